#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot host-buffer call pp_plan_tick_batch (upload + tick + download),
for DESIGN.md §7 — never the headline `value` of bench.py, whose inputs are resident in HBM."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dmpp_amd as dm

n, n_obs = 1024, 64
cfg = dm.default_config(512)
sc = dm.gen_scenes(cfg, 0, n, n_obs, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * n_obs)
st = sc["state"].copy()
for _ in range(3):
    pl.plan_tick_batch(sc, st)
t0 = time.perf_counter()
K = 20
for _ in range(K):
    pl.plan_tick_batch(sc, st)
dt = (time.perf_counter() - t0) / K
up = sum(sc[k].nbytes for k in ("scene_in", "lane_pool", "attr_pool", "ref_pool", "obs_pool", "mot_pool")) + st.nbytes
down = n * (dm.PlanOut.itemsize + dm.SceneState.itemsize + dm.GridOut.itemsize)
print(f"host-buffer tick: {dt*1e3:.3f} ms for {n} scenes = {n/dt:.0f} ticks/s ; {up/1e6:.1f} MB up + {down/1e6:.1f} MB down per tick")
