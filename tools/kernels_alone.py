#!/usr/bin/env python3
"""Runs ON THE GPU BOX: every kernel of the tick timed with nothing beside it (one tick, host wait, next tick), and the same in
the pipelined tick - what the kernels cost each other.
    gpurun -- 'python tools/kernels_alone.py [scenes] [n_obs] [dynamic]'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dmpp_amd as dm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_obs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dyn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = dm.default_config(512)
cfg["dynamic_obstacles"] = dyn
cfg["force_replan"] = dyn
sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=8)
pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
pl.set_scenes(sc)
pl.set_state(sc["state"])
for _ in range(6):
    pl.tick()
pl.sync()
pl.set_profile(1)
for mode in ("alone", "pipelined"):
    pl.reset_kernel_ms()
    for _ in range(30):
        pl.tick(sync=(mode == "alone"))
    pl.sync()
    print("%-10s" % mode, " ".join("%s=%.3f" % (k.replace("k_", ""), v[0] / max(v[1], 1)) for k, v in pl.kernel_ms().items() if v[1] > 0), flush=True)
