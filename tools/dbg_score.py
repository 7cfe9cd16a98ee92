"""Runs ON THE GPU BOX (debug build): how many obstacles survive the scoring pass's cull, and how often the bucket grid is used.
    gpurun -- 'python tools/dbg_score.py [obstacles] [dynamic]'"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import dmpp_amd as dm
lib = dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))
n_obs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dyn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = dm.default_config(512)
cfg["dynamic_obstacles"] = dyn
cfg["force_replan"] = dyn
n = 1024
sc = dm.gen_scenes(cfg, 0, n, n_obs, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * n_obs)
pl.set_scenes(sc); pl.set_state(sc['state'])
out = (C.c_int * 8)()
lib.pp_debug_score_counters(out, 1)
for t in range(10):
    pl.tick(sync=True)
    lib.pp_debug_score_counters(out, 1)
    v = list(out)
    print("tick %d: scenes %d  mean n_rel %.1f  max %d  bucketed %d  bucket overflow %d  not culled %d" % (t, v[0], v[1] / max(v[0], 1), v[5], v[2], v[3], v[4]))
