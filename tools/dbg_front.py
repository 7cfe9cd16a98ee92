"""Phase timing of k_decision / k_planning (debug build: make -C .../csrc debug): cycle stamps per scene, alone on the GPU."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))
cfg = dm.default_config(512)
cfg["grid_stage"] = 0
n = 1024
sc = dm.gen_scenes(cfg, 0, n, 64, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * 64)
pl.set_scenes(sc); pl.set_state(sc['state'])
for _ in range(3):
    pl.tick(sync=True)
D, P = [], []
for s in range(n):
    r = pl.get_refpath(s, 512)
    D.append([r["x"][480], r["y"][480], r["x"][481], r["y"][481], r["x"][482], r["y"][482]])
    P.append([r["x"][490], r["y"][490], r["x"][491], r["y"][491], r["x"][492], r["y"][492]])
D, P = np.array(D), np.array(P)
pos = sc["scene_in"]["loc"]["pos"]
attr = sc["scene_in"]["lanes"]["lanechg_attribute"]
for name, sel in (("road, no lane change", (pos == 0) & (attr == 0)), ("road, lane change allowed", (pos == 0) & (attr != 0)), ("junction", pos != 0)):
    print(name, int(sel.sum()), "scenes")
    print("  k_decision stamps (cycles): stage obs, LoadRefPath, AroundObstacle, sweep, rem-length, tree, end: median", np.median(D[sel], axis=0).astype(int).tolist())
    print("  k_planning stamps (cycles): aim, initial, localise, path, SearchObstacle, end: median", np.median(P[sel], axis=0).astype(int).tolist())
