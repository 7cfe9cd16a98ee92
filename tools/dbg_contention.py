"""How much slower a scene's search runs inside the 1024-scene batch than alone on the GPU (debug build)."""
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))
cfg = dm.default_config(512)
mp = int(cfg['max_path'][0])
n = 1024
sc = dm.gen_scenes(cfg, 0, n, 64, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * 64)
pl.set_scenes(sc); pl.set_state(sc['state'])
for _ in range(4):
    pl.tick(sync=True)
batch = np.array([pl.get_path(s, mp)[-16:] for s in range(n)])
g = pl.get_grid_out()
pl.close()
heavy = np.argsort(-batch[:, 13])[:6].tolist()
for s in heavy + [0, 1, 2, 3]:
    one = dm.gen_scenes(cfg, s, 1, 64, 8)
    p1 = dm.Planner(cfg, max_scenes=1, max_obs_total=64)
    p1.set_scenes(one); p1.set_state(one['state'])
    for _ in range(3):
        p1.tick(sync=True)
    alone = p1.get_path(0, mp)[-16:]
    p1.close()
    print("scene %4d expanded %4d steps %3d: kernel cycles alone %8d, in the batch %8d  (x%.2f)" %
          (s, int(g['n_expanded'][s]), int(batch[s, 2]), int(alone[13]) * 16, int(batch[s, 13]) * 16, batch[s, 13] / max(alone[13], 1)))
