import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import os
import numpy as np, dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))
cfg = dm.default_config(512)
names=['n_exp','n_push','iter','jump_iters','jobs','passes','cyc_nz','cyc_pop','cyc_closed','cyc_cand','cyc_jump','cyc_push','cyc_walk','cyc_total','cyc_setup','cyc_kernel','cyc_pack','cyc_tr']
print(names)
for first in (0,1,2,3):
    sc = dm.gen_scenes(cfg, first, 1, 64, 0)
    pl = dm.Planner(cfg, max_scenes=1, max_obs_total=64)
    pl.set_scenes(sc); pl.set_state(sc['state']); pl.tick(sync=True); pl.tick(sync=True)
    g = pl.get_grid_out(); mp=int(cfg['max_path'][0])
    p=pl.get_path(0, mp)[-16:]
    print(first, [int(g['n_expanded'][0]), int(g['n_pushed'][0])]+p[:16].tolist())
    pl.close()
