
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import os
import numpy as np, dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))   # make -C .../csrc debug
cfg = dm.default_config(512)
n=1024
sc = dm.gen_scenes(cfg, 0, n, 64, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n*64)
pl.set_scenes(sc); pl.set_state(sc['state']); pl.tick(sync=True); pl.tick(sync=True)
g = pl.get_grid_out()
mp=int(cfg['max_path'][0])
rows=[]
for s in range(n):
    p=pl.get_path(s, mp)[-16:]
    rows.append([int(g['n_expanded'][s]), int(g['n_pushed'][s])]+p[:16].tolist())
rows=np.array(rows)[g['status']==0]
names=['n_exp','n_push','iter','jump_iters','jobs','passes','cyc_nz','cyc_pop','cyc_closed','cyc_cand','cyc_jump','cyc_push','cyc_walk','cyc_total','cyc_setup','cyc_kernel','cyc_pack','cyc_transpose']
print(names, '(cycles/16)')
idx=np.argsort(-rows[:,15])[:6]
for i in idx: print(rows[i].tolist())
print('median', np.median(rows,axis=0).astype(int).tolist())
print('sum', rows.sum(axis=0).tolist())
