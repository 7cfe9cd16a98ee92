
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import os
import numpy as np, dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))   # make -C .../csrc debug
cfg = dm.default_config(512)
n=1024
N_OBS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DYN = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg["dynamic_obstacles"] = DYN; cfg["force_replan"] = DYN
sc = dm.gen_scenes(cfg, 0, n, N_OBS, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n*N_OBS)
pl.set_scenes(sc); pl.set_state(sc['state']); pl.tick(sync=True); pl.tick(sync=True)
g = pl.get_grid_out()
mp=int(cfg['max_path'][0])
rows=[]
setup=[]
for s in range(n):
    pp_=pl.get_path(s, mp)
    p=pp_[-16:]
    setup.append(pp_[-32:-24].tolist())
    rows.append([int(g['n_expanded'][s]), int(g['n_pushed'][s])]+p[:16].tolist())
rows=np.array(rows)[g['status']==0]
names=['n_exp','n_push','iter','popped','jump_iters','diag_jobs','rounds','cyc_pop','cyc_closed','cyc_cand','cyc_jump','cyc_push','cyc_walk','cyc_search','cyc_setup','cyc_kernel','hw_id','xcc_id']
print(names, '(cycles/16)')
idx=np.argsort(-rows[:,15])[:6]
for i in idx: print(rows[i].tolist())
print('median', np.median(rows,axis=0).astype(int).tolist())
print('sum', rows.sum(axis=0).tolist())

hw = rows[:,16].astype(np.int64) & 0xFFFFFFFF
simd = (hw >> 4) & 3
print('searching waves per SIMD id:', np.bincount(simd, minlength=4).tolist(), ' wave slot ids:', np.bincount(hw & 15, minlength=16).tolist())

setup=np.array(setup)[g['status']==0]
print('setup phases (cycles) entry, clear, footprints, pass1, offsets, zero, pass2, tail: median', np.median(setup,axis=0).astype(int).tolist(), 'max', setup.max(axis=0).tolist())
