# Finer phase split of the search step than tools/dbg_timing.py: needs the debug build (make -C .../csrc debug), whose extra
# marks (DBG_MARK 7..11 in kernels_s.hpp) stamp job hand-over, line descriptors, cell tests, scans and successor rules.
#   gpurun -- 'python tools/dbg_fine.py'
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import os
import numpy as np, dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))
cfg = dm.default_config(512)
n=1024
sc = dm.gen_scenes(cfg, 0, n, 64, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n*64)
pl.set_scenes(sc); pl.set_state(sc['state']); pl.tick(sync=True); pl.tick(sync=True)
g = pl.get_grid_out()
mp=int(cfg['max_path'][0])
rows=[]
for s in range(n):
    pp_=pl.get_path(s, mp)
    rows.append(pp_[-16:].tolist()+pp_[-64+7:-64+16].tolist())
rows=np.array(rows)[g['status']==0]
names=['iter','popped','jump_iters','diag_jobs','rounds','pop','probe','cand','results','push','walk','sum','setup','kernel','hw','xcc','post','jobread+metas','celltests','jump_lane','rules','t12','t13','t14','t15']
idx=np.argsort(-rows[:,13])[:4]
for i in idx: print(dict(zip(names, rows[i].tolist())))
tot=rows.sum(axis=0)
it=tot[0]
print('per step (cycles):', {k: round(16*v/it) for k,v in zip(names,tot) if k not in ('hw','xcc','iter')})
