import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, dmpp_amd as dm, oracle_binding
O = oracle_binding.Oracle('oracle/liboracle.so')
cfg = dm.default_config(512)
n=96
sc = dm.gen_scenes(cfg, 1000, n, 1, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n, order_cap=512*512)
pl.set_scenes(sc); pl.set_state(sc['state']); pl.tick(sync=True)
g = pl.get_grid_out()
st=sc['state'].copy()
po,go,_ = O.plan_tick_batch(cfg, sc, st, n_threads=8)
print('gpu status', np.bincount(g['status'],minlength=7), 'oracle', np.bincount(go['status'],minlength=7))
bad=[s for s in range(n) if g['n_expanded'][s]!=go['n_expanded'][s] or g['order_digest'][s]!=go['order_digest'][s] or g['status'][s]!=go['status'][s]]
print('bad scenes', bad[:20], len(bad))
for s in bad[:3]:
    ne=int(go['n_expanded'][s]); og=pl.get_order(s, ne)
    st1=sc['state'].copy(); _,go1,grid,oo,pp=O.plan_tick_one(cfg, sc, s, st1, order_cap=512*512)
    k=int(np.argmax(og!=oo)) if (og!=oo).any() else -1
    print(s,'gpu', g['status'][s], g['n_expanded'][s], g['n_pushed'][s],'orc', go['status'][s], ne, go['n_pushed'][s], 'first diff at', k, og[max(k-2,0):k+3], oo[max(k-2,0):k+3])

for s in bad[:3]:
    pth = pl.get_path(s, 232)
    print('scene', s, 'n_exp,b,cb,gb,first,fcur,om,avail =', pth[:8].tolist())
    print('  cnts', pth[8:24].tolist()); print('  gcn ', pth[24:40].tolist())
    print('  win[b]', [hex(int(v) & 0xffffffff) for v in pth[40:104]])
    print('  all rows first4', [hex(int(v) & 0xffffffff) for v in pth[104:168]])
    print('  readback', [hex(int(v) & 0xffffffff) for v in pth[168:232]][:12])
