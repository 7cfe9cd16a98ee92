import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, dmpp_amd as dm, time
cfg = dm.default_config(512)
for first in (0, 1, 2, 3):
    pl = dm.Planner(cfg, max_scenes=1, max_obs_total=64)
    sc = dm.gen_scenes(cfg, first, 1, 64, junction_every=0)
    pl.set_scenes(sc); pl.set_state(sc["state"])
    for _ in range(10): pl.tick(sync=True)
    lat=[]
    for _ in range(200):
        a=time.perf_counter(); pl.tick(sync=True); lat.append((time.perf_counter()-a)*1e3)
    pl.set_profile(True); pl.reset_kernel_ms()
    for _ in range(50): pl.tick(sync=True)
    k = pl.kernel_ms()
    g = pl.get_grid_out()
    print("scene", first, "p50 %.3f ms" % np.percentile(lat,50), "n_exp", int(g["n_expanded"][0]), {n: round(v[0]/max(v[1],1),4) for n,v in k.items()})
    pl.close()
