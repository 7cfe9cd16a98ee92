"""Launch timeline of k_search (debug build, `make -C .../csrc debug`): when each scene's wave started and ended
(100 MHz wall clock), in launch order, for the 1024-scene benchmark workload."""
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dmpp_amd as dm
dm.load_library(os.path.join(os.path.dirname(dm.LIB_PATH), 'libdmpp_dbg.so'))
cfg = dm.default_config(512)
n = 1024
sc = dm.gen_scenes(cfg, 0, n, 64, 8)
pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * 64)
pl.set_scenes(sc); pl.set_state(sc['state'])
for _ in range(4):
    pl.tick(sync=True)
mp = int(cfg['max_path'][0])
rows = np.array([pl.get_path(s, mp)[-16:] for s in range(n)])
t0, t1, cyc = rows[:, 14].astype(np.int64), rows[:, 15].astype(np.int64), rows[:, 13].astype(np.int64) * 16
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
order = np.argsort(t0)
print("kernel span %.1f us; last start %.1f us; starts: p50 %.1f p90 %.1f" % (t1.max(), t0.max(), np.percentile(t0, 50), np.percentile(t0, 90)))
dur = t1 - t0
print("durations us: max %.1f p99 %.1f p90 %.1f p50 %.1f mean %.1f; sum/512 = %.1f" % (dur.max(), np.percentile(dur, 99), np.percentile(dur, 90), np.percentile(dur, 50), dur.mean(), dur.sum() / 512))
late = np.argsort(-t1)[:8]
for s in late:
    print("scene %4d start %.1f end %.1f dur %.1f cycles %d  (MHz %.0f)" % (s, t0[s], t1[s], dur[s], cyc[s], cyc[s] / max(dur[s], 1e-9)))
print("scenes started after 100 us:", int((t0 > 100).sum()), " first-wave (start < 20 us):", int((t0 < 20).sum()))
for t in (5, 20, 50, 100, 150, 200, 300, 400, 500):
    print("t=%3d us: running %4d  started %4d  finished %4d" % (t, int(((t0 <= t) & (t1 > t)).sum()), int((t0 <= t).sum()), int((t1 <= t).sum())))
