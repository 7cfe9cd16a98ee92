"""Randomised soak of the map store (pp_set_map / pp_set_egos / k_resolve_map) against the numpy resolver of
tests/map_scenes.py and the oracle (test infrastructure).  Usage (GPU box): python tests/soak_map.py [seconds] [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch                                   # noqa: F401
torch.cuda.is_available()
import dmpp_amd as dm
import oracle_binding
import map_scenes as ms
from parity_util import compare


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    orc = oracle_binding.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    rng = np.random.default_rng(seed0)
    t0, it, bad_total, ticks = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        grid = int(rng.choice([64, 128, 256]))
        cfg = dm.default_config(grid)
        cfg["dynamic_obstacles"] = int(rng.integers(0, 2))
        n_roads, mseed, eseed = int(rng.integers(2, 9)), int(rng.integers(0, 1 << 30)), int(rng.integers(0, 1 << 30))
        n, n_obs = int(rng.choice([1, 8, 96, 300])), int(rng.choice([0, 4, 16, 40]))
        m = ms.build_map(dm, n_roads=n_roads, seed=mseed)
        sc = ms.make_egos(dm, cfg, m, n, n_obs, seed=eseed)
        pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=max(n * n_obs, 1), max_lane_pts_total=len(m["points"]),
                        max_ref_pts_total=max(len(m["jpoints"]), 1))
        pl.set_map(m)
        pl.set_egos(sc)
        want = ms.resolve(dm, m, sc["scene_in"])
        bad = compare(pl.get_scene_in(), want, "scene_in")
        sc_o = dict(sc)
        sc_o["scene_in"] = want
        st_o = sc["state"].copy()
        pl.set_state(sc["state"])
        n_ticks = int(rng.integers(1, 5))
        for t in range(n_ticks):
            pl.tick(sync=bool(rng.integers(0, 2)))
            plan_o, gout_o, _ = orc.plan_tick_batch(cfg, sc_o, st_o, n_threads=16, want_grid=True)
        pl.sync()
        gout_g = pl.get_grid_out()
        bad += compare(pl.get_plan(), plan_o, "plan") + compare(pl.get_state(), st_o, "state")
        bad += compare(gout_g["status"], gout_o["status"], "grid.status")
        keep = (gout_o["status"] != 3) & (gout_o["status"] != 7)
        bad += compare(gout_g[keep], gout_o[keep], "grid")
        pl.close()
        it += 1
        ticks += n * n_ticks
        tag = f"it {it} grid {grid} roads {n_roads} mseed {mseed} eseed {eseed} n {n} obs {n_obs} ticks {n_ticks} dyn {int(cfg['dynamic_obstacles'][0])}"
        if bad:
            bad_total += 1
            print("MISMATCH", tag, bad[:6], flush=True)
        elif it % 10 == 0:
            print(f"ok {tag} [{ticks} scene-ticks, {time.time() - t0:.0f} s]", flush=True)
    print(f"MAP SOAK DONE iterations {it} scene-ticks {ticks} mismatching batches {bad_total}", flush=True)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
