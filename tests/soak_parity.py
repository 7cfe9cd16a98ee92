"""Randomised soak of the HIP path against the CPU oracle (test infrastructure): many seeds, grid sizes,
obstacle counts, static / dynamic obstacles, moving egos, several ticks per batch.  Every PlanOut / SceneState /
GridOut field is compared exactly as the GPU tests do.  Usage (GPU box): python tests/soak_parity.py [seconds] [seed0].  Lives under tests/ because it loads the oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if os.environ.get("SOAK_TORCH"):               # the library on the HIP runtime bundled with torch (what a caller that imports torch first gets)
    import torch                               # noqa: F401
    torch.cuda.is_available()
import dmpp_amd as dm
import oracle_binding
from parity_util import compare, move_ego


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    orc = oracle_binding.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    rng = np.random.default_rng(seed0)
    t0, it, scenes_done, bad_total = time.time(), 0, 0, 0
    threads = 32
    while time.time() - t0 < budget:
        grid = int(rng.choice([64, 128, 256, 512, 512, 1024, 2048] if seed0 % 2 == 0 else [64, 128, 256, 512, 512, 1024]))
        if grid <= 512:
            n = int(rng.choice([1, 3, 17, 64, 200, 256, 300, 700]))
        elif grid == 1024:
            n = int(rng.choice([2, 9, 40, 260]))          # 260: consecutive searches on two streams
        else:
            n = int(rng.choice([2, 5]))
        n_obs = int(rng.choice([0, 1, 8, 24, 64, 64, 130, 256, 520]))
        if n * n_obs > 120000:
            n_obs = 64
        dynamic = int(rng.integers(0, 2))
        first = int(rng.integers(0, 1 << 20))
        jevery = int(rng.choice([0, 3, 8]))
        n_ticks = int(rng.integers(1, 6))
        gw = gh = grid
        if seed0 >= 5 and grid <= 512:                      # seeds from 5 on: any width / height in multiples of 32
            gw, gh = int(rng.integers(1, 21)) * 32, int(rng.integers(1, 21)) * 32
        cfg = dm.default_config(gw, gh)
        cfg["dynamic_obstacles"] = dynamic
        cfg["force_replan"] = int(rng.integers(0, 2))
        cfg["decision_stage"] = int(rng.integers(0, 4) != 0)
        cfg["lanechg_stage"] = int(rng.integers(0, 4) != 0)
        if seed0 >= 14:                                     # seeds from 14 on: the limits and weights of the grid engine as well
            if rng.integers(0, 4) == 0:
                cfg["max_expansions"] = int(rng.choice([1, 7, 40, 100, 300]))
            if rng.integers(0, 4) == 0:
                cfg["bucket_cap"] = int(rng.choice([16, 24, 40, 100]))
            if rng.integers(0, 4) == 0:
                cfg["max_path"] = int(rng.choice([2, 17, 100, 250]))
            if rng.integers(0, 3) == 0:
                cfg["n_lattice"] = int(rng.integers(0, 17))
            if rng.integers(0, 3) == 0:
                cfg["lookahead_cells"] = int(rng.choice([1, 5, 40, 120, 199, 400]))
            if rng.integers(0, 4) == 0:
                cfg["inflate"] = float(rng.choice([0.0, 0.3, 1.1]))
            if rng.integers(0, 4) == 0:
                cfg["d_safe"] = float(rng.choice([0.2, 1.0, 3.0]))
            if rng.integers(0, 4) == 0:
                cfg["ID_MORE"] = int(rng.integers(0, 6))
        sync_each = bool(rng.integers(0, 2))
        sc = dm.gen_scenes(cfg, first, n, n_obs, junction_every=jevery)
        if gw != gh:                                        # the generator spreads goals over the width
            sc["scene_in"]["goal"]["y"] = np.clip(sc["scene_in"]["goal"]["y"], 0.0, gh * float(cfg["cell"][0]) - 0.01)
        if seed0 >= 14 and rng.integers(0, 3) == 0:         # egos (and goals) outside their grids: clamped start / goal cells
            k = len(sc["scene_in"])
            sc["scene_in"]["grid_origin"]["x"] += rng.choice([0.0, 0.0, 7.0, -9.0, 30.0], size=k)
            sc["scene_in"]["grid_origin"]["y"] += rng.choice([0.0, 0.0, 5.0, -12.0, -40.0], size=k)
        if seed0 >= 18 and rng.integers(0, 2) == 0:         # ragged road inputs on a random subset of the scenes (cf. test_edge_cases)
            si = sc["scene_in"]
            k = len(si)
            pick = lambda frac: rng.random(k) < frac
            m = pick(0.15); si["lanes"]["lane_width"][m] = 7.0
            m = pick(0.3); si["lanes"]["lanechg_attribute"][m] = rng.integers(0, 4, size=int(m.sum()))
            m = pick(0.1); si["ref_n"][m] = 2; si["dec"]["refpath_n"][m] = 2
            m = pick(0.1); si["ref_n"][m] = 0; si["dec"]["refpath_n"][m] = 0
            m = pick(0.1); si["loc"]["id"][m] = dm.GEN_LANE_PTS - 2
            m = pick(0.05); si["loc"]["id"][m] = dm.GEN_LANE_PTS + 5
            m = pick(0.05); si["lanes"]["cur_n"][m] = 1
            m = pick(0.1); si["lanes"]["left_n"][m] = 0
            m = pick(0.1); si["lanes"]["right_n"][m] = 0
            m = pick(0.1); si["obs_n"][m] = 0
            m = pick(0.1); si["obs_n"][m] = np.minimum(si["obs_n"][m], 3)
            m = pick(0.1); si["loc"]["velocity"][m] = rng.choice([0.0, 0.5, 35.0], size=int(m.sum()))
        if rng.integers(0, 3) == 0:
            sc["scene_in"]["period_last"] = float(rng.choice([100.0, 900.0, 1700.0]))
        one_shot = bool(rng.integers(0, 3) == 0)            # host buffers through pp_plan_tick_batch, state carried by the caller
        streamed = seed0 >= 40 and not one_shot and bool(rng.integers(0, 3) == 0)   # seeds from 40 on: pp_update_async / pp_fetch_async, every tick compared
        if seed0 >= 40 and rng.integers(0, 12) == 0 and gw == gh and grid in (256, 512, 1024):
            # a field of small discs behind walls: thousands of live open-list entries (the spill area of the open list)
            from grid_scenes import pebble_field
            cfg["bucket_cap"] = int(rng.choice([700, 2000, 16384]))
            cfg["max_path"] = 32768
            sc = pebble_field(dm, cfg, n_side=int(rng.integers(8, grid // 24)), radius=float(rng.choice([0.05, 0.2, 0.6])),
                              jitter=float(rng.choice([0.5, 2.0])), seed=int(rng.integers(0, 1 << 20)), walls=int(rng.integers(1, 4)))
            n, n_obs, dynamic = 1, sc["n_obs"], 0
            cfg["dynamic_obstacles"] = 0
            sc["n_obs"] = n_obs
        pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=max(n * n_obs, 1))
        pl.set_state(sc["state"])
        st_o = sc["state"].copy()
        st_h = sc["state"].copy()
        bad = []
        fetched = []
        if streamed:
            pl.set_scenes(sc)
        for t in range(n_ticks):
            if t and rng.integers(0, 2):
                move_ego(sc, int(rng.integers(1, 9)), dlat=float(rng.choice([0.0, 0.2, -0.4])))
            if streamed:
                if n_obs:
                    sc["obs_pool"]["x"] += rng.uniform(-0.2, 0.2, len(sc["obs_pool"]))
                a, b = dm.pinned_copy(sc["scene_in"]), dm.pinned_copy(sc["obs_pool"])
                mo = dm.pinned_copy(sc["mot_pool"])
                pl.update_async(a, b, mo, n_obs_total=n * n_obs)
                pl.tick()
                pg, gg = dm.pinned_empty(n, dm.PlanOut), dm.pinned_empty(n, dm.GridOut)
                tid = pl.fetch_async(pg, gg)
                plan_o, gout_o, _ = orc.plan_tick_batch(cfg, sc, st_o, n_threads=threads, want_grid=True)
                fetched.append((tid, pg, gg, plan_o, gout_o, a, b, mo))
                if t == n_ticks - 1:
                    for (tid, pg, gg, po_, go_, _, _, _) in fetched:
                        pl.wait_tick(tid)
                        bad += compare(pg, po_, "plan") + compare(gg["status"], go_["status"], "grid.status")
                        keep = (go_["status"] != 3) & (go_["status"] != 7)
                        bad += compare(gg[keep], go_[keep], "grid")
                    pl.sync()
                    bad += compare(pl.get_state(), st_o, "state")
                continue
            if one_shot:
                plan_g, gout_g = pl.plan_tick_batch(sc, st_h)
            else:
                pl.set_scenes(sc)
                pl.tick(sync=sync_each)
            plan_o, gout_o, _ = orc.plan_tick_batch(cfg, sc, st_o, n_threads=threads, want_grid=True)
            if one_shot:
                bad += compare(plan_g, plan_o, "plan") + compare(st_h, st_o, "state")
                bad += compare(gout_g["status"], gout_o["status"], "grid.status")
                keep = (gout_o["status"] != 3) & (gout_o["status"] != 7)
                bad += compare(gout_g[keep], gout_o[keep], "grid")
            elif sync_each or t == n_ticks - 1:
                pl.sync()
                plan_g, st_g, gout_g = pl.get_plan(), pl.get_state(), pl.get_grid_out()
                bad += compare(plan_g, plan_o, "plan") + compare(st_g, st_o, "state")
                bad += compare(gout_g["status"], gout_o["status"], "grid.status")
                keep = (gout_o["status"] != 3) & (gout_o["status"] != 7)
                bad += compare(gout_g[keep], gout_o[keep], "grid")
        pl.close()
        it += 1
        scenes_done += n * n_ticks
        tag = f"it {it} grid {gw}x{gh} n {n} obs {n_obs} dyn {dynamic} first {first} jevery {jevery} ticks {n_ticks} sync {int(sync_each)} oneshot {int(one_shot)} streamed {int(streamed)} cap {int(cfg['bucket_cap'][0])} " \
              f"dec {int(cfg['decision_stage'][0])} lc {int(cfg['lanechg_stage'][0])} force {int(cfg['force_replan'][0])}"
        if bad:
            bad_total += 1
            print("MISMATCH", tag, bad[:6], flush=True)
        elif it % 10 == 0:
            print(f"ok {tag}  [{scenes_done} scene-ticks, {time.time() - t0:.0f} s]", flush=True)
    print(f"SOAK DONE iterations {it} scene-ticks {scenes_done} mismatching batches {bad_total}", flush=True)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
