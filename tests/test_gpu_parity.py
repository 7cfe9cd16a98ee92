"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded scenes.

Bar (BASELINE.json north_star): integer outputs bit-exact — ids, flags, behaviour codes, replan
cause, occupancy grid, node-expansion order, digest, counters, chosen path; floats within 1e-6
relative.  PARITY UNPINNED versus the reference itself: it ships no tests or golden vectors and
cannot be built here (see oracle/dmpp_oracle.h); the oracle is a line-cited restatement.
"""
import os

import numpy as np
import pytest

from parity_util import compare, move_ego, bit_identical_fraction

pytestmark = pytest.mark.gpu


def _tick_both(dm, oracle, cfg, sc, n_ticks=1, order_cap=0, mutate=None, with_motion=True):
    n = len(sc["scene_in"])
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=max(n * sc["n_obs"], 1), order_cap=order_cap)
    st_o = sc["state"].copy()
    pl.set_state(sc["state"])
    res = []
    for t in range(n_ticks):
        if mutate is not None:
            mutate(sc, t)
        pl.set_scenes(sc, with_motion=with_motion)
        pl.tick(sync=True)
        plan_g, st_g = pl.get_plan(), pl.get_state()
        gout_g = pl.get_grid_out() if cfg["grid_stage"][0] else None
        sc_o = dict(sc)
        if not with_motion:
            sc_o["mot_pool"] = None
        plan_o, gout_o, grids_o = oracle.plan_tick_batch(cfg, sc_o, st_o, n_threads=8, want_grid=bool(cfg["grid_stage"][0]),
                                                         keep_grids=bool(cfg["grid_stage"][0]) and n <= 64)
        res.append((plan_g, st_g, gout_g, plan_o, st_o.copy(), gout_o, grids_o))
    return pl, res


def _assert_tick(res, tag=""):
    plan_g, st_g, gout_g, plan_o, st_o, gout_o, _ = res
    bad = compare(plan_g, plan_o, "plan") + compare(st_g, st_o, "state")
    if gout_g is not None:
        bad += compare(gout_g["status"], gout_o["status"], "grid.status")
        keep = (gout_o["status"] != 3) & (gout_o["status"] != 7)   # DMPP_G_OVERFLOW / COST_RANGE: only the status is specified
        bad += compare(gout_g[keep], gout_o[keep], "grid")
    assert not bad, tag + "\n" + "\n".join(bad[:20])


# ---- stand-alone operators (rows R6, R12-R15) ---------------------------------------------------
def test_geom_ops(dm, oracle):
    cfg = dm.default_config(128)
    pl = dm.Planner(cfg, max_scenes=1)
    rng = np.random.default_rng(1)
    n = 4096
    a, b, c = (np.zeros(n, dm.GlobalPoint2D) for _ in range(3))
    for arr in (a, b, c):
        arr["x"] = rng.uniform(-50, 50, n)
        arr["y"] = rng.uniform(-50, 50, n)
    # degenerate cases of the EPSILON branches (Planning.cpp:690,722,724)
    b["x"][:64] = a["x"][:64]
    b["y"][32:64] = a["y"][32:64]
    c["x"][64:96] = b["x"][64:96] + 1e-7
    lat = pl.geom_batch(0, a, b, c)
    ang = pl.geom_batch(1, a, b)
    d = np.zeros(n, dm.GlobalPoint2D)
    d["x"] = rng.uniform(0, 360, n)
    d["y"] = rng.uniform(0, 360, n)
    err = pl.geom_batch(2, d)
    for i in range(n):
        ai, bi, ci_ = (a["x"][i], a["y"][i]), (b["x"][i], b["y"][i]), (c["x"][i], c["y"][i])
        assert lat[i] == oracle.GetLatDis(cfg, ai, bi, ci_)                     # +,-,*,/,sqrt only: bit-exact
        assert abs(ang[i] - oracle.GetRoadAngle(cfg, ai, bi)) <= 1e-6 * max(1.0, abs(ang[i]))   # atan: 1e-6
        assert err[i] == oracle.GetAngleErr(d["x"][i], d["y"][i])


def test_helper_ops(dm, oracle):
    cfg = dm.default_config(128)
    pl = dm.Planner(cfg, max_scenes=1)
    rng = np.random.default_rng(2)
    for _ in range(20):
        s = (rng.uniform(0, 100), rng.uniform(0, 100), rng.uniform(0, 360))
        e = (rng.uniform(0, 100), rng.uniform(0, 100), rng.uniform(0, 360))
        assert not compare(pl.bezier(s, e), oracle.BezierPlanning(cfg, s, e), "bezier")
        n_in = int(rng.integers(0, 300))
        pts = np.zeros(n_in, dm.GlobalPoint2D)
        pts["x"] = np.cumsum(rng.uniform(0.0, 1.0, n_in))
        pts["y"] = np.cumsum(rng.uniform(-0.5, 0.5, n_in))
        assert not compare(pl.mean_points(pts), oracle.MeanPoints(cfg, pts), "mean")
        if n_in:
            off = float(rng.uniform(-4, 4))
            g, o = pl.create_new_path(pts, off), oracle.CreateNewPath(cfg, pts, off)
            assert g.tobytes() == o.tobytes()                                    # no transcendental: bit-exact
    # degenerate inputs: repeated points, single point, empty
    rep = np.zeros(5, dm.GlobalPoint2D)
    assert not compare(pl.mean_points(rep), oracle.MeanPoints(cfg, rep), "mean-rep")
    assert not compare(pl.mean_points(rep[:1]), oracle.MeanPoints(cfg, rep[:1]), "mean-1")
    assert not compare(pl.mean_points(rep[:0]), oracle.MeanPoints(cfg, rep[:0]), "mean-0")
    assert pl.create_new_path(rep, 1.0).tobytes() == oracle.CreateNewPath(cfg, rep, 1.0).tobytes()


def test_search_obstacle_batch(dm, oracle):
    cfg = dm.default_config(128)
    pl = dm.Planner(cfg, max_scenes=1)
    rng = np.random.default_rng(3)
    paths, obs, poff, ooff, lo, hi = [], [], [0], [0], [], []
    for q in range(200):
        n = int(rng.choice([0, 1, 2, 3, 40, 120, 200, 513]))
        m = int(rng.choice([0, 1, 7, 64, 65, 256]))
        p = np.zeros(n, dm.GlobalPoint2D)
        p["x"] = np.cumsum(rng.uniform(0.2, 0.8, n))
        p["y"] = 3 * np.sin(p["x"] / 9.0)
        o = np.zeros(m, dm.ObPoint)
        o["x"] = rng.uniform(-10, (p["x"][-1] if n else 10) + 10, m)
        o["y"] = rng.uniform(-6, 6, m)
        o["radius"] = 1
        if m > 2 and n > 2:      # exact ties: two obstacles at the same place, one exactly on a path point
            o[1] = o[0]
            o["x"][2], o["y"][2] = p["x"][n // 2], p["y"][n // 2]
        paths.append(p); obs.append(o)
        poff.append(poff[-1] + n); ooff.append(ooff[-1] + m)
        w = float(rng.choice([0.9, 1.1, 1.875]))
        lo.append(-w); hi.append(float(rng.choice([0.9, 1.875])))
    P, Ob = np.concatenate(paths), np.concatenate(obs)
    out = pl.search_obstacle_batch(P, np.array(poff, np.int32), Ob, np.array(ooff, np.int32), np.array(lo), np.array(hi))
    for q in range(200):
        r = oracle.SearchObstacle(cfg, paths[q], obs[q], lo[q], hi[q])
        g = out[q]
        assert int(g["Obs_flag"]) == r["flag"] and int(g["Ob_Pathid"]) == r["path_id"], q
        assert g["Ob_Pose"]["dis_lat"] == r["dis_lat"] and g["Ob_Pose"]["dis_lng"] == r["dis_lng"], q   # bit-exact
        assert g["Ob_Attr"].tobytes() == r["ob"].tobytes(), q


# ---- whole tick -------------------------------------------------------------------------------------
def test_config0_single_scene_128(dm, oracle):
    """BASELINE configs[0]: single 128x128 grid, 8 static obstacles, one ego."""
    cfg = dm.default_config(128)
    sc = dm.gen_scenes(cfg, 0, 1, 8, junction_every=0)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=3, order_cap=128 * 128)
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")
    st = sc["state"].copy()
    for _ in range(3):
        _, go, grid_o, order_o, path_o = oracle.plan_tick_one(cfg, sc, 0, st, order_cap=128 * 128)
    assert (pl.get_grid(0) == grid_o).all()
    assert (pl.get_order(0, int(go["n_expanded"])) == order_o).all()
    assert (pl.get_path(0, int(go["path_len"])) == path_o).all()


@pytest.mark.parametrize("n_obs", [0, 1, 64, 256])
def test_batch_512(dm, oracle, n_obs):
    """512x512, mixed road / pre-junction / junction scenes, every lane-change attribute."""
    cfg = dm.default_config(512)
    n = 96
    sc = dm.gen_scenes(cfg, 1000, n, n_obs, junction_every=8)
    cap = 512 * 512
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=1, order_cap=cap)
    _assert_tick(res[0], f"n_obs={n_obs}")
    plan_g, st_g, gout_g, plan_o, st_o, gout_o, grids = res[0]
    st = sc["state"].copy()
    for s in range(0, n, 7):                      # full expansion order, grid and path of a sample of scenes
        st1 = sc["state"].copy()
        _, go, grid_o, order_o, path_o = oracle.plan_tick_one(cfg, sc, s, st1, order_cap=cap)
        assert (pl.get_grid(s) == grid_o).all(), s
        assert (pl.get_order(s, int(go["n_expanded"])) == order_o).all(), s
        assert (pl.get_path(s, int(go["path_len"])) == path_o).all(), s
    if n_obs == 64:
        print("byte-identical fraction plan/grid:", bit_identical_fraction(plan_g, plan_o), bit_identical_fraction(gout_g, gout_o))


def test_multi_tick_state_and_replan_causes(dm, oracle):
    """State carried over ticks while the ego moves: exercises replan causes 1-4, the obstacle
    counters of the sweep (Decision.cpp:915-917,936,998) and the stale-aim quirk."""
    cfg = dm.default_config(512)
    sc = dm.gen_scenes(cfg, 5000, 64, 64, junction_every=8)

    def mutate(sc, t):
        if t:
            move_ego(sc, 6, dlat=0.35 if t % 3 == 2 else 0.0)
        if t == 4:
            sc["scene_in"]["loc"]["globalpoint"]["dir"] += 60.0          # heading error > 45 deg
            sc["scene_in"]["loc"]["globalpoint"]["dir"] %= 360.0
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=8, mutate=mutate)
    causes = set()
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")
        causes |= set(np.unique(r[4]["afresh_cause"]).tolist())
    assert {0, 2, 3}.issubset(causes), causes
    beh = np.concatenate([r[4]["z_behavior"] for r in res])
    assert (beh == 4).any() or (beh == 5).any(), "no avoidance sweep was accepted in the sample"


def test_decision_stage_off_and_behaviours(dm, oracle):
    """DecisionOut supplied by the caller (lane-change behaviours 2/3 walk the left/right lane, Planning.cpp:443-501)."""
    cfg = dm.default_config(512)
    cfg["decision_stage"] = 0
    sc = dm.gen_scenes(cfg, 7000, 48, 64, junction_every=6)
    si = sc["scene_in"]
    for s in range(48):
        ln = int(si["loc"]["lane_num"][s])
        if s % 3 == 1 and ln > 1:
            si["dec"]["behavior"][s], si["dec"]["target_lanenum"][s] = 2, ln - 1
        if s % 3 == 2 and ln < 3:
            si["dec"]["behavior"][s], si["dec"]["target_lanenum"][s] = 3, ln + 1
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=3, mutate=lambda sc, t: move_ego(sc, 4) if t else None)
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")


def test_dynamic_obstacles_replan_every_tick(dm, oracle):
    """BASELINE configs[3] in small: dynamic obstacles, 30-step horizon, replan forced every tick."""
    cfg = dm.default_config(512)
    cfg["dynamic_obstacles"] = 1
    cfg["force_replan"] = 1
    sc = dm.gen_scenes(cfg, 9000, 16, 256, junction_every=0)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=30)
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")
    assert (res[-1][1]["tick"] == 30).all()
    # the grid must actually change from tick to tick
    assert res[0][2]["order_digest"].tolist() != res[-1][2]["order_digest"].tolist()


def test_search_limits(dm, oracle):
    """LIMIT, OVERFLOW, GOAL_BLOCKED, PATH_TRUNC and NO_PATH exits of the search."""
    base = dm.default_config(512)
    sc = dm.gen_scenes(base, 11000, 32, 64, junction_every=0)
    for key, val in (("max_expansions", 100), ("bucket_cap", 24), ("max_path", 100)):
        cfg = base.copy()
        cfg[key] = val
        pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=1)
        plan_g, st_g, gout_g, plan_o, st_o, gout_o, _ = res[0]
        assert (gout_g["status"] == gout_o["status"]).all(), key
        if key != "bucket_cap":          # on OVERFLOW only the status is specified
            _assert_tick(res[0], key)
        want = {"max_expansions": dm.G_LIMIT, "bucket_cap": dm.G_OVERFLOW, "max_path": dm.G_PATH_TRUNC}[key]
        assert (gout_o["status"] == want).any(), (key, np.bincount(gout_o["status"]))
    # goal walled in: ring of obstacles around the goal -> NO_PATH; goal inside an obstacle -> GOAL_BLOCKED
    cfg = base.copy()
    sc2 = dm.gen_scenes(cfg, 12000, 4, 64, junction_every=0)
    for s in range(4):
        gx, gy = sc2["scene_in"]["goal"]["x"][s], sc2["scene_in"]["goal"]["y"][s]
        o = sc2["obs_pool"][s * 64:(s + 1) * 64]
        if s < 2:
            ang = np.linspace(0, 2 * np.pi, 64, endpoint=False)
            o["x"], o["y"], o["radius"] = gx + 6 * np.cos(ang), gy + 6 * np.sin(ang), 0.5
        else:
            o["x"][0], o["y"][0], o["radius"][0] = gx, gy, 1.0
    pl, res = _tick_both(dm, oracle, cfg, sc2, n_ticks=1)
    _assert_tick(res[0], "walled")
    assert res[0][5]["status"].tolist()[:2] == [dm.G_NO_PATH] * 2 and res[0][5]["status"].tolist()[2:] == [dm.G_GOAL_BLOCKED] * 2


def test_one_shot_batch_call_and_errors(dm, oracle):
    cfg = dm.default_config(128)
    sc = dm.gen_scenes(cfg, 300, 8, 8, junction_every=4)
    pl = dm.Planner(cfg, max_scenes=8, max_obs_total=64)
    st_g, st_o = sc["state"].copy(), sc["state"].copy()
    plan_g, gout_g = pl.plan_tick_batch(sc, st_g)
    plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st_o)
    assert not (compare(plan_g, plan_o) + compare(st_g, st_o) + compare(gout_g, gout_o))
    big = dm.gen_scenes(cfg, 0, 9, 8)
    with pytest.raises(dm.PlannerError):
        pl.set_scenes(big)                         # more scenes than caps.max_scenes
    bad = cfg.copy(); bad["grid_w"] = 100
    with pytest.raises(dm.PlannerError):
        dm.Planner(bad, max_scenes=1)


def test_scalar_stage_ops(dm, oracle):
    """CPlanning::UpdatePlanJudge / SpeedPlanning / CalculateRadius as stand-alone device stages."""
    cfg = dm.default_config(128)
    pl = dm.Planner(cfg, max_scenes=1)
    rng = np.random.default_rng(4)
    dec, loc, st = np.zeros(1, dm.DecisionOutPod), np.zeros(1, dm.LocationOut), np.zeros(1, dm.SceneState)
    for _ in range(60):
        hb, b, pos = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(0, 4))
        lat, derr, rem = float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-90, 90)), float(rng.uniform(0, 20))
        dec["behavior"], loc["pos"] = b, pos
        st["path_lat_dis"], st["path_dir_err"], st["remain_dis"] = lat, derr, rem
        cause = np.zeros(1, np.int32)
        want = oracle.L.orc_UpdatePlanJudge(cfg.ctypes.data, dec.ctypes.data, loc.ctypes.data, hb, st.ctypes.data, cause.ctypes.data)
        got = pl.scalar_stage(0, [hb, b, pos, lat, derr, rem], n_out=2)
        assert (int(got[0]), int(got[1])) == (want, int(cause[0]))
        ob, lon, far, ve = int(rng.integers(0, 2)), float(rng.uniform(0, 30)), float(np.float32(rng.uniform(8, 40))), float(rng.uniform(3, 15))
        dec["velocity_expect"] = ve
        want = oracle.SpeedPlanning(ob, dec, loc, lon, 0.0, far, init=(1.5, 1, -0.5))
        got = pl.scalar_stage(1, [pos, ob, lon, far, ve, 1.5, 1, -0.5], n_out=3)
        assert (got[0], int(got[1]), got[2]) == want or (np.isinf(got[0]) and np.isinf(want[0]))
    pts = np.zeros(200, dm.GlobalPoint2D)
    pts["x"], pts["y"] = np.cumsum(rng.uniform(0.3, 0.7, 200)), np.cumsum(rng.uniform(-0.2, 0.2, 200))
    for nid, fid in ((0, 8), (50, 58), (195, 203), (199, 207)):
        got = pl.scalar_stage(2, [nid, fid], last_Bpoints=pts, n_out=1)[0]
        want = oracle.CalculateRadius(pts, nid, fid)
        assert got == want or (np.isnan(got) and np.isnan(want))
    a, b = np.zeros(8, dm.GlobalPoint2D), np.zeros(8, dm.GlobalPoint2D)
    a["x"], a["y"], b["x"], b["y"] = rng.uniform(0, 99, 8), rng.uniform(0, 99, 8), rng.uniform(0, 99, 8), rng.uniform(0, 99, 8)
    assert np.array_equal(pl.geom_batch(3, a, b), np.hypot(a["x"] - b["x"], a["y"] - b["y"]) * 0 + np.sqrt((a["x"] - b["x"]) ** 2 + (a["y"] - b["y"]) ** 2))
    assert np.array_equal(pl.geom_batch(4, a), cfg["wgs_lat0"][0] + a["y"] * cfg["wgs_deg_per_m_lat"][0])
    assert np.array_equal(pl.geom_batch(5, a), cfg["wgs_lng0"][0] + a["x"] * cfg["wgs_deg_per_m_lng"][0])


def test_decision_refpath_export(dm, oracle):
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    sc = dm.gen_scenes(cfg, 33, 4, 8, junction_every=2)
    pl = dm.Planner(cfg, max_scenes=4, max_obs_total=32)
    st = sc["state"].copy()
    plan, _ = pl.plan_tick_batch(sc, st, want_grid=False)
    for s in range(4):
        n = int(plan["dec"]["refpath_n"][s])
        rp = pl.get_refpath(s, n)
        assert n > 0 and np.isfinite(rp["x"]).all()
        if sc["scene_in"]["loc"]["pos"][s] == 0:      # RefPath: the front corridor of the current lane (Decision.cpp:1813)
            lo = int(sc["scene_in"]["lanes"]["cur_off"][s]) + int(sc["scene_in"]["loc"]["id"][s][0])
            assert np.array_equal(rp["x"], sc["lane_pool"]["x"][lo:lo + n])


def test_cpp_host_classes_example():
    """The C++ CDecision/CPlanning surface (host/) driving the GPU through the C-ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "decision-making-and-path-planning_amd", "host", "example_tick")
    assert os.path.exists(exe), "build it with make -C decision-making-and-path-planning_amd/host"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout[-1500:], r.stderr[-500:])
    assert r.returncode == 0 and "example ok" in r.stdout


def test_cpp_streamed_example():
    """The streamed calls of the C-ABI from C++ (host/example_stream.cpp): 16 ticks of 320 scenes with a new snapshot each, six
    in flight; the published records of the last tick must equal those of the synchronous calls on a second handle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "decision-making-and-path-planning_amd", "host", "example_stream")
    assert os.path.exists(exe), "build it with make -C decision-making-and-path-planning_amd/host"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout[-1500:], r.stderr[-500:])
    assert r.returncode == 0 and "example_stream ok" in r.stdout


def test_grid_2048_bitmap_in_hbm(dm, oracle):
    """BASELINE configs[4] in small: 2048x2048 grid (64 line-mask bits per line: k_search_lds<2>; the sparse views need no more
    LDS than at 512 x 512 - the stored words follow the obstacles, not the grid)."""
    cfg = dm.default_config(2048)
    sc = dm.gen_scenes(cfg, 21000, 6, 64, junction_every=3)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=2, order_cap=1 << 20, mutate=lambda sc, t: move_ego(sc, 5) if t else None)
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")
    st1 = sc["state"].copy()
    _, go, grid_o, order_o, path_o = oracle.plan_tick_one(cfg, sc, 1, st1, order_cap=1 << 20)
    assert (pl.get_grid(1) == grid_o).all()


def test_grid_1024_lds_optin(dm, oracle):
    """1024x1024: 32-bit line masks (k_search_lds<1>)."""
    cfg = dm.default_config(1024)
    sc = dm.gen_scenes(cfg, 22000, 6, 64, junction_every=0)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=1)
    _assert_tick(res[0], "1024")


def test_edge_cases(dm, oracle):
    """Long obstacle lists (read from HBM, not staged in LDS), ID_MORE, wide lanes (more sweep
    candidates), short / empty refpaths and lanes, an ego at the very end of its lane."""
    cfg = dm.default_config(512)
    cfg["ID_MORE"] = 3
    sc = dm.gen_scenes(cfg, 31000, 24, 600, junction_every=4)          # 600 > 512 obstacles per scene
    si = sc["scene_in"]
    si["lanes"]["lane_width"][::3] = 7.0                               # (7.0 - 1.8) / 0.6 -> 9 candidates, capped at 8
    si["lanes"]["lanechg_attribute"][::3] = 0
    si["ref_n"][1], si["dec"]["refpath_n"][1] = 2, 2                   # fewer than 3 refpath points (Planning.cpp:507 fence)
    si["ref_n"][2], si["dec"]["refpath_n"][2] = 0, 0
    si["loc"]["id"][3] = dm.GEN_LANE_PTS - 2                           # two points left on the lane
    si["loc"]["id"][4] = dm.GEN_LANE_PTS + 5                           # id past the end (Decision.cpp:581 min())
    si["lanes"]["cur_n"][5] = 1
    si["lanes"]["left_n"][6] = 0
    si["obs_n"][7] = 0                                                 # no obstacle at all
    for stage in (1, 0):
        cfg["decision_stage"] = stage
        pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=4, mutate=lambda sc, t: None)
        for t, r in enumerate(res):
            _assert_tick(r, f"stage {stage} tick {t}")


def test_zero_and_one_scene(dm, oracle):
    cfg = dm.default_config(128)
    pl = dm.Planner(cfg, max_scenes=2, max_obs_total=16)
    pl.tick(sync=True)                                                  # nothing resident: a no-op
    sc = dm.gen_scenes(cfg, 77, 1, 8, junction_every=0)
    st_g, st_o = sc["state"].copy(), sc["state"].copy()
    plan_g, gout_g = pl.plan_tick_batch(sc, st_g)
    plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st_o)
    assert not (compare(plan_g, plan_o) + compare(st_g, st_o) + compare(gout_g, gout_o))


# ---- lane-change rule tree (Decision.cpp:1011-1772) --------------------------------
def test_lanechange_known_scenes(dm, oracle):
    """Every hand-built scene of test_lanechange_kat.py, device against oracle, 7 ticks each, with the
    localisation switching to the target lane once a change has been decided."""
    import lanechange_scenes as lcs
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    pl = dm.Planner(cfg, device=0, max_scenes=1, max_obs_total=8)
    changed = 0
    for k, kw in enumerate(lcs.SCENARIOS):
        sc = lcs.make_scene(dm, cfg, **kw)
        st_o = sc["state"].copy()
        pl.set_state(sc["state"])
        for t in range(7):
            pl.set_scenes(sc)
            pl.tick(sync=True)
            plan_g, st_g = pl.get_plan(), pl.get_state()
            plan_o, _, _ = oracle.plan_tick_batch(cfg, sc, st_o, want_grid=False)
            bad = compare(plan_g, plan_o, "plan") + compare(st_g, st_o, "state")
            assert not bad, f"scenario {k} {kw} tick {t}\n" + "\n".join(bad[:10])
            tgt = int(st_o["z_target_lanenum"][0])
            if int(st_o["z_segment_lanechg_status"][0]) == 1 and int(st_o["z_behavior"][0]) in (2, 3) and 1 <= tgt <= 3 \
                    and tgt != int(sc["scene_in"]["loc"]["lane_num"][0]) and t >= 4:
                lcs.switch_lane(dm, sc, tgt)
                changed += 1
    assert changed >= 5            # the scenes do reach the "changing lanes" state


def test_lanechange_remaining_length_fallbacks(dm, oracle):
    """The remaining lane-change length flags are decided from a parallel sum unless that sum lies within its rounding bound of
    a threshold or a length is NaN - then the in-order walk decides.  Exact thresholds (0.5 m spacing: 120 points = 60.0 m) and a
    NaN lane point in the middle of the run, against the oracle's in-order loop; and a lane of 700 points (three blocks)."""
    import lanechange_scenes as lcs
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    pl = dm.Planner(cfg, device=0, max_scenes=1, max_obs_total=8, max_lane_pts_total=4096)
    cases = []
    for run in (119, 120, 121, 29, 30, 31, 99, 100, 101, 19, 20, 21):
        for attr in (1, 3):
            cases.append((dict(lane_num=2, map_attr=1, out_lanes=(2,) if attr == 1 else (1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=attr, attr_run=run), None))
    cases.append((dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=250), 40))      # NaN 40 points ahead
    cases.append((dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=250), 150))
    for k, (kw, nan_at) in enumerate(cases):
        sc = lcs.make_scene(dm, cfg, **kw)
        if nan_at is not None:
            sc["lane_pool"]["x"][lcs.EGO_ID + nan_at] = np.nan
        st_o = sc["state"].copy()
        pl.set_state(sc["state"])
        for t in range(4):
            pl.set_scenes(sc)
            pl.tick(sync=True)
            plan_o, _, _ = oracle.plan_tick_batch(cfg, sc, st_o, want_grid=False)
            bad = compare(pl.get_plan(), plan_o, "plan") + compare(pl.get_state(), st_o, "state")
            assert not bad, f"case {k} {kw} nan {nan_at} tick {t}\n" + "\n".join(bad[:10])
    # a long lane: the run crosses two block boundaries of the reduction (irregular spacing: no exact sums)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)])
    n_long = 700
    rng = np.random.default_rng(3)
    lane = np.zeros(n_long, dm.GlobalPoint3D)
    lane["x"] = 100.0 + np.cumsum(rng.uniform(0.05, 0.12, n_long))
    lane["y"] = sc["lane_pool"]["y"][0]
    pool = np.concatenate([lane, sc["lane_pool"]])
    attrs = np.concatenate([np.ones(n_long, np.uint8), sc["attr_pool"]])
    attrs[600:n_long] = 0
    sc2 = dict(sc, lane_pool=pool, attr_pool=attrs)
    si = sc2["scene_in"] = sc["scene_in"].copy()
    si["lanes"]["cur_off"], si["lanes"]["cur_n"] = 0, n_long
    si["lanes"]["left_off"] += n_long
    si["lanes"]["right_off"] += n_long
    si["loc"]["id"][:] = 5
    si["loc"]["globalpoint"]["x"] = lane["x"][5]
    for j in range(len(sc2["obs_pool"])):
        sc2["obs_pool"]["x"][j] = lane["x"][5] + 10.0
    st_o = sc["state"].copy()
    pl.set_state(sc["state"])
    for t in range(4):
        pl.set_scenes(sc2)
        pl.tick(sync=True)
        plan_o, _, _ = oracle.plan_tick_batch(cfg, sc2, st_o, want_grid=False)
        bad = compare(pl.get_plan(), plan_o, "plan") + compare(pl.get_state(), st_o, "state")
        assert not bad, f"long lane tick {t}\n" + "\n".join(bad[:10])


def test_lanechange_generated_scenes(dm, oracle):
    """Generated scenes with mixed decision periods: 30 ticks, the ego advancing along its lane, every tick compared."""
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    n = 1024
    sc = dm.gen_scenes(cfg, 4096, n, 64, junction_every=8)
    sc["scene_in"]["period_last"] = np.array([100.0, 700.0, 900.0, 1600.0, 2500.0])[np.arange(n) % 5]
    seen_b, seen_dlg = set(), set()

    def mutate(sc, t):
        if t and t % 3 == 0:
            move_ego(sc, 2)

    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=30, mutate=mutate)
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")
        seen_b |= set(np.unique(r[4]["z_behavior"]).tolist())
        seen_dlg |= set(np.unique(r[4]["z_behavior_to_dlg"]).tolist())
    assert {1, 2, 3, 4, 5} <= seen_b, seen_b                  # keep, both lane changes, both avoidance sides
    assert {2, 3, 4, 5, 6, 8, 9} <= seen_dlg, seen_dlg        # every panel code of the rule tree that generated scenes reach


def test_lanechange_needs_attributes(dm):
    cfg = dm.default_config(128)
    sc = dm.gen_scenes(cfg, 0, 2, 8)
    pl = dm.Planner(cfg, device=0, max_scenes=2, max_obs_total=16)
    noattr = dict(sc)
    noattr["attr_pool"] = None
    pl.set_scenes(noattr)
    pl.set_state(sc["state"])
    with pytest.raises(dm.PlannerError, match="lane attribute"):
        pl.tick(sync=True)
    cfg["lanechg_stage"] = 0
    pl.set_config(cfg)
    pl.tick(sync=True)              # the rule tree switched off: no attributes needed


# ---- ticks enqueued back to back (no host sync in between): the pipelined three-stream tick -----------------
@pytest.mark.parametrize("n,grid,n_obs", [(320, 128, 24), (64, 128, 24), (300, 512, 64), (272, 128, 136)])   # the last: consecutive searches overlap
def test_back_to_back_ticks_match_synchronised_ones(dm, oracle, n, grid, n_obs):
    """Dynamic obstacles, replanning every tick, 9 ticks without a host sync: the front of tick t+1 and the scoring of
    tick t overlap the search (double-buffered by tick parity).  The final state, plan and grid results must be those
    of the oracle ticking one tick at a time; the device is also run once more with a sync after every tick."""
    cfg = dm.default_config(grid)
    cfg["dynamic_obstacles"] = 1
    cfg["force_replan"] = 1
    sc = dm.gen_scenes(cfg, 7000, n, n_obs, junction_every=8)
    ticks = 9
    st_o = sc["state"].copy()
    for _ in range(ticks):
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st_o, n_threads=8)
    for synced in (False, True):
        pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
        pl.set_scenes(sc)
        pl.set_state(sc["state"])
        for _ in range(ticks):
            pl.tick(sync=synced)
        pl.sync()
        res = (pl.get_plan(), pl.get_state(), pl.get_grid_out(), plan_o, st_o, gout_o, None)
        _assert_tick(res, f"synced={synced}")
        g = pl.get_grid(3)
        obs = oracle.effective_obstacles(cfg, sc["obs_pool"][3 * n_obs:4 * n_obs], sc["mot_pool"][3 * n_obs:4 * n_obs], ticks - 1)
        assert np.array_equal(g, oracle.rasterise(cfg, (0.0, 0.0), obs))       # the bitmaps of the LAST tick
        pl.close()


# ---- map store: one resident map, lane views derived on the device ---------------------
def test_map_store_views_and_ticks(dm, oracle):
    import map_scenes as ms
    cfg = dm.default_config(128)
    m = ms.build_map(dm, n_roads=5)
    n, n_obs = 96, 16
    sc = ms.make_egos(dm, cfg, m, n, n_obs)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs, max_lane_pts_total=len(m["points"]),
                    max_ref_pts_total=max(len(m["jpoints"]), 1))
    pl.set_map(m)
    pl.set_egos(sc)
    want = ms.resolve(dm, m, sc["scene_in"])
    got = pl.get_scene_in()
    assert not compare(got, want, "scene_in")                   # lane views, attribute / width at the ego point, junction slices
    assert set(np.unique(want["loc"]["pos"])) == {0, 1, 2} and (want["lanes"]["left_n"] == 0).any() and (want["lanes"]["right_n"] == 0).any()
    # the same scenes through the oracle, which indexes the shared pools with the views derived above
    sc_o = dict(sc)
    sc_o["scene_in"] = want
    st_o = sc["state"].copy()
    pl.set_state(sc["state"])
    for t in range(6):
        pl.tick(sync=True)
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc_o, st_o, n_threads=8)
        _assert_tick((pl.get_plan(), pl.get_state(), pl.get_grid_out(), plan_o, st_o, gout_o, None), f"tick {t}")
    seen = set(np.unique(st_o["z_behavior"]).tolist())
    assert 1 in seen and len(seen) >= 2                         # more than lane keeping happens on this map


def test_map_store_errors(dm):
    import map_scenes as ms
    cfg = dm.default_config(128)
    m = ms.build_map(dm, n_roads=3)
    sc = ms.make_egos(dm, cfg, m, 4, 4)
    pl = dm.Planner(cfg, device=0, max_scenes=4, max_obs_total=16, max_lane_pts_total=len(m["points"]), max_ref_pts_total=len(m["jpoints"]))
    with pytest.raises(dm.PlannerError, match="resident map"):
        pl.set_egos(sc)
    bad = dict(m)
    bad["lanes"] = m["lanes"].copy()
    bad["lanes"]["n_points"][2] = len(m["points"])              # slice runs past the point pool
    with pytest.raises(dm.PlannerError, match="outside the point pool"):
        pl.set_map(bad)
    pl.set_map(m)
    sc["scene_in"]["loc"]["road_num"][1] = 9                    # no such road: the reference would index out of its vectors
    with pytest.raises(dm.PlannerError, match="outside the map"):
        pl.set_egos(sc)
    small = dm.Planner(cfg, device=0, max_scenes=4, max_obs_total=16, max_lane_pts_total=100, max_ref_pts_total=10)
    with pytest.raises(dm.PlannerError, match="larger than caps"):
        small.set_map(m)


def test_back_to_back_ticks_across_config_changes(dm, oracle):
    """The grid stage switched off and on again between un-synchronised ticks (three-stream form -> one stream -> back),
    and the scene inputs replaced in between: every hand-over between the chains must still be ordered."""
    n, n_obs = 288, 16
    cfg = dm.default_config(128)
    cfg["dynamic_obstacles"] = 1
    cfg["force_replan"] = 1
    sc = dm.gen_scenes(cfg, 8100, n, n_obs, junction_every=8)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
    pl.set_scenes(sc)
    pl.set_state(sc["state"])
    st_o = sc["state"].copy()
    plan_o = gout_o = None
    schedule = [1, 1, 1, 0, 0, 1, 1, 0, 1, 1]
    for t, g in enumerate(schedule):
        cfg["grid_stage"] = g
        pl.set_config(cfg)
        if t == 5:
            move_ego(sc, 3)
            pl.set_scenes(sc)
        pl.tick()
        plan_o, go, _ = oracle.plan_tick_batch(cfg, sc, st_o, n_threads=8, want_grid=bool(g))
        if g:
            gout_o = go
    pl.sync()
    _assert_tick((pl.get_plan(), pl.get_state(), pl.get_grid_out(), plan_o, st_o, gout_o, None), "after the schedule")


@pytest.mark.parametrize("gw,gh", [(96, 160), (224, 64), (32, 32), (640, 384), (160, 96)])      # 96 rows: one band that must not be halved
def test_non_square_grids(dm, oracle, gw, gh):
    """Line widths that are not a power of two (3, 7, 20 words), very small and non-square grids: the word summaries,
    the band layout of the rasteriser and the candidate-word scans must not depend on the 512 x 512 shape."""
    cfg = dm.default_config(gw, gh)
    n = 48
    sc = dm.gen_scenes(cfg, 9100, n, 12, junction_every=0)
    # the generator spreads goals over the width: keep them inside the shorter grids
    sc["scene_in"]["goal"]["y"] = np.clip(sc["scene_in"]["goal"]["y"], 0.0, gh * float(cfg["cell"][0]) - 0.01)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=2, order_cap=gw * gh)
    for t, r in enumerate(res):
        _assert_tick(r, f"{gw}x{gh} tick {t}")
    for s_ in (0, 7, 31):
        assert np.array_equal(pl.get_grid(s_), r[6][s_])
    found = int((res[-1][5]["status"] == dm.G_FOUND).sum())
    assert found >= (4 if gw > 32 else 0)


@pytest.mark.parametrize("grid,cell,origin", [(512, 0.2, (0.0, 0.0)), (512, 0.05, (4.1e6, -6.3e5)), (256, 1.0, (-2.5e6, 3.0e6)),
                                              (2048, 0.2, (5.0e5, 4.4e6)), (128, 0.01, (1.0e6, 1.0e6))])
def test_rasteriser_adversarial_footprints(dm, oracle, grid, cell, origin):
    """Footprints chosen against the span estimates of the rasteriser (pass 1 in single precision on coordinates relative to the
    grid origin, pass 2 from a raw square root, both certified by the oracle's own predicate): radii from a fraction of a cell to
    several grid widths, centres on cell centres and cell borders, discs that graze a line exactly (dv^2 == R^2), centres outside
    the grid that reach in, origins at 1e6 metres, cells from 1 cm to 1 m.  The exported grid (built by the code the search uses)
    must equal the oracle's, brute-force form included."""
    cfg0 = dm.default_config(grid)
    cfg = cfg0.copy()
    cfg["cell"] = cell
    cfg["inflate"] = 0.35 * cell                                  # footprints down to less than a cell
    infl = float(cfg["inflate"][0])
    ext = grid * cell
    rng = np.random.default_rng(int(grid * 1000 + cell * 100))
    discs = []
    for k in range(10):                                           # grazing: the disc's edge exactly on a line of cell centres
        r = float(rng.uniform(2.5, 12.0)) * cell
        cy = (int(rng.integers(20, grid - 20)) + 0.5) * cell
        discs.append((float(rng.uniform(0.1, 0.9)) * ext, cy + (r + infl) * (1 if k % 2 else -1), r))          # its centre line +- R
        discs.append(((int(rng.integers(20, grid - 20)) + 0.5) * cell - (r + infl), float(rng.uniform(0.1, 0.9)) * ext, r))
    for k in range(8):                                            # centres on cell centres / cell borders, tiny and sub-cell radii
        ix, iy = int(rng.integers(4, grid - 4)), int(rng.integers(4, grid - 4))
        discs.append(((ix + (0.5 if k % 2 else 0.0)) * cell, (iy + (0.5 if k % 4 < 2 else 0.0)) * cell, [0.0, 0.3 * cell, 0.5 * cell, 1.0 * cell][k % 4]))
    discs += [(-0.3 * ext, 0.4 * ext, 0.32 * ext), (0.5 * ext, 1.2 * ext, 0.25 * ext), (1.05 * ext, 1.05 * ext, 0.1 * ext),
              (0.7 * ext, 0.3 * ext, 0.08 * ext), (-5.0 * ext, 0.5 * ext, 4.9 * ext), (0.2 * ext, -1.0 * ext, 0.9 * ext)]      # large, outside, reaching in (or not)
    discs += [(float(rng.uniform(0, 1)) * ext, float(rng.uniform(0, 1)) * ext, float(rng.uniform(0.2, 3.0)) * cell) for _ in range(12)]
    # span ends ON cell centres (within the rounding of the float32 radius): centre on a cell centre and R = k cells - on the disc's
    # own row and column, and (k = 5, 13: 3-4-5, 5-12-13) on the lines 3, 4, 5, 12 cells away - or centre on a cell border and
    # R = k + 0.5 cells.  The error bound of exact_span cannot certify these ends: the predicate itself decides them, as in the oracle.
    for k, half_cell in ((1, 0), (2, 0), (5, 0), (13, 0), (3, 1), (7, 1)):
        if k + 2 >= grid // 4:
            continue
        ix, iy = int(rng.integers(k + 2, grid - k - 2)), int(rng.integers(k + 2, grid - k - 2))
        discs.append(((ix + (0.0 if half_cell else 0.5)) * cell, (iy + (0.0 if half_cell else 0.5)) * cell, (k + 0.5 * half_cell) * cell - infl))
    m = len(discs)
    sc = dm.gen_scenes(cfg0, 0, 1, m, junction_every=0)            # (the generator places its own obstacles for the default cell: all overwritten)
    si = sc["scene_in"]
    si["grid_origin"]["x"], si["grid_origin"]["y"] = origin
    for j, (x, y, r) in enumerate(discs):
        sc["obs_pool"][j]["x"], sc["obs_pool"][j]["y"], sc["obs_pool"][j]["radius"], sc["obs_pool"][j]["type"] = origin[0] + x, origin[1] + y, r, 0
    sc["mot_pool"][:] = 0
    si["loc"]["globalpoint"]["x"], si["loc"]["globalpoint"]["y"] = origin[0] + 0.05 * ext, origin[1] + 0.05 * ext
    si["goal"]["x"], si["goal"]["y"] = origin[0] + 0.95 * ext, origin[1] + 0.95 * ext
    pl = dm.Planner(cfg, max_scenes=1, max_obs_total=m)
    pl.set_scenes(sc)
    pl.set_state(sc["state"])
    pl.tick(sync=True)
    obs = sc["obs_pool"][:m]
    grid_o = oracle.rasterise(cfg, origin, obs)
    assert np.array_equal(grid_o, oracle.rasterise(cfg, origin, obs, brute=True))
    grid_g = pl.get_grid(0)
    assert grid_g.shape == grid_o.shape and 0 < int(grid_o.sum()) < grid_o.size
    assert np.array_equal(grid_g, grid_o), f"{int((grid_g != grid_o).sum())} cells differ"
    # and the search on it: same expansion count, same path
    st = sc["state"].copy()
    _, go, _, _, path_o = oracle.plan_tick_one(cfg, sc, 0, st, order_cap=0)
    g = pl.get_grid_out()[0]
    assert int(g["status"]) == int(go["status"]) and int(g["n_expanded"]) == int(go["n_expanded"]) and int(g["path_len"]) == int(go["path_len"])


# ---- the bench workloads at their full size (BASELINE configs[1], [3], [4]) -----------------------------
@pytest.mark.parametrize("grid,n_obs,dynamic,n_ticks", [(512, 64, 0, 4), (512, 256, 1, 30), (2048, 64, 0, 2)],
                         ids=["configs1", "configs3", "configs4"])
def test_bench_workloads_full_size(dm, oracle, grid, n_obs, dynamic, n_ticks):
    """Exactly what bench.py times — 1024 scenes of seed 0, ticks queued back to back without a host sync —
    against the oracle ticking the same scenes: every PlanOut / SceneState / GridOut field (the digest in GridOut
    covers the expansion order and the path cells), plus the full order and path of a sample of scenes."""
    import os
    cfg = dm.default_config(grid)
    if dynamic:
        cfg["dynamic_obstacles"] = 1
        cfg["force_replan"] = 1
    n = 1024
    sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=8)
    cap = grid * grid if grid <= 512 and not dynamic else 0
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs, order_cap=cap)
    pl.set_scenes(sc)
    pl.set_state(sc["state"])
    for _ in range(n_ticks):
        pl.tick(sync=False)
    pl.sync()
    plan_g, st_g, gout_g = pl.get_plan(), pl.get_state(), pl.get_grid_out()
    st_o = sc["state"].copy()
    threads = min(64, os.cpu_count() or 8)
    for _ in range(n_ticks):
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st_o, n_threads=threads, want_grid=True)
    _assert_tick((plan_g, st_g, gout_g, plan_o, st_o, gout_o, None), f"{grid}/{n_obs}/{n_ticks} ticks")
    print("status histogram:", np.bincount(gout_o["status"], minlength=5).tolist(),
          "max expanded:", int(gout_o["n_expanded"].max()))
    if cap and not dynamic:
        for s in (0, 406, 1023):                          # 406: the slowest search of the batch
            st1 = sc["state"].copy()
            for _ in range(n_ticks):
                _, go, grid_o, order_o, path_o = oracle.plan_tick_one(cfg, sc, s, st1, order_cap=cap)
            assert (pl.get_grid(s) == grid_o).all(), s
            assert (pl.get_order(s, int(go["n_expanded"])) == order_o).all(), s
            assert (pl.get_path(s, int(go["path_len"])) == path_o).all(), s


def test_bench_two_ranks_rehearsal(dm, oracle, tmp_path):
    """`python bench.py --gpus 2` - plain python, no torchrun: the bench starts its own ranks - at the per-rank workload of
    BASELINE configs[2] (1024 scenes, 512 x 512, 64 obstacles).  The N > 1 code path (scatter from rank 0, device pointers
    into pp_set_scenes, max-over-ranks timing, gather of PlanOut + SceneState + GridOut, bit-for-bit check of every shard
    on rank 0) runs with two ranks sharing this box's GPU over gloo; the measured runs use RCCL with one rank per GPU.
    Both gathered shards are then compared with the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    dump = str(tmp_path / "gathered.npz")
    n, grid, n_obs, warm, steps = 1024, 512, 64, 1, 2
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm),
           "--scenes", str(n), "--grid", str(grid), "--obstacles", str(n_obs), "--backend", "gloo", "--no-cpu-baseline",
           "--latency-ticks", "0", "--dump-gathered", dump]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    print(r.stdout[-3000:], r.stderr[-1500:])
    assert r.returncode == 0
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_scenes"] == 2 * n and line["value"] > 0
    mg = line["multi_gpu"]
    per_scene = dm.PlanOut.itemsize + dm.SceneState.itemsize + dm.GridOut.itemsize
    assert mg["gathered_bytes"] == 2 * n * per_scene
    assert mg["shards_verified"] == 2 and mg["shard_mismatches"] == 0
    assert [p["rank"] for p in mg["per_rank"]] == [0, 1]
    assert all(p["frac"] > 0 and p["avg_launch_ms"] > 0 for p in mg["per_rank"])
    got = np.load(dump)
    cfg = dm.default_config(grid)
    threads = min(64, os.cpu_count() or 8)
    counts = np.zeros(dm.G_STATUS_COUNT, np.int64)
    for rk in range(2):                                    # rank rk's shard is scenes rk*n .. rk*n + n - 1
        sc = dm.gen_scenes(cfg, rk * n, n, n_obs, junction_every=8)
        st = sc["state"].copy()
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st, n_threads=threads, want_grid=True, n_ticks=line["ticks_run"])
        sl = slice(rk * n, (rk + 1) * n)
        bad = (compare(got["plan"][sl], plan_o, "plan") + compare(got["state"][sl], st, "state")
               + compare(got["grid_out"][sl], gout_o, "grid"))
        assert not bad, f"rank {rk}:\n" + "\n".join(bad[:10])
        counts += np.bincount(gout_o["status"], minlength=dm.G_STATUS_COUNT)
    assert line["search_status_counts"] == counts.tolist()
    assert line["searched_scenes"] == int(counts.sum() - counts[dm.G_GOAL_BLOCKED])


@pytest.mark.parametrize("dx,dy", [(25.0, 0.0), (-30.0, 0.0), (0.0, -40.0)])
def test_ego_outside_the_grid(dm, oracle, dx, dy):
    """An ego outside its grid starts the search from a clamped border cell, far from where it stands: the scoring
    kernel's obstacle culling must follow the path, not the ego (found by tests/soak_parity.py: with the box around
    the ego, an obstacle beside the far part of the path was dropped from the path candidate's penalty)."""
    cfg = dm.default_config(64, 384)
    n = 512
    sc = dm.gen_scenes(cfg, 4242, n, 130, junction_every=3)
    si = sc["scene_in"]
    si["goal"]["y"] = np.clip(si["goal"]["y"], 0.0, 384 * float(cfg["cell"][0]) - 0.01)
    si["grid_origin"]["x"] += dx
    si["grid_origin"]["y"] += dy
    pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * 130)
    st_g, st_o = sc["state"].copy(), sc["state"].copy()
    plan_g, gout_g = pl.plan_tick_batch(sc, st_g)
    plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st_o, n_threads=8, want_grid=True)
    bad = compare(gout_g, gout_o, "grid") + compare(plan_g, plan_o, "plan") + compare(st_g, st_o, "state")
    assert not bad, "\n".join(bad[:10])
    assert int((gout_o["status"] == dm.G_FOUND).sum()) >= 20


def test_no_candidate_at_all(dm, oracle):
    """n_lattice = 0 leaves the grid path as the only candidate; a scene without a path then has none: n_candidates 0,
    best_candidate 0 and a best_path of zeros (found by tests/soak_parity.py: the kernel resampled an empty path)."""
    cfg = dm.default_config(128)
    cfg["n_lattice"] = 0
    sc = dm.gen_scenes(cfg, 5150, 200, 24, junction_every=0)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=2)
    for t, r in enumerate(res):
        _assert_tick(r, f"tick {t}")
    gout_o = res[-1][5]
    none = gout_o["n_candidates"] == 0
    assert none.any() and (~none).any(), np.bincount(gout_o["status"])
    assert not gout_o["best_path"]["x"][none].any()


@pytest.mark.parametrize("n_walls,want", [(5, 0), (7, 7)])
def test_cost_range_serpentine_2048(dm, oracle, n_walls, want):
    """f = g + h close to and beyond DMPP_F_LIMIT (the device keeps f/2 in 16 bits): five serpentine walls on 2048 x 2048
    cost 124,162 and must match the oracle in every field, expansion order included; seven walls exceed the limit and end
    with DMPP_G_COST_RANGE in both (only the status is defined there).  Thousands of closed cells: the closed set spills
    from the LDS hash to HBM."""
    from grid_scenes import serpentine
    cfg = dm.default_config(2048)
    cfg["max_expansions"] = 400000
    cfg["max_path"] = 32768
    sc = serpentine(dm, cfg, n_walls)
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=1, order_cap=8192)
    _assert_tick(res[0], f"{n_walls} walls")
    gout_o = res[0][5]
    assert int(gout_o["status"][0]) == want
    if want == 0:
        assert int(gout_o["path_cost"][0]) > 120000 and int(gout_o["n_expanded"][0]) > 768
        st1 = sc["state"].copy()
        _, go, grid_o, order_o, path_o = oracle.plan_tick_one(cfg, sc, 0, st1, order_cap=8192)
        assert (pl.get_order(0, int(go["n_expanded"])) == order_o).all()
        assert (pl.get_path(0, int(go["path_len"])) == path_o).all()


@pytest.mark.parametrize("grid,n_side,radius,min_peak", [(1024, 40, 0.05, 1500), (2048, 44, 0.2, 2000)])
def test_open_list_spills_to_hbm(dm, oracle, grid, n_side, radius, min_peak):
    """Thousands of live open-list entries (a field of small discs behind two walls: the search floods it): the device keeps
    DMPP_OPEN_CAP slots in LDS and spills the entries that pop last to HBM, merging them back as the minimum rises.  Status,
    expansion order, path and every counter must be the oracle's - and with bucket_cap below the peak both must overflow."""
    from grid_scenes import pebble_field
    cfg = dm.default_config(grid)
    cfg["bucket_cap"] = 16384
    cfg["max_path"] = 32768
    sc = pebble_field(dm, cfg, n_side=n_side, radius=radius)
    cap_order = 1 << 18
    pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=1, order_cap=cap_order)
    _assert_tick(res[0], "spilling open list")
    st1 = sc["state"].copy()
    _, go, _, order_o, path_o = oracle.plan_tick_one(cfg, sc, 0, st1, order_cap=cap_order)
    peak = oracle.last_peak_open()
    assert int(go["status"]) == 0 and peak > min_peak, (int(go["status"]), peak)
    assert (pl.get_order(0, int(go["n_expanded"])) == order_o).all()
    assert (pl.get_path(0, int(go["path_len"])) == path_o).all()
    # the same scene with fewer entries allowed than it needs: OVERFLOW in both (only the status is specified)
    cfg["bucket_cap"] = peak - 50
    pl.set_config(cfg)
    pl.set_state(sc["state"])
    pl.tick(sync=True)
    st2 = sc["state"].copy()
    _, go2, _, _, _ = oracle.plan_tick_one(cfg, sc, 0, st2)
    assert int(go2["status"]) == dm.G_OVERFLOW and int(pl.get_grid_out()["status"][0]) == dm.G_OVERFLOW
    # a handle made for bucket_cap <= DMPP_OPEN_CAP has no spill area and refuses a larger cap later
    small = dm.default_config(128)
    small["bucket_cap"] = 512
    p2 = dm.Planner(small, max_scenes=1)
    small["bucket_cap"] = 4096
    with pytest.raises(dm.PlannerError, match="bucket_cap"):
        p2.set_config(small)


def test_unsynced_ticks_then_grid_and_stage_switch(dm, oracle):
    """What a caller may do between ticks without a host sync (the three-stream tick, n >= 256): read the grid of the last
    tick (pp_get_grid expands the bitmaps the rasteriser wrote on another stream), then switch the grid stage off and on
    again - Decision + Planning move to the handle's stream and back to the front chain - and keep ticking."""
    cfg = dm.default_config(128)
    cfg["dynamic_obstacles"] = 1
    n, n_obs = 384, 128                                    # >= 128 obstacles per scene: consecutive searches overlap on two streams
    sc = dm.gen_scenes(cfg, 777, n, n_obs, junction_every=5)
    pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * n_obs)
    pl.set_scenes(sc)
    pl.set_state(sc["state"])
    st_o = sc["state"].copy()
    cfg_off = cfg.copy()
    cfg_off["grid_stage"] = 0
    plan_o = gout_o = grids_o = None
    for phase, (c_, ticks) in enumerate([(cfg, 3), (cfg_off, 2), (cfg, 2), (cfg_off, 1), (cfg, 3)]):
        pl.set_config(c_)
        for _ in range(ticks):
            pl.tick(sync=False)
            plan_o, g2, gr2 = oracle.plan_tick_batch(c_, sc, st_o, n_threads=8, want_grid=bool(c_["grid_stage"][0]),
                                                     keep_grids=bool(c_["grid_stage"][0]))
            if c_["grid_stage"][0]:
                gout_o, grids_o = g2, gr2
        if c_["grid_stage"][0]:
            for s in (0, 5, n - 1):                         # no sync before this: pp_get_grid orders itself after the rasteriser
                assert (pl.get_grid(s) == grids_o[s]).all(), (phase, s)
        # after a phase without the grid stage the getters still return the LAST search (a grid-off tick does not advance
        # the search buffer sets): GridOut and the path cells of a found scene
        bad = compare(pl.get_grid_out(), gout_o, "grid")
        assert not bad, f"phase {phase}\n" + "\n".join(bad[:10])
        found = np.flatnonzero(gout_o["status"] == 0)
        if len(found):
            s0 = int(found[0])
            path = pl.get_path(s0, int(gout_o["path_len"][s0]))
            assert path[0] == gout_o["start_cell"][s0] and path[-1] == gout_o["goal_cell"][s0], phase
        bad = compare(pl.get_plan(), plan_o, "plan") + compare(pl.get_state(), st_o, "state")
        assert not bad, f"phase {phase}\n" + "\n".join(bad[:10])


def test_scene_slices_are_validated(dm):
    """A SceneIn whose obstacle / lane / refpath slice leaves its pool is refused by pp_set_scenes (PP_ERR_ARG): nothing is
    left resident and no kernel ever follows the slice.  An empty slice may carry any offset."""
    cfg = dm.default_config(128)
    n, n_obs = 8, 6
    pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * n_obs)
    for field, sub, val in [("obs_off", None, n * n_obs - 2), ("obs_n", None, -1), ("obs_off", None, -3),
                            ("lanes", "cur_n", 10 ** 7), ("lanes", "left_off", 2 ** 30), ("ref_off", None, n * dm.GEN_REF_PTS)]:
        sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=2)
        si = sc["scene_in"]
        if sub:
            if sub == "left_off":
                si["lanes"]["left_n"][3] = 5
            si[field][sub][3] = val
        else:
            if field == "ref_off":
                si["ref_n"][3] = 4
            si[field][3] = val
        with pytest.raises(dm.PlannerError, match="outside its pool"):
            pl.set_scenes(sc)
        pl.tick(sync=True)                                   # nothing resident: a no-op, not a fault
    sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=2)
    sc["scene_in"]["lanes"]["left_off"][2] = 2 ** 30         # empty slice, wild offset: accepted
    sc["scene_in"]["lanes"]["left_n"][2] = 0
    pl.set_scenes(sc)
    pl.tick(sync=True)


def test_device_pointer_inputs_without_a_motion_pool(dm, oracle):
    """Inputs written straight into the handle's buffers (pp_device_ptr, as an RCCL scatter would) and declared with
    pp_set_n_scenes: without a motion pool the obstacles stand still even when cfg.dynamic_obstacles is set (velocities
    nobody uploaded are zero, not uninitialised memory); with one they move."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    cfg = dm.default_config(128)
    cfg["dynamic_obstacles"] = 1
    n, n_obs = 32, 12
    sc = dm.gen_scenes(cfg, 31, n, n_obs, junction_every=4)
    for with_motion in (False, True):
        pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * n_obs)
        bufs = [(dm.BUF_SCENE_IN, sc["scene_in"]), (dm.BUF_LANE_POOL, sc["lane_pool"]), (dm.BUF_LANE_ATTR, sc["attr_pool"]),
                (dm.BUF_REF_POOL, sc["ref_pool"]), (dm.BUF_OBS_POOL, sc["obs_pool"]), (dm.BUF_STATE, sc["state"])]
        if with_motion:
            bufs.append((dm.BUF_MOT_POOL, sc["mot_pool"]))
        for which, arr in bufs:
            ptr, size = pl.device_ptr(which)
            raw = np.frombuffer(arr.tobytes(), np.uint8).copy()
            assert raw.size <= size
            assert hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(raw.ctypes.data), C.c_size_t(raw.size), 1) == 0     # hipMemcpyHostToDevice
        dm._check(pl.lib.pp_set_n_scenes(pl.h, n, len(sc["lane_pool"]), len(sc["ref_pool"]), n * n_obs, int(with_motion), 1))
        pl.n = n
        sc_o = dict(sc)
        if not with_motion:
            sc_o["mot_pool"] = None
        st_o = sc["state"].copy()
        for _ in range(3):
            pl.tick(sync=True)
            plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc_o, st_o, n_threads=8, want_grid=True)
        bad = compare(pl.get_plan(), plan_o, "plan") + compare(pl.get_state(), st_o, "state") + compare(pl.get_grid_out(), gout_o, "grid")
        assert not bad, f"with_motion={with_motion}\n" + "\n".join(bad[:10])
        pl.close()


@pytest.mark.parametrize("env,grid,n_obs", [({"DMPP_SEARCH_GBM": "1"}, 512, 64), ({"DMPP_LDS_BUDGET": "600"}, 512, 64),
                                            ({"DMPP_LDS_BUDGET": "64"}, 128, 24), ({"DMPP_SEARCH_GBM": "1"}, 2048, 64)])
def test_search_fallback_paths(dm, oracle, env, grid, n_obs):
    """The dense form of the search's bitmaps (written to HBM by the scene's own workgroup) that takes the scenes whose obstacle
    words do not fit the LDS budget of a launch: forced for every scene (DMPP_SEARCH_GBM), and with a budget so small that
    only some scenes fit (DMPP_LDS_BUDGET: both forms in one launch).  Same results as the oracle either way."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cfg = dm.default_config(grid)
        n = 96 if grid < 2048 else 6
        sc = dm.gen_scenes(cfg, 8800, n, n_obs, junction_every=4)
        if "DMPP_LDS_BUDGET" in env:                      # a few sparse scenes that do fit next to the dense ones
            keep = sc["scene_in"]["obs_n"].copy()
            keep[::3] = max(n_obs // 8, 1)
            sc["scene_in"]["obs_n"] = keep
        pl, res = _tick_both(dm, oracle, cfg, sc, n_ticks=2, order_cap=4096 if grid < 2048 else 0,
                             mutate=lambda sc_, t: move_ego(sc_, 4) if t else None)
        for t, r in enumerate(res):
            _assert_tick(r, f"{env} tick {t}")
        budget, need, dense = pl.search_info()
        assert dense > 0 and (dense == n or "DMPP_LDS_BUDGET" in env), (budget, need, dense)      # the dense path really ran
        if "DMPP_LDS_BUDGET" in env:
            assert 0 < dense < n and budget == int(env["DMPP_LDS_BUDGET"])
        if grid < 2048:
            for s_ in (0, 1, 5):
                st1 = sc["state"].copy()
                go = None
                for _ in range(2):
                    _, go, grid_o, order_o, path_o = oracle.plan_tick_one(cfg, sc, s_, st1, order_cap=4096)
                assert (pl.get_grid(s_) == grid_o).all(), s_
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_lds_budget_adapts(dm, oracle):
    """The LDS budget of the search follows the scenes: sparse scenes first, then much denser ones in the same handle - the
    first ticks after the change run the dense scenes through the fallback, later ones fit again; results equal the oracle's
    throughout."""
    cfg = dm.default_config(256)
    n = 64
    pl = dm.Planner(cfg, max_scenes=n, max_obs_total=n * 200)
    seen = []
    for n_obs, seed in ((6, 100), (200, 200), (6, 300)):
        sc = dm.gen_scenes(cfg, seed, n, n_obs, junction_every=0)
        st_o = sc["state"].copy()
        pl.set_scenes(sc)
        pl.set_state(sc["state"])
        for t in range(5):
            pl.tick(sync=True)
            plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc, st_o, n_threads=8, want_grid=True)
            bad = compare(pl.get_grid_out(), gout_o, "grid") + compare(pl.get_state(), st_o, "state")
            assert not bad, f"{n_obs} obstacles, tick {t}\n" + "\n".join(bad[:10])
            seen.append(pl.search_info())
    budgets = [b for b, _, _ in seen]
    assert max(budgets[5:10]) > 2 * budgets[4], seen       # the budget grew with the dense scenes ...
    assert seen[9][2] == 0, seen                           # ... until they all fit again
    assert budgets[14] < max(budgets[5:10]), seen          # ... and shrank when they left
