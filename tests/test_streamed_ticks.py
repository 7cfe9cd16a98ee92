"""GPU parity of the STREAMED tick path: pp_update_async -> pp_plan_tick -> pp_fetch_async, tick after tick with no host
wait in between (the reference reads its blackboard and publishes on every tick: Planning.cpp:95-112,186,214;
Decision.cpp:155-160,203).  Egos AND obstacles change on the host every tick; the results of EVERY tick are downloaded
asynchronously into their own pinned buffers and compared with the oracle ticking one tick at a time on the same inputs.
"""
import numpy as np
import pytest

from parity_util import compare, move_ego

pytestmark = pytest.mark.gpu


def _check_tick(plan_g, gout_g, plan_o, gout_o, tag):
    bad = compare(plan_g, plan_o, "plan")
    if gout_g is not None:
        bad += compare(gout_g["status"], gout_o["status"], "grid.status")
        keep = (gout_o["status"] != 3) & (gout_o["status"] != 7)      # OVERFLOW / COST_RANGE: only the status is specified
        bad += compare(gout_g[keep], gout_o[keep], "grid")
    assert not bad, tag + "\n" + "\n".join(bad[:20])


def _mutate(sc, rng, t, n_obs, world):
    """New snapshot of tick t: every ego advances along its lane, every obstacle moves, some are replaced, and on some
    ticks scenes see fewer obstacles."""
    move_ego(sc, 1 + (t % 3), dlat=0.05 * ((t % 5) - 2))
    ob = sc["obs_pool"]
    ob["x"] += rng.uniform(-0.4, 0.4, len(ob))
    ob["y"] += rng.uniform(-0.4, 0.4, len(ob))
    k = rng.integers(0, len(ob), max(len(ob) // 10, 1))
    ob["x"][k] = rng.uniform(0.0, world, len(k))
    ob["y"][k] = rng.uniform(0.0, world, len(k))
    ego = sc["scene_in"]["loc"]["globalpoint"]
    n = len(sc["scene_in"])
    if n_obs:
        # keep the generator's rule: no obstacle within 3 m of its ego (the start cell would be inside a footprint)
        d2 = (ob["x"].reshape(n, n_obs) - ego["x"][:, None]) ** 2 + (ob["y"].reshape(n, n_obs) - ego["y"][:, None]) ** 2
        close = (d2 < 16.0).reshape(-1)
        ob["x"][close] += 9.0
        sc["scene_in"]["obs_n"][:] = n_obs if t % 4 != 3 else rng.integers(0, n_obs + 1, n)
    sc["scene_in"]["period_last"][:] = 90.0 + 10.0 * (t % 4)


@pytest.mark.parametrize("n,grid,n_obs,dynamic", [(320, 128, 24, 0), (304, 512, 64, 0), (40, 128, 24, 0), (288, 128, 32, 1)])
def test_streamed_ticks_every_tick_matches_oracle(dm, oracle, n, grid, n_obs, dynamic):
    """>= 12 ticks at n >= 300 (the three-stream tick) and at n = 40 (one-stream tick): new egos and obstacles every tick,
    no pp_sync in between, every tick's PlanOut + GridOut fetched asynchronously."""
    cfg = dm.default_config(grid)
    cfg["force_replan"] = 1
    cfg["dynamic_obstacles"] = dynamic
    world = float(grid) * float(cfg["cell"][0])
    sc = dm.gen_scenes(cfg, 9100, n, n_obs, junction_every=8)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
    pl.set_scenes(sc, with_motion=bool(dynamic))
    pl.set_state(sc["state"])
    rng = np.random.default_rng(77)
    T = 14
    ins, ids = [], []
    plans = [dm.pinned_empty(n, dm.PlanOut) for _ in range(T)]
    grids = [dm.pinned_empty(n, dm.GridOut) for _ in range(T)]
    for t in range(T):
        _mutate(sc, rng, t, n_obs, world)
        in_t, ob_t = dm.pinned_copy(sc["scene_in"]), dm.pinned_copy(sc["obs_pool"])
        mo_t = dm.pinned_copy(sc["mot_pool"]) if dynamic else None
        ins.append((in_t, ob_t, mo_t))
        pl.update_async(in_t, ob_t, mo_t)
        pl.tick()
        ids.append(pl.fetch_async(plans[t], grids[t]))
    assert ids == list(range(ids[0], ids[0] + T))
    for t in reversed(range(T)):                       # any order: each wait covers its own tick only
        assert pl.wait_tick(ids[t]) == 0
    st_o = sc["state"].copy()
    for t in range(T):
        sc_t = dict(sc, scene_in=np.array(ins[t][0]), obs_pool=np.array(ins[t][1]), mot_pool=np.array(ins[t][2]) if dynamic else None)
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, sc_t, st_o, n_threads=8)
        _check_tick(plans[t], grids[t], plan_o, gout_o, f"tick {t}")
    pl.sync()
    bad = compare(pl.get_state(), st_o, "state")
    assert not bad, "\n".join(bad[:20])
    # the synchronous getters still see the last tick
    _check_tick(pl.get_plan(), pl.get_grid_out(), plan_o, gout_o, "getters after the stream")
    pl.close()


def test_streamed_partial_updates_and_idle_ticks(dm, oracle):
    """Obstacle-only and ego-only updates carry the other half over; ticks without an update reuse the last snapshot; the
    downloads of PlanOut and GridOut can be asked for separately."""
    n, n_obs, grid = 300, 16, 128
    cfg = dm.default_config(grid)
    cfg["force_replan"] = 1
    sc = dm.gen_scenes(cfg, 9300, n, n_obs, junction_every=8)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
    pl.set_scenes(sc, with_motion=False)
    pl.set_state(sc["state"])
    rng = np.random.default_rng(5)
    st_o = sc["state"].copy()
    cur_in, cur_ob = sc["scene_in"].copy(), sc["obs_pool"].copy()
    keep, want, got = [], [], []
    for t in range(10):
        kind = ("obs", "ego", "none", "both", "none")[t % 5]
        if kind in ("obs", "both"):
            cur_ob = cur_ob.copy()
            cur_ob["x"] += rng.uniform(-0.5, 0.5, len(cur_ob))
        if kind in ("ego", "both"):
            tmp = dict(sc, scene_in=cur_in.copy())
            move_ego(tmp, 2)
            cur_in = tmp["scene_in"]
        a = dm.pinned_copy(cur_in) if kind in ("ego", "both") else None
        b = dm.pinned_copy(cur_ob) if kind in ("obs", "both") else None
        keep.append((a, b))
        if kind != "none":
            pl.update_async(a, b)
        pl.tick()
        p = dm.pinned_empty(n, dm.PlanOut) if t % 2 == 0 else None
        g = dm.pinned_empty(n, dm.GridOut) if t % 3 != 1 else None
        tid = pl.fetch_async(p, g) if (p is not None or g is not None) else None
        got.append((tid, p, g))
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, dict(sc, scene_in=cur_in, obs_pool=cur_ob, mot_pool=None), st_o, n_threads=8)
        want.append((plan_o, gout_o))
    for t, (tid, p, g) in enumerate(got):
        if tid is None:
            continue
        pl.wait_tick(tid)
        if p is not None:
            assert not compare(p, want[t][0], "plan"), f"tick {t}"
        if g is not None:
            assert not compare(g, want[t][1], "grid"), f"tick {t}"
    pl.sync()
    assert not compare(pl.get_state(), st_o, "state")
    pl.close()


def test_streamed_update_poisons_bad_scenes_instead_of_faulting(dm, oracle):
    """A slice outside its pool cannot be refused asynchronously: the scene runs with empty inputs, the tick reports it."""
    n, n_obs = 280, 8
    cfg = dm.default_config(128)
    sc = dm.gen_scenes(cfg, 9400, n, n_obs, junction_every=8)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
    pl.set_scenes(sc, with_motion=False)
    pl.set_state(sc["state"])
    bad_in = sc["scene_in"].copy()
    bad_in["obs_off"][5] = n * n_obs - 2                    # runs past the obstacle pool
    bad_in["lanes"]["cur_off"][9] = -7                      # negative lane offset
    bad_in["ref_n"][17] = 10 ** 9
    a, b = dm.pinned_copy(bad_in), dm.pinned_copy(sc["obs_pool"])
    plan, gout = dm.pinned_empty(n, dm.PlanOut), dm.pinned_empty(n, dm.GridOut)
    pl.update_async(a, b)
    pl.tick()
    tid = pl.fetch_async(plan, gout)
    with pytest.raises(dm.PlannerError, match="3 scene"):
        pl.wait_tick(tid)
    assert pl.wait_tick(tid, allow_poisoned=True) == 3
    # the oracle on the sanitised records (the three scenes emptied): every scene matches, poisoned ones included
    clean = bad_in.copy()
    for s in (5, 9, 17):
        for f in ("obs_off", "obs_n", "ref_off", "ref_n"):
            clean[f][s] = 0
        for f in ("cur_off", "cur_n", "left_off", "left_n", "right_off", "right_n"):
            clean["lanes"][f][s] = 0
    st_o = sc["state"].copy()
    plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, dict(sc, scene_in=clean, mot_pool=None), st_o, n_threads=8)
    _check_tick(plan, gout, plan_o, gout_o, "poisoned tick")
    # a good update afterwards: nothing sticks
    a2 = dm.pinned_copy(sc["scene_in"])
    pl.update_async(a2, b)
    pl.tick()
    tid = pl.fetch_async(plan, gout)
    assert pl.wait_tick(tid) == 0
    plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, dict(sc, mot_pool=None), st_o, n_threads=8)
    _check_tick(plan, gout, plan_o, gout_o, "tick after the poisoned one")
    # argument checks
    with pytest.raises(dm.PlannerError, match="resident count"):
        dm._check(pl.lib.pp_update_async(pl.h, n - 1, a.ctypes.data, b.ctypes.data, None, n * n_obs))
    with pytest.raises(dm.PlannerError, match="no pp_fetch_async"):
        pl.tick()
        pl.wait_tick(pl.tick_id())
    pl.close()


def test_streamed_egos_on_the_resident_map(dm, oracle):
    """After pp_set_map / pp_set_egos an update carries only egos + obstacles: the lane views are derived on the upload
    stream, a road off the map poisons its scene."""
    import map_scenes as ms
    cfg = dm.default_config(128)
    m = ms.build_map(dm, n_roads=5)
    n, n_obs = 272, 12
    sc = ms.make_egos(dm, cfg, m, n, n_obs)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs, max_lane_pts_total=len(m["points"]),
                    max_ref_pts_total=max(len(m["jpoints"]), 1))
    pl.set_map(m)
    pl.set_egos(sc, with_motion=False)
    pl.set_state(sc["state"])
    rng = np.random.default_rng(9)
    st_o = sc["state"].copy()
    T = 12
    res, want, keep = [], [], []
    for t in range(T):
        si = sc["scene_in"].copy()
        # every ego on a road advances along its lane (the point ids move, position from the map); the ones inside a junction creep
        lane0 = m["road_first_lane"][si["loc"]["road_num"] - 1] + si["loc"]["lane_num"] - 1
        for s in range(n):
            if si["loc"]["pos"][s] == 2:
                si["loc"]["globalpoint"]["x"][s] += 0.05
                continue
            ml = m["lanes"][lane0[s]]
            k = si["loc"]["lane_num"][s] - 1
            nid = min(int(si["loc"]["id"][s][k]) + 1 + t % 2, int(ml["n_points"]) - 2)
            si["loc"]["id"][s][:] = nid
            p = m["points"][int(ml["point_off"]) + nid]
            si["loc"]["globalpoint"]["x"][s], si["loc"]["globalpoint"]["y"][s], si["loc"]["globalpoint"]["dir"][s] = p["x"], p["y"], p["dir"]
        if t == 7:
            si["loc"]["road_num"][3] = 99            # off the map
        sc["scene_in"] = si
        ob = sc["obs_pool"].copy()
        ob["x"] += rng.uniform(-0.3, 0.3, len(ob))
        sc["obs_pool"] = ob
        a, b = dm.pinned_copy(si), dm.pinned_copy(ob)
        keep.append((a, b))
        pl.update_async(a, b)
        pl.tick()
        p_, g_ = dm.pinned_empty(n, dm.PlanOut), dm.pinned_empty(n, dm.GridOut)
        res.append((pl.fetch_async(p_, g_), p_, g_))
        si_ok = si.copy()
        if t == 7:
            si_ok["loc"]["road_num"][3] = 1
            si_ok["loc"]["lane_num"][3] = 1
        w = ms.resolve(dm, m, si_ok)
        if t == 7:
            w["loc"][3] = si["loc"][3]
            for f in ("cur_off", "cur_n", "left_off", "left_n", "right_off", "right_n", "lane_sum", "lanechg_attribute"):
                w["lanes"][f][3] = 0
            w["lanes"]["lane_width"][3] = 0
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, dict(sc, scene_in=w, mot_pool=None), st_o, n_threads=8)
        want.append((plan_o, gout_o))
        if t == 7:
            sc["scene_in"]["loc"]["road_num"][3] = si["loc"]["road_num"][4]
            sc["scene_in"]["loc"]["lane_num"][3] = si["loc"]["lane_num"][4]
            sc["scene_in"]["loc"]["id"][3] = si["loc"]["id"][4]
    for t, (tid, p_, g_) in enumerate(res):
        assert pl.wait_tick(tid, allow_poisoned=True) == (1 if t == 7 else 0), t
        _check_tick(p_, g_, want[t][0], want[t][1], f"tick {t}")
    pl.sync()
    assert not compare(pl.get_state(), st_o, "state")
    pl.close()


def test_streamed_published_records_only(dm, oracle):
    """pp_fetch_published_async: what the reference publishes per tick - PlanningOut (SetUdpSendCtrl) and PlanningStatus
    (SetPlanningStatus) - as two contiguous arrays, strided copies out of the PlanOut records."""
    n, n_obs = 300, 16
    cfg = dm.default_config(128)
    cfg["force_replan"] = 1
    sc = dm.gen_scenes(cfg, 9500, n, n_obs, junction_every=8)
    pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
    pl.set_scenes(sc, with_motion=False)
    pl.set_state(sc["state"])
    st_o = sc["state"].copy()
    got, keep = [], []
    for t in range(8):
        move_ego(sc, 1)
        a, b = dm.pinned_copy(sc["scene_in"]), dm.pinned_copy(sc["obs_pool"])
        keep.append((a, b))
        pl.update_async(a, b)
        pl.tick()
        res, show = dm.pinned_empty(n, dm.PlanningOut), dm.pinned_empty(n, dm.PlanningStatus)
        grid = dm.pinned_empty(n, dm.GridOut) if t % 2 == 0 else None
        tid = pl.fetch_published_async(res, show if t % 3 else None, grid)
        plan_o, gout_o, _ = oracle.plan_tick_batch(cfg, dict(sc, mot_pool=None), st_o, n_threads=8)
        got.append((tid, res, show if t % 3 else None, grid, plan_o, gout_o))
    for t, (tid, res, show, grid, plan_o, gout_o) in enumerate(got):
        assert pl.wait_tick(tid) == 0
        assert not compare(res, plan_o["result"], "result"), t
        if show is not None:
            assert not compare(show, plan_o["show"], "show"), t
        if grid is not None:
            assert not compare(grid, gout_o, "grid"), t
    pl.close()
