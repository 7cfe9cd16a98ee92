"""The C++ class surface - CDecision::decide -> CPlanning::plan, the literal drop-in for the reference's two threads
(Decision.cpp:172-205, Planning.cpp:114-223; Planning.h:38-85) - against the oracle, tick by tick.

Drives libdmpp_host.so through its `extern "C"` entry points (host/dmpp_host.cpp: one or two method calls on the
singletons each).  Compared every tick: PlanningOut, PlanningStatus, the 200 road points, the public members of
CPlanning (Planning.h:42-52), DecisionOut with its refpath, the six corridor results, and the cross-tick state both
classes keep.  PARITY UNPINNED versus the reference itself (oracle/dmpp_oracle.h)."""
import os

import numpy as np
import pytest

from parity_util import compare, move_ego

pytestmark = pytest.mark.gpu

PLANNING_STATE = ("last_Bpoints", "aimpoint_far", "aimpoint_near", "path_lat_dis", "remain_dis", "path_dir_err", "brakespeed", "des_acc",
                  "faraim_dis", "nearaim_dis", "path_near_id", "path_front_near_id", "his_behavior", "afresh_planning", "afresh_cause",
                  "acc_flag", "count")
DECISION_STATE = ("z_behavior", "z_light_status", "z_target_lanenum", "z_target_roadnum", "z_behavior_to_dlg", "z_segment_lanechg_status",
                  "z_segment_obsavoid_status", "d_his_behavior", "d_his_light_status", "d_his_target_lanenum", "obsavoid_time",
                  "no_obsaviod_time", "frontobs_time", "z_velocity_expect", "leftlight_time", "rightlight_time")


@pytest.fixture(scope="module")
def surface(dm):
    from dmpp_amd_pkg import host_surface
    return host_surface


def _one(sc, s, dm):
    """Scene `s` of a batch as a batch of one with its own pools (offsets rebased)."""
    si = sc["scene_in"][s:s + 1].copy()
    lv = si["lanes"][0]
    lo = min(int(lv["cur_off"]), int(lv["left_off"]), int(lv["right_off"]))
    hi = max(int(lv["cur_off"]) + int(lv["cur_n"]), int(lv["left_off"]) + int(lv["left_n"]), int(lv["right_off"]) + int(lv["right_n"]))
    out = dict(scene_in=si, lane_pool=sc["lane_pool"][lo:hi].copy(), attr_pool=sc["attr_pool"][lo:hi].copy(),
               ref_pool=sc["ref_pool"][int(si["ref_off"][0]):int(si["ref_off"][0]) + int(si["ref_n"][0])].copy(),
               obs_pool=sc["obs_pool"][int(si["obs_off"][0]):int(si["obs_off"][0]) + max(int(si["obs_n"][0]), 1)].copy(),
               state=sc["state"][s:s + 1].copy(), n_obs=int(si["obs_n"][0]))
    out["mot_pool"] = np.zeros(len(out["obs_pool"]), dm.ObMotion)
    for k in ("cur_off", "left_off", "right_off"):
        si["lanes"][k] -= lo
    si["ref_off"] = 0
    si["obs_off"] = 0
    return out


def _run(dm, oracle, surface, cfg, sc, n_ticks, mutate=None, tag=""):
    hs = surface.HostSurface(dm, cfg)
    hs.set_scene_map(sc, 0)
    st_o = sc["state"].copy()
    seen = set()
    for t in range(n_ticks):
        if mutate is not None:
            mutate(sc, t, st_o)
            hs.set_scene_map(sc, 0)
        got = hs.tick(sc, 0)
        plan_o, _, _, _, _ = oracle.plan_tick_one(cfg, sc, 0, st_o)
        ref_o = oracle.last_refpath()
        where = f"{tag} tick {t}"
        bad = (compare(got["result"], plan_o["result"], "PlanningOut") + compare(got["show"], plan_o["show"], "PlanningStatus")
               + compare(got["road_points"], plan_o["road_points"], "road_points") + compare(got["around"], plan_o["around"], "around")
               + compare(got["dec"], plan_o["dec"], "DecisionOut"))
        assert not bad, where + "\n" + "\n".join(bad[:10])
        assert len(got["refpath"]) == len(ref_o) == int(plan_o["dec"]["refpath_n"]), where
        assert not compare(got["refpath"], ref_o, "DecisionOut.refpath"), where
        m = got["members"]
        for f in ("path_lat_dis", "remain_dis", "path_dir_err", "brakespeed", "des_acc"):
            a, b = m[f], float(st_o[f][0])
            assert a == b or abs(a - b) <= 1e-9 + 1e-6 * max(abs(a), abs(b)) or (np.isnan(a) and np.isnan(b)), (where, f, a, b)
        for f in ("afresh_planning", "afresh_cause", "path_near_id", "path_front_near_id", "his_behavior", "acc_flag"):
            assert m[f] == int(st_o[f][0]), (where, f, m[f], int(st_o[f][0]))
        stp, std = hs.planning_state(), hs.decision_state()
        bad = []
        for f in PLANNING_STATE:
            bad += compare(stp[f], st_o[f], "CPlanning." + f)
        for f in DECISION_STATE:
            bad += compare(std[f], st_o[f], "CDecision." + f)
        assert not bad, where + "\n" + "\n".join(bad[:10])
        seen.add((int(sc["scene_in"]["loc"]["pos"][0]), int(plan_o["dec"]["behavior"]), int(st_o["afresh_cause"][0])))
    return seen


def test_decide_then_plan_generated_scenes(dm, oracle, surface):
    """Road, pre-junction and junction scenes with obstacles near the lane, nine ticks each with the ego advancing: every
    `LocationOut.pos` value, the lateral sweep, replans by several causes."""
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    batch = dm.gen_scenes(cfg, 900, 24, 40, junction_every=3)
    seen = set()
    for s in range(24):
        sc = _one(batch, s, dm)

        def mutate(sc_, t, st_):
            if t and int(sc_["scene_in"]["loc"]["pos"][0]) == 0:
                move_ego(sc_, 3, dlat=0.05 * t)
        seen |= _run(dm, oracle, surface, cfg, sc, 9, mutate, tag=f"scene {s}")
    assert {p for p, _, _ in seen} == {0, 1, 2}
    assert len({c for _, _, c in seen}) >= 3, seen               # no replan, and at least two replan causes


def test_decide_then_plan_lane_changes(dm, oracle, surface):
    """The lane-change rule tree on the class surface: the hand-built scenes of tests/lanechange_scenes.py, eight ticks each,
    the localisation moving to the target lane once a change has started."""
    import lanechange_scenes as lcs
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    behaviours = set()
    for k, kw in enumerate(lcs.SCENARIOS):
        sc = lcs.make_scene(dm, cfg, **kw)

        def mutate(sc_, t, st_):          # two ticks after a change has started the localisation reports the target lane
            lane, target = int(sc_["scene_in"]["loc"]["lane_num"][0]), int(st_["z_target_lanenum"][0])
            if t >= 2 and int(st_["z_segment_lanechg_status"][0]) == 1 and target != lane and 1 <= target <= 3:
                lcs.switch_lane(dm, sc_, target)
        hs_seen = _run(dm, oracle, surface, cfg, sc, 8, mutate, tag=f"scenario {k}")
        behaviours |= {b for _, b, _ in hs_seen}
    assert {1, 2, 3} <= behaviours, behaviours


def test_plan_with_the_grid_stage(dm, oracle, surface):
    """CPlanning::plan(..., GridOut*) runs the grid stage on the frame given by SetGridFrame: the search and the scored
    candidates of the class surface against the oracle's tick on the same scene."""
    cfg = dm.default_config(256)
    batch = dm.gen_scenes(cfg, 4100, 6, 48, junction_every=0)
    for s in range(6):
        sc = _one(batch, s, dm)
        hs = surface.HostSurface(dm, cfg)
        hs.set_scene_map(sc, 0)
        st_o = sc["state"].copy()
        for t in range(3):
            got = hs.tick(sc, 0, want_grid=True)
            plan_o, gout_o, _, _, _ = oracle.plan_tick_one(cfg, sc, 0, st_o)
            bad = compare(got["grid"], gout_o, "GridOut") + compare(got["result"], plan_o["result"], "PlanningOut")
            assert not bad, f"scene {s} tick {t}\n" + "\n".join(bad[:10])


def test_class_surface_latency_helper(dm, surface):
    cfg = dm.default_config(512)
    sc = dm.gen_scenes(cfg, 0, 1, 64, junction_every=0)
    ms = surface.p50_plan_ms(dm, cfg, sc, ticks=20)
    assert 0 < ms < 50
