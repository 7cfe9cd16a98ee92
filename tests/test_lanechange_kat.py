"""CPU: known-answer sequences for the lane-change rule tree (Decision.cpp:1011-1772) and Nav_LaneChange
(Decision.cpp:685-738).  Every expected value is derived by hand from the cited reference lines on the
straight-lane scenes of lanechange_scenes.py (all distances exact) — analytic answers, not reference
outputs: the reference cannot be built here, parity stays "unpinned".

A state tuple is (behavior, target_lanenum, light_status, lanechg_status, behavior_to_dlg) after a tick."""
import numpy as np
import pytest

import lanechange_scenes as lcs


@pytest.fixture(scope="module")
def cfg(dm):
    c = dm.default_config(128)
    c["grid_stage"] = 0
    return c


def run(oracle, cfg, sc, ticks, st=None):
    st = sc["state"].copy() if st is None else st
    seq, plan = [], None
    for _ in range(ticks):
        plan, _, _ = oracle.plan_tick_batch(cfg, sc, st, want_grid=False)
        seq.append(lcs.behaviour(st))
    return seq, st, plan


def test_nav_lanechange_directions(oracle, dm, cfg):            # Decision.cpp:685-738, 498-538
    def navi(lane_num, out, lane_sum=5):
        sc = lcs.make_scene(dm, cfg, lane_num=lane_num, lane_sum=lane_sum, map_attr=0, out_lanes=out)
        _, _, plan = run(oracle, cfg, sc, 1)
        return int(plan["navi_lanechg"][0]), int(plan["navi_lanechg_times"][0])
    assert navi(2, (2,)) == (0, 0)                  # ego lane is an exit lane (:696-703)
    assert navi(2, (3, 4)) == (2, 1)                # below out_lane_no[0] -> right, fewest changes 3-2 (:722-726)
    assert navi(1, (3, 4)) == (2, 2)
    assert navi(4, (1, 2)) == (1, 2)                # above the largest exit lane -> left, 4-2 (:728-732)
    assert navi(3, (4, 1)) == (2, 254)              # "min" is out_lane_no[0] = 4 (:708): right; 1 - 3 = -2 as a BYTE (:498,522)
    assert navi(3, (2, 4)) == (0, 0)                # between: "should not happen" branch (:733-736)
    assert navi(1, ()) == (0, 0)                    # no exit lanes: min 0, max stays 1 (:706-707)
    assert navi(2, ()) == (1, 5)                    # 2 > max(=1): left; the times loop never runs, 5 stays (:500)


def test_navigation_left_change_then_arrival(oracle, dm, cfg):
    # lane 2 of 3, map allows left, navigation exits via lane 1, nothing around: corridors report 999 m.
    # :1030-1035 left light on, timer += 700 per tick; :1044 needs > 2000 ms -> change on the 3rd tick.
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(1,), period=700.0)
    seq, st, plan = run(oracle, cfg, sc, 4)
    assert seq[0] == (1, 2, 1, 0, 2) and seq[1] == (1, 2, 1, 0, 2)
    assert seq[2] == (2, 1, 1, 1, 2)                            # behavior 2, target lane 1, changing (:1046-1048)
    assert seq[3] == (2, 1, 1, 1, 9)                            # :1760-1770 copies the last published decision
    assert float(st["leftlight_time"][0]) == 2100.0
    assert int(plan["navi_lanechg"][0]) == 1 and int(plan["navi_lanechg_times"][0]) == 1
    assert int(plan["dec"]["behavior"][0]) == 2 and int(plan["dec"]["light"][0]) == 1
    # the published reference path is the left lane's forward slice while changing (RefPath :1801-1816)
    assert int(plan["dec"]["refpath_n"][0]) == 120
    # localisation now reports lane 1: target reached (:1763-1767), but the decision of that tick is still the
    # history copy (:1768-1770); one tick later the tree is back to lane keeping with no obstacle ahead (:1749-1756)
    lcs.switch_lane(dm, sc, 1)
    seq2, st, _ = run(oracle, cfg, sc, 2, st)
    assert seq2[0] == (2, 1, 1, 0, 9)
    assert seq2[1] == (1, 1, 1, 0, 8)                           # nobody switches the light off on this path


def test_navigation_blocked_by_map_or_traffic(oracle, dm, cfg):
    # navigation wants the right lane but the map only allows left: :1135-1141
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(3,), period=700.0)
    assert run(oracle, cfg, sc, 3)[0][-1] == (1, 2, 0, 0, 4)
    # navigation left, a car 12 m behind in the left lane: LR = 12 <= 15 (:1041) -> never changes
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(1,), period=900.0, obstacles=[(1, -12.0)])
    seq, st, plan = run(oracle, cfg, sc, 5)
    assert float(plan["around"][0][3]["Ob_Pose"]["dis_lng"]) == 12.0
    assert all(s == (1, 2, 1, 0, 2) for s in seq) and float(st["leftlight_time"][0]) == 4500.0   # this timer is not capped
    # navigation left, left-front car at 30 m, own-lane car at 25 m: 30 > 25 + 10 fails and 30 > 40 fails (:1038)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(1,), period=900.0, obstacles=[(1, 30.0), (2, 25.0)])
    assert run(oracle, cfg, sc, 4)[0][-1] == (1, 2, 1, 0, 2)
    # ... but with the own-lane car at 15 m, 30 > 15 + 10 holds
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(1,), period=900.0, obstacles=[(1, 30.0), (2, 15.0)])
    assert run(oracle, cfg, sc, 3)[0][-1] == (2, 1, 1, 1, 2)


def test_navigation_right_change_quirk(oracle, dm, cfg):
    # :1092-1096 writes lanechg_status where light_status is meant: the right light never comes on, so the
    # timer restarts every tick and only a single period >= 2000 ms can trigger the change (:1106 uses >=)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=2, out_lanes=(3,), period=700.0)
    seq, st, _ = run(oracle, cfg, sc, 6)
    assert all(s == (1, 2, 0, 0, 3) for s in seq) and float(st["rightlight_time"][0]) == 700.0
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=2, out_lanes=(3,), period=2000.0)
    assert run(oracle, cfg, sc, 1)[0][0] == (3, 3, 0, 1, 3)


def test_front_obstacle_left_change(oracle, dm, cfg):
    # no navigation demand, a car 10 m ahead (< 25, :1149), map allows left, 125 m of attribute-1 lane ahead
    # (a) lane 1 is not an exit lane -> "must come back" branch (:1176-1208): its light test reads the member
    #     z_light_status that :308 overwrites, so the timer restarts every tick; 800 ms never passes 1500
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,), period=800.0, obstacles=[(2, 10.0)])
    seq, st, _ = run(oracle, cfg, sc, 6)
    assert seq[0] == (1, 2, 0, 0, 0) and seq[1] == (1, 2, 0, 0, 0)      # frontobs_time 1, 2: :1741-1746
    assert all(s == (1, 2, 0, 0, 5) for s in seq[2:])
    assert float(st["leftlight_time"][0]) == 800.0 and int(st["frontobs_time"][0]) == 3
    #     a 1600 ms period passes 1500 at once (:1250)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)])
    seq, st, _ = run(oracle, cfg, sc, 3)
    assert seq[2] == (2, 1, 0, 1, 5) and int(st["frontobs_time"][0]) == 0
    #     fewer than 60 m of attribute-1 lane: no change (:1190)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=120)
    assert run(oracle, cfg, sc, 4)[0][-1] == (1, 2, 0, 0, 5)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=121)
    assert run(oracle, cfg, sc, 3)[0][-1] == (2, 1, 0, 1, 5)                # 121 points = 60.5 m > 60
    # (b) lane 1 is an exit lane too (:1209-1239): the light really comes on and the timer accumulates
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(1, 2), period=800.0, obstacles=[(2, 10.0)])
    seq, st, _ = run(oracle, cfg, sc, 4)
    assert seq[2] == (1, 2, 1, 0, 5) and seq[3] == (2, 1, 1, 1, 5)
    assert float(st["leftlight_time"][0]) == 1600.0
    #     attribute 3 ahead fails `attr == 1` of (a) but passes `attr & 1` of (b)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3)
    assert run(oracle, cfg, sc, 4)[0][-1] == (1, 2, 0, 0, 5)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3)
    assert run(oracle, cfg, sc, 3)[0][-1] == (2, 1, 1, 1, 5)
    # (c) leftmost lane: nothing to change into (:1287-1292)
    sc = lcs.make_scene(dm, cfg, lane_num=1, map_attr=1, out_lanes=(1,), period=1600.0, obstacles=[(1, 10.0)])
    assert run(oracle, cfg, sc, 4)[0][-1] == (1, 1, 0, 0, 0)


def test_front_obstacle_right_change(oracle, dm, cfg):
    # map allows right; `attr & 0x02 == 0x02` parses as attr & 1 (:1316), so attribute 2 ahead yields 0 m
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)])
    assert run(oracle, cfg, sc, 4)[0][-1] == (1, 2, 0, 0, 6)
    # attribute 3 ahead passes; lane 1 not an exit lane -> > 50 m branch, right light, change on the 3rd tick
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3)
    seq, st, plan = run(oracle, cfg, sc, 3)
    assert seq[2] == (3, 3, 2, 1, 6) and int(st["frontobs_time"][0]) == 0
    assert int(plan["dec"]["refpath_n"][0]) == 120              # the right lane's forward slice
    # a car 8 m behind in the right lane blocks it (:1376: RR > 10)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0), (3, -8.0)], attr_ahead=3)
    assert run(oracle, cfg, sc, 4)[0][-1] == (1, 2, 2, 0, 6)
    # lane 1 an exit lane -> the > 10 m branch, which switches the LEFT light on (:1356-1360)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=2, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3, attr_run=21)
    assert run(oracle, cfg, sc, 3)[0][-1] == (3, 3, 1, 1, 6)


def test_both_sides_branch_never_fires(oracle, dm, cfg):
    # map allows both: every timer is capped at 2000 and then tested with > 2000 (:1553-1564 ...), and with
    # attribute 3 LoadRefPath loads no right paths (:636), so this branch cannot start a change.  Its
    # side effect at :1640-1644 (lanechg_status written for light_status) makes the state oscillate.
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=3, out_lanes=(2,), period=2500.0, obstacles=[(2, 10.0)])
    seq, st, plan = run(oracle, cfg, sc, 6)
    assert float(plan["around"][0][4]["Ob_Pose"]["dis_lng"]) == 0.0          # right-front corridor never searched
    assert seq[2] == (1, 2, 0, 1, 0)                            # keep lane, but lanechg_status 1
    assert seq[3] == (1, 2, 0, 0, 9)                            # :1760-1767 "arrived"
    assert seq[4] == (1, 2, 0, 1, 9) and seq[5] == (1, 2, 0, 0, 9)
    assert float(st["leftlight_time"][0]) == 2000.0
    # right-only candidate (leftmost lane): light 2 comes on, RF = 0 fails RF > F + 10, nothing is assigned
    sc = lcs.make_scene(dm, cfg, lane_num=1, map_attr=3, out_lanes=(1,), period=900.0, obstacles=[(1, 10.0)])
    seq, st, _ = run(oracle, cfg, sc, 4)
    assert seq[2] == (1, 1, 2, 0, 0) and seq[3] == (1, 1, 2, 0, 0)
    assert float(st["leftlight_time"][0]) == 1800.0


def test_stage_switch_and_counters(oracle, dm, cfg):
    # lanechg_stage = 0 keeps the earlier behaviour: counters reset (:1014-1015), decision passes through
    c = cfg.copy()
    c["lanechg_stage"] = 0
    sc = lcs.make_scene(dm, c, lane_num=2, map_attr=1, out_lanes=(1,), period=2500.0)
    st = sc["state"].copy()
    st["obsavoid_time"], st["no_obsaviod_time"] = 5, 6
    seq, st, _ = run(oracle, c, sc, 2, st)
    assert seq[-1] == (1, 2, 0, 0, 0) and int(st["obsavoid_time"][0]) == 0 and int(st["no_obsaviod_time"][0]) == 0
    # an obstacle-free road with no navigation demand: frontobs_time reset, panel code 8 (:1749-1756)
    sc = lcs.make_scene(dm, cfg, lane_num=2, map_attr=1, out_lanes=(2,))
    st = sc["state"].copy()
    st["frontobs_time"] = 3
    seq, st, _ = run(oracle, cfg, sc, 1, st)
    assert seq[0] == (1, 2, 0, 0, 8) and int(st["frontobs_time"][0]) == 0
