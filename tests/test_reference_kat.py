"""CPU: known-answer tests for the functions whose bodies ARE in the reference, with expected values
derived by hand from the cited source lines (the reference ships no tests or vectors, and cannot be
built here — these are analytic answers, not reference outputs: parity stays "unpinned")."""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cfg(dm):
    return dm.default_config(128)


def test_GetRoadAngle_quadrants(oracle, cfg):          # Planning.cpp:719-750: degrees CCW from east, [0,360)
    A = lambda b: oracle.GetRoadAngle(cfg, (0.0, 0.0), b)
    assert A((1.0, 0.0)) == 0.0
    assert A((1.0, 1.0)) == pytest.approx(45.0, abs=1e-12)
    assert A((0.0, 2.0)) == pytest.approx(90.0, abs=1e-12)            # |dx| < EPSILON, dy > 0  (:726-728)
    assert A((-1.0, 1.0)) == pytest.approx(135.0, abs=1e-12)          # bx < ax: + PI           (:739-741)
    assert A((-1.0, 0.0)) == pytest.approx(180.0, abs=1e-12)
    assert A((-1.0, -1.0)) == pytest.approx(225.0, abs=1e-12)
    assert A((0.0, -3.0)) == pytest.approx(270.0, abs=1e-12)          # |dx| < EPSILON, dy < 0  (:730-732)
    assert A((1.0, -1.0)) == pytest.approx(315.0, abs=1e-12)          # 4th quadrant: + 2 PI    (:743-746)
    assert A((1e-7, 1e-7)) == 0.0                                     # both below EPSILON      (:722-723)
    assert A((1e-7, 1.0)) == pytest.approx(90.0, abs=1e-12)           # EPSILON branch, not atan


def test_GetAngleErr_wrap(oracle):                      # Planning.cpp:760-786: (-180, 180]
    E = oracle.GetAngleErr
    assert E(10.0, 30.0) == 20.0
    assert E(350.0, 10.0) == 20.0                       # dir1 >= 180, diff = -340 <= -180 -> +360
    assert E(10.0, 350.0) == -20.0                      # dir1 < 180, diff = 340 > 180 -> -360
    assert E(0.0, 180.0) == 180.0                       # diff == 180 stays (<=)
    assert E(180.0, 0.0) == 180.0                       # diff == -180 is not > -180 -> +360
    assert E(200.0, 100.0) == -100.0


def test_GetLatDis_sign_and_epsilon(oracle, cfg):       # Planning.cpp:686-709: left of the path is positive
    L = oracle.GetLatDis
    assert L(cfg, (5.0, 2.0), (0.0, 0.0), (10.0, 0.0)) == pytest.approx(2.0, abs=1e-15)      # left of +x heading
    assert L(cfg, (5.0, -3.0), (0.0, 0.0), (10.0, 0.0)) == pytest.approx(-3.0, abs=1e-15)
    assert L(cfg, (2.0, 5.0), (0.0, 0.0), (0.0, 10.0)) == -2.0      # vertical segment branch (:697): right of +y heading
    assert L(cfg, (-2.0, 5.0), (0.0, 0.0), (0.0, 10.0)) == 2.0
    assert L(cfg, (5.0, 1e-7), (0.0, 0.0), (10.0, 0.0)) == 0.0      # below EPSILON -> exactly 0 (:699-702)
    assert L(cfg, (0.0, 1.0), (0.0, 0.0), (1.0, 1.0)) == pytest.approx(math.sqrt(0.5), rel=1e-15)


def test_Calculate_aim_dis(oracle, dm, cfg):            # Planning.cpp:242-290; FLOAT results
    loc = np.zeros(1, dm.LocationOut)
    for v, want in ((0.0, 10.0), (60.0, 40.0), (18.0, float(np.float32(18.0 / 3.6 * 5 + 4)))):
        loc["velocity"] = v
        assert oracle.Calculate_aim_dis(cfg, loc) == (want, want)
    loc["pos"] = 1
    assert oracle.Calculate_aim_dis(cfg, loc) == (15.0, 15.0)
    loc["pos"] = 2
    assert oracle.Calculate_aim_dis(cfg, loc) == (10.0, 10.0)
    loc["pos"] = 7
    assert oracle.Calculate_aim_dis(cfg, loc) == (0.0, 0.0)          # default: stays at the initial 0 (:250-251,287)


def test_CalculateRadius(oracle, dm):                   # Planning.cpp:1000-1019
    pts = np.zeros(200, dm.GlobalPoint2D)
    th = np.arange(200) * 0.01
    R = 25.0
    pts["x"], pts["y"] = R * np.sin(th), R * (1 - np.cos(th))
    assert oracle.CalculateRadius(pts, 10, 18) == pytest.approx(R, rel=1e-9)     # circumradius of 3 points on a circle
    pts["x"], pts["y"] = np.arange(200) * 0.5, 0.0
    assert oracle.CalculateRadius(pts, 10, 18) == 1000.0                          # sinA < 0.001 (:1010-1012)
    assert oracle.CalculateRadius(pts, 195, 203) == 1000.0                        # front id 203 fenced to 199
    # middle index uses integer division before round(): (10+19)/2 = 14, not 15 (:1003)
    pts["y"] = (np.arange(200) == 14) * 1.0
    r14 = oracle.CalculateRadius(pts, 10, 19)
    assert r14 != 1000.0 and np.isfinite(r14)


def test_SpeedPlanning_branches(oracle, dm):            # Planning.cpp:888-990
    dec, loc = np.zeros(1, dm.DecisionOutPod), np.zeros(1, dm.LocationOut)
    dec["velocity_expect"] = 10.0
    S = lambda flag, lon, far=30.0: oracle.SpeedPlanning(flag, dec, loc, lon, 0.0, far)
    assert S(0, 999.0) == (10.0, 0, 0.0)                                  # no obstacle -> expected speed
    assert S(1, 20.0) == (pytest.approx(3 + (20 - 9) / (30 - 9) * 7), 0, 0.0)       # lon-4 > 9
    assert S(1, 12.0) == (3.0, 0, 0.0)                                    # 5 < lon-4 <= 9
    assert S(1, 9.0) == (0.0, 1, -3.0)                                    # AEB
    assert S(1, 13.0) == (3.0, 0, 0.0) and S(1, 13.000001)[0] > 3.0       # boundary lon-4 > 9 is strict
    assert math.isinf(S(1, 20.0, far=9.0)[0])                             # division by (faraim-9), quirk :898
    loc["pos"] = 5
    assert oracle.SpeedPlanning(1, dec, loc, 1.0, 0.0, 30.0, init=(7.0, 1, -1.0)) == (7.0, 1, -1.0)   # default: untouched


def test_GetVhclLocalState_and_UpdatePlanJudge(oracle, dm, cfg):     # Planning.cpp:623-676, 797-832
    last = np.zeros(200, dm.GlobalPoint2D)
    last["x"] = np.arange(200) * 0.5
    loc = np.zeros(1, dm.LocationOut)
    loc["globalpoint"]["x"], loc["globalpoint"]["y"], loc["globalpoint"]["dir"] = 10.1, 0.3, 20.0
    lat, derr, mid, fid, rem = oracle.GetVhclLocalState(cfg, loc, last)
    assert (mid, fid) == (20, 28)                                        # nearest point, +8 (:649)
    assert lat == pytest.approx(0.3, abs=1e-12) and derr == 20.0         # left of the path; heading error
    assert rem == pytest.approx((199 - 28) * 0.5, abs=1e-9)              # arc length from the front id (:668-671)
    loc["globalpoint"]["x"] = 10.25                                      # tie between points 20 and 21: strict < keeps the first
    assert oracle.GetVhclLocalState(cfg, loc, last)[2] == 20
    loc["globalpoint"]["x"] = 1e5                                        # farther than 9999 from every point: id keeps its old value
    assert oracle.GetVhclLocalState(cfg, loc, last, near_id_in=7)[2] == 7
    loc["globalpoint"]["x"] = 99.4                                       # last point: index 199 -> segment 198-199 (:656-659)
    assert oracle.GetVhclLocalState(cfg, loc, last)[2:4] == (199, 207)


def test_tick_counters_and_first_tick(oracle, dm):      # Planning.cpp:124-128, 216-223
    cfg = dm.default_config(128)
    cfg["grid_stage"] = 0
    sc = dm.gen_scenes(cfg, 0, 1, 8, junction_every=0)
    st = sc["state"].copy()
    counts, cnts = [], []
    for t in range(205):
        plan, _, _ = oracle.plan_tick_batch(cfg, sc, st, want_grid=False)
        counts.append(int(st["count"][0]))
        cnts.append(int(plan["result"]["cnt"][0]))
    assert counts[:3] == [1, 2, 3] and counts[99] == 100 and counts[100] == 1 and counts[101] == 2   # BYTE count wraps 101 -> 1
    assert cnts[0] == 0 and cnts[1] == 1 and cnts[100] == 0 and cnts[101] == 1                       # cnt = count % 100 before the increment
    # the published points are every 2nd path point (:180-183) and their WGS84 image (:205-212)
    assert plan["show"]["path_points"][0].tobytes() == plan["road_points"][0][::2].tobytes()
    lat = cfg["wgs_lat0"][0] + plan["road_points"][0]["y"][::2] * cfg["wgs_deg_per_m_lat"][0]
    assert np.array_equal(plan["result"]["pnts"][0]["x"], lat)
