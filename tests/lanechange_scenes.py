"""Hand-built road-segment scenes for the lane-change rule tree (Decision.cpp:1011-1772).

Three parallel straight lanes along +x at 0.5 m spacing (so every arc length is an exact binary
fraction), the ego on point 50 of its lane, obstacles placed on lane points so that their
longitudinal distances are known without running any code.  Used by the CPU known-answer tests
(oracle) and by the GPU parity tests (device vs oracle)."""
import numpy as np

LANE_W = 3.75
EGO_ID = 50


def make_scene(dm, cfg, lane_num=2, lane_sum=3, map_attr=1, out_lanes=(2,), period=100.0,
               attr_ahead=None, attr_run=250, obstacles=()):
    """obstacles: iterable of (lane_no, metres_ahead) — lane_no is 1-based, negative metres = behind the ego.
    attr_ahead: lanechg_attribute of the `attr_run` points after the ego point (default: map_attr)."""
    n_obs = len(obstacles)
    sc = dm.gen_scenes(cfg, 0, 1, n_obs, junction_every=0)
    n = dm.GEN_LANE_PTS
    x = 100.0 + 0.5 * np.arange(n)
    lane_y = {k: 200.0 - LANE_W * (k - 1) for k in range(1, lane_sum + 1)}     # lane 1 is the leftmost (largest y)
    pool = sc["lane_pool"]
    si = sc["scene_in"]

    def fill(slot, lane_no):
        seg = pool[slot * n:(slot + 1) * n]
        seg["x"], seg["y"], seg["dir"] = x, lane_y.get(lane_no, 0.0), 0.0

    fill(0, lane_num), fill(1, lane_num - 1), fill(2, lane_num + 1)
    lv = si["lanes"]
    lv["cur_off"], lv["cur_n"] = 0, n
    lv["left_off"], lv["left_n"] = n, (n if lane_num > 1 else 0)
    lv["right_off"], lv["right_n"] = 2 * n, (n if lane_num < lane_sum else 0)
    lv["lane_sum"], lv["lanechg_attribute"], lv["lane_width"] = lane_sum, map_attr, LANE_W
    loc = si["loc"]
    loc["globalpoint"]["x"], loc["globalpoint"]["y"], loc["globalpoint"]["dir"] = x[EGO_ID], lane_y[lane_num], 0.0
    loc["velocity"], loc["pos"], loc["lane_num"], loc["road_num"] = 20.0, 0, lane_num, 1
    loc["last_lanenum"], loc["next_lanenum"] = lane_num, lane_num
    loc["id"][:] = EGO_ID
    si["out_lane_no"][:] = 0
    for k, v in enumerate(out_lanes):
        si["out_lane_no"][0, k] = v
    si["period_last"] = period
    a = sc["attr_pool"]
    a[:] = map_attr
    a[EGO_ID + 1:n] = 0
    a[EGO_ID + 1:min(n, EGO_ID + 1 + attr_run)] = map_attr if attr_ahead is None else attr_ahead
    si["obs_off"], si["obs_n"] = 0, n_obs
    for j, (lane_no, ahead) in enumerate(obstacles):
        o = sc["obs_pool"][j]
        o["x"], o["y"], o["type"], o["radius"] = x[EGO_ID] + ahead, lane_y[lane_no], 0, 0.5
    sc["mot_pool"][:] = 0
    si["goal"]["x"], si["goal"]["y"] = x[EGO_ID] + 20.0, lane_y[lane_num]
    si["grid_origin"]["x"], si["grid_origin"]["y"] = x[EGO_ID] - 5.0, lane_y[lane_num] - 16.0
    st = sc["state"]
    st["z_target_lanenum"], st["d_his_target_lanenum"] = lane_num, lane_num
    return sc


def switch_lane(dm, sc, new_lane, lane_sum=3):
    """The localisation reports the ego on `new_lane` (the lane change has happened)."""
    n = dm.GEN_LANE_PTS
    si = sc["scene_in"]
    pool = sc["lane_pool"]
    lane_y = {k: 200.0 - LANE_W * (k - 1) for k in range(1, lane_sum + 1)}
    for slot, lane_no in ((0, new_lane), (1, new_lane - 1), (2, new_lane + 1)):
        pool[slot * n:(slot + 1) * n]["y"] = lane_y.get(lane_no, 0.0)
    si["lanes"]["left_n"] = n if new_lane > 1 else 0
    si["lanes"]["right_n"] = n if new_lane < lane_sum else 0
    si["loc"]["lane_num"] = new_lane
    si["loc"]["globalpoint"]["y"] = lane_y[new_lane]


def behaviour(st, plan=None):
    """(behavior, target_lanenum, light, lanechg_status, behavior_to_dlg) of scene 0 after a tick."""
    return (int(st["z_behavior"][0]), int(st["z_target_lanenum"][0]), int(st["z_light_status"][0]),
            int(st["z_segment_lanechg_status"][0]), int(st["z_behavior_to_dlg"][0]))


# every scene of tests/test_lanechange_kat.py (plus a few variations), for the device-vs-oracle runs
SCENARIOS = [
    dict(lane_num=2, map_attr=1, out_lanes=(1,), period=700.0),
    dict(lane_num=2, map_attr=1, out_lanes=(3,), period=700.0),
    dict(lane_num=2, map_attr=1, out_lanes=(1,), period=900.0, obstacles=[(1, -12.0)]),
    dict(lane_num=2, map_attr=1, out_lanes=(1,), period=900.0, obstacles=[(1, 30.0), (2, 25.0)]),
    dict(lane_num=2, map_attr=1, out_lanes=(1,), period=900.0, obstacles=[(1, 30.0), (2, 15.0)]),
    dict(lane_num=2, map_attr=3, out_lanes=(1,), period=1100.0),
    dict(lane_num=2, map_attr=2, out_lanes=(3,), period=700.0),
    dict(lane_num=2, map_attr=2, out_lanes=(3,), period=2000.0),
    dict(lane_num=2, map_attr=3, out_lanes=(3,), period=2000.0),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=800.0, obstacles=[(2, 10.0)]),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)]),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=120),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=121),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_run=269),   # runs to the lane end
    dict(lane_num=2, map_attr=1, out_lanes=(1, 2), period=800.0, obstacles=[(2, 10.0)]),
    dict(lane_num=2, map_attr=1, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_run=30),
    dict(lane_num=2, map_attr=1, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_run=31),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3),
    dict(lane_num=2, map_attr=1, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3),
    dict(lane_num=1, map_attr=1, out_lanes=(1,), period=1600.0, obstacles=[(1, 10.0)]),
    dict(lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)]),
    dict(lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3),
    dict(lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3, attr_run=100),
    dict(lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3, attr_run=101),
    dict(lane_num=2, map_attr=2, out_lanes=(2,), period=1600.0, obstacles=[(2, 10.0), (3, -8.0)], attr_ahead=3),
    dict(lane_num=2, map_attr=2, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3, attr_run=21),
    dict(lane_num=2, map_attr=2, out_lanes=(1, 2), period=1600.0, obstacles=[(2, 10.0)], attr_ahead=3, attr_run=20),
    dict(lane_num=3, map_attr=2, out_lanes=(3,), period=1600.0, obstacles=[(3, 10.0)], attr_ahead=3),
    dict(lane_num=2, map_attr=3, out_lanes=(2,), period=2500.0, obstacles=[(2, 10.0)]),
    dict(lane_num=2, map_attr=3, out_lanes=(2, 3), period=900.0, obstacles=[(2, 10.0)]),
    dict(lane_num=2, map_attr=3, out_lanes=(1, 2), period=900.0, obstacles=[(2, 10.0)]),
    dict(lane_num=1, map_attr=3, out_lanes=(1,), period=900.0, obstacles=[(1, 10.0)]),
    dict(lane_num=1, map_attr=3, out_lanes=(1, 2), period=900.0, obstacles=[(1, 10.0)], attr_run=25),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=100.0, obstacles=[(2, 24.5)]),
    dict(lane_num=2, map_attr=1, out_lanes=(2,), period=100.0, obstacles=[(2, 25.0)]),
    dict(lane_num=2, map_attr=1, out_lanes=(), period=700.0),
]
