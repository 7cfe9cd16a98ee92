"""A small road network for the map-store tests: `n_roads` roads of three parallel lanes on
gentle sinusoids, a junction polyline from every lane of road r to the same lane of road r+1, and egos spread
over roads / lanes / point ids / positions (road, pre-junction, junction).  Everything the device derives from
the map is also derived here in plain numpy (`resolve`), the way the reference indexes its vectors
(Planning.cpp:331-380, Decision.cpp:346-348,562-578)."""
import numpy as np

LANE_W_CM = 375
PTS = 260          # points per lane
JPTS = 40          # points per junction polyline


def build_map(dm, n_roads=4, seed=5):
    rng = np.random.default_rng(seed)
    lanes, first = [], [0]
    pts, attr, width = [], [], []
    for r in range(n_roads):
        n_l = 3 if r % 3 != 2 else 2                      # one road with two lanes only
        x0, y0 = 20.0 + 8.0 * r, 14.0 + 20.0 * r
        A, lam = rng.uniform(0.0, 2.0), rng.uniform(50.0, 90.0)
        x = x0 + 0.5 * np.arange(PTS)
        for l in range(n_l):
            p = np.zeros(PTS, dm.GlobalPoint3D)
            p["x"] = x
            p["y"] = y0 + A * np.sin(2 * np.pi * (x - x0) / lam) - 3.75 * l      # lane 1 leftmost
            p["dir"] = np.degrees(np.arctan(A * 2 * np.pi / lam * np.cos(2 * np.pi * (x - x0) / lam))) % 360.0
            a = np.full(PTS, [0, 1, 2, 3][(r + l) % 4], np.uint8)
            a[rng.integers(60, 200):] = 0                                          # the attribute ends somewhere
            w = np.full(PTS, LANE_W_CM - 5 * l, np.uint16)
            lanes.append((sum(len(q) for q in pts), PTS, n_l))
            pts.append(p); attr.append(a); width.append(w)
        first.append(len(lanes))
    junc, jpts = [], []
    for r in range(n_roads - 1):
        for l in range(2):
            src = pts[first[r] + l][-1]
            dst = pts[first[r + 1] + l][0]
            t = np.linspace(0.0, 1.0, JPTS)
            jp = np.zeros(JPTS, dm.GlobalPoint2D)
            jp["x"] = src["x"] + (dst["x"] - src["x"]) * t
            jp["y"] = src["y"] + (dst["y"] - src["y"]) * t * t            # a bend
            junc.append((r + 1, r + 2, l + 1, l + 1, sum(len(q) for q in jpts), JPTS))
            jpts.append(jp)
    m = dict(road_first_lane=np.array(first, np.int32),
             lanes=np.array([(o, n, ls, 0) for o, n, ls in lanes], dm.MapLane),
             points=np.concatenate(pts), lanechg_attribute=np.concatenate(attr), lane_width_cm=np.concatenate(width),
             junctions=np.array(junc, dm.MapJunction), jpoints=np.concatenate(jpts))
    return m


def make_egos(dm, cfg, m, n, n_obs, seed=11):
    """SceneIn records with lanes / ref slices left zero (the device derives them), obstacles near the egos."""
    rng = np.random.default_rng(seed)
    sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=0)             # obstacle radii / motion, goal etc. as generated
    si = sc["scene_in"]
    n_roads = len(m["road_first_lane"]) - 1
    S = float(cfg["grid_w"][0]) * float(cfg["cell"][0])
    for s in range(n):
        road = int(rng.integers(1, n_roads + 1))
        n_l = int(m["road_first_lane"][road] - m["road_first_lane"][road - 1])
        lane = int(rng.integers(1, n_l + 1))
        pos = int(rng.choice([0, 0, 0, 1, 2]))
        if pos == 1 and (road == n_roads or lane > 2):
            pos = 0
        if pos == 2 and (road == 1 or lane > 2):
            pos = 0
        L = m["lanes"][m["road_first_lane"][road - 1] + lane - 1]
        loc = si["loc"][s]
        loc["pos"], loc["road_num"], loc["lane_num"] = pos, road, lane
        if pos == 2:                                       # in the junction leading INTO this road
            loc["last_roadnum"], loc["next_roadnum"], loc["last_lanenum"], loc["next_lanenum"] = road - 1, road, lane, lane
            jid = int(rng.integers(2, 20))
            J = [q for q in m["junctions"] if q["last_road"] == road - 1 and q["last_lane"] == lane][0]
            p = m["jpoints"][int(J["point_off"]) + jid]
            loc["id"][:] = jid
            ex, ey, ed = float(p["x"]), float(p["y"]), 10.0
        else:
            loc["last_roadnum"], loc["next_roadnum"], loc["last_lanenum"], loc["next_lanenum"] = road, road + 1, lane, lane
            pid = int(rng.integers(20, 120)) if pos == 0 else int(rng.integers(200, 250))
            p = m["points"][int(L["point_off"]) + pid]
            loc["id"][:] = pid
            ex, ey, ed = float(p["x"]), float(p["y"]) + float(rng.uniform(-0.1, 0.1)), float(p["dir"])
        loc["globalpoint"]["x"], loc["globalpoint"]["y"], loc["globalpoint"]["dir"] = ex, ey, ed
        loc["velocity"] = float(rng.uniform(5.0, 50.0))
        si["lanes"][s] = np.zeros(1, dm.LaneView)[0]
        si["ref_off"][s], si["ref_n"][s] = 0, 0
        si["out_lane_no"][s] = 0
        si["out_lane_no"][s, 0] = int(rng.integers(1, n_l + 1))
        si["period_last"][s] = float(rng.choice([100.0, 900.0, 1600.0]))
        si["stub_attribute"][s] = int(rng.integers(0, 4))
        si["grid_origin"][s]["x"], si["grid_origin"][s]["y"] = ex - 0.1 * S, ey - 0.5 * S
        si["goal"][s]["x"], si["goal"][s]["y"] = ex + 0.8 * S, ey + float(rng.uniform(-0.3, 0.3)) * S
        o = sc["obs_pool"][s * n_obs:(s + 1) * n_obs]
        o["x"] = ex + rng.uniform(4.0, 0.85 * S, n_obs)
        o["y"] = ey + rng.uniform(-0.45 * S, 0.45 * S, n_obs)
        near = min(3, n_obs)
        o["x"][:near] = ex + rng.uniform(6.0, 30.0, near)
        o["y"][:near] = ey + rng.uniform(-4.0, 4.0, near)
        st = sc["state"][s]
        st["z_target_lanenum"], st["d_his_target_lanenum"] = lane, lane
    sc["lane_pool"], sc["attr_pool"], sc["ref_pool"] = m["points"], m["lanechg_attribute"], m["jpoints"]
    return sc


def resolve(dm, m, scene_in):
    """The lane views and junction slices as the reference's vector indexing gives them."""
    out = scene_in.copy()
    for s in range(len(out)):
        loc = out["loc"][s]
        road, lane = int(loc["road_num"]), int(loc["lane_num"])
        L0 = int(m["road_first_lane"][road - 1])
        L1 = int(m["road_first_lane"][road])
        cur = m["lanes"][L0 + lane - 1]
        lv = np.zeros(1, dm.LaneView)[0]
        lv["cur_off"], lv["cur_n"], lv["lane_sum"] = cur["point_off"], cur["n_points"], cur["lane_sum"]
        if lane > 1:
            lv["left_off"], lv["left_n"] = m["lanes"][L0 + lane - 2]["point_off"], m["lanes"][L0 + lane - 2]["n_points"]
        if lane < int(cur["lane_sum"]) and L0 + lane < L1:
            lv["right_off"], lv["right_n"] = m["lanes"][L0 + lane]["point_off"], m["lanes"][L0 + lane]["n_points"]
        pid = min(max(int(loc["id"][lane - 1]), 0), int(cur["n_points"]) - 1)
        lv["lanechg_attribute"] = m["lanechg_attribute"][int(cur["point_off"]) + pid]
        lv["lane_width"] = float(m["lane_width_cm"][int(cur["point_off"]) + pid]) / 100.0
        out["lanes"][s] = lv
        out["ref_off"][s], out["ref_n"][s] = 0, 0
        for q in m["junctions"]:
            if (int(q["last_road"]), int(q["next_road"]), int(q["last_lane"]), int(q["next_lane"])) == \
                    (int(loc["last_roadnum"]), int(loc["next_roadnum"]), int(loc["last_lanenum"]), int(loc["next_lanenum"])):
                out["ref_off"][s], out["ref_n"][s] = q["point_off"], q["n_points"]
                break
    return out
