#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_r1.json from the CPU oracle.

These are REGRESSION vectors produced by this repo's own oracle on seeded synthetic scenes, NOT
outputs of the reference: the reference ships no tests or fixtures and cannot be built in this
image, so nothing here pins the reference's arithmetic ("parity unpinned").  They
freeze the specification so that a later change to the oracle or the kernels is noticed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dmpp_amd as dm          # noqa: E402
import oracle_binding          # noqa: E402

CASES = [  # name, grid, first scene, scenes, obstacles, junction_every, ticks, config overrides, decision period (ms)
    ("config0_128_8obs", 128, 0, 1, 8, 0, 3, {}, 100.0),
    ("batch_512_64obs", 512, 1000, 12, 64, 4, 2, {}, 100.0),
    ("dynamic_512_256obs", 512, 9000, 4, 256, 0, 5, {"dynamic_obstacles": 1, "force_replan": 1}, 100.0),
    ("lanechange_128_64obs", 128, 1320, 48, 64, 8, 10, {}, 900.0),   # window with left and right lane changes
]


def run_case(orc, name, grid, first, n, n_obs, je, ticks, over, period):
    cfg = dm.default_config(grid)
    for k, v in over.items():
        cfg[k] = v
    sc = dm.gen_scenes(cfg, first, n, n_obs, je)
    sc["scene_in"]["period_last"] = period
    st = sc["state"].copy()
    out = []
    for _ in range(ticks):
        plan, gout, _ = orc.plan_tick_batch(cfg, sc, st)
        out.append({
            "status": gout["status"].tolist(), "n_expanded": gout["n_expanded"].tolist(), "n_pushed": gout["n_pushed"].tolist(),
            "n_rounds": gout["n_rounds"].tolist(), "path_len": gout["path_len"].tolist(), "path_cost": gout["path_cost"].tolist(),
            "order_digest": [str(int(d)) for d in gout["order_digest"]], "best_candidate": gout["best_candidate"].tolist(),
            "best_cost": [float(g["cand_cost"][int(g["best_candidate"])]) for g in gout],
            "afresh_cause": st["afresh_cause"].tolist(), "path_near_id": st["path_near_id"].tolist(),
            "z_behavior": st["z_behavior"].tolist(), "z_light_status": st["z_light_status"].tolist(),
            "z_target_lanenum": st["z_target_lanenum"].tolist(), "lanechg_status": st["z_segment_lanechg_status"].tolist(),
            "behavior_to_dlg": st["z_behavior_to_dlg"].tolist(), "frontobs_time": st["frontobs_time"].tolist(),
            "navi_lanechg": plan["navi_lanechg"].tolist(), "leftlight_time": st["leftlight_time"].tolist(),
            "rightlight_time": st["rightlight_time"].tolist(), "ob_flag": plan["ob_flag"].tolist(), "ob_pathid": plan["ob_pathid"].tolist(),
            "ob_dis_lng": plan["ob_dis_lng"].tolist(), "desspd": plan["result"]["desspd"].tolist(),
            "radius": plan["result"]["radius"].tolist(), "around_flag": plan["around"]["Obs_flag"].tolist(),
            "road_x_sum": plan["road_points"]["x"].sum(axis=1).tolist(), "road_y_sum": plan["road_points"]["y"].sum(axis=1).tolist(),
        })
    return {"grid": grid, "first_scene": first, "scenes": n, "obstacles": n_obs, "junction_every": je, "overrides": over, "period_last": period,
            "config": {k: (float(cfg[k][0]) if cfg.dtype[k].kind == "f" else int(cfg[k][0])) for k in cfg.dtype.names}, "ticks": out}


if __name__ == "__main__":
    orc = oracle_binding.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    res = {c[0]: run_case(orc, *c) for c in CASES}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_r1.json"), "w") as f:
        json.dump(res, f, indent=0, separators=(",", ":"))
    print("wrote", len(res), "cases")
