"""The committed regression vectors (tests/golden/oracle_r1.json, made by make_golden.py from this
repo's oracle — not reference outputs) against (a) the oracle on CPU and (b) the HIP path on GPU.
Each fixture carries the full PlannerConfig, since the reference fixes none of its macros."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_r1.json")))
INT_KEYS = ["status", "n_expanded", "n_pushed", "n_rounds", "path_len", "path_cost", "best_candidate", "afresh_cause",
            "path_near_id", "z_behavior", "ob_flag", "ob_pathid", "around_flag", "z_light_status", "z_target_lanenum",
            "lanechg_status", "behavior_to_dlg", "frontobs_time", "navi_lanechg"]
FLT_KEYS = ["ob_dis_lng", "desspd", "radius", "road_x_sum", "road_y_sum", "best_cost", "leftlight_time", "rightlight_time"]


def _cfg(dm, case):
    cfg = dm.default_config(case["grid"])
    for k, v in case["config"].items():
        cfg[k] = v
    return cfg


def _check(case, t, plan, gout, st):
    want = case["ticks"][t]
    got = {"status": gout["status"], "n_expanded": gout["n_expanded"], "n_pushed": gout["n_pushed"], "n_rounds": gout["n_rounds"],
           "path_len": gout["path_len"], "path_cost": gout["path_cost"], "best_candidate": gout["best_candidate"],
           "afresh_cause": st["afresh_cause"], "path_near_id": st["path_near_id"], "z_behavior": st["z_behavior"],
           "ob_flag": plan["ob_flag"], "ob_pathid": plan["ob_pathid"], "around_flag": plan["around"]["Obs_flag"],
           "ob_dis_lng": plan["ob_dis_lng"], "desspd": plan["result"]["desspd"], "radius": plan["result"]["radius"],
           "road_x_sum": plan["road_points"]["x"].sum(axis=1), "road_y_sum": plan["road_points"]["y"].sum(axis=1),
           "best_cost": np.array([float(g["cand_cost"][int(g["best_candidate"])]) for g in gout]),
           "z_light_status": st["z_light_status"], "z_target_lanenum": st["z_target_lanenum"],
           "lanechg_status": st["z_segment_lanechg_status"], "behavior_to_dlg": st["z_behavior_to_dlg"],
           "frontobs_time": st["frontobs_time"], "navi_lanechg": plan["navi_lanechg"],
           "leftlight_time": st["leftlight_time"], "rightlight_time": st["rightlight_time"]}
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (k, t)
    assert [str(int(d)) for d in gout["order_digest"]] == want["order_digest"], t
    for k in FLT_KEYS:
        assert np.allclose(np.asarray(got[k], float), np.asarray(want[k], float), rtol=1e-6, atol=1e-9, equal_nan=True), (k, t)


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_reproduces_golden(dm, oracle, name):
    case = GOLD[name]
    cfg = _cfg(dm, case)
    sc = dm.gen_scenes(cfg, case["first_scene"], case["scenes"], case["obstacles"], case["junction_every"])
    sc["scene_in"]["period_last"] = case["period_last"]
    st = sc["state"].copy()
    for t in range(len(case["ticks"])):
        plan, gout, _ = oracle.plan_tick_batch(cfg, sc, st)
        _check(case, t, plan, gout, st)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_hip_reproduces_golden(dm, name):
    case = GOLD[name]
    cfg = _cfg(dm, case)
    sc = dm.gen_scenes(cfg, case["first_scene"], case["scenes"], case["obstacles"], case["junction_every"])
    sc["scene_in"]["period_last"] = case["period_last"]
    pl = dm.Planner(cfg, max_scenes=case["scenes"], max_obs_total=max(case["scenes"] * case["obstacles"], 1))
    pl.set_scenes(sc)
    pl.set_state(sc["state"])
    for t in range(len(case["ticks"])):
        pl.tick(sync=True)
        _check(case, t, pl.get_plan(), pl.get_grid_out(), pl.get_state())
