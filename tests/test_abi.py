"""CPU: the C-ABI library loads, exports every symbol the header declares, mirrors the struct sizes,
and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dmpp_planner.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(dm):
    lib = dm.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 30, syms
    for s in syms:
        assert hasattr(lib, s), f"libdmpp.so does not export {s}"


def test_struct_sizes_match_header(dm):
    lib = dm.load_library()
    want = {2: dm.SceneIn, 3: dm.SceneState, 4: dm.PlanOut, 5: dm.GridOut, 6: dm.ObPoint, 12: dm.PlanningOut,
            13: dm.PlanningStatus}
    for which, dt in want.items():
        assert lib.pp_sizeof(which) == dt.itemsize
    # the sizes SURVEY.md §8 prices the algorithmic bytes with
    assert dm.GlobalPoint2D.itemsize == 16 and dm.ObPoint.itemsize == 24
    assert dm.SceneState.fields["last_Bpoints"][0].itemsize == 3200
    assert dm.PlanningOut.itemsize == 1680


def test_default_config_records_every_undefined_macro(dm):
    cfg = dm.default_config(512)
    assert cfg["ROAD_FARAIM_MAX"][0] == 40 and cfg["ROAD_FARAIM_MIN"][0] == 10
    assert cfg["PRE_INTER_FARAIM"][0] == 15 and cfg["INTER_FARAIM"][0] == 10
    assert cfg["ROAD_REMAIN_DISTANCE"][0] == 10 and cfg["INTER_REMAIN_DISTANCE"][0] == 5
    assert cfg["EPSILON"][0] == 1e-6 and cfg["Vehicle_Width"][0] == 1.8 and cfg["ID_MORE"][0] == 0
    assert cfg["NO_OBSTACLE_DIS"][0] == 999          # the value Planning.cpp:161-162 pre-loads
    assert cfg["grid_w"][0] == 512 and cfg["cell"][0] == 0.25


def test_scene_generator_is_seeded_and_sharded_consistently(dm):
    cfg = dm.default_config(512)
    a = dm.gen_scenes(cfg, 0, 16, 64)
    b = dm.gen_scenes(cfg, 8, 8, 64)
    # scene k of a batch starting at 8 == scene 8+k of a batch starting at 0 (offsets are shard-local)
    for f in ("loc", "dec", "goal", "grid_origin", "stub_attribute"):
        assert a["scene_in"][f][8:].tobytes() == b["scene_in"][f].tobytes()
    assert a["obs_pool"][8 * 64:].tobytes() == b["obs_pool"].tobytes()
    assert a["lane_pool"][8 * 3 * dm.GEN_LANE_PTS:].tobytes() == b["lane_pool"].tobytes()
    assert (b["scene_in"]["obs_off"] == np.arange(8) * 64).all()
    ego = a["scene_in"]["loc"]["globalpoint"]
    d2 = (a["obs_pool"]["x"].reshape(16, 64) - ego["x"][:, None]) ** 2 + (a["obs_pool"]["y"].reshape(16, 64) - ego["y"][:, None]) ** 2
    assert (d2 >= 9.0).all()                         # obstacles at least 3 m from the ego
    assert set(np.unique(a["scene_in"]["loc"]["pos"])) == {0, 1, 2}


def test_no_cpu_fallback(dm):
    """Without a GPU pp_create must fail loudly; with one it must succeed."""
    import torch
    cfg = dm.default_config(128)
    if torch.cuda.is_available():
        dm.Planner(cfg, max_scenes=1).close()
    else:
        with pytest.raises(dm.PlannerError, match="HIP|device"):
            dm.Planner(cfg, max_scenes=1)


def test_argument_checks_without_gpu(dm):
    lib = dm.load_library()
    h = ctypes.c_void_p()
    caps = np.zeros(1, dm.PlannerCaps)
    assert lib.pp_create(None, 0, caps.ctypes.data, ctypes.byref(h)) != 0
    assert b"null" in lib.pp_last_error() or b"config" in lib.pp_last_error()
    assert lib.pp_plan_tick(None) != 0 and lib.pp_sync(None) != 0
    assert lib.pp_destroy(None) == 0
