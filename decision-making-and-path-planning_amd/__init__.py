"""Python host binding of the MI355X planning engine (ctypes over include/dmpp_planner.h).

The product is libdmpp.so (hand-written HIP kernels behind a C-ABI).  This module only
mirrors the ABI structs as numpy dtypes and forwards calls; it never computes planning
results itself and it raises if the library is missing (no CPU fallback).

Reference surface mirrored: CPlanning / CDecision (Planning.h:38-85, Decision.h:106-107)
as one batched `Planner.tick()`; see INTEGRATION.md.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdmpp.so")

PATH_POINTS, OUT_POINTS, LANESUM, MAX_REFPATH, MAX_LATTICE = 200, 100, 8, 512, 17
GEN_LANE_PTS, GEN_REF_PTS = 320, 128

G_FOUND, G_NO_PATH, G_LIMIT, G_OVERFLOW, G_GOAL_BLOCKED, G_PATH_TRUNC, G_INTERNAL, G_COST_RANGE = range(8)
G_STATUS_COUNT = 8
K_NAMES = ["k_effective_obstacles", "k_decision", "k_planning", "k_search", "k_score"]
(BUF_SCENE_IN, BUF_LANE_POOL, BUF_REF_POOL, BUF_OBS_POOL, BUF_MOT_POOL, BUF_STATE, BUF_PLAN_OUT, BUF_GRID_OUT,
 BUF_GRID, BUF_PATH, BUF_ORDER, BUF_LANE_ATTR) = range(12)


def _dt(fields):
    return np.dtype(fields, align=True)


f8, i4, u4, f4, u8, u2 = np.float64, np.int32, np.uint32, np.float32, np.uint64, np.uint16
GlobalPoint2D = _dt([("x", f8), ("y", f8)])
GlobalPoint3D = _dt([("x", f8), ("y", f8), ("dir", f8)])
AimPoint = _dt([("Aim_point", GlobalPoint3D), ("Aim_id", i4), ("_pad", i4)])
ObPoint = _dt([("x", f8), ("y", f8), ("type", i4), ("radius", f4)])
ObMotion = _dt([("vx", f8), ("vy", f8)])
LaneView = _dt([("cur_off", i4), ("cur_n", i4), ("left_off", i4), ("left_n", i4), ("right_off", i4), ("right_n", i4),
                ("lane_sum", i4), ("lanechg_attribute", i4), ("lane_width", f8)])
LocationOut = _dt([("globalpoint", GlobalPoint3D), ("velocity", f8), ("pos", i4), ("road_num", i4), ("lane_num", i4),
                   ("last_roadnum", i4), ("next_roadnum", i4), ("last_lanenum", i4), ("next_lanenum", i4),
                   ("path_num", i4), ("id", i4, (LANESUM,))])
DecisionOutPod = _dt([("velocity_expect", f8), ("behavior", i4), ("target_roadnum", i4), ("target_lanenum", i4),
                   ("light", i4), ("behavior_to_dlg", i4), ("refpath_n", i4)])
Obs_To_Veh = _dt([("dis_lat", f8), ("dis_lng", f8)])
Path_Obs = _dt([("Ob_Pose", Obs_To_Veh), ("Ob_Attr", ObPoint), ("Obs_flag", i4), ("Ob_Pathid", i4)])
PlanningOut = _dt([("brakedis", f8), ("brake_speed", f8), ("desacc", f8), ("desspd", f8), ("desstr", f8), ("radius", f8),
                   ("cnt", i4), ("APA", i4), ("desaccVd", i4), ("desstrVd", i4), ("light", i4), ("road_type", i4),
                   ("sstop", i4), ("_pad", i4), ("pnts", GlobalPoint2D, (OUT_POINTS,))])
PlanningStatus = _dt([("near_ob_dist", f8), ("planspeed", f8), ("planacc", f8), ("afresh_cause", i4), ("trafficlight", i4),
                      ("path_points", GlobalPoint2D, (OUT_POINTS,))])
SceneState = _dt([("last_Bpoints", GlobalPoint2D, (PATH_POINTS,)), ("aimpoint_far", AimPoint), ("aimpoint_near", AimPoint),
                  ("path_lat_dis", f8), ("remain_dis", f8), ("path_dir_err", f8), ("brakespeed", f8), ("des_acc", f8),
                  ("faraim_dis", f4), ("nearaim_dis", f4), ("path_near_id", i4), ("path_front_near_id", i4),
                  ("his_behavior", i4), ("afresh_planning", i4), ("afresh_cause", i4), ("acc_flag", i4), ("count", i4),
                  ("z_behavior", i4), ("z_light_status", i4), ("z_target_lanenum", i4), ("z_target_roadnum", i4),
                  ("z_behavior_to_dlg", i4), ("z_segment_lanechg_status", i4), ("z_segment_obsavoid_status", i4),
                  ("d_his_behavior", i4), ("d_his_light_status", i4), ("d_his_target_lanenum", i4),
                  ("obsavoid_time", u4), ("no_obsaviod_time", u4), ("frontobs_time", u4), ("tick", i4), ("_pad", i4),
                  ("z_velocity_expect", f8), ("leftlight_time", f8), ("rightlight_time", f8)])
SceneIn = _dt([("loc", LocationOut), ("dec", DecisionOutPod), ("lanes", LaneView), ("ref_off", i4), ("ref_n", i4),
               ("obs_off", i4), ("obs_n", i4), ("stub_attribute", i4), ("_pad", i4), ("out_lane_no", u2, (LANESUM,)),
               ("period_last", f8), ("grid_origin", GlobalPoint2D),
               ("goal", GlobalPoint2D)])
GridOut = _dt([("order_digest", u8), ("status", i4), ("n_expanded", i4), ("n_pushed", i4), ("n_rounds", i4),
               ("path_len", i4), ("path_cost", i4), ("start_cell", i4), ("goal_cell", i4), ("best_candidate", i4),
               ("n_candidates", i4), ("cand_cost", f8, (MAX_LATTICE,)), ("cand_col", f8, (MAX_LATTICE,)),
               ("cand_curv", f8, (MAX_LATTICE,)), ("cand_prog", f8, (MAX_LATTICE,)),
               ("best_path", GlobalPoint2D, (PATH_POINTS,))])
PlanOut = _dt([("result", PlanningOut), ("show", PlanningStatus), ("road_points", GlobalPoint2D, (PATH_POINTS,)),
               ("around", Path_Obs, (6,)), ("dec", DecisionOutPod), ("ob_dis_lat", f8), ("ob_dis_lng", f8), ("ob", ObPoint),
               ("ob_flag", i4), ("ob_pathid", i4), ("sweep_side", i4), ("sweep_index", i4), ("navi_lanechg", i4),
               ("navi_lanechg_times", i4)])
PlannerConfig = _dt([("ROAD_FARAIM_MAX", f8), ("ROAD_FARAIM_MIN", f8), ("PRE_INTER_FARAIM", f8), ("INTER_FARAIM", f8),
                     ("ROAD_REMAIN_DISTANCE", f8), ("INTER_REMAIN_DISTANCE", f8), ("EPSILON", f8), ("PI", f8),
                     ("Vehicle_Width", f8), ("NO_OBSTACLE_DIS", f8), ("wgs_lat0", f8), ("wgs_lng0", f8),
                     ("wgs_deg_per_m_lat", f8), ("wgs_deg_per_m_lng", f8), ("ID_MORE", i4), ("decision_stage", i4),
                     ("lanechg_stage", i4), ("grid_stage", i4), ("grid_w", i4), ("grid_h", i4), ("max_expansions", i4), ("bucket_cap", i4),
                     ("max_path", i4), ("n_lattice", i4), ("lookahead_cells", i4), ("dynamic_obstacles", i4),
                     ("force_replan", i4), ("_cfg_pad", i4), ("cell", f8), ("inflate", f8), ("lattice_step", f8), ("d_safe", f8),
                     ("w_col", f8), ("w_curv", f8), ("w_prog", f8), ("w_off", f8), ("dyn_dt", f8)])
PlannerCaps = _dt([("max_scenes", i4), ("max_obs_total", i4), ("max_lane_pts_total", i4), ("max_ref_pts_total", i4),
                   ("order_cap", i4), ("_pad", i4)])

MapLane = _dt([("point_off", i4), ("n_points", i4), ("lane_sum", i4), ("_pad", i4)])
MapJunction = _dt([("last_road", i4), ("next_road", i4), ("last_lane", i4), ("next_lane", i4), ("point_off", i4), ("n_points", i4)])

_SIZEOF_ORDER = [PlannerConfig, PlannerCaps, SceneIn, SceneState, PlanOut, GridOut, ObPoint, ObMotion, Path_Obs,
                 LocationOut, DecisionOutPod, LaneView, PlanningOut, PlanningStatus, AimPoint, MapLane, MapJunction]


class MapDesc(C.Structure):          # include/dmpp_types.h: the map store handed to pp_set_map
    _fields_ = [("n_roads", C.c_int32), ("n_lanes", C.c_int32), ("n_points", C.c_int32), ("n_junctions", C.c_int32),
                ("n_jpoints", C.c_int32), ("_pad", C.c_int32), ("road_first_lane", C.c_void_p), ("lanes", C.c_void_p),
                ("points", C.c_void_p), ("lanechg_attribute", C.c_void_p), ("lane_width_cm", C.c_void_p),
                ("junctions", C.c_void_p), ("jpoints", C.c_void_p)]



class _P3(C.Structure):          # GlobalPoint3D passed by value
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("dir", C.c_double)]


_lib = None


class PlannerError(RuntimeError):
    pass


def load_library(path=None):
    """Load libdmpp.so (built by `make -C csrc` / __graft_entry__.build()). Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("DMPP_LIB") or LIB_PATH          # DMPP_LIB: another build of the library (A/B measurements)
    if not os.path.exists(path):
        raise PlannerError(f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path)
    vp, ci, cz = C.c_void_p, C.c_int, C.c_size_t
    lib.pp_last_error.restype = C.c_char_p
    lib.pp_sizeof.restype = cz
    lib.pp_sizeof.argtypes = [ci]
    lib.pp_default_config.argtypes = [vp, ci, ci]
    lib.pp_default_config.restype = None
    lib.pp_init_state.argtypes = [vp, ci]
    lib.pp_init_state.restype = None
    lib.pp_gen_scenes.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.pp_create.argtypes = [vp, ci, vp, C.POINTER(vp)]
    lib.pp_destroy.argtypes = [vp]
    lib.pp_set_config.argtypes = [vp, vp]
    lib.pp_set_scenes.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, vp, vp, ci]
    lib.pp_set_n_scenes.argtypes = [vp, ci, ci, ci, ci, ci, ci]
    lib.pp_join.argtypes = [vp]
    lib.pp_get_search_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    lib.pp_set_map.argtypes = [vp, vp]
    lib.pp_set_egos.argtypes = [vp, ci, vp, vp, vp, ci]
    lib.pp_get_scene_in.argtypes = [vp, vp, ci]
    lib.pp_set_state.argtypes = [vp, vp, ci]
    lib.pp_plan_tick.argtypes = [vp]
    lib.pp_sync.argtypes = [vp]
    lib.pp_device_synchronize.argtypes = [vp]
    lib.pp_get_plan.argtypes = [vp, vp, ci]
    lib.pp_get_state.argtypes = [vp, vp, ci]
    lib.pp_get_grid_out.argtypes = [vp, vp, ci]
    lib.pp_get_grid.argtypes = [vp, ci, vp]
    lib.pp_get_order.argtypes = [vp, ci, vp, ci]
    lib.pp_get_path.argtypes = [vp, ci, vp, ci]
    lib.pp_get_refpath.argtypes = [vp, ci, vp, ci]
    lib.pp_plan_tick_batch.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, ci, vp, ci, vp, vp, vp]
    lib.pp_search_obstacle_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.pp_geom_batch.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    lib.pp_scalar_stage.argtypes = [vp, ci, vp, ci, vp, vp, ci]
    lib.pp_bezier.argtypes = [vp, _P3, _P3, vp, ci]
    lib.pp_mean_points.argtypes = [vp, vp, ci, vp, ci]
    lib.pp_create_new_path.argtypes = [vp, vp, ci, C.c_double, vp]
    lib.pp_set_profile.argtypes = [vp, ci]
    lib.pp_get_kernel_ms.argtypes = [vp, ci, C.POINTER(C.c_float), C.POINTER(ci)]
    lib.pp_reset_kernel_ms.argtypes = [vp]
    lib.pp_device_ptr.argtypes = [vp, ci, C.POINTER(cz)]
    lib.pp_device_ptr.restype = vp
    lib.pp_stream.argtypes = [vp]
    lib.pp_stream.restype = vp
    lib.pp_update_async.argtypes = [vp, ci, vp, vp, vp, ci]
    lib.pp_fetch_async.argtypes = [vp, vp, vp, C.POINTER(C.c_longlong)]
    lib.pp_fetch_published_async.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_longlong)]
    lib.pp_wait_tick.argtypes = [vp, C.c_longlong, C.POINTER(ci)]
    lib.pp_tick_id.argtypes = [vp]
    lib.pp_tick_id.restype = C.c_longlong
    lib.pp_host_alloc.argtypes = [cz]
    lib.pp_host_alloc.restype = vp
    lib.pp_host_free.argtypes = [vp]
    lib.pp_host_free.restype = None
    lib.pp_host_register.argtypes = [vp, cz]
    lib.pp_host_unregister.argtypes = [vp]
    if lib.pp_sizeof(17) != C.sizeof(MapDesc):
        raise PlannerError(f"ABI mismatch for MapDesc: C {lib.pp_sizeof(17)} B, binding {C.sizeof(MapDesc)} B")
    for which, dt in enumerate(_SIZEOF_ORDER):
        if lib.pp_sizeof(which) != dt.itemsize:
            raise PlannerError(f"ABI mismatch for struct #{which}: C {lib.pp_sizeof(which)} B, binding {dt.itemsize} B")
    _lib = lib
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):          # raw host/device address
        return a
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def _check(rc):
    if rc != 0:
        raise PlannerError(f"libdmpp error {rc}: {load_library().pp_last_error().decode()}")


def default_config(grid_w=512, grid_h=None):
    """PlannerConfig record with the build-chosen values for every undefined reference macro."""
    cfg = np.zeros(1, PlannerConfig)
    load_library().pp_default_config(_ptr(cfg), grid_w, grid_h or grid_w)
    return cfg


def gen_scenes(cfg, first_scene, n_scenes, n_obs, junction_every=8):
    """Seeded synthetic scenes. Returns a dict of numpy arrays."""
    lib = load_library()
    sc = dict(
        scene_in=np.zeros(n_scenes, SceneIn),
        lane_pool=np.zeros(n_scenes * 3 * GEN_LANE_PTS, GlobalPoint3D),
        attr_pool=np.zeros(n_scenes * 3 * GEN_LANE_PTS, np.uint8),
        ref_pool=np.zeros(n_scenes * GEN_REF_PTS, GlobalPoint2D),
        obs_pool=np.zeros(max(n_scenes * n_obs, 1), ObPoint),
        mot_pool=np.zeros(max(n_scenes * n_obs, 1), ObMotion),
        state=np.zeros(n_scenes, SceneState),
    )
    _check(lib.pp_gen_scenes(_ptr(cfg), first_scene, n_scenes, n_obs, junction_every, _ptr(sc["scene_in"]),
                             _ptr(sc["lane_pool"]), _ptr(sc["attr_pool"]), _ptr(sc["ref_pool"]), _ptr(sc["obs_pool"]),
                             _ptr(sc["mot_pool"]), _ptr(sc["state"])))
    sc["n_obs"] = n_obs
    return sc


class _Pinned:
    """Owner of one pp_host_alloc block (freed when the last numpy view of it goes)."""

    def __init__(self, nbytes):
        self.lib = load_library()
        self.ptr = self.lib.pp_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise PlannerError("pp_host_alloc failed: " + self.lib.pp_last_error().decode())
        self.buf = (C.c_char * max(int(nbytes), 1)).from_address(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            self.lib.pp_host_free(self.ptr)
            self.ptr = None


def pinned_empty(n, dtype):
    """numpy array of n records in pinned host memory (pp_host_alloc): what pp_update_async / pp_fetch_async want."""
    dtype = np.dtype(dtype)
    owner = _Pinned(n * dtype.itemsize)
    a = np.frombuffer(owner.buf, dtype=dtype, count=n)      # keeps `owner.buf` (and through it nothing else) alive ...
    a = a.view(_PinnedArray)
    a._owner = owner                                          # ... so the owner rides on the array
    return a


class _PinnedArray(np.ndarray):
    _owner = None

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def pinned_copy(a):
    out = pinned_empty(len(a), a.dtype)
    out[...] = a
    return out


class Planner:
    """One GPU, one stream, resident scenes: the batched stand-in for the CDecision + CPlanning threads."""

    def __init__(self, cfg, device=0, max_scenes=1024, max_obs_total=None, max_lane_pts_total=None,
                 max_ref_pts_total=None, order_cap=0):
        self.lib = load_library()
        self.cfg = np.array(cfg, PlannerConfig).reshape(1).copy()
        caps = np.zeros(1, PlannerCaps)
        caps["max_scenes"] = max_scenes
        caps["max_obs_total"] = max_obs_total if max_obs_total is not None else max_scenes * 256
        caps["max_lane_pts_total"] = max_lane_pts_total if max_lane_pts_total is not None else max_scenes * 3 * GEN_LANE_PTS
        caps["max_ref_pts_total"] = max_ref_pts_total if max_ref_pts_total is not None else max_scenes * GEN_REF_PTS
        caps["order_cap"] = order_cap
        self.caps = caps
        h = C.c_void_p()
        _check(self.lib.pp_create(_ptr(self.cfg), device, _ptr(caps), C.byref(h)))
        self.h = h
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.pp_destroy(self.h)
            self.h = None

    __del__ = close

    def set_config(self, cfg):
        self.cfg = np.array(cfg, PlannerConfig).reshape(1).copy()
        _check(self.lib.pp_set_config(self.h, _ptr(self.cfg)))

    def set_scenes(self, sc, with_motion=True):
        n = len(sc["scene_in"])
        _check(self.lib.pp_set_scenes(self.h, n, _ptr(sc["scene_in"]), _ptr(sc["lane_pool"]), _ptr(sc.get("attr_pool")),
                                      len(sc["lane_pool"]),
                                      _ptr(sc["ref_pool"]), len(sc["ref_pool"]), _ptr(sc["obs_pool"]),
                                      _ptr(sc["mot_pool"]) if with_motion else None, n * sc["n_obs"]))
        self.n = n

    def set_map(self, m):
        """Resident map store (pp_set_map). `m`: dict with road_first_lane (int32, n_roads + 1), lanes (MapLane),
        points (GlobalPoint3D), lanechg_attribute (uint8), lane_width_cm (uint16), junctions (MapJunction),
        jpoints (GlobalPoint2D)."""
        keep = {k: np.ascontiguousarray(m[k]) for k in ("road_first_lane", "lanes", "points", "lanechg_attribute", "lane_width_cm",
                                                        "junctions", "jpoints")}
        d = MapDesc(len(keep["road_first_lane"]) - 1, len(keep["lanes"]), len(keep["points"]), len(keep["junctions"]),
                    len(keep["jpoints"]), 0, *[_ptr(keep[k]) if len(keep[k]) else None for k in
                                               ("road_first_lane", "lanes", "points", "lanechg_attribute", "lane_width_cm",
                                                "junctions", "jpoints")])
        _check(self.lib.pp_set_map(self.h, C.addressof(d)))

    def set_egos(self, sc, with_motion=True):
        """Scenes on the resident map (pp_set_egos): lane views and junction slices are derived on the device."""
        n = len(sc["scene_in"])
        _check(self.lib.pp_set_egos(self.h, n, _ptr(sc["scene_in"]), _ptr(sc["obs_pool"]),
                                    _ptr(sc["mot_pool"]) if with_motion else None, n * sc["n_obs"]))
        self.n = n

    def get_scene_in(self):
        out = np.zeros(self.n, SceneIn)
        _check(self.lib.pp_get_scene_in(self.h, _ptr(out), self.n))
        return out

    def set_state(self, state):
        _check(self.lib.pp_set_state(self.h, _ptr(state), len(state)))

    def tick(self, sync=False):
        _check(self.lib.pp_plan_tick(self.h))
        if sync:
            self.sync()

    def sync(self):
        _check(self.lib.pp_sync(self.h))

    # ---- streamed ticks (no host wait) ---------------------------------------------------
    def update_async(self, scene_in=None, obs_pool=None, mot_pool=None, n_obs_total=None):
        """pp_update_async: the inputs of the next tick.  Arrays should be pinned (pinned_empty / pinned_copy) and must stay
        untouched until wait_tick of the tick that adopts them."""
        if n_obs_total is None:
            n_obs_total = len(obs_pool) if obs_pool is not None else 0
        _check(self.lib.pp_update_async(self.h, self.n, _ptr(scene_in), _ptr(obs_pool), _ptr(mot_pool), int(n_obs_total)))

    def fetch_async(self, plan=None, grid=None):
        """pp_fetch_async of the last enqueued tick into `plan` / `grid` (pinned arrays of self.n records). Returns the tick id."""
        t = C.c_longlong()
        _check(self.lib.pp_fetch_async(self.h, _ptr(plan), _ptr(grid), C.byref(t)))
        return t.value

    def fetch_published_async(self, result=None, show=None, grid=None):
        """pp_fetch_published_async: PlanningOut / PlanningStatus arrays (what the reference publishes) and / or GridOut of the last tick."""
        t = C.c_longlong()
        _check(self.lib.pp_fetch_published_async(self.h, _ptr(result), _ptr(show), _ptr(grid), C.byref(t)))
        return t.value

    def wait_tick(self, tick_id, allow_poisoned=False):
        """pp_wait_tick: host wait for that tick's downloads. Returns the number of poisoned scenes (raises on any unless allowed)."""
        bad = C.c_int()
        rc = self.lib.pp_wait_tick(self.h, tick_id, C.byref(bad))
        if rc != 0 and not (allow_poisoned and bad.value > 0):
            _check(rc)
        return bad.value

    def tick_id(self):
        return self.lib.pp_tick_id(self.h)

    def device_synchronize(self):
        """pp_device_synchronize: pp_sync + hipDeviceSynchronize (what torch.cuda.synchronize() does, without torch)."""
        _check(self.lib.pp_device_synchronize(self.h))

    def join(self):
        """pp_join: the handle's stream waits (on the device) for every tick enqueued so far."""
        _check(self.lib.pp_join(self.h))

    def get_plan(self):
        out = np.zeros(self.n, PlanOut)
        _check(self.lib.pp_get_plan(self.h, _ptr(out), self.n))
        return out

    def get_state(self):
        out = np.zeros(self.n, SceneState)
        _check(self.lib.pp_get_state(self.h, _ptr(out), self.n))
        return out

    def get_grid_out(self):
        out = np.zeros(self.n, GridOut)
        _check(self.lib.pp_get_grid_out(self.h, _ptr(out), self.n))
        return out

    def get_grid(self, scene):
        w, hgt = int(self.cfg["grid_w"][0]), int(self.cfg["grid_h"][0])
        out = np.zeros((hgt, w), np.uint8)
        _check(self.lib.pp_get_grid(self.h, scene, _ptr(out)))
        return out

    def search_info(self):
        """(LDS budget in words per view, words the densest scene needed, scenes of the last tick on the dense path)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.pp_get_search_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_order(self, scene, n):
        out = np.zeros(max(n, 1), np.int32)
        _check(self.lib.pp_get_order(self.h, scene, _ptr(out), n))
        return out[:n]

    def get_path(self, scene, n):
        out = np.zeros(max(n, 1), np.int32)
        _check(self.lib.pp_get_path(self.h, scene, _ptr(out), n))
        return out[:n]

    def get_refpath(self, scene, n):
        out = np.zeros(max(n, 1), GlobalPoint2D)
        _check(self.lib.pp_get_refpath(self.h, scene, _ptr(out), n))
        return out[:n]

    def plan_tick_batch(self, sc, state, with_motion=True, want_grid=True):
        """One-shot upload + tick + download (pp_plan_tick_batch). Updates `state` in place."""
        n = len(sc["scene_in"])
        plan = np.zeros(n, PlanOut)
        gout = np.zeros(n, GridOut) if want_grid else None
        _check(self.lib.pp_plan_tick_batch(self.h, n, _ptr(sc["scene_in"]), _ptr(sc["obs_pool"]),
                                           _ptr(sc["mot_pool"]) if with_motion else None, n * sc["n_obs"],
                                           _ptr(sc["lane_pool"]), _ptr(sc.get("attr_pool")), len(sc["lane_pool"]),
                                           _ptr(sc["ref_pool"]),
                                           len(sc["ref_pool"]), _ptr(state), _ptr(plan), _ptr(gout)))
        self.n = n
        return plan, gout

    # ---- stand-alone operators --------------------------------------------------------
    def search_obstacle_batch(self, paths, path_off, obs, obs_off, lat_lo, lat_hi):
        nq = len(lat_lo)
        out = np.zeros(nq, Path_Obs)
        _check(self.lib.pp_search_obstacle_batch(self.h, nq, _ptr(paths), _ptr(path_off), _ptr(obs), _ptr(obs_off),
                                                 _ptr(lat_lo), _ptr(lat_hi), _ptr(out)))
        return out

    def geom_batch(self, op, a, b=None, c=None):
        n = len(a)
        out = np.zeros(n, np.float64)
        _check(self.lib.pp_geom_batch(self.h, op, n, _ptr(a), _ptr(b), _ptr(c), _ptr(out)))
        return out

    def scalar_stage(self, op, args, last_Bpoints=None, n_out=3):
        a = np.asarray(args, np.float64).copy()
        out = np.zeros(n_out, np.float64)
        _check(self.lib.pp_scalar_stage(self.h, op, _ptr(a), len(a), _ptr(last_Bpoints), _ptr(out), n_out))
        return out

    def bezier(self, start, end, n=PATH_POINTS):
        out = np.zeros(n, GlobalPoint2D)
        _check(self.lib.pp_bezier(self.h, _P3(*start), _P3(*end), _ptr(out), n))
        return out

    def mean_points(self, pts, n_out=PATH_POINTS):
        out = np.zeros(n_out, GlobalPoint2D)
        _check(self.lib.pp_mean_points(self.h, _ptr(pts) if len(pts) else None, len(pts), _ptr(out), n_out))
        return out

    def create_new_path(self, path, offset):
        out = np.zeros(len(path), GlobalPoint2D)
        _check(self.lib.pp_create_new_path(self.h, _ptr(path), len(path), float(offset), _ptr(out)))
        return out

    # ---- measurement -------------------------------------------------------------------
    def set_profile(self, on):
        _check(self.lib.pp_set_profile(self.h, int(on)))

    def reset_kernel_ms(self):
        _check(self.lib.pp_reset_kernel_ms(self.h))

    def kernel_ms(self):
        out = {}
        for k, name in enumerate(K_NAMES):
            ms, cnt = C.c_float(), C.c_int()
            _check(self.lib.pp_get_kernel_ms(self.h, k, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def device_ptr(self, which):
        sz = C.c_size_t()
        p = self.lib.pp_device_ptr(self.h, which, C.byref(sz))
        return p, sz.value
