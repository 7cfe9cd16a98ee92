"""ctypes binding of oracle/liboracle.so — the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED: see oracle/dmpp_oracle.h.
"""
import ctypes as C

import numpy as np

import dmpp_amd as dm

vp, ci, cd = C.c_void_p, C.c_int, C.c_double


class P2(C.Structure):
    _fields_ = [("x", cd), ("y", cd)]


class P3(C.Structure):
    _fields_ = [("x", cd), ("y", cd), ("dir", cd)]


def _p(a):
    return None if a is None else a.ctypes.data


class Oracle:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        L.orc_GetLatDis.restype = cd
        L.orc_GetLatDis.argtypes = [vp, P2, P2, P2]
        L.orc_GetRoadAngle.restype = cd
        L.orc_GetRoadAngle.argtypes = [vp, P2, P2]
        L.orc_GetAngleErr.restype = cd
        L.orc_GetAngleErr.argtypes = [cd, cd]
        L.orc_CalcDistance.restype = cd
        L.orc_CalcDistance.argtypes = [P2, P2]
        L.orc_Calculate_aim_dis.argtypes = [vp, vp, vp, vp]
        L.orc_GetVhclLocalState.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_UpdatePlanJudge.argtypes = [vp, vp, vp, ci, vp, vp]
        L.orc_SpeedPlanning.argtypes = [ci, vp, vp, cd, cd, C.c_float, vp, vp, vp]
        L.orc_CalculateRadius.restype = cd
        L.orc_CalculateRadius.argtypes = [vp, ci, ci]
        L.orc_BezierPlanning.argtypes = [vp, P3, P3, vp, ci]
        L.orc_MeanPoints.argtypes = [vp, vp, ci, vp, ci]
        L.orc_CreateNewPath.argtypes = [vp, vp, ci, cd, vp]
        L.orc_SearchObstacle.argtypes = [vp, vp, ci, vp, ci, cd, cd, vp, vp, vp, vp]
        L.orc_rasterise.argtypes = [vp, P2, vp, ci, vp]
        L.orc_rasterise_bruteforce.argtypes = [vp, P2, vp, ci, vp]
        L.orc_grid_search.argtypes = [vp, vp, ci, ci, vp, vp, ci, vp, ci]
        L.orc_grid_score.argtypes = [vp, vp, vp, ci, vp, vp]
        L.orc_cell_of.argtypes = [vp, P2, cd, cd]
        L.orc_effective_obstacles.argtypes = [vp, vp, vp, ci, ci, vp]
        L.orc_plan_tick.argtypes = [vp] * 10 + [vp, vp, ci, vp, ci]
        L.orc_plan_tick_batch.argtypes = [vp, ci] + [vp] * 10 + [ci]
        L.orc_plan_ticks_batch.argtypes = [vp, ci] + [vp] * 10 + [ci, ci]
        L.orc_last_refpath.argtypes = [vp, ci]
        L.orc_last_peak_open.restype = ci

    # ---- scalar helpers ----
    def GetLatDis(self, cfg, cur, pt, nxt):
        return self.L.orc_GetLatDis(_p(cfg), P2(*cur), P2(*pt), P2(*nxt))

    def GetRoadAngle(self, cfg, a, b):
        return self.L.orc_GetRoadAngle(_p(cfg), P2(*a), P2(*b))

    def GetAngleErr(self, d1, d2):
        return self.L.orc_GetAngleErr(d1, d2)

    def CalculateRadius(self, pts, near_id, front_id):
        return self.L.orc_CalculateRadius(_p(pts), near_id, front_id)

    def Calculate_aim_dis(self, cfg, loc):
        far, near = np.zeros(1, np.float32), np.zeros(1, np.float32)
        self.L.orc_Calculate_aim_dis(_p(cfg), _p(loc), _p(far), _p(near))
        return float(far[0]), float(near[0])

    def GetVhclLocalState(self, cfg, loc, last, near_id_in=0):
        lat, derr, rem = np.zeros(1), np.zeros(1), np.zeros(1)
        mid, fid = np.array([near_id_in], np.int32), np.zeros(1, np.int32)
        self.L.orc_GetVhclLocalState(_p(cfg), _p(loc), _p(last), _p(lat), _p(derr), _p(mid), _p(fid), _p(rem))
        return float(lat[0]), float(derr[0]), int(mid[0]), int(fid[0]), float(rem[0])

    def SpeedPlanning(self, ob_flag, dec, loc, lon, lat, faraim, init=(0.0, 0, 0.0)):
        bs, af, da = np.array([init[0]]), np.array([init[1]], np.int32), np.array([init[2]])
        self.L.orc_SpeedPlanning(ob_flag, _p(dec), _p(loc), lon, lat, faraim, _p(bs), _p(af), _p(da))
        return float(bs[0]), int(af[0]), float(da[0])

    def BezierPlanning(self, cfg, s, e, n=200):
        out = np.zeros(n, dm.GlobalPoint2D)
        self.L.orc_BezierPlanning(_p(cfg), P3(*s), P3(*e), _p(out), n)
        return out

    def MeanPoints(self, cfg, pts, n_out=200):
        out = np.zeros(n_out, dm.GlobalPoint2D)
        self.L.orc_MeanPoints(_p(cfg), _p(pts) if len(pts) else None, len(pts), _p(out), n_out)
        return out

    def CreateNewPath(self, cfg, path, offset):
        out = np.zeros(len(path), dm.GlobalPoint2D)
        self.L.orc_CreateNewPath(_p(cfg), _p(path), len(path), offset, _p(out))
        return out

    def SearchObstacle(self, cfg, path, obs, lo, hi):
        dl, dg = np.zeros(1), np.zeros(1)
        ob, pid = np.zeros(1, dm.ObPoint), np.zeros(1, np.int32)
        f = self.L.orc_SearchObstacle(_p(cfg), _p(path), len(path), _p(obs) if len(obs) else None, len(obs), lo, hi,
                                      _p(dl), _p(dg), _p(ob), _p(pid))
        return dict(flag=int(f), dis_lat=float(dl[0]), dis_lng=float(dg[0]), ob=ob[0], path_id=int(pid[0]))

    # ---- grid engine ----
    def rasterise(self, cfg, origin, obs, brute=False):
        w, h = int(cfg["grid_w"][0]), int(cfg["grid_h"][0])
        g = np.zeros((h, w), np.uint8)
        fn = self.L.orc_rasterise_bruteforce if brute else self.L.orc_rasterise
        fn(_p(cfg), P2(*origin), _p(obs) if len(obs) else None, len(obs), _p(g))
        return g

    def cell_of(self, cfg, origin, x, y):
        return self.L.orc_cell_of(_p(cfg), P2(*origin), x, y)

    def grid_search(self, cfg, grid, start, goal, order_cap=0):
        out = np.zeros(1, dm.GridOut)
        order = np.zeros(max(order_cap, 1), np.int32)
        mp = int(cfg["max_path"][0])
        path = np.zeros(mp, np.int32)
        self.L.orc_grid_search(_p(cfg), _p(grid), start, goal, _p(out), _p(order) if order_cap else None, order_cap, _p(path), mp)
        return out, order[: min(order_cap, int(out["n_expanded"][0]))], path[: int(out["path_len"][0])]

    def last_peak_open(self):
        """Largest number of live open-list entries during the last grid_search / plan_tick_one on this thread."""
        return self.L.orc_last_peak_open()

    def effective_obstacles(self, cfg, obs, mot, tick):
        out = np.zeros(len(obs), dm.ObPoint)
        self.L.orc_effective_obstacles(_p(cfg), _p(obs), _p(mot), len(obs), tick, _p(out))
        return out

    # ---- whole tick ----
    def plan_tick_batch(self, cfg, sc, state, n_threads=1, want_grid=True, keep_grids=False, n_ticks=1):
        """Runs n_ticks oracle ticks over all scenes (outputs of the last one); updates `state` in place."""
        n = len(sc["scene_in"])
        plan = np.zeros(n, dm.PlanOut)
        gout = np.zeros(n, dm.GridOut) if want_grid else None
        grids = None
        if keep_grids:
            grids = np.zeros((n, int(cfg["grid_h"][0]), int(cfg["grid_w"][0])), np.uint8)
        self.L.orc_plan_ticks_batch(_p(cfg), n, _p(sc["scene_in"]), _p(sc["lane_pool"]), _p(sc.get("attr_pool")), _p(sc["ref_pool"]),
                                    _p(sc["obs_pool"]), _p(sc["mot_pool"]), _p(state), _p(plan), _p(gout), _p(grids), n_threads, n_ticks)
        return plan, gout, grids

    def last_refpath(self):
        """DecisionOut.refpath of the last plan_tick_one on this thread."""
        out = np.zeros(dm.MAX_REFPATH, dm.GlobalPoint2D)
        n = self.L.orc_last_refpath(_p(out), len(out))
        return out[:n]

    def plan_tick_one(self, cfg, sc, s, state, order_cap=0):
        """One scene with expansion order and path kept."""
        plan, gout = np.zeros(1, dm.PlanOut), np.zeros(1, dm.GridOut)
        order = np.zeros(max(order_cap, 1), np.int32)
        mp = int(cfg["max_path"][0])
        path = np.zeros(mp, np.int32)
        grid = np.zeros((int(cfg["grid_h"][0]), int(cfg["grid_w"][0])), np.uint8)
        self.L.orc_plan_tick(_p(cfg), sc["scene_in"][s:s + 1].ctypes.data, _p(sc["lane_pool"]), _p(sc.get("attr_pool")), _p(sc["ref_pool"]),
                             _p(sc["obs_pool"]), _p(sc["mot_pool"]), state[s:s + 1].ctypes.data, _p(plan), _p(gout), _p(grid),
                             _p(order) if order_cap else None, order_cap, _p(path), mp)
        ne = int(gout["n_expanded"][0])
        return plan[0], gout[0], grid, order[: min(order_cap, ne)], path[: int(gout["path_len"][0])]
