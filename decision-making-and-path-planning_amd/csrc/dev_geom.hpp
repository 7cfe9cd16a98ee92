// dev_geom.hpp — device-side geometry of the planning path (gfx950, wave64).
//
// Scalar helpers follow the reference bodies (Planning.cpp:686-786,1000-1019); the CShare
// helpers (BezierPlanning, MeanPoints, CreateNewPath, SearchObstacle) follow the repo's
// specification in DESIGN.md §4.  All arithmetic is f64 and the translation unit is built
// with -ffp-contract=off, so +,-,*,/,sqrt round exactly as on the host.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dmpp_types.h"

#define DMPP_WAVE 64

namespace dmpp {

// LDS / global visibility inside ONE wave: LDS and VMEM operations of a wave execute in
// program order; this only stops the compiler from moving or caching across the point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Compiler-only ordering inside one wave (no s_waitcnt): LDS operations of a wave execute in
// issue order, and so do its vector-memory operations on one address.
__device__ __forceinline__ void wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64); hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

// Planning.h:54
__device__ __forceinline__ int Sgn(double a) { return a > 0 ? 1 : -1; }

__device__ __forceinline__ double CalcDistance(GlobalPoint2D a, GlobalPoint2D b)
{
    double dx = a.x - b.x, dy = a.y - b.y;
    return sqrt(dx * dx + dy * dy);
}

// Planning.cpp:686-709
__device__ __forceinline__ double GetLatDis(const PlannerConfig& c, GlobalPoint2D cur_pt, GlobalPoint2D pt, GlobalPoint2D pt_next)
{
    double lat_dis;
    if (fabs(pt.x - pt_next.x) > c.EPSILON) {
        double k = (pt.y - pt_next.y) / (pt.x - pt_next.x);
        lat_dis = fabs((cur_pt.y - pt.y) - k * (cur_pt.x - pt.x)) / sqrt(1 + k * k);
    } else {
        lat_dis = fabs(pt.x - cur_pt.x);
    }
    if (lat_dis < c.EPSILON) lat_dis = 0;
    else lat_dis = lat_dis * Sgn((pt_next.x - pt.x) * (cur_pt.y - pt.y) - (pt_next.y - pt.y) * (cur_pt.x - pt.x));
    return lat_dis;
}

// Planning.cpp:719-750 — keeps the atan branch structure (no atan2: the EPSILON branches differ)
__device__ __forceinline__ double GetRoadAngle(const PlannerConfig& c, GlobalPoint2D a, GlobalPoint2D b)
{
    double angle;
    const double PI = c.PI, EPS = c.EPSILON;
    if (fabs(b.x - a.x) < EPS && fabs(b.y - a.y) < EPS) angle = 0;
    else if (fabs(b.x - a.x) < EPS) angle = (b.y > a.y) ? PI / 2 : 3 * PI / 2;
    else {
        angle = atan((b.y - a.y) / (b.x - a.x));
        if (b.x < a.x) angle = angle + PI;
        else if ((b.x > a.x) && (b.y < a.y)) angle = angle + 2 * PI;
    }
    return angle * 180 / PI;
}

// Planning.cpp:760-786
__device__ __forceinline__ double GetAngleErr(double dir1, double dir2)
{
    double e = dir2 - dir1;
    if (dir1 < 180) { if (!(dir2 - dir1 <= 180)) e = dir2 - dir1 - 360; }
    else            { if (!(dir2 - dir1 > -180)) e = dir2 - dir1 + 360; }
    return e;
}

// Planning.cpp:1004-1017 with the NaN / zero-length cases fenced to 1000 (grid scoring only)
__device__ __forceinline__ double radius3_fenced(GlobalPoint2D a, GlobalPoint2D m, GlobalPoint2D f)
{
    double dis1 = sqrt((a.x - m.x) * (a.x - m.x) + (a.y - m.y) * (a.y - m.y));
    double dis2 = sqrt((m.x - f.x) * (m.x - f.x) + (m.y - f.y) * (m.y - f.y));
    double dis3 = sqrt((a.x - f.x) * (a.x - f.x) + (a.y - f.y) * (a.y - f.y));
    double den = 2 * dis1 * dis2;
    if (!(den > 0)) return 1000;
    double cosA = (dis1 * dis1 + dis2 * dis2 - dis3 * dis3) / den;
    double sinA = sqrt(1 - cosA * cosA);
    if (!(sinA >= 0.001)) return 1000;
    return 0.5 * dis3 / sinA;
}

// CShare::BezierPlanning control data, computed once per curve
struct Bezier {
    double x0, y0, x1, y1, x2, y2, x3, y3;
};
__device__ __forceinline__ Bezier bezier_setup(const PlannerConfig& c, GlobalPoint3D s, GlobalPoint3D e)
{
    Bezier b;
    double th0 = s.dir * c.PI / 180, th1 = e.dir * c.PI / 180;
    double dx = e.x - s.x, dy = e.y - s.y;
    double d = sqrt(dx * dx + dy * dy) / 3;
    b.x0 = s.x; b.y0 = s.y; b.x3 = e.x; b.y3 = e.y;
    b.x1 = s.x + d * cos(th0); b.y1 = s.y + d * sin(th0);
    b.x2 = e.x - d * cos(th1); b.y2 = e.y - d * sin(th1);
    return b;
}
__device__ __forceinline__ GlobalPoint2D bezier_point(const Bezier& b, int i, int n)
{
    double t = (n > 1) ? (double)i / (double)(n - 1) : 0.0;
    double u = 1 - t;
    double b0 = u * u * u, b1 = 3 * u * u * t, b2 = 3 * u * t * t, b3 = t * t * t;
    GlobalPoint2D p;
    p.x = b0 * b.x0 + b1 * b.x1 + b2 * b.x2 + b3 * b.x3;
    p.y = b0 * b.y0 + b1 * b.y1 + b2 * b.y2 + b3 * b.y3;
    return p;
}

// CShare::CreateNewPath, one output point (path may be LDS or global)
__device__ __forceinline__ GlobalPoint2D offset_point(const PlannerConfig& c, const GlobalPoint2D* path, int n, int i, double offset)
{
    int ia = i > 0 ? i - 1 : 0, ib = i < n - 1 ? i + 1 : n - 1;
    double tx = path[ib].x - path[ia].x, ty = path[ib].y - path[ia].y;
    double L = sqrt(tx * tx + ty * ty);
    GlobalPoint2D o = path[i];
    if (!(L < c.EPSILON)) {
        o.x = path[i].x + offset * (ty / L);
        o.y = path[i].y + offset * (-tx / L);
    }
    return o;
}

// CShare::MeanPoints given the cumulative arc length cum[0..n_in) (summed left to right
// by ONE lane so the rounding matches a sequential host loop).
__device__ __forceinline__ GlobalPoint2D mean_point(const PlannerConfig& c, const GlobalPoint2D* in, const double* cum,
                                                    int n_in, int k, int n_out)
{
    GlobalPoint2D o;
    if (n_in <= 0) { o.x = 0; o.y = 0; return o; }
    double L = cum[n_in - 1];
    if (n_in == 1 || L < c.EPSILON) return in[0];
    double s = (n_out > 1) ? (L * (double)k) / (double)(n_out - 1) : 0.0;
    // smallest j in [0, n_in-2] with cum[j+1] >= s (binary search: cum is non-decreasing)
    int lo = 0, hi = n_in - 2;
    if (!(cum[n_in - 1] >= s)) lo = n_in - 2;
    else while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid + 1] >= s) hi = mid; else lo = mid + 1; }
    int j = lo;
    double seg = cum[j + 1] - cum[j];
    double r = (seg > c.EPSILON) ? (s - cum[j]) / seg : 0.0;
    o.x = in[j].x + r * (in[j + 1].x - in[j].x);
    o.y = in[j].y + r * (in[j + 1].y - in[j].y);
    return o;
}

// Cumulative arc length of a polyline, executed by one full wave: segment lengths in
// parallel, the running sum by lane 0 in index order.  cum must hold n doubles.
__device__ __forceinline__ void wave_cumlen(const GlobalPoint2D* path, int n, double* cum, int lane)
{
    for (int i = 1 + lane; i < n; i += DMPP_WAVE) cum[i] = CalcDistance(path[i], path[i - 1]);
    wave_sync();
    if (lane == 0 && n > 0) {
        double acc = 0; cum[0] = 0;
        for (int i = 1; i < n; i++) { acc += cum[i]; cum[i] = acc; }
    }
    wave_sync();
}

struct SoResult {
    double dis_lat, dis_lng;
    int flag, path_id, ob_index;
};

// CShare::SearchObstacle for one polyline, executed by one full wave (all 64 lanes active).
// Lanes take obstacles j = lane, lane+64, ...; each walks the n path points (LDS broadcast
// reads) keeping the first minimum of the squared distance, exactly as a sequential loop
// would; the wave then reduces on (arc length, obstacle index).  s: n doubles of scratch.
__device__ __forceinline__ SoResult wave_search_obstacle(const PlannerConfig& c, const GlobalPoint2D* path, int n, double* s,
                                                         const ObPoint* obs, int m, double lat_lo, double lat_hi, int lane)
{
    SoResult r;
    r.flag = 0; r.path_id = 0; r.ob_index = -1; r.dis_lat = c.NO_OBSTACLE_DIS; r.dis_lng = c.NO_OBSTACLE_DIS;
    if (n < 2 || m < 1) return r;
    wave_cumlen(path, n, s, lane);
    double best_lng = __builtin_inf(), best_lat = 0;
    int best_j = 0x7fffffff, best_id = 0;
    for (int j = lane; j < m; j += DMPP_WAVE) {
        const double ox = obs[j].x, oy = obs[j].y;
        double best = __builtin_inf(); int bi = 0;
        for (int i = 0; i < n; i++) {
            double dx = ox - path[i].x, dy = oy - path[i].y;
            double d2 = dx * dx + dy * dy;
            if (d2 < best) { best = d2; bi = i; }
        }
        int idx = (bi == n - 1) ? n - 2 : bi;
        GlobalPoint2D a = path[idx], b = path[idx + 1], o = { ox, oy };
        bool ok = true;
        if (bi == 0) { double t = (ox - a.x) * (b.x - a.x) + (oy - a.y) * (b.y - a.y); if (t < 0) ok = false; }
        if (bi == n - 1) { double t = (ox - b.x) * (b.x - a.x) + (oy - b.y) * (b.y - a.y); if (t > 0) ok = false; }
        if (ok) {
            double lat = GetLatDis(c, o, a, b);
            if (!(lat < lat_lo || lat > lat_hi)) {
                double lng = s[bi];
                if (best_j == 0x7fffffff || lng < best_lng) { best_lng = lng; best_lat = lat; best_j = j; best_id = bi; }
            }
        }
    }
    // reduce on (lng, j): the sequential loop keeps the first obstacle with the smallest lng
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        double o_lng = shfl_xor_f64(best_lng, sft);
        double o_lat = shfl_xor_f64(best_lat, sft);
        int o_j = __shfl_xor(best_j, sft, 64), o_id = __shfl_xor(best_id, sft, 64);
        bool take = (o_j != 0x7fffffff) && (best_j == 0x7fffffff || o_lng < best_lng || (o_lng == best_lng && o_j < best_j));
        if (take) { best_lng = o_lng; best_lat = o_lat; best_j = o_j; best_id = o_id; }
    }
    if (best_j != 0x7fffffff) { r.flag = 1; r.dis_lat = best_lat; r.dis_lng = best_lng; r.path_id = best_id; r.ob_index = best_j; }
    return r;
}

}  // namespace dmpp
