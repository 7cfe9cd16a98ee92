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

// __ballot() takes an int: a bool predicate would be materialised as 0 / 1 and compared again; this keeps it a lane mask
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

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


// Order-dependent sums (arc lengths) are carried by ONE lane in index order, so their rounding equals a sequential host loop's;
// what makes them fast is that the lane works on blocks of eight values: the eight LDS reads are issued together (one wait),
// then eight dependent adds follow - ~10 cycles per element instead of an LDS round trip per element.

// in place: v[i] <- v[lo] + ... + v[i] for i in [lo, hi), starting from `acc`; returns the total.  One lane.
__device__ __forceinline__ double serial_prefix_inplace(double* v, int lo, int hi, double acc)
{
    int i = lo;
    for (; i + 8 <= hi; i += 8) {
        double a0 = v[i], a1 = v[i + 1], a2 = v[i + 2], a3 = v[i + 3], a4 = v[i + 4], a5 = v[i + 5], a6 = v[i + 6], a7 = v[i + 7];
        a0 = acc + a0; a1 = a0 + a1; a2 = a1 + a2; a3 = a2 + a3; a4 = a3 + a4; a5 = a4 + a5; a6 = a5 + a6; a7 = a6 + a7;
        v[i] = a0; v[i + 1] = a1; v[i + 2] = a2; v[i + 3] = a3; v[i + 4] = a4; v[i + 5] = a5; v[i + 6] = a6; v[i + 7] = a7;
        acc = a7;
    }
    for (; i < hi; i++) { acc += v[i]; v[i] = acc; }
    return acc;
}
// v[lo] + ... + v[hi-1] added to `acc` in index order.  One lane.
__device__ __forceinline__ double serial_sum(const double* v, int lo, int hi, double acc)
{
    int i = lo;
    for (; i + 8 <= hi; i += 8) {
        const double a0 = v[i], a1 = v[i + 1], a2 = v[i + 2], a3 = v[i + 3], a4 = v[i + 4], a5 = v[i + 5], a6 = v[i + 6], a7 = v[i + 7];
        acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
    }
    for (; i < hi; i++) acc += v[i];
    return acc;
}
// Adds v[0 .. cnt) to `sum` in index order and returns the first k with  sum - 4 > limit  (sum then holds the value at k),
// or -1 with sum = the total.  One lane.  (The walk of SearchAimPoint, Planning.cpp:410-432.)
__device__ __forceinline__ int serial_walk(const double* v, int cnt, double limit, double& sum)
{
    double acc = sum;
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {
        const double a0 = v[i], a1 = v[i + 1], a2 = v[i + 2], a3 = v[i + 3], a4 = v[i + 4], a5 = v[i + 5], a6 = v[i + 6], a7 = v[i + 7];
        const double s0 = acc + a0, s1 = s0 + a1, s2 = s1 + a2, s3 = s2 + a3, s4 = s3 + a4, s5 = s4 + a5, s6 = s5 + a6, s7 = s6 + a7;
        if (s7 - 4 > limit) {                              // the sums never decrease (lengths >= 0) but may be NaN: test each in order
            const double ss[8] = { s0, s1, s2, s3, s4, s5, s6, s7 };
#pragma unroll
            for (int k = 0; k < 8; k++) if (ss[k] - 4 > limit) { sum = ss[k]; return i + k; }
        }
        if (!(s7 - 4 <= limit)) {                          // a NaN among them: fall back to the one-by-one test for this block
            const double ss[8] = { s0, s1, s2, s3, s4, s5, s6, s7 };
#pragma unroll
            for (int k = 0; k < 8; k++) if (ss[k] - 4 > limit) { sum = ss[k]; return i + k; }
        }
        acc = s7;
    }
    for (; i < cnt; i++) { acc += v[i]; if (acc - 4 > limit) { sum = acc; return i; } }
    sum = acc;
    return -1;
}

// Cumulative arc length of a polyline, executed by one full wave: cum[0] = 0, cum[i] = cum[i-1] + |P[i] - P[i-1]|, the
// additions in index order.  Segment lengths in parallel, the running sum by lane 0 (serial_prefix_inplace).
// cum must hold n doubles.
__device__ __forceinline__ void wave_cumlen(const GlobalPoint2D* path, int n, double* cum, int lane)
{
    for (int i = 1 + lane; i < n; i += DMPP_WAVE) cum[i] = CalcDistance(path[i], path[i - 1]);
    wave_sync();
    if (lane == 0 && n > 0) { cum[0] = 0; serial_prefix_inplace(cum, 1, n, 0.0); }
    wave_sync();
}

__device__ __forceinline__ double readlane_f64(double v, int src)       // src must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
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
        int i = 0;
        for (; i + 4 <= n; i += 4) {                         // four points per round: the (broadcast) reads are issued together
            const GlobalPoint2D p0 = path[i], p1 = path[i + 1], p2 = path[i + 2], p3 = path[i + 3];
            const double a0 = ox - p0.x, b0 = oy - p0.y, a1 = ox - p1.x, b1 = oy - p1.y, a2 = ox - p2.x, b2 = oy - p2.y, a3 = ox - p3.x, b3 = oy - p3.y;
            const double e0 = a0 * a0 + b0 * b0, e1 = a1 * a1 + b1 * b1, e2 = a2 * a2 + b2 * b2, e3 = a3 * a3 + b3 * b3;
            if (e0 < best) { best = e0; bi = i; }
            if (e1 < best) { best = e1; bi = i + 1; }
            if (e2 < best) { best = e2; bi = i + 2; }
            if (e3 < best) { best = e3; bi = i + 3; }
        }
        for (; i < n; i++) {
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

// CShare::SearchObstacle for one polyline by a TEAM of TW waves of a 256-thread block (TW = 4: the whole block on one
// polyline; TW = 2: two polylines side by side).  The team's first wave sums the arc lengths while every wave of the team
// scans its share [n*k/TW, n*(k+1)/TW) of the points for all obstacles; the shares are merged per obstacle in point order
// with the strict < of the sequential loop (so the first minimum wins), and the first wave finishes as wave_search_obstacle
// does.  EVERY thread of the block must call it (block-wide barriers), with the same m; a team without a job passes
// active = false.  part_d2 / part_bi: 256 entries each.  The result is valid in the team's first wave.
template <int TW>
__device__ __forceinline__ SoResult team_search_obstacle(const PlannerConfig& c, const GlobalPoint2D* path, int n, double* s,
                                                         const ObPoint* obs, int m, double lat_lo, double lat_hi, bool active,
                                                         double* part_d2, int* part_bi)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rank = wave % TW;
    const bool leader = rank == 0;
    SoResult r;
    r.flag = 0; r.path_id = 0; r.ob_index = -1; r.dis_lat = c.NO_OBSTACLE_DIS; r.dis_lng = c.NO_OBSTACLE_DIS;
    const bool run = active && n >= 2 && m >= 1;
    if (run && leader) wave_cumlen(path, n, s, lane);
    const int q0 = run ? (int)((long long)n * rank / TW) : 0, q1 = run ? (int)((long long)n * (rank + 1) / TW) : 0;
    double best_lng = __builtin_inf(), best_lat = 0;
    int best_j = 0x7fffffff, best_id = 0;
    for (int j0 = 0; j0 < m; j0 += DMPP_WAVE) {
        const int j = j0 + lane;
        double best = __builtin_inf(); int bi = 0;
        double ox = 0, oy = 0;
        if (run && j < m) {
            ox = obs[j].x; oy = obs[j].y;
            int i = q0;
            for (; i + 4 <= q1; i += 4) {                    // four points per round: the (broadcast) reads are issued together
                const GlobalPoint2D p0 = path[i], p1 = path[i + 1], p2 = path[i + 2], p3 = path[i + 3];
                const double a0 = ox - p0.x, b0 = oy - p0.y, a1 = ox - p1.x, b1 = oy - p1.y, a2 = ox - p2.x, b2 = oy - p2.y, a3 = ox - p3.x, b3 = oy - p3.y;
                const double e0 = a0 * a0 + b0 * b0, e1 = a1 * a1 + b1 * b1, e2 = a2 * a2 + b2 * b2, e3 = a3 * a3 + b3 * b3;
                if (e0 < best) { best = e0; bi = i; }
                if (e1 < best) { best = e1; bi = i + 1; }
                if (e2 < best) { best = e2; bi = i + 2; }
                if (e3 < best) { best = e3; bi = i + 3; }
            }
            for (; i < q1; i++) {
                const double dx = ox - path[i].x, dy = oy - path[i].y, d2 = dx * dx + dy * dy;
                if (d2 < best) { best = d2; bi = i; }
            }
        }
        part_d2[tid] = best; part_bi[tid] = bi;
        __syncthreads();
        if (run && leader && j < m) {
            best = __builtin_inf(); bi = 0;
#pragma unroll
            for (int k = 0; k < TW; k++) { const double d = part_d2[tid + k * DMPP_WAVE]; if (d < best) { best = d; bi = part_bi[tid + k * DMPP_WAVE]; } }
            const int idx = (bi == n - 1) ? n - 2 : bi;
            const GlobalPoint2D a = path[idx], b = path[idx + 1], o = { ox, oy };
            bool ok = true;
            if (bi == 0) { double t = (ox - a.x) * (b.x - a.x) + (oy - a.y) * (b.y - a.y); if (t < 0) ok = false; }
            if (bi == n - 1) { double t = (ox - b.x) * (b.x - a.x) + (oy - b.y) * (b.y - a.y); if (t > 0) ok = false; }
            if (ok) {
                const double lat = GetLatDis(c, o, a, b);
                if (!(lat < lat_lo || lat > lat_hi)) {
                    const double lng = s[bi];
                    if (best_j == 0x7fffffff || lng < best_lng) { best_lng = lng; best_lat = lat; best_j = j; best_id = bi; }
                }
            }
        }
        __syncthreads();
    }
    if (run && leader) {
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
    }
    return r;
}

}  // namespace dmpp
