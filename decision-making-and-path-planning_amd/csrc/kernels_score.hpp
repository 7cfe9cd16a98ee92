// kernels_score.hpp — G3: candidate scoring (specification: DESIGN.md §5, oracle/dmpp_grid_oracle.c).  One device function,
// used by the search kernel for the scene it has just searched (all waves of the workgroup) and by the stand-alone k_score.
#pragma once
#include "dev_geom.hpp"

namespace dmpp {

// G3.  256 threads = 4 waves per scene; wave w scores candidates w, w+4, ...
// NW waves per scene score the candidates NW at a time: 4 for batches (four scenes per CU), 16 for the few scenes of a
// latency-bound tick (17 candidates in 2 rounds instead of 5).
constexpr int kMaxRelObs = 128;      // culled obstacle list kept in LDS; a scene with more candidates near its paths reads the whole list from HBM
template <int NW>
struct ScoreShared {
    GlobalPoint2D cand[NW][DMPP_PATH_POINTS];
    double seg[NW][DMPP_PATH_POINTS];        // |P_i - P_{i+1}| of the wave's current candidate
    GlobalPoint2D pts[DMPP_PATH_POINTS];     // grid-path prefix in metres (lookahead_cells+1 <= 200)
    double cum[DMPP_PATH_POINTS];
    double rx[kMaxRelObs], ry[kMaxRelObs], rr[kMaxRelObs], rt2[kMaxRelObs];   // obstacles that can matter: x, y, radius, cutoff^2
    double bx0[DMPP_MAX_LATTICE], bx1[DMPP_MAX_LATTICE], by0[DMPP_MAX_LATTICE], by1[DMPP_MAX_LATTICE];
    double cost[DMPP_MAX_LATTICE];
    int best, n_rel;
};

__device__ __forceinline__ double wave_tree_sum(double acc)
{
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) acc += shfl_xor_f64(acc, sft);
    return acc;
}

// Scores the candidates of one scene and writes cand_* / best_* of its GridOut.  All NW waves of the workgroup; `status` and
// `path_len` are the search's results for the scene, `path` its cells, `gobs` / `m` the obstacle snapshot of the tick.
template <int NW>
__device__ __forceinline__ void score_body(const PlannerConfig& c, const SceneIn& si, const ObPoint* __restrict__ gobs, int m,
                                           const int32_t* __restrict__ path, GridOut& go, int status, int path_len, ScoreShared<NW>& sh)
{
    constexpr int kThreads = NW * DMPP_WAVE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int W = c.grid_w;
    const GlobalPoint3D ego = si.loc.globalpoint;
    const bool have_path = (status == DMPP_G_FOUND) && path_len >= 1;
    int a = 0;
    GlobalPoint2D T; double thT;
    if (have_path) {
        a = min(path_len - 1, c.lookahead_cells);
        if (a > DMPP_PATH_POINTS - 1) a = DMPP_PATH_POINTS - 1;
        const int a0 = max(a - 4, 0);
        const int pa = path[a], p0 = path[a0];
        T.x = si.grid_origin.x + ((double)(pa % W) + 0.5) * c.cell;
        T.y = si.grid_origin.y + ((double)(pa / W) + 0.5) * c.cell;
        if (a0 == a) thT = ego.dir;
        else {
            GlobalPoint2D P0 = { si.grid_origin.x + ((double)(p0 % W) + 0.5) * c.cell, si.grid_origin.y + ((double)(p0 / W) + 0.5) * c.cell };
            thT = GetRoadAngle(c, P0, T);
        }
        for (int i = tid; i <= a; i += kThreads) {
            const int pc = path[i];
            sh.pts[i].x = si.grid_origin.x + ((double)(pc % W) + 0.5) * c.cell;
            sh.pts[i].y = si.grid_origin.y + ((double)(pc / W) + 0.5) * c.cell;
        }
    } else {
        T = si.goal;
        GlobalPoint2D e2 = { ego.x, ego.y };
        thT = GetRoadAngle(c, e2, si.goal);
    }
    __syncthreads();
    if (have_path && wave == 0) wave_cumlen(sh.pts, a + 1, sh.cum, lane);
    __syncthreads();
    const int nl = min(c.n_lattice, DMPP_MAX_LATTICE - 1);
    const int nc = nl + (have_path ? 1 : 0);
    // the trigonometry is the same for every candidate: start heading, terminal heading
    const double th0 = ego.dir * c.PI / 180, c0 = cos(th0), s0 = sin(th0);
    const double th = thT * c.PI / 180, cs = cos(th), sn = sin(th);
    const double half_w = 0.5 * c.Vehicle_Width;
    auto lattice_curve = [&](int k, double& off) -> Bezier {        // BezierPlanning(ego -> terminal k), as bezier_setup
        off = (double)(k - (nl - 1) / 2) * c.lattice_step;
        Bezier bz;
        bz.x0 = ego.x; bz.y0 = ego.y; bz.x3 = T.x + off * sn; bz.y3 = T.y + off * (-cs);
        const double dx = bz.x3 - bz.x0, dy = bz.y3 - bz.y0;
        const double d = sqrt(dx * dx + dy * dy) / 3;
        bz.x1 = bz.x0 + d * c0; bz.y1 = bz.y0 + d * s0;
        bz.x2 = bz.x3 - d * cs; bz.y2 = bz.y3 - d * sn;
        return bz;
    };
    // ---- obstacles that can matter at all: inside the box of every candidate grown by their cutoff.
    //      A Bezier lies in the hull of its control points; the grid path stays within a+1 cells of its first cell.
    if (tid < nl) {
        double off; const Bezier bz = lattice_curve(tid, off);
        sh.bx0[tid] = fmin(fmin(bz.x0, bz.x1), fmin(bz.x2, bz.x3)); sh.bx1[tid] = fmax(fmax(bz.x0, bz.x1), fmax(bz.x2, bz.x3));
        sh.by0[tid] = fmin(fmin(bz.y0, bz.y1), fmin(bz.y2, bz.y3)); sh.by1[tid] = fmax(fmax(bz.y0, bz.y1), fmax(bz.y2, bz.y3));
    }
    if (tid == 0) sh.n_rel = 0;
    __syncthreads();
    double X0 = __builtin_inf(), X1 = -__builtin_inf(), Y0 = __builtin_inf(), Y1 = -__builtin_inf();
    for (int k = 0; k < nl; k++) { X0 = fmin(X0, sh.bx0[k]); X1 = fmax(X1, sh.bx1[k]); Y0 = fmin(Y0, sh.by0[k]); Y1 = fmax(Y1, sh.by1[k]); }
    if (have_path) {
        // around the centre of the path's first cell, not around the ego: an ego outside the grid is clamped to a border cell
        const double ext = (double)(a + 2) * c.cell;
        const int pc0 = path[0];
        const double px0 = si.grid_origin.x + ((double)(pc0 % W) + 0.5) * c.cell, py0 = si.grid_origin.y + ((double)(pc0 / W) + 0.5) * c.cell;
        X0 = fmin(X0, px0 - ext); X1 = fmax(X1, px0 + ext); Y0 = fmin(Y0, py0 - ext); Y1 = fmax(Y1, py0 + ext);
    }
    // the obstacles near the candidates, collected in LDS (kMaxRelObs of them); when more qualify, the scoring loop reads the
    // whole list from HBM instead (the same minimum over the same thresholds either way)
    for (int j = tid; j < m; j += kThreads) {
        const ObPoint o = gobs[j];
        const double thr = (double)o.radius + half_w + c.d_safe;
        if (o.x >= X0 - thr && o.x <= X1 + thr && o.y >= Y0 - thr && o.y <= Y1 + thr) {
            const int q = atomicAdd(&sh.n_rel, 1);           // order is irrelevant: only a minimum is taken
            if (q < kMaxRelObs) { sh.rx[q] = o.x; sh.ry[q] = o.y; sh.rr[q] = (double)o.radius; sh.rt2[q] = thr * thr; }
        }
    }
    __syncthreads();
    const bool culled = sh.n_rel <= kMaxRelObs;
    const int n_rel = culled ? sh.n_rel : m;
    GlobalPoint2D* cand = sh.cand[wave];
    double* seg = sh.seg[wave];
    for (int k = wave; k < nc; k += NW) {
        double off = 0;
        if (k < nl) {
            const Bezier bz = lattice_curve(k, off);
            for (int i = lane; i < DMPP_PATH_POINTS; i += DMPP_WAVE) cand[i] = bezier_point(bz, i, DMPP_PATH_POINTS);
        } else {
            for (int i = lane; i < DMPP_PATH_POINTS; i += DMPP_WAVE) cand[i] = mean_point(c, sh.pts, sh.cum, a + 1, i, DMPP_PATH_POINTS);
        }
        wave_sync();
        // |P_i - P_{i+1}|: the dis1 / dis2 of the circumradius of neighbouring triples, computed once
        for (int i = lane; i < DMPP_PATH_POINTS - 1; i += DMPP_WAVE) {
            const GlobalPoint2D p = cand[i], q = cand[i + 1];
            seg[i] = sqrt((p.x - q.x) * (p.x - q.x) + (p.y - q.y) * (p.y - q.y));
        }
        wave_sync();
        double pen_acc = 0, k2_acc = 0; int first_hit = DMPP_PATH_POINTS;
        for (int q = 0; q < 4; q++) {
            const int i = lane + 64 * q;
            if (i < DMPP_PATH_POINTS) {
                const GlobalPoint2D p = cand[i];
                double clear = __builtin_inf();
                if (culled) {
                    for (int j = 0; j < n_rel; j++) {
                        const double dx = p.x - sh.rx[j], dy = p.y - sh.ry[j];
                        const double d2 = dx * dx + dy * dy;
                        if (d2 > sh.rt2[j]) continue;                 // cannot produce a penalty (no sqrt needed)
                        const double v = sqrt(d2) - sh.rr[j];
                        if (v < clear) clear = v;
                    }
                } else {
                    for (int j = 0; j < m; j++) {
                        const double dx = p.x - gobs[j].x, dy = p.y - gobs[j].y;
                        const double d2 = dx * dx + dy * dy;
                        const double thr = (double)gobs[j].radius + half_w + c.d_safe;
                        if (d2 > thr * thr) continue;
                        const double v = sqrt(d2) - (double)gobs[j].radius;
                        if (v < clear) clear = v;
                    }
                }
                clear = clear - half_w;
                double pen;
                if (clear <= 0) { pen = 1000.0; if (i < first_hit) first_hit = i; }
                else if (clear < c.d_safe) { const double qq = (c.d_safe - clear) / c.d_safe; pen = qq * qq; }
                else pen = 0;
                pen_acc += pen;
                if (i >= 1 && i <= DMPP_PATH_POINTS - 2) {
                    // radius3_fenced(P[i-1], P[i], P[i+1]) with dis1 = seg[i-1], dis2 = seg[i]
                    const GlobalPoint2D pa = cand[i - 1], pf = cand[i + 1];
                    const double dis1 = seg[i - 1], dis2 = seg[i];
                    const double dis3 = sqrt((pa.x - pf.x) * (pa.x - pf.x) + (pa.y - pf.y) * (pa.y - pf.y));
                    const double den = 2 * dis1 * dis2;
                    // curvature 1 / R with R = 0.5 * dis3 / sinA (fenced to 1000), as one division: sinA / (0.5 * dis3) - the
                    // oracle's 1 / (0.5 * dis3 / sinA) to an ulp or two (the scores are compared to 1e-6)
                    double kk = 0.001;
                    if (den > 0) {
                        const double cosA = (dis1 * dis1 + dis2 * dis2 - dis3 * dis3) / den;
                        const double sinA = sqrt(1 - cosA * cosA);
                        if (sinA >= 0.001) kk = sinA / (0.5 * dis3);
                    }
                    k2_acc += kk * kk;
                }
            }
        }
        const double col = wave_tree_sum(pen_acc), curv = wave_tree_sum(k2_acc);
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) first_hit = min(first_hit, __shfl_xor(first_hit, sft, 64));
        const double prog = (first_hit == DMPP_PATH_POINTS) ? 0.0 : (double)(DMPP_PATH_POINTS - first_hit) / (double)DMPP_PATH_POINTS;
        const double cost = c.w_col * col + c.w_curv * curv + c.w_prog * prog + c.w_off * fabs(off);
        if (lane == 0) { go.cand_col[k] = col; go.cand_curv[k] = curv; go.cand_prog[k] = prog; go.cand_cost[k] = cost; sh.cost[k] = cost; }
        wave_sync();
    }
    __syncthreads();
    if (tid == 0) {
        double best = 0; int bi = 0;
        for (int k = 0; k < nc; k++) if (k == 0 || sh.cost[k] < best) { best = sh.cost[k]; bi = k; }
        sh.best = bi; go.best_candidate = bi; go.n_candidates = nc;
        for (int k = nc; k < DMPP_MAX_LATTICE; k++) { go.cand_cost[k] = 0; go.cand_col[k] = 0; go.cand_curv[k] = 0; go.cand_prog[k] = 0; }
    }
    __syncthreads();
    {   // regenerate the winner into the output
        const int k = sh.best;
        if (tid < DMPP_PATH_POINTS) {
            GlobalPoint2D p = { 0.0, 0.0 };                 // no candidate at all (n_lattice = 0 and no grid path): zeros
            if (k < nl) {
                double off; const Bezier bz = lattice_curve(k, off);
                p = bezier_point(bz, tid, DMPP_PATH_POINTS);
            } else if (nc > 0) p = mean_point(c, sh.pts, sh.cum, a + 1, tid, DMPP_PATH_POINTS);
            go.best_path[tid] = p;
        }
    }
}

}  // namespace dmpp
