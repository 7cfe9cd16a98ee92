// kernels_score.hpp — G3: candidate scoring (specification: DESIGN.md §5, oracle/dmpp_grid_oracle.c).  One device function,
// used by the search kernel for the scene it has just searched (all waves of the workgroup) and by the stand-alone k_score.
#pragma once
#include "dev_geom.hpp"

namespace dmpp {

// G3.  256 threads = 4 waves per scene; wave w scores candidates w, w+4, ...
// NW waves per scene score the candidates NW at a time: 4 for batches (four scenes per CU), 16 for the few scenes of a
// latency-bound tick (17 candidates in 2 rounds instead of 5).
constexpr int kBoxWaves = 4;
constexpr int kTailRounds = 4, kTailFirst = 190;      // packed fourth pass: candidates per wave, first point kept for it
template <int NW> constexpr bool kPackTail = NW <= 4;
template <int NW> constexpr int kTailWaves = NW <= 4 ? NW : 1;
static_assert(DMPP_PATH_POINTS - 192 == 8 && kTailFirst == 190 && kTailRounds * 8 <= DMPP_WAVE, "packed fourth pass: lane 8 c + j = point 192 + j of candidate c");
static_assert(DMPP_PATH_POINTS <= kBoxWaves * DMPP_WAVE && DMPP_MAX_LATTICE <= 32, "score_body: lane layouts");
constexpr int kMaxRelObs = 128;      // culled obstacle list kept in LDS; a scene with more candidates near its paths reads the whole list from HBM
// Many obstacles near the candidates (256-obstacle scenes keep ~10 - 30 after the cull): a coarse bucket grid over the box of
// the candidates, each culled obstacle entered into every bucket its cutoff disc can touch, so that a point only looks at the
// list of ITS bucket.  The clearance is a minimum over the obstacles within their cutoff of the point - every one of them is in
// the point's bucket - so the result is the same number (a minimum does not depend on the order or on extra far members).
constexpr int kBucketN = 12, kBucketCap = 12, kBucketMinObs = 8;
#ifdef DMPP_DEBUG_SEARCH
__device__ int g_dbg_score[8];       // debug build: scenes scored, sum of n_rel, bucketed, bucket overflow, not culled, max n_rel
#endif
template <int NW>
struct ScoreShared {
    GlobalPoint2D cand[NW][DMPP_PATH_POINTS];
    GlobalPoint2D pts[DMPP_PATH_POINTS];     // grid-path prefix in metres (lookahead_cells+1 <= 200)
    double cum[DMPP_PATH_POINTS];
    double rx[kMaxRelObs], ry[kMaxRelObs], rr[kMaxRelObs], rt2[kMaxRelObs];   // obstacles that can matter: x, y, radius, cutoff^2
    Bezier bz[DMPP_MAX_LATTICE];             // control points of the lattice candidates
    double trig[4];                          // cos, sin of the start heading; cos, sin of the terminal heading
    double bb[4];                            // box of all lattice candidates: x0, x1, y0, y1
    GlobalPoint2D T;                         // terminal point
    double cost[DMPP_MAX_LATTICE];
    int box[4][4];                           // per wave: min / max cell column and row of the grid-path prefix
    GlobalPoint2D tail[kTailWaves<NW>][kTailRounds][DMPP_PATH_POINTS - kTailFirst];   // points 190 .. 199 of a wave's candidates (packed fourth pass)
    int best, n_rel;
    int bcnt[kBucketN * kBucketN];                             // bucket fill counts (beyond kBucketCap: the bucket's points take the plain loop)
    unsigned char bent[kBucketN * kBucketN][kBucketCap];       // indices into rx / ry / rr / rt2
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
    const int nl = min(c.n_lattice, DMPP_MAX_LATTICE - 1);
    const int nc = nl + (have_path ? 1 : 0);
    const double half_w = 0.5 * c.Vehicle_Width;
    int a = 0;
    if (have_path) {
        a = min(path_len - 1, c.lookahead_cells);
        if (a > DMPP_PATH_POINTS - 1) a = DMPP_PATH_POINTS - 1;
        // the prefix in metres, and the box of its cells (per wave here, combined below): the path candidate lies inside it
        int cx0 = 0x7fffffff, cx1 = -1, cy0 = 0x7fffffff, cy1 = -1;
        for (int i = tid; i <= a; i += kThreads) {
            const int pc = path[i];
            const int px = pc % W, py = pc / W;
            sh.pts[i].x = si.grid_origin.x + ((double)px + 0.5) * c.cell;
            sh.pts[i].y = si.grid_origin.y + ((double)py + 0.5) * c.cell;
            cx0 = min(cx0, px); cx1 = max(cx1, px); cy0 = min(cy0, py); cy1 = max(cy1, py);
        }
        if (wave < kBoxWaves) {            // DMPP_PATH_POINTS <= 256 points: only the first four waves hold any
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                cx0 = min(cx0, __shfl_xor(cx0, sft, 64)); cx1 = max(cx1, __shfl_xor(cx1, sft, 64));
                cy0 = min(cy0, __shfl_xor(cy0, sft, 64)); cy1 = max(cy1, __shfl_xor(cy1, sft, 64));
            }
            if (lane == 0) { sh.box[wave][0] = cx0; sh.box[wave][1] = cx1; sh.box[wave][2] = cy0; sh.box[wave][3] = cy1; }
        }
    }
    // What is the same for every candidate is computed ONCE, by the last wave (the others have the path prefix to convert):
    // the terminal pose, and the sine and cosine of the start and terminal headings - lane 0 takes the start heading, lane 1
    // the terminal one, so one pass through sincos yields all four values.
    if (wave == NW - 1) {
        GlobalPoint2D T; double thT;
        if (have_path) {
            const int a0 = max(a - 4, 0);
            const int pa = path[a], p0 = path[a0];
            T.x = si.grid_origin.x + ((double)(pa % W) + 0.5) * c.cell;
            T.y = si.grid_origin.y + ((double)(pa / W) + 0.5) * c.cell;
            if (a0 == a) thT = ego.dir;
            else {
                GlobalPoint2D P0 = { si.grid_origin.x + ((double)(p0 % W) + 0.5) * c.cell, si.grid_origin.y + ((double)(p0 / W) + 0.5) * c.cell };
                thT = GetRoadAngle(c, P0, T);
            }
        } else {
            T = si.goal;
            GlobalPoint2D e2 = { ego.x, ego.y };
            thT = GetRoadAngle(c, e2, si.goal);
        }
        const double ang = (lane == 0 ? ego.dir : thT) * c.PI / 180;
        double sv, cv;
        sincos(ang, &sv, &cv);
        if (lane == 0) { sh.trig[0] = cv; sh.trig[1] = sv; sh.T = T; }
        if (lane == 1) { sh.trig[2] = cv; sh.trig[3] = sv; }
    }
    __syncthreads();
    if (have_path && wave == 0) wave_cumlen(sh.pts, a + 1, sh.cum, lane);
    // ---- the lattice: candidate k is the cubic Bezier of BezierPlanning(ego -> terminal k) (as bezier_setup); lane k of the
    //      last wave sets it up for every wave to read.  Obstacles that can matter at all lie inside the box of every candidate
    //      grown by their cutoff: a Bezier lies in the hull of its control points. ----
    if (wave == NW - 1) {
        double X0 = __builtin_inf(), X1 = -__builtin_inf(), Y0 = __builtin_inf(), Y1 = -__builtin_inf();
        if (lane < nl) {
            const double c0 = sh.trig[0], s0 = sh.trig[1], cs = sh.trig[2], sn = sh.trig[3];
            const GlobalPoint2D T = sh.T;
            const double off = (double)(lane - (nl - 1) / 2) * c.lattice_step;
            Bezier bz;
            bz.x0 = ego.x; bz.y0 = ego.y; bz.x3 = T.x + off * sn; bz.y3 = T.y + off * (-cs);
            const double dx = bz.x3 - bz.x0, dy = bz.y3 - bz.y0;
            const double d = sqrt(dx * dx + dy * dy) / 3;
            bz.x1 = bz.x0 + d * c0; bz.y1 = bz.y0 + d * s0;
            bz.x2 = bz.x3 - d * cs; bz.y2 = bz.y3 - d * sn;
            sh.bz[lane] = bz;
            X0 = fmin(fmin(bz.x0, bz.x1), fmin(bz.x2, bz.x3)); X1 = fmax(fmax(bz.x0, bz.x1), fmax(bz.x2, bz.x3));
            Y0 = fmin(fmin(bz.y0, bz.y1), fmin(bz.y2, bz.y3)); Y1 = fmax(fmax(bz.y0, bz.y1), fmax(bz.y2, bz.y3));
        }
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) {          // DMPP_MAX_LATTICE <= 32 candidates: lanes 0..31
            X0 = fmin(X0, shfl_xor_f64(X0, sft)); X1 = fmax(X1, shfl_xor_f64(X1, sft));
            Y0 = fmin(Y0, shfl_xor_f64(Y0, sft)); Y1 = fmax(Y1, shfl_xor_f64(Y1, sft));
        }
        if (lane == 0) { sh.bb[0] = X0; sh.bb[1] = X1; sh.bb[2] = Y0; sh.bb[3] = Y1; sh.n_rel = 0; }
    }
    __syncthreads();
    double X0 = sh.bb[0], X1 = sh.bb[1], Y0 = sh.bb[2], Y1 = sh.bb[3];
    if (have_path) {
        // the box of the path's own cells (not a box around the ego: an ego outside the grid is clamped to a border cell),
        // one cell wider - the resampled points lie on the segments between the cell centres
        int cx0 = 0x7fffffff, cx1 = -1, cy0 = 0x7fffffff, cy1 = -1;
        for (int w = 0; w < (NW < kBoxWaves ? NW : kBoxWaves); w++) {
            cx0 = min(cx0, sh.box[w][0]); cx1 = max(cx1, sh.box[w][1]); cy0 = min(cy0, sh.box[w][2]); cy1 = max(cy1, sh.box[w][3]);
        }
        X0 = fmin(X0, si.grid_origin.x + ((double)cx0 - 0.5) * c.cell); X1 = fmax(X1, si.grid_origin.x + ((double)cx1 + 1.5) * c.cell);
        Y0 = fmin(Y0, si.grid_origin.y + ((double)cy0 - 0.5) * c.cell); Y1 = fmax(Y1, si.grid_origin.y + ((double)cy1 + 1.5) * c.cell);
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
    // ---- bucket grid (uniform decision per scene) ----
    const double inv_bx = (X1 > X0) ? (double)kBucketN / (X1 - X0) : 0.0, inv_by = (Y1 > Y0) ? (double)kBucketN / (Y1 - Y0) : 0.0;
    auto bucket_x = [&](double v) { return clampi((int)floor((v - X0) * inv_bx), 0, kBucketN - 1); };      // monotone in v: the obstacle's range and the
    auto bucket_y = [&](double v) { return clampi((int)floor((v - Y0) * inv_by), 0, kBucketN - 1); };      // point's bucket come from the same expression
    // culled scenes: entries index the LDS list; a scene with more than kMaxRelObs obstacles near its candidates (not culled) is
    // bucketed straight from the snapshot in HBM (entries index it: <= 256 obstacles), so that its points read a few records
    // instead of all of them.  A bucket that overflows sends only ITS points to the plain loop.
    const bool bucketed = culled ? n_rel >= kBucketMinObs : m <= 256;
    if (bucketed) {
        for (int i = tid; i < kBucketN * kBucketN; i += kThreads) sh.bcnt[i] = 0;
        __syncthreads();
        for (int j = tid; j < n_rel; j += kThreads) {
            double ox, oy, r;
            if (culled) { ox = sh.rx[j]; oy = sh.ry[j]; r = sh.rr[j]; }
            else { const ObPoint o = gobs[j]; ox = o.x; oy = o.y; r = (double)o.radius; }
            // the cutoff, a hair wider: a point exactly at the cutoff (where the penalty is zero anyway) stays inside
            const double thr = (r + half_w + c.d_safe) * (1.0 + 1e-9) + 1e-9;
            if (!(ox >= X0 - thr && ox <= X1 + thr && oy >= Y0 - thr && oy <= Y1 + thr)) continue;       // (always true for the culled list)
            const int bx0 = bucket_x(ox - thr), bx1 = bucket_x(ox + thr), by0 = bucket_y(oy - thr), by1 = bucket_y(oy + thr);
            for (int by = by0; by <= by1; by++)
                for (int bx = bx0; bx <= bx1; bx++) {
                    const int b = by * kBucketN + bx;
                    const int q = atomicAdd(&sh.bcnt[b], 1);
                    if (q < kBucketCap) sh.bent[b][q] = (unsigned char)j;
                }
        }
        __syncthreads();
    }
#ifdef DMPP_DEBUG_SEARCH
    if (tid == 0) {
        atomicAdd(&g_dbg_score[0], 1); atomicAdd(&g_dbg_score[1], sh.n_rel); atomicAdd(&g_dbg_score[2], bucketed ? 1 : 0);
        { int ov = 0; if (bucketed) for (int i = 0; i < kBucketN * kBucketN; i++) ov += sh.bcnt[i] > kBucketCap; atomicAdd(&g_dbg_score[3], ov); }
        atomicAdd(&g_dbg_score[4], culled ? 0 : 1);
        atomicMax(&g_dbg_score[5], sh.n_rel);
    }
#endif
    GlobalPoint2D* cand = sh.cand[wave];
    // The Bezier parameter of a point, and with it the four basis weights, are the same for every lattice candidate: each
    // lane keeps those of its (<= 4) points (bezier_point's own expressions, so the points come out bit for bit the same).
    double wb0[4], wb1[4], wb2[4], wb3[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = lane + 64 * q;
        const double t = (double)i / (double)(DMPP_PATH_POINTS - 1), u = 1 - t;
        wb0[q] = u * u * u; wb1[q] = 3 * u * u * t; wb2[q] = 3 * u * t * t; wb3[q] = t * t * t;
    }
    const double kk2_fenced = 0.001 * 0.001;           // R fenced to 1000
    // penalty and squared curvature of point i (= p) of the candidate whose points lie in cnd[]
    auto point_terms = [&](const GlobalPoint2D* cnd, int i, const GlobalPoint2D p, double& pen, double& kk2, bool& hit) {
        double clear = __builtin_inf();
        int nb = 0, b = 0;
        if (bucketed) { b = bucket_y(p.y) * kBucketN + bucket_x(p.x); nb = sh.bcnt[b]; }
        if (bucketed && nb <= kBucketCap) {
            for (int e = 0; e < nb; e++) {
                const int j = sh.bent[b][e];
                double ox, oy, r, t2;
                if (culled) { ox = sh.rx[j]; oy = sh.ry[j]; r = sh.rr[j]; t2 = sh.rt2[j]; }
                else { ox = gobs[j].x; oy = gobs[j].y; r = (double)gobs[j].radius; const double thr = r + half_w + c.d_safe; t2 = thr * thr; }
                const double dx = p.x - ox, dy = p.y - oy;
                const double d2 = dx * dx + dy * dy;
                if (d2 > t2) continue;
                const double v = sqrt(d2) - r;
                if (v < clear) clear = v;
            }
        } else if (__builtin_expect(culled, 1)) {
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
        hit = false;
        if (clear <= 0) { pen = 1000.0; hit = true; }
        else if (clear < c.d_safe) { const double qq = (c.d_safe - clear) / c.d_safe; pen = qq * qq; }
        else pen = 0;
        kk2 = 0;
        if (i >= 1 && i <= DMPP_PATH_POINTS - 2) {
            // (1 / R)^2 of the circumradius of (P[i-1], P[i], P[i+1]), R = 0.5 * dis3 / sinA fenced to 1000 - from the
            // SQUARED side lengths, with one division and no square root:
            //   cosA = (d1 + d2 - d3) / (2 sqrt(d1 d2))   =>   sinA^2 = (4 d1 d2 - (d1 + d2 - d3)^2) / (4 d1 d2),
            //   (1 / R)^2 = sinA^2 / (0.25 d3) = (4 d1 d2 - (d1 + d2 - d3)^2) / (d1 d2 d3).
            // Against the oracle's sqrt / divide chain this differs by rounding only (<= ~1e-9 relative where sinA is at
            // its fence of 0.001, less elsewhere; the scores are compared to 1e-6) - except AT the fence, see below.
            const GlobalPoint2D pa = cnd[i - 1], pf = cnd[i + 1];
            const double d1 = (pa.x - p.x) * (pa.x - p.x) + (pa.y - p.y) * (pa.y - p.y);
            const double d2 = (p.x - pf.x) * (p.x - pf.x) + (p.y - pf.y) * (p.y - pf.y);
            const double d3 = (pa.x - pf.x) * (pa.x - pf.x) + (pa.y - pf.y) * (pa.y - pf.y);
            const double den2 = 4 * d1 * d2, num = d1 + d2 - d3;
            const double diff = den2 - num * num, fence = kk2_fenced * den2;        // sinA >= 0.001  <=>  diff >= fence
            kk2 = kk2_fenced;
            if (den2 > 0) {
                if (__builtin_expect(fabs(diff - fence) <= 1e-6 * fence, 0)) {
                    // The fence is a discontinuity of the specification (1 / R jumps from 0.001 to sinA / (0.5 dis3)), so the side a
                    // point falls on must be the oracle's own decision, not one taken from a differently rounded expression: where
                    // the squared form cannot tell (its rounding error is ~1e-9 of the fence for sides of similar length; this band is 1e-6) the
                    // oracle's chain of operations is evaluated as it stands.  About one point in 1e6 comes here
                    // (tests/test_oracle_properties.py replays both forms around the fence).
                    const double dis1 = sqrt(d1), dis2 = sqrt(d2), dis3 = sqrt(d3);
                    const double den = 2 * dis1 * dis2;
                    if (den > 0) {
                        const double cosA = (dis1 * dis1 + dis2 * dis2 - dis3 * dis3) / den;
                        const double sinA = sqrt(1 - cosA * cosA);
                        if (sinA >= 0.001) { const double R = 0.5 * dis3 / sinA; const double kq = 1 / R; kk2 = kq * kq; }
                    }
                } else if (diff > fence) kk2 = diff / (d1 * d2 * d3);
            }
        }
    };
    auto publish = [&](int k, double off, double pen_acc, double k2_acc, int first_hit) {      // the whole wave
        const double col = wave_tree_sum(pen_acc), curv = wave_tree_sum(k2_acc);
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) first_hit = min(first_hit, __shfl_xor(first_hit, sft, 64));
        const double prog = (first_hit == DMPP_PATH_POINTS) ? 0.0 : (double)(DMPP_PATH_POINTS - first_hit) / (double)DMPP_PATH_POINTS;
        const double cost = c.w_col * col + c.w_curv * curv + c.w_prog * prog + c.w_off * fabs(off);
        if (lane == 0) { go.cand_col[k] = col; go.cand_curv[k] = curv; go.cand_prog[k] = prog; go.cand_cost[k] = cost; sh.cost[k] = cost; }
    };
    // Whole rounds: one candidate per wave.  What is left over (17 candidates on 4 waves leave one) is split by quarter of the
    // points, one quarter per wave, when the waves suffice (4 * left <= NW); the per-lane partial sums are then added in the
    // order of the quarters, which is the order one wave would have added them in.
    const int left = nc % NW, n_whole = (4 * left <= NW) ? nc - left : nc;
    // One whole candidate: passes 0 .. n_pass - 1 over its points (64 per pass).  The fourth pass holds only the points 192 .. 199:
    // a wave that scores several candidates (four waves per scene) leaves it out and runs ONE packed pass for all of them at the
    // end - lane 8 c + j takes point 192 + j of its c-th candidate - and then adds each term to the lane that would have added it
    // (lane j, after its three earlier terms: the order of the sums does not change).
    // (The loops over a wave's candidates and over a candidate's passes are NOT unrolled: point_terms is ~1,000 instructions, and
    // eighteen copies of it made this kernel 129 KB of code; rolled it is 36 KB and exactly as fast, alone and beside the searches.)
    auto whole_candidate = [&](int k, int n_pass, double& off, double& pen_acc, double& k2_acc, int& first_hit, GlobalPoint2D* tail) {
        off = 0;
        if (k < nl) {
            const Bezier bz = sh.bz[k];
            off = (double)(k - (nl - 1) / 2) * c.lattice_step;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                GlobalPoint2D P;
                P.x = wb0[q] * bz.x0 + wb1[q] * bz.x1 + wb2[q] * bz.x2 + wb3[q] * bz.x3;
                P.y = wb0[q] * bz.y0 + wb1[q] * bz.y1 + wb2[q] * bz.y2 + wb3[q] * bz.y3;
                if (lane + 64 * q < DMPP_PATH_POINTS) cand[lane + 64 * q] = P;
            }
        } else {
#pragma nounroll
            for (int q = 0; q < 4; q++)
                if (lane + 64 * q < DMPP_PATH_POINTS) cand[lane + 64 * q] = mean_point(c, sh.pts, sh.cum, a + 1, lane + 64 * q, DMPP_PATH_POINTS);
        }
        wave_sync();
        if (tail && lane < DMPP_PATH_POINTS - kTailFirst) tail[lane] = cand[kTailFirst + lane];      // points 190 .. 199 for the packed pass
        pen_acc = 0; k2_acc = 0; first_hit = DMPP_PATH_POINTS;
#pragma nounroll
        for (int q = 0; q < n_pass; q++) {
            const int i = lane + 64 * q;
            if (i < DMPP_PATH_POINTS) {
                double pen, kk2; bool hit;
                point_terms(cand, i, cand[i], pen, kk2, hit);
                pen_acc += pen; k2_acc += kk2;
                if (hit && i < first_hit) first_hit = i;
            }
        }
        wave_sync();
    };
    const int rounds = (n_whole - wave + NW - 1) / NW;                 // whole candidates of this wave
    const bool pack_tail = kPackTail<NW> && (n_whole + NW - 1) / NW >= 2 && (n_whole + NW - 1) / NW <= kTailRounds;       // (uniform over the block)
    {
        double accP[kTailRounds], accK[kTailRounds], accOff[kTailRounds]; int accH[kTailRounds];
#pragma unroll
        for (int t = 0; t < kTailRounds; t++) { accP[t] = 0; accK[t] = 0; accOff[t] = 0; accH[t] = DMPP_PATH_POINTS; }
#pragma nounroll
        for (int ci = 0; ci < rounds; ci++) {
            const int k = wave + ci * NW;
            double off, pen_acc, k2_acc; int first_hit;
            whole_candidate(k, pack_tail ? 3 : 4, off, pen_acc, k2_acc, first_hit, pack_tail ? &sh.tail[wave % kTailWaves<NW>][ci][0] : nullptr);
            if (!pack_tail) publish(k, off, pen_acc, k2_acc, first_hit);
            else {
#pragma unroll
                for (int t = 0; t < kTailRounds; t++) if (t == ci) { accP[t] = pen_acc; accK[t] = k2_acc; accOff[t] = off; accH[t] = first_hit; }
            }
        }
        if (pack_tail) {
            {   // the packed pass: lane 8 c + j = point 192 + j of candidate c
                const int cl = lane >> 3, i = 192 + (lane & 7);
                double pen = 0, kk2 = 0; bool hit = false;
                if (cl < rounds) {
                    const GlobalPoint2D* cnd = &sh.tail[wave % kTailWaves<NW>][cl][0] - kTailFirst;      // cnd[i] = point i, i = 190 .. 199
                    point_terms(cnd, i, cnd[i], pen, kk2, hit);
                }
                const int hv = hit ? i : DMPP_PATH_POINTS;
#pragma unroll
                for (int ci = 0; ci < kTailRounds; ci++) {
                    const double vp = shfl_f64(pen, 8 * ci + (lane & 7)), vk = shfl_f64(kk2, 8 * ci + (lane & 7));
                    const int vh = __shfl(hv, 8 * ci + (lane & 7), 64);
                    if (ci < rounds && lane < DMPP_PATH_POINTS - 192) { accP[ci] += vp; accK[ci] += vk; accH[ci] = min(accH[ci], vh); }
                }
            }
#pragma unroll
            for (int ci = 0; ci < kTailRounds; ci++) if (ci < rounds) publish(wave + ci * NW, accOff[ci], accP[ci], accK[ci], accH[ci]);
        }
    }
    if (n_whole < nc) {
        struct Part { double pen[DMPP_WAVE], k2[DMPP_WAVE]; int hit[DMPP_WAVE]; };
        static_assert(sizeof(Part) * NW <= sizeof(GlobalPoint2D) * DMPP_PATH_POINTS * (NW - NW / 4), "the partial sums live in the candidate arrays the left-over candidates do not use");
        Part* parts = reinterpret_cast<Part*>(&sh.cand[NW / 4][0]);
        __syncthreads();                                   // every wave is done with its candidate array
        const bool mine = wave < 4 * left;
        const int j = wave >> 2, q = wave & 3, k = n_whole + j, i = lane + 64 * q;
        GlobalPoint2D p = { 0.0, 0.0 };
        if (mine) {
            if (k < nl) {
                const Bezier bz = sh.bz[k];
                const double b0 = q == 0 ? wb0[0] : q == 1 ? wb0[1] : q == 2 ? wb0[2] : wb0[3], b1 = q == 0 ? wb1[0] : q == 1 ? wb1[1] : q == 2 ? wb1[2] : wb1[3];
                const double b2 = q == 0 ? wb2[0] : q == 1 ? wb2[1] : q == 2 ? wb2[2] : wb2[3], b3 = q == 0 ? wb3[0] : q == 1 ? wb3[1] : q == 2 ? wb3[2] : wb3[3];
                p.x = b0 * bz.x0 + b1 * bz.x1 + b2 * bz.x2 + b3 * bz.x3;
                p.y = b0 * bz.y0 + b1 * bz.y1 + b2 * bz.y2 + b3 * bz.y3;
            } else p = mean_point(c, sh.pts, sh.cum, a + 1, min(i, DMPP_PATH_POINTS - 1), DMPP_PATH_POINTS);
            if (i < DMPP_PATH_POINTS) sh.cand[j][i] = p;
        }
        __syncthreads();
        if (mine) {
            double pen = 0, kk2 = 0; bool hit = false;
            if (i < DMPP_PATH_POINTS) point_terms(sh.cand[j], i, p, pen, kk2, hit);
            parts[wave].pen[lane] = pen; parts[wave].k2[lane] = kk2; parts[wave].hit[lane] = hit ? i : DMPP_PATH_POINTS;
        }
        __syncthreads();
        if (wave < left) {                                 // wave j2 = wave adds the quarters of left-over candidate j2 up, in order
            const int k2i = n_whole + wave;
            double pen_acc = 0, k2_acc = 0; int first_hit = DMPP_PATH_POINTS;
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                if (lane + 64 * qq < DMPP_PATH_POINTS) {
                    const Part& pt = parts[wave * 4 + qq];
                    pen_acc += pt.pen[lane]; k2_acc += pt.k2[lane]; first_hit = min(first_hit, pt.hit[lane]);
                }
            }
            const double off2 = k2i < nl ? (double)(k2i - (nl - 1) / 2) * c.lattice_step : 0.0;
            publish(k2i, off2, pen_acc, k2_acc, first_hit);
        }
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
                p = bezier_point(sh.bz[k], tid, DMPP_PATH_POINTS);
            } else if (nc > 0) p = mean_point(c, sh.pts, sh.cum, a + 1, tid, DMPP_PATH_POINTS);
            go.best_path[tid] = p;
        }
    }
}

}  // namespace dmpp
