// kernels_g.hpp — HIP kernels of the grid engine (specification in
// DESIGN.md §5 and oracle/dmpp_grid_oracle.c — the reference has no grid code).
//
//   k_rasterise : obstacle list -> bit-packed occupancy grid in HBM, row-major and column-major.  One workgroup
//                 per (scene, band of rows): footprints are OR-ed into two LDS bit bands (the same cell in both
//                 orientations), then the bands are written out word by word (HBM-write bound; 1/8 of a byte grid).
//   k_expand_grid: one scene's bitmap -> u8 grid, on demand (pp_get_grid).
//   k_order     : launch order of the search, heaviest scenes first (counting sort on the previous search times).
//   k_search    : jump-point A*.  Four waves set a scene up (bitmaps HBM -> LDS by LDS-DMA, word summaries, empty
//                 closed set), then ONE wave searches: a straight jump is one lane walking the candidate words of its
//                 line; a diagonal jump is a group of 16 lanes; the open list and the closed-set hash are LDS resident
//                 (DPP minimum, ballot + prefix-popcount compaction); up to 4 nodes of the minimal f per step.
//   k_score     : lattice candidates (cubic Beziers to laterally shifted terminals + the grid
//                 path) scored on collision / curvature / progress, 4 (or 16) waves per scene.
#pragma once
#include "dev_geom.hpp"

namespace dmpp {

__device__ __forceinline__ uint64_t mix64(uint64_t v)
{
    uint64_t z = v + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int cell_of(const PlannerConfig& c, GlobalPoint2D origin, double x, double y)
{
    int ix = (int)floor((x - origin.x) / c.cell);
    int iy = (int)floor((y - origin.y) / c.cell);
    ix = clampi(ix, 0, c.grid_w - 1); iy = clampi(iy, 0, c.grid_h - 1);
    return iy * c.grid_w + ix;
}

// ---------------------------------------------------------------------------------------
// G1.  grid: per scene H x W/32 words row-major, then W x H/32 words column-major; bit = 1 occupied (the host
// guarantees W % 32 == 0 and H % 32 == 0, and bands of a multiple of 32 rows: whole words in both orientations).
constexpr int kRasterBlock = 256;

__global__ void __launch_bounds__(kRasterBlock)
k_rasterise(PlannerConfig c, int n_scenes, int band_rows, const SceneIn* __restrict__ in,
            const ObPoint* __restrict__ obs_now, uint32_t* __restrict__ gbits)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int scene = blockIdx.x, band = blockIdx.y;
    if (scene >= n_scenes) return;
    const int W = c.grid_w, H = c.grid_h, WW = W >> 5, HW = H >> 5;
    const int row0 = band * band_rows;                               // band_rows is a multiple of 32
    const int rows = min(band_rows, H - row0);
    if (rows <= 0) return;
    const int words = (rows * W) >> 5;
    const int bw = rows >> 5;                                        // words per column inside this band
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem_raw);          // row-major band: rows x WW words
    uint32_t* bitsT = bits + ((band_rows * W) >> 5);                 // column-major band: W x bw words
    const int tid = threadIdx.x;
    for (int w = tid; w < words; w += kRasterBlock) { bits[w] = 0; bitsT[w] = 0; }
    __syncthreads();
    const SceneIn& si = in[scene];
    const double ox = si.grid_origin.x, oy = si.grid_origin.y;
    const int m = si.obs_n;
    const ObPoint* obs = obs_now + si.obs_off;
    // Footprints that touch this band: one thread per obstacle computes the clipped bounding box once,
    // the hits are compacted into an LDS list (chunks of 256 obstacles); then, per listed footprint, the
    // cells of its box are spread over the 256 threads.
    __shared__ int s_box[kRasterBlock][4];
    __shared__ double s_par[kRasterBlock][3];      // x, y, R^2
    __shared__ int s_cnt;
    for (int j0 = 0; j0 < m; j0 += kRasterBlock) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const int j = j0 + tid;
        if (j < m) {
            const ObPoint o = obs[j];
            const double R = (double)o.radius + c.inflate;
            int ix0 = (int)floor((o.x - R - ox) / c.cell) - 1, ix1 = (int)floor((o.x + R - ox) / c.cell) + 1;
            int iy0 = (int)floor((o.y - R - oy) / c.cell) - 1, iy1 = (int)floor((o.y + R - oy) / c.cell) + 1;
            ix0 = max(ix0, 0); ix1 = min(ix1, W - 1);
            iy0 = max(iy0, row0); iy1 = min(iy1, row0 + rows - 1);
            if (ix1 >= ix0 && iy1 >= iy0) {
                const int k = atomicAdd(&s_cnt, 1);        // order does not matter: the bits are OR-ed
                s_box[k][0] = ix0; s_box[k][1] = ix1; s_box[k][2] = iy0; s_box[k][3] = iy1;
                s_par[k][0] = o.x; s_par[k][1] = o.y; s_par[k][2] = R * R;
            }
        }
        __syncthreads();
        const int nhit = s_cnt;
        for (int k = 0; k < nhit; k++) {
            const int ix0 = s_box[k][0], iy0 = s_box[k][2];
            const int bwid = s_box[k][1] - ix0 + 1, bh = s_box[k][3] - iy0 + 1;
            const double cxo = s_par[k][0], cyo = s_par[k][1], R2 = s_par[k][2];
            for (int t = tid; t < bwid * bh; t += kRasterBlock) {
                const int iy = iy0 + t / bwid, ix = ix0 + t % bwid;
                const double cx = ox + ((double)ix + 0.5) * c.cell, cy = oy + ((double)iy + 0.5) * c.cell;
                const double dx = cx - cxo, dy = cy - cyo;
                if (dx * dx + dy * dy <= R2) {
                    const int r = iy - row0, b = r * W + ix;
                    atomicOr(&bits[b >> 5], 1u << (b & 31));
                    atomicOr(&bitsT[ix * bw + (r >> 5)], 1u << (r & 31));     // the same cell in the column-major copy
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // the occupancy grid, bit-packed: row-major (H x WW words) then column-major (W x HW words)
    uint32_t* brow = gbits + (size_t)scene * 2 * ((size_t)(W * H) >> 5);
    uint32_t* bcol = brow + ((size_t)(W * H) >> 5);
    for (int w = tid; w < words; w += kRasterBlock) brow[row0 * WW + w] = bits[w];
    for (int w = tid; w < words; w += kRasterBlock) bcol[(w / bw) * HW + (row0 >> 5) + (w % bw)] = bitsT[w];
}

// The occupancy grid of one scene as bytes (0 free / 1 occupied), for pp_get_grid: 16 bits -> 16 bytes per thread.
__global__ void __launch_bounds__(kRasterBlock)
k_expand_grid(int grid_w, int grid_h, const uint32_t* __restrict__ brow, uint8_t* __restrict__ out)
{
    const int chunks = (grid_w * grid_h) >> 4;
    uint4* o4 = reinterpret_cast<uint4*>(out);
    for (int k = blockIdx.x * kRasterBlock + threadIdx.x; k < chunks; k += gridDim.x * kRasterBlock) {
        const uint32_t w = brow[k >> 1];
        const uint32_t h16 = (k & 1) ? (w >> 16) : (w & 0xFFFFu);
        uint4 v;
        v.x = ((h16 & 0xFu) * 0x00204081u) & 0x01010101u;
        v.y = (((h16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
        v.z = (((h16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
        v.w = (((h16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
        o4[k] = v;
    }
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int hfun(int x, int y, int gx, int gy)
{
    int dx = abs(x - gx), dy = abs(y - gy);
    return 10 * max(dx, dy) + 4 * min(dx, dy);
}

// G2: jump-point A* (specification: oracle/dmpp_grid_oracle.c).  ONE searching wave per scene.
//   * the bit-packed grid of k_rasterise arrives in LDS by LDS-DMA, both orientations: E/W and N/S jumps are both scans
//     along a line of words; per line a summary word says which of its words are non-zero;
//   * a straight jump is ONE lane (jump_lane): it visits, in travel order, only the words where the line or one of its
//     two neighbours has an obstacle bit (or the goal) and builds  blocked | forced | goal  with two shifts;
//   * a diagonal jump (<= DMPP_DIAG_JUMP cells) is a group of 16 lanes: lane pair k scans horizontally | vertically from
//     cell k+1 of the diagonal; the pair of cell 8 is free for the straight successors, so one round serves a step;
//   * a step takes up to 4 open entries of the minimal f (their g is final, so the search stays optimal), closes them
//     and expands them on 4 x 8 lanes (node x direction);
//   * the open list lives in LDS in push order (f/2, x|y|dir, run): minimum by DPP wave reduction, ties
//     picked with ballots, dead slots squeezed out with ballot + prefix-popcount compaction;
//   * closed cells: an LDS hash (atomicCAS insertion, direction + run length beside the cell) answers while it has room;
//     a scene that outgrows it moves to a bit set + direction/run array in HBM (spill), zeroed only then.
constexpr int kOpenCap = DMPP_OPEN_CAP;
constexpr int kClosedLog = 10, kClosedTab = 1 << kClosedLog, kClosedMax = 768;      // LDS closed-set hash; beyond kClosedMax the HBM bit set answers
constexpr int kDiagK = DMPP_DIAG_JUMP;                  // cells a diagonal jump looks ahead
constexpr int kDiagGroup = 2 * kDiagK;                  // lanes per diagonal jump: (cell, horizontal | vertical component)
constexpr int kDiagPerRound = DMPP_WAVE / kDiagGroup;
constexpr int kMaxDiag = 16;                            // diagonal jumps of a step: <= 4 nodes x 4 (start) / x 3

// minimum over the 64 lanes, returned in every lane: DPP prefix-min inside each row of 16 lanes
// (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row results to lane 63.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#define DMPP_DPP_MIN(ctrl, rowmask)                                                                              \
    { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, ctrl, rowmask, 0xf, false); \
      v = t < v ? t : v; }
    DMPP_DPP_MIN(0x111, 0xf) DMPP_DPP_MIN(0x112, 0xf) DMPP_DPP_MIN(0x114, 0xf) DMPP_DPP_MIN(0x118, 0xf)
    DMPP_DPP_MIN(0x142, 0xa) DMPP_DPP_MIN(0x143, 0xc)
#undef DMPP_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <bool GBM>
struct Bits {
    const uint32_t* bm; int W, H, WW;

    __device__ __forceinline__ bool blk(int x, int y) const
    {
        if (x < 0 || y < 0 || x >= W || y >= H) return true;
        return (bm[y * WW + (x >> 5)] >> (x & 31)) & 1u;
    }
};

// One straight jump = one lane.  A "view" is a bit matrix of NL lines x LW words: the row-major bitmap for
// E/W travel (line = y, position along the line = x) or the transposed one for N/S (line = x, position = y);
// the forced-neighbour test only needs the two neighbouring lines, so both axes share the code.  `nz` holds
// one bit per word of the view (word != 0), SW summary words per line: after the word the jump starts in, the
// scan goes straight to the next word where the line or one of its two neighbours has any obstacle bit (or
// to the goal's word) instead of walking the free words in between.
struct View {
    const uint32_t* base; const uint32_t* nz;
    int LW, NL, SW;
};
__device__ __forceinline__ uint32_t view_word(const View& V, int line, int w)
{
    const bool ok = (unsigned)line < (unsigned)V.NL && (unsigned)w < (unsigned)V.LW;
    const uint32_t v = V.base[ok ? line * V.LW + w : 0];
    return ok ? v : 0xFFFFFFFFu;
}
__device__ __forceinline__ uint32_t view_nz(const View& V, int line, int sw)
{
    const bool ok = (unsigned)line < (unsigned)V.NL;
    const uint32_t v = V.nz[ok ? line * V.SW + sw : 0];
    return ok ? v : 0u;
}
// run = cells travelled from `pos` along `line` in direction sgn to the first stop (blocked | forced | goal);
// 0 = none (the first stop is a wall or the edge of the grid).  Safe for starts outside the grid (returns 0).
// Only words whose summary bit is set in the line or one of its two neighbours (or that hold the goal) can
// contain a stop: those candidate words are visited in travel order, nothing else is read.  LW <= 64.
__device__ __forceinline__ int jump_lane(const View& V, bool active, int line, int pos, int sgn, int gline, int gpos, int* iters = nullptr)
{
    int run = 0;
    bool go = active && (unsigned)line < (unsigned)V.NL && (unsigned)pos < (unsigned)(V.LW << 5);
    const int w0 = pos >> 5;
    unsigned long long cand = 0;
    if (go) {
        uint32_t lo = view_nz(V, line, 0) | view_nz(V, line + 1, 0) | view_nz(V, line - 1, 0), hi = 0;
        if (V.SW > 1) hi = view_nz(V, line, 1) | view_nz(V, line + 1, 1) | view_nz(V, line - 1, 1);
        cand = ((unsigned long long)hi << 32) | lo;
        if (gline == line) cand |= 1ull << (gpos >> 5);
        cand &= sgn > 0 ? ~((1ull << w0) - 1ull) : ((2ull << w0) - 1ull);     // the start word and everything ahead of it
        go = cand != 0;
    }
    for (int it = 0; it <= V.LW; it++) {
        if (!__ballot(go)) break;
        if (iters) ++*iters;
        if (go) {
            const int wi = sgn > 0 ? __ffsll((long long)cand) - 1 : 63 - __clzll((long long)cand);
            cand &= ~(1ull << wi);
            const int nwi = wi + sgn;
            const uint32_t B0 = view_word(V, line, wi);
            const uint32_t P = view_word(V, line + 1, wi), M = view_word(V, line - 1, wi);
            const uint32_t Pw = view_word(V, line + 1, nwi), Mw = view_word(V, line - 1, nwi);
            uint32_t Pn, Mn;
            if (sgn > 0) { Pn = (P >> 1) | (Pw << 31); Mn = (M >> 1) | (Mw << 31); }
            else         { Pn = (P << 1) | (Pw >> 31); Mn = (M << 1) | (Mw >> 31); }
            uint32_t stop = B0 | (P & ~Pn) | (M & ~Mn);
            if (gline == line && (gpos >> 5) == wi) stop |= 1u << (gpos & 31);
            if (wi == w0) {                                  // only the cells strictly ahead of the start
                const int bp = pos & 31;
                if (sgn > 0) stop &= (bp == 31) ? 0u : ~((2u << bp) - 1u);
                else         stop &= (1u << bp) - 1u;
            }
            if (stop) {
                const int bit = sgn > 0 ? (__ffs((int)stop) - 1) : (31 - __clz((int)stop));
                if (!((B0 >> bit) & 1u)) { const int np = (wi << 5) + bit; run = sgn > 0 ? np - pos : pos - np; }
                go = false;
            } else if (cand == 0) go = false;                // free all the way to the edge of the grid: no jump point
        }
    }
    return run;
}

// Launch order of the scenes of k_search: heaviest first, by the time the scene's search took on the previous tick (in
// units of 8 Ki cycles, written by k_search; it changes little from tick to tick).  One wave per scene and two waves per CU means the kernel ends with its
// slowest scene; starting that scene first keeps it off the tail.  Counting sort into 1024 cost classes, one block.
constexpr int kOrderBlock = 1024, kOrderClasses = 1024, kOrderShift = 13;     // classes of 8 Ki cycles, up to 8.4 M cycles
__global__ void __launch_bounds__(kOrderBlock)
k_order(int n_scenes, const int32_t* __restrict__ cost, int32_t* __restrict__ perm)
{
    __shared__ int cnt[kOrderClasses], base[kOrderClasses];
    const int tid = threadIdx.x;
    cnt[tid] = 0;
    __syncthreads();
    for (int s = tid; s < n_scenes; s += kOrderBlock) atomicAdd(&cnt[kOrderClasses - 1 - min(max(cost[s], 0), kOrderClasses - 1)], 1);
    __syncthreads();
    // exclusive prefix sum over the classes (heaviest class first): Hillis-Steele in LDS, one class per thread
    base[tid] = cnt[tid];
    __syncthreads();
    for (int d = 1; d < kOrderClasses; d <<= 1) {
        const int v = tid >= d ? base[tid - d] : 0;
        __syncthreads();
        base[tid] += v;
        __syncthreads();
    }
    base[tid] -= cnt[tid];
    __syncthreads();
    for (int s = tid; s < n_scenes; s += kOrderBlock)
        perm[atomicAdd(&base[kOrderClasses - 1 - min(max(cost[s], 0), kOrderClasses - 1)], 1)] = s;
}

// Four waves set a scene up (bitmaps into LDS, word summaries, empty closed set); then waves 1..3 leave and wave 0
// searches alone - the search itself is one serial chain of steps.
constexpr int kSearchSetupWaves = 4, kSearchBlock = kSearchSetupWaves * DMPP_WAVE;
template <bool GBM>
__global__ void __launch_bounds__(kSearchBlock)
k_search(PlannerConfig c, int n_scenes, int order_cap, const int32_t* __restrict__ perm, const SceneIn* __restrict__ in,
         uint32_t* __restrict__ gclosed, uint16_t* __restrict__ pinfo, int32_t* __restrict__ orders,
         int32_t* __restrict__ paths, GridOut* __restrict__ gout, uint32_t* __restrict__ gbitmaps, int32_t* __restrict__ cost_out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ uint32_t o_ent[kOpenCap];       // x | y << 12 | arriving direction << 24
    __shared__ uint16_t o_f2[kOpenCap];        // f / 2, 0xFFFF = dead slot
    __shared__ uint16_t o_run[kOpenCap];       // run length of the move that reached the cell
    __shared__ int dc_owner[kMaxDiag];                                         // the (node, s) lanes of the diagonal jumps of the current step
    __shared__ uint32_t sj_job[kDiagGroup];                                    // its straight jumps (<= 8), packed
    __shared__ uint32_t c_tab[kClosedTab];     // closed cells (cell + 1, 0 = empty): open addressing, linear probing
    __shared__ uint16_t c_info[kClosedTab];    // arriving direction | run length << 4 of the cell in the same slot
    if ((int)blockIdx.x >= n_scenes) return;
    const long long t_begin = clock64();
    __builtin_amdgcn_s_setprio(3);             // one latency-bound wave per scene: issue ahead of the kernels that run beside it
    const int scene = perm ? perm[blockIdx.x] : (int)blockIdx.x;      // heaviest scenes first (k_order) when they do not all fit at once
#ifdef DMPP_DEBUG_SEARCH
    const long long t_entry = clock64(); long long t_loop = t_entry;
    const long long w_entry = wall_clock64();          // 100 MHz, the same counter on every CU: launch timeline of the scenes
#endif
    const int tid = threadIdx.x, lane = tid & (DMPP_WAVE - 1), wv = tid >> 6;
    const int W = c.grid_w, H = c.grid_h, N = W * H, WW = W >> 5;
    const int HW = H >> 5;
    // dynamic LDS: the word summaries of both views (one bit per bitmap word), then - when they fit - the bitmaps
    const int SWr = (WW + 31) >> 5, SWc = (HW + 31) >> 5;
    uint32_t* nz_row = reinterpret_cast<uint32_t*>(smem_raw);
    uint32_t* nz_col = nz_row + H * SWr;
    // obstacle bits twice: row-major for E/W scans, column-major for N/S scans (N/32 words each)
    uint32_t* bm = GBM ? gbitmaps + (size_t)scene * 2 * (N >> 5) : nz_col + W * SWc;
    uint32_t* bmT = bm + (N >> 5);
    const SceneIn& si = in[scene];
    GridOut& go = gout[scene];
    uint32_t* closed = gclosed + (size_t)scene * (N >> 5);
    {   // this scene's closed bit set starts empty (16 bytes per lane; done with before the first atomicOr, see the wait below)
        uint4* c4 = reinterpret_cast<uint4*>(closed);
        const uint4 z = { 0u, 0u, 0u, 0u };
        for (int i = tid; i < (N >> 7); i += kSearchBlock) c4[i] = z;
    }
    uint16_t* pin = pinfo + (size_t)scene * N;
    int32_t* order = orders ? orders + (size_t)scene * order_cap : nullptr;
    int32_t* path = paths + (size_t)scene * c.max_path;

    // ---- the bit-packed occupancy grid (both orientations, written by k_rasterise) -> LDS, 16 B per lane ----
    if (!GBM) {
        const uint4* src4 = reinterpret_cast<const uint4*>(gbitmaps + (size_t)scene * 2 * (N >> 5));
        uint4* dst4 = reinterpret_cast<uint4*>(bm);
        const int chunks = (2 * (N >> 5)) >> 2;
        // LDS-DMA (global_load_lds_dwordx4): 1 KiB per wave instruction straight into LDS (destination = uniform base +
        // lane * 16), no VGPR staging, so every piece of the 2 * N / 8 bytes is in flight at once
        const int full = chunks & ~(DMPP_WAVE - 1);
        for (int c0 = wv * DMPP_WAVE; c0 < full; c0 += kSearchBlock)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src4 + c0 + lane),
                                             (__attribute__((address_space(3))) void*)(dst4 + c0), 16, 0, 0);
        if (wv == 0 && full + lane < chunks) dst4[full + lane] = src4[full + lane];
    }
    __builtin_amdgcn_s_waitcnt(0);             // the bitmaps have landed in LDS and the zeroes of the closed bit set in L2
    __syncthreads();
#ifdef DMPP_DEBUG_SEARCH
    const long long t_pack = clock64(); long long t_tr = t_pack, t_nz = t_pack;
#endif

    const int start = cell_of(c, si.grid_origin, si.loc.globalpoint.x, si.loc.globalpoint.y);
    const int goal = cell_of(c, si.grid_origin, si.goal.x, si.goal.y);
    const int gx = goal % W, gy = goal / W;
    const int cap = min(c.bucket_cap, kOpenCap);
    int status = -1, n_exp = 0, n_push = 0, n_rounds = 0, path_cost = 0;
    bool hash_complete = true;                 // every closed cell is in the LDS hash (with its direction and run)
    uint32_t start_was_set = 0;                // lane 0: the start cell was occupied in the grid (restored in HBM at the end)
    uint64_t digest = 0;
#ifdef DMPP_DEBUG_SEARCH
    long long t0 = clock64(), t_pop = 0, t_closed = 0, t_cand = 0, t_jump = 0, t_push = 0, t_done = 0; int c_iter = 0, c_jobs = 0, c_pass = 0, c_scan = 0, c_nt = 0;
#endif

    const bool goal_blocked = ((bm[goal >> 5] >> (goal & 31)) & 1u) != 0;      // the same in every wave
    if (!goal_blocked) {
        if (tid == 0) {                                                     // the vehicle is where it is
            const int sx = start % W, sy = start / W;
            start_was_set = (bm[start >> 5] >> (start & 31)) & 1u;
            bm[start >> 5] &= ~(1u << (start & 31));
            bmT[sx * HW + (sy >> 5)] &= ~(1u << (sy & 31));
        }
        if (GBM) __threadfence();
        __syncthreads();
#ifdef DMPP_DEBUG_SEARCH
        t_tr = clock64();
#endif
        // word summaries of both views (bit w of a line's summary = word w of the line is non-zero): one lane per line,
        // 16-byte reads when a line is a whole number of them
        for (int v = 0; v < 2; v++) {
            const uint32_t* src = v ? bmT : bm;
            uint32_t* nz = v ? nz_col : nz_row;
            const int LW = v ? HW : WW, NL = v ? W : H, SW = v ? SWc : SWr;
            for (int line = tid; line < NL; line += kSearchBlock) {
                uint32_t lo = 0, hi = 0;
                if ((LW & 3) == 0) {
                    const uint4* p4 = reinterpret_cast<const uint4*>(src + line * LW);
                    for (int q = 0; q < (LW >> 2); q++) {
                        const uint4 a = p4[q];
                        const uint32_t nib = min(a.x, 1u) | (min(a.y, 1u) << 1) | (min(a.z, 1u) << 2) | (min(a.w, 1u) << 3);
                        if (q < 8) lo |= nib << (4 * q); else hi |= nib << (4 * (q - 8));
                    }
                } else {
                    for (int w = 0; w < LW; w++) {
                        const uint32_t bit = min(src[line * LW + w], 1u);
                        if (w < 32) lo |= bit << w; else hi |= bit << (w - 32);
                    }
                }
                nz[line * SW] = lo;
                if (SW > 1) nz[line * SW + 1] = hi;
            }
        }
#ifdef DMPP_DEBUG_SEARCH
        t_nz = clock64();
#endif
        for (int i = tid; i < kClosedTab; i += kSearchBlock) c_tab[i] = 0;
        if (tid == 0) {
            o_ent[0] = (uint32_t)(start % W) | ((uint32_t)(start / W) << 12) | (8u << 24);
            o_f2[0] = (uint16_t)(hfun(start % W, start / W, gx, gy) >> 1);
            o_run[0] = 0;
        }
    }
    __syncthreads();
    if (wv != 0) return;                       // set-up done: the search is wave 0's
    if (goal_blocked) {
        status = DMPP_G_GOAL_BLOCKED;
    } else {
        Bits<GBM> B{ bm, W, H, WW };     // single-cell tests of the diagonal steps
        int n_open = 1, live = 1, fmax = -1;
        n_push = 1;
        const View Vrow{ bm, nz_row, WW, H, SWr }, Vcol{ bmT, nz_col, HW, W, SWc };
        // lane = node * 8 + s: the eight directions of each of the (<= 4) nodes of a step
        const int s = lane & 7;
        const int sdx = (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0);
        const int sdy = (s >= 1 && s <= 3) ? 1 : ((s >= 5) ? -1 : 0);
        long long guard = 16ll * N + 1024;                 // every iteration pops an entry; entries <= 8 per closed cell
#ifdef DMPP_DEBUG_SEARCH
        t_loop = clock64();
#endif
        while (status < 0) {
            if (--guard < 0) { status = DMPP_G_INTERNAL; break; }
#ifdef DMPP_DEBUG_SEARCH
            c_iter++; long long ta = clock64();
#endif
            if (live == 0) { status = DMPP_G_NO_PATH; break; }
            // ---- pop: up to 4 entries of the smallest f, the latest pushes first ----
            // (1) squeeze the dead slots out when they outnumber the live ones: the scans below stay short
            if (n_open - live > 64 && n_open > 2 * live) {
                int w = 0;
                for (int q0 = 0; q0 < n_open; q0 += DMPP_WAVE) {
                    const int i = q0 + lane;
                    uint32_t f2 = 0xFFFFu, ee = 0; uint16_t rr = 0;
                    if (i < n_open) { f2 = o_f2[i]; ee = o_ent[i]; rr = o_run[i]; }
                    const bool alive = f2 != 0xFFFFu;
                    const unsigned long long am = __ballot(alive);
                    wave_order();
                    if (alive) {
                        const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
                        o_f2[w + r] = (uint16_t)f2; o_ent[w + r] = ee; o_run[w + r] = rr;
                    }
                    w += __popcll(am);
                    wave_order();
                }
                n_open = w;
            }
            // (2) the first 256 slots are cached in registers: one LDS pass serves both the minimum and the ties
            uint32_t v0 = 0xFFFFu, v1 = 0xFFFFu, v2 = 0xFFFFu, v3 = 0xFFFFu;
            if (lane < n_open) v0 = o_f2[lane];
            if (lane + 64 < n_open) v1 = o_f2[lane + 64];
            if (lane + 128 < n_open) v2 = o_f2[lane + 128];
            if (lane + 192 < n_open) v3 = o_f2[lane + 192];
            uint32_t key = min(min(v0, v1), min(v2, v3));
            for (int i = lane + 256; i < n_open; i += DMPP_WAVE) { const uint32_t f2 = o_f2[i]; if (f2 < key) key = f2; }
            const uint32_t fmin2 = wave_min_u32(key);
            if (fmin2 == 0xFFFFu) { status = DMPP_G_INTERNAL; break; }
            const int f = (int)fmin2 << 1;
            int nt = 0, i0 = 0, i1 = 0, i2 = 0, i3 = 0;
            for (int c0 = ((n_open - 1) >> 6) << 6; c0 >= 0 && nt < DMPP_JPS_BATCH; c0 -= DMPP_WAVE) {
                const int i = c0 + lane;
                bool tie;
                if (c0 == 0) tie = v0 == fmin2; else if (c0 == 64) tie = v1 == fmin2; else if (c0 == 128) tie = v2 == fmin2;
                else if (c0 == 192) tie = v3 == fmin2; else tie = i < n_open && o_f2[i] == fmin2;
                unsigned long long tm = __ballot(tie);
                while (tm && nt < DMPP_JPS_BATCH) {
                    const int L = 63 - __clzll((long long)tm);
                    tm &= ~(1ull << L);
                    const int idx = c0 + L;
                    if (nt == 0) i0 = idx; else if (nt == 1) i1 = idx; else if (nt == 2) i2 = idx; else i3 = idx;
                    nt++;
                }
            }
            const int myi = lane == 0 ? i0 : lane == 1 ? i1 : lane == 2 ? i2 : i3;
            const bool have = lane < nt;
            uint32_t e = 0; int run_in = 0;
            if (have) { e = o_ent[myi]; run_in = o_run[myi]; }
            wave_order();
            if (have) o_f2[myi] = 0xFFFFu;
            live -= nt;
            if (i0 == n_open - 1) n_open--;
            const int x = (int)(e & 0xFFFu), y = (int)((e >> 12) & 0xFFFu), d = (int)(e >> 24);
            const int cell = y * W + x;
#ifdef DMPP_DEBUG_SEARCH
            long long tb = clock64(); t_pop += tb - ta; c_nt += nt;
#endif
            // ---- closed?  duplicates inside the batch: the earlier one wins; then the closed set ----
            bool valid = have;
            {
                const int c0_ = __builtin_amdgcn_readlane(cell, 0), c1_ = __builtin_amdgcn_readlane(cell, 1), c2_ = __builtin_amdgcn_readlane(cell, 2);
                if ((lane == 1 && cell == c0_) || (lane == 2 && (cell == c0_ || cell == c1_)) ||
                    (lane == 3 && (cell == c0_ || cell == c1_ || cell == c2_))) valid = false;
            }
            if (n_exp + DMPP_JPS_BATCH <= kClosedMax) {
                if (valid) {
                    atomicOr(&closed[cell >> 5], 1u << (cell & 31));          // the HBM set stays complete (fire and forget)
                    const uint32_t keyc = (uint32_t)cell + 1u;
                    uint32_t hh = ((uint32_t)cell * 2654435761u) >> (32 - kClosedLog);
                    for (int probe = 0; probe < kClosedTab; probe++) {
                        const uint32_t old = atomicCAS(&c_tab[hh], 0u, keyc);
                        if (old == 0u) { c_info[hh] = (uint16_t)(d | (run_in << 4)); break; }   // inserted: was open
                        if (old == keyc) { valid = false; break; }            // already closed
                        hh = (hh + 1) & (kClosedTab - 1);
                    }
                }
            } else {
                hash_complete = false;
                if (valid) {
                    const uint32_t old = atomicOr(&closed[cell >> 5], 1u << (cell & 31));
                    if ((old >> (cell & 31)) & 1u) valid = false;
                }
            }
            // the goal, or the entry that reaches the expansion limit, ends the search at once
            unsigned vm = (unsigned)__ballot(valid) & 0xFu;
            {
                const int nvb = __popc(vm & ((1u << lane) - 1u));
                const unsigned stop = (unsigned)__ballot(valid && (cell == goal || n_exp + nvb + 1 >= c.max_expansions)) & 0xFu;
                if (stop) {
                    const int last = __ffs((int)stop) - 1;
                    if (lane > last) valid = false;
                    vm = (unsigned)__ballot(valid) & 0xFu;
                }
            }
            if (valid) {
                const int seq = n_exp + __popc(vm & ((1u << lane) - 1u));
                pin[cell] = (uint16_t)(d | (run_in << 4));
                if (order && seq < order_cap) order[seq] = cell;
                digest += mix64(((uint64_t)(uint32_t)seq << 32) | (uint32_t)cell);
            }
            if (vm && f > fmax) { fmax = f; n_rounds++; }
            n_exp += __popc(vm);
            if (__ballot(valid && cell == goal)) { status = DMPP_G_FOUND; path_cost = f; break; }
            if (n_exp >= c.max_expansions) { status = DMPP_G_LIMIT; break; }
            if (vm == 0) continue;
#ifdef DMPP_DEBUG_SEARCH
            long long tc = clock64(); t_closed += tc - tb;
#endif
            // ---- successors: lane = node * 8 + s for the (<= 4) batch nodes ----
            const int node = lane >> 3;
            const int nx0 = __shfl(x, node, 64), ny0 = __shfl(y, node, 64), nd = __shfl(d, node, 64);
            const bool nvalid = lane < 32 && ((vm >> node) & 1u);
            const int gcur = f - hfun(nx0, ny0, gx, gy);
            bool want_jump = false, want_diag = false; int run = 0;
            {
                const int dd = nd & 7;
                const int ddx = (dd == 0 || dd == 1 || dd == 7) ? 1 : ((dd >= 3 && dd <= 5) ? -1 : 0);
                const int ddy = (dd >= 1 && dd <= 3) ? 1 : ((dd >= 5) ? -1 : 0);
                const int rel = (s - nd) & 7;
                const bool is_start = nd == 8, d_odd = (nd & 1) != 0 && !is_start, d_even = !d_odd && !is_start;
                want_jump = nvalid && ((is_start && (s & 1) == 0) || (d_even && rel == 0) || (d_odd && (rel == 1 || rel == 7)));
                const bool plain = nvalid && ((is_start && (s & 1) != 0) || (d_odd && rel == 0));
                const bool sided = nvalid && ((d_even && (rel == 1 || rel == 7)) || (d_odd && (rel == 2 || rel == 6)));
                const int px = d_odd ? (sdx - ddx) / 2 : sdx - ddx, py = d_odd ? (sdy - ddy) / 2 : sdy - ddy;
                const bool t_free = (plain || sided) && !B.blk(nx0 + sdx, ny0 + sdy);
                const bool side_blk = sided && B.blk(nx0 + px, ny0 + py);
                want_diag = (plain && t_free) || (sided && side_blk && t_free);
            }
#ifdef DMPP_DEBUG_SEARCH
            long long td = clock64(); t_cand += td - tc;
#endif
            // ---- jumps.  Every straight scan is one lane (jump_lane).  A diagonal jump takes a group of kDiagGroup = 16
            //      lanes: lane pair k = 0..6 scans horizontally | vertically from cell k+1 of the diagonal and its even lane
            //      tests that cell (blocked / goal / forced); cell 8 only needs the test (whatever a scan found there, the
            //      jump ends at that cell), so the last pair of every group is free for the straight successors of the
            //      batch (<= 8; 4 for the start node).  Four diagonal jumps + all straight ones per round; a second round
            //      only when a step has more than four diagonal successors.  The first cell with a finding ends a jump. ----
            const unsigned smask = (unsigned)__ballot(want_jump), dmask = (unsigned)__ballot(want_diag);
            const int n_sj = __popc(smask), n_dc = __popc(dmask);
            const int my_sj = __popc(smask & ((1u << (lane & 31)) - 1u)), my_dc = __popc(dmask & ((1u << (lane & 31)) - 1u));
            if (want_jump) {
                const bool horiz = s == 0 || s == 4;
                // view | sgn | line | pos, 12 bits each for line and pos (always inside the grid here)
                sj_job[my_sj] = (horiz ? 0u : 1u) | ((s == 0 || s == 2) ? 2u : 0u) | ((uint32_t)(horiz ? ny0 : nx0) << 2) | ((uint32_t)(horiz ? nx0 : ny0) << 14);
            }
            if (want_diag) dc_owner[my_dc] = lane;
            wave_order();
            const int n_rounds_j = (n_sj | n_dc) ? max(1, (n_dc + kDiagPerRound - 1) / kDiagPerRound) : 0;
            for (int rnd = 0; rnd < n_rounds_j; rnd++) {
                const int grp = lane / kDiagGroup, t = lane % kDiagGroup, kk = t >> 1;
                const bool last = kk == kDiagK - 1;                                // the pair of cell 8 = the straight-jump lanes
                const int dci = rnd * kDiagPerRound + grp;
                const int sji = grp * 2 + (t & 1);
                const bool dact = dci < n_dc, sact = rnd == 0 && last && sji < n_sj;
                const int ol = dc_owner[dact ? dci : 0];
                const int ox = __shfl(nx0, ol, 64), oy = __shfl(ny0, ol, 64);
                const int os = ol & 7;
                const int odx = (os == 1 || os == 7) ? 1 : -1, ody = (os == 1 || os == 3) ? 1 : -1;
                const int cx = ox + (kk + 1) * odx, cy = oy + (kk + 1) * ody;
                const uint32_t sj = sj_job[sact ? sji : 0];
                bool hv = (t & 1) != 0;                                            // vertical scan?
                int jl = hv ? cx : cy, jp = hv ? cy : cx, jsg = hv ? ody : odx;
                if (last) { hv = (sj & 1u) != 0; jsg = (sj & 2u) ? 1 : -1; jl = (int)((sj >> 2) & 0xFFFu); jp = (int)(sj >> 14); }
                const View V = hv ? Vcol : Vrow;
#ifdef DMPP_DEBUG_SEARCH
                const int r = jump_lane(V, last ? sact : dact, jl, jp, jsg, hv ? gx : gy, hv ? gy : gx, &c_scan);
#else
                const int r = jump_lane(V, last ? sact : dact, jl, jp, jsg, hv ? gx : gy, hv ? gy : gx);
#endif
                bool cblk = false, cstop = false;
                if (dact && (t & 1) == 0) {
                    cblk = B.blk(cx, cy);
                    const bool forced = (B.blk(cx - odx, cy) && !B.blk(cx - odx, cy + ody)) || (B.blk(cx, cy - ody) && !B.blk(cx + odx, cy - ody));
                    cstop = cblk || (cx == gx && cy == gy) || forced;
                }
                const unsigned long long sm = __ballot(dact && (cstop || (!last && r > 0))), bk = __ballot(cblk);
                const unsigned gs = (unsigned)(sm >> (grp * kDiagGroup)) & ((1u << kDiagGroup) - 1u);
                const unsigned gb = (unsigned)(bk >> (grp * kDiagGroup)) & ((1u << kDiagGroup) - 1u);
                int drun = kDiagK;
                if (gs) { const int k1 = (__ffs((int)gs) - 1) >> 1; drun = ((gb >> (2 * k1)) & 1u) ? 0 : k1 + 1; }
                // results back to the owner lanes: a straight one sits on lane 14 | 15 of group my_sj / 2, a diagonal one on its whole group
                const int from_s = __shfl(r, ((my_sj >> 1) & (kDiagPerRound - 1)) * kDiagGroup + (kDiagGroup - 2) + (my_sj & 1), 64);
                const int from_d = __shfl(drun, (my_dc & (kDiagPerRound - 1)) * kDiagGroup, 64);
                if (rnd == 0 && want_jump) run = from_s;
                if (want_diag && (my_dc / kDiagPerRound) == rnd) run = from_d;
            }
#ifdef DMPP_DEBUG_SEARCH
            long long te2 = clock64(); t_jump += te2 - td; c_jobs += n_dc; c_pass += n_rounds_j;
#endif
            // ---- push in batch order, then direction order ----
            const bool push = run > 0;
            const unsigned pm = (unsigned)__ballot(push);
            const int cnt = __popc(pm);
            if (cnt) {
                const int nx = nx0 + run * sdx, ny = ny0 + run * sdy;
                const int fn = gcur + run * ((s & 1) ? 14 : 10) + hfun(nx, ny, gx, gy);
                // f/2 lives in 16 bits (0xFFFF = dead slot): a push at or beyond DMPP_F_LIMIT ends the search.  The oracle tests
                // each push in turn, the range before the capacity: the earlier of the two failing pushes decides the status.
                const unsigned rm = (unsigned)__ballot(push && fn >= DMPP_F_LIMIT);
                if (rm || live + cnt > cap) {
                    const int k_range = rm ? __popc(pm & ((1u << (__ffs((int)rm) - 1)) - 1u)) : 0x7FFFFFFF;
                    const int k_cap = live + cnt > cap ? cap - live : 0x7FFFFFFF;
                    status = k_range <= k_cap ? DMPP_G_COST_RANGE : DMPP_G_OVERFLOW;
                    break;
                }
                if (n_open + cnt > kOpenCap) {
                    // squeeze the dead slots out, keeping the push order (ballot + prefix popcount)
                    int w = 0;
                    for (int q0 = 0; q0 < n_open; q0 += DMPP_WAVE) {
                        const int i = q0 + lane;
                        uint32_t f2 = 0xFFFFu, ee = 0; uint16_t rr = 0;
                        if (i < n_open) { f2 = o_f2[i]; ee = o_ent[i]; rr = o_run[i]; }
                        const bool alive = f2 != 0xFFFFu;
                        const unsigned long long am = __ballot(alive);
                        wave_order();
                        if (alive) {
                            const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
                            o_f2[w + r] = (uint16_t)f2; o_ent[w + r] = ee; o_run[w + r] = rr;
                        }
                        w += __popcll(am);
                        wave_order();
                    }
                    n_open = w;
                }
                if (push) {
                    const int slot = n_open + __popc(pm & ((1u << lane) - 1u));
                    o_f2[slot] = (uint16_t)(fn >> 1);
                    o_ent[slot] = (uint32_t)nx | ((uint32_t)ny << 12) | ((uint32_t)s << 24);
                    o_run[slot] = (uint16_t)run;
                }
                n_open += cnt; live += cnt; n_push += cnt;
                wave_order();
            }
#ifdef DMPP_DEBUG_SEARCH
            t_push += clock64() - te2;
#endif
        }
    }

#ifdef DMPP_DEBUG_SEARCH
    t_done = clock64();
#endif
    // ---- reduce the digest, rebuild the path from the runs, publish ----
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        uint32_t lo = (uint32_t)digest, hi = (uint32_t)(digest >> 32);
        lo = __shfl_xor((int)lo, sft, 64); hi = __shfl_xor((int)hi, sft, 64);
        digest += ((uint64_t)hi << 32) | lo;
    }
    int path_len = 0;
    bool slow_walk = !hash_complete;           // the HBM copy of direction | run answers instead of the LDS hash
    if (status == DMPP_G_FOUND && hash_complete) {
        // Walk the runs back from the goal: lane 0 looks each closed cell up in the LDS hash (direction and run sit in
        // the same slot) and lists the hops in the LDS arrays of the dead open list.  The length is then known before
        // a cell is written, so every cell goes straight to its final place, path[keep-1-k] for the k-th cell counted
        // from the goal: one lane per hop, offsets from a wave prefix sum of the run lengths.
        int L = 1, hops = 0, bad = 0;          // bad: 1 = inconsistent closed set, 2 = more hops than the LDS list holds
        if (lane == 0) {
            int cur = goal;
            while (cur != start) {
                if (hops >= kOpenCap) { bad = 2; break; }
                uint32_t hh = ((uint32_t)cur * 2654435761u) >> (32 - kClosedLog);
                int v = -1;
                for (int probe = 0; probe < kClosedTab; probe++) {
                    const uint32_t e2 = c_tab[hh]; const int inf = c_info[hh];
                    if (e2 == (uint32_t)cur + 1u) { v = inf; break; }
                    if (e2 == 0u) break;
                    hh = (hh + 1) & (kClosedTab - 1);
                }
                const int pd = v & 15, rn = v >> 4;
                if (v < 0 || rn == 0 || pd > 7) { bad = 1; break; }
                // dx, dy of direction pd from two packed tables (2 bits each, value + 1)
                const int dx = (int)((0x901Au >> (2 * pd)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * pd)) & 3u) - 1;
                const int step = dy * W + dx;
                o_ent[hops] = (uint32_t)cur; o_run[hops] = (uint16_t)rn; o_f2[hops] = (uint16_t)pd;
                hops++; L += rn; cur -= rn * step;
            }
        }
        L = __builtin_amdgcn_readfirstlane(L);
        hops = __builtin_amdgcn_readfirstlane(hops);
        bad = __builtin_amdgcn_readfirstlane(bad);
        wave_sync();
        if (bad == 1) status = DMPP_G_INTERNAL;
        else if (bad == 2) slow_walk = true;
        else {
            int keep = L;
            if (L > c.max_path) { keep = c.max_path; status = DMPP_G_PATH_TRUNC; }
            path_len = keep;
            int kbase = 0;
            for (int j0 = 0; j0 < hops; j0 += DMPP_WAVE) {
                const int j = j0 + lane;
                const int rn = j < hops ? (int)o_run[j] : 0;
                int incl = rn;
#pragma unroll
                for (int sft = 1; sft < DMPP_WAVE; sft <<= 1) { const int t = __shfl_up(incl, sft, 64); if (lane >= sft) incl += t; }
                int ec = 0, step = 0;
                const int idx0 = kbase + incl - rn;
                if (rn) {
                    const int pd = o_f2[j];
                    ec = (int)o_ent[j];
                    const int dx = (int)((0x901Au >> (2 * pd)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * pd)) & 3u) - 1;
                    step = dy * W + dx;
                }
                // short runs (diagonal jumps, hops between close jump points): the hop's lane writes its cells; long
                // straight runs: the whole wave writes one run together
                constexpr int kLongRun = 12;
                if (rn && rn <= kLongRun) { int idx = idx0; for (int r = 0; r < rn && idx < keep; r++, idx++) path[keep - 1 - idx] = ec - r * step; }
                unsigned long long lm = __ballot(rn > kLongRun);
                while (lm) {
                    const int src = __ffsll((long long)lm) - 1;
                    lm &= lm - 1;
                    const int h_ec = __shfl(ec, src, 64), h_step = __shfl(step, src, 64), h_rn = __shfl(rn, src, 64), h_idx = __shfl(idx0, src, 64);
                    for (int r = lane; r < h_rn; r += DMPP_WAVE) if (h_idx + r < keep) path[keep - 1 - (h_idx + r)] = h_ec - r * h_step;
                }
                kbase += __shfl(incl, DMPP_WAVE - 1, 64);
            }
            if (lane == 0 && keep == L) path[0] = start;
        }
    }
    if (status == DMPP_G_FOUND && slow_walk) {
        // Walk the runs back from the goal (lane 0 follows dir/run of each closed cell; it wrote them
        // itself), a chunk of runs at a time through the LDS arrays of the dead open list; all lanes
        // write the cells of a chunk goal-first into path[], which is reversed in place at the end.
        int cur = goal, k = 0;                            // k = cells written so far (goal side)
        bool done = false, broken = false;
        while (!done && !broken) {
            int hops = 0;
            if (lane == 0) {
                int kk = k;
                while (cur != start && hops < kOpenCap) {
                    const int v = pin[cur];
                    const int pd = v & 15, rn = v >> 4;
                    if (rn == 0 || pd > 7 || kk > N) { broken = true; break; }
                    const int dx = (pd == 0 || pd == 1 || pd == 7) ? 1 : ((pd >= 3 && pd <= 5) ? -1 : 0);
                    const int dy = (pd >= 1 && pd <= 3) ? 1 : ((pd >= 5) ? -1 : 0);
                    o_ent[hops] = (uint32_t)cur; o_run[hops] = (uint16_t)rn; o_f2[hops] = (uint16_t)pd;
                    hops++; kk += rn;
                    cur -= rn * (dy * W + dx);
                }
                done = cur == start;
            }
            hops = __builtin_amdgcn_readfirstlane(hops);
            cur = __builtin_amdgcn_readfirstlane(cur);
            done = __builtin_amdgcn_readfirstlane((int)done) != 0;
            broken = __builtin_amdgcn_readfirstlane((int)broken) != 0;
            wave_sync();
            for (int j = 0; j < hops; j++) {
                const int ec = (int)o_ent[j], rn = o_run[j], pd = o_f2[j];
                const int dx = (pd == 0 || pd == 1 || pd == 7) ? 1 : ((pd >= 3 && pd <= 5) ? -1 : 0);
                const int dy = (pd >= 1 && pd <= 3) ? 1 : ((pd >= 5) ? -1 : 0);
                const int step = dy * W + dx;
                for (int r = lane; r < rn; r += DMPP_WAVE)
                    if (k + r < c.max_path) path[k + r] = ec - r * step;
                k += rn;
            }
            wave_sync();
        }
        if (broken) { status = DMPP_G_INTERNAL; }
        else {
            if (lane == 0 && k < c.max_path) path[k] = start;
            const int L = k + 1;
            int keep = L;
            if (L > c.max_path) { keep = c.max_path; status = DMPP_G_PATH_TRUNC; }
            path_len = keep;
            wave_sync();
            for (int i = lane; i < keep / 2; i += DMPP_WAVE) {     // goal-first -> start-first
                const int a0 = path[i], b0 = path[keep - 1 - i];
                path[i] = b0; path[keep - 1 - i] = a0;
            }
        }
    }
#ifdef DMPP_DEBUG_SEARCH
    if (lane == 0) { long long te = clock64(); int32_t* dbg = path + c.max_path - 16; dbg[0] = c_iter; dbg[1] = c_scan; dbg[2] = c_jobs; dbg[3] = c_pass; dbg[4] = c_scan;
        dbg[5] = (int)(t_pop >> 4); dbg[6] = (int)(t_closed >> 4); dbg[7] = (int)(t_cand >> 4); dbg[8] = (int)(t_jump >> 4); dbg[9] = (int)(t_push >> 4); dbg[10] = (int)((te - t_done) >> 4); dbg[11] = (int)((te - t0) >> 4); dbg[12] = (int)((t_loop - t_entry) >> 4); dbg[13] = (int)((te - t_entry) >> 4); dbg[14] = (int)(w_entry & 0x7FFFFFFF); dbg[15] = (int)(wall_clock64() & 0x7FFFFFFF); dbg[4] = (int)((t_nz - t_tr) >> 4); }
#endif
    if (GBM && lane == 0 && start_was_set) {           // the bitmaps in HBM are the grid pp_get_grid returns: leave them as rasterised
        const int sx = start % W, sy = start / W;
        bm[start >> 5] |= 1u << (start & 31);
        bmT[sx * HW + (sy >> 5)] |= 1u << (sy & 31);
    }
    if (lane == 0) {
        cost_out[scene] = (int32_t)min((clock64() - t_begin) >> kOrderShift, (long long)(kOrderClasses - 1));   // launch-order key of the next tick (k_order)
        go.order_digest = digest; go.status = status; go.n_expanded = n_exp; go.n_pushed = n_push; go.n_rounds = n_rounds;
        go.path_len = path_len; go.path_cost = path_cost; go.start_cell = start; go.goal_cell = goal;
    }
}

// ---------------------------------------------------------------------------------------
// G3.  256 threads = 4 waves per scene; wave w scores candidates w, w+4, ...
// NW waves per scene score the candidates NW at a time: 4 for batches (four scenes per CU), 16 for the few scenes of a
// latency-bound tick (17 candidates in 2 rounds instead of 5).
template <int NW>
struct ScoreShared {
    GlobalPoint2D cand[NW][DMPP_PATH_POINTS];
    double seg[NW][DMPP_PATH_POINTS];        // |P_i - P_{i+1}| of the wave's current candidate
    GlobalPoint2D pts[DMPP_PATH_POINTS];     // grid-path prefix in metres (lookahead_cells+1 <= 200)
    double cum[DMPP_PATH_POINTS];
    double rx[kMaxObsLds], ry[kMaxObsLds], rr[kMaxObsLds], rt2[kMaxObsLds];   // obstacles that can matter: x, y, radius, cutoff^2
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

template <int NW>
__global__ void __launch_bounds__(NW * DMPP_WAVE)
k_score(PlannerConfig c, int n_scenes, const SceneIn* __restrict__ in, const ObPoint* __restrict__ obs_now,
        const int32_t* __restrict__ paths, GridOut* __restrict__ gout)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ScoreShared<NW>& sh = *reinterpret_cast<ScoreShared<NW>*>(smem_raw);
    constexpr int kThreads = NW * DMPP_WAVE;
    const int scene = blockIdx.x;
    if (scene >= n_scenes) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const SceneIn& si = in[scene];
    GridOut& go = gout[scene];
    const int W = c.grid_w;
    const int m = si.obs_n;
    const ObPoint* gobs = obs_now + si.obs_off;
    const int32_t* path = paths + (size_t)scene * c.max_path;
    const GlobalPoint3D ego = si.loc.globalpoint;
    const int status = go.status, path_len = go.path_len;
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
    const bool culled = m <= kMaxObsLds;          // longer lists are read from HBM without culling
    if (culled) {
        for (int j = tid; j < m; j += kThreads) {
            const ObPoint o = gobs[j];
            const double thr = (double)o.radius + half_w + c.d_safe;
            if (o.x >= X0 - thr && o.x <= X1 + thr && o.y >= Y0 - thr && o.y <= Y1 + thr) {
                const int q = atomicAdd(&sh.n_rel, 1);           // order is irrelevant: only a minimum is taken
                sh.rx[q] = o.x; sh.ry[q] = o.y; sh.rr[q] = (double)o.radius; sh.rt2[q] = thr * thr;
            }
        }
    }
    __syncthreads();
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
                    double R = 1000;
                    if (den > 0) {
                        const double cosA = (dis1 * dis1 + dis2 * dis2 - dis3 * dis3) / den;
                        const double sinA = sqrt(1 - cosA * cosA);
                        if (sinA >= 0.001) R = 0.5 * dis3 / sinA;
                    }
                    const double kk = 1 / R;
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
