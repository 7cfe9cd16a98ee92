// kernels_g.hpp — HIP kernels of the grid engine (specification in
// DESIGN.md §5 and oracle/dmpp_grid_oracle.c — the reference has no grid code).
//
//   k_rasterise : obstacle list -> u8 occupancy grid in HBM.  One workgroup per (scene, band of
//                 rows): footprints are OR-ed into an LDS bit band, then the band is expanded to
//                 bytes and written with 16-B-per-lane coalesced stores (HBM-write bound).
//   k_search    : bucketed A* with LIFO levels, batches of 8.  ONE WAVE per scene: the u8 grid is read
//                 once with 16-B-per-lane coalesced loads and packed into an LDS bitmap (blocked-
//                 or-closed, 1 bit per cell); the open set is 16 stacks (f/2 mod 16): the top <= 96
//                 entries of each in LDS, older ones spilled to HBM in 32-entry chunks, the heights
//                 in a lane-spread VGPR.  A step takes the top 8 entries of the current level,
//                 closes the open distinct ones and expands them on 8 x 8 lanes (node x direction);
//                 successors are compacted per level with ballot + mbcnt prefix ranks, in batch
//                 order then direction order.  No memory wait except LDS in the common step.
//   k_score     : lattice candidates (cubic Beziers to laterally shifted terminals + the grid
//                 path) scored on collision / curvature / progress, 4 waves per scene.
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
// G1.  grid: [n_scenes][H][W] u8, 0 free / 1 occupied.  band_rows*W must be a multiple of 4096
// bits... (host guarantees band_rows*W % 32 == 0 and W % 16 == 0).
constexpr int kRasterBlock = 256;
constexpr int kRasterMaxObs = 1024;

__global__ void __launch_bounds__(kRasterBlock)
k_rasterise(PlannerConfig c, int n_scenes, int band_rows, const SceneIn* __restrict__ in,
            const ObPoint* __restrict__ obs_now, uint8_t* __restrict__ grid)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem_raw);          // band_rows*W/32 words
    const int scene = blockIdx.x, band = blockIdx.y;
    if (scene >= n_scenes) return;
    const int W = c.grid_w, H = c.grid_h;
    const int row0 = band * band_rows;
    const int rows = min(band_rows, H - row0);
    if (rows <= 0) return;
    const int words = (rows * W) >> 5;
    const int tid = threadIdx.x;
    for (int w = tid; w < words; w += kRasterBlock) bits[w] = 0;
    __syncthreads();
    const SceneIn& si = in[scene];
    const double ox = si.grid_origin.x, oy = si.grid_origin.y;
    const int m = si.obs_n;
    const ObPoint* obs = obs_now + si.obs_off;
    // every thread walks the obstacle list; the cells of one footprint's bounding box (clipped
    // to this band) are spread over the 256 threads
    for (int j = 0; j < m; j++) {
        const ObPoint o = obs[j];
        const double R = (double)o.radius + c.inflate;
        int ix0 = (int)floor((o.x - R - ox) / c.cell) - 1, ix1 = (int)floor((o.x + R - ox) / c.cell) + 1;
        int iy0 = (int)floor((o.y - R - oy) / c.cell) - 1, iy1 = (int)floor((o.y + R - oy) / c.cell) + 1;
        ix0 = max(ix0, 0); ix1 = min(ix1, W - 1);
        iy0 = max(iy0, row0); iy1 = min(iy1, row0 + rows - 1);
        const int bw = ix1 - ix0 + 1, bh = iy1 - iy0 + 1;
        if (bw <= 0 || bh <= 0) continue;
        const double R2 = R * R;
        for (int t = tid; t < bw * bh; t += kRasterBlock) {
            const int iy = iy0 + t / bw, ix = ix0 + t % bw;
            const double cx = ox + ((double)ix + 0.5) * c.cell, cy = oy + ((double)iy + 0.5) * c.cell;
            const double dx = cx - o.x, dy = cy - o.y;
            if (dx * dx + dy * dy <= R2) {
                const int b = (iy - row0) * W + ix;
                atomicOr(&bits[b >> 5], 1u << (b & 31));
            }
        }
    }
    __syncthreads();
    // expand 16 bits -> 16 bytes per lane per store (uint4), fully coalesced
    uint4* out = reinterpret_cast<uint4*>(grid + ((size_t)scene * H + row0) * W);
    const int chunks = (rows * W) >> 4;
    for (int k = tid; k < chunks; k += kRasterBlock) {
        const uint32_t w = bits[k >> 1];
        const uint32_t h16 = (k & 1) ? (w >> 16) : (w & 0xFFFFu);
        uint4 v;
        v.x = ((h16 & 0xFu) * 0x00204081u) & 0x01010101u;
        v.y = (((h16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
        v.z = (((h16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
        v.w = (((h16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
        out[k] = v;
    }
}

// ---------------------------------------------------------------------------------------
// G2.
__device__ __forceinline__ int hfun(int x, int y, int gx, int gy)
{
    int dx = abs(x - gx), dy = abs(y - gy);
    return 10 * max(dx, dy) + 4 * min(dx, dy);
}
__device__ __forceinline__ uint32_t pack_nz4(uint32_t x)   // 4 bytes -> 4 bits (bit k = byte k != 0)
{
    uint32_t nz = ((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u) >> 7;
    return (nz * 0x01020408u) >> 24;
}

struct SearchScratch {          // per scene, HBM
    uint32_t* bucket;           // 16 * bucket_cap entries: cell | dir << 24
    uint8_t*  parent;           // W*H arriving directions (written only for closed cells)
    int32_t*  order;            // order_cap cells or null
    int32_t*  path;             // max_path cells
};

// blockDim = 64 (one wave).  GBM = false: the bitmap (W*H/8 bytes) is dynamic LDS — grids up to
// 1024x1024.  GBM = true: the bitmap is a per-scene HBM/L2 scratch (2048x2048 = 512 KiB does not
// fit the 160 KiB of LDS); same code, global loads/atomics instead of ds_ operations.
constexpr int kWin = 96, kSpill = 32;   // LDS window per level: 8 kept + up to 64 pushed per step + one chunk

template <bool GBM>
__global__ void __launch_bounds__(DMPP_WAVE)
k_search(PlannerConfig c, int n_scenes, int order_cap, const SceneIn* __restrict__ in, const uint8_t* __restrict__ grid,
         uint32_t* __restrict__ buckets, uint8_t* __restrict__ parents, int32_t* __restrict__ orders,
         int32_t* __restrict__ paths, GridOut* __restrict__ gout, uint32_t* __restrict__ gbitmaps)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ uint32_t win[16][kWin];           // LDS tops of the 16 level stacks
    const int scene = blockIdx.x;
    if (scene >= n_scenes) return;
    const int lane = threadIdx.x;
    const int W = c.grid_w, H = c.grid_h, N = W * H, cap = c.bucket_cap;
    uint32_t* bm = GBM ? gbitmaps + (size_t)scene * (N >> 5) : reinterpret_cast<uint32_t*>(smem_raw);   // N/32 words
    const SceneIn& si = in[scene];
    GridOut& go = gout[scene];
    const uint8_t* g = grid + (size_t)scene * N;
    uint32_t* bucket = buckets + (size_t)scene * 16 * cap;
    uint8_t* parent = parents + (size_t)scene * N;
    int32_t* order = orders ? orders + (size_t)scene * order_cap : nullptr;
    int32_t* path = paths + (size_t)scene * c.max_path;

    // ---- occupancy bytes -> LDS bits: lane i of a load covers bytes [16 i, 16 i + 16) of a 1-KiB
    //      span (fully coalesced), packs them to 16 bits and stores one ds_write_b16 ----
    {
        const uint4* g4 = reinterpret_cast<const uint4*>(g);
        uint16_t* bm16 = reinterpret_cast<uint16_t*>(bm);
        const int chunks = N >> 4;
        constexpr int U = 8;                                   // 8 KiB in flight per wave
        for (int c0 = 0; c0 < chunks; c0 += DMPP_WAVE * U) {
            uint4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = c0 + u * DMPP_WAVE + lane;
                if (k < chunks) a[u] = g4[k];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = c0 + u * DMPP_WAVE + lane;
                if (k < chunks)
                    bm16[k] = (uint16_t)(pack_nz4(a[u].x) | (pack_nz4(a[u].y) << 4) | (pack_nz4(a[u].z) << 8) | (pack_nz4(a[u].w) << 12));
            }
        }
    }
    wave_order();

    const int start = cell_of(c, si.grid_origin, si.loc.globalpoint.x, si.loc.globalpoint.y);
    const int goal = cell_of(c, si.grid_origin, si.goal.x, si.goal.y);
    const int gx = goal % W, gy = goal / W;
    int status = -1, n_exp = 0, n_push = 0, n_rounds = 0, path_cost = 0;
    uint64_t digest = 0;       // per-lane partial, summed at the end

    if ((bm[goal >> 5] >> (goal & 31)) & 1u) {
        status = DMPP_G_GOAL_BLOCKED;
    } else {
        if (lane == 0) bm[start >> 5] &= ~(1u << (start & 31));            // the vehicle is where it is
        int fcur = hfun(start % W, start / W, gx, gy);
        // Open set: level k (= f/2 mod 16) is a stack whose top (<= kWin entries) lives in LDS (win[k])
        // and whose older entries are spilled to HBM.  The two heights of level k are kept in lane k
        // of two VGPRs and read with v_readlane.  An entry is x | y << 12 | arriving direction << 24.
        int cnts = 0, gcn = 0;
        {
            const int b0 = (fcur >> 1) & 15;
            if (lane == 0) win[b0][0] = (uint32_t)(start % W) | ((uint32_t)(start / W) << 12) | (8u << 24);
            if (lane == b0) cnts = 1;
        }
        n_push = 1; n_rounds = 1;
        wave_order();
        // lane = node * 8 + direction: 8 batch nodes x 8 directions
        const int node = lane >> 3, my_dir = lane & 7;
        const int ddx = (my_dir == 0 || my_dir == 1 || my_dir == 7) ? 1 : ((my_dir >= 3 && my_dir <= 5) ? -1 : 0);
        const int ddy = (my_dir >= 1 && my_dir <= 3) ? 1 : ((my_dir >= 5) ? -1 : 0);
        const int dcost = (my_dir & 1) ? 14 : 10;
        const unsigned long long below_node = (1ull << (node * 8)) - 1ull;      // lanes of earlier nodes
        // every step consumes at least one entry; there are at most 8N+1 entries
        long long guard = 10ll * N + 64;
        while (status < 0) {
            if (--guard < 0) { status = DMPP_G_INTERNAL; break; }
            const int b = (fcur >> 1) & 15;
            int cb = __builtin_amdgcn_readlane(cnts, b);
            int gb = __builtin_amdgcn_readlane(gcn, b);
            if (cb == 0 && gb == 0) {
                // next non-empty level: 16-bit occupancy mask rotated so that bit k = level b+k
                const unsigned m = (unsigned)(__ballot(lane < 16 && (cnts != 0 || gcn != 0)) & 0xFFFFull);
                if (m == 0) { status = DMPP_G_NO_PATH; break; }
                const unsigned r = ((m >> b) | (m << (16 - b))) & 0xFFFFu;
                fcur += 2 * (__ffs((int)r) - 1);
                n_rounds++;
                continue;
            }
            if (cb < 8 && gb > 0) {
                // slide the (< 8) window entries up and pull a chunk of the spilled part in underneath
                const int take = min(kSpill, gb);
                uint32_t v = 0;
                if (lane < cb) v = win[b][lane];
                wave_order();
                if (lane < cb) win[b][lane + take] = v;
                if (lane < take) win[b][lane] = bucket[(size_t)b * cap + (gb - take) + lane];
                cb += take; gb -= take;
                if (lane == b) gcn = gb;
                wave_order();
            }
            // ---- pop phase: the top min(8, height) entries, node q = q-th from the top ----
            const int avail = min(cb, 8);
            const bool have = node < avail;
            uint32_t e = 0;
            if (have) e = win[b][cb - 1 - node];
            const int x = (int)(e & 0xFFFu), y = (int)((e >> 12) & 0xFFFu), pd = (int)(e >> 24);
            const int cell = y * W + x;
            bool valid = have && !((bm[cell >> 5] >> (cell & 31)) & 1u);
            {   // a cell that appears twice among the 8: the one nearer the top wins
                const unsigned long long om = __ballot(valid);
                bool dup = false;
#pragma unroll
                for (int q = 0; q < 7; q++) {
                    const int cq = __builtin_amdgcn_readlane(cell, q * 8);
                    if (q < node && ((om >> (q * 8)) & 1ull) && cq == cell) dup = true;
                }
                if (dup) valid = false;
            }
            // the goal, or the node that reaches the expansion limit, ends the search at once:
            // entries below it are not looked at
            unsigned long long vm = __ballot(valid && my_dir == 0);                 // bit 8q = node q joins the batch
            {
                const int nvb = __popcll(vm & below_node);
                const unsigned long long stop = __ballot(valid && my_dir == 0 && (cell == goal || n_exp + nvb + 1 >= c.max_expansions));
                if (stop) {
                    const int last = (__ffsll((long long)stop) - 1) >> 3;
                    if (node > last) valid = false;
                    vm = __ballot(valid && my_dir == 0);
                }
            }
            // ---- close: bitmap, parent, order, digest (one lane per batch node) ----
            if (valid && my_dir == 0) {
                const int seq = n_exp + __popcll(vm & below_node);
                atomicOr(&bm[cell >> 5], 1u << (cell & 31));
                parent[cell] = (uint8_t)pd;
                if (order && seq < order_cap) order[seq] = cell;
                digest += mix64(((uint64_t)(uint32_t)seq << 32) | (uint32_t)cell);
            }
            n_exp += __popcll(vm);
            if (lane == b) cnts = cb - avail;
            wave_order();
            if (__ballot(valid && cell == goal)) { status = DMPP_G_FOUND; path_cost = fcur; break; }
            if (n_exp >= c.max_expansions) { status = DMPP_G_LIMIT; break; }
            // ---- expand phase: every lane tests one (node, direction) ----
            const int nx = x + ddx, ny = y + ddy;
            const bool inb = valid && nx >= 0 && ny >= 0 && nx < W && ny < H;
            const int ncell = inb ? ny * W + nx : 0;
            const bool push = inb && !((bm[ncell >> 5] >> (ncell & 31)) & 1u);
            const int kb = (b + ((dcost + hfun(nx, ny, gx, gy) - hfun(x, y, gx, gy)) >> 1)) & 15;   // level of the successor
            unsigned long long rem = __ballot(push);
            n_push += __popcll(rem);
            // ---- push per level in lane order (= batch order, then direction order) ----
            bool overflow = false;
            while (rem) {
                const int q = __ffsll((long long)rem) - 1;
                const int kk = __builtin_amdgcn_readlane(kb, q);
                const bool mine = push && kb == kk;
                const unsigned long long mk = __ballot(mine);
                rem &= ~mk;
                const int cnt = __popcll(mk);
                int base = __builtin_amdgcn_readlane(cnts, kk);
                int gbase = __builtin_amdgcn_readlane(gcn, kk);
                if (base + gbase + cnt > cap) { overflow = true; break; }
                if (base + cnt > kWin) {
                    // spill the bottom 32 or 64 entries of the window to HBM, slide the rest down
                    const int sp = ((base + cnt - kWin + kSpill - 1) / kSpill) * kSpill;
                    if (lane < sp) bucket[(size_t)kk * cap + gbase + lane] = win[kk][lane];
                    uint32_t v = 0;
                    if (lane < base - sp) v = win[kk][sp + lane];
                    wave_order();
                    if (lane < base - sp) win[kk][lane] = v;
                    base -= sp; gbase += sp;
                    wave_order();
                }
                if (mine) {
                    const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    win[kk][base + r] = (uint32_t)nx | ((uint32_t)ny << 12) | ((uint32_t)my_dir << 24);
                }
                if (lane == kk) { cnts = base + cnt; gcn = gbase; }
            }
            wave_order();
            if (overflow) { status = DMPP_G_OVERFLOW; break; }
        }
    }

    // ---- reduce the digest, walk the path back, publish ----
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        uint32_t lo = (uint32_t)digest, hi = (uint32_t)(digest >> 32);
        lo = __shfl_xor((int)lo, sft, 64); hi = __shfl_xor((int)hi, sft, 64);
        digest += ((uint64_t)hi << 32) | lo;
    }
    int path_len = 0;
    if (status == DMPP_G_FOUND) {
        // parent[] was written by lanes of this wave through global memory; order those stores
        // before lane 0's loads (same CU: workgroup scope is enough)
        wave_sync();
        int32_t* rev = reinterpret_cast<int32_t*>(bucket);     // open set is dead now: reuse as scratch (16*cap >= max_path)
        int L = 1;
        if (lane == 0) {
            int cur = goal;
            rev[0] = cur;
            while (cur != start && L <= N) {
                const int pd = parent[cur] & 7;
                const int dx = (pd == 0 || pd == 1 || pd == 7) ? 1 : ((pd >= 3 && pd <= 5) ? -1 : 0);
                const int dy = (pd >= 1 && pd <= 3) ? 1 : ((pd >= 5) ? -1 : 0);
                cur -= dy * W + dx;
                if (L < c.max_path) rev[L] = cur;
                L++;
            }
        }
        L = __shfl(L, 0, 64);
        if (L > N) { status = DMPP_G_INTERNAL; L = 1; }
        int keep = L;
        if (status == DMPP_G_FOUND && L > c.max_path) { keep = c.max_path; status = DMPP_G_PATH_TRUNC; }
        path_len = keep;
        wave_sync();
        for (int k = lane; k < keep; k += DMPP_WAVE) path[keep - 1 - k] = rev[k];
    }
    if (lane == 0) {
        go.order_digest = digest; go.status = status; go.n_expanded = n_exp; go.n_pushed = n_push; go.n_rounds = n_rounds;
        go.path_len = path_len; go.path_cost = path_cost; go.start_cell = start; go.goal_cell = goal;
    }
}

// ---------------------------------------------------------------------------------------
// G3.  256 threads = 4 waves per scene; wave w scores candidates w, w+4, ...
struct ScoreShared {
    GlobalPoint2D cand[4][DMPP_PATH_POINTS];
    GlobalPoint2D pts[DMPP_PATH_POINTS];     // grid-path prefix in metres (lookahead_cells+1 <= 200)
    double cum[DMPP_PATH_POINTS];
    ObPoint obs[kMaxObsLds];
    double cost[DMPP_MAX_LATTICE];
    int best;
};

__device__ __forceinline__ double wave_tree_sum(double acc)
{
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) acc += shfl_xor_f64(acc, sft);
    return acc;
}

__global__ void __launch_bounds__(kBlock)
k_score(PlannerConfig c, int n_scenes, const SceneIn* __restrict__ in, const ObPoint* __restrict__ obs_now,
        const int32_t* __restrict__ paths, GridOut* __restrict__ gout)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ScoreShared& sh = *reinterpret_cast<ScoreShared*>(smem_raw);
    const int scene = blockIdx.x;
    if (scene >= n_scenes) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const SceneIn& si = in[scene];
    GridOut& go = gout[scene];
    const int W = c.grid_w;
    const int m = si.obs_n;
    const ObPoint* gobs = obs_now + si.obs_off;
    const ObPoint* obs = gobs;
    if (m <= kMaxObsLds) { for (int j = tid; j < m; j += kBlock) sh.obs[j] = gobs[j]; obs = sh.obs; }
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
        for (int i = tid; i <= a; i += kBlock) {
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
    const double th = thT * c.PI / 180, cs = cos(th), sn = sin(th);
    const double half_w = 0.5 * c.Vehicle_Width;
    GlobalPoint2D* cand = sh.cand[wave];
    for (int k = wave; k < nc; k += 4) {
        double off = 0;
        if (k < nl) {
            off = (double)(k - (nl - 1) / 2) * c.lattice_step;
            GlobalPoint3D e = { T.x + off * sn, T.y + off * (-cs), thT };
            Bezier bz = bezier_setup(c, ego, e);
            for (int i = lane; i < DMPP_PATH_POINTS; i += DMPP_WAVE) cand[i] = bezier_point(bz, i, DMPP_PATH_POINTS);
        } else {
            for (int i = lane; i < DMPP_PATH_POINTS; i += DMPP_WAVE) cand[i] = mean_point(c, sh.pts, sh.cum, a + 1, i, DMPP_PATH_POINTS);
        }
        wave_sync();
        double pen_acc = 0, k2_acc = 0; int first_hit = DMPP_PATH_POINTS;
        for (int q = 0; q < 4; q++) {
            const int i = lane + 64 * q;
            if (i < DMPP_PATH_POINTS) {
                const GlobalPoint2D p = cand[i];
                double clear = __builtin_inf();
                for (int j = 0; j < m; j++) {
                    const double dx = p.x - obs[j].x, dy = p.y - obs[j].y;
                    const double v = sqrt(dx * dx + dy * dy) - (double)obs[j].radius;
                    if (v < clear) clear = v;
                }
                clear = clear - half_w;
                double pen;
                if (clear <= 0) { pen = 1000.0; if (i < first_hit) first_hit = i; }
                else if (clear < c.d_safe) { const double qq = (c.d_safe - clear) / c.d_safe; pen = qq * qq; }
                else pen = 0;
                pen_acc += pen;
                if (i >= 1 && i <= DMPP_PATH_POINTS - 2) {
                    const double R = radius3_fenced(cand[i - 1], p, cand[i + 1]);
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
            GlobalPoint2D p;
            if (k < nl) {
                const double off = (double)(k - (nl - 1) / 2) * c.lattice_step;
                GlobalPoint3D e = { T.x + off * sn, T.y + off * (-cs), thT };
                Bezier bz = bezier_setup(c, ego, e);
                p = bezier_point(bz, tid, DMPP_PATH_POINTS);
            } else p = mean_point(c, sh.pts, sh.cum, a + 1, tid, DMPP_PATH_POINTS);
            go.best_path[tid] = p;
        }
    }
}

}  // namespace dmpp
