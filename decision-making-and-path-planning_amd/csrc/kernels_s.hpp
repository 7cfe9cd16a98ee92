// kernels_s.hpp — G2, the jump-point A* of the grid engine (specification: oracle/dmpp_grid_oracle.c, DESIGN.md §5;
// the reference has no grid code).  ONE searching wave per scene.
//
//   k_search<K, SW> : the scene's workgroup (SW = 4 waves; 16 when few scenes have the chip to themselves) rasterises its own obstacle list straight into LDS as a SPARSE two-view
//                 bitmap - row-major for E/W scans, column-major for N/S scans; per line a mask of the words that hold an
//                 obstacle bit and the index of the line's first stored word (per-line CSR); only non-zero words are stored.
//                 No occupancy grid crosses HBM, and a scene needs ~12 KB (64 obstacles) instead of 64 KB of bitmaps,
//                 whatever the grid size.  Then wave 0 searches.  A scene whose non-zero words do not fit the launch's LDS
//                 budget writes the DENSE form of the two views to HBM instead and is searched there, by the same loop.
//
// The search loop (search_core) is shared by the two forms:
//   * a straight jump is ONE lane (jump_lane): it visits, in travel order, only the words where the line or one of its
//     two neighbours has an obstacle bit (or the goal) and builds  blocked | forced | goal  with two shifts;
//   * a diagonal jump (<= DMPP_DIAG_JUMP cells) is a group of 16 lanes: lane pair k scans horizontally | vertically from
//     cell k+1 of the diagonal; the pair of cell 8 is free for the straight successors, so one round serves a step;
//   * a step takes up to 4 open entries of the minimal f (their g is final, so the search stays optimal), closes them
//     and expands them on 4 x 8 lanes (node x direction);
//   * the open list lives in LDS in push order (f/2, x|y|dir, run): minimum by DPP wave reduction, ties picked with
//     ballots, dead slots squeezed out with ballot + prefix-popcount compaction;
//   * closed cells: an LDS hash (atomicCAS insertion, direction + run length beside the cell).  A scene that outgrows it
//     (> 768 closed cells) spills: it zeroes its bit set in HBM only then, replays the hash into it and goes on there.
#pragma once
#include <type_traits>
#include "dev_geom.hpp"

#ifndef DMPP_PRIO_T2
#define DMPP_PRIO_T2 12         // steps after which a searching wave raises its issue priority to 2 ...
#define DMPP_PRIO_T3 36         // ... and to 3 (experiment knobs)
#endif
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

__device__ __forceinline__ int hfun(int x, int y, int gx, int gy)
{
    int dx = abs(x - gx), dy = abs(y - gy);
    return 10 * max(dx, dy) + 4 * min(dx, dy);
}

constexpr int kOpenCap = DMPP_OPEN_CAP;
constexpr uint32_t kEntryForcedA = 1u, kEntryForcedB = 2u, kEntryAheadBlocked = 4u;      // bits 28.. of an open-list entry
// LDS closed-set hash: 2^CL slots, at most 3/4 full; beyond that the scene spills to a bit set in HBM (zeroed only then: W * H / 8
// bytes - 512 KiB at 2048 x 2048, where the line metas already limit a CU to two scenes and a larger hash costs no occupancy)
template <int K> constexpr int closed_log_of() { return K == 2 ? 10 : 9; }      // LDS closed-set hash; beyond kClosedMax the scene spills to HBM
constexpr int kDiagK = DMPP_DIAG_JUMP;                  // cells a diagonal jump looks ahead
constexpr int kDiagGroup = 2 * kDiagK;                  // lanes per diagonal jump: (cell, horizontal | vertical component)
constexpr int kDiagPerRound = DMPP_WAVE / kDiagGroup;
constexpr int kMaxDiag = 16;                            // diagonal jumps of a step: <= 4 nodes x 4 (start) / x 3
constexpr int kSearchSetupWaves = 4, kSearchBlock = kSearchSetupWaves * DMPP_WAVE;   // set-up waves per scene in a batch ...
constexpr int kSearchSetupWavesWide = 16;                                             // ... and when a handful of scenes has the chip to itself
constexpr int kOrderShift = 13;                         // search-time classes of 8 Ki cycles (k_order)
constexpr int kOrderClasses = 1024;

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

__device__ __forceinline__ int popc_m(uint32_t v) { return __popc(v); }
__device__ __forceinline__ int popc_m(uint64_t v) { return __popcll(v); }
__device__ __forceinline__ int lsb_m(uint32_t v) { return __ffs((int)v) - 1; }
__device__ __forceinline__ int lsb_m(uint64_t v) { return __ffsll((long long)v) - 1; }
__device__ __forceinline__ int msb_m(uint32_t v) { return 31 - __clz((int)v); }
__device__ __forceinline__ int msb_m(uint64_t v) { return 63 - __clzll((long long)v); }

// ---------------------------------------------------------------------------------------
// Views.  A "view" is a bit matrix of NL lines x LW words of 32 cells: the row-major one for E/W travel (line = y,
// position along the line = x) or the column-major one for N/S (line = x, position = y); the forced-neighbour test only
// needs the two neighbouring lines, so both axes share the code.  A line is described by LineM: `mask` has bit w set when
// word w of the line may hold an obstacle bit (a scan only ever visits such words), `off` locates the line's words.
constexpr uint32_t kLineOutside = 0xFFFFFFFFu;          // LineM.off of a line outside the grid: every cell blocked
template <class M> struct LineM { M mask; uint32_t off; };

// Sparse view in LDS: only the words flagged in `mask` are stored, in word order, from data[off].
//   K = 0: LW <= 16, one u32 per line (mask | off << 16);  K = 1: LW <= 32, uint2 {mask, off};
//   K = 2: LW <= 64, u64 mask[] and u16 off[] (a view's data never exceeds the 160 KB of LDS: < 65536 words).
#define DMPP_LDS __attribute__((address_space(3)))
template <int K>
struct SparseView {
    using M = std::conditional_t<K == 2, uint64_t, uint32_t>;
    // address_space(3): the compiler must know these are LDS (ds_read / ds_or), not generic pointers (flat_load)
    DMPP_LDS uint32_t* data; DMPP_LDS unsigned char* meta; DMPP_LDS uint16_t* off2; int LW, NL;

    __device__ __forceinline__ LineM<M> line(int l) const
    {
        // branch-free (an unconditional read of line 0 for a line outside the grid, then selects): reads of several lines
        // issue together and share one wait
        LineM<M> r;
        const bool in = (unsigned)l < (unsigned)NL;
        const int li = in ? l : 0;
        if constexpr (K == 0) { const uint32_t v = ((DMPP_LDS const uint32_t*)meta)[li]; r.mask = v & 0xFFFFu; r.off = v >> 16; }
        else if constexpr (K == 1) { r.mask = ((DMPP_LDS const uint32_t*)meta)[2 * li]; r.off = ((DMPP_LDS const uint32_t*)meta)[2 * li + 1]; }
        else { r.mask = ((DMPP_LDS const uint64_t*)meta)[li]; r.off = off2[li]; }
        if (!in) { r.mask = 0; r.off = kLineOutside; }
        return r;
    }
    __device__ __forceinline__ uint32_t word(const LineM<M>& m, int w) const
    {   // branch-free: one unconditional LDS read (index 0 when the word is not stored), the rest are selects
        const bool outside = m.off == kLineOutside || (unsigned)w >= (unsigned)LW;
        const M bit = (M)1 << (w & (int)(8 * sizeof(M) - 1));
        const bool present = !outside && (m.mask & bit) != 0;
        const uint32_t idx = present ? m.off + (uint32_t)popc_m((M)(m.mask & (bit - 1))) : 0u;
        const uint32_t v = data[idx];
        return outside ? 0xFFFFFFFFu : (present ? v : 0u);
    }
    // ---- construction (setup waves) ----
    __device__ __forceinline__ void clear_line(int l) const
    {
        if constexpr (K == 0) ((DMPP_LDS uint32_t*)meta)[l] = 0;
        else if constexpr (K == 1) { ((DMPP_LDS uint32_t*)meta)[2 * l] = 0; ((DMPP_LDS uint32_t*)meta)[2 * l + 1] = 0; }
        else { ((DMPP_LDS uint64_t*)meta)[l] = 0; off2[l] = (uint16_t)0; }
    }
    __device__ __forceinline__ void or_mask(int l, M bits) const
    {
        if constexpr (K == 0) __hip_atomic_fetch_or(&((DMPP_LDS uint32_t*)meta)[l], (uint32_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if constexpr (K == 1) __hip_atomic_fetch_or(&((DMPP_LDS uint32_t*)meta)[2 * l], (uint32_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_or(&((DMPP_LDS uint64_t*)meta)[l], (uint64_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ M mask_of(int l) const
    {
        if constexpr (K == 0) return ((DMPP_LDS const uint32_t*)meta)[l] & 0xFFFFu;
        else if constexpr (K == 1) return ((DMPP_LDS const uint32_t*)meta)[2 * l];
        else return ((DMPP_LDS const uint64_t*)meta)[l];
    }
    __device__ __forceinline__ void set_off(int l, uint32_t off) const
    {
        if constexpr (K == 0) ((DMPP_LDS uint32_t*)meta)[l] |= off << 16;
        else if constexpr (K == 1) ((DMPP_LDS uint32_t*)meta)[2 * l + 1] = off;
        else off2[l] = (uint16_t)off;
    }
};
template <int K> __host__ __device__ constexpr int sparse_meta_bytes_per_line() { return K == 0 ? 4 : (K == 1 ? 8 : 10); }

// Dense view: the bitmap of k_rasterise in HBM, one u64 mask per line in LDS (bit w = word w is non-zero).
struct DenseView {
    using M = uint64_t;
    const uint32_t* base; DMPP_LDS const uint64_t* nz; int LW, NL;
    __device__ __forceinline__ LineM<M> line(int l) const
    {
        LineM<M> r; r.mask = 0; r.off = kLineOutside;
        if ((unsigned)l < (unsigned)NL) { r.mask = nz[l]; r.off = (uint32_t)(l * LW); }
        return r;
    }
    __device__ __forceinline__ uint32_t word(const LineM<M>& m, int w) const
    {
        if (m.off == kLineOutside || (unsigned)w >= (unsigned)LW) return 0xFFFFFFFFu;
        if (!((m.mask >> w) & 1ull)) return 0u;
        return base[m.off + w];
    }
};

// single-cell test on the row-major view: outside the grid = blocked
template <class V>
__device__ __forceinline__ bool cell_blocked(const V& rowv, int x, int y)
{
    const auto m = rowv.line(y);
    return (rowv.word(m, x >> 5) >> (x & 31)) & 1u;          // word(): all ones beyond either end of the line
}
template <class V, class L>
__device__ __forceinline__ bool cell_blocked_m(const V& v, const L& m, int p)       // cell at position p of a line already described by m
{
    return (v.word(m, p >> 5) >> (p & 31)) & 1u;             // word(): all ones beyond either end of the line
}

// run (bits 0..11 of the result) = cells travelled from `pos` along `line` in direction sgn to the first stop (blocked | forced |
// goal); 0 = none (the first stop is a wall or the edge of the grid); bits 12 / 13: see the end of the loop.  Safe for starts outside the grid (returns 0).
// Only words whose mask bit is set in the line or one of its two neighbours (or that hold the goal) can contain a stop:
// those candidate words are visited in travel order, nothing else is read.  m0 / mP / mM: the line and its neighbours
// line + 1 / line - 1, loaded by the caller (the diagonal jump's cell tests share them).
template <class V, class L>
__device__ __forceinline__ int jump_lane(const V& vw, const L& m0, const L& mP, const L& mM, bool active, int line, int pos, int sgn,
                                         int gline, int gpos, int* iters = nullptr)
{
    using M = typename V::M;
    int run = 0;
    bool go = active && (unsigned)line < (unsigned)vw.NL && (unsigned)pos < (unsigned)(vw.LW << 5);
    const bool fwd = sgn > 0;
    const int w0 = pos >> 5;
    const bool goal_line = gline == line;
    M cand = 0;
    if (go) {
        cand = m0.mask | mP.mask | mM.mask;
        if (goal_line) cand |= (M)1 << (gpos >> 5);
        cand &= fwd ? (M)~((((M)1) << w0) - (M)1) : (M)((((M)2) << w0) - (M)1);     // the start word and everything ahead of it
        go = cand != 0;
    }
    // Both directions run the same code: a lane that travels towards lower positions bit-reverses the words it reads, so that
    // "the next cell" is always the next higher bit and the first stop the lowest set bit.  All five reads of a visited word
    // (the line, its two neighbours, and the neighbours' next word along the travel, which only matters for the last cell of
    // the word) are issued together, unconditionally: one wait per visited word, no branches around the reads.
    const int bp = fwd ? (pos & 31) : 31 - (pos & 31);                     // the start cell in travel order
    const uint32_t ahead = (bp == 31) ? 0u : ~((2u << bp) - 1u);           // cells strictly ahead of it, in travel order
    const int gb = fwd ? (gpos & 31) : 31 - (gpos & 31);
    for (int it = 0; it <= vw.LW; it++) {
        if (!wave_ballot(go)) break;
        if (iters) ++*iters;
        const int wlo = lsb_m(cand), whi = msb_m(cand);
        const int wi = !go ? 0 : (fwd ? wlo : whi);            // (a lane that is done reads word 0 of its lines: harmless)
        cand &= (M)~(((M)1) << wi);
        const int wn = wi + sgn;
        uint32_t B0 = vw.word(m0, wi), P = vw.word(mP, wi), Mi = vw.word(mM, wi), Pw = vw.word(mP, wn), Mw = vw.word(mM, wn);
        if (!fwd) { B0 = __brev(B0); P = __brev(P); Mi = __brev(Mi); Pw = __brev(Pw); Mw = __brev(Mw); }
        const uint32_t Pn = (P >> 1) | (Pw << 31), Mn = (Mi >> 1) | (Mw << 31);
        uint32_t stop = B0 | (P & ~Pn) | (Mi & ~Mn);
        if (goal_line && (gpos >> 5) == wi) stop |= 1u << gb;
        if (wi == w0) stop &= ahead;
        {   // selects, no branches: the first stop of the word ends the scan (a blocked one with run 0); no stop and no
            // candidate word left: free all the way to the edge of the grid, no jump point
            const int bit = (__ffs((int)stop) - 1) & 31;
            const int np = (wi << 5) + (fwd ? bit : 31 - bit);
            const int rr = fwd ? np - pos : pos - np;
            // with the run: was the stop forced from the line + 1 / line - 1 side (bits 12 / 13)?  The node the run ends at
            // needs exactly these two bits for its own successor rules (search_core), so they travel with it.
            const uint32_t fP = ((P & ~Pn) >> bit) & 1u, fM = ((Mi & ~Mn) >> bit) & 1u;
            run = (go && stop != 0u && !((B0 >> bit) & 1u)) ? (int)((uint32_t)rr | (fP << 12) | (fM << 13)) : run;
            go = go && stop == 0u && cand != 0;
        }
    }
    return run;
}

// ---------------------------------------------------------------------------------------
// G1 inside the search: the footprint of one obstacle, exactly as oracle/dmpp_grid_oracle.c defines it - cell (ix, iy)
// is occupied iff  dx*dx + dy*dy <= R*R  for its centre, R = radius + inflate - but produced span by span.
// For a fixed row the predicate is true on a contiguous run of columns (every step of its evaluation is monotone in the
// distance of the column from the obstacle, rounding included), so a row is described by its first and last true column;
// both are found from a sqrt estimate and then MOVED BY THE PREDICATE ITSELF until exact.  Columns likewise.
struct Footprint { double ox, oy, R2; int ix0, ix1, iy0, iy1; };

__device__ __forceinline__ Footprint footprint_of(const PlannerConfig& c, GlobalPoint2D origin, const ObPoint& o, int W, int H)
{
    Footprint f;
    const double R = (double)o.radius + c.inflate;
    f.ox = o.x; f.oy = o.y; f.R2 = R * R;
    // the conservative box of the oracle's rasteriser, clipped to the grid
    f.ix0 = max((int)floor((o.x - R - origin.x) / c.cell) - 1, 0);
    f.ix1 = min((int)floor((o.x + R - origin.x) / c.cell) + 1, W - 1);
    f.iy0 = max((int)floor((o.y - R - origin.y) / c.cell) - 1, 0);
    f.iy1 = min((int)floor((o.y + R - origin.y) / c.cell) + 1, H - 1);
    return f;
}

// Exact run [a, b] of indices i in [lo, hi] with  du(i)^2 + dv2 <= R2,  du(i) = (org + (i + 0.5) * cell) - ou.
// `ic` = floor((ou - org) / cell), the index whose centre is nearest to ou (or next to it).  Returns false when empty.
// The true indices form ONE run (see above).  Its ends come from a sqrt estimate of the half-chord, and are certified
//   (1) by an error bound: x_a = (ou - half - org) / cell - 0.5 and x_b = (ou + half - org) / cell - 0.5 are where the run begins
//       and ends on the real line; the predicate as evaluated in double precision can differ from the real-number one only
//       for an index within `eps` of x_a or x_b, and the estimates are themselves within eps of the true values, with
//         eps = (half * 2^-24  [raw hardware sqrt, ~2^-26 relative]  +  cq) / cell,
//         cq  = 2^-47 (2 |org| + extent) + 2^-36 cell  [rounding of the coordinates, of du and of x itself; >= 16 times the bound]
//               + 2^-30 (1 + R2)  [rounding of du^2 + dv2 moved to the boundary: 2^-53 R2 / half <= 2^-33 R for half >= 2^-20 R]
//       so when neither x_a nor x_b lies within eps of an integer (and half >= 2^-20 R), a = ceil(x_a), b = floor(x_b) ARE the
//       run (empty when a > b) - about ten instructions, no evaluation of the predicate (tests/test_oracle_properties.py
//       ::test_span_ends_certified_by_margin replays this against the predicate; the adversarial rasteriser tests on the GPU);
//   (2) otherwise (a few lines in a million) by the predicate itself: true at the end, false just outside; when that fails
//       too, the ends are walked to their place.
__device__ __forceinline__ bool exact_span(double ou, double org, double cell, double inv_cell, double dv2, double R2, double cq, int ic,
                                           int lo, int hi, int& a, int& b)
{
    if (dv2 > R2) return false;                            // du*du >= 0: no index can satisfy the predicate
    auto pred = [&](int i) { const double du = (org + ((double)i + 0.5) * cell) - ou; return du * du + dv2 <= R2; };
    // only an estimate: the raw hardware square root (~2^-26 relative), without the refinement to a correctly rounded one
    const double h2 = R2 - dv2;
    const double half = __builtin_amdgcn_sqrt(h2);
    const double xa = (ou - half - org) * inv_cell - 0.5, xb = (ou + half - org) * inv_cell - 0.5;
    a = clampi((int)ceil(xa), lo - 1, hi + 1);
    b = clampi((int)floor(xb), lo - 1, hi + 1);
    {
        const double lim = 0.5 - (half * 0x1p-24 + cq) * inv_cell;
        const bool sure = (h2 >= R2 * 0x1p-40) & (fabs((xa - floor(xa)) - 0.5) < lim) & (fabs((xb - floor(xb)) - 0.5) < lim);
        if (__builtin_expect(sure, 1)) { a = max(a, lo); b = min(b, hi); return a <= b; }
    }
    // (the rare path: what it evaluates at ic, lo and hi does not depend on the line, and the compiler would hoist those thirty
    // double-precision operations out of the loop over the lines - which runs once per obstacle - into every pass)
    asm volatile("" : "+v"(ic), "+v"(lo), "+v"(hi));
    const bool pa = pred(a), pa1 = pred(a - 1), pb = pred(b), pb1 = pred(b + 1);          // four independent evaluations, no short-circuit
    if (__builtin_expect(!((a <= b) & pa & !pa1 & pb & !pb1), 0)) {
        // an empty run (the footprint only grazes the line), or an estimate one off.  A run that
        // is not empty contains the index nearest to ou: three evaluations certify emptiness; otherwise both ends are walked.
        const bool c0 = pred(ic), c1 = pred(ic - 1), c2 = pred(ic + 1);
        if (!(c0 | c1 | c2)) return false;
        const int t = c0 ? ic : (c1 ? ic - 1 : ic + 1);
        // the run contains t; outside the window [lo, hi] it only matters whether it reaches in
        if ((t < lo && !pred(lo)) || (t > hi && !pred(hi))) return false;
        a = clampi(min(a, t), lo - 1, hi + 1); b = clampi(max(b, t), lo - 1, hi + 1);
        // every walk stays inside the window (plus one cell), so it is bounded by the width of the grid
#pragma nounroll
        while (a >= lo && pred(a - 1)) a--;
#pragma nounroll
        while (a <= hi && !pred(a)) a++;
#pragma nounroll
        while (b <= hi && pred(b + 1)) b++;
#pragma nounroll
        while (b >= lo && b >= a && !pred(b)) b--;
    }
    a = max(a, lo); b = min(b, hi);
    return a <= b;
}

// bits [p0, p1] of a line as (word, mask) pieces
template <class F>
__device__ __forceinline__ void for_words(int p0, int p1, F&& f)
{
    for (int w = p0 >> 5; w <= (p1 >> 5); w++) {
        const int lo = max(p0 - (w << 5), 0), hi = min(p1 - (w << 5), 31);
        const uint32_t bits = (hi == 31 ? 0xFFFFFFFFu : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
        f(w, bits);
    }
}

// exclusive prefix sum over the 256 threads of the setup block (wave scan by shuffles, waves joined through LDS)
template <int SW>
__device__ __forceinline__ int block_excl_scan(int v, int tid, int* s_wave /* [SW + 1] */)
{
    const int lane = tid & 63, wv = tid >> 6;
    int incl = v;
#pragma unroll
    for (int sft = 1; sft < DMPP_WAVE; sft <<= 1) { const int t = __shfl_up(incl, sft, 64); if (lane >= sft) incl += t; }
    if (lane == DMPP_WAVE - 1) s_wave[wv] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int q = 0; q < SW; q++) { const int t = s_wave[q]; if (q < wv) base += t; total += t; }
    if (tid == 0) s_wave[SW] = total;
    __syncthreads();
    return base + incl - v;
}

// Walks the exact spans of every footprint of a scene: emit(colhalf, line, a, b) is called, by the lane that owns the line,
// for every line a footprint touches, [a, b] being its first and last occupied position on that line (row-major half:
// line = y, positions x; column-major half: line = x, positions y).  All kSearchBlock threads.
// A wave owns the obstacles wv, wv + 4, ...; it computes 64 footprints at once (one per lane: the divisions are paid once
// per obstacle, not per line) and then walks them, lanes 0..31 on a footprint's rows and lanes 32..63 on its columns.
template <int SW, class Emit>
__device__ __forceinline__ void for_each_span(const PlannerConfig& c, const SceneIn& si, const ObPoint* __restrict__ obs, int m, Emit&& emit)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = c.grid_w, H = c.grid_h;
    const bool colhalf = lane >= 32;
    const int l32 = lane & 31;
    const double cell = c.cell, inv_cell = 1.0 / c.cell;
    const GlobalPoint2D origin = si.grid_origin;
    const double org_u = colhalf ? origin.y : origin.x, org_v = colhalf ? origin.x : origin.y;
    // the scene's share of the error bound that certifies span ends (exact_span): coordinate rounding, with both axes' origins
    const double cq_scene = 0x1p-47 * (2.0 * (fabs(origin.x) + fabs(origin.y)) + (double)max(W, H) * cell) + 0x1p-36 * cell;
    auto bcast_d = [](double v, int src) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    };
    for (int base = 0; base < m; base += SW * DMPP_WAVE) {
        // this lane's obstacle of the chunk
        Footprint f; f.ox = f.oy = f.R2 = 0; f.ix0 = f.iy0 = 0; f.ix1 = f.iy1 = -1;
        int icx = 0, icy = 0;
        const int j = base + wv + SW * lane;
        if (j < m) {
            f = footprint_of(c, origin, obs[j], W, H);
            icx = (int)floor((f.ox - origin.x) / cell);
            icy = (int)floor((f.oy - origin.y) / cell);
        }
        const unsigned long long anym = wave_ballot(j < m && f.ix1 >= f.ix0 && f.iy1 >= f.iy0);
        const int left = m - base - wv;
        const int cnt = left <= 0 ? 0 : min(DMPP_WAVE, (left + SW - 1) / SW);
        for (int q = 0; q < cnt; q++) {
            if (!((anym >> q) & 1ull)) continue;
            const double fx = bcast_d(f.ox, q), fy = bcast_d(f.oy, q), R2 = bcast_d(f.R2, q);
            const int ix0 = __builtin_amdgcn_readlane(f.ix0, q), ix1 = __builtin_amdgcn_readlane(f.ix1, q);
            const int iy0 = __builtin_amdgcn_readlane(f.iy0, q), iy1 = __builtin_amdgcn_readlane(f.iy1, q);
            const int jcx = __builtin_amdgcn_readlane(icx, q), jcy = __builtin_amdgcn_readlane(icy, q);
            const double ou = colhalf ? fy : fx, ov = colhalf ? fx : fy;
            const int ic = colhalf ? jcy : jcx, l0 = colhalf ? ix0 : iy0, l1 = colhalf ? ix1 : iy1, lo = colhalf ? iy0 : ix0, hi = colhalf ? iy1 : ix1;
            const double cq = cq_scene + 0x1p-30 * (1.0 + R2);
            for (int l = l0 + l32; l <= l1; l += 32) {
                const double dv = (org_v + ((double)l + 0.5) * cell) - ov;
                int a, b;
                const bool any = exact_span(ou, org_u, cell, inv_cell, dv * dv, R2, cq, ic, lo, hi, a, b);
                if (any) emit(colhalf, l, a, b);
            }
        }
    }
}

// Pass 1 of build_sparse_views: a SUPERSET of every exact span, in single precision.  The coordinates are taken relative to
// the grid's origin first (|rel| <= S = grid extent + R), so their rounding is <= 2^-24 S; the squared radius carries a margin
// that covers what this does to R2 - dv^2 (<= 2 R * 2^-22 S + 2^-23 R2, taken four times over), so the half-chord is never
// shorter than the exact one, and the two cells the run is widened by on both sides absorb the rounding of the division
// (<= 2^-22 * grid width cells).  emit() as for_each_span.
template <int SW, class Emit>
__device__ __forceinline__ void for_each_rough_span(const PlannerConfig& c, const SceneIn& si, const ObPoint* __restrict__ obs, int m, Emit&& emit)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = c.grid_w, H = c.grid_h;
    const bool colhalf = lane >= 32;
    const int l32 = lane & 31;
    const float cellf = (float)c.cell, inv_cellf = (float)(1.0 / c.cell);
    const GlobalPoint2D origin = si.grid_origin;
    const double extent = (double)max(W, H) * c.cell;
    auto bcast_f = [](float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
    for (int base = 0; base < m; base += SW * DMPP_WAVE) {
        float rx = 0, ry = 0, r2m = -1.0f;
        int ix0 = 0, ix1 = -1, iy0 = 0, iy1 = -1;
        const int j = base + wv + SW * lane;
        if (j < m) {
            const Footprint f = footprint_of(c, origin, obs[j], W, H);
            ix0 = f.ix0; ix1 = f.ix1; iy0 = f.iy0; iy1 = f.iy1;
            const double R = sqrt(f.R2), S = extent + R;
            rx = (float)(f.ox - origin.x); ry = (float)(f.oy - origin.y);
            r2m = (float)((f.R2 + 4.0 * (R * S * 0x1p-21 + f.R2 * 0x1p-23)) * (1.0 + 0x1p-22));
        }
        const unsigned long long anym = wave_ballot(j < m && ix1 >= ix0 && iy1 >= iy0);
        const int left = m - base - wv;
        const int cnt = left <= 0 ? 0 : min(DMPP_WAVE, (left + SW - 1) / SW);
        for (int q = 0; q < cnt; q++) {
            if (!((anym >> q) & 1ull)) continue;
            const float fx = bcast_f(rx, q), fy = bcast_f(ry, q), R2 = bcast_f(r2m, q);
            const int jx0 = __builtin_amdgcn_readlane(ix0, q), jx1 = __builtin_amdgcn_readlane(ix1, q);
            const int jy0 = __builtin_amdgcn_readlane(iy0, q), jy1 = __builtin_amdgcn_readlane(iy1, q);
            const float ou = colhalf ? fy : fx, ov = colhalf ? fx : fy;
            const int l0 = colhalf ? jx0 : jy0, l1 = colhalf ? jx1 : jy1, lo = colhalf ? jy0 : jx0, hi = colhalf ? jy1 : jx1;
            for (int l = l0 + l32; l <= l1; l += 32) {
                const float dv = ((float)l + 0.5f) * cellf - ov;
                const float h2 = R2 - dv * dv;
                if (h2 < 0.0f) continue;
                const float half = __builtin_sqrtf(h2);
                const int a = max((int)floorf((ou - half) * inv_cellf) - 2, lo), b = min((int)floorf((ou + half) * inv_cellf) + 2, hi);
                if (a <= b) emit(colhalf, l, a, b);
            }
        }
    }
}

// Builds both sparse views of a scene from its obstacle list.  All kSearchBlock threads; returns the words the larger view
// needs (> budget: nothing was filled, the views are unusable).
//   pass 1: which words of which lines the footprints can touch (single-precision span estimates, widened: a superset);
//   offsets: exclusive prefix sum of the word counts over the lines;  pass 2: the exact spans, OR-ed into the words.
template <int K, int SW>
__device__ __forceinline__ int build_sparse_views(const PlannerConfig& c, const SceneIn& si, const ObPoint* __restrict__ obs, int m,
                                                  const SparseView<K>& vr, const SparseView<K>& vc, int budget, int* s_wave,
                                                  long long* tmark = nullptr)
{
    using M = typename SparseView<K>::M;
    const int tid = threadIdx.x;
    const int W = c.grid_w, H = c.grid_h;
    int tm_i = 0;
    auto mark = [&]() { if (tmark) tmark[tm_i++] = clock64(); };
    mark();
    for (int l = tid; l < H; l += (SW * DMPP_WAVE)) vr.clear_line(l);
    for (int l = tid; l < W; l += (SW * DMPP_WAVE)) vc.clear_line(l);
    __syncthreads();
    mark(); mark();
    auto mark_words = [&](bool colhalf, int l, int a, int b) {
        const int wa = a >> 5, wb = b >> 5;
        const M bits = (M)((((M)2) << (wb - wa)) - (M)1) << wa;
        if (colhalf) vc.or_mask(l, bits); else vr.or_mask(l, bits);
    };
#ifdef DMPP_EXACT_PASS1          // experiment: the exact spans in pass 1 too (fewer words allocated, more arithmetic)
    for_each_span<SW>(c, si, obs, m, mark_words);
#else
    for_each_rough_span<SW>(c, si, obs, m, mark_words);
#endif
    __syncthreads();
    mark();
    // ---- offsets: exclusive prefix sum of the word counts over the lines, per view ----
    auto place = [&](const SparseView<K>& vw) {
        const int NL = vw.NL, per = (NL + (SW * DMPP_WAVE) - 1) / (SW * DMPP_WAVE);
        const int l0 = tid * per, l1 = min(l0 + per, NL);
        int cnt = 0;
        for (int l = l0; l < l1; l++) cnt += popc_m(vw.mask_of(l));
        int off = block_excl_scan<SW>(cnt, tid, s_wave);
        const int total = s_wave[SW];
        for (int l = l0; l < l1; l++) { const int k = popc_m(vw.mask_of(l)); vw.set_off(l, (uint32_t)off); off += k; }
        __syncthreads();
        return total;
    };
    const int total_r = place(vr), total_c = place(vc);
    const int need = max(total_r, total_c);
    mark();
    if (need > budget) return need;
    for (int i = tid; i < total_r; i += (SW * DMPP_WAVE)) vr.data[i] = 0;
    for (int i = tid; i < total_c; i += (SW * DMPP_WAVE)) vc.data[i] = 0;
    __syncthreads();
    mark();
    const SparseView<K> vw = (tid & 32) ? vc : vr;           // this lane's half, selected once (by value: per-lane field selects, no stack object)
    for_each_span<SW>(c, si, obs, m, [&](bool, int l, int a, int b) {
        const LineM<M> lm = vw.line(l);
        for_words(a, b, [&](int w, uint32_t bits) {
            __hip_atomic_fetch_or(&vw.data[lm.off + popc_m((M)(lm.mask & ((((M)1) << w) - (M)1)))], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
    });
    __syncthreads();
    mark();
    return need;
}

// The dense form of the same two views, in HBM: H x W/32 words row-major at bm, W x H/32 words column-major behind them;
// one u64 per line in LDS says which words are non-zero.  Only for the scenes whose non-zero words do not fit the LDS
// budget of the launch.  All kSearchBlock threads.
template <int SW>
__device__ __forceinline__ void build_dense_views(const PlannerConfig& c, const SceneIn& si, const ObPoint* __restrict__ obs, int m,
                                                  uint32_t* __restrict__ bm, DMPP_LDS uint64_t* nz_row, DMPP_LDS uint64_t* nz_col)
{
    const int tid = threadIdx.x;
    const int W = c.grid_w, H = c.grid_h, WW = W >> 5, HW = H >> 5, NW = (W * H) >> 5;
    uint32_t* bmT = bm + NW;
    {
        uint4* z4 = reinterpret_cast<uint4*>(bm);
        const uint4 z = { 0u, 0u, 0u, 0u };
        for (int i = tid; i < (2 * NW) >> 2; i += (SW * DMPP_WAVE)) z4[i] = z;
    }
    __threadfence();
    __syncthreads();
    for_each_span<SW>(c, si, obs, m, [&](bool colhalf, int l, int a, int b) {
        uint32_t* line = colhalf ? bmT + (size_t)l * HW : bm + (size_t)l * WW;
        for_words(a, b, [&](int w, uint32_t bits) { atomicOr(&line[w], bits); });
    });
    __threadfence();
    __syncthreads();
    for (int v = 0; v < 2; v++) {
        const uint32_t* src = v ? bmT : bm;
        DMPP_LDS uint64_t* nz = v ? nz_col : nz_row;
        const int LW = v ? HW : WW, NL = v ? W : H;
        for (int line = tid; line < NL; line += (SW * DMPP_WAVE)) {
            uint64_t msk = 0;
            for (int w = 0; w < LW; w++) msk |= (uint64_t)min(src[line * LW + w], 1u) << w;
            nz[line] = msk;
        }
    }
    __syncthreads();
}

// clears one cell in a sparse view (the start cell is always free)
template <int K>
__device__ __forceinline__ void sparse_clear_cell(const SparseView<K>& vw, int line, int pos)
{
    using M = typename SparseView<K>::M;
    const LineM<M> lm = vw.line(line);
    const int w = pos >> 5;
    if (lm.mask & (((M)1) << w)) vw.data[lm.off + popc_m((M)(lm.mask & ((((M)1) << w) - (M)1)))] &= ~(1u << (pos & 31));
}

// ---------------------------------------------------------------------------------------
// The successor rules of jump-point search as lookup tables: what direction s is for a node reached in
// direction d (8 = the start node) - 1: a straight jump, 2: a diagonal jump in the natural direction, 3: a diagonal jump
// that needs a blocked side cell, 0: nothing - and where that side cell lies.  The tables are generated at compile time from
// the rules as the oracle states them (rule_code / side_offset below), so they cannot drift from them.
namespace rules {
constexpr int dxs(int s) { return (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0); }
constexpr int dys(int s) { return (s >= 1 && s <= 3) ? 1 : ((s >= 5) ? -1 : 0); }
constexpr int rule_code(int d, int s)
{
    const int rel = (s - d) & 7;
    const bool is_start = d == 8, d_odd = (d & 1) != 0 && !is_start, d_even = !d_odd && !is_start;
    const bool jump0 = (is_start && (s & 1) == 0) || (d_even && rel == 0) || (d_odd && (rel == 1 || rel == 7));
    const bool plain = (is_start && (s & 1) != 0) || (d_odd && rel == 0);
    const bool sided = (d_even && (rel == 1 || rel == 7)) || (d_odd && (rel == 2 || rel == 6));
    return jump0 ? 1 : (plain ? 2 : (sided ? 3 : 0));
}
constexpr int side_offset(int d, int s, bool want_y)       // the side cell of a "3" successor, relative to the node
{
    const bool d_odd = (d & 1) != 0 && d != 8;
    const int a = want_y ? dys(s) - dys(d & 7) : dxs(s) - dxs(d & 7);
    return d_odd ? a / 2 : a;
}
constexpr uint64_t code_rows(int d0)                        // rows d0 .. d0 + 3: 16 bits per row, 2 bits per direction
{
    uint64_t t = 0;
    for (int d = d0; d < d0 + 4; d++) for (int s = 0; s < 8; s++) t |= (uint64_t)rule_code(d, s) << ((d - d0) * 16 + 2 * s);
    return t;
}
constexpr uint32_t start_row() { uint32_t t = 0; for (int s = 0; s < 8; s++) t |= (uint32_t)rule_code(8, s) << (2 * s); return t; }
constexpr uint32_t kDxTab = 0x901Au, kDyTab = 0x01A9u;     // dx + 1 / dy + 1 of a direction, 2 bits each
constexpr bool tables_ok()
{
    for (int d = 0; d < 8; d++) if (((int)((kDxTab >> (2 * d)) & 3u) - 1) != dxs(d) || ((int)((kDyTab >> (2 * d)) & 3u) - 1) != dys(d)) return false;
    // the device code takes the side offset as (ds - dd) >> (d & 1): equal to side_offset wherever the code is 3
    for (int d = 0; d <= 8; d++) for (int s = 0; s < 8; s++) if (rule_code(d, s) == 3) {
        const int ax = dxs(s) - dxs(d & 7), ay = dys(s) - dys(d & 7);
        if ((ax >> (d & 1)) != side_offset(d, s, false) || (ay >> (d & 1)) != side_offset(d, s, true)) return false;
    }
    return true;
}
static_assert(tables_ok(), "direction tables / side offsets");
constexpr uint64_t kCodeLo = code_rows(0), kCodeHi = code_rows(4);
constexpr uint32_t kCodeStart = start_row();
}  // namespace rules

// ---------------------------------------------------------------------------------------
template <int CL>
struct SearchLds {                               // the static LDS of a searching workgroup (8.2 KB at CL = 9)
    static constexpr int kClosedLog = CL, kClosedTab = 1 << CL, kClosedMax = 3 << (CL - 2);
    uint32_t o_ent[kOpenCap];       // x | y << 12 | arriving direction << 24 | kEntry* flags << 28
    uint16_t o_f2[kOpenCap];        // f / 2, 0xFFFF = dead slot
    uint16_t o_run[kOpenCap];       // run length of the move that reached the cell
    uint32_t c_tab[kClosedTab];     // closed cells (cell + 1, 0 = empty): open addressing, linear probing
    uint16_t c_info[kClosedTab];    // arriving direction | run length << 4 of the cell in the same slot
    uint16_t c_seq[kClosedTab];     // its expansion number (0xFFFF: inserted but never expanded - the search ended in that step)
    uint32_t job[kMaxDiag + 8];     // the jumps of the current step, x | y << 12 | s << 24: diagonal ones (<= 16), then straight ones (<= 8)
    int s_wave[kSearchSetupWavesWide + 1];
    int s_flag;
};

struct SearchOut { int status, n_exp, n_push, n_rounds, path_cost, path_len; uint64_t digest; };

// order_digest of the expansions recorded in the closed-set hash: this lane's share of  sum mix64(seq << 32 | cell)
template <class LDS>
__device__ __forceinline__ uint64_t hash_digest(const LDS& L, int lane)
{
    uint64_t dg = 0;
    for (int i = lane; i < LDS::kClosedTab; i += DMPP_WAVE) {
        const uint32_t e = L.c_tab[i]; const uint32_t sq = L.c_seq[i];
        if (e && sq != 0xFFFFu) dg += mix64(((uint64_t)sq << 32) | (uint64_t)(e - 1u));
    }
    return dg;
}

// Squeezes the dead slots out of the open list, keeping the push order (ballot + prefix popcount); the slots that fall free
// are marked dead again (the pop relies on 0xFFFF at and beyond n_open).  Returns the new n_open.
template <class LDS>
__device__ __forceinline__ int squeeze_open(LDS& L, int n_open, int lane)
{
    int w = 0;
    for (int q0 = 0; q0 < n_open; q0 += DMPP_WAVE) {
        const int i = q0 + lane;
        uint32_t f2 = 0xFFFFu, ee = 0; uint16_t rr = 0;
        if (i < n_open) { f2 = L.o_f2[i]; ee = L.o_ent[i]; rr = L.o_run[i]; }
        const bool alive = f2 != 0xFFFFu;
        const unsigned long long am = wave_ballot(alive);
        wave_order();
        if (alive) {
            const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
            L.o_f2[w + r] = (uint16_t)f2; L.o_ent[w + r] = ee; L.o_run[w + r] = rr;
        }
        w += __popcll(am);
        wave_order();
    }
    for (int i = w + lane; i < n_open; i += DMPP_WAVE) L.o_f2[i] = 0xFFFFu;
    wave_order();
    return w;
}

// ---- the open list beyond LDS --------------------------------------------------------------------------------------------
// The specification allows bucket_cap live entries; LDS holds kOpenCap slots.  When a push does not fit, the entries that would
// be popped LAST - largest f, and among equal f the oldest - go to the scene's spill area in HBM (spill_open keeps the
// kOpenCap / 2 that pop first), appended in push order; when the minimum of the list in LDS reaches the smallest spilled f (or
// the list cannot fill a batch of ties at that f), the newest spilled entries of that f come back IN FRONT of the list
// (merge_open).  Invariant: among entries of equal f, every spilled one is older than every one in LDS, and the spill area
// keeps push order - so "the latest push first" is decided inside LDS exactly as if nothing had been spilled.
// A spill-area entry: x = the o_ent word, y = f / 2 | run << 16.  Both paths are rare (no generated scene spills).
template <class LDS>
__device__ __forceinline__ int spill_open(LDS& L, int n_open /* == live: squeezed */, int lane, uint2* __restrict__ ospill, int& sp_n, uint32_t& sp_min)
{
    constexpr int keep = kOpenCap / 2;
    if (n_open <= keep) return n_open;
    // pop priority key: f first, then the newer (higher slot) first; all keys distinct
    uint32_t K[kOpenCap / DMPP_WAVE];
#pragma unroll
    for (int k = 0; k < kOpenCap / DMPP_WAVE; k++) {
        const int slot = lane + DMPP_WAVE * k;
        K[k] = slot < n_open ? (((uint32_t)L.o_f2[slot] << 16) | (0xFFFFu - (uint32_t)slot)) : 0xFFFFFFFFu;
    }
    uint32_t lo = 0u, hi = 0xFFFFFFFEu;                       // the keep-th smallest key
    for (int it = 0; it < 32 && lo < hi; it++) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kOpenCap / DMPP_WAVE; k++) cnt += __popcll(wave_ballot(K[k] <= mid));
        if (cnt >= keep) hi = mid; else lo = mid + 1u;
    }
    uint32_t mn = 0xFFFFu;
    int base = sp_n;
#pragma unroll
    for (int k = 0; k < kOpenCap / DMPP_WAVE; k++) {
        const int slot = lane + DMPP_WAVE * k;
        const bool out = slot < n_open && K[k] > lo;
        const unsigned long long bm = wave_ballot(out);
        if (out) {
            const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
            const uint32_t f2 = L.o_f2[slot];
            ospill[base + r] = make_uint2(L.o_ent[slot], f2 | ((uint32_t)L.o_run[slot] << 16));
            L.o_f2[slot] = 0xFFFFu;
            mn = min(mn, f2);
        }
        base += __popcll(bm);
    }
    sp_n = base;
    sp_min = min(sp_min, wave_min_u32(mn));
    __threadfence();                                          // the wave reads its own spill area back later (merge_open)
    wave_order();
    return squeeze_open(L, n_open, lane);
}

// Brings the newest spilled entries of f / 2 == sp_min back, in front of the list.  Returns the new n_open (== live).
template <class LDS>
__device__ __forceinline__ int merge_open(LDS& L, int n_open, int lane, uint2* __restrict__ ospill, int& sp_n, uint32_t& sp_min)
{
    n_open = squeeze_open(L, n_open, lane);
    if (kOpenCap - n_open < 2 * DMPP_WAVE) n_open = spill_open(L, n_open, lane, ospill, sp_n, sp_min);    // room for the batch's pushes and for what comes back
    __threadfence();
    const uint32_t X = sp_min;
    int cntX = 0;
    for (int q0 = 0; q0 < sp_n; q0 += DMPP_WAVE) {
        const int i = q0 + lane;
        const uint32_t f = i < sp_n ? (ospill[i].y & 0xFFFFu) : 0x10000u;
        cntX += __popcll(wave_ballot(f == X));
    }
    const int room = kOpenCap - n_open - DMPP_WAVE;           // (>= 64: a step pushes at most 32)
    const int take = min(cntX, room), skip = cntX - take;     // the oldest `skip` of them stay
    // the list moves up by `take` slots (all reads, then all writes)
    {
        uint32_t f2[kOpenCap / DMPP_WAVE], ee[kOpenCap / DMPP_WAVE]; uint16_t rr[kOpenCap / DMPP_WAVE];
#pragma unroll
        for (int k = 0; k < kOpenCap / DMPP_WAVE; k++) {
            const int slot = lane + DMPP_WAVE * k;
            f2[k] = 0xFFFFu; ee[k] = 0; rr[k] = 0;
            if (slot < n_open) { f2[k] = L.o_f2[slot]; ee[k] = L.o_ent[slot]; rr[k] = L.o_run[slot]; }
        }
        wave_order();
#pragma unroll
        for (int k = 0; k < kOpenCap / DMPP_WAVE; k++) {
            const int slot = lane + DMPP_WAVE * k;
            if (slot < n_open) { L.o_f2[slot + take] = (uint16_t)f2[k]; L.o_ent[slot + take] = ee[k]; L.o_run[slot + take] = rr[k]; }
        }
        wave_order();
    }
    // one pass over the spill area: the newest `take` entries of f == X go to slots 0 .. take - 1 (in push order), the others
    // close up (stable), and the new minimum is taken on the way
    int r = 0, w = 0; uint32_t nmin = 0xFFFFu;
    for (int q0 = 0; q0 < sp_n; q0 += DMPP_WAVE) {
        const int i = q0 + lane;
        const bool valid = i < sp_n;
        uint2 e = make_uint2(0u, 0xFFFFu);
        if (valid) e = ospill[i];
        const uint32_t f = e.y & 0xFFFFu;
        const bool isX = valid && f == X;
        const unsigned long long bx = wave_ballot(isX);
        const int myr = r + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bx >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bx, 0u));
        const bool tk = isX && myr >= skip;
        if (tk) { const int dst = myr - skip; L.o_f2[dst] = (uint16_t)f; L.o_ent[dst] = e.x; L.o_run[dst] = (uint16_t)(e.y >> 16); }
        const bool kp = valid && !tk;
        const unsigned long long bk = wave_ballot(kp);
        if (kp) {
            ospill[w + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bk, 0u))] = e;
            nmin = min(nmin, f);
        }
        r += __popcll(bx); w += __popcll(bk);
    }
    sp_n = w;
    sp_min = wave_min_u32(nmin);
    __threadfence();
    wave_order();
    return n_open + take;
}

// The search proper, one wave.  Vrow / Vcol: the two views; closed / pin: this scene's spill area in HBM.
// SPILL = false: the open list lives in LDS only - the loop every generated scene runs; a scene that needs more live entries
// than LDS holds ends with OVERFLOW there and, when the specification allows more (bucket_cap), is searched again by the
// SPILL = true instance (k_search_spill), whose loop carries the spill state - kept out of the common loop: it cost 5 %
// there in scalar-register spills.
template <bool SPILL, class V, class LDS>
__device__ __forceinline__ SearchOut search_core(const PlannerConfig& c, LDS& L, const V& Vrow, const V& Vcol, int start, int goal, int order_cap,
                                        uint32_t* __restrict__ closed, uint16_t* __restrict__ pin, int32_t* __restrict__ order,
                                        int32_t* __restrict__ path, uint2* __restrict__ ospill, int lane
#ifdef DMPP_DEBUG_SEARCH
                                        , long long* dbg_t, int* dbg_c
#endif
                                        )
{
#ifdef DMPP_DEBUG_SEARCH
#define DBG_MARK(slot) { const long long t__ = clock64(); dbg_t[slot] += t__ - dbg_last; dbg_last = t__; }
    long long dbg_last = clock64();
#else
#define DBG_MARK(slot)
#endif
    constexpr int kClosedLog = LDS::kClosedLog, kClosedTab = LDS::kClosedTab, kClosedMax = LDS::kClosedMax;
    const int W = c.grid_w, H = c.grid_h, N = W * H;
    const int gx = goal % W, gy = goal / W;
    // live entries allowed: the specification's bucket_cap; without a spill area (a handle made for bucket_cap <= kOpenCap) what LDS holds
    const int cap = SPILL ? c.bucket_cap : min(c.bucket_cap, kOpenCap);
    int sp_n = 0; uint32_t sp_min = 0xFFFFu;       // (SPILL) spilled entries (all live) and the smallest f / 2 among them
    SearchOut R; R.status = -1; R.n_exp = 0; R.n_push = 1; R.n_rounds = 0; R.path_cost = 0; R.path_len = 0; R.digest = 0;
    int status = -1, n_exp = 0, n_rounds = 0, path_cost = 0;
    int my_pushes = 0;                                   // per lane (a counter nobody reads inside the loop need not be a scalar register)
    uint64_t digest = 0;
    bool hash_complete = true;                 // every closed cell is in the LDS hash (with its direction and run)
    int n_open = 1, live = 1, fmax = -1;
    // lane = node * 8 + s: the eight directions of each of the (<= 4) nodes of a step
    const int s = lane & 7;
    const int sdx = (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0);
    const int sdy = (s >= 1 && s <= 3) ? 1 : ((s >= 5) ? -1 : 0);
    // ---- lane roles, fixed for the whole search ----
    // (node, s) lanes 0..31: the eight directions of each of the (<= 4) nodes of a step; lanes with s == 0 also stand for
    // their node in the closed-set test.
    const int node = (lane >> 3) & 3;
    // scan lanes of the jumps: lanes 0..27 the horizontal scans of (diagonal job j, cell k) = (lane / 7, lane % 7 + 1), lanes
    // 28..55 the vertical ones, lanes 56..63 the straight jumps (<= 8) - four diagonal jumps and every straight jump of a
    // step in ONE round.  Cell 8 of a diagonal needs no scan: whatever one would find, the jump ends there.
    const bool is_dscan = lane < 56, bhv = lane >= 28;
    const int bq = bhv ? lane - 28 : lane;
    const int bj = is_dscan ? bq / 7 : 0, bk = is_dscan ? bq % 7 + 1 : 0;
    // cell-test lanes of the diagonal jumps: lanes 0..31 = (job, cell 1..8)
    const int aj = node, ak = (lane & 7) + 1;
    int guard = 16 * N + 1024;                         // every iteration pops an entry; entries <= 8 per closed cell (N <= 2^24: fits)
    int steps = 0;
    while (status < 0) {
        if (__builtin_expect(--guard < 0, 0)) { status = DMPP_G_INTERNAL; break; }
        // The kernel ends with its longest search, and several searching waves share a SIMD: a search that has already run
        // long issues ahead of the fresh ones (and of the set-up waves, which run at the lowest priority)
        if (steps == DMPP_PRIO_T2) __builtin_amdgcn_s_setprio(2);
        if (steps == DMPP_PRIO_T3) __builtin_amdgcn_s_setprio(3);
        steps++;
#ifdef DMPP_DEBUG_SEARCH
        dbg_c[0]++;
#endif
        if (__builtin_expect((SPILL ? live + sp_n : live) == 0, 0)) { status = DMPP_G_NO_PATH; break; }
        // ---- pop: up to 4 entries of the smallest f, the latest pushes first ----
        // (1) squeeze the dead slots out when they outnumber the live ones: the scans below stay short
        if (__builtin_expect(n_open - live > 64 && n_open > 2 * live, 0)) n_open = squeeze_open(L, n_open, lane);
        // (2) the keys, 64 slots per register; slots at and beyond n_open always hold 0xFFFF, so nothing is range-checked
        const uint32_t v0 = L.o_f2[lane], v1 = L.o_f2[lane + 64], v2 = L.o_f2[lane + 128], v3 = L.o_f2[lane + 192];
        uint32_t v4 = 0xFFFFu, v5 = 0xFFFFu, v6 = 0xFFFFu, v7 = 0xFFFFu;
        const bool upper = n_open > 256;
        if (__builtin_expect(upper, 0)) { v4 = L.o_f2[lane + 256]; v5 = L.o_f2[lane + 320]; v6 = L.o_f2[lane + 384]; v7 = L.o_f2[lane + 448]; }
        const uint32_t fmin2 = wave_min_u32(min(min(min(v0, v1), min(v2, v3)), min(min(v4, v5), min(v6, v7))));
        // spilled entries pop before the list's minimum: they come back first (the list in LDS may even be empty)
        if constexpr (SPILL) if (__builtin_expect(sp_n > 0 && fmin2 > sp_min, 0)) { n_open = merge_open(L, n_open, lane, ospill, sp_n, sp_min); live = n_open; continue; }
        if (__builtin_expect(fmin2 == 0xFFFFu, 0)) { status = DMPP_G_INTERNAL; break; }
        const int f = (int)fmin2 << 1;
        // the (<= 4) slots taken, 16 bits each, first taken in the low bits (scalar: the tie masks are wave-uniform)
        int nt = 0; unsigned long long sel = 0;
#define DMPP_TAKE(vreg, base)                                                                        \
        {                                                                                            \
            unsigned long long tm = wave_ballot((vreg) == fmin2);                                    \
            while (tm && nt < DMPP_JPS_BATCH) {                                                      \
                const int b = 63 - __clzll((long long)tm);                                           \
                tm &= ~(1ull << b);                                                                  \
                sel |= (unsigned long long)((base) + b) << (16 * nt);                                \
                nt++;                                                                                \
            }                                                                                        \
        }
        if (__builtin_expect(upper, 0)) { DMPP_TAKE(v7, 448) DMPP_TAKE(v6, 384) DMPP_TAKE(v5, 320) DMPP_TAKE(v4, 256) }
        DMPP_TAKE(v3, 192) DMPP_TAKE(v2, 128) DMPP_TAKE(v1, 64) DMPP_TAKE(v0, 0)
#undef DMPP_TAKE
        static_assert(DMPP_JPS_BATCH == 4 && kOpenCap <= 65536, "four 16-bit slot numbers in one 64-bit scalar");
        // ties at the smallest spilled f: the list's own (newer) ones go first, but a batch it cannot fill needs the spilled ones
        if constexpr (SPILL) if (__builtin_expect(sp_n > 0 && fmin2 == sp_min && nt < DMPP_JPS_BATCH, 0)) { n_open = merge_open(L, n_open, lane, ospill, sp_n, sp_min); live = n_open; continue; }
        // every (node, s) lane reads its node's entry itself (a broadcast read): no cross-lane traffic afterwards
        const int i0 = (int)(sel & 0xFFFFu);
        const int myi = (int)((sel >> (16 * node)) & 0xFFFFu);
        const bool have = lane < 32 && node < nt;
        const uint32_t e = L.o_ent[myi]; const int run_in = L.o_run[myi];
        wave_order();
        if (have && s == 0) L.o_f2[myi] = 0xFFFFu;
        live -= nt;
        n_open -= (i0 == n_open - 1) ? 1 : 0;
        const int x = (int)(e & 0xFFFu), y = (int)((e >> 12) & 0xFFFu), d = (int)((e >> 24) & 15u);
        const uint32_t eflags = e >> 28;         // what the jump that created the entry saw beside / ahead of it (kEntry*)
        const int cell = y * W + x;
        DBG_MARK(0)
#ifdef DMPP_DEBUG_SEARCH
        dbg_c[1] += nt;
#endif
        // ---- successor rules, evaluated for every popped entry before its closed-set test (the reads overlap the hash probe) ----
        const int gcur = f - hfun(x, y, gx, gy);
        bool jump0, diag0;
        {
            // the rule for (arriving direction d, direction s): a table lookup (namespace rules).  A forced diagonal needs a blocked
            // side cell and a free target; the plain diagonal a free target: the jump that created the entry has seen those cells
            // and left the answers in the entry (kEntryForcedA / B: the diagonal on the counter-clockwise / clockwise side of d,
            // kEntryAheadBlocked: the next cell of the diagonal) - no cell is read here.  The start node has no such record: its
            // diagonals are simply tried (a blocked target ends the jump with run 0).
            const uint32_t row = d >= 8 ? rules::kCodeStart : (uint32_t)((d < 4 ? rules::kCodeLo : rules::kCodeHi) >> ((d & 3) * 16));
            const uint32_t code = (row >> (2 * s)) & 3u;
            jump0 = code == 1u;
            const int rel = (s - d) & 7;
            const bool forced = ((rel <= 2 ? eflags >> 0 : eflags >> 1) & 1u) != 0;
            const bool ahead_free = (eflags & kEntryAheadBlocked) == 0;
            diag0 = (code == 2u && ahead_free) || (code == 3u && forced);
        }
        DBG_MARK(11)
        // ---- closed?  duplicates inside the batch: the earlier one wins; then the closed set (lanes with s == 0) ----
        bool valid = have && s == 0;
        {
            const int c0_ = __builtin_amdgcn_readlane(cell, 0), c1_ = __builtin_amdgcn_readlane(cell, 8), c2_ = __builtin_amdgcn_readlane(cell, 16);
            // (no short-circuits: three compares and lane masks that never change)
            const bool dup = ((node >= 1) & (cell == c0_)) | ((node >= 2) & (cell == c1_)) | ((node >= 3) & (cell == c2_));
            valid = valid & !dup;
        }
        uint32_t my_slot = 0;
        if (__builtin_expect(n_exp + DMPP_JPS_BATCH <= kClosedMax, 1)) {
            // first probe in straight-line code (nearly always the last); a collision chain is walked in a wave-uniform loop
            const uint32_t keyc = (uint32_t)cell + 1u;
            uint32_t hh = ((uint32_t)cell * 2654435761u) >> (32 - kClosedLog);
            uint32_t old = 0u;
            if (valid) old = atomicCAS(&L.c_tab[hh], 0u, keyc);
            if (old == keyc) valid = false;                               // already closed
            bool collide = valid && old != 0u;
            for (int probe = 1; probe < kClosedTab && wave_ballot(collide); probe++) {
                if (collide) {
                    hh = (hh + 1) & (kClosedTab - 1);
                    old = atomicCAS(&L.c_tab[hh], 0u, keyc);
                    if (old == keyc) valid = false;
                    collide = old != 0u && old != keyc;
                }
            }
            my_slot = hh;                                                 // valid: inserted here (its direction | run and number follow below)
        } else {
            if (hash_complete) {
                // ---- spill: this scene outgrows the LDS hash.  Its bit set in HBM is zeroed now, not at launch (most scenes
                //      never get here), the hash is replayed into it, and from here on the bit set answers. ----
                hash_complete = false;
                digest += hash_digest(L, lane);             // the expansions so far (from here on the digest is kept step by step)
                uint4* c4 = reinterpret_cast<uint4*>(closed);
                const uint4 z = { 0u, 0u, 0u, 0u };
                for (int i = lane; i < (N >> 7); i += DMPP_WAVE) c4[i] = z;
                __threadfence();
                wave_sync();
                for (int i = lane; i < kClosedTab; i += DMPP_WAVE) {
                    const uint32_t e2 = L.c_tab[i];
                    if (e2) { const int cc = (int)e2 - 1; atomicOr(&closed[cc >> 5], 1u << (cc & 31)); pin[cc] = L.c_info[i]; }
                }
                __threadfence();
                wave_sync();
            }
            if (valid) {
                const uint32_t old = atomicOr(&closed[cell >> 5], 1u << (cell & 31));
                if ((old >> (cell & 31)) & 1u) valid = false;
            }
        }
        const bool inserted = valid;                 // (while the LDS hash is complete: valid <=> the cell was inserted just now)
        // node mask from the s == 0 lanes (bits 0, 8, 16, 24 of a ballot -> bits 0..3)
        // (one multiply gathers them: bit 8k lands on bit 24 + k)
        auto nodes_of = [](unsigned long long bm) { return (((unsigned)bm & 0x01010101u) * 0x01020408u) >> 24; };
        unsigned vm = nodes_of(wave_ballot(valid));
        {   // the goal, or the entry that reaches the expansion limit, ends the search at once
            const int nvb = __popc(vm & ((1u << node) - 1u));
            const unsigned stop = nodes_of(wave_ballot(valid && (cell == goal || n_exp + nvb + 1 >= c.max_expansions)));
            if (__builtin_expect(stop != 0, 0)) {
                const int last = __ffs((int)stop) - 1;
                if (node > last) valid = false;
                vm = nodes_of(wave_ballot(valid));
            }
        }
        {
            const int seq = n_exp + __popc(vm & ((1u << node) - 1u));
            // the hash entry of every cell inserted in this step: direction | run, and its expansion number (0xFFFF for one that
            // was inserted but cut off by the end of the search) - the digest is summed from the hash at the end, off the chain of steps
            if (inserted && __builtin_expect(hash_complete, 1)) { L.c_info[my_slot] = (uint16_t)(d | (run_in << 4)); L.c_seq[my_slot] = valid ? (uint16_t)seq : (uint16_t)0xFFFFu; }
            if (valid) {
                if (__builtin_expect(!hash_complete, 0)) { pin[cell] = (uint16_t)(d | (run_in << 4)); digest += mix64(((uint64_t)(uint32_t)seq << 32) | (uint32_t)cell); }
                if (order && seq < order_cap) order[seq] = cell;
            }
        }
        if (vm && f > fmax) { fmax = f; n_rounds++; }
        n_exp += __popc(vm);
        if (__builtin_expect(wave_ballot(valid && cell == goal) != 0, 0)) { status = DMPP_G_FOUND; path_cost = f; break; }
        if (__builtin_expect(n_exp >= c.max_expansions, 0)) { status = DMPP_G_LIMIT; break; }
        if (__builtin_expect(vm == 0, 0)) continue;
        DBG_MARK(1)
        const bool nvalid = lane < 32 && ((vm >> node) & 1u);
        const bool want_jump = nvalid && jump0, want_diag = nvalid && diag0;
        int run = 0;
        uint32_t nflags = 0;
        DBG_MARK(2)
        // ---- jumps.  The successor lanes post their jobs (x | y << 12 | s << 24) in LDS, diagonal ones in slots 0..15 and
        //      straight ones in 16..23, in lane order; the scan lanes and the cell-test lanes pick them up. ----
        const unsigned smask = (unsigned)wave_ballot(want_jump), dmask = (unsigned)wave_ballot(want_diag);
        const int n_sj = __popc(smask), n_dc = __popc(dmask);
        const int my_sj = __popc(smask & ((1u << (lane & 31)) - 1u)), my_dc = __popc(dmask & ((1u << (lane & 31)) - 1u));
        if (want_jump || want_diag) L.job[want_diag ? my_dc : kMaxDiag + my_sj] = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24);
        wave_order();
        const int n_rounds_j = (n_sj | n_dc) ? max(1, (n_dc + 3) >> 2) : 0;
        DBG_MARK(7)
        for (int rnd = 0; rnd < n_rounds_j; rnd++) {
            // -- cell tests: lanes 0..31 = (job aj, cell ak) of this round --
            const int dca = rnd * 4 + aj;
            const bool a_act = lane < 32 && dca < n_dc;
            const uint32_t Ja = L.job[a_act ? dca : 0];
            // -- scans --
            const int dcb = rnd * 4 + bj;
            const bool b_act = is_dscan ? dcb < n_dc : (rnd == 0 && lane - 56 < n_sj);
            const uint32_t Jb = L.job[b_act ? (is_dscan ? dcb : kMaxDiag + lane - 56) : 0];
            int jl, jp, jsg; bool hv;
            {
                const int ox = (int)(Jb & 0xFFFu), oy = (int)((Jb >> 12) & 0xFFFu), os = (int)(Jb >> 24);
                if (is_dscan) {
                    const int odx = (os == 1 || os == 7) ? 1 : -1, ody = (os == 1 || os == 3) ? 1 : -1;
                    const int cx = ox + bk * odx, cy = oy + bk * ody;
                    hv = bhv; jl = hv ? cx : cy; jp = hv ? cy : cx; jsg = hv ? ody : odx;
                } else {
                    hv = !(os == 0 || os == 4); jsg = (os == 0 || os == 2) ? 1 : -1; jl = hv ? ox : oy; jp = hv ? oy : ox;
                }
            }
            const V vw = hv ? Vcol : Vrow;                                     // by value: per-lane field selects
            // the six line descriptors of the round (three for the scan, three for the cell test) are read together: one wait
            const int aox = (int)(Ja & 0xFFFu), aoy = (int)((Ja >> 12) & 0xFFFu), aos = (int)(Ja >> 24);
            const int aodx = (aos == 1 || aos == 7) ? 1 : -1, aody = (aos == 1 || aos == 3) ? 1 : -1;
            const int acx = aox + ak * aodx, acy = aoy + ak * aody;
            const auto m0 = vw.line(b_act ? jl : -1), mP = vw.line(b_act ? jl + 1 : -1), mM = vw.line(b_act ? jl - 1 : -1);
            const auto r0 = Vrow.line(a_act ? acy : -1), ra = Vrow.line(a_act ? acy + aody : -1), rb = Vrow.line(a_act ? acy - aody : -1);
            DBG_MARK(8)
            bool cblk = false, cstop = false, cf1 = false, cf2 = false;
            {
                // five independent reads, combined without short-circuits: one wait for all of them, no branches
                const bool b0 = cell_blocked_m(Vrow, r0, acx), b1 = cell_blocked_m(Vrow, r0, acx - aodx), b2 = cell_blocked_m(Vrow, ra, acx - aodx);
                const bool b3 = cell_blocked_m(Vrow, rb, acx), b4 = cell_blocked_m(Vrow, rb, acx + aodx);
                cf1 = a_act & b1 & !b2;            // forced: the cell beside (against the x travel) is blocked, the one diagonally past it free
                cf2 = a_act & b3 & !b4;            // forced: the cell beside (against the y travel) is blocked, the one diagonally past it free
                cstop = a_act & (b0 | ((acx == gx) & (acy == gy)) | cf1 | cf2);
                cblk = b0 & a_act;
            }
            DBG_MARK(9)
#ifdef DMPP_DEBUG_SEARCH
            const int r = jump_lane(vw, m0, mP, mM, b_act, jl, jp, jsg, hv ? gx : gy, hv ? gy : gx, &dbg_c[2]);
#else
            const int r = jump_lane(vw, m0, mP, mM, b_act, jl, jp, jsg, hv ? gx : gy, hv ? gy : gx);
#endif
            DBG_MARK(10)
            // -- results.  Diagonal job j of the round: the first cell k with a cell-test stop (bits j*8 + k-1 of the test ballot)
            //    or a scan hit (bits j*7 + k-1 of the horizontal / 28 + j*7 + k-1 of the vertical half of the scan ballot) ends
            //    the jump; every owner lane works its own job out of the three wave-uniform masks. --
            //    What the new node will need for its own successor rules goes with it (nflags, see the rules above): for the cell a
            //    diagonal jump ends at, the two forced tests of that cell and whether the next cell of the diagonal is blocked; for
            //    the end of a straight jump, the sides its stop was forced from.
            const unsigned long long am = wave_ballot(cstop), kb = wave_ballot(cblk), sb = wave_ballot(is_dscan && r > 0);
            const unsigned long long f1b = wave_ballot(cf1), f2b = wave_ballot(cf2);
            if (want_diag && (my_dc >> 2) == rnd) {
                const int j = my_dc & 3;
                const unsigned ta = (unsigned)(am >> (8 * j)) & 0xFFu, tk = (unsigned)(kb >> (8 * j)) & 0xFFu;
                const unsigned sc = ((unsigned)(sb >> (7 * j)) | (unsigned)(sb >> (28 + 7 * j))) & 0x7Fu;
                const unsigned any = ta | sc;
                const int k1 = any ? __ffs((int)any) - 1 : kDiagK - 1;                  // the cell the jump ends at (0-based)
                run = ((tk >> k1) & 1u) ? 0 : k1 + 1;
                const unsigned t1 = ((unsigned)(f1b >> (8 * j)) >> k1) & 1u, t2 = ((unsigned)(f2b >> (8 * j)) >> k1) & 1u;
                const unsigned ahead_blk = (tk >> (k1 + 1)) & 1u;                       // (cell 9 is not tested: 0, the jump from there will tell)
                const bool ccw_is_1 = s == 1 || s == 5;                                 // which of the two tests opens the diagonal s + 2
                nflags = (ccw_is_1 ? t1 : t2) * kEntryForcedA | (ccw_is_1 ? t2 : t1) * kEntryForcedB | ahead_blk * kEntryAheadBlocked;
            }
            const int from_s = __shfl(r, 56 + (my_sj & 7), 64);
            if (rnd == 0 && want_jump) {
                run = from_s & 0xFFF;
                const unsigned fP = ((unsigned)from_s >> 12) & 1u, fM = ((unsigned)from_s >> 13) & 1u;
                const bool ccw_is_p = s == 0 || s == 6;                                 // line + 1 lies on the counter-clockwise side of E and S travel
                nflags = (ccw_is_p ? fP : fM) * kEntryForcedA | (ccw_is_p ? fM : fP) * kEntryForcedB;
            }
        }
        DBG_MARK(3)
#ifdef DMPP_DEBUG_SEARCH
        dbg_c[3] += n_dc; dbg_c[4] += n_rounds_j;
#endif
        // ---- push in batch order, then direction order ----
        const bool push = run > 0;
        const unsigned pm = (unsigned)wave_ballot(push);
        const int cnt = __popc(pm);
        if (__builtin_expect(cnt != 0, 1)) {
            const int nx = x + run * sdx, ny = y + run * sdy;
            const int fn = gcur + run * ((s & 1) ? 14 : 10) + hfun(nx, ny, gx, gy);
            // f/2 lives in 16 bits (0xFFFF = dead slot): a push at or beyond DMPP_F_LIMIT ends the search.  The oracle tests
            // each push in turn, the range before the capacity: the earlier of the two failing pushes decides the status.
            const unsigned rm = (unsigned)wave_ballot(push && fn >= DMPP_F_LIMIT);
            const int live_all = SPILL ? live + sp_n : live;
            if (__builtin_expect(rm || live_all + cnt > cap, 0)) {
                const int k_range = rm ? __popc(pm & ((1u << (__ffs((int)rm) - 1)) - 1u)) : 0x7FFFFFFF;
                const int k_cap = live_all + cnt > cap ? cap - live_all : 0x7FFFFFFF;
                status = k_range <= k_cap ? DMPP_G_COST_RANGE : DMPP_G_OVERFLOW;
                break;          // (SPILL = false and bucket_cap > kOpenCap: the caller sends the scene to the SPILL = true instance)
            }
            if (__builtin_expect(n_open + cnt > kOpenCap, 0)) {
                n_open = squeeze_open(L, n_open, lane);                                                    // keeps the push order
                if constexpr (SPILL) { if (n_open + cnt > kOpenCap) n_open = spill_open(L, n_open, lane, ospill, sp_n, sp_min); live = n_open; }
            }
            if (push) {
                const int slot = n_open + __popc(pm & ((1u << lane) - 1u));
                L.o_f2[slot] = (uint16_t)(fn >> 1);
                L.o_ent[slot] = (uint32_t)nx | ((uint32_t)ny << 12) | ((uint32_t)s << 24) | (nflags << 28);
                L.o_run[slot] = (uint16_t)run;
            }
            n_open += cnt; live += cnt; my_pushes += push ? 1 : 0;
            wave_order();
        }
        DBG_MARK(4)
    }
    DBG_MARK(5)

    // ---- the digest (from the hash unless the scene spilled), reduced over the wave; then the path from the runs ----
    if (hash_complete) digest += hash_digest(L, lane);
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
        int Lc = 1, hops = 0, bad = 0;         // bad: 1 = inconsistent closed set, 2 = more hops than the LDS list holds
        if (lane == 0) {
            int cur = goal;
            while (cur != start) {
                if (hops >= kOpenCap) { bad = 2; break; }
                uint32_t hh = ((uint32_t)cur * 2654435761u) >> (32 - kClosedLog);
                int v = -1;
                for (int probe = 0; probe < kClosedTab; probe++) {
                    const uint32_t e2 = L.c_tab[hh]; const int inf = L.c_info[hh];
                    if (e2 == (uint32_t)cur + 1u) { v = inf; break; }
                    if (e2 == 0u) break;
                    hh = (hh + 1) & (kClosedTab - 1);
                }
                const int pd = v & 15, rn = v >> 4;
                if (v < 0 || rn == 0 || pd > 7) { bad = 1; break; }
                // dx, dy of direction pd from two packed tables (2 bits each, value + 1)
                const int dx = (int)((0x901Au >> (2 * pd)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * pd)) & 3u) - 1;
                const int step = dy * W + dx;
                L.o_ent[hops] = (uint32_t)cur; L.o_run[hops] = (uint16_t)rn; L.o_f2[hops] = (uint16_t)pd;
                hops++; Lc += rn; cur -= rn * step;
            }
        }
        Lc = __builtin_amdgcn_readfirstlane(Lc);
        hops = __builtin_amdgcn_readfirstlane(hops);
        bad = __builtin_amdgcn_readfirstlane(bad);
        wave_sync();
        if (bad == 1) status = DMPP_G_INTERNAL;
        else if (bad == 2) {
            // more hops than the LDS list holds: spill the hash to HBM and take the chunked walk below
            uint4* c4 = reinterpret_cast<uint4*>(closed);
            (void)c4;
            for (int i = lane; i < kClosedTab; i += DMPP_WAVE) {
                const uint32_t e2 = L.c_tab[i];
                if (e2) pin[(int)e2 - 1] = L.c_info[i];
            }
            __threadfence();
            wave_sync();
            slow_walk = true;
        } else {
            int keep = Lc;
            if (Lc > c.max_path) { keep = c.max_path; status = DMPP_G_PATH_TRUNC; }
            path_len = keep;
            int kbase = 0;
            for (int j0 = 0; j0 < hops; j0 += DMPP_WAVE) {
                const int j = j0 + lane;
                const int rn = j < hops ? (int)L.o_run[j] : 0;
                int incl = rn;
#pragma unroll
                for (int sft = 1; sft < DMPP_WAVE; sft <<= 1) { const int t = __shfl_up(incl, sft, 64); if (lane >= sft) incl += t; }
                int ec = 0, step = 0;
                const int idx0 = kbase + incl - rn;
                if (rn) {
                    const int pd = L.o_f2[j];
                    ec = (int)L.o_ent[j];
                    const int dx = (int)((0x901Au >> (2 * pd)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * pd)) & 3u) - 1;
                    step = dy * W + dx;
                }
                // short runs (diagonal jumps, hops between close jump points): the hop's lane writes its cells; long
                // straight runs: the whole wave writes one run together
                constexpr int kLongRun = 12;
                if (rn && rn <= kLongRun) { int idx = idx0; for (int r = 0; r < rn && idx < keep; r++, idx++) path[keep - 1 - idx] = ec - r * step; }
                unsigned long long lm = wave_ballot(rn > kLongRun);
                while (lm) {
                    const int src = __ffsll((long long)lm) - 1;
                    lm &= lm - 1;
                    const int h_ec = __shfl(ec, src, 64), h_step = __shfl(step, src, 64), h_rn = __shfl(rn, src, 64), h_idx = __shfl(idx0, src, 64);
                    for (int r = lane; r < h_rn; r += DMPP_WAVE) if (h_idx + r < keep) path[keep - 1 - (h_idx + r)] = h_ec - r * h_step;
                }
                kbase += __shfl(incl, DMPP_WAVE - 1, 64);
            }
            if (lane == 0 && keep == Lc) path[0] = start;
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
                    L.o_ent[hops] = (uint32_t)cur; L.o_run[hops] = (uint16_t)rn; L.o_f2[hops] = (uint16_t)pd;
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
                const int ec = (int)L.o_ent[j], rn = L.o_run[j], pd = L.o_f2[j];
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
            const int Lc = k + 1;
            int keep = Lc;
            if (Lc > c.max_path) { keep = c.max_path; status = DMPP_G_PATH_TRUNC; }
            path_len = keep;
            wave_sync();
            for (int i = lane; i < keep / 2; i += DMPP_WAVE) {     // goal-first -> start-first
                const int a0 = path[i], b0 = path[keep - 1 - i];
                path[i] = b0; path[keep - 1 - i] = a0;
            }
        }
    }
    DBG_MARK(6)
    int n_push = my_pushes;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) n_push += __shfl_xor(n_push, sft, 64);
    n_push += 1;                                         // the start node
    R.status = status; R.n_exp = n_exp; R.n_push = n_push; R.n_rounds = n_rounds; R.path_cost = path_cost; R.path_len = path_len; R.digest = digest;
    return R;
#undef DBG_MARK
}

// the searching wave puts its LDS state back to "nothing expanded, the start node open" (before the second attempt)
template <class LDS>
__device__ __forceinline__ void reset_search_lds(LDS& L, int start, int goal, int W, int lane)
{
    for (int i = lane; i < LDS::kClosedTab; i += DMPP_WAVE) L.c_tab[i] = 0;
    for (int i = lane; i < kOpenCap; i += DMPP_WAVE) L.o_f2[i] = 0xFFFFu;
    wave_order();
    if (lane == 0) {
        const int sx = start % W, sy = start / W;
        L.o_ent[0] = (uint32_t)sx | ((uint32_t)sy << 12) | (8u << 24);
        L.o_f2[0] = (uint16_t)(hfun(sx, sy, goal % W, goal / W) >> 1);
        L.o_run[0] = 0;
    }
    wave_order();
}

__device__ __forceinline__ void publish_search(GridOut& go, const SearchOut& R, int start, int goal)
{
    go.order_digest = R.digest; go.status = R.status; go.n_expanded = R.n_exp; go.n_pushed = R.n_push; go.n_rounds = R.n_rounds;
    go.path_len = R.path_len; go.path_cost = R.path_cost; go.start_cell = start; go.goal_cell = goal;
}

#ifdef DMPP_DEBUG_SEARCH
__device__ __forceinline__ void publish_debug(int32_t* path, int max_path, const long long* t, const int* cn, long long t_setup, long long t_total)
{   // [iter, popped, jump_iters, diag_jobs, rounds, pop, closed, cand, jump, push, walk, loop+walk, setup, total] in units of 16 cycles
    int32_t* dbg = path + max_path - 16;
    dbg[0] = cn[0]; dbg[1] = cn[1]; dbg[2] = cn[2]; dbg[3] = cn[3]; dbg[4] = cn[4];
    dbg[5] = (int)(t[0] >> 4); dbg[6] = (int)(t[1] >> 4); dbg[7] = (int)(t[2] >> 4); dbg[8] = (int)(t[3] >> 4); dbg[9] = (int)(t[4] >> 4);
    dbg[10] = (int)(t[6] >> 4); dbg[11] = (int)((t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6]) >> 4); dbg[12] = (int)(t_setup >> 4); dbg[13] = (int)(t_total >> 4);
    dbg[14] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    dbg[15] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    for (int i = 7; i < 16; i++) path[max_path - 64 + i] = (int)(t[i] >> 4);
}
#endif

// ---------------------------------------------------------------------------------------
// The search kernel.  Dynamic LDS: the line metas of both sparse views, then `budget` data words per view (and at least
// (W + H) * 8 bytes: the line masks of the dense form).
//   A scene whose non-zero words fit `budget` is searched on the sparse views in LDS; one that does not fit (overflow[scene]
//   = 1) rasterises the dense form into gbitmaps (HBM) and is searched there - same loop, slower reads.  budget = 0 sends
//   every scene that way (test knob).
//   need_max: running maximum of the words a scene needed (the host sizes the next launches from it).
//   retry_list / retry_cnt: scenes whose open list outgrew LDS (kStatusRetrySpill): searched again by k_search_spill.
template <int K, int SW, bool SPILL>
__device__ __forceinline__ void search_scene(const PlannerConfig& c, int scene, int order_cap, int budget, const SceneIn* __restrict__ in,
         const ObPoint* __restrict__ obs_now, uint32_t* __restrict__ gclosed, uint16_t* __restrict__ pinfo, int32_t* __restrict__ orders,
         int32_t* __restrict__ paths, GridOut* __restrict__ gout, uint32_t* __restrict__ gbitmaps, int32_t* __restrict__ cost_out,
         int32_t* __restrict__ overflow, int32_t* __restrict__ need_max, uint2* __restrict__ ospill_all, int spill_cap,
         int32_t* __restrict__ retry_list, int32_t* __restrict__ retry_cnt,
         SearchLds<closed_log_of<K>()>& L, unsigned char* smem_raw)
{
    using LdsT = SearchLds<closed_log_of<K>()>;
    constexpr unsigned kViewsAt = 0;
    const long long t_begin = clock64();
    const int tid = threadIdx.x, lane = tid & (DMPP_WAVE - 1), wv = tid >> 6;
    const int W = c.grid_w, H = c.grid_h, N = W * H, WW = W >> 5, HW = H >> 5;
    const SceneIn& si = in[scene];
    SparseView<K> vr, vc;
    {
        DMPP_LDS unsigned char* p = (DMPP_LDS unsigned char*)smem_raw + kViewsAt;
        unsigned used = 0;
        if constexpr (K == 2) {
            vr.meta = p; vc.meta = p + (unsigned)H * 8u;
            vr.off2 = (DMPP_LDS uint16_t*)(p + (unsigned)(H + W) * 8u); vc.off2 = vr.off2 + H;
            used = (unsigned)(H + W) * 10u;
        } else {
            constexpr unsigned mb = (unsigned)sparse_meta_bytes_per_line<K>();
            vr.meta = p; vc.meta = p + (unsigned)H * mb;
            vr.off2 = nullptr; vc.off2 = nullptr;
            used = (unsigned)(H + W) * mb;
        }
        used = (used + 15u) & ~15u;
        vr.data = (DMPP_LDS uint32_t*)(p + used); vc.data = vr.data + budget;
        vr.LW = WW; vr.NL = H; vc.LW = HW; vc.NL = W;
    }
    for (int i = tid; i < LdsT::kClosedTab; i += (SW * DMPP_WAVE)) L.c_tab[i] = 0;
    for (int i = tid; i < kOpenCap; i += (SW * DMPP_WAVE)) L.o_f2[i] = 0xFFFFu;      // dead slots everywhere: the pop never range-checks
    const ObPoint* obs = obs_now + si.obs_off;
#ifdef DMPP_DEBUG_SEARCH
    long long tmark[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const int need = budget > 0 ? build_sparse_views<K, SW>(c, si, obs, si.obs_n, vr, vc, budget, L.s_wave, tmark) : 1;
#else
    const int need = budget > 0 ? build_sparse_views<K, SW>(c, si, obs, si.obs_n, vr, vc, budget, L.s_wave) : 1;
#endif
    const bool dense = need > budget;          // uniform over the block
    if (tid == 0) { if (budget > 0) atomicMax(need_max, need); overflow[scene] = dense ? 1 : 0; }
    uint32_t* bm = gbitmaps + (size_t)scene * 2 * (N >> 5);
    DMPP_LDS uint64_t* nz_row = (DMPP_LDS uint64_t*)((DMPP_LDS unsigned char*)smem_raw + kViewsAt);
    DMPP_LDS uint64_t* nz_col = nz_row + H;
    if (dense) { __syncthreads(); build_dense_views<SW>(c, si, obs, si.obs_n, bm, nz_row, nz_col); }
    const DenseView dr{ bm, nz_row, WW, H }, dc{ bm + (N >> 5), nz_col, HW, W };
    const int start = cell_of(c, si.grid_origin, si.loc.globalpoint.x, si.loc.globalpoint.y);
    const int goal = cell_of(c, si.grid_origin, si.goal.x, si.goal.y);
    const bool goal_blocked = dense ? cell_blocked(dr, goal % W, goal / W) : cell_blocked(vr, goal % W, goal / W);       // the same in every wave
    __syncthreads();
    if (!goal_blocked && tid == 0) {                                      // the vehicle is where it is
        const int sx = start % W, sy = start / W;
        if (dense) {
            bm[start >> 5] &= ~(1u << (start & 31));
            bm[(N >> 5) + sx * HW + (sy >> 5)] &= ~(1u << (sy & 31));
            __threadfence();
        } else {
            sparse_clear_cell(vr, sy, sx);
            sparse_clear_cell(vc, sx, sy);
        }
        L.o_ent[0] = (uint32_t)sx | ((uint32_t)sy << 12) | (8u << 24);
        L.o_f2[0] = (uint16_t)(hfun(sx, sy, goal % W, goal / W) >> 1);
        L.o_run[0] = 0;
    }
    __syncthreads();
    int32_t* path = paths + (size_t)scene * c.max_path;
    if (wv != 0) return;                       // set-up done: the search is wave 0's (k_search_spill: the others wait at its barrier for the workgroup's next scene)
    __builtin_amdgcn_s_setprio(1);             // a latency-bound wave: ahead of the set-up waves and of the kernels that run beside it
    const long long t_setup = clock64() - t_begin;
    SearchOut R; R.status = DMPP_G_GOAL_BLOCKED; R.n_exp = 0; R.n_push = 0; R.n_rounds = 0; R.path_cost = 0; R.path_len = 0; R.digest = 0;
    uint32_t* closed = gclosed + (size_t)scene * (N >> 5);
    uint16_t* pin = pinfo + (size_t)scene * N;
    int32_t* order = orders ? orders + (size_t)scene * order_cap : nullptr;
    uint2* ospill = ospill_all ? ospill_all + (size_t)scene * spill_cap : nullptr;
#ifdef DMPP_DEBUG_SEARCH
    long long dbg_t[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; int dbg_c[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (!goal_blocked) R = dense ? search_core<SPILL>(c, L, dr, dc, start, goal, order_cap, closed, pin, order, path, SPILL ? ospill : nullptr, lane, dbg_t, dbg_c)
                                 : search_core<SPILL>(c, L, vr, vc, start, goal, order_cap, closed, pin, order, path, SPILL ? ospill : nullptr, lane, dbg_t, dbg_c);
    if (lane == 0) {
        publish_debug(path, c.max_path, dbg_t, dbg_c, t_setup, clock64() - t_begin);
        int32_t* d2 = path + c.max_path - 32;      // set-up phases: entry->clear, (unused x2), pass 1, offsets, zero, pass 2, tail
        d2[0] = (int)(tmark[0] - t_begin); for (int i = 1; i < 7; i++) d2[i] = (int)(tmark[i] - tmark[i - 1]); d2[7] = (int)(t_begin + t_setup - tmark[6]);
    }
#else
    if (!goal_blocked) R = dense ? search_core<SPILL>(c, L, dr, dc, start, goal, order_cap, closed, pin, order, path, SPILL ? ospill : nullptr, lane)
                                 : search_core<SPILL>(c, L, vr, vc, start, goal, order_cap, closed, pin, order, path, SPILL ? ospill : nullptr, lane);
#endif
    if (!SPILL && __builtin_expect(R.status == DMPP_G_OVERFLOW && retry_cnt != nullptr && c.bucket_cap > kOpenCap, 0)) {
        // more live open-list entries than LDS holds, and the specification allows more: the scene goes on the list of
        // k_search_spill, which follows on the stream (retry_cnt is null when the handle has no spill area)
        if (lane == 0) retry_list[atomicAdd(retry_cnt, 1)] = scene;
    }
    (void)t_setup;
    if (lane == 0) {
        cost_out[scene] = (int32_t)min((clock64() - t_begin) >> kOrderShift, (long long)(kOrderClasses - 1));   // launch-order key of the next tick (k_order)
        publish_search(gout[scene], R, start, goal);
    }
}

#ifdef DMPP_SEARCH_VGPR          // experiment: cap the registers of the search kernel (occupancy while the set-up waves are resident)
#define DMPP_SEARCH_ATTR __attribute__((amdgpu_waves_per_eu(DMPP_SEARCH_VGPR, DMPP_SEARCH_VGPR)))
#else
#define DMPP_SEARCH_ATTR
#endif
template <int K, int SW>
__global__ void __launch_bounds__(SW * DMPP_WAVE) DMPP_SEARCH_ATTR
k_search(PlannerConfig c, int n_scenes, int order_cap, int budget, const int32_t* __restrict__ perm, const SceneIn* __restrict__ in,
         const ObPoint* __restrict__ obs_now, uint32_t* __restrict__ gclosed, uint16_t* __restrict__ pinfo, int32_t* __restrict__ orders,
         int32_t* __restrict__ paths, GridOut* __restrict__ gout, uint32_t* __restrict__ gbitmaps, int32_t* __restrict__ cost_out,
         int32_t* __restrict__ overflow, int32_t* __restrict__ need_max, uint2* __restrict__ ospill_all, int spill_cap,
         int32_t* __restrict__ retry_list, int32_t* __restrict__ retry_cnt)
{
    // static LDS: SearchLds; dynamic LDS: [line metas of both views | `budget` data words per view]
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ SearchLds<closed_log_of<K>()> L;
    if ((int)blockIdx.x >= n_scenes) return;
    const int scene = perm ? perm[blockIdx.x] : (int)blockIdx.x;      // heaviest scenes first (k_order) when they do not all fit at once
    search_scene<K, SW, false>(c, scene, order_cap, budget, in, obs_now, gclosed, pinfo, orders, paths, gout, gbitmaps, cost_out, overflow, need_max,
                               ospill_all, spill_cap, retry_list, retry_cnt, L, smem_raw);
}

// The scenes the search kernel put on the retry list (their open list outgrew LDS), once more with the spill area.  Launched
// behind every k_search of a handle that has a spill area, with TWO workgroups that take the list's scenes in turn (no generated
// scene ever comes here: they find the list empty and leave - but each has to wait for LDS behind the searching workgroups of
// the neighbouring ticks first, which is why there are so few of them: one per scene cost 25 - 130 us per tick).
// Measured and not kept: both instances of the search loop in one kernel (104 -> 132 VGPRs, twice the scalar spills in the
// common loop: -5 %), and the second attempt as a call from the kernel's tail (scratch + 142 VGPRs: -10 %).
template <int K>
__global__ void __launch_bounds__(kSearchBlock)
k_search_spill(PlannerConfig c, int n_scenes, int order_cap, int budget, const SceneIn* __restrict__ in,
         const ObPoint* __restrict__ obs_now, uint32_t* __restrict__ gclosed, uint16_t* __restrict__ pinfo, int32_t* __restrict__ orders,
         int32_t* __restrict__ paths, GridOut* __restrict__ gout, uint32_t* __restrict__ gbitmaps, int32_t* __restrict__ cost_out,
         int32_t* __restrict__ overflow, int32_t* __restrict__ need_max, uint2* __restrict__ ospill_all, int spill_cap,
         const int32_t* __restrict__ retry_list, const int32_t* __restrict__ retry_cnt)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ SearchLds<closed_log_of<K>()> L;
    const int cnt = min(*retry_cnt, n_scenes);
    for (int i = (int)blockIdx.x; i < cnt; i += (int)gridDim.x) {      // (a handful of workgroups: nearly always there is nothing to do, and they must not queue for LDS behind the searches)
        const int scene = retry_list[i];
        search_scene<K, kSearchSetupWaves, true>(c, scene, order_cap, budget, in, obs_now, gclosed, pinfo, orders, paths, gout, gbitmaps, cost_out, overflow,
                                                 need_max, ospill_all, spill_cap, nullptr, nullptr, L, smem_raw);
        __syncthreads();                       // every wave is done with this scene's LDS
    }
}

// pp_get_grid: one scene's occupancy grid as bytes, produced by the SAME footprint code the search uses (so the tests see
// what the search sees): the workgroup builds the sparse views and expands the row-major one; the column-major one is
// checked against it cell by cell (mismatches counted in bad[0]).  A scene that does not fit the LDS given is written span
// by span instead (bad[1] = 1: no cross-check of the two views then).
template <int K>
__global__ void __launch_bounds__(kSearchBlock)
k_export_grid(PlannerConfig c, int scene, int budget, const SceneIn* __restrict__ in, const ObPoint* __restrict__ obs_now,
              uint8_t* __restrict__ out, int* __restrict__ bad)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int s_wave[kSearchSetupWaves + 1];
    const int tid = threadIdx.x;
    const int W = c.grid_w, H = c.grid_h, WW = W >> 5, HW = H >> 5;
    const SceneIn& si = in[scene];
    SparseView<K> vr, vc;
    {
        DMPP_LDS unsigned char* p = (DMPP_LDS unsigned char*)smem_raw;
        unsigned used = 0;
        if constexpr (K == 2) {
            vr.meta = p; vc.meta = p + (unsigned)H * 8u;
            vr.off2 = (DMPP_LDS uint16_t*)(p + (unsigned)(H + W) * 8u); vc.off2 = vr.off2 + H;
            used = (unsigned)(H + W) * 10u;
        } else {
            constexpr unsigned mb = (unsigned)sparse_meta_bytes_per_line<K>();
            vr.meta = p; vc.meta = p + (unsigned)H * mb;
            vr.off2 = nullptr; vc.off2 = nullptr;
            used = (unsigned)(H + W) * mb;
        }
        used = (used + 15u) & ~15u;
        vr.data = (DMPP_LDS uint32_t*)(p + used); vc.data = vr.data + budget;
        vr.LW = WW; vr.NL = H; vc.LW = HW; vc.NL = W;
    }
    const ObPoint* obs = obs_now + si.obs_off;
    const int need = build_sparse_views<K, kSearchSetupWaves>(c, si, obs, si.obs_n, vr, vc, budget, s_wave);
    if (need > budget) {
        if (tid == 0) bad[1] = 1;
        for (int cell = tid; cell < W * H; cell += kSearchBlock) out[cell] = 0;
        __threadfence();
        __syncthreads();
        for_each_span<kSearchSetupWaves>(c, si, obs, si.obs_n, [&](bool colhalf, int l, int a, int b) {
            if (!colhalf) for (int x = a; x <= b; x++) out[l * W + x] = 1;
        });
        return;
    }
    int mism = 0;
    for (int cell = tid; cell < W * H; cell += kSearchBlock) {
        const int x = cell % W, y = cell / W;
        const auto mr = vr.line(y); const auto mc = vc.line(x);
        const uint32_t a = (vr.word(mr, x >> 5) >> (x & 31)) & 1u, b = (vc.word(mc, y >> 5) >> (y & 31)) & 1u;
        out[cell] = (uint8_t)a;
        mism += a != b;
    }
    if (mism) atomicAdd(&bad[0], mism);
}

}  // namespace dmpp
