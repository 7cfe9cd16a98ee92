// kernels_g.hpp — HIP kernels of the grid engine (specification in
// DESIGN.md §5 and oracle/dmpp_grid_oracle.c — the reference has no grid code).
//
//   k_order     : launch order of the search, heaviest scenes first (counting sort on the previous search times).
//   k_score     : lattice candidates (cubic Beziers to laterally shifted terminals + the grid
//                 path) scored on collision / curvature / progress, 4 (or 16) waves per scene.
// The search itself, and the rasterisation (G1) it does for itself, is kernels_s.hpp.
#pragma once
#include "dev_geom.hpp"
#include "kernels_s.hpp"
#include "kernels_score.hpp"

namespace dmpp {

// Launch order of the scenes of k_search: heaviest first, by the time the scene's search took on the previous tick (in
// units of 8 Ki cycles, written by k_search; it changes little from tick to tick).  One wave per scene and two waves per CU means the kernel ends with its
// slowest scene; starting that scene first keeps it off the tail.  Counting sort into 1024 cost classes, one block.
constexpr int kOrderBlock = 1024;     // kOrderClasses classes of 8 Ki cycles (kOrderShift), up to 8.4 M cycles: kernels_s.hpp
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

// ---------------------------------------------------------------------------------------
// G3.  The kernel follows its tick's search on the stream, so it also hands that search's LDS need to the host (a 4-byte
// store into pinned memory: no copy command on the stream) and clears the counter for the next search that uses it.
template <int NW>
__global__ void __launch_bounds__(NW * DMPP_WAVE)
k_score(PlannerConfig c, int n_scenes, const SceneIn* __restrict__ in, const ObPoint* __restrict__ obs_now,
        const int32_t* __restrict__ paths, GridOut* __restrict__ gout, int32_t* __restrict__ need_dev, int32_t* __restrict__ need_host)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ScoreShared<NW>& sh = *reinterpret_cast<ScoreShared<NW>*>(smem_raw);
    const int scene = blockIdx.x;
    if (scene == 0 && threadIdx.x == 0 && need_dev) {
        const int32_t v = need_dev[0];
        need_dev[0] = 0;
        need_dev[1] = 0;                       // the retry count of k_search / k_search_spill (same buffer set, three ticks on)
        if (need_host) *need_host = v;
    }
    if (scene >= n_scenes) return;
#ifdef DMPP_SCORE_PRIO                         // experiment knob: wave issue priority of the scoring pass (default 0)
    __builtin_amdgcn_s_setprio(DMPP_SCORE_PRIO);
#endif
    const SceneIn& si = in[scene];
    GridOut& go = gout[scene];
    score_body<NW>(c, si, obs_now + si.obs_off, si.obs_n, paths + (size_t)scene * c.max_path, go, go.status, go.path_len, sh);
}

}  // namespace dmpp
