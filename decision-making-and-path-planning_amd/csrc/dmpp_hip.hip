// dmpp_hip.hip — the C-ABI of include/dmpp_planner.h over the HIP kernels (gfx950 only).
//
// One handle = one device and all device buffers.  Streams: the handle's stream (copies, stand-alone operators, the whole
// tick of a small batch), kBuf search streams (stream_m[0] is the handle's stream: the searches of consecutive ticks run
// side by side, each preceded by its launch order and followed by its own scoring pass), the FRONT stream stream_r (obstacle
// snapshot, Decision, Planning: highest priority, running ahead of the searches), stream_s (scoring on its own stream: a measurement knob), and - once streamed
// ticks are in use (pp_update_async / pp_fetch_async) - one upload and two download streams.  pp_plan_tick describes the
// launch order and the events between the chains; batches below pipeline_min scenes run on the handle's stream with only
// Decision + Planning forked beside the grid engine.  pp_set_* / pp_get_* join the chains first (join_all) and wait on
// the host; the streamed calls never wait on the host (pp_wait_tick waits for one tick's downloads only).
// Nothing here computes planning results on the host; without a GPU pp_create fails.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>
#include <cstdlib>
#include <vector>
#include <new>
#include <algorithm>
#include "../../include/dmpp_planner.h"
#include "kernels_r.hpp"
#include "kernels_g.hpp"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(PP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
    } while (0)

struct EvPair { hipEvent_t a, b; int k; };

// Buffers that a tick's search and scoring touch exist kBuf times, used round-robin ("parity"): up to kBuf searches of
// consecutive ticks are in flight at once, each on its own stream, and scoring of tick t never holds up the search of t + 1.
#ifndef DMPP_KBUF
#define DMPP_KBUF 3
#endif
constexpr int kBuf = DMPP_KBUF;
// The obstacle snapshot exists 2 * kBuf times: the front chain of tick t writes its snapshot while the scoring pass of tick
// t - kBuf still reads its own, so that chain need not wait for that pass (only for the search before it, see pp_plan_tick).
constexpr int kObs = 2 * kBuf;
// Streamed ticks (pp_update_async / pp_fetch_async): the per-tick inputs - SceneIn records, obstacle pool, motion pool - exist
// kIn times.  An update is copied into the set after the current one while the ticks in flight still read theirs (a search
// runs up to kBuf ticks behind the front chain of its tick); a set is written again only after every kernel of the ticks that
// read it (TickRec) has finished.  PlanOut exists kPlan times, so that Planning(t + 1) does not wait for the download of tick t.
constexpr int kIn = 12;
constexpr int kPlan = 12;
constexpr int kGout = 12;          // GridOut sets: a download is issued when its scoring pass has finished, up to ~8 ticks behind the newest front chain
constexpr int kDone = 32;          // ticks whose downloads pp_wait_tick can still name

struct InputSet {
    SceneIn* d_in = nullptr; ObPoint* d_obs = nullptr; ObMotion* d_mot = nullptr;
    bool have_motion = false; int n_obs_total = 0;
    hipEvent_t ev_up = nullptr; bool up_recorded = false;
};
// one tick in flight: which input set it reads, and the events that close its front chain and its last kernel
struct TickRec { long long tick; int in_set; hipEvent_t ev_front, ev_tail; };
// A download asked for by pp_fetch_async.  Its copies are ISSUED only once the kernels they follow have finished (polled by
// every streamed call; forced by pp_wait_tick and by the tick that is about to overwrite the device set): a copy command
// enqueued behind an unfinished dependency sits at the head of its hardware queue / DMA queue as a blocked barrier, and
// streams that share that queue - there are more HIP streams here than hardware queues - stall behind it (measured: the
// front chain's kernels took 2 - 6 times as long with the GridOut copy of each tick enqueued three ticks ahead of its
// scoring pass).
struct PendingFetch {
    long long tick; int slot;
    PlanOut* plan_dst; const PlanOut* plan_src; int plan_set; bool plan_issued;
    PlanningOut* res_dst; PlanningStatus* show_dst;      // pp_fetch_published_async: only what the reference publishes (strided copies out of PlanOut)
    GridOut* grid_dst; const GridOut* grid_src; int grid_set; bool grid_issued;
    hipEvent_t ev_front, ev_tail;      // own references: taken out of the tick's TickRec so that they are not recycled early
    size_t n;
};

}  // namespace

struct pp_planner {
    PlannerConfig cfg;
    PlannerCaps caps;
    int device = 0;
    int n_scenes = 0;
    hipStream_t stream = nullptr;
    // inputs
    SceneIn* d_in = nullptr; GlobalPoint3D* d_lane = nullptr; uint8_t* d_attr = nullptr; bool have_attr = false; GlobalPoint2D* d_ref = nullptr;
    ObPoint* d_obs = nullptr; ObMotion* d_mot = nullptr; ObPoint* d_obs_now[kObs] = {};
    bool have_motion = false;
    // state / outputs
    SceneState* d_state = nullptr; PlanOut* d_plan = nullptr; GridOut* d_gout[kGout] = {}; int gout_set = 0;   // GridOut: kGout sets (a download of tick t must not hold up the search of tick t + kBuf); gout_set: the last grid tick's
    GlobalPoint2D* d_dec_ref = nullptr;
    // grid engine
    uint8_t* d_grid = nullptr; uint16_t* d_pinfo[kBuf] = {}; uint32_t* d_closed[kBuf] = {};
    int32_t* d_order[kBuf] = {}; int32_t* d_path[kObs] = {}; uint32_t* d_gbm[kBuf] = {};     // d_path, d_need: per snapshot set (the scoring pass of tick t reads them beside the search of tick t + kBuf)
    int path_set = 0;                                  // the set of the last tick with the grid stage
    int32_t* d_perm[kBuf] = {}; int32_t* d_cost[kBuf] = {};
    uint2* d_ospill[kBuf] = {}; int spill_cap = 0; int32_t* d_retry[kBuf] = {};     // open-list spill areas (bucket_cap0 entries per scene; none when bucket_cap0 <= the LDS list)
    // Long searches (many obstacles, large grids) overlap their tails: the searches of odd ticks run on a second stream, and
    // every buffer a search touches exists per tick parity
    hipStream_t stream_m[kBuf] = {}; int n_obs_total = 0; int overlap_override = -1;     // stream_m[0] is the handle's stream int overlap_override = -1;
    size_t grid_cells = 0;       // per scene, at creation
    int bucket_cap0 = 0, max_path0 = 0;
    // search: k_search_lds<kind> with `lds_budget` data words per view in LDS; scenes that need more go to k_search_gbm
    int search_kind = 0; int search_meta_bytes = 0; int lds_budget = 0, lds_budget_max = 0; bool lds_budget_fixed = false, search_force_gbm = false, budget_from_need = false;
    int gbm_lds = 0; int search_slots = 512; size_t search_static_lds = 0;   // static LDS of k_search<kind>
    int32_t* d_ovf[kBuf] = {}; int32_t* d_need[kObs] = {}; int32_t* h_need = nullptr;   // h_need: pinned, [kObs], written by k_score (-1: nothing yet)
    int need_seen = 0;
    int* d_gridbad = nullptr;

    hipStream_t stream_r = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the R kernels run beside the grid engine
    // k_score of tick t runs on its own stream beside the rasterise / search of tick t+1: the obstacle snapshot, the path
    // cells and GridOut are double-buffered by tick parity; ev_score[p] = the last k_score that used the buffers p
    hipStream_t stream_s = nullptr; hipEvent_t ev_search[kBuf] = {}, ev_score[kObs] = {};      // ev_score: per snapshot set
    hipEvent_t ev_raster = nullptr;
    bool score_recorded[kObs] = {}, search_recorded[kBuf] = {}, front_recorded = false, front_unjoined = false;
    int parity = 0;              // buffers of the last tick
    bool last_piped = false;     // the last tick ran as three chains on several streams (else it ended on the handle's stream)
    int obs_set = 0;             // obstacle snapshot of the last tick (d_obs_now[obs_set])
    bool score_own_stream = false;   // env DMPP_SCORE_STREAM=1 | 2 (measurement knob): k_score on one stream of its own, lowest | highest queue priority
    bool score_stream_high = false;
    bool front_wait = false;         // env DMPP_FRONT_WAIT=1 (measurement knob): the front chain waits for the search kBuf ticks back, as in rounds 1 - 2
    int n_cus = 256;
    int pipeline_min = 256;      // batches at least this large run the three chains on three streams (env DMPP_PIPELINE_MIN)
    bool r_on_main = false;      // the last tick ran Decision + Planning on the handle's stream (grid stage off)
    int n_lane_pts = 0, n_ref_pts = 0;   // pool sizes of the resident scenes (slice validation)
    int* d_bad = nullptr;        // k_validate_scenes: scenes with a slice outside its pool
    // map store (pp_set_map): lane / junction tables; the point pools are d_lane / d_attr / d_ref
    int32_t* d_map_first = nullptr; MapLane* d_map_lanes = nullptr; uint16_t* d_map_width = nullptr; MapJunction* d_map_junc = nullptr;
    int* d_map_bad = nullptr; int map_roads = 0, map_lanes = 0, map_junctions = 0; bool have_map = false;
    // op scratch (stand-alone operators)
    void* d_scratch = nullptr; size_t scratch_bytes = 0;
    // streamed ticks (allocated by the first pp_update_async / pp_fetch_async)
    bool streaming = false;
    InputSet in_sets[kIn]; int in_cur = 0, in_staged = -1;       // d_in / d_obs / d_mot / have_motion / n_obs_total alias in_sets[in_cur]
    int resident_mode = 0;       // 0: scenes with their own slices (pp_set_scenes), 1: egos on the resident map (pp_set_egos)
    PlanOut* d_plan_ring[kPlan] = {}; int plan_cur = 0;          // d_plan aliases d_plan_ring[plan_cur]
    hipStream_t stream_up = nullptr, stream_dp = nullptr, stream_dg = nullptr;   // upload; download of PlanOut; download of GridOut
    hipEvent_t ev_fetched_plan[kPlan] = {}, ev_fetched_grid[kGout] = {}; bool fetched_plan_rec[kPlan] = {}, fetched_grid_rec[kGout] = {};
    hipEvent_t ev_done_p[kDone] = {}, ev_done_g[kDone] = {}; long long done_tick[kDone]; bool done_p_rec[kDone] = {}, done_g_rec[kDone] = {};
    int32_t* h_bad = nullptr;    // pinned, [kDone]: poisoned scenes of the tick (copied down with its PlanOut)
    long long tick_seq = 0;      // ticks enqueued so far on this handle (the id of the last one)
    TickRec last_rec = { -1, 0, nullptr, nullptr };
    std::vector<TickRec> inflight; std::vector<hipEvent_t> sync_events;   // sync_events: a pool of timing-disabled events
    std::vector<PendingFetch> fetches;
    // profiling
    int profile = 0;             // 0 off, 1 every kernel of a tick between HIP events, 2 only the search kernel
    std::vector<EvPair> pending; std::vector<hipEvent_t> free_events;
    float k_ms[PP_K_COUNT] = {0}; int k_launches[PP_K_COUNT] = {0};
};

namespace {

template <class T> int dmalloc(T** p, size_t count)
{
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return PP_OK;
}

int check_cfg(const PlannerConfig* c)
{
    if (!c) return fail(PP_ERR_ARG, "config is null");
    if (c->grid_stage) {
        if (c->grid_w <= 0 || c->grid_h <= 0 || (c->grid_w % 32) != 0 || (c->grid_h % 32) != 0)
            return fail(PP_ERR_ARG, "grid_w and grid_h must be positive multiples of 32");
        if (c->grid_w > 2048 || c->grid_h > 2048) return fail(PP_ERR_ARG, "grids above 2048x2048 are not supported (one wave scans 64 words of a line)");
        if ((long long)c->grid_w * c->grid_h > (1ll << 24)) return fail(PP_ERR_ARG, "grid larger than 2^24 cells (cell index is 24 bits in an open-set entry)");
        if (c->bucket_cap < 16 || c->max_path < 2) return fail(PP_ERR_ARG, "bucket_cap/max_path too small");
        if (!(c->cell > 0)) return fail(PP_ERR_ARG, "cell size must be positive");
        if (c->n_lattice < 0 || c->n_lattice > DMPP_MAX_LATTICE - 1) return fail(PP_ERR_ARG, "n_lattice out of range");
    }
    return PP_OK;
}

hipEvent_t get_event(pp_planner* h)
{
    if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// timing-disabled events for cross-stream ordering, pooled
hipEvent_t get_sync_event(pp_planner* h)
{
    if (!h->sync_events.empty()) { hipEvent_t e = h->sync_events.back(); h->sync_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

// h->d_in / d_obs / d_mot / have_motion / n_obs_total are the input set the next tick reads
void adopt_input_set(pp_planner* h, int s)
{
    const InputSet& I = h->in_sets[s];
    h->in_cur = s; h->d_in = I.d_in; h->d_obs = I.d_obs; h->d_mot = I.d_mot; h->have_motion = I.have_motion; h->n_obs_total = I.n_obs_total;
}
void note_current_set(pp_planner* h)          // after a pp_set_* call changed what the current set holds
{
    InputSet& I = h->in_sets[h->in_cur];
    I.have_motion = h->have_motion; I.n_obs_total = h->n_obs_total;
    h->in_staged = -1;                        // an update staged before it is superseded
}

// Issues the copies of the pending downloads whose kernels have finished.  force_tick: that tick's copies are issued whatever
// the state of its kernels (behind a stream wait); force_plan_set / force_grid_set: likewise the copies that read that
// PlanOut / GridOut set (the next tick is about to overwrite it).
int pump_fetches(pp_planner* h, long long force_tick = -1, int force_plan_set = -1, int force_grid_set = -1)
{
    for (PendingFetch& f : h->fetches) {
        if ((f.plan_dst || f.res_dst || f.show_dst) && !f.plan_issued) {
            const bool ready = hipEventQuery(f.ev_front) == hipSuccess;
            if (ready || f.tick == force_tick || f.plan_set == force_plan_set) {
                (void)hipGetLastError();
                if (!ready) HIP_TRY(hipStreamWaitEvent(h->stream_dp, f.ev_front, 0));
                if (f.plan_dst) HIP_TRY(hipMemcpyAsync(f.plan_dst, f.plan_src, f.n * sizeof(PlanOut), hipMemcpyDefault, h->stream_dp));
                const char* base = reinterpret_cast<const char*>(f.plan_src);
                if (f.res_dst) HIP_TRY(hipMemcpy2DAsync(f.res_dst, sizeof(PlanningOut), base + offsetof(PlanOut, result), sizeof(PlanOut), sizeof(PlanningOut), f.n, hipMemcpyDefault, h->stream_dp));
                if (f.show_dst) HIP_TRY(hipMemcpy2DAsync(f.show_dst, sizeof(PlanningStatus), base + offsetof(PlanOut, show), sizeof(PlanOut), sizeof(PlanningStatus), f.n, hipMemcpyDefault, h->stream_dp));
                HIP_TRY(hipEventRecord(h->ev_fetched_plan[f.plan_set], h->stream_dp)); h->fetched_plan_rec[f.plan_set] = true;
                HIP_TRY(hipEventRecord(h->ev_done_p[f.slot], h->stream_dp));
                f.plan_issued = true;
            }
        }
        if (f.grid_dst && !f.grid_issued) {
            hipEvent_t dep = f.ev_tail ? f.ev_tail : f.ev_front;
            const bool ready = hipEventQuery(dep) == hipSuccess;
            if (ready || f.tick == force_tick || f.grid_set == force_grid_set) {
                (void)hipGetLastError();
                if (!ready) HIP_TRY(hipStreamWaitEvent(h->stream_dg, dep, 0));
                HIP_TRY(hipMemcpyAsync(f.grid_dst, f.grid_src, f.n * sizeof(GridOut), hipMemcpyDefault, h->stream_dg));
                HIP_TRY(hipEventRecord(h->ev_fetched_grid[f.grid_set], h->stream_dg)); h->fetched_grid_rec[f.grid_set] = true;
                HIP_TRY(hipEventRecord(h->ev_done_g[f.slot], h->stream_dg));
                f.grid_issued = true;
            }
        }
    }
    (void)hipGetLastError();                  // hipErrorNotReady is not an error here
    size_t k = 0;                             // completed requests leave from the front (prune_inflight looks at the first one left)
    while (k < h->fetches.size() && ((!h->fetches[k].plan_dst && !h->fetches[k].res_dst && !h->fetches[k].show_dst) || h->fetches[k].plan_issued) &&
           (!h->fetches[k].grid_dst || h->fetches[k].grid_issued)) k++;
    h->fetches.erase(h->fetches.begin(), h->fetches.begin() + (long)k);
    return PP_OK;
}

// ticks whose kernels have all finished leave the in-flight list (oldest first; the list stays short: a tick's front chain
// waits for the scoring pass 2 * kBuf ticks before it).  Their events go back to the pool: last_rec may still name the
// events of the last tick, which are recorded again by the next tick at the earliest - and then last_rec names that one.
int prune_inflight(pp_planner* h)
{
    size_t k = 0;
    while (k < h->inflight.size()) {
        const TickRec& r = h->inflight[k];
        if (!h->fetches.empty() && r.tick >= h->fetches.front().tick) break;      // its events are still named by a download not issued yet
        if (hipEventQuery(r.ev_front) != hipSuccess || (r.ev_tail && hipEventQuery(r.ev_tail) != hipSuccess)) break;
        h->sync_events.push_back(r.ev_front); if (r.ev_tail) h->sync_events.push_back(r.ev_tail);
        k++;
    }
    (void)hipGetLastError();                  // hipErrorNotReady is not an error here
    h->inflight.erase(h->inflight.begin(), h->inflight.begin() + (long)k);
    if (h->inflight.size() > 256) {           // a caller that never lets the device catch up: wait for the oldest
        HIP_TRY(hipEventSynchronize(h->inflight.front().ev_front));
        if (h->inflight.front().ev_tail) HIP_TRY(hipEventSynchronize(h->inflight.front().ev_tail));
    }
    return PP_OK;
}

constexpr int kScoreWideMaxScenes = 128;     // up to here k_score runs 16 waves per scene (one scene per CU at most)

// Everything the ticks enqueued so far started - on any of the four streams - is ordered before whatever the handle's
// stream does next: ev_score[q] closes the raster -> search -> score chain of the last tick of parity q, ev_join the
// Decision -> Planning chain (stream order covers the earlier ticks).  No host wait.
int join_all(pp_planner* h)
{
    for (int q = 0; q < kObs; q++) if (h->score_recorded[q]) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_score[q], 0));
    if (h->front_recorded) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return PP_OK;
}

struct Timed {
    pp_planner* h; int k; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on = false;
    Timed(pp_planner* h_, int k_, hipStream_t st_ = nullptr) : h(h_), k(k_), st(st_ ? st_ : h_->stream) {
        on = h->profile == 1 || (h->profile == 2 && k == PP_K_SEARCH);
        if (on) { a = get_event(h); b = get_event(h); if (a) (void)hipEventRecord(a, st); }
    }
    ~Timed() {
        if (on && a && b) { (void)hipEventRecord(b, st); h->pending.push_back({a, b, k}); }
    }
};

int drain_events(pp_planner* h)
{
    if (h->pending.empty()) return PP_OK;
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& p : h->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { h->k_ms[p.k] += ms; h->k_launches[p.k] += 1; }
        h->free_events.push_back(p.a); h->free_events.push_back(p.b);
    }
    h->pending.clear();
    return PP_OK;
}

int setup_grid_launch(pp_planner* h)
{
    const PlannerConfig& c = h->cfg;
    if (!c.grid_stage) return PP_OK;
    const size_t N = (size_t)c.grid_w * c.grid_h;
    // search: per-line metas of both sparse views + the budgeted data words in dynamic LDS (kernels_s.hpp)
    const int lw = (int)std::max((size_t)c.grid_w / 32, (size_t)c.grid_h / 32);
    h->search_kind = lw <= 16 ? 0 : (lw <= 32 ? 1 : 2);
    const size_t per_line = h->search_kind == 0 ? 4 : (h->search_kind == 1 ? 8 : 10);
    h->search_meta_bytes = (int)((((size_t)c.grid_w + c.grid_h) * per_line + 15) & ~(size_t)15);
    h->search_static_lds = h->search_kind == 2 ? sizeof(dmpp::SearchLds<dmpp::closed_log_of<2>()>) : sizeof(dmpp::SearchLds<dmpp::closed_log_of<0>()>);
    static_assert(dmpp::closed_log_of<0>() == dmpp::closed_log_of<1>(), "one static LDS size for the kinds 0 and 1");
    const size_t static_lds = h->search_static_lds + 64;
    size_t lds_max = 64u * 1024u;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) == hipSuccess && (size_t)v > lds_max) lds_max = (size_t)v;
    }
    if (lds_max > 160u * 1024u) lds_max = 160u * 1024u;
    h->gbm_lds = (int)(((size_t)c.grid_w + c.grid_h) * 8);
    if (std::max((size_t)h->search_meta_bytes + 2 * 64 * 4, (size_t)h->gbm_lds) + static_lds > lds_max)
        return fail(PP_ERR_CAPACITY, "grid too large for the search kernel's LDS tables");
    {
        const size_t room = (lds_max - static_lds - (size_t)h->search_meta_bytes) / 8;        // data words per view that fit at all
        const size_t dense = N / 32;                                                            // ... that a view can ever need
        h->lds_budget_max = (int)std::min(room, dense);
        const void* fns[3] = { reinterpret_cast<const void*>(&dmpp::k_search<0, dmpp::kSearchSetupWaves>), reinterpret_cast<const void*>(&dmpp::k_search<1, dmpp::kSearchSetupWaves>),
                               reinterpret_cast<const void*>(&dmpp::k_search<2, dmpp::kSearchSetupWaves>) };
        const void* fnw[3] = { reinterpret_cast<const void*>(&dmpp::k_search<0, dmpp::kSearchSetupWavesWide>), reinterpret_cast<const void*>(&dmpp::k_search<1, dmpp::kSearchSetupWavesWide>),
                               reinterpret_cast<const void*>(&dmpp::k_search<2, dmpp::kSearchSetupWavesWide>) };
        const void* fnr[3] = { reinterpret_cast<const void*>(&dmpp::k_search_spill<0>), reinterpret_cast<const void*>(&dmpp::k_search_spill<1>),
                               reinterpret_cast<const void*>(&dmpp::k_search_spill<2>) };
        const void* fne[3] = { reinterpret_cast<const void*>(&dmpp::k_export_grid<0>), reinterpret_cast<const void*>(&dmpp::k_export_grid<1>),
                               reinterpret_cast<const void*>(&dmpp::k_export_grid<2>) };
        const int dyn_max = std::max(h->search_meta_bytes + 8 * h->lds_budget_max, h->gbm_lds);
        if (dyn_max + (int)static_lds > 48 * 1024) {
            if (hipFuncSetAttribute(fns[h->search_kind], hipFuncAttributeMaxDynamicSharedMemorySize, dyn_max) != hipSuccess ||
                hipFuncSetAttribute(fnw[h->search_kind], hipFuncAttributeMaxDynamicSharedMemorySize, dyn_max) != hipSuccess ||
                hipFuncSetAttribute(fnr[h->search_kind], hipFuncAttributeMaxDynamicSharedMemorySize, dyn_max) != hipSuccess ||
                hipFuncSetAttribute(fne[h->search_kind], hipFuncAttributeMaxDynamicSharedMemorySize, dyn_max) != hipSuccess) {
                (void)hipGetLastError();
                h->lds_budget_max = (int)std::min((size_t)h->lds_budget_max, (48u * 1024u - static_lds - (size_t)h->search_meta_bytes) / 8);
            }
        }
    }
    if (const char* e = std::getenv("DMPP_SEARCH_GBM")) h->search_force_gbm = std::atoi(e) != 0;          // test / measurement knob: the dense form in HBM for every scene
    if (const char* e = std::getenv("DMPP_LDS_BUDGET")) {                                                // ... a fixed budget (words per view)
        h->lds_budget = std::max(1, std::min(std::atoi(e), h->lds_budget_max)); h->lds_budget_fixed = true;
    } else h->lds_budget = 0;                                                                            // chosen at the first tick (obstacle density), then adaptive
    for (int q = 0; q < kBuf; q++) {
        if (!h->d_ovf[q]) {
            int r = dmalloc(&h->d_ovf[q], (size_t)h->caps.max_scenes); if (r) return r;
            HIP_TRY(hipMemsetAsync(h->d_ovf[q], 0, (size_t)h->caps.max_scenes * sizeof(int32_t), h->stream));
        }
        if (!h->d_retry[q]) { int r = dmalloc(&h->d_retry[q], (size_t)h->caps.max_scenes); if (r) return r; }
    }
    for (int q = 0; q < kObs; q++)
        if (!h->d_need[q]) { int r = dmalloc(&h->d_need[q], (size_t)2); if (r) return r; HIP_TRY(hipMemsetAsync(h->d_need[q], 0, 2 * sizeof(int32_t), h->stream)); }   // [0] LDS need of the search, [1] its retry count
    if (!h->h_need) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_need), kObs * sizeof(int32_t), hipHostMallocDefault));
    for (int q = 0; q < kObs; q++) h->h_need[q] = -1;       // (a new configuration: what earlier searches needed says nothing; no search is in flight here)
    h->budget_from_need = false;
    if (!h->d_gridbad) { int r = dmalloc(&h->d_gridbad, (size_t)2); if (r) return r; }
    if (sizeof(dmpp::ScoreShared<16>) > 48u * 1024u)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dmpp::k_score<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(dmpp::ScoreShared<16>));
    for (int q = 0; q < kBuf; q++) {
        if (!h->d_perm[q]) { int r = dmalloc(&h->d_perm[q], (size_t)h->caps.max_scenes); if (r) return r; }
        if (!h->d_cost[q]) {
            int r = dmalloc(&h->d_cost[q], (size_t)h->caps.max_scenes); if (r) return r;
            HIP_TRY(hipMemsetAsync(h->d_cost[q], 0, (size_t)h->caps.max_scenes * sizeof(int32_t), h->stream));
        }
    }
    for (int q = 0; q < kBuf; q++) if (!h->d_gbm[q]) {   // the dense form of the two views (row- then column-major), only written by the scenes that do not fit the LDS budget
        int r = dmalloc(&h->d_gbm[q], (size_t)h->caps.max_scenes * 2 * (h->grid_cells / 32));
        if (r) return r;
    }
    return PP_OK;
}

}  // namespace

extern "C" {

const char* pp_last_error(void) { return g_err.c_str(); }

int pp_create(const PlannerConfig* cfg, int device, const PlannerCaps* caps, pp_handle* out)
{
    if (!out || !caps) return fail(PP_ERR_ARG, "null argument");
    *out = nullptr;
    int r = check_cfg(cfg); if (r) return r;
    if (caps->max_scenes <= 0) return fail(PP_ERR_ARG, "max_scenes must be positive");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(PP_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(PP_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    pp_planner* h = new (std::nothrow) pp_planner();
    if (!h) return fail(PP_ERR_HIP, "out of host memory");
    h->cfg = *cfg; h->caps = *caps; h->device = device;
    if (const char* e = std::getenv("DMPP_PIPELINE_MIN")) h->pipeline_min = std::atoi(e);      // tuning knob: 0 = always, large = never
    auto bail = [&](int code) { pp_destroy(h); return code; };
    // Queue priorities.  The FRONT chain (obstacle snapshot, Decision, Planning) is a short serial chain that every tick's
    // search waits for: it gets the highest dispatch priority (and its waves raise their issue priority, s_setprio).  The
    // searches are the bulk of the work and overlap each other: normal priority.  Scoring fills what is left: lowest.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio_normal = (prio_least + prio_greatest) / 2;
    if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_normal) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipStreamCreate failed"));
    h->stream_m[0] = h->stream;
    for (int q = 1; q < kBuf; q++)
        if (hipStreamCreateWithPriority(&h->stream_m[q], hipStreamNonBlocking, prio_normal) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipStreamCreate failed"));
    // DMPP_SIDE_CUS=<n>: measurement knob - the front and score streams may only use the first n CUs (0 / unset: all of them)
    int side_cus = 0;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cus = prop.multiProcessorCount;
        if (const char* e = std::getenv("DMPP_SIDE_CUS")) side_cus = std::atoi(e);
    }
    uint32_t cu_mask[32] = { 0 };
    if (side_cus > 1024) side_cus = 1024;
    for (int i = 0; i < side_cus; i++) cu_mask[i >> 5] |= 1u << (i & 31);
    const uint32_t cu_words = (uint32_t)((side_cus + 31) / 32);
    if (side_cus > 0 && hipExtStreamCreateWithCUMask(&h->stream_r, cu_words, cu_mask) != hipSuccess) { (void)hipGetLastError(); h->stream_r = nullptr; side_cus = 0; }
    if (side_cus > 0 && hipExtStreamCreateWithCUMask(&h->stream_s, cu_words, cu_mask) != hipSuccess) { (void)hipGetLastError(); h->stream_s = nullptr; }
    if (const char* e = std::getenv("DMPP_OVERLAP")) h->overlap_override = std::atoi(e);
    if (const char* e = std::getenv("DMPP_SCORE_STREAM")) { h->score_own_stream = std::atoi(e) != 0; h->score_stream_high = std::atoi(e) == 2; }
    if (const char* e = std::getenv("DMPP_FRONT_WAIT")) h->front_wait = std::atoi(e) != 0;
    if (!h->stream_r)
    if (hipStreamCreateWithPriority(&h->stream_r, hipStreamNonBlocking, prio_greatest) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipStreamCreate failed"));
    if (!h->stream_s)
    if (hipStreamCreateWithPriority(&h->stream_s, hipStreamNonBlocking, h->score_stream_high ? prio_greatest : prio_least) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipStreamCreate failed"));
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_raster, hipEventDisableTiming) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipEventCreate failed"));
    for (int q = 0; q < kBuf; q++)
        if (hipEventCreateWithFlags(&h->ev_search[q], hipEventDisableTiming) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipEventCreate failed"));
    for (int q = 0; q < kObs; q++)
        if (hipEventCreateWithFlags(&h->ev_score[q], hipEventDisableTiming) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipEventCreate failed"));
    const size_t ns = (size_t)caps->max_scenes;
    if ((r = dmalloc(&h->in_sets[0].d_in, ns))) return bail(r);
    h->d_in = h->in_sets[0].d_in;
    if ((r = dmalloc(&h->d_lane, (size_t)caps->max_lane_pts_total))) return bail(r);
    if ((r = dmalloc(&h->d_attr, (size_t)caps->max_lane_pts_total + 1))) return bail(r);
    if ((r = dmalloc(&h->d_ref, (size_t)caps->max_ref_pts_total))) return bail(r);
    if ((r = dmalloc(&h->in_sets[0].d_obs, (size_t)caps->max_obs_total))) return bail(r);
    h->d_obs = h->in_sets[0].d_obs;
    if ((r = dmalloc(&h->in_sets[0].d_mot, (size_t)caps->max_obs_total))) return bail(r);
    h->d_mot = h->in_sets[0].d_mot;
    if (hipMemsetAsync(h->d_mot, 0, (size_t)(caps->max_obs_total > 0 ? caps->max_obs_total : 1) * sizeof(ObMotion), h->stream) != hipSuccess)
        return bail(fail(PP_ERR_HIP, "memset failed"));     // velocities nobody uploaded are zero, never uninitialised
    if ((r = dmalloc(&h->d_bad, (size_t)1))) return bail(r);
    for (int q = 0; q < kObs; q++) if ((r = dmalloc(&h->d_obs_now[q], (size_t)caps->max_obs_total))) return bail(r);
    if ((r = dmalloc(&h->d_state, ns))) return bail(r);
    if ((r = dmalloc(&h->d_plan_ring[0], ns))) return bail(r);
    h->d_plan = h->d_plan_ring[0];
    if (hipMemsetAsync(h->d_plan, 0, ns * sizeof(PlanOut), h->stream) != hipSuccess) return bail(fail(PP_ERR_HIP, "memset failed"));
    for (int q = 0; q < kDone; q++) h->done_tick[q] = -1;
    for (int q = 0; q < kGout; q++) if ((r = dmalloc(&h->d_gout[q], ns))) return bail(r);
    if ((r = dmalloc(&h->d_dec_ref, ns * DMPP_MAX_REFPATH))) return bail(r);
    for (int q = 0; q < kGout; q++)
        if (hipMemsetAsync(h->d_gout[q], 0, ns * sizeof(GridOut), h->stream) != hipSuccess) return bail(fail(PP_ERR_HIP, "memset failed"));
    if (cfg->grid_stage) {
        h->grid_cells = (size_t)cfg->grid_w * cfg->grid_h;
        h->bucket_cap0 = cfg->bucket_cap; h->max_path0 = cfg->max_path;
        if ((r = dmalloc(&h->d_grid, h->grid_cells))) return bail(r);             // one scene as bytes, filled on demand (pp_get_grid)
        for (int q = 0; q < kBuf; q++) {
            if ((r = dmalloc(&h->d_pinfo[q], ns * h->grid_cells))) return bail(r);
            if ((r = dmalloc(&h->d_closed[q], ns * (h->grid_cells / 32)))) return bail(r);
        }
        for (int q = 0; q < kObs; q++) if ((r = dmalloc(&h->d_path[q], ns * (size_t)cfg->max_path))) return bail(r);
        for (int q = 0; q < kBuf; q++) if (caps->order_cap > 0 && (r = dmalloc(&h->d_order[q], ns * (size_t)caps->order_cap))) return bail(r);
        if (cfg->bucket_cap > DMPP_OPEN_CAP) {
            h->spill_cap = cfg->bucket_cap;
            for (int q = 0; q < kBuf; q++) if ((r = dmalloc(&h->d_ospill[q], ns * (size_t)h->spill_cap))) return bail(r);
        }
        if ((r = setup_grid_launch(h))) return bail(r);
    }
    h->scratch_bytes = 4u << 20;
    if (hipMalloc(&h->d_scratch, h->scratch_bytes) != hipSuccess) return bail(fail(PP_ERR_HIP, "hipMalloc(scratch) failed"));
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(PP_ERR_HIP, "sync failed"));
    *out = h;
    return PP_OK;
}

int pp_destroy(pp_handle h)
{
    if (!h) return PP_OK;
    (void)hipSetDevice(h->device);
    for (hipStream_t st : { h->stream, h->stream_r, h->stream_s }) if (st) (void)hipStreamSynchronize(st);
    for (int q = 1; q < kBuf; q++) if (h->stream_m[q]) (void)hipStreamSynchronize(h->stream_m[q]);
    for (auto& p : h->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : h->free_events) (void)hipEventDestroy(e);
    if (h->stream_s) (void)hipStreamSynchronize(h->stream_s);
    if (h->stream_dg == h->stream_dp) h->stream_dg = nullptr;
    if (h->stream_dp == h->stream_up) h->stream_dp = nullptr;
    for (hipStream_t st : { h->stream_up, h->stream_dp, h->stream_dg }) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int q = 0; q < kObs; q++) if (h->d_obs_now[q]) (void)hipFree(h->d_obs_now[q]);
    for (int q = 0; q < kIn; q++) {
        InputSet& I = h->in_sets[q];
        for (void* b : { (void*)I.d_in, (void*)I.d_obs, (void*)I.d_mot }) if (b) (void)hipFree(b);
        if (I.ev_up) (void)hipEventDestroy(I.ev_up);
    }
    for (int q = 0; q < kPlan; q++) { if (h->d_plan_ring[q]) (void)hipFree(h->d_plan_ring[q]); if (h->ev_fetched_plan[q]) (void)hipEventDestroy(h->ev_fetched_plan[q]); }
    for (int q = 0; q < kGout; q++) { if (h->ev_fetched_grid[q]) (void)hipEventDestroy(h->ev_fetched_grid[q]); if (h->d_gout[q]) (void)hipFree(h->d_gout[q]); }
    for (int q = 0; q < kDone; q++) { if (h->ev_done_p[q]) (void)hipEventDestroy(h->ev_done_p[q]); if (h->ev_done_g[q]) (void)hipEventDestroy(h->ev_done_g[q]); }
    for (auto e : h->sync_events) (void)hipEventDestroy(e);
    for (auto& r : h->inflight) { (void)hipEventDestroy(r.ev_front); if (r.ev_tail) (void)hipEventDestroy(r.ev_tail); }
    if (h->h_bad) (void)hipHostFree(h->h_bad);
    void* bufs[] = { h->d_lane, h->d_attr, h->d_ref, h->d_state,
                     h->d_dec_ref, h->d_grid, h->d_scratch, h->d_map_first, h->d_map_lanes, h->d_map_width, h->d_map_junc, h->d_map_bad, h->d_bad,
                     h->d_gridbad };
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (int q = 0; q < kBuf; q++)
        for (void* b : { (void*)h->d_ospill[q], (void*)h->d_retry[q], (void*)h->d_pinfo[q], (void*)h->d_closed[q], (void*)h->d_order[q],
                         (void*)h->d_gbm[q], (void*)h->d_perm[q], (void*)h->d_cost[q], (void*)h->d_ovf[q] })
            if (b) (void)hipFree(b);
    for (int q = 0; q < kObs; q++) for (void* b : { (void*)h->d_path[q], (void*)h->d_need[q] }) if (b) (void)hipFree(b);
    for (int q = 0; q < kBuf; q++) if (h->ev_search[q]) (void)hipEventDestroy(h->ev_search[q]);
    for (int q = 0; q < kObs; q++) if (h->ev_score[q]) (void)hipEventDestroy(h->ev_score[q]);
    if (h->ev_raster) (void)hipEventDestroy(h->ev_raster);
    if (h->h_need) (void)hipHostFree(h->h_need);
    if (h->stream_s) (void)hipStreamDestroy(h->stream_s);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream_r) { (void)hipStreamSynchronize(h->stream_r); (void)hipStreamDestroy(h->stream_r); }
    for (int q = 1; q < kBuf; q++) if (h->stream_m[q]) (void)hipStreamDestroy(h->stream_m[q]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PP_OK;
}

int pp_set_config(pp_handle h, const PlannerConfig* cfg)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    int r = check_cfg(cfg); if (r) return r;
    if (cfg->grid_stage) {
        if (!h->d_grid) return fail(PP_ERR_STATE, "handle was created without the grid stage");
        if ((size_t)cfg->grid_w * cfg->grid_h > h->grid_cells || cfg->max_path > h->max_path0)
            return fail(PP_ERR_CAPACITY, "grid size / max_path may not grow after pp_create");
        if (cfg->bucket_cap > DMPP_OPEN_CAP && cfg->bucket_cap > h->spill_cap)
            return fail(PP_ERR_CAPACITY, "bucket_cap may not grow after pp_create beyond the larger of its value then and DMPP_OPEN_CAP (it sizes the open list's spill area)");
    }
    HIP_TRY(hipSetDevice(h->device));
    // a configuration change may move the next tick's kernels to other streams (grid stage on / off): finish what is queued
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->cfg = *cfg;
    return setup_grid_launch(h);
}

// Slices of the resident SceneIn records against the resident pools (k_validate_scenes); syncs the handle's stream.
static int validate_resident(pp_handle h, int n_scenes, const char* who)
{
    int bad = 0;
    if (n_scenes > 0) {
        HIP_TRY(hipMemsetAsync(h->d_bad, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(dmpp::k_validate_scenes, dim3((unsigned)((n_scenes + dmpp::kBlock - 1) / dmpp::kBlock)), dim3(dmpp::kBlock), 0, h->stream,
                           n_scenes, h->d_in, h->n_obs_total, h->n_lane_pts, h->n_ref_pts, h->d_bad);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&bad, h->d_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bad) {
        h->n_scenes = 0;                                // nothing a tick could follow out of its pool stays resident
        return fail(PP_ERR_ARG, std::string(who) + ": " + std::to_string(bad) + " scene(s) with an obstacle / lane / refpath slice outside its pool");
    }
    return PP_OK;
}

int pp_set_scenes(pp_handle h, int n_scenes, const SceneIn* in, const GlobalPoint3D* lane_pool, const uint8_t* lane_attr_pool,
                  int n_lane_pts, const GlobalPoint2D* ref_pool, int n_ref_pts, const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total)
{
    if (!h || !in) return fail(PP_ERR_ARG, "null argument");
    if (n_scenes < 0 || n_scenes > h->caps.max_scenes) return fail(PP_ERR_CAPACITY, "n_scenes exceeds caps.max_scenes");
    if (n_lane_pts > h->caps.max_lane_pts_total || n_ref_pts > h->caps.max_ref_pts_total || n_obs_total > h->caps.max_obs_total)
        return fail(PP_ERR_CAPACITY, "pool larger than the capacity given to pp_create");
    if (n_lane_pts < 0 || n_ref_pts < 0 || n_obs_total < 0) return fail(PP_ERR_ARG, "negative size");
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipMemcpyAsync(h->d_in, in, (size_t)n_scenes * sizeof(SceneIn), hipMemcpyDefault, h->stream));
    if (n_lane_pts && lane_pool) HIP_TRY(hipMemcpyAsync(h->d_lane, lane_pool, (size_t)n_lane_pts * sizeof(GlobalPoint3D), hipMemcpyDefault, h->stream));
    h->have_attr = false;
    if (n_lane_pts && lane_attr_pool) {
        HIP_TRY(hipMemcpyAsync(h->d_attr, lane_attr_pool, (size_t)n_lane_pts, hipMemcpyDefault, h->stream));
        h->have_attr = true;
    }
    if (n_ref_pts && ref_pool) HIP_TRY(hipMemcpyAsync(h->d_ref, ref_pool, (size_t)n_ref_pts * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    if (n_obs_total && obs_pool) HIP_TRY(hipMemcpyAsync(h->d_obs, obs_pool, (size_t)n_obs_total * sizeof(ObPoint), hipMemcpyDefault, h->stream));
    h->have_motion = false;
    if (n_obs_total && mot_pool) {
        HIP_TRY(hipMemcpyAsync(h->d_mot, mot_pool, (size_t)n_obs_total * sizeof(ObMotion), hipMemcpyDefault, h->stream));
        h->have_motion = true;
    }
    h->n_scenes = n_scenes; h->n_obs_total = n_obs_total; h->n_lane_pts = n_lane_pts; h->n_ref_pts = n_ref_pts;
    h->resident_mode = 0; note_current_set(h);
    return validate_resident(h, n_scenes, "pp_set_scenes");     // syncs: the caller may reuse its buffers
}

static int fetch(pp_handle h, void* dst, const void* src, size_t bytes);

int pp_set_map(pp_handle h, const MapDesc* m)
{
    if (!h || !m) return fail(PP_ERR_ARG, "null argument");
    if (m->n_roads < 0 || m->n_lanes < 0 || m->n_points < 0 || m->n_junctions < 0 || m->n_jpoints < 0) return fail(PP_ERR_ARG, "negative size");
    if (m->n_points > h->caps.max_lane_pts_total || m->n_jpoints > h->caps.max_ref_pts_total)
        return fail(PP_ERR_CAPACITY, "map larger than caps.max_lane_pts_total / max_ref_pts_total");
    if ((m->n_lanes && (!m->road_first_lane || !m->lanes)) || (m->n_points && (!m->points || !m->lanechg_attribute || !m->lane_width_cm)) ||
        (m->n_junctions && !m->junctions) || (m->n_jpoints && !m->jpoints)) return fail(PP_ERR_ARG, "null map array");
    // the tables are host arrays: every slice is checked here, so that no scene can index outside the pools
    if (m->n_roads && (m->road_first_lane[0] != 0 || m->road_first_lane[m->n_roads] != m->n_lanes)) return fail(PP_ERR_ARG, "road_first_lane must run from 0 to n_lanes");
    for (int r = 0; r < m->n_roads; r++) if (m->road_first_lane[r + 1] < m->road_first_lane[r]) return fail(PP_ERR_ARG, "road_first_lane must not decrease");
    for (int l = 0; l < m->n_lanes; l++) {
        const MapLane& L = m->lanes[l];
        if (L.point_off < 0 || L.n_points < 0 || (long long)L.point_off + L.n_points > m->n_points) return fail(PP_ERR_ARG, "lane slice outside the point pool");
    }
    for (int j = 0; j < m->n_junctions; j++) {
        const MapJunction& J = m->junctions[j];
        if (J.point_off < 0 || J.n_points < 0 || (long long)J.point_off + J.n_points > m->n_jpoints) return fail(PP_ERR_ARG, "junction slice outside the junction point pool");
    }
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    void* old[] = { h->d_map_first, h->d_map_lanes, h->d_map_junc };
    for (void* b : old) if (b) (void)hipFree(b);
    h->d_map_first = nullptr; h->d_map_lanes = nullptr; h->d_map_junc = nullptr; h->have_map = false;
    int r;
    if ((r = dmalloc(&h->d_map_first, (size_t)m->n_roads + 1))) return r;
    if ((r = dmalloc(&h->d_map_lanes, (size_t)(m->n_lanes > 0 ? m->n_lanes : 1)))) return r;
    if ((r = dmalloc(&h->d_map_junc, (size_t)(m->n_junctions > 0 ? m->n_junctions : 1)))) return r;
    if (!h->d_map_width && (r = dmalloc(&h->d_map_width, (size_t)h->caps.max_lane_pts_total + 1))) return r;
    if (!h->d_map_bad && (r = dmalloc(&h->d_map_bad, (size_t)1))) return r;
    if (m->n_roads) HIP_TRY(hipMemcpyAsync(h->d_map_first, m->road_first_lane, ((size_t)m->n_roads + 1) * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    else HIP_TRY(hipMemsetAsync(h->d_map_first, 0, sizeof(int32_t), h->stream));
    if (m->n_lanes) HIP_TRY(hipMemcpyAsync(h->d_map_lanes, m->lanes, (size_t)m->n_lanes * sizeof(MapLane), hipMemcpyHostToDevice, h->stream));
    if (m->n_junctions) HIP_TRY(hipMemcpyAsync(h->d_map_junc, m->junctions, (size_t)m->n_junctions * sizeof(MapJunction), hipMemcpyHostToDevice, h->stream));
    if (m->n_points) {
        HIP_TRY(hipMemcpyAsync(h->d_lane, m->points, (size_t)m->n_points * sizeof(GlobalPoint3D), hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_attr, m->lanechg_attribute, (size_t)m->n_points, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_map_width, m->lane_width_cm, (size_t)m->n_points * sizeof(uint16_t), hipMemcpyDefault, h->stream));
    }
    if (m->n_jpoints) HIP_TRY(hipMemcpyAsync(h->d_ref, m->jpoints, (size_t)m->n_jpoints * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->map_roads = m->n_roads; h->map_lanes = m->n_lanes; h->map_junctions = m->n_junctions;
    h->n_lane_pts = m->n_points; h->n_ref_pts = m->n_jpoints;
    h->have_map = true; h->have_attr = true;
    return PP_OK;
}

int pp_set_egos(pp_handle h, int n_scenes, const SceneIn* in, const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total)
{
    if (!h || !in) return fail(PP_ERR_ARG, "null argument");
    if (!h->have_map) return fail(PP_ERR_STATE, "pp_set_egos needs a resident map (pp_set_map)");
    if (n_scenes < 0 || n_scenes > h->caps.max_scenes) return fail(PP_ERR_CAPACITY, "n_scenes exceeds caps.max_scenes");
    if (n_obs_total < 0 || n_obs_total > h->caps.max_obs_total) return fail(PP_ERR_CAPACITY, "obstacle pool larger than caps.max_obs_total");
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipMemcpyAsync(h->d_in, in, (size_t)n_scenes * sizeof(SceneIn), hipMemcpyDefault, h->stream));
    if (n_obs_total && obs_pool) HIP_TRY(hipMemcpyAsync(h->d_obs, obs_pool, (size_t)n_obs_total * sizeof(ObPoint), hipMemcpyDefault, h->stream));
    h->have_motion = false;
    if (n_obs_total && mot_pool) {
        HIP_TRY(hipMemcpyAsync(h->d_mot, mot_pool, (size_t)n_obs_total * sizeof(ObMotion), hipMemcpyDefault, h->stream));
        h->have_motion = true;
    }
    HIP_TRY(hipMemsetAsync(h->d_map_bad, 0, sizeof(int), h->stream));
    if (n_scenes)
        hipLaunchKernelGGL(dmpp::k_resolve_map, dim3((unsigned)((n_scenes + dmpp::kBlock - 1) / dmpp::kBlock)), dim3(dmpp::kBlock), 0, h->stream,
                           n_scenes, h->d_in, h->map_roads, h->d_map_first, h->d_map_lanes, h->d_attr, h->d_map_width,
                           h->map_junctions, h->d_map_junc, h->d_map_bad);
    HIP_TRY(hipGetLastError());
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, h->d_map_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->n_scenes = n_scenes; h->n_obs_total = n_obs_total;
    h->resident_mode = 1; note_current_set(h);
    if (bad) { h->n_scenes = 0; return fail(PP_ERR_ARG, "pp_set_egos: " + std::to_string(bad) + " scene(s) name a road or lane outside the map"); }
    return validate_resident(h, n_scenes, "pp_set_egos");       // the obstacle slices are still the caller's
}

int pp_get_scene_in(pp_handle h, SceneIn* out, int n)
{
    if (!h || !out) return fail(PP_ERR_ARG, "null argument");
    if (n < 0 || n > h->n_scenes) return fail(PP_ERR_ARG, "n exceeds the resident scenes");
    return fetch(h, out, h->d_in, (size_t)n * sizeof(SceneIn));
}

int pp_set_n_scenes(pp_handle h, int n_scenes, int n_lane_pts, int n_ref_pts, int n_obs_total, int have_motion, int have_lane_attr)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    if (n_scenes < 0 || n_scenes > h->caps.max_scenes) return fail(PP_ERR_CAPACITY, "n_scenes exceeds caps.max_scenes");
    if (n_lane_pts < 0 || n_ref_pts < 0 || n_obs_total < 0) return fail(PP_ERR_ARG, "negative size");
    if (n_lane_pts > h->caps.max_lane_pts_total || n_ref_pts > h->caps.max_ref_pts_total || n_obs_total > h->caps.max_obs_total)
        return fail(PP_ERR_CAPACITY, "pool larger than the capacity given to pp_create");
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    h->n_scenes = n_scenes; h->n_obs_total = n_obs_total; h->n_lane_pts = n_lane_pts; h->n_ref_pts = n_ref_pts;
    h->have_motion = have_motion != 0; h->have_attr = have_lane_attr != 0;
    h->resident_mode = 0; note_current_set(h);
    return validate_resident(h, n_scenes, "pp_set_n_scenes");
}

int pp_set_state(pp_handle h, const SceneState* state, int n)
{
    if (!h || !state) return fail(PP_ERR_ARG, "null argument");
    if (n < 0 || n > h->caps.max_scenes) return fail(PP_ERR_CAPACITY, "n exceeds caps.max_scenes");
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipMemcpyAsync(h->d_state, state, (size_t)n * sizeof(SceneState), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_plan_tick(pp_handle h)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    const int n = h->n_scenes;
    if (n <= 0) return PP_OK;
    HIP_TRY(hipSetDevice(h->device));
    const PlannerConfig& c = h->cfg;
    if (c.decision_stage && c.lanechg_stage && !h->have_attr)
        return fail(PP_ERR_ARG, "cfg.lanechg_stage needs the lane attribute pool (pp_set_scenes lane_attr_pool)");
    // One tick = two chains.  FRONT (stream_r, highest priority): obstacle snapshot, Decision, Planning - short kernels, serial
    // from tick to tick through SceneState.  SEARCH (stream_m[t % kBuf]): (launch order,) k_search, k_search_spill, k_score - the
    // long one; the searches of kBuf consecutive ticks run side by side on their own streams.
    //   search(t) waits for the snapshot of front(t) [ev_raster] and follows score(t - kBuf) on its stream;
    //   front(t)  waits for score(t - kObs), whose snapshot set it overwrites - so it may run up to kObs ticks ahead of the scoring.
    // What a search and its scoring pass write exists kBuf times (closed sets, orders, ...), what the scoring pass reads beside
    // a later search kObs times (snapshot, path cells, LDS need), GridOut kGout times.
    // Small batches and ticks without the grid stage stay on one stream (a cross-stream hand-over costs tens of
    // microseconds; only Decision + Planning run beside the grid engine there).
    // (a tick without the grid stage leaves the search buffers alone: pp_get_grid_out / pp_get_path keep returning the last search)
    const int p = c.grid_stage ? (h->parity + 1) % kBuf : h->parity, p_prev = h->parity;
    const int gs = c.grid_stage ? (h->gout_set + 1) % kGout : h->gout_set;
    const bool piped = c.grid_stage && n >= h->pipeline_min;
    // streamed inputs: the update staged by pp_update_async becomes the set this tick (and the following ones) read
    bool adopted = false;
    if (h->in_staged >= 0) { adopt_input_set(h, h->in_staged); h->in_staged = -1; adopted = true; }
    if (h->streaming) {
        // downloads whose kernels have finished are issued now; one that still reads a set this tick overwrites is issued whatever
        int r = pump_fetches(h, -1, (h->plan_cur + 1) % kPlan, c.grid_stage ? (h->gout_set + 1) % kGout : -1); if (r) return r;
        r = prune_inflight(h); if (r) return r;
        h->plan_cur = (h->plan_cur + 1) % kPlan; h->d_plan = h->d_plan_ring[h->plan_cur];
        if (!adopted) h->h_bad[(h->tick_seq + 1) % kDone] = 0;
    }
    if (c.grid_stage && !h->search_force_gbm) {
        // LDS budget of the search (data words per view).  First tick: from the obstacle density; afterwards from what the
        // densest scene of an earlier tick needed (+ 1/8): the scoring pass behind each search stores it in pinned memory, which
        // is simply read here - whatever has landed; never waited for.
        if (!h->lds_budget_fixed) {
            int need = -1;
            for (int q = 0; q < kBuf; q++) need = std::max(need, (int)reinterpret_cast<volatile int32_t*>(h->h_need)[(h->obs_set + kObs - q) % kObs]);     // the last kBuf ticks' sets
            int want = h->lds_budget;
            if (need >= 0) {
                h->need_seen = need;
                const int fit = std::min(h->lds_budget_max, (need + need / 8 + 64 + 63) / 64 * 64);
                // Workgroups per CU at a budget (the 160 KB of LDS are handed out in 128 granules of 1,280 bytes - measured: 54,000 bytes per
                // workgroup are two per CU, 52,976 three, 26,864 six).
                // When the scenes outnumber the workgroup slots and a smaller - still safe - slack over the need lets one more
                // searching workgroup onto every CU, the largest budget that does is taken: 256 moving obstacles need ~4,650 words,
                // 5,312 with the usual eighth on top = two workgroups of 55 KB per CU; three fit at <= 5,120 (configs[3]: 1.14 -> 1.23 M
                // ticks/s); 4096 scenes of 64 obstacles: six of 26.9 KB instead of five of 27.9 (5.17 -> 5.30 M).  With a slot for every
                // scene the eighth stays: the room it leaves on the CU is what the front kernels start in.  A scene that outgrows the
                // budget takes the dense form in HBM.
                const size_t fixed_lds = h->search_static_lds + 64 + (size_t)h->search_meta_bytes;
                constexpr size_t kLdsGranule = 1280;
                auto wgs_at = [&](int b) { return (int)std::min<size_t>(8, (160u * 1024u) / ((fixed_lds + 8 * (size_t)b + kLdsGranule - 1) / kLdsGranule * kLdsGranule)); };
                int target = fit;
                const int tight = std::min(h->lds_budget_max, (need + std::max(need / 32, 96) + 63) / 64 * 64);
                if (tight < fit && wgs_at(tight) > wgs_at(fit) && n > wgs_at(fit) * std::max(1, h->n_cus)) {
                    const size_t room = (160u * 1024u) / (size_t)wgs_at(tight) / kLdsGranule * kLdsGranule;
                    const int lim = room > fixed_lds ? (int)((room - fixed_lds) / 8 / 64 * 64) : 0;
                    target = std::max(tight, std::min(lim, fit));
                }
                // (the first need that arrives replaces the first tick's guess outright: 64 obstacles were guessed at 2,048 words, need
                // 1,635, and the hysteresis kept the guess - 28.2 KB per workgroup, 18.8 KB free beside five of them, 0.5 KB short of
                // a k_decision workgroup; at 1,920 it fits: +1 % at 1024 scenes, +3 % at 4096)
                if (!h->budget_from_need || target > h->lds_budget || target < h->lds_budget - h->lds_budget / 4 || (h->lds_budget > 0 && wgs_at(target) > wgs_at(h->lds_budget))) want = target;
                h->budget_from_need = true;
            }
            if (want <= 0) {
                const long long per_scene = ((long long)h->n_obs_total + n - 1) / n;
                want = (int)std::min<long long>(h->lds_budget_max, (28 * per_scene + 256 + 63) / 64 * 64);
            }
            h->lds_budget = std::max(64, std::min(want, h->lds_budget_max));
        }
        const size_t per_wg = h->search_static_lds + 64 + std::max((size_t)h->search_meta_bytes + 8 * (size_t)h->lds_budget, (size_t)h->gbm_lds);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / per_wg));       // 4 waves per workgroup while it sets up: <= 8 per CU
        h->search_slots = per_cu * std::max(1, h->n_cus);
    } else if (c.grid_stage) {
        const size_t per_wg = h->search_static_lds + 64 + (size_t)h->gbm_lds;
        h->search_slots = (int)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / per_wg)) * std::max(1, h->n_cus);
    }
    // Consecutive searches overlap: a search ends with a handful of long scenes and would leave most of the chip idle; the searches
    // of up to kBuf consecutive ticks run on their own streams (every buffer a search or a scoring pass touches exists kBuf times).
    bool overlap = piped;
    if (h->overlap_override >= 0) overlap = piped && h->overlap_override != 0;      // env DMPP_OVERLAP (measurement knob)
    hipStream_t sm = overlap ? h->stream_m[p] : h->stream;             // search chain
    hipStream_t sf = piped ? h->stream_r : h->stream;                  // front chain
    hipStream_t ss = piped ? ((overlap && !h->score_own_stream) ? sm : h->stream_s) : h->stream; // score chain: behind its own search when the searches overlap
    hipStream_t sr = c.grid_stage ? h->stream_r : h->stream;           // Decision + Planning
    // Buffers p were last used by tick t - kBuf (snapshot set po_b), snapshot set po - and with it the path cells and the LDS need
    // of the search - by tick t - 2 kBuf.  The search follows the scoring pass of tick t - kBuf on its stream; the front chain waits
    // for the scoring pass of tick t - 2 kBuf (it overwrites that pass's snapshot) and, see below, for the search of tick t - kBuf.
    const int po = (h->obs_set + 1) % kObs, po_b = (po + kBuf) % kObs;
    if (h->score_recorded[po_b] && ss == sm) HIP_TRY(hipStreamWaitEvent(sm, h->ev_score[po_b], 0));     // (own scoring stream: path cells and LDS need exist per snapshot set, the search of tick t does not wait for the scoring of t - kBuf)
    if (h->score_recorded[po]) HIP_TRY(hipStreamWaitEvent(sf, h->ev_score[po], 0));
    // The front chain does not wait for the search of tick t - kBuf: it runs ahead - up to kObs ticks, bounded by the snapshot
    // sets - so its kernels no longer start together with the scoring pass that follows that search (+ 2 %), and the launch order,
    // which needs that search's times, is computed on the search's own stream.  DMPP_FRONT_WAIT=1: the old hand-over (measurement knob).
    const bool front_wait = h->front_wait;
    if (h->search_recorded[p] && front_wait) HIP_TRY(hipStreamWaitEvent(sf, h->ev_search[p], 0));
    if (!overlap)                                                      // one search at a time (also after a switch of mode)
        for (int q = 0; q < kBuf; q++) if (q != p && h->search_recorded[q]) HIP_TRY(hipStreamWaitEvent(sm, h->ev_search[q], 0));
    if (h->front_recorded && sf == h->stream && h->front_unjoined) HIP_TRY(hipStreamWaitEvent(sf, h->ev_join, 0));   // Planning(t-1) -> snapshot(t) when not on the same stream
    h->front_unjoined = false;
    if (h->r_on_main && sf != h->stream) {         // Planning(t-1) ran on the handle's stream (grid stage off then): the front chain reads its state
        HIP_TRY(hipEventRecord(h->ev_fork, h->stream)); HIP_TRY(hipStreamWaitEvent(sf, h->ev_fork, 0));
    }
    h->r_on_main = sr == h->stream;
    // (an event that has completed needs no barrier packet on the front chain: uploads and downloads run ticks ahead / behind)
    auto wait_unless_done = [](hipStream_t st, hipEvent_t e) { if (hipEventQuery(e) == hipSuccess) return hipSuccess; (void)hipGetLastError(); return hipStreamWaitEvent(st, e, 0); };
    if (adopted && h->in_sets[h->in_cur].up_recorded) HIP_TRY(wait_unless_done(sf, h->in_sets[h->in_cur].ev_up));   // every kernel of the tick follows the snapshot kernel
    if (h->streaming && h->fetched_plan_rec[h->plan_cur]) HIP_TRY(wait_unless_done(sr, h->ev_fetched_plan[h->plan_cur]));   // PlanOut set still being downloaded (kPlan ticks ago)
    ObPoint* obs_now = h->d_obs_now[po];
    {
        Timed t(h, PP_K_OBSTACLES, sf);
        hipLaunchKernelGGL(dmpp::k_effective_obstacles, dim3(n), dim3(dmpp::kBlock), 0, sf, c, n, h->d_in, h->d_state,
                           h->d_obs, h->have_motion ? h->d_mot : nullptr, obs_now);
    }
    if (sr != sf) { HIP_TRY(hipEventRecord(h->ev_fork, sf)); HIP_TRY(hipStreamWaitEvent(sr, h->ev_fork, 0)); }   // small batches: Decision + Planning beside the grid engine
    // launch order of the search (heaviest scenes first) - pointless while every scene is resident at once.  Keyed by the times of
    // the search kBuf ticks back, the one before it on its stream (one-stream tick: by the previous tick's).
    const bool order_scenes = c.grid_stage && n > h->search_slots;
    const bool order_in_front = order_scenes && piped && front_wait;
    if (order_in_front) hipLaunchKernelGGL(dmpp::k_order, dim3(1), dim3(dmpp::kOrderBlock), 0, sf, n, h->d_cost[p], h->d_perm[p]);
    if (sf != sm) HIP_TRY(hipEventRecord(h->ev_raster, sf));        // the search rasterises for itself: it only needs the obstacle snapshot (and its launch order)
    if (c.decision_stage) {
        Timed t(h, PP_K_DECISION, sr);
        hipLaunchKernelGGL(dmpp::k_decision, dim3(n), dim3(dmpp::kBlock), sizeof(dmpp::DecShared), sr, c, n, h->d_in, h->d_lane,
                           h->d_attr, h->d_ref, obs_now, h->d_state, h->d_plan, h->d_dec_ref);
    }
    {
        Timed t(h, PP_K_PLANNING, sr);
        hipLaunchKernelGGL(dmpp::k_planning, dim3(n), dim3(dmpp::kBlock), sizeof(dmpp::PlanShared), sr, c, n, h->d_in, h->d_lane,
                           h->d_ref, h->d_dec_ref, obs_now, h->d_state, h->d_plan);
    }
    if (sr != h->stream) { HIP_TRY(hipEventRecord(h->ev_join, sr)); h->front_recorded = true; h->front_unjoined = piped; }
    if (c.grid_stage) {
        if (sf != sm) HIP_TRY(hipStreamWaitEvent(sm, h->ev_raster, 0));
        if (h->streaming && h->fetched_grid_rec[gs]) HIP_TRY(wait_unless_done(sm, h->ev_fetched_grid[gs]));   // GridOut set still being downloaded (kGout grid ticks ago)
        const int32_t* perm = order_scenes ? h->d_perm[p] : nullptr;
        if (perm && !order_in_front)
            hipLaunchKernelGGL(dmpp::k_order, dim3(1), dim3(dmpp::kOrderBlock), 0, sm, n, h->d_cost[overlap ? p : p_prev], h->d_perm[p]);
        {
            const int budget = h->search_force_gbm ? 0 : h->lds_budget;
            const bool wide = n <= kScoreWideMaxScenes;       // a few scenes: sixteen waves set each scene up (the latency-bound tick)
            const size_t dyn = std::max((size_t)h->search_meta_bytes + 8 * (size_t)budget, (size_t)h->gbm_lds);
            const bool use_spill = h->d_ospill[p] != nullptr && c.bucket_cap > DMPP_OPEN_CAP;     // scenes whose open list outgrows LDS are searched again, spilling
            {
                Timed t(h, PP_K_SEARCH, sm);
                switch (h->search_kind) {
#define DMPP_LAUNCH_SEARCH(K)                                                                                                                  \
                case K:                                                                                                                        \
                    if (wide) hipLaunchKernelGGL((dmpp::k_search<K, dmpp::kSearchSetupWavesWide>), dim3(n), dim3(dmpp::kSearchSetupWavesWide * DMPP_WAVE), dyn, sm, c, n, h->caps.order_cap, budget, perm, \
                                           h->d_in, obs_now, h->d_closed[p], h->d_pinfo[p], h->d_order[p], h->d_path[po], h->d_gout[gs], h->d_gbm[p], \
                                           h->d_cost[p], h->d_ovf[p], h->d_need[po], h->d_ospill[p], h->spill_cap, h->d_retry[p], use_spill ? h->d_need[po] + 1 : nullptr); \
                    else hipLaunchKernelGGL((dmpp::k_search<K, dmpp::kSearchSetupWaves>), dim3(n), dim3(dmpp::kSearchBlock), dyn, sm, c, n, h->caps.order_cap, budget, perm,  \
                                           h->d_in, obs_now, h->d_closed[p], h->d_pinfo[p], h->d_order[p], h->d_path[po], h->d_gout[gs], h->d_gbm[p], \
                                           h->d_cost[p], h->d_ovf[p], h->d_need[po], h->d_ospill[p], h->spill_cap, h->d_retry[p], use_spill ? h->d_need[po] + 1 : nullptr); \
                    if (use_spill) hipLaunchKernelGGL((dmpp::k_search_spill<K>), dim3(std::min(n, 2)), dim3(dmpp::kSearchBlock), dyn, sm, c, n, h->caps.order_cap, budget,                       \
                                           h->d_in, obs_now, h->d_closed[p], h->d_pinfo[p], h->d_order[p], h->d_path[po], h->d_gout[gs], h->d_gbm[p],                              \
                                           h->d_cost[p], h->d_ovf[p], h->d_need[po], h->d_ospill[p], h->spill_cap, h->d_retry[p], h->d_need[po] + 1);                           \
                    break;
                DMPP_LAUNCH_SEARCH(0) DMPP_LAUNCH_SEARCH(1) DMPP_LAUNCH_SEARCH(2)
#undef DMPP_LAUNCH_SEARCH
                }
            }
        }
        h->search_recorded[p] = piped;                   // (one-stream mode: stream order is enough, no events on the latency path)
        if (piped) { HIP_TRY(hipEventRecord(h->ev_search[p], sm)); HIP_TRY(hipStreamWaitEvent(ss, h->ev_search[p], 0)); }
        {
            Timed t(h, PP_K_SCORE, ss);
            int32_t* need_host = (!h->lds_budget_fixed && !h->search_force_gbm) ? &h->h_need[po] : nullptr;
            if (n <= kScoreWideMaxScenes)     // few scenes: sixteen waves per scene (17 candidates in two rounds)
                hipLaunchKernelGGL(dmpp::k_score<16>, dim3(n), dim3(16 * DMPP_WAVE), sizeof(dmpp::ScoreShared<16>), ss, c, n, h->d_in, obs_now,
                                   h->d_path[po], h->d_gout[gs], h->d_need[po], need_host);
            else
                hipLaunchKernelGGL(dmpp::k_score<4>, dim3(n), dim3(4 * DMPP_WAVE), sizeof(dmpp::ScoreShared<4>), ss, c, n, h->d_in, obs_now,
                                   h->d_path[po], h->d_gout[gs], h->d_need[po], need_host);
        }
        h->score_recorded[po] = piped;
        if (piped) HIP_TRY(hipEventRecord(h->ev_score[po], ss));
        if (!piped && sr != h->stream) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));   // one-stream mode: the tick is complete on the handle's stream
    }
    h->parity = p; h->obs_set = po; h->gout_set = gs; h->last_piped = piped;
    if (c.grid_stage) h->path_set = po;
    h->tick_seq++;
    if (h->streaming) {          // what a later update of this tick's input set, and a download of its results, wait for
        TickRec rec = { h->tick_seq, h->in_cur, get_sync_event(h), c.grid_stage ? get_sync_event(h) : nullptr };
        if (!rec.ev_front || (c.grid_stage && !rec.ev_tail)) return fail(PP_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(rec.ev_front, sr));
        if (rec.ev_tail) HIP_TRY(hipEventRecord(rec.ev_tail, piped ? ss : h->stream));
        h->inflight.push_back(rec); h->last_rec = rec;
    }
    HIP_TRY(hipGetLastError());
    return PP_OK;
}

int pp_join(pp_handle h)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return join_all(h);
}

int pp_sync(pp_handle h)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->streaming) {          // staged updates and downloads too (everything has finished: every pending copy is issued)
        int r = pump_fetches(h); if (r) return r;
        for (hipStream_t st : { h->stream_up, h->stream_dp, h->stream_dg }) HIP_TRY(hipStreamSynchronize(st));
    }
    return PP_OK;
}

int pp_device_synchronize(pp_handle h)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int r = pp_sync(h); if (r) return r; }
    HIP_TRY(hipDeviceSynchronize());
    return PP_OK;
}

static int fetch(pp_handle h, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_get_plan(pp_handle h, PlanOut* out, int n)
{
    if (!h || !out) return fail(PP_ERR_ARG, "null argument");
    if (n < 0 || n > h->n_scenes) return fail(PP_ERR_ARG, "n exceeds the resident scenes");
    return fetch(h, out, h->d_plan, (size_t)n * sizeof(PlanOut));
}
int pp_get_state(pp_handle h, SceneState* st, int n)
{
    if (!h || !st) return fail(PP_ERR_ARG, "null argument");
    if (n < 0 || n > h->caps.max_scenes) return fail(PP_ERR_ARG, "n exceeds capacity");
    return fetch(h, st, h->d_state, (size_t)n * sizeof(SceneState));
}
int pp_get_grid_out(pp_handle h, GridOut* out, int n)
{
    if (!h || !out) return fail(PP_ERR_ARG, "null argument");
    if (n < 0 || n > h->n_scenes) return fail(PP_ERR_ARG, "n exceeds the resident scenes");
    return fetch(h, out, h->d_gout[h->gout_set], (size_t)n * sizeof(GridOut));
}
int pp_get_grid(pp_handle h, int scene, uint8_t* grid)
{
    if (!h || !grid || !h->d_grid) return fail(PP_ERR_ARG, "no grid");
    if (scene < 0 || scene >= h->n_scenes) return fail(PP_ERR_ARG, "scene out of range");
    const PlannerConfig& c = h->cfg;
    const size_t N = (size_t)c.grid_w * c.grid_h;
    // No occupancy grid exists after a tick: the search builds its sparse bitmaps in LDS and drops them.  The grid asked for is
    // produced here, from the obstacle snapshot of the last tick, by the same footprint code (k_export_grid), which also checks
    // the column-major view against the row-major one (unless the scene is too dense for one workgroup's LDS).
    HIP_TRY(hipSetDevice(h->device));
    { int r = join_all(h); if (r) return r; }       // the snapshot of the last tick was written on another stream
    const ObPoint* obs_now = h->d_obs_now[h->obs_set];
    int bad[2] = { 0, 0 };
    HIP_TRY(hipMemsetAsync(h->d_gridbad, 0, 2 * sizeof(int), h->stream));
    const int budget = h->search_force_gbm ? 1 : h->lds_budget_max;      // (1 word: nothing fits, the span-by-span path)
    const size_t dyn = (size_t)h->search_meta_bytes + 8 * (size_t)budget;
    switch (h->search_kind) {
    case 0: hipLaunchKernelGGL(dmpp::k_export_grid<0>, dim3(1), dim3(dmpp::kSearchBlock), dyn, h->stream, c, scene, budget, h->d_in, obs_now, h->d_grid, h->d_gridbad); break;
    case 1: hipLaunchKernelGGL(dmpp::k_export_grid<1>, dim3(1), dim3(dmpp::kSearchBlock), dyn, h->stream, c, scene, budget, h->d_in, obs_now, h->d_grid, h->d_gridbad); break;
    default: hipLaunchKernelGGL(dmpp::k_export_grid<2>, dim3(1), dim3(dmpp::kSearchBlock), dyn, h->stream, c, scene, budget, h->d_in, obs_now, h->d_grid, h->d_gridbad); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bad, h->d_gridbad, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bad[0]) return fail(PP_ERR_STATE, "pp_get_grid: the row-major and column-major views of the scene differ in " + std::to_string(bad[0]) + " cell(s)");
    return fetch(h, grid, h->d_grid, N);
}
int pp_get_search_info(pp_handle h, int32_t* lds_budget_words, int32_t* need_words, int32_t* dense_scenes)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    if (!h->d_ovf[0]) return fail(PP_ERR_STATE, "handle was created without the grid stage");
    if (lds_budget_words) *lds_budget_words = h->search_force_gbm ? 0 : h->lds_budget;
    if (need_words) {                      // what the densest scene of the last tick needed (its scoring pass has stored it by now)
        HIP_TRY(hipSetDevice(h->device));
        { int r = join_all(h); if (r) return r; }
        HIP_TRY(hipStreamSynchronize(h->stream));
        const int32_t v = reinterpret_cast<volatile int32_t*>(h->h_need)[h->parity];
        if (v >= 0) h->need_seen = v;
        *need_words = h->need_seen;
    }
    if (dense_scenes) {
        std::vector<int32_t> ovf((size_t)std::max(h->n_scenes, 1));
        int r = fetch(h, ovf.data(), h->d_ovf[h->parity], (size_t)h->n_scenes * sizeof(int32_t)); if (r) return r;
        int k = 0; for (int i = 0; i < h->n_scenes; i++) k += ovf[(size_t)i] != 0;
        *dense_scenes = k;
    }
    return PP_OK;
}
int pp_get_order(pp_handle h, int scene, int32_t* order, int cap)
{
    if (!h || !order || !h->d_order[0]) return fail(PP_ERR_STATE, "expansion order was not requested (caps.order_cap == 0)");
    if (scene < 0 || scene >= h->n_scenes) return fail(PP_ERR_ARG, "scene out of range");
    if (cap > h->caps.order_cap) cap = h->caps.order_cap;
    return fetch(h, order, h->d_order[h->parity] + (size_t)scene * h->caps.order_cap, (size_t)cap * sizeof(int32_t));
}
int pp_get_refpath(pp_handle h, int scene, GlobalPoint2D* pts, int cap)
{
    if (!h || !pts) return fail(PP_ERR_ARG, "null argument");
    if (scene < 0 || scene >= h->n_scenes) return fail(PP_ERR_ARG, "scene out of range");
    if (cap > DMPP_MAX_REFPATH) cap = DMPP_MAX_REFPATH;
    return fetch(h, pts, h->d_dec_ref + (size_t)scene * DMPP_MAX_REFPATH, (size_t)cap * sizeof(GlobalPoint2D));
}
int pp_get_path(pp_handle h, int scene, int32_t* path, int cap)
{
    if (!h || !path || !h->d_path[0]) return fail(PP_ERR_ARG, "no path buffer");
    if (scene < 0 || scene >= h->n_scenes) return fail(PP_ERR_ARG, "scene out of range");
    if (cap > h->cfg.max_path) cap = h->cfg.max_path;
    return fetch(h, path, h->d_path[h->path_set] + (size_t)scene * h->cfg.max_path, (size_t)cap * sizeof(int32_t));
}

int pp_plan_tick_batch(pp_handle h, int n_scenes, const SceneIn* in, const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total,
                       const GlobalPoint3D* lane_pool, const uint8_t* lane_attr_pool, int n_lane_pts,
                       const GlobalPoint2D* ref_pool, int n_ref_pts, SceneState* state_inout, PlanOut* out, GridOut* grid_out)
{
    if (!h || !state_inout || !out) return fail(PP_ERR_ARG, "null argument");
    int r;
    if ((r = pp_set_scenes(h, n_scenes, in, lane_pool, lane_attr_pool, n_lane_pts, ref_pool, n_ref_pts, obs_pool, mot_pool, n_obs_total))) return r;
    if ((r = pp_set_state(h, state_inout, n_scenes))) return r;
    if ((r = pp_plan_tick(h))) return r;
    if ((r = pp_get_plan(h, out, n_scenes))) return r;
    if ((r = pp_get_state(h, state_inout, n_scenes))) return r;
    if (grid_out && h->cfg.grid_stage && (r = pp_get_grid_out(h, grid_out, n_scenes))) return r;
    return PP_OK;
}

// ---------------------------------------------------------------------------------------
// Streamed ticks: new inputs every tick, results out every tick, no host wait in between.
// The reference reads its blackboard at the top of every tick (Planning.cpp:95-112, Decision.cpp:155-160) and publishes at
// the end of it (Planning.cpp:186,214; Decision.cpp:203).

static int ensure_streaming(pp_handle h)
{
    if (h->streaming) return PP_OK;
    const size_t ns = (size_t)h->caps.max_scenes, no = (size_t)(h->caps.max_obs_total > 0 ? h->caps.max_obs_total : 1);
    int r;
    for (int q = 0; q < kIn; q++) {
        InputSet& I = h->in_sets[q];
        if (!I.d_in && (r = dmalloc(&I.d_in, ns))) return r;
        if (!I.d_obs && (r = dmalloc(&I.d_obs, no))) return r;
        if (!I.d_mot) { if ((r = dmalloc(&I.d_mot, no))) return r; HIP_TRY(hipMemsetAsync(I.d_mot, 0, no * sizeof(ObMotion), h->stream)); }
        if (!I.ev_up) HIP_TRY(hipEventCreateWithFlags(&I.ev_up, hipEventDisableTiming));
    }
    for (int q = 1; q < kPlan; q++) if (!h->d_plan_ring[q]) {
        if ((r = dmalloc(&h->d_plan_ring[q], ns))) return r;
        HIP_TRY(hipMemsetAsync(h->d_plan_ring[q], 0, ns * sizeof(PlanOut), h->stream));
    }
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // the upload carries two small kernels (map views, slice check) that the next front chain waits for: top priority
    int io_streams = 3;
    if (const char* e = std::getenv("DMPP_IO_STREAMS")) io_streams = std::atoi(e);       // measurement knob: 1 = one stream for upload and both downloads, 2 = one for the downloads
    if (!h->stream_up) HIP_TRY(hipStreamCreateWithPriority(&h->stream_up, hipStreamNonBlocking, prio_greatest));
    if (!h->stream_dp) { if (io_streams <= 1) h->stream_dp = h->stream_up; else HIP_TRY(hipStreamCreateWithPriority(&h->stream_dp, hipStreamNonBlocking, prio_least)); }
    if (!h->stream_dg) { if (io_streams <= 2) h->stream_dg = h->stream_dp; else HIP_TRY(hipStreamCreateWithPriority(&h->stream_dg, hipStreamNonBlocking, prio_least)); }
    for (int q = 0; q < kPlan; q++) if (!h->ev_fetched_plan[q]) HIP_TRY(hipEventCreateWithFlags(&h->ev_fetched_plan[q], hipEventDisableTiming));
    for (int q = 0; q < kGout; q++) if (!h->ev_fetched_grid[q]) HIP_TRY(hipEventCreateWithFlags(&h->ev_fetched_grid[q], hipEventDisableTiming));
    for (int q = 0; q < kDone; q++) {
        if (!h->ev_done_p[q]) HIP_TRY(hipEventCreateWithFlags(&h->ev_done_p[q], hipEventDisableTiming));
        if (!h->ev_done_g[q]) HIP_TRY(hipEventCreateWithFlags(&h->ev_done_g[q], hipEventDisableTiming));
    }
    if (!h->h_bad) { HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_bad), kDone * sizeof(int32_t), hipHostMallocDefault)); for (int q = 0; q < kDone; q++) h->h_bad[q] = 0; }
    // everything enqueued before streaming began (ticks, the memsets above) is ordered before the three new streams
    { int r2 = join_all(h); if (r2) return r2; }
    hipEvent_t e = get_sync_event(h);
    if (!e) return fail(PP_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(e, h->stream));
    for (hipStream_t st : { h->stream_up, h->stream_dp, h->stream_dg }) HIP_TRY(hipStreamWaitEvent(st, e, 0));
    HIP_TRY(hipStreamSynchronize(h->stream));      // (once per handle) so that e can go back to the pool
    h->sync_events.push_back(e);
    h->streaming = true;
    return PP_OK;
}

int pp_update_async(pp_handle h, int n_scenes, const SceneIn* in, const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    if (n_scenes <= 0 || n_scenes != h->n_scenes)
        return fail(PP_ERR_STATE, "pp_update_async replaces the per-tick inputs of the resident scenes: n_scenes must be the resident count (pp_set_scenes / pp_set_egos come first)");
    if (!in && !obs_pool) return fail(PP_ERR_ARG, "nothing to update");
    if (obs_pool && (n_obs_total < 0 || n_obs_total > h->caps.max_obs_total)) return fail(PP_ERR_CAPACITY, "obstacle pool larger than caps.max_obs_total");
    if (!obs_pool && mot_pool) return fail(PP_ERR_ARG, "a motion pool without its obstacle pool");
    HIP_TRY(hipSetDevice(h->device));
    { int r = ensure_streaming(h); if (r) return r; }
    { int r = pump_fetches(h); if (r) return r; }
    { int r = prune_inflight(h); if (r) return r; }
    const InputSet& C = h->in_sets[h->in_cur];
    const int s = h->in_staged >= 0 ? h->in_staged : (h->in_cur + 1) % kIn;
    // what the update leaves alone is carried over from the set it replaces (the staged one if there is one)
    const InputSet& P = h->in_staged >= 0 ? h->in_sets[h->in_staged] : C;
    InputSet& I = h->in_sets[s];
    hipStream_t su = h->stream_up;
    for (const TickRec& r : h->inflight) if (r.in_set == s) {      // the ticks that still read set s (a whole ring ago: long finished, as a rule)
        HIP_TRY(hipStreamWaitEvent(su, r.ev_front, 0));
        if (r.ev_tail) HIP_TRY(hipStreamWaitEvent(su, r.ev_tail, 0));
    }
    const int n = n_scenes;
    if (in) HIP_TRY(hipMemcpyAsync(I.d_in, in, (size_t)n * sizeof(SceneIn), hipMemcpyDefault, su));
    else if (&P != &I) HIP_TRY(hipMemcpyAsync(I.d_in, P.d_in, (size_t)n * sizeof(SceneIn), hipMemcpyDeviceToDevice, su));
    int n_obs = n_obs_total; bool have_motion = false;
    if (obs_pool) {
        if (n_obs) HIP_TRY(hipMemcpyAsync(I.d_obs, obs_pool, (size_t)n_obs * sizeof(ObPoint), hipMemcpyDefault, su));
        if (n_obs && mot_pool) { HIP_TRY(hipMemcpyAsync(I.d_mot, mot_pool, (size_t)n_obs * sizeof(ObMotion), hipMemcpyDefault, su)); have_motion = true; }
    } else {
        n_obs = P.n_obs_total; have_motion = P.have_motion;
        if (&P != &I && n_obs) {
            HIP_TRY(hipMemcpyAsync(I.d_obs, P.d_obs, (size_t)n_obs * sizeof(ObPoint), hipMemcpyDeviceToDevice, su));
            if (have_motion) HIP_TRY(hipMemcpyAsync(I.d_mot, P.d_mot, (size_t)n_obs * sizeof(ObMotion), hipMemcpyDeviceToDevice, su));
        }
    }
    // the count of poisoned scenes is written by the two kernels below straight into pinned host memory (the slot of the tick
    // that will adopt this update; zeroed here by the host: that tick is not enqueued yet, nothing else writes the slot) - no
    // memset and no copy command on the upload stream
    int32_t* bad_slot = &h->h_bad[(h->tick_seq + 1) % kDone];
    if (in || s != h->in_staged) *reinterpret_cast<volatile int32_t*>(bad_slot) = 0;      // (a second, obstacles-only update of a staged set keeps the count of the first)
    const dim3 grid((unsigned)((n + dmpp::kBlock - 1) / dmpp::kBlock)), block(dmpp::kBlock);
    if (h->resident_mode == 1 && in)      // egos on the resident map: lane views and junction slices from road / lane numbers
        hipLaunchKernelGGL(dmpp::k_resolve_map, grid, block, 0, su, n, I.d_in, h->map_roads, h->d_map_first, h->d_map_lanes, h->d_attr, h->d_map_width,
                           h->map_junctions, h->d_map_junc, bad_slot);
    hipLaunchKernelGGL(dmpp::k_sanitise_scenes, grid, block, 0, su, n, I.d_in, n_obs, h->n_lane_pts, h->n_ref_pts, bad_slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(I.ev_up, su));
    I.up_recorded = true; I.have_motion = have_motion; I.n_obs_total = n_obs;
    h->in_staged = s;
    return PP_OK;
}

static int fetch_async(pp_handle h, PlanOut* plan, PlanningOut* result, PlanningStatus* show, GridOut* grid, long long* tick_id)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    if (!plan && !grid && !result && !show) return fail(PP_ERR_ARG, "nothing to fetch");
    if (h->tick_seq == 0 || h->n_scenes <= 0) return fail(PP_ERR_STATE, "pp_fetch_async: no tick has been enqueued");
    if (grid && !h->cfg.grid_stage) return fail(PP_ERR_STATE, "pp_fetch_async: the last tick ran without the grid stage");
    HIP_TRY(hipSetDevice(h->device));
    { int r = ensure_streaming(h); if (r) return r; }
    const long long T = h->tick_seq;
    const int slot = (int)(T % kDone), n = h->n_scenes;
    if (h->last_rec.tick != T) {
        // the last tick was enqueued before streaming began: one event behind everything stands in for its two
        { int r = join_all(h); if (r) return r; }
        hipEvent_t e = get_sync_event(h);
        if (!e) return fail(PP_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(e, h->stream));
        TickRec rec = { T, h->in_cur, e, nullptr };
        h->inflight.push_back(rec); h->last_rec = rec;
    }
    const TickRec& R = h->last_rec;
    if (h->done_tick[slot] != T) { h->done_tick[slot] = T; h->done_g_rec[slot] = false; h->done_p_rec[slot] = false; }
    const bool want_plan = plan || result || show;
    PendingFetch f{};
    f.tick = T; f.slot = slot; f.n = (size_t)n; f.ev_front = R.ev_front; f.ev_tail = R.ev_tail;
    if (want_plan) { f.plan_dst = plan; f.res_dst = result; f.show_dst = show; f.plan_src = h->d_plan; f.plan_set = h->plan_cur; h->done_p_rec[slot] = true; }
    if (grid) { f.grid_dst = grid; f.grid_src = h->d_gout[h->gout_set]; f.grid_set = h->gout_set; h->done_g_rec[slot] = true; }
    // a second request for the same tick (PlanOut and GridOut asked for separately) joins the first
    bool joined = false;
    for (PendingFetch& g : h->fetches) if (g.tick == T) {
        if (want_plan && !g.plan_dst && !g.res_dst && !g.show_dst) { g.plan_dst = f.plan_dst; g.res_dst = f.res_dst; g.show_dst = f.show_dst; g.plan_src = f.plan_src; g.plan_set = f.plan_set; g.plan_issued = false; joined = true; }
        if (grid && !g.grid_dst) { g.grid_dst = f.grid_dst; g.grid_src = f.grid_src; g.grid_set = f.grid_set; g.grid_issued = false; joined = true; }
    }
    if (!joined) h->fetches.push_back(f);
    { int r = pump_fetches(h); if (r) return r; }
    if (tick_id) *tick_id = T;
    return PP_OK;
}

int pp_fetch_async(pp_handle h, PlanOut* plan, GridOut* grid, long long* tick_id) { return fetch_async(h, plan, nullptr, nullptr, grid, tick_id); }
int pp_fetch_published_async(pp_handle h, PlanningOut* result, PlanningStatus* show, GridOut* grid, long long* tick_id)
{ return fetch_async(h, nullptr, result, show, grid, tick_id); }

long long pp_tick_id(pp_handle h) { return h ? h->tick_seq : -1; }

int pp_wait_tick(pp_handle h, long long tick_id, int* n_poisoned)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    if (n_poisoned) *n_poisoned = 0;
    if (tick_id <= 0 || tick_id > h->tick_seq) return fail(PP_ERR_ARG, "pp_wait_tick: no such tick");
    const int slot = (int)(tick_id % kDone);
    if (!h->streaming || h->done_tick[slot] != tick_id)
        return fail(PP_ERR_ARG, "pp_wait_tick: no pp_fetch_async was issued for that tick (or more than " + std::to_string(kDone) + " ticks ago)");
    HIP_TRY(hipSetDevice(h->device));
    { int r = pump_fetches(h, tick_id); if (r) return r; }
    if (h->done_p_rec[slot]) HIP_TRY(hipEventSynchronize(h->ev_done_p[slot]));
    if (h->done_g_rec[slot]) HIP_TRY(hipEventSynchronize(h->ev_done_g[slot]));
    const int bad = (int)reinterpret_cast<volatile int32_t*>(h->h_bad)[slot];
    if (n_poisoned) *n_poisoned = bad;
    if (bad) return fail(PP_ERR_ARG, "tick " + std::to_string(tick_id) + ": " + std::to_string(bad) + " scene(s) of its update had a slice outside its pool (or a road / lane outside the map) and ran with empty inputs");
    return PP_OK;
}

int pp_tick_io(pp_handle h, PpSceneIo* io)
{
    if (!h || !io) return fail(PP_ERR_ARG, "null argument");
    if (h->n_scenes != 1) return fail(PP_ERR_STATE, "pp_tick_io moves ONE resident scene (pp_set_scenes with n_scenes = 1 comes first: its lane pool stays)");
    if ((io->want & PP_IO_WANT_GRID) && !h->cfg.grid_stage) return fail(PP_ERR_STATE, "PP_IO_WANT_GRID: the handle's configuration has the grid stage off");
    HIP_TRY(hipSetDevice(h->device));
    const int max_obs = std::min(PP_IO_MAX_OBS, h->caps.max_obs_total), max_ref = std::min(DMPP_MAX_REFPATH, h->caps.max_ref_pts_total);
    h->in_staged = -1;
    hipLaunchKernelGGL(dmpp::k_io_in, dim3(1), dim3(dmpp::kBlock), 0, h->stream, io, max_obs, max_ref, h->n_lane_pts, h->d_in, h->d_state, h->d_obs, h->d_ref);
    HIP_TRY(hipGetLastError());
    h->n_obs_total = std::min(std::max((int)io->n_obs, 0), max_obs); h->have_motion = false;
    h->n_ref_pts = std::max(h->n_ref_pts, max_ref);
    note_current_set(h);
    { int r = pp_plan_tick(h); if (r) return r; }
    if (h->last_piped) { int r = join_all(h); if (r) return r; }      // (a one-scene tick ends on the handle's stream, as a rule: nothing to join)
    hipLaunchKernelGGL(dmpp::k_io_out, dim3(1), dim3(dmpp::kBlock), 0, h->stream, io, (int)io->want, h->d_plan, h->d_state,
                       h->cfg.grid_stage ? h->d_gout[h->gout_set] : nullptr, h->d_dec_ref);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (reinterpret_cast<volatile int32_t*>(&io->status)[0]) return fail(PP_ERR_ARG, "pp_tick_io: a lane slice of io->in lies outside the resident lane pool (the scene ran with empty lanes)");
    return PP_OK;
}

#ifdef DMPP_DEBUG_SEARCH
int pp_debug_score_counters(int* out8, int reset)          // debug build only (tools/dbg_score.py)
{
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(dmpp::g_dbg_score), 8 * sizeof(int)));
    if (reset) { int z[8] = { 0 }; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dmpp::g_dbg_score), z, sizeof(z))); }
    return PP_OK;
}
#endif

void* pp_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g_err = "hipHostMalloc failed"; return nullptr; }
    return p;
}
void pp_host_free(void* p) { if (p) (void)hipHostFree(p); }
int pp_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return fail(PP_ERR_ARG, "null argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return PP_OK;
}
int pp_host_unregister(void* p) { if (!p) return PP_OK; HIP_TRY(hipHostUnregister(p)); return PP_OK; }

// ---------------------------------------------------------------------------------------
// stand-alone operators
}  // extern "C"

namespace dmpp {

constexpr int kSoMaxPts = 2048;

__global__ void __launch_bounds__(DMPP_WAVE)
k_search_obstacle_batch(PlannerConfig c, int nq, const GlobalPoint2D* __restrict__ paths, const int32_t* __restrict__ path_off,
                        const ObPoint* __restrict__ obs, const int32_t* __restrict__ obs_off, const double* __restrict__ lo,
                        const double* __restrict__ hi, Path_Obs* __restrict__ out)
{
    __shared__ double s[kSoMaxPts];
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int lane = threadIdx.x;
    const int p0 = path_off[q], n = path_off[q + 1] - p0, o0 = obs_off[q], m = obs_off[q + 1] - o0;
    SoResult r = wave_search_obstacle(c, paths + p0, n, s, obs + o0, m, lo[q], hi[q], lane);
    if (lane == 0) store_path_obs(&out[q], r, obs + o0, true);
}

__global__ void k_geom_batch(PlannerConfig c, int op, int n, const GlobalPoint2D* a, const GlobalPoint2D* b, const GlobalPoint2D* cc, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 0) out[i] = GetLatDis(c, a[i], b[i], cc[i]);
    else if (op == 1) out[i] = GetRoadAngle(c, a[i], b[i]);
    else if (op == 2) out[i] = GetAngleErr(a[i].x, a[i].y);
    else if (op == 3) out[i] = CalcDistance(a[i], b[i]);
    else if (op == 4) out[i] = c.wgs_lat0 + a[i].y * c.wgs_deg_per_m_lat;      // GlobalToWGS84 .lat
    else out[i] = c.wgs_lng0 + a[i].x * c.wgs_deg_per_m_lng;                   // GlobalToWGS84 .lng
}

// One scalar stage of the planning tick on explicit arguments (the CPlanning methods of the same name).
//   op 0 UpdatePlanJudge : in = {last_behavior, behavior, pos, path_lat_dis, path_dir_err, remain_dis} -> out = {afresh, cause}
//   op 1 SpeedPlanning   : in = {pos, ob_flag, mindist_lon, faraim_dis, velocity_expect, brake_speed, acc_flag, des_acc} -> out = {brake_speed, acc_flag, des_acc}
//   op 2 CalculateRadius : in = {path_near_id, path_front_near_id}, pts = last_Bpoints[200] -> out = {radius}
__global__ void k_scalar_stage(PlannerConfig c, int op, const double* in, const GlobalPoint2D* pts, double* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (op == 0) {
        int cause = 0;
        const int afresh = d_UpdatePlanJudge(c, (int)in[0], (int)in[1], (int)in[2], in[3], in[4], in[5], cause);
        out[0] = afresh; out[1] = cause;
    } else if (op == 1) {
        double bs = in[5], da = in[7]; int af = (int)in[6];
        d_SpeedPlanning((int)in[0], (int)in[1], in[2], (float)in[3], in[4], bs, af, da);
        out[0] = bs; out[1] = af; out[2] = da;
    } else {
        out[0] = d_CalculateRadius(pts, (int)in[0], (int)in[1]);
    }
}

__global__ void k_bezier(PlannerConfig c, GlobalPoint3D s, GlobalPoint3D e, GlobalPoint2D* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Bezier bz = bezier_setup(c, s, e);
    out[i] = bezier_point(bz, i, n);
}

__global__ void __launch_bounds__(DMPP_WAVE)
k_cumlen(const GlobalPoint2D* in, int n, double* cum) { wave_cumlen(in, n, cum, threadIdx.x); }

__global__ void k_mean_points(PlannerConfig c, const GlobalPoint2D* in, const double* cum, int n_in, GlobalPoint2D* out, int n_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_out) out[k] = mean_point(c, in, cum, n_in, k, n_out);
}

__global__ void k_create_new_path(PlannerConfig c, const GlobalPoint2D* path, int n, double offset, GlobalPoint2D* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = offset_point(c, path, n, i, offset);
}

}  // namespace dmpp

extern "C" {

static int need_scratch(pp_handle h, size_t bytes)
{
    if (bytes <= h->scratch_bytes) return PP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_scratch) HIP_TRY(hipFree(h->d_scratch));
    h->d_scratch = nullptr; h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->d_scratch, bytes));
    h->scratch_bytes = bytes;
    return PP_OK;
}
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

int pp_search_obstacle_batch(pp_handle h, int nq, const GlobalPoint2D* paths, const int32_t* path_off, const ObPoint* obs,
                             const int32_t* obs_off, const double* lat_lo, const double* lat_hi, Path_Obs* out)
{
    if (!h || !paths || !path_off || !obs_off || !lat_lo || !lat_hi || !out) return fail(PP_ERR_ARG, "null argument");
    if (nq <= 0) return PP_OK;
    HIP_TRY(hipSetDevice(h->device));
    // offsets are small host arrays by contract here (they size the copies)
    const int np = path_off[nq], no = obs_off[nq];
    for (int q = 0; q < nq; q++) if (path_off[q + 1] - path_off[q] > dmpp::kSoMaxPts) return fail(PP_ERR_CAPACITY, "a path longer than 2048 points");
    size_t o_paths = 0, o_poff = o_paths + al256((size_t)np * sizeof(GlobalPoint2D)), o_obs = o_poff + al256((size_t)(nq + 1) * 4),
           o_ooff = o_obs + al256((size_t)no * sizeof(ObPoint)), o_lo = o_ooff + al256((size_t)(nq + 1) * 4),
           o_hi = o_lo + al256((size_t)nq * 8), o_out = o_hi + al256((size_t)nq * 8), total = o_out + al256((size_t)nq * sizeof(Path_Obs));
    int r = need_scratch(h, total); if (r) return r;
    char* d = (char*)h->d_scratch;
    if (np) HIP_TRY(hipMemcpyAsync(d + o_paths, paths, (size_t)np * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    HIP_TRY(hipMemcpyAsync(d + o_poff, path_off, (size_t)(nq + 1) * 4, hipMemcpyDefault, h->stream));
    if (no && obs) HIP_TRY(hipMemcpyAsync(d + o_obs, obs, (size_t)no * sizeof(ObPoint), hipMemcpyDefault, h->stream));
    HIP_TRY(hipMemcpyAsync(d + o_ooff, obs_off, (size_t)(nq + 1) * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(hipMemcpyAsync(d + o_lo, lat_lo, (size_t)nq * 8, hipMemcpyDefault, h->stream));
    HIP_TRY(hipMemcpyAsync(d + o_hi, lat_hi, (size_t)nq * 8, hipMemcpyDefault, h->stream));
    hipLaunchKernelGGL(dmpp::k_search_obstacle_batch, dim3(nq), dim3(DMPP_WAVE), 0, h->stream, h->cfg, nq,
                       (const GlobalPoint2D*)(d + o_paths), (const int32_t*)(d + o_poff), (const ObPoint*)(d + o_obs),
                       (const int32_t*)(d + o_ooff), (const double*)(d + o_lo), (const double*)(d + o_hi), (Path_Obs*)(d + o_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d + o_out, (size_t)nq * sizeof(Path_Obs), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_geom_batch(pp_handle h, int op, int n, const GlobalPoint2D* a, const GlobalPoint2D* b, const GlobalPoint2D* c3, double* out)
{
    if (!h || !a || !out || op < 0 || op > 5) return fail(PP_ERR_ARG, "bad argument");
    if (((op <= 1 || op == 3) && !b) || (op == 0 && !c3)) return fail(PP_ERR_ARG, "missing operand");
    if (n <= 0) return PP_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t pb = al256((size_t)n * sizeof(GlobalPoint2D));
    int r = need_scratch(h, 3 * pb + al256((size_t)n * 8)); if (r) return r;
    char* d = (char*)h->d_scratch;
    HIP_TRY(hipMemcpyAsync(d, a, (size_t)n * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    if (b) HIP_TRY(hipMemcpyAsync(d + pb, b, (size_t)n * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    if (c3) HIP_TRY(hipMemcpyAsync(d + 2 * pb, c3, (size_t)n * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    hipLaunchKernelGGL(dmpp::k_geom_batch, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->cfg, op, n, (const GlobalPoint2D*)d,
                       (const GlobalPoint2D*)(d + pb), (const GlobalPoint2D*)(d + 2 * pb), (double*)(d + 3 * pb));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d + 3 * pb, (size_t)n * 8, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_scalar_stage(pp_handle h, int op, const double* in, int n_in, const GlobalPoint2D* last_Bpoints, double* out, int n_out)
{
    if (!h || !in || !out || op < 0 || op > 2 || n_in < 1 || n_in > 8 || n_out < 1 || n_out > 4) return fail(PP_ERR_ARG, "bad argument");
    if (op == 2 && !last_Bpoints) return fail(PP_ERR_ARG, "CalculateRadius needs the 200 path points");
    HIP_TRY(hipSetDevice(h->device));
    int r = need_scratch(h, 4096 + DMPP_PATH_POINTS * sizeof(GlobalPoint2D)); if (r) return r;
    char* d = (char*)h->d_scratch;
    HIP_TRY(hipMemcpyAsync(d, in, (size_t)n_in * 8, hipMemcpyDefault, h->stream));
    if (op == 2) HIP_TRY(hipMemcpyAsync(d + 4096, last_Bpoints, DMPP_PATH_POINTS * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    hipLaunchKernelGGL(dmpp::k_scalar_stage, dim3(1), dim3(64), 0, h->stream, h->cfg, op, (const double*)d,
                       (const GlobalPoint2D*)(d + 4096), (double*)(d + 2048));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d + 2048, (size_t)n_out * 8, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_bezier(pp_handle h, GlobalPoint3D s, GlobalPoint3D e, GlobalPoint2D* out, int n)
{
    if (!h || !out || n <= 0) return fail(PP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    int r = need_scratch(h, (size_t)n * sizeof(GlobalPoint2D)); if (r) return r;
    hipLaunchKernelGGL(dmpp::k_bezier, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->cfg, s, e, (GlobalPoint2D*)h->d_scratch, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->d_scratch, (size_t)n * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_mean_points(pp_handle h, const GlobalPoint2D* in, int n_in, GlobalPoint2D* out, int n_out)
{
    if (!h || !out || n_out <= 0 || n_in < 0 || (n_in > 0 && !in)) return fail(PP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t ib = al256((size_t)(n_in > 0 ? n_in : 1) * sizeof(GlobalPoint2D)), cb = al256((size_t)(n_in > 0 ? n_in : 1) * 8);
    int r = need_scratch(h, ib + cb + (size_t)n_out * sizeof(GlobalPoint2D)); if (r) return r;
    char* d = (char*)h->d_scratch;
    if (n_in) HIP_TRY(hipMemcpyAsync(d, in, (size_t)n_in * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    hipLaunchKernelGGL(dmpp::k_cumlen, dim3(1), dim3(DMPP_WAVE), 0, h->stream, (const GlobalPoint2D*)d, n_in, (double*)(d + ib));
    hipLaunchKernelGGL(dmpp::k_mean_points, dim3((n_out + 255) / 256), dim3(256), 0, h->stream, h->cfg, (const GlobalPoint2D*)d,
                       (const double*)(d + ib), n_in, (GlobalPoint2D*)(d + ib + cb), n_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d + ib + cb, (size_t)n_out * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

int pp_create_new_path(pp_handle h, const GlobalPoint2D* path, int n, double offset, GlobalPoint2D* out)
{
    if (!h || !path || !out || n < 0) return fail(PP_ERR_ARG, "bad argument");
    if (n == 0) return PP_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t pb = al256((size_t)n * sizeof(GlobalPoint2D));
    int r = need_scratch(h, 2 * pb); if (r) return r;
    char* d = (char*)h->d_scratch;
    HIP_TRY(hipMemcpyAsync(d, path, (size_t)n * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    hipLaunchKernelGGL(dmpp::k_create_new_path, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->cfg, (const GlobalPoint2D*)d, n, offset,
                       (GlobalPoint2D*)(d + pb));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d + pb, (size_t)n * sizeof(GlobalPoint2D), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
}

// ---------------------------------------------------------------------------------------
int pp_set_profile(pp_handle h, int on)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    int r = drain_events(h); if (r) return r;
    h->profile = on < 0 ? 0 : (on > 2 ? 1 : on);
    return PP_OK;
}
int pp_get_kernel_ms(pp_handle h, int k, float* ms_total, int* launches)
{
    if (!h || k < 0 || k >= PP_K_COUNT) return fail(PP_ERR_ARG, "bad argument");
    int r = drain_events(h); if (r) return r;
    if (ms_total) *ms_total = h->k_ms[k];
    if (launches) *launches = h->k_launches[k];
    return PP_OK;
}
int pp_reset_kernel_ms(pp_handle h)
{
    if (!h) return fail(PP_ERR_ARG, "null handle");
    int r = drain_events(h); if (r) return r;
    for (int k = 0; k < PP_K_COUNT; k++) { h->k_ms[k] = 0; h->k_launches[k] = 0; }
    return PP_OK;
}
void* pp_device_ptr(pp_handle h, int which, size_t* bytes)
{
    if (!h) return nullptr;
    const size_t ns = (size_t)h->caps.max_scenes;
    void* p = nullptr; size_t b = 0;
    switch (which) {
    case PP_BUF_SCENE_IN: p = h->d_in; b = ns * sizeof(SceneIn); break;
    case PP_BUF_LANE_POOL: p = h->d_lane; b = (size_t)h->caps.max_lane_pts_total * sizeof(GlobalPoint3D); break;
    case PP_BUF_REF_POOL: p = h->d_ref; b = (size_t)h->caps.max_ref_pts_total * sizeof(GlobalPoint2D); break;
    case PP_BUF_OBS_POOL: p = h->d_obs; b = (size_t)h->caps.max_obs_total * sizeof(ObPoint); break;
    case PP_BUF_MOT_POOL: p = h->d_mot; b = (size_t)h->caps.max_obs_total * sizeof(ObMotion); break;
    case PP_BUF_STATE: p = h->d_state; b = ns * sizeof(SceneState); break;
    case PP_BUF_PLAN_OUT: p = h->d_plan; b = ns * sizeof(PlanOut); break;
    case PP_BUF_GRID_OUT: p = h->d_gout[h->gout_set]; b = ns * sizeof(GridOut); break;      // the buffers of the last tick
    case PP_BUF_GRID: p = nullptr; b = 0; break;      // no occupancy grid is kept after a tick (the search builds it in LDS): use pp_get_grid
    case PP_BUF_PATH: p = h->d_path[h->path_set]; b = ns * (size_t)h->max_path0 * 4; break;
    case PP_BUF_LANE_ATTR: p = h->d_attr; b = (size_t)h->caps.max_lane_pts_total; break;
    case PP_BUF_ORDER: p = h->d_order[h->parity]; b = ns * (size_t)h->caps.order_cap * 4; break;
    default: break;
    }
    if (bytes) *bytes = b;
    return p;
}
void* pp_stream(pp_handle h) { return h ? (void*)h->stream : nullptr; }


size_t pp_sizeof(int which)
{
    switch (which) {
    case 0: return sizeof(PlannerConfig); case 1: return sizeof(PlannerCaps); case 2: return sizeof(SceneIn);
    case 3: return sizeof(SceneState); case 4: return sizeof(PlanOut); case 5: return sizeof(GridOut);
    case 6: return sizeof(ObPoint); case 7: return sizeof(ObMotion); case 8: return sizeof(Path_Obs);
    case 9: return sizeof(LocationOut); case 10: return sizeof(DecisionOutPod); case 11: return sizeof(LaneView);
    case 12: return sizeof(PlanningOut); case 13: return sizeof(PlanningStatus); case 14: return sizeof(AimPoint);
    case 15: return sizeof(MapLane); case 16: return sizeof(MapJunction); case 17: return sizeof(MapDesc); case 18: return sizeof(PpSceneIo);
    default: return 0;
    }
}

}  // extern "C"
