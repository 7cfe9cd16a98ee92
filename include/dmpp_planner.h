/*
 * dmpp_planner.h — C-ABI of the MI355X planning engine (libdmpp.so).
 *
 * The reference has no plugin/FFI interface: its boundary is the C++ class surface of
 * CPlanning / CDecision (Planning.h:38-85, Decision.h:106-107), fed from an MFC
 * blackboard (Planning.cpp:95-112) and drained into it (Planning.cpp:186,214).  This
 * header is what a binding for that path binds instead: plain pointers
 * and sizes, POD structs from dmpp_types.h, int status codes, no C++ or torch types.
 * Every data pointer may be a host pointer or a HIP device pointer (copies use
 * hipMemcpyDefault).  A handle owns its GPU streams and all device scratch; calls on one
 * handle are serialised by the caller (the reference is one thread per module,
 * Planning.cpp:24-39).  All cross-tick state is the caller-visible SceneState.
 *
 * Return value: 0 = ok, negative = error (pp_last_error() has the text).  Nothing here
 * falls back to the CPU: without a usable GPU pp_create fails.
 */
#ifndef DMPP_PLANNER_H
#define DMPP_PLANNER_H

#include "dmpp_types.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK            0
#define PP_ERR_ARG      -1
#define PP_ERR_HIP      -2
#define PP_ERR_CAPACITY -3
#define PP_ERR_STATE    -4

/* strides of the pools filled by pp_gen_scenes */
#define PP_GEN_LANE_PTS 320
#define PP_GEN_REF_PTS  128

typedef struct pp_planner* pp_handle;

typedef struct PlannerCaps {
    int32_t max_scenes;         /* scenes per batch */
    int32_t max_obs_total;      /* obstacle-pool entries */
    int32_t max_lane_pts_total; /* lane-pool points */
    int32_t max_ref_pts_total;  /* refpath-pool points */
    int32_t order_cap;          /* expansion-order cells kept per scene (0 = digest only) */
    int32_t _pad;
} PlannerCaps;

/* kernels of one tick, in launch order; index into pp_get_kernel_ms */
enum { PP_K_OBSTACLES = 0, PP_K_DECISION, PP_K_PLANNING, PP_K_SEARCH, PP_K_SCORE, PP_K_COUNT };
/* device buffers addressable through pp_device_ptr (for RCCL scatter/gather by the caller) */
enum { PP_BUF_SCENE_IN = 0, PP_BUF_LANE_POOL, PP_BUF_REF_POOL, PP_BUF_OBS_POOL, PP_BUF_MOT_POOL, PP_BUF_STATE,
       PP_BUF_PLAN_OUT, PP_BUF_GRID_OUT, PP_BUF_GRID, PP_BUF_PATH, PP_BUF_ORDER, PP_BUF_LANE_ATTR, PP_BUF_COUNT };

const char* pp_last_error(void);

/* ---- host-side helpers (no GPU needed) ------------------------------------------------ */
/* Values for every macro the reference leaves undefined + grid-engine defaults. */
void pp_default_config(PlannerConfig* cfg, int grid_w, int grid_h);
/* Zeroed state with the constructor values of Planning.cpp:10. */
void pp_init_state(SceneState* st, int lane_num);
/* Seeded synthetic scenes.  Pools must hold n*3*PP_GEN_LANE_PTS lane points (and as many
 * lane attribute bytes), n*PP_GEN_REF_PTS refpath points, n*n_obs obstacles
 * (lane_attr_pool/mot_pool/state may be NULL). */
int  pp_gen_scenes(const PlannerConfig* cfg, int first_scene, int n_scenes, int n_obs, int junction_every,
                   SceneIn* in, GlobalPoint3D* lane_pool, uint8_t* lane_attr_pool, GlobalPoint2D* ref_pool,
                   ObPoint* obs_pool, ObMotion* mot_pool, SceneState* state);

/* ---- lifetime -------------------------------------------------------------------------- */
/* Replaces CPlanning::Instance()/CDecision::Instance() + thread start (Planning.cpp:18-33). */
int  pp_create(const PlannerConfig* cfg, int device, const PlannerCaps* caps, pp_handle* out);
int  pp_destroy(pp_handle h);
int  pp_set_config(pp_handle h, const PlannerConfig* cfg);   /* grid size / caps may not grow */

/* ---- resident-data path ------------------------------------------------------------------
 * Replaces the blackboard reads of Planning.cpp:95-112 / Decision.cpp:155-160. */
/* lane_attr_pool[i] is decision_MapData[..][..][id].lanechg_attribute of lane point i (Decision.cpp:1179,1212):
 * one byte per lane-pool point, same indexing.  Required when cfg.lanechg_stage is 1. */
/* Every scene's slices (obs_off/obs_n, lanes.*_off/_n, ref_off/ref_n) are checked on the device against the pool
 * sizes given here; a slice outside its pool is PP_ERR_ARG and leaves no scenes resident (no kernel ever follows it). */
int  pp_set_scenes(pp_handle h, int n_scenes, const SceneIn* in,
                   const GlobalPoint3D* lane_pool, const uint8_t* lane_attr_pool, int n_lane_pts,
                   const GlobalPoint2D* ref_pool, int n_ref_pts,
                   const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total);
int  pp_set_state(pp_handle h, const SceneState* state, int n_scenes);
/* Map store: the whole map once, shared by every scene of the handle - replaces the
 * app->planning_MapData / planning_InterMapData members the threads index (Planning.cpp:331-356,
 * Decision.cpp:346-348,562-578).  Needs caps.max_lane_pts_total >= map->n_points and
 * caps.max_ref_pts_total >= map->n_jpoints.  Host or device pointers. */
int  pp_set_map(pp_handle h, const MapDesc* map);
/* Scenes on the resident map: SceneIn.lanes and SceneIn.ref_off / ref_n are IGNORED on input and derived on the
 * device from loc.road_num / lane_num / id[] (current, left, right lane; lane_sum; lanechg_attribute and
 * lane_width at the ego point) and from loc.last_roadnum / next_roadnum / last_lanenum / next_lanenum (the
 * junction polyline; none -> empty).  A road or lane outside the map is an error (the reference would index
 * out of its vectors). */
int  pp_set_egos(pp_handle h, int n_scenes, const SceneIn* in,
                 const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total);
/* The resident SceneIn records (after pp_set_egos: with the derived lane views). */
int  pp_get_scene_in(pp_handle h, SceneIn* out, int n_scenes);

/* One Decision+Planning(+grid) tick for every resident scene: the bodies of
 * Decision.cpp:172-205 and Planning.cpp:114-223.  Asynchronous: the kernels of a tick run on several streams of the
 * handle (large batches: three chains side by side, consecutive ticks overlapping), so work a caller enqueues on
 * pp_stream(h) is NOT ordered after a tick by itself.  Every pp_get_* / pp_set_* call orders itself after all ticks
 * enqueued so far; for anything else on pp_stream(h) (an RCCL gather out of pp_device_ptr buffers) call pp_join first. */
int  pp_plan_tick(pp_handle h);
/* Makes pp_stream(h) wait, on the device, for every tick enqueued so far; returns at once (no host wait). */
int  pp_join(pp_handle h);
/* pp_join + host wait (and, once streamed ticks are in use, every staged update and download). */
int  pp_sync(pp_handle h);
/* pp_sync + hipDeviceSynchronize: every stream of the handle's device, the caller's included. */
int  pp_device_synchronize(pp_handle h);
/* Replaces SetPlanningStatus/SetUdpSendCtrl (Planning.cpp:186,214) and SetDecisionOut (Decision.cpp:203). */
int  pp_get_plan(pp_handle h, PlanOut* out, int n_scenes);
int  pp_get_state(pp_handle h, SceneState* state, int n_scenes);
int  pp_get_grid_out(pp_handle h, GridOut* out, int n_scenes);
int  pp_get_grid(pp_handle h, int scene, uint8_t* grid);                 /* grid_w*grid_h bytes, 0 free / 1 occupied: rasterised on demand from the last tick's obstacle snapshot by the search's own footprint code */
int  pp_get_order(pp_handle h, int scene, int32_t* order, int cap);      /* needs caps.order_cap > 0 */
int  pp_get_path(pp_handle h, int scene, int32_t* path, int cap);
/* DecisionOut.refpath published by the decision stage (Decision.cpp:195); PlanOut.dec.refpath_n points are valid */
int  pp_get_refpath(pp_handle h, int scene, GlobalPoint2D* pts, int cap);

/* ---- one-shot batch call (the SURVEY §8b signature): upload, tick, download ---------------- */
int  pp_plan_tick_batch(pp_handle h, int n_scenes, const SceneIn* in,
                        const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total,
                        const GlobalPoint3D* lane_pool, const uint8_t* lane_attr_pool, int n_lane_pts,
                        const GlobalPoint2D* ref_pool, int n_ref_pts,
                        SceneState* state_inout, PlanOut* out, GridOut* grid_out /* may be NULL */);

/* ---- streamed ticks: new inputs in, results out, every tick, with no host wait in between --------------------------
 * The reference reads obstacles / location / decision from its blackboard at the top of every tick (Planning.cpp:95-112,
 * Decision.cpp:155-160) and publishes at the end of it (Planning.cpp:186,214; Decision.cpp:203).  pp_set_* / pp_get_* do that
 * with a host wait each, which drains the pipeline of overlapping ticks; the calls below never wait on the host:
 *
 *     pp_update_async(h, n, in_t, obs_t, NULL, n_obs);     // snapshot of tick t (pinned host memory, or device memory)
 *     pp_plan_tick(h);
 *     pp_fetch_async(h, plan_t, grid_t, &id_t);            // into pinned host memory (or device memory)
 *     ... the same for t + 1, t + 2 ...                    // a few ticks deep
 *     pp_wait_tick(h, id_t, NULL);                         // plan_t / grid_t are complete; in_t / obs_t may be reused
 *
 * Inputs are ring-buffered on the device (the searches of three ticks are in flight behind the newest front chain), PlanOut
 * and GridOut too; uploads and downloads run on their own streams beside the kernels.  SceneState stays on the device.
 * Host buffers should come from pp_host_alloc (or be pinned with pp_host_register): copies from / to pageable memory work
 * but are staged by the runtime and wait on the host. */
/* Replaces the per-tick inputs of the resident scenes for the NEXT pp_plan_tick (and the ticks after it, until the next
 * update).  n_scenes must equal the resident count.  `in` NULL: SceneIn records carried over; obs_pool NULL: obstacles (and
 * motion) carried over; mot_pool NULL with an obs_pool: static obstacles.  After pp_set_scenes the SceneIn slices are the
 * caller's (lane and refpath pools stay resident); after pp_set_egos they are derived from the resident map as there.
 * Nothing is refused asynchronously: a scene whose slices fall outside their pools (or whose road / lane is off the map) is
 * POISONED - it runs that tick with empty lanes, refpath and obstacles - and pp_wait_tick reports how many there were.
 * The source buffers must stay untouched until pp_wait_tick of the tick that adopts them (or pp_sync) has returned. */
int  pp_update_async(pp_handle h, int n_scenes, const SceneIn* in, const ObPoint* obs_pool, const ObMotion* mot_pool, int n_obs_total);
/* Downloads PlanOut and / or GridOut (either may be NULL) of the LAST enqueued tick, ordered after that tick's kernels only:
 * PlanOut as soon as its Planning kernel has finished, GridOut after its scoring pass.  *tick_id (may be NULL) names the tick
 * for pp_wait_tick.  The device copies are overwritten 12 ticks later, which those ticks order
 * themselves after - the destination buffers are the caller's to rotate. */
int  pp_fetch_async(pp_handle h, PlanOut* plan, GridOut* grid, long long* tick_id);
/* The same for only what the reference publishes on every tick - PlanningOut (SetUdpSendCtrl, Planning.cpp:214) and
 * PlanningStatus (SetPlanningStatus, Planning.cpp:186) of every scene, as two contiguous arrays (strided copies out of the
 * PlanOut records: 3.3 of their 6.9 KB) - and / or GridOut.  Any of the three may be NULL. */
int  pp_fetch_published_async(pp_handle h, PlanningOut* result, PlanningStatus* show, GridOut* grid, long long* tick_id);
/* Host wait for the downloads of one tick (at most 32 ticks back).  *n_poisoned (may be NULL): scenes of that tick's update
 * that were poisoned; PP_ERR_ARG when there were any (the other scenes' results are valid), PP_OK otherwise. */
int  pp_wait_tick(pp_handle h, long long tick_id, int* n_poisoned);
long long pp_tick_id(pp_handle h);                   /* ticks enqueued on this handle so far = the id of the last one */
/* Pinned host memory for the buffers above (hipHostMalloc / hipHostRegister). */
void* pp_host_alloc(size_t bytes);
void  pp_host_free(void* p);
int   pp_host_register(void* p, size_t bytes);
int   pp_host_unregister(void* p);

/* ---- one scene, one call, one host wait: the latency path of the class surface ------------------------------------------
 * CPlanning::plan(...) / CDecision::decide(...) take everything by value on every call (Planning.h:57-75) and own the
 * cross-tick state as members.  A PpSceneIo block (pinned host memory: pp_host_alloc(sizeof(PpSceneIo))) carries exactly that
 * - location + decision, obstacle list, refpath / junction polyline, state in; PlanOut, state, GridOut, published refpath
 * out - and pp_tick_io moves it with two small kernels that read / write the block over PCIe on the handle's stream, around
 * one pp_plan_tick: no copy command, one host wait.  The handle must hold ONE resident scene (pp_set_scenes: its lane pool
 * stays; io->in.lanes must lie inside it).  io->in.obs_off / obs_n / ref_off / ref_n are set from n_obs / n_ref. */
#define PP_IO_MAX_OBS       1024
#define PP_IO_WANT_GRID     1     /* io->grid is filled (the tick must run the grid stage) */
#define PP_IO_WANT_REFPATH  2     /* io->dec_ref[0 .. plan.dec.refpath_n) is filled (decision stage) */
typedef struct PpSceneIo {
    SceneIn       in;
    SceneState    state;                      /* in / out */
    int32_t       n_obs, n_ref, want, status; /* status out: 0 ok, 1 a lane slice outside the resident pool (scene ran with empty lanes) */
    ObPoint       obs[PP_IO_MAX_OBS];
    GlobalPoint2D ref[DMPP_MAX_REFPATH];
    PlanOut       plan;                       /* out */
    GridOut       grid;                       /* out (PP_IO_WANT_GRID) */
    GlobalPoint2D dec_ref[DMPP_MAX_REFPATH];  /* out (PP_IO_WANT_REFPATH) */
} PpSceneIo;
int  pp_tick_io(pp_handle h, PpSceneIo* io);

/* ---- stand-alone operators on the path ------------------------------------------------------
 * CShare::SearchObstacle (11 call sites, e.g. Planning.cpp:168, Decision.cpp:811): query q
 * uses path points [path_off[q], path_off[q+1]) and obstacles [obs_off[q], obs_off[q+1]). */
int  pp_search_obstacle_batch(pp_handle h, int n_queries,
                              const GlobalPoint2D* paths, const int32_t* path_off,
                              const ObPoint* obs, const int32_t* obs_off,
                              const double* lat_lo, const double* lat_hi, Path_Obs* out);
/* CPlanning::GetLatDis / GetRoadAngle / GetAngleErr (Planning.cpp:686-786), n independent items.
 * op 0: out = GetLatDis(a[i], b[i], c[i]);  op 1: out = GetRoadAngle(a[i], b[i]);
 * op 2: out = GetAngleErr(a[i].x, a[i].y);  op 3: out = CShare::CalcDistance(a[i], b[i]);
 * op 4 / 5: the .lat / .lng of CShare::GlobalToWGS84(a[i]) (Planning.cpp:209). */
int  pp_geom_batch(pp_handle h, int op, int n, const GlobalPoint2D* a, const GlobalPoint2D* b,
                   const GlobalPoint2D* c, double* out);
/* One scalar stage of the planning tick on explicit arguments:
 *  op 0 CPlanning::UpdatePlanJudge (Planning.cpp:797-832): in = {last_behavior, behavior, pos, path_lat_dis,
 *       path_dir_err, remain_dis} -> out = {afresh, cause}
 *  op 1 CPlanning::SpeedPlanning (Planning.cpp:888-990): in = {pos, ob_flag, mindist_lon, faraim_dis,
 *       velocity_expect, brake_speed, acc_flag, des_acc} -> out = {brake_speed, acc_flag, des_acc}
 *  op 2 CPlanning::CalculateRadius (Planning.cpp:1000-1019): in = {path_near_id, path_front_near_id},
 *       last_Bpoints = 200 points -> out = {radius} */
int  pp_scalar_stage(pp_handle h, int op, const double* in, int n_in, const GlobalPoint2D* last_Bpoints,
                     double* out, int n_out);
/* CShare::BezierPlanning / MeanPoints / CreateNewPath on one polyline each (host pointers). */
int  pp_bezier(pp_handle h, GlobalPoint3D start, GlobalPoint3D end, GlobalPoint2D* out, int n);
int  pp_mean_points(pp_handle h, const GlobalPoint2D* in, int n_in, GlobalPoint2D* out, int n_out);
int  pp_create_new_path(pp_handle h, const GlobalPoint2D* path, int n, double offset, GlobalPoint2D* out);

/* ---- measurement / multi-GPU plumbing --------------------------------------------------------- */
/* on = 1: every kernel launch of pp_plan_tick is bracketed by HIP events on the stream it is launched on; on = 2: only the
 * search kernel (an event pair costs a few microseconds of queue time: 2 is what a throughput measurement wants); 0: off. */
int   pp_set_profile(pp_handle h, int on);
/* Sum of event-measured durations (ms) and launch count of kernel k since the last reset. */
int   pp_get_kernel_ms(pp_handle h, int k, float* ms_total, int* launches);
int   pp_reset_kernel_ms(pp_handle h);
/* Raw device pointer + byte size of an internal buffer, so a caller can scatter inputs into /
 * gather results out of it with RCCL without a host hop. */
void* pp_device_ptr(pp_handle h, int which, size_t* bytes);
void* pp_stream(pp_handle h);       /* hipStream_t; ordered after the ticks only after pp_join / pp_sync (see pp_plan_tick) */
/* sizeof of an ABI struct, for bindings to check their mirror: 0 PlannerConfig, 1 PlannerCaps,
 * 2 SceneIn, 3 SceneState, 4 PlanOut, 5 GridOut, 6 ObPoint, 7 ObMotion, 8 Path_Obs, 9 LocationOut,
 * 10 DecisionOutPod, 11 LaneView, 12 PlanningOut, 13 PlanningStatus, 14 AimPoint, 15 MapLane, 16 MapJunction,
 * 17 MapDesc, 18 PpSceneIo */
size_t pp_sizeof(int which);
/* The search keeps a scene's obstacle bitmaps sparse in LDS; a launch gives every scene `lds_budget_words` words per view
 * (sized from what the densest scene of an earlier tick needed) and a scene that needs more is searched on dense bitmaps
 * in HBM instead.  After a tick: the budget of that tick, the words the densest scene seen so far needed, and how many
 * scenes of the last tick took the dense path.  Any pointer may be NULL. */
int   pp_get_search_info(pp_handle h, int32_t* lds_budget_words, int32_t* need_words, int32_t* dense_scenes);
/* After filling the input buffers through pp_device_ptr (an RCCL scatter straight into them): declares what is resident -
 * scenes, the used part of each pool, whether the motion pool / the lane attribute pool were filled - and checks every
 * scene's slices against those pools, like pp_set_scenes does. */
int   pp_set_n_scenes(pp_handle h, int n_scenes, int n_lane_pts, int n_ref_pts, int n_obs_total, int have_motion, int have_lane_attr);

#ifdef __cplusplus
}
#endif
#endif /* DMPP_PLANNER_H */
