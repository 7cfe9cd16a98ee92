// kernels_r.hpp — HIP kernels for the reference-real part of the tick (SURVEY.md §8 rows
// R1-R18): one 256-thread workgroup (4 waves) per scene.
//
//   k_effective_obstacles : obstacle snapshot of this tick (G4 constant-velocity model)
//   k_decision            : CDecision tick — LoadRefPath, AroundObstacle, the lateral-offset
//                           sweep of BehaviorDecision, SpeedDecision, RefPath, and the two
//                           junction handlers (Decision.cpp:216-486,553-673,759-1010,1781-1816)
//   k_planning            : CPlanning tick body (Planning.cpp:114-223)
//
// Layout: everything a scene touches per tick is staged once into LDS (paths as 16-B
// points read as broadcast ds_read_b128, the obstacle list as 24-B records); global reads
// are coalesced point-per-lane copies; the 200-point steps run one point per thread; sums
// whose rounding depends on order (arc lengths) are accumulated by one lane in index order.
#pragma once
#include "dev_geom.hpp"

#ifndef DMPP_FRONT_PRIO
#define DMPP_FRONT_PRIO 3          // wave issue priority of Decision / Planning (experiment knob: -DDMPP_FRONT_PRIO=2)
#endif
namespace dmpp {

constexpr int kBlock = 256;

// Debug build (make debug): cycle stamps of the phases of k_decision / k_planning, written by thread 0 into the unused tail of
// the scene's decision-refpath slots (tools/dbg_front.py reads them with pp_get_refpath).
#ifdef DMPP_DEBUG_SEARCH
#define FRONT_MARK(slot) { if (threadIdx.x == 0) dbg_stamp[slot] = (double)(clock64() - dbg_t0); }
#else
#define FRONT_MARK(slot)
#endif
constexpr int kMaxObsLds = 512;   // obstacle records staged in LDS; longer lists are read from HBM

// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_effective_obstacles(PlannerConfig c, int n_scenes, const SceneIn* __restrict__ in, const SceneState* __restrict__ st,
                      const ObPoint* __restrict__ obs, const ObMotion* __restrict__ mot, ObPoint* __restrict__ now)
{
    const int s = blockIdx.x;
    if (s >= n_scenes) return;
    __builtin_amdgcn_s_setprio(3);             // the front chain is short and every search waits for it: issue ahead of the searching waves
    const int off = in[s].obs_off, m = in[s].obs_n;
    const double t = c.dyn_dt * (double)st[s].tick;
    for (int j = threadIdx.x; j < m; j += kBlock) {
        ObPoint o = obs[off + j];
        if (c.dynamic_obstacles && mot) {
            ObMotion v = mot[off + j];
            o.x = o.x + v.vx * t;
            o.y = o.y + v.vy * t;
        }
        now[off + j] = o;
    }
}

// ---- scalar stages of the planning tick, shared by k_planning and the stand-alone stage operator ----
// UpdatePlanJudge, Planning.cpp:797-832 (reads the members path_lat_dis / path_dir_err / remain_dis)
__device__ inline int d_UpdatePlanJudge(const PlannerConfig& c, int last_behavior, int behavior, int pos,
                                        double path_lat_dis, double path_dir_err, double remain_dis, int& cause)
{
    cause = 0;
    if (last_behavior != behavior) { cause = 1; return 1; }
    if (fabs(path_lat_dis) > 0.2) { cause = 2; return 1; }
    if (fabs(path_dir_err) > 45) { cause = 3; return 1; }
    if ((pos == 0) && remain_dis < c.ROAD_REMAIN_DISTANCE) { cause = 4; return 1; }
    else if (pos != 0 && remain_dis < c.INTER_REMAIN_DISTANCE) { cause = 4; return 1; }
    return 0;
}
// SpeedPlanning, Planning.cpp:888-990 (three identical cases; other pos values leave the outputs alone)
__device__ inline void d_SpeedPlanning(int pos, int ob_flag, double lon, float faraim, double velocity_expect,
                                       double& brakespeed, int& acc_flag, double& des_acc)
{
    if (pos == 0 || pos == 1 || pos == 2) {
        if (ob_flag) {
            if (lon - 4 > 9) { brakespeed = 3 + (lon - 9) / (faraim - 9) * (velocity_expect - 3); acc_flag = 0; des_acc = 0; }
            else if (lon - 4 > 5) { brakespeed = 3; acc_flag = 0; des_acc = 0; }
            else { brakespeed = 0; acc_flag = 1; des_acc = -3; }
        } else { brakespeed = velocity_expect; acc_flag = 0; des_acc = 0; }
    }
}
// CalculateRadius, Planning.cpp:1000-1019 (ids fenced to [0,199]; a NaN sinA gives a NaN radius as in the reference)
__device__ inline double d_CalculateRadius(const GlobalPoint2D* last, int near_id, int front_id)
{
    const int nid = clampi(near_id, 0, 199), f2 = clampi(front_id, 0, 199);
    const unsigned mnum = (unsigned)round((double)((nid + f2) / 2));
    const GlobalPoint2D a = last[nid], mm = last[mnum], f = last[f2];
    double dis1 = sqrt((a.x - mm.x) * (a.x - mm.x) + (a.y - mm.y) * (a.y - mm.y));
    double dis2 = sqrt((mm.x - f.x) * (mm.x - f.x) + (mm.y - f.y) * (mm.y - f.y));
    double dis3 = sqrt((a.x - f.x) * (a.x - f.x) + (a.y - f.y) * (a.y - f.y));
    double dis = dis1 * dis1 + dis2 * dis2 - dis3 * dis3;
    double cosA = dis / (2 * dis1 * dis2);
    double sinA = sqrt(1 - cosA * cosA);
    return (sinA < 0.001) ? 1000 : 0.5 * dis3 / sinA;
}

// Block-wide arc-length walk shared by every branch of SearchAimPoint (Planning.cpp:410-432,
// 448-469,478-499,507-538): accumulate |P[i+1]-P[i]| for i = i0 .. iend-1 in index order and
// stop at the first i with sum - 4 > faraim.  Segment lengths are computed 256 at a time in
// parallel; thread 0 adds them in order, eight per block of reads (serial_walk), so the rounding equals the scalar loop.
// base/stride: point i is (base[i*stride], base[i*stride+1]).  Returns the index or -1.
__device__ inline int block_aim_walk(const double* base, int stride, int i0, int iend, double faraim,
                                     double* seg, int* sh_found, double* sh_sum)
{
    const int tid = threadIdx.x;
    if (i0 < 0) i0 = 0;
    if (tid == 0) { *sh_found = -1; *sh_sum = 0; }
    __syncthreads();
    for (int c0 = i0; c0 < iend; c0 += kBlock) {
        const int i = c0 + tid;
        if (i < iend) {
            double dx = base[(size_t)(i + 1) * stride] - base[(size_t)i * stride];
            double dy = base[(size_t)(i + 1) * stride + 1] - base[(size_t)i * stride + 1];
            seg[tid] = sqrt(dx * dx + dy * dy);
        }
        __syncthreads();
        if (tid == 0) {
            double sum = *sh_sum;
            const int k = serial_walk(seg, iend - c0 < kBlock ? iend - c0 : kBlock, faraim, sum);
            if (k >= 0) *sh_found = c0 + k;
            *sh_sum = sum;
        }
        __syncthreads();
        if (*sh_found >= 0) break;
    }
    return *sh_found;
}

// ---------------------------------------------------------------------------------------
struct DecShared {                                 // 19 KB: small enough to sit beside the searching workgroups of a CU
    union {
        struct {                                    // road segment: the six corridors, two offset candidates, scratch
            GlobalPoint2D F[DMPP_FRONT_POINTS], R[DMPP_REAR_POINTS], LF[DMPP_FRONT_POINTS], LR[DMPP_REAR_POINTS],
                          RF[DMPP_FRONT_POINTS], RR[DMPP_REAR_POINTS];
            GlobalPoint2D tmp[2][DMPP_FRONT_POINTS];    // one offset candidate per team of two waves
            double s0[DMPP_FRONT_POINTS], s1[DMPP_FRONT_POINTS];   // arc-length scratch of the two teams
            double seg[kBlock]; unsigned char sa[kBlock];          // remaining lane-change length: segment lengths + attributes
        } road;
        struct {                                    // (pre-)junction: the front path (<= 512 points) and its arc lengths
            GlobalPoint2D ref[DMPP_MAX_REFPATH];
            double s0[DMPP_MAX_REFPATH];
        } junc;
    } u;
    double part_d2[kBlock]; int part_bi[kBlock];      // team_search_obstacle
    SoResult around[6];
    double sweep_lng[2 * DMPP_MAX_SWEEP];
    int n[6];
    int n_ref, do_sweep, n_cand;
    int rem_go, rem_flags;                        // remaining lane-change length tests (lane-change rule tree)
    int rem_w[2][kBlock / 64]; double rem_s[2][kBlock / 64];   // ... their per-wave run ends and partial sums
};


__device__ inline int dev_load_front(const PlannerConfig& c, const GlobalPoint3D* lane, int IdSum, int Id, GlobalPoint2D* out)
{   // Decision.cpp:581-587
    int lo = min(IdSum, Id + c.ID_MORE), hi = min(IdSum, Id + 120 + c.ID_MORE);
    if (lo < 0) lo = 0;
    int n = hi - lo; if (n < 0) n = 0;
    for (int k = threadIdx.x; k < n; k += kBlock) { out[k].x = lane[lo + k].x; out[k].y = lane[lo + k].y; }
    return n;
}
__device__ inline int dev_load_rear(const PlannerConfig& c, const GlobalPoint3D* lane, int IdSum, int Id, GlobalPoint2D* out)
{   // Decision.cpp:590-596 (walks backwards from the ego point; index IdSum is fenced to IdSum-1)
    int hi = min(IdSum, Id + c.ID_MORE), lo = max(0, Id + c.ID_MORE - 40);
    int n = hi - lo; if (n < 0) n = 0;
    if (IdSum <= 0) n = 0;
    for (int k = threadIdx.x; k < n; k += kBlock) {
        int i = hi - k; int q = i >= IdSum ? IdSum - 1 : i;
        out[k].x = lane[q].x; out[k].y = lane[q].y;
    }
    return n;
}

__device__ inline void store_path_obs(Path_Obs* dst, const SoResult& r, const ObPoint* obs, bool searched)
{
    Path_Obs p;
    p.Ob_Pose.dis_lat = searched ? r.dis_lat : 0; p.Ob_Pose.dis_lng = searched ? r.dis_lng : 0;
    p.Obs_flag = searched ? r.flag : 0; p.Ob_Pathid = searched ? r.path_id : 0;
    if (searched && r.flag) p.Ob_Attr = obs[r.ob_index];
    else { p.Ob_Attr.x = 0; p.Ob_Attr.y = 0; p.Ob_Attr.type = 0; p.Ob_Attr.radius = 0; }
    *dst = p;
}

// ---------------------------------------------------------------------------------------
// Map store -> per-scene views.  One thread per scene does what the reference does with
// app->planning_MapData[road-1][lane-1] / [lane-2] / [lane] (Planning.cpp:331-380, Decision.cpp:562-578) and
// planning_InterMapData[last_road-1][next_road-1][last_lane-1][next_lane-1] (Planning.cpp:342, Decision.cpp:348):
// it writes LaneView and the junction slice into the resident SceneIn.  bad[0] counts scenes off the map.
__global__ void __launch_bounds__(kBlock)
k_resolve_map(int n_scenes, SceneIn* __restrict__ in, int n_roads, const int32_t* __restrict__ road_first_lane,
              const MapLane* __restrict__ lanes, const uint8_t* __restrict__ attr, const uint16_t* __restrict__ width_cm,
              int n_junctions, const MapJunction* __restrict__ junctions, int* __restrict__ bad)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_scenes) return;
    SceneIn& si = in[s];
    const LocationOut loc = si.loc;
    LaneView lv;
    lv.cur_off = lv.cur_n = lv.left_off = lv.left_n = lv.right_off = lv.right_n = 0;
    lv.lane_sum = 0; lv.lanechg_attribute = 0; lv.lane_width = 0;
    const int road = loc.road_num, lane = loc.lane_num;
    bool ok = road >= 1 && road <= n_roads;
    int L0 = 0, L1 = 0;
    if (ok) { L0 = road_first_lane[road - 1]; L1 = road_first_lane[road]; ok = lane >= 1 && lane <= L1 - L0; }
    if (ok) {
        const MapLane cur = lanes[L0 + lane - 1];
        lv.cur_off = cur.point_off; lv.cur_n = cur.n_points; lv.lane_sum = cur.lane_sum;
        if (lane > 1) { const MapLane l = lanes[L0 + lane - 2]; lv.left_off = l.point_off; lv.left_n = l.n_points; }            // Planning.cpp:359-368
        if (lane < cur.lane_sum && L0 + lane < L1) { const MapLane r = lanes[L0 + lane]; lv.right_off = r.point_off; lv.right_n = r.n_points; }   // :371-380
        const int id = clampi(loc.id[clampi(lane - 1, 0, DMPP_LANESUM - 1)], 0, max(cur.n_points - 1, 0));
        if (cur.n_points > 0) {
            lv.lanechg_attribute = attr[cur.point_off + id];                        // Decision.cpp:566
            lv.lane_width = (double)width_cm[cur.point_off + id] / 100.0;           // Decision.cpp:578
        }
    } else atomicAdd_system(bad, 1);       // (bad may be pinned host memory: the streamed path)
    si.lanes = lv;
    int roff = 0, rn = 0;
    for (int j = 0; j < n_junctions; j++) {
        const MapJunction q = junctions[j];
        if (q.last_road == loc.last_roadnum && q.next_road == loc.next_roadnum && q.last_lane == loc.last_lanenum && q.next_lane == loc.next_lanenum) {
            roff = q.point_off; rn = q.n_points; break;
        }
    }
    si.ref_off = roff; si.ref_n = rn;
}

// One thread per scene: every slice a tick follows out of a SceneIn record - obstacles, the three lane views, the
// junction polyline - must lie inside its pool (an empty slice may carry any offset: it is never dereferenced).
// bad[0] counts the scenes that fail; the host refuses the batch (no tick is launched on it).
__global__ void __launch_bounds__(kBlock)
k_validate_scenes(int n_scenes, const SceneIn* __restrict__ in, int n_obs_total, int n_lane_pts, int n_ref_pts, int* __restrict__ bad)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_scenes) return;
    const SceneIn& si = in[s];
    auto inside = [](int off, int n, int pool) { return n == 0 || (n > 0 && off >= 0 && (long long)off + n <= (long long)pool); };
    const bool ok = inside(si.obs_off, si.obs_n, n_obs_total) && inside(si.lanes.cur_off, si.lanes.cur_n, n_lane_pts) &&
                    inside(si.lanes.left_off, si.lanes.left_n, n_lane_pts) && inside(si.lanes.right_off, si.lanes.right_n, n_lane_pts) &&
                    inside(si.ref_off, si.ref_n, n_ref_pts);
    if (!ok) atomicAdd(bad, 1);
}

// Streamed updates (pp_update_async) cannot refuse a batch - nobody waits for the answer - so the same test POISONS a scene
// instead: every slice of a failing scene is made empty (an empty slice is never dereferenced, whatever its offset), the tick
// runs on it like on a scene without lanes and obstacles, and bad[0] counts it; pp_wait_tick reports the count of that tick.
__global__ void __launch_bounds__(kBlock)
k_sanitise_scenes(int n_scenes, SceneIn* __restrict__ in, int n_obs_total, int n_lane_pts, int n_ref_pts, int* __restrict__ bad)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_scenes) return;
    SceneIn& si = in[s];
    auto inside = [](int off, int n, int pool) { return n == 0 || (n > 0 && off >= 0 && (long long)off + n <= (long long)pool); };
    const bool ok = inside(si.obs_off, si.obs_n, n_obs_total) && inside(si.lanes.cur_off, si.lanes.cur_n, n_lane_pts) &&
                    inside(si.lanes.left_off, si.lanes.left_n, n_lane_pts) && inside(si.lanes.right_off, si.lanes.right_n, n_lane_pts) &&
                    inside(si.ref_off, si.ref_n, n_ref_pts);
    if (!ok) {
        si.obs_n = 0; si.lanes.cur_n = 0; si.lanes.left_n = 0; si.lanes.right_n = 0; si.ref_n = 0;
        si.obs_off = 0; si.lanes.cur_off = 0; si.lanes.left_off = 0; si.lanes.right_off = 0; si.ref_off = 0;
        atomicAdd_system(bad, 1);          // (bad may be pinned host memory: the streamed path)
    }
}

// ---- pp_tick_io: the per-call inputs / outputs of ONE scene between a pinned host block and the device buffers ----
// One workgroup; every thread moves 8-byte words (all the records are 8-byte aligned and sized).  The block lives in host
// memory: the loads / stores below cross PCIe, all of them in flight together (no dependent round trips except the counts).
__device__ inline void copy_words(void* __restrict__ dst, const void* __restrict__ src, size_t bytes)
{
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
    for (size_t i = threadIdx.x; i < bytes / 8; i += kBlock) d[i] = s[i];
}
__global__ void __launch_bounds__(kBlock)
k_io_in(PpSceneIo* __restrict__ io, int max_obs, int max_ref, int n_lane_pts, SceneIn* __restrict__ in, SceneState* __restrict__ state,
        ObPoint* __restrict__ obs, GlobalPoint2D* __restrict__ ref)
{
    const int n_obs = min(max(io->n_obs, 0), max_obs), n_ref = min(max(io->n_ref, 0), max_ref);
    copy_words(in, &io->in, sizeof(SceneIn));
    copy_words(state, &io->state, sizeof(SceneState));
    copy_words(obs, io->obs, (size_t)n_obs * sizeof(ObPoint));
    copy_words(ref, io->ref, (size_t)n_ref * sizeof(GlobalPoint2D));
    __syncthreads();
    if (threadIdx.x == 0) {
        SceneIn& si = in[0];
        si.obs_off = 0; si.obs_n = n_obs; si.ref_off = 0; si.ref_n = n_ref;
        auto inside = [](int off, int n, int pool) { return n == 0 || (n > 0 && off >= 0 && (long long)off + n <= (long long)pool); };
        const bool ok = inside(si.lanes.cur_off, si.lanes.cur_n, n_lane_pts) && inside(si.lanes.left_off, si.lanes.left_n, n_lane_pts) &&
                        inside(si.lanes.right_off, si.lanes.right_n, n_lane_pts);
        if (!ok) { si.lanes.cur_n = 0; si.lanes.left_n = 0; si.lanes.right_n = 0; si.lanes.cur_off = 0; si.lanes.left_off = 0; si.lanes.right_off = 0; }
        io->status = ok ? 0 : 1;
    }
}
__global__ void __launch_bounds__(kBlock)
k_io_out(PpSceneIo* __restrict__ io, int want, const PlanOut* __restrict__ plan, const SceneState* __restrict__ state,
         const GridOut* __restrict__ gout, const GlobalPoint2D* __restrict__ dec_ref)
{
    copy_words(&io->plan, plan, sizeof(PlanOut));
    copy_words(&io->state, state, sizeof(SceneState));
    if ((want & PP_IO_WANT_GRID) && gout) copy_words(&io->grid, gout, sizeof(GridOut));
    if (want & PP_IO_WANT_REFPATH) {
        const int n = min(max(plan[0].dec.refpath_n, 0), DMPP_MAX_REFPATH);
        copy_words(io->dec_ref, dec_ref, (size_t)n * sizeof(GlobalPoint2D));
    }
}

// ---------------------------------------------------------------------------------------
// Lane-change rule tree of CDecision::BehaviorDecision, Decision.cpp:1017-1772, run by one thread per scene
// on the corridor distances k_decision has just produced.  `rem` holds the remaining-length tests (see
// k_decision).  The tree's leaves come in three shapes, named here:
//   go(b, t)  : start the lane change          (behavior b, target lane t, lanechg_status 1)
//   keep()    : stay, lanechg_status cleared   (behavior 1, target = current lane, lanechg_status 0)
//   keep_ns() : stay, lanechg_status untouched (the branches of :1623-1634, :1667-1685, :1717-1728)
// Oddities of the reference text are kept and marked "as written".
__device__ inline void lane_change_tree(const SceneIn& si, int LaneNum_Cur, int LaneSum, int Map, unsigned navi, int rem,
                                        double F, double LF, double LR, double RF, double RR,
                                        SceneState& st, Behavior_Dec& Cur)
{
    const double period = si.period_last;
    double left_t = st.leftlight_time, right_t = st.rightlight_time;
    unsigned frontobs = st.frontobs_time;
    auto go = [&](int b, int t) { Cur.behavior = b; Cur.target_lanenum = t; Cur.lanechg_status = 1; };
    auto keep = [&]() { Cur.behavior = 1; Cur.target_lanenum = LaneNum_Cur; Cur.lanechg_status = 0; };
    auto keep_ns = [&]() { Cur.behavior = 1; Cur.target_lanenum = LaneNum_Cur; };
    auto exits_via = [&](int lane_no) {           // is lane_no one of the navigation's exit lanes (:1166-1172)
        bool hit = false;
        for (int i = 0; i < DMPP_LANESUM && si.out_lane_no[i] != 0; i++) hit = hit || ((int)si.out_lane_no[i] == lane_no);
        return hit;
    };
    // signal bookkeeping shared by the obstacle-triggered branches: restart the timer when `restart`, add the period, cap
    auto run_left_timer = [&](bool restart, double cap_to) {
        if (restart) left_t = 0;
        left_t += period;
        if (left_t > 2000) left_t = cap_to;
    };
    const bool A60 = rem & 1, B10 = rem & 2, B15 = rem & 4, B50 = rem & 8;

    if (Cur.lanechg_status == 0) {
        if (navi == 1) {                                                       // :1024 navigation wants the left lane
            if (Map == 1 || Map == 3) {
                Cur.behavior_to_dlg = 2;
                if (Cur.light_status != 1) { Cur.light_status = 1; left_t = 0; }
                left_t += period;
                if ((LF > F + 10 || LF > 40) && LR > 15 && left_t > 2000) go(2, LaneNum_Cur - 1); else keep();
            } else { keep(); Cur.behavior_to_dlg = 4; }
        } else if (navi == 2) {                                                // :1086 navigation wants the right lane
            if (Map == 2 || Map == 3) {
                Cur.behavior_to_dlg = 3;
                if (Cur.light_status != 2) { Cur.lanechg_status = 2; right_t = 0; }   // :1094 as written (status, not light)
                right_t += period;
                if ((RF > F + 10 || RF > 40) && RR > 15 && right_t >= 2000) go(3, LaneNum_Cur + 1); else keep();
            } else { keep(); Cur.behavior_to_dlg = 4; }
        } else if (navi == 0) {                                                // :1146 no navigation demand
            if (F < 25) {                                                      // :1149
                frontobs++;
                if (frontobs > 2) {
                    frontobs = 3;
                    if (Map == 1) {                                            // :1157
                        if (LaneNum_Cur > 1) {
                            Cur.behavior_to_dlg = 5;
                            bool chg;
                            if (!exits_via(LaneNum_Cur - 1)) {                 // must come back: needs 60 m of attribute-1 lane
                                chg = A60;
                                // :1193-1206 test and set the member z_light_status, which :308 then overwrites with
                                // Cur.light_status: only the timer restart survives
                                if (chg) run_left_timer(Cur.light_status != 1, 2000);
                            } else {
                                chg = B15;
                                if (chg) { const bool r = Cur.light_status != 1; Cur.light_status = 1; run_left_timer(r, 2100); }  // :1232 as written
                                else Cur.light_status = 0;
                            }
                            if (chg && LF > F + 10 && LR > 10 && left_t > 1500) { frontobs = 0; go(2, LaneNum_Cur - 1); } else keep();
                        } else keep();
                    } else if (Map == 2) {                                     // :1296
                        if (LaneNum_Cur < LaneSum) {
                            Cur.behavior_to_dlg = 6;
                            bool chg;
                            if (!exits_via(LaneNum_Cur - 1)) {                 // :1307 as written (the left neighbour)
                                chg = B50;
                                if (chg) { const bool r = Cur.light_status != 2; Cur.light_status = 2; run_left_timer(r, 2000); }
                            } else {
                                chg = B10;
                                if (chg) { const bool r = Cur.light_status != 1; Cur.light_status = 1; run_left_timer(r, 2000); }  // :1356 as written
                            }
                            if (chg && RF > F + 10 && RR > 10 && left_t > 1500) { frontobs = 0; go(3, LaneNum_Cur + 1); } else keep();
                        } else keep();
                    } else if (Map == 3) {                                     // :1426
                        const bool back_l = !exits_via(LaneNum_Cur - 1), back_r = !exits_via(LaneNum_Cur + 1);
                        // :1477 parses to a constant-false predicate: a left change that need not come back never qualifies
                        const bool left_ok = (LaneNum_Cur > 1) && back_l && B50;
                        const bool right_ok = (LaneNum_Cur < LaneSum) && (back_r ? B50 : B10);
                        if (left_ok && !back_l) {                              // :1545 (unreachable given the line above; kept)
                            const bool r = Cur.lanechg_status != 1;
                            if (r) Cur.light_status = 1;
                            run_left_timer(r, 2000);
                            if (LF > F + 10 && LR > 10 && left_t > 2000) { frontobs = 0; go(2, LaneNum_Cur - 1); } else keep();
                        } else if (right_ok && !back_r) {                      // :1596
                            const bool r = Cur.light_status != 2; Cur.light_status = 2;
                            run_left_timer(r, 2000);
                            if (RF > F + 10) {                                 // no else in the reference
                                if (RR > 10 && left_t > 2000) { frontobs = 0; go(3, LaneNum_Cur + 1); } else keep_ns();
                            }
                        } else if (left_ok) {                                  // :1638
                            const bool r = Cur.light_status != 1;
                            if (r) Cur.lanechg_status = 1;                     // :1642 as written (status, not light)
                            run_left_timer(r, 2000);
                            if (LF > F + 10 && LR > 10 && left_t > 2000) { frontobs = 0; go(2, LaneNum_Cur - 1); } else keep_ns();
                        } else if (right_ok) {                                 // :1688
                            const bool r = Cur.light_status != 2; Cur.light_status = 2;
                            run_left_timer(r, 2000);
                            if (RF > F + 10) {                                 // no else in the reference
                                if (RR > 10 && left_t > 2000) { frontobs = 0; go(3, 1); }   // :1712 as written: target lane 1
                                else keep_ns();
                            }
                        } else keep();
                    }
                } else keep();                                                 // :1741-1746
            } else { frontobs = 0; Cur.behavior_to_dlg = 8; keep(); }          // :1749-1756
        }
    } else if (Cur.lanechg_status == 1) {                                      // :1760 changing lanes
        Cur.behavior_to_dlg = 9;
        if (Cur.target_lanenum == LaneNum_Cur) Cur.lanechg_status = 0;         // (the light reset of :1766 is overwritten at :1770)
        Cur.behavior = st.d_his_behavior; Cur.target_lanenum = st.d_his_target_lanenum; Cur.light_status = st.d_his_light_status;
    }
    st.leftlight_time = left_t; st.rightlight_time = right_t; st.frontobs_time = frontobs;
}

__global__ void __launch_bounds__(kBlock)
k_decision(PlannerConfig c, int n_scenes, const SceneIn* __restrict__ in, const GlobalPoint3D* __restrict__ lane_pool,
           const uint8_t* __restrict__ attr_pool, const GlobalPoint2D* __restrict__ ref_pool, const ObPoint* __restrict__ obs_now,
           SceneState* __restrict__ state, PlanOut* __restrict__ plan, GlobalPoint2D* __restrict__ dec_ref)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    DecShared& sh = *reinterpret_cast<DecShared*>(smem_raw);
    const int scene = blockIdx.x;
    if (scene >= n_scenes) return;
    __builtin_amdgcn_s_setprio(DMPP_FRONT_PRIO);             // front chain: ahead of the searching waves on the same SIMD
#ifdef DMPP_DEBUG_SEARCH
    const long long dbg_t0 = clock64();
    double* dbg_stamp = reinterpret_cast<double*>(dec_ref + (size_t)blockIdx.x * DMPP_MAX_REFPATH + 480);
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const SceneIn& si = in[scene];
    SceneState& st = state[scene];
    PlanOut& po = plan[scene];
    const LocationOut& loc = si.loc;
    const int m = si.obs_n;
    const ObPoint* obs = obs_now + si.obs_off;       // read where they are: every lane needs its obstacle once per query
    FRONT_MARK(0)
    const double hv = 0.5 * c.Vehicle_Width;
    GlobalPoint2D* out_ref = dec_ref + (size_t)scene * DMPP_MAX_REFPATH;
    const int pos = loc.pos;

    if (pos == 0) {
        // ---- LoadRefPath, Decision.cpp:553-673 ----
        const int LaneNum_Cur = loc.lane_num, LaneSum = si.lanes.lane_sum, LaneChg = si.lanes.lanechg_attribute;
        const int Id_Cur = loc.id[clampi(LaneNum_Cur - 1, 0, DMPP_LANESUM - 1)];
        const double W = si.lanes.lane_width;
        const GlobalPoint3D* cur = lane_pool + si.lanes.cur_off;
        int nF = dev_load_front(c, cur, si.lanes.cur_n, Id_Cur, sh.u.road.F);
        int nR = dev_load_rear(c, cur, si.lanes.cur_n, Id_Cur, sh.u.road.R);
        int nLF = 0, nLR = 0, nRF = 0, nRR = 0;
        int synth_left = 0, synth_right = 0;
        if (LaneChg == 1 || LaneChg == 3) {
            if (LaneNum_Cur > 1) {
                int Id_L = loc.id[clampi(LaneNum_Cur - 2, 0, DMPP_LANESUM - 1)], Sum_L = si.lanes.left_n;
                if (Id_L > 0 && Id_L < Sum_L) {
                    const GlobalPoint3D* left = lane_pool + si.lanes.left_off;
                    nLF = dev_load_front(c, left, Sum_L, Id_L, sh.u.road.LF);
                    nLR = dev_load_rear(c, left, Sum_L, Id_L, sh.u.road.LR);
                }
            } else synth_left = 1;
        }
        if (LaneChg == 2) {
            if (LaneNum_Cur < LaneSum) {
                int Id_Rt = loc.id[clampi(LaneNum_Cur, 0, DMPP_LANESUM - 1)], Sum_Rt = si.lanes.right_n;
                if (Id_Rt > 0 && Id_Rt < Sum_Rt) {
                    const GlobalPoint3D* right = lane_pool + si.lanes.right_off;
                    nRF = dev_load_front(c, right, Sum_Rt, Id_Rt, sh.u.road.RF);
                    nRR = dev_load_rear(c, right, Sum_Rt, Id_Rt, sh.u.road.RR);
                }
            } else synth_right = 1;
        }
        __syncthreads();
        if (synth_left) {          // Decision.cpp:629-631
            for (int i = tid; i < nF; i += kBlock) sh.u.road.LF[i] = offset_point(c, sh.u.road.F, nF, i, -1 * W);
            for (int i = tid; i < nR; i += kBlock) sh.u.road.LR[i] = offset_point(c, sh.u.road.R, nR, i, -1 * W);
            nLF = nF; nLR = nR;
        }
        if (synth_right) {         // Decision.cpp:667-669
            for (int i = tid; i < nF; i += kBlock) sh.u.road.RF[i] = offset_point(c, sh.u.road.F, nF, i, W);
            for (int i = tid; i < nR; i += kBlock) sh.u.road.RR[i] = offset_point(c, sh.u.road.R, nR, i, W);
            nRF = nF; nRR = nR;
        }
        __syncthreads();
        FRONT_MARK(1)
        // ---- AroundObstacle, Decision.cpp:759-881: six corridor queries over four waves ----
        const GlobalPoint2D* P[6] = { sh.u.road.F, sh.u.road.R, sh.u.road.LF, sh.u.road.LR, sh.u.road.RF, sh.u.road.RR };
        const int N[6] = { nF, nR, nLF, nLR, nRF, nRR };
        const double LO[6] = { -hv, -hv, -hv, -hv, -0.5 * W, -0.5 * W };
        const double HI[6] = { hv, hv, 0.5 * W, 0.5 * W, hv, hv };
        {   // two teams of two waves, three rounds: corridor t = 2 * round + team
            const int team = wave >> 1;
            for (int rd = 0; rd < 3; rd++) {
                const int t = 2 * rd + team;
                const SoResult r = team_search_obstacle<2>(c, P[t], N[t], team == 0 ? sh.u.road.s0 : sh.u.road.s1, obs, m, LO[t], HI[t], true, sh.part_d2, sh.part_bi);
                if ((tid & 127) == 0) { sh.around[t] = r; sh.n[t] = N[t]; }
            }
        }
        __syncthreads();
        if (tid < 6) store_path_obs(&po.around[tid], sh.around[tid], obs, sh.n[tid] != 0);
        FRONT_MARK(2)
        // ---- BehaviorDecision, no-lane-change map: Decision.cpp:920-1010 ----
        // (an empty front path leaves dis_lng = 0 from the memset at Decision.cpp:794)
        const double F_lng = (nF != 0) ? sh.around[0].dis_lng : 0.0;
        const double lim = (W - c.Vehicle_Width) / 0.6;
        int n_cand = 0;
        for (int i = 0; i < DMPP_MAX_SWEEP; i++) if ((double)i < lim) n_cand = i + 1;
        const int do_sweep = (LaneChg == 0) && (F_lng < 15) && (st.obsavoid_time + 1 > 2);
        if (do_sweep) {
            // the candidates of both sides are independent: evaluate all, pick the first accepted
            const int team = wave >> 1;                                        // two teams of two waves, one candidate each per round
            for (int t0 = 0; t0 < 2 * n_cand; t0 += 2) {
                const int t = t0 + team;
                const bool act = t < 2 * n_cand;
                const int side = act ? t / n_cand : 0, i = act ? t - side * n_cand : 0;
                const double off = (side == 0 ? -0.3 : 0.3) * (double)i;      // Decision.cpp:942,961
                if (act) for (int k = tid & 127; k < nF; k += 128) sh.u.road.tmp[team][k] = offset_point(c, sh.u.road.F, nF, k, off);
                __syncthreads();
                const SoResult r = team_search_obstacle<2>(c, sh.u.road.tmp[team], nF, team == 0 ? sh.u.road.s0 : sh.u.road.s1, obs, m, -hv, hv, act, sh.part_d2, sh.part_bi);
                if (act && (tid & 127) == 0) sh.sweep_lng[side * DMPP_MAX_SWEEP + i] = r.dis_lng;
            }
        }
        __syncthreads();
        FRONT_MARK(3)
        // ---- lane-change rule tree, part 1: the "remaining lane-change length" tests ----
        // Decision.cpp:1178-1187 (and :1212, 1316, 1344, 1459, 1477, 1506, 1523) walk the current lane from the ego point
        // while the next point's lanechg_attribute passes a predicate, summing segment lengths, and compare the sum with
        // 60 / 50 / 15 / 10 m.  As the compiler parses them the predicates are `attr == 1` (A), `attr & 1` (B) and
        // constant false.
        //   rem_flags bit0: A > 60, bit1: B > 10, bit2: B > 15, bit3: B > 50
        const bool run_tree = (LaneChg != 0) && c.lanechg_stage;
        if (run_tree) {
            double* seg = sh.u.road.seg;
            unsigned char* sa = sh.u.road.sa;
            const uint8_t* attr = attr_pool + si.lanes.cur_off;
            const int IdSum = si.lanes.cur_n;
            const int base0 = max(Id_Cur, 0);
            // The flags only ask whether the IN-ORDER sum over a whole run of segments exceeds a threshold (the reference sums the run,
            // then compares: Decision.cpp:1178-1190); the run's end comes from the attribute bytes (exact), and a sum taken in any other
            // order differs from the in-order one by at most n u sum.  So the block reduces in parallel and decides every flag whose
            // sum is farther than that bound from its threshold - all of them, in practice; should one be closer, thread 0 adds the
            // run in order as the reference does.  A NaN length inside a run makes its sum NaN and every comparison false.
            // (Thread 0 adding up to 256 lengths in order was 29 k of the kernel's 73 k cycles for a scene that may change lanes.)
            bool ambiguous = false;
            int flags = 0;
            {
                double sumA = 0, sumB = 0; bool liveA = true, liveB = true, nanA = false, nanB = false; int n_el = 0;
                for (int base = base0; (liveA || liveB) && base < IdSum - 1; base += kBlock) {      // (uniform over the block)
                    const int cnt = min(kBlock, IdSum - 1 - base);
                    const bool in = tid < cnt;
                    double d = 0; unsigned a = 0;
                    if (in) {
                        const int i = base + tid;
                        GlobalPoint2D p0 = { cur[i].x, cur[i].y }, p1 = { cur[i + 1].x, cur[i + 1].y };
                        d = CalcDistance(p0, p1); a = attr[i + 1];
                    }
                    const bool passA = in && a == 1, passB = in && (a & 1) != 0;
                    const unsigned long long fa = wave_ballot(!passA), fb = wave_ballot(!passB);
                    if (lane == 0) { sh.rem_w[0][wave] = fa ? __ffsll((long long)fa) - 1 : DMPP_WAVE; sh.rem_w[1][wave] = fb ? __ffsll((long long)fb) - 1 : DMPP_WAVE; }
                    __syncthreads();
                    int rA = 0, rB = 0;
                    { bool go = true; for (int w = 0; w < kBlock / DMPP_WAVE; w++) if (go) { rA += sh.rem_w[0][w]; go = sh.rem_w[0][w] == DMPP_WAVE; } }
                    { bool go = true; for (int w = 0; w < kBlock / DMPP_WAVE; w++) if (go) { rB += sh.rem_w[1][w]; go = sh.rem_w[1][w] == DMPP_WAVE; } }
                    if (!liveA) rA = 0;
                    if (!liveB) rB = 0;
                    const bool isn = d != d;
                    double pa = (tid < rA && !isn) ? d : 0.0, pb = (tid < rB && !isn) ? d : 0.0;
                    const unsigned long long na = wave_ballot(tid < rA && isn), nb = wave_ballot(tid < rB && isn);
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1) { pa += shfl_xor_f64(pa, sft); pb += shfl_xor_f64(pb, sft); }
                    __syncthreads();
                    if (lane == 0) { sh.rem_s[0][wave] = pa; sh.rem_s[1][wave] = pb; sh.rem_w[0][wave] = na ? 1 : 0; sh.rem_w[1][wave] = nb ? 1 : 0; }
                    __syncthreads();
                    for (int w = 0; w < kBlock / DMPP_WAVE; w++) {
                        sumA += sh.rem_s[0][w]; sumB += sh.rem_s[1][w];
                        nanA = nanA || sh.rem_w[0][w] != 0; nanB = nanB || sh.rem_w[1][w] != 0;
                    }
                    __syncthreads();
                    n_el += cnt;
                    if (rA < cnt || cnt < kBlock) liveA = false;          // the run (or the lane) ends in this block
                    if (rB < cnt || cnt < kBlock) liveB = false;
                }
                const double epsA = (double)n_el * 2.3e-16 * sumA + 1e-300, epsB = (double)n_el * 2.3e-16 * sumB + 1e-300;
                if (!nanA) { if (fabs(sumA - 60) <= epsA) ambiguous = true; else if (sumA > 60) flags |= 1; }
                if (!nanB) {
                    if (fabs(sumB - 10) <= epsB || fabs(sumB - 15) <= epsB || fabs(sumB - 50) <= epsB) ambiguous = true;
                    if (sumB > 10) flags |= 2;
                    if (sumB > 15) flags |= 4;
                    if (sumB > 50) flags |= 8;
                }
            }
            if (tid == 0) { sh.rem_go = ambiguous ? 1 : 0; sh.rem_flags = flags; }
            if (__builtin_expect(ambiguous, 0)) {
                // the in-order sums of both runs: segment lengths 256 at a time by the block, thread 0 adds them in order
                int base = base0;
                double sumA = 0, sumB = 0; bool liveA = true, liveB = true;
                for (;;) {
                    __syncthreads();
                    if (!sh.rem_go) break;
                    const int i = base + tid;
                    if (i < IdSum - 1) {
                        GlobalPoint2D a = { cur[i].x, cur[i].y }, b = { cur[i + 1].x, cur[i + 1].y };
                        seg[tid] = CalcDistance(a, b); sa[tid] = attr[i + 1];
                    }
                    __syncthreads();
                    if (tid == 0) {
                        const int cnt = min(kBlock, IdSum - 1 - base);
                        for (int k = 0; k < cnt && (liveA || liveB); k++) {
                            const unsigned a = sa[k]; const double d = seg[k];
                            if (liveA) { if (a == 1) sumA += d; else liveA = false; }
                            if (liveB) { if (a & 1) sumB += d; else liveB = false; }
                        }
                        if (cnt < kBlock) { liveA = false; liveB = false; }
                        sh.rem_go = (liveA || liveB) ? 1 : 0;
                        sh.rem_flags = (sumA > 60 ? 1 : 0) | (sumB > 10 ? 2 : 0) | (sumB > 15 ? 4 : 0) | (sumB > 50 ? 8 : 0);
                    }
                    base += kBlock;
                }
            }
        }
        FRONT_MARK(4)
        if (tid == 0) {
            // Nav_LaneChange, Decision.cpp:685-738 (+ CalcNaviLaneChgTimes :498-538)
            unsigned navi = 4, navi_times = 0;
            {
                const unsigned L = (unsigned)LaneNum_Cur;
                for (int i = 0; i < DMPP_LANESUM && si.out_lane_no[i] != 0; i++) if (L == si.out_lane_no[i]) { navi = 0; break; }
                const unsigned lane_min = si.out_lane_no[0];
                unsigned lane_max = 1;
                for (int i = 0; i < DMPP_LANESUM; i++) lane_max = max(lane_max, (unsigned)si.out_lane_no[i]);
                if (navi == 4) {
                    const int dir = (L < lane_min) ? 2 : (L > lane_max) ? 1 : 0;
                    navi = (unsigned)dir;
                    if (dir) {
                        int times = 5;
                        const int lc = (int)(L & 0xffu);
                        for (int i = 0; i < DMPP_LANESUM && si.out_lane_no[i] != 0; i++) {
                            const int t = (dir == 1) ? lc - (int)si.out_lane_no[i] : (int)si.out_lane_no[i] - lc;
                            times = min(times, t);
                        }
                        navi_times = (unsigned)(times & 0xff);
                    }
                }
            }
            po.navi_lanechg = (int)navi; po.navi_lanechg_times = (int)navi_times;
            Behavior_Dec Cur;
            Cur.behavior = st.z_behavior; Cur.light_status = st.z_light_status; Cur.target_lanenum = st.z_target_lanenum;
            Cur.lanechg_status = st.z_segment_lanechg_status; Cur.obsavoid_status = st.z_segment_obsavoid_status;
            Cur.behavior_to_dlg = st.z_behavior_to_dlg;
            int sweep_side = 0, sweep_index = -1;
            if (LaneChg == 0) {
                if (F_lng < 15) {
                    st.no_obsaviod_time = 0;
                    st.obsavoid_time = st.obsavoid_time + 1;
                    if (st.obsavoid_time > 2) {
                        int left_flag = 0;
                        for (int i = 0; i < n_cand; i++) if (sh.sweep_lng[i] > 25) {
                            Cur.behavior = 4; Cur.target_lanenum = LaneNum_Cur; Cur.light_status = 1;
                            Cur.obsavoid_status = 1; Cur.behavior_to_dlg = 11;
                            left_flag = 1; sweep_side = -1; sweep_index = i; break;
                        }
                        if (!left_flag) for (int i = 0; i < n_cand; i++) if (sh.sweep_lng[DMPP_MAX_SWEEP + i] > 25) {
                            Cur.behavior = 5; Cur.target_lanenum = LaneNum_Cur; Cur.light_status = 2;
                            Cur.obsavoid_status = 1; Cur.behavior_to_dlg = 12;
                            sweep_side = 1; sweep_index = i; break;
                        }
                    } else {
                        Cur.behavior = 1; Cur.target_lanenum = LaneNum_Cur; Cur.light_status = 0; Cur.behavior_to_dlg = 1;
                    }
                } else {
                    if (st.z_segment_obsavoid_status == 0) {
                        Cur.behavior = 1; Cur.target_lanenum = LaneNum_Cur; Cur.light_status = 0; Cur.behavior_to_dlg = 1;
                    } else {
                        st.no_obsaviod_time = st.no_obsaviod_time + 1;
                        if (st.no_obsaviod_time > 3) {
                            Cur.behavior = 1; Cur.target_lanenum = LaneNum_Cur; Cur.light_status = 0;
                            Cur.behavior_to_dlg = 1; Cur.obsavoid_status = 0;
                        }
                    }
                    Cur.behavior_to_dlg = 1;
                }
            } else {
                st.no_obsaviod_time = 0; st.obsavoid_time = 0;      // Decision.cpp:1014-1015
                if (run_tree)
                    lane_change_tree(si, LaneNum_Cur, LaneSum, LaneChg, navi, sh.rem_flags,
                                     F_lng, (nLF != 0) ? sh.around[2].dis_lng : 0.0, (nLR != 0) ? sh.around[3].dis_lng : 0.0,
                                     (nRF != 0) ? sh.around[4].dis_lng : 0.0, (nRR != 0) ? sh.around[5].dis_lng : 0.0, st, Cur);
            }
            st.z_velocity_expect = (Cur.behavior == 4 || Cur.behavior == 5) ? 5 : 10;   // SpeedDecision
            st.z_behavior = Cur.behavior; st.z_light_status = Cur.light_status; st.z_target_lanenum = Cur.target_lanenum;
            st.z_segment_lanechg_status = Cur.lanechg_status; st.z_segment_obsavoid_status = Cur.obsavoid_status;
            st.z_behavior_to_dlg = Cur.behavior_to_dlg; st.z_target_roadnum = loc.road_num;
            po.sweep_side = sweep_side; po.sweep_index = sweep_index;
            sh.n_ref = (Cur.behavior == 2) ? nLF : (Cur.behavior == 3) ? nRF : nF;      // RefPath
            sh.do_sweep = Cur.behavior;
        }
        __syncthreads();
        {
            const int beh = sh.do_sweep;
            const GlobalPoint2D* src = (beh == 2) ? sh.u.road.LF : (beh == 3) ? sh.u.road.RF : sh.u.road.F;
            for (int i = tid; i < sh.n_ref; i += kBlock) out_ref[i] = src[i];
        }
    } else if (pos == 1 || pos == 2) {
        // ---- PreStubDecision / StubDecision, Decision.cpp:323-486 ----
        const GlobalPoint3D* cur = lane_pool + si.lanes.cur_off;
        const GlobalPoint2D* inter = ref_pool + si.ref_off;
        int n = 0;
        if (pos == 1) {
            int Id_Cur = max(loc.id[clampi(loc.lane_num - 1, 0, DMPP_LANESUM - 1)], 0);
            int n1 = max(si.lanes.cur_n - Id_Cur, 0); n1 = min(n1, DMPP_MAX_REFPATH);
            int n2 = min(max(si.ref_n, 0), DMPP_MAX_REFPATH - n1);
            for (int k = tid; k < n1; k += kBlock) { sh.u.junc.ref[k].x = cur[Id_Cur + k].x; sh.u.junc.ref[k].y = cur[Id_Cur + k].y; }
            for (int k = tid; k < n2; k += kBlock) sh.u.junc.ref[n1 + k] = inter[k];
            n = n1 + n2;
        } else {
            int Id_Inter = max(loc.id[clampi(loc.last_lanenum - 1, 0, DMPP_LANESUM - 1)], 0);
            int n1 = max(si.ref_n - Id_Inter, 0); n1 = min(n1, DMPP_MAX_REFPATH);
            int n2 = min(min(60, si.lanes.cur_n), DMPP_MAX_REFPATH - n1); n2 = max(n2, 0);
            for (int k = tid; k < n1; k += kBlock) sh.u.junc.ref[k] = inter[Id_Inter + k];
            for (int k = tid; k < n2; k += kBlock) { sh.u.junc.ref[n1 + k].x = cur[k].x; sh.u.junc.ref[n1 + k].y = cur[k].y; }
            n = n1 + n2;
        }
        __syncthreads();
        {
            const SoResult r = team_search_obstacle<4>(c, sh.u.junc.ref, n, sh.u.junc.s0, obs, m, -hv, hv, true, sh.part_d2, sh.part_bi);
            if (tid == 0) sh.around[0] = r;
        }
        __syncthreads();
        if (tid < 6) {
            if (tid == 0) store_path_obs(&po.around[0], sh.around[0], obs, true);
            else { SoResult z; z.flag = 0; z.path_id = 0; z.ob_index = -1; z.dis_lat = 0; z.dis_lng = 0; store_path_obs(&po.around[tid], z, obs, false); }
        }
        if (tid == 0) {
            const double d = sh.around[0].dis_lng;
            if (d < 13) { double v = d - 3; st.z_velocity_expect = v > 0 ? v : 0; st.z_behavior_to_dlg = 13; }
            else { st.z_velocity_expect = 10; st.z_behavior_to_dlg = 1; }
            st.z_light_status = (si.stub_attribute == 3) ? 1 : si.stub_attribute;
            st.z_behavior = 1; st.z_target_roadnum = loc.road_num; st.z_target_lanenum = loc.lane_num;
            po.sweep_side = 0; po.sweep_index = -1; po.navi_lanechg = 0; po.navi_lanechg_times = 0;
            sh.n_ref = n;
        }
        for (int i = tid; i < n; i += kBlock) out_ref[i] = sh.u.junc.ref[i];
        __syncthreads();
    } else {
        if (tid < 6) { SoResult z; z.flag = 0; z.path_id = 0; z.ob_index = -1; z.dis_lat = 0; z.dis_lng = 0; store_path_obs(&po.around[tid], z, obs, false); }
        if (tid == 0) { sh.n_ref = 0; po.sweep_side = 0; po.sweep_index = -1; po.navi_lanechg = 0; po.navi_lanechg_times = 0; }
        __syncthreads();
    }
    FRONT_MARK(5)
    if (tid == 0) {       // Decision.cpp:187-201
        DecisionOutPod d;
        d.velocity_expect = st.z_velocity_expect; d.behavior = st.z_behavior; d.target_roadnum = st.z_target_roadnum;
        d.target_lanenum = st.z_target_lanenum; d.light = st.z_light_status; d.behavior_to_dlg = st.z_behavior_to_dlg;
        d.refpath_n = sh.n_ref;
        po.dec = d;
        st.d_his_behavior = st.z_behavior; st.d_his_light_status = st.z_light_status; st.d_his_target_lanenum = st.z_target_lanenum;
    }
}

// ---------------------------------------------------------------------------------------
struct PlanShared {
    GlobalPoint2D last[DMPP_PATH_POINTS];
    GlobalPoint2D road[DMPP_PATH_POINTS];
    double dist[kBlock];           // also the 256-wide segment buffer of block_aim_walk
    double seg[DMPP_PATH_POINTS];
    double s[DMPP_PATH_POINTS + 8];
    double part_d2[kBlock]; int part_bi[kBlock];      // team_search_obstacle
    AimPoint aim_far;
    SoResult so;
    double sh_sum;
    int sh_found, afresh, near_id, na;
};

__global__ void __launch_bounds__(kBlock)
k_planning(PlannerConfig c, int n_scenes, const SceneIn* __restrict__ in, const GlobalPoint3D* __restrict__ lane_pool,
           const GlobalPoint2D* __restrict__ ref_pool, const GlobalPoint2D* __restrict__ dec_ref,
           const ObPoint* __restrict__ obs_now, SceneState* __restrict__ state, PlanOut* __restrict__ plan)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    PlanShared& sh = *reinterpret_cast<PlanShared*>(smem_raw);
    const int scene = blockIdx.x;
    if (scene >= n_scenes) return;
    __builtin_amdgcn_s_setprio(DMPP_FRONT_PRIO);             // front chain: ahead of the searching waves on the same SIMD
#ifdef DMPP_DEBUG_SEARCH
    const long long dbg_t0 = clock64();
    double* dbg_stamp = const_cast<double*>(reinterpret_cast<const double*>(dec_ref + (size_t)blockIdx.x * DMPP_MAX_REFPATH + 490));
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const SceneIn& si = in[scene];
    SceneState& st = state[scene];
    PlanOut& po = plan[scene];
    const LocationOut& loc = si.loc;
    const int pos = loc.pos;
    const int m = si.obs_n;
    const ObPoint* obs = obs_now + si.obs_off;

    // DecisionOut: published by k_decision into PlanOut.dec, or the caller's (decision stage off)
    DecisionOutPod dec;
    const GlobalPoint2D* refpath;
    if (c.decision_stage) { dec = po.dec; refpath = dec_ref + (size_t)scene * DMPP_MAX_REFPATH; }
    else {
        dec = si.dec;
        if (dec.refpath_n > si.ref_n) dec.refpath_n = si.ref_n;
        if (dec.refpath_n > DMPP_MAX_REFPATH) dec.refpath_n = DMPP_MAX_REFPATH;
        refpath = ref_pool + si.ref_off;
    }
    const GlobalPoint3D ego = loc.globalpoint;

    // ---- Calculate_aim_dis, Planning.cpp:242-290 (FLOAT members) ----
    float faraim = 0, nearaim = 0;
    switch (pos) {
    case 0:
        faraim = (float)((loc.velocity / 3.6) * 5 + 4);
        if (faraim > c.ROAD_FARAIM_MAX) faraim = (float)c.ROAD_FARAIM_MAX;
        else if (faraim < c.ROAD_FARAIM_MIN) faraim = (float)c.ROAD_FARAIM_MIN;
        nearaim = faraim; break;
    case 1: faraim = (float)c.PRE_INTER_FARAIM; nearaim = faraim; break;
    case 2: faraim = (float)c.INTER_FARAIM; nearaim = faraim; break;
    default: break;
    }

    // ---- SearchAimPoint, Planning.cpp:303-583 ----
    if (tid == 0) sh.aim_far = st.aimpoint_far;
    AimPoint aim_near_new = st.aimpoint_near;
    bool near_follows_far = false;
    {
        const int LaneNum_Cur = loc.lane_num, LaneSum = si.lanes.lane_sum;
        const int curpoint_id = loc.id[clampi(LaneNum_Cur - 1, 0, DMPP_LANESUM - 1)];
        const int curpoint_sum = si.lanes.cur_n;
        int leftpoint_id = 0, leftpoint_sum = 0, rightpoint_id = 0, rightpoint_sum = 0;
        if (LaneNum_Cur > 1) { leftpoint_id = loc.id[clampi(LaneNum_Cur - 2, 0, DMPP_LANESUM - 1)]; leftpoint_sum = si.lanes.left_n; }
        if (LaneNum_Cur < LaneSum) { rightpoint_id = loc.id[clampi(LaneNum_Cur, 0, DMPP_LANESUM - 1)]; rightpoint_sum = si.lanes.right_n; }
        const bool plan_div = !(fabs(faraim - nearaim) < 1);
        const GlobalPoint3D* cur = lane_pool + si.lanes.cur_off;
        const GlobalPoint3D* left = lane_pool + si.lanes.left_off;
        const GlobalPoint3D* right = lane_pool + si.lanes.right_off;
        __syncthreads();
        if (pos == 0) {
            if (dec.target_lanenum == loc.lane_num) {
                if (dec.behavior == 1 && !plan_div) {
                    int iend = curpoint_sum - 1;
                    int f = block_aim_walk(&cur[0].x, 3, curpoint_id, iend, (double)faraim, sh.dist, &sh.sh_found, &sh.sh_sum);
                    if (tid == 0) {
                        if (f >= 0) { sh.aim_far.Aim_point = cur[f]; sh.aim_far.Aim_id = f; }
                        else if (max(curpoint_id, 0) < iend) { sh.aim_far.Aim_point = cur[curpoint_sum - 1]; sh.aim_far.Aim_id = curpoint_sum - 1; }
                    }
                    near_follows_far = true;                               // Planning.cpp:434
                }
            } else if (dec.behavior == 2) {
                if (!plan_div) {
                    int iend = leftpoint_sum - 1;
                    int f = block_aim_walk(&left[0].x, 3, leftpoint_id, iend, (double)faraim, sh.dist, &sh.sh_found, &sh.sh_sum);
                    if (tid == 0) {
                        if (f >= 0) { sh.aim_far.Aim_point = left[f]; sh.aim_far.Aim_id = f; }
                        else if (max(leftpoint_id, 0) < iend) {             // quirk Planning.cpp:464-467
                            int q = clampi(leftpoint_sum - 2, 0, curpoint_sum > 0 ? curpoint_sum - 1 : 0);
                            sh.aim_far.Aim_point = cur[q]; sh.aim_far.Aim_id = leftpoint_sum - 1;
                        }
                    }
                }
            } else if (dec.behavior == 3) {
                if (!plan_div) {
                    int iend = min(leftpoint_sum - 1, rightpoint_sum - 1);   // quirk Planning.cpp:478 + fence
                    int i0 = rightpoint_id;
                    int f = (i0 >= 0) ? block_aim_walk(&right[0].x, 3, i0, iend, (double)faraim, sh.dist, &sh.sh_found, &sh.sh_sum) : -1;
                    if (tid == 0) {
                        if (f >= 0) { sh.aim_far.Aim_point = right[f]; sh.aim_far.Aim_id = f; }
                        else if (i0 >= 0 && i0 < iend) { sh.aim_far.Aim_point = right[rightpoint_sum - 1]; sh.aim_far.Aim_id = rightpoint_sum - 1; }
                    }
                }
            }
        } else if (pos == 1 || pos == 2) {
            const int n = dec.refpath_n;
            if (n >= 3) {
                int f = block_aim_walk(&refpath[0].x, 2, 0, n - 1, (double)faraim, sh.dist, &sh.sh_found, &sh.sh_sum);
                if (tid == 0) {
                    if (f >= 0) {
                        sh.aim_far.Aim_point.x = refpath[f].x; sh.aim_far.Aim_point.y = refpath[f].y;
                        if (f < n - 4) sh.aim_far.Aim_point.dir = GetRoadAngle(c, refpath[f], refpath[f + 2]);
                        else sh.aim_far.Aim_point.dir = GetRoadAngle(c, refpath[f - 2 < 0 ? 0 : f - 2], refpath[f]);
                        sh.aim_far.Aim_id = f;
                    } else {
                        sh.aim_far.Aim_point.x = refpath[n - 1].x; sh.aim_far.Aim_point.y = refpath[n - 1].y;
                        sh.aim_far.Aim_point.dir = GetRoadAngle(c, refpath[n - 3], refpath[n - 1]);
                        sh.aim_far.Aim_id = n - 1;
                    }
                }
            }
            near_follows_far = true;                                       // Planning.cpp:539,577
        }
    }
    __syncthreads();
    const AimPoint aim_far = sh.aim_far;
    if (near_follows_far) aim_near_new = aim_far;

    FRONT_MARK(0)
    // ---- first tick: InitialPlanning, Planning.cpp:124-128,596-611 ----
    const int count = st.count;
    if (count == 0) {
        Bezier bz = bezier_setup(c, ego, aim_far.Aim_point);
        if (tid < DMPP_PATH_POINTS) sh.last[tid] = bezier_point(bz, tid, DMPP_PATH_POINTS);
    } else {
        if (tid < DMPP_PATH_POINTS) sh.last[tid] = st.last_Bpoints[tid];
    }
    __syncthreads();

    FRONT_MARK(1)
    // ---- GetVhclLocalState, Planning.cpp:623-676 ----
    if (tid < DMPP_PATH_POINTS) {
        double dx = ego.x - sh.last[tid].x, dy = ego.y - sh.last[tid].y;
        sh.dist[tid] = sqrt(dx * dx + dy * dy);
        if (tid < DMPP_PATH_POINTS - 1) {
            double sx = sh.last[tid + 1].x - sh.last[tid].x, sy = sh.last[tid + 1].y - sh.last[tid].y;
            sh.seg[tid] = sqrt(sx * sx + sy * sy);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // first minimum of the 200 distances below 9999 (Planning.cpp:640-650: strict <, so ties keep the lower index; a NaN is
        // never smaller): every lane over its four indices in order, then a (value, index) reduction over the wave
        double mind = 9999; int mid = -1;
        for (int k = 0; k < 4; k++) { const int i = lane + 64 * k; if (i < DMPP_PATH_POINTS && sh.dist[i] < mind) { mind = sh.dist[i]; mid = i; } }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const double o_d = shfl_xor_f64(mind, sft); const int o_i = __shfl_xor(mid, sft, 64);
            if (o_i >= 0 && (mid < 0 || o_d < mind || (o_d == mind && o_i < mid))) { mind = o_d; mid = o_i; }
        }
        if (mid < 0) mid = clampi(st.path_near_id, 0, DMPP_PATH_POINTS - 1);      // nothing within 9999 m: the member keeps its value
        const int fid = mid + 8;
        // remaining length: seg[fid] + ... + seg[198] in index order (Planning.cpp:668-671), one lane, eight reads per block
      if (lane == 0) {
        const double remain = serial_sum(sh.seg, min(fid, DMPP_PATH_POINTS - 1), DMPP_PATH_POINTS - 1, 0.0);
        const int index = (mid == 199) ? mid - 1 : mid;
        const GlobalPoint2D pt = sh.last[index], pt_next = sh.last[index + 1], vp = { ego.x, ego.y };
        const double lat = GetLatDis(c, vp, pt, pt_next);
        const double pt_dir = GetRoadAngle(c, pt, pt_next);
        const double dir_err = GetAngleErr(pt_dir, ego.dir);
        // ---- UpdatePlanJudge, Planning.cpp:797-832 ----
        int cause = 0;
        int afresh = d_UpdatePlanJudge(c, st.his_behavior, dec.behavior, pos, lat, dir_err, remain, cause);
        if (c.force_replan && !afresh) { afresh = 1; cause = 5; }
        st.path_lat_dis = lat; st.path_dir_err = dir_err; st.remain_dis = remain;
        st.path_near_id = mid; st.path_front_near_id = fid;
        st.afresh_planning = afresh; st.afresh_cause = cause;
        st.faraim_dis = faraim; st.nearaim_dis = nearaim;
        st.aimpoint_far = aim_far; st.aimpoint_near = aim_near_new;
        sh.afresh = afresh; sh.near_id = mid;
        int na = aim_far.Aim_id; if (na > 200) na = 200; if (na > dec.refpath_n) na = dec.refpath_n; if (na < 0) na = 0;
        sh.na = na;
        // CalculateRadius, Planning.cpp:1000-1019: runs on the OLD path (called at :199, path saved at :217)
        po.result.radius = d_CalculateRadius(sh.last, mid, fid);
      }
    }
    __syncthreads();

    FRONT_MARK(2)
    // ---- PathPlanning, Planning.cpp:845-877 (or reuse of the last path, :144-145) ----
    const int afresh = sh.afresh;
    if (afresh) {
        if (pos == 0) {
            Bezier bz = bezier_setup(c, ego, aim_far.Aim_point);
            if (tid < DMPP_PATH_POINTS) sh.road[tid] = bezier_point(bz, tid, DMPP_PATH_POINTS);
        } else if (pos == 1 || pos == 2) {
            const int na = sh.na;
            if (wave == 0) wave_cumlen(refpath, na, sh.s, lane);
            __syncthreads();
            if (tid < DMPP_PATH_POINTS) sh.road[tid] = mean_point(c, refpath, sh.s, na, tid, DMPP_PATH_POINTS);
        } else {
            if (tid < DMPP_PATH_POINTS) { sh.road[tid].x = 0; sh.road[tid].y = 0; }
        }
    } else {
        if (tid < DMPP_PATH_POINTS) sh.road[tid] = sh.last[tid];
    }
    __syncthreads();

    FRONT_MARK(3)
    // ---- SearchObstacle on the remaining path, Planning.cpp:153-168 ----
    const int near_id = clampi(sh.near_id, 0, 200);
    {   // the whole block on the one polyline: four shares of the points
        const SoResult r = team_search_obstacle<4>(c, sh.road + near_id, DMPP_PATH_POINTS - near_id, sh.s, obs, m,
                                                   (double)(float)(-1.1), (double)(float)(1.1), true, sh.part_d2, sh.part_bi);
        if (tid == 0) sh.so = r;
    }
    __syncthreads();

    FRONT_MARK(4)
    // ---- SpeedPlanning + publication, Planning.cpp:171-223 ----
    if (tid == 0) {
        const SoResult r = sh.so;
        double brakespeed = st.brakespeed, des_acc = st.des_acc; int acc_flag = st.acc_flag;
        const double lon = r.dis_lng;
        d_SpeedPlanning(pos, r.flag, lon, faraim, dec.velocity_expect, brakespeed, acc_flag, des_acc);
        st.brakespeed = brakespeed; st.des_acc = des_acc; st.acc_flag = acc_flag;
        po.show.afresh_cause = st.afresh_cause; po.show.near_ob_dist = lon; po.show.planspeed = brakespeed;
        po.show.planacc = des_acc; po.show.trafficlight = dec.light;
        po.result.cnt = count % 100; po.result.APA = 0; po.result.brakedis = lon; po.result.brake_speed = 0;
        po.result.desaccVd = acc_flag; po.result.desacc = des_acc; po.result.desspd = brakespeed; po.result.desstr = 0;
        po.result.desstrVd = 0; po.result.light = dec.light; po.result.road_type = 0; po.result.sstop = 1; po.result._pad = 0;
        po.ob_dis_lat = r.dis_lat; po.ob_dis_lng = lon; po.ob_flag = r.flag; po.ob_pathid = r.path_id;
        if (r.flag) po.ob = obs[r.ob_index]; else { po.ob.x = 0; po.ob.y = 0; po.ob.type = 0; po.ob.radius = 0; }
        st.his_behavior = dec.behavior;
        int cn = (count + 1) & 0xFF; if (cn % 100 == 1) cn = 1;
        st.count = cn;
        st.tick = st.tick + 1;
        if (!c.decision_stage) {
            po.dec = dec; po.sweep_side = 0; po.sweep_index = -1; po.navi_lanechg = 0; po.navi_lanechg_times = 0;
        }
    }
    if (!c.decision_stage && tid < 6) {
        SoResult z; z.flag = 0; z.path_id = 0; z.ob_index = -1; z.dis_lat = 0; z.dis_lng = 0;
        store_path_obs(&po.around[tid], z, obs, false);
    }
    if (tid < DMPP_PATH_POINTS) {
        const GlobalPoint2D p = sh.road[tid];
        po.road_points[tid] = p;
        st.last_Bpoints[tid] = p;                                           // Planning.cpp:217
        if ((tid & 1) == 0) {
            const int k = tid >> 1;
            po.show.path_points[k] = p;                                     // Planning.cpp:180-183
            GlobalPoint2D g;                                                // GlobalToWGS84, Planning.cpp:205-212
            g.x = c.wgs_lat0 + p.y * c.wgs_deg_per_m_lat;
            g.y = c.wgs_lng0 + p.x * c.wgs_deg_per_m_lng;
            po.result.pnts[k] = g;
        }
    }
    FRONT_MARK(5)
}

}  // namespace dmpp
