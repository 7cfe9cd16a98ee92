/*
 * dmpp_types.h — plain-C data types of the planning hot path.
 *
 * The reference defines none of these: every struct below lives in its unshipped
 * `Share.h` (included at Planning.h:2 / Decision.h:2).  Field sets are exactly the
 * fields the reference touches; scalar widths and layouts are
 * chosen here and are part of this library's ABI.  Coordinates are `double`
 * (the only width consistent with every use in Planning.cpp / Decision.cpp).
 *
 * This header is pure C99, shared by the C-ABI (`dmpp_planner.h`), the C++ host
 * classes, the HIP kernels and — as data layout only — the CPU oracle.
 */
#ifndef DMPP_TYPES_H
#define DMPP_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fixed sizes taken from the reference ------------------------------------ */
#define DMPP_PATH_POINTS   200 /* GlobalPoint2D road_points[200]    Planning.cpp:115, Planning.h:33 */
#define DMPP_OUT_POINTS    100 /* every 2nd point is published      Planning.cpp:180-183,205-212    */
#define DMPP_FRONT_POINTS  120 /* corridor ahead  "120 points ~60m" Decision.cpp:581                */
#define DMPP_REAR_POINTS    40 /* corridor behind "40 points  ~20m" Decision.cpp:590                */
#define DMPP_STUB_NEXT_PTS  60 /* exit-lane points added in a junction, Decision.cpp:446            */
#define DMPP_LANESUM         8 /* LocationOut.id[LANESUM]; value is build-chosen       */
#define DMPP_MAX_SWEEP       8 /* lateral sweep candidates per side bound (BYTE i loop, Decision.cpp:940) */
#define DMPP_MAX_REFPATH   512 /* cap on DecisionOut.refpath length carried on device               */

/* ---- geometry (Share.h types) ------------------------------------- */
typedef struct GlobalPoint2D { double x, y; } GlobalPoint2D;             /* 16 B */
typedef struct GlobalPoint3D { double x, y, dir; } GlobalPoint3D;        /* 24 B; dir: degrees CCW from east */
typedef struct GPSPoint2D    { double lat, lng; } GPSPoint2D;

/* AimPoint{Aim_point, Aim_id}: Planning.cpp:418-421 */
typedef struct AimPoint { GlobalPoint3D Aim_point; int32_t Aim_id; int32_t _pad; } AimPoint; /* 32 B */

/* ObPoint is opaque in the reference (no field is ever read).  24 B as SURVEY §8
 * prices it: position + type, the pad word carries the footprint radius the grid
 * engine rasterises (metres, f32). */
typedef struct ObPoint { double x, y; int32_t type; float radius; } ObPoint;        /* 24 B */
/* constant-velocity model for dynamic obstacles (row G4; Decision.h:14-15 declares
 * dynamic-obstacle lists the reference never fills). m/s in world axes. */
typedef struct ObMotion { double vx, vy; } ObMotion;

/* map element: planning_MapData[road][lane][id] -> {global_point, lane_sum,
 * lanechg_attribute, lane_width(cm)}  (Decision.cpp:563,566,578; Planning.cpp:413-420).
 * On this ABI a lane is a flat array of GlobalPoint3D; the per-lane scalars travel in
 * LaneView. */
typedef struct LaneView {
    int32_t cur_off,   cur_n;   /* current lane  : points [cur_off, cur_off+cur_n) of the lane-point pool */
    int32_t left_off,  left_n;  /* left  lane (lane_num-2); n = 0 when absent   Planning.cpp:359-368 */
    int32_t right_off, right_n; /* right lane (lane_num);   n = 0 when absent   Planning.cpp:371-380 */
    int32_t lane_sum;           /* [0].lane_sum                                 Planning.cpp:331      */
    int32_t lanechg_attribute;  /* 0 none,1 left,2 right,3 both at ego id       Decision.cpp:566      */
    double  lane_width;         /* metres (= lane_width/100.0)                  Decision.cpp:578      */
} LaneView;                                                                  /* 40 B */

/* LocationOut: fields read on the path. */
typedef struct LocationOut {
    GlobalPoint3D globalpoint;        /* ego pose                                   Planning.cpp:599,637 */
    double  velocity;                 /* km/h                                       Planning.cpp:258     */
    int32_t pos;                      /* 0 road, 1 pre-junction, 2 junction         Planning.cpp:247     */
    int32_t road_num, lane_num;       /* 1-based                                    Planning.cpp:329-330 */
    int32_t last_roadnum, next_roadnum, last_lanenum, next_lanenum;
    int32_t path_num;
    int32_t id[DMPP_LANESUM];         /* ego point id on each lane                  Planning.cpp:338     */
} LocationOut;                                                              /* 96 B */

/* DecisionOut (Decision.cpp:187-196) minus the std::vector and the two period fields: refpath is flattened to
 * (offset,len) into a point pool at the C-ABI.  The C++ host surface (host/dmpp_share.hpp) declares
 * `struct DecisionOut` - the type the reference's methods take, Planning.h:57-75 - on top of this POD. */
typedef struct DecisionOutPod {
    double  velocity_expect;
    int32_t behavior;         /* 1 keep,2 left chg,3 right chg,4 left avoid,5 right avoid,6 grid search (Decision.h:36) */
    int32_t target_roadnum, target_lanenum;
    int32_t light;            /* 0 none,1 left,2 right,3 hazard (Decision.h:39) */
    int32_t behavior_to_dlg;
    int32_t refpath_n;        /* number of valid refpath points (<= DMPP_MAX_REFPATH) */
} DecisionOutPod;                                                           /* 32 B */

/* Obs_To_Veh / Path_Obs: Decision.cpp:847-850 */
typedef struct Obs_To_Veh { double dis_lat, dis_lng; } Obs_To_Veh;
typedef struct Path_Obs {
    Obs_To_Veh Ob_Pose;
    ObPoint    Ob_Attr;
    int32_t    Obs_flag;
    int32_t    Ob_Pathid;
} Path_Obs;                                                                 /* 48 B */

/* Behavior_Dec: Decision.cpp:286-291 */
typedef struct Behavior_Dec {
    int32_t behavior, light_status, target_lanenum;
    int32_t lanechg_status, obsavoid_status, behavior_to_dlg;
} Behavior_Dec;

/* PlanningOut (Planning.cpp:189-212) */
typedef struct PlanningOut {
    double  brakedis, brake_speed, desacc, desspd, desstr, radius;
    int32_t cnt, APA, desaccVd, desstrVd, light, road_type, sstop, _pad;
    GlobalPoint2D pnts[DMPP_OUT_POINTS];      /* .x = lat, .y = lng */
} PlanningOut;                                                              /* 1680 B */

/* PlanningStatus (Planning.cpp:174-183) */
typedef struct PlanningStatus {
    double  near_ob_dist, planspeed, planacc;
    int32_t afresh_cause, trafficlight;
    GlobalPoint2D path_points[DMPP_OUT_POINTS];
} PlanningStatus;                                                           /* 1632 B */

/* ---- per-scene cross-tick state ------------------------------------------------
 * Everything the reference keeps in statics / members between ticks
 * (Planning.cpp:6,52-53,216-223; Planning.h:20-23,42-52; Decision.cpp:199-201,915-917;
 * Decision.h:27-44) made explicit so scenes are independent and batchable. */
typedef struct SceneState {
    GlobalPoint2D last_Bpoints[DMPP_PATH_POINTS];   /* Planning.h:33               3200 B */
    AimPoint aimpoint_far, aimpoint_near;           /* Planning.h:22-23 (stale across ticks by design) */
    double  path_lat_dis, remain_dis, path_dir_err; /* Planning.h:42-46 */
    double  brakespeed, des_acc;                    /* Planning.h:50-52 */
    float   faraim_dis, nearaim_dis;                /* FLOAT members, Planning.h:20-21 */
    int32_t path_near_id, path_front_near_id;       /* Planning.h:47-48 */
    int32_t his_behavior;                           /* Planning.h:49, ctor sets 1 (Planning.cpp:10) */
    int32_t afresh_planning, afresh_cause;          /* Planning.h:43-44 */
    int32_t acc_flag;                               /* Planning.h:51 */
    int32_t count;                                  /* BYTE count, Planning.cpp:51,219-223 */
    /* Decision side */
    int32_t z_behavior, z_light_status, z_target_lanenum, z_target_roadnum; /* Decision.h:36-39 */
    int32_t z_behavior_to_dlg;                                              /* Decision.h:27    */
    int32_t z_segment_lanechg_status, z_segment_obsavoid_status;            /* Decision.h:28-29 */
    int32_t d_his_behavior, d_his_light_status, d_his_target_lanenum;       /* Decision.h:41-44 */
    uint32_t obsavoid_time, no_obsaviod_time, frontobs_time;                /* statics, Decision.cpp:915-917 */
    int32_t tick;                                   /* ticks elapsed; drives the dynamic-obstacle model (G4) */
    int32_t _pad;
    double  z_velocity_expect;                      /* Decision.h:47 */
    double  leftlight_time, rightlight_time;        /* Decision.h:31-32: turn-signal timers, ms */
} SceneState;

/* ---- per-scene per-tick input ---------------------------------------------------- */
typedef struct SceneIn {
    LocationOut loc;          /* app->GetLocationOut()   Planning.cpp:101 */
    DecisionOutPod dec;       /* app->GetDesicionOut()   Planning.cpp:96  — used as is when the decision stage is off */
    LaneView    lanes;        /* slice of the lane-point pool standing in for planning_MapData */
    int32_t ref_off, ref_n;   /* DecisionOut.refpath / junction polyline: slice of the refpath pool */
    int32_t obs_off, obs_n;   /* app->GetObj() snapshot: slice of the obstacle pool (Planning.cpp:111) */
    int32_t stub_attribute;   /* z_RoadNavi[path_num].stub_attribute, Decision.cpp:385 */
    int32_t _pad;
    /* lane-change rule tree inputs (Decision.cpp:685-738, 1011-1772) */
    uint16_t out_lane_no[DMPP_LANESUM]; /* z_RoadNavi[path_num].out_lane_no[]: exit lanes of this road, 0-terminated (Decision.cpp:696) */
    double   period_last;     /* z_period_last: milliseconds since the previous decision tick (Decision.cpp:137) */
    /* grid engine (rows G1-G4; absent from the reference) */
    GlobalPoint2D grid_origin;/* world coordinates of the corner of cell (0,0) */
    GlobalPoint2D goal;       /* goal position, world */
} SceneIn;

/* ---- map store ------------------------------------------------------
 * planning_MapData[road][lane][id] / decision_MapData (Planning.cpp:331-356; Decision.cpp:562-578) and
 * planning_InterMapData[last_road][next_road][last_lane][next_lane][id] (Planning.cpp:342; Decision.cpp:348)
 * flattened: one point pool shared by every scene, structure-of-arrays for the per-point attributes, a lane
 * table indexed through the roads' first-lane offsets, and a junction table.  Roads and lanes are 1-based as
 * in the reference; lane l of road r is lanes[road_first_lane[r-1] + l-1]. */
typedef struct MapLane {
    int32_t point_off, n_points;   /* points [point_off, point_off + n_points) of the point pool        */
    int32_t lane_sum;              /* [road][lane][0].lane_sum                       Planning.cpp:331    */
    int32_t _pad;
} MapLane;
typedef struct MapJunction {
    int32_t last_road, next_road, last_lane, next_lane;   /* the four indices of planning_InterMapData, 1-based */
    int32_t point_off, n_points;   /* slice of the junction point pool                                   */
} MapJunction;
typedef struct MapDesc {
    int32_t n_roads, n_lanes, n_points, n_junctions, n_jpoints, _pad;
    const int32_t*       road_first_lane;     /* n_roads + 1 entries                                        */
    const MapLane*       lanes;               /* n_lanes                                                    */
    const GlobalPoint3D* points;              /* n_points: global_point                                     */
    const uint8_t*       lanechg_attribute;   /* n_points: 0 none, 1 left, 2 right, 3 both  Decision.cpp:566 */
    const uint16_t*      lane_width_cm;       /* n_points: centimetres                      Decision.cpp:578 */
    const MapJunction*   junctions;           /* n_junctions                                                */
    const GlobalPoint2D* jpoints;             /* n_jpoints: junction polylines                              */
} MapDesc;

/* ---- grid-engine result (rows G2,G3) ---------------------------------------------- */
#define DMPP_G_FOUND      0
#define DMPP_G_NO_PATH    1   /* open set exhausted */
#define DMPP_G_LIMIT      2   /* max_expansions reached */
#define DMPP_G_OVERFLOW   3   /* more than bucket_cap live open-list entries */
#define DMPP_G_GOAL_BLOCKED 4 /* goal cell occupied: no search run */
#define DMPP_G_PATH_TRUNC 5   /* path longer than max_path cells */
#define DMPP_G_INTERNAL   6   /* a loop bound of the device search was hit: never expected, reported instead of hanging */
#define DMPP_G_COST_RANGE 7   /* a successor's f = g + h reached DMPP_F_LIMIT: only `status` is defined (as for OVERFLOW) */
#define DMPP_G_STATUS_COUNT 8

#define DMPP_JPS_BATCH 4       /* entries of the minimal f taken per step of the jump-point search */
#define DMPP_DIAG_JUMP 8       /* cells a diagonal jump of the jump-point search looks ahead before it settles for a plain node */
#define DMPP_OPEN_CAP 512      /* open-list slots the device keeps in LDS (496 is the most any generated scene needs); beyond them the entries
                                  that pop last live in a spill area in HBM, up to bucket_cap: a storage detail, not a limit of the specification */

/* f = g + h of an open-list entry must stay below this (the device keeps f/2 in 16 bits, 0xFFFF = dead slot): ~13,000
 * straight cells of detour.  A push that would reach it ends the search with DMPP_G_COST_RANGE, checked - per push, in
 * push order - before the open-list capacity. */
#define DMPP_F_LIMIT 131070

#define DMPP_MAX_LATTICE 17   /* n_lattice Bezier candidates + 1 grid-path candidate */

typedef struct GridOut {
    uint64_t order_digest;    /* sum over expansions of mix64(seq<<32 | cell)   */
    int32_t  status;
    int32_t  n_expanded;      /* closed cells, in expansion order                      */
    int32_t  n_pushed;        /* open-set insertions                                   */
    int32_t  n_rounds;        /* distinct f levels visited                             */
    int32_t  path_len;        /* cells on the chosen path, start..goal inclusive       */
    int32_t  path_cost;       /* integer g at the goal (10 straight / 14 diagonal)     */
    int32_t  start_cell, goal_cell;
    int32_t  best_candidate;  /* argmin of cand_cost, ties to the lowest index         */
    int32_t  n_candidates;
    double   cand_cost[DMPP_MAX_LATTICE];
    double   cand_col[DMPP_MAX_LATTICE], cand_curv[DMPP_MAX_LATTICE], cand_prog[DMPP_MAX_LATTICE];
    GlobalPoint2D best_path[DMPP_PATH_POINTS];
} GridOut;

/* ---- per-scene per-tick output ------------------------------------------------------ */
typedef struct PlanOut {
    PlanningOut    result;      /* what SetUdpSendCtrl receives, Planning.cpp:214 */
    PlanningStatus show;        /* what SetPlanningStatus receives, Planning.cpp:186 */
    GlobalPoint2D  road_points[DMPP_PATH_POINTS];  /* the 200-point path of this tick */
    Path_Obs       around[6];   /* F,R,LF,LR,RF,RR  (Decision.cpp:847-880) */
    DecisionOutPod dec;         /* decision published this tick (Decision.cpp:187-196) */
    double         ob_dis_lat, ob_dis_lng;   /* SearchObstacle outputs of Planning.cpp:168 */
    ObPoint        ob;
    int32_t        ob_flag, ob_pathid;
    int32_t        sweep_side, sweep_index;  /* which candidate of Decision.cpp:940-973 was accepted (-1 none) */
    int32_t        navi_lanechg;             /* Nav_LaneChange result, Decision.cpp:685-738: 0 none, 1 left, 2 right */
    int32_t        navi_lanechg_times;       /* CalcNaviLaneChgTimes, Decision.cpp:498-538 */
} PlanOut;

/* ---- every macro the reference uses but never defines ------------------ */
typedef struct PlannerConfig {
    double ROAD_FARAIM_MAX, ROAD_FARAIM_MIN;      /* Planning.cpp:260,264 */
    double PRE_INTER_FARAIM, INTER_FARAIM;        /* Planning.cpp:275,283 */
    double ROAD_REMAIN_DISTANCE, INTER_REMAIN_DISTANCE; /* Planning.cpp:821,826 */
    double EPSILON, PI;                           /* Planning.cpp:690,728 */
    double Vehicle_Width;                         /* Decision.cpp:370 */
    double NO_OBSTACLE_DIS;                       /* value SearchObstacle leaves in dis_lat/dis_lng when nothing is found */
    /* GlobalToWGS84: local equirectangular frame (projection is unpinned) */
    double wgs_lat0, wgs_lng0, wgs_deg_per_m_lat, wgs_deg_per_m_lng;
    int32_t ID_MORE;                              /* Decision.cpp:581 */
    int32_t decision_stage;                       /* 1: run the CDecision corridor queries + sweep on device */
    int32_t lanechg_stage;                        /* 1: run the lane-change rule tree (Decision.cpp:1017-1772) when the map allows a change */
    /* grid engine (build-defined) */
    int32_t grid_stage;                           /* 1: run G1-G3 every tick */
    int32_t grid_w, grid_h;                       /* cells */
    int32_t max_expansions, bucket_cap, max_path;
    int32_t n_lattice, lookahead_cells;
    int32_t dynamic_obstacles;                    /* 1: obstacle j sits at p0 + v*(dyn_dt*tick) */
    int32_t force_replan;                         /* 1: config "replan every tick" (BASELINE configs[3]) */
    int32_t _cfg_pad;
    double cell;                                  /* metres per cell */
    double inflate;                               /* added to every footprint radius when rasterising */
    double lattice_step, d_safe, w_col, w_curv, w_prog, w_off, dyn_dt;
} PlannerConfig;

#ifdef __cplusplus
}
#endif
#endif /* DMPP_TYPES_H */
