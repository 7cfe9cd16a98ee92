/*
 * dmpp_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED: the reference (/root/reference) ships no tests, golden vectors or
 * fixtures, and cannot be built here (it needs stdafx.h, Share.h, GAC_Auotpilot_DP.h,
 * GAC_Auotpilot_DPDlg.h, V2XTCP.h and Win32/MFC, none of which exist; writing
 * stand-ins for them is not allowed).  This oracle is therefore a line-cited
 * restatement of the reference's own function bodies (Part R) plus this repo's
 * documented specification for what the reference leaves undefined (the CShare
 * helpers, rows R12-R15, and the whole grid engine, rows G1-G4).  See DESIGN.md §3.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef DMPP_ORACLE_H
#define DMPP_ORACLE_H

#include "../include/dmpp_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Part R: functions whose bodies are in the reference ------------------------- */
int    orc_Sgn(double a);                                                     /* Planning.h:54        */
double orc_GetLatDis(const PlannerConfig*, GlobalPoint2D cur, GlobalPoint2D pt, GlobalPoint2D pt_next); /* Planning.cpp:686-709 */
double orc_GetRoadAngle(const PlannerConfig*, GlobalPoint2D a, GlobalPoint2D b);                       /* Planning.cpp:719-750 */
double orc_GetAngleErr(double dir1, double dir2);                                                      /* Planning.cpp:760-786 */
void   orc_Calculate_aim_dis(const PlannerConfig*, const LocationOut*, float* far_, float* near_);     /* Planning.cpp:242-290 */
int    orc_last_refpath(GlobalPoint2D* out, int cap);   /* refpath published by this thread's last orc_plan_tick; returns its length */
void   orc_SearchAimPoint(const PlannerConfig*, const SceneIn*, const DecisionOutPod*, const GlobalPoint2D* refpath,
                          const GlobalPoint3D* lane_pool, SceneState*);                                /* Planning.cpp:303-583 */
void   orc_GetVhclLocalState(const PlannerConfig*, const LocationOut*, const GlobalPoint2D last_Bpoints[DMPP_PATH_POINTS],
                             double* mindist_lat, double* path_dir_err, int* mindist_id, int* front_mindist_id,
                             double* remain_dis);                                                      /* Planning.cpp:623-676 */
int    orc_UpdatePlanJudge(const PlannerConfig*, const DecisionOutPod*, const LocationOut*, int last_behavior,
                           const SceneState*, int* afreshcause);                                       /* Planning.cpp:797-832 */
void   orc_SpeedPlanning(int ob_flag, const DecisionOutPod*, const LocationOut*, double mindist_lon, double mindist_lat,
                         float faraim_dis, double* brake_speed, int* acc_flag, double* des_acc);       /* Planning.cpp:888-990 */
double orc_CalculateRadius(const GlobalPoint2D last_Bpoints[DMPP_PATH_POINTS], int near_id, int front_id); /* Planning.cpp:1000-1019 */

/* ---- Part R: CShare helpers, bodies missing from the reference; spec in DESIGN.md §4 */
double orc_CalcDistance(GlobalPoint2D a, GlobalPoint2D b);
void   orc_BezierPlanning(const PlannerConfig*, GlobalPoint3D start, GlobalPoint3D end, GlobalPoint2D* out, int n);
void   orc_MeanPoints(const PlannerConfig*, const GlobalPoint2D* in, int n_in, GlobalPoint2D* out, int n_out);
int    orc_CreateNewPath(const PlannerConfig*, const GlobalPoint2D* path, int n, double offset, GlobalPoint2D* out);
int    orc_SearchObstacle(const PlannerConfig*, const GlobalPoint2D* path, int n, const ObPoint* obs, int m,
                          double lat_lo, double lat_hi, double* dis_lat, double* dis_lng, ObPoint* ob, int* path_id);
GPSPoint2D orc_GlobalToWGS84(const PlannerConfig*, GlobalPoint2D p);

/* ---- grid engine (rows G1-G4; specification in DESIGN.md §5) ---------------------- */
void orc_effective_obstacles(const PlannerConfig*, const ObPoint* obs, const ObMotion* mot, int m, int tick, ObPoint* out);
void orc_rasterise(const PlannerConfig*, GlobalPoint2D origin, const ObPoint* obs, int m, uint8_t* grid);
void orc_rasterise_bruteforce(const PlannerConfig*, GlobalPoint2D origin, const ObPoint* obs, int m, uint8_t* grid);
/* order (n_expanded cells) and path (path_len cells) may be NULL; caps are in cells */
void orc_grid_search(const PlannerConfig*, const uint8_t* grid, int start_cell, int goal_cell,
                     GridOut* out, int32_t* order, int order_cap, int32_t* path, int path_cap);
void orc_grid_score(const PlannerConfig*, const SceneIn*, const ObPoint* obs, int m,
                    const int32_t* path, GridOut* out);
int  orc_cell_of(const PlannerConfig*, GlobalPoint2D origin, double x, double y);

/* ---- whole tick for one scene / a batch ------------------------------------------- */
/* grid_scratch: grid_w*grid_h bytes (may be NULL when cfg->grid_stage == 0);
 * grid_out may be NULL.  order/path as in orc_grid_search. */
void orc_plan_tick(const PlannerConfig*, const SceneIn*, const GlobalPoint3D* lane_pool, const uint8_t* lane_attr_pool,
                   const GlobalPoint2D* ref_pool, const ObPoint* obs_pool, const ObMotion* mot_pool, SceneState*, PlanOut*, GridOut*,
                   uint8_t* grid_scratch, int32_t* order, int order_cap, int32_t* path, int path_cap);
/* runs scenes [0,n) on n_threads pthreads; grids (n * w*h bytes) may be NULL -> internal scratch */
void orc_plan_tick_batch(const PlannerConfig*, int n, const SceneIn*, const GlobalPoint3D* lane_pool,
                         const uint8_t* lane_attr_pool, const GlobalPoint2D* ref_pool, const ObPoint* obs_pool, const ObMotion* mot_pool,
                         SceneState*, PlanOut*, GridOut*, uint8_t* grids, int n_threads);
/* the same for n_ticks consecutive ticks: every thread runs its scenes through all of them (outputs of the last tick) */
void orc_plan_ticks_batch(const PlannerConfig*, int n, const SceneIn*, const GlobalPoint3D* lane_pool,
                          const uint8_t* lane_attr_pool, const GlobalPoint2D* ref_pool, const ObPoint* obs_pool, const ObMotion* mot_pool,
                          SceneState*, PlanOut*, GridOut*, uint8_t* grids, int n_threads, int n_ticks);

#ifdef __cplusplus
}
#endif
#endif
