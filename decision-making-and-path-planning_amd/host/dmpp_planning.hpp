// dmpp_planning.hpp — C++ host surface, part 2: class CPlanning with the public surface of the
// reference (Planning.h:38-85: same method names, parameter order and in/out conventions) plus plan(),
// the factored tick body of CPlanningThread (Planning.cpp:114-223).  All arithmetic runs on the GPU.
#pragma once
#include "dmpp_share.hpp"

class CPlanning : public CShare {
public:
    static CPlanning& Instance();                 // Planning.cpp:18-22
    BYTE startCPlanningThread();                  // Planning.cpp:24-33: no thread here, the caller drives plan(); returns 1

    // public state of the reference (Planning.h:42-52)
    double path_lat_dis = 0;
    bool   afresh_planning = false;
    int    afresh_cause = 0;
    double remain_dis = 0;
    double path_dir_err = 0;
    int    path_near_id = 0;
    int    path_front_near_id = 0;
    int    his_behavior = 1;                      // Planning.cpp:10
    double brakespeed = 0;
    bool   acc_flag = false;
    double des_acc = 0;

    inline int Sgn(double a) { return a > 0 ? 1 : -1; }          // Planning.h:54

    void Calculate_aim_dis(DecisionOut decision_result, LocationOut vhcl_location, VehStatus vhcl_status,
                           FLOAT& faraim_dis, FLOAT& nearaim_dis);
    void SearchAimPoint(DecisionOut decision_result, LocationOut vhcl_location, VehStatus vhcl_status,
                        AimPoint& aimpoint_far, AimPoint& aimpoint_near);
    void InitialPlanning(DecisionOut decision_result, LocationOut vhcl_location, VehStatus vhcl_status,
                         const AimPoint aimpoint_far, const AimPoint aimpoint_near, GlobalPoint2D Bezier_points[]);
    void GetVhclLocalState(LocationOut vhcl_location, const GlobalPoint2D last_Bpoints[], double& mindist_lat,
                           double& path_dir_err, int& mindist_id, int& front_mindist_id, double& remain_dis);
    bool UpdatePlanJudge(const DecisionOut decision_result, const LocationOut vhcl_location, const int his_behavior,
                         int& afreshcause);
    void PathPlanning(const DecisionOut z_DecisionOut, int afresh_cause, LocationOut vhcl_location,
                      const AimPoint aimpoint_far, const AimPoint aimpoint_near, GlobalPoint2D road_points[]);
    void SpeedPlanning(const bool ob_flag, const DecisionOut decision_result, const LocationOut vhcl_location,
                       const double mindist_lon, const double mindist_lat, const FLOAT faraim_dis, double& brakespeed,
                       bool& acc_flag, double& des_acc);
    double GetLatDis(GlobalPoint2D cur_pt, GlobalPoint2D pt, GlobalPoint2D pt_next);
    double GetRoadAngle(GlobalPoint2D apoint, GlobalPoint2D bpoint);
    double GetAngleErr(double dir1, double dir2);
    double CalculateRadius();

    // ---- what replaces the blackboard (Planning.cpp:95-112) and the while(true) body ----
    // (SetMap: CShare)
    // Frame of the grid stage (the search the reference only reserves behaviour code 6 "A*" for, Decision.h:36): world
    // position of the corner of cell (0,0) and the goal.  Without it plan(..., grid) searches from the ego to itself.
    void SetGridFrame(GlobalPoint2D origin, GlobalPoint2D goal) { m_origin = origin; m_goal = goal; m_frame = true; }
    // One planning tick: the inputs the thread copies from the app, the outputs it hands back
    // (SetUdpSendCtrl / SetPlanningStatus, Planning.cpp:186,214).  road_points: 200 points, may be null.
    void plan(const DecisionOut& decision, const LocationOut& location, const VehStatus& status,
              const vector<ObPoint>& obstacles, PlanningOut& result, PlanningStatus& show,
              GlobalPoint2D road_points[] = nullptr, GridOut* grid = nullptr);
    const SceneState& State() const { return m_state; }
    void Reset();

private:
    CPlanning();
    ~CPlanning() {}
    void tick(const DecisionOut& dec, const LocationOut& loc, const vector<ObPoint>& obs, SceneState& st,
              PlanOut& out, GridOut* grid, bool decision_stage);
    friend class CDecision;
    GlobalPoint2D m_origin{0, 0}, m_goal{0, 0}; bool m_frame = false;
    SceneState m_state;             // last_Bpoints, count, aim points ... (Planning.cpp:6,216-223)
    FLOAT faraim_dis = 0, nearaim_dis = 0;       // Planning.h:20-21
    AimPoint aimpoint_near{}, aimpoint_far{};    // Planning.h:22-23
};
