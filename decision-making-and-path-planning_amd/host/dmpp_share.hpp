// dmpp_share.hpp — C++ host surface, part 1: the types and the CShare base class.
//
// The reference's Share.h is not shipped; this header supplies what Planning.h:2 /
// Decision.h:2 expect from it, on top of the C-ABI (include/dmpp_planner.h).  Every helper runs on the
// GPU through libdmpp.so — there is no host implementation of the arithmetic.
#pragma once
#include <vector>
#include <mutex>
#include "../../include/dmpp_planner.h"

using std::vector;

// Win32 scalar names the reference uses (stdafx.h is not shipped either)
typedef unsigned char  BYTE;
typedef unsigned short WORD;
typedef unsigned long  DWORD;
typedef unsigned int   UINT;
typedef int            INT;
typedef int            BOOL;
typedef float          FLOAT;
typedef double         DOUBLE;

// VehStatus is copied around but never read on the path (Planning.cpp:106)
struct VehStatus { double reserved[4]; };

// DecisionOut exactly as the reference builds and passes it (Decision.cpp:187-196; Planning.h:57-75 take it by value):
// period_max, period_last, behavior, target_roadnum, target_lanenum, light, velocity_expect, refpath, behavior_to_dlg.
// The scalar fields are those of the C-ABI's DecisionOutPod (refpath_n is kept equal to refpath.size() at the boundary).
struct DecisionOut : public DecisionOutPod {
    double period_max = 0, period_last = 0;
    vector<GlobalPoint2D> refpath;
    DecisionOut() : DecisionOutPod{} {}
};

// planning_MapData[road][lane] as this boundary sees it: the current lane and its neighbours
struct LaneMap {
    vector<GlobalPoint3D> cur, left, right;
    int lane_sum = 1, lanechg_attribute = 0;
    double lane_width = 3.75;
    // decision_MapData[..][..][id].lanechg_attribute of every point of `cur` (Decision.cpp:1179); when shorter than
    // `cur` the remainder takes `lanechg_attribute`
    vector<BYTE> cur_lanechg_attribute;
    // z_RoadNavi[path_num].out_lane_no (Decision.cpp:696): exit lanes of this road, 0-terminated
    WORD out_lane_no[DMPP_LANESUM] = { 0 };
};

// Raised into a flag instead of AfxMessageBox (Decision.cpp:514): the reference's methods return void
struct DmppStatus { int code = 0; const char* text = ""; };

// Threads.  The reference runs CDecision and CPlanning on a thread each (Decision.cpp:45, Planning.cpp:27), every object used
// by its own thread only.  Here every object owns its GPU context - a pp_handle with its resident lane map and one pinned
// PpSceneIo block - behind its own mutex: decide() and plan() may run concurrently on two threads, and two threads sharing
// ONE object are serialised call by call.  The last status is per thread.  Config() is process-wide and read when a context
// is (re)created and on every tick: set it before the threads start (the reference's macros are compile-time constants).
class CShare {
public:
    CShare();
    virtual ~CShare();
    CShare(const CShare&) = delete;
    CShare& operator=(const CShare&) = delete;
    // the planning singleton's context (created on first use); device: env DMPP_DEVICE, default 0
    static pp_handle Device();
    static PlannerConfig& Config();
    static DmppStatus& LastStatus();                                                                 // of the calling thread

    void   BezierPlanning(GlobalPoint3D start, GlobalPoint3D end, GlobalPoint2D out[], int n);       // Planning.cpp:606,863
    void   MeanPoints(GlobalPoint2D in[], int n_in, GlobalPoint2D out[], int n_out);                 // Planning.cpp:872
    bool   SearchObstacle(vector<GlobalPoint2D> path, vector<ObPoint> obs, double lat_lo, double lat_hi,
                          double& dis_lat, double& dis_lng, ObPoint& ob, WORD& path_id);             // Planning.cpp:168
    vector<GlobalPoint2D> CreateNewPath(vector<GlobalPoint2D> path, double offset);                  // Decision.cpp:629,942
    double CalcDistance(GlobalPoint2D a, GlobalPoint2D b);                                           // Planning.cpp:509
    double CalcGlobalDir(GlobalPoint2D a, GlobalPoint2D b);                                          // Planning.cpp:519
    GPSPoint2D GlobalToWGS84(GlobalPoint2D p);                                                       // Planning.cpp:209
    int    NearestId(GlobalPoint2D p, vector<GlobalPoint2D> path);                                   // Decision.cpp:1889
    double LatDis(GlobalPoint2D p, GlobalPoint2D a, GlobalPoint2D b);                                // Decision.cpp:1895
    void   SetMap(const LaneMap& map);             // planning_MapData / decision_MapData of this object: uploaded once, resident
protected:
    static void note(int rc);
    // this object's context; the methods below expect m_mu to be held
    pp_handle handle(bool decision_stage, bool grid_stage);    // (re)creates the context / applies the stage switches when they differ
    int upload_map(pp_handle h);                               // the LaneMap as the handle's ONE resident scene
    // one tick of this object's scene through its PpSceneIo block (one host wait); state in / out, PlanOut (+ GridOut, refpath) out
    int tick_io(bool decision_stage, const LocationOut& loc, const DecisionOutPod& dec, const vector<GlobalPoint2D>& ref,
                const vector<ObPoint>& obs, int stub_attribute, double period_last_ms, const GlobalPoint2D* origin, const GlobalPoint2D* goal,
                SceneState& st, PlanOut& out, GridOut* grid, vector<GlobalPoint2D>* refpath_out);
    pp_handle op_handle();                                     // this object's context for a stand-alone operator (created when absent)
    std::recursive_mutex m_mu;
    LaneMap m_map;
    bool m_grid_capable = false;                               // the context is created with the buffers of the grid stage (CPlanning)
    bool m_decision_object = false;                            // the object ticks with the decision stage on (CDecision)
private:
    pp_handle m_h = nullptr;
    PpSceneIo* m_io = nullptr;
    PlannerConfig m_applied{}; bool m_have_applied = false;
    bool m_map_dirty = true;
    SceneIn m_in_template{};                     // lane slices of the resident map
};
