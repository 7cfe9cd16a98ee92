// dmpp_host.cpp — the C++ class surface over libdmpp.so.  No arithmetic of the planning path is done
// here: every method marshals its arguments into the C-ABI and reads the result back.
#include "dmpp_decision.hpp"
#include <cstdlib>
#include <string>
#include <cstring>
#include <algorithm>

namespace {
PlannerConfig g_cfg;
std::once_flag g_cfg_once;
thread_local DmppStatus t_status;
thread_local std::string t_text;
typedef std::lock_guard<std::recursive_mutex> Lock;
}  // namespace

// ---------------------------------------------------------------------------------------- CShare
PlannerConfig& CShare::Config()
{
    std::call_once(g_cfg_once, [] { pp_default_config(&g_cfg, 512, 512); });
    return g_cfg;
}
DmppStatus& CShare::LastStatus() { return t_status; }
void CShare::note(int rc)
{
    t_status.code = rc;
    if (rc) { t_text = pp_last_error(); t_status.text = t_text.c_str(); } else t_status.text = "";
}
CShare::CShare() {}
CShare::~CShare()
{
    Lock g(m_mu);
    if (m_io) pp_host_free(m_io);
    if (m_h) pp_destroy(m_h);
}
pp_handle CShare::Device()
{
    CPlanning& p = CPlanning::Instance();
    Lock g(p.m_mu);
    return p.handle(false, false);
}
pp_handle CShare::op_handle() { return m_h ? m_h : handle(m_decision_object, false); }
void CShare::SetMap(const LaneMap& map)
{
    Lock g(m_mu);
    m_map = map; m_map_dirty = true;
}

// This object's context, configured for the tick about to run.  pp_set_config drains the handle, so it is only called when the
// configuration really differs from the one applied (CDecision always ticks with the decision stage on and no grid; CPlanning
// with the decision stage off: neither flips anything from call to call).
pp_handle CShare::handle(bool decision_stage, bool grid_stage)
{
    PlannerConfig c = Config();
    c.decision_stage = decision_stage ? 1 : 0;
    if (!grid_stage) c.grid_stage = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (!m_h) {
            PlannerCaps caps{};
            caps.max_scenes = 1; caps.max_obs_total = PP_IO_MAX_OBS; caps.max_lane_pts_total = 1 << 16; caps.max_ref_pts_total = DMPP_MAX_REFPATH;
            const char* dv = std::getenv("DMPP_DEVICE");
            PlannerConfig cc = c;
            cc.grid_stage = (m_grid_capable || c.grid_stage) ? 1 : 0;      // the buffers of the grid stage exist whether or not the first call uses it
            if (cc.grid_stage) { cc.grid_w = Config().grid_w; cc.grid_h = Config().grid_h; }
            int rc = pp_create(&cc, dv ? std::atoi(dv) : 0, &caps, &m_h);
            note(rc);
            if (rc) { m_h = nullptr; return nullptr; }
            m_have_applied = false; m_map_dirty = true;
            if (c.grid_stage) m_grid_capable = true;
            if (!m_io) m_io = static_cast<PpSceneIo*>(pp_host_alloc(sizeof(PpSceneIo)));
            if (!m_io) { note(PP_ERR_HIP); return nullptr; }
        }
        if (m_have_applied && std::memcmp(&m_applied, &c, sizeof(c)) == 0) break;
        const int rc = pp_set_config(m_h, &c);
        if (rc == PP_OK) { m_applied = c; m_have_applied = true; break; }
        if ((rc == PP_ERR_CAPACITY || rc == PP_ERR_STATE) && attempt == 0) {   // a larger grid than the context was made for (or none): a new one
            pp_destroy(m_h); m_h = nullptr;
            if (c.grid_stage) m_grid_capable = true;
            continue;
        }
        note(rc);
        return nullptr;
    }
    if (m_map_dirty) { const int rc = upload_map(m_h); if (rc) { note(rc); return nullptr; } }
    return m_h;
}

// The LaneMap as the handle's one resident scene: lane points (current | left | right), their lane-change attributes, and room
// for any refpath in the refpath pool.  Per tick only location, decision, obstacles, refpath and state travel (tick_io).
int CShare::upload_map(pp_handle h)
{
    const LaneMap& m = m_map;
    vector<GlobalPoint3D> lanes;
    SceneIn in{};
    in.lanes.cur_off = 0; in.lanes.cur_n = (int)m.cur.size();
    lanes.insert(lanes.end(), m.cur.begin(), m.cur.end());
    in.lanes.left_off = (int)lanes.size(); in.lanes.left_n = (int)m.left.size();
    lanes.insert(lanes.end(), m.left.begin(), m.left.end());
    in.lanes.right_off = (int)lanes.size(); in.lanes.right_n = (int)m.right.size();
    lanes.insert(lanes.end(), m.right.begin(), m.right.end());
    if (lanes.empty()) lanes.push_back(GlobalPoint3D{0, 0, 0});
    vector<uint8_t> attr(lanes.size(), (uint8_t)m.lanechg_attribute);
    for (size_t i = 0; i < m.cur_lanechg_attribute.size() && i < m.cur.size(); i++) attr[i] = m.cur_lanechg_attribute[i];
    for (int i = 0; i < DMPP_LANESUM; i++) in.out_lane_no[i] = m.out_lane_no[i];
    in.lanes.lane_sum = m.lane_sum; in.lanes.lanechg_attribute = m.lanechg_attribute; in.lanes.lane_width = m.lane_width;
    vector<GlobalPoint2D> ref(DMPP_MAX_REFPATH, GlobalPoint2D{0, 0});
    ObPoint ob{0, 0, 0, 0};
    const int rc = pp_set_scenes(h, 1, &in, lanes.data(), attr.data(), (int)lanes.size(), ref.data(), (int)ref.size(), &ob, nullptr, 1);
    if (rc) return rc;
    m_in_template = in; m_map_dirty = false;
    return PP_OK;
}

int CShare::tick_io(bool decision_stage, const LocationOut& loc, const DecisionOutPod& dec, const vector<GlobalPoint2D>& ref,
                    const vector<ObPoint>& obs, int stub_attribute, double period_last_ms, const GlobalPoint2D* origin, const GlobalPoint2D* goal,
                    SceneState& st, PlanOut& out, GridOut* grid, vector<GlobalPoint2D>* refpath_out)
{
    pp_handle h = handle(decision_stage, grid != nullptr);
    if (!h) return t_status.code ? t_status.code : PP_ERR_HIP;
    if (obs.size() > (size_t)PP_IO_MAX_OBS) {
        t_text = "more obstacles than one PpSceneIo block carries (PP_IO_MAX_OBS)";
        t_status.code = PP_ERR_CAPACITY; t_status.text = t_text.c_str();
        return PP_ERR_CAPACITY;
    }
    PpSceneIo& io = *m_io;
    io.in = m_in_template;
    io.in.loc = loc; io.in.dec = dec;
    const size_t n_ref = std::min(ref.size(), (size_t)DMPP_MAX_REFPATH);
    io.in.dec.refpath_n = (int)n_ref;
    io.in.stub_attribute = stub_attribute; io.in.period_last = period_last_ms;
    io.in.goal = GlobalPoint2D{loc.globalpoint.x, loc.globalpoint.y};
    if (origin) io.in.grid_origin = *origin;
    if (goal) io.in.goal = *goal;
    io.n_obs = (int)obs.size(); io.n_ref = (int)n_ref;
    if (!obs.empty()) std::memcpy(io.obs, obs.data(), obs.size() * sizeof(ObPoint));
    if (n_ref) std::memcpy(io.ref, ref.data(), n_ref * sizeof(GlobalPoint2D));
    io.state = st;
    io.want = (grid && Config().grid_stage ? PP_IO_WANT_GRID : 0) | (refpath_out ? PP_IO_WANT_REFPATH : 0);
    const int rc = pp_tick_io(h, &io);
    note(rc);
    if (rc && rc != PP_ERR_ARG) return rc;
    st = io.state; out = io.plan;
    if (grid && (io.want & PP_IO_WANT_GRID)) *grid = io.grid;
    if (refpath_out) {
        const int n = std::min(std::max(io.plan.dec.refpath_n, 0), DMPP_MAX_REFPATH);
        refpath_out->assign(io.dec_ref, io.dec_ref + n);
    }
    return rc;
}

void CShare::BezierPlanning(GlobalPoint3D start, GlobalPoint3D end, GlobalPoint2D out[], int n)
{ Lock g(m_mu); note(pp_bezier(op_handle(), start, end, out, n)); }
void CShare::MeanPoints(GlobalPoint2D in[], int n_in, GlobalPoint2D out[], int n_out)
{ Lock g(m_mu); note(pp_mean_points(op_handle(), in, n_in, out, n_out)); }
bool CShare::SearchObstacle(vector<GlobalPoint2D> path, vector<ObPoint> obs, double lat_lo, double lat_hi,
                            double& dis_lat, double& dis_lng, ObPoint& ob, WORD& path_id)
{
    Lock g(m_mu);
    int32_t poff[2] = {0, (int32_t)path.size()}, ooff[2] = {0, (int32_t)obs.size()};
    Path_Obs r{};
    GlobalPoint2D dp{}; ObPoint dob{};
    note(pp_search_obstacle_batch(op_handle(), 1, path.empty() ? &dp : path.data(), poff, obs.empty() ? &dob : obs.data(), ooff,
                                  &lat_lo, &lat_hi, &r));
    dis_lat = r.Ob_Pose.dis_lat; dis_lng = r.Ob_Pose.dis_lng; ob = r.Ob_Attr; path_id = (WORD)r.Ob_Pathid;
    return r.Obs_flag != 0;
}
vector<GlobalPoint2D> CShare::CreateNewPath(vector<GlobalPoint2D> path, double offset)
{
    Lock g(m_mu);
    vector<GlobalPoint2D> out(path.size());
    if (!path.empty()) note(pp_create_new_path(op_handle(), path.data(), (int)path.size(), offset, out.data()));
    return out;
}
double CShare::CalcDistance(GlobalPoint2D a, GlobalPoint2D b)
{ Lock g(m_mu); double v = 0; note(pp_geom_batch(op_handle(), 3, 1, &a, &b, nullptr, &v)); return v; }
double CShare::CalcGlobalDir(GlobalPoint2D a, GlobalPoint2D b)
{ Lock g(m_mu); double v = 0; note(pp_geom_batch(op_handle(), 1, 1, &a, &b, nullptr, &v)); return v; }
GPSPoint2D CShare::GlobalToWGS84(GlobalPoint2D p)
{
    Lock lk(m_mu);
    GPSPoint2D g{};
    note(pp_geom_batch(op_handle(), 4, 1, &p, nullptr, nullptr, &g.lat));
    note(pp_geom_batch(op_handle(), 5, 1, &p, nullptr, nullptr, &g.lng));
    return g;
}
int CShare::NearestId(GlobalPoint2D p, vector<GlobalPoint2D> path)
{   // the argmin loop of Planning.cpp:640-650 (first minimum); distances from the device
    if (path.empty()) return 0;
    Lock g(m_mu);
    vector<GlobalPoint2D> a(path.size(), p);
    vector<double> d(path.size());
    note(pp_geom_batch(op_handle(), 3, (int)path.size(), a.data(), path.data(), nullptr, d.data()));
    return (int)(std::min_element(d.begin(), d.end()) - d.begin());
}
double CShare::LatDis(GlobalPoint2D p, GlobalPoint2D a, GlobalPoint2D b)
{ Lock g(m_mu); double v = 0; note(pp_geom_batch(op_handle(), 0, 1, &p, &a, &b, &v)); return v; }

// ------------------------------------------------------------------------------------- CPlanning
CPlanning::CPlanning() { m_grid_capable = true; Reset(); }
CPlanning& CPlanning::Instance() { static CPlanning thePlanning; return thePlanning; }
BYTE CPlanning::startCPlanningThread() { Lock g(m_mu); return handle(false, false) ? 1 : 0; }
void CPlanning::Reset()
{
    Lock g(m_mu);
    pp_init_state(&m_state, 1);
    his_behavior = 1; path_near_id = path_front_near_id = 0;
}

void CPlanning::tick(const DecisionOut& dec, const LocationOut& loc, const vector<ObPoint>& obs, SceneState& st,
                     PlanOut& out, GridOut* grid, bool decision_stage)
{
    (void)tick_io(decision_stage, loc, dec, dec.refpath, obs, 0, 100.0, m_frame ? &m_origin : nullptr, m_frame ? &m_goal : nullptr,
                  st, out, grid, nullptr);
}

void CPlanning::plan(const DecisionOut& decision, const LocationOut& location, const VehStatus&, const vector<ObPoint>& obstacles,
                     PlanningOut& result, PlanningStatus& show, GlobalPoint2D road_points[], GridOut* grid)
{
    Lock g(m_mu);
    PlanOut out{};
    m_state.his_behavior = his_behavior;
    tick(decision, location, obstacles, m_state, out, grid, false);
    result = out.result; show = out.show;
    if (road_points) std::memcpy(road_points, out.road_points, sizeof(out.road_points));
    // mirror the public members the reference exposes (Planning.h:42-52)
    path_lat_dis = m_state.path_lat_dis; afresh_planning = m_state.afresh_planning != 0; afresh_cause = m_state.afresh_cause;
    remain_dis = m_state.remain_dis; path_dir_err = m_state.path_dir_err; path_near_id = m_state.path_near_id;
    path_front_near_id = m_state.path_front_near_id; his_behavior = m_state.his_behavior; brakespeed = m_state.brakespeed;
    acc_flag = m_state.acc_flag != 0; des_acc = m_state.des_acc;
    faraim_dis = m_state.faraim_dis; nearaim_dis = m_state.nearaim_dis;
    aimpoint_far = m_state.aimpoint_far; aimpoint_near = m_state.aimpoint_near;
}

// The stage methods below run the device tick on a SCRATCH copy of the state and return the stage's own
// outputs (each of these stages stores its outputs in SceneState untouched by later stages).
void CPlanning::Calculate_aim_dis(DecisionOut dec, LocationOut loc, VehStatus, FLOAT& far_, FLOAT& near_)
{
    Lock g(m_mu);
    SceneState st = m_state; PlanOut out{};
    tick(dec, loc, {}, st, out, nullptr, false);
    far_ = st.faraim_dis; near_ = st.nearaim_dis;
    faraim_dis = far_; nearaim_dis = near_;           // the reference writes the members (Planning.cpp:118)
}
void CPlanning::SearchAimPoint(DecisionOut dec, LocationOut loc, VehStatus, AimPoint& far_, AimPoint& near_)
{
    Lock g(m_mu);
    SceneState st = m_state; PlanOut out{};
    st.aimpoint_far = far_; st.aimpoint_near = near_;  // branches that assign nothing keep the caller's values
    tick(dec, loc, {}, st, out, nullptr, false);
    far_ = st.aimpoint_far; near_ = st.aimpoint_near;
}
void CPlanning::InitialPlanning(DecisionOut, LocationOut loc, VehStatus, const AimPoint aimpoint_far, const AimPoint,
                                GlobalPoint2D Bezier_points[])
{ BezierPlanning(loc.globalpoint, aimpoint_far.Aim_point, Bezier_points, DMPP_PATH_POINTS); }           // Planning.cpp:596-611
void CPlanning::GetVhclLocalState(LocationOut loc, const GlobalPoint2D last_Bpoints[], double& mindist_lat, double& dir_err,
                                  int& mindist_id, int& front_mindist_id, double& remain)
{
    Lock g(m_mu);
    SceneState st = m_state; PlanOut out{};
    std::memcpy(st.last_Bpoints, last_Bpoints, sizeof(st.last_Bpoints));
    st.count = 1;                                      // not the first tick: no InitialPlanning
    st.path_near_id = mindist_id;
    DecisionOut dec; dec.behavior = st.his_behavior; dec.target_lanenum = loc.lane_num;
    tick(dec, loc, {}, st, out, nullptr, false);
    mindist_lat = st.path_lat_dis; dir_err = st.path_dir_err; mindist_id = st.path_near_id;
    front_mindist_id = st.path_front_near_id; remain = st.remain_dis;
}
bool CPlanning::UpdatePlanJudge(const DecisionOut dec, const LocationOut loc, const int last_behavior, int& afreshcause)
{
    Lock g(m_mu);
    const double in[6] = {(double)last_behavior, (double)dec.behavior, (double)loc.pos, path_lat_dis, path_dir_err, remain_dis};
    double out[2] = {0, 0};
    note(pp_scalar_stage(op_handle(), 0, in, 6, nullptr, out, 2));
    afreshcause = (int)out[1];
    return out[0] != 0;
}
void CPlanning::PathPlanning(const DecisionOut dec, int, LocationOut loc, const AimPoint aimpoint_far, const AimPoint,
                             GlobalPoint2D road_points[])
{   // Planning.cpp:845-877
    if (loc.pos == 0) BezierPlanning(loc.globalpoint, aimpoint_far.Aim_point, road_points, DMPP_PATH_POINTS);
    else if (loc.pos == 1 || loc.pos == 2) {
        int na = std::min(std::min(aimpoint_far.Aim_id, DMPP_PATH_POINTS), (int)dec.refpath.size());
        if (na < 0) na = 0;
        vector<GlobalPoint2D> ref(dec.refpath.begin(), dec.refpath.begin() + na);
        GlobalPoint2D dummy{};
        MeanPoints(ref.empty() ? &dummy : ref.data(), na, road_points, DMPP_PATH_POINTS);
    } else std::memset(road_points, 0, sizeof(GlobalPoint2D) * DMPP_PATH_POINTS);
}
void CPlanning::SpeedPlanning(const bool ob_flag, const DecisionOut dec, const LocationOut loc, const double mindist_lon,
                              const double, const FLOAT far_, double& brake_speed, bool& accf, double& desacc)
{
    Lock g(m_mu);
    const double in[8] = {(double)loc.pos, ob_flag ? 1.0 : 0.0, mindist_lon, (double)far_, dec.velocity_expect, brake_speed,
                          accf ? 1.0 : 0.0, desacc};
    double out[3] = {0, 0, 0};
    note(pp_scalar_stage(op_handle(), 1, in, 8, nullptr, out, 3));
    brake_speed = out[0]; accf = out[1] != 0; desacc = out[2];
}
double CPlanning::GetLatDis(GlobalPoint2D cur_pt, GlobalPoint2D pt, GlobalPoint2D pt_next) { return LatDis(cur_pt, pt, pt_next); }
double CPlanning::GetRoadAngle(GlobalPoint2D a, GlobalPoint2D b) { return CalcGlobalDir(a, b); }
double CPlanning::GetAngleErr(double dir1, double dir2)
{ Lock g(m_mu); GlobalPoint2D a{dir1, dir2}; double v = 0; note(pp_geom_batch(op_handle(), 2, 1, &a, nullptr, nullptr, &v)); return v; }
double CPlanning::CalculateRadius()
{
    Lock g(m_mu);
    const double in[2] = {(double)path_near_id, (double)path_front_near_id};
    double out[1] = {0};
    note(pp_scalar_stage(op_handle(), 2, in, 2, m_state.last_Bpoints, out, 1));
    return out[0];
}

// ------------------------------------------------------------------------------------- CDecision
CDecision::CDecision() { m_decision_object = true; Reset(); }
CDecision& CDecision::Instance() { static CDecision theDecision; return theDecision; }
BYTE CDecision::startCDecisionThread() { Lock g(m_mu); return handle(true, false) ? 1 : 0; }
void CDecision::Reset() { Lock g(m_mu); pp_init_state(&m_state, 1); }

DecisionOut CDecision::decide(const LocationOut& location, const vector<ObPoint>& obstacles,
                               const vector<GlobalPoint2D>& junction_polyline, int stub_attribute, Path_Obs around[6],
                               double period_last_ms)
{
    Lock g(m_mu);
    DecisionOutPod none{};
    if (m_state.tick == 0 && m_state.z_target_lanenum != location.lane_num) {
        m_state.z_target_lanenum = location.lane_num; m_state.d_his_target_lanenum = location.lane_num;
    }
    PlanOut out{};
    DecisionOut d;
    (void)tick_io(true, location, none, junction_polyline, obstacles, stub_attribute, period_last_ms, nullptr, nullptr, m_state, out, nullptr, &d.refpath);
    static_cast<DecisionOutPod&>(d) = out.dec;
    d.refpath_n = (int32_t)d.refpath.size();
    if (around) std::memcpy(around, out.around, sizeof(out.around));
    return d;
}

// ------------------------------------------------------------------------------------- C entry points
// The class surface for callers without a C++ compiler at hand (the ctypes harness of tests/ and bench.py): each function
// below is one or two calls on the singletons above and nothing else.
extern "C" {

struct DmppHostMembers {          // the public members of CPlanning (Planning.h:42-52) after a plan()
    double path_lat_dis, remain_dis, path_dir_err, brakespeed, des_acc;
    int32_t afresh_planning, afresh_cause, path_near_id, path_front_near_id, his_behavior, acc_flag;
};

int dmpp_host_start(void)
{ return (CDecision::Instance().startCDecisionThread() && CPlanning::Instance().startCPlanningThread()) ? 0 : CShare::LastStatus().code; }

const char* dmpp_host_error(void) { return CShare::LastStatus().text; }

PlannerConfig* dmpp_host_config(void) { return &CShare::Config(); }

void dmpp_host_reset(int lane_num)
{
    CDecision::Instance().Reset(); CPlanning::Instance().Reset();
    (void)lane_num;
}

void dmpp_host_set_map(const GlobalPoint3D* cur, int n_cur, const GlobalPoint3D* left, int n_left, const GlobalPoint3D* right, int n_right,
                       int lane_sum, int lanechg_attribute, double lane_width, const uint8_t* cur_attr, int n_attr, const uint16_t* out_lane_no)
{
    LaneMap m;
    if (n_cur > 0) m.cur.assign(cur, cur + n_cur);
    if (n_left > 0) m.left.assign(left, left + n_left);
    if (n_right > 0) m.right.assign(right, right + n_right);
    m.lane_sum = lane_sum; m.lanechg_attribute = lanechg_attribute; m.lane_width = lane_width;
    if (n_attr > 0) m.cur_lanechg_attribute.assign(cur_attr, cur_attr + n_attr);
    for (int i = 0; i < DMPP_LANESUM; i++) m.out_lane_no[i] = out_lane_no ? out_lane_no[i] : 0;
    CDecision::Instance().SetMap(m); CPlanning::Instance().SetMap(m);
}

// CDecision::decide -> CPlanning::plan: one pass of the two reference threads (Decision.cpp:172-205, Planning.cpp:114-223)
int dmpp_host_tick(const LocationOut* loc, const ObPoint* obs, int n_obs, const GlobalPoint2D* junction, int n_junction,
                   int stub_attribute, double period_last_ms,
                   DecisionOutPod* dec_out, GlobalPoint2D* refpath_out, int refpath_cap, Path_Obs* around_out,
                   PlanningOut* result, PlanningStatus* show, GlobalPoint2D* road_points, DmppHostMembers* members, GridOut* grid)
{
    vector<ObPoint> o; if (n_obs > 0) o.assign(obs, obs + n_obs);
    vector<GlobalPoint2D> j; if (n_junction > 0) j.assign(junction, junction + n_junction);
    DecisionOut d = CDecision::Instance().decide(*loc, o, j, stub_attribute, around_out, period_last_ms);
    if (CShare::LastStatus().code) return CShare::LastStatus().code;
    if (dec_out) { *dec_out = d; dec_out->refpath_n = (int32_t)d.refpath.size(); }
    for (int i = 0; i < refpath_cap && i < (int)d.refpath.size(); i++) refpath_out[i] = d.refpath[(size_t)i];
    CPlanning& pl = CPlanning::Instance();
    pl.plan(d, *loc, VehStatus{}, o, *result, *show, road_points, grid);
    if (members) {
        members->path_lat_dis = pl.path_lat_dis; members->remain_dis = pl.remain_dis; members->path_dir_err = pl.path_dir_err;
        members->brakespeed = pl.brakespeed; members->des_acc = pl.des_acc; members->afresh_planning = pl.afresh_planning;
        members->afresh_cause = pl.afresh_cause; members->path_near_id = pl.path_near_id; members->path_front_near_id = pl.path_front_near_id;
        members->his_behavior = pl.his_behavior; members->acc_flag = pl.acc_flag;
    }
    return CShare::LastStatus().code;
}

// CPlanning::plan alone, on a DecisionOut the caller holds (the p50 of the class surface is quoted on this call)
int dmpp_host_plan(const DecisionOutPod* dec, const GlobalPoint2D* refpath, int n_refpath, const LocationOut* loc, const ObPoint* obs, int n_obs,
                   PlanningOut* result, PlanningStatus* show, GlobalPoint2D* road_points, GridOut* grid)
{
    DecisionOut d; static_cast<DecisionOutPod&>(d) = *dec;
    if (n_refpath > 0) d.refpath.assign(refpath, refpath + n_refpath);
    vector<ObPoint> o; if (n_obs > 0) o.assign(obs, obs + n_obs);
    CPlanning::Instance().plan(d, *loc, VehStatus{}, o, *result, *show, road_points, grid);
    return CShare::LastStatus().code;
}

void dmpp_host_set_grid_frame(double ox, double oy, double gx, double gy)
{ CPlanning::Instance().SetGridFrame(GlobalPoint2D{ox, oy}, GlobalPoint2D{gx, gy}); }

const SceneState* dmpp_host_planning_state(void) { return &CPlanning::Instance().State(); }
const SceneState* dmpp_host_decision_state(void) { return &CDecision::Instance().State(); }

}  // extern "C"
