// example_tick.cpp — the two reference threads as two calls: CDecision::decide -> CPlanning::plan, on one
// synthetic scene, plus a few of the CShare / CPlanning helper methods.  Exit code 0 = ran on the GPU.
#include "dmpp_decision.hpp"
#include <cstdio>
#include <cmath>

int main()
{
    PlannerConfig& cfg = CShare::Config();
    cfg.grid_stage = 0;
    if (!CShare::Device()) { std::fprintf(stderr, "no GPU: %s\n", CShare::LastStatus().text); return 2; }
    // one generated scene -> the map / location / obstacle inputs of the class surface
    SceneIn in; SceneState st;
    std::vector<GlobalPoint3D> lanes(3 * PP_GEN_LANE_PTS);
    std::vector<uint8_t> attr(3 * PP_GEN_LANE_PTS);
    std::vector<GlobalPoint2D> ref(PP_GEN_REF_PTS);
    std::vector<ObPoint> obs(64);
    std::vector<ObMotion> mot(64);
    pp_gen_scenes(&cfg, 2, 1, 64, 0, &in, lanes.data(), attr.data(), ref.data(), obs.data(), mot.data(), &st);
    LaneMap map;
    map.cur.assign(lanes.begin(), lanes.begin() + PP_GEN_LANE_PTS);
    if (in.lanes.left_n) map.left.assign(lanes.begin() + PP_GEN_LANE_PTS, lanes.begin() + 2 * PP_GEN_LANE_PTS);
    if (in.lanes.right_n) map.right.assign(lanes.begin() + 2 * PP_GEN_LANE_PTS, lanes.end());
    map.lane_sum = in.lanes.lane_sum; map.lanechg_attribute = in.lanes.lanechg_attribute; map.lane_width = in.lanes.lane_width;
    map.cur_lanechg_attribute.assign(attr.begin(), attr.begin() + PP_GEN_LANE_PTS);
    for (int i = 0; i < DMPP_LANESUM; i++) map.out_lane_no[i] = in.out_lane_no[i];

    CDecision& dec = CDecision::Instance();
    CPlanning& pl = CPlanning::Instance();
    if (!dec.startCDecisionThread() || !pl.startCPlanningThread()) return 3;
    dec.SetMap(map); pl.SetMap(map);
    PlanningOut result; PlanningStatus show; GlobalPoint2D road[DMPP_PATH_POINTS];
    for (int t = 0; t < 5; t++) {
        DecisionOut d = dec.decide(in.loc, obs);
        pl.plan(d, in.loc, VehStatus{}, obs, result, show, road);
        std::printf("tick %d: behavior %d refpath %zu  desspd %.3f  radius %.2f  replan %d cause %d  near_ob %.2f\n", t, d.behavior,
                    d.refpath.size(), result.desspd, result.radius, (int)pl.afresh_planning, pl.afresh_cause, show.near_ob_dist);
        if (CShare::LastStatus().code) { std::fprintf(stderr, "error: %s\n", CShare::LastStatus().text); return 4; }
    }
    // helper methods of the reference surface, each evaluated on the device
    GlobalPoint2D a{0, 0}, b{1, 1}, c{0, 1};
    const double ang = pl.GetRoadAngle(a, b), err = pl.GetAngleErr(350, 10), lat = pl.GetLatDis(c, a, b), dist = pl.CalcDistance(a, b);
    int cause = 0; const bool judge = pl.UpdatePlanJudge(DecisionOut(), in.loc, 7, cause);
    std::printf("GetRoadAngle %.6f GetAngleErr %.1f GetLatDis %.6f CalcDistance %.6f UpdatePlanJudge %d/%d CalculateRadius %.3f\n",
                ang, err, lat, dist, (int)judge, cause, pl.CalculateRadius());
    const bool ok = std::fabs(ang - 45.0) < 1e-9 && err == 20.0 && std::fabs(lat - std::sqrt(0.5)) < 1e-12 &&
                    std::fabs(dist - std::sqrt(2.0)) < 1e-15 && judge && cause == 1 && std::isfinite(result.desspd);
    std::printf(ok ? "example ok\n" : "example FAILED\n");
    return ok ? 0 : 1;
}
