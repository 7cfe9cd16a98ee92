// scenegen.cpp — synthetic scene generator and default PlannerConfig (host code).
//
// The reference ships no maps, logs or recorded scenes, and fixes none of
// its tuning macros (§2.3).  This file is the one place both are chosen; SURVEY.md §8(d)
// gives the recipe: SplitMix64 seeded 0x5EED0000 + scene_index, world = W*cell metres,
// ego in the left 10 % band heading +-15 deg about +x, goal in the right 10 % band, lane =
// straight + sinusoid at 0.5 m spacing, obstacles uniform (>= 3 m from the ego), radius
// U[0.3,1.5] m, dynamic obstacles with constant velocity U[0,5] m/s.
#include "../../include/dmpp_planner.h"
#include <cmath>
#include <cstring>

namespace {
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
    double range(double a, double b) { return a + (b - a) * uni(); }
};
const double kPi = 3.14159265358979323846;
double wrap360(double d) { while (d < 0) d += 360.0; while (d >= 360.0) d -= 360.0; return d; }
}  // namespace

extern "C" void pp_default_config(PlannerConfig* c, int grid_w, int grid_h)
{
    std::memset(c, 0, sizeof(*c));
    // values the reference uses but never defines: build-chosen
    c->ROAD_FARAIM_MAX = 40; c->ROAD_FARAIM_MIN = 10;
    c->PRE_INTER_FARAIM = 15; c->INTER_FARAIM = 10;
    c->ROAD_REMAIN_DISTANCE = 10; c->INTER_REMAIN_DISTANCE = 5;
    c->EPSILON = 1e-6; c->PI = kPi;
    c->Vehicle_Width = 1.8;
    c->NO_OBSTACLE_DIS = 999;                 // the value Planning.cpp:161-162 pre-loads
    c->wgs_lat0 = 23.0; c->wgs_lng0 = 113.0;  // local frame origin
    c->wgs_deg_per_m_lat = 1.0 / 111320.0;
    c->wgs_deg_per_m_lng = 1.0 / (111320.0 * std::cos(c->wgs_lat0 * kPi / 180.0));
    c->ID_MORE = 0;
    c->decision_stage = 1;
    c->lanechg_stage = 1;
    c->grid_stage = 1;
    c->grid_w = grid_w; c->grid_h = grid_h;
    c->max_expansions = grid_w * grid_h;      // never binds unless lowered
    c->bucket_cap = 4096;                     // live open-list entries (DMPP_OPEN_CAP of them in LDS, the rest in the spill area)
    c->max_path = 4 * (grid_w > grid_h ? grid_w : grid_h);
    c->n_lattice = 16; c->lookahead_cells = 120;
    c->dynamic_obstacles = 0; c->force_replan = 0;
    c->cell = 0.25;
    c->inflate = 0.5 * c->Vehicle_Width + 0.25;
    c->lattice_step = 0.3;                    // the sweep step of Decision.cpp:942
    c->d_safe = 1.0; c->w_col = 1.0; c->w_curv = 1.0; c->w_prog = 100.0; c->w_off = 0.1;
    c->dyn_dt = 0.1;
}

extern "C" void pp_init_state(SceneState* st, int lane_num)
{
    std::memset(st, 0, sizeof(*st));
    st->his_behavior = 1;                     // CPlanning::CPlanning(), Planning.cpp:10
    st->z_behavior = 1;
    st->z_target_lanenum = lane_num;
    st->d_his_behavior = 1;
    st->d_his_target_lanenum = lane_num;
}

// Layout of the pools this generator fills (fixed strides so scene s is self-contained):
//   lane_pool : scene s owns [s*3*PP_GEN_LANE_PTS, (s+1)*3*PP_GEN_LANE_PTS): current, left, right lane
//   lane_attr_pool : one lanechg_attribute byte per lane_pool point, same indexing
//   ref_pool  : scene s owns [s*PP_GEN_REF_PTS,   (s+1)*PP_GEN_REF_PTS)
//   obs_pool / mot_pool : scene s owns [s*n_obs, (s+1)*n_obs)
extern "C" int pp_gen_scenes(const PlannerConfig* c, int first_scene, int n_scenes, int n_obs, int junction_every,
                             SceneIn* in, GlobalPoint3D* lane_pool, uint8_t* lane_attr_pool, GlobalPoint2D* ref_pool,
                             ObPoint* obs_pool, ObMotion* mot_pool, SceneState* state)
{
    if (!c || !in || !lane_pool || !ref_pool || n_scenes < 0 || n_obs < 0) return -1;
    const double S = (double)c->grid_w * c->cell;         // world edge, metres
    const double Sy = (double)c->grid_h * c->cell;
    static const int attr_cycle[8] = { 0, 0, 0, 1, 2, 3, 0, 0 };
    for (int s = 0; s < n_scenes; s++) {
        const int scene = first_scene + s;
        SplitMix64 rng(0x5EED0000ull + (uint64_t)scene);
        SceneIn& si = in[s];
        std::memset(&si, 0, sizeof(si));
        // ego
        const double ex = rng.range(0.02 * S, 0.10 * S), ey = rng.range(0.10 * Sy, 0.90 * Sy);
        const double ehead = rng.range(-15.0, 15.0);
        const double speed = rng.range(0.0, 60.0);
        // lane: y(x) = y0 + A sin(2 pi (x - ex) / lambda); ego sits lat_off beside it
        const double A = rng.range(0.0, 3.0), lambda = rng.range(40.0, 120.0), lat_off = rng.range(-0.15, 0.15);
        const double y0 = ey - lat_off;
        const double lane_w = 3.75;
        const int lane_sum = 3;
        int lane_num = 2;
        if (scene % 5 == 0) lane_num = 1;                 // no left neighbour
        else if (scene % 7 == 0) lane_num = 3;            // no right neighbour
        GlobalPoint3D* cur = lane_pool + (size_t)s * 3 * PP_GEN_LANE_PTS;
        GlobalPoint3D* left = cur + PP_GEN_LANE_PTS;
        GlobalPoint3D* right = left + PP_GEN_LANE_PTS;
        const double xs = ex - 25.0;                      // 50 points (25 m) behind the ego
        for (int i = 0; i < PP_GEN_LANE_PTS; i++) {
            const double x = xs + 0.5 * i;
            const double ph = 2 * kPi * (x - ex) / lambda;
            const double y = y0 + A * std::sin(ph);
            const double slope = A * (2 * kPi / lambda) * std::cos(ph);
            const double dir = wrap360(std::atan(slope) * 180.0 / kPi);
            cur[i] = { x, y, dir };
            left[i] = { x, y + lane_w, dir };
            right[i] = { x, y - lane_w, dir };
        }
        si.lanes.cur_off = (int)(cur - lane_pool); si.lanes.cur_n = PP_GEN_LANE_PTS;
        si.lanes.left_off = (int)(left - lane_pool); si.lanes.left_n = (lane_num > 1) ? PP_GEN_LANE_PTS : 0;
        si.lanes.right_off = (int)(right - lane_pool); si.lanes.right_n = (lane_num < lane_sum) ? PP_GEN_LANE_PTS : 0;
        si.lanes.lane_sum = lane_sum;
        si.lanes.lanechg_attribute = attr_cycle[scene & 7];
        si.lanes.lane_width = lane_w;
        // location
        si.loc.globalpoint = { ex, ey, wrap360(ehead) };
        si.loc.velocity = speed;
        si.loc.pos = 0;
        if (junction_every > 0 && scene % junction_every == junction_every - 1) si.loc.pos = 1 + ((scene / junction_every) & 1);
        si.loc.road_num = 1; si.loc.lane_num = lane_num;
        si.loc.last_roadnum = 1; si.loc.next_roadnum = 2; si.loc.last_lanenum = lane_num; si.loc.next_lanenum = lane_num;
        si.loc.path_num = 0;
        for (int l = 0; l < DMPP_LANESUM; l++) si.loc.id[l] = 50;
        if (si.loc.pos == 2) si.loc.id[lane_num - 1] = 4;  // ego a few points into the junction polyline
        // decision input (used verbatim when the decision stage is off)
        si.dec.velocity_expect = 10; si.dec.behavior = 1; si.dec.target_roadnum = 1; si.dec.target_lanenum = lane_num;
        si.dec.light = 0; si.dec.behavior_to_dlg = 1;
        si.stub_attribute = scene % 4;                    // 0 straight, 1 left, 2 right, 3 u-turn (left light)
        // refpath pool slice: a gentle arc leaving the ego position (the junction polyline /
        // DecisionOut.refpath of the pre-junction and junction scenes)
        GlobalPoint2D* ref = ref_pool + (size_t)s * PP_GEN_REF_PTS;
        {
            const double turn = rng.range(-0.4, 0.4);     // degrees per point
            double hx = ex - 2.0, hy = ey, hd = ehead;
            for (int i = 0; i < PP_GEN_REF_PTS; i++) {
                ref[i] = { hx, hy };
                hx += 0.5 * std::cos(hd * kPi / 180.0); hy += 0.5 * std::sin(hd * kPi / 180.0);
                hd += turn;
            }
        }
        si.ref_off = (int)(ref - ref_pool); si.ref_n = PP_GEN_REF_PTS;
        si.dec.refpath_n = PP_GEN_REF_PTS;
        // obstacles
        si.obs_off = s * n_obs; si.obs_n = n_obs;
        const int near_lane = n_obs >= 8 ? n_obs / 16 + 1 : 0;
        for (int j = 0; j < n_obs; j++) {
            double ox, oy;
            for (;;) {
                if (j < near_lane) {
                    ox = ex + rng.range(5.0, 60.0);
                    oy = y0 + A * std::sin(2 * kPi * (ox - ex) / lambda) + rng.range(-2.5, 2.5);
                } else {
                    ox = rng.range(0.0, S); oy = rng.range(0.0, Sy);
                }
                const double dx = ox - ex, dy = oy - ey;
                if (dx * dx + dy * dy >= 9.0) break;
            }
            ObPoint& o = obs_pool[(size_t)si.obs_off + j];
            o.x = ox; o.y = oy; o.type = 0; o.radius = (float)rng.range(0.3, 1.5);
            const double vh = rng.range(0.0, 2 * kPi), vs = rng.range(0.0, 5.0);
            if (mot_pool) mot_pool[(size_t)si.obs_off + j] = { vs * std::cos(vh), vs * std::sin(vh) };
        }
        // grid engine
        si.grid_origin = { 0.0, 0.0 };
        si.goal = { rng.range(0.90 * S, 0.98 * S), rng.range(0.10 * Sy, 0.90 * Sy) };
        // lane-change inputs (drawn last so that everything above keeps its values): the exit lanes the
        // navigation allows, the decision period, and the per-point lane-change attribute of the map
        {
            static const uint16_t pat[8][3] = { {0, 0, 0}, {0, 0, 0}, {1, 0, 0}, {3, 0, 0}, {1, 2, 0}, {2, 3, 0}, {1, 2, 3}, {0, 0, 0} };
            const int p = (int)(rng.next() & 7);
            for (int k = 0; k < 3; k++) si.out_lane_no[k] = pat[p][k];
            if (p < 2) si.out_lane_no[0] = (uint16_t)lane_num;      // ego already in an exit lane
            si.period_last = 100.0;
            const int run = 20 + (int)(rng.next() % 240);           // points ahead of the ego that keep the attribute
            const int base = si.lanes.lanechg_attribute;
            const int ahead = (rng.next() & 1) ? base : (base ? 3 : 0);
            if (lane_attr_pool) {
                uint8_t* a = lane_attr_pool + (size_t)s * 3 * PP_GEN_LANE_PTS;
                for (int i = 0; i < 3 * PP_GEN_LANE_PTS; i++) a[i] = (uint8_t)base;
                for (int i = 51; i < PP_GEN_LANE_PTS; i++) a[i] = (uint8_t)(i <= 50 + run ? ahead : 0);
            }
        }
        if (state) pp_init_state(&state[s], lane_num);
    }
    return 0;
}
