// example_stream.cpp — the streamed tick of the C-ABI from C++: a fleet of scenes gets a NEW snapshot (egos + obstacles)
// from the host on every tick and every tick's results come back, with no host wait in between (what the reference's threads
// do with their blackboard: Planning.cpp:95-112,186,214).  The same inputs through the synchronous calls on a second handle
// must give the same PlanningOut records.  Exit code 0 = ran on the GPU and the two agree.
#include "../../include/dmpp_planner.h"
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(expr) do { int rc__ = (expr); if (rc__) { std::fprintf(stderr, "%s: %s\n", #expr, pp_last_error()); return 2; } } while (0)

int main()
{
    const int n = 320, n_obs = 24, ticks = 16;
    constexpr int D = 6;                                           // ticks the host runs ahead (buffers in rotation)
    PlannerConfig cfg; pp_default_config(&cfg, 128, 128);
    PlannerCaps caps{}; caps.max_scenes = n; caps.max_obs_total = n * n_obs; caps.max_lane_pts_total = n * 3 * PP_GEN_LANE_PTS; caps.max_ref_pts_total = n * PP_GEN_REF_PTS;
    std::vector<SceneIn> in(n); std::vector<SceneState> st(n);
    std::vector<GlobalPoint3D> lanes((size_t)n * 3 * PP_GEN_LANE_PTS); std::vector<uint8_t> attr(lanes.size());
    std::vector<GlobalPoint2D> ref((size_t)n * PP_GEN_REF_PTS); std::vector<ObPoint> obs((size_t)n * n_obs); std::vector<ObMotion> mot(obs.size());
    CHECK(pp_gen_scenes(&cfg, 100, n, n_obs, 8, in.data(), lanes.data(), attr.data(), ref.data(), obs.data(), mot.data(), st.data()));

    pp_handle h = nullptr, h2 = nullptr;
    CHECK(pp_create(&cfg, 0, &caps, &h));
    CHECK(pp_create(&cfg, 0, &caps, &h2));
    for (pp_handle q : { h, h2 }) {
        CHECK(pp_set_scenes(q, n, in.data(), lanes.data(), attr.data(), (int)lanes.size(), ref.data(), (int)ref.size(), obs.data(), nullptr, (int)obs.size()));
        CHECK(pp_set_state(q, st.data(), n));
    }
    // rotating pinned buffers: inputs the host fills, outputs the library fills
    SceneIn* pin[D]; ObPoint* pob[D]; PlanningOut* res[D]; PlanningStatus* show[D]; GridOut* grid[D]; long long id[D];
    for (int k = 0; k < D; k++) {
        pin[k] = (SceneIn*)pp_host_alloc(sizeof(SceneIn) * n); pob[k] = (ObPoint*)pp_host_alloc(sizeof(ObPoint) * n * n_obs);
        res[k] = (PlanningOut*)pp_host_alloc(sizeof(PlanningOut) * n); show[k] = (PlanningStatus*)pp_host_alloc(sizeof(PlanningStatus) * n);
        grid[k] = (GridOut*)pp_host_alloc(sizeof(GridOut) * n);
        if (!pin[k] || !pob[k] || !res[k] || !show[k] || !grid[k]) { std::fprintf(stderr, "pp_host_alloc: %s\n", pp_last_error()); return 2; }
    }
    std::vector<PlanningOut> want((size_t)n), got((size_t)n);
    std::vector<PlanOut> plan2((size_t)n);
    int mismatches = 0, poisoned = 0;
    for (int t = 0; t < ticks + D; t++) {
        const int k = t % D;
        if (t >= D) {                                              // tick t - D is complete: its records may be published, its buffers reused
            CHECK(pp_wait_tick(h, id[k], &poisoned));
            if (t - D == ticks - 1) std::memcpy(got.data(), res[k], sizeof(PlanningOut) * n);
        }
        if (t >= ticks) continue;
        for (int s = 0; s < n; s++) {                              // the snapshot of this tick: egos creep forward, obstacles drift
            in[s].loc.globalpoint.x += 0.05; in[s].loc.velocity = 20.0 + (t % 5);
            for (int j = 0; j < n_obs; j++) obs[(size_t)s * n_obs + j].y += 0.02 * ((j & 1) ? 1 : -1);
        }
        std::memcpy(pin[k], in.data(), sizeof(SceneIn) * n); std::memcpy(pob[k], obs.data(), sizeof(ObPoint) * obs.size());
        CHECK(pp_update_async(h, n, pin[k], pob[k], nullptr, (int)obs.size()));
        CHECK(pp_plan_tick(h));
        CHECK(pp_fetch_published_async(h, res[k], show[k], grid[k], &id[k]));
        // the same tick through the synchronous calls
        CHECK(pp_set_scenes(h2, n, in.data(), lanes.data(), attr.data(), (int)lanes.size(), ref.data(), (int)ref.size(), obs.data(), nullptr, (int)obs.size()));
        CHECK(pp_plan_tick(h2));
        if (t == ticks - 1) { CHECK(pp_get_plan(h2, plan2.data(), n)); for (int s = 0; s < n; s++) want[(size_t)s] = plan2[(size_t)s].result; }
    }
    for (int s = 0; s < n; s++) if (std::memcmp(&want[(size_t)s], &got[(size_t)s], sizeof(PlanningOut)) != 0) mismatches++;
    std::printf("streamed %d ticks of %d scenes, %d in flight: last tick desspd[0] %.3f radius[0] %.2f; %d of %d records differ from the synchronous run\n",
                ticks, n, D, got[0].desspd, got[0].radius, mismatches, n);
    for (int k = 0; k < D; k++) { pp_host_free(pin[k]); pp_host_free(pob[k]); pp_host_free(res[k]); pp_host_free(show[k]); pp_host_free(grid[k]); }
    pp_destroy(h); pp_destroy(h2);
    std::printf(mismatches ? "example_stream FAILED\n" : "example_stream ok\n");
    return mismatches ? 1 : 0;
}
