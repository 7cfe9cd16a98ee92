/*
 * dmpp_oracle_batch.c — CPU ORACLE (test infrastructure, NOT product code): runs the
 * single-scene oracle tick over a batch on pthreads.  Used by tests and by bench.py's
 * cpu_baseline leg only.  PARITY UNPINNED — see dmpp_oracle.h.
 */
#include "dmpp_oracle.h"
#include <pthread.h>
#include <stdlib.h>

typedef struct Job {
    const PlannerConfig* c; int lo, hi;
    const SceneIn* in; const GlobalPoint3D* lane_pool; const uint8_t* attr_pool; const GlobalPoint2D* ref_pool;
    const ObPoint* obs_pool; const ObMotion* mot_pool;
    SceneState* st; PlanOut* po; GridOut* go; uint8_t* grids; int n_ticks;
} Job;

static void* worker(void* arg)
{
    Job* j = (Job*)arg;
    size_t cells = (size_t)j->c->grid_w * (size_t)j->c->grid_h;
    uint8_t* scratch = NULL;
    if (j->c->grid_stage && !j->grids) scratch = (uint8_t*)malloc(cells);
    for (int t = 0; t < j->n_ticks; t++)
    for (int s = j->lo; s < j->hi; s++) {
        uint8_t* g = j->grids ? j->grids + (size_t)s * cells : scratch;
        orc_plan_tick(j->c, &j->in[s], j->lane_pool, j->attr_pool, j->ref_pool, j->obs_pool, j->mot_pool, &j->st[s], &j->po[s],
                      j->go ? &j->go[s] : NULL, g, NULL, 0, NULL, 0);
    }
    free(scratch);
    return NULL;
}

/* n_ticks consecutive ticks of every scene: scenes are independent (all cross-tick state is the scene's own SceneState),
 * so each thread takes its block of scenes through all the ticks without meeting the others — the CPU's best case, and
 * what bench.py times as the cpu_baseline (thread start-up amortised over the ticks). */
void orc_plan_ticks_batch(const PlannerConfig* c, int n, const SceneIn* in, const GlobalPoint3D* lane_pool,
                          const uint8_t* attr_pool, const GlobalPoint2D* ref_pool, const ObPoint* obs_pool, const ObMotion* mot_pool,
                          SceneState* st, PlanOut* po, GridOut* go, uint8_t* grids, int n_threads, int n_ticks)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    Job* jobs = (Job*)malloc(sizeof(Job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        Job* j = &jobs[t];
        j->c = c; j->lo = (int)((long long)n * t / n_threads); j->hi = (int)((long long)n * (t + 1) / n_threads);
        j->in = in; j->lane_pool = lane_pool; j->attr_pool = attr_pool; j->ref_pool = ref_pool; j->obs_pool = obs_pool; j->mot_pool = mot_pool;
        j->st = st; j->po = po; j->go = go; j->grids = grids; j->n_ticks = n_ticks;
        if (n_threads == 1) worker(j); else pthread_create(&th[t], NULL, worker, j);
    }
    if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(jobs); free(th);
}

void orc_plan_tick_batch(const PlannerConfig* c, int n, const SceneIn* in, const GlobalPoint3D* lane_pool,
                         const uint8_t* attr_pool, const GlobalPoint2D* ref_pool, const ObPoint* obs_pool, const ObMotion* mot_pool,
                         SceneState* st, PlanOut* po, GridOut* go, uint8_t* grids, int n_threads)
{
    orc_plan_ticks_batch(c, n, in, lane_pool, attr_pool, ref_pool, obs_pool, mot_pool, st, po, go, grids, n_threads, 1);
}
