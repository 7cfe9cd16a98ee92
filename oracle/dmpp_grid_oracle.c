/*
 * dmpp_grid_oracle.c — CPU ORACLE, grid engine rows G1-G4 (test infrastructure, NOT product).
 *
 * PARITY UNPINNED versus the reference: the reference contains no grid, lattice or
 * search code at all (only the comment "6: A*" on the behaviour code, Decision.h:36,41).
 * This file IS the specification: sequential, scalar, written for
 * clarity.  The HIP kernels must reproduce its integer outputs bit for bit
 * (occupancy, expansion order, digest, counters, path) and its float scores to 1e-6
 * relative.
 */
#include "dmpp_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* G4: constant-velocity obstacles; the position used by EVERY stage of tick t is
 * p0 + v * (dyn_dt * t). */
void orc_effective_obstacles(const PlannerConfig* c, const ObPoint* obs, const ObMotion* mot, int m, int tick, ObPoint* out)
{
    for (int j = 0; j < m; j++) {
        out[j] = obs[j];
        if (c->dynamic_obstacles && mot) {
            double t = c->dyn_dt * (double)tick;
            out[j].x = obs[j].x + mot[j].vx * t;
            out[j].y = obs[j].y + mot[j].vy * t;
        }
    }
}

int orc_cell_of(const PlannerConfig* c, GlobalPoint2D origin, double x, double y)
{
    int ix = (int)floor((x - origin.x) / c->cell);
    int iy = (int)floor((y - origin.y) / c->cell);
    if (ix < 0) ix = 0; if (ix > c->grid_w - 1) ix = c->grid_w - 1;
    if (iy < 0) iy = 0; if (iy > c->grid_h - 1) iy = c->grid_h - 1;
    return iy * c->grid_w + ix;
}

static int blocked_by(const PlannerConfig* c, GlobalPoint2D origin, int ix, int iy, const ObPoint* o)
{
    double cx = origin.x + ((double)ix + 0.5) * c->cell;
    double cy = origin.y + ((double)iy + 0.5) * c->cell;
    double R = (double)o->radius + c->inflate;
    double dx = cx - o->x, dy = cy - o->y;
    return dx * dx + dy * dy <= R * R;
}

/* G1 definition: a cell is occupied (1) iff its centre lies within radius+inflate of
 * some obstacle centre. */
void orc_rasterise_bruteforce(const PlannerConfig* c, GlobalPoint2D origin, const ObPoint* obs, int m, uint8_t* grid)
{
    for (int iy = 0; iy < c->grid_h; iy++)
        for (int ix = 0; ix < c->grid_w; ix++) {
            uint8_t v = 0;
            for (int j = 0; j < m && !v; j++) v = (uint8_t)blocked_by(c, origin, ix, iy, &obs[j]);
            grid[(size_t)iy * c->grid_w + ix] = v;
        }
}

/* Same result through a conservative bounding box per obstacle. */
void orc_rasterise(const PlannerConfig* c, GlobalPoint2D origin, const ObPoint* obs, int m, uint8_t* grid)
{
    memset(grid, 0, (size_t)c->grid_w * (size_t)c->grid_h);
    for (int j = 0; j < m; j++) {
        double R = (double)obs[j].radius + c->inflate;
        int ix0 = (int)floor((obs[j].x - R - origin.x) / c->cell) - 1;
        int ix1 = (int)floor((obs[j].x + R - origin.x) / c->cell) + 1;
        int iy0 = (int)floor((obs[j].y - R - origin.y) / c->cell) - 1;
        int iy1 = (int)floor((obs[j].y + R - origin.y) / c->cell) + 1;
        if (ix0 < 0) ix0 = 0; if (iy0 < 0) iy0 = 0;
        if (ix1 > c->grid_w - 1) ix1 = c->grid_w - 1;
        if (iy1 > c->grid_h - 1) iy1 = c->grid_h - 1;
        for (int iy = iy0; iy <= iy1; iy++)
            for (int ix = ix0; ix <= ix1; ix++)
                if (blocked_by(c, origin, ix, iy, &obs[j])) grid[(size_t)iy * c->grid_w + ix] = 1;
    }
}

/* ------------------------------------------------------------------------------ */
/* G2: bucketed A* with LIFO levels, expanded in batches of 8 (Dial's buckets over f = g + h,
 * depth-first tie-breaking, batch-synchronous so that 8 nodes x 8 directions fill a 64-lane wave).
 *   moves   : 8-connected, direction d = 0..7 = E,NE,N,NW,W,SW,S,SE, cost 10 (even d) / 14 (odd d);
 *             a move is legal iff the target cell is inside the grid, not occupied, not closed.
 *   h       : octile, 10*max(|dx|,|dy|) + 4*min(|dx|,|dy|)  (consistent, so f never decreases
 *             and a successor's f is within [f, f+28]; all f are even -> ring of 16 levels).
 *   open set: one STACK per f level.  Entries are (cell, arriving direction).  A cell may be
 *             stacked several times.
 *   step    : take the top min(8, height) entries of the stack of the lowest non-empty f, from the
 *             top down; each one whose cell is not closed is closed (expansion k = the k-th cell
 *             closed) and joins the batch; the others are dropped.  Then every batch node, in the
 *             order it was closed, pushes its legal successors in direction order 0..7 (legality
 *             is tested after the whole batch has been closed).
 *   stop    : goal closed (FOUND, at once: later entries of the step are not looked at) | open set
 *             empty (NO_PATH) | n_expanded == max_expansions (LIMIT, at once) | a stack would hold
 *             more than bucket_cap entries (OVERFLOW).
 *   path    : follow the arriving directions back from the goal.
 */
#define ORC_BATCH 8
static const int DX[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
static const int DY[8] = { 0, 1, 1, 1, 0, -1, -1, -1 };

static uint64_t mix64(uint64_t v)
{
    uint64_t z = v + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static int hfun(int x, int y, int gx, int gy)
{
    int dx = abs(x - gx), dy = abs(y - gy);
    int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
    return 10 * mx + 4 * mn;
}

void orc_grid_search(const PlannerConfig* c, const uint8_t* grid, int start_cell, int goal_cell,
                     GridOut* out, int32_t* order, int order_cap, int32_t* path, int path_cap)
{
    const int W = c->grid_w, H = c->grid_h, N = W * H, cap = c->bucket_cap;
    out->start_cell = start_cell; out->goal_cell = goal_cell;
    out->status = DMPP_G_NO_PATH; out->n_expanded = 0; out->n_pushed = 0; out->n_rounds = 0;
    out->path_len = 0; out->path_cost = 0; out->order_digest = 0;
    if (grid[goal_cell]) { out->status = DMPP_G_GOAL_BLOCKED; return; }

    uint8_t* closed = (uint8_t*)malloc((size_t)N);
    uint8_t* parent = (uint8_t*)malloc((size_t)N);
    uint32_t* bucket = (uint32_t*)malloc(sizeof(uint32_t) * 16u * (size_t)cap);
    int tail[16];
    for (int i = 0; i < N; i++) closed[i] = grid[i] ? 1 : 0;
    closed[start_cell] = 0;                       /* the vehicle is where it is */
    memset(tail, 0, sizeof(tail));
    const int gx = goal_cell % W, gy = goal_cell / W;

    int fcur = hfun(start_cell % W, start_cell / W, gx, gy);
    bucket[(size_t)((fcur / 2) % 16) * cap + 0] = (uint32_t)start_cell | (8u << 24);
    tail[(fcur / 2) % 16] = 1;
    out->n_pushed = 1; out->n_rounds = 1;

    int status = -1;
    while (status < 0) {
        int b = (fcur / 2) % 16;
        if (tail[b] == 0) {
            int k;
            for (k = 1; k < 16; k++) { int bb = ((fcur / 2) + k) % 16; if (tail[bb] != 0) break; }
            if (k == 16) { status = DMPP_G_NO_PATH; break; }
            fcur += 2 * k; out->n_rounds++;
            continue;
        }
        /* pop phase */
        int take = tail[b] < ORC_BATCH ? tail[b] : ORC_BATCH;
        int batch[ORC_BATCH], nb = 0;
        for (int q = 0; q < take && status < 0; q++) {
            uint32_t e = bucket[(size_t)b * cap + (tail[b] - 1 - q)];
            int cell = (int)(e & 0xFFFFFFu), pd = (int)(e >> 24);
            if (closed[cell]) continue;
            closed[cell] = 1; parent[cell] = (uint8_t)pd;
            int seq = out->n_expanded++;
            if (order && seq < order_cap) order[seq] = cell;
            out->order_digest += mix64(((uint64_t)(uint32_t)seq << 32) | (uint32_t)cell);
            batch[nb++] = cell;
            if (cell == goal_cell) { status = DMPP_G_FOUND; out->path_cost = fcur; }
            else if (out->n_expanded >= c->max_expansions) status = DMPP_G_LIMIT;
        }
        tail[b] -= take;
        if (status >= 0) break;
        /* expand phase */
        for (int i = 0; i < nb && status < 0; i++) {
            int cell = batch[i];
            int x = cell % W, y = cell / W;
            int g = fcur - hfun(x, y, gx, gy);
            for (int d = 0; d < 8; d++) {
                int nx = x + DX[d], ny = y + DY[d];
                if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                int n = ny * W + nx;
                if (closed[n]) continue;
                int fn = g + ((d & 1) ? 14 : 10) + hfun(nx, ny, gx, gy);
                int bb = (fn / 2) % 16;
                if (tail[bb] >= cap) { status = DMPP_G_OVERFLOW; break; }
                bucket[(size_t)bb * cap + tail[bb]++] = (uint32_t)n | ((uint32_t)d << 24);
                out->n_pushed++;
            }
        }
    }
    out->status = status;
    if (status == DMPP_G_FOUND) {
        int L = 1, cur = goal_cell;
        while (cur != start_cell) { int pd = parent[cur]; cur -= DY[pd] * W + DX[pd]; L++; }
        int keep = L;
        if (L > path_cap || L > c->max_path) { keep = path_cap < c->max_path ? path_cap : c->max_path; out->status = DMPP_G_PATH_TRUNC; }
        out->path_len = keep;
        cur = goal_cell;
        for (int k = 0; k < keep; k++) {
            if (path) path[keep - 1 - k] = cur;
            if (cur != start_cell) { int pd = parent[cur]; cur -= DY[pd] * W + DX[pd]; }
        }
    }
    free(bucket); free(parent); free(closed);
}

/* ------------------------------------------------------------------------------ */
/* G3: candidate scoring.  Candidates 0..n_lattice-1 are cubic Beziers (the BezierPlanning
 * of Part R) from the ego pose to a terminal pose shifted sideways in steps of
 * lattice_step (the 0.3 m of Decision.cpp:942); candidate n_lattice is the grid path itself,
 * resampled to 200 points (MeanPoints).  cost = w_col*col + w_curv*curv + w_prog*prog +
 * w_off*|offset|; the minimum wins, ties to the lowest index.
 * Sums over the 200 points use a fixed 64-leaf tree (leaf l takes points l, l+64, l+128,
 * l+192 in that order; leaves are folded 32,16,8,4,2,1) so that a 64-lane wave reproduces
 * them bit for bit. */
static double tree_sum(const double* v, int lo, int hi)
{
    double part[64];
    for (int l = 0; l < 64; l++) {
        double acc = 0;
        for (int q = 0; q < 4; q++) { int i = l + 64 * q; if (i >= lo && i < hi) acc += v[i]; }
        part[l] = acc;
    }
    for (int s = 32; s >= 1; s >>= 1) for (int l = 0; l < s; l++) part[l] += part[l + s];
    return part[0];
}

static double radius3(GlobalPoint2D a, GlobalPoint2D m, GlobalPoint2D f)
{   /* the circumradius expression of Planning.cpp:1004-1017, NaN and 0/0 fenced to 1000 */
    double dis1 = sqrt((a.x - m.x) * (a.x - m.x) + (a.y - m.y) * (a.y - m.y));
    double dis2 = sqrt((m.x - f.x) * (m.x - f.x) + (m.y - f.y) * (m.y - f.y));
    double dis3 = sqrt((a.x - f.x) * (a.x - f.x) + (a.y - f.y) * (a.y - f.y));
    double den = 2 * dis1 * dis2;
    if (!(den > 0)) return 1000;
    double cosA = (dis1 * dis1 + dis2 * dis2 - dis3 * dis3) / den;
    double sinA = sqrt(1 - cosA * cosA);
    if (!(sinA >= 0.001)) return 1000;
    return 0.5 * dis3 / sinA;
}

static void score_one(const PlannerConfig* c, const GlobalPoint2D* p, const ObPoint* obs, int m,
                      double* col, double* curv, double* prog)
{
    double pen[DMPP_PATH_POINTS], k2[DMPP_PATH_POINTS];
    int first_hit = DMPP_PATH_POINTS;
    for (int i = 0; i < DMPP_PATH_POINTS; i++) {
        double clear = INFINITY;
        for (int j = 0; j < m; j++) {
            double dx = p[i].x - obs[j].x, dy = p[i].y - obs[j].y;
            double v = sqrt(dx * dx + dy * dy) - (double)obs[j].radius;
            if (v < clear) clear = v;
        }
        clear = clear - 0.5 * c->Vehicle_Width;
        if (clear <= 0) { pen[i] = 1000.0; if (i < first_hit) first_hit = i; }
        else if (clear < c->d_safe) { double q = (c->d_safe - clear) / c->d_safe; pen[i] = q * q; }
        else pen[i] = 0;
        if (i >= 1 && i <= DMPP_PATH_POINTS - 2) {
            double R = radius3(p[i - 1], p[i], p[i + 1]);
            double k = 1 / R; k2[i] = k * k;
        } else k2[i] = 0;
    }
    *col = tree_sum(pen, 0, DMPP_PATH_POINTS);
    *curv = tree_sum(k2, 1, DMPP_PATH_POINTS - 1);
    *prog = (first_hit == DMPP_PATH_POINTS) ? 0.0 : (double)(DMPP_PATH_POINTS - first_hit) / (double)DMPP_PATH_POINTS;
}

void orc_grid_score(const PlannerConfig* c, const SceneIn* in, const ObPoint* obs, int m, const int32_t* path, GridOut* out)
{
    const int W = c->grid_w;
    GlobalPoint3D ego = in->loc.globalpoint;
    GlobalPoint2D ego2 = { ego.x, ego.y };
    GlobalPoint2D T; double thT;
    int have_path = (out->status == DMPP_G_FOUND) && path && out->path_len >= 1;
    int a = 0;
    if (have_path) {
        a = out->path_len - 1 < c->lookahead_cells ? out->path_len - 1 : c->lookahead_cells;
        if (a > DMPP_PATH_POINTS - 1) a = DMPP_PATH_POINTS - 1;
        int a0 = a - 4 > 0 ? a - 4 : 0;
        T.x = in->grid_origin.x + ((double)(path[a] % W) + 0.5) * c->cell;
        T.y = in->grid_origin.y + ((double)(path[a] / W) + 0.5) * c->cell;
        if (a0 == a) thT = ego.dir;
        else {
            GlobalPoint2D P0 = { in->grid_origin.x + ((double)(path[a0] % W) + 0.5) * c->cell,
                                 in->grid_origin.y + ((double)(path[a0] / W) + 0.5) * c->cell };
            thT = orc_GetRoadAngle(c, P0, T);
        }
    } else {
        T = in->goal;
        thT = orc_GetRoadAngle(c, ego2, in->goal);
    }
    int nl = c->n_lattice; if (nl > DMPP_MAX_LATTICE - 1) nl = DMPP_MAX_LATTICE - 1;
    int nc = nl + (have_path ? 1 : 0);
    out->n_candidates = nc;
    double th = thT * c->PI / 180, cs = cos(th), sn = sin(th);
    double best = 0; int bi = 0;
    GlobalPoint2D cand[DMPP_PATH_POINTS];
    for (int k = 0; k < nc; k++) {
        double off = 0;
        if (k < nl) {
            off = (double)(k - (nl - 1) / 2) * c->lattice_step;
            GlobalPoint3D e = { T.x + off * sn, T.y + off * (-cs), thT };      /* right of the heading for off > 0 */
            orc_BezierPlanning(c, ego, e, cand, DMPP_PATH_POINTS);
        } else {
            GlobalPoint2D* pts = (GlobalPoint2D*)malloc(sizeof(GlobalPoint2D) * (size_t)(a + 1));
            for (int i = 0; i <= a; i++) {
                pts[i].x = in->grid_origin.x + ((double)(path[i] % W) + 0.5) * c->cell;
                pts[i].y = in->grid_origin.y + ((double)(path[i] / W) + 0.5) * c->cell;
            }
            orc_MeanPoints(c, pts, a + 1, cand, DMPP_PATH_POINTS);
            free(pts);
        }
        double col, curv, prog;
        score_one(c, cand, obs, m, &col, &curv, &prog);
        double cost = c->w_col * col + c->w_curv * curv + c->w_prog * prog + c->w_off * fabs(off);
        out->cand_col[k] = col; out->cand_curv[k] = curv; out->cand_prog[k] = prog; out->cand_cost[k] = cost;
        if (k == 0 || cost < best) { best = cost; bi = k; memcpy(out->best_path, cand, sizeof(cand)); }
    }
    out->best_candidate = bi;
}
