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
/* G2: jump-point A* (JPS with straight jumps of any length and diagonal jumps bounded to
 * DMPP_DIAG_JUMP cells per node).  Optimal on the 8-connected grid with costs 10 / 14 and corner
 * cutting allowed; it expands only jump points (and one plain node every DMPP_DIAG_JUMP cells of a long
 * diagonal) instead of every cell of the A* ellipse.
 *   grid    : blocked(x,y) = outside the grid or occupied; the start cell is always free.
 *   dirs    : d = 0..7 = E,NE,N,NW,W,SW,S,SE (even = straight, cost 10 per cell; odd = diagonal, 14).
 *   h       : octile, 10*max(|dx|,|dy|) + 4*min(|dx|,|dy|).
 *   jump(p,s), s straight: walk p+s, p+2s, ...; at each cell c: blocked -> none; c = goal -> c;
 *             c has a forced neighbour for travel s -> c, where forced means: a cell beside c
 *             (perpendicular to s) is blocked and the cell beside c+s on the same side is free.
 *   jump(p,s), s diagonal = (sx,sy): walk c = p+s, p+2s, ... for at most DMPP_DIAG_JUMP cells; at each c:
 *             blocked -> none; c = goal -> c; c has a forced neighbour for travel s -> c, where forced
 *             means: (c-(sx,0) blocked and c+(-sx,sy) free) or (c-(0,sy) blocked and c+(sx,-sy) free);
 *             jump(c,(sx,0)) or jump(c,(0,sy)) finds something -> c; after DMPP_DIAG_JUMP cells -> that cell
 *             (a plain node: the diagonal goes on from there when it is expanded).
 *   successors of a node p closed with arriving direction d (8 = start), evaluated for s = 0..7 in turn:
 *             start      : jump(p,s) for every s;
 *             d straight : s = d -> jump(p,s);  s = d+-1 (the two diagonals next to d) -> jump(p,s),
 *                          only if forced: the cell beside p on that side is blocked and p+s is free;
 *             d diagonal : s = d and s = d+-1 (its two straight components) -> jump(p,s);
 *                          s = d+-2 -> jump(p,s) only if forced: the cell p + (s-d)/2 is blocked and p+s free.
 *   open set: a list in push order; an entry is (f, cell, arriving direction, run length); g is
 *             recovered as f - h(cell).
 *   step    : let fmin be the smallest f in the open set.  Up to DMPP_JPS_BATCH (4) entries with
 *             f = fmin are taken, the most recently pushed first; an entry whose cell is already closed
 *             is dropped, the others are closed in that order (the k-th cell closed is expansion k) and
 *             form the batch.  Then the batch nodes, in that order, push their successors (s = 0..7).
 *             Every batch node has the minimal f, so its g is final: the search stays optimal while a
 *             64-lane wave expands four nodes at once.
 *   stop    : goal closed (FOUND, at once: the rest of the step is skipped) | open set empty (NO_PATH)
 *             | n_expanded == max_expansions (LIMIT, at once)
 *             | more than bucket_cap live entries (OVERFLOW; the device keeps DMPP_OPEN_CAP of them in LDS and spills the
 *               rest - the entries that would be popped last - to HBM: a storage detail, not a limit of the specification)
 *             | a successor with f >= DMPP_F_LIMIT (COST_RANGE; per push, tested before the capacity).
 *   path    : from the goal, each closed cell knows its arriving direction and run length; the cells
 *             of every run are written out, start..goal.
 *   n_rounds: number of pops whose f exceeds every f popped before (+1 for the first).
 */
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

typedef struct JGrid { const uint8_t* g; int W, H, start; } JGrid;
static int jblk(const JGrid* G, int x, int y)
{
    if (x < 0 || y < 0 || x >= G->W || y >= G->H) return 1;
    int c = y * G->W + x;
    return c != G->start && G->g[c] != 0;
}
/* straight jump from (x,y) in direction s (even); returns the run length k > 0 or 0 for none */
static int jump_straight(const JGrid* G, int x, int y, int s, int gx, int gy)
{
    const int dx = DX[s], dy = DY[s];
    for (int k = 1;; k++) {
        x += dx; y += dy;
        if (jblk(G, x, y)) return 0;
        if (x == gx && y == gy) return k;
        if (dx != 0) {
            if ((jblk(G, x, y + 1) && !jblk(G, x + dx, y + 1)) || (jblk(G, x, y - 1) && !jblk(G, x + dx, y - 1))) return k;
        } else {
            if ((jblk(G, x + 1, y) && !jblk(G, x + 1, y + dy)) || (jblk(G, x - 1, y) && !jblk(G, x - 1, y + dy))) return k;
        }
    }
}

/* diagonal jump from (x,y) in direction s (odd), at most DMPP_DIAG_JUMP cells; run length or 0 for none */
static int jump_diagonal(const JGrid* G, int x, int y, int s, int gx, int gy)
{
    const int dx = DX[s], dy = DY[s];
    const int sh = dx > 0 ? 0 : 4, sv = dy > 0 ? 2 : 6;          /* its straight components */
    for (int k = 1; k <= DMPP_DIAG_JUMP; k++) {
        x += dx; y += dy;
        if (jblk(G, x, y)) return 0;
        if (x == gx && y == gy) return k;
        if ((jblk(G, x - dx, y) && !jblk(G, x - dx, y + dy)) || (jblk(G, x, y - dy) && !jblk(G, x + dx, y - dy))) return k;
        if (jump_straight(G, x, y, sh, gx, gy) || jump_straight(G, x, y, sv, gx, gy)) return k;
    }
    return DMPP_DIAG_JUMP;
}

typedef struct OEnt { int f, cell, dir, run; } OEnt;

/* test aid: the largest number of live open-list entries of the last search on this thread (tests pick scenes that exceed what
 * the device keeps in LDS) */
static __thread int g_peak_open = 0;
int orc_last_peak_open(void) { return g_peak_open; }

void orc_grid_search(const PlannerConfig* c, const uint8_t* grid, int start_cell, int goal_cell,
                     GridOut* out, int32_t* order, int order_cap, int32_t* path, int path_cap)
{
    const int W = c->grid_w, H = c->grid_h, N = W * H;
    const int cap = c->bucket_cap;
    out->start_cell = start_cell; out->goal_cell = goal_cell;
    out->status = DMPP_G_NO_PATH; out->n_expanded = 0; out->n_pushed = 0; out->n_rounds = 0;
    out->path_len = 0; out->path_cost = 0; out->order_digest = 0;
    if (grid[goal_cell] && goal_cell != start_cell) { out->status = DMPP_G_GOAL_BLOCKED; return; }
    if (grid[goal_cell]) { out->status = DMPP_G_GOAL_BLOCKED; return; }

    JGrid G = { grid, W, H, start_cell };
    uint8_t* closed = (uint8_t*)calloc((size_t)N, 1);
    uint8_t* pdir = (uint8_t*)malloc((size_t)N);
    uint16_t* prun = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    OEnt* open = (OEnt*)malloc(sizeof(OEnt) * (size_t)(cap + 1));
    int n_open = 0;
    const int gx = goal_cell % W, gy = goal_cell / W;

    open[n_open++] = (OEnt){ hfun(start_cell % W, start_cell / W, gx, gy), start_cell, 8, 0 };
    out->n_pushed = 1;
    int status = -1, fmax = -1;
    g_peak_open = 1;
    while (status < 0) {
        if (n_open > g_peak_open) g_peak_open = n_open;
        if (n_open == 0) { status = DMPP_G_NO_PATH; break; }
        int fmin = open[0].f;
        for (int i = 1; i < n_open; i++) if (open[i].f < fmin) fmin = open[i].f;
        OEnt batch[DMPP_JPS_BATCH]; int nb = 0;
        for (int t = 0; t < DMPP_JPS_BATCH && status < 0; t++) {
            int bi = -1;
            for (int i = n_open - 1; i >= 0; i--) if (open[i].f == fmin) { bi = i; break; }    /* latest push first */
            if (bi < 0) break;
            OEnt e = open[bi];
            memmove(&open[bi], &open[bi + 1], sizeof(OEnt) * (size_t)(n_open - bi - 1));        /* keeps push order */
            n_open--;
            if (closed[e.cell]) continue;
            closed[e.cell] = 1; pdir[e.cell] = (uint8_t)e.dir; prun[e.cell] = (uint16_t)e.run;
            if (e.f > fmax) { fmax = e.f; out->n_rounds++; }
            int seq = out->n_expanded++;
            if (order && seq < order_cap) order[seq] = e.cell;
            out->order_digest += mix64(((uint64_t)(uint32_t)seq << 32) | (uint32_t)e.cell);
            batch[nb++] = e;
            if (e.cell == goal_cell) { status = DMPP_G_FOUND; out->path_cost = e.f; }
            else if (out->n_expanded >= c->max_expansions) status = DMPP_G_LIMIT;
        }
        if (status >= 0) break;
        for (int bn = 0; bn < nb && status < 0; bn++) {
            const OEnt e = batch[bn];
            const int x = e.cell % W, y = e.cell / W, d = e.dir;
            const int g = e.f - hfun(x, y, gx, gy);
            for (int s = 0; s < 8 && status < 0; s++) {
                int run = 0;                                   /* 0 = no successor in direction s */
                const int tx = x + DX[s], ty = y + DY[s];
                if (d == 8) {
                    if ((s & 1) == 0) run = jump_straight(&G, x, y, s, gx, gy);
                    else run = jump_diagonal(&G, x, y, s, gx, gy);
                } else if ((d & 1) == 0) {
                    if (s == d) run = jump_straight(&G, x, y, s, gx, gy);
                    else if (s == ((d + 1) & 7) || s == ((d + 7) & 7)) {
                        const int px = DX[s] - DX[d], py = DY[s] - DY[d];          /* the side the diagonal leans to */
                        if (jblk(&G, x + px, y + py) && !jblk(&G, tx, ty)) run = jump_diagonal(&G, x, y, s, gx, gy);
                    }
                } else {
                    if (s == ((d + 1) & 7) || s == ((d + 7) & 7)) run = jump_straight(&G, x, y, s, gx, gy);
                    else if (s == d) run = jump_diagonal(&G, x, y, s, gx, gy);
                    else if (s == ((d + 2) & 7) || s == ((d + 6) & 7)) {
                        const int px = (DX[s] - DX[d]) / 2, py = (DY[s] - DY[d]) / 2;
                        if (jblk(&G, x + px, y + py) && !jblk(&G, tx, ty)) run = jump_diagonal(&G, x, y, s, gx, gy);
                    }
                }
                if (!run) continue;
                const int nx = x + run * DX[s], ny = y + run * DY[s];
                const int fn = g + run * ((s & 1) ? 14 : 10) + hfun(nx, ny, gx, gy);
                if (fn >= DMPP_F_LIMIT) { status = DMPP_G_COST_RANGE; break; }     /* checked per push, before the capacity */
                if (n_open >= cap) { status = DMPP_G_OVERFLOW; break; }
                open[n_open++] = (OEnt){ fn, ny * W + nx, s, run };
                out->n_pushed++;
            }
        }
    }
    out->status = status;
    if (status == DMPP_G_FOUND) {
        int L = 1, cur = goal_cell;
        while (cur != start_cell) { L += prun[cur]; cur -= prun[cur] * (DY[pdir[cur]] * W + DX[pdir[cur]]); }
        int keep = L;
        const int lim = path_cap < c->max_path ? path_cap : c->max_path;
        if (L > lim) { keep = lim; out->status = DMPP_G_PATH_TRUNC; }
        out->path_len = keep;
        /* write the last `keep` cells, goal backwards */
        int k = 0; cur = goal_cell;
        while (k < keep) {
            if (path) path[keep - 1 - k] = cur;
            k++;
            if (cur == start_cell) break;
            const int step = DY[pdir[cur]] * W + DX[pdir[cur]];
            int run = prun[cur], c2 = cur;
            for (int r = 1; r < run && k < keep; r++) { c2 -= step; if (path) path[keep - 1 - k] = c2; k++; }
            cur -= run * step;
        }
    }
    free(open); free(prun); free(pdir); free(closed);
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
            double d2 = dx * dx + dy * dy;
            /* an obstacle farther than radius + half width + d_safe cannot produce a penalty: ignored */
            double thr = (double)obs[j].radius + 0.5 * c->Vehicle_Width + c->d_safe;
            if (d2 > thr * thr) continue;
            double v = sqrt(d2) - (double)obs[j].radius;
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
    if (nc == 0) memset(out->best_path, 0, sizeof(out->best_path));     /* nothing to choose from: zeros, best_candidate 0 */
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
