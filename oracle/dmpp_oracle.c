/*
 * dmpp_oracle.c — CPU ORACLE, Part R (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED — see dmpp_oracle.h.  Scalar, sequential restatement of the
 * per-tick planning path of the reference.  Each function cites the reference lines
 * it follows; where the reference has undefined behaviour the fence chosen is
 * stated at the spot (and listed in DESIGN.md §3.3).  Compile with -ffp-contract=off:
 * the device code is built the same way so that +,-,*,/,sqrt agree bit for bit.
 */
#include "dmpp_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------------------ */
/* Planning.h:54 — "precondition fabs(a) > EPSILON" */
int orc_Sgn(double a) { return a > 0 ? 1 : -1; }

/* CShare::CalcDistance: Euclidean, same expression the reference inlines at
 * Planning.cpp:413-414 (pow(d,2) is an exact square). */
double orc_CalcDistance(GlobalPoint2D a, GlobalPoint2D b)
{
    double dx = a.x - b.x, dy = a.y - b.y;
    return sqrt(dx * dx + dy * dy);
}

/* Planning.cpp:686-709 */
double orc_GetLatDis(const PlannerConfig* c, GlobalPoint2D cur_pt, GlobalPoint2D pt, GlobalPoint2D pt_next)
{
    double lat_dis = 0;
    if (fabs(pt.x - pt_next.x) > c->EPSILON) {
        double k = (pt.y - pt_next.y) / (pt.x - pt_next.x);
        lat_dis = fabs((cur_pt.y - pt.y) - k * (cur_pt.x - pt.x)) / sqrt(1 + k * k);
    } else {
        lat_dis = fabs(pt.x - cur_pt.x);
    }
    if (lat_dis < c->EPSILON) {
        lat_dis = 0;
    } else {
        lat_dis = lat_dis * orc_Sgn((pt_next.x - pt.x) * (cur_pt.y - pt.y) - (pt_next.y - pt.y) * (cur_pt.x - pt.x));
    }
    return lat_dis;
}

/* Planning.cpp:719-750 — degrees, CCW from east, [0,360) */
double orc_GetRoadAngle(const PlannerConfig* c, GlobalPoint2D apoint, GlobalPoint2D bpoint)
{
    double angle;
    const double PI = c->PI, EPSILON = c->EPSILON;
    if (fabs(bpoint.x - apoint.x) < EPSILON && fabs(bpoint.y - apoint.y) < EPSILON)
        angle = 0;
    else if (fabs(bpoint.x - apoint.x) < EPSILON) {
        if (bpoint.y > apoint.y) angle = PI / 2;
        else angle = 3 * PI / 2;
    } else {
        angle = atan((bpoint.y - apoint.y) / (bpoint.x - apoint.x));
        if (bpoint.x < apoint.x) angle = angle + PI;
        else if ((bpoint.x > apoint.x) && (bpoint.y < apoint.y)) angle = angle + 2 * PI;
    }
    angle = angle * 180 / PI;
    return angle;
}

/* Planning.cpp:760-786 */
double orc_GetAngleErr(double dir1, double dir2)
{
    double angle_err = dir2 - dir1;
    if (dir1 < 180) {
        if (dir2 - dir1 <= 180) angle_err = dir2 - dir1;
        else angle_err = dir2 - dir1 - 360;
    } else if (dir1 >= 180) {
        if (dir2 - dir1 > -180) angle_err = dir2 - dir1;
        else angle_err = dir2 - dir1 + 360;
    }
    return angle_err;
}

/* Planning.cpp:242-290.  faraim/nearaim are FLOAT (Planning.h:20-21): the double
 * expression is rounded to float on assignment, the clamps compare the float. */
void orc_Calculate_aim_dis(const PlannerConfig* c, const LocationOut* loc, float* faraim_dis, float* nearaim_dis)
{
    *faraim_dis = 0;
    *nearaim_dis = 0;
    switch (loc->pos) {
    case 0:
        *faraim_dis = (float)((loc->velocity / 3.6) * 5 + 4);
        if (*faraim_dis > c->ROAD_FARAIM_MAX) *faraim_dis = (float)c->ROAD_FARAIM_MAX;
        else if (*faraim_dis < c->ROAD_FARAIM_MIN) *faraim_dis = (float)c->ROAD_FARAIM_MIN;
        *nearaim_dis = *faraim_dis;
        break;
    case 1:
        *faraim_dis = (float)c->PRE_INTER_FARAIM;
        *nearaim_dis = *faraim_dis;
        break;
    case 2:
        *faraim_dis = (float)c->INTER_FARAIM;
        *nearaim_dis = *faraim_dis;
        break;
    default:
        break;
    }
}

static void aim_set(AimPoint* a, GlobalPoint3D p, int id) { a->Aim_point = p; a->Aim_id = id; }

/* clamp used only where the reference indexes past an array (fences) */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Planning.cpp:303-583.  planning_MapData[road][lane] -> LaneView slices of lane_pool;
 * faraim/nearaim are the members written by Calculate_aim_dis (Planning.cpp:118). */
void orc_SearchAimPoint(const PlannerConfig* c, const SceneIn* in, const DecisionOutPod* dec, const GlobalPoint2D* refpath,
                        const GlobalPoint3D* lane_pool, SceneState* st)
{
    const LocationOut* loc = &in->loc;
    const GlobalPoint3D* cur = lane_pool + in->lanes.cur_off;
    const GlobalPoint3D* left = lane_pool + in->lanes.left_off;
    const GlobalPoint3D* right = lane_pool + in->lanes.right_off;
    double sum_dis = 0;
    int LaneNum_Cur = loc->lane_num, LaneSum = in->lanes.lane_sum;
    int lane_i = clampi(LaneNum_Cur - 1, 0, DMPP_LANESUM - 1);
    int curpoint_id = loc->id[lane_i];                            /* :355 */
    int curpoint_sum = in->lanes.cur_n;                           /* :356 */
    int leftpoint_id, leftpoint_sum, rightpoint_id, rightpoint_sum;
    int plan_divflag;
    float faraim_dis = st->faraim_dis, nearaim_dis = st->nearaim_dis;
    AimPoint* aimpoint_far = &st->aimpoint_far;
    AimPoint* aimpoint_near = &st->aimpoint_near;

    if (LaneNum_Cur > 1) {                                        /* :359-368 */
        leftpoint_id = loc->id[clampi(LaneNum_Cur - 2, 0, DMPP_LANESUM - 1)];
        leftpoint_sum = in->lanes.left_n;
    } else { leftpoint_id = 0; leftpoint_sum = 0; }
    if (LaneNum_Cur < LaneSum) {                                  /* :371-380 */
        rightpoint_id = loc->id[clampi(LaneNum_Cur, 0, DMPP_LANESUM - 1)];
        rightpoint_sum = in->lanes.right_n;
    } else { rightpoint_id = 0; rightpoint_sum = 0; }

    plan_divflag = (fabs(faraim_dis - nearaim_dis) < 1) ? 0 : 1;  /* :386-393 */

    switch (loc->pos) {
    case 0:
        if (dec->target_lanenum == loc->lane_num) {               /* :401 */
            if (dec->behavior == 1 && !plan_divflag) {            /* :404-407 */
                for (int i = curpoint_id; i < curpoint_sum - 1; i++) {         /* :410 */
                    if (i < 0) continue;                          /* fence: negative id */
                    double dx = cur[i + 1].x - cur[i].x, dy = cur[i + 1].y - cur[i].y;
                    sum_dis += sqrt(dx * dx + dy * dy);           /* :413-414 */
                    if (sum_dis - 4 > faraim_dis) {               /* :416 */
                        aim_set(aimpoint_far, cur[i], i);
                        break;
                    } else {
                        aim_set(aimpoint_far, cur[curpoint_sum - 1], curpoint_sum - 1);  /* :426-429 */
                    }
                }
                if (dec->behavior == 1 && !plan_divflag) *aimpoint_near = *aimpoint_far;  /* :434 */
            }
        } else {                                                  /* :440 */
            if (dec->behavior == 2) {                             /* :443 */
                if (!plan_divflag) {
                    for (int i = leftpoint_id; i < leftpoint_sum - 1; i++) {   /* :448 */
                        if (i < 0) continue;
                        double dx = left[i + 1].x - left[i].x, dy = left[i + 1].y - left[i].y;
                        sum_dis += sqrt(dx * dx + dy * dy);
                        if (sum_dis - 4 > faraim_dis) {
                            aim_set(aimpoint_far, left[i], i);
                            break;
                        } else {
                            /* quirk :464-467: the default reads the CURRENT lane at leftpoint_sum-2,
                             * Aim_id = leftpoint_sum-1.  Fence: index clamped into the current lane. */
                            int q = clampi(leftpoint_sum - 2, 0, curpoint_sum > 0 ? curpoint_sum - 1 : 0);
                            aim_set(aimpoint_far, cur[q], leftpoint_sum - 1);
                        }
                    }
                }
            } else if (dec->behavior == 3) {                      /* :473 */
                if (!plan_divflag) {
                    /* quirk :478: loop bound is leftpoint_sum.  Fence: i+1 kept inside the right lane. */
                    for (int i = rightpoint_id; i < leftpoint_sum - 1; i++) {
                        if (i < 0 || i + 1 >= rightpoint_sum) break;
                        double dx = right[i + 1].x - right[i].x, dy = right[i + 1].y - right[i].y;
                        sum_dis += sqrt(dx * dx + dy * dy);
                        if (sum_dis - 4 > faraim_dis) {
                            aim_set(aimpoint_far, right[i], i);
                            break;
                        } else {
                            aim_set(aimpoint_far, right[rightpoint_sum - 1], rightpoint_sum - 1);  /* :494-497 */
                        }
                    }
                }
            }
        }
        break;
    case 1:                                                       /* :505-540 */
    case 2: {                                                     /* :541-578, same body */
        int n = dec->refpath_n;
        sum_dis = 0;
        /* fence: refpath.size()-1 is unsigned in the reference; fewer than 3 points would index out of range */
        if (n >= 3) {
            for (int i = 0; i < n - 1; i++) {
                sum_dis += orc_CalcDistance(refpath[i], refpath[i + 1]);      /* :509 */
                if ((sum_dis - 4) > faraim_dis) {                 /* :512 */
                    aimpoint_far->Aim_point.x = refpath[i].x;
                    aimpoint_far->Aim_point.y = refpath[i].y;
                    if (i < n - 4) {                              /* :517 */
                        aimpoint_far->Aim_point.dir = orc_GetRoadAngle(c, refpath[i], refpath[i + 2]);
                    } else {
                        int im2 = i - 2 < 0 ? 0 : i - 2;          /* fence: i-2 < 0 */
                        aimpoint_far->Aim_point.dir = orc_GetRoadAngle(c, refpath[im2], refpath[i]);
                    }
                    aimpoint_far->Aim_id = i;
                    break;
                } else {                                          /* :529-536 */
                    aimpoint_far->Aim_point.x = refpath[n - 1].x;
                    aimpoint_far->Aim_point.y = refpath[n - 1].y;
                    aimpoint_far->Aim_point.dir = orc_GetRoadAngle(c, refpath[n - 3], refpath[n - 1]);
                    aimpoint_far->Aim_id = n - 1;
                }
            }
        }
        *aimpoint_near = *aimpoint_far;                           /* :539,577 */
        break;
    }
    default:
        break;
    }
}

/* Planning.cpp:623-676.  mindist_id is a reference to the member path_near_id: when no
 * point is closer than 9999 it keeps its previous value (zero on the first tick). */
void orc_GetVhclLocalState(const PlannerConfig* c, const LocationOut* loc, const GlobalPoint2D last_Bpoints[DMPP_PATH_POINTS],
                           double* mindist_lat, double* path_dir_err, int* mindist_id, int* front_mindist_id, double* remain_dis)
{
    GlobalPoint2D vhcl_pose, pt, pt_next;
    double dist;
    *remain_dis = 0;
    *mindist_lat = 9999;
    vhcl_pose.x = loc->globalpoint.x;
    vhcl_pose.y = loc->globalpoint.y;
    *mindist_id = clampi(*mindist_id, 0, DMPP_PATH_POINTS - 1);   /* fence for a garbage carried id */
    for (unsigned i = 0; i < 200; i++) {                          /* :640-650 */
        double dx = vhcl_pose.x - last_Bpoints[i].x, dy = vhcl_pose.y - last_Bpoints[i].y;
        dist = sqrt(dx * dx + dy * dy);
        if (dist < *mindist_lat) { *mindist_lat = dist; *mindist_id = (int)i; }
        *front_mindist_id = *mindist_id + 8;                      /* :649 */
    }
    int index = *mindist_id;                                      /* :654-663 */
    if (*mindist_id == 199) index = *mindist_id - 1;
    pt = last_Bpoints[index];
    pt_next = last_Bpoints[index + 1];
    *mindist_lat = orc_GetLatDis(c, vhcl_pose, pt, pt_next);      /* :666 */
    for (int i = *front_mindist_id; i < 199; i++) {               /* :668-671 */
        double dx = last_Bpoints[i + 1].x - last_Bpoints[i].x, dy = last_Bpoints[i + 1].y - last_Bpoints[i].y;
        *remain_dis += sqrt(dx * dx + dy * dy);
    }
    double pt_dir = orc_GetRoadAngle(c, pt, pt_next);             /* :673 */
    *path_dir_err = orc_GetAngleErr(pt_dir, loc->globalpoint.dir);/* :675 */
}

/* Planning.cpp:797-832 — reads the MEMBERS path_lat_dis/path_dir_err/remain_dis (:810,815,821) */
int orc_UpdatePlanJudge(const PlannerConfig* c, const DecisionOutPod* dec, const LocationOut* loc, int last_behavior,
                        const SceneState* st, int* afreshcause)
{
    *afreshcause = 0;
    if (last_behavior != dec->behavior) { *afreshcause = 1; return 1; }
    if (fabs(st->path_lat_dis) > 0.2) { *afreshcause = 2; return 1; }
    if (fabs(st->path_dir_err) > 45) { *afreshcause = 3; return 1; }
    if ((loc->pos == 0) && st->remain_dis < c->ROAD_REMAIN_DISTANCE) { *afreshcause = 4; return 1; }
    else if (loc->pos != 0 && st->remain_dis < c->INTER_REMAIN_DISTANCE) { *afreshcause = 4; return 1; }
    return 0;
}

/* Planning.cpp:888-990 — the three cases are identical; other pos values leave the outputs alone */
void orc_SpeedPlanning(int ob_flag, const DecisionOutPod* dec, const LocationOut* loc, double mindist_lon, double mindist_lat,
                       float faraim_dis, double* brake_speed, int* acc_flag, double* des_acc)
{
    (void)mindist_lat;
    switch (loc->pos) {
    case 0: case 1: case 2:
        if (ob_flag) {
            if (mindist_lon - 4 > 9) {
                *brake_speed = 3 + (mindist_lon - 9) / (faraim_dis - 9) * (dec->velocity_expect - 3);   /* :898 */
                *acc_flag = 0; *des_acc = 0;
            } else if (mindist_lon - 4 > 5) {
                *brake_speed = 3; *acc_flag = 0; *des_acc = 0;
            } else {
                *brake_speed = 0; *acc_flag = 1; *des_acc = -3;
            }
        } else {
            *brake_speed = dec->velocity_expect; *acc_flag = 0; *des_acc = 0;
        }
        break;
    default:
        break;
    }
}

/* Planning.cpp:1000-1019.  Fence: ids clamped to [0,199] (front id reaches 207, :649).
 * sqrt(1-cosA*cosA) may be NaN; NaN < 0.001 is false, so radius = 0.5*dis3/NaN = NaN is
 * reproduced as the reference would compute it. */
double orc_CalculateRadius(const GlobalPoint2D last_Bpoints[DMPP_PATH_POINTS], int path_near_id, int path_front_near_id)
{
    double radius;
    int nid = clampi(path_near_id, 0, 199), fid = clampi(path_front_near_id, 0, 199);
    unsigned midlle_num = (unsigned)round((double)((nid + fid) / 2));     /* integer division first, :1003 */
    GlobalPoint2D a = last_Bpoints[nid], m = last_Bpoints[midlle_num], f = last_Bpoints[fid];
    double dis1 = sqrt((a.x - m.x) * (a.x - m.x) + (a.y - m.y) * (a.y - m.y));
    double dis2 = sqrt((m.x - f.x) * (m.x - f.x) + (m.y - f.y) * (m.y - f.y));
    double dis3 = sqrt((a.x - f.x) * (a.x - f.x) + (a.y - f.y) * (a.y - f.y));
    double dis = dis1 * dis1 + dis2 * dis2 - dis3 * dis3;
    double cosA = dis / (2 * dis1 * dis2);
    double sinA = sqrt(1 - cosA * cosA);
    if (sinA < 0.001) radius = 1000;
    else radius = 0.5 * dis3 / sinA;
    return radius;
}

/* ------------------------------------------------------------------------------ */
/* CShare helpers — bodies NOT in the reference; this is the repo's specification
 * (DESIGN.md §4).  Call sites constrain only signatures and caller-side invariants. */

/* BezierPlanning (calls: Planning.cpp:606,863): cubic Bezier, control arms |P3-P0|/3
 * along the start/end headings, uniform parameter. */
void orc_BezierPlanning(const PlannerConfig* c, GlobalPoint3D s, GlobalPoint3D e, GlobalPoint2D* out, int n)
{
    double th0 = s.dir * c->PI / 180, th1 = e.dir * c->PI / 180;
    double dx = e.x - s.x, dy = e.y - s.y;
    double d = sqrt(dx * dx + dy * dy) / 3;
    double x1 = s.x + d * cos(th0), y1 = s.y + d * sin(th0);
    double x2 = e.x - d * cos(th1), y2 = e.y - d * sin(th1);
    for (int i = 0; i < n; i++) {
        double t = (n > 1) ? (double)i / (double)(n - 1) : 0.0;
        double u = 1 - t;
        double b0 = u * u * u, b1 = 3 * u * u * t, b2 = 3 * u * t * t, b3 = t * t * t;
        out[i].x = b0 * s.x + b1 * x1 + b2 * x2 + b3 * e.x;
        out[i].y = b0 * s.y + b1 * y1 + b2 * y2 + b3 * e.y;
    }
}

/* MeanPoints (call: Planning.cpp:872, "make the points uniform"): arc-length resampling,
 * linear interpolation, cumulative length summed left to right. */
void orc_MeanPoints(const PlannerConfig* c, const GlobalPoint2D* in, int n_in, GlobalPoint2D* out, int n_out)
{
    if (n_in <= 0) { for (int k = 0; k < n_out; k++) { out[k].x = 0; out[k].y = 0; } return; }
    double* cum = (double*)malloc(sizeof(double) * (size_t)n_in);
    cum[0] = 0;
    for (int i = 1; i < n_in; i++) cum[i] = cum[i - 1] + orc_CalcDistance(in[i], in[i - 1]);
    double L = cum[n_in - 1];
    if (n_in == 1 || L < c->EPSILON) { for (int k = 0; k < n_out; k++) out[k] = in[0]; free(cum); return; }
    for (int k = 0; k < n_out; k++) {
        double s = (n_out > 1) ? (L * (double)k) / (double)(n_out - 1) : 0.0;
        int j = n_in - 2;
        for (int q = 0; q <= n_in - 2; q++) if (cum[q + 1] >= s) { j = q; break; }
        double seg = cum[j + 1] - cum[j];
        double r = (seg > c->EPSILON) ? (s - cum[j]) / seg : 0.0;
        out[k].x = in[j].x + r * (in[j + 1].x - in[j].x);
        out[k].y = in[j].y + r * (in[j + 1].y - in[j].y);
    }
    free(cum);
}

/* CreateNewPath (calls: Decision.cpp:629,631,667,669,942,961): lateral offset curve,
 * negative = left, positive = right; central-difference tangent, one-sided at the ends. */
int orc_CreateNewPath(const PlannerConfig* c, const GlobalPoint2D* path, int n, double offset, GlobalPoint2D* out)
{
    for (int i = 0; i < n; i++) {
        int ia = i > 0 ? i - 1 : 0, ib = i < n - 1 ? i + 1 : n - 1;
        double tx = path[ib].x - path[ia].x, ty = path[ib].y - path[ia].y;
        double L = sqrt(tx * tx + ty * ty);
        if (L < c->EPSILON) { out[i] = path[i]; }
        else {
            out[i].x = path[i].x + offset * (ty / L);
            out[i].y = path[i].y + offset * (-tx / L);
        }
    }
    return n;
}

/* SearchObstacle (11 call sites): nearest obstacle by arc length whose
 * signed lateral offset (left +) lies in [lat_lo, lat_hi].  Writes NO_OBSTACLE_DIS to
 * both distances when nothing qualifies (callers test dis_lng without the flag:
 * Decision.cpp:373,922,944). */
int orc_SearchObstacle(const PlannerConfig* c, const GlobalPoint2D* path, int n, const ObPoint* obs, int m,
                       double lat_lo, double lat_hi, double* dis_lat, double* dis_lng, ObPoint* ob, int* path_id)
{
    int found = 0, bj = 0, bid = 0;
    double best_lng = 0, best_lat = 0;
    if (n >= 2 && m >= 1) {
        double* s = (double*)malloc(sizeof(double) * (size_t)n);
        s[0] = 0;
        for (int i = 1; i < n; i++) s[i] = s[i - 1] + orc_CalcDistance(path[i], path[i - 1]);
        for (int j = 0; j < m; j++) {
            double ox = obs[j].x, oy = obs[j].y;
            double best = INFINITY; int bi = 0;
            for (int i = 0; i < n; i++) {
                double dx = ox - path[i].x, dy = oy - path[i].y;
                double d2 = dx * dx + dy * dy;
                if (d2 < best) { best = d2; bi = i; }
            }
            int idx = (bi == n - 1) ? n - 2 : bi;
            GlobalPoint2D a = path[idx], b = path[idx + 1], o = { ox, oy };
            if (bi == 0) {          /* behind the first point: not on this path */
                double t = (ox - a.x) * (b.x - a.x) + (oy - a.y) * (b.y - a.y);
                if (t < 0) continue;
            }
            if (bi == n - 1) {      /* beyond the last point */
                double t = (ox - b.x) * (b.x - a.x) + (oy - b.y) * (b.y - a.y);
                if (t > 0) continue;
            }
            double lat = orc_GetLatDis(c, o, a, b);
            if (lat < lat_lo || lat > lat_hi) continue;
            double lng = s[bi];
            if (!found || lng < best_lng) { found = 1; best_lng = lng; best_lat = lat; bj = j; bid = bi; }
        }
        free(s);
    }
    if (found) { *dis_lat = best_lat; *dis_lng = best_lng; *ob = obs[bj]; *path_id = bid; }
    else { *dis_lat = c->NO_OBSTACLE_DIS; *dis_lng = c->NO_OBSTACLE_DIS; memset(ob, 0, sizeof(*ob)); *path_id = 0; }
    return found;
}

/* GlobalToWGS84 (call: Planning.cpp:209): local equirectangular frame about (lat0,lng0). */
GPSPoint2D orc_GlobalToWGS84(const PlannerConfig* c, GlobalPoint2D p)
{
    GPSPoint2D g;
    g.lat = c->wgs_lat0 + p.y * c->wgs_deg_per_m_lat;
    g.lng = c->wgs_lng0 + p.x * c->wgs_deg_per_m_lng;
    return g;
}

/* ------------------------------------------------------------------------------ */
/* Decision side: LoadRefPath, AroundObstacle, BehaviorDecision - the no-lane-change
 * branch (the lateral sweep) and, further down, the lane-change rule tree with
 * Nav_LaneChange (Decision.cpp:685-738, 1011-1772) -, SpeedDecision, RefPath, and the two
 * junction handlers. */

typedef struct DecScratch {
    GlobalPoint2D F[DMPP_FRONT_POINTS], R[DMPP_REAR_POINTS], LF[DMPP_FRONT_POINTS], LR[DMPP_REAR_POINTS],
                  RF[DMPP_FRONT_POINTS], RR[DMPP_REAR_POINTS];
    int nF, nR, nLF, nLR, nRF, nRR;
    GlobalPoint2D refpath[DMPP_MAX_REFPATH];
    int n_ref;
} DecScratch;

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* forward / rearward slices, Decision.cpp:581-596 (and :611-622, :649-660 for the side lanes) */
static int load_front(const PlannerConfig* c, const GlobalPoint3D* lane, int IdSum, int Id, GlobalPoint2D* out)
{
    int n = 0;
    for (int i = imin(IdSum, Id + c->ID_MORE); i < imin(IdSum, Id + 120 + c->ID_MORE); i++) {
        if (i < 0) continue;
        out[n].x = lane[i].x; out[n].y = lane[i].y; n++;
    }
    return n;
}
static int load_rear(const PlannerConfig* c, const GlobalPoint3D* lane, int IdSum, int Id, GlobalPoint2D* out)
{
    int n = 0;
    for (int i = imin(IdSum, Id + c->ID_MORE); i > imax(0, Id + c->ID_MORE - 40); i--) {
        int q = i >= IdSum ? IdSum - 1 : i;       /* fence: the reference reads [IdSum] when Id+ID_MORE >= IdSum */
        if (q < 0) break;
        out[n].x = lane[q].x; out[n].y = lane[q].y; n++;
    }
    return n;
}

/* Decision.cpp:553-673 */
static void orc_LoadRefPath(const PlannerConfig* c, const SceneIn* in, const GlobalPoint3D* lane_pool, DecScratch* d,
                            double* Width_CurLane)
{
    const LocationOut* loc = &in->loc;
    int LaneNum_Cur = loc->lane_num, LaneSum = in->lanes.lane_sum, LaneChg = in->lanes.lanechg_attribute;
    int Id_CurLane = loc->id[clampi(LaneNum_Cur - 1, 0, DMPP_LANESUM - 1)];
    int IdSum_CurLane = in->lanes.cur_n;
    const GlobalPoint3D* cur = lane_pool + in->lanes.cur_off;
    d->nF = d->nR = d->nLF = d->nLR = d->nRF = d->nRR = 0;
    *Width_CurLane = in->lanes.lane_width;                                   /* :578 */
    d->nF = load_front(c, cur, IdSum_CurLane, Id_CurLane, d->F);              /* :581-587 */
    d->nR = load_rear(c, cur, IdSum_CurLane, Id_CurLane, d->R);               /* :590-596 */
    if (LaneChg == 1 || LaneChg == 3) {                                       /* :599 */
        if (LaneNum_Cur > 1) {
            int Id_LeftLane = loc->id[clampi(LaneNum_Cur - 2, 0, DMPP_LANESUM - 1)];
            int IdSum_LeftLane = in->lanes.left_n;
            if (Id_LeftLane > 0 && Id_LeftLane < IdSum_LeftLane) {            /* :606 */
                const GlobalPoint3D* left = lane_pool + in->lanes.left_off;
                d->nLF = load_front(c, left, IdSum_LeftLane, Id_LeftLane, d->LF);
                d->nLR = load_rear(c, left, IdSum_LeftLane, Id_LeftLane, d->LR);
            }
        } else {                                                              /* :626-632 */
            d->nLF = orc_CreateNewPath(c, d->F, d->nF, -1 * *Width_CurLane, d->LF);
            d->nLR = orc_CreateNewPath(c, d->R, d->nR, -1 * *Width_CurLane, d->LR);
        }
    }
    if (LaneChg == 2) {                                                       /* :636 (attribute 3 loads no right paths) */
        if (LaneNum_Cur < LaneSum) {
            int Id_RightLane = loc->id[clampi(LaneNum_Cur, 0, DMPP_LANESUM - 1)];
            int IdSum_RightLane = in->lanes.right_n;
            if (Id_RightLane > 0 && Id_RightLane < IdSum_RightLane) {         /* :644 */
                const GlobalPoint3D* right = lane_pool + in->lanes.right_off;
                d->nRF = load_front(c, right, IdSum_RightLane, Id_RightLane, d->RF);
                d->nRR = load_rear(c, right, IdSum_RightLane, Id_RightLane, d->RR);
            }
        } else {                                                              /* :664-670 */
            d->nRF = orc_CreateNewPath(c, d->F, d->nF, *Width_CurLane, d->RF);
            d->nRR = orc_CreateNewPath(c, d->R, d->nR, *Width_CurLane, d->RR);
        }
    }
}

/* Decision.cpp:759-881: outputs zeroed (:794-806), a corridor is searched only when its path is non-empty */
static void around_one(const PlannerConfig* c, const GlobalPoint2D* p, int n, const ObPoint* obs, int m,
                       double lo, double hi, Path_Obs* out)
{
    memset(out, 0, sizeof(*out));
    if (n != 0) {
        int id = 0;
        out->Obs_flag = orc_SearchObstacle(c, p, n, obs, m, lo, hi, &out->Ob_Pose.dis_lat, &out->Ob_Pose.dis_lng, &out->Ob_Attr, &id);
        out->Ob_Pathid = id;
    }
}
static void orc_AroundObstacle(const PlannerConfig* c, const DecScratch* d, const ObPoint* obs, int m, double W,
                               Path_Obs around[6])
{
    double hv = 0.5 * c->Vehicle_Width;
    around_one(c, d->F,  d->nF,  obs, m, -hv, hv, &around[0]);                 /* :811 */
    around_one(c, d->R,  d->nR,  obs, m, -hv, hv, &around[1]);                 /* :817 */
    around_one(c, d->LF, d->nLF, obs, m, -hv, 0.5 * W, &around[2]);            /* :823 */
    around_one(c, d->LR, d->nLR, obs, m, -hv, 0.5 * W, &around[3]);            /* :830 */
    around_one(c, d->RF, d->nRF, obs, m, -0.5 * W, hv, &around[4]);            /* :836 */
    around_one(c, d->RR, d->nRR, obs, m, -0.5 * W, hv, &around[5]);            /* :842 */
}

/* CalcNaviLaneChgTimes, Decision.cpp:498-538 (the AfxMessageBox diagnostics are UI and dropped; BYTE return) */
static unsigned orc_CalcNaviLaneChgTimes(unsigned lanenum_cur, const uint16_t* outlanenum, int lanechgdir)
{
    int times = 5, times_tmp = 0;
    lanenum_cur &= 0xffu;                                                      /* const BYTE lanenum_cur */
    if (lanechgdir == 1) {
        for (int i = 0; i < DMPP_LANESUM && outlanenum[i] != 0; i++) {
            times_tmp = (int)lanenum_cur - (int)outlanenum[i];
            if (times_tmp < times) times = times_tmp;
        }
    } else if (lanechgdir == 2) {
        for (int i = 0; i < DMPP_LANESUM && outlanenum[i] != 0; i++) {
            times_tmp = (int)outlanenum[i] - (int)lanenum_cur;
            if (times_tmp < times) times = times_tmp;
        }
    }
    return (unsigned)(times & 0xff);
}

/* Nav_LaneChange, Decision.cpp:685-738; the caller's initial values are those of Decision.cpp:234-235 */
static void orc_Nav_LaneChange(const SceneIn* in, unsigned* Navi_LaneChg, unsigned* Navi_LaneChg_times)
{
    unsigned LaneNum_Cur = (unsigned)in->loc.lane_num;
    const uint16_t* out = in->out_lane_no;
    *Navi_LaneChg = 4; *Navi_LaneChg_times = 0;
    for (unsigned i = 0; i < DMPP_LANESUM && out[i] != 0; i++)                 /* :696-703 */
        if (LaneNum_Cur == out[i]) { *Navi_LaneChg = 0; break; }
    unsigned out_lane_min = out[0], out_lane_max = 1;                          /* :706-715 */
    for (int i = 0; i < DMPP_LANESUM; i++) if (out[i] > out_lane_max) out_lane_max = out[i];
    if (*Navi_LaneChg == 4) {                                                  /* :719-737 */
        if (LaneNum_Cur < out_lane_min) {
            *Navi_LaneChg = 2;
            *Navi_LaneChg_times = orc_CalcNaviLaneChgTimes(LaneNum_Cur, out, 2);
        } else if (LaneNum_Cur > out_lane_max) {
            *Navi_LaneChg = 1;
            *Navi_LaneChg_times = orc_CalcNaviLaneChgTimes(LaneNum_Cur, out, 1);
        } else {
            *Navi_LaneChg = 0;
        }
    }
}

/* The "remaining lane-change length" loops of Decision.cpp:1178-1187 and the like:
 *   for (WORD i = Id_CurLane; i < IdSum_CurLane - 1 && PRED(map[i + 1].lanechg_attribute); i++) dis += |p[i] p[i+1]|
 * PRED as the compiler parses the reference's text (== binds tighter than &):
 *   pred 0: attr == 1                      (:1179)
 *   pred 1: attr & (0x01 == 0x01) = attr&1 (:1212,1459,1506) — and also attr & (0x02 == 0x02) (:1316,1344,1523)
 *   pred 2: attr & (0x01 != 0x01) = 0      (:1477)
 * A negative ego id starts at 0 (fence: the reference would wrap it to 65535 and skip the loop body on
 * out-of-range reads). */
static double orc_lanechg_remaining(const GlobalPoint3D* cur, const uint8_t* attr, int Id_CurLane, int IdSum_CurLane, int pred)
{
    double dis_chg1 = 0;
    for (int i = Id_CurLane < 0 ? 0 : Id_CurLane; i < IdSum_CurLane - 1; i++) {
        int a = attr ? attr[i + 1] : 0, ok;
        if (pred == 0) ok = (a == 1);
        else if (pred == 1) ok = (a & 1);
        else ok = (a & 0);
        if (!ok) break;
        GlobalPoint2D pt = { cur[i].x, cur[i].y }, pt1 = { cur[i + 1].x, cur[i + 1].y };
        dis_chg1 += orc_CalcDistance(pt, pt1);
    }
    return dis_chg1;
}

#define KEEP_LANE()   do { Cur->behavior = 1; Cur->target_lanenum = LaneNum_Cur; Cur->lanechg_status = 0; } while (0)
#define KEEP_LANE_NS() do { Cur->behavior = 1; Cur->target_lanenum = LaneNum_Cur; } while (0)   /* branches that leave lanechg_status alone */
#define CAP_LEFT(v)   do { st->leftlight_time += period; if (st->leftlight_time > 2000) st->leftlight_time = (v); } while (0)

/* Lane-change rule tree, Decision.cpp:1017-1772 (the map allows a lane change at the ego point). */
static void orc_LaneChangeTree(const SceneIn* in, const GlobalPoint3D* lane_pool, const uint8_t* attr_pool,
                               const Path_Obs* around, unsigned Navi_LaneChg, const Behavior_Dec* His,
                               SceneState* st, Behavior_Dec* Cur)
{
    int LaneNum_Cur = in->loc.lane_num;
    const int LaneChg_Map = in->lanes.lanechg_attribute, LaneSum = in->lanes.lane_sum;
    const int Id_CurLane = in->loc.id[clampi(LaneNum_Cur - 1, 0, DMPP_LANESUM - 1)];
    const int IdSum_CurLane = in->lanes.cur_n;
    const GlobalPoint3D* cur = lane_pool + in->lanes.cur_off;
    const uint8_t* attr = attr_pool ? attr_pool + in->lanes.cur_off : NULL;
    const uint16_t* out = in->out_lane_no;
    const double period = in->period_last;
    const double F = around[0].Ob_Pose.dis_lng, LF = around[2].Ob_Pose.dis_lng, LR = around[3].Ob_Pose.dis_lng;
    const double RF = around[4].Ob_Pose.dis_lng, RR = around[5].Ob_Pose.dis_lng;
    int z_light_status = Cur->light_status;       /* the member the tree reads at :1193 (= Cur.light_status, :287) */

    if (Cur->lanechg_status == 0) {                                            /* :1018 */
        if (Navi_LaneChg != 0) {                                               /* :1021 */
            if (Navi_LaneChg == 1) {
                if (LaneChg_Map == 1 || LaneChg_Map == 3) {                    /* :1027 */
                    Cur->behavior_to_dlg = 2;
                    if (Cur->light_status != 1) { Cur->light_status = 1; st->leftlight_time = 0; }
                    st->leftlight_time += period;                              /* :1035 */
                    if (LF > F + 10 || LF > 40) {                              /* :1038 */
                        if (LR > 15) {
                            if (st->leftlight_time > 2000) {                   /* :1044 */
                                Cur->behavior = 2; Cur->target_lanenum = LaneNum_Cur - 1; Cur->lanechg_status = 1;
                            } else KEEP_LANE();
                        } else KEEP_LANE();
                    } else KEEP_LANE();
                } else { KEEP_LANE(); Cur->behavior_to_dlg = 4; }              /* :1077-1083 */
            } else if (Navi_LaneChg == 2) {                                    /* :1086 */
                if (LaneChg_Map == 2 || LaneChg_Map == 3) {
                    Cur->behavior_to_dlg = 3;
                    if (Cur->light_status != 2) { Cur->lanechg_status = 2; st->rightlight_time = 0; }  /* :1092-1096 as written */
                    st->rightlight_time += period;
                    if (RF > F + 10 || RF > 40) {                              /* :1100 */
                        if (RR > 15) {
                            if (st->rightlight_time >= 2000) {                 /* :1106 */
                                Cur->behavior = 3; Cur->target_lanenum = LaneNum_Cur + 1; Cur->lanechg_status = 1;
                            } else KEEP_LANE();
                        } else KEEP_LANE();
                    } else KEEP_LANE();
                } else { KEEP_LANE(); Cur->behavior_to_dlg = 4; }              /* :1135-1141 */
            }
        } else {                                                               /* :1146 no navigation demand */
            if (F < (2 * 10 + 5)) {                                            /* :1149 */
                st->frontobs_time++;
                if (st->frontobs_time > 2) {
                    st->frontobs_time = 3;
                    if (LaneChg_Map == 1) {                                    /* :1157 */
                        if (LaneNum_Cur > 1) {
                            Cur->behavior_to_dlg = 5;
                            int no_back_flag = 1;                              /* :1165-1172 */
                            for (int i = 0; i < DMPP_LANESUM && out[i] != 0; i++) if (out[i] == LaneNum_Cur - 1) no_back_flag = 0;
                            int chg_condition_flag = 0;
                            if (no_back_flag) {
                                double dis_chg1 = orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 0);  /* :1179 */
                                if (dis_chg1 > 60) {
                                    chg_condition_flag = 1;
                                    if (z_light_status != 1) { z_light_status = 1; st->leftlight_time = 0; }   /* :1193-1197 */
                                    CAP_LEFT(2000);
                                } else z_light_status = 0;
                            } else {
                                double dis_chg1 = orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 1);  /* :1212 */
                                if (dis_chg1 > 15) {
                                    chg_condition_flag = 1;
                                    if (Cur->light_status != 1) { Cur->light_status = 1; st->leftlight_time = 0; }
                                    CAP_LEFT(2100);                            /* :1230-1233 */
                                } else Cur->light_status = 0;
                            }
                            if (chg_condition_flag) {                          /* :1242 */
                                if (LF > F + 10) {
                                    if (LR > 10) {
                                        if (st->leftlight_time > 1500) {
                                            st->frontobs_time = 0;
                                            Cur->behavior = 2; Cur->target_lanenum = LaneNum_Cur - 1; Cur->lanechg_status = 1;
                                        } else KEEP_LANE();
                                    } else KEEP_LANE();
                                } else KEEP_LANE();
                            } else KEEP_LANE();
                        } else KEEP_LANE();                                    /* :1287-1292 */
                    } else if (LaneChg_Map == 2) {                             /* :1296 */
                        if (LaneNum_Cur < LaneSum) {
                            Cur->behavior_to_dlg = 6;
                            int no_back_flag = 1;                              /* :1304-1309 (tests the LEFT neighbour, as written) */
                            for (int i = 0; i < DMPP_LANESUM && out[i] != 0; i++) if (out[i] == LaneNum_Cur - 1) no_back_flag = 0;
                            int chg_condition_flag = 0;
                            if (no_back_flag) {
                                double dis_chg1 = orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 1);  /* :1316 */
                                if (dis_chg1 > 50) {
                                    chg_condition_flag = 1;
                                    if (Cur->light_status != 2) { Cur->light_status = 2; st->leftlight_time = 0; }
                                    CAP_LEFT(2000);
                                }
                            } else {
                                double dis_chg1 = orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 1);  /* :1344 */
                                if (dis_chg1 > 10) {
                                    chg_condition_flag = 1;
                                    if (Cur->light_status != 1) { Cur->light_status = 1; st->leftlight_time = 0; }  /* :1356-1360 */
                                    CAP_LEFT(2000);
                                }
                            }
                            if (chg_condition_flag) {                          /* :1370 */
                                if (RF > F + 10) {
                                    if (RR > 10) {
                                        if (st->leftlight_time > 1500) {
                                            st->frontobs_time = 0;
                                            Cur->behavior = 3; Cur->target_lanenum = LaneNum_Cur + 1; Cur->lanechg_status = 1;
                                        } else KEEP_LANE();
                                    } else KEEP_LANE();
                                } else KEEP_LANE();
                            } else KEEP_LANE();
                        } else KEEP_LANE();                                    /* :1417-1422 */
                    } else if (LaneChg_Map == 3) {                             /* :1426 (z_behavior_to_dlg = 7 is overwritten at :312) */
                        int no_back_leftchg_flag = 1, no_back_rightchg_flag = 1;
                        for (int i = 0; i < DMPP_LANESUM && out[i] != 0; i++) if (out[i] == LaneNum_Cur - 1) no_back_leftchg_flag = 0;
                        for (int i = 0; i < DMPP_LANESUM && out[i] != 0; i++) if (out[i] == LaneNum_Cur + 1) no_back_rightchg_flag = 0;
                        int leftchg_condition_flag = 0;                        /* :1453-1495 */
                        if (LaneNum_Cur > 1) {
                            if (no_back_leftchg_flag) {
                                if (orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 1) > 50) leftchg_condition_flag = 1;
                            } else {
                                if (orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 2) > 10) leftchg_condition_flag = 1;
                            }
                        }
                        int rightchg_condition_flag = 0;                       /* :1500-1541 */
                        if (LaneNum_Cur < LaneSum) {
                            if (no_back_rightchg_flag) {
                                if (orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 1) > 50) rightchg_condition_flag = 1;
                            } else {
                                if (orc_lanechg_remaining(cur, attr, Id_CurLane, IdSum_CurLane, 1) > 10) rightchg_condition_flag = 1;
                            }
                        }
                        if (leftchg_condition_flag && !no_back_leftchg_flag) { /* :1545 */
                            if (Cur->lanechg_status != 1) { Cur->light_status = 1; st->leftlight_time = 0; }
                            CAP_LEFT(2000);
                            if (LF > F + 10) {
                                if (LR > 10) {
                                    if (st->leftlight_time > 2000) {
                                        st->frontobs_time = 0;
                                        Cur->behavior = 2; Cur->target_lanenum = LaneNum_Cur - 1; Cur->lanechg_status = 1;
                                    } else KEEP_LANE();
                                } else KEEP_LANE();
                            } else KEEP_LANE();
                        } else if (rightchg_condition_flag && !no_back_rightchg_flag) {   /* :1596 */
                            if (Cur->light_status != 2) { Cur->light_status = 2; st->leftlight_time = 0; }
                            CAP_LEFT(2000);
                            if (RF > F + 10) {                                 /* no else branch, :1609-1635 */
                                if (RR > 10) {
                                    if (st->leftlight_time > 2000) {
                                        st->frontobs_time = 0;
                                        Cur->behavior = 3; Cur->target_lanenum = LaneNum_Cur + 1; Cur->lanechg_status = 1;
                                    } else KEEP_LANE_NS();
                                } else KEEP_LANE_NS();
                            }
                        } else if (leftchg_condition_flag) {                   /* :1638 */
                            if (Cur->light_status != 1) { Cur->lanechg_status = 1; st->leftlight_time = 0; }   /* :1640-1644 as written */
                            CAP_LEFT(2000);
                            if (LF > F + 10) {
                                if (LR > 10) {
                                    if (st->leftlight_time > 2000) {
                                        st->frontobs_time = 0;
                                        Cur->behavior = 2; Cur->target_lanenum = LaneNum_Cur - 1; Cur->lanechg_status = 1;
                                    } else KEEP_LANE_NS();
                                } else KEEP_LANE_NS();
                            } else KEEP_LANE_NS();
                        } else if (rightchg_condition_flag) {                  /* :1688 */
                            if (Cur->light_status != 2) { Cur->light_status = 2; st->leftlight_time = 0; }
                            CAP_LEFT(2000);
                            if (RF > F + 10) {                                 /* no else branch, :1702-1729 */
                                if (RR > 10) {
                                    if (st->leftlight_time > 2000) {
                                        st->frontobs_time = 0;
                                        Cur->behavior = 3;
                                        Cur->target_lanenum = LaneNum_Cur = 1; /* :1712 as written */
                                        Cur->lanechg_status = 1;
                                    } else KEEP_LANE_NS();
                                } else KEEP_LANE_NS();
                            }
                        } else KEEP_LANE();                                    /* :1731-1736 */
                    }
                } else KEEP_LANE();                                            /* :1741-1746 */
            } else {                                                           /* :1749-1756 */
                st->frontobs_time = 0;
                Cur->behavior_to_dlg = 8;
                KEEP_LANE();
            }
        }
    } else if (Cur->lanechg_status == 1) {                                     /* :1760-1771 */
        Cur->behavior_to_dlg = 9;
        if (Cur->target_lanenum == LaneNum_Cur) { Cur->lanechg_status = 0; Cur->light_status = 0; }
        Cur->behavior = His->behavior;
        Cur->target_lanenum = His->target_lanenum;
        Cur->light_status = His->light_status;
    }
    (void)z_light_status;
}
#undef KEEP_LANE
#undef KEEP_LANE_NS
#undef CAP_LEFT

/* Decision.cpp:898-1010 (LaneChg_Map == 0) and :1012-1015 (counters reset otherwise) */
static void orc_BehaviorDecision(const PlannerConfig* c, const SceneIn* in, const GlobalPoint3D* lane_pool, const uint8_t* attr_pool,
                                 const DecScratch* d, const ObPoint* obs, int m,
                                 double Width_CurLane, const Path_Obs* around, unsigned Navi_LaneChg, const Behavior_Dec* His,
                                 SceneState* st, Behavior_Dec* Cur, int* sweep_side, int* sweep_index)
{
    const Path_Obs* Path_Obs_F = &around[0];
    int LaneNum_Cur = in->loc.lane_num;
    int LaneChg_Map = in->lanes.lanechg_attribute;
    *sweep_side = 0; *sweep_index = -1;
    if (LaneChg_Map == 0) {
        if (Path_Obs_F->Ob_Pose.dis_lng < 15) {                               /* :922 */
            st->no_obsaviod_time = 0;
            st->obsavoid_time++;
            if (st->obsavoid_time > 2) {                                      /* :936 */
                GlobalPoint2D newpath[DMPP_FRONT_POINTS];
                double dl, dg; ObPoint ob; int id;
                int left_flag = 0;
                /* BYTE i; compared as double against (W_lane - W_veh)/0.6, :940 */
                for (unsigned i = 0; (double)i < (Width_CurLane - c->Vehicle_Width) / 0.6 && i < DMPP_MAX_SWEEP; i++) {
                    int n = orc_CreateNewPath(c, d->F, d->nF, -0.3 * (double)i, newpath);
                    orc_SearchObstacle(c, newpath, n, obs, m, -0.5 * c->Vehicle_Width, 0.5 * c->Vehicle_Width, &dl, &dg, &ob, &id);
                    if (dg > 25) {                                            /* :944-952 */
                        Cur->behavior = 4; Cur->target_lanenum = LaneNum_Cur; Cur->light_status = 1;
                        Cur->obsavoid_status = 1; Cur->behavior_to_dlg = 11;
                        left_flag = 1; *sweep_side = -1; *sweep_index = (int)i;
                        break;
                    }
                }
                if (!left_flag) {                                             /* :957-973 */
                    for (unsigned i = 0; (double)i < (Width_CurLane - c->Vehicle_Width) / 0.6 && i < DMPP_MAX_SWEEP; i++) {
                        int n = orc_CreateNewPath(c, d->F, d->nF, 0.3 * (double)i, newpath);
                        orc_SearchObstacle(c, newpath, n, obs, m, -0.5 * c->Vehicle_Width, 0.5 * c->Vehicle_Width, &dl, &dg, &ob, &id);
                        if (dg > 25) {
                            Cur->behavior = 5; Cur->target_lanenum = LaneNum_Cur; Cur->light_status = 2;
                            Cur->obsavoid_status = 1; Cur->behavior_to_dlg = 12;
                            *sweep_side = 1; *sweep_index = (int)i;
                            break;
                        }
                    }
                }
            } else {                                                          /* :977-983 */
                Cur->behavior = 1; Cur->target_lanenum = LaneNum_Cur; Cur->light_status = 0; Cur->behavior_to_dlg = 1;
            }
        } else {                                                              /* :985-1009 */
            if (st->z_segment_obsavoid_status == 0) {
                Cur->behavior = 1; Cur->target_lanenum = LaneNum_Cur; Cur->light_status = 0; Cur->behavior_to_dlg = 1;
            } else {
                st->no_obsaviod_time++;
                if (st->no_obsaviod_time > 3) {
                    Cur->behavior = 1; Cur->target_lanenum = LaneNum_Cur; Cur->light_status = 0;
                    Cur->behavior_to_dlg = 1; Cur->obsavoid_status = 0;
                }
            }
            Cur->behavior_to_dlg = 1;                                         /* :1008 */
        }
    } else {
        st->no_obsaviod_time = 0;                                             /* :1014-1015 */
        st->obsavoid_time = 0;
        if (c->lanechg_stage) orc_LaneChangeTree(in, lane_pool, attr_pool, around, Navi_LaneChg, His, st, Cur);  /* :1017-1772 */
    }
}

/* Decision.cpp:216-315 restricted to the in-scope calls */
static void orc_SegmentDecision(const PlannerConfig* c, const SceneIn* in, const GlobalPoint3D* lane_pool, const uint8_t* attr_pool,
                                const ObPoint* obs, int m, SceneState* st, DecScratch* d, PlanOut* po)
{
    double Width_CurLane = 0;
    Behavior_Dec Cur, His;
    unsigned Navi_LaneChg = 4, Navi_LaneChg_times = 0;                         /* :234-235 */
    orc_Nav_LaneChange(in, &Navi_LaneChg, &Navi_LaneChg_times);                /* :268 */
    po->navi_lanechg = (int32_t)Navi_LaneChg; po->navi_lanechg_times = (int32_t)Navi_LaneChg_times;
    orc_LoadRefPath(c, in, lane_pool, d, &Width_CurLane);                      /* :271 */
    orc_AroundObstacle(c, d, obs, m, Width_CurLane, po->around);               /* :274 */
    Cur.behavior = st->z_behavior;                                             /* :286-291 */
    Cur.light_status = st->z_light_status;
    Cur.target_lanenum = st->z_target_lanenum;
    Cur.lanechg_status = st->z_segment_lanechg_status;
    Cur.obsavoid_status = st->z_segment_obsavoid_status;
    Cur.behavior_to_dlg = st->z_behavior_to_dlg;
    memset(&His, 0, sizeof(His));                                              /* :265, :293-295 */
    His.behavior = st->d_his_behavior; His.light_status = st->d_his_light_status; His.target_lanenum = st->d_his_target_lanenum;
    orc_BehaviorDecision(c, in, lane_pool, attr_pool, d, obs, m, Width_CurLane, po->around, Navi_LaneChg, &His, st, &Cur,
                         &po->sweep_side, &po->sweep_index);                   /* :298 */
    st->z_velocity_expect = (Cur.behavior == 4 || Cur.behavior == 5) ? 5 : 10; /* SpeedDecision :1781-1793 */
    /* RefPath :1801-1816 */
    const GlobalPoint2D* src = d->F; int n = d->nF;
    if (Cur.behavior == 2) { src = d->LF; n = d->nLF; }
    else if (Cur.behavior == 3) { src = d->RF; n = d->nRF; }
    memcpy(d->refpath, src, sizeof(GlobalPoint2D) * (size_t)n);
    d->n_ref = n;
    st->z_behavior = Cur.behavior;                                             /* :307-313 */
    st->z_light_status = Cur.light_status;
    st->z_target_lanenum = Cur.target_lanenum;
    st->z_segment_lanechg_status = Cur.lanechg_status;
    st->z_segment_obsavoid_status = Cur.obsavoid_status;
    st->z_behavior_to_dlg = Cur.behavior_to_dlg;
    st->z_target_roadnum = in->loc.road_num;
}

/* PreStubDecision Decision.cpp:323-402 (pos 1) and StubDecision :409-486 (pos 2).
 * The junction polyline decision_InterMapData[...] is the scene's refpath-pool slice. */
static void orc_StubDecisions(const PlannerConfig* c, const SceneIn* in, const GlobalPoint3D* lane_pool,
                              const GlobalPoint2D* ref_pool, const ObPoint* obs, int m, SceneState* st, DecScratch* d, PlanOut* po)
{
    const LocationOut* loc = &in->loc;
    const GlobalPoint3D* cur = lane_pool + in->lanes.cur_off;
    const GlobalPoint2D* inter = ref_pool + in->ref_off;
    int n = 0;
    if (loc->pos == 1) {
        int Id_CurLane = loc->id[clampi(loc->lane_num - 1, 0, DMPP_LANESUM - 1)];
        for (int i = imax(Id_CurLane, 0); i < in->lanes.cur_n && n < DMPP_MAX_REFPATH; i++) {   /* :352-358 */
            d->refpath[n].x = cur[i].x; d->refpath[n].y = cur[i].y; n++;
        }
        for (int j = 0; j < in->ref_n && n < DMPP_MAX_REFPATH; j++) d->refpath[n++] = inter[j]; /* :361-367 */
    } else {
        int Id_Inter = loc->id[clampi(loc->last_lanenum - 1, 0, DMPP_LANESUM - 1)];             /* :423 */
        for (int j = imax(Id_Inter, 0); j < in->ref_n && n < DMPP_MAX_REFPATH; j++) d->refpath[n++] = inter[j]; /* :438-444 */
        for (int i = 0; i < imin(60, in->lanes.cur_n) && n < DMPP_MAX_REFPATH; i++) {           /* :446-452 */
            d->refpath[n].x = cur[i].x; d->refpath[n].y = cur[i].y; n++;
        }
    }
    d->n_ref = n;
    memset(po->around, 0, sizeof(po->around));
    int id = 0;
    Path_Obs* F = &po->around[0];
    F->Obs_flag = orc_SearchObstacle(c, d->refpath, n, obs, m, -0.5 * c->Vehicle_Width, 0.5 * c->Vehicle_Width,
                                     &F->Ob_Pose.dis_lat, &F->Ob_Pose.dis_lng, &F->Ob_Attr, &id);   /* :370,455 */
    F->Ob_Pathid = id;
    if (F->Ob_Pose.dis_lng < 13) {                                             /* :373-382 */
        double v = F->Ob_Pose.dis_lng - 3;
        st->z_velocity_expect = v > 0 ? v : 0;
        st->z_behavior_to_dlg = 13;
    } else {
        st->z_velocity_expect = 10;
        st->z_behavior_to_dlg = 1;
    }
    st->z_light_status = (in->stub_attribute == 3) ? 1 : in->stub_attribute;   /* :385-392 */
    st->z_behavior = 1;                                                        /* :394-399 */
    st->z_target_roadnum = loc->road_num;
    st->z_target_lanenum = loc->lane_num;
    po->sweep_side = 0; po->sweep_index = -1;
}

/* ------------------------------------------------------------------------------ */
/* The tick: Decision.cpp:172-205 (decision stage) then Planning.cpp:114-223. */
/* DecisionOut.refpath of the calling thread's last tick (Decision.cpp:195), for the tests of the refpath hand-off */
static __thread GlobalPoint2D g_last_refpath[DMPP_MAX_REFPATH];
static __thread int g_last_refpath_n;
int orc_last_refpath(GlobalPoint2D* out, int cap)
{
    int n = g_last_refpath_n < cap ? g_last_refpath_n : cap;
    if (n > 0) memcpy(out, g_last_refpath, sizeof(GlobalPoint2D) * (size_t)n);
    return g_last_refpath_n;
}

void orc_plan_tick(const PlannerConfig* c, const SceneIn* in, const GlobalPoint3D* lane_pool, const uint8_t* attr_pool,
                   const GlobalPoint2D* ref_pool, const ObPoint* obs_pool, const ObMotion* mot_pool, SceneState* st, PlanOut* po, GridOut* go,
                   uint8_t* grid_scratch, int32_t* order, int order_cap, int32_t* path, int path_cap)
{
    const LocationOut* loc = &in->loc;
    int m = in->obs_n;
    ObPoint* obs = (ObPoint*)malloc(sizeof(ObPoint) * (size_t)(m > 0 ? m : 1));
    DecScratch* d = (DecScratch*)malloc(sizeof(DecScratch));
    DecisionOutPod dec;
    const GlobalPoint2D* refpath;

    memset(po, 0, sizeof(*po));
    orc_effective_obstacles(c, obs_pool + in->obs_off, mot_pool ? mot_pool + in->obs_off : NULL, m, st->tick, obs);

    /* ---- decision stage ---- */
    if (c->decision_stage) {
        switch (loc->pos) {                                                    /* Decision.cpp:172-185 */
        case 0: orc_SegmentDecision(c, in, lane_pool, attr_pool, obs, m, st, d, po); break;
        case 1: case 2: orc_StubDecisions(c, in, lane_pool, ref_pool, obs, m, st, d, po); break;
        default: d->n_ref = 0; break;
        }
        dec.velocity_expect = st->z_velocity_expect;                           /* Decision.cpp:187-196 */
        dec.behavior = st->z_behavior;
        dec.target_roadnum = st->z_target_roadnum;
        dec.target_lanenum = st->z_target_lanenum;
        dec.light = st->z_light_status;
        dec.behavior_to_dlg = st->z_behavior_to_dlg;
        dec.refpath_n = d->n_ref;
        st->d_his_behavior = st->z_behavior;                                   /* Decision.cpp:199-201 */
        st->d_his_light_status = st->z_light_status;
        st->d_his_target_lanenum = st->z_target_lanenum;
        refpath = d->refpath;
    } else {
        dec = in->dec;
        if (dec.refpath_n > in->ref_n) dec.refpath_n = in->ref_n;
        if (dec.refpath_n > DMPP_MAX_REFPATH) dec.refpath_n = DMPP_MAX_REFPATH;
        refpath = ref_pool + in->ref_off;
        po->sweep_side = 0; po->sweep_index = -1;
    }
    po->dec = dec;
    g_last_refpath_n = dec.refpath_n > 0 ? (dec.refpath_n < DMPP_MAX_REFPATH ? dec.refpath_n : DMPP_MAX_REFPATH) : 0;
    if (g_last_refpath_n) memcpy(g_last_refpath, refpath, sizeof(GlobalPoint2D) * (size_t)g_last_refpath_n);

    /* ---- planning tick, Planning.cpp:114-223 ---- */
    GlobalPoint2D road_points[DMPP_PATH_POINTS];
    memset(road_points, 0, sizeof(road_points));                               /* :115 */
    orc_Calculate_aim_dis(c, loc, &st->faraim_dis, &st->nearaim_dis);          /* :118 */
    orc_SearchAimPoint(c, in, &dec, refpath, lane_pool, st);                   /* :121 */
    if (st->count == 0) {                                                      /* :124-128 */
        orc_BezierPlanning(c, loc->globalpoint, st->aimpoint_far.Aim_point, road_points, 200);   /* InitialPlanning :596-611 */
        memcpy(st->last_Bpoints, road_points, sizeof(road_points));
    }
    orc_GetVhclLocalState(c, loc, st->last_Bpoints, &st->path_lat_dis, &st->path_dir_err, &st->path_near_id,
                          &st->path_front_near_id, &st->remain_dis);           /* :131 */
    st->afresh_planning = orc_UpdatePlanJudge(c, &dec, loc, st->his_behavior, st, &st->afresh_cause);   /* :134 */
    if (c->force_replan && !st->afresh_planning) { st->afresh_planning = 1; st->afresh_cause = 5; }     /* BASELINE configs[3] */
    if (st->afresh_planning) {                                                 /* PathPlanning :845-877 */
        memset(road_points, 0, sizeof(road_points));
        if (loc->pos == 0) {
            orc_BezierPlanning(c, loc->globalpoint, st->aimpoint_far.Aim_point, road_points, 200);      /* :863 */
        } else if (loc->pos == 1 || loc->pos == 2) {
            int na = st->aimpoint_far.Aim_id;                                  /* :867-872; fence: Ref_points[200] */
            if (na > 200) na = 200;
            if (na > dec.refpath_n) na = dec.refpath_n;
            if (na < 0) na = 0;
            orc_MeanPoints(c, refpath, na, road_points, 200);
        }
    } else {
        memcpy(road_points, st->last_Bpoints, sizeof(road_points));            /* :144-145 */
    }
    /* rem_path = road_points[path_near_id..199], :153-158 */
    int near_id = clampi(st->path_near_id, 0, 200);
    double mindist_lat = 999, mindist_lon = 999;                               /* :161-162 */
    ObPoint min_ob_pt; int mindist_pathid = 0;
    int ob_flag = orc_SearchObstacle(c, road_points + near_id, 200 - near_id, obs, m, (double)(float)(-1.1), (double)(float)(1.1),
                                     &mindist_lat, &mindist_lon, &min_ob_pt, &mindist_pathid);           /* :168 */
    orc_SpeedPlanning(ob_flag, &dec, loc, mindist_lon, mindist_lat, st->faraim_dis, &st->brakespeed, &st->acc_flag, &st->des_acc); /* :171 */

    po->show.afresh_cause = st->afresh_cause;                                  /* :174-183 */
    po->show.near_ob_dist = mindist_lon;
    po->show.planspeed = st->brakespeed;
    po->show.planacc = st->des_acc;
    po->show.trafficlight = dec.light;
    for (unsigned i = 0; i < 100; i++) po->show.path_points[i] = road_points[2 * i];

    po->result.cnt = st->count % 100;                                          /* :189-201 */
    po->result.APA = 0;
    po->result.brakedis = mindist_lon;
    po->result.brake_speed = 0;
    po->result.desaccVd = st->acc_flag;
    po->result.desacc = st->des_acc;
    po->result.desspd = st->brakespeed;
    po->result.desstr = 0;
    po->result.desstrVd = 0;
    po->result.light = dec.light;
    po->result.radius = orc_CalculateRadius(st->last_Bpoints, st->path_near_id, st->path_front_near_id);   /* :199, on the OLD path */
    po->result.road_type = 0;
    po->result.sstop = 1;
    for (unsigned i = 0; i < 100; i++) {                                       /* :205-212 */
        GPSPoint2D g = orc_GlobalToWGS84(c, road_points[2 * i]);
        po->result.pnts[i].x = g.lat;
        po->result.pnts[i].y = g.lng;
    }
    memcpy(po->road_points, road_points, sizeof(road_points));
    po->ob_dis_lat = mindist_lat; po->ob_dis_lng = mindist_lon; po->ob = min_ob_pt;
    po->ob_flag = ob_flag; po->ob_pathid = mindist_pathid;

    st->his_behavior = dec.behavior;                                           /* :216-217 */
    memcpy(st->last_Bpoints, road_points, sizeof(road_points));
    st->count = (st->count + 1) & 0xFF;                                        /* BYTE count, :219-223 */
    if (st->count % 100 == 1) st->count = 1;

    /* ---- grid engine (rows G1-G3) ---- */
    if (c->grid_stage && go) {
        memset(go, 0, sizeof(*go));
        uint8_t* grid = grid_scratch;
        int own = 0;
        if (!grid) { grid = (uint8_t*)malloc((size_t)c->grid_w * (size_t)c->grid_h); own = 1; }
        orc_rasterise(c, in->grid_origin, obs, m, grid);
        int sc = orc_cell_of(c, in->grid_origin, loc->globalpoint.x, loc->globalpoint.y);
        int gc = orc_cell_of(c, in->grid_origin, in->goal.x, in->goal.y);
        int32_t* p = path; int own_p = 0;
        if (!p) { p = (int32_t*)malloc(sizeof(int32_t) * (size_t)c->max_path); own_p = 1; path_cap = c->max_path; }
        orc_grid_search(c, grid, sc, gc, go, order, order_cap, p, path_cap);
        orc_grid_score(c, in, obs, m, p, go);
        if (own_p) free(p);
        if (own) free(grid);
    }
    st->tick += 1;
    free(d);
    free(obs);
}
