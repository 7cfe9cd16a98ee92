"""CPU: properties of the oracle's build-defined parts (the CShare helpers and the grid engine),
checked against independent brute-force formulations in numpy / pure Python."""
import heapq

import numpy as np
import pytest

DX = [1, 1, 0, -1, -1, -1, 0, 1]
DY = [0, 1, 1, 1, 0, -1, -1, -1]


@pytest.fixture(scope="module")
def cfg(dm):
    return dm.default_config(128)


def _wavy(dm, n, rng):
    p = np.zeros(n, dm.GlobalPoint2D)
    p["x"] = np.cumsum(rng.uniform(0.3, 0.7, n))
    p["y"] = 2.5 * np.sin(p["x"] / 7.0)
    return p


def test_search_obstacle_matches_bruteforce_definition(dm, oracle, cfg):
    rng = np.random.default_rng(7)
    eps = cfg["EPSILON"][0]
    for _ in range(200):
        n, m = int(rng.integers(2, 150)), int(rng.integers(0, 40))
        p = _wavy(dm, n, rng)
        o = np.zeros(m, dm.ObPoint)
        o["x"], o["y"] = rng.uniform(-5, p["x"][-1] + 5, m), rng.uniform(-6, 6, m)
        lo, hi = -float(rng.uniform(0.5, 2)), float(rng.uniform(0.5, 2))
        r = oracle.SearchObstacle(cfg, p, o, lo, hi)
        s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(p["x"]), np.diff(p["y"])))])
        best = None
        for j in range(m):
            d2 = (o["x"][j] - p["x"]) ** 2 + (o["y"][j] - p["y"]) ** 2
            bi = int(np.argmin(d2))
            idx = n - 2 if bi == n - 1 else bi
            ax, ay, bx, by = p["x"][idx], p["y"][idx], p["x"][idx + 1], p["y"][idx + 1]
            if bi == 0 and (o["x"][j] - ax) * (bx - ax) + (o["y"][j] - ay) * (by - ay) < 0:
                continue
            if bi == n - 1 and (o["x"][j] - bx) * (bx - ax) + (o["y"][j] - by) * (by - ay) > 0:
                continue
            cross = (bx - ax) * (o["y"][j] - ay) - (by - ay) * (o["x"][j] - ax)
            lat = abs(cross) / np.hypot(bx - ax, by - ay)
            lat = 0.0 if lat < eps else (lat if cross > 0 else -lat)
            if lo <= lat <= hi and (best is None or s[bi] < best[0]):
                best = (s[bi], lat, j, bi)
        if best is None:
            assert r["flag"] == 0 and r["dis_lng"] == 999 and r["dis_lat"] == 999      # sentinel callers rely on
        else:
            assert r["flag"] == 1 and r["path_id"] == best[3]
            assert r["dis_lng"] == pytest.approx(best[0], rel=1e-12) and r["dis_lat"] == pytest.approx(best[1], rel=1e-9, abs=1e-12)
            assert r["ob"]["x"] == o["x"][best[2]]


def test_helper_curves(dm, oracle, cfg):
    rng = np.random.default_rng(8)
    s, e = (3.0, 4.0, 30.0), (40.0, 12.0, 350.0)
    b = oracle.BezierPlanning(cfg, s, e)
    assert (b["x"][0], b["y"][0]) == (3.0, 4.0) and b["x"][-1] == pytest.approx(40.0) and b["y"][-1] == pytest.approx(12.0)
    h0 = np.degrees(np.arctan2(b["y"][1] - b["y"][0], b["x"][1] - b["x"][0]))
    h1 = np.degrees(np.arctan2(b["y"][-1] - b["y"][-2], b["x"][-1] - b["x"][-2])) % 360
    assert h0 == pytest.approx(30.0, abs=0.5) and h1 == pytest.approx(350.0, abs=0.5)     # leaves/arrives along the headings
    p = _wavy(dm, 77, rng)
    m = oracle.MeanPoints(cfg, p)
    seg = np.hypot(np.diff(m["x"]), np.diff(m["y"]))
    assert m[0] == p[0] and m["x"][-1] == pytest.approx(p["x"][-1]) and seg.std() / seg.mean() < 0.02   # uniform spacing
    for off in (-3.75, 0.9):
        q = oracle.CreateNewPath(cfg, p, off)
        d = np.hypot(q["x"] - p["x"], q["y"] - p["y"])
        assert np.allclose(d, abs(off), rtol=1e-12)
        # negative = left of the direction of travel (Decision.cpp:629 vs :667)
        tx, ty = np.gradient(p["x"]), np.gradient(p["y"])
        side = tx * (q["y"] - p["y"]) - ty * (q["x"] - p["x"])
        assert (np.sign(side) == (1 if off < 0 else -1)).all()


def test_rasterise_equals_bruteforce(dm, oracle, cfg):
    rng = np.random.default_rng(9)
    for _ in range(5):
        o = np.zeros(12, dm.ObPoint)
        o["x"], o["y"] = rng.uniform(-3, 35, 12), rng.uniform(-3, 35, 12)        # some straddle the border
        o["radius"] = rng.uniform(0.3, 1.5, 12).astype(np.float32)
        g = oracle.rasterise(cfg, (0.0, 0.0), o)
        assert np.array_equal(g, oracle.rasterise(cfg, (0.0, 0.0), o, brute=True))
        assert set(np.unique(g)) <= {0, 1}
    assert not oracle.rasterise(cfg, (0.0, 0.0), o[:0]).any()


def _dijkstra(grid, start, goal):
    H, W = grid.shape
    dist = {start: 0}
    pq = [(0, start)]
    while pq:
        d, c = heapq.heappop(pq)
        if c == goal:
            return d
        if d > dist[c]:
            continue
        x, y = c % W, c // W
        for k in range(8):
            nx, ny = x + DX[k], y + DY[k]
            if 0 <= nx < W and 0 <= ny < H and not grid[ny, nx]:
                nd = d + (14 if k & 1 else 10)
                n = ny * W + nx
                if nd < dist.get(n, 1 << 60):
                    dist[n] = nd
                    heapq.heappush(pq, (nd, n))
    return None


def test_grid_search_is_optimal_and_well_formed(dm, oracle, cfg):
    """The batched LIFO bucket search returns a minimum-cost 8-connected path (checked against an
    independent Dijkstra), each cell is expanded once, and the digest matches its definition."""
    W = 128
    sc = dm.gen_scenes(cfg, 40, 24, 24, junction_every=0)
    found = 0
    for s in range(24):
        obs = sc["obs_pool"][s * 24:(s + 1) * 24]
        grid = oracle.rasterise(cfg, (0.0, 0.0), obs)
        si = sc["scene_in"][s]
        st = oracle.cell_of(cfg, (0.0, 0.0), float(si["loc"]["globalpoint"]["x"]), float(si["loc"]["globalpoint"]["y"]))
        go = oracle.cell_of(cfg, (0.0, 0.0), float(si["goal"]["x"]), float(si["goal"]["y"]))
        out, order, path = oracle.grid_search(cfg, grid, st, go, order_cap=W * W)
        status = int(out["status"][0])
        g2 = grid.copy()
        g2.reshape(-1)[st] = 0
        ref = None if grid.reshape(-1)[go] else _dijkstra(g2, st, go)
        if grid.reshape(-1)[go]:
            assert status == dm.G_GOAL_BLOCKED
            continue
        if ref is None:
            assert status == dm.G_NO_PATH
            continue
        found += 1
        assert status == dm.G_FOUND and int(out["path_cost"][0]) == ref                      # optimal
        assert len(set(order.tolist())) == len(order) == int(out["n_expanded"][0])              # closed once
        assert path[0] == st and path[-1] == go
        cost = 0
        for a, b in zip(path[:-1], path[1:]):
            dx, dy = b % W - a % W, b // W - a // W
            assert max(abs(dx), abs(dy)) == 1 and not g2.reshape(-1)[b]
            cost += 14 if dx and dy else 10
        assert cost == ref
        mix = lambda v: _mix64(v)
        dig = sum(_mix64((k << 32) | int(c)) for k, c in enumerate(order)) & ((1 << 64) - 1)
        assert dig == int(out["order_digest"][0])
    assert found >= 12


def _mix64(v):
    M = (1 << 64) - 1
    z = (v + 0x9e3779b97f4a7c15) & M
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
    return z ^ (z >> 31)


def test_grid_search_optimal_on_random_mazes(dm, oracle):
    """Jump points, forced neighbours and bounded diagonal jumps on cluttered grids: cost equals Dijkstra's,
    NO_PATH exactly when Dijkstra finds none."""
    cfg = dm.default_config(64)
    rng = np.random.default_rng(21)
    n_found = n_none = 0
    for trial in range(250):
        dens = float(rng.choice([0.05, 0.15, 0.25, 0.35, 0.45, 0.55, 0.62]))
        grid = (rng.random((64, 64)) < dens).astype(np.uint8)
        if trial % 5 == 0:                       # walls with gaps: long corridors, forced neighbours at the gaps
            grid[:] = 0
            for x in range(8, 64, 8):
                grid[:, x] = 1
                grid[rng.integers(0, 64, 3), x] = 0
        st, go = int(rng.integers(0, 64 * 64)), int(rng.integers(0, 64 * 64))
        grid.reshape(-1)[go] = 0
        out, order, path = oracle.grid_search(cfg, grid, st, go, order_cap=64 * 64)
        g2 = grid.copy()
        g2.reshape(-1)[st] = 0
        ref = _dijkstra(g2, st, go)
        if ref is None:
            assert int(out["status"][0]) == dm.G_NO_PATH, trial
            n_none += 1
            continue
        n_found += 1
        assert int(out["status"][0]) == dm.G_FOUND and int(out["path_cost"][0]) == ref, (trial, dens)
        assert path[0] == st and path[-1] == go
        steps = np.diff(np.stack([path % 64, path // 64]), axis=1)
        assert (np.abs(steps) <= 1).all() and not g2.reshape(-1)[path].any()               # 8-connected, through free cells
        assert int((np.abs(steps).sum(axis=0) == 2).sum()) * 14 + int((np.abs(steps).sum(axis=0) == 1).sum()) * 10 == ref
    assert n_found > 120 and n_none > 20


def test_scoring_prefers_clear_low_curvature_candidates(dm, oracle, cfg):
    sc = dm.gen_scenes(cfg, 70, 16, 8, junction_every=0)
    st = sc["state"].copy()
    _, gout, _ = oracle.plan_tick_batch(cfg, sc, st)
    for g in gout:
        nc = int(g["n_candidates"])
        cost = g["cand_cost"][:nc]
        assert int(g["best_candidate"]) == int(np.argmin(cost))                      # argmin, ties to the lowest index
        assert (g["cand_col"][:nc] >= 0).all() and (g["cand_prog"][:nc] >= 0).all() and (g["cand_prog"][:nc] <= 1).all()
        assert nc == (17 if g["status"] == dm.G_FOUND else 16)                        # the grid path joins only when found
        assert np.isfinite(g["best_path"]["x"]).all()


def test_dynamic_obstacles_move_with_the_tick(dm, oracle):
    cfg = dm.default_config(128)
    cfg["dynamic_obstacles"] = 1
    sc = dm.gen_scenes(cfg, 5, 1, 8, junction_every=0)
    o3 = oracle.effective_obstacles(cfg, sc["obs_pool"], sc["mot_pool"], 3)
    assert np.allclose(o3["x"], sc["obs_pool"]["x"] + sc["mot_pool"]["vx"] * 0.3, rtol=1e-15)
    cfg["dynamic_obstacles"] = 0
    assert oracle.effective_obstacles(cfg, sc["obs_pool"], sc["mot_pool"], 3).tobytes() == sc["obs_pool"].tobytes()


def test_curvature_fence_of_the_scoring_kernel_is_decided_by_the_oracle_chain():
    """k_score sums (1 / R)^2 from SQUARED side lengths (one division, no square root; kernels_score.hpp).  R is fenced to
    1000 where sinA < 0.001 - a discontinuity of the specification - so a point must fall on the same side of the fence as in
    the oracle's chain of operations (radius3 in oracle/dmpp_grid_oracle.c), not on the side a differently rounded expression
    says.  This replays the kernel's arithmetic in numpy (IEEE double, no contraction - as the device build) on triples whose
    sinA sits within 1e-12 ... 1e-3 (relative) of the fence: the fast form may only ever be used where it agrees with the
    chain, i.e. every disagreement about the side must lie inside the band the kernel hands to the chain itself."""
    rng = np.random.default_rng(11)
    n = 400000
    L1 = rng.uniform(0.05, 3.0, n); L2 = rng.uniform(0.05, 3.0, n)
    rel = 10.0 ** rng.uniform(-12, -3, n) * rng.choice([-1.0, 1.0], n)
    theta = 0.001 * (1.0 + rel)                                   # turning angle ~ sinA
    phi = rng.uniform(0, 2 * np.pi, n)
    ax = rng.uniform(-500, 500, n); ay = rng.uniform(-500, 500, n)
    mx = ax + L1 * np.cos(phi); my = ay + L1 * np.sin(phi)
    fx = mx + L2 * np.cos(phi + theta); fy = my + L2 * np.sin(phi + theta)
    # the oracle's chain
    dis1 = np.sqrt((ax - mx) * (ax - mx) + (ay - my) * (ay - my))
    dis2 = np.sqrt((mx - fx) * (mx - fx) + (my - fy) * (my - fy))
    dis3 = np.sqrt((ax - fx) * (ax - fx) + (ay - fy) * (ay - fy))
    den = 2 * dis1 * dis2
    cosA = (dis1 * dis1 + dis2 * dis2 - dis3 * dis3) / den
    with np.errstate(invalid="ignore"):
        sinA = np.sqrt(1 - cosA * cosA)
    chain_unfenced = sinA >= 0.001
    k = 1 / (0.5 * dis3 / np.where(chain_unfenced, sinA, 1.0))
    chain = np.where(chain_unfenced, k * k, 0.001 * 0.001)
    # the kernel's squared form
    d1 = (ax - mx) * (ax - mx) + (ay - my) * (ay - my)
    d2 = (mx - fx) * (mx - fx) + (my - fy) * (my - fy)
    d3 = (ax - fx) * (ax - fx) + (ay - fy) * (ay - fy)
    den2 = 4 * d1 * d2; num = d1 + d2 - d3
    diff = den2 - num * num; fence = (0.001 * 0.001) * den2
    band = np.abs(diff - fence) <= 1e-6 * fence
    fast_unfenced = diff > fence
    fast = np.where(fast_unfenced, diff / (d1 * d2 * d3), 0.001 * 0.001)
    sides_differ = fast_unfenced != chain_unfenced
    assert band.sum() > 100                                      # the sample really probes the band
    assert not (sides_differ & ~band).any(), int((sides_differ & ~band).sum())
    kernel = np.where(band, chain, fast)
    assert np.allclose(kernel, chain, rtol=1e-7, atol=0)         # (outside the band: the squared form, to rounding - ~1e-9 for the
                                                                 # near-equal sides of a resampled path, more for sides 1 : 60 as drawn here)


def test_span_ends_certified_by_margin():
    """The rasteriser inside k_search (csrc/kernels_s.hpp `exact_span`) takes a = ceil(x_a), b = floor(x_b) as the run of a line
    without evaluating the predicate when neither estimate lies within `eps` of an integer.  This replays that rule in numpy -
    the raw hardware square root modelled as the exact one with up to 2^-26 of relative error - against the predicate itself,
    `du*du + dv2 <= R2` in double precision (the oracle's, dmpp_grid_oracle.c `orc_rasterise`): cells from 1 cm to 1 m, origins
    up to 1e7 m, radii from 1 mm to 30 m, grazing lines, and boundaries placed 1e-12 ... 1e-4 cells from a cell centre.  Every
    run the rule calls certain must be the predicate's run; what it does not call certain goes to the predicate on the GPU."""
    rng = np.random.default_rng(20261005)

    def pred(i, org, cell, ou, dv2, R2):
        du = (org + (i.astype(np.float64) + 0.5) * cell) - ou
        return du * du + dv2 <= R2

    for adversarial in (False, True):
        n = 400_000
        cell = rng.choice([0.01, 0.05, 0.07, 0.1, 0.2, 0.25, 0.3, 0.5, 1.0], n)
        W = rng.choice([32, 128, 512, 2048], n)
        org = rng.choice([0.0, 1e3, -5e4, 3.3e5, 1e6, -1e6, 1e7], n) + rng.uniform(-100, 100, n)
        extent = W * cell
        R = np.exp(rng.uniform(np.log(1e-3), np.log(30.0), n))
        R2 = R * R
        ou = org + rng.uniform(-0.2, 1.2, n) * extent
        graze = rng.random(n) < 0.2
        dv = np.where(graze, R * (1 - np.exp(rng.uniform(np.log(1e-16), np.log(1e-2), n))), rng.uniform(0, 1, n) * R)
        dv2 = dv * dv
        keep = dv2 <= R2
        h2 = R2 - dv2
        if adversarial:
            half0 = np.sqrt(np.maximum(h2, 0))
            k = np.floor((ou - half0 - org) / cell)
            d = np.exp(rng.uniform(np.log(1e-12), np.log(1e-4), n)) * rng.choice([-1, 1], n)
            ou = np.where(rng.random(n) < 0.5, org + (k + 0.5 + d) * cell + half0, org + (k + 0.5 + d) * cell - half0)
        half = np.sqrt(np.maximum(h2, 0)) * (1 + rng.uniform(-1, 1, n) * 2.0 ** -26)
        inv = 1.0 / cell
        xa = (ou - half - org) * inv - 0.5
        xb = (ou + half - org) * inv - 0.5
        # (the kernel's constant sums both axes' origins; one axis here: the smaller bound is the harder test)
        cq = 2.0 ** -47 * (2 * np.abs(org) + extent) + 2.0 ** -36 * cell + 2.0 ** -30 * (1.0 + R2)
        lim = 0.5 - (half * 2.0 ** -24 + cq) * inv
        sure = keep & (h2 >= R2 * 2.0 ** -40) & (np.abs((xa - np.floor(xa)) - 0.5) < lim) & (np.abs((xb - np.floor(xb)) - 0.5) < lim)
        idx = np.nonzero(sure)[0]
        assert len(idx) > n // 5
        ai = np.ceil(xa[idx]).astype(np.int64); bi = np.floor(xb[idx]).astype(np.int64)
        o, c, u, d2, r2 = org[idx], cell[idx], ou[idx], dv2[idx], R2[idx]
        nonempty = ai <= bi
        for off in range(-3, 4):
            inside_a = nonempty & (off >= 0) & (ai + off <= bi)
            inside_b = nonempty & (off <= 0) & (bi + off >= ai)
            assert np.array_equal(pred(ai + off, o, c, u, d2, r2), inside_a), (adversarial, off)
            assert np.array_equal(pred(bi + off, o, c, u, d2, r2), inside_b), (adversarial, off)
        # a run called empty: the predicate fails at the indices nearest to the obstacle too
        ic = np.floor((u - o) / c).astype(np.int64)
        for off in (-1, 0, 1):
            assert not np.any(pred(ic + off, o, c, u, d2, r2) & ~nonempty), (adversarial, off)
