"""Hand-built grid scenes for the search's edge cases (shared by GPU tests and CPU oracle tests)."""
import numpy as np


def serpentine(dm, cfg, n_walls=7, radius=20.0, gap=40.0):
    """One scene: horizontal walls of large discs across the grid with a gap at alternating ends, ego in the bottom-left
    corner, goal beyond the last wall.  On 2048 x 2048 the optimal path crosses the 512 m world once per wall: five
    walls cost 124,162 (just inside DMPP_F_LIMIT = 131,070), seven walls need more than the limit (DMPP_G_COST_RANGE).
    Either way the search closes thousands of cells, far beyond the LDS closed-set hash (the HBM spill)."""
    W, H, cell = int(cfg["grid_w"][0]), int(cfg["grid_h"][0]), float(cfg["cell"][0])
    Lx, Ly = W * cell, H * cell
    discs = []
    pitch = Ly / (n_walls + 1)
    for w in range(n_walls):
        y = pitch * (w + 1)
        x0, x1 = (0.0, Lx - gap) if w % 2 == 0 else (gap, Lx)
        discs += [(x, y, radius) for x in np.arange(x0, x1 + 1e-9, radius * 1.8)]
    sc = dm.gen_scenes(cfg, 0, 1, len(discs), junction_every=0)
    si = sc["scene_in"]
    ox, oy = float(si["grid_origin"]["x"][0]), float(si["grid_origin"]["y"][0])
    for j, (x, y, r) in enumerate(discs):
        sc["obs_pool"][j]["x"], sc["obs_pool"][j]["y"], sc["obs_pool"][j]["radius"], sc["obs_pool"][j]["type"] = ox + x, oy + y, r, 0
    sc["mot_pool"][:] = 0
    si["loc"]["globalpoint"]["x"], si["loc"]["globalpoint"]["y"] = ox + 5.0, oy + 5.0
    si["goal"]["x"] = ox + (Lx - 5.0 if n_walls % 2 == 0 else 5.0)
    si["goal"]["y"] = oy + Ly - 5.0
    return sc


def pebble_field(dm, cfg, n_side=26, radius=0.2, jitter=2.0, seed=4, walls=2):
    """One scene with a very wide search frontier: a jittered lattice of small discs (every one a source of forced neighbours,
    i.e. of jump points) crossed by `walls` walls of large discs with the gap at alternating ends, so that the search has to
    flood most of the field before it finds the way round.  The open list grows to thousands of live entries - far more than
    the DMPP_OPEN_CAP slots the device keeps in LDS."""
    rng = np.random.default_rng(seed)
    W, H, cell = int(cfg["grid_w"][0]), int(cfg["grid_h"][0]), float(cfg["cell"][0])
    Lx, Ly = W * cell, H * cell
    discs = []
    px, py = Lx / (n_side + 1), Ly / (n_side + 1)
    for iy in range(n_side):
        for ix in range(n_side):
            discs.append((px * (ix + 1) + rng.uniform(-jitter, jitter), py * (iy + 1) + rng.uniform(-jitter, jitter), radius))
    R, gap = 0.06 * Ly, 0.12 * Lx
    for w in range(walls):
        y = Ly * (w + 1) / (walls + 1)
        x0, x1 = (0.0, Lx - gap) if w % 2 == 0 else (gap, Lx)
        discs += [(x, y, R) for x in np.arange(x0, x1 + 1e-9, R * 1.8)]
    sc = dm.gen_scenes(cfg, 0, 1, len(discs), junction_every=0)
    si = sc["scene_in"]
    ox, oy = float(si["grid_origin"]["x"][0]), float(si["grid_origin"]["y"][0])
    ex, ey = 4.0, 4.0
    gx, gy = (Lx - 4.0 if walls % 2 == 0 else 4.0), Ly - 4.0
    keep = [(x, y, r) for (x, y, r) in discs if (x - ex) ** 2 + (y - ey) ** 2 > (r + 3.0) ** 2 and (x - gx) ** 2 + (y - gy) ** 2 > (r + 3.0) ** 2]
    sc = dm.gen_scenes(cfg, 0, 1, len(keep), junction_every=0)
    si = sc["scene_in"]
    for j, (x, y, r) in enumerate(keep):
        sc["obs_pool"][j]["x"], sc["obs_pool"][j]["y"], sc["obs_pool"][j]["radius"], sc["obs_pool"][j]["type"] = ox + x, oy + y, r, 0
    sc["mot_pool"][:] = 0
    si["loc"]["globalpoint"]["x"], si["loc"]["globalpoint"]["y"] = ox + ex, oy + ey
    si["goal"]["x"], si["goal"]["y"] = ox + gx, oy + gy
    return sc
