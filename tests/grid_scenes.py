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
