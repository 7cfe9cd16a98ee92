"""Field-by-field comparison of ABI records: integers bit-exact, floats within rtol (NaN == NaN)."""
import numpy as np

RTOL = 1e-6      # BASELINE.json north_star: "trajectory cost within 1e-6 relative for the float scoring"
ATOL = 1e-9


def compare(a, b, path="", rtol=RTOL, atol=ATOL, skip=()):
    """Returns a list of mismatch strings; also counts fields that are not bit-identical."""
    bad = []
    dt = a.dtype
    if dt.names:
        for name in dt.names:
            if name.startswith("_pad") or (path + "." + name).lstrip(".") in skip or name in skip:
                continue
            bad += compare(a[name], b[name], path + "." + name, rtol, atol, skip)
        return bad
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        return [f"{path}: shape {a.shape} vs {b.shape}"]
    if np.issubdtype(dt, np.floating):
        both_nan = np.isnan(a) & np.isnan(b)
        both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
        ok = both_nan | both_inf | (np.abs(a - b) <= atol + rtol * np.maximum(np.abs(a), np.abs(b)))
        if not ok.all():
            idx = np.argwhere(~ok)[0]
            bad.append(f"{path}: {int((~ok).sum())} float mismatches, first at {tuple(idx)}: {a[tuple(idx)]!r} vs {b[tuple(idx)]!r}")
    else:
        ne = a != b
        if ne.any():
            idx = np.argwhere(ne)[0]
            bad.append(f"{path}: {int(ne.sum())} integer mismatches, first at {tuple(idx)}: {a[tuple(idx)]!r} vs {b[tuple(idx)]!r}")
    return bad


def bit_identical_fraction(a, b):
    """Fraction of bytes that are identical (diagnostic only)."""
    ra = np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint8)
    rb = np.frombuffer(np.ascontiguousarray(b).tobytes(), np.uint8)
    return float((ra == rb).mean())


def move_ego(sc, step, dlat=0.0):
    """Advance every ego along its current lane by `step` lane points (keeps scenes consistent)."""
    import dmpp_amd as dm
    si = sc["scene_in"]
    for s in range(len(si)):
        lv = si["lanes"][s]
        ids = si["loc"]["id"][s]
        new_id = min(int(ids[0]) + step, int(lv["cur_n"]) - 2)
        p = sc["lane_pool"][int(lv["cur_off"]) + new_id]
        si["loc"]["globalpoint"]["x"][s] = p["x"]
        si["loc"]["globalpoint"]["y"][s] = p["y"] + dlat
        si["loc"]["globalpoint"]["dir"][s] = p["dir"]
        si["loc"]["id"][s] = new_id
