#!/usr/bin/env python3
"""Runs ON THE GPU BOX: where the streamed tick's time goes - the same loop with the upload, the PlanOut download and the
GridOut download switched on one by one, the host's enqueue time per call, and the raw PCIe rates of this box.
    gpurun -- 'python tools/stream_probe.py [scenes] [steps]'"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if os.environ.get("TORCH"):          # bench.py initialises torch's HIP context before the library's
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
import dmpp_amd as dm
from parity_util import move_ego

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n_obs, depth = 64, int(os.environ.get("DEPTH", "4"))
cfg = dm.default_config(512)
sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=8)
pl = dm.Planner(cfg, device=0, max_scenes=n, max_obs_total=n * n_obs)
pl.set_scenes(sc)
pl.set_state(sc["state"])
snaps = []
rng = np.random.default_rng(3)
if os.environ.get("PARTS"):          # what bench.py does before its streamed leg: grid stage off and on again
    c2 = cfg.copy(); c2["grid_stage"] = 0
    pl.set_config(c2)
    for _ in range(5):
        pl.tick()
    pl.sync()
    pl.set_config(cfg)
for k in range(depth + 2):
    move_ego(sc, 1)
    if os.environ.get("JITTER"):
        sc["obs_pool"]["x"] += rng.uniform(-0.3, 0.3, len(sc["obs_pool"]))
        sc["obs_pool"]["y"] += rng.uniform(-0.3, 0.3, len(sc["obs_pool"]))
    snaps.append((dm.pinned_copy(sc["scene_in"]), dm.pinned_copy(sc["obs_pool"])))
WIRE = bool(os.environ.get("WIRE"))      # only the published records (PlanningOut + PlanningStatus) instead of the whole PlanOut
ress = [dm.pinned_empty(n, dm.PlanningOut) for _ in range(depth)]
shows = [dm.pinned_empty(n, dm.PlanningStatus) for _ in range(depth)]
plans = [dm.pinned_empty(n, dm.PlanOut) for _ in range(depth)]
grids = [dm.pinned_empty(n, dm.GridOut) for _ in range(depth)]


def run(k_steps, up, dp, dg, acc=None):
    ids = []
    for t in range(k_steps):
        if len(ids) == depth:
            a0 = time.perf_counter()
            pl.wait_tick(ids.pop(0))
            if acc is not None:
                acc["wait"] += time.perf_counter() - a0
        a0 = time.perf_counter()
        if up:
            pl.update_async(*snaps[t % len(snaps)])
        a1 = time.perf_counter()
        pl.tick()
        a2 = time.perf_counter()
        if dp or dg:
            if WIRE:
                ids.append(pl.fetch_published_async(ress[t % depth] if dp else None, shows[t % depth] if dp else None, grids[t % depth] if dg else None))
            else:
                ids.append(pl.fetch_async(plans[t % depth] if dp else None, grids[t % depth] if dg else None))
        a3 = time.perf_counter()
        if acc is not None:
            acc["update"] += a1 - a0
            acc["tick"] += a2 - a1
            acc["fetch"] += a3 - a2
    for i in ids:
        pl.wait_tick(i)


CASES = (("plain ticks", (0, 0, 0)), ("upload only", (1, 0, 0)), ("PlanOut down only", (0, 1, 0)), ("GridOut down only", (0, 0, 1)),
         ("both down", (0, 1, 1)), ("upload + PlanOut", (1, 1, 0)), ("all", (1, 1, 1)), ("plain, streaming on", (0, 0, 0)))
if os.environ.get("UP_PART"):        # upload only the egos / only the obstacles
    part = os.environ["UP_PART"]
    snaps = [((a if part == "in" else None), (b if part == "obs" else None)) for a, b in snaps]
if os.environ.get("ONLY_ALL"):
    CASES = (CASES[0], CASES[-2], CASES[-1])
for name, (up, dp, dg) in CASES:
    run(12, up, dp, dg)
    pl.sync()
    acc = {"update": 0.0, "tick": 0.0, "fetch": 0.0, "wait": 0.0}
    t0 = time.perf_counter()
    run(steps, up, dp, dg, acc)
    pl.sync()
    dt = time.perf_counter() - t0
    print("%-20s %.3f ms/tick  %.2f M ticks/s   host us/tick: %s" % (name, dt / steps * 1e3, n * steps / dt / 1e6,
                                                                      " ".join("%s=%.0f" % (k, v / steps * 1e6) for k, v in acc.items())), flush=True)
    if os.environ.get("KERNELS"):
        pl.set_profile(1)
        pl.reset_kernel_ms()
        run(steps, up, dp, dg)
        pl.sync()
        print("      kernels us:", " ".join("%s=%.0f" % (k.replace("k_", ""), v[0] / max(v[1], 1) * 1e3) for k, v in pl.kernel_ms().items()), flush=True)
        pl.set_profile(0)

if os.environ.get("NO_RAW"):
    sys.exit(0)
# raw PCIe rates of this box: the HIP runtime the library runs on, pinned host memory, one stream
import ctypes as C
print("HIP runtime(s) mapped:", sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}))
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
st = pl.lib.pp_stream(pl.h)
for mb in (2, 4, 7, 11, 64):
    nbytes = mb << 20
    hbuf = dm.pinned_empty(nbytes, np.uint8)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), nbytes) == 0
    for direction, kind in (("h2d", 1), ("d2h", 2)):
        hip.hipStreamSynchronize(st)
        t0 = time.perf_counter()
        for _ in range(20):
            if kind == 1:
                hip.hipMemcpyAsync(d, hbuf.ctypes.data, nbytes, 1, st)
            else:
                hip.hipMemcpyAsync(hbuf.ctypes.data, d, nbytes, 2, st)
        t1 = time.perf_counter()
        hip.hipStreamSynchronize(st)
        dt = (time.perf_counter() - t0) / 20
        print("raw %s %d MiB: %.3f ms  %.1f GB/s   (host enqueue %.0f us / copy)" % (direction, mb, dt * 1e3, nbytes / dt / 1e9, (t1 - t0) / 20 * 1e6), flush=True)
