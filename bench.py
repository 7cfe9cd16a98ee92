#!/usr/bin/env python3
"""bench.py — planning ticks/s on batched synthetic scenes (BASELINE.json metric).

A "step" is one Decision+Planning+grid tick (pp_plan_tick) over one resident batch of scenes:
BASELINE configs[1] — 1024 scenes, 512x512 occupancy grid, 64 obstacles — per GPU.

N > 1 (BASELINE configs[2] at N = 8): one process per GPU.  Either the driver starts the ranks
(`python -m torch.distributed.run ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) or a plain `python bench.py --gpus N` starts them itself: the parent, which never imports torch or
loads libdmpp.so (no HIP call, so nothing that has touched a GPU is forked or exec'ed), starts N fresh children
with that environment, relays rank 0's JSON line and exits non-zero if any child does.  The scenes of all ranks
are generated on rank 0 and SCATTERED over RCCL, every rank ticks its own shard with no data-path collective
(scenes are independent: weak scaling), and PlanOut + SceneState + GridOut of every scene are GATHERED back to
rank 0 over RCCL, where every shard is checked bit for bit against a single-GPU run of the same
scenes.  Scatter / gather / check are outside the timed region (inputs resident in HBM when timing starts).

Timing: W warm-up steps, barrier + device sync, K steps, barrier + device sync, MAX over ranks.
Per-kernel durations come from HIP events recorded by the library on the streams its kernels run on.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
GRID_KERNELS = ("k_effective_obstacles", "k_order", "k_search", "k_score")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenes", type=int, default=1024, help="scenes per GPU")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--obstacles", type=int, default=64)
    ap.add_argument("--dynamic", type=int, default=0, help="1: BASELINE configs[3] (dynamic obstacles, replan every tick)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--latency-ticks", type=int, default=200)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path with ranks sharing the GPUs of a smaller box (never a measured number)")
    ap.add_argument("--dump-gathered", default=None, help="rank 0 writes the gathered PlanOut / SceneState / GridOut of all ranks to this .npz (tests)")
    ap.add_argument("--no-verify-gather", action="store_true", help="skip the bit-for-bit check of the gathered shards on rank 0")
    ap.add_argument("--bucket-cap", type=int, default=0, help="experiment: PlannerConfig.bucket_cap (0: the default); <= 512 means no open-list spill area and no retry kernel")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the streamed / one-shot / configs[3] / configs[4] legs behind the timed region")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no HIP events around the kernels of the timed region (no per-kernel times, no roofline)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# plain `python bench.py --gpus N`: this process only starts the ranks (no torch, no HIP in here)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n_ranks):
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    for ln in (out0 or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if any(codes) or not lines:
        print("bench.py: rank exit codes %s, %d result line(s)" % (codes, len(lines)), file=sys.stderr)
        return max([c for c in codes if c > 0] + [1])
    print(lines[-1])
    return 0


def algorithmic_bytes(cfg, n_obs):
    """per-tick algorithmic bytes (planning tick alone, no junction refpath)."""
    W, H = int(cfg["grid_w"][0]), int(cfg["grid_h"][0])
    b_r = 3200 + 24 * n_obs + 128 + 3200 + 1700
    b_g = W * H + 24 * n_obs + 3200
    per_kernel = {                                   # per scene, per launch
        "k_effective_obstacles": 2 * 24 * n_obs,
        "k_decision": 480 * 24 + 24 * n_obs + 6 * 48 + 120 * 16,
        "k_planning": b_r,
        "k_front": 480 * 24 + 24 * n_obs + 6 * 48 + 120 * 16 + b_r,   # Decision + Planning in one launch
        "k_search": W * H + 24 * n_obs + 3200,       # SURVEY 8(d) B_G: one byte per grid cell + the obstacle list in + the path out (the kernel
                                                     # rasterises into LDS and never moves the grid through HBM: see roofline.traffic)
        "k_score": 24 * n_obs + 3200 + 3200,
    }
    return b_r, b_g, per_kernel


def pmc_traffic(kernel, tags):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/profile.sh: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of the same workload; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md): first tag that has it."""
    for tag in tags:
        path = os.path.join(ROOT, "profiles", tag + "_pmc.json")
        if os.path.exists(path):
            for name, k in json.load(open(path))["kernels"].items():
                if name.split("<")[0] == kernel and "hbm_bytes_gfx950_corrected" in k:
                    return k["hbm_bytes_gfx950_corrected"], "profiles/%s_pmc.json (rocprofv3 --pmc, 2*FETCH_SIZE + WRITE_SIZE)" % tag
    return None, None


def pmc_instructions(path=os.path.join(ROOT, "profiles", "r3_pmc_insts.txt")):
    """Vector / scalar instructions per launch of every kernel of the tick, from the committed rocprofv3 PMC pass (tools/pmc_insts.sh
    on configs[1], 1024 scenes): {kernel: (SQ_INSTS_VALU, SQ_INSTS_SALU)}.  None when the file is missing."""
    import ast
    if not os.path.exists(path):
        return None
    out = {}
    for ln in open(path):
        if " dispatches " not in ln or "{" not in ln:
            continue
        name = ln.split(" dispatches ")[0].replace("void ", "").replace("dmpp::", "").split("<")[0].strip()
        if name.startswith("k_") and name != "k_validate_scenes":       # (validation runs once per pp_set_scenes, not per tick)
            d = ast.literal_eval(ln[ln.index("{"):].strip())
            out[name] = (d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_SALU", 0))
    return out or None


def time_ticks(pl, steps, warmup, events=True):
    """W warm-up ticks, then K ticks between two full syncs; per-kernel averages from a second pass with events around
    every launch (the search kernel's from the timed pass).  Returns (seconds, {kernel: (ms_total, launches)})."""
    for _ in range(warmup):
        pl.tick()
    pl.sync()
    pl.set_profile(2 if events else 0)
    pl.reset_kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.tick()
    pl.sync()
    dt = time.perf_counter() - t0
    kms = pl.kernel_ms()
    if events:
        pl.set_profile(1)
        pl.reset_kernel_ms()
        for _ in range(steps):
            pl.tick()
        pl.sync()
        kms_all = pl.kernel_ms()
        kms = {k: (kms[k] if k == "k_search" else v) for k, v in kms_all.items()}
    pl.set_profile(0)
    return dt, kms


def streamed_leg(dm, np, pl, sc, n, n_obs, steps, depth=6):
    """New egos AND obstacles from the host every tick, PlanOut + GridOut of every tick downloaded, no pp_sync in between:
    pp_update_async -> pp_plan_tick -> pp_fetch_async, `depth` ticks in flight before the oldest is waited for
    (the reference reads its blackboard and publishes every tick: Planning.cpp:95-112,186,214)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_util import move_ego
    rng = np.random.default_rng(3)
    work = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in sc.items()}
    snaps = []
    for k in range(depth + 2):                       # distinct snapshots in pinned memory, rotated (never modified while in flight)
        move_ego(work, 1)
        work["obs_pool"]["x"] += rng.uniform(-0.3, 0.3, len(work["obs_pool"]))
        work["obs_pool"]["y"] += rng.uniform(-0.3, 0.3, len(work["obs_pool"]))
        snaps.append((dm.pinned_copy(work["scene_in"]), dm.pinned_copy(work["obs_pool"])))
    plans = [dm.pinned_empty(n, dm.PlanOut) for _ in range(depth)]
    grids = [dm.pinned_empty(n, dm.GridOut) for _ in range(depth)]
    ress = [dm.pinned_empty(n, dm.PlanningOut) for _ in range(depth)]
    shows = [dm.pinned_empty(n, dm.PlanningStatus) for _ in range(depth)]

    def run(k_steps, published_only):
        ids = []
        for t in range(k_steps):
            if len(ids) == depth:
                pl.wait_tick(ids.pop(0))             # the output buffers of tick t - depth are about to be reused
            a, b = snaps[t % len(snaps)]
            pl.update_async(a, b)
            pl.tick()
            if published_only:
                ids.append(pl.fetch_published_async(ress[t % depth], shows[t % depth], grids[t % depth]))
            else:
                ids.append(pl.fetch_async(plans[t % depth], grids[t % depth]))
        for i in ids:
            pl.wait_tick(i)

    def timed(published_only):
        run(3 * depth, published_only)
        pl.sync()
        t0 = time.perf_counter()
        run(steps, published_only)
        pl.sync()
        return time.perf_counter() - t0
    dt = timed(False)
    dt_pub = timed(True)
    up = n * dm.SceneIn.itemsize + n * n_obs * dm.ObPoint.itemsize
    down = n * (dm.PlanOut.itemsize + dm.GridOut.itemsize)
    down_pub = n * (dm.PlanningOut.itemsize + dm.PlanningStatus.itemsize + dm.GridOut.itemsize)
    return {"published_only": {"ticks_per_s": n * steps / dt_pub, "ms_per_step": dt_pub / steps * 1e3, "MB_down": down_pub / 1e6,
                               "note": "pp_fetch_published_async: PlanningOut + PlanningStatus (what SetUdpSendCtrl / SetPlanningStatus receive, "
                                       "Planning.cpp:186,214) + GridOut instead of the whole PlanOut + GridOut"},
            "ticks_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "ticks_in_flight": depth,
            "MB_up": up / 1e6, "MB_down": down / 1e6, "PCIe_GBps_up": up * steps / dt / 1e9, "PCIe_GBps_down": down * steps / dt / 1e9,
            "note": "per tick: SceneIn + obstacle pool uploaded from pinned host memory, PlanOut + GridOut of every scene downloaded "
                    "into pinned host memory; no host wait except for the tick `ticks_in_flight` behind"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a profiler's preloaded library initialises the GPU before this script starts: starting the ranks from here would be
        # an exec from a process that has touched the GPU (which takes this pool's machines down)
        preload = os.environ.get("LD_PRELOAD", "").lower()
        if "rocprof" in preload or "roctracer" in preload or any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ):
            sys.exit("bench.py: --gpus %d under a profiler: refusing to start ranks from a profiled process "
                     "(profile one rank: --gpus 1, or let torch.distributed.run start the ranks)" % args.gpus)
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import dmpp_amd as dm

    # N = 1 needs neither torch.distributed nor torch's device memory: the library is loaded alone, so that it runs on the HIP
    # runtime it was built against (/opt/rocm).  `import torch` first would bind libdmpp.so to the older runtime bundled in the
    # torch wheel (same SONAME, first one loaded wins), on which asynchronous copies behave differently (tools/stream_probe.py:
    # the streamed leg loses ~15 %).  The device-wide synchronisation around the timed region is hipDeviceSynchronize through
    # pp_device_synchronize - what torch.cuda.synchronize() calls.  N > 1: torch.distributed over RCCL, as before.
    torch = None
    rehearsal = args.backend != "nccl"
    dist = None
    comm_dev = None
    if world > 1:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the planning path has no CPU fallback")
        # --backend gloo is a rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (the ranks share the
        # cards, the scatter / gather go through host tensors); the measured runs use RCCL, one rank per GPU
        if rehearsal:
            local_rank %= torch.cuda.device_count()
        elif torch.cuda.device_count() < world:
            raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, this box shows %d (use --backend gloo to rehearse)"
                             % (world, world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        comm_dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(args.backend)
        else:
            dist.init_process_group("nccl", device_id=comm_dev)   # nccl == RCCL on ROCm

    cfg = dm.default_config(args.grid)
    if args.dynamic:
        cfg["dynamic_obstacles"] = 1
        cfg["force_replan"] = 1
    if args.bucket_cap:
        cfg["bucket_cap"] = args.bucket_cap
    n = args.scenes
    n_obs = args.obstacles
    pl = dm.Planner(cfg, device=local_rank, max_scenes=n, max_obs_total=max(n * n_obs, 1))

    # ---- inputs: generated on rank 0, scattered over RCCL, handed to the library as device pointers ----
    scatter_ms = None
    if world == 1:
        sc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=8)
        pl.set_scenes(sc)
        pl.set_state(sc["state"])
    else:
        from dmpp_amd_pkg import sharding
        recv, scatter_ms = sharding.scatter_scenes(dm, dist, torch, cfg, n, n_obs, rank, world, comm_dev, timed=True)
        recv = {k: v.cuda() for k, v in recv.items()}
        torch.cuda.synchronize()
        lib = pl.lib
        dm._check(lib.pp_set_scenes(pl.h, n, recv["scene_in"].data_ptr(), recv["lane_pool"].data_ptr(), recv["attr_pool"].data_ptr(),
                                    n * 3 * dm.GEN_LANE_PTS,
                                    recv["ref_pool"].data_ptr(), n * dm.GEN_REF_PTS, recv["obs_pool"].data_ptr(),
                                    recv["mot_pool"].data_ptr(), n * n_obs))
        dm._check(lib.pp_set_state(pl.h, recv["state"].data_ptr(), n))
        pl.n = n

    def barrier():
        pl.device_synchronize()         # the library's own streams, then hipDeviceSynchronize: everything else on this device
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        pl.tick()
    barrier()
    # HIP events around the search kernel only inside the timed region (the dominant kernel: roofline); the other kernels' average
    # durations come from a second, untimed pass of the same K steps with events around every launch (an event pair costs queue time)
    pl.set_profile(0 if args.no_kernel_events else 2)
    pl.reset_kernel_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pl.tick()
    barrier()
    dt_local = dt = time.perf_counter() - t0
    kms = pl.kernel_ms()
    ticks_run = args.warmup + args.steps
    if not args.no_kernel_events:
        ticks_run += args.steps
        pl.set_profile(1)
        pl.reset_kernel_ms()
        for _ in range(args.steps):
            pl.tick()
        barrier()
        kms_all = pl.kernel_ms()
        kms = {k: (kms[k] if k == "k_search" else v) for k, v in kms_all.items()}
    pl.set_profile(0)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    b_r, b_g, per_kernel = algorithmic_bytes(cfg, n_obs)

    def roofline_of(kms_):
        dom = max(kms_, key=lambda k: kms_[k][0])
        dom_ms, dom_launches = kms_[dom]
        avg_ms = dom_ms / max(dom_launches, 1)
        achieved = per_kernel[dom] * n / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        return dom, avg_ms, achieved

    # ---- results gathered over RCCL: PlanOut + SceneState + GridOut of every scene, checked on rank 0 ----
    gout = pl.get_grid_out()
    gathered = None
    multi = None
    if dist is not None:
        from dmpp_amd_pkg import sharding
        mine = sharding.result_tensors(dm, torch, pl, n, comm_dev)      # device-to-device copies out of the handle's buffers
        torch.cuda.synchronize()
        if not rehearsal:
            dist.barrier()
        g0 = time.perf_counter()
        allg = {k: sharding.gather_results(dist, torch, v, rank, world) for k, v in mine.items()}
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        dom, avg_ms, achieved = roofline_of(kms)
        stats = [None] * world
        dist.all_gather_object(stats, {"rank": rank, "ms_per_step": dt_local / args.steps * 1e3, "kernel": dom, "avg_launch_ms": avg_ms,
                                       "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS,
                                       "search_status_counts": np.bincount(gout["status"], minlength=dm.G_STATUS_COUNT).tolist()})
        if rank == 0:
            dts = {"plan": dm.PlanOut, "state": dm.SceneState, "grid_out": dm.GridOut}
            gathered = {k: [np.frombuffer(t.cpu().numpy().tobytes(), dts[k]) for t in allg[k]] for k in dts}
            gathered_bytes = sum(t.numel() for k in allg for t in allg[k])
            checked, mism = 0, []
            if not args.no_verify_gather:
                # every shard again on THIS GPU alone, the same W + K ticks from the same generated scenes: the gathered
                # records must be identical (integers and floats bit for bit; NaN == NaN)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from parity_util import compare
                plv = dm.Planner(cfg, device=local_rank, max_scenes=n, max_obs_total=max(n * n_obs, 1))
                for r in range(world):
                    scr = dm.gen_scenes(cfg, r * n, n, n_obs, junction_every=8)
                    plv.set_scenes(scr)
                    plv.set_state(scr["state"])
                    for _ in range(ticks_run):
                        plv.tick()
                    plv.sync()
                    want = {"plan": plv.get_plan(), "state": plv.get_state(), "grid_out": plv.get_grid_out()}
                    for k in want:
                        mism += ["rank %d %s" % (r, m) for m in compare(gathered[k][r], want[k], k, rtol=0.0, atol=0.0)]
                    checked += 1
                plv.close()
                if mism:
                    print("bench.py: gathered results differ from a single-GPU run:\n" + "\n".join(mism[:10]), file=sys.stderr)
            multi = {"scatter_ms": scatter_ms, "gather_ms": gather_ms, "gathered_bytes": gathered_bytes,
                     "gathered_records": "PlanOut + SceneState + GridOut per scene (%d B)" % (dm.PlanOut.itemsize + dm.SceneState.itemsize + dm.GridOut.itemsize),
                     "shards_verified": checked, "shard_mismatches": len(mism), "per_rank": stats}
            if args.dump_gathered:
                np.savez(args.dump_gathered, **{k: np.concatenate(v) for k, v in gathered.items()})
            if mism:
                pl.close()
                dist.destroy_process_group()
                sys.exit(3)

    # ---- Part R alone (Decision + Planning ticks, grid engine off) on the same scenes, rank 0; SURVEY 8(d) asks for
    #      the two parts separately as well as combined.  Part G's kernels never run without Part R's inputs, so its
    #      figure is the serial sum of its kernels' average launch times from the timed region above. ----
    parts = None
    if rank == 0:
        cfg_r = cfg.copy()
        cfg_r["grid_stage"] = 0
        pl.set_config(cfg_r)
        for _ in range(max(args.warmup, 2)):
            pl.tick()
        pl.sync()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            pl.tick()
        pl.sync()
        r_dt = time.perf_counter() - r0
        pl.set_config(cfg)
        g_ms = sum(v[0] / max(v[1], 1) for k, v in kms.items() if k in GRID_KERNELS)
        front_ms = sum(v[0] / max(v[1], 1) for k, v in kms.items() if k in ("k_effective_obstacles", "k_decision", "k_planning", "k_front"))
        parts = {"R_only_ticks_per_s": n * args.steps / r_dt, "R_only_ms_per_step": r_dt / args.steps * 1e3,
                 "G_kernels_serial_ms": g_ms, "G_kernels_serial_ticks_per_s": (n / (g_ms * 1e-3)) if g_ms > 0 else None,
                 "front_chain_ms": front_ms,
                 "note": "per GPU; R = Decision + Planning with the grid stage off; G = sum of the grid-engine kernels' average launch times "
                         "inside the combined tick; front_chain = obstacle snapshot + Decision + Planning, the kernels that share one stream beside the searches"}

    # ---- the tick with a NEW host snapshot every tick and every tick's results downloaded (streamed), the one-shot batch
    #      call (upload, tick, download with host waits: the SURVEY 8(b) signature), and BASELINE configs[3] / configs[4] in the
    #      same process, so that the driver's record carries them ----
    streamed = batch_call = other = None
    if rank == 0 and world == 1 and not args.no_extra_legs:
        pl.set_scenes(sc)
        pl.set_state(sc["state"])
        streamed = streamed_leg(dm, np, pl, sc, n, n_obs, max(args.steps, 60))
        stb = sc["state"].copy()
        pl.plan_tick_batch(sc, stb)
        b0 = time.perf_counter()
        for _ in range(5):
            pl.plan_tick_batch(sc, stb)
        bdt = (time.perf_counter() - b0) / 5
        batch_call = {"ticks_per_s": n / bdt, "ms_per_step": bdt * 1e3,
                      "note": "pp_plan_tick_batch: pageable host buffers in, SceneIn + lane / refpath / obstacle pools + SceneState uploaded, "
                              "PlanOut + SceneState + GridOut downloaded, host waits between the stages (PCIe-inclusive; never the headline)"}
        if n == 1024 and n_obs == 64 and not args.dynamic and args.grid == 512:
            other = {}
            pl.close()      # one handle at a time: the streams of a second handle share the process's few hardware queues with the first one's,
                            # and its three searches then queue behind each other instead of overlapping (measured: 0.68 instead of 1.15 M ticks/s)
            for name, (g2, o2, dyn2, tag2) in (("configs[3]", (512, 256, 1, "c3")), ("configs[4]", (2048, 64, 0, "c4"))):
                cfg2 = dm.default_config(g2)
                if dyn2:
                    cfg2["dynamic_obstacles"] = 1
                    cfg2["force_replan"] = 1
                pl2 = dm.Planner(cfg2, device=local_rank, max_scenes=n, max_obs_total=n * o2)
                sc2 = dm.gen_scenes(cfg2, 0, n, o2, junction_every=8)
                pl2.set_scenes(sc2)
                pl2.set_state(sc2["state"])
                steps2 = 24
                dt2, kms2 = time_ticks(pl2, steps2, 6)
                g2o = pl2.get_grid_out()
                _, _, pk2 = algorithmic_bytes(cfg2, o2)
                s_ms = kms2["k_search"][0] / max(kms2["k_search"][1], 1)
                traffic2, src2 = pmc_traffic("k_search", ["r3_" + tag2, "r2_" + tag2])
                ach2 = pk2["k_search"] * n / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
                other[name] = {"workload": "%d scenes, %dx%d grid, %d %s obstacles%s" % (n, g2, g2, o2, "dynamic" if dyn2 else "static",
                                                                                        ", replan every tick" if dyn2 else ""),
                               "ticks_per_s": n * steps2 / dt2, "ms_per_step": dt2 / steps2 * 1e3, "steps": steps2, "warmup": 6,
                               "kernel_ms_avg": {k: v[0] / max(v[1], 1) for k, v in kms2.items() if v[1] > 0},
                               "search_status_counts": [int(v) for v in np.bincount(g2o["status"], minlength=dm.G_STATUS_COUNT)],
                               "roofline": {"bound": "hbm", "kernel": "k_search", "achieved": ach2, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                            "frac": ach2 / HBM_PEAK_GBPS, "traffic": traffic2, "traffic_source": src2,
                                            "algorithmic_bytes_per_launch": pk2["k_search"] * n, "avg_launch_ms": s_ms}}
                pl2.close()

    # ---- p50 plan latency, batch = 1 (rank 0): the C-ABI tick, and the C++ class surface CDecision::decide -> CPlanning::plan ----
    p50_ms = None
    p50_class_ms = None
    sc1 = None
    if rank == 0 and args.latency_ticks > 0:
        pl1 = dm.Planner(cfg, device=local_rank, max_scenes=1, max_obs_total=max(n_obs, 1))
        sc1 = dm.gen_scenes(cfg, 0, 1, n_obs, junction_every=0)
        pl1.set_scenes(sc1)
        pl1.set_state(sc1["state"])
        for _ in range(10):
            pl1.tick(sync=True)
        lat = []
        for _ in range(args.latency_ticks):
            a = time.perf_counter()
            pl1.tick(sync=True)
            lat.append((time.perf_counter() - a) * 1e3)
        p50_ms = float(np.percentile(lat, 50))
        pl1.close()
        try:
            from dmpp_amd_pkg import host_surface
            p50_class_ms = host_surface.p50_plan_ms(dm, cfg, sc1, device=local_rank, ticks=args.latency_ticks)
        except (ImportError, OSError) as e:          # the C++ host library is optional for the bench (never for the tests)
            print("bench.py: class-surface latency skipped: %s" % e, file=sys.stderr)

    # ---- CPU baseline: the oracle (a port of the path), timed on this box's host cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding
        orc = oracle_binding.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
        ncpu = os.cpu_count() or 1
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            pass
        scc = dm.gen_scenes(cfg, 0, n, n_obs, junction_every=8)       # the GPU's own batch
        stc = scc["state"].copy()
        orc.plan_tick_batch(cfg, scc, stc, n_threads=min(ncpu, 16))  # warm-up tick
        # A GPU box may grant this job fewer CPUs than it shows (a cgroup quota): more threads than that only get
        # throttled.  Time a few ticks at several thread counts and keep the fastest; `cores` reports that count.
        cores, per_tick = 1, None
        for cand in sorted({min(ncpu, k) for k in (8, 16, 32, 64, 128, ncpu)}):
            c0 = time.perf_counter()
            orc.plan_tick_batch(cfg, scc, stc, n_threads=cand, n_ticks=4)
            t = (time.perf_counter() - c0) / 4
            if per_tick is None or t < per_tick:
                cores, per_tick = cand, t
        # every thread takes its block of scenes through all the ticks on its own (scenes are independent): no barrier
        # between ticks and the thread start-up is paid once - the CPU's best case for this workload
        ticks = int(min(max(args.cpu_seconds / per_tick, 1), 40000))
        c0 = time.perf_counter()
        orc.plan_tick_batch(cfg, scc, stc, n_threads=cores, n_ticks=ticks)
        cdt = time.perf_counter() - c0
        ns1 = min(n, 64)
        sc1t = dm.gen_scenes(cfg, 0, ns1, n_obs, junction_every=8)
        st1 = sc1t["state"].copy()
        orc.plan_tick_batch(cfg, sc1t, st1, n_threads=1)
        one0 = time.perf_counter()
        one_ticks = 20
        orc.plan_tick_batch(cfg, sc1t, st1, n_threads=1, n_ticks=one_ticks)
        one_dt = time.perf_counter() - one0
        # p50 latency of ONE scene on one thread: the scene the GPU's batch-1 latency is measured on
        cpu_p50 = None
        if args.latency_ticks > 0:
            scl = sc1 if sc1 is not None else dm.gen_scenes(cfg, 0, 1, n_obs, junction_every=0)
            stl = scl["state"].copy()
            for _ in range(3):
                orc.plan_tick_batch(cfg, scl, stl, n_threads=1)
            clat = []
            for _ in range(min(args.latency_ticks, 100)):
                a = time.perf_counter()
                orc.plan_tick_batch(cfg, scl, stl, n_threads=1)
                clat.append((time.perf_counter() - a) * 1e3)
            cpu_p50 = float(np.percentile(clat, 50))
        cpu = {"value": n * ticks / cdt, "unit": "ticks/s", "cores": cores, "kind": "port",
               "single_thread_ticks_per_s": ns1 * one_ticks / one_dt, "p50_ms_batch1": cpu_p50,
               "cpu_count": ncpu, "cgroup_cpu_quota": quota,
               "sample": f"{ticks} consecutive ticks x {n} scenes of the same workload ({args.grid}x{args.grid}, {n_obs} obstacles), "
                         f"oracle C port, {cores} pthreads (fastest of 8..{ncpu} on this box: os.cpu_count() = {ncpu}, cgroup CPU quota = {quota}) each ticking its own block of scenes, {cdt:.1f} s; "
                         f"p50_ms_batch1 = one scene (seed 0, the scene of p50_plan_latency_ms_batch1), one thread, one tick per call"}

    if rank == 0:
        dom, avg_ms, achieved = roofline_of(kms)
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (tools/profile.sh:
        # separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload; FETCH_SIZE doubled per the
        # gfx950 note of MI355X_MICROARCH.md).  Only quoted for the workload it was measured on.
        traffic, traffic_src = None, None
        if n == 1024 and n_obs == 64 and not args.dynamic and args.grid in (512, 2048):
            sfx = "" if args.grid == 512 else "_c4"
            traffic, traffic_src = pmc_traffic(dom, ["r3" + sfx, "r2" + sfx, "r1" + sfx])
        elif n == 1024 and n_obs == 256 and args.dynamic and args.grid == 512:
            traffic, traffic_src = pmc_traffic(dom, ["r3_c3", "r2_c3"])
        status_counts = np.bincount(gout["status"], minlength=dm.G_STATUS_COUNT)
        if multi:
            status_counts = np.sum([s["search_status_counts"] for s in multi["per_rank"]], axis=0)
        searched = int(status_counts.sum() - status_counts[dm.G_GOAL_BLOCKED])
        total = n * world
        # how busy the vector units are over the whole tick: instructions of every kernel of the tick (committed PMC pass, configs[1]
        # at 1024 scenes) x 4 cycles per wave64 instruction, over the SIMD-cycles of the tick time (1024 SIMDs, 2.4 GHz)
        issue = None
        pi = pmc_instructions()
        if pi and n == 1024 and not args.dynamic and args.obstacles == 64 and args.grid == 512:
            valu = sum(v for v, _ in pi.values()); salu = sum(sc_ for _, sc_ in pi.values())
            issue = {"valu_insts_per_tick": valu, "salu_insts_per_tick": salu,
                     "frac_of_valu_issue_slots": valu * 4.0 / (1024 * 2.4e9 * (dt / args.steps)),
                     "source": "profiles/r3_pmc_insts.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU, per launch of each kernel of the tick)"}
        line = {
            "metric": "planning ticks/sec (batched scenes), %dx%d grid" % (args.grid, args.grid),
            "value": total * args.steps / dt,
            "unit": "ticks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ticks_run": ticks_run,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d scenes/GPU, %dx%d grid, %d %s obstacles"
                                   % (3 if args.dynamic else (4 if args.grid == 2048 else (2 if world > 1 else 1)), n, args.grid, args.grid, n_obs,
                                      "dynamic" if args.dynamic else "static"),
                       "scenes_per_gpu": n, "global_scenes": total, "grid": args.grid, "obstacles": n_obs,
                       "parallelism": ("scene-sharded x%d, RCCL scatter/gather outside the timed region" % world)
                                      + (" [REHEARSAL: gloo, ranks share GPUs - not a measurement]" if rehearsal else ""),
                       "algorithmic_bytes_per_tick": b_r + b_g,
                       "tick_GBps": (b_r + b_g) * total * args.steps / dt / 1e9},
            "p50_plan_latency_ms_batch1": p50_ms,
            "p50_class_surface_ms": p50_class_ms,
            "parts": parts,
            "search_status_counts": [int(v) for v in status_counts],
            # scenes whose goal cell is occupied end with GOAL_BLOCKED before any expansion: they are ticks (Decision, Planning,
            # rasterise and scoring run) but no search - the rate over the scenes that did search is quoted beside the headline
            "searched_scenes": searched,
            "ticks_per_s_searched": searched * args.steps / dt,
            "kernel_ms_avg": {k: (v[0] / max(v[1], 1)) for k, v in kms.items() if v[1] > 0},
            # algorithmic bytes of each kernel over its own average launch time; the kernels of neighbouring
            # ticks overlap on three streams, so these are per-kernel rates, not shares of the tick
            "kernel_algorithmic_GBps": {k: (per_kernel[k] * n / (v[0] / max(v[1], 1) * 1e-3) / 1e9 if v[0] > 0 else 0.0)
                                        for k, v in kms.items() if k in per_kernel and v[1] > 0},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": per_kernel[dom] * n, "avg_launch_ms": avg_ms,
                         # what the kernel really moves through HBM, over its launch time (traffic is far below the algorithmic bytes: the
                         # grid is rasterised into LDS): the kernel is bound by the latency of its longest scene's chain of steps, not by HBM
                         "measured_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                         "limiter": "latency: one wave per scene, the launch ends with its slowest scene (dependent LDS round trips and ~1,350 instructions per step)"
                                    + ("; the ticks in flight beside it take the share of the chip's vector issue slots given under `issue`" if issue else ""),
                         "issue": issue,
                         # up to three launches of the search run side by side (consecutive ticks on three streams): what the chip
                         # sustains is the same bytes over the tick time; `achieved` stays the per-launch figure
                         "launches_in_flight": 3 if (n >= 256 and dom == "k_search") else 1,
                         "achieved_overlapped": per_kernel[dom] * n / (dt / args.steps) / 1e9,
                         "frac_overlapped": per_kernel[dom] * n / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS,
                         "note": "algorithmic bytes per SURVEY 8(d): one byte per grid cell + the obstacle list + the path per scene; the kernel "
                                 "rasterises the obstacle list into LDS and never moves a grid through HBM (traffic = measured HBM bytes per launch)"
                                 + ("; rank 0's kernel, every rank's own figure under multi_gpu.per_rank" if multi else "")},
            "streamed": streamed,
            "plan_tick_batch": batch_call,
            "other_configs": other,
            "cpu_baseline": cpu,
        }
        if multi:
            line["multi_gpu"] = multi
        print(json.dumps(line))
    pl.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
