"""Scene sharding over ranks (one process per GPU): scenes are independent once the reference's
statics are per-scene state, so the only exchange is a scatter of inputs from rank 0
and a gather of results — torch.distributed over RCCL/xGMI on GPUs (backend "nccl"), gloo on CPU.
No collective sits on the data path of a tick."""
import time

import numpy as np

KEYS = ("scene_in", "lane_pool", "attr_pool", "ref_pool", "obs_pool", "mot_pool", "state")
RESULT_KEYS = ("plan", "state", "grid_out")


def shard_nbytes(dm, n_scenes, n_obs):
    """Bytes of each per-rank input buffer (fixed strides of pp_gen_scenes)."""
    return {
        "scene_in": n_scenes * dm.SceneIn.itemsize,
        "lane_pool": n_scenes * 3 * dm.GEN_LANE_PTS * dm.GlobalPoint3D.itemsize,
        "attr_pool": n_scenes * 3 * dm.GEN_LANE_PTS,
        "ref_pool": n_scenes * dm.GEN_REF_PTS * dm.GlobalPoint2D.itemsize,
        "obs_pool": max(n_scenes * n_obs, 1) * dm.ObPoint.itemsize,
        "mot_pool": max(n_scenes * n_obs, 1) * dm.ObMotion.itemsize,
        "state": n_scenes * dm.SceneState.itemsize,
    }


def _layout(sizes):
    """Offsets of the buffers inside one packed shard (256-byte aligned: the views are handed to the device as pointers)."""
    off, o = {}, 0
    for k in KEYS:
        off[k] = o
        o += (sizes[k] + 255) // 256 * 256
    return off, o


def scatter_scenes(dm, dist, torch, cfg, n_scenes, n_obs, rank, world, device, junction_every=8, timed=False):
    """Rank 0 generates every rank's shard (offsets are shard-local), packs each into ONE byte buffer and scatters them
    with one collective (rank 0 -> rank k directly: one xGMI link each, all at once); returns this rank's buffers as
    uint8 views of the received shard on `device` (and the milliseconds the collective took when `timed`)."""
    sizes = shard_nbytes(dm, n_scenes, n_obs)
    off, total = _layout(sizes)
    src = None
    if rank == 0:
        src = []
        for r in range(world):
            s = dm.gen_scenes(cfg, r * n_scenes, n_scenes, n_obs, junction_every)
            buf = np.zeros(total, np.uint8)
            for k in KEYS:
                buf[off[k]:off[k] + sizes[k]] = np.frombuffer(s[k].tobytes(), np.uint8)
            src.append(torch.from_numpy(buf).to(device))
    dst = torch.empty(total, dtype=torch.uint8, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.scatter(dst, src, src=0)
    if device.type == "cuda":
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    out = {k: dst[off[k]:off[k] + sizes[k]] for k in KEYS}
    return (out, ms) if timed else out


def result_tensors(dm, torch, pl, n_scenes, device):
    """PlanOut, SceneState and GridOut of the handle's resident scenes as uint8 tensors on `device`.  For a GPU
    `device` the library copies device-to-device straight into the tensors (pp_get_* take host or device pointers),
    so the RCCL gather needs no host hop."""
    dts = {"plan": dm.PlanOut, "state": dm.SceneState, "grid_out": dm.GridOut}
    getters = {"plan": pl.lib.pp_get_plan, "state": pl.lib.pp_get_state, "grid_out": pl.lib.pp_get_grid_out}
    out = {}
    for k in RESULT_KEYS:
        t = torch.empty(n_scenes * dts[k].itemsize, dtype=torch.uint8, device=device)
        dm._check(getters[k](pl.h, t.data_ptr(), n_scenes))
        out[k] = t
    return out


def gather_results(dist, torch, mine, rank, world):
    """Gathers one equally-shaped tensor per rank onto rank 0 (list of tensors there, None elsewhere)."""
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bufs, dst=0)
    return bufs
