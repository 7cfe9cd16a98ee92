"""Scene sharding over ranks (one process per GPU): scenes are independent once the reference's
statics are per-scene state, so the only exchange is a scatter of inputs from rank 0
and a gather of results — torch.distributed over RCCL/xGMI on GPUs (backend "nccl"), gloo on CPU.
No collective sits on the data path of a tick."""
import numpy as np

KEYS = ("scene_in", "lane_pool", "attr_pool", "ref_pool", "obs_pool", "mot_pool", "state")


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


def scatter_scenes(dm, dist, torch, cfg, n_scenes, n_obs, rank, world, device, junction_every=8):
    """Rank 0 generates every rank's shard (offsets are shard-local) and scatters it; returns this
    rank's buffers as uint8 tensors on `device`."""
    sizes = shard_nbytes(dm, n_scenes, n_obs)
    shards = None
    if rank == 0:
        shards = [dm.gen_scenes(cfg, r * n_scenes, n_scenes, n_obs, junction_every) for r in range(world)]
    out = {}
    for k in KEYS:
        dst = torch.empty(sizes[k], dtype=torch.uint8, device=device)
        src = None
        if rank == 0:
            src = [torch.from_numpy(np.frombuffer(s[k].tobytes(), np.uint8).copy()).to(device) for s in shards]
        dist.scatter(dst, src, src=0)
        out[k] = dst
    return out


def gather_results(dist, torch, mine, rank, world):
    """Gathers one equally-shaped tensor per rank onto rank 0 (list of tensors there, None elsewhere)."""
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bufs, dst=0)
    return bufs
