"""CPU, world_size 2, gloo: the N>1 path of bench.py — scenes generated on rank 0, scattered, ticked
per rank with no data-path collective, results gathered — checked against a single-process run.
(The per-rank tick is the oracle here because this box has no GPU; on GPUs it is pp_plan_tick.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, n_obs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import dmpp_amd as dm
    import oracle_binding
    from dmpp_amd_pkg import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dm.default_config(128)
    recv = sharding.scatter_scenes(dm, dist, torch, cfg, n, n_obs, rank, world, torch.device("cpu"))
    # what this rank received must be exactly the shard it would have generated itself
    local = dm.gen_scenes(cfg, rank * n, n, n_obs, 8)
    same = all(recv[k].numpy().tobytes() == local[k].tobytes() for k in sharding.KEYS)
    sc = {k: np.frombuffer(recv[k].numpy().tobytes(), local[k].dtype).copy() for k in sharding.KEYS}
    sc["n_obs"] = n_obs
    orc = oracle_binding.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    st = sc["state"].copy()
    plan, gout, _ = orc.plan_tick_batch(cfg, sc, st)
    mine = torch.from_numpy(gout["order_digest"].astype(np.int64))
    allg = sharding.gather_results(dist, torch, mine, rank, world)
    dist.barrier()
    if rank == 0:
        q.put((same, [t.numpy().copy() for t in allg]))
    else:
        q.put((same, None))
    dist.destroy_process_group()


def test_scatter_tick_gather_world2(dm, oracle):
    import torch.multiprocessing as mp
    n, n_obs, world = 6, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, n_obs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[0] for r in res), "a rank received bytes that differ from its own shard"
    gathered = [r[1] for r in res if r[1] is not None][0]
    cfg = dm.default_config(128)
    whole = dm.gen_scenes(cfg, 0, n * world, n_obs, 8)          # the same scenes in one process
    st = whole["state"].copy()
    _, gout, _ = oracle.plan_tick_batch(cfg, whole, st)
    want = gout["order_digest"].astype(np.int64)
    assert np.array_equal(np.concatenate(gathered), want)
