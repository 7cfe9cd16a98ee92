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
    # the gather of SURVEY 8(e): PlanOut + SceneState + GridOut of every scene, as bytes
    allg = {}
    for k, a in (("plan", plan), ("state", st), ("grid_out", gout)):
        mine = torch.from_numpy(np.frombuffer(a.tobytes(), np.uint8).copy())
        allg[k] = sharding.gather_results(dist, torch, mine, rank, world)
    dist.barrier()
    if rank == 0:
        q.put((same, {k: [t.numpy().copy() for t in v] for k, v in allg.items()}))
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
    plan, gout, _ = oracle.plan_tick_batch(cfg, whole, st)
    from parity_util import compare
    for k, want in (("plan", plan), ("state", st), ("grid_out", gout)):
        got = np.frombuffer(np.concatenate(gathered[k]).tobytes(), want.dtype)
        assert len(got) == n * world
        bad = compare(got, want, k, rtol=0.0, atol=0.0)
        assert not bad, bad[:5]


def test_bench_spawns_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) starts two ranks itself; on a box without GPUs every rank
    refuses to run (no CPU fallback) and the parent reports the exit codes and exits non-zero."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--scenes", "4", "--grid", "128", "--obstacles", "4", "--no-cpu-baseline", "--latency-ticks", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU: the spawn path is covered by the -m gpu rehearsal")
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2, r.stderr[-2000:]
    assert "rank exit codes" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
