"""The N > 1 measurement path of bench.py on CPU: two gloo ranks, barrier + max-time / sum-units aggregation."""
import os
import socket

import pytest


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vasp_amd.dist import aggregate, init_from_env
    r, lr, w, dist = init_from_env(prefer_gpu=False)
    assert (r, w) == (rank, world) and dist is not None
    dist.barrier()
    t, u = aggregate(dist, elapsed_s=1.0 + rank, units=10.0 * (rank + 1))
    out.put((rank, t, u))
    dist.destroy_process_group()


def test_two_rank_aggregation():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(t, u) for _, t, u in res] == [(2.0, 30.0), (2.0, 30.0)]     # max of {1, 2} s, sum of {10, 20} units


def test_single_process_is_identity():
    from vasp_amd.dist import aggregate
    assert aggregate(None, 1.5, 7.0) == (1.5, 7.0)
