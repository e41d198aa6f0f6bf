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


def _stop_worker(rank, world, port, folder, out):
    """Rank 1 alone is past its killtime; rank 0 alone would have seen the sentinel first.  Both must stop together."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    from pathlib import Path
    from vasp_amd.dist import agree_flags, init_from_env
    from vasp_amd.monolithic import stop_controls
    _, _, _, dist = init_from_env(prefer_gpu=False)
    agree = lambda flags: agree_flags(dist, flags)
    lines = []
    t_loop = time.perf_counter() - (100.0 if rank == 1 else 0.0)          # only rank 1's clock is over the budget
    first = stop_controls(Path(folder), 50.0, t_loop, agree, rank == 0, lines.append)
    dist.barrier()
    if rank == 0:
        (Path(folder) / "killturtle").write_text("")
    dist.barrier()
    second = stop_controls(Path(folder), None, time.perf_counter(), agree, rank == 0, lines.append)
    dist.barrier()
    out.put((rank, first, second, (Path(folder) / "killturtle").exists(), len(lines)))
    dist.destroy_process_group()


def test_stop_controls_agree_across_ranks(tmp_path):
    """killtime compares each rank's own clock and the sentinel is unlinked by rank 0 while others look for it;
    without an agreement step ranks disagree on `stop` and the job hangs in the next collective."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stop_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, first, second, still_there, nlines in res:
        assert first is True and second is True            # both ranks stop in both situations
        assert still_there is False                        # rank 0 removed the sentinel after the agreement
        assert nlines == 2
