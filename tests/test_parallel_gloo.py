"""Data-parallel host logic on CPU: 2 processes over gloo (the GPU path uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from occm_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1000, generator=g)
    local = flat.clone()
    red = parallel.FlatGradAllReducer(flat, bucket_bytes=1024)          # 4 buckets of 256 floats
    assert len(red.buckets) == 4 and red.grad_scale == 0.5
    red.all_reduce()
    expect = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 + k)) for k in range(world))
    ok = torch.allclose(flat, expect, atol=1e-6)
    works = parallel.FlatGradAllReducer(local, bucket_bytes=4000).all_reduce(async_op=True)
    for wk in works:
        wk.wait()
    ok = ok and torch.allclose(local, expect, atol=1e-6)
    # overlapped form: slices reported ready out of order during "backward" (last layer first), the rest at the end, every element
    # reduced exactly once
    g2 = torch.Generator().manual_seed(100 + rank)
    over = torch.randn(1000, generator=g2)
    red2 = parallel.FlatGradAllReducer(over, bucket_bytes=400)          # 100-float buckets: ranges are split further
    red2.reduce_range(700, 900); red2.reduce_range(400, 700); red2.reduce_range(0, 0)
    red2.all_reduce()                                                   # covers [0,400) and [900,1000)
    ok = ok and torch.allclose(over, expect, atol=1e-6) and red2._started == [] and red2._works == []
    red2.all_reduce()                                                   # second step with nothing pre-started: one more full sum
    ok = ok and torch.allclose(over, expect * world, atol=1e-5)
    mx = parallel.max_over_ranks(1.5 + rank, torch.device("cpu"))
    lo, hi = parallel.shard_groups(5, rank, world)
    parallel.barrier()
    torch.distributed.destroy_process_group()
    out.put((rank, ok, mx, (lo, hi)))


def test_flat_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.5, 2.5]
    assert [r[3] for r in res] == [(0, 3), (3, 5)]          # 5 groups over 2 ranks: contiguous, none split


def test_shard_groups_covers_everything_once():
    from occm_amd.parallel import shard_groups
    for n in (1, 7, 8, 43):
        for w in (1, 2, 3, 8):
            cuts = [shard_groups(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
