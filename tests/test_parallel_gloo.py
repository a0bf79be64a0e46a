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
    assert len(red.buckets) == 4 and red.grad_scale == 1.0 / world
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
    # the overlapped and the one-shot exchange add the same ranks' values element by element: bit-identical results
    one = torch.randn(1000, generator=torch.Generator().manual_seed(100 + rank))
    parallel.FlatGradAllReducer(one, bucket_bytes=1 << 20).all_reduce()
    two = torch.randn(1000, generator=torch.Generator().manual_seed(100 + rank))
    r3 = parallel.FlatGradAllReducer(two, bucket_bytes=400)
    r3.reduce_range(900, 1000); r3.reduce_range(0, 250); r3.all_reduce()
    ok = ok and (torch.equal(one, two) if world == 2 else torch.allclose(one, two, atol=1e-5))       # (more ranks: gloo's ring order per bucket size)
    # bf16 on the wire (half the payload): equals the f32 exchange to bf16 round-off of the summands and of the sum; results land in
    # the f32 buffer; overlapped slices and the remainder are all widened back exactly once
    wb = torch.randn(1000, generator=torch.Generator().manual_seed(100 + rank))
    r4 = parallel.FlatGradAllReducer(wb, bucket_bytes=400, wire_dtype=torch.bfloat16)
    r4.reduce_range(300, 600)
    r4.all_reduce()
    ok = ok and wb.dtype == torch.float32 and r4._pending == [] and bool(((wb - expect).abs() <= 2.0 ** -7 * (expect.abs() + 4.0)).all())
    ok = ok and not torch.equal(wb, expect)                              # (it really went through bf16)
    exact = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 + k)).bfloat16() for k in range(world)).float()
    ok = ok and (torch.equal(wb, exact) if world == 2 else True)         # two ranks: bf16(a) + bf16(b) rounded once to bf16 (more ranks round per partial sum)
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


def test_flat_grad_allreduce_world4():
    """The same exchange over four ranks (the largest world the 8-core container rehearses): sums of four ranks' gradients in the f32,
    overlapped and bf16-wire forms, grad_scale 1/4, five layer groups dealt 2 / 1 / 1 / 1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True] * 4
    assert [r[2] for r in res] == [4.5] * 4
    spans = [r[3] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == 5 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all(hi > lo for lo, hi in spans)


def test_shard_groups_covers_everything_once():
    from occm_amd.parallel import shard_groups
    for n in (1, 7, 8, 43):
        for w in (1, 2, 3, 8):
            cuts = [shard_groups(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


class _FakeScorer:
    """Stands in for AModel on the CPU: (emb [B,160], logits [B,2]) as a deterministic function of each utterance."""

    def eval(self):
        return self

    def __call__(self, x):
        m, s = x.mean(dim=1, keepdim=True), x.std(dim=1, keepdim=True)
        emb = m * torch.arange(160.0)[None] + s
        return emb, torch.cat([m, s], dim=1)


class _Utts(torch.utils.data.Dataset):
    LENS = [9000, 8720, 9039, 12000, 8800, 12100, 8900, 15000, 8999, 12239, 8721]

    def __len__(self):
        return len(self.LENS)

    def __getitem__(self, i):
        return torch.randn(self.LENS[i], generator=torch.Generator().manual_seed(i)), torch.tensor([0])


def _score_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from occm_amd import parallel
    from occm_amd.oc_classifier import embed_dataset, shard_dataset
    parallel.init_from_env(backend="gloo")
    res = []
    for bs in (1, 3):
        dl = torch.utils.data.DataLoader(shard_dataset(_Utts(), rank, world), batch_size=1, shuffle=False)
        e, l = embed_dataset(_FakeScorer(), dl, "cpu", batch_size=bs, rank=rank, world=world)
        res.append((e.numpy().copy(), l.numpy().copy()))         # plain arrays: tensors in a Queue live in shared memory of a process that exits
    torch.distributed.destroy_process_group()
    out.put((rank, res))


def test_sharded_scoring_world2_equals_single_process():
    """oc_classifier.embed_dataset with the file list sharded over two ranks (gloo): every rank ends up with the embeddings of the whole
    set, in dataset order, equal to the single-process result -- one at a time and bucketed by frame count."""
    from occm_amd.oc_classifier import canonical_len, embed_dataset, n_frames
    single = {}
    for bs in (1, 3):
        dl = torch.utils.data.DataLoader(_Utts(), batch_size=1, shuffle=False)
        single[bs] = embed_dataset(_FakeScorer(), dl, "cpu", batch_size=bs)
    # bucketing crops to the canonical length, so the fake scorer (which looks at every sample) differs between bs 1 and 3 by design
    x0 = _Utts()[0][0]
    assert single[3][0].shape == (11, 160) and torch.allclose(single[3][0][0], _FakeScorer()(x0[None, :canonical_len(n_frames(x0.numel()))])[0][0])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_score_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        for k, bs in enumerate((1, 3)):
            assert torch.allclose(torch.from_numpy(got[rank][k][0]), single[bs][0], atol=1e-6) and torch.allclose(torch.from_numpy(got[rank][k][1]), single[bs][1], atol=1e-6)
