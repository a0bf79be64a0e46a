"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference uses single-process ``nn.DataParallel`` (oc_training.py:328: parameters broadcast and gradients
reduced through GPU 0 every step).  Here every rank owns a replica, shards the utterance groups and all-reduces
its FLAT gradient buffer in a few large buckets (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: fewer,
larger collectives).  BatchNorm statistics stay per rank, as under DataParallel.  Device agnostic: the same code
runs on CPU tensors over gloo in the tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:     # OCC_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("OCC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_groups(n_groups, rank, world):
    """Contiguous split of utterance groups (groups of 12 are never split: compactness_loss[:6] is per group)."""
    per, extra = divmod(n_groups, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class FlatGradAllReducer:
    """Sum-all-reduce of one flat gradient tensor in buckets of ``bucket_bytes``; the 1/world scaling is folded into
    the optimizer step (occ_adam_multi grad_scale) instead of an extra pass over the gradients.

    Overlap with backward: ``reduce_range(lo, hi)`` starts the collective for ``flat[lo:hi]`` as soon as the caller says
    those gradients are final (the XLS-R backward reports each transformer layer when it is done, last layer first);
    RCCL runs it on its own stream behind the kernels enqueued so far, under the rest of the backward pass.
    ``all_reduce()`` then covers whatever has not been started and waits for everything."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20, group=None, wire_dtype=None):
        """wire_dtype=torch.bfloat16: the gradients cross the links as bf16 (half the payload: 0.63 GB instead of 1.26 GB for XLS-R-300M)
        -- each slice is rounded into a bf16 staging buffer, summed by the collective in bf16 and widened back into the f32 buffer the
        optimizer reads (f32 master accumulation after the reduce).  Default (None): f32 on the wire, the bit-exact sum of the ranks' gradients.
        (``OCC_GRAD_WIRE=bf16`` is read by ``OcTrainer`` and applied to the XLS-R reducer only: the small back-end buffer stays f32.)"""
        self.flat = flat_grad
        self.group = group
        self.bucket = max(1, bucket_bytes // flat_grad.element_size())
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._started, self._works = [], []
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("wire_dtype must be None / float32 / bfloat16")
        self.wire = wire_dtype if (wire_dtype == torch.bfloat16 and flat_grad.dtype == torch.float32) else None
        self._stage = None                                    # bf16 mirror of the flat buffer, allocated on first use (world > 1 only)
        self._pending = []                                    # (lo, hi) slices whose bf16 sums still have to be widened back

    @property
    def buckets(self):
        n = self.bucket
        return [self.flat[i:i + n] for i in range(0, self.flat.numel(), n)]

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def _launch(self, lo, hi):
        if self.wire is not None:
            if self._stage is None:
                self._stage = torch.empty(self.flat.numel(), device=self.flat.device, dtype=self.wire)
            _cast(self.flat[lo:hi], self._stage[lo:hi])       # on the compute stream, behind the kernels that produced the slice
            self._pending.append((lo, hi))
        buf = self.flat if self.wire is None else self._stage
        mine = []
        for a in range(lo, hi, self.bucket):
            b = min(a + self.bucket, hi)
            mine.append(dist.all_reduce(buf[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._works += mine
        return mine

    def reduce_range(self, lo, hi):
        """Gradients in flat[lo:hi] are final: start their all-reduce now (no-op on one rank).  Returns the collective's work handles
        (f32 wire only; a caller that wants to consume the slice early waits on them on its own stream)."""
        if self.world == 1 or hi <= lo:
            return []
        self._started.append((int(lo), int(hi)))
        works = self._launch(int(lo), int(hi))
        return works if self.wire is None else []            # bf16 wire: the handles cover the STAGING buffer, not the f32 slice a caller would read

    def all_reduce(self, async_op=False):
        if self.world == 1:
            return []
        cur, n = 0, self.flat.numel()
        for lo, hi in sorted(self._started):                 # whatever reduce_range() has not covered
            if lo > cur:
                self._launch(cur, lo)
            cur = max(cur, hi)
        if cur < n:
            self._launch(cur, n)
        works, self._works, self._started = self._works, [], []
        if async_op and self.wire is None:
            return works
        for w in works:
            w.wait()
        pend, self._pending = self._pending, []
        for lo, hi in pend:                                  # f32 master accumulation: the optimizer reads the widened sums
            _cast(self._stage[lo:hi], self.flat[lo:hi])
        return []


def _cast(src, dst):
    """dst[:] = src with dtype conversion (f32 <-> bf16): the library's cast kernel on the GPU, a plain copy on CPU tensors (gloo tests)."""
    if src.is_cuda:
        from ._lib import check, dtype_code, lib, ptr, stream_ptr
        check(lib().occ_cast(ptr(src), dtype_code(src), ptr(dst), dtype_code(dst), src.numel(), stream_ptr()), "occ_cast")
    else:
        dst.copy_(src)


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
