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
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_groups(n_groups, rank, world):
    """Contiguous split of utterance groups (groups of 12 are never split: compactness_loss[:6] is per group)."""
    per, extra = divmod(n_groups, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class FlatGradAllReducer:
    """Sum-all-reduce of one flat gradient tensor in buckets of ``bucket_bytes``; the 1/world scaling is folded into
    the optimizer step (occ_adam_multi grad_scale) instead of an extra pass over the gradients."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        n = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets = [flat_grad[i:i + n] for i in range(0, flat_grad.numel(), n)]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def all_reduce(self, async_op=False):
        if self.world == 1:
            return []
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
        if not async_op:
            for w in works:
                w.wait()
            return []
        return works


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
