"""Data-parallel gradient exchange for the shared Q-network: ONE flat bucket, ONE all-reduce per step.

The reference has no collective at all (independent processes talking through files, README.md:53;
SURVEY.md section 8e).  The MI355X-native design shards envs / replay / batches across the GPUs of a
node (one process per GPU) and exchanges only the parameter gradients: GNN-L is 486 974 fp32 = 1.95 MB,
which is latency-bound on xGMI, so everything goes into a single RCCL all-reduce (backend "nccl" is RCCL
on ROCm) instead of per-parameter buckets.  Works unchanged with the ``gloo`` backend on CPU tensors
(used by the CPU tests of the N>1 path).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradSync:
    """Flatten -> all_reduce(SUM) -> scale -> unflatten, over the parameters that received a gradient.

    Parameters whose ``.grad`` is None on this step (e.g. the head of the side not to move) contribute
    zeros and keep ``.grad is None`` only if no rank produced a gradient for them -- every rank must
    therefore process the same side per step (Env_manager keeps all envs on one side, and the replay
    shards are sampled per side), which makes the participating set identical on all ranks."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 average: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.average = average
        self._flat: Optional[torch.Tensor] = None
        self._key = None

    def _bucket(self, active: List[torch.nn.Parameter]) -> torch.Tensor:
        key = tuple(id(p) for p in active)
        total = sum(p.numel() for p in active)
        if self._flat is None or self._key != key or self._flat.numel() != total \
                or self._flat.device != active[0].device:
            self._flat = torch.empty(total, dtype=torch.float32, device=active[0].device)
            self._key = key
        return self._flat

    @staticmethod
    def _adopt_flat(active: List[torch.nn.Parameter]) -> Optional[torch.Tensor]:
        """If the gradients already are consecutive views of ONE fp32 buffer (the fused backward allocates them
        that way), return that buffer as a 1-D tensor: the all-reduce then needs no copy in or out."""
        g0 = active[0].grad
        if g0.dtype != torch.float32:
            return None
        st = g0.untyped_storage()
        base = st.data_ptr()
        off = g0.storage_offset()
        start = off
        for p in active:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base \
                    or g.storage_offset() != off:
                return None
            off += g.numel()
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, start, (off - start,))

    def all_reduce(self) -> int:
        """Sum (or average) gradients over the process group in place.  Returns the bucket size in elements."""
        active = [p for p in self.params if p.grad is not None]
        if not active:
            return 0
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return sum(p.numel() for p in active)
        flat = self._adopt_flat(active)
        if flat is not None:            # zero-copy: one collective on the buffer the backward wrote
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat.mul_(1.0 / world)
            return flat.numel()
        flat = self._bucket(active)
        off = 0
        views = []
        for p in active:
            v = flat[off:off + p.numel()].view_as(p.grad)
            views.append(v)
            off += p.numel()
        torch._foreach_copy_(views, [p.grad for p in active])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat.mul_(1.0 / world)
        torch._foreach_copy_([p.grad for p in active], views)
        return off


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of ``total`` units (envs, graphs) owned by ``rank``."""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def balance_by_edges(edge_counts, world: int):
    """Greedy partition of graphs across ranks balanced by EDGE count (ragged MIX batches).
    Returns a list of index lists, one per rank; deterministic (largest first, ties by index)."""
    order = sorted(range(len(edge_counts)), key=lambda i: (-int(edge_counts[i]), i))
    loads = [0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(edge_counts[i])
    for p in parts:
        p.sort()
    return parts
