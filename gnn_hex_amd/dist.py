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

    The bucket holds ONLY the parameters whose ``.grad`` is not None on this step (the head of the side
    not to move stays out and keeps ``.grad is None``).  Every rank must therefore train the same side and
    mode per step -- Env_manager keeps all envs on one side and the replay shards are sampled per side --
    so that the participating set is identical on all ranks; ``check=True`` verifies that with one small
    extra collective per step and raises on a mismatch instead of reducing mismatched buckets."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 average: bool = True, check: bool = False, single_rank_collectives: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.average = average
        self.check = check
        # world size 1 normally skips every collective; True issues them anyway (a 1-rank RCCL group on one GPU exercises
        # the exact call sequence of the N-rank path: async AVG all-reduce on buffer segments, waits, checks)
        self._min_world = 0 if single_rank_collectives else 1
        self._flat: Optional[torch.Tensor] = None
        self._key = None
        self._segments = []          # (data_ptr, numel, work) of the segments reduced from inside the backward

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

    # ---- overlap with the backward -----------------------------------------------------------
    def enable_overlap(self, enabled: bool = True) -> None:
        """Start the all-reduce of a gradient segment from INSIDE the fused backward, as soon as that segment's kernels
        are enqueued (gnn_hex_amd.ops.set_grad_stage_hook): the upper layers' + head's half of the flat buffer travels
        over xGMI while the lower layers' weight-gradient GEMM still computes; ``all_reduce()`` then only waits.  With
        RCCL the collective runs on the process group's own stream (async work, 1/world folded in as ReduceOp.AVG); the
        gloo rehearsal backend reduces the segment synchronously through the host (same results, no overlap)."""
        from . import ops, _lib
        ops.set_grad_stage_hook(self._on_segment if enabled else None)
        # the one-launch SAGE stack kernels need every workgroup resident at once: leave the CUs of the RCCL channels that
        # run beside the backward to them (the residency guard falls back to per-layer launches when the grid no longer fits)
        if torch.cuda.is_available():
            _lib.lib().hexgnn_stack_reserve_cus(64 if enabled else 0)

    def reduce_segment(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        """Start the all-reduce of ``flat[lo:hi]`` now (asynchronously with RCCL); ``all_reduce()`` later only waits.  The
        call a step captured as two graphs makes between its replays (graphs.GraphedSplitStep)."""
        self._on_segment(flat, lo, hi)

    def _on_segment(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world <= self._min_world or hi <= lo:
            return
        seg = flat[lo:hi]
        work = None
        if flat.is_cuda and dist.get_backend(self.group) != "gloo":
            op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
            work = dist.all_reduce(seg, op=op, group=self.group, async_op=True)
        else:
            self._reduce(seg, world)
        self._segments.append((seg.data_ptr(), seg.numel(), work))

    def _finish_segments(self, active: List[torch.nn.Parameter]) -> Optional[int]:
        """Wait for the segments reduced during the backward; returns their element count if they are exactly this
        step's gradients (one contiguous flat buffer), else None (the caller then reduces the remaining way)."""
        segs, self._segments = self._segments, []
        if not segs:
            return None
        for _, _, work in segs:
            if work is not None:
                work.wait()               # the current stream waits for the collective's stream
        flat = self._adopt_flat(active)
        covered = sum(n for _, n, _ in segs)
        lo = min(p for p, _, _ in segs)
        if flat is None or flat.data_ptr() != lo or flat.numel() != covered:
            raise RuntimeError("GradSync: gradient segments were reduced during the backward, but they are not this step's "
                               "gradient buffer (more than one backward per all_reduce()?)")
        return covered

    def all_reduce(self) -> int:
        """Sum (or average) gradients over the process group in place.  Returns the bucket size in elements."""
        active = [p for p in self.params if p.grad is not None]
        if not active:
            return 0
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world <= self._min_world:
            self._segments = []
            return sum(p.numel() for p in active)
        if self.check:
            self._check_same_set(active, world)
        done = self._finish_segments(active)
        if done is not None:
            return done
        flat = self._adopt_flat(active)
        if flat is not None:            # zero-copy: one collective on the buffer the backward wrote
            self._reduce(flat, world)
            return flat.numel()
        flat = self._bucket(active)
        off = 0
        views = []
        for p in active:
            v = flat[off:off + p.numel()].view_as(p.grad)
            views.append(v)
            off += p.numel()
        torch._foreach_copy_(views, [p.grad for p in active])
        self._reduce(flat, world)
        torch._foreach_copy_([p.grad for p in active], views)
        return off

    def _reduce(self, flat: torch.Tensor, world: int) -> None:
        """SUM all-reduce (+ 1/world) of one flat fp32 buffer, in place.  RCCL takes the device buffer as it is; the gloo
        backend (CPU tests, and rehearsals of more ranks than GPUs) is given a host copy of a device buffer."""
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                host.mul_(1.0 / world)
            flat.copy_(host)
            return
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat.mul_(1.0 / world)

    def _check_same_set(self, active: List[torch.nn.Parameter], world: int) -> None:
        """``check=True``: one tiny extra collective per step that verifies every rank reduces the SAME parameter set
        (same side to move, same mode) before the gradient bucket goes out; a mismatch would otherwise mix the maker and
        the breaker head's gradients (equal sizes) or hang on unequal counts."""
        index = {id(p): i for i, p in enumerate(self.params)}
        sig = 0
        for p in active:
            sig = (sig * 1000003 + index[id(p)] + 1) % 2147483629
        t = torch.tensor([sig, -sig, len(active), -len(active)], dtype=torch.int64)      # MAX of (v, -v): equal iff all equal
        if dist.get_backend(self.group) != "gloo":
            t = t.to(active[0].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        t = t.cpu()
        if int(t[0]) != -int(t[1]) or int(t[2]) != -int(t[3]):
            raise RuntimeError("GradSync: the ranks hold gradients for different parameter sets this step (every rank "
                               "must train the same side / mode per step)")


def launch_ranks(argv: List[str], world: int, timeout_s: Optional[float] = None) -> int:
    """Start ``world`` copies of ``argv`` (one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, rendezvous on 127.0.0.1) and wait for them.  The caller must NOT have touched the
    GPU: the ranks are fresh child processes (no exec of a process that holds a HIP context).  Rank 0 inherits stdout
    (it prints the result line), every rank inherits stderr.  If a rank fails, the others are terminated by PID and the
    failing exit code is returned."""
    import os
    import socket
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.monotonic()
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
            if rc == 0:
                rc = 124
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
    return rc


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
