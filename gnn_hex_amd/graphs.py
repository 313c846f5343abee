"""HIP-graph capture of a whole training / acting step (launch-bound loops: ~25 kernel launches per step).

The step of the hot path is a fixed sequence of stream-ordered launches with static shapes (CSR build, weight pack,
fused forward, loss, fused backward, weight-gradient GEMM, reduces): nothing in it reads device results on the host,
so the whole thing is capturable.  ``GraphedStep`` warms the callable up on a side stream (first-use attribute
calls, allocator growth), captures it into a ``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) and replays it; the host
cost per step drops from ~0.7 ms of Python + launches to one ``hipGraphLaunch``.  Gradients produced inside the capture
live in the graph's private pool: ``replay()`` re-points every parameter's ``.grad`` at the buffers of THIS graph, so
several captured steps (maker batch / breaker batch) can share one model.  Collectives stay outside the graph
(``GradSync.all_reduce()`` after ``replay()``).

Caveat (PyTorch whole-network capture rule, fatal on ROCm): autograd caches one AccumulateGrad node per parameter
together with the stream it was created on, and keeps it while ANY graph that used the parameter is alive.  If a loss
tensor from an eager step on the default stream is still referenced when a step is captured, the captured backward
runs those nodes on the default (null) stream, which joins the capture, and ``hipStreamEndCapture`` crashes.  Drop such
tensors (``del loss``) or run eager steps under a side stream before constructing a ``GraphedStep``.  For the same
reason the captured callable's tensor outputs are returned DETACHED (the captured autograd graph is dropped as soon
as the capture ends, which also returns its saved buffers to the graph's pool).

ROCm 7.2 runtime bug: with the default AQL packet capture of hipGraphExec, eager kernel launches between the replays
of two instantiated graphs corrupt the second graph's kernel arguments (garbage pointers -> GPU memory fault; found
with tests/test_gpu_model.py::test_graphed_step_replays_bit_identical_gradients).  Export
``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` BEFORE the HIP runtime initialises: ``import gnn_hex_amd`` does (HIP starts lazily at
the first device call, so importing the package before touching the GPU is enough); ``GraphedStep`` refuses to run when
HIP was already up without it.
"""
from __future__ import annotations

import gc
import os
from typing import Callable, Iterable, List, Optional

import torch


def _detach(out):
    if torch.is_tensor(out):
        return out.detach()
    if isinstance(out, (tuple, list)):
        return type(out)(_detach(o) for o in out)
    return out


class GraphedStep:
    def __init__(self, fn: Callable[[], object], params: Optional[Iterable[torch.nn.Parameter]] = None,
                 warmup: int = 3, pool=None):
        import gnn_hex_amd
        if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0" or not gnn_hex_amd._graph_env_ok:
            raise RuntimeError("GraphedStep needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the HIP "
                               "runtime starts (ROCm 7.2 hipGraph packet-capture bug, see gnn_hex_amd/graphs.py): "
                               "`import gnn_hex_amd` sets it, but here HIP was already initialised without it")
        self.params: List[torch.nn.Parameter] = list(params) if params is not None else []
        gc.collect()            # unreachable autograd graphs (and their cached AccumulateGrad nodes) go now
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            out = fn()
        self.out = _detach(out)
        del out
        self.grads = [p.grad for p in self.params]

    def pool(self):
        return self.graph.pool()

    def replay(self):
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return self.out

    __call__ = replay


class GraphedSplitStep:
    """A training step captured as TWO HIP graphs split at the staged backward's hand-over (the one exchange per step
    is the gradient all-reduce, and it should travel while the backward still computes).  ``first()`` issues the step up to the
    point where the TAIL of the flat gradient buffer is final and returns ``(out, call)``; ``second(call)`` issues the rest
    (``ops.td_step(..., defer_lower=True)`` / ``ops.finish_backward``).  Between the two replays the caller starts the tail's
    all-reduce (a collective cannot be issued from inside a captured graph); it then runs on RCCL's stream beside the second
    graph's weight-gradient GEMM:

        out = step.replay_first(); sync.reduce_segment(step.flat, step.cut, step.total)
        step.replay_second();     sync.reduce_segment(step.flat, 0, step.cut); sync.all_reduce()

    Both graphs share one memory pool; the buffers the first graph allocates (gradients, workspace) are the second's inputs."""

    def __init__(self, first: Callable[[], tuple], second: Callable[[object], tuple],
                 params: Optional[Iterable[torch.nn.Parameter]] = None, warmup: int = 3, pool=None):
        self._call = None

        def fn_first():
            out, call = first()
            self._call = call
            return out

        def fn_second():
            flat, cut = second(self._call)
            self._fc = (flat, cut)
            return None

        self.g1 = GraphedStep(fn_first, params, warmup=warmup, pool=pool)
        self._fc = (None, 0)
        self.g2 = GraphedStep(fn_second, None, warmup=warmup, pool=self.g1.pool())
        self.flat, self.cut = self._fc
        self.total = int(self.flat.numel()) if self.flat is not None else 0
        self.out = self.g1.out

    def pool(self):
        return self.g1.pool()

    def replay_first(self):
        return self.g1.replay()

    def replay_second(self):
        self.g2.graph.replay()

    def replay(self):
        out = self.replay_first()
        self.replay_second()
        return out

    __call__ = replay
