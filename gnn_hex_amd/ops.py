"""Host side of the HIP path: graph-structure build + ``torch.autograd.Function`` wrappers that hand raw
device pointers and the current HIP stream to libhexgnn.so (include/hexgnn.h).

PyTorch is plumbing here: allocator, stream, autograd bookkeeping.  All arithmetic of the hot path
runs in the hand-written kernels under gnn_hex_amd/csrc; there is no eager/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_USE_EDGE_PTR = os.environ.get("HEXGNN_NO_EDGE_PTR", "") in ("", "0")      # (A/B switch: ignore the collation's edge offsets)
_DEVICE_BLOCKS = os.environ.get("HEXGNN_NO_DEVICE_BLOCKS", "") in ("", "0")  # (A/B switch: no device-built row-block tables)


def _stream() -> int:
    """Raw hipStream_t of the current torch stream (the private fast accessor when this torch has it: the public
    ``torch.cuda.current_stream()`` costs ~17 us per call, three times per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.HexGnnError(
            "%s is on %s: gnn_hex_amd runs only on the MI355X HIP path (no CPU fallback)" % (what, t.device))


def attach_hints(x: torch.Tensor, is_maker=None, max_nodes=None) -> torch.Tensor:
    """Host-known metadata of a batch (side to move, largest graph) as attributes of its feature tensor, stamped with the
    tensor's version counter: an in-place edit of ``x`` afterwards invalidates them (``hints_of`` then returns nothing and
    the model falls back to the reference's device checks)."""
    if is_maker is not None:
        x._hex_is_maker = bool(is_maker)
    if max_nodes is not None:
        x._hex_max_nodes = int(max_nodes)
    x._hex_hint_version = x._version
    return x


def hints_of(x: torch.Tensor):
    """(is_maker, max_nodes) hints of ``x`` or (None, None) when absent or stale (x modified in place since)."""
    ver = getattr(x, "_hex_hint_version", None)
    if ver is not None and ver != x._version:
        return None, None
    return getattr(x, "_hex_is_maker", None), getattr(x, "_hex_max_nodes", None)


def padded_width(hidden: int) -> int:
    hp = _lib.lib().hexgnn_padded_width(int(hidden))
    if hp < 0:
        raise _lib.HexGnnError("hidden_channels=%d not supported by the compiled kernels (1..256)" % hidden)
    return hp


def _bytes(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _ptr_array(tensors: Sequence[torch.Tensor]):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


_STICKY = {}


def sticky_status(device) -> torch.Tensor:
    """One int32 error word per device, zero unless a kernel found broken input (bits: GraphStructure.check).  The
    one-launch CSR build and the fused kernels OR into it and never clear it, so no per-batch memset is needed; the
    first ``check()`` after an error reports it and clears the word (errors are sticky across batches until then)."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _STICKY.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=dev)
        _STICKY[key] = t
    return t


def _blocks_of(edge_index, n: int):
    """The collation's row-block table of a batch (``edge_index._hex_blocks`` = (int32 device tensor [nb + 1], nb), attached by
    ``Batch.from_data_list(..., pack=True)`` / ``data.attach_blocks``): graph-aligned blocks for the one-launch stack kernels
    (``hexgnn_sage_stack_*_blocks``).  Anything malformed is ignored (the default 128-row blocks then)."""
    blk = getattr(edge_index, "_hex_blocks", None)
    if blk is None:
        return None
    try:
        t, nb = blk
        nb = int(nb)
    except (TypeError, ValueError):
        return None
    if not (torch.is_tensor(t) and t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.dim() == 1
            and t.numel() == nb + 1 and (int(n) + 127) // 128 <= nb <= min(512, max(int(n), 1))):
        return None
    return (t, nb)


class GraphStructure:
    """Target-major CSR of a batch + its transpose + 1/deg, built once per batch on the device.

    rowptr/col: in-neighbours (sources) of every node, ascending; rowptr_t/col_t: out-neighbours.
    Replaces the per-layer x[edge_index[0]] / scatter(edge_index[1]) indexing of the reference
    (GN0/models.py:276)."""

    __slots__ = ("n", "e", "rowptr", "col", "rowptr_t", "col_t", "invdeg", "status", "_ptrs", "blocks")

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, gptr: Optional[torch.Tensor] = None, b: int = 0,
                 ptr64: Optional[torch.Tensor] = None):
        """``gptr`` (int32 [b+1] node ranges) selects the single-launch build for batches whose edges are grouped by
        graph in graph order (``Batch.from_data_list``, the env builder); without it the general build runs.  With
        ``ptr64`` (the batch's int64 ``ptr``) the node ranges are read from it and ``gptr`` (preallocated) receives the
        int32 copy."""
        _require_cuda(edge_index, "edge_index")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        blocks = _blocks_of(edge_index, num_nodes)
        if edge_index.dtype != torch.int64:
            edge_index = edge_index.long()
        edge_index = edge_index.contiguous()
        dev = edge_index.device
        n, e = int(num_nodes), int(edge_index.shape[1])
        L = _lib.lib()
        self.n, self.e = n, e
        self._ptrs = None
        self.blocks = blocks
        if gptr is not None and b > 0:
            ibuf = torch.empty(2 * (n + 1), dtype=torch.int32, device=dev)
            self.rowptr, self.rowptr_t = ibuf[:n + 1], ibuf[n + 1:2 * (n + 1)]
            self.status = sticky_status(dev)
            self.col = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
            self.col_t = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
            self.invdeg = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            _lib.check(L.hexgnn_csr_build_grouped(n, e, int(b), edge_index[0].data_ptr(), edge_index[1].data_ptr(),
                                                  gptr.data_ptr() if ptr64 is None else None,
                                                  ptr64.data_ptr() if ptr64 is not None else None,
                                                  gptr.data_ptr() if ptr64 is not None else None,
                                                  self.rowptr.data_ptr(), self.col.data_ptr(),
                                                  self.rowptr_t.data_ptr(), self.col_t.data_ptr(), self.invdeg.data_ptr(),
                                                  self.status.data_ptr(), _stream()), "hexgnn_csr_build_grouped")
            return
        ws_bytes = L.hexgnn_csr_workspace_bytes(n, e)
        # one int32 buffer [rowptr | rowptr_t | status | workspace]: the build zeroes it with a single memset
        ibuf = torch.empty(2 * (n + 1) + 1 + (ws_bytes + 3) // 4, dtype=torch.int32, device=dev)
        self.rowptr = ibuf[:n + 1]
        self.rowptr_t = ibuf[n + 1:2 * (n + 1)]
        self.status = ibuf[2 * (n + 1):2 * (n + 1) + 1]
        ws = ibuf[2 * (n + 1) + 1:]
        self.col = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
        self.col_t = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
        self.invdeg = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        src = edge_index[0]
        dst = edge_index[1]
        _lib.check(L.hexgnn_csr_build(n, e, src.data_ptr(), dst.data_ptr(), self.rowptr.data_ptr(),
                                      self.col.data_ptr(), self.rowptr_t.data_ptr(), self.col_t.data_ptr(),
                                      self.invdeg.data_ptr(), self.status.data_ptr(), ws.data_ptr(), ws_bytes,
                                      _stream()), "hexgnn_csr_build")

    @classmethod
    def grouped(cls, edge_index: torch.Tensor, num_nodes: int, b: int, ptr64: torch.Tensor, pack=None, device_blocks: bool = False):
        """The one-launch build for a collated batch (edges grouped by graph, int64 ``ptr`` on the device) with ONE int32
        allocation [rowptr | rowptr_t | col | col_t | gptr] + one for 1/deg instead of six, and the raw pointers kept in
        ``_ptrs`` = (rowptr, col, rowptr_t, col_t, invdeg, gptr, status) so that the fused calls need no ``data_ptr()``: the
        eager hot path (a ``torch.empty`` costs ~2.5 us of host time, a slice ~1 us).  Returns ``(gs, gptr)``."""
        self = _GroupedStructure.__new__(_GroupedStructure)
        dev = edge_index.device
        n, e = int(num_nodes), int(edge_index.shape[1])
        self.n, self.e = n, e
        self.blocks = _blocks_of(edge_index, n)
        e1 = e if e > 0 else 1
        o1, o2 = n + 1, 2 * (n + 1)
        o3, o4 = o2 + e1, o2 + 2 * e1
        ibuf = torch.empty(o4 + b + 1, dtype=torch.int32, device=dev)
        invdeg = torch.empty(n if n > 0 else 1, dtype=torch.float32, device=dev)
        status = sticky_status(dev)
        base = ibuf.data_ptr()
        src = edge_index.data_ptr()
        self._ibuf, self._offs = ibuf, (o1, o2, o3, o4)
        self.invdeg, self.status = invdeg, status
        self._ptrs = (base, base + 4 * o2, base + 4 * o1, base + 4 * o3, invdeg.data_ptr(), base + 4 * o4, status.data_ptr())
        if pack is not None:
            # + the weight pack of the network call that follows, in the same launch: pack = (c_in, hidden, layers, wl, bl, wr
            # pointer arrays, wpack address); that call is then given no weight arrays
            # (the collation's per-graph edge offsets, when the batch carries them: Batch.from_data_list attaches them)
            ep = getattr(edge_index, "_hex_edge_ptr", None) if _USE_EDGE_PTR else None
            if ep is not None and not (torch.is_tensor(ep) and ep.dtype == torch.long and ep.device == dev and ep.is_contiguous()
                                       and ep.numel() == int(b) + 1):
                ep = None
            tbl, budget = None, 0
            if device_blocks and self.blocks is None and _DEVICE_BLOCKS:
                # the layer-major path of a batch that carries no block table (raw tensors: another collation): built on the device
                # in this launch, in the batch's own graph order (the host never sees the sizes)
                budget = stack_block_budget()
                if 0 < budget and (n + 127) // 128 <= budget:
                    tbl = torch.empty(budget + 1, dtype=torch.int32, device=dev)
                    self.blocks = (tbl, budget)
                else:
                    budget = 0
            _lib.check(_lib.lib().hexgnn_csr_build_grouped_pack_b(
                n, e, int(b), src, src + 8 * e, None, ptr64.data_ptr(), ep.data_ptr() if ep is not None else None,
                base + 4 * o4, base, base + 4 * o2, base + 4 * o1, base + 4 * o3, self._ptrs[4], self._ptrs[6], pack[0], pack[1],
                pack[2], pack[3], pack[4], pack[5], pack[6], tbl.data_ptr() if tbl is not None else None, budget, _stream()),
                "hexgnn_csr_build_grouped_pack_b")
            return self
        _lib.check(_lib.lib().hexgnn_csr_build_grouped(
            n, e, int(b), src, src + 8 * e, None, ptr64.data_ptr(), base + 4 * o4, base, base + 4 * o2, base + 4 * o1,
            base + 4 * o3, self._ptrs[4], self._ptrs[6], _stream()), "hexgnn_csr_build_grouped")
        return self

    @classmethod
    def from_csr(cls, n: int, e: int, rowptr, col, invdeg, rowptr_t=None, col_t=None) -> "GraphStructure":
        """Adopt an existing sorted CSR (the env builder emits one).  Without a transpose the graph is taken to be
        symmetric (board graphs are), i.e. its own transpose."""
        self = cls.__new__(cls)
        self.n, self.e = int(n), int(e)
        self.rowptr, self.col, self.invdeg = rowptr, col, invdeg
        self.rowptr_t = rowptr if rowptr_t is None else rowptr_t
        self.col_t = col if col_t is None else col_t
        self.status = sticky_status(rowptr.device)
        self._ptrs = None
        self.blocks = None
        return self

    def check(self) -> None:
        """Host-synchronising validity check (debug aid; not called on the hot path).  Batches built by the one-launch
        CSR build share the device's sticky error word: an error is reported by the first check after it, then cleared."""
        code = int(self.status.item())              # (synchronises)
        rc = _lib.lib().hexgnn_stack_status(1)      # a one-launch stack kernel that ran out of its poll budget since the last check
        if rc != 0:
            _lib.check(rc, "a one-launch SAGE stack kernel (its output rows were poisoned with NaN)")
        if code != 0:
            if self.status is _STICKY.get((self.status.device.type, self.status.device.index)):
                self.status.zero_()
            raise IndexError("invalid batch structure (status=%d): 1 = node id outside [0,n), 2 = graph larger than "
                             "128 nodes reached the fused kernel, 4 = an edge connects two graphs (or a batch passed as grouped is "
                             "not), 8 = graph above 2048 nodes in the grouped build, 16 = td_step: a selected node outside its graph.  This batch has n = %d; the word is shared "
                             "by every batch of this device since the last check(), so the error may stem from an EARLIER "
                             "batch (it is cleared now)" % (code, self.n))


class _GroupedStructure(GraphStructure):
    """GraphStructure.grouped(): the index arrays are views of ONE buffer, made when somebody asks for them."""
    __slots__ = ("_ibuf", "_offs")

    @property
    def rowptr(self):
        return self._ibuf[:self._offs[0]]

    @property
    def rowptr_t(self):
        return self._ibuf[self._offs[0]:self._offs[1]]

    @property
    def col(self):
        return self._ibuf[self._offs[1]:self._offs[2]]

    @property
    def col_t(self):
        return self._ibuf[self._offs[2]:self._offs[3]]

    @property
    def gptr(self):
        return self._ibuf[self._offs[3]:]


def graph_ptr(graph_indices: Optional[torch.Tensor], ptr: Optional[torch.Tensor], n: int, device):
    """(gptr int32 [B+1] on device, B).  Uses ``ptr`` when given (no host sync); otherwise derives it from
    the sorted ``graph_indices`` with B = max+1 -- the same host sync the reference has at
    GN0/models.py:576."""
    if ptr is not None:
        b = int(ptr.numel()) - 1
        return ptr.to(device=device, dtype=torch.int32), b
    if graph_indices is None:
        return torch.tensor([0, n], dtype=torch.int32, device=device), 1
    _require_cuda(graph_indices, "graph_indices")
    b = int(graph_indices.max()) + 1 if n > 0 else 0
    gi = graph_indices.long().contiguous()
    gptr = torch.empty(b + 1, dtype=torch.int32, device=device)
    _lib.check(_lib.lib().hexgnn_graph_ptr(n, b, gi.data_ptr(), gptr.data_ptr(), _stream()), "hexgnn_graph_ptr")
    return gptr, b


# ------------------------------------------------------------------------------------------------
# padded layout helpers
# ------------------------------------------------------------------------------------------------

def _is_padded_view(t: torch.Tensor, hidden: int, hp: int) -> bool:
    if t.dim() != 2 or t.shape[1] != hidden or t.dtype != torch.float32:
        return False
    if t.shape[0] > 0 and t.stride() != (hp, 1):
        return False
    need = (t.storage_offset() + t.shape[0] * hp) * 4
    return t.untyped_storage().nbytes() >= need and t.data_ptr() % 16 == 0


def as_padded(t: torch.Tensor, hidden: int, trust_pads: bool) -> torch.Tensor:
    """Return a [n, HP] tensor sharing memory with ``t`` when ``t`` already is a padded-layout view whose pad
    columns are known to be zero (tensors produced by this module carry ``_hexgnn_hp``), or whose pads do
    not matter (``trust_pads``: gradients, which the kernels mask); otherwise a zero-padded copy."""
    hp = padded_width(hidden)
    n = t.shape[0]
    tagged = getattr(t, "_hexgnn_hp", None) == hp
    if (tagged or trust_pads) and _is_padded_view(t, hidden, hp):
        return torch.as_strided(t, (n, hp), (hp, 1))
    _require_cuda(t, "feature matrix")
    src = t if (t.stride(1) == 1 and t.dtype == torch.float32) else t.float().contiguous()
    out = torch.empty((n, hp), dtype=torch.float32, device=t.device)
    _lib.check(_lib.lib().hexgnn_pad_rows(n, hidden, src.data_ptr(), src.stride(0) if n > 0 else hidden,
                                          out.data_ptr(), _stream()), "hexgnn_pad_rows")
    return out


def _logical(padded: torch.Tensor, hidden: int) -> torch.Tensor:
    v = padded[:, :hidden]
    v._hexgnn_hp = padded.shape[1]
    return v


# ------------------------------------------------------------------------------------------------
# GraphSAGE stack
# ------------------------------------------------------------------------------------------------

class SageStackFn(torch.autograd.Function):
    """y = relu(SAGE_L(... relu(SAGE_1(x)) ...)); CachifiedGNN.forward, GN0/models.py:261-294."""

    @staticmethod
    def forward(ctx, x, gs: GraphStructure, c_in: int, hidden: int, num_layers: int, flags: int, *params):
        L = _lib.lib()
        dev = x.device
        n = int(x.shape[0])
        hp = padded_width(hidden)
        small = c_in != hidden
        if small:
            if x.dtype != torch.float32 or x.stride(1) != 1:
                x = x.float().contiguous()
            xin, x_stride = x, (x.stride(0) if n > 0 else c_in)
        else:
            xin = as_padded(x, hidden, trust_pads=False)
            x_stride = hp
        params = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous() for p in params]
        wl, bl, wr = params[0::3], params[1::3], params[2::3]
        need_bwd = any(ctx.needs_input_grad)
        acts = torch.empty((num_layers, n, hp), dtype=torch.float32, device=dev)
        wpack = _bytes(L.hexgnn_sage_stack_pack_bytes(c_in, hidden, num_layers), dev)
        # (hidden > 128: the plain kernels materialise every layer's aggregate, with or without a backward)
        saved = _bytes(L.hexgnn_sage_stack_saved_bytes(n, c_in, hidden, num_layers), dev) if (need_bwd or hidden > 128) else None
        blk = gs.blocks
        _lib.check(L.hexgnn_sage_stack_forward_blocks(
            n, c_in, hidden, num_layers, gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.invdeg.data_ptr(),
            xin.data_ptr(), x_stride, _ptr_array(wl), _ptr_array(bl), _ptr_array(wr), wpack.data_ptr(),
            acts.data_ptr(), saved.data_ptr() if saved is not None else None, int(need_bwd), int(flags),
            blk[0].data_ptr() if blk else None, blk[1] if blk else 0, _stream()), "hexgnn_sage_stack_forward_blocks")
        if need_bwd:
            ctx.gs = gs
            ctx.flags = int(flags)
            ctx.dims = (n, c_in, hidden, num_layers, hp, x_stride)
            ctx.bufs = (xin, acts, saved, wpack)
            ctx.param_shapes = [p.shape for p in params]
        return acts[num_layers - 1][:, :hidden]

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.lib()
        n, c_in, hidden, num_layers, hp, x_stride = ctx.dims
        xin, acts, saved, wpack = ctx.bufs
        gs = ctx.gs
        dev = acts.device
        dy = as_padded(grad_out, hidden, trust_pads=True)
        want_dx = ctx.needs_input_grad[0] and c_in == hidden
        dx = torch.empty((n, hp), dtype=torch.float32, device=dev) if want_dx else None
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.param_shapes]
        ws_bytes = L.hexgnn_sage_stack_backward_workspace_bytes(n, c_in, hidden, num_layers)
        ws = _bytes(ws_bytes, dev)
        blk = gs.blocks
        _lib.check(L.hexgnn_sage_stack_backward_blocks(
            n, c_in, hidden, num_layers, gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.rowptr_t.data_ptr(),
            gs.col_t.data_ptr(), gs.invdeg.data_ptr(), xin.data_ptr(), x_stride, acts.data_ptr(),
            saved.data_ptr(), wpack.data_ptr(), dy.data_ptr(), dx.data_ptr() if dx is not None else None,
            _ptr_array(grads[0::3]), _ptr_array(grads[1::3]), _ptr_array(grads[2::3]), ws.data_ptr(), ws_bytes,
            ctx.flags, -1, None, blk[0].data_ptr() if blk else None, blk[1] if blk else 0, _stream()),
            "hexgnn_sage_stack_backward_blocks")
        gx = _logical(dx, hidden) if dx is not None else None
        return (gx, None, None, None, None, None) + tuple(grads)


SAGE_LINEAR_LAST = 1      # HEXGNN_SAGE_LINEAR_LAST: no ReLU after the last layer of the stack


def sage_stack(x: torch.Tensor, gs: GraphStructure, c_in: int, hidden: int, convs, linear_last: bool = False) -> torch.Tensor:
    """Run a stack of SAGEConv parameter holders (objects with lin_l.weight/bias, lin_r.weight): ReLU after every layer
    (CachifiedGNN.forward), except after the last one when ``linear_last`` (a bare SAGEConv.forward)."""
    _require_cuda(x, "x")
    params: List[torch.Tensor] = []
    for conv in convs:
        params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
    out = SageStackFn.apply(x, gs, c_in, hidden, len(convs), SAGE_LINEAR_LAST if linear_last else 0, *params)
    out._hexgnn_hp = padded_width(hidden)
    return out


def stack_block_budget(device=None) -> int:
    """Row blocks the one-launch stack kernels may use on ``device`` right now (CUs minus those reserved for kernels that run
    beside them, ``GradSync.enable_overlap``): what ``data.pack_order`` packs against."""
    if device is not None and torch.device(device).type == "cuda":
        with torch.cuda.device(device):
            return int(_lib.lib().hexgnn_stack_block_budget())
    return int(_lib.lib().hexgnn_stack_block_budget())


class live_rows:
    """``with ops.live_rows(count):`` -- inside, the whole-batch LayerNorm calls (``sage_norm_stack``, ``graph_layernorm``) take
    their row count from ``count`` (a 1-element int32 device tensor, at most the buffers' row count) instead of the tensors'
    shapes: the closed acting loop (``multi_env_manager.DeviceRollout``) hands the model capacity-sized buffers whose live
    node total exists on the device only.  Forward-only: a call that would need a backward raises."""
    _tls = threading.local()

    def __init__(self, count: Optional[torch.Tensor]):
        if count is not None and (count.dtype != torch.int32 or count.numel() != 1 or not count.is_cuda):
            raise ValueError("live_rows: a 1-element int32 device tensor")
        self.count = count

    def __enter__(self):
        self.prev = getattr(live_rows._tls, "count", None)
        live_rows._tls.count = self.count
        return self

    def __exit__(self, *exc):
        live_rows._tls.count = self.prev
        return False

    @staticmethod
    def current(*tensors) -> Optional[torch.Tensor]:
        """The count in force (called by the wrappers OUTSIDE the autograd function: there the grad mode is the caller's)."""
        c = getattr(live_rows._tls, "count", None)
        if c is not None and torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            raise RuntimeError("ops.live_rows is forward-only (acting loop): run the model under torch.no_grad()")
        return c


class SageNormStackFn(torch.autograd.Function):
    """x = relu(norm_l(SAGE_l(x))) for every layer: CachifiedGNN.forward with the LayerNorm of --norm=True (GN0/models.py:
    261-294, 935, 945) as one call per direction (``hexgnn_sage_norm_stack_*``).  params: per layer lin_l.weight, lin_l.bias,
    lin_r.weight, norm.weight, norm.bias."""

    @staticmethod
    def forward(ctx, x, gs: GraphStructure, c_in: int, hidden: int, num_layers: int, eps: float, live, *params):
        L = _lib.lib()
        dev = x.device
        n = int(x.shape[0])
        hp = padded_width(hidden)
        small = c_in != hidden
        if small:
            if x.dtype != torch.float32 or x.stride(1) != 1:
                x = x.float().contiguous()
            xin, x_stride = x, (x.stride(0) if n > 0 else c_in)
        else:
            xin = as_padded(x, hidden, trust_pads=False)
            x_stride = hp
        params = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous() for p in params]
        wl, bl, wr, nw, nb = params[0::5], params[1::5], params[2::5], params[3::5], params[4::5]
        # live: ops.live_rows' device-side row count (forward-only)
        need_bwd = any(ctx.needs_input_grad) and live is None
        # [pre | acts]: contraction outputs and norm + ReLU outputs of every layer
        both = torch.empty((2, num_layers, n, hp), dtype=torch.float32, device=dev)
        pre, acts = both[0], both[1]
        wpack = _bytes(L.hexgnn_sage_stack_pack_bytes(c_in, hidden, num_layers), dev)
        saved = _bytes(L.hexgnn_sage_stack_saved_bytes(n, c_in, hidden, num_layers), dev) if need_bwd else None
        stats = torch.empty((num_layers, 2), dtype=torch.float32, device=dev)
        nws_bytes = L.hexgnn_graph_layernorm_workspace_bytes(hidden)
        nws = _bytes(nws_bytes, dev)
        _lib.check(L.hexgnn_sage_norm_stack_forward_live(
            n, live.data_ptr() if live is not None else None, c_in, hidden, num_layers, gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.invdeg.data_ptr(), xin.data_ptr(),
            x_stride, _ptr_array(wl), _ptr_array(bl), _ptr_array(wr), _ptr_array(nw), _ptr_array(nb), float(eps),
            wpack.data_ptr(), pre.data_ptr(), acts.data_ptr(), saved.data_ptr() if saved is not None else None,
            stats.data_ptr(), nws.data_ptr(), nws_bytes, int(need_bwd), _stream()), "hexgnn_sage_norm_stack_forward_live")
        if need_bwd:
            ctx.gs = gs
            ctx.dims = (n, c_in, hidden, num_layers, hp, x_stride, float(eps))
            ctx.bufs = (xin, pre, acts, saved, wpack, stats, nw)
            ctx.param_shapes = [p.shape for p in params]
        return acts[num_layers - 1][:, :hidden]

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.lib()
        n, c_in, hidden, num_layers, hp, x_stride, eps = ctx.dims
        xin, pre, acts, saved, wpack, stats, nw = ctx.bufs
        gs = ctx.gs
        dev = acts.device
        dy = as_padded(grad_out, hidden, trust_pads=True)
        want_dx = ctx.needs_input_grad[0] and c_in == hidden
        dx = torch.empty((n, hp), dtype=torch.float32, device=dev) if want_dx else None
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.param_shapes]
        ws_bytes = L.hexgnn_sage_norm_stack_backward_workspace_bytes(n, c_in, hidden, num_layers)
        ws = _bytes(ws_bytes, dev)
        nws_bytes = L.hexgnn_graph_layernorm_workspace_bytes(hidden)
        nws = _bytes(nws_bytes, dev)
        _lib.check(L.hexgnn_sage_norm_stack_backward(
            n, c_in, hidden, num_layers, gs.rowptr_t.data_ptr(), gs.col_t.data_ptr(), gs.invdeg.data_ptr(), xin.data_ptr(),
            x_stride, pre.data_ptr(), acts.data_ptr(), saved.data_ptr(), wpack.data_ptr(), stats.data_ptr(), _ptr_array(nw),
            eps, dy.data_ptr(), dx.data_ptr() if dx is not None else None, _ptr_array(grads[0::5]), _ptr_array(grads[1::5]),
            _ptr_array(grads[2::5]), _ptr_array(grads[3::5]), _ptr_array(grads[4::5]), ws.data_ptr(), ws_bytes,
            nws.data_ptr(), nws_bytes, _stream()), "hexgnn_sage_norm_stack_backward")
        gx = _logical(dx, hidden) if dx is not None else None
        return (gx, None, None, None, None, None, None) + tuple(grads)


def sage_norm_stack(x: torch.Tensor, gs: GraphStructure, c_in: int, hidden: int, convs, norms) -> torch.Tensor:
    """conv -> LayerNorm (whole batch) -> ReLU for every layer of a stack (``--norm=True``)."""
    _require_cuda(x, "x")
    eps = float(norms[0].eps)
    params: List[torch.Tensor] = []
    for conv, norm in zip(convs, norms):
        if float(norm.eps) != eps:
            raise ValueError("the norms of one stack must share eps")
        params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight, norm.weight, norm.bias]
    out = SageNormStackFn.apply(x, gs, c_in, hidden, len(convs), eps, live_rows.current(x, *params), *params)
    out._hexgnn_hp = padded_width(hidden)
    return out


class GraphLayerNormFn(torch.autograd.Function):
    """torch_geometric LayerNorm(mode="graph") without a batch vector: normalise over ALL nodes and channels of the batch,
    affine, optional fused ReLU (GN0/models.py:286-289 norm -> act).  C ABI ``hexgnn_graph_layernorm_*``."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, relu: bool, live=None):
        L = _lib.lib()
        hidden = int(x.shape[1])
        hp = padded_width(hidden)
        n = int(x.shape[0])
        xp = as_padded(x, hidden, trust_pads=False)
        w = weight if (weight.is_contiguous() and weight.dtype == torch.float32) else weight.float().contiguous()
        b = bias if (bias.is_contiguous() and bias.dtype == torch.float32) else bias.float().contiguous()
        y = torch.empty((n, hp), dtype=torch.float32, device=x.device)
        stats = torch.empty(2, dtype=torch.float32, device=x.device)
        ws_bytes = L.hexgnn_graph_layernorm_workspace_bytes(hidden)
        ws = _bytes(ws_bytes, x.device)
        _lib.check(L.hexgnn_graph_layernorm_forward_live(n, live.data_ptr() if live is not None else None, hidden,
                                                         xp.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), int(relu),
                                                         y.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
                   "hexgnn_graph_layernorm_forward_live")
        ctx.dims = (n, hidden, hp, float(eps), bool(relu))
        ctx.bufs = (xp, y, w, stats)
        return _logical(y, hidden)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        n, hidden, hp, eps, relu = ctx.dims
        xp, y, w, stats = ctx.bufs
        dyp = as_padded(dy, hidden, trust_pads=True)
        dx = torch.empty((n, hp), dtype=torch.float32, device=xp.device)
        dw = torch.empty(hidden, dtype=torch.float32, device=xp.device)
        db = torch.empty(hidden, dtype=torch.float32, device=xp.device)
        ws_bytes = L.hexgnn_graph_layernorm_workspace_bytes(hidden)
        ws = _bytes(ws_bytes, xp.device)
        _lib.check(L.hexgnn_graph_layernorm_backward(n, hidden, xp.data_ptr(), y.data_ptr(), w.data_ptr(), stats.data_ptr(),
                                                     dyp.data_ptr(), eps, int(relu), dx.data_ptr(), dw.data_ptr(),
                                                     db.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
                   "hexgnn_graph_layernorm_backward")
        return _logical(dx, hidden), dw, db, None, None, None


def graph_layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, relu: bool = False):
    _require_cuda(x, "x")
    out = GraphLayerNormFn.apply(x, weight, bias, eps, relu, live_rows.current(x, weight, bias))
    out._hexgnn_hp = padded_width(int(x.shape[1]))
    return out


class GraphColNormFn(torch.autograd.Function):
    """CachedGraphNorm (GN0/models.py:644-670) without a batch vector: per-channel mean / variance over ALL nodes of the
    batch, ``weight * (x - mean * mean_scale) / sqrt(var + eps) + bias``, optional fused ReLU.  ``cache`` = None: fresh
    statistics (returned as the second output, [2, hidden] = mean | var, not differentiable); ``cache`` = a [2, hidden]
    tensor: the cached statistics are constants.  C ABI ``hexgnn_graph_colnorm_*``."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean_scale, eps: float, relu: bool, cache):
        L = _lib.lib()
        hidden = int(x.shape[1])
        hp = padded_width(hidden)
        n = int(x.shape[0])
        xp = as_padded(x, hidden, trust_pads=False)
        ps = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous()
              for p in (weight, bias, mean_scale)]
        y = torch.empty((n, hp), dtype=torch.float32, device=x.device)
        stats = torch.zeros((2, hp), dtype=torch.float32, device=x.device)
        use_cache = cache is not None
        if use_cache:
            stats[:, :hidden] = cache.reshape(2, hidden).to(device=x.device, dtype=torch.float32)
        ws_bytes = L.hexgnn_graph_colnorm_workspace_bytes(hidden)
        ws = _bytes(ws_bytes, x.device)
        _lib.check(L.hexgnn_graph_colnorm_forward(n, hidden, xp.data_ptr(), ps[0].data_ptr(), ps[1].data_ptr(),
                                                  ps[2].data_ptr(), float(eps), int(relu), int(use_cache), y.data_ptr(),
                                                  stats.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
                   "hexgnn_graph_colnorm_forward")
        ctx.dims = (n, hidden, hp, float(eps), bool(relu), use_cache)
        ctx.bufs = (xp, y, ps[0], ps[2], stats)
        out_stats = stats[:, :hidden]
        ctx.mark_non_differentiable(out_stats)
        return _logical(y, hidden), out_stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        L = _lib.lib()
        n, hidden, hp, eps, relu, use_cache = ctx.dims
        xp, y, w, ms, stats = ctx.bufs
        dyp = as_padded(dy, hidden, trust_pads=True)
        dx = torch.empty((n, hp), dtype=torch.float32, device=xp.device)
        dw = torch.empty(hidden, dtype=torch.float32, device=xp.device)
        db = torch.empty(hidden, dtype=torch.float32, device=xp.device)
        dms = torch.empty(hidden, dtype=torch.float32, device=xp.device)
        ws_bytes = L.hexgnn_graph_colnorm_workspace_bytes(hidden)
        ws = _bytes(ws_bytes, xp.device)
        _lib.check(L.hexgnn_graph_colnorm_backward(n, hidden, xp.data_ptr(), y.data_ptr(), w.data_ptr(), ms.data_ptr(),
                                                   stats.data_ptr(), dyp.data_ptr(), eps, int(relu), int(use_cache),
                                                   dx.data_ptr(), dw.data_ptr(), db.data_ptr(), dms.data_ptr(),
                                                   ws.data_ptr(), ws_bytes, _stream()), "hexgnn_graph_colnorm_backward")
        return _logical(dx, hidden), dw, db, dms, None, None, None


def graph_colnorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, mean_scale: torch.Tensor, eps: float = 1e-5,
                  relu: bool = False, cache: Optional[torch.Tensor] = None):
    """-> (y [n, hidden], stats [2, hidden] = mean | var as used)."""
    _require_cuda(x, "x")
    out, stats = GraphColNormFn.apply(x, weight, bias, mean_scale, eps, relu, cache)
    out._hexgnn_hp = padded_width(int(x.shape[1]))
    return out, stats


# ------------------------------------------------------------------------------------------------
# head tail (advantage linear + pooling + value MLP + dueling combine)
# ------------------------------------------------------------------------------------------------

class HeadTailFn(torch.autograd.Function):
    """mode 0: Q [n]; mode 1: (V [b], A - mean(A) [n]); mode 2: 2*tanh(adv) [n]; mode 3: raw (value [b], advantages [n])
    of HeadNetwork.forward; mode 4: raw advantages [n] only.  GN0/models.py:368-384,567-584."""

    @staticmethod
    def forward(ctx, h, gptr, b: int, hidden: int, mode: int, lin_w, lin_b, v0_w, v0_b, v1_w, v1_b):
        L = _lib.lib()
        dev = h.device
        n = int(h.shape[0])
        hp = padded_width(hidden)
        hpad = as_padded(h, hidden, trust_pads=False)
        ps = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous()
              for p in (lin_w, lin_b, v0_w, v0_b, v1_w, v1_b)]
        q = torch.empty(n, dtype=torch.float32, device=dev)
        out_v = torch.empty(b, dtype=torch.float32, device=dev) if mode in (1, 3) else None
        saved = _bytes(L.hexgnn_head_saved_bytes(n, b, hidden), dev)
        _lib.check(L.hexgnn_head_forward(
            n, b, hidden, mode, gptr.data_ptr(), hpad.data_ptr(), ps[0].data_ptr(), ps[1].data_ptr(),
            ps[2].data_ptr(), ps[3].data_ptr(), ps[4].data_ptr(), ps[5].data_ptr(), q.data_ptr(),
            out_v.data_ptr() if out_v is not None else None, saved.data_ptr(), _stream()), "hexgnn_head_forward")
        ctx.dims = (n, b, hidden, mode, hp)
        ctx.bufs = (hpad, gptr, saved, ps)
        if mode in (1, 3):
            return out_v, q
        return q

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.lib()
        n, b, hidden, mode, hp = ctx.dims
        hpad, gptr, saved, ps = ctx.bufs
        dev = hpad.device
        if mode in (1, 3):
            d_v, dq = gouts
            if d_v is None:
                d_v = torch.zeros(b, dtype=torch.float32, device=dev)
            if dq is None:
                dq = torch.zeros(n, dtype=torch.float32, device=dev)
            d_v = d_v.float().contiguous()
        else:
            (dq,) = gouts
            d_v = None
        dq = dq.float().contiguous()
        dh = torch.empty((n, hp), dtype=torch.float32, device=dev)
        g = [torch.empty_like(p) for p in ps]
        ws_bytes = L.hexgnn_head_backward_workspace_bytes(n, b, hidden)
        ws = _bytes(ws_bytes, dev)
        _lib.check(L.hexgnn_head_backward(
            n, b, hidden, mode, gptr.data_ptr(), hpad.data_ptr(), ps[0].data_ptr(), ps[2].data_ptr(),
            ps[4].data_ptr(), saved.data_ptr(), dq.data_ptr(), d_v.data_ptr() if d_v is not None else None,
            dh.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), g[4].data_ptr(),
            g[5].data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "hexgnn_head_backward")
        if mode in (2, 4):   # value path unused: no gradient for the value head (reference: grads stay None)
            g[2] = g[3] = g[4] = g[5] = None
        return (_logical(dh, hidden), None, None, None, None) + tuple(g)


class HeadLinearTailFn(torch.autograd.Function):
    """The head tail of the ``two_headed`` family: value_head_type="linear" over ("mean",) pooling (GN0/models.py:319-330,
    374-384), same modes as HeadTailFn.  C ABI ``hexgnn_head_linear_*``."""

    @staticmethod
    def forward(ctx, h, gptr, b: int, hidden: int, mode: int, lin_w, lin_b, val_w, val_b):
        L = _lib.lib()
        dev = h.device
        n = int(h.shape[0])
        hp = padded_width(hidden)
        hpad = as_padded(h, hidden, trust_pads=False)
        ps = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous()
              for p in (lin_w, lin_b, val_w, val_b)]
        q = torch.empty(n, dtype=torch.float32, device=dev)
        out_v = torch.empty(b, dtype=torch.float32, device=dev) if mode in (1, 3) else None
        saved = _bytes(L.hexgnn_head_linear_saved_bytes(n, b), dev)
        _lib.check(L.hexgnn_head_linear_forward(
            n, b, hidden, mode, gptr.data_ptr(), hpad.data_ptr(), ps[0].data_ptr(), ps[1].data_ptr(), ps[2].data_ptr(),
            ps[3].data_ptr(), q.data_ptr(), out_v.data_ptr() if out_v is not None else None, saved.data_ptr(), _stream()),
            "hexgnn_head_linear_forward")
        ctx.dims = (n, b, hidden, mode, hp)
        ctx.bufs = (hpad, gptr, saved, ps)
        if mode in (1, 3):
            return out_v, q
        return q

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.lib()
        n, b, hidden, mode, hp = ctx.dims
        hpad, gptr, saved, ps = ctx.bufs
        dev = hpad.device
        if mode in (1, 3):
            d_v, dq = gouts
            if d_v is None:
                d_v = torch.zeros(b, dtype=torch.float32, device=dev)
            if dq is None:
                dq = torch.zeros(n, dtype=torch.float32, device=dev)
            d_v = d_v.float().contiguous()
        else:
            (dq,) = gouts
            d_v = None
        dq = dq.float().contiguous()
        dh = torch.empty((n, hp), dtype=torch.float32, device=dev)
        g = [torch.empty_like(p) for p in ps]
        ws_bytes = L.hexgnn_head_linear_backward_workspace_bytes(n, b, hidden)
        ws = _bytes(ws_bytes, dev)
        _lib.check(L.hexgnn_head_linear_backward(
            n, b, hidden, mode, gptr.data_ptr(), hpad.data_ptr(), ps[0].data_ptr(), ps[2].data_ptr(), saved.data_ptr(),
            dq.data_ptr(), d_v.data_ptr() if d_v is not None else None, dh.data_ptr(), g[0].data_ptr(), g[1].data_ptr(),
            g[2].data_ptr(), g[3].data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "hexgnn_head_linear_backward")
        if mode in (2, 4):
            g[2] = g[3] = None
        return (_logical(dh, hidden), None, None, None, None) + tuple(g)


# ------------------------------------------------------------------------------------------------
# HexAra policy head pieces (GN0/torch_script_models.py:286-379)
# ------------------------------------------------------------------------------------------------

class SageScalarFn(torch.autograd.Function):
    """SAGEConv(H, 1), the last layer of the HexAra policy head: ``b + wr . h_i + mean_{j in N(i)} wl . h_j`` -> [n]."""

    @staticmethod
    def forward(ctx, h, gs: GraphStructure, hidden: int, wl, bl, wr):
        L = _lib.lib()
        dev = h.device
        n = int(h.shape[0])
        hpad = as_padded(h, hidden, trust_pads=False)
        ps = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous() for p in (wl, bl, wr)]
        out = torch.empty(n, dtype=torch.float32, device=dev)
        dots = torch.empty((n, 2), dtype=torch.float32, device=dev)
        _lib.check(L.hexgnn_sage_scalar_forward(n, hidden, gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.invdeg.data_ptr(),
                                                hpad.data_ptr(), ps[0].data_ptr(), ps[2].data_ptr(), ps[1].data_ptr(),
                                                out.data_ptr(), dots.data_ptr(), _stream()), "hexgnn_sage_scalar_forward")
        ctx.gs = gs
        ctx.dims = (n, hidden)
        ctx.bufs = (hpad, ps)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        n, hidden = ctx.dims
        hpad, ps = ctx.bufs
        gs = ctx.gs
        dev = hpad.device
        hp = padded_width(hidden)
        dout = dout.float().contiguous()
        dh = torch.empty((n, hp), dtype=torch.float32, device=dev)
        g = [torch.empty_like(p) for p in ps]
        ws_bytes = L.hexgnn_sage_scalar_backward_workspace_bytes(n, hidden)
        ws = _bytes(ws_bytes, dev)
        _lib.check(L.hexgnn_sage_scalar_backward(n, hidden, gs.rowptr_t.data_ptr(), gs.col_t.data_ptr(), gs.invdeg.data_ptr(),
                                                 hpad.data_ptr(), ps[0].data_ptr(), ps[2].data_ptr(), dout.data_ptr(),
                                                 dh.data_ptr(), g[0].data_ptr(), g[2].data_ptr(), g[1].data_ptr(),
                                                 ws.data_ptr(), ws_bytes, _stream()), "hexgnn_sage_scalar_backward")
        return _logical(dh, hidden), None, None, g[0], g[1], g[2]


def sage_scalar(h: torch.Tensor, gs: GraphStructure, hidden: int, conv) -> torch.Tensor:
    _require_cuda(h, "h")
    return SageScalarFn.apply(h, gs, hidden, conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight)


class PolicyLogSoftmaxFn(torch.autograd.Function):
    """Terminal rows dropped, swap logits inserted, scatter_log_softmax per graph (torch_script_models.py:326-378).
    -> (pi [capacity], output_graph_indices [capacity], output_batch_ptr [b+1]); the caller trims to output_batch_ptr[-1]."""

    @staticmethod
    def forward(ctx, pi_raw, should_swap, x, gptr, b: int, swap_allowed: bool):
        L = _lib.lib()
        dev = pi_raw.device
        n = int(pi_raw.shape[0])
        pi_raw = pi_raw.float().contiguous()
        ss = should_swap.float().contiguous() if should_swap is not None else None
        cap = max(n - 2 * b + (b if swap_allowed else 0), 0)
        out_pi = torch.empty(cap, dtype=torch.float32, device=dev)
        out_gi = torch.empty(cap, dtype=torch.int64, device=dev)
        out_ptr = torch.empty(b + 1, dtype=torch.int64, device=dev)
        xs = x if (x.dtype == torch.float32 and x.stride(1) == 1) else x.float().contiguous()
        _lib.check(L.hexgnn_policy_log_softmax_forward(
            n, b, gptr.data_ptr(), xs.data_ptr(), xs.stride(0) if n > 0 else 3, int(swap_allowed), pi_raw.data_ptr(),
            ss.data_ptr() if ss is not None else None, out_pi.data_ptr(), out_gi.data_ptr(), out_ptr.data_ptr(), _stream()),
            "hexgnn_policy_log_softmax_forward")
        ctx.dims = (n, b, bool(swap_allowed), cap)
        ctx.bufs = (gptr, xs, out_ptr, out_pi)
        ctx.mark_non_differentiable(out_gi, out_ptr)
        return out_pi, out_gi, out_ptr

    @staticmethod
    def backward(ctx, d_pi, _dgi, _dptr):
        L = _lib.lib()
        n, b, swap_allowed, cap = ctx.dims
        gptr, xs, out_ptr, out_pi = ctx.bufs
        dev = out_pi.device
        d_pi = d_pi.float().contiguous()
        d_raw = torch.empty(n, dtype=torch.float32, device=dev)
        d_ss = torch.empty(b, dtype=torch.float32, device=dev) if swap_allowed else None
        _lib.check(L.hexgnn_policy_log_softmax_backward(
            n, b, gptr.data_ptr(), xs.data_ptr(), xs.stride(0) if n > 0 else 3, int(swap_allowed), out_ptr.data_ptr(),
            out_pi.data_ptr(), d_pi.data_ptr(), d_raw.data_ptr(), d_ss.data_ptr() if d_ss is not None else None, _stream()),
            "hexgnn_policy_log_softmax_backward")
        return d_raw, d_ss, None, None, None, None


# ------------------------------------------------------------------------------------------------
# fused per-graph path: whole network in one launch per direction (graphs <= 128 nodes, hidden <= 112)
# ------------------------------------------------------------------------------------------------

_FUSED_ENABLED = True
_MATH = 0     # 0 = exact fp32 MFMA, 1 = split precision "f16x3" (fused kernels only)


def set_math(mode) -> None:
    """Arithmetic of the fused kernels' contractions: "fp32" (exact fp32 MFMA, default) or "f16x3" (fp32 operands
    scaled by exact powers of two and split into fp16 hi+lo pairs = 22 significand bits, three f16 MFMAs per product
    block, fp32 accumulation; ~1e-6 relative)."""
    global _MATH
    _MATH = {"fp32": 0, 0: 0, "f16x3": 1, 1: 1}[mode]


def get_math() -> str:
    return "f16x3" if _MATH == 1 else "fp32"


def set_fused(enabled: bool) -> None:
    """Enable/disable the fused per-graph kernels (tests exercise both paths)."""
    global _FUSED_ENABLED
    _FUSED_ENABLED = bool(enabled)


_GRAD_STAGE_HOOK = None


def set_grad_stage_hook(fn) -> None:
    """``fn(flat, lo, hi)`` is called INSIDE the fused backward as soon as the gradient elements ``flat[lo:hi]`` are final
    (their kernels are enqueued on the current stream) while the weight-gradient GEMM of the remaining layers is still
    to be enqueued: ``gnn_hex_amd.dist.GradSync.enable_overlap`` starts that segment's all-reduce there.
    None switches the staging off (one backward call, one reduce launch)."""
    global _GRAD_STAGE_HOOK
    _GRAD_STAGE_HOOK = fn


_SUPPORTED = {}


def qnet_fused_supported(c_in: int, hidden: int, max_nodes: int) -> bool:
    if not _FUSED_ENABLED:
        return False
    key = (c_in, hidden, max_nodes)
    ok = _SUPPORTED.get(key)
    if ok is None:
        if len(_SUPPORTED) > 4096:
            _SUPPORTED.clear()
        ok = _SUPPORTED[key] = bool(_lib.lib().hexgnn_qnet_supported(int(c_in), int(hidden), int(max_nodes)))
    return ok


class QNetFusedFn(torch.autograd.Function):
    """DuellingTwoHeaded.forward (GN0/models.py:537-584) as ONE kernel launch; backward = one data-chain launch +
    the batched weight-gradient GEMM.  Outputs: mode 0 -> (Q, embeds); 1 -> (V, A-mean(A), embeds);
    2 -> (2tanh(adv), embeds).  ``embeds`` (final_conv_acts) is returned non-differentiable; its gradient is handed
    to ``grad_sink(d_embeds)`` during backward (final_conv_grads / Grad-CAM)."""

    @staticmethod
    def forward(ctx, x, gs: GraphStructure, gptr, b: int, c_in: int, hidden: int, body_layers: int,
                head_layers: int, mode: int, grad_sink, *params):
        L = _lib.lib()
        ctx.set_materialize_grads(False)      # unused outputs (embeds, V) arrive as None instead of freshly zeroed tensors
        dev = x.device
        n = int(x.shape[0])
        hp = padded_width(hidden)
        tot = body_layers + head_layers
        if x.dtype != torch.float32 or x.stride(1) != 1:
            x = x.float().contiguous()
        x_stride = x.stride(0) if n > 0 else c_in
        # a parameter that is computed per forward (FactorizedNoisyLinear's mu + sigma * eps) is no leaf: its gradient is
        # an intermediate that autograd still has to carry to mu / sigma, so it must not be all-reduced in place
        ctx.nonleaf_params = any(p.grad_fn is not None for p in params)
        params = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.float().contiguous() for p in params]
        convs, tail = params[:3 * tot], params[3 * tot:]
        wl, bl, wr = convs[0::3], convs[1::3], convs[2::3]
        need_bwd = any(ctx.needs_input_grad)
        acts = torch.empty((tot, n, hp), dtype=torch.float32, device=dev)
        wpack = _bytes(L.hexgnn_sage_stack_pack_bytes(c_in, hidden, tot), dev)
        saved = _bytes(L.hexgnn_qnet_saved_bytes(n, b, c_in, hidden, tot), dev)
        q = torch.empty(n, dtype=torch.float32, device=dev)
        out_v = torch.empty(b, dtype=torch.float32, device=dev) if mode == 1 else None
        status = gs.status      # shared status word (OR-ed into by the kernels)
        _lib.check(L.hexgnn_qnet_forward(
            n, b, c_in, hidden, tot, mode, gptr.data_ptr(), gs.rowptr.data_ptr(), gs.col.data_ptr(),
            gs.invdeg.data_ptr(), x.data_ptr(), x_stride, _ptr_array(wl), _ptr_array(bl), _ptr_array(wr),
            tail[0].data_ptr(), tail[1].data_ptr(), tail[2].data_ptr(), tail[3].data_ptr(), tail[4].data_ptr(),
            tail[5].data_ptr(), wpack.data_ptr(), acts.data_ptr(), saved.data_ptr(), int(need_bwd), body_layers - 1, _MATH,
            q.data_ptr(),
            out_v.data_ptr() if out_v is not None else None, status.data_ptr(), _stream()), "hexgnn_qnet_forward")
        embeds = acts[body_layers - 1][:, :hidden]
        ctx.mark_non_differentiable(embeds)
        if need_bwd:
            ctx.gs, ctx.gptr = gs, gptr
            ctx.dims = (n, b, c_in, hidden, tot, body_layers, mode, hp, x_stride)
            ctx.math = _MATH
            ctx.bufs = (x, acts, saved, wpack, tail, status)
            ctx.params = params
            # flat gradient layout follows model.parameters() order (convs..., value_head.layers.*, linear.*) so that
            # GradSync can all-reduce the buffer in place; the call's tail order is (lin_w, lin_b, v0_w, v0_b, v1_w, v1_b)
            nconv = 3 * tot
            flat_order = list(range(nconv)) + [nconv + 2, nconv + 3, nconv + 4, nconv + 5, nconv + 0, nconv + 1]
            offs, o = [0] * len(params), 0
            for i in flat_order:
                offs[i] = o
                o += params[i].numel()
            ctx.param_offsets = (offs, o, flat_order)
            ctx.grad_sink = grad_sink
        if mode == 1:
            return out_v, q, embeds
        return q, embeds

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.lib()
        n, b, c_in, hidden, tot, body_layers, mode, hp, x_stride = ctx.dims
        x, acts, saved, wpack, tail, status = ctx.bufs
        gs, gptr = ctx.gs, ctx.gptr
        dev = acts.device
        if mode == 1:
            d_v, dq = gouts[0], gouts[1]
            d_v = torch.zeros(b, dtype=torch.float32, device=dev) if d_v is None else d_v.float().contiguous()
        else:
            dq, d_v = gouts[0], None
        dq = torch.zeros(n, dtype=torch.float32, device=dev) if dq is None else dq.float().contiguous()
        # ONE flat gradient buffer for all parameters (views are handed to autograd): one allocation instead of
        # 66, and the layout a single RCCL all-reduce wants (gnn_hex_amd.dist.GradSync adopts it without copies).
        params = ctx.params
        offs, total, flat_order = ctx.param_offsets
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        ordered = torch._C._nn.unflatten_dense_tensors(flat, [params[i] for i in flat_order])
        grads = [None] * len(params)
        for i, gview in zip(flat_order, ordered):
            grads[i] = gview
        base = flat.data_ptr()
        ptrs = [base + 4 * o for o in offs]
        cp, tp = ptrs[:3 * tot], ptrs[3 * tot:]
        vp_arr = C.c_void_p * tot
        d_emb = torch.empty((n, hp), dtype=torch.float32, device=dev) if ctx.grad_sink is not None else None
        ws_bytes = L.hexgnn_qnet_backward_workspace_bytes(n, b, c_in, hidden, tot)
        ws = _bytes(ws_bytes, dev)
        common = (n, b, c_in, hidden, tot, body_layers, mode, ctx.math, gptr.data_ptr(), gs.rowptr_t.data_ptr(),
                  gs.col_t.data_ptr(), gs.invdeg.data_ptr(), x.data_ptr(), x_stride, acts.data_ptr(), saved.data_ptr(),
                  wpack.data_ptr(), tail[0].data_ptr(), tail[2].data_ptr(), tail[4].data_ptr(), dq.data_ptr(),
                  d_v.data_ptr() if d_v is not None else None, d_emb.data_ptr() if d_emb is not None else None,
                  vp_arr(*cp[0::3]), vp_arr(*cp[1::3]), vp_arr(*cp[2::3]), tp[0], tp[1], tp[2], tp[3], tp[4], tp[5],
                  ws.data_ptr(), ws_bytes, status.data_ptr())
        hook = _GRAD_STAGE_HOOK
        if hook is None or tot < 3 or mode == 2 or ctx.nonleaf_params:
            # (non-leaf parameters, --noisy_dqn=True: d_sigma = d_w * eps with per-rank noise is not the average of the ranks'
            # d_sigma if d_w is averaged first, and the mu / sigma gradients live outside the flat buffer: no staging, GradSync
            # reduces the finished .grad tensors through its bucket)
            _lib.check(L.hexgnn_qnet_backward(*common, _stream()), "hexgnn_qnet_backward")
        else:
            # two stages: the upper half of the hidden layers + everything small first -- with the head tail they are the
            # TAIL of the flat buffer, handed to the hook (all-reduce on the collective's own stream) while the lower
            # half's weight-gradient GEMM is enqueued behind them; then the head of the buffer
            mid = 1 + tot // 2
            _lib.check(L.hexgnn_qnet_backward_staged(*common, 1 | 2 | 4, mid, tot, _stream()), "hexgnn_qnet_backward_staged")
            hook(flat, offs[3 * mid], total)
            _lib.check(L.hexgnn_qnet_backward_staged(*common, 4, 1, mid, _stream()), "hexgnn_qnet_backward_staged")
            hook(flat, 0, offs[3 * mid])
        cg, tg = grads[:3 * tot], grads[3 * tot:]
        if ctx.grad_sink is not None:
            ctx.grad_sink(d_emb[:, :hidden])
        if mode == 2:
            tg[2] = tg[3] = tg[4] = tg[5] = None
        return (None,) * 10 + tuple(cg) + tuple(tg)


# ------------------------------------------------------------------------------------------------
# direct-gradient form of the fused path: the eager step an unmodified train.py issues
# ------------------------------------------------------------------------------------------------
# QNetFusedFn hands 57 parameter tensors to autograd and receives 57 gradients back: ~0.45 ms of Python + autograd per step
# (pointer arrays rebuilt, 66 AccumulateGrad nodes, six allocations), which left GNN-S host-bound at 0.8 M graphs/s against
# 2 M replayed.  Here the parameters are NOT autograd inputs: the function has ONE differentiable input (a per-model
# anchor), the forward call uses pointer arrays cached per (model, head) -- re-validated by data_ptr every call, 57 C-level
# calls -- and the backward writes every gradient into ONE flat buffer (``hexgnn_qnet_backward_flat``: base pointer + cached
# offset table) and assigns the views to ``p.grad`` itself (accumulating when a gradient is already there, like
# AccumulateGrad).  Same kernels, same bits.  Not usable with ``torch.autograd.grad(loss, parameters)`` or tensor hooks on
# parameters (the model falls back to QNetFusedFn when it finds hooks, frozen or non-leaf parameters);
# ``set_direct_grads(False)`` switches it off.
_DIRECT_GRADS = True


def set_direct_grads(enabled: bool) -> None:
    global _DIRECT_GRADS
    _DIRECT_GRADS = bool(enabled)


_FREE_RC = None


def _free_refcount() -> int:
    """What ``sys.getrefcount(v)`` returns inside ``for v in views`` for an object that only the tuple ``views`` holds (3 on
    CPython 3.10: the tuple, the loop variable, the call's argument) -- measured, not assumed."""
    global _FREE_RC
    if _FREE_RC is None:
        probe = (object(),)
        for v in probe:
            _FREE_RC = sys.getrefcount(v)
    return _FREE_RC


class QNetParamCache:
    """Per (model, head): the parameter list of the fused call, its cached pointer arrays and the flat gradient layout."""
    __slots__ = ("params", "ptrs", "wl", "bl", "wr", "tail", "flat_params", "offsets", "total", "direct_ok", "tot",
                 "sizes", "cut", "grad_ring")

    def __init__(self, params, tot):
        self.params = params
        self.tot = tot
        self.ptrs = None
        self.sizes = {}          # (n, b) -> buffer sizes of the fused calls (three C queries per new batch shape)
        self.grad_ring = []      # [(flat gradient buffer, its per-parameter views, stream)]: see grad_buffer()
        self.refresh()

    def grad_buffer(self, dev):
        """The flat buffer the backward writes and its per-parameter views.  A training loop drops its gradients every step
        (``zero_grad(set_to_none=True)`` / ``p.grad = None``) and the next backward allocates them again: 60 views through
        ``unflatten_dense_tensors`` cost 25 us per step -- a sixth of the host time of an eager GNN-S step.  A buffer of an
        earlier backward is handed out again when NOBODY references any of its views any more (Python reference counts: the
        ring's own tuple only), which is exactly when autograd's freshly allocated gradients would be indistinguishable
        from it; a view still held anywhere (``p.grad`` not cleared, a list of gradients kept for logging) leaves that buffer
        alone (``Tensor._use_count()`` sees the holders Python reference counts do not: ``.grad`` itself).  Not inside a HIP-graph capture (those gradients must come from the graph's pool)."""
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(self.total, dtype=torch.float32, device=dev), None
        rc = sys.getrefcount
        free_rc = _free_refcount()
        st = _stream()
        for flat, views, owner in self.grad_ring:
            if owner == st and flat.device == dev:      # (same stream only: the caching allocator's rule for a freed block)
                for v in views:
                    # Python side: the tuple, the loop variable, the argument (measured once, _free_refcount); C++ side (a .grad, a
                    # saved tensor): the wrapper only
                    if rc(v) != free_rc or v._use_count() != 1:
                        break
                else:
                    return flat, views
        flat = torch.empty(self.total, dtype=torch.float32, device=dev)
        views = tuple(torch._C._nn.unflatten_dense_tensors(flat, self.flat_params))
        if len(self.grad_ring) >= 3:
            self.grad_ring.pop(0)
        self.grad_ring.append((flat, views, st))
        return flat, views

    def refresh(self):
        params, tot = self.params, self.tot
        self.ptrs = list(map(torch.Tensor.data_ptr, params))
        vp = C.c_void_p * tot
        pt = self.ptrs
        self.wl, self.bl, self.wr = vp(*pt[0:3 * tot:3]), vp(*pt[1:3 * tot:3]), vp(*pt[2:3 * tot:3])
        self.tail = tuple(pt[3 * tot:])
        nconv = 3 * tot
        order = list(range(nconv)) + [nconv + 2, nconv + 3, nconv + 4, nconv + 5, nconv + 0, nconv + 1]   # parameters() order
        offs, o = [0] * len(params), 0
        for i in order:
            offs[i] = o
            o += params[i].numel()
        self.flat_params = [params[i] for i in order]
        self.offsets = (C.c_int64 * len(params))(*offs)
        self.total = o
        self.grad_ring = []
        self.cut = offs[3 * (1 + tot // 2)] if tot >= 3 else 0      # flat position where the staged backward splits
        # (post-accumulate-grad hooks -- optimizer-in-backward, FSDP-style reducers -- hang off AccumulateGrad, which the direct
        # path never runs: such parameters take the autograd form, like tensor hooks do)
        self.direct_ok = all(p.is_leaf and p.requires_grad and p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda
                             and not p._backward_hooks and not getattr(p, "_post_accumulate_grad_hooks", None) for p in params)

    def valid(self) -> bool:
        """Pointers unchanged (parameters updated in place keep them; ``.to()`` / ``p.data = ...`` do not) and the flags
        ``direct_ok`` was derived from still hold (a parameter frozen or given a tensor hook since)."""
        ps = self.params
        return list(map(torch.Tensor.data_ptr, ps)) == self.ptrs and \
            self.direct_ok == all(p.requires_grad and not p._backward_hooks and
                                  not getattr(p, "_post_accumulate_grad_hooks", None) for p in ps)


class _QNetCall:
    """Everything one fused forward leaves behind for its backward (plain attributes: cheaper than ctx.save_for_backward)."""
    __slots__ = ("cache", "gs", "gptr", "dims", "x", "bufs", "math", "sink", "gp", "done", "layered", "td", "pending", "versions")


_HP_CACHE = {}


def qnet_direct_forward(cache: QNetParamCache, x, gs: GraphStructure, gptr, b: int, c_in: int, hidden: int, body_layers: int,
                        head_layers: int, mode: int, need_bwd: bool, layered: bool = False):
    """Launch the fused forward with cached pointer arrays; returns (q, out_v, call) -- ``call`` feeds QNetDirectFn.
    ``gptr`` may be None when ``gs`` comes from ``GraphStructure.grouped`` (it carries the pointer).  ``layered``: the same
    network on the layer-major kernels (graphs above 128 nodes, hidden 113..128): body and head SAGE layers as ONE stack +
    the head-tail kernels, see ``qnet_layered_forward``."""
    if layered:
        return qnet_layered_forward(cache, x, gs, gptr, b, c_in, hidden, body_layers, head_layers, mode, need_bwd)
    L = _lib.lib()
    dev = x.device
    n = x.shape[0]
    tot = body_layers + head_layers
    if x.dtype != torch.float32 or x.stride(1) != 1:
        x = x.float().contiguous()
    x_stride = x.stride(0) if n > 0 else c_in
    sizes = cache.sizes.get((n, b))
    if sizes is None:
        hp = padded_width(hidden)
        a_bytes = (4 * tot * n * hp + 255) & ~255
        w_bytes = (L.hexgnn_sage_stack_pack_bytes(c_in, hidden, tot) + 255) & ~255
        s_bytes = L.hexgnn_qnet_saved_bytes(n, b, c_in, hidden, tot)
        ws_bytes = L.hexgnn_qnet_backward_workspace_bytes(n, b, c_in, hidden, tot)
        if len(cache.sizes) > 64:
            cache.sizes.clear()
        sizes = cache.sizes[(n, b)] = (hp, a_bytes, w_bytes, max(s_bytes, 16), max(ws_bytes, 16))
    hp, a_bytes, w_bytes, s_bytes, ws_bytes = sizes
    # [acts | wpack | saved] in ONE allocation (256-byte aligned parts)
    buf = torch.empty(a_bytes + w_bytes + s_bytes, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    q = torch.empty(n, dtype=torch.float32, device=dev)
    out_v = torch.empty(b, dtype=torch.float32, device=dev) if mode == 1 else None
    wl, bl, wr = cache.wl, cache.bl, cache.wr
    if type(gs) is tuple:       # deferred grouped build (models.py): CSR + weight pack in one launch, now that wpack exists
        if _MATH == 0:
            gs = GraphStructure.grouped(gs[0], gs[1], gs[2], gs[3], pack=(c_in, hidden, tot, wl, bl, wr, base + a_bytes))
            wl = bl = wr = None
        else:
            gs = GraphStructure.grouped(*gs)
    gp = gs._ptrs
    if gp is None:
        gp = (gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.rowptr_t.data_ptr(), gs.col_t.data_ptr(), gs.invdeg.data_ptr(),
              gptr.data_ptr(), gs.status.data_ptr())
    t = cache.tail
    stream = _stream()
    td = None
    tda = getattr(_TD_STEP, "args", None)
    if tda is not None and mode == 0 and need_bwd and b > 0 and n > 0:
        # td_step(): the update's loss is formed in the forward kernel's tail (one selected node per graph)
        sel, tgt, w, lfn = tda
        if sel.numel() == b:
            # one buffer: dq [n] | td [b] | loss terms [b] | loss [1]
            tb = torch.empty(n + 2 * b + 1, dtype=torch.float32, device=dev)
            tp = tb.data_ptr()
            _lib.check(L.hexgnn_qnet_forward_td(
                n, b, c_in, hidden, tot, gp[5], gp[0], gp[1], gp[4], x.data_ptr(), x_stride, wl, bl, wr,
                t[0], t[1], t[2], t[3], t[4], t[5], base + a_bytes, base, base + a_bytes + w_bytes, _MATH, q.data_ptr(), gp[6],
                sel.data_ptr(), tgt.data_ptr(), w.data_ptr() if w is not None else None, lfn, tp, tp + 4 * n,
                tp + 4 * (n + b), stream), "hexgnn_qnet_forward_td")
            td = (tb, n, b)
    if td is None:
        _lib.check(L.hexgnn_qnet_forward(
            n, b, c_in, hidden, tot, mode, gp[5], gp[0], gp[1], gp[4], x.data_ptr(), x_stride, wl, bl, wr,
            t[0], t[1], t[2], t[3], t[4], t[5], base + a_bytes, base, base + a_bytes + w_bytes, int(need_bwd), body_layers - 1,
            _MATH, q.data_ptr(), out_v.data_ptr() if out_v is not None else None, gp[6], stream), "hexgnn_qnet_forward")
    call = _QNetCall()
    call.cache, call.gs, call.gptr, call.x = cache, gs, gptr, x
    call.dims = (n, b, c_in, hidden, tot, body_layers, mode, hp, x_stride, a_bytes, w_bytes, ws_bytes)
    call.bufs, call.math, call.gp = buf, _MATH, gp
    call.sink = None
    call.done = False
    call.layered = False
    call.td = td
    # the backward reads the head tail's weights LIVE (the SAGE layers' from the pack made by this forward): an in-place update
    # between the two would mix old and new weights without autograd's saved-tensor version check to notice it
    call.versions = tuple(p._version for p in cache.params[-6:]) if need_bwd else None
    return q, out_v, call


def qnet_layered_forward(cache: QNetParamCache, x, gs: GraphStructure, gptr, b: int, c_in: int, hidden: int, body_layers: int,
                         head_layers: int, mode: int, need_bwd: bool):
    """DuellingTwoHeaded.forward on the layer-major kernels with the direct-gradient bookkeeping: body + head SAGE layers run
    as ONE stack of body_layers + head_layers layers (``hexgnn_sage_stack_forward``; the head's gnn is just more SAGE layers
    with ReLU), then the head tail (``hexgnn_head_forward``).  Against the per-module composition (body stack, head stack,
    head tail as three autograd functions) the backward has ONE batched weight-gradient GEMM + ONE slab reduce over all 16
    hidden layers instead of two of each, no stack-boundary combine, and ~1 ms less Python per step."""
    L = _lib.lib()
    dev = x.device
    n = x.shape[0]
    tot = body_layers + head_layers
    if x.dtype != torch.float32 or x.stride(1) != 1:
        x = x.float().contiguous()
    x_stride = x.stride(0) if n > 0 else c_in
    sizes = cache.sizes.get(("L", n, b))
    if sizes is None:
        hp = padded_width(hidden)
        al = lambda v: (max(int(v), 16) + 255) & ~255
        sizes = cache.sizes[("L", n, b)] = (
            hp, al(4 * tot * n * hp), al(L.hexgnn_sage_stack_pack_bytes(c_in, hidden, tot)),
            al(L.hexgnn_sage_stack_saved_bytes(n, c_in, hidden, tot)), al(L.hexgnn_head_saved_bytes(n, b, hidden)),
            al(L.hexgnn_sage_stack_backward_workspace_bytes(n, c_in, hidden, tot)),
            al(L.hexgnn_head_backward_workspace_bytes(n, b, hidden)))
    hp, a_bytes, w_bytes, s_bytes, hs_bytes, ws_bytes, hws_bytes = sizes
    buf = torch.empty(a_bytes + w_bytes + s_bytes + hs_bytes, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    q = torch.empty(n, dtype=torch.float32, device=dev)
    out_v = torch.empty(b, dtype=torch.float32, device=dev) if mode == 1 else None
    wl, bl, wr = cache.wl, cache.bl, cache.wr
    if type(gs) is tuple:       # deferred grouped build: CSR + weight pack in one launch (hidden <= 128: the packed layout)
        if hidden <= 128:
            gs = GraphStructure.grouped(gs[0], gs[1], gs[2], gs[3], pack=(c_in, hidden, tot, wl, bl, wr, base + a_bytes),
                                        device_blocks=True)
            wl = bl = wr = None
        else:
            gs = GraphStructure.grouped(*gs)
    gp = gs._ptrs
    if gp is None:
        gp = (gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.rowptr_t.data_ptr(), gs.col_t.data_ptr(), gs.invdeg.data_ptr(),
              gptr.data_ptr(), gs.status.data_ptr())
    t = cache.tail
    stream = _stream()
    blk = gs.blocks
    _lib.check(L.hexgnn_sage_stack_forward_blocks(n, c_in, hidden, tot, gp[0], gp[1], gp[4], x.data_ptr(), x_stride, wl, bl,
                                                  wr, base + a_bytes, base, base + a_bytes + w_bytes, int(need_bwd), 0,
                                                  blk[0].data_ptr() if blk else None, blk[1] if blk else 0, stream),
               "hexgnn_sage_stack_forward_blocks")
    h_top = base + 4 * (tot - 1) * n * hp
    _lib.check(L.hexgnn_head_forward(n, b, hidden, mode, gp[5], h_top, t[0], t[1], t[2], t[3], t[4], t[5], q.data_ptr(),
                                     out_v.data_ptr() if out_v is not None else None, base + a_bytes + w_bytes + s_bytes,
                                     stream), "hexgnn_head_forward")
    call = _QNetCall()
    call.cache, call.gs, call.gptr, call.x = cache, gs, gptr, x
    call.dims = (n, b, c_in, hidden, tot, body_layers, mode, hp, x_stride, a_bytes, w_bytes, (s_bytes, ws_bytes, hws_bytes))
    call.bufs, call.math, call.gp = buf, 0, gp
    call.sink = None
    call.done = False
    call.layered = True
    call.td = None
    call.versions = tuple(p._version for p in cache.params[-6:]) if need_bwd else None
    return q, out_v, call


def _assign_flat_grads(cache: QNetParamCache, flat: torch.Tensor, mode: int, views=None) -> None:
    """Views of the flat gradient buffer -> ``p.grad`` (accumulating into a gradient that is already there)."""
    fp = cache.flat_params
    if views is None:
        views = torch._C._nn.unflatten_dense_tensors(flat, fp)
    skip = mode == 2          # advantages only: the value head (flat positions -6 .. -3) has no gradient
    k_lo, k_hi = len(fp) - 6, len(fp) - 2
    for k, (p, v) in enumerate(zip(fp, views)):
        if skip and k_lo <= k < k_hi:
            continue
        g = p.grad
        p.grad = v if g is None else g + v


def qnet_layered_backward(call: "_QNetCall", dq, d_v=None) -> None:
    L = _lib.lib()
    cache = call.cache
    n, b, c_in, hidden, tot, body_layers, mode, hp, x_stride, a_bytes, w_bytes, (s_bytes, ws_bytes, hws_bytes) = call.dims
    dev = call.x.device
    if mode == 1:
        d_v = torch.zeros(b, dtype=torch.float32, device=dev) if d_v is None else d_v.float().contiguous()
    else:
        d_v = None
    dq = torch.zeros(n, dtype=torch.float32, device=dev) if dq is None else \
        (dq if (dq.dtype == torch.float32 and dq.is_contiguous()) else dq.float().contiguous())
    flat, gviews = cache.grad_buffer(dev)
    ws = torch.empty(ws_bytes + hws_bytes, dtype=torch.uint8, device=dev)
    # the head tail writes dh * [h > 0] = G of the top layer straight into its slab of the stack's workspace
    # (HEXGNN_HEAD_MASK_DH / HEXGNN_SAGE_DY_IN_PLACE: no masked copy in between)
    dh_ptr = ws.data_ptr() + 4 * (tot - 1) * n * hp
    d_emb = torch.empty((n, hp), dtype=torch.float32, device=dev) if (call.sink is not None and body_layers < tot) else None
    base, gp, t = call.bufs.data_ptr(), call.gp, cache.tail
    fb = flat.data_ptr()
    offs = cache.offsets
    tp = [fb + 4 * offs[3 * tot + k] for k in range(6)]          # d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w, d_v1_b
    vh = mode != 2
    stream = _stream()
    h_top = base + 4 * (tot - 1) * n * hp
    _lib.check(L.hexgnn_head_backward(n, b, hidden, mode | 8, gp[5], h_top, t[0], t[2], t[4], base + a_bytes + w_bytes + s_bytes,
                                      dq.data_ptr(), d_v.data_ptr() if d_v is not None else None, dh_ptr, tp[0], tp[1],
                                      tp[2] if vh else None, tp[3] if vh else None, tp[4] if vh else None,
                                      tp[5] if vh else None, ws.data_ptr() + ws_bytes, hws_bytes, stream),
               "hexgnn_head_backward")
    vp = C.c_void_p * tot
    d_wl = vp(*[fb + 4 * offs[3 * l] for l in range(tot)])
    d_bl = vp(*[fb + 4 * offs[3 * l + 1] for l in range(tot)])
    d_wr = vp(*[fb + 4 * offs[3 * l + 2] for l in range(tot)])
    blk = call.gs.blocks
    _lib.check(L.hexgnn_sage_stack_backward_blocks(
        n, c_in, hidden, tot, gp[0], gp[1], gp[2], gp[3], gp[4], call.x.data_ptr(), x_stride, base, base + a_bytes + w_bytes,
        base + a_bytes, dh_ptr, None, d_wl, d_bl, d_wr, ws.data_ptr(), ws_bytes, 2,
        body_layers - 1 if d_emb is not None else -1, d_emb.data_ptr() if d_emb is not None else None,
        blk[0].data_ptr() if blk else None, blk[1] if blk else 0, stream), "hexgnn_sage_stack_backward_blocks")
    _assign_flat_grads(cache, flat, mode, gviews)
    del gviews
    if call.sink is not None and d_emb is not None:
        call.sink(d_emb[:, :hidden])


def qnet_embeds(call: _QNetCall) -> torch.Tensor:
    """final_conv_acts of a direct forward: the body output as a [n, hidden] view of the activation slab."""
    n, b, c_in, hidden, tot, body_layers, mode, hp = call.dims[:8]
    acts = call.bufs[:4 * tot * n * hp].view(torch.float32).view(tot, n, hp)
    return acts[body_layers - 1][:, :hidden]


class QNetDirectFn(torch.autograd.Function):
    """Autograd node of a direct forward: ONE differentiable input (the per-model anchor, whose gradient is never produced);
    ``fargs`` = the arguments of ``qnet_direct_forward``, ``holder`` = a list that receives the call record.  The backward
    runs the fused backward kernels and assigns the parameter gradients (see the section comment)."""

    @staticmethod
    def forward(ctx, anchor, holder, fargs):
        ctx.set_materialize_grads(False)
        q, out_v, call = qnet_direct_forward(*fargs)
        ctx.call = call
        holder.append(call)
        if out_v is None:
            return q
        return q, out_v

    @staticmethod
    def backward(ctx, dq, d_v=None):
        qnet_direct_backward(ctx.call, dq, d_v)
        return None, None, None


def qnet_direct_backward(call: "_QNetCall", dq, d_v=None, defer_lower: bool = False) -> None:
    """The fused backward of a direct forward: every parameter gradient into ONE flat buffer, views assigned to ``p.grad``
    (accumulated when a gradient is already there).  Called by autograd (QNetDirectFn.backward, on the engine's device
    thread) or straight from ``ops.backward(loss)`` on the caller's thread.

    ``defer_lower``: only the first stage of the staged backward runs (data chain, small reduces, weight gradients of the upper
    half of the hidden layers: the TAIL of the flat buffer is final); ``finish_backward(call)`` runs the rest.  A step captured
    as two HIP graphs split there lets the tail's all-reduce travel while the second graph computes (graphs.GraphedSplitStep)."""
    call.pending = None
    if call.versions is not None and call.versions != tuple(p._version for p in call.cache.params[-6:]):
        raise RuntimeError("a head parameter of the model was modified in place between this forward and its backward (the "
                           "fused backward reads the head tail's weights live); run the backward before optimizer.step()")
    if call.done:
        raise RuntimeError("this forward's backward already ran through ops.backward(loss) (its graph is spent, as after "
                           "loss.backward() without retain_graph)")
    if call.layered:
        return qnet_layered_backward(call, dq, d_v)
    L = _lib.lib()
    cache = call.cache
    n, b, c_in, hidden, tot, body_layers, mode, hp, x_stride, a_bytes, w_bytes, ws_bytes = call.dims
    dev = call.x.device
    if mode == 1:
        d_v = torch.zeros(b, dtype=torch.float32, device=dev) if d_v is None else d_v.float().contiguous()
    else:
        d_v = None
    dq = torch.zeros(n, dtype=torch.float32, device=dev) if dq is None else \
        (dq if (dq.dtype == torch.float32 and dq.is_contiguous()) else dq.float().contiguous())
    flat, gviews = cache.grad_buffer(dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d_emb = torch.empty((n, hp), dtype=torch.float32, device=dev) if call.sink is not None else None
    base = call.bufs.data_ptr()
    gp, t = call.gp, cache.tail
    common = (n, b, c_in, hidden, tot, body_layers, mode, call.math, gp[5], gp[2], gp[3], gp[4], call.x.data_ptr(),
              x_stride, base, base + a_bytes + w_bytes, base + a_bytes, t[0], t[2], t[4], dq.data_ptr(),
              d_v.data_ptr() if d_v is not None else None, d_emb.data_ptr() if d_emb is not None else None,
              flat.data_ptr(), cache.offsets, ws.data_ptr(), ws_bytes, gp[6])
    hook = _GRAD_STAGE_HOOK
    td = call.td
    if td is not None and dq.data_ptr() == td[0].data_ptr():
        # td_step(): d loss / d Q came from the forward launch; the reduce launch also writes the loss (mean of the graphs' terms)
        tb, tn, tbb = td
        tp = tb.data_ptr()
        ctd = common[:6] + common[7:21] + common[22:]          # (no mode, no d_out_v: mode 0)
        if defer_lower and tot >= 3:
            mid = 1 + tot // 2
            _lib.check(L.hexgnn_qnet_backward_flat_td(*ctd, 7, mid, tot, tp + 4 * (tn + tbb), tp + 4 * (tn + 2 * tbb), _stream()),
                       "hexgnn_qnet_backward_flat_td")
            call.pending = (common, mid, flat, cache.cut, ws, d_emb)
        elif hook is None or tot < 3:
            _lib.check(L.hexgnn_qnet_backward_flat_td(*ctd, 7, 1, tot, tp + 4 * (tn + tbb), tp + 4 * (tn + 2 * tbb), _stream()),
                       "hexgnn_qnet_backward_flat_td")
        else:
            mid, cut = 1 + tot // 2, cache.cut
            _lib.check(L.hexgnn_qnet_backward_flat_td(*ctd, 7, mid, tot, tp + 4 * (tn + tbb), tp + 4 * (tn + 2 * tbb), _stream()),
                       "hexgnn_qnet_backward_flat_td")
            hook(flat, cut, cache.total)
            _lib.check(L.hexgnn_qnet_backward_flat(*common, 4, 1, mid, _stream()), "hexgnn_qnet_backward_flat")
            hook(flat, 0, cut)
    elif defer_lower and tot >= 3 and mode != 2:
        mid = 1 + tot // 2
        _lib.check(L.hexgnn_qnet_backward_flat(*common, 7, mid, tot, _stream()), "hexgnn_qnet_backward_flat")
        call.pending = (common, mid, flat, cache.cut, ws, d_emb)
    elif hook is None or tot < 3 or mode == 2:
        _lib.check(L.hexgnn_qnet_backward_flat(*common, 7, 1, tot, _stream()), "hexgnn_qnet_backward_flat")
    else:
        mid, cut = 1 + tot // 2, cache.cut
        _lib.check(L.hexgnn_qnet_backward_flat(*common, 7, mid, tot, _stream()), "hexgnn_qnet_backward_flat")
        hook(flat, cut, cache.total)
        _lib.check(L.hexgnn_qnet_backward_flat(*common, 4, 1, mid, _stream()), "hexgnn_qnet_backward_flat")
        hook(flat, 0, cut)
    _assign_flat_grads(cache, flat, mode, gviews)
    del gviews
    if call.sink is not None:
        call.sink(d_emb[:, :hidden])


def finish_backward(call: "_QNetCall"):
    """Second stage of a backward started with ``defer_lower=True``: the weight-gradient GEMM + slab reduce of the lower half of
    the hidden layers (the head of the flat gradient buffer).  Returns ``(flat, cut)``: the buffer and the position where its
    two segments meet (None, 0 when nothing was deferred).  May be issued again on the same call (same result)."""
    pend = call.pending
    if pend is None:
        return None, 0
    common, mid, flat, cut, _ws, _d_emb = pend
    _lib.check(_lib.lib().hexgnn_qnet_backward_flat(*common, 4, 1, mid, _stream()), "hexgnn_qnet_backward_flat")
    return flat, cut


# d loss / d q of the TdLossFn forward that just ran on THIS thread (picked up by td_loss()); "the loss is the root of the backward
# pass" travels on the loss's own autograd node (ctx._hex_unit, set by backward()): TdLossFn.backward runs on the autograd
# engine's device thread, where neither a module global nor a thread-local of the caller is safe to read
_TD_TLS = threading.local()
_ONES = {}


class TdLossFn(torch.autograd.Function):
    """``loss_fn(Q[sel], target)`` of the DQN update: mean of importance-weighted squared ("mse") or Huber errors over the
    selected nodes.  When Q needs a gradient the forward is ONE launch that also produces ``d loss / d Q``
    (``hexgnn_td_loss_forward_backward``); the backward then hands that buffer on when the loss is the root of the pass
    (``ops.backward(loss)``: nothing is launched at all) and runs the two-launch form's scatter with the upstream scale
    otherwise (``loss.backward()`` / a scaled loss: ``hexgnn_td_loss_backward``).  Same bits either way."""

    @staticmethod
    def forward(ctx, q, sel, target, weights, loss_fn: int):
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        dev = q.device
        qf = q if q.dim() == 1 else q.reshape(-1)
        if qf.dtype != torch.float32 or not qf.is_contiguous():
            qf = qf.float().contiguous()
        # (.to() costs ~2 us even when nothing changes: the usual case -- device tensors of the right type -- skips it)
        if sel.dtype != torch.long or sel.device != dev or not sel.is_contiguous():
            sel = sel.to(device=dev, dtype=torch.long).contiguous()
        tgt = target
        if tgt.dtype != torch.float32 or tgt.device != dev or not tgt.is_contiguous():
            tgt = tgt.to(device=dev, dtype=torch.float32).contiguous()
        w = weights
        if w is not None and (w.dtype != torch.float32 or w.device != dev or not w.is_contiguous()):
            w = w.to(device=dev, dtype=torch.float32).contiguous()
        k, n = sel.numel(), qf.numel()
        if tgt.numel() != k or (w is not None and w.numel() != k):
            raise ValueError("sel / target / weights must have the same length")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        td = torch.empty(k, dtype=torch.float32, device=dev)
        dq = None
        if ctx.needs_input_grad[0]:
            dq = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(L.hexgnn_td_loss_forward_backward(n, k, qf.data_ptr(), sel.data_ptr(), tgt.data_ptr(),
                                                         w.data_ptr() if w is not None else None, loss_fn,
                                                         loss.data_ptr(), td.data_ptr(), dq.data_ptr(), _stream()),
                       "hexgnn_td_loss_forward_backward")
        else:
            _lib.check(L.hexgnn_td_loss_forward(n, k, qf.data_ptr(), sel.data_ptr(), tgt.data_ptr(),
                                                w.data_ptr() if w is not None else None, loss_fn, loss.data_ptr(),
                                                td.data_ptr(), _stream()), "hexgnn_td_loss_forward")
        ctx.save_for_backward(sel, td, w if w is not None else td)
        ctx.dq = dq
        _TD_TLS.last = dq
        ctx.has_w, ctx.loss_fn, ctx.shape = w is not None, loss_fn, q.shape
        ctx.mark_non_differentiable(td)
        return loss, td

    @staticmethod
    def backward(ctx, gloss, _gtd):
        if gloss is None:
            return None, None, None, None, None
        if getattr(ctx, "_hex_unit", False) and ctx.dq is not None:   # the loss is the root: d loss / d q came with the forward launch
            dq, ctx.dq = ctx.dq, None
            return dq.view(ctx.shape), None, None, None, None
        sel, td, w = ctx.saved_tensors
        n = 1
        for d in ctx.shape:
            n *= int(d)
        dq = torch.empty(n, dtype=torch.float32, device=td.device)
        g = gloss.reshape(1).float().contiguous()
        _lib.check(_lib.lib().hexgnn_td_loss_backward(n, int(sel.numel()), sel.data_ptr(), td.data_ptr(),
                                                      w.data_ptr() if ctx.has_w else None, ctx.loss_fn, g.data_ptr(),
                                                      dq.data_ptr(), _stream()), "hexgnn_td_loss_backward")
        return dq.view(ctx.shape), None, None, None, None


def td_loss(q: torch.Tensor, sel: torch.Tensor, target: torch.Tensor, weights: Optional[torch.Tensor] = None,
            loss_fn: str = "mse"):
    """``(loss, td_errors)`` with ``loss = mean(weights * l(q[sel] - target))``, ``l`` = squared error ("mse", the
    reference's ``--loss_fn=mse``) or Huber with delta 1 ("huber"); ``td_errors = q[sel] - target`` (detached) feed
    ``GraphReplayBuffer.update_priorities``.  Equal to ``F.mse_loss(q[sel], target)`` / ``F.huber_loss`` when
    ``weights`` is None."""
    if q.device.type != "cuda":
        raise _lib.HexGnnError("td_loss runs only on the MI355X HIP path (no CPU fallback)")
    q = q.__dict__.get("_hex_plain", q)      # (the model's QValues wrapper: the ordinary tensor underneath, qvalues.py)
    _TD_TLS.last = None
    out = TdLossFn.apply(q, sel, target, weights, {"mse": 0, "huber": 1}[loss_fn])
    loss = out[0]
    loss._hex_td_root = True
    call = getattr(q, "_hex_call", None)
    last = getattr(_TD_TLS, "last", None)
    if call is not None and last is not None and q.dim() == 1 and q.dtype == torch.float32 and q.is_contiguous():
        # q is the fused network's own output and d loss / d q exists already: ops.backward(loss) can run the network's
        # backward directly on this thread (no autograd engine, no hop to its device thread and back: ~100 us of host time)
        loss._hex_direct = (call, last)
    _TD_TLS.last = None
    return out


def backward(loss: torch.Tensor) -> None:
    """``loss.backward()`` for a loss returned by ``td_loss`` without the two launches autograd adds around it: the
    ``ones_like(loss)`` fill that seeds the pass, and the scaling scatter of ``d loss / d Q`` (the fused forward launch
    already produced it for a unit seed).  Any other tensor falls through to ``loss.backward()``.  Gradients are
    bit-identical to ``loss.backward()``."""
    if not getattr(loss, "_hex_td_root", False) or not loss.is_cuda:
        loss.backward()
        return
    direct = loss.__dict__.pop("_hex_direct", None) if _GRAD_STAGE_HOOK is None else None
    if direct is not None and not direct[0].done:
        call, dq = direct
        qnet_direct_backward(call, dq, None)
        call.done = True            # like autograd without retain_graph: a second backward through this forward is an error
        return
    key = (loss.device.type, loss.device.index)
    one = _ONES.get(key)
    if one is None:
        one = torch.ones((), dtype=torch.float32, device=loss.device)
        _ONES[key] = one
    node = loss.grad_fn
    if node is not None:
        node._hex_unit = True          # (read by TdLossFn.backward on the engine's thread: this loss IS the root, seed == 1)
    try:
        torch.autograd.backward((loss,), (one,))
    finally:
        if node is not None:
            node._hex_unit = False


_TD_STEP = threading.local()      # .args = (sel, target, weights, loss_fn) while td_step() runs the model's forward


def td_step(model, x: torch.Tensor, edge_index, graph_indices=None, ptr=None, *, sel: torch.Tensor, target: torch.Tensor,
            weights: Optional[torch.Tensor] = None, loss_fn: str = "mse", defer_lower: bool = False):
    """One DQN update on a batch in its fused form: ``q = model(x, edge_index, graph_indices, ptr)``,
    ``loss, td = td_loss(q, sel, target, weights, loss_fn)``, ``backward(loss)`` -- returns ``(loss, td, q)`` with every
    parameter's ``.grad`` set, same values as those three calls (``td`` and the gradients bit-identical).

    When the update selects ONE node per graph, ``sel[g]`` a node of graph g (the action of the sampled transition: what the
    RainbowDQN step gathers, README.md:5,7) and the batch runs on the fused per-graph kernels, the loss is graph-local and the
    forward kernel forms it in its tail: no launch between the network's forward and backward, ``loss`` is written by the
    backward's reduce launch (``hexgnn_qnet_forward_td`` / ``hexgnn_qnet_backward_flat_td``).  A ``sel[g]`` outside graph g
    makes ``loss`` / ``td[g]`` NaN and raises at the next ``GraphStructure.check()`` (status 16).  Anything else (several
    selections per graph, graphs above 128 nodes, frozen parameters, ``--noisy_dqn``) runs the three calls.

    ``defer_lower=True`` returns ``(loss, td, q, call)`` with only the first stage of the staged backward issued (see
    ``qnet_direct_backward``); ``finish_backward(call)`` issues the rest (a no-op returning (None, 0) where the step did not
    take the fused path and ran whole)."""
    if not x.is_cuda:
        raise _lib.HexGnnError("td_step runs only on the MI355X HIP path (no CPU fallback)")
    dev = x.device
    if sel.dtype != torch.long or sel.device != dev or not sel.is_contiguous():
        sel = sel.to(device=dev, dtype=torch.long).contiguous()
    if target.dtype != torch.float32 or target.device != dev or not target.is_contiguous():
        target = target.to(device=dev, dtype=torch.float32).contiguous()
    if weights is not None and (weights.dtype != torch.float32 or weights.device != dev or not weights.is_contiguous()):
        weights = weights.to(device=dev, dtype=torch.float32).contiguous()
    if target.numel() != sel.numel() or (weights is not None and weights.numel() != sel.numel()):
        raise ValueError("sel / target / weights must have the same length")
    lfn = {"mse": 0, "huber": 1}[loss_fn]
    _TD_STEP.args = (sel, target, weights, lfn)
    try:
        q = model(x, edge_index, graph_indices, ptr)
    finally:
        _TD_STEP.args = None
    q = q.__dict__.get("_hex_plain", q)      # (td_step hands back the ordinary tensor, not the QValues wrapper)
    call = getattr(q, "_hex_call", None)
    td = call.td if call is not None else None
    if td is None:
        loss, tde = td_loss(q, sel, target, weights, loss_fn)
        direct = getattr(loss, "_hex_direct", None)
        if defer_lower and direct is not None and _GRAD_STAGE_HOOK is None and not direct[0].layered:
            loss.__dict__.pop("_hex_direct", None)
            qnet_direct_backward(direct[0], direct[1], None, defer_lower=True)
            direct[0].done = True
            return loss, tde, q, direct[0]
        backward(loss)
        if defer_lower:
            if call is None:
                call = _QNetCall()
            call.pending = None
            return loss, tde, q, call
        return loss, tde, q
    tb, n, b = td
    qnet_direct_backward(call, tb[:n], None, defer_lower=defer_lower and _GRAD_STAGE_HOOK is None)
    call.done = True
    if defer_lower:
        return tb[n + 2 * b], tb[n:n + b], q, call
    return tb[n + 2 * b], tb[n:n + b], q


def greedy_nodes(q: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    """For every graph of a batch the global index of its best non-terminal node: ``ptr[g] + 2 + argmax(q[ptr[g]+2 :
    ptr[g+1]])`` (first maximum) -- the per-graph python argmax of GN0/RainbowDQN/evaluate_elo.py:253-266, and the
    double-DQN action selection over a sampled batch, as one launch (``hexgnn_select_actions`` without a backmap)."""
    _require_cuda(q, "q")
    b = int(ptr.numel()) - 1
    gptr = ptr.to(device=q.device, dtype=torch.int32)
    qf = q.reshape(-1)
    if qf.dtype != torch.float32 or not qf.is_contiguous():
        qf = qf.float().contiguous()
    rank = torch.empty(b, dtype=torch.int32, device=q.device)
    _lib.check(_lib.lib().hexgnn_select_actions(b, gptr.data_ptr(), qf.data_ptr(), None, 0.0, None, None, rank.data_ptr(),
                                                None, _stream()), "hexgnn_select_actions")
    return gptr[:-1].long() + rank.long()
