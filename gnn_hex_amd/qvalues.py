"""The DQN update of an UNMODIFIED training loop on the fused loss path.

The reference's loop (RainbowDQN agent driven by train.py, README.md:5,7) forms its loss with plain torch calls on the
model's output:

    q = model(x, edge_index, batch, ptr)                    # GN0/models.py:537-584
    td_est = q[actions]                                     # torch indexing (or index_select / gather / take)
    loss = F.mse_loss(td_est, td_target)                    # --loss_fn=mse; or reduction='none', `weights * losses`, .mean()
    loss.backward()                                         # --prioritized_er=True: importance weights

Run literally, that is five to ten tiny torch kernels forward, a sort-based ``index_put`` and a hop to the autograd engine's
device thread backward: GNN-S steps at 0.97 M graphs/s that way against 2.0 M with ``ops.td_loss`` + ``ops.backward``.  The
model therefore returns its Q-values as a ``torch.Tensor`` subclass that recognises exactly this expression:

* ``q[idx]`` / ``q.index_select(0, idx)`` / ``q.gather(0, idx)`` / ``q.take(idx)`` with a 1-D integer tensor: computed as usual,
  the result remembers ``(q, idx)``;
* ``mse_loss`` / ``huber_loss(delta=1)`` / ``smooth_l1_loss(beta=1)`` on it: ``reduction='mean'`` IS ``ops.td_loss`` (one launch
  for loss, td and d loss / d Q); ``reduction='none'`` is computed as usual and remembers its operands, a following ``weights *
  losses`` (either order) remembers the weights, and ``.mean()`` / ``torch.mean`` then is ``ops.td_loss`` with them;
* ``loss.backward()`` on that loss is ``ops.backward(loss)``: the network's backward on the caller's thread, straight from the
  gradient the loss launch produced.

Every intermediate is still a real tensor with the reference's value, anything else done with them (``td_est - td_target`` for
the priorities, logging, other losses) is plain torch, and ``torch.autograd.backward`` / ``retain_graph`` / ``create_graph`` /
``inputs=`` keep autograd's own path.  ``set_enabled(False)`` returns plain tensors from the model.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

_ENABLED = True


def set_enabled(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


def enabled() -> bool:
    return _ENABLED


def _plain(fn, *args, **kwargs):
    with torch._C.DisableTorchFunctionSubclass():
        return fn(*args, **kwargs)


class _Base(torch.Tensor):
    """Tensor subclass whose unrecognised operations are ordinary torch operations returning ordinary tensors."""
    _HANDLERS = {}

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        h = cls._HANDLERS.get(func)
        if h is not None:
            out = h(*args, **kwargs)
            if out is not NotImplemented:
                return out
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _is_index(idx) -> bool:
    return torch.is_tensor(idx) and idx.dim() == 1 and idx.dtype in (torch.int64, torch.int32) and type(idx) is torch.Tensor


class QValues(_Base):
    """Q-values of a fused forward (1-D, one per node); ``_hex_plain`` is the ordinary tensor that carries ``_hex_call``."""
    _HANDLERS = {}


class QSelected(_Base):
    """``q[idx]``: remembers (plain q, idx)."""
    _HANDLERS = {}


class QLosses(_Base):
    """Per-sample losses ``l(q[idx] - target)`` (reduction='none'), possibly times importance weights."""
    _HANDLERS = {}


class TdLoss(_Base):
    """The scalar loss ``ops.td_loss`` returned; ``backward()`` runs ``ops.backward``."""
    _HANDLERS = {}

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        plain = self.__dict__.get("_hex_plain")
        if plain is not None and gradient is None and not retain_graph and not create_graph and inputs is None:
            from . import ops
            ops.backward(plain)
            return
        _plain(torch.Tensor.backward, self, gradient, retain_graph, create_graph, inputs)


def wrap_q(q: torch.Tensor) -> torch.Tensor:
    """The model's output as a ``QValues`` (same storage, same autograd history)."""
    if not _ENABLED or q.dim() != 1 or not q.requires_grad:
        return q
    out = _plain(torch.Tensor.as_subclass, q, QValues)
    out.__dict__["_hex_plain"] = q
    call = getattr(q, "_hex_call", None)
    if call is not None:
        out.__dict__["_hex_call"] = call
    return out


def _select(q, idx, computed):
    out = _plain(torch.Tensor.as_subclass, computed, QSelected)
    out.__dict__["_hex_sel"] = (q.__dict__["_hex_plain"], idx)
    return out


def _getitem(q, idx):
    if type(q) is not QValues or not _is_index(idx):
        return NotImplemented
    return _select(q, idx, _plain(torch.Tensor.__getitem__, q, idx))


def _index_select(q, dim, index):
    if type(q) is not QValues or dim not in (0, -1) or not _is_index(index):
        return NotImplemented
    return _select(q, index, _plain(torch.index_select, q, dim, index))


def _gather(q, dim, index, *, sparse_grad=False):
    if type(q) is not QValues or dim not in (0, -1) or not _is_index(index) or sparse_grad:
        return NotImplemented
    return _select(q, index, _plain(torch.gather, q, dim, index))


def _take(q, index):
    if type(q) is not QValues or not _is_index(index):
        return NotImplemented
    return _select(q, index, _plain(torch.take, q, index))


QValues._HANDLERS.update({torch.Tensor.__getitem__: _getitem, torch.index_select: _index_select,
                          torch.Tensor.index_select: _index_select, torch.gather: _gather, torch.Tensor.gather: _gather,
                          torch.take: _take, torch.Tensor.take: _take})


def _as_loss(loss, td):
    out = _plain(torch.Tensor.as_subclass, loss, TdLoss)
    out.__dict__["_hex_plain"] = loss
    out.__dict__["_hex_td"] = td
    return out


def _loss_handler(kind, torch_fn, param_name, param_default):
    def handler(inp, target, *args, **kwargs):
        if type(inp) is not QSelected or not torch.is_tensor(target) or isinstance(target, _Base):
            return NotImplemented
        # positional legacy arguments (size_average, reduce) or anything unusual: torch's own path
        if args or any(k not in ("reduction", param_name, "weight", "size_average", "reduce") for k in kwargs) or \
                any(kwargs.get(k) is not None for k in ("weight", "size_average", "reduce")):
            return NotImplemented
        if param_name and float(kwargs.get(param_name, param_default)) != 1.0:
            return NotImplemented
        reduction = kwargs.get("reduction", "mean")
        q, idx = inp.__dict__["_hex_sel"]
        if target.shape != inp.shape or not target.is_floating_point():
            return NotImplemented
        if reduction == "mean":
            from . import ops
            loss, td = ops.td_loss(q, idx, target.detach(), None, kind)
            return _as_loss(loss, td)
        if reduction == "none":
            kw = {k: v for k, v in kwargs.items() if v is not None}
            out = _plain(torch.Tensor.as_subclass, _plain(torch_fn, inp, target, **kw), QLosses)
            out.__dict__["_hex_terms"] = (q, idx, target, None, kind)
            return out
        return NotImplemented
    return handler


QSelected._HANDLERS.update({
    F.mse_loss: _loss_handler("mse", F.mse_loss, None, None),
    F.huber_loss: _loss_handler("huber", F.huber_loss, "delta", 1.0),
    F.smooth_l1_loss: _loss_handler("huber", F.smooth_l1_loss, "beta", 1.0),
})


def _mul(a, b):
    losses, w = (a, b) if type(a) is QLosses else (b, a)
    if type(losses) is not QLosses or not torch.is_tensor(w) or isinstance(w, _Base) or w.requires_grad:
        return NotImplemented
    q, idx, target, w0, kind = losses.__dict__["_hex_terms"]
    if w0 is not None or w.shape != losses.shape or not w.is_floating_point():
        return NotImplemented
    out = _plain(torch.Tensor.as_subclass, _plain(torch.mul, losses, w), QLosses)
    out.__dict__["_hex_terms"] = (q, idx, target, w, kind)
    return out


def _mean(losses, *args, **kwargs):
    if type(losses) is not QLosses or args or kwargs:
        return NotImplemented
    q, idx, target, w, kind = losses.__dict__["_hex_terms"]
    from . import ops
    loss, td = ops.td_loss(q, idx, target.detach(), w, kind)
    return _as_loss(loss, td)


QLosses._HANDLERS.update({torch.mul: _mul, torch.Tensor.mul: _mul, torch.Tensor.__mul__: _mul, torch.Tensor.__rmul__: _mul,
                          torch.mean: _mean, torch.Tensor.mean: _mean})
