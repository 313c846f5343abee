"""Drop-in mirror of the HexAra policy/value network of ``GN0.torch_script_models``: ``get_current_model("SAGE", ...)`` ->
``SAGE_torch_script`` (GN0/torch_script_models.py:286-379, 495-507), same module tree and state-dict keys
(``gnn.convs.*``, ``my_modules.{value_head,policy_head}.convs.*``, ``my_modules.{value_linear,swap_linear}.layers.*``),
same call signature ``model(x, edge_index, graph_indices, batch_ptr) -> (pi, value, output_graph_indices,
output_batch_ptr)``.

Every forward / backward runs in libhexgnn.so (CUDA(HIP) tensors only, no CPU fallback):

* body, value head and the policy head's hidden layers: the layer-major SAGE stack (``ops.sage_stack``; ModifiedBaseNet
  applies the activation after every layer but the last, torch_script_models.py:170-175);
* the policy head's last layer SAGEConv(H, 1): ``hexgnn_sage_scalar_*``;
* 4-way pooling + the value MLP with tanh, and the swap MLP: the head-tail kernel of the RainbowDQN path (its advantage
  linear fed with zeros);
* terminal-node removal, swap-logit insertion and ``scatter_log_softmax``: ``hexgnn_policy_log_softmax_*``.

``norm=LayerNorm`` (``gnn_hex_amd.models.LayerNorm`` for pyg's, the option rl_loop/train_config.py:10,131 offers; round 4): conv ->
whole-batch LayerNorm -> relu for all layers of a net but the last (``ops.sage_norm_stack``), the last conv on its own, and
``before_head_norm`` on the embeddings (GN0/torch_script_models.py:151-160, 179-187, 306, 316-317).

Out of scope (NotImplementedError): other norms, the PNA / CNN / Unet / Gao variants of the same file.  TorchScript tracing (``rl_loop/trace_model.py``) does not apply: the model already is native code.
"""
from __future__ import annotations

from typing import Optional

import copy

import torch
from torch import Tensor
from torch.nn import ModuleList

from . import ops
from .models import MLP, LayerNorm, SAGEConv


class ModifiedSAGEConv(SAGEConv):
    """GN0/torch_script_models.py:52-73: pyg SAGEConv(aggr="mean") restated; parameter holder + stand-alone forward."""


class ModifiedBaseNet(torch.nn.Module):
    """GN0/torch_script_models.py:75-189 with ``conv_class=ModifiedSAGEConv``, ``norm=None``: ``num_layers`` SAGE layers, the
    activation after every layer but the last; ``out_channels`` (1 for the policy head) replaces the last layer's width."""

    def __init__(self, in_channels: int, hidden_channels: int, num_layers: int, out_channels: Optional[int] = None,
                 conv_class=ModifiedSAGEConv, norm=None, norm_kwargs=None, **kwargs):
        super().__init__()
        if norm is not None and not isinstance(norm, LayerNorm):
            raise NotImplementedError("ModifiedBaseNet: norm is None or a gnn_hex_amd.models.LayerNorm instance (pyg's LayerNorm "
                                      "in the batch-less call form, rl_loop/train_config.py:10,131)")
        if conv_class is not ModifiedSAGEConv or kwargs:
            raise NotImplementedError("ModifiedBaseNet: only ModifiedSAGEConv without extra arguments")
        if out_channels not in (None, 1):
            raise NotImplementedError("ModifiedBaseNet: out_channels is None or 1 (the policy head)")
        self.conv_class = conv_class
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.act = torch.nn.ReLU()
        self.norm = None
        self.norm_kwargs = norm_kwargs
        self.out_channels = out_channels if out_channels is not None else hidden_channels
        self.convs = ModuleList()
        c = in_channels
        if num_layers > 1:
            self.convs.append(self.init_conv(c, hidden_channels))
            c = hidden_channels
        for _ in range(num_layers - 2):
            self.convs.append(self.init_conv(c, hidden_channels))
            c = hidden_channels
        if out_channels is not None:
            self._is_conv_to_out = True
            self.convs.append(self.init_conv(c, out_channels))
        else:
            self.convs.append(self.init_conv(c, hidden_channels))
        self.norms = None
        if norm is not None:             # GN0/torch_script_models.py:151-160: one copy per layer but the last
            if hidden_channels > 128:
                raise NotImplementedError("norms need hidden_channels <= 128")
            self.norms = ModuleList([copy.deepcopy(norm) for _ in range(num_layers - 1)])

    def init_conv(self, in_channels: int, out_channels: int, **kwargs):
        return self.conv_class(in_channels, out_channels, **kwargs)

    def reset_parameters(self):
        for conv in self.convs:
            conv.lin_l.reset_parameters()
            conv.lin_r.reset_parameters()

    def forward(self, x: Tensor, edge_index: Tensor, _graph: Optional[ops.GraphStructure] = None) -> Tensor:
        """[n, hidden] (or [n, 1] with out_channels=1); no activation after the last layer."""
        ops._require_cuda(x, "x")
        gs = _graph if _graph is not None else ops.GraphStructure(edge_index, x.shape[0])
        h = self.hidden_channels
        if self.norms is not None and len(self.norms) > 0:
            # GN0/torch_script_models.py:179-187: conv -> norm -> relu for all layers but the last (one stack call with the
            # whole-batch LayerNorm between a layer's contraction and its ReLU), then the last conv on its own
            convs = list(self.convs)
            x = ops.sage_norm_stack(x, gs, self.in_channels, h, convs[:-1], list(self.norms))
            if self.out_channels == 1 and h != 1:
                return ops.sage_scalar(x, gs, h, convs[-1]).view(-1, 1)
            return ops.sage_stack(x, gs, h, h, convs[-1:], linear_last=True)
        if self.out_channels == 1 and h != 1:
            if len(self.convs) > 1:
                x = ops.sage_stack(x, gs, self.in_channels, h, list(self.convs)[:-1])
            elif self.in_channels != h:
                raise NotImplementedError("a one-layer policy head needs in_channels == hidden_channels")
            return ops.sage_scalar(x, gs, h, self.convs[-1]).view(-1, 1)
        return ops.sage_stack(x, gs, self.in_channels, h, self.convs, linear_last=True)


class SAGE_torch_script(torch.nn.Module):
    """GN0/torch_script_models.py:286-379."""

    def __init__(self, hidden_channels, hidden_layers, policy_layers, value_layers, in_channels=3, swap_allowed=False,
                 norm=None, **gnn_kwargs):
        super().__init__()
        if gnn_kwargs:
            raise NotImplementedError("SAGE_torch_script: no extra conv arguments on the HIP path")
        if norm is not None and norm is not LayerNorm:
            raise NotImplementedError("SAGE_torch_script: norm is None or the LayerNorm class (gnn_hex_amd.models.LayerNorm for "
                                      "torch_geometric.nn.norm.LayerNorm, rl_loop/train_config.py:10,131)")
        mk = (lambda: None) if norm is None else (lambda: norm(hidden_channels))
        self.final_conv_acts = None
        self.final_conv_grad = None
        self.swap_allowed = swap_allowed
        self.gnn = ModifiedBaseNet(in_channels=in_channels, norm=mk(), hidden_channels=hidden_channels,
                                   num_layers=hidden_layers, conv_class=ModifiedSAGEConv)
        self.my_modules = torch.nn.ModuleDict()
        self.my_modules["value_head"] = ModifiedBaseNet(in_channels=hidden_channels, norm=mk(), hidden_channels=hidden_channels,
                                                        conv_class=ModifiedSAGEConv, num_layers=value_layers)
        self.my_modules["policy_head"] = ModifiedBaseNet(in_channels=hidden_channels, norm=mk(),
                                                         hidden_channels=hidden_channels, num_layers=policy_layers,
                                                         conv_class=ModifiedSAGEConv, out_channels=1)
        self.my_modules["value_linear"] = MLP(hidden_channels // 2, 1, hidden_channels * 4, 1)
        self.my_modules["swap_linear"] = MLP(hidden_channels // 2, 1, hidden_channels * 4, 1)
        self.before_head_norm = mk()          # GN0/torch_script_models.py:306
        self.value_activation = torch.nn.Tanh()

    def activations_hook(self, grad):
        self.final_conv_grads = grad

    def _pooled_mlp(self, v, gptr, b, mlp: MLP, mode: int):
        """[sum|max|min|mean] pooling + Linear(4H, H/2) -> relu -> Linear(H/2, 1) in the head-tail kernel (mode 1: tanh of
        it, mode 3: raw); its per-node advantage linear gets zero weights and its output is dropped."""
        h = self.gnn.hidden_channels
        zw = self.__dict__.get("_zero_lin")
        if zw is None or zw[0].device != v.device or zw[0].shape[1] != h:
            zw = (torch.zeros(1, h, device=v.device), torch.zeros(1, device=v.device))
            self.__dict__["_zero_lin"] = zw
        out_v, _ = ops.HeadTailFn.apply(v, gptr, b, h, mode, zw[0], zw[1], mlp.layers[0].weight, mlp.layers[0].bias,
                                        mlp.layers[1].weight, mlp.layers[1].bias)
        return out_v

    def forward(self, x: Tensor, edge_index: Tensor, graph_indices: Tensor, batch_ptr: Tensor):
        ops._require_cuda(x, "x")
        assert ((batch_ptr[1:] - batch_ptr[:-1]) > 2).all()     # with only 2 nodes left, someone must have won before
        n = x.shape[0]
        b = int(batch_ptr.numel()) - 1
        gs = ops.GraphStructure(edge_index, n)
        gptr = batch_ptr.to(torch.int32)
        embeds = self.gnn(x, edge_index, _graph=gs)
        if self.before_head_norm is not None:                   # GN0/torch_script_models.py:316-317
            embeds = self.before_head_norm(embeds)
        self.final_conv_acts = embeds.detach()
        if embeds.requires_grad:
            embeds.register_hook(self.activations_hook)
        pi_raw = self.my_modules["policy_head"](embeds, edge_index, _graph=gs).view(-1)
        value_embeds = self.my_modules["value_head"](embeds, edge_index, _graph=gs)
        value = self._pooled_mlp(value_embeds, gptr, b, self.my_modules["value_linear"], 1)
        should_swap = None
        if self.swap_allowed:
            should_swap = self._pooled_mlp(value_embeds, gptr, b, self.my_modules["swap_linear"], 3)
        pi, out_gi, out_ptr = ops.PolicyLogSoftmaxFn.apply(pi_raw, should_swap, x, gptr, b, bool(self.swap_allowed))
        if self.swap_allowed:
            count = int(out_ptr[-1])                            # data dependent (the reference's boolean indexing syncs too)
            pi, out_gi = pi[:count], out_gi[:count]
        return pi, value.reshape(value.size(0)), out_gi, out_ptr


def get_current_model(net_type="SAGE", hidden_channels=60, hidden_layers=15, policy_layers=2, value_layers=2, in_channels=3,
                      swap_allowed=False, norm=None):
    """GN0/torch_script_models.py:495-507; only the graph network of the HexAra pipeline is on the MI355X path."""
    if net_type == "SAGE":
        return SAGE_torch_script(hidden_channels=hidden_channels, hidden_layers=hidden_layers, policy_layers=policy_layers,
                                 value_layers=value_layers, in_channels=in_channels, swap_allowed=swap_allowed, norm=norm)
    if net_type in ("PNA", "PV_CNN"):
        raise NotImplementedError("%r: only net_type='SAGE' is part of the MI355X hot path (SURVEY.md section 8)" % (net_type,))
    raise ValueError("Invalid net type")
