"""Drop-in mirror of ``GN0.models`` for the RainbowDQN hot path: ``get_pre_defined("modern_two_headed", args)``.

Same module tree, attribute names, method signatures and state-dict keys as the reference
(GN0/models.py:36-82 MLP, 144-295 CachifiedGNN, 318-384 HeadNetwork, 477-590 DuellingTwoHeaded,
892-947 get_pre_defined), so reference checkpoints ``{"state_dict","args","cache"}`` load with
``load_state_dict`` and ``train.py`` uses the model unchanged.  The arithmetic does NOT go through
torch_geometric / torch_scatter: every forward/backward is a call into the hand-written HIP kernels
of libhexgnn.so (gnn_hex_amd/ops.py).  CUDA(HIP) tensors only -- there is no CPU fallback.

``--noisy_dqn=True`` (FactorizedNoisyLinear as the heads' advantage linear, GN0/models.py:84-141,331-334) is
supported on every kernel path: the effective weight ``mu + sigma * eps`` is formed per forward and handed to the
kernels like a plain Linear; autograd carries the kernel's gradient back to ``mu`` and ``sigma``.
``--norm=True`` (torch_geometric LayerNorm, which the model calls in its whole-batch "graph" form, GN0/models.py:286-287,
550-551,935,945) runs on the layer-major kernels: one SAGE layer without ReLU, then the ``hexgnn_graph_layernorm`` kernels
with the ReLU fused; a batch-global statistic rules the per-graph fused kernels out.
``get_pre_defined("two_headed")`` (GN0/models.py:901-918: CachedGraphNorm with ``cached_norm=True``, linear value head over
mean pooling) runs on the same layer-major kernels, one SAGE layer + one ``hexgnn_graph_colnorm`` call per layer, with the
statistics cache of the reference (``set_cache`` / ``export_norm_cache`` / ``import_norm_cache``).
Out of scope here (raise NotImplementedError): the other model families of the reference factory (CNN, PNA, Unet, Gao, the
single-headed Duelling / ActionValue / PolicyValue variants); all BASELINE configs run ``modern_two_headed`` with
``--norm=False --noisy_dqn=False`` (README.md:5,7).
"""
from __future__ import annotations

import copy
from argparse import Namespace
from math import sqrt
from typing import Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn import Linear, ModuleList, Tanh

from . import ops, qvalues
from .data import Batch, Data


class MLP(torch.nn.Module):
    """Parameter holder with the reference layout (GN0/models.py:36-82); evaluated inside the head kernel."""

    def __init__(self, hidden_channels, num_hidden_layers, num_input, num_output, output_activation=None):
        super().__init__()
        self.layers = ModuleList()
        self.num_input = num_input
        self.num_output = num_output
        self.hidden_channels = hidden_channels
        if num_hidden_layers == 0:
            self.layers.append(Linear(num_input, num_output))
        else:
            self.layers.append(Linear(num_input, hidden_channels))
            for _ in range(num_hidden_layers - 1):
                self.layers.append(Linear(hidden_channels, hidden_channels))
            self.layers.append(Linear(hidden_channels, num_output))
        self.output_activation = output_activation

    def grow_input_width(self, new_input_width, new_hidden_channels):
        """GN0/models.py:51-73 for the 1-hidden-layer value head (zero-padded copy of the old weights)."""
        if len(self.layers) != 2:
            raise NotImplementedError("grow_input_width: only the 1-hidden-layer value head is supported")
        old0, old1 = self.layers[0], self.layers[1]
        dev = old0.weight.device
        new0 = Linear(new_input_width, new_hidden_channels).to(dev)
        new0.weight.data.fill_(0)
        new0.bias.data.fill_(0)
        new0.weight.data[:self.hidden_channels, :self.num_input] = old0.weight.data
        new0.bias.data[:self.hidden_channels] = old0.bias.data
        new1 = Linear(new_hidden_channels, 1).to(dev)
        new1.weight.data.fill_(0)
        new1.weight.data[:, :self.hidden_channels] = old1.weight.data
        new1.bias.data[:] = old1.bias.data
        self.layers[0], self.layers[1] = new0, new1
        self.num_input = new_input_width
        self.hidden_channels = new_hidden_channels


class FactorizedNoisyLinear(torch.nn.Module):
    """Factorised Gaussian noise layer of noisy-net DQN, same parameters / buffers / methods as GN0/models.py:84-141
    (state-dict keys weight_mu, weight_sigma, bias_mu, bias_sigma + buffers weight_epsilon, bias_epsilon).  As the heads'
    advantage linear it is evaluated inside the head-tail kernels: ``effective()`` forms w = mu_w + sigma_w * eps_w and
    b = mu_b + sigma_b * eps_b (two tiny element-wise ops on [out, in] = [1, H]) and autograd routes the kernel's
    d w / d b to the four parameters."""

    def __init__(self, in_features: int, out_features: int, sigma_0: float) -> None:
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.sigma_0 = sigma_0
        self.weight_mu = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.weight_sigma = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.register_buffer("weight_epsilon", torch.empty(out_features, in_features))
        self.bias_mu = torch.nn.Parameter(torch.empty(out_features))
        self.bias_sigma = torch.nn.Parameter(torch.empty(out_features))
        self.register_buffer("bias_epsilon", torch.empty(out_features))
        self.reset_parameters()
        self.reset_noise()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        scale = 1 / sqrt(self.in_features)
        torch.nn.init.uniform_(self.weight_mu, -scale, scale)
        torch.nn.init.uniform_(self.bias_mu, -scale, scale)
        torch.nn.init.constant_(self.weight_sigma, self.sigma_0 * scale)
        torch.nn.init.constant_(self.bias_sigma, self.sigma_0 * scale)

    @torch.no_grad()
    def _get_noise(self, size: int) -> Tensor:
        noise = torch.randn(size, device=self.weight_mu.device)
        return noise.sign().mul_(noise.abs().sqrt_())            # f(x) = sgn(x) sqrt(|x|)

    @torch.no_grad()
    def reset_noise(self) -> None:
        epsilon_in = self._get_noise(self.in_features)
        epsilon_out = self._get_noise(self.out_features)
        self.weight_epsilon.copy_(epsilon_out.outer(epsilon_in))
        self.bias_epsilon.copy_(epsilon_out)

    @torch.no_grad()
    def disable_noise(self) -> None:
        self.weight_epsilon[:] = 0
        self.bias_epsilon[:] = 0

    def effective(self) -> Tuple[Tensor, Tensor]:
        return (self.weight_mu + self.weight_sigma * self.weight_epsilon,
                self.bias_mu + self.bias_sigma * self.bias_epsilon)

    def forward(self, input: Tensor) -> Tensor:
        """Stand-alone use (not the hot path, which evaluates the layer inside the head-tail kernel)."""
        w, b = self.effective()
        return torch.nn.functional.linear(input, w, b)


class LayerNorm(torch.nn.Module):
    """torch_geometric.nn.norm.LayerNorm (pyg 2.2.0; imported at GN0/models.py:8, built at 935,945 for --norm=True) in the
    one call form the model uses: mode "graph" and NO batch vector (GN0/models.py:286-287,550-551), i.e. normalisation
    over all nodes and channels of the whole batch, ``(x - mean) / (std + eps) * weight + bias``.  Evaluated by the
    ``hexgnn_graph_layernorm_*`` kernels; ``_relu`` fuses the activation CachifiedGNN applies after the norm."""

    def __init__(self, in_channels: int, eps: float = 1e-5, affine: bool = True, mode: str = "graph"):
        super().__init__()
        if mode != "graph" or not affine:
            raise NotImplementedError("LayerNorm: the model uses mode='graph', affine=True")
        self.in_channels = in_channels
        self.eps = eps
        self.mode = mode
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x: Tensor, batch: Optional[Tensor] = None, _relu: bool = False) -> Tensor:
        if batch is not None:
            raise NotImplementedError("LayerNorm with a batch vector is not on the model's path (GN0/models.py:287)")
        return ops.graph_layernorm(x, self.weight, self.bias, self.eps, _relu)

    def __repr__(self):
        return "%s(%d, mode=%s)" % (self.__class__.__name__, self.in_channels, self.mode)


class CachedGraphNorm(torch.nn.Module):
    """GN0/models.py:644-670: torch_geometric GraphNorm (weight, bias, mean_scale; eps added to the VARIANCE) plus a cache of
    the statistics.  The model calls it without a batch vector (GN0/models.py:282-283, 550-551), i.e. the whole batch is one
    graph: per-channel mean and variance over all nodes.  ``set_cache`` stores the statistics of this call
    (``mean_cache`` / ``var_cache``, shape [1, C] as scatter_mean returns them); ``use_cache and not set_cache`` normalises
    with the stored ones.  Evaluated by the ``hexgnn_graph_colnorm_*`` kernels; ``_relu`` fuses CachifiedGNN's activation."""

    supports_cache = True

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        self.in_channels = in_channels
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = torch.nn.Parameter(torch.ones(in_channels))
        self.mean_cache = None
        self.var_cache = None

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)
        torch.nn.init.zeros_(self.bias)
        torch.nn.init.ones_(self.mean_scale)

    def forward(self, x: Tensor, batch: Optional[Tensor] = None, set_cache=False, use_cache=False,
                _relu: bool = False) -> Tensor:
        if batch is not None:
            raise NotImplementedError("CachedGraphNorm with a batch vector is not on the model's path (GN0/models.py:283)")
        cache = None
        if use_cache and not set_cache:
            if self.mean_cache is None or self.var_cache is None:
                raise RuntimeError("CachedGraphNorm: use_cache without a cache (run a set_cache forward or import_norm_cache)")
            cache = torch.cat([self.mean_cache.reshape(1, -1), self.var_cache.reshape(1, -1)]).detach()
        y, stats = ops.graph_colnorm(x, self.weight, self.bias, self.mean_scale, self.eps, _relu, cache)
        if set_cache:
            self.mean_cache = stats[0:1].clone()
            self.var_cache = stats[1:2].clone()
        return y

    def __repr__(self):
        return "%s(%d)" % (self.__class__.__name__, self.in_channels)


class SAGEConv(torch.nn.Module):
    """Parameter holder for pyg SAGEConv(aggr='mean', root_weight=True, bias=True): ``lin_l`` (with bias)
    acts on the neighbour mean, ``lin_r`` (no bias) on the root (GN0/torch_script_models.py:52-73)."""

    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.lin_l = Linear(in_channels, out_channels, bias=True)
        self.lin_r = Linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index, _graph: Optional[ops.GraphStructure] = None) -> Tensor:
        """``lin_l(mean_{j in N(i)} x_j) + lin_r(x_i)``, no activation (GN0/torch_script_models.py:52-73): a 1-layer
        stack of the layer-major kernels with the ReLU switched off.  Shapes the kernels cover: in_channels <= 8 (raw
        features) or in_channels == out_channels (hidden layer)."""
        ops._require_cuda(x, "x")
        if not (self.in_channels <= 8 or self.in_channels == self.out_channels):
            raise NotImplementedError("SAGEConv(%d, %d): the kernels cover in_channels <= 8 or in_channels == out_channels"
                                      % (self.in_channels, self.out_channels))
        gs = _graph if _graph is not None else ops.GraphStructure(edge_index, x.shape[0])
        return ops.sage_stack(x, gs, self.in_channels, self.out_channels, [self], linear_last=True)


class GraphSAGE(torch.nn.Module):
    """torch_geometric BasicGNN/GraphSAGE layout (restated at GN0/torch_script_models.py:96-144)."""

    supports_edge_weight = False
    supports_edge_attr = False

    def __init__(self, in_channels: int, hidden_channels: int, num_layers: int, out_channels: Optional[int] = None,
                 dropout: float = 0.0, act="relu", norm=None, jk=None, **kwargs):
        super().__init__()
        if norm is not None and not isinstance(norm, (LayerNorm, CachedGraphNorm)):
            raise NotImplementedError("only norm=None, the LayerNorm of modern_two_headed --norm=True (GN0/models.py:935,945) "
                                      "or the CachedGraphNorm of two_headed (GN0/models.py:908,916)")
        if act != "relu" or jk is not None or dropout != 0.0:
            raise NotImplementedError("only act='relu', jk=None, dropout=0 (the modern_two_headed configuration)")
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.dropout = dropout
        self.act = torch.nn.ReLU()
        self.jk_mode = jk
        self.act_first = False
        self.out_channels = out_channels if out_channels is not None else hidden_channels
        self.convs = ModuleList()
        c = in_channels
        if num_layers > 1:
            self.convs.append(self.init_conv(c, hidden_channels, **kwargs))
            c = hidden_channels
        for _ in range(num_layers - 2):
            self.convs.append(self.init_conv(c, hidden_channels, **kwargs))
            c = hidden_channels
        self.convs.append(self.init_conv(c, self.out_channels, **kwargs))
        self.norms = None
        if norm is not None:        # BasicGNN: one norm per hidden layer (CachifiedGNN adds the one of the last layer)
            self.norms = ModuleList()
            for _ in range(num_layers - 1):
                self.norms.append(copy.deepcopy(norm))

    def init_conv(self, in_channels: int, out_channels: int, **kwargs) -> SAGEConv:
        return SAGEConv(in_channels, out_channels, **kwargs)


def _widened_norm(norm, new_width: int, old_width: int, device):
    """A norm of the same class at ``new_width`` whose first ``old_width`` affine parameters are the old ones (the rest keep
    their initial values): what grow_width does to every norm (GN0/models.py:226-235, 499-508)."""
    new_norm = norm.__class__(new_width).to(device)
    new_norm.weight.data[:old_width] = norm.weight.data
    new_norm.bias.data[:old_width] = norm.bias.data
    if hasattr(norm, "mean_scale"):
        new_norm.mean_scale.data[:old_width] = norm.mean_scale.data
    if hasattr(norm, "eps"):
        new_norm.eps = norm.eps
    return new_norm


def cachify_gnn(gnn):
    """GN0/models.py:144-295.  Only GraphSAGE is accelerated."""
    if gnn is not GraphSAGE:
        raise NotImplementedError("cachify_gnn: only GraphSAGE is supported by the HIP path")

    class CachifiedGNN(gnn):
        supports_edge_weight = False
        supports_edge_attr = False
        supports_cache = True

        def __init__(self, *args, out_channels: Optional[int] = None, cached_norm=False, **kwargs):
            if "conv_kwargs" in kwargs:
                self.conv_kwargs = kwargs["conv_kwargs"]
                kwargs.update(kwargs["conv_kwargs"])
                del kwargs["conv_kwargs"]
            else:
                self.conv_kwargs = dict()
            super().__init__(*args, out_channels=out_channels, **kwargs)
            self.cached_norm = cached_norm
            self.has_output = out_channels is not None
            self.has_cache = False
            if self.has_output:
                raise NotImplementedError("out_channels != None (linear last layer) is not on the hot path")
            if cached_norm and self.norms is not None and not isinstance(self.norms[0] if len(self.norms) else kwargs["norm"],
                                                                         CachedGraphNorm):
                raise NotImplementedError("cached_norm=True needs a CachedGraphNorm (GN0/models.py:282-283)")
            if self.norms is not None:      # final norm after the last hidden layer (GN0/models.py:158-162)
                self.norms.append(copy.deepcopy(self.norms[0] if len(self.norms) > 0 else kwargs["norm"]))

        def grow_depth(self, additional_layers):
            """GN0/models.py:166-185: append identity layers (lin_l = 0, lin_r = I)."""
            assert not self.has_output
            self.num_layers += additional_layers
            device = self.convs[0].lin_l.weight.device
            for _ in range(additional_layers):
                conv = self.init_conv(self.hidden_channels, self.hidden_channels, **self.conv_kwargs).to(device)
                conv.lin_l.weight.data[:] = 0
                conv.lin_l.bias.data[:] = 0
                conv.lin_r.weight.data[:] = torch.eye(self.hidden_channels)
                self.convs.append(conv)
                if self.norms is not None:       # GN0/models.py:180-183: a fresh norm of the same class per new layer
                    new_norm = self.norms[0].__class__(self.hidden_channels).to(device)
                    if hasattr(new_norm, "mean_scale"):          # CachedGraphNorm: start without mean subtraction
                        new_norm.mean_scale.data[:] = 0
                    self.norms.append(new_norm)
            self.has_cache = False

        def grow_width(self, new_width, new_in_channels=None):
            """GN0/models.py:187-238: widen every layer, old weights in the top-left block, new input
            columns zero, new output rows freshly initialised."""
            device = self.convs[0].lin_l.weight.device
            old_convs = self.convs
            old_in_channels = self.in_channels
            if new_in_channels is not None:
                self.in_channels = new_in_channels
            self.convs = ModuleList()
            self.convs.append(self.init_conv(self.in_channels, new_width, **self.conv_kwargs).to(device))
            for _ in range(self.num_layers - 1):
                self.convs.append(self.init_conv(new_width, new_width, **self.conv_kwargs).to(device))
            h = self.hidden_channels
            for i, (conv, old) in enumerate(zip(self.convs, old_convs)):
                if i == 0:
                    if new_in_channels is None:
                        conv.lin_l.weight.data[:h, :] = old.lin_l.weight.data
                        conv.lin_r.weight.data[:h, :] = old.lin_r.weight.data
                    else:
                        conv.lin_l.weight.data[:h, :old_in_channels] = old.lin_l.weight.data
                        conv.lin_r.weight.data[:h, :old_in_channels] = old.lin_r.weight.data
                        conv.lin_l.weight.data[:h, old_in_channels:] = 0
                        conv.lin_r.weight.data[:h, old_in_channels:] = 0
                else:
                    conv.lin_l.weight.data[:h, :h] = old.lin_l.weight.data
                    conv.lin_r.weight.data[:h, :h] = old.lin_r.weight.data
                    conv.lin_l.weight.data[:h, h:] = 0
                    conv.lin_r.weight.data[:h, h:] = 0
                conv.lin_l.bias.data[:h] = old.lin_l.bias.data
            if self.norms is not None:           # GN0/models.py:226-235: wider norms, old affine parameters in front
                new_norms = ModuleList()
                for i in range(self.num_layers):
                    new_norms.append(_widened_norm(self.norms[i], new_width, h, device))
                self.norms = new_norms
            self.has_cache = False
            self.hidden_channels = new_width
            self.out_channels = new_width

        def export_norm_cache(self):
            """GN0/models.py:165-174: (stack of mean caches, stack of var caches), [layers, 1, C] each."""
            if self.norms is None:
                return
            assert self.has_cache
            return (torch.stack([norm.mean_cache for norm in self.norms]),
                    torch.stack([norm.var_cache for norm in self.norms]))

        def import_norm_cache(self, mean_cache, var_cache):
            """GN0/models.py:176-182."""
            if self.norms is None or not self.cached_norm:
                return
            self.has_cache = True
            for i, norm in enumerate(self.norms):
                norm.mean_cache = mean_cache[i].to(norm.weight.device)
                norm.var_cache = var_cache[i].to(norm.weight.device)

        def forward(self, x: Tensor, edge_index, *, edge_weight=None, edge_attr=None, set_cache: bool = False,
                    _graph: Optional[ops.GraphStructure] = None) -> Tensor:
            """conv -> [norm] -> relu for every layer (GN0/models.py:261-294 with has_output False)."""
            gs = _graph if _graph is not None else ops.GraphStructure(edge_index, x.shape[0])
            if set_cache and self.cached_norm:
                self.has_cache = True
            if self.norms is None:
                return ops.sage_stack(x, gs, self.in_channels, self.hidden_channels, self.convs)
            if isinstance(self.norms[0], CachedGraphNorm):
                # two_headed family: conv -> CachedGraphNorm -> relu per layer (GN0/models.py:261-294), one SAGE layer without
                # ReLU + one per-channel norm call with the ReLU fused; the cache flags exactly as the reference passes them
                use_cache = self.has_cache and not self.training
                for i in range(self.num_layers):
                    c_in = self.in_channels if i == 0 else self.hidden_channels
                    x = ops.sage_stack(x, gs, c_in, self.hidden_channels, [self.convs[i]], linear_last=True)
                    if self.cached_norm:
                        x = self.norms[i](x, set_cache=set_cache, use_cache=use_cache, _relu=True)
                    else:
                        x = self.norms[i](x, _relu=True)
                return x
            # --norm=True: the whole-batch statistics sit between a layer's contraction and its activation: SAGE layer
            # without ReLU, then LayerNorm with the ReLU fused, layer by layer inside one call per direction
            # (convs[i](x, edge_index) and norms[i](x) on their own remain available as modules)
            return ops.sage_norm_stack(x, gs, self.in_channels, self.hidden_channels, self.convs, self.norms)

    return CachifiedGNN


class HeadNetwork(torch.nn.Module):
    """GN0/models.py:318-384 (noisy_dqn=False)."""

    def __init__(self, in_channels, hidden_channels, out_channels, GNN, value_head_type="linear",
                 value_aggr_types=("mean",), noisy_dqn=True, noise_sigma=0, **gnn_kwargs):
        super().__init__()
        value_aggr_types = tuple(value_aggr_types)
        mlp4 = value_head_type == "mlp" and value_aggr_types == ("sum", "max", "min", "mean")     # modern_two_headed
        lin1 = value_head_type == "linear" and value_aggr_types == ("mean",)                       # two_headed
        if not (mlp4 or lin1) or out_channels != 1:
            raise NotImplementedError("head kernels implement value_head_type='mlp' over (sum,max,min,mean) and "
                                      "value_head_type='linear' over (mean,), out=1")
        self.gnn = GNN(in_channels=in_channels, hidden_channels=hidden_channels, **gnn_kwargs)
        self.supports_cache = hasattr(self.gnn, "supports_cache") and self.gnn.supports_cache
        self.value_head_type = value_head_type
        self.hidden_channels = hidden_channels
        if value_head_type == "linear":
            self.value_head = Linear(self.hidden_channels * len(value_aggr_types), 1)
        else:
            self.value_head = MLP(self.hidden_channels // 2, 1, self.hidden_channels * len(value_aggr_types), 1)
        self.out_channels = out_channels
        self.value_aggr_types = value_aggr_types
        if noisy_dqn:
            self.linear = FactorizedNoisyLinear(hidden_channels, out_channels, noise_sigma)
        else:
            self.linear = Linear(hidden_channels, out_channels)

    def _lin_params(self) -> Tuple[Tensor, Tensor]:
        """(weight [1,H], bias [1]) of the advantage linear as the kernels take them."""
        if isinstance(self.linear, FactorizedNoisyLinear):
            return self.linear.effective()
        return self.linear.weight, self.linear.bias

    def grow_width(self, new_width, new_in_channels=None):
        """GN0/models.py:336-357."""
        assert isinstance(self.linear, Linear)
        self.gnn.grow_width(new_width, new_in_channels=new_in_channels)
        old = self.linear
        self.linear = Linear(new_width, self.out_channels).to(old.weight.device)
        self.linear.weight.data.fill_(0)
        self.linear.weight.data[:, :self.hidden_channels] = old.weight.data
        self.linear.bias.data[:] = old.bias.data[:]
        if self.value_head_type == "linear":       # GN0/models.py:348-353
            old_head = self.value_head
            self.value_head = Linear(new_width, self.out_channels).to(old_head.weight.device)
            self.value_head.weight.data.fill_(0)
            self.value_head.weight.data[:, :self.hidden_channels] = old_head.weight.data
            self.value_head.bias.data[:] = old_head.bias.data[:]
        else:
            self.value_head.grow_input_width(new_width * len(self.value_aggr_types), new_width // 2)
        self.hidden_channels = new_width

    def grow_depth(self, additional_layers):
        self.gnn.grow_depth(additional_layers)

    def export_norm_cache(self, *args, **kwargs):
        return self.gnn.export_norm_cache(*args, **kwargs)

    def import_norm_cache(self, *args, **kwargs):
        return self.gnn.import_norm_cache(*args, **kwargs)

    def _tail(self, x, gptr, b, mode):
        vh = self.value_head
        lin_w, lin_b = self._lin_params()
        if self.value_head_type == "linear":
            return ops.HeadLinearTailFn.apply(x, gptr, b, self.hidden_channels, mode, lin_w, lin_b, vh.weight, vh.bias)
        return ops.HeadTailFn.apply(x, gptr, b, self.hidden_channels, mode, lin_w, lin_b,
                                    vh.layers[0].weight, vh.layers[0].bias, vh.layers[1].weight, vh.layers[1].bias)

    def forward(self, x: Tensor, edge_index: Tensor, graph_indices, advantages_only=False, set_cache=False,
                _graph: Optional[ops.GraphStructure] = None):
        """GN0/models.py:368-384: raw (pre-activation) ``advantages [N,1]`` and ``value [B,1]`` from the body embedding
        ``x [N,H]`` -- the head's SAGE layers on the layer-major kernels, then the head-tail kernel in its raw-output mode.
        (DuellingTwoHeaded does not go through here: its fused kernels evaluate body, head and dueling combine at once.)"""
        ops._require_cuda(x, "x")
        n = x.shape[0]
        gs = _graph if _graph is not None else ops.GraphStructure(edge_index, n)
        hx = self.gnn(x, edge_index, set_cache=set_cache, _graph=gs)
        gptr, b = ops.graph_ptr(graph_indices, None, n, x.device)
        if advantages_only:
            return self._tail(hx, gptr, b, 4).view(-1, 1)
        value, adv = self._tail(hx, gptr, b, 3)
        return adv.view(-1, 1), value.view(-1, 1)


class DuellingTwoHeaded(torch.nn.Module):
    """GN0/models.py:477-590."""

    def __init__(self, GNN, advantage_head, gnn_kwargs, head_kwargs):
        super().__init__()
        self.gnn = GNN(**gnn_kwargs)
        if "norm" in gnn_kwargs and gnn_kwargs["norm"]:
            self.after_embed_norm = copy.deepcopy(gnn_kwargs["norm"])
        else:
            self.after_embed_norm = None
        self.supports_cache = hasattr(self.gnn, "supports_cache") and self.gnn.supports_cache
        self.value_activation = Tanh()
        self.advantage_activation = Tanh()
        h = gnn_kwargs["hidden_channels"]
        self.maker_head = advantage_head(in_channels=h, hidden_channels=h, out_channels=1, **head_kwargs)
        self.breaker_head = advantage_head(in_channels=h, hidden_channels=h, out_channels=1, **head_kwargs)
        self.final_conv_acts = None
        self.final_conv_grad = None

    # final_conv_acts (GN0/models.py:553): the body's output.  The direct fused path stores the call record and builds the
    # [n, hidden] view only when somebody reads the attribute (two view ops per step that Grad-CAM alone looks at).
    @property
    def final_conv_acts(self):
        v = self.__dict__.get("_fca")
        if isinstance(v, ops._QNetCall):
            v = ops.qnet_embeds(v)
            self.__dict__["_fca"] = v
        return v

    @final_conv_acts.setter
    def final_conv_acts(self, value):
        self.__dict__["_fca"] = value

    def grow_depth(self, additional_layers):
        self.gnn.grow_depth(additional_layers)

    def grow_width(self, new_width):
        """GN0/models.py:497-508."""
        old_width = self.gnn.hidden_channels
        self.gnn.grow_width(new_width)
        self.maker_head.grow_width(new_width, new_in_channels=new_width)
        self.breaker_head.grow_width(new_width, new_in_channels=new_width)
        if self.after_embed_norm is not None:
            self.after_embed_norm = _widened_norm(self.after_embed_norm, new_width, old_width,
                                                  self.after_embed_norm.weight.device)

    def export_norm_cache(self, *args):
        cache_list = []
        for m in (self.gnn, self.maker_head, self.breaker_head):
            if hasattr(m, "supports_cache") and m.supports_cache:
                cache_list.append(m.export_norm_cache(*args))
        return cache_list

    def import_norm_cache(self, *args):
        ind = 0
        for m in (self.gnn, self.maker_head, self.breaker_head):
            if hasattr(m, "supports_cache") and m.supports_cache:
                if ind < len(args) and args[ind] is not None:
                    m.import_norm_cache(*args[ind])
                ind += 1

    def activations_hook(self, grad):
        self.final_conv_grads = grad

    def forward(self, x: Tensor, edge_index, graph_indices: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                set_cache: bool = False, advantages_only=False, seperate=False
                ) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        """Same contract as GN0/models.py:537-584.

        Host syncs of the reference (``x[0,2]`` at 538-539, ``graph_indices.max()`` at 576) are avoided when the
        caller already knows the answers: ``ptr`` gives the graph count, and a ``Batch``/``Data`` built by
        ``gnn_hex_amd`` (env manager, ``Batch.from_data_list``) carries the side to move as ``x._hex_is_maker``.
        Without the hint the reference behaviour (assert + sync) is kept."""
        ops._require_cuda(x, "x")
        hint, max_nodes = ops.hints_of(x)          # (None, None) when absent or stale (x edited in place since)
        if hint is None:
            assert torch.all(x[:, 2] == x[0, 2])
            is_maker = bool(x[0, 2] == 1)
        else:
            is_maker = bool(hint)
        n = x.shape[0]
        x2 = x[:, :2]

        gs = getattr(edge_index, "_hex_csr", None)          # CSR emitted by the env builder, if any
        deferred = None
        # edge_index CSR-sorted once per batch; collated batches (edges grouped by graph) take the one-launch build
        grouped = (gs is None or gs.n != n) and getattr(edge_index, "_hex_grouped", False) \
            and max_nodes is not None and max_nodes <= 2048
        if grouped and ptr is not None and ptr.dtype == torch.int64 and ptr.is_cuda and ptr.is_contiguous():
            b = int(ptr.numel()) - 1            # the build reads the int64 ptr itself and emits the int32 copy
            if b > 0 and edge_index.dtype == torch.int64 and edge_index.dim() == 2 and edge_index.is_contiguous() \
                    and edge_index.is_cuda:
                # built below: together with the weight pack when the direct path takes the call (one launch), else right away
                gs = None
                deferred = (edge_index, n, b, ptr)
                gptr = None                                                 # (gs.gptr: a view made on demand)
            else:
                gptr = torch.empty(b + 1, dtype=torch.int32, device=x.device)
                gs = ops.GraphStructure(edge_index, n, gptr, b, ptr64=ptr) if b > 0 else ops.GraphStructure(edge_index, n)
                if b == 0:
                    gptr.zero_()
        else:
            gptr, b = ops.graph_ptr(graph_indices, ptr, n, x.device)
            if gs is None or gs.n != n:
                gs = ops.GraphStructure(edge_index, n, gptr, b) if grouped else ops.GraphStructure(edge_index, n)
        mods = self._modules
        head = mods["maker_head"] if is_maker else mods["breaker_head"]
        mode = 2 if advantages_only else (1 if seperate else 0)

        # fused per-graph path (one launch per direction) when every graph fits a workgroup's LDS
        if max_nodes is None and ops._FUSED_ENABLED:
            if gptr is None:
                gptr = gs.gptr
            max_nodes = int((gptr[1:] - gptr[:-1]).max()) if b > 0 else 0     # host sync (no size hint given)
        ent = self._fused_entry(head)         # (sig, params, pointer cache, (c_in, hidden, body layers, head layers, fusable, noisy))
        c_in, h, n_body, n_head, fusable, noisy = ent[3]
        fused = fusable and max_nodes is not None and ops.qnet_fused_supported(c_in, h, max_nodes) and x2.shape[1] == c_in
        # graphs above 128 nodes / hidden 113..128: the same network on the layer-major kernels, as ONE stack + head tail
        layered = (not fused) and fusable and not noisy and ops._DIRECT_GRADS and c_in <= 8 and c_in != h \
            and x2.shape[1] == c_in and n_body >= 1
        if fused or layered:
            grad_on = torch.is_grad_enabled()
            if ops._DIRECT_GRADS and not noisy:
                # direct-gradient form (ops.QNetDirectFn): cached pointer arrays, one autograd input, gradients assigned to
                # p.grad by the backward itself -- the eager step of an unmodified train.py
                cache = ent[2]
                if not cache.valid():
                    cache.refresh()
                if cache.direct_ok or not grad_on:
                    # (a deferred grouped CSR build goes in as the tuple: built WITH the weight pack, in one launch)
                    fargs = (cache, x2, gs if gs is not None else deferred, gptr, b, c_in, h, n_body, n_head, mode, grad_on,
                             layered)
                    if grad_on:
                        anchor = self.__dict__.get("_hex_anchor")
                        if anchor is None or anchor.device != x.device:
                            anchor = torch.zeros(1, device=x.device, requires_grad=True)
                            self.__dict__["_hex_anchor"] = anchor
                        holder = []
                        outs = ops.QNetDirectFn.apply(anchor, holder, fargs)
                        call = holder[0]
                        call.sink = self.activations_hook
                        q, out_v = outs if mode == 1 else (outs, None)
                        if mode == 0:
                            q._hex_call = call          # lets ops.td_loss / ops.backward run this backward directly
                            # (the loss expression of an unmodified training loop -- q[actions], F.mse_loss, loss.backward()
                            # -- recognised on the tensor itself: gnn_hex_amd/qvalues.py)
                            q = qvalues.wrap_q(q)
                    else:
                        q, out_v, call = ops.qnet_direct_forward(*fargs)
                    self.__dict__["_fca"] = call
                    if mode == 2:
                        return q.view(-1, 1)
                    if mode == 1:
                        return out_v, q
                    return q
            if layered:
                fused = False       # (frozen / hooked parameters: the per-module composition below)
        if gs is None:              # every other path: the structure now
            gs = ops.GraphStructure.grouped(*deferred)
        if fused:
            grad_on = torch.is_grad_enabled()
            params = ent[1]
            if noisy:      # effective weights are formed per forward
                params = params[:-6] + list(head._lin_params()) + params[-4:]
            sink = self.activations_hook if grad_on else None
            if gptr is None:
                gptr = gs.gptr
            outs = ops.QNetFusedFn.apply(x2, gs, gptr, b, c_in, h, n_body, n_head, mode, sink, *params)
            self.final_conv_acts = outs[-1]
            if mode == 2:
                return outs[0].view(-1, 1)
            if mode == 1:
                return outs[0].squeeze(), outs[1].squeeze()
            return outs[0].squeeze()

        if gptr is None:
            gptr = gs.gptr
        # general layer-major path (any graph size; hidden 129..256: the plain kernels of csrc/wide.hip behind the same calls)
        embeds = self.gnn(x2, edge_index, set_cache=set_cache, _graph=gs)
        if self.after_embed_norm is not None:
            embeds = self.after_embed_norm(embeds)
        # values only (as the fused path): keeping the differentiable tensor on the module would keep the whole autograd
        # graph -- and its per-parameter AccumulateGrad nodes -- alive across steps (see gnn_hex_amd/graphs.py)
        self.final_conv_acts = embeds.detach()
        if embeds.requires_grad:
            embeds.register_hook(self.activations_hook)
        hx = head.gnn(embeds, edge_index, set_cache=set_cache, _graph=gs)
        if advantages_only:
            return head._tail(hx, gptr, b, 2).view(-1, 1)
        if seperate:
            v, a = head._tail(hx, gptr, b, 1)
            return v.squeeze(), a.squeeze()
        return head._tail(hx, gptr, b, 0).squeeze()

    def _fused_entry(self, head):
        """Per head: the parameter list of the fused call (body convs, head convs, head tail), its pointer cache and the
        shape constants; rebuilt when the module tree changes (grow_*, parameter replacement by ``.to()`` /
        ``load_state_dict(assign=True)``).  The signature is read through the modules' own dictionaries: a chain of
        ``nn.Module.__getattr__`` calls cost 17 us per forward here."""
        cache = self.__dict__.setdefault("_fused_cache", {})
        key = id(head)
        ent = cache.get(key)
        gnn = self._modules["gnn"]
        convs = gnn._modules["convs"]._modules
        hconvs = head._modules["gnn"]._modules["convs"]._modules
        lin = head._modules["linear"]
        lin0 = lin._parameters.get("weight_mu") if "weight_mu" in lin._parameters else lin._parameters.get("weight")
        first = convs["0"]._modules["lin_l"]._parameters["weight"]
        last = hconvs[str(len(hconvs) - 1)]._modules["lin_r"]._parameters["weight"] if len(hconvs) else first
        sig = (len(convs), len(hconvs), id(first), id(lin0), id(last), gnn.hidden_channels, gnn.norms is None)
        if (ent is None or ent[0] != sig) and head.value_head_type != "mlp":
            # two_headed family (linear value head): per-module composition on the layer-major kernels only
            ent = (sig, [], None, (self.gnn.in_channels, self.gnn.hidden_channels, len(self.gnn.convs), len(head.gnn.convs),
                                   False, isinstance(head.linear, FactorizedNoisyLinear)))
            cache[key] = ent
        if ent is None or ent[0] != sig:
            params = []
            for conv in list(self.gnn.convs) + list(head.gnn.convs):
                params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
            vh = head.value_head
            params += list(head._lin_params()) + [vh.layers[0].weight, vh.layers[0].bias,
                                                   vh.layers[1].weight, vh.layers[1].bias]
            noisy = isinstance(head.linear, FactorizedNoisyLinear)
            n_body, n_head = len(self.gnn.convs), len(head.gnn.convs)
            qcache = None if noisy else ops.QNetParamCache(params, n_body + n_head)
            h = self.gnn.hidden_channels
            fusable = head.gnn.hidden_channels == h and self.gnn.norms is None and head.gnn.norms is None
            ent = (sig, params, qcache, (self.gnn.in_channels, h, n_body, n_head, fusable, noisy))
            cache[key] = ent
        return ent

    def _fused_params(self, head):
        return self._fused_entry(head)[1]

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_fused_cache", None)
        return super()._apply(fn, *args, **kwargs)

    def simple_forward(self, data: Union[Data, Batch]):
        if isinstance(data, Batch) or getattr(data, "batch", None) is not None:
            return self.forward(data.x, data.edge_index, data.batch, getattr(data, "ptr", None))
        return self.forward(data.x, data.edge_index)


def get_pre_defined(name, args: Optional[Namespace] = None) -> torch.nn.Module:
    """GN0/models.py:892-980: ``modern_two_headed`` (the RainbowDQN GNN of README.md:5,7) and its predecessor ``two_headed``
    (CachedGraphNorm with the statistics cache, linear value head over mean pooling)."""
    if name == "two_headed":
        use_norm = bool(getattr(args, "norm", False))
        return DuellingTwoHeaded(
            cachify_gnn(GraphSAGE), HeadNetwork,
            gnn_kwargs=dict(in_channels=2, num_layers=args.num_layers, hidden_channels=args.hidden_channels,
                            cached_norm=True, norm=CachedGraphNorm(args.hidden_channels) if use_norm else None, act="relu"),
            head_kwargs=dict(GNN=cachify_gnn(GraphSAGE),
                             num_layers=args.num_head_layers if hasattr(args, "num_head_layers") else 2,
                             noisy_dqn=args.noisy_dqn, noise_sigma=args.noisy_sigma0, cached_norm=True,
                             norm=CachedGraphNorm(args.hidden_channels) if use_norm else None, act="relu"))
    if name == "modern_two_headed":
        use_norm = bool(getattr(args, "norm", False))
        return DuellingTwoHeaded(
            cachify_gnn(GraphSAGE), HeadNetwork,
            gnn_kwargs=dict(in_channels=2, num_layers=args.num_layers, hidden_channels=args.hidden_channels,
                            cached_norm=False, norm=LayerNorm(args.hidden_channels) if use_norm else None, act="relu"),
            head_kwargs=dict(GNN=cachify_gnn(GraphSAGE), value_head_type="mlp",
                             value_aggr_types=("sum", "max", "min", "mean"),
                             num_layers=args.num_head_layers if hasattr(args, "num_head_layers") else 2,
                             noisy_dqn=args.noisy_dqn, noise_sigma=args.noisy_sigma0, cached_norm=False,
                             norm=LayerNorm(args.hidden_channels) if use_norm else None, act="relu"))
    raise NotImplementedError(
        "%r: only 'modern_two_headed' and 'two_headed' are part of the MI355X hot path (SURVEY.md section 8)" % (name,))
