"""CPU restatement of the reference's ``modern_two_headed`` Q-network.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED: the
reference executes this path through torch_geometric 2.2.0 (``GraphSAGE`` /
``BasicGNN`` / ``SAGEConv``) and torch_scatter 2.1.0 (``scatter``), neither of
which is vendored or installed; the reference's tests pin no number for it.
This file restates the published algorithm of those packages op for op, anchored
on the reference's own call sites:

* ``MLPRef``                <- GN0/models.py:36-82
* ``SAGEConvRef``           <- GN0/torch_script_models.py:52-73 (in-tree restatement
                               of pyg SAGEConv: lin_l(mean_j x_j) + lin_r(x_i),
                               bias on lin_l only, GN0/models.py:175-177)
* ``GraphSAGERef``          <- GN0/torch_script_models.py:96-144 (layer layout) and
                               GN0/models.py:144-164,261-294 (CachifiedGNN loop:
                               ReLU after EVERY layer because out_channels=None)
* ``LayerNormRef``          <- torch_geometric 2.2.0 ``nn.norm.LayerNorm`` (--norm=True,
                               GN0/models.py:8,935,945), mode "graph", called WITHOUT a batch
                               vector (GN0/models.py:286-287,550-551): statistics over all
                               nodes and channels of the batch, eps added to the std
* ``CachedGraphNormRef``    <- GN0/models.py:644-670 over torch_geometric 2.2.0 ``GraphNorm`` (the ``two_headed``
                               family, GN0/models.py:901-918; called without a batch vector,
                               GN0/models.py:282-283: per-channel statistics over all nodes)
* ``FactorizedNoisyLinearRef`` <- GN0/models.py:84-141 (--noisy_dqn=True: the heads'
                               advantage linear, GN0/models.py:331-334)
* ``HeadNetworkRef``        <- GN0/models.py:318-384
* ``DuellingTwoHeadedRef``  <- GN0/models.py:477-590
* ``get_pre_defined_ref``   <- GN0/models.py:892-947 (``modern_two_headed`` and ``two_headed``)
* ``scatter_ref``           <- torch_scatter 2.1.0 ``scatter`` (sum / mean / max /
                               min over dim 0; mean divides by max(count,1); max/min
                               send their gradient to the FIRST index that attains
                               the extremum, as the CPU kernel's strict comparison does)

Execution structure mirrors the reference on purpose (materialised ``[E, C]``
gather, scatter, two linears per layer): ``bench.py`` times it as the CPU
baseline ("CPU restatement of the torch_geometric path").
"""
from __future__ import annotations

from argparse import Namespace
from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor


def scatter_ref(src: Tensor, index: Tensor, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    """torch_scatter.scatter(src, index, dim=0, dim_size=..., reduce=...)."""
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    tail = src.shape[1:]
    if reduce in ("sum", "add", "mean"):
        out = src.new_zeros((dim_size,) + tuple(tail))
        out = out.index_add(0, index, src)
        if reduce == "mean":
            cnt = src.new_zeros(dim_size).index_add(0, index, src.new_ones(index.numel()))
            cnt = cnt.clamp(min=1)
            out = out / cnt.view((-1,) + (1,) * len(tail))
        return out
    if reduce in ("max", "min"):
        flat = src.reshape(src.shape[0], -1)
        n, c = flat.shape
        idx = index.view(-1, 1).expand(n, c)
        red = "amax" if reduce == "max" else "amin"
        with torch.no_grad():
            ext = torch.full((dim_size, c), float("-inf") if reduce == "max" else float("inf"),
                             dtype=src.dtype).scatter_reduce(0, idx, flat, red, include_self=True)
            hit = flat == ext.index_select(0, index)
            pos = torch.arange(n).view(-1, 1).expand(n, c)
            cand = torch.where(hit, pos, torch.full_like(pos, n))
            arg = torch.full((dim_size, c), n, dtype=torch.long).scatter_reduce(
                0, idx, cand, "amin", include_self=True)
            empty = arg >= n
            arg = arg.clamp(max=max(n - 1, 0))
        out = flat.gather(0, arg)
        out = torch.where(empty, torch.zeros_like(out), out)  # empty segments -> 0
        return out.view((dim_size,) + tuple(tail))
    raise ValueError(reduce)


class MLPRef(torch.nn.Module):
    """GN0/models.py:36-82 (grow_input_width omitted: parameter surgery, not arithmetic)."""

    def __init__(self, hidden_channels, num_hidden_layers, num_input, num_output, output_activation=None):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        self.num_input = num_input
        self.num_output = num_output
        self.hidden_channels = hidden_channels
        if num_hidden_layers == 0:
            self.layers.append(torch.nn.Linear(num_input, num_output))
        else:
            self.layers.append(torch.nn.Linear(num_input, hidden_channels))
            for _ in range(num_hidden_layers - 1):
                self.layers.append(torch.nn.Linear(hidden_channels, hidden_channels))
            self.layers.append(torch.nn.Linear(hidden_channels, num_output))
        self.output_activation = output_activation

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
            if layer is not self.layers[-1]:
                x = F.relu(x)
        if self.output_activation is not None:
            x = self.output_activation(x)
        return x


class SAGEConvRef(torch.nn.Module):
    """pyg SAGEConv(aggr='mean', root_weight=True, bias=True), torch_script_models.py:52-73."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.lin_l = torch.nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = torch.nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        x_j = x.index_select(0, src)                                     # propagate: gather [E, C]
        out = scatter_ref(x_j, dst, dim_size=x.size(0), reduce="mean")   # aggregate at the targets
        out = self.lin_l(out)
        out = out + self.lin_r(x)
        return out


class LayerNormRef(torch.nn.Module):
    """pyg 2.2.0 LayerNorm(in_channels, eps=1e-5, affine=True, mode="graph").forward(x, batch=None):
    ``x = x - x.mean(); out = x / (x.std(unbiased=False) + eps); out = out * weight + bias``."""

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))

    def forward(self, x: Tensor) -> Tensor:
        x = x - x.mean()
        out = x / (x.std(unbiased=False) + self.eps)
        return out * self.weight + self.bias


class CachedGraphNormRef(torch.nn.Module):
    """GN0/models.py:644-670 on top of pyg 2.2.0 GraphNorm (weight = 1, bias = 0, mean_scale = 1, eps = 1e-5 inside the
    square root), statement for statement; scatter_mean restated through scatter_ref."""

    supports_cache = True

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = torch.nn.Parameter(torch.ones(in_channels))
        self.mean_cache = None
        self.var_cache = None

    def forward(self, x: Tensor, batch: Optional[Tensor] = None, set_cache=False, use_cache=False) -> Tensor:
        if batch is None:
            batch = x.new_zeros(x.size(0), dtype=torch.long)
        batch_size = int(batch.max()) + 1
        if use_cache and not set_cache:
            mean = self.mean_cache
        else:
            mean = scatter_ref(x, batch, dim_size=batch_size, reduce="mean")
            if set_cache:
                self.mean_cache = mean
        out = x - mean.index_select(0, batch) * self.mean_scale
        if use_cache and not set_cache:
            var = self.var_cache
        else:
            var = scatter_ref(out.pow(2), batch, dim_size=batch_size, reduce="mean")
            if set_cache:
                self.var_cache = var
        std = (var + self.eps).sqrt().index_select(0, batch)
        return self.weight * out / std + self.bias


class GraphSAGERef(torch.nn.Module):
    """BasicGNN layout (torch_script_models.py:118-144) + CachifiedGNN loop (models.py:261-294)."""

    supports_cache = True

    def __init__(self, in_channels: int, hidden_channels: int, num_layers: int,
                 out_channels: Optional[int] = None, norm=None, act="relu", cached_norm=False, **_):
        super().__init__()
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.has_output = out_channels is not None
        self.out_channels = out_channels if out_channels is not None else hidden_channels
        self.convs = torch.nn.ModuleList()
        c = in_channels
        if num_layers > 1:
            self.convs.append(SAGEConvRef(c, hidden_channels))
            c = hidden_channels
        for _ in range(num_layers - 2):
            self.convs.append(SAGEConvRef(c, hidden_channels))
            c = hidden_channels
        self.convs.append(SAGEConvRef(c, self.out_channels))
        self.norms = None
        self.cached_norm = cached_norm
        self.has_cache = False
        if norm is not None:     # BasicGNN: num_layers - 1 norms; CachifiedGNN appends the last one (GN0/models.py:158-162)
            make = CachedGraphNormRef if norm == "cached_graph_norm" else LayerNormRef
            self.norms = torch.nn.ModuleList(make(hidden_channels) for _ in range(num_layers - 1))
            if not self.has_output:
                self.norms.append(make(hidden_channels))

    def export_norm_cache(self):                          # GN0/models.py:165-174
        if self.norms is None:
            return
        assert self.has_cache
        return (torch.stack([n.mean_cache for n in self.norms]), torch.stack([n.var_cache for n in self.norms]))

    def import_norm_cache(self, mean_cache, var_cache):   # GN0/models.py:176-182
        if self.norms is None or not self.cached_norm:
            return
        self.has_cache = True
        for i, norm in enumerate(self.norms):
            norm.mean_cache = mean_cache[i]
            norm.var_cache = var_cache[i]

    def forward(self, x: Tensor, edge_index: Tensor, set_cache: bool = False) -> Tensor:
        if set_cache and self.cached_norm:
            self.has_cache = True
        for i in range(self.num_layers):
            x = self.convs[i](x, edge_index)
            if i == self.num_layers - 1 and self.has_output:
                break
            if self.norms is not None:          # act_first is False: norm, then activation (GN0/models.py:281-291)
                if self.cached_norm:
                    x = self.norms[i](x, set_cache=set_cache, use_cache=self.has_cache and not self.training)
                else:
                    x = self.norms[i](x)
            x = F.relu(x)
        return x


class FactorizedNoisyLinearRef(torch.nn.Module):
    """GN0/models.py:84-141: y = (mu_w + sigma_w * eps_w) x + (mu_b + sigma_b * eps_b), factorised noise
    eps_w = f(e_out) f(e_in)^T, eps_b = f(e_out), f(x) = sgn(x) sqrt(|x|)."""

    def __init__(self, in_features: int, out_features: int, sigma_0: float) -> None:
        super().__init__()
        self.in_features, self.out_features, self.sigma_0 = in_features, out_features, sigma_0
        self.weight_mu = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.weight_sigma = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.register_buffer("weight_epsilon", torch.empty(out_features, in_features))
        self.bias_mu = torch.nn.Parameter(torch.empty(out_features))
        self.bias_sigma = torch.nn.Parameter(torch.empty(out_features))
        self.register_buffer("bias_epsilon", torch.empty(out_features))
        with torch.no_grad():
            scale = 1 / (in_features ** 0.5)
            torch.nn.init.uniform_(self.weight_mu, -scale, scale)
            torch.nn.init.uniform_(self.bias_mu, -scale, scale)
            torch.nn.init.constant_(self.weight_sigma, sigma_0 * scale)
            torch.nn.init.constant_(self.bias_sigma, sigma_0 * scale)
        self.reset_noise()

    @torch.no_grad()
    def reset_noise(self) -> None:
        def f(size):
            noise = torch.randn(size)
            return noise.sign().mul_(noise.abs().sqrt_())
        e_in, e_out = f(self.in_features), f(self.out_features)
        self.weight_epsilon.copy_(e_out.outer(e_in))
        self.bias_epsilon.copy_(e_out)

    def forward(self, input: Tensor) -> Tensor:
        return F.linear(input, self.weight_mu + self.weight_sigma * self.weight_epsilon,
                        self.bias_mu + self.bias_sigma * self.bias_epsilon)


class HeadNetworkRef(torch.nn.Module):
    """GN0/models.py:318-384."""

    def __init__(self, in_channels, hidden_channels, out_channels, value_head_type="linear",
                 value_aggr_types=("mean",), num_layers=2, noisy_dqn=False, noise_sigma=0, norm=None,
                 cached_norm=False, **_):
        super().__init__()
        self.gnn = GraphSAGERef(in_channels=in_channels, hidden_channels=hidden_channels, num_layers=num_layers,
                                norm=norm, cached_norm=cached_norm)
        self.supports_cache = True
        self.value_head_type = value_head_type
        self.hidden_channels = hidden_channels
        if value_head_type == "linear":
            self.value_head = torch.nn.Linear(hidden_channels * len(value_aggr_types), 1)
        else:
            self.value_head = MLPRef(hidden_channels // 2, 1, hidden_channels * len(value_aggr_types), 1)
        self.out_channels = out_channels
        self.value_aggr_types = value_aggr_types
        if noisy_dqn:
            self.linear = FactorizedNoisyLinearRef(hidden_channels, out_channels, noise_sigma)
        else:
            self.linear = torch.nn.Linear(hidden_channels, out_channels)

    def export_norm_cache(self):
        return self.gnn.export_norm_cache()

    def import_norm_cache(self, *args):
        return self.gnn.import_norm_cache(*args)

    def forward(self, x, edge_index, graph_indices, advantages_only=False, set_cache=False):
        x = self.gnn(x, edge_index, set_cache=set_cache)
        advantages = self.linear(x)
        if advantages_only:
            return advantages
        parts = [scatter_ref(x, graph_indices, reduce=a) for a in self.value_aggr_types]
        graph_parts = torch.cat(parts, dim=1)
        value = self.value_head(graph_parts)
        return advantages, value


class DuellingTwoHeadedRef(torch.nn.Module):
    """GN0/models.py:477-590."""

    def __init__(self, gnn_kwargs, head_kwargs):
        super().__init__()
        self.gnn = GraphSAGERef(**gnn_kwargs)
        self.after_embed_norm = None
        if gnn_kwargs.get("norm"):
            make = CachedGraphNormRef if gnn_kwargs["norm"] == "cached_graph_norm" else LayerNormRef
            self.after_embed_norm = make(gnn_kwargs["hidden_channels"])
        self.supports_cache = True
        self.value_activation = torch.nn.Tanh()
        self.advantage_activation = torch.nn.Tanh()
        h = gnn_kwargs["hidden_channels"]
        self.maker_head = HeadNetworkRef(in_channels=h, hidden_channels=h, out_channels=1, **head_kwargs)
        self.breaker_head = HeadNetworkRef(in_channels=h, hidden_channels=h, out_channels=1, **head_kwargs)
        self.final_conv_acts = None
        self.final_conv_grads = None

    def activations_hook(self, grad):
        self.final_conv_grads = grad

    def forward(self, x: Tensor, edge_index: Tensor, graph_indices: Optional[Tensor] = None,
                ptr: Optional[Tensor] = None, set_cache: bool = False, advantages_only=False,
                seperate=False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        assert torch.all(x[:, 2] == x[0, 2])
        is_maker = x[0, 2]
        x = x[:, :2]
        if graph_indices is None:
            graph_indices = x.new_zeros(x.size(0), dtype=torch.long)
        embeds = self.gnn(x, edge_index, set_cache=set_cache)
        if self.after_embed_norm is not None:
            embeds = self.after_embed_norm(embeds)
        self.final_conv_acts = embeds
        if embeds.requires_grad:
            embeds.register_hook(self.activations_hook)
        head = self.maker_head if is_maker == 1 else self.breaker_head
        head_res = head(embeds, edge_index, graph_indices, advantages_only=advantages_only, set_cache=set_cache)
        if advantages_only:
            return 2 * self.advantage_activation(head_res)
        advantages = 2 * self.advantage_activation(head_res[0])
        value = self.value_activation(head_res[1])
        batch_size = int(graph_indices.max()) + 1
        adv_means = scatter_ref(advantages, graph_indices, dim_size=batch_size, reduce="mean")
        if seperate:
            return value.squeeze(), (advantages - adv_means.index_select(0, graph_indices)).squeeze()
        return (value.index_select(0, graph_indices)
                + (advantages - adv_means.index_select(0, graph_indices))).squeeze()

    def export_norm_cache(self):                           # GN0/models.py:513-521
        return [m.export_norm_cache() for m in (self.gnn, self.maker_head, self.breaker_head)]

    def import_norm_cache(self, *args):                    # GN0/models.py:523-535
        for m, a in zip((self.gnn, self.maker_head, self.breaker_head), args):
            if a is not None:
                m.import_norm_cache(*a)

    def simple_forward(self, data):
        if hasattr(data, "batch") and data.batch is not None:
            return self.forward(data.x, data.edge_index, data.batch, getattr(data, "ptr", None))
        return self.forward(data.x, data.edge_index)


def get_pre_defined_ref(name: str, args: Optional[Namespace] = None) -> torch.nn.Module:
    """GN0/models.py:892-947, the ``modern_two_headed`` and ``two_headed`` branches."""
    if name == "two_headed":                               # GN0/models.py:901-918
        norm = "cached_graph_norm" if getattr(args, "norm", False) else None
        return DuellingTwoHeadedRef(
            gnn_kwargs=dict(in_channels=2, num_layers=args.num_layers, hidden_channels=args.hidden_channels,
                            cached_norm=True, norm=norm, act="relu"),
            head_kwargs=dict(num_layers=args.num_head_layers if hasattr(args, "num_head_layers") else 2,
                             noisy_dqn=getattr(args, "noisy_dqn", False), noise_sigma=getattr(args, "noisy_sigma0", 0.5),
                             cached_norm=True, norm=norm))
    if name != "modern_two_headed":
        raise NotImplementedError(name)
    norm = True if getattr(args, "norm", False) else None
    return DuellingTwoHeadedRef(
        gnn_kwargs=dict(in_channels=2, num_layers=args.num_layers, hidden_channels=args.hidden_channels,
                        cached_norm=False, norm=norm, act="relu"),
        head_kwargs=dict(value_head_type="mlp", value_aggr_types=("sum", "max", "min", "mean"),
                         num_layers=args.num_head_layers if hasattr(args, "num_head_layers") else 2,
                         noisy_dqn=getattr(args, "noisy_dqn", False), noise_sigma=getattr(args, "noisy_sigma0", 0.5),
                         norm=norm))
