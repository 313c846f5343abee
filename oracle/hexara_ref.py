"""CPU restatement of the HexAra policy/value network ``SAGE_torch_script`` (GN0/torch_script_models.py:286-379).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  The arithmetic (SAGEConv, scatter, scatter_log_softmax) is
torch_geometric / torch_scatter code that is absent here: PARITY UNPINNED for the floating-point values.  The INDEX
surgery of the output (terminal nodes removed, swap logit inserted, output batch pointer / graph indices) IS pinned: the
reference's own ``rl_loop/unittest_model.py:16-92`` holds exact expected sizes, graph indices and batch pointers for three
hand-made batches and the invariants of a randomized test; ``tests/test_oracle_hexara.py`` checks this file against them.

* ``ModifiedBaseNetRef``   <- GN0/torch_script_models.py:75-189 (layer layout 123-144; forward 167-189: activation after
                              every layer but the last; with ``norm`` -- an INSTANCE, deep-copied per layer like
                              normalization_resolver does (151-160) -- conv -> norm -> relu for all layers but the last, then
                              the last conv alone (177-187).  rl_loop/train_config.py:10,131 offers pyg's LayerNorm, called
                              without a batch vector: ``model_ref.LayerNormRef``)
* ``SageTorchScriptRef``   <- GN0/torch_script_models.py:286-379; the output surgery (326-376) restated graph by graph
* ``scatter_log_softmax_ref`` <- torch_scatter 2.1.0 ``composite.scatter_log_softmax``: per group, x - max, then
                              minus log(sum(exp(.)))
* ``get_current_model_ref`` <- GN0/torch_script_models.py:495-507, ``net_type="SAGE"``
"""
from __future__ import annotations

import copy

import torch
from torch import Tensor

from .model_ref import MLPRef, SAGEConvRef, scatter_ref


def scatter_log_softmax_ref(src: Tensor, index: Tensor) -> Tensor:
    n_groups = int(index.max()) + 1 if index.numel() > 0 else 0
    mx = scatter_ref(src.detach().view(-1, 1), index, dim_size=n_groups, reduce="max").view(-1)
    rec = src - mx.index_select(0, index)
    se = src.new_zeros(n_groups).index_add(0, index, rec.exp())
    return rec - se.log().index_select(0, index)


class ModifiedBaseNetRef(torch.nn.Module):
    def __init__(self, in_channels: int, hidden_channels: int, num_layers: int, out_channels=None, norm=None, **_):
        super().__init__()
        self.in_channels, self.hidden_channels, self.num_layers = in_channels, hidden_channels, num_layers
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
        if norm is not None:                 # GN0/torch_script_models.py:151-160
            self.norms = torch.nn.ModuleList([copy.deepcopy(norm) for _ in range(num_layers - 1)])

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        if self.norms is None:               # GN0/torch_script_models.py:173-178
            for i, conv in enumerate(self.convs):
                x = conv(x, edge_index)
                if i != self.num_layers - 1:
                    x = torch.relu(x)
            return x
        assert len(self.norms) == len(self.convs) - 1          # 179-187
        for i, (norm, conv) in enumerate(zip(self.norms, self.convs)):
            x = conv(x, edge_index)
            if i != self.num_layers - 1:
                x = torch.relu(norm(x))
        return self.convs[-1](x, edge_index)


class SageTorchScriptRef(torch.nn.Module):
    def __init__(self, hidden_channels, hidden_layers, policy_layers, value_layers, in_channels=3, swap_allowed=False,
                 norm=None):
        super().__init__()
        self.final_conv_acts = None
        self.final_conv_grads = None
        self.swap_allowed = swap_allowed
        mk = (lambda: None) if norm is None else (lambda: norm(hidden_channels))      # (292-298: a fresh instance per net)
        self.gnn = ModifiedBaseNetRef(in_channels, hidden_channels, hidden_layers, norm=mk())
        self.my_modules = torch.nn.ModuleDict()
        self.my_modules["value_head"] = ModifiedBaseNetRef(hidden_channels, hidden_channels, value_layers, norm=mk())
        self.my_modules["policy_head"] = ModifiedBaseNetRef(hidden_channels, hidden_channels, policy_layers, out_channels=1,
                                                            norm=mk())
        self.my_modules["value_linear"] = MLPRef(hidden_channels // 2, 1, hidden_channels * 4, 1)
        self.my_modules["swap_linear"] = MLPRef(hidden_channels // 2, 1, hidden_channels * 4, 1)
        self.before_head_norm = mk()          # 306
        self.value_activation = torch.nn.Tanh()

    def activations_hook(self, grad):
        self.final_conv_grads = grad

    def forward(self, x: Tensor, edge_index: Tensor, graph_indices: Tensor, batch_ptr: Tensor):
        assert ((batch_ptr[1:] - batch_ptr[:-1]) > 2).all()
        embeds = self.gnn(x, edge_index)
        if self.before_head_norm is not None:                  # 316-317
            embeds = self.before_head_norm(embeds)
        self.final_conv_acts = embeds
        if embeds.requires_grad:
            embeds.register_hook(self.activations_hook)
        pi = self.my_modules["policy_head"](embeds, edge_index)
        value_embeds = self.my_modules["value_head"](embeds, edge_index)
        nb = batch_ptr.numel() - 1
        parts = [scatter_ref(value_embeds, graph_indices, dim_size=nb, reduce=r) for r in ("sum", "max", "min", "mean")]
        graph_parts = torch.cat(parts, dim=1)
        value = self.value_activation(self.my_modules["value_linear"](graph_parts))
        pi = pi.reshape(pi.size(0))
        # Output surgery of GN0/torch_script_models.py:326-376, restated graph by graph (the reference does it with boolean
        # masks and a cumulative sum over the whole batch; rl_loop/unittest_model.py pins that both give the same indices):
        #   * the first two rows of every graph are the terminal nodes: no move, dropped;
        #   * with swap_allowed, a graph in which swapping is possible gets ONE more entry behind its nodes, the swap logit of
        #     its pooled features.  "Possible" is read from feature 2 -- of the graph's LAST row for every graph but the last
        #     one (lines 337-338: x[batch_ptr[1:-1] - 1, 2]), of its FIRST row for the last graph (line 347: x[batch_ptr[-2], 2]);
        #   * output_batch_ptr = running start of the segments, output_graph_indices = the segment index of every entry;
        #   * log-softmax per segment (scatter_log_softmax).
        should_swap = None
        if self.swap_allowed:
            should_swap = self.my_modules["swap_linear"](graph_parts).reshape(nb)
        pieces, seg_ids, starts = [], [], [0]
        for gi in range(nb):
            lo, hi = int(batch_ptr[gi]), int(batch_ptr[gi + 1])
            seg = pi[lo + 2:hi]
            if self.swap_allowed:
                probe_row = hi - 1 if gi < nb - 1 else lo
                if bool(x[probe_row, 2] != 0):
                    seg = torch.cat((seg, should_swap[gi:gi + 1]))
            pieces.append(seg)
            seg_ids.append(torch.full((seg.numel(),), gi, dtype=graph_indices.dtype))
            starts.append(starts[-1] + seg.numel())
        pi = torch.cat(pieces)
        output_graph_indices = torch.cat(seg_ids)
        output_batch_ptr = torch.tensor(starts, dtype=batch_ptr.dtype)
        pi = scatter_log_softmax_ref(pi, output_graph_indices)
        return pi, value.reshape(value.size(0)), output_graph_indices, output_batch_ptr


def get_current_model_ref(net_type="SAGE", hidden_channels=60, hidden_layers=15, policy_layers=2, value_layers=2,
                          in_channels=3, swap_allowed=False, norm=None):
    if net_type != "SAGE":
        raise NotImplementedError(net_type)
    return SageTorchScriptRef(hidden_channels=hidden_channels, hidden_layers=hidden_layers, policy_layers=policy_layers,
                              value_layers=value_layers, in_channels=in_channels, swap_allowed=swap_allowed, norm=norm)
