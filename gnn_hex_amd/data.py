"""Minimal stand-ins for ``torch_geometric.data.Data`` / ``Batch`` (pyg is absent on the GPU box).

Only what the hot path touches: attribute bag with ``x``, ``edge_index``, optional ``backmap`` /
``batch`` / ``ptr``, ``.to(device)``, ``__delattr__`` (Env_manager.get_transitions deletes
``backmap`` in place, graph_game/multi_env_manager.py:145,155,163) and
``Batch.from_data_list`` (collation as torch_geometric 2.2.0 / cpp_hex/hex_graph_game/util.cpp:22-41:
x concatenated, edge_index offset by the running node count, ``batch`` vector, ``ptr`` prefix sums;
other per-graph attributes such as ``backmap`` concatenated un-offset).
"""
from __future__ import annotations

from typing import List, Optional

import torch


class Data:
    def __init__(self, x: Optional[torch.Tensor] = None, edge_index: Optional[torch.Tensor] = None, **kwargs):
        if x is not None:
            self.x = x
        if edge_index is not None:
            self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def to(self, device, non_blocking: bool = False):
        for k in self.keys():
            v = self.__dict__[k]
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device, non_blocking=non_blocking)
        return self

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else v
        return out

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = self.__dict__[k]
            parts.append("%s=%s" % (k, list(v.shape) if torch.is_tensor(v) else v))
        return "%s(%s)" % (self.__class__.__name__, ", ".join(parts))


class Batch(Data):
    """``Batch.from_data_list`` + ``batch`` / ``ptr`` / ``num_graphs``."""

    @classmethod
    def from_data_list(cls, data_list: List[Data]) -> "Batch":
        if hasattr(data_list, "to_batch"):      # an Env_manager observation is already batched on the device
            return data_list.to_batch()
        out = cls()
        if len(data_list) == 0:
            raise ValueError("empty data_list")
        sizes = [int(d.x.shape[0]) for d in data_list]
        device = data_list[0].x.device
        ptr = torch.zeros(len(sizes) + 1, dtype=torch.long)
        ptr[1:] = torch.tensor(sizes, dtype=torch.long).cumsum(0)
        out.x = torch.cat([d.x for d in data_list], dim=0)
        offs = ptr[:-1].tolist()
        out.edge_index = torch.cat([d.edge_index + o for d, o in zip(data_list, offs)], dim=1)
        out.batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(device)
        out.ptr = ptr.to(device)
        skip = {"x", "edge_index", "batch", "ptr"}
        for k in data_list[0].keys():
            if k in skip:
                continue
            vals = [getattr(d, k, None) for d in data_list]
            if all(torch.is_tensor(v) for v in vals):
                out.__dict__[k] = torch.cat([v if v.dim() > 0 else v.view(1) for v in vals], dim=0)
            else:
                out.__dict__[k] = vals
        out._num_graphs = len(sizes)
        # host-known batch metadata spares the model two device->host syncs (side to move, largest graph)
        sides = {getattr(d.x, "_hex_is_maker", None) for d in data_list}
        if len(sides) == 1 and None not in sides:
            out.x._hex_is_maker = sides.pop()
        out.x._hex_max_nodes = max(sizes)
        out.x._hex_hint_version = out.x._version
        out.edge_index._hex_grouped = True      # collated graph by graph: the one-launch CSR build applies
        # ... and knows every graph's edge range without searching for it (torch_geometric keeps the same slices in _slice_dict)
        ecnt = torch.tensor([0] + [int(d.edge_index.shape[1]) for d in data_list], dtype=torch.long)
        out.edge_index._hex_edge_ptr = ecnt.cumsum(0).to(device)
        return out

    @property
    def num_graphs(self) -> int:
        if "_num_graphs" in self.__dict__:
            return self._num_graphs
        return int(self.ptr.numel()) - 1
