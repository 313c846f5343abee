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


BLOCK_ROWS = 128         # rows of one workgroup of the layer-major kernels (gnn_hex_amd/csrc/sage.hip)


def blocks_for_order(sizes: List[int], block: int = BLOCK_ROWS, head: int = 64) -> List[int]:
    """Row-block table for graphs in a GIVEN order (no reordering): consecutive whole graphs share a block while they fit; a
    graph above ``block`` rows gets blocks of its own -- a first piece of ``head`` rows, the rest in equal pieces of at most
    ``block`` rows (``head = 0``: pieces of ``block`` rows, and the last, partial one stays open for the graphs that follow).

    Why a short first piece: the blocks of one large graph exchange rows layer by layer, and the rows with the longest remote
    neighbour lists are the two terminal nodes at the graph's start (half of a Hex terminal's 2 x size neighbours lie in another
    block wherever the graph is cut).  With 64 rows = four waves that block has every SIMD to a single wave: the wave that
    fetches the terminals' remote rows is not competing for issue slots with a partner's matrix instructions, and the block is
    through a layer before its partners need it.  Measured on MIX256 (profiles/r04/mix_blocks.txt): equal pieces 305 k
    graphs/s, 64-row head 337 k."""
    starts, row, fill = [0], 0, 0
    for sz in sizes:
        sz = int(sz)
        if sz > block:
            if fill:
                starts.append(row)
            h = head if 0 < head < sz else 0
            if h:
                row += h
                starts.append(row)
                rest = sz - h
                k = -(-rest // block)
                base, extra = divmod(rest, k)
                for i in range(k):
                    row += base + (1 if i < extra else 0)
                    starts.append(row)
                fill = 0
            else:
                full, tail = divmod(sz, block)
                for _ in range(full):
                    row += block
                    starts.append(row)
                row += tail
                fill = tail
        else:
            if fill + sz > block:
                starts.append(row)
                fill = 0
            fill += sz
            row += sz
    if starts[-1] != row:
        starts.append(row)
    return starts


def pack_order(sizes: List[int], block: int = BLOCK_ROWS, max_blocks: Optional[int] = None, head: int = 64):
    """Order of the graphs of a batch, and the row blocks that go with it, for the one-launch SAGE stack kernels
    (``hexgnn_sage_stack_*_blocks``): a workgroup owns a block of at most ``block`` consecutive rows, and a block that holds
    only WHOLE graphs never waits for another block.  The order in which a batch lists its graphs is the collation's to
    choose (a replay batch is a random draw; the reference's loss is a mean over it), so the graphs are packed:

    * graphs above ``block`` rows first, each in blocks of its own (``blocks_for_order``: a ``head``-row piece + the rest);
    * the other graphs best-fit-decreasing into blocks; a block is as long as what it holds (no padding rows).

    Returns ``(order, starts)``: ``order[k]`` = index (into ``sizes``) of the graph at position k of the packed batch,
    ``starts`` = row offsets of the blocks (``len(starts) - 1`` blocks, ``starts[-1] == sum(sizes)``).  Deterministic.  With
    ``max_blocks`` (the workgroups that can be resident at once) the large graphs fall back to ``block``-row pieces whose
    partial last block is filled up with small graphs; if that still needs more blocks: ``(identity, None)`` -- the kernels'
    default blocks of exactly ``block`` rows need the fewest."""
    n_graphs = len(sizes)
    if n_graphs == 0:
        return [], None
    sizes = [int(v) for v in sizes]
    if max_blocks is not None and -(-sum(sizes) // block) > max_blocks:
        return list(range(n_graphs)), None           # (not even full blocks fit the budget: nothing to pack for)
    big = sorted((g for g in range(n_graphs) if sizes[g] > block), key=lambda g: (-sizes[g], g))
    small = sorted((g for g in range(n_graphs) if sizes[g] <= block), key=lambda g: (-sizes[g], g))

    def build(hd: int):
        bins = []            # [free rows, big graph whose tail opens the bin or None, small graphs]
        order = []
        for g in big:
            if hd:
                order.append(g)
            else:
                tail = sizes[g] % block
                bins.append([block - tail if tail else 0, g, []])
        for g in small:
            best = None
            for b in bins:                       # best fit: the fullest bin that still takes the graph
                if b[0] >= sizes[g] and (best is None or b[0] < best[0]):
                    best = b
            if best is None:
                best = [block, None, []]
                bins.append(best)
            best[0] -= sizes[g]
            best[2].append(g)
        for _free, bg, members in bins:
            if bg is not None:
                order.append(bg)
            order.extend(members)
        return order, blocks_for_order([sizes[g] for g in order], block, hd)

    order, starts = build(head if head and head > 0 else 0)
    if max_blocks is not None and len(starts) - 1 > max_blocks:
        order, starts = build(0)
        if len(starts) - 1 > max_blocks:
            return list(range(n_graphs)), None
    return order, starts


def attach_blocks(edge_index: torch.Tensor, starts: Optional[List[int]]) -> None:
    """Attach a row-block table (``pack_order``'s ``starts``) to a collated batch's ``edge_index`` (device tensor): the model
    forward hands it to the one-launch stack kernels."""
    if starts is None or len(starts) < 2:
        return
    edge_index._hex_blocks = (torch.tensor(starts, dtype=torch.int32).to(edge_index.device), len(starts) - 1)


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
    def from_data_list(cls, data_list: List[Data], pack: bool = False, max_blocks: Optional[int] = None) -> "Batch":
        """``pack=True`` (graphs above 128 nodes in the batch -- Hex-12 and larger, mixed sizes): the graphs are collated in
        ``pack_order`` order with its row-block table attached, and ``batch.order`` (LongTensor [num_graphs]) says which entry
        of ``data_list`` sits at each position -- per-graph quantities of the caller (actions, targets, weights) go through
        ``t[batch.order]``.  Without ``pack`` the order is the caller's and the table follows its graph boundaries when that fits
        (``blocks_for_order``).  ``max_blocks``: see ``pack_order`` (default: ``ops.stack_block_budget``, the CUs the one-launch
        kernels may fill)."""
        if hasattr(data_list, "to_batch"):      # an Env_manager observation is already batched on the device
            return data_list.to_batch()
        out = cls()
        if len(data_list) == 0:
            raise ValueError("empty data_list")
        starts = None
        dev0 = data_list[0].x.device
        sizes = [int(d.x.shape[0]) for d in data_list]
        if dev0.type == "cuda" and max(sizes) > BLOCK_ROWS:
            if max_blocks is None:
                from . import ops
                max_blocks = ops.stack_block_budget(dev0)
            if pack:
                order, starts = pack_order(sizes, BLOCK_ROWS, max_blocks)
                data_list = [data_list[g] for g in order]
                out.order = torch.tensor(order, dtype=torch.long)
            elif -(-sum(sizes) // BLOCK_ROWS) <= max_blocks:
                # the caller's order is kept; blocks follow the graph boundaries of THAT order where the budget allows (more,
                # shorter blocks than a packing needs: the Hex-5..13 round robin takes 254 of 256)
                starts = blocks_for_order(sizes)
                if len(starts) - 1 > max_blocks:
                    starts = None
        elif pack:
            out.order = torch.arange(len(sizes))
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
        if starts is not None and out.edge_index.is_cuda:
            attach_blocks(out.edge_index, starts)
        return out

    @property
    def num_graphs(self) -> int:
        if "_num_graphs" in self.__dict__:
            return self._num_graphs
        return int(self.ptr.numel()) - 1
