"""Graph-aligned row blocks for the one-launch SAGE stack kernels (hexgnn_sage_stack_*_blocks, data.pack_order).

* a packed MIX batch with its block table gives the Q-values and gradients of the same batch on the default 128-row blocks
  (and of the per-graph reference order, graph by graph through ``batch.order``) to fp32 rounding, bit-identically run to run,
  also with the blocks delayed unevenly (the stress mode of tests/test_gpu_stack_stress.py);
* the table's content is device data: a table that is not a partition in pieces of at most 128 rows is caught by the kernel
  (HEXGNN_EINVAL through hexgnn_stack_status), nothing is read or written out of range."""
from argparse import Namespace

import pytest
import torch

from helpers import make_pair, sel_and_targets

pytestmark = pytest.mark.gpu


def _data_list(sizes, maker=True):
    from gnn_hex_amd.data import Data
    from helpers import batch_tensors
    out = []
    for s in sizes:
        x, ei, _, _ = batch_tensors("D0", [s], maker=maker)
        d = Data(x=x.cuda(), edge_index=ei.cuda())
        d.x._hex_is_maker = maker
        out.append(d)
    return out


def _step(model, bt, sel, tgt):
    model.zero_grad(set_to_none=True)
    q = model(bt.x, bt.edge_index, bt.batch, bt.ptr)
    torch.nn.functional.mse_loss(torch.as_tensor(q).reshape(-1)[sel], tgt).backward()
    torch.cuda.synchronize()
    return q.detach().clone(), [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]


def _close(a, b, tol=3e-5):
    scale = max(1.0, b.abs().max().item())
    return (a - b).abs().max().item() < tol * scale


def test_packed_batch_equals_default_blocks_and_caller_order():
    from gnn_hex_amd import _lib
    from gnn_hex_amd.data import Batch
    hip, _ = make_pair(5, 110, seed=3)
    sizes = [5 + (g % 9) for g in range(63)]
    dl = _data_list(sizes)
    plain = Batch.from_data_list(dl)
    packed = Batch.from_data_list(dl, pack=True)
    assert getattr(packed.edge_index, "_hex_blocks", None) is not None and sorted(packed.order.tolist()) == list(range(63))
    nb = packed.edge_index._hex_blocks[1]
    assert (int(plain.x.shape[0]) + 127) // 128 <= nb <= 256
    # per-graph selections / targets defined on the caller's order, carried to the packed order through batch.order
    sel0, tgt = sel_and_targets(plain.ptr.cpu(), seed=5)
    local = sel0 - plain.ptr.cpu()[:-1]
    order = packed.order
    sel_p = (packed.ptr.cpu()[:-1] + local[order]).cuda()
    tgt_p = tgt[order].cuda()
    q0, g0 = _step(hip, plain, sel0.cuda(), tgt.cuda())
    # (the caller's order gets a table too -- blocks along ITS graph boundaries, no reordering; same results as without it)
    tbl0 = plain.edge_index._hex_blocks
    assert tbl0[1] > nb and not hasattr(plain, "order")
    del plain.edge_index._hex_blocks
    q0b, g0b = _step(hip, plain, sel0.cuda(), tgt.cuda())
    assert _close(q0, q0b) and all(_close(a, b, 1e-4) for a, b in zip(g0, g0b))
    q1, g1 = _step(hip, packed, sel_p, tgt_p)
    # graph by graph: position k of the packed batch is graph order[k] of the caller's list
    pp, p0 = packed.ptr.tolist(), plain.ptr.tolist()
    for k, g in enumerate(order.tolist()):
        assert _close(q1[pp[k]:pp[k + 1]], q0[p0[g]:p0[g + 1]]), "graph %d" % g
    for a, b in zip(g1, g0):
        assert _close(a, b, 1e-4)
    # the same packed batch WITHOUT its table (default blocks) and on the per-layer launches
    tbl = packed.edge_index._hex_blocks
    del packed.edge_index._hex_blocks
    q2, g2 = _step(hip, packed, sel_p, tgt_p)
    packed.edge_index._hex_blocks = tbl
    assert _close(q1, q2) and all(_close(a, b, 1e-4) for a, b in zip(g1, g2))
    L = _lib.lib()
    try:
        L.hexgnn_debug_stack_mode(0, 0)
        q3, g3 = _step(hip, packed, sel_p, tgt_p)
    finally:
        L.hexgnn_debug_stack_mode(-1, 0)
    assert _close(q1, q3) and all(_close(a, b, 1e-4) for a, b in zip(g1, g3))
    # bit-identical run to run, also with the blocks delayed unevenly (a new pattern per launch)
    try:
        for seed in (0, 11, 12, 13):
            L.hexgnn_debug_stack_mode(-1, seed)
            q4, g4 = _step(hip, packed, sel_p, tgt_p)
            assert torch.equal(q4, q1) and all(torch.equal(a, b) for a, b in zip(g4, g1)), seed
    finally:
        L.hexgnn_debug_stack_mode(-1, 0)
    assert L.hexgnn_stack_status(1) == 0


def test_large_boards_only():
    """Hex-13 boards alone (171 rows: a 64-row head block + a 107-row block each)."""
    from gnn_hex_amd.data import Batch
    hip, _ = make_pair(4, 110, seed=4)
    dl = _data_list([13] * 24)
    plain, packed = Batch.from_data_list(dl), Batch.from_data_list(dl, pack=True)
    assert packed.edge_index._hex_blocks[1] == 48      # 64 + 107 rows each
    sel, tgt = sel_and_targets(plain.ptr.cpu(), seed=2)
    q0, g0 = _step(hip, plain, sel.cuda(), tgt.cuda())
    q1, g1 = _step(hip, packed, sel.cuda(), tgt.cuda())            # (equal graphs: the order is the identity up to ties)
    assert _close(q1, q0) and all(_close(a, b, 1e-4) for a, b in zip(g1, g0))


@pytest.mark.parametrize("bad", ["long_piece", "not_ascending", "short_end", "bad_start"])
def test_malformed_block_table_is_caught_by_the_kernel(bad):
    from gnn_hex_amd import _lib
    from gnn_hex_amd.data import Batch, attach_blocks
    hip, _ = make_pair(3, 110, seed=5)
    dl = _data_list([5 + (g % 9) for g in range(36)])
    bt = Batch.from_data_list(dl, pack=True)
    t, nb = bt.edge_index._hex_blocks
    starts = t.cpu().tolist()
    n = starts[-1]
    if bad == "long_piece":
        starts[3] = starts[2] + 129
    elif bad == "not_ascending":
        starts[5], starts[6] = starts[6], starts[5]
    elif bad == "short_end":
        starts[-1] = n - 7
    else:
        starts[0] = 3
    attach_blocks(bt.edge_index, starts)
    L = _lib.lib()
    L.hexgnn_stack_status(1)
    with torch.no_grad():
        hip(bt.x, bt.edge_index, bt.batch, bt.ptr)
    torch.cuda.synchronize()
    assert L.hexgnn_stack_status(1) == -1          # HEXGNN_EINVAL
    # and the library is fine afterwards
    good = Batch.from_data_list(dl, pack=True)
    with torch.no_grad():
        q = hip(good.x, good.edge_index, good.batch, good.ptr)
    torch.cuda.synchronize()
    assert torch.isfinite(torch.as_tensor(q)).all() and L.hexgnn_stack_status(1) == 0


def test_replay_draws_of_large_boards_are_packed():
    """GraphReplayBuffer on Hex-13 boards (171 nodes at the start): a draw comes in pack order with block tables on both
    batches; indices, actions, rewards, done flags and the two batches stay aligned; the model's Q-values on the packed batch
    equal those of the same transitions collated in the draw's order without tables."""
    import numpy as np
    from gnn_hex_amd import _lib
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    from test_gpu_replay import _play
    rng = np.random.default_rng(7)
    mgr = Env_manager(6, 13, gamma=0.97, n_steps=[1])
    obs0, states, actions, rewards, dones, expl = _play(mgr, 40, rng)
    maker, _ = mgr.get_transitions(obs0, states, actions, rewards, dones, expl)
    assert len(maker) >= 48
    buf = GraphReplayBuffer(256, 13, prioritized=False)
    buf.put(maker)
    idx, w, s, s2, a, r, d = buf.sample(48)
    ih = idx.cpu().tolist()
    for bt, col in ((s, 0), (s2, 3)):
        blk = bt.edge_index._hex_csr.blocks
        assert blk is not None
        starts = blk[0].cpu().tolist()
        assert starts[0] == 0 and starts[-1] == int(bt.x.shape[0]) and all(0 < q - p <= 128 for p, q in zip(starts, starts[1:]))
        ref = Batch.from_data_list([maker[i][col] for i in ih])
        assert torch.equal(bt.x, ref.x) and torch.equal(bt.edge_index, ref.edge_index) and torch.equal(bt.ptr, ref.ptr)
    assert a.cpu().tolist() == [int(maker[i][1]) for i in ih]
    assert np.allclose(r.cpu().numpy(), [maker[i][2] for i in ih])
    sizes = (s.ptr[1:] - s.ptr[:-1]).cpu().tolist()
    nbig = sum(v > 128 for v in sizes)
    assert all(v > 128 for v in sizes[:nbig]) and all(v <= 128 for v in sizes[nbig:])      # large graphs first
    hip, _ = make_pair(4, 110, seed=6)
    with torch.no_grad():
        q1 = hip(s.x, s.edge_index, s.batch, s.ptr).clone()
        ref = Batch.from_data_list([maker[i][0] for i in ih])
        q0 = hip(ref.x, ref.edge_index, ref.batch, ref.ptr)
    torch.cuda.synchronize()
    assert _close(torch.as_tensor(q1), torch.as_tensor(q0)) and _lib.lib().hexgnn_stack_status(1) == 0


def test_table_over_the_block_budget_falls_back_to_default_blocks():
    """CUs reserved for kernels running beside the stack kernels (GradSync.enable_overlap) shrink the resident-workgroup budget:
    a table with more blocks than that is ignored (default 128-row blocks, still one launch when those fit), never launched."""
    from gnn_hex_amd import _lib, ops
    from gnn_hex_amd.data import Batch
    hip, _ = make_pair(3, 110, seed=8)
    dl = _data_list([5 + (g % 9) for g in range(63)])
    bt = Batch.from_data_list(dl, pack=True)
    nb = bt.edge_index._hex_blocks[1]
    dflt = (int(bt.x.shape[0]) + 127) // 128
    assert nb > dflt + 2
    L = _lib.lib()
    prev = L.hexgnn_stack_reserve_cus(0)          # (an earlier test may have left CUs reserved: start from none, restore below)
    budget0 = ops.stack_block_budget(bt.x.device)
    assert budget0 >= nb
    with torch.no_grad():
        q0 = torch.as_tensor(hip(bt.x, bt.edge_index, bt.batch, bt.ptr)).clone()
    try:
        L.hexgnn_stack_reserve_cus(budget0 - (dflt + 1))
        assert ops.stack_block_budget(bt.x.device) == dflt + 1
        with torch.no_grad():
            q1 = torch.as_tensor(hip(bt.x, bt.edge_index, bt.batch, bt.ptr)).clone()
        # a batch collated now packs against the smaller budget: no table that cannot run
        again = Batch.from_data_list(dl, pack=True)
        blk = getattr(again.edge_index, "_hex_blocks", None)
        assert blk is None or blk[1] <= dflt + 1
    finally:
        L.hexgnn_stack_reserve_cus(prev)
    torch.cuda.synchronize()
    assert _close(q1, q0) and L.hexgnn_stack_status(1) == 0


@pytest.mark.parametrize("sizes", [[5 + (g % 9) for g in range(63)], [13] * 20 + [5] * 7, [12] * 30, [15, 14, 6, 6, 6, 13]])
def test_device_built_table_equals_blocks_for_order(sizes):
    """Raw tensors of another collation (no table, the host never saw the sizes): the CSR launch builds the row-block table on
    the device, in the batch's own order -- entry for entry what data.blocks_for_order gives, padded with empty blocks -- and the
    forward / backward on it equal those on the attached table bit for bit."""
    from gnn_hex_amd import ops
    from gnn_hex_amd.data import Batch, blocks_for_order
    hip, _ = make_pair(3, 110, seed=11)
    dl = _data_list(sizes)
    bt = Batch.from_data_list(dl)
    want = blocks_for_order([int(d.x.shape[0]) for d in dl])
    n = int(bt.x.shape[0])
    sel, tgt = sel_and_targets(bt.ptr.cpu(), seed=3)
    q0, g0 = _step(hip, bt, sel.cuda(), tgt.cuda())                   # the attached table
    assert bt.edge_index._hex_blocks[0].cpu().tolist() == want
    del bt.edge_index._hex_blocks
    q1, g1 = _step(hip, bt, sel.cuda(), tgt.cuda())                   # no table: built on the device
    tbl, nb = hip._fca.gs.blocks
    got = tbl.cpu().tolist()
    budget = ops.stack_block_budget(bt.x.device)
    assert nb == budget and len(got) == budget + 1
    assert got[:len(want)] == want and all(v == n for v in got[len(want):])
    assert torch.equal(q1, q0) and all(torch.equal(a, b) for a, b in zip(g1, g0))


def test_device_built_table_falls_back_to_plain_blocks_over_budget():
    """More aligned blocks than the budget: the device writes the plain 128-row partition (which the caller made sure fits)."""
    from gnn_hex_amd import _lib, ops
    from gnn_hex_amd.data import Batch
    hip, _ = make_pair(3, 110, seed=12)
    dl = _data_list([5 + (g % 9) for g in range(63)])
    bt = Batch.from_data_list(dl)
    del bt.edge_index._hex_blocks
    n = int(bt.x.shape[0])
    dflt = (n + 127) // 128
    L = _lib.lib()
    with torch.no_grad():
        q0 = torch.as_tensor(hip(bt.x, bt.edge_index, bt.batch, bt.ptr)).clone()
    prev = L.hexgnn_stack_reserve_cus(0)
    try:
        L.hexgnn_stack_reserve_cus(ops.stack_block_budget(bt.x.device) - (dflt + 1))
        with torch.no_grad():
            q1 = torch.as_tensor(hip(bt.x, bt.edge_index, bt.batch, bt.ptr)).clone()
        got = hip._fca.gs.blocks[0].cpu().tolist()
    finally:
        L.hexgnn_stack_reserve_cus(prev)
    assert got == [min(128 * i, n) for i in range(dflt + 1)] + [n]
    assert _close(q1, q0) and L.hexgnn_stack_status(1) == 0
