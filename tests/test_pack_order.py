"""gnn_hex_amd.data.pack_order: the graph order + row-block table of a packed batch (host logic, no GPU).

Properties the one-launch stack kernels rely on (include/hexgnn.h, hexgnn_sage_stack_forward_blocks): the blocks partition the
rows, none is longer than 128, and only graphs above 128 rows are cut -- at their own multiples of 128."""
import random

import pytest

from gnn_hex_amd.data import blocks_for_order, pack_order


def _check(sizes, order, starts, block=128):
    assert sorted(order) == list(range(len(sizes)))
    assert starts[0] == 0 and starts[-1] == sum(sizes)
    lens = [b - a for a, b in zip(starts, starts[1:])]
    assert all(0 < v <= block for v in lens)
    pos, cuts = 0, set(starts)
    for g in order:
        lo, hi = pos, pos + sizes[g]
        inner = sorted(c for c in cuts if lo < c < hi)
        if sizes[g] <= block:
            assert not inner, "a graph of %d rows is cut" % sizes[g]
        else:
            assert lo in cuts, "a large graph opens a block"
            pieces = [b - a for a, b in zip([lo] + inner, inner + [hi])]
            if hi in cuts and pieces[0] == 64:
                assert max(pieces[1:]) - min(pieces[1:]) <= 1      # a 64-row head + equal pieces, blocks of its own
            else:
                assert all(p == block for p in pieces[:-1])        # 128-row pieces + a tail (whose block may hold more graphs)
        pos = hi


def test_mix_batch_packs_into_fewer_blocks_than_workgroups_available():
    sizes = [(5 + g % 9) ** 2 + 2 for g in range(256)]          # BASELINE config 5: Hex-5..13 round robin
    order, starts = pack_order(sizes)
    _check(sizes, order, starts)
    assert len(starts) - 1 <= 256                                 # 178 is the floor (22 778 rows); 256 CUs on the device
    assert pack_order(sizes) == (order, starts)                   # deterministic
    # under a tighter budget: large graphs in 128-row pieces whose tail block takes small graphs
    order2, starts2 = pack_order(sizes, max_blocks=200)
    _check(sizes, order2, starts2)
    assert len(starts2) - 1 <= 200


@pytest.mark.parametrize("seed", range(6))
def test_random_sizes(seed):
    rng = random.Random(seed)
    sizes = [rng.choice([3, 17, 27, 51, 64, 100, 128, 129, 171, 256, 300, 402]) for _ in range(rng.randint(1, 90))]
    order, starts = pack_order(sizes)
    _check(sizes, order, starts)


def test_uniform_small_and_exact_multiples():
    order, starts = pack_order([51] * 10)
    _check([51] * 10, order, starts)
    assert len(starts) - 1 == 5                                   # two 51-row graphs per block
    order, starts = pack_order([256, 128, 128])
    _check([256, 128, 128], order, starts)
    assert starts == [0, 64, 160, 256, 384, 512]
    order, starts = pack_order([256, 128, 128], max_blocks=4)
    assert starts == [0, 128, 256, 384, 512]


def test_block_budget_and_empty():
    sizes = [123] * 300
    order, starts = pack_order(sizes, max_blocks=256)             # 300 single-graph blocks do not fit 256 workgroups
    assert starts is None and order == list(range(300))
    assert pack_order([], max_blocks=4) == ([], None)


def test_blocks_for_a_given_order():
    """No reordering (a next-state batch must keep its state batch's order): consecutive whole graphs share blocks."""
    sizes = [171, 27, 27, 123, 51, 51, 51, 200, 128, 1]
    starts = blocks_for_order(sizes)
    assert starts == [0, 64, 171, 225, 348, 450, 501, 565, 633, 701, 829, 830]
    _check(sizes, list(range(len(sizes))), starts)
    assert blocks_for_order([171, 27], head=0) == [0, 128, 198]          # 128-row pieces, the tail block takes the next graph
    assert blocks_for_order([]) == [0]
