"""CPU: known answers for the prioritized-replay oracle (oracle/replay_ref.py)."""
import numpy as np

from oracle.replay_ref import SegmentTreePER


def test_tree_sums_and_prefix_search_by_hand():
    t = SegmentTreePER(4)
    t.update([0, 1, 2, 3], [1.0, 2.0, 3.0, 4.0])
    assert t.sum[1] == 10.0 and t.sum[2] == 3.0 and t.sum[3] == 7.0 and t.min[1] == 1.0
    # B = 4 strata of width 2.5: masses 0.5*2.5=1.25 -> leaf 1 ([1,3)), 2.5+1.25=3.75 -> leaf 2 ([3,6)),
    # 5+1.25=6.25 -> leaf 3 ([6,10)), 7.5+1.25=8.75 -> leaf 3
    idx, w = t.sample([0.5, 0.5, 0.5, 0.5], size=4, beta=1.0)
    assert idx.tolist() == [1, 2, 3, 3]
    # w_i = (N p_i)^-1 / (N p_min)^-1 = p_min / p_i
    assert np.allclose(w, [1 / 2, 1 / 3, 1 / 4, 1 / 4])
    idx, w = t.sample([0.0, 0.99], size=4, beta=0.5)        # masses 0 -> leaf 0, 5 + 4.95 = 9.95 -> leaf 3
    assert idx.tolist() == [0, 3] and np.allclose(w, [1.0, (1 / 4) ** 0.5])


def test_update_overwrites_and_partial_fill():
    t = SegmentTreePER(8)
    t.update([0, 1, 2], [1.0, 1.0, 1.0])
    t.update([1], [4.0])
    assert t.sum[1] == 6.0 and t.min[1] == 1.0
    idx, _ = t.sample(np.full(6, 0.5), size=3, beta=0.4)
    # strata of width 1: masses .5 1.5 2.5 3.5 4.5 5.5 over cumulative [0,1) [1,5) [5,6)
    assert idx.tolist() == [0, 1, 1, 1, 1, 2]
    # empirical frequencies follow p_i / total
    rng = np.random.default_rng(0)
    cnt = np.zeros(3)
    for _ in range(400):
        i, _ = t.sample(rng.random(8), size=3, beta=0.4)
        np.add.at(cnt, i, 1)
    assert np.allclose(cnt / cnt.sum(), [1 / 6, 4 / 6, 1 / 6], atol=0.02)
