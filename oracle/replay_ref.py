"""CPU restatement of the prioritized-replay sampler (numpy, fp64).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's replay buffer is in the un-vendored submodule GN0/RainbowDQN/Rainbow
(.gitmodules:1-4, fork of schmidtdominik/Rainbow, itself built on the OpenAI-baselines segment trees); only the flags
are visible in /root/reference (README.md:5,7).  This file restates the published algorithm (Schaul et al. 2016,
proportional variant, baselines form) and is pinned by hand-worked known answers in tests/test_oracle_replay.py:

* priorities are stored as p^alpha in a sum tree and a min tree of capacity 2^k (node 1 = root, leaves [cap, 2cap));
* sample i of a batch of B draws mass = (i + u_i) * total / B and descends the sum tree (go left while left > mass);
* importance weight w_i = (N * p_i / total)^-beta / (N * p_min / total)^-beta, N = number of stored transitions.
"""
import numpy as np


class SegmentTreePER:
    def __init__(self, capacity_pow2: int):
        assert capacity_pow2 >= 1 and capacity_pow2 & (capacity_pow2 - 1) == 0
        self.cap = capacity_pow2
        self.sum = np.zeros(2 * capacity_pow2, dtype=np.float64)
        self.min = np.full(2 * capacity_pow2, np.inf, dtype=np.float64)

    def update(self, idx, prio_alpha):
        idx = np.asarray(idx, dtype=np.int64)
        pa = np.asarray(prio_alpha, dtype=np.float64)
        for i, p in zip(idx, pa):                 # later duplicates win, as in a sequential put
            self.sum[self.cap + i] = p
            self.min[self.cap + i] = p
        nodes = np.unique((self.cap + idx) >> 1)
        while nodes.size and nodes[0] >= 1:
            for n in nodes:
                self.sum[n] = self.sum[2 * n] + self.sum[2 * n + 1]
                self.min[n] = min(self.min[2 * n], self.min[2 * n + 1])
            nodes = np.unique(nodes >> 1)
            nodes = nodes[nodes >= 1]

    def sample(self, u, size, beta):
        u = np.asarray(u, dtype=np.float64)
        b = len(u)
        total = self.sum[1]
        idx = np.empty(b, dtype=np.int64)
        for i in range(b):
            mass = (i + u[i]) * (total / b)
            node = 1
            while node < self.cap:
                left = self.sum[2 * node]
                if left > mass:
                    node = 2 * node
                else:
                    mass -= left
                    node = 2 * node + 1
            idx[i] = min(node - self.cap, size - 1)
        p_min = self.min[1] / total
        max_w = (p_min * size) ** (-beta)
        w = ((self.sum[self.cap + idx] / total) * size) ** (-beta) / max_w
        return idx, w.astype(np.float32)
