#!/usr/bin/env python
"""Per-wave timeline of chosen workgroups of the one-launch stack kernels on the PACKED MIX batch (profiling builds only:
make -C gnn_hex_amd/csrc clean && make -C gnn_hex_amd/csrc STAMPS=1; rebuild without STAMPS afterwards).

    python tools/stack_stamps_blocks.py [block ...]      (default: the two pieces of the first Hex-13 graph, the two of a Hex-12
                                                          graph, and a block of whole small graphs)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_pair  # noqa: E402
from gnn_hex_amd import _lib  # noqa: E402
from gnn_hex_amd import ops as hexops  # noqa: E402

PN = ["layer start", "stores acked / W_l landed / flag / remote wait (gap 3)", "self half + gather done", "long rows, aggregate stored",
      "W_l of the layer in place", "aggregate half done", "epilogue stores issued", "every self half over", "W_r pieces claimed + landed",
      "barrier 1", "rows -> LDS, W_l pieces issued", "barrier 2", "tail: other blocks' counters seen"]


def main():
    dev = torch.device("cuda", 0)
    hip, _ = make_pair(15, 110, seed=0, device=dev)
    bt = bench.make_batches("MIX", "D0", 256, dev)[0]
    starts = bt["ei"]._hex_blocks[0].cpu().tolist()
    L = _lib.lib()
    L.hexgnn_debug_stamp_block.restype = C.c_int
    L.hexgnn_debug_stamp_block.argtypes = [C.c_int]
    fn = L.hexgnn_debug_layer_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int]
    blocks = [int(a) for a in sys.argv[1:]] or [0, 1, 60, 61, len(starts) - 20]
    for blk in blocks:
        L.hexgnn_debug_stamp_block(blk)
        for _ in range(3):
            hexops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"], target=bt["tgt"])
        torch.cuda.synchronize()
        buf = np.zeros(384, dtype=np.uint64)
        assert fn(buf.ctypes.data, 384) == 384
        ps = buf[128:].reshape(2, 16, 8).astype(np.float64)
        print("block %d: rows %d..%d (%d)" % (blk, starts[blk], starts[blk + 1], starts[blk + 1] - starts[blk]))
        for k, name in enumerate(["sage_stack_fwd_kernel", "sage_stack_bwd_kernel"]):
            if ps[k].max() == 0:
                continue
            act = ps[k, 0] > 0
            t0 = ps[k, 0][act].min()
            print(" %s, layer 8: ticks from the workgroup's first stamp, per wave" % name)
            for p in range(13):
                if ps[k, p].max() == 0:
                    continue
                print("  %-62s %s" % (PN[p], " ".join("%6d" % v if v > 0 else "     -" for v in np.where(ps[k, p] > 0, ps[k, p] - t0, 0))))


if __name__ == "__main__":
    main()
