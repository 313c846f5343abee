"""What would graph-aligned 128-row blocks buy the one-launch stack kernels?  (experiment, not part of the product)

The layer-major stack kernels cut the batch's rows into fixed 128-row blocks; a block whose rows all belong to graphs that lie
inside it never waits for another block.  This script times the SAME GNN-L step (layered path forced) on
  (a) B Hex-11 boards, 123 rows each              -- every block cuts a graph (the MIX situation),
  (b) the same boards + 5 isolated rows each = 128 -- every block is exactly one graph: no row crosses a block,
and prints the stack kernels' average launch times (hexgnn_profile_*).  (b) has 4 % more rows; the difference beyond that is
the price of the cross-block hand-off.

    python tools/exp_aligned_blocks.py [B]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def padded_batch(size, B, pad_to, maker, dev):
    from helpers import batch_tensors, sel_and_targets
    x, ei, bv, ptr = batch_tensors("D0", [size] * B, maker=maker)
    nv = int(ptr[1])
    if pad_to:
        assert pad_to >= nv
        e_per = ei.shape[1] // B
        xs = torch.zeros((B, pad_to, x.shape[1]), dtype=x.dtype)
        xs[:, :nv] = x.view(B, nv, -1)
        xs[:, nv:, 2] = x[0, 2]
        x = xs.view(B * pad_to, -1).contiguous()
        g = torch.arange(B).repeat_interleave(e_per)
        ei = ei - g * nv + g * pad_to
        bv = torch.arange(B).repeat_interleave(pad_to)
        ptr = torch.arange(B + 1) * pad_to
    sel, tgt = sel_and_targets(ptr, seed=1)
    xd = x.to(dev)
    xd._hex_is_maker = maker
    xd._hex_max_nodes = 200            # (a lie: keeps the batch off the fused per-graph kernels)
    eid = ei.to(dev)
    eid._hex_grouped = True
    return dict(x=xd, ei=eid, bv=bv.to(dev), ptr=ptr.to(dev), sel=sel.to(dev), tgt=tgt.to(dev), n=int(x.shape[0]))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 178
    import gnn_hex_amd  # noqa: F401
    from gnn_hex_amd import _lib, ops
    from gnn_hex_amd.graphs import GraphedStep
    from helpers import make_pair
    dev = torch.device("cuda:0")
    L = _lib.lib()
    hip, _ = make_pair(15, 110, seed=0, device=dev)
    plist = list(hip.parameters())
    runs = []
    for name, pad in (("123 rows per graph (blocks cut graphs)", 0), ("128 rows per graph (aligned blocks)", 128),
                      ("123 again", 0)):
        bt = padded_batch(11, B, pad, True, dev)

        def fn(bt=bt):
            for p in plist:
                p.grad = None
            return ops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"], target=bt["tgt"])[0]

        step = GraphedStep(fn, plist, warmup=3)
        for _ in range(300):
            step.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            step.replay()
        e1.record()
        torch.cuda.synchronize()
        runs.append((name, bt, fn, e0.elapsed_time(e1) / 200))
        del step
    # per-kernel HIP-event passes on eager steps (after ALL captures: graphs.py, AccumulateGrad caveat)
    for name, bt, fn, ms in runs:
        out = ["%-42s n=%d  step %.1f us" % (name, bt["n"], ms * 1e3)]
        for cls, kn in ((0, "stack fwd"), (1, "stack bwd"), (2, "dW"), (7, "csr+pack")):
            L.hexgnn_profile_enable(cls)
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            cnt, tot = C.c_int(0), C.c_float(0)
            L.hexgnn_profile_read(C.byref(cnt), C.byref(tot))
            L.hexgnn_profile_enable(-1)
            out.append("%s %.1f us x %.1f/step" % (kn, tot.value * 1e3 / max(cnt.value, 1), cnt.value / 20))
        print("  |  ".join(out), flush=True)


if __name__ == "__main__":
    main()
