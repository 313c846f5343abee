#!/usr/bin/env python
"""Per-wave timeline of the fused kernels (profiling builds only).

    make -C gnn_hex_amd/csrc clean && make -C gnn_hex_amd/csrc STAMPS=1
    python tools/stamps.py            # on the GPU box
    make -C gnn_hex_amd/csrc clean && make -C gnn_hex_amd/csrc      # back to the shipped build

Workgroup 0 records s_memtime at fixed points of every layer (QSTAMP in qnet_fused_kernels.h); this script runs
one GNN-L Hex-11 B=256 forward + backward, reads the stamps and prints, per kernel, for every wave the time of each
stamp point, averaged over the hidden layers, in s_memtime ticks from the layer's start (about 2.29 ticks per ns).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import batch_tensors, make_pair, sel_and_targets  # noqa: E402
from gnn_hex_amd import _lib  # noqa: E402
from gnn_hex_amd import ops as hexops  # noqa: E402

KMAXL, POINTS = 64, 10
NAMES = {0: "layer top", 1: "phase S done", 3: "before barrier 1", 4: "barrier 1", 5: "phase A MFMAs done", 6: "rows published",
         7: "barrier 2"}


def main():
    dev = torch.device("cuda", 0)
    cfg = sys.argv[1] if len(sys.argv) > 1 else "L256"          # L256 (GNN-L Hex-11) or S256 (GNN-S Hex-7)
    layers_, hidden_, size_ = (15, 110, 11) if cfg == "L256" else (10, 35, 7)
    hip, _ = make_pair(layers_, hidden_, seed=0, device=dev)
    x, ei, bv, ptr = batch_tensors("D0", [size_] * 256, maker=True)
    sel, tgt = sel_and_targets(ptr)
    xd, eid, bvd, ptrd, seld, tgtd = (t.to(dev) for t in (x, ei, bv, ptr, sel, tgt))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(5):
        hip.zero_grad(set_to_none=True)
        ev[0].record()
        q = hip(xd, eid, bvd, ptrd)
        loss, _ = hexops.td_loss(q, seld, tgtd)
        loss.backward()
        ev[1].record()
    torch.cuda.synchronize()
    lib = _lib.lib()
    fn = lib.hexgnn_debug_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int]
    total = 2 * (KMAXL + 2) * POINTS * 8
    buf = np.zeros(total, dtype=np.uint64)
    rc = fn(buf.ctypes.data, total)
    assert rc == total, rc
    st = buf.reshape(2, KMAXL + 2, POINTS, 8).astype(np.float64)
    L = layers_ + 2
    for k, kname in enumerate(["qnet_fwd_kernel", "qnet_bwd_kernel"]):
        s = st[k]
        t0 = s[0, 0].min()
        end = s[0, 2 if k == 0 else 3].max()
        # s_memtime ticks -> us: calibrated below against the span, printed raw as well
        print("== %s: start->end %.0f ticks" % (kname, end - t0))
        layers = range(1, L)
        rows = []
        for l in layers:
            base = s[l, 0].min()       # earliest wave entering the layer
            rows.append(s[l, :POINTS, :] - base)
        a = np.stack(rows)              # [layers][point][wave]
        lay_span = np.array([st[k][l, 7].max() - st[k][l, 0].min() for l in layers])
        print("   per-layer span (ticks): mean %.0f  min %.0f  max %.0f" % (lay_span.mean(), lay_span.min(), lay_span.max()))
        mean = a.mean(0)                # [point][wave]
        print("   point                 " + "  ".join("w%d    " % w for w in range(8)))
        for p in sorted(NAMES):
            print("   %-20s " % NAMES[p] + "  ".join("%6.0f" % mean[p, w] for w in range(8)))
        if k == 0:
            print("   prologue %.0f ticks, tail %.0f ticks" % (s[0, 1].max() - t0, end - s[L - 1, 7].max()))
        else:
            print("   prologue (head-tail bwd) %.0f, publish top %.0f, epilogue %.0f ticks" % (
                s[0, 1].max() - t0, s[0, 2].max() - s[0, 1].max(), end - s[1, 7].max()))
    print("wall of the last fwd+bwd step (events): %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))


if __name__ == "__main__":
    main()
