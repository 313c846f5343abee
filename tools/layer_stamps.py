#!/usr/bin/env python
"""Timeline of one workgroup of the layer-major kernels (profiling builds only: make -C gnn_hex_amd/csrc clean && make -C
gnn_hex_amd/csrc STAMPS=1; python tools/layer_stamps.py on the GPU box; then rebuild without STAMPS).  Runs one MIX
(Hex-5..13, 256 graphs) forward + backward on the layer-major path and prints, per wave of the mid-grid workgroup, the
s_memtime ticks (about the shader clock: 1.9 per ns in these runs) between the stamp points of the last hidden-layer launch of each kind."""
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

NAMES = ["start", "weights / self rows / ids landed", "self half + gather done", "barrier", "tail + aggregate half done", "epilogue stores issued"]


def main():
    dev = torch.device("cuda", 0)
    hip, _ = make_pair(15, 110, seed=0, device=dev)
    x, ei, bv, ptr = batch_tensors("D0", [5 + (g % 9) for g in range(256)], maker=True)
    sel, tgt = sel_and_targets(ptr)
    xd, eid, bvd, ptrd, seld, tgtd = (t.to(dev) for t in (x, ei, bv, ptr, sel, tgt))
    hexops.set_fused(False)
    for _ in range(5):
        hip.zero_grad(set_to_none=True)
        q = hip(xd, eid, bvd, ptrd)
        loss, _ = hexops.td_loss(q, seld, tgtd)
        loss.backward()
    torch.cuda.synchronize()
    fn = _lib.lib().hexgnn_debug_layer_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(384, dtype=np.uint64)
    assert fn(buf.ctypes.data, 384) == 384
    st = buf[:128].reshape(2, 8, 8).astype(np.float64)
    ps = buf[128:].reshape(2, 16, 8).astype(np.float64)
    PN = ["layer start", "stores acked / W_l landed / flag / remote wait (gap 3)", "self half + gather done", "long rows, aggregate stored",
          "W_l of the layer in place", "aggregate half done", "epilogue stores issued", "every self half over", "W_r pieces claimed + landed",
          "barrier 1", "rows -> LDS, W_l pieces issued", "barrier 2", "tail: other blocks' counters seen"]
    for k, name in enumerate(["sage_stack_fwd_kernel", "sage_stack_bwd_kernel"]):
        if ps[k].max() == 0:
            continue
        t0 = ps[k, 0][ps[k, 0] > 0].min()
        print("%s, layer 8: ticks from the workgroup's first stamp, per wave 0..7" % name)
        for p in range(13):
            if ps[k, p].max() == 0:
                continue                     # (a point this kernel's hand-over does not pass)
            print("  %-62s %s" % (PN[p], " ".join("%6d" % v for v in (ps[k, p] - t0))))
    for k, name in enumerate(["sage_hidden_fwd_kernel", "sage_hidden_bwd_kernel"]):
        t0 = st[k, 0].min()
        print("%s: ticks from the workgroup's first stamp, per wave 0..7" % name)
        for p in range(6):
            print("  %-62s %s" % (NAMES[p], " ".join("%6d" % v for v in (st[k, p] - t0))))


if __name__ == "__main__":
    main()
