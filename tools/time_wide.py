"""Step time of the hidden 129..256 path (csrc/wide.hip): GNN-L-shaped model at a given width, Hex-11, B = 256, eager
forward + TD loss + backward.    python tools/time_wide.py [hidden] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import batch_tensors, make_pair, sel_and_targets  # noqa: E402
from gnn_hex_amd import ops  # noqa: E402


def main():
    hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    hip, _ = make_pair(15, hidden, seed=0, device="cuda")
    x, ei, bv, ptr = batch_tensors("D0", [11] * 256)
    sel, tgt = sel_and_targets(ptr)
    xd = ops.attach_hints(x.cuda(), True, int((ptr[1:] - ptr[:-1]).max()))
    eid = ei.cuda()
    eid._hex_grouped = True
    bvd, ptrd, seld, tgtd = bv.cuda(), ptr.cuda(), sel.cuda(), tgt.cuda()
    plist = list(hip.parameters())

    def step():
        for p in plist:
            p.grad = None
        q = hip(xd, eid, bvd, ptrd)
        loss, _ = ops.td_loss(q, seld, tgtd)
        loss.backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = int(x.shape[0])
    flops = 3 * 2.0 * n * 16 * hidden * 2 * hidden            # fwd + bwd data + dW over 16 hidden-input layers (logical width)
    print("hidden %d: %.3f ms per step (%.0f graphs/s, %.1f TFLOP/s of contractions)" % (hidden, dt * 1e3, 256 / dt, flops / dt * 1e-12))


if __name__ == "__main__":
    main()
