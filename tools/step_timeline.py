"""How the eager step time evolves from a cold start: consecutive windows of 20 steps (sync per window), then the same
after idle gaps -- separates launch / allocator warm-up from clock behaviour (DVFS) on the box.
    python tools/step_timeline.py [L256|S256] [fp32|f16x3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

from helpers import batch_tensors, make_pair, sel_and_targets  # noqa: E402
from gnn_hex_amd import ops  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "L256"
    ops.set_math(sys.argv[2] if len(sys.argv) > 2 else "fp32")
    layers, hidden, size = (10, 35, 7) if cfg == "S256" else (15, 110, 11)
    hip, _ = make_pair(layers, hidden, seed=0, device="cuda")
    batches = []
    for maker in (True, False):
        x, ei, bv, ptr = batch_tensors("D0", [size] * 256, maker=maker)
        sel, tgt = sel_and_targets(ptr)
        xd = ops.attach_hints(x.cuda(), maker, int((ptr[1:] - ptr[:-1]).max()))
        eid = ei.cuda()
        eid._hex_grouped = True
        batches.append((xd, eid, bv.cuda(), ptr.cuda(), sel.cuda(), tgt.cuda()))
    plist = list(hip.parameters())

    def step(i):
        bt = batches[i & 1]
        for p in plist:
            p.grad = None
        q = hip(bt[0], bt[1], bt[2], bt[3])
        loss, _ = ops.td_loss(q, bt[4], bt[5])
        loss.backward()

    def window(k=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    for i in range(5):
        step(i)
    out = [window() for _ in range(30)]
    print("%s %s cold start, 30 windows of 20 steps (ms/step):" % (cfg, ops.get_math()))
    print(" ".join("%.3f" % v for v in out))
    for gap in (0.05, 0.3, 1.0, 3.0):
        time.sleep(gap)
        out = [window() for _ in range(4)]
        print("after %.2f s idle:" % gap, " ".join("%.3f" % v for v in out))
    # one long window for reference
    print("200-step window: %.3f" % window(200))


if __name__ == "__main__":
    main()
