"""Host-side profile of the eager training step (cProfile over N steps of `bench.py`'s step on one config): where the
Python time of a launch-bound configuration (GNN-S B=256: kernels 0.13 ms, step 0.6 ms) goes.
    python tools/host_profile.py [S256|L256] [steps] [plain]"""
import cProfile
import os
import pstats
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
    cfg = sys.argv[1] if len(sys.argv) > 1 else "S256"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    layers, hidden, size = (10, 35, 7) if cfg == "S256" else (15, 110, 11)
    hip, _ = make_pair(layers, hidden, seed=0, device="cuda")
    x, ei, bv, ptr = batch_tensors("D0", [size] * 256)
    sel, tgt = sel_and_targets(ptr)
    xd = ops.attach_hints(x.cuda(), True, int((ptr[1:] - ptr[:-1]).max()))
    eid = ei.cuda()
    eid._hex_grouped = True
    bvd, ptrd, seld, tgtd = bv.cuda(), ptr.cuda(), sel.cuda(), tgt.cuda()
    plist = list(hip.parameters())

    plain = len(sys.argv) > 3 and sys.argv[3] == "plain"      # the step as an unmodified train.py writes it

    def step():
        for p in plist:
            p.grad = None
        q = hip(xd, eid, bvd, ptrd)
        if plain:
            torch.nn.functional.mse_loss(q[seld], tgtd).backward()
            return
        loss, _ = ops.td_loss(q, seld, tgtd)
        ops.backward(loss)

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s eager: %.1f us/step (%.0f graphs/s)" % (cfg, dt / steps * 1e6, 256 * steps / dt))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("host-side issue time: %.1f us/step" % (host / steps * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(35)
    st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
