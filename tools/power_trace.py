"""Clocks / power while the training step runs back to back for a few seconds in each arithmetic mode (rocm-smi sampled
from a side thread): evidence for the sustained-load behaviour DESIGN.md 7.2 describes.
    python tools/power_trace.py [seconds_per_mode]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnn_hex_amd  # noqa: F401  (graph env switch before HIP starts)
import torch
from helpers import batch_tensors, make_pair, sel_and_targets
from gnn_hex_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
hip, _ = make_pair(15, 110, seed=0, device="cuda")
x, ei, bv, ptr = batch_tensors("D0", [11] * 256)
sel, tgt = sel_and_targets(ptr)
xd = ops.attach_hints(x.cuda(), True, 123); eid = ei.cuda(); eid._hex_grouped = True
bvd, ptrd, seld, tgtd = bv.cuda(), ptr.cuda(), sel.cuda(), tgt.cuda()
plist = list(hip.parameters())

def step():
    for p in plist: p.grad = None
    q = hip(xd, eid, bvd, ptrd)
    ops.td_loss(q, seld, tgtd)[0].backward()

samples, stop = [], False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out.strip().splitlines()[-1] if out.strip() else ""))
        except Exception as exc:  # noqa: BLE001
            samples.append((time.perf_counter(), "error %s" % exc))
        time.sleep(0.25)

hdr = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print("rocm-smi columns:", hdr[0] if hdr else "(none)")
for mode in ("fp32", "f16x3", "fp32"):
    ops.set_math(mode)
    for _ in range(5): step()
    torch.cuda.synchronize(); time.sleep(2.0)          # idle gap: boost budget refilled
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); wins = []
    while time.perf_counter() - t0 < secs:
        tw = time.perf_counter()
        for _ in range(50): step()
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0, (time.perf_counter() - tw) / 50 * 1e3))
    stop = True; th.join()
    print("== %s: ms/step per 50-step window (t s: ms)" % mode)
    print(" ".join("%.2f:%.3f" % w for w in wins))
    for ts, line in samples:
        print("   t=%.2f %s" % (ts - t0, line))
