#!/usr/bin/env python
"""Eager forward + backward step time of model variants the bench has no configuration for (--norm=True, --noisy_dqn=True,
Hex-13, hidden 128 / 160 / 256): python tools/time_variants.py [plain norm noisy norm+noisy S-norm hex13 wide]"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from argparse import Namespace
from helpers import batch_tensors, sel_and_targets
from gnn_hex_amd.models import get_pre_defined
from gnn_hex_amd import ops
def run(name, norm, noisy, sizes, layers=15, hidden=110):
    args = Namespace(num_layers=layers, hidden_channels=hidden, norm=norm, noisy_dqn=noisy, noisy_sigma0=0.5, num_head_layers=2)
    torch.manual_seed(0)
    m = get_pre_defined("modern_two_headed", args).cuda()
    x, ei, bv, ptr = batch_tensors("D0", sizes, maker=True)
    sel, tgt = sel_and_targets(ptr)
    x, ei, bv, ptr, sel, tgt = (t.cuda() for t in (x, ei, bv, ptr, sel, tgt))
    ops.attach_hints(x, True, int((ptr[1:]-ptr[:-1]).max()))
    ei._hex_grouped = True
    def step():
        m.zero_grad(set_to_none=True)
        q = m(x, ei, bv, ptr)
        loss, _ = ops.td_loss(q, sel, tgt)
        loss.backward()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-28s %.3f ms/step  %.0f graphs/s" % (name, dt * 1e3, len(sizes) / dt))
if __name__ == "__main__":
    which = sys.argv[1:] or ["plain", "norm", "noisy", "norm+noisy", "S-norm", "hex13", "wide"]
    if "plain" in which: run("L256 plain", False, False, [11] * 256)
    if "norm" in which: run("L256 norm", True, False, [11] * 256)
    if "noisy" in which: run("L256 noisy", False, True, [11] * 256)
    if "norm+noisy" in which: run("L256 norm+noisy", True, True, [11] * 256)
    if "S-norm" in which: run("S256 norm", True, False, [7] * 256, 10, 35)
    if "hex13" in which: run("Hex-13 x256 plain", False, False, [13] * 256)
    if "wide" in which:
        run("L256 hidden 128 (layer-major)", False, False, [11] * 256, 15, 128)
        run("L256 hidden 160 (wide.hip)", False, False, [11] * 256, 15, 160)
        run("L256 hidden 256 (wide.hip)", False, False, [11] * 256, 15, 256)
