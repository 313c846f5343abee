#!/usr/bin/env python
"""Eager forward + backward step time of model variants the bench has no configuration for (--norm=True, --noisy_dqn=True,
Hex-13, hidden 128 / 160 / 256): python tools/time_variants.py [plain norm noisy norm+noisy S-norm hex13 wide hexara two_headed]"""
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
def run_hexara(b):
    """HexAra network (get_current_model() defaults: hidden 60, 15 + 2 + 2 layers): inference forward on b Hex-11 positions
    (the MCTS search threads' mini-batch) and a policy / value training step."""
    from gnn_hex_amd.torch_script_models import get_current_model
    torch.manual_seed(0)
    m = get_current_model().cuda()
    x, ei, bv, ptr = (t.cuda() for t in batch_tensors("D1", [11] * b, maker=True))
    tp = torch.rand(int(x.shape[0]) - 2 * b, device="cuda"); tv = torch.rand(b, device="cuda") * 2 - 1
    def fwd():
        with torch.no_grad():
            return m(x, ei, bv, ptr)
    def step():
        m.zero_grad(set_to_none=True)
        pi, v, _, _ = m(x, ei, bv, ptr)
        (-(pi * tp).sum() / b + torch.nn.functional.mse_loss(v, tv)).backward()
    for name, fn in (("forward", fwd), ("train step", step)):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print("HexAra SAGE b=%-4d %-10s  %.3f ms  %.0f positions/s" % (b, name, dt * 1e3, b / dt))
def run_two_headed(norm):
    args = Namespace(num_layers=13, hidden_channels=32, norm=norm, noisy_dqn=False, noisy_sigma0=0.5, num_head_layers=2)
    torch.manual_seed(0)
    m = get_pre_defined("two_headed", args).cuda()
    x, ei, bv, ptr = batch_tensors("D1", [11] * 256, maker=True)
    sel, tgt = sel_and_targets(ptr)
    x, ei, bv, ptr, sel, tgt = (t.cuda() for t in (x, ei, bv, ptr, sel, tgt))
    def step():
        m.zero_grad(set_to_none=True)
        q = m(x, ei, bv, ptr)
        loss, _ = ops.td_loss(q, sel, tgt)
        loss.backward()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print("two_headed L=13 H=32 norm=%-5s B=256  %.3f ms/step  %.0f graphs/s" % (norm, dt * 1e3, 256 / dt))
if __name__ == "__main__":
    which = sys.argv[1:] or ["plain", "norm", "noisy", "norm+noisy", "S-norm", "hex13", "wide"]
    if "plain" in which: run("L256 plain", False, False, [11] * 256)
    if "norm" in which: run("L256 norm", True, False, [11] * 256)
    if "noisy" in which: run("L256 noisy", False, True, [11] * 256)
    if "norm+noisy" in which: run("L256 norm+noisy", True, True, [11] * 256)
    if "S-norm" in which: run("S256 norm", True, False, [7] * 256, 10, 35)
    if "hex13" in which: run("Hex-13 x256 plain", False, False, [13] * 256)
    if "hexara" in which:
        run_hexara(64); run_hexara(512)
    if "two_headed" in which:
        run_two_headed(False); run_two_headed(True)
    if "wide" in which:
        run("L256 hidden 128 (layer-major)", False, False, [11] * 256, 15, 128)
        run("L256 hidden 160 (wide.hip)", False, False, [11] * 256, 15, 160)
        run("L256 hidden 256 (wide.hip)", False, False, [11] * 256, 15, 256)
