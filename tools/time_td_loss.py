"""Kernel time of ops.td_loss forward / backward for a few (n, k): python tools/time_td_loss.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_hex_amd import ops

def t(n, k, dup=False, reps=200):
    q = torch.randn(n, device="cuda", requires_grad=True)
    sel = torch.randint(0, n, (k,), device="cuda")
    if dup:
        sel[: k // 2] = sel[k // 2: 2 * (k // 2)]
    tgt = torch.randn(k, device="cuda")
    loss, _ = ops.td_loss(q, sel, tgt)
    g = torch.ones_like(loss)
    for _ in range(10):
        torch.autograd.grad(loss, q, g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.autograd.grad(loss, q, g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print("n=%6d k=%5d dup=%d: backward %.2f us per call (incl. launch gaps)" % (n, k, dup, e0.elapsed_time(e1) / reps * 1e3))

for n, k in ((31488, 256), (31488, 16), (1024, 256), (13056, 256), (31488, 1024), (200000, 256)):
    t(n, k)
t(31488, 256, dup=True)

# prioritized-replay tree update with and without duplicates (kernel times: run under rocprofv3 --kernel-trace --stats)
from gnn_hex_amd import _lib
L = _lib.lib()
cap = 1 << 18
st = torch.empty(2 * cap, dtype=torch.float64, device="cuda"); mt = torch.empty_like(st)
_lib.check(L.hexgnn_per_init(cap, st.data_ptr(), mt.data_ptr(), ops._stream()))
for k in (256, 2048):
    idx = torch.randint(0, 260000, (k,), dtype=torch.int32, device="cuda")
    pa = torch.rand(k, dtype=torch.float64, device="cuda")
    for _ in range(50):
        _lib.check(L.hexgnn_per_update(cap, k, idx.data_ptr(), pa.data_ptr(), st.data_ptr(), mt.data_ptr(), ops._stream()))
torch.cuda.synchronize()
