import os, sys, time, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
from helpers import batch_tensors, make_pair
from gnn_hex_amd import _lib
hip, ref = make_pair(15, 110)
x, ei, bv, ptr = batch_tensors("D0", [11] * 256)
xd = x.cuda(); xd._hex_is_maker = True; xd._hex_max_nodes = 123
ei, bv, ptr = ei.cuda(), bv.cuda(), ptr.cuda()
L = _lib.lib()
for dbg in [0, 1, 2, 4, 8, 1 | 2, 1 | 2 | 4, 8 | 1, 15, 0]:
    os.environ["HEXGNN_DBG_ABLATE"] = str(dbg)
    L.hexgnn_profile_enable(8)
    for _ in range(12):
        q = hip(xd, ei, bv, ptr)
        q.sum().backward()
    torch.cuda.synchronize()
    cnt, ms = C.c_int(0), C.c_float(0)
    L.hexgnn_profile_read(C.byref(cnt), C.byref(ms))
    print("dbg=%2d fwd kernel avg %.1f us" % (dbg, ms.value / cnt.value * 1e3))
