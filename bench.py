#!/usr/bin/env python
"""bench.py -- board-graphs/s, forward + backward, GNN-L (modern_two_headed, 15 layers, hc=110) on a
batch of 256 Hex-11 board graphs per GPU (BASELINE.json metric).

One step = CSR build from edge_index + Q = model(x, edge_index, batch, ptr) + mse(Q[sel], tgt) +
loss.backward() (all parameter gradients), maker and breaker batches alternating per step; on N > 1
GPUs additionally ONE flat RCCL all-reduce of the gradients (weak scaling: 256 graphs per GPU).
Inputs (synthetic D0 start positions, or D1 random playouts with --data D1) are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config L256|S256|MIX] [--data D0|D1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # ROCm 7.2 hipGraph bug, see gnn_hex_amd/graphs.py (--graph)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3

CONFIGS = {
    # name: (num_layers, hidden, sizes-per-graph fn, label)
    "L256": (15, 110, lambda b: [11] * b, "GNN-L Hex-11 L=15 hc=110 batch=256"),
    "S256": (10, 35, lambda b: [7] * b, "GNN-S Hex-7 L=10 hc=35 batch=256"),
    "MIX": (15, 110, lambda b: [5 + (g % 9) for g in range(b)], "GNN-L mixed Hex-5..13 ragged batch=256"),
}

KNAMES = {0: "sage_hidden_fwd_kernel", 1: "sage_hidden_bwd_kernel", 2: "sage_dw_kernel",
          8: "qnet_fwd_kernel", 9: "qnet_bwd_kernel"}


def bytes_fwd(n, e, c):
    """algorithmic aggregation bytes of one SAGE layer with input width c."""
    return e * (4 * c + 4) + 4 * (n + 1) + 4 * n * c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="L256", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--data", default="D0", choices=["D0", "D1"])
    ap.add_argument("--math", default="fp32", choices=["fp32", "f16x3"],
                    help="contraction arithmetic of the fused kernels: exact fp32 MFMA (default) or split f16x3")
    ap.add_argument("--mode", default="train", choices=["train", "selfplay"],
                    help="train: the BASELINE metric (default).  selfplay: closed device loop env -> Q-network -> "
                         "epsilon-greedy -> env step, reports frames/s (secondary metric, SURVEY 8d)")
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank path "
                         "on a box with fewer GPUs than ranks)")
    ap.add_argument("--graph", action="store_true",
                    help="capture each step (maker batch / breaker batch) into a HIP graph and replay it; the gradient "
                         "all-reduce stays outside the graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the informational split-precision timing (clean profiles)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus=%d" % (world, args.gpus))
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPUs (RCCL needs one GPU per rank)" % (world, ndev))
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from helpers import batch_tensors, make_pair, sel_and_targets
    from gnn_hex_amd import _lib
    from gnn_hex_amd.dist import GradSync
    from gnn_hex_amd import ops as hexops
    hexops.set_math(args.math)

    num_layers, hidden, sizes_fn, label = CONFIGS[args.config]
    if args.mode == "selfplay":
        return selfplay(args, num_layers, hidden, label, dev)
    B = args.batch
    hip, ref = make_pair(num_layers, hidden, seed=0, device=dev)   # identical replicas on every rank
    sync = GradSync(hip.parameters())

    # two resident batches (maker to move / breaker to move), alternated per step; each rank draws its own
    # graphs for D1 (seed offset by rank), D0 is the same start position everywhere.
    batches = []
    for maker in (True, False):
        if args.data == "D0":
            x, ei, bv, ptr = batch_tensors("D0", sizes_fn(B), maker=maker)
        else:
            from oracle import env_ref
            import numpy as np
            xs, eis, bvs, ptrs, off = [], [], [], [0], 0
            for g, size in enumerate(sizes_fn(B)):
                game = env_ref.random_position(size, 100000 * rank + g, maker)
                gx, gei, _ = game.observe()
                xs.append(gx); eis.append(gei + off); bvs.append(np.full(gx.shape[0], g, dtype=np.int64))
                off += gx.shape[0]; ptrs.append(off)
            x, ei, bv, ptr = (torch.from_numpy(np.concatenate(xs, 0)), torch.from_numpy(np.concatenate(eis, 1)),
                              torch.from_numpy(np.concatenate(bvs)), torch.tensor(ptrs, dtype=torch.long))
        sel, tgt = sel_and_targets(ptr)
        xd = x.to(dev)
        xd._hex_is_maker = maker          # side to move known to the host (env / replay metadata)
        xd._hex_max_nodes = int((ptr[1:] - ptr[:-1]).max())   # largest graph, known from the board size
        eid = ei.to(dev)
        eid._hex_grouped = True           # collated graph by graph (what Batch.from_data_list produces and marks)
        batches.append(dict(x=xd, ei=eid, bv=bv.to(dev), ptr=ptr.to(dev), sel=sel.to(dev), tgt=tgt.to(dev),
                            cpu=(x, ei, bv, ptr, sel, tgt), n=int(x.shape[0]), e=int(ei.shape[1])))

    plist = list(hip.parameters())

    def step(i):
        bt = batches[i & 1]
        for p in plist:              # optimizer.zero_grad(set_to_none=True) over a cached parameter list
            p.grad = None
        q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
        loss, _ = hexops.td_loss(q, bt["sel"], bt["tgt"])
        loss.backward()
        if world > 1:
            sync.all_reduce()

    if args.graph:
        from gnn_hex_amd.graphs import GraphedStep

        def local_step(bt):
            def fn():
                for p in plist:
                    p.grad = None
                q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
                loss, _ = hexops.td_loss(q, bt["sel"], bt["tgt"])
                loss.backward()
                return loss
            return fn

        g0 = GraphedStep(local_step(batches[0]), plist)
        g1 = GraphedStep(local_step(batches[1]), plist, pool=g0.pool())
        graphs = (g0, g1)

        def step(i):
            graphs[i & 1].replay()
            if world > 1:
                sync.all_reduce()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    out = None
    if rank == 0:
        # ---- live per-kernel timing (HIP events on the launch stream) over the same steps ---------------
        L = _lib.lib()
        per_kernel = {}
        for cls in (0, 1, 2, 8, 9):
            L.hexgnn_profile_enable(cls)
            for i in range(args.steps):
                step_local(hip, batches, i)
            torch.cuda.synchronize()
            cnt, ms = C.c_int(0), C.c_float(0.0)
            _lib.check(L.hexgnn_profile_read(C.byref(cnt), C.byref(ms)), "profile_read")
            per_kernel[cls] = (cnt.value, ms.value)
        L.hexgnn_profile_enable(-1)
        n = (batches[0]["n"] + batches[1]["n"]) / 2.0
        e = (batches[0]["e"] + batches[1]["e"]) / 2.0
        per_kernel = {k: v for k, v in per_kernel.items() if v[0] > 0}
        dom = max(per_kernel, key=lambda k: per_kernel[k][1])
        launches, tot_ms = per_kernel[dom]
        avg_s = tot_ms / max(launches, 1) * 1e-3
        hidden_layers = num_layers + 1          # hidden-input SAGE layers per step: (L-1) body + 2 head
        if dom in (0, 1):   # gather kernels: HBM roofline on the un-fused aggregation bytes of ONE layer
            alg = bytes_fwd(n, e, hidden)          # bytes per launch (one layer, one direction)
            roof = dict(bound="hbm", achieved=alg / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        elif dom in (8, 9):  # fused per-graph kernels: one launch = all layers of one direction
            alg = bytes_fwd(n, e, 2) + hidden_layers * bytes_fwd(n, e, hidden)
            roof = dict(bound="hbm", achieved=alg / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        else:               # batched weight-gradient GEMM: fp32 MFMA roofline (flops per launch = per-step / launches)
            flops = 2.0 * n * (2 * hidden) * hidden * hidden_layers / (launches / args.steps)
            roof = dict(bound="mfma", achieved=flops / avg_s / 1e12, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        # measured HBM bytes per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs,
        # gfx950 FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes), committed under profiles/; same config only
        roof["traffic"] = None
        tpath = os.path.join(ROOT, "profiles", "r01", "traffic_pmc.json")
        if args.config == "L256" and args.data == "D0" and B == 256 and os.path.exists(tpath):
            mid = 1 if args.math == "f16x3" else 0
            want = {2: "sage_dw16_kernel<" if mid else "sage_dw_kernel<"}.get(dom, "%s<7, %d>" % (KNAMES[dom], mid))
            for k, v in json.load(open(tpath)).items():
                if want in k:
                    roof["traffic"] = v["hbm_bytes_per_launch"]
        roof["kernel"] = KNAMES[dom]
        roof["avg_launch_us"] = avg_s * 1e6
        roof["launches_per_step"] = launches / args.steps
        step_bytes = 2 * (bytes_fwd(n, e, 2) + (num_layers + 1) * bytes_fwd(n, e, hidden))
        roof["step_aggregation_GBps"] = step_bytes / (ms_per_step * 1e-3) / 1e9
        roof["kernel_ms_per_step"] = {KNAMES[k]: per_kernel[k][1] / args.steps for k in per_kernel}

        # informational, never `value`: the same steps in the opt-in split-precision arithmetic (three f16 MFMAs per
        # product on power-of-two scaled operands, 22-bit products, fp32 accumulate; tests hold it to the same 1e-4 bar and
        # measure it against a float64 oracle next to the exact-fp32 path)
        split = None
        if world == 1 and args.math == "fp32" and not args.no_split:
            hexops.set_math("f16x3")
            for i in range(args.warmup):
                step_local(hip, batches, i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step_local(hip, batches, i)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            hexops.set_math("fp32")
            split = {"math": "f16x3", "value": B * args.steps / dt2, "unit": "graphs/s", "ms_per_step": dt2 / args.steps * 1e3}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:      # rank 0 at N=1 only (bench contract)
            cpu = cpu_baseline(ref, batches, B)

        out = {
            "metric": "board-graphs/sec fwd+bwd", "value": value, "unit": "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.math == "fp32" else "f32 operands split into scaled f16 hi+lo (f16x3 MFMA, 22-bit products), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "%s, %s board graphs, %d graphs per GPU (N=%d nodes, E=%d directed edges)"
                                   % (label, "start-position" if args.data == "D0" else "random-playout",
                                      B, batches[0]["n"], batches[0]["e"]),
                       "parallelism": "dp%d" % world, "hip_graph": bool(args.graph), "global_batch": B * world},
            "roofline": roof, "cpu_baseline": cpu, "split_precision_mode": split,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def selfplay(args, num_layers, hidden, label, dev):
    """Closed device loop on one GPU: observation (HIP builder) -> Q-network forward (advantages only) -> epsilon-greedy
    (eps 0.05) -> env step with dead/captured removal and auto reset.  One step = one move in every env."""
    from helpers import make_pair
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import Env_manager
    size = 11 if args.config != "S256" else 7
    hip, _ = make_pair(num_layers, hidden, seed=0, device=dev)
    mgr = Env_manager(args.envs, size, gamma=0.97, device=dev)
    obs = mgr.reset()

    def one(obs):
        b = Batch.from_data_list(obs)
        with torch.no_grad():
            adv = hip(b.x, b.edge_index, b.batch, b.ptr, advantages_only=True)
        vert, _, _ = mgr.select_actions(adv, obs, eps=0.05)
        return mgr.step(vert)[0]

    if args.graph:
        # device-resident rollout: 16 moves of every env per HIP-graph launch, one read-back per launch
        from gnn_hex_amd.multi_env_manager import DeviceRollout
        chunk = 16
        ro = DeviceRollout(mgr, hip, steps=chunk, eps=0.05, graph=True)
        launches_w, launches = max(1, args.warmup // chunk), max(1, args.steps // chunk)
        for _ in range(launches_w):
            ro.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(launches):
            ro.run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        args.steps = launches * chunk
    else:
        for _ in range(args.warmup):
            obs = one(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            obs = one(obs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = {"metric": "self-play frames/sec (env step + observation + Q forward + action selection)",
           "value": args.envs * args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic (self-generated games)",
           "config": {"workload": "%s acting, Hex-%d, %d parallel envs" % (label.split(" batch")[0], size, args.envs),
                      "parallelism": "dp1", "hip_graph": bool(args.graph), "math": args.math}}
    print(json.dumps(out))
    return out


def step_local(hip, batches, i):
    from gnn_hex_amd import ops as hexops
    bt = batches[i & 1]
    for p in hip.parameters():
        p.grad = None
    q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
    hexops.td_loss(q, bt["sel"], bt["tgt"])[0].backward()


def cpu_baseline(ref, batches, B):
    """CPU restatement of the torch_geometric path (oracle/model_ref.py) timed on the host cores of this
    box, on a bounded sample: the first 32 graphs of the same batch, all torch threads."""
    import numpy as np
    sample_graphs = min(32, B)
    x, ei, bv, ptr, sel, tgt = batches[0]["cpu"]
    n_s = int(ptr[sample_graphs])
    keep = (ei[0] < n_s) & (ei[1] < n_s)
    xs, eis, bvs, ptrs = x[:n_s], ei[:, keep], bv[:n_s], ptr[:sample_graphs + 1]
    sels, tgts = sel[:sample_graphs], tgt[:sample_graphs]
    threads = torch.get_num_threads()

    def one():
        ref.zero_grad(set_to_none=True)
        q = ref(xs, eis, bvs, ptrs)
        torch.nn.functional.mse_loss(q[sels], tgts).backward()

    one()
    t0 = time.perf_counter()
    it = 0
    while True:
        one()
        it += 1
        el = time.perf_counter() - t0
        if el > 10.0 or it >= 20:
            break
    return {"value": sample_graphs * it / el, "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": "%d iterations of fwd+bwd on the first %d graphs of the same batch (%d nodes), "
                      "CPU restatement of the torch_geometric path, %d torch threads of %d host cpus"
                      % (it, sample_graphs, n_s, threads, os.cpu_count())}


if __name__ == "__main__":
    main()
