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
# useful-FLOP peak of each arithmetic: exact fp32 MFMA; split modes issue 3 (f16x3) f16 MFMAs (2516 TFLOP/s dense) per product
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "f16x3": 2516.6 / 3}
PROFILE_ROUND = "r04"

CONFIGS = {
    # name: (num_layers, hidden, sizes-per-graph fn, label)
    "L256": (15, 110, lambda b: [11] * b, "GNN-L Hex-11 L=15 hc=110 batch=256"),
    "S256": (10, 35, lambda b: [7] * b, "GNN-S Hex-7 L=10 hc=35 batch=256"),
    "MIX": (15, 110, lambda b: [5 + (g % 9) for g in range(b)], "GNN-L mixed Hex-5..13 ragged batch=256"),
}

PACK_BATCHES = True      # --no-pack: batches with graphs above 128 nodes keep the round-robin order (default 128-row blocks)

KNAMES = {0: "sage_hidden_fwd_kernel", 1: "sage_hidden_bwd_kernel", 2: "sage_dw_kernel",
          8: "qnet_fwd_kernel", 9: "qnet_bwd_kernel"}
# classes 0 / 1 time whichever kernel runs the hidden layers of the layer-major path: one launch per layer, or (batch fits one
# resident workgroup per CU) ONE launch for all of them -- told apart by the launches per step
KNAMES_STACK = {0: "sage_stack_fwd_kernel", 1: "sage_stack_bwd_kernel"}


def make_batches(config, data, B, dev, rank=0, subset=None, world=1):
    """The two resident batches (maker to move / breaker to move) of a configuration.  Weak scaling: every rank holds its own
    B graphs (D1 graphs differ per rank).  Strong scaling (`subset` = this rank's graph indices of ONE global B-graph batch,
    gnn_hex_amd.dist.balance_by_edges): the rank holds only those graphs; D1 seeds are the global graph indices."""
    from helpers import batch_tensors, sel_and_targets
    from gnn_hex_amd.data import attach_blocks, blocks_for_order, pack_order
    sizes_fn = CONFIGS[config][2]
    all_sizes = sizes_fn(B)
    graphs = list(range(B)) if subset is None else list(subset)
    sizes = [all_sizes[g] for g in graphs]
    # Batches with graphs above 128 nodes (MIX: Hex-12 / Hex-13) run on the one-launch stack kernels, whose workgroups own
    # blocks of at most 128 rows: the collation lists the graphs in gnn_hex_amd.data.pack_order order (blocks of whole graphs
    # wherever possible; the same multiset of graphs -- what Batch.from_data_list(pack=True) does) unless --no-pack
    want_pack = PACK_BATCHES and max(sizes) ** 2 + 2 > 128
    from gnn_hex_amd import ops as _ops
    cus = _ops.stack_block_budget(dev) if torch.device(dev).type == "cuda" else None      # CUs minus those reserved (overlap)
    batches = []
    for maker in (True, False):
        starts = None
        if data == "D0":
            if want_pack:
                order, starts = pack_order([s * s + 2 for s in sizes], max_blocks=cus)
                sizes_m = [sizes[k] for k in order]
            else:
                sizes_m = sizes
                if max(sizes) ** 2 + 2 > 128 and cus:
                    # the caller's order: blocks along ITS graph boundaries when they fit (Batch.from_data_list does the same)
                    st2 = blocks_for_order([s_ * s_ + 2 for s_ in sizes])
                    starts = st2 if len(st2) - 1 <= cus else None
            x, ei, bv, ptr = batch_tensors("D0", sizes_m, maker=maker)
        else:
            from oracle import env_ref          # input generation only (before any timed region)
            import numpy as np
            items = []
            for g, size in zip(graphs, sizes):
                game = env_ref.random_position(size, (100000 * rank + g) if subset is None else g, maker)
                gx, gei, _ = game.observe()
                items.append((gx, gei))
            if want_pack:
                order, starts = pack_order([it[0].shape[0] for it in items], max_blocks=cus)
                items = [items[k] for k in order]
            elif max(it[0].shape[0] for it in items) > 128 and cus:
                st2 = blocks_for_order([it[0].shape[0] for it in items])
                starts = st2 if len(st2) - 1 <= cus else None
            xs, eis, bvs, ptrs, off = [], [], [], [0], 0
            for gx, gei in items:
                bvs.append(np.full(gx.shape[0], len(xs), dtype=np.int64)); xs.append(gx); eis.append(gei + off)
                off += gx.shape[0]; ptrs.append(off)
            x, ei, bv, ptr = (torch.from_numpy(np.concatenate(xs, 0)), torch.from_numpy(np.concatenate(eis, 1)),
                              torch.from_numpy(np.concatenate(bvs)), torch.tensor(ptrs, dtype=torch.long))
        sel, tgt = sel_and_targets(ptr, seed=1 + rank)       # every rank regresses on its own targets
        # strong scaling: the loss is the mean over the GLOBAL batch -- a rank's mean over its k graphs times k * world / B,
        # so that the gradient average over the ranks is the global mean even when the edge-balanced parts differ in count
        wts = None if subset is None else torch.full((len(graphs),), float(len(graphs)) * world / B)
        xd = x.to(dev)
        xd._hex_is_maker = maker          # side to move known to the host (env / replay metadata)
        xd._hex_max_nodes = int((ptr[1:] - ptr[:-1]).max())   # largest graph, known from the board size
        eid = ei.to(dev)
        eid._hex_grouped = True           # collated graph by graph (what Batch.from_data_list produces and marks)
        # ... with the collation's per-graph edge offsets (Batch.from_data_list attaches them; here the graphs were concatenated by
        # hand): edges are grouped by graph, so graph g's range is the running count of the edges whose source lies in it
        ecnt = torch.bincount(bv[ei[0]], minlength=int(ptr.numel()) - 1)
        eid._hex_edge_ptr = torch.cat([torch.zeros(1, dtype=torch.long), ecnt.cumsum(0)]).to(dev)
        if starts is not None and eid.is_cuda:
            attach_blocks(eid, starts)
        batches.append(dict(x=xd, ei=eid, bv=bv.to(dev), ptr=ptr.to(dev), sel=sel.to(dev), tgt=tgt.to(dev),
                            w=None if wts is None else wts.to(dev), graphs=len(graphs),
                            blocks=None if starts is None else len(starts) - 1, packed=bool(want_pack and starts is not None),
                            cpu=(x, ei, bv, ptr, sel, tgt), n=int(x.shape[0]), e=int(ei.shape[1])))
    return batches


def secondary_config(config, data, B, dev, steps, warmup, preheat_ms, _order_too=True):
    """One more BASELINE configuration measured the same way as the headline (HIP-graph replay, preheat, W warm-up steps,
    K timed steps), so that the driver's record carries it too.  Returns a small dict; never raises.  A configuration whose
    batch is collated packed (MIX) is measured in the caller's round-robin order as well (`callers_order`)."""
    try:
        from helpers import make_pair
        from gnn_hex_amd import ops as hexops
        from gnn_hex_amd.graphs import GraphedStep
        num_layers, hidden, _, label = CONFIGS[config]
        hip, _ = make_pair(num_layers, hidden, seed=0, device=dev)
        batches = make_batches(config, data, B, dev)
        plist = list(hip.parameters())

        def local_step(bt):
            def fn():
                for p in plist:
                    p.grad = None
                if True:      # (the fused form of the three calls; falls back by itself where it does not apply, e.g. MIX)
                    return hexops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"], target=bt["tgt"])[0]
                q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
                loss, _ = hexops.td_loss(q, bt["sel"], bt["tgt"])
                hexops.backward(loss)
                return loss
            return fn

        g0 = GraphedStep(local_step(batches[0]), plist)
        g1 = GraphedStep(local_step(batches[1]), plist, pool=g0.pool())
        graphs = (g0, g1)
        t_pre, k = time.perf_counter(), 0
        while (time.perf_counter() - t_pre) * 1e3 < preheat_ms:
            for i in range(16):
                graphs[i & 1].replay()
            torch.cuda.synchronize()
            k += 16
        for i in range(warmup):
            graphs[i & 1].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            graphs[i & 1].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = (batches[0]["n"] + batches[1]["n"]) / 2.0
        e = (batches[0]["e"] + batches[1]["e"]) / 2.0
        step_bytes = 2 * (bytes_fwd(n, e, 2) + (num_layers + 1) * bytes_fwd(n, e, hidden))
        out = {"workload": "%s, %s board graphs (N=%d, E=%d)%s" % (
                   label, "start-position" if data == "D0" else "random-playout", batches[0]["n"], batches[0]["e"],
                   "" if batches[0]["blocks"] is None else
                   ", %s (%d graph-aligned row blocks)" % ("graphs collated in data.pack_order order" if batches[0]["packed"]
                                                           else "caller's graph order", batches[0]["blocks"])),
               "value": B * steps / dt, "unit": "graphs/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
               "warmup": warmup, "preheat_steps": k,
               "step_hbm_roofline_frac": step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS}
        if batches[0]["packed"] and _order_too:
            global PACK_BATCHES
            del graphs, g0, g1, batches
            keep, PACK_BATCHES = PACK_BATCHES, False
            try:
                alt = secondary_config(config, data, B, dev, steps, warmup, min(preheat_ms, 300.0), _order_too=False)
            finally:
                PACK_BATCHES = keep
            out["callers_order"] = {k2: alt[k2] for k2 in ("value", "ms_per_step", "error") if k2 in alt}
        return out
    except Exception as exc:  # noqa: BLE001  (a secondary line must never take the headline down)
        return {"workload": "%s %s" % (config, data), "error": "%s: %s" % (type(exc).__name__, exc)}


def flops_fwd(n, b, h, layers):
    """dense FLOPs of one forward (= of one backward data chain, = of the weight-gradient GEMMs)."""
    return 4.0 * n * 2 * h + (layers + 1) * 4.0 * n * h * h + 2.0 * n * h + 2.0 * b * (4 * h * (h // 2) + h // 2)


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
    ap.add_argument("--no-pack", action="store_true",
                    help="MIX: keep the round-robin graph order (default: gnn_hex_amd.data.pack_order, graph-aligned row blocks)")
    ap.add_argument("--math", default="fp32", choices=["fp32", "f16x3"],
                    help="contraction arithmetic of the fused kernels: exact fp32 MFMA (default) or split f16x3")
    ap.add_argument("--mode", default="train", choices=["train", "selfplay"],
                    help="train: the BASELINE metric (default).  selfplay: closed device loop env -> Q-network -> "
                         "epsilon-greedy -> env step, reports frames/s (secondary metric, SURVEY 8d)")
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank path "
                         "on a box with fewer GPUs than ranks)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="capture each step (maker batch / breaker batch) into a HIP graph and replay it (gnn_hex_amd.graphs."
                         "GraphedStep: every kernel of the step still runs per replay); the gradient all-reduce stays "
                         "outside the graph.  Default for --mode train: the step is a fixed sequence of ~10 launches, and "
                         "replaying it removes the host from the loop (GNN-S is host-bound in eager mode)")
    ap.add_argument("--eager", dest="graph", action="store_false",
                    help="issue every step through Python / torch.autograd (what an unmodified train.py does); with N > 1 "
                         "this also overlaps the all-reduce with the backward")
    ap.add_argument("--preheat-ms", type=float, default=400.0,
                    help="untimed device preheat before the W warm-up steps: the step is repeated until this much wall time "
                         "has passed, so that the clocks, the caching allocator and the TLBs are in their steady state when "
                         "the K timed steps start (a cold start runs its first 20 steps ~5 %% slower); 0 disables it")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the BASELINE metric): every GPU steps its own --batch graphs.  strong: ONE global "
                         "--batch-graph batch per step, split over the GPUs by edge count (gnn_hex_amd.dist.balance_by_edges); "
                         "value = --batch graphs per step time")
    ap.add_argument("--sustain-s", type=float, default=2.0,
                    help="after the K timed steps, repeat the step for at least this many seconds and report that window as "
                         "`sustained` (never `value`); 0 disables it")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: one all-reduce after the backward instead of the staged backward whose first gradient "
                         "segment is reduced while the rest of the weight-gradient GEMM computes")
    ap.add_argument("--split-graph", action="store_true",
                    help="capture the step as TWO graphs split at the staged backward's hand-over and start the first gradient "
                         "segment's all-reduce between their replays (N > 1: the collective overlaps the second graph's "
                         "weight-gradient GEMM); default: one graph, one all-reduce behind it")
    ap.add_argument("--no-td-step", action="store_true",
                    help="issue the step as model(...), ops.td_loss, ops.backward (three calls, a TD-loss launch between the "
                         "network's two) instead of ops.td_step, which forms the same loss in the forward kernel's tail")
    ap.add_argument("--plain-autograd", action="store_true",
                    help="with --eager: the step exactly as an unmodified train.py issues it -- q[sel] by torch indexing, "
                         "F.mse_loss, loss.backward() through the autograd engine (no gnn_hex_amd.ops call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-collective-probe", action="store_true",
                    help="N = 1: skip the 1-rank RCCL all-reduce latency probe of the gradient bucket (config.collective)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra lines for the other single-GPU BASELINE configurations (D1 boards, GNN-S, MIX) that "
                         "the default N=1 run appends as `other_configs`")
    ap.add_argument("--no-split", action="store_true", help="skip the informational split-precision timing (clean profiles)")
    args = ap.parse_args()
    if args.graph is None:
        args.graph = args.mode == "train"
    if args.no_pack:
        global PACK_BATCHES
        PACK_BATCHES = False

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: bring up the N ranks ourselves, as fresh child processes.  This process calls NOTHING in
        # torch.cuda (device_count() may fall back to hipGetDeviceCount and bring the HIP runtime up): it only waits and
        # passes rank 0's JSON line through.  The visible-GPU check runs inside every rank.
        from gnn_hex_amd.dist import launch_ranks
        raise SystemExit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    ndev = torch.cuda.device_count()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus=%d: refusing to report a line whose n_gpus is not the requested one"
                         % (world, args.gpus))
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but only %d GPU(s) visible (RCCL needs one GPU per rank; --backend gloo rehearses the "
                         "multi-rank path on fewer GPUs)" % (world, ndev))
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)

    def init_group():
        """Called AFTER the step graphs are captured: the replicas are built from the same seed on every rank (nothing to
        broadcast), and no collective library thread is alive while a HIP-graph capture is in progress."""
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
    overlap = world > 1 and not args.no_overlap and not args.graph
    if overlap:
        sync.enable_overlap()

    # two resident batches (maker to move / breaker to move), alternated per step; each rank draws its own
    # graphs for D1 (seed offset by rank), D0 is the same start position everywhere.
    subset = None
    strong = args.scaling == "strong"
    if strong:
        # ONE global batch; graph g's directed edge count from the closed form of its board size (start positions) -- the
        # partition must be known to every rank without building the other ranks' graphs; mid-game boards keep the same
        # order of sizes, so the start-position counts balance them as well
        from gnn_hex_amd.dist import balance_by_edges
        def edges_of(nn):
            return 2 * (2 * nn + (nn - 2) * (nn - 1) + nn * (nn - 1) + (nn - 1) ** 2)
        parts = balance_by_edges([edges_of(sz) for sz in sizes_fn(B)], world)
        subset = parts[rank]
    batches = make_batches(args.config, args.data, B, dev, rank, subset, world)
    gfactor = 1 if strong else world          # graphs per step over all ranks = B * gfactor

    plist = list(hip.parameters())
    td_fused = not args.no_td_step and not overlap and not args.plain_autograd
    if args.plain_autograd and args.graph:
        raise SystemExit("--plain-autograd is an --eager measurement (the step an unmodified train.py issues)")

    def step(i):
        bt = batches[i & 1]
        for p in plist:              # optimizer.zero_grad(set_to_none=True) over a cached parameter list
            p.grad = None
        if args.plain_autograd:
            # an UNMODIFIED train.py: torch indexing, torch's loss, loss.backward() -- no gnn_hex_amd.ops call in the step
            q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
            d = q[bt["sel"]]
            loss = torch.nn.functional.mse_loss(d, bt["tgt"]) if bt["w"] is None else (bt["w"] * (d - bt["tgt"]) ** 2).mean()
            loss.backward()
        elif td_fused:
            # the same three calls in their fused form (ops.td_step: the loss is formed in the forward kernel's tail)
            hexops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"], target=bt["tgt"], weights=bt["w"])
        else:
            q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
            loss, _ = hexops.td_loss(q, bt["sel"], bt["tgt"], bt["w"])
            hexops.backward(loss)   # == loss.backward(), minus autograd's ones-fill and the TD scatter launch (ops.backward)
        if world > 1:
            sync.all_reduce()

    if args.graph:
        from gnn_hex_amd.graphs import GraphedStep

        def local_step(bt):
            def fn():
                for p in plist:
                    p.grad = None
                if td_fused:
                    return hexops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"], target=bt["tgt"],
                                          weights=bt["w"])[0]
                q = hip(bt["x"], bt["ei"], bt["bv"], bt["ptr"])
                loss, _ = hexops.td_loss(q, bt["sel"], bt["tgt"], bt["w"])
                hexops.backward(loss)
                return loss
            return fn

        split = args.split_graph
        if split:
            # opt-in: the step as TWO graphs split at the staged backward's hand-over, so that the all-reduce of the finished
            # gradient segment (RCCL, its own stream) travels beside the second graph's weight-gradient GEMM.  Not the default:
            # measured at N = 1 the split itself costs 0.682 -> 0.725 ms per step (two half-size weight-gradient launches at
            # twice the slices each, two slab reduces, a second graph launch: profiles/r04), i.e. about what hiding a
            # latency-bound 2-MB all-reduce can win back -- DESIGN.md section 6
            from gnn_hex_amd.graphs import GraphedSplitStep
            from gnn_hex_amd import _lib as hexlib
            hexlib.lib().hexgnn_stack_reserve_cus(64)      # (the one-launch stack kernels leave the RCCL channels their CUs)

            def first_of(bt):
                def fn():
                    for p in plist:
                        p.grad = None
                    loss, _, _, call = hexops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"],
                                                      target=bt["tgt"], weights=bt["w"], defer_lower=True)
                    return loss, call
                return fn

            g0 = GraphedSplitStep(first_of(batches[0]), hexops.finish_backward, plist)
            g1 = GraphedSplitStep(first_of(batches[1]), hexops.finish_backward, plist, pool=g0.pool())
            graphs = (g0, g1)
            overlap = world > 1

            def step(i):
                gr = graphs[i & 1]
                gr.replay_first()
                if gr.flat is not None and world > 1:
                    sync.reduce_segment(gr.flat, gr.cut, gr.total)
                gr.replay_second()
                if world > 1:
                    if gr.flat is not None:
                        sync.reduce_segment(gr.flat, 0, gr.cut)
                    sync.all_reduce()
        else:
            g0 = GraphedStep(local_step(batches[0]), plist)
            g1 = GraphedStep(local_step(batches[1]), plist, pool=g0.pool())
            graphs = (g0, g1)

            def step(i):
                graphs[i & 1].replay()
                if world > 1:
                    sync.all_reduce()

    with _stdout_to_stderr():
        init_group()
    # the first step of every run verifies with one tiny collective that all ranks reduce the SAME parameter set (same side to
    # move): ranks that disagree would otherwise hang or mix the two heads' gradients (GradSync check)
    sync.check = world > 1
    if world > 1:
        with _stdout_to_stderr():          # (the communicator is created lazily by the first collective)
            step(0)
            torch.cuda.synchronize()
        sync.check = False

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    note("setup done (%s, %d ranks); preheat + warm-up" % (label, world))
    preheat_steps = 0
    if args.preheat_ms > 0:
        # same number of steps on every rank (collectives inside): chunks of 16 steps until rank 0's clock says stop
        t_pre = time.perf_counter()
        while True:
            for i in range(16):
                step(i)
            preheat_steps += 16
            torch.cuda.synchronize()
            go = torch.tensor([1.0 if (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms else 0.0])
            if world > 1:
                go = go.to(dev) if args.backend == "nccl" else go
                dist.broadcast(go, 0)
            if float(go.item()) == 0.0 or preheat_steps >= 4096:
                break
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * gfactor * args.steps / dt

    # what holds under load: the same step for >= `--sustain-s` seconds (same count on every rank), reported beside the
    # K-step line (`value` stays the driver's protocol).  The short window can sit inside a boost period of the box.
    sustained = None
    if args.sustain_s > 0:
        ks = max(args.steps, int(args.sustain_s / (dt / args.steps)) + 1)
        barrier()
        t1 = time.perf_counter()
        for i in range(ks):
            step(i)
        barrier()
        dts = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sustained = {"seconds": dts, "steps": ks, "ms_per_step": dts / ks * 1e3, "value": B * gfactor * ks / dts}

    # N > 1: the replicas must stay identical -- one more step, a plain SGD update from the all-reduced gradients on every
    # rank, then the parameter checksums of all ranks are compared (a rank that reduced a different bucket would diverge)
    replicas_identical = None
    if world > 1:
        step(0)
        with torch.no_grad():
            for p in plist:
                if p.grad is not None:
                    p.add_(p.grad, alpha=-1e-3)
            flat = torch.cat([p.detach().double().reshape(-1) for p in plist])
            cs = torch.stack([flat.sum(), flat.abs().sum(), (flat * flat).sum()])
        if args.backend == "gloo":
            cs = cs.cpu()
        gathered = [torch.empty_like(cs) for _ in range(world)]
        dist.all_gather(gathered, cs)
        replicas_identical = all(torch.equal(g, gathered[0]) for g in gathered)
        if not replicas_identical:
            raise SystemExit("rank %d: parameter replicas diverged after the gradient all-reduce: %s" % (rank, gathered))
        sync.enable_overlap(False)       # the rank-0-only passes below must not start collectives

    out = None
    if rank == 0:
        note("timed region done: %.4f ms/step; per-kernel HIP-event passes" % ms_per_step)
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
        layers_per_launch = 1.0
        if dom in (0, 1):   # gather kernels: HBM roofline on the un-fused aggregation bytes of the layers ONE launch runs
            layers_per_launch = max(1.0, round(hidden_layers / max(launches / args.steps, 1e-9)))
            alg = layers_per_launch * bytes_fwd(n, e, hidden)      # bytes per layer and direction x layers per launch
            roof = dict(bound="hbm", achieved=alg / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        elif dom in (8, 9):  # fused per-graph kernels: one launch = all layers of one direction
            alg = bytes_fwd(n, e, 2) + hidden_layers * bytes_fwd(n, e, hidden)
            roof = dict(bound="hbm", achieved=alg / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        else:               # batched weight-gradient GEMM: fp32 MFMA roofline (flops per launch = per-step / launches)
            flops = 2.0 * n * (2 * hidden) * hidden * hidden_layers / (launches / args.steps)
            roof = dict(bound="mfma", achieved=flops / avg_s / 1e12, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        # the same launch priced against the OTHER roof as well: the fused kernels keep a graph's rows in LDS, so their HBM
        # traffic is far below the un-fused algorithmic bytes and the binding roof is the MFMA pipe of the arithmetic in use
        launch_flops = (flops_fwd(n, batches[0]["graphs"], hidden, num_layers) if dom in (8, 9)
                        else 2.0 * n * (2 * hidden) * hidden * (hidden_layers / max(launches / args.steps, 1)) if dom == 2
                        else 4.0 * n * hidden * hidden * layers_per_launch)
        mfma_peak = MFMA_PEAK_TFLOPS[args.math]
        roof["mfma_tflops"] = launch_flops / avg_s / 1e12
        roof["mfma_peak_tflops"] = mfma_peak
        roof["mfma_frac"] = roof["mfma_tflops"] / mfma_peak
        roof["binding_roof"] = "mfma"     # refined below once the measured HBM traffic of the launch is known
        # measured HBM bytes per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs,
        # gfx950 FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes), committed under profiles/; same config only
        roof["traffic"] = None
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic_pmc.json")
        if not os.path.exists(tpath):       # counters not re-collected this round yet: the previous round's passes
            tpath = os.path.join(ROOT, "profiles", "r02", "traffic_pmc.json")
        kname = KNAMES_STACK[dom] if (dom in (0, 1) and layers_per_launch > 1) else KNAMES[dom]
        if args.data == "D0" and B == 256 and os.path.exists(tpath):
            mid = 1 if args.math == "f16x3" else 0
            table = json.load(open(tpath))
            table = table.get(args.config, table if args.config == "L256" and "L256" not in table else {})   # (r04: per configuration)
            want = {2: "sage_dw16_kernel<" if mid else "sage_dw_kernel<"}.get(dom, kname + "<")
            for k, v in table.items():
                if want in k and (dom == 2 or dom in (0, 1) or k.rstrip(">").endswith(", %d" % mid)):
                    roof["traffic"] = v["hbm_bytes_per_launch"]
        if roof["traffic"]:
            roof["measured_hbm_frac"] = roof["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS
            roof["binding_roof"] = "mfma" if roof["mfma_frac"] >= roof["measured_hbm_frac"] else "hbm"
        roof["kernel"] = kname
        roof["avg_launch_us"] = avg_s * 1e6
        roof["launches_per_step"] = launches / args.steps
        roof["layers_per_launch"] = layers_per_launch if dom in (0, 1) else (hidden_layers if dom in (8, 9) else
                                                                              hidden_layers / max(launches / args.steps, 1))
        step_bytes = 2 * (bytes_fwd(n, e, 2) + (num_layers + 1) * bytes_fwd(n, e, hidden))
        roof["step_aggregation_GBps"] = step_bytes / (ms_per_step * 1e-3) / 1e9
        roof["kernel_ms_per_step"] = {(KNAMES_STACK[k] if (k in (0, 1) and per_kernel[k][0] / args.steps < hidden_layers / 2)
                                       else KNAMES[k]): per_kernel[k][1] / args.steps for k in per_kernel}

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

        others = None
        if world == 1 and not args.no_other_configs and args.config == "L256" and args.data == "D0" and B == 256 \
                and args.math == "fp32":
            note("the other single-GPU BASELINE configurations (same protocol)")
            others = [secondary_config(c, d, 256, dev, args.steps, args.warmup, min(args.preheat_ms, 150.0))
                      for c, d in (("L256", "D1"), ("S256", "D0"), ("MIX", "D0"))]

        cpu = None
        if world == 1 and not args.no_cpu_baseline:      # rank 0 at N=1 only (bench contract)
            note("CPU baseline sweep (bounded: ~40 s)")
            cpu = cpu_baseline(ref, batches, B)

        probe = None
        if world == 1 and not args.no_collective_probe:
            probe = collective_probe_child(4 * sum(p.numel() for p in plist if p.grad is not None))

        out = {
            "metric": "board-graphs/sec fwd+bwd", "value": value, "unit": "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "preheat_steps": preheat_steps, "ms_per_step": ms_per_step,
            "higher_is_better": True, "sustained": sustained,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.math == "fp32" else "f32 operands split into scaled f16 hi+lo (f16x3 MFMA, 22-bit products), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "%s, %s board graphs, %d graphs per GPU (N=%d nodes, E=%d directed edges)"
                                   % (label, "start-position" if args.data == "D0" else "random-playout",
                                      batches[0]["graphs"], batches[0]["n"], batches[0]["e"]),
                       "parallelism": "dp%d" % world, "hip_graph": bool(args.graph), "global_batch": B * gfactor,
                       "graphs_per_gpu": batches[0]["graphs"],
                       "collation": "caller's order" if batches[0]["blocks"] is None else
                                    "%s (%d graph-aligned row blocks)" % ("gnn_hex_amd.data.pack_order" if batches[0]["packed"] else
                                                                          "caller's order, data.blocks_for_order", batches[0]["blocks"]),
                       "step_issue": "model(...), F.mse_loss(q[sel], target), loss.backward()" if args.plain_autograd else
                                     "ops.td_step (model forward, TD loss in its tail, backward)" if td_fused else
                                     "model(...), ops.td_loss, ops.backward"},
            "roofline": roof, "cpu_baseline": cpu, "split_precision_mode": split, "other_configs": others,
        }
        if probe is not None:
            out["config"]["collective"] = probe
        if world > 1:
            out["replicas_identical"] = replicas_identical
            out["config"]["collective"] = {"backend": "rccl" if args.backend == "nccl" else "gloo (rehearsal)",
                                           "overlapped_with_backward": overlap,
                                           "bucket_bytes": 4 * sum(p.numel() for p in plist if p.grad is not None)}
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


class _stdout_to_stderr:
    """RCCL prints its version banner on STDOUT when a communicator is created: route file descriptor 1 to stderr for the
    duration, so that rank 0's stdout carries the ONE JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def collective_probe_child(nbytes, timeout_s=90.0):
    """The probe below in a FRESH child process (started, never exec'ed, by this one) with a time limit: an RCCL or rendezvous
    hang cannot take the bench line down, and this process's MASTER_* environment stays untouched.  Never raises."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-child", str(int(nbytes))], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ))
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"backend": "rccl", "world": 1, "error": "no result (rc %d): %s" % (r.returncode, r.stderr[-200:])}
    except Exception as exc:  # noqa: BLE001
        return {"backend": "rccl", "world": 1, "error": "%s: %s" % (type(exc).__name__, exc)}


def collective_probe(dev, nbytes, reps=50):
    """N = 1 only: what ONE all-reduce of the step's gradient bucket costs on this GPU through the exact call the N-rank
    path makes (a 1-rank RCCL group, in-place SUM on a flat fp32 buffer of the bucket's size): launch + kernel latency of the
    collective without any wire time, i.e. the floor the per-step exchange adds when it is not overlapped.  Never raises."""
    try:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        with _stdout_to_stderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        try:
            flat = torch.zeros(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
            with _stdout_to_stderr():
                for _ in range(5):
                    dist.all_reduce(flat)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                dist.all_reduce(flat)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
        finally:
            dist.destroy_process_group()
        return {"backend": "rccl", "world": 1, "bucket_bytes": int(nbytes), "all_reduce_us_one_rank": us,
                "note": "1-rank RCCL group on this GPU: launch + kernel latency of the per-step gradient all-reduce, no wire time"}
    except Exception as exc:  # noqa: BLE001
        return {"backend": "rccl", "world": 1, "error": "%s: %s" % (type(exc).__name__, exc)}


def step_local(hip, batches, i):
    from gnn_hex_amd import ops as hexops
    bt = batches[i & 1]
    for p in hip.parameters():
        p.grad = None
    # (the instrumented passes issue the step the way the timed region does: ops.td_step)
    hexops.td_step(hip, bt["x"], bt["ei"], bt["bv"], bt["ptr"], sel=bt["sel"], target=bt["tgt"], weights=bt.get("w"))


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ref, batches, B, total_budget_s=40.0):
    """CPU restatement of the torch_geometric path (oracle/model_ref.py) timed on the host cores of this box, on the
    FULL batch of the same workload (same graphs, weights, targets), fwd + bwd, for torch thread counts 1, 8, 16, 32 and
    all usable cores (SURVEY 8d / BASELINE.md section 3): per thread count 1 warm-up iteration, then the median of up to
    10 timed iterations inside its share of a ~40 s budget (at least 2; a thread count is skipped once the budget is
    spent; more threads than 64 are not tried: the GPU boxes expose 256 logical cpus of which a 1-GPU lease owns a
    16-core share, and 128 oversubscribed threads ran slower than one).  The best thread count is reported; the whole
    sweep is kept beside it."""
    import statistics
    x, ei, bv, ptr, sel, tgt = batches[0]["cpu"]
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = ncpu
    saved = torch.get_num_threads()

    def one():
        ref.zero_grad(set_to_none=True)
        q = ref(x, ei, bv, ptr)
        torch.nn.functional.mse_loss(q[sel], tgt).backward()

    counts = sorted({c for c in (8, 16, 32, 1, min(usable, 64)) if c <= usable}, key=lambda c: (c == 1, c))   # 1 thread last
    sweep, t_all = {}, time.perf_counter()
    for k, th in enumerate(counts):
        left = total_budget_s - (time.perf_counter() - t_all)
        if left <= 0:
            break
        share = left / (len(counts) - k)
        torch.set_num_threads(th)
        t_start = time.perf_counter()
        one()
        times = []
        if time.perf_counter() - t_start > share:        # one iteration already exceeds this count's share: keep it, move on
            times.append(time.perf_counter() - t_start)
        while len(times) < 10 and (len(times) < 2 or time.perf_counter() - t_start < share) \
                and not (times and time.perf_counter() - t_start > share):
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
            if len(times) >= 2 and time.perf_counter() - t_all > 2.5 * total_budget_s:
                break
        sweep[th] = {"graphs_per_s": B / statistics.median(times), "median_s": statistics.median(times),
                     "iterations": len(times)}
        print("[bench] cpu baseline: %d threads %.1f graphs/s (%d iterations)" % (th, sweep[th]["graphs_per_s"], len(times)),
              file=sys.stderr, flush=True)
    torch.set_num_threads(saved)
    best = max(sweep, key=lambda k: sweep[k]["graphs_per_s"])
    return {"value": sweep[best]["graphs_per_s"], "unit": "graphs/s", "cores": best, "kind": "port",
            "sample": "median of %d fwd+bwd iterations on the full %d-graph batch of the same workload (%d nodes, %d edges), "
                      "CPU restatement of the torch_geometric path (oracle/model_ref.py), best of torch threads %s; "
                      "host: %s, %d cpus (%d usable)"
                      % (sweep[best]["iterations"], B, int(x.shape[0]), int(ei.shape[1]), sorted(sweep), _cpu_model(),
                         ncpu, usable),
            "threads_sweep": {str(k): v for k, v in sweep.items()}}


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--probe-child":
        # child of collective_probe_child(): a 1-rank RCCL all-reduce of the bucket, result as one JSON line
        import gnn_hex_amd  # noqa: F401
        print(json.dumps(collective_probe(torch.device("cuda", 0), int(sys.argv[2]))), flush=True)
        sys.exit(0)
    main()
