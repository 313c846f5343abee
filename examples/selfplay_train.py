#!/usr/bin/env python
"""A complete RainbowDQN-style self-play loop on the HIP path (what GN0/RainbowDQN's train.py does with the reference
pieces, README.md:5,7 flags): 128 parallel Hex-11 envs, eps-greedy acting with the online network, n-step (2) transitions
into two prioritized replay rings (maker / breaker), double-DQN targets from a target network, importance-weighted MSE,
Adam, priority updates.  Everything between two prints stays on the GPU except one read-back per 16-move rollout.

    python examples/selfplay_train.py --iters 50

The reference's training script itself lives in an un-vendored submodule and is not rebuilt; this file shows how its loop
maps onto the drop-in API and measures end-to-end frames/s and updates/s."""
import argparse
import copy
import os
import sys
import time
from argparse import Namespace

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")      # gnn_hex_amd/graphs.py
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_hex_amd import ops  # noqa: E402
from gnn_hex_amd.models import get_pre_defined  # noqa: E402
from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager, RolloutStitcher  # noqa: E402
from gnn_hex_amd.replay import GraphReplayBuffer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hex-size", type=int, default=11)
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--layers", type=int, default=15)
    ap.add_argument("--hidden", type=int, default=110)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warm", type=int, default=5, help="untimed iterations first (graph capture, first launches, allocator)")
    ap.add_argument("--rollout", type=int, default=16, help="moves per env between updates")
    ap.add_argument("--updates", type=int, default=2, help="gradient steps per side and iteration")
    ap.add_argument("--math", default="fp32", choices=["fp32", "f16x3"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--async-rollout", action="store_true",
                    help="issue the next rollout before this iteration's updates: the GPU plays while the host prepares "
                         "the update batches; the actor's weights are then one iteration (4 updates) old")
    args = ap.parse_args()
    ops.set_math(args.math)
    gamma, n_step = 0.97, 2
    margs = Namespace(num_layers=args.layers, hidden_channels=args.hidden, norm=False, noisy_dqn=False, noisy_sigma0=0.5,
                      num_head_layers=2)
    torch.manual_seed(0)
    q_net = get_pre_defined("modern_two_headed", margs).cuda()
    target_net = copy.deepcopy(q_net)
    opt = torch.optim.Adam(q_net.parameters(), lr=4e-4, fused=True)     # one launch for all 66 parameter tensors
    mgr = Env_manager(args.envs, args.hex_size, gamma=gamma, n_steps=[n_step])
    mgr.reset()
    cap = 65536
    bufs = {True: GraphReplayBuffer(cap, args.hex_size, prioritized=True, alpha=0.5),
            False: GraphReplayBuffer(cap, args.hex_size, prioritized=True, alpha=0.5)}
    rollout = DeviceRollout(mgr, q_net, steps=args.rollout, eps=0.12, graph=not args.no_graph)
    stitch = RolloutStitcher(mgr)      # carries the last 2 * n_step moves of a rollout into the next assembly
    frames = updates = games = 0
    last_loss = float("nan")
    t0 = time.perf_counter()
    ahead = None
    for it in range(-args.warm, args.iters):
        if it == 0:       # start of the timed part
            torch.cuda.synchronize()
            frames = updates = games = 0
            t0 = time.perf_counter()
        res = rollout.run_end(ahead) if ahead is not None else rollout.run()
        frames += args.envs * args.rollout
        games += int(res.dones.sum())
        mb, bb = stitch.assemble(res)
        bufs[True].put_block(mb)
        bufs[False].put_block(bb)
        # Updates alternate between the two buffers, and a buffer's next draw (stream-ordered behind its own priority
        # update) is started right after its update is issued: while the host waits for those indices and builds that
        # batch, the GPU runs the OTHER buffer's update.
        sides = [sd for sd in (True, False) if len(bufs[sd]) >= args.batch]
        pending = {sd: bufs[sd].sample_begin(args.batch, beta=0.6) for sd in sides}
        # (after put_block: the run overwrites the snapshot ring; after the first draws: they need not wait for the rollout)
        ahead = rollout.run_begin() if args.async_rollout else None
        for k in range(args.updates):
            for side in sides:
                buf = bufs[side]
                idx, w, s, s2, act, r, d = buf.sample_end(pending[side])
                with torch.no_grad():
                    q_on = q_net(s2.x, s2.edge_index, s2.batch, s2.ptr)
                    q_tg = target_net(s2.x, s2.edge_index, s2.batch, s2.ptr)
                    # double DQN: argmax of the online net over each next state's non-terminal nodes, target net's value
                    y = r + (gamma ** n_step) * q_tg[ops.greedy_nodes(q_on, s2.ptr)] * (~d).float()
                q = q_net(s.x, s.edge_index, s.batch, s.ptr)
                loss, td = ops.td_loss(q, s.ptr[:-1] + act.long(), y, w, "mse")
                opt.zero_grad(set_to_none=True)
                ops.backward(loss)
                opt.step()
                buf.update_priorities(idx, td.abs() + 1e-3)
                if k + 1 < args.updates:
                    pending[side] = buf.sample_begin(args.batch, beta=0.6)
                updates += 1
                last_loss = loss.detach()
        if (it + 1) % 10 == 0 and it >= 0:
            target_net.load_state_dict(q_net.state_dict())
    if ahead is not None:
        rollout.run_end(ahead)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("hex %d, %d envs, GNN %dx%d, math %s%s: %.0f env frames/s, %.1f updates/s (batch %d), %d games finished, last loss %.4f"
          % (args.hex_size, args.envs, args.layers, args.hidden, args.math, ", rollout one iteration ahead" if args.async_rollout else "", frames / dt, updates / dt, args.batch, games,
             float(last_loss)))


if __name__ == "__main__":
    main()
