"""End-to-end slice of the RainbowDQN loop on the HIP path only: device rollout -> n-step transition assembly -> device
prioritized replay -> sampled batches -> double-DQN target -> fused TD loss -> fused backward -> Adam -> priority update.
The reference's train.py (un-vendored Rainbow submodule) is not rebuilt; this is the smallest loop that exercises every
component of SURVEY section 8 together and checks that learning signals flow (finite, changing parameters, sane TD errors)."""
import copy

import numpy as np
import pytest
import torch

from helpers import model_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("math", ["fp32", "f16x3"])
def test_rollout_replay_update_loop(math):
    from gnn_hex_amd import ops
    from gnn_hex_amd.models import get_pre_defined
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    torch.manual_seed(0)
    ops.set_fused(True)
    ops.set_math(math)
    try:
        hex_size, envs, gamma, n_step = 5, 32, 0.97, 2
        q_net = get_pre_defined("modern_two_headed", model_args(4, 32)).cuda()
        target_net = copy.deepcopy(q_net)
        opt = torch.optim.Adam(q_net.parameters(), lr=4e-4)
        mgr = Env_manager(envs, hex_size, gamma=gamma, n_steps=[n_step])
        mgr.reset()
        bufs = {True: GraphReplayBuffer(4096, hex_size, prioritized=True, alpha=0.5),
                False: GraphReplayBuffer(4096, hex_size, prioritized=True, alpha=0.5)}
        rollout = DeviceRollout(mgr, q_net, steps=8, eps=0.3, graph=False)
        before = [p.detach().clone() for p in q_net.parameters()]
        losses, finished = [], 0
        for it in range(6):
            res = rollout.run()
            finished += int(res.dones.sum())
            mb, bb = mgr.assemble_transitions(res.states[0], res.states[1:], list(res.actions), list(res.rewards),
                                              list(res.dones), list(res.exploratories))
            bufs[True].put_block(mb)
            bufs[False].put_block(bb)
            for side in (True, False):
                buf = bufs[side]
                if len(buf) < 64:
                    continue
                idx, w, s, s2, act, r, d = buf.sample(64, beta=0.6)
                with torch.no_grad():
                    # the next state has the SAME side to move (2n plies later): double DQN over the non-terminal nodes
                    q_next_online = q_net(s2.x, s2.edge_index, s2.batch, s2.ptr)
                    q_next_target = target_net(s2.x, s2.edge_index, s2.batch, s2.ptr)
                    best = ops.greedy_nodes(q_next_online, s2.ptr)
                    p2 = s2.ptr.tolist()
                    g0 = 5
                    assert int(best[g0]) == p2[g0] + 2 + int(torch.argmax(q_next_online[p2[g0] + 2:p2[g0 + 1]]))
                    y = r + (gamma ** n_step) * q_next_target[best] * (~d).float()
                q = q_net(s.x, s.edge_index, s.batch, s.ptr)
                sel = s.ptr[:-1] + act.long()
                loss, td = ops.td_loss(q, sel, y, w, "mse")
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                buf.update_priorities(idx, td.abs() + 1e-3)
                losses.append(float(loss.detach()))
                assert torch.isfinite(td).all() and float(td.abs().max()) < 10.0
            target_net.load_state_dict(q_net.state_dict())
        assert finished > 0, "Hex-5 games should finish within 48 plies"
        assert len(losses) >= 6 and all(np.isfinite(losses))
        moved = sum(float((p.detach() - b).abs().max()) > 0 for p, b in zip(q_net.parameters(), before))
        assert moved >= len(before) - 2, "every parameter tensor of the body and of both heads should have been updated"
    finally:
        ops.set_math("fp32")
