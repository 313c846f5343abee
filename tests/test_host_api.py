"""CPU: the drop-in surface (module tree, state-dict compatibility, grow_*), the Data/Batch stand-ins, and the
C-ABI library: it loads and exports every symbol include/hexgnn.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

from helpers import batch_tensors, model_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gnn_hex_amd import _lib
    header = open(os.path.join(ROOT, "include", "hexgnn.h")).read()
    declared = sorted(set(re.findall(r"\b(hexgnn_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libhexgnn.so does not export %s" % name
    assert set(declared) == set(_lib.exported_symbols()), "ctypes signature table out of sync with the header"
    assert L.hexgnn_abi_version() == _lib.ABI_VERSION == 6
    assert L.hexgnn_padded_width(110) == 112 and L.hexgnn_padded_width(35) == 48 and L.hexgnn_padded_width(129) == 144
    assert L.hexgnn_padded_width(256) == 256 and L.hexgnn_padded_width(257) < 0
    assert L.hexgnn_strerror(-3).decode() == "workspace too small"
    assert L.hexgnn_qnet_supported(2, 110, 123) == 1 and L.hexgnn_qnet_supported(2, 110, 146) == 0
    assert L.hexgnn_qnet_supported(2, 128, 51) == 0
    # workspace queries are pure host arithmetic
    assert L.hexgnn_sage_stack_pack_bytes(2, 110, 15) > 14 * 2 * 100352
    assert L.hexgnn_sage_stack_pack_bytes(2, 129, 3) > 2 * 4 * 4 * 144 * 144 and L.hexgnn_sage_stack_pack_bytes(2, 257, 3) == 0


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from gnn_hex_amd import _lib
    L = _lib.lib()
    assert L.hexgnn_csr_build(-1, 0, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert L.hexgnn_graph_ptr(4, 2, None, None, None) == -1
    assert L.hexgnn_pad_rows(4, 300, None, 300, None, None) == -2
    assert L.hexgnn_profile_enable(99) == -1 and L.hexgnn_profile_enable(-1) == 0


def test_model_tree_matches_reference_layout():
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    for (l, h) in ((10, 35), (15, 110)):
        a = get_pre_defined("modern_two_headed", model_args(l, h))
        b = get_pre_defined_ref("modern_two_headed", model_args(l, h))
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)
        a.load_state_dict(sb)                       # a reference-format checkpoint loads unchanged
    for attr in ("gnn", "maker_head", "breaker_head", "after_embed_norm", "supports_cache", "value_activation",
                 "advantage_activation", "final_conv_acts"):
        assert hasattr(a, attr)
    assert a.after_embed_norm is None and a.supports_cache
    for name in ("forward", "simple_forward", "grow_depth", "grow_width", "export_norm_cache", "import_norm_cache"):
        assert callable(getattr(a, name))
    assert a.export_norm_cache() == [None, None, None]


def test_unsupported_configurations_fail_loudly():
    from gnn_hex_amd.models import get_pre_defined
    with pytest.raises(NotImplementedError):
        get_pre_defined("pna_two_headed", model_args(3, 8))
    with pytest.raises(NotImplementedError):
        get_pre_defined("sage+norm", model_args(3, 8))


def test_two_headed_module_tree_and_state_dict_keys():
    """get_pre_defined("two_headed") (GN0/models.py:901-918): CachedGraphNorm (weight, bias, mean_scale) per body / head
    layer + after_embed_norm with cached_norm=True, Linear(H, 1) value head over mean pooling; keys as the oracle's."""
    from gnn_hex_amd.models import CachedGraphNorm, get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    args = model_args(4, 16)
    args.norm = True
    m, r = get_pre_defined("two_headed", args), get_pre_defined_ref("two_headed", args)
    assert list(m.state_dict().keys()) == list(r.state_dict().keys())
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for p in r.parameters())
    assert len(m.gnn.norms) == 4 and isinstance(m.after_embed_norm, CachedGraphNorm) and m.gnn.cached_norm
    assert m.maker_head.value_head.weight.shape == (1, 16) and m.maker_head.value_head_type == "linear"
    assert torch.equal(m.gnn.norms[0].mean_scale, torch.ones(16)) and m.gnn.norms[0].mean_cache is None
    with pytest.raises(AssertionError):
        m.export_norm_cache()              # no cache set yet (GN0/models.py:168)
    args.norm = False
    assert get_pre_defined("two_headed", args).gnn.norms is None


def test_norm_module_tree_and_state_dict_keys():
    """--norm=True: one LayerNorm per body layer (BasicGNN's num_layers - 1 plus CachifiedGNN's final one,
    GN0/models.py:158-162), after_embed_norm (482-483), two per head; keys and order as the oracle's restatement."""
    from gnn_hex_amd.models import LayerNorm, get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    args = model_args(4, 16)
    args.norm = True
    m, r = get_pre_defined("modern_two_headed", args), get_pre_defined_ref("modern_two_headed", args)
    assert list(m.state_dict().keys()) == list(r.state_dict().keys())
    assert len(m.gnn.norms) == 4 and len(m.maker_head.gnn.norms) == 2 and isinstance(m.after_embed_norm, LayerNorm)
    assert m.gnn.norms[0] is not m.gnn.norms[1] and m.gnn.norms[0].weight is not m.after_embed_norm.weight
    assert torch.equal(m.gnn.norms[3].weight, torch.ones(16)) and m.gnn.norms[3].eps == 1e-5
    with pytest.raises(AssertionError):
        m.export_norm_cache()              # as the reference: no cache was ever set
    m.import_norm_cache(None, None, None)  # cached_norm=False: a no-op


def test_noisy_dqn_module_tree_and_state_dict_keys():
    """--noisy_dqn=True: the heads' advantage linear is a FactorizedNoisyLinear with the reference's parameter / buffer
    names (GN0/models.py:84-141), initialised as there (sigma = sigma_0 / sqrt(in), factorised epsilon)."""
    from gnn_hex_amd.models import FactorizedNoisyLinear, get_pre_defined
    args = model_args(3, 16)
    args.noisy_dqn = True
    torch.manual_seed(0)
    m = get_pre_defined("modern_two_headed", args)
    for head in (m.maker_head, m.breaker_head):
        lin = head.linear
        assert isinstance(lin, FactorizedNoisyLinear)
        assert [k for k, _ in lin.named_parameters()] == ["weight_mu", "weight_sigma", "bias_mu", "bias_sigma"]
        assert [k for k, _ in lin.named_buffers()] == ["weight_epsilon", "bias_epsilon"]
        assert torch.allclose(lin.weight_sigma, torch.full((1, 16), 0.5 / 4.0)) and lin.weight_mu.abs().max() <= 0.25
        assert torch.allclose(lin.weight_epsilon, lin.bias_epsilon.outer(lin.weight_epsilon[0] / lin.bias_epsilon[0]))
        w, b = lin.effective()
        assert torch.equal(w, lin.weight_mu + lin.weight_sigma * lin.weight_epsilon) and b.shape == (1,)
        lin.disable_noise()
        assert torch.equal(lin.effective()[0], lin.weight_mu)
    assert sum(p.numel() for p in m.parameters()) == sum(
        p.numel() for p in get_pre_defined("modern_two_headed", model_args(3, 16)).parameters()) + 2 * (16 + 1)


def test_cpu_tensors_raise_not_fall_back():
    from gnn_hex_amd._lib import HexGnnError
    from gnn_hex_amd.models import get_pre_defined
    m = get_pre_defined("modern_two_headed", model_args(3, 8))
    x, ei, batch, ptr = batch_tensors("D0", [5])
    with pytest.raises(HexGnnError):
        m(x, ei, batch, ptr)


def test_grow_depth_and_width_parameter_surgery():
    """GN0/models.py:166-238,336-360,494-508: identity layers / zero-padded widening (host logic only)."""
    from gnn_hex_amd.models import get_pre_defined
    torch.manual_seed(0)
    m = get_pre_defined("modern_two_headed", model_args(3, 8))
    old = {k: v.clone() for k, v in m.state_dict().items()}
    m.grow_depth(2)
    assert m.gnn.num_layers == 5 and len(m.gnn.convs) == 5
    c = m.gnn.convs[4]
    assert torch.equal(c.lin_r.weight.data, torch.eye(8)) and c.lin_l.weight.abs().sum() == 0 and c.lin_l.bias.abs().sum() == 0
    m.grow_width(12)
    sd = m.state_dict()
    assert sd["gnn.convs.0.lin_l.weight"].shape == (12, 2) and sd["gnn.convs.1.lin_l.weight"].shape == (12, 12)
    assert torch.equal(sd["gnn.convs.1.lin_l.weight"][:8, :8], old["gnn.convs.1.lin_l.weight"])
    assert sd["gnn.convs.1.lin_l.weight"][:8, 8:].abs().sum() == 0
    assert torch.equal(sd["gnn.convs.0.lin_r.weight"][:8], old["gnn.convs.0.lin_r.weight"])
    assert torch.equal(sd["gnn.convs.2.lin_l.bias"][:8], old["gnn.convs.2.lin_l.bias"])
    assert sd["maker_head.linear.weight"].shape == (1, 12)
    assert torch.equal(sd["maker_head.linear.weight"][:, :8], old["maker_head.linear.weight"])
    assert sd["maker_head.linear.weight"][:, 8:].abs().sum() == 0
    assert sd["maker_head.value_head.layers.0.weight"].shape == (6, 48)
    assert torch.equal(sd["maker_head.value_head.layers.0.weight"][:4, :32], old["maker_head.value_head.layers.0.weight"])
    assert sd["maker_head.gnn.convs.0.lin_l.weight"].shape == (12, 12)
    assert m.gnn.hidden_channels == 12 and m.maker_head.hidden_channels == 12


def test_data_and_batch_stand_ins():
    from gnn_hex_amd.data import Batch, Data
    d1 = Data(x=torch.zeros(3, 3), edge_index=torch.tensor([[0, 1], [1, 0]]), backmap=torch.tensor([0, 1, 5]))
    d2 = Data(x=torch.ones(2, 3), edge_index=torch.tensor([[0], [1]]), backmap=torch.tensor([0, 1]))
    b = Batch.from_data_list([d1, d2])
    assert b.num_graphs == 2
    assert torch.equal(b.ptr, torch.tensor([0, 3, 5]))
    assert torch.equal(b.batch, torch.tensor([0, 0, 0, 1, 1]))
    assert torch.equal(b.edge_index, torch.tensor([[0, 1, 3], [1, 0, 4]]))
    assert torch.equal(b.backmap, torch.tensor([0, 1, 5, 0, 1]))      # concatenated un-offset
    d1.__delattr__("backmap")                                          # Env_manager.get_transitions does this
    assert not hasattr(d1, "backmap")
    assert d1.to("cpu") is d1 and d1.num_nodes == 3 and d1.num_edges == 2


def test_reference_checkpoint_format_round_trip(tmp_path):
    """RainbowDQN checkpoints are ``{"state_dict", "args" (argparse Namespace), optional "cache"}``
    (GN0/RainbowDQN/evaluate_elo.py:98-102,177-183): a file written from the oracle/reference layout loads into the
    mirror through the same three lines the reference uses, and back."""
    from argparse import Namespace
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    torch.manual_seed(3)
    args = model_args(10, 35)
    ref = get_pre_defined_ref("modern_two_headed", args)
    path = str(tmp_path / "checkpoint_14395392.pt")
    torch.save({"state_dict": ref.state_dict(), "args": args, "cache": ref.state_dict() and None}, path)
    stuff = torch.load(path, weights_only=False)
    assert isinstance(stuff["args"], Namespace)
    model = get_pre_defined("modern_two_headed", args=stuff["args"])
    missing = model.load_state_dict(stuff["state_dict"])
    assert not missing.missing_keys and not missing.unexpected_keys
    if "cache" in stuff and stuff["cache"] is not None:
        model.import_norm_cache(*stuff["cache"])
    for k, v in ref.state_dict().items():
        assert torch.equal(model.state_dict()[k], v)
    # and the other way round: a checkpoint written by the mirror loads into the reference layout
    path2 = str(tmp_path / "back.pt")
    torch.save({"state_dict": model.state_dict(), "args": args}, path2)
    ref2 = get_pre_defined_ref("modern_two_headed", args)
    ref2.load_state_dict(torch.load(path2, weights_only=False)["state_dict"])
    model.import_norm_cache(None, None, None)          # norm=False: a no-op, as in the reference


def test_transition_assembly_matches_the_oracle_loop_on_host():
    """Env_manager.assemble_transitions (numpy over envs) and the list form built on it (get_transitions) against the
    ORACLE's restatement of the reference loop (oracle/env_ref.py RefEnvManager.get_transitions <-
    graph_game/multi_env_manager.py:113-165), on fabricated histories -- no GPU involved: the manager is built without
    its device handles and its start observation is stubbed."""
    import numpy as np
    from gnn_hex_amd.data import Data
    from gnn_hex_amd.multi_env_manager import Env_manager, ObsList
    from oracle import env_ref

    class FakeObs(ObsList):
        def __init__(self, k, maker):
            super().__init__(None, None, None, None, None, list(range(0, 3 * k + 1, 3)), list(range(0, 2 * k + 1, 2)), None,
                             maker, 3, None)
            self._d = [Data(x=torch.tensor([[1.0, 1.0, float(maker)]] * 3), edge_index=torch.zeros((2, 2), dtype=torch.long))
                       for _ in range(k)]

        def __getitem__(self, i):
            return self._d[i]

    def fresh_start():
        return Data(x=torch.zeros((3, 3)), edge_index=torch.zeros((2, 2), dtype=torch.long), backmap=torch.arange(3))

    class HostOnlyManager(Env_manager):      # no device handles: only the transition maths is exercised
        def __init__(self, n_steps, prune):
            self.gamma, self.n_steps, self.prune_exploratories = 0.9, n_steps, prune
            self._base, self._base_sizes, self._h = None, None, None

        def _observe_handle(self, *a, **k):
            return FakeObs(1, True)

        @property
        def starting_obs(self):
            return fresh_start()

    rng = np.random.default_rng(3)
    for prune in (True, False):
        for n_steps in ([1], [2], [1, 3]):
            mgr = HostOnlyManager(n_steps, prune)
            class OracleSelf:                      # what RefEnvManager.get_transitions reads from `self`
                starting_obs = property(lambda self: fresh_start())
            oracle_self = OracleSelf()
            oracle_self.n_steps, oracle_self.gamma, oracle_self.prune_exploratories = n_steps, 0.9, prune
            T, E = 14, 6
            start = FakeObs(E, True)
            states = [FakeObs(E, (t % 2) == 1) for t in range(T)]       # side alternates: state t+1 after move t
            actions = [rng.integers(0, 3, E) for _ in range(T)]
            dones = [rng.random(E) < 0.15 for _ in range(T)]
            rewards = [np.where(d, rng.choice([-1.0, 1.0], E), 0.0) for d in dones]
            expl = [rng.random(E) < 0.25 for _ in range(T)]
            mb, bb = mgr.assemble_transitions(start, states, actions, rewards, dones, expl)
            ml, bl = mgr.get_transitions(start, states, actions, rewards, dones, expl)
            om, ob = env_ref.RefEnvManager.get_transitions(oracle_self, start, states, actions, rewards, dones, expl)
            sh = [start] + states
            for block, lst, want in ((mb, ml, om), (bb, bl, ob)):
                assert len(block) == len(lst) == len(want) and len(want) > 0
                assert block.action.tolist() == [int(t[1]) for t in want] == [int(t[1]) for t in lst]
                assert np.allclose(block.reward, [t[2] for t in want]) and np.allclose([t[2] for t in lst], [t[2] for t in want])
                assert block.done.tolist() == [bool(t[4]) for t in want] == [bool(t[4]) for t in lst]
                assert (block.next_step[block.done] == -1).all() and (block.next_step[~block.done] >= 0).all()
                for k, (got, exp) in enumerate(zip(lst, want)):
                    assert got[0] is exp[0]                                     # the very same state object
                    assert got[0] is sh[int(block.src_step[k])][int(block.env[k])]
                    if exp[4]:                                                  # terminal: a fresh start observation
                        assert not hasattr(got[3], "backmap") and float(got[3].x[0, 2]) == float(exp[3].x[0, 2])
                    else:
                        assert got[3] is exp[3]


def test_rollout_stitcher_on_host():
    """RolloutStitcher: three consecutive rollouts assembled with the carried-over tail must give exactly the transitions
    of ONE assembly over the concatenated histories (each window once, also with two different n_step values), on
    fabricated histories without a GPU."""
    import numpy as np
    from gnn_hex_amd.data import Data
    from gnn_hex_amd.multi_env_manager import Env_manager, ObsList, RolloutResult, RolloutStitcher

    class FakeObs(ObsList):
        def __init__(self, k, maker):
            super().__init__(None, None, None, None, None, list(range(0, 3 * k + 1, 3)), list(range(0, 2 * k + 1, 2)), None,
                             maker, 3, None)
            self._d = [Data(x=torch.tensor([[1.0, 1.0, float(maker)]] * 3), edge_index=torch.zeros((2, 2), dtype=torch.long))
                       for _ in range(k)]

        def __getitem__(self, i):
            return self._d[i]

    class HostOnlyManager(Env_manager):
        def __init__(self, n_steps, prune):
            self.gamma, self.n_steps, self.prune_exploratories = 0.9, n_steps, prune
            self._base, self._base_sizes, self._h = None, None, None

        def _observe_handle(self, *a, **k):
            return FakeObs(1, True)

    for n_steps in ([2], [2, 1], [1], [3, 1]):
        rng = np.random.default_rng(sum(n_steps))
        T, E = 8, 5
        mgr = HostOnlyManager(n_steps, True)
        stitch = RolloutStitcher(mgr)
        assert stitch.keep == 2 * max(n_steps) - 1
        allstates, runs, got = [FakeObs(E, True)], [], {True: [], False: []}
        for r in range(3):
            states = [allstates[-1]] + [FakeObs(E, ((r * T + t + 1) % 2) == 0) for t in range(T)]
            allstates += states[1:]
            acts = [rng.integers(0, 3, E) for _ in range(T)]
            dones = [rng.random(E) < 0.15 for _ in range(T)]
            rews = [np.where(d, rng.choice([-1.0, 1.0], E), 0.0) for d in dones]
            expl = [rng.random(E) < 0.25 for _ in range(T)]
            res = RolloutResult(states, np.array(acts), np.array(acts), np.array(rews), np.array(dones), np.array(expl), None)
            runs.append(res)
            off = max(0, r * T - stitch.keep)
            for side, blk in zip((True, False), stitch.assemble(res)):
                for s_, e_, a_, rw, nx, d_ in zip(blk.src_step.tolist(), blk.env.tolist(), blk.action.tolist(),
                                                  blk.reward.tolist(), blk.next_step.tolist(), blk.done.tolist()):
                    got[side].append((s_ + off, e_, a_, round(rw, 9), -1 if nx < 0 else nx + off, d_))
        cat = lambda name: [x for res in runs for x in list(getattr(res, name))]       # noqa: E731
        blocks = mgr.assemble_transitions(allstates[0], allstates[1:], cat("actions"), cat("rewards"), cat("dones"),
                                          cat("exploratories"))
        for side, blk in zip((True, False), blocks):
            want = sorted(zip(blk.src_step.tolist(), blk.env.tolist(), blk.action.tolist(),
                              [round(v, 9) for v in blk.reward.tolist()],
                              [-1 if v < 0 else v for v in blk.next_step.tolist()], blk.done.tolist()))
            assert sorted(got[side]) == want and len(want) > 0, (n_steps, side)
