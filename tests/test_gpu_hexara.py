"""The HexAra policy/value network SAGE_torch_script on the device (GN0/torch_script_models.py:286-379).
Index surgery against the reference's own known answers (rl_loop/unittest_model.py:16-92, replayed through
tests/test_oracle_hexara.py's fixtures), values and every gradient against the oracle restatement."""
import numpy as np
import pytest
import torch

from helpers import batch_tensors
from test_oracle_hexara import FIXTURES, check_random_case, fixture_inputs, random_case

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pair(hidden, layers, policy_layers, value_layers, swap_allowed, seed=0, norm=False):
    from gnn_hex_amd.models import LayerNorm
    from gnn_hex_amd.torch_script_models import get_current_model
    from oracle.hexara_ref import get_current_model_ref
    from oracle.model_ref import LayerNormRef
    torch.manual_seed(seed)
    ref = get_current_model_ref(hidden_channels=hidden, hidden_layers=layers, policy_layers=policy_layers,
                                value_layers=value_layers, swap_allowed=swap_allowed, norm=LayerNormRef if norm else None)
    if norm:
        with torch.no_grad():                      # non-trivial affine parameters (the default is weight 1, bias 0)
            for k, p in ref.named_parameters():
                if "norm" in k:
                    p.add_(torch.randn(p.shape) * 0.2)
    hip = get_current_model(hidden_channels=hidden, hidden_layers=layers, policy_layers=policy_layers,
                            value_layers=value_layers, swap_allowed=swap_allowed, norm=LayerNorm if norm else None)
    assert list(hip.state_dict().keys()) == list(ref.state_dict().keys())
    hip.load_state_dict(ref.state_dict())
    return hip.cuda(), ref


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_reference_unittest_fixtures_on_device(name):
    x, ei, gi, bp, want = fixture_inputs(name)
    hip, ref = _pair(12, 3, 2, 2, True)
    with torch.no_grad():
        pi, value, ogi, obp = hip(x.cuda(), ei.cuda(), gi.cuda(), bp.cuda())
        pi_r, value_r, _, _ = ref(x, ei, gi, bp)
    assert pi.shape == (want["pi"],) and value.shape == (want["value"],)
    assert ogi.tolist() == want["gi"] and obp.tolist() == want["bp"] and ogi.dtype == torch.int64
    assert (pi.cpu() - pi_r).abs().max().item() < TOL and (value.cpu() - value_r).abs().max().item() < TOL


def test_reference_randomized_property_on_device():
    """The invariants of rl_loop/unittest_model.py:54-92 (random multigraphs, random swap flags): sizes, segment structure,
    pointer arithmetic and sum(exp(pi)) == 1 per graph -- on the device model, plus agreement with the oracle."""
    gen = np.random.default_rng(7)
    hip, ref = _pair(12, 3, 2, 2, True, seed=1)
    for _ in range(15):
        case = random_case(gen)
        with torch.no_grad():
            check_random_case(lambda x, ei, gi, bp: tuple(t.cpu() for t in hip(x.cuda(), ei.cuda(), gi.cuda(), bp.cuda())), case)
            pi, value, ogi, obp = hip(*[t.cuda() for t in case[:4]])
            pi_r, value_r, ogi_r, obp_r = ref(*case[:4])
        assert torch.equal(ogi.cpu(), ogi_r) and torch.equal(obp.cpu(), obp_r)
        assert (pi.cpu() - pi_r).abs().max().item() < 2e-4 and (value.cpu() - value_r).abs().max().item() < TOL


def _loss(pi, value, gen_pi, tv):
    return -(pi * gen_pi).sum() / max(pi.numel(), 1) + torch.nn.functional.mse_loss(value, tv)


@pytest.mark.parametrize("hidden,layers,pl,vl,sizes", [(60, 4, 2, 2, [5, 7, 6, 9]), (35, 3, 1, 1, [7, 5]), (110, 3, 3, 2, [11, 9]),
                                                       (16, 1, 2, 2, [5, 6, 5])])
@pytest.mark.parametrize("swap_allowed", [False, True])
def test_sage_torch_script_matches_oracle(hidden, layers, pl, vl, sizes, swap_allowed):
    _check_against_oracle(hidden, layers, pl, vl, sizes, swap_allowed, norm=False)


@pytest.mark.parametrize("hidden,layers,pl,vl,sizes", [(60, 4, 2, 2, [5, 7, 6, 9]), (35, 3, 1, 3, [7, 5]), (110, 3, 3, 2, [11, 9]),
                                                       (16, 1, 2, 2, [5, 6, 5])])
@pytest.mark.parametrize("swap_allowed", [False, True])
def test_sage_torch_script_with_layernorm_matches_oracle(hidden, layers, pl, vl, sizes, swap_allowed):
    """``norm=LayerNorm`` (rl_loop/train_config.py:10,131; GN0/torch_script_models.py:151-160,179-187,306,316-317): whole-batch
    LayerNorm between every contraction and its ReLU but the last layer's, and on the embeddings in front of the heads."""
    _check_against_oracle(hidden, layers, pl, vl, sizes, swap_allowed, norm=True)


def _check_against_oracle(hidden, layers, pl, vl, sizes, swap_allowed, norm):
    hip, ref = _pair(hidden, layers, pl, vl, swap_allowed, seed=hidden, norm=norm)
    x, ei, batch, ptr = batch_tensors("D1", sizes, maker=True)
    x = x.clone()
    b = len(sizes)
    flags = [(g * 7 + 3) % 3 != 0 for g in range(b)]        # swap allowed in some graphs (feature 2 of every node of the graph)
    for g in range(b):
        x[ptr[g]:ptr[g + 1], 2] = 1.0 if flags[g] else 0.0
    pi_r, value_r, ogi_r, obp_r = ref(x, ei, batch, ptr)
    gen = torch.Generator().manual_seed(5)
    tp = torch.rand(pi_r.numel(), generator=gen)
    tv = torch.rand(b, generator=gen) * 2 - 1
    ref.zero_grad()
    _loss(pi_r, value_r, tp, tv).backward()
    pi, value, ogi, obp = hip(x.cuda(), ei.cuda(), batch.cuda(), ptr.cuda())
    assert torch.equal(ogi.cpu(), ogi_r) and torch.equal(obp.cpu(), obp_r)
    assert pi.shape == pi_r.shape and pi.numel() == x.shape[0] - 2 * b + (sum(flags) if swap_allowed else 0)
    assert (pi.detach().cpu() - pi_r.detach()).abs().max().item() < TOL
    assert (value.detach().cpu() - value_r.detach()).abs().max().item() < TOL
    hip.zero_grad()
    _loss(pi, value, tp.cuda(), tv.cuda()).backward()
    torch.cuda.synchronize()
    gr = dict(ref.named_parameters())
    for k, p in hip.named_parameters():
        want = gr[k].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        err = (p.grad.cpu() - want).abs().max().item()
        assert err < TOL * max(1.0, want.abs().max().item()), "%s grad err %g" % (k, err)
    assert (hip.final_conv_acts.cpu() - ref.final_conv_acts.detach()).abs().max().item() < TOL
    assert (hip.final_conv_grads.cpu() - ref.final_conv_grads).abs().max().item() < TOL


def test_hexara_default_model_at_a_search_batch():
    """get_current_model() defaults (hidden 60, 15 + 2 + 2 layers, torch_script_models.py:495) on 64 Hex-11 positions, the
    mini-batch shape of the MCTS search threads: values vs the oracle, and the deterministic repeat."""
    hip, ref = _pair(60, 15, 2, 2, False, seed=2)
    x, ei, batch, ptr = batch_tensors("D1", [11] * 64, maker=True)
    with torch.no_grad():
        pi_r, value_r, _, obp_r = ref(x, ei, batch, ptr)
        dev = [t.cuda() for t in (x, ei, batch, ptr)]
        pi, value, _, obp = hip(*dev)
        pi2, value2, _, _ = hip(*dev)
    assert torch.equal(obp.cpu(), obp_r)
    assert (pi.cpu() - pi_r).abs().max().item() < 2e-4 and (value.cpu() - value_r).abs().max().item() < TOL
    assert torch.equal(pi, pi2) and torch.equal(value, value2)


def test_unsupported_variants_fail_loudly():
    from gnn_hex_amd.torch_script_models import get_current_model
    with pytest.raises(NotImplementedError):
        get_current_model("PNA")
    with pytest.raises(NotImplementedError):
        get_current_model("SAGE", norm=torch.nn.LayerNorm)        # (only the whole-batch LayerNorm of gnn_hex_amd.models)
    with pytest.raises(ValueError):
        get_current_model("nope")
