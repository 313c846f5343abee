"""Shared test helpers: synthetic batches and model pairs (HIP path + CPU oracle)."""
from argparse import Namespace

import numpy as np
import torch


def model_args(num_layers, hidden, head_layers=2):
    return Namespace(num_layers=num_layers, hidden_channels=hidden, norm=False, noisy_dqn=False,
                     noisy_sigma0=0.5, num_head_layers=head_layers)


def make_pair(num_layers, hidden, seed=0, device="cuda"):
    """(HIP model on device, CPU oracle) with identical weights (torch.manual_seed(seed))."""
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    torch.manual_seed(seed)
    ref = get_pre_defined_ref("modern_two_headed", model_args(num_layers, hidden))
    hip = get_pre_defined("modern_two_headed", model_args(num_layers, hidden))
    hip.load_state_dict(ref.state_dict())
    return hip.to(device), ref


def sharpen_(model, alpha=0.8, beta=0.5, gain=2.0, vscale=0.02):
    """In-place change of a freshly initialised model into a state whose signal does NOT collapse (with the
    default init a 15-layer mean-aggregating ReLU stack over-smooths -- std(A - mean A) ~ 1e-5, advantage-linear gradients
    ~ 1e-5 -- so absolute 1e-4 gates cannot see an error there).  Every hidden SAGE layer becomes a high-pass filter,
    lin_r <- gain * lin_r, lin_l <- beta * gain * lin_l - alpha * lin_r (node minus most of its neighbourhood mean keeps
    the differences between nodes alive through the stack), and the value MLP's first layer is scaled down so that
    tanh(v) does not saturate on the sum-pooled inputs.  Measured with the oracle on GNN-L Hex-11 and GNN-S Hex-7, start
    and mid-game boards, both heads: std(A - mean A) 0.3-0.6, Q spread > 1.5, every gradient tensor's |g|max >= 1e-2.
    Works on the oracle and on the HIP mirror alike (same parameter names)."""
    with torch.no_grad():
        sd = dict(model.named_parameters())
        for k, p in sd.items():
            if k.endswith("lin_r.weight") and p.shape[1] > 2:
                p.mul_(gain)
                l = sd[k.replace("lin_r", "lin_l")]
                l.mul_(beta * gain).add_(p, alpha=-alpha)
            if "value_head.layers.0.weight" in k:
                p.mul_(vscale)
    return model


def batch_tensors(kind, sizes, maker=True):
    from oracle import env_ref
    x, ei, batch, ptr = env_ref.make_batch(kind, sizes, maker_turn=maker)
    return (torch.from_numpy(x), torch.from_numpy(ei), torch.from_numpy(batch), torch.from_numpy(ptr))


def sel_and_targets(ptr, seed=1):
    """one non-terminal node per graph, index 2 + (g*7919 mod (n_g-2)); targets ~ U(-1,1)"""
    ptr = ptr.tolist()
    sel = []
    for g in range(len(ptr) - 1):
        n_g = ptr[g + 1] - ptr[g]
        sel.append(ptr[g] + 2 + (g * 7919) % max(n_g - 2, 1))
    gen = torch.Generator().manual_seed(seed)
    tgt = torch.rand(len(sel), generator=gen) * 2 - 1
    return torch.tensor(sel, dtype=torch.long), tgt
