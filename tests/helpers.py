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
