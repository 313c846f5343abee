"""gnn_hex_amd -- MI355X-native hot path of GNN_Hex's RainbowDQN loop.

``models.get_pre_defined("modern_two_headed", args)`` mirrors ``GN0.models.get_pre_defined``;
``multi_env_manager.Env_manager`` mirrors ``graph_game.multi_env_manager.Env_manager``.
Everything numeric runs in libhexgnn.so (hand-written HIP kernels for gfx950, C ABI in
include/hexgnn.h); importing this package does not load the library, calling into it does and
fails loudly when it is missing.
"""
import os as _os

# ROCm 7.2: hipGraphExec's default AQL packet capture corrupts kernel arguments when eager launches run between graph
# replays (gnn_hex_amd/graphs.py).  The switch must be in the environment BEFORE the HIP runtime initialises, so it is
# set on import of this package (a no-op for processes that never capture a graph; GraphedStep refuses to run if HIP was
# already up without it).
import torch as _torch

_graph_env_ok = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" or not _torch.cuda.is_initialized()
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .data import Batch, Data  # noqa: F401
from .models import get_pre_defined  # noqa: F401

__all__ = ["Batch", "Data", "get_pre_defined"]
