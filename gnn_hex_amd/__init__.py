"""gnn_hex_amd -- MI355X-native hot path of GNN_Hex's RainbowDQN loop.

``models.get_pre_defined("modern_two_headed", args)`` mirrors ``GN0.models.get_pre_defined``;
``multi_env_manager.Env_manager`` mirrors ``graph_game.multi_env_manager.Env_manager``.
Everything numeric runs in libhexgnn.so (hand-written HIP kernels for gfx950, C ABI in
include/hexgnn.h); importing this package does not load the library, calling into it does and
fails loudly when it is missing.
"""
from .data import Batch, Data  # noqa: F401
from .models import get_pre_defined  # noqa: F401

__all__ = ["Batch", "Data", "get_pre_defined"]
