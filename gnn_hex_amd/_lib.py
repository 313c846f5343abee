"""ctypes binding of libhexgnn.so (C ABI: include/hexgnn.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.
torch is imported first on purpose: libhexgnn.so links libamdhip64.so.7 by soname, and loading
torch first makes the dynamic loader hand it the HIP runtime torch already uses, so streams and
device pointers are shared with torch's allocator.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhexgnn.so")
_lib = None
ABI_VERSION = 6          # HEXGNN_ABI_VERSION of include/hexgnn.h this binding was written against

vp = C.c_void_p
ci = C.c_int
sz = C.c_size_t

_SIGS = {
    "hexgnn_abi_version": (ci, []),
    "hexgnn_strerror": (C.c_char_p, [ci]),
    "hexgnn_last_hip_error": (ci, []),
    "hexgnn_padded_width": (ci, [ci]),
    "hexgnn_csr_workspace_bytes": (sz, [ci, ci]),
    "hexgnn_csr_build": (ci, [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "hexgnn_csr_build_grouped": (ci, [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_csr_build_grouped_pack": (ci, [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp]),
    "hexgnn_csr_build_grouped_pack_e": (ci, [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp]),
    "hexgnn_csr_build_grouped_pack_b": (ci, [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp,
                                             vp, ci, vp]),
    "hexgnn_graph_ptr": (ci, [ci, ci, vp, vp, vp]),
    "hexgnn_sage_stack_pack_bytes": (sz, [ci, ci, ci]),
    "hexgnn_sage_stack_saved_bytes": (sz, [ci, ci, ci, ci]),
    "hexgnn_sage_stack_forward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    "hexgnn_sage_stack_backward_workspace_bytes": (sz, [ci, ci, ci, ci]),
    "hexgnn_sage_stack_backward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp,
                                        vp, vp, vp, vp, sz, ci, vp]),
    "hexgnn_sage_stack_backward_tap": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp,
                                            vp, vp, vp, vp, sz, ci, ci, vp, vp]),
    "hexgnn_sage_stack_forward_blocks": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp, ci, vp]),
    "hexgnn_sage_stack_backward_blocks": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp,
                                               vp, vp, vp, vp, sz, ci, ci, vp, vp, ci, vp]),
    "hexgnn_sage_norm_stack_forward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp,
                                            vp, vp, sz, ci, vp]),
    "hexgnn_sage_norm_stack_forward_live": (ci, [ci, vp, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, C.c_float, vp, vp,
                                                 vp, vp, vp, vp, sz, ci, vp]),
    "hexgnn_sage_norm_stack_backward_workspace_bytes": (sz, [ci, ci, ci, ci]),
    "hexgnn_sage_norm_stack_backward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp,
                                             vp, vp, vp, vp, vp, sz, vp, sz, vp]),
    "hexgnn_head_saved_bytes": (sz, [ci, ci, ci]),
    "hexgnn_head_forward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_head_backward_workspace_bytes": (sz, [ci, ci, ci]),
    "hexgnn_head_backward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                  vp, sz, vp]),
    "hexgnn_graph_layernorm_workspace_bytes": (sz, [ci]),
    "hexgnn_graph_layernorm_forward": (ci, [ci, ci, vp, vp, vp, C.c_float, ci, vp, vp, vp, sz, vp]),
    "hexgnn_graph_layernorm_forward_live": (ci, [ci, vp, ci, vp, vp, vp, C.c_float, ci, vp, vp, vp, sz, vp]),
    "hexgnn_graph_layernorm_backward": (ci, [ci, ci, vp, vp, vp, vp, vp, C.c_float, ci, vp, vp, vp, vp, sz, vp]),
    "hexgnn_graph_colnorm_workspace_bytes": (sz, [ci]),
    "hexgnn_graph_colnorm_forward": (ci, [ci, ci, vp, vp, vp, vp, C.c_float, ci, ci, vp, vp, vp, sz, vp]),
    "hexgnn_graph_colnorm_backward": (ci, [ci, ci, vp, vp, vp, vp, vp, vp, C.c_float, ci, ci, vp, vp, vp, vp, vp, sz, vp]),
    "hexgnn_head_linear_saved_bytes": (sz, [ci, ci]),
    "hexgnn_head_linear_forward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_head_linear_backward_workspace_bytes": (sz, [ci, ci, ci]),
    "hexgnn_head_linear_backward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "hexgnn_sage_scalar_forward": (ci, [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_sage_scalar_backward_workspace_bytes": (sz, [ci, ci]),
    "hexgnn_sage_scalar_backward": (ci, [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "hexgnn_policy_log_softmax_forward": (ci, [ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]),
    "hexgnn_policy_log_softmax_backward": (ci, [ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]),
    "hexgnn_qnet_supported": (ci, [ci, ci, ci]),
    "hexgnn_qnet_saved_bytes": (sz, [ci, ci, ci, ci, ci]),
    "hexgnn_qnet_forward": (ci, [ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                 vp, vp, vp, ci, ci, ci, vp, vp, vp, vp]),
    "hexgnn_qnet_backward_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "hexgnn_qnet_backward": (ci, [ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp,
                                  vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]),
    "hexgnn_qnet_backward_staged": (ci, [ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp,
                                         vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, ci, ci, ci, vp]),
    "hexgnn_qnet_backward_flat": (ci, [ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, sz, vp, ci, ci, ci, vp]),
    "hexgnn_stack_status": (ci, [ci]),
    "hexgnn_stack_reserve_cus": (ci, [ci]),
    "hexgnn_stack_block_budget": (ci, []),
    "hexgnn_debug_stack_mode": (ci, [ci, C.c_uint]),
    "hexgnn_debug_occupy": (ci, [ci, ci, vp, sz, vp, vp]),
    "hexgnn_qnet_forward_td": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci,
                                    vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
    "hexgnn_qnet_backward_flat_td": (ci, [ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                          vp, vp, sz, vp, ci, ci, ci, vp, vp, vp]),
    "hexgnn_env_create": (ci, [ci, ci, vp]),
    "hexgnn_env_destroy": (None, [vp]),
    "hexgnn_env_num_vertices": (ci, [vp]),
    "hexgnn_env_words": (ci, [vp]),
    "hexgnn_env_reset": (ci, [vp, vp, ci, vp, vp]),
    "hexgnn_env_set_maker_turn": (ci, [vp, ci, vp]),
    "hexgnn_env_step": (ci, [vp, vp, ci, ci, ci, vp, vp]),
    "hexgnn_env_observe": (ci, [vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_env_export": (ci, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_env_offsets": (ci, [ci, vp, vp, vp, vp]),
    "hexgnn_env_import": (ci, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_states_observe": (ci, [ci, ci, vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hexgnn_per_init": (ci, [ci, vp, vp, vp]),
    "hexgnn_per_update": (ci, [ci, ci, vp, vp, vp, vp, vp]),
    "hexgnn_per_update_td": (ci, [ci, ci, vp, ci, vp, C.c_double, C.c_double, vp, vp, vp, vp]),
    "hexgnn_per_sample": (ci, [ci, ci, ci, C.c_double, vp, vp, vp, vp, vp, vp]),
    "hexgnn_select_actions": (ci, [ci, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp]),
    "hexgnn_td_loss_forward": (ci, [ci, ci, vp, vp, vp, vp, ci, vp, vp, vp]),
    "hexgnn_td_loss_backward": (ci, [ci, ci, vp, vp, vp, ci, vp, vp, vp]),
    "hexgnn_td_loss_forward_backward": (ci, [ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
    "hexgnn_profile_enable": (ci, [ci]),
    "hexgnn_profile_read": (ci, [vp, vp]),
    "hexgnn_pad_rows": (ci, [ci, ci, vp, ci, vp, vp]),
    "hexgnn_unpad_rows": (ci, [ci, ci, vp, vp, ci, vp]),
}


class HexGnnError(RuntimeError):
    pass


def lib():
    """Load libhexgnn.so once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HexGnnError(
                "libhexgnn.so not found at %s -- build it with `make -C gnn_hex_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "gnn_hex_amd has no CPU/PyTorch fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.hexgnn_abi_version() != ABI_VERSION:
            raise HexGnnError("libhexgnn.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        L = lib()
        msg = L.hexgnn_strerror(rc).decode()
        if rc == -4:
            msg += " (hipError_t %d)" % L.hexgnn_last_hip_error()
        raise HexGnnError("%s failed: %s" % (what or "hexgnn call", msg))


def exported_symbols():
    """Names every entry point include/hexgnn.h declares (used by the CPU-side ABI test)."""
    return sorted(_SIGS)
