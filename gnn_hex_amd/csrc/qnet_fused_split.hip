// Split-precision ("f16x3") instantiations of the fused per-graph kernels (kernels: qnet_fused_kernels.h).
#include "qnet_fused_kernels.h"

namespace hexgnn {

#define HEXGNN_NT_SWITCH7S(nt, CALL)                    \
    switch (nt) {                                      \
        case 1: { constexpr int NT_ = 1; CALL; } break; \
        case 2: { constexpr int NT_ = 2; CALL; } break; \
        case 3: { constexpr int NT_ = 3; CALL; } break; \
        case 4: { constexpr int NT_ = 4; CALL; } break; \
        case 5: { constexpr int NT_ = 5; CALL; } break; \
        case 6: { constexpr int NT_ = 6; CALL; } break; \
        case 7: { constexpr int NT_ = 7; CALL; } break; \
        default: return HEXGNN_EUNSUPPORTED;           \
    }

int launch_qfwd_split(int nt, const QFwdArgs& a, hipStream_t st) {
    HEXGNN_NT_SWITCH7S(nt, (launch_qfwd_m<NT_, 1>(a, st)));
    return HEXGNN_OK;
}
int launch_qbwd_split(int nt, const QBwdArgs& a, hipStream_t st) {
    HEXGNN_NT_SWITCH7S(nt, (launch_qbwd_m<NT_, 1>(a, st)));
    return HEXGNN_OK;
}

}  // namespace hexgnn
