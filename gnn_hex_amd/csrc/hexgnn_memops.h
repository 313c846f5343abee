// Raw-buffer / LDS-DMA memory helpers shared by the fused per-graph kernels and the layer-major kernels (gfx950 only).
#pragma once
#include "hexgnn_common.h"

namespace hexgnn {

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt) but NOT for its global
// stores/loads (vmcnt).  __syncthreads() would drain vmcnt(0) first, exposing the latency of the saved-tensor stores
// (acts / agg / G are consumed by LATER kernels, never through this barrier) every layer.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Saved-tensor traffic goes through raw buffer instructions: a wave-uniform resource (SGPRs) + one 32-bit lane offset +
// an immediate, instead of a 64-bit address pair per access (the fused kernels run at the register ceiling).  A lane
// that must not take part (pad row, tensor not requested) uses the offset kOob: the hardware drops stores and returns
// zeros for loads beyond num_records, so a filler is ONE instruction, with no exec-mask branch around it.
constexpr unsigned kOob = 0x80000000u;
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void buf_store(const f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, v), r, off, 0, 0);
}
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// LDS-DMA of one 1-KiB piece (64 lanes x 16 B, lane-linear on both sides) as an asm statement: hipcc then neither
// tracks it nor drains vmcnt before later LDS accesses; the issuing wave waits with wait_vmem() before the barrier that
// publishes the bytes.  lds_dst = wave-uniform LDS byte address.
__device__ __forceinline__ void dma_piece(const void* gsrc_piece /* wave-uniform */, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    const unsigned long long sb = (unsigned long long)gsrc_piece;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const unsigned long long sbase = ((unsigned long long)hi << 32) | lo;
    // s_nop 4: lds_dst / sbase may come straight from v_readfirstlane, and a VALU-written SGPR needs 5 wait states before
    // a vector-memory instruction reads it as its base (hipcc pads nothing inside an asm string)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace hexgnn
