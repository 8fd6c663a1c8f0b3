// Device-side building blocks shared by the exact-integer GEMM sources (gemm_i8.hip: the default kernels; gemm_i8_simple.hip: the
// register-only reference kernels; fold_quantize_i8.hip: folds and digit-plane quantisation).
#ifndef GPCA_GEMM_I8_COMMON_H
#define GPCA_GEMM_I8_COMMON_H
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

namespace gpca {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define GPCA_RSRC_FLAGS 0x00020000
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc8(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, GPCA_RSRC_FLAGS);
}

// BITS = 7: four signed base-128 digits (28-bit fixed point, the default);  BITS = 8: three signed base-256 digits (24-bit,
// the packed kernels' fast mode -- plane 3 is all zero and its accumulators are never touched)
// c = b^T T is summed per 32-row unit (fixed order: 16 rows per lane half, then the two halves), one partial per unit at
// cunit[unit][32]: whichever wave of whichever launch computes a unit writes the same bits, so resident, sharded and
// streamed-panel runs agree on c exactly
// (the two halves of the wave meet through v_permlane32_swap: lanes 0-31 end up with (own, partner), lanes 32-63 with (partner, own) --
//  no lane id, which a __shfl_xor keeps in a register for the whole kernel)
__device__ __forceinline__ void halves_pair(float x, float& lo, float& hi) {
    const unsigned v = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}
#define GPCA_STORE_CUNIT(UNIT) { float lo_, hi_; halves_pair(ct, lo_, hi_); const float co_ = lo_ + hi_; if (h == 0) cunit[(int64_t)(UNIT) * 32 + c] = co_; }

// Shapes the hand-counted DMA pipelines were derived for: sample pitch a multiple of `npad_mult`, row count a multiple of `rows_mult`.
// The launchers refuse anything else (hipErrorInvalidValue -> GPCA_ERR_HIP with the kernel's name) rather than compute garbage.
static inline bool dma_shape_ok(int64_t Npad, int64_t npad_mult, int64_t rows, int64_t rows_mult) {
    return Npad > 0 && Npad % npad_mult == 0 && rows > 0 && rows % rows_mult == 0;
}

template <int BITS = 7>
__device__ __forceinline__ double combine_digits(const i32x16 (&a)[kDigits], int e) {
    // exact: each |a| < 2^31, weights are powers of two, total < 2^53
    if (BITS == 8) return (double)a[0][e] + 256.0 * (double)a[1][e] + 65536.0 * (double)a[2][e];
    return (double)a[0][e] + 128.0 * (double)a[1][e] + 16384.0 * (double)a[2][e] + 2097152.0 * (double)a[3][e];
}

// ask the scheduler for N x { 1 MFMA, V VALU } so that the next step's decode issues in the shadow of the MFMAs
template <int N, int V>
__device__ __forceinline__ void interleave_mfma_valu() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, V, 0);   // VALU
    }
}

// digit planes of one MFMA step (32 samples x 32 columns), one 16-byte operand per plane
struct Gq8Q { i32x4 q[kDigits]; };
template <int ND = kDigits>
__device__ __forceinline__ void gq8_load_q(Gq8Q& b, __amdgpu_buffer_rsrc_t rq, uint32_t qvo, uint32_t qoff) {
#pragma unroll
    for (int d = 0; d < ND; ++d) b.q[d] = __builtin_amdgcn_raw_buffer_load_b128(rq, qvo, qoff + d * 1024, 0);
}

// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {hi: 4..7, lo: 0..3}
__device__ __forceinline__ int permb(int hi, int lo, unsigned sel) { return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, sel); }

// the four k-contiguous operands of a 32-row block of G^T (sample byte t of rows 0..15 per lane half)
struct Gtt2Ops { i32x4 bt[4]; };

}  // namespace gpca
#endif
