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
#define GPCA_STORE_CUNIT(UNIT) { const float co_ = ct + __shfl_xor(ct, 32); if (h == 0) cunit[(int64_t)(UNIT) * 32 + c] = co_; }

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

// One tile of a K1 epilogue: T = r o (G Q) + b s^T from the exact digit-plane sums, the unit's share of c, the column abs-max, and
// the tile on its way out through a wave-private 4 KiB of LDS so that it leaves as four 16-byte stores per lane (a lane's 16
// elements are 16 different rows: written directly they are 16 dword stores with a 64-bit address each).  `rrow` / `brow` hold r
// and b of row 32 t + c in lane c.  The sched_barrier keeps one tile's accumulators live at a time: without it hipcc read all 256
// accumulators into VGPRs first, spilled the address arithmetic to scratch and waited (vmcnt(0)) on every reload behind the store it
// had just issued -- one store round trip per element, ~19 us per round at any N (the per-round cost the shape sweep showed).
template <int BITS, bool RB_LDS, bool RELANE = false>
__device__ __forceinline__ void gq_tile_out(const i32x16 (&a)[kDigits], float rrow, float brow, const float* rl, const float* bl,
                                            double qs, float sj, int scale_out,
                                            float* __restrict__ tile, float* __restrict__ Tout, int64_t ldt, int64_t unit,
                                            float* __restrict__ cunit, float& amax, int lane_in) {
    __builtin_amdgcn_sched_barrier(0);
    // the lane id is made opaque here: everything derived from it (LDS offsets, cross-lane indices, store addresses) is then computed
    // where it is used instead of being hoisted out of the round loop as ~40 loop-invariant registers that the stage loop's register
    // pressure sent to scratch
    // (RELANE: the lane id is recomputed from an opaque mask instead of being passed in -- in k_gq_2bit a register carried across the
    //  sweep for it was spilled; in k_gq_d the passed-in form allocates better)
    int lane = lane_in;
    if constexpr (RELANE) {
        unsigned all = ~0u;
        asm volatile("" : "+s"(all));
        lane = (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
    } else {
        asm volatile("" : "+v"(lane));
    }
    const int c = lane & 31, h = lane >> 5;
    float ct = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rin = (e & 3) + 8 * (e >> 2) + 4 * h;          // row of element e inside the tile
        // r and b of that row: from the wave's LDS staging (k_gq_d: DMA-ed at the start of the round, no compiler-visible load whose
        // wait would drain the DMA queue), or from lane `rin` of registers loaded one row per lane
        const float ri = RB_LDS ? rl[rin] : __shfl(rrow, rin), bi = RB_LDS ? bl[rin] : __shfl(brow, rin);
        const float gq = (float)(combine_digits<BITS>(a, e) * qs);
        const float tv = __fmaf_rn(ri, gq, __fmul_rn(bi, sj));   // roundings pinned: every K1 variant returns the same bits
        ct = __fmaf_rn(bi, tv, ct);
        const float ov = scale_out ? __fmul_rn(ri, tv) : tv;
        amax = fmaxf(amax, fabsf(ov));
        tile[rin * 32 + c] = ov;
    }
    GPCA_STORE_CUNIT(unit)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (one wave's LDS operations execute in order: no barrier)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (lane >> 3) + 8 * i;
        const float4 v = *reinterpret_cast<const float4*>(tile + rr * 32 + 4 * (lane & 7));
        *reinterpret_cast<float4*>(Tout + (unit * 32 + rr) * ldt + 4 * (lane & 7)) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (the tile is free again before the next one is written)
    __builtin_amdgcn_sched_barrier(0);
}


}  // namespace gpca
#endif
