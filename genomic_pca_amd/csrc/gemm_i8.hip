// Exact-integer variants of the two tall-skinny products on v_mfma_i32_32x32x32_i8 (GPCA_PREC_I8_EXACT): the kernels the engine runs
// by default.  (The register-only reference kernels: gemm_i8_simple.hip; folds and quantisation: fold_quantize_i8.hip.)
//
// The dosage bytes 0/1/2 ARE the int8 MFMA operand -- no conversion instruction at all -- and the skinny f32/f64
// operand X (Q or T') is split per column into kDigits = 4 signed base-128 digits of a fixed-point value
//     x_int = rint(x / colmax_j * S),  S = 0.49 * 128^4,   x_int = sum_d digit_d * 128^d,  digit_d in [-64, 63]
// (resolution colmax * 7.6e-9, finer than f32's 6e-8).  Every digit plane is one int8 MFMA with exact i32
// accumulation (|acc| <= 128 * K < 2^31 for K <= 1.6e7), planes are recombined exactly in f64
// (|sum| <= K * 2^28 < 2^53) and scaled once.  Consequences: the products are exact for the quantised operand,
// bitwise independent of the grid partition, and need 4 x 32-cycle MFMAs per 1 KiB of G instead of
// 16 x 64-cycle f32 MFMAs -> ~10x less matrix-core time per byte, so both kernels are HBM-bound
// (1 B per genotype per pass; roofline = 8 TB/s).
//
//   K1  T = r o (G Q) + b s^T     k_gq_d (int8 rows, LDS-DMA, chained rounds), k_gq_n (<= 256 samples), k_gq_2bit (2-bit rows)
//   K2  Y^T tiles = T'^T G        k_gtt_d (int8 rows, LDS-DMA, chained tasks), k_gtt_p (2-bit rows, cooperative LDS-DMA)
#include "gemm_i8_common.h"

namespace gpca {

// K1 for at most 256 samples (configs[2]'s shape class: 1 066 557 SNPs x 64 samples).  The wide kernels sweep rows padded to 256
// samples in 128-sample stages and spend most of a launch on padding and per-round prologues (0.5 TB/s at N = 64).  Here the
// digit planes of ALL of Q sit in registers (NSU x 4 steps x 4 planes), a wave streams its own range of 32-row units -- only the
// 128-byte lines that hold samples -- PF units ahead, and the launch is bound by its output (128 B of T per row against 64-128 B
// of genotypes).  Same arithmetic, same epilogue, same per-unit c partials and per-wave abs-max as the other K1 kernels.
template <int NSU>
struct GqnG { i32x4 g[4 * NSU]; float r, b; };
// (r and b of the unit's rows ride along, one row per lane: asked for in the epilogue they cost a dependent global-load latency per
//  unit -- 105 us per launch at 1 066 557 x 64 against 55 with the prefetch)
template <int NSU>
__device__ __forceinline__ void gqn_load(GqnG<NSU>& b, const int8_t* __restrict__ G, int64_t ldg, int64_t unit, int c, int h,
                                         const float* __restrict__ rv, const float* __restrict__ bv) {
    const __amdgpu_buffer_rsrc_t rg = make_rsrc8(G + unit * 32 * ldg);
    const uint32_t vo = (uint32_t)(c * ldg + 16 * h);
#pragma unroll
    for (int j = 0; j < 4 * NSU; ++j) b.g[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, vo, 32u * j, 2);   // nt: streamed once per pass
    b.r = rv[unit * 32 + c]; b.b = bv[unit * 32 + c];
}
// (On gfx9 stores count in vmcnt like loads.  With the epilogue's 16 dword stores per unit between a unit's loads and the wait for
//  them, six units in flight overran the 6-bit counter and the effective prefetch depth fell to under three units: 85 us per launch
//  at 1 066 557 x 64.  The unit's 32 x 32 tile of T is therefore turned through a wave-private 4 KiB of LDS and leaves as four
//  16-byte stores per lane -- 5 stores + 6 loads per unit, PF = 5 units stay inside the counter.)
template <int NSU, int PF>
__global__ __launch_bounds__(256, 1) void k_gq_n(const int8_t* __restrict__ G, int64_t ldg, int64_t units,
                                                  const int8_t* __restrict__ Qd, const double* __restrict__ qscale,
                                                  const float* __restrict__ rv, const float* __restrict__ bv,
                                                  const float* __restrict__ sv, float* __restrict__ Tout,
                                                  float* __restrict__ cunit, double* __restrict__ apart, int scale_out, int64_t ldt) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, waves = (int64_t)gridDim.x * 4;
    const int64_t u0 = (units * wave) / waves, u1 = (units * (wave + 1)) / waves;
    const float sj = sv[c];
    const double qs = qscale[c];
    float amax = 0.f;
    __shared__ __attribute__((aligned(16))) float tile_all[4][32 * 32];
    float* tile = tile_all[wv];
    if (u0 < u1) {
        i32x4 q[4 * NSU][kDigits];
        {
            const __amdgpu_buffer_rsrc_t rq = make_rsrc8(Qd);
#pragma unroll
            for (int j = 0; j < 4 * NSU; ++j)
#pragma unroll
                for (int d = 0; d < kDigits; ++d) q[j][d] = __builtin_amdgcn_raw_buffer_load_b128(rq, (uint32_t)(lane * 16), (uint32_t)((j * kDigits + d) * 1024), 0);
        }
        GqnG<NSU> gb[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) gqn_load<NSU>(gb[p], G, ldg, (u0 + p < u1) ? u0 + p : u1 - 1, c, h, rv, bv);
        for (int64_t u = u0; u < u1; u += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int64_t unit = u + p;
                if (unit < u1) {                                        // (wave-uniform)
                    i32x16 acc[kDigits];
#pragma unroll
                    for (int d = 0; d < kDigits; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[d][e] = 0;
#pragma unroll
                    for (int j = 0; j < 4 * NSU; ++j)
#pragma unroll
                        for (int d = 0; d < kDigits; ++d) acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(gb[p].g[j], q[j][d], acc[d], 0, 0, 0);
                    const float rrow = gb[p].r, brow = gb[p].b;
                    const int64_t nxt = unit + PF;
                    gqn_load<NSU>(gb[p], G, ldg, nxt < u1 ? nxt : u1 - 1, c, h, rv, bv);   // the slot's next unit, PF - 1 others still in flight
                    float ct = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rin = (e & 3) + 8 * (e >> 2) + 4 * h;          // row of element e inside the unit
                        const float ri = __shfl(rrow, rin), bi = __shfl(brow, rin);
                        const float gq = (float)(combine_digits(acc, e) * qs);
                        const float tv = __fmaf_rn(ri, gq, __fmul_rn(bi, sj));   // roundings pinned: every K1 variant returns the same bits
                        ct = __fmaf_rn(bi, tv, ct);
                        const float ov = scale_out ? __fmul_rn(ri, tv) : tv;
                        amax = fmaxf(amax, fabsf(ov));
                        tile[rin * 32 + c] = ov;
                    }
                    GPCA_STORE_CUNIT(unit)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (one wave's LDS operations execute in order: no barrier)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = (lane >> 3) + 8 * i;
                        const float4 v = *reinterpret_cast<const float4*>(tile + rr * 32 + 4 * (lane & 7));
                        *reinterpret_cast<float4*>(Tout + (unit * 32 + rr) * ldt + 4 * (lane & 7)) = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the tile is free again before the next unit writes it)
                }
            }
        }
    }
    float am_lo, am_hi; halves_pair(amax, am_lo, am_hi);
    const float am = fmaxf(am_lo, am_hi);
    if (h == 0) apart[wave * 32 + c] = (double)am;
}
// Npad = padded sample count of Qd's planes (multiple of 256); N = samples (<= 256)
int launch_gq_n(hipStream_t st, const int8_t* G, int64_t ldg, const GqPlan& plan, int64_t N, const int8_t* Qd,
                const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart, double* apart,
                int scale_out, int64_t ldt) {
    if (N < 1 || N > 256 || ldg < 256 || plan.units < 1) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(plan.waves / 4)), blk(256);
    if (N <= 128) hipLaunchKernelGGL((k_gq_n<1, 5>), grid, blk, 0, st, G, ldg, plan.units, Qd, qscale, r, b, s, Tout, cpart, apart, scale_out, ldt);
    else hipLaunchKernelGGL((k_gq_n<2, 4>), grid, blk, 0, st, G, ldg, plan.units, Qd, qscale, r, b, s, Tout, cpart, apart, scale_out, ldt);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------
Gtt8Plan gtt8_plan(int64_t Mpad, int64_t Npad, int target_waves) {
    Gtt8Plan p{};
    p.nblocks_n = Npad / 128;
    int64_t W = target_waves / p.nblocks_n;
    if (W < 1) W = 1;
    // a slice's digit-plane sums live in i32 accumulators: |g| <= 2 times |digit| <= 128 per row keeps 2^22 rows a factor 2 inside
    // 2^31 (only a resident matrix of > 4M rows AND > 260k samples would get there: more than one GPU holds)
    const int64_t minW = (Mpad + ((int64_t)1 << 22) - 1) >> 22;
    if (W < minW) W = minW;
    const int64_t maxW = Mpad / 128;
    if (W > maxW) W = maxW;
    int64_t rpw = (Mpad + W - 1) / W;
    rpw = (rpw + 127) / 128 * 128;       // k-blocks per wave: multiple of 4 (Mpad is a multiple of 128)
    W = (Mpad + rpw - 1) / rpw;
    p.W = (int)W;
    p.rows_per_wave = rpw;
    p.grid = ((p.nblocks_n + 3) / 4) * W;
    return p;
}

// several consecutive (row chunk, n-group) tasks per workgroup (kernels.h): W row chunks such that one batch of `grid0` workgroups
// covers the tasks evenly.  Cost model per candidate W, relative to the bytes of one sweep: batch fill (tasks rounded up to whole
// workgroup loads), ~3 stages of prologue / drain / tile store per task, the fold's read of W partial tiles, and 2 % when a row chunk's
// T' planes (16 KiB per stage) outgrow the share of an XCD's L2 they can expect to keep.
Gtt8Plan gtt8_plan_batched(int64_t Mpad, int64_t Npad, int target_waves) {
    Gtt8Plan p{};
    p.nblocks_n = Npad / 128;
    p.ngroups = (p.nblocks_n + 3) / 4;
    p.S = Mpad / 128;
    int64_t grid0 = target_waves / 8;                // the default target (2 048) = 256 workgroups = one per CU of an MI355X
    if (grid0 < 1) grid0 = 1;
    // a task's digit-plane sums live in i32 accumulators: at most 2^22 rows (32 768 stages) per task
    const int64_t wmin = std::max<int64_t>(1, (p.S + 32767) / 32768), wmax = std::min<int64_t>(p.S, 1024);
    double best = 1e300;
    int64_t bestW = wmin;
    for (int64_t W = wmin; W <= std::max(wmin, wmax); ++W) {
        const int64_t T = W * p.ngroups, k = (T + grid0 - 1) / grid0;
        double f = (double)(k * grid0) / (double)T;                                      // batch fill: k tasks per workgroup against T / grid0
        f *= 1.0 + 3.0 * (double)W / (double)p.S;                                        // per-task prologue / drain / store
        f += (double)W * 256.0 / (double)Mpad;                                           // the fold reads W x Npad x 256 B against Mpad x Npad
        if ((double)p.S / (double)W * 16384.0 > 3.0 * 1048576.0) f += 0.02;               // T' planes of a row chunk vs L2
        if (f < best - 1e-12) { best = f; bestW = W; }
    }
    p.C = (p.S + bestW - 1) / bestW;                  // stages per row chunk (the last chunk may be shorter)
    p.W = (int)((p.S + p.C - 1) / p.C);
    const int64_t T = (int64_t)p.W * p.ngroups;
    p.tasks_per_wg = (int)((T + grid0 - 1) / grid0);
    p.grid = (T + p.tasks_per_wg - 1) / p.tasks_per_wg;
    p.strided = 1;
    p.rows_per_wave = p.C * 128;                      // rows of a full task
    return p;
}

// (the slice-form plan of the narrow K2, gtt8_plan_narrow, lives with that kernel in gemm_i8_simple.hip)

// ================================================================================================
// 2-bit resident genotypes (GPCA_STORE_2BIT): the same two products with the dosage codes decoded in the prologue.
// G2 [Mpad][ld2]: 4 samples per byte (codes 0/1/2; 3 = missing, only in SNPs whose r = b = 0).  HBM traffic drops to
// 0.25 B per genotype; the kernels become matrix-core / VALU bound.
// ================================================================================================

template <int NT> __device__ __forceinline__ void gqd_dma(uint32_t lds_addr, uint32_t voff, i32x4 rsrc, uint32_t soff);   // (defined with k_gq_d below)
__device__ __forceinline__ i32x4 gqd_rsrc(const void* p);
#ifndef GPCA_ABLATE
#define GPCA_ABLATE 0   // scripts/kbench/kbench_gq2.hip: bit0 no decode, bit1 no Q loads, bit2 no G loads, bit3 no MFMA,
#endif                  // bit4 clock stamps (s_memtime / s_memrealtime per workgroup into g_kbench_stamp)
#if GPCA_ABLATE & 16
__device__ unsigned long long g_kbench_stamp[2 * 4096];
__device__ unsigned long long g_kbench_abs[2 * 4096];     // s_memrealtime (100 MHz) at the start and at the end of every workgroup
#endif
// 16 samples (one 32-bit word of 2-bit codes) -> 16 int8 bytes:  per output dword 5 VALU ops (bfe, 2 x (lshl_or, and))
__device__ __forceinline__ i32x4 spread16(unsigned w) {
    i32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned t = (w >> (8 * q)) & 0xffu;
        t = (t | (t << 12)) & 0x000F000Fu;
        t = (t | (t << 6)) & 0x03030303u;
        o[q] = (int)t;
    }
    return o;
}

// ---- K1, packed.  A lane's 16-byte load = 64 consecutive samples of its SNP row (half h of a 128-sample block);
// MFMA step s of block b contracts samples {128b + 64h + 16s + j}: the digit planes of Q are stored in that order
// (quantize layout 1).  Four blocks (one 128-byte line of the row) are requested back to back.
template <int R>
struct Gq2G { i32x4 g[4][R]; };

template <int R>
__device__ __forceinline__ void gq2_load_g(Gq2G<R>& b, __amdgpu_buffer_rsrc_t rg, const uint32_t (&gvo)[R], uint32_t s0) {
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) b.g[j][t] = __builtin_amdgcn_raw_buffer_load_b128(rg, gvo[t], s0 + 32u * j, 0);
}

// decode the R operands of MFMA step S of one 128-sample block (one dword per tile)
template <int R, int S>
__device__ __forceinline__ void gq2_decode(const i32x4 (&g)[R], i32x4 (&op)[R]) {
#pragma unroll
    for (int t = 0; t < R; ++t) op[t] = spread16((unsigned)g[t][S]);
}
__device__ __forceinline__ int spread4(unsigned w, int q) {   // byte q of w (4 samples) -> 4 int8 bytes
    unsigned t = (w >> (8 * q)) & 0xffu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    return (int)t;
}
// The R x 4 int8 MFMAs of the current step, each followed by one micro-step (5 VALU ops = one output dword) of the
// NEXT step's decode; sched_barrier(0) pins the order so the VALU work issues in the shadow of the matrix pipe.
template <int R, int SN, int ND = kDigits>
__device__ __forceinline__ void gq2_mfma_decode(const i32x4 (&op)[R], const Gq8Q& q, i32x16 (&acc)[R][kDigits],
                                                const i32x4 (&gn)[R], i32x4 (&opn)[R], const unsigned* lut) {
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
#pragma unroll
        for (int t = 0; t < R; ++t) {
            if (d >= ND) { /* three-plane mode: this slot carries only its look-up */ }
            else if (GPCA_ABLATE & 8) acc[t][d][0] ^= op[t][d] ^ q.q[d][t & 3];
            else acc[t][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(op[t], q.q[d], acc[t][d], 0, 0, 0);
            const int m = d * R + t;          // 4R MFMAs <-> 4R output dwords (tile m / 4, dword m % 4)
            if (GPCA_ABLATE & 1) opn[m >> 2][m & 3] = gn[m >> 2][SN];
            else opn[m >> 2][m & 3] = (int)lut[((((unsigned)gn[m >> 2][SN] >> (8 * (m & 3))) & 0xffu) << 5)];   // 2 VALU + 1 LDS read (lut = this lane's copy)
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <int R, int ND = kDigits>
__device__ __forceinline__ void gq2_group(const uint8_t* __restrict__ G2, int64_t ld2, int64_t nsuper,
                                          const int8_t* __restrict__ Qd, const double* __restrict__ qscale, const float* __restrict__ rv,
                                          const float* __restrict__ bv, const float* __restrict__ sv, float* __restrict__ Tout, int scale_out, int64_t ldt,
                                          float* __restrict__ cunit, float& amax, int64_t row0, int lane_in, const unsigned* lut) {
    // (the lane's place is made opaque per group and the column's scales are fetched in the epilogue: less is carried through the decode
    //  loop, where the four-plane kernel has no register to spare.  It parked 9 VGPRs in scratch; now 5, all written before the loop and
    //  read back in the epilogues -- the loop itself never touched scratch, before or after)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int c = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rg = make_rsrc8(G2 + row0 * ld2);
    uint32_t gvo[R];
#pragma unroll
    for (int t = 0; t < R; ++t) gvo[t] = (uint32_t)((32 * t + c) * ld2 + 16 * h);
    const uint32_t qvo = (uint32_t)(lane * 16);
    constexpr uint32_t QCH = kDigits * 1024;   // bytes of digit planes per MFMA step
    const uint32_t nsteps = (uint32_t)(nsuper * 16);

    i32x16 acc[R][kDigits];
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int d = 0; d < kDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][d][e] = 0;

    Gq2G<R> GA, GB;
    Gq8Q Q0, Q1, Q2, Q3;
    i32x4 opA[R], opB[R];
    const __amdgpu_buffer_rsrc_t rq = make_rsrc8(Qd);   // digit planes are < 2 GiB: one descriptor
    gq2_load_g<R>(GA, rg, gvo, 0u);
    gq8_load_q<ND>(Q0, rq, qvo, 0u); gq8_load_q<ND>(Q1, rq, qvo, QCH); gq8_load_q<ND>(Q2, rq, qvo, 2 * QCH);
    if (GPCA_ABLATE & 2) Q3 = Q0;
    if (GPCA_ABLATE & 4) GB = GA;
    gq2_decode<R, 0>(GA.g[0], opA);
    // One phase = one MFMA step: prefetch the digit planes 3 steps ahead; the R x 4 int8 MFMAs of this step (matrix
    // pipe) and the bit-spreading of the NEXT step's operands (VALU, 5 ops per dword) sit in one scheduling region so
    // that they interleave.  Operand sets alternate opA / opB.
#if GPCA_ABLATE & 32
    // upper bound of sharing the planes through LDS (timing only, nothing synchronised): each wave brings ONE plane of the step into
    // a ring by LDS-DMA and reads all ND back with ds_read_b128
    __shared__ i32x4 abl_qring[4][kDigits][64];
    const int abl_wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i32x4 abl_rsrc = gqd_rsrc(Qd);
#define GQ2_ABL_SHARED(QN, NST) \
    if (abl_wv < ND) gqd_dma<0>((uint32_t)(uintptr_t)&abl_qring[(NST) & 3u][abl_wv][0], qvo, abl_rsrc, ((NST) < nsteps ? (NST) : 0u) * QCH + abl_wv * 1024u); \
    _Pragma("unroll") for (int d_ = 0; d_ < ND; ++d_) QN.q[d_] = abl_qring[(NST) & 3u][d_][lane];
#else
#define GQ2_ABL_SHARED(QN, NST)
#endif
#define GQ2_PHASE(OPCUR, OPNXT, GNXT, BN, SN, QCUR, QNEXT, STEP)                            \
    { const uint32_t nst_ = (STEP) + 3u;                                                     \
      if (GPCA_ABLATE & 32) { GQ2_ABL_SHARED(QNEXT, nst_) }                                    \
      else if (!(GPCA_ABLATE & 2)) gq8_load_q<ND>(QNEXT, rq, qvo, (nst_ < nsteps ? nst_ : 0u) * QCH); }  \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    gq2_mfma_decode<R, SN, ND>(OPCUR, QCUR, acc, GNXT.g[BN], OPNXT, lut);                        \
    __builtin_amdgcn_sched_barrier(0);
    // block B of buffer GCUR: steps (B,0..3); the step after (B,3) is (BNX,0) of buffer GNX
#define GQ2_BLOCK(GCUR, B, GNX, BNX, STEP0)                                   \
    GQ2_PHASE(opA, opB, GCUR, B, 1, Q0, Q3, (STEP0) + 0u)                       \
    GQ2_PHASE(opB, opA, GCUR, B, 2, Q1, Q0, (STEP0) + 1u)                       \
    GQ2_PHASE(opA, opB, GCUR, B, 3, Q2, Q1, (STEP0) + 2u)                       \
    GQ2_PHASE(opB, opA, GNX, BNX, 0, Q3, Q2, (STEP0) + 3u)
    for (int64_t sc = 0; sc < nsuper; sc += 2) {        // nsuper (512-sample super-chunks) is even
        const uint32_t st0 = (uint32_t)(sc * 16);
        const uint32_t more = (sc + 2 < nsuper) ? 1u : 0u;
        if (!(GPCA_ABLATE & 4)) gq2_load_g<R>(GB, rg, gvo, (uint32_t)((sc + 1) * 128));
        GQ2_BLOCK(GA, 0, GA, 1, st0) GQ2_BLOCK(GA, 1, GA, 2, st0 + 4u) GQ2_BLOCK(GA, 2, GA, 3, st0 + 8u)
        GQ2_BLOCK(GA, 3, GB, 0, st0 + 12u)
        if (!(GPCA_ABLATE & 4)) gq2_load_g<R>(GA, rg, gvo, (uint32_t)((sc + 2 * more) * 128));   // last trip: re-loads its own first chunk (unused)
        GQ2_BLOCK(GB, 0, GB, 1, st0 + 16u) GQ2_BLOCK(GB, 1, GB, 2, st0 + 20u) GQ2_BLOCK(GB, 2, GB, 3, st0 + 24u)
        GQ2_BLOCK(GB, 3, GA, 0, st0 + 28u)
    }
#undef GQ2_BLOCK
#undef GQ2_PHASE
#undef GQ2_ABL_SHARED
    const float sj = sv[c];           // the column's sum of Q and digit scale, fetched here (two registers less through the loop above)
    const double qs = qscale[c];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        float ct = 0.f;     // this tile's share of c = b^T T: one partial per 32-row unit, so c does not depend on the grid partition
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = row0 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float ri = rv[row], bi = bv[row];
            const float gq = (float)(combine_digits<ND == 3 ? 8 : 7>(acc[t], e) * qs);
            const float tv = __fmaf_rn(ri, gq, __fmul_rn(bi, sj));   // roundings pinned: every K1 variant returns the same bits
            ct = __fmaf_rn(bi, tv, ct);
            const float ov = scale_out ? __fmul_rn(ri, tv) : tv;
            amax = fmaxf(amax, fabsf(ov));
            Tout[row * ldt + c] = ov;      // (a streaming store measured the same here: the packed kernel is bound by its decode, not by HBM)
        }
        GPCA_STORE_CUNIT(row0 / 32 + t)
    }
}

template <int ND, int RMAX = 4>
__global__ __launch_bounds__(256, RMAX == 4 ? 1 : 2) void k_gq_2bit(const uint8_t* __restrict__ G2, int64_t ld2, int64_t units, int64_t nsuper,
                                                     const int8_t* __restrict__ Qd, const double* __restrict__ qscale,
                                                     const float* __restrict__ rv, const float* __restrict__ bv,
                                                     const float* __restrict__ sv, float* __restrict__ Tout,
                                                     float* __restrict__ cpart, double* __restrict__ apart, int scale_out, int64_t ldt) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // ONE register carries the thread's place through the kernel; what the epilogues and the last store need is derived there
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t waves = (int64_t)gridDim.x * 4;
    // (32-bit unit counters, scalars by construction: the 64-bit `u + 4 <= u_end` is a VALU compare -- there is no scalar signed 64-bit
    //  less-than -- whose operands lived in VGPR pairs through the whole kernel and were spilled to scratch by the four-plane form)
    int u = __builtin_amdgcn_readfirstlane((int)((units * wave) / waves));
    const int u_end = __builtin_amdgcn_readfirstlane((int)((units * (wave + 1)) / waves));
    float amax = 0.f;
    // byte (4 two-bit codes) -> 4 int8 bytes: 256-entry table in LDS; the spread then costs 2 VALU + 1 ds_read per dword
    // 32 interleaved copies (entry v of copy j at word 32 v + j): lane l reads copy l % 32, i.e. always bank l % 32 -- a single
    // 1-KiB table made 69 % of this kernel's LDS cycles bank conflicts (random bytes of 32 lanes over 32 banks)
    __shared__ unsigned lut_all[256 * 32];
    for (int e = threadIdx.x; e < 256 * 32; e += 256) lut_all[e] = (unsigned)spread4((unsigned)(e >> 5), 0);
    const unsigned* lut = lut_all + (lane & 31);
    __syncthreads();
#if GPCA_ABLATE & 16
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (RMAX == 4)
        for (; u + 4 <= u_end; u += 4) gq2_group<4, ND>(G2, ld2, nsuper, Qd, qscale, rv, bv, sv, Tout, scale_out, ldt, cpart, amax, (int64_t)u * 32, lane, lut);
    else        // (kbench experiment: two tiles per sweep, two waves per SIMD)
        for (; u + 2 <= u_end; u += 2) gq2_group<2, ND>(G2, ld2, nsuper, Qd, qscale, rv, bv, sv, Tout, scale_out, ldt, cpart, amax, (int64_t)u * 32, lane, lut);
    // the 1-3 units left over go in ONE more sweep over the samples (a wave with 31 units used to make two, of 2 and of 1 tile:
    // every sweep re-reads all of Q's planes and pays its prologue)
    if (RMAX == 4 && u + 3 <= u_end) { gq2_group<3, ND>(G2, ld2, nsuper, Qd, qscale, rv, bv, sv, Tout, scale_out, ldt, cpart, amax, (int64_t)u * 32, lane, lut); u += 3; }
    else if (u + 2 <= u_end) { gq2_group<2, ND>(G2, ld2, nsuper, Qd, qscale, rv, bv, sv, Tout, scale_out, ldt, cpart, amax, (int64_t)u * 32, lane, lut); u += 2; }
    else if (u + 1 <= u_end) { gq2_group<1, ND>(G2, ld2, nsuper, Qd, qscale, rv, bv, sv, Tout, scale_out, ldt, cpart, amax, (int64_t)u * 32, lane, lut); u += 1; }
    float am_lo, am_hi; halves_pair(amax, am_lo, am_hi);
    const float am = fmaxf(am_lo, am_hi);
    {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        if ((t2 & 32) == 0) apart[wave * 32 + (t2 & 31)] = (double)am;
    }
#if GPCA_ABLATE & 16
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_kbench_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
        const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();
        g_kbench_stamp[2 * blockIdx.x + 1] = r1_ - st_r0;
        g_kbench_abs[2 * blockIdx.x] = st_r0; g_kbench_abs[2 * blockIdx.x + 1] = r1_;
    }
#endif
}

void launch_gq_2bit(hipStream_t st, const uint8_t* G2, int64_t ld2, const GqPlan& plan, int64_t Npad, const int8_t* Qd,
                    const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart,
                    double* apart, int scale_out, int nd, int64_t ldt) {
    const dim3 grid((unsigned)(plan.waves / 4)), blk(256);
    const int64_t nsuper = Npad / 512;   // Npad is a multiple of 1024 in 2-bit mode -> even
    if (nd == 3) hipLaunchKernelGGL(k_gq_2bit<3>, grid, blk, 0, st, G2, ld2, plan.units, nsuper, Qd, qscale, r, b, s, Tout, cpart, apart, scale_out, ldt);
    else hipLaunchKernelGGL(k_gq_2bit<kDigits>, grid, blk, 0, st, G2, ld2, plan.units, nsuper, Qd, qscale, r, b, s, Tout, cpart, apart, scale_out, ldt);
}

// ================================================================================================
// K1 by LDS-DMA (int8-resident genotypes).  Operand-shaped loads (one wave instruction = 32 rows x 32 B) keep the
// texture-address unit twice as busy as full-line requests and cap this kernel near 4.9 TB/s; here every genotype byte
// arrives as 8-row x 128-byte pieces written straight into LDS by `buffer_load_dwordx4 ... lds` (1 KiB per wave
// instruction, no VGPR destination) and is read back in operand shape with ds_read_b128.
//   * unit = one 32-row tile x one 128-sample stage = 4 KiB = 4 pieces.  The LDS image of a piece is lane-linear
//     (lane l -> base + 16 l); lane l fetches row r = 8 i + (l >> 3), 16-byte chunk (l & 7) ^ ((r >> 1) & 7), so row r keeps
//     chunk k at position k ^ ((r >> 1) & 7) and the operand read (row c, chunk 2 s + h) is bank-conflict free for
//     ds_read_b128's lane groups.
//   * each wave owns a ring of 6 unit slots and walks its R = 4 tiles tile-outer inside a stage (the 16 digit-plane
//     operands of the stage sit in registers), so units are consumed strictly in order; the unit just finished is
//     re-filled with the unit 6 ahead.  The digit planes of stage s+1 are DMA-ed (wave w: plane w) at the start of
//     stage s behind the workgroup barrier.
//   * the DMAs are inline asm, invisible to the compiler's s_waitcnt bookkeeping: completion is counted by hand.
//     Issue order per stage:  Q(s+1), G(4s+6), G(4s+7), G(4s+8), G(4s+9)   (4 instructions each).
//     - stage start, vmcnt(16): Q(s) was issued one stage ago, 4 unit fills (16) are younger.
//     - before the first read of unit m (at step 3 of unit m-1), G(m+1..m+4) are younger (16) plus one plane batch
//       (4), two when m = 1 mod 4:  vmcnt(20) / vmcnt(24).
//     The prologue issues G0 G1 Q0 G2 G3 G4 G5 so that the same counts hold from the first stage on.
//   A wave reads only its own unit slots: its counted vmcnt orders those reads; the planes are shared, so their wait is
//   followed by the workgroup barrier.
//   * rounds of fewer tiles (R = 3, 2, 1; ring of 6).  A workgroup's range rarely ends on a full round of 16 units (1M rows on
//     256 workgroups: 122 units = 7 full rounds + 10 units); the short last round used to sweep the samples with four tiles per
//     wave all the same (the missing ones re-reading tile 0), an eighth round's time for 0.63 rounds of work = 4.6 % of the launch.
//     Now a wave with nv < 4 units runs the round with R = nv tiles -- its own instantiation, beside waves of the same workgroup
//     that run another R: the stage barrier and the plane DMAs (wave w: plane w) are the same for every R.  Counts for R tiles:
//     - stage start: the R refills of the previous stage are younger than Q(s): vmcnt(4 R) (stricter than needed in stage 0,
//       where the prologue's 4 fills follow Q(0)).
//     - before the first read of unit m + 1 (during unit m, before m's own refill): fills m+2..m+5 are younger (16) plus 4 per
//       stage start since fill m+1 was issued at the end of unit m-5, i.e. per unit u in [m-4, m] with u mod R = 0.  In the steady
//       state that depends on t = m mod R only (gqd_q_young); the first units of a round have fewer stage starts behind them, so
//       the stages before gqd_early_stages(R) wait with the smallest count any of them needs (a stricter wait is always safe).
// ================================================================================================
constexpr int kGqdSlots = 6;
#ifndef GPCA_STAMP
#define GPCA_STAMP 0     // scripts/kbench/kbench_gqd.hip: s_memrealtime stamps (100 MHz) of wave 0 of every workgroup at the round boundaries
#endif
#if GPCA_STAMP
__device__ unsigned long long g_gqd_stamp[1024 * 64];     // [workgroup][slot]: 0 = start, 1 + 2 r = stage loop of round r done, 2 + 2 r = its epilogue done
__device__ int g_gqd_stamp_n;
#define GQD_STAMP(SLOT) { if (threadIdx.x == 0 && blockIdx.x < 1024 && (SLOT) < 64) g_gqd_stamp[blockIdx.x * 64 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); }
__device__ unsigned long long g_gqd_stage_stamp[256 * 3 * 128];     // [workgroup][round < 3][stage < 128]: start of every stage (GPCA_STAMP=2)
#define GQD_STAGE_STAMP(RND, ST) { if (GPCA_STAMP == 2 && threadIdx.x == 0 && blockIdx.x < 256 && (RND) < 3 && (ST) < 128) g_gqd_stage_stamp[(blockIdx.x * 3 + (RND)) * 128 + (ST)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define GQD_STAMP(SLOT)
#define GQD_STAGE_STAMP(RND, ST)
#endif
// plane batches younger than the fill of unit m + 1 while unit m (tile t = m mod R) is consumed, steady state (m >= 4)
constexpr int gqd_q_young(int R, int t) {
    int n = 0;
    for (int j = 0; j <= 4; ++j) n += (((t - j) % R + R) % R == 0) ? 1 : 0;
    return n;
}
constexpr int gqd_early_stages(int R) { return R == 3 ? 2 : (R == 2 ? 2 : 4); }   // stages holding a unit m < 4 (R = 1: 0..3)
constexpr int gqd_q_young_early(int R) { return R == 1 ? 2 : 1; }                  // fewest plane batches behind any unit of those stages
static_assert(gqd_q_young(4, 0) == 2 && gqd_q_young(4, 1) == 1 && gqd_q_young(4, 2) == 1 && gqd_q_young(4, 3) == 1, "R = 4: vmcnt 24/20/20/20");
static_assert(gqd_q_young(3, 0) == 2 && gqd_q_young(3, 1) == 2 && gqd_q_young(3, 2) == 1, "R = 3: vmcnt 24/24/20");
static_assert(gqd_q_young(2, 0) == 3 && gqd_q_young(2, 1) == 2 && gqd_q_young(1, 0) == 5, "R = 2: vmcnt 28/24; R = 1: 36");
struct GqdSmem {
    i32x4 q[2][4][kDigits][64];          // digit planes: [slot][step][digit][lane]          32 KiB
    i32x4 g[4][kGqdSlots][256];          // genotype units: [wave][slot][piece i][lane]      96 KiB
    float tile[4][32 * 32];              // a wave's 32 x 32 tile of T on its way out        16 KiB
    float rb[4][2][128];                 // r and b of the wave's rows of the round             4 KiB
};

template <int NT = 0>
__device__ __forceinline__ void gqd_dma(uint32_t lds_addr, uint32_t voff, i32x4 rsrc, uint32_t soff) {
    unsigned keep;
    if (NT)      // streamed once per pass: non-temporal
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// 64 consecutive dwords (one per lane) straight into LDS at lds_addr + 4 * lane
__device__ __forceinline__ void gqd_dma_dword(uint32_t lds_addr, uint32_t voff, i32x4 rsrc, uint32_t soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void gqd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// the same with the count chosen by a value that is a constant after unrolling (16 + 4 x plane batches)
__device__ __forceinline__ void gqd_wait_young(int qb) {
    switch (qb) {
        case 1: gqd_wait_vm<20>(); break;
        case 2: gqd_wait_vm<24>(); break;
        case 3: gqd_wait_vm<28>(); break;
        case 4: gqd_wait_vm<32>(); break;
        default: gqd_wait_vm<36>(); break;
    }
}
__device__ __forceinline__ i32x4 gqd_rsrc(const void* p) {
    const uint64_t a = (uint64_t)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
    r.z = 0x7fffffff;
    r.w = GPCA_RSRC_FLAGS;
    return r;
}

// One tile of a K1 epilogue: T = r o (G Q) + b s^T from the exact digit-plane sums, the unit's share of c, the column abs-max, and
// the tile on its way out through a wave-private 4 KiB of LDS so that it leaves as four 16-byte stores per lane (a lane's 16
// elements are 16 different rows: written directly they are 16 dword stores with a 64-bit address each).  `rrow` / `brow` hold r
// and b of row 32 t + c in lane c.  The sched_barrier keeps one tile's accumulators live at a time: without it hipcc read all 256
// accumulators into VGPRs first, spilled the address arithmetic to scratch and waited (vmcnt(0)) on every reload behind the store it
// had just issued -- one store round trip per element, ~19 us per round at any N (the per-round cost the shape sweep showed).
#ifndef GPCA_T_NT_MODE
#define GPCA_T_NT_MODE 2      // (harness: 0 never, 1 every round, 2 every round but a workgroup's last)
#endif
template <int BITS, bool RB_LDS>
__device__ __forceinline__ void gq_tile_out(const i32x16 (&a)[kDigits], float rrow, float brow, const float* rl, const float* bl,
                                            double qs, float sj, int scale_out,
                                            float* __restrict__ tile, float* __restrict__ Tout, int64_t ldt, int64_t unit,
                                            float* __restrict__ cunit, float& amax, int lane_in, bool stream_store = false) {
    __builtin_amdgcn_sched_barrier(0);
    // the lane id is made opaque here: everything derived from it (LDS offsets, cross-lane indices, store addresses) is then computed
    // where it is used instead of being hoisted out of the round loop as ~40 loop-invariant registers that the stage loop's register
    // pressure sent to scratch
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
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
        // Streaming store when another round follows: written through instead of left dirty in L2 for the next round's genotype
        // reads to evict one line at a time -- that write-back, interleaved with the reads, made the first ten stages of every round
        // ~25 % slower (stage stamps, profiles/r4_kbench_summary.md section 6; k_gq_d 1.563 -> 1.550 ms at 1M x 10k).  A workgroup's
        // LAST round keeps the plain store: nothing streams behind it, and the launch does not have to wait for HBM to take the tile.
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v vv = {v.x, v.y, v.z, v.w};
        f4v* dst = reinterpret_cast<f4v*>(Tout + (unit * 32 + rr) * ldt + 4 * (lane & 7));
        // (the streaming form is inline asm: written as two C++ stores under an if, the optimiser merges them into one plain store --
        //  the common metadata of the two -- and the hint is gone)
        if (GPCA_T_NT_MODE == 1 || (GPCA_T_NT_MODE == 2 && stream_store)) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(dst), "v"(vv) : "memory");   // (s_nop: a VALU write of the data registers needs two wait states behind a store wider than 8 bytes, and the hazard recogniser does not read asm)
        else *dst = vv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (the tile is free again before the next one is written)
    __builtin_amdgcn_sched_barrier(0);
}

// What a wave does in one round: R tiles starting at row unit `unit0`, `nv` of them real (nv = 0: the wave rides along on the round's
// first unit for the planes and the barriers and stores nothing).
struct GqdRound { int64_t unit0; int nv; int R; };

// CHAINED rounds.  The last six refills of a round used to wrap to stage 0 of the same tiles and were thrown away, the ring was
// drained, and the next round started cold (prologue, first-fill latency).  Now those six refills fetch the first six units of the
// wave's NEXT round (in that round's own order: stage-major over its R' tiles) and the plane batch issued in the last stage is
// Q(0) of the next round -- the planes depend on the stage only -- so the issue order across the boundary is the steady state's
//     ... G'0 G'1 | Q'(0) G'2 G'3 G'4 G'5 |        (= the prologue's order G0 G1 Q0 G2 G3 G4 G5)
// and the next round starts where a prologue would have left it, with its data landing behind this round's epilogue.  The ring
// position carries over (`rslot`: unit m + 6 always goes into the slot of unit m).  Nothing is drained at the boundary: the epilogue's
// stores (and the next round's r / b fetch) are younger than every prefetch, and a counted wait stays correct with MORE operations
// younger than its target -- it needs at least `count` LOADS younger than the load it waits for (loads complete in order), which the
// steady-state issue order provides.
// Needs more than 6 units per round (nstage >= 8); shorter sample axes keep the drained form.
template <int NT, int R>
__device__ __forceinline__ void gqd_round(const int8_t* __restrict__ G, int64_t ldg, int64_t nstage,
                                          const int8_t* __restrict__ Qd, GqdSmem* sm, int wv, int lane, int c, int h,
                                          int64_t unit0, int nvalid, bool prologue, bool chain, GqdRound nx, uint32_t& rslot, int round_ix, uint32_t ph,
                                          int64_t rows_total, double qs, const float* __restrict__ rv,
                                          const float* __restrict__ bv, float sj, float* __restrict__ Tout, int scale_out, int64_t ldt,
                                          float* __restrict__ cunit, float& amax) {
    constexpr int S = kGqdSlots;
    static_assert(R >= 1 && R <= 4, "tiles per wave");
    // the workgroup sweeps the sample axis starting at stage `ph` (and wraps): the integer sums do not depend on the order, and
    // workgroups that start at different columns do not all ask the memory system for the same column of 131 072 rows at once
    auto phys = [&](uint32_t st_) -> uint32_t { const uint32_t p_ = st_ + ph; return p_ >= (uint32_t)nstage ? p_ - (uint32_t)nstage : p_; };
    const int64_t row0 = unit0 * 32;
    const i32x4 rg = gqd_rsrc(G + row0 * ldg);
    const i32x4 rgn = gqd_rsrc(G + nx.unit0 * 32 * ldg);          // the next round's rows (chain)
    const i32x4 rq = gqd_rsrc(Qd + wv * 1024);
    const uint32_t ld32 = (uint32_t)ldg;
    // row r keeps chunk k at position k ^ ((r >> 1) & 7): with ds_read_b128's lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...)
    // the 8 even and the 8 odd rows of a group then land on 16 distinct bank quads (k ^ (r & 7) was 2-way: SQ_LDS_BANK_CONFLICT =
    // a third of the LDS cycles).  Piece i holds rows 8 i + (l >> 3), so (r >> 1) & 7 = (4 i + (l >> 4)) & 7: one offset per parity of i.
    const uint32_t gvo_e = (uint32_t)(lane >> 3) * ld32 + 16u * (uint32_t)((lane & 7) ^ (lane >> 4));
    const uint32_t gvo_o = (uint32_t)(lane >> 3) * ld32 + 16u * (uint32_t)((lane & 7) ^ (lane >> 4) ^ 4);
    const uint32_t qvo = (uint32_t)(lane * 16);
    uint32_t toff[R];                                   // wave-uniform byte offset of tile t (tiles past the range: tile 0)
#pragma unroll
    for (int t = 0; t < R; ++t) toff[t] = (uint32_t)(t < nvalid ? 32 * t : 0) * ld32;
    // r and b of the round's rows (row0 ... row0 + 32 R) go into the wave's LDS staging by DMA, issued before anything else of the
    // round; the epilogue reads them with ds_read.  (As global loads into registers their first use -- the epilogue -- carried a
    // compiler-inserted vmcnt wait that knew nothing of the DMAs in flight and drained them.)  Extra DMAs never invalidate a counted
    // wait: a count stays correct as long as at least that many LOADS are younger than the one waited for.  Rows past the end of
    // the launch's row range read as zero (num_records).
    {
        const uint32_t lds_rb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sm->rb[wv][0][0];
        i32x4 rr_ = gqd_rsrc(rv + row0), rb_ = gqd_rsrc(bv + row0);
        rr_.z = rb_.z = (int)((rows_total - row0) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the previous round's epilogue has read its r and b
#pragma unroll
        for (int j = 0; j < (R + 1) / 2; ++j) {
            gqd_dma_dword(lds_rb + 256u * j, qvo >> 2, rr_, 256u * j);
            gqd_dma_dword(lds_rb + 512u + 256u * j, qvo >> 2, rb_, 256u * j);
        }
    }
    constexpr uint32_t QCH = kDigits * 1024;
    const uint32_t lds_q = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sm->q[0][0][0][0] + (uint32_t)wv * 1024u;
    const uint32_t lds_g = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sm->g[0][0][0] + (uint32_t)wv * (S * 4096u);
    const char* gl = reinterpret_cast<const char*>(&sm->g[wv][0][0]);
    uint32_t loff[4];                                   // operand read: row c, chunk 2 s + h at position (2 s + h) ^ ((c >> 1) & 7)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) loff[s4] = (uint32_t)(c * 128 + (((2 * s4 + h) ^ ((c >> 1) & 7)) * 16));

    i32x16 acc[R][kDigits];
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int d = 0; d < kDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][d][e] = 0;

    // fill one unit (rows of resource RSRC at byte offset SO) into ring slot `slot`
#define GQD_ISSUE_G(RSRC, SO, SLOT)                                                                       \
    {                                                                                                     \
        const uint32_t so_ = (SO);                                                                        \
        const uint32_t la_ = lds_g + (uint32_t)(SLOT) * 4096u;                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) gqd_dma<NT>(la_ + 1024u * i, (i & 1) ? gvo_o : gvo_e, RSRC, so_ + 8u * i * ld32); \
    }
#define GQD_ISSUE_Q(ST, QS)                                                                               \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
            gqd_dma(lds_q + (uint32_t)(((QS) * 4 + j) * kDigits) * 1024u, qvo, rq, (phys((uint32_t)(ST)) * 4u + j) * QCH); \
    }
    // ring of S slots: the unit consumed is re-filled with the unit S ahead, units in the order they are consumed (stage-major,
    // R tiles per stage).  Past the last stage the fills go on with the next round's units (chain) or wrap to stage 0 and are
    // never read (last round, short sample axes).  S - 4 units precede Q(0) in the prologue so that the steady-state counts hold
    // from the first stage on (G0 G1 Q0 G2..G5).
    int64_t ist = 0;         // next unit to issue: (stage ist, tile it) of this round ...
    int it = 0;
    bool in_next = false;    // ... or (stage pst, tile pt) of the next
    uint32_t pst = 0; int pt = 0;
#define GQD_ISSUE_NEXT(SLOT)                                                                              \
    {                                                                                                     \
        if (!in_next) {                                                                                   \
            GQD_ISSUE_G(rg, toff[it] + phys((uint32_t)ist) * 128u, SLOT)                                  \
            if (++it == R) { it = 0; if (++ist == nstage) { if (chain) in_next = true; else ist = 0; } }  \
        } else {                                                                                          \
            const uint32_t tn_ = (uint32_t)(pt < nx.nv ? 32 * pt : 0) * ld32;                             \
            GQD_ISSUE_G(rgn, tn_ + phys(pst) * 128u, SLOT)                                                \
            if (++pt == nx.R) { pt = 0; ++pst; }                                                          \
        }                                                                                                 \
    }
    if (prologue) {
        rslot = 0;
        GQD_ISSUE_NEXT(0) GQD_ISSUE_NEXT(1)
        GQD_ISSUE_Q(0, 0)
#pragma unroll
        for (int sl = S - 4; sl < S; ++sl) GQD_ISSUE_NEXT(sl)
    } else {                 // the previous round left units 0..5 of this one in flight (slots rslot ...) and Q(0) in plane slot 0
        ist = S / R; it = S % R;
    }
    i32x4 gcur, gnxt;

    for (int64_t st = 0; st < nstage; ++st) {
        // Q(st) landed in every wave's plane: the R refills of the previous stage are younger
        if constexpr (R == 4) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        else if constexpr (R == 3) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        else if constexpr (R == 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        GQD_STAGE_STAMP(round_ix, st)
        GQD_ISSUE_Q(st + 1 < nstage ? st + 1 : 0, (int)((st + 1) & 1))
        i32x4 q[4][kDigits];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int d = 0; d < kDigits; ++d) q[s4][d] = sm->q[st & 1][s4][d][lane];
        if (st == 0) {                                   // unit 0 landed with Q(0)
            gcur = *reinterpret_cast<const i32x4*>(gl + rslot * 4096u + loff[0]);
            gnxt = *reinterpret_cast<const i32x4*>(gl + rslot * 4096u + loff[1]);
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const char* ub = gl + rslot * 4096u;
            const uint32_t nslot = rslot == S - 1 ? 0u : rslot + 1u;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                // operand reads run two MFMA groups ahead of their use (one group = 128 cycles did not always cover the LDS
                // latency beside the DMA writes): steps 2 and 3 already read steps 0 and 1 of the next unit
                i32x4 gfar;
                if (s4 < 2) {
                    gfar = *reinterpret_cast<const i32x4*>(ub + loff[s4 + 2]);
                } else {
                    if (s4 == 2) {
                        // younger than the next unit: S - 2 unit fills + 1 plane batch (2 when a stage start falls in the window)
                        if constexpr (R == 4) {
                            if (t == 0) gqd_wait_vm<24>(); else gqd_wait_vm<20>();
                        } else {
                            if (st < gqd_early_stages(R)) gqd_wait_young(gqd_q_young_early(R));
                            else gqd_wait_young(gqd_q_young(R, t));
                        }
                    }
                    gfar = *reinterpret_cast<const i32x4*>(gl + nslot * 4096u + loff[s4 - 2]);
                }
#pragma unroll
                for (int d = 0; d < kDigits; ++d)
                    acc[t][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(gcur, q[s4][d], acc[t][d], 0, 0, 0);
                if (s4 == 3) {
                    // every read of this unit has returned (its last operand fed the MFMAs above): re-fill its slot
                    asm volatile("" :: "v"(gcur));
                    GQD_ISSUE_NEXT(rslot)
                }
                gcur = gnxt; gnxt = gfar;
            }
            rslot = nslot;
        }
    }
    // not chained: ring and plane slots quiescent before the next round's prologue (or the end of the kernel)
    if (!chain) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    GQD_STAMP(1 + 2 * round_ix)
#undef GQD_ISSUE_NEXT
#undef GQD_ISSUE_G
#undef GQD_ISSUE_Q
    // chained: no drain.  The epilogue's stores are younger than the prefetched fills, which only makes the next round's counted waits
    // stricter than they need to be for a few stages.
    float* tile = sm->tile[wv];
#pragma unroll
    for (int t = 0; t < R; ++t)
        if (t < nvalid) gq_tile_out<7, true>(acc[t], 0.f, 0.f, &sm->rb[wv][0][32 * t], &sm->rb[wv][1][32 * t], qs, sj, scale_out, tile, Tout, ldt, unit0 + t, cunit, amax, lane, nx.unit0 != unit0);      // (= chain, from scalars that are live anyway: `chain` itself, needed this late, was kept as 0 / 1 in a VGPR, spilled to scratch, and read back behind a vmcnt(0))
    GQD_STAMP(2 + 2 * round_ix)
}

template <int NT>
__global__ __launch_bounds__(256, 1) void k_gq_d(const int8_t* __restrict__ G, int64_t ldg, int64_t units, int64_t nstage,
                                                  const int8_t* __restrict__ Qd, const double* __restrict__ qscale,
                                                  const float* __restrict__ rv, const float* __restrict__ bv,
                                                  const float* __restrict__ sv, float* __restrict__ Tout,
                                                  float* __restrict__ cpart, double* __restrict__ apart, int scale_out, int64_t ldt,
                                                  int chain_ok, int phase_mul) {
    extern __shared__ __attribute__((aligned(16))) char gqd_smem[];
    GqdSmem* sm = reinterpret_cast<GqdSmem*>(gqd_smem);
    // (launch_gq_d refuses anything else.  Told to the compiler because it hoisted the test `nstage != 1` of the fill bookkeeping out of
    //  the round loop as a 0 / 1 VGPR, spilled that register to scratch, read it back in every epilogue and put vmcnt(0) -- a drain of
    //  the chained prefetches AND of the epilogue's stores, whose acknowledgements take microseconds -- at the head of every round.)
    __builtin_assume(nstage >= 2 && (nstage & 1) == 0);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const uint32_t ph = (uint32_t)(((int64_t)(blockIdx.x & 7) * (phase_mul & 0xffff) + (int64_t)(blockIdx.x >> 3) * (phase_mul >> 16)) % nstage);   // this workgroup's first stage: XCD (b % 8) and place in the XCD (b / 8)
    const int64_t u0 = (units * (int64_t)blockIdx.x) / gridDim.x;          // this workgroup's range of 32-row units
    const int64_t u1 = (units * ((int64_t)blockIdx.x + 1)) / gridDim.x;
    float amax = 0.f;
    const float sj = sv[c];
    const double qs = qscale[c];
    asm volatile("" :: "v"(sj), "v"(qs));      // (their wait falls here, before any DMA is in flight, not in the first epilogue)
    // a round = up to 16 units, four per wave; a short last round (1..15 units) is split evenly -- (3,3,2,2) rather than (4,4,2,0) --
    // and every wave sweeps the samples with as many tiles as it has units (a wave without a unit rides along)
    auto round_at = [&](int64_t u) -> GqdRound {
        const int64_t rem = u1 - u, take = rem < 16 ? rem : 16;
        const int64_t base = take >> 2, extra = take & 3;
        const int nv = (int)(base + (wv < extra ? 1 : 0));
        GqdRound r;
        r.nv = nv; r.R = nv ? nv : 1;
        r.unit0 = nv ? u + wv * base + (wv < extra ? wv : extra) : u;
        return r;
    };
    const bool chain_all = chain_ok && nstage > kGqdSlots;    // (more than 6 units in every round, whatever its R: a round's own refills start inside it)
    uint32_t rslot = 0;
    int round_ix = 0;
    GQD_STAMP(0)
    int64_t u = u0;
    while (u < u1) {
        const int64_t un = u + (u1 - u < 16 ? u1 - u : 16);
        const GqdRound cur = round_at(u);
        const bool chain = chain_all && un < u1;
        // a round starts cold unless the round before it was chained to it -- every round but a workgroup's first, when rounds chain at
        // all.  (Recomputed from scalars every round: carried through the loop as a flag, the compiler kept it in a VGPR, spilled that
        // to scratch and drained the whole memory queue -- vmcnt(0): the chain's prefetches and the epilogue's stores -- before every
        // round to read it back.)
        const bool prologue = !chain_all || u == u0;
        GqdRound nx = cur;
        if (chain) nx = round_at(un);
#define GPCA_GQD_ROUND(RR) gqd_round<NT, RR>(G, ldg, nstage, Qd, sm, wv, lane, c, h, cur.unit0, cur.nv, prologue, chain, nx, rslot, round_ix, ph, units * 32, qs, rv, bv, sj, Tout, scale_out, ldt, cpart, amax)
        if (cur.R == 4) GPCA_GQD_ROUND(4);
        else if (cur.R == 3) GPCA_GQD_ROUND(3);
        else if (cur.R == 2) GPCA_GQD_ROUND(2);
        else GPCA_GQD_ROUND(1);
#undef GPCA_GQD_ROUND
        u = un;
        ++round_ix;
    }
#if GPCA_STAMP
    if (threadIdx.x == 0 && blockIdx.x == 0) g_gqd_stamp_n = round_ix;
#endif
    float am_lo, am_hi; halves_pair(amax, am_lo, am_hi);
    const float am = fmaxf(am_lo, am_hi);
    if (h == 0) apart[((int64_t)blockIdx.x * 4 + wv) * 32 + c] = (double)am;
}

int launch_gq_d(hipStream_t st, const int8_t* G, int64_t ldg, const GqPlan& plan, int64_t Npad, const int8_t* Qd,
                const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart, double* apart,
                int scale_out, int64_t ldt, const KernelOpts& ko) {
    // the hand-counted vmcnt pipeline is only correct for an EVEN number of 128-sample stages (9 stages gave wrong eigenvalues,
    // profiles/r1_kbench_summary.md section 9): refuse any other pitch instead of answering wrongly
    if (!dma_shape_ok(Npad, 256, plan.units * 32, 32) || ldg < Npad) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(plan.waves / 4)), blk(256);
    const int64_t nstage = Npad / 128;
    // Where a workgroup starts its sweep of the sample axis.  In lockstep every workgroup asks for the same 128-byte column of its
    // rows at the same time (131 072 rows x 128 B, one row pitch apart); with the workgroups of an XCD (b % 8) spread over 8 starting
    // stages the launch is 3-5 % faster at 10k-40k samples (scripts/kbench/kbench_gqd.hip ab, profiles/r4_kbench_summary.md).  Eight
    // starting points keep the digit planes of Q an L2 hit -- one sweep of the planes per phase in flight; 16 phases at 40k samples
    // (5 MB of planes) and any phase at 100k (12.8 MB) measured slower than lockstep, so long sample axes stay in lockstep.
    int phase = ko.gq_phase;
    if (phase < 0) phase = (nstage >= 16 && nstage <= 400) ? (int)((nstage / 8) << 16) : 0;
    hipLaunchKernelGGL((k_gq_d<1>), grid, blk, sizeof(GqdSmem), st, G, ldg, plan.units, nstage, Qd, qscale, r, b, s, Tout, cpart, apart, scale_out, ldt, ko.gq_chain, phase);      // (genotype DMAs non-temporal: every line is read once per pass; measured 1.82 -> 1.71 ms per pass)
    return 0;
}

// ================================================================================================
// K2 building blocks shared by the two DMA kernels (k_gtt_d on int8 rows, k_gtt_p on 2-bit rows): the four waves of a workgroup own
// four adjacent 128-sample blocks and the SAME row range, so they consume the same digit planes of T' -- wave w brings plane w of a
// stage into LDS, everyone reads its operands back with conflict-free ds_read_b128 -- and a block's operands are decoded (byte
// transpose / 2-bit spread) in micro-steps pinned between the MFMAs of the block before.
// ================================================================================================
template <bool PACKED>
struct GttXG { unsigned g[16]; };

// int8 rows: every 128-byte line is consumed whole by one instruction of one wave and never again in the pass -> nt
// (measured 1.90 -> 1.80 ms per pass).  Packed rows: the four waves of a workgroup share lines -> default policy
// (nt there measured 3.91 -> 4.22 ms per step).
#ifndef GPCA_GTTX_AUX
#define GPCA_GTTX_AUX 2
#endif
template <bool PACKED>
__device__ __forceinline__ void gttx_load_g(GttXG<PACKED>& b, __amdgpu_buffer_rsrc_t rg, uint32_t gvo, uint32_t row_off, uint32_t ldr) {
#pragma unroll
    for (int i = 0; i < 16; ++i) b.g[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rg, gvo, row_off + (uint32_t)i * ldr, PACKED ? 0 : GPCA_GTTX_AUX);
}
// decode micro-step m (0..15) of the next block's operands
// ABL (scripts/kbench/kbench_gtd.hip only; 0 in the product): bit0 no transpose (the dwords as loaded are the operands), bit1 the T' planes
// are fetched once, bit2 no genotype refills, bit3 no MFMA, bit4 no Ypart stores
template <bool PACKED, int ABL = 0>
__device__ __forceinline__ void gttx_decode_step(const GttXG<PACKED>& bn, Gtt2Ops& on, unsigned (&x)[4], unsigned (&y)[4], int m, unsigned bsh) {
    const int w = m >> 2, ph = m & 3;
    if (ABL & 1) { on.bt[ph][w] = (int)bn.g[4 * w + ph]; return; }
    if (PACKED) {
        if (ph == 0) x[w] = ((bn.g[4 * w] >> bsh) & 0xffu) | (((bn.g[4 * w + 1] >> bsh) & 0xffu) << 8) |
                            (((bn.g[4 * w + 2] >> bsh) & 0xffu) << 16) | ((bn.g[4 * w + 3] >> bsh) << 24);
        if (ph == 1) { on.bt[0][w] = (int)(x[w] & 0x03030303u); on.bt[1][w] = (int)((x[w] >> 2) & 0x03030303u); }
        if (ph == 2) on.bt[2][w] = (int)((x[w] >> 4) & 0x03030303u);
        if (ph == 3) on.bt[3][w] = (int)((x[w] >> 6) & 0x03030303u);
    } else {   // 4x4 byte transpose of rows 4w..4w+3 (two v_perm stages)
        const int r0 = (int)bn.g[4 * w], r1 = (int)bn.g[4 * w + 1], r2 = (int)bn.g[4 * w + 2], r3 = (int)bn.g[4 * w + 3];
        if (ph == 0) { x[w] = (unsigned)permb(r1, r0, 0x05010400u); y[w] = (unsigned)permb(r1, r0, 0x07030602u); }
        if (ph == 1) { x[w] = x[w]; }
        if (ph == 2) {
            const int y0 = permb(r3, r2, 0x05010400u);
            on.bt[0][w] = permb(y0, (int)x[w], 0x05040100u); on.bt[1][w] = permb(y0, (int)x[w], 0x07060302u);
        }
        if (ph == 3) {
            const int y1 = permb(r3, r2, 0x07030602u);
            on.bt[2][w] = permb(y1, (int)y[w], 0x05040100u); on.bt[3][w] = permb(y1, (int)y[w], 0x07060302u);
        }
    }
}
template <bool PACKED>
__device__ __forceinline__ void gttx_decode(const GttXG<PACKED>& b, Gtt2Ops& o, unsigned bsh) {
    unsigned x[4], y[4];
#pragma unroll
    for (int m = 0; m < 16; ++m) gttx_decode_step<PACKED>(b, o, x, y, m, bsh);
}
struct GttXT { i32x4 t[kDigits]; };
// the 16 MFMAs of the current block; after each one a micro-step of the next block's decode, and (first four slots) the
// LDS reads of the next block's digit operands
// ZERO: the first block of a task -- the sums start from the MFMA's constant-zero C operand instead of from zeroed registers, so that
// nothing of the accumulators is live across a task boundary of the chained k_gtt_d (hipcc otherwise carried the 256 registers through
// the task loop's phi nodes, shuffled them between AGPRs at every boundary and sent four of them to scratch)
template <bool PACKED, int ND = kDigits, int ABL = 0, bool ZERO = false>
__device__ __forceinline__ void gttx_phase(const GttXT& tc, const Gtt2Ops& oc, i32x16 (&acc)[4][kDigits],
                                           const GttXG<PACKED>& gn, Gtt2Ops& on, GttXT& tn, const i32x4* lds_next, unsigned bsh) {
    unsigned x[4], y[4];
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (ABL & 8) { if (d < ND) { if (ZERO) acc[t][d] = zero16; acc[t][d][0] ^= tc.t[d][t] ^ oc.bt[t][d]; } }
            else if (d < ND) acc[t][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(tc.t[d], oc.bt[t], ZERO ? zero16 : acc[t][d], 0, 0, 0);   // (three-plane mode: the last four slots carry only decode work)
            const int m = d * 4 + t;
            if (m < ND) tn.t[m] = lds_next[m * 64];
            gttx_decode_step<PACKED, ABL>(gn, on, x, y, m, bsh);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ================================================================================================
// K2 by LDS-DMA (int8-resident genotypes).  k_gtt_x keeps four 32-row blocks of dwords per wave in registers (64 VGPRs)
// plus 32 VGPRs of digit planes on their way to LDS, fills the 256-VGPR budget and so has ~12 KiB per wave in flight:
// latency-bound at ~5.5 TB/s while the same request pattern alone streams at 6.9.  Here both streams arrive by
// `buffer_load_dwordx4 ... lds` (no VGPR destination): a ring of 6 row-major [32 rows][128 B] units per wave (20 KiB in
// flight) and wave w's plane of T' for the next stage; the dwords of a block are read back from LDS two phases before
// they are multiplied (16 ds_read_b32, rows contiguous across lanes: conflict-free) and transposed by the same v_perm
// micro-steps in the shadow of the MFMAs.
//   phase b:  [vmcnt] read block b+2 from LDS | 16 MFMAs of block b || T' planes + decode of block b+1 | re-fill the slot
//             of block b+1 (all of its reads have been consumed) with block b+7.
//   before the last phase of stage s (4 phases): vmcnt + workgroup barrier -> T'(s+1) visible, slot of T'(s) free:
//             issue T'(s+2).
// Hand-counted completion (the DMAs are invisible to the compiler).  Issue order in steady state:
//   ... T'(s+2), G(4s+10), G(4s+11), G(4s+12), G(4s+13), T'(s+3), ...      (4 instructions each)
//   - barrier of stage s: T'(s+1) was issued one stage earlier, 4 unit fills are younger           -> vmcnt(16)
//   - start of phase b: G(b+2) must have landed; younger: G(b+3..b+6) and one plane batch, two when b = 3 mod 4
//                                                                                                 -> vmcnt(20) / (24)
// The prologue issues G0 G1 T'0 G2 G3 G4 G5 T'1, loads blocks 0 and 1 into registers, then G6: the same counts hold.
// ================================================================================================
template <int NT>
__device__ __forceinline__ void gtd_issue_g(const uint8_t* gp, int64_t ldr, int64_t blk, int64_t kblocks, uint32_t lds_slot, uint32_t gvo) {
    const int64_t ub = blk < kblocks ? blk : 0;
    const i32x4 rg = gqd_rsrc(gp + ub * 32 * ldr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot's last LDS reads have returned
#pragma unroll
    for (int i = 0; i < 4; ++i) gqd_dma<NT>(lds_slot + 1024u * i, gvo, rg, 8u * i * (uint32_t)ldr);
}
__device__ __forceinline__ void gtd_issue_t(const int8_t* tp, int64_t st, int64_t nstage, uint32_t lds_t_slot, uint32_t tvo) {
    constexpr uint32_t TKB = kDigits * 1024;
    const int64_t us = st < nstage ? st : 0;
    const i32x4 rt = gqd_rsrc(tp + us * 4 * TKB);
#pragma unroll
    for (int j = 0; j < 4; ++j) gqd_dma<0>(lds_t_slot + (uint32_t)(j * kDigits) * 1024u, tvo, rt, j * TKB);
}
__device__ __forceinline__ void gtd_read_g(GttXG<false>& b, const char* unit) {
#pragma unroll
    for (int i = 0; i < 16; ++i) b.g[i] = *reinterpret_cast<const unsigned*>(unit + i * 128);
}

#if GPCA_STAMP
__device__ unsigned long long g_gtd_stamp[2 * 4096];      // s_memrealtime at the start and at the end of every workgroup of k_gtt_d
#endif
// this workgroup's virtual id: workgroup b runs on XCD b % 8; the workgroups of one XCD get consecutive virtual ids, so that the
// workgroups that share a row chunk (and its T' planes) run behind one L2
__device__ __forceinline__ int64_t k2_virtual_wg(int xcd_remap) {
    int64_t vb = blockIdx.x;
    if (xcd_remap) {
        const int64_t q = gridDim.x / 8, r = gridDim.x % 8, x = blockIdx.x % 8;
        vb = x * q + (x < r ? x : r) + blockIdx.x / 8;
    }
    return vb;
}

// A workgroup's tasks run as ONE pipeline (chained, like k_gq_d's rounds).  The blocks of task i + 1 follow the blocks of task i in the
// ring -- the last refills of a task used to wrap to its own block 0 and were thrown away, now they fetch the next task's first
// blocks -- and the T' planes two stages ahead are the next task's stages 0 and 1, so the issue order across the boundary is the steady
// state's and the registers (the dwords of blocks b + 1 and b + 2, the operands of block b) already hold the next task's first blocks
// when a task ends.  At the boundary only the accumulators change hands: the finished task's tile goes to Ypart (stores younger than
// every load in flight: the counted waits only get stricter) and the sums restart.  A boundary used to cost a chip-wide drain and a
// cold prologue (every workgroup reaches it at the same time): ~25 us per task in the decomposition A/B of kbench_gtd.
struct GtdTask { const uint8_t* gp; const int8_t* tp; int64_t kblocks; int64_t n0; int64_t slice; bool live; };

// a wave's four 32 x 32 tiles of exact integer sums -> Ypart[slice][n0 + 4 c + t][j] as f64.  D[j][col]: j = (reg&3) + 8*(reg>>2) + 4*h,
// col = c -> sample n0 + 4c + t.  The lane id is made opaque so that the address arithmetic is done here, per task, instead of being
// hoisted out of the task loop into registers that the stage loop's pressure sends to scratch (their reloads carried vmcnt(0) waits
// that drained the DMA queue at every task boundary).
template <int BITS>
__device__ __forceinline__ void gtt_tiles_out(const i32x16 (&acc)[4][kDigits], double* __restrict__ Ypart, int64_t slice, int64_t Npad, int64_t n0,
                                              char* stage = nullptr, int piece_stride = 1024) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned all = ~0u;
    asm volatile("" : "+s"(all));                     // (opaque mask: the lane id is recomputed HERE -- hoisted out of the task loop it would be kept across the stage loop, i.e. spilled)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
    const int c = lane & 31, h = lane >> 5;
#ifndef GPCA_YPART_DIRECT
    if (stage) {
        // The wave's tile of Y^T is one contiguous 32 KiB of Ypart (128 samples x 32 doubles), but a lane holds 16-byte pieces of it 1 KiB
        // apart: written directly, each store instruction touches 64 lines with 16 bytes each (2 048 partial-line writes per tile;
        // the stores were worth 2.3 % of a launch for 1.3 % of its bytes).  Through a wave-private 4 KiB of LDS a store instruction
        // writes 8 full 128-byte lines: per (sample t of the lane's four, half of the row) lane (c, h) parks its 8 doubles in line c
        // (16-byte chunks XOR-swizzled by c; the 4 KiB are four 1-KiB pieces `piece_stride` bytes apart: contiguous in k_gq_d's
        // tile buffer, the wave's own DMA pieces of a drained stage buffer in k_gtt_p), lane L takes chunk L & 7 of lines L >> 3
        // (+ 8, 16, 24) back and stores it -- as a
        // STREAMING store: the same full lines written with plain stores measured no gain at all (1.5245 against 1.5246 ms direct);
        // what costs is the dirty tile evicted from L2 between the genotype reads, as in K1.  1M x 10k: 1.525 -> 1.498 ms (-1.8 %),
        // 125k x 100k: 1.937 -> 1.869 (-3.5 %) (profiles/r4_kbench_summary.md section 8); the fold that reads the tiles next now
        // reads them from HBM (22 -> 30 us at configs[1], part of it won back there).
        double* ybase = Ypart + (slice * Npad + n0) * 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int ip = 0; ip < 2; ++ip) {
                        const int e = 8 * half + 4 * q + 2 * ip;
                        double2 o;
                        o.x = combine_digits<BITS>(acc[t], e); o.y = combine_digits<BITS>(acc[t], e + 1);
                        const int ch = 4 * q + 2 * h + ip;
                        *reinterpret_cast<double2*>(stage + (c >> 3) * piece_stride + (c & 7) * 128 + ((ch ^ (c & 7)) << 4)) = o;
                    }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (one wave's LDS operations execute in order: no barrier)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int line = (lane >> 3) + 8 * i, ch = lane & 7;
                    const double2 v = *reinterpret_cast<const double2*>(stage + i * piece_stride + (lane >> 3) * 128 + ((ch ^ (line & 7)) << 4));
                    typedef double d2v __attribute__((ext_vector_type(2)));
                    const d2v vv = {v.x, v.y};
                    __builtin_nontemporal_store(vv, reinterpret_cast<d2v*>(ybase + (4 * line + t) * 32 + 16 * half + 2 * ch));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_sched_barrier(0);                            // (a quarter of a tile's accumulators live at a time)
            }
        }
        return;
    }
#endif
    double* yp = Ypart + (slice * Npad + n0 + 4 * c) * 32 + 4 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int j = (e & 3) + 8 * (e >> 2);
            double2 o;
            o.x = combine_digits<BITS>(acc[t], e); o.y = combine_digits<BITS>(acc[t], e + 1);
            *reinterpret_cast<double2*>(yp + t * 32 + j) = o;
            if (e == 6) __builtin_amdgcn_sched_barrier(0);       // (half a tile's accumulators live at a time)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the walk over a workgroup's tasks: task = (row chunk, n-group), n-group fastest; workgroup v takes tasks v, v + grid, v + 2 grid ...
// (strided): at any time the workgroups of an XCD (consecutive v) are on the n-groups of the same one or two row chunks and fetch those
// rows' T' planes from HBM once (consecutive tasks per workgroup spread an XCD over 6 row chunks: 15 MB of planes against 4 MB of L2,
// + 4.6 % at configs[1]).  Row chunk w = stages [w C, min(S, (w + 1) C)).  Stepping from a task to the next is additions only: a
// division inside the pipeline costs VALU registers the stage loop does not have.
struct K2Walk {
    int64_t wch, g, wch_step, g_step, ngroups;
    int ntask;
    __device__ __forceinline__ void init(int64_t vwg, int64_t T, int64_t ngroups_, int tasks_per_wg, int strided) {
        ngroups = ngroups_;
        const int64_t t_first = strided ? vwg : vwg * tasks_per_wg, t_step = strided ? (int64_t)gridDim.x : 1;
        ntask = 0;
        if (t_first < T) { const int64_t left = (T - 1 - t_first) / t_step + 1; ntask = (int)(left < tasks_per_wg ? left : tasks_per_wg); }
        wch = t_first / ngroups; g = t_first - wch * ngroups;
        wch_step = t_step / ngroups; g_step = t_step - wch_step * ngroups;
    }
    __device__ __forceinline__ void advance() { wch += wch_step; g += g_step; if (g >= ngroups) { g -= ngroups; ++wch; } }
};

template <int NT, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_gtt_d(const uint8_t* __restrict__ Gb, int64_t ldr, int64_t Npad,
                                                   const int8_t* __restrict__ Td, double* __restrict__ Ypart,
                                                   int64_t S, int64_t C, int64_t ngroups, int W, int tasks_per_wg, int strided, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) char gqd_smem[];
    GqdSmem* sm = reinterpret_cast<GqdSmem*>(gqd_smem);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
#if GPCA_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_gtd_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr uint32_t TKB = kDigits * 1024;
    K2Walk cw;
    cw.init(k2_virtual_wg(xcd_remap), (int64_t)W * ngroups, ngroups, tasks_per_wg, strided);
    const int ntask = cw.ntask;
    if (ntask == 0) return;
    auto task_of = [&](const K2Walk& w) -> GtdTask {
        const int64_t s0 = w.wch * C, s1 = (s0 + C < S) ? s0 + C : S;
        GtdTask k;
        k.n0 = (w.g * 4 + wv) * 128;
        k.live = k.n0 < Npad;             // a dead wave (ragged last n-group) still moves planes and joins the barriers
        if (!k.live) k.n0 = 0;
        k.gp = Gb + (s0 * 128) * ldr + k.n0;
        k.tp = Td + (s0 * 4) * TKB + wv * 1024;                 // this wave's plane
        k.kblocks = (s1 - s0) * 4;
        k.slice = w.wch;
        return k;
    };

    const uint32_t gvo = (uint32_t)(lane >> 3) * (uint32_t)ldr + 16u * (uint32_t)(lane & 7);
    const uint32_t tvo = (uint32_t)(lane * 16);
    const uint32_t lds_t = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sm->q[0][0][0][0] + (uint32_t)wv * 1024u;
    const uint32_t lds_g = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sm->g[0][0][0] + (uint32_t)wv * (kGqdSlots * 4096u);
    const char* gl = reinterpret_cast<const char*>(&sm->g[wv][0][0]) + (16 * h) * 128 + 4 * c;   // this lane's dword column
    i32x4 (*tds)[4][kDigits][64] = sm->q;

    // issue cursors: the next block / the next stage of T' planes to fetch, walking the workgroup's tasks in order (past the last task
    // they re-read its first blocks, which are never used)
    K2Walk gw = cw, tw = cw;
    int gi = 0; GtdTask gk = task_of(gw); int64_t gblk = 0;
    int ti = 0; GtdTask tk = gk; int64_t tst = 0;
    auto issue_g = [&](uint32_t slot) {
        const i32x4 rg = gqd_rsrc(gk.gp + gblk * 32 * ldr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot's last LDS reads have returned
#pragma unroll
        for (int i = 0; i < 4; ++i) gqd_dma<NT>(lds_g + slot * 4096u + 1024u * i, gvo, rg, 8u * i * (uint32_t)ldr);
        if (++gblk == gk.kblocks) { gblk = 0; if (gi + 1 < ntask) { ++gi; gw.advance(); gk = task_of(gw); } }
    };
    auto issue_t = [&](uint32_t tslot) {
        const i32x4 rt = gqd_rsrc(tk.tp + tst * 4 * TKB);
#pragma unroll
        for (int j = 0; j < 4; ++j) gqd_dma<0>(lds_t + tslot * 4u * TKB + (uint32_t)(j * kDigits) * 1024u, tvo, rt, j * TKB);
        if (++tst == (tk.kblocks >> 2)) { tst = 0; if (ti + 1 < ntask) { ++ti; tw.advance(); tk = task_of(tw); } }
    };

    i32x16 acc[4][kDigits];
    GttXG<false> GA, GB;
    Gtt2Ops OA, OB;
    GttXT TA, TB;
    // prologue (once per workgroup): G0 G1 T'0 G2 G3 G4 G5 T'1, blocks 0 and 1 into registers, then G6: the steady-state counts hold
    issue_g(0); issue_g(1);
    issue_t(0);
    issue_g(2); issue_g(3); issue_g(4); issue_g(5);
    issue_t(1);
    asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");      // G0, G1 and every wave's plane of T'(0) landed
    gtd_read_g(GA, gl);
    gtd_read_g(GB, gl + 4096);
#pragma unroll
    for (int d = 0; d < kDigits; ++d) TA.t[d] = tds[0][0][d][lane];
    gttx_decode<false>(GA, OA, 0u);
    asm volatile("" :: "v"(GB.g[15]));
    issue_g(0);
    uint32_t s1 = 1;            // ring slot of block b + 1 (re-filled at the end of phase b)
    uint32_t slot = 0;          // plane slot of the current stage (alternates over the workgroup's stages, across tasks)

    // one phase: TC/OC = operands of block b; GN (registers of block b+1) -> ON, TN; GR receives block b+2
#define GTD_PHASE(TC, OC, GN, ON, TN, GR, WV, TSLOT, TBLK, ZERO)                                           \
    {                                                                                                    \
        const uint32_t s2_ = s1 == kGqdSlots - 1 ? 0u : s1 + 1u;                                         \
        gqd_wait_vm<WV>();                                                                               \
        gtd_read_g(GR, gl + s2_ * 4096u);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        gttx_phase<false, kDigits, ABL, ZERO>(TC, OC, acc, GN, ON, TN, &tds[(TSLOT)][(TBLK)][0][lane], 0u); \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (!(ABL & 4)) issue_g(s1);                                                                     \
        s1 = s2_;                                                                                        \
    }
#define GTD_STAGE(ZERO)                                                                                  \
    {                                                                                                    \
        GTD_PHASE(TA, OA, GB, OB, TB, GA, 20, slot, 1, ZERO)                                             \
        GTD_PHASE(TB, OB, GA, OA, TA, GB, 20, slot, 2, false)                                            \
        GTD_PHASE(TA, OA, GB, OB, TB, GA, 20, slot, 3, false)                                            \
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");   /* T' of the next stage visible; everyone is done with this stage's */ \
        if (!(ABL & 2)) issue_t(slot);                                                                   \
        GTD_PHASE(TB, OB, GA, OA, TA, GB, 24, slot ^ 1u, 0, false)                                       \
        slot ^= 1u;                                                                                      \
    }
    for (int i = 0; i < ntask; ++i) {
        if (i) cw.advance();
        const GtdTask k = task_of(cw);
        const int64_t nstage = k.kblocks >> 2;
        GTD_STAGE(true)                                        // the task's first block starts the sums (C = 0)
        for (int64_t st = 1; st < nstage; ++st) GTD_STAGE(false)
        if (k.live && !((ABL & 16) && acc[0][0][0] != 0x7fffffff)) gtt_tiles_out<7>(acc, Ypart, k.slice, Npad, k.n0, reinterpret_cast<char*>(&sm->tile[wv][0]));
    }
#undef GTD_STAGE
#undef GTD_PHASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may land after this workgroup's LDS is released
#if GPCA_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_gtd_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ================================================================================================
// K2 for packed (2-bit) genotypes by cooperative LDS-DMA.  k_gtt_x<true> keeps 4 blocks of dwords per wave in
// registers = 16 KiB per CU in flight: at 2.8 TB/s of real traffic that is exactly one loaded HBM latency, i.e. the
// kernel is latency-bound (matrix cores 46 % busy).  Here a stage (128 SNP rows) is brought in by the whole workgroup:
// wave w DMAs rows 8w..8w+7 of each 32-row block as one full-line piece (8 rows x 128 B = the 512 samples of the
// n-group) and plane w of T' -- 8 instructions per wave and stage -- into a ring of four 32-KiB stage buffers, two
// stages (64 KiB per CU) in flight beyond the one being consumed and the one the register pipeline looks ahead into.
// Every wave then reads its own 32-byte slab of each row from LDS (dword reads, 4 lanes per dword) and decodes as
// before.  Per stage:  vmcnt(8) [stages s, s+1 landed; 8 DMAs of s+2 may be outstanding]; barrier; issue stage s+3.
// ================================================================================================
struct GtpStage { i32x4 t[4][kDigits][64]; char g[4][4096]; };     // 16 KiB of planes + 16 KiB of genotype rows
struct GtpSmem { GtpStage stg[4]; };

template <int ND>
__device__ __forceinline__ void gtp_segment(const uint8_t* __restrict__ G2, int64_t ld2, int64_t Npad, const int8_t* __restrict__ Td,
                                            double* __restrict__ Ypart, GtpSmem* sm, int wv, int lane, int c, int h,
                                            int64_t g, int64_t s0, int64_t nstage, int64_t slice) {
    const int64_t n0 = (g * 4 + wv) * 128;               // Npad is a multiple of 1024 in packed mode: every wave is live
    const int64_t m_begin = s0 * 128;

    i32x16 acc[4][kDigits];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < kDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][d][e] = 0;

    constexpr uint32_t TKB = kDigits * 1024;
    const uint8_t* gp = G2 + m_begin * ld2 + g * 128;              // the n-group's 128-byte column of the row range
    const int8_t* tp = Td + (m_begin >> 5) * TKB + wv * 1024;      // this wave's plane
    const uint32_t gvo = (uint32_t)(8 * wv + (lane >> 3)) * (uint32_t)ld2 + 16u * (uint32_t)(lane & 7);
    const uint32_t tvo = (uint32_t)(lane * 16);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sm->stg[0];
    const unsigned bsh = 8u * (unsigned)(c & 3);
    // this lane's dword of row 16 h + i of a block: + i * 128
    const uint32_t rdo = (uint32_t)((16 * h) * 128 + 32 * wv + (c & ~3));

    // stage st -> ring slot: 4 genotype pieces (one per 32-row block) + 4 plane pieces
    auto issue_stage = [&](int64_t st, uint32_t slot) {
        const int64_t us = st < nstage ? st : 0;
        const i32x4 rg = gqd_rsrc(gp + us * 128 * ld2);
        const i32x4 rt = gqd_rsrc(tp + us * 4 * TKB);
        const uint32_t base = lds0 + slot * (uint32_t)sizeof(GtpStage);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
            gqd_dma<0>(base + 4u * TKB + (uint32_t)bb * 4096u + (uint32_t)wv * 1024u, gvo, rg, (uint32_t)(32 * bb) * (uint32_t)ld2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            gqd_dma<0>(base + (uint32_t)(j * kDigits) * 1024u + (uint32_t)wv * 1024u, tvo, rt, j * TKB);
    };
    auto read_g = [&](GttXG<true>& b, uint32_t slot, int bb) {
        const char* unit = &sm->stg[slot].g[bb][0] + rdo;
#pragma unroll
        for (int i = 0; i < 16; ++i) b.g[i] = *reinterpret_cast<const unsigned*>(unit + i * 128);
    };

    GttXG<true> GA, GB;
    Gtt2Ops OA, OB;
    GttXT TA, TB;
    issue_stage(0, 0); issue_stage(1, 1); issue_stage(2, 2);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");       // stages 0 and 1 landed for every wave
    issue_stage(3, 3);
    read_g(GA, 0, 0);
    read_g(GB, 0, 1);
#pragma unroll
    for (int d = 0; d < ND; ++d) TA.t[d] = sm->stg[0].t[0][d][lane];
    gttx_decode<true>(GA, OA, bsh);

    // one phase: TC/OC = operands of block b; GN (registers of block b+1) -> ON, TN (planes of block b+1 at TSLOT/TBLK);
    // GR receives block b+2 = block RB of stage slot RS
#define GTP_PHASE(TC, OC, GN, ON, TN, GR, TSLOT, TBLK, RS, RB)                                             \
    {                                                                                                    \
        read_g(GR, (RS), (RB));                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        gttx_phase<true, ND>(TC, OC, acc, GN, ON, TN, &sm->stg[(TSLOT)].t[(TBLK)][0][lane], bsh);            \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }
    for (int64_t st = 0; st < nstage; ++st) {
        const uint32_t q0 = (uint32_t)(st & 3), qn = (uint32_t)((st + 1) & 3);
        if (st > 0) {
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage st+1 landed; everyone is done with stage st-1
            issue_stage(st + 3, (uint32_t)((st + 3) & 3));
        }
        GTP_PHASE(TA, OA, GB, OB, TB, GA, q0, 1, q0, 2)
        GTP_PHASE(TB, OB, GA, OA, TA, GB, q0, 2, q0, 3)
        GTP_PHASE(TA, OA, GB, OB, TB, GA, q0, 3, qn, 0)
        GTP_PHASE(TB, OB, GA, OA, TA, GB, qn, 0, qn, 1)
    }
#undef GTP_PHASE
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");       // no DMA may land after this segment; every wave has left the ring
    // the ring is drained and every wave has left it: the wave's own DMA pieces of stage buffer 3 (1 KiB in each of its four blocks of
    // genotype rows, written by nobody else -- a faster wave may already be filling the ring for the next task) stage the tile on its
    // way out (gtt_tiles_out: full lines, streaming stores)
    gtt_tiles_out<ND == 3 ? 8 : 7>(acc, Ypart, slice, Npad, n0, &sm->stg[3].g[0][1024 * wv], (int)sizeof(sm->stg[3].g[0]));
}

template <int ND>
__global__ __launch_bounds__(256, 1) void k_gtt_p(const uint8_t* __restrict__ G2, int64_t ld2, int64_t Npad,
                                                   const int8_t* __restrict__ Td, double* __restrict__ Ypart,
                                                   int64_t S, int64_t C, int64_t ngroups, int W, int tasks_per_wg, int strided, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) char gqd_smem[];
    GtpSmem* sm = reinterpret_cast<GtpSmem*>(gqd_smem);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    K2Walk cw;
    cw.init(k2_virtual_wg(xcd_remap), (int64_t)W * ngroups, ngroups, tasks_per_wg, strided);     // (see k_gtt_d)
    for (int i = 0; i < cw.ntask; ++i) {
        if (i) cw.advance();
        const int64_t s0 = cw.wch * C, s1 = (s0 + C < S) ? s0 + C : S;
        gtp_segment<ND>(G2, ld2, Npad, Td, Ypart, sm, wv, lane, c, h, cw.g, s0, s1 - s0, cw.wch);       // (ends with a workgroup barrier)
    }
}

int launch_gtt_p(hipStream_t st, const uint8_t* G2, int64_t ld2, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Ypart,
                 const Gtt8Plan& plan, int nd, const KernelOpts& ko) {
    // stages of 128 SNP rows, 1 024-sample row padding of the packed store (its DMA pieces are whole 128-byte lines of codes)
    if (plan.tasks_per_wg < 1 || plan.S * 128 != Mpad || !dma_shape_ok(Npad, kSamplePad2bit, Mpad, kGQRowsPerWave) || ld2 * 4 < Npad) return (int)hipErrorInvalidValue;
    const int remap = ko.gtt_xcd;
    if (nd == 3) hipLaunchKernelGGL(k_gtt_p<3>, dim3((unsigned)plan.grid), dim3(256), sizeof(GtpSmem), st, G2, ld2, Npad, Td, Ypart, plan.S, plan.C, plan.ngroups, plan.W, plan.tasks_per_wg, plan.strided, remap);
    else hipLaunchKernelGGL(k_gtt_p<kDigits>, dim3((unsigned)plan.grid), dim3(256), sizeof(GtpSmem), st, G2, ld2, Npad, Td, Ypart, plan.S, plan.C, plan.ngroups, plan.W, plan.tasks_per_wg, plan.strided, remap);
    return 0;
}

int launch_gtt_d(hipStream_t st, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Ypart,
                 const Gtt8Plan& plan, const KernelOpts& ko) {
    if (plan.tasks_per_wg < 1 || plan.S * 128 != Mpad || !dma_shape_ok(Npad, 256, Mpad, kGQRowsPerWave) || ldg < Npad) return (int)hipErrorInvalidValue;   // 128-row stages, 256-sample pitch
    const dim3 grid((unsigned)plan.grid), blk(256);
    const int remap = ko.gtt_xcd;   // (1 in the product; the kbench harness turns it off: + 5-15 % without it)
    hipLaunchKernelGGL((k_gtt_d<1>), grid, blk, sizeof(GqdSmem), st, (const uint8_t*)G, ldg, Npad, Td, Ypart, plan.S, plan.C, plan.ngroups, plan.W, plan.tasks_per_wg, plan.strided, remap);
    return 0;
}

// The LDS-DMA kernels use more than 64 KiB of dynamic LDS: the opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is recorded
// per device by the runtime, so gpca_create calls this once per handle after hipSetDevice (a process may open handles on
// several GPUs).  Returns a hipError_t value (0 = ok).
int init_device_kernels_i8() {
    int e = 0;
#define GPCA_OPT_IN(KERNEL, BYTES) \
    if (e == 0) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));
    GPCA_OPT_IN((k_gq_d<1>), sizeof(GqdSmem))
    GPCA_OPT_IN((k_gtt_p<kDigits>), sizeof(GtpSmem)) GPCA_OPT_IN((k_gtt_p<3>), sizeof(GtpSmem))
    GPCA_OPT_IN((k_gtt_d<1>), sizeof(GqdSmem))
#undef GPCA_OPT_IN
    return e;
}

}  // namespace gpca
