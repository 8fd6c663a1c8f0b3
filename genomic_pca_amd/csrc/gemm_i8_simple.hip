// The register-only kernels of the exact-integer path: every operand travels HBM -> registers through compiler-visible loads, no LDS,
// no hand-counted waits.  k_gtt_i8<.., NARROW> is the default K2 for matrices of at most 256 samples; the others (k_gq_i8, k_gtt_i8,
// k_gtt_2bit) are the REFERENCE kernels behind gpca_config.reserved[0] & GPCA_CFG_SIMPLE_KERNELS: same integers, same pinned f32
// roundings, same per-unit c partials as the DMA kernels of gemm_i8.hip, which the parity tests hold to their bits.
#include "gemm_i8_common.h"

namespace gpca {

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
// G is loaded a 128-sample super-chunk at a time: the four 32-byte pieces of a row's 128-byte line are requested
// back to back (one L1 miss + three hits) instead of one per compute phase (four L2->L1 line fills, which made
// the L2->L1 path, not HBM, the limit: 3.7 TB/s).  Q digit planes (L2-resident, full-line reads) ride a 4-stage ring.
template <int R>
struct Gq8G { i32x4 g[4][R]; };

template <int R, int AUX>
__device__ __forceinline__ void gq8_load_g(Gq8G<R>& b, __amdgpu_buffer_rsrc_t rg, const uint32_t (&gvo)[R], uint32_t s0) {
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) b.g[j][t] = __builtin_amdgcn_raw_buffer_load_b128(rg, gvo[t], s0 + 32u * j, AUX);
}
template <int R>
__device__ __forceinline__ void gq8_compute(const i32x4 (&g)[R], const Gq8Q& q, i32x16 (&acc)[R][kDigits]) {
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
#pragma unroll
        for (int t = 0; t < R; ++t) acc[t][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(g[t], q.q[d], acc[t][d], 0, 0, 0);
}

template <int R, int AUX>
__device__ __forceinline__ void gq8_group(const int8_t* __restrict__ G, int64_t ldg, int64_t nsuper,
                                          const int8_t* __restrict__ Qd, double qs, const float* __restrict__ rv,
                                          const float* __restrict__ bv, float sj, float* __restrict__ Tout, int scale_out, int64_t ldt,
                                          float* __restrict__ cunit, int64_t row0, int c, int h, int lane) {
    const __amdgpu_buffer_rsrc_t rg = make_rsrc8(G + row0 * ldg);
    uint32_t gvo[R];
#pragma unroll
    for (int t = 0; t < R; ++t) gvo[t] = (uint32_t)((32 * t + c) * ldg + 16 * h);
    const uint32_t qvo = (uint32_t)(lane * 16);
    constexpr uint32_t QCH = kDigits * 1024;   // bytes of digit planes per 32-sample chunk

    i32x16 acc[R][kDigits];
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int d = 0; d < kDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][d][e] = 0;

    // nsuper (128-sample super-chunks) is even: samples are padded to a multiple of 256
    Gq8G<R> GA, GB;
    Gq8Q Q0, Q1, Q2, Q3;
    {
        const __amdgpu_buffer_rsrc_t rq0 = make_rsrc8(Qd);
        gq8_load_g<R, AUX>(GA, rg, gvo, 0u);
        gq8_load_q(Q0, rq0, qvo, 0u); gq8_load_q(Q1, rq0, qvo, QCH); gq8_load_q(Q2, rq0, qvo, 2 * QCH);
    }
#define GQ8_PHASE(GCUR, J, QCUR, QNEXT, QNEXT_OFF)                         \
    gq8_load_q(QNEXT, rq, qvo, (QNEXT_OFF));                                \
    __builtin_amdgcn_sched_barrier(0);                                      \
    gq8_compute<R>(GCUR.g[J], QCUR, acc);                                   \
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t sc = 0; sc < nsuper; sc += 2) {
        const __amdgpu_buffer_rsrc_t rq = make_rsrc8(Qd + sc * 4 * QCH);
        const uint32_t s0 = (uint32_t)(sc * 128);
        const uint32_t more = (sc + 2 < nsuper) ? 1u : 0u;   // the last trip re-loads its own data (unused)
        gq8_load_g<R, AUX>(GB, rg, gvo, s0 + 128u);
        GQ8_PHASE(GA, 0, Q0, Q3, 3 * QCH)
        GQ8_PHASE(GA, 1, Q1, Q0, 4 * QCH)
        GQ8_PHASE(GA, 2, Q2, Q1, 5 * QCH)
        GQ8_PHASE(GA, 3, Q3, Q2, 6 * QCH)
        gq8_load_g<R, AUX>(GA, rg, gvo, s0 + 256u * more);
        GQ8_PHASE(GB, 0, Q0, Q3, 7 * QCH)
        GQ8_PHASE(GB, 1, Q1, Q0, 8 * QCH * more)
        GQ8_PHASE(GB, 2, Q2, Q1, 8 * QCH * more + QCH)
        GQ8_PHASE(GB, 3, Q3, Q2, 8 * QCH * more + 2 * QCH)
    }
#undef GQ8_PHASE
#pragma unroll
    for (int t = 0; t < R; ++t) {
        float ct = 0.f;     // this tile's share of c = b^T T: one partial per 32-row unit, so c does not depend on the grid partition
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = row0 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float ri = rv[row], bi = bv[row];
            const float gq = (float)(combine_digits(acc[t], e) * qs);
            const float tv = __fmaf_rn(ri, gq, __fmul_rn(bi, sj));   // roundings pinned: every K1 variant returns the same bits
            ct = __fmaf_rn(bi, tv, ct);
            Tout[row * ldt + c] = scale_out ? __fmul_rn(ri, tv) : tv;
        }
        GPCA_STORE_CUNIT(row0 / 32 + t)
    }
}

template <int AUX>
__global__ __launch_bounds__(256, 1) void k_gq_i8(const int8_t* __restrict__ G, int64_t ldg, int64_t units, int64_t nsuper,
                                                   const int8_t* __restrict__ Qd, const double* __restrict__ qscale,
                                                   const float* __restrict__ rv, const float* __restrict__ bv,
                                                   const float* __restrict__ sv, float* __restrict__ Tout,
                                                   float* __restrict__ cpart, int scale_out, int64_t ldt) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t waves = (int64_t)gridDim.x * 4;
    int64_t u = (units * wave) / waves;
    const int64_t u_end = (units * (wave + 1)) / waves;
    const float sj = sv[c];
    const double qs = qscale[c];
    for (; u + 4 <= u_end; u += 4) gq8_group<4, AUX>(G, ldg, nsuper, Qd, qs, rv, bv, sj, Tout, scale_out, ldt, cpart, u * 32, c, h, lane);
    if (u + 2 <= u_end) { gq8_group<2, AUX>(G, ldg, nsuper, Qd, qs, rv, bv, sj, Tout, scale_out, ldt, cpart, u * 32, c, h, lane); u += 2; }
    if (u + 1 <= u_end) { gq8_group<1, AUX>(G, ldg, nsuper, Qd, qs, rv, bv, sj, Tout, scale_out, ldt, cpart, u * 32, c, h, lane); u += 1; }
}

void launch_gq_i8(hipStream_t st, const int8_t* G, int64_t ldg, const GqPlan& plan, int64_t N, const int8_t* Qd,
                  const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart,
                  int scale_out, int64_t ldt, const KernelOpts& ko) {
    const dim3 grid((unsigned)(plan.waves / 4)), blk(256);
    const int64_t nsuper = (N + 255) / 256 * 2;   // 128-sample super-chunks, even count (= Npad / 128)
    hipLaunchKernelGGL((k_gq_i8<0>), grid, blk, 0, st, G, ldg, plan.units, nsuper, Qd, qscale, r, b, s, Tout, cpart, scale_out, ldt);
}

struct Gtt8Buf { int g[16]; i32x4 t[kDigits]; };

template <int AUX>
__device__ __forceinline__ void gtt8_load(Gtt8Buf& b, __amdgpu_buffer_rsrc_t rg, uint32_t gvo, uint32_t row_off, uint32_t ldg,
                                          __amdgpu_buffer_rsrc_t rt, uint32_t tvo, uint32_t toff) {
#pragma unroll
    for (int i = 0; i < 16; ++i) b.g[i] = __builtin_amdgcn_raw_buffer_load_b32(rg, gvo, row_off + (uint32_t)i * ldg, AUX);
#pragma unroll
    for (int d = 0; d < kDigits; ++d) b.t[d] = __builtin_amdgcn_raw_buffer_load_b128(rt, tvo, toff + d * 1024, 0);
}


__device__ __forceinline__ void gtt8_compute(const Gtt8Buf& b, i32x16 (&acc)[4][kDigits]) {
    // 16 rows x 4 samples of bytes -> 4 operands of 16 k-contiguous bytes (operand t = sample byte t of rows 0..15)
    i32x4 bt[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int r0 = b.g[4 * w], r1 = b.g[4 * w + 1], r2 = b.g[4 * w + 2], r3 = b.g[4 * w + 3];
        const int x0 = permb(r1, r0, 0x05010400u), x1 = permb(r1, r0, 0x07030602u);   // [r0.b0 r1.b0 r0.b1 r1.b1], [..b2 ..b3]
        const int y0 = permb(r3, r2, 0x05010400u), y1 = permb(r3, r2, 0x07030602u);
        bt[0][w] = permb(y0, x0, 0x05040100u);   // [r0.b0 r1.b0 r2.b0 r3.b0]
        bt[1][w] = permb(y0, x0, 0x07060302u);   // byte 1 of rows 4w..4w+3
        bt[2][w] = permb(y1, x1, 0x05040100u);
        bt[3][w] = permb(y1, x1, 0x07060302u);
    }
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.t[d], bt[t], acc[t][d], 0, 0, 0);
}

// NARROW (at most 256 samples, configs[2]'s shape class): a row has one or two 128-sample blocks, so the four waves of a workgroup
// take four different ROW chunks (`ngroups` then holds the number of 128-sample blocks that hold samples, 1 or 2) instead of four
// adjacent sample blocks of one row chunk -- three of which would be padding.
template <int AUX, bool NARROW = false>
__global__ __launch_bounds__(256, 1) void k_gtt_i8(const int8_t* __restrict__ G, int64_t ldg, int64_t Mpad, int64_t Npad,
                                                    const int8_t* __restrict__ Td, double* __restrict__ Ypart,
                                                    int64_t ngroups, int64_t rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    // (an XCD-aware block -> (row chunk, n-group) remap was measured: no gain -- the skinny operand is already
    //  L2/MALL-served -- and its padded grid broke the all-blocks-resident property, so the plain mapping stays)
    int64_t wchunk, nblock;
    if (NARROW) {
        const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
        nblock = wave % ngroups; wchunk = wave / ngroups;
        if (wchunk * rows_per_wave >= Mpad) return;
    } else {
        const int64_t ngroup = blockIdx.x % ngroups;
        wchunk = blockIdx.x / ngroups;
        nblock = ngroup * 4 + wv;
    }
    const int64_t n0 = nblock * 128;
    if (n0 >= Npad) return;
    const int64_t m_begin = wchunk * rows_per_wave;
    const int64_t m_end = (m_begin + rows_per_wave < Mpad) ? m_begin + rows_per_wave : Mpad;
    const int64_t kblocks = (m_end - m_begin) >> 5;   // multiple of 4

    i32x16 acc[4][kDigits];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < kDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][d][e] = 0;

    const uint32_t gvo = (uint32_t)(16 * h * ldg + 4 * c);
    const uint32_t tvo = (uint32_t)(lane * 16);
    constexpr uint32_t TKB = kDigits * 1024;                // bytes of digit planes per 32-row k-block
    const int8_t* gp = G + m_begin * ldg + n0;
    const int8_t* tp = Td + (m_begin >> 5) * TKB;
    // 4-stage register ring over 32-row k-blocks (kblocks is a multiple of 4): 3 blocks in flight per wave
    Gtt8Buf B0, B1, B2, B3;
    {
        const __amdgpu_buffer_rsrc_t rg0 = make_rsrc8(gp), rt0 = make_rsrc8(tp);
        gtt8_load<AUX>(B0, rg0, gvo, 0u, (uint32_t)ldg, rt0, tvo, 0u);
        gtt8_load<AUX>(B1, rg0, gvo, 32u * (uint32_t)ldg, (uint32_t)ldg, rt0, tvo, TKB);
        gtt8_load<AUX>(B2, rg0, gvo, 64u * (uint32_t)ldg, (uint32_t)ldg, rt0, tvo, 2 * TKB);
    }
    for (int64_t kb = 0; kb < kblocks; kb += 4) {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc8(gp + kb * 32 * ldg);   // re-based every trip: offsets stay < 256 * ldg
        const __amdgpu_buffer_rsrc_t rt = make_rsrc8(tp + kb * TKB);
        const uint32_t more = (kb + 4 < kblocks) ? 1u : 0u;
        const uint32_t L32 = 32u * (uint32_t)ldg;
        gtt8_load<AUX>(B3, rg, gvo, 3u * L32, (uint32_t)ldg, rt, tvo, 3 * TKB);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_compute(B0, acc);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_load<AUX>(B0, rg, gvo, 4u * L32 * more, (uint32_t)ldg, rt, tvo, 4 * TKB * more);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_compute(B1, acc);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_load<AUX>(B1, rg, gvo, (4u * more + 1u) * L32, (uint32_t)ldg, rt, tvo, (4 * more + 1) * TKB);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_compute(B2, acc);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_load<AUX>(B2, rg, gvo, (4u * more + 2u) * L32, (uint32_t)ldg, rt, tvo, (4 * more + 2) * TKB);
        __builtin_amdgcn_sched_barrier(0);
        gtt8_compute(B3, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    // D[j][col]: j = (reg&3) + 8*(reg>>2) + 4*h, col = c -> sample n0 + 4c + t.  Exact integers as f64.
    double* yp = Ypart + (wchunk * Npad) * 32;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t n = n0 + 4 * c + t;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int j = (e & 3) + 8 * (e >> 2) + 4 * h;
            double2 o;
            o.x = combine_digits(acc[t], e); o.y = combine_digits(acc[t], e + 1);
            *reinterpret_cast<double2*>(yp + n * 32 + j) = o;
        }
    }
}

// K2 for at most 256 samples (int8 rows): plan = gtt8_plan_narrow; every wave owns a row chunk of its own
Gtt8Plan gtt8_plan_narrow(int64_t Mpad, int64_t N, int target_waves) {
    Gtt8Plan p{};
    p.nblocks_n = (N + 127) / 128;                      // 128-sample blocks that hold samples: 1 or 2
    int64_t W = target_waves / p.nblocks_n;
    if (W < 1) W = 1;
    const int64_t maxW = Mpad / 128;
    if (W > maxW) W = maxW;
    int64_t rpw = (Mpad + W - 1) / W;
    rpw = (rpw + 127) / 128 * 128;
    W = (Mpad + rpw - 1) / rpw;
    p.W = (int)W;
    p.rows_per_wave = rpw;
    p.grid = (W * p.nblocks_n + 3) / 4;
    return p;
}
void launch_gtt_n(hipStream_t st, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const int8_t* Td,
                  double* Ypart, const Gtt8Plan& plan) {
    hipLaunchKernelGGL((k_gtt_i8<2, true>), dim3((unsigned)plan.grid), dim3(256), 0, st, G, ldg, Mpad, Npad, Td, Ypart, plan.nblocks_n, plan.rows_per_wave);
}

void launch_gtt_i8(hipStream_t st, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const int8_t* Td,
                   double* Ypart, const Gtt8Plan& plan, const KernelOpts& ko) {
    const int64_t ngroups = (plan.nblocks_n + 3) / 4;
    hipLaunchKernelGGL((k_gtt_i8<0>), dim3((unsigned)plan.grid), dim3(256), 0, st, G, ldg, Mpad, Npad, Td, Ypart, ngroups, plan.rows_per_wave);
}

// ---- K2, packed.  Lane (c, h) loads ONE byte (4 samples) from each of its 16 SNP rows; the four waves of a workgroup
// cover 128 adjacent bytes of every row.  Four rows are OR-ed into a dword, and operand t is (x >> 2t) & 0x03030303.
struct Gtt2Buf { unsigned g[16]; i32x4 t[kDigits]; };

// (one-byte loads made the address unit the bottleneck -- 16 buffer_load_ubyte per 32-row block cost ~2700 cycles;
//  lanes 4j..4j+3 now load the same dword and each extracts its byte)
__device__ __forceinline__ void gtt2_load(Gtt2Buf& b, __amdgpu_buffer_rsrc_t rg, uint32_t gvo, uint32_t row_off, uint32_t ld2,
                                          __amdgpu_buffer_rsrc_t rt, uint32_t tvo, uint32_t toff) {
#pragma unroll
    for (int i = 0; i < 16; ++i) b.g[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rg, gvo, row_off + (uint32_t)i * ld2, 0);
#pragma unroll
    for (int d = 0; d < kDigits; ++d) b.t[d] = __builtin_amdgcn_raw_buffer_load_b128(rt, tvo, toff + d * 1024, 0);
}
__device__ __forceinline__ unsigned gtt2_quad(const Gtt2Buf& b, int w, unsigned bsh) {
    // this lane's byte (bit offset bsh = 8 * (c & 3)) of rows 4w..4w+3 -> one dword
    return ((b.g[4 * w] >> bsh) & 0xffu) | (((b.g[4 * w + 1] >> bsh) & 0xffu) << 8) |
           (((b.g[4 * w + 2] >> bsh) & 0xffu) << 16) | ((b.g[4 * w + 3] >> bsh) << 24);
}
__device__ __forceinline__ void gtt2_decode(const Gtt2Buf& b, Gtt2Ops& o, unsigned bsh) {
    unsigned x[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) x[w] = gtt2_quad(b, w, bsh);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int w = 0; w < 4; ++w) o.bt[t][w] = (int)((x[w] >> (2 * t)) & 0x03030303u);
}
// 16 int8 MFMAs of block k, each followed by one of 16 micro-steps of block k+1's decode (per row quad w: combine 4 byte
// loads into a dword, then the four shift/mask operands); sched_barrier(0) pins the interleave.
__device__ __forceinline__ void gtt2_mfma_decode(const Gtt2Buf& b, const Gtt2Ops& o, i32x16 (&acc)[4][kDigits],
                                                 const Gtt2Buf& bn, Gtt2Ops& on, unsigned bsh) {
    unsigned x[4];
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.t[d], o.bt[t], acc[t][d], 0, 0, 0);
            const int m = d * 4 + t, w = m >> 2, ph = m & 3;
            if (ph == 0) x[w] = gtt2_quad(bn, w, bsh);
            if (ph == 1) { on.bt[0][w] = (int)(x[w] & 0x03030303u); on.bt[1][w] = (int)((x[w] >> 2) & 0x03030303u); }
            if (ph == 2) on.bt[2][w] = (int)((x[w] >> 4) & 0x03030303u);
            if (ph == 3) on.bt[3][w] = (int)((x[w] >> 6) & 0x03030303u);
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <int ND>
__global__ __launch_bounds__(256, 1) void k_gtt_2bit(const uint8_t* __restrict__ G2, int64_t ld2, int64_t Mpad, int64_t Npad,
                                                      const int8_t* __restrict__ Td, double* __restrict__ Ypart,
                                                      int64_t ngroups, int64_t rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t ngroup = blockIdx.x % ngroups;
    const int64_t wchunk = blockIdx.x / ngroups;
    const int64_t nblock = ngroup * 4 + wv;
    const int64_t n0 = nblock * 128;
    if (n0 >= Npad) return;
    const int64_t m_begin = wchunk * rows_per_wave;
    const int64_t m_end = (m_begin + rows_per_wave < Mpad) ? m_begin + rows_per_wave : Mpad;
    const int64_t kblocks = (m_end - m_begin) >> 5;   // multiple of 4

    i32x16 acc[4][kDigits];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < kDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][d][e] = 0;

    const uint32_t gvo = (uint32_t)(16 * h * ld2 + (c & ~3));   // the dword holding this lane's byte
    const unsigned bsh = 8u * (unsigned)(c & 3);
    const uint32_t tvo = (uint32_t)(lane * 16);
    constexpr uint32_t TKB = kDigits * 1024;
    const uint8_t* gp = G2 + m_begin * ld2 + (n0 >> 2);
    const int8_t* tp = Td + (m_begin >> 5) * TKB;
    Gtt2Buf B0, B1, B2, B3;
    Gtt2Ops OA, OB;
    {
        const __amdgpu_buffer_rsrc_t rg0 = make_rsrc8(gp), rt0 = make_rsrc8(tp);
        gtt2_load(B0, rg0, gvo, 0u, (uint32_t)ld2, rt0, tvo, 0u);
        gtt2_load(B1, rg0, gvo, 32u * (uint32_t)ld2, (uint32_t)ld2, rt0, tvo, TKB);
        gtt2_load(B2, rg0, gvo, 64u * (uint32_t)ld2, (uint32_t)ld2, rt0, tvo, 2 * TKB);
    }
    gtt2_decode(B0, OA, bsh);
    // phase k: load block k+3; the 16 int8 MFMAs of block k and the decode (OR + shift/mask, VALU) of block k+1 share one
    // scheduling region so that they interleave; operand sets alternate OA / OB
    for (int64_t kb = 0; kb < kblocks; kb += 4) {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc8(gp + kb * 32 * ld2);
        const __amdgpu_buffer_rsrc_t rt = make_rsrc8(tp + kb * TKB);
        const uint32_t more = (kb + 4 < kblocks) ? 1u : 0u;
        const uint32_t L32 = 32u * (uint32_t)ld2;
        gtt2_load(B3, rg, gvo, 3u * L32, (uint32_t)ld2, rt, tvo, 3 * TKB);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_mfma_decode(B0, OA, acc, B1, OB, bsh);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_load(B0, rg, gvo, 4u * L32 * more, (uint32_t)ld2, rt, tvo, 4 * TKB * more);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_mfma_decode(B1, OB, acc, B2, OA, bsh);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_load(B1, rg, gvo, (4u * more + 1u) * L32, (uint32_t)ld2, rt, tvo, (4 * more + 1) * TKB);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_mfma_decode(B2, OA, acc, B3, OB, bsh);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_load(B2, rg, gvo, (4u * more + 2u) * L32, (uint32_t)ld2, rt, tvo, (4 * more + 2) * TKB);
        __builtin_amdgcn_sched_barrier(0);
        gtt2_mfma_decode(B3, OB, acc, B0, OA, bsh);
        __builtin_amdgcn_sched_barrier(0);
    }
    double* yp = Ypart + (wchunk * Npad) * 32;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t n = n0 + 4 * c + t;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int j = (e & 3) + 8 * (e >> 2) + 4 * h;
            double2 o;
            o.x = combine_digits<ND == 3 ? 8 : 7>(acc[t], e); o.y = combine_digits<ND == 3 ? 8 : 7>(acc[t], e + 1);
            *reinterpret_cast<double2*>(yp + n * 32 + j) = o;
        }
    }
}

void launch_gtt_2bit(hipStream_t st, const uint8_t* G2, int64_t ld2, int64_t Mpad, int64_t Npad, const int8_t* Td,
                     double* Ypart, const Gtt8Plan& plan, int nd) {
    const int64_t ngroups = (plan.nblocks_n + 3) / 4;
    if (nd == 3) hipLaunchKernelGGL(k_gtt_2bit<3>, dim3((unsigned)plan.grid), dim3(256), 0, st, G2, ld2, Mpad, Npad, Td, Ypart, ngroups, plan.rows_per_wave);   // (plane 3 of a three-plane operand is all zero: its MFMAs add nothing)
    else hipLaunchKernelGGL(k_gtt_2bit<kDigits>, dim3((unsigned)plan.grid), dim3(256), 0, st, G2, ld2, Mpad, Npad, Td, Ypart, ngroups, plan.rows_per_wave);
}

}  // namespace gpca
