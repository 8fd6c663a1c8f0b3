// Standalone microbenchmark for the G^T T (K2) inner loop on gfx950: isolates what keeps the fp32 MFMA
// pipe from 100 % (loads, cvt+nop, register moves, occupancy).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o kbench_gtt kbench_gtt.hip && ./kbench_gtt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// VAR bit0: real loads each group (else reuse first group's registers); bit1: cvt from bytes (else use raw float bits)
// bit2: copy nxt->cur with v_mov (else swap by 2x unroll)
template <int VAR, int WPS>
__global__ __launch_bounds__(256, WPS) void k_gtt(const int8_t* __restrict__ G, int64_t ldg, int64_t Mpad, int64_t Npad,
                                                   const float* __restrict__ Tp, float* __restrict__ Ypart, int64_t ngroups,
                                                   int64_t rows_per_wave, unsigned long long* clk) {
    constexpr int L = 32;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t ngroup = blockIdx.x % ngroups;
    const int64_t wchunk = blockIdx.x / ngroups;
    const int64_t nblock = ngroup * 4 + wv;
    const int64_t n0 = nblock * 256;
    if (n0 >= Npad) return;
    const int64_t m_begin = wchunk * rows_per_wave;
    const int64_t m_end = (m_begin + rows_per_wave < Mpad) ? m_begin + rows_per_wave : Mpad;
    const int64_t groups = (m_end - m_begin) >> 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const char* gs = reinterpret_cast<const char*>(G) + m_begin * ldg + n0;
    const uint32_t gvo = (uint32_t)(h * ldg + 8 * c);
    const char* ts = reinterpret_cast<const char*>(Tp + m_begin * L);
    const uint32_t tvo = (uint32_t)((h * L + c) * 4);
    uint2 ga[8], gb[8];
    float ta[8], tb[8];
    float bvs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        ga[u] = *reinterpret_cast<const uint2*>(gs + (2 * u) * ldg + gvo);
        ta[u] = *reinterpret_cast<const float*>(ts + tvo + (2 * u * L) * 4);
    }
    for (int64_t g = 0; g < groups; ++g) {
        const int64_t gn = (g + 1 < groups) ? g + 1 : g;
        if (VAR & 1) {
            const char* g1 = gs + gn * 16 * ldg;
            const char* t1 = ts + gn * 16 * (L * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                gb[u] = *reinterpret_cast<const uint2*>(g1 + (2 * u) * ldg + gvo);
                tb[u] = *reinterpret_cast<const float*>(t1 + tvo + (2 * u * L) * 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (VAR & 64) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            float bv[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f32x2 p0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)ga[u].x, false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)ga[u].x, true);
                f32x2 p2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)ga[u].y, false), p3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)ga[u].y, true);
                bv[u][0] = p0[0]; bv[u][1] = p0[1]; bv[u][2] = p1[0]; bv[u][3] = p1[1];
                bv[u][4] = p2[0]; bv[u][5] = p2[1]; bv[u][6] = p3[0]; bv[u][7] = p3[1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[u], bv[u][t], acc[t], 0, 0, 0);
        } else if (VAR & 8) {
            float bv[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned w = (t < 4) ? ga[u].x : ga[u].y;
                    bv[u][t] = (float)((w >> (8 * (t & 3))) & 0xffu);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[u], bv[u][t], acc[t], 0, 0, 0);
        } else if (VAR & 32) {
            // software-pipelined conversion, hand-placed: each asm statement = one MFMA + the cvt of the operand
            // that the MFMA 8 statements later will read (no RAW nops, 1 VALU filler per 64-cycle MFMA)
            float bv[8];
            if (g == 0) {
#pragma unroll
                for (int t = 0; t < 8; ++t) { const unsigned w = (t < 4) ? ga[0].x : ga[0].y; bv[t] = (float)((w >> (8 * (t & 3))) & 0xffu); }
#pragma unroll
                for (int t = 0; t < 8; ++t) bvs[t] = bv[t];
                asm volatile("s_nop 1");
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) bv[t] = bvs[t];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float bn[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned w = (u < 7) ? ((t < 4) ? ga[u + 1].x : ga[u + 1].y) : ((t < 4) ? gb[0].x : gb[0].y);
                    if ((t & 3) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\tv_cvt_f32_ubyte0_e32 %1, %4" : "+v"(acc[t]), "=&v"(bn[t]) : "v"(ta[u]), "v"(bv[t]), "v"(w));
                    if ((t & 3) == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\tv_cvt_f32_ubyte1_e32 %1, %4" : "+v"(acc[t]), "=&v"(bn[t]) : "v"(ta[u]), "v"(bv[t]), "v"(w));
                    if ((t & 3) == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\tv_cvt_f32_ubyte2_e32 %1, %4" : "+v"(acc[t]), "=&v"(bn[t]) : "v"(ta[u]), "v"(bv[t]), "v"(w));
                    if ((t & 3) == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\tv_cvt_f32_ubyte3_e32 %1, %4" : "+v"(acc[t]), "=&v"(bn[t]) : "v"(ta[u]), "v"(bv[t]), "v"(w));
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) bv[t] = bn[t];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) bvs[t] = bv[t];
        } else if (VAR & 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float bv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned w = (t < 4) ? ga[u].x : ga[u].y;
                    bv[t] = (float)((w >> (8 * (t & 3))) & 0xffu);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[u], bv[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const unsigned w = (t < 4) ? ga[u].x : ga[u].y;
                float bval;
                if (VAR & 2) bval = (float)((w >> (8 * (t & 3))) & 0xffu);
                else bval = __builtin_bit_cast(float, w);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[u], bval, acc[t], 0, 0, 0);
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (VAR & 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { ga[u] = gb[u]; ta[u] = tb[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(ga[u].x), "+v"(ga[u].y), "+v"(ta[u]));
        }
    }
    float* yp = Ypart + (wchunk * Npad) * L;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int64_t n = n0 + 8 * c + t;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const int j = 8 * (e >> 2) + 4 * h;
            float4 o; o.x = acc[t][e]; o.y = acc[t][e + 1]; o.z = acc[t][e + 2]; o.w = acc[t][e + 3];
            *reinterpret_cast<float4*>(yp + n * L + j) = o;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x == 0 && wv == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}


// v4: 1 wave/SIMD; raw buffer loads (SGPR row offsets, no per-lane address VALU); T' in a blocked layout
// [group][lane][8] (two 16-byte loads per group); bulk cvt_pk_f32_fp8; 2x unrolled buffer swap (no v_mov).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct GBuf { i32x2 g[8]; i32x4 t0, t1; };

__device__ __forceinline__ void v4_load(GBuf& b, __amdgpu_buffer_rsrc_t rg, uint32_t gvo, uint32_t row_off, uint32_t ldg,
                                        __amdgpu_buffer_rsrc_t rt, uint32_t tvo, uint32_t t_off) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b.g[u] = __builtin_amdgcn_raw_buffer_load_b64(rg, gvo, row_off + (uint32_t)(2 * u) * ldg, 0);
    b.t0 = __builtin_amdgcn_raw_buffer_load_b128(rt, tvo, t_off, 0);
    b.t1 = __builtin_amdgcn_raw_buffer_load_b128(rt, tvo + 16, t_off, 0);
}
__device__ __forceinline__ void v4_compute(const GBuf& b, f32x16 (&acc)[8]) {
    float bv[8][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        f32x2 p0 = __builtin_amdgcn_cvt_pk_f32_fp8(b.g[u][0], false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8(b.g[u][0], true);
        f32x2 p2 = __builtin_amdgcn_cvt_pk_f32_fp8(b.g[u][1], false), p3 = __builtin_amdgcn_cvt_pk_f32_fp8(b.g[u][1], true);
        bv[u][0] = p0[0]; bv[u][1] = p0[1]; bv[u][2] = p1[0]; bv[u][3] = p1[1];
        bv[u][4] = p2[0]; bv[u][5] = p2[1]; bv[u][6] = p3[0]; bv[u][7] = p3[1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float ta = __builtin_bit_cast(float, u < 4 ? b.t0[u] : b.t1[u - 4]);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta, bv[u][t], acc[t], 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 1) void k_gtt_v4(const int8_t* __restrict__ G, int64_t ldg, int64_t Mpad, int64_t Npad,
                                                    const float* __restrict__ Tb, float* __restrict__ Ypart, int64_t ngroups,
                                                    int64_t rows_per_wave, unsigned long long* clk) {
    constexpr int L = 32;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t ngroup = blockIdx.x % ngroups;
    const int64_t wchunk = blockIdx.x / ngroups;
    const int64_t nblock = ngroup * 4 + wv;
    const int64_t n0 = nblock * 256;
    if (n0 >= Npad) return;
    const int64_t m_begin = wchunk * rows_per_wave;
    const int64_t m_end = (m_begin + rows_per_wave < Mpad) ? m_begin + rows_per_wave : Mpad;
    const int64_t groups = (m_end - m_begin) >> 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const uint32_t gvo = (uint32_t)(h * ldg + 8 * c);
    const uint32_t tvo = (uint32_t)(lane * 32);
    // descriptors are re-based every 2 groups (32 rows), so 32-bit offsets never overflow for any ldg < 2^26
    const int8_t* gp = G + m_begin * ldg + n0;
    const float* tp = Tb + (m_begin >> 4) * 512;     // blocked: 512 floats per 16-row group
    GBuf A, B;
    {
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)tp, 0, 0x7fffffff, 0x00020000);
        v4_load(A, rg, gvo, 0, (uint32_t)ldg, rt, tvo, 0);
    }
    for (int64_t g = 0; g < groups; g += 2) {
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gp + g * 16 * ldg), 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(tp + g * 512), 0, 0x7fffffff, 0x00020000);
        // groups is even (rows_per_wave and Mpad are multiples of 32): straight-line body, no branches
        const uint32_t o2 = (g + 2 < groups) ? 32u : 0u;
        v4_load(B, rg, gvo, 16u * (uint32_t)ldg, (uint32_t)ldg, rt, tvo, 16u * 128u);
        __builtin_amdgcn_sched_barrier(0);
        v4_compute(A, acc);
        __builtin_amdgcn_sched_barrier(0);
        v4_load(A, rg, gvo, o2 * (uint32_t)ldg, (uint32_t)ldg, rt, tvo, o2 * 128u);
        __builtin_amdgcn_sched_barrier(0);
        v4_compute(B, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    float* yp = Ypart + (wchunk * Npad) * L;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int64_t n = n0 + 8 * c + t;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const int j = 8 * (e >> 2) + 4 * h;
            float4 o; o.x = acc[t][e]; o.y = acc[t][e + 1]; o.z = acc[t][e + 2]; o.w = acc[t][e + 3];
            *reinterpret_cast<float4*>(yp + n * L + j) = o;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x == 0 && wv == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

void run_v4(const char* name, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const float* Tp, float* Ypart,
            int target_waves, unsigned long long* d_clk) {
    const int64_t nb = Npad / 256;
    int64_t W = target_waves / nb; if (W < 1) W = 1;
    int64_t rpw = (Mpad + W - 1) / W; rpw = (rpw + 31) / 32 * 32; W = (Mpad + rpw - 1) / rpw;
    const int64_t ngroups = (nb + 3) / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_gtt_v4, dim3((unsigned)(ngroups * W)), dim3(256), 0, 0, G, ldg, Mpad, Npad, Tp, Ypart, ngroups, rpw, d_clk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0) { sum += ms; if (ms < best) best = ms; }
    }
    unsigned long long clk[2]; CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    const double flops = 2.0 * Mpad * Npad * 32;
    printf("%-34s waves=%5lld W=%3lld  avg %.3f ms  best %.3f ms  %.1f TF(padded)  in-kernel clock %.2f GHz\n", name,
           (long long)(ngroups * W * 4), (long long)W, sum / (reps - 1), best, flops / (best * 1e-3) / 1e12,
           (double)clk[0] / (double)clk[1] * 0.1);
}

template <int VAR, int WPS>
void run(const char* name, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const float* Tp, float* Ypart,
         int target_waves, unsigned long long* d_clk) {
    const int64_t nb = Npad / 256;
    int64_t W = target_waves / nb; if (W < 1) W = 1;
    int64_t rpw = (Mpad + W - 1) / W; rpw = (rpw + 15) / 16 * 16; W = (Mpad + rpw - 1) / rpw;
    const int64_t ngroups = (nb + 3) / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gtt<VAR, WPS>), dim3((unsigned)(ngroups * W)), dim3(256), 0, 0, G, ldg, Mpad, Npad, Tp, Ypart,
                           ngroups, rpw, d_clk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0) { sum += ms; if (ms < best) best = ms; }
    }
    unsigned long long clk[2]; CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    const double flops = 2.0 * Mpad * Npad * 32;
    printf("%-34s waves=%5lld W=%3lld  avg %.3f ms  best %.3f ms  %.1f TF(padded)  in-kernel clock %.2f GHz\n", name,
           (long long)(ngroups * W * 4), (long long)W, sum / (reps - 1), best, flops / (best * 1e-3) / 1e12,
           (double)clk[0] / (double)clk[1] * 0.1);
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000, N = argc > 2 ? atoll(argv[2]) : 10000;
    const int64_t Mpad = (M + 127) / 128 * 128, Npad = (N + 255) / 256 * 256, ldg = Npad;
    int8_t* G; float *Tp, *Ypart; unsigned long long* d_clk;
    CK(hipMalloc(&G, (size_t)Mpad * ldg)); CK(hipMalloc(&Tp, (size_t)Mpad * 32 * 4)); CK(hipMalloc(&Ypart, (size_t)64 * Npad * 32 * 4));
    CK(hipMalloc(&d_clk, 16));
    {   // random 0/1/2 bytes and random floats
        std::vector<int8_t> hg((size_t)1 << 26);
        for (size_t i = 0; i < hg.size(); ++i) hg[i] = (int8_t)(rand() % 3);
        for (size_t off = 0; off < (size_t)Mpad * ldg; off += hg.size())
            CK(hipMemcpy(G + off, hg.data(), std::min(hg.size(), (size_t)Mpad * ldg - off), hipMemcpyHostToDevice));
        std::vector<float> ht((size_t)Mpad * 32);
        for (size_t i = 0; i < ht.size(); ++i) ht[i] = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(Tp, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    }
    printf("M=%lld N=%lld (Mpad %lld Npad %lld)\n", (long long)M, (long long)N, (long long)Mpad, (long long)Npad);
    for (int rep = 0; rep < 2; ++rep) {
        run<65, 1>("bulk cvt_pk_fp8, loads 1w", G, ldg, Mpad, Npad, Tp, Ypart, 1024, d_clk);
        run_v4("v4 buffer loads + blocked T 1w", G, ldg, Mpad, Npad, Tp, Ypart, 1024, d_clk);
        run<64, 1>("bulk cvt_pk_fp8, no loads 1w", G, ldg, Mpad, Npad, Tp, Ypart, 1024, d_clk);
    }
    return 0;
}
