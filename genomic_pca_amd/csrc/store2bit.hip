// 2-bit resident genotypes (GPCA_STORE_2BIT): G2 [Mpad][ld2] bytes, 4 samples per byte LSB-first like PLINK, but each
// field holds the DOSAGE code  0, 1, 2  (3 = missing), so the GEMM prologues only have to spread bit fields.
// ld2 = round_up(N, 512) / 4 (rows are 128-byte aligned); pad fields are 0.
//
// Producers: PLINK .bed payload -> recode in place (count_a1: 00->2, 01->missing, 10->1, 11->0, prepare.rs:622-629);
// int8 dosages (upload / synthetic generator, chunk by chunk through a scratch buffer) -> pack.
// Consumers here: SNP QC statistics straight from the packed words (popcounts) and the pull-API gather.
#include "kernels.h"

namespace gpca {

#define DEVINL __device__ __forceinline__

// PLINK 2-bit codes -> dosage codes, 16 samples per 32-bit word:  new_hi = ~c1, new_lo = c1 ^ c0
DEVINL unsigned plink_to_dosage_codes(unsigned x) {
    const unsigned c0 = x & 0x55555555u, c1 = (x >> 1) & 0x55555555u;
    return (((~c1) & 0x55555555u) << 1) | (c1 ^ c0);
}

// bed: [M][bpr] PLINK bytes (bpr = ceil(N/4)); G2: [Mpad][ld2].  One thread = one 32-bit word (16 samples).
__global__ __launch_bounds__(256) void k_bed_to_codes(const uint8_t* __restrict__ bed, int64_t bpr, uint8_t* __restrict__ G2,
                                                      int64_t M, int64_t N, int64_t ld2) {
    const int64_t wpr = ld2 >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * wpr) return;
    const int64_t i = t / wpr, wi = t - i * wpr;
    const int64_t b0 = wi * 4;
    unsigned x = 0;
    const uint8_t* src = bed + i * bpr + b0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (b0 + k < bpr) x |= (unsigned)src[k] << (8 * k);
    unsigned y = plink_to_dosage_codes(x);
    const int64_t n0 = wi * 16;
    if (n0 + 16 > N) {   // zero the fields of samples >= N
        const int valid = (int)(N > n0 ? N - n0 : 0);
        y &= valid >= 16 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (2 * valid)) - 1u));
    }
    *reinterpret_cast<unsigned*>(G2 + i * ld2 + b0) = y;
}
void launch_bed_to_codes(hipStream_t st, const uint8_t* bed, int64_t bpr, uint8_t* G2, int64_t M, int64_t N, int64_t ld2) {
    const int64_t total = M * (ld2 >> 2);
    hipLaunchKernelGGL(k_bed_to_codes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bed, bpr, G2, M, N, ld2);
}

// int8 dosages (rows [0, rows) of a scratch buffer with pitch ld8, pad bytes 0) -> packed rows starting at G2.
// -127 -> code 3; any other value outside {0,1,2} -> code 3 and flags[0] |= 2 (invalid genotype).
__global__ __launch_bounds__(256) void k_pack_i8(const int8_t* __restrict__ G8, int64_t ld8, uint8_t* __restrict__ G2,
                                                 int64_t rows, int64_t N, int64_t ld2, unsigned* __restrict__ flags) {
    const int64_t wpr = ld2 >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * wpr) return;
    const int64_t i = t / wpr, wi = t - i * wpr;
    const int64_t n0 = wi * 16;
    unsigned y = 0;
    bool bad = false;
    if (n0 < ld8) {
        const uint4 q = *reinterpret_cast<const uint4*>(G8 + i * ld8 + n0);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int v = (int)(int8_t)((w[s >> 2] >> (8 * (s & 3))) & 0xffu);
            unsigned c = (unsigned)v;
            if (v == -127) c = 3u;
            else if (v < 0 || v > 2) { c = 3u; bad = true; }
            if (n0 + s >= N) c = 0u;
            y |= c << (2 * s);
        }
    }
    *reinterpret_cast<unsigned*>(G2 + i * ld2 + wi * 4) = y;
    if (bad) atomicOr(flags, 2u);
}
void launch_pack_i8(hipStream_t st, const int8_t* G8, int64_t ld8, uint8_t* G2, int64_t rows, int64_t N, int64_t ld2,
                    unsigned* flags) {
    const int64_t total = rows * (ld2 >> 2);
    hipLaunchKernelGGL(k_pack_i8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, G8, ld8, G2, rows, N, ld2, flags);
}

// ------------------------------------------------------------------------------------------------
// a1 on packed words: per SNP  n1 = #01, n2 = #10, n_missing = #11, n0 = N - rest; then the same QC chain
// (prepare.rs:1281-1364) as k_snp_stats.  One wave per SNP row, 16 B (64 samples) per lane per step.
// ------------------------------------------------------------------------------------------------
__device__ double hwe_p_dev2(unsigned n1h, unsigned nhet, unsigned n2h) {   // prepare.rs:1641-1745 (copy of kernels.hip)
    const unsigned long long tot = (unsigned long long)n1h + nhet + n2h;
    if (tot == 0) return 1.0;
    const double c1 = 2.0 * (double)n1h + (double)nhet, c2 = 2.0 * (double)n2h + (double)nhet;
    const double ta = c1 + c2;
    if (ta <= 1e-9) return 1.0;
    const double f1 = c1 / ta, f2 = c2 / ta;
    if (f1 < 1e-9 || f2 < 1e-9) return 1.0;
    if (fabs(f1 + f2 - 1.0) > 1e-6) return 1.0;
    const double e1 = f1 * f1 * (double)tot, eh = 2.0 * f1 * f2 * (double)tot, e2 = f2 * f2 * (double)tot;
    double chi = 0.0;
    const double MINE = 1e-9;
    if (e1 > MINE) { const double d = (double)n1h - e1; chi += d * d / e1; } else if ((double)n1h > MINE) chi = INFINITY;
    if (isfinite(chi)) { if (eh > MINE) { const double d = (double)nhet - eh; chi += d * d / eh; } else if ((double)nhet > MINE) chi = INFINITY; }
    if (isfinite(chi)) { if (e2 > MINE) { const double d = (double)n2h - e2; chi += d * d / e2; } else if ((double)n2h > MINE) chi = INFINITY; }
    if (isnan(chi)) return 1.0;
    if (chi == INFINITY) return 0.0;
    const double cdf = erf(sqrt(chi * 0.5));
    if (isnan(cdf)) return 1.0;
    const double p = 1.0 - cdf;
    return p > 0.0 ? p : 0.0;
}

__global__ __launch_bounds__(256) void k_snp_stats_2bit(const uint8_t* __restrict__ G2, int64_t M, int64_t N, int64_t ld2,
                                                         QcParams qc, float* __restrict__ mu, float* __restrict__ sigma,
                                                         float* __restrict__ rr, float* __restrict__ bb,
                                                         uint8_t* __restrict__ keep, uint8_t* __restrict__ reason,
                                                         uint32_t* __restrict__ counts, uint32_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // KiB-aligned wave loads over the row's samples (the pitch is an odd multiple of 256 B: see k_snp_stats)
    const uintptr_t rstart = reinterpret_cast<uintptr_t>(G2 + row * ld2);
    const int64_t skip = (int64_t)((rstart & 1023u) >> 4);
    const uint4* p = reinterpret_cast<const uint4*>(rstart & ~(uintptr_t)1023u);
    const int64_t nvec = skip + ((N + 63) >> 6);        // 64 samples per 16-byte vector; fields between N and the pitch are 0
    int c1 = 0, c2 = 0, c3 = 0;
    for (int64_t v0 = lane; v0 < nvec; v0 += 64) {
        if (v0 < skip) continue;
        const uint4 q = p[v0];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lo = w[k] & 0x55555555u, hi = (w[k] >> 1) & 0x55555555u;
            c1 += __builtin_popcount(lo & ~hi); c2 += __builtin_popcount(hi & ~lo); c3 += __builtin_popcount(hi & lo);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); c3 += __shfl_xor(c3, o); }
    if (lane != 0) return;
    const long long nv = (long long)N - c3;
    const unsigned n1 = (unsigned)c1, n2 = (unsigned)c2, n0 = (unsigned)(nv - c1 - c2);
    const long long s1 = (long long)c1 + 2ll * c2, s2 = (long long)c1 + 4ll * c2;
    counts[4 * row + 0] = (unsigned)nv; counts[4 * row + 1] = n0; counts[4 * row + 2] = n1; counts[4 * row + 3] = n2;
    uint8_t why = 0;
    double mean = 0.0;
    do {
        const double call_rate = (double)nv / (double)N;
        if (call_rate < qc.min_call_rate) { why = 1; break; }
        if (nv == 0) { why = 2; break; }
        mean = (double)s1 / (double)nv;
        const double pfr = mean / 2.0;
        const double maf = pfr < 1.0 - pfr ? pfr : 1.0 - pfr;
        if (maf < qc.min_maf) { why = 3; break; }
        if (fabs(pfr) < 1e-9 || fabs(1.0 - pfr) < 1e-9) { why = 4; break; }
        if (qc.max_hwe_p < 1.0) { if (hwe_p_dev2(n0, n1, n2) <= qc.max_hwe_p) { why = 5; break; } }
    } while (0);
    float m32 = 0.f, s32 = 0.f, r32 = 0.f, b32 = 0.f;
    if (!why) {
        double var = 0.0;
        if (nv >= 2) var = (((double)nv * (double)s2 - (double)s1 * (double)s1) / (double)nv) / (double)(nv - 1);
        if (var <= 1e-9) why = 6;
        else {
            m32 = (float)mean; s32 = (float)sqrt(var); r32 = 1.0f / s32; b32 = -m32 * r32;
            if (nv != N) atomicOr(flags, 1u);
        }
    }
    mu[row] = m32; sigma[row] = s32; rr[row] = r32; bb[row] = b32;
    keep[row] = why ? 0 : 1; reason[row] = why;
}
void launch_snp_stats_2bit(hipStream_t st, const uint8_t* G2, int64_t M, int64_t N, int64_t ld2, QcParams qc, float* mu,
                           float* sigma, float* r, float* b, uint8_t* keep, uint8_t* reason, uint32_t* counts, uint32_t* flags) {
    hipLaunchKernelGGL(k_snp_stats_2bit, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, G2, M, N, ld2, qc, mu, sigma, r, b,
                       keep, reason, counts, flags);
}

// a2 gather from packed storage (prepare.rs:1884-2016)
__global__ __launch_bounds__(256) void k_standardize_block_2bit(const uint8_t* __restrict__ G2, int64_t ld2,
                                                                 const float* __restrict__ mu, const float* __restrict__ sigma,
                                                                 const int64_t* __restrict__ rows, int64_t ns,
                                                                 const int64_t* __restrict__ cols, int64_t nj,
                                                                 float* __restrict__ out, unsigned long long* err_idx) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ns * nj) return;
    const int64_t a = t / nj, c = t - a * nj;
    const int64_t i = rows[a], n = cols[c];
    const unsigned code = (G2[i * ld2 + (n >> 2)] >> (2 * (n & 3))) & 3u;
    if (code == 3u) { atomicMin(err_idx, (unsigned long long)t); return; }
    const float sd = sigma[i];
    float o = 0.0f;
    if (!(fabsf(sd) < 1e-9f)) {
        const float rs = 1.0f / sd;
        const float bt = -mu[i] * rs;
        o = __builtin_fmaf((float)code, rs, bt);
    }
    out[t] = o;
}
void launch_standardize_block_2bit(hipStream_t st, const uint8_t* G2, int64_t ld2, const float* mu, const float* sigma,
                                   const int64_t* rows, int64_t ns, const int64_t* cols, int64_t nj, float* out,
                                   unsigned long long* err_idx) {
    const int64_t total = ns * nj;
    hipLaunchKernelGGL(k_standardize_block_2bit, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, G2, ld2, mu, sigma,
                       rows, ns, cols, nj, out, err_idx);
}

}  // namespace gpca
