// gfx950 (MI355X, CDNA4) kernels of the randomized-PCA hot path.  wave = 64 lanes everywhere.
//
// Data layout in HBM
//   G      int8  [M][ldg]      SNP-major dosages, ldg = round_up(N, 256), pad bytes are 0
//   Q      f32   [Npad][L]     sample-side orthonormal basis, Npad = ldg, L = 32 or 64, pad rows/cols 0
//   T / B  f32   [M][L]        SNP-side sketch (r o T for power iterations, T itself for the projection)
//   Y      f64   [N][L]        sketch accumulator (sum over SNP shards / GPUs happens in f64)
//
// The standardised matrix A[i][n] = g*r_i + b_i (r = 1/sigma, b = -mu r: prepare.rs:1948-1949) is never
// materialised:  A Q = r o (G Q) + b (1^T Q)   and   A^T T = G^T (r o T) + 1 (b^T T).
// So both tall-skinny GEMMs run on the raw 0/1/2 bytes (one v_cvt_f32_ubyteN per MFMA operand) and the
// standardisation is a per-row epilogue -- the reference's per-block f32 standardise pass
// (prepare.rs:1884-2016) disappears from the hot loop.
#include "kernels.h"
#include "philox.hpp"
#include "omega_math.h"

#ifndef GPCA_OMEGA_ABLATE
#define GPCA_OMEGA_ABLATE 0     // scripts/kbench/kbench_omega.hip
#endif
namespace gpca {

__device__ const OmegaLnEntry kOmegaLnTable[kOmegaLnEntries] = {
#include "omega_table.inc"
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DEVINL __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// Synthetic genotypes.  One thread = 8 consecutive samples of one SNP (one 8-byte store).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth(int8_t* __restrict__ G, int64_t M, int64_t N, int64_t ld,
                                                int64_t snp_offset, uint64_t seed,
                                                const uint32_t* __restrict__ thresh, int P) {
    const int64_t per_row = ld >> 3;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * per_row) return;
    const int64_t i = t / per_row;
    const int64_t n0 = (t - i * per_row) << 3;
    const uint64_t gi = (uint64_t)(i + snp_offset);
    const uint32_t* th = thresh + i * P;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t n = n0 + 2 * q;
        if (n < N) {
            philox_out o = philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)(n >> 1), GPCA_STREAM_GENO,
                                         (uint32_t)seed, (uint32_t)(seed >> 32));
            uint32_t ta = th[n % P];
            uint32_t g0 = (uint32_t)(o.v[0] < ta) + (uint32_t)(o.v[1] < ta);
            uint32_t g1 = 0;
            if (n + 1 < N) {
                uint32_t tb = th[(n + 1) % P];
                g1 = (uint32_t)(o.v[2] < tb) + (uint32_t)(o.v[3] < tb);
            }
            w[q >> 1] |= (g0 | (g1 << 8)) << (16 * (q & 1));
        }
    }
    *reinterpret_cast<uint2*>(G + i * ld + n0) = make_uint2(w[0], w[1]);
}

void launch_synth(hipStream_t st, int8_t* G, int64_t M, int64_t N, int64_t ld, int64_t snp_offset, uint64_t seed,
                  const uint32_t* d_thresh, int P) {
    const int64_t total = M * (ld >> 3);
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, G, M, N, ld, snp_offset, seed,
                       d_thresh, P);
}

// Fast generator for streamed panels: thread = 16 consecutive samples of one SNP row = four SplitMix64 outputs; writes 16 int8
// bytes or one 32-bit word of 2-bit dosage codes (no int8 scratch + pack pass).  The 16 samples of a thread belong to ONE
// population -- pop(n) = (n / 16) % P -- so the two 16-bit thresholds are loaded once per thread and a genotype costs two
// sub-dword compares and two carry-adds (the first version, with pop(n) = n % P and a threshold load per sample, spent 80 %
// of its instructions outside Philox: 1.45e12 genotypes/s).  Bit-identical to oracle/gpca_oracle.c:orc_synth16_genotypes.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_synth16(void* __restrict__ Gv, int64_t rows, int64_t N, int64_t ld, int64_t snp0,
                                                  uint64_t seed, const uint32_t* __restrict__ thresh, int P) {
    // grid = (rows, ceil(words per row / 256)): no 64-bit division per thread
    const uint32_t per_row = (uint32_t)(PACKED ? (ld >> 2) : (ld >> 4));
    const int64_t i = blockIdx.x;
    const uint32_t wi = blockIdx.y * 256u + threadIdx.x;    // 16-sample word of the row
    if (wi >= per_row) return;
    const int64_t n0 = (int64_t)wi << 4;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t codes = 0u;
    if (n0 < N) {
        const uint64_t gi = (uint64_t)(i + snp0);
        const uint32_t tw = thresh[i * P + (int)(wi % (uint32_t)P)];
        const uint32_t t1 = tw >> 16, t2 = tw & 0xffffu;
        // four 16-bit uniforms per SplitMix64 output; the thread's four outputs are consecutive in the stream (one multiply,
        // then 64-bit adds).  Philox4x32-10 here cost 170 of the kernel's 300 instructions (20 wide multiplies per 8 genotypes).
        uint64_t st = seed + (((gi << 26) + 4ull * wi) + 1ull) * GPCA_SPLITMIX_GAMMA;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t z = splitmix64_mix(st);
            st += GPCA_SPLITMIX_GAMMA;
            const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = 4 * q + j;
                const uint32_t half = j < 2 ? zl : zh;
                const uint32_t u = (j & 1) ? (half >> 16) : (half & 0xffffu);
                const uint32_t g = (uint32_t)(u < t1) + (uint32_t)(u < t2);
                if (PACKED) codes |= g << (2 * s);
                else w[s >> 2] |= g << (8 * (s & 3));
            }
        }
        if (n0 + 16 > N) {                                   // the row's last, partial word: samples >= N are 0
            const int valid = (int)(N - n0);                 // 1..15
            if (PACKED) codes &= (1u << (2 * valid)) - 1u;
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int v = valid - 4 * q;
                    w[q] = v >= 4 ? w[q] : (v <= 0 ? 0u : (w[q] & ((1u << (8 * v)) - 1u)));
                }
            }
        }
    }
    if (PACKED) *reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(Gv) + i * ld + (n0 >> 2)) = codes;
    else *reinterpret_cast<uint4*>(static_cast<int8_t*>(Gv) + i * ld + n0) = make_uint4(w[0], w[1], w[2], w[3]);
}
void launch_synth16(hipStream_t st, void* G, int packed, int64_t rows, int64_t N, int64_t ld, int64_t snp0, uint64_t seed,
                    const uint32_t* d_thresh16, int P) {
    const int64_t per_row = packed ? (ld >> 2) : (ld >> 4);
    const dim3 grid((unsigned)rows, (unsigned)((per_row + 255) / 256)), blk(256);
    if (packed) hipLaunchKernelGGL(k_synth16<true>, grid, blk, 0, st, G, rows, N, ld, snp0, seed, d_thresh16, P);
    else hipLaunchKernelGGL(k_synth16<false>, grid, blk, 0, st, G, rows, N, ld, snp0, seed, d_thresh16, P);
}

// ------------------------------------------------------------------------------------------------
// PLINK .bed 2-bit -> int8 dosage, count_a1 semantics (prepare.rs:622-629: .i8().count_a1()):
//   code 00 -> 2, 10 -> 1, 11 -> 0, 01 -> missing (-127).  One thread = 4 packed bytes = 16 samples.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bed_decode(const uint8_t* __restrict__ bed, int64_t bpr, int8_t* __restrict__ G,
                                                    int64_t M, int64_t N, int64_t ld) {
    const int64_t per_row = ld >> 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * per_row) return;
    const int64_t i = t / per_row;
    const int64_t n0 = (t - i * per_row) << 4;
    uint32_t packed = 0;
    const uint8_t* src = bed + i * bpr + (n0 >> 2);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((n0 >> 2) + k < bpr) packed |= (uint32_t)src[k] << (8 * k);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const uint32_t code = (packed >> (2 * s)) & 3u;
        // 00->2, 01->0x81, 10->1, 11->0
        uint32_t v = code == 0u ? 2u : (code == 1u ? 0x81u : (code == 2u ? 1u : 0u));
        if (n0 + s >= N) v = 0u;
        w[s >> 2] |= v << (8 * (s & 3));
    }
    *reinterpret_cast<uint4*>(G + i * ld + n0) = make_uint4(w[0], w[1], w[2], w[3]);
}

void launch_bed_decode(hipStream_t st, const uint8_t* bed, int64_t bytes_per_row, int8_t* G, int64_t M, int64_t N,
                       int64_t ld) {
    const int64_t total = M * (ld >> 4);
    hipLaunchKernelGGL(k_bed_decode, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bed, bytes_per_row, G, M, N,
                       ld);
}

// ------------------------------------------------------------------------------------------------
// a1: SNP QC + standardisation parameters.  One wave per SNP row, 16 B per lane per step.
//
// Restates prepare.rs:1216-1375.  Integer sums are exact, so instead of the reference's second f64
// pass the sum of squared deviations is formed as (n*sum(g^2) - sum(g)^2)/n from exact integers
// (< 2^53) with a single rounding -- the exact value the two-pass f64 loop approximates.
// Fast path handles bytes in {0,1,2,-127} with packed-byte logic + v_dot4; any other byte value
// flags the row, which is then recounted byte by byte (signed), keeping prepare.rs:1272-1277's semantics.
// ------------------------------------------------------------------------------------------------
DEVINL double hwe_p_dev(unsigned n1h, unsigned nhet, unsigned n2h) {  // prepare.rs:1641-1745
    const unsigned long long tot = (unsigned long long)n1h + nhet + n2h;
    if (tot == 0) return 1.0;
    const double c1 = 2.0 * (double)n1h + (double)nhet;
    const double c2 = 2.0 * (double)n2h + (double)nhet;
    const double ta = c1 + c2;
    if (ta <= 1e-9) return 1.0;
    const double f1 = c1 / ta, f2 = c2 / ta;
    if (f1 < 1e-9 || f2 < 1e-9) return 1.0;
    if (fabs(f1 + f2 - 1.0) > 1e-6) return 1.0;
    const double e1 = f1 * f1 * (double)tot, eh = 2.0 * f1 * f2 * (double)tot, e2 = f2 * f2 * (double)tot;
    double chi = 0.0;
    const double MINE = 1e-9;
    if (e1 > MINE) { const double d = (double)n1h - e1; chi += d * d / e1; }
    else if ((double)n1h > MINE) chi = INFINITY;
    if (isfinite(chi)) {
        if (eh > MINE) { const double d = (double)nhet - eh; chi += d * d / eh; }
        else if ((double)nhet > MINE) chi = INFINITY;
    }
    if (isfinite(chi)) {
        if (e2 > MINE) { const double d = (double)n2h - e2; chi += d * d / e2; }
        else if ((double)n2h > MINE) chi = INFINITY;
    }
    if (isnan(chi)) return 1.0;
    if (chi == INFINITY) return 0.0;
    const double cdf = erf(sqrt(chi * 0.5));
    if (isnan(cdf)) return 1.0;
    const double p = 1.0 - cdf;
    return p > 0.0 ? p : 0.0;
}

DEVINL int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
DEVINL long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_snp_stats(const int8_t* __restrict__ G, int64_t M, int64_t N, int64_t ld,
                                                    QcParams qc, float* __restrict__ mu, float* __restrict__ sigma,
                                                    float* __restrict__ rr, float* __restrict__ bb,
                                                    uint8_t* __restrict__ keep, uint8_t* __restrict__ reason,
                                                    uint32_t* __restrict__ counts, uint32_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // The row pitch is an odd multiple of 256 B (alloc_genotypes), so rows start 0/256/512/768 B into a KiB: the sweep runs over
    // KiB-ALIGNED wave loads (64 lanes x 16 B = 8 whole lines; unaligned they touch 9 and the pass was 13 % slower) starting at the
    // KiB that holds the row start, and the lanes in front of the row start sit the first trip out.
    const uintptr_t rstart = reinterpret_cast<uintptr_t>(G + row * ld);
    const int64_t skip = (int64_t)((rstart & 1023u) >> 4);                       // 16-byte vectors between the KiB boundary and the row
    const uint4* p = reinterpret_cast<const uint4*>(rstart & ~(uintptr_t)1023u);
    const int64_t nvec = skip + ((N + 15) >> 4);  // the row's samples; bytes between N and the pitch are zero pads
    int nmiss = 0, sum = 0, sq = 0;
    unsigned weird = 0;
    for (int64_t v0 = lane; v0 < nvec; v0 += 64) {
        if (v0 < skip) continue;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + v0);   // one pass, nothing re-read: nt
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned v = w[k];
            const unsigned m7 = v & 0x80808080u;
            nmiss += __builtin_popcount(m7);
            weird |= (v & 0x7C7C7C7Cu) | ((v & (v >> 1)) & 0x01010101u) | ((m7 >> 7) & (~v | (v >> 1)) & 0x01010101u);
            const unsigned vm = v & ~(m7 | (m7 >> 7));
            sum = __builtin_amdgcn_sdot4((int)vm, 0x01010101, sum, false);
            sq = __builtin_amdgcn_sdot4((int)vm, (int)vm, sq, false);
        }
    }
    const unsigned long long anyweird = __ballot(weird != 0);
    long long nv, s1, s2;
    unsigned n0, n1, n2;
    if (anyweird == 0ull) {
        nmiss = wave_sum_i32(nmiss); sum = wave_sum_i32(sum); sq = wave_sum_i32(sq);
        nv = (long long)N - nmiss;  // pad bytes are 0 = valid hom-ref: removed from n0 below
        s1 = sum; s2 = sq;
        n2 = (unsigned)((s2 - s1) >> 1);
        n1 = (unsigned)(s1 - 2 * (long long)n2);
        n0 = (unsigned)(nv - n1 - n2);
    } else {  // exact byte-wise recount, signed values (prepare.rs:1267-1279)
        long long a_nv = 0, a_s1 = 0, a_s2 = 0; int a0 = 0, a1 = 0, a2 = 0;
        const int8_t* rb = G + row * ld;
        for (int64_t n = lane; n < N; n += 64) {
            const int v = rb[n];
            if (v != -127) { a_nv++; a_s1 += v; a_s2 += v * v; a0 += (v == 0); a1 += (v == 1); a2 += (v == 2); }
        }
        nv = wave_sum_i64(a_nv); s1 = wave_sum_i64(a_s1); s2 = wave_sum_i64(a_s2);
        n0 = (unsigned)wave_sum_i32(a0); n1 = (unsigned)wave_sum_i32(a1); n2 = (unsigned)wave_sum_i32(a2);
    }
    if (lane != 0) return;
    counts[4 * row + 0] = (unsigned)nv; counts[4 * row + 1] = n0; counts[4 * row + 2] = n1; counts[4 * row + 3] = n2;
    uint8_t why = 0;
    double mean = 0.0;
    do {
        const double call_rate = (double)nv / (double)N;                       // prepare.rs:1283
        if (call_rate < qc.min_call_rate) { why = 1; break; }
        if (nv == 0) { why = 2; break; }                                       // :1292
        mean = (double)s1 / (double)nv;                                        // :1294
        const double pfr = mean / 2.0;
        const double maf = pfr < 1.0 - pfr ? pfr : 1.0 - pfr;                  // :1296
        if (maf < qc.min_maf) { why = 3; break; }                              // :1299
        if (fabs(pfr) < 1e-9 || fabs(1.0 - pfr) < 1e-9) { why = 4; break; }    // :1302
        if (qc.max_hwe_p < 1.0) {
            if (hwe_p_dev(n0, n1, n2) <= qc.max_hwe_p) { why = 5; break; }     // :1306-1311
        }
    } while (0);
    float m32 = 0.f, s32 = 0.f, r32 = 0.f, b32 = 0.f;
    if (!why) {
        double var = 0.0;
        if (nv >= 2) {
            const double num = (double)nv * (double)s2 - (double)s1 * (double)s1;  // exact integers < 2^53
            var = (num / (double)nv) / (double)(nv - 1);                           // :1358
        }
        if (var <= 1e-9) why = 6;                                                  // :1363
        else {
            m32 = (float)mean;                                                     // :1313
            s32 = (float)sqrt(var);                                                // :1364
            r32 = 1.0f / s32;                                                      // :1948
            b32 = -m32 * r32;                                                      // :1949
            unsigned f = 0;
            if (nv != N) f |= 1u;
            if ((long long)n0 + n1 + n2 != nv) f |= 2u;
            if (f) atomicOr(flags, f);
        }
    }
    mu[row] = m32; sigma[row] = s32; rr[row] = r32; bb[row] = b32;
    keep[row] = why ? 0 : 1; reason[row] = why;
}

void launch_snp_stats(hipStream_t st, const int8_t* G, int64_t M, int64_t N, int64_t ld, QcParams qc, float* mu,
                      float* sigma, float* r, float* b, uint8_t* keep, uint8_t* reason, uint32_t* counts,
                      uint32_t* flags) {
    hipLaunchKernelGGL(k_snp_stats, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, G, M, N, ld, qc, mu, sigma, r, b,
                       keep, reason, counts, flags);
}

__global__ void k_set_scale(int64_t M, const float* mu, const float* sigma, const uint8_t* keep, float* r, float* b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float rv = 0.f, bv = 0.f;
    if (keep[i] && !(fabsf(sigma[i]) < 1e-9f)) { rv = 1.0f / sigma[i]; bv = -mu[i] * rv; }
    r[i] = rv; b[i] = bv;
}
void launch_set_scale(hipStream_t st, int64_t M, const float* mu, const float* sigma, const uint8_t* keep, float* r,
                      float* b) {
    hipLaunchKernelGGL(k_set_scale, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, M, mu, sigma, keep, r, b);
}

// ------------------------------------------------------------------------------------------------
// a2: the pull API, prepare.rs:1884-2016.  Gather (row, col) -> fma((f32)g, 1/sigma, -mu * (1/sigma)).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_standardize_block(const int8_t* __restrict__ G, int64_t ld,
                                                            const float* __restrict__ mu, const float* __restrict__ sigma,
                                                            const int64_t* __restrict__ rows, int64_t ns,
                                                            const int64_t* __restrict__ cols, int64_t nj,
                                                            float* __restrict__ out, unsigned long long* err_idx) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ns * nj) return;
    const int64_t a = t / nj, c = t - a * nj;
    const int64_t i = rows[a];
    const int v = G[i * ld + cols[c]];
    if (v == -127) { atomicMin(err_idx, (unsigned long long)t); return; }   // :1909-1911
    const float sd = sigma[i];
    float o = 0.0f;                                                          // :1899 zero-sigma branch
    if (!(fabsf(sd) < 1e-9f)) {
        const float rs = 1.0f / sd;                                          // :1948
        const float bt = -mu[i] * rs;                                        // :1949
        o = __builtin_fmaf((float)v, rs, bt);                                // :1988 / :2011
    }
    out[t] = o;
}
void launch_standardize_block(hipStream_t st, const int8_t* G, int64_t ld, const float* mu, const float* sigma,
                              const int64_t* rows, int64_t ns, const int64_t* cols, int64_t nj, float* out,
                              unsigned long long* err_idx) {
    const int64_t total = ns * nj;
    hipLaunchKernelGGL(k_standardize_block, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, G, ld, mu, sigma, rows,
                       ns, cols, nj, out, err_idx);
}

// ------------------------------------------------------------------------------------------------
// Sketch operand.  Omega[i][j] ~ N(0,1): Philox4x32-10, counter (i_lo, i_hi, j/4, stream), two
// Box-Muller pairs in f64 (bit-compatible recipe with oracle/gpca_oracle.c:omega4).
// One wave = 64 rows; writes Tp = r o Omega and the wave's partial of c = b^T Omega.
// ------------------------------------------------------------------------------------------------
int64_t omega_num_parts(int64_t Mpad) { return (Mpad + 63) / 64; }
// sketches wider than 64 columns: the plain any-L kernels of wide_sketch.hip
void launch_gram_any_f64(hipStream_t st, const double* X, int64_t rows, int64_t rpb, int64_t parts, int L, double* part);
void launch_gram_any_f32(hipStream_t st, const float* X, int64_t rows, int64_t rpb, int64_t parts, int L, double* part);
int launch_chol_inv_any(hipStream_t st, const double* W, int n, int L, double* Z, int* flag);
void launch_apply_right_any(hipStream_t st, double* X, int64_t rows, int64_t parts, int L, const double* Z, double* csum_part, double* amax_part);
void launch_rightmul_any_f64(hipStream_t st, const double* X, int64_t rows, int L, const double* Z, int K, double* out64, float* out32);
void launch_rightmul_any_gather_f32(hipStream_t st, const float* X, const int64_t* row_ids, int64_t nrows, int L, const double* Z, int K, float* out32);

// LP = LDS pitch of the staged tile: L + 1 (33 for a 32-column sketch -- half the LDS of the 64-column form, so four waves per SIMD
// stay resident beside the f64 log / sincospi chains instead of two; 90 VGPRs would allow five).
template <int LP, int NW = 2>
__global__ __launch_bounds__(64 * NW) void k_omega(int64_t M, int64_t Mpad, int l, int L, int64_t snp_offset, uint64_t seed,
                                               const float* __restrict__ r, const float* __restrict__ b,
                                               float* __restrict__ Tb, float* __restrict__ cpart, double* __restrict__ apart,
                                               int blocked, int8_t* __restrict__ Td, const float* __restrict__ rmax,
                                               double* __restrict__ tscale, double* __restrict__ tinv, int nd,
                                               const int64_t* __restrict__ row_ids) {
    // Each lane draws the L normals of its own SNP row; the wave's 64 x L tile is staged in LDS (pitch L + 1) so that
    //   * T' = r o Omega leaves as full rows, lane-contiguous (a lane writing its row 4 bytes at a time at a 128-byte
    //     stride cost 4.3x write amplification), and
    //   * the wave's partial of c = b^T Omega is a conflict-free column walk instead of 6 cross-lane steps per column.
    __shared__ float zt[NW][64][LP];
    __shared__ float rs[NW][64], bs[NW][64];
    __shared__ OmegaLnEntry lntab[kOmegaLnEntries];      // (omega_math.h: the ln table, one 16-byte LDS read per draw)
    for (int e = threadIdx.x; e < kOmegaLnEntries; e += 64 * NW) lntab[e] = kOmegaLnTable[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * NW + wv;
    const int64_t i0 = wave * 64, i = i0 + lane;
    const bool live = i < M;
    rs[wv][lane] = live ? r[i] : 0.f;
    bs[wv][lane] = live ? b[i] : 0.f;
    // the row's GLOBAL index keys its normals: a matrix of gathered rows (the kept SNPs of a larger one, row_ids) draws what they
    // would draw in place
    const uint64_t gi = (uint64_t)(((row_ids && live) ? row_ids[i] : i) + snp_offset);
    const int LT = L >> 5;
    for (int jq = 0; jq < L / 4; ++jq) {
        double z[4] = {0, 0, 0, 0};
        if (4 * jq < l && live) {
#if GPCA_OMEGA_ABLATE & 2
            philox_out o; o.v[0] = (uint32_t)gi * 2654435761u + jq; o.v[1] = o.v[0] ^ 0x9e3779b9u; o.v[2] = o.v[0] * 3u; o.v[3] = ~o.v[0];
#else
            philox_out o = philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)jq, GPCA_STREAM_OMEGA,
                                         (uint32_t)seed, (uint32_t)(seed >> 32));
#endif
#if GPCA_OMEGA_ABLATE & 1
            z[0] = o.v[0]; z[1] = o.v[1]; z[2] = o.v[2]; z[3] = o.v[3];
#else
            // Box-Muller on the 32-bit uniforms (u = (v + 1) 2^-32), the oracle's recipe (gpca_oracle.c:omega4) through transcendentals
            // written for a 33-bit integer argument (omega_math.h): within 2e-16 of long-double libm, a third of the device library's cost
            omg_box_muller(o.v[0], o.v[1], lntab, z[0], z[1]);
            omg_box_muller(o.v[2], o.v[3], lntab, z[2], z[3]);
#endif
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float zf = (4 * jq + t < l) ? (float)z[t] : 0.f;
            zt[wv][lane][4 * jq + t] = zf;
            // the blocked operand layout of the f32 path interleaves rows: its lane-per-row stores are already contiguous
            if (blocked && Tb && i < Mpad) Tb[blocked_t_index(i, 4 * jq + t, LT)] = rs[wv][lane] * zf;
        }
    }
    __syncthreads();
    if (Td && !(GPCA_OMEGA_ABLATE & 4)) {
        // Exact-integer path: the digit planes of T' = r o Omega leave this kernel directly (no f32 T', no quantisation pass over
        // it).  The column scale is an analytic bound instead of the measured maximum: |z| <= sqrt(-2 ln 2^-32) = 6.6604 for the
        // Box-Muller draw above and r <= rmax, so |T'| <= 6.67 rmax -- at M = 10^6 rows that is within a factor ~2 of the measured
        // maximum (one bit of the 28), and the same for every column, shard and panel.
        const double bound = 6.67 * (double)rmax[0];
        const double S = nd == 3 ? kDigitScale3 : kDigitScale;
        const double inv = bound > 0.0 ? S / bound : 0.0;
        if (blockIdx.x == 0)
            for (int cj = threadIdx.x; cj < L; cj += 64 * NW) {
                tscale[cj] = (cj < l && bound > 0.0) ? bound / S : 0.0;
                tinv[cj] = (cj < l) ? inv : 0.0;
            }
        const int cc = lane & 31, hh = lane >> 5;
        for (int hf = 0; hf < LT; ++hf)
            for (int b2 = 0; b2 < 2; ++b2) {
                const int64_t blk = (i0 >> 5) + b2;
                if (blk * 32 >= Mpad) break;
                unsigned w[kDigits][4];
#pragma unroll
                for (int d = 0; d < kDigits; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[d][q] = 0u;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = 32 * b2 + 16 * hh + j;
                    const float t = __fmul_rn(rs[wv][row], zt[wv][row][32 * hf + cc]);
                    int v = __double2int_rn((double)t * inv);
#pragma unroll
                    for (int d = 0; d < kDigits; ++d) {
                        int dg;
                        if (nd == 3) { if (d < 2) { dg = ((v + 128) & 255) - 128; v = (v - dg) >> 8; } else { dg = v; v = 0; } }
                        else if (d < kDigits - 1) { dg = ((v + 64) & 127) - 64; v = (v - dg) >> 7; } else dg = v;
                        w[d][j >> 2] |= ((unsigned)(dg & 0xff)) << (8 * (j & 3));
                    }
                }
                int8_t* out = Td + (size_t)hf * (size_t)Mpad * 32 * kDigits;
#pragma unroll
                for (int d = 0; d < kDigits; ++d)
                    *reinterpret_cast<uint4*>(out + ((blk * kDigits + d) * 64 + lane) * 16) = make_uint4(w[d][0], w[d][1], w[d][2], w[d][3]);
            }
    }
    // T' rows: element e = row * L + col of the tile, lanes take consecutive elements
    if (!blocked && Tb)
        for (int e = lane; e < 64 * L; e += 64) {
            const int row = e / L, col = e % L;
            const int64_t gr = i0 + row;
            if (gr < Mpad) Tb[gr * L + col] = rs[wv][row] * zt[wv][row][col];
        }
    // c partial: lane = column (two half-waves split the rows for L = 32), rows in a fixed order
    if (wave < (Mpad + 63) / 64) {
        if (L == 32) {
            const int col = lane & 31, hh = lane >> 5;
            float cv = 0.f, am = 0.f;
            for (int row = 32 * hh; row < 32 * hh + 32; ++row) {
                const float zv = zt[wv][row][col];
                cv += bs[wv][row] * zv;
                am = fmaxf(am, fabsf(rs[wv][row] * zv));
            }
            cv += __shfl_xor(cv, 32);
            am = fmaxf(am, __shfl_xor(am, 32));
            if (hh == 0) {
                cpart[wave * L + col] = cv;
                // |T'| column max (digit scale of the first exact product): one 32-entry array for the whole grid.  max is
                // order-independent, so the atomic keeps the result deterministic; non-negative doubles order like their
                // bit patterns; the plain pre-read skips the atomic once the running max has settled (it only grows).
                if (apart) {
                    const double amd = (double)am;
                    if (amd > apart[col]) atomicMax(reinterpret_cast<unsigned long long*>(apart) + col, (unsigned long long)__double_as_longlong(amd));
                }
            }
        } else {
            for (int col = lane; col < L; col += 64) {
                float cv = 0.f;
                for (int row = 0; row < 64; ++row) cv += bs[wv][row] * zt[wv][row][col];
                cpart[wave * L + col] = cv;
            }
        }
    }
}

void launch_omega(hipStream_t st, int64_t M, int64_t Mpad, int l, int L, int64_t snp_offset, uint64_t seed, const float* r,
                  const float* b, float* Tb, float* cpart, double* apart, int blocked, const int64_t* row_ids) {
    const int64_t waves = omega_num_parts(Mpad);
    if (L == 32) hipLaunchKernelGGL(k_omega<33>, dim3((unsigned)((waves + 1) / 2)), dim3(128), 0, st, M, Mpad, l, L, snp_offset, seed, r, b, Tb,
                                    cpart, apart, blocked, (int8_t*)nullptr, (const float*)nullptr, (double*)nullptr, (double*)nullptr, 4, row_ids);
    else hipLaunchKernelGGL(k_omega<65>, dim3((unsigned)((waves + 1) / 2)), dim3(128), 0, st, M, Mpad, l, L, snp_offset, seed, r, b, Tb,
                            cpart, apart, blocked, (int8_t*)nullptr, (const float*)nullptr, (double*)nullptr, (double*)nullptr, 4, row_ids);      // (f32 path: L <= 64)
}
// Exact-integer path: T' = r o Omega straight into digit planes Td ([L/32 halves][Mpad/32][kDigits][64][16 B]) against the analytic
// column bound 6.67 * rmax; tscale / tinv [L] receive the scale (columns >= l: 0); cpart as above.  No f32 copy of T'.
void launch_omega_planes(hipStream_t st, int64_t M, int64_t Mpad, int l, int L, int64_t snp_offset, uint64_t seed, const float* r,
                         const float* b, float* cpart, int8_t* Td, const float* rmax, double* tscale, double* tinv, int nd,
                         const int64_t* row_ids) {
    const int64_t waves = omega_num_parts(Mpad);
    if (L == 32) hipLaunchKernelGGL(k_omega<33>, dim3((unsigned)((waves + 1) / 2)), dim3(128), 0, st, M, Mpad, l, L, snp_offset, seed, r, b, (float*)nullptr,
                                    cpart, (double*)nullptr, 0, Td, rmax, tscale, tinv, nd, row_ids);
    else if (L == 64) hipLaunchKernelGGL(k_omega<65>, dim3((unsigned)((waves + 1) / 2)), dim3(128), 0, st, M, Mpad, l, L, snp_offset, seed, r, b, (float*)nullptr,
                                         cpart, (double*)nullptr, 0, Td, rmax, tscale, tinv, nd, row_ids);
    else hipLaunchKernelGGL((k_omega<129, 1>), dim3((unsigned)waves), dim3(64), 0, st, M, Mpad, l, L, snp_offset, seed, r, b, (float*)nullptr,
                            cpart, (double*)nullptr, 0, Td, rmax, tscale, tinv, nd, row_ids);        // L = 128 (one wave per workgroup: 33 KB of LDS for its tile)
}
// dst row i <- src row ids[i] (rows of `pitch` bytes, a multiple of 16): the kept SNPs of a matrix gathered into one of their own
__global__ __launch_bounds__(256) void k_gather_rows(const uint8_t* __restrict__ src, int64_t pitch, const int64_t* __restrict__ ids,
                                                     int64_t n, uint8_t* __restrict__ dst) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)ids[i] * (size_t)pitch);
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)i * (size_t)pitch);
    for (int64_t v = threadIdx.x; v < pitch / 16; v += 256) d[v] = s[v];
}
void launch_gather_rows(hipStream_t st, const void* src, int64_t pitch, const int64_t* ids, int64_t n, void* dst) {
    if (n > 0) hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n), dim3(256), 0, st, (const uint8_t*)src, pitch, ids, n, (uint8_t*)dst);
}
// dst[i] <- src[ids[i]] for elements of `eb` bytes (1, 4 or 16)
__global__ __launch_bounds__(256) void k_gather_elems(const uint8_t* __restrict__ src, int eb, const int64_t* __restrict__ ids, int64_t n,
                                                      uint8_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t j = ids[i];
    if (eb == 4) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[j];
    else if (eb == 16) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[j];
    else dst[i] = src[j];
}
void launch_gather_elems(hipStream_t st, const void* src, int elem_bytes, const int64_t* ids, int64_t n, void* dst) {
    if (n > 0) hipLaunchKernelGGL(k_gather_elems, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t*)src, elem_bytes, ids, n, (uint8_t*)dst);
}
// rmax[0] = max_i r[i] (r >= 0: non-negative floats order like their bit patterns); the caller zeroes rmax first
__global__ __launch_bounds__(256) void k_max_f32(const float* __restrict__ r, int64_t n, float* __restrict__ out) {
    __shared__ float red[256];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a = fmaxf(a, r[i]);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) { if ((int)threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]); __syncthreads(); }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(red[0]));
}
void launch_max_f32(hipStream_t st, const float* r, int64_t n, float* out) {
    int64_t blocks = (n + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(k_max_f32, dim3((unsigned)blocks), dim3(256), 0, st, r, n, out);
}

__global__ __launch_bounds__(256) void k_reduce_y(const float* __restrict__ Ypart, int W, int64_t Npad, int64_t N, int L,
                                                  const double* __restrict__ cvec, double* __restrict__ Y) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * L) return;
    const int j = (int)(e % L);
    double a = cvec[j];
    const int64_t stride = Npad * L;
    double sacc = 0.0;
    for (int w = 0; w < W; ++w) sacc += (double)Ypart[w * stride + e];
    Y[e] = a + 512.0 * sacc;   // 2^9 of the fp8-subnormal dosage trick (gemm_f32.hip), exact
}
void launch_reduce_y(hipStream_t st, const float* Ypart, int W, int64_t Npad, int64_t N, int L, const double* c, double* Y) {
    const int64_t total = N * L;
    hipLaunchKernelGGL(k_reduce_y, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Ypart, W, Npad, N, L, c, Y);
}

// ------------------------------------------------------------------------------------------------
// Small helpers (HBM/latency bound, negligible next to K1/K2).
// ------------------------------------------------------------------------------------------------
// out[e] = sum_p part[p][e] in f64, deterministic (fixed summation tree), two stages so that thousands of
// parts do not serialise behind one block's load latency: stage 1 = S slices of the part axis, stage 2 = S -> 1.
template <typename T>
__global__ __launch_bounds__(256) void k_sum_partials(const T* __restrict__ part, int64_t P, int64_t E,
                                                      double* __restrict__ out, int S) {
    __shared__ double red[4][64];
    sum_partials_body<T>(part, P, E, out, S, blockIdx.x, blockIdx.y, red);
}
// slices of the part axis in stage 1.  Few elements per part (E <= 64: the c = b^T T partials, one per 32-row unit, 31 250 of them at
// a million SNPs) put a single column of workgroups on the grid: up to 256 slices there instead of 64 (14 us -> ~5 us for 4 MB).
// S depends on (P, E) only, so the summation tree -- and with it every bit of the result -- is the same for every run and partition.
int sum_slices(int64_t P, int64_t E) {
    const int64_t cap = E <= 64 ? 256 : 64;
    int64_t s = (P + 63) / 64;
    return (int)(s < 1 ? 1 : (s > cap ? cap : s));
}
template <typename T>
static void launch_sum_partials_t(hipStream_t st, const T* part, int64_t P, int64_t E, double* out, double* scratch) {
    const int S = sum_slices(P, E);
    const dim3 blk(256);
    if (S == 1) {
        hipLaunchKernelGGL((k_sum_partials<T>), dim3((unsigned)((E + 63) / 64), 1), blk, 0, st, part, P, E, out, 1);
    } else {
        hipLaunchKernelGGL((k_sum_partials<T>), dim3((unsigned)((E + 63) / 64), S), blk, 0, st, part, P, E, scratch, S);
        hipLaunchKernelGGL((k_sum_partials<double>), dim3((unsigned)((E + 63) / 64), 1), blk, 0, st, (const double*)scratch,
                           (int64_t)S, E, out, 1);
    }
}
void launch_sum_partials_f32(hipStream_t st, const float* part, int64_t P, int64_t E, double* out, double* scratch) {
    launch_sum_partials_t<float>(st, part, P, E, out, scratch);
}
void launch_sum_partials_f64(hipStream_t st, const double* part, int64_t P, int64_t E, double* out, double* scratch) {
    launch_sum_partials_t<double>(st, part, P, E, out, scratch);
}
void launch_sum_partials_f64_stage1(hipStream_t st, const double* part, int64_t P, int64_t E, double* scratch, const double** src, int* slices) {
    const int S = sum_slices(P, E);
    if (S == 1) { *src = part; *slices = (int)P; return; }      // (P <= 64: the parts are the slices)
    hipLaunchKernelGGL((k_sum_partials<double>), dim3((unsigned)((E + 63) / 64), S), dim3(256), 0, st, part, P, E, scratch, S);
    *src = scratch; *slices = S;
}

// Gram: part[blk][a][c] = sum over the block's rows of X[n][a] X[n][c]  (f64 accumulate, k_gram_mfma below).
// Rows per block adapt to the problem so that ~1024 blocks are in flight (N = 10^4 used to get 20 blocks).
static int64_t gram_rows_per_block(int64_t rows) {
    // The sample-side Grams of CholeskyQR (N rows, four per call) sit on the critical path between two GEMM passes, and their partial
    // sums are folded by the ONE workgroup that factors the result (k_chol_inv_fold32): 32 parts up to 8k rows, rising to at most 64, which
    // one k_sum_partials launch still finishes (the matrix-core Gram of 10 000 x 32 takes 5 us with 63 workgroups, 11 with 16).
    if (rows <= 262144) {
        int64_t parts = rows / 256;
        parts = parts < 32 ? 32 : (parts > 64 ? 64 : parts);
        const int64_t q = (rows + parts - 1) / parts;
        return q < 32 ? 32 : (q + 31) / 32 * 32;
    }
    int64_t r = (rows + 1023) / 1024;
    r = (r + 31) / 32 * 32;
    return r < 32 ? 32 : (r > 2048 ? 2048 : r);
}
int64_t gram_num_parts(int64_t rows) { const int64_t rpb = gram_rows_per_block(rows); return (rows + rpb - 1) / rpb; }

// Gram of a tall factor (f32: B = A Q, M rows; f64: the sample-side sketch, N rows) on the f64 matrix cores: W = X^T X as 16x16x4 MFMAs.  Lane (i = lane & 15,
// k = lane >> 4) converts X[n + k][16 g + i] once and uses it both as the A element (X^T tile g) and as the B element
// (X tile g); only the upper-triangular tiles are computed, the lower ones are mirrored on store.  Four waves take
// interleaved 4-row groups and are combined through LDS in a fixed order.  (The VALU version spent 132 us on the
// 128 MB factor -- LDS-read bound; this one is HBM-bound.)
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename TX, int L>
__global__ __launch_bounds__(256) void k_gram_mfma(const TX* __restrict__ X, int64_t rows, int64_t rpb,
                                                   double* __restrict__ part) {
    constexpr int G = L / 16, NT = G * (G + 1) / 2;
    __shared__ double red[3][NT][64][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, k = lane >> 4;
    f64x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = (r0 + rpb < rows) ? r0 + rpb : rows;
    // 16 rows per wave and trip (4 MFMA k-steps); the next trip's rows are requested before this trip's MFMAs.  (Three trips ahead
    // measured the same 41.7 us at a million rows: half of that is the f64 matrix pipe itself -- 750k 16x16x4 MFMAs at 64 cycles.)
    TX xc[4][G], xn[4][G];
    auto load_rows = [&](int64_t n, TX (&dst)[4][G]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = n + 4 * u + k;
#pragma unroll
            for (int g = 0; g < G; ++g) dst[u][g] = row < r1 ? X[row * L + 16 * g + i] : (TX)0;
        }
    };
    load_rows(r0 + 16 * wv, xc);
    for (int64_t n = r0 + 16 * wv; n < r1; n += 64) {
        load_rows(n + 64, xn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double x[G];
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = (double)xc[u][g];
            int t = 0;
#pragma unroll
            for (int ga = 0; ga < G; ++ga)
#pragma unroll
                for (int gc = ga; gc < G; ++gc, ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ga], x[gc], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) xc[u][g] = xn[u][g];
    }
    if (wv > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv - 1][t][lane][r] = acc[t][r];
    }
    __syncthreads();
    if (wv != 0) return;
    double* out = part + (int64_t)blockIdx.x * L * L;
    int t = 0;
#pragma unroll
    for (int ga = 0; ga < G; ++ga)
#pragma unroll
        for (int gc = ga; gc < G; ++gc, ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = ((acc[t][r] + red[0][t][lane][r]) + red[1][t][lane][r]) + red[2][t][lane][r];
                const int a = 16 * ga + 4 * r + k, c = 16 * gc + i;     // f64 16x16x4: D[4 r + lane / 16][lane % 16] (probe_mfma_f64.hip)
                out[a * L + c] = v;
                if (ga != gc) out[c * L + a] = v;
            }
}
void launch_gram_f64(hipStream_t st, const double* X, int64_t rows, int L, double* part) {
    const int64_t rpb = gram_rows_per_block(rows);
    const dim3 grid((unsigned)gram_num_parts(rows)), blk(256);
    // the f64 matrix cores (v_mfma_f64_16x16x4), as for the f32 factor below: the VALU kernel k_gram<double, L> (an LDS row tile, 8 reads per 16
    // FMAs) took 12.5 us for 10 000 x 32 -- four of them sit between the GEMM sweeps of a call
    if (L == 32) hipLaunchKernelGGL((k_gram_mfma<double, 32>), grid, blk, 0, st, X, rows, rpb, part);
    else if (L == 64) hipLaunchKernelGGL((k_gram_mfma<double, 64>), grid, blk, 0, st, X, rows, rpb, part);
    else launch_gram_any_f64(st, X, rows, rpb, gram_num_parts(rows), L, part);
}
void launch_gram_f32(hipStream_t st, const float* X, int64_t rows, int L, double* part) {
    const int64_t rpb = gram_rows_per_block(rows);
    const dim3 grid((unsigned)gram_num_parts(rows)), blk(256);
    if (L == 32) hipLaunchKernelGGL((k_gram_mfma<float, 32>), grid, blk, 0, st, X, rows, rpb, part);
    else if (L == 64) hipLaunchKernelGGL((k_gram_mfma<float, 64>), grid, blk, 0, st, X, rows, rpb, part);
    else launch_gram_any_f32(st, X, rows, rpb, gram_num_parts(rows), L, part);
}

// X[n][:] <- X[n][:] Z, in place through an LDS row tile; optional f32 copy into Qout (pad rows zeroed)
template <int L>
__global__ __launch_bounds__(256) void k_apply_right(double* __restrict__ X, int64_t rows, const double* __restrict__ Z,
                                                     float* __restrict__ Qout, int64_t rows_pad) {
    constexpr int RPB = 256 / L;
    __shared__ double zs[L][L + 1];
    __shared__ double xs[RPB][L];
    for (int e = threadIdx.x; e < L * L; e += 256) zs[e / L][e % L] = Z[e];
    const int rr = threadIdx.x / L, cc = threadIdx.x % L;
    const int64_t n = (int64_t)blockIdx.x * RPB + rr;
    xs[rr][cc] = (n < rows) ? X[n * L + cc] : 0.0;
    __syncthreads();
    double a = 0.0;
#pragma unroll 8
    for (int j = 0; j < L; ++j) a += xs[rr][j] * zs[j][cc];
    if (n < rows) X[n * L + cc] = a;
    if (Qout && n < rows_pad) Qout[blocked_q_index(n, cc, L >> 5)] = (n < rows) ? (float)a : 0.f;
}
// Last right-multiplication of CholeskyQR2, fused with everything the next stage needs from the orthonormal basis:
// X <- X Z in place (f64), the f32 blocked copy Qb (pad rows zeroed), and per-workgroup partials of the column sums
// (s = Q^T 1, the centring term of A Q) and of the column abs-max (digit scale of the exact-integer path).  A workgroup
// walks kTailRows rows so that Z is staged once per 64 rows and the partial arrays stay small; k_finish_q reduces them
// in a fixed order.  Replaces k_colsum + 2 x k_sum_partials + k_f64_to_f32 + k_col_absmax + k_finish_scale.
constexpr int kTailRows = 64;
int64_t tail_num_parts(int64_t rows_pad) { return (rows_pad + kTailRows - 1) / kTailRows; }
template <int L>
__global__ __launch_bounds__(256) void k_apply_right_tail(double* __restrict__ X, int64_t rows, const double* __restrict__ Z,
                                                          float* __restrict__ Qout, int64_t rows_pad,
                                                          double* __restrict__ csum_part, double* __restrict__ amax_part) {
    constexpr int RPB = 256 / L;
    __shared__ double zs[L][L + 1];
    __shared__ double xs[RPB][L];
    __shared__ double red[2][RPB][L];
    for (int e = threadIdx.x; e < L * L; e += 256) zs[e / L][e % L] = Z[e];
    const int rr = threadIdx.x / L, cc = threadIdx.x % L;
    double cs = 0.0, am = 0.0;
    const int64_t n0 = (int64_t)blockIdx.x * kTailRows;
    for (int sub = 0; sub < kTailRows; sub += RPB) {
        const int64_t n = n0 + sub + rr;
        __syncthreads();                                   // zs ready (first trip) / xs free (later trips)
        xs[rr][cc] = (n < rows) ? X[n * L + cc] : 0.0;
        __syncthreads();
        double a = 0.0;
#pragma unroll 8
        for (int j = 0; j < L; ++j) a += xs[rr][j] * zs[j][cc];
        if (n < rows) { X[n * L + cc] = a; cs += a; am = fmax(am, fabs(a)); }
        if (Qout && n < rows_pad) Qout[blocked_q_index(n, cc, L >> 5)] = (n < rows) ? (float)a : 0.f;
    }
    red[0][rr][cc] = cs; red[1][rr][cc] = am;
    __syncthreads();
    if (rr == 0) {
        for (int g = 1; g < RPB; ++g) { cs += red[0][g][cc]; am = fmax(am, red[1][g][cc]); }
        csum_part[(int64_t)blockIdx.x * L + cc] = cs;
        amax_part[(int64_t)blockIdx.x * L + cc] = am;
    }
}
void launch_apply_right_tail(hipStream_t st, double* X, int64_t rows, int L, const double* Z, float* Qout, int64_t rows_pad,
                             double* csum_part, double* amax_part) {
    const dim3 grid((unsigned)tail_num_parts(rows_pad)), blk(256);
    if (L == 32) hipLaunchKernelGGL((k_apply_right_tail<32>), grid, blk, 0, st, X, rows, Z, Qout, rows_pad, csum_part, amax_part);
    else if (L == 64) hipLaunchKernelGGL((k_apply_right_tail<64>), grid, blk, 0, st, X, rows, Z, Qout, rows_pad, csum_part, amax_part);
    else launch_apply_right_any(st, X, rows, tail_num_parts(rows_pad), L, Z, csum_part, amax_part);      // (exact path only: no f32 copy of the basis)
}
// s64/s32[c] = sum_p csum_part[p][c] (fixed order);  digit scale of column c from max_p amax_part[p][c]:
// scale = max / S, inv = S / max (0 for an all-zero column), S = kDigitScale.  One workgroup.
__global__ __launch_bounds__(1024) void k_finish_q(const double* __restrict__ csum_part, const double* __restrict__ amax_part,
                                                   int64_t P, int L, double* __restrict__ s64, float* __restrict__ s32,
                                                   double* __restrict__ scale, double* __restrict__ inv, double S) {
    __shared__ double rs[1024], rm[1024];
    const int cc = threadIdx.x % L, pg = threadIdx.x / L, G = 1024 / L;
    double a = 0.0, m = 0.0;
    for (int64_t p = pg; p < P; p += G) { a += csum_part[p * L + cc]; m = fmax(m, amax_part[p * L + cc]); }
    rs[threadIdx.x] = a; rm[threadIdx.x] = m;
    __syncthreads();
    if (pg != 0) return;
    for (int g = 1; g < G; ++g) { a += rs[g * L + cc]; m = fmax(m, rm[g * L + cc]); }
    s64[cc] = a; s32[cc] = (float)a;
    if (scale) { scale[cc] = m > 0.0 ? m / S : 0.0; inv[cc] = m > 0.0 ? S / m : 0.0; }
}
void launch_finish_q(hipStream_t st, const double* csum_part, const double* amax_part, int64_t P, int L, double* s64, float* s32,
                     double* scale, double* inv, int nd) {
    hipLaunchKernelGGL(k_finish_q, dim3(1), dim3(1024), 0, st, csum_part, amax_part, P, L, s64, s32, scale, inv, digit_scale(nd));
}

// CholeskyQR's small factorisation on the device: W (n x n, pitch NN, upper triangle used) = R^T R, Z = R^-1 (upper,
// zero elsewhere, the whole NN x NN block written), so that no host round trip (and no stream sync) sits between the
// Gram matrix and the right-multiplication.  ONE WAVE, the matrix in registers: lane c owns column c of R and of R^-1.
//   * Cholesky, right-looking: step j scales row j by 1/sqrt(pivot) and subtracts its outer product from the trailing
//     rows (each element sees the same subtractions, in the same order, as a row-by-row Cholesky-Crout).
//   * R^-1: lane c back-substitutes R x = e_c from the bottom row up; x[k] = 0 for k > c falls out by itself.
// Rows / columns n..NN-1 are treated as identity.  A pivot that is not finite records (j + 1) in *flag (first failure wins)
// and the factorisation carries on with pivot 1, so nothing downstream spins or faults; the caller checks the flag
// once, at the end of the rSVD.  A pivot that is zero to rounding drops its column from the basis (see the step below).
constexpr double kCholRankTol = 1e-13;   // relative to the column's own squared norm (Gram rounding is ~32 x 2.2e-16)
// 1 / sqrt(x) in f64 from the hardware estimate and two Newton steps (the correctly rounded sqrt + divide pair costs
// ~500 dependent cycles per pivot; this chain ~100)
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
// How an element of another column reaches a lane.  Rounds 1-3: one v_readlane pair per element (~1 000 of them for NN = 32, each
// with its hazard nops) -- 18.9 us per launch (scripts/kbench/kbench_chol.hip, both forms alternating in one process).  Round 4: a finished row of R is broadcast through LDS: step j writes its scaled row once
// (ds_write_b64, lane c -> rs[j][c]) and every lane reads the entries it needs back from a wave-uniform address (a broadcast read,
// no bank conflict), one batch of wide reads per step; the back substitution reads the same rows again.  Same operations on the
// same values in the same order, so the bits do not change (the harness compares the results; scripts/fingerprint.py before / after) -- 12.5 us (the chain of a
// step: pivot, 1/sqrt by Newton, the row through LDS, 31 - j dependent-free FMAs; what is left is one wave's f64 latency).
// The asm pins keep a step's FMAs in the step: left free, the scheduler sinks them behind the last row's reads with every row live.
// (Dropped after measurement: the whole matrix in LDS (round 3, 100 us); the fold of the Gram partials ahead of the factorisation
//  (round 3, 45 us against 5 + 23); Gram + fold + factorisation in one launch, the last workgroup to finish doing the small work
//  (round 4: 58 us against 12.5 + 4.7 + 17 -- one CU folds 63 partials of 8 KiB more slowly than 16 do).)
__device__ __forceinline__ double bcast_lane(double v, int lane) {      // lane: a constant once the loop around the call is unrolled
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// One wave (lanes c = 0..63); Wg may be global or LDS; rs = NN x NN doubles of LDS (rs[j][c] = R[j][c], row j final after step j).
// wave_sync(): orders this wave's LDS write of a row before its reads of it (one wave's LDS operations execute in order: the
// compiler's fence and wait, no barrier needed -- a block-wide barrier would also hang the fused caller, whose other waves have left).
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
template <int NN>
__device__ __forceinline__ void chol_inv_wave(const double* Wg, int n, double* __restrict__ Zg, int* __restrict__ flag, double* rs, int c) {
    double col[NN], x[NN], dinv[NN];
#pragma unroll
    for (int r = 0; r < NN; ++r) col[r] = Wg[r * NN + (c & (NN - 1))];      // (unconditional: the loads stay in flight together)
#pragma unroll
    for (int r = 0; r < NN; ++r) col[r] = (r < n && c < n) ? col[r] : ((r == c) ? 1.0 : 0.0);
    double diag0 = 0.0;
#pragma unroll
    for (int r = 0; r < NN; ++r) diag0 = (r == c) ? col[r] : diag0;
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        __builtin_amdgcn_sched_barrier(0);      // (a step's row reads stay in the step: hoisted, they fill the register file and spill)
        // pivot = R[j][j] so far (lane j's col[j]); d0 = column j's own squared norm before the elimination
        double piv = bcast_lane(col[j], j);
        const double d0 = bcast_lane(diag0, j);
        if (!isfinite(piv) || !isfinite(d0)) {
            if (c == 0) atomicCAS(flag, 0, j + 1);
            piv = 1.0;
        }
        // Column j lies in the span of the columns before it (what is left of its squared norm is rounding noise, possibly
        // negative): a sketch wider than the rank of the matrix -- k + oversample = N samples of centred rows have rank N - 1.
        // The column leaves the basis: row j of R and of R^-1 become zero, so Q's column j is zero and every later product
        // carries a zero column (zero singular value) instead of the call failing.
        const bool dependent = !(piv > kCholRankTol * d0);
        dinv[j] = dependent ? 0.0 : rsqrt_nr(piv);
        col[j] = (c == j) ? piv * dinv[j] : col[j] * dinv[j];
        if (c < NN) rs[j * NN + c] = col[j];
        wave_lds_sync();
        double row[NN];                         // the row, read in one batch of wide LDS reads ahead of the FMAs
#pragma unroll
        for (int r = j + 1; r < NN; ++r) row[r] = rs[j * NN + r];
#pragma unroll
        for (int r = j + 1; r < NN; ++r) asm volatile("" : "+v"(row[r]));
#pragma unroll
        for (int r = j + 1; r < NN; ++r) { col[r] -= row[r] * col[j]; asm volatile("" : "+v"(col[r])); }   // (pinned to its step, as x[i] below)
    }
    // R^-1: lane c back-substitutes R x = e_c from the bottom row up
#pragma unroll
    for (int i = NN - 1; i >= 0; --i) {
        __builtin_amdgcn_sched_barrier(0);
        double acc = (c == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = i + 1; k < NN; ++k) acc -= rs[i * NN + k] * x[k];
        x[i] = acc * dinv[i];
        asm volatile("" : "+v"(x[i]));          // (the step's FMAs stay in the step: left free they all sink behind the last row's reads, every row live)
    }
    if (c < NN) {
#pragma unroll
        for (int i = 0; i < NN; ++i) Zg[i * NN + c] = (i < n && c < n) ? x[i] : 0.0;
    }
}
template <int NN>
__global__ __launch_bounds__(64) void k_chol_inv(const double* __restrict__ Wg, int n, double* __restrict__ Zg,
                                                 int* __restrict__ flag) {
    __shared__ double rs[NN * NN];
    chol_inv_wave<NN>(Wg, n, Zg, flag, rs, (int)threadIdx.x);
}
// The same with the fold of the Gram's partial sums in front (k_sum_partials was a launch of its own between k_gram and this one, four
// times a call): 256 threads sum part[p][e] over the <= 64 parts, four elements and sixteen parts of each in flight per thread, into the
// LDS block the factorisation then reads -- and reuses for its rows, the wave has every column in registers by then.  Wave 0 factors,
// the others leave.  256 threads, not 1 024: the factorisation keeps four 32-entry f64 arrays in registers, and a 1 024-thread
// workgroup is compiled for 128 VGPRs -- it spilled them (33 us; 16 us for the 64-thread kernel + 5 for the sum).  32 x 32 only.
__global__ __launch_bounds__(256) void k_chol_inv_fold32(const double* __restrict__ part, int P, int n, double* __restrict__ Zg, int* __restrict__ flag) {
    __shared__ double rs[32 * 32];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < P; p0 += 16) {
        double v[4][16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double* pp = part + (size_t)(p0 + u < P ? p0 + u : p0) * 1024 + threadIdx.x;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q][u] = pp[256 * q];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += (p0 + u < P) ? v[q][u] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) rs[threadIdx.x + 256 * q] = a[q];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    chol_inv_wave<32>(rs, n, Zg, flag, rs, (int)threadIdx.x);
}
// part: P <= 64 partial Gram matrices [32 * 32] (k_gram_mfma's output)
void launch_chol_inv_fold(hipStream_t st, const double* part, int P, int n, int ld, double* Z, int* flag) {
    (void)ld;
    hipLaunchKernelGGL(k_chol_inv_fold32, dim3(1), dim3(256), 0, st, part, P, n, Z, flag);
}
void launch_chol_inv(hipStream_t st, const double* W, int n, int ld, double* Z, int* flag) {
    if (ld == 32) hipLaunchKernelGGL(k_chol_inv<32>, dim3(1), dim3(64), 0, st, W, n, Z, flag);
    else if (ld == 64) hipLaunchKernelGGL(k_chol_inv<64>, dim3(1), dim3(64), 0, st, W, n, Z, flag);
    else (void)launch_chol_inv_any(st, W, n, ld, Z, flag);       // (ld = 128 = kMaxSketchCols: the factor lives in LDS)
}

void launch_apply_right_inplace(hipStream_t st, double* X, int64_t rows, int L, const double* Z, float* Qout,
                                int64_t rows_pad) {
    const int64_t span = Qout ? rows_pad : rows;
    if (L == 32) hipLaunchKernelGGL((k_apply_right<32>), dim3((unsigned)((span + 7) / 8)), dim3(256), 0, st, X, rows, Z, Qout, rows_pad);
    else if (L == 64) hipLaunchKernelGGL((k_apply_right<64>), dim3((unsigned)((span + 3) / 4)), dim3(256), 0, st, X, rows, Z, Qout, rows_pad);
    else launch_apply_right_any(st, X, rows, (rows + 63) / 64, L, Z, nullptr, nullptr);
}

// out[n][kc] = sum_j X[row(n)][j] Z[j][kc].  One thread per row: the row sits in registers (L/4 16-byte loads), Z is
// broadcast from LDS.  The f32 outputs of a block (256 rows x K, one contiguous run of the output) are staged in LDS and
// written back coalesced -- per-thread runs of K floats at a K-float stride cost 6.7x write amplification (WRITE_SIZE
// 538 MB for an 80 MB result).  The f64 output (scores: N rows only) is written directly.
template <typename TX, int L>
__global__ __launch_bounds__(256) void k_rightmul(const TX* __restrict__ X, const int64_t* __restrict__ row_ids,
                                                  int64_t nrows, const double* __restrict__ Z, int K,
                                                  double* __restrict__ out64, float* __restrict__ out32) {
    extern __shared__ double zsm[];          // Z [L][K], then (f32 output only) the staging tile [256][K | 1]
    float* osm = reinterpret_cast<float*>(zsm + L * K);
    const int KP = K | 1;                    // odd pitch: conflict-free column writes
    for (int e = threadIdx.x; e < L * K; e += 256) zsm[e] = Z[e];
    __syncthreads();
    const int64_t n0 = (int64_t)blockIdx.x * 256;
    const int64_t n = n0 + threadIdx.x;
    if (n < nrows) {
        const int64_t src = row_ids ? row_ids[n] : n;
        double x[L];
#pragma unroll
        for (int j = 0; j < L; ++j) x[j] = (double)X[src * L + j];
        for (int kc = 0; kc < K; ++kc) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < L; ++j) a += x[j] * zsm[j * K + kc];
            if (out64) out64[n * K + kc] = a;
            if (out32) osm[threadIdx.x * KP + kc] = (float)a;
        }
    }
    if (!out32) return;
    __syncthreads();
    const int64_t rows_here = (nrows - n0 < 256) ? nrows - n0 : 256;
    const int total = (int)rows_here * K;
    float* dst = out32 + n0 * K;
    for (int e = threadIdx.x; e < total; e += 256) dst[e] = osm[(e / K) * KP + (e % K)];
}
static size_t rightmul_lds(int L, int K, bool f32out) { return sizeof(double) * L * K + (f32out ? sizeof(float) * 256 * (size_t)(K | 1) : 0); }
void launch_rightmul_f64(hipStream_t st, const double* X, int64_t rows, int L, const double* Z, int K, double* out64,
                         float* out32) {
    const dim3 grid((unsigned)((rows + 255) / 256)), blk(256);
    const size_t lds = rightmul_lds(L, K, out32 != nullptr);
    if (L == 32) hipLaunchKernelGGL((k_rightmul<double, 32>), grid, blk, lds, st, X, (const int64_t*)nullptr, rows, Z, K, out64, out32);
    else if (L == 64) {   // (> 64 KiB of dynamic LDS: opted in per device by init_device_kernels_common)
        hipLaunchKernelGGL((k_rightmul<double, 64>), grid, blk, lds, st, X, (const int64_t*)nullptr, rows, Z, K, out64, out32);
    } else launch_rightmul_any_f64(st, X, rows, L, Z, K, out64, out32);
}
// Loadings = B[rows] (V S^-1): the tall f32 factor times an L x K f64 matrix, on the f64 matrix cores (the VALU kernel
// above is LDS-broadcast bound: 640 ds_reads per row).  One wave = 16 rows per tile: lane (i = lane & 15, kq = lane >> 4)
// loads the E = L/4 consecutive floats X[row_i][E kq ..] (the k-order of a dot product is free, so step s pairs
// A_s[i][kq] = X[row_i][E kq + s] with B_s[kq][j] = Z[E kq + s][j]); Z sits in registers for the whole kernel.
// D[4 r + lane / 16][lane % 16] (probe_mfma_f64.hip): a store covers 4 rows x 64 contiguous bytes.
template <int L, int NJ>
__global__ __launch_bounds__(256) void k_rightmul_mfma(const float* __restrict__ X, const int64_t* __restrict__ row_ids,
                                                       int64_t nrows, const double* __restrict__ Z, int K,
                                                       float* __restrict__ out32, int64_t tiles_per_wave, const int* __restrict__ sign) {
    constexpr int E = L / 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    double zb[E][NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int col = 16 * jt + i;
        const double sg = (sign && col < K) ? (double)sign[col] : 1.0;      // (+-1: exact)
#pragma unroll
        for (int s2 = 0; s2 < E; ++s2) zb[s2][jt] = col < K ? Z[(E * kq + s2) * K + col] * sg : 0.0;
    }
    __shared__ float osm[4][16 * 64];                   // per-wave output tile, written back as one contiguous run
    const int64_t ntiles = (nrows + 15) >> 4;
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + wv) * tiles_per_wave;
    const int64_t t1 = (t0 + tiles_per_wave < ntiles) ? t0 + tiles_per_wave : ntiles;
    float xa[E], xn[E];
    auto load_tile = [&](int64_t tile, float (&dst)[E]) {
        const int64_t row = tile * 16 + i;
        const bool valid = row < nrows && tile < t1;
        const int64_t src = valid ? (row_ids ? row_ids[row] : row) : 0;
        const float4* xp = reinterpret_cast<const float4*>(X + src * L + E * kq);
#pragma unroll
        for (int v = 0; v < E / 4; ++v) {
            const float4 q = xp[v];
            dst[4 * v] = valid ? q.x : 0.f; dst[4 * v + 1] = valid ? q.y : 0.f; dst[4 * v + 2] = valid ? q.z : 0.f; dst[4 * v + 3] = valid ? q.w : 0.f;
        }
    };
    if (t0 < t1) load_tile(t0, xa);
    for (int64_t tile = t0; tile < t1; ++tile) {
        load_tile(tile + 1, xn);                        // next tile's rows are in flight behind this tile's MFMAs (two tiles ahead and the
                                                        // row ids three ahead measured the same: 55.6 against 55.3 - 55.8 us, kbench_tail.hip)
        f64x4 acc[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < E; ++s2) {
            const double a = (double)xa[s2];
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zb[s2][jt], acc[jt], 0, 0, 0);
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int col = 16 * jt + i;
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) osm[wv][(4 * r + kq) * K + col] = (float)acc[jt][r];
            }
        }
        // (one wave: its LDS operations execute in issue order, so the reads below see the writes above)
        const int64_t rows_here = (nrows - tile * 16 < 16) ? nrows - tile * 16 : 16;
        float* dst = out32 + tile * 16 * K;
        for (int e = lane; e < (int)rows_here * K; e += 64) dst[e] = osm[wv][e];
#pragma unroll
        for (int s2 = 0; s2 < E; ++s2) xa[s2] = xn[s2];
    }
}
void launch_rightmul_gather_f32(hipStream_t st, const float* X, const int64_t* row_ids, int64_t nrows, int L,
                                const double* Z, int K, float* out32, const int* sign) {
    if (nrows == 0) return;
    if (L > 64) { launch_rightmul_any_gather_f32(st, X, row_ids, nrows, L, Z, K, out32); return; }      // (wide sketches: the caller scales Z by the sign first)
    const int64_t ntiles = (nrows + 15) / 16;
    int64_t tpw = ntiles / (4 * 2048);            // ~2048 workgroups, at least one tile per wave
    if (tpw < 1) tpw = 1;
    const int64_t waves = (ntiles + tpw - 1) / tpw;
    const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    const int nj = (K + 15) / 16;
#define GPCA_RM(LL, NN) hipLaunchKernelGGL((k_rightmul_mfma<LL, NN>), grid, blk, 0, st, X, row_ids, nrows, Z, K, out32, tpw, sign)
    if (L == 32) { if (nj == 1) GPCA_RM(32, 1); else if (nj == 2) GPCA_RM(32, 2); else if (nj == 3) GPCA_RM(32, 3); else GPCA_RM(32, 4); }
    else { if (nj == 1) GPCA_RM(64, 1); else if (nj == 2) GPCA_RM(64, 2); else if (nj == 3) GPCA_RM(64, 3); else GPCA_RM(64, 4); }
#undef GPCA_RM
}

// ---- sample scores and their sign rule (L <= 64) ---------------------------------------------------------------------------
// scores = X Z (X = the orthonormal basis, f64 [rows][L]; Z [L][K]); a column's sign is fixed so that its entry of largest
// magnitude is positive (the first such row on a tie).  k_scores writes the unsigned product and one candidate per workgroup and
// column -- the workgroups walk 256-row chunks with a stride, at most kScoreParts of them, so the consumer's fold stays small at any
// sample count; k_scores_sign folds the candidates in workgroup order (ascending rows within a workgroup's chunks, so "first row"
// needs the row index, not the fold order), applies the sign in place and writes the f32 copy.  Replaces k_rightmul + k_col_sign
// (one workgroup per column walking all rows: 17 us) + 2 x k_scale_cols.
constexpr int kScoreParts = 48;                       // (x 64 columns x 16 B of candidates = 48 KiB of LDS in k_scores_sign)
int64_t scores_num_parts(int64_t rows) { const int64_t c = (rows + 255) / 256; return c < kScoreParts ? (c < 1 ? 1 : c) : kScoreParts; }
template <int L>
__global__ __launch_bounds__(256) void k_scores(const double* __restrict__ X, int64_t nrows, const double* __restrict__ Z, int K,
                                                double* __restrict__ out64, double* __restrict__ cand_val, int64_t* __restrict__ cand_idx) {
    extern __shared__ double zsc[];                       // Z [L][K] | a 256 x 17 tile of the chunk's scores (16 columns at a time) | the workgroup's winners
    double* tile = zsc + L * K;
    double* win_val = tile + 256 * 17;                    // [64]
    long long* win_idx = reinterpret_cast<long long*>(win_val + 64);   // [64]
    __shared__ double seg_val[256];
    __shared__ int seg_row[256];
    for (int e = threadIdx.x; e < L * K; e += 256) zsc[e] = Z[e];
    if (threadIdx.x < 64) { win_val[threadIdx.x] = 0.0; win_idx[threadIdx.x] = -1; }
    __syncthreads();
    const int64_t nchunks = (nrows + 255) / 256;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t n0 = chunk * 256, n = n0 + threadIdx.x;
        const bool live = n < nrows;
        double x[L];
#pragma unroll
        for (int j = 0; j < L; ++j) x[j] = live ? X[n * L + j] : 0.0;
        for (int k0 = 0; k0 < K; k0 += 16) {
            const int kn = K - k0 < 16 ? K - k0 : 16;
            for (int kc = 0; kc < kn; ++kc) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < L; ++j) a += x[j] * zsc[j * K + k0 + kc];
                if (live) out64[n * K + k0 + kc] = a;
                tile[threadIdx.x * 17 + kc] = a;
            }
            __syncthreads();
            // a column's winner: the largest |score|, the lowest row on a tie (a wave-shuffle argmax per column cost six LDS-crossbar round
            // trips per column and chunk: 32 us for 10 000 x 20; one thread per column walking all 256 staged rows: 48 us).  Thread t
            // scans the 16 rows of segment t / 16 for column t % 16, then one thread per column folds the 16 segment winners, rows ascending;
            // the workgroup's earlier chunks hold lower rows, so they keep a tie.
            {
                const int kc = threadIdx.x & 15, seg = threadIdx.x >> 4;
                const int rows_here = nrows - n0 < 256 ? (int)(nrows - n0) : 256;
                double cv = 0.0; int cr = -1;
                if (kc < kn) {
                    double av[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) av[r] = tile[(16 * seg + r) * 17 + kc];
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (16 * seg + r < rows_here && (cr < 0 || fabs(av[r]) > fabs(cv))) { cv = av[r]; cr = 16 * seg + r; }
                }
                seg_val[threadIdx.x] = cv; seg_row[threadIdx.x] = cr;
            }
            __syncthreads();
            if ((int)threadIdx.x < kn) {
                double cv = win_val[k0 + threadIdx.x]; long long ci = win_idx[k0 + threadIdx.x];
                for (int sg2 = 0; sg2 < 16; ++sg2) {
                    const double a = seg_val[16 * sg2 + threadIdx.x]; const int r = seg_row[16 * sg2 + threadIdx.x];
                    if (r >= 0 && (ci < 0 || fabs(a) > fabs(cv))) { cv = a; ci = n0 + r; }
                }
                win_val[k0 + threadIdx.x] = cv; win_idx[k0 + threadIdx.x] = ci;
            }
            __syncthreads();
        }
    }
    for (int kc = threadIdx.x; kc < K; kc += 256) { cand_val[(int64_t)blockIdx.x * K + kc] = win_val[kc]; cand_idx[(int64_t)blockIdx.x * K + kc] = win_idx[kc]; }
}
__global__ __launch_bounds__(256) void k_scores_sign(double* __restrict__ X64, float* __restrict__ X32, int64_t total, int K,
                                                     const double* __restrict__ cand_val, const int64_t* __restrict__ cand_idx, int parts,
                                                     int* __restrict__ sign) {
    // the candidates (parts x K <= 128 x 64 pairs) come into LDS with every thread loading its share -- a thread walking its column's
    // candidates one dependent load at a time waited 40 L2 round trips: 17 us for 40 parts
    extern __shared__ double cv_s[];                      // values [parts * K], then indices
    long long* ci_s = reinterpret_cast<long long*>(cv_s + parts * K);
    __shared__ int sg[kMaxSketchCols];
    for (int e = threadIdx.x; e < parts * K; e += 256) { cv_s[e] = cand_val[e]; ci_s[e] = cand_idx[e]; }
    __syncthreads();
    for (int kc = threadIdx.x; kc < K; kc += 256) {
        double bv = 0.0; long long bi = -1;
        for (int p = 0; p < parts; ++p) {
            const double ov = cv_s[p * K + kc]; const long long oi = ci_s[p * K + kc];
            const bool take = oi >= 0 && (bi < 0 || fabs(ov) > fabs(bv) || (fabs(ov) == fabs(bv) && oi < bi));
            bv = take ? ov : bv; bi = take ? oi : bi;
        }
        const int s1 = (bi >= 0 && bv < 0.0) ? -1 : 1;
        sg[kc] = s1;
        if (blockIdx.x == 0) sign[kc] = s1;
    }
    __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const double v = X64[t] * (double)sg[t % K];
        X64[t] = v;
        if (X32) X32[t] = (float)v;
    }
}
void launch_scores(hipStream_t st, const double* X, int64_t rows, int L, const double* Z, int K, double* out64, double* cand_val, int64_t* cand_idx) {
    const dim3 grid((unsigned)scores_num_parts(rows)), blk(256);
    const size_t lds = sizeof(double) * ((size_t)L * K + 256 * 17 + 64) + sizeof(long long) * 64;      // (L = K = 64: 68 KiB, opted in by init_device_kernels_common)
    if (L == 32) hipLaunchKernelGGL((k_scores<32>), grid, blk, lds, st, X, rows, Z, K, out64, cand_val, cand_idx);
    else hipLaunchKernelGGL((k_scores<64>), grid, blk, lds, st, X, rows, Z, K, out64, cand_val, cand_idx);
}
void launch_scores_sign(hipStream_t st, double* out64, float* out32, int64_t rows, int K, const double* cand_val, const int64_t* cand_idx,
                        int64_t parts, int* sign) {
    const int64_t total = rows * K;
    int64_t blocks = (total + 1023) / 1024;                          // four elements per thread
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);         // (every workgroup loads the candidates: keep them few)
    const size_t lds = (sizeof(double) + sizeof(long long)) * (size_t)parts * K;
    hipLaunchKernelGGL(k_scores_sign, dim3((unsigned)blocks), dim3(256), lds, st, out64, out32, total, K, cand_val, cand_idx, (int)parts, sign);
}

constexpr int kColsumRowsPerBlock = 256;
int64_t colsum_num_parts(int64_t rows) { return (rows + kColsumRowsPerBlock - 1) / kColsumRowsPerBlock; }
__global__ __launch_bounds__(256) void k_colsum(const double* __restrict__ X, int64_t rows, int L, double* __restrict__ part) {
    __shared__ double red[256];
    const int cc = threadIdx.x % L, rg = threadIdx.x / L, nrg = 256 / L;
    const int64_t r0 = (int64_t)blockIdx.x * kColsumRowsPerBlock;
    const int64_t r1 = (r0 + kColsumRowsPerBlock < rows) ? r0 + kColsumRowsPerBlock : rows;
    double a = 0.0;
    for (int64_t n = r0 + rg; n < r1; n += nrg) a += (double)X[n * L + cc];
    red[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < nrg; ++g) a += red[g * L + cc];
        part[(int64_t)blockIdx.x * L + cc] = a;
    }
}
void launch_colsum_f64(hipStream_t st, const double* X, int64_t rows, int L, double* part) {
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)colsum_num_parts(rows)), dim3(256), 0, st, X, rows, L, part);
}

// sign of the first element with maximal |x| per column; one block per column.  1 024 threads and four independent loads per trip:
// the 256-thread form walked 10 000 rows in 39 dependent round trips to L2 (17 us for 10 columns)
__global__ __launch_bounds__(1024) void k_col_sign(const double* __restrict__ X, int64_t rows, int K, int* __restrict__ sign) {
    __shared__ double bv[1024];
    __shared__ long long bi[1024];
    const int col = blockIdx.x;
    double best = -1.0; long long idx = -1;
    for (int64_t n0 = threadIdx.x; n0 < rows; n0 += 4096) {
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t n = n0 + 1024 * u; a[u] = n < rows ? fabs(X[n * K + col]) : -1.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (a[u] > best) { best = a[u]; idx = n0 + 1024 * u; }      // (ascending rows: the first maximum stays)
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double ob = bv[threadIdx.x + s]; const long long oi = bi[threadIdx.x + s];
            if (ob > bv[threadIdx.x] || (ob == bv[threadIdx.x] && oi >= 0 && (bi[threadIdx.x] < 0 || oi < bi[threadIdx.x]))) {
                bv[threadIdx.x] = ob; bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) sign[col] = (bi[0] >= 0 && X[bi[0] * K + col] < 0.0) ? -1 : 1;
}
void launch_col_sign(hipStream_t st, const double* X, int64_t rows, int K, int* sign) {
    hipLaunchKernelGGL(k_col_sign, dim3(K), dim3(1024), 0, st, X, rows, K, sign);
}
__global__ void k_scale_cols(double* X64, float* X32, int64_t rows, int K, const int* sign) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * K) return;
    const double v = X64[t] * (double)sign[t % K];
    X64[t] = v;
    if (X32) X32[t] = (float)v;
}
void launch_scale_cols(hipStream_t st, double* X64, float* X32, int64_t rows, int K, const int* sign) {
    const int64_t total = rows * K;
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X64, X32, rows, K, sign);
}
__global__ void k_f64_to_f32(const double* in, float* out, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = (float)in[t];
}
void launch_f64_to_f32(hipStream_t st, const double* in, float* out, int64_t n) {
    hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}
__global__ void k_fill_f32(float* p, int64_t n, float v) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) p[t] = v;
}
void launch_fill_f32(hipStream_t st, float* p, int64_t n, float v) {
    hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}


__global__ void k_expand_loadings(const float* __restrict__ load, const int64_t* __restrict__ rows, int64_t n_pca, int k,
                                  int L, float* __restrict__ Tp) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_pca * k) return;
    const int64_t a = t / k;
    const int c = (int)(t - a * k);
    Tp[rows[a] * L + c] = load[t];
}
void launch_expand_loadings(hipStream_t st, const float* load, const int64_t* rows, int64_t n_pca, int k, int L, float* Tp) {
    const int64_t total = n_pca * k;
    if (total == 0) return;
    hipLaunchKernelGGL(k_expand_loadings, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, load, rows, n_pca, k, L, Tp);
}

// one wave = 64 rows; lane = row; loops over the L columns (tiny kernel: transform path only)
__global__ __launch_bounds__(256) void k_scale_rows(const float* __restrict__ X, int64_t M, int64_t Mpad, int L,
                                                    const float* __restrict__ r, const float* __restrict__ b,
                                                    float* __restrict__ Tb, float* __restrict__ cpart, int blocked) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (Mpad + 63) / 64) return;
    const int64_t i = wave * 64 + lane;
    const bool live = i < M;
    const float ri = live ? r[i] : 0.f, bi = live ? b[i] : 0.f;
    const int LT = L >> 5;
    for (int j = 0; j < L; ++j) {
        const float x = live ? X[i * L + j] : 0.f;
        if (i < Mpad) Tb[blocked ? blocked_t_index(i, j, LT) : i * L + j] = ri * x;
        float cv = bi * x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cv += __shfl_xor(cv, o);
        if (lane == 0) cpart[wave * L + j] = cv;
    }
}
void launch_scale_rows(hipStream_t st, const float* X, int64_t M, int64_t Mpad, int L, const float* r, const float* b,
                       float* Tb, float* cpart, int blocked) {
    const int64_t waves = (Mpad + 63) / 64;
    hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, X, M, Mpad, L, r, b, Tb, cpart, blocked);
}


// ------------------------------------------------------------------------------------------------
// EigenSNP stages.  The condensed feature matrix C* = Lambda^-1 U_blk^T X (one row per local eigenSNP of an
// LD block) is never formed: W = U_blk Lambda^-1 is block diagonal -- SNP row i of block b carries the <= cmax coefficients
// W[i][0..cmax) of the block's features [feat0[i], feat0[i] + cmax) -- so C* Q = W^T (X Q) and C*^T Z = X^T (W Z) run through the
// genotype GEMMs with these two O(M l cmax) kernels in between.
// ------------------------------------------------------------------------------------------------
// out[i][j] = sum_c W[i][c] P[feat0[i] + c][j]   (rows without a block: 0)
__global__ __launch_bounds__(256) void k_bd_expand(const float* __restrict__ W, const int32_t* __restrict__ feat0, int cmax,
                                                   const double* __restrict__ P, int64_t M, int L, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * L) return;
    const int64_t i = t / L;
    const int j = (int)(t - i * L);
    const int32_t f = feat0[i];
    double a = 0.0;
    if (f >= 0)
        for (int c = 0; c < cmax; ++c) a = fma((double)W[i * cmax + c], P[(int64_t)(f + c) * L + j], a);
    out[t] = (float)a;
}
void launch_bd_expand(hipStream_t st, const float* W, const int32_t* feat0, int cmax, const double* P, int64_t M, int L, float* out) {
    hipLaunchKernelGGL(k_bd_expand, dim3((unsigned)((M * L + 255) / 256)), dim3(256), 0, st, W, feat0, cmax, P, M, L, out);
}
// P[bf + c][j] = sum over the rows i of block b (feat0[i] == bf, rows [row0, row1) in ascending order: one fixed summation order)
// of W[i][c] T[i][j].  One workgroup per block; thread = (c mod 8, j mod 32).
__global__ __launch_bounds__(256) void k_bd_reduce(const float* __restrict__ W, const int32_t* __restrict__ feat0, int cmax,
                                                   const float* __restrict__ T, int L, const int64_t* __restrict__ blk_row0,
                                                   const int64_t* __restrict__ blk_row1, const int32_t* __restrict__ blk_feat0,
                                                   const int32_t* __restrict__ blk_c, double* __restrict__ P) {
    const int b = blockIdx.x;
    const int64_t r0 = blk_row0[b], r1 = blk_row1[b];
    const int32_t bf = blk_feat0[b];
    const int cb = blk_c[b];        // the block's own feature count (<= cmax): rows bf + cb .. of P belong to the next block
    const int cg = threadIdx.x >> 5, jl = threadIdx.x & 31;
    for (int c = cg; c < cb; c += 8)
        for (int j = jl; j < L; j += 32) {
            double a = 0.0;
            for (int64_t i = r0; i < r1; ++i)
                if (feat0[i] == bf) a = fma((double)W[i * cmax + c], (double)T[i * L + j], a);
            P[(int64_t)(bf + c) * L + j] = a;
        }
}
void launch_bd_reduce(hipStream_t st, const float* W, const int32_t* feat0, int cmax, const float* T, int L, const int64_t* blk_row0,
                      const int64_t* blk_row1, const int32_t* blk_feat0, const int32_t* blk_c, int B, double* P) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_bd_reduce, dim3((unsigned)B), dim3(256), 0, st, W, feat0, cmax, T, L, blk_row0, blk_row1, blk_feat0, blk_c, P);
}
// X[i][:] <- X[i][:] Z   (X: rows x L f32 in place, Z: L x L f64 row-major): the triangular factor of a CholeskyQR of a tall f32 matrix
template <int L>
__global__ __launch_bounds__(256) void k_rightmul_inplace_f32(float* __restrict__ X, int64_t rows, const double* __restrict__ Z) {
    __shared__ double zs[L * L];
    for (int e = threadIdx.x; e < L * L; e += 256) zs[e] = Z[e];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    double x[L];
#pragma unroll
    for (int k = 0; k < L; ++k) x[k] = (double)X[i * L + k];
    for (int j = 0; j < L; ++j) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < L; ++k) a = fma(x[k], zs[k * L + j], a);
        X[i * L + j] = (float)a;
    }
}
void launch_rightmul_inplace_f32(hipStream_t st, float* X, int64_t rows, int L, const double* Z) {
    const dim3 grid((unsigned)((rows + 255) / 256)), blk(256);
    if (L == 32) hipLaunchKernelGGL((k_rightmul_inplace_f32<32>), grid, blk, 0, st, X, rows, Z);
    else hipLaunchKernelGGL((k_rightmul_inplace_f32<64>), grid, blk, 0, st, X, rows, Z);
}
// rows of Y (N x L f64) whose sample is outside the subset become zero
__global__ __launch_bounds__(256) void k_mask_rows(double* __restrict__ Y, int64_t N, int L, const uint8_t* __restrict__ mask) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= N * L) return;
    if (!mask[t / L]) Y[t] = 0.0;
}
void launch_mask_rows(hipStream_t st, double* Y, int64_t N, int L, const uint8_t* mask) {
    hipLaunchKernelGGL(k_mask_rows, dim3((unsigned)((N * L + 255) / 256)), dim3(256), 0, st, Y, N, L, mask);
}
__global__ __launch_bounds__(256) void k_f32_to_f64(const float* __restrict__ in, double* __restrict__ out, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = (double)in[t];
}
void launch_f32_to_f64(hipStream_t st, const float* in, double* out, int64_t n) {
    hipLaunchKernelGGL(k_f32_to_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

// > 64 KiB of dynamic LDS is an opt-in the runtime records per device
int init_device_kernels_common() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rightmul<double, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scores<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    return (int)e;
}

}  // namespace gpca
