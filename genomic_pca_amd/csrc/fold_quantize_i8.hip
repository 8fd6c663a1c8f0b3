// Folds of K2's partial tiles and the quantisation of the skinny operand into digit planes (exact-integer path).
#include "gemm_i8_common.h"

namespace gpca {

// Y[n][j] = c[j] + tscale[j] * sum_w Ypart[w][n][j]    (the integer sum is exact and order-independent)
// SPLIT = 8: the W slices are shared between 8 thread groups of a block (32 consecutive elements each) and folded through LDS --
// for few samples the one-thread-per-element form leaves a handful of blocks walking W (thousands of) slices one load at a
// time: 224 us per launch at 1 066 557 x 64 (configs[2]), 30 % of that call.  Same bits either way: the partial sums are integers.
template <int SPLIT>
__device__ __forceinline__ double reduce_slices(const double* __restrict__ Ypart, int W, int64_t stride, int64_t total, int64_t& e, bool& live) {
    if (SPLIT == 1) {
        e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        live = e < total;
        double s = 0.0;
        if (live) {
            // eight loads in flight per thread (the sums are integers: any order gives the same bits).  K2 writes its partial tiles with
            // streaming stores, so this read comes from HBM, not from L2 / MALL: latency, not the adds, is what a thread waits for
            const double* p = Ypart + e;
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int w = 0;
            for (; w + 8 <= W; w += 8) {
                const double v0 = p[(w + 0) * stride], v1 = p[(w + 1) * stride], v2 = p[(w + 2) * stride], v3 = p[(w + 3) * stride];
                const double v4 = p[(w + 4) * stride], v5 = p[(w + 5) * stride], v6 = p[(w + 6) * stride], v7 = p[(w + 7) * stride];
                s += v0; s1 += v1; s2 += v2; s3 += v3; s += v4; s1 += v5; s2 += v6; s3 += v7;
            }
            for (; w < W; ++w) s += p[w * stride];
            s = (s + s1) + (s2 + s3);
        }
        return s;
    }
    __shared__ double part[256];
    const int grp = threadIdx.x >> 5, le = threadIdx.x & 31;
    e = (int64_t)blockIdx.x * 32 + le;
    live = e < total;
    double s = 0.0;
    if (live) for (int w = grp; w < W; w += 8) s += Ypart[w * stride + e];
    part[threadIdx.x] = s;
    __syncthreads();
    if (grp != 0) { live = false; return 0.0; }
#pragma unroll
    for (int g = 1; g < 8; ++g) s += part[g * 32 + le];
    return s;
}
template <int SPLIT>
__global__ __launch_bounds__(256) void k_reduce_y_i8(const double* __restrict__ Ypart, int W, int64_t Npad, int64_t N,
                                                     const double* __restrict__ cvec, const double* __restrict__ tscale,
                                                     double* __restrict__ Y, int64_t ldy) {
    int64_t e; bool live;
    const double s = reduce_slices<SPLIT>(Ypart, W, Npad * 32, N * 32, e, live);
    if (!live) return;
    const int j = (int)(e & 31);
    Y[(e >> 5) * ldy + j] = fma(tscale[j], s, cvec[j]);   // one rounding, the same in k_finish_y_i8
}
// one thread per element while that still fills the chip (>= 1024 blocks), 8 threads per element below
static inline bool reduce_split(int64_t total) { return total < (int64_t)256 * 1024; }
void launch_reduce_y_i8(hipStream_t st, const double* Ypart, int W, int64_t Npad, int64_t N, const double* c,
                        const double* tscale, double* Y, int64_t ldy) {
    const int64_t total = N * 32;
    if (reduce_split(total)) hipLaunchKernelGGL(k_reduce_y_i8<8>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, Ypart, W, Npad, N, c, tscale, Y, ldy);
    else hipLaunchKernelGGL(k_reduce_y_i8<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Ypart, W, Npad, N, c, tscale, Y, ldy);
}

// Streamed panels: the integer partial sums of every panel are added into Yint (exact in f64 while |sum| < 2^53, so the
// order of the panels does not matter) and scaled once at the end -- the same value, bit for bit, as the one-launch reduce.
template <int SPLIT>
__global__ __launch_bounds__(256) void k_accum_y_i8(const double* __restrict__ Ypart, int W, int64_t Npad, int64_t N,
                                                    double* __restrict__ Yint, int first) {
    int64_t e; bool live;
    const double s = reduce_slices<SPLIT>(Ypart, W, Npad * 32, N * 32, e, live);
    if (!live) return;
    Yint[e] = first ? s : Yint[e] + s;
}
__global__ __launch_bounds__(256) void k_finish_y_i8(const double* __restrict__ Yint, int64_t N, const double* __restrict__ cvec,
                                                     const double* __restrict__ tscale, double* __restrict__ Y, int64_t ldy) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * 32) return;
    const int j = (int)(e & 31);
    Y[(e >> 5) * ldy + j] = fma(tscale[j], Yint[e], cvec[j]);
}
void launch_accum_y_i8(hipStream_t st, const double* Ypart, int W, int64_t Npad, int64_t N, double* Yint, int first) {
    const int64_t total = N * 32;
    if (reduce_split(total)) hipLaunchKernelGGL(k_accum_y_i8<8>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, Ypart, W, Npad, N, Yint, first);
    else hipLaunchKernelGGL(k_accum_y_i8<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Ypart, W, Npad, N, Yint, first);
}
void launch_finish_y_i8(hipStream_t st, const double* Yint, int64_t N, const double* c, const double* tscale, double* Y, int64_t ldy) {
    const int64_t total = N * 32;
    hipLaunchKernelGGL(k_finish_y_i8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Yint, N, c, tscale, Y, ldy);
}
// Fused streamed power iteration: every panel quantises its rows of T' against its OWN column maximum, so its integer sums carry
// their own scale: Yacc[n][j] (+)= tscale_p[j] * sum_w Ypart[w][n][j]  (exact integer sum, one fma per panel, fixed panel order).
template <int SPLIT>
__global__ __launch_bounds__(256) void k_accum_y_scaled(const double* __restrict__ Ypart, int W, int64_t Npad, int64_t N,
                                                        const double* __restrict__ tscale, double* __restrict__ Yacc, int first) {
    int64_t e; bool live;
    const double s = reduce_slices<SPLIT>(Ypart, W, Npad * 32, N * 32, e, live);
    if (!live) return;
    Yacc[e] = fma(tscale[e & 31], s, first ? 0.0 : Yacc[e]);
}
__global__ __launch_bounds__(256) void k_finish_y_sum(const double* __restrict__ Yacc, int64_t N, const double* __restrict__ cvec,
                                                      double* __restrict__ Y, int64_t ldy) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * 32) return;
    Y[(e >> 5) * ldy + (e & 31)] = cvec[e & 31] + Yacc[e];
}
void launch_accum_y_scaled(hipStream_t st, const double* Ypart, int W, int64_t Npad, int64_t N, const double* tscale, double* Yacc, int first) {
    const int64_t total = N * 32;
    if (reduce_split(total)) hipLaunchKernelGGL(k_accum_y_scaled<8>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, Ypart, W, Npad, N, tscale, Yacc, first);
    else hipLaunchKernelGGL(k_accum_y_scaled<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Ypart, W, Npad, N, tscale, Yacc, first);
}
void launch_finish_y_sum(hipStream_t st, const double* Yacc, int64_t N, const double* c, double* Y, int64_t ldy) {
    const int64_t total = N * 32;
    hipLaunchKernelGGL(k_finish_y_sum, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Yacc, N, c, Y, ldy);
}
// run[c] = max(run[c], max_p apart[p][c]): the column abs-max of T' over the panels seen so far (max is exact in any order)
__global__ __launch_bounds__(1024) void k_absmax_fold(const double* __restrict__ apart, int64_t P, double* __restrict__ run) {
    __shared__ double red[1024];
    const int cc = threadIdx.x & 31, pg = threadIdx.x >> 5;
    double a = 0.0;
    for (int64_t p = pg; p < P; p += 32) { const double v = apart[p * 32 + cc]; a = v > a ? v : a; }
    red[threadIdx.x] = a;
    __syncthreads();
    if (pg != 0) return;
    for (int g = 1; g < 32; ++g) { const double v = red[g * 32 + cc]; a = v > a ? v : a; }
    const double r = run[cc];
    run[cc] = a > r ? a : r;
}
void launch_absmax_fold(hipStream_t st, const double* apart, int64_t P, double* run) {
    hipLaunchKernelGGL(k_absmax_fold, dim3(1), dim3(1024), 0, st, apart, P, run);
}

// ------------------------------------------------------------------------------------------------
// Quantisation of the skinny operand: column abs-max -> scale; X[rows][32] -> digit planes
// blocked [block = row/32][d][lane = 32*((row%32)/16) + col][j = row%16] (16 B per lane per plane).
// ------------------------------------------------------------------------------------------------
constexpr int kAbsmaxRowsPerBlock = 1024;
int64_t absmax_num_parts(int64_t rows) { return (rows + kAbsmaxRowsPerBlock - 1) / kAbsmaxRowsPerBlock; }

template <typename T>
__global__ __launch_bounds__(256) void k_col_absmax(const T* __restrict__ X, int64_t rows, double* __restrict__ part, int64_t ldx) {
    __shared__ double red[256];
    const int cc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kAbsmaxRowsPerBlock;
    const int64_t r1 = (r0 + kAbsmaxRowsPerBlock < rows) ? r0 + kAbsmaxRowsPerBlock : rows;
    double a = 0.0;
    for (int64_t n = r0 + rg; n < r1; n += 8) { const double v = fabs((double)X[n * ldx + cc]); a = v > a ? v : a; }
    red[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < 8; ++g) { const double v = red[g * 32 + cc]; a = v > a ? v : a; }
        part[(int64_t)blockIdx.x * 32 + cc] = a;
    }
}
// scale[j] = colmax_j / S (multiplier back to real units), inv[j] = S / colmax_j; colmax 0 -> scale 0, inv 0
__global__ __launch_bounds__(1024) void k_finish_scale(const double* __restrict__ part, int64_t P, double* __restrict__ scale,
                                                       double* __restrict__ inv, double S) {
    __shared__ double red[1024];
    const int cc = threadIdx.x & 31, pg = threadIdx.x >> 5;   // 32 part-groups
    double a = 0.0;
    int64_t p = pg;
    for (; p + 96 < P; p += 128) {       // four loads in flight per thread (the one-load loop waited for each: 10 us for 256 KB)
        const double v0 = part[p * 32 + cc], v1 = part[(p + 32) * 32 + cc], v2 = part[(p + 64) * 32 + cc], v3 = part[(p + 96) * 32 + cc];
        a = fmax(fmax(a, fmax(v0, v1)), fmax(v2, v3));
    }
    for (; p < P; p += 32) { const double v = part[p * 32 + cc]; a = v > a ? v : a; }
    red[threadIdx.x] = a;
    __syncthreads();
    if (pg != 0) return;
    for (int g = 1; g < 32; ++g) { const double v = red[g * 32 + cc]; a = v > a ? v : a; }
    scale[cc] = a > 0.0 ? a / S : 0.0;
    inv[cc] = a > 0.0 ? S / a : 0.0;
}

// layout 0: block = 32 consecutive rows, lane half hh, element j -> row 32*blk + 16*hh + j
// layout 1 (packed K1): block = MFMA step (b, s) of a 128-row group -> row 128*(blk/4) + 64*hh + 16*(blk%4) + j
// The 16 loads of a lane are unconditional (row index clamped, the value zeroed afterwards) and issued together: with a
// predicated load per element the compiler waited for each one before asking for the next (16 serial latencies per lane,
// 3.8 TB/s on the 256 MB of a T' pass); ND is a template parameter so that the digit loop carries no runtime branch.
// the tail of the orthonormalisation that k_quantize<double> can carry (launch_quantize_f64_finishq); amax_part = NULL: nothing to carry
struct QFold { const double* amax_part; const double* csum_part; int64_t P; int ldp; double S; double* s64; float* s32; double* scale; double* inv; };
template <int ND>
__device__ __forceinline__ void split_digits(int v, unsigned (&w)[kDigits][4], int j) {
#pragma unroll
    for (int d = 0; d < kDigits; ++d) {
        int dg;
        if (ND == 3) { if (d < 2) { dg = ((v + 128) & 255) - 128; v = (v - dg) >> 8; } else { dg = v; v = 0; } }    // base 256, plane 3 = 0
        else if (d < kDigits - 1) { dg = ((v + 64) & 127) - 64; v = (v - dg) >> 7; } else dg = v;
        w[d][j >> 2] |= ((unsigned)(dg & 0xff)) << (8 * (j & 3));
    }
}
template <typename T, int ND>
__global__ __launch_bounds__(256) void k_quantize(const T* __restrict__ X, int64_t rows, int64_t rows_pad,
                                                  const double* __restrict__ inv, int8_t* __restrict__ Xd, int layout, int64_t ldx,
                                                  const double* __restrict__ cscratch, int cslices, double* __restrict__ cout, QFold qf) {
    __shared__ double sinv[32];
    if (qf.amax_part) {
        // The orthonormal basis' tail rides along (it was k_finish_q, one more launch): every workgroup folds the column abs-max partials of
        // k_apply_right_tail for its 32 columns itself (max is exact in any order: every workgroup gets the same scale); the first one also
        // folds the column sums s = Q^T 1 in a fixed order and publishes scale / inv / s for the kernels that follow.
        __shared__ double rm[256], rs[256];
        const int cc = threadIdx.x & 31, pg = threadIdx.x >> 5;
        double m = 0.0, a = 0.0;
        int64_t p = pg;
        for (; p < qf.P; p += 64) {                        // eight loads in flight per thread (one L2 round trip each: the fold is all latency)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = qf.amax_part[(p + 8 * u < qf.P ? p + 8 * u : p) * qf.ldp + cc];
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmax(m, v[u]);
        }
        if (blockIdx.x == 0)
            for (int64_t p2 = pg; p2 < qf.P; p2 += 64) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = qf.csum_part[(p2 + 8 * u < qf.P ? p2 + 8 * u : p2) * qf.ldp + cc];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (p2 + 8 * u < qf.P) ? v[u] : 0.0;
            }
        rm[threadIdx.x] = m; rs[threadIdx.x] = a;
        __syncthreads();
        if (pg == 0) {
            for (int g = 1; g < 8; ++g) { m = fmax(m, rm[g * 32 + cc]); a += rs[g * 32 + cc]; }
            const double iv = m > 0.0 ? qf.S / m : 0.0;
            sinv[cc] = iv;
            if (blockIdx.x == 0) { qf.scale[cc] = m > 0.0 ? m / qf.S : 0.0; qf.inv[cc] = iv; qf.s64[cc] = a; qf.s32[cc] = (float)a; }
        }
        __syncthreads();
    }
    if (cscratch && blockIdx.x == 0) {      // the second stage of c = b^T T rides along (the first ran beside the scale's fold: k_post_k1)
        __shared__ double red[4][64];
        sum_partials_body<double>(cscratch, cslices, 32, cout, 1, 0, 0, red);
    }
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk * 32 >= rows_pad) return;
    const int cc = lane & 31, hh = lane >> 5;
    const double sc = qf.amax_part ? sinv[cc] : inv[cc];
    const int64_t rbase = layout ? (blk >> 2) * 128 + 64 * hh + 16 * (blk & 3) : blk * 32 + 16 * hh;
    T xv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t row = rbase + j;
        xv[j] = X[(row < rows ? row : rows - 1) * ldx + cc];
    }
    unsigned w[kDigits][4];
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[d][q] = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double x = rbase + j < rows ? (double)xv[j] : 0.0;
        split_digits<ND>(__double2int_rn(x * sc), w, j);          // |x * sc| <= 0.49 * 2^28 (2^24 in three-plane mode): 32-bit digit arithmetic
    }
#pragma unroll
    for (int d = 0; d < kDigits; ++d)
        *reinterpret_cast<uint4*>(Xd + ((blk * kDigits + d) * 64 + lane) * 16) = make_uint4(w[d][0], w[d][1], w[d][2], w[d][3]);
}
template <typename T>
static void launch_k_quantize(hipStream_t st, const T* X, int64_t rows, int64_t rows_pad, const double* inv, int8_t* Xd, int layout, int nd, int64_t ldx,
                              const double* cscratch = nullptr, int cslices = 0, double* c = nullptr, QFold qf = QFold{}) {
    const int64_t blocks = rows_pad / 32;
    const dim3 grid((unsigned)((blocks + 3) / 4)), blk(256);
    if (nd == 3) hipLaunchKernelGGL((k_quantize<T, 3>), grid, blk, 0, st, X, rows, rows_pad, inv, Xd, layout, ldx, cscratch, cslices, c, qf);
    else hipLaunchKernelGGL((k_quantize<T, kDigits>), grid, blk, 0, st, X, rows, rows_pad, inv, Xd, layout, ldx, cscratch, cslices, c, qf);
}
// the digit planes of 32 columns of the orthonormal basis X (f64) with k_finish_q's work folded in: the partials of k_apply_right_tail
// (amax_part / csum_part [P][ldp], already offset to these columns) -> scale / inv / s64 / s32 of these columns.  For P <= kFinishQFoldMax.
void launch_quantize_f64_finishq(hipStream_t st, const double* X, int64_t rows, int64_t rows_pad, int8_t* Xd, int layout, int nd, int64_t ldx,
                                 const double* csum_part, const double* amax_part, int64_t P, int ldp, double* s64, float* s32, double* scale, double* inv) {
    QFold qf{amax_part, csum_part, P, ldp, digit_scale(nd), s64, s32, scale, inv};
    launch_k_quantize<double>(st, X, rows, rows_pad, inv, Xd, layout, nd, ldx, nullptr, 0, nullptr, qf);
}
// K1's tail: the first stage of c over the units' partials (workgroups 0 .. S - 1: sum_partials_body, E = 32) and the fold of the waves'
// column abs-max into the digit scale (workgroup S), side by side in one launch
__global__ __launch_bounds__(256) void k_post_k1(const float* __restrict__ cpart, int64_t units, double* __restrict__ cscratch, int S,
                                                 const double* __restrict__ apart, int64_t P, double* __restrict__ scale, double* __restrict__ inv, double DS) {
    __shared__ double red[4][64];
    if ((int)blockIdx.x < S) { sum_partials_body<float>(cpart, units, 32, cscratch, S, 0, blockIdx.x, red); return; }
    double* rf = &red[0][0];                               // 256 doubles
    const int cc = threadIdx.x & 31, pg = threadIdx.x >> 5;   // 8 part-groups
    double a = 0.0;
    int64_t p = pg;
    for (; p + 24 < P; p += 32) {                          // four loads in flight per thread
        const double v0 = apart[p * 32 + cc], v1 = apart[(p + 8) * 32 + cc], v2 = apart[(p + 16) * 32 + cc], v3 = apart[(p + 24) * 32 + cc];
        a = fmax(fmax(a, fmax(v0, v1)), fmax(v2, v3));
    }
    for (; p < P; p += 8) { const double v = apart[p * 32 + cc]; a = v > a ? v : a; }
    rf[threadIdx.x] = a;
    __syncthreads();
    if (pg != 0) return;
    for (int g = 1; g < 8; ++g) { const double v = rf[g * 32 + cc]; a = v > a ? v : a; }
    scale[cc] = a > 0.0 ? a / DS : 0.0;
    inv[cc] = a > 0.0 ? DS / a : 0.0;
}
int post_k1_slices(int64_t units) { return sum_slices(units, 32); }
void launch_post_k1(hipStream_t st, const float* cpart, int64_t units, double* cscratch, const double* apart, int64_t P, double* scale, double* inv, int nd) {
    const int S = post_k1_slices(units);
    hipLaunchKernelGGL(k_post_k1, dim3((unsigned)(S + 1)), dim3(256), 0, st, cpart, units, cscratch, S, apart, P, scale, inv, digit_scale(nd));
}
void launch_quantize_f32_cfold(hipStream_t st, const float* X, int64_t rows, int64_t rows_pad, const double* inv, int8_t* Xd, int layout, int nd,
                               int64_t ldx, const double* cscratch, int cslices, double* c) {
    launch_k_quantize<float>(st, X, rows, rows_pad, inv, Xd, layout, nd, ldx, cscratch, cslices, c);
}

template <typename T>
static void quantize_t(hipStream_t st, const T* X, int64_t rows, int64_t rows_pad, double* part, double* scale, double* inv,
                       int8_t* Xd, int layout, int nd, int64_t ldx) {
    const int64_t P = absmax_num_parts(rows);
    hipLaunchKernelGGL((k_col_absmax<T>), dim3((unsigned)P), dim3(256), 0, st, X, rows, part, ldx);
    hipLaunchKernelGGL(k_finish_scale, dim3(1), dim3(1024), 0, st, (const double*)part, P, scale, inv, digit_scale(nd));
    launch_k_quantize<T>(st, X, rows, rows_pad, (const double*)inv, Xd, layout, nd, ldx);
}
// abs-max partials already produced by the kernel that wrote X (K1 epilogue): finish the scale and quantise
void launch_quantize_f32_premax(hipStream_t st, const float* X, int64_t rows, int64_t rows_pad, const double* apart, int64_t P,
                                double* scale, double* inv, int8_t* Xd, int layout, int nd, int64_t ldx) {
    hipLaunchKernelGGL(k_finish_scale, dim3(1), dim3(1024), 0, st, apart, P, scale, inv, digit_scale(nd));
    launch_k_quantize<float>(st, X, rows, rows_pad, (const double*)inv, Xd, layout, nd, ldx);
}
// quantise X (f64) with a column scale that is already on the device (k_finish_q)
void launch_quantize_f64_prescaled(hipStream_t st, const double* X, int64_t rows, int64_t rows_pad, const double* inv, int8_t* Xd, int layout,
                                   int nd, int64_t ldx) {
    launch_k_quantize<double>(st, X, rows, rows_pad, inv, Xd, layout, nd, ldx);
}
void launch_quantize_f32(hipStream_t st, const float* X, int64_t rows, int64_t rows_pad, double* part, double* scale,
                         double* inv, int8_t* Xd, int layout, int nd, int64_t ldx) { quantize_t<float>(st, X, rows, rows_pad, part, scale, inv, Xd, layout, nd, ldx); }
void launch_quantize_f64(hipStream_t st, const double* X, int64_t rows, int64_t rows_pad, double* part, double* scale,
                         double* inv, int8_t* Xd, int layout, int nd, int64_t ldx) { quantize_t<double>(st, X, rows, rows_pad, part, scale, inv, Xd, layout, nd, ldx); }

}  // namespace gpca
