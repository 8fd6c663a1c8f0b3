// Tall-skinny helpers for sketches wider than 64 columns (L = 128 or 256 padded columns; k + oversample up to 256).
//
// The reference clamps k only to min(samples, variants) and always adds 10 (main.rs:621-628, 636), and its authors sweep
// components_per_block to 50 (tests/sweep_run.py:61): --components 60 is an ordinary call there.  The two GEMMs of the exact-integer
// path are 32 columns wide and simply run ceil(l / 32) column blocks; the helpers around them (Gram, Cholesky + inverse, the right
// multiplications) are specialised for L = 32 / 64 in kernels.hip -- register-resident factorisations, one thread per output column.
// Wider sketches are rare and small next to the GEMM sweeps, so the versions here are plain: any L that is a multiple of 32, LDS
// tiles, one barrier per step, same arithmetic contracts (f64 accumulation, fixed summation orders, the same treatment of dependent
// and non-finite pivots as k_chol_inv).
#include "kernels.h"

namespace gpca {

// part[blk][a][c] = sum over the block's rows of X[n][a] X[n][c] (f64).  256 threads; the L x L outputs are dealt out 64 per thread
// in (a, c) order (L <= 128: one pass; L = 256: four passes over the tile); rows staged 16 at a time.
template <typename T>
__global__ __launch_bounds__(256) void k_gram_any(const T* __restrict__ X, int64_t rows, int64_t rpb, int L, double* __restrict__ part) {
    extern __shared__ double tile_any[];                 // [16][L]
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = (r0 + rpb < rows) ? r0 + rpb : rows;
    const int total = L * L;
    for (int base = 0; base < total; base += 256 * 64) {
        double acc[64];
#pragma unroll
        for (int o = 0; o < 64; ++o) acc[o] = 0.0;
        for (int64_t rb = r0; rb < r1; rb += 16) {
            __syncthreads();
            for (int e = threadIdx.x; e < 16 * L; e += 256) {
                const int rr = e / L, cc = e - rr * L;
                tile_any[e] = (rb + rr < r1) ? (double)X[(rb + rr) * L + cc] : 0.0;
            }
            __syncthreads();
#pragma unroll 4
            for (int o = 0; o < 64; ++o) {
                const int idx = base + o * 256 + (int)threadIdx.x;
                if (idx < total) {
                    const int a = idx / L, c = idx - a * L;
                    double s = acc[o];
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) s += tile_any[rr * L + a] * tile_any[rr * L + c];
                    acc[o] = s;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 64; ++o) {
            const int idx = base + o * 256 + (int)threadIdx.x;
            if (idx < total) part[(int64_t)blockIdx.x * total + idx] = acc[o];
        }
    }
}
void launch_gram_any_f64(hipStream_t st, const double* X, int64_t rows, int64_t rpb, int64_t parts, int L, double* part) {
    hipLaunchKernelGGL((k_gram_any<double>), dim3((unsigned)parts), dim3(256), sizeof(double) * 16 * L, st, X, rows, rpb, L, part);
}
void launch_gram_any_f32(hipStream_t st, const float* X, int64_t rows, int64_t rpb, int64_t parts, int L, double* part) {
    hipLaunchKernelGGL((k_gram_any<float>), dim3((unsigned)parts), dim3(256), sizeof(double) * 16 * L, st, X, rows, rpb, L, part);
}

// W (n x n used, pitch L, upper triangle) = R^T R; Z (L x L) = R^-1 (upper, zero elsewhere).  One workgroup, the matrices in global
// memory (L2-resident: 128 KiB at L = 128), a barrier per elimination step.  Same contracts as k_chol_inv (kernels.hip): a pivot that
// is not finite records (j + 1) in *flag and carries on with pivot 1; a pivot that is zero to rounding against the column's own
// squared norm (1e-13) drops its column from the basis (zero row in R and R^-1).  `work` = L x L doubles of scratch (R).
__global__ __launch_bounds__(256) void k_chol_inv_any(const double* __restrict__ Wg, int n, int L, double* __restrict__ Zg, double* __restrict__ work,
                                                      int* __restrict__ flag) {
    __shared__ double rowj[256];
    __shared__ double sh_dinv;
    const int tid = threadIdx.x;
    double* R = work;
    // R := upper triangle of W on the n x n block, identity outside
    for (int e = tid; e < L * L; e += 256) {
        const int r = e / L, c = e - r * L;
        R[e] = (r < n && c < n) ? ((c >= r) ? Wg[r * L + c] : 0.0) : ((r == c) ? 1.0 : 0.0);
        Zg[e] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            double piv = R[j * L + j];
            const double d0 = Wg[j * L + j];
            if (!isfinite(piv) || !isfinite(d0)) { atomicCAS(flag, 0, j + 1); piv = 1.0; R[j * L + j] = 1.0; }
            const bool dependent = !(piv > 1e-13 * d0);
            sh_dinv = dependent ? 0.0 : 1.0 / sqrt(piv);
        }
        __syncthreads();
        const double dinv = sh_dinv;
        // row j of R: R[j][c] = (c == j ? piv : R[j][c]) * dinv
        for (int c = j + tid; c < n; c += 256) { const double v = R[j * L + c] * dinv; R[j * L + c] = v; rowj[c & 255] = v; }
        __syncthreads();
        // trailing update: R[r][c] -= R[j][r] R[j][c] for j < r <= c < n   (L <= 256: rowj holds row j's entries by column)
        for (int r = j + 1; r < n; ++r) {
            const double a = rowj[r & 255];
            for (int c = r + tid; c < n; c += 256) R[r * L + c] -= a * rowj[c & 255];
        }
        __syncthreads();
    }
    // Z = R^-1 by back substitution, one column per thread: R x = e_c, x[k] = 0 for k > c; a dropped row (R[i][i] = 0) gives x[i] = 0
    for (int c = tid; c < n; c += 256) {
        for (int i = c; i >= 0; --i) {
            double acc = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k <= c; ++k) acc -= R[i * L + k] * Zg[k * L + c];
            const double d = R[i * L + i];
            Zg[i * L + c] = d != 0.0 ? acc / d : 0.0;
        }
    }
}
void launch_chol_inv_any(hipStream_t st, const double* W, int n, int L, double* Z, double* work, int* flag) {
    hipLaunchKernelGGL(k_chol_inv_any, dim3(1), dim3(256), 0, st, W, n, L, Z, work, flag);
}

// X[n][:] <- X[n][:] Z in place (f64): a workgroup takes kTailRows = 64 rows (the partial-array granularity of k_apply_right_tail,
// so that callers size and fold the partials the same way for every L), staged 32 at a time in LDS; with csum_part / amax_part: the
// workgroup's partial column sums and column abs-max of the result (k_finish_q folds them).
__global__ __launch_bounds__(256) void k_apply_right_any(double* __restrict__ X, int64_t rows, const double* __restrict__ Z, int L,
                                                         double* __restrict__ csum_part, double* __restrict__ amax_part) {
    extern __shared__ double xs_any[];                    // [32][L]
    const int cc = threadIdx.x;                           // (L <= 256: one output column per thread)
    double cs = 0.0, am = 0.0;
    for (int sub = 0; sub < 64; sub += 32) {
        const int64_t n0 = (int64_t)blockIdx.x * 64 + sub;
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * L; e += 256) {
            const int rr = e / L, c2 = e - rr * L;
            xs_any[e] = (n0 + rr < rows) ? X[(n0 + rr) * L + c2] : 0.0;
        }
        __syncthreads();
        if (cc < L)
            for (int rr = 0; rr < 32; ++rr) {
                double a = 0.0;
                for (int j = 0; j < L; ++j) a += xs_any[rr * L + j] * Z[j * L + cc];
                if (n0 + rr < rows) { X[(n0 + rr) * L + cc] = a; cs += a; am = fmax(am, fabs(a)); }
            }
    }
    if (csum_part && cc < L) { csum_part[(int64_t)blockIdx.x * L + cc] = cs; amax_part[(int64_t)blockIdx.x * L + cc] = am; }
}
void launch_apply_right_any(hipStream_t st, double* X, int64_t rows, int64_t parts, int L, const double* Z, double* csum_part, double* amax_part) {
    hipLaunchKernelGGL(k_apply_right_any, dim3((unsigned)parts), dim3(256), sizeof(double) * 32 * L, st, X, rows, Z, L, csum_part, amax_part);
}

// out[n][kc] = sum_j X[row(n)][j] Z[j][kc] for kc < K: one thread per (row, kc) pair, rows gathered through row_ids when given
template <typename TX>
__global__ __launch_bounds__(256) void k_rightmul_any(const TX* __restrict__ X, const int64_t* __restrict__ row_ids, int64_t nrows, int L,
                                                      const double* __restrict__ Z, int K, double* __restrict__ out64, float* __restrict__ out32) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * K) return;
    const int64_t n = e / K;
    const int kc = (int)(e - n * K);
    const int64_t src = row_ids ? row_ids[n] : n;
    double a = 0.0;
    for (int j = 0; j < L; ++j) a += (double)X[src * L + j] * Z[j * K + kc];
    if (out64) out64[e] = a;
    if (out32) out32[e] = (float)a;
}
void launch_rightmul_any_f64(hipStream_t st, const double* X, int64_t rows, int L, const double* Z, int K, double* out64, float* out32) {
    const int64_t total = rows * K;
    if (total > 0) hipLaunchKernelGGL((k_rightmul_any<double>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, (const int64_t*)nullptr, rows, L, Z, K, out64, out32);
}
void launch_rightmul_any_gather_f32(hipStream_t st, const float* X, const int64_t* row_ids, int64_t nrows, int L, const double* Z, int K, float* out32) {
    const int64_t total = nrows * K;
    if (total > 0) hipLaunchKernelGGL((k_rightmul_any<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, row_ids, nrows, L, Z, K, (double*)nullptr, out32);
}

}  // namespace gpca
