// Tall-skinny helpers for sketches wider than 64 columns (L = 128 padded columns = kMaxSketchCols; k + oversample up to 128, i.e. 118
// components with the reference's fixed oversampling of 10).
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
// in (a, c) order (L <= 128: one pass); rows staged 16 at a time.
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

// W (n x n used, pitch L, upper triangle) = R^T R; Z (L x L) = R^-1 (upper, zero elsewhere).  One workgroup, L <= 128, everything in
// LDS: one (L x (L + 1)) square of doubles (129 KiB at L = 128, opted in) holds R in its upper triangle and, once the factor is
// complete, the transpose of R^-1 in its strictly lower one (the inverse is upper triangular too: its column c, above the diagonal,
// is row c of the lower triangle -- the thread that owns the column walks its own row) with the inverse's diagonal beside it.
// Same contracts as k_chol_inv (kernels.hip): a pivot that is not finite records (j + 1) in *flag and carries on with pivot 1; a pivot
// that is zero to rounding against the column's own squared norm (1e-13) drops its column from the basis (zero row in R and R^-1).
// Every element sees the arithmetic of the first form of this kernel (one workgroup reading and writing global memory, a serial loop
// over the rows of every elimination step, milliseconds at n = 100) in the same order: R[r][c] -= R[j][r] R[j][c] for j ascending,
// the substitution sums for k ascending.
//   * elimination step j: thread (c = tid mod 128, half = tid / 128) updates column c of the rows j + 1 + half, j + 3 + half, ...
//     (R[j][r] is a broadcast read, column c of consecutive rows is conflict-free at the odd pitch); two barriers per step.
//   * back substitution with the row index i as the uniform outer loop: thread c (c >= i) forms x_c[i] from R[i][k] (broadcast) and its
//     own x_c[k] (row c of the lower triangle).
// scripts/kbench/kbench_chol.hip `wide` times both forms.
constexpr int kCholAnyMaxL = 128;
__global__ __launch_bounds__(256) void k_chol_inv_any(const double* __restrict__ Wg, int n, int L, double* __restrict__ Zg, int* __restrict__ flag) {
    extern __shared__ double chol_any_lds[];              // [L][L + 1] + [L] (diagonal of the inverse) + [1]
    const int P = L + 1;
    double* R = chol_any_lds;
    double* zdiag = R + (size_t)L * P;
    double* sh_dinv = zdiag + L;
    const int tid = threadIdx.x;
    // R := upper triangle of W on the n x n block, identity outside
    for (int e = tid; e < L * L; e += 256) {
        const int r = e / L, c = e - r * L;
        R[r * P + c] = (r < n && c < n) ? ((c >= r) ? Wg[r * L + c] : 0.0) : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    const int cc = tid & (kCholAnyMaxL - 1), half = tid >> 7;
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            double piv = R[j * P + j];
            const double d0 = Wg[j * L + j];
            if (!isfinite(piv) || !isfinite(d0)) { atomicCAS(flag, 0, j + 1); piv = 1.0; R[j * P + j] = 1.0; }
            const bool dependent = !(piv > 1e-13 * d0);
            *sh_dinv = dependent ? 0.0 : 1.0 / sqrt(piv);
        }
        __syncthreads();
        const double dinv = *sh_dinv;
        if (half == 0 && cc >= j && cc < n) R[j * P + cc] *= dinv;          // row j of the factor
        __syncthreads();
        if (cc > j && cc < n) {
            const double rc = R[j * P + cc];
            for (int r = j + 1 + half; r <= cc; r += 2) R[r * P + cc] -= R[j * P + r] * rc;
        }
        // (the next step's pivot R[j+1][j+1] and row j + 1 were written by this step: the barrier at the top of the loop body orders them)
        __syncthreads();
    }
    // Z = R^-1 by back substitution: R x = e_c, x[k] = 0 for k > c; a dropped row (R[i][i] = 0) gives x[i] = 0
    if (tid < n) zdiag[tid] = R[tid * P + tid] != 0.0 ? 1.0 / R[tid * P + tid] : 0.0;
    __syncthreads();
    if (half == 0 && cc < n) {
        double* xrow = R + (size_t)cc * P;                // x_c[k] for k < c lives at [c][k]
        for (int i = cc - 1; i >= 0; --i) {
            // k ascends from i + 1 to c; the last term is the column's own diagonal entry x_c[c]
            double acc = 0.0;
            for (int k = i + 1; k < cc; ++k) acc -= R[i * P + k] * xrow[k];
            acc -= R[i * P + cc] * zdiag[cc];
            const double d = R[i * P + i];
            xrow[i] = d != 0.0 ? acc / d : 0.0;
        }
    }
    __syncthreads();
    for (int e = tid; e < L * L; e += 256) {
        const int r = e / L, c = e - r * L;
        Zg[e] = (r < n && c < n) ? (r < c ? R[c * P + r] : (r == c ? zdiag[r] : 0.0)) : 0.0;
    }
}
int launch_chol_inv_any(hipStream_t st, const double* W, int n, int L, double* Z, int* flag) {
    if (L > kCholAnyMaxL || n > L) return (int)hipErrorInvalidValue;
    const size_t lds = sizeof(double) * ((size_t)L * (L + 1) + L + 1);
    static bool opted = false;
    if (!opted) { if (hipFuncSetAttribute((const void*)k_chol_inv_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * ((size_t)kCholAnyMaxL * (kCholAnyMaxL + 1) + kCholAnyMaxL + 1))) != hipSuccess) return (int)hipErrorInvalidValue; opted = true; }
    hipLaunchKernelGGL(k_chol_inv_any, dim3(1), dim3(256), lds, st, W, n, L, Z, flag);
    return 0;
}

// X[n][:] <- X[n][:] Z in place (f64): a workgroup takes kTailRows = 64 rows (the partial-array granularity of k_apply_right_tail,
// so that callers size and fold the partials the same way for every L), staged 32 at a time in LDS; with csum_part / amax_part: the
// workgroup's partial column sums and column abs-max of the result (k_finish_q folds them).
__global__ __launch_bounds__(256) void k_apply_right_any(double* __restrict__ X, int64_t rows, const double* __restrict__ Z, int L,
                                                         double* __restrict__ csum_part, double* __restrict__ amax_part) {
    extern __shared__ double xs_any[];                    // [32][L]
    const int cc = threadIdx.x;                           // (L <= 128 < 256: one output column per thread)
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
