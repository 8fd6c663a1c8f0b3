// The one-wave Cholesky / inverse of rounds 1-3 (an element of another lane's column by v_readlane) beside the current kernel (a finished
// row through LDS), alternating in ONE process on the same matrices: per-launch time of each and whether the bits agree.  The old kernel
// is kept here, in the harness, for this comparison only (git show 8ae9932~1:genomic_pca_amd/csrc/kernels.hip).
//   hipcc --offload-arch=gfx950 -O3 -o kbench_chol kbench_chol.hip && ./kbench_chol
#include "../../genomic_pca_amd/csrc/kernels.hip"
#include "../../genomic_pca_amd/csrc/wide_sketch.hip"
#include <cstdio>
#include <vector>
#include <cstring>
namespace gpca {
constexpr double kCholRankTol_old = 1e-13;   // relative to the column's own squared norm (Gram rounding is ~32 x 2.2e-16)
template <int LANE>
__device__ __forceinline__ double lane_bcast_old(double v, double& dep) {
    int lo, hi;
    // the compiler's hazard recogniser does not look inside the string: the wait states a VALU-written VGPR needs before
    // v_readlane reads it, and a VALU-written SGPR needs before a VALU reads it as an operand, are supplied here
    asm volatile("s_nop 1\n\tv_readlane_b32 %0, %3, %5\n\tv_readlane_b32 %1, %4, %5\n\ts_nop 1"
                 : "=s"(lo), "=s"(hi), "+v"(dep) : "v"(__double2loint(v)), "v"(__double2hiint(v)), "n"(LANE));
    return __hiloint2double(hi, lo);
}
// 1 / sqrt(x) in f64 from the hardware estimate and two Newton steps (the correctly rounded sqrt + divide pair costs
// ~500 dependent cycles per pivot; this chain ~100)
__device__ __forceinline__ double rsqrt_nr_old(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
template <int NN, int J, int R>
struct CholRow_old {   // col[r] -= R[j][r] * R[j][c] for r = R .. NN-1 (compile-time lanes)
    static __device__ __forceinline__ void run(double (&col)[NN]) {
        if constexpr (R < NN) { const double a = lane_bcast_old<R>(col[J], col[R]); col[R] -= a * col[J]; CholRow_old<NN, J, R + 1>::run(col); }
    }
};
template <int NN, int J>
struct CholStep_old {
    static __device__ __forceinline__ void run(double (&col)[NN], double (&dinv)[NN], double& diag0, int c, int* flag) {
        if constexpr (J < NN) {
            double piv = lane_bcast_old<J>(col[J], col[J]);
            const double d0 = lane_bcast_old<J>(diag0, diag0);          // column J's own squared norm before the elimination
            if (!isfinite(piv) || !isfinite(d0)) {
                if (c == 0) atomicCAS(flag, 0, J + 1);
                piv = 1.0;
            }
            // Column J lies in the span of the columns before it (what is left of its squared norm is rounding noise, possibly
            // negative): a sketch wider than the rank of the matrix -- k + oversample = N samples of centred rows have rank N - 1.
            // The column leaves the basis: row J of R and of R^-1 become zero, so Q's column J is zero and every later product
            // carries a zero column (zero singular value) instead of the call failing.
            const bool dependent = !(piv > kCholRankTol_old * d0);
            dinv[J] = dependent ? 0.0 : rsqrt_nr_old(piv);
            col[J] = (c == J) ? piv * dinv[J] : col[J] * dinv[J];
            CholRow_old<NN, J, J + 1>::run(col);
            CholStep_old<NN, J + 1>::run(col, dinv, diag0, c, flag);
        }
    }
};
template <int NN, int I, int K>
struct InvRow_old {
    static __device__ __forceinline__ void run(const double (&col)[NN], const double (&x)[NN], double& acc) {
        if constexpr (K < NN) { const double a = lane_bcast_old<K>(col[I], acc); acc -= a * x[K]; InvRow_old<NN, I, K + 1>::run(col, x, acc); }
    }
};
template <int NN, int I>
struct InvStep_old {
    static __device__ __forceinline__ void run(const double (&col)[NN], double (&x)[NN], const double (&dinv)[NN], int c) {
        if constexpr (I >= 0) {
            double acc = (c == I) ? 1.0 : 0.0;
            InvRow_old<NN, I, I + 1>::run(col, x, acc);
            x[I] = acc * dinv[I];
            InvStep_old<NN, I - 1>::run(col, x, dinv, c);
        }
    }
};
template <int NN>
__global__ __launch_bounds__(64) void k_chol_inv_old(const double* __restrict__ Wg, int n, double* __restrict__ Zg,
                                                 int* __restrict__ flag) {
    const int c = threadIdx.x;
    double col[NN], x[NN], dinv[NN];
#pragma unroll
    for (int r = 0; r < NN; ++r) col[r] = Wg[r * NN + (c & (NN - 1))];      // (unconditional: the loads stay in flight together)
#pragma unroll
    for (int r = 0; r < NN; ++r) col[r] = (r < n && c < n) ? col[r] : ((r == c) ? 1.0 : 0.0);
    double diag0 = 0.0;
#pragma unroll
    for (int r = 0; r < NN; ++r) diag0 = (r == c) ? col[r] : diag0;
    CholStep_old<NN, 0>::run(col, dinv, diag0, c, flag);
    InvStep_old<NN, NN - 1>::run(col, x, dinv, c);
    if (c < NN) {
#pragma unroll
        for (int i = 0; i < NN; ++i) Zg[i * NN + c] = (i < n && c < n) ? x[i] : 0.0;
    }
}
}  // namespace gpca

// The wide-sketch (L = 128) factorisation in its first form -- one workgroup on global memory, a serial loop over the rows of every
// elimination step -- kept here for the comparison with the LDS-resident kernel of csrc/wide_sketch.hip.
namespace gpca {
__global__ __launch_bounds__(256) void k_chol_inv_any_old(const double* __restrict__ Wg, int n, int L, double* __restrict__ Zg, double* __restrict__ work,
                                                          int* __restrict__ flag) {
    __shared__ double rowj[256];
    __shared__ double sh_dinv;
    const int tid = threadIdx.x;
    double* R = work;
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
        for (int c = j + tid; c < n; c += 256) { const double v = R[j * L + c] * dinv; R[j * L + c] = v; rowj[c & 255] = v; }
        __syncthreads();
        for (int r = j + 1; r < n; ++r) {
            const double a = rowj[r & 255];
            for (int c = r + tid; c < n; c += 256) R[r * L + c] -= a * rowj[c & 255];
        }
        __syncthreads();
    }
    for (int c = tid; c < n; c += 256) {
        for (int i = c; i >= 0; --i) {
            double acc = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k <= c; ++k) acc -= R[i * L + k] * Zg[k * L + c];
            const double d = R[i * L + i];
            Zg[i * L + c] = d != 0.0 ? acc / d : 0.0;
        }
    }
}
}  // namespace gpca
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int NN>
static int run(double scale = 1.0) {
    std::vector<double> A(NN * NN), W(NN * NN, 0.0);
    unsigned x = 12345u;
    for (auto& a : A) { x = x * 1664525u + 1013904223u; a = ((x >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < NN; ++i) for (int j = 0; j < NN; ++j) { double s = (i == j) ? 0.5 : 0.0; for (int k = 0; k < NN; ++k) s += A[k * NN + i] * A[k * NN + j]; W[i * NN + j] = s * scale; }
    if (scale != 1.0) printf("(matrix scaled by %.1e) ", scale);
    double *dW, *dZ0, *dZ1; int* flag;
    CK(hipMalloc(&dW, NN * NN * 8)); CK(hipMalloc(&dZ0, NN * NN * 8)); CK(hipMalloc(&dZ1, NN * NN * 8)); CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    CK(hipMemcpy(dW, W.data(), NN * NN * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double t[2] = {0, 0};
    const int n = NN - 2;
    for (int rep = 0; rep < 6; ++rep)
        for (int v = 0; v < 2; ++v) {
            auto go = [&]() {
                if (v == 0) hipLaunchKernelGGL(gpca::k_chol_inv_old<NN>, dim3(1), dim3(64), 0, 0, dW, n, dZ0, flag);
                else hipLaunchKernelGGL(gpca::k_chol_inv<NN>, dim3(1), dim3(64), 0, 0, dW, n, dZ1, flag);
            };
            go();
            hipEventRecord(e0);
            for (int it = 0; it < 50; ++it) go();
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); t[v] += ms / 50 * 1e3;
        }
    std::vector<double> Z0(NN * NN), Z1(NN * NN);
    CK(hipMemcpy(Z0.data(), dZ0, NN * NN * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Z1.data(), dZ1, NN * NN * 8, hipMemcpyDeviceToHost));
    printf("k_chol_inv<%d> (n = %d): readlane form %.2f us, rows through LDS %.2f us per launch (back to back, launch overhead included); results %s\n",
           NN, n, t[0] / 6, t[1] / 6, memcmp(Z0.data(), Z1.data(), NN * NN * 8) == 0 ? "bit-identical" : "DIFFER");
    return 0;
}

static int run_wide(int n) {
    const int L = 128;
    std::vector<double> A(L * L), W(L * L, 0.0);
    unsigned x = 777u + n;
    for (auto& a : A) { x = x * 1664525u + 1013904223u; a = ((x >> 8) & 0xffff) / 65536.0 - 0.5; }
    // (column n - 3 = column 1 + column 2: one dependent pivot, the case the rank contract is for)
    for (int k = 0; k < L; ++k) A[k * L + n - 3] = A[k * L + 1] + A[k * L + 2];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < L; ++k) s += A[k * L + i] * A[k * L + j]; W[i * L + j] = s; }
    double *dW, *dZ0, *dZ1, *dwork; int* flag;
    CK(hipMalloc(&dW, L * L * 8)); CK(hipMalloc(&dZ0, L * L * 8)); CK(hipMalloc(&dZ1, L * L * 8)); CK(hipMalloc(&dwork, L * L * 8)); CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    CK(hipMemcpy(dW, W.data(), L * L * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double t[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep)
        for (int v = 0; v < 2; ++v) {
            auto go = [&]() {
                if (v == 0) hipLaunchKernelGGL(gpca::k_chol_inv_any_old, dim3(1), dim3(256), 0, 0, dW, n, L, dZ0, dwork, flag);
                else (void)gpca::launch_chol_inv_any(0, dW, n, L, dZ1, flag);
            };
            go();
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) go();
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); t[v] += ms / 10 * 1e3;
        }
    std::vector<double> Z0(L * L), Z1(L * L);
    CK(hipMemcpy(Z0.data(), dZ0, L * L * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Z1.data(), dZ1, L * L * 8, hipMemcpyDeviceToHost));
    int zero_rows = 0;
    for (int i = 0; i < n; ++i) { bool z = true; for (int c = 0; c < n; ++c) z = z && Z1[i * L + c] == 0.0; zero_rows += z; }
    printf("k_chol_inv_any (L = 128, n = %d): global-memory form %.1f us, LDS-resident %.1f us per launch; results %s; dropped columns %d\n",
           n, t[0] / 3, t[1] / 3, memcmp(Z0.data(), Z1.data(), L * L * 8) == 0 ? "bit-identical" : "DIFFER", zero_rows);
    return 0;
}
int main() { return run<32>() || run<64>() || run<64>(1e23) || run<64>(1e-23) || run_wide(70) || run_wide(100) || run_wide(128); }
