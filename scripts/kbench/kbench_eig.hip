// The device eigen step of gpca_rsvd (csrc/small_eig.hip) alone: per-launch time back to back, the phases of one launch from
// s_memrealtime stamps (fold + prescale | Jacobi sweeps | sort + outputs), and the residual of what it returns.
//   hipcc --offload-arch=gfx950 -O3 -DGPCA_EIG_STAMP=1 -o kbench_eig kbench_eig.hip && ./kbench_eig
#include "../../genomic_pca_amd/csrc/small_eig.hip"
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static int run(int n, int k, int slices) {
    const int L = n <= 32 ? 32 : (n <= 64 ? 64 : 128);
    const int ns = slices > 0 ? slices : 1;      // (slices = 0: the matrix itself at src)
    std::vector<double> B((size_t)3 * n * n), W((size_t)ns * L * L, 0.0), A((size_t)n * n);
    unsigned x = 12345u + n;
    for (auto& b : B) { x = x * 1664525u + 1013904223u; b = ((x >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0.0; for (int r = 0; r < 3 * n; ++r) s += B[(size_t)r * n + i] * B[(size_t)r * n + j] * (1.0 + 50.0 * (r < 3));   // a few strong directions
        A[(size_t)i * n + j] = s * 1e6;
        for (int sl = 0; sl < ns; ++sl) W[(size_t)sl * L * L + (size_t)i * L + j] = s * 1e6 / ns;
    }
    double *dW, *dZ, *dR, *dV;
    CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dZ, 2 * (size_t)L * k * 8)); CK(hipMalloc(&dR, gpca::kEigResCount * 8)); CK(hipMalloc(&dV, (size_t)n * n * 8));
    CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    if (gpca::init_device_kernels_eig() != 0) { printf("attribute failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { gpca::launch_small_eigh(0, dW, slices, n, L, k, 0, (double)(1000 - 1), nullptr, dZ, dR, dV); };
    go(); CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) go();
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> R(gpca::kEigResCount), V((size_t)n * n);
    CK(hipMemcpy(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(V.data(), dV, V.size() * 8, hipMemcpyDeviceToHost));
    double resid = 0.0, orth = 0.0;
    for (int i = 0; i < n; ++i) for (int c = 0; c < n; ++c) {
        double av = 0.0, vv = 0.0;
        for (int j = 0; j < n; ++j) { av += A[(size_t)i * n + j] * V[(size_t)j * n + c]; vv += V[(size_t)j * n + i] * V[(size_t)j * n + c]; }
        resid = std::fmax(resid, std::fabs(av - V[(size_t)i * n + c] * R[gpca::kEigResW + c]));
        orth = std::fmax(orth, std::fabs(vv - (i == c ? 1.0 : 0.0)));
    }
    unsigned long long sp[20];
    CK(hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_eig_stamp), sizeof sp));
    auto us = [&](int a, int b) { return (double)(sp[2 * b] - sp[2 * a]) * 0.01; };
    const double ghz = (double)(sp[2 * 4 + 1] - sp[1]) / ((double)(sp[2 * 4] - sp[0]) * 10.0);
    printf("k_small_eigh n = %3d (L = %3d, %2d slices): %.1f us per launch back to back | stamps: fold %.1f, sweeps %.1f, sort+out %.1f us, shader clock %.2f GHz, %llu Jacobi sweeps, %llu steps = %.0f cycles each | "
           "w0 = %.6e, residual / w0 = %.1e, |V^T V - I| = %.1e, cap flag %.0f\n", n, L, slices, ms / 50 * 1e3,
           us(0, 1), us(1, 3), us(3, 4), ghz, sp[10], sp[11], (double)(sp[2 * 3 + 1] - sp[2 * 1 + 1]) / (double)(sp[11] ? sp[11] : 1), R[gpca::kEigResW], resid / R[gpca::kEigResW], orth, R[gpca::kEigResFlag + 1]);
    if (n > 32 && sp[2 * 2] > sp[2 * 1] && sp[2 * 2] < sp[2 * 3])      // the QL form stamps the end of its tridiagonalisation
        printf("    (tridiagonalisation + accumulation of Q %.1f us, QL chain with its rotations applied %.1f us = %.0f cycles per rotation; chain waited for ring room %llu polls; applying wave: %llu batches, %llu empty polls)\n",
               us(1, 2), us(2, 3), (double)(sp[2 * 3 + 1] - sp[2 * 2 + 1]) / (double)(sp[11] ? sp[11] : 1), sp[12], sp[13], sp[14]);
    hipFree(dW); hipFree(dZ); hipFree(dR); hipFree(dV);
    return 0;
}
// accuracy of v_rsq_f64 / v_rcp_f64 with 0, 1, 2 Newton steps (how many does the rotation chain need?)
__global__ void k_probe_rsq(double* out) {
    const int t = threadIdx.x + blockIdx.x * 256;
    const double x = 0.3 + 1e-4 * t + 1e-9 * t * t;
    double y = __builtin_amdgcn_rsq(x);
    const double e0 = fabs(y * y * x - 1.0) * 0.5;
    y = y * (1.5 - 0.5 * x * y * y);
    const double e1 = fabs(y * y * x - 1.0) * 0.5;
    y = y * (1.5 - 0.5 * x * y * y);
    const double e2 = fabs(y * y * x - 1.0) * 0.5;
    double r = __builtin_amdgcn_rcp(x);
    const double r0 = fabs(fma(r, x, -1.0));
    r = fma(r, fma(-x, r, 1.0), r);
    const double r1 = fabs(fma(r, x, -1.0));
    out[t * 5 + 0] = e0; out[t * 5 + 1] = e1; out[t * 5 + 2] = e2; out[t * 5 + 3] = r0; out[t * 5 + 4] = r1;
}
static int probe() {
    double* d; CK(hipMalloc(&d, 4096 * 5 * 8));
    hipLaunchKernelGGL(k_probe_rsq, dim3(16), dim3(256), 0, 0, d);
    std::vector<double> h(4096 * 5); CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double m[5] = {0, 0, 0, 0, 0};
    for (int t = 0; t < 4096; ++t) for (int j = 0; j < 5; ++j) m[j] = std::fmax(m[j], h[t * 5 + j]);
    printf("v_rsq_f64 relative error: raw %.2e, 1 Newton step %.2e, 2 steps %.2e | v_rcp_f64: raw %.2e, 1 step %.2e\n", m[0], m[1], m[2], m[3], m[4]);
    return 0;
}
int main() { return probe() || run(30, 20, 16) || run(30, 20, 0) || run(50, 40, 16) || run(64, 40, 16) || run(100, 90, 4) || run(128, 100, 4); }
