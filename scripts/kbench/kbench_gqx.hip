// Does matrix-core power (DVFS) limit the HBM-bound int8-resident K1 kernel (k_gq_x)?  Times the product kernel on a
// 1M x 10k problem with random digit planes vs all-zero digit planes (same memory traffic, no operand toggling), and
// with 4 / 2 / 1 of the planes non-zero.  Not product code.
#include "../../genomic_pca_amd/csrc/gemm_i8.hip"
#include <cstdio>
#include <vector>
__global__ void k_fill_g(uint32_t* p, int64_t n, uint32_t seed) {     // bytes in {0,1,2}
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        uint32_t o = 0;
        for (int b = 0; b < 4; ++b) { uint32_t v = (x >> (8 * b)) & 3u; if (v == 3) v = 0; o |= v << (8 * b); }
        p[i] = o;
    }
}
// digit planes [step][digit][64 lanes][16 B]: random signed bytes in planes < nz, zero elsewhere
__global__ void k_fill_q(uint32_t* p, int64_t n, int nz) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int digit = (int)((i / 256) % 4);
        uint32_t x = (uint32_t)i * 2654435761u ^ 77u; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = digit < nz ? (x & 0x7f7f7f7fu) ^ ((x >> 1) & 0x80808080u & ((x & 0x40404040u) << 1)) : 0u;
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int64_t M = 1000064, Npad = 10240, ldg = Npad;
    const int waves = argc > 1 ? atoi(argv[1]) : 1024;
    int8_t *G, *Qd; double* qs; float *r, *b, *s, *T, *cp; double* ap;
    CK(hipMalloc(&G, M * ldg)); hipLaunchKernelGGL(k_fill_g, dim3(8192), dim3(256), 0, 0, (uint32_t*)G, M * ldg / 4, 1u);
    CK(hipMalloc(&Qd, Npad * 32 * 4));
    CK(hipMalloc(&qs, 32 * 8)); CK(hipMemset(qs, 0, 32 * 8));
    CK(hipMalloc(&r, M * 4)); CK(hipMalloc(&b, M * 4)); CK(hipMalloc(&s, 32 * 4)); CK(hipMalloc(&T, M * 32 * 4));
    CK(hipMemset(r, 0, M * 4)); CK(hipMemset(b, 0, M * 4)); CK(hipMemset(s, 0, 32 * 4));
    CK(hipMalloc(&cp, (M / 32) * 32 * 4));   /* one c partial per 32-row unit (not per wave: the kernels write cunit[unit][32]) */ CK(hipMalloc(&ap, waves * 32 * 8));
    gpca::GqPlan plan{M / 32, waves};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nz = 4; nz >= 0; --nz) {
        if (nz == 3) continue;
        hipLaunchKernelGGL(k_fill_q, dim3(1024), dim3(256), 0, 0, (uint32_t*)Qd, Npad * 32, nz);
        for (int it = 0; it < 20; ++it) gpca::launch_gq_x(0, G, ldg, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1);
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) gpca::launch_gq_x(0, G, ldg, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("k_gq_x waves %d, %d non-zero digit planes: %.3f ms  = %.2f TB/s of genotype bytes\n", waves, nz, ms, (double)M * ldg / ms * 1e-9);
        fflush(stdout);
    }
    return 0;
}
