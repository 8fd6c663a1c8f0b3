// Is a row panel that has just been streamed still worth anything to the next sweep over it?  (the Infinity-Cache
// panel fusion.)  NB panels of `rows` x 10 000 int8 genotypes; K2 (k_gtt_d) and K1 (k_gq_d) launched (a) on ONE panel over and over
// -- the panel is on the die from the previous launch if the part keeps it -- and (b) round-robin over all NB panels (NB x panel bytes
// well past the 256 MiB Infinity Cache: every launch streams from HBM).  Same launch shapes in both; nt and default-policy DMA loads.
//   hipcc --offload-arch=gfx950 -O3 -o kbench_mall kbench_mall.hip && ./kbench_mall [rows]
#include "../../genomic_pca_amd/csrc/gemm_i8.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, int64_t n, uint32_t seed, uint32_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = x & mask;
    }
}
template <int NT>
static void k2(const int8_t* G, int64_t ld8, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Yp, const gpca::Gtt8Plan& p) {
    hipLaunchKernelGGL((gpca::k_gtt_d<NT, 0>), dim3((unsigned)p.grid), dim3(256), sizeof(gpca::GqdSmem), 0, (const uint8_t*)G, ld8, Npad, Td, Yp, p.S, p.C, p.ngroups, p.W,
                       p.tasks_per_wg, p.strided, 1);
}
int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 16384;      // 16 384 rows x 10 496 B = 172 MB
    const int NB = 8;
    const int64_t N = 10000, Npad = 10240, ld8 = 10496, Mpad = rows, bytes = rows * ld8;
    std::vector<int8_t*> G(NB);
    for (int b = 0; b < NB; ++b) { CK(hipMalloc(&G[b], bytes)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G[b], bytes / 4, 1u + b, 0x01010101u); }
    const gpca::Gtt8Plan p2 = gpca::gtt8_plan_batched(Mpad, Npad, 2048);
    int8_t *Qd, *Td; double *qs, *ap, *Yp; float *r, *bb, *s, *T, *cp;
    CK(hipMalloc(&Qd, Npad * 32 * 4)); CK(hipMalloc(&Td, Mpad * 32 * 4)); CK(hipMalloc(&Yp, (size_t)p2.W * Npad * 32 * 8));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t*)Qd, Npad * 32, 2u, 0x3f3f3f3fu);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)Td, Mpad * 32, 3u, 0x3f3f3f3fu);
    CK(hipMalloc(&qs, 256)); CK(hipMemset(qs, 0, 256)); CK(hipMalloc(&r, rows * 4)); CK(hipMalloc(&bb, rows * 4)); CK(hipMalloc(&s, 128)); CK(hipMalloc(&T, rows * 32 * 4));
    CK(hipMemset(r, 0, rows * 4)); CK(hipMemset(bb, 0, rows * 4)); CK(hipMemset(s, 0, 128)); CK(hipMalloc(&cp, (rows / 32) * 32 * 4)); CK(hipMalloc(&ap, 1024 * 32 * 8));
    if (gpca::init_device_kernels_i8() != 0) { printf("LDS opt-in failed\n"); return 1; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gpca::k_gtt_d<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(gpca::GqdSmem)));
    const gpca::GqPlan p1{rows / 32, 1024};
    gpca::KernelOpts ko;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto&& launch, bool cycle) {
        const int n = 64;
        for (int i = 0; i < 16; ++i) launch(G[cycle ? i % NB : 0]);
        hipEventRecord(e0);
        for (int i = 0; i < n; ++i) launch(G[cycle ? i % NB : 0]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n * 1e3;
    };
    printf("panels of %lld rows x %lld samples = %.0f MB; K2 plan: %lld workgroups x %d tasks (W %d)\n", (long long)rows, (long long)N, bytes * 1e-6, (long long)p2.grid, p2.tasks_per_wg, p2.W);
    for (int pass = 0; pass < 2; ++pass) {
        const double a = run([&](const int8_t* g) { k2<1>(g, ld8, Mpad, Npad, Td, Yp, p2); }, false), b = run([&](const int8_t* g) { k2<1>(g, ld8, Mpad, Npad, Td, Yp, p2); }, true);
        const double c = run([&](const int8_t* g) { k2<0>(g, ld8, Mpad, Npad, Td, Yp, p2); }, false), d = run([&](const int8_t* g) { k2<0>(g, ld8, Mpad, Npad, Td, Yp, p2); }, true);
        const double e = run([&](const int8_t* g) { gpca::launch_gq_d(0, g, ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); }, false);
        const double f = run([&](const int8_t* g) { gpca::launch_gq_d(0, g, ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); }, true);
        printf("K2 nt loads:      same panel %.1f us = %.2f TB/s | %d panels round-robin %.1f us = %.2f TB/s\n", a, bytes / a * 1e-6, NB, b, bytes / b * 1e-6);
        printf("K2 default loads: same panel %.1f us = %.2f TB/s | %d panels round-robin %.1f us = %.2f TB/s\n", c, bytes / c * 1e-6, NB, d, bytes / d * 1e-6);
        printf("K1 (nt loads):    same panel %.1f us = %.2f TB/s | %d panels round-robin %.1f us = %.2f TB/s\n", e, bytes / e * 1e-6, NB, f, bytes / f * 1e-6);
    }
    return 0;
}
