// Does it pay to pick the genotype buffer among several allocations?  (r1 section 7, r3 section 2: plain hipMalloc buffers of the C2 matrix
// stream at 6.68 ... 6.95 TB/s depending on where they land; two engines of a process differ by up to 8 % on K1 and K2 alike.)
// NB buffers of the C2 int8 matrix side by side; for each: a linear nt read probe, the product's K1 (k_gq_d) and K2 (k_gtt_d) on it,
// round-robin over the buffers, several passes.  The question: how wide is the spread of K1 / K2, and does the cheap probe rank it?
//   hipcc --offload-arch=gfx950 -O3 -o kbench_pick kbench_pick.hip && ./kbench_pick [NB]
#include "../../genomic_pca_amd/csrc/gemm_i8.hip"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, int64_t n, uint32_t seed, uint32_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = x & mask;
    }
}
typedef int i32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_linear(const i32x4v* __restrict__ p, int64_t n16, int* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    i32x4v acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n16; i += 4 * stride) {
        i32x4v a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        i32x4v c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 6;
    const int64_t M = 1000064, N = 10000, Npad = 10240, ld8 = 10496, Mpad = M;
    const int64_t bytes = M * ld8;
    std::vector<int8_t*> G(NB);
    for (int b = 0; b < NB; ++b) { CK(hipMalloc(&G[b], bytes)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G[b], bytes / 4, 1u, 0x01010101u); }
    int8_t *Qd, *Td; double *qs, *ap, *Yp; float *r, *bb, *s, *T, *cp; int* out;
    const gpca::Gtt8Plan p2 = gpca::gtt8_plan_batched(Mpad, Npad, 2048);
    CK(hipMalloc(&Qd, Npad * 32 * 4)); CK(hipMalloc(&Td, Mpad * 32 * 4)); CK(hipMalloc(&Yp, (size_t)p2.W * Npad * 32 * 8)); CK(hipMalloc(&out, 4));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t*)Qd, Npad * 32, 2u, 0x3f3f3f3fu);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)Td, Mpad * 32, 3u, 0x3f3f3f3fu);
    CK(hipMalloc(&qs, 256)); CK(hipMemset(qs, 0, 256)); CK(hipMalloc(&r, M * 4)); CK(hipMalloc(&bb, M * 4)); CK(hipMalloc(&s, 128)); CK(hipMalloc(&T, M * 32 * 4));
    CK(hipMemset(r, 0, M * 4)); CK(hipMemset(bb, 0, M * 4)); CK(hipMemset(s, 0, 128)); CK(hipMalloc(&cp, (M / 32) * 32 * 4)); CK(hipMalloc(&ap, 1024 * 32 * 8));
    if (gpca::init_device_kernels_i8() != 0) { printf("LDS opt-in failed\n"); return 1; }
    const gpca::GqPlan p1{M / 32, 1024};
    gpca::KernelOpts ko;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time_us = [&](auto&& f, int n) { f(); hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n * 1e3; };
    std::vector<std::vector<double>> t(NB, std::vector<double>(3, 0.0));
    const int passes = 3;
    for (int pass = 0; pass < passes; ++pass)
        for (int b = 0; b < NB; ++b) {
            t[b][0] += time_us([&] { hipLaunchKernelGGL(k_linear, dim3(16384), dim3(256), 0, 0, (const i32x4v*)G[b], bytes / 16, out); }, 3);
            t[b][1] += time_us([&] { gpca::launch_gq_d(0, G[b], ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); }, 6);
            t[b][2] += time_us([&] { gpca::launch_gtt_d(0, G[b], ld8, Mpad, Npad, Td, Yp, p2, ko); }, 6);
        }
    {   // Does it matter where K2's OTHER buffers land?  Free every genotype buffer but the fastest, allocate a second set of T' planes and
        // partial tiles (they land in the holes), and run K2 on the kept buffer with either set.
        int best = 0; for (int b = 1; b < NB; ++b) if (t[b][0] < t[best][0]) best = b;
        auto k12 = [&](const char* what) {
            double a1 = 0, a2 = 0;
            for (int pass = 0; pass < 4; ++pass) {
                a1 += time_us([&] { gpca::launch_gq_d(0, G[best], ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); }, 6);
                a2 += time_us([&] { gpca::launch_gtt_d(0, G[best], ld8, Mpad, Npad, Td, Yp, p2, ko); }, 6);
            }
            printf("kept buffer %d, %-58s K1 %.1f us | K2 %.1f us\n", best, what, a1 / 4, a2 / 4);
        };
        k12("every candidate still allocated:");
        if (argc > 2) {   // how large does a free have to be?   ./kbench_pick NB sizes   (a fresh allocation of each size, written once, freed, K1 / K2 right after)
            for (const double gb : {0.016, 0.25, 1.0, 4.0, 16.0}) {
                int8_t* q = nullptr; const size_t nb = (size_t)(gb * (1 << 30));
                CK(hipMalloc(&q, nb)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)q, (int64_t)(nb / 4), 1u, 0x01010101u);
                for (int i = 0; i < 300; ++i) { gpca::launch_gq_d(0, G[best], ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); gpca::launch_gtt_d(0, G[best], ld8, Mpad, Npad, Td, Yp, p2, ko); }
                CK(hipDeviceSynchronize());
                char what[96]; snprintf(what, sizeof what, "settled (1 s of load), then");  k12(what);
                CK(hipFree(q));
                snprintf(what, sizeof what, "... right after hipFree of %.3f GiB:", gb);  k12(what);
            }
        }
        {   // Is it the idle gaps?  A product call = 3 x (K2, K1) with two host synchronisations (the l x l eigenproblem, the end of the call).
            // The same six launches per "call" here, timed by events around each launch, with and without a host sync + ~80 us pause per call.
            hipEvent_t ea[6], eb[6];
            for (int i = 0; i < 6; ++i) { hipEventCreateWithFlags(&ea[i], hipEventDisableSystemFence); hipEventCreateWithFlags(&eb[i], hipEventDisableSystemFence); }
            for (int mode = 0; mode < 3; ++mode) {
                double k1 = 0, k2 = 0; int n = 0;
                const auto t_begin = std::chrono::steady_clock::now();
                for (int call = 0; call < 200; ++call) {
                    for (int i = 0; i < 3; ++i) {
                        hipEventRecord(ea[2 * i], 0); gpca::launch_gtt_d(0, G[best], ld8, Mpad, Npad, Td, Yp, p2, ko); hipEventRecord(eb[2 * i], 0);
                        hipEventRecord(ea[2 * i + 1], 0); gpca::launch_gq_d(0, G[best], ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); hipEventRecord(eb[2 * i + 1], 0);
                        if (mode == 2 && i == 1) { hipStreamSynchronize(0); const auto t1 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() < 70.0) {} }
                    }
                    if (mode >= 1) { hipStreamSynchronize(0); const auto t1 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() < 30.0) {} }
                    else if (call % 20 == 19) hipStreamSynchronize(0);
                    if (call >= 100 && (mode >= 1 || call % 20 == 19)) {
                        hipEventSynchronize(eb[5]);
                        for (int i = 0; i < 3; ++i) { float a, b; hipEventElapsedTime(&a, ea[2 * i], eb[2 * i]); hipEventElapsedTime(&b, ea[2 * i + 1], eb[2 * i + 1]); k2 += a; k1 += b; }
                        n += 3;
                    }
                }
                const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
                printf("200 'calls' of 3 x (K2, K1), %-62s K1 %.1f us | K2 %.1f us (events, calls 100-199)  wall %.2f ms per call\n",
                       mode == 0 ? "no host synchronisation:" : (mode == 1 ? "host sync + 30 us pause after every call:" : "... and a sync + 70 us pause inside every call:"),
                       k1 / n * 1e3, k2 / n * 1e3, wall / 200);
            }
        }
        for (int b = 0; b < NB; ++b) if (b != best) { CK(hipFree(G[b])); G[b] = nullptr; }
        k12("the other candidates freed:");
        k12("(again)");
        for (int i = 0; i < 600; ++i) { gpca::launch_gq_d(0, G[best], ld8, p1, Npad, Qd, qs, r, bb, s, T, cp, ap, 1, 32, ko); gpca::launch_gtt_d(0, G[best], ld8, Mpad, Npad, Td, Yp, p2, ko); }
        CK(hipDeviceSynchronize());
        k12("after 2 s of K1 / K2 back to back:");
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_linear, dim3(16384), dim3(256), 0, 0, (const i32x4v*)G[best], bytes / 16, out);
        k12("after 40 linear read passes over it:");
        int8_t* w1 = nullptr; CK(hipMalloc(&w1, (size_t)1 << 30));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)w1, ((int64_t)1 << 30) / 4, 1u, 0x01010101u);
        k12("after 10 fills of a fresh 1 GiB buffer:");
        std::vector<int8_t*> refill;
        for (int b = 0; b + 1 < NB; ++b) { int8_t* q = nullptr; if (hipMalloc(&q, bytes) == hipSuccess) refill.push_back(q); }
        k12("as many buffers allocated again (untouched):");
        for (auto q : refill) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)q, bytes / 4, 1u, 0x01010101u);
        CK(hipDeviceSynchronize());
        k12("... and written once:");
        for (auto q : refill) CK(hipFree(q));
        k12("... and freed again:");
    }
    printf("%d buffers of %.1f GB (plain hipMalloc, side by side), %d passes round-robin:\n", NB, bytes * 1e-9, passes);
    for (int b = 0; b < NB; ++b)
        if (t[b][0] > 0) printf("  buffer %d at %p: linear nt probe %7.1f us = %.2f TB/s | K1 k_gq_d %7.1f us | K2 k_gtt_d %7.1f us\n", b, (void*)G[b], t[b][0] / passes,
               bytes / (t[b][0] / passes) * 1e-6, t[b][1] / passes, t[b][2] / passes);
    return 0;
}
