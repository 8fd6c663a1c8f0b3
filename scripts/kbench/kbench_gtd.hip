// Where a launch of the int8-resident K2 (k_gtt_d: Y^T tiles = T'^T G by LDS-DMA) spends its time.  The product source compiled with
// GPCA_STAMP=1 and the kernel instantiated with ablation bits (gemm_i8.hip, ABL): bit0 no byte transpose, bit1 T' planes fetched once,
// bit2 no genotype refills (no HBM stream), bit3 no MFMA, bit4 no Ypart stores.  All variants run round-robin in ONE process (run-to-run
// noise of these kernels is +-2 %).  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -DGPCA_STAMP=1 -o kbench_gtd kbench_gtd.hip && ./kbench_gtd [M N reps W]
#ifndef GPCA_STAMP
#define GPCA_STAMP 1
#endif
#include "../../genomic_pca_amd/csrc/gemm_i8.hip"
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, int64_t n, uint32_t seed, uint32_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = x & mask;
    }
}
template <int ABL>
static void launch(const int8_t* G, int64_t ld8, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Yp, const gpca::Gtt8Plan& plan, int remap) {
    hipLaunchKernelGGL((gpca::k_gtt_d<1, ABL>), dim3((unsigned)plan.grid), dim3(256), sizeof(gpca::GqdSmem), 0, (const uint8_t*)G, ld8, Npad, Td, Yp, plan.S,
                       plan.C, plan.ngroups, plan.W, plan.tasks_per_wg, plan.strided, remap);
}
template <int ABL>
static int opt_in() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gpca::k_gtt_d<1, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(gpca::GqdSmem));
}
// decomposition A/B in one process:  kbench_gtd plans M N reps W [W ...]   W = 0: gtt8_plan_batched's choice; W > 0: that many row chunks,
// tasks spread evenly over <= 256 workgroups (workgroup v takes tasks v, v + grid, ...; 1000 + W: consecutive tasks instead); W < 0: the slice form of rounds 1-3 (one task per workgroup, gtt8_plan's W)
static int plans_main(int argc, char** argv) {
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]);
    const int reps = atoi(argv[4]);
    const int64_t Npad = (N + 255) / 256 * 256, ld8 = ((Npad / 256) % 2 == 0) ? Npad + 256 : Npad, Mpad = (M + 127) / 128 * 128;
    std::vector<gpca::Gtt8Plan> plans;
    std::vector<std::string> names;
    int maxW = 1;
    for (int i = 5; i < argc; ++i) {
        const int W = atoi(argv[i]);
        gpca::Gtt8Plan p = gpca::gtt8_plan_batched(Mpad, Npad, 2048);
        char nm[96];
        if (W < 0) {
            const gpca::Gtt8Plan o = gpca::gtt8_plan(Mpad, Npad, 2048);
            p.C = (p.S + o.W - 1) / o.W; p.W = (int)((p.S + p.C - 1) / p.C); p.tasks_per_wg = 1; p.grid = (int64_t)p.W * p.ngroups;
            snprintf(nm, sizeof nm, "slice form: W %d, %lld workgroups x 1 task", p.W, (long long)p.grid);
        } else {
            const int Wr = W >= 1000 ? W - 1000 : W;
            if (Wr > 0) { p.C = (p.S + Wr - 1) / Wr; p.W = (int)((p.S + p.C - 1) / p.C); const int64_t T = (int64_t)p.W * p.ngroups; p.tasks_per_wg = (int)((T + 255) / 256); p.grid = (T + p.tasks_per_wg - 1) / p.tasks_per_wg; }
            p.strided = W >= 1000 ? 0 : 1;
            snprintf(nm, sizeof nm, "%s%s W %d, %lld workgroups x %d tasks of %lld stages", W ? "batched" : "batched (auto)", p.strided ? " strided" : " consecutive", p.W, (long long)p.grid, p.tasks_per_wg, (long long)(p.S / p.W));
        }
        plans.push_back(p); names.push_back(nm); maxW = std::max(maxW, p.W);
    }
    int8_t* G; int8_t* Td; double* Yp;
    CK(hipMalloc(&G, Mpad * ld8)); CK(hipMalloc(&Td, Mpad * 32 * 4)); CK(hipMalloc(&Yp, (size_t)maxW * Npad * 32 * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G, Mpad * ld8 / 4, 1u, 0x01010101u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)Td, Mpad * 32, 2u, 0x3f3f3f3fu);
    if (opt_in<0>()) { printf("LDS opt-in failed\n"); return 1; }
    std::vector<std::vector<double>> ms(plans.size());
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 10; ++it) launch<0>(G, ld8, Mpad, Npad, Td, Yp, plans[0], 1);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < reps; ++rep)
        for (size_t v = 0; v < plans.size(); ++v) {
            launch<0>(G, ld8, Mpad, Npad, Td, Yp, plans[v], 1);
            (void)hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) launch<0>(G, ld8, Mpad, Npad, Td, Yp, plans[v], 1);
            (void)hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            ms[v].push_back(t / 10);
        }
    printf("k_gtt_d %lld x %lld: decompositions, %d x 10 launches each, round-robin (the fold of the W partial tiles is NOT in these times: W x %lld x 256 B per launch)\n",
           (long long)M, (long long)N, reps, (long long)Npad);
    for (size_t v = 0; v < plans.size(); ++v) {
        std::vector<double> x = ms[v]; std::sort(x.begin(), x.end());
        double m = 0; for (double y : x) m += y; m /= x.size();
        printf("  %-62s mean %.4f ms (min %.4f max %.4f) = %.2f TB/s\n", names[v].c_str(), m, x.front(), x.back(), (double)M * N / (m * 1e-3) / 1e12);
    }
    return 0;
}
// packed K2 (k_gtt_p, 2-bit rows): the product's plan, four and three digit planes round-robin      ./kbench_gtd packed M N reps
static int packed_main(int argc, char** argv) {
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]);
    const int reps = atoi(argv[4]);
    const int64_t Npad = (N + 1023) / 1024 * 1024, ld2 = Npad / 4, Mpad = (M + 127) / 128 * 128;
    const gpca::Gtt8Plan p = gpca::gtt8_plan_batched(Mpad, Npad, 2048);
    uint8_t* G2; int8_t* Td; double* Yp;
    CK(hipMalloc(&G2, Mpad * ld2)); CK(hipMalloc(&Td, Mpad * 32 * 4)); CK(hipMalloc(&Yp, (size_t)p.W * Npad * 32 * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G2, Mpad * ld2 / 4, 1u, 0xaaaaaaaau ^ 0xffffffffu);   // codes 0 / 1 only
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)Td, Mpad * 32, 2u, 0x3f3f3f3fu);
    if (gpca::init_device_kernels_i8()) { printf("LDS opt-in failed\n"); return 1; }
    gpca::KernelOpts ko;
    double sum[2] = {0, 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) if (gpca::launch_gtt_p(0, G2, ld2, Mpad, Npad, Td, Yp, p, 4, ko)) { printf("launch refused\n"); return 1; }
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < reps; ++rep)
        for (int v = 0; v < 2; ++v) {
            const int nd = v ? 3 : 4;
            gpca::launch_gtt_p(0, G2, ld2, Mpad, Npad, Td, Yp, p, nd, ko);
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) gpca::launch_gtt_p(0, G2, ld2, Mpad, Npad, Td, Yp, p, nd, ko);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float t; hipEventElapsedTime(&t, e0, e1); sum[v] += t / 10;
        }
    printf("k_gtt_p %lld x %lld (W %d, %lld workgroups x %d tasks), %d x 10 launches per variant, round-robin: four planes %.4f ms, three planes %.4f ms\n",
           (long long)M, (long long)N, p.W, (long long)p.grid, p.tasks_per_wg, reps, sum[0] / reps, sum[1] / reps);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 4 && std::string(argv[1]) == "packed") return packed_main(argc, argv);
    if (argc > 5 && std::string(argv[1]) == "plans") return plans_main(argc, argv);
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000064, N = argc > 2 ? atoll(argv[2]) : 10000;
    const int reps = argc > 3 ? atoi(argv[3]) : 4;
    const int target = argc > 4 ? atoi(argv[4]) : 2048;
    const int64_t Npad = (N + 255) / 256 * 256, ld8 = ((Npad / 256) % 2 == 0) ? Npad + 256 : Npad, Mpad = (M + 127) / 128 * 128;
    int8_t* G; int8_t* Td; double* Yp;
    const gpca::Gtt8Plan plan = gpca::gtt8_plan_batched(Mpad, Npad, target);
    CK(hipMalloc(&G, Mpad * ld8)); CK(hipMalloc(&Td, Mpad * 32 * 4)); CK(hipMalloc(&Yp, (size_t)(plan.W + 2) * Npad * 32 * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G, Mpad * ld8 / 4, 1u, 0x01010101u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)Td, Mpad * 32, 2u, 0x3f3f3f3fu);
    if (opt_in<0>() || opt_in<1>() || opt_in<2>() || opt_in<4>() || opt_in<8>() || opt_in<9>() || opt_in<16>() || opt_in<12>() || opt_in<7>()) { printf("LDS opt-in failed\n"); return 1; }
    struct V { const char* name; void (*fn)(const int8_t*, int64_t, int64_t, int64_t, const int8_t*, double*, const gpca::Gtt8Plan&, int); int remap; };
    const V vs[] = {{"full kernel", launch<0>, 1}, {"no XCD remap", launch<0>, 0}, {"no byte transpose", launch<1>, 1}, {"T' planes fetched once", launch<2>, 1},
                    {"no genotype refills (no HBM stream)", launch<4>, 1}, {"no MFMA", launch<8>, 1}, {"no MFMA, no transpose", launch<9>, 1},
                    {"no Ypart stores", launch<16>, 1}, {"no refills, no MFMA (LDS reads + decode only)", launch<12>, 1},
                    {"no transpose, planes once, no refills (MFMA + LDS reads)", launch<7>, 1}};
    const int nv = (int)(sizeof vs / sizeof vs[0]);
    std::vector<std::vector<double>> ms(nv);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 10; ++it) vs[0].fn(G, ld8, Mpad, Npad, Td, Yp, plan, 1);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < reps; ++rep)
        for (int v = 0; v < nv; ++v) {
            vs[v].fn(G, ld8, Mpad, Npad, Td, Yp, plan, vs[v].remap);
            (void)hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) vs[v].fn(G, ld8, Mpad, Npad, Td, Yp, plan, vs[v].remap);
            (void)hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            ms[v].push_back(t / 10);
        }
    printf("k_gtt_d %lld x %lld (pitch %lld): %lld workgroups x %d tasks of <= %lld stages (%lld n-groups x %d row chunks of %lld stages); %d x 10 launches per variant, round-robin\n",
           (long long)M, (long long)N, (long long)ld8, (long long)plan.grid, plan.tasks_per_wg, (long long)(plan.rows_per_wave / 128), (long long)plan.ngroups, plan.W, (long long)plan.S, reps);
    for (int v = 0; v < nv; ++v) {
        std::vector<double> x = ms[v]; std::sort(x.begin(), x.end());
        double m = 0; for (double y : x) m += y; m /= x.size();
        printf("  %-58s mean %.4f ms (min %.4f max %.4f) = %.2f TB/s of genotype bytes\n", vs[v].name, m, x.front(), x.back(), (double)M * N / (m * 1e-3) / 1e12);
    }
    // the full kernel once more for its stamps: spans of the workgroups, and when the second batch (grid > 256 CUs) starts
    vs[0].fn(G, ld8, Mpad, Npad, Td, Yp, plan, 1);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(2 * 4096);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(gpca::g_gtd_stamp), st.size() * 8));
    const int nb = (int)std::min<int64_t>(plan.grid, 4096);
    unsigned long long s0 = ~0ull, e1x = 0;
    for (int b = 0; b < nb; ++b) { s0 = std::min(s0, st[2 * b]); e1x = std::max(e1x, st[2 * b + 1]); }
    std::vector<double> dur, start, end;
    for (int b = 0; b < nb; ++b) { dur.push_back((st[2 * b + 1] - st[2 * b]) * 0.01); start.push_back((st[2 * b] - s0) * 0.01); end.push_back((st[2 * b + 1] - s0) * 0.01); }
    std::vector<double> d2 = dur, s2 = start, e2 = end;
    std::sort(d2.begin(), d2.end()); std::sort(s2.begin(), s2.end()); std::sort(e2.begin(), e2.end());
    printf("  workgroup spans: min %.1f median %.1f max %.1f us; launch (first start -> last end) %.1f us\n", d2.front(), d2[nb / 2], d2.back(), (e1x - s0) * 0.01);
    printf("  start times: 256th workgroup %.1f us, 257th %.1f us, last %.1f us; end times: first %.1f, median %.1f, last %.1f us\n", s2[std::min(255, nb - 1)],
           nb > 256 ? s2[256] : -1.0, s2.back(), e2.front(), e2[nb / 2], e2.back());
    return 0;
}
