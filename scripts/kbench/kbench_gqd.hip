// Where a launch of the int8-resident K1 (k_gq_d) spends its time, round by round: the product source compiled with GPCA_STAMP=1
// (s_memrealtime stamps of wave 0 of every workgroup after each round's stage loop and after its epilogue).  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -DGPCA_STAMP=1 -o kbench_gqd kbench_gqd.hip && ./kbench_gqd [M N chain pitch brief phase]
#ifndef GPCA_STAMP
#define GPCA_STAMP 1
#endif
#include "../../genomic_pca_amd/csrc/gemm_i8.hip"
#include "gqs_skew.inc"
#include "gqt_drip.inc"
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>
#include <cstring>
#define CK_V(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); } } while (0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = x & 0x01010101u;          // dosages 0 / 1
    }
}
// interleaved comparison of phase settings in ONE process (run-to-run noise of this kernel is +-2 %: placement, clocks):
//   kbench_gqd ab M N reps A:B [A:B ...]     every setting runs `reps` times 10 launches, round-robin
static int ab_main(int argc, char** argv) {
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]);
    const int reps = atoi(argv[4]);
    const int64_t Npad = (N + 255) / 256 * 256, ld8 = ((Npad / 256) % 2 == 0) ? Npad + 256 : Npad;
    const int waves = 1024;
    int8_t* G; int8_t* Qd; double* qs; float *r, *b, *s, *T, *cp; double* ap;
    CK(hipMalloc(&G, M * ld8)); CK(hipMalloc(&Qd, Npad * 32 * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G, M * ld8 / 4, 1u);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t*)Qd, Npad * 32, 2u);
    CK(hipMalloc(&qs, 32 * 8));
    CK(hipMalloc(&r, M * 4)); CK(hipMalloc(&b, M * 4)); CK(hipMalloc(&s, 32 * 4)); CK(hipMalloc(&T, M * 32 * 4));
    {   // operands that make every output element distinct: the settings' outputs are compared bit by bit at the end
        std::vector<double> hq(32); std::vector<float> hs(32), hr(M), hb(M);
        for (int i = 0; i < 32; ++i) { hq[i] = 1e-3 * (1 + i); hs[i] = 0.25f + 0.01f * i; }
        for (int64_t i = 0; i < M; ++i) { hr[i] = 1.f + (float)(i % 977) * 1e-3f; hb[i] = -0.5f + (float)(i % 331) * 3e-3f; }
        CK(hipMemcpy(qs, hq.data(), 32 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), 32 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(r, hr.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), M * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&cp, (M / 32) * 32 * 4)); CK(hipMalloc(&ap, waves * 32 * 8));
    if (gpca::init_device_kernels_i8() != 0) { printf("LDS opt-in failed\n"); return 1; }
    gpca::GqPlan plan{M / 32, waves};
    std::vector<gpca::KernelOpts> kos;
    std::vector<int> skews;
    auto go = [&](size_t c_) -> int {
        if (skews[c_] == 100) return gpca::launch_gq_t(0, G, ld8, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1, 32, kos[c_].gq_phase);
        if (skews[c_]) return gpca::launch_gq_s(0, G, ld8, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1, 32, kos[c_].gq_phase, skews[c_] - 1);
        return gpca::launch_gq_d(0, G, ld8, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1, 32, kos[c_]);
    };
    std::vector<std::string> names;
    for (int i = 5; i < argc; ++i) {
        int a = 0, bb = 0, ch = 1, sk = 0;
        sscanf(argv[i], "%d:%d:%d:%d", &a, &bb, &ch, &sk);
        gpca::KernelOpts ko; ko.gq_chain = ch; ko.gq_phase = a | (bb << 16);
        kos.push_back(ko); skews.push_back(sk); names.push_back(argv[i]);
    }
    std::vector<std::vector<double>> ms(kos.size());
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 10; ++it) if (go(0) != 0) { printf("launch refused\n"); return 1; }
    for (int rep = 0; rep < reps; ++rep)
        for (size_t c = 0; c < kos.size(); ++c) {
            go(c);
            (void)hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) go(c);
            (void)hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            ms[c].push_back(t / 10);
        }
    {   // every setting's T, c partials and column maxima against the first setting's, bit by bit
        const size_t nt = (size_t)M * 32, nc = (size_t)(M / 32) * 32, na = (size_t)waves * 32;
        std::vector<float> T0(nt), T1(nt), c0(nc), c1(nc); std::vector<double> a0(32, 0.0), a1(32), ah(na);
        auto colmax = [&](std::vector<double>& out) { CK_V(hipMemcpy(ah.data(), ap, na * 8, hipMemcpyDeviceToHost)); out.assign(32, 0.0); for (size_t i = 0; i < na; ++i) out[i % 32] = std::max(out[i % 32], ah[i]); };
        for (size_t c = 0; c < kos.size(); ++c) {
            CK(hipMemset(T, 0xff, nt * 4)); CK(hipMemset(cp, 0xff, nc * 4)); CK(hipMemset(ap, 0, na * 8));
            go(c);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(c ? T1.data() : T0.data(), T, nt * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c ? c1.data() : c0.data(), cp, nc * 4, hipMemcpyDeviceToHost));
            colmax(c ? a1 : a0);
            if (c) printf("  %-10s vs %s: T %s, c partials %s, column maxima %s\n", names[c].c_str(), names[0].c_str(), memcmp(T0.data(), T1.data(), nt * 4) ? "DIFFER" : "bit-identical",
                          memcmp(c0.data(), c1.data(), nc * 4) ? "DIFFER" : "bit-identical", memcmp(a0.data(), a1.data(), 32 * 8) ? "DIFFER" : "bit-identical");
        }
        double sum = 0; for (size_t i = 0; i < nt; i += 9973) sum += T0[i];
        printf("  (sample sum of T: %.6g)\n", sum);
    }
    printf("k_gq_d %lld x %lld, %d x 10 launches per setting, round-robin (A:B[:chain[:skew]] -> workgroup b starts at stage ((b %% 8) A + (b / 8) B) mod stages; skew 1 = k_gq_s)\n", (long long)M, (long long)N, reps);
    for (size_t c = 0; c < kos.size(); ++c) {
        std::vector<double> v = ms[c]; std::sort(v.begin(), v.end());
        double m = 0; for (double x : v) m += x; m /= v.size();
        printf("  %-10s mean %.4f ms  median %.4f  min %.4f  max %.4f   = %.2f TB/s (mean)\n", names[c].c_str(), m, v[v.size() / 2], v.front(), v.back(), (double)M * N / (m * 1e-3) / 1e12);
    }
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 5 && std::string(argv[1]) == "ab") return ab_main(argc, argv);
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000064, N = argc > 2 ? atoll(argv[2]) : 10000;
    const int chain = argc > 3 ? atoi(argv[3]) : 1;
    const int64_t Npad = (N + 255) / 256 * 256;
    const int64_t ld8 = (argc > 4 && atoll(argv[4]) > 0) ? atoll(argv[4]) : (((Npad / 256) % 2 == 0) ? Npad + 256 : Npad);       // the product's pitch: an odd multiple of 256 B
    const int waves = 1024;
    int8_t* G; int8_t* Qd; double* qs; float *r, *b, *s, *T, *cp; double* ap;
    CK(hipMalloc(&G, M * ld8)); CK(hipMalloc(&Qd, Npad * 32 * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G, M * ld8 / 4, 1u);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t*)Qd, Npad * 32, 2u);
    CK(hipMalloc(&qs, 32 * 8)); CK(hipMemset(qs, 0, 32 * 8));
    CK(hipMalloc(&r, M * 4)); CK(hipMalloc(&b, M * 4)); CK(hipMalloc(&s, 32 * 4)); CK(hipMalloc(&T, M * 32 * 4));
    CK(hipMemset(r, 0, M * 4)); CK(hipMemset(b, 0, M * 4)); CK(hipMemset(s, 0, 32 * 4));
    CK(hipMalloc(&cp, (M / 32) * 32 * 4)); CK(hipMalloc(&ap, waves * 32 * 8));
    if (gpca::init_device_kernels_i8() != 0) { printf("LDS opt-in failed\n"); return 1; }
    gpca::GqPlan plan{M / 32, waves};
    gpca::KernelOpts ko; ko.gq_chain = chain; ko.gq_phase = argc > 6 ? (atoi(argv[6]) | (argc > 7 ? atoi(argv[7]) << 16 : 0)) : 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) if (gpca::launch_gq_d(0, G, ld8, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1, 32, ko) != 0) { printf("launch refused\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) gpca::launch_gq_d(0, G, ld8, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1, 32, ko);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("k_gq_d %lld x %lld (pitch %lld) chain %d phase %d: %.4f ms per launch = %.2f TB/s of genotype bytes (M x N)\n", (long long)M, (long long)N, (long long)ld8, chain, (ko.gq_phase & 0xffff) * 1000 + (ko.gq_phase >> 16), ms,
           (double)M * N / (ms * 1e-3) / 1e12);
    std::vector<unsigned long long> st(1024 * 64);
    int nr = 0;
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(gpca::g_gqd_stamp), st.size() * 8));
    CK(hipMemcpyFromSymbol(&nr, HIP_SYMBOL(gpca::g_gqd_stamp_n), 4));
    const int nb = waves / 4;
    unsigned long long s0 = ~0ull, s1 = 0, f0 = ~0ull, f1 = 0;
    for (int w = 0; w < nb; ++w) {
        s0 = std::min(s0, st[w * 64]); s1 = std::max(s1, st[w * 64]);
        f0 = std::min(f0, st[w * 64 + 2 * nr]); f1 = std::max(f1, st[w * 64 + 2 * nr]);
    }
    printf("  %d rounds per workgroup; first start -> last start %.1f us, first end -> last end %.1f us, first start -> last end %.1f us\n", nr,
           (s1 - s0) * 0.01, (f1 - f0) * 0.01, (f1 - s0) * 0.01);
    {   // by XCD (workgroup b runs on XCD b % 8): mean span of a workgroup's whole launch
        double sum[8] = {0}, mx[8] = {0}, mn[8]; int cnt[8] = {0};
        for (int x = 0; x < 8; ++x) mn[x] = 1e30;
        for (int w = 0; w < nb; ++w) {
            const double d = (st[w * 64 + 2 * nr] - st[w * 64]) * 0.01;
            sum[w & 7] += d; cnt[w & 7]++; mx[w & 7] = std::max(mx[w & 7], d); mn[w & 7] = std::min(mn[w & 7], d);
        }
        printf("  workgroup span by XCD (mean / min / max us):");
        for (int x = 0; x < 8; ++x) printf("  %d: %.0f / %.0f / %.0f", x, sum[x] / cnt[x], mn[x], mx[x]);
        printf("\n");
    }
#if GPCA_STAMP == 2
    {   // stage by stage: median over the workgroups of the time between consecutive stage starts, rounds 0-2
        std::vector<unsigned long long> ss(256 * 3 * 128);
        CK(hipMemcpyFromSymbol(ss.data(), HIP_SYMBOL(gpca::g_gqd_stage_stamp), ss.size() * 8));
        const int ns = (int)std::min<int64_t>(Npad / 128, 128);
        for (int rd = 0; rd < 3 && rd < nr; ++rd) {
            printf("  round %d, median stage duration (us) by stage:", rd);
            for (int st = 0; st + 1 < ns; ++st) {
                std::vector<double> d;
                for (int w = 0; w < nb; ++w) d.push_back((ss[(w * 3 + rd) * 128 + st + 1] - ss[(w * 3 + rd) * 128 + st]) * 0.01);
                std::sort(d.begin(), d.end());
                if (st < 12 || st % 8 == 0 || st + 2 >= ns) printf(" %d:%.2f", st, d[nb / 2]);
            }
            printf("\n");
        }
    }
#endif
    const int brief = argc > 5 ? atoi(argv[5]) : 0;
    for (int rd = 0; rd < nr && rd < 31; ++rd) {
        if (brief && rd != 1 && rd != nr - 1) continue;
        std::vector<double> loop, epi;
        for (int w = 0; w < nb; ++w) {
            const unsigned long long a = st[w * 64 + 2 * rd], m = st[w * 64 + 2 * rd + 1], z = st[w * 64 + 2 * rd + 2];
            loop.push_back((m - a) * 0.01); epi.push_back((z - m) * 0.01);
        }
        std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end());
        printf("  round %2d: stage loop min %.1f median %.1f max %.1f us | epilogue min %.2f median %.2f max %.2f us\n", rd, loop.front(), loop[nb / 2], loop.back(),
               epi.front(), epi[nb / 2], epi.back());
    }
    return 0;
}
