// Ablation harness for the packed K1 kernel (k_gq_2bit): the product source is compiled with GPCA_ABLATE = bitmask
// (bit0 no decode, bit1 no Q loads, bit2 no G loads, bit3 no MFMA) and timed on a 1M x 10k problem.  Not product code.
//   for a in 0 1 2 4 6 7 8; do hipcc --offload-arch=gfx950 -O3 -DGPCA_ABLATE=$a -o kb_$a kbench_gq2.hip; ./kb_$a; done
#include "../../genomic_pca_amd/csrc/gemm_i8.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_fill(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = x;
    }
}
// "real" operands (argv[2] = real): what the product kernel sees -- 2-bit codes in Hardy-Weinberg-like proportions (55 % / 35 % /
// 10 % of 0 / 1 / 2, never 3) and the digit planes of an orthonormal basis (planes 0-2 uniform in [-64, 63], the top plane a small
// signed number): far fewer toggling operand bits than uniform random bytes, which is what the chip's clock responds to.
__global__ void k_fill_codes(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t w = 0;
        for (int f = 0; f < 16; ++f) {
            uint32_t x = ((uint32_t)i * 16u + f) * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
            const uint32_t u = x & 255u;
            w |= (u < 141u ? 0u : (u < 230u ? 1u : 2u)) << (2 * f);
        }
        p[i] = w;
    }
}
__global__ void k_fill_planes(int8_t* q, int64_t nsteps, uint32_t seed) {   // [step][4 planes][64 lanes][16 B]
    const int64_t n = nsteps * 4 * 1024;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        const int plane = (int)((i >> 10) & 3);
        int v;
        if (plane < 3) v = (int)(x & 127u) - 64;
        else v = (int)__popc(x & 0xffffu) + (int)__popc((x >> 16) & 0xffu) - 12;     // ~N(0, 2.4^2): the top digit of a ~4.5 sigma column maximum
        q[i] = (int8_t)v;
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int64_t M = 1000064, N = 10000, Npad = 10240, ld2 = Npad / 4;
    const int waves = argc > 1 ? atoi(argv[1]) : 1024;
    uint8_t* G2; int8_t* Qd; double* qs; float *r, *b, *s, *T, *cp; double* ap;
    const bool real = argc > 2 && argv[2][0] == 'r';
    CK(hipMalloc(&G2, M * ld2)); CK(hipMalloc(&Qd, Npad * 32 * 4));
    if (real) {
        hipLaunchKernelGGL(k_fill_codes, dim3(4096), dim3(256), 0, 0, (uint32_t*)G2, M * ld2 / 4, 1u);
        hipLaunchKernelGGL(k_fill_planes, dim3(1024), dim3(256), 0, 0, Qd, Npad / 32, 2u);
    } else {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)G2, M * ld2 / 4, 1u);
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t*)Qd, Npad * 32, 2u);
    }
    CK(hipMalloc(&qs, 32 * 8)); CK(hipMemset(qs, 0, 32 * 8));
    CK(hipMalloc(&r, M * 4)); CK(hipMalloc(&b, M * 4)); CK(hipMalloc(&s, 32 * 4)); CK(hipMalloc(&T, M * 32 * 4));
    CK(hipMemset(r, 0, M * 4)); CK(hipMemset(b, 0, M * 4)); CK(hipMemset(s, 0, 32 * 4));
    CK(hipMalloc(&cp, (M / 32) * 32 * 4));   /* one c partial per 32-row unit (not per wave: the kernels write cunit[unit][32]) */ CK(hipMalloc(&ap, waves * 32 * 8));
    gpca::GqPlan plan{M / 32, waves};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 3 && argv[3][0] == 'o') {
        // occupancy A/B (round 4), round-robin in one process: the product form (four tiles per sweep, one wave per SIMD, 1 024 waves)
        // against two tiles per sweep with two waves per SIMD (2 048 waves), for both plane counts
        struct V { const char* name; int nd, rmax, w; };
        const V vs[] = {{"4 planes, 4 tiles/sweep, 1 wave/SIMD", 4, 4, 1024}, {"4 planes, 2 tiles/sweep, 2 waves/SIMD", 4, 2, 2048},
                        {"3 planes, 4 tiles/sweep, 1 wave/SIMD", 3, 4, 1024}, {"3 planes, 2 tiles/sweep, 2 waves/SIMD", 3, 2, 2048}};
        double sum[4] = {0, 0, 0, 0};
        auto go = [&](const V& v) {
            const dim3 grid((unsigned)(v.w / 4)), blk(256);
            const int64_t nsuper = Npad / 512;
            if (v.nd == 4 && v.rmax == 4) hipLaunchKernelGGL((gpca::k_gq_2bit<4, 4>), grid, blk, 0, 0, G2, ld2, M / 32, nsuper, Qd, qs, r, b, s, T, cp, ap, 1, (int64_t)32);
            else if (v.nd == 4) hipLaunchKernelGGL((gpca::k_gq_2bit<4, 2>), grid, blk, 0, 0, G2, ld2, M / 32, nsuper, Qd, qs, r, b, s, T, cp, ap, 1, (int64_t)32);
            else if (v.rmax == 4) hipLaunchKernelGGL((gpca::k_gq_2bit<3, 4>), grid, blk, 0, 0, G2, ld2, M / 32, nsuper, Qd, qs, r, b, s, T, cp, ap, 1, (int64_t)32);
            else hipLaunchKernelGGL((gpca::k_gq_2bit<3, 2>), grid, blk, 0, 0, G2, ld2, M / 32, nsuper, Qd, qs, r, b, s, T, cp, ap, 1, (int64_t)32);
        };
        double* ap2; CK(hipMalloc(&ap2, 2048 * 32 * 8)); ap = ap2;
        for (int it = 0; it < 10; ++it) go(vs[0]);
        CK(hipDeviceSynchronize());
        const int reps = 4;
        for (int rep = 0; rep < reps; ++rep)
            for (int v = 0; v < 4; ++v) {
                go(vs[v]);
                hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) go(vs[v]);
                hipEventRecord(e1); CK(hipEventSynchronize(e1));
                float t; hipEventElapsedTime(&t, e0, e1); sum[v] += t / 10;
            }
        printf("k_gq_2bit %lld x %lld, %s operands, %d x 10 launches per variant, round-robin\n", (long long)M, (long long)N, real ? "real-like" : "random", reps);
        for (int v = 0; v < 4; ++v) printf("  %-42s %.4f ms\n", vs[v].name, sum[v] / reps);
        return 0;
    }
    for (int it = 0; it < 2; ++it) gpca::launch_gq_2bit(0, G2, ld2, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) gpca::launch_gq_2bit(0, G2, ld2, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double mf = (double)M * Npad * 32 * 4 / 32768.0;   // MFMAs
    printf("ablate %d %s waves %d: %.3f ms  (%.1f %% of the i8 MFMA floor %.3f ms)\n", GPCA_ABLATE, real ? "real-like operands" : "random bytes", waves, ms,
           100.0 * (mf * 32 / (1024 * 2.39e9) * 1e3) / ms, mf * 32 / (1024 * 2.39e9) * 1e3);
#if GPCA_ABLATE & 16
    {   // steady state: 300 more launches, then read the stamps of the last one
        for (int it = 0; it < 300; ++it) gpca::launch_gq_2bit(0, G2, ld2, plan, Npad, Qd, qs, r, b, s, T, cp, ap, 1);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(2 * 4096);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(gpca::g_kbench_stamp), st.size() * 8));
        std::vector<double> ghz;
        for (int b2 = 0; b2 < waves / 4; ++b2) ghz.push_back((double)st[2 * b2] / (double)st[2 * b2 + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        printf("  in-kernel clock: median %.3f GHz (min %.3f max %.3f), median wave cycles %.0f\n", ghz[ghz.size() / 2], ghz.front(), ghz.back(), (double)st[0]);
        // where the launch's time goes: every workgroup's own span (100 MHz ticks) against first start -> last end
        std::vector<unsigned long long> ab(2 * 4096);
        CK(hipMemcpyFromSymbol(ab.data(), HIP_SYMBOL(gpca::g_kbench_abs), ab.size() * 8));
        const int nb = waves / 4;
        unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1x = 0;
        std::vector<double> dur;
        for (int b2 = 0; b2 < nb; ++b2) {
            s0 = std::min(s0, ab[2 * b2]); s1 = std::max(s1, ab[2 * b2]); e0 = std::min(e0, ab[2 * b2 + 1]); e1x = std::max(e1x, ab[2 * b2 + 1]);
            dur.push_back((double)(ab[2 * b2 + 1] - ab[2 * b2]) * 0.01);
        }
        std::sort(dur.begin(), dur.end());
        printf("  workgroup spans: min %.1f median %.1f max %.1f us; first start -> last start %.1f us, first end -> last end %.1f us, first start -> last end %.1f us\n",
               dur.front(), dur[dur.size() / 2], dur.back(), (double)(s1 - s0) * 0.01, (double)(e1x - e0) * 0.01, (double)(e1x - s0) * 0.01);
    }
#endif
    return 0;
}
