// Host check of genomic_pca_amd/csrc/omega_math.h (the same source the device compiles): -2 ln u, sin / cos of 2 pi u and the Box-Muller
// pair against long-double libm, over every binade edge, the table's interval edges and N random arguments.  Prints the worst errors;
// exit code 1 when a bar is missed.  Also emits the ln table (argument "table") that csrc/omega_table.inc holds.
//   g++ -O2 -std=c++17 -I genomic_pca_amd/csrc tests/cpp/omega_math_check.cpp -o omega_math_check && ./omega_math_check 20000000
#include "omega_math.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
namespace gpca {
static OmegaLnEntry make_entry(int k) {
    OmegaLnEntry t;
    t.inv_c = 128.0 / (double)k;                               // rounded
    t.ln_c = (double)(-logl((long double)t.inv_c));
    return t;
}
}  // namespace gpca
int main(int argc, char** argv) {
    using namespace gpca;
    std::vector<OmegaLnEntry> tab(kOmegaLnEntries);
    for (int k = 91; k <= 181; ++k) tab[k - 91] = make_entry(k);
    if (argc > 1 && !strcmp(argv[1], "table")) {
        for (int k = 0; k < kOmegaLnEntries; ++k) printf("    {%a, %a},%s", tab[k].inv_c, tab[k].ln_c, (k % 2) ? "\n" : "");
        printf("\n");
        return 0;
    }
    const long N = argc > 1 ? atol(argv[1]) : 2000000;
    const long double twopi = 6.283185307179586476925286766559L;
    double worst_ln = 0, worst_sc = 0, worst_z = 0;
    uint32_t at_ln = 0, at_sc = 0;
    auto check = [&](uint32_t v) {
        const long double u = ((long double)v + 1.0L) / 4294967296.0L;
        const long double ref = -2.0L * logl(u);
        const double got = omg_neg2ln(v, tab.data());
        const double err = ref > 0 ? (double)(fabsl(got - ref) / ref) : std::fabs(got);
        if (err > worst_ln) { worst_ln = err; at_ln = v; }
        double c, s;
        omg_sincos2pi(v, c, s);
        const double e2 = (double)fmaxl(fabsl(c - cosl(twopi * u)), fabsl(s - sinl(twopi * u)));
        if (e2 > worst_sc) { worst_sc = e2; at_sc = v; }
        double z0, z1;
        omg_box_muller(v, v * 2654435761u + 12345u, tab.data(), z0, z1);
        const long double u1 = ((long double)(uint32_t)(v * 2654435761u + 12345u) + 1.0L) / 4294967296.0L;
        const long double r = sqrtl(ref);
        const double e3 = (double)fmaxl(fabsl(z0 - r * cosl(twopi * u1)), fabsl(z1 - r * sinl(twopi * u1)));
        if (e3 > worst_z) worst_z = e3;
    };
    // edges: every power of two and its neighbours, the table's interval edges in the top binade, the extremes
    for (int e = 0; e <= 32; ++e)
        for (long d = -3; d <= 3; ++d) { const long long w = (1LL << e) + d; if (w >= 1 && w <= (1LL << 32)) check((uint32_t)(w - 1)); }
    for (int k = 128; k <= 256; ++k)           // interval edges (2k - 1) / 256 of the top binade and of the one below
        for (long d = -2; d <= 2; ++d) for (int sh = 22; sh <= 23; ++sh) { const long long w = (long long)((2 * k - 1) * (1LL << sh)) + d; if (w >= 1 && w <= (1LL << 32)) check((uint32_t)(w - 1)); }
    for (long d = 0; d < 4096; ++d) { check((uint32_t)d); check(0xffffffffu - (uint32_t)d); check(0x7fffffffu + (uint32_t)d - 2048u); check(0x3fffffffu + (uint32_t)d - 2048u); }
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (long i = 0; i < N; ++i) { st = st * 6364136223846793005ull + 1442695040888963407ull; check((uint32_t)(st >> 32)); }
    printf("omega_math: %ld random + edges | -2 ln u: worst relative error %.3e (v = %u) | sin / cos(2 pi u): worst absolute error %.3e (v = %u) | "
           "Box-Muller z: worst absolute error %.3e\n", N, worst_ln, at_ln, worst_sc, at_sc, worst_z);
    return (worst_ln < 5e-16 && worst_sc < 3e-16 && worst_z < 2.5e-15) ? 0 : 1;      // (|z| <= 6.67: 2.5e-15 absolute is 4e-16 of the largest draw)
}
