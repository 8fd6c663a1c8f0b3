// The transcendentals of the sketch's Box-Muller draw, written for what they are fed: a 32-bit uniform u = (v + 1) 2^-32.
//   z0 = sqrt(-2 ln u0) cos(2 pi u1),  z1 = sqrt(-2 ln u0) sin(2 pi u1)            (oracle/gpca_oracle.c:omega4)
// The general-purpose f64 log / sincospi / sqrt of the device library took 76 of k_omega's 149 us (M = 10^6, l = 30:
// scripts/kbench/kbench_omega.hip).  With a 33-bit integer argument the reductions are integer arithmetic:
//   * ln: w = v + 1 = m 2^e with m in (0.709, 1.418]; a 91-entry table of (1 / c_k, ln c_k), c_k = k / 128, k = round(128 m) = 91 .. 181;
//     ln m = ln c_k + log1p(r), r = m / c_k - 1, |r| <= 0.0055, degree-8 polynomial.  u -> 1 lands on e = 32, c = 1 (ln c = 0 exactly):
//     -2 ln u keeps its relative accuracy down to u = 1 - 2^-32;
//   * sin / cos of 2 pi w 2^-32: the nearest quadrant and the signed remainder are integer operations on w; Taylor polynomials to
//     x^15 / x^16 on [-pi/4, pi/4];
//   * sqrt by v_rsq_f64 and one cubic step.
// Every function is within 2e-16 relative (sin / cos: 2e-16 absolute) of the long-double value -- tests/cpp/omega_math_check.cpp runs
// the same code on the host over every exponent and 20M random arguments -- i.e. as close to the oracle's libm as libm is to itself; the
// sketch is rounded to f32 / 28-bit digits afterwards.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define OMG_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define OMG_HD inline
#endif

namespace gpca {

// (1 / c_k, ln c_k) for c_k = k / 128, k = 91 .. 181; 1 / c_k is the rounded double and ln c_k = -ln(that double), to long-double accuracy
// (csrc/omega_table.inc, printed by tests/cpp/omega_math_check.cpp)
struct OmegaLnEntry { double inv_c, ln_c; };
constexpr int kOmegaLnEntries = 91;

OMG_HD double omg_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}
OMG_HD double omg_rsqrt(double x) {     // x normal, positive
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    const double h = omg_fma(-(x * y), y, 1.0);
    return omg_fma(y * h, omg_fma(0.375, h, 0.5), y);
#else
    return 1.0 / std::sqrt(x);
#endif
}

// -2 ln((v + 1) 2^-32) >= 0
OMG_HD double omg_neg2ln(uint32_t v, const OmegaLnEntry* __restrict__ tab) {
    const uint64_t w = (uint64_t)v + 1u;                       // 1 .. 2^32
    int e = 63 - __builtin_clzll(w);                           // floor(log2 w): 0 .. 32
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_ldexp((double)w, -e);                 // w 2^-e in [1, 2), exact
#else
    double m = std::ldexp((double)w, -e);
#endif
    if (m > 181.5 / 128.0) { m *= 0.5; e += 1; }               // m in (0.709, 1.418]: |ln m| <= 0.35, and u -> 1 ends on e = 32, c = 1 (no cancellation)
    int k = (int)omg_fma(m, 128.0, 0.5);                       // round(128 m): 91 .. 181
    k = k < 91 ? 91 : (k > 181 ? 181 : k);
    const OmegaLnEntry t = tab[k - 91];
    const double r = omg_fma(m, t.inv_c, -1.0);
    double p = -0.125;                                         // log1p(r) = r - r^2/2 + r^3/3 - ... + r^7/7 - r^8/8
    p = omg_fma(p, r, 1.0 / 7.0); p = omg_fma(p, r, -1.0 / 6.0); p = omg_fma(p, r, 0.2); p = omg_fma(p, r, -0.25);
    p = omg_fma(p, r, 1.0 / 3.0); p = omg_fma(p, r, -0.5); p = omg_fma(p, r, 1.0);
    const double lnm = omg_fma(p, r, t.ln_c);
    const double lnu = omg_fma((double)(e - 32), 0.6931471805599453094, lnm);
    return -2.0 * lnu;
}

// cos and sin of 2 pi (v + 1) 2^-32
OMG_HD void omg_sincos2pi(uint32_t v, double& c, double& s) {
    const uint32_t a = v + 1u;                                 // (v + 1) mod 2^32: the angle in units of 2 pi 2^-32
    const uint32_t q = (a + 0x20000000u) >> 30;                // nearest quadrant, 0 .. 4 (4 = a full turn)
    const int32_t d = (int32_t)(a - (q << 30));                // signed remainder in [-2^29, 2^29)
    const double x = (double)d * 1.4629180792671596e-09;       // 2 pi 2^-32
    const double x2 = x * x;
    double ps = -7.647163731819816e-13;                        // -1/15!
    ps = omg_fma(ps, x2, 1.6059043836821613e-10); ps = omg_fma(ps, x2, -2.505210838544172e-08); ps = omg_fma(ps, x2, 2.7557319223985893e-06);
    ps = omg_fma(ps, x2, -1.984126984126984e-04); ps = omg_fma(ps, x2, 8.333333333333333e-03); ps = omg_fma(ps, x2, -1.6666666666666666e-01);
    const double sx = omg_fma(ps * x2, x, x);
    double pc = 4.779477332387385e-14;                         // 1/16!
    pc = omg_fma(pc, x2, -1.1470745597729725e-11); pc = omg_fma(pc, x2, 2.08767569878681e-09); pc = omg_fma(pc, x2, -2.755731922398589e-07);
    pc = omg_fma(pc, x2, 2.48015873015873e-05); pc = omg_fma(pc, x2, -1.388888888888889e-03); pc = omg_fma(pc, x2, 4.1666666666666664e-02);
    pc = omg_fma(pc, x2, -0.5);
    const double cx = omg_fma(pc, x2, 1.0);
    const uint32_t qq = q & 3u;
    c = qq == 0 ? cx : (qq == 1 ? -sx : (qq == 2 ? -cx : sx));
    s = qq == 0 ? sx : (qq == 1 ? cx : (qq == 2 ? -sx : -cx));
}

// the Box-Muller pair of two 32-bit uniforms
OMG_HD void omg_box_muller(uint32_t v0, uint32_t v1, const OmegaLnEntry* __restrict__ tab, double& z0, double& z1) {
    const double x = omg_neg2ln(v0, tab);
    const double r = x > 0.0 ? x * omg_rsqrt(x) : 0.0;
    double c, s;
    omg_sincos2pi(v1, c, s);
    z0 = r * c; z1 = r * s;
}

}  // namespace gpca
