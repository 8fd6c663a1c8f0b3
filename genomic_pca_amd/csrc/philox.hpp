// Philox4x32-10 counter RNG (Salmon, Moraes, Dror, Shaw -- SC'11), host + gfx950 device.
// Streams used by the engine (4th counter word):
//   0x47454E4F  synthetic genotypes  counter = (snp_lo, snp_hi, sample/2)
//   0x4F4D4547  sketch matrix Omega  counter = (snp_lo, snp_hi, column/4)
//   0x47454E31  fast panel generator counter = (snp_lo, snp_hi, sample/8): eight 16-bit uniforms per call
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GPCA_HD __host__ __device__ __forceinline__
#else
#define GPCA_HD inline
#endif

struct philox_out { uint32_t v[4]; };

GPCA_HD philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox_out o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

#define GPCA_STREAM_GENO 0x47454E4Fu
#define GPCA_STREAM_OMEGA 0x4F4D4547u
#define GPCA_STREAM_GEN16 0x47454E31u   // fast panel generator: counter = (snp_lo, snp_hi, sample/8)
