// Philox4x32-10 counter RNG (Salmon, Moraes, Dror, Shaw -- SC'11), host + gfx950 device.
// Streams used by the engine (4th counter word):
//   0x47454E4F  synthetic genotypes  counter = (snp_lo, snp_hi, sample/2)
//   0x4F4D4547  sketch matrix Omega  counter = (snp_lo, snp_hi, column/4)
// The fast panel generator (GPCA_PANEL_SYNTH16) uses SplitMix64 in counter mode instead (below): output (snp << 26) + sample / 4
// of the stream seeded with `seed` = four 16-bit uniforms.
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
// SplitMix64 (Steele, Lea, Flood 2014; the generator that seeds xoshiro) in counter mode: output number i (0-based) of the stream
// seeded with `seed` is mix(seed + (i + 1) * gamma) -- any i can be computed directly, consecutive i cost one 64-bit add.
// Known answers (seed 1234567): 6457827717110365317, 3203168211198807973, 9817491932198370423, ...
#define GPCA_SPLITMIX_GAMMA 0x9E3779B97F4A7C15ull
GPCA_HD uint64_t splitmix64_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
