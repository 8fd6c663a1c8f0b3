// What one lone wave pays per instruction on gfx950: dependent / independent f64 FMA, v_rsq_f64, v_readlane -> VALU, LDS read -> use,
// DPP move -> use, v_cndmask.  Cycles from s_memtime around 256 repetitions.  (The device eigen-solver is one wave of serial f64 work.)
//   hipcc --offload-arch=gfx950 -O3 -o probe_lat probe_lat.hip && ./probe_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
__global__ __launch_bounds__(64) void k_probe(double* out, unsigned long long* cyc, double seed) {
    __shared__ double lds[256];
    const int lane = threadIdx.x;
    lds[lane] = seed + lane; lds[lane + 64] = seed * 2 + lane;
    __syncthreads();
    double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9;
    unsigned long long t0, t1;
    int slot = 0;
#define BEGIN t0 = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory");
#define END asm volatile("s_nop 0" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[slot] = t1 - t0; slot++;
    // 0: empty
    BEGIN END
    // 1: dependent fma chain
    BEGIN
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    END
    // 2: four independent chains (REP instructions in all)
    { double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    BEGIN
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
    }
    END
    a += a1 + a2 + a3; }
    // 3: dependent v_mul_f64
    BEGIN
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    END
    // 4: dependent v_rsq_f64
    { double x = 1.5 + lane;
    BEGIN
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
    END
    a += x; }
    // 5: dependent f32 fma
    { float f = (float)a, g = 1.000001f, h = 1e-6f;
    BEGIN
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));
    END
    a += f; }
    // 6: readlane (uniform SGPR index) -> fma with the SGPR pair, dependent through a
    { int idx = lane & 0;   // 0, but not a compile-time constant to the assembler
      int sidx = __builtin_amdgcn_readfirstlane(idx + 5);
    BEGIN
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) {
        double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), sidx), __builtin_amdgcn_readlane(__double2loint(a), sidx));
        a = fma(a, b, s * 1e-30);
    }
    END }
    // 7: LDS read -> dependent use (address from the value)
    { int addr = lane;
    BEGIN
#pragma unroll 16
    for (int i = 0; i < REP / 4; ++i) { const double v = lds[addr & 127]; addr = (int)v & 63; }
    END
    a += addr; }
    // 8: DPP row_shr:1 mov pair + add, dependent
    BEGIN
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, true), lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, true);
        a += __hiloint2double(hi, lo) * 1e-30;
    }
    END
    // 9: v_cmp + 2 cndmask (select by lane), dependent
    BEGIN
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { a = (lane == (i & 63)) ? a * b : a; }
    END
    // 10: LDS write then read of the same address (one wave)
    BEGIN
#pragma unroll 16
    for (int i = 0; i < REP / 4; ++i) { lds[lane] = a; asm volatile("" ::: "memory"); a = lds[lane ^ 1] * b; }
    END
    // 11-15: issue rate of the K1 epilogue's conversions (8 independent destinations, 256 instructions in all)
#define INDEP8(INSTR, OUTC, INC, SRC)                                                                         \
    BEGIN                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < REP / 8; ++i) {                                                     \
        asm volatile(INSTR " %0, %1" : OUTC(r0) : INC(SRC)); asm volatile(INSTR " %0, %1" : OUTC(r1) : INC(SRC));  \
        asm volatile(INSTR " %0, %1" : OUTC(r2) : INC(SRC)); asm volatile(INSTR " %0, %1" : OUTC(r3) : INC(SRC));  \
        asm volatile(INSTR " %0, %1" : OUTC(r4) : INC(SRC)); asm volatile(INSTR " %0, %1" : OUTC(r5) : INC(SRC));  \
        asm volatile(INSTR " %0, %1" : OUTC(r6) : INC(SRC)); asm volatile(INSTR " %0, %1" : OUTC(r7) : INC(SRC));  \
    }                                                                                                         \
    END
    { int isrc = lane * 977 + 13; double r0, r1, r2, r3, r4, r5, r6, r7;
      INDEP8("v_cvt_f64_i32", "=v", "v", isrc)
      a += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; }
    { float r0, r1, r2, r3, r4, r5, r6, r7;
      INDEP8("v_cvt_f32_f64", "=v", "v", a)
      a += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; }
    { double r0 = a, r1 = a, r2 = a, r3 = a, r4 = a, r5 = a, r6 = a, r7 = a;
      BEGIN
#pragma unroll
      for (int i = 0; i < REP / 8; ++i) {
          asm volatile("v_add_f64 %0, %0, %1" : "+v"(r0) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(r1) : "v"(b));
          asm volatile("v_add_f64 %0, %0, %1" : "+v"(r2) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(r3) : "v"(b));
          asm volatile("v_add_f64 %0, %0, %1" : "+v"(r4) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(r5) : "v"(b));
          asm volatile("v_add_f64 %0, %0, %1" : "+v"(r6) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(r7) : "v"(b));
      }
      END
      a += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; }
    { double r0 = a, r1 = a, r2 = a, r3 = a, r4 = a, r5 = a, r6 = a, r7 = a;
      BEGIN
#pragma unroll
      for (int i = 0; i < REP / 8; ++i) {
          asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r0) : "v"(b)); asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r1) : "v"(b));
          asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r2) : "v"(b)); asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r3) : "v"(b));
          asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r4) : "v"(b)); asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r5) : "v"(b));
          asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r6) : "v"(b)); asm volatile("v_fmac_f64 %0, 0x40600000, %1" : "+v"(r7) : "v"(b));
      }
      END
      a += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; }
    { float r0 = (float)a, r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = r0, r7 = r0; float fb = 1.0001f;
      BEGIN
#pragma unroll
      for (int i = 0; i < REP / 8; ++i) {
          asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r0) : "v"(fb)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r1) : "v"(fb));
          asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r2) : "v"(fb)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r3) : "v"(fb));
          asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r4) : "v"(fb)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r5) : "v"(fb));
          asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r6) : "v"(fb)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r7) : "v"(fb));
      }
      END
      a += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; }
    out[lane] = a;
}
int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 64 * 8); hipMalloc(&c, 16 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, c, 1.25);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"empty", "dependent v_fma_f64 x256", "4 independent v_fma_f64 chains, 256 in all", "dependent v_mul_f64 x256", "dependent v_rsq_f64 x256",
                        "dependent v_fma_f32 x256", "(2 readlane + mul + fma) x128", "LDS read -> address of the next x64", "(2 DPP mov + mul + add) x64",
                        "(cmp + mul + 2 cndmask) x64", "(LDS write, read other lane's, mul) x64",
                        "independent v_cvt_f64_i32 x256", "independent v_cvt_f32_f64 x256", "independent v_add_f64 x256 (8 chains)",
                        "independent v_fmac_f64 with a literal x256", "independent v_fmac_f32 x256 (8 chains)"};
    const int reps[] = {1, 256, 256, 256, 256, 256, 128, 64, 64, 64, 64, 256, 256, 256, 256, 256};
    for (int i = 0; i < 16; ++i) printf("%-48s %6llu cycles  = %.1f per repetition\n", nm[i], h[i], (double)(h[i] - h[0]) / reps[i]);
    return 0;
}
