// HBM read-bandwidth ceiling probe for gfx950: streaming-read kernels over a 10 GiB buffer with varying loads in flight,
// grid sizes and cache policies.  Calibrates what "speed of light" means for the HBM-bound genotype passes (K1/K2).
//   hipcc --offload-arch=gfx950 -O3 -o kbench_stream kbench_stream.hip && ./kbench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <initializer_list>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U, int NT>
__global__ __launch_bounds__(256) void k_read(const i32x4* __restrict__ p, int64_t n16, int* __restrict__ out) {
    // grid-stride over 16-byte elements; U independent loads in flight per lane
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    i32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        i32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
// contiguous-per-block variant: each block owns one contiguous slab (like a wave owning SNP rows)
template <int U>
__global__ __launch_bounds__(256) void k_read_slab(const i32x4* __restrict__ p, int64_t n16, int* __restrict__ out) {
    const int64_t per = n16 / gridDim.x;
    const i32x4* q = p + per * blockIdx.x;
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        i32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = q[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
// K1's access pattern: a wave owns 32*R rows of a row-major [M][ld] byte matrix and sweeps the columns 128 B at a time.
// PAT 0: lane (c = lane & 31, h = lane >> 5) loads 16 B at row c, byte 32 j + 16 h  (the MFMA A-operand mapping; one wave
//        instruction touches 32 rows x 32 B), four j back to back = one 128-B line per row.
// PAT 1: line-coalesced: lane l loads 16 B at row (l >> 3) + 8 j, byte 16 (l & 7)  (one instruction = 8 full lines).
template <int R, int PAT>
__global__ __launch_bounds__(256, 1) void k_read_rows(const char* __restrict__ p, int64_t M, int64_t ld, int* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t units = M / (32 * R);
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t u = (units * wave) / waves; u < (units * (wave + 1)) / waves; ++u) {
        const char* base = p + u * 32 * R * ld;
        for (int64_t col = 0; col < ld; col += 128) {
            i32x4 v[R][4];
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t row = PAT == 0 ? 32 * t + (lane & 31) : 32 * t + 8 * j + (lane >> 3);
                    const int64_t off = PAT == 0 ? 32 * j + 16 * (lane >> 5) : 16 * (lane & 7);
                    v[t][j] = *reinterpret_cast<const i32x4*>(base + row * ld + col + off);
                }
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc ^= v[t][j];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
// operand-map pattern with J back-to-back 32-byte pieces per row per sweep step (J = 4: one 128-B line, 8: two, 16: four)
template <int R, int J>
__global__ __launch_bounds__(256, 1) void k_read_rows_j(const char* __restrict__ p, int64_t M, int64_t ld, int* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t units = M / (32 * R);
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t u = (units * wave) / waves; u < (units * (wave + 1)) / waves; ++u) {
        const char* base = p + u * 32 * R * ld;
        for (int64_t col = 0; col < ld; col += 32 * J) {
            i32x4 v[R][J];
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < J; ++j)
                    v[t][j] = *reinterpret_cast<const i32x4*>(base + (32 * t + (lane & 31)) * ld + col + 32 * j + 16 * (lane >> 5));
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < J; ++j) acc ^= v[t][j];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
// K2's access pattern: a wave owns a 128-byte column slab of a row range; lane (c = lane & 31, h = lane >> 5) loads the
// dword of row 16 h + i, bytes 4 c .. 4 c + 3 (one wave instruction = 2 rows x 128 B), 16 i per 32-row block, NB blocks in
// flight.  Four waves of a workgroup take adjacent slabs.  NT: non-temporal.
template <int NB, int NT>
__global__ __launch_bounds__(256, 1) void k_read_cols(const char* __restrict__ p, int64_t M, int64_t ld, int64_t rows_per_wg, int* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ngroups = ld / 512;
    const int64_t ngroup = blockIdx.x % ngroups, wchunk = blockIdx.x / ngroups;
    const int64_t m0 = wchunk * rows_per_wg, m1 = (m0 + rows_per_wg < M) ? m0 + rows_per_wg : M;
    const char* base = p + (ngroup * 4 + wv) * 128 + 4 * (lane & 31) + (int64_t)(16 * (lane >> 5)) * ld;
    unsigned acc = 0;
    for (int64_t m = m0; m + 32 * NB <= m1; m += 32 * NB) {
        unsigned v[NB][16];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned* q = reinterpret_cast<const unsigned*>(base + (m + 32 * b + i) * ld);
                v[b][i] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc ^= v[b][i];
    }
    if (acc == 0x12345678u) out[0] = 1;
}
// same slab ownership, but the bytes arrive as 8-row x 128-byte b128 pieces (what an LDS-DMA version would request)
template <int NB, int NT>
__global__ __launch_bounds__(256, 1) void k_read_cols_wide(const char* __restrict__ p, int64_t M, int64_t ld, int64_t rows_per_wg, int* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ngroups = ld / 512;
    const int64_t ngroup = blockIdx.x % ngroups, wchunk = blockIdx.x / ngroups;
    const int64_t m0 = wchunk * rows_per_wg, m1 = (m0 + rows_per_wg < M) ? m0 + rows_per_wg : M;
    const char* base = p + (ngroup * 4 + wv) * 128 + 16 * (lane & 7) + (int64_t)(lane >> 3) * ld;
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t m = m0; m + 32 * NB <= m1; m += 32 * NB) {
        i32x4 v[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x4* q = reinterpret_cast<const i32x4*>(base + (m + 32 * b + 8 * i) * ld);
                v[b][i] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc ^= v[b][i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
// K1's DMA fill shape generalised: one wave instruction (64 lanes x 16 B = 1 KiB) covers ROWS = 1024 / WB rows x WB contiguous
// bytes (WB = 128: 8 rows x one line, what k_gq_d issues; 256: 4 rows x two lines; 512: 2 rows x four lines).  A wave owns
// 32*R rows and sweeps the columns WB bytes at a time; NT: non-temporal.
template <int R, int WB, int NT>
__global__ __launch_bounds__(256, 1) void k_read_pieces(const char* __restrict__ p, int64_t M, int64_t ld, int* __restrict__ out) {
    constexpr int ROWS = 1024 / WB, LPR = WB / 16;     // rows per instruction, lanes per row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t units = M / (32 * R);
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t u = (units * wave) / waves; u < (units * (wave + 1)) / waves; ++u) {
        const char* base = p + u * 32 * R * ld;
        for (int64_t col = 0; col < ld; col += WB) {
            i32x4 v[R][32 / ROWS];
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < 32 / ROWS; ++j) {
                    const i32x4* q = reinterpret_cast<const i32x4*>(base + (int64_t)(32 * t + ROWS * j + lane / LPR) * ld + col + 16 * (lane % LPR));
                    v[t][j] = NT ? __builtin_nontemporal_load(q) : *q;
                }
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < 32 / ROWS; ++j) acc ^= v[t][j];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int64_t bytes = 12LL << 30; const int64_t n16 = bytes / 16;
    i32x4* p; int* out; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(p, 1, bytes)); CK(hipMemset(out, 0, 4));
    const int grids[] = {1024, 16384};
    for (int g : grids) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((k_read<1, 0>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 5);
        float t4 = time_ms([&] { hipLaunchKernelGGL((k_read<4, 0>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 5);
        float t8 = time_ms([&] { hipLaunchKernelGGL((k_read<8, 0>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 5);
        float n4 = time_ms([&] { hipLaunchKernelGGL((k_read<4, 1>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 5);
        float s4 = time_ms([&] { hipLaunchKernelGGL((k_read_slab<4>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 5);
        float s8 = time_ms([&] { hipLaunchKernelGGL((k_read_slab<8>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 5);
        printf("grid %6d  stride U1 %.2f  U4 %.2f  U8 %.2f  U4nt %.2f | slab U4 %.2f  U8 %.2f  TB/s\n", g, bytes / t1 * 1e-9,
               bytes / t4 * 1e-9, bytes / t8 * 1e-9, bytes / n4 * 1e-9, bytes / s4 * 1e-9, bytes / s8 * 1e-9);
        fflush(stdout);
    }
    {
        const int64_t M = 1000064, ld = 10240;   // 10.24 GB, the C2 genotype matrix
        const int wgs[] = {256};
        for (int g : wgs) {
            float a4 = time_ms([&] { hipLaunchKernelGGL((k_read_rows<4, 0>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float a2 = time_ms([&] { hipLaunchKernelGGL((k_read_rows<2, 0>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float b4 = time_ms([&] { hipLaunchKernelGGL((k_read_rows<4, 1>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float b2 = time_ms([&] { hipLaunchKernelGGL((k_read_rows<2, 1>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            const double gb = (double)M * ld;
            printf("rows grid %5d  operand-map R4 %.2f R2 %.2f | line-coalesced R4 %.2f R2 %.2f  TB/s\n", g, gb / a4 * 1e-9, gb / a2 * 1e-9,
                   gb / b4 * 1e-9, gb / b2 * 1e-9);
            fflush(stdout);
            float c1 = time_ms([&] { hipLaunchKernelGGL((k_read_rows_j<4, 8>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float c2 = time_ms([&] { hipLaunchKernelGGL((k_read_rows_j<2, 8>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float c3 = time_ms([&] { hipLaunchKernelGGL((k_read_rows_j<2, 16>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float c4 = time_ms([&] { hipLaunchKernelGGL((k_read_rows_j<1, 16>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float c5 = time_ms([&] { hipLaunchKernelGGL((k_read_rows_j<1, 8>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            float c6 = time_ms([&] { hipLaunchKernelGGL((k_read_rows_j<1, 4>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            printf("           operand-map  R4x256B %.2f  R2x256B %.2f  R2x512B %.2f  R1x512B %.2f  R1x256B %.2f  R1x128B %.2f TB/s\n",
                   gb / c1 * 1e-9, gb / c2 * 1e-9, gb / c3 * 1e-9, gb / c4 * 1e-9, gb / c5 * 1e-9, gb / c6 * 1e-9);
            fflush(stdout);
        }
    }
    {
        const int64_t M = 1000064, ld = 10240, rpw = 40064;   // K2's plan at C2: 20 n-groups x 25 row chunks = 500 workgroups
        const int g = 500;
        const double gb = (double)M * ld;
        float a0 = time_ms([&] { hipLaunchKernelGGL((k_read_cols<4, 0>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
        float a1 = time_ms([&] { hipLaunchKernelGGL((k_read_cols<4, 1>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
        float a2 = time_ms([&] { hipLaunchKernelGGL((k_read_cols<2, 1>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
        float b0 = time_ms([&] { hipLaunchKernelGGL((k_read_cols_wide<4, 0>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
        float b1 = time_ms([&] { hipLaunchKernelGGL((k_read_cols_wide<4, 1>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
        float b2 = time_ms([&] { hipLaunchKernelGGL((k_read_cols_wide<8, 1>), dim3(g), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
        printf("column slabs (K2 plan): dword loads NB4 %.2f  NB4 nt %.2f  NB2 nt %.2f | b128 pieces NB4 %.2f  NB4 nt %.2f  NB8 nt %.2f TB/s\n",
               gb / a0 * 1e-9, gb / a1 * 1e-9, gb / a2 * 1e-9, gb / b0 * 1e-9, gb / b1 * 1e-9, gb / b2 * 1e-9);
    }
    {
        const int64_t M = 1000064, ld = 10240;
        const double gb = (double)M * ld;
#define PIECES(R, WB, NT) time_ms([&] { hipLaunchKernelGGL((k_read_pieces<R, WB, NT>), dim3(256), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5)
        printf("K1 fill shape, 256 WGs, R2 (8 KiB in flight per wave-step): 8r x 128B %.2f nt %.2f | 4r x 256B %.2f nt %.2f | 2r x 512B %.2f nt %.2f TB/s\n",
               gb / PIECES(2, 128, 0) * 1e-9, gb / PIECES(2, 128, 1) * 1e-9, gb / PIECES(2, 256, 0) * 1e-9, gb / PIECES(2, 256, 1) * 1e-9,
               gb / PIECES(2, 512, 0) * 1e-9, gb / PIECES(2, 512, 1) * 1e-9);
        printf("K1 fill shape, 256 WGs, R4 (16 KiB in flight per wave-step): 8r x 128B %.2f nt %.2f | 4r x 256B %.2f nt %.2f | 2r x 512B %.2f nt %.2f TB/s\n",
               gb / PIECES(4, 128, 0) * 1e-9, gb / PIECES(4, 128, 1) * 1e-9, gb / PIECES(4, 256, 0) * 1e-9, gb / PIECES(4, 256, 1) * 1e-9,
               gb / PIECES(4, 512, 0) * 1e-9, gb / PIECES(4, 512, 1) * 1e-9);
        fflush(stdout);
    }
    {
        // Row pitch vs HBM channel interleave: the same 8-row x 128-B fill shape over rows 10 240 B apart (40 x 256 B: what a
        // 10 000-sample int8 matrix padded to 256 gets) and over odd multiples of 256 B.
        const int64_t M = 1000064;
        for (int64_t ld : {10240, 10496, 10368, 12288, 12544}) {
            const double gb = (double)M * ld;
            float a = time_ms([&] { hipLaunchKernelGGL((k_read_pieces<4, 128, 1>), dim3(256), dim3(256), 0, 0, (const char*)p, M, ld, out); }, 5);
            const int64_t rpw = 40064;
            float b = time_ms([&] { hipLaunchKernelGGL((k_read_cols_wide<4, 1>), dim3((unsigned)(ld / 512 * 25)), dim3(256), 0, 0, (const char*)p, M, ld, rpw, out); }, 5);
            printf("pitch %6lld B (%lld x 256): K1 fill shape nt %.2f TB/s | K2 column slabs nt %.2f TB/s\n", (long long)ld, (long long)(ld / 256), gb / a * 1e-9, gb / b * 1e-9);
        }
        fflush(stdout);
    }
    {
        // Infinity Cache (256 MiB MALL).  (a) a buffer that stays resident, re-read 20 x back to back: the on-die read rate.
        // (b) what pairing K1 -> K2 per row chunk would buy: the 10 GiB buffer read twice, either as two full
        // passes (2 launches: the second pass misses everywhere) or chunk by chunk with the second read of a chunk right behind the
        // first (2 x 10 GiB / chunk launches: the second read of a chunk can hit the MALL).  grid 4096 keeps a chunk launch
        // from being all ramp-up.
        for (int64_t mib : {32, 64, 128, 192, 512}) {
            const int64_t nb = (mib << 20) / 16;
            float t = time_ms([&] { for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k_read<4, 0>), dim3(4096), dim3(256), 0, 0, p, nb, out); }, 3);
            printf("re-read of a %4lld MiB buffer (20 launches back to back): %.2f TB/s, %.1f us per launch\n", (long long)mib,
                   20.0 * (double)(mib << 20) / t * 1e-9, t * 1e3 / 20);
        }
        const double two = 2.0 * (double)(10LL << 30);
        const int64_t n10 = (10LL << 30) / 16;
        float full = time_ms([&] { for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_read<4, 0>), dim3(16384), dim3(256), 0, 0, p, n10, out); }, 3);
        float fullnt = time_ms([&] { for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_read<4, 1>), dim3(16384), dim3(256), 0, 0, p, n10, out); }, 3);
        printf("10 GiB read twice, two full passes: %.3f ms (%.2f TB/s); nt: %.3f ms (%.2f TB/s)\n", full, two / full * 1e-9, fullnt, two / fullnt * 1e-9);
        for (int64_t mib : {64, 128, 192}) {
            const int64_t nb = (mib << 20) / 16, chunks = (10LL << 30) / (mib << 20);
            float t = time_ms([&] {
                for (int64_t c = 0; c < chunks; ++c) {
                    hipLaunchKernelGGL((k_read<4, 0>), dim3(4096), dim3(256), 0, 0, p + c * nb, nb, out);
                    hipLaunchKernelGGL((k_read<4, 0>), dim3(4096), dim3(256), 0, 0, p + c * nb, nb, out);
                }
            }, 3);
            printf("10 GiB read twice in paired %3lld MiB chunks (%lld launches): %.3f ms (%.2f TB/s)\n", (long long)mib, (long long)(2 * chunks), t,
                   two / t * 1e-9);
        }
    }
    return 0;
}
