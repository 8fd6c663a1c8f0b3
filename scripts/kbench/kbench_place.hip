// Does the PLACEMENT of a 10.5 GB buffer move a streaming kernel?  (Round 2: two engines holding the same matrix in one process
// differed by up to 8 % on K1 and K2 alike.)  Allocates NB buffers of the C2 matrix size side by side and streams each of them with
// K1's fill shape (8 rows x 128 B pieces, nt) and with plain contiguous 16-byte loads; prints the device address of each.
//   hipcc --offload-arch=gfx950 -O3 -o kbench_place kbench_place.hip && ./kbench_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int R>
__global__ __launch_bounds__(256, 1) void k_pieces(const char* __restrict__ p, int64_t M, int64_t ld, int* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t units = M / (32 * R);
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t u = (units * wave) / waves; u < (units * (wave + 1)) / waves; ++u) {
        const char* base = p + u * 32 * R * ld;
        for (int64_t col = 0; col + 128 <= ld; col += 128) {
            i32x4 v[R][4];
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[t][j] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(base + (int64_t)(32 * t + 8 * j + (lane >> 3)) * ld + col + 16 * (lane & 7)));
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc ^= v[t][j];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_linear(const i32x4* __restrict__ p, int64_t n16, int* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    i32x4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n16; i += 4 * stride) {
        i32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        i32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
// usage: kbench_place [contig]   (contig: every second buffer is asked for with hipDeviceMallocContiguous -- physically contiguous,
// the largest page fragments the driver can give -- to see whether the spread between buffers is a matter of fragments)
int main(int argc, char** argv) {
    const bool contig = argc > 1 && argv[1][0] == 'c';
    const int64_t M = 1000064, ld = 10496;
    const int64_t bytes = M * ld;
    const int NB = 10;
    std::vector<char*> buf(NB);
    int* out; CK(hipMalloc(&out, 4)); CK(hipMemset(out, 0, 4));
    for (int b = 0; b < NB; ++b) {
        if (contig && (b & 1)) {
            const hipError_t e = hipExtMallocWithFlags((void**)&buf[b], bytes, hipDeviceMallocContiguous);
            printf("buffer %2d: hipDeviceMallocContiguous -> %s\n", b, hipGetErrorString(e));
            if (e != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc(&buf[b], bytes)); }
        } else CK(hipMalloc(&buf[b], bytes));
        CK(hipMemset(buf[b], 1, bytes));
    }
    for (int pass = 0; pass < 2; ++pass)
        for (int b = 0; b < NB; ++b) {
            float t1 = time_ms([&] { hipLaunchKernelGGL((k_pieces<4>), dim3(256), dim3(256), 0, 0, (const char*)buf[b], M, ld, out); }, 5);
            float t2 = time_ms([&] { hipLaunchKernelGGL(k_linear, dim3(16384), dim3(256), 0, 0, (const i32x4*)buf[b], bytes / 16, out); }, 5);
            printf("pass %d buffer %2d at %p: row pieces nt %.3f ms = %.2f TB/s | linear nt %.3f ms = %.2f TB/s\n", pass, b, (void*)buf[b], t1,
                   bytes / t1 * 1e-9, t2, bytes / t2 * 1e-9);
            fflush(stdout);
        }
    return 0;
}
