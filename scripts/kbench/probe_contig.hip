// What is different about memory from hipExtMallocWithFlags(hipDeviceMallocContiguous)?  (The parity suite failed with the genotype
// matrix in such memory.)  Alignment of the pointer, memset / kernel write / copy round trips, reuse after free.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -o probe_contig probe_contig.hip && ./probe_contig
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i * 2654435761u ^ seed;
}
__global__ void k_check(const uint32_t* p, size_t n, uint32_t seed, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (p[i] != ((uint32_t)i * 2654435761u ^ seed)) atomicAdd(bad, 1ull);
}
__global__ void k_count_nonzero(const uint32_t* p, size_t n, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) if (p[i]) atomicAdd(bad, 1ull);
}
static int run(bool contig, size_t bytes) {
    const size_t n = bytes / 4;
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    auto badcount = [&]() -> unsigned long long { unsigned long long v = 0; (void)hipMemcpy(&v, bad, 8, hipMemcpyDeviceToHost); (void)hipMemset(bad, 0, 8); return v; };
    CK(hipMemset(bad, 0, 8));
    for (int round = 0; round < 3; ++round) {
        uint32_t* p = nullptr;
        if (contig) CK(hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocContiguous)); else CK(hipMalloc((void**)&p, bytes));
        hipPointerAttribute_t at{}; (void)hipPointerGetAttributes(&at, p);
        hipLaunchKernelGGL(k_count_nonzero, dim3(4096), dim3(256), 0, 0, p, n, bad); CK(hipDeviceSynchronize());
        printf("%s round %d: ptr %p (mod 2 MiB = %zu, mod 4 KiB = %zu) type %d, nonzero words on arrival %llu of %zu\n", contig ? "contig" : "plain ", round, (void*)p,
               (size_t)((uintptr_t)p & ((2u << 20) - 1)), (size_t)((uintptr_t)p & 4095), (int)at.type, badcount(), n);
        CK(hipMemset(p, 0, bytes));
        hipLaunchKernelGGL(k_count_nonzero, dim3(4096), dim3(256), 0, 0, p, n, bad); CK(hipDeviceSynchronize());
        printf("   after hipMemset 0: nonzero %llu\n", badcount());
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        CK(hipMemsetAsync(p, 0, bytes, st));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, p, n, 7u + round);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, st, p, n, 7u + round, bad); CK(hipStreamSynchronize(st));
        printf("   kernel write -> kernel read on a non-blocking stream: %llu bad\n", badcount());
        std::vector<uint32_t> hst(n);
        CK(hipMemcpy(hst.data(), p, bytes, hipMemcpyDeviceToHost));
        size_t hb = 0; for (size_t i = 0; i < n; ++i) hb += hst[i] != ((uint32_t)i * 2654435761u ^ (7u + round));
        printf("   hipMemcpy D2H: %zu bad\n", hb);
        // 2-D copy of a sub-rectangle, row by row copies, host -> device -> kernel check
        const size_t pitch = 2816, rows = bytes / pitch / 2, width = 700;
        std::vector<uint8_t> h2(rows * width);
        CK(hipMemcpy2D(h2.data(), width, p, pitch, width, rows, hipMemcpyDeviceToHost));
        hb = 0; for (size_t r = 0; r < rows; r += 97) for (size_t c = 0; c < width; ++c) hb += h2[r * width + c] != ((const uint8_t*)hst.data())[r * pitch + c];
        printf("   hipMemcpy2D D2H: %zu bad\n", hb);
        for (size_t i = 0; i < n; ++i) hst[i] = (uint32_t)i * 2654435761u ^ (99u + round);
        CK(hipMemcpyAsync(p, hst.data(), bytes, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, st, p, n, 99u + round, bad); CK(hipStreamSynchronize(st));
        printf("   H2D async (pageable) -> kernel read: %llu bad\n", badcount());
        uint32_t* q = nullptr; CK(hipMalloc((void**)&q, bytes));
        CK(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, st, q, n, 99u + round, bad); CK(hipStreamSynchronize(st));
        printf("   D2D async out of it -> kernel read: %llu bad\n", badcount());
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, q, n, 5u);
        CK(hipMemcpyAsync(p, q, bytes, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, st, p, n, 5u, bad); CK(hipStreamSynchronize(st));
        printf("   kernel -> D2D async into it -> kernel read: %llu bad\n", badcount());
        CK(hipFree(q)); CK(hipFree(p)); CK(hipStreamDestroy(st));
    }
    return 0;
}
int main() {
    if (run(false, (size_t)64 << 20)) return 1;
    if (run(true, (size_t)64 << 20)) return 1;
    if (run(true, (size_t)17 << 20)) return 1;
    return 0;
}
