// Does the way a 10.5 GB buffer is BUILT decide how fast it streams?  (summary section 9: a process's first large hipMalloc streams at
// 6.75 TB/s in some processes and 6.93 in others.)  The same size from: plain hipMalloc; the virtual-memory API (hipMemAddressReserve +
// hipMemCreate / hipMemMap) in physical chunks of 2 MiB ... 1 GiB.  Each buffer written once, then the linear nt read probe and
// K1's fill shape, two passes round-robin.
//   hipcc --offload-arch=gfx950 -O3 -o kbench_vmm kbench_vmm.hip && ./kbench_vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, int64_t n) { for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = (uint32_t)i * 2654435761u; }
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
static int vmm_alloc(char** out, size_t bytes, size_t chunk, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 1;
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, chunk < ((size_t)1 << 30) ? chunk : ((size_t)1 << 30), nullptr, 0) != hipSuccess) return 2;
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t hnd;
        if (hipMemCreate(&hnd, chunk, &prop, 0) != hipSuccess) return 3;
        if (hipMemMap((char*)va + off, chunk, 0, hnd, 0) != hipSuccess) return 4;
        (void)hipMemRelease(hnd);
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) return 5;
    *out = (char*)va;
    return 0;
}
int main() {
    const size_t bytes = (size_t)1000064 * 10496;
    struct B { std::string name; char* p; };
    std::vector<B> bufs;
    int* out; CK(hipMalloc(&out, 4));
    { char* p; CK(hipMalloc(&p, bytes)); bufs.push_back({"hipMalloc #1", p}); }
    for (size_t chunk : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
        char* p = nullptr; const int rc = vmm_alloc(&p, bytes, chunk, 0);
        char nm[64]; snprintf(nm, sizeof nm, "VMM, %zu MiB chunks", chunk >> 20);
        if (rc) { printf("%s: failed at step %d (%s)\n", nm, rc, hipGetErrorString(hipGetLastError())); continue; }
        bufs.push_back({nm, p});
    }
    { char* p; CK(hipMalloc(&p, bytes)); bufs.push_back({"hipMalloc #2", p}); }
    { char* p; CK(hipMalloc(&p, bytes)); bufs.push_back({"hipMalloc #3", p}); }
    for (auto& b : bufs) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)b.p, (int64_t)(bytes / 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 3; ++pass)
        for (auto& b : bufs) {
            hipLaunchKernelGGL(k_linear, dim3(16384), dim3(256), 0, 0, (const i32x4*)b.p, (int64_t)(bytes / 16), out);
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_linear, dim3(16384), dim3(256), 0, 0, (const i32x4*)b.p, (int64_t)(bytes / 16), out);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("pass %d  %-22s at %p: linear nt read %.3f ms = %.2f TB/s\n", pass, b.name.c_str(), (void*)b.p, ms, bytes / ms * 1e-9);
        }
    return 0;
}
