// Where k_omega's time goes (the sketch of the exact path: Philox -> Box-Muller in f64 -> f32 -> digit planes of T' = r o Omega).
// The product source compiled with GPCA_OMEGA_ABLATE bits: 1 no transcendentals (z = u), 2 no Philox (counter as bits), 4 no digit planes.
//   hipcc --offload-arch=gfx950 -O3 -DGPCA_OMEGA_ABLATE=<bits> -o kbench_omega kbench_omega.hip && ./kbench_omega [M]
#include "../../genomic_pca_amd/csrc/kernels.hip"
#include "../../genomic_pca_amd/csrc/wide_sketch.hip"
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000064, Mpad = (M + 127) / 128 * 128;
    float *r, *b, *cpart, *rmax; int8_t* Td; double *tscale, *tinv;
    CK(hipMalloc(&r, Mpad * 4)); CK(hipMalloc(&b, Mpad * 4)); CK(hipMalloc(&cpart, Mpad * 4)); CK(hipMalloc(&rmax, 4));
    CK(hipMalloc(&Td, Mpad * 32 * 4)); CK(hipMalloc(&tscale, 64 * 8)); CK(hipMalloc(&tinv, 64 * 8));
    CK(hipMemset(r, 0x3f, Mpad * 4)); CK(hipMemset(b, 0, Mpad * 4));
    const float one = 1.f; CK(hipMemcpy(rmax, &one, 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) gpca::launch_omega_planes(0, M, Mpad, 30, 32, 0, 1, r, b, cpart, Td, rmax, tscale, tinv, 4, nullptr);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) gpca::launch_omega_planes(0, M, Mpad, 30, 32, 0, 1, r, b, cpart, Td, rmax, tscale, tinv, 4, nullptr);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_omega ablate %d, %lld rows x 30 of 32 columns: %.1f us per launch\n", GPCA_OMEGA_ABLATE, (long long)M, ms / 20 * 1e3);
    return 0;
}
