// prints the D layout of v_mfma_f64_16x16x4_f64 on gfx950: D[i][j] = (i+1) * 100 + (j+1) built from A[i][0] = 1, A[i][1] = i+1,
// B[0][j] = j+1, B[1][j] = 100 (other k zero)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const double a = kk == 0 ? 1.0 : (kk == 1 ? (double)(i + 1) : 0.0);
    const double b = kk == 0 ? (double)(i + 1) : (kk == 1 ? 100.0 : 0.0);
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane += 1) if (lane < 3 || (lane & 15) == 0) {
        printf("lane %2d:", lane);
        for (int r = 0; r < 4; ++r) { const int v = (int)h[lane * 4 + r]; printf("  r%d -> D[i=%d][j=%d]", r, v / 100 - 1, v % 100 - 1); }
        printf("\n");
    }
    return 0;
}
