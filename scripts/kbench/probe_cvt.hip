// probe: what does v_cvt_pk_f32_fp8 return for raw bytes 0..3 in each byte position (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int* in, float* out) {
    int w = in[threadIdx.x];
    f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    out[4 * threadIdx.x + 0] = lo[0]; out[4 * threadIdx.x + 1] = lo[1]; out[4 * threadIdx.x + 2] = hi[0]; out[4 * threadIdx.x + 3] = hi[1];
}
int main() {
    int h[4] = {0x03020100, 0x00010203, 0x02020101, 0x07060504}; int* d; float* o; float r[16];
    hipMalloc(&d, 16); hipMalloc(&o, 64); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o); hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("word %08x -> %g %g %g %g   (x512: %g %g %g %g)\n", h[i], r[4*i], r[4*i+1], r[4*i+2], r[4*i+3], r[4*i]*512, r[4*i+1]*512, r[4*i+2]*512, r[4*i+3]*512);
    return 0;
}
