// Old and new forms of two launches of the call's tail, alternating in ONE process (cross-box comparisons of 15-60 us launches are
// worthless: the clock a part holds through its short launches differs by box).  The old kernels live here for the comparison only.
//   k_rightmul_mfma (loadings = gathered rows of T x Z): the product's (one tile ahead, id load in front of the row load) vs rows two tiles
//   ahead and ids three (built in round 4, measured level, not adopted: it lives HERE as k_rightmul_mfma_deep)
//   k_col_sign: 256 threads vs 1 024 threads with four loads per trip
//   hipcc --offload-arch=gfx950 -O3 -o kbench_tail kbench_tail.hip && ./kbench_tail
#include "../../genomic_pca_amd/csrc/kernels.hip"
#include "../../genomic_pca_amd/csrc/wide_sketch.hip"
#include "../../genomic_pca_amd/csrc/fold_quantize_i8.hip"
#include <cstdio>
#include <vector>
#include <cstring>
namespace gpca {
template <int L, int NJ>
__global__ __launch_bounds__(256) void k_rightmul_mfma_deep(const float* __restrict__ X, const int64_t* __restrict__ row_ids,
                                                       int64_t nrows, const double* __restrict__ Z, int K,
                                                       float* __restrict__ out32, int64_t tiles_per_wave) {
    constexpr int E = L / 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    double zb[E][NJ];
#pragma unroll
    for (int s2 = 0; s2 < E; ++s2)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int col = 16 * jt + i;
            zb[s2][jt] = col < K ? Z[(E * kq + s2) * K + col] : 0.0;
        }
    __shared__ float osm[4][16 * 64];                   // per-wave output tile, written back as one contiguous run
    const int64_t ntiles = (nrows + 15) >> 4;
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + wv) * tiles_per_wave;
    const int64_t t1 = (t0 + tiles_per_wave < ntiles) ? t0 + tiles_per_wave : ntiles;
    // A tile's rows are requested two tiles ahead and their ids (the gather through row_ids) three: the id load and the row load it
    // feeds used to sit back to back in front of every tile -- two dependent round trips with 2 KiB per wave in flight, 3.1 TB/s.
    float xa[E], xb[E], xc[E];
    auto tile_src = [&](int64_t tile) -> int64_t {          // source row of this lane's row of the tile, -1 = none
        const int64_t row = tile * 16 + i;
        if (!(row < nrows && tile < t1)) return -1;
        return row_ids ? row_ids[row] : row;
    };
    auto load_tile = [&](int64_t src, float (&dst)[E]) {
        const bool valid = src >= 0;
        const float4* xp = reinterpret_cast<const float4*>(X + (valid ? src : 0) * L + E * kq);
#pragma unroll
        for (int v = 0; v < E / 4; ++v) {
            const float4 q = xp[v];
            dst[4 * v] = valid ? q.x : 0.f; dst[4 * v + 1] = valid ? q.y : 0.f; dst[4 * v + 2] = valid ? q.z : 0.f; dst[4 * v + 3] = valid ? q.w : 0.f;
        }
    };
    int64_t s2a = -1, s3a = -1;                              // ids of tiles +2 and +3
    if (t0 < t1) {
        load_tile(tile_src(t0), xa);
        load_tile(tile_src(t0 + 1), xb);
        s2a = tile_src(t0 + 2);
    }
    for (int64_t tile = t0; tile < t1; ++tile) {
        s3a = tile_src(tile + 3);
        load_tile(s2a, xc);                                  // tile + 2: in flight behind two tiles of MFMAs and stores
        s2a = s3a;
        f64x4 acc[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < E; ++s2) {
            const double a = (double)xa[s2];
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zb[s2][jt], acc[jt], 0, 0, 0);
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int col = 16 * jt + i;
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) osm[wv][(4 * r + kq) * K + col] = (float)acc[jt][r];
            }
        }
        // (one wave: its LDS operations execute in issue order, so the reads below see the writes above)
        const int64_t rows_here = (nrows - tile * 16 < 16) ? nrows - tile * 16 : 16;
        float* dst = out32 + tile * 16 * K;
        for (int e = lane; e < (int)rows_here * K; e += 64) dst[e] = osm[wv][e];
#pragma unroll
        for (int s2 = 0; s2 < E; ++s2) { xa[s2] = xb[s2]; xb[s2] = xc[s2]; }
    }
}
// sign of the first element with maximal |x| per column; one block per column
__global__ __launch_bounds__(256) void k_col_sign_old(const double* __restrict__ X, int64_t rows, int K, int* __restrict__ sign) {
    __shared__ double bv[256];
    __shared__ long long bi[256];
    const int col = blockIdx.x;
    double best = -1.0; long long idx = -1;
    for (int64_t n = threadIdx.x; n < rows; n += 256) {
        const double a = fabs(X[n * K + col]);
        if (a > best) { best = a; idx = n; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double ob = bv[threadIdx.x + s]; const long long oi = bi[threadIdx.x + s];
            if (ob > bv[threadIdx.x] || (ob == bv[threadIdx.x] && oi >= 0 && (bi[threadIdx.x] < 0 || oi < bi[threadIdx.x]))) {
                bv[threadIdx.x] = ob; bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) sign[col] = (bi[0] >= 0 && X[bi[0] * K + col] < 0.0) ? -1 : 1;
}
// the fold of K2's partial tiles as rounds 1-3 wrote it (the compiler's own unroll of a dependent chain of adds)
__global__ __launch_bounds__(256) void k_reduce_y_old(const double* __restrict__ Ypart, int W, int64_t Npad, int64_t N,
                                                      const double* __restrict__ cvec, const double* __restrict__ tscale,
                                                      double* __restrict__ Y, int64_t ldy) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * 32) return;
    double s = 0.0;
    for (int w = 0; w < W; ++w) s += Ypart[w * (Npad * 32) + e];
    const int j = (int)(e & 31);
    Y[(e >> 5) * ldy + j] = fma(tscale[j], s, cvec[j]);
}
}  // namespace gpca
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fillf(float* p, int64_t n) { for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = (float)((i * 2654435761u) & 0xffff) / 65536.f - 0.5f; }
__global__ void k_filld(double* p, int64_t n) { for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5; }
__global__ void k_ids(int64_t* p, int64_t n) { for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = i; }
int main() {
    const int64_t M = 1000000, N = 10000; const int L = 32, K = 20;
    float *T, *o0, *o1; int64_t* ids; double *Z, *S; int *sg0, *sg1;
    CK(hipMalloc(&T, M * L * 4)); CK(hipMalloc(&o0, M * K * 4)); CK(hipMalloc(&o1, M * K * 4)); CK(hipMalloc(&ids, M * 8)); CK(hipMalloc(&Z, L * K * 8));
    CK(hipMalloc(&S, N * K * 8)); CK(hipMalloc(&sg0, 256)); CK(hipMalloc(&sg1, 256));
    hipLaunchKernelGGL(k_fillf, dim3(4096), dim3(256), 0, 0, T, M * L); hipLaunchKernelGGL(k_filld, dim3(8), dim3(256), 0, 0, Z, (int64_t)L * K);
    hipLaunchKernelGGL(k_filld, dim3(256), dim3(256), 0, 0, S, N * K); hipLaunchKernelGGL(k_ids, dim3(1024), dim3(256), 0, 0, ids, M);
    const int64_t ntiles = (M + 15) / 16; int64_t tpw = ntiles / (4 * 2048); if (tpw < 1) tpw = 1;
    const int64_t waves = (ntiles + tpw - 1) / tpw; const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double t[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 6; ++rep)
        for (int v = 0; v < 4; ++v) {
            auto go = [&]() {
                if (v == 0) hipLaunchKernelGGL((gpca::k_rightmul_mfma<32, 2>), grid, blk, 0, 0, T, ids, M, Z, K, o0, tpw);
                else if (v == 1) hipLaunchKernelGGL((gpca::k_rightmul_mfma_deep<32, 2>), grid, blk, 0, 0, T, ids, M, Z, K, o1, tpw);
                else if (v == 2) hipLaunchKernelGGL(gpca::k_col_sign_old, dim3(K), dim3(256), 0, 0, S, N, K, sg0);
                else hipLaunchKernelGGL(gpca::k_col_sign, dim3(K), dim3(1024), 0, 0, S, N, K, sg1);
            };
            go();
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) go();
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); t[v] += ms / 20 * 1e3;
        }
    std::vector<float> a(M * K), b(M * K); std::vector<int> s0(K), s1(K);
    CK(hipMemcpy(a.data(), o0, M * K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o1, M * K * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(s0.data(), sg0, K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s1.data(), sg1, K * 4, hipMemcpyDeviceToHost));
    printf("k_rightmul_mfma<32, 2> 1M gathered rows x 20: one tile ahead %.2f us, two ahead + ids three ahead %.2f us; results %s\n", t[0] / 6, t[1] / 6,
           memcmp(a.data(), b.data(), a.size() * 4) == 0 ? "bit-identical" : "DIFFER");
    {   // fold: 51 partial tiles of 10 240 x 32 doubles (134 MB), written just before by a streaming kernel's worth of stores
        const int W = 51; const int64_t Npad = 10240;
        double *Yp, *Y0, *Y1, *cv, *ts;
        CK(hipMalloc(&Yp, (size_t)W * Npad * 32 * 8)); CK(hipMalloc(&Y0, N * 32 * 8)); CK(hipMalloc(&Y1, N * 32 * 8)); CK(hipMalloc(&cv, 256)); CK(hipMalloc(&ts, 256));
        hipLaunchKernelGGL(k_filld, dim3(4096), dim3(256), 0, 0, Yp, (int64_t)W * Npad * 32); hipLaunchKernelGGL(k_filld, dim3(1), dim3(64), 0, 0, cv, (int64_t)32); hipLaunchKernelGGL(k_filld, dim3(1), dim3(64), 0, 0, ts, (int64_t)32);
        double tf[2] = {0, 0};
        for (int rep = 0; rep < 6; ++rep)
            for (int v = 0; v < 2; ++v) {
                auto go = [&]() {
                    hipLaunchKernelGGL(k_fillf, dim3(4096), dim3(256), 0, 0, T, M * L);      // 128 MB through L2 between folds: the tiles are not cache-resident
                    if (v == 0) hipLaunchKernelGGL(gpca::k_reduce_y_old, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, 0, Yp, W, Npad, N, cv, ts, Y0, (int64_t)32);
                    else gpca::launch_reduce_y_i8(0, Yp, W, Npad, N, cv, ts, Y1, 32);
                };
                go();
                hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) go();
                hipEventRecord(e1); CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1); tf[v] += ms / 10 * 1e3;
            }
        std::vector<double> ya(N * 32), yb(N * 32);
        CK(hipMemcpy(ya.data(), Y0, N * 32 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(yb.data(), Y1, N * 32 * 8, hipMemcpyDeviceToHost));
        printf("fold of 51 partial tiles (+ a 128 MB fill in front of each, same for both): plain loop %.2f us, eight loads in flight %.2f us; results %s\n", tf[0] / 6, tf[1] / 6,
               memcmp(ya.data(), yb.data(), ya.size() * 8) == 0 ? "bit-identical" : "DIFFER");
    }
    printf("k_col_sign 10 000 x 20: 256 threads %.2f us, 1 024 threads %.2f us; results %s\n", t[2] / 6, t[3] / 6, s0 == s1 ? "identical" : "DIFFER");
    return 0;
}
