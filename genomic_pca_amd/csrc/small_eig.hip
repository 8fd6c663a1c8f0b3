// The small dense step of the randomized PCA on the device: the symmetric eigenproblem of the l x l Gram of the
// projection B = A Q, and everything that used to hang on its result on the host -- the descending sort, singular values, eigenvalues,
// the two l x k factors that turn Q into scores and B into loadings.  One workgroup; the call's stream never waits for the host
// (main.rs:648-660 is one opaque call in the reference too).
//
// Algorithm: Householder tridiagonalisation + implicit QL (the EISPACK tred2 / tql2 pair, the same pair gpca_host_eigh_desc runs on
// the CPU as the pin of tests/test_abi.py), arranged for one wave (n <= 64) or two (n <= 128):
//   * the matrix lives in LDS with an odd row pitch (a column walk across lanes and a row walk across lanes are both conflict-free);
//   * tred2: thread t owns row / column t.  The inner products and the rank-2 update of a step run across the threads, the three
//     scalar reductions of a step (scale, h, f) are summed by every thread from LDS in the host's order;
//   * tql2: thread t owns ROW t of the eigenvector matrix, so a plane rotation touches only the thread's own two elements -- no
//     barrier inside the QL sweeps.  The rotation parameters are a serial chain (1 / sqrt by the hardware estimate + two Newton
//     steps instead of hypot + two divisions: ~110 dependent cycles per rotation); every lane computes them redundantly from its
//     wave's private copy of d[] / e[] (lane 0 writes), the next rotation's d[i], e[i] and matrix element are fetched one rotation ahead.
//   * every wave reaches every exit: the QL loop is bounded (200 sweeps per eigenvalue, as on the host), NaN input compares false
//     in the deflation test and falls through.
// The input is scaled by a power of two (exact) so that its largest entry is in [0.5, 1): the sums of squares of the rotation chain
// can neither overflow nor underflow for any finite Gram matrix.
#include "kernels.h"

namespace gpca {

__device__ __forceinline__ void eig_wsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
template <int NT> __device__ __forceinline__ void eig_tsync() { if (NT <= 64) eig_wsync(); else __syncthreads(); }

__device__ __forceinline__ double eig_rsqrt(double x) {       // 1 / sqrt(x), x normal and positive: hardware estimate + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// Householder reduction to tridiagonal form; V (n x n, pitch P, symmetric on entry) ends as the accumulated transformation,
// d = diagonal, e = sub-diagonal (e[0] = 0).  t = thread index; threads >= n idle along (they still reach every barrier).
template <int NT>
__device__ void eig_tred2(int n, int P, double* __restrict__ V, double* __restrict__ d, double* __restrict__ e, int t) {
    if (t < n) d[t] = V[(n - 1) * P + t];
    eig_tsync<NT>();
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int kk = 0; kk < i; ++kk) scale += fabs(d[kk]);
        if (scale == 0.0) {
            const double dim1 = d[i - 1];
            eig_tsync<NT>();
            if (t == 0) e[i] = dim1;
            if (t < i) { d[t] = V[(i - 1) * P + t]; V[i * P + t] = 0.0; V[t * P + i] = 0.0; }
        } else {
            eig_tsync<NT>();                                   // every thread has summed |d|
            if (t < i) d[t] = d[t] / scale;
            eig_tsync<NT>();
            for (int kk = 0; kk < i; ++kk) h = fma(d[kk], d[kk], h);
            const double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            h -= f * g;
            eig_tsync<NT>();                                   // ... and read d[i - 1]
            if (t == i - 1) d[i - 1] = f - g;
            if (t == 0) e[i] = scale * g;
            eig_tsync<NT>();
            // p = A u / h with A = the current symmetric matrix (lower triangle valid), u = d[0..i)
            double pj = 0.0;
            if (t < i) {
                V[t * P + i] = d[t];                           // the Householder vector stays in column i for the accumulation below
                for (int kk = 0; kk <= t; ++kk) pj = fma(V[t * P + kk], d[kk], pj);
                for (int kk = t + 1; kk < i; ++kk) pj = fma(V[kk * P + t], d[kk], pj);
                pj /= h;
                e[t] = pj;
            }
            eig_tsync<NT>();
            double ff = 0.0;
            for (int j = 0; j < i; ++j) ff = fma(e[j], d[j], ff);
            const double hh = ff / (h + h);
            eig_tsync<NT>();
            if (t < i) e[t] = pj - hh * d[t];
            eig_tsync<NT>();
            if (t < i) {                                       // rank-2 update of the thread's own row of the lower triangle
                const double et = e[t], dt = d[t];
                for (int j = 0; j <= t; ++j) V[t * P + j] -= (d[j] * et + e[j] * dt);
            }
            eig_tsync<NT>();
            if (t < i) { d[t] = V[(i - 1) * P + t]; V[i * P + t] = 0.0; }
        }
        if (t == 0) d[i] = h;
        eig_tsync<NT>();
    }
    for (int i = 0; i < n - 1; ++i) {                          // accumulate the transformations
        if (t == 0) { V[(n - 1) * P + i] = V[i * P + i]; V[i * P + i] = 1.0; }
        const double h = d[i + 1];
        eig_tsync<NT>();
        if (h != 0.0) {
            if (t <= i) d[t] = V[t * P + (i + 1)] / h;
            eig_tsync<NT>();
            if (t <= i) {                                      // the thread's own column
                double g = 0.0;
                for (int kk = 0; kk <= i; ++kk) g = fma(V[kk * P + (i + 1)], V[kk * P + t], g);
                for (int kk = 0; kk <= i; ++kk) V[kk * P + t] -= g * d[kk];
            }
            eig_tsync<NT>();
        }
        if (t <= i) V[t * P + (i + 1)] = 0.0;
        eig_tsync<NT>();
    }
    if (t < n) { d[t] = V[(n - 1) * P + t]; V[(n - 1) * P + t] = 0.0; }
    eig_tsync<NT>();
    if (t == 0) { V[(n - 1) * P + (n - 1)] = 1.0; e[0] = 0.0; }
    eig_tsync<NT>();
}

// Implicit QL on the tridiagonal (dd, ee: THIS WAVE's private copies); thread t rotates row t of V.  Returns 1 if a sweep count hit the cap.
__device__ int eig_tql2(int n, int P, double* __restrict__ V, double* __restrict__ dd, double* __restrict__ ee, int t, int lane) {
    const double eps = 2.220446049250313e-16;
    int capped = 0;
    {   // e[i - 1] = e[i], e[n - 1] = 0
        double e0 = (lane + 1 < n) ? ee[lane + 1] : 0.0, e1 = (lane + 65 < n) ? ee[lane + 65] : 0.0;
        eig_wsync();
        if (lane < n) ee[lane] = e0;
        if (lane + 64 < n) ee[lane + 64] = e1;
        eig_wsync();
    }
    const bool row = t < n;
    double f = 0.0, tst1 = 0.0;
    for (int l = 0; l < n; ++l) {
        tst1 = fmax(tst1, fabs(dd[l]) + fabs(ee[l]));
        int m = l;
        while (m < n - 1 && fabs(ee[m]) > eps * tst1) ++m;     // (ee[n - 1] = 0 ends the search; NaN compares false)
        if (m > l) {
            int iter = 0;
            double el_cur;
            do {
                ++iter;
                double g = dd[l];
                const double el0 = ee[l];
                double p = (dd[l + 1] - g) / (2.0 * el0);
                double r = fabs(p) < 1e150 ? sqrt(fma(p, p, 1.0)) : fabs(p);
                if (p < 0) r = -r;
                const double dl_new = el0 / (p + r), dl1 = el0 * (p + r);
                double h = g - dl_new;
                eig_wsync();                                   // every lane has read dd[l], dd[l + 1]
                for (int i0 = l + 2 + lane; i0 < n; i0 += 64) dd[i0] -= h;
                if (lane == 0) { dd[l] = dl_new; dd[l + 1] = dl1; }
                eig_wsync();
                f += h;
                p = dd[m];
                double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                const double el1 = ee[l + 1];
                double xhi = row ? V[t * P + m] : 0.0;
                double di = dd[m - 1], ei = ee[m - 1];
                for (int i = m - 1; i >= l; --i) {
                    const int ip = i > l ? i - 1 : l;          // the next rotation's operands, one rotation ahead
                    const double dn = dd[ip], en = ee[ip];
                    const double xlo = row ? V[t * P + i] : 0.0;
                    c3 = c2; c2 = c; s2 = s;
                    g = c * ei; h = c * p;
                    const double sq = fma(p, p, ei * ei);
                    double rr, rinv;
                    if (sq > 1e-290 && sq < 1e290) { rinv = eig_rsqrt(sq); rr = sq * rinv; }
                    else { rr = hypot(p, ei); rinv = rr > 0.0 ? 1.0 / rr : 0.0; }
                    const double e_up = s * rr;                // (the previous rotation's s)
                    s = ei * rinv; c = p * rinv;
                    p = c * di - s * g;
                    const double d_up = h + s * (c * g + s * di);
                    if (lane == 0) { ee[i + 1] = e_up; dd[i + 1] = d_up; }
                    if (row) { V[t * P + (i + 1)] = fma(s, xlo, c * xhi); xhi = fma(c, xlo, -(s * xhi)); }
                    di = dn; ei = en;
                }
                if (row) V[t * P + l] = xhi;
                p = -s * s2 * c3 * el1 * el0 / dl1;
                el_cur = s * p;
                eig_wsync();
                if (lane == 0) { ee[l] = el_cur; dd[l] = c * p; }
                eig_wsync();
            } while (fabs(el_cur) > eps * tst1 && iter < 200);
            if (fabs(el_cur) > eps * tst1) capped = 1;
        }
        const double dl = dd[l];
        eig_wsync();
        if (lane == 0) { dd[l] = dl + f; ee[l] = 0.0; }
        eig_wsync();
    }
    return capped;
}

// src: the Gram W [L][L] (nslices == 0) or `nslices` partial sums of it [nslices][L * L] (summed here in slice order); only the
// leading n x n block is used, symmetrised as (W + W^T) / 2.  Outputs:
//   res[kEigResSv + j]   = sqrt(max(w_j, 0)), j < n (0 beyond)          res[kEigResEig + c] = w_c / denom, c < k
//   res[kEigResW + j]    = w_j (descending)                             res[kEigResFlag] = *cholflag, res[kEigResFlag + 1] = sweep cap hit
//   Z [2][L][k]: zmode 0: Z0 = V_k diag(sv), Z1 = V_k diag(1 / sv) (0 where sv = 0); zmode 1: Z0 = Z1 = V_k.  Rows >= n are zero.
//   Vout (may be NULL) [n][n]: the eigenvectors in columns, sorted like w.
template <int NT>
__global__ __launch_bounds__(256) void k_small_eigh(const double* __restrict__ src, int nslices, int n, int L, int k, int zmode, double denom,
                                                    const int* __restrict__ cholflag, double* __restrict__ Z, double* __restrict__ res,
                                                    double* __restrict__ Vout) {
    extern __shared__ double eig_sm[];
    const int P = n | 1;
    double* V = eig_sm;                                        // [n][P]
    double* dsh = V + (size_t)n * P;                           // [2][128]: d per wave
    double* esh = dsh + 256;                                   // [2][128]: e per wave
    int* order = reinterpret_cast<int*>(esh + 256);            // [128]
    __shared__ double redmax[256];
    __shared__ int capsh;
    const int tid = threadIdx.x;
    const int S = nslices > 0 ? nslices : 1;
    const size_t LL = (size_t)L * L;
    // fold + symmetrise into LDS, all 256 threads; the loads of the slices are independent (the sum keeps slice order)
    double amax = 0.0;
    for (int e0 = tid; e0 < n * n; e0 += 256) {
        const int a = e0 / n, c = e0 - a * n;
        double s1 = 0.0, s2 = 0.0;
        const double* p1 = src + (size_t)a * L + c;
        const double* p2 = src + (size_t)c * L + a;
#pragma unroll 8
        for (int s = 0; s < S; ++s) { s1 += p1[s * LL]; s2 += p2[s * LL]; }
        const double v = 0.5 * (s1 + s2);
        V[a * P + c] = v;
        const double av = fabs(v);
        amax = (av > amax && av < INFINITY) ? av : amax;
    }
    redmax[tid] = amax;
    if (tid == 0) capsh = 0;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) redmax[tid] = fmax(redmax[tid], redmax[tid + st]); __syncthreads(); }
    amax = redmax[0];
    int ex = 0;
    if (amax > 0.0) (void)frexp(amax, &ex);
    const double sc = ldexp(1.0, -ex), unsc = ldexp(1.0, ex);
    for (int e0 = tid; e0 < n * n; e0 += 256) { const int a = e0 / n, c = e0 - a * n; V[a * P + c] *= sc; }
    __syncthreads();
    if (NT <= 64 && tid >= 64) return;                         // one-wave solver: no block barrier from here on
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NTHR = NT <= 64 ? 64 : 256;                  // threads still here (NT = 128: waves 2, 3 walk tred2's barriers as idle threads)
    eig_tred2<NT>(n, P, V, dsh, esh, tid);
    if (NT > 64) {                                             // wave 1's private copy of the tridiagonal
        if (tid < n) { dsh[128 + tid] = dsh[tid]; esh[128 + tid] = esh[tid]; }
        __syncthreads();
    }
    if (tid < NT) {
        const int capped = eig_tql2(n, P, V, dsh + 128 * wv, esh + 128 * wv, tid, lane);
        if (capped && lane == 0 && wv == 0) capsh = 1;
    }
    for (int j = tid; j < n; j += NTHR) order[j] = j;
    eig_tsync<NT>();
    // descending order: rank_j = #{i : w_i > w_j or (w_i == w_j and i < j)}  (= the host's stable selection sort)
    for (int j = tid; j < n; j += NTHR) {
        const double wj = dsh[j];
        int rank = 0;
        for (int i = 0; i < n; ++i) { const double wi = dsh[i]; rank += (wi > wj || (wi == wj && i < j)) ? 1 : 0; }
                order[rank] = j;
    }
    eig_tsync<NT>();
    for (int j = tid; j < kMaxSketchCols; j += NTHR) {
        double w = 0.0;
        if (j < n) w = dsh[order[j]] * unsc;
        res[kEigResW + j] = w;
        res[kEigResSv + j] = w > 0.0 ? sqrt(w) : 0.0;
        res[kEigResEig + j] = j < k ? w / denom : 0.0;
    }
    if (tid == 0) { res[kEigResFlag] = cholflag ? (double)cholflag[0] : 0.0; res[kEigResFlag + 1] = (double)capsh; }
    for (int e0 = tid; e0 < L * k; e0 += NTHR) {
        const int r = e0 / k, c = e0 - r * k;
        double z0 = 0.0, z1 = 0.0;
        if (r < n && c < n) {
            const double v = V[r * P + order[c]];
            if (zmode == 0) {
                const double w = dsh[order[c]] * unsc;
                const double sv = w > 0.0 ? sqrt(w) : 0.0;
                z0 = v * sv; z1 = sv > 0.0 ? v / sv : 0.0;
            } else { z0 = v; z1 = v; }
        }
        Z[e0] = z0; Z[(size_t)L * k + e0] = z1;
    }
    if (Vout)
        for (int e0 = tid; e0 < n * n; e0 += NTHR) { const int r = e0 / n, c = e0 - r * n; Vout[e0] = V[r * P + order[c]]; }
}

static size_t small_eigh_lds(int n) { return sizeof(double) * ((size_t)n * (n | 1) + 512) + sizeof(int) * 128; }
int init_device_kernels_eig() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds(128));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds(64));
    return (int)e;
}
void launch_small_eigh(hipStream_t st, const double* src, int nslices, int n, int L, int k, int zmode, double denom, const int* cholflag,
                       double* Z, double* res, double* Vout) {
    const size_t lds = small_eigh_lds(n);
    if (n <= 64) hipLaunchKernelGGL(k_small_eigh<64>, dim3(1), dim3(256), lds, st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else hipLaunchKernelGGL(k_small_eigh<128>, dim3(1), dim3(256), lds, st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
}

}  // namespace gpca
