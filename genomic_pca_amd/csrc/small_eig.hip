// The small dense step of the randomized PCA on the device: the symmetric eigenproblem of the l x l Gram of the
// projection B = A Q, and everything that used to hang on its result on the host -- the descending sort, singular values, eigenvalues,
// the two l x k factors that turn Q into scores and B into loadings.  One workgroup; the call's stream never waits for the host
// (main.rs:648-660 is one opaque call in the reference too).
//
// Algorithm: Householder tridiagonalisation + implicit QL (the EISPACK tred2 / tql2 pair, the same pair gpca_host_eigh_desc runs on
// the CPU as the pin of tests/test_abi.py), arranged for ONE wave (lane j owns rows / columns j and, for 64 < n <= 128, j + 64):
//   * the matrix lives in LDS with an odd row pitch (a column walk across lanes and a row walk across lanes are both conflict-free);
//     the diagonal and sub-diagonal live in registers, entry i in lane i, read with v_readlane (i is wave-uniform) and written
//     with a compare-and-select: the serial parts of both phases touch no LDS and carry no predicated region;
//   * tred2: both triangles are kept current, so the matrix-vector product and the rank-2 update of a step walk the lane's own row;
//     the step's two scalar sums are DPP wave reductions;
//   * tql2: lane t owns ROW t of the eigenvector matrix, so a plane rotation touches only the lane's own two elements.  The
//     rotation parameters are a serial chain every lane computes redundantly (1 / sqrt by the hardware estimate + one cubic step
//     instead of hypot + two divisions); the next rotation's d[i], e[i] and matrix element are fetched one rotation ahead;
//   * the wave reaches every exit: the QL loop is bounded (200 sweeps per eigenvalue, as on the host), NaN input compares false
//     in the deflation test and falls through.
// First form (round 5, two waves, d / e in LDS, lane-0 writes): 353 us at n = 30, 6 ms at n = 128, at a shader clock of 2.40 GHz --
// every rotation waited for an LDS write to land (profiles/r5_kbench_summary.md).
// The input is scaled by a power of two (exact) so that its largest entry is in [0.5, 1): the sums of squares of the rotation chain
// can neither overflow nor underflow for any finite Gram matrix.
#include "kernels.h"

#ifndef GPCA_EIG_ABL
#define GPCA_EIG_ABL 0          // harness only (wrong results): 1 no d / e write-back in a rotation, 2 no matrix traffic, 4 no operand prefetch, 8 raw v_rsq
#endif
#ifndef GPCA_EIG_STAMP
#define GPCA_EIG_STAMP 0        // scripts/kbench/kbench_eig.hip: s_memrealtime (100 MHz) at the phase boundaries into res[kEigResFlag + 2 ..]
#endif
#if GPCA_EIG_STAMP
__device__ unsigned long long g_eig_stamp[16];      // [2 s]: s_memrealtime (100 MHz), [2 s + 1]: s_memtime (shader clock) at stamp s
#define EIG_STAMP(SLOT) { if (threadIdx.x == 0) { g_eig_stamp[2 * (SLOT)] = __builtin_amdgcn_s_memrealtime(); g_eig_stamp[2 * (SLOT) + 1] = __builtin_amdgcn_s_memtime(); } }
#else
#define EIG_STAMP(SLOT)
#endif
#if GPCA_EIG_STAMP
#define EIG_COUNT(SLOT, N) { if (threadIdx.x == 0) g_eig_stamp[10 + (SLOT)] += (N); }
#else
#define EIG_COUNT(SLOT, N)
#endif

namespace gpca {

#define EIGDEV __device__ __forceinline__
EIGDEV void eig_wsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// value of lane `i` (wave-uniform index): two v_readlane into SGPRs
EIGDEV double eig_rl(double v, int i) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), i), __builtin_amdgcn_readlane(__double2loint(v), i));
}
// A vector of up to 64 * NB entries lives in registers, entry i in lane i % 64 of r[i / 64].
template <int NB> EIGDEV double eig_get(const double (&r)[NB], int i) {
    if (NB == 1) return eig_rl(r[0], i);
    return i < 64 ? eig_rl(r[0], i) : eig_rl(r[NB - 1], i - 64);
}
template <int NB> EIGDEV void eig_set(double (&r)[NB], int i, double v, int lane) {
#pragma unroll
    for (int b = 0; b < NB; ++b) r[b] = (lane + 64 * b == i) ? v : r[b];
}
template <int CTRL, int ROWMASK> EIGDEV double eig_dpp0(double v) {     // the DPP-selected lane's value, 0 where the selection leaves the row / the row is masked
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, returned in every lane: inclusive scan inside the rows of 16 (row_shr 1, 2, 4, 8), row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3, lane 63 read back -- ~20 VALU operations instead of six LDS crossbar round trips
EIGDEV double eig_wave_sum(double x) {
    x += eig_dpp0<0x111, 0xf>(x); x += eig_dpp0<0x112, 0xf>(x); x += eig_dpp0<0x114, 0xf>(x); x += eig_dpp0<0x118, 0xf>(x);
    x += eig_dpp0<0x142, 0xa>(x); x += eig_dpp0<0x143, 0xc>(x);
    return eig_rl(x, 63);
}
// 1 / sqrt(x) for normal positive x: v_rsq_f64 (3e-8 relative, measured: scripts/kbench/kbench_eig.hip) + one cubic (Halley) step:
// y (1 + h (1/2 + 3/8 h)), h = 1 - x y^2 -- five operations, four deep, full precision (the two Newton steps it replaces: eight, eight deep)
EIGDEV double eig_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double h = fma(-(x * y), y, 1.0);
    return fma(y * h, fma(0.375, h, 0.5), y);
}

// Householder reduction to tridiagonal form, ONE wave: lane j owns rows / columns j + 64 b of V (n x n, pitch P, symmetric on entry, both
// triangles kept current so that every inner product and update walks the lane's own row).  On return V is the accumulated transformation,
// td = diagonal, e = sub-diagonal (e[0] = 0), both distributed over the lanes.  The input is prescaled (largest entry < 1): JAMA's
// per-step rescaling is not needed.
template <int NB>
EIGDEV void eig_tred2(int n, int P, double* __restrict__ V, double (&td)[NB], double (&e)[NB], int lane) {
    double hreg[NB];
    int row[NB];                                               // the lane's rows, clamped into the matrix (lanes beyond n repeat row n - 1: same values, same writes)
#pragma unroll
    for (int b = 0; b < NB; ++b) { hreg[b] = 0.0; e[b] = 0.0; td[b] = 0.0; row[b] = (lane + 64 * b < n) ? lane + 64 * b : n - 1; }
    for (int i = n - 1; i > 0; --i) {
        double u[NB], q[NB], p[NB];
        double hs = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) { u[b] = (lane + 64 * b < i) ? V[i * P + row[b]] : 0.0; hs = fma(u[b], u[b], hs); }
        double h = eig_wave_sum(hs);
        const double f = eig_get<NB>(u, i - 1);
        if (h == 0.0) {                                        // nothing to annihilate (or below 1e-162 of the largest entry)
            eig_set<NB>(e, i, f, lane);
        } else {
            double g = sqrt(h);
            if (f > 0) g = -g;
            eig_set<NB>(e, i, g, lane);
            h -= f * g;
            eig_set<NB>(u, i - 1, f - g, lane);
#pragma unroll
            for (int b = 0; b < NB; ++b) { if (lane + 64 * b < i) V[row[b] * P + i] = u[b]; p[b] = 0.0; }      // the Householder vector stays in column i
            // p = A u over the leading i x i block, the lane's own rows.  Eight columns per trip, written out (v_readlane is a convergent
            // operation: the compiler will not unroll a runtime loop around it, and one LDS round trip per column is the whole cost);
            // columns past i - 1 are clamped into the matrix and weighted 0
            for (int k0 = 0; k0 < i; k0 += 8) {
                double av[NB][8], uk[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = k0 + t < n ? k0 + t : n - 1;
#pragma unroll
                    for (int b = 0; b < NB; ++b) av[b][t] = V[row[b] * P + kk];
                    const double x = eig_get<NB>(u, kk);
                    uk[t] = k0 + t < i ? x : 0.0;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int b = 0; b < NB; ++b) p[b] = fma(av[b][t], uk[t], p[b]);
            }
            double fs = 0.0;
            const double hinv = 1.0 / h;
#pragma unroll
            for (int b = 0; b < NB; ++b) { p[b] = (lane + 64 * b < i) ? p[b] * hinv : 0.0; fs = fma(p[b], u[b], fs); }
            const double hh = eig_wave_sum(fs) / (h + h);
#pragma unroll
            for (int b = 0; b < NB; ++b) q[b] = p[b] - hh * u[b];
            for (int k0 = 0; k0 < i; k0 += 8) {               // A -= u q^T + q u^T, the lane's own rows (u = q = 0 in the lanes past i - 1 and for the clamped columns)
                double av[NB][8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = k0 + t < n ? k0 + t : n - 1;
#pragma unroll
                    for (int b = 0; b < NB; ++b) av[b][t] = V[row[b] * P + kk];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = k0 + t < n ? k0 + t : n - 1;
                    const double x = eig_get<NB>(u, kk), y = eig_get<NB>(q, kk);
                    const double uk = k0 + t < i ? x : 0.0, qk = k0 + t < i ? y : 0.0;
                    if (k0 + t < i) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) V[row[b] * P + kk] = av[b][t] - (u[b] * qk + q[b] * uk);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) if (lane + 64 * b < i) V[i * P + row[b]] = 0.0;
        eig_set<NB>(hreg, i, h, lane);
        eig_wsync();
    }
    for (int i = 0; i < n - 1; ++i) {                          // accumulate the transformations
#pragma unroll
        for (int b = 0; b < NB; ++b) if (lane + 64 * b == i) { td[b] = V[i * P + i]; V[i * P + i] = 1.0; }
        eig_wsync();
        const double h = eig_get<NB>(hreg, i + 1);
        if (h != 0.0) {
            double uk[NB], wk[NB], g[NB];
            const double hinv = 1.0 / h;
#pragma unroll
            for (int b = 0; b < NB; ++b) { uk[b] = (lane + 64 * b <= i) ? V[row[b] * P + (i + 1)] : 0.0; wk[b] = uk[b] * hinv; g[b] = 0.0; }
            for (int k0 = 0; k0 <= i; k0 += 8) {               // g = u^T (the lane's own column), rows 0 .. i
                double av[NB][8], ut[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = k0 + t < n ? k0 + t : n - 1;
#pragma unroll
                    for (int b = 0; b < NB; ++b) av[b][t] = V[kk * P + row[b]];
                    const double x = eig_get<NB>(uk, kk);
                    ut[t] = k0 + t <= i ? x : 0.0;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int b = 0; b < NB; ++b) g[b] = fma(ut[t], av[b][t], g[b]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) g[b] = (lane + 64 * b <= i) ? g[b] : 0.0;     // (the columns past i stay as they are)
            for (int k0 = 0; k0 <= i; k0 += 8) {
                double av[NB][8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = k0 + t < n ? k0 + t : n - 1;
#pragma unroll
                    for (int b = 0; b < NB; ++b) av[b][t] = V[kk * P + row[b]];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = k0 + t < n ? k0 + t : n - 1;
                    const double wkk = eig_get<NB>(wk, kk);
                    if (k0 + t <= i) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) V[kk * P + row[b]] = av[b][t] - g[b] * wkk;
                    }
                }
            }
        }
        eig_wsync();
#pragma unroll
        for (int b = 0; b < NB; ++b) if (lane + 64 * b <= i) V[row[b] * P + (i + 1)] = 0.0;
        eig_wsync();
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) if (lane + 64 * b == n - 1) { td[b] = V[(n - 1) * P + (n - 1)]; V[(n - 1) * P + (n - 1)] = 1.0; }
    eig_wsync();
}

// Implicit QL on the tridiagonal (d, e distributed over the lanes as above); lane t rotates rows t + 64 b of V.  Returns 1 if a sweep count
// hit the cap.  Inside a sweep nothing is predicated and nothing waits for LDS but the lane's own row: the rotation's operands d[i], e[i]
// come by v_readlane one rotation ahead, its results go back by a compare-and-select, the history the sweep's last step needs (the sine
// before the last rotation, the cosine before the last two) is taken by peeling those two rotations off the loop.
template <int NB>
EIGDEV int eig_tql2(int n, int P, double* __restrict__ V, double (&d)[NB], double (&e)[NB], int lane) {
    const double eps = 2.220446049250313e-16;
    int capped = 0;
    int row[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) row[b] = (lane + 64 * b < n) ? lane + 64 * b : n - 1;
    {   // e[i - 1] = e[i], e[n - 1] = 0
        double en[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double nx = __shfl_down(e[b], 1);
            if (lane == 63) nx = (b + 1 < NB) ? eig_rl(e[NB - 1], 0) : 0.0;
            en[b] = (lane + 64 * b + 1 < n) ? nx : 0.0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) e[b] = en[b];
    }
    double f = 0.0, tst1 = 0.0;
    for (int l = 0; l < n; ++l) {
        tst1 = fmax(tst1, fabs(eig_get<NB>(d, l)) + fabs(eig_get<NB>(e, l)));
        // an off-diagonal entry at or below thr is zero: eps * tst1 as on the host, and never below 1e-140 of the (prescaled) matrix, so
        // that p^2 + e^2 of a rotation stays a normal number
        const double thr = fmax(eps * tst1, 1e-140);
        int m = n - 1;                                         // first index >= l whose e is negligible (e[n - 1] = 0)
#pragma unroll
        for (int b = NB - 1; b >= 0; --b) {
            const int idx = lane + 64 * b;
            const unsigned long long mk = __ballot(idx >= l && idx < n && !(fabs(e[b]) > thr));
            if (mk) m = 64 * b + __ffsll((long long)mk) - 1;
        }
        if (m > l) {
            int iter = 0;
            double el_cur;
            do {
                ++iter;
                const double g0 = eig_get<NB>(d, l), el0 = eig_get<NB>(e, l);
                double p = (eig_get<NB>(d, l + 1) - g0) / (2.0 * el0);
                double r = fabs(p) < 1e150 ? sqrt(fma(p, p, 1.0)) : fabs(p);
                if (p < 0) r = -r;
                const double dl_new = el0 / (p + r), dl1 = el0 * (p + r);
                const double hsh = g0 - dl_new;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int idx = lane + 64 * b;
                    d[b] = idx == l ? dl_new : (idx == l + 1 ? dl1 : ((idx > l + 1 && idx < n) ? d[b] - hsh : d[b]));
                }
                f += hsh;
                p = eig_get<NB>(d, m);
                double c = 1.0, s = 0.0, c3 = 1.0, s2 = 0.0;
                const double el1 = eig_get<NB>(e, l + 1);
                double xhi[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) xhi[b] = V[row[b] * P + m];
                double di = eig_get<NB>(d, m - 1), ei = eig_get<NB>(e, m - 1);
                // A rotation = a serial chain (p, e[i] -> 1 / sqrt -> c, s -> the next p: ten dependent operations, the wave has nothing
                // else to issue meanwhile) and a tail nothing waits for (the new d[i + 1], e[i + 1], the two matrix elements).  The loop body
                // is the chain of rotation i beside the TAIL of rotation i + 1, so the scheduler has independent work for the chain's bubbles.
                double t_c = 1.0, t_s = 0.0, t_sp = 0.0, t_g = 0.0, t_h = 0.0, t_di = 0.0, t_rr = 0.0;     // the pending tail's inputs
                int t_i = -1;
                double xlo_p[NB];
                auto tail = [&]() {                            // rotation t_i's results: e[t_i + 1], d[t_i + 1], V[:, t_i + 1], the running V[:, t_i]
                    const double e_up = t_sp * t_rr;
                    const double d_up = t_h + t_s * fma(t_c, t_g, t_s * t_di);
                    if (!(GPCA_EIG_ABL & 1)) { eig_set<NB>(e, t_i + 1, e_up, lane); eig_set<NB>(d, t_i + 1, d_up, lane); }
                    else { t_rr += e_up + d_up; }
#pragma unroll
                    for (int b = 0; b < NB; ++b) { if (!(GPCA_EIG_ABL & 2)) V[row[b] * P + (t_i + 1)] = fma(t_s, xlo_p[b], t_c * xhi[b]); xhi[b] = fma(t_c, xlo_p[b], -(t_s * xhi[b])); }
                };
                auto rotate = [&](int i) {
                    const int ip = i > l ? i - 1 : l;          // the next rotation's operands, one rotation ahead
                    const double dn = (GPCA_EIG_ABL & 4) ? di * 0.999 : eig_get<NB>(d, ip), en = (GPCA_EIG_ABL & 4) ? ei * 0.999 : eig_get<NB>(e, ip);
                    double xlo[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) xlo[b] = (GPCA_EIG_ABL & 2) ? xhi[b] * 0.5 : V[row[b] * P + i];
                    // chain
                    const double g = c * ei, h = c * p;
                    const double sq = fma(p, p, ei * ei);
                    const double rinv = (GPCA_EIG_ABL & 8) ? __builtin_amdgcn_rsq(sq) : eig_rsqrt(sq);
                    const double s_prev = s;
                    s = ei * rinv; c = p * rinv;
                    p = fma(c, di, -(s * g));
                    // the previous rotation's tail (independent of everything above)
                    if (t_i >= 0) tail();
                    t_i = i; t_c = c; t_s = s; t_sp = s_prev; t_g = g; t_h = h; t_di = di; t_rr = sq * rinv;
#pragma unroll
                    for (int b = 0; b < NB; ++b) xlo_p[b] = xlo[b];
                    di = dn; ei = en;
                };
                int i = m - 1;
                EIG_COUNT(0, 1) EIG_COUNT(1, m - l)
                for (; i >= l + 2; --i) rotate(i);
                if (i == l + 1) { c3 = c; rotate(l + 1); }
                s2 = s;
                rotate(l);
                tail();
#pragma unroll
                for (int b = 0; b < NB; ++b) V[row[b] * P + l] = xhi[b];
                p = -s * s2 * c3 * el1 * el0 / dl1;
                el_cur = s * p;
                eig_set<NB>(e, l, el_cur, lane);
                eig_set<NB>(d, l, c * p, lane);
            } while (fabs(el_cur) > thr && iter < 200);
            if (fabs(el_cur) > thr) capped = 1;
        }
        eig_set<NB>(d, l, eig_get<NB>(d, l) + f, lane);
        eig_set<NB>(e, l, 0.0, lane);
    }
    return capped;
}

// src: the Gram W [L][L] (nslices == 0) or `nslices` partial sums of it [nslices][L * L] (summed here in slice order); only the
// leading n x n block is used, symmetrised as (W + W^T) / 2.  Outputs:
//   res[kEigResSv + j]   = sqrt(max(w_j, 0)), j < n (0 beyond)          res[kEigResEig + c] = w_c / denom, c < k
//   res[kEigResW + j]    = w_j (descending)                             res[kEigResFlag] = *cholflag, res[kEigResFlag + 1] = sweep cap hit
//   Z [2][L][k]: zmode 0: Z0 = V_k diag(sv), Z1 = V_k diag(1 / sv) (0 where sv = 0); zmode 1: Z0 = Z1 = V_k.  Rows >= n are zero.
//   Vout (may be NULL) [n][n]: the eigenvectors in columns, sorted like w.
template <int NB>      // rows per lane of the one-wave solver: n <= 64 NB
__global__ __launch_bounds__(256) void k_small_eigh(const double* __restrict__ src, int nslices, int n, int L, int k, int zmode, double denom,
                                                    const int* __restrict__ cholflag, double* __restrict__ Z, double* __restrict__ res,
                                                    double* __restrict__ Vout) {
    extern __shared__ double eig_sm[];
    const int P = n | 1;
    double* V = eig_sm;                                        // [n][P]
    double* wsh = V + (size_t)n * P;                           // [128]: sorted eigenvalues (scaled)
    int* order = reinterpret_cast<int*>(wsh + 128);            // [128]: column of V that holds eigenvector c
    __shared__ double redmax[256];
    const int tid = threadIdx.x;
#if GPCA_EIG_STAMP
    if (tid == 0) { g_eig_stamp[10] = 0; g_eig_stamp[11] = 0; }
#endif
    EIG_STAMP(0)
    const int S = nslices > 0 ? nslices : 1;
    const size_t LL = (size_t)L * L;
    // fold the slices into LDS, all 256 threads, four elements and eight slices of each in flight per thread (the sum keeps slice order)
    for (int e0 = tid; e0 < n * n; e0 += 1024) {
        const double* p[4];
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e1 = e0 + 256 * q; const int e2 = e1 < n * n ? e1 : e0; p[q] = src + (size_t)(e2 / n) * L + (e2 % n); }
        for (int s0 = 0; s0 < S; s0 += 8) {
            double v[4][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sl = s0 + u < S ? s0 + u : s0;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q][u] = p[q][sl * LL];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += (s0 + u < S) ? v[q][u] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e1 = e0 + 256 * q; if (e1 < n * n) V[(e1 / n) * P + (e1 % n)] = acc[q]; }
    }
    __syncthreads();
    // symmetrise (W + W^T) / 2 in place, largest finite magnitude for the prescale
    double amax = 0.0;
    for (int e0 = tid; e0 < n * n; e0 += 256) {
        const int a = e0 / n, c = e0 - a * n;
        if (a <= c) {
            const double v = 0.5 * (V[a * P + c] + V[c * P + a]);
            V[a * P + c] = v; V[c * P + a] = v;
            const double av = fabs(v);
            amax = (av > amax && av < INFINITY) ? av : amax;
        }
    }
    redmax[tid] = amax;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) redmax[tid] = fmax(redmax[tid], redmax[tid + st]); __syncthreads(); }
    amax = redmax[0];
    int ex = 0;
    if (amax > 0.0) (void)frexp(amax, &ex);
    const double sc = ldexp(1.0, -ex), unsc = ldexp(1.0, ex);
    for (int e0 = tid; e0 < n * n; e0 += 256) { const int a = e0 / n, c = e0 - a * n; V[a * P + c] *= sc; }
    __syncthreads();
    if (tid >= 64) return;                                     // one wave from here on: no block barrier below
    const int lane = tid;
    EIG_STAMP(1)
    double d[NB], e[NB];
    eig_tred2<NB>(n, P, V, d, e, lane);
    EIG_STAMP(2)
    const int capped = eig_tql2<NB>(n, P, V, d, e, lane);
    EIG_STAMP(3)
    // descending order: rank_j = #{i : w_i > w_j or (w_i == w_j and i < j)}  (= the host's stable selection sort)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        int rank = 0;
        for (int i = 0; i < n; ++i) { const double wi = eig_get<NB>(d, i); rank += (wi > d[b] || (wi == d[b] && i < j)) ? 1 : 0; }
        if (j < n) { rank = rank < n ? rank : n - 1; order[rank] = j; wsh[rank] = d[b]; }      // (NaN input: keep the writes in range)
    }
    eig_wsync();
    for (int j = lane; j < kMaxSketchCols; j += 64) {
        const double w = j < n ? wsh[j] * unsc : 0.0;
        res[kEigResW + j] = w;
        res[kEigResSv + j] = w > 0.0 ? sqrt(w) : 0.0;
        res[kEigResEig + j] = j < k ? w / denom : 0.0;
    }
    if (lane == 0) { res[kEigResFlag] = cholflag ? (double)cholflag[0] : 0.0; res[kEigResFlag + 1] = (double)capped; }
    for (int e0 = lane; e0 < L * k; e0 += 64) {
        const int r = e0 / k, c = e0 - r * k;
        double z0 = 0.0, z1 = 0.0;
        if (r < n && c < n) {
            const double v = V[r * P + order[c]];
            if (zmode == 0) {
                const double w = wsh[c] * unsc;
                const double sv = w > 0.0 ? sqrt(w) : 0.0;
                z0 = v * sv; z1 = sv > 0.0 ? v / sv : 0.0;
            } else { z0 = v; z1 = v; }
        }
        Z[e0] = z0; Z[(size_t)L * k + e0] = z1;
    }
    EIG_STAMP(4)
    if (Vout)
        for (int e0 = lane; e0 < n * n; e0 += 64) { const int r = e0 / n, c = e0 - r * n; Vout[e0] = V[r * P + order[c]]; }
}

static size_t small_eigh_lds(int n) { return sizeof(double) * ((size_t)n * (n | 1) + 128) + sizeof(int) * 128; }
int init_device_kernels_eig() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds(128));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds(64));
    return (int)e;
}
void launch_small_eigh(hipStream_t st, const double* src, int nslices, int n, int L, int k, int zmode, double denom, const int* cholflag,
                       double* Z, double* res, double* Vout) {
    const size_t lds = small_eigh_lds(n);
    if (n <= 64) hipLaunchKernelGGL(k_small_eigh<1>, dim3(1), dim3(256), lds, st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else hipLaunchKernelGGL(k_small_eigh<2>, dim3(1), dim3(256), lds, st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
}

}  // namespace gpca
