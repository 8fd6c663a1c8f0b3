// The small dense step of the randomized PCA on the device ("one-sided Jacobi in a single workgroup, f64"): the
// symmetric eigenproblem of the l x l Gram of the projection B = A Q, and everything that used to hang on its result on the host -- the
// descending sort, singular values, eigenvalues, the two l x k factors that turn Q into scores and B into loadings.  One workgroup; the
// call's stream never waits for the host (main.rs:648-660 is one opaque call in the reference too).
//
// Algorithm: cyclic two-sided Jacobi in the round-robin (Brent-Luk) order, arranged so that NO rotation needs another thread's data:
//   * the matrix (padded with zero rows / columns to L = 32, 64 or 128) is cut into 2 x 2 blocks and a thread OWNS block (I, J) of A and
//     of the eigenvector matrix V in registers; a step rotates the index pairs (2I, 2I + 1) for all I at once, so the row rotation of a
//     block needs (c, s) of pair I, its column rotation (c, s) of pair J, and both act inside the block;
//   * the 2 x 2 diagonal block of a pair holds a_pp, a_pq, a_qq: every thread reads the diagonal blocks of ITS row pair and ITS column pair
//     from the exchange buffer and computes both rotations itself (the same numbers in every thread of a block row / column; 1 / sqrt and
//     1 / x from the hardware estimates + a cubic / one Newton step, branch-free so that the two chains interleave): no (c, s) to post,
//     no barrier for them;
//   * the next pairing is made adjacent again by moving every element to its new place THROUGH LDS (rows and columns of A, columns of
//     V, one fixed permutation every step; every other pair of rows is skewed by one double so that the scattered 8-byte stores of a
//     wave spread over all bank slots), after which every thread reads its new block back.  The exchange buffers alternate, so ONE
//     barrier per step is enough (L = 128 has room for one buffer: four barriers per step); L - 1 steps per sweep, ~8 sweeps; a sweep
//     in which no pair was above 2e-15 before its rotation was the last one;
//   * padded indices never mix with real ones (their off-diagonal entries are exactly zero), so their unit eigenvectors are told from
//     genuine zero eigenvalues by where their columns live.
// The input is scaled by a power of two (exact) so that its largest entry is in [0.5, 1): squares can neither overflow nor underflow, and
// "negligible" is an absolute 1e-17.  Every thread reaches every exit: NaN compares false (no rotation, the sweep count ends at once),
// and the sweep count is capped.
//
// Why not QL: the first two forms of this file were the EISPACK tred2 / tql2 pair the host pin (gpca_host_eigh_desc) runs, on one or two
// waves.  A lone wave issues one instruction every ~5 cycles and waits 8 for a dependent f64 result, 20 for v_rsq_f64, ~75 for an LDS
// round trip (scripts/kbench/probe_lat.hip): the 923 serial plane rotations of n = 30 cost ~370 cycles each however they were arranged --
// 250 us a call against 126 us for the host round trip they replace, and 7 ms at n = 128 (profiles/r5_kbench_summary.md).  Jacobi spends
// more flops and keeps four to sixteen waves busy: 138 us at n = 30 (248 steps of ~1 200 cycles: two rotation chains ~550, the LDS
// exchange and its barrier the rest), 0.9 ms at n = 64; at n = 128 the exchange is LDS-bandwidth bound (8.5 ms: open).
#include "kernels.h"

#ifndef GPCA_EIG_ABL
#define GPCA_EIG_ABL 0          // harness only (wrong results, exactly 8 sweeps): 1 fixed rotation instead of the chains, 2 no V update / exchange,
#endif                          // 4 no exchange of A
#ifndef GPCA_EIG_STAMP
#define GPCA_EIG_STAMP 0        // scripts/kbench/kbench_eig.hip: s_memrealtime (100 MHz) at the phase boundaries into res[kEigResFlag + 2 ..]
#endif
#if GPCA_EIG_STAMP
__device__ unsigned long long g_eig_stamp[16];      // [2 s]: s_memrealtime (100 MHz), [2 s + 1]: s_memtime (shader clock) at stamp s; [10] sweeps, [11] steps
#define EIG_STAMP(SLOT) { if (threadIdx.x == 0) { g_eig_stamp[2 * (SLOT)] = __builtin_amdgcn_s_memrealtime(); g_eig_stamp[2 * (SLOT) + 1] = __builtin_amdgcn_s_memtime(); } }
#else
#define EIG_STAMP(SLOT)
#endif

namespace gpca {

#define EIGDEV __device__ __forceinline__
// 1 / sqrt(x) for normal positive x: v_rsq_f64 (3e-8 relative, measured: scripts/kbench/kbench_eig.hip) + one cubic (Halley) step:
// y (1 + h (1/2 + 3/8 h)), h = 1 - x y^2 -- five operations, four deep, full precision (the two Newton steps it replaces: eight, eight deep)
EIGDEV double eig_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double h = fma(-(x * y), y, 1.0);
    return fma(y * h, fma(0.375, h, 0.5), y);
}

// 1 / x for normal x: v_rcp_f64 (4e-8 relative) + two Newton steps
EIGDEV double eig_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}

// Off-diagonal entries of the prescaled matrix (largest entry in [0.5, 1)): at or below kJacSkip a pair is left alone; the iteration ends
// after the sweep that leaves nothing above kJacDone off the diagonal.
constexpr double kJacSkip = 1e-17, kJacDone = 2e-15;
template <int L> struct JacCfg {
    static constexpr int H = L / 2;                            // index pairs = 2 x 2 blocks per side
    static constexpr int NT = L == 32 ? 256 : 1024;            // threads
    static constexpr int NBK = H * H / NT;                     // blocks per thread: 1, 1, 4
    static constexpr int LP = L + 2;                           // LDS row pitch in doubles (room for the skew below)
    static constexpr bool DB = L <= 64;                        // exchange buffers of A and of V, each twice (one barrier per step); L = 128: one buffer in all
    static constexpr int NBUF = DB ? 4 : 1;
};
// round-robin with index 0 fixed: the slot the index at slot s moves to after a step (pairs are slots (2i, 2i + 1))
// Le = the even number of slots that take part (n rounded up to even: the padded pairs beyond it never move and never meet anybody)
EIGDEV int jac_dest(int s, int Le) { return s >= Le ? s : (s == 0 ? 0 : (s == 1 ? (Le > 2 ? 2 : 1) : ((s & 1) ? s - 2 : (s == Le - 2 ? Le - 1 : s + 2)))); }
// LDS offset (doubles) of element (r, c): every other pair of rows starts one double later, so that the scattered 8-byte stores of a
// step (64 lanes: 4 block rows x 16 block columns, column stride 2) fill all sixteen 8-byte bank slots instead of eight
template <int L> EIGDEV int jac_at(int r, int c) { return r * JacCfg<L>::LP + ((r >> 1) & 1) + c; }

// The rotation that annihilates a_pq of [[app, apq], [apq, aqq]]: J = [[c, s], [-s, c]] with tan 2t = 2 apq / (aqq - app), |t| <= pi / 4.
EIGDEV void jac_rot(double app, double a01, double a10, double aqq, double& c, double& s) {
    const double b2 = a01 + a10, dd = aqq - app;               // 2 apq of the symmetrised pair
#if GPCA_EIG_ABL & 1
    c = 0.8; s = 0.6; return;
#endif
    // By the double angle -- cos 2t = |d| / r, sin 2t = +-2 apq / r, r = sqrt(d^2 + 4 apq^2); c = sqrt((1 + cos 2t) / 2), s = sin 2t / (2 c) -- the
    // chain is two reciprocal square roots and six other operations deep (the tangent form t = 2 apq / (d + sign(d) r), c = 1 / sqrt(1 + t^2)
    // needs a reciprocal between them: twice as deep).  c^2 + s^2 = 1 to the accuracy of the second root, whatever the first one's.
    // No branch; a pair at or below kJacSkip computes with sq = 1 and is deselected at the end (NaN input: deselected too).
    const bool live = fabs(b2) > 2.0 * kJacSkip;
    const double sq = live ? fma(dd, dd, b2 * b2) : 1.0;
    const double rinv = eig_rsqrt(sq);
    const double c2 = fabs(dd) * rinv, s2 = (dd >= 0.0 ? b2 : -b2) * rinv;
    const double h = fma(0.5, c2, 0.5);                        // cos^2 t, in [0.5, 1]
    const double ci = eig_rsqrt(h);
    c = live ? h * ci : 1.0; s = live ? 0.5 * s2 * ci : 0.0;
}

// src: the Gram W [Lw][Lw] (nslices == 0) or `nslices` partial sums of it [nslices][Lw * Lw] (summed here in slice order); only the
// leading n x n block is used, symmetrised as (W + W^T) / 2.  Outputs:
//   res[kEigResSv + j]   = sqrt(max(w_j, 0)), j < n (0 beyond)          res[kEigResEig + c] = w_c / denom, c < k
//   res[kEigResW + j]    = w_j (descending)                             res[kEigResFlag] = *cholflag, res[kEigResFlag + 1] = sweep cap hit
//   Z [2][Lw][k]: zmode 0: Z0 = V_k diag(sv), Z1 = V_k diag(1 / sv) (0 where sv = 0); zmode 1: Z0 = Z1 = V_k.  Rows >= n are zero.
//   Vout (may be NULL) [n][n]: the eigenvectors in columns, sorted like w.
template <int L>
__global__ __launch_bounds__(JacCfg<L>::NT) void k_small_eigh(const double* __restrict__ src, int nslices, int n, int Lw, int k, int zmode, double denom,
                                                             const int* __restrict__ cholflag, double* __restrict__ Z, double* __restrict__ res,
                                                             double* __restrict__ Vout) {
    using C = JacCfg<L>;
    constexpr int H = C::H, NT = C::NT, NBK = C::NBK, LP = C::LP, BUF = L * LP;
    extern __shared__ double eig_sm[];                         // DB: [A0 | V0 | A1 | V1] of L x LP each; else one buffer, A's then V's
    __shared__ double wsl[128];                                // eigenvalue by slot, then sorted
    __shared__ int order[128], genuine[128];
    __shared__ double redmax[NT];
    const int tid = threadIdx.x;
#if GPCA_EIG_STAMP
    if (tid == 0) { g_eig_stamp[10] = 0; g_eig_stamp[11] = 0; }
#endif
    EIG_STAMP(0)
    const int S = nslices > 0 ? nslices : 1;
    const size_t LL = (size_t)Lw * Lw;
    double* buf0 = eig_sm;
    // fold the slices into LDS, four elements and eight slices of each in flight per thread (the sum keeps slice order); pad = 0
    for (int e0 = tid; e0 < BUF; e0 += NT) buf0[e0] = 0.0;
    __syncthreads();
    for (int e0 = tid; e0 < n * n; e0 += 4 * NT) {
        const double* p[4];
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e1 = e0 + NT * q; const int e2 = e1 < n * n ? e1 : e0; p[q] = src + (size_t)(e2 / n) * Lw + (e2 % n); }
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
        for (int q = 0; q < 4; ++q) { const int e1 = e0 + NT * q; if (e1 < n * n) buf0[jac_at<L>(e1 / n, e1 % n)] = acc[q]; }
    }
    __syncthreads();
    // symmetrise in place (the upper-triangle thread of a pair writes both), largest finite magnitude for the prescale
    double amax = 0.0;
    for (int e0 = tid; e0 < n * n; e0 += NT) {
        const int r = e0 / n, c = e0 - r * n;
        if (r <= c) {
            const double x = 0.5 * (buf0[jac_at<L>(r, c)] + buf0[jac_at<L>(c, r)]);
            buf0[jac_at<L>(r, c)] = x; buf0[jac_at<L>(c, r)] = x;
            const double ax = fabs(x);
            amax = (ax > amax && ax < INFINITY) ? ax : amax;
        }
    }
    redmax[tid] = amax;
    __syncthreads();
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) redmax[tid] = fmax(redmax[tid], redmax[tid + st]); __syncthreads(); }
    amax = redmax[0];
    int ex = 0;
    if (amax > 0.0) (void)frexp(amax, &ex);
    const double sc = ldexp(1.0, -ex), unsc = ldexp(1.0, ex);
    // the thread's blocks and where their elements go / come from in an exchange (the same every step)
    const int Le = (n + 1) & ~1;                               // slots in the round-robin: Le - 1 steps per sweep
    double a[NBK][4], v[NBK][4], dJ[NBK][4];                   // dJ: the 2 x 2 diagonal block of the thread's column pair
    int wA[NBK][4], wV[NBK][4], rB[NBK][2], rJ[NBK][2], srcI[NBK];     // srcI: byte address (ds_bpermute) of the lane of this wave whose column pair is the thread's row pair
#pragma unroll
    for (int q = 0; q < NBK; ++q) {
        const int bk = tid + NT * q, I = bk / H, J = bk - I * H;
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) {
            const int r = 2 * I + (e1 >> 1), c = 2 * J + (e1 & 1);
            wA[q][e1] = jac_at<L>(jac_dest(r, Le), jac_dest(c, Le)); wV[q][e1] = jac_at<L>(r, jac_dest(c, Le));
            v[q][e1] = r == c ? 1.0 : 0.0;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) { rB[q][h] = jac_at<L>(2 * I + h, 2 * J); rJ[q][h] = jac_at<L>(2 * J + h, 2 * J); }
        srcI[q] = 4 * I;                                       // (a wave holds every column pair: lanes 0 .. H - 1 have J = lane)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g) { a[q][2 * h + g] = buf0[rB[q][h] + g] * sc; dJ[q][2 * h + g] = buf0[rJ[q][h] + g] * sc; }
    }
    __syncthreads();                                           // (everybody has read buf0: it is the first exchange buffer)
    EIG_STAMP(1)
    // ---- sweeps ----
    int capped = 1, sweeps = 0, cur = 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        for (int step = 0; step < Le - 1; ++step) {
            // (1) the rotation of the thread's COLUMN pair from that pair's diagonal block (every thread of a block column computes the same
            //     numbers); the rotation of its ROW pair I is the one the lane with column pair I of this wave has just computed: an
            //     LDS-crossbar read (ds_bpermute), no memory, no barrier.  A <- R_I^T A R_J, V <- V R_J
#pragma unroll
            for (int q = 0; q < NBK; ++q) {
                double cJ, sJ;
                jac_rot(dJ[q][0], dJ[q][1], dJ[q][2], dJ[q][3], cJ, sJ);
                const double cI = __hiloint2double(__builtin_amdgcn_ds_bpermute(srcI[q], __double2hiint(cJ)), __builtin_amdgcn_ds_bpermute(srcI[q], __double2loint(cJ)));
                const double sI = __hiloint2double(__builtin_amdgcn_ds_bpermute(srcI[q], __double2hiint(sJ)), __builtin_amdgcn_ds_bpermute(srcI[q], __double2loint(sJ)));
                const double b00 = fma(cI, a[q][0], -(sI * a[q][2])), b01 = fma(cI, a[q][1], -(sI * a[q][3]));
                const double b10 = fma(sI, a[q][0], cI * a[q][2]), b11 = fma(sI, a[q][1], cI * a[q][3]);
                a[q][0] = fma(cJ, b00, -(sJ * b01)); a[q][1] = fma(sJ, b00, cJ * b01);
                a[q][2] = fma(cJ, b10, -(sJ * b11)); a[q][3] = fma(sJ, b10, cJ * b11);
#if !(GPCA_EIG_ABL & 2)
                const double v00 = v[q][0], v01 = v[q][1], v10 = v[q][2], v11 = v[q][3];
                v[q][0] = fma(cJ, v00, -(sJ * v01)); v[q][1] = fma(sJ, v00, cJ * v01);
                v[q][2] = fma(cJ, v10, -(sJ * v11)); v[q][3] = fma(sJ, v10, cJ * v11);
#endif
            }
            // (2) the next pairing: every element to its new place through LDS, then the thread's new blocks (and its pairs' diagonal blocks) back
            double* bA = eig_sm + (C::DB ? 2 * cur * BUF : 0);
            double* bV = C::DB ? bA + BUF : bA;
#pragma unroll
            for (int q = 0; q < NBK; ++q)
#pragma unroll
                for (int e1 = 0; e1 < 4; ++e1) {
                    if (!(GPCA_EIG_ABL & 4)) bA[wA[q][e1]] = a[q][e1];
                    if (C::DB && !(GPCA_EIG_ABL & 2)) bV[wV[q][e1]] = v[q][e1];
                }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NBK; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        if (!(GPCA_EIG_ABL & 4)) { dJ[q][2 * h + g] = bA[rJ[q][h] + g]; a[q][2 * h + g] = bA[rB[q][h] + g]; }
                        if (C::DB && !(GPCA_EIG_ABL & 2)) v[q][2 * h + g] = bV[rB[q][h] + g];
                    }
            if (C::DB) cur ^= 1;                               // (the next step writes the other pair of buffers: a wave one barrier ahead cannot overwrite what a slower one still reads)
            else {                                             // L = 128: the same buffer once more for V
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NBK; ++q)
#pragma unroll
                    for (int e1 = 0; e1 < 4; ++e1) bV[wV[q][e1]] = v[q][e1];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NBK; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int g = 0; g < 2; ++g) v[q][2 * h + g] = bV[rB[q][h] + g];
                __syncthreads();
            }
        }
        ++sweeps;
        // Done when nothing off the diagonal is above kJacDone any more: every thread looks at its own elements (a diagonal block's own
        // diagonal excepted), one maximum over the workgroup.  (A flag "some pair of this sweep was large before its rotation" needs one more
        // sweep to come back empty: 31 steps that rotate nothing.)
        double om = 0.0;
#pragma unroll
        for (int q = 0; q < NBK; ++q) {
            const int bk = tid + NT * q; const bool dg = (bk / H) == (bk % H);
            om = fmax(om, fmax(fabs(a[q][1]), fabs(a[q][2])));
            if (!dg) om = fmax(om, fmax(fabs(a[q][0]), fabs(a[q][3])));
        }
        redmax[tid] = om;                                      // (NaN: fmax drops it -- NaN input ends after the first sweep)
        __syncthreads();
        for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) redmax[tid] = fmax(redmax[tid], redmax[tid + st]); __syncthreads(); }
        const double offmax = redmax[0];
        __syncthreads();
        if (GPCA_EIG_ABL ? sweep == 7 : !(offmax > kJacDone)) { capped = 0; break; }
    }
#if GPCA_EIG_STAMP
    if (tid == 0) { g_eig_stamp[10] = sweeps; g_eig_stamp[11] = (unsigned long long)sweeps * (Le > 1 ? Le - 1 : 0); }
#endif
    (void)sweeps;
    EIG_STAMP(3)
    // ---- results: eigenvalue of slot s = the diagonal, its vector = column s of V (rows in the original order) ----
    __syncthreads();
    double* bufV = eig_sm;                                     // [L][LP], plain layout from here on
#pragma unroll
    for (int q = 0; q < NBK; ++q) {
        const int bk = tid + NT * q, I = bk / H, J = bk - I * H;
        if (I == J) { wsl[2 * I] = a[q][0]; wsl[2 * I + 1] = a[q][3]; }
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) bufV[(2 * I + (e1 >> 1)) * LP + 2 * J + (e1 & 1)] = v[q][e1];
    }
    if (tid < 128) order[tid] = 0;
    __syncthreads();
    // a padded index keeps a unit vector outside the first n rows: slot t is genuine when its column has weight inside them
    if (tid < L) { double ss = 0.0; for (int r = 0; r < n; ++r) { const double x = bufV[r * LP + tid]; ss = fma(x, x, ss); } genuine[tid] = ss > 0.25; }
    __syncthreads();
    // descending order over the genuine slots: rank_s = #{t genuine : w_t > w_s or (w_t == w_s and t < s)}  (= the host's stable selection sort)
    double wmine = 0.0; int rank = -1;
    if (tid < L && genuine[tid]) {
        wmine = wsl[tid]; rank = 0;
        for (int t = 0; t < L; ++t) { const double wt = wsl[t]; if (t != tid && genuine[t] && (wt > wmine || (wt == wmine && t < tid))) ++rank; }
    }
    __syncthreads();
    if (rank >= 0 && rank < n) { order[rank] = tid; wsl[rank] = wmine; }       // (wsl re-used: every thread has read what it needs)
    __syncthreads();
    for (int j = tid; j < kMaxSketchCols; j += NT) {
        const double w = j < n ? wsl[j] * unsc : 0.0;
        res[kEigResW + j] = w;
        res[kEigResSv + j] = w > 0.0 ? sqrt(w) : 0.0;
        res[kEigResEig + j] = j < k ? w / denom : 0.0;
    }
    if (tid == 0) { res[kEigResFlag] = cholflag ? (double)cholflag[0] : 0.0; res[kEigResFlag + 1] = (double)capped; }
    for (int e0 = tid; e0 < Lw * k; e0 += NT) {
        const int r = e0 / k, c = e0 - r * k;
        double z0 = 0.0, z1 = 0.0;
        if (r < n && c < n) {
            const double x = bufV[r * LP + order[c]];
            if (zmode == 0) {
                const double w = wsl[c] * unsc;
                const double sv = w > 0.0 ? sqrt(w) : 0.0;
                z0 = x * sv; z1 = sv > 0.0 ? x / sv : 0.0;
            } else { z0 = x; z1 = x; }
        }
        Z[e0] = z0; Z[(size_t)Lw * k + e0] = z1;
    }
    EIG_STAMP(4)
    if (Vout)
        for (int e0 = tid; e0 < n * n; e0 += NT) { const int r = e0 / n, c = e0 - r * n; Vout[e0] = bufV[r * LP + order[c]]; }
}

template <int L> static size_t small_eigh_lds() { return sizeof(double) * (size_t)L * JacCfg<L>::LP * JacCfg<L>::NBUF; }
int init_device_kernels_eig() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds<128>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds<64>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds<32>());
    return (int)e;
}
void launch_small_eigh(hipStream_t st, const double* src, int nslices, int n, int L, int k, int zmode, double denom, const int* cholflag,
                       double* Z, double* res, double* Vout) {
    if (n <= 32) hipLaunchKernelGGL(k_small_eigh<32>, dim3(1), dim3(JacCfg<32>::NT), small_eigh_lds<32>(), st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else if (n <= 64) hipLaunchKernelGGL(k_small_eigh<64>, dim3(1), dim3(JacCfg<64>::NT), small_eigh_lds<64>(), st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else hipLaunchKernelGGL(k_small_eigh<128>, dim3(1), dim3(JacCfg<128>::NT), small_eigh_lds<128>(), st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
}

}  // namespace gpca
