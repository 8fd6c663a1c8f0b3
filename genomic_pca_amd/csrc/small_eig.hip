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
// more flops and keeps four to sixteen waves busy: 117 us at n = 30 (203 steps of ~1 200 cycles: the rotation chain ~550, the LDS
// exchange and its barrier the rest), 0.8 ms at n = 64; at n = 128 the exchange is LDS-bandwidth bound (6.3 ms), and sketches of
// 65-128 columns go through k_small_eigh_ql below: QL again, but with its serial chain on one wave and the eigenvector updates on
// two others (3.6 ms).
#include "kernels.h"

#ifndef GPCA_EIG_ABL
#define GPCA_EIG_ABL 0          // harness only (wrong results, exactly 8 sweeps): 1 fixed rotation instead of the chains, 2 no V update / exchange,
#endif                          // 4 no exchange of A
#ifndef GPCA_EIG_STAMP
#define GPCA_EIG_STAMP 0        // scripts/kbench/kbench_eig.hip: s_memrealtime (100 MHz) at the phase boundaries into res[kEigResFlag + 2 ..]
#endif
#if GPCA_EIG_STAMP
__device__ unsigned long long g_eig_stamp[20];      // [2 s]: s_memrealtime (100 MHz), [2 s + 1]: s_memtime (shader clock) at stamp s; [10] sweeps, [11] steps
#define EIG_STAMP(SLOT) { if (threadIdx.x == 0) { g_eig_stamp[2 * (SLOT)] = __builtin_amdgcn_s_memrealtime(); g_eig_stamp[2 * (SLOT) + 1] = __builtin_amdgcn_s_memtime(); } }
#define EIG_COUNT(X) (++(X))            // harness counters of the QL form: iterations, polls, batches
#else
#define EIG_STAMP(SLOT)
#define EIG_COUNT(X) ((void)(X))
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

// The results from eigenvalue-by-slot `wsl` (prescaled by 1 / unsc), the eigenvector of slot s = getV(row, s), and which slots are genuine
// (a padded index of the Jacobi form keeps a unit vector outside the first n rows).  Called by every thread of the workgroup; `wsl`,
// `order` are LDS arrays of 128.
template <int NT, class GetV>
EIGDEV void eig_emit(GetV getV, double* wsl, int* order, const int* genuine, int slots, int n, int Lw, int k, int zmode, double denom, double unsc,
                     const int* __restrict__ cholflag, int capped, double* __restrict__ Z, double* __restrict__ res, double* __restrict__ Vout, int tid) {
    // descending order over the genuine slots: rank_s = #{t genuine : w_t > w_s or (w_t == w_s and t < s)}  (= the host's stable selection sort)
    double wmine = 0.0; int rank = -1;
    if (tid < slots && genuine[tid]) {
        wmine = wsl[tid]; rank = 0;
        for (int t = 0; t < slots; ++t) { const double wt = wsl[t]; if (t != tid && genuine[t] && (wt > wmine || (wt == wmine && t < tid))) ++rank; }
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
            const double x = getV(r, order[c]);
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
        for (int e0 = tid; e0 < n * n; e0 += NT) { const int r = e0 / n, c = e0 - r * n; Vout[e0] = getV(r, order[c]); }
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
    eig_emit<NT>([&](int r, int slot) { return bufV[r * LP + slot]; }, wsl, order, genuine, L, n, Lw, k, zmode, denom, unsc, cholflag, capped, Z, res, Vout, tid);
}


// ================================================================================================
// Sketches of 65 .. 128 columns: Householder tridiagonalisation + implicit QL (the pair the host pin runs), arranged for one workgroup.
// The Jacobi form above moves the whole matrix through LDS every step: 0.57 / 0.79 ms at n = 50 / 64, 5.0 / 6.3 ms at n = 100 / 128
// (LDS-bandwidth bound).  This form: 0.62 / 0.95 ms at n = 50 / 64 (so Jacobi keeps L = 64), **2.2 / 3.6 ms at n = 100 / 128**
// (scripts/kbench/kbench_eig.hip, profiles/r5_kbench_summary.md section 1): 0.34 / 0.53 ms of tridiagonalisation + Q, the rest the
// serial chain at ~407 cycles per rotation (10 753 / 18 074 rotations).
//   * tridiagonalisation and the accumulation of Q: n - 2 reflectors, each a matrix-vector product and a rank-2 (rank-1) update
//     spread over all 1 024 threads, four (three) barriers per reflector; the matrix sits in LDS with row r rotated by r places
//     (ql_at), so rows and columns are both conflict-free without padding (128 x 128 doubles = 128 KiB of the 160);
//   * the QL iteration on (d, e) is a serial chain of plane rotations that does not depend on the eigenvectors: ONE wave runs it and posts every rotation (index, c, s) into a ring in LDS; one or
//     two other waves apply the rotations to the rows of Z (a lane owns a row; the element two consecutive rotations share stays
//     in a register), lagging behind by what the ring holds.  The other waves wait at the barrier.
//   * every wait is bounded: a side that polls too long sets the cap flag (-> GPCA_ERR_NOT_CONVERGED) and leaves.
// Results through eig_emit like the Jacobi form.  Same prescale (largest entry in [0.5, 1)): "negligible" is relative to a running
// |d| + |e| with a floor of 1e-20, so no square underflows.
// ================================================================================================
template <int L> struct QlCfg {
    static constexpr int NT = 1024;
    static constexpr int RING = 512;                           // rotations in flight between the chain and the waves that apply them (ql_tag assumes 512)
    static constexpr int NCONS = L / 64;                       // applying waves: one row of Z per lane
    static constexpr int kDoubles = L * L + 6 * L + 64 + 2 * RING;
    static constexpr size_t kBytes = sizeof(double) * kDoubles + sizeof(int) * (RING + 16);
};
template <int L> EIGDEV int ql_at(int r, int c) { return r * L + ((c + r) & (L - 1)); }
EIGDEV int ql_tag(int seq) { return (((seq >> 9) + 1) & 0xffff) << 12; }                  // (RING = 512 entries per lap)
// the DPP-selected lane's value (0 where the selection leaves the row or the row is masked): two VALU moves, no LDS crossbar trip
template <int CTRL, int ROWMASK> EIGDEV double eig_dpp0(double v) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
EIGDEV double eig_lane(double v, int i) {                     // lane i's value in every lane (i wave-uniform): two v_readlane
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), i), __builtin_amdgcn_readlane(__double2loint(v), i));
}
// sum over the 64 lanes, in every lane: inclusive scan inside the rows of 16 (row_shr 1, 2, 4, 8), row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3, lane 63 read back -- ~20 VALU operations instead of six LDS crossbar round trips of ~100 cycles
EIGDEV double wave_sum(double x) {
    x += eig_dpp0<0x111, 0xf>(x); x += eig_dpp0<0x112, 0xf>(x); x += eig_dpp0<0x114, 0xf>(x); x += eig_dpp0<0x118, 0xf>(x);
    x += eig_dpp0<0x142, 0xa>(x); x += eig_dpp0<0x143, 0xc>(x);
    return eig_lane(x, 63);
}
// sum over the 8 lanes of an aligned group of eight, in each of them: xor 1, xor 2 inside the quads (quad_perm), then the other quad of the
// eight (row_half_mirror: lane i <-> 7 - i)
EIGDEV double oct_sum(double x) {
    x += eig_dpp0<0xb1, 0xf>(x);          // quad_perm [1, 0, 3, 2]
    x += eig_dpp0<0x4e, 0xf>(x);          // quad_perm [2, 3, 0, 1]
    x += eig_dpp0<0x141, 0xf>(x);         // row_half_mirror
    return x;
}
EIGDEV double wave_max(double x) { for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o)); return x; }
// a ring entry's tag word: index of the rotation (bits 0-7), first / last of its sweep, end of the stream, and the entry's sequence
// number divided by the ring size (bits 12-27: which lap of the ring it belongs to) + 1, so that a slot that still holds an older lap --
// or the zeros the ring starts with -- is told from one that has been written
constexpr int kQlFirst = 1 << 30, kQlLast = 1 << 29, kQlDoneFlag = 1 << 28, kQlIndex = 0xff, kQlTagMask = 0xffff << 12;
constexpr int kQlSpinCap = 4000000;                            // polls (tens of cycles each) before a waiting side gives up
enum { kQlTail0 = 2, kQlTail1 = 3, kQlGaveUp = 4 };

template <int L>
__global__ __launch_bounds__(QlCfg<L>::NT) void k_small_eigh_ql(const double* __restrict__ src, int nslices, int n, int Lw, int k, int zmode, double denom,
                                                                const int* __restrict__ cholflag, double* __restrict__ Z, double* __restrict__ res,
                                                                double* __restrict__ Vout) {
    using C = QlCfg<L>;
    constexpr int NT = C::NT, RING = C::RING;
    extern __shared__ double eig_sm[];
    // (LDS pointers by type: a volatile access through a generic pointer is a FLAT instruction with a full wait behind it -- the first
    //  form of this kernel spent 1 400 cycles per rotation that way)
    typedef __attribute__((address_space(3))) double lds_d;
    typedef __attribute__((address_space(3))) int lds_i;
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2d lds_d2;
    lds_d* A = (lds_d*)eig_sm;                                 // [L x L], element (r, c) at ql_at(r, c); becomes Q, then Z
    lds_d* de = A + L * L;                                     // d[i] = de[2 i] (diagonal / eigenvalues), e[i] = de[2 i + 1] (couples i and i + 1): one 16-byte access for the pair
    lds_d* vv = A + L * L + 2 * L;                             // the reflector of the step
    lds_d* pp = vv + L;
    lds_d* ww = pp + L;
    lds_d* bb = ww + L;                                        // beta of reflector k
    lds_d* red = bb + L;                                       // [64] reduction scratch
    volatile lds_d2* ringcs = (volatile lds_d2*)(red + 64);    // (c, s) of a rotation
    volatile lds_i* ringi = (volatile lds_i*)(red + 64 + 2 * RING);
    volatile lds_i* ctrl = ringi + RING;
    __shared__ double wsl[128];
    __shared__ int order[128], genuine[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    EIG_STAMP(0)
    const int S = nslices > 0 ? nslices : 1;
    const size_t LL = (size_t)Lw * Lw;
    for (int e0 = tid; e0 < L * L; e0 += NT) A[e0] = 0.0;
    if (tid < 16) ctrl[tid] = 0;
    for (int e0 = tid; e0 < RING; e0 += NT) ringi[e0] = 0;     // (no slot carries a valid tag yet)
    if (tid < 128) { order[tid] = 0; genuine[tid] = tid < n; }
    __syncthreads();
    // fold the slices (slice order), eight of them in flight per element
    for (int e0 = tid; e0 < n * n; e0 += NT) {
        const int r = e0 / n, c = e0 - r * n;
        const double* p0 = src + (size_t)r * Lw + c;
        double acc = 0.0;
        for (int s0 = 0; s0 < S; s0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p0[(s0 + u < S ? s0 + u : s0) * LL];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (s0 + u < S) ? v[u] : 0.0;
        }
        A[ql_at<L>(r, c)] = acc;
    }
    __syncthreads();
    // symmetrise (the upper-triangle thread of a pair writes both), largest finite magnitude, prescale by a power of two
    double amax = 0.0;
    for (int e0 = tid; e0 < n * n; e0 += NT) {
        const int r = e0 / n, c = e0 - r * n;
        if (r <= c) {
            const double x = 0.5 * (A[ql_at<L>(r, c)] + A[ql_at<L>(c, r)]);
            A[ql_at<L>(r, c)] = x; A[ql_at<L>(c, r)] = x;
            const double ax = fabs(x);
            amax = (ax > amax && ax < INFINITY) ? ax : amax;
        }
    }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = 0.0;
    for (int w = 0; w < NT / 64; ++w) amax = fmax(amax, red[w]);
    int ex = 0;
    if (amax > 0.0) (void)frexp(amax, &ex);
    const double sc = ldexp(1.0, -ex), unsc = ldexp(1.0, ex);
    for (int e0 = tid; e0 < n * n; e0 += NT) { const int r = e0 / n, c = e0 - r * n; A[ql_at<L>(r, c)] *= sc; }
    __syncthreads();
    EIG_STAMP(1)
    // a row (column) of the active block and an eighth of its elements per thread.  Element (r, c) sits in bank (r + c) mod 32 (8-byte
    // banks), so the 32 lanes one LDS pass serves take 4 rows x 8 column groups with the columns of a group 4 apart:
    // r + 4 part + const is a different bank for each of them; a thread's columns are 4 part + a + 32 b, a < 4
    const int r8 = wave * 8 + ((lane >> 3) & 3) + 4 * (lane >> 5), part = lane & 7;         // (the eight parts of a row in eight adjacent lanes: oct_sum)
    // ---- tridiagonalisation: reflector kk annihilates column kk below the subdiagonal ----
    for (int kk = 0; kk + 2 < n; ++kk) {
        const int m = n - kk - 1;
        if (wave == 0) {
            const int t1 = lane + 64;
            const double x0 = lane < m ? A[ql_at<L>(kk + 1 + lane, kk)] : 0.0, x1 = t1 < m ? A[ql_at<L>(kk + 1 + t1, kk)] : 0.0;
            const double sig = wave_sum(fma(x0, x0, x1 * x1));
            const double xf = eig_lane(x0, 0);
            double alpha = 0.0, beta = 0.0;
            if (sig > 1e-280 && sig < INFINITY) {              // H = I - beta v v^T, v = x - alpha e_1, H x = alpha e_1 (a column below 1e-140 of the scale is left alone)
                const double nrm = sig * eig_rsqrt(sig);
                alpha = xf >= 0.0 ? -nrm : nrm;
                beta = eig_rcp(sig - alpha * xf);
            }
            if (lane < m) vv[lane] = lane == 0 ? x0 - alpha : x0;
            if (t1 < m) vv[t1] = x1;
            if (lane == 0) { de[2 * kk] = A[ql_at<L>(kk, kk)]; de[2 * kk + 1] = alpha; bb[kk] = beta; }
        }
        __syncthreads();
        const double beta = bb[kk];
        {   // p = beta A22 v
            double acc = 0.0;
            if (r8 < m)
                for (int b = 0; 32 * b < m; ++b) {             // four loads in flight per trip (a lone dependent LDS read is ~100 cycles)
                    double av[4], xv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) { const int c = 4 * part + a + 32 * b, cc = c < m ? c : 0; av[a] = A[ql_at<L>(kk + 1 + r8, kk + 1 + cc)]; xv[a] = c < m ? vv[cc] : 0.0; }
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc = fma(av[a], xv[a], acc);
                }
            acc = oct_sum(acc);
            if (part == 0 && r8 < m) pp[r8] = beta * acc;
        }
        __syncthreads();
        if (wave == 0) {                                       // w = p - (beta / 2) (v^T p) v
            const int t1 = lane + 64;
            const double v0 = lane < m ? vv[lane] : 0.0, v1 = t1 < m ? vv[t1] : 0.0, p0 = lane < m ? pp[lane] : 0.0, p1 = t1 < m ? pp[t1] : 0.0;
            const double kq = 0.5 * beta * wave_sum(fma(v0, p0, v1 * p1));
            if (lane < m) ww[lane] = fma(-kq, v0, p0);
            if (t1 < m) ww[t1] = fma(-kq, v1, p1);
        }
        __syncthreads();
        if (r8 < m) {                                          // A22 -= v w^T + w v^T
            const double vr = vv[r8], wr = ww[r8];
            for (int b = 0; 32 * b < m; ++b) {
                double av[4], wv[4], xv[4]; int at[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { const int c = 4 * part + a + 32 * b, cc = c < m ? c : 0; at[a] = ql_at<L>(kk + 1 + r8, kk + 1 + cc); av[a] = A[at[a]]; wv[a] = ww[cc]; xv[a] = vv[cc]; }
#pragma unroll
                for (int a = 0; a < 4; ++a) if (4 * part + a + 32 * b < m) A[at[a]] = fma(-vr, wv[a], fma(-wr, xv[a], av[a]));
            }
        }
        if (tid < m) A[ql_at<L>(kk + 1 + tid, kk)] = vv[tid];  // the reflector stays in its column for the accumulation below
        __syncthreads();
    }
    if (tid == 0) {
        if (n >= 2) { de[2 * (n - 2)] = A[ql_at<L>(n - 2, n - 2)]; de[2 * (n - 2) + 1] = A[ql_at<L>(n - 1, n - 2)]; }
        de[2 * (n - 1)] = A[ql_at<L>(n - 1, n - 1)]; de[2 * (n - 1) + 1] = 0.0;
    }
    __syncthreads();
    // ---- Q = H_0 H_1 ... in place, from the last reflector back: the trailing block holds the product so far ----
    if (tid < 4 && n >= 2) { const int r = n - 2 + (tid >> 1), c = n - 2 + (tid & 1); A[ql_at<L>(r, c)] = r == c ? 1.0 : 0.0; }
    if (tid == 0 && n == 1) A[0] = 1.0;
    __syncthreads();
    for (int kk = n - 3; kk >= 0; --kk) {
        const int m = n - kk - 1;
        const double beta = bb[kk];
        if (tid < m) vv[tid] = A[ql_at<L>(kk + 1 + tid, kk)];
        __syncthreads();
        {   // t = beta v^T Q22  (column r8 of the block, an eighth of its rows per thread)
            double acc = 0.0;
            if (r8 < m)
                for (int b = 0; 32 * b < m; ++b) {
                    double av[4], xv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) { const int r = 4 * part + a + 32 * b, rr = r < m ? r : 0; av[a] = A[ql_at<L>(kk + 1 + rr, kk + 1 + r8)]; xv[a] = r < m ? vv[rr] : 0.0; }
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc = fma(xv[a], av[a], acc);
                }
            acc = oct_sum(acc);
            if (part == 0 && r8 < m) pp[r8] = beta * acc;
        }
        __syncthreads();
        if (r8 < m) {                                          // Q22 -= v t
            const double vr = vv[r8];
            for (int b = 0; 32 * b < m; ++b) {
                double av[4], tv[4]; int at[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { const int c = 4 * part + a + 32 * b, cc = c < m ? c : 0; at[a] = ql_at<L>(kk + 1 + r8, kk + 1 + cc); av[a] = A[at[a]]; tv[a] = pp[cc]; }
#pragma unroll
                for (int a = 0; a < 4; ++a) if (4 * part + a + 32 * b < m) A[at[a]] = fma(-vr, tv[a], av[a]);
            }
        }
        if (tid < m) { A[ql_at<L>(kk, kk + 1 + tid)] = 0.0; A[ql_at<L>(kk + 1 + tid, kk)] = 0.0; }
        if (tid == 0) A[ql_at<L>(kk, kk)] = 1.0;
        __syncthreads();
    }
    EIG_STAMP(2)
    // ---- implicit QL on (dd, ee); rotations through the ring to the waves that own the rows of Z ----
    int capped = 0;
    if (wave == 0) {
        // The chain.  A lone wave issues an instruction every ~5 cycles whatever it is (scripts/kbench/probe_lat.hip), so a rotation
        // costs its instruction COUNT (profiles/r5_kbench_summary.md section 1 has the forms that were tried).  d and e in LDS through
        // LDS-typed plain pointers, the next rotation's entries fetched one rotation ahead; loop counters in SGPRs; room in the ring
        // checked once per sweep (a sweep is at most n - 1 < RING rotations); four stores of lane 0 per rotation -- the entry's tag
        // carries its sequence number, there is no separate head to publish.
        const double eps = 2.220446049250313e-16;
        lds_d2* dep = (lds_d2*)de;                               // (this wave is the only one that touches d and e from here on)
        const int nu = __builtin_amdgcn_readfirstlane(n);
        double f = 0.0, tst1 = 1e-20;
        int head = 0, tail_seen = 0, iters_all = 0; long long full_spins = 0;
        bool gave_up = false;
        for (int l = 0; l < nu && !gave_up; ++l) {
            { const v2d x_ = dep[l]; tst1 = fmax(tst1, fabs(x_.x) + fabs(x_.y)); }
            auto first_small = [&]() -> int {                  // smallest m >= l with |e[m]| <= eps tst1 (e[n - 1] = 0 ends the search)
                int mine = nu - 1;
                const int i0 = l + lane, i1 = i0 + 64;
                if (i1 < nu && !(fabs(de[2 * i1 + 1]) > eps * tst1)) mine = i1;
                if (i0 < nu && !(fabs(de[2 * i0 + 1]) > eps * tst1)) mine = i0;
                for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(mine, o); mine = other < mine ? other : mine; }
                return __builtin_amdgcn_readfirstlane(mine);
            };
            int m = first_small();
            if (m > l) {
                int iter = 0;
                bool more = true;
                while (more) {
                    ++iter; EIG_COUNT(iters_all);
                    // room for the whole sweep (bounded wait)
                    if (head + (m - l) - tail_seen > RING) {
                        int spins = 0;
                        for (;;) {
                            int t = ctrl[kQlTail0];
                            if (C::NCONS > 1) { const int t1 = ctrl[kQlTail1]; t = t1 < t ? t1 : t; }
                            tail_seen = __builtin_amdgcn_readfirstlane(t);
                            if (head + (m - l) - tail_seen <= RING) break;
                            EIG_COUNT(full_spins);
                            if (++spins > kQlSpinCap) { gave_up = true; break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (gave_up) break;
                    }
                    const v2d x0_ = dep[l];
                    double g = x0_.x;
                    const double el = x0_.y;
                    double p = (de[2 * (l + 1)] - g) / (2.0 * el);
                    double r = sqrt(fma(p, p, 1.0));
                    if (p < 0.0) r = -r;
                    const double dl = el / (p + r), dl1 = el * (p + r);
                    double h = g - dl;
                    __builtin_amdgcn_wave_barrier();
                    for (int i = l + lane; i < nu; i += 64) { const double x = de[2 * i]; de[2 * i] = i == l ? dl : (i == l + 1 ? dl1 : x - h); }
                    __builtin_amdgcn_wave_barrier();
                    f += h;
                    p = de[2 * m];
                    double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                    const double el1 = de[2 * (l + 1) + 1];
                    double e_i, d_i;
                    { const v2d x_ = dep[m - 1]; d_i = x_.x; e_i = x_.y; }
                    for (int i = m - 1; i >= l; --i) {
                        const int ip = i > l ? i - 1 : i;
                        const v2d nx_ = dep[ip];                                  // one rotation ahead
                        const double e_nx = nx_.y, d_nx = nx_.x;
                        c3 = c2; c2 = c; s2 = s;
                        g = c * e_i; h = c * p;
                        const double q2 = fma(p, p, e_i * e_i);
                        const double rinv = eig_rsqrt(q2);
                        r = q2 * rinv;
                        const double s_old = s;
                        s = e_i * rinv; c = p * rinv;
                        p = fma(c, d_i, -(s * g));
                        if (lane == 0) {
                            const int slot = head & (RING - 1);
                            dep[i + 1] = v2d{fma(s, fma(c, g, s * d_i), h), s_old * r};
                            ringcs[slot] = v2d{c, s};
                            ringi[slot] = i | (i == m - 1 ? kQlFirst : 0) | (i == l ? kQlLast : 0) | ql_tag(head);
                        }
                        ++head;
                        e_i = e_nx; d_i = d_nx;
                    }
                    p = -s * s2 * c3 * el1 * de[2 * l + 1] / dl1;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) dep[l] = v2d{c * p, s * p};
                    __builtin_amdgcn_wave_barrier();
                    more = fabs(s * p) > eps * tst1;
                    if (more && iter >= 60) { capped = 1; more = false; }
                    if (more) m = first_small();               // (the block may have split during the sweep)
                    if (more && m == l) more = false;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) dep[l] = v2d{de[2 * l] + f, 0.0};
            __builtin_amdgcn_wave_barrier();
        }
        if (gave_up) capped = 1;
#if GPCA_EIG_STAMP
        if (lane == 0) { g_eig_stamp[10] = (unsigned long long)iters_all; g_eig_stamp[11] = (unsigned long long)head; g_eig_stamp[12] = (unsigned long long)full_spins; }
#endif
        (void)iters_all; (void)full_spins;
        // the end of the stream: an entry with the Done flag (the ring has room for it: wait like a sweep of one)
        if (lane == 0) {
            int spins = 0;
            while (head + 1 - (C::NCONS > 1 ? (ctrl[kQlTail1] < ctrl[kQlTail0] ? ctrl[kQlTail1] : ctrl[kQlTail0]) : ctrl[kQlTail0]) > RING && ++spins < kQlSpinCap) __builtin_amdgcn_s_sleep(2);
            const int slot = head & (RING - 1);
            ringcs[slot] = v2d{0.0, 0.0};
            ringi[slot] = kQlDoneFlag | ql_tag(head);
            if (capped) ctrl[kQlGaveUp] = 1;
        }
    } else if (wave <= C::NCONS) {
        const int row = (wave - 1) * 64 + lane;
        const bool live = row < n;
        lds_d* zr = A + row * L;                               // element c of the row at (c + row) & (L - 1)
        int tail = 0, spins = 0;
        long long batches = 0, empty_polls = 0;
        double carry = 0.0;
        bool finished = false;
        while (!finished) {
            // how many entries from `tail` on carry the tag of their sequence number?  (up to four; a wave's LDS stores land in order:
            // a tag that is there has its c and s behind it)
            int ii[4], navail = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) ii[j] = ringi[(tail + j) & (RING - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) if (navail == j && (ii[j] & kQlTagMask) == ql_tag(tail + j)) navail = j + 1;
            if (navail == 0) {
                if (++spins > kQlSpinCap) { ctrl[kQlGaveUp] = 1; break; }
                EIG_COUNT(empty_polls);
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            spins = 0;
            int len = navail;
#pragma unroll
            for (int j = 3; j >= 1; --j) if (j < len && (ii[j] & (kQlFirst | kQlDoneFlag))) len = j;     // a new sweep (or the end) starts a new batch
            if (ii[0] & kQlDoneFlag) { finished = true; break; }
            v2d cs[4]; double lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = ringcs[(tail + (j < len ? j : 0)) & (RING - 1)];
            if (live) {
                const int i0 = ii[0] & kQlIndex;
#pragma unroll
                for (int j = 0; j < 4; ++j) lo[j] = zr[((ii[j < len ? j : 0] & kQlIndex) + row) & (L - 1)];
                double hi = (ii[0] & kQlFirst) ? (double)zr[(i0 + 1 + row) & (L - 1)] : carry;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < len) {
                        const int i = ii[j] & kQlIndex;
                        zr[(i + 1 + row) & (L - 1)] = fma(cs[j].y, lo[j], cs[j].x * hi);
                        hi = fma(cs[j].x, lo[j], -(cs[j].y * hi));
                        if (ii[j] & kQlLast) zr[(i + row) & (L - 1)] = hi;
                    }
                carry = hi;
            }
            tail += len; EIG_COUNT(batches);
            if ((tail & 63) < len && lane == 0) ctrl[kQlTail0 + wave - 1] = tail;          // (the chain looks at it once per sweep: every 64 entries is often enough)
        }
#if GPCA_EIG_STAMP
        if (wave == 1 && lane == 0) { g_eig_stamp[13] = (unsigned long long)batches; g_eig_stamp[14] = (unsigned long long)empty_polls; }
#endif
        (void)batches; (void)empty_polls;
    }
    __syncthreads();
    capped = ctrl[kQlGaveUp] != 0 || (wave == 0 && capped);
    if (wave == 0 && lane == 0 && capped) ctrl[kQlGaveUp] = 1;
    __syncthreads();
    capped = ctrl[kQlGaveUp];
    EIG_STAMP(3)
    if (tid < 128) wsl[tid] = tid < n ? de[2 * tid] : 0.0;
    __syncthreads();
    eig_emit<NT>([&](int r, int slot) { return A[ql_at<L>(r, slot)]; }, wsl, order, genuine, L, n, Lw, k, zmode, denom, unsc, cholflag, capped, Z, res, Vout, tid);
}

template <int L> static size_t small_eigh_lds() { return sizeof(double) * (size_t)L * JacCfg<L>::LP * JacCfg<L>::NBUF; }
#ifndef GPCA_EIG_QL
#define GPCA_EIG_QL 1           // 1: sketches of 65-128 columns by tridiagonalisation + QL (k_small_eigh_ql), the rest by Jacobi; harness A/B: 0 = Jacobi
#endif                          // everywhere, 2 = QL from 33 columns
int init_device_kernels_eig() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds<128>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh_ql<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)QlCfg<128>::kBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh_ql<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)QlCfg<64>::kBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds<64>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_eigh<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_eigh_lds<32>());
    return (int)e;
}
void launch_small_eigh(hipStream_t st, const double* src, int nslices, int n, int L, int k, int zmode, double denom, const int* cholflag,
                       double* Z, double* res, double* Vout) {
    if (n <= 32) hipLaunchKernelGGL(k_small_eigh<32>, dim3(1), dim3(JacCfg<32>::NT), small_eigh_lds<32>(), st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else if (GPCA_EIG_QL == 2 && n <= 64) hipLaunchKernelGGL(k_small_eigh_ql<64>, dim3(1), dim3(QlCfg<64>::NT), QlCfg<64>::kBytes, st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else if (GPCA_EIG_QL && n > 64) hipLaunchKernelGGL(k_small_eigh_ql<128>, dim3(1), dim3(QlCfg<128>::NT), QlCfg<128>::kBytes, st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else if (n <= 64) hipLaunchKernelGGL(k_small_eigh<64>, dim3(1), dim3(JacCfg<64>::NT), small_eigh_lds<64>(), st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
    else hipLaunchKernelGGL(k_small_eigh<128>, dim3(1), dim3(JacCfg<128>::NT), small_eigh_lds<128>(), st, src, nslices, n, L, k, zmode, denom, cholflag, Z, res, Vout);
}

}  // namespace gpca
