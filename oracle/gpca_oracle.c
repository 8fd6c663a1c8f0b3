/*
 * gpca_oracle.c -- CPU restatement of the genomic_pca hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (genomic_pca_amd/, libgpca.so) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - snp_stats / HWE / standardise: restated line by line from the reference's own source
 *     (file:line given at each function).  The reference holds no golden vectors for them
 *     (SURVEY.md F4), so they are pinned only by hand-derived known answers in tests/.
 *   - rSVD: "parity unpinned".  The reference's arithmetic lives in the un-vendored,
 *     un-pinned git dependency `efficient_pca` (Cargo.toml:30, branch = "main", no Cargo.lock).
 *     This file restates the published randomized-SVD recipe (Halko/Martinsson/Tropp 2011,
 *     alg. 4.4 + 5.1: Gaussian sketch, QR-stabilised power iterations, projection, small SVD)
 *     with the parameters the reference's call sites fix: l = k + 10 (main.rs:636, :317),
 *     q = 2 (main.rs:318), seed (main.rs:637), standardisation (g-mu)/sigma with sample (n-1)
 *     sigma (prepare.rs:1294,1357-1364,1948-1988), eigenvalues s^2/(N-1).
 *
 *   - orc_rsvd below is the SAME recipe the HIP engine runs (CholeskyQR2 + cyclic Jacobi): it is the timed CPU baseline
 *     ("port") and a cross-check.  The parity CHECKER of tests/ is oracle.py:rsvd(method="lapack"): the same sketch and the
 *     same two products (orc_prod_AQ / orc_prod_AtT), but Householder QR (LAPACK geqrf) for the tall orthonormalisation and
 *     LAPACK gesdd for the small factorisation -- no small-dense code in common with the product.
 *
 * Build: see oracle/Makefile (REAL=double -> checker, REAL=float -> timed CPU baseline).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define GPCA_MISSING ((int8_t)-127) /* prepare.rs:1224 */

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11).  Same constants and round structure as the device
 * generator in genomic_pca_amd/csrc/philox.hpp, written independently here.
 * ---------------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                uint32_t* out) {
    philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

/* Synthetic genotypes: sample n belongs to population n % P; per SNP i and
 * population c the caller supplies thresh[i*P+c] = floor(p_ic * 2^32); each genotype is the
 * number of two independent uniform u32 draws below the threshold (= Binomial(2, p_ic)).
 * Counter = (snp_lo, snp_hi, n/2, stream 0x47454E4F), key = seed; draws 0,1 -> even n, 2,3 -> odd. */
void orc_synth_genotypes(int8_t* G, int64_t M, int64_t N, int64_t ld, int64_t snp_offset,
                         uint64_t seed, const uint32_t* thresh, int P) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        uint64_t gi = (uint64_t)(i + snp_offset);
        for (int64_t n2 = 0; n2 < (N + 1) / 2; ++n2) {
            uint32_t o[4];
            philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)n2, 0x47454E4Fu,
                          (uint32_t)seed, (uint32_t)(seed >> 32), o);
            int64_t n = 2 * n2;
            uint32_t t = thresh[i * P + (n % P)];
            G[i * ld + n] = (int8_t)((o[0] < t) + (o[1] < t));
            if (n + 1 < N) {
                t = thresh[i * P + ((n + 1) % P)];
                G[i * ld + n + 1] = (int8_t)((o[2] < t) + (o[3] < t));
            }
        }
    }
}

/* SplitMix64 (Steele, Lea, Flood 2014): output i (0-based) of the stream seeded with `seed` is mix(seed + (i + 1) * gamma).
 * Written independently of genomic_pca_amd/csrc/philox.hpp; pinned by the published outputs for seed 1234567. */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t orc_splitmix64_at(uint64_t seed, uint64_t i) { return splitmix64_at(seed, i); }

/* Fast panel generator (GPCA_PANEL_SYNTH16; device twin: genomic_pca_amd/csrc/kernels.hip:k_synth16).  One 16-bit uniform
 * per genotype: sample n of SNP (global row) gi takes bits 16 (n % 4) .. +15 of SplitMix64 output number (gi << 26) + n / 4
 * of the stream seeded with `seed`;  g = (u < t1) + (u < t2) with the thresholds of the sample's population (n / 16) % P --
 * blocks of 16 consecutive samples share a population -- packed in thresh[i*P + pop]: high half
 * t1 = floor(P(g >= 1) * 65536), low half t2 = floor(P(g = 2) * 65536). */
void orc_synth16_genotypes(int8_t* G, int64_t M, int64_t N, int64_t ld, int64_t snp_offset,
                           uint64_t seed, const uint32_t* thresh, int P) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        uint64_t gi = (uint64_t)(i + snp_offset);
        for (int64_t q = 0; q < (N + 3) / 4; ++q) {
            uint64_t z = splitmix64_at(seed, (gi << 26) + (uint64_t)q);
            for (int j = 0; j < 4 && 4 * q + j < N; ++j) {
                int64_t n = 4 * q + j;
                uint32_t tw = thresh[i * P + ((n >> 4) % P)];
                uint32_t u = (uint32_t)(z >> (16 * j)) & 0xffffu;
                G[i * ld + n] = (int8_t)((u < (tw >> 16)) + (u < (tw & 0xffffu)));
            }
        }
    }
}

/* Standard normal for the sketch matrix Omega[i][j]: counter = (i_lo, i_hi, j/4, 0x4F4D4547),
 * the 4 outputs give 4 normals for columns 4*(j/4)..+3 by two Box-Muller pairs.
 * u in (0,1]: (x + 1) * 2^-32. */
static inline void omega4(uint64_t i, uint32_t jq, uint64_t seed, double z[4]) {
    uint32_t o[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), jq, 0x4F4D4547u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), o);
    const double s = 1.0 / 4294967296.0, twopi = 6.283185307179586476925286766559;
    double u0 = ((double)o[0] + 1.0) * s, u1 = ((double)o[1] + 1.0) * s;
    double u2 = ((double)o[2] + 1.0) * s, u3 = ((double)o[3] + 1.0) * s;
    double r0 = sqrt(-2.0 * log(u0)), r1 = sqrt(-2.0 * log(u2));
    z[0] = r0 * cos(twopi * u1); z[1] = r0 * sin(twopi * u1);
    z[2] = r1 * cos(twopi * u3); z[3] = r1 * sin(twopi * u3);
}

void orc_omega(double* Om, int64_t M, int l, int64_t snp_offset, uint64_t seed) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        for (int jq = 0; jq < (l + 3) / 4; ++jq) {
            double z[4];
            omega4((uint64_t)(i + snp_offset), (uint32_t)jq, seed, z);
            for (int t = 0; t < 4 && 4 * jq + t < l; ++t) Om[i * l + 4 * jq + t] = z[t];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * HWE chi-squared p-value.  Follows prepare.rs:1641-1745 branch for branch.
 * statrs ChiSquared(1).cdf(x) = P(1/2, x/2) = erf(sqrt(x/2)); p = max(0, 1 - cdf).
 * ---------------------------------------------------------------------------------------- */
double orc_hwe_p(uint64_t n_hom1, uint64_t n_het, uint64_t n_hom2) {
    uint64_t tot = n_hom1 + n_het + n_hom2;
    if (tot == 0) return 1.0;                                         /* :1647-1650 */
    double c1 = 2.0 * (double)n_hom1 + (double)n_het;                 /* :1652 */
    double c2 = 2.0 * (double)n_hom2 + (double)n_het;                 /* :1653 */
    double ta = c1 + c2;
    if (ta <= 1e-9) return 1.0;                                       /* :1656-1659 */
    double f1 = c1 / ta, f2 = c2 / ta;                                /* :1661-1662 */
    if (f1 < 1e-9 || f2 < 1e-9) return 1.0;                           /* :1664-1667 */
    if (fabs(f1 + f2 - 1.0) > 1e-6) return 1.0;                       /* :1668-1675 */
    double e1 = f1 * f1 * (double)tot;                                /* :1677-1682 */
    double eh = 2.0 * f1 * f2 * (double)tot;
    double e2 = f2 * f2 * (double)tot;
    double chi = 0.0;
    const double MINE = 1e-9;
    if (e1 > MINE) { double d = (double)n_hom1 - e1; chi += d * d / e1; }
    else if ((double)n_hom1 > MINE) chi = INFINITY;                   /* :1687-1693 */
    if (isfinite(chi)) {
        if (eh > MINE) { double d = (double)n_het - eh; chi += d * d / eh; }
        else if ((double)n_het > MINE) chi = INFINITY;                /* :1695-1703 */
    }
    if (isfinite(chi)) {
        if (e2 > MINE) { double d = (double)n_hom2 - e2; chi += d * d / e2; }
        else if ((double)n_hom2 > MINE) chi = INFINITY;               /* :1705-1713 */
    }
    if (isnan(chi)) return 1.0;                                       /* :1715-1721 */
    if (chi == INFINITY) return 0.0;                                  /* :1723-1725 */
    double cdf = erf(sqrt(chi * 0.5));                                /* :1727-1729 */
    if (isnan(cdf)) return 1.0;
    double p = 1.0 - cdf;                                             /* :1737 */
    return p > 0.0 ? p : 0.0;
}

/* ------------------------------------------------------------------------------------------
 * Per-SNP QC + standardisation parameters.  Follows prepare.rs:1216-1375 (the SIMD body; the
 * 32-lane chunking only changes f64 summation order of pass 2, documented in DESIGN.md).
 * G is SNP-major (each SNP's N samples contiguous, like the F-order column of prepare.rs:626).
 * Outputs per SNP: mu, sigma (f32, 0 if dropped), keep (1/0), reason code, counts.
 *   reason: 0 kept, 1 call-rate, 2 no valid, 3 MAF, 4 monomorphic, 5 HWE, 6 variance
 * ---------------------------------------------------------------------------------------- */
void orc_snp_stats(const int8_t* G, int64_t M, int64_t N, int64_t ld, double min_call_rate,
                   double min_maf, double max_hwe_p, float* mu, float* sigma, uint8_t* keep,
                   uint8_t* reason, uint32_t* counts /* [M][4]: n_valid,n0,n1,n2 */,
                   double* sum_out, double* ss_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        const int8_t* row = G + i * ld;
        uint32_t nv = 0, n0 = 0, n1 = 0, n2 = 0;
        double sum = 0.0;
        for (int64_t n = 0; n < N; ++n) {                              /* pass 1 :1233-1279 */
            int8_t v = row[n];
            if (v != GPCA_MISSING) {
                nv++; sum += (double)v;
                if (v == 0) n0++; else if (v == 1) n1++; else if (v == 2) n2++;
            }
        }
        mu[i] = 0.f; sigma[i] = 0.f; keep[i] = 0;
        if (counts) { counts[4*i] = nv; counts[4*i+1] = n0; counts[4*i+2] = n1; counts[4*i+3] = n2; }
        if (sum_out) sum_out[i] = sum;
        if (ss_out) ss_out[i] = 0.0;
        uint8_t why = 0;
        double mean = 0.0;
        do {
            double call_rate = (double)nv / (double)N;                 /* :1283 */
            if (call_rate < min_call_rate) { why = 1; break; }         /* :1284 */
            if (nv == 0) { why = 2; break; }                           /* :1292 */
            mean = sum / (double)nv;                                   /* :1294 */
            double p = mean / 2.0;                                     /* :1295 */
            double maf = p < 1.0 - p ? p : 1.0 - p;                    /* :1296 */
            if (maf < min_maf) { why = 3; break; }                     /* :1299 */
            if (fabs(p) < 1e-9 || fabs(1.0 - p) < 1e-9) { why = 4; break; } /* :1302 */
            if (max_hwe_p < 1.0) {                                     /* :1306 */
                double hp = orc_hwe_p(n0, n1, n2);                     /* :1307-1309 */
                if (hp <= max_hwe_p) { why = 5; break; }               /* :1310 */
            }
        } while (0);
        if (why) { if (reason) reason[i] = why; continue; }
        double ss = 0.0;                                               /* pass 2 :1316-1352 */
        for (int64_t n = 0; n < N; ++n) {
            int8_t v = row[n];
            if (v != GPCA_MISSING) { double d = (double)v - mean; ss += d * d; }
        }
        if (ss_out) ss_out[i] = ss;
        double var = nv >= 2 ? ss / (double)(nv - 1) : 0.0;            /* :1357-1361 */
        if (var <= 1e-9) { if (reason) reason[i] = 6; continue; }      /* :1363 */
        mu[i] = (float)mean;                                           /* :1313 */
        sigma[i] = (float)sqrt(var);                                   /* :1364 */
        keep[i] = 1;
        if (reason) reason[i] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * Standardised block, the L2 boundary.  Follows prepare.rs:1884-2016:
 *   sigma < 1e-9  -> zeros (:1899-1945);  else out = fma((f32)g, 1/sigma, -mu * (1/sigma))
 *   (:1948-1949, :1988, :2011).  Any -127 -> error (:1909-1911).  Returns 0, or 1 + flat index of
 *   the first missing genotype in (snp-major) scan order.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_standardize_block(const int8_t* G, int64_t ld, const float* mu, const float* sigma,
                              const int64_t* snp_ids, int64_t ns, const int64_t* sample_ids,
                              int64_t nj, float* out) {
    for (int64_t a = 0; a < ns; ++a) {
        int64_t i = snp_ids[a];
        float m = mu[i], sd = sigma[i];
        const int8_t* row = G + i * ld;
        if (fabsf(sd) < 1e-9f) {
            for (int64_t c = 0; c < nj; ++c) {
                if (row[sample_ids[c]] == GPCA_MISSING) return 1 + a * nj + c;
                out[a * nj + c] = 0.0f;
            }
        } else {
            float rs = 1.0f / sd;
            float bt = -m * rs;
            for (int64_t c = 0; c < nj; ++c) {
                int8_t v = row[sample_ids[c]];
                if (v == GPCA_MISSING) return 1 + a * nj + c;
                out[a * nj + c] = fmaf((float)v, rs, bt);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Small dense helpers (double): Cholesky, upper-triangular inverse, cyclic Jacobi eigensolver.
 * ---------------------------------------------------------------------------------------- */
static int chol_upper(double* W, int n) { /* in place: W = R^T R, R upper; returns 0 ok */
    for (int j = 0; j < n; ++j) {
        double d = W[j * n + j];
        for (int k = 0; k < j; ++k) d -= W[k * n + j] * W[k * n + j];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        W[j * n + j] = d;
        for (int c = j + 1; c < n; ++c) {
            double s = W[j * n + c];
            for (int k = 0; k < j; ++k) s -= W[k * n + j] * W[k * n + c];
            W[j * n + c] = s / d;
        }
        for (int r = j + 1; r < n; ++r) W[r * n + j] = 0.0;
    }
    return 0;
}
static void upper_inverse(const double* R, double* X, int n) {
    memset(X, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        X[j * n + j] = 1.0 / R[j * n + j];
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
            for (int k = i + 1; k <= j; ++k) s += R[i * n + k] * X[k * n + j];
            X[i * n + j] = -s / R[i * n + i];
        }
    }
}
/* A symmetric n x n (row-major, destroyed); V gets eigenvectors in columns; w eigenvalues, sorted desc */
static void jacobi_eigh(double* A, double* V, double* w, int n) {
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i) { dg += A[i*n+i]*A[i*n+i]; for (int j = i + 1; j < n; ++j) off += A[i*n+j]*A[i*n+j]; }
        if (off <= 1e-30 * dg || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            double apq = A[p * n + q];
            if (apq == 0.0) continue;
            double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < n; ++k) {
                double akp = A[k*n+p], akq = A[k*n+q];
                A[k*n+p] = c * akp - s * akq; A[k*n+q] = s * akp + c * akq;
            }
            for (int k = 0; k < n; ++k) {
                double apk = A[p*n+k], aqk = A[q*n+k];
                A[p*n+k] = c * apk - s * aqk; A[q*n+k] = s * apk + c * aqk;
            }
            for (int k = 0; k < n; ++k) {
                double vkp = V[k*n+p], vkq = V[k*n+q];
                V[k*n+p] = c * vkp - s * vkq; V[k*n+q] = s * vkp + c * vkq;
            }
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; ++i) { /* selection sort, descending */
        int m = i;
        for (int j = i + 1; j < n; ++j) if (w[j] > w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; ++k) { double u = V[k*n+i]; V[k*n+i] = V[k*n+m]; V[k*n+m] = u; }
        }
    }
}
void orc_jacobi_eigh(double* A, double* V, double* w, int n) { jacobi_eigh(A, V, w, n); }

/* ------------------------------------------------------------------------------------------
 * The two tall-skinny products on the implicitly standardised matrix
 *   A[i][n] = g[i][n] * r[i] + b[i],  r = 1/sigma, b = -mu * r (both f32, prepare.rs:1948-1949).
 * A is never materialised:  (A Q)[i] = r_i (g_i . Q) + b_i (1^T Q);  A^T T = G^T (r o T) + 1 (b^T T).
 * ---------------------------------------------------------------------------------------- */
/* Both products are tiled so that every thread owns its outputs outright (rows of T; a sample tile of Y): no per-thread copies of
 * the N x l block and no reduction over threads, whose cost grew with the thread count (307 MB of partials at 256 threads).  The
 * skinny operand is padded to LP = 32 or 64 columns so that the inner loops have a compile-time trip count (gcc vectorises them). */
#define ORC_ROWS_PER_BLOCK 2
#define ORC_SAMPLE_TILE 64
static int padded_cols(int l) { return l <= 32 ? 32 : (l <= 64 ? 64 : ((l + 15) / 16) * 16); }

#define ORC_DEFINE_AQ(NAME, LP)                                                                                         \
static void NAME(const int8_t* G, int64_t M, int64_t N, int64_t ld, const float* r, const float* b, const real* Qp,      \
                 const real* s, int l, real* T) {                                                                       \
    _Pragma("omp parallel for schedule(static)")                                                                         \
    for (int64_t i0 = 0; i0 < M; i0 += ORC_ROWS_PER_BLOCK) {                                                             \
        real acc[ORC_ROWS_PER_BLOCK][LP];                                                                                \
        const int nr = (int)(M - i0 < ORC_ROWS_PER_BLOCK ? M - i0 : ORC_ROWS_PER_BLOCK);                                 \
        for (int a = 0; a < ORC_ROWS_PER_BLOCK; ++a) for (int j = 0; j < LP; ++j) acc[a][j] = 0;                         \
        const int8_t* row0 = G + i0 * ld;                                                                                \
        const int8_t* row1 = G + (i0 + (nr > 1 ? 1 : 0)) * ld;                                                           \
        for (int64_t n = 0; n < N; ++n) {                                                                                \
            const real* q = Qp + n * LP;                                                                                 \
            const real g0 = (real)row0[n], g1 = (real)row1[n];                                                           \
            for (int j = 0; j < LP; ++j) { acc[0][j] += g0 * q[j]; acc[1][j] += g1 * q[j]; }                              \
        }                                                                                                                \
        for (int a = 0; a < nr; ++a) {                                                                                   \
            const real ri = (real)r[i0 + a], bi = (real)b[i0 + a];                                                       \
            for (int j = 0; j < l; ++j) T[(i0 + a) * l + j] = ri * acc[a][j] + bi * s[j];                                \
        }                                                                                                                \
    }                                                                                                                    \
}
ORC_DEFINE_AQ(prod_AQ_32, 32)
ORC_DEFINE_AQ(prod_AQ_64, 64)

/* T[M][l] = A Q ; Q is [N][l] */
static void prod_AQ(const int8_t* G, int64_t M, int64_t N, int64_t ld, const float* r,
                    const float* b, const real* Q, int l, real* T) {
    const int LP = padded_cols(l);
    real* s = (real*)calloc(l, sizeof(real));
    for (int64_t n = 0; n < N; ++n) for (int j = 0; j < l; ++j) s[j] += Q[n * l + j];
    if (LP > 64) {   /* wider than the engine ever runs: the plain loop */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < M; ++i) {
            const int8_t* row = G + i * ld;
            real ri = (real)r[i], bi = (real)b[i];
            for (int j = 0; j < l; ++j) {
                real acc = 0;
                for (int64_t n = 0; n < N; ++n) acc += (real)row[n] * Q[n * l + j];
                T[i * l + j] = ri * acc + bi * s[j];
            }
        }
        free(s);
        return;
    }
    real* Qp = (real*)calloc((size_t)N * LP, sizeof(real));
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) for (int j = 0; j < l; ++j) Qp[n * LP + j] = Q[n * l + j];
    if (LP == 32) prod_AQ_32(G, M, N, ld, r, b, Qp, s, l, T); else prod_AQ_64(G, M, N, ld, r, b, Qp, s, l, T);
    free(Qp); free(s);
}

#define ORC_DEFINE_ATT(NAME, LP)                                                                                         \
static void NAME(const int8_t* G, int64_t M, int64_t N, int64_t ld, const real* TP, const double* c, int l, double* Y) {  \
    const int64_t tiles = (N + ORC_SAMPLE_TILE - 1) / ORC_SAMPLE_TILE;                                                    \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                                                       \
    for (int64_t t = 0; t < tiles; ++t) {                                                                                 \
        const int64_t n0 = t * ORC_SAMPLE_TILE;                                                                           \
        const int nn = (int)(N - n0 < ORC_SAMPLE_TILE ? N - n0 : ORC_SAMPLE_TILE);                                        \
        real y[ORC_SAMPLE_TILE][LP];                                                                                      \
        for (int a = 0; a < ORC_SAMPLE_TILE; ++a) for (int j = 0; j < LP; ++j) y[a][j] = 0;                               \
        for (int64_t i = 0; i < M; ++i) {                                                                                 \
            const int8_t* g = G + i * ld + n0;                                                                            \
            const real* tp = TP + i * LP;                                                                                 \
            for (int a = 0; a < nn; ++a) {                                                                                \
                const real ga = (real)g[a];                                                                               \
                for (int j = 0; j < LP; ++j) y[a][j] += ga * tp[j];                                                       \
            }                                                                                                             \
        }                                                                                                                 \
        for (int a = 0; a < nn; ++a) for (int j = 0; j < l; ++j) Y[(n0 + a) * l + j] = c[j] + (double)y[a][j];            \
    }                                                                                                                     \
}
ORC_DEFINE_ATT(prod_AtT_32, 32)
ORC_DEFINE_ATT(prod_AtT_64, 64)

/* Y[N][l] = A^T T ; T is [M][l].  T' = r o T is formed once (padded to LP columns), c = b^T T in double; every thread then owns a
 * tile of 64 samples and sweeps all rows for it. */
static void prod_AtT(const int8_t* G, int64_t M, int64_t N, int64_t ld, const float* r,
                     const float* b, const real* T, int l, double* Y) {
    const int LP = padded_cols(l);
    real* TP = (real*)calloc((size_t)M * LP, sizeof(real));
    double* c = (double*)calloc(l, sizeof(double));
#pragma omp parallel
    {
        double* cp = (double*)calloc(l, sizeof(double));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < M; ++i) {
            const real ri = (real)r[i]; const double bi = (double)b[i];
            for (int j = 0; j < l; ++j) { TP[i * LP + j] = ri * T[i * l + j]; cp[j] += bi * (double)T[i * l + j]; }
        }
#pragma omp critical
        for (int j = 0; j < l; ++j) c[j] += cp[j];
        free(cp);
    }
    if (LP == 32) prod_AtT_32(G, M, N, ld, TP, c, l, Y);
    else if (LP == 64) prod_AtT_64(G, M, N, ld, TP, c, l, Y);
    else {
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; ++n)
            for (int j = 0; j < l; ++j) {
                double a = c[j];
                for (int64_t i = 0; i < M; ++i) a += (double)((real)G[i * ld + n] * TP[i * LP + j]);
                Y[n * l + j] = a;
            }
    }
    free(TP); free(c);
}
void orc_prod_AQ(const int8_t* G, int64_t M, int64_t N, int64_t ld, const float* r, const float* b,
                 const real* Q, int l, real* T) { prod_AQ(G, M, N, ld, r, b, Q, l, T); }
void orc_prod_AtT(const int8_t* G, int64_t M, int64_t N, int64_t ld, const float* r, const float* b,
                  const real* T, int l, double* Y) { prod_AtT(G, M, N, ld, r, b, T, l, Y); }

/* CholeskyQR2 of Y[N][l] (double) -> Q[N][l] (real).  Returns 0, or j+1 if pivot j failed. */
static int cholqr2(double* Y, int64_t N, int l, real* Q) {
    double* W = (double*)malloc(sizeof(double) * l * l);
    double* X = (double*)malloc(sizeof(double) * l * l);
    double* tmp = (double*)malloc(sizeof(double) * l);
    for (int round = 0; round < 2; ++round) {
        memset(W, 0, sizeof(double) * l * l);
        for (int64_t n = 0; n < N; ++n)
            for (int a = 0; a < l; ++a) { double ya = Y[n*l+a]; for (int c = a; c < l; ++c) W[a*l+c] += ya * Y[n*l+c]; }
        for (int a = 0; a < l; ++a) for (int c = 0; c < a; ++c) W[a*l+c] = W[c*l+a];
        int rc = chol_upper(W, l);
        if (rc) { free(W); free(X); free(tmp); return rc; }
        upper_inverse(W, X, l);
        for (int64_t n = 0; n < N; ++n) {
            for (int c = 0; c < l; ++c) { double s = 0; for (int a = 0; a <= c; ++a) s += Y[n*l+a] * X[a*l+c]; tmp[c] = s; }
            for (int c = 0; c < l; ++c) Y[n*l+c] = tmp[c];
        }
    }
    for (int64_t n = 0; n < N; ++n) for (int c = 0; c < l; ++c) Q[n*l+c] = (real)Y[n*l+c];
    free(W); free(X); free(tmp);
    return 0;
}
int orc_cholqr2(double* Y, int64_t N, int l, real* Q) { return cholqr2(Y, N, l, Q); }

/* ------------------------------------------------------------------------------------------
 * Full randomized PCA (same recipe the HIP engine runs; header comment gives provenance).
 *   r[i], b[i]: per-SNP scale/shift (0,0 for dropped SNPs).  snp_offset: global index of row 0
 *   (so a row-shard draws the same Omega rows it would in the unsharded run).
 * Outputs: scores[N][k], eigenvalues[k], loadings[M][k], singular values sv[l] (all double).
 * Returns 0 ok; 100+j CholQR pivot failure.
 * ---------------------------------------------------------------------------------------- */
int orc_rsvd(const int8_t* G, int64_t M, int64_t N, int64_t ld, const float* r, const float* b,
             int k, int oversample, int power_iters, uint64_t seed, int64_t snp_offset,
             double* scores, double* eigenvalues, double* loadings, double* sv) {
    int l = k + oversample;
    real* T = (real*)malloc(sizeof(real) * (size_t)M * l);
    real* Q = (real*)malloc(sizeof(real) * (size_t)N * l);
    double* Y = (double*)malloc(sizeof(double) * (size_t)N * l);
    int rc = 0;
    {   /* sketch: Y = A^T Omega */
        double* Om = (double*)malloc(sizeof(double) * (size_t)M * l);
        orc_omega(Om, M, l, snp_offset, seed);
        for (int64_t t = 0; t < M * (int64_t)l; ++t) T[t] = (real)Om[t];
        free(Om);
        prod_AtT(G, M, N, ld, r, b, T, l, Y);
        rc = cholqr2(Y, N, l, Q);
    }
    for (int it = 0; it < power_iters && !rc; ++it) {
        prod_AQ(G, M, N, ld, r, b, Q, l, T);
        prod_AtT(G, M, N, ld, r, b, T, l, Y);
        rc = cholqr2(Y, N, l, Q);
    }
    if (rc) { free(T); free(Q); free(Y); return 100 + rc; }
    prod_AQ(G, M, N, ld, r, b, Q, l, T);            /* B = A Q */
    double* C = (double*)calloc((size_t)l * l, sizeof(double));
    for (int64_t i = 0; i < M; ++i)
        for (int a = 0; a < l; ++a) { double ta = (double)T[i*l+a]; for (int c = a; c < l; ++c) C[a*l+c] += ta * (double)T[i*l+c]; }
    for (int a = 0; a < l; ++a) for (int c = 0; c < a; ++c) C[a*l+c] = C[c*l+a];
    double* V = (double*)malloc(sizeof(double) * l * l);
    double* w = (double*)malloc(sizeof(double) * l);
    jacobi_eigh(C, V, w, l);
    for (int j = 0; j < l; ++j) { double s = w[j] > 0 ? sqrt(w[j]) : 0.0; if (sv) sv[j] = s; }
    for (int c = 0; c < k; ++c) {
        double s = w[c] > 0 ? sqrt(w[c]) : 0.0;
        eigenvalues[c] = w[c] / (double)(N - 1);
        double best = 0.0; int sgn = 1;
        for (int64_t n = 0; n < N; ++n) {
            double a = 0; for (int j = 0; j < l; ++j) a += (double)Q[n*l+j] * V[j*l+c];
            a *= s; scores[n * k + c] = a;
            if (fabs(a) > best) { best = fabs(a); sgn = a < 0 ? -1 : 1; }
        }
        if (sgn < 0) for (int64_t n = 0; n < N; ++n) scores[n * k + c] = -scores[n * k + c];
        double inv = s > 0 ? 1.0 / s : 0.0;
        for (int64_t i = 0; i < M; ++i) {
            double a = 0; for (int j = 0; j < l; ++j) a += (double)T[i*l+j] * V[j*l+c];
            loadings[i * k + c] = sgn * a * inv;
        }
    }
    free(C); free(V); free(w); free(T); free(Q); free(Y);
    return 0;
}

int orc_sizeof_real(void) { return (int)sizeof(real); }
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
