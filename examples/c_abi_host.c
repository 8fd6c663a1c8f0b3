/* A plain C99 client of libgpca.so's C ABI (include/gpca.h): what a non-Python, non-C++ host links against.
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_abi_host.c -Lgenomic_pca_amd -lgpca -Wl,-rpath,$PWD/genomic_pca_amd -lm
 * Without a GPU it exercises the host-only entry points and shows that gpca_create fails loudly (no CPU fallback); with
 * `./a.out gpu` it runs the whole path on a small synthetic matrix: upload -> QC -> randomized PCA -> results, plus the same
 * matrix out of core through a host panel callback. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpca.h"

typedef struct { const int8_t* g; int64_t n; int calls; } rows_t;

static int fill_rows(void* user, int64_t row0, int64_t rows, void* dst, int64_t ld) {
    rows_t* r = (rows_t*)user;
    int64_t i;
    for (i = 0; i < rows; ++i) memcpy((int8_t*)dst + i * ld, r->g + (row0 + i) * r->n, (size_t)r->n);
    r->calls++;
    return 0;
}

static int run_gpu(void) {
    enum { M = 4096, N = 384, K = 5 };
    gpca_config cfg;
    gpca_qc_config qc = {0.98, 0.01, 1e-6};
    gpca_handle* h = NULL;
    gpca_panel_source src;
    rows_t rows;
    int8_t* g = (int8_t*)malloc((size_t)M * N);
    double ev[K], ev2[K];
    float* scores = (float*)malloc(sizeof(float) * N * K);
    unsigned s = 12345u;
    int64_t i, n;
    int rc, c;
    int32_t cached = 0;
    double ev0[K];
    int32_t store = -1, prec = -1;
    memset(&cfg, 0, sizeof cfg);                    /* a zeroed config = the fast exact path, residency chosen by the library */
    cfg.device = -1;
    for (i = 0; i < M; ++i) {                       /* two populations with different allele frequencies */
        double p0, p1;
        s = s * 1664525u + 1013904223u; p0 = 0.1 + 0.4 * (s >> 8) / 16777216.0;
        s = s * 1664525u + 1013904223u; p1 = 0.1 + 0.4 * (s >> 8) / 16777216.0;
        for (n = 0; n < N; ++n) {
            const double p = (n & 1) ? p1 : p0;
            int a, b;
            s = s * 1664525u + 1013904223u; a = (s >> 8) / 16777216.0 < p;
            s = s * 1664525u + 1013904223u; b = (s >> 8) / 16777216.0 < p;
            g[i * N + n] = (int8_t)(a + b);
        }
    }
    if ((rc = gpca_create(&cfg, &h)) != GPCA_OK) { fprintf(stderr, "gpca_create: %s\n", gpca_last_error(NULL)); return 1; }
    if ((rc = gpca_upload_genotypes_i8(h, g, M, N, N)) != GPCA_OK || (rc = gpca_snp_stats(h, &qc, NULL, NULL, NULL)) != GPCA_OK ||
        (rc = gpca_rsvd(h, K, 10, 2, 7)) != GPCA_OK || (rc = gpca_get_eigenvalues(h, ev)) != GPCA_OK ||
        (rc = gpca_get_scores(h, scores)) != GPCA_OK) {
        fprintf(stderr, "resident path failed: [%d] %s\n", rc, gpca_last_error(h)); return 1;
    }
    printf("resident : %lld PCA SNPs, eigenvalues %.6f %.6f %.6f\n", (long long)gpca_num_pca_snps(h), ev[0], ev[1], ev[2]);
    /* {0} selected GPCA_PREC_I8_EXACT, and GPCA_STORE_AUTO resolved to int8 rows for 384 samples (2-bit codes start at 1 024): the same
     * bits as the explicit choice */
    if (gpca_get_storage(h, &store, &prec) != GPCA_OK || store != GPCA_STORE_INT8 || prec != GPCA_PREC_I8_EXACT) { fprintf(stderr, "default config: storage %d precision %d\n", (int)store, (int)prec); return 1; }
    {
        gpca_handle* h2 = NULL;
        gpca_config ex;
        memset(&ex, 0, sizeof ex);
        ex.device = -1; ex.precision = GPCA_PREC_I8_EXACT; ex.storage = GPCA_STORE_INT8;
        if (gpca_create(&ex, &h2) != GPCA_OK || gpca_upload_genotypes_i8(h2, g, M, N, N) != GPCA_OK || gpca_snp_stats(h2, &qc, NULL, NULL, NULL) != GPCA_OK ||
            gpca_rsvd(h2, K, 10, 2, 7) != GPCA_OK || gpca_get_eigenvalues(h2, ev0) != GPCA_OK) { fprintf(stderr, "explicit config failed: %s\n", gpca_last_error(h2)); return 1; }
        gpca_destroy(h2);
        for (c = 0; c < K; ++c) if (ev[c] != ev0[c]) { fprintf(stderr, "default config != explicit exact path\n"); return 1; }
        puts("default  : gpca_config {0} = GPCA_PREC_I8_EXACT, GPCA_STORE_AUTO -> int8 rows here; same bits as the explicit exact path");
    }
    memset(&src, 0, sizeof src);
    rows.g = g; rows.n = N; rows.calls = 0;
    src.kind = GPCA_PANEL_HOST_I8; src.fill = fill_rows; src.user = &rows;
    if ((rc = gpca_stream_open(h, &src, M, N, 1024, 2)) != GPCA_OK || (rc = gpca_stream_set_fused(h, 0)) != GPCA_OK ||
        (rc = gpca_snp_stats(h, &qc, NULL, NULL, NULL)) != GPCA_OK || (rc = gpca_rsvd(h, K, 10, 2, 7)) != GPCA_OK ||
        (rc = gpca_get_eigenvalues(h, ev2)) != GPCA_OK) {
        fprintf(stderr, "streamed path failed: [%d] %s\n", rc, gpca_last_error(h)); return 1;
    }
    printf("streamed : %d panel fills, eigenvalues %.6f %.6f %.6f\n", rows.calls, ev2[0], ev2[1], ev2[2]);
    for (c = 0; c < K; ++c) if (ev[c] != ev2[c]) { fprintf(stderr, "streamed != resident\n"); return 1; }
    /* spare HBM keeps the panels: after one more pass the callback is not asked again, and the bits stay the same */
    if ((rc = gpca_stream_set_cache(h, -1, &cached)) != GPCA_OK || (rc = gpca_rsvd(h, K, 10, 2, 7)) != GPCA_OK) {
        fprintf(stderr, "panel cache failed: [%d] %s\n", rc, gpca_last_error(h)); return 1;
    }
    rows.calls = 0;
    if ((rc = gpca_rsvd(h, K, 10, 2, 7)) != GPCA_OK || (rc = gpca_get_eigenvalues(h, ev2)) != GPCA_OK) return 1;
    printf("cached   : %d panels in HBM, %d panel fills, eigenvalues %.6f %.6f %.6f\n", (int)cached, rows.calls, ev2[0], ev2[1], ev2[2]);
    for (c = 0; c < K; ++c) if (ev[c] != ev2[c] || rows.calls != 0) { fprintf(stderr, "cached != resident\n"); return 1; }
    if (!(ev[0] > 4.0 * ev[1])) { fprintf(stderr, "expected one structured eigenvalue\n"); return 1; }
    gpca_destroy(h);
    free(g); free(scores);
    puts("ok");
    return 0;
}

int main(int argc, char** argv) {
    double a[9] = {4, 1, 0, 1, 3, 1, 0, 1, 2}, w[3], v[9];
    gpca_handle* h = NULL;
    int rc;
    printf("libgpca version %d; status(-5) = \"%s\"\n", gpca_version(), gpca_status_string(GPCA_ERR_MISSING_GENOTYPE));
    if (gpca_version() != GPCA_VERSION) return 2;
    if (gpca_host_eigh_desc(a, 3, w, v) != GPCA_OK || fabs(w[0] + w[1] + w[2] - 9.0) > 1e-12 || !(w[0] >= w[1] && w[1] >= w[2])) return 3;
    if (fabs(gpca_hwe_chi_squared_p_value(25, 50, 25) - 1.0) > 0.0) return 4;          /* exact HWE proportions: p = 1 */
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) return run_gpu();
    rc = gpca_create(NULL, &h);
    if (rc == GPCA_OK) { gpca_destroy(h); puts("a GPU is present: run with `gpu` for the full path"); return 0; }
    printf("gpca_create without a GPU: [%d] %s\n", rc, gpca_last_error(NULL));
    return rc == GPCA_ERR_NO_DEVICE ? 0 : 5;
}
