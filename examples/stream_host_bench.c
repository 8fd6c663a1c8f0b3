/* Out-of-core streaming from HOST memory through the C ABI (include/gpca.h): how close to the host link does a randomized PCA
 * run when the matrix never fits the device?  BASELINE.json configs[4]'s mode with the source a real host would have -- a PLINK
 * .bed payload in RAM / the page cache (what the reference's bed_reader opens, prepare.rs:622-629, and re-reads strip by strip on
 * every pass, main.rs:322, prepare.rs:682-693) -- instead of the device generator bench.py --streamed uses.
 *
 *   gcc -std=c99 -O2 -Wall -Wextra -pedantic -pthread -Iinclude examples/stream_host_bench.c -Lgenomic_pca_amd -lgpca \
 *       -Wl,-rpath,$PWD/genomic_pca_amd -lm -o examples/stream_host_bench
 *   ./examples/stream_host_bench [M = 2000000] [N = 65536] [steps = 2] [k = 20]
 *
 * A valid 2-bit matrix (no missing codes) of M x N genotypes is built in host memory, then the same gpca_snp_stats + gpca_rsvd job
 * runs over it four ways and one JSON line per way reports the link rate = bytes the passes moved / wall time of gpca_rsvd:
 *   callback_1thread : GPCA_PANEL_HOST_BED, the callback memcpy's the panel on the library's worker thread (one core)
 *   callback_mt      : the same callback splitting the copy over pthreads
 *   mapped_staged    : GPCA_PANEL_MAPPED_BED, the library's own copy threads fill the pinned staging ring
 *   mapped_registered: GPCA_PANEL_MAPPED_BED + GPCA_SOURCE_REGISTER, pages locked once, DMA straight from the matrix (no staging)
 * and checks that all four return the same eigenvalues bit for bit. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gpca.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- the matrix: every byte holds four valid codes (00 = 2, 10 = 1, 11 = 0 copies of A1; 01 = missing is never produced) */
typedef struct { uint8_t* base; int64_t bpr, row0, row1; uint64_t seed; } gen_job;
static uint8_t g_valid[81];
static void* gen_rows(void* p) {
    gen_job* j = (gen_job*)p;
    int64_t r, c;
    for (r = j->row0; r < j->row1; ++r) {
        /* per-row allele frequency so that rows differ in variance; xorshift64* stream per row */
        uint64_t s = (j->seed + (uint64_t)r) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
        uint8_t* row = j->base + r * j->bpr;
        unsigned bias;
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        bias = (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 58);        /* 0..63 */
        for (c = 0; c < j->bpr; c += 8) {
            int b;
            uint64_t v;
            s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
            v = s * 0x2545F4914F6CDD1Dull;
            for (b = 0; b < 8 && c + b < j->bpr; ++b) {
                /* four base-3 digits from 8 bits, skewed by the row's bias towards code 11 (dosage 0) */
                unsigned x = (unsigned)(v >> (8 * b)) & 255u;
                unsigned d0 = x % 3u, d1 = (x / 3u) % 3u, d2 = (x / 9u) % 3u, d3 = (x / 27u) % 3u;
                if ((x & 63u) < bias) d0 = 2u;
                row[c + b] = g_valid[d0 + 3u * d1 + 9u * d2 + 27u * d3];
            }
        }
    }
    return NULL;
}

/* ---- callbacks */
typedef struct { const uint8_t* base; int64_t bpr; int threads; long calls; } src_t;
typedef struct { uint8_t* dst; const uint8_t* src; size_t bytes; } cp_job;
static void* cp_run(void* p) { cp_job* j = (cp_job*)p; memcpy(j->dst, j->src, j->bytes); return NULL; }
static int fill_rows(void* user, int64_t row0, int64_t rows, void* dst, int64_t ld) {
    src_t* s = (src_t*)user;
    const size_t total = (size_t)rows * (size_t)ld;
    const uint8_t* from = s->base + (size_t)row0 * (size_t)s->bpr;
    int t, T = s->threads;
    pthread_t th[64];
    cp_job jobs[64];
    s->calls++;
    if (ld != s->bpr) return 1;
    if (T <= 1) { memcpy(dst, from, total); return 0; }
    for (t = 0; t < T; ++t) {
        const size_t a = total / (size_t)T * (size_t)t, b = t == T - 1 ? total : total / (size_t)T * (size_t)(t + 1);
        jobs[t].dst = (uint8_t*)dst + a; jobs[t].src = from + a; jobs[t].bytes = b - a;
        if (t > 0) pthread_create(&th[t], NULL, cp_run, &jobs[t]);
    }
    cp_run(&jobs[0]);
    for (t = 1; t < T; ++t) pthread_join(th[t], NULL);
    return 0;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != GPCA_OK) { fprintf(stderr, "%s: [%d] %s\n", #call, rc_, gpca_last_error(h)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 2000000, N = argc > 2 ? atoll(argv[2]) : 65536;
    const int steps = argc > 3 ? atoi(argv[3]) : 2, K = argc > 4 ? atoi(argv[4]) : 20;
    const int64_t bpr = (N + 3) / 4;
    const size_t bytes = (size_t)M * (size_t)bpr;
    const int GT = 16;
    const char* names[4] = {"callback_1thread", "callback_mt", "mapped_staged", "mapped_registered"};
    double ev_ref[64];
    uint8_t* bed;
    int mode, i, t;
    double t0;
    {
        int a, b, c, d;   /* dosage digit 0/1/2 -> code 11/10/00 */
        static const unsigned code[3] = {3u, 2u, 0u};
        for (a = 0; a < 3; ++a) for (b = 0; b < 3; ++b) for (c = 0; c < 3; ++c) for (d = 0; d < 3; ++d)
            g_valid[a + 3 * b + 9 * c + 27 * d] = (uint8_t)(code[2 - a] | (code[2 - b] << 2) | (code[2 - c] << 4) | (code[2 - d] << 6));
    }
    if (K < 1 || K > 54 || steps < 1) { fprintf(stderr, "bad arguments\n"); return 2; }
    bed = (uint8_t*)malloc(bytes);
    if (!bed) { fprintf(stderr, "cannot allocate %.1f GB of host memory\n", (double)bytes / 1e9); return 2; }
    t0 = now_s();
    {
        pthread_t th[16];
        gen_job jobs[16];
        for (t = 0; t < GT; ++t) {
            jobs[t].base = bed; jobs[t].bpr = bpr; jobs[t].row0 = M * t / GT; jobs[t].row1 = M * (t + 1) / GT; jobs[t].seed = 2025;
            pthread_create(&th[t], NULL, gen_rows, &jobs[t]);
        }
        for (t = 0; t < GT; ++t) pthread_join(th[t], NULL);
    }
    fprintf(stderr, "matrix: %lld SNPs x %lld samples = %.2f GB of 2-bit rows in host memory, generated in %.1f s\n", (long long)M, (long long)N,
            (double)bytes / 1e9, now_s() - t0);

    for (mode = 0; mode < 4; ++mode) {
        gpca_config cfg;
        gpca_qc_config qc = {0.0, 0.0, 1.0};
        gpca_handle* h = NULL;
        gpca_panel_source src;
        gpca_stream_info info;
        src_t user;
        double ev[64], t_open, t_stats, t_rsvd, passes;
        memset(&cfg, 0, sizeof cfg);
        cfg.device = -1; cfg.precision = GPCA_PREC_I8_EXACT; cfg.storage = GPCA_STORE_2BIT;
        memset(&src, 0, sizeof src);
        user.base = bed; user.bpr = bpr; user.calls = 0; user.threads = mode == 1 ? 8 : 1;
        if (mode < 2) { src.kind = GPCA_PANEL_HOST_BED; src.fill = fill_rows; src.user = &user; }
        else { src.kind = GPCA_PANEL_MAPPED_BED; src.user = bed; src.host_ld = bpr; src.flags = mode == 3 ? GPCA_SOURCE_REGISTER : 0; }
        if (gpca_create(&cfg, &h) != GPCA_OK) { fprintf(stderr, "gpca_create: %s\n", gpca_last_error(NULL)); return 1; }
        t0 = now_s();
        CHECK(gpca_stream_open(h, &src, M, N, 0, 3));      /* fused (4 passes per call at q = 2), no panel cache: every pass crosses the link */
        t_open = now_s() - t0;
        t0 = now_s();
        CHECK(gpca_snp_stats(h, &qc, NULL, NULL, NULL));
        t_stats = now_s() - t0;
        CHECK(gpca_rsvd(h, K, 10, 2, 1));                   /* warm-up: workspace allocation */
        t0 = now_s();
        for (i = 0; i < steps; ++i) CHECK(gpca_rsvd(h, K, 10, 2, 1));
        t_rsvd = (now_s() - t0) / steps;
        CHECK(gpca_get_eigenvalues(h, ev));
        CHECK(gpca_stream_get_info(h, &info));
        passes = 4.0;
        printf("{\"mode\": \"%s\", \"snps\": %lld, \"samples\": %lld, \"k\": %d, \"bed_GB\": %.3f, \"open_s\": %.3f, \"register_s\": %.3f, "
               "\"snp_stats_s\": %.3f, \"snp_stats_link_GBs\": %.2f, \"rsvd_s\": %.3f, \"passes_per_rsvd\": %.0f, \"rsvd_link_GBs\": %.2f, "
               "\"snps_x_samples_per_s\": %.4e, \"panels\": %d, \"panel_rows\": %lld, \"staging_buffers\": %d, \"zero_staging\": %d, "
               "\"copy_threads\": %d, \"fills\": %lld, \"fill_host_ms_per_panel\": %.2f, \"fill_wait_ms_total\": %.1f, \"eigenvalues\": [%.9g, %.9g, %.9g]}\n",
               names[mode], (long long)M, (long long)N, K, (double)bytes / 1e9, t_open, info.register_ms * 1e-3, t_stats, (double)bytes / 1e9 / t_stats,
               t_rsvd, passes, passes * (double)bytes / 1e9 / t_rsvd, (double)M * (double)N / t_rsvd, (int)info.n_panels, (long long)info.panel_rows,
               (int)info.staging_buffers, (int)info.zero_staging, (int)info.copy_threads, (long long)info.fills,
               info.fills ? info.fill_host_ms / (double)info.fills : 0.0, info.fill_wait_ms, ev[0], ev[1], ev[2]);
        fflush(stdout);
        if (mode == 0) memcpy(ev_ref, ev, sizeof(double) * (size_t)K);
        else for (i = 0; i < K; ++i) if (ev[i] != ev_ref[i]) { fprintf(stderr, "mode %s: eigenvalue %d differs from the callback run\n", names[mode], i); return 1; }
        gpca_destroy(h);
    }
    free(bed);
    return 0;
}
