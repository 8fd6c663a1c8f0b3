/*
 * gpca.h -- C ABI of the MI355X-native randomized-PCA engine (libgpca.so).
 *
 * Drop-in boundary for ONE hot path of SauersML/genomic_pca: per-SNP standardisation fused into
 * the randomized-SVD core.  Plain C types only: opaque handle, caller-owned host buffers,
 * integer status codes, no exceptions, no C++/torch types.  A Rust host binds these with a
 * 30-line `extern "C"` block.
 *
 * The reference's seam is a PULL model (the solver calls the accessor for f32 blocks):
 *   trait PcaReadyGenotypeAccessor          /root/reference/src/prepare.rs:1838-2030
 *   PCA::new / rfit / transform             /root/reference/src/main.rs:602,648-660
 *   EigenSNPCoreAlgorithm::compute_pca      /root/reference/src/main.rs:359-366
 * This ABI is a PUSH model: genotypes are uploaded once (1 B or 0.25 B per genotype), stay in
 * HBM, and every pass over them runs on the device; the pull API survives as
 * gpca_standardize_block() for boundary parity.
 *
 * Threading: one handle = one GPU.  Every entry point takes the handle's (recursive) lock, so a handle may be shared
 * between host threads the way the reference shares its `Clone + Send + Sync` accessor between rayon workers
 * (prepare.rs:1770-1779, 1838): calls are serialised per handle, different handles run concurrently (one host thread may
 * drive handles on several GPUs: every entry point makes the handle's device the calling thread's current HIP device).
 * The pull API is the exception, as in the reference, whose accessor is served by 1-16 actor threads in parallel
 * (main.rs:279-283): gpca_standardize_block holds the lock only while it checks its ids, then runs its copies and its
 * kernel on one of up to 16 lanes (a stream and scratch of its own) -- calls from different threads overlap; every other
 * entry point first waits until no pull is in flight, so the matrix and its statistics never change under one.  gpca_destroy
 * must not race with other calls on the same handle; gpca_last_error() returns the calling thread's own last failure on
 * the handle (or, for a thread that has not failed on it, the handle's last message).  Functions return GPCA_OK (0) or a
 * negative gpca_status.
 *
 * Limits: k + oversample <= 128 sketch columns on GPCA_PREC_I8_EXACT (<= 64 on GPCA_PREC_F32_MFMA and in the EigenSNP stage calls: the
 * reference adds 10 to any k <= min(samples, variants), main.rs:621-628, 636); GPCA_PREC_I8_EXACT holds up to 2^22 (4 194 304) samples per matrix (i32
 * accumulators; GPCA_PREC_F32_MFMA has no such bound); SNP rows per handle are bounded by device memory only (64M rows x 1 000
 * samples and 10M x 100k as 2-bit codes were run on one MI355X), the bit-for-bit guarantees between partitions of the same
 * matrix (streamed = resident, any kernel variant) hold up to 2^25 (33.5M) rows per handle, where integer sums stay below 2^53.
 */
#ifndef GPCA_H
#define GPCA_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define GPCA_API __attribute__((visibility("default")))
#else
#define GPCA_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GPCA_VERSION 250 /* 0.2.5: a zeroed gpca_config is the fast exact path with automatic residency (the enum values of gpca_precision /
                            gpca_storage changed: 0 now means "the library's choice"); the l x l eigen step runs on the device */
#define GPCA_MISSING_I8 (-127) /* bed_reader i8 missing code, prepare.rs:1224 */

typedef struct gpca_handle gpca_handle;

typedef enum gpca_status {
    GPCA_OK = 0,
    GPCA_ERR_BAD_ARG = -1,
    GPCA_ERR_OOM = -2,
    GPCA_ERR_HIP = -3,
    GPCA_ERR_RCCL = -4,
    /* replaces the hard error of prepare.rs:1909-1911, 1961-1963, 2008-2009 */
    GPCA_ERR_MISSING_GENOTYPE = -5,
    GPCA_ERR_NOT_CONVERGED = -6, /* sketch lost rank (CholQR pivot <= 0) */
    GPCA_ERR_STATE = -7,         /* call order: e.g. rsvd before stats */
    GPCA_ERR_NO_DEVICE = -8,
    GPCA_ERR_INVALID_GENOTYPE = -9 /* a kept SNP holds a value outside {0,1,2} */
} gpca_status;

/* Arithmetic used by the two tall-skinny products.  0 -- what `gpca_config cfg = {0}`, a Rust `GpcaConfig::default()` or
 * gpca_create(NULL) give -- is the library's choice: the exact-integer path, the fastest parity-green one (a randomized PCA of 1M x 10k
 * in 10 ms against 28 ms on the f32 matrix cores). */
typedef enum gpca_precision {
    GPCA_PREC_DEFAULT = 0,  /* = GPCA_PREC_I8_EXACT */
    GPCA_PREC_I8_EXACT = 1, /* v_mfma_i32_32x32x32_i8 on fixed-point digits of the skinny operand, exact integer accumulation */
    GPCA_PREC_F32_MFMA = 2  /* v_mfma_f32_32x32x2_f32, exact f32 FMA chains (north_star's MFMA-fp32 path) */
} gpca_precision;

/* How the genotypes stay resident in HBM.  0 = GPCA_STORE_AUTO: decided when the genotypes arrive, the way both command lines decide it
 * (main.rs has no such choice: its matrix is f64): rows of >= 1 024 samples stay as 2-bit codes -- a quarter of the HBM and the faster
 * kernels there -- narrower ones as int8 (2-bit rows pad to 1 024 samples, int8 rows to 256).  An int8 upload that turns out to hold a
 * value outside {0, 1, 2, -127} is kept as int8 (2-bit codes could only store it as "missing"); panel streams cannot look ahead and
 * follow the sample count alone.  gpca_get_storage reports what was chosen. */
typedef enum gpca_storage {
    GPCA_STORE_AUTO = 0,
    GPCA_STORE_2BIT = 1, /* 0.25 B per genotype (PLINK-like packing of dosage codes), decoded in the GEMM prologues of either
                            precision.  10M SNPs x 100k samples = 250 GB: fits one MI355X. */
    GPCA_STORE_INT8 = 2  /* 1 B per genotype */
} gpca_storage;

typedef struct gpca_config {
    int32_t device;    /* HIP ordinal; -1 = current device */
    int32_t precision; /* gpca_precision */
    int32_t storage;   /* gpca_storage */
    int32_t digit_planes; /* GPCA_PREC_I8_EXACT only (3 needs an explicit GPCA_STORE_2BIT).  4: four signed base-128 digit planes of the skinny operand (28-bit fixed point per
                             column).  3: three signed base-256 planes (24-bit; exact integer accumulation as before) -- a quarter
                             less matrix-core work; implemented for GPCA_STORE_2BIT, whose kernels are matrix-core bound.
                             0 = the library's choice: 4 on int8 rows (HBM-bound), 3 on 2-bit rows (measured max|dPC| <= 3e-7 against
                             the f64 checker on every parity shape, tighter than GPCA_PREC_F32_MFMA). */
    int32_t reserved[4];  /* [0] = GPCA_CFG_* flags below (0 = the defaults); [1], [2] = resident-wave targets of the two GEMM grids (0 = the
                             values tuned on MI355X: 1 024 / 2 048; smaller values give small grids -- how the tests reach every round and
                             task pattern on small matrices); [3] must be 0.  Which kernels run is decided here, by the caller: the library
                             reads no environment variable for it. */
} gpca_config;
/* gpca_config.reserved[0] */
#define GPCA_CFG_SIMPLE_KERNELS 1 /* GPCA_PREC_I8_EXACT: the register-only reference kernels (no LDS staging, no DMA, compiler-counted waits)
                                     instead of the LDS-DMA ones: same integers, same pinned roundings -- the same bits, more slowly */
#define GPCA_CFG_NO_COMPACT 2     /* never gather the kept rows into a matrix of their own when QC drops most of them */
#define GPCA_CFG_NO_NARROW 4      /* matrices of <= 256 samples on the wide kernels (rows padded to 256 samples) */
#define GPCA_CFG_NO_SPIN_SYNC 8   /* wait for the device with hipStreamSynchronize instead of a busy-polled stream (one host core less, ~0.1 ms per call more) */
#define GPCA_CFG_ALL 15

/* SNP QC thresholds = MicroarrayDataPreparerConfig, main.rs:302-309 / prepare.rs:1281-1311,1363.
 * Effective reference defaults (clap, main.rs:545-560): 0.98 / 0.01 / 1e-6.
 * "no filtering" = {0, 0, 1.0}: only the nv==0, monomorphic (1e-9) and variance (1e-9) guards act. */
typedef struct gpca_qc_config {
    double min_snp_call_rate;   /* drop if n_valid/N <  this         prepare.rs:1283-1284 */
    double min_snp_maf;         /* drop if min(p,1-p) < this         prepare.rs:1296-1299 */
    double max_snp_hwe_p_value; /* if < 1: drop if HWE p <= this     prepare.rs:1306-1311 */
} gpca_qc_config;

/* ---- lifecycle ------------------------------------------------------------------------- */
GPCA_API int gpca_version(void);
GPCA_API const char* gpca_status_string(int status);
GPCA_API int gpca_create(const gpca_config* cfg, gpca_handle** out);
GPCA_API int gpca_destroy(gpca_handle* h);
GPCA_API const char* gpca_last_error(gpca_handle* h);

/* ---- genotype residency (replaces IoService + bed_reader reads, prepare.rs:622-629,682-693,
 *      and build_matrix's N x M f64, vcf.rs:317-345) ------------------------------------- */
/* SNP-major int8 dosages (count of allele 1: 0/1/2, -127 missing); row i at snp_major + i*ld. */
GPCA_API int gpca_upload_genotypes_i8(gpca_handle* h, const int8_t* snp_major, int64_t M, int64_t N, int64_t ld);
/* PLINK .bed payload after the 3-byte magic: M rows of ceil(N/4) bytes, 2 bits/sample LSB-first;
 * decoded on the device with count_a1 semantics (00->2, 10->1, 11->0, 01->missing). */
GPCA_API int gpca_upload_bed2bit(gpca_handle* h, const uint8_t* bed_rows, int64_t M, int64_t N);
/* Synthetic workload of SURVEY.md 8(d), generated on the device (bit-identical to
 * oracle/gpca_oracle.c:orc_synth_genotypes).  thresh: host uint32 [M][P] = floor(p*2^32). */
GPCA_API int gpca_synth_genotypes(gpca_handle* h, int64_t M, int64_t N, uint64_t seed, const uint32_t* thresh,
                         int32_t P, int64_t snp_offset);
GPCA_API int gpca_download_genotypes_i8(gpca_handle* h, int8_t* out, int64_t ld);

/* ---- g: panel sources and out-of-core streaming (BASELINE.json configs[4]; the reference never holds the matrix either:
 *      its solver pulls snp_processing_strip_size-row strips through the accessor, main.rs:322,584, prepare.rs:1839-2022) */
typedef enum gpca_panel_kind {
    GPCA_PANEL_HOST_I8 = 0,  /* `fill` writes int8 SNP-major rows (0/1/2, -127 missing), row pitch ld = N */
    GPCA_PANEL_HOST_BED = 1, /* `fill` writes PLINK .bed rows (2 bits/sample, count_a1 decode), row pitch ld = ceil(N/4) */
    GPCA_PANEL_SYNTH = 2,    /* device generator of gpca_synth_genotypes: thresh = uint32 [M][n_pop] = floor(p * 2^32) */
    GPCA_PANEL_SYNTH16 = 3,  /* fast device generator, one 16-bit uniform per genotype (SplitMix64 in counter mode): thresh = uint32 [M][n_pop],
                                high half = floor(P(g >= 1) * 65536), low half = floor(P(g = 2) * 65536); sample n belongs to
                                population (n / 16) % n_pop */
    /* The whole matrix sits in the caller's address space (malloc'ed, or a memory-mapped .bed payload: what bed_reader opens at
     * prepare.rs:622-629): `user` = address of row 0, `host_ld` = row pitch in bytes (0 = tight: N, or ceil(N/4)).  No callback:
     * the library's own copy threads move the rows of a panel into its pinned staging ring, or -- GPCA_SOURCE_REGISTER -- the
     * mapping is page-locked once at open and every panel is DMA-ed straight out of it (no staging copy at all).  The memory must
     * stay valid and unchanged until the stream is closed (gpca_stream_open) or the call returns (gpca_load_from_source). */
    GPCA_PANEL_MAPPED_I8 = 4,
    GPCA_PANEL_MAPPED_BED = 5
} gpca_panel_kind;
/* gpca_panel_source.flags */
#define GPCA_SOURCE_REGISTER 1 /* MAPPED_*: hipHostRegister the mapping at open (zero staging); if the pages cannot be locked the
                                  staged path is used instead -- gpca_stream_get_info reports which */
/* Write rows [row0, row0 + rows) of the matrix into dst (pinned host memory owned by the library).  Return 0, or
 * non-zero to abort the pass (reported as GPCA_ERR_BAD_ARG with the row range in the message).
 * Threading: called from ONE library-owned worker thread per source (never concurrently with itself), rows ascending within a
 * pass, up to staging_buffers - 1 panels ahead of the panel being copied to the device -- so the host copy of panel p + 2
 * overlaps the H2D copy of p + 1 and the GEMMs of p.  It runs while a pass (gpca_snp_stats / gpca_rsvd / gpca_transform /
 * gpca_load_from_source) is in progress on another thread, which holds the handle's lock: it may block on I/O, it must not
 * call into the same handle.  Every panel is asked exactly once per pass (cached panels once in all); after a failure no
 * further panel is asked in that pass. */
#define GPCA_SOURCE_BENCH_HOLD 2 /* SYNTH16, MEASUREMENT ONLY: a ring / cache buffer that already holds a generated panel of the same height is not
                                   generated again -- every pass then multiplies whatever panels the buffers held first.  The numbers a call
                                   returns are NOT a PCA of the source; what it measures is the engine's rate over streamed panels with the
                                   generator out of the way (bench.py: config5_per_gpu_shard_streamed.engine_rate_with_fills_hidden). */
typedef int (*gpca_panel_fn)(void* user, int64_t row0, int64_t rows, void* dst, int64_t ld);
typedef struct gpca_panel_source {
    int32_t kind;           /* gpca_panel_kind */
    int32_t n_pop;          /* SYNTH*: populations (columns of thresh) */
    gpca_panel_fn fill;     /* HOST_*: the callback above */
    void* user;             /* HOST_*: passed to fill; MAPPED_*: address of row 0 */
    const uint32_t* thresh; /* SYNTH*: host table, copied to the device at open */
    uint64_t seed;          /* SYNTH* */
    int64_t snp_offset;     /* SYNTH*: global index of row 0 (row shards of one matrix draw the rows they would unsharded) */
    int64_t host_ld;        /* MAPPED_*: row pitch in bytes (0 = tight) */
    int64_t flags;          /* GPCA_SOURCE_* */
} gpca_panel_source;
/* Resident load through a panel source (chunked through bounded staging: a 250 GB .bed needs no second device copy). */
GPCA_API int gpca_load_from_source(gpca_handle* h, const gpca_panel_source* src, int64_t M, int64_t N);
/* Out-of-core mode: the matrix is never resident.  Every pass of gpca_snp_stats / gpca_rsvd / gpca_transform walks
 * ceil(M / panel_rows) panels through a ring of `ring_slots` (>= 2) HBM panel buffers; panel p + 1 is generated or
 * copied on a second stream while panel p is multiplied.  panel_rows is rounded up to a multiple of 128; 0 picks
 * 131 072 rows (a full grid of the row-parallel GEMM) or as many as fit the ring in half of the free HBM (host sources: at most
 * 2 GiB of pinned host staging per panel, three staging panels -- GPCA_STAGE_BUFFERS=2..8).  Requires GPCA_PREC_I8_EXACT (either storage).  With gpca_stream_set_fused(h, 0) the
 * results are bit-identical to the resident engine on the same matrix; the default (fused) form is described below.  The pull API
 * (gpca_standardize_block) and gpca_download_genotypes_i8 need a resident matrix. */
GPCA_API int gpca_stream_open(gpca_handle* h, const gpca_panel_source* src, int64_t M, int64_t N, int64_t panel_rows,
                              int32_t ring_slots);
/* fused = 1 (the default after gpca_stream_open): a power iteration reads every panel ONCE -- G Q, quantisation and G^T T per panel
 * while it sits in HBM -- so gpca_rsvd makes 2 + power_iters passes over the source instead of 2 + 2 * power_iters (4 instead of 6
 * at q = 2: what counts when the source is a disk or the host link).  Every panel then quantises its rows against its own column
 * maxima: results differ from the resident engine at the 1e-9 level of the 28-bit fixed point (as a row-sharded run does).
 * fused = 0: the 6-pass form, bit-identical to the resident engine. */
GPCA_API int gpca_stream_set_fused(gpca_handle* h, int32_t fused);
/* Panel cache: HBM the ring and the solver's workspace leave free holds the LEADING panels for good -- each is asked of the
 * source once (normally during gpca_snp_stats) and read in place on every later pass, so a matrix 3x the HBM costs 2/3 of the
 * source traffic per pass, and one that fits is read once.  max_bytes: upper bound for the cache (whole panels are taken);
 * < 0 = all free device memory less the solver's workspace estimate and a 4 GiB margin; 0 drops the cache.  The source must
 * return the same rows every time it is asked (it must anyway: every pass re-reads it).  Results do not change.
 * *n_cached (may be NULL) receives the number of cached panels.  GPCA_ERR_OOM keeps the panels allocated so far. */
GPCA_API int gpca_stream_set_cache(gpca_handle* h, int64_t max_bytes, int32_t* n_cached);
/* What the open panel stream looks like, and where its time went since gpca_stream_open (host-side, always collected). */
typedef struct gpca_stream_info {
    int64_t panel_rows;
    int32_t n_panels, ring_slots, n_cached;
    int32_t staging_buffers; /* pinned host staging panels (0: device generator, or zero-staging) */
    int32_t zero_staging;    /* 1 = MAPPED_* source page-locked in place, panels DMA-ed straight from the caller's memory */
    int32_t copy_threads;    /* MAPPED_* staged: threads that copy a panel into staging */
    int64_t fills;           /* panels asked of the source so far */
    double fill_host_ms;     /* host time spent producing panels into staging (callback / copy threads), summed over the worker's jobs */
    double fill_wait_ms;     /* time the pass thread was blocked on the staging ring (panel not staged yet, or every buffer still in flight on
                                the link): compare with fill_host_ms to tell a slow source from a saturated link */
    double register_ms;      /* hipHostRegister at open (zero-staging) */
    int64_t reserved[4];
} gpca_stream_info;
GPCA_API int gpca_stream_get_info(gpca_handle* h, gpca_stream_info* out);
GPCA_API int gpca_dims(gpca_handle* h, int64_t* M, int64_t* N);
/* The residency in use (GPCA_STORE_INT8 / GPCA_STORE_2BIT; GPCA_STORE_AUTO until the first genotypes have arrived) and the precision. */
GPCA_API int gpca_get_storage(gpca_handle* h, int32_t* storage, int32_t* precision);
/* Free and total memory of the handle's device in bytes (hipMemGetInfo): what a host needs to choose between a resident load
 * (M x N bytes int8, M x N / 4 packed, plus about 1 KiB per SNP row and 8 KiB per sample of workspace) and gpca_stream_open. */
GPCA_API int gpca_get_device_memory(gpca_handle* h, int64_t* free_bytes, int64_t* total_bytes);

/* ---- a1/a3: SNP QC + standardisation parameters (prepare.rs:1100-1422, 1641-1745) -------- */
/* Any of mu/sigma/keep may be NULL.  mu, sigma are the f32 values the reference stores
 * (prepare.rs:1313,1364); 0 for dropped SNPs. */
GPCA_API int gpca_snp_stats(gpca_handle* h, const gpca_qc_config* qc, float* mu, float* sigma, uint8_t* keep);
/* counts[i] = {n_valid, n_hom0, n_het, n_hom2}; reason[i]: 0 kept, 1 call-rate, 2 no valid,
 * 3 MAF, 4 monomorphic, 5 HWE, 6 variance.  Either may be NULL. */
GPCA_API int gpca_get_snp_qc_detail(gpca_handle* h, uint32_t* counts, uint8_t* reason);
/* Caller-supplied parameters instead of gpca_snp_stats (e.g. LD-block restriction via keep). */
GPCA_API int gpca_set_standardization(gpca_handle* h, const float* mu, const float* sigma, const uint8_t* keep);
/* Current parameters, length M each (any may be NULL): what gpca_snp_stats computed or gpca_set_standardization set. */
GPCA_API int gpca_get_standardization(gpca_handle* h, float* mu, float* sigma, uint8_t* keep);
/* Host helper (no GPU): a symmetric eigen-solver (Householder tridiagonalisation + implicit QL) -- what gpca_rsvd ran for its l x l
 * step on the host up to round 4, kept as the pin of the device solver: CPU-only tests hold it to LAPACK, GPU tests hold the device
 * solver to it.  a_sym: n x n row-major (n <= 128); w: eigenvalues descending; v: eigenvectors in columns, row-major. */
GPCA_API int gpca_host_eigh_desc(const double* a_sym, int32_t n, double* w, double* v);
/* Test hook (GPU): the DEVICE eigen-solver gpca_rsvd runs for its l x l step (csrc/small_eig.hip: cyclic two-sided Jacobi in LDS up to
 * 64 columns, tridiagonalisation + QL split over waves for 65-128; the call's stream never waits for the host), on a caller's matrix;
 * same conventions as above. */
GPCA_API int gpca_device_eigh_desc(gpca_handle* h, const double* a_sym, int32_t n, double* w, double* v);
/* Host helper, same branches as prepare.rs:1641-1745. */
GPCA_API double gpca_hwe_chi_squared_p_value(uint64_t n_hom1, uint64_t n_het, uint64_t n_hom2);

/* ---- a2/a7: the pull API (prepare.rs:1838-2030) ------------------------------------------ */
/* PcaSnpId = rank among kept SNPs; QcSampleId = sample column.  out: f32 [ns][nj], C order.
 * Returns GPCA_ERR_MISSING_GENOTYPE if any requested genotype is -127 (message names the ids). */
GPCA_API int gpca_standardize_block(gpca_handle* h, const int64_t* pca_snp_ids, int64_t ns,
                           const int64_t* qc_sample_ids, int64_t nj, float* out);
GPCA_API int64_t gpca_num_pca_snps(gpca_handle* h);   /* prepare.rs:2024-2026 */
GPCA_API int64_t gpca_num_qc_samples(gpca_handle* h); /* prepare.rs:2027-2029 */
/* original row (BIM index) of every PCA SNP, length gpca_num_pca_snps; prepare.rs:1833-1835 */
GPCA_API int gpca_get_pca_snp_rows(gpca_handle* h, int64_t* rows);

/* ---- a5/a6: randomized PCA (PCA::rfit main.rs:648-656; compute_pca main.rs:365) ---------- */
/* l = k + oversample columns (<= 128, see Limits); power_iters QR-stabilised iterations; Omega from
 * Philox4x32-10 keyed by seed.  Requires stats.  Results stay on the device until fetched.
 * Row-sharded runs (gpca_comm_init / gpca_set_allreduce_hook): the ranks agree on a status word before the first and after
 * the last exchange of the call, so a rank-local failure (missing genotype in one shard, out of memory, a failed launch)
 * is returned by EVERY rank instead of leaving the others inside a collective. */
GPCA_API int gpca_rsvd(gpca_handle* h, int32_t k, int32_t oversample, int32_t power_iters, uint64_t seed);
GPCA_API int gpca_get_scores(gpca_handle* h, float* out /* [N][k] */);       /* main.rs:389 */
GPCA_API int gpca_get_scores_f64(gpca_handle* h, double* out /* [N][k] */);  /* main.rs:659 (f64 path) */
GPCA_API int gpca_get_eigenvalues(gpca_handle* h, double* out /* [k] */);    /* main.rs:394 */
GPCA_API int gpca_get_singular_values(gpca_handle* h, double* out /* [k+oversample] */);
GPCA_API int gpca_get_loadings(gpca_handle* h, float* out /* [num_pca_snps][k] */); /* main.rs:407 */
/* PCA::transform (main.rs:659) on the resident matrix: scores = A^T * loadings, f64 [N][k]. */
GPCA_API int gpca_transform(gpca_handle* h, double* out);

/* ---- f3: the stages of EigenSNPCoreAlgorithm::compute_pca (main.rs:311-327, 359-366) ------------------------------------------
 * The algorithm lives in the un-vendored efficient_pca crate (Cargo.toml:30, branch "main", no pinned revision): what follows is
 * the stage structure of its published description -- per-LD-block local bases learnt on a sample subset, condensed features of
 * all samples, row standardisation, an initial global randomized PCA of the condensed features, refinement passes on the full
 * matrix -- with every pass over the genotypes on the device.  Parity with the crate is UNPINNED (no source, no golden vectors);
 * the host mirror (EigenSNPCoreAlgorithm.compute_pca, local_stage = True) drives these calls and is checked against a numpy
 * restatement of the same stages and against exact PCA.
 *
 * gpca_copy_rows: dst (same device and storage mode) receives rows [row0, row0 + rows) of src's RESIDENT matrix, device to device:
 *   an LD block as a matrix of its own (follow with gpca_set_standardization on dst).
 * gpca_set_sample_mask: mask[n] != 0 = sample n takes part in learning the basis; gpca_rsvd then learns scores / loadings from
 *   those columns only (its eigenvalues are variances over the subset: / (subset size - 1)), gpca_transform still projects every
 *   sample.  NULL clears.
 * gpca_set_condensed_basis: W[i][0..cmax) = SNP row i's coefficients (local loading / feature s.d.) on the condensed features
 *   [feat0[i], feat0[i] + cmax) of its block, zero-padded; feat0[i] < 0 = in no block; R = number of condensed features.
 * gpca_rsvd_condensed: randomized PCA of the row-standardised condensed feature matrix C* = W^T A (R x N, never formed: its
 *   products run through the genotype GEMMs).  Leaves N x k sample scores and C*'s eigenvalues; no loadings.
 * gpca_refine: one refinement pass from sample scores S0 [N][k]: L = orth(A S0), S = A^T L, S^T S = W Sigma^2 W^T;
 *   scores = S W, loadings = L W, eigenvalues = Sigma^2 / (N - 1), through the usual getters. */
GPCA_API int gpca_copy_rows(gpca_handle* dst, gpca_handle* src, int64_t row0, int64_t rows);
GPCA_API int gpca_set_sample_mask(gpca_handle* h, const uint8_t* mask /* [N] or NULL */);
GPCA_API int gpca_set_condensed_basis(gpca_handle* h, const float* W /* [M][cmax] */, const int32_t* feat0 /* [M] */, int32_t cmax, int64_t R);
GPCA_API int gpca_rsvd_condensed(gpca_handle* h, int32_t k, int32_t oversample, int32_t power_iters, uint64_t seed);
GPCA_API int gpca_refine(gpca_handle* h, const double* S0 /* [N][k] */, int32_t k);

/* ---- e: SNP-row sharding across GPUs ------------------------------------------------------- */
#define GPCA_UNIQUE_ID_BYTES 128
GPCA_API int gpca_comm_get_unique_id(void* out_id /* GPCA_UNIQUE_ID_BYTES */);
/* This handle holds rows [snp_offset, snp_offset + M) of a matrix sharded over `world` ranks.
 * Creates an RCCL communicator; the N x l sketch and l x l Gram blocks are all-reduced. */
GPCA_API int gpca_comm_init(gpca_handle* h, int32_t world, int32_t rank, const void* unique_id, int64_t snp_offset);
/* Host-staged all-reduce hook (sum, in place, f64) used instead of RCCL when set: lets any
 * transport (MPI, gloo) carry the exchange; also how the CPU tests exercise the N>1 path.  Called with the handle's lock held
 * (other handles -- the peers' -- are unaffected). */
typedef int (*gpca_allreduce_fn)(void* user, double* host_buf, int64_t count);
GPCA_API int gpca_set_allreduce_hook(gpca_handle* h, gpca_allreduce_fn fn, void* user, int32_t world,
                            int32_t rank, int64_t snp_offset);

/* How many ranks the handle's exchange actually reaches: a 1.0 per rank summed through the same transport as the sketch (RCCL
 * communicator or hook).  Collective: every rank of the sharded matrix must call it.  1 for an unsharded handle. */
GPCA_API int gpca_comm_count_ranks(gpca_handle* h, int32_t* ranks);

/* ---- d: measurement ------------------------------------------------------------------------- */
typedef struct gpca_kernel_timing {
    char name[32];
    int64_t launches;
    double total_ms;   /* sum of HIP-event durations on the engine's stream */
    double flops;      /* algorithmic (un-padded) flops summed over those launches */
    double bytes;      /* algorithmic HBM bytes summed over those launches */
} gpca_kernel_timing;
/* Timings are OFF by default (gpca_enable_timings(h, 1) turns them on; n > 1: only every n-th gpca_rsvd call records its events -- a
 * recorded event pair costs the stream about 5 us of idle time, which a caller who wants per-kernel figures of a long run need not
 * pay on every call: launches / total_ms then cover the sampled calls only); records are folded into per-name totals
 * once 32768 are pending, so a long-running host never accumulates events.
 * Timings accumulated since the last gpca_reset_timings (events resolved lazily here). */
GPCA_API int gpca_get_timings(gpca_handle* h, gpca_kernel_timing* out, int32_t cap, int32_t* n);
GPCA_API int gpca_reset_timings(gpca_handle* h);
GPCA_API int gpca_enable_timings(gpca_handle* h, int32_t on);
GPCA_API int gpca_synchronize(gpca_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* GPCA_H */
