#pragma once
// Internals shared by the translation units of libgpca.so's host driver (gpca_api.cpp: lifecycle, QC statistics, pull API, exchange,
// timings; gpca_residency.cpp: uploads, panel sources, out-of-core ring; gpca_rsvd.cpp: the randomized-PCA stages).  Nothing here is
// exported: the library is built with -fvisibility=hidden and only the extern "C" entry points of include/gpca.h carry GPCA_API.
//
// C ABI + host driver of the randomized-PCA engine (declared in include/gpca.h).
//
// The driver keeps genotypes, basis, sketches and results in HBM behind the opaque handle and only
// moves l x l (<= 64 x 64) f64 blocks to the host for Cholesky / Jacobi.  Per randomized-PCA call:
//   sketch Y = A^T Omega, orth;  q x { T = A Q, Y = A^T T, orth };  B = A Q;  eig(B^T B)
// = 4 passes over the int8 matrix, 12*l flop per genotype.
#include "../../include/gpca.h"
#include "kernels.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace gpca;

// ---- minimal RCCL surface, resolved with dlopen so that libgpca.so loads on hosts without a GPU ----
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef int (*pfn_ncclGetUniqueId)(ncclUniqueId_t*);
typedef int (*pfn_ncclCommInitRank)(ncclComm_t*, int, ncclUniqueId_t, int);
typedef int (*pfn_ncclAllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*pfn_ncclCommDestroy)(ncclComm_t);
typedef const char* (*pfn_ncclGetErrorString)(int);
struct RcclApi {
    void* lib = nullptr;
    pfn_ncclGetUniqueId GetUniqueId = nullptr;
    pfn_ncclCommInitRank CommInitRank = nullptr;
    pfn_ncclAllReduce AllReduce = nullptr;
    pfn_ncclCommDestroy CommDestroy = nullptr;
    pfn_ncclGetErrorString GetErrorString = nullptr;
    bool load() {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        GetUniqueId = (pfn_ncclGetUniqueId)dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (pfn_ncclCommInitRank)dlsym(lib, "ncclCommInitRank");
        AllReduce = (pfn_ncclAllReduce)dlsym(lib, "ncclAllReduce");
        CommDestroy = (pfn_ncclCommDestroy)dlsym(lib, "ncclCommDestroy");
        GetErrorString = (pfn_ncclGetErrorString)dlsym(lib, "ncclGetErrorString");
        return GetUniqueId && CommInitRank && AllReduce && CommDestroy;
    }
};
extern RcclApi g_rccl;
enum { kNcclFloat64 = 8, kNcclSum = 0 };  // ncclDataType_t / ncclRedOp_t values (rccl.h)

struct TimingRec { std::string name; hipEvent_t a, b; double flops, bytes; };

// One view of genotype rows in HBM: the whole resident matrix, or the panel currently in a ring slot (streamed mode).
struct PanelView { const int8_t* g8; const uint8_t* g2; int64_t row0, rows, rows_pad; int index; };

// Turns a gpca_panel_source into rows in device memory: device generators run on `st`; host sources go through a ring of pinned
// staging panels that a library-owned worker thread fills (callback, or the library's copy threads for MAPPED_* sources) up to
// n_stage - 1 panels ahead of the one being copied to the device (and, for .bed bytes / 2-bit storage, recoded) on `st`; a MAPPED_*
// source that could be page-locked in place is DMA-ed straight from the caller's memory (no staging, no worker).
struct Filler {
    gpca_panel_source src{};
    int64_t chunk_rows = 0;          // most rows one fill() call may ask for
    int64_t stage_ld = 0;            // bytes per row of the host staging buffers (= bytes per row that travel)
    uint32_t* d_thresh = nullptr;    // SYNTH*: [M][n_pop]
    int8_t* d_scratch8 = nullptr;    // int8 rows on their way to 2-bit storage: [chunk_rows][ldg]
    uint8_t* d_raw = nullptr;        // .bed bytes on the device: [chunk_rows][bpr]
    unsigned* d_flags = nullptr;     // invalid-genotype flag of the pack kernel
    bool open = false;
    // host sources
    bool host = false, mapped = false, registered = false;
    const uint8_t* map_base = nullptr; int64_t map_ld = 0; size_t reg_bytes = 0;
    int device = 0, copy_threads = 1;
    int n_stage = 0;
    std::vector<void*> h_stage;
    std::vector<hipEvent_t> ev_stage;
    // staging ring state, guarded by m.  Both sides walk the buffers round-robin from a common origin: job #i uses buffer i % n_stage.
    enum { kFree = 0, kReady = 1, kInFlight = 2 };
    std::vector<int> st_state, st_rc;
    std::vector<int64_t> st_row0;
    std::deque<std::pair<int64_t, int64_t>> jobs;    // posted (row0, rows), not yet started
    uint64_t posted = 0, produced = 0, consumed = 0;
    bool busy = false, quit = false;
    std::mutex m; std::condition_variable cv; std::thread worker;
    // host-side accounting (gpca_stream_get_info)
    int64_t fills = 0; double fill_host_ms = 0.0, fill_wait_ms = 0.0, register_ms = 0.0;
    std::map<void*, int64_t> held;   // GPCA_SOURCE_BENCH_HOLD: device buffers that hold a generated panel (rows)
};

struct StreamState {
    bool on = false;
    Filler fl;
    int64_t panel_rows = 0;
    int n_panels = 0, ring = 0;
    std::vector<void*> slot;
    std::vector<hipEvent_t> ev_filled, ev_free;
    std::vector<char> free_pending;
    hipStream_t st_fill = nullptr;
    int64_t seq = 0;                 // panels filled so far: slot = seq % ring
    int fused = 1;                   // power iterations read every panel once (K1 -> quantise -> K2 per panel): 4 passes per call, not 6
    // panel cache (gpca_stream_set_cache): the leading panels keep a buffer of their own in the HBM the ring and the workspace
    // leave free; they are asked of the source once and read in place on every later pass
    std::vector<void*> cache;
    std::vector<hipEvent_t> ev_cache;
    std::vector<char> cache_filled;
};

struct gpca_handle {
    std::recursive_mutex mu;     // every entry point locks it: a handle may be shared between host threads (gpca.h, "Threading")
    KernelOpts ko;
    StreamState sm;
    int device = 0;
    int precision = GPCA_PREC_I8_EXACT;
    int storage = GPCA_STORE_INT8;         // the residency in use: GPCA_STORE_INT8 or GPCA_STORE_2BIT, never AUTO
    int storage_cfg = GPCA_STORE_AUTO;     // what the caller asked for; AUTO is resolved by alloc_genotypes when rows arrive
    int nd_cfg = 0;                        // gpca_config.digit_planes (0 = follow the residency)
    int auto_pin = 0;                      // AUTO: residency forced for the allocation in progress (an int8 upload with values 2-bit codes cannot
                                           // carry; gpca_copy_rows from a source of the other kind); 0 = follow the sample count
    hipStream_t st = nullptr;
    std::string err;

    // genotypes
    int64_t M = 0, N = 0, ldg = 0, Mpad = 0;   // ldg = samples padded to the kernels' tiles; Mpad = round_up(M, 128): zero rows, so the GEMM loops carry no predicates
    int64_t ld8 = 0;           // byte pitch of the int8 rows: ldg, plus 256 when ldg / 256 is even (see alloc_genotypes)
    int64_t cap_rows_pad = 0;  // rows the resident genotype buffer was allocated for (gpca_copy_rows reuses it for a smaller block)
    int64_t cap_stats_pad = 0; // rows the per-row statistics arrays (d_mu .. d_counts) were allocated for
    int8_t* dG = nullptr;      // GPCA_STORE_INT8: [Mpad][ldg]
    uint8_t* dG2 = nullptr;    // GPCA_STORE_2BIT: [Mpad][ld2], ld2 = ldg / 4, dosage codes (3 = missing)
    int64_t ld2 = 0;
    uint32_t pack_flags = 0;   // invalid genotypes seen while packing int8 input

    // stats
    bool have_stats = false;
    float *d_mu = nullptr, *d_sigma = nullptr, *d_r = nullptr, *d_b = nullptr;
    uint8_t *d_keep = nullptr, *d_reason = nullptr;
    uint32_t *d_counts = nullptr, *d_flags = nullptr;
    int64_t n_pca = 0;
    std::vector<int64_t> pca_rows;
    int64_t* d_pca_rows = nullptr;
    uint32_t flags = 0;

    // rsvd workspace / results
    int k = 0, l = 0, L = 0;
    bool have_rsvd = false;
    float *dQ = nullptr, *dT = nullptr, *dTb = nullptr, *dYpart = nullptr, *d_cpart = nullptr, *d_s32 = nullptr;
    double* h_pin = nullptr;     // pinned host staging: the result block of the device eigen step; gpca_transform's selection matrix
    int spin_sync = 1;           // busy-poll the stream at the two syncs of gpca_rsvd (GPCA_CFG_NO_SPIN_SYNC: hipStreamSynchronize)
    int* d_cholflag = nullptr;   // first failed CholeskyQR pivot + 1 (0 = ok), written by k_chol_inv
    double *d_scratch64 = nullptr, *dY = nullptr, *d_c = nullptr, *d_part64 = nullptr, *dW = nullptr, *dZ = nullptr, *d_s64 = nullptr;
    double* d_tr64 = nullptr; size_t cap_tr64 = 0;   // [N][k] compacted output of gpca_transform
    double* d_scores64 = nullptr; float* d_scores32 = nullptr; float* d_load32 = nullptr; int* d_sign = nullptr;
    double* d_eigres = nullptr;      // [kEigResCount] result block of the device eigen step (singular values, eigenvalues, flags): read back once per call
    double* d_cand_val = nullptr; int64_t* d_cand_idx = nullptr;   // [scores_num_parts][kMaxSketch] candidates of the scores' sign rule
    size_t cap_Q = 0, cap_T = 0, cap_Tb = 0, cap_Ypart = 0, cap_cpart = 0, cap_Y = 0, cap_part64 = 0, cap_scores = 0, cap_load = 0;
    std::vector<double> eig, sv;
    GttPlan plan{};
    GqPlan gqplan{};
    Gtt8Plan plan8{};
    // exact-integer path
    int8_t *dQd = nullptr, *dTd = nullptr;
    bool c_fold_pending = false;     // K1's tail ran as launch_post_k1: the digit scale is made, c's second stage waits in d_scratch64 for the quantisation
    double* d_apart = nullptr; size_t cap_apart = 0; bool apart_valid = false; int64_t apart_parts = 0;   // column abs-max partials of T' from the K1 epilogue
    const double* apart_src[4] = {nullptr, nullptr, nullptr, nullptr};   // where the partials of each 32-column block sit (kMaxSketch / 32 blocks)
    float* d_rmax = nullptr; bool rmax_valid = false;   // max_i r[i]: the sketch's digit scale is 6.67 * rmax (k_omega); recomputed when r changes
    double* d_amax_run = nullptr;    // [2][32] running column abs-max over the panels of a streamed K1 sweep
    double* d_yint = nullptr; size_t cap_yint = 0;   // [halves][N][32] integer partial sums of a streamed K2 sweep
    // EigenSNP stages (gpca_set_sample_mask / gpca_set_condensed_basis / gpca_rsvd_condensed / gpca_refine, gpca_rsvd.cpp)
    uint8_t* d_smask = nullptr;      // [N] 1 = the sample takes part in learning the basis (nullptr: all samples)
    int64_t n_smask = 0;             // samples in the subset (eigenvalues of a masked gpca_rsvd are variances over these: / (n_smask - 1))
    float* d_cw = nullptr; int32_t* d_cfeat0 = nullptr; int c_cmax = 0; int64_t c_R = 0; int c_B = 0;   // block-diagonal W = U_blk Lambda^-1
    int64_t *d_cblk_row0 = nullptr, *d_cblk_row1 = nullptr; int32_t *d_cblk_feat0 = nullptr, *d_cblk_c = nullptr;
    double* dP = nullptr; size_t cap_P = 0;         // [(R + 16)][L] condensed-side factor
    float* d_lqr = nullptr; size_t cap_lqr = 0;     // [Mpad][L] orthonormal SNP-side factor of a refinement pass
    float *d_ones = nullptr, *d_zeros = nullptr; size_t cap_ones = 0, cap_zeros = 0;
    bool loadings_valid = true;      // (gpca_rsvd_condensed leaves scores only)
    double* d_status = nullptr;      // [16] status word the ranks agree on
    double* h_status = nullptr;      // pinned [32]: contribution | agreed histogram
    hipEvent_t ev_status = nullptr;  // after the agreed histogram has landed in h_status (agree_status_begin / _end)
    std::string status_own;          // this rank's error text when it contributed
    // The pull API (gpca_standardize_block) is served concurrently, the way the reference's accessor is called from many workers at once
    // (prepare.rs:1770-1779, 1838; 1-16 actor threads, main.rs:279-283): a call validates and claims a lane under the handle's lock,
    // then runs its copies and its kernel on the lane's own stream WITHOUT the lock.  Every other entry point waits until no pull is
    // in flight (LOCK -> drain_pulls), so the matrix and its statistics never change under a running pull.
    struct PullLane {
        hipStream_t st = nullptr;
        hipEvent_t dep = nullptr;                                // recorded on the engine's stream when the lane is claimed: the pull starts after the work queued there
        int64_t *d_rows = nullptr, *d_cols = nullptr; float* d_out = nullptr; unsigned long long* d_err = nullptr;
        size_t cap_rows = 0, cap_cols = 0, cap_out = 0;          // persistent scratch (grown on demand, never freed per call)
    };
    std::mutex pull_mu;                    // lanes_free, lanes_total, pulls_in_flight
    std::condition_variable pull_cv;
    std::vector<PullLane*> lanes_free;
    int lanes_total = 0;
    std::atomic<int> pulls_in_flight{0};
    double *dYpart64 = nullptr, *d_qscale = nullptr, *d_qinv = nullptr, *d_tscale = nullptr, *d_tinv = nullptr;
    size_t cap_Qd = 0, cap_Td = 0, cap_Ypart64 = 0;
    int nd = 4;           // digit planes of the exact path (gpca_config.digit_planes): 4 x base 128, or 3 x base 256 (packed storage)
    uint64_t gen = 0;         // generation id of this handle (per-thread error texts are keyed by it)
    int simple_kernels = 0;   // gpca_config.reserved[0] & GPCA_CFG_SIMPLE_KERNELS: the register-only reference kernels (gemm_i8_simple.hip)
    int narrow_ok = 1;    // matrices of at most 256 samples (int8 rows) run the narrow K1 / K2 (GPCA_CFG_NO_NARROW: the wide kernels on padded rows)
    int gq_waves_target = 1024, gtt_waves_target = 2048;   // resident-wave targets (256 CUs x 4 SIMDs x 1 or 2), tuned on MI355X

    // Compact child: when QC dropped most SNP rows (the reference's solver only ever sees the PCA SNPs, prepare.rs:1465-1469), the kept
    // rows are gathered into a matrix of their own and gpca_rsvd / gpca_transform / the getters run on it: every pass, the sketch and
    // the quantisations then cost n_pca rows instead of M (configs[2]: 203 512 of 1 066 557).  The child shares the parent's stream,
    // draws Omega by the ORIGINAL row index of its rows (row_ids), and is rebuilt whenever the parent's rows or keep mask change.
    gpca_handle* child = nullptr;
    bool child_valid = false, rsvd_on_child = false;
    bool is_child = false;           // (a child never compacts again and does not own its stream)
    const int64_t* d_row_ids = nullptr;   // child: original row of every row (the parent's d_pca_rows; not owned)
    int compact_ok = 1;              // GPCA_CFG_NO_COMPACT: never compact

    // comm
    int world = 1, rank = 0;
    int64_t snp_offset = 0;
    ncclComm_t comm = nullptr;
    gpca_allreduce_fn hook = nullptr;
    void* hook_user = nullptr;
    std::vector<double> hook_buf;

    // timings (off by default; bounded: pending records are folded into `agg` once kMaxTimingRecs are outstanding)
    bool timing_on = false;
    int timing_every = 1;        // gpca_enable_timings(h, n > 1): only every n-th gpca_rsvd call records its events (an event pair costs the stream ~5 us of idle time)
    int64_t timing_calls = 0;
    bool timing_skip = false;    // the gpca_rsvd call in progress is not one of the sampled ones
    int open_timers = 0;         // ScopedTimers alive (their records must not be folded away under them)
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> ev_pool;
    std::vector<gpca_kernel_timing> agg;
};
constexpr size_t kMaxTimingRecs = 32768;
constexpr size_t kPostK1Scratch = 256 * 32;     // doubles of d_scratch64 per 32-column block of the sketch (<= 256 slices x 32 columns; kSumScratchElems holds 32 of them)
// every entry point: take the handle's lock and make its device the calling thread's current one (a host thread that drives
// several handles on different GPUs, or that last touched another device, would otherwise launch on the wrong one)
#define LOCK(h) std::lock_guard<std::recursive_mutex> lock_guard_(h->mu); (void)hipSetDevice(h->device); drain_pulls(h)
// read-only getters and the first half of a pull: the lock without waiting for the pulls in flight
#define LOCK_SHARED(h) std::lock_guard<std::recursive_mutex> lock_guard_(h->mu); (void)hipSetDevice(h->device)
constexpr int kMaxPullLanes = 16;        // the reference serves its accessor from at most 16 actor threads (main.rs:279-283)
inline void drain_pulls(gpca_handle* h) {
    if (h->pulls_in_flight.load(std::memory_order_acquire) == 0) return;
    std::unique_lock<std::mutex> lk(h->pull_mu);   // (no new pull can start: its first half needs the handle's lock, which the caller holds)
    h->pull_cv.wait(lk, [&] { return h->pulls_in_flight.load(std::memory_order_acquire) == 0; });
}

extern thread_local std::string g_last_global_err;
int fail(gpca_handle* h, int code, const std::string& msg);
#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            char buf_[512];                                                                            \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return fail(h, e_ == hipErrorOutOfMemory ? GPCA_ERR_OOM : GPCA_ERR_HIP, buf_);             \
        }                                                                                              \
    } while (0)
#define CHK(x) do { int rc_ = (x); if (rc_ != GPCA_OK) return rc_; } while (0)

template <typename T>
inline int ensure(gpca_handle* h, T*& p, size_t& cap, size_t need_elems) {
    if (cap >= need_elems && p) return GPCA_OK;
    if (p) { HIPCHK(hipFree(p)); p = nullptr; cap = 0; }
    HIPCHK(hipMalloc((void**)&p, need_elems * sizeof(T)));
    cap = need_elems;
    return GPCA_OK;
}
template <typename T>
inline void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

// Genotype storage (resident matrix, panel ring, panel cache).  Plain hipMalloc on purpose: physically contiguous allocations
// (hipExtMallocWithFlags + hipDeviceMallocContiguous) stream 0 - 4 % faster in a microbenchmark (profiles/r3_kbench_place_contig.log)
// but a SECOND such allocation in one process, after the first was freed, gave kernels stale contents while copies read the right
// bytes (profiles/r3_diag_contig.log: wrong mu / sigma / eigenvalues on the second engine of a process, never on the first) -- not
// something this library can repair, so it does not use them.
inline hipError_t malloc_genotypes(const gpca_handle*, void** p, size_t bytes) { return hipMalloc(p, bytes); }

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static_assert(gpca::kMaxSketchCols == 128, "kernels.h");
constexpr int kMaxSketch = gpca::kMaxSketchCols;   // most columns (k + oversample) a sketch may have: L = 32, 64 (specialised helpers) or 128 (wide_sketch.hip)

// ---- timing ---------------------------------------------------------------------------------------
void fold_timings(gpca_handle* h);   // resolve pending records into per-name totals and recycle their events
struct ScopedTimer {
    gpca_handle* h; bool on; size_t idx = 0; hipStream_t st;
    ScopedTimer(gpca_handle* h_, const char* name, double flops, double bytes, hipStream_t st_ = nullptr, bool enable = true)
        : h(h_), on(h_->timing_on && !h_->timing_skip && enable), st(st_ ? st_ : h_->st) {
        if (!on) return;
        // (never while another timer is open: a fold clears recs and recycles the open record's events -- nested timers are the
        //  sweep timers of the streamed passes around their panel_fill / per-launch records)
        if (h->recs.size() >= kMaxTimingRecs && h->open_timers == 0) fold_timings(h);
        TimingRec r; r.name = name; r.flops = flops; r.bytes = bytes; r.a = r.b = nullptr;
        for (hipEvent_t* e : {&r.a, &r.b}) {
            if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); }
            else if (hipEventCreateWithFlags(e, hipEventDisableSystemFence) != hipSuccess) { on = false; return; }   // (timing only: no system-scope fence when it records)
        }
        (void)hipEventRecord(r.a, st);
        h->recs.push_back(r); idx = h->recs.size() - 1; h->open_timers++;
    }
    ~ScopedTimer() { if (on) { (void)hipEventRecord(h->recs[idx].b, st); h->open_timers--; } }
};


// ---- shared between the translation units ---------------------------------------------------------------------
inline bool have_genotypes(const gpca_handle* h) { return h->dG || h->dG2 || h->sm.on; }
inline bool multi_rank(const gpca_handle* h) { return h->world > 1 || h->hook != nullptr; }
constexpr size_t kPlaneBytesPerBlock = (size_t)gpca::kDigits * 1024;   // digit planes of one 32-row (or 32-sample) block
// gpca_residency.cpp
void free_stats(gpca_handle* h);
int alloc_stats(gpca_handle* h);
int refresh_pca_rows(gpca_handle* h);
void free_eigensnp(gpca_handle* h);
void free_ws(gpca_handle* h);
void stream_close(gpca_handle* h);
int finish_pack_flags(gpca_handle* h, unsigned* d_flags, hipStream_t st);
int filler_fill(gpca_handle* h, Filler& f, int64_t row0, int64_t rows, void* dst, hipStream_t st);
void filler_post(Filler& f, int64_t row0, int64_t rows);   // tell the worker which rows filler_fill will ask for next (in this order)
void filler_cancel(Filler& f);                              // drop what was posted and not consumed (a pass that ended early)
// gpca_api.cpp
int allreduce_f64(gpca_handle* h, double* dbuf, int64_t count);   // in-place sum across the ranks that share the sharded matrix
hipError_t stream_wait(gpca_handle* h);
int agree_status(gpca_handle* h, int local_rc, const char* where);
int agree_status_begin(gpca_handle* h, int local_rc);                        // enqueue contribution + exchange + copy back, record ev_status
int agree_status_end(gpca_handle* h, int local_rc, const char* where);       // wait for ev_status, return what the ranks agreed on
void status_histogram(double* slots16, int local_rc);
int status_verdict(gpca_handle* h, const double* v, int local_rc, const std::string& own, const char* where);
void drop_child(gpca_handle* h);    // the compact child is stale (rows, statistics or keep mask changed) or the handle goes away

// fn(view) once for the resident matrix, or once per panel (generated / copied ahead on the fill stream)
template <class F>
inline int for_each_panel_walk(gpca_handle* h, F&& fn) {
    StreamState& sm = h->sm;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const int64_t row_bytes = packed ? h->ld2 : h->ld8;
    for (int p = 0; p < sm.n_panels; ++p) {
        const int64_t row0 = (int64_t)p * sm.panel_rows;
        const int64_t rows = std::min(sm.panel_rows, h->M - row0);
        const int64_t rows_pad = round_up(rows, kGQRowsPerWave);
        if (p < (int)sm.cache.size()) {                      // cached panel: filled on first use, never again
            void* buf = sm.cache[p];
            if (!sm.cache_filled[p]) {
                {
                    ScopedTimer t(h, "panel_fill", 0.0, (double)rows * (double)h->N, sm.st_fill);
                    if (rows_pad > rows) HIPCHK(hipMemsetAsync((char*)buf + (size_t)rows * row_bytes, 0, (size_t)(rows_pad - rows) * row_bytes, sm.st_fill));
                    CHK(filler_fill(h, sm.fl, row0, rows, buf, sm.st_fill));
                }
                HIPCHK(hipEventRecord(sm.ev_cache[p], sm.st_fill));
                HIPCHK(hipStreamWaitEvent(h->st, sm.ev_cache[p], 0));
                sm.cache_filled[p] = 1;
            }
            const PanelView pv{packed ? nullptr : (const int8_t*)buf, packed ? (const uint8_t*)buf : nullptr, row0, rows, rows_pad, p};
            CHK(fn(pv));
            continue;
        }
        const int s = (int)(sm.seq % sm.ring); sm.seq++;
        if (sm.free_pending[s]) HIPCHK(hipStreamWaitEvent(sm.st_fill, sm.ev_free[s], 0));   // the slot's last reader has finished
        {
            ScopedTimer t(h, "panel_fill", 0.0, (double)rows * (double)h->N, sm.st_fill);
            if (rows_pad > rows) HIPCHK(hipMemsetAsync((char*)sm.slot[s] + (size_t)rows * row_bytes, 0, (size_t)(rows_pad - rows) * row_bytes, sm.st_fill));
            CHK(filler_fill(h, sm.fl, row0, rows, sm.slot[s], sm.st_fill));
        }
        HIPCHK(hipEventRecord(sm.ev_filled[s], sm.st_fill));
        HIPCHK(hipStreamWaitEvent(h->st, sm.ev_filled[s], 0));
        const PanelView pv{packed ? nullptr : (const int8_t*)sm.slot[s], packed ? (const uint8_t*)sm.slot[s] : nullptr, row0, rows, rows_pad, p};
        CHK(fn(pv));
        HIPCHK(hipEventRecord(sm.ev_free[s], h->st)); sm.free_pending[s] = 1;
    }
    return GPCA_OK;
}
template <class F>
inline int for_each_panel(gpca_handle* h, F&& fn) {
    if (!h->sm.on) { const PanelView pv{h->dG, h->dG2, 0, h->M, h->Mpad, 0}; return fn(pv); }
    StreamState& sm = h->sm;
    // the worker of a host source learns the whole sweep up front (every panel that will be asked of the source, in order) and
    // stages ahead of the walk below
    for (int p = 0; p < sm.n_panels; ++p) {
        if (p < (int)sm.cache.size() && sm.cache_filled[p]) continue;
        const int64_t row0 = (int64_t)p * sm.panel_rows;
        filler_post(sm.fl, row0, std::min(sm.panel_rows, h->M - row0));
    }
    const int rc = for_each_panel_walk(h, fn);
    if (rc != GPCA_OK) filler_cancel(sm.fl);    // a sweep that ended early leaves nothing posted behind
    return rc;
}

