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
static struct RcclApi {
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
} g_rccl;
enum { kNcclFloat64 = 8, kNcclSum = 0 };  // ncclDataType_t / ncclRedOp_t values (rccl.h)

struct TimingRec { std::string name; hipEvent_t a, b; double flops, bytes; };

// One view of genotype rows in HBM: the whole resident matrix, or the panel currently in a ring slot (streamed mode).
struct PanelView { const int8_t* g8; const uint8_t* g2; int64_t row0, rows, rows_pad; int index; };

// Turns a gpca_panel_source into rows in device memory: device generators run on `st`; host callbacks fill one of two
// pinned staging buffers, which is then copied (and, for .bed bytes / 2-bit storage, recoded) on `st`.
struct Filler {
    gpca_panel_source src{};
    int64_t chunk_rows = 0;          // most rows one fill() call may ask for
    int64_t stage_ld = 0;            // bytes per row of the host staging buffers
    uint32_t* d_thresh = nullptr;    // SYNTH*: [M][n_pop]
    int8_t* d_scratch8 = nullptr;    // int8 rows on their way to 2-bit storage: [chunk_rows][ldg]
    uint8_t* d_raw = nullptr;        // .bed bytes on the device: [chunk_rows][bpr]
    unsigned* d_flags = nullptr;     // invalid-genotype flag of the pack kernel
    void* h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    char stage_pending[2] = {0, 0};
    int stage_idx = 0;
    bool open = false;
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
};

struct gpca_handle {
    std::recursive_mutex mu;     // every entry point locks it: a handle may be shared between host threads (gpca.h, "Threading")
    KernelOpts ko;
    StreamState sm;
    int device = 0;
    int precision = GPCA_PREC_F32_MFMA;
    int storage = GPCA_STORE_INT8;
    hipStream_t st = nullptr;
    std::string err;

    // genotypes
    int64_t M = 0, N = 0, ldg = 0, Mpad = 0;   // ldg = samples padded to the kernels' tiles; Mpad = round_up(M, 128): zero rows, so the GEMM loops carry no predicates
    int64_t ld8 = 0;           // byte pitch of the int8 rows: ldg, plus 256 when ldg / 256 is even (see alloc_genotypes)
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
    double* h_pin = nullptr;     // pinned host staging for the l x l blocks of the final eigenproblem (W | Z | flag)
    int spin_sync = 1;           // busy-poll the stream at the two syncs of gpca_rsvd (GPCA_SPIN_SYNC=0: hipStreamSynchronize)
    int* d_cholflag = nullptr;   // first failed CholeskyQR pivot + 1 (0 = ok), written by k_chol_inv
    double *d_scratch64 = nullptr, *dY = nullptr, *d_c = nullptr, *d_part64 = nullptr, *dW = nullptr, *dZ = nullptr, *d_s64 = nullptr;
    double* d_scores64 = nullptr; float* d_scores32 = nullptr; float* d_load32 = nullptr; int* d_sign = nullptr;
    size_t cap_Q = 0, cap_T = 0, cap_Tb = 0, cap_Ypart = 0, cap_cpart = 0, cap_Y = 0, cap_part64 = 0, cap_scores = 0, cap_load = 0;
    std::vector<double> eig, sv;
    GttPlan plan{};
    GqPlan gqplan{};
    Gtt8Plan plan8{};
    // exact-integer path
    int8_t *dQd = nullptr, *dTd = nullptr;
    double* d_apart = nullptr; size_t cap_apart = 0; bool apart_valid = false; int64_t apart_parts = 0;   // column abs-max partials of T' from the K1 epilogue
    const double* apart_src[2] = {nullptr, nullptr};   // where the partials of each 32-column half sit
    double* d_amax_run = nullptr;    // [2][32] running column abs-max over the panels of a streamed K1 sweep
    double* d_yint = nullptr; size_t cap_yint = 0;   // [halves][N][32] integer partial sums of a streamed K2 sweep
    double* d_status = nullptr;      // [16] status word the ranks agree on
    double* h_status = nullptr;      // pinned [32]: contribution | agreed histogram
    // persistent scratch of the pull API (no allocation per call)
    int64_t *d_blk_rows = nullptr, *d_blk_cols = nullptr; float* d_blk_out = nullptr; unsigned long long* d_blk_err = nullptr;
    size_t cap_blk_rows = 0, cap_blk_cols = 0, cap_blk_out = 0;
    double *dYpart64 = nullptr, *d_qscale = nullptr, *d_qinv = nullptr, *d_tscale = nullptr, *d_tinv = nullptr;
    size_t cap_Qd = 0, cap_Td = 0, cap_Ypart64 = 0;
    int nd = 4;           // digit planes of the exact path (gpca_config.digit_planes): 4 x base 128, or 3 x base 256 (packed storage)
    int gtt_dma = 1;      // K2 (int8-resident) by LDS-DMA (GPCA_GTT_DMA=0: register-staged k_gtt_x)
    int gq_dma = 1;       // K1 (int8-resident) genotype loads by LDS-DMA, full-line pieces (GPCA_GQ_DMA=0: register-staged k_gq_x)
    int lds_planes = 1;   // share the digit planes of the exact GEMMs through LDS (GPCA_LDS_PLANES=0 disables)
    int gq_waves_target = 1024, gtt_waves_target = 2048;   // resident-wave targets (256 CUs x 4 SIMDs x 1 or 2), tuned on MI355X

    // comm
    int world = 1, rank = 0;
    int64_t snp_offset = 0;
    ncclComm_t comm = nullptr;
    gpca_allreduce_fn hook = nullptr;
    void* hook_user = nullptr;
    std::vector<double> hook_buf;

    // timings (off by default; bounded: pending records are folded into `agg` once kMaxTimingRecs are outstanding)
    bool timing_on = false;
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> ev_pool;
    std::vector<gpca_kernel_timing> agg;
};
constexpr size_t kMaxTimingRecs = 32768;
#define LOCK(h) std::lock_guard<std::recursive_mutex> lock_guard_(h->mu)

static thread_local std::string g_last_global_err;

static int fail(gpca_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_last_global_err = msg;
    return code;
}
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
static int ensure(gpca_handle* h, T*& p, size_t& cap, size_t need_elems) {
    if (cap >= need_elems && p) return GPCA_OK;
    if (p) { HIPCHK(hipFree(p)); p = nullptr; cap = 0; }
    HIPCHK(hipMalloc((void**)&p, need_elems * sizeof(T)));
    cap = need_elems;
    return GPCA_OK;
}
template <typename T>
static void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- timing ---------------------------------------------------------------------------------------
static void fold_timings(gpca_handle* h) {   // resolve pending records into per-name totals and recycle their events
    for (auto& r : h->recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            size_t i = 0;
            for (; i < h->agg.size(); ++i) if (r.name == h->agg[i].name) break;
            if (i == h->agg.size()) { gpca_kernel_timing t{}; snprintf(t.name, sizeof t.name, "%s", r.name.c_str()); h->agg.push_back(t); }
            h->agg[i].launches++; h->agg[i].total_ms += ms; h->agg[i].flops += r.flops; h->agg[i].bytes += r.bytes;
        }
        h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b);
    }
    h->recs.clear();
}
struct ScopedTimer {
    gpca_handle* h; bool on; size_t idx = 0; hipStream_t st;
    ScopedTimer(gpca_handle* h_, const char* name, double flops, double bytes, hipStream_t st_ = nullptr, bool enable = true)
        : h(h_), on(h_->timing_on && enable), st(st_ ? st_ : h_->st) {
        if (!on) return;
        if (h->recs.size() >= kMaxTimingRecs) fold_timings(h);
        TimingRec r; r.name = name; r.flops = flops; r.bytes = bytes; r.a = r.b = nullptr;
        for (hipEvent_t* e : {&r.a, &r.b}) {
            if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) { on = false; return; }
        }
        (void)hipEventRecord(r.a, st);
        h->recs.push_back(r); idx = h->recs.size() - 1;
    }
    ~ScopedTimer() { if (on) (void)hipEventRecord(h->recs[idx].b, st); }
};

// ---- lifecycle --------------------------------------------------------------------------------------
extern "C" int gpca_version(void) { return GPCA_VERSION; }

extern "C" const char* gpca_status_string(int s) {
    switch (s) {
        case GPCA_OK: return "ok";
        case GPCA_ERR_BAD_ARG: return "bad argument";
        case GPCA_ERR_OOM: return "out of device memory";
        case GPCA_ERR_HIP: return "HIP runtime error";
        case GPCA_ERR_RCCL: return "RCCL error";
        case GPCA_ERR_MISSING_GENOTYPE: return "missing genotype in a PCA SNP";
        case GPCA_ERR_NOT_CONVERGED: return "sketch lost rank";
        case GPCA_ERR_STATE: return "call out of order";
        case GPCA_ERR_NO_DEVICE: return "no HIP device";
        case GPCA_ERR_INVALID_GENOTYPE: return "genotype outside {0,1,2} in a PCA SNP";
        default: return "unknown status";
    }
}

extern "C" const char* gpca_last_error(gpca_handle* h) { return h ? h->err.c_str() : g_last_global_err.c_str(); }

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

extern "C" int gpca_create(const gpca_config* cfg, gpca_handle** out) {
    if (!out) return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, GPCA_ERR_NO_DEVICE, "gpca_create: no HIP device visible (this engine has no CPU fallback)");
    gpca_handle* h = new gpca_handle();
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: device ordinal out of range"); }
    h->device = dev;
    h->precision = cfg ? cfg->precision : GPCA_PREC_F32_MFMA;
    h->storage = cfg ? cfg->storage : GPCA_STORE_INT8;
    if (h->storage != GPCA_STORE_INT8 && h->storage != GPCA_STORE_2BIT) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: unknown storage mode"); }
    if (h->precision != GPCA_PREC_F32_MFMA && h->precision != GPCA_PREC_I8_EXACT) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: unknown precision mode"); }
    {
        const int dp = cfg ? cfg->digit_planes : 0;
        if (dp != 0 && dp != 3 && dp != 4) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: digit_planes must be 0, 3 or 4"); }
        if (dp == 3 && !(h->precision == GPCA_PREC_I8_EXACT && h->storage == GPCA_STORE_2BIT)) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: digit_planes = 3 is implemented for GPCA_PREC_I8_EXACT with GPCA_STORE_2BIT"); }
        h->nd = dp == 3 ? 3 : 4;
    }
    // diagnostic switches, read once per handle (defaults are the tuned values; DESIGN.md "Diagnostic switches")
    h->gq_waves_target = std::max(4, env_int("GPCA_GQ_WAVES", h->gq_waves_target));
    h->gtt_waves_target = std::max(4, env_int("GPCA_GTT_WAVES", h->gtt_waves_target));
    h->lds_planes = env_int("GPCA_LDS_PLANES", h->lds_planes);
    h->gq_dma = env_int("GPCA_GQ_DMA", h->gq_dma);
    h->spin_sync = env_int("GPCA_SPIN_SYNC", h->spin_sync);
    h->gtt_dma = env_int("GPCA_GTT_DMA", h->gtt_dma);
    h->ko.stream_nt = env_int("GPCA_STREAM_NT", 0) != 0;
    h->ko.dma_nt = env_int("GPCA_GQ_DMA_NT", 1) != 0;
    h->ko.gq_r = env_int("GPCA_GQ_R", 4);
    h->ko.gq_slots = env_int("GPCA_GQ_SLOTS", 6) == 7 ? 7 : 6;
    h->ko.gtt_xcd = env_int("GPCA_GTT_XCD", 1);
    h->ko.gttx_xcd = env_int("GPCA_GTTX_XCD", 0);
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) {
        delete h; return fail(nullptr, GPCA_ERR_HIP, "gpca_create: hipSetDevice/hipStreamCreate failed");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            // kernels are compiled for gfx950 only; any other device cannot run them
            std::string m = std::string("gpca_create: device is ") + prop.gcnArchName + ", this library only carries gfx950 code";
            (void)hipStreamDestroy(h->st); delete h; return fail(nullptr, GPCA_ERR_NO_DEVICE, m);
        }
    }
    // > 64 KiB of dynamic LDS is an opt-in the runtime records per device: once per handle, on this handle's device
    int e = init_device_kernels_i8();
    if (e == 0) e = init_device_kernels_common();
    if (e != 0) {
        std::string m = std::string("gpca_create: cannot reserve the LDS the DMA kernels need (hipFuncSetAttribute: ") + hipGetErrorString((hipError_t)e) + ")";
        (void)hipStreamDestroy(h->st); delete h; return fail(nullptr, GPCA_ERR_HIP, m);
    }
    *out = h;
    return GPCA_OK;
}

static void free_stats(gpca_handle* h) {
    dfree(h->d_mu); dfree(h->d_sigma); dfree(h->d_r); dfree(h->d_b); dfree(h->d_keep); dfree(h->d_reason);
    dfree(h->d_counts); dfree(h->d_flags); dfree(h->d_pca_rows);
    h->have_stats = false; h->n_pca = 0; h->pca_rows.clear();
}
static void free_ws(gpca_handle* h) {
    dfree(h->dQ); dfree(h->dT); dfree(h->dTb); dfree(h->dYpart); dfree(h->d_cpart); dfree(h->d_s32); dfree(h->dY); dfree(h->d_c);
    dfree(h->d_part64); dfree(h->dW); dfree(h->dZ); dfree(h->d_s64); dfree(h->d_scores64); dfree(h->d_scores32);
    dfree(h->d_load32); dfree(h->d_sign); dfree(h->d_scratch64);
    dfree(h->dQd); dfree(h->dTd); dfree(h->dYpart64); dfree(h->d_apart); h->cap_apart = 0; dfree(h->d_cholflag); if (h->h_pin) { (void)hipHostFree(h->h_pin); h->h_pin = nullptr; } dfree(h->d_qscale); dfree(h->d_qinv); dfree(h->d_tscale); dfree(h->d_tinv);
    dfree(h->d_amax_run); dfree(h->d_yint); h->cap_yint = 0;
    h->cap_Qd = h->cap_Td = h->cap_Ypart64 = 0;
    h->cap_Q = h->cap_T = h->cap_Tb = h->cap_Ypart = h->cap_cpart = h->cap_Y = h->cap_part64 = h->cap_scores = h->cap_load = 0;
    h->have_rsvd = false;
}
static void filler_close(Filler& f) {
    dfree(f.d_thresh); dfree(f.d_scratch8); dfree(f.d_raw); dfree(f.d_flags);
    for (int i = 0; i < 2; ++i) {
        if (f.ev_stage[i]) { (void)hipEventSynchronize(f.ev_stage[i]); (void)hipEventDestroy(f.ev_stage[i]); f.ev_stage[i] = nullptr; }
        if (f.h_stage[i]) { (void)hipHostFree(f.h_stage[i]); f.h_stage[i] = nullptr; }
        f.stage_pending[i] = 0;
    }
    f.open = false;
}
static void stream_close(gpca_handle* h) {
    StreamState& sm = h->sm;
    if (sm.st_fill) (void)hipStreamSynchronize(sm.st_fill);
    if (h->st) (void)hipStreamSynchronize(h->st);
    filler_close(sm.fl);
    for (void* p : sm.slot) if (p) (void)hipFree(p);
    for (auto e : sm.ev_filled) (void)hipEventDestroy(e);
    for (auto e : sm.ev_free) (void)hipEventDestroy(e);
    sm.slot.clear(); sm.ev_filled.clear(); sm.ev_free.clear(); sm.free_pending.clear();
    if (sm.st_fill) { (void)hipStreamDestroy(sm.st_fill); sm.st_fill = nullptr; }
    sm.on = false; sm.seq = 0; sm.n_panels = 0;
}

extern "C" int gpca_destroy(gpca_handle* h) {
    if (!h) return GPCA_OK;
    { LOCK(h);
      (void)hipSetDevice(h->device);
      (void)hipStreamSynchronize(h->st);
      stream_close(h);
      for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
      for (auto e : h->ev_pool) (void)hipEventDestroy(e);
      if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
      free_stats(h); free_ws(h); dfree(h->dG); dfree(h->dG2);
      dfree(h->d_blk_rows); dfree(h->d_blk_cols); dfree(h->d_blk_out); dfree(h->d_blk_err); dfree(h->d_status); if (h->h_status) { (void)hipHostFree(h->h_status); h->h_status = nullptr; }
      (void)hipStreamDestroy(h->st);
    }
    delete h;
    return GPCA_OK;
}

extern "C" int gpca_synchronize(gpca_handle* h) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (h->sm.st_fill) HIPCHK(hipStreamSynchronize(h->sm.st_fill));
    HIPCHK(hipStreamSynchronize(h->st));
    return GPCA_OK;
}

// ---- genotype residency -------------------------------------------------------------------------------
static inline bool have_genotypes(const gpca_handle* h) { return h->dG || h->dG2 || h->sm.on; }

// dimensions + (resident = true) the device matrix; streamed mode only records the dimensions
static int alloc_genotypes(gpca_handle* h, int64_t M, int64_t N, bool resident = true) {
    if (M <= 0 || N <= 0) return fail(h, GPCA_ERR_BAD_ARG, "genotype matrix must have M > 0 SNPs and N > 0 samples");
    HIPCHK(hipSetDevice(h->device));
    stream_close(h);
    free_stats(h); free_ws(h); dfree(h->dG); dfree(h->dG2);
    h->M = M; h->N = N; h->Mpad = round_up(M, kGQRowsPerWave); h->pack_flags = 0;
    if (h->storage == GPCA_STORE_2BIT) {
        h->ldg = round_up(N, kSamplePad2bit); h->ld2 = h->ldg / 4; h->ld8 = h->ldg;
        if (!resident) return GPCA_OK;
        HIPCHK(hipMalloc((void**)&h->dG2, (size_t)h->Mpad * (size_t)h->ld2));
        if (h->Mpad > M) HIPCHK(hipMemsetAsync(h->dG2 + (size_t)M * (size_t)h->ld2, 0, (size_t)(h->Mpad - M) * (size_t)h->ld2, h->st));
        return GPCA_OK;
    }
    h->ldg = round_up(N, kSamplePad); h->ld2 = 0;
    // Row pitch vs HBM channel interleave: rows an EVEN multiple of 256 B apart (10 240 B for 10 000 samples) stream 2-5 % slower
    // than rows an odd multiple apart (profiles/r1_kbench_summary.md section 6): 8 rows of one DMA piece then spread over fewer
    // channels.  The pitch gets one extra 256-byte block in that case; the kernels never read past ldg.
    h->ld8 = ((h->ldg / 256) & 1) ? h->ldg : h->ldg + 256;
    if (getenv("GPCA_PITCH_PAD") && atoi(getenv("GPCA_PITCH_PAD")) == 0) h->ld8 = h->ldg;
    if (!resident) return GPCA_OK;
    HIPCHK(hipMalloc((void**)&h->dG, (size_t)h->Mpad * (size_t)h->ld8));
    if (h->Mpad > M) HIPCHK(hipMemsetAsync(h->dG + (size_t)M * (size_t)h->ld8, 0, (size_t)(h->Mpad - M) * (size_t)h->ld8, h->st));
    return GPCA_OK;
}

// rows per chunk of the bounded staging buffers (<= 256 MiB of int8 rows)
static int64_t pack_chunk_rows(gpca_handle* h) {
    int64_t r = ((int64_t)256 << 20) / h->ldg;
    if (r < 1) r = 1;
    return r < h->M ? r : h->M;
}
static int finish_pack_flags(gpca_handle* h, unsigned* d_flags, hipStream_t st) {
    unsigned f = 0;
    HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    h->pack_flags |= f;
    return GPCA_OK;
}

extern "C" int gpca_upload_genotypes_i8(gpca_handle* h, const int8_t* src, int64_t M, int64_t N, int64_t ld) {
    if (!h || !src || ld < N) return fail(h, GPCA_ERR_BAD_ARG, "gpca_upload_genotypes_i8: bad arguments");
    LOCK(h);
    CHK(alloc_genotypes(h, M, N));
    if (h->storage == GPCA_STORE_2BIT) {
        const int64_t cr = pack_chunk_rows(h);
        int8_t* scratch = nullptr; unsigned* d_flags = nullptr;
        HIPCHK(hipMalloc((void**)&scratch, (size_t)cr * (size_t)h->ldg));
        hipError_t e = hipMalloc((void**)&d_flags, 16);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 16, h->st);
        if (e == hipSuccess) e = hipMemsetAsync(scratch, 0, (size_t)cr * (size_t)h->ldg, h->st);
        for (int64_t r0 = 0; r0 < M && e == hipSuccess; r0 += cr) {
            const int64_t rows = std::min(cr, M - r0);
            e = hipMemcpy2DAsync(scratch, (size_t)h->ldg, src + r0 * ld, (size_t)ld, (size_t)N, (size_t)rows, hipMemcpyHostToDevice, h->st);
            if (e == hipSuccess) { launch_pack_i8(h->st, scratch, h->ldg, h->dG2 + (size_t)r0 * h->ld2, rows, N, h->ld2, d_flags); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipStreamSynchronize(h->st);   // the host source of the next chunk may be pageable
        }
        int rc = e == hipSuccess ? finish_pack_flags(h, d_flags, h->st) : GPCA_OK;
        (void)hipFree(scratch); (void)hipFree(d_flags);
        HIPCHK(e);
        return rc;
    }
    if (h->ld8 != N) HIPCHK(hipMemsetAsync(h->dG, 0, (size_t)M * (size_t)h->ld8, h->st));
    HIPCHK(hipMemcpy2DAsync(h->dG, (size_t)h->ld8, src, (size_t)ld, (size_t)N, (size_t)M, hipMemcpyHostToDevice, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    return GPCA_OK;
}

// The .bed payload travels in row chunks through a bounded device staging buffer (<= 256 MiB) and is recoded chunk by
// chunk, so the peak is the resident matrix + 256 MiB (a 250 GB .bed of 10M x 100k fits one MI355X as 2-bit codes) and no
// launch exceeds 2^32 work-items.
extern "C" int gpca_upload_bed2bit(gpca_handle* h, const uint8_t* bed_rows, int64_t M, int64_t N) {
    if (!h || !bed_rows) return fail(h, GPCA_ERR_BAD_ARG, "gpca_upload_bed2bit: bad arguments");
    LOCK(h);
    CHK(alloc_genotypes(h, M, N));
    const int64_t bpr = (N + 3) / 4;
    int64_t cr = ((int64_t)256 << 20) / bpr;
    cr = std::max<int64_t>(1, std::min(cr, M));
    uint8_t* d_bed = nullptr;
    HIPCHK(hipMalloc((void**)&d_bed, (size_t)cr * (size_t)bpr));
    hipError_t e = hipSuccess;
    for (int64_t r0 = 0; r0 < M && e == hipSuccess; r0 += cr) {
        const int64_t rows = std::min(cr, M - r0);
        e = hipMemcpyAsync(d_bed, bed_rows + (size_t)r0 * (size_t)bpr, (size_t)rows * (size_t)bpr, hipMemcpyHostToDevice, h->st);
        if (e != hipSuccess) break;
        if (h->storage == GPCA_STORE_2BIT) launch_bed_to_codes(h->st, d_bed, bpr, h->dG2 + (size_t)r0 * h->ld2, rows, N, h->ld2);   // stays 2-bit
        else launch_bed_decode(h->st, d_bed, bpr, h->dG + (size_t)r0 * h->ld8, rows, N, h->ld8);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->st);   // the staging buffer is reused; the host source may be pageable
    }
    (void)hipFree(d_bed);
    HIPCHK(e);
    return GPCA_OK;
}

extern "C" int gpca_synth_genotypes(gpca_handle* h, int64_t M, int64_t N, uint64_t seed, const uint32_t* thresh,
                                    int32_t P, int64_t snp_offset) {
    if (!h || !thresh || P <= 0) return fail(h, GPCA_ERR_BAD_ARG, "gpca_synth_genotypes: bad arguments");
    gpca_panel_source src{};
    src.kind = GPCA_PANEL_SYNTH; src.n_pop = P; src.thresh = thresh; src.seed = seed; src.snp_offset = snp_offset;
    return gpca_load_from_source(h, &src, M, N);
}

// ---- panel sources ------------------------------------------------------------------------------------------
static int check_source(gpca_handle* h, const gpca_panel_source* src, const char* who) {
    if (!src) return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": source is NULL");
    switch (src->kind) {
        case GPCA_PANEL_HOST_I8: case GPCA_PANEL_HOST_BED:
            if (!src->fill) return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": host panel source without a fill callback");
            return GPCA_OK;
        case GPCA_PANEL_SYNTH: case GPCA_PANEL_SYNTH16:
            if (!src->thresh || src->n_pop <= 0 || src->snp_offset < 0) return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": generator source needs thresh, n_pop > 0, snp_offset >= 0");
            return GPCA_OK;
        default: return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": unknown panel kind");
    }
}

// staging of one source for chunks of up to chunk_rows rows of the handle's current M x N matrix
static int filler_open(gpca_handle* h, Filler& f, const gpca_panel_source& src, int64_t chunk_rows, hipStream_t st) {
    filler_close(f);
    f.src = src; f.chunk_rows = chunk_rows; f.stage_idx = 0;
    f.open = true;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    if (src.kind == GPCA_PANEL_SYNTH || src.kind == GPCA_PANEL_SYNTH16) {
        const size_t tb = (size_t)h->M * (size_t)src.n_pop * sizeof(uint32_t);
        HIPCHK(hipMalloc((void**)&f.d_thresh, tb));
        HIPCHK(hipMemcpyAsync(f.d_thresh, src.thresh, tb, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));   // the caller's table may be freed after open
    }
    if (packed && (src.kind == GPCA_PANEL_SYNTH || src.kind == GPCA_PANEL_HOST_I8)) {
        HIPCHK(hipMalloc((void**)&f.d_scratch8, (size_t)chunk_rows * (size_t)h->ldg));
        HIPCHK(hipMemsetAsync(f.d_scratch8, 0, (size_t)chunk_rows * (size_t)h->ldg, st));   // pad columns stay 0
        HIPCHK(hipMalloc((void**)&f.d_flags, 16));
        HIPCHK(hipMemsetAsync(f.d_flags, 0, 16, st));
    }
    if (src.kind == GPCA_PANEL_HOST_BED) {
        f.stage_ld = (h->N + 3) / 4;
        HIPCHK(hipMalloc((void**)&f.d_raw, (size_t)chunk_rows * (size_t)f.stage_ld));
    }
    if (src.kind == GPCA_PANEL_HOST_I8) f.stage_ld = h->N;
    if (src.kind == GPCA_PANEL_HOST_I8 || src.kind == GPCA_PANEL_HOST_BED) {
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipHostMalloc(&f.h_stage[i], (size_t)chunk_rows * (size_t)f.stage_ld, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&f.ev_stage[i], hipEventDisableTiming));
        }
    }
    return GPCA_OK;
}

// rows [row0, row0 + rows) of the matrix -> dst (int8 rows of pitch ldg, or 2-bit rows of pitch ld2), enqueued on st
static int filler_fill(gpca_handle* h, Filler& f, int64_t row0, int64_t rows, void* dst, hipStream_t st) {
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const gpca_panel_source& s = f.src;
    switch (s.kind) {
        case GPCA_PANEL_SYNTH:
            if (packed) {
                launch_synth(st, f.d_scratch8, rows, h->N, h->ldg, s.snp_offset + row0, s.seed, f.d_thresh + (size_t)row0 * s.n_pop, s.n_pop);
                launch_pack_i8(st, f.d_scratch8, h->ldg, (uint8_t*)dst, rows, h->N, h->ld2, f.d_flags);
            } else launch_synth(st, (int8_t*)dst, rows, h->N, h->ld8, s.snp_offset + row0, s.seed, f.d_thresh + (size_t)row0 * s.n_pop, s.n_pop);
            HIPCHK(hipGetLastError());
            return GPCA_OK;
        case GPCA_PANEL_SYNTH16:
            launch_synth16(st, dst, packed ? 1 : 0, rows, h->N, packed ? h->ld2 : h->ld8, s.snp_offset + row0, s.seed, f.d_thresh + (size_t)row0 * s.n_pop, s.n_pop);
            HIPCHK(hipGetLastError());
            return GPCA_OK;
        case GPCA_PANEL_HOST_I8: case GPCA_PANEL_HOST_BED: {
            const int b = f.stage_idx; f.stage_idx ^= 1;
            if (f.stage_pending[b]) { HIPCHK(hipEventSynchronize(f.ev_stage[b])); f.stage_pending[b] = 0; }   // its last copy has left the buffer
            if (s.fill(s.user, row0, rows, f.h_stage[b], f.stage_ld) != 0) {
                char buf[160];
                snprintf(buf, sizeof buf, "panel source callback failed for rows [%lld, %lld)", (long long)row0, (long long)(row0 + rows));
                return fail(h, GPCA_ERR_BAD_ARG, buf);
            }
            if (s.kind == GPCA_PANEL_HOST_BED) {
                HIPCHK(hipMemcpyAsync(f.d_raw, f.h_stage[b], (size_t)rows * (size_t)f.stage_ld, hipMemcpyHostToDevice, st));
                if (packed) launch_bed_to_codes(st, f.d_raw, f.stage_ld, (uint8_t*)dst, rows, h->N, h->ld2);
                else launch_bed_decode(st, f.d_raw, f.stage_ld, (int8_t*)dst, rows, h->N, h->ld8);
                HIPCHK(hipGetLastError());
            } else if (packed) {
                HIPCHK(hipMemcpy2DAsync(f.d_scratch8, (size_t)h->ldg, f.h_stage[b], (size_t)f.stage_ld, (size_t)h->N, (size_t)rows, hipMemcpyHostToDevice, st));
                launch_pack_i8(st, f.d_scratch8, h->ldg, (uint8_t*)dst, rows, h->N, h->ld2, f.d_flags);
                HIPCHK(hipGetLastError());
            } else {
                HIPCHK(hipMemcpy2DAsync(dst, (size_t)h->ld8, f.h_stage[b], (size_t)f.stage_ld, (size_t)h->N, (size_t)rows, hipMemcpyHostToDevice, st));
            }
            HIPCHK(hipEventRecord(f.ev_stage[b], st)); f.stage_pending[b] = 1;
            return GPCA_OK;
        }
        default: return fail(h, GPCA_ERR_BAD_ARG, "unknown panel kind");
    }
}

extern "C" int gpca_load_from_source(gpca_handle* h, const gpca_panel_source* src, int64_t M, int64_t N) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    CHK(check_source(h, src, "gpca_load_from_source"));
    CHK(alloc_genotypes(h, M, N));
    const bool packed = h->storage == GPCA_STORE_2BIT;
    if (!packed && h->ld8 != N) HIPCHK(hipMemsetAsync(h->dG, 0, (size_t)M * (size_t)h->ld8, h->st));   // pad columns of host-copied rows
    const int64_t cr = pack_chunk_rows(h);
    Filler f;
    int rc = filler_open(h, f, *src, cr, h->st);
    for (int64_t r0 = 0; r0 < M && rc == GPCA_OK; r0 += cr) {
        const int64_t rows = std::min(cr, M - r0);
        void* dst = packed ? (void*)(h->dG2 + (size_t)r0 * h->ld2) : (void*)(h->dG + (size_t)r0 * h->ld8);
        rc = filler_fill(h, f, r0, rows, dst, h->st);
    }
    if (rc == GPCA_OK && f.d_flags) rc = finish_pack_flags(h, f.d_flags, h->st);
    if (hipStreamSynchronize(h->st) != hipSuccess && rc == GPCA_OK) rc = fail(h, GPCA_ERR_HIP, "gpca_load_from_source: stream failed");
    filler_close(f);
    return rc;
}

extern "C" int gpca_stream_open(gpca_handle* h, const gpca_panel_source* src, int64_t M, int64_t N, int64_t panel_rows,
                                int32_t ring_slots) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    CHK(check_source(h, src, "gpca_stream_open"));
    if (h->precision != GPCA_PREC_I8_EXACT) return fail(h, GPCA_ERR_BAD_ARG, "gpca_stream_open: streamed panels need GPCA_PREC_I8_EXACT (integer partial sums make the panel order irrelevant)");
    if (ring_slots == 0) ring_slots = 3;
    if (ring_slots < 2 || ring_slots > 16 || panel_rows < 0) return fail(h, GPCA_ERR_BAD_ARG, "gpca_stream_open: ring_slots must be in [2, 16], panel_rows >= 0");
    CHK(alloc_genotypes(h, M, N, /*resident=*/false));
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const int64_t row_bytes = packed ? h->ld2 : h->ld8;
    if (panel_rows == 0) {
        // K1 gives every wave 128 SNP rows and sweeps all samples with them: a panel needs gq_waves_target x 128 rows (131 072)
        // to fill the chip, however wide the rows are (a 1 GiB panel of 500k-sample rows holds 8k rows and leaves three quarters
        // of the CUs idle).  Take that many rows when the ring fits in half of the free HBM after the M- and N-sized workspace.
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const double workspace = (double)h->Mpad * (64 * 4 + 2 * 32 * kDigits + 48) + (double)h->ldg * (64 * 8 * 3 + 2 * 32 * kDigits + 32 * 8 * 4);
        const double budget = 0.5 * ((double)free_b - workspace);
        int64_t rows = (int64_t)h->gq_waves_target * kGQRowsPerWave;
        const int64_t fit = (int64_t)(budget / ((double)ring_slots * (double)row_bytes));
        if (rows > fit) rows = fit;
        if (src->kind == GPCA_PANEL_HOST_I8 || src->kind == GPCA_PANEL_HOST_BED) {
            // callback sources also need two pinned host staging panels: keep each within 2 GiB (such a source is bound by the
            // host link, ~55 GB/s, long before the row-parallel K1 runs out of rows)
            const int64_t host_ld = src->kind == GPCA_PANEL_HOST_BED ? (N + 3) / 4 : N;
            const int64_t cap = ((int64_t)2 << 30) / host_ld;
            if (rows > cap) rows = cap;
        }
        panel_rows = rows / kGQRowsPerWave * kGQRowsPerWave;
        if (panel_rows < kGQRowsPerWave) panel_rows = kGQRowsPerWave;
    }
    panel_rows = round_up(std::max<int64_t>(panel_rows, 1), kGQRowsPerWave);
    if (panel_rows > h->Mpad) panel_rows = h->Mpad;
    StreamState& sm = h->sm;
    sm.panel_rows = panel_rows;
    sm.n_panels = (int)((M + panel_rows - 1) / panel_rows);
    sm.ring = ring_slots; sm.seq = 0; sm.fused = 1;
    HIPCHK(hipStreamCreateWithFlags(&sm.st_fill, hipStreamNonBlocking));
    sm.on = true;   // from here on stream_close() releases whatever was set up
    int rc = GPCA_OK;
    for (int i = 0; i < ring_slots && rc == GPCA_OK; ++i) {
        void* p = nullptr; hipEvent_t a = nullptr, b = nullptr;
        hipError_t e = hipMalloc(&p, (size_t)panel_rows * (size_t)row_bytes);
        if (e == hipSuccess) { sm.slot.push_back(p); e = hipMemsetAsync(p, 0, (size_t)panel_rows * (size_t)row_bytes, sm.st_fill); }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
        if (e == hipSuccess) { sm.ev_filled.push_back(a); e = hipEventCreateWithFlags(&b, hipEventDisableTiming); }
        if (e == hipSuccess) { sm.ev_free.push_back(b); sm.free_pending.push_back(0); }
        if (e != hipSuccess) rc = fail(h, e == hipErrorOutOfMemory ? GPCA_ERR_OOM : GPCA_ERR_HIP, std::string("gpca_stream_open: panel ring: ") + hipGetErrorString(e));
    }
    if (rc == GPCA_OK) rc = filler_open(h, sm.fl, *src, panel_rows, sm.st_fill);
    if (rc == GPCA_OK && hipStreamSynchronize(sm.st_fill) != hipSuccess) rc = fail(h, GPCA_ERR_HIP, "gpca_stream_open: stream failed");
    if (rc != GPCA_OK) { std::string keep = h->err; stream_close(h); h->M = h->N = 0; h->err = keep; }
    return rc;
}

extern "C" int gpca_stream_set_fused(gpca_handle* h, int32_t fused) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!h->sm.on) return fail(h, GPCA_ERR_STATE, "gpca_stream_set_fused: no panel stream open");
    h->sm.fused = fused != 0;
    return GPCA_OK;
}

// fn(view) once for the resident matrix, or once per panel (generated / copied one panel ahead on the fill stream)
template <class F>
static int for_each_panel(gpca_handle* h, F&& fn) {
    if (!h->sm.on) { const PanelView pv{h->dG, h->dG2, 0, h->M, h->Mpad, 0}; return fn(pv); }
    StreamState& sm = h->sm;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const int64_t row_bytes = packed ? h->ld2 : h->ld8;
    for (int p = 0; p < sm.n_panels; ++p) {
        const int64_t row0 = (int64_t)p * sm.panel_rows;
        const int64_t rows = std::min(sm.panel_rows, h->M - row0);
        const int64_t rows_pad = round_up(rows, kGQRowsPerWave);
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

extern "C" int gpca_download_genotypes_i8(gpca_handle* h, int8_t* out, int64_t ld) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (h->sm.on) return fail(h, GPCA_ERR_STATE, "gpca_download_genotypes_i8: the matrix is streamed, not resident");
    if (!out || (!h->dG && !h->dG2) || ld < h->N) return fail(h, GPCA_ERR_BAD_ARG, "gpca_download_genotypes_i8: bad arguments / nothing resident");
    if (h->storage == GPCA_STORE_2BIT) {   // debug/test path: copy the packed rows and unpack on the host
        HIPCHK(hipStreamSynchronize(h->st));
        std::vector<uint8_t> row((size_t)h->ld2);
        static const int8_t lut[4] = {0, 1, 2, -127};
        for (int64_t i = 0; i < h->M; ++i) {
            HIPCHK(hipMemcpy(row.data(), h->dG2 + (size_t)i * h->ld2, (size_t)h->ld2, hipMemcpyDeviceToHost));
            for (int64_t n = 0; n < h->N; ++n) out[i * ld + n] = lut[(row[(size_t)(n >> 2)] >> (2 * (n & 3))) & 3];
        }
        return GPCA_OK;
    }
    HIPCHK(hipMemcpy2D(out, (size_t)ld, h->dG, (size_t)h->ld8, (size_t)h->N, (size_t)h->M, hipMemcpyDeviceToHost));
    return GPCA_OK;
}

extern "C" int gpca_dims(gpca_handle* h, int64_t* M, int64_t* N) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (M) *M = h->M;
    if (N) *N = h->N;
    return GPCA_OK;
}

// ---- a1 ---------------------------------------------------------------------------------------------------
static int alloc_stats(gpca_handle* h) {
    const size_t M = (size_t)h->Mpad;   // pad rows: r = b = 0
    if (h->d_mu) return GPCA_OK;
    HIPCHK(hipMalloc((void**)&h->d_mu, M * 4)); HIPCHK(hipMalloc((void**)&h->d_sigma, M * 4));
    HIPCHK(hipMalloc((void**)&h->d_r, M * 4)); HIPCHK(hipMalloc((void**)&h->d_b, M * 4));
    HIPCHK(hipMalloc((void**)&h->d_keep, M)); HIPCHK(hipMalloc((void**)&h->d_reason, M));
    HIPCHK(hipMalloc((void**)&h->d_counts, M * 16)); HIPCHK(hipMalloc((void**)&h->d_flags, 16));
    HIPCHK(hipMemsetAsync(h->d_counts, 0, M * 16, h->st)); HIPCHK(hipMemsetAsync(h->d_r, 0, M * 4, h->st)); HIPCHK(hipMemsetAsync(h->d_b, 0, M * 4, h->st));
    HIPCHK(hipMemsetAsync(h->d_mu, 0, M * 4, h->st)); HIPCHK(hipMemsetAsync(h->d_sigma, 0, M * 4, h->st)); HIPCHK(hipMemsetAsync(h->d_keep, 0, M, h->st));
    return GPCA_OK;
}

static int refresh_pca_rows(gpca_handle* h) {
    std::vector<uint8_t> keep((size_t)h->M);
    HIPCHK(hipMemcpy(keep.data(), h->d_keep, (size_t)h->M, hipMemcpyDeviceToHost));
    h->pca_rows.clear();
    for (int64_t i = 0; i < h->M; ++i) if (keep[(size_t)i]) h->pca_rows.push_back(i);
    h->n_pca = (int64_t)h->pca_rows.size();
    dfree(h->d_pca_rows);
    if (h->n_pca) {
        HIPCHK(hipMalloc((void**)&h->d_pca_rows, (size_t)h->n_pca * 8));
        HIPCHK(hipMemcpy(h->d_pca_rows, h->pca_rows.data(), (size_t)h->n_pca * 8, hipMemcpyHostToDevice));
    }
    return GPCA_OK;
}

extern "C" int gpca_snp_stats(gpca_handle* h, const gpca_qc_config* qc, float* mu, float* sigma, uint8_t* keep) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!have_genotypes(h)) return fail(h, GPCA_ERR_STATE, "gpca_snp_stats: no genotypes resident and no panel stream open");
    HIPCHK(hipSetDevice(h->device));
    CHK(alloc_stats(h));
    QcParams q{0.0, 0.0, 1.0};
    if (qc) { q.min_call_rate = qc->min_snp_call_rate; q.min_maf = qc->min_snp_maf; q.max_hwe_p = qc->max_snp_hwe_p_value; }
    HIPCHK(hipMemsetAsync(h->d_flags, 0, 16, h->st));
    {
        ScopedTimer t(h, "snp_stats", 0.0, (double)h->M * (double)h->N);
        CHK(for_each_panel(h, [&](const PanelView& pv) -> int {
            const int64_t o = pv.row0;
            if (h->storage == GPCA_STORE_2BIT)
                launch_snp_stats_2bit(h->st, pv.g2, pv.rows, h->N, h->ld2, q, h->d_mu + o, h->d_sigma + o, h->d_r + o, h->d_b + o, h->d_keep + o,
                                      h->d_reason + o, h->d_counts + 4 * o, h->d_flags);
            else
                launch_snp_stats(h->st, pv.g8, pv.rows, h->N, h->ld8, q, h->d_mu + o, h->d_sigma + o, h->d_r + o, h->d_b + o, h->d_keep + o,
                                 h->d_reason + o, h->d_counts + 4 * o, h->d_flags);
            HIPCHK(hipGetLastError());
            return GPCA_OK;
        }));
    }
    HIPCHK(hipMemcpyAsync(&h->flags, h->d_flags, 4, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    if (h->sm.on && h->sm.fl.d_flags) CHK(finish_pack_flags(h, h->sm.fl.d_flags, h->sm.st_fill));   // int8 panels packed on the way in
    h->flags |= h->pack_flags;   // 2-bit mode: values outside {0,1,2,-127} were seen (and stored as missing) at upload
    CHK(refresh_pca_rows(h));
    h->have_stats = true; h->have_rsvd = false;
    if (mu) HIPCHK(hipMemcpy(mu, h->d_mu, (size_t)h->M * 4, hipMemcpyDeviceToHost));
    if (sigma) HIPCHK(hipMemcpy(sigma, h->d_sigma, (size_t)h->M * 4, hipMemcpyDeviceToHost));
    if (keep) HIPCHK(hipMemcpy(keep, h->d_keep, (size_t)h->M, hipMemcpyDeviceToHost));
    return GPCA_OK;
}

extern "C" int gpca_get_snp_qc_detail(gpca_handle* h, uint32_t* counts, uint8_t* reason) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_get_snp_qc_detail: run gpca_snp_stats first");
    if (counts) HIPCHK(hipMemcpy(counts, h->d_counts, (size_t)h->M * 16, hipMemcpyDeviceToHost));
    if (reason) HIPCHK(hipMemcpy(reason, h->d_reason, (size_t)h->M, hipMemcpyDeviceToHost));
    return GPCA_OK;
}

extern "C" int gpca_get_standardization(gpca_handle* h, float* mu, float* sigma, uint8_t* keep) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_get_standardization: run gpca_snp_stats or gpca_set_standardization first");
    HIPCHK(hipStreamSynchronize(h->st));
    if (mu) HIPCHK(hipMemcpy(mu, h->d_mu, (size_t)h->M * 4, hipMemcpyDeviceToHost));
    if (sigma) HIPCHK(hipMemcpy(sigma, h->d_sigma, (size_t)h->M * 4, hipMemcpyDeviceToHost));
    if (keep) HIPCHK(hipMemcpy(keep, h->d_keep, (size_t)h->M, hipMemcpyDeviceToHost));
    return GPCA_OK;
}

extern "C" int gpca_set_standardization(gpca_handle* h, const float* mu, const float* sigma, const uint8_t* keep) {
    if (!h || !mu || !sigma) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_standardization: mu and sigma are required");
    LOCK(h);
    if (!have_genotypes(h)) return fail(h, GPCA_ERR_STATE, "gpca_set_standardization: no genotypes resident and no panel stream open");
    HIPCHK(hipSetDevice(h->device));
    // a stats pass supplies the missing/invalid flags for the rows the caller keeps
    if (!h->have_stats) { gpca_qc_config none{0.0, 0.0, 1.0}; CHK(gpca_snp_stats(h, &none, nullptr, nullptr, nullptr)); }
    const size_t M = (size_t)h->M;
    HIPCHK(hipStreamSynchronize(h->st));   // blocking copies below run on the null stream
    HIPCHK(hipMemcpy(h->d_mu, mu, M * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_sigma, sigma, M * 4, hipMemcpyHostToDevice));
    if (keep) HIPCHK(hipMemcpy(h->d_keep, keep, M, hipMemcpyHostToDevice));
    else HIPCHK(hipMemsetAsync(h->d_keep, 1, M, h->st));
    launch_set_scale(h->st, h->M, h->d_mu, h->d_sigma, h->d_keep, h->d_r, h->d_b);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->st));
    CHK(refresh_pca_rows(h));
    // recompute flags for the caller's keep set from the counts of the stats pass
    std::vector<uint32_t> counts(M * 4);
    HIPCHK(hipMemcpy(counts.data(), h->d_counts, M * 16, hipMemcpyDeviceToHost));
    h->flags = 0;
    for (int64_t i : h->pca_rows) {
        const uint32_t* c = &counts[(size_t)i * 4];
        if ((int64_t)c[0] != h->N) h->flags |= 1u;
        if ((uint64_t)c[1] + c[2] + c[3] != c[0]) h->flags |= 2u;
    }
    h->have_rsvd = false;
    return GPCA_OK;
}

extern "C" double gpca_hwe_chi_squared_p_value(uint64_t n1h, uint64_t nhet, uint64_t n2h) {
    // same branches as prepare.rs:1641-1745 (host restatement; device copy: kernels.hip hwe_p_dev)
    const uint64_t tot = n1h + nhet + n2h;
    if (tot == 0) return 1.0;
    const double c1 = 2.0 * (double)n1h + (double)nhet, c2 = 2.0 * (double)n2h + (double)nhet;
    const double ta = c1 + c2;
    if (ta <= 1e-9) return 1.0;
    const double f1 = c1 / ta, f2 = c2 / ta;
    if (f1 < 1e-9 || f2 < 1e-9) return 1.0;
    if (std::fabs(f1 + f2 - 1.0) > 1e-6) return 1.0;
    const double e1 = f1 * f1 * (double)tot, eh = 2.0 * f1 * f2 * (double)tot, e2 = f2 * f2 * (double)tot;
    double chi = 0.0;
    const double MINE = 1e-9;
    if (e1 > MINE) { double d = (double)n1h - e1; chi += d * d / e1; } else if ((double)n1h > MINE) chi = INFINITY;
    if (std::isfinite(chi)) { if (eh > MINE) { double d = (double)nhet - eh; chi += d * d / eh; } else if ((double)nhet > MINE) chi = INFINITY; }
    if (std::isfinite(chi)) { if (e2 > MINE) { double d = (double)n2h - e2; chi += d * d / e2; } else if ((double)n2h > MINE) chi = INFINITY; }
    if (std::isnan(chi)) return 1.0;
    if (chi == INFINITY) return 0.0;
    const double cdf = std::erf(std::sqrt(chi * 0.5));
    if (std::isnan(cdf)) return 1.0;
    const double p = 1.0 - cdf;
    return p > 0.0 ? p : 0.0;
}


// ---- a2 ---------------------------------------------------------------------------------------------------
extern "C" int64_t gpca_num_pca_snps(gpca_handle* h) { if (!h) return 0; LOCK(h); return h->have_stats ? h->n_pca : 0; }
extern "C" int64_t gpca_num_qc_samples(gpca_handle* h) { if (!h) return 0; LOCK(h); return h->N; }
extern "C" int gpca_get_pca_snp_rows(gpca_handle* h, int64_t* rows) {
    if (!h || !rows) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_get_pca_snp_rows: run gpca_snp_stats first");
    std::copy(h->pca_rows.begin(), h->pca_rows.end(), rows);
    return GPCA_OK;
}

// The reference's solver calls this once per strip from many rayon workers (prepare.rs:1838): the id lists, the output block
// and the error word live in scratch that persists on the handle (grown on demand, never freed per call).
extern "C" int gpca_standardize_block(gpca_handle* h, const int64_t* snp_ids, int64_t ns, const int64_t* sample_ids,
                                      int64_t nj, float* out) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_standardize_block: run gpca_snp_stats first");
    if (h->sm.on) return fail(h, GPCA_ERR_STATE, "gpca_standardize_block: the pull API needs a resident matrix (this handle streams panels)");
    if (ns < 0 || nj < 0) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: negative block size");
    if (ns == 0 || nj == 0) return GPCA_OK;  // prepare.rs:1848-1850: empty block, nothing to fill
    if (!snp_ids || !sample_ids || !out) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: NULL pointer");
    HIPCHK(hipSetDevice(h->device));
    std::vector<int64_t> rows((size_t)ns);
    for (int64_t a = 0; a < ns; ++a) {
        if (snp_ids[a] < 0 || snp_ids[a] >= h->n_pca) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: PcaSnpId out of range");
        rows[(size_t)a] = h->pca_rows[(size_t)snp_ids[a]];
    }
    for (int64_t c = 0; c < nj; ++c)
        if (sample_ids[c] < 0 || sample_ids[c] >= h->N) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: QcSampleId out of range");
    CHK(ensure(h, h->d_blk_rows, h->cap_blk_rows, (size_t)ns));
    CHK(ensure(h, h->d_blk_cols, h->cap_blk_cols, (size_t)nj));
    CHK(ensure(h, h->d_blk_out, h->cap_blk_out, (size_t)ns * (size_t)nj));
    if (!h->d_blk_err) HIPCHK(hipMalloc((void**)&h->d_blk_err, 8));
    unsigned long long err_idx = ~0ull;
    HIPCHK(hipMemcpyAsync(h->d_blk_rows, rows.data(), (size_t)ns * 8, hipMemcpyHostToDevice, h->st));
    HIPCHK(hipMemcpyAsync(h->d_blk_cols, sample_ids, (size_t)nj * 8, hipMemcpyHostToDevice, h->st));
    HIPCHK(hipMemcpyAsync(h->d_blk_err, &err_idx, 8, hipMemcpyHostToDevice, h->st));
    if (h->storage == GPCA_STORE_2BIT) launch_standardize_block_2bit(h->st, h->dG2, h->ld2, h->d_mu, h->d_sigma, h->d_blk_rows, ns, h->d_blk_cols, nj, h->d_blk_out, h->d_blk_err);
    else launch_standardize_block(h->st, h->dG, h->ld8, h->d_mu, h->d_sigma, h->d_blk_rows, ns, h->d_blk_cols, nj, h->d_blk_out, h->d_blk_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&err_idx, h->d_blk_err, 8, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipMemcpyAsync(out, h->d_blk_out, (size_t)ns * (size_t)nj * 4, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    if (err_idx != ~0ull) {
        const int64_t a = (int64_t)(err_idx / (unsigned long long)nj), c = (int64_t)(err_idx % (unsigned long long)nj);
        char buf[400];  // wording of prepare.rs:1910-1911
        snprintf(buf, sizeof buf,
                 "Unexpected missing genotype (-127i8) in SnpBlockData for PCA SNP ID %lld (original BIM index %lld), "
                 "requested sample index %lld. This should have been filtered by QC.",
                 (long long)snp_ids[a], (long long)rows[(size_t)a], (long long)sample_ids[c]);
        return fail(h, GPCA_ERR_MISSING_GENOTYPE, buf);
    }
    return GPCA_OK;
}

// ---- e: exchange step ------------------------------------------------------------------------------------
extern "C" int gpca_comm_get_unique_id(void* out_id) {
    if (!out_id) return GPCA_ERR_BAD_ARG;
    if (!g_rccl.load()) return fail(nullptr, GPCA_ERR_RCCL, "librccl.so could not be loaded");
    ncclUniqueId_t id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, GPCA_ERR_RCCL, "ncclGetUniqueId failed");
    memcpy(out_id, &id, GPCA_UNIQUE_ID_BYTES);
    return GPCA_OK;
}

extern "C" int gpca_comm_init(gpca_handle* h, int32_t world, int32_t rank, const void* unique_id, int64_t snp_offset) {
    if (!h || world < 1 || rank < 0 || rank >= world || !unique_id || snp_offset < 0)
        return fail(h, GPCA_ERR_BAD_ARG, "gpca_comm_init: bad arguments");
    LOCK(h);
    if (!g_rccl.load()) return fail(h, GPCA_ERR_RCCL, "librccl.so could not be loaded");
    HIPCHK(hipSetDevice(h->device));
    if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId_t id;
    memcpy(&id, unique_id, GPCA_UNIQUE_ID_BYTES);
    const int rc = g_rccl.CommInitRank(&h->comm, world, id, rank);
    if (rc != 0) {
        std::string m = "ncclCommInitRank failed: ";
        m += g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?";
        return fail(h, GPCA_ERR_RCCL, m);
    }
    if (!h->d_status) HIPCHK(hipMalloc((void**)&h->d_status, 16 * sizeof(double)));   // allocated here so that the agreement itself cannot run out of memory
    if (!h->h_status) HIPCHK(hipHostMalloc((void**)&h->h_status, 32 * sizeof(double), hipHostMallocDefault));
    h->world = world; h->rank = rank; h->snp_offset = snp_offset; h->hook = nullptr;
    return GPCA_OK;
}

extern "C" int gpca_set_allreduce_hook(gpca_handle* h, gpca_allreduce_fn fn, void* user, int32_t world, int32_t rank,
                                       int64_t snp_offset) {
    if (!h || world < 1 || rank < 0 || rank >= world || snp_offset < 0) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_allreduce_hook: bad arguments");
    LOCK(h);
    HIPCHK(hipSetDevice(h->device));
    if (!h->d_status) HIPCHK(hipMalloc((void**)&h->d_status, 16 * sizeof(double)));
    if (!h->h_status) HIPCHK(hipHostMalloc((void**)&h->h_status, 32 * sizeof(double), hipHostMallocDefault));
    h->hook = fn; h->hook_user = user; h->world = world; h->rank = rank; h->snp_offset = snp_offset;
    return GPCA_OK;
}

static inline bool multi_rank(const gpca_handle* h) { return h->world > 1 || h->hook != nullptr; }

// in-place sum of a device f64 buffer across the ranks that share the sharded matrix
static int allreduce_f64(gpca_handle* h, double* dbuf, int64_t count) {
    if (!multi_rank(h)) return GPCA_OK;
    if (h->hook) {
        h->hook_buf.resize((size_t)count);
        HIPCHK(hipMemcpyAsync(h->hook_buf.data(), dbuf, (size_t)count * 8, hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        if (h->hook(h->hook_user, h->hook_buf.data(), count) != 0) return fail(h, GPCA_ERR_RCCL, "all-reduce hook reported failure");
        HIPCHK(hipMemcpyAsync(dbuf, h->hook_buf.data(), (size_t)count * 8, hipMemcpyHostToDevice, h->st));
        return GPCA_OK;
    }
    if (!h->comm) return fail(h, GPCA_ERR_STATE, "world > 1 but no communicator: call gpca_comm_init");
    ScopedTimer t(h, "allreduce", 0.0, (double)count * 8.0);
    const int rc = g_rccl.AllReduce(dbuf, dbuf, (size_t)count, kNcclFloat64, kNcclSum, h->comm, h->st);
    if (rc != 0) {
        std::string m = "ncclAllReduce failed: ";
        m += g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?";
        return fail(h, GPCA_ERR_RCCL, m);
    }
    return GPCA_OK;
}

// wait for the engine's stream: busy-poll (lowest wake-up latency, one host core spins) or hipStreamSynchronize
static hipError_t stream_wait(gpca_handle* h) {
    if (!h->spin_sync) return hipStreamSynchronize(h->st);
    hipError_t e;
    while ((e = hipStreamQuery(h->st)) == hipErrorNotReady) {}
    return e;
}

// Row-sharded runs: every rank contributes its local status (0 or a negative gpca_status) to a 16-slot histogram that is
// summed through the same transport as the sketch; all ranks then return the most severe (smallest) status any rank saw.
// A rank whose own shard is clean therefore leaves the call with the failing rank's code instead of waiting in a collective
// that the failing rank never enters.  Single-rank handles return local_rc untouched (no device work).
static int agree_status(gpca_handle* h, int local_rc, const char* where) {
    if (!multi_rank(h)) return local_rc;
    const std::string own = h->err;
    if (!h->d_status || !h->h_status) return fail(h, GPCA_ERR_STATE, "agree_status: no status buffer (gpca_comm_init / gpca_set_allreduce_hook allocate it)");
    double* v = h->h_status + 16;
    for (int i = 0; i < 16; ++i) h->h_status[i] = 0.0;
    h->h_status[local_rc == GPCA_OK ? 0 : std::min(15, -local_rc)] = 1.0;
    // pinned staging both ways and one wait: H2D, exchange and D2H are stream-ordered
    HIPCHK(hipMemcpyAsync(h->d_status, h->h_status, 16 * sizeof(double), hipMemcpyHostToDevice, h->st));
    CHK(allreduce_f64(h, h->d_status, 16));
    HIPCHK(hipMemcpyAsync(v, h->d_status, 16 * sizeof(double), hipMemcpyDeviceToHost, h->st));
    HIPCHK(stream_wait(h));
    int agreed = GPCA_OK;
    for (int i = 15; i >= 1; --i) if (v[i] > 0.5) agreed = -i;    // ends on the smallest index = GPCA_ERR_BAD_ARG first ... any is fine, all ranks pick the same
    if (agreed == GPCA_OK) return GPCA_OK;
    if (agreed == local_rc) { h->err = own; return agreed; }
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %d of %d rank(s) reported \"%s\" (%s); this rank (%d): %s%s%s", where, (int)(v[-agreed] + 0.5), h->world,
             gpca_status_string(agreed), "all ranks leave the call together", h->rank, local_rc == GPCA_OK ? "ok" : gpca_status_string(local_rc),
             local_rc == GPCA_OK ? "" : " -- ", local_rc == GPCA_OK ? "" : own.c_str());
    return fail(h, agreed, buf);
}

// ---- small dense (host, f64) ----------------------------------------------------------------------------------
// Symmetric eigenproblem of the l x l Gram of the projection (l <= 64): Householder tridiagonalisation + implicit QL
// (the EISPACK tred2 / tql2 pair).  It replaced a cyclic Jacobi solver: at l = 30 Jacobi's ~8 sweeps of 435 rotations kept
// the stream idle for ~190 us per call while the host worked; this pair needs ~20 us.  (The parity checker of tests/ uses LAPACK,
// oracle/oracle.py:rsvd -- no code in common.)  A: symmetric, row-major, destroyed; V: eigenvectors in columns; w: eigenvalues,
// sorted descending.
static void tred2(int n, double* V, double* d, double* e) {
    for (int j = 0; j < n; ++j) d[j] = V[(n - 1) * n + j];
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = V[(i - 1) * n + j]; V[i * n + j] = 0.0; V[j * n + i] = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g; h -= f * g; d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j]; V[j * n + i] = f; g = e[j] + V[j * n + j] * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += V[k * n + j] * d[k]; e[k] += V[k * n + j] * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) V[k * n + j] -= (f * e[k] + g * d[k]);
                d[j] = V[(i - 1) * n + j]; V[i * n + j] = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {        // accumulate the transformations
        V[(n - 1) * n + i] = V[i * n + i]; V[i * n + i] = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = V[k * n + (i + 1)] / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += V[k * n + (i + 1)] * V[k * n + j];
                for (int k = 0; k <= i; ++k) V[k * n + j] -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) V[k * n + (i + 1)] = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = V[(n - 1) * n + j]; V[(n - 1) * n + j] = 0.0; }
    V[(n - 1) * n + (n - 1)] = 1.0; e[0] = 0.0;
}
static void tql2(int n, double* V, double* d, double* e) {
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n - 1 && std::fabs(e[m]) > eps * tst1) ++m;     // e[n-1] = 0 ends the search
        if (m > l) {
            int iter = 0;
            do {
                ++iter;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r); d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i]; h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r; s = e[i] / r; c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = V[k * n + i + 1];
                        V[k * n + i + 1] = s * V[k * n + i] + c * h;
                        V[k * n + i] = c * V[k * n + i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p; d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] += f; e[l] = 0.0;
    }
}
static void host_eigh_desc(std::vector<double>& A, std::vector<double>& V, std::vector<double>& w, int n) {
    std::vector<double> e((size_t)n);
    V = A;
    if (n == 1) { w[0] = A[0]; V[0] = 1.0; return; }
    tred2(n, V.data(), w.data(), e.data());
    tql2(n, V.data(), w.data(), e.data());
    for (int i = 0; i < n - 1; ++i) {        // selection sort, descending
        int m = i;
        for (int j = i + 1; j < n; ++j) if (w[j] > w[m]) m = j;
        if (m != i) { std::swap(w[i], w[m]); for (int k = 0; k < n; ++k) std::swap(V[k * n + i], V[k * n + m]); }
    }
}
// test hook (host only, no GPU): eigen-decomposition of a symmetric n x n row-major matrix, eigenvalues descending
extern "C" int gpca_host_eigh_desc(const double* a_sym, int32_t n, double* w, double* v) {
    if (!a_sym || !w || !v || n < 1 || n > 64) return GPCA_ERR_BAD_ARG;
    std::vector<double> A(a_sym, a_sym + (size_t)n * n), V((size_t)n * n), W((size_t)n);
    host_eigh_desc(A, V, W, n);
    std::copy(W.begin(), W.end(), w); std::copy(V.begin(), V.end(), v);
    return GPCA_OK;
}

// ---- rsvd stages -----------------------------------------------------------------------------------------------
static int stage_sum_c(gpca_handle* h, int64_t parts) {
    launch_sum_partials_f32(h->st, h->d_cpart, parts, h->L, h->d_c, h->d_scratch64);
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}

constexpr size_t kPlaneBytesPerBlock = (size_t)kDigits * 1024;   // digit planes of one 32-row (or 32-sample) block

// K2 of one 32-column half over one panel: Ypart = (digit planes of T')^T G, exact integers
static int k2_panel(gpca_handle* h, const PanelView& pv, const int8_t* Td_half, const Gtt8Plan& plan) {
    const int8_t* Td = Td_half + (size_t)(pv.row0 >> 5) * kPlaneBytesPerBlock;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    if (h->lds_planes && h->gtt_dma && !packed) {
        const int e = launch_gtt_d(h->st, pv.g8, h->ld8, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->ko);
        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gtt_d launch failed (hip error " + std::to_string(e) + ")");
    } else if (((h->lds_planes && h->gtt_dma) || h->nd == 3) && packed) {   // (three planes: only this kernel)
        const int e = launch_gtt_p(h->st, pv.g2, h->ld2, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->nd, h->ko);
        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gtt_p launch failed (hip error " + std::to_string(e) + ")");
    } else if (h->lds_planes) launch_gtt_x(h->st, packed ? (const void*)pv.g2 : (const void*)pv.g8, packed, packed ? h->ld2 : h->ld8, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->ko);
    else if (packed) launch_gtt_2bit(h->st, pv.g2, h->ld2, pv.rows_pad, h->ldg, Td, h->dYpart64, plan);
    else launch_gtt_i8(h->st, pv.g8, h->ld8, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->ko);
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}

// Y = A^T T  (T' = r o T already in dT, c = b^T T in d_c); rank-local part, the exchange of Y follows in the caller
static int stage_AtT_local(gpca_handle* h) {
    const double elems = (double)h->M * (double)h->N;
    if (h->precision == GPCA_PREC_I8_EXACT) {
        // T' (f32 row-major in dT) -> digit planes; exact int8 product; integer partials summed exactly in f64.
        // The kernels are 32 columns wide: a 64-column sketch (32 < l <= 64) runs as two column halves over the same genotypes.
        const int L = h->L, halves = L / 32;
        const size_t td_half = (size_t)h->Mpad * 32 * kDigits;
        for (int hf = 0; hf < halves; ++hf) {
            const float* Th = h->dT + 32 * hf;
            double* tsc = h->d_tscale + 32 * hf; double* tin = h->d_tinv + 32 * hf;
            if (h->apart_valid) launch_quantize_f32_premax(h->st, Th, h->Mpad, h->Mpad, h->apart_src[hf], h->apart_parts, tsc, tin, h->dTd + hf * td_half, 0, h->nd, L);
            else launch_quantize_f32(h->st, Th, h->Mpad, h->Mpad, h->d_part64, tsc, tin, h->dTd + hf * td_half, 0, h->nd, L);
            HIPCHK(hipGetLastError());
        }
        h->apart_valid = false;
        const bool streamed = h->sm.on;
        const size_t yint_half = (size_t)h->N * 32;
        {
            // resident: one record per launch (the roofline figure of bench.py); streamed: one record per sweep over the panels
            const double by = h->storage == GPCA_STORE_2BIT ? elems / 4 : elems;
            ScopedTimer sweep(h, "gemm_GtT", 2.0 * elems * h->l, by * halves, nullptr, streamed);
            CHK(for_each_panel(h, [&](const PanelView& pv) -> int {
                const Gtt8Plan plan = streamed ? gtt8_plan(pv.rows_pad, h->ldg, h->gtt_waves_target) : h->plan8;
                for (int hf = 0; hf < halves; ++hf) {
                    {
                        ScopedTimer t(h, "gemm_GtT", 2.0 * elems * h->l / halves, by, nullptr, !streamed);
                        CHK(k2_panel(h, pv, h->dTd + hf * td_half, plan));
                    }
                    if (streamed) launch_accum_y_i8(h->st, h->dYpart64, plan.W, h->ldg, h->N, h->d_yint + hf * yint_half, pv.index == 0);
                    else launch_reduce_y_i8(h->st, h->dYpart64, plan.W, h->ldg, h->N, h->d_c + 32 * hf, h->d_tscale + 32 * hf, h->dY + 32 * hf, L);
                    HIPCHK(hipGetLastError());
                }
                return GPCA_OK;
            }));
        }
        if (streamed)
            for (int hf = 0; hf < halves; ++hf) {
                launch_finish_y_i8(h->st, h->d_yint + hf * yint_half, h->N, h->d_c + 32 * hf, h->d_tscale + 32 * hf, h->dY + 32 * hf, L);
                HIPCHK(hipGetLastError());
            }
        return GPCA_OK;
    }
    {
        const bool packed = h->storage == GPCA_STORE_2BIT;
        ScopedTimer t(h, "gemm_GtT", 2.0 * elems * h->l, packed ? elems / 4 : elems);
        launch_gtt_f32(h->st, packed ? (const void*)h->dG2 : (const void*)h->dG, packed, packed ? h->ld2 : h->ld8, h->Mpad, h->ldg, h->dTb, h->L, h->dYpart, h->plan);
    }
    HIPCHK(hipGetLastError());
    launch_reduce_y(h->st, h->dYpart, h->plan.W, h->ldg, h->N, h->L, h->d_c, h->dY);
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}

// T = A Q (scale_out: r o T and c)
static int stage_AQ(gpca_handle* h, int scale_out) {
    const double elems = (double)h->M * (double)h->N;
    if (h->precision == GPCA_PREC_I8_EXACT) {
        const int L = h->L, halves = L / 32;
        const bool packed = h->storage == GPCA_STORE_2BIT, streamed = h->sm.on;
        const size_t qhalf = (size_t)h->ldg * 32 * kDigits;            // digit planes of one 32-column half of Q
        const size_t chalf = (size_t)h->Mpad;                           // per-unit partials of c of one half: [Mpad / 32][32]
        const size_t ahalf = (size_t)h->gqplan.waves * 32;              // per-wave abs-max partials of one launch
        if (streamed && scale_out) HIPCHK(hipMemsetAsync(h->d_amax_run, 0, 64 * 8, h->st));
        {
            const double by = packed ? elems / 4 : elems;
            ScopedTimer sweep(h, "gemm_GQ", 2.0 * elems * h->l, by * halves, nullptr, streamed);
            CHK(for_each_panel(h, [&](const PanelView& pv) -> int {
                const GqPlan plan = streamed ? gq_plan(pv.rows_pad, h->gq_waves_target) : h->gqplan;
                for (int hf = 0; hf < halves; ++hf) {
                    const int8_t* Qd = h->dQd + hf * qhalf;
                    const double* qsc = h->d_qscale + 32 * hf;
                    const float* s32 = h->d_s32 + 32 * hf;
                    const float* rr = h->d_r + pv.row0; const float* bb = h->d_b + pv.row0;
                    float* Th = h->dT + (size_t)pv.row0 * L + 32 * hf;
                    float* cp = h->d_cpart + hf * chalf + (size_t)pv.row0;          // (row0 / 32) units x 32 columns
                    double* ap = h->d_apart + hf * ahalf;
                    ScopedTimer t(h, "gemm_GQ", 2.0 * elems * h->l / halves, by, nullptr, !streamed);
                    if (packed) launch_gq_2bit(h->st, pv.g2, h->ld2, plan, h->ldg, Qd, qsc, rr, bb, s32, Th, cp, ap, scale_out, h->nd, L);
                    else if (h->lds_planes && h->gq_dma) {
                        const int e = launch_gq_d(h->st, pv.g8, h->ld8, plan, h->ldg, Qd, qsc, rr, bb, s32, Th, cp, ap, scale_out, L, h->ko);
                        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gq_d launch failed (hip error " + std::to_string(e) + ")");
                    }
                    else if (h->lds_planes) launch_gq_x(h->st, pv.g8, h->ld8, plan, h->ldg, Qd, qsc, rr, bb, s32, Th, cp, ap, scale_out, L, h->ko);
                    else launch_gq_i8(h->st, pv.g8, h->ld8, plan, h->N, Qd, qsc, rr, bb, s32, Th, cp, scale_out, L, h->ko);
                    HIPCHK(hipGetLastError());
                    if (streamed && scale_out && (packed || h->lds_planes)) {   // (streamed: the per-launch timer above is disabled) fold this panel's column abs-max before the next launch reuses ap
                        launch_absmax_fold(h->st, ap, plan.waves, h->d_amax_run + 32 * hf);
                        HIPCHK(hipGetLastError());
                    }
                }
                return GPCA_OK;
            }));
        }
        if (scale_out)
            for (int hf = 0; hf < halves; ++hf) {   // c = b^T T of this half: one partial per 32-row unit, summed in a fixed order
                launch_sum_partials_f32(h->st, h->d_cpart + hf * chalf, h->Mpad / 32, 32, h->d_c + 32 * hf, h->d_scratch64);
                HIPCHK(hipGetLastError());
            }
        h->apart_valid = scale_out != 0 && (packed || h->lds_planes);   // (k_gq_i8 has no abs-max epilogue)
        for (int hf = 0; hf < 2; ++hf) h->apart_src[hf] = streamed ? h->d_amax_run + 32 * hf : h->d_apart + hf * ahalf;
        h->apart_parts = streamed ? 1 : h->gqplan.waves;
        return GPCA_OK;
    }
    {
        const bool packed = h->storage == GPCA_STORE_2BIT;
        ScopedTimer t(h, "gemm_GQ", 2.0 * elems * h->l, packed ? elems / 4 : elems);
        launch_gq_f32(h->st, packed ? (const void*)h->dG2 : (const void*)h->dG, packed, packed ? h->ld2 : h->ld8, h->gqplan, h->ldg, h->dQ, h->L, h->d_r, h->d_b,
                      h->d_s32, h->dT, scale_out ? h->dTb : nullptr, h->d_cpart);
    }
    HIPCHK(hipGetLastError());
    if (scale_out) CHK(stage_sum_c(h, h->gqplan.waves));
    return GPCA_OK;
}

// One power iteration Y = A^T (A Q) of a STREAMED matrix with every panel read once: K1 on the panel (its rows of T' = r o (A Q), its
// units' shares of c, its column maxima), the panel's rows of T' quantised against the panel's own maxima, K2 on the same panel,
// and the panel's integer sums added into Yacc with the panel's scale.  4 passes over the source per call instead of 6 -- the
// "fused read" SURVEY.md 8(d) counts, which HBM-resident data cannot use (no on-chip room for the N x l accumulators) but a
// panel that sits in HBM between its two kernels can.  Per-panel scales put the 28-bit fixed point on a per-panel grid, so the
// result differs from the resident engine at the 1e-9 level, like a row-sharded run does; gpca_stream_set_fused(h, 0) selects
// the 6-pass form that is bit-identical to the resident engine.
static int stage_power_fused(gpca_handle* h) {
    const double elems = (double)h->M * (double)h->N;
    const int L = h->L, halves = L / 32;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const size_t qhalf = (size_t)h->ldg * 32 * kDigits, chalf = (size_t)h->Mpad, ahalf = (size_t)h->gqplan.waves * 32;
    const size_t td_half = (size_t)h->Mpad * 32 * kDigits, yint_half = (size_t)h->N * 32;
    {
        const double by = packed ? elems / 4 : elems;
        ScopedTimer sweep(h, "gemm_fused", 4.0 * elems * h->l, by * halves);
        CHK(for_each_panel(h, [&](const PanelView& pv) -> int {
            const GqPlan plan1 = gq_plan(pv.rows_pad, h->gq_waves_target);
            const Gtt8Plan plan2 = gtt8_plan(pv.rows_pad, h->ldg, h->gtt_waves_target);
            const float* rr = h->d_r + pv.row0; const float* bb = h->d_b + pv.row0;
            for (int hf = 0; hf < halves; ++hf) {
                const int8_t* Qd = h->dQd + hf * qhalf;
                float* Th = h->dT + (size_t)pv.row0 * L + 32 * hf;
                float* cp = h->d_cpart + hf * chalf + (size_t)pv.row0;
                double* ap = h->d_apart + hf * ahalf;
                if (packed) launch_gq_2bit(h->st, pv.g2, h->ld2, plan1, h->ldg, Qd, h->d_qscale + 32 * hf, rr, bb, h->d_s32 + 32 * hf, Th, cp, ap, 1, h->nd, L);
                else if (h->gq_dma) {
                    const int e = launch_gq_d(h->st, pv.g8, h->ld8, plan1, h->ldg, Qd, h->d_qscale + 32 * hf, rr, bb, h->d_s32 + 32 * hf, Th, cp, ap, 1, L, h->ko);
                    if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gq_d launch failed (hip error " + std::to_string(e) + ")");
                } else launch_gq_x(h->st, pv.g8, h->ld8, plan1, h->ldg, Qd, h->d_qscale + 32 * hf, rr, bb, h->d_s32 + 32 * hf, Th, cp, ap, 1, L, h->ko);
                HIPCHK(hipGetLastError());
                // this panel's rows of T' -> digit planes against this panel's column maxima
                launch_quantize_f32_premax(h->st, Th, pv.rows_pad, pv.rows_pad, ap, plan1.waves, h->d_tscale + 32 * hf, h->d_tinv + 32 * hf,
                                           h->dTd + hf * td_half + (size_t)(pv.row0 >> 5) * kPlaneBytesPerBlock, 0, h->nd, L);
                HIPCHK(hipGetLastError());
            }
            for (int hf = 0; hf < halves; ++hf) {
                CHK(k2_panel(h, pv, h->dTd + hf * td_half, plan2));
                launch_accum_y_scaled(h->st, h->dYpart64, plan2.W, h->ldg, h->N, h->d_tscale + 32 * hf, h->d_yint + hf * yint_half, pv.index == 0);
                HIPCHK(hipGetLastError());
            }
            return GPCA_OK;
        }));
    }
    for (int hf = 0; hf < halves; ++hf) {
        launch_sum_partials_f32(h->st, h->d_cpart + hf * chalf, h->Mpad / 32, 32, h->d_c + 32 * hf, h->d_scratch64);
        HIPCHK(hipGetLastError());
        launch_finish_y_sum(h->st, h->d_yint + hf * yint_half, h->N, h->d_c + 32 * hf, h->dY + 32 * hf, L);
        HIPCHK(hipGetLastError());
    }
    h->apart_valid = false;
    return GPCA_OK;
}

// CholeskyQR2 of dY -> dQ (f32, padded), s = 1^T Q
static int stage_orth(gpca_handle* h) {
    const int L = h->L, l = h->l;
    for (int round = 0; round < 2; ++round) {      // CholeskyQR2, entirely on the stream (no host round trip)
        const int64_t parts = gram_num_parts(h->N);
        launch_gram_f64(h->st, h->dY, h->N, L, h->d_part64);
        HIPCHK(hipGetLastError());
        launch_sum_partials_f64(h->st, h->d_part64, parts, (int64_t)L * L, h->dW, h->d_scratch64);
        HIPCHK(hipGetLastError());
        launch_chol_inv(h->st, h->dW, l, L, h->dZ, h->d_cholflag);
        HIPCHK(hipGetLastError());
        if (round == 0) launch_apply_right_inplace(h->st, h->dY, h->N, L, h->dZ, nullptr, h->ldg);
        else launch_apply_right_tail(h->st, h->dY, h->N, L, h->dZ, h->dQ, h->ldg, h->d_part64, h->d_part64 + tail_num_parts(h->ldg) * L);
        HIPCHK(hipGetLastError());
    }
    // dY now holds the orthonormal basis in f64: s = Q^T 1 and (exact-integer path) the digit scale of Q, then its planes
    const bool i8 = h->precision == GPCA_PREC_I8_EXACT;
    launch_finish_q(h->st, h->d_part64, h->d_part64 + tail_num_parts(h->ldg) * L, tail_num_parts(h->ldg), L, h->d_s64, h->d_s32,
                    i8 ? h->d_qscale : nullptr, i8 ? h->d_qinv : nullptr, h->nd);
    HIPCHK(hipGetLastError());
    if (i8) {
        for (int hf = 0; hf < L / 32; ++hf) {
            launch_quantize_f64_prescaled(h->st, h->dY + 32 * hf, h->N, h->ldg, h->d_qinv + 32 * hf, h->dQd + (size_t)hf * h->ldg * 32 * kDigits,
                                          h->storage == GPCA_STORE_2BIT ? 1 : 0, h->nd, L);
            HIPCHK(hipGetLastError());
        }
    }
    return GPCA_OK;
}

static int ensure_workspace(gpca_handle* h) {
    const int L = h->L;
    const int64_t Npad = h->ldg, M = h->M, N = h->N;
    // streamed mode: the GEMM grids are sized per panel (all panels but the last have panel_rows rows)
    const int64_t gemm_rows = h->sm.on ? h->sm.panel_rows : h->Mpad;
    h->plan = gtt_plan(h->Mpad, Npad, L, h->gtt_waves_target);
    h->gqplan = gq_plan(gemm_rows, h->gq_waves_target);
    CHK(ensure(h, h->dQ, h->cap_Q, (size_t)Npad * L));
    CHK(ensure(h, h->dT, h->cap_T, (size_t)h->Mpad * L));
    if (h->precision == GPCA_PREC_F32_MFMA) CHK(ensure(h, h->dTb, h->cap_Tb, (size_t)h->Mpad * L));
    if (h->Mpad > M) HIPCHK(hipMemsetAsync(h->dT + (size_t)M * L, 0, (size_t)(h->Mpad - M) * L * 4, h->st));
    if (h->precision == GPCA_PREC_F32_MFMA) CHK(ensure(h, h->dYpart, h->cap_Ypart, (size_t)h->plan.W * (size_t)Npad * L));
    // c partials: per wave x L (f32 path, Omega: 64-row groups), or per 32-row unit x 32 per column half (exact path)
    const int64_t cparts = std::max({h->gqplan.waves * (int64_t)L, omega_num_parts(h->Mpad) * (int64_t)L, h->Mpad * (int64_t)(L / 32)});
    CHK(ensure(h, h->d_cpart, h->cap_cpart, (size_t)cparts));
    CHK(ensure(h, h->dY, h->cap_Y, (size_t)N * L));
    const int64_t p64 = std::max({gram_num_parts(N) * (int64_t)L * L, gram_num_parts(M) * (int64_t)L * L, colsum_num_parts(Npad) * (int64_t)L, absmax_num_parts(h->Mpad) * (int64_t)32, 2 * tail_num_parts(Npad) * (int64_t)L});
    CHK(ensure(h, h->d_part64, h->cap_part64, (size_t)p64));
    if (!h->d_c) {
        HIPCHK(hipMalloc((void**)&h->d_c, 64 * 8)); HIPCHK(hipMalloc((void**)&h->d_s64, 64 * 8));
        HIPCHK(hipMalloc((void**)&h->d_s32, 64 * 4)); HIPCHK(hipMalloc((void**)&h->dW, 64 * 64 * 8));
        HIPCHK(hipMalloc((void**)&h->dZ, 2 * 64 * 64 * 8));
        HIPCHK(hipHostMalloc((void**)&h->h_pin, (3 * 64 * 64 + 16) * 8, hipHostMallocDefault)); HIPCHK(hipMalloc((void**)&h->d_sign, 64 * 4));
        HIPCHK(hipMalloc((void**)&h->d_scratch64, kSumScratchElems * 8));
        HIPCHK(hipMalloc((void**)&h->d_cholflag, 4));
    }
    if (h->precision == GPCA_PREC_I8_EXACT) {
        h->plan8 = gtt8_plan(gemm_rows, Npad, h->gtt_waves_target);
        CHK(ensure(h, h->dQd, h->cap_Qd, (size_t)Npad * 32 * kDigits * (size_t)(L / 32)));
        CHK(ensure(h, h->dTd, h->cap_Td, (size_t)h->Mpad * 32 * kDigits * (size_t)(L / 32)));
        CHK(ensure(h, h->dYpart64, h->cap_Ypart64, (size_t)h->plan8.W * (size_t)Npad * 32));
        CHK(ensure(h, h->d_apart, h->cap_apart, (size_t)h->gqplan.waves * 32 * (size_t)(L / 32)));
        if (h->sm.on) CHK(ensure(h, h->d_yint, h->cap_yint, (size_t)N * 32 * (size_t)(L / 32)));
        if (!h->d_qscale) {
            HIPCHK(hipMalloc((void**)&h->d_qscale, 64 * 8)); HIPCHK(hipMalloc((void**)&h->d_qinv, 64 * 8));
            HIPCHK(hipMalloc((void**)&h->d_tscale, 64 * 8)); HIPCHK(hipMalloc((void**)&h->d_tinv, 64 * 8));
            HIPCHK(hipMalloc((void**)&h->d_amax_run, 64 * 8));
        }
    }
    size_t cap2 = h->cap_scores;
    CHK(ensure(h, h->d_scores64, h->cap_scores, (size_t)N * h->k));
    CHK(ensure(h, h->d_scores32, cap2, (size_t)N * h->k));
    CHK(ensure(h, h->d_load32, h->cap_load, (size_t)std::max<int64_t>(h->n_pca, 1) * h->k));
    return GPCA_OK;
}

// argument / state checks of gpca_rsvd + workspace: everything that can fail on one rank only before the first exchange
static int rsvd_preflight(gpca_handle* h, int32_t k, int32_t oversample, int32_t power_iters) {
    if (!have_genotypes(h)) return fail(h, GPCA_ERR_STATE, "gpca_rsvd: no genotypes resident and no panel stream open");
    if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_rsvd: run gpca_snp_stats or gpca_set_standardization first");
    if (k <= 0) return fail(h, GPCA_ERR_BAD_ARG, "Number of components (-k) must be > 0.");  // main.rs:607-609
    if (oversample < 0 || power_iters < 0) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: negative oversample/power_iters");
    if (h->N < 2) return fail(h, GPCA_ERR_BAD_ARG, "PCA requires at least 2 samples.");   // main.rs:614-616
    if (!multi_rank(h) && h->n_pca == 0) return fail(h, GPCA_ERR_BAD_ARG, "PCA requires at least 1 variant (feature), found 0.");  // main.rs:617-619
    const int l = k + oversample;
    if (l > 64) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: k + oversample must be <= 64");
    if (l > h->N || (!multi_rank(h) && l > h->n_pca)) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: k + oversample exceeds min(samples, PCA SNPs)");
    if (h->flags & 1u) return fail(h, GPCA_ERR_MISSING_GENOTYPE,
        "Unexpected missing genotype (-127i8) in a PCA SNP. This should have been filtered by QC.");  // prepare.rs:1909-1911
    if (h->flags & 2u) return fail(h, GPCA_ERR_INVALID_GENOTYPE, "a PCA SNP holds a dosage outside {0,1,2}");
    if (h->sm.on && h->precision != GPCA_PREC_I8_EXACT) return fail(h, GPCA_ERR_STATE, "gpca_rsvd: streamed panels need GPCA_PREC_I8_EXACT");
    HIPCHK(hipSetDevice(h->device));
    h->k = k; h->l = l; h->L = l <= 32 ? 32 : 64;
    h->have_rsvd = false;
    CHK(ensure_workspace(h));
    HIPCHK(hipMemsetAsync(h->d_cholflag, 0, 4, h->st));
    return GPCA_OK;
}

extern "C" int gpca_rsvd(gpca_handle* h, int32_t k, int32_t oversample, int32_t power_iters, uint64_t seed) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    // Ranks of a sharded run leave together: agree on the preflight status before the first exchange ...
    int lrc = agree_status(h, rsvd_preflight(h, k, oversample, power_iters), "gpca_rsvd (before the sketch)");
    if (lrc != GPCA_OK) return lrc;
    const int l = h->l, L = h->L;
    const bool mr = multi_rank(h);
    // ... and from here on a rank-local failure is remembered (lrc) while the rank keeps entering every exchange of the call, so
    // that its peers are not left inside a collective; the second agreement below returns the failure on every rank.
#define LOCAL(x) do { if (lrc == GPCA_OK) lrc = (x); if (lrc != GPCA_OK && !mr) return lrc; } while (0)
#define EXCHANGE(buf, count) do { const int xrc_ = allreduce_f64(h, (buf), (count)); if (xrc_ != GPCA_OK) return xrc_; } while (0)
    auto omega = [&]() -> int {
        // 1. sketch: T' = r o Omega, c = b^T Omega
        ScopedTimer t(h, "omega", 0.0, (double)h->M * L * 4.0);
        if (h->precision == GPCA_PREC_I8_EXACT) {
            HIPCHK(hipMemsetAsync(h->d_apart, 0, 32 * 8, h->st));
            launch_omega(h->st, h->M, h->Mpad, l, L, h->snp_offset, seed, h->d_r, h->d_b, h->dT, h->d_cpart, L == 32 ? h->d_apart : nullptr, 0);
            h->apart_valid = L == 32; h->apart_parts = 1; h->apart_src[0] = h->d_apart; h->apart_src[1] = nullptr;
        } else launch_omega(h->st, h->M, h->Mpad, l, L, h->snp_offset, seed, h->d_r, h->d_b, h->dTb, h->d_cpart, nullptr, 1);
        HIPCHK(hipGetLastError());
        return GPCA_OK;
    };
    LOCAL(omega());
    LOCAL(stage_sum_c(h, omega_num_parts(h->Mpad)));
    LOCAL(stage_AtT_local(h));                       // Y = A^T Omega
    EXCHANGE(h->dY, h->N * (int64_t)L);
    LOCAL(stage_orth(h));
    // 2. power iterations
    const bool fused = h->sm.on && h->sm.fused && (h->storage == GPCA_STORE_2BIT || h->lds_planes);   // (k_gq_i8 has no abs-max epilogue)
    for (int it = 0; it < power_iters; ++it) {
        if (fused) LOCAL(stage_power_fused(h));
        else { LOCAL(stage_AQ(h, 1)); LOCAL(stage_AtT_local(h)); }
        EXCHANGE(h->dY, h->N * (int64_t)L);
        LOCAL(stage_orth(h));
    }
    // 3. projection B = A Q, small eigenproblem of B^T B
    LOCAL(stage_AQ(h, 0));
    auto gram_b = [&]() -> int {
        const int64_t parts = gram_num_parts(h->M);
        launch_gram_f32(h->st, h->dT, h->M, L, h->d_part64);
        HIPCHK(hipGetLastError());
        launch_sum_partials_f64(h->st, h->d_part64, parts, (int64_t)L * L, h->dW, h->d_scratch64);
        HIPCHK(hipGetLastError());
        return GPCA_OK;
    };
    LOCAL(gram_b());
    EXCHANGE(h->dW, (int64_t)L * L);
    // The only host step: the l x l eigenproblem.  Pinned staging + a busy-polled stream keep the round trip short
    // (pageable copies and a sleeping hipStreamSynchronize cost ~220 us here); everything after it is enqueued at once.
    std::vector<double> C((size_t)l * l), V((size_t)l * l), w((size_t)l);
    double* Wfull = h->h_pin;
    double* Zpin = h->h_pin + 64 * 64;                    // [scores Z (L x k) | loadings Z (L x k)]
    int* flagpin = reinterpret_cast<int*>(h->h_pin + 3 * 64 * 64);
    auto fetch_w = [&]() -> int {
        HIPCHK(hipMemcpyAsync(Wfull, h->dW, sizeof(double) * L * L, hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipMemcpyAsync(flagpin, h->d_cholflag, 4, hipMemcpyDeviceToHost, h->st));
        HIPCHK(stream_wait(h));
        if (*flagpin) {
            char buf[160];
            snprintf(buf, sizeof buf, "CholeskyQR: pivot %d of the %d-column sketch is not positive (rank-deficient sketch)", *flagpin - 1, l);
            return fail(h, GPCA_ERR_NOT_CONVERGED, buf);
        }
        return GPCA_OK;
    };
    LOCAL(fetch_w());
    lrc = agree_status(h, lrc, "gpca_rsvd (after the last exchange)");
    if (lrc != GPCA_OK) return lrc;
#undef LOCAL
#undef EXCHANGE
    for (int a2 = 0; a2 < l; ++a2) for (int c = 0; c < l; ++c) C[(size_t)a2 * l + c] = 0.5 * (Wfull[(size_t)a2 * L + c] + Wfull[(size_t)c * L + a2]);
    host_eigh_desc(C, V, w, l);
    h->sv.assign((size_t)l, 0.0); h->eig.assign((size_t)k, 0.0);
    for (int j = 0; j < l; ++j) h->sv[(size_t)j] = w[(size_t)j] > 0 ? std::sqrt(w[(size_t)j]) : 0.0;
    for (int c = 0; c < k; ++c) h->eig[(size_t)c] = w[(size_t)c] / (double)(h->N - 1);
    // 4. scores = Q V_k diag(s) ; sign ; loadings = B V_k diag(sign/s)   (the sign is applied to the loadings' Z on the device)
    const size_t zk = (size_t)L * k;
    for (size_t e = 0; e < 2 * zk; ++e) Zpin[e] = 0.0;
    for (int j = 0; j < l; ++j) for (int c = 0; c < k; ++c) {
        Zpin[(size_t)j * k + c] = V[(size_t)j * l + c] * h->sv[(size_t)c];
        Zpin[zk + (size_t)j * k + c] = h->sv[(size_t)c] > 0 ? V[(size_t)j * l + c] / h->sv[(size_t)c] : 0.0;
    }
    HIPCHK(hipMemcpyAsync(h->dZ, Zpin, sizeof(double) * 2 * zk, hipMemcpyHostToDevice, h->st));
    launch_rightmul_f64(h->st, h->dY, h->N, L, h->dZ, k, h->d_scores64, (float*)nullptr);   // dY = Q in f64
    HIPCHK(hipGetLastError());
    launch_col_sign(h->st, h->d_scores64, h->N, k, h->d_sign);
    HIPCHK(hipGetLastError());
    launch_scale_cols(h->st, h->d_scores64, h->d_scores32, h->N, k, h->d_sign);
    HIPCHK(hipGetLastError());
    launch_scale_cols(h->st, h->dZ + zk, (float*)nullptr, L, k, h->d_sign);
    HIPCHK(hipGetLastError());
    launch_rightmul_gather_f32(h->st, h->dT, h->d_pca_rows, h->n_pca, L, h->dZ + zk, k, h->d_load32);
    HIPCHK(hipGetLastError());
    HIPCHK(stream_wait(h));
    h->have_rsvd = true;
    return GPCA_OK;
}

#define NEED_RSVD(name) \
    if (!h || !out) return GPCA_ERR_BAD_ARG; \
    LOCK(h); \
    if (!h->have_rsvd) return fail(h, GPCA_ERR_STATE, name ": run gpca_rsvd first")

extern "C" int gpca_get_scores(gpca_handle* h, float* out) {
    NEED_RSVD("gpca_get_scores");
    HIPCHK(hipMemcpy(out, h->d_scores32, (size_t)h->N * h->k * 4, hipMemcpyDeviceToHost));
    return GPCA_OK;
}
extern "C" int gpca_get_scores_f64(gpca_handle* h, double* out) {
    NEED_RSVD("gpca_get_scores_f64");
    HIPCHK(hipMemcpy(out, h->d_scores64, (size_t)h->N * h->k * 8, hipMemcpyDeviceToHost));
    return GPCA_OK;
}
extern "C" int gpca_get_eigenvalues(gpca_handle* h, double* out) {
    NEED_RSVD("gpca_get_eigenvalues");
    std::copy(h->eig.begin(), h->eig.end(), out);
    return GPCA_OK;
}
extern "C" int gpca_get_singular_values(gpca_handle* h, double* out) {
    NEED_RSVD("gpca_get_singular_values");
    std::copy(h->sv.begin(), h->sv.end(), out);
    return GPCA_OK;
}
extern "C" int gpca_get_loadings(gpca_handle* h, float* out) {
    NEED_RSVD("gpca_get_loadings");
    if (h->n_pca) HIPCHK(hipMemcpy(out, h->d_load32, (size_t)h->n_pca * h->k * 4, hipMemcpyDeviceToHost));
    return GPCA_OK;
}

// PCA::transform (main.rs:659): scores = A^T U on the resident (or streamed) matrix, U = loadings.
extern "C" int gpca_transform(gpca_handle* h, double* out) {
    NEED_RSVD("gpca_transform");
    HIPCHK(hipSetDevice(h->device));
    const int L = h->L, k = h->k;
    const bool mr = multi_rank(h);
    int lrc = GPCA_OK;
#define LOCAL(x) do { if (lrc == GPCA_OK) lrc = (x); if (lrc != GPCA_OK && !mr) return lrc; } while (0)
    auto prep = [&]() -> int {
        // T' = r o U (zero rows for dropped SNPs), c = b^T U
        HIPCHK(hipMemsetAsync(h->dT, 0, (size_t)h->Mpad * L * 4, h->st));
        launch_expand_loadings(h->st, h->d_load32, h->d_pca_rows, h->n_pca, k, L, h->dT);
        HIPCHK(hipGetLastError());
        if (h->precision == GPCA_PREC_I8_EXACT) launch_scale_rows(h->st, h->dT, h->M, h->Mpad, L, h->d_r, h->d_b, h->dT, h->d_cpart, 0);   // in place, row-major
        else launch_scale_rows(h->st, h->dT, h->M, h->Mpad, L, h->d_r, h->d_b, h->dTb, h->d_cpart);
        HIPCHK(hipGetLastError());
        h->apart_valid = false;
        return stage_sum_c(h, omega_num_parts(h->Mpad));
    };
    LOCAL(prep());
    LOCAL(stage_AtT_local(h));
    { const int xrc = allreduce_f64(h, h->dY, h->N * (int64_t)L); if (xrc != GPCA_OK) return xrc; }
    lrc = agree_status(h, lrc, "gpca_transform");
    if (lrc != GPCA_OK) return lrc;
#undef LOCAL
    std::vector<double> Y((size_t)h->N * L);
    HIPCHK(hipMemcpyAsync(Y.data(), h->dY, (size_t)h->N * L * 8, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    for (int64_t n = 0; n < h->N; ++n) for (int c = 0; c < k; ++c) out[n * k + c] = Y[(size_t)n * L + c];
    h->have_rsvd = true;  // dT (=B) is consumed, but scores/loadings/eigenvalues stay valid
    return GPCA_OK;
}

// ---- d: timings ------------------------------------------------------------------------------------------------
extern "C" int gpca_enable_timings(gpca_handle* h, int32_t on) { if (!h) return GPCA_ERR_BAD_ARG; LOCK(h); h->timing_on = on != 0; return GPCA_OK; }
extern "C" int gpca_reset_timings(gpca_handle* h) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (h->sm.st_fill) HIPCHK(hipStreamSynchronize(h->sm.st_fill));
    HIPCHK(hipStreamSynchronize(h->st));
    for (auto& r : h->recs) { h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b); }
    h->recs.clear(); h->agg.clear();
    return GPCA_OK;
}
extern "C" int gpca_get_timings(gpca_handle* h, gpca_kernel_timing* out, int32_t cap, int32_t* n) {
    if (!h || !n) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (h->sm.st_fill) HIPCHK(hipStreamSynchronize(h->sm.st_fill));
    HIPCHK(hipStreamSynchronize(h->st));
    fold_timings(h);
    *n = (int32_t)h->agg.size();
    if (out) for (int32_t i = 0; i < *n && i < cap; ++i) out[i] = h->agg[(size_t)i];
    return GPCA_OK;
}
