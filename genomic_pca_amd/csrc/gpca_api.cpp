// C ABI of the randomized-PCA engine (include/gpca.h): lifecycle, SNP QC statistics, the pull API, the exchange step of row-sharded
// runs, timings.  Residency / panel sources: gpca_residency.cpp; the randomized-PCA stages: gpca_rsvd.cpp; shared internals:
// gpca_internal.h.
#include "gpca_internal.h"

using namespace gpca;

RcclApi g_rccl;
thread_local std::string g_last_global_err;

// The failing call's text is kept twice: on the handle, and for the calling thread -- several threads may be inside
// gpca_standardize_block on one handle at once, and each must read ITS failure back from gpca_last_error.
// (keyed by the handle's generation id, not its address: a new handle at a recycled address must not read a stale text)
static thread_local uint64_t g_tls_err_gen = 0;
static thread_local std::string g_tls_err;
static std::atomic<uint64_t> g_next_gen{1};
int fail(gpca_handle* h, int code, const std::string& msg) {
    if (h) { h->err = msg; g_tls_err_gen = h->gen; g_tls_err = msg; } else g_last_global_err = msg;
    return code;
}

void fold_timings(gpca_handle* h) {   // resolve pending records into per-name totals and recycle their events
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

extern "C" const char* gpca_last_error(gpca_handle* h) {
    if (!h) return g_last_global_err.c_str();
    // this thread's own last failure on this handle (a thread that never failed here reads the handle's text)
    if (g_tls_err_gen == h->gen) return g_tls_err.c_str();
    return h->err.c_str();
}

extern "C" int gpca_create(const gpca_config* cfg, gpca_handle** out) {
    if (!out) return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, GPCA_ERR_NO_DEVICE, "gpca_create: no HIP device visible (this engine has no CPU fallback)");
    gpca_handle* h = new gpca_handle();
    h->gen = g_next_gen.fetch_add(1, std::memory_order_relaxed);
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: device ordinal out of range"); }
    h->device = dev;
    // A zeroed config (or none) is the fast exact path with automatic residency -- the boundary's default is the path both command
    // lines and the Python mirror default to, not the slowest one (it was: 28 ms against 10 ms per call at configs[1]).
    h->precision = (!cfg || cfg->precision == GPCA_PREC_DEFAULT) ? GPCA_PREC_I8_EXACT : cfg->precision;
    h->storage_cfg = cfg ? cfg->storage : GPCA_STORE_AUTO;
    if (h->storage_cfg != GPCA_STORE_AUTO && h->storage_cfg != GPCA_STORE_INT8 && h->storage_cfg != GPCA_STORE_2BIT) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: unknown storage mode"); }
    if (h->precision != GPCA_PREC_F32_MFMA && h->precision != GPCA_PREC_I8_EXACT) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: unknown precision mode"); }
    h->storage = h->storage_cfg == GPCA_STORE_2BIT ? GPCA_STORE_2BIT : GPCA_STORE_INT8;     // (AUTO: provisional until rows arrive)
    {
        const int dp = cfg ? cfg->digit_planes : 0;
        if (dp != 0 && dp != 3 && dp != 4) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: digit_planes must be 0, 3 or 4"); }
        if (dp == 3 && !(h->precision == GPCA_PREC_I8_EXACT && h->storage_cfg == GPCA_STORE_2BIT)) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: digit_planes = 3 is implemented for GPCA_PREC_I8_EXACT with GPCA_STORE_2BIT"); }
        h->nd_cfg = dp;
        // digit_planes = 0 leaves the choice to the library: four planes on int8 rows (HBM-bound: the planes cost nothing there), THREE
        // on 2-bit rows, whose kernels are matrix-core bound (a quarter less work).  Round 3 ran the whole parity suite and
        // scripts/planes3_parity.py under both settings: three planes sit within 3e-7 (max|dPC|) / 5e-8 (eigenvalues) of the f64
        // checker on every shape -- closer than the f32-MFMA path (7e-7 / 2e-7).  digit_planes = 4 asks for four.
        h->nd = dp == 3 ? 3 : ((dp == 0 && h->precision == GPCA_PREC_I8_EXACT && h->storage == GPCA_STORE_2BIT) ? 3 : 4);
    }
    // The kernel choice is the CALLER's (gpca_config.reserved[0..2], include/gpca.h), never the process environment's:
    // reserved[0] = GPCA_CFG_* flags, reserved[1] / reserved[2] = resident-wave targets of K1 / K2 (0 = the tuned defaults).
    {
        const int32_t flags = cfg ? cfg->reserved[0] : 0, gqw = cfg ? cfg->reserved[1] : 0, gtw = cfg ? cfg->reserved[2] : 0;
        if ((flags & ~GPCA_CFG_ALL) != 0 || gqw < 0 || gtw < 0 || (cfg && cfg->reserved[3] != 0)) { delete h; return fail(nullptr, GPCA_ERR_BAD_ARG, "gpca_create: unknown bits in gpca_config.reserved"); }
        h->simple_kernels = (flags & GPCA_CFG_SIMPLE_KERNELS) != 0;
        h->compact_ok = (flags & GPCA_CFG_NO_COMPACT) == 0;
        h->narrow_ok = (flags & GPCA_CFG_NO_NARROW) == 0;
        h->spin_sync = (flags & GPCA_CFG_NO_SPIN_SYNC) == 0;
        if (gqw) h->gq_waves_target = std::max(4, (int)gqw);
        if (gtw) h->gtt_waves_target = std::max(4, (int)gtw);
    }
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
    if (e == 0) e = init_device_kernels_eig();
    if (e != 0) {
        std::string m = std::string("gpca_create: cannot reserve the LDS the DMA kernels need (hipFuncSetAttribute: ") + hipGetErrorString((hipError_t)e) + ")";
        (void)hipStreamDestroy(h->st); delete h; return fail(nullptr, GPCA_ERR_HIP, m);
    }
    *out = h;
    return GPCA_OK;
}

void drop_child(gpca_handle* h) {
    h->child_valid = false; h->rsvd_on_child = false;
    if (!h->child) return;
    gpca_handle* c = h->child;
    h->child = nullptr;
    c->d_row_ids = nullptr;
    if (h->timing_on || !c->agg.empty() || !c->recs.empty()) {     // the passes that ran on the child stay in this handle's timings
        (void)hipStreamSynchronize(h->st);
        fold_timings(c);
        for (const gpca_kernel_timing& t : c->agg) {
            size_t i = 0;
            for (; i < h->agg.size(); ++i) if (strcmp(h->agg[i].name, t.name) == 0) break;
            if (i == h->agg.size()) h->agg.push_back(t);
            else { h->agg[i].launches += t.launches; h->agg[i].total_ms += t.total_ms; h->agg[i].flops += t.flops; h->agg[i].bytes += t.bytes; }
        }
    }
    (void)gpca_destroy(c);
}

extern "C" int gpca_destroy(gpca_handle* h) {
    if (!h) return GPCA_OK;
    { LOCK(h);
      (void)hipSetDevice(h->device);
      (void)hipStreamSynchronize(h->st);
      drop_child(h);
      stream_close(h);
      for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
      for (auto e : h->ev_pool) (void)hipEventDestroy(e);
      if (h->ev_status) (void)hipEventDestroy(h->ev_status);
      if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
      free_stats(h); free_ws(h); dfree(h->dG); dfree(h->dG2);
      for (auto* ln : h->lanes_free) {       // (LOCK drained the pulls: every lane is back in the pool)
          (void)hipStreamDestroy(ln->st); (void)hipEventDestroy(ln->dep); dfree(ln->d_rows); dfree(ln->d_cols); dfree(ln->d_out); dfree(ln->d_err); delete ln;
      }
      h->lanes_free.clear(); h->lanes_total = 0;
      dfree(h->d_status); if (h->h_status) { (void)hipHostFree(h->h_status); h->h_status = nullptr; }
      if (!h->is_child) (void)hipStreamDestroy(h->st);      // (a compact child runs on its parent's stream)
    }
    { std::lock_guard<std::mutex> lk(h->pull_mu); }        // a pull that has just counted itself out may still be inside its critical section
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

// ---- a1 ---------------------------------------------------------------------------------------------------
int alloc_stats(gpca_handle* h) {
    const size_t M = (size_t)h->Mpad;   // pad rows: r = b = 0
    if (h->d_mu && h->cap_stats_pad >= h->Mpad) return GPCA_OK;
    if (h->d_mu) free_stats(h);         // (arrays of an earlier, smaller matrix on a reused buffer: gpca_copy_rows)
    h->cap_stats_pad = h->Mpad;
    HIPCHK(hipMalloc((void**)&h->d_mu, M * 4)); HIPCHK(hipMalloc((void**)&h->d_sigma, M * 4));
    HIPCHK(hipMalloc((void**)&h->d_r, M * 4)); HIPCHK(hipMalloc((void**)&h->d_b, M * 4));
    HIPCHK(hipMalloc((void**)&h->d_keep, M)); HIPCHK(hipMalloc((void**)&h->d_reason, M));
    HIPCHK(hipMalloc((void**)&h->d_counts, M * 16)); HIPCHK(hipMalloc((void**)&h->d_flags, 16));
    HIPCHK(hipMemsetAsync(h->d_counts, 0, M * 16, h->st)); HIPCHK(hipMemsetAsync(h->d_r, 0, M * 4, h->st)); HIPCHK(hipMemsetAsync(h->d_b, 0, M * 4, h->st));
    HIPCHK(hipMemsetAsync(h->d_mu, 0, M * 4, h->st)); HIPCHK(hipMemsetAsync(h->d_sigma, 0, M * 4, h->st)); HIPCHK(hipMemsetAsync(h->d_keep, 0, M, h->st));
    return GPCA_OK;
}

int refresh_pca_rows(gpca_handle* h) {
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
    h->have_stats = true; h->have_rsvd = false; h->rmax_valid = false; drop_child(h);
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
    h->have_rsvd = false; h->rmax_valid = false; drop_child(h);
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
extern "C" int64_t gpca_num_pca_snps(gpca_handle* h) { if (!h) return 0; LOCK_SHARED(h); return h->have_stats ? h->n_pca : 0; }
extern "C" int64_t gpca_num_qc_samples(gpca_handle* h) { if (!h) return 0; LOCK_SHARED(h); return h->N; }
extern "C" int gpca_get_pca_snp_rows(gpca_handle* h, int64_t* rows) {
    if (!h || !rows) return GPCA_ERR_BAD_ARG;
    LOCK_SHARED(h);
    if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_get_pca_snp_rows: run gpca_snp_stats first");
    std::copy(h->pca_rows.begin(), h->pca_rows.end(), rows);
    return GPCA_OK;
}

// The reference's solver calls this once per strip from many rayon workers (prepare.rs:1838) and serves the calls from 1-16 actor
// threads in parallel (main.rs:279-283).  Here a call holds the handle's lock only while it validates its ids and claims a lane; the
// id lists, the output block and the error word live in the lane's persistent scratch, and the copies and the kernel run on the
// lane's own stream with the lock released -- calls from different threads overlap (H2D of one, kernel of another, D2H of a third).
static int lane_ensure(gpca_handle::PullLane* ln, int64_t ns, int64_t nj) {
    auto grow = [](void** p, size_t& cap, size_t need, size_t elem) -> hipError_t {
        if (cap >= need) return hipSuccess;
        if (*p) { (void)hipFree(*p); *p = nullptr; cap = 0; }
        const hipError_t e = hipMalloc(p, need * elem);
        if (e == hipSuccess) cap = need;
        return e;
    };
    if (grow((void**)&ln->d_rows, ln->cap_rows, (size_t)ns, 8) != hipSuccess) return GPCA_ERR_OOM;
    if (grow((void**)&ln->d_cols, ln->cap_cols, (size_t)nj, 8) != hipSuccess) return GPCA_ERR_OOM;
    if (grow((void**)&ln->d_out, ln->cap_out, (size_t)ns * (size_t)nj, 4) != hipSuccess) return GPCA_ERR_OOM;
    if (!ln->d_err && hipMalloc((void**)&ln->d_err, 8) != hipSuccess) return GPCA_ERR_OOM;
    return GPCA_OK;
}

extern "C" int gpca_standardize_block(gpca_handle* h, const int64_t* snp_ids, int64_t ns, const int64_t* sample_ids,
                                      int64_t nj, float* out) {
    if (!h) return GPCA_ERR_BAD_ARG;
    std::vector<int64_t> rows;
    gpca_handle::PullLane* ln = nullptr;
    {
        LOCK_SHARED(h);
        if (!h->have_stats) return fail(h, GPCA_ERR_STATE, "gpca_standardize_block: run gpca_snp_stats first");
        if (h->sm.on) return fail(h, GPCA_ERR_STATE, "gpca_standardize_block: the pull API needs a resident matrix (this handle streams panels)");
        if (ns < 0 || nj < 0) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: negative block size");
        if (ns == 0 || nj == 0) return GPCA_OK;  // prepare.rs:1848-1850: empty block, nothing to fill
        if (!snp_ids || !sample_ids || !out) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: NULL pointer");
        rows.resize((size_t)ns);
        for (int64_t a = 0; a < ns; ++a) {
            if (snp_ids[a] < 0 || snp_ids[a] >= h->n_pca) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: PcaSnpId out of range");
            rows[(size_t)a] = h->pca_rows[(size_t)snp_ids[a]];
        }
        for (int64_t c = 0; c < nj; ++c)
            if (sample_ids[c] < 0 || sample_ids[c] >= h->N) return fail(h, GPCA_ERR_BAD_ARG, "gpca_standardize_block: QcSampleId out of range");
        // claim a lane (a 17th concurrent caller waits for one to come back)
        std::unique_lock<std::mutex> lk(h->pull_mu);
        if (h->lanes_free.empty() && h->lanes_total < kMaxPullLanes) {
            auto* fresh = new gpca_handle::PullLane();
            if (hipStreamCreateWithFlags(&fresh->st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&fresh->dep, hipEventDisableTiming) != hipSuccess) {
                if (fresh->st) (void)hipStreamDestroy(fresh->st);
                delete fresh;
                return fail(h, GPCA_ERR_HIP, "gpca_standardize_block: stream creation failed");
            }
            h->lanes_free.push_back(fresh); ++h->lanes_total;
        }
        h->pull_cv.wait(lk, [&] { return !h->lanes_free.empty(); });
        ln = h->lanes_free.back(); h->lanes_free.pop_back();
        // statistics or an upload may still be running on the engine's stream (gpca_snp_stats without a fetch returns at once)
        if (hipEventRecord(ln->dep, h->st) != hipSuccess || hipStreamWaitEvent(ln->st, ln->dep, 0) != hipSuccess) {
            h->lanes_free.push_back(ln);
            return fail(h, GPCA_ERR_HIP, "gpca_standardize_block: could not order the pull behind the engine's stream");
        }
        h->pulls_in_flight.fetch_add(1, std::memory_order_acq_rel);
    }
    // ---- the handle's lock is released: the matrix and its statistics stay put until this pull has left (drain_pulls) ----
    int rc = GPCA_OK;
    std::string msg;
    unsigned long long err_idx = ~0ull;
    auto run = [&]() -> hipError_t {
        hipError_t e;
        if ((e = hipMemcpyAsync(ln->d_rows, rows.data(), (size_t)ns * 8, hipMemcpyHostToDevice, ln->st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(ln->d_cols, sample_ids, (size_t)nj * 8, hipMemcpyHostToDevice, ln->st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(ln->d_err, &err_idx, 8, hipMemcpyHostToDevice, ln->st)) != hipSuccess) return e;
        if (h->storage == GPCA_STORE_2BIT) launch_standardize_block_2bit(ln->st, h->dG2, h->ld2, h->d_mu, h->d_sigma, ln->d_rows, ns, ln->d_cols, nj, ln->d_out, ln->d_err);
        else launch_standardize_block(ln->st, h->dG, h->ld8, h->d_mu, h->d_sigma, ln->d_rows, ns, ln->d_cols, nj, ln->d_out, ln->d_err);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(&err_idx, ln->d_err, 8, hipMemcpyDeviceToHost, ln->st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(out, ln->d_out, (size_t)ns * (size_t)nj * 4, hipMemcpyDeviceToHost, ln->st)) != hipSuccess) return e;
        return hipStreamSynchronize(ln->st);
    };
    rc = lane_ensure(ln, ns, nj);
    if (rc != GPCA_OK) msg = "gpca_standardize_block: out of device memory for the block";
    else {
        const hipError_t e = run();
        if (e != hipSuccess) { rc = GPCA_ERR_HIP; msg = std::string("gpca_standardize_block: ") + hipGetErrorString(e); (void)hipStreamSynchronize(ln->st); }
        else if (err_idx != ~0ull) {
            const int64_t a = (int64_t)(err_idx / (unsigned long long)nj), c = (int64_t)(err_idx % (unsigned long long)nj);
            char buf[400];  // wording of prepare.rs:1910-1911
            snprintf(buf, sizeof buf,
                     "Unexpected missing genotype (-127i8) in SnpBlockData for PCA SNP ID %lld (original BIM index %lld), "
                     "requested sample index %lld. This should have been filtered by QC.",
                     (long long)snp_ids[a], (long long)rows[(size_t)a], (long long)sample_ids[c]);
            rc = GPCA_ERR_MISSING_GENOTYPE; msg = buf;
        }
    }
    {   // the lane goes back and the pull leaves BEFORE the handle's lock is asked for again (a caller that waits in drain_pulls holds it)
        std::lock_guard<std::mutex> lk(h->pull_mu);
        h->lanes_free.push_back(ln);
        h->pulls_in_flight.fetch_sub(1, std::memory_order_acq_rel);
        h->pull_cv.notify_all();     // (under pull_mu: a gpca_destroy woken in drain_pulls cannot delete the handle before this returns)
    }
    if (rc != GPCA_OK) { LOCK_SHARED(h); return fail(h, rc, msg); }
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
    if (h->snp_offset != snp_offset) drop_child(h);      // (a compact child draws Omega by global row index: rebuilt with the new offset)
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
    if (h->snp_offset != snp_offset) drop_child(h);
    h->hook = fn; h->hook_user = user; h->world = world; h->rank = rank; h->snp_offset = snp_offset;
    return GPCA_OK;
}


// in-place sum of a device f64 buffer across the ranks that share the sharded matrix
int allreduce_f64(gpca_handle* h, double* dbuf, int64_t count) {
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
hipError_t stream_wait(gpca_handle* h) {
    if (!h->spin_sync) return hipStreamSynchronize(h->st);
    hipError_t e;
    while ((e = hipStreamQuery(h->st)) == hipErrorNotReady) {}
    return e;
}

// Row-sharded runs: every rank contributes its local status (0 or a negative gpca_status) to a 16-slot histogram that is
// summed through the same transport as the sketch; all ranks then return the most severe (smallest) status any rank saw.
// A rank whose own shard is clean therefore leaves the call with the failing rank's code instead of waiting in a collective
// that the failing rank never enters.  Single-rank handles return local_rc untouched (no device work).
void status_histogram(double* slots16, int local_rc) {
    for (int i = 0; i < 16; ++i) slots16[i] = 0.0;
    slots16[local_rc == GPCA_OK ? 0 : std::min(15, -local_rc)] = 1.0;
}
// what the ranks agreed on, from the summed histogram v[16]; `own` = this rank's error text at the time of its contribution
int status_verdict(gpca_handle* h, const double* v, int local_rc, const std::string& own, const char* where) {
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
// The agreement in two halves, so that the host does not stand still for it: _begin enqueues contribution, exchange and the copy back
// and records an event; the caller goes on enqueueing rank-local work and asks for the verdict (_end) only before it would enter the
// next exchange -- by then the event is long past and the stream has work queued behind it.
int agree_status_begin(gpca_handle* h, int local_rc) {
    if (!multi_rank(h)) return GPCA_OK;
    if (!h->d_status || !h->h_status) return fail(h, GPCA_ERR_STATE, "agree_status: no status buffer (gpca_comm_init / gpca_set_allreduce_hook allocate it)");
    h->status_own = h->err;
    status_histogram(h->h_status, local_rc);
    // pinned staging both ways: H2D, exchange and D2H are stream-ordered
    HIPCHK(hipMemcpyAsync(h->d_status, h->h_status, 16 * sizeof(double), hipMemcpyHostToDevice, h->st));
    CHK(allreduce_f64(h, h->d_status, 16));
    HIPCHK(hipMemcpyAsync(h->h_status + 16, h->d_status, 16 * sizeof(double), hipMemcpyDeviceToHost, h->st));
    if (!h->ev_status) HIPCHK(hipEventCreateWithFlags(&h->ev_status, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->ev_status, h->st));
    return GPCA_OK;
}
int agree_status_end(gpca_handle* h, int local_rc, const char* where) {
    if (!multi_rank(h)) return local_rc;
    hipError_t e;
    if (h->spin_sync) { while ((e = hipEventQuery(h->ev_status)) == hipErrorNotReady) {} }
    else e = hipEventSynchronize(h->ev_status);
    HIPCHK(e);
    return status_verdict(h, h->h_status + 16, local_rc, h->status_own, where);
}
int agree_status(gpca_handle* h, int local_rc, const char* where) {
    if (!multi_rank(h)) return local_rc;
    CHK(agree_status_begin(h, local_rc));
    return agree_status_end(h, local_rc, where);
}

extern "C" int gpca_comm_count_ranks(gpca_handle* h, int32_t* ranks) {
    if (!h || !ranks) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    *ranks = 1;
    if (!multi_rank(h)) return GPCA_OK;
    if (!h->d_status || !h->h_status) return fail(h, GPCA_ERR_STATE, "gpca_comm_count_ranks: no status buffer");
    for (int i = 0; i < 16; ++i) h->h_status[i] = i == 0 ? 1.0 : 0.0;
    HIPCHK(hipMemcpyAsync(h->d_status, h->h_status, 16 * sizeof(double), hipMemcpyHostToDevice, h->st));
    CHK(allreduce_f64(h, h->d_status, 16));
    HIPCHK(hipMemcpyAsync(h->h_status + 16, h->d_status, 16 * sizeof(double), hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    *ranks = (int32_t)(h->h_status[16] + 0.5);
    return GPCA_OK;
}

// ---- d: timings ------------------------------------------------------------------------------------------------
extern "C" int gpca_enable_timings(gpca_handle* h, int32_t on) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    h->timing_on = on != 0;
    h->timing_every = on > 1 ? on : 1; h->timing_calls = 0;
    if (h->child) h->child->timing_on = h->timing_on;
    return GPCA_OK;
}
extern "C" int gpca_reset_timings(gpca_handle* h) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (h->sm.st_fill) HIPCHK(hipStreamSynchronize(h->sm.st_fill));
    HIPCHK(hipStreamSynchronize(h->st));
    for (auto& r : h->recs) { h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b); }
    h->recs.clear(); h->agg.clear();
    if (h->child) CHK(gpca_reset_timings(h->child));
    return GPCA_OK;
}
extern "C" int gpca_get_timings(gpca_handle* h, gpca_kernel_timing* out, int32_t cap, int32_t* n) {
    if (!h || !n) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (h->sm.st_fill) HIPCHK(hipStreamSynchronize(h->sm.st_fill));
    HIPCHK(hipStreamSynchronize(h->st));
    fold_timings(h);
    std::vector<gpca_kernel_timing> all = h->agg;
    if (h->child) {      // the passes that ran on the compact child count as this handle's
        fold_timings(h->child);
        for (const gpca_kernel_timing& t : h->child->agg) {
            size_t i = 0;
            for (; i < all.size(); ++i) if (strcmp(all[i].name, t.name) == 0) break;
            if (i == all.size()) all.push_back(t);
            else { all[i].launches += t.launches; all[i].total_ms += t.total_ms; all[i].flops += t.flops; all[i].bytes += t.bytes; }
        }
    }
    *n = (int32_t)all.size();
    if (out) for (int32_t i = 0; i < *n && i < cap; ++i) out[i] = all[(size_t)i];
    return GPCA_OK;
}
