// Genotype residency: uploads, panel sources, the out-of-core panel ring (include/gpca.h sections "genotype residency" and "g").
#include "gpca_internal.h"

using namespace gpca;

void free_stats(gpca_handle* h) {
    dfree(h->d_mu); dfree(h->d_sigma); dfree(h->d_r); dfree(h->d_b); dfree(h->d_keep); dfree(h->d_reason);
    dfree(h->d_counts); dfree(h->d_flags); dfree(h->d_pca_rows);
    h->have_stats = false; h->n_pca = 0; h->pca_rows.clear(); h->cap_stats_pad = 0;
}
void free_eigensnp(gpca_handle* h) {
    dfree(h->d_smask); h->n_smask = 0; dfree(h->d_cw); dfree(h->d_cfeat0); dfree(h->d_cblk_row0); dfree(h->d_cblk_row1); dfree(h->d_cblk_feat0); dfree(h->d_cblk_c);
    dfree(h->dP); dfree(h->d_lqr); dfree(h->d_ones); dfree(h->d_zeros);
    h->cap_P = h->cap_lqr = h->cap_ones = h->cap_zeros = 0; h->c_cmax = 0; h->c_R = 0; h->c_B = 0; h->loadings_valid = true;
}
void free_ws(gpca_handle* h) {
    free_eigensnp(h);
    dfree(h->dQ); dfree(h->dT); dfree(h->dTb); dfree(h->dYpart); dfree(h->d_cpart); dfree(h->d_s32); dfree(h->dY); dfree(h->d_c);
    dfree(h->d_part64); dfree(h->dW); dfree(h->dZ); dfree(h->d_s64); dfree(h->d_scores64); dfree(h->d_scores32);
    dfree(h->d_load32); dfree(h->d_sign); dfree(h->d_eigres); dfree(h->d_cand_val); dfree(h->d_cand_idx); dfree(h->d_scratch64); dfree(h->d_tr64); h->cap_tr64 = 0;
    dfree(h->dQd); dfree(h->dTd); dfree(h->dYpart64); dfree(h->d_apart); h->cap_apart = 0; dfree(h->d_cholflag); if (h->h_pin) { (void)hipHostFree(h->h_pin); h->h_pin = nullptr; } dfree(h->d_qscale); dfree(h->d_qinv); dfree(h->d_tscale); dfree(h->d_tinv);
    dfree(h->d_amax_run); dfree(h->d_rmax); h->rmax_valid = false; dfree(h->d_yint); h->cap_yint = 0;
    h->cap_Qd = h->cap_Td = h->cap_Ypart64 = 0;
    h->cap_Q = h->cap_T = h->cap_Tb = h->cap_Ypart = h->cap_cpart = h->cap_Y = h->cap_part64 = h->cap_scores = h->cap_load = 0;
    h->have_rsvd = false;
}
static inline bool host_kind(int kind) { return kind == GPCA_PANEL_HOST_I8 || kind == GPCA_PANEL_HOST_BED || kind == GPCA_PANEL_MAPPED_I8 || kind == GPCA_PANEL_MAPPED_BED; }
static inline bool bed_kind(int kind) { return kind == GPCA_PANEL_HOST_BED || kind == GPCA_PANEL_MAPPED_BED; }
static inline bool mapped_kind(int kind) { return kind == GPCA_PANEL_MAPPED_I8 || kind == GPCA_PANEL_MAPPED_BED; }
static inline double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// rows [row0, row0 + rows) of a MAPPED_* source -> staging, by f.copy_threads threads (one memcpy stream runs at 10-15 GB/s on
// this class of host; the link wants > 50)
static void copy_mapped_rows(const Filler& f, int64_t row0, int64_t rows, void* dst) {
    const int64_t w = f.stage_ld;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(f.copy_threads, rows * w / (4 << 20)));   // >= 4 MiB per thread
    auto part = [&](int t) {
        const int64_t a = rows * t / T, b = rows * (t + 1) / T;
        const uint8_t* src = f.map_base + (size_t)(row0 + a) * (size_t)f.map_ld;
        uint8_t* d = (uint8_t*)dst + (size_t)a * (size_t)w;
        if (f.map_ld == w) memcpy(d, src, (size_t)(b - a) * (size_t)w);
        else for (int64_t r = a; r < b; ++r, src += f.map_ld, d += w) memcpy(d, src, (size_t)w);
    };
    if (T == 1) { part(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(part, t);
    part(0);
    for (auto& x : th) x.join();
}

// The worker of a host source: stages posted panels in order, up to n_stage - 1 ahead of the pass thread.
static void filler_worker(Filler* f) {
    (void)hipSetDevice(f->device);
    std::unique_lock<std::mutex> lk(f->m);
    for (;;) {
        // next job when its buffer is not holding a staged panel the pass thread has yet to take
        f->cv.wait(lk, [&] { return f->quit || (!f->jobs.empty() && f->st_state[(size_t)(f->produced % (uint64_t)f->n_stage)] != Filler::kReady); });
        if (f->quit) return;
        const size_t b = (size_t)(f->produced % (uint64_t)f->n_stage);
        const std::pair<int64_t, int64_t> job = f->jobs.front();
        f->jobs.pop_front();
        f->busy = true;
        const bool in_flight = f->st_state[b] == Filler::kInFlight;
        lk.unlock();
        if (in_flight) (void)hipEventSynchronize(f->ev_stage[b]);      // its last H2D copy has left the buffer
        const auto t0 = std::chrono::steady_clock::now();
        int rc = 0;
        if (f->mapped) copy_mapped_rows(*f, job.first, job.second, f->h_stage[b]);
        else rc = f->src.fill(f->src.user, job.first, job.second, f->h_stage[b], f->stage_ld);
        const double ms = ms_since(t0);
        lk.lock();
        f->fill_host_ms += ms;
        f->st_state[b] = Filler::kReady; f->st_rc[b] = rc; f->st_row0[b] = job.first;
        f->produced++; f->busy = false;
        if (rc != 0) f->jobs.clear();          // no further panel is asked after a failure
        f->cv.notify_all();
    }
}

void filler_post(Filler& f, int64_t row0, int64_t rows) {
    if (!f.worker.joinable()) return;          // device generator or zero-staging: nothing to stage
    std::lock_guard<std::mutex> lk(f.m);
    f.jobs.emplace_back(row0, rows); f.posted++;
    f.cv.notify_all();
}

void filler_cancel(Filler& f) {
    if (!f.worker.joinable()) return;
    std::unique_lock<std::mutex> lk(f.m);
    f.jobs.clear();
    f.cv.wait(lk, [&] { return !f.busy; });
    for (size_t b = 0; b < f.st_state.size(); ++b) {
        if (f.st_state[b] == Filler::kInFlight) (void)hipEventSynchronize(f.ev_stage[b]);
        f.st_state[b] = Filler::kFree;
    }
    f.posted = f.produced = f.consumed = 0;
}

static void filler_close(Filler& f) {
    if (f.worker.joinable()) {
        { std::lock_guard<std::mutex> lk(f.m); f.quit = true; f.jobs.clear(); }
        f.cv.notify_all();
        f.worker.join();
    }
    f.quit = false; f.busy = false; f.jobs.clear(); f.posted = f.produced = f.consumed = 0; f.held.clear();
    dfree(f.d_thresh); dfree(f.d_scratch8); dfree(f.d_raw); dfree(f.d_flags);
    for (size_t i = 0; i < f.h_stage.size(); ++i) {
        if (f.ev_stage[i]) { (void)hipEventSynchronize(f.ev_stage[i]); (void)hipEventDestroy(f.ev_stage[i]); }
        if (f.h_stage[i]) (void)hipHostFree(f.h_stage[i]);
    }
    f.h_stage.clear(); f.ev_stage.clear(); f.st_state.clear(); f.st_rc.clear(); f.st_row0.clear(); f.n_stage = 0;
    if (f.registered) { (void)hipHostUnregister((void*)f.map_base); f.registered = false; }
    f.host = f.mapped = false; f.map_base = nullptr;
    f.open = false;
}
void stream_close(gpca_handle* h) {
    StreamState& sm = h->sm;
    if (sm.st_fill) (void)hipStreamSynchronize(sm.st_fill);
    if (h->st) (void)hipStreamSynchronize(h->st);
    filler_close(sm.fl);
    for (void* p : sm.slot) if (p) (void)hipFree(p);
    for (auto e : sm.ev_filled) (void)hipEventDestroy(e);
    for (auto e : sm.ev_free) (void)hipEventDestroy(e);
    sm.slot.clear(); sm.ev_filled.clear(); sm.ev_free.clear(); sm.free_pending.clear();
    for (void* p : sm.cache) if (p) (void)hipFree(p);
    for (auto e : sm.ev_cache) (void)hipEventDestroy(e);
    sm.cache.clear(); sm.ev_cache.clear(); sm.cache_filled.clear();
    if (sm.st_fill) { (void)hipStreamDestroy(sm.st_fill); sm.st_fill = nullptr; }
    sm.on = false; sm.seq = 0; sm.n_panels = 0;
}

// ---- genotype residency -------------------------------------------------------------------------------

// dimensions + (resident = true) the device matrix; streamed mode only records the dimensions
static int alloc_genotypes(gpca_handle* h, int64_t M, int64_t N, bool resident = true) {
    if (M <= 0 || N <= 0) return fail(h, GPCA_ERR_BAD_ARG, "genotype matrix must have M > 0 SNPs and N > 0 samples");
    HIPCHK(hipSetDevice(h->device));
    stream_close(h);
    drop_child(h);
    free_stats(h); free_ws(h); dfree(h->dG); dfree(h->dG2);
    h->M = M; h->N = N; h->Mpad = round_up(M, kGQRowsPerWave); h->pack_flags = 0;
    h->cap_rows_pad = resident ? h->Mpad : 0;
    if (h->storage_cfg == GPCA_STORE_AUTO) {
        // the residency follows the rows (include/gpca.h, GPCA_STORE_AUTO): 2-bit codes from 1 024 samples on, int8 below
        h->storage = h->auto_pin ? h->auto_pin : (N >= kSamplePad2bit ? GPCA_STORE_2BIT : GPCA_STORE_INT8);
        if (h->nd_cfg == 0) h->nd = (h->precision == GPCA_PREC_I8_EXACT && h->storage == GPCA_STORE_2BIT) ? 3 : 4;
    }
    if (h->storage == GPCA_STORE_2BIT) {
        h->ldg = round_up(N, kSamplePad2bit); h->ld2 = h->ldg / 4; h->ld8 = h->ldg;
        // row pitch an odd multiple of 256 B, like the int8 rows below: 1.9 % on the packed K1 (in-process A/B, both engine orders:
        // 1.145 / 1.142 ms padded vs 1.167 / 1.164), nothing on K2; (measured with an environment switch that is gone)
        if (!((h->ld2 / 256) & 1)) h->ld2 += 256;
        if (!resident) return GPCA_OK;
        HIPCHK(malloc_genotypes(h, (void**)&h->dG2, (size_t)h->Mpad * (size_t)h->ld2));
        if (h->Mpad > M) HIPCHK(hipMemsetAsync(h->dG2 + (size_t)M * (size_t)h->ld2, 0, (size_t)(h->Mpad - M) * (size_t)h->ld2, h->st));
        return GPCA_OK;
    }
    h->ldg = round_up(N, kSamplePad); h->ld2 = 0;
    // Row pitch vs HBM channel interleave: rows an EVEN multiple of 256 B apart (10 240 B for 10 000 samples) stream 2-5 % slower
    // than rows an odd multiple apart (profiles/r1_kbench_summary.md section 6): 8 rows of one DMA piece then spread over fewer
    // channels.  The pitch gets one extra 256-byte block in that case; the kernels never read past ldg.
    h->ld8 = ((h->ldg / 256) & 1) ? h->ldg : h->ldg + 256;
    if (!resident) return GPCA_OK;
    HIPCHK(malloc_genotypes(h, (void**)&h->dG, (size_t)h->Mpad * (size_t)h->ld8));
    if (h->Mpad > M) HIPCHK(hipMemsetAsync(h->dG + (size_t)M * (size_t)h->ld8, 0, (size_t)(h->Mpad - M) * (size_t)h->ld8, h->st));
    return GPCA_OK;
}

// rows per chunk of the bounded staging buffers (<= 256 MiB of int8 rows)
static int64_t pack_chunk_rows(gpca_handle* h) {
    int64_t r = ((int64_t)256 << 20) / h->ldg;
    if (r < 1) r = 1;
    return r < h->M ? r : h->M;
}
int finish_pack_flags(gpca_handle* h, unsigned* d_flags, hipStream_t st) {
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
        if (rc == GPCA_OK && h->storage_cfg == GPCA_STORE_AUTO && (h->pack_flags & 2u)) {
            // AUTO chose 2-bit codes and the rows hold a value outside {0, 1, 2, -127}: codes could only store it as "missing", int8
            // rows keep it (the statistics then see what prepare.rs:1267-1279 sees).  Once more, as int8.
            h->auto_pin = GPCA_STORE_INT8;
            rc = gpca_upload_genotypes_i8(h, src, M, N, ld);
            h->auto_pin = 0;
        }
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
        case GPCA_PANEL_MAPPED_I8: case GPCA_PANEL_MAPPED_BED:
            if (!src->user || src->host_ld < 0) return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": mapped source needs user = address of row 0 and host_ld >= 0");
            return GPCA_OK;
        case GPCA_PANEL_SYNTH: case GPCA_PANEL_SYNTH16:
            if (!src->thresh || src->n_pop <= 0 || src->snp_offset < 0) return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": generator source needs thresh, n_pop > 0, snp_offset >= 0");
            return GPCA_OK;
        default: return fail(h, GPCA_ERR_BAD_ARG, std::string(who) + ": unknown panel kind");
    }
}

static int env_int_r(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// staging of one source for chunks of up to chunk_rows rows of the handle's current M x N matrix
static int filler_open(gpca_handle* h, Filler& f, const gpca_panel_source& src, int64_t chunk_rows, hipStream_t st) {
    filler_close(f);
    f.src = src; f.chunk_rows = chunk_rows;
    f.open = true;
    f.device = h->device;
    f.fills = 0; f.fill_host_ms = f.fill_wait_ms = f.register_ms = 0.0;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    if (src.kind == GPCA_PANEL_SYNTH || src.kind == GPCA_PANEL_SYNTH16) {
        const size_t tb = (size_t)h->M * (size_t)src.n_pop * sizeof(uint32_t);
        HIPCHK(hipMalloc((void**)&f.d_thresh, tb));
        HIPCHK(hipMemcpyAsync(f.d_thresh, src.thresh, tb, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));   // the caller's table may be freed after open
    }
    const bool i8_rows = src.kind == GPCA_PANEL_SYNTH || src.kind == GPCA_PANEL_HOST_I8 || src.kind == GPCA_PANEL_MAPPED_I8;
    if (packed && i8_rows) {
        HIPCHK(hipMalloc((void**)&f.d_scratch8, (size_t)chunk_rows * (size_t)h->ldg));
        HIPCHK(hipMemsetAsync(f.d_scratch8, 0, (size_t)chunk_rows * (size_t)h->ldg, st));   // pad columns stay 0
        HIPCHK(hipMalloc((void**)&f.d_flags, 16));
        HIPCHK(hipMemsetAsync(f.d_flags, 0, 16, st));
    }
    if (!host_kind(src.kind)) return GPCA_OK;
    f.host = true;
    f.stage_ld = bed_kind(src.kind) ? (h->N + 3) / 4 : h->N;
    if (bed_kind(src.kind)) HIPCHK(hipMalloc((void**)&f.d_raw, (size_t)chunk_rows * (size_t)f.stage_ld));
    if (mapped_kind(src.kind)) {
        f.mapped = true;
        f.map_base = (const uint8_t*)src.user;
        f.map_ld = src.host_ld > 0 ? src.host_ld : f.stage_ld;
        if (f.map_ld < f.stage_ld) return fail(h, GPCA_ERR_BAD_ARG, "mapped panel source: host_ld is smaller than a row");
        f.copy_threads = std::max(1, std::min({env_int_r("GPCA_COPY_THREADS", 8), 64, (int)std::max(1u, std::thread::hardware_concurrency())}));
        if ((src.flags & GPCA_SOURCE_REGISTER) && env_int_r("GPCA_SOURCE_REGISTER", 1) != 0) {
            // zero staging: lock the caller's pages once, DMA every panel straight out of them.  A mapping that cannot be locked
            // (a file larger than RAM, RLIMIT_MEMLOCK, a read-only mapping the driver refuses) falls back to the staging ring.
            const size_t bytes = (size_t)(h->M - 1) * (size_t)f.map_ld + (size_t)f.stage_ld;
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipHostRegister((void*)f.map_base, bytes, hipHostRegisterDefault);
            if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister((void*)f.map_base, bytes, hipHostRegisterReadOnly); }
            if (e == hipSuccess) { f.registered = true; f.reg_bytes = bytes; f.register_ms = ms_since(t0); return GPCA_OK; }
            (void)hipGetLastError();
        }
    }
    f.n_stage = std::max(2, std::min(8, env_int_r("GPCA_STAGE_BUFFERS", 3)));
    for (int i = 0; i < f.n_stage; ++i) {
        void* p = nullptr; hipEvent_t e = nullptr;
        HIPCHK(hipHostMalloc(&p, (size_t)chunk_rows * (size_t)f.stage_ld, hipHostMallocDefault));
        f.h_stage.push_back(p); f.ev_stage.push_back(nullptr);
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        f.ev_stage.back() = e;
    }
    f.st_state.assign((size_t)f.n_stage, Filler::kFree); f.st_rc.assign((size_t)f.n_stage, 0); f.st_row0.assign((size_t)f.n_stage, -1);
    f.posted = f.produced = f.consumed = 0; f.quit = false; f.busy = false;
    f.worker = std::thread(filler_worker, &f);
    return GPCA_OK;
}

// host rows (pinned staging, or the caller's page-locked mapping) of pitch src_ld -> dst on the device, enqueued on st
static int host_rows_to_device(gpca_handle* h, Filler& f, const void* src, int64_t src_ld, int64_t rows, void* dst, hipStream_t st) {
    const bool packed = h->storage == GPCA_STORE_2BIT;
    if (bed_kind(f.src.kind)) {
        if (src_ld == f.stage_ld) HIPCHK(hipMemcpyAsync(f.d_raw, src, (size_t)rows * (size_t)f.stage_ld, hipMemcpyHostToDevice, st));
        else HIPCHK(hipMemcpy2DAsync(f.d_raw, (size_t)f.stage_ld, src, (size_t)src_ld, (size_t)f.stage_ld, (size_t)rows, hipMemcpyHostToDevice, st));
        if (packed) launch_bed_to_codes(st, f.d_raw, f.stage_ld, (uint8_t*)dst, rows, h->N, h->ld2);
        else launch_bed_decode(st, f.d_raw, f.stage_ld, (int8_t*)dst, rows, h->N, h->ld8);
        HIPCHK(hipGetLastError());
    } else if (packed) {
        HIPCHK(hipMemcpy2DAsync(f.d_scratch8, (size_t)h->ldg, src, (size_t)src_ld, (size_t)h->N, (size_t)rows, hipMemcpyHostToDevice, st));
        launch_pack_i8(st, f.d_scratch8, h->ldg, (uint8_t*)dst, rows, h->N, h->ld2, f.d_flags);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)h->ld8, src, (size_t)src_ld, (size_t)h->N, (size_t)rows, hipMemcpyHostToDevice, st));
    }
    return GPCA_OK;
}

// rows [row0, row0 + rows) of the matrix -> dst (int8 rows of pitch ldg, or 2-bit rows of pitch ld2), enqueued on st
int filler_fill(gpca_handle* h, Filler& f, int64_t row0, int64_t rows, void* dst, hipStream_t st) {
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const gpca_panel_source& s = f.src;
    f.fills++;
    switch (s.kind) {
        case GPCA_PANEL_SYNTH:
            if (packed) {
                launch_synth(st, f.d_scratch8, rows, h->N, h->ldg, s.snp_offset + row0, s.seed, f.d_thresh + (size_t)row0 * s.n_pop, s.n_pop);
                launch_pack_i8(st, f.d_scratch8, h->ldg, (uint8_t*)dst, rows, h->N, h->ld2, f.d_flags);
            } else launch_synth(st, (int8_t*)dst, rows, h->N, h->ld8, s.snp_offset + row0, s.seed, f.d_thresh + (size_t)row0 * s.n_pop, s.n_pop);
            HIPCHK(hipGetLastError());
            return GPCA_OK;
        case GPCA_PANEL_SYNTH16:
            if (s.flags & GPCA_SOURCE_BENCH_HOLD) {      // measurement only: a buffer that already holds a generated panel of this height keeps it
                auto it = f.held.find(dst);
                if (it != f.held.end() && it->second == rows) return GPCA_OK;
                f.held[dst] = rows;
            }
            // In launches of 8 192 rows: one grid over a whole panel keeps the dispatcher busy for ~24 ms at 500k samples, and a small kernel
            // of the pass that becomes ready meanwhile (a panel's column scales between its K1 and its K2) waited for it -- 3.9 ms on
            // average, 1.5 s of a configs[4] step (profiles/r5_kbench_summary.md section 6).
            for (int64_t r0 = 0; r0 < rows; r0 += 8192) {
                const int64_t nr = std::min<int64_t>(8192, rows - r0);
                const size_t pitch = (size_t)(packed ? h->ld2 : h->ld8);
                launch_synth16(st, (char*)dst + (size_t)r0 * pitch, packed ? 1 : 0, nr, h->N, packed ? h->ld2 : h->ld8, s.snp_offset + row0 + r0, s.seed,
                               f.d_thresh + (size_t)(row0 + r0) * s.n_pop, s.n_pop);
            }
            HIPCHK(hipGetLastError());
            return GPCA_OK;
        case GPCA_PANEL_HOST_I8: case GPCA_PANEL_HOST_BED: case GPCA_PANEL_MAPPED_I8: case GPCA_PANEL_MAPPED_BED: {
            if (rows > f.chunk_rows) return fail(h, GPCA_ERR_BAD_ARG, "panel source: more rows asked than the staging holds");
            if (f.registered)      // zero staging: straight out of the caller's page-locked mapping
                return host_rows_to_device(h, f, f.map_base + (size_t)row0 * (size_t)f.map_ld, f.map_ld, rows, dst, st);
            size_t b;
            int rc;
            {
                std::unique_lock<std::mutex> lk(f.m);
                if (f.consumed == f.posted) { f.jobs.emplace_back(row0, rows); f.posted++; f.cv.notify_all(); }   // (not announced by filler_post)
                b = (size_t)(f.consumed % (uint64_t)f.n_stage);
                const auto t0 = std::chrono::steady_clock::now();
                const bool waited = !(f.st_state[b] == Filler::kReady && f.produced > f.consumed);
                f.cv.wait(lk, [&] { return f.st_state[b] == Filler::kReady && f.produced > f.consumed; });
                if (waited) f.fill_wait_ms += ms_since(t0);
                rc = f.st_rc[b];
                if (rc == 0 && f.st_row0[b] != row0) rc = -1;      // (the walk asked for rows other than the ones it posted)
            }
            if (rc != 0) {
                filler_cancel(f);
                char buf[200];
                if (rc == -1) snprintf(buf, sizeof buf, "panel source: internal order mismatch at rows [%lld, %lld)", (long long)row0, (long long)(row0 + rows));
                else snprintf(buf, sizeof buf, "panel source callback failed for rows [%lld, %lld)", (long long)row0, (long long)(row0 + rows));
                return fail(h, GPCA_ERR_BAD_ARG, buf);
            }
            int erc = host_rows_to_device(h, f, f.h_stage[b], f.stage_ld, rows, dst, st);
            hipError_t ee = erc == GPCA_OK ? hipEventRecord(f.ev_stage[b], st) : hipSuccess;
            {
                std::lock_guard<std::mutex> lk(f.m);
                // (on failure the buffer goes back as "in flight" too: its event is whatever was recorded last, long complete)
                f.st_state[b] = Filler::kInFlight; f.consumed++;
                f.cv.notify_all();
            }
            if (erc != GPCA_OK) return erc;
            HIPCHK(ee);
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
    if (rc == GPCA_OK) for (int64_t r0 = 0; r0 < M; r0 += cr) filler_post(f, r0, std::min(cr, M - r0));   // the worker stages ahead of the loop below
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

static double workspace_estimate(const gpca_handle* h);
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
        const double budget = 0.5 * ((double)free_b - workspace_estimate(h));
        int64_t rows = (int64_t)h->gq_waves_target * kGQRowsPerWave;
        const int64_t fit = (int64_t)(budget / ((double)ring_slots * (double)row_bytes));
        if (rows > fit) rows = fit;
        if (host_kind(src->kind)) {
            // host sources also need pinned host staging panels (three by default): keep each within 2 GiB (such a source is bound by
            // the host link, ~55 GB/s, long before the row-parallel K1 runs out of rows)
            const int64_t host_ld = bed_kind(src->kind) ? (N + 3) / 4 : N;
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
    {   // The stream that produces panels runs at the LOWEST priority the device offers: a device generator launches grids that fill the
        // chip for ~20 ms at a time, and behind them the pass's small kernels (a panel's column scales between its K1 and its K2: 0.13 ms of
        // work) waited ~4 ms each for a wave slot -- 1.5 s of a 5.9 s configs[4] step (profiles/r5_kbench_summary.md section 6).
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        HIPCHK(hipStreamCreateWithPriority(&sm.st_fill, hipStreamNonBlocking, lo));      // (lo = numerically greatest = least urgent)
    }
    sm.on = true;   // from here on stream_close() releases whatever was set up
    int rc = GPCA_OK;
    for (int i = 0; i < ring_slots && rc == GPCA_OK; ++i) {
        void* p = nullptr; hipEvent_t a = nullptr, b = nullptr;
        hipError_t e = malloc_genotypes(h, &p, (size_t)panel_rows * (size_t)row_bytes);
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

// M- and N-sized device memory gpca_snp_stats / gpca_rsvd will ask for at up to 64 sketch columns.  An upper estimate: per SNP
// row T (256 B), its digit planes (256 B), the loadings (<= 256 B), mu/sigma/r/b/keep/reason/counts (48 B); per sample Q, Y,
// the digit planes and the K2 partial sums of up to 16 column slices.
static double workspace_estimate(const gpca_handle* h) {
    return (double)h->Mpad * 1024.0 + (double)h->ldg * 8192.0 + 1073741824.0;
}

extern "C" int gpca_stream_set_cache(gpca_handle* h, int64_t max_bytes, int32_t* n_cached) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    StreamState& sm = h->sm;
    if (!sm.on) return fail(h, GPCA_ERR_STATE, "gpca_stream_set_cache: no panel stream open");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(sm.st_fill));
    HIPCHK(hipStreamSynchronize(h->st));
    const size_t panel_bytes = (size_t)sm.panel_rows * (size_t)(h->storage == GPCA_STORE_2BIT ? h->ld2 : h->ld8);
    if (max_bytes < 0) {   // what is free now, less the workspace the solver has not allocated yet and a 4 GiB margin
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const double left = (double)free_b + (double)sm.cache.size() * (double)panel_bytes - workspace_estimate(h) - 4.0 * 1073741824.0;
        max_bytes = left > 0 ? (int64_t)left : 0;
    }
    size_t want = (size_t)max_bytes / panel_bytes;
    if (want > (size_t)sm.n_panels) want = (size_t)sm.n_panels;
    while (sm.cache.size() > want) {
        (void)hipFree(sm.cache.back()); (void)hipEventDestroy(sm.ev_cache.back());
        sm.cache.pop_back(); sm.ev_cache.pop_back(); sm.cache_filled.pop_back();
    }
    int rc = GPCA_OK;
    while (sm.cache.size() < want) {
        void* p = nullptr; hipEvent_t e = nullptr;
        hipError_t err = malloc_genotypes(h, &p, panel_bytes);
        // zeroed like the ring slots: a HOST_I8 source writes N bytes per row, and the kernels' vector loads assume the bytes between
        // N and the row pitch are 0 (recycled device memory need not be)
        if (err == hipSuccess) err = hipMemsetAsync(p, 0, panel_bytes, sm.st_fill);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (err != hipSuccess) {
            if (p) (void)hipFree(p);
            (void)hipGetLastError();
            rc = fail(h, err == hipErrorOutOfMemory ? GPCA_ERR_OOM : GPCA_ERR_HIP, std::string("gpca_stream_set_cache: ") + hipGetErrorString(err)
                      + " after " + std::to_string(sm.cache.size()) + " panels (they stay cached)");
            break;
        }
        sm.cache.push_back(p); sm.ev_cache.push_back(e); sm.cache_filled.push_back(0);
    }
    if (n_cached) *n_cached = (int32_t)sm.cache.size();
    return rc;
}

extern "C" int gpca_stream_get_info(gpca_handle* h, gpca_stream_info* out) {
    if (!h || !out) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    StreamState& sm = h->sm;
    if (!sm.on) return fail(h, GPCA_ERR_STATE, "gpca_stream_get_info: no panel stream open");
    memset(out, 0, sizeof *out);
    out->panel_rows = sm.panel_rows; out->n_panels = sm.n_panels; out->ring_slots = sm.ring; out->n_cached = (int32_t)sm.cache.size();
    Filler& f = sm.fl;
    std::lock_guard<std::mutex> lk(f.m);
    out->staging_buffers = f.n_stage; out->zero_staging = f.registered ? 1 : 0; out->copy_threads = f.mapped && !f.registered ? f.copy_threads : 0;
    out->fills = f.fills; out->fill_host_ms = f.fill_host_ms; out->fill_wait_ms = f.fill_wait_ms; out->register_ms = f.register_ms;
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

// dst <- rows [row0, row0 + rows) of src's resident matrix, device to device (same device, same storage mode, same sample count by
// construction): an LD block becomes a matrix of its own, so that every M-sized step of a call on it costs the block's rows only.
extern "C" int gpca_copy_rows(gpca_handle* dst, gpca_handle* src, int64_t row0, int64_t rows) {
    if (!dst || !src || dst == src) return fail(dst, GPCA_ERR_BAD_ARG, "gpca_copy_rows: two different handles are required");
    std::lock(dst->mu, src->mu);
    std::lock_guard<std::recursive_mutex> g1(dst->mu, std::adopt_lock), g2(src->mu, std::adopt_lock);
    drain_pulls(dst);        // (pulls in flight on src only read it: they may go on)
    gpca_handle* h = dst;
    if (src->sm.on || (!src->dG && !src->dG2)) return fail(h, GPCA_ERR_STATE, "gpca_copy_rows: the source matrix is not resident");
    if (dst->storage_cfg == GPCA_STORE_AUTO && dst->storage != src->storage) {      // an AUTO destination takes the source's residency
        if (dst->sm.on) stream_close(dst);
        dfree(dst->dG); dfree(dst->dG2); dst->cap_rows_pad = 0;
        dst->storage = src->storage;
        if (dst->nd_cfg == 0) dst->nd = (dst->precision == GPCA_PREC_I8_EXACT && dst->storage == GPCA_STORE_2BIT) ? 3 : 4;
    }
    if (dst->device != src->device || dst->storage != src->storage) return fail(h, GPCA_ERR_BAD_ARG, "gpca_copy_rows: handles differ in device or storage mode");
    if (row0 < 0 || rows <= 0 || row0 + rows > src->M) return fail(h, GPCA_ERR_BAD_ARG, "gpca_copy_rows: row range outside the source matrix");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(src->st));
    const int64_t new_pad = round_up(rows, kGQRowsPerWave);
    const bool packed = dst->storage == GPCA_STORE_2BIT;
    const int64_t pitch = packed ? src->ld2 : src->ld8;
    if (!dst->sm.on && (packed ? dst->dG2 != nullptr : dst->dG != nullptr) && dst->N == src->N && new_pad <= dst->cap_rows_pad) {
        // the block fits what an earlier, larger block allocated: keep the genotype buffer, the statistics arrays and the
        // workspace (a per-LD-block loop would otherwise spend a third of its time in hipMalloc / hipFree); everything that
        // depends on the rows is marked stale, the rows past the block are zeroed (pad rows: zero genotypes, r = b = 0)
        HIPCHK(hipStreamSynchronize(dst->st));
        dst->M = rows; dst->Mpad = new_pad;
        dst->have_stats = false; dst->have_rsvd = false; dst->n_pca = 0; dst->pca_rows.clear(); dst->flags = 0; dst->apart_valid = false; dst->rmax_valid = false;
        drop_child(dst);
        free_eigensnp(dst);
        if (dst->d_r && dst->cap_stats_pad < new_pad) free_stats(dst);      // (allocated for a smaller block: the next stats pass re-makes them)
        if (dst->d_r) {
            const size_t n = (size_t)dst->cap_stats_pad;
            HIPCHK(hipMemsetAsync(dst->d_r, 0, n * 4, dst->st)); HIPCHK(hipMemsetAsync(dst->d_b, 0, n * 4, dst->st));
            HIPCHK(hipMemsetAsync(dst->d_keep, 0, n, dst->st)); HIPCHK(hipMemsetAsync(dst->d_counts, 0, n * 16, dst->st));
        }
        if (new_pad > rows) HIPCHK(hipMemsetAsync((packed ? (char*)dst->dG2 : (char*)dst->dG) + (size_t)rows * pitch, 0, (size_t)(new_pad - rows) * pitch, dst->st));
    } else {
        dst->auto_pin = dst->storage_cfg == GPCA_STORE_AUTO ? src->storage : 0;
        const int arc = alloc_genotypes(dst, rows, src->N);
        dst->auto_pin = 0;
        CHK(arc);
    }
    if (packed) HIPCHK(hipMemcpyAsync(dst->dG2, src->dG2 + (size_t)row0 * src->ld2, (size_t)rows * src->ld2, hipMemcpyDeviceToDevice, dst->st));
    else HIPCHK(hipMemcpyAsync(dst->dG, src->dG + (size_t)row0 * src->ld8, (size_t)rows * src->ld8, hipMemcpyDeviceToDevice, dst->st));
    dst->pack_flags = src->pack_flags;
    if (src->have_stats) {
        // the rows come with their statistics (mu, sigma, scale / shift, QC decision, genotype counts): the block needs no stats
        // pass of its own, gpca_set_standardization on dst only narrows the keep set
        CHK(alloc_stats(dst));
        const size_t o = (size_t)row0, n = (size_t)rows;
        HIPCHK(hipMemcpyAsync(dst->d_mu, src->d_mu + o, n * 4, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipMemcpyAsync(dst->d_sigma, src->d_sigma + o, n * 4, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipMemcpyAsync(dst->d_r, src->d_r + o, n * 4, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipMemcpyAsync(dst->d_b, src->d_b + o, n * 4, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipMemcpyAsync(dst->d_keep, src->d_keep + o, n, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipMemcpyAsync(dst->d_reason, src->d_reason + o, n, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipMemcpyAsync(dst->d_counts, src->d_counts + 4 * o, n * 16, hipMemcpyDeviceToDevice, dst->st));
        HIPCHK(hipStreamSynchronize(dst->st));
        CHK(refresh_pca_rows(dst));
        std::vector<uint32_t> counts(n * 4);
        HIPCHK(hipMemcpy(counts.data(), dst->d_counts, n * 16, hipMemcpyDeviceToHost));
        dst->flags = 0;                    // missing / invalid genotypes among the rows the block keeps (as gpca_set_standardization does)
        for (int64_t i : dst->pca_rows) {
            const uint32_t* c = &counts[(size_t)i * 4];
            if ((int64_t)c[0] != dst->N) dst->flags |= 1u;
            if ((uint64_t)c[1] + c[2] + c[3] != c[0]) dst->flags |= 2u;
        }
        dst->have_stats = true;
    }
    HIPCHK(hipStreamSynchronize(dst->st));
    return GPCA_OK;
}

extern "C" int gpca_get_device_memory(gpca_handle* h, int64_t* free_bytes, int64_t* total_bytes) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return GPCA_OK;
}

extern "C" int gpca_get_storage(gpca_handle* h, int32_t* storage, int32_t* precision) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK_SHARED(h);
    if (storage) *storage = (h->storage_cfg == GPCA_STORE_AUTO && !have_genotypes(h)) ? GPCA_STORE_AUTO : h->storage;
    if (precision) *precision = h->precision;
    return GPCA_OK;
}

extern "C" int gpca_dims(gpca_handle* h, int64_t* M, int64_t* N) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (M) *M = h->M;
    if (N) *N = h->N;
    return GPCA_OK;
}

