// The randomized-PCA driver: host eigen step, the stages over the resident matrix or the streamed panels, gpca_rsvd, result getters,
// gpca_transform (include/gpca.h section a5/a6).
#include "gpca_internal.h"

using namespace gpca;

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
    if (!a_sym || !w || !v || n < 1 || n > kMaxSketch) return GPCA_ERR_BAD_ARG;
    std::vector<double> A(a_sym, a_sym + (size_t)n * n), V((size_t)n * n), W((size_t)n);
    host_eigh_desc(A, V, W, n);
    std::copy(W.begin(), W.end(), w); std::copy(V.begin(), V.end(), v);
    return GPCA_OK;
}

// test hook (GPU): the device solver of gpca_rsvd (small_eig.hip) on a caller's symmetric n x n matrix, same conventions as above
extern "C" int gpca_device_eigh_desc(gpca_handle* h, const double* a_sym, int32_t n, double* w, double* v) {
    if (!h || !a_sym || !w || !v || n < 1 || n > kMaxSketch) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    const int L = n <= 32 ? 32 : (n <= 64 ? 64 : kMaxSketch);
    double *dA = nullptr, *dZz = nullptr, *dR = nullptr, *dV = nullptr;
    std::vector<double> A((size_t)L * L, 0.0), R((size_t)kEigResCount);
    for (int a = 0; a < n; ++a) for (int c = 0; c < n; ++c) A[(size_t)a * L + c] = a_sym[(size_t)a * n + c];
    int rc = GPCA_OK;
    auto run = [&]() -> int {
        HIPCHK(hipMalloc((void**)&dA, A.size() * 8)); HIPCHK(hipMalloc((void**)&dZz, 2 * (size_t)L * n * 8));
        HIPCHK(hipMalloc((void**)&dR, R.size() * 8)); HIPCHK(hipMalloc((void**)&dV, (size_t)n * n * 8));
        HIPCHK(hipMemcpyAsync(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice, h->st));
        launch_small_eigh(h->st, dA, 0, n, L, n, 1, 1.0, nullptr, dZz, dR, dV);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipMemcpyAsync(v, dV, (size_t)n * n * 8, hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        return GPCA_OK;
    };
    rc = run();
    dfree(dA); dfree(dZz); dfree(dR); dfree(dV);
    if (rc != GPCA_OK) return rc;
    std::copy(R.begin() + kEigResW, R.begin() + kEigResW + n, w);
    if (R[kEigResFlag + 1] != 0.0) return fail(h, GPCA_ERR_NOT_CONVERGED, "gpca_device_eigh_desc: the QL sweeps hit their cap");
    return GPCA_OK;
}

// ---- rsvd stages -----------------------------------------------------------------------------------------------
static int stage_sum_c(gpca_handle* h, int64_t parts) {
    launch_sum_partials_f32(h->st, h->d_cpart, parts, h->L, h->d_c, h->d_scratch64);
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}


// At most 256 samples on int8 rows (configs[2]'s shape class): the narrow kernels (gemm_i8.hip: k_gq_n, k_gtt_i8<.., NARROW>)
static inline bool narrow_shape(const gpca_handle* h) {
    return h->narrow_ok && h->precision == GPCA_PREC_I8_EXACT && h->storage == GPCA_STORE_INT8 && h->N <= kNarrowSamples;
}
// the DMA kernels (k_gtt_d on int8 rows, k_gtt_p on 2-bit rows) take several consecutive tasks per workgroup: one batch of workgroups per launch
static inline bool k2_batched(const gpca_handle* h) {
    return !narrow_shape(h) && !h->simple_kernels;      // (the choices of k2_panel)
}
static inline Gtt8Plan k2_plan(const gpca_handle* h, int64_t rows_pad) {
    return narrow_shape(h) ? gtt8_plan_narrow(rows_pad, h->N, std::min(h->gtt_waves_target, 1024)) :   // (one wave per SIMD: k_gtt_i8 holds 1 workgroup per CU)
           k2_batched(h) ? gtt8_plan_batched(rows_pad, h->ldg, h->gtt_waves_target) : gtt8_plan(rows_pad, h->ldg, h->gtt_waves_target);
}

static inline GqPlan k1_plan(const gpca_handle* h, int64_t rows_pad) { return gq_plan(rows_pad, h->gq_waves_target); }

// K2 of one 32-column half over one panel: Ypart = (digit planes of T')^T G, exact integers
static int k2_panel(gpca_handle* h, const PanelView& pv, const int8_t* Td_half, const Gtt8Plan& plan) {
    const int8_t* Td = Td_half + (size_t)(pv.row0 >> 5) * kPlaneBytesPerBlock;
    const bool packed = h->storage == GPCA_STORE_2BIT;
    if (narrow_shape(h)) launch_gtt_n(h->st, pv.g8, h->ld8, pv.rows_pad, h->ldg, Td, h->dYpart64, plan);
    else if (h->simple_kernels) {        // the register-only reference kernels
        if (packed) launch_gtt_2bit(h->st, pv.g2, h->ld2, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->nd);
        else launch_gtt_i8(h->st, pv.g8, h->ld8, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->ko);
    } else if (packed) {
        const int e = launch_gtt_p(h->st, pv.g2, h->ld2, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->nd, h->ko);
        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gtt_p launch failed (hip error " + std::to_string(e) + ")");
    } else {
        const int e = launch_gtt_d(h->st, pv.g8, h->ld8, pv.rows_pad, h->ldg, Td, h->dYpart64, plan, h->ko);
        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gtt_d launch failed (hip error " + std::to_string(e) + ")");
    }
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}

// Y = A^T T  (T' = r o T already in dT, c = b^T T in d_c); rank-local part, the exchange of Y follows in the caller
static int stage_AtT_local(gpca_handle* h, bool planes_ready = false) {
    const double elems = (double)h->M * (double)h->N;
    if (h->precision == GPCA_PREC_I8_EXACT) {
        // T' (f32 row-major in dT) -> digit planes; exact int8 product; integer partials summed exactly in f64.
        // The kernels are 32 columns wide: a 64-column sketch (32 < l <= 64) runs as two column halves over the same genotypes.
        const int L = h->L, halves = L / 32;
        const size_t td_half = (size_t)h->Mpad * 32 * kDigits;
        for (int hf = 0; hf < halves && !planes_ready; ++hf) {      // (planes_ready: the sketch's digit planes came straight out of k_omega)
            const float* Th = h->dT + 32 * hf;
            double* tsc = h->d_tscale + 32 * hf; double* tin = h->d_tinv + 32 * hf;
            if (h->c_fold_pending) launch_quantize_f32_cfold(h->st, Th, h->Mpad, h->Mpad, tin, h->dTd + hf * td_half, 0, h->nd, L, h->d_scratch64 + (size_t)hf * kPostK1Scratch,
                                                             post_k1_slices(h->Mpad / 32), h->d_c + 32 * hf);      // (scale by launch_post_k1; c's second stage rides along)
            else if (h->apart_valid) launch_quantize_f32_premax(h->st, Th, h->Mpad, h->Mpad, h->apart_src[hf], h->apart_parts, tsc, tin, h->dTd + hf * td_half, 0, h->nd, L);
            else launch_quantize_f32(h->st, Th, h->Mpad, h->Mpad, h->d_part64, tsc, tin, h->dTd + hf * td_half, 0, h->nd, L);
            HIPCHK(hipGetLastError());
        }
        h->apart_valid = false; h->c_fold_pending = false;
        const bool streamed = h->sm.on;
        const size_t yint_half = (size_t)h->N * 32;
        {
            // resident: one record per launch (the roofline figure of bench.py); streamed: one record per sweep over the panels
            const double by = h->storage == GPCA_STORE_2BIT ? elems / 4 : elems;
            ScopedTimer sweep(h, "gemm_GtT", 2.0 * elems * h->l, by * halves, nullptr, streamed);
            CHK(for_each_panel(h, [&](const PanelView& pv) -> int {
                const Gtt8Plan plan = streamed ? k2_plan(h, pv.rows_pad) : h->plan8;
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
        if (streamed && scale_out) HIPCHK(hipMemsetAsync(h->d_amax_run, 0, kMaxSketch * 8, h->st));
        {
            const double by = packed ? elems / 4 : elems;
            ScopedTimer sweep(h, "gemm_GQ", 2.0 * elems * h->l, by * halves, nullptr, streamed);
            CHK(for_each_panel(h, [&](const PanelView& pv) -> int {
                const GqPlan plan = streamed ? k1_plan(h, pv.rows_pad) : h->gqplan;
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
                    else if (narrow_shape(h)) {
                        const int e = launch_gq_n(h->st, pv.g8, h->ld8, plan, h->N, Qd, qsc, rr, bb, s32, Th, cp, ap, scale_out, L);
                        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gq_n launch failed (hip error " + std::to_string(e) + ")");
                    }
                    else if (!h->simple_kernels) {
                        const int e = launch_gq_d(h->st, pv.g8, h->ld8, plan, h->ldg, Qd, qsc, rr, bb, s32, Th, cp, ap, scale_out, L, h->ko);
                        if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gq_d launch failed (hip error " + std::to_string(e) + ")");
                    }
                    else launch_gq_i8(h->st, pv.g8, h->ld8, plan, h->N, Qd, qsc, rr, bb, s32, Th, cp, scale_out, L, h->ko);
                    HIPCHK(hipGetLastError());
                    if (streamed && scale_out && (packed || !h->simple_kernels || narrow_shape(h))) {   // (streamed: the per-launch timer above is disabled) fold this panel's column abs-max before the next launch reuses ap
                        launch_absmax_fold(h->st, ap, plan.waves, h->d_amax_run + 32 * hf);
                        HIPCHK(hipGetLastError());
                    }
                }
                return GPCA_OK;
            }));
        }
        h->apart_valid = scale_out != 0 && (packed || !h->simple_kernels || narrow_shape(h));   // (k_gq_i8 has no abs-max epilogue)
        // c = b^T T of every half: one partial per 32-row unit, summed in a fixed two-stage tree.  Resident matrices whose K1 left the waves'
        // column abs-max: the first stage runs beside the fold of those into the digit scale (one launch), the second rides the
        // quantisation that follows in stage_AtT_local -- two launches between K1 and K2 instead of four, the same bits.
        h->c_fold_pending = scale_out != 0 && !streamed && h->apart_valid;
        if (scale_out)
            for (int hf = 0; hf < halves; ++hf) {
                if (h->c_fold_pending)
                    launch_post_k1(h->st, h->d_cpart + hf * chalf, h->Mpad / 32, h->d_scratch64 + (size_t)hf * kPostK1Scratch, h->d_apart + hf * ahalf, h->gqplan.waves,
                                   h->d_tscale + 32 * hf, h->d_tinv + 32 * hf, h->nd);
                else launch_sum_partials_f32(h->st, h->d_cpart + hf * chalf, h->Mpad / 32, 32, h->d_c + 32 * hf, h->d_scratch64);
                HIPCHK(hipGetLastError());
            }
        for (int hf = 0; hf < kMaxSketch / 32; ++hf) h->apart_src[hf] = streamed ? h->d_amax_run + 32 * hf : h->d_apart + hf * ahalf;
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
            const GqPlan plan1 = k1_plan(h, pv.rows_pad);
            const Gtt8Plan plan2 = k2_plan(h, pv.rows_pad);
            const float* rr = h->d_r + pv.row0; const float* bb = h->d_b + pv.row0;
            for (int hf = 0; hf < halves; ++hf) {
                const int8_t* Qd = h->dQd + hf * qhalf;
                float* Th = h->dT + (size_t)pv.row0 * L + 32 * hf;
                float* cp = h->d_cpart + hf * chalf + (size_t)pv.row0;
                double* ap = h->d_apart + hf * ahalf;
                if (packed) launch_gq_2bit(h->st, pv.g2, h->ld2, plan1, h->ldg, Qd, h->d_qscale + 32 * hf, rr, bb, h->d_s32 + 32 * hf, Th, cp, ap, 1, h->nd, L);
                else if (narrow_shape(h)) {
                    const int e = launch_gq_n(h->st, pv.g8, h->ld8, plan1, h->N, Qd, h->d_qscale + 32 * hf, rr, bb, h->d_s32 + 32 * hf, Th, cp, ap, 1, L);
                    if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gq_n launch failed (hip error " + std::to_string(e) + ")");
                }
                else {
                    const int e = launch_gq_d(h->st, pv.g8, h->ld8, plan1, h->ldg, Qd, h->d_qscale + 32 * hf, rr, bb, h->d_s32 + 32 * hf, Th, cp, ap, 1, L, h->ko);
                    if (e != 0) return fail(h, GPCA_ERR_HIP, "k_gq_d launch failed (hip error " + std::to_string(e) + ")");
                }
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

// CholeskyQR of dY -> dQ (f32, padded), s = 1^T Q.  rounds = 2 (CholeskyQR2): orthonormal to rounding -- the basis the projection and
// the scores are built on.  rounds = 1: the basis of an INTERMEDIATE power iteration, which only has to span range(Y) with a modest
// condition number: after one round ||Q^T Q - I|| ~ cond(Y)^2 eps (1e-11 at a spectrum ratio of 500), and wherever that is not small
// the second round could not have repaired the first either.  A third of the Gram / Cholesky / apply launches of a call (q = 2) go.
static int stage_orth(gpca_handle* h, int rounds = 2) {
    const int L = h->L, l = h->l;
    if (h->d_smask) {   // basis learning on a sample subset (gpca_set_sample_mask): the other samples' rows leave the sketch
        launch_mask_rows(h->st, h->dY, h->N, L, h->d_smask);
        HIPCHK(hipGetLastError());
    }
    for (int round = 0; round < rounds; ++round) {      // entirely on the stream (no host round trip)
        const int64_t parts = gram_num_parts(h->N);
        launch_gram_f64(h->st, h->dY, h->N, L, h->d_part64);
        HIPCHK(hipGetLastError());
        if (L == 32 && parts <= 64) launch_chol_inv_fold(h->st, h->d_part64, (int)parts, l, L, h->dZ, h->d_cholflag);     // (the fold of the partial Grams rides in front)
        else {
            launch_sum_partials_f64(h->st, h->d_part64, parts, (int64_t)L * L, h->dW, h->d_scratch64);
            HIPCHK(hipGetLastError());
            launch_chol_inv(h->st, h->dW, l, L, h->dZ, h->d_cholflag);
        }
        HIPCHK(hipGetLastError());
        if (round + 1 < rounds) launch_apply_right_inplace(h->st, h->dY, h->N, L, h->dZ, nullptr, h->ldg);
        else launch_apply_right_tail(h->st, h->dY, h->N, L, h->dZ, h->dQ, h->ldg, h->d_part64, h->d_part64 + tail_num_parts(h->ldg) * L);
        HIPCHK(hipGetLastError());
    }
    // dY now holds the orthonormal basis in f64: s = Q^T 1 and (exact-integer path) the digit scale of Q, then its planes
    const bool i8 = h->precision == GPCA_PREC_I8_EXACT;
    const int64_t tparts = tail_num_parts(h->ldg);
    const double* csum_part = h->d_part64;
    const double* amax_part = h->d_part64 + tparts * L;
    if (i8 && tparts <= kFinishQFoldMax) {      // one launch per 32 columns: the quantisation folds the tail's partials itself
        for (int hf = 0; hf < L / 32; ++hf) {
            launch_quantize_f64_finishq(h->st, h->dY + 32 * hf, h->N, h->ldg, h->dQd + (size_t)hf * h->ldg * 32 * kDigits, h->storage == GPCA_STORE_2BIT ? 1 : 0, h->nd, L,
                                        csum_part + 32 * hf, amax_part + 32 * hf, tparts, L, h->d_s64 + 32 * hf, h->d_s32 + 32 * hf, h->d_qscale + 32 * hf, h->d_qinv + 32 * hf);
            HIPCHK(hipGetLastError());
        }
        return GPCA_OK;
    }
    launch_finish_q(h->st, csum_part, amax_part, tparts, L, h->d_s64, h->d_s32, i8 ? h->d_qscale : nullptr, i8 ? h->d_qinv : nullptr, h->nd);
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
    h->gqplan = k1_plan(h, gemm_rows);
    CHK(ensure(h, h->dQ, h->cap_Q, (size_t)Npad * L));
    CHK(ensure(h, h->dT, h->cap_T, (size_t)h->Mpad * L));
    if (h->precision == GPCA_PREC_F32_MFMA) CHK(ensure(h, h->dTb, h->cap_Tb, (size_t)h->Mpad * L));
    if (h->Mpad > M) HIPCHK(hipMemsetAsync(h->dT + (size_t)M * L, 0, (size_t)(h->Mpad - M) * L * 4, h->st));
    if (h->precision == GPCA_PREC_F32_MFMA) CHK(ensure(h, h->dYpart, h->cap_Ypart, (size_t)h->plan.W * (size_t)Npad * L));
    // c partials: per wave x L (f32 path, Omega: 64-row groups), or per 32-row unit x 32 per column half (exact path)
    const int64_t cparts = std::max({h->gqplan.waves * (int64_t)L, omega_num_parts(h->Mpad) * (int64_t)L, h->Mpad * (int64_t)(L / 32)});
    CHK(ensure(h, h->d_cpart, h->cap_cpart, (size_t)cparts));
    CHK(ensure(h, h->dY, h->cap_Y, (size_t)N * L + 16));   // (+ 16 status slots: gpca_transform's agreement rides its exchange)
    const int64_t p64 = std::max({gram_num_parts(N) * (int64_t)L * L, gram_num_parts(M) * (int64_t)L * L, colsum_num_parts(Npad) * (int64_t)L, absmax_num_parts(h->Mpad) * (int64_t)32, 2 * tail_num_parts(Npad) * (int64_t)L});
    CHK(ensure(h, h->d_part64, h->cap_part64, (size_t)p64));
    if (!h->d_c) {
        constexpr size_t LL = (size_t)kMaxSketch * kMaxSketch;
        HIPCHK(hipMalloc((void**)&h->d_c, kMaxSketch * 8)); HIPCHK(hipMalloc((void**)&h->d_s64, kMaxSketch * 8));
        HIPCHK(hipMalloc((void**)&h->d_s32, kMaxSketch * 4)); HIPCHK(hipMalloc((void**)&h->dW, (LL + 16) * 8));   // (+ the 16 status slots that ride the Gram's exchange)
        HIPCHK(hipMalloc((void**)&h->dZ, 2 * LL * 8));
        HIPCHK(hipHostMalloc((void**)&h->h_pin, (3 * LL + 16) * 8, hipHostMallocDefault)); HIPCHK(hipMalloc((void**)&h->d_sign, kMaxSketch * 4));
        HIPCHK(hipMalloc((void**)&h->d_scratch64, kSumScratchElems * 8));
        HIPCHK(hipMalloc((void**)&h->d_eigres, kEigResCount * 8));
        HIPCHK(hipMalloc((void**)&h->d_cand_val, (size_t)64 * kMaxSketch * 8)); HIPCHK(hipMalloc((void**)&h->d_cand_idx, (size_t)64 * kMaxSketch * 8));      // (scores_num_parts <= 48)
        HIPCHK(hipMalloc((void**)&h->d_cholflag, 4));
    }
    if (h->precision == GPCA_PREC_I8_EXACT) {
        h->plan8 = k2_plan(h, gemm_rows);
        CHK(ensure(h, h->dQd, h->cap_Qd, (size_t)Npad * 32 * kDigits * (size_t)(L / 32)));
        CHK(ensure(h, h->dTd, h->cap_Td, (size_t)h->Mpad * 32 * kDigits * (size_t)(L / 32)));
        // (streamed: a shorter last panel may be cut into more row chunks than the full panels: the plan of every panel size is asked)
        int64_t yslices = h->plan8.W;
        if (h->sm.on && h->M % h->sm.panel_rows) yslices = std::max<int64_t>(yslices, k2_plan(h, round_up(h->M % h->sm.panel_rows, kGQRowsPerWave)).W);
        CHK(ensure(h, h->dYpart64, h->cap_Ypart64, (size_t)yslices * (size_t)Npad * 32));
        CHK(ensure(h, h->d_apart, h->cap_apart, (size_t)h->gqplan.waves * 32 * (size_t)(L / 32)));
        if (h->sm.on) CHK(ensure(h, h->d_yint, h->cap_yint, (size_t)N * 32 * (size_t)(L / 32)));
        if (!h->d_qscale) {
            HIPCHK(hipMalloc((void**)&h->d_qscale, kMaxSketch * 8)); HIPCHK(hipMalloc((void**)&h->d_qinv, kMaxSketch * 8));
            HIPCHK(hipMalloc((void**)&h->d_tscale, kMaxSketch * 8)); HIPCHK(hipMalloc((void**)&h->d_tinv, kMaxSketch * 8));
            HIPCHK(hipMalloc((void**)&h->d_amax_run, kMaxSketch * 8));
            HIPCHK(hipMalloc((void**)&h->d_rmax, 4));
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
    if (l > kMaxSketch) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: k + oversample must be <= 128");
    if (l > 64 && h->precision != GPCA_PREC_I8_EXACT)
        return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: sketches wider than 64 columns (k + oversample > 64) run on GPCA_PREC_I8_EXACT; GPCA_PREC_F32_MFMA holds up to 64");
    if (l > h->N || (!multi_rank(h) && l > h->n_pca)) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: k + oversample exceeds min(samples, PCA SNPs)");
    if (h->flags & 1u) return fail(h, GPCA_ERR_MISSING_GENOTYPE,
        "Unexpected missing genotype (-127i8) in a PCA SNP. This should have been filtered by QC.");  // prepare.rs:1909-1911
    if (h->flags & 2u) return fail(h, GPCA_ERR_INVALID_GENOTYPE, "a PCA SNP holds a dosage outside {0,1,2}");
    if (h->sm.on && h->precision != GPCA_PREC_I8_EXACT) return fail(h, GPCA_ERR_STATE, "gpca_rsvd: streamed panels need GPCA_PREC_I8_EXACT");
    // K1 sums a row's products over all samples in i32 digit-plane accumulators: |g| <= 2 times |digit| <= 128 per sample
    if (h->precision == GPCA_PREC_I8_EXACT && h->N > ((int64_t)1 << 22))
        return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd: the exact-integer path holds up to 4 194 304 samples per matrix (i32 accumulators); use GPCA_PREC_F32_MFMA beyond");
    HIPCHK(hipSetDevice(h->device));
    h->k = k; h->l = l; h->L = l <= 32 ? 32 : (l <= 64 ? 64 : kMaxSketch);
    h->have_rsvd = false;
    CHK(ensure_workspace(h));
    HIPCHK(hipMemsetAsync(h->d_cholflag, 0, 4, h->st));
    return GPCA_OK;
}

// ---- the small dense step and what hangs on it, enqueued without a host wait (small_eig.hip, kernels.hip) -----------------------
// eigen-decomposition of the l x l Gram (at src: the matrix itself, or `nslices` partial sums of it) -> dZ = [Z0 | Z1] (L x k each), d_eigres
static int enqueue_small_eigh(gpca_handle* h, const double* src, int nslices, int zmode, double denom) {
    launch_small_eigh(h->st, src, nslices, h->l, h->L, h->k, zmode, denom, h->d_cholflag, h->dZ, h->d_eigres, nullptr);
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}
// scores = Y Z0 with the sign rule (dY holds the orthonormal basis or the refined sample factor); loadings = X[pca rows] (Z1 o sign) when asked
static int enqueue_scores_loadings(gpca_handle* h, const float* Xload, bool loadings) {
    const int L = h->L, k = h->k;
    const size_t zk = (size_t)L * k;
    if (L <= 64) {
        launch_scores(h->st, h->dY, h->N, L, h->dZ, k, h->d_scores64, h->d_cand_val, h->d_cand_idx);
        HIPCHK(hipGetLastError());
        launch_scores_sign(h->st, h->d_scores64, h->d_scores32, h->N, k, h->d_cand_val, h->d_cand_idx, scores_num_parts(h->N), h->d_sign);
        HIPCHK(hipGetLastError());
        if (loadings) launch_rightmul_gather_f32(h->st, Xload, h->d_pca_rows, h->n_pca, L, h->dZ + zk, k, h->d_load32, h->d_sign);
    } else {                                                     // wide sketches: the plain any-L kernels
        launch_rightmul_f64(h->st, h->dY, h->N, L, h->dZ, k, h->d_scores64, (float*)nullptr);
        HIPCHK(hipGetLastError());
        launch_col_sign(h->st, h->d_scores64, h->N, k, h->d_sign);
        launch_scale_cols(h->st, h->d_scores64, h->d_scores32, h->N, k, h->d_sign);
        HIPCHK(hipGetLastError());
        if (loadings) {
            launch_scale_cols(h->st, h->dZ + zk, (float*)nullptr, L, k, h->d_sign);
            launch_rightmul_gather_f32(h->st, Xload, h->d_pca_rows, h->n_pca, L, h->dZ + zk, k, h->d_load32);
        }
    }
    HIPCHK(hipGetLastError());
    return GPCA_OK;
}
// the call's host wait: the result block comes back through pinned staging behind everything enqueued so far.  take = false: wait only.
static int finish_small_eigh(gpca_handle* h, bool take) {
    double* res = h->h_pin;
    HIPCHK(hipMemcpyAsync(res, h->d_eigres, sizeof(double) * kEigResCount, hipMemcpyDeviceToHost, h->st));
    HIPCHK(stream_wait(h));
    if (!take) return GPCA_OK;
    const int flag = (int)res[kEigResFlag];
    if (flag) {   // (computed redundantly on the replicated Y: the same on every rank)
        char buf[160];
        snprintf(buf, sizeof buf, "CholeskyQR: pivot %d of the %d-column sketch is not finite (overflow or NaN in the sketch)", flag - 1, h->l);
        return fail(h, GPCA_ERR_NOT_CONVERGED, buf);
    }
    h->sv.assign(res + kEigResSv, res + kEigResSv + h->l);
    h->eig.assign(res + kEigResEig, res + kEigResEig + h->k);
    return GPCA_OK;
}

// ---- compact child (gpca_internal.h): the kept rows as a matrix of their own -------------------------------------------------------
// Worth it when QC dropped at least half of the rows; only for a resident, unsharded matrix without a sample mask (the EigenSNP stage
// calls work on the handle itself), and only when the copy fits beside the parent with room for the solver's workspace.
static bool wants_child(gpca_handle* h, int32_t k, int32_t oversample) {
    if (!h->compact_ok || h->is_child || h->sm.on || multi_rank(h) || h->d_smask || !h->have_stats || (!h->dG && !h->dG2)) return false;
    if (h->n_pca < 1 || k + oversample > h->n_pca) return false;                       // (the usual preflight reports these)
    if ((h->flags & 3u) != 0) return false;                                             // (missing / invalid genotypes: ditto)
    // ... and only when it pays.  A child costs a workspace of its own (~35 allocations), a gather and, when the keep mask changes, the
    // matching frees: milliseconds.  A call over a few thousand rows gains microseconds from losing half of them -- the per-block local
    // stage of the EigenSNP hosts re-targets ONE scratch handle at thousands of ~300-row LD blocks (copy_rows + set_standardization +
    // rsvd), where compaction turned a 1.7 ms block into allocator traffic.  Worth it from 32 Ki dropped rows AND 8 MiB of genotype
    // bytes that no longer stream six times per call (configs[2]: 860k of 1 066 557 rows, 220 MB).
    const int64_t npad = round_up(h->n_pca, kGQRowsPerWave), dropped = h->Mpad - npad;
    const int64_t pitch = h->storage == GPCA_STORE_2BIT ? h->ld2 : h->ld8;
    if (dropped < 32768 || dropped * pitch < ((int64_t)8 << 20)) return false;
    return 2 * npad <= h->Mpad;
}
static int ensure_child(gpca_handle* h) {
    if (h->child && h->child_valid) return GPCA_OK;
    drop_child(h);
    const bool packed = h->storage == GPCA_STORE_2BIT;
    const int64_t n = h->n_pca, npad = round_up(n, kGQRowsPerWave), pitch = packed ? h->ld2 : h->ld8;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const double need = (double)npad * (double)pitch + (double)npad * 1024.0 + (double)h->ldg * 8192.0 + 1073741824.0;
    if (need > (double)free_b) return GPCA_ERR_OOM;                                     // (the caller carries on without the child)
    gpca_handle* c = new gpca_handle();
    c->is_child = true; c->compact_ok = 0;
    c->device = h->device; c->precision = h->precision; c->storage = h->storage; c->nd = h->nd; c->ko = h->ko;
    c->st = h->st;                                                                      // same stream: ordered with the parent's work
    c->simple_kernels = h->simple_kernels; c->narrow_ok = h->narrow_ok; c->spin_sync = h->spin_sync;
    c->gq_waves_target = h->gq_waves_target; c->gtt_waves_target = h->gtt_waves_target;
    c->timing_on = h->timing_on;
    c->M = n; c->N = h->N; c->Mpad = npad; c->ldg = h->ldg; c->ld8 = h->ld8; c->ld2 = h->ld2; c->cap_rows_pad = npad; c->pack_flags = h->pack_flags;
    h->child = c;                                                                       // (from here on drop_child releases whatever exists)
    auto build = [&]() -> int {
        gpca_handle* hh = h;
        (void)hh;
        if (packed) { HIPCHK(malloc_genotypes(h, (void**)&c->dG2, (size_t)npad * (size_t)pitch)); }
        else { HIPCHK(malloc_genotypes(h, (void**)&c->dG, (size_t)npad * (size_t)pitch)); }
        char* gdst = packed ? (char*)c->dG2 : (char*)c->dG;
        if (npad > n) HIPCHK(hipMemsetAsync(gdst + (size_t)n * pitch, 0, (size_t)(npad - n) * pitch, h->st));
        launch_gather_rows(h->st, packed ? (const void*)h->dG2 : (const void*)h->dG, pitch, h->d_pca_rows, n, gdst);
        HIPCHK(hipGetLastError());
        {
            gpca_handle* h = c;     // (alloc_stats reports through the handle it is given)
            CHK(alloc_stats(h));
        }
        launch_gather_elems(h->st, h->d_mu, 4, h->d_pca_rows, n, c->d_mu); launch_gather_elems(h->st, h->d_sigma, 4, h->d_pca_rows, n, c->d_sigma);
        launch_gather_elems(h->st, h->d_r, 4, h->d_pca_rows, n, c->d_r); launch_gather_elems(h->st, h->d_b, 4, h->d_pca_rows, n, c->d_b);
        launch_gather_elems(h->st, h->d_counts, 16, h->d_pca_rows, n, c->d_counts);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(c->d_keep, 1, (size_t)n, h->st)); HIPCHK(hipMemsetAsync(c->d_reason, 0, (size_t)npad, h->st));
        c->pca_rows.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) c->pca_rows[(size_t)i] = i;
        c->n_pca = n;
        HIPCHK(hipMalloc((void**)&c->d_pca_rows, (size_t)n * 8));
        HIPCHK(hipMemcpyAsync(c->d_pca_rows, c->pca_rows.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        c->d_row_ids = h->d_pca_rows;                                                   // Omega by the rows' ORIGINAL index
        c->snp_offset = h->snp_offset;
        c->flags = 0; c->have_stats = true;
        return GPCA_OK;
    };
    const int rc = build();
    if (rc != GPCA_OK) { const std::string keep = c->err.empty() ? h->err : c->err; drop_child(h); h->err = keep; return rc; }
    h->child_valid = true;
    return GPCA_OK;
}

extern "C" int gpca_rsvd(gpca_handle* h, int32_t k, int32_t oversample, int32_t power_iters, uint64_t seed) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    h->rsvd_on_child = false;
    // sampled timings (gpca_enable_timings(h, n)): a call that is not the n-th records no events
    struct SkipGuard { gpca_handle* h; ~SkipGuard() { h->timing_skip = false; } } skip_guard{h};
    if (!h->is_child) h->timing_skip = h->timing_on && h->timing_every > 1 && (h->timing_calls++ % h->timing_every) != 0;
    if (wants_child(h, k, oversample) && ensure_child(h) == GPCA_OK) {
        // QC dropped most rows: the whole call runs on the kept rows gathered into a matrix of their own
        gpca_handle* c = h->child;
        c->timing_on = h->timing_on; c->timing_skip = h->timing_skip;
        const int rc = gpca_rsvd(c, k, oversample, power_iters, seed);
        if (rc != GPCA_OK) { h->err = c->err; h->have_rsvd = false; return rc; }
        h->k = c->k; h->l = c->l; h->L = c->L; h->eig = c->eig; h->sv = c->sv;
        h->have_rsvd = true; h->loadings_valid = true; h->rsvd_on_child = true;
        return GPCA_OK;
    }
    // Ranks of a sharded run leave together: they agree on the preflight status before the first exchange.  The agreement is
    // enqueued here and read just before that exchange (agree_status_begin / _end): a clean rank enqueues its sketch meanwhile, so the
    // host never stands still for the round trip; a rank whose preflight failed waits for the verdict and leaves.
    const bool mr = multi_rank(h);
    const int pre = rsvd_preflight(h, k, oversample, power_iters);
    if (!mr) { if (pre != GPCA_OK) return pre; }
    else {
        CHK(agree_status_begin(h, pre));
        if (pre != GPCA_OK) return agree_status_end(h, pre, "gpca_rsvd (before the sketch)");
    }
    bool agreement_pending = mr;
    int lrc = GPCA_OK;
    const int l = h->l, L = h->L;
    // ... and from here on a rank-local failure is remembered (lrc) while the rank keeps entering every exchange of the call, so
    // that its peers are not left inside a collective; the second agreement below returns the failure on every rank.
#define LOCAL(x) do { if (lrc == GPCA_OK) lrc = (x); if (lrc != GPCA_OK && !mr) return lrc; } while (0)
#define EXCHANGE(buf, count) do { \
        if (agreement_pending) { \
            agreement_pending = false; \
            const int arc_ = agree_status_end(h, GPCA_OK, "gpca_rsvd (before the sketch)"); \
            if (arc_ != GPCA_OK) { (void)hipStreamSynchronize(h->st); return arc_; }   /* a peer's preflight failed: nobody enters the exchange */ \
        } \
        const int xrc_ = allreduce_f64(h, (buf), (count)); if (xrc_ != GPCA_OK) return xrc_; } while (0)
    auto omega = [&]() -> int {
        // 1. sketch: T' = r o Omega, c = b^T Omega
        ScopedTimer t(h, "omega", 0.0, (double)h->M * L * 4.0);
        if (h->precision == GPCA_PREC_I8_EXACT) {
            // the digit planes of T' leave k_omega directly, scaled by the analytic bound 6.67 * max r (no f32 T', no quantisation pass)
            if (!h->rmax_valid) {
                HIPCHK(hipMemsetAsync(h->d_rmax, 0, 4, h->st));
                launch_max_f32(h->st, h->d_r, h->Mpad, h->d_rmax);
                HIPCHK(hipGetLastError());
                h->rmax_valid = true;
            }
            launch_omega_planes(h->st, h->M, h->Mpad, l, L, h->snp_offset, seed, h->d_r, h->d_b, h->d_cpart, h->dTd, h->d_rmax, h->d_tscale, h->d_tinv, h->nd, h->d_row_ids);
            h->apart_valid = false;
        } else launch_omega(h->st, h->M, h->Mpad, l, L, h->snp_offset, seed, h->d_r, h->d_b, h->dTb, h->d_cpart, nullptr, 1, h->d_row_ids);
        HIPCHK(hipGetLastError());
        return GPCA_OK;
    };
    LOCAL(omega());
    LOCAL(stage_sum_c(h, omega_num_parts(h->Mpad)));
    LOCAL(stage_AtT_local(h, h->precision == GPCA_PREC_I8_EXACT));                       // Y = A^T Omega (exact path: planes already made)
    EXCHANGE(h->dY, h->N * (int64_t)L);
    LOCAL(stage_orth(h, power_iters == 0 ? 2 : 1));   // (the last basis before the projection gets CholeskyQR2, the earlier ones one round)
    // 2. power iterations
    const bool fused = h->sm.on && h->sm.fused && (h->storage == GPCA_STORE_2BIT || !h->simple_kernels || narrow_shape(h));   // (k_gq_i8 has no abs-max epilogue)
    for (int it = 0; it < power_iters; ++it) {
        if (fused) LOCAL(stage_power_fused(h));
        else { LOCAL(stage_AQ(h, 1)); LOCAL(stage_AtT_local(h)); }
        EXCHANGE(h->dY, h->N * (int64_t)L);
        LOCAL(stage_orth(h, it + 1 == power_iters ? 2 : 1));
    }
    // 3. projection B = A Q, the l x l eigenproblem of B^T B, scores, loadings: all enqueued, no host step in between.
    LOCAL(stage_AQ(h, 0));
    const double* gsrc = h->dW; int gslices = 0;
    auto gram_b = [&]() -> int {
        const int64_t parts = gram_num_parts(h->M);
        launch_gram_f32(h->st, h->dT, h->M, L, h->d_part64);
        HIPCHK(hipGetLastError());
        if (mr) launch_sum_partials_f64(h->st, h->d_part64, parts, (int64_t)L * L, h->dW, h->d_scratch64);      // the exchange needs the rank's sum in one place
        else launch_sum_partials_f64_stage1(h->st, h->d_part64, parts, (int64_t)L * L, h->d_scratch64, &gsrc, &gslices);   // (the eigen kernel folds the slices)
        HIPCHK(hipGetLastError());
        return GPCA_OK;
    };
    LOCAL(gram_b());
    // the second agreement rides the Gram's exchange: every rank appends its status histogram to the l x l block (no round trip of its own)
    if (mr) {
        h->status_own = h->err;
        status_histogram(h->h_status, lrc);
        if (hipMemcpyAsync(h->dW + (size_t)L * L, h->h_status, 16 * sizeof(double), hipMemcpyHostToDevice, h->st) != hipSuccess && lrc == GPCA_OK)
            lrc = fail(h, GPCA_ERR_HIP, "gpca_rsvd: status copy failed");
    }
    EXCHANGE(h->dW, (int64_t)L * L + (mr ? 16 : 0));
    // variance over the samples that took part: all N, or the subset of gpca_set_sample_mask (the other rows of the sketch are zero)
    const double n_eff = h->d_smask ? (double)h->n_smask : (double)h->N;
    // 4. C = V diag(w) V^T on the device (small_eig.hip); scores = Q V_k diag(s), sign (largest |score| positive), loadings = B V_k diag(sign / s)
    auto tail = [&]() -> int {
        CHK(enqueue_small_eigh(h, gsrc, gslices, 0, n_eff - 1.0));
        return enqueue_scores_loadings(h, h->dT, true);
    };
    LOCAL(tail());
    // The call's one host wait: singular values, eigenvalues, the pivot flag and (sharded runs) the agreed status arrive together.
    if (mr && hipMemcpyAsync(h->h_status + 16, h->dW + (size_t)L * L, 16 * sizeof(double), hipMemcpyDeviceToHost, h->st) != hipSuccess)
        return lrc != GPCA_OK ? lrc : fail(h, GPCA_ERR_HIP, "gpca_rsvd: status fetch failed");
    {
        const int frc = finish_small_eigh(h, lrc == GPCA_OK);      // (waits for the stream; a rank that failed earlier only waits)
        if (lrc == GPCA_OK) lrc = frc;
    }
    if (mr) {
        const std::string own_now = h->err;
        const int agreed = status_verdict(h, h->h_status + 16, lrc, h->status_own, "gpca_rsvd (after the last exchange)");
        if (agreed != GPCA_OK) return agreed;
        h->err = own_now;
    }
    if (lrc != GPCA_OK) return lrc;
#undef LOCAL
#undef EXCHANGE
    h->have_rsvd = true; h->loadings_valid = true;
    return GPCA_OK;
}

#define NEED_RSVD(name) \
    if (!h || !out) return GPCA_ERR_BAD_ARG; \
    LOCK(h); \
    if (!h->have_rsvd) return fail(h, GPCA_ERR_STATE, name ": run gpca_rsvd first")
// the last gpca_rsvd ran on the compact child: its results are this handle's (scores per sample; loadings per PCA SNP, in their order)
#define ON_CHILD(call) \
    if (h->rsvd_on_child && h->child) { const int rc_ = (call); if (rc_ != GPCA_OK) h->err = h->child->err; return rc_; }

extern "C" int gpca_get_scores(gpca_handle* h, float* out) {
    NEED_RSVD("gpca_get_scores");
    ON_CHILD(gpca_get_scores(h->child, out));
    HIPCHK(hipMemcpy(out, h->d_scores32, (size_t)h->N * h->k * 4, hipMemcpyDeviceToHost));
    return GPCA_OK;
}
extern "C" int gpca_get_scores_f64(gpca_handle* h, double* out) {
    NEED_RSVD("gpca_get_scores_f64");
    ON_CHILD(gpca_get_scores_f64(h->child, out));
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
    ON_CHILD(gpca_get_loadings(h->child, out));
    if (!h->loadings_valid) return fail(h, GPCA_ERR_STATE, "gpca_get_loadings: the last call (gpca_rsvd_condensed) produced sample scores only");
    if (h->n_pca) HIPCHK(hipMemcpy(out, h->d_load32, (size_t)h->n_pca * h->k * 4, hipMemcpyDeviceToHost));
    return GPCA_OK;
}

// PCA::transform (main.rs:659): scores = A^T U on the resident (or streamed) matrix, U = loadings.
extern "C" int gpca_transform(gpca_handle* h, double* out) {
    NEED_RSVD("gpca_transform");
    ON_CHILD(gpca_transform(h->child, out));
    if (!h->loadings_valid) return fail(h, GPCA_ERR_STATE, "gpca_transform: the last call (gpca_rsvd_condensed) produced sample scores only");
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
    const size_t nY = (size_t)h->N * L;
    if (mr) {
        // the agreement rides the one exchange of the call: 16 status slots behind the N x L block (no round trip of its own)
        if (h->cap_Y < nY + 16) return fail(h, GPCA_ERR_STATE, "gpca_transform: workspace of the last gpca_rsvd is gone");
        h->status_own = h->err;
        status_histogram(h->h_status, lrc);
        if (hipMemcpyAsync(h->dY + nY, h->h_status, 16 * sizeof(double), hipMemcpyHostToDevice, h->st) != hipSuccess && lrc == GPCA_OK)
            lrc = fail(h, GPCA_ERR_HIP, "gpca_transform: status copy failed");
    }
    { const int xrc = allreduce_f64(h, h->dY, (int64_t)nY + (mr ? 16 : 0)); if (xrc != GPCA_OK) return xrc; }
    if (mr) {
        const int own_rc = lrc;
        if (hipMemcpyAsync(h->h_status + 16, h->dY + nY, 16 * sizeof(double), hipMemcpyDeviceToHost, h->st) != hipSuccess || stream_wait(h) != hipSuccess)
            return own_rc != GPCA_OK ? own_rc : fail(h, GPCA_ERR_HIP, "gpca_transform: status fetch failed");
        lrc = status_verdict(h, h->h_status + 16, own_rc, h->status_own, "gpca_transform");
        if (lrc == GPCA_OK) lrc = own_rc;
    }
    if (lrc != GPCA_OK) return lrc;
#undef LOCAL
    // only the k columns asked for leave the device (the whole N x L block was 256 MB at N = 500k, L = 64): compacted on the device
    // by a right-multiplication with the L x k selection matrix, then one contiguous copy into the caller's buffer
    CHK(ensure(h, h->d_tr64, h->cap_tr64, (size_t)h->N * k));
    double* Zsel = h->h_pin + (size_t)kMaxSketch * kMaxSketch;
    for (int j = 0; j < L; ++j) for (int c = 0; c < k; ++c) Zsel[(size_t)j * k + c] = j == c ? 1.0 : 0.0;
    HIPCHK(hipMemcpyAsync(h->dZ, Zsel, sizeof(double) * (size_t)L * k, hipMemcpyHostToDevice, h->st));
    launch_rightmul_f64(h->st, h->dY, h->N, L, h->dZ, k, h->d_tr64, (float*)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->d_tr64, (size_t)h->N * k * 8, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    h->have_rsvd = true;  // dT (=B) is consumed, but scores/loadings/eigenvalues stay valid
    return GPCA_OK;
}


// ---- EigenSNP stages (efficient_pca's EigenSNPCoreAlgorithm, un-vendored: the stage structure of the published
//      algorithm, parity unpinned -- DESIGN.md 7c) ---------------------------------------------------------------------------------
// Basis learning on a sample subset (main.rs:314-316 subset_factor / min / max_subset_size_for_local_basis_learning): rows of the
// N x l sketch that belong to samples outside the subset are zeroed before every orthonormalisation, so T = A Q and everything
// learnt from it sees the subset's columns only; gpca_transform still projects ALL samples.  mask = NULL: all samples again.
extern "C" int gpca_set_sample_mask(gpca_handle* h, const uint8_t* mask) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!have_genotypes(h)) return fail(h, GPCA_ERR_STATE, "gpca_set_sample_mask: no genotypes resident and no panel stream open");
    HIPCHK(hipStreamSynchronize(h->st));
    if (!mask) { dfree(h->d_smask); h->n_smask = 0; return GPCA_OK; }
    int64_t n_in = 0;
    for (int64_t n = 0; n < h->N; ++n) n_in += mask[n] != 0;
    if (n_in < 2) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_sample_mask: fewer than 2 samples in the subset");
    if (!h->d_smask) HIPCHK(hipMalloc((void**)&h->d_smask, (size_t)h->N));
    HIPCHK(hipMemcpy(h->d_smask, mask, (size_t)h->N, hipMemcpyHostToDevice));
    h->n_smask = n_in;
    h->have_rsvd = false;
    return GPCA_OK;
}

// The block-diagonal condensed basis W = U_blk Lambda^-1: W[i][0..cmax) are SNP row i's coefficients on the condensed features
// [feat0[i], feat0[i] + cmax) of its LD block (zero-padded when the block has fewer local components; feat0[i] < 0: the SNP is in no
// block); R = total number of condensed features.
extern "C" int gpca_set_condensed_basis(gpca_handle* h, const float* W, const int32_t* feat0, int32_t cmax, int64_t R) {
    if (!h || !W || !feat0 || cmax < 1 || cmax > 64 || R < 1) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_condensed_basis: bad arguments (1 <= cmax <= 64, R >= 1)");
    LOCK(h);
    if (h->sm.on || (!h->dG && !h->dG2)) return fail(h, GPCA_ERR_STATE, "gpca_set_condensed_basis: needs a resident matrix");
    if (multi_rank(h)) return fail(h, GPCA_ERR_STATE, "gpca_set_condensed_basis: not available on row-sharded handles");
    const int64_t M = h->M;
    // blocks = distinct feat0 values; each block's row range [first, last + 1) bounds the reduction over its SNPs
    std::map<int32_t, std::pair<int64_t, int64_t>> range;
    for (int64_t i = 0; i < M; ++i) {
        const int32_t f = feat0[i];
        if (f < 0) continue;
        if ((int64_t)f + cmax > R + 64 || f >= R) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_condensed_basis: feat0 out of range");
        auto it = range.find(f);
        if (it == range.end()) range[f] = {i, i + 1}; else it->second.second = i + 1;
    }
    if (range.empty()) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_condensed_basis: no SNP belongs to a block");
    if (R > M) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_condensed_basis: more condensed features than SNP rows");
    std::vector<int64_t> r0, r1; std::vector<int32_t> bf, bc;
    for (const auto& kv : range) { bf.push_back(kv.first); r0.push_back(kv.second.first); r1.push_back(kv.second.second); }
    // a block's feature count = distance to the next block's first feature (the map is ordered by feat0), at most cmax: features
    // bf + c_b .. belong to the next block and must not be written by this one
    for (size_t b = 0; b < bf.size(); ++b) {
        const int64_t next = b + 1 < bf.size() ? (int64_t)bf[b + 1] : R;
        const int64_t cb = next - bf[b];
        if (cb < 1) return fail(h, GPCA_ERR_BAD_ARG, "gpca_set_condensed_basis: feat0 values of different blocks overlap");
        bc.push_back((int32_t)std::min<int64_t>(cb, cmax));
    }
    HIPCHK(hipStreamSynchronize(h->st));
    dfree(h->d_cw); dfree(h->d_cfeat0); dfree(h->d_cblk_row0); dfree(h->d_cblk_row1); dfree(h->d_cblk_feat0); dfree(h->d_cblk_c);
    const size_t B = bf.size();
    HIPCHK(hipMalloc((void**)&h->d_cw, (size_t)M * cmax * 4)); HIPCHK(hipMalloc((void**)&h->d_cfeat0, (size_t)M * 4));
    HIPCHK(hipMalloc((void**)&h->d_cblk_row0, B * 8)); HIPCHK(hipMalloc((void**)&h->d_cblk_row1, B * 8)); HIPCHK(hipMalloc((void**)&h->d_cblk_feat0, B * 4));
    HIPCHK(hipMalloc((void**)&h->d_cblk_c, B * 4));
    HIPCHK(hipMemcpy(h->d_cblk_c, bc.data(), B * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_cw, W, (size_t)M * cmax * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_cfeat0, feat0, (size_t)M * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_cblk_row0, r0.data(), B * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_cblk_row1, r1.data(), B * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_cblk_feat0, bf.data(), B * 4, hipMemcpyHostToDevice));
    h->c_cmax = cmax; h->c_R = R; h->c_B = (int)B;
    return GPCA_OK;
}

// T' = r o X, c = b^T X for an arbitrary M x L f32 factor X sitting in dT (what gpca_transform does for the loadings)
static int prep_custom_T(gpca_handle* h) {
    const int L = h->L;
    if (h->precision == GPCA_PREC_I8_EXACT) launch_scale_rows(h->st, h->dT, h->M, h->Mpad, L, h->d_r, h->d_b, h->dT, h->d_cpart, 0);
    else launch_scale_rows(h->st, h->dT, h->M, h->Mpad, L, h->d_r, h->d_b, h->dTb, h->d_cpart);
    HIPCHK(hipGetLastError());
    h->apart_valid = false;
    return stage_sum_c(h, omega_num_parts(h->Mpad));
}

// Initial global PCs from the row-standardised condensed features C* = W^T X (never formed): randomized PCA of C* with its
// products factored through the genotype GEMMs -- C*^T Z = A^T (W Z), C* Q = W^T (A Q).  Leaves the N x k sample scores
// (gpca_get_scores / _f64) and the eigenvalues of C*; loadings are not defined for this call.
extern "C" int gpca_rsvd_condensed(gpca_handle* h, int32_t k, int32_t oversample, int32_t power_iters, uint64_t seed) {
    if (!h) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (!h->d_cw) return fail(h, GPCA_ERR_STATE, "gpca_rsvd_condensed: call gpca_set_condensed_basis first");
    if (h->sm.on || multi_rank(h)) return fail(h, GPCA_ERR_STATE, "gpca_rsvd_condensed: needs a resident, unsharded matrix");
    if ((int64_t)k + oversample > h->c_R) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd_condensed: k + oversample exceeds the number of condensed features");
    if ((int64_t)k + oversample > 64) return fail(h, GPCA_ERR_BAD_ARG, "gpca_rsvd_condensed: k + oversample must be <= 64 (the EigenSNP stages hold 64 sketch columns)");
    CHK(rsvd_preflight(h, k, oversample, power_iters));
    const int l = h->l, L = h->L, cmax = h->c_cmax;
    const int64_t R = h->c_R, Rpad = round_up(R, 128);
    CHK(ensure(h, h->dP, h->cap_P, (size_t)(Rpad + 64) * L));
    CHK(ensure(h, h->d_ones, h->cap_ones, (size_t)Rpad)); CHK(ensure(h, h->d_zeros, h->cap_zeros, (size_t)Rpad));
    CHK(ensure(h, h->d_lqr, h->cap_lqr, (size_t)h->Mpad * L));      // (f32 staging of the condensed sketch; R <= M)
    launch_fill_f32(h->st, h->d_ones, Rpad, 1.0f); launch_fill_f32(h->st, h->d_zeros, Rpad, 0.0f);
    HIPCHK(hipMemsetAsync(h->dP, 0, (size_t)(Rpad + 64) * L * 8, h->st));
    // 1. sketch of C*: Omega_R (R x l, the engine's Philox stream over feature indices), Y = C*^T Omega = A^T (W Omega)
    launch_omega(h->st, R, Rpad, l, L, 0, seed, h->d_ones, h->d_zeros, h->d_lqr, h->d_cpart, nullptr, 0);
    HIPCHK(hipGetLastError());
    launch_f32_to_f64(h->st, h->d_lqr, h->dP, R * (int64_t)L);
    HIPCHK(hipGetLastError());
    auto through_W_back = [&]() -> int {            // dT = W dP ; T' = r o dT, c ; Y = A^T T'
        launch_bd_expand(h->st, h->d_cw, h->d_cfeat0, cmax, h->dP, h->M, L, h->dT);
        HIPCHK(hipGetLastError());
        CHK(prep_custom_T(h));
        return stage_AtT_local(h);
    };
    auto through_W_forward = [&]() -> int {         // dT = A Q ; dP = W^T dT
        CHK(stage_AQ(h, 0));
        launch_bd_reduce(h->st, h->d_cw, h->d_cfeat0, cmax, h->dT, L, h->d_cblk_row0, h->d_cblk_row1, h->d_cblk_feat0, h->d_cblk_c, h->c_B, h->dP);
        HIPCHK(hipGetLastError());
        return GPCA_OK;
    };
    CHK(through_W_back());
    CHK(stage_orth(h));
    for (int it = 0; it < power_iters; ++it) { CHK(through_W_forward()); CHK(through_W_back()); CHK(stage_orth(h)); }
    // 2. projection P = C* Q (R x l) and the l x l eigenproblem of P^T P
    CHK(through_W_forward());
    launch_gram_f64(h->st, h->dP, R, L, h->d_part64);
    HIPCHK(hipGetLastError());
    const double* gsrc = nullptr; int gslices = 0;
    launch_sum_partials_f64_stage1(h->st, h->d_part64, gram_num_parts(R), (int64_t)L * L, h->d_scratch64, &gsrc, &gslices);
    HIPCHK(hipGetLastError());
    CHK(enqueue_small_eigh(h, gsrc, gslices, 0, (double)(h->N - 1)));      // scores = Q V diag(s), eigenvalues of C*
    CHK(enqueue_scores_loadings(h, nullptr, false));
    CHK(finish_small_eigh(h, true));
    h->have_rsvd = true; h->loadings_valid = false; h->rsvd_on_child = false;
    return GPCA_OK;
}

// One refinement pass from given sample scores S0 (N x k, any basis of the current estimate of the PC subspace):
//   L = orth(A S0) (SNP side, CholeskyQR2 over the M rows), S = A^T L (N x k), S^T S = W Sigma^2 W^T,
//   scores = S W, loadings = L W, eigenvalues = Sigma^2 / (N - 1).
// Results through the usual getters.  (compute_refined_snp_loadings / compute_rotated_final_outputs of the published algorithm.)
extern "C" int gpca_refine(gpca_handle* h, const double* S0, int32_t k) {
    if (!h || !S0) return GPCA_ERR_BAD_ARG;
    LOCK(h);
    if (multi_rank(h)) return fail(h, GPCA_ERR_STATE, "gpca_refine: not available on row-sharded handles");
    const uint8_t* saved_mask = h->d_smask;
    if (saved_mask) return fail(h, GPCA_ERR_STATE, "gpca_refine: clear the sample mask first (refinement uses every sample)");
    if (k > 64) return fail(h, GPCA_ERR_BAD_ARG, "gpca_refine: k must be <= 64 (the EigenSNP stages hold 64 sketch columns)");
    CHK(rsvd_preflight(h, k, 0, 0));
    const int l = h->l, L = h->L;
    CHK(ensure(h, h->d_lqr, h->cap_lqr, (size_t)h->Mpad * L));
    // Q = orth(S0) (a basis of the same subspace; keeps the digit scales of the exact path well conditioned)
    {
        std::vector<double> Y((size_t)h->N * L, 0.0);
        for (int64_t n = 0; n < h->N; ++n) for (int c = 0; c < k; ++c) Y[(size_t)n * L + c] = S0[(size_t)n * k + c];
        HIPCHK(hipMemcpyAsync(h->dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
    }
    CHK(stage_orth(h));
    CHK(stage_AQ(h, 0));                                  // dT = A Q   (M x L f32)
    for (int round = 0; round < 2; ++round) {             // CholeskyQR2 over the SNP rows
        launch_gram_f32(h->st, h->dT, h->M, L, h->d_part64);
        HIPCHK(hipGetLastError());
        launch_sum_partials_f64(h->st, h->d_part64, gram_num_parts(h->M), (int64_t)L * L, h->dW, h->d_scratch64);
        HIPCHK(hipGetLastError());
        launch_chol_inv(h->st, h->dW, l, L, h->dZ, h->d_cholflag);
        HIPCHK(hipGetLastError());
        launch_rightmul_inplace_f32(h->st, h->dT, h->M, L, h->dZ);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(h->d_lqr, h->dT, (size_t)h->Mpad * L * 4, hipMemcpyDeviceToDevice, h->st));   // L, kept for the loadings
    CHK(prep_custom_T(h));
    CHK(stage_AtT_local(h));                              // dY = S = A^T L   (N x L f64)
    launch_gram_f64(h->st, h->dY, h->N, L, h->d_part64);
    HIPCHK(hipGetLastError());
    const double* gsrc = nullptr; int gslices = 0;
    launch_sum_partials_f64_stage1(h->st, h->d_part64, gram_num_parts(h->N), (int64_t)L * L, h->d_scratch64, &gsrc, &gslices);
    HIPCHK(hipGetLastError());
    CHK(enqueue_small_eigh(h, gsrc, gslices, 1, (double)(h->N - 1)));      // S^T S = W Sigma^2 W^T: scores = S W, loadings = L W
    CHK(enqueue_scores_loadings(h, h->d_lqr, true));
    CHK(finish_small_eigh(h, true));
    h->have_rsvd = true; h->loadings_valid = true; h->rsvd_on_child = false;
    return GPCA_OK;
}
