/* gpca.hpp -- C++17 host mirror of the reference's operator interface over the C ABI of libgpca.so (gpca.h).
 *
 * Header only; nothing here touches the device except through gpca.h.  The names, argument meaning and error behaviour
 * follow the reference's Rust types at the L2 boundary, so that a host written against them reads the same:
 *
 *   MicroarrayGenotypeAccessor  : impl PcaReadyGenotypeAccessor      /root/reference/src/prepare.rs:1770-1779, 1838-2030
 *   PcaSnpId / QcSampleId       : dense 0-based ids                   /root/reference/src/prepare.rs:1485, 1854, 1858
 *   LdBlockSpecification        :                                     /root/reference/src/prepare.rs:1540-1543
 *   EigenSNPCoreAlgorithmConfig : the 14 fields                       /root/reference/src/main.rs:311-327
 *   EigenSNPCoreAlgorithm       : ::new(cfg).compute_pca(&acc, &blk)  /root/reference/src/main.rs:359-366
 *   PCA                         : ::new / rfit / transform            /root/reference/src/main.rs:602, 648-660
 *
 * Errors: every failed C call becomes a gpca::Error carrying the integer status and the handle's message (the reference
 * returns Result<_, Box<dyn Error + Send + Sync>> with a human string, prepare.rs:61, 1879-1881).
 */
#ifndef GPCA_HPP
#define GPCA_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gpca.h"

namespace gpca {

class Error : public std::runtime_error {
public:
    Error(int status, const std::string& msg) : std::runtime_error("[gpca status " + std::to_string(status) + "] " + msg), status_(status) {}
    int status() const { return status_; }
private:
    int status_;
};

using PcaSnpId = int64_t;     // index into the post-QC SNP list   (prepare.rs:1485)
using QcSampleId = int64_t;   // index into the QC'd sample list   (prepare.rs:1854)

struct QcConfig {             // clap's effective defaults with --eigensnp (main.rs:545-560)
    double min_snp_call_rate_threshold = 0.98;
    double min_snp_maf_threshold = 0.01;
    double max_snp_hwe_p_value_threshold = 1e-6;
    static QcConfig none() { return QcConfig{0.0, 0.0, 1.0}; }
};

struct SnpStats { std::vector<float> mu, sigma; std::vector<uint8_t> keep; };

/* One opaque gpca_handle: one GPU, one SNP-row shard of the genotype matrix. */
class Engine {
public:
    explicit Engine(int device = -1, int precision = GPCA_PREC_I8_EXACT, int storage = GPCA_STORE_INT8, int digit_planes = 0)
        : device_(device), precision_(precision), storage_(storage), digit_planes_(digit_planes) {
        gpca_config cfg{};
        cfg.device = device; cfg.precision = precision; cfg.storage = storage; cfg.digit_planes = digit_planes;
        const int rc = gpca_create(&cfg, &h_);
        if (rc != GPCA_OK) throw Error(rc, gpca_last_error(nullptr));
    }
    ~Engine() { if (h_) gpca_destroy(h_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    Engine(Engine&& o) noexcept : h_(o.h_), k_(o.k_), device_(o.device_), precision_(o.precision_), storage_(o.storage_), digit_planes_(o.digit_planes_) { o.h_ = nullptr; }

    gpca_handle* handle() const { return h_; }
    void check(int rc) const { if (rc != GPCA_OK) throw Error(rc, gpca_last_error(h_)); }

    /* residency */
    void upload_genotypes_i8(const int8_t* snp_major, int64_t M, int64_t N, int64_t ld) { check(gpca_upload_genotypes_i8(h_, snp_major, M, N, ld)); }
    void upload_bed2bit(const uint8_t* bed_rows, int64_t M, int64_t N) { check(gpca_upload_bed2bit(h_, bed_rows, M, N)); }
    void load_from_source(const gpca_panel_source& src, int64_t M, int64_t N) { check(gpca_load_from_source(h_, &src, M, N)); }
    /* out of core: see gpca.h (gpca_stream_open / _set_fused / _set_cache) */
    void stream_open(const gpca_panel_source& src, int64_t M, int64_t N, int64_t panel_rows = 0, int ring_slots = 3, bool fused = true,
                     int64_t cache_bytes = 0) {
        check(gpca_stream_open(h_, &src, M, N, panel_rows, ring_slots));
        check(gpca_stream_set_fused(h_, fused ? 1 : 0));
        if (cache_bytes != 0) check(gpca_stream_set_cache(h_, cache_bytes, nullptr));
    }
    std::pair<int64_t, int64_t> device_memory() const { int64_t f = 0, t = 0; check(gpca_get_device_memory(h_, &f, &t)); return {f, t}; }   // (free, total) bytes
    std::pair<int64_t, int64_t> dims() const { int64_t M = 0, N = 0; check(gpca_dims(h_, &M, &N)); return {M, N}; }

    /* a1/a3 */
    SnpStats snp_stats(const QcConfig& qc, bool fetch = true) {
        const gpca_qc_config c{qc.min_snp_call_rate_threshold, qc.min_snp_maf_threshold, qc.max_snp_hwe_p_value_threshold};
        SnpStats st;
        if (!fetch) { check(gpca_snp_stats(h_, &c, nullptr, nullptr, nullptr)); return st; }
        const int64_t M = dims().first;
        st.mu.resize((size_t)M); st.sigma.resize((size_t)M); st.keep.resize((size_t)M);
        check(gpca_snp_stats(h_, &c, st.mu.data(), st.sigma.data(), st.keep.data()));
        return st;
    }
    SnpStats get_standardization() const {
        SnpStats st; const int64_t M = dims().first;
        st.mu.resize((size_t)M); st.sigma.resize((size_t)M); st.keep.resize((size_t)M);
        check(gpca_get_standardization(h_, st.mu.data(), st.sigma.data(), st.keep.data()));
        return st;
    }
    void set_standardization(const std::vector<float>& mu, const std::vector<float>& sigma, const std::vector<uint8_t>& keep) {
        const size_t M = (size_t)dims().first;
        if (mu.size() != M || sigma.size() != M || keep.size() != M) throw std::invalid_argument("set_standardization: need one entry per SNP row");
        check(gpca_set_standardization(h_, mu.data(), sigma.data(), keep.data()));
    }
    int64_t num_pca_snps() const { return gpca_num_pca_snps(h_); }
    int64_t num_qc_samples() const { return gpca_num_qc_samples(h_); }
    std::vector<int64_t> pca_snp_rows() const {
        std::vector<int64_t> r((size_t)std::max<int64_t>(num_pca_snps(), 0));
        if (!r.empty()) check(gpca_get_pca_snp_rows(h_, r.data()));
        return r;
    }

    /* a2: the pull API (prepare.rs:1839-2022); out is [snps x samples], C order */
    std::vector<float> standardize_block(const std::vector<PcaSnpId>& snps, const std::vector<QcSampleId>& samples) const {
        std::vector<float> out(snps.size() * samples.size());
        check(gpca_standardize_block(h_, snps.data(), (int64_t)snps.size(), samples.data(), (int64_t)samples.size(), out.data()));
        return out;
    }

    /* f3: the stages of EigenSNPCoreAlgorithm (gpca.h) */
    int device() const { return device_; }
    int precision() const { return precision_; }
    int storage() const { return storage_; }
    int digit_planes() const { return digit_planes_; }
    void copy_rows_from(const Engine& src, int64_t row0, int64_t rows) { check(gpca_copy_rows(h_, src.h_, row0, rows)); }
    void set_sample_mask(const std::vector<uint8_t>* mask) {
        if (mask && (int64_t)mask->size() != dims().second) throw std::invalid_argument("set_sample_mask: one entry per sample");
        check(gpca_set_sample_mask(h_, mask ? mask->data() : nullptr));
    }
    void set_condensed_basis(const std::vector<float>& W, const std::vector<int32_t>& feat0, int cmax, int64_t R) {
        const size_t M = (size_t)dims().first;
        if (feat0.size() != M || W.size() != M * (size_t)cmax) throw std::invalid_argument("set_condensed_basis: W is [SNPs][cmax], feat0 [SNPs]");
        check(gpca_set_condensed_basis(h_, W.data(), feat0.data(), cmax, R));
    }
    void rsvd_condensed(int k, int oversample, int power_iters, uint64_t seed) { check(gpca_rsvd_condensed(h_, k, oversample, power_iters, seed)); k_ = k; }
    void refine(const std::vector<double>& scores, int k) {
        if ((int64_t)scores.size() != dims().second * (int64_t)k) throw std::invalid_argument("refine: scores are [samples][k]");
        check(gpca_refine(h_, scores.data(), k)); k_ = k;
    }

    /* a5/a6 */
    void rsvd(int k, int oversample, int power_iters, uint64_t seed) { check(gpca_rsvd(h_, k, oversample, power_iters, seed)); k_ = k; }
    int components() const { return k_; }
    std::vector<float> scores() const { std::vector<float> s((size_t)num_qc_samples() * (size_t)k_); check(gpca_get_scores(h_, s.data())); return s; }
    std::vector<double> eigenvalues() const { std::vector<double> e((size_t)k_); check(gpca_get_eigenvalues(h_, e.data())); return e; }
    std::vector<float> loadings() const { std::vector<float> l((size_t)num_pca_snps() * (size_t)k_); check(gpca_get_loadings(h_, l.data())); return l; }
    std::vector<double> scores_f64() const { std::vector<double> s((size_t)num_qc_samples() * (size_t)k_); check(gpca_get_scores_f64(h_, s.data())); return s; }
    std::vector<double> transform() const { std::vector<double> s((size_t)num_qc_samples() * (size_t)k_); check(gpca_transform(h_, s.data())); return s; }

private:
    gpca_handle* h_ = nullptr;
    int k_ = 0;
    int device_ = -1, precision_ = GPCA_PREC_I8_EXACT, storage_ = GPCA_STORE_INT8, digit_planes_ = 0;
};

struct LdBlockSpecification {                       // prepare.rs:1540-1543
    std::string user_defined_block_tag;
    std::vector<PcaSnpId> pca_snp_ids_in_block;
};

/* impl PcaReadyGenotypeAccessor for MicroarrayGenotypeAccessor (prepare.rs:1838-2030), backed by genotypes resident in
 * HBM (or streamed panels) instead of the IoService actor pool.  Copyable like the reference's `Clone` accessor: copies
 * share the engine: pulls from different threads run concurrently on the handle's lanes, every other call waits for them (gpca.h,
 * "Threading"). */
class MicroarrayGenotypeAccessor {
public:
    explicit MicroarrayGenotypeAccessor(Engine& e) : eng_(&e) {}
    std::vector<float> get_standardized_snp_sample_block(const std::vector<PcaSnpId>& pca_snp_ids_to_fetch,
                                                         const std::vector<QcSampleId>& qc_sample_ids_to_fetch) const {
        return eng_->standardize_block(pca_snp_ids_to_fetch, qc_sample_ids_to_fetch);
    }
    int64_t num_pca_snps() const { return eng_->num_pca_snps(); }
    int64_t num_qc_samples() const { return eng_->num_qc_samples(); }
    std::vector<int64_t> original_indices_of_pca_snps() const { return eng_->pca_snp_rows(); }
    Engine& engine() const { return *eng_; }
private:
    Engine* eng_;
};

struct EigenSNPCoreAlgorithmConfig {                // main.rs:311-327 with clap's effective defaults (main.rs:561-588)
    int target_num_global_pcs = 10;
    int components_per_ld_block = 7;
    double subset_factor_for_local_basis_learning = 0.075;
    int64_t min_subset_size_for_local_basis_learning = 10000;
    int64_t max_subset_size_for_local_basis_learning = 40000;
    int global_pca_sketch_oversampling = 10;
    int global_pca_num_power_iterations = 2;
    int local_rsvd_sketch_oversampling = 10;
    int local_rsvd_num_power_iterations = 2;
    uint64_t random_seed = 2025;
    int64_t snp_processing_strip_size = 2000;
    int refine_pass_count = 1;
    bool collect_diagnostics = false;
    int64_t diagnostic_block_list_id_to_trace = -1;
};

struct EigenSNPCoreOutput {
    std::vector<float> final_sample_principal_component_scores;    // [N][K]  main.rs:389
    std::vector<double> final_principal_component_eigenvalues;     // [K]     main.rs:394
    std::vector<float> final_snp_principal_component_loadings;     // [D][K]  main.rs:407
    int64_t num_qc_samples_used = 0, num_pca_snps_used = 0;
    int num_principal_components_computed = 0;
};

/* EigenSNPCoreAlgorithm::new(cfg).compute_pca(&accessor, &blocks) (main.rs:359-365): top-K PCA of the standardised matrix
 * restricted to the SNPs the blocks name (a PCA SNP in no block leaves the PCA, prepare.rs:1465-1469).
 *   local_stage = false (default): ONE global randomized PCA over the union of the blocks (target_num_global_pcs,
 *     global_pca_sketch_oversampling, global_pca_num_power_iterations, random_seed act).
 *   local_stage = true: the multi-stage algorithm the 14 config fields parameterise, as published for the un-vendored
 *     efficient_pca crate (parity UNPINNED): sample subset -> per-block local bases -> condensed features of all samples, row
 *     standardised -> initial global randomized PCA of the condensed features -> refine_pass_count refinement passes on the full
 *     matrix.  Same stages, same calls, same seeds as genomic_pca_amd.engine.EigenSNPCoreAlgorithm._multi_stage. */
class EigenSNPCoreAlgorithm {
public:
    explicit EigenSNPCoreAlgorithm(const EigenSNPCoreAlgorithmConfig& cfg) : cfg_(cfg) {}

    static int64_t subset_size(const EigenSNPCoreAlgorithmConfig& cfg, int64_t n_samples) {
        int64_t want = (int64_t)std::nearbyint(cfg.subset_factor_for_local_basis_learning * (double)n_samples);
        want = std::max(cfg.min_subset_size_for_local_basis_learning, std::min(cfg.max_subset_size_for_local_basis_learning, want));
        return std::max<int64_t>(2, std::min(n_samples, want));
    }
    /* the first ns entries of a Fisher-Yates shuffle driven by SplitMix64(seed); empty = every sample */
    static std::vector<uint8_t> subset_mask(const EigenSNPCoreAlgorithmConfig& cfg, int64_t n_samples) {
        const int64_t ns = subset_size(cfg, n_samples);
        if (ns >= n_samples) return {};
        std::vector<int64_t> idx((size_t)n_samples);
        for (int64_t i = 0; i < n_samples; ++i) idx[(size_t)i] = i;
        uint64_t state = cfg.random_seed;
        for (int64_t i = 0; i < ns; ++i) {
            state += 0x9E3779B97F4A7C15ull;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            const int64_t j = i + (int64_t)(z % (uint64_t)(n_samples - i));
            std::swap(idx[(size_t)i], idx[(size_t)j]);
        }
        std::vector<uint8_t> mask((size_t)n_samples, 0);
        for (int64_t i = 0; i < ns; ++i) mask[(size_t)idx[(size_t)i]] = 1;
        return mask;
    }

    EigenSNPCoreOutput compute_pca(const MicroarrayGenotypeAccessor& accessor, const std::vector<LdBlockSpecification>& ld_blocks,
                                   bool local_stage = false) const {
        Engine& eng = accessor.engine();
        const int64_t n_pca = accessor.num_pca_snps();
        if (ld_blocks.empty()) throw std::invalid_argument("compute_pca: ld_block_specifications is empty");
        std::vector<PcaSnpId> ids;
        for (const auto& b : ld_blocks) ids.insert(ids.end(), b.pca_snp_ids_in_block.begin(), b.pca_snp_ids_in_block.end());
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        if (ids.empty()) throw std::invalid_argument("compute_pca: the LD blocks hold no PCA SNP");
        if (ids.front() < 0 || ids.back() >= n_pca)
            throw std::invalid_argument("compute_pca: PcaSnpId " + std::to_string(ids.front() < 0 ? ids.front() : ids.back()) + " out of range [0, " +
                                        std::to_string(n_pca) + ")");
        SnpStats saved;
        std::vector<int64_t> rows;
        const bool restrict_rows = (int64_t)ids.size() < n_pca;
        if (restrict_rows || local_stage) { saved = eng.get_standardization(); rows = eng.pca_snp_rows(); }
        if (restrict_rows) {   // keep mask = union of the blocks (same mu / sigma); the accessor's numbering is restored afterwards
            std::vector<uint8_t> keep2(saved.keep.size(), 0);
            for (PcaSnpId id : ids) keep2[(size_t)rows[(size_t)id]] = 1;
            eng.set_standardization(saved.mu, saved.sigma, keep2);
        }
        EigenSNPCoreOutput out;
        try {
            if (local_stage) multi_stage(eng, ld_blocks, saved, rows);
            else eng.rsvd(cfg_.target_num_global_pcs, cfg_.global_pca_sketch_oversampling, cfg_.global_pca_num_power_iterations, cfg_.random_seed);
            out.final_sample_principal_component_scores = eng.scores();
            out.final_principal_component_eigenvalues = eng.eigenvalues();
            out.final_snp_principal_component_loadings = eng.loadings();
            out.num_qc_samples_used = accessor.num_qc_samples();
            out.num_pca_snps_used = (int64_t)ids.size();
            out.num_principal_components_computed = eng.components();   // (the local stage may leave fewer: min(K, condensed features))
        } catch (...) {
            if (restrict_rows) eng.set_standardization(saved.mu, saved.sigma, saved.keep);
            throw;
        }
        if (restrict_rows) eng.set_standardization(saved.mu, saved.sigma, saved.keep);
        return out;
    }

private:
    void multi_stage(Engine& eng, const std::vector<LdBlockSpecification>& ld_blocks, const SnpStats& st, const std::vector<int64_t>& pca_rows) const {
        const auto dm = eng.dims();
        const int64_t M = dm.first, N = dm.second;
        const int K = cfg_.target_num_global_pcs;
        const std::vector<uint8_t> mask = subset_mask(cfg_, N);
        int64_t n_sub = N;
        if (!mask.empty()) { n_sub = 0; for (uint8_t m : mask) n_sub += m; }
        std::vector<std::vector<int64_t>> blocks;
        int64_t longest = 0;
        for (const auto& b : ld_blocks) {
            if (b.pca_snp_ids_in_block.empty()) continue;
            std::vector<int64_t> r;
            for (PcaSnpId id : b.pca_snp_ids_in_block) r.push_back(pca_rows[(size_t)id]);
            std::sort(r.begin(), r.end());
            longest = std::max<int64_t>(longest, (int64_t)r.size());
            blocks.push_back(std::move(r));
        }
        const int64_t cp = std::max(1, cfg_.components_per_ld_block);
        const int cmax = (int)std::min<int64_t>({cp, longest, n_sub});
        std::vector<float> W((size_t)M * (size_t)cmax, 0.f);
        std::vector<int32_t> feat0((size_t)M, -1);
        int64_t R = 0;
        {
            Engine sub(eng.device(), eng.precision(), eng.storage(), eng.digit_planes());
            {   // sized once for the widest block: every block reuses its buffers
                size_t big = 0;
                for (size_t bi = 1; bi < blocks.size(); ++bi)
                    if (blocks[bi].back() - blocks[bi].front() > blocks[big].back() - blocks[big].front()) big = bi;
                sub.copy_rows_from(eng, blocks[big].front(), blocks[big].back() + 1 - blocks[big].front());
            }
            for (size_t bi = 0; bi < blocks.size(); ++bi) {
                const std::vector<int64_t>& rows = blocks[bi];
                const int64_t r0 = rows.front(), r1 = rows.back() + 1, D = (int64_t)rows.size();
                sub.copy_rows_from(eng, r0, r1 - r0);
                std::vector<uint8_t> keep((size_t)(r1 - r0), 0);
                for (int64_t r : rows) keep[(size_t)(r - r0)] = 1;
                sub.set_standardization(std::vector<float>(st.mu.begin() + r0, st.mu.begin() + r1), std::vector<float>(st.sigma.begin() + r0, st.sigma.begin() + r1), keep);
                sub.set_sample_mask(mask.empty() ? nullptr : &mask);
                int c = (int)std::min<int64_t>({cp, D, n_sub});
                const int lo = (int)std::max<int64_t>(0, std::min<int64_t>(cfg_.local_rsvd_sketch_oversampling, std::min(D, n_sub) - c));
                sub.rsvd(c, lo, cfg_.local_rsvd_num_power_iterations, cfg_.random_seed + 1 + (uint64_t)bi);
                const std::vector<float> U = sub.loadings();          // [D][c]
                const std::vector<double> feats = sub.transform();    // [N][c]
                int kept_cols = 0;
                for (int j = 0; j < c; ++j) {
                    double mean = 0.0;
                    for (int64_t n = 0; n < N; ++n) mean += feats[(size_t)n * c + j];
                    mean /= (double)N;
                    double ss = 0.0;
                    for (int64_t n = 0; n < N; ++n) { const double d = feats[(size_t)n * c + j] - mean; ss += d * d; }
                    const double sd = std::sqrt(ss / (double)(N - 1));
                    if (!(sd > 1e-12)) continue;
                    const float sdf = (float)sd;
                    for (int64_t a = 0; a < D; ++a) W[(size_t)rows[(size_t)a] * cmax + kept_cols] = U[(size_t)a * c + j] / sdf;
                    ++kept_cols;
                }
                if (kept_cols == 0) continue;
                for (int64_t r : rows) feat0[(size_t)r] = (int32_t)R;
                R += kept_cols;
            }
        }
        if (R < 1) throw std::invalid_argument("compute_pca: no condensed features (every local component is constant)");
        eng.set_sample_mask(nullptr);
        eng.set_condensed_basis(W, feat0, cmax, R);
        const int k0 = (int)std::min<int64_t>(K, R);
        const int go = (int)std::max<int64_t>(0, std::min<int64_t>(cfg_.global_pca_sketch_oversampling, std::min(R, N) - k0));
        eng.rsvd_condensed(k0, go, cfg_.global_pca_num_power_iterations, cfg_.random_seed);
        std::vector<double> scores = eng.scores_f64();
        for (int pass = 0; pass < std::max(1, cfg_.refine_pass_count); ++pass) { eng.refine(scores, k0); scores = eng.scores_f64(); }
    }

    EigenSNPCoreAlgorithmConfig cfg_;
};

/* PCA::new(), .rfit(x, k, n_oversamples, seed, tol), .transform(x) (main.rs:602, 648-660).  The reference hands rfit an
 * Array2<f64> samples x variants (vcf.rs:329-342) by value; here the same matrix goes in as the int8 dosages it was built
 * from, SNP-major (variants x samples): 1 B per genotype instead of build_matrix's 8 B + clone (main.rs:640). */
class PCA {
public:
    explicit PCA(int device = -1, int precision = GPCA_PREC_I8_EXACT, int storage = GPCA_STORE_INT8) : eng_(device, precision, storage) {}
    PCA& rfit(const int8_t* variants_by_samples, int64_t n_features, int64_t n_samples, int k, int n_oversamples = 10, uint64_t seed = 0,
              int power_iters = 2) {
        if (k == 0) throw std::invalid_argument("Number of components (-k) must be > 0.");                                       // main.rs:607-609
        if (n_samples < 2) throw std::invalid_argument("PCA requires at least 2 samples, found " + std::to_string(n_samples) + ".");  // main.rs:614-616
        if (n_features == 0) throw std::invalid_argument("PCA requires at least 1 variant (feature), found 0.");                 // main.rs:617-619
        k = (int)std::min<int64_t>({(int64_t)k, n_samples, n_features});                                                          // main.rs:621-628
        eng_.upload_genotypes_i8(variants_by_samples, n_features, n_samples, n_samples);
        eng_.snp_stats(QcConfig::none(), false);
        // zero-variance rows leave the PCA even without QC thresholds: the reference's clamp (main.rs:621-628), applied to what is left
        if (eng_.num_pca_snps() == 0) throw std::invalid_argument("PCA requires at least 1 variant (feature), found 0.");
        k = (int)std::min<int64_t>((int64_t)k, eng_.num_pca_snps());
        const int64_t l = std::min<int64_t>({(int64_t)k + n_oversamples, n_samples, eng_.num_pca_snps()});
        eng_.rsvd(k, (int)(l - k), power_iters, seed);
        fitted_ = true;
        return *this;
    }
    std::vector<double> transform() const {       // N x k, f64 like the reference's Array2<f64> (main.rs:659)
        if (!fitted_) throw std::logic_error("PCA.transform before rfit");
        return eng_.transform();
    }
    std::vector<double> explained_variance() const { return eng_.eigenvalues(); }
    std::vector<float> rotation() const { return eng_.loadings(); }
    int components() const { return eng_.components(); }
    Engine& engine() { return eng_; }
private:
    Engine eng_;
    bool fitted_ = false;
};

}  // namespace gpca

#endif /* GPCA_HPP */
