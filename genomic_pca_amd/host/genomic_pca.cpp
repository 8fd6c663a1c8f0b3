// genomic_pca -- the reference's command line (main.rs:501-593) over the MI355X engine, as a native host program.
//
// Two workflows, dispatched on --eigensnp like main.rs:109-122:
//   * VCF  (run_vcf_workflow, main.rs:133-247):  --vcf-dir D -k K [--maf f] [--rfit-seed s] --out P
//         -> P.vcf.pca.tsv, P.eigenvalues.tsv (header only, as main.rs:676 leaves the vector empty;
//            --write-eigenvalues is an extension that fills it)
//   * BED  (run_eigensnp_rust_workflow, main.rs:250-442):  --eigensnp --bed-file B --ld-block-file L --out P [--eigensnp-*]
//         -> P.eigensnp.pca.tsv, P.eigenvalues.tsv, P.eigensnp.loadings.tsv
// Everything numerical happens behind include/gpca.h (libgpca.so, hand-written HIP); this file parses text, maps the
// .bed, and writes TSVs.  Same flags, defaults, messages and output bytes as `python -m genomic_pca_amd`
// (genomic_pca_amd/cli.py), which tests/test_cpp_host.py holds it to.
#include <dirent.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "formats.hpp"
#include "gpca.hpp"

namespace {

struct Args {
    std::string output_prefix, vcf_dir, bed_file, ld_block_file, sample_keep_file, log_level = "Info";
    bool have_components = false, have_maf = false, have_seed = false, eigensnp = false, collect_diagnostics = false, write_eigenvalues = false;
    int64_t components = 0, threads = 0;
    double maf = 0.01;
    uint64_t rfit_seed = 0;
    // clap's effective defaults when --eigensnp is given (main.rs:545-588)
    double min_call_rate = 0.98, min_maf = 0.01, max_hwe_p = 1e-6, subset_factor = 0.075;
    int64_t k_global = 10, components_per_block = 7, min_subset = 10000, max_subset = 40000, global_oversampling = 10, global_power_iter = 2,
            local_oversampling = 10, local_power_iter = 2, strip_size = 2000, refine_passes = 1, rfit_power_iters = 2;
    uint64_t seed = 2025;
    // extensions
    int device = -1;
    std::string precision = "i8", storage = "auto", stream = "auto";
    int64_t panel_rows = 0;
    bool local_stage = false;
};

[[noreturn]] void usage_error(const std::string& msg) {
    std::fprintf(stderr, "error: %s\n\nUsage: genomic_pca --out <OUTPUT_PREFIX> (--vcf-dir <DIR> --components <K> | --eigensnp --bed-file <BED> --ld-block-file <FILE>) [options]\n"
                         "For more information, try '--help'.\n", msg.c_str());
    std::exit(2);
}

void print_help() {
    std::puts(
        "Genomic PCA Tool from VCF or BED/LD-block files.\n\n"
        "Usage: genomic_pca [OPTIONS] --out <OUTPUT_PREFIX>\n\n"
        "Options:\n"
        "  -o, --out <OUTPUT_PREFIX>            Output file prefix.\n"
        "  -t, --threads <THREADS>              accepted for compatibility (the GPU does the work)\n"
        "      --log-level <LOG_LEVEL>          [default: Info]\n"
        "  -d, --vcf-dir <VCF_DIR>              Directory containing VCF files (required if not using --eigensnp).\n"
        "  -k, --components <COMPONENTS>        Number of principal components to compute (for VCF workflow); at most 118 here\n"
        "                                       (the sketch holds components + 10 <= 128 columns; the reference has no cap).\n"
        "      --maf <MAF>                      Minimum MAF for VCF variant filtering [default: 0.01 in VCF mode]\n"
        "      --rfit-seed <RFIT_SEED>          Seed for the randomized SVD (VCF workflow).\n"
        "      --eigensnp                       Run PCA on BED + LD block files.\n"
        "      --bed-file <BED_FILE>            Path to the BED file (required if --eigensnp is used).\n"
        "      --ld-block-file <LD_BLOCK_FILE>  Path to the LD block definition file (required if --eigensnp is used).\n"
        "      --eigensnp-sample-keep-file <F>  Optional: file listing sample IDs to keep.\n"
        "      --eigensnp-min-call-rate <X>     [default: 0.98]\n"
        "      --eigensnp-min-maf <X>           [default: 0.01]\n"
        "      --eigensnp-max-hwe-p <X>         (1.0 to disable) [default: 1e-6]\n"
        "      --eigensnp-k-global <K>          [default: 10]\n"
        "      --eigensnp-components-per-block <C>  [default: 7]\n"
        "      --eigensnp-subset-factor <X>     [default: 0.075]\n"
        "      --eigensnp-min-subset-size <N>   [default: 10000]\n"
        "      --eigensnp-max-subset-size <N>   [default: 40000]\n"
        "      --eigensnp-global-oversampling <N>  [default: 10]\n"
        "      --eigensnp-global-power-iter <N> [default: 2]\n"
        "      --eigensnp-local-oversampling <N>   [default: 10]\n"
        "      --eigensnp-local-power-iter <N>  [default: 2]\n"
        "      --eigensnp-seed <SEED>           [default: 2025]\n"
        "      --eigensnp-snp-strip-size <N>    [default: 2000]\n"
        "      --eigensnp-refine-passes <N>     [default: 1]\n"
        "      --eigensnp-collect-diagnostics\n"
        "Extensions:\n"
        "      --device <N>                     HIP device ordinal\n"
        "      --write-eigenvalues              VCF workflow: fill P.eigenvalues.tsv (the reference leaves it header-only)\n"
        "      --gpca-precision <i8|f32>        i8 = exact-integer GEMMs (default); f32 = f32 matrix cores\n"
        "      --gpca-storage <auto|int8|2bit>  HBM residency of the genotypes (auto = 2bit for a .bed of >= 1024 samples, else int8)\n"
        "      --gpca-stream <auto|on|off>      walk the .bed out of core (auto = when it does not fit the device)\n"
        "      --gpca-panel-rows <N>            SNP rows per panel for --gpca-stream (0 = engine default)\n"
        "      --gpca-rfit-power-iters <N>      VCF workflow: power iterations of the randomized PCA [default: 2]\n"
        "      --gpca-eigensnp-local-stage      run the multi-stage algorithm of the --eigensnp-* local / refine flags instead of\n"
        "                                       one global randomized PCA over all blocks (the default)\n"
        "  -h, --help                           Print help");
}

int64_t to_i64(const std::string& flag, const std::string& v) {
    try { size_t n = 0; const long long x = std::stoll(v, &n); if (n != v.size()) throw 1; return x; }
    catch (...) { usage_error("invalid value '" + v + "' for '" + flag + "'"); }
}
uint64_t to_u64(const std::string& flag, const std::string& v) {
    try { size_t n = 0; if (!v.empty() && v[0] == '-') throw 1; const unsigned long long x = std::stoull(v, &n); if (n != v.size()) throw 1; return x; }
    catch (...) { usage_error("invalid value '" + v + "' for '" + flag + "'"); }
}
double to_f64(const std::string& flag, const std::string& v) {
    try { size_t n = 0; const double x = std::stod(v, &n); if (n != v.size()) throw 1; return x; }
    catch (...) { usage_error("invalid value '" + v + "' for '" + flag + "'"); }
}

Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string f = argv[i], inline_val;
        bool has_inline = false;
        if (f.compare(0, 2, "--") == 0) { const size_t eq = f.find('='); if (eq != std::string::npos) { inline_val = f.substr(eq + 1); f = f.substr(0, eq); has_inline = true; } }
        auto val = [&]() -> std::string {
            if (has_inline) return inline_val;
            if (i + 1 >= argc) usage_error("a value is required for '" + f + "' but none was supplied");
            return argv[++i];
        };
        if (f == "-h" || f == "--help") { print_help(); std::exit(0); }
        else if (f == "-o" || f == "--out") a.output_prefix = val();
        else if (f == "-t" || f == "--threads") a.threads = to_i64(f, val());
        else if (f == "--log-level") a.log_level = val();
        else if (f == "-d" || f == "--vcf-dir") a.vcf_dir = val();
        else if (f == "-k" || f == "--components") { a.components = to_i64(f, val()); a.have_components = true; if (a.components < 0) usage_error("invalid value for '--components'"); }
        else if (f == "--maf") { a.maf = to_f64(f, val()); a.have_maf = true; }
        else if (f == "--rfit-seed") { a.rfit_seed = to_u64(f, val()); a.have_seed = true; }
        else if (f == "--eigensnp") a.eigensnp = true;
        else if (f == "--bed-file") a.bed_file = val();
        else if (f == "--ld-block-file") a.ld_block_file = val();
        else if (f == "--eigensnp-sample-keep-file") a.sample_keep_file = val();
        else if (f == "--eigensnp-min-call-rate") a.min_call_rate = to_f64(f, val());
        else if (f == "--eigensnp-min-maf") a.min_maf = to_f64(f, val());
        else if (f == "--eigensnp-max-hwe-p") a.max_hwe_p = to_f64(f, val());
        else if (f == "--eigensnp-k-global") a.k_global = to_i64(f, val());
        else if (f == "--eigensnp-components-per-block") a.components_per_block = to_i64(f, val());
        else if (f == "--eigensnp-subset-factor") a.subset_factor = to_f64(f, val());
        else if (f == "--eigensnp-min-subset-size") a.min_subset = to_i64(f, val());
        else if (f == "--eigensnp-max-subset-size") a.max_subset = to_i64(f, val());
        else if (f == "--eigensnp-global-oversampling") a.global_oversampling = to_i64(f, val());
        else if (f == "--eigensnp-global-power-iter") a.global_power_iter = to_i64(f, val());
        else if (f == "--eigensnp-local-oversampling") a.local_oversampling = to_i64(f, val());
        else if (f == "--eigensnp-local-power-iter") a.local_power_iter = to_i64(f, val());
        else if (f == "--eigensnp-seed") a.seed = to_u64(f, val());
        else if (f == "--eigensnp-snp-strip-size") a.strip_size = to_i64(f, val());
        else if (f == "--eigensnp-refine-passes") a.refine_passes = to_i64(f, val());
        else if (f == "--eigensnp-collect-diagnostics") a.collect_diagnostics = true;
        else if (f == "--device") a.device = (int)to_i64(f, val());
        else if (f == "--write-eigenvalues") a.write_eigenvalues = true;
        else if (f == "--gpca-precision") { a.precision = val(); if (a.precision != "i8" && a.precision != "f32") usage_error("invalid value '" + a.precision + "' for '--gpca-precision' (i8, f32)"); }
        else if (f == "--gpca-storage") { a.storage = val(); if (a.storage != "auto" && a.storage != "int8" && a.storage != "2bit") usage_error("invalid value '" + a.storage + "' for '--gpca-storage' (auto, int8, 2bit)"); }
        else if (f == "--gpca-stream") { a.stream = val(); if (a.stream != "auto" && a.stream != "on" && a.stream != "off") usage_error("invalid value '" + a.stream + "' for '--gpca-stream' (auto, on, off)"); }
        else if (f == "--gpca-panel-rows") a.panel_rows = to_i64(f, val());
        else if (f == "--gpca-rfit-power-iters") a.rfit_power_iters = to_i64(f, val());
        else if (f == "--gpca-eigensnp-local-stage") a.local_stage = true;
        else usage_error("unexpected argument '" + f + "' found");
    }
    if (a.output_prefix.empty()) usage_error("the following required arguments were not provided:\n  --out <OUTPUT_PREFIX>");
    return a;
}

void logmsg(const std::string& m) { std::fprintf(stderr, "[genomic_pca] %s\n", m.c_str()); }

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int engine_precision(const Args& a) { return a.precision == "i8" ? GPCA_PREC_I8_EXACT : GPCA_PREC_F32_MFMA; }
// storage "auto": a .bed of >= 1 024 samples stays in its own 2-bit form (a quarter of the HBM, faster packed kernels there);
// narrower matrices and VCF input are int8 (cli.py:_engine_modes)
int engine_storage(Args& a, int64_t bed_samples = 0) {
    if (a.storage == "auto") a.storage = bed_samples >= 1024 ? "2bit" : "int8";
    return a.storage == "2bit" ? GPCA_STORE_2BIT : GPCA_STORE_INT8;
}

// ------------------------------------------------------------------------------------------------ VCF workflow
int run_vcf_workflow(Args a) {
    if (a.vcf_dir.empty() || !a.have_components) {
        std::fprintf(stderr, "error: --vcf-dir and --components are required unless --eigensnp is given\n");
        return 2;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::string> files;
    if (DIR* d = opendir(a.vcf_dir.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (gpca_host::ends_with(n, ".vcf") || gpca_host::ends_with(n, ".vcf.gz")) files.push_back(a.vcf_dir + "/" + n);
        }
        closedir(d);
    }
    std::sort(files.begin(), files.end());
    if (files.empty()) { std::fprintf(stderr, "No VCF files found in %s\n", a.vcf_dir.c_str()); return 1; }          // main.rs:153-155
    gpca_host::VcfData v;
    for (size_t i = 0; i < files.size(); ++i) gpca_host::read_vcf(files[i], a.have_maf ? a.maf : 0.01, v, i == 0);
    const int64_t n_samples = (int64_t)v.samples.size(), n_variants = (int64_t)v.variant_ids.size();
    char buf[256];
    std::snprintf(buf, sizeof buf, "%zu VCF files, %lld variants x %lld samples in %.2fs", files.size(), (long long)n_variants, (long long)n_samples, seconds_since(t0));
    logmsg(buf);
    if (n_variants == 0) { std::fprintf(stderr, "No variants available to build matrix.\n"); return 1; }              // vcf.rs:321-323
    gpca::PCA model(a.device, engine_precision(a), engine_storage(a));
    model.rfit(v.dosages.data(), n_variants, n_samples, (int)a.components, 10, a.have_seed ? a.rfit_seed : 0, (int)a.rfit_power_iters);          // main.rs:636-656
    const std::vector<double> pcs = model.transform();
    gpca_host::ensure_parent(a.output_prefix);
    gpca_host::write_principal_components(a.output_prefix, "vcf.pca.tsv", v.samples, pcs.data(), n_samples, model.components());   // main.rs:231
    gpca_host::write_eigenvalues(a.output_prefix, a.write_eigenvalues ? model.explained_variance() : std::vector<double>());        // main.rs:232, 676
    std::snprintf(buf, sizeof buf, "VCF workflow done in %.2fs", seconds_since(t0));
    logmsg(buf);
    return 0;
}

// ------------------------------------------------------------------------------------------------ EigenSNP workflow
struct KeptColumns {            // host-side decode of the kept sample columns (a sample keep file): the panel source of that case
    const gpca_host::PlinkFileset* fs;
    std::vector<int64_t> cols;
};
extern "C" int fill_kept_columns(void* user, int64_t row0, int64_t rows, void* dst, int64_t ld) {
    static const int8_t lut[4] = {2, -127, 1, 0};                // count_a1 (prepare.rs:622-629)
    const KeptColumns* k = static_cast<const KeptColumns*>(user);
    int8_t* out = static_cast<int8_t*>(dst);
    for (int64_t r = 0; r < rows; ++r) {
        const uint8_t* src = k->fs->bed_rows + (row0 + r) * k->fs->bytes_per_row;
        int8_t* o = out + r * ld;
        for (size_t c = 0; c < k->cols.size(); ++c) { const int64_t s = k->cols[c]; o[c] = lut[(src[s >> 2] >> (2 * (s & 3))) & 3]; }
    }
    return 0;
}
// The .bed payload into the engine: resident, or -- when it does not fit the device, or on request -- out of core with the
// HBM panel cache on (cli.py:_load_bed; the reference pulls strips through the accessor on every pass, main.rs:322).
void load_bed(gpca::Engine& eng, const Args& a, const gpca_host::PlinkFileset& fs, KeptColumns* kept) {
    gpca_panel_source src;
    std::memset(&src, 0, sizeof src);
    const int64_t n_samples = kept ? (int64_t)kept->cols.size() : fs.n_samples;
    if (kept) { src.kind = GPCA_PANEL_HOST_I8; src.fill = fill_kept_columns; src.user = kept; }
    else {   // the memory-mapped payload itself is the source: the library's copy threads stage its panels, no callback
        src.kind = GPCA_PANEL_MAPPED_BED; src.user = const_cast<uint8_t*>(fs.bed_rows); src.host_ld = fs.bytes_per_row;
    }
    std::string mode = a.stream;
    if (mode == "auto") {
        // resident needs the matrix (1 B or 0.25 B per genotype, rows padded) plus the solver's workspace (gpca.h,
        // gpca_get_device_memory): a load that fits with nothing to spare would only fail later, in gpca_rsvd
        int64_t free_b = eng.device_memory().first;
        if (const char* e = std::getenv("GPCA_CLI_FREE_BYTES")) free_b = std::atoll(e);       // (test hook: pretend the device is smaller)
        const int64_t per_row = (n_samples + 1023) / 1024 * 1024 / (a.storage == "2bit" ? 4 : 1) + 512;
        const double need = (double)fs.n_snps * (double)(per_row + 1024) + (double)n_samples * 8192.0 + 1073741824.0;
        if (need > (double)free_b) {
            char buf[200];
            std::snprintf(buf, sizeof buf, "the genotype matrix needs about %.1f GiB resident, %.1f GiB are free: walking it out of core",
                          need / 1073741824.0, (double)free_b / 1073741824.0);
            logmsg(buf);
            mode = "on";
        }
    }
    if (mode != "on") {
        try {
            if (kept) eng.load_from_source(src, fs.n_snps, n_samples);
            else eng.upload_bed2bit(fs.bed_rows, fs.n_snps, fs.n_samples);     // the memory map goes up in 256 MiB row chunks, decoded on the GPU
            return;
        } catch (const gpca::Error& e) {
            if (mode == "off" || e.status() != GPCA_ERR_OOM) throw;
            logmsg("the genotype matrix does not fit the device: walking it out of core");
        }
    }
    eng.stream_open(src, fs.n_snps, n_samples, a.panel_rows, 3, true, -1);
}

int run_eigensnp_workflow(Args a) {
    if (a.bed_file.empty() || a.ld_block_file.empty()) {
        std::fprintf(stderr, "error: --bed-file and --ld-block-file are required when --eigensnp is used\n");           // main.rs:296-301
        return 2;
    }
    const auto t0 = std::chrono::steady_clock::now();
    gpca_host::PlinkFileset fs;
    gpca_host::read_plink(a.bed_file, fs);
    const int store = engine_storage(a, fs.n_samples);
    gpca::Engine eng(a.device, engine_precision(a), store);
    std::vector<std::string> sample_ids = fs.sample_ids;
    KeptColumns kept{&fs, {}};
    bool use_kept = false;
    if (!a.sample_keep_file.empty()) {                                                                                  // prepare.rs:1058-1096
        const auto ids = gpca_host::read_sample_keep_file(a.sample_keep_file);
        const std::set<std::string> keep_ids(ids.begin(), ids.end());
        sample_ids.clear();
        for (size_t i = 0; i < fs.sample_ids.size(); ++i)
            if (keep_ids.count(fs.sample_ids[i])) { kept.cols.push_back((int64_t)i); sample_ids.push_back(fs.sample_ids[i]); }
        if (kept.cols.empty()) { logmsg("No samples available after sample QC."); return 0; }
        use_kept = true;
    }
    load_bed(eng, a, fs, use_kept ? &kept : nullptr);
    const gpca::SnpStats st = eng.snp_stats(gpca::QcConfig{a.min_call_rate, a.min_maf, a.max_hwe_p});
    const auto blocks = gpca_host::parse_ld_block_file(a.ld_block_file);
    std::vector<uint8_t> keep;
    const auto by_tag = gpca_host::map_snps_to_ld_blocks(blocks, fs.chromosomes, fs.positions, st.keep, keep);
    int64_t n_qc = 0, n_in = 0;
    for (uint8_t k : st.keep) n_qc += k;
    for (uint8_t k : keep) n_in += k;
    char buf[256];
    std::snprintf(buf, sizeof buf, "%lld / %zu SNPs passed QC; %lld fall in %zu LD blocks", (long long)n_qc, st.keep.size(), (long long)n_in, by_tag.size());
    logmsg(buf);
    if (sample_ids.empty() || n_in == 0) { logmsg("No samples or SNPs available for EigenSNP PCA after preparation."); return 0; }   // main.rs:349-352
    eng.set_standardization(st.mu, st.sigma, keep);
    gpca::MicroarrayGenotypeAccessor acc(eng);
    const std::vector<int64_t> rows = acc.original_indices_of_pca_snps();
    std::unordered_map<int64_t, int64_t> row_to_id;
    row_to_id.reserve(rows.size() * 2);
    for (size_t i = 0; i < rows.size(); ++i) row_to_id[rows[i]] = (int64_t)i;
    std::vector<gpca::LdBlockSpecification> specs;
    for (const auto& tr : by_tag) {
        gpca::LdBlockSpecification s; s.user_defined_block_tag = tr.first;
        for (int64_t r : tr.second) s.pca_snp_ids_in_block.push_back(row_to_id.at(r));
        specs.push_back(std::move(s));
    }
    gpca::EigenSNPCoreAlgorithmConfig cfg;
    const int64_t lim = std::min<int64_t>((int64_t)sample_ids.size(), (int64_t)rows.size());
    const int64_t k = std::min<int64_t>(a.k_global, lim);
    cfg.target_num_global_pcs = (int)k;
    cfg.components_per_ld_block = (int)a.components_per_block;
    cfg.subset_factor_for_local_basis_learning = a.subset_factor;
    cfg.min_subset_size_for_local_basis_learning = a.min_subset; cfg.max_subset_size_for_local_basis_learning = a.max_subset;
    cfg.global_pca_sketch_oversampling = (int)std::max<int64_t>(0, std::min<int64_t>(a.global_oversampling, lim - k));
    cfg.global_pca_num_power_iterations = (int)a.global_power_iter;
    cfg.local_rsvd_sketch_oversampling = (int)a.local_oversampling; cfg.local_rsvd_num_power_iterations = (int)a.local_power_iter;
    cfg.random_seed = a.seed; cfg.snp_processing_strip_size = a.strip_size; cfg.refine_pass_count = (int)a.refine_passes;
    cfg.collect_diagnostics = a.collect_diagnostics;
    const gpca::EigenSNPCoreOutput out = gpca::EigenSNPCoreAlgorithm(cfg).compute_pca(acc, specs, a.local_stage);
    // the column count comes from the result: the local stage may leave fewer than k components (min(k, condensed features))
    const int kc = (int)out.num_principal_components_computed;
    gpca_host::ensure_parent(a.output_prefix);
    gpca_host::write_principal_components(a.output_prefix, "eigensnp.pca.tsv", sample_ids, out.final_sample_principal_component_scores.data(),
                                          out.num_qc_samples_used, kc);
    gpca_host::write_eigenvalues(a.output_prefix, out.final_principal_component_eigenvalues);
    std::vector<std::string> vids, chroms; std::vector<int64_t> pos;
    vids.reserve(rows.size()); chroms.reserve(rows.size()); pos.reserve(rows.size());
    for (int64_t r : rows) { vids.push_back(fs.variant_ids[(size_t)r]); chroms.push_back(fs.chromosomes[(size_t)r]); pos.push_back(fs.positions[(size_t)r]); }
    gpca_host::write_loadings(a.output_prefix, vids, chroms, pos, out.final_snp_principal_component_loadings.data(), (int64_t)rows.size(), kc);
    std::snprintf(buf, sizeof buf, "EigenSNP workflow done in %.2fs", seconds_since(t0));
    logmsg(buf);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const Args a = parse(argc, argv);
    try {
        return a.eigensnp ? run_eigensnp_workflow(a) : run_vcf_workflow(a);
    } catch (const gpca::Error& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
