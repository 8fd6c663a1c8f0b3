// Test-only client of include/gpca.hpp on the GPU: the mirrored types used the way the reference's host uses its own
// (main.rs:344-366, 598-679).  Reads an int8 SNP-major matrix from a raw file, prints results as text for tests/test_cpp_host.py.
//   hpp_client <matrix.i8> <M> <N>
#include <cstdio>
#include <fstream>
#include <vector>

#include "gpca.hpp"

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    const int64_t M = std::atoll(argv[2]), N = std::atoll(argv[3]);
    std::vector<int8_t> g((size_t)(M * N));
    std::ifstream f(argv[1], std::ios::binary);
    f.read(reinterpret_cast<char*>(g.data()), (std::streamsize)g.size());
    try {
        gpca::Engine eng;
        eng.upload_genotypes_i8(g.data(), M, N, N);
        const gpca::SnpStats st = eng.snp_stats(gpca::QcConfig{0.9, 0.01, 1e-6});
        gpca::MicroarrayGenotypeAccessor acc(eng);
        gpca::MicroarrayGenotypeAccessor acc2 = acc;                       // `Clone`: copies share the engine
        std::printf("dims\t%lld\t%lld\n", (long long)acc2.num_pca_snps(), (long long)acc2.num_qc_samples());
        const std::vector<float> blk = acc.get_standardized_snp_sample_block({0, 3, 5}, {1, 0, 7, 2});
        std::printf("block");
        for (float v : blk) std::printf("\t%.9g", (double)v);
        std::printf("\n");
        // two LD blocks that cover only part of the PCA SNPs: the PCA runs over their union, the accessor is left as it was
        const int64_t D = acc.num_pca_snps();
        gpca::LdBlockSpecification b1{"a", {}}, b2{"b", {}};
        for (int64_t i = 0; i < D / 3; ++i) b1.pca_snp_ids_in_block.push_back(i);
        for (int64_t i = D / 2; i < D; i += 2) b2.pca_snp_ids_in_block.push_back(i);
        gpca::EigenSNPCoreAlgorithmConfig cfg;
        cfg.target_num_global_pcs = 4; cfg.random_seed = 9;
        const gpca::EigenSNPCoreOutput out = gpca::EigenSNPCoreAlgorithm(cfg).compute_pca(acc, {b1, b2});
        std::printf("used\t%lld\t%lld\t%d\n", (long long)out.num_pca_snps_used, (long long)out.num_qc_samples_used, out.num_principal_components_computed);
        std::printf("eig");
        for (double v : out.final_principal_component_eigenvalues) std::printf("\t%.17g", v);
        std::printf("\nscores0");
        for (int c = 0; c < 4; ++c) std::printf("\t%.9g", (double)out.final_sample_principal_component_scores[(size_t)c]);
        std::printf("\nloadings\t%zu\n", out.final_snp_principal_component_loadings.size());
        std::printf("restored\t%lld\n", (long long)acc.num_pca_snps());
        try { gpca::EigenSNPCoreAlgorithm(cfg).compute_pca(acc, {gpca::LdBlockSpecification{"bad", {D}}}); }
        catch (const std::invalid_argument& e) { std::printf("range\t%s\n", e.what()); }
        try { acc.get_standardized_snp_sample_block({D + 5}, {0}); }
        catch (const gpca::Error& e) { std::printf("pull\t%d\n", e.status()); }
        // PCA::new().rfit(..).transform()
        gpca::PCA model;
        const std::vector<double> pcs = model.rfit(g.data(), M, N, 3, 10, 1).transform();
        std::printf("pca\t%d\t%zu", model.components(), pcs.size());
        for (double v : model.explained_variance()) std::printf("\t%.17g", v);
        std::printf("\n");
        try { gpca::PCA().rfit(g.data(), M, N, 0); } catch (const std::invalid_argument& e) { std::printf("k0\t%s\n", e.what()); }
        try { gpca::PCA().rfit(g.data(), M, 1, 2); } catch (const std::invalid_argument& e) { std::printf("n1\t%s\n", e.what()); }
        try { gpca::PCA().transform(); } catch (const std::logic_error& e) { std::printf("unfitted\t%s\n", e.what()); }
    } catch (const std::exception& e) {
        std::printf("error\t%s\n", e.what());
        return 1;
    }
    return 0;
}
