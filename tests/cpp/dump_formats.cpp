// Test-only driver for genomic_pca_amd/host/formats.hpp (no GPU, no libgpca): parses the files named on the command line and
// prints what it read as plain text, one record per line, so that tests/test_cpp_host.py can hold the C++ host to io.py.
//   dump_formats ld <blocks> <bim-like: chrom pos keep per line>
//   dump_formats vcf <maf> <file>...
//   dump_formats plink <prefix.bed>
//   dump_formats writers <prefix>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "formats.hpp"

using namespace gpca_host;

int main(int argc, char** argv) {
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "ld" && argc == 4) {
            const auto blocks = parse_ld_block_file(argv[2]);
            for (const auto& b : blocks) std::printf("block\t%s\t%lld\t%lld\t%s\n", b.chrom.c_str(), (long long)b.start, (long long)b.end, b.tag.c_str());
            std::ifstream f(argv[3]);
            std::vector<std::string> chroms; std::vector<int64_t> pos; std::vector<uint8_t> qc, keep;
            std::string line;
            while (std::getline(f, line)) { const auto p = split_ws(line); if (p.size() == 3) { chroms.push_back(p[0]); pos.push_back(std::stoll(p[1])); qc.push_back((uint8_t)std::stoi(p[2])); } }
            const auto by_tag = map_snps_to_ld_blocks(blocks, chroms, pos, qc, keep);
            std::printf("keep");
            for (uint8_t k : keep) std::printf("\t%d", (int)k);
            std::printf("\n");
            for (const auto& t : by_tag) { std::printf("tag\t%s", t.first.c_str()); for (int64_t r : t.second) std::printf("\t%lld", (long long)r); std::printf("\n"); }
            return 0;
        }
        if (mode == "vcf" && argc >= 4) {
            VcfData v;
            for (int i = 3; i < argc; ++i) read_vcf(argv[i], std::atof(argv[2]), v, i == 3);
            std::printf("samples");
            for (const auto& s : v.samples) std::printf("\t%s", s.c_str());
            std::printf("\n");
            const size_t ns = v.samples.size();
            for (size_t i = 0; i < v.variant_ids.size(); ++i) {
                std::printf("%s", v.variant_ids[i].c_str());
                for (size_t s = 0; s < ns; ++s) std::printf("\t%d", (int)v.dosages[i * ns + s]);
                std::printf("\n");
            }
            return 0;
        }
        if (mode == "plink" && argc == 3) {
            PlinkFileset fs;
            read_plink(argv[2], fs);
            std::printf("dims\t%lld\t%lld\t%lld\n", (long long)fs.n_snps, (long long)fs.n_samples, (long long)fs.bytes_per_row);
            for (size_t i = 0; i < fs.sample_ids.size(); ++i) std::printf("sample\t%s\n", fs.sample_ids[i].c_str());
            for (int64_t i = 0; i < fs.n_snps; ++i) {
                std::printf("snp\t%s\t%s\t%lld", fs.chromosomes[(size_t)i].c_str(), fs.variant_ids[(size_t)i].c_str(), (long long)fs.positions[(size_t)i]);
                for (int64_t b = 0; b < fs.bytes_per_row; ++b) std::printf("\t%d", (int)fs.bed_rows[i * fs.bytes_per_row + b]);
                std::printf("\n");
            }
            return 0;
        }
        if (mode == "writers" && argc == 3) {
            const std::string pre = argv[2];
            ensure_parent(pre);
            const float pcs[4] = {1.23456789f, -0.5f, 2.0f, 1e-7f};
            write_principal_components(pre, "eigensnp.pca.tsv", {"s1", "s2", "s3"}, pcs, 2, 2);
            write_eigenvalues(pre + "_empty", {});
            write_eigenvalues(pre, {12.5, 0.1234567});
            const float ld[4] = {0.5f, -0.25f, 0.125f, 1.0f};
            write_loadings(pre, {"rs1", "rs2"}, {"1", "2"}, {10, 20}, ld, 2, 2);
            try { write_loadings(pre + "_bad", {"rs1"}, {"1", "2"}, {10, 20}, ld, 2, 1); } catch (const std::exception& e) { std::printf("mismatch\t%s\n", e.what()); }
            return 0;
        }
        std::fprintf(stderr, "usage: dump_formats ld|vcf|plink|writers ...\n");
        return 2;
    } catch (const std::exception& e) {
        std::printf("error\t%s\n", e.what());
        return 1;
    }
}
