// Host-side file formats either side of the hot path, C++ twin of genomic_pca_amd/io.py:
// PLINK .bed/.bim/.fam, LD-block files, sample keep lists, a VCF genotype reader (plain or gzip/bgzf) and the TSV writers.
// Text parsing stays on the host; genotype bytes go to the GPU untouched (the memory-mapped 2-bit .bed payload, or int8
// dosages) and are decoded / QC'd there.
//
// Reference behaviour restated (file:line):
//   * BED: 3-byte magic 6c 1b 01 (SNP-major), ceil(N/4) bytes per SNP, 2 bits per sample LSB-first (tests/disk.py:89-135);
//     .bim chrom/sid/bp columns, .fam iid column (prepare.rs:940-970 via bed_reader).
//   * LD blocks: prepare.rs:1565-1616 (skip '#', "chr\t", "chromosome\t" headers; tag "chr:start-end"; chromosome names
//     lower-cased with leading "chr" stripped); SNP -> first matching block (prepare.rs:1447-1463).
//   * VCF: biallelic single-base REF/ALT only (vcf.rs:109-121); GT "a/b" or "a|b" with alleles 0/1, anything else drops
//     the variant (vcf.rs:52-63, 153-240); MAF filter default 0.01 (vcf.rs:244-266); id chr:pos:ref:alt.
//   * writers: main.rs:696-839 ("{:.6}" fixed formatting, the exact headers and file suffixes).
#ifndef GPCA_HOST_FORMATS_HPP
#define GPCA_HOST_FORMATS_HPP

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace gpca_host {

inline std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
        size_t j = i;
        while (j < s.size() && !std::isspace((unsigned char)s[j])) ++j;
        if (j > i) out.emplace_back(s, i, j - i);
        i = j;
    }
    return out;
}

inline bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// ---------------------------------------------------------------------------------------------- PLINK
struct PlinkFileset {
    const uint8_t* bed_rows = nullptr;   // [M][ceil(N/4)]: the memory-mapped .bed payload (after the 3-byte magic)
    int64_t n_snps = 0, n_samples = 0, bytes_per_row = 0;
    std::vector<std::string> sample_ids, variant_ids, chromosomes;
    std::vector<int64_t> positions;
    void* map_base = nullptr; size_t map_len = 0;
    PlinkFileset() = default;
    PlinkFileset(const PlinkFileset&) = delete;
    PlinkFileset& operator=(const PlinkFileset&) = delete;
    ~PlinkFileset() { if (map_base) munmap(map_base, map_len); }
};

inline std::string strip_plink_ext(const std::string& p) {
    for (const char* e : {".bed", ".bim", ".fam"}) if (ends_with(p, e)) return p.substr(0, p.size() - 4);
    return p;
}

inline void read_plink(const std::string& bed_path, PlinkFileset& fs) {
    const std::string prefix = strip_plink_ext(bed_path);
    std::string line;
    {
        std::ifstream f(prefix + ".fam");
        if (!f) throw std::runtime_error("cannot open " + prefix + ".fam");
        while (std::getline(f, line)) {
            const auto p = split_ws(line);
            if (!p.empty()) fs.sample_ids.push_back(p.size() > 1 ? p[1] : p[0]);
        }
    }
    {
        std::ifstream f(prefix + ".bim");
        if (!f) throw std::runtime_error("cannot open " + prefix + ".bim");
        while (std::getline(f, line)) {
            const auto p = split_ws(line);
            if (p.size() >= 4) { fs.chromosomes.push_back(p[0]); fs.variant_ids.push_back(p[1]); fs.positions.push_back(std::stoll(p[3])); }
        }
    }
    fs.n_samples = (int64_t)fs.sample_ids.size(); fs.n_snps = (int64_t)fs.variant_ids.size();
    fs.bytes_per_row = (fs.n_samples + 3) / 4;
    const std::string bed = prefix + ".bed";
    const int fd = open(bed.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + bed);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); throw std::runtime_error("cannot stat " + bed); }
    unsigned char magic[3] = {0, 0, 0};
    if (pread(fd, magic, 3, 0) != 3 || magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01) {
        close(fd); throw std::runtime_error(bed + ": not a SNP-major PLINK .bed");
    }
    if ((int64_t)sb.st_size != 3 + fs.n_snps * fs.bytes_per_row) {
        close(fd);
        throw std::runtime_error(bed + ": size " + std::to_string((long long)sb.st_size) + " does not match " + std::to_string(fs.n_snps) + " SNPs x " +
                                 std::to_string(fs.n_samples) + " samples");
    }
    fs.map_len = (size_t)sb.st_size;
    fs.map_base = mmap(nullptr, fs.map_len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (fs.map_base == MAP_FAILED) { fs.map_base = nullptr; throw std::runtime_error("cannot mmap " + bed); }
    fs.bed_rows = static_cast<const uint8_t*>(fs.map_base) + 3;
}

// ---------------------------------------------------------------------------------------------- LD blocks
inline std::string normalize_chromosome_name(std::string name) {   // prepare.rs:1610-1616
    std::transform(name.begin(), name.end(), name.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    while (name.compare(0, 3, "chr") == 0) name = name.substr(3);
    return name;
}

struct LdBlock { std::string chrom; int64_t start, end; std::string tag; };

inline std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

inline std::vector<LdBlock> parse_ld_block_file(const std::string& path) {   // prepare.rs:1565-1607
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open LD block file " + path);
    std::vector<LdBlock> blocks;
    std::string line;
    while (std::getline(f, line)) {
        const std::string t = trim(line);
        if (t.empty() || t[0] == '#' || t.compare(0, 4, "chr\t") == 0 || t.compare(0, 11, "chromosome\t") == 0) continue;
        const auto p = split_ws(t);
        if (p.size() < 3) continue;
        LdBlock b;
        b.chrom = normalize_chromosome_name(p[0]); b.start = std::stoll(p[1]); b.end = std::stoll(p[2]);
        b.tag = b.chrom + ":" + std::to_string(b.start) + "-" + std::to_string(b.end);
        blocks.push_back(b);
    }
    return blocks;
}

// prepare.rs:1424-1563: every QC-passing SNP goes to the FIRST block (file order) that contains it.  Returns the keep mask
// restricted to SNPs inside some block and the blocks' original row lists, sorted by tag.
inline std::vector<std::pair<std::string, std::vector<int64_t>>> map_snps_to_ld_blocks(
    const std::vector<LdBlock>& blocks, const std::vector<std::string>& chromosomes, const std::vector<int64_t>& positions,
    const std::vector<uint8_t>& qc_keep, std::vector<uint8_t>& keep_out) {
    const size_t M = positions.size();
    std::vector<std::string> norm(M);
    for (size_t i = 0; i < M; ++i) norm[i] = normalize_chromosome_name(chromosomes[i]);
    // blocks per chromosome, in file order
    std::map<std::string, std::vector<size_t>> by_chrom;
    for (size_t b = 0; b < blocks.size(); ++b) by_chrom[blocks[b].chrom].push_back(b);
    std::vector<int64_t> assigned(M, -1);
    for (size_t i = 0; i < M; ++i) {
        if (!qc_keep[i]) continue;
        const auto it = by_chrom.find(norm[i]);
        if (it == by_chrom.end()) continue;
        for (size_t b : it->second)
            if (positions[i] >= blocks[b].start && positions[i] <= blocks[b].end) { assigned[i] = (int64_t)b; break; }
    }
    keep_out.assign(M, 0);
    std::map<std::string, std::vector<int64_t>> by_tag;
    for (size_t i = 0; i < M; ++i)
        if (assigned[i] >= 0) { keep_out[i] = 1; by_tag[blocks[(size_t)assigned[i]].tag].push_back((int64_t)i); }
    std::vector<std::pair<std::string, std::vector<int64_t>>> out(by_tag.begin(), by_tag.end());   // std::map: sorted by tag, rows ascending
    return out;
}

inline std::vector<std::string> read_sample_keep_file(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open sample keep file " + path);
    std::vector<std::string> ids;
    std::string line;
    while (std::getline(f, line)) { const auto p = split_ws(line); if (!p.empty()) ids.push_back(p[0]); }
    return ids;
}

// ---------------------------------------------------------------------------------------------- VCF
// A line reader over zlib's gzFile: reads plain text, gzip and bgzf (concatenated gzip members) alike.
class LineReader {
public:
    explicit LineReader(const std::string& path) : gz_(gzopen(path.c_str(), "rb")) {
        if (!gz_) throw std::runtime_error("cannot open " + path);
        gzbuffer(gz_, 1 << 20);
        buf_.resize(1 << 20);
    }
    ~LineReader() { if (gz_) gzclose(gz_); }
    LineReader(const LineReader&) = delete;
    LineReader& operator=(const LineReader&) = delete;
    bool next(std::string& line) {
        line.clear();
        for (;;) {
            if (pos_ == len_) {
                const int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
                if (n < 0) throw std::runtime_error("read error in compressed stream");
                if (n == 0) return !line.empty();
                pos_ = 0; len_ = (size_t)n;
            }
            const char* p = static_cast<const char*>(std::memchr(buf_.data() + pos_, '\n', len_ - pos_));
            if (p) { line.append(buf_.data() + pos_, (size_t)(p - (buf_.data() + pos_))); pos_ = (size_t)(p - buf_.data()) + 1; break; }
            line.append(buf_.data() + pos_, len_ - pos_);
            pos_ = len_;
        }
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        return true;
    }
private:
    gzFile gz_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
};

struct VcfData {
    std::vector<std::string> samples, variant_ids;
    std::vector<int8_t> dosages;      // [variants][samples], SNP-major
};

// the genotype of one sample field whose GT sits at FORMAT position gi: exactly 3 bytes a/b or a|b, alleles 0/1 (vcf.rs:52-63)
inline int gt_dosage(const char* f, const char* fend, int gi) {
    for (int c = 0; c < gi; ++c) {
        const char* q = static_cast<const char*>(std::memchr(f, ':', (size_t)(fend - f)));
        if (!q) return -1;
        f = q + 1;
    }
    const char* q = static_cast<const char*>(std::memchr(f, ':', (size_t)(fend - f)));
    const char* e = q ? q : fend;
    if (e - f != 3 || (f[1] != '/' && f[1] != '|') || (f[0] != '0' && f[0] != '1') || (f[2] != '0' && f[2] != '1')) return -1;
    return (f[0] - '0') + (f[2] - '0');
}

// vcf.rs:65-286: appends the variants of one file that pass the rules; the first file defines the sample list
inline void read_vcf(const std::string& path, double maf_threshold, VcfData& out, bool first_file) {
    LineReader rd(path);
    std::string line;
    std::vector<std::string> samples;
    std::vector<int8_t> d;
    while (rd.next(line)) {
        if (line.compare(0, 2, "##") == 0) continue;
        if (line.compare(0, 6, "#CHROM") == 0) {
            size_t pos = 0; int col = 0;
            while (pos <= line.size()) {
                const size_t tab = line.find('\t', pos);
                const size_t e = tab == std::string::npos ? line.size() : tab;
                if (col >= 9) samples.emplace_back(line, pos, e - pos);
                ++col;
                if (tab == std::string::npos) break;
                pos = tab + 1;
            }
            if (samples.empty()) throw std::runtime_error("VCF header from " + path + " contains no samples.");      // vcf.rs:31-36
            if (first_file) out.samples = samples;
            else if (samples != out.samples) throw std::runtime_error("Sample mismatch between VCF files: " + path);   // vcf.rs:78-95
            d.resize(samples.size());
            continue;
        }
        if (samples.empty()) continue;
        // the first nine columns
        size_t col_start[10]; size_t pos = 0; int ncol = 0;
        while (ncol < 10) {
            col_start[ncol++] = pos;
            const size_t tab = line.find('\t', pos);
            if (tab == std::string::npos) break;
            pos = tab + 1;
        }
        if (ncol < 10) continue;
        auto col = [&](int c) { return std::string(line, col_start[c], col_start[c + 1] - col_start[c] - 1); };
        const std::string ref = col(3), alt = col(4);
        if (ref.size() != 1 || alt.size() != 1 || alt == ",") continue;                                                // vcf.rs:109-121
        const std::string fmt = col(8);
        int gi = -1, idx = 0;
        for (size_t a = 0; a <= fmt.size();) {
            const size_t c = fmt.find(':', a);
            const size_t e = c == std::string::npos ? fmt.size() : c;
            if (fmt.compare(a, e - a, "GT") == 0 && gi < 0) gi = idx;
            ++idx;
            if (c == std::string::npos) break;
            a = c + 1;
        }
        if (gi < 0) continue;
        const size_t ns = samples.size();
        const char* p = line.data() + col_start[9];
        const char* end = line.data() + line.size();
        bool ok = true;
        int64_t sum = 0;
        size_t si = 0;
        while (ok) {
            const char* t = static_cast<const char*>(std::memchr(p, '\t', (size_t)(end - p)));
            const char* fe = t ? t : end;
            if (si >= ns) { ok = false; break; }
            const int v = gt_dosage(p, fe, gi);
            if (v < 0) { ok = false; break; }                                                                            // any bad GT drops the variant
            d[si++] = (int8_t)v; sum += v;
            if (!t) break;
            p = t + 1;
        }
        if (!ok || si != ns) continue;
        const double af = (double)sum / (2.0 * (double)ns);
        if (std::min(af, 1.0 - af) < maf_threshold) continue;                                                           // vcf.rs:244-266
        out.variant_ids.push_back(col(0) + ":" + col(1) + ":" + ref + ":" + alt);
        out.dosages.insert(out.dosages.end(), d.begin(), d.end());
    }
    if (first_file && out.samples.empty() && !samples.empty()) out.samples = samples;
}

// ---------------------------------------------------------------------------------------------- writers
inline void ensure_parent(const std::string& prefix) {   // main.rs:372-377
    const size_t s = prefix.find_last_of('/');
    if (s == std::string::npos || s == 0) return;
    const std::string dir = prefix.substr(0, s);
    std::string cur;
    size_t pos = 0;
    while (pos <= dir.size()) {
        const size_t n = dir.find('/', pos);
        cur = dir.substr(0, n == std::string::npos ? dir.size() : n);
        if (!cur.empty()) mkdir(cur.c_str(), 0777);
        if (n == std::string::npos) break;
        pos = n + 1;
    }
}

struct OutFile {
    FILE* f;
    explicit OutFile(const std::string& path) : f(std::fopen(path.c_str(), "w")) {
        if (!f) throw std::runtime_error("cannot create " + path);
        std::setvbuf(f, nullptr, _IOFBF, 1 << 20);
    }
    ~OutFile() { if (f) std::fclose(f); }
};

// main.rs:696-762: header SampleID\tPC1..; "{:.6}".  suffix = "vcf.pca.tsv" or "eigensnp.pca.tsv"; pcs is [rows][k]
template <typename T>
inline void write_principal_components(const std::string& prefix, const std::string& suffix, const std::vector<std::string>& sample_names,
                                       const T* pcs, int64_t rows, int k) {
    if (k == 0) return;
    OutFile o(prefix + "." + suffix);
    std::fputs("SampleID", o.f);
    for (int c = 1; c <= k; ++c) std::fprintf(o.f, "\tPC%d", c);
    std::fputc('\n', o.f);
    for (size_t i = 0; i < sample_names.size(); ++i) {
        std::fputs(sample_names[i].c_str(), o.f);
        for (int c = 0; c < k; ++c) {
            if ((int64_t)i < rows) std::fprintf(o.f, "\t%.6f", (double)pcs[i * (size_t)k + (size_t)c]);
            else std::fputs("\tNA", o.f);
        }
        std::fputc('\n', o.f);
    }
}

inline void write_eigenvalues(const std::string& prefix, const std::vector<double>& ev) {   // main.rs:765-784: header even when empty
    OutFile o(prefix + ".eigenvalues.tsv");
    std::fputs("PC\tEigenvalue\n", o.f);
    for (size_t i = 0; i < ev.size(); ++i) std::fprintf(o.f, "%zu\t%.6f\n", i + 1, ev[i]);
}

inline void write_loadings(const std::string& prefix, const std::vector<std::string>& variant_ids, const std::vector<std::string>& chroms,
                           const std::vector<int64_t>& positions, const float* loadings, int64_t rows, int k) {   // main.rs:787-839
    if (k == 0) return;
    if (!variant_ids.empty() && !(variant_ids.size() == chroms.size() && chroms.size() == positions.size() && (int64_t)positions.size() == rows))
        throw std::runtime_error("Mismatch in lengths of variant metadata and loadings matrix rows.");   // main.rs:817-824
    OutFile o(prefix + ".eigensnp.loadings.tsv");
    std::fputs("VariantID\tChrom\tPos", o.f);
    for (int c = 1; c <= k; ++c) std::fprintf(o.f, "\tPC%d_loading", c);
    std::fputc('\n', o.f);
    for (size_t i = 0; i < variant_ids.size(); ++i) {
        std::fprintf(o.f, "%s\t%s\t%lld", variant_ids[i].c_str(), chroms[i].c_str(), (long long)positions[i]);
        for (int c = 0; c < k; ++c) std::fprintf(o.f, "\t%.6f", (double)loadings[i * (size_t)k + (size_t)c]);
        std::fputc('\n', o.f);
    }
}

}  // namespace gpca_host

#endif
