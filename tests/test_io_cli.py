"""Host-side formats and the CLI surface.  Parsers/writers run on CPU; the two
end-to-end workflows need the GPU."""
import gzip
import os

import numpy as np
import pytest

from genomic_pca_amd import io as gio


def test_ld_block_parsing_and_mapping(tmp_path):
    p = tmp_path / "blocks.txt"
    p.write_text("# comment\nchr\tstart\tend\nchromosome\tstart\tend\nchr1 100 200\nCHR1\t150\t400\n2 1 50 extra\nbad line\n\nX 5 9\n")
    blocks = gio.parse_ld_block_file(str(p))
    # prepare.rs:1575-1602: headers/comments skipped, tag auto = chr:start-end on the normalised name
    assert blocks == [("1", 100, 200, "1:100-200"), ("1", 150, 400, "1:150-400"), ("2", 1, 50, "2:1-50"), ("x", 5, 9, "x:5-9")]
    assert gio.normalize_chromosome_name("Chr22") == "22" and gio.normalize_chromosome_name("X") == "x"
    chroms = ["1", "chr1", "1", "2", "2", "X", "3"]
    pos = [100, 180, 300, 50, 51, 7, 7]
    qc = np.array([1, 1, 1, 1, 1, 0, 1], np.uint8)
    keep, by_tag = gio.map_snps_to_ld_blocks(blocks, chroms, pos, qc)
    assert keep.tolist() == [1, 1, 1, 1, 0, 0, 0]          # 51 outside, X failed QC, chr3 has no block
    # first matching block wins (prepare.rs:1447-1463): 180 is in both chr1 blocks -> the first; tags sorted
    assert by_tag == [("1:100-200", [0, 1]), ("1:150-400", [2]), ("2:1-50", [3])]


def test_writers_match_reference_format(tmp_path):
    pre = str(tmp_path / "out" / "run")
    os.makedirs(os.path.dirname(pre))
    pcs = np.array([[1.23456789, -0.5], [2.0, 1e-7]], np.float32)
    gio.write_principal_components(pre, "eigensnp.pca.tsv", ["s1", "s2", "s3"], pcs)
    assert open(pre + ".eigensnp.pca.tsv").read() == "SampleID\tPC1\tPC2\ns1\t1.234568\t-0.500000\ns2\t2.000000\t0.000000\ns3\tNA\tNA\n"
    gio.write_eigenvalues(pre, [])
    assert open(pre + ".eigenvalues.tsv").read() == "PC\tEigenvalue\n"                 # main.rs:769-774
    gio.write_eigenvalues(pre, [12.5, 0.1234567])
    assert open(pre + ".eigenvalues.tsv").read() == "PC\tEigenvalue\n1\t12.500000\n2\t0.123457\n"
    gio.write_loadings(pre, ["rs1", "rs2"], ["1", "2"], [10, 20], np.array([[0.5, -0.25], [0.125, 1.0]], np.float32))
    assert open(pre + ".eigensnp.loadings.tsv").read() == \
        "VariantID\tChrom\tPos\tPC1_loading\tPC2_loading\nrs1\t1\t10\t0.500000\t-0.250000\nrs2\t2\t20\t0.125000\t1.000000\n"
    with pytest.raises(ValueError):
        gio.write_loadings(pre, ["rs1"], ["1", "2"], [10, 20], np.zeros((2, 1), np.float32))


def test_plink_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    G = rng.integers(0, 3, size=(50, 13), dtype=np.int8)
    G[3, 4] = -127; G[7, 12] = -127
    pre = str(tmp_path / "toy")
    gio.write_plink(pre, G, [f"s{i}" for i in range(13)], [f"rs{i}" for i in range(50)], ["1"] * 50, list(range(100, 150)))
    fs = gio.read_plink(pre + ".bed")
    assert fs.n_samples == 13 and fs.bed_rows.shape == (50, 4) and fs.sample_ids[2] == "s2" and fs.positions[49] == 149
    lut = np.array([2, -127, 1, 0], np.int8)                      # count_a1: 00->2, 01->missing, 10->1, 11->0
    dec = np.empty((50, 16), np.int8)
    for s in range(4):
        dec[:, s::4] = lut[(np.asarray(fs.bed_rows) >> (2 * s)) & 3]
    assert np.array_equal(dec[:, :13], G)
    open(pre + ".bed", "r+b").write(b"\x6c\x1b\x00")              # sample-major magic is rejected
    with pytest.raises(ValueError):
        gio.read_plink(pre + ".bed")


VCF_HEAD = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\tD\n"


def test_vcf_rules(tmp_path):
    body = ("1\t100\t.\tA\tG\t.\t.\t.\tGT\t0/0\t0|1\t1/1\t0/1\n"          # kept: dosages 0,1,2,1
            "1\t101\t.\tAT\tG\t.\t.\t.\tGT\t0/0\t0|1\t1/1\t0/1\n"         # REF not single base (vcf.rs:109-121)
            "1\t102\t.\tA\tG,T\t.\t.\t.\tGT\t0/0\t0|1\t1/1\t0/1\n"        # multi-allelic
            "1\t103\t.\tA\tG\t.\t.\t.\tGT\t0/0\t./.\t1/1\t0/1\n"          # missing GT drops the variant (vcf.rs:52-63)
            "1\t104\t.\tA\tG\t.\t.\t.\tGT\t0/0\t0/2\t1/1\t0/1\n"          # allele 2
            "1\t105\t.\tC\tT\t.\t.\t.\tGT:DP\t0/0:3\t0/0:4\t0/0:9\t0/0:1\n"  # MAF 0 < threshold (vcf.rs:244-266)
            "2\t7\t.\tC\tT\t.\t.\t.\tDP:GT\t3:1|0\t4:0/0\t9:0/0\t1:1|1\n")   # GT not first in FORMAT
    p = tmp_path / "a.vcf.gz"
    with gzip.open(p, "wt") as f:
        f.write(VCF_HEAD + body)
    samples, ids, G = gio.read_vcf(str(p), 0.01)
    assert samples == ["A", "B", "C", "D"]
    assert ids == ["1:100:A:G", "2:7:C:T"]                          # id = chr:pos:ref:alt (vcf.rs:268-276)
    assert G.tolist() == [[0, 1, 2, 1], [1, 0, 0, 2]]
    _, ids2, _ = gio.read_vcf(str(p), 0.4)                          # MAF threshold
    assert ids2 == ["1:100:A:G"]


# ------------------------------------------------------------------------------------------------ GPU end-to-end
def _read_tsv(path):
    lines = open(path).read().rstrip("\n").split("\n")
    return lines[0].split("\t"), [ln.split("\t") for ln in lines[1:]]


@pytest.mark.gpu
def test_cli_eigensnp_workflow(tmp_path, gpca, oracle):
    from genomic_pca_amd.cli import main
    M, N, P = 3000, 96, 6
    th = gpca.synth_thresholds(M, P, seed=3, fst=0.3)
    G = oracle.synth_genotypes(M, N, 3, th)
    G[::50] = 0                                                    # monomorphic rows fail QC
    pre = str(tmp_path / "cohort")
    chroms = ["1"] * 2000 + ["chr2"] * 1000
    pos = list(range(1, 2001)) + list(range(1, 1001))
    gio.write_plink(pre, G, [f"id{i}" for i in range(N)], [f"rs{i}" for i in range(M)], chroms, pos)
    ld = tmp_path / "ld.txt"
    ld.write_text("chr\tstart\tend\n1 1 1500\n2 1 600\n")         # SNPs 1501..2000 of chr1 and 601.. of chr2 are unblocked
    out = str(tmp_path / "res" / "run1")
    assert main(["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--out", out, "--eigensnp-k-global", "4",
                 "--eigensnp-max-hwe-p", "1.0", "--eigensnp-seed", "11"]) == 0
    hdr, rows = _read_tsv(out + ".eigensnp.pca.tsv")
    assert hdr == ["SampleID", "PC1", "PC2", "PC3", "PC4"] and len(rows) == N and rows[5][0] == "id5"
    ehdr, erows = _read_tsv(out + ".eigenvalues.tsv")
    assert ehdr == ["PC", "Eigenvalue"] and [r[0] for r in erows] == ["1", "2", "3", "4"]
    lhdr, lrows = _read_tsv(out + ".eigensnp.loadings.tsv")
    assert lhdr == ["VariantID", "Chrom", "Pos", "PC1_loading", "PC2_loading", "PC3_loading", "PC4_loading"]
    # expected PCA SNP set: QC (call rate .98, MAF .01, HWE off) AND inside an LD block
    st = oracle.snp_stats(G, N, 0.98, 0.01, 1.0)
    inblock = np.array([(i < 1500) or (2000 <= i < 2600) for i in range(M)])
    keep = st["keep"].astype(bool) & inblock
    assert [r[0] for r in lrows] == [f"rs{i}" for i in np.nonzero(keep)[0]]
    assert lrows[0][1] == "1" and lrows[-1][1] == "chr2"
    r, b = oracle.scale_shift(st["mu"], st["sigma"], keep)
    R = oracle.rsvd(G, N, r, b, 4, 10, 2, seed=11)
    sc = np.array([[float(x) for x in rw[1:]] for rw in rows])
    ev = np.array([float(rw[1]) for rw in erows])
    assert np.max(np.abs(ev - R["eigenvalues"]) / R["eigenvalues"]) < 1e-4
    al = oracle.sign_align(sc, R["scores"])
    assert np.max(np.abs(al - R["scores"])) < 2e-6 + 1e-4 * np.max(np.abs(R["scores"]))   # 6-decimal TSV rounding
    ldv = np.array([[float(x) for x in rw[3:]] for rw in lrows])
    assert np.max(np.abs(oracle.sign_align(ldv, R["loadings"][keep]) - R["loadings"][keep])) < 2e-6


@pytest.mark.gpu
def test_cli_vcf_workflow(tmp_path, gpca, oracle):
    from genomic_pca_amd.cli import main
    M, N = 400, 40
    th = gpca.synth_thresholds(M, 4, seed=8, fst=0.3)
    G = oracle.synth_genotypes(M, N, 8, th)
    names = [f"S{i}" for i in range(N)]
    gt = {0: "0/0", 1: "0|1", 2: "1/1"}
    d = tmp_path / "vcfs"; d.mkdir()
    for ci, (lo, hi) in enumerate([(0, 250), (250, 400)]):        # two chromosomes = two files, sorted by name
        with gzip.open(d / f"chr{ci + 1}.vcf.gz", "wt") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
            for i in range(lo, hi):
                f.write(f"{ci + 1}\t{i + 1}\t.\tA\tC\t.\t.\t.\tGT\t" + "\t".join(gt[int(v)] for v in G[i]) + "\n")
    out = str(tmp_path / "o" / "v")
    assert main(["--vcf-dir", str(d), "-k", "3", "--maf", "0.05", "--rfit-seed", "1", "--out", out]) == 0
    assert open(out + ".eigenvalues.tsv").read() == "PC\tEigenvalue\n"      # main.rs:676: empty vector -> header only
    hdr, rows = _read_tsv(out + ".vcf.pca.tsv")
    assert hdr == ["SampleID", "PC1", "PC2", "PC3"] and [r[0] for r in rows] == names
    af = G.sum(axis=1) / (2 * N)
    kept = np.minimum(af, 1 - af) >= 0.05
    Gk = G[kept]
    st = oracle.snp_stats(Gk, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(Gk, N, r, b, 3, 10, 2, seed=1)
    ref = oracle.standardized_dense(Gk, N, r, b).T @ R["loadings"]
    sc = np.array([[float(x) for x in rw[1:]] for rw in rows])
    assert oracle.max_abs_dpc(sc, ref) < 1e-4


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] on its own data: the reference's data/chr22_subset50 PLINK set through --eigensnp
# ------------------------------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _chr22_fileset(tmp_path):
    """120 000 consecutive SNPs of the reference's data/chr22_subset50.bed with its own .fam ids; the .bim is synthesised
    (chromosome 22, increasing positions) because data/chr22_subset50.bim.zip is missing from the reference checkout."""
    z = np.load(os.path.join(GOLD, "chr22_subset50_120k.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"]); M = rows.shape[0]
    pre = str(tmp_path / "chr22_subset50")
    with open(pre + ".bed", "wb") as f:
        f.write(b"\x6c\x1b\x01"); f.write(rows.tobytes())
    with open(pre + ".fam", "w") as f:
        for fid, iid in zip(z["fids"], z["iids"]):
            f.write(f"{fid} {iid} 0 0 0 -9\n")
    pos = 16_050_000 + 25 * np.arange(M)
    with open(pre + ".bim", "w") as f:
        for i in range(M):
            f.write(f"22\t22:{pos[i]}\t0\t{pos[i]}\tA\tG\n")
    lut = np.array([2, -127, 1, 0], np.int8)
    G = np.empty((M, rows.shape[1] * 4), np.int8)
    for s in range(4):
        G[:, s::4] = lut[(rows >> (2 * s)) & 3]
    return pre, z, G[:, :n], pos


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["int8", "2bit"])
def test_cli_config3_chr22_subset50(tmp_path, gpca, oracle, storage):
    """`genomic_pca --eigensnp --bed-file data/chr22_subset50.bed --ld-block-file <one genome-wide block> --eigensnp-k-global 20`
    (README.md:106-115 usage; main.rs:250-442) on real genotypes, N = 64 samples: QC decisions, eigenvalues, scores and
    loadings against the oracle (same seed) and the committed fixture, for both residencies."""
    from genomic_pca_amd.cli import main
    pre, z, G, pos = _chr22_fileset(tmp_path)
    n = G.shape[1]
    ld = tmp_path / "ld.txt"
    ld.write_text("22 1 500000000\n")
    out = str(tmp_path / "res" / storage)
    assert main(["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "20", "--out", out,
                 "--gpca-storage", storage]) == 0
    hdr, rows = _read_tsv(out + ".eigensnp.pca.tsv")
    assert hdr == ["SampleID"] + [f"PC{i}" for i in range(1, 21)] and [r[0] for r in rows] == [str(x) for x in z["iids"]]
    _, erows = _read_tsv(out + ".eigenvalues.tsv")
    _, lrows = _read_tsv(out + ".eigensnp.loadings.tsv")
    keep = z["keep"].astype(bool)                                   # QC with clap's effective defaults (main.rs:545-560)
    assert [r[0] for r in lrows] == [f"22:{p}" for p in pos[keep]]   # the PCA SNP set = the oracle's QC decisions
    st = oracle.snp_stats(G, n, 0.98, 0.01, 1e-6)
    assert np.array_equal(st["keep"], z["keep"])
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, n, r, b, 20, 10, 2, seed=2025)                # --eigensnp-seed default (main.rs:580)
    ev = np.array([float(rw[1]) for rw in erows])
    assert np.max(np.abs(ev - R["eigenvalues"]) / R["eigenvalues"]) < 1e-4
    assert np.allclose(ev, z["eigenvalues"], rtol=1e-4)             # ... and the committed numbers
    sc = np.array([[float(x) for x in rw[1:]] for rw in rows])
    al = oracle.sign_align(sc, R["scores"])
    assert np.max(np.abs(al - R["scores"])) < 2e-6 + 1e-4 * np.max(np.abs(R["scores"]))    # 6-decimal TSV rounding
    ldv = np.array([[float(x) for x in rw[3:]] for rw in lrows])
    assert np.max(np.abs(oracle.sign_align(ldv, R["loadings"][keep]) - R["loadings"][keep])) < 2e-6 + 1e-4
    # the converged answer (exact PCA, tests/pca.py:81-141 pattern): N = 64 gives a flat noise spectrum, so only the leading
    # PCs of the randomized PCA have converged at q = 2
    E = oracle.exact_pca(G, n, r, b, 20)
    assert abs(ev[0] - E["eigenvalues"][0]) / E["eigenvalues"][0] < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("keep_file", [False, True])
def test_cli_out_of_core_equals_resident(tmp_path, gpca, keep_file):
    """--gpca-stream on: the CLI walks the memory-mapped .bed panel by panel (8 panels here, the HBM cache on) instead of
    holding it resident -- the mode a .bed larger than the device gets by itself -- and writes the same three files as the
    resident run (scores / loadings within the 6-decimal TSV rounding; the streamed power iterations quantise per panel).
    With a sample keep file the panels are the kept columns, decoded on the host."""
    from genomic_pca_amd.cli import main
    pre, z, G, pos = _chr22_fileset(tmp_path)
    ld = tmp_path / "ld.txt"
    ld.write_text("22 1 500000000\n")
    extra = []
    if keep_file:
        kf = tmp_path / "keep.txt"
        kf.write_text("".join(f"{i}\n" for i in list(z["iids"])[::2] + ["not_in_the_fam"]))
        extra = ["--eigensnp-sample-keep-file", str(kf)]
    outs = {}
    for mode in ("off", "on"):
        out = str(tmp_path / "res" / mode)
        assert main(["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "8", "--out", out,
                     "--gpca-stream", mode, "--gpca-panel-rows", "16384"] + extra) == 0
        outs[mode] = [_read_tsv(out + sfx) for sfx in (".eigensnp.pca.tsv", ".eigenvalues.tsv", ".eigensnp.loadings.tsv")]
    (h0, sc0), (_, ev0), (_, ld0) = outs["off"]
    (h1, sc1), (_, ev1), (_, ld1) = outs["on"]
    assert h0 == h1 and [r[0] for r in sc0] == [r[0] for r in sc1] and [r[:3] for r in ld0] == [r[:3] for r in ld1]
    assert len(sc0) == (32 if keep_file else 64)
    num = lambda rows, c: np.array([[float(x) for x in r[c:]] for r in rows])
    assert np.allclose(num(ev0, 1), num(ev1, 1), rtol=1e-6)
    assert np.max(np.abs(num(sc0, 1) - num(sc1, 1))) < 5e-6 * max(1.0, np.max(np.abs(num(sc0, 1))))
    assert np.max(np.abs(num(ld0, 3) - num(ld1, 3))) < 5e-6


@pytest.mark.gpu
def test_engine_config3_full_precision(gpca, oracle):
    """The same data through the C ABI without the TSV rounding: unit-norm sign-aligned PCs within 1e-4 of the oracle on all
    three GEMM paths (N = 64 is padded to 256 / 1024 samples by the kernels' tiles)."""
    from genomic_pca_amd import _lib
    z = np.load(os.path.join(GOLD, "chr22_subset50_120k.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"])
    lut = np.array([2, -127, 1, 0], np.int8)
    G = np.empty((rows.shape[0], rows.shape[1] * 4), np.int8)
    for s in range(4):
        G[:, s::4] = lut[(rows >> (2 * s)) & 3]
    G = G[:, :n]
    st = oracle.snp_stats(G, n, 0.98, 0.01, 1e-6)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, n, r, b, 20, 10, 2, seed=2025)
    kept = st["keep"].astype(bool)
    for prec, store in ((_lib.PREC_I8_EXACT, _lib.STORE_INT8), (_lib.PREC_I8_EXACT, _lib.STORE_2BIT), (_lib.PREC_F32_MFMA, _lib.STORE_INT8)):
        with gpca.GpcaEngine(precision=prec, storage=store) as e:
            e.upload_bed2bit(rows, n)
            s2 = e.snp_stats(gpca.QcConfig())
            _, reason = e.snp_qc_detail()
            assert np.array_equal(s2["keep"], z["keep"]) and np.array_equal(reason, z["reason"]) and np.array_equal(s2["mu"], z["mu"])
            e.rsvd(20, 10, 2, seed=2025)
            assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < 1e-4
            assert oracle.max_abs_dpc(e.scores(f64=True), R["scores"]) < 1e-4
            assert oracle.max_abs_dpc(e.loadings().astype(np.float64), R["loadings"][kept]) < 1e-4


def test_vcf_fast_path_equals_per_sample_rules(tmp_path):
    """The vectorised GT-first path against the per-sample rule (_gt_to_dosage, vcf.rs:52-63) on random records, including
    broken genotypes that must drop the variant; and a throughput floor so that a chr22-scale file stays usable."""
    import time
    rng = np.random.default_rng(0)
    ns, nv = 300, 400
    names = [f"S{i}" for i in range(ns)]
    gts = np.array(["0|0", "0|1", "1|0", "1|1", "0/1", "1/1"])
    bad = ["./.", "0|2", "1", "0|1|1", ".|1", "01"]
    lines, expect = [], []
    for v in range(nv):
        g = gts[rng.integers(0, len(gts), ns)].tolist()
        fmt = ["GT", "GT:DP", "GT:DP:GQ"][v % 3]
        if v % 7 == 3:
            g[int(rng.integers(0, ns))] = bad[v % len(bad)]
        fields = [x + ("" if fmt == "GT" else ":7" + (":30" if fmt.endswith("GQ") else "")) for x in g]
        lines.append(f"1\t{v + 1}\t.\tA\tC\t.\t.\t.\t{fmt}\t" + "\t".join(fields))
        d = [gio._gt_to_dosage(x) for x in g]
        if None not in d:
            af = sum(d) / (2 * ns)
            if min(af, 1 - af) >= 0.01:
                expect.append((f"1:{v + 1}:A:C", d))
    p = tmp_path / "r.vcf"
    p.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n" + "\n".join(lines) + "\n")
    samples, ids, G = gio.read_vcf(str(p), 0.01)
    assert samples == names and ids == [e[0] for e in expect]
    assert np.array_equal(G, np.array([e[1] for e in expect], np.int8))
    # throughput: 2 000 variants x 2 504 samples (the 1000 Genomes sample count), GT-only columns
    ns2, nv2 = 2504, 2000
    row = "\t".join(gts[rng.integers(0, 4, ns2)].tolist())
    with open(tmp_path / "big.vcf", "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"s{i}" for i in range(ns2)) + "\n")
        for v in range(nv2):
            f.write(f"22\t{v + 1}\t.\tA\tG\t.\t.\t.\tGT\t{row}\n")
    t0 = time.perf_counter()
    _, ids2, G2 = gio.read_vcf(str(tmp_path / "big.vcf"), 0.0)
    dt = time.perf_counter() - t0
    assert G2.shape == (nv2, ns2) and dt < 5.0, dt          # 5M genotypes; the per-genotype loop needed ~10 s for this


@pytest.mark.gpu
def test_against_the_references_own_exact_pca_definition(gpca, oracle):
    """The reference validates its runs against tests/pca.py ("Exact PCA Reference": centre only, GRM / kept, eigh,
    PCs = evecs * sqrt(evals)).  The engine with caller-supplied standardisation (mu = per-SNP mean, sigma = 1) and enough
    power iterations to converge on N = 64 samples reproduces that definition on the reference's own chr22_subset50 genotypes:
    eigenvalues (rescaled (N-1)/kept) and the leading PCs."""
    from genomic_pca_amd import _lib
    z = np.load(os.path.join(GOLD, "chr22_subset50_120k.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"])
    lut = np.array([2, -127, 1, 0], np.int8)
    G = np.empty((rows.shape[0], rows.shape[1] * 4), np.int8)
    for s in range(4):
        G[:, s::4] = lut[(rows >> (2 * s)) & 3]
    G = G[:, :n]
    k = 6
    for store in (_lib.STORE_INT8, _lib.STORE_2BIT):
        with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=store) as e:
            e.upload_bed2bit(rows, n)
            st = e.snp_stats(gpca.QcConfig())                       # the QC both the Rust path and pca.py apply
            E = oracle.exact_pca_centred_only(G, n, st["keep"], k)
            e.set_standardization(st["mu"], np.ones_like(st["sigma"]), st["keep"])      # centre only
            e.rsvd(k, 20, 12, seed=1)                               # l = 26 of 63 dimensions, 12 power iterations: converged
            ev = e.eigenvalues() * (n - 1) / E["kept"]              # s^2/(N-1)  ->  s^2/kept
            pcs = e.scores(f64=True) / np.sqrt(E["kept"])           # V s        ->  V s / sqrt(kept)
            assert np.max(np.abs(ev[:4] - E["evals"][:4]) / E["evals"][:4]) < 1e-6          # (the oracle's rsvd reaches 1e-10 here)
            assert oracle.max_abs_dpc(pcs[:, :3], E["pcs"][:, :3]) < 1e-4                    # north_star's bar, against the EXACT answer
            al = oracle.sign_align(pcs[:, :3], E["pcs"][:, :3])
            assert np.max(np.abs(al - E["pcs"][:, :3])) < 1e-4 * np.max(np.abs(E["pcs"][:, :3]))


def test_cli_defaults_are_claps_effective_defaults():
    """main.rs:545-588 (`default_value_if("eigensnp", "true", ...)`), the values the authors' own sweep driver treats as the
    baseline (tests/sweep_run.py:31-45): the flag surface and its effective defaults."""
    from genomic_pca_amd.cli import build_parser
    a = build_parser().parse_args(["--eigensnp", "--bed-file", "x.bed", "--ld-block-file", "l.txt", "--out", "p"])
    assert (a.eigensnp_min_call_rate, a.eigensnp_min_maf, a.eigensnp_max_hwe_p) == (0.98, 0.01, 1e-6)
    assert (a.eigensnp_k_global, a.eigensnp_components_per_block, a.eigensnp_subset_factor) == (10, 7, 0.075)
    assert (a.eigensnp_min_subset_size, a.eigensnp_max_subset_size) == (10000, 40000)
    assert (a.eigensnp_global_oversampling, a.eigensnp_global_power_iter) == (10, 2)
    assert (a.eigensnp_local_oversampling, a.eigensnp_local_power_iter) == (10, 2)
    assert (a.eigensnp_seed, a.eigensnp_snp_strip_size, a.eigensnp_refine_passes) == (2025, 2000, 1)
    v = build_parser().parse_args(["-d", "vcfs", "-k", "10", "--maf", "0.05", "--rfit-seed", "1", "-o", "p", "-t", "8"])   # BASELINE configs[0]
    assert (v.vcf_dir, v.components, v.maf, v.rfit_seed, v.output_prefix, v.eigensnp) == ("vcfs", 10, 0.05, 1, "p", False)


def test_outputs_are_what_the_references_consumers_read(tmp_path):
    """The reference's own downstream scripts define what the output files must look like: tests/metrics.py:227-234 reads the PCA
    table with pandas (whitespace separated) and requires the columns SampleID, PC1..PCk; tests/plot.py:15-18,214-231 looks for
    `<prefix>.eigenvalues.tsv` (columns PC, Eigenvalue), `.eigensnp.pca.tsv` and `.eigensnp.loadings.tsv` (columns Pos,
    PC1_loading).  Written by the writers here, read back the way they read them."""
    import pandas as pd
    pre = str(tmp_path / "run")
    rng = np.random.default_rng(0)
    ids = [f"HG{i:05d}" for i in range(7)]
    pcs = rng.standard_normal((7, 4)).astype(np.float32)
    gio.write_principal_components(pre, "eigensnp.pca.tsv", ids, pcs)
    gio.write_eigenvalues(pre, [3.5, 2.25, 1.0, 0.5])
    gio.write_loadings(pre, [f"22:{p}" for p in (100, 200, 300)], ["22"] * 3, [100, 200, 300], rng.standard_normal((3, 4)).astype(np.float32))
    pca_table = pd.read_csv(pre + ".eigensnp.pca.tsv", sep=r"\s+")                                   # metrics.py:227
    assert list(pca_table.columns) == ["SampleID"] + [f"PC{i + 1}" for i in range(4)]                  # metrics.py:230
    assert pca_table["SampleID"].astype(str).str.strip().tolist() == ids                               # plot.py:219
    assert np.allclose(pca_table[[f"PC{i + 1}" for i in range(4)]].to_numpy(), pcs, atol=5e-7)          # '{:.6}' rounding
    ev = pd.read_csv(pre + ".eigenvalues.tsv", sep="\t")
    assert list(ev.columns) == ["PC", "Eigenvalue"] and len(ev) == 4                                   # plot.py:215-216
    ld = pd.read_csv(pre + ".eigensnp.loadings.tsv", sep="\t")
    assert {"Pos", "PC1_loading"} <= set(ld.columns) and ld["Pos"].tolist() == [100, 200, 300]          # plot.py:230
    pc_cols = sorted([c for c in pca_table.columns if c.startswith("PC") and c[2:].isdigit()], key=lambda x: int(x[2:]))
    assert int(pc_cols[-1][2:]) == 4                                                                    # plot.py:221-224
