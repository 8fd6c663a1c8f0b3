"""The native host program (genomic_pca_amd/host: include/gpca.hpp + formats.hpp + genomic_pca.cpp -> bin/genomic_pca): the
reference's command line as a compiled program over the C ABI.  CPU: its parsers and writers against genomic_pca_amd/io.py
(through a test-only dump driver), its argument handling, and that it fails loudly without a GPU.  GPU: both workflows end to
end, byte for byte the files `python -m genomic_pca_amd` writes."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from genomic_pca_amd import io as gio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SANITIZE = os.environ.get("GPCA_HOST_SANITIZE") == "1"            # scripts/sanitize_cpu.sh: the host program and the parser harness under ASan + UBSan
BIN = os.path.join(ROOT, "genomic_pca_amd", "bin", "genomic_pca_asan" if SANITIZE else "genomic_pca")
SANFLAGS = ["-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if SANITIZE else []
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def host_bin(gpca):
    gpca.load()                                                   # (builds libgpca.so when it is missing)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "genomic_pca_amd", "host"), "-s"] + (["asan"] if SANITIZE else []))
    assert os.path.exists(BIN)
    return BIN


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp") / "dump_formats")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror"] + SANFLAGS + ["-I" + os.path.join(ROOT, "genomic_pca_amd", "host"),
                           os.path.join(ROOT, "tests", "cpp", "dump_formats.cpp"), "-lz", "-o", exe])

    def run(*args):
        out = subprocess.run([exe, *map(str, args)], capture_output=True, text=True)
        return out.returncode, [ln.split("\t") for ln in out.stdout.rstrip("\n").split("\n")]
    return run


def test_cpp_ld_blocks_match_io_py(tmp_path, dump):
    p = tmp_path / "blocks.txt"
    p.write_text("# comment\nchr\tstart\tend\nchromosome\tstart\tend\nchr1 100 200\nCHR1\t150\t400\n2 1 50 extra\nbad line\n\nX 5 9\nchrchr7 1 2\n")
    chroms = ["1", "chr1", "1", "2", "2", "X", "3", "Chr7"]
    pos = [100, 180, 300, 50, 51, 7, 7, 2]
    qc = [1, 1, 1, 1, 1, 0, 1, 1]
    (tmp_path / "snps.txt").write_text("".join(f"{c} {q} {k}\n" for c, q, k in zip(chroms, pos, qc)))
    rc, rows = dump("ld", p, tmp_path / "snps.txt")
    assert rc == 0
    blocks = gio.parse_ld_block_file(str(p))
    assert [(r[1], int(r[2]), int(r[3]), r[4]) for r in rows if r[0] == "block"] == blocks
    keep, by_tag = gio.map_snps_to_ld_blocks(blocks, chroms, pos, np.array(qc, np.uint8))
    assert [int(x) for x in next(r for r in rows if r[0] == "keep")[1:]] == keep.tolist()
    assert [(r[1], [int(x) for x in r[2:]]) for r in rows if r[0] == "tag"] == by_tag


def test_cpp_vcf_reader_matches_io_py(tmp_path, dump):
    """The rules of vcf.rs:52-63, 109-121, 244-266 on the records of tests/test_io_cli.py::test_vcf_rules plus random ones
    (GT first / not first, extra FORMAT keys, broken genotypes), plain and gzip, two files with one sample list."""
    rng = np.random.default_rng(5)
    ns = 37
    names = [f"S{i}" for i in range(ns)]
    head = "##fileformat=VCFv4.2\n##contig=<ID=1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    gts = np.array(["0|0", "0|1", "1|0", "1|1", "0/1", "1/1"])
    bad = ["./.", "0|2", "1", "0|1|1", ".|1", "01", ""]

    def body(n, chrom):
        lines = []
        for v in range(n):
            g = gts[rng.integers(0, len(gts), ns)].tolist()
            fmt = ["GT", "GT:DP", "GT:DP:GQ", "DP:GT", "DP:GQ:GT"][v % 5]
            if v % 7 == 3:
                g[int(rng.integers(0, ns))] = bad[v % len(bad)]
            gi = fmt.split(":").index("GT")
            fields = [":".join(["7"] * gi + [x] + ["9"] * (len(fmt.split(":")) - gi - 1)) for x in g]
            ref, alt = [("A", "C"), ("AT", "C"), ("A", "C,T"), ("G", "T")][v % 4 if v % 11 == 0 else 0]
            if v % 13 == 5:
                fields = fields[:-1]                                  # a short record
            lines.append(f"{chrom}\t{v + 1}\trs{v}\t{ref}\t{alt}\t.\tPASS\t.\t{fmt}\t" + "\t".join(fields))
        return "\n".join(lines) + "\n"
    f1, f2 = tmp_path / "a.vcf", tmp_path / "b.vcf.gz"
    f1.write_text(head + body(120, "1"))
    with gzip.open(f2, "wt") as f:
        f.write(head + body(90, "chr2"))
    for maf in (0.0, 0.05, 0.3):
        want_ids, want_rows = [], []
        for f in (f1, f2):
            s, ids, G = gio.read_vcf(str(f), maf)
            assert s == names
            want_ids += ids; want_rows += G.tolist()
        rc, rows = dump("vcf", maf, f1, f2)
        assert rc == 0 and rows[0] == ["samples"] + names
        assert [r[0] for r in rows[1:]] == want_ids and len(want_ids) > 20
        assert [[int(x) for x in r[1:]] for r in rows[1:]] == want_rows
    other = tmp_path / "c.vcf"
    other.write_text(head.replace("S3\t", "X3\t") + body(5, "3"))
    rc, rows = dump("vcf", 0.0, f1, other)
    assert rc == 1 and "Sample mismatch between VCF files" in rows[-1][1]           # vcf.rs:78-95


def test_cpp_plink_reader_and_writers(tmp_path, dump):
    rng = np.random.default_rng(0)
    G = rng.integers(0, 3, size=(50, 13), dtype=np.int8)
    G[3, 4] = -127
    pre = str(tmp_path / "toy")
    gio.write_plink(pre, G, [f"s{i}" for i in range(13)], [f"rs{i}" for i in range(50)], ["1"] * 25 + ["chrX"] * 25, list(range(100, 150)))
    fs = gio.read_plink(pre + ".bed")
    rc, rows = dump("plink", pre + ".bed")
    assert rc == 0 and rows[0] == ["dims", "50", "13", "4"]
    assert [r[1] for r in rows if r[0] == "sample"] == fs.sample_ids
    snps = [r for r in rows if r[0] == "snp"]
    assert [r[1] for r in snps] == fs.chromosomes and [r[2] for r in snps] == fs.variant_ids
    assert [int(r[3]) for r in snps] == fs.positions.tolist()
    assert np.array_equal(np.array([[int(x) for x in r[4:]] for r in snps], np.uint8), np.asarray(fs.bed_rows))
    open(pre + ".bed", "r+b").write(b"\x6c\x1b\x00")                              # sample-major magic is rejected
    rc, rows = dump("plink", pre + ".bed")
    assert rc == 1 and "not a SNP-major PLINK .bed" in rows[-1][1]
    # writers: the bytes of io.py's (main.rs:696-839 formats)
    cpp, py = str(tmp_path / "c" / "run"), str(tmp_path / "p" / "run")
    os.makedirs(os.path.dirname(py))
    rc, rows = dump("writers", cpp)
    assert rc == 0 and rows[-1][0] == "mismatch" and "Mismatch in lengths" in rows[-1][1]
    pcs = np.array([[1.23456789, -0.5], [2.0, 1e-7]], np.float32)
    gio.write_principal_components(py, "eigensnp.pca.tsv", ["s1", "s2", "s3"], pcs)
    gio.write_eigenvalues(py + "_empty", [])
    gio.write_eigenvalues(py, [12.5, 0.1234567])
    gio.write_loadings(py, ["rs1", "rs2"], ["1", "2"], [10, 20], np.array([[0.5, -0.25], [0.125, 1.0]], np.float32))
    for sfx in (".eigensnp.pca.tsv", "_empty.eigenvalues.tsv", ".eigenvalues.tsv", ".eigensnp.loadings.tsv"):
        assert open(cpp + sfx).read() == open(py + sfx).read(), sfx


def test_cpp_cli_arguments_and_no_gpu(tmp_path, host_bin):
    run = lambda *a: subprocess.run([host_bin, *a], capture_output=True, text=True)
    h = run("--help")
    assert h.returncode == 0
    for flag in ("--vcf-dir", "--bed-file", "--eigensnp", "--out", "--components", "--maf", "--rfit-seed", "--ld-block-file",
                 "--eigensnp-k-global", "--eigensnp-sample-keep-file", "--eigensnp-seed"):
        assert flag in h.stdout
    assert run().returncode == 2 and "--out" in run().stderr
    assert run("--out", "x").returncode == 2                                        # neither workflow selected
    assert run("--out", "x", "--eigensnp").returncode == 2                          # main.rs:296-301
    r = run("--out", "x", "--bogus")
    assert r.returncode == 2 and "unexpected argument '--bogus'" in r.stderr
    assert run("--out", "x", "-k", "abc", "-d", ".").returncode == 2
    r = run("--out", str(tmp_path / "o"), "--eigensnp", "--bed-file", str(tmp_path / "none.bed"), "--ld-block-file", "x")
    assert r.returncode == 1 and "cannot open" in r.stderr
    (tmp_path / "empty").mkdir()
    r = run("--out", str(tmp_path / "o"), "-d", str(tmp_path / "empty"), "-k", "2")
    assert r.returncode == 1 and "No VCF files found" in r.stderr                  # main.rs:153-155
    from conftest import gpu_present
    if gpu_present():
        return
    # a parsable input and no GPU: the program stops at gpca_create -- there is no CPU path behind it
    d = tmp_path / "v"; d.mkdir()
    (d / "a.vcf").write_text("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\n1\t5\t.\tA\tG\t.\t.\t.\tGT\t0/1\t1/1\t0/0\n")
    r = run("--out", str(tmp_path / "o"), "-d", str(d), "-k", "1")
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    assert not os.path.exists(str(tmp_path / "o") + ".vcf.pca.tsv")


# ------------------------------------------------------------------------------------------------ GPU end to end
def _chr22_fileset(tmp_path):
    z = np.load(os.path.join(GOLD, "chr22_subset50_120k.npz"))
    rows = z["bed_rows"]; M = rows.shape[0]
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
    return pre, z


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--gpca-storage", "2bit"], ["--gpca-stream", "on", "--gpca-panel-rows", "16384"], ["KEEP"],
                                   ["--gpca-precision", "f32"]])
def test_cpp_eigensnp_workflow_equals_python_cli(tmp_path, host_bin, extra):
    """configs[2] data (the reference's chr22_subset50 genotypes, 120k-SNP slice) with several LD blocks, through the compiled
    host program and through `python -m genomic_pca_amd`: the same three files, byte for byte."""
    from genomic_pca_amd.cli import main
    pre, z = _chr22_fileset(tmp_path)
    ld = tmp_path / "ld.txt"
    ld.write_text("chr\tstart\tend\nchr22 16050000 16900000\n22 16800000 17500000\n22 18000000 500000000\n")
    extra = list(extra)
    if extra == ["KEEP"]:
        kf = tmp_path / "keep.txt"
        kf.write_text("".join(f"{i}\n" for i in list(z["iids"])[1::2]))
        extra = ["--eigensnp-sample-keep-file", str(kf)]
    common = ["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "12", "--eigensnp-seed", "7"] + extra
    out_c, out_p = str(tmp_path / "c" / "run"), str(tmp_path / "p" / "run")
    r = subprocess.run([host_bin, "--out", out_c] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "SNPs passed QC" in r.stderr
    assert main(["--out", out_p] + common) == 0
    for sfx in (".eigensnp.pca.tsv", ".eigenvalues.tsv", ".eigensnp.loadings.tsv"):
        a, b = open(out_c + sfx).read(), open(out_p + sfx).read()
        assert a == b, sfx
        assert len(a.split("\n")) > 10


@pytest.mark.gpu
def test_both_clis_go_out_of_core_when_the_device_is_too_small(tmp_path, host_bin, monkeypatch):
    """--gpca-stream auto (the default) asks the device how much memory is free (gpca_get_device_memory) and walks the .bed out
    of core when matrix + workspace would not fit: pretending the device holds 16 MiB, both programs say so and still write
    the files of the resident run (within the TSV rounding: the streamed power iterations quantise per panel)."""
    from genomic_pca_amd.cli import main
    pre, z = _chr22_fileset(tmp_path)
    ld = tmp_path / "ld.txt"
    ld.write_text("22 1 500000000\n")
    common = ["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "6", "--gpca-panel-rows", "32768"]
    ref = str(tmp_path / "ref" / "run")
    assert main(["--out", ref] + common) == 0
    monkeypatch.setenv("GPCA_CLI_FREE_BYTES", str(16 << 20))
    out_c, out_p = str(tmp_path / "c" / "run"), str(tmp_path / "p" / "run")
    r = subprocess.run([host_bin, "--out", out_c] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "walking it out of core" in r.stderr, r.stderr
    assert main(["--out", out_p] + common) == 0
    num = lambda path, c: np.array([[float(x) for x in ln.split("\t")[c:]] for ln in open(path).read().strip().split("\n")[1:]])
    for sfx, c in ((".eigensnp.pca.tsv", 1), (".eigenvalues.tsv", 1), (".eigensnp.loadings.tsv", 3)):
        assert open(out_c + sfx).read() == open(out_p + sfx).read(), sfx
        a, b = num(out_c + sfx, c), num(ref + sfx, c)
        assert np.max(np.abs(a - b)) <= 5e-6 * max(1.0, np.max(np.abs(b))), sfx


@pytest.mark.gpu
@pytest.mark.parametrize("subset", [False, True])
def test_both_clis_multi_stage_eigensnp(tmp_path, host_bin, subset):
    """--gpca-eigensnp-local-stage through both command lines on the reference's chr22_subset50 genotypes with 9 LD blocks (the
    --eigensnp-* local / refine flags acting): the same stages, seeds and -- with a sample subset -- the same drawn samples in
    the C++ and the Python host, so the files agree to the TSV rounding; and they sit close to the one-stage default."""
    from genomic_pca_amd.cli import main
    pre, z = _chr22_fileset(tmp_path)
    ld = tmp_path / "ld.txt"
    edges = np.linspace(16_050_000, 16_050_000 + 25 * 120_000, 10).astype(int)
    ld.write_text("".join(f"22 {a} {b - 1}\n" for a, b in zip(edges[:-1], edges[1:])))
    common = ["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "5", "--eigensnp-seed", "3",
              "--eigensnp-components-per-block", "6", "--eigensnp-refine-passes", "2"]
    if subset:
        common += ["--eigensnp-subset-factor", "0.5", "--eigensnp-min-subset-size", "20", "--eigensnp-max-subset-size", "40"]
    ref = str(tmp_path / "ref" / "run")
    assert main(["--out", ref] + common) == 0                       # the one-stage default
    out_c, out_p = str(tmp_path / "c" / "run"), str(tmp_path / "p" / "run")
    r = subprocess.run([host_bin, "--out", out_c, "--gpca-eigensnp-local-stage"] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert main(["--out", out_p, "--gpca-eigensnp-local-stage"] + common) == 0
    num = lambda path, c: np.array([[float(x) for x in ln.split("\t")[c:]] for ln in open(path).read().strip().split("\n")[1:]])
    for sfx, c in ((".eigensnp.pca.tsv", 1), (".eigenvalues.tsv", 1), (".eigensnp.loadings.tsv", 3)):
        a, b, g0 = num(out_c + sfx, c), num(out_p + sfx, c), num(ref + sfx, c)
        assert a.shape == b.shape == g0.shape
        assert np.max(np.abs(a - b)) <= 1e-5 * max(1.0, np.max(np.abs(b))), sfx          # two hosts, one algorithm
    ev_m, ev_g = num(out_p + ".eigenvalues.tsv", 1)[:, 0], num(ref + ".eigenvalues.tsv", 1)[:, 0]
    # N = 64 real samples, a flat spectrum, local bases from as few as 32 samples: both runs approximate the same PCs
    assert np.all(np.abs(ev_m - ev_g)[:2] / ev_g[:2] < 0.05) and np.all(np.abs(ev_m - ev_g) / ev_g < 0.2)


@pytest.mark.gpu
def test_both_clis_local_stage_with_fewer_components_than_k_global(tmp_path, host_bin):
    """One LD block and components_per_block (3) < k_global (5): the local stage leaves 3 condensed features, so 3 PCs come back.
    Both hosts must take the column count from the result -- the C++ host once wrote k_global columns from 3-column arrays (wrong
    stride, reads past the end) -- and produce the same 3-column files."""
    from genomic_pca_amd.cli import main
    pre, z = _chr22_fileset(tmp_path)
    ld = tmp_path / "ld.txt"
    ld.write_text("22 1 500000000\n")
    common = ["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "5", "--eigensnp-seed", "3",
              "--eigensnp-components-per-block", "3", "--gpca-eigensnp-local-stage"]
    out_c, out_p = str(tmp_path / "c" / "run"), str(tmp_path / "p" / "run")
    r = subprocess.run([host_bin, "--out", out_c] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert main(["--out", out_p] + common) == 0
    num = lambda path, c: np.array([[float(x) for x in ln.split("\t")[c:]] for ln in open(path).read().strip().split("\n")[1:]])
    for sfx, c, cols in ((".eigensnp.pca.tsv", 1, 3), (".eigenvalues.tsv", 1, 1), (".eigensnp.loadings.tsv", 3, 3)):
        a, b = num(out_c + sfx, c), num(out_p + sfx, c)
        assert a.shape == b.shape and a.shape[1] == cols, (sfx, a.shape, b.shape)
        assert open(out_c + sfx).readline() == open(out_p + sfx).readline()           # same header (PC1..PC3)
        assert np.max(np.abs(a - b)) <= 1e-5 * max(1.0, np.max(np.abs(b))), sfx
    assert num(out_p + ".eigenvalues.tsv", 1).shape[0] == 3


@pytest.mark.gpu
def test_both_clis_pick_2bit_residency_for_a_wide_bed(tmp_path, host_bin, gpca, oracle):
    """--gpca-storage auto (the default): a .bed of >= 1 024 samples stays in its own 2-bit form on the device, a narrower one is
    decoded to int8 -- both hosts make the same choice (byte-identical files, and identical to the explicit setting)."""
    from genomic_pca_amd.cli import main
    M, N = 2500, 1100
    G = oracle.synth_genotypes(M, N, 6, gpca.synth_thresholds(M, 5, seed=6, fst=0.3))
    pre = str(tmp_path / "wide")
    gio.write_plink(pre, G, [f"s{i}" for i in range(N)], [f"rs{i}" for i in range(M)], ["7"] * M, list(range(1, M + 1)))
    ld = tmp_path / "ld.txt"
    ld.write_text("7 1 100000\n")
    common = ["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", str(ld), "--eigensnp-k-global", "4"]
    outs = {}
    for name, extra in (("auto", []), ("two", ["--gpca-storage", "2bit"]), ("eight", ["--gpca-storage", "int8"])):
        out = str(tmp_path / name / "run")
        assert main(["--out", out] + common + extra) == 0
        outs[name] = [open(out + sfx).read() for sfx in (".eigensnp.pca.tsv", ".eigenvalues.tsv", ".eigensnp.loadings.tsv")]
    out_c = str(tmp_path / "c" / "run")
    r = subprocess.run([host_bin, "--out", out_c] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert [open(out_c + sfx).read() for sfx in (".eigensnp.pca.tsv", ".eigenvalues.tsv", ".eigensnp.loadings.tsv")] == outs["auto"]
    assert outs["auto"] == outs["two"]                                # 1 100 samples: 2-bit
    # 2-bit rows run three digit planes (24-bit fixed point per column), int8 rows four: the same PCs to the 6th printed decimal or so
    num = lambda txt, c: np.array([[float(x) for x in ln.split("\t")[c:]] for ln in txt.strip().split("\n")[1:]])
    for a, b, c in zip(outs["two"], outs["eight"], (1, 1, 3)):
        assert a.split("\n")[0] == b.split("\n")[0]
        A, B = num(a, c), num(b, c)
        assert A.shape == B.shape and np.max(np.abs(A - B)) <= 5e-6 * max(1.0, np.max(np.abs(B)))


@pytest.mark.gpu
def test_cpp_vcf_workflow_equals_python_cli(tmp_path, host_bin, gpca, oracle):
    from genomic_pca_amd.cli import main
    M, N = 600, 48
    G = oracle.synth_genotypes(M, N, 8, gpca.synth_thresholds(M, 4, seed=8, fst=0.3))
    names = [f"S{i}" for i in range(N)]
    gt = {0: "0/0", 1: "0|1", 2: "1/1"}
    d = tmp_path / "vcfs"; d.mkdir()
    for ci, (lo, hi) in enumerate([(0, 350), (350, 600)]):
        opener = gzip.open if ci == 0 else open
        with opener(d / (f"chr{ci + 1}.vcf" + (".gz" if ci == 0 else "")), "wt") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
            for i in range(lo, hi):
                f.write(f"{ci + 1}\t{i + 1}\t.\tA\tC\t.\t.\t.\tGT:DP\t" + "\t".join(gt[int(v)] + ":5" for v in G[i]) + "\n")
    # (-k 38 with 48 samples: the reference clamps k to min(samples, variants) and adds 10 -- a 48-column sketch; -k 60 asks for more
    #  components than there are samples: clamped to 48, the oversampling to what is left)
    for kk, extra in (("4", []), ("4", ["--write-eigenvalues"]), ("38", ["--write-eigenvalues"]), ("60", ["--write-eigenvalues"]),
                      ("4", ["--write-eigenvalues", "--gpca-rfit-power-iters", "4"])):
        out_c, out_p = str(tmp_path / "c" / "v"), str(tmp_path / "p" / "v")
        common = ["--vcf-dir", str(d), "-k", kk, "--maf", "0.05", "--rfit-seed", "3"] + extra
        r = subprocess.run([host_bin, "--out", out_c] + common, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert main(["--out", out_p] + common) == 0
        for sfx in (".vcf.pca.tsv", ".eigenvalues.tsv"):
            assert open(out_c + sfx).read() == open(out_p + sfx).read(), sfx
        assert (len(open(out_c + ".eigenvalues.tsv").read().split("\n")) > 3) == bool(extra)


@pytest.mark.gpu
def test_both_clis_components_60(tmp_path, host_bin, gpca, oracle):
    """`--components 60` (missing #2): the reference clamps k to min(samples, variants) and adds 10 (main.rs:621-628, 636), a
    70-column sketch; both command lines run it (128 padded columns on the exact-integer path), write the same bytes, and the leading PCs
    are the oracle's."""
    from genomic_pca_amd.cli import main
    M, N, P = 900, 130, 6
    G = oracle.synth_genotypes(M, N, 5, gpca.synth_thresholds(M, P, seed=5, fst=0.3))
    names = [f"S{i}" for i in range(N)]
    gt = {0: "0/0", 1: "0|1", 2: "1/1"}
    d = tmp_path / "vcfs"; d.mkdir()
    with open(d / "chr1.vcf", "wt") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for i in range(M):
            f.write(f"1\t{i + 1}\t.\tA\tC\t.\t.\t.\tGT\t" + "\t".join(gt[int(v)] for v in G[i]) + "\n")
    out_c, out_p = str(tmp_path / "c" / "v"), str(tmp_path / "p" / "v")
    common = ["--vcf-dir", str(d), "--components", "60", "--maf", "0.0", "--rfit-seed", "3", "--write-eigenvalues"]
    r = subprocess.run([host_bin, "--out", out_c] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert main(["--out", out_p] + common) == 0
    for sfx in (".vcf.pca.tsv", ".eigenvalues.tsv"):
        assert open(out_c + sfx).read() == open(out_p + sfx).read(), sfx
    rows = [ln.split("\t") for ln in open(out_p + ".vcf.pca.tsv").read().strip().split("\n")]
    assert rows[0] == ["SampleID"] + [f"PC{i + 1}" for i in range(60)] and len(rows) == N + 1
    pcs = np.array([[float(x) for x in rw[1:]] for rw in rows[1:]])
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    keep = st["keep"].astype(bool)
    r_, b_ = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r_, b_, 60, 10, 2, seed=3)
    assert keep.sum() > 800
    ref = oracle.standardized_dense(G, N, r_, b_).T @ R["loadings"]                     # the VCF workflow writes PCA::transform(x) (main.rs:659)
    assert oracle.max_abs_dpc(pcs[:, :P - 1], ref[:, :P - 1]) < 1e-4 + 2e-6             # ({:.6} in the file)


@pytest.mark.gpu
def test_gpca_hpp_mirror_types(tmp_path, gpca, oracle):
    """include/gpca.hpp used directly (accessor pull API and its `Clone`, compute_pca over a union of LD blocks with the
    accessor restored afterwards, PCA::rfit/transform, the reference's argument errors): the numbers the Python mirror gives."""
    gpca.load()
    exe = str(tmp_path / "hpp_client")
    pkg = os.path.join(ROOT, "genomic_pca_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "hpp_client.cpp"), "-L" + pkg, "-lgpca", "-Wl,-rpath," + pkg, "-o", exe])
    M, N = 3000, 200
    G = oracle.synth_genotypes(M, N, 4, gpca.synth_thresholds(M, 6, seed=4, fst=0.3))
    G[::40] = 1                                                     # monomorphic rows leave the PCA
    G.tofile(tmp_path / "g.i8")
    out = subprocess.run([exe, str(tmp_path / "g.i8"), str(M), str(N)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = {r[0]: r[1:] for r in (ln.split("\t") for ln in out.stdout.strip().split("\n"))}
    with gpca.GpcaEngine() as e:
        e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6))
        acc = gpca.MicroarrayGenotypeAccessor(e)
        D = acc.num_pca_snps()
        assert rows["dims"] == [str(D), str(N)] and D < M
        blk = acc.get_standardized_snp_sample_block([0, 3, 5], [1, 0, 7, 2])
        assert np.array_equal(np.array([float(x) for x in rows["block"]], np.float32), blk.reshape(-1))
        b1 = gpca.LdBlockSpecification("a", list(range(D // 3)))
        b2 = gpca.LdBlockSpecification("b", list(range(D // 2, D, 2)))
        cfg = gpca.EigenSNPCoreAlgorithmConfig(target_num_global_pcs=4, random_seed=9)
        res, _ = gpca.EigenSNPCoreAlgorithm(cfg).compute_pca(acc, [b1, b2])
        used = len(b1.pca_snp_ids_in_block) + len(b2.pca_snp_ids_in_block)
        assert rows["used"] == [str(used), str(N), "4"] and rows["loadings"] == [str(used * 4)] and rows["restored"] == [str(D)]
        assert np.array_equal(np.array([float(x) for x in rows["eig"]]), res.final_principal_component_eigenvalues)
        assert np.array_equal(np.array([float(x) for x in rows["scores0"]], np.float32), res.final_sample_principal_component_scores[0])
    assert "out of range" in rows["range"][0] and rows["pull"] == ["-1"]
    model = gpca.PCA().rfit(G.T, 3, 10, 1)
    assert rows["pca"][:2] == ["3", str(N * 3)]
    assert np.array_equal(np.array([float(x) for x in rows["pca"][2:]]), model.explained_variance())
    assert "must be > 0" in rows["k0"][0] and "at least 2 samples" in rows["n1"][0] and "before rfit" in rows["unfitted"][0]
