"""CPU-only: pins the oracle (tests/ infrastructure) against known answers.

The reference holds no golden vectors for this path, so the pins are (a) published
known-answer vectors (Philox, Random123 kat_vectors), (b) closed-form answers derived by hand from the
reference's formulas (file:line cited), (c) the committed fixtures under tests/golden/."""
import json
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert [int(x) for x in oracle.philox([0, 0, 0, 0], [0, 0])] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in oracle.philox([0xffffffff] * 4, [0xffffffff] * 2)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_hwe_branches(oracle):
    # prepare.rs:1647-1650 total 0 -> 1.0 ; :1664-1667 monomorphic -> 1.0
    assert oracle.hwe_p(0, 0, 0) == 1.0
    assert oracle.hwe_p(50, 0, 0) == 1.0
    assert oracle.hwe_p(0, 0, 7) == 1.0
    # exact HWE proportions p=0.5, n=100: (25, 50, 25) -> chi2 = 0 -> p = 1
    assert oracle.hwe_p(25, 50, 25) == 1.0
    # no hets at p=0.5: expected (50,100,50) observed (100,0,100): chi2 = 50+100+50 = 200 -> p ~ 0 (1 - erf(10) == 0 in f64)
    assert oracle.hwe_p(100, 0, 100) == 0.0
    # hand value: (10, 50, 40): p1 = 70/200 = .35; E = (12.25, 45.5, 42.25); chi2 = .413265+.445055+.119822 = .978142
    chi = (10 - 12.25) ** 2 / 12.25 + (50 - 45.5) ** 2 / 45.5 + (40 - 42.25) ** 2 / 42.25
    assert oracle.hwe_p(10, 50, 40) == pytest.approx(1.0 - math.erf(math.sqrt(chi / 2)), rel=1e-14)
    # symmetric in the two homozygote classes
    assert oracle.hwe_p(40, 50, 10) == pytest.approx(oracle.hwe_p(10, 50, 40), rel=1e-14)


def test_snp_stats_known_rows(oracle):
    M = oracle.MISSING
    G = np.array([
        [0, 1, 2, 1, 0, 1, 2, 1],        # mean 1, SS = 4, var = 4/7
        [0, 0, 0, 0, 0, 0, 0, 0],        # monomorphic -> dropped (MAF / monomorphic)
        [2, 2, 2, 2, 2, 2, 2, 2],        # monomorphic alt
        [0, 1, M, 1, 0, 1, 2, 1],        # one missing: call rate 7/8
        [M, M, M, M, M, M, M, M],        # nothing valid
        [0, 0, 0, 0, 0, 0, 0, 1],        # MAF 1/16
    ], np.int8)
    st = oracle.snp_stats(G, min_call_rate=0.0, min_maf=0.0, max_hwe_p=1.0)
    assert st["keep"].tolist() == [1, 0, 0, 1, 0, 1]
    assert st["reason"].tolist() == [0, 4, 4, 0, 1 if False else 2, 0]
    assert st["counts"][0].tolist() == [8, 2, 4, 2]
    assert st["counts"][3].tolist() == [7, 2, 4, 1]
    assert st["mu"][0] == np.float32(1.0)
    assert st["sigma"][0] == np.float32(math.sqrt(4 / 7))
    # row 3: values 0,1,1,0,1,2,1: mean 6/7, SS = sum (v - 6/7)^2
    v = np.array([0, 1, 1, 0, 1, 2, 1], float)
    assert st["mu"][3] == np.float32(6 / 7)
    assert st["sigma"][3] == np.float32(math.sqrt(((v - 6 / 7) ** 2).sum() / 6))
    # thresholds: call rate (prepare.rs:1283-1284) and MAF (:1296-1299)
    st2 = oracle.snp_stats(G, min_call_rate=0.98, min_maf=0.1, max_hwe_p=1.0)
    assert st2["reason"].tolist() == [0, 3, 3, 1, 1, 3]
    # exact-rational cross-check of every kept row
    for i in np.nonzero(st["keep"])[0]:
        m, s = oracle.snp_sigma_exact(G[i])
        assert st["mu"][i] == np.float32(m)
        assert abs(float(st["sigma"][i]) - s) <= np.spacing(np.float32(s))


def test_snp_stats_hwe_filter(oracle):
    # 200 samples, no heterozygotes, p = 0.5 -> HWE p = 0 -> dropped when the filter is active (prepare.rs:1306-1311)
    G = np.array([[0] * 100 + [2] * 100, [0] * 50 + [1] * 100 + [2] * 50], np.int8)
    st = oracle.snp_stats(G, 200, 0.0, 0.0, 1e-6)
    assert st["keep"].tolist() == [0, 1] and st["reason"].tolist() == [5, 0]
    st = oracle.snp_stats(G, 200, 0.0, 0.0, 1.0)  # filter off (threshold == 1.0)
    assert st["keep"].tolist() == [1, 1]


def test_standardize_block_formula_and_error(oracle):
    G = np.array([[0, 1, 2, 1], [2, 2, 0, -127], [1, 1, 1, 1]], np.int8)
    mu = np.array([1.0, 1.3333334, 1.0], np.float32)
    sg = np.array([0.8164966, 1.1547005, 0.0], np.float32)
    out, err = oracle.standardize_block(G, mu, sg, [0, 2], [3, 0, 2])
    assert err is None
    rs = np.float32(1.0) / sg[0]
    bt = -mu[0] * rs
    # fma((f32) g, 1/sigma, -mu * (1/sigma))  (prepare.rs:1948-1949, 1988)
    exp0 = [np.float32(np.float64(g) * np.float64(rs) + np.float64(bt)) for g in (1, 0, 2)]
    assert out[0].tolist() == [float(x) for x in exp0]
    assert out[1].tolist() == [0.0, 0.0, 0.0]          # sigma < 1e-9 -> zeros (prepare.rs:1899-1945)
    out, err = oracle.standardize_block(G, mu, sg, [0, 1], [0, 3])
    assert err == (1, 1)                                # missing genotype -> error (prepare.rs:1909-1911)
    # empty request (prepare.rs:1848-1850)
    out, err = oracle.standardize_block(G, mu, sg, [], [0, 1])
    assert out.shape == (0, 2) and err is None


def test_rsvd_converges_to_exact_pca(oracle, gpca):
    """The restated randomized PCA against exact f64 PCA (pattern of the reference's tests/pca.py:81-141)
    on strongly structured data: eigenvalues to 1e-3, PCs to 2e-2 (q = 2 power iterations; convergence bound,
    not the kernel-parity bar)."""
    M, N, P, k = 6000, 300, 8, 6
    th = gpca.synth_thresholds(M, P, seed=11, fst=0.25)
    G = oracle.synth_genotypes(M, N, 11, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=1)
    E = oracle.exact_pca(G, N, r, b, k)
    assert np.max(np.abs(R["eigenvalues"] - E["eigenvalues"]) / E["eigenvalues"]) < 1e-3
    assert oracle.max_abs_dpc(R["scores"], E["scores"]) < 2e-2
    # f32 restatement (the timed CPU baseline) agrees with the f64 checker far inside the 1e-4 bar
    R32 = oracle.rsvd(G, N, r, b, k, 10, 2, seed=1, real="f32")
    assert oracle.max_abs_dpc(R32["scores"], R["scores"]) < 1e-5
    assert np.max(np.abs(R32["eigenvalues"] - R["eigenvalues"]) / R["eigenvalues"]) < 1e-5


def _structured_case(oracle, gpca, M=5000, N=400, P=6, seed=13):
    th = gpca.synth_thresholds(M, P, seed=seed, fst=0.3)
    G = oracle.synth_genotypes(M, N, seed, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    return G, N, r, b, P - 1          # P populations -> P - 1 structured PCs with clean gaps


def test_checker_shares_no_small_dense_code_with_the_port(oracle, gpca):
    """oracle.rsvd (LAPACK Householder QR + gesdd) against oracle.rsvd_port (the product's recipe: CholeskyQR2 + cyclic
    Jacobi, restated in C): same sketch, same products, independent factorisations -> agreement at rounding level.  A defect
    in the transcribed Jacobi / Cholesky would show up here instead of cancelling in the GPU parity tests."""
    G, N, r, b, ks = _structured_case(oracle, gpca)
    for k, ov, q in ((8, 10, 2), (5, 0, 0), (1, 3, 1), (20, 10, 2)):
        A = oracle.rsvd(G, N, r, b, k, ov, q, seed=3)
        Pt = oracle.rsvd_port(G, N, r, b, k, ov, q, seed=3)
        assert np.max(np.abs(A["eigenvalues"] - Pt["eigenvalues"]) / Pt["eigenvalues"]) < 1e-11
        assert np.max(np.abs(A["singular_values"] - Pt["singular_values"]) / Pt["singular_values"][0]) < 1e-11
        kk = min(k, ks)               # the structured PCs are well conditioned; noise PCs are compared through the subspace below
        assert oracle.max_abs_dpc(A["scores"][:, :kk], Pt["scores"][:, :kk]) < 1e-9
        assert oracle.max_abs_dpc(A["loadings"][:, :kk], Pt["loadings"][:, :kk]) < 1e-9
        PA = A["scores"] @ np.linalg.pinv(A["scores"]); PP = Pt["scores"] @ np.linalg.pinv(Pt["scores"])
        assert np.max(np.abs(PA - PP)) < 1e-8                                # same k-dimensional score subspace


def test_checker_against_scipy_svd_of_the_dense_matrix(oracle, gpca):
    """exact SVD (scipy.linalg.svd -> LAPACK gesdd) of the dense standardised matrix X = A^T.
    With q = 2 the structured PCs of the randomized PCA have converged to it; the noise PCs agree as a subspace bound."""
    import scipy.linalg
    G, N, r, b, ks = _structured_case(oracle, gpca)
    A = oracle.standardized_dense(G, N, r, b)                                 # M x N
    U, s, Vt = scipy.linalg.svd(A.T, full_matrices=False)                     # X = U diag(s) Vt, U: samples
    R = oracle.rsvd(G, N, r, b, 10, 10, 2, seed=7)
    # convergence bounds of q = 2 on this spectrum (measured: eigenvalues 1e-7 .. 5e-6, scores 4e-4, loadings 3e-5): this pins the
    # ALGORITHM's answer to the exact decomposition; kernel parity (1e-4 against the same-sketch checker) is a different bar
    assert np.max(np.abs(R["eigenvalues"][:ks] - s[:ks] ** 2 / (N - 1)) / (s[:ks] ** 2 / (N - 1))) < 2e-5
    assert oracle.max_abs_dpc(R["scores"][:, :ks], U[:, :ks] * s[:ks]) < 2e-3
    assert oracle.max_abs_dpc(R["loadings"][:, :ks], Vt[:ks].T) < 2e-4
    # all 10 eigenvalues are bounded by the exact ones (Rayleigh-Ritz) and within 20 % even in the noise bulk
    ex = s[:10] ** 2 / (N - 1)
    assert np.all(R["eigenvalues"] <= ex * (1 + 1e-9)) and np.max(np.abs(R["eigenvalues"] - ex) / ex) < 0.2
    # exact_pca (eigh of the Gram, the reference's tests/pca.py pattern) is the same decomposition
    E = oracle.exact_pca(G, N, r, b, 10)
    assert np.allclose(E["eigenvalues"], ex, rtol=1e-9)


def test_checker_against_sklearn_randomized_svd(oracle, gpca):
    """SURVEY.md 8c(3), the independent third opinion: sklearn.utils.extmath.randomized_svd (its own Gaussian sketch, its own
    LU/QR-normalised power iterations) on the dense standardised matrix, n_iter = 2, same oversampling."""
    from sklearn.utils.extmath import randomized_svd
    G, N, r, b, ks = _structured_case(oracle, gpca)
    X = oracle.standardized_dense(G, N, r, b).T                               # samples x variants
    U, s, Vt = randomized_svd(X, n_components=10, n_oversamples=10, n_iter=2, power_iteration_normalizer="QR", random_state=0)
    R = oracle.rsvd(G, N, r, b, 10, 10, 2, seed=7)
    assert np.max(np.abs(R["eigenvalues"][:ks] - s[:ks] ** 2 / (N - 1)) / R["eigenvalues"][:ks]) < 1e-5
    assert oracle.max_abs_dpc(R["scores"][:, :ks], U[:, :ks] * s[:ks]) < 2e-3          # (two different sketches, q = 2 each)
    assert oracle.max_abs_dpc(R["loadings"][:, :ks], Vt[:ks].T) < 1e-3
    # different sketches -> different noise PCs, but comparable captured variance
    assert abs(R["eigenvalues"].sum() - (s ** 2).sum() / (N - 1)) / R["eigenvalues"].sum() < 0.02


def test_synth16_known_answers(oracle, gpca):
    """Fast panel generator: SplitMix64 against its published outputs, then field order and threshold halves by hand."""
    assert [oracle.splitmix64_at(1234567, i) for i in range(5)] == \
        [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]
    assert oracle.splitmix64_at(0, 0) == 0xE220A8397B1DCDAF
    th = np.array([[(40000 << 16) | 10000, (65535 << 16) | 65535, 0]], np.uint32)          # pop 0: mixed; pop 1: always 2; pop 2: always 0
    G = oracle.synth16_genotypes(1, 70, 99, th, snp_offset=7)
    assert np.all(G[0, 16:32] == 2) and np.all(G[0, 32:48] == 0) and np.all(G[0, 64:70] == 2)   # pop(n) = (n / 16) % 3
    for n in (0, 3, 6, 13, 48, 63):                                                          # samples of population 0
        z = oracle.splitmix64_at(99, (7 << 26) + n // 4)
        u = (z >> (16 * (n % 4))) & 0xffff
        assert G[0, n] == (u < 40000) + (u < 10000)
    assert np.array_equal(oracle.synth16_genotypes(3, 50, 5, gpca.synth_thresholds16(3, 2, seed=5, snp_offset=11), snp_offset=11),
                          oracle.synth16_genotypes(14, 50, 5, gpca.synth_thresholds16(14, 2, seed=5), snp_offset=0)[11:])


def test_chr22_fixture_pins_the_checker(oracle):
    """tests/golden/chr22_subset50_120k.npz: 120 000 SNPs of the reference's own data/chr22_subset50.bed (BASELINE.json
    configs[2]'s data; 64 samples): the oracle reproduces the QC decisions and randomized-PCA outputs committed with it."""
    z = np.load(os.path.join(GOLD, "chr22_subset50_120k.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"])
    lut = np.array([2, -127, 1, 0], np.int8)                                  # count_a1 (prepare.rs:622-629)
    G = np.empty((rows.shape[0], rows.shape[1] * 4), np.int8)
    for s4 in range(4):
        G[:, s4::4] = lut[(rows >> (2 * s4)) & 3]
    G = G[:, :n]
    st = oracle.snp_stats(G, n, 0.98, 0.01, 1e-6)
    assert np.array_equal(st["keep"], z["keep"]) and np.array_equal(st["reason"], z["reason"])
    assert np.array_equal(st["mu"], z["mu"]) and np.array_equal(st["sigma"], z["sigma"])
    assert 20_000 < int(st["keep"].sum()) < 30_000
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, n, r, b, int(z["k"]), 10, 2, seed=int(z["seed"]))
    assert np.allclose(R["eigenvalues"], z["eigenvalues"], rtol=1e-9)
    assert oracle.max_abs_dpc(R["scores"][:, :5], z["scores"][:, :5]) < 1e-7


def test_rsvd_shard_invariance(oracle, gpca):
    """A row shard draws the Omega rows and genotypes of its global SNP indices."""
    M, N, P = 512, 64, 3
    th = gpca.synth_thresholds(M, P, seed=5)
    G = oracle.synth_genotypes(M, N, 5, th)
    G2 = oracle.synth_genotypes(M - 200, N, 5, gpca.synth_thresholds(M - 200, P, seed=5, snp_offset=200), snp_offset=200)
    assert np.array_equal(G[200:], G2)
    Om = oracle.omega(M, 30, seed=9)
    assert np.array_equal(Om[200:], oracle.omega(M - 200, 30, seed=9, snp_offset=200))
    assert abs(Om.mean()) < 0.02 and abs(Om.std() - 1.0) < 0.02


def test_golden_fixture(oracle):
    """Committed fixture (tests/golden/make_golden.py): the oracle reproduces its own pinned outputs."""
    path = os.path.join(GOLD, "synth_2048x192.npz")
    if not os.path.exists(path):
        pytest.skip("golden fixture not generated yet")
    z = np.load(path)
    G = z["G"]; N = G.shape[1]
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    assert np.array_equal(st["mu"], z["mu"]) and np.array_equal(st["sigma"], z["sigma"])
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, int(z["k"]), 10, 2, seed=int(z["seed"]))
    assert np.allclose(R["eigenvalues"], z["eigenvalues"], rtol=1e-10)
    assert oracle.max_abs_dpc(R["scores"], z["scores"]) < 1e-9


def test_checker_against_the_references_exact_pca_script(oracle):
    """The reference's own validation target (tests/pca.py:81-141, "Exact PCA Reference": centre only, GRM / kept, eigh) on the
    reference's own chr22_subset50 genotypes: the oracle's randomized PCA with mu = mean, sigma = 1 and 12 power iterations
    converges to it (eigenvalues 1e-9, leading PCs 1e-5) -- the only place where a number defined by the reference's own files
    pins this path's PCA output."""
    z = np.load(os.path.join(GOLD, "chr22_subset50_120k.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"])
    lut = np.array([2, -127, 1, 0], np.int8)
    G = np.empty((rows.shape[0], rows.shape[1] * 4), np.int8)
    for s4 in range(4):
        G[:, s4::4] = lut[(rows >> (2 * s4)) & 3]
    G = G[:, :n]
    st = oracle.snp_stats(G, n, 0.98, 0.01, 1e-6)
    E = oracle.exact_pca_centred_only(G, n, st["keep"], 6)
    assert E["kept"] == int(st["keep"].sum()) and not (G[st["keep"].astype(bool)] == -127).any()
    r, b = oracle.scale_shift(st["mu"], np.ones_like(st["sigma"]), st["keep"])
    R = oracle.rsvd(G, n, r, b, 6, 20, 12, seed=1)
    ev = R["eigenvalues"] * (n - 1) / E["kept"]
    assert np.max(np.abs(ev[:4] - E["evals"][:4]) / E["evals"][:4]) < 1e-8
    assert oracle.max_abs_dpc(R["scores"][:, :3] / np.sqrt(E["kept"]), E["pcs"][:, :3]) < 1e-5


def test_hwe_against_the_references_python_definition(oracle, gpca):
    """The reference holds a second, independent definition of the HWE test in its own validation script (tests/pca.py:54-66:
    allele frequency, expected counts, chi-squared statistic, 1 - scipy chi2.cdf(., 1)).  Where both definitions are regular (all
    three expected counts > 0) the Rust restatement (prepare.rs:1641-1745) and the library's host helper agree with it."""
    from scipy.stats import chi2 as chi2_dist

    def pca_py_hwe(a_aa, a_ab, a_bb):                       # restated from tests/pca.py:54-66
        n = a_aa + a_ab + a_bb
        p = (2 * a_aa + a_ab) / (2 * n)
        q = 1.0 - p
        exp = np.array([n * p * p, 2 * n * p * q, n * q * q])
        obs = np.array([a_aa, a_ab, a_bb])
        return 1.0 - chi2_dist.cdf(((obs - exp) ** 2 / exp).sum(), 1)
    rng = np.random.default_rng(0)
    for _ in range(300):
        n0, n1, n2 = (int(x) for x in rng.integers(1, 4000, 3))
        ref = pca_py_hwe(n0, n1, n2)
        assert abs(oracle.hwe_p(n0, n1, n2) - ref) <= 1e-12 + 1e-9 * ref
        assert gpca.GpcaEngine.hwe_chi_squared_p_value(n0, n1, n2) == oracle.hwe_p(n0, n1, n2)
    for n in ((25, 50, 25), (10, 50, 40), (640, 320, 40), (1, 2, 400)):
        assert abs(oracle.hwe_p(*n) - pca_py_hwe(*n)) <= 1e-12


def test_bed_fixture_against_the_references_own_decoder():
    """tests/disk.py:89-135 is the reference's own reading of the .bed bytes (2 bits per sample, LSB first, codes 00 -> 0,
    10 -> 1, 11 -> 2, 01 -> missing: the count of the OTHER allele).  The count-A1 dosages committed with the chr22 slice (what
    bed_reader's count_a1 read of prepare.rs:622-629 returns, and what the device decode is tested against) must be its mirror
    image: 2 - g, missing <-> missing."""
    z = np.load(os.path.join(GOLD, "chr22_subset50_slice.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"]); a1 = z["dosage_count_a1"]
    disk_py = np.array([0, 255, 1, 2], np.int16)            # indexed by the 2-bit code, as tests/disk.py maps it
    other = np.empty((rows.shape[0], rows.shape[1] * 4), np.int16)
    for s4 in range(4):
        other[:, s4::4] = disk_py[(rows >> (2 * s4)) & 3]
    other = other[:, :n]
    assert np.array_equal(np.where(a1 == -127, 255, 2 - a1.astype(np.int16)), other)
    assert (a1 == -127).sum() == (other == 255).sum() and set(np.unique(a1)) <= {-127, 0, 1, 2}


def test_box_muller_math_of_the_sketch_against_long_double_libm(tmp_path):
    """genomic_pca_amd/csrc/omega_math.h -- the transcendentals k_omega uses for the sketch's Box-Muller draw, written for a 33-bit integer
    argument -- compiled for the HOST (the same source the device compiles) and held to long-double libm over every binade edge, the
    ln table's interval edges and 2M random arguments: -2 ln u within 5e-16 relative, sin / cos(2 pi u) within 3e-16 absolute, the draw
    within 2.5e-15 absolute (|z| <= 6.67) -- as close to the oracle's libm (gpca_oracle.c:omega4) as libm is to itself.  The committed
    table (csrc/omega_table.inc) is the one the check program prints."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "omega_math_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "genomic_pca_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "omega_math_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    tab = subprocess.run([exe, "table"], capture_output=True, text=True, timeout=60).stdout
    inc = open(os.path.join(root, "genomic_pca_amd", "csrc", "omega_table.inc")).read()
    assert tab.strip() == "\n".join(inc.strip().split("\n")[1:]).strip()
