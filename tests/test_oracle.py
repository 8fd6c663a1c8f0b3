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
