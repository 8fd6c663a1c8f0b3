"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs.  Bars: bit-exact for integer/byte/index work and for the f32 standardise formula;
1e-4 (north_star) for the floating-point randomized PCA, sign-aligned, against the oracle's f64
restatement with the same seed."""
import numpy as np
import pytest

from conftest import engine_defaults

pytestmark = pytest.mark.gpu

TOL_PC = 1e-4   # BASELINE.json: max|dPC| vs ref, unit-norm sign-aligned
TOL_EV = 1e-4   # relative eigenvalue error


def _make(gpca, oracle, engine, M, N, P, seed, fst=0.1, missing=0.0):
    th = gpca.synth_thresholds(M, P, seed=seed, fst=fst)
    engine.synth_genotypes(M, N, seed, th)
    G = engine.download_genotypes_i8()
    return th, G


def test_native_library_is_the_one_loaded(gpca):
    lib = gpca.load()
    assert lib._name.endswith("genomic_pca_amd/libgpca.so")
    maps = open("/proc/self/maps").read()
    assert "libgpca.so" in maps


@pytest.mark.parametrize("M,N,P", [(1000, 333, 3), (257, 256, 5), (64, 1000, 2), (4096, 64, 3), (3, 7, 2)])
def test_synth_bit_exact(gpca, oracle, engine, M, N, P):
    th, G = _make(gpca, oracle, engine, M, N, P, seed=42)
    assert np.array_equal(G, oracle.synth_genotypes(M, N, 42, th))
    assert set(np.unique(G)) <= {0, 1, 2}


def test_synth_shard_offset(gpca, oracle, engine):
    M, N, P, off = 300, 130, 3, 777
    th = gpca.synth_thresholds(M, P, seed=7, snp_offset=off)
    engine.synth_genotypes(M, N, 7, th, snp_offset=off)
    assert np.array_equal(engine.download_genotypes_i8(), oracle.synth_genotypes(M, N, 7, th, snp_offset=off))


def test_upload_roundtrip_ragged(gpca, engine):
    rng = np.random.default_rng(0)
    for M, N in [(5, 1), (17, 255), (33, 257), (2, 1024)]:
        G = rng.integers(0, 3, size=(M, N), dtype=np.int8)
        engine.upload_genotypes_i8(G)
        assert engine.dims() == (M, N)
        assert np.array_equal(engine.download_genotypes_i8(), G)
    # strided source (ld > N)
    big = rng.integers(0, 3, size=(9, 700), dtype=np.int8)
    engine.upload_genotypes_i8(big[:, :300])
    assert np.array_equal(engine.download_genotypes_i8(), big[:, :300])


def _inject(G, rng, frac_missing=0.01):
    G = G.copy()
    m = rng.random(G.shape) < frac_missing
    G[m] = -127
    return G


@pytest.mark.parametrize("M,N", [(2000, 500), (300, 1023), (129, 64), (50, 2049)])
def test_snp_stats_parity(gpca, oracle, engine, M, N):
    rng = np.random.default_rng(M + N)
    th = gpca.synth_thresholds(M, 3, seed=3, fst=0.1)
    G = oracle.synth_genotypes(M, N, 3, th)
    G = _inject(G, rng, 0.02)
    G[0] = 0; G[1] = 2; G[2] = -127; G[3, : N // 2] = -127          # monomorphic / all-missing / low call rate
    G[4] = np.where(np.arange(N) % 2 == 0, 0, 2)                     # no hets -> HWE failure
    engine.upload_genotypes_i8(G)
    for qc in [gpca.QcConfig.none(), gpca.QcConfig(), gpca.QcConfig(0.9, 0.05, 1e-3)]:
        st = engine.snp_stats(qc)
        counts, reason = engine.snp_qc_detail()
        ref = oracle.snp_stats(G, N, qc.min_snp_call_rate, qc.min_snp_maf, qc.max_snp_hwe_p_value)
        assert np.array_equal(counts, ref["counts"])                 # integers: bit-exact
        assert np.array_equal(st["keep"], ref["keep"])
        assert np.array_equal(reason, ref["reason"])
        assert np.array_equal(st["mu"], ref["mu"])                   # exact integer sum / n, one rounding
        # sigma: device forms SS exactly from integers; the reference's two-pass f64 agrees to <= 1 f32 ulp
        d = np.abs(st["sigma"].astype(np.float64) - ref["sigma"].astype(np.float64))
        assert np.all(d <= np.spacing(ref["sigma"]))
        kept = np.nonzero(ref["keep"])[0]
        for i in kept[:40]:
            m, s = oracle.snp_sigma_exact(G[i, :N])
            assert st["mu"][i] == np.float32(m) and st["sigma"][i] == np.float32(s)   # bit-exact vs exact rational
        assert engine.num_pca_snps() == int(ref["keep"].sum())
        assert np.array_equal(engine.pca_snp_rows(), kept)


def test_snp_stats_weird_values(gpca, oracle, engine):
    """Bytes outside {0,1,2,-127} take the byte-wise path and keep prepare.rs:1267-1279 semantics."""
    rng = np.random.default_rng(5)
    G = rng.integers(0, 3, size=(40, 300), dtype=np.int8)
    G[3, 17] = 3; G[5, 0] = -1; G[7, 299] = 127; G[9, 100] = -128; G[11, 5] = 5; G[11, 6] = -127
    engine.upload_genotypes_i8(G)
    st = engine.snp_stats(gpca.QcConfig.none())
    counts, reason = engine.snp_qc_detail()
    ref = oracle.snp_stats(G, 300, 0.0, 0.0, 1.0)
    assert np.array_equal(counts, ref["counts"]) and np.array_equal(st["keep"], ref["keep"])
    assert np.array_equal(st["mu"], ref["mu"])
    assert np.all(np.abs(st["sigma"] - ref["sigma"]) <= np.spacing(ref["sigma"]))
    with pytest.raises(gpca.GpcaError) as e:
        engine.rsvd(2, 2, 1, 1)
    assert e.value.status in (-5, -9)


def test_standardize_block_bit_exact(gpca, oracle, engine):
    M, N = 600, 257
    th, G = _make(gpca, oracle, engine, M, N, 3, seed=9)
    st = engine.snp_stats(gpca.QcConfig(0.0, 0.05, 1.0))
    rows = engine.pca_snp_rows()
    acc = gpca.MicroarrayGenotypeAccessor(engine)
    assert acc.num_pca_snps() == len(rows) and acc.num_qc_samples() == N
    rng = np.random.default_rng(1)
    for ns, nj in [(1, 1), (7, 257), (len(rows), 3), (200, 100)]:
        sid = rng.permutation(len(rows))[:ns]
        cid = rng.permutation(N)[:nj]
        out = acc.get_standardized_snp_sample_block(sid, cid)
        ref, err = oracle.standardize_block(G, st["mu"], st["sigma"], rows[sid], cid)
        assert err is None and out.dtype == np.float32
        assert np.array_equal(out, ref)                              # fma formula: bit-exact
    assert acc.get_standardized_snp_sample_block([], [0, 1]).shape == (0, 2)   # prepare.rs:1848-1850
    with pytest.raises(gpca.GpcaError):
        acc.get_standardized_snp_sample_block([len(rows)], [0])


def test_standardize_block_missing_is_hard_error(gpca, oracle, engine):
    G = np.array([[0, 1, 2, 1, 0, 2], [0, 1, -127, 1, 2, 2], [1, 1, 0, 2, 0, 1]], np.int8)
    engine.upload_genotypes_i8(G)
    engine.snp_stats(gpca.QcConfig(0.5, 0.0, 1.0))
    with pytest.raises(gpca.GpcaError) as e:
        engine.standardize_block([0, 1, 2], [0, 2, 4])
    assert e.value.status == -5
    # message wording of prepare.rs:1910-1911
    assert "Unexpected missing genotype (-127i8) in SnpBlockData for PCA SNP ID 1 (original BIM index 1), requested sample index 2" in e.value.message
    assert np.array_equal(engine.standardize_block([0, 2], [0, 2, 4]),
                          oracle.standardize_block(G, *[engine.snp_stats(gpca.QcConfig(0.5, 0.0, 1.0))[k] for k in ("mu", "sigma")], [0, 2], [0, 2, 4])[0])
    with pytest.raises(gpca.GpcaError) as e:     # rsvd refuses kept SNPs with missing genotypes, like the reference's accessor
        engine.rsvd(1, 1, 1, 1)
    assert e.value.status == -5


def _rsvd_case(gpca, oracle, engine, M, N, P, k, seed, fst, oversample=10, q=2):
    th, G = _make(gpca, oracle, engine, M, N, P, seed=seed, fst=fst)
    st = engine.snp_stats(gpca.QcConfig.none())
    ref_st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(ref_st["mu"], ref_st["sigma"], ref_st["keep"])
    engine.rsvd(k, oversample, q, seed=seed)
    R = oracle.rsvd(G, N, r, b, k, oversample, q, seed=seed)
    return G, r, b, R


@pytest.mark.parametrize("M,N,P,k", [(4096, 512, 12, 8), (20000, 1000, 16, 10), (3000, 1500, 10, 6), (999, 257, 8, 4)])
def test_rsvd_parity(gpca, oracle, engine, M, N, P, k):
    G, r, b, R = _rsvd_case(gpca, oracle, engine, M, N, P, k, seed=1, fst=0.2)
    ev = engine.eigenvalues(); sc = engine.scores(); sc64 = engine.scores(f64=True); ld = engine.loadings()
    assert np.max(np.abs(ev - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
    assert oracle.max_abs_dpc(sc64, R["scores"]) < TOL_PC
    assert oracle.max_abs_dpc(sc.astype(np.float64), R["scores"]) < TOL_PC
    assert oracle.max_abs_dpc(ld.astype(np.float64), R["loadings"]) < TOL_PC
    # absolute scale too (scores = V s, loadings unit norm)
    al = oracle.sign_align(sc64, R["scores"])
    assert np.max(np.abs(al - R["scores"])) < 1e-4 * np.max(np.abs(R["scores"]))
    assert np.allclose(np.linalg.norm(ld.astype(np.float64), axis=0), 1.0, atol=1e-5)
    # sign convention: largest |score| of each PC is positive
    assert np.all(sc64[np.abs(sc64).argmax(axis=0), np.arange(k)] > 0)
    sv = engine.singular_values()
    assert np.allclose(sv[:k] ** 2 / (N - 1), ev, rtol=1e-12)
    assert np.all(np.diff(sv) <= 1e-9 * sv[0])


def test_rsvd_l64_path(gpca, oracle, engine):
    """k = 40 -> l = 50 -> two 32-column MFMA tiles (config C5's shape class)."""
    G, r, b, R = _rsvd_case(gpca, oracle, engine, 6000, 700, 48, 40, seed=3, fst=0.3)
    ev = engine.eigenvalues()
    assert np.max(np.abs(ev - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
    # the trailing PCs of this small case sit in the noise bulk: compare the structured ones
    assert oracle.max_abs_dpc(engine.scores(f64=True)[:, :20], R["scores"][:, :20]) < TOL_PC


@pytest.mark.parametrize("store", ["int8", "2bit", "2bit4"])
@pytest.mark.parametrize("M,N,P,k", [(8000, 900, 60, 90), (5000, 700, 40, 60)])
def test_rsvd_wide_sketch(gpca, oracle, store, M, N, P, k):
    """Sketches wider than 64 columns: k = 90 -> l = 100 and k = 60 -> l = 70, both padded to 128 columns = four 32-column GEMM blocks and
    the any-L helpers of wide_sketch.hip (Gram, Cholesky + inverse in global memory, right multiplications).  The reference clamps k only
    to min(samples, variants) and always adds 10 (main.rs:621-628, 636), so -k 60 is an ordinary call there; it was refused here up to
    round 3.  Bars: the oracle's, on every structured PC; the orthonormality of what comes back on all k."""
    from genomic_pca_amd import _lib
    th = gpca.synth_thresholds(M, P, seed=7, fst=0.3)
    G = oracle.synth_genotypes(M, N, 7, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=7)
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_INT8 if store == "int8" else _lib.STORE_2BIT,
                         digit_planes=4 if store == "2bit4" else 0) as e:
        e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, 10, 2, seed=7)
        ev, sc, ld, sv = e.eigenvalues(), e.scores(f64=True), e.loadings().astype(np.float64), e.singular_values()
        tr = e.transform()
    assert ev.shape == (k,) and sc.shape == (N, k) and ld.shape == (M, k) and sv.shape == (k + 10,)
    assert np.max(np.abs(ev - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
    ns = P - 1                                                       # the population PCs; the rest sit in the noise bulk
    assert oracle.max_abs_dpc(sc[:, :ns], R["scores"][:, :ns]) < TOL_PC
    assert oracle.max_abs_dpc(ld[:, :ns], R["loadings"][:, :ns]) < TOL_PC
    gram = sc.T @ sc
    assert np.max(np.abs(gram - np.diag(np.diag(gram)))) < 1e-8 * gram[0, 0]          # scores orthogonal, norms = singular values
    assert np.max(np.abs(np.sqrt(np.diag(gram)) - sv[:k]) / sv[:k]) < 1e-8
    assert np.max(np.abs(ld.T @ ld - np.eye(k))) < 1e-5                                # loadings orthonormal (f32 storage)
    assert oracle.max_abs_dpc(tr[:, :ns], sc[:, :ns]) < 1e-2                           # PCA::transform on the fitted matrix
    with gpca.GpcaEngine(precision=_lib.PREC_F32_MFMA) as e:                           # the f32 path holds 64 columns: a clear refusal, not garbage
        e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig.none())
        with pytest.raises(gpca.GpcaError, match="wider than 64 columns"):
            e.rsvd(k, 10, 2, seed=7)
        with pytest.raises(gpca.GpcaError, match="<= 128"):
            e.rsvd(120, 10, 2, seed=7)


@pytest.mark.parametrize("planes", [3, 4])
def test_three_planes_under_a_wide_maf_spectrum(gpca, oracle, planes):
    """three digit planes are the default on 2-bit rows, and the sketch's planes are scaled by the analytic bound 6.67 max r over
    ALL kept rows.  With rare variants in the PCA (MAF down to the clap default's QC floor, 0.01) r = 1/sigma spans ~5x, so the typical
    row sits several bits below the scale of the rarest one.  A realistic spectrum -- ancestral frequencies log-uniform on [0.012, 0.5],
    half of the SNPs below 8 % -- through the default QC: the three-plane engine must still hold the oracle's bar (recorded: see the
    assertion message), as the four-plane form does."""
    from genomic_pca_amd import _lib
    M, N, P, k = 30000, 1200, 6, 10
    rng = np.random.default_rng(11)
    p_anc = np.exp(rng.uniform(np.log(0.012), np.log(0.5), M))
    fst = 0.1
    a = p_anc * (1 - fst) / fst; bb = (1 - p_anc) * (1 - fst) / fst
    p_pop = np.clip(rng.beta(a[:, None], bb[:, None], size=(M, P)), 0.0, 1.0)
    th = np.minimum(np.floor(p_pop * 4294967296.0), 4294967295.0).astype(np.uint32)
    G = oracle.synth_genotypes(M, N, 11, th)
    qc = (0.98, 0.01, 1e-6)
    st = oracle.snp_stats(G, N, *qc)
    keep = st["keep"].astype(bool)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    spread = r[keep].max() / r[keep].min()
    assert keep.sum() > 0.6 * M and spread > 4.0, (keep.sum(), spread)
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=2)
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT, digit_planes=planes) as e:
        e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig(*qc)); e.rsvd(k, 10, 2, seed=2)
        dsc = oracle.max_abs_dpc(e.scores(f64=True)[:, :P - 1], R["scores"][:, :P - 1])
        dld = oracle.max_abs_dpc(e.loadings().astype(np.float64)[:, :P - 1], R["loadings"][keep][:, :P - 1])
        dev = float(np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]))
    msg = f"planes {planes}: r spread {spread:.1f}x, max|dPC| scores {dsc:.2e} loadings {dld:.2e}, eigenvalues {dev:.2e}"
    print(msg)
    assert dsc < (1e-5 if planes == 3 else 1e-7) and dld < 1e-4 and dev < (1e-5 if planes == 3 else 1e-7), msg


def test_rsvd_vs_exact_pca(gpca, oracle, engine):
    """Converged answer: exact f64 PCA (reference's own cross-check pattern, tests/pca.py:81-141)."""
    G, r, b, R = _rsvd_case(gpca, oracle, engine, 8000, 400, 8, 6, seed=2, fst=0.3)
    E = oracle.exact_pca(G, 400, r, b, 6)
    assert np.max(np.abs(engine.eigenvalues() - E["eigenvalues"]) / E["eigenvalues"]) < 1e-3
    assert oracle.max_abs_dpc(engine.scores(f64=True), E["scores"]) < 2e-2


def test_rsvd_with_qc_dropped_snps(gpca, oracle, engine):
    M, N = 3000, 400
    th = gpca.synth_thresholds(M, 6, seed=4, fst=0.2)
    G = oracle.synth_genotypes(M, N, 4, th)
    G[::7] = 0                                   # monomorphic rows are dropped by QC
    G[5::11, :3] = -127                          # low call rate rows dropped at 0.999
    engine.upload_genotypes_i8(G)
    qc = gpca.QcConfig(0.999, 0.02, 1e-6)
    st = engine.snp_stats(qc)
    ref = oracle.snp_stats(G, N, qc.min_snp_call_rate, qc.min_snp_maf, qc.max_snp_hwe_p_value)
    assert np.array_equal(st["keep"], ref["keep"]) and 0 < ref["keep"].sum() < M
    r, b = oracle.scale_shift(ref["mu"], ref["sigma"], ref["keep"])
    engine.rsvd(5, 10, 2, seed=8)
    R = oracle.rsvd(G, N, r, b, 5, 10, 2, seed=8)
    kept = np.nonzero(ref["keep"])[0]
    ld = engine.loadings()
    assert ld.shape == (len(kept), 5)            # D x K over PCA SNPs only (main.rs:407)
    assert oracle.max_abs_dpc(ld.astype(np.float64), R["loadings"][kept]) < TOL_PC
    assert oracle.max_abs_dpc(engine.scores(f64=True), R["scores"]) < TOL_PC
    assert np.max(np.abs(engine.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV


def test_transform_matches_projection(gpca, oracle, engine):
    """PCA::transform (main.rs:659): scores = standardised X * rotation."""
    G, r, b, R = _rsvd_case(gpca, oracle, engine, 5000, 300, 8, 5, seed=6, fst=0.3)
    ld = engine.loadings().astype(np.float64)
    tr = engine.transform()
    A = oracle.standardized_dense(G, 300, r, b)
    ref = A.T @ ld
    assert np.max(np.abs(tr - ref)) < 1e-4 * np.max(np.abs(ref))
    # and it is the rSVD's own scores up to one more power-iteration's worth of convergence
    assert oracle.max_abs_dpc(tr, engine.scores(f64=True)) < 1e-2


def test_pca_class_mirror(gpca, oracle):
    """PCA::new / rfit / transform call pattern of main.rs:602,648-660 (x = samples x variants)."""
    M, N = 2500, 200
    th = gpca.synth_thresholds(M, 6, seed=12, fst=0.3)
    G = oracle.synth_genotypes(M, N, 12, th)
    x = G.T.astype(np.float64)                                  # build_matrix orientation (vcf.rs:329-342)
    model = gpca.PCA()
    model.rfit(x, 4, 10, 1, None)
    pcs = model.transform(x)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, 4, 10, 2, seed=1)
    A = oracle.standardized_dense(G, N, r, b)
    ref = A.T @ R["loadings"]
    assert pcs.shape == (N, 4)
    assert oracle.max_abs_dpc(pcs, ref) < TOL_PC
    with pytest.raises(ValueError):
        gpca.PCA().rfit(x, 0)                                    # main.rs:607-609
    with pytest.raises(ValueError):
        gpca.PCA().rfit(x[:1], 2)                                # main.rs:614-616
    model.rfit(x[:, :3], 10, 10, 1)                              # k clamped to min(n, m) (main.rs:621-628)
    assert model.transform().shape == (N, 3)
    y = x.copy(); y[7, 11] = 0.5
    with pytest.raises(ValueError, match="whole number"):        # dosage codes only: nothing is rounded behind the caller's back
        gpca.PCA().rfit(y, 4)
    y[7, 11] = 3.0
    with pytest.raises(gpca.GpcaError) as ei:                    # a whole number that is no dosage: the device's own check
        gpca.PCA().rfit(y, 4)
    assert ei.value.status == -9
    # a sketch as wide as the sample count (k + 10 >= N): the centred matrix has rank N - 1, the call still succeeds
    small = gpca.PCA().rfit(x[:12], 4, 10, 1)
    assert small.transform().shape == (12, 4) and np.all(np.isfinite(small.transform()))


def test_eigensnp_mirror(gpca, oracle, engine):
    M, N = 3000, 256
    th, G = _make(gpca, oracle, engine, M, N, 8, seed=21, fst=0.3)
    engine.snp_stats(gpca.QcConfig())
    acc = gpca.MicroarrayGenotypeAccessor(engine)
    cfg = gpca.EigenSNPCoreAlgorithmConfig(target_num_global_pcs=5)
    out, diag = gpca.EigenSNPCoreAlgorithm(cfg).compute_pca(acc, [gpca.LdBlockSpecification("1:1-500000000", list(range(acc.num_pca_snps())))])
    assert out.final_sample_principal_component_scores.shape == (N, 5) and out.final_sample_principal_component_scores.dtype == np.float32
    assert out.final_principal_component_eigenvalues.shape == (5,) and out.final_principal_component_eigenvalues.dtype == np.float64
    assert out.final_snp_principal_component_loadings.shape == (acc.num_pca_snps(), 5)
    ref = oracle.snp_stats(G, N, 0.98, 0.01, 1e-6)
    r, b = oracle.scale_shift(ref["mu"], ref["sigma"], ref["keep"])
    R = oracle.rsvd(G, N, r, b, 5, 10, 2, seed=2025)
    assert oracle.max_abs_dpc(out.final_sample_principal_component_scores.astype(np.float64), R["scores"]) < TOL_PC


def test_errors_and_state(gpca, engine):
    with pytest.raises(gpca.GpcaError) as e:
        engine.snp_stats()
    assert e.value.status == -7
    engine.upload_genotypes_i8(np.random.default_rng(0).integers(0, 3, size=(50, 40), dtype=np.int8))
    with pytest.raises(gpca.GpcaError) as e:
        engine.rsvd(2)
    assert e.value.status == -7                   # stats first
    engine.snp_stats()
    with pytest.raises(gpca.GpcaError):
        engine.rsvd(0)                            # main.rs:607-609
    with pytest.raises(gpca.GpcaError):
        engine.rsvd(60, 10)                       # l > 64 on the f32 path (and l > N)
    with pytest.raises(gpca.GpcaError):
        engine.rsvd(35, 10)                       # l > N
    with pytest.raises(ValueError):
        engine.upload_genotypes_i8(np.zeros((3, 3), np.float32))


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit"), ("f32", "int8"), ("f32", "2bit")])
def test_degenerate_inputs(gpca, oracle, prec, store):
    """The smallest and the emptiest inputs on every path: the reference's own messages where it has one (main.rs:607-619:
    no variant left / fewer than 2 samples), an argument error when the sketch is wider than the matrix, and the right answer
    for a 2 x 3 and a 5 x 3 matrix (rank 2: the third eigenvalue is zero to rounding)."""
    from genomic_pca_amd import _lib
    kw = dict(precision=_lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA,
              storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8)
    rng = np.random.default_rng(1)

    def run(G, k, ov):
        with gpca.GpcaEngine(**kw) as e:
            e.upload_genotypes_i8(G)
            st = e.snp_stats(gpca.QcConfig.none())
            ref = oracle.snp_stats(G, G.shape[1], 0.0, 0.0, 1.0)
            assert np.array_equal(st["keep"], ref["keep"]) and np.array_equal(st["mu"], ref["mu"])
            e.rsvd(k, ov, 2, 1)
            return e.eigenvalues(), e.scores(f64=True), ref
    with pytest.raises(gpca.GpcaError, match="at least 1 variant"):
        run(np.ones((300, 50), np.int8), 2, 2)                       # every SNP monomorphic: nothing left for the PCA
    for M, N in ((1, 1), (5, 1)):
        with pytest.raises(gpca.GpcaError, match="at least 2 samples"):
            run(rng.integers(0, 3, size=(M, N), dtype=np.int8), 1, 0)
    for M, N, k, ov in ((5, 3, 2, 5), (3, 40, 5, 0)):
        with pytest.raises(gpca.GpcaError, match="exceeds min"):
            run(rng.integers(0, 3, size=(M, N), dtype=np.int8), k, ov)
    for G, k in ((np.array([[0, 1, 2], [2, 0, 1]], np.int8), 2), (rng.integers(0, 3, size=(5, 3), dtype=np.int8), 3)):
        ev, sc, ref = run(G, k, 0)
        r, b = oracle.scale_shift(ref["mu"], ref["sigma"], ref["keep"])
        E = oracle.exact_pca(G, G.shape[1], r, b, k)                 # the sketch spans the whole sample space: the exact answer
        assert np.allclose(ev[:2], E["eigenvalues"][:2], rtol=1e-5)
        assert oracle.max_abs_dpc(sc[:, :2], E["scores"][:, :2]) < 1e-4
        if k == 3:
            assert abs(ev[2]) < 1e-6 * ev[0]                          # centred rows: rank <= N - 1


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit"), ("f32", "int8")])
def test_handle_reuse_across_widths_and_shapes(gpca, oracle, prec, store):
    """One handle, many calls: sketch widths 30 -> 50 -> 15 -> 30 (workspaces and launch plans are re-made as l crosses 32),
    then a different matrix shape on the same handle, then the first one again.  Every call gives the bits a fresh handle
    gives -- nothing of an earlier call (column halves, partial sums, digit planes, status flags) leaks into a later one."""
    from genomic_pca_amd import _lib
    kw = dict(precision=_lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA,
              storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8)
    shapes = [(6000, 700, 48, 1), (1500, 2300, 48, 2)]
    Gs = [oracle.synth_genotypes(M, N, sd, gpca.synth_thresholds(M, P, seed=sd, fst=0.3)) for M, N, P, sd in shapes]

    def fresh(G, k, seed):
        with gpca.GpcaEngine(**kw) as e:
            e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6)); e.rsvd(k, 10, 2, seed=seed)
            return e.eigenvalues(), e.scores(f64=True), e.loadings(), e.transform()
    want = {(g, k, sd): fresh(Gs[g], k, sd) for g, k, sd in ((0, 20, 1), (0, 40, 1), (0, 5, 7), (1, 20, 1), (1, 40, 3))}
    with gpca.GpcaEngine(**kw) as e:
        for g, k, sd in ((0, 20, 1), (0, 40, 1), (0, 5, 7), (0, 20, 1), (1, 40, 3), (1, 20, 1), (0, 40, 1), (0, 5, 7)):
            if e.dims() != Gs[g].shape:
                e.upload_genotypes_i8(Gs[g]); e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6))
            e.rsvd(k, 10, 2, seed=sd)
            got = (e.eigenvalues(), e.scores(f64=True), e.loadings(), e.transform())
            for a, b in zip(got, want[(g, k, sd)]):
                assert np.array_equal(a, b), (g, k, sd)


def test_allreduce_hook_two_shards_one_gpu(gpca, oracle):
    """N>1 exchange step on one GPU: two engines each hold a row shard; the host hook sums their sketches.
    Sharded result must equal the unsharded one (same Omega rows via snp_offset)."""
    import threading
    M, N, P, k = 4000, 384, 8, 6
    th = gpca.synth_thresholds(M, P, seed=31, fst=0.3)
    G = oracle.synth_genotypes(M, N, 31, th)
    from genomic_pca_amd import _lib
    mk = lambda: gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_INT8)     # the headline path, named explicitly
    full = mk(); full.upload_genotypes_i8(G); full.snp_stats(); full.rsvd(k, 10, 2, seed=5)
    ref_scores, ref_ev, ref_ld = full.scores(f64=True), full.eigenvalues(), full.loadings(); full.close()
    world = 2
    spans = [gpca.shard_rows(M, world, r) for r in range(world)]
    barrier = threading.Barrier(world); bufs = [None] * world; res = [None] * world

    def run(rank):
        a, b_ = spans[rank]
        e = mk(); e.upload_genotypes_i8(G[a:b_]); e.snp_stats()

        def hook(buf):
            bufs[rank] = buf.copy(); barrier.wait()
            buf[:] = sum(bufs[r] for r in range(world)); barrier.wait()
        e.set_allreduce_hook(hook, world, rank, a)
        e.rsvd(k, 10, 2, seed=5)
        res[rank] = (e.scores(f64=True), e.eigenvalues(), e.loadings()); e.close()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])   # replicated results identical
    # each shard quantises its rows of T' against its OWN column maximum (no extra exchange), so the 28-bit fixed-point
    # rounding differs from the unsharded run at the 1e-9 level; everything else is exact integers
    assert np.max(np.abs(res[0][1] - ref_ev) / ref_ev) < 5e-8
    assert oracle.max_abs_dpc(res[0][0], ref_scores) < 1e-7
    ld = np.concatenate([res[0][2], res[1][2]], axis=0)
    assert oracle.max_abs_dpc(ld.astype(np.float64), ref_ld.astype(np.float64)) < 1e-6


def test_rccl_world1(gpca, oracle, engine):
    """RCCL path with a 1-rank communicator: same answer as no communicator."""
    M, N = 2000, 300
    th, G = _make(gpca, oracle, engine, M, N, 6, seed=13, fst=0.3)
    engine.snp_stats(); engine.rsvd(4, 10, 2, seed=1)
    ev0, sc0 = engine.eigenvalues(), engine.scores(f64=True)
    uid = gpca.GpcaEngine.comm_unique_id()
    assert len(uid) == 128
    engine.comm_init(1, 0, uid, 0)
    engine.rsvd(4, 10, 2, seed=1)
    assert np.array_equal(engine.eigenvalues(), ev0) and np.array_equal(engine.scores(f64=True), sc0)


def test_timings_exposed(gpca, oracle, engine):
    th, G = _make(gpca, oracle, engine, 3000, 512, 4, seed=2)
    engine.snp_stats(fetch=False)
    engine.rsvd(10, 10, 2, seed=1)
    assert engine.timings() == {}                      # off by default: a long-running host accumulates nothing
    engine.enable_timings(True); engine.reset_timings()
    engine.rsvd(10, 10, 2, seed=1)
    t = engine.timings()
    assert t["gemm_GQ"]["launches"] == 3 and t["gemm_GtT"]["launches"] == 3
    assert t["gemm_GQ"]["flops"] == pytest.approx(3 * 2.0 * 3000 * 512 * 20)
    assert t["gemm_GQ"]["total_ms"] > 0


# ------------------------------------------------------------------------------------------------
# committed golden fixtures (tests/golden/make_golden.py)
# ------------------------------------------------------------------------------------------------
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_synth_fixture(gpca, oracle, engine):
    z = np.load(os.path.join(GOLD, "synth_2048x192.npz"))
    G = z["G"]; M, N = G.shape; k = int(z["k"]); seed = int(z["seed"])
    engine.synth_genotypes(M, N, seed, z["thresh"])
    assert np.array_equal(engine.download_genotypes_i8(), G)            # generator: bit-exact vs the pinned bytes
    st = engine.snp_stats(gpca.QcConfig.none())
    counts, _ = engine.snp_qc_detail()
    assert np.array_equal(counts, z["counts"]) and np.array_equal(st["mu"], z["mu"]) and np.array_equal(st["keep"], z["keep"])
    assert np.all(np.abs(st["sigma"] - z["sigma"]) <= np.spacing(z["sigma"]))
    engine.rsvd(k, 10, 2, seed=seed)
    assert np.max(np.abs(engine.eigenvalues() - z["eigenvalues"]) / z["eigenvalues"]) < TOL_EV
    assert oracle.max_abs_dpc(engine.scores(f64=True), z["scores"]) < TOL_PC
    assert oracle.max_abs_dpc(engine.loadings().astype(np.float64), z["loadings"]) < TOL_PC


def test_bed2bit_decode_reference_fixture(gpca, oracle, engine):
    """PLINK 2-bit decode on a slice of the reference's own data/chr22_subset50.bed (64 samples),
    count_a1 semantics of prepare.rs:622-629; byte layout per the reference's tests/disk.py:89-135."""
    z = np.load(os.path.join(GOLD, "chr22_subset50_slice.npz"))
    rows = z["bed_rows"]; n = int(z["n_samples"]); ref = z["dosage_count_a1"]
    engine.upload_bed2bit(rows, n)
    assert engine.dims() == (rows.shape[0], n)
    assert np.array_equal(engine.download_genotypes_i8(), ref)
    for qc in (gpca.QcConfig.none(), gpca.QcConfig()):
        st = engine.snp_stats(qc)
        counts, reason = engine.snp_qc_detail()
        o = oracle.snp_stats(ref, n, qc.min_snp_call_rate, qc.min_snp_maf, qc.max_snp_hwe_p_value)
        assert np.array_equal(counts, o["counts"]) and np.array_equal(st["keep"], o["keep"]) and np.array_equal(reason, o["reason"])
        assert np.array_equal(st["mu"], o["mu"]) and np.all(np.abs(st["sigma"] - o["sigma"]) <= np.spacing(o["sigma"]))
    # synthetic ragged BED: N not a multiple of 4, all four codes present
    rng = np.random.default_rng(3)
    for n2 in (1, 5, 63, 130, 257):
        b = rng.integers(0, 256, size=(37, (n2 + 3) // 4), dtype=np.uint8)
        lut = np.array([2, -127, 1, 0], np.int8)
        exp = np.empty((37, b.shape[1] * 4), np.int8)
        for s in range(4):
            exp[:, s::4] = lut[(b >> (2 * s)) & 3]
        engine.upload_bed2bit(b, n2)
        assert np.array_equal(engine.download_genotypes_i8(), exp[:, :n2])


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] at full size (1M SNPs x 10k samples): size-independent properties
# ------------------------------------------------------------------------------------------------
_FULL = {}


def _full_size_case(gpca, oracle, prec, store):
    """One (precision, residency) pair at BASELINE.json configs[1]'s full size: every size-independent check, and the structured
    part of the result (eigenvalues, two leading PCs, their loadings) for the cross-path comparisons.  Cached per process, so the
    comparisons below do not depend on the order pytest runs the parameters in."""
    if (prec, store) in _FULL:
        return _FULL[(prec, store)]
    from genomic_pca_amd import _lib
    M, N, k, seed = 1_000_000, 10_000, 20, 1
    th = gpca.synth_thresholds(M, 3, seed=seed)
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA,
                         storage=_lib.STORE_2BIT if store.startswith("2bit") else _lib.STORE_INT8,
                         digit_planes=4 if store == "2bit4" else 0) as e:      # "2bit4": four digit planes on 2-bit rows (28-bit, as on int8 rows)
        e.synth_genotypes(M, N, seed, th)
        st = e.snp_stats(gpca.QcConfig.none())
        counts, _ = e.snp_qc_detail()
        # (1) generator + stats: spot rows against the oracle (same global row index -> same bytes)
        rows = np.array([0, 1, 65535, 65536, 123457, 777777, M - 1])
        pca_rows = e.pca_snp_rows()
        for i in rows:
            g_row = oracle.synth_genotypes(1, N, seed, th[i:i + 1], snp_offset=int(i))
            o = oracle.snp_stats(g_row, N, 0.0, 0.0, 1.0)
            assert np.array_equal(counts[i], o["counts"][0]) and st["mu"][i] == o["mu"][0]
            assert abs(float(st["sigma"][i]) - float(o["sigma"][0])) <= np.spacing(o["sigma"][0])
            blk = e.standardize_block([int(np.searchsorted(pca_rows, i))], np.arange(0, N, 997))
            ref, err = oracle.standardize_block(g_row, o["mu"], o["sigma"], [0], np.arange(0, N, 997))
            assert err is None and np.array_equal(blk, ref)
        assert counts[:, 0].min() == N and int(st["keep"].sum()) == M         # no missing, everything kept
        # checksum of checksums: total allele count = sum of row sums
        tot = int(counts[:, 2].astype(np.int64).sum() + 2 * counts[:, 3].astype(np.int64).sum())
        assert tot == int(np.rint((st["mu"].astype(np.float64) * N).sum()))
        e.rsvd(k, 10, 2, seed=seed)
        sc = e.scores(f64=True); ev = e.eigenvalues(); sv = e.singular_values(); ld = e.loadings()
        # (2) scores = V * s with orthonormal V; loadings orthonormal
        gram = sc.T @ sc
        assert np.allclose(np.diag(gram), sv[:k] ** 2, rtol=1e-6)
        off = gram - np.diag(np.diag(gram))
        assert np.max(np.abs(off)) < 1e-6 * sv[0] ** 2
        lg = ld.astype(np.float64).T @ ld.astype(np.float64)
        assert np.max(np.abs(lg - np.eye(k))) < 1e-4
        assert np.allclose(ev, sv[:k] ** 2 / (N - 1), rtol=1e-12) and np.all(np.diff(ev) <= 0)
        # (3) the standardised matrix is row-centred: every PC is orthogonal to the all-ones sample vector
        assert np.max(np.abs(sc.sum(axis=0))) < 1e-6 * np.abs(sc).sum(axis=0).max()
        # (4) idempotence: same seed -> bitwise same answer; 3 populations -> exactly 2 structured eigenvalues
        e.rsvd(k, 10, 2, seed=seed)
        assert np.array_equal(e.eigenvalues(), ev) and np.array_equal(e.scores(f64=True), sc)
        assert ev[1] > 20 * ev[2]
        # (5) PCA::transform consistency: A^T U = V s up to convergence of the trailing (noise) PCs
        tr = e.transform()
        assert oracle.max_abs_dpc(tr[:, :2], sc[:, :2]) < 1e-4
        # (6) loadings of spot rows against a direct f64 evaluation  u_i = a_i . V_k / s  from the oracle's bytes
        V = sc[:, :2] / sv[:2]
        for i in rows:
            g_row = oracle.synth_genotypes(1, N, seed, th[i:i + 1], snp_offset=int(i)).astype(np.float64)[0]
            a_i = (g_row - float(st["mu"][i])) / float(st["sigma"][i])
            assert np.max(np.abs(a_i @ V / sv[:2] - ld[i, :2].astype(np.float64))) < 1e-5
    _FULL[(prec, store)] = (ev, sc[:, :2].copy(), ld[:, :2].astype(np.float64))
    return _FULL[(prec, store)]


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit"), ("i8", "2bit4"), ("f32", "int8"), ("f32", "2bit")])
def test_full_size_properties(gpca, oracle, prec, store):
    """BASELINE.json configs[1] at full size on EVERY GEMM path, named explicitly: ("i8", "int8") is the headline path of
    bench.py (k_gq_d / k_gtt_d, LDS-DMA), ("i8", "2bit") the packed kernels (k_gq_2bit / k_gtt_p), ("f32", "int8") the
    f32 matrix-core kernels.  The oracle cannot run at this size, so: spot rows against the oracle, orthogonality,
    centring, idempotence, PCA::transform consistency (_full_size_case), and agreement of the structured PCs between the paths:
    every path against the headline path -- 1e-6 for the exact-integer pair (the same integers; 24-bit against 28-bit digit planes),
    1e-5 for the f32 paths -- and f32 on 2-bit rows == f32 on int8 rows bit for bit.  The headline result is computed on demand
    (and cached), so no comparison depends on the order the parameters run in."""
    ev, sc2, ld2 = _full_size_case(gpca, oracle, prec, store)
    ev0, sc0, ld0 = _full_size_case(gpca, oracle, "i8", "int8")
    if (prec, store) != ("i8", "int8"):
        # i8 on 2-bit rows: three digit planes by default (a 24-bit fixed point per column): 1e-6; with four planes the packed
        # kernels multiply the same integers as the int8-resident ones: 1e-8 (only the f32 partials of c may be grouped differently)
        tol = 1e-8 if store == "2bit4" else (1e-6 if prec == "i8" else 1e-5)
        assert np.max(np.abs(ev[:2] - ev0[:2]) / ev0[:2]) < tol
        assert oracle.max_abs_dpc(sc2, sc0) < tol and oracle.max_abs_dpc(ld2, ld0) < 10 * tol
    if (prec, store) == ("f32", "2bit"):      # same f32 FMA chains on the decoded codes: the same bits
        evf, scf, _ = _full_size_case(gpca, oracle, "f32", "int8")
        assert np.array_equal(ev, evf) and np.array_equal(sc2, scf)


@pytest.mark.parametrize("M,N", [(1_250_000, 100_000)])
def test_c4_per_gpu_shard_i8_and_2bit(gpca, oracle, M, N):
    """BASELINE.json configs[3]'s PER-GPU workload at full size: 1.25M SNPs x 100k samples (one of the eight row shards of
    10M x 100k) = 125 GB of int8 genotypes resident, then the same rows as 2-bit codes.  No oracle at this size: property
    checks, bitwise repeat, and int8-resident == 2-bit-resident (the same exact integers), plus spot rows of the loadings
    against a direct f64 evaluation from the oracle's bytes."""
    from genomic_pca_amd import _lib
    k, seed = 20, 5
    th = gpca.synth_thresholds(M, 3, seed=seed)
    res = {}
    for store in ("int8", "2bit"):
        with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8) as e:
            e.synth_genotypes(M, N, seed, th)
            st = e.snp_stats(gpca.QcConfig.none())
            e.rsvd(k, 10, 2, seed=seed)
            sc = e.scores(f64=True); ev = e.eigenvalues(); sv = e.singular_values(); ld = e.loadings()
            e.rsvd(k, 10, 2, seed=seed)
            assert np.array_equal(e.eigenvalues(), ev) and np.array_equal(e.scores(f64=True), sc) and np.array_equal(e.loadings(), ld)
            gram = sc.T @ sc
            assert np.allclose(np.diag(gram), sv[:k] ** 2, rtol=1e-6)
            assert np.max(np.abs(gram - np.diag(np.diag(gram)))) < 1e-6 * sv[0] ** 2
            lg = ld.astype(np.float64).T @ ld.astype(np.float64)
            assert np.max(np.abs(lg - np.eye(k))) < 1e-4
            assert np.max(np.abs(sc.sum(axis=0))) < 1e-6 * np.abs(sc).sum(axis=0).max()
            assert ev[1] > 20 * ev[2]
            V = sc[:, :2] / sv[:2]
            for i in (0, 77_777, 654_321, M - 1):
                g_row = oracle.synth_genotypes(1, N, seed, th[i:i + 1], snp_offset=int(i)).astype(np.float64)[0]
                a_i = (g_row - float(st["mu"][i])) / float(st["sigma"][i])
                assert np.max(np.abs(a_i @ V / sv[:2] - ld[i, :2].astype(np.float64))) < 1e-5
            tr = e.transform()
            assert oracle.max_abs_dpc(tr[:, :2], sc[:, :2]) < 1e-4
            res[store] = (ev, sc, ld)
    # (2-bit rows run three digit planes by default: a 24-bit fixed point per column against the 28 bits of the int8-resident run)
    assert np.max(np.abs(res["int8"][0][:2] - res["2bit"][0][:2]) / res["int8"][0][:2]) < 1e-6
    assert oracle.max_abs_dpc(res["int8"][1][:, :2], res["2bit"][1][:, :2]) < 1e-6


# ------------------------------------------------------------------------------------------------
# exact-integer GEMM path (GPCA_PREC_I8_EXACT): same parity bars
# ------------------------------------------------------------------------------------------------
@pytest.fixture()
def engine_i8(gpca):
    from genomic_pca_amd import _lib
    e = gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT)
    yield e
    e.close()


@pytest.mark.parametrize("M,N,P,k", [(4096, 512, 12, 8), (20000, 1000, 16, 10), (3000, 1500, 10, 6), (999, 257, 8, 4), (130, 70, 4, 3)])
def test_rsvd_parity_i8(gpca, oracle, engine_i8, M, N, P, k):
    G, r, b, R = _rsvd_case(gpca, oracle, engine_i8, M, N, P, k, seed=1, fst=0.2)
    e = engine_i8
    assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
    assert oracle.max_abs_dpc(e.scores(f64=True), R["scores"]) < TOL_PC
    assert oracle.max_abs_dpc(e.loadings().astype(np.float64), R["loadings"]) < TOL_PC
    al = oracle.sign_align(e.scores(f64=True), R["scores"])
    assert np.max(np.abs(al - R["scores"])) < 1e-4 * np.max(np.abs(R["scores"]))


def test_i8_matches_f32_and_is_partition_independent(gpca, oracle, engine, engine_i8, monkeypatch):
    M, N = 6000, 640
    th = gpca.synth_thresholds(M, 8, seed=5, fst=0.3)
    G = oracle.synth_genotypes(M, N, 5, th)
    out = {}
    for name, e in (("f32", engine), ("i8", engine_i8)):
        e.upload_genotypes_i8(G); e.snp_stats(); e.rsvd(6, 10, 2, seed=3)
        out[name] = (e.eigenvalues(), e.scores(f64=True), e.transform())
    assert np.max(np.abs(out["i8"][0] - out["f32"][0]) / out["f32"][0]) < 1e-5
    assert oracle.max_abs_dpc(out["i8"][1], out["f32"][1]) < 1e-5
    assert oracle.max_abs_dpc(out["i8"][2], out["f32"][2]) < 1e-5
    # the integer GEMM partial sums are exact and the centring term c = b^T T is summed per 32-row unit in a fixed order, so a
    # different grid partition returns the same bits
    from genomic_pca_amd import _lib
    engine_defaults(monkeypatch, gq_waves=64, gtt_waves=96)
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e2:
        e2.upload_genotypes_i8(G); e2.snp_stats(); e2.rsvd(6, 10, 2, seed=3)
        assert np.array_equal(e2.eigenvalues(), out["i8"][0])
        assert np.array_equal(e2.scores(f64=True), out["i8"][1])


@pytest.mark.parametrize("store,planes", [("int8", 0), ("2bit", 0), ("2bit", 3)])
def test_i8_wide_sketch_two_column_halves(gpca, oracle, store, planes):
    """32 < l <= 64 on the exact path: the 32-column kernels run over the two halves of the sketch (k = 40 -> l = 50, the
    shape class of BASELINE config 5), same answers as the oracle and as the f32-MFMA path."""
    from genomic_pca_amd import _lib
    M, N, k = 6000, 700, 40
    th = gpca.synth_thresholds(M, 48, seed=3, fst=0.3)
    G = oracle.synth_genotypes(M, N, 3, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=3)
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8,
                         digit_planes=planes) as e:
        e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, 10, 2, seed=3)
        assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
        # the trailing PCs of this small case sit in the noise bulk: compare the structured ones (as test_rsvd_l64_path does)
        assert oracle.max_abs_dpc(e.scores(f64=True)[:, :20], R["scores"][:, :20]) < TOL_PC
        assert oracle.max_abs_dpc(e.loadings().astype(np.float64)[:, :20], R["loadings"][:, :20]) < TOL_PC
        tr = e.transform()
    with gpca.GpcaEngine(precision=_lib.PREC_F32_MFMA) as f:      # PCA::transform on the wide sketch: same numbers as the f32 path
        f.upload_genotypes_i8(G); f.snp_stats(gpca.QcConfig.none()); f.rsvd(k, 10, 2, seed=3)
        assert oracle.max_abs_dpc(tr[:, :20], f.transform()[:, :20]) < TOL_PC
    with pytest.raises(gpca.GpcaError):
        with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e2:
            e2.upload_genotypes_i8(G); e2.snp_stats(gpca.QcConfig.none()); e2.rsvd(120, 10)     # l = 130 > 128 (l = 70: test_rsvd_wide_sketch)


# ------------------------------------------------------------------------------------------------
# 2-bit resident genotypes (GPCA_STORE_2BIT, decode in the GEMM prologues): same answers as int8 residency
# ------------------------------------------------------------------------------------------------
@pytest.fixture()
def engine_2bit(gpca):
    from genomic_pca_amd import _lib
    e = gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT)
    yield e
    e.close()


@pytest.mark.parametrize("M,N,P", [(1000, 333, 3), (257, 1025, 5), (64, 2100, 2), (4096, 64, 3), (3, 7, 2)])
def test_2bit_synth_and_upload_roundtrip(gpca, oracle, engine_2bit, M, N, P):
    th = gpca.synth_thresholds(M, P, seed=42)
    engine_2bit.synth_genotypes(M, N, 42, th)
    ref = oracle.synth_genotypes(M, N, 42, th)
    assert np.array_equal(engine_2bit.download_genotypes_i8(), ref)
    G = _inject(ref, np.random.default_rng(1), 0.03)
    engine_2bit.upload_genotypes_i8(G)
    assert np.array_equal(engine_2bit.download_genotypes_i8(), G)


def test_2bit_stats_and_block_parity(gpca, oracle, engine_2bit):
    M, N = 1500, 777
    rng = np.random.default_rng(9)
    G = _inject(oracle.synth_genotypes(M, N, 3, gpca.synth_thresholds(M, 3, seed=3, fst=0.1)), rng, 0.02)
    G[0] = 0; G[1] = 2; G[2] = -127; G[3, : N // 2] = -127; G[4] = np.where(np.arange(N) % 2 == 0, 0, 2)
    engine_2bit.upload_genotypes_i8(G)
    for qc in [gpca.QcConfig.none(), gpca.QcConfig(), gpca.QcConfig(0.9, 0.05, 1e-3)]:
        st = engine_2bit.snp_stats(qc)
        counts, reason = engine_2bit.snp_qc_detail()
        ref = oracle.snp_stats(G, N, qc.min_snp_call_rate, qc.min_snp_maf, qc.max_snp_hwe_p_value)
        assert np.array_equal(counts, ref["counts"]) and np.array_equal(st["keep"], ref["keep"]) and np.array_equal(reason, ref["reason"])
        assert np.array_equal(st["mu"], ref["mu"]) and np.all(np.abs(st["sigma"] - ref["sigma"]) <= np.spacing(ref["sigma"]))
    st = engine_2bit.snp_stats(gpca.QcConfig(1.0, 0.05, 1.0))          # only fully called SNPs
    rows = engine_2bit.pca_snp_rows()
    sid = rng.permutation(len(rows))[:50]; cid = rng.permutation(N)[:123]
    out = engine_2bit.standardize_block(sid, cid)
    ref, err = oracle.standardize_block(G, st["mu"], st["sigma"], rows[sid], cid)
    assert err is None and np.array_equal(out, ref)
    engine_2bit.snp_stats(gpca.QcConfig(0.9, 0.05, 1.0))               # now SNPs with missing genotypes are kept
    with pytest.raises(gpca.GpcaError) as e:
        engine_2bit.standardize_block(np.arange(engine_2bit.num_pca_snps()), np.arange(N))
    assert e.value.status == -5
    with pytest.raises(gpca.GpcaError) as e:
        engine_2bit.rsvd(2, 2, 1, 1)
    assert e.value.status == -5


def test_2bit_bed_upload_stays_packed(gpca, oracle, engine_2bit):
    z = np.load(os.path.join(GOLD, "chr22_subset50_slice.npz"))
    engine_2bit.upload_bed2bit(z["bed_rows"], int(z["n_samples"]))
    assert np.array_equal(engine_2bit.download_genotypes_i8(), z["dosage_count_a1"])
    rng = np.random.default_rng(3)
    for n2 in (1, 5, 63, 130, 1030):
        b = rng.integers(0, 256, size=(37, (n2 + 3) // 4), dtype=np.uint8)
        lut = np.array([2, -127, 1, 0], np.int8)
        exp = np.empty((37, b.shape[1] * 4), np.int8)
        for s in range(4):
            exp[:, s::4] = lut[(b >> (2 * s)) & 3]
        engine_2bit.upload_bed2bit(b, n2)
        assert np.array_equal(engine_2bit.download_genotypes_i8(), exp[:, :n2])
        st = engine_2bit.snp_stats(gpca.QcConfig.none())
        o = oracle.snp_stats(exp[:, :n2], n2, 0.0, 0.0, 1.0)
        assert np.array_equal(engine_2bit.snp_qc_detail()[0], o["counts"]) and np.array_equal(st["keep"], o["keep"])


def test_2bit_invalid_values_flagged(gpca, engine_2bit):
    G = np.random.default_rng(5).integers(0, 3, size=(40, 300), dtype=np.int8)
    G[3, 17] = 3
    engine_2bit.upload_genotypes_i8(G)
    engine_2bit.snp_stats(gpca.QcConfig.none())
    with pytest.raises(gpca.GpcaError) as e:
        engine_2bit.rsvd(2, 2, 1, 1)
    assert e.value.status in (-5, -9)


@pytest.mark.parametrize("M,N,P,k", [(4096, 512, 12, 8), (20000, 1000, 16, 10), (3000, 1500, 10, 6), (999, 257, 8, 4), (130, 70, 4, 3)])
def test_rsvd_parity_2bit(gpca, oracle, engine_2bit, M, N, P, k):
    G, r, b, R = _rsvd_case(gpca, oracle, engine_2bit, M, N, P, k, seed=1, fst=0.2)
    e = engine_2bit
    assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
    assert oracle.max_abs_dpc(e.scores(f64=True), R["scores"]) < TOL_PC
    assert oracle.max_abs_dpc(e.loadings().astype(np.float64), R["loadings"]) < TOL_PC
    tr = e.transform()
    A = oracle.standardized_dense(G, N, r, b)
    ref = A.T @ e.loadings().astype(np.float64)
    assert np.max(np.abs(tr - ref)) < 1e-4 * np.max(np.abs(ref))


def test_2bit_equals_int8_residency(gpca, oracle, engine_i8, engine_2bit):
    """Same exact-integer arithmetic on the same codes: with the same four digit planes the packed path reproduces the int8-resident
    path (1e-8: only the quantisation grid of T' per launch differs); the packed default (three planes) sits within 1e-6."""
    from genomic_pca_amd import _lib
    M, N = 6000, 900
    G = oracle.synth_genotypes(M, N, 5, gpca.synth_thresholds(M, 8, seed=5, fst=0.3))
    res = []
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT, digit_planes=4) as engine_2bit_4p:
        for e in (engine_i8, engine_2bit_4p, engine_2bit):
            e.upload_genotypes_i8(G); e.snp_stats(); e.rsvd(6, 10, 2, seed=3)
            res.append((e.eigenvalues(), e.scores(f64=True)))
    assert np.max(np.abs(res[0][0] - res[1][0]) / res[0][0]) < 1e-8
    assert oracle.max_abs_dpc(res[0][1], res[1][1]) < 1e-8
    assert np.max(np.abs(res[0][0] - res[2][0]) / res[0][0]) < 1e-6
    assert oracle.max_abs_dpc(res[0][1], res[2][1]) < 1e-6


@pytest.mark.parametrize("M,N,P,k", [(4096, 512, 12, 8), (20000, 1000, 16, 10), (3000, 1500, 10, 6), (999, 257, 8, 4), (130, 70, 4, 3), (6000, 700, 48, 40)])
def test_f32_path_on_2bit_residency(gpca, oracle, engine, M, N, P, k):
    """GPCA_PREC_F32_MFMA on GPCA_STORE_2BIT (north_star's literal configuration: MFMA-fp32 GEMMs on a matrix that only fits one
    GPU packed): the codes are spread in the conversion block and take the same f32 FMA chains, so the answers are the
    int8-resident f32 path's, BIT FOR BIT, and hold the oracle's 1e-4 bar."""
    from genomic_pca_amd import _lib
    th = gpca.synth_thresholds(M, P, seed=1, fst=0.2)
    G = oracle.synth_genotypes(M, N, 1, th)
    G[5::11, :3] = -127                                        # packed code 3 inside rows that QC drops
    qc = gpca.QcConfig(0.999, 0.0, 1.0)
    with gpca.GpcaEngine(precision=_lib.PREC_F32_MFMA, storage=_lib.STORE_2BIT) as e2:
        e2.upload_genotypes_i8(G); st2 = e2.snp_stats(qc); e2.rsvd(k, 10, 2, seed=1)
        engine.upload_genotypes_i8(G); st8 = engine.snp_stats(qc); engine.rsvd(k, 10, 2, seed=1)
        assert np.array_equal(st2["keep"], st8["keep"]) and np.array_equal(st2["mu"], st8["mu"])
        assert np.array_equal(e2.eigenvalues(), engine.eigenvalues())
        assert np.array_equal(e2.scores(f64=True), engine.scores(f64=True)) and np.array_equal(e2.loadings(), engine.loadings())
        assert np.array_equal(e2.transform(), engine.transform())
        ref = oracle.snp_stats(G, N, qc.min_snp_call_rate, qc.min_snp_maf, qc.max_snp_hwe_p_value)
        r, b = oracle.scale_shift(ref["mu"], ref["sigma"], ref["keep"])
        R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=1)
        kk = min(k, 20)                                        # (k = 40 case: the trailing PCs sit in the noise bulk)
        assert np.max(np.abs(e2.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
        assert oracle.max_abs_dpc(e2.scores(f64=True)[:, :kk], R["scores"][:, :kk]) < TOL_PC
        assert oracle.max_abs_dpc(e2.loadings().astype(np.float64)[:, :kk], R["loadings"][ref["keep"].astype(bool)][:, :kk]) < TOL_PC


# ------------------------------------------------------------------------------------------------
# wide sample axis (N ~ 5e4, BASELINE configs[3] shape class): all three GEMM paths against the oracle
# ------------------------------------------------------------------------------------------------
def test_wide_sample_axis_all_paths(gpca, oracle):
    from genomic_pca_amd import _lib
    M, N, P, k = 6000, 50_000, 12, 8
    th = gpca.synth_thresholds(M, P, seed=19, fst=0.2)
    G = oracle.synth_genotypes(M, N, 19, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=4)
    for prec, store in ((_lib.PREC_F32_MFMA, _lib.STORE_INT8), (_lib.PREC_F32_MFMA, _lib.STORE_2BIT), (_lib.PREC_I8_EXACT, _lib.STORE_INT8),
                        (_lib.PREC_I8_EXACT, _lib.STORE_2BIT)):
        with gpca.GpcaEngine(precision=prec, storage=store) as e:
            e.upload_genotypes_i8(G)
            s2 = e.snp_stats(gpca.QcConfig.none())
            assert np.array_equal(s2["mu"], st["mu"]) and np.array_equal(s2["keep"], st["keep"])
            e.rsvd(k, 10, 2, seed=4)
            assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
            assert oracle.max_abs_dpc(e.scores(f64=True), R["scores"]) < TOL_PC
            assert oracle.max_abs_dpc(e.loadings().astype(np.float64), R["loadings"]) < TOL_PC


# ------------------------------------------------------------------------------------------------
# bitwise repeatability: hand-counted waits (LDS-DMA K1) and inline-asm wait states show up as runs that differ
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,store,M,N,k", [("i8", "int8", 300_000, 2048, 20), ("i8", "int8", 40_000, 10_000, 10),
                                               ("i8", "2bit", 300_000, 2048, 20), ("f32", "int8", 300_000, 2048, 20),
                                               ("f32", "int8", 100_000, 1024, 40), ("f32", "2bit", 300_000, 2048, 20)])
def test_rsvd_bitwise_repeatable(gpca, prec, store, M, N, k):
    """Every kernel has a fixed reduction tree and no float atomics, so the same call must return the same bits.  Sizes
    are chosen so that each K1 workgroup runs full LDS-DMA rounds (> 8 row units per workgroup)."""
    from genomic_pca_amd import _lib
    p = _lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA
    s = _lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8
    th = gpca.synth_thresholds(M, 3, seed=7)
    with gpca.GpcaEngine(precision=p, storage=s) as e:
        e.synth_genotypes(M, N, 7, th)
        e.snp_stats(gpca.QcConfig.none())
        ref = None
        for _ in range(6):
            e.rsvd(k, 10, 2, seed=3)
            cur = (e.eigenvalues(), e.scores(f64=True), e.loadings())
            if ref is None:
                ref = cur
            else:
                assert all(np.array_equal(a, b) for a, b in zip(ref, cur))


# ------------------------------------------------------------------------------------------------
# the alternative kernels behind the diagnostic switches stay correct ("Diagnostic switches")
# ------------------------------------------------------------------------------------------------
def _fuzz_cases():
    rng = np.random.default_rng(20261004)
    cases = []
    for i in range(28):
        k = int(rng.integers(1, 9))
        N = int(rng.choice([rng.integers(k + 14, 90), rng.integers(90, 700), rng.integers(700, 2600)]))
        M = int(rng.choice([rng.integers(k + 14, 200), rng.integers(200, 3000), rng.integers(3000, 9000)]))
        cases.append((i, M, N, k, int(rng.integers(0, 13)), int(rng.integers(0, 4)), ["i8", "f32"][i % 2 if i % 5 else 0],
                      ["int8", "2bit"][(i // 2) % 2], float(rng.choice([0.0, 0.0, 0.01])), int(rng.integers(1, 10**6))))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"{c[0]}-{c[1]}x{c[2]}-k{c[3]}+{c[4]}-q{c[5]}-{c[6]}-{c[7]}-miss{c[8]}")
def test_random_shapes_against_the_oracle(gpca, oracle, monkeypatch, case):
    """Seeded random shapes, sketch widths, power-iteration counts, residencies, precisions and missing rates (rows with missing
    genotypes leave through the call-rate filter): QC decisions bit-exact, eigenvalues / unit-norm PCs / loadings within the
    1e-4 bar of the oracle with the same sketch.  Population structure with more groups than components keeps the requested
    eigenvalues apart, so that the comparison is not one of rotations inside a degenerate subspace."""
    from genomic_pca_amd import _lib
    idx, M, N, k, ov, q, prec, store, miss, seed = case
    ov = min(ov, min(M, N) - k - 1)
    if idx % 3 == 0:      # a grid of 2 workgroups: the full LDS-DMA rounds (>= 9 units of 32 rows per workgroup) run on these small matrices too
        engine_defaults(monkeypatch, gq_waves=8, gtt_waves=64)
    th = gpca.synth_thresholds(M, k + 3, seed=seed, fst=0.35)
    G = oracle.synth_genotypes(M, N, seed, th)
    if miss:
        rng = np.random.default_rng(seed)
        rows = rng.choice(M, max(1, int(miss * M)), replace=False)
        G[rows, rng.integers(0, N, len(rows))] = -127
    qc = (1.0 if miss else 0.0, 0.0, 1.0)                              # missing rows fail a call rate of 1.0
    ref = oracle.snp_stats(G, N, *qc)
    if int(ref["keep"].sum()) < k + ov + 1:
        pytest.skip("too few SNPs pass QC for this draw")
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA,
                         storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8) as e:
        e.upload_genotypes_i8(G)
        st = e.snp_stats(gpca.QcConfig(*qc))
        assert np.array_equal(st["keep"], ref["keep"]) and np.array_equal(st["mu"], ref["mu"])
        e.rsvd(k, ov, q, seed=seed)
        ev, sc, ld = e.eigenvalues(), e.scores(f64=True), e.loadings().astype(np.float64)
    r, b = oracle.scale_shift(ref["mu"], ref["sigma"], ref["keep"])
    R = oracle.rsvd(G, N, r, b, k, ov, q, seed=seed)
    assert np.max(np.abs(ev - R["eigenvalues"]) / R["eigenvalues"]) < 1e-4
    gaps = np.abs(np.diff(np.append(R["singular_values"][:k], R["singular_values"][k] if k < len(R["singular_values"]) else 0.0))) / R["singular_values"][:k]
    sep = gaps > 1e-2                                                  # (a PC whose singular value sits within 1 % of the next is compared by its eigenvalue only)
    if sep.any():
        assert oracle.max_abs_dpc(sc[:, sep], R["scores"][:, sep]) < 1e-4
        assert oracle.max_abs_dpc(ld[:, sep], R["loadings"][ref["keep"].astype(bool)][:, sep]) < 1e-4


@pytest.mark.parametrize("env", [dict(simple=1), dict(simple=1, gq_waves=8, gtt_waves=64), dict(gq_waves=8, gtt_waves=64), dict(gq_waves=8),
                                 dict(gq_waves=12, gtt_waves=24), dict(gq_waves=4), dict(gtt_waves=8), dict(gtt_waves=2048 + 8), dict(spin_sync=0)])
def test_alternative_kernels_same_answer(gpca, oracle, monkeypatch, env):
    """The register-only reference kernels (k_gq_i8 / k_gtt_i8: gpca_config.reserved[0] & GPCA_CFG_SIMPLE_KERNELS) and small grids that
    force full LDS-DMA rounds, chained rounds and multi-task workgroups on a small matrix, against the default configuration: the
    integer products are exact and every f32 rounding is pinned, so the results are the SAME BITS, and the oracle parity bar holds."""
    from genomic_pca_amd import _lib
    M, N, k = 20000, 1000, 10
    th = gpca.synth_thresholds(M, 16, seed=1, fst=0.2)
    G = oracle.synth_genotypes(M, N, 1, th)
    res = {}
    for name, e_env in (("default", {}), ("alt", env)):
        with monkeypatch.context() as mp:
            engine_defaults(mp, **e_env)
            with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e:
                e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, 10, 2, seed=1)
                res[name] = (e.eigenvalues(), e.scores(f64=True), e.loadings().astype(np.float64))
    for a, b in zip(res["alt"], res["default"]):
        assert np.array_equal(a, b)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=1)
    assert oracle.max_abs_dpc(res["alt"][1], R["scores"]) < TOL_PC


@pytest.mark.parametrize("M,N", [(m, n) for n in (200, 1000, 1300) for m in (500, 1500, 2500, 3700)] +
                         [(1500, 500), (3700, 500), (1500, 700), (3700, 700), (2500, 1800), (1500, 2504), (3700, 2504)])
def test_short_dma_rounds_same_bits_as_the_register_staged_kernel(gpca, oracle, monkeypatch, M, N):
    """k_gq_d's rounds of fewer than four tiles per wave (R = 3, 2, 1 and waves that only ride along): grids of 1, 2 and 3 workgroups
    over 16 ... 116 row units leave every remainder class behind the full rounds -- (3,3,2,2), (2,2,1,1), (1,1,1,1), (2,1,1,1),
    (1,1,0,0) ... --, on 2 stages (the ring's prologue wraps around the sample axis) and on 8.  The register-staged kernel
    (GPCA_CFG_SIMPLE_KERNELS) computes the same integers with the same pinned f32 roundings: every result must be the same bits, and the
    oracle bar holds."""
    from genomic_pca_amd import _lib
    k = 5
    th = gpca.synth_thresholds(M, 3, seed=5, fst=0.1)
    G = oracle.synth_genotypes(M, N, 5, th)
    res = {}
    # (N = 1000, 1300, 1800 and 2504 are 8, 12, 16 and 20 stages of 128 samples: the rounds of a workgroup are CHAINED there -- the next
    #  round's first units and planes are fetched behind the current round's epilogue, whatever the two rounds' tile counts -- and from
    #  16 stages the workgroups start their sweeps at different stages; 2, 4 and 6 stages (N = 200, 500, 700) keep the drained form:
    #  the sample axes of 1000 Genomes (2 504) and of a few hundred to a few thousand samples that the shape sweep is about)
    for name, env in (("staged", dict(simple=1)), ("w4", dict(gq_waves=4)), ("w8", dict(gq_waves=8)), ("w12", dict(gq_waves=12)),
                      ("w4_k2w8", dict(gq_waves=4, gtt_waves=8)), ("w16", dict(gq_waves=16, gtt_waves=32)), ("default", {})):
        with monkeypatch.context() as mp:
            engine_defaults(mp, **env)
            with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e:
                e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, 5, 2, seed=3)
                res[name] = (e.eigenvalues().copy(), e.scores(f64=True).copy(), e.loadings().copy())
    for name in ("w4", "w8", "w12", "w4_k2w8", "w16", "default"):
        for a, b in zip(res[name], res["staged"]):
            assert np.array_equal(a, b), name
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 5, 2, seed=3)
    assert np.max(np.abs(res["default"][0] - R["eigenvalues"]) / R["eigenvalues"]) < 1e-4
    assert oracle.max_abs_dpc(res["default"][1][:, :2], R["scores"][:, :2]) < TOL_PC


@pytest.mark.parametrize("M,N,k", [(70_001, 64, 20), (70_001, 128, 20), (33_333, 129, 10), (50_000, 256, 40), (40_000, 50, 5), (300, 2, 1)])
def test_narrow_kernels_same_bits_as_the_wide_ones(gpca, oracle, monkeypatch, M, N, k):
    """At most 256 samples on int8 rows (BASELINE.json configs[2]'s shape class: 1 066 557 x 64): k_gq_n keeps all of Q's digit
    planes in registers and streams only the lines that hold samples, the narrow k_gtt_i8 gives every wave a row chunk of its
    own.  Same integers, same pinned f32 roundings, same per-unit c partials: the results are the wide kernels' bits
    (GPCA_CFG_NO_NARROW runs those on the padded rows) -- resident, streamed 6-pass and fused -- and hold the oracle's parity bar."""
    from genomic_pca_amd import _lib
    th = gpca.synth_thresholds(M, 5, seed=2, fst=0.25)
    l = min(k + 10, N)
    res = {}
    for name, env in (("narrow", {}), ("wide", dict(narrow=0))):
        with monkeypatch.context() as mp:
            engine_defaults(mp, **env)
            with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e:
                e.synth_genotypes(M, N, 2, th)
                st = e.snp_stats(gpca.QcConfig.none())
                e.rsvd(k, l - k, 2, seed=4)
                res[name] = (e.eigenvalues(), e.scores(f64=True), e.loadings(), e.transform())
                if name == "narrow":
                    G = e.download_genotypes_i8()
                    e.stream_open(gpca.PanelSource.synth(th, 2), M, N, panel_rows=8192, ring_slots=2, fused=False)
                    e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, l - k, 2, seed=4)
                    assert np.array_equal(e.eigenvalues(), res[name][0]) and np.array_equal(e.scores(f64=True), res[name][1])
                    e.stream_open(gpca.PanelSource.synth(th, 2), M, N, panel_rows=8192, ring_slots=2, fused=True)
                    e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, l - k, 2, seed=4)
                    assert np.max(np.abs(e.eigenvalues() - res[name][0]) / res[name][0]) < 1e-7
    for a, b in zip(res["narrow"], res["wide"]):
        assert np.array_equal(a, b)
    if N >= 50:
        r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
        R = oracle.rsvd(G, N, r, b, k, l - k, 2, seed=4)
        nz = R["eigenvalues"] > 1e-9 * R["eigenvalues"][0]
        assert np.max(np.abs(res["narrow"][0][nz] - R["eigenvalues"][nz]) / R["eigenvalues"][nz]) < TOL_EV
        assert oracle.max_abs_dpc(res["narrow"][1][:, :3], R["scores"][:, :3]) < TOL_PC


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit"), ("f32", "int8")])
def test_compact_child_when_qc_drops_most_rows(gpca, oracle, monkeypatch, prec, store):
    """QC that drops most SNP rows (configs[2] keeps 203 512 of 1 066 557): gpca_rsvd gathers the kept rows into a matrix of their
    own and runs there -- the passes, the sketch and the quantisations cost n_pca rows instead of M.  Omega is drawn by the rows'
    ORIGINAL index, so the sketch is the one of the uncompacted run (GPCA_CFG_NO_COMPACT): same answers (only the grouping of the
    f32 partial sums of c moves), the oracle's parity bar, loadings in PCA-SNP order, PCA::transform, the pull API still served
    by the parent, and a changed keep mask rebuilds the child."""
    from genomic_pca_amd import _lib
    M, N, k = 60_000, 700, 8
    th = gpca.synth_thresholds(M, 6, seed=3, fst=0.25)
    G = oracle.synth_genotypes(M, N, 3, th)
    rng = np.random.default_rng(0)
    keep = (rng.random(M) < 0.2).astype(np.uint8)            # 20 % of the rows stay
    kw = dict(precision=_lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA, storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8)
    res = {}
    for name, env in (("compact", {}), ("plain", dict(compact=0))):
        with monkeypatch.context() as mp:
            engine_defaults(mp, **env)
            with gpca.GpcaEngine(**kw) as e:
                e.upload_genotypes_i8(G)
                st = e.snp_stats(gpca.QcConfig.none())
                e.set_standardization(st["mu"], st["sigma"], keep)
                e.enable_timings(True)
                e.rsvd(k, 10, 2, seed=5)
                tim = e.timings()
                res[name] = (e.eigenvalues(), e.scores(f64=True), e.loadings(), e.transform(), e.singular_values())
                assert e.num_pca_snps() == int(keep.sum()) and res[name][2].shape == (int(keep.sum()), k)
                # algorithmic bytes of a GEMM launch: the rows it swept
                per = 0.25 if store == "2bit" else 1.0
                rows_swept = tim["gemm_GQ"]["bytes"] / tim["gemm_GQ"]["launches"] / (N * per)
                assert abs(rows_swept - (int(keep.sum()) if name == "compact" else M)) < 1
                blk = e.standardize_block([0, 5], [0, 1, 2])             # the pull API: served by the parent, PCA-SNP numbering
                rows = e.pca_snp_rows()
                ref, err = oracle.standardize_block(G, st["mu"], st["sigma"], rows[[0, 5]], [0, 1, 2])
                assert err is None and np.array_equal(blk, ref)
                if name == "compact":                                    # a different keep mask: the child is rebuilt
                    keep2 = keep.copy(); keep2[::3] = 0
                    e.set_standardization(st["mu"], st["sigma"], keep2)
                    e.rsvd(k, 10, 2, seed=5)
                    assert e.loadings().shape == (int(keep2.sum()), k)
                    r2, b2 = oracle.scale_shift(st["mu"], st["sigma"], keep2)
                    R2 = oracle.rsvd(G, N, r2, b2, k, 10, 2, seed=5)
                    assert oracle.max_abs_dpc(e.scores(f64=True)[:, :5], R2["scores"][:, :5]) < TOL_PC
    # ... and only when it pays: a matrix that loses fewer than 32 Ki rows is swept as it is -- the EigenSNP hosts re-target one
    # scratch handle at thousands of ~300-row LD blocks, where a child per block was milliseconds of allocator traffic for microseconds saved
    with gpca.GpcaEngine(**kw) as e:
        Ms = 20_000
        e.upload_genotypes_i8(G[:Ms]); st_s = e.snp_stats(gpca.QcConfig.none())
        e.set_standardization(st_s["mu"], st_s["sigma"], keep[:Ms]); e.enable_timings(True); e.rsvd(k, 10, 2, seed=5)
        tim = e.timings()
        assert abs(tim["gemm_GQ"]["bytes"] / tim["gemm_GQ"]["launches"] / (N * (0.25 if store == "2bit" else 1.0)) - Ms) < 1
    tol = 1e-7 if prec == "i8" else 1e-5
    assert np.max(np.abs(res["compact"][0] - res["plain"][0]) / res["plain"][0]) < tol
    assert oracle.max_abs_dpc(res["compact"][1][:, :5], res["plain"][1][:, :5]) < tol
    assert oracle.max_abs_dpc(res["compact"][2].astype(np.float64)[:, :5], res["plain"][2].astype(np.float64)[:, :5]) < 10 * tol
    assert oracle.max_abs_dpc(res["compact"][3][:, :5], res["plain"][3][:, :5]) < 10 * tol
    r, b = oracle.scale_shift(st["mu"], st["sigma"], keep)
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=5)
    assert np.max(np.abs(res["compact"][0] - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
    assert oracle.max_abs_dpc(res["compact"][1][:, :5], R["scores"][:, :5]) < TOL_PC
    assert oracle.max_abs_dpc(res["compact"][2].astype(np.float64)[:, :5], R["loadings"][keep.astype(bool)][:, :5]) < TOL_PC


@pytest.mark.parametrize("store", ["int8", "2bit"])
@pytest.mark.parametrize("k,oversample,q", [(22, 10, 2), (6, 0, 0), (1, 3, 1)])
def test_rsvd_i8_sketch_width_and_iteration_edges(gpca, oracle, store, k, oversample, q):
    """l = 32 uses every column of the 32-wide tiles (no zero padding columns), l = k (no oversampling) with q = 0 is a
    single sketch + projection, k = 1 is the narrowest call the reference accepts (main.rs:607-619)."""
    from genomic_pca_amd import _lib
    M, N = 5000, 600
    th = gpca.synth_thresholds(M, 40, seed=4, fst=0.25)
    G = oracle.synth_genotypes(M, N, 4, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, oversample, q, seed=9)
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8) as e:
        e.upload_genotypes_i8(G); e.snp_stats(gpca.QcConfig.none()); e.rsvd(k, oversample, q, seed=9)
        assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
        assert oracle.max_abs_dpc(e.scores(f64=True), R["scores"]) < TOL_PC
        assert oracle.max_abs_dpc(e.loadings().astype(np.float64), R["loadings"]) < TOL_PC


# ------------------------------------------------------------------------------------------------
# three digit planes (gpca_config.digit_planes = 3): packed residency, 24-bit fixed point, exact integer accumulation
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,P,k", [(20000, 1000, 16, 10), (3000, 1500, 10, 6), (130, 70, 4, 3)])
def test_rsvd_parity_2bit_three_planes(gpca, oracle, M, N, P, k):
    from genomic_pca_amd import _lib
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT, digit_planes=3) as e3, \
         gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT, digit_planes=4) as e4, \
         gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT) as e0:
        G, r, b, R = _rsvd_case(gpca, oracle, e3, M, N, P, k, seed=1, fst=0.2)
        assert np.max(np.abs(e3.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV
        assert oracle.max_abs_dpc(e3.scores(f64=True), R["scores"]) < TOL_PC
        assert oracle.max_abs_dpc(e3.loadings().astype(np.float64), R["loadings"]) < TOL_PC
        # against the four-plane engine: the same integers, a coarser fixed point (2^-24 instead of 2^-28 of the column max)
        e4.upload_genotypes_i8(G); e4.snp_stats(gpca.QcConfig.none()); e4.rsvd(k, 10, 2, seed=1)
        assert oracle.max_abs_dpc(e3.scores(f64=True), e4.scores(f64=True)) < 2e-6
        assert oracle.max_abs_dpc(e3.transform(), e4.transform()) < 2e-6
        # digit_planes = 0 (the library's choice) on 2-bit rows IS three planes: the same bits
        e0.upload_genotypes_i8(G); e0.snp_stats(gpca.QcConfig.none()); e0.rsvd(k, 10, 2, seed=1)
        assert np.array_equal(e0.eigenvalues(), e3.eigenvalues()) and np.array_equal(e0.scores(f64=True), e3.scores(f64=True))


def test_three_planes_need_packed_exact_path(gpca):
    from genomic_pca_amd import _lib
    with pytest.raises(gpca.GpcaError):
        gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_INT8, digit_planes=3)
    with pytest.raises(gpca.GpcaError):
        gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT, digit_planes=5)


@pytest.mark.parametrize("store,planes", [("int8", 0), ("2bit", 0), ("2bit", 3)])
def test_exact_path_with_qc_dropped_and_missing_rows(gpca, oracle, store, planes):
    """SNP rows that QC drops (monomorphic, low call rate with -127 / packed code 3 in them) stay resident but carry r = b = 0:
    the exact kernels must ignore whatever bytes they hold (for packed rows the decode turns code 3 into the value 3)."""
    from genomic_pca_amd import _lib
    M, N = 3000, 400
    th = gpca.synth_thresholds(M, 6, seed=4, fst=0.2)
    G = oracle.synth_genotypes(M, N, 4, th)
    G[::7] = 0
    G[5::11, :3] = -127
    G[6::13, 100:140] = -127
    qc = gpca.QcConfig(0.999, 0.02, 1e-6)
    ref = oracle.snp_stats(G, N, qc.min_snp_call_rate, qc.min_snp_maf, qc.max_snp_hwe_p_value)
    r, b = oracle.scale_shift(ref["mu"], ref["sigma"], ref["keep"])
    R = oracle.rsvd(G, N, r, b, 5, 10, 2, seed=8)
    kept = np.nonzero(ref["keep"])[0]
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8,
                         digit_planes=planes) as e:
        e.upload_genotypes_i8(G)
        st = e.snp_stats(qc)
        assert np.array_equal(st["keep"], ref["keep"]) and 0 < ref["keep"].sum() < M
        e.rsvd(5, 10, 2, seed=8)
        assert oracle.max_abs_dpc(e.loadings().astype(np.float64), R["loadings"][kept]) < TOL_PC
        assert oracle.max_abs_dpc(e.scores(f64=True), R["scores"]) < TOL_PC
        assert np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]) < TOL_EV


def test_zeroed_config_is_the_exact_path_with_automatic_residency(gpca, oracle):
    """gpca_create(NULL) / a zeroed gpca_config: GPCA_PREC_I8_EXACT, and GPCA_STORE_AUTO decided when the rows arrive -- 2-bit codes from
    1 024 samples on (three digit planes, the packed default), int8 below, int8 as well for an upload that holds a value 2-bit codes
    cannot carry; every choice returns the bits of the same choice made explicitly."""
    import ctypes as C
    from genomic_pca_amd import _lib
    lib = gpca.load()

    def run(e, G, k=6):
        e.upload_genotypes_i8(G)
        e.snp_stats(gpca.QcConfig(0.5, 0.0, 1.0))
        e.rsvd(k, 10, 2, seed=3)
        return e.eigenvalues(), e.scores(f64=True)
    rng = np.random.default_rng(5)
    for N, want in ((1500, _lib.STORE_2BIT), (700, _lib.STORE_INT8)):
        th = gpca.synth_thresholds(4000, 3, seed=2, fst=0.2)
        G = oracle.synth_genotypes(4000, N, 2, th)
        with gpca.GpcaEngine(precision=_lib.PREC_DEFAULT, storage=_lib.STORE_AUTO) as e0, gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=want) as e1:
            assert e0.storage_in_use() == (_lib.STORE_AUTO, _lib.PREC_I8_EXACT)
            ev0, sc0 = run(e0, G)
            assert e0.storage_in_use() == (want, _lib.PREC_I8_EXACT)
            ev1, sc1 = run(e1, G)
            assert np.array_equal(ev0, ev1) and np.array_equal(sc0, sc1)
            # the same handle re-resolves for the next matrix
            G2 = oracle.synth_genotypes(3000, 2500 - N, 2, gpca.synth_thresholds(3000, 3, seed=2, fst=0.2))
            run(e0, G2)
            assert e0.storage_in_use()[0] == (_lib.STORE_2BIT if 2500 - N >= 1024 else _lib.STORE_INT8)
    # a value outside {0, 1, 2, -127}: AUTO keeps the rows as int8 (the statistics see the value, as prepare.rs:1267-1279 does)
    G = oracle.synth_genotypes(2000, 1200, 4, gpca.synth_thresholds(2000, 3, seed=4, fst=0.2))
    G[7, 5] = 3
    with gpca.GpcaEngine(precision=_lib.PREC_DEFAULT, storage=_lib.STORE_AUTO) as e0, gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_INT8) as e1:
        e0.upload_genotypes_i8(G); e1.upload_genotypes_i8(G)
        assert e0.storage_in_use()[0] == _lib.STORE_INT8 and np.array_equal(e0.download_genotypes_i8(), G)
        s0, s1 = e0.snp_stats(gpca.QcConfig.none()), e1.snp_stats(gpca.QcConfig.none())
        assert np.array_equal(s0["mu"], s1["mu"]) and np.array_equal(s0["keep"], s1["keep"])
    # gpca_create(NULL)
    h = C.c_void_p()
    assert lib.gpca_create(None, C.byref(h)) == 0
    st, pr = C.c_int32(-1), C.c_int32(-1)
    assert lib.gpca_get_storage(h, C.byref(st), C.byref(pr)) == 0 and (st.value, pr.value) == (_lib.STORE_AUTO, _lib.PREC_I8_EXACT)
    lib.gpca_destroy(h)


def test_device_eigensolver_against_the_host_pin_and_lapack(gpca):
    """The l x l step of gpca_rsvd runs on the device (csrc/small_eig.hip: tridiagonalisation + implicit QL on one or two waves, the
    call's stream never waits for the host).  Held to gpca_host_eigh_desc -- the CPU pin of tests/test_abi.py, itself held to LAPACK
    there -- and to numpy.linalg.eigh directly, at 1e-12 of the largest eigenvalue, on Gram-like matrices of every size class the
    engine asks for (one wave up to 64, two waves up to 128), clustered / repeated / rank-deficient / widely spread spectra, a zero
    matrix and a huge-magnitude one (the power-of-two prescale)."""
    import ctypes as C
    lib = gpca.load()
    rng = np.random.default_rng(0)
    with gpca.GpcaEngine() as e:
        for n in (1, 2, 3, 7, 16, 30, 31, 32, 33, 50, 64, 65, 70, 100, 127, 128):
            for kind in ("gram", "spread", "rank_deficient", "repeated", "clustered", "zero", "huge"):
                B = rng.standard_normal((max(n, 2) * 3, n))
                if kind == "spread":
                    B = B * np.logspace(0, -6, n)
                if kind == "rank_deficient" and n > 2:
                    B[:, -2:] = B[:, :2]
                A = B.T @ B
                if kind == "repeated":
                    A = np.diag(np.repeat([4.0, 1.0], [n // 2, n - n // 2])) if n > 1 else np.array([[2.0]])
                if kind == "clustered":          # eigenvalues in tight clusters under a random rotation
                    Qr, _ = np.linalg.qr(rng.standard_normal((n, n)))
                    lam = np.repeat([1.0, 1.0 + 1e-9, 0.5, 1e-3], (n + 3) // 4)[:n]
                    A = (Qr * lam) @ Qr.T
                if kind == "zero":
                    A = np.zeros((n, n))
                if kind == "huge":
                    A = A * 1e200
                A = np.ascontiguousarray(0.5 * (A + A.T))
                w = np.empty(n); V = np.empty((n, n)); wh = np.empty(n); Vh = np.empty((n, n))
                rc = lib.gpca_device_eigh_desc(e._h, A.ctypes.data_as(C.c_void_p), n, w.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
                assert rc == 0, (n, kind, lib.gpca_last_error(e._h))
                assert lib.gpca_host_eigh_desc(A.ctypes.data_as(C.c_void_p), n, wh.ctypes.data_as(C.c_void_p), Vh.ctypes.data_as(C.c_void_p)) == 0
                ref = np.linalg.eigvalsh(A)[::-1]
                scale = max(abs(ref[0]), 1e-300)
                assert np.all(np.diff(w) <= 0), (n, kind)
                assert np.max(np.abs(w - ref)) < 1e-12 * scale, (n, kind, np.max(np.abs(w - ref)) / scale)
                if kind != "huge":               # (the host pair overflows at 1e200: its products are not prescaled; the device solver is)
                    assert np.max(np.abs(w - wh)) < 1e-12 * scale, (n, kind)
                assert np.max(np.abs(V.T @ V - np.eye(n))) < 1e-12, (n, kind)              # orthonormal eigenvectors
                assert np.max(np.abs(A @ V - V * w)) < 1e-11 * scale, (n, kind)            # residual
                if kind in ("gram", "spread"):      # simple spectra: the vectors themselves agree with the host pin up to sign
                    sg = np.sign(np.sum(V * Vh, axis=0))
                    gaps = np.min(np.abs(np.diff(ref))) / scale if n > 1 else 1.0
                    if gaps > 1e-6:
                        assert np.max(np.abs(V * sg - Vh)) < 1e-9, (n, kind)
        assert lib.gpca_device_eigh_desc(e._h, None, 3, None, None) == -1


@pytest.mark.gpu
def test_product_bits_did_not_move():
    """sha256 of eigenvalues, f64 scores and loadings of five fixed problems (L = 32 / 64 / 128, int8 and 2-bit rows, a sketch as wide
    as the sample count) against tests/golden/product_fingerprint.txt.  This is a guard, not a parity claim: a kernel rewritten without
    a change of arithmetic must leave every line alone (round 4: the Cholesky through LDS, the streaming stores, the prefetch depths
    all did); a change that is MEANT to move bits regenerates the file with `python scripts/fingerprint.py` and says so."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("fingerprint", os.path.join(os.path.dirname(here), "scripts", "fingerprint.py"))
    fp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fp)
    want = open(os.path.join(here, "golden", "product_fingerprint.txt")).read().split("\n")
    want = [w for w in want if w.strip()]
    assert fp.lines() == want
