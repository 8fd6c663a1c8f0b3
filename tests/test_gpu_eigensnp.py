"""the stages of EigenSNPCoreAlgorithm::compute_pca (main.rs:311-327, 359-366) through the C ABI's section f3
(gpca_copy_rows, gpca_set_sample_mask, gpca_set_condensed_basis, gpca_rsvd_condensed, gpca_refine).  The algorithm lives in the
un-vendored efficient_pca crate: parity with the crate is UNPINNED; what is pinned here is (i) every device stage against a numpy
restatement of the same stage with the same sketches (oracle.eigensnp_*), and (ii) the end result against exact PCA."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _modes(prec, store):
    from genomic_pca_amd import _lib
    return dict(precision=_lib.PREC_I8_EXACT if prec == "i8" else _lib.PREC_F32_MFMA,
                storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8)


def _data(gpca, oracle, M=6000, N=400, P=6, seed=4, fst=0.3):
    G = oracle.synth_genotypes(M, N, seed, gpca.synth_thresholds(M, P, seed=seed, fst=fst))
    G[::37] = 1                                                   # monomorphic rows: QC drops them inside the blocks
    return G


@pytest.mark.parametrize("store", ["int8", "2bit"])
def test_copy_rows_makes_a_block_a_matrix_of_its_own(gpca, oracle, store):
    G = _data(gpca, oracle, 3000, 333)
    with gpca.GpcaEngine(**_modes("i8", store)) as e, gpca.GpcaEngine(**_modes("i8", store)) as sub:
        e.upload_genotypes_i8(G)
        st = e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6))
        # growing, shrinking and growing again: a smaller block reuses the buffers of a larger one (no hipMalloc per LD block);
        # every block must give the bits a fresh handle gives for the same rows
        for r0, n in ((0, 1), (17, 500), (2999, 1), (1000, 2000), (40, 130), (0, 3000), (2000, 900), (5, 129)):
            sub.copy_rows_from(e, r0, n)
            assert sub.dims() == (n, 333) and np.array_equal(sub.download_genotypes_i8(), G[r0:r0 + n])
            if n < 100:
                continue
            sub.set_standardization(st["mu"][r0:r0 + n], st["sigma"][r0:r0 + n], st["keep"][r0:r0 + n])
            sub.rsvd(5, 8, 2, seed=r0 + 1)
            got = (sub.eigenvalues(), sub.scores(f64=True), sub.loadings(), sub.transform())
            with gpca.GpcaEngine(**_modes("i8", store)) as fresh:
                fresh.upload_genotypes_i8(G[r0:r0 + n])
                fresh.set_standardization(st["mu"][r0:r0 + n], st["sigma"][r0:r0 + n], st["keep"][r0:r0 + n])
                fresh.rsvd(5, 8, 2, seed=r0 + 1)
                want = (fresh.eigenvalues(), fresh.scores(f64=True), fresh.loadings(), fresh.transform())
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (r0, n)
        with pytest.raises(gpca.GpcaError):
            sub.copy_rows_from(e, 2990, 20)                        # past the end
        with pytest.raises(gpca.GpcaError):
            sub.copy_rows_from(sub, 0, 1)                          # onto itself
    with gpca.GpcaEngine(**_modes("i8", "int8")) as a, gpca.GpcaEngine(**_modes("i8", "2bit")) as b:
        a.upload_genotypes_i8(G)
        with pytest.raises(gpca.GpcaError):
            b.copy_rows_from(a, 0, 10)                             # different storage modes


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit"), ("f32", "int8")])
def test_local_basis_on_a_sample_subset(gpca, oracle, prec, store):
    """gpca_set_sample_mask: the basis (loadings) is learnt from the subset's columns only, gpca_transform projects every
    sample -- against the numpy restatement with the same sketch, for a block whose row range holds SNPs of other blocks."""
    G = _data(gpca, oracle)
    M, N = G.shape
    r0, r1, c = 1200, 2100, 7
    rng = np.random.default_rng(3)
    member = np.zeros(r1 - r0, np.uint8); member[rng.choice(r1 - r0, 700, replace=False)] = 1
    mask = np.zeros(N, np.uint8); mask[rng.choice(N, 150, replace=False)] = 1
    with gpca.GpcaEngine(**_modes(prec, store)) as e, gpca.GpcaEngine(**_modes(prec, store)) as sub:
        e.upload_genotypes_i8(G)
        st = e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6))
        keep = st["keep"][r0:r1] & member
        sub.copy_rows_from(e, r0, r1 - r0)
        sub.set_standardization(st["mu"][r0:r1], st["sigma"][r0:r1], keep)
        sub.set_sample_mask(mask)
        sub.rsvd(c, 10, 2, seed=11)
        U, feats, sc = sub.loadings().astype(np.float64), sub.transform(), sub.scores(f64=True)
        assert np.all(sc[mask == 0] == 0) and np.any(feats[mask == 0] != 0)      # learnt on the subset, projected for everyone
        sub.set_sample_mask(None)
        sub.rsvd(c, 10, 2, seed=11)
        assert np.all(np.any(sub.scores(f64=True) != 0, axis=1))                 # the mask is gone
    r, b = oracle.scale_shift(st["mu"][r0:r1], st["sigma"][r0:r1], keep)
    A = oracle.standardized_dense(G[r0:r1], N, r, b)
    Uo, fo = oracle.eigensnp_local_basis(A, mask, c, 10, 2, 11)
    kept = keep.astype(bool)
    tol = 1e-4
    assert oracle.max_abs_dpc(U, Uo[kept]) < tol
    assert oracle.max_abs_dpc(feats, fo) < tol
    assert np.allclose(U.T @ U, np.eye(c), atol=1e-5)


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit"), ("f32", "int8")])
def test_condensed_global_pca_and_refinement(gpca, oracle, prec, store):
    """gpca_rsvd_condensed + gpca_refine against the numpy restatement, given the same block-diagonal basis W: the condensed
    matrix C* = W^T A is never formed on the device (its products run through the genotype GEMMs); initial scores, refined
    scores, loadings and eigenvalues agree, and a second refinement pass moves on from the first."""
    G = _data(gpca, oracle)
    M, N = G.shape
    K, cmax = 5, 4
    rng = np.random.default_rng(8)
    bounds = [0, 700, 1500, 1501, 2600, 4000, 5200, 6000]          # 7 blocks, one of a single SNP; rows 5200.. stay outside
    W = np.zeros((M, cmax), np.float32); feat0 = np.full(M, -1, np.int32); R = 0
    with gpca.GpcaEngine(**_modes(prec, store)) as e:
        e.upload_genotypes_i8(G)
        st = e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6))
        keepb = st["keep"].astype(bool)
        for a, z in zip(bounds[:-2], bounds[1:-1]):
            rows = np.nonzero(keepb[a:z])[0] + a
            c = min(cmax, len(rows))
            if c == 0:
                continue
            W[rows, :c] = rng.standard_normal((len(rows), c)).astype(np.float32) / np.sqrt(len(rows))
            feat0[rows] = R; R += c
        keep2 = st["keep"].copy(); keep2[5200:] = 0; keep2[feat0 < 0] = 0
        e.set_standardization(st["mu"], st["sigma"], keep2)
        e.set_condensed_basis(W, feat0, R)
        e.rsvd_condensed(K, 10, 2, seed=5)
        s0, ev0 = e.scores(f64=True), e.eigenvalues()
        for _ in range(4):                                         # (blocks with fewer features than cmax sit before larger ones:
            e.rsvd_condensed(K, 10, 2, seed=5)                     #  every block writes its own features only -- same bits every time)
            assert np.array_equal(e.scores(f64=True), s0)
        with pytest.raises(gpca.GpcaError):
            e.loadings()                                           # scores only
        e.refine(s0)
        s1, l1, ev1 = e.scores(f64=True), e.loadings().astype(np.float64), e.eigenvalues()
        e.refine(s1)
        s2, ev2 = e.scores(f64=True), e.eigenvalues()
    r, b = oracle.scale_shift(st["mu"], st["sigma"], keep2)
    A = oracle.standardized_dense(G, N, r, b)
    Wd = np.zeros((M, R))
    for i in np.nonzero(feat0 >= 0)[0]:
        Wd[i, feat0[i]:feat0[i] + cmax] += W[i, :min(cmax, R - feat0[i])]
    O1 = oracle.eigensnp_global_and_refine(A, Wd, K, 10, 2, 5, refine_passes=1)
    O2 = oracle.eigensnp_global_and_refine(A, Wd, K, 10, 2, 5, refine_passes=2)
    tol = 1e-4
    assert oracle.max_abs_dpc(s0, O1["initial_scores"]) < tol
    assert oracle.max_abs_dpc(s1, O1["scores"]) < tol and np.max(np.abs(ev1 - O1["eigenvalues"][:K]) / O1["eigenvalues"][:K]) < tol
    assert oracle.max_abs_dpc(l1, O1["loadings"][keep2.astype(bool)]) < tol
    assert oracle.max_abs_dpc(s2, O2["scores"]) < tol and np.max(np.abs(ev2 - O2["eigenvalues"][:K]) / O2["eigenvalues"][:K]) < tol
    assert np.all(ev2 >= ev1 * (1 - 1e-9))                          # a refinement pass is a power step: Ritz values do not decrease
    assert np.allclose(l1.T @ l1, np.eye(K), atol=1e-5)
    assert ev0.shape == (K,)


@pytest.mark.parametrize("prec,store", [("i8", "int8"), ("i8", "2bit")])
def test_compute_pca_multi_stage_end_to_end(gpca, oracle, prec, store):
    """EigenSNPCoreAlgorithm(cfg).compute_pca(accessor, ld_blocks, local_stage=True) with all 14 config fields acting: 12 LD blocks,
    local bases learnt on a 160-sample subset, one refinement pass.  The population PCs match exact PCA and the one-stage global
    path; the accessor comes back as it was; two refinement passes are at least as good as one."""
    G = _data(gpca, oracle, M=8000, N=400, P=5, seed=9, fst=0.3)
    M, N = G.shape
    K = 6
    with gpca.GpcaEngine(**_modes(prec, store)) as e:
        e.upload_genotypes_i8(G)
        st = e.snp_stats(gpca.QcConfig(0.9, 0.01, 1e-6))
        acc = gpca.MicroarrayGenotypeAccessor(e)
        D = acc.num_pca_snps()
        edges = np.linspace(0, D - 400, 13).astype(int)             # the last 400 PCA SNPs are in no block
        blocks = [gpca.LdBlockSpecification(f"b{i}", list(range(edges[i], edges[i + 1]))) for i in range(12)]
        blocks += [gpca.LdBlockSpecification("one_snp", [D - 300]), gpca.LdBlockSpecification("three_snps", [D - 200, D - 190, D - 180]),
                   gpca.LdBlockSpecification("empty", [])]   # blocks smaller than components_per_ld_block, and an empty one
        cfg = gpca.EigenSNPCoreAlgorithmConfig(target_num_global_pcs=K, components_per_ld_block=5, subset_factor_for_local_basis_learning=0.4,
                                               min_subset_size_for_local_basis_learning=50, max_subset_size_for_local_basis_learning=200,
                                               random_seed=77, refine_pass_count=1, collect_diagnostics=True)
        assert gpca.EigenSNPCoreAlgorithm.subset_size(cfg, N) == 160
        out, diag = gpca.EigenSNPCoreAlgorithm(cfg).compute_pca(acc, blocks, local_stage=True)
        assert diag["stage"] == "multi-stage" and diag["subset_size"] == 160 and diag["num_condensed_features"] == 60 + 1 + 3
        assert acc.num_pca_snps() == D                               # the accessor is as it was
        one, _ = gpca.EigenSNPCoreAlgorithm(cfg).compute_pca(acc, blocks)
        cfg2 = gpca.EigenSNPCoreAlgorithmConfig(**{**cfg.__dict__, "refine_pass_count": 2})
        two, _ = gpca.EigenSNPCoreAlgorithm(cfg2).compute_pca(acc, blocks, local_stage=True)
        rows = e.pca_snp_rows()
    used = np.concatenate([rows[:edges[-1]], rows[[D - 300, D - 200, D - 190, D - 180]]])
    keep = np.zeros(M, np.uint8); keep[used] = 1
    r, b = oracle.scale_shift(st["mu"], st["sigma"], keep)
    E = oracle.exact_pca(G, N, r, b, K)
    assert out.final_sample_principal_component_scores.shape == (N, K) and out.final_snp_principal_component_loadings.shape == (len(used), K)
    assert out.num_pca_snps_used == len(used)
    ns = 4                                                           # 5 populations: 4 structured PCs
    err1 = np.max(np.abs(out.final_principal_component_eigenvalues[:ns] - E["eigenvalues"][:ns]) / E["eigenvalues"][:ns])
    err2 = np.max(np.abs(two.final_principal_component_eigenvalues[:ns] - E["eigenvalues"][:ns]) / E["eigenvalues"][:ns])
    assert err1 < 2e-3 and err2 <= err1 * 1.001 and err2 < 5e-4
    assert oracle.max_abs_dpc(out.final_sample_principal_component_scores[:, :ns].astype(np.float64), E["scores"][:, :ns]) < 2e-2
    assert oracle.max_abs_dpc(two.final_sample_principal_component_scores[:, :ns].astype(np.float64), E["scores"][:, :ns]) < 5e-3
    assert oracle.max_abs_dpc(out.final_sample_principal_component_scores[:, :ns].astype(np.float64),
                              one.final_sample_principal_component_scores[:, :ns].astype(np.float64)) < 2e-2
    ld = out.final_snp_principal_component_loadings.astype(np.float64)
    assert np.allclose(ld.T @ ld, np.eye(K), atol=1e-4)
