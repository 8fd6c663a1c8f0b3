#!/usr/bin/env python3
"""The multi-stage EigenSNP algorithm (compute_pca(local_stage=True)) next to the one-stage global randomized PCA on the same
matrix: wall time of each stage and accuracy of both against each other.  One JSON line.
usage: python scripts/eigensnp_stages_bench.py [M] [N] [blocks]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 200
K = 10
th = g.synth_thresholds(M, 6, seed=1, fst=0.1)
out = {"shape": f"{M} x {N}", "ld_blocks": B, "K": K}
with g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_INT8) as e:
    e.synth_genotypes(M, N, 1, th)
    e.snp_stats(g.QcConfig())
    acc = g.MicroarrayGenotypeAccessor(e)
    D = acc.num_pca_snps()
    edges = np.linspace(0, D, B + 1).astype(int)
    blocks = [g.LdBlockSpecification(f"b{i}", np.arange(edges[i], edges[i + 1])) for i in range(B)]
    cfg = g.EigenSNPCoreAlgorithmConfig(target_num_global_pcs=K, collect_diagnostics=True)
    for name, local in (("one_stage_global", False), ("multi_stage", True)):
        for rep in range(2):
            t0 = time.perf_counter()
            res, diag = g.EigenSNPCoreAlgorithm(cfg).compute_pca(acc, blocks, local_stage=local)
            dt = time.perf_counter() - t0
        out[name + "_s"] = round(dt, 4)
        out[name + "_eigenvalues"] = [round(float(x), 4) for x in res.final_principal_component_eigenvalues[:6]]
        if local:
            out["diag"] = {k: v for k, v in diag.items() if k in ("num_condensed_features", "subset_size", "refine_passes")}
            sc_m = res.final_sample_principal_component_scores.astype(np.float64)
        else:
            sc_g = res.final_sample_principal_component_scores.astype(np.float64)
nrm = lambda X: X / np.linalg.norm(X, axis=0)
a, b = nrm(sc_m[:, :5]), nrm(sc_g[:, :5])
out["max_abs_dPC_multi_vs_one_stage_top5"] = float(np.max(np.abs(a * np.sign(np.sum(a * b, axis=0)) - b)))
print(json.dumps(out))
