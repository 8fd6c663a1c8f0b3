#!/usr/bin/env python3
"""A whole-genome-sized tall matrix on one GPU: 64M SNPs x 1 000 samples as 2-bit codes (16 GB + ~50 GB of M-sized workspace).
Checks that nothing in the M-sized index arithmetic overflows: finite descending eigenvalues, orthogonal scores, unit-norm
orthogonal loadings on a row sample, the same bits on a second call.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402

M, N, k = int(os.environ.get("TALL_M", 64_000_000)), 1000, 10
th = g.synth_thresholds(M, 3, seed=1)
with g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT) as e:
    t0 = time.perf_counter(); e.synth_genotypes(M, N, 1, th); e.synchronize(); t_syn = time.perf_counter() - t0
    t0 = time.perf_counter(); e.snp_stats(g.QcConfig(), fetch=False); e.synchronize(); t_qc = time.perf_counter() - t0
    e.rsvd(k, 10, 2, 1)
    ev0, sc0 = e.eigenvalues(), e.scores(f64=True)
    t0 = time.perf_counter(); e.rsvd(k, 10, 2, 1); e.synchronize(); t_rsvd = time.perf_counter() - t0
    ev, sc = e.eigenvalues(), e.scores(f64=True)
    t0 = time.perf_counter(); ld = e.loadings(); t_ld = time.perf_counter() - t0
    n_pca = e.num_pca_snps()
ok = {"finite": bool(np.all(np.isfinite(ev)) and np.all(np.isfinite(sc))), "descending": bool(np.all(np.diff(ev) <= 0) and ev[-1] > 0),
      "same_bits_twice": bool(np.array_equal(ev, ev0) and np.array_equal(sc, sc0))}
gram = sc.T @ sc
ok["scores_orthogonal"] = bool(np.max(np.abs(gram - np.diag(np.diag(gram)))) < 1e-6 * gram[0, 0])
ok["scores_norms_are_eigenvalues"] = bool(np.allclose(np.diag(gram) / (N - 1), ev, rtol=1e-9))
lg = ld.astype(np.float64).T @ ld.astype(np.float64)
ok["loadings_orthonormal"] = bool(np.max(np.abs(lg - np.eye(k))) < 1e-4)
ok["last_rows_nonzero"] = bool(np.any(ld[-1000:] != 0))
print(json.dumps({"shape": f"{M} x {N} 2-bit", "pca_snps": int(n_pca), "synth_s": round(t_syn, 2), "snp_stats_s": round(t_qc, 3), "rsvd_s": round(t_rsvd, 3),
                  "fetch_loadings_s": round(t_ld, 2), "genotypes_per_s": M * N / t_rsvd, "top_eigenvalues": [float(x) for x in ev[:3]], "checks": ok}))
sys.exit(0 if all(ok.values()) else 1)
