#!/usr/bin/env python3
"""How accurate are three digit planes (24-bit fixed point per column of the skinny operand) next to four (28-bit) on the packed
kernels?  max|dPC| of scores and loadings and the relative eigenvalue error against the f64 checker (oracle.rsvd: same sketch,
Householder QR + LAPACK SVD) on the parity suite's shapes, for both settings.  The oracle is the checker only.  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402
from oracle import oracle as O       # noqa: E402  (checker)

out = {}
for M, N, P, k, fst in ((4096, 512, 12, 8, 0.2), (20000, 1000, 16, 10, 0.2), (3000, 1500, 10, 6, 0.2), (999, 257, 8, 4, 0.2), (60000, 2000, 3, 20, 0.05),
                        (6000, 700, 48, 40, 0.3)):
    th = g.synth_thresholds(M, P, seed=1, fst=fst)
    G = O.synth_genotypes(M, N, 1, th)
    st = O.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = O.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = O.rsvd(G, N, r, b, k, 10, 2, seed=1)
    kk = min(k, P - 1)                       # the structured PCs (the rest is the noise bulk: direction-degenerate)
    row = {}
    for planes in (4, 3):
        with g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT, digit_planes=planes) as e:
            e.upload_genotypes_i8(G); e.snp_stats(g.QcConfig.none()); e.rsvd(k, 10, 2, seed=1)
            row[f"planes{planes}"] = {
                "max_abs_dPC_scores": O.max_abs_dpc(e.scores(f64=True)[:, :kk], R["scores"][:, :kk]),
                "max_abs_dPC_loadings": O.max_abs_dpc(e.loadings().astype(np.float64)[:, :kk], R["loadings"][:, :kk]),
                "max_rel_d_eigenvalue": float(np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]))}
    with g.GpcaEngine(precision=_lib.PREC_F32_MFMA, storage=_lib.STORE_2BIT) as e:
        e.upload_genotypes_i8(G); e.snp_stats(g.QcConfig.none()); e.rsvd(k, 10, 2, seed=1)
        row["f32_mfma"] = {"max_abs_dPC_scores": O.max_abs_dpc(e.scores(f64=True)[:, :kk], R["scores"][:, :kk]),
                           "max_rel_d_eigenvalue": float(np.max(np.abs(e.eigenvalues() - R["eigenvalues"]) / R["eigenvalues"]))}
    out[f"{M}x{N} k={k} ({kk} structured PCs)"] = row
print(json.dumps(out))
