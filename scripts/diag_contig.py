#!/usr/bin/env python3
"""Where does a run with GPCA_CONTIG=1 (genotype storage from hipDeviceMallocContiguous) go wrong?  The generator's bytes against the
oracle's, the statistics and a randomized PCA against a plain-allocation engine, in both residencies, twice in one process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402
from oracle import oracle as O       # noqa: E402  (checker)

M, N = 6000, 700
th = g.synth_thresholds(M, 3, seed=7, fst=0.1)
G = O.synth_genotypes(M, N, 7, th)
for rep in range(2):
    for store, sname in ((_lib.STORE_INT8, "int8"), (_lib.STORE_2BIT, "2bit")):
        res = {}
        for contig in ("0", "1"):
            os.environ["GPCA_CONTIG"] = contig
            with g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=store) as e:
                e.synth_genotypes(M, N, 7, th)
                D = e.download_genotypes_i8()
                bad = np.argwhere(D != G)
                print(f"rep {rep} {sname} contig={contig}: generator vs oracle: {len(bad)} mismatches" +
                      (f"; first {bad[:5].tolist()}, rows touched {len(np.unique(bad[:, 0]))}, cols min/max {bad[:, 1].min()}/{bad[:, 1].max()}, "
                       f"device values {D[tuple(bad[:5].T)].tolist()} oracle {G[tuple(bad[:5].T)].tolist()}" if len(bad) else ""), flush=True)
                e.upload_genotypes_i8(G)
                D = e.download_genotypes_i8()
                print(f"    upload -> download: {int((D != G).sum())} mismatches", flush=True)
                st = e.snp_stats(g.QcConfig.none())
                e.rsvd(8, 10, 2, seed=1)
                res[contig] = (st["mu"].copy(), st["sigma"].copy(), e.eigenvalues().copy(), e.scores(f64=True).copy())
        for name, a, b in zip(("mu", "sigma", "eigenvalues", "scores"), res["0"], res["1"]):
            print(f"    {name}: contig == plain bitwise: {bool(np.array_equal(a, b))}  max|d| {float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))):.3e}", flush=True)
