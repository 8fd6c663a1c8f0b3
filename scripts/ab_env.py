#!/usr/bin/env python3
"""In-process A/B of two kernel-switch settings on the default workload: two engines (the switches are read per handle at
gpca_create) hold the same matrix and take turns, so clock / thermal drift hits both alike.
usage: ab_env.py NAME=VALUE_A NAME=VALUE_B [rounds]      e.g.  ab_env.py GPCA_GQ_SLOTS=7 GPCA_GQ_SLOTS=6 12"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402

a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
M, N, k = 1_000_000, 10_000, 20
th = g.synth_thresholds(M, 3, seed=1)
engs = []
for setting in (a, b):
    name, val = setting.split("=")
    os.environ[name] = val
    e = g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if os.environ.get("GPCA_AB_STORAGE") == "2bit" else _lib.STORE_INT8)
    e.synth_genotypes(M, N, 1, th)          # (GPCA_PITCH_PAD is read when the matrix is allocated)
    os.environ.pop(name)
    e.snp_stats(g.QcConfig.none(), fetch=False)
    e.rsvd(k, 10, 2, 1)
    e.enable_timings(True)
    engs.append(e)
res = {a: [], b: []}
for r in range(rounds):
    for setting, e in zip((a, b), engs):
        e.reset_timings()
        for _ in range(3):
            e.rsvd(k, 10, 2, 1)
        t = e.timings()
        res[setting].append((t["gemm_GQ"]["total_ms"] / t["gemm_GQ"]["launches"], t["gemm_GtT"]["total_ms"] / t["gemm_GtT"]["launches"]))
for setting in (a, b):
    v = np.array(res[setting])
    print(f"{setting:28s} K1 {v[:, 0].mean():.4f} ms (min {v[:, 0].min():.4f}, sd {v[:, 0].std():.4f})   K2 {v[:, 1].mean():.4f} ms (min {v[:, 1].min():.4f})")
assert np.array_equal(engs[0].eigenvalues(), engs[1].eigenvalues()), "the two settings disagree"
