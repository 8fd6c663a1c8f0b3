#!/usr/bin/env python3
"""In-process A/B of two engine configurations on the default workload: two engines hold the same matrix and take turns, so clock /
thermal drift hits both alike (the first and the second engine of a process sit at different places in HBM: compare within a position,
scripts/gpu.sh runs both orders).  A configuration is a comma-separated list of GpcaEngine keyword settings (gpca_config.reserved):
usage: ab_env.py CONFIG_A CONFIG_B [rounds]      e.g.  ab_env.py simple=1 default 8      ab_env.py gq_waves=512,gtt_waves=1024 default
keys: simple, compact, narrow, spin_sync (0 / 1), gq_waves, gtt_waves, storage (int8 / 2bit), planes (3 / 4)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402


def make(setting):
    kw = dict(kv.split("=") for kv in setting.split(",") if "=" in kv)
    flags = 0
    if int(kw.get("simple", 0)):
        flags |= _lib.CFG_SIMPLE_KERNELS
    if int(kw.get("compact", 1)) == 0:
        flags |= _lib.CFG_NO_COMPACT
    if int(kw.get("narrow", 1)) == 0:
        flags |= _lib.CFG_NO_NARROW
    if int(kw.get("spin_sync", 1)) == 0:
        flags |= _lib.CFG_NO_SPIN_SYNC
    return g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if kw.get("storage") == "2bit" else _lib.STORE_INT8,
                        digit_planes=int(kw.get("planes", 0)), flags=flags, gq_waves=int(kw.get("gq_waves", 0)), gtt_waves=int(kw.get("gtt_waves", 0)))


a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
M, N, k = 1_000_000, 10_000, 20
th = g.synth_thresholds(M, 3, seed=1)
engs = []
for setting in (a, b):
    e = make(setting)
    e.synth_genotypes(M, N, 1, th)
    e.snp_stats(g.QcConfig.none(), fetch=False)
    e.rsvd(k, 10, 2, 1)
    e.enable_timings(True)
    engs.append(e)
res = {0: [], 1: []}
for r in range(rounds):
    for i, e in enumerate(engs):
        e.reset_timings()
        for _ in range(3):
            e.rsvd(k, 10, 2, 1)
        t = e.timings()
        res[i].append((t["gemm_GQ"]["total_ms"] / t["gemm_GQ"]["launches"], t["gemm_GtT"]["total_ms"] / t["gemm_GtT"]["launches"]))
for i, setting in enumerate((a, b)):
    v = np.array(res[i])
    print(f"{setting:28s} K1 {v[:, 0].mean():.4f} ms (min {v[:, 0].min():.4f}, sd {v[:, 0].std():.4f})   K2 {v[:, 1].mean():.4f} ms (min {v[:, 1].min():.4f})")
same = np.array_equal(engs[0].eigenvalues(), engs[1].eigenvalues())
print("eigenvalues of the two configurations:", "same bits" if same else f"differ, max rel {np.max(np.abs(engs[0].eigenvalues() - engs[1].eigenvalues()) / engs[1].eigenvalues()):.2e}")
