#!/usr/bin/env python3
"""Sample-count sweep at a fixed matrix size (about 1e10 genotypes, k = 20): how far from the headline shape's rates the
exact-integer path sits when the rows are short or very long.  One JSON line per shape on stdout.
usage: python scripts/shape_sweep.py [int8|2bit]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402

store = sys.argv[1] if len(sys.argv) > 1 else "int8"
per_b = 1.0 if store == "int8" else 0.25
for N in (64, 256, 1000, 2504, 5000, 10_000, 25_000, 50_000, 100_000, 250_000):
    M = min(int(1e10) // N, 8_000_000)
    M -= M % 128
    th = g.synth_thresholds(M, 3, seed=1)
    with g.GpcaEngine(precision=_lib.PREC_I8_EXACT, storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8) as e:
        e.synth_genotypes(M, N, 1, th)
        e.snp_stats(g.QcConfig.none(), fetch=False)
        e.rsvd(20, 10, 2, 1)
        e.enable_timings(True); e.reset_timings()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            e.rsvd(20, 10, 2, 1)
        dt = (time.perf_counter() - t0) / reps
        tim = e.timings()
    gq = tim["gemm_GQ"]["total_ms"] / tim["gemm_GQ"]["launches"]
    gt = tim["gemm_GtT"]["total_ms"] / tim["gemm_GtT"]["launches"]
    by = M * N * per_b
    print(json.dumps({"store": store, "M": M, "N": N, "rsvd_ms": round(dt * 1e3, 3), "genotypes_per_s": M * N / dt,
                      "K1_ms": round(gq, 4), "K2_ms": round(gt, 4), "K1_GBs": round(by / gq / 1e6, 1), "K2_GBs": round(by / gt / 1e6, 1),
                      "outside_the_gemms_ms": round(dt * 1e3 - 3 * (gq + gt), 3)}), flush=True)
