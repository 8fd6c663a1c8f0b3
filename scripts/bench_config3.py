#!/usr/bin/env python3
"""BASELINE.json configs[2] at its own size: the reference's data/chr22_subset50.bed is 1 066 557 SNPs x 64 samples.  The
GPU box has no /root/reference, so the committed 120 000-SNP slice (tests/golden/chr22_subset50_120k.npz) is tiled to the
full row count.  N = 64 is far below the sample tiles of the GEMMs (row pitch 256 B int8 / 1024 samples packed): this
measures what the 4x / 16x padded sweeps cost, next to upload + QC.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genomic_pca_amd as g          # noqa: E402
from genomic_pca_amd import _lib     # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "chr22_subset50_120k.npz"))
rows = z["bed_rows"]; n = int(z["n_samples"])
M = 1_066_557
bed = np.tile(rows, (M // rows.shape[0] + 1, 1))[:M]
out = {"workload": f"chr22_subset50-shaped: {M} SNPs x {n} samples (.bed rows of the committed slice, tiled), --eigensnp k=20, QC defaults"}
only = os.environ.get("CFG3_ONLY")          # e.g. "i8/int8": one path only (for a rocprofv3 kernel trace)
for name, prec, store in (("i8/int8", _lib.PREC_I8_EXACT, _lib.STORE_INT8), ("i8/2bit", _lib.PREC_I8_EXACT, _lib.STORE_2BIT),
                          ("f32/int8", _lib.PREC_F32_MFMA, _lib.STORE_INT8)):
    if only and name != only:
        continue
    with g.GpcaEngine(precision=prec, storage=store) as e:
        t0 = time.perf_counter(); e.upload_bed2bit(bed, n); t_up = time.perf_counter() - t0
        t0 = time.perf_counter(); e.snp_stats(g.QcConfig(), fetch=False); t_qc = time.perf_counter() - t0
        e.rsvd(20, 10, 2, seed=2025)
        e.enable_timings(True); e.reset_timings()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            e.rsvd(20, 10, 2, seed=2025)
        dt = (time.perf_counter() - t0) / reps
        tim = e.timings()
        out[name] = {"upload_s": t_up, "snp_stats_s": t_qc, "rsvd_ms": dt * 1e3, "pca_snps": e.num_pca_snps(),
                     "gemm_ms_per_call": (tim["gemm_GQ"]["total_ms"] + tim["gemm_GtT"]["total_ms"]) / reps,
                     "gemm_launch_us": {k: v["total_ms"] / v["launches"] * 1e3 for k, v in tim.items() if k.startswith("gemm")},
                     "stages_us_per_call": {k: round(v["total_ms"] / reps * 1e3, 1) for k, v in tim.items()},
                     "genotypes_per_s": M * n / dt, "top_eigenvalues": [float(x) for x in e.eigenvalues()[:3]]}
print(json.dumps(out))
