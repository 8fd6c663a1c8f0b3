"""Host-link-inclusive rates at the drop-in boundary ("PCIe-inclusive note"): how long the C ABI's
host-buffer entry points take for BASELINE.json configs[1] (1M SNPs x 10k samples), next to the resident gpca_rsvd call.
usage: python scripts/pcie_rates.py  (on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomic_pca_amd as g  # noqa: E402
from genomic_pca_amd import _lib  # noqa: E402

M, N, k = 1_000_000, 10_000, 20
out = {"shape": f"{M} x {N}", "k": k}
th = g.synth_thresholds(M, 3, seed=1)
with g.GpcaEngine(storage=_lib.STORE_INT8) as e:
    e.synth_genotypes(M, N, 1, th)
    t0 = time.perf_counter(); G = e.download_genotypes_i8(); out["download_i8_s"] = time.perf_counter() - t0
bpr = (N + 3) // 4
code = np.array([3, 2, 0, 1], np.uint8)          # dosage 0,1,2 -> .bed codes 11,10,00 (count_a1); missing (01) does not occur here
bed = np.zeros((M, bpr), np.uint8)
for s in range(4):
    bed |= code[G[:, s::4]] << np.uint8(2 * s)
for store, name in ((_lib.STORE_INT8, "int8"), (_lib.STORE_2BIT, "2bit")):
    with g.GpcaEngine(storage=store) as e:
        for rep in range(2):                       # second call: pages touched, staging allocated
            t0 = time.perf_counter(); e.upload_genotypes_i8(G); e.synchronize(); dt = time.perf_counter() - t0
        out[f"upload_i8_to_{name}_s"] = dt; out[f"upload_i8_to_{name}_GBs"] = M * N / dt / 1e9
        for rep in range(2):
            t0 = time.perf_counter(); e.upload_bed2bit(bed, N); e.synchronize(); dt = time.perf_counter() - t0
        out[f"upload_bed_to_{name}_s"] = dt; out[f"upload_bed_to_{name}_GBs"] = M * bpr / dt / 1e9
        t0 = time.perf_counter(); e.snp_stats(g.QcConfig.none(), fetch=False); e.synchronize(); out[f"snp_stats_{name}_s"] = time.perf_counter() - t0
        e.rsvd(k, 10, 2, 1)
        t0 = time.perf_counter(); e.rsvd(k, 10, 2, 1); e.synchronize(); out[f"rsvd_{name}_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); sc = e.scores(); ev = e.eigenvalues(); ld = e.loadings(); out[f"fetch_results_{name}_s"] = time.perf_counter() - t0
# out of core from host memory: every pass crosses the link (2-bit rows, .bed bytes recoded on the device)
with g.GpcaEngine(storage=_lib.STORE_2BIT) as e:
    e.stream_open(g.PanelSource.host_bed(lambda r0, r: bed[r0:r0 + r]), M, N)
    e.snp_stats(g.QcConfig.none(), fetch=False)
    e.rsvd(k, 10, 2, 1)
    t0 = time.perf_counter(); e.rsvd(k, 10, 2, 1); e.synchronize(); dt = time.perf_counter() - t0
    out["streamed_host_bed_rsvd_s"] = dt; out["streamed_host_bed_link_GBs"] = 4 * M * bpr / dt / 1e9
    e.stream_set_cache(-1)
    e.rsvd(k, 10, 2, 1)
    t0 = time.perf_counter(); e.rsvd(k, 10, 2, 1); e.synchronize(); out["streamed_host_bed_all_cached_rsvd_s"] = time.perf_counter() - t0
print(json.dumps(out))
