#!/usr/bin/env python3
"""BASELINE.json configs[2] end to end through the command line: a chr22_subset50-sized PLINK fileset (1 066 557 SNPs x 64
samples: the committed 120k-SNP slice of the reference's data, tiled; synthesised .bim) -> --eigensnp -> three TSV files,
with the native host and with the Python mirror.  Prints wall times as one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "tests", "golden", "chr22_subset50_120k.npz"))
rows = z["bed_rows"]; M = 1_066_557
bed = np.tile(rows, (M // rows.shape[0] + 1, 1))[:M]
d = tempfile.mkdtemp()
pre = os.path.join(d, "chr22_subset50")
with open(pre + ".bed", "wb") as f:
    f.write(b"\x6c\x1b\x01"); f.write(bed.tobytes())
with open(pre + ".fam", "w") as f:
    for fid, iid in zip(z["fids"], z["iids"]):
        f.write(f"{fid} {iid} 0 0 0 -9\n")
pos = 16_050_000 + 25 * np.arange(M)
with open(pre + ".bim", "w") as f:
    f.write("".join(f"22\t22:{p}\t0\t{p}\tA\tG\n" for p in pos))
ld = os.path.join(d, "ld.txt")
open(ld, "w").write("22 1 500000000\n")
out = {"workload": f"{M} SNPs x 64 samples .bed, --eigensnp --eigensnp-k-global 20, one genome-wide LD block"}
common = ["--eigensnp", "--bed-file", pre + ".bed", "--ld-block-file", ld, "--eigensnp-k-global", "20"]
for name, cmd in (("native_host", [os.path.join(ROOT, "genomic_pca_amd", "bin", "genomic_pca")]),
                  ("python_mirror", [sys.executable, "-m", "genomic_pca_amd"])):
    for rep in range(2):                      # second run: page cache warm, GPU driver up
        t0 = time.perf_counter()
        r = subprocess.run(cmd + common + ["--out", os.path.join(d, name)], capture_output=True, text=True, cwd=ROOT)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
    out[name + "_wall_s"] = round(dt, 3)
    out[name + "_log"] = [ln for ln in r.stderr.strip().split("\n") if "genomic_pca" in ln][-2:]
same = all(open(os.path.join(d, "native_host" + s)).read() == open(os.path.join(d, "python_mirror" + s)).read()
           for s in (".eigensnp.pca.tsv", ".eigenvalues.tsv", ".eigensnp.loadings.tsv"))
out["identical_output_files"] = same
out["loadings_rows"] = sum(1 for _ in open(os.path.join(d, "native_host.eigensnp.loadings.tsv"))) - 1
print(json.dumps(out))
