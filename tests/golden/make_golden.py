#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference (Rust; no toolchain here) cannot be run and holds no golden vectors for
this path, so these fixtures pin the ORACLE's outputs on seeded inputs: a later change to the
oracle or to the synthetic generator that alters any number is caught by tests/test_oracle.py, and the GPU
parity tests compare against the same arrays.  Run from the repo root:  python tests/golden/make_golden.py

Also extracts two slices of the reference's own data fixture data/chr22_subset50.bed(.zip) (64 samples; a data
file, not source) when /root/reference is present: 2000 SNPs with their decoded int8 dosages, and 120 000 SNPs (1.9 MB
of .bed bytes, BASELINE.json configs[2]'s data at a size the CPU oracle handles in seconds) with the .fam sample ids
and the oracle's QC + randomized-PCA + exact-PCA outputs on it.  data/chr22_subset50.bim.zip is missing from the
reference checkout (.MISSING_LARGE_BLOBS), so the tests synthesise a .bim (chromosome 22, increasing positions).
"""
import io
import os
import sys
import zipfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
import genomic_pca_amd as g               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synth_fixture():
    M, N, P, k, seed = 2048, 192, 8, 6, 7
    th = g.synth_thresholds(M, P, seed=seed, fst=0.25)
    G = O.synth_genotypes(M, N, seed, th)
    st = O.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = O.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = O.rsvd(G, N, r, b, k, 10, 2, seed=seed)
    E = O.exact_pca(G, N, r, b, k)
    np.savez_compressed(os.path.join(HERE, "synth_2048x192.npz"), G=G, thresh=th, mu=st["mu"], sigma=st["sigma"],
                        keep=st["keep"], counts=st["counts"], k=k, seed=seed, eigenvalues=R["eigenvalues"],
                        scores=R["scores"], loadings=R["loadings"], exact_eigenvalues=E["eigenvalues"],
                        exact_scores=E["scores"])
    print("synth_2048x192.npz", os.path.getsize(os.path.join(HERE, "synth_2048x192.npz")))


def decode_bed_rows(rows: np.ndarray, n: int) -> np.ndarray:
    """PLINK 2-bit, LSB-first, count_a1 (prepare.rs:622-629): 00->2, 10->1, 11->0, 01->missing(-127).
    Byte layout as documented by the reference's tests/disk.py:89-135."""
    lut = np.array([2, -127, 1, 0], np.int8)
    out = np.empty((rows.shape[0], rows.shape[1] * 4), np.int8)
    for s in range(4):
        out[:, s::4] = lut[(rows >> (2 * s)) & 3]
    return out[:, :n]


def bed_fixture():
    zp = "/root/reference/data/chr22_subset50.bed.zip"
    fp = "/root/reference/data/chr22_subset50.fam.zip"
    if not (os.path.exists(zp) and os.path.exists(fp)):
        print("reference data not present; skipping BED slice")
        return
    fam = zipfile.ZipFile(fp).read(zipfile.ZipFile(fp).namelist()[0]).decode().strip().splitlines()
    n = len(fam)
    bed = zipfile.ZipFile(zp).read(zipfile.ZipFile(zp).namelist()[0])
    assert bed[:3] == b"\x6c\x1b\x01", "not a SNP-major PLINK .bed"
    bpr = (n + 3) // 4
    m_total = (len(bed) - 3) // bpr
    start, count = 500_000, 2000
    rows = np.frombuffer(bed, np.uint8, count * bpr, 3 + start * bpr).reshape(count, bpr).copy()
    G = decode_bed_rows(rows, n)
    np.savez_compressed(os.path.join(HERE, "chr22_subset50_slice.npz"), bed_rows=rows, n_samples=n, snp_start=start,
                        m_total=m_total, dosage_count_a1=G, iids=np.array([l.split()[1] for l in fam]))
    print("chr22_subset50_slice.npz", os.path.getsize(os.path.join(HERE, "chr22_subset50_slice.npz")), "N =", n, "M_total =", m_total)
    # configs[2] at test size: 120 000 consecutive SNPs, QC with the reference's effective defaults, k = 20
    start, count, k, seed = 300_000, 120_000, 20, 2025
    rows = np.frombuffer(bed, np.uint8, count * bpr, 3 + start * bpr).reshape(count, bpr).copy()
    G = decode_bed_rows(rows, n)
    st = O.snp_stats(G, n, 0.98, 0.01, 1e-6)
    r, b = O.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = O.rsvd(G, n, r, b, k, 10, 2, seed=seed)
    E = O.exact_pca(G, n, r, b, k)
    np.savez_compressed(os.path.join(HERE, "chr22_subset50_120k.npz"), bed_rows=rows, n_samples=n, snp_start=start,
                        iids=np.array([l.split()[1] for l in fam]), fids=np.array([l.split()[0] for l in fam]),
                        keep=st["keep"], reason=st["reason"], mu=st["mu"], sigma=st["sigma"], k=k, seed=seed,
                        eigenvalues=R["eigenvalues"], scores=R["scores"], exact_eigenvalues=E["eigenvalues"],
                        exact_scores=E["scores"])
    print("chr22_subset50_120k.npz", os.path.getsize(os.path.join(HERE, "chr22_subset50_120k.npz")), "kept", int(st["keep"].sum()),
          "rel d(eigenvalue) rsvd vs exact", float(np.max(np.abs(R["eigenvalues"] - E["eigenvalues"]) / E["eigenvalues"])))


if __name__ == "__main__":
    synth_fixture()
    bed_fixture()
