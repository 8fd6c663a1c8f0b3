"""Host-side file formats either side of the hot path: PLINK BED/BIM/FAM, LD-block
files, a minimal VCF genotype reader, and the TSV writers.  Text parsing stays on the host (tiny); genotype
bytes go to the GPU untouched (2-bit BED payload or int8 dosages) and are decoded / QC'd there.

Reference behaviour restated (file:line):
  * BED: 3-byte magic 6c 1b 01 (SNP-major), ceil(N/4) bytes per SNP, 2 bits per sample LSB-first
    (tests/disk.py:89-135); .bim chrom/sid/bp columns, .fam iid column (prepare.rs:940-970 via bed_reader).
  * LD blocks: prepare.rs:1565-1616 (skip '#', 'chr\\t', 'chromosome\\t' headers; tag 'chr:start-end'; chromosome
    names lower-cased with a leading 'chr' stripped); SNP -> first matching block (prepare.rs:1447-1463).
  * VCF: biallelic single-base REF/ALT only (vcf.rs:109-121); GT 'a/b' or 'a|b' with alleles 0/1, anything else
    drops the variant (vcf.rs:52-63, 153-240); MAF filter default 0.01 (vcf.rs:244-266); id chr:pos:ref:alt.
  * writers: main.rs:696-839 ('{:.6}' fixed formatting, the exact headers and file suffixes).
"""
from __future__ import annotations

import gzip
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

BED_MAGIC = b"\x6c\x1b\x01"


# ------------------------------------------------------------------------------------------------ PLINK
@dataclass
class PlinkFileset:
    bed_rows: np.ndarray          # uint8 [M, ceil(N/4)] (a memory map of the .bed payload)
    n_samples: int
    sample_ids: List[str]         # .fam IID
    variant_ids: List[str]        # .bim sid
    chromosomes: List[str]        # .bim chrom
    positions: np.ndarray         # .bim bp (int64)


def _strip_ext(path: str) -> str:
    for ext in (".bed", ".bim", ".fam"):
        if path.endswith(ext):
            return path[: -len(ext)]
    return path


def read_plink(bed_path: str) -> PlinkFileset:
    prefix = _strip_ext(bed_path)
    iids = []
    with open(prefix + ".fam") as f:
        for line in f:
            p = line.split()
            if p:
                iids.append(p[1] if len(p) > 1 else p[0])
    sids, chroms, pos = [], [], []
    with open(prefix + ".bim") as f:
        for line in f:
            p = line.split()
            if len(p) >= 4:
                chroms.append(p[0]); sids.append(p[1]); pos.append(int(p[3]))
    n, m = len(iids), len(sids)
    bpr = (n + 3) // 4
    with open(prefix + ".bed", "rb") as f:
        magic = f.read(3)
    if magic != BED_MAGIC:
        raise ValueError(f"{prefix}.bed: not a SNP-major PLINK .bed (magic {magic.hex()})")
    size = os.path.getsize(prefix + ".bed")
    if size != 3 + m * bpr:
        raise ValueError(f"{prefix}.bed: size {size} does not match {m} SNPs x {n} samples")
    rows = np.memmap(prefix + ".bed", dtype=np.uint8, mode="r", offset=3, shape=(m, bpr))
    return PlinkFileset(rows, n, iids, sids, chroms, np.asarray(pos, np.int64))


def write_plink(prefix: str, dosage_count_a1: np.ndarray, sample_ids: Sequence[str], variant_ids: Sequence[str],
                chromosomes: Sequence[str], positions: Sequence[int]) -> None:
    """Test/fixture helper: int8 [M, N] count-A1 dosages (-127 missing) -> .bed/.bim/.fam."""
    g = np.asarray(dosage_count_a1, np.int8)
    m, n = g.shape
    code = np.full(g.shape, 1, np.uint8)          # 01 = missing
    code[g == 2] = 0; code[g == 1] = 2; code[g == 0] = 3
    pad = (-n) % 4
    if pad:
        code = np.concatenate([code, np.zeros((m, pad), np.uint8)], axis=1)
    c4 = code.reshape(m, -1, 4)
    rows = (c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)).astype(np.uint8)
    with open(prefix + ".bed", "wb") as f:
        f.write(BED_MAGIC); f.write(rows.tobytes())
    with open(prefix + ".bim", "w") as f:
        for c, s, p in zip(chromosomes, variant_ids, positions):
            f.write(f"{c}\t{s}\t0\t{p}\tA\tG\n")
    with open(prefix + ".fam", "w") as f:
        for s in sample_ids:
            f.write(f"{s}\t{s}\t0\t0\t0\t-9\n")


# ------------------------------------------------------------------------------------------------ LD blocks
def normalize_chromosome_name(name: str) -> str:
    """prepare.rs:1610-1616."""
    name = name.lower()
    while name.startswith("chr"):
        name = name[3:]
    return name


def parse_ld_block_file(path: str) -> List[Tuple[str, int, int, str]]:
    """prepare.rs:1565-1607 -> [(chrom, start, end, tag)]."""
    blocks = []
    with open(path) as f:
        for line in f:
            t = line.strip()
            if not t or t.startswith("#") or t.startswith("chr\t") or t.startswith("chromosome\t"):
                continue
            p = t.split()
            if len(p) < 3:
                continue
            c = normalize_chromosome_name(p[0])
            start, end = int(p[1]), int(p[2])
            blocks.append((c, start, end, f"{c}:{start}-{end}"))
    return blocks


def map_snps_to_ld_blocks(blocks, chromosomes: Sequence[str], positions: Sequence[int], qc_keep: np.ndarray):
    """prepare.rs:1424-1563.  Returns (keep mask restricted to SNPs inside some block, [(tag, original_rows)] sorted
    by tag).  Each kept SNP goes to the FIRST block (file order) that contains it."""
    qc_keep = np.asarray(qc_keep).astype(bool)
    pos = np.asarray(positions, np.int64)
    norm = np.array([normalize_chromosome_name(c) for c in chromosomes])
    assigned = np.full(len(pos), -1, np.int64)
    for bi, (c, start, end, _) in enumerate(blocks):
        hit = qc_keep & (assigned < 0) & (norm == c) & (pos >= start) & (pos <= end)
        assigned[hit] = bi
    keep = assigned >= 0
    by_tag = {}
    for bi, (_, _, _, tag) in enumerate(blocks):
        rows = np.nonzero(assigned == bi)[0]
        if len(rows):
            by_tag.setdefault(tag, []).extend(rows.tolist())
    return keep.astype(np.uint8), sorted(((t, sorted(r)) for t, r in by_tag.items()), key=lambda x: x[0])


def read_sample_keep_file(path: str) -> List[str]:
    with open(path) as f:
        return [ln.split()[0] for ln in f if ln.strip()]


# ------------------------------------------------------------------------------------------------ VCF
def _gt_to_dosage(gt: str) -> Optional[int]:
    """vcf.rs:52-63: exactly 3 bytes, separator / or |, alleles 0/1."""
    if len(gt) != 3 or gt[1] not in "/|":
        return None
    a, b = gt[0], gt[2]
    if a not in "01" or b not in "01":
        return None
    return (a == "1") + (b == "1")


def _dosages_gt_first(rest: bytes, ns: int) -> Optional[np.ndarray]:
    """All sample columns of one record at once when GT is the first FORMAT key (the layout of 1000 Genomes-style files):
    every sample field must start with a 3-byte genotype `a/b` or `a|b`, a, b in {0, 1}, followed by ':' or the column end
    (vcf.rs:52-63).  Returns int8 dosages, or None if any sample breaks the rule (the variant is dropped, as in the reference)."""
    a = np.frombuffer(rest, np.uint8)
    if len(rest) == 4 * ns - 1:                                   # FORMAT = GT only: fixed 4-byte stride
        b = np.frombuffer(rest + b"\t", np.uint8).reshape(ns, 4)
        c0, sep, c1, end = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        if not (end == 9).all():
            return None
    else:
        tabs = np.flatnonzero(a == 9)
        if len(tabs) != ns - 1:
            return None
        starts = np.concatenate(([0], tabs + 1))
        ends = np.concatenate((tabs, [len(a)]))
        if ((ends - starts) < 3).any():
            return None
        c0, sep, c1 = a[starts], a[starts + 1], a[starts + 2]
        long_ = (ends - starts) > 3
        if long_.any() and not (a[starts[long_] + 3] == 58).all():   # a 3-byte GT must be followed by ':'
            return None
    if not (((c0 == 48) | (c0 == 49)).all() and ((c1 == 48) | (c1 == 49)).all() and ((sep == 47) | (sep == 124)).all()):
        return None
    return ((c0 - 48) + (c1 - 48)).astype(np.int8)


def read_vcf(path: str, maf_threshold: float = 0.01):
    """Returns (sample_names, variant_ids, int8 [variants, samples]) following vcf.rs:65-286.  Records are parsed as bytes;
    when GT leads the FORMAT column the whole sample row is converted with a handful of numpy operations (a chr22-scale file of
    2 504 samples parses at text-I/O speed instead of one Python call per genotype); any other FORMAT order takes the
    per-sample path."""
    opener = gzip.open if path.endswith(".gz") else open
    samples: List[str] = []
    ids: List[str] = []
    rows: List[np.ndarray] = []
    with opener(path, "rb") as f:
        for line in f:
            if line.startswith(b"##"):
                continue
            line = line.rstrip(b"\r\n")
            if line.startswith(b"#CHROM"):
                samples = [x.decode() for x in line.split(b"\t")[9:]]
                if not samples:
                    raise ValueError(f"VCF header from {path} contains no samples.")      # vcf.rs:31-36
                continue
            p = line.split(b"\t", 9)
            if len(p) < 10:
                continue
            chrom, pos, ref, alt, fmt_col, rest = p[0], p[1], p[3], p[4], p[8], p[9]
            if len(ref) != 1 or len(alt) != 1 or b"," in alt:                              # vcf.rs:109-121
                continue
            fmt = fmt_col.split(b":")
            if b"GT" not in fmt:
                continue
            gi = fmt.index(b"GT")
            ns = len(samples)
            if gi == 0:
                d = _dosages_gt_first(rest, ns)
            else:                                                                          # GT not first: per-sample path
                d = np.empty(ns, np.int8)
                fields = rest.split(b"\t")
                ok = len(fields) == ns
                if ok:
                    for si, field in enumerate(fields):
                        parts = field.split(b":")
                        v = _gt_to_dosage(parts[gi].decode()) if gi < len(parts) else None
                        if v is None:                                                      # any bad GT drops the variant
                            ok = False
                            break
                        d[si] = v
                if not ok:
                    d = None
            if d is None:
                continue
            af = float(d.sum(dtype=np.int64)) / (2 * ns)
            if min(af, 1.0 - af) < maf_threshold:                                          # vcf.rs:244-266
                continue
            ids.append(f"{chrom.decode()}:{pos.decode()}:{ref.decode()}:{alt.decode()}")
            rows.append(d)
    G = np.stack(rows) if rows else np.zeros((0, len(samples)), np.int8)
    return samples, ids, G


# ------------------------------------------------------------------------------------------------ writers
def _fmt6(x) -> str:
    return f"{float(x):.6f}"


def write_principal_components(prefix: str, suffix: str, sample_names: Sequence[str], pcs: np.ndarray) -> str:
    """main.rs:696-762: header SampleID\\tPC1..; '{:.6}'.  suffix = 'vcf.pca.tsv' or 'eigensnp.pca.tsv'."""
    path = f"{prefix}.{suffix}"
    if pcs.shape[1] == 0:
        return path
    with open(path, "w") as f:
        f.write("SampleID" + "".join(f"\tPC{i}" for i in range(1, pcs.shape[1] + 1)) + "\n")
        for i, name in enumerate(sample_names):
            if i < pcs.shape[0]:
                f.write(name + "".join("\t" + _fmt6(v) for v in pcs[i]) + "\n")
            else:
                f.write(name + "\tNA" * pcs.shape[1] + "\n")
    return path


def write_eigenvalues(prefix: str, eigenvalues: Sequence[float]) -> str:
    """main.rs:765-784: header written even when empty."""
    path = f"{prefix}.eigenvalues.tsv"
    with open(path, "w") as f:
        f.write("PC\tEigenvalue\n")
        for i, v in enumerate(eigenvalues):
            f.write(f"{i + 1}\t{_fmt6(v)}\n")
    return path


def write_loadings(prefix: str, variant_ids: Sequence[str], chromosomes: Sequence[str], positions: Sequence[int],
                   loadings: np.ndarray) -> str:
    """main.rs:787-839."""
    path = f"{prefix}.eigensnp.loadings.tsv"
    if loadings.shape[1] == 0:
        return path
    if not (len(variant_ids) == len(chromosomes) == len(positions) == loadings.shape[0]) and len(variant_ids):
        raise ValueError("Mismatch in lengths of variant metadata and loadings matrix rows.")     # main.rs:817-824
    with open(path, "w") as f:
        f.write("VariantID\tChrom\tPos" + "".join(f"\tPC{i}_loading" for i in range(1, loadings.shape[1] + 1)) + "\n")
        for i in range(len(variant_ids)):
            f.write(f"{variant_ids[i]}\t{chromosomes[i]}\t{positions[i]}" + "".join("\t" + _fmt6(v) for v in loadings[i]) + "\n")
    return path
