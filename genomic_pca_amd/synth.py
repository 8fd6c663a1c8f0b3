"""Synthetic genotype workload of SURVEY.md 8(d) / BASELINE.md: int8 dosages in {0,1,2}, SNP-major,
P populations with drift, sample n in population n % P.  This module only builds the small per-SNP
threshold table on the host; the M x N genotypes are generated on the GPU (gpca_synth_genotypes)."""
from __future__ import annotations

import numpy as np

_BLOCK = 1 << 16


def _block_table(seed: int, blk: int, P: int, fst: float) -> np.ndarray:
    """Allele frequencies p[j, c] for global SNP rows [blk*65536, (blk+1)*65536): independent stream per block,
    so any row shard reproduces exactly the rows of the unsharded table."""
    rng = np.random.Generator(np.random.Philox(key=[int(seed) & (2**64 - 1), int(blk)]))
    p = rng.uniform(0.05, 0.5, size=_BLOCK)
    z = rng.standard_normal(size=(_BLOCK, P))
    # ancestral p_j ~ U(0.05, 0.5); drift ~ Balding-Nichols (normal approximation), F_c grows with the
    # population index so the P-1 structured eigenvalues are distinct
    F = fst * (0.5 + 1.5 * np.arange(P) / max(P - 1, 1))
    pc = p[:, None] + np.sqrt(F[None, :] * (p * (1 - p))[:, None]) * z
    return np.clip(pc, 0.02, 0.98)


def synth_thresholds(M: int, P: int = 3, seed: int = 1, fst: float = 0.05, snp_offset: int = 0) -> np.ndarray:
    """uint32 [M, P]: floor(p * 2^32) for SNP rows snp_offset .. snp_offset + M - 1."""
    out = np.empty((M, P), np.uint32)
    done = 0
    while done < M:
        g = snp_offset + done
        blk, off = divmod(g, _BLOCK)
        take = min(M - done, _BLOCK - off)
        pc = _block_table(seed, blk, P, fst)[off:off + take]
        out[done:done + take] = np.minimum(np.floor(pc * 4294967296.0), 4294967295.0).astype(np.uint32)
        done += take
    return out


def synth_thresholds16(M: int, P: int = 3, seed: int = 1, fst: float = 0.05, snp_offset: int = 0) -> np.ndarray:
    """uint32 [M, P] for the fast panel generator (GPCA_PANEL_SYNTH16): one 16-bit uniform u per genotype,
    g = (u < t1) + (u < t2) with t2 = floor(p^2 * 65536) in the low half (P(g = 2)) and
    t1 = floor((1 - (1 - p)^2) * 65536) in the high half (P(g >= 1)): Hardy-Weinberg proportions of the same
    per-population allele frequencies synth_thresholds() uses.  Sample n belongs to population (n // 16) % P."""
    out = np.empty((M, P), np.uint32)
    done = 0
    while done < M:
        g = snp_offset + done
        blk, off = divmod(g, _BLOCK)
        take = min(M - done, _BLOCK - off)
        pc = _block_table(seed, blk, P, fst)[off:off + take]
        t2 = np.minimum(np.floor(pc * pc * 65536.0), 65535.0).astype(np.uint32)
        t1 = np.minimum(np.floor((1.0 - (1.0 - pc) ** 2) * 65536.0), 65535.0).astype(np.uint32)
        out[done:done + take] = (t1 << np.uint32(16)) | t2
        done += take
    return out
