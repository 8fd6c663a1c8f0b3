"""Product-only fingerprint: sha256 of the eigenvalues, scores and loadings the engine returns for fixed synthetic inputs and seeds.
Two builds that print the same lines compute the same bits (used when a kernel is rewritten without a change of arithmetic:
scripts/gpu.sh <tag> py=scripts/fingerprint.py under each build).  tests/golden/product_fingerprint.txt holds the current lines;
tests/test_gpu_parity.py::test_product_bits_did_not_move compares (regenerate the file when a change of arithmetic is intended)."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
import genomic_pca_amd as gpca  # noqa: E402


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def lines():
    out = []
    for (M, N, k, over, q, store) in CASES:
        from genomic_pca_amd import _lib
        with gpca.GpcaEngine(storage=_lib.STORE_2BIT if store == "2bit" else _lib.STORE_INT8) as e:
            e.synth_genotypes(M, N, 3, gpca.synth_thresholds(M, 4, seed=5, fst=0.1))
            e.snp_stats()
            e.rsvd(k, over, q, seed=11)
            out.append(f"{M} {N} {k} {over} {q} {store} {h(e.eigenvalues())} {h(e.scores(f64=True))} {h(e.loadings())}")
    return out


CASES = [(20000, 1500, 10, 10, 2, "int8"), (20000, 1500, 40, 10, 2, "int8"), (6000, 700, 90, 10, 1, "int8"),
         (20000, 2048, 10, 10, 2, "2bit"), (3000, 40, 30, 10, 2, "int8")]

if __name__ == "__main__":
    print("\n".join(lines()))
