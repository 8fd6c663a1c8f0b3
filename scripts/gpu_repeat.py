"""Bitwise repeatability screen: the same rSVD call repeated on one engine must return identical bits every time (all
kernels are deterministic: fixed reduction trees, no float atomics).  A timing-dependent hazard (hand-counted vmcnt of
the LDS-DMA kernel, wait states around inline asm) shows up here as a run that differs.
usage: python scripts/gpu_repeat.py [repeats]"""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genomic_pca_amd as gpca
from genomic_pca_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
BIG = ((_lib.PREC_F32_MFMA, _lib.STORE_INT8, 1_000_000, 10_000, 20), (_lib.PREC_I8_EXACT, _lib.STORE_INT8, 1_000_000, 10_000, 20)) if os.environ.get("BIG") else ()
for (prec, store, M, N, k) in BIG + ((_lib.PREC_I8_EXACT, _lib.STORE_INT8, 300_000, 2048, 20), (_lib.PREC_I8_EXACT, _lib.STORE_INT8, 40_000, 10_000, 10),
                               (_lib.PREC_I8_EXACT, _lib.STORE_2BIT, 300_000, 2048, 20), (_lib.PREC_F32_MFMA, _lib.STORE_INT8, 300_000, 2048, 20),
                               (_lib.PREC_F32_MFMA, _lib.STORE_INT8, 100_000, 1024, 40)):
    th = gpca.synth_thresholds(M, 3, seed=7)
    with gpca.GpcaEngine(precision=prec, storage=store) as e:
        e.synth_genotypes(M, N, 7, th)
        e.snp_stats(gpca.QcConfig.none())
        seen = {}
        first = None
        for r in range(reps):
            e.rsvd(k, 10, 2, seed=3)
            ev, sc, ld = e.eigenvalues(), e.scores(f64=True), e.loadings()
            h = hashlib.sha1(ev.tobytes() + sc.tobytes() + ld.tobytes()).hexdigest()[:12]
            seen[h] = seen.get(h, 0) + 1
            if first is None:
                first = (ev, sc, ld)
            elif h != next(iter(seen)):
                print("   run", r, "differs: max|d ev|/ev", float(np.max(np.abs(ev - first[0]) / first[0])), "max|d sc|", float(np.max(np.abs(sc - first[1]))),
                      "max|d ld|", float(np.max(np.abs(ld - first[2]))))
        print(f"precision {prec} storage {store} {M}x{N} k={k}: {len(seen)} distinct result(s) over {reps} runs {seen}", flush=True)
        bad += len(seen) != 1
print("REPEATABLE" if bad == 0 else f"NOT REPEATABLE in {bad} configuration(s)")
sys.exit(1 if bad else 0)
