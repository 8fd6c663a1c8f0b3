"""What the per-kernel HIP events of enable_timings cost a call (the bench needs them on: roofline.achieved is measured live)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import genomic_pca_amd as g
M, N, k = 1_000_000, 10_000, 20
e = g.GpcaEngine()
e.synth_genotypes(M, N, 1, g.synth_thresholds(M, 3, seed=1)); e.snp_stats(g.QcConfig.none(), fetch=False); e.rsvd(k, 10, 2, 1)
res = {False: [], True: []}
for r in range(10):
    for on in (False, True):
        e.enable_timings(on); e.reset_timings()
        e.rsvd(k, 10, 2, 1); e.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            e.rsvd(k, 10, 2, 1)
        e.synchronize()
        res[on].append((time.perf_counter() - t0) / 5 * 1e3)
for on in (False, True):
    a = np.array(res[on]); print(f"timings {'on ' if on else 'off'}: {a.mean():.4f} ms per call (min {a.min():.4f})")
