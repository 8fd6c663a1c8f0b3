import sys; sys.path.insert(0,'.')
import genomic_pca_amd as g
from genomic_pca_amd import _lib
th=g.synth_thresholds(1_000_000,3,seed=1)
for store in (_lib.STORE_INT8,_lib.STORE_2BIT):
    e=g.GpcaEngine(precision=_lib.PREC_I8_EXACT,storage=store); e.synth_genotypes(1_000_000,10_000,1,th)
    e.snp_stats(g.QcConfig.none(),fetch=False); e.enable_timings(True); e.reset_timings()
    for _ in range(10): e.snp_stats(g.QcConfig.none(),fetch=False)
    t=e.timings()['snp_stats']; print('store',store,'snp_stats ms',t['total_ms']/t['launches']); e.close()
