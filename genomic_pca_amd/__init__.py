"""genomic_pca_amd -- MI355X-native randomized PCA for genotype matrices (hot path of SauersML/genomic_pca).

Product code: libgpca.so (HIP, gfx950) behind include/gpca.h + this thin host mirror.  No CPU fallback."""
from ._lib import GpcaError, GpcaLibraryError, LIB_PATH, build, load  # noqa: F401
from .engine import (EigenSNPCoreAlgorithm, EigenSNPCoreAlgorithmConfig, EigenSNPCoreOutput, GpcaEngine,  # noqa: F401
                     LdBlockSpecification, MicroarrayGenotypeAccessor, PCA, PanelSource, QcConfig)
from .synth import synth_thresholds, synth_thresholds16  # noqa: F401
from .distributed import shard_rows  # noqa: F401

__version__ = "0.2.2"
