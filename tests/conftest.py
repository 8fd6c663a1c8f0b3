import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/oracle.py)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpca():
    import genomic_pca_amd as g
    return g


@pytest.fixture()
def engine(gpca):
    """The f32-MFMA path on int8-resident genotypes, named explicitly (GpcaEngine's own default is the exact-integer path)."""
    from genomic_pca_amd import _lib
    e = gpca.GpcaEngine(precision=_lib.PREC_F32_MFMA, storage=_lib.STORE_INT8)
    yield e
    e.close()
