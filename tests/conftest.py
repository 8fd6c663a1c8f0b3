import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def gpu_present() -> bool:
    """Is there an AMD GPU on this machine?  Asked of the kernel driver's device node, NOT of torch.cuda: torch ships its own
    copy of the HIP runtime and loads it by absolute path, so torch.cuda.is_available() in a process that has already loaded
    libgpca.so (linked against /opt/rocm's HIP) brings up a second runtime, and whichever of the two initialises second sees
    no device ("PyTorch in the same process")."""
    return os.path.exists("/dev/kfd")


def gpu_count() -> int:
    """GPUs a HIP process can use here, asked of a throw-away child process (so that no HIP runtime comes up in this one)."""
    global _GPU_COUNT
    if _GPU_COUNT is None:
        _GPU_COUNT = 0
        if gpu_present():
            import subprocess
            try:
                # the HIP runtime itself (what libgpca.so links against), not torch: a cold `import torch` costs a fresh box a minute or two
                code = ("import ctypes\n"
                        "for n in ('libamdhip64.so.7', '/opt/rocm/lib/libamdhip64.so', 'libamdhip64.so'):\n"
                        "    try:\n"
                        "        l = ctypes.CDLL(n); break\n"
                        "    except OSError:\n"
                        "        l = None\n"
                        "c = ctypes.c_int(0)\n"
                        "print(c.value if (l is None or l.hipGetDeviceCount(ctypes.byref(c)) != 0) else c.value)\n")
                out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
                _GPU_COUNT = int(out.stdout.strip().splitlines()[-1])
            except Exception:
                _GPU_COUNT = 1
    return _GPU_COUNT


_GPU_COUNT = None


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


def engine_defaults(mp, **kw):
    """Kernel-choice defaults for every GpcaEngine created while the monkeypatch (context) `mp` is active: simple=1 (register-only
    reference kernels), compact=0, narrow=0, gq_waves=..., gtt_waves=... -> genomic_pca_amd.engine.DEFAULT_* (gpca_config.reserved)."""
    import genomic_pca_amd.engine as E
    from genomic_pca_amd import _lib
    flags = 0
    if kw.get("simple"):
        flags |= _lib.CFG_SIMPLE_KERNELS
    if kw.get("compact", 1) == 0:
        flags |= _lib.CFG_NO_COMPACT
    if kw.get("narrow", 1) == 0:
        flags |= _lib.CFG_NO_NARROW
    if kw.get("spin_sync", 1) == 0:
        flags |= _lib.CFG_NO_SPIN_SYNC
    mp.setattr(E, "DEFAULT_FLAGS", flags)
    mp.setattr(E, "DEFAULT_GQ_WAVES", int(kw.get("gq_waves", 0)))
    mp.setattr(E, "DEFAULT_GTT_WAVES", int(kw.get("gtt_waves", 0)))
