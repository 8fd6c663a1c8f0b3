"""CPU-only: the C-ABI library loads and exports every symbol include/gpca.h declares (no compute)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gpca.h")).read()
    return sorted(set(re.findall(r"GPCA_API[^;(]*?\b(gpca_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(gpca):
    lib = gpca.load()
    decl = _declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"libgpca.so does not export {name}"
    from genomic_pca_amd import _lib
    assert sorted(_lib.PROTOTYPES) == decl, "ctypes prototypes and gpca.h disagree"


def test_no_undeclared_exports(gpca):
    out = subprocess.check_output(["nm", "-D", "--defined-only", gpca.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "gpca_" in l)
    assert exported == _declared()


def test_version_and_strings(gpca):
    lib = gpca.load()
    assert lib.gpca_version() == 250
    assert lib.gpca_status_string(0) == b"ok"
    assert b"missing genotype" in lib.gpca_status_string(-5)


def test_enum_values_are_pinned_and_zero_is_the_fast_default():
    """The boundary's zero value is the library's choice -- a Rust `GpcaConfig::default()` or a C `{0}` must not land on the slowest
    path.  The numbers are ABI: pinned here against the header text and the Python mirror."""
    import re
    from genomic_pca_amd import _lib
    src = open(os.path.join(ROOT, "include", "gpca.h")).read()
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(GPCA_(?:PREC|STORE)_[A-Z0-9_]+)\s*=\s*(\d+)", src)}
    assert vals == {"GPCA_PREC_DEFAULT": 0, "GPCA_PREC_I8_EXACT": 1, "GPCA_PREC_F32_MFMA": 2, "GPCA_STORE_AUTO": 0, "GPCA_STORE_2BIT": 1, "GPCA_STORE_INT8": 2}
    assert (_lib.PREC_DEFAULT, _lib.PREC_I8_EXACT, _lib.PREC_F32_MFMA) == (0, 1, 2) and (_lib.STORE_AUTO, _lib.STORE_2BIT, _lib.STORE_INT8) == (0, 1, 2)
    assert "#define GPCA_VERSION 250" in src


def test_header_cites_the_reference_interfaces():
    """Every entry point group of include/gpca.h names the reference interface it replaces (file:line)."""
    src = open(os.path.join(ROOT, "include", "gpca.h")).read()
    for cite in ("prepare.rs:1838-2030", "main.rs:602,648-660", "main.rs:359-366", "prepare.rs:1100-1422", "prepare.rs:1641-1745",
                 "main.rs:322,584", "prepare.rs:1770-1779"):
        assert cite in src, cite


def test_fails_loudly_without_gpu(gpca):
    from conftest import gpu_present
    if gpu_present():
        pytest.skip("GPU present")
    with pytest.raises(gpca.GpcaError) as e:
        gpca.GpcaEngine()
    assert e.value.status == -8 and "no CPU fallback" in str(e.value)


def test_host_hwe_helper_matches_oracle(gpca, oracle):
    for n in [(0, 0, 0), (10, 50, 40), (100, 0, 100), (25, 50, 25), (3, 0, 0), (1, 2, 400), (1234, 5000, 4100)]:
        assert gpca.GpcaEngine.hwe_chi_squared_p_value(*n) == oracle.hwe_p(*n)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "genomic_pca_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_shard_rows_cover(gpca):
    for M, W in [(1000000, 8), (1000, 3), (127, 4), (128, 2), (5, 8)]:
        spans = [gpca.shard_rows(M, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == M
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 2 * 128 or M < 128 * W


def test_host_eigensolver_against_lapack(gpca):
    """The l x l host step of gpca_rsvd (Householder tridiagonalisation + implicit QL in libgpca.so) against numpy.linalg.eigh
    (LAPACK) on Gram-like matrices of every size the engine can ask for, including repeated and zero eigenvalues."""
    import ctypes as C
    import numpy as np
    lib = gpca.load()
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 7, 16, 30, 31, 32, 50, 64, 70, 100, 128):
        for kind in ("gram", "spread", "rank_deficient", "repeated"):
            B = rng.standard_normal((max(n, 2) * 3, n))
            if kind == "spread":
                B = B * np.logspace(0, -6, n)
            if kind == "rank_deficient" and n > 2:
                B[:, -2:] = B[:, :2]
            A = B.T @ B
            if kind == "repeated":
                A = np.diag(np.repeat([4.0, 1.0], [n // 2, n - n // 2])) if n > 1 else np.array([[2.0]])
            w = np.empty(n); V = np.empty((n, n))
            rc = lib.gpca_host_eigh_desc(A.ctypes.data_as(C.c_void_p), n, w.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
            assert rc == 0
            ref = np.linalg.eigvalsh(A)[::-1]
            scale = max(abs(ref[0]), 1e-300)
            assert np.all(np.diff(w) <= 0) and np.max(np.abs(w - ref)) < 1e-12 * scale
            assert np.max(np.abs(V.T @ V - np.eye(n))) < 1e-12                         # orthonormal eigenvectors
            assert np.max(np.abs(A @ V - V * w)) < 1e-11 * scale                       # residual
    assert lib.gpca_host_eigh_desc(None, 3, None, None) == -1


def _build_c_client(tmp_path):
    exe = str(tmp_path / "c_abi_host")
    pkg = os.path.join(ROOT, "genomic_pca_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_host.c"), "-L" + pkg, "-lgpca", "-Wl,-rpath," + pkg, "-lm", "-o", exe])
    return exe


def test_header_is_plain_c99_and_links(tmp_path, gpca):
    """include/gpca.h compiles as strict C99 (no C++ in the boundary) and a C program links against libgpca.so; without a GPU
    it runs the host-only entry points and sees gpca_create fail loudly with GPCA_ERR_NO_DEVICE."""
    gpca.load()
    exe = _build_c_client(tmp_path)
    from conftest import gpu_present
    if gpu_present():
        pytest.skip("GPU present: covered by test_c_client_full_path")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "no CPU fallback" in out.stdout and "libgpca version 250" in out.stdout


@pytest.mark.gpu
def test_c_client_full_path(tmp_path, gpca):
    """The same C program end to end on the GPU: upload -> QC -> randomized PCA, then the same matrix out of core through a host
    panel callback written in C, bit-identical eigenvalues."""
    exe = _build_c_client(tmp_path)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout.splitlines()[-1] and "streamed :" in out.stdout
    assert "default  : gpca_config {0} = GPCA_PREC_I8_EXACT" in out.stdout


def test_rccl_constants_match_the_installed_header():
    """libgpca.so resolves RCCL with dlopen and therefore restates four facts of rccl.h (gpca_internal.h: kNcclFloat64,
    kNcclSum, the 128-byte unique id passed by value, ncclAllReduce's argument order): pin them to the installed header."""
    import re
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    txt = open(hdr).read()
    mine = open(os.path.join(ROOT, "genomic_pca_amd", "csrc", "gpca_internal.h")).read()
    k = dict(re.findall(r"(kNccl\w+) = (\d+)", mine))
    assert re.search(r"ncclFloat64\s*=\s*%s\b" % k["kNcclFloat64"], txt) and re.search(r"ncclSum\s*=\s*%s\b" % k["kNcclSum"], txt)
    assert re.search(r"#define NCCL_UNIQUE_ID_BYTES 128", txt)
    assert re.search(r"#define GPCA_UNIQUE_ID_BYTES 128", open(os.path.join(ROOT, "include", "gpca.h")).read())
    proto = re.search(r"ncclResult_t\s+ncclAllReduce\(([^;]*?)\);", txt, re.S).group(1)
    names = [a.split()[-1].lstrip("*") for a in proto.replace("\n", " ").split(",")]
    assert names == ["sendbuff", "recvbuff", "count", "datatype", "op", "comm", "stream"]
    init = re.search(r"ncclResult_t\s+ncclCommInitRank\(([^;]*?)\);", txt, re.S).group(1)
    assert [a.split()[-1].lstrip("*") for a in init.replace("\n", " ").split(",")] == ["comm", "nranks", "commId", "rank"]
    assert "ncclUniqueId commId" in init                       # by value


def test_integration_md_names_every_exported_symbol():
    """INTEGRATION.md shows the binding a maintainer of the reference would add: it has to name every entry point of include/gpca.h."""
    import re
    hdr = open(os.path.join(ROOT, "include", "gpca.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = re.findall(r"GPCA_API\s+[\w\s\*]+?\b(gpca_\w+)\s*\(", hdr)
    assert len(names) > 40
    assert [n for n in names if n not in doc] == []
