"""CPU oracle for the genomic_pca hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``genomic_pca_amd/`` imports it; the product fails loudly without
its HIP library.

Two layers:
  * ctypes bindings to ``oracle/gpca_oracle.c`` (the line-by-line restatement; citations there);
  * numpy: exact f64 PCA via ``eigh`` of the N x N Gram of the standardised matrix -- the
    pattern of the reference's own cross-check ``tests/pca.py:81-141`` but with the Rust
    normalisation ``(g - mu) / sigma`` (``prepare.rs:1294,1357-1364,1948-1988``).

Parity status: rSVD is "parity unpinned" (algorithm lives in the un-vendored ``efficient_pca``
crate, ``Cargo.toml:30``; the reference holds no golden vectors).  What stands in for the pin:
``rsvd()`` shares no small-dense code with the product (LAPACK QR / SVD vs CholeskyQR2 / Jacobi), and
tests/test_oracle.py checks it against ``scipy.linalg.svd`` of the dense standardised matrix and against
``sklearn.utils.extmath.randomized_svd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from fractions import Fraction

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MISSING = -127  # prepare.rs:1224
# GPCA_ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build of the same source (oracle/_asan/, `make -C oracle asan`); the process must
# run with LD_PRELOAD=$(gcc -print-file-name=libasan.so) (scripts/sanitize_cpu.sh does)
_LIBDIR = os.path.join(_HERE, "_asan") if os.environ.get("GPCA_ORACLE_SANITIZE") == "1" else _HERE


def build(force: bool = False) -> None:
    need = force or any(
        not os.path.exists(os.path.join(_LIBDIR, f)) for f in ("liboracle_f64.so", "liboracle_f32.so")
    )
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _LIBDIR != _HERE else []) + (["-B"] if force else []))


_libs: dict = {}


def lib(real: str = "f64") -> C.CDLL:
    if real not in _libs:
        build()
        L = C.CDLL(os.path.join(_LIBDIR, f"liboracle_{real}.so"))
        L.orc_hwe_p.restype = C.c_double
        L.orc_hwe_p.argtypes = [C.c_uint64] * 3
        L.orc_standardize_block.restype = C.c_int64
        L.orc_rsvd.restype = C.c_int
        L.orc_cholqr2.restype = C.c_int
        L.orc_sizeof_real.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        _libs[real] = L
    return _libs[real]


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def _real_dtype(real: str):
    return np.float64 if real == "f64" else np.float32


# ----------------------------------------------------------------------------- philox / synth
def philox(c, k) -> np.ndarray:
    out = np.zeros(4, np.uint32)
    lib().orc_philox(*[C.c_uint32(int(x)) for x in c], *[C.c_uint32(int(x)) for x in k], _p(out, C.c_uint32))
    return out


def synth_genotypes(M: int, N: int, seed: int, thresh: np.ndarray, snp_offset: int = 0, ld: int | None = None) -> np.ndarray:
    ld = ld or N
    G = np.zeros((M, ld), np.int8)
    thresh = np.ascontiguousarray(thresh, np.uint32)
    lib().orc_synth_genotypes(_p(G, C.c_int8), C.c_int64(M), C.c_int64(N), C.c_int64(ld), C.c_int64(snp_offset),
                              C.c_uint64(seed), _p(thresh, C.c_uint32), C.c_int(thresh.shape[1]))
    return G


def splitmix64_at(seed: int, i: int) -> int:
    """Output number i (0-based) of SplitMix64 seeded with `seed`."""
    f = lib().orc_splitmix64_at
    f.restype = C.c_uint64
    return int(f(C.c_uint64(seed), C.c_uint64(i)))


def synth16_genotypes(M: int, N: int, seed: int, thresh16: np.ndarray, snp_offset: int = 0, ld: int | None = None) -> np.ndarray:
    """Fast panel generator (GPCA_PANEL_SYNTH16), restated in gpca_oracle.c:orc_synth16_genotypes."""
    ld = ld or N
    G = np.zeros((M, ld), np.int8)
    thresh16 = np.ascontiguousarray(thresh16, np.uint32)
    lib().orc_synth16_genotypes(_p(G, C.c_int8), C.c_int64(M), C.c_int64(N), C.c_int64(ld), C.c_int64(snp_offset),
                                C.c_uint64(seed), _p(thresh16, C.c_uint32), C.c_int(thresh16.shape[1]))
    return G


def omega(M: int, l: int, seed: int, snp_offset: int = 0) -> np.ndarray:
    Om = np.zeros((M, l), np.float64)
    lib().orc_omega(_p(Om, C.c_double), C.c_int64(M), C.c_int(l), C.c_int64(snp_offset), C.c_uint64(seed))
    return Om


# ----------------------------------------------------------------------------- a1 / a3 / a2
def hwe_p(n0: int, n1: int, n2: int) -> float:
    """prepare.rs:1641-1745."""
    return float(lib().orc_hwe_p(int(n0), int(n1), int(n2)))


def snp_stats(G: np.ndarray, N: int | None = None, min_call_rate=0.98, min_maf=0.01, max_hwe_p=1e-6):
    """prepare.rs:1216-1375 (two-pass f64).  G: int8 [M, ld] SNP-major.  Returns dict."""
    G = np.ascontiguousarray(G, np.int8)
    M, ld = G.shape
    N = N or ld
    mu = np.zeros(M, np.float32); sg = np.zeros(M, np.float32)
    keep = np.zeros(M, np.uint8); reason = np.zeros(M, np.uint8)
    counts = np.zeros((M, 4), np.uint32); s = np.zeros(M); ss = np.zeros(M)
    lib().orc_snp_stats(_p(G, C.c_int8), C.c_int64(M), C.c_int64(N), C.c_int64(ld), C.c_double(min_call_rate),
                        C.c_double(min_maf), C.c_double(max_hwe_p), _p(mu, C.c_float), _p(sg, C.c_float),
                        _p(keep, C.c_uint8), _p(reason, C.c_uint8), _p(counts, C.c_uint32), _p(s, C.c_double),
                        _p(ss, C.c_double))
    return dict(mu=mu, sigma=sg, keep=keep, reason=reason, counts=counts, sum=s, ss=ss)


def snp_sigma_exact(row: np.ndarray) -> tuple[float, float]:
    """Exact-rational mean and sample s.d. of the valid genotypes of one SNP (python ints),
    rounded once to f64 then f32 -- the mathematically exact value prepare.rs:1294-1364 approximates."""
    v = [int(x) for x in row if int(x) != MISSING]
    n = len(v)
    s1 = sum(v); s2 = sum(x * x for x in v)
    mean = Fraction(s1, n)
    var = Fraction(n * s2 - s1 * s1, n * (n - 1)) if n >= 2 else Fraction(0)
    return float(np.float32(float(mean))), float(np.float32(np.sqrt(float(var))))


def standardize_block(G: np.ndarray, mu, sigma, snp_ids, sample_ids):
    """prepare.rs:1884-2016.  Returns (block f32 [ns, nj], err) ; err = None or (snp_pos, sample_pos)."""
    G = np.ascontiguousarray(G, np.int8)
    snp_ids = np.ascontiguousarray(snp_ids, np.int64); sample_ids = np.ascontiguousarray(sample_ids, np.int64)
    mu = np.ascontiguousarray(mu, np.float32); sigma = np.ascontiguousarray(sigma, np.float32)
    out = np.zeros((len(snp_ids), len(sample_ids)), np.float32)
    rc = lib().orc_standardize_block(_p(G, C.c_int8), C.c_int64(G.shape[1]), _p(mu, C.c_float), _p(sigma, C.c_float),
                                     _p(snp_ids, C.c_int64), C.c_int64(len(snp_ids)), _p(sample_ids, C.c_int64),
                                     C.c_int64(len(sample_ids)), _p(out, C.c_float))
    if rc:
        return out, divmod(rc - 1, len(sample_ids))
    return out, None


def scale_shift(mu, sigma, keep=None):
    """r = 1/sigma, b = -mu * r in f32 (prepare.rs:1948-1949); zeros for dropped / sigma<1e-9 SNPs (:1899)."""
    mu = np.asarray(mu, np.float32); sigma = np.asarray(sigma, np.float32)
    ok = np.abs(sigma) >= np.float32(1e-9)
    if keep is not None:
        ok &= np.asarray(keep).astype(bool)
    r = np.zeros_like(sigma); b = np.zeros_like(sigma)
    r[ok] = np.float32(1.0) / sigma[ok]
    b[ok] = -mu[ok] * r[ok]
    return r, b


# ----------------------------------------------------------------------------- a5 / a6
def prod_AQ(G, N, r, b, Q, real="f64"):
    G = np.ascontiguousarray(G, np.int8); dt = _real_dtype(real)
    Q = np.ascontiguousarray(Q, dt); M, ld = G.shape; l = Q.shape[1]
    T = np.zeros((M, l), dt)
    ct = C.c_double if real == "f64" else C.c_float
    lib(real).orc_prod_AQ(_p(G, C.c_int8), C.c_int64(M), C.c_int64(N), C.c_int64(ld), _p(np.ascontiguousarray(r, np.float32), C.c_float),
                          _p(np.ascontiguousarray(b, np.float32), C.c_float), _p(Q, ct), C.c_int(l), _p(T, ct))
    return T


def prod_AtT(G, N, r, b, T, real="f64"):
    G = np.ascontiguousarray(G, np.int8); dt = _real_dtype(real)
    T = np.ascontiguousarray(T, dt); M, ld = G.shape; l = T.shape[1]
    Y = np.zeros((N, l), np.float64)
    ct = C.c_double if real == "f64" else C.c_float
    lib(real).orc_prod_AtT(_p(G, C.c_int8), C.c_int64(M), C.c_int64(N), C.c_int64(ld), _p(np.ascontiguousarray(r, np.float32), C.c_float),
                           _p(np.ascontiguousarray(b, np.float32), C.c_float), _p(T, ct), C.c_int(l), _p(Y, C.c_double))
    return Y


def cholqr2(Y, real="f64"):
    Y = np.array(Y, np.float64, order="C", copy=True); N, l = Y.shape
    dt = _real_dtype(real); Q = np.zeros((N, l), dt)
    ct = C.c_double if real == "f64" else C.c_float
    rc = lib(real).orc_cholqr2(_p(Y, C.c_double), C.c_int64(N), C.c_int(l), _p(Q, ct))
    if rc:
        raise np.linalg.LinAlgError(f"CholQR pivot {rc - 1} not positive")
    return Q


def rsvd_port(G, N, r, b, k, oversample=10, power_iters=2, seed=1, snp_offset=0, real="f64"):
    """The recipe the HIP engine runs, restated in C (oracle/gpca_oracle.c:orc_rsvd: CholeskyQR2 + cyclic Jacobi).
    Timed as bench.py's cpu_baseline ("port", real="f32"); cross-checked against rsvd() in tests/test_oracle.py."""
    G = np.ascontiguousarray(G, np.int8); M, ld = G.shape
    r = np.ascontiguousarray(r, np.float32); b = np.ascontiguousarray(b, np.float32)
    l = k + oversample
    scores = np.zeros((N, k)); ev = np.zeros(k); load = np.zeros((M, k)); sv = np.zeros(l)
    rc = lib(real).orc_rsvd(_p(G, C.c_int8), C.c_int64(M), C.c_int64(N), C.c_int64(ld), _p(r, C.c_float), _p(b, C.c_float),
                            C.c_int(k), C.c_int(oversample), C.c_int(power_iters), C.c_uint64(seed), C.c_int64(snp_offset),
                            _p(scores, C.c_double), _p(ev, C.c_double), _p(load, C.c_double), _p(sv, C.c_double))
    if rc:
        raise np.linalg.LinAlgError(f"orc_rsvd failed rc={rc}")
    return dict(scores=scores, eigenvalues=ev, loadings=load, singular_values=sv)


def rsvd(G, N, r, b, k, oversample=10, power_iters=2, seed=1, snp_offset=0, real="f64", method="lapack"):
    """THE PARITY CHECKER: randomized PCA with the engine's sketch (same Philox Omega) and the plain-loop f64 products
    of gpca_oracle.c, but with LAPACK for every small-dense step -- Householder QR (numpy.linalg.qr -> geqrf/orgqr) for
    the tall orthonormalisation and an SVD of the projection B = A Q (numpy.linalg.svd -> gesdd) instead of the
    product's CholeskyQR2 + Jacobi.  The product shares no small-dense code with this function, so a defect in either
    side's factorisations cannot cancel in a parity test.  (method="port" = rsvd_port, the same-recipe C restatement.)

    Q from Householder QR spans the same subspace as the product's CholeskyQR2 basis; scores = Q W s, loadings = U and
    eigenvalues = s^2/(N-1) do not depend on the basis chosen inside that subspace."""
    if method == "port" or real != "f64":
        return rsvd_port(G, N, r, b, k, oversample, power_iters, seed, snp_offset, real)
    G = np.ascontiguousarray(G, np.int8); M = G.shape[0]
    r = np.ascontiguousarray(r, np.float32); b = np.ascontiguousarray(b, np.float32)
    l = k + oversample
    Y = prod_AtT(G, N, r, b, omega(M, l, seed, snp_offset))            # N x l sketch  A^T Omega
    Q, _ = np.linalg.qr(Y)
    for _ in range(power_iters):
        Q, _ = np.linalg.qr(prod_AtT(G, N, r, b, prod_AQ(G, N, r, b, Q)))
    B = prod_AQ(G, N, r, b, Q)                                         # M x l
    U, s, Wt = np.linalg.svd(B, full_matrices=False)                   # B = U diag(s) Wt
    scores = (Q @ Wt.T[:, :k]) * s[:k]
    sgn = np.sign(scores[np.abs(scores).argmax(axis=0), np.arange(k)]); sgn[sgn == 0] = 1
    return dict(scores=scores * sgn, eigenvalues=s[:k] ** 2 / (N - 1), loadings=U[:, :k] * sgn, singular_values=s)


# ----------------------------------------------------------------------------- f3: the EigenSNP stages (checker of gpca.h section f3)
def _sign_by_scores(scores):
    k = scores.shape[1]
    sgn = np.sign(scores[np.abs(scores).argmax(axis=0), np.arange(k)]); sgn[sgn == 0] = 1
    return sgn


def eigensnp_local_basis(A_rows, mask, c, oversample, power_iters, seed):
    """Local eigenSNP basis of one LD block: randomized PCA of the block's standardised rows (A_rows: D x N f64, zero rows for
    SNPs of the row range that are not in the block) learnt on the subset's columns (mask, or None), with the engine's Philox
    sketch over the rows of the range.  Returns (U [D, c] orthonormal, feats [N, c] = A^T U for ALL samples)."""
    D, N = A_rows.shape
    l = c + oversample
    m = np.ones(N, bool) if mask is None else np.asarray(mask).astype(bool)

    def masked(Y):
        Y = Y.copy(); Y[~m] = 0.0
        return Y
    Q, _ = np.linalg.qr(masked(A_rows.T @ omega(D, l, seed)))
    for _ in range(power_iters):
        Q, _ = np.linalg.qr(masked(A_rows.T @ (A_rows @ Q)))
    B = A_rows @ Q
    U, s, Wt = np.linalg.svd(B, full_matrices=False)
    sgn = _sign_by_scores((Q @ Wt.T[:, :c]) * s[:c])
    U = U[:, :c] * sgn
    return U, A_rows.T @ U


def eigensnp_global_and_refine(A, Wd, k, oversample, power_iters, seed, refine_passes=1):
    """Stages 4-5 on the dense standardised matrix A (M x N f64) with the block-diagonal condensed basis Wd (M x R, dense here):
    randomized PCA of C* = Wd^T A (sketch over the R features with the engine's Philox stream), then refinement passes
    L = orth(A S), S = A^T L, S^T S = V w V^T -> scores S V, loadings L V, eigenvalues w / (N - 1)."""
    C = Wd.T @ A
    R, N = C.shape
    l = k + oversample
    Q, _ = np.linalg.qr(C.T @ omega(R, l, seed))
    for _ in range(power_iters):
        Q, _ = np.linalg.qr(C.T @ (C @ Q))
    P = C @ Q
    w, V = np.linalg.eigh(P.T @ P)
    w = w[::-1]; V = V[:, ::-1]
    s0 = (Q @ V[:, :k]) * np.sqrt(np.maximum(w[:k], 0))
    s0 = s0 * _sign_by_scores(s0)
    scores = s0
    out = None
    for _ in range(max(1, refine_passes)):
        Q0, _ = np.linalg.qr(scores)
        Lq, _ = np.linalg.qr(A @ Q0)
        S = A.T @ Lq
        w2, V2 = np.linalg.eigh(S.T @ S)
        w2 = w2[::-1]; V2 = V2[:, ::-1]
        scores = S @ V2
        sgn = _sign_by_scores(scores)
        scores = scores * sgn
        out = dict(scores=scores, loadings=(Lq @ V2) * sgn, eigenvalues=w2 / (N - 1), initial_scores=s0)
    return out


def standardized_dense(G, N, r, b) -> np.ndarray:
    """A[i, n] = g * r_i + b_i in f64 from the f32 r, b (M x N dense; small cases only)."""
    return np.asarray(G[:, :N], np.float64) * np.asarray(r, np.float64)[:, None] + np.asarray(b, np.float64)[:, None]


def exact_pca(G, N, r, b, k):
    """Exact f64 PCA of X = A^T (samples x variants): eigh of the N x N Gram A^T A
    (pattern of tests/pca.py:81-141).  scores = V * s, loadings = A V / s, eigenvalues = s^2/(N-1)."""
    A = standardized_dense(G, N, r, b)
    gram = A.T @ A
    w, V = np.linalg.eigh(gram)
    w = w[::-1][:k]; V = V[:, ::-1][:, :k]
    s = np.sqrt(np.maximum(w, 0))
    scores = V * s
    load = (A @ V) / np.where(s > 0, s, 1)
    sgn = np.sign(scores[np.abs(scores).argmax(axis=0), np.arange(k)])
    return dict(scores=scores * sgn, eigenvalues=w / (N - 1), loadings=load * sgn, singular_values=s)


def exact_pca_centred_only(G, N, keep, k):
    """The reference's own "Exact PCA Reference" (tests/pca.py:81-141, the script its sweep analysis compares every run
    against): per-variant CENTRING only (no sigma scaling), missing -> 0 after centring, GRM = sum X X^T / kept over the variants
    that pass QC, eigh, PCs = evecs * sqrt(evals).  (pca.py reads with count_A1=False, i.e. 2 - g: centring makes that a sign
    flip of every variant, which the GRM does not see.)  `keep` = the QC decisions (its filters are the Rust ones of
    prepare.rs:1281-1363 up to the monomorphic guard, which the MAF filter subsumes)."""
    X = np.asarray(G[:, :N], np.float64)[np.asarray(keep).astype(bool)]
    miss = X == MISSING
    X = np.where(miss, np.nan, X)
    X = X - np.nanmean(X, axis=1, keepdims=True)
    X = np.nan_to_num(X)
    kept = X.shape[0]
    gram = (X.T @ X) / kept
    w, V = np.linalg.eigh(gram)
    w = w[::-1][:k]; V = V[:, ::-1][:, :k]
    pcs = V * np.sqrt(np.maximum(w, 0))
    sgn = np.sign(pcs[np.abs(pcs).argmax(axis=0), np.arange(k)])
    return dict(pcs=pcs * sgn, evals=w, kept=kept)


def sign_align(X, ref):
    """Flip columns of X to maximise agreement with ref."""
    s = np.sign(np.sum(X * ref, axis=0)); s[s == 0] = 1
    return X * s


def max_abs_dpc(X, ref):
    """BASELINE.json metric: max over PCs of max |unit-norm, sign-aligned PC difference|."""
    Xn = X / np.linalg.norm(X, axis=0); Rn = ref / np.linalg.norm(ref, axis=0)
    return float(np.max(np.abs(sign_align(Xn, Rn) - Rn)))


def num_threads(real="f64") -> int:
    return int(lib(real).orc_num_threads())


def set_num_threads(n: int, real="f64") -> None:
    lib(real).orc_set_num_threads(C.c_int(int(n)))


def usable_cpus() -> dict:
    """CPUs this process can really run on: the scheduler affinity mask, capped by the cgroup CPU quota (a container on a 256-thread
    host may own 16 CPUs' worth of time: 256 OpenMP threads on it spend their time descheduled) and by the physical core count (SMT
    siblings add nothing to an FMA-bound loop)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    cores = set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    physical = len(cores) or aff
    use = max(1, min(aff, physical, int(quota + 0.5) if quota else aff))
    return {"threads": use, "affinity": aff, "cgroup_quota_cpus": quota, "physical_cores": physical}
