"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

The reference is Rust; this image has no Rust toolchain, so the host side above
``include/gpca.h`` is written here with the reference's names and argument meaning:

===============================  =====================================================
reference (file:line)            here
===============================  =====================================================
PCA::new/rfit/transform          :class:`PCA`                       main.rs:602,648-660
PcaReadyGenotypeAccessor trait   :class:`MicroarrayGenotypeAccessor` prepare.rs:1838-2030
PcaSnpId / QcSampleId            plain ints (dense 0-based)          prepare.rs:1485,1854,1858
LdBlockSpecification             :class:`LdBlockSpecification`       prepare.rs:1540-1543
EigenSNPCoreAlgorithmConfig      :class:`EigenSNPCoreAlgorithmConfig` main.rs:311-327
EigenSNPCoreAlgorithm            :class:`EigenSNPCoreAlgorithm`      main.rs:359-366
===============================  =====================================================

Everything numeric happens in libgpca.so on the GPU; numpy is only the host buffer type.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import GpcaError


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class QcConfig:
    """MicroarrayDataPreparerConfig thresholds; defaults = clap's effective ones (main.rs:545-560)."""
    min_snp_call_rate: float = 0.98
    min_snp_maf: float = 0.01
    max_snp_hwe_p_value: float = 1e-6

    @staticmethod
    def none() -> "QcConfig":
        return QcConfig(0.0, 0.0, 1.0)


class PanelSource:
    """``gpca_panel_source``: where the SNP rows of a matrix come from, panel by panel.

    * ``PanelSource.host_i8(fn)``  -- ``fn(row0, rows) -> int8 [rows, N]`` (0/1/2, -127 missing)
    * ``PanelSource.host_bed(fn)`` -- ``fn(row0, rows) -> uint8 [rows, ceil(N/4)]`` PLINK .bed rows
    * ``PanelSource.synth(thresh, seed, snp_offset)`` / ``.synth16(...)`` -- device generators

    Used for resident loading (``GpcaEngine.load_from_source``) and for out-of-core streaming
    (``GpcaEngine.stream_open``; BASELINE.json configs[4]).  The reference's analogue is the accessor's strip pull loop
    (main.rs:322,584; prepare.rs:1839-2022)."""

    def __init__(self, kind: int, fill=None, thresh: Optional[np.ndarray] = None, seed: int = 0, snp_offset: int = 0,
                 mapped: Optional[np.ndarray] = None, flags: int = 0):
        self.kind = kind
        self.mapped, self.flags = mapped, int(flags)
        self.thresh = None if thresh is None else np.ascontiguousarray(thresh, np.uint32)
        self.seed, self.snp_offset = int(seed), int(snp_offset)
        self._fill = fill
        self._cb = _lib.PANEL_FN(self._tramp) if fill is not None else _lib.PANEL_FN()
        self.error: Optional[BaseException] = None

    def _tramp(self, _user, row0, rows, dst, ld):
        try:
            a = np.asarray(self._fill(int(row0), int(rows)))
            want = np.int8 if self.kind == _lib.PANEL_HOST_I8 else np.uint8
            if a.dtype != want or a.shape != (rows, ld):
                raise ValueError(f"panel callback must return {np.dtype(want).name} [{rows}, {ld}], got {a.dtype} {a.shape}")
            view = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_uint8)), shape=(rows, ld))
            view[...] = a.view(np.uint8)
            return 0
        except BaseException as e:  # noqa: BLE001 -- reported through the C status
            self.error = e
            return 1

    def c_struct(self) -> "_lib.gpca_panel_source":
        st = _lib.gpca_panel_source()
        st.kind = self.kind
        st.fill = self._cb
        st.seed, st.snp_offset = self.seed, self.snp_offset
        if self.thresh is not None:
            st.n_pop = self.thresh.shape[1]
            st.thresh = self.thresh.ctypes.data_as(C.c_void_p)
        if self.mapped is not None:
            st.user = self.mapped.ctypes.data_as(C.c_void_p)
            st.host_ld = self.mapped.strides[0]
        st.flags = self.flags
        return st

    @staticmethod
    def host_i8(fn):
        return PanelSource(_lib.PANEL_HOST_I8, fill=fn)

    @staticmethod
    def host_bed(fn):
        return PanelSource(_lib.PANEL_HOST_BED, fill=fn)

    @staticmethod
    def _mapped(kind, a, want, register):
        a = np.asarray(a)
        if a.dtype != want or a.ndim != 2 or (a.shape[1] > 1 and a.strides[1] != 1) or a.strides[0] < a.shape[1]:
            raise ValueError(f"mapped panel source: a 2-D {np.dtype(want).name} array with contiguous rows is required")
        return PanelSource(kind, mapped=a, flags=_lib.SOURCE_REGISTER if register else 0)

    @staticmethod
    def mapped_i8(a, register=False):
        """The whole int8 matrix [M, N] in host memory (an ndarray or np.memmap; rows contiguous, any row pitch): no callback, the
        library's copy threads stage the panels -- or, register=True, the pages are locked once and DMA-ed in place."""
        return PanelSource._mapped(_lib.PANEL_MAPPED_I8, a, np.int8, register)

    @staticmethod
    def mapped_bed(a, register=False):
        """The .bed payload [M, ceil(N/4)] uint8 in host memory (e.g. np.memmap of the file past its 3-byte magic)."""
        return PanelSource._mapped(_lib.PANEL_MAPPED_BED, a, np.uint8, register)

    @staticmethod
    def synth(thresh, seed, snp_offset=0):
        return PanelSource(_lib.PANEL_SYNTH, thresh=thresh, seed=seed, snp_offset=snp_offset)

    @staticmethod
    def synth16(thresh16, seed, snp_offset=0, bench_hold=False):
        """bench_hold: MEASUREMENT ONLY (gpca.h GPCA_SOURCE_BENCH_HOLD): buffers that hold a generated panel are not generated again."""
        return PanelSource(_lib.PANEL_SYNTH16, thresh=thresh16, seed=seed, snp_offset=snp_offset, flags=_lib.SOURCE_BENCH_HOLD if bench_hold else 0)


# Defaults added to every GpcaEngine's gpca_config.reserved (flags OR-ed in, wave targets used when the caller passes 0).  The library
# itself reads no environment variable for its kernel choice; a harness that wants every engine of a run on, say, the reference
# kernels sets these (the parity tests do, through monkeypatch.setattr).
DEFAULT_FLAGS = 0
DEFAULT_GQ_WAVES = 0
DEFAULT_GTT_WAVES = 0


class GpcaEngine:
    """One opaque ``gpca_handle``: one GPU, one SNP-row shard of the genotype matrix.

    Default = the exact-integer GEMM path on int8-resident genotypes (``PREC_I8_EXACT`` / ``STORE_INT8``), the path
    bench.py's headline times; ``PREC_F32_MFMA`` selects the f32 matrix-core path, ``STORE_2BIT`` packed residency."""

    def __init__(self, device: int = -1, precision: int = _lib.PREC_I8_EXACT, storage: int = _lib.STORE_INT8,
                 digit_planes: int = 0, flags: int = 0, gq_waves: int = 0, gtt_waves: int = 0):
        """flags: _lib.CFG_* bits of gpca_config.reserved[0]; gq_waves / gtt_waves: grid targets of the two GEMMs (0 = tuned defaults)."""
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self._device, self.precision, self.storage, self.digit_planes = device, precision, storage, digit_planes
        cfg = _lib.gpca_config(device=device, precision=precision, storage=storage, digit_planes=digit_planes)
        cfg.reserved[0], cfg.reserved[1], cfg.reserved[2] = flags | DEFAULT_FLAGS, gq_waves or DEFAULT_GQ_WAVES, gtt_waves or DEFAULT_GTT_WAVES
        rc = self._lib.gpca_create(C.byref(cfg), C.byref(self._h))
        if rc != _lib.GPCA_OK:
            raise GpcaError(rc, self._lib.gpca_last_error(None).decode())
        self._hook_ref = None
        self._source_ref = None

    # -- plumbing
    def _chk(self, rc: int):
        if rc != _lib.GPCA_OK:
            src = self._source_ref[0] if self._source_ref else None
            if src is not None and src.error is not None:      # a panel callback raised: show its exception
                err, src.error = src.error, None
                raise err
            raise GpcaError(rc, self._lib.gpca_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.gpca_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- residency
    def upload_genotypes_i8(self, snp_major: np.ndarray):
        """int8 [M, N] SNP-major dosages (0/1/2, -127 missing)."""
        g = np.asarray(snp_major)
        if g.dtype != np.int8 or g.ndim != 2:
            raise ValueError("snp_major must be a 2-D int8 array [SNPs, samples]")
        if g.strides[1] != 1:
            g = np.ascontiguousarray(g)
        self._chk(self._lib.gpca_upload_genotypes_i8(self._h, _vp(g), g.shape[0], g.shape[1], g.strides[0]))

    def upload_bed2bit(self, bed_rows: np.ndarray, n_samples: int):
        b = np.ascontiguousarray(bed_rows, np.uint8)
        if b.ndim != 2 or b.shape[1] != (n_samples + 3) // 4:
            raise ValueError("bed_rows must be uint8 [SNPs, ceil(N/4)]")
        self._chk(self._lib.gpca_upload_bed2bit(self._h, _vp(b), b.shape[0], n_samples))

    def synth_genotypes(self, M: int, N: int, seed: int, thresh: np.ndarray, snp_offset: int = 0):
        t = np.ascontiguousarray(thresh, np.uint32)
        if t.shape[0] != M:
            raise ValueError("thresh must be uint32 [M, P]")
        self._chk(self._lib.gpca_synth_genotypes(self._h, M, N, seed, _vp(t), t.shape[1], snp_offset))

    def load_from_source(self, src: "PanelSource", M: int, N: int):
        """Resident load through a panel source, chunked through bounded staging."""
        cs = src.c_struct()
        if src.thresh is not None and src.thresh.shape[0] != M:
            raise ValueError("thresh must be uint32 [M, P]")
        rc = self._lib.gpca_load_from_source(self._h, C.byref(cs), M, N)
        if rc != _lib.GPCA_OK and src.error is not None:
            raise src.error
        self._chk(rc)

    def stream_open(self, src: "PanelSource", M: int, N: int, panel_rows: int = 0, ring_slots: int = 3, fused: bool = True,
                    cache_bytes: int = 0):
        """Out-of-core mode: later snp_stats / rsvd / transform calls walk the matrix panel by panel through a ring of
        HBM buffers (the source object must outlive them; it is kept referenced here).  fused = True: a power iteration
        reads every panel once (4 passes per rsvd at q = 2); False: 6 passes, bit-identical to the resident engine.
        cache_bytes: HBM for the panel cache (stream_set_cache; -1 = what is free, 0 = none)."""
        cs = src.c_struct()
        if src.thresh is not None and src.thresh.shape[0] != M:
            raise ValueError("thresh must be uint32 [M, P]")
        self._source_ref = (src, cs)
        self._chk(self._lib.gpca_stream_open(self._h, C.byref(cs), M, N, panel_rows, ring_slots))
        self._chk(self._lib.gpca_stream_set_fused(self._h, int(fused)))
        if cache_bytes:
            self.stream_set_cache(cache_bytes)

    def stream_info(self) -> dict:
        """gpca_stream_get_info: shape of the open panel stream and where its host-side time went."""
        info = _lib.gpca_stream_info()
        self._chk(self._lib.gpca_stream_get_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}

    def stream_set_cache(self, max_bytes: int = -1) -> int:
        """Keep the leading panels in spare HBM (asked of the source once, read in place afterwards); returns how many."""
        n = C.c_int32()
        self._chk(self._lib.gpca_stream_set_cache(self._h, int(max_bytes), C.byref(n)))
        return n.value

    def device_memory(self):
        """(free, total) bytes of the engine's device."""
        f, t = C.c_int64(), C.c_int64()
        self._chk(self._lib.gpca_get_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def download_genotypes_i8(self) -> np.ndarray:
        M, N = self.dims()
        out = np.empty((M, N), np.int8)
        self._chk(self._lib.gpca_download_genotypes_i8(self._h, _vp(out), N))
        return out

    def storage_in_use(self):
        """(storage, precision) the handle runs with: STORE_AUTO resolves to STORE_2BIT / STORE_INT8 when the genotypes arrive."""
        st, pr = C.c_int32(), C.c_int32()
        self._chk(self._lib.gpca_get_storage(self._h, C.byref(st), C.byref(pr)))
        return st.value, pr.value

    def dims(self):
        M, N = C.c_int64(), C.c_int64()
        self._chk(self._lib.gpca_dims(self._h, C.byref(M), C.byref(N)))
        return M.value, N.value

    # -- a1 / a3
    def snp_stats(self, qc: Optional[QcConfig] = None, fetch: bool = True):
        M, _ = self.dims()
        q = qc or QcConfig.none()
        cq = _lib.gpca_qc_config(q.min_snp_call_rate, q.min_snp_maf, q.max_snp_hwe_p_value)
        if not fetch:
            self._chk(self._lib.gpca_snp_stats(self._h, C.byref(cq), None, None, None))
            return None
        mu = np.empty(M, np.float32); sg = np.empty(M, np.float32); keep = np.empty(M, np.uint8)
        self._chk(self._lib.gpca_snp_stats(self._h, C.byref(cq), _vp(mu), _vp(sg), _vp(keep)))
        return dict(mu=mu, sigma=sg, keep=keep)

    def snp_qc_detail(self):
        M, _ = self.dims()
        counts = np.empty((M, 4), np.uint32); reason = np.empty(M, np.uint8)
        self._chk(self._lib.gpca_get_snp_qc_detail(self._h, _vp(counts), _vp(reason)))
        return counts, reason

    def set_standardization(self, mu, sigma, keep=None):
        mu = np.ascontiguousarray(mu, np.float32); sigma = np.ascontiguousarray(sigma, np.float32)
        keep = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        self._chk(self._lib.gpca_set_standardization(self._h, _vp(mu), _vp(sigma), _vp(keep)))

    def get_standardization(self):
        M, _ = self.dims()
        mu = np.empty(M, np.float32); sg = np.empty(M, np.float32); keep = np.empty(M, np.uint8)
        self._chk(self._lib.gpca_get_standardization(self._h, _vp(mu), _vp(sg), _vp(keep)))
        return dict(mu=mu, sigma=sg, keep=keep)

    @staticmethod
    def hwe_chi_squared_p_value(n_hom1: int, n_het: int, n_hom2: int) -> float:
        """prepare.rs:1641-1745 (host helper in libgpca.so)."""
        return float(_lib.load().gpca_hwe_chi_squared_p_value(int(n_hom1), int(n_het), int(n_hom2)))

    # -- a2
    def standardize_block(self, pca_snp_ids: Sequence[int], qc_sample_ids: Sequence[int]) -> np.ndarray:
        s = np.ascontiguousarray(pca_snp_ids, np.int64); c = np.ascontiguousarray(qc_sample_ids, np.int64)
        out = np.zeros((len(s), len(c)), np.float32)
        self._chk(self._lib.gpca_standardize_block(self._h, _vp(s), len(s), _vp(c), len(c), _vp(out)))
        return out

    def num_pca_snps(self) -> int:
        return int(self._lib.gpca_num_pca_snps(self._h))

    def num_qc_samples(self) -> int:
        return int(self._lib.gpca_num_qc_samples(self._h))

    def pca_snp_rows(self) -> np.ndarray:
        rows = np.empty(self.num_pca_snps(), np.int64)
        self._chk(self._lib.gpca_get_pca_snp_rows(self._h, _vp(rows)))
        return rows

    # -- a5 / a6
    def rsvd(self, k: int, oversample: int = 10, power_iters: int = 2, seed: int = 1):
        self._chk(self._lib.gpca_rsvd(self._h, k, oversample, power_iters, seed))
        self._k, self._l = k, k + oversample

    def scores(self, f64: bool = False) -> np.ndarray:
        _, N = self.dims()
        out = np.empty((N, self._k), np.float64 if f64 else np.float32)
        fn = self._lib.gpca_get_scores_f64 if f64 else self._lib.gpca_get_scores
        self._chk(fn(self._h, _vp(out)))
        return out

    def eigenvalues(self) -> np.ndarray:
        out = np.empty(self._k, np.float64)
        self._chk(self._lib.gpca_get_eigenvalues(self._h, _vp(out)))
        return out

    def singular_values(self) -> np.ndarray:
        out = np.empty(self._l, np.float64)
        self._chk(self._lib.gpca_get_singular_values(self._h, _vp(out)))
        return out

    def loadings(self) -> np.ndarray:
        out = np.empty((self.num_pca_snps(), self._k), np.float32)
        self._chk(self._lib.gpca_get_loadings(self._h, _vp(out)))
        return out

    def transform(self) -> np.ndarray:
        _, N = self.dims()
        out = np.empty((N, self._k), np.float64)
        self._chk(self._lib.gpca_transform(self._h, _vp(out)))
        return out

    # -- f3: the stages of EigenSNPCoreAlgorithm (gpca.h)
    def copy_rows_from(self, src: "GpcaEngine", row0: int, rows: int):
        """This engine receives rows [row0, row0 + rows) of src's resident matrix (device to device)."""
        self._chk(self._lib.gpca_copy_rows(self._h, src._h, row0, rows))

    def set_sample_mask(self, mask: Optional[np.ndarray]):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        if m is not None and m.shape != (self.dims()[1],):
            raise ValueError("mask must have one entry per sample")
        self._chk(self._lib.gpca_set_sample_mask(self._h, _vp(m)))

    def set_condensed_basis(self, W: np.ndarray, feat0: np.ndarray, R: int):
        W = np.ascontiguousarray(W, np.float32); feat0 = np.ascontiguousarray(feat0, np.int32)
        M, _ = self.dims()
        if W.ndim != 2 or W.shape[0] != M or feat0.shape != (M,):
            raise ValueError("W must be [SNPs, cmax] and feat0 [SNPs]")
        self._chk(self._lib.gpca_set_condensed_basis(self._h, _vp(W), _vp(feat0), W.shape[1], R))

    def rsvd_condensed(self, k: int, oversample: int = 10, power_iters: int = 2, seed: int = 1):
        self._chk(self._lib.gpca_rsvd_condensed(self._h, k, oversample, power_iters, seed))
        self._k, self._l = k, k + oversample

    def refine(self, scores: np.ndarray):
        s = np.ascontiguousarray(scores, np.float64)
        if s.ndim != 2 or s.shape[0] != self.dims()[1]:
            raise ValueError("scores must be [samples, k]")
        self._chk(self._lib.gpca_refine(self._h, _vp(s), s.shape[1]))
        self._k = self._l = s.shape[1]

    # -- e
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.GPCA_UNIQUE_ID_BYTES)
        rc = _lib.load().gpca_comm_get_unique_id(buf)
        if rc != _lib.GPCA_OK:
            raise GpcaError(rc, _lib.load().gpca_last_error(None).decode())
        return buf.raw

    def comm_init(self, world: int, rank: int, unique_id: bytes, snp_offset: int):
        self._chk(self._lib.gpca_comm_init(self._h, world, rank, C.c_char_p(unique_id), snp_offset))

    def comm_count_ranks(self) -> int:
        """Ranks the exchange actually reaches (collective: a 1.0 per rank summed through RCCL or the hook)."""
        n = C.c_int32()
        self._chk(self._lib.gpca_comm_count_ranks(self._h, C.byref(n)))
        return n.value

    def set_allreduce_hook(self, fn, world: int, rank: int, snp_offset: int):
        """fn(np.ndarray f64 view) sums the buffer in place across ranks (any transport)."""
        def _tramp(_user, ptr, count):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception:  # pragma: no cover
                return 1
        self._hook_ref = _lib.ALLREDUCE_FN(_tramp)
        self._chk(self._lib.gpca_set_allreduce_hook(self._h, self._hook_ref, None, world, rank, snp_offset))

    # -- d
    def reset_timings(self):
        self._chk(self._lib.gpca_reset_timings(self._h))

    def enable_timings(self, on):
        """True / 1: HIP events around every timed kernel; n > 1: only every n-th rsvd call records them (the others run without the
        ~5 us of idle stream an event pair costs); False / 0: off."""
        self._chk(self._lib.gpca_enable_timings(self._h, int(on)))

    def timings(self) -> dict:
        n = C.c_int32()
        arr = (_lib.gpca_kernel_timing * 32)()
        self._chk(self._lib.gpca_get_timings(self._h, arr, 32, C.byref(n)))   # (timings are off until enable_timings(True))
        return {arr[i].name.decode(): dict(launches=arr[i].launches, total_ms=arr[i].total_ms, flops=arr[i].flops,
                                           bytes=arr[i].bytes) for i in range(min(n.value, 32))}

    def synchronize(self):
        self._chk(self._lib.gpca_synchronize(self._h))


# ------------------------------------------------------------------------------------------------
# efficient_pca::PCA as called at main.rs:602, 648-660
# ------------------------------------------------------------------------------------------------
class PCA:
    """``PCA::new()``, ``.rfit(x, k, n_oversamples, seed, tol)``, ``.transform(x)``.

    ``x`` is samples x variants (vcf.rs:329-342 orientation) with dosages 0/1/2.  The device keeps
    it as int8 SNP-major (1 B/genotype instead of build_matrix's 8 B + clone, main.rs:640).
    Standardisation: column mean and sample (n-1) standard deviation (UNVERIFIED for the
    un-vendored crate; matches the in-tree BED path prepare.rs:1294,1357-1364).  ``tol`` is
    accepted for signature parity and ignored (main.rs:638 always passes None).
    """

    def __init__(self, device: int = -1, precision: int = _lib.PREC_I8_EXACT, storage: int = _lib.STORE_INT8):
        self._eng = GpcaEngine(device=device, precision=precision, storage=storage)
        self._fitted = False

    def rfit(self, x: np.ndarray, k: int, n_oversamples: int = 10, seed: Optional[int] = None, tol=None,
             power_iters: int = 2):
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("x must be samples x variants")
        n_samples, n_features = x.shape
        if k == 0:
            raise ValueError("Number of components (-k) must be > 0.")          # main.rs:607-609
        if n_samples < 2:
            raise ValueError(f"PCA requires at least 2 samples, found {n_samples}.")  # main.rs:614-616
        if n_features == 0:
            raise ValueError("PCA requires at least 1 variant (feature), found 0.")   # main.rs:617-619
        k = min(k, n_samples, n_features)                                              # main.rs:621-628
        if x.dtype == np.int8:
            g = np.ascontiguousarray(x.T)
        else:
            # build_matrix (vcf.rs:317-345) fills x with u8 dosages; this engine keeps them as codes, so anything that is not
            # a whole number in the int8 range is refused here (values other than 0/1/2 are refused by the device, status -9)
            g = np.empty((n_features, n_samples), np.int8)
            step = max(1, (1 << 24) // max(n_features, 1))
            for s0 in range(0, n_samples, step):
                blk = np.asarray(x[s0:s0 + step])
                r = np.rint(blk)
                if not (np.all(np.abs(r) <= 127) and np.array_equal(r, blk)):
                    raise ValueError("PCA.rfit: x must hold genotype dosages 0/1/2 (found a value that is not a whole number)")
                g[:, s0:s0 + step] = r.T.astype(np.int8)
        self._eng.upload_genotypes_i8(g)
        self._eng.snp_stats(QcConfig.none(), fetch=False)
        # zero-variance rows leave the PCA even without QC thresholds: the reference's clamp (main.rs:621-628), applied to what is left
        if self._eng.num_pca_snps() == 0:
            raise ValueError("PCA requires at least 1 variant (feature), found 0.")
        k = min(k, self._eng.num_pca_snps())
        l = min(k + n_oversamples, n_samples, self._eng.num_pca_snps())
        self._eng.rsvd(k, l - k, power_iters, 0 if seed is None else int(seed))
        self._fitted = True
        self.k = k
        return self

    def transform(self, x: Optional[np.ndarray] = None) -> np.ndarray:
        """Scores of the fitted matrix (the reference passes the clone of the same x, main.rs:640,659)."""
        if not self._fitted:
            raise RuntimeError("PCA.transform before rfit")
        return self._eng.transform()

    def explained_variance(self) -> np.ndarray:
        return self._eng.eigenvalues()

    def rotation(self) -> np.ndarray:
        return self._eng.loadings()


# ------------------------------------------------------------------------------------------------
# The L2 boundary types, prepare.rs:1771-2030
# ------------------------------------------------------------------------------------------------
@dataclass
class LdBlockSpecification:
    user_defined_block_tag: str
    pca_snp_ids_in_block: list


class MicroarrayGenotypeAccessor:
    """``impl PcaReadyGenotypeAccessor for MicroarrayGenotypeAccessor`` (prepare.rs:1838-2030),
    backed by genotypes resident in HBM instead of the IoService actor pool."""

    def __init__(self, engine: GpcaEngine):
        self.engine = engine

    def get_standardized_snp_sample_block(self, pca_snp_ids_to_fetch, qc_sample_ids_to_fetch) -> np.ndarray:
        return self.engine.standardize_block(pca_snp_ids_to_fetch, qc_sample_ids_to_fetch)

    def num_pca_snps(self) -> int:
        return self.engine.num_pca_snps()

    def num_qc_samples(self) -> int:
        return self.engine.num_qc_samples()

    def original_indices_of_pca_snps(self) -> np.ndarray:
        return self.engine.pca_snp_rows()


@dataclass
class EigenSNPCoreAlgorithmConfig:
    """The 14 fields of main.rs:311-327 with clap's effective defaults (main.rs:561-588)."""
    target_num_global_pcs: int = 10
    components_per_ld_block: int = 7
    subset_factor_for_local_basis_learning: float = 0.075
    min_subset_size_for_local_basis_learning: int = 10_000
    max_subset_size_for_local_basis_learning: int = 40_000
    global_pca_sketch_oversampling: int = 10
    global_pca_num_power_iterations: int = 2
    local_rsvd_sketch_oversampling: int = 10
    local_rsvd_num_power_iterations: int = 2
    random_seed: int = 2025
    snp_processing_strip_size: int = 2000
    refine_pass_count: int = 1
    collect_diagnostics: bool = False
    diagnostic_block_list_id_to_trace: Optional[int] = None


@dataclass
class EigenSNPCoreOutput:
    final_sample_principal_component_scores: np.ndarray   # [N, K] f32   main.rs:389
    final_principal_component_eigenvalues: np.ndarray     # [K] f64      main.rs:394
    final_snp_principal_component_loadings: np.ndarray    # [D, K] f32   main.rs:407
    num_qc_samples_used: int = 0
    num_pca_snps_used: int = 0
    num_principal_components_computed: int = 0


class EigenSNPCoreAlgorithm:
    """``EigenSNPCoreAlgorithm::new(cfg).compute_pca(&accessor, &blocks)`` (main.rs:359-365).

    Two ways to the same quantity (top-K PCA of the standardised matrix restricted to the SNPs the caller's ``ld_blocks`` name --
    a PCA SNP in no block does not enter, prepare.rs:1465-1469):

    * ``local_stage=False`` (default): ONE global randomized PCA over the union of the blocks -- what the GPU does best (6 passes
      over the genotypes, every power iteration on the full matrix) and the more accurate of the two.  Acting config fields:
      ``target_num_global_pcs``, ``global_pca_sketch_oversampling``, ``global_pca_num_power_iterations``, ``random_seed``.
    * ``local_stage=True``: the multi-stage algorithm the 14 config fields (main.rs:311-327) parameterise, as published for the
      un-vendored ``efficient_pca`` crate (Cargo.toml:30, no pinned revision -- parity UNPINNED):
        1. a sample subset of ``clamp(subset_factor * N, min_subset, max_subset)`` samples (seeded);
        2. per LD block, a local randomized PCA on the subset's columns -> ``components_per_ld_block`` local eigenSNP loadings
           (``local_rsvd_sketch_oversampling`` / ``local_rsvd_num_power_iterations``);
        3. every sample projected on the local bases -> condensed features, row-standardised;
        4. an initial global randomized PCA of the condensed features (``global_pca_*``) -> sample scores;
        5. ``refine_pass_count`` refinement passes on the full matrix: loadings = orth(X scores), scores = X^T loadings, small SVD.
      Every pass over genotypes runs on the device (gpca.h section f3); the host only assembles the block-diagonal basis.
      ``snp_processing_strip_size`` stays a no-op (HBM-resident rows); ``collect_diagnostics`` returns a small dict of what ran.
    """

    def __init__(self, config: EigenSNPCoreAlgorithmConfig):
        self.config = config

    @staticmethod
    def _union_of_blocks(ld_blocks: Sequence[LdBlockSpecification], n_pca: int) -> np.ndarray:
        if len(ld_blocks) == 0:
            raise ValueError("compute_pca: ld_block_specifications is empty")
        ids = np.unique(np.concatenate([np.asarray(b.pca_snp_ids_in_block, np.int64).reshape(-1) for b in ld_blocks]))
        if ids.size == 0:
            raise ValueError("compute_pca: the LD blocks hold no PCA SNP")
        if ids[0] < 0 or ids[-1] >= n_pca:
            raise ValueError(f"compute_pca: PcaSnpId {int(ids[0] if ids[0] < 0 else ids[-1])} out of range [0, {n_pca})")
        return ids

    def compute_pca(self, accessor: MicroarrayGenotypeAccessor, ld_blocks: Sequence[LdBlockSpecification], local_stage: bool = False):
        eng = accessor.engine
        cfg = self.config
        n_pca = accessor.num_pca_snps()
        ids = self._union_of_blocks(ld_blocks, n_pca)
        restore = None
        if ids.size < n_pca or local_stage:
            # SNPs outside every block leave the PCA: keep mask = union of the blocks (same mu/sigma); the accessor's
            # own PCA-SNP numbering is restored afterwards, as the reference never mutates its accessor
            st = eng.get_standardization()
            rows = eng.pca_snp_rows()
            keep2 = np.zeros_like(st["keep"])
            keep2[rows[ids]] = 1
            if ids.size < n_pca:
                eng.set_standardization(st["mu"], st["sigma"], keep2)
                restore = st
        diag = None
        try:
            if local_stage:
                diag = self._multi_stage(eng, ld_blocks, st, rows)
            else:
                eng.rsvd(cfg.target_num_global_pcs, cfg.global_pca_sketch_oversampling, cfg.global_pca_num_power_iterations,
                         cfg.random_seed)
            sc = eng.scores()     # (the local stage may leave fewer components than target_num_global_pcs: min(K, condensed features))
            out = EigenSNPCoreOutput(sc, eng.eigenvalues(), eng.loadings(), accessor.num_qc_samples(),
                                     int(ids.size), int(sc.shape[1]))
        finally:
            if restore is not None:
                eng.set_standardization(restore["mu"], restore["sigma"], restore["keep"])
        if cfg.collect_diagnostics:
            diag = dict(diag or {}, stage="multi-stage" if local_stage else "global", num_ld_blocks=len(ld_blocks),
                        num_pca_snps_in_blocks=int(ids.size), pca_snp_ids_used=ids)
        else:
            diag = None
        return out, diag

    # ---- the multi-stage algorithm ---------------------------------------------------------------------------------
    @staticmethod
    def subset_size(cfg: EigenSNPCoreAlgorithmConfig, n_samples: int) -> int:
        want = int(round(cfg.subset_factor_for_local_basis_learning * n_samples))
        want = max(cfg.min_subset_size_for_local_basis_learning, min(cfg.max_subset_size_for_local_basis_learning, want))
        return max(2, min(n_samples, want))

    @staticmethod
    def subset_mask(cfg: EigenSNPCoreAlgorithmConfig, n_samples: int) -> Optional[np.ndarray]:
        """Seeded sample subset for the local bases (None = every sample): the first ns entries of a Fisher-Yates shuffle driven
        by SplitMix64(seed) -- defined here rather than taken from a library generator so that the C++ host (gpca.hpp) draws the
        same samples.  The crate uses ChaCha: WHICH samples are drawn cannot match the reference, only how many."""
        ns = EigenSNPCoreAlgorithm.subset_size(cfg, n_samples)
        if ns >= n_samples:
            return None
        m64 = (1 << 64) - 1
        state = cfg.random_seed & m64
        idx = list(range(n_samples))
        for i in range(ns):
            state = (state + 0x9E3779B97F4A7C15) & m64
            z = state
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
            z ^= z >> 31
            j = i + z % (n_samples - i)
            idx[i], idx[j] = idx[j], idx[i]
        mask = np.zeros(n_samples, np.uint8); mask[idx[:ns]] = 1
        return mask

    def _multi_stage(self, eng: "GpcaEngine", ld_blocks, st, pca_rows):
        cfg = self.config
        M, N = eng.dims()
        K = cfg.target_num_global_pcs
        mask = self.subset_mask(cfg, N)
        n_sub = N if mask is None else int(mask.sum())
        blocks = [(b.user_defined_block_tag, np.sort(pca_rows[np.asarray(b.pca_snp_ids_in_block, np.int64)])) for b in ld_blocks
                  if len(b.pca_snp_ids_in_block)]
        cp = max(1, cfg.components_per_ld_block)
        # stages 1-3: local bases on the subset, all samples projected, feature standard deviations
        cmax = min(cp, max(len(r) for _, r in blocks), n_sub)
        W = np.zeros((M, cmax), np.float32)
        feat0 = np.full(M, -1, np.int32)
        R = 0
        kw = dict(device=eng._device, precision=eng.precision, storage=eng.storage, digit_planes=eng.digit_planes)
        with GpcaEngine(**kw) as sub:
            big = max(blocks, key=lambda b: int(b[1][-1]) - int(b[1][0]))[1]
            sub.copy_rows_from(eng, int(big[0]), int(big[-1]) + 1 - int(big[0]))     # sized once for the widest block: every block reuses its buffers
            for bi, (_, rows) in enumerate(blocks):
                r0, r1 = int(rows[0]), int(rows[-1]) + 1
                sub.copy_rows_from(eng, r0, r1 - r0)
                keep = np.zeros(r1 - r0, np.uint8); keep[rows - r0] = 1
                sub.set_standardization(st["mu"][r0:r1], st["sigma"][r0:r1], keep)
                sub.set_sample_mask(mask)
                c = min(cp, len(rows), n_sub)
                lo = max(0, min(cfg.local_rsvd_sketch_oversampling, min(len(rows), n_sub) - c))
                sub.rsvd(c, lo, cfg.local_rsvd_num_power_iterations, cfg.random_seed + 1 + bi)
                U = sub.loadings()                                   # [block SNPs, c], orthonormal columns
                feats = sub.transform()                              # [N, c]: every sample on the local basis
                sd = feats.std(axis=0, ddof=1)
                ok = sd > 1e-12
                U = U[:, ok] / sd[ok].astype(np.float32)
                c = U.shape[1]
                if c == 0:
                    continue
                W[rows, :c] = U
                feat0[rows] = R
                R += c
        if R < 1:
            raise ValueError("compute_pca: no condensed features (every local component is constant)")
        # stage 4: initial global PCs from the condensed features (never formed on the host)
        eng.set_sample_mask(None)
        eng.set_condensed_basis(W, feat0, R)
        k0 = min(K, R)
        go = max(0, min(cfg.global_pca_sketch_oversampling, min(R, N) - k0))
        eng.rsvd_condensed(k0, go, cfg.global_pca_num_power_iterations, cfg.random_seed)
        scores = eng.scores(f64=True)
        # stage 5: refinement on the full matrix
        for _ in range(max(1, cfg.refine_pass_count)):
            eng.refine(scores)
            scores = eng.scores(f64=True)
        return dict(num_condensed_features=R, subset_size=n_sub, refine_passes=max(1, cfg.refine_pass_count))


