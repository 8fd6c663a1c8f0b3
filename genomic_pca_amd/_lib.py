"""ctypes binding of libgpca.so (include/gpca.h).  No fallback: if the HIP library is missing or
fails to load, importing callers get a loud GpcaLibraryError -- there is no CPU path in the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libgpca.so")
CSRC = os.path.join(_PKG, "csrc")

GPCA_OK = 0
GPCA_ERR_BAD_ARG = -1
GPCA_ERR_OOM = -2
GPCA_ERR_HIP = -3
GPCA_ERR_RCCL = -4
GPCA_ERR_MISSING_GENOTYPE = -5
GPCA_ERR_NOT_CONVERGED = -6
GPCA_ERR_STATE = -7
GPCA_ERR_NO_DEVICE = -8
GPCA_ERR_INVALID_GENOTYPE = -9
GPCA_UNIQUE_ID_BYTES = 128
PREC_DEFAULT = 0       # = PREC_I8_EXACT: what a zeroed gpca_config selects
PREC_I8_EXACT = 1
PREC_F32_MFMA = 2
STORE_AUTO = 0         # 2-bit codes from 1 024 samples on, int8 below (decided when the genotypes arrive)
STORE_2BIT = 1
STORE_INT8 = 2
CFG_SIMPLE_KERNELS = 1     # gpca_config.reserved[0] flags (include/gpca.h)
CFG_NO_COMPACT = 2
CFG_NO_NARROW = 4
CFG_NO_SPIN_SYNC = 8


class GpcaLibraryError(RuntimeError):
    pass


class GpcaError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[gpca status {status}] {message}")
        self.status = status
        self.message = message


class gpca_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("precision", C.c_int32), ("storage", C.c_int32), ("digit_planes", C.c_int32),
                ("reserved", C.c_int32 * 4)]


class gpca_qc_config(C.Structure):
    _fields_ = [("min_snp_call_rate", C.c_double), ("min_snp_maf", C.c_double), ("max_snp_hwe_p_value", C.c_double)]


class gpca_kernel_timing(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
PANEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64)
PANEL_HOST_I8 = 0
PANEL_HOST_BED = 1
PANEL_SYNTH = 2
PANEL_SYNTH16 = 3
PANEL_MAPPED_I8 = 4
PANEL_MAPPED_BED = 5
SOURCE_REGISTER = 1
SOURCE_BENCH_HOLD = 2     # SYNTH16, measurement only: generated panels stay in their buffers (gpca.h)


class gpca_panel_source(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_pop", C.c_int32), ("fill", PANEL_FN), ("user", C.c_void_p),
                ("thresh", C.c_void_p), ("seed", C.c_uint64), ("snp_offset", C.c_int64), ("host_ld", C.c_int64), ("flags", C.c_int64)]


class gpca_stream_info(C.Structure):
    _fields_ = [("panel_rows", C.c_int64), ("n_panels", C.c_int32), ("ring_slots", C.c_int32), ("n_cached", C.c_int32),
                ("staging_buffers", C.c_int32), ("zero_staging", C.c_int32), ("copy_threads", C.c_int32), ("fills", C.c_int64),
                ("fill_host_ms", C.c_double), ("fill_wait_ms", C.c_double), ("register_ms", C.c_double), ("reserved", C.c_int64 * 4)]


# name -> (restype, argtypes); the "not gpu" test checks every one of these is exported
_H = C.c_void_p
PROTOTYPES = {
    "gpca_version": (C.c_int, []),
    "gpca_status_string": (C.c_char_p, [C.c_int]),
    "gpca_create": (C.c_int, [C.POINTER(gpca_config), C.POINTER(_H)]),
    "gpca_destroy": (C.c_int, [_H]),
    "gpca_last_error": (C.c_char_p, [_H]),
    "gpca_upload_genotypes_i8": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "gpca_upload_bed2bit": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int64]),
    "gpca_synth_genotypes": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_int32, C.c_int64]),
    "gpca_download_genotypes_i8": (C.c_int, [_H, C.c_void_p, C.c_int64]),
    "gpca_load_from_source": (C.c_int, [_H, C.POINTER(gpca_panel_source), C.c_int64, C.c_int64]),
    "gpca_stream_open": (C.c_int, [_H, C.POINTER(gpca_panel_source), C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "gpca_stream_set_fused": (C.c_int, [_H, C.c_int32]),
    "gpca_stream_set_cache": (C.c_int, [_H, C.c_int64, C.POINTER(C.c_int32)]),
    "gpca_stream_get_info": (C.c_int, [_H, C.POINTER(gpca_stream_info)]),
    "gpca_get_device_memory": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gpca_copy_rows": (C.c_int, [_H, _H, C.c_int64, C.c_int64]),
    "gpca_set_sample_mask": (C.c_int, [_H, C.c_void_p]),
    "gpca_set_condensed_basis": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64]),
    "gpca_rsvd_condensed": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]),
    "gpca_refine": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "gpca_dims": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gpca_get_storage": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gpca_snp_stats": (C.c_int, [_H, C.POINTER(gpca_qc_config), C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpca_get_snp_qc_detail": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "gpca_set_standardization": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpca_get_standardization": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpca_hwe_chi_squared_p_value": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "gpca_host_eigh_desc": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gpca_device_eigh_desc": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gpca_standardize_block": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "gpca_num_pca_snps": (C.c_int64, [_H]),
    "gpca_num_qc_samples": (C.c_int64, [_H]),
    "gpca_get_pca_snp_rows": (C.c_int, [_H, C.c_void_p]),
    "gpca_rsvd": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]),
    "gpca_get_scores": (C.c_int, [_H, C.c_void_p]),
    "gpca_get_scores_f64": (C.c_int, [_H, C.c_void_p]),
    "gpca_get_eigenvalues": (C.c_int, [_H, C.c_void_p]),
    "gpca_get_singular_values": (C.c_int, [_H, C.c_void_p]),
    "gpca_get_loadings": (C.c_int, [_H, C.c_void_p]),
    "gpca_transform": (C.c_int, [_H, C.c_void_p]),
    "gpca_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "gpca_comm_init": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "gpca_set_allreduce_hook": (C.c_int, [_H, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32, C.c_int64]),
    "gpca_comm_count_ranks": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "gpca_get_timings": (C.c_int, [_H, C.POINTER(gpca_kernel_timing), C.c_int32, C.POINTER(C.c_int32)]),
    "gpca_reset_timings": (C.c_int, [_H]),
    "gpca_enable_timings": (C.c_int, [_H, C.c_int32]),
    "gpca_synchronize": (C.c_int, [_H]),
}


def build(force: bool = False) -> str:
    """Compile libgpca.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-s"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpcaLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  genomic_pca_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise GpcaLibraryError(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GpcaLibraryError(f"{LIB_PATH} does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
