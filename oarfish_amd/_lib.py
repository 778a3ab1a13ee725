"""ctypes loader of liboarfish_em.so (the C ABI of include/oarfish_em.h).

There is no Python or CPU fallback: if the shared library is missing the import
of the product path fails loudly, and every compute entry point of the library
returns OEM_ERR_NO_DEVICE when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboarfish_em.so")

OEM_OK = 0
OEM_ERR_ARG = 1
OEM_ERR_OOM = 2
OEM_ERR_HIP = 3
OEM_ERR_RCCL = 4
OEM_ERR_NO_DEVICE = 5
OEM_ERR_STATE = 6
OEM_UNIQUE_ID_BYTES = 128
OEM_P2P_HANDLE_BYTES = 128
OEM_OPT_BATCH_BOOTSTRAP = 1
OEM_OPT_BOOTSTRAP_FIRST_REPLICA = 2
OEM_COMM_OPT_P2P_MAX_BYTES = 1
OEM_COMM_OPT_P2P_SHAPE = 2
OEM_COMM_OPT_P2P_TIMEOUT_MS = 3
OEM_COMM_OPT_P2P_SELF_CHECK = 4
OEM_COMM_INFO_RANKS = 1
OEM_COMM_INFO_RCCL_RANKS = 2
OEM_COMM_INFO_P2P_CONNECTED = 3
OEM_INFO_WEIGHT_DICT_ENTRIES = 1
OEM_INFO_TILES = 2
OEM_INFO_REMOTE_ALIGNMENTS = 3

# every symbol include/oarfish_em.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "oem_abi_version", "oem_last_error", "oem_device_count",
    "oem_store_create", "oem_store_destroy", "oem_store_dims", "oem_store_bytes", "oem_store_info", "oem_store_set_option",
    "oem_builder_create", "oem_builder_destroy", "oem_builder_add_group", "oem_builder_dims",
    "oem_builder_discard_table", "oem_builder_export", "oem_builder_coverage_probs",
    "oem_builder_coverage_probs_binomial", "oem_coverage_probs_device", "oem_builder_coverage_probs_device",
    "oem_builder_store_create",
    "oem_m_step", "oem_em_run", "oem_aux_counts", "oem_assignment_probs",
    "oem_bootstrap_weights", "oem_bootstrap",
    "oem_em_run_cells",
    "oem_comm_unique_id", "oem_comm_create", "oem_comm_destroy", "oem_comm_p2p_export", "oem_comm_p2p_connect",
    "oem_comm_set_option", "oem_comm_info", "oem_store_attach_comm",
    "oem_time_m_step", "oem_time_em_iters", "oem_time_bootstrap_passes", "oem_time_allreduce",
    "oem_cells_last_timing",
]


class OemError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"oarfish_em error {code}: {msg}")
        self.code = code


class RunInfoC(C.Structure):
    _fields_ = [
        ("niter", C.c_uint32),
        ("n_passes", C.c_uint32),
        ("converged", C.c_uint32),
        ("reserved", C.c_uint32),
        ("rel_diff", C.c_double),
    ]


class FiltersC(C.Structure):
    _fields_ = [("five_prime_clip", C.c_uint32), ("three_prime_clip", C.c_int64),
                ("score_threshold", C.c_float), ("min_aligned_fraction", C.c_float),
                ("min_aligned_len", C.c_uint32), ("which_strand", C.c_int32),
                ("score_prob_denom", C.c_float), ("reserved", C.c_uint32)]


class AlnRecordC(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("aln_start", C.c_uint32), ("aln_end", C.c_uint32),
                ("aln_span", C.c_uint32), ("score", C.c_int64), ("seq_len", C.c_int64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class DiscardTableC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("discard_5p", "discard_3p", "discard_score", "discard_aln_frac",
                                          "discard_aln_len", "discard_ori", "discard_supp", "valid_best_aln",
                                          "no_mapping", "no_valid_aln")]


REC_UNMAPPED, REC_REVERSE, REC_SUPPLEMENTARY, REC_HAS_SCORE = 1, 2, 4, 8


class StoreOptsC(C.Structure):
    _fields_ = [("reorder_rows", C.c_uint32), ("problem_size", C.c_uint32), ("window_cap", C.c_uint32),
                ("layout_build", C.c_uint32), ("weight_coding", C.c_uint32), ("reserved", C.c_uint32 * 3)]


_lib = None       # the library the package calls into (the product, unless inside `testing()`)
_product = None
_testing = None
TESTING_LIB_PATH = os.path.join(HERE, "liboarfish_em_testing.so")


def _load(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m oarfish_amd.build` "
            "(or __graft_entry__.build()).  oarfish_amd has no CPU fallback."
        )
    try:  # make sure the process has ONE HIP runtime / RCCL: torch's, if torch is around
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI
        pass
    L = C.CDLL(path)
    vp, u64, u32, i32, f64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
    L.oem_abi_version.restype = i32
    L.oem_last_error.restype = C.c_char_p
    L.oem_device_count.argtypes = [C.POINTER(i32)]
    L.oem_store_create.argtypes = [vp, vp, vp, vp, u64, u64, u32, i32, vp, C.POINTER(vp)]
    L.oem_store_destroy.argtypes = [vp]
    L.oem_store_destroy.restype = None
    L.oem_store_dims.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    L.oem_store_bytes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.oem_store_info.argtypes = [vp, u32, C.POINTER(u64)]
    L.oem_store_set_option.argtypes = [vp, u32, u64]
    L.oem_builder_create.argtypes = [vp, vp, u32, C.POINTER(vp)]
    L.oem_builder_destroy.argtypes = [vp]
    L.oem_builder_destroy.restype = None
    L.oem_builder_add_group.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.oem_builder_dims.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.oem_builder_discard_table.argtypes = [vp, vp]
    L.oem_builder_export.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.oem_builder_coverage_probs.argtypes = [vp, u32, f64, vp]
    L.oem_builder_coverage_probs_binomial.argtypes = [vp, u32, vp]
    L.oem_coverage_probs_device.argtypes = [vp, vp, vp, vp, vp, u64, u64, u32, u32, i32, f64, i32, vp]
    L.oem_builder_coverage_probs_device.argtypes = [vp, u32, i32, f64, i32, vp]
    L.oem_builder_store_create.argtypes = [vp, vp, i32, vp, C.POINTER(vp)]
    L.oem_m_step.argtypes = [vp, vp, vp, vp]
    L.oem_em_run.argtypes = [vp, vp, u32, f64, u32, vp, C.POINTER(RunInfoC)]
    L.oem_aux_counts.argtypes = [vp, vp, vp]
    L.oem_assignment_probs.argtypes = [vp, vp, f64, vp]
    L.oem_bootstrap_weights.argtypes = [vp, u64, u32, vp]
    L.oem_bootstrap.argtypes = [vp, u32, u64, vp, vp, u32, f64, vp, vp]
    L.oem_em_run_cells.argtypes = [vp, u32, vp, vp, vp, vp, u64, u64, u32, i32, u32, f64, vp, vp]
    L.oem_comm_unique_id.argtypes = [vp]
    L.oem_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.oem_comm_destroy.argtypes = [vp]
    L.oem_comm_destroy.restype = None
    L.oem_comm_p2p_export.argtypes = [vp, u64, vp]
    L.oem_comm_p2p_connect.argtypes = [vp, vp]
    L.oem_comm_set_option.argtypes = [vp, u32, u64]
    L.oem_comm_info.argtypes = [vp, u32, C.POINTER(u64)]
    L.oem_time_allreduce.argtypes = [vp, u32, C.POINTER(C.c_float)]
    L.oem_store_attach_comm.argtypes = [vp, vp, u64, u64]
    L.oem_time_m_step.argtypes = [vp, u32, C.POINTER(C.c_float)]
    L.oem_time_em_iters.argtypes = [vp, u32, C.POINTER(C.c_float)]
    L.oem_time_bootstrap_passes.argtypes = [vp, u32, C.POINTER(C.c_float), C.POINTER(u32), C.POINTER(u64)]
    L.oem_cells_last_timing.argtypes = [C.POINTER(C.c_float), C.POINTER(u64)]
    for name in ABI_SYMBOLS:
        getattr(L, name)
    return L


def lib() -> C.CDLL:
    """The library in use: the product (liboarfish_em.so), loaded once; raises if it has not been
    built.  Inside `with testing():` it is the test-only build instead."""
    global _lib, _product
    if _lib is None:
        if _product is None:
            _product = _load(LIB_PATH)
        _lib = _product
    return _lib


def testing_lib() -> C.CDLL:
    """liboarfish_em_testing.so: the product objects plus the hooks of csrc/oem_testing.hip and the
    environment-driven knobs (-DOEM_TESTING).  Only tests/ and scripts/ load it."""
    global _testing
    if _testing is None:
        L = _load(TESTING_LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.oem_debug_layout_hash.argtypes = [vp, vp, u32]
        L.oem_debug_local_comm_create.argtypes = [i32, i32, vp]
        L.oem_test_reldiff_stress.argtypes = [u32, u32, u32, i32, vp]
        _testing = L
    return _testing


class testing:
    """Context manager: every call of the package goes to the test-only library inside the block
    (handles created inside must be closed inside: the two libraries do not share state)."""

    def __enter__(self):
        global _lib
        lib()
        self._prev = _lib
        _lib = testing_lib()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev


def check(rc: int) -> None:
    if rc != OEM_OK:
        msg = lib().oem_last_error()
        raise OemError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    check(lib().oem_device_count(C.byref(n)))
    return int(n.value)
