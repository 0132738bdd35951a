"""ctypes binding of libvdd_hip.so (C ABI: include/vdd_hip.h).  No fallback: a missing
library is an error, never a silent CPU path."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 3

VDD_F32, VDD_F16, VDD_BF16 = 0, 1, 2
PICK_ARGMAX, CUTOFF_F32_SCALAR, TEMP_RECIPROCAL, NO_SAMPLE, TOPP_FP32_MASS = 1, 2, 4, 8, 16
ROW_OK, ROW_EMPTY = 0, 1


class VddLibraryError(RuntimeError):
    pass


class VddSampleParams(C.Structure):
    """Field-for-field mirror of `vdd_sample_params` (layout checked by tests/test_abi.py)."""
    _fields_ = [
        ("abi_version", C.c_uint32), ("flags", C.c_uint32),
        ("logit_v", C.c_void_p), ("logit_cd", C.c_void_p), ("logit_dd", C.c_void_p),
        ("stride_v", C.c_int64), ("stride_cd", C.c_int64), ("stride_dd", C.c_int64),
        ("B", C.c_int32), ("V", C.c_int32), ("dtype", C.c_int32), ("min_keep", C.c_int32),
        ("alpha", C.c_double), ("log_beta", C.c_double), ("temperature", C.c_double), ("top_p", C.c_double),
        ("top_k", C.c_int32), ("n_eos", C.c_int32),
        ("philox_seed", C.c_uint64), ("philox_offset", C.c_uint64),
        ("uniforms", C.c_void_p),
        ("eos_ids", C.c_void_p), ("pad_id", C.c_int64), ("unfinished", C.c_void_p),
        ("next_tokens", C.c_void_p), ("stride_tokens", C.c_int64),
        ("scores_out", C.c_void_p), ("stride_scores", C.c_int64),
        ("top_prob", C.c_void_p), ("top_tok", C.c_void_p), ("n_top", C.c_int32), ("_pad0", C.c_int32),
        ("row_status", C.c_void_p),
        ("workspace", C.c_void_p), ("stride_workspace", C.c_int64),
        ("philox_offset_ptr", C.c_void_p),
        ("eos_min_step", C.c_void_p), ("step", C.c_int64), ("step_ptr", C.c_void_p),
        ("force_eos", C.c_void_p), ("force_eos_id", C.c_int64), ("force_eos_value", C.c_double),
    ]


def lib_path() -> str:
    return os.environ.get("VDD_HIP_LIB", os.path.join(PKG, "lib", "libvdd_hip.so"))


_lib = None


def load_lib():
    """Loads the HIP library or raises VddLibraryError (build it with
    `python -c "import __graft_entry__ as g; g.build()"`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise VddLibraryError(f"{path} not found: the HIP extension is not built. There is no CPU fallback; "
                              f"run `python __graft_entry__.py build` (needs hipcc).")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise VddLibraryError(f"cannot load {path}: {e}") from e
    lib.vdd_abi_version.restype = C.c_int
    if lib.vdd_abi_version() != ABI_VERSION:
        raise VddLibraryError(f"{path}: ABI {lib.vdd_abi_version()} != binding {ABI_VERSION}; rebuild")
    lib.vdd_contrast_sample.argtypes = [C.POINTER(VddSampleParams), C.c_void_p]
    lib.vdd_contrast_sample.restype = C.c_int
    lib.vdd_last_error.restype = C.c_char_p
    lib.vdd_lds_row_capacity.argtypes = [C.c_int]
    lib.vdd_lds_row_capacity.restype = C.c_int
    lib.vdd_topp_exact_max.restype = C.c_int
    lib.vdd_kernel_name.argtypes = [C.c_int, C.c_int]
    lib.vdd_kernel_name.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int):
    """Maps vdd_status to the Python exception types the reference surfaces."""
    if rc == 0:
        return
    msg = load_lib().vdd_last_error().decode()
    if rc == -1:
        raise ValueError(f"vdd_hip: invalid argument: {msg}")
    raise RuntimeError(f"vdd_hip: status {rc}: {msg}")
