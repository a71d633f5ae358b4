"""ctypes binding of libcasync_hip.so (include/casync_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a
RuntimeError is raised with the library's own message."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

_lib: Optional[C.CDLL] = None

c_f32p = C.c_void_p       # device pointers travel as integers
c_i64 = C.c_int64


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kernel", C.c_char * 64), ("ms", C.c_float), ("ms_raw", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


_PROTOS = {
    "casync_abi_version": (C.c_int, []),
    "casync_last_error": (C.c_char_p, []),
    "casync_packed_count": (C.c_int, []),
    "casync_packed_name": (C.c_char_p, [C.c_int]),
    "casync_packed_offset": (c_i64, [C.c_int]),
    "casync_packed_size": (c_i64, [C.c_int]),
    "casync_packed_total": (c_i64, []),
    "casync_workspace_bytes": (c_i64, [C.c_int]),
    "casync_workspace_bytes_dt": (c_i64, [C.c_int, C.c_int]),
    "casync_workspace_bytes_h": (c_i64, [C.c_void_p, C.c_int]),
    "casync_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "casync_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "casync_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "casync_create_ex": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "casync_destroy": (None, [C.c_void_p]),
    "casync_load_weights_host": (C.c_int, [C.c_void_p, C.c_void_p, c_i64]),
    "casync_load_weights_device": (C.c_int, [C.c_void_p, c_f32p, c_i64]),
    "casync_forward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p, c_i64,
                                 C.c_void_p]),
    "casync_forward_windows": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_void_p, c_f32p, C.c_int,
                                         C.c_void_p, c_i64, C.c_void_p]),
    "casync_tap": (c_i64, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, c_f32p, c_i64, C.c_void_p]),
    "casync_profile_forward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p,
                                         c_i64, C.c_void_p, C.POINTER(KernelTime), C.c_int]),
    "casync_op_set_dtype": (C.c_int, [C.c_int]),
    "casync_debug_gemm_stamps": (C.c_int, [C.c_void_p]),
    "casync_debug_ir_stamps": (C.c_int, [C.c_void_p]),
    "casync_op_pw_gemm": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p,
                                    C.c_int, c_f32p, c_f32p, C.c_void_p]),
    "casync_op_dw3x3": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p]),
    "casync_op_dw3x3_ups": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "casync_op_pw_dw": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, c_f32p, C.c_int, C.c_void_p]),
    "casync_op_pw_gemm_ups": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        c_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "casync_op_ir_fused": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                     c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "casync_op_ir_fused_up": (C.c_int, [c_f32p, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p,
                                        c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "casync_op_ir_fused_upg": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                         c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]),
    "casync_op_upsample2x": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "casync_op_cross_attention": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int,
                                            c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int,
                                            C.c_void_p]),
    "casync_op_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "casync_op_audio_windows": (C.c_int, [c_f32p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "casync_op_crop_to_input": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.c_void_p]),
    "casync_op_pred_to_u8": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_void_p]),
    "casync_frame_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, c_f32p, C.c_void_p]),
    "casync_frame_paste_back": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_f32p, C.c_int,
                                          C.c_int, C.c_int, C.c_int, c_i64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "casync_op_nchw_to_nhwc": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "casync_op_inc": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_void_p]),
    "casync_op_outc": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p]),
}

EXPORTS = tuple(_PROTOS)
ABI_VERSION = 6          # == CASYNC_ABI_VERSION of include/casync_hip.h this file was written against


def lib_path() -> str:
    """The in-tree library; CASYNC_LIB names another build of the same ABI (A/B runs of two kernel versions in one
    GPU session: tools/experiments).  An override is never rebuilt."""
    return os.environ.get("CASYNC_LIB") or _build.LIB_PATH


def load() -> C.CDLL:
    """Load the engine library or raise -- there is no CPU / eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64: it must be in the process before this library resolves its
    # HIP dependency, or two HIP runtimes get loaded and the second one sees no device
    # ("create: no HIP device visible" when the engine was loaded before `import torch`).
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"casync HIP engine not built: {path} is missing. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    if path == _build.LIB_PATH and _build.is_stale():
        # a library older than its sources can carry an old struct stride / prototype: rebuild where a
        # compiler exists (the build container, the GPU box), refuse it otherwise
        try:
            _build.build()
        except Exception as exc:
            raise RuntimeError(f"{path} is older than calipsync_amd/csrc and could not be rebuilt: {exc}") from exc
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.casync_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def set_option(name: str, value: int, handle=None) -> None:
    """Set an engine switch by name (include/casync_hip.h, casync_set_option): on `handle`, or on the
    process defaults (single-operator calls and engines created later) when handle is None."""
    check(load().casync_set_option(handle, name.encode(), int(value)), f"casync_set_option({name})")


def get_option(name: str, handle=None) -> int:
    v = C.c_int()
    check(load().casync_get_option(handle, name.encode(), C.byref(v)), f"casync_get_option({name})")
    return v.value


def check(status: int, what: str) -> int:
    if status < 0:
        msg = load().casync_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")
    return status


def packed_layout():
    """[(name, offset, size)] in floats, and the total -- the engine owns the layout."""
    lib = load()
    n = lib.casync_packed_count()
    items = [(lib.casync_packed_name(i).decode(), lib.casync_packed_offset(i),
              lib.casync_packed_size(i)) for i in range(n)]
    return items, lib.casync_packed_total()
