"""ctypes binding of libcabinet_hip.so (the C ABI declared in include/cabinet_hip.h).

There is no fallback: if the library is missing, was built for another ABI
version, or a call returns an error code, a ``RuntimeError`` is raised.
"""

from __future__ import annotations

import ctypes
import os
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libcabinet_hip.so")
ABI_VERSION = 7

_c_float_p = ctypes.c_void_p  # device pointers travel as integers
_INT, _FLT, _SZ, _PTR = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/cabinet_hip.h one to one
SIGNATURES = {
    "cabinet_abi_version": (_INT, []),
    "cabinet_last_error": (ctypes.c_char_p, []),
    "cabinet_cab_attn_supported": (_INT, [_INT] * 2),
    "cabinet_cab_attn_precision_supported": (_INT, [_INT] * 3),
    "cabinet_cab_attn_fwd_workspace_bytes": (_SZ, [_INT] * 5),
    "cabinet_cab_attn_fwd": (_INT, [_PTR, _PTR, _PTR, _FLT, _INT, _INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _SZ, _PTR]),
    "cabinet_cab_attn_proj_supported": (_INT, [_INT] * 5),
    "cabinet_cab_attn_proj_fwd": (_INT, [_PTR] * 4 + [_FLT] + [_INT] * 5 + [_PTR] * 3 + [_PTR]),
    "cabinet_cab_attn_bwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_cab_attn_bwd": (_INT, [_PTR] * 6 + [_FLT] + [_INT] * 4 + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
    "cabinet_ffm_fwd_workspace_bytes": (_SZ, [_INT] * 7),
    "cabinet_ffm_fwd": (_INT, [_PTR] * 9 + [_INT] * 7 + [_INT, _FLT, _FLT] + [_PTR] * 6 + [_PTR, _SZ, _PTR]),
    "cabinet_ffm_bwd_workspace_bytes": (_SZ, [_INT] * 7),
    "cabinet_ffm_bwd": (_INT, [_PTR] * 13 + [_INT] * 7 + [_INT] + [_PTR] * 7 + [_PTR, _SZ, _PTR]),
    "cabinet_ffm_up_fwd_workspace_bytes": (_SZ, [_INT] * 9),
    "cabinet_ffm_up_fwd": (_INT, [_PTR] * 9 + [_INT] * 9 + [_INT, _FLT, _FLT, _INT] + [_PTR] * 6 + [_PTR, _SZ, _PTR]),
    "cabinet_ffm_up_bwd_workspace_bytes": (_SZ, [_INT] * 9),
    "cabinet_ffm_up_bwd": (_INT, [_PTR] * 13 + [_INT] * 9 + [_INT] + [_PTR] * 7 + [_PTR, _SZ, _PTR]),
    "cabinet_ohem_up_blocks": (_INT, [_INT] * 3),
    "cabinet_ohem_stats": (_INT, [_PTR, _PTR, _INT, _INT, _PTR, _PTR]),
    "cabinet_ohem_up_fwd": (_INT, [_PTR, _PTR] + [_INT] * 6 + [_FLT, _INT] + [_PTR] * 3 + [_PTR]),
    "cabinet_ohem_up_bwd_workspace_bytes": (_SZ, [_INT] * 6),
    "cabinet_ohem_up_bwd": (_INT, [_PTR] * 3 + [_INT] * 6 + [_FLT, _INT, _FLT] + [_PTR] + [_PTR, _SZ, _PTR]),
    "cabinet_ohem_up_pair_fwd": (_INT, [_PTR] * 3 + [_INT] * 6 + [_FLT, _INT] + [_PTR] * 3 + [_PTR]),
    "cabinet_ohem_up_pair_bwd_workspace_bytes": (_SZ, [_INT] * 6),
    "cabinet_ohem_up_pair_bwd": (_INT, [_PTR] * 4 + [_INT] * 6 + [_FLT, _INT, _FLT] + [_PTR] + [_PTR, _SZ, _PTR]),
    "cabinet_cab_qkv_supported": (_INT, [_INT] * 7 + [_PTR]),
    "cabinet_cab_qkv_padded_bins": (_INT, [_INT, _PTR]),
    "cabinet_cab_qkv_fwd_workspace_bytes": (_SZ, [_INT] * 7 + [_PTR]),
    "cabinet_cab_qkv_fwd": (_INT, [_PTR] * 14 + [_INT] * 7 + [_PTR] + [_INT, _FLT, _FLT] + [_PTR] * 10
                            + [_PTR, _SZ, _PTR]),
    "cabinet_cab_qkv_bwd_workspace_bytes": (_SZ, [_INT] * 7 + [_PTR]),
    "cabinet_cab_qkv_bwd": (_INT, [_PTR] * 20 + [_INT] * 7 + [_PTR] + [_INT] + [_PTR] * 9 + [_PTR, _SZ, _PTR]),
    "cabinet_conv1x1_fwd_workspace_bytes": (_SZ, [_INT] * 2),
    "cabinet_conv1x1_fwd": (_INT, [_PTR] * 2 + [_INT] * 4 + [_PTR] + [_PTR, _SZ, _PTR]),
    "cabinet_conv1x1_bias_supported": (_INT, [_INT] * 4),
    "cabinet_conv1x1_bias_fwd": (_INT, [_PTR] * 3 + [_INT] * 4 + [_PTR] + [_PTR, _SZ, _PTR]),
    "cabinet_channel_sum": (_INT, [_PTR] + [_INT] * 3 + [_PTR] + [_PTR]),
    "cabinet_conv1x1_bwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_conv1x1_bwd": (_INT, [_PTR] * 3 + [_INT] * 4 + [_PTR] * 2 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_act_workspace_bytes": (_SZ, [_INT] * 3),
    "cabinet_bn_act_fwd": (_INT, [_PTR] * 6 + [_INT] * 5 + [_FLT, _FLT] + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_act_bwd": (_INT, [_PTR] * 6 + [_INT] * 5 + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_cls_supported": (_INT, [_INT] * 3),
    "cabinet_bn_cls_table_floats": (_INT, [_INT] * 2),
    "cabinet_bn_cls_fwd_workspace_bytes": (_SZ, [_INT] * 3),
    "cabinet_bn_cls_fwd": (_INT, [_PTR] * 8 + [_INT] * 6 + [_FLT, _FLT] + [_PTR] * 2 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_cls_bwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_bn_cls_bwd": (_INT, [_PTR] * 3 + [_INT] * 6 + [_PTR] * 5 + [_PTR, _SZ, _PTR]),
    "cabinet_gate_act_fwd": (_INT, [_PTR] * 2 + [_INT] * 4 + [_PTR] + [_PTR]),
    "cabinet_gate_act_bwd_workspace_bytes": (_SZ, [_INT] * 3),
    "cabinet_gate_act_bwd": (_INT, [_PTR] * 3 + [_INT] * 4 + [_PTR] * 2 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_dwconv_fwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_bn_dwconv_fwd": (_INT, [_PTR] * 6 + [_INT] * 8 + [_FLT, _FLT] + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_dwconv_bwd_workspace_bytes": (_SZ, [_INT] * 6),
    "cabinet_bn_dwconv_bwd": (_INT, [_PTR] * 7 + [_INT] * 8 + [_PTR] * 4 + [_PTR, _SZ, _PTR]),
    "cabinet_stem_conv_fwd": (_INT, [_PTR] * 2 + [_INT] * 3 + [_PTR] + [_PTR]),
    "cabinet_stem_conv_wrw_workspace_bytes": (_SZ, [_INT] * 3),
    "cabinet_stem_conv_wrw": (_INT, [_PTR] * 2 + [_INT] * 3 + [_PTR] + [_PTR, _SZ, _PTR]),
    "cabinet_pwconv_supported": (_INT, [_INT] * 3),
    "cabinet_pwconv_fwd": (_INT, [_PTR] * 2 + [_INT] * 4 + [_PTR] + [_PTR]),
    "cabinet_pwconv_bwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_pwconv_bwd": (_INT, [_PTR] * 3 + [_INT] * 4 + [_PTR] * 2 + [_PTR, _SZ, _PTR]),
    "cabinet_bn_act_fwd_part": (_INT, [_PTR] * 7 + [_INT] * 6 + [_FLT, _FLT] + [_PTR] * 3 + [_PTR]),
    "cabinet_conv3x3_supported": (_INT, [_INT] * 3),
    "cabinet_conv3x3_tile_blocks": (_INT, [_INT] * 3),
    "cabinet_conv3x3_fwd_workspace_bytes": (_SZ, [_INT] * 6),
    "cabinet_conv3x3_fwd": (_INT, [_PTR] * 3 + [_INT] * 6 + [_PTR] * 2 + [_PTR, _SZ, _PTR]),
    "cabinet_conv3x3_bwd_workspace_bytes": (_SZ, [_INT] * 6),
    "cabinet_conv3x3_bwd": (_INT, [_PTR] * 4 + [_INT] * 6 + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
    "cabinet_dwconv_supported": (_INT, [_INT] * 2),
    "cabinet_dwconv_fwd": (_INT, [_PTR] * 2 + [_INT] * 6 + [_PTR] + [_PTR]),
    "cabinet_dwconv_bwd_workspace_bytes": (_SZ, [_INT] * 6),
    "cabinet_dwconv_bwd": (_INT, [_PTR] * 3 + [_INT] * 6 + [_PTR] * 2 + [_PTR, _SZ, _PTR]),
    "cabinet_cab_local_supported": (_INT, [_INT] * 4),
    "cabinet_cab_local_fwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_cab_local_bwd_workspace_bytes": (_SZ, [_INT] * 4),
    "cabinet_cab_local_fwd": (_INT, [_PTR] * 3 + [_PTR] * 5 + [_INT] * 5 + [_FLT, _FLT] + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
    "cabinet_cab_local_bwd": (_INT, [_PTR] * 4 + [_PTR] * 3 + [_PTR] * 2 + [_INT] * 5 + [_PTR] * 3 + [_PTR] * 3 + [_PTR, _SZ, _PTR]),
}

_lock = threading.Lock()
_lib = None


def _preload_hip_runtime():
    """Make sure ONE libamdhip64.so.7 is in the process before ours resolves it.

    Inside a PyTorch process that is PyTorch's bundled runtime (so streams, events and
    allocations are shared with torch); otherwise the system ROCm one.
    """
    import sys

    cands = []
    if "torch" in sys.modules:
        cands.append(os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "libamdhip64.so"))
    cands += ["libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so.7"]
    for c in cands:
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            return c
        except OSError:
            continue
    raise RuntimeError("cabinet_amd: no HIP runtime (libamdhip64.so.7) could be loaded")


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError when unusable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"cabinet_amd: {LIB_PATH} is missing -- build it with `python -m cabinet_amd.build` "
                "(there is no non-HIP fallback for device tensors)")
        _preload_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RuntimeError(f"cabinet_amd: {LIB_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        got = lib.cabinet_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"cabinet_amd: ABI version {got} != expected {ABI_VERSION}; rebuild the library")
        _lib = lib
    return _lib


def available() -> bool:
    return os.path.exists(LIB_PATH)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().cabinet_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"cabinet_amd: {what} failed (code {rc}): {msg}")
