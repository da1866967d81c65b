"""The C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares.
No compute call is made (there is no GPU in the CPU test tier)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(cabinet_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib_path():
    from cabinet_amd import build

    return build.build(verbose=False)


def test_header_declares_the_hot_path():
    syms = _declared_symbols()
    for s in ("cabinet_cab_attn_fwd", "cabinet_cab_attn_bwd", "cabinet_ffm_fwd", "cabinet_ffm_bwd",
              "cabinet_cab_attn_fwd_workspace_bytes", "cabinet_ffm_bwd_workspace_bytes", "cabinet_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib_path):
    from cabinet_amd import _lib

    lib = _lib.load()
    assert lib.cabinet_abi_version() == _lib.ABI_VERSION
    for s in _declared_symbols():
        assert hasattr(lib, s), s
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_library_is_gfx950_only(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data


def test_workspace_queries_and_argument_errors_need_no_gpu(lib_path):
    from cabinet_amd import _lib

    lib = _lib.load()
    assert lib.cabinet_cab_attn_fwd_workspace_bytes(8, 128, 128, 1024, 0) == 0        # 256 workgroups: no split
    assert lib.cabinet_cab_attn_fwd_workspace_bytes(2, 128, 128, 2048, 0) > 0         # split over keys
    # split-bf16 forms: the workspace also holds q, k, v as 2 / 3 bf16 pieces each; (256,128) has no such form
    assert lib.cabinet_cab_attn_fwd_workspace_bytes(8, 128, 128, 1024, 1) == 8 * 2 * 384 * 1024 * 2
    assert lib.cabinet_cab_attn_fwd_workspace_bytes(8, 128, 128, 1024, 2) == 8 * 3 * 384 * 1024 * 2
    assert lib.cabinet_cab_attn_fwd_workspace_bytes(1, 256, 128, 64, 2) == 0
    assert [lib.cabinet_cab_attn_precision_supported(128, 128, p) for p in (0, 1, 2, 3)] == [1, 1, 1, 0]
    assert [lib.cabinet_cab_attn_precision_supported(256, 128, p) for p in (0, 1, 2)] == [1, 0, 0]
    rc = lib.cabinet_cab_attn_fwd(None, None, None, 1.0, 1, 256, 128, 16, 2, None, None, None, 0, None)
    assert rc == -2 and b"precision" in lib.cabinet_last_error()
    assert lib.cabinet_cab_attn_bwd_workspace_bytes(8, 128, 128, 1024) >= 8 * 1024 * 4
    assert lib.cabinet_ffm_bwd_workspace_bytes(8, 128, 256, 256, 64, 128, 128) >= 8 * 256 * 128 * 128 * 4
    assert lib.cabinet_ffm_fwd_workspace_bytes(0, 128, 256, 256, 64, 8, 8) == 0
    # invalid arguments are rejected before any HIP call
    rc = lib.cabinet_cab_attn_fwd(None, None, None, 1.0, 1, 128, 128, 16, 0, None, None, None, 0, None)
    assert rc == -1 and b"null" in lib.cabinet_last_error()
    rc = lib.cabinet_cab_attn_fwd(None, None, None, 1.0, 1, 48, 128, 16, 0, None, None, None, 0, None)
    assert rc == -2 and b"instantiation" in lib.cabinet_last_error()
    # 16-byte alignment contract of the attention / OHEM entry points (128-bit loads and stores): a pointer 4 bytes into an
    # allocation is an invalid argument, not a fault -- checked before any HIP call (the addresses below are never touched)
    A, M = 0x10000, 0x10004
    rc = lib.cabinet_cab_attn_fwd(A, A, M, 1.0, 1, 128, 128, 16, 0, A, A, None, 0, None)
    assert rc == -1 and b"16-byte aligned" in lib.cabinet_last_error()
    rc = lib.cabinet_cab_attn_bwd(A, A, A, A, A, A, 1.0, 1, 128, 128, 16, A, M, A, None, 0, None)
    assert rc == -1 and b"16-byte aligned" in lib.cabinet_last_error()
    rc = lib.cabinet_ohem_up_pair_fwd(A, A, A, 1, 8, 4, 4, 32, 32, 0.7, 255, M, A, A, None)
    assert rc == -1 and b"16-byte aligned" in lib.cabinet_last_error()
    rc = lib.cabinet_ffm_fwd(*([None] * 9), 1, 100, 256, 256, 64, 8, 8, 1, 0.1, 1e-5, *([None] * 6), None, 0, None)
    assert rc == -2
    with pytest.raises(RuntimeError, match="code -2"):
        _lib.check(rc, "cabinet_ffm_fwd")


def test_device_tensors_never_fall_back(monkeypatch):
    """If the library is unusable the operators raise; they do not route to PyTorch ops."""
    import torch

    from cabinet_amd import _lib, functional

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcabinet_hip.so")
    with pytest.raises(RuntimeError, match="missing"):
        _lib.load()
    q = torch.randn(1, 128, 8)
    # host tensors use the host path and are unaffected
    assert functional.cab_attention(q, q, q, 0.1).shape == (1, 128, 8)
