"""Build libcabinet_hip.so (gfx950 only) in-tree with hipcc.

    python -m cabinet_amd.build [--force] [--save-temps]

hipcc cross-compiles without a GPU.  Each .hip file is compiled to an object
(in parallel), then linked into ``cabinet_amd/libcabinet_hip.so``.  The library
links against ``libamdhip64.so.7`` by SONAME with no RPATH, so inside a PyTorch
process it binds to the HIP runtime PyTorch already loaded (one runtime, shared
streams and allocations); stand-alone it resolves through the system loader path.
"""

from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "build")
LIB_PATH = os.path.join(PKG_DIR, "libcabinet_hip.so")
ARCH = "gfx950"

SOURCES = ["capi.hip", "cab_attn_fwd.hip", "cab_attn_bf16.hip", "cab_attn_bwd.hip", "ffm.hip", "ffm_bwd_fused.hip", "ffm_bwd_adj.hip", "ffm_fwd_fused.hip", "gemm_bf16.hip", "ohem.hip", "cab_local.hip", "cab_local_tiled.hip", "cab_qkv.hip", "cab_qkv_fused.hip", "bn_act.hip", "dwconv.hip", "stem_conv.hip", "pwconv.hip", "small_gemm.hip", "conv3x3_wino.hip", "bn_cls.hip"]
HEADERS = ["common.hpp", "cab_local.hpp", "cab_qkv.hpp", "blocks.hpp", "act.hpp", "bn_finalize.hpp", os.path.join("..", "..", "include", "cabinet_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libcabinet_hip.so")


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_digest():
    paths = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return _digest(paths)


def is_fresh():
    stamp = LIB_PATH + ".stamp"
    if not (os.path.exists(LIB_PATH) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _stamp()


def _stamp():
    """What the built library is checked against: the sources AND the extra flags (a -D experiment build must not pass for the product)."""
    return (source_digest() + " " + os.environ.get("CABINET_EXTRA_HIPCC_FLAGS", "").strip()).strip()


def build(force=False, save_temps=False, verbose=True):
    if not force and is_fresh():
        if verbose:
            print(f"[cabinet_amd.build] up to date: {LIB_PATH}")
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
             "-Wall", "-Wno-unused-function"]
    if save_temps:
        flags.append("-save-temps=obj")
    flags += os.environ.get("CABINET_EXTRA_HIPCC_FLAGS", "").split()  # dev aid: -D switches of A/B experiments

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # -fno-rtlib-add-rpath: drop the RUNPATH hipcc injects, so the HIP runtime the host
    # process already loaded (PyTorch's bundled libamdhip64.so.7) is the one that binds
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH,
            "-fno-rtlib-add-rpath", "-Wl,-soname,libcabinet_hip.so", "--hip-link"]
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(LIB_PATH + ".stamp", "w") as f:
        f.write(_stamp())
    if verbose:
        print(f"[cabinet_amd.build] built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    build(force=a.force, save_temps=a.save_temps)
