"""bench.py's per-kernel roofline loop at a tiny size: every kernel group must launch and report finite numbers
(the default `python bench.py` runs this loop at config 3 after the timed region)."""
import importlib.util
import math
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_kernel_rooflines_run_at_a_small_size():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = bench.kernel_rooflines(2, 256, 2)
    names = [r["kernel"].split(" ")[0] for r in rows]
    for want in ("cab_attn_fwd", "cab_attn_bwd", "ffm_up_fwd", "ffm_up_bwd", "bn_act_fwd", "bn_act_bwd", "bn_dwconv_fwd",
                 "bn_dwconv_bwd", "stem_conv_fwd", "stem_conv_wrw", "pwconv_fwd", "pwconv_bwd", "ohem_up_fwd",
                 "ohem_up_bwd", "cab_local_fwd", "cab_local_bwd", "cab_qkv_fwd", "cab_qkv_bwd"):
        assert want in names, want
    for r in rows:
        assert r["bound"] in ("hbm", "mfma") and r["ms_per_launch"] > 0
        assert all(math.isfinite(r[k]) for k in ("achieved", "peak", "frac", "tflops", "gbytes_per_s"))
        assert r["traffic"] is None  # PMC traffic is only attached at the shape it was measured at (config 3)
