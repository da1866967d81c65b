"""HIP attention core (K1/K2, through the C ABI) vs golden vectors and the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32


def _load(path):
    d = np.load(path)
    return {k: torch.from_numpy(d[k]) if d[k].ndim else d[k] for k in d.files}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_attn_*.npz"))), ids=os.path.basename)
def test_attn_fwd_golden(path):
    from cabinet_amd.functional import attn_fwd_hip
    from oracle.cab_math import attn_core_fwd

    g = _load(path)
    q, k, v = (g[n].flatten(2).cuda() for n in ("q", "k", "v"))
    scale = float(g["scale"])
    ctx, lse = attn_fwd_hip(q, k, v, scale)
    torch.cuda.synchronize()
    want = g["ctx"].flatten(2)
    assert rel_err(ctx, want) < TOL
    _, lse_ref = attn_core_fwd(q.double().cpu(), k.double().cpu(), v.double().cpu(), scale)
    assert rel_err(lse, lse_ref) < 1e-5
    # fp32 HIP should sit as close to the fp64 truth as the fp32 reference does (few ulp)
    ctx64, _ = attn_core_fwd(q.double().cpu(), k.double().cpu(), v.double().cpu(), scale)
    assert rel_err(ctx, ctx64) < 2e-5
