"""HIP attention core (K1/K2, through the C ABI) vs golden vectors and the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close, rel_err

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


SPLIT_SHAPES = [(2, 128, 128, 256), (1, 128, 128, 1000), (1, 128, 128, 31), (3, 64, 64, 130), (8, 128, 128, 1024),
                (2, 128, 128, 2048), (1, 128, 128, 8704)]


@pytest.mark.parametrize("B,Kc,Vc,n", SPLIT_SHAPES)
def test_attn_fwd_split_bf16_vs_fp64(B, Kc, Vc, n):
    """K1 on the bf16 matrix pipe (include/cabinet_hip.h CABINET_PREC_*): every fp32 operand as two (bf16x3) or three (bf16x6)
    bf16 pieces.  Against the fp64 oracle of cab.py:149-154, next to the exact-fp32-MFMA kernel on the same inputs:
    bf16x6 must be as accurate as fp32 (<= 2x its error, or 1e-6), bf16x3 within 5e-5 (measured ~1e-5); lse alike.  Shapes:
    ragged n, kv-split grids (2 x 2048), the un-tiled validation frame (n = 8704), production grids of configs 3 and 5."""
    from cabinet_amd.functional import PREC_BF16X3, PREC_BF16X6, PREC_FP32, attn_fwd_hip
    from oracle.cab_math import attn_core_fwd

    g0 = torch.Generator().manual_seed(n + Kc)
    q = torch.randn(B, Kc, n, generator=g0).relu()
    k = torch.randn(B, Kc, n, generator=g0) + 0.5     # a common-mode key component, as PSP outputs have
    v = torch.randn(B, Vc, n, generator=g0)
    scale = Kc ** -0.5
    ctx64, lse64 = attn_core_fwd(q.double(), k.double(), v.double(), scale)
    err = {}
    for prec in (PREC_FP32, PREC_BF16X6, PREC_BF16X3):
        ctx, lse = attn_fwd_hip(q.cuda(), k.cuda(), v.cuda(), scale, prec)
        torch.cuda.synchronize()
        err[prec] = (rel_err(ctx, ctx64), rel_err(lse, lse64))
    assert err[PREC_FP32][0] < 2e-6, err
    assert err[PREC_BF16X6][0] <= max(2 * err[PREC_FP32][0], 1e-6) and err[PREC_BF16X6][1] <= max(2 * err[PREC_FP32][1], 1e-6), err
    assert err[PREC_BF16X3][0] < 5e-5 and err[PREC_BF16X3][1] < 5e-5, err


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_attn_*.npz"))), ids=os.path.basename)
def test_attn_fwd_split_bf16_golden(path):
    """The split-bf16 forms against the reference-generated vectors (1e-3, the north-star bar) and bit-reproducible."""
    from cabinet_amd.functional import PREC_BF16X3, PREC_BF16X6, attn_fwd_hip

    from cabinet_amd import _lib

    g = _load(path)
    q, k, v = (g[n].flatten(2).cuda() for n in ("q", "k", "v"))
    if not _lib.load().cabinet_cab_attn_precision_supported(q.shape[1], v.shape[1], PREC_BF16X6):
        pytest.skip("no split-bf16 instantiation for this channel pair (fp32 MFMA only)")
    for prec in (PREC_BF16X6, PREC_BF16X3):
        ctx, lse = attn_fwd_hip(q, k, v, float(g["scale"]), prec)
        ctx2, lse2 = attn_fwd_hip(q, k, v, float(g["scale"]), prec)
        torch.cuda.synchronize()
        assert rel_err(ctx, g["ctx"].flatten(2)) < (2e-6 if prec == PREC_BF16X6 else 5e-5)
        assert torch.equal(ctx, ctx2) and torch.equal(lse, lse2)


def test_attn_split_bf16_autograd_and_dispatch():
    """cab_attention(..., precision=...) runs K1 in the chosen arithmetic and K2 in fp32; gradients stay within 1e-3 of the
    oracle for both split forms; channel pairs without a split instantiation refuse an explicit request and ignore the
    process-wide default."""
    from cabinet_amd import functional as Fn
    from oracle.cab_math import attn_core_bwd, attn_core_fwd

    g0 = torch.Generator().manual_seed(5)
    q, k, v, dy = (torch.randn(2, 128, 200, generator=g0) for _ in range(4))
    q = q.relu()
    ctx64, lse64 = attn_core_fwd(q.double(), k.double(), v.double(), 128 ** -0.5)
    want = attn_core_bwd(dy.double(), q.double(), k.double(), v.double(), ctx64, lse64, 128 ** -0.5)
    for prec in ("bf16x6", "bf16x3"):
        t = [x.cuda().requires_grad_(True) for x in (q, k, v)]
        out = Fn.cab_attention(t[0], t[1], t[2], 128 ** -0.5, precision=prec)
        out.backward(dy.cuda())
        assert rel_err(out, ctx64) < (2e-6 if prec == "bf16x6" else 5e-5)
        for got, ref, name in zip(t, want, ("dq", "dk", "dv")):
            assert rel_err(got.grad, ref) < (1e-5 if prec == "bf16x6" else 2e-4), (prec, name)
    lib = Fn._lib.load()
    assert lib.cabinet_cab_attn_precision_supported(128, 128, 2) == 1 and lib.cabinet_cab_attn_precision_supported(256, 128, 2) == 0
    q2 = torch.randn(1, 256, 64, device="cuda")
    with pytest.raises(RuntimeError, match="not built"):
        Fn.cab_attention(q2, q2, torch.randn(1, 128, 64, device="cuda"), 0.1, precision="bf16x6")
    old = Fn.ATTN_PRECISION
    try:
        Fn.ATTN_PRECISION = "bf16x6"   # process-wide default: (256,128) silently stays fp32, (128,128) takes the split form
        assert Fn.cab_attention(q2, q2, torch.randn(1, 128, 64, device="cuda"), 0.1).shape == (1, 128, 64)
        assert Fn._resolve_precision(None, 128, 128) == Fn.PREC_BF16X6 and Fn._resolve_precision(None, 256, 128) == Fn.PREC_FP32
    finally:
        Fn.ATTN_PRECISION = old


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_attn_*.npz"))), ids=os.path.basename)
def test_attn_bwd_golden(path):
    from cabinet_amd.functional import attn_bwd_hip, attn_fwd_hip

    g = _load(path)
    q, k, v, dctx = (g[n].flatten(2).cuda() for n in ("q", "k", "v", "g"))
    scale = float(g["scale"])
    ctx, lse = attn_fwd_hip(q, k, v, scale)
    dq, dk, dv = attn_bwd_hip(dctx, q, k, v, ctx, lse, scale)
    torch.cuda.synchronize()
    for name, got in (("dq", dq), ("dk", dk), ("dv", dv)):
        assert rel_err(got, g[name].flatten(2)) < TOL, name


@pytest.mark.parametrize("B,Kc,Vc,n", [(1, 128, 128, 1), (1, 128, 128, 31), (2, 128, 128, 33), (3, 64, 64, 130),
                                       (4, 128, 128, 256), (1, 256, 128, 2048), (8, 128, 128, 1024),
                                       (2, 128, 128, 2048), (1, 256, 128, 66), (2, 256, 128, 1000),
                                       # n % 4 == 0 (the two-waves-per-SIMD kernels): one partial tile, two tiles for eight
                                       # waves (six waves merge an empty result), a ragged last tile, a key-split batch
                                       (1, 128, 128, 4), (1, 128, 128, 36), (2, 128, 128, 1000), (5, 64, 64, 520),
                                       (1, 128, 128, 3076)])
def test_attn_autograd_vs_oracle(B, Kc, Vc, n):
    """cab_attention (autograd Function over the C ABI) vs the explicit-formula oracle, fp64 truth."""
    from cabinet_amd.functional import cab_attention
    from oracle.cab_math import attn_core_bwd, attn_core_fwd

    gen = torch.Generator().manual_seed(1000 + n)
    q = torch.randn(B, Kc, n, generator=gen).relu()
    k = torch.randn(B, Kc, n, generator=gen)
    v = torch.randn(B, Vc, n, generator=gen)
    g = torch.randn(B, Vc, n, generator=gen)
    scale = Kc ** -0.5
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    ctx = cab_attention(qd, kd, vd, scale)
    ctx.backward(g.cuda())
    torch.cuda.synchronize()
    ctx_ref, lse_ref = attn_core_fwd(q, k, v, scale)  # fp32 CPU oracle
    dq_ref, dk_ref, dv_ref = attn_core_bwd(g, q, k, v, ctx_ref, lse_ref, scale)
    assert_close(ctx, ctx_ref, TOL, "ctx")
    assert_close(qd.grad, dq_ref, TOL, "dq")
    assert_close(kd.grad, dk_ref, TOL, "dk")
    assert_close(vd.grad, dv_ref, TOL, "dv")
    if B * n * n <= 4 * 1024 * 1024:  # fp64 truth where it is cheap
        ctx64, lse64 = attn_core_fwd(q.double(), k.double(), v.double(), scale)
        d64 = attn_core_bwd(g.double(), q.double(), k.double(), v.double(), ctx64, lse64, scale)
        assert_close(ctx, ctx64, 2e-5, "ctx vs fp64")
        for nm, got, want in zip(("dq", "dk", "dv"), (qd.grad, kd.grad, vd.grad), d64):
            assert_close(got, want, 5e-5, nm + " vs fp64")


def test_attn_properties_full_size():
    """Size-independent properties at BASELINE config-3 size (B=8, n=1024)."""
    from cabinet_amd.functional import attn_fwd_hip

    B, Kc, Vc, n = 8, 128, 128, 1024
    gen = torch.Generator().manual_seed(7)
    q = torch.randn(B, Kc, n, generator=gen).relu().cuda()
    k = torch.randn(B, Kc, n, generator=gen).cuda()
    v = torch.randn(B, Vc, n, generator=gen).cuda()
    scale = Kc ** -0.5
    ctx, lse = attn_fwd_hip(q, k, v, scale)
    # rows of P sum to one: a constant value field is reproduced
    ones = torch.ones_like(v)
    c1, _ = attn_fwd_hip(q, k, ones, scale)
    assert (c1 - 1).abs().max() < 1e-5
    # linear in v
    v2 = torch.randn(B, Vc, n, generator=gen).cuda()
    ca, _ = attn_fwd_hip(q, k, v + 2 * v2, scale)
    cb, _ = attn_fwd_hip(q, k, v2, scale)
    assert rel_err(ca, ctx + 2 * cb) < 1e-5
    # permuting the keys (and values alike) leaves ctx and lse unchanged
    perm = torch.randperm(n, generator=gen).cuda()
    cp, lp = attn_fwd_hip(q, k[:, :, perm].contiguous(), v[:, :, perm].contiguous(), scale)
    assert rel_err(cp, ctx) < 1e-5 and rel_err(lp, lse) < 1e-6
    # zero keys -> uniform attention: ctx = mean_j v, lse = log n
    c0, l0 = attn_fwd_hip(q, torch.zeros_like(k), v, scale)
    assert rel_err(c0, v.mean(dim=2, keepdim=True).expand_as(v)) < 1e-5
    assert (l0 - float(np.log(n))).abs().max() < 1e-5
    # bitwise reproducible
    c2, l2 = attn_fwd_hip(q, k, v, scale)
    assert torch.equal(c2, ctx) and torch.equal(l2, lse)


def test_attn_rejects_bad_input():
    from cabinet_amd.functional import attn_fwd_hip, cab_attention

    q = torch.randn(1, 48, 64).cuda()
    with pytest.raises(RuntimeError, match="no gfx950 instantiation"):
        attn_fwd_hip(q, q, q, 1.0)
    with pytest.raises(RuntimeError, match="bad shapes"):
        cab_attention(q, q[:, :, :32], q, 1.0)


@pytest.mark.parametrize("B,Kc,Vc,n", [(2, 128, 128, 512), (1, 64, 64, 200), (1, 256, 128, 160), (2, 128, 128, 2048)])
def test_attn_bwd_dq_is_conditioned_for_common_mode_keys(B, Kc, Vc, n):
    """Keys with a large component common to all positions and near-uniform attention -- what PSP-pooled keys look like
    in the model.  sum_j dS_ij = 0, so dq = scale * sum_j dS_ij (k_j - mean k) exactly; K2a feeds the product with the
    centred keys, which removes the coherent error of the flash-style D_i.  Uncentred fp32 (oracle formulas run in fp32)
    is ~7e-5 off the fp64 dq on these inputs, the kernel must be within 5e-6 (measured 4e-7 for the centred formula);
    (2,128,128,2048) takes the key-range-split path.  dk, dv are unaffected and stay at the usual level."""
    from cabinet_amd.functional import attn_bwd_hip, attn_fwd_hip
    from oracle.cab_math import attn_core_bwd, attn_core_fwd

    g0 = torch.Generator().manual_seed(3)
    q = torch.randn(B, Kc, n, generator=g0).relu() * 0.2
    k = 3.0 * torch.randn(B, Kc, 1, generator=g0) + 0.02 * torch.randn(B, Kc, n, generator=g0)
    v = torch.randn(B, Vc, n, generator=g0)
    g = torch.randn(B, Vc, n, generator=g0)
    sc = Kc ** -0.5
    c64, l64 = attn_core_fwd(q.double(), k.double(), v.double(), sc)
    dq64, dk64, dv64 = attn_core_bwd(g.double(), q.double(), k.double(), v.double(), c64, l64, sc)
    c32, l32 = attn_core_fwd(q, k, v, sc)
    dq32 = attn_core_bwd(g, q, k, v, c32, l32, sc)[0]
    naive = float((dq32.double() - dq64).norm() / dq64.norm())
    qd, kd, vd, gd = (t.cuda() for t in (q, k, v, g))
    ctx, lse = attn_fwd_hip(qd, kd, vd, sc)
    dq, dk, dv = attn_bwd_hip(gd, qd, kd, vd, ctx, lse, sc)
    torch.cuda.synchronize()
    err = float((dq.double().cpu() - dq64).norm() / dq64.norm())
    assert err <= 5e-6, (err, naive)
    assert naive > 4 * err  # the test inputs do exercise the cancellation
    assert_close(dk, dk64, 1e-5, "dk")
    assert_close(dv, dv64, 1e-5, "dv")


def test_attn_bwd_lds_attribute_covers_a_later_larger_n():
    """The stored-dS dk/dv kernel's dynamic LDS grows with n (8 n bytes of row constants beside the k | v tiles) while
    hipFuncAttributeMaxDynamicSharedMemorySize is set once per device: a small n first must not pin the attribute to its
    size.  (Kc, Vc) = (256, 128): 48 KB of tiles, so n = 4096 needs 80 KB -- above both the 64 KB default and what n = 128
    asked for."""
    from cabinet_amd.functional import cab_attention
    from oracle.cab_math import attn_core_bwd, attn_core_fwd

    for n in (128, 4096):
        gen = torch.Generator().manual_seed(n)
        q = torch.randn(1, 256, n, generator=gen).relu()
        k = torch.randn(1, 256, n, generator=gen)
        v = torch.randn(1, 128, n, generator=gen)
        g = torch.randn(1, 128, n, generator=gen)
        qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
        ctx = cab_attention(qd, kd, vd, 256 ** -0.5)
        ctx.backward(g.cuda())
        torch.cuda.synchronize()
        ctx_ref, lse_ref = attn_core_fwd(q, k, v, 256 ** -0.5)
        dq_ref, dk_ref, dv_ref = attn_core_bwd(g, q, k, v, ctx_ref, lse_ref, 256 ** -0.5)
        assert_close(ctx, ctx_ref, TOL, f"ctx n={n}")
        assert_close(qd.grad, dq_ref, TOL, f"dq n={n}")
        assert_close(kd.grad, dk_ref, TOL, f"dk n={n}")
        assert_close(vd.grad, dv_ref, TOL, f"dv n={n}")


def _proj_reference(q, k, v, w, g, scale):
    """project_out(attention(q, k, v)) and its gradients in fp64: the reference's own ops (cab.py:149-155)."""
    q64, k64, v64, w64 = (t.double().requires_grad_(True) for t in (q, k, v, w))
    attn = torch.softmax(torch.bmm(q64.transpose(1, 2), k64) * scale, dim=-1)
    ctx = torch.bmm(v64, attn.transpose(1, 2))
    glob = torch.einsum("oc,bcn->bon", w64, ctx)
    glob.backward(g.double())
    return glob.detach(), ctx.detach(), q64.grad, k64.grad, v64.grad, w64.grad


@pytest.mark.parametrize("B,Kc,Vc,Co,n,fused", [(8, 128, 128, 256, 1024, True),   # BASELINE config 3's CAB
                                                (8, 128, 128, 256, 1000, True),   # ragged last query tile
                                                (1, 128, 128, 64, 4, True), (1, 128, 128, 256, 36, True),
                                                (8, 64, 64, 512, 1024, True),     # two output blocks per wave
                                                (16, 128, 128, 32, 512, True),
                                                (2, 128, 128, 256, 2048, False),  # key split (config 5): K1 + small GEMM
                                                (2, 128, 128, 256, 1001, False),  # n % 4 != 0
                                                (1, 256, 128, 512, 160, False)])  # Kc = 256: one wave per SIMD kernel
def test_attn_proj_autograd_vs_fp64(B, Kc, Vc, Co, n, fused):
    """cab_attention_proj = project_out(attention): K1 with the projection in its epilogue (one launch) where
    cabinet_cab_attn_proj_supported says so, K1 + conv1x1 elsewhere -- both against the fp64 composite, forward and all four
    gradients; the fused form is also bit-reproducible and writes no ctx under no_grad."""
    from cabinet_amd.functional import cab_attention_proj, cab_attention_proj_supported

    gen = torch.Generator().manual_seed(77 + n + Co)
    q = torch.randn(B, Kc, n, generator=gen).relu()
    k = torch.randn(B, Kc, n, generator=gen)
    v = torch.randn(B, Vc, n, generator=gen)
    w = torch.randn(Co, Vc, 1, 1, generator=gen) * Vc ** -0.5
    g = torch.randn(B, Co, n, generator=gen)
    scale = Kc ** -0.5
    qd, kd, vd, wd = (t.cuda().requires_grad_(True) for t in (q, k, v, w))
    assert cab_attention_proj_supported(qd, vd, wd) == fused
    glob = cab_attention_proj(qd, kd, vd, wd, scale)
    glob.backward(g.cuda())
    torch.cuda.synchronize()
    ref, _, dq, dk, dv, dw = _proj_reference(q, k, v, w.flatten(1), g, scale)
    assert_close(glob, ref, 2e-5, "glob vs fp64")
    assert_close(qd.grad, dq, 1e-4, "dq")
    assert_close(kd.grad, dk, 1e-4, "dk")
    assert_close(vd.grad, dv, 1e-4, "dv")
    assert_close(wd.grad.flatten(1), dw, 1e-4, "dw_out")
    with torch.no_grad():
        again = cab_attention_proj(qd, kd, vd, wd, scale)
    assert torch.equal(again, glob.detach())


def test_attn_proj_equals_the_two_launch_form():
    """Fused epilogue vs cab_attention followed by conv1x1 on the same tensors (config-3 grid): the projection contracts the SAME
    fp32 context values, in another order -- agreement to fp32 rounding, and the C entry point refuses what it does not take."""
    import cabinet_amd.functional as Fn
    from cabinet_amd import _lib

    B, C, n, Co = 8, 128, 1024, 256
    gen = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, C, n, generator=gen).cuda() for _ in range(3))
    w = (torch.randn(Co, C, generator=gen) * C ** -0.5).cuda()
    fused = Fn.cab_attention_proj(q, k, v, w, C ** -0.5)
    ctx = Fn.cab_attention(q, k, v, C ** -0.5)
    two = Fn.conv1x1(ctx.reshape(B, C, n, 1), w).reshape(B, Co, n)
    assert rel_err(fused, two) < 2e-6
    lib = _lib.load()
    assert lib.cabinet_cab_attn_proj_supported(2, 128, 128, 256, 2048) == 0   # key split
    assert lib.cabinet_cab_attn_proj_supported(8, 128, 128, 250, 1024) == 0   # Co % 32
    assert lib.cabinet_cab_attn_proj_supported(8, 256, 128, 256, 1024) == 0   # Kc = 256
    glob = torch.empty(2, 256, 2048, device="cuda")
    lse = torch.empty(2, 2048, device="cuda")
    q2 = torch.randn(2, 128, 2048, device="cuda")
    rc = lib.cabinet_cab_attn_proj_fwd(q2.data_ptr(), q2.data_ptr(), q2.data_ptr(), w.data_ptr(), 1.0, 2, 128, 128, 256, 2048, None,
                                       glob.data_ptr(), lse.data_ptr(), None)
    assert rc == -2 and b"outside the fused form" in lib.cabinet_last_error()
