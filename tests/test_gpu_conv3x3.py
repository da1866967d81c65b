"""K11 -- dense 3x3 convolution (Winograd F(2x2,3x3), fp32 MFMA; cabinet_amd/csrc/conv3x3_wino.hip) through the C ABI against the
oracle: what torch's CPU convolution computes in fp64 (the reference's nn.Conv2d at src/models/cabinet.py:59, :68 + :88-89, :160 IS
``F.conv2d``; SURVEY.md section 8(c): "the oracle is whatever torch CPU computes").

Tolerance: 1e-3 relative per tensor (||a-b|| / ||b||), the contract of BASELINE.json's north_star; measured 2e-7 .. 5e-7.
Full-size grids (BASELINE configs 3 and 5) are checked through size-independent properties -- the adjoint identities
<conv(x,w), dy> = <x, dgrad(dy,w)> = <w, wgrad(dy,x)> in fp64 and spot outputs recomputed directly -- because a CPU fp64
convolution of 155 GFLOP is minutes, not seconds.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _case(B, C0, C1, K, H, W, seed=0):
    g = torch.Generator().manual_seed(seed + 17 * H + W)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(K, C0 + C1, 3, 3, generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    dy = torch.randn(B, K, H, W, generator=g)
    return x0, x1, w, dy


def _oracle(x0, x1, w, dy):
    xin = (torch.cat([x0, x1], 1) if x1 is not None else x0).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv2d(xin, wd, padding=1)
    y.backward(dy.double())
    return y.detach(), xin.grad, wd.grad


SHAPES = [
    (2, 64, 0, 64, 8, 8),       # smallest supported block
    (1, 64, 64, 64, 6, 34),     # two inputs; 17 tile columns: a second, ragged tile-block column
    (2, 128, 0, 64, 7, 9),      # odd height and width: masked last row / column, scalar stores
    (1, 64, 0, 128, 33, 20),    # odd height, two channel blocks
    (3, 64, 0, 64, 1, 1),       # one pixel: every tap but the centre is padding
    (1, 64, 0, 64, 1, 40),      # one row
    (1, 64, 0, 64, 37, 1),      # one column
    (2, 192, 64, 128, 16, 32),  # C0 = 3 blocks + C1 = 1 block, full-width tile blocks
    (1, 576, 256, 256, 8, 16),  # CABiNet-Small fusion head channels
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv3x3_fwd_bwd_vs_fp64_oracle(shape):
    from cabinet_amd.functional import conv3x3_bwd_hip, conv3x3_fwd_hip

    B, C0, C1, K, H, W = shape
    x0, x1, w, dy = _case(*shape)
    y_ref, dx_ref, dw_ref = _oracle(x0, x1, w, dy)
    dev = torch.device("cuda", 0)
    d = lambda t: t.to(dev) if t is not None else None  # noqa: E731
    y = conv3x3_fwd_hip(d(x0), d(x1), d(w))
    dx0, dx1, dw = conv3x3_bwd_hip(d(dy), d(x0), d(x1), d(w))
    torch.cuda.synchronize()
    errs = {"y": rel(y, y_ref), "dx0": rel(dx0, dx_ref[:, :C0]), "dw": rel(dw, dw_ref)}
    if C1:
        errs["dx1"] = rel(dx1, dx_ref[:, C0:])
    assert all(e < TOL for e in errs.values()), errs
    assert max(errs.values()) < 1e-5, errs   # what fp32 Winograd delivers here; a regression shows long before 1e-3


K128_SHAPES = [
    (1, 64, 0, 128, 256, 256),    # 512 workgroups of 128 channels: the one-wave-per-SIMD kernel by its own rule
    (2, 64, 64, 256, 130, 132),   # two inputs, two channel blocks, ragged tile blocks (33 x 5 per image, last ones partial)
    (1, 192, 0, 128, 6, 64),      # forced onto a small grid: one tile-block row and a half, twelve chunks
]


@pytest.mark.parametrize("shape", K128_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv3x3_128_channel_kernel_vs_fp64_oracle_and_equals_64_channel_kernel(shape, monkeypatch):
    """Round 6: wino_conv128_kernel (128 output channels per workgroup, one wave per SIMD, 16 accumulator tiles per wave) serves the
    forward and the data gradient of grids with at least two rounds of workgroups (conv_out, cabinet.py:160).  It contracts the
    channels in the order of the 64-channel kernel -- chunk by chunk, k-step by k-step on one fma chain per output -- so outputs,
    gradients and the BatchNorm partials of its epilogue must equal that kernel's BIT FOR BIT, and the fp64 oracle to 1e-5."""
    from cabinet_amd import _lib
    from cabinet_amd.functional import conv3x3_bwd_hip, conv3x3_fwd_hip

    B, C0, C1, K, H, W = shape
    x0, x1, w, dy = _case(*shape, seed=23)
    y_ref, dx_ref, dw_ref = _oracle(x0, x1, w, dy)
    d = lambda t: t.cuda() if t is not None else None  # noqa: E731
    nblk = _lib.load().cabinet_conv3x3_tile_blocks(B, H, W)
    res = {}
    for flag in ("1", "2", "0"):   # 1: the 128-channel kernel; 2: its persistent all-xi-per-wave form (measured slower, kept); 0: 64 channels
        monkeypatch.setenv("CABINET_WINO_128", flag)
        part = torch.full((2, K, nblk), float("nan"), device="cuda")
        y = conv3x3_fwd_hip(d(x0), d(x1), d(w), bn_part=part)
        res[flag] = (y, part) + tuple(conv3x3_bwd_hip(d(dy), d(x0), d(x1), d(w)))
    torch.cuda.synchronize()
    y, part, dx0, dx1, dw = res["1"]
    errs = {"y": rel(y, y_ref), "dx0": rel(dx0, dx_ref[:, :C0]), "dw": rel(dw, dw_ref)}
    if C1:
        errs["dx1"] = rel(dx1, dx_ref[:, C0:])
    assert max(errs.values()) < 1e-5, errs
    assert torch.isfinite(part).all()
    for flag in ("1", "2"):
        for a, b, name in zip(res[flag], res["0"], ("y", "bn_part", "dx0", "dx1", "dw")):
            if a is not None:
                assert torch.equal(a, b), f"{name}: CABINET_WINO_128={flag} differs from the 64-channel kernel's bits (rel {rel(a, b):.2e})"


def test_conv3x3_autograd_function_matches_stock_module_and_skips_unneeded_grads():
    """The autograd wrapper the model uses, against nn.Conv2d's own autograd in fp64 on the host; weight-only and input-only
    gradient requests run only their half."""
    from cabinet_amd.functional import conv3x3

    torch.manual_seed(3)
    conv = torch.nn.Conv2d(128, 64, 3, padding=1, bias=False)
    x = torch.randn(2, 64, 12, 20)
    f = torch.randn(2, 64, 12, 20)
    ref_x, ref_f = x.double().requires_grad_(True), f.double().requires_grad_(True)
    ref = conv.double()(torch.cat([ref_x, ref_f], 1))
    g = torch.randn_like(ref)
    ref.backward(g)
    wref = conv.weight.grad.clone()
    conv = conv.float().cuda()
    conv.weight.grad = None
    xd, fd = x.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
    y = conv3x3(xd, conv.weight, fd)
    y.backward(g.float().cuda())
    assert rel(y, ref) < 1e-5 and rel(xd.grad, ref_x.grad) < 1e-5 and rel(fd.grad, ref_f.grad) < 1e-5
    assert rel(conv.weight.grad, wref) < 1e-5
    # no input gradient requested: only dw
    conv.weight.grad = None
    conv3x3(x.cuda(), conv.weight, f.cuda()).backward(g.float().cuda())
    assert rel(conv.weight.grad, wref) < 1e-5
    # frozen weight: only the data gradient
    wfix = conv.weight.detach()
    xd2 = x.cuda().requires_grad_(True)
    conv3x3(xd2, wfix, f.cuda()).backward(g.float().cuda())
    assert rel(xd2.grad, ref_x.grad) < 1e-5


def test_conv3x3_is_bit_reproducible_and_two_pointer_equals_concat():
    """No atomics anywhere (ordered slab sums): two runs agree bit for bit; reading (x, feat) through two pointers gives the
    bits of the same kernel on the materialised concat."""
    from cabinet_amd.functional import conv3x3_bwd_hip, conv3x3_fwd_hip

    x0, x1, w, dy = (t.cuda() for t in _case(2, 192, 64, 128, 20, 36, seed=5))
    a = (conv3x3_fwd_hip(x0, x1, w),) + conv3x3_bwd_hip(dy, x0, x1, w)
    b = (conv3x3_fwd_hip(x0, x1, w),) + conv3x3_bwd_hip(dy, x0, x1, w)
    for s, t in zip(a, b):
        assert torch.equal(s, t)
    xc = torch.cat([x0, x1], 1)
    yc = conv3x3_fwd_hip(xc, None, w)
    dxc, _, dwc = conv3x3_bwd_hip(dy, xc, None, w)
    assert torch.equal(yc, a[0]) and torch.equal(dxc[:, :192], a[1]) and torch.equal(dxc[:, 192:], a[2]) and torch.equal(dwc, a[3])


def test_conv3x3_bn_partials_give_the_batch_statistics():
    """The forward's optional epilogue output: per channel and tile block the mean and the sum of squared deviations of the block's
    valid outputs.  Chan-merged in double they must reproduce BatchNorm2d's batch mean and biased variance of y -- including for
    a channel whose mean is 100x its standard deviation (one-pass E[y^2] - mean^2 in fp32 would not)."""
    from cabinet_amd import _lib
    from cabinet_amd.functional import conv3x3_fwd_hip

    lib = _lib.load()
    for (B, C, K, H, W) in [(2, 64, 64, 9, 35), (3, 64, 128, 32, 32)]:
        x0, _, w, _ = _case(B, C, 0, K, H, W, seed=11)
        x0 = x0 + 3.0
        w[0] = w[0].abs() * 10.0   # channel 0: large positive mean
        nblk = lib.cabinet_conv3x3_tile_blocks(B, H, W)
        part = torch.full((2, K, nblk), float("nan"), device="cuda")
        y = conv3x3_fwd_hip(x0.cuda(), None, w.cuda(), bn_part=part)
        torch.cuda.synchronize()
        nby, nbx = ((H + 1) // 2 + 1) // 2, ((W + 1) // 2 + 15) // 16
        cnt = torch.tensor([min(4, H - 4 * by) * min(32, W - 32 * bx) for _ in range(B) for by in range(nby) for bx in range(nbx)],
                           dtype=torch.float64)
        assert cnt.numel() == nblk and int(cnt.sum()) == B * H * W
        mean_b, m2_b = part[0].double().cpu(), part[1].double().cpu()
        n = cnt.sum()
        mean = (mean_b * cnt).sum(1) / n
        m2 = m2_b.sum(1) + (cnt * (mean_b - mean[:, None]) ** 2).sum(1)
        yd = y.double().cpu()
        assert rel(mean, yd.mean((0, 2, 3))) < 1e-6
        assert rel(m2 / n, yd.var((0, 2, 3), unbiased=False)) < 1e-5


FULL = {
    "config3-conva": (8, 960, 0, 256, 32, 32), "config3-b1": (8, 960, 256, 256, 32, 32), "config3-conv_out": (8, 256, 0, 256, 128, 128),
    "config5-conva": (2, 960, 0, 256, 128, 64), "config5-b1": (2, 960, 256, 256, 128, 64), "config5-conv_out": (2, 256, 0, 256, 256, 128),
}


@pytest.mark.parametrize("name", sorted(FULL))
def test_conv3x3_production_grids_adjoint_identities_and_spot_values(name):
    """At the grids of BASELINE configs 3 and 5: (i) 64 output values recomputed directly in fp64 from the 3x3x C window,
    (ii) 64 data-gradient and 64 weight-gradient values likewise, (iii) the adjoint identities in fp64 -- forward, data gradient
    and weight gradient are three views of ONE trilinear form, so <y, dy> = <x, dx> = <w, dw> ties the three kernels together
    at full size."""
    from cabinet_amd.functional import conv3x3_bwd_hip, conv3x3_fwd_hip

    B, C0, C1, K, H, W = FULL[name]
    C = C0 + C1
    x0, x1, w, dy = _case(B, C0, C1, K, H, W, seed=23)
    d = lambda t: t.cuda() if t is not None else None  # noqa: E731
    y = conv3x3_fwd_hip(d(x0), d(x1), d(w))
    dx0, dx1, dw = conv3x3_bwd_hip(d(dy), d(x0), d(x1), d(w))
    torch.cuda.synchronize()
    y, dx0, dw = y.cpu(), dx0.cpu(), dw.cpu()
    x = torch.cat([x0, x1], 1) if C1 else x0
    dx = torch.cat([dx0, dx1.cpu()], 1) if C1 else dx0
    xp, dyp = F.pad(x.double(), (1, 1, 1, 1)), F.pad(dy.double(), (1, 1, 1, 1))
    wd = w.double()
    g = torch.Generator().manual_seed(1)
    pick = lambda n: int(torch.randint(0, n, (1,), generator=g))  # noqa: E731
    ey, edx, edw = [], [], []
    for i in range(64):
        b, k, c = pick(B), pick(K), pick(C)
        oy, ox = (0, H - 1, pick(H))[i % 3], (0, W - 1, pick(W))[(i // 3) % 3]   # borders and interior
        ref = (xp[b, :, oy:oy + 3, ox:ox + 3] * wd[k]).sum()
        ey.append((float(y[b, k, oy, ox]), float(ref)))
        # dx[b,c,iy,ix] = sum_{k,r,s} dy[b,k,iy+1-r,ix+1-s] w[k,c,r,s]
        ref = (dyp[b, :, oy:oy + 3, ox:ox + 3] * wd[:, c].flip(1, 2)).sum()
        edx.append((float(dx[b, c, oy, ox]), float(ref)))
        r, s = pick(3), pick(3)
        ref = (dy[:, k].double() * xp[:, c, r:r + H, s:s + W]).sum()
        edw.append((float(dw[k, c, r, s]), float(ref)))
    for nm, pairs in (("y", ey), ("dx", edx), ("dw", edw)):
        a, r = torch.tensor(pairs, dtype=torch.float64).unbind(1)
        assert float((a - r).norm() / r.norm()) < 1e-5, (name, nm)
    ip_y = float((y.double() * dy.double()).sum())
    ip_x = float((x.double() * dx.double()).sum())
    ip_w = float((wd * dw.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(ip_y - ip_x) < 1e-6 * scale and abs(ip_y - ip_w) < 1e-6 * scale, (name, ip_y, ip_x, ip_w, scale)


def test_bn_act_from_conv_partials_equals_bn_act_with_its_own_statistics_pass():
    """cabinet_bn_act_fwd_part (statistics from K11's epilogue) against cabinet_bn_act_fwd (its own pass over x) and against
    torch's BatchNorm2d in fp64: output, saved statistics and running buffers; odd sizes (ragged blocks) included."""
    from cabinet_amd.functional import bn_act, conv3x3, conv3x3_bn_part

    for (B, C, K, H, W) in [(2, 64, 64, 9, 35), (4, 64, 128, 32, 32), (1, 128, 64, 5, 70)]:
        x0, _, w, _ = _case(B, C, 0, K, H, W, seed=31)
        xd, wd = x0.cuda(), w.cuda()
        bn_a, bn_b = torch.nn.BatchNorm2d(K).cuda().train(), torch.nn.BatchNorm2d(K).cuda().train()
        with torch.no_grad():
            bn_a.weight.uniform_(0.5, 1.5), bn_a.bias.normal_()
            bn_b.load_state_dict(bn_a.state_dict())
        part = conv3x3_bn_part(xd, K)
        y = conv3x3(xd, wd, None, part)
        out_a = bn_act(y, bn_a, "relu", conv_part=part)
        out_b = bn_act(y, bn_b, "relu")
        torch.cuda.synchronize()
        assert rel(out_a, out_b) < 1e-6
        assert rel(bn_a.running_mean, bn_b.running_mean) < 1e-6 and rel(bn_a.running_var, bn_b.running_var) < 1e-6
        ref_bn = torch.nn.BatchNorm2d(K).double().train()
        ref_bn.load_state_dict({k: v.double().cpu() for k, v in bn_b.state_dict().items() if k not in ("running_mean", "running_var", "num_batches_tracked")}, strict=False)
        ref = torch.relu(ref_bn(y.double().cpu()))
        assert rel(out_a, ref) < 1e-5
        assert rel(bn_a.running_var, ref_bn.running_var) < 1e-5 and int(bn_a.num_batches_tracked) == 1


def test_conv3x3_c_abi_errors():
    """Error behaviour of the entry points (include/cabinet_hip.h): unsupported channel counts, a short workspace and a misaligned
    pointer are error codes with a message, never a launch."""
    from cabinet_amd import _lib

    lib = _lib.load()
    assert lib.cabinet_conv3x3_supported(960, 256, 256) == 1 and lib.cabinet_conv3x3_supported(576, 256, 256) == 1
    assert lib.cabinet_conv3x3_supported(256, 0, 256) == 1
    assert lib.cabinet_conv3x3_supported(64, 0, 19) == 0       # Co not a multiple of 64
    assert lib.cabinet_conv3x3_supported(3, 0, 64) == 0        # the image stem: K9's job
    assert lib.cabinet_conv3x3_supported(96, 32, 64) == 0      # two inputs: C0 must be whole 64-channel blocks
    x = torch.zeros(1, 64, 8, 8, device="cuda")
    w = torch.zeros(64, 64, 3, 3, device="cuda")
    y = torch.empty(1, 64, 8, 8, device="cuda")
    need = lib.cabinet_conv3x3_fwd_workspace_bytes(1, 64, 0, 64, 8, 8)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.cabinet_conv3x3_fwd(x.data_ptr(), None, w.data_ptr(), 1, 64, 0, 64, 8, 8, y.data_ptr(), None, ws.data_ptr(), need - 1, st) == -3
    assert b"workspace" in lib.cabinet_last_error()
    assert lib.cabinet_conv3x3_fwd(x.data_ptr() + 4, None, w.data_ptr(), 1, 64, 0, 64, 8, 8, y.data_ptr(), None, ws.data_ptr(), need, st) == -1
    assert b"16-byte aligned" in lib.cabinet_last_error()
    assert lib.cabinet_conv3x3_fwd(x.data_ptr(), None, w.data_ptr(), 1, 48, 0, 64, 8, 8, y.data_ptr(), None, ws.data_ptr(), need, st) == -2
    assert lib.cabinet_conv3x3_fwd(x.data_ptr(), None, w.data_ptr(), 1, 64, 64, 64, 8, 8, y.data_ptr(), None, ws.data_ptr(), need, st) == -1  # x1 missing
    assert lib.cabinet_conv3x3_fwd(x.data_ptr(), None, w.data_ptr(), 1, 64, 0, 64, 8, 8, y.data_ptr(), None, ws.data_ptr(), need, st) == 0
    torch.cuda.synchronize()
    assert float(y.abs().sum()) == 0.0


def test_model_uses_k11_for_its_three_plain_3x3_convolutions(monkeypatch):
    """AttentionBranch.conva / .b1 and CABiNetOutput.conv go through cabinet_conv3x3_* (no torch.cat in the fusion head); the strided
    3x3 of the spatial branch does not; CABINET_CONV3X3=0 (functional.CONV3X3_ENABLED) restores the stock operator with results that
    agree to fp32 accuracy."""
    import cabinet_amd.functional as Fn
    from cabinet_amd.train import build_model

    calls = []
    orig = Fn._Conv3x3.apply

    def spy(x0, x1, weight, *part):
        calls.append((tuple(x0.shape), None if x1 is None else tuple(x1.shape), tuple(weight.shape), len(part)))
        return orig(x0, x1, weight, *part)

    monkeypatch.setattr(Fn._Conv3x3, "apply", staticmethod(spy))
    net = build_model("small", n_classes=8, device="cuda", seed=0, gamma=0.5).train()
    im = torch.randn(2, 3, 128, 128, device="cuda")
    out, out16 = net(im)
    assert [c[2] for c in calls] == [(256, 576, 3, 3), (256, 832, 3, 3), (256, 256, 3, 3)], calls
    assert calls[1][1] == (2, 256, 4, 4)   # feat rides along as the second pointer
    assert all(c[3] == 1 for c in calls)   # training mode: each hands its BatchNorm the statistics partials of its epilogue
    monkeypatch.setattr(Fn, "CONV3X3_ENABLED", False)
    net2 = build_model("small", n_classes=8, device="cuda", seed=0, gamma=0.5).train()
    n_before = len(calls)
    ref, ref16 = net2(im)
    assert len(calls) == n_before
    assert rel(out, ref) < 1e-4 and rel(out16, ref16) < 1e-4
