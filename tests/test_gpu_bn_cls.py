"""K12 -- BatchNorm2d -> ReLU -> 1x1 classifier as one streaming operator (cabinet_bn_cls_fwd / _bwd, through the C ABI) vs the
reference's own ops in fp64 (`nn.BatchNorm2d -> nn.ReLU -> nn.Conv2d(1x1)`: cabinet.py:90-92 and :161-172 are exactly F.batch_norm,
F.relu, F.conv2d): all three class-count instantiations (8, 20, 32), with / without bias, both BatchNorm modes, running-buffer
side effects, statistics from K11's epilogue, the production grids of BASELINE configs 3 and 5, the fall-back for shapes outside
the coverage, bit-reproducibility, error codes."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import assert_close, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32; measured values next to each assert


def _modules(C, K, bias, gen):
    bn, cls = nn.BatchNorm2d(C), nn.Conv2d(C, K, 1, bias=bias)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5)
        bn.bias.copy_(torch.rand(C, generator=gen) - 0.5)
        bn.running_mean.copy_(torch.rand(C, generator=gen) - 0.5)
        bn.running_var.copy_(torch.rand(C, generator=gen) + 0.5)
        cls.weight.copy_(torch.randn(K, C, 1, 1, generator=gen) * C ** -0.5)
        if bias:
            cls.bias.copy_(torch.randn(K, generator=gen) * 0.1)
    return bn, cls


def _oracle(z, g, bn, cls, training):
    zo = z.detach().cpu().double().requires_grad_(True)
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in
         (("gw", bn.weight), ("gb", bn.bias), ("w", cls.weight))}
    b = cls.bias.detach().cpu().double().requires_grad_(True) if cls.bias is not None else None
    rm, rv = bn.running_mean.detach().cpu().double().clone(), bn.running_var.detach().cpu().double().clone()
    y = F.conv2d(F.relu(F.batch_norm(zo, rm, rv, p["gw"], p["gb"], training, 0.1, 1e-5)), p["w"], b)
    y.backward(g.cpu().double())
    return dict(y=y.detach(), dz=zo.grad, dgw=p["gw"].grad, dgb=p["gb"].grad, dw=p["w"].grad, db=(b.grad if b is not None else None),
                rm=rm, rv=rv)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape,K,bias", [((8, 256, 32, 32), 8, True),      # ab.b2 -> b4 at config 3
                                          ((2, 256, 64, 32), 19, True),     # ... at config 5 (19 classes: the 20-wide instantiation)
                                          ((2, 64, 8, 8), 8, False), ((1, 64, 2, 2), 1, True), ((3, 128, 6, 6), 3, True),
                                          ((2, 64, 10, 10), 20, True), ((2, 64, 12, 8), 32, False), ((1, 192, 66, 70), 5, True)])
def test_bn_relu_cls_vs_fp64(shape, K, bias, training):
    from cabinet_amd.functional import bn_relu_cls, bn_relu_cls_supported

    gen = torch.Generator().manual_seed(11 + K)
    bn, cls = _modules(shape[1], K, bias, gen)
    z = torch.randn(*shape, generator=gen) * 1.3 + 0.4   # non-zero mean: exercises the variance merge
    g = torch.randn(shape[0], K, shape[2], shape[3], generator=gen)
    ref = _oracle(z, g, bn, cls, training)
    bn, cls = bn.cuda().train(training), cls.cuda()
    zd = z.cuda().requires_grad_(True)
    assert bn_relu_cls_supported(zd, bn, cls)
    y = bn_relu_cls(zd, bn, cls)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert_close(y, ref["y"], 2e-5, "y")                      # measured ~3e-7
    assert_close(zd.grad, ref["dz"], TOL, "dz")               # measured ~5e-7 (no unit of these maps sits within rounding of zero)
    assert_close(bn.weight.grad, ref["dgw"], TOL, "dgamma")
    assert_close(bn.bias.grad, ref["dgb"], TOL, "dbeta")
    assert_close(cls.weight.grad, ref["dw"], 2e-5, "dw_cls")
    if bias:
        assert_close(cls.bias.grad, ref["db"], 2e-5, "dbias")
    assert_close(bn.running_mean, ref["rm"], 1e-5, "running_mean")
    assert_close(bn.running_var, ref["rv"], 1e-5, "running_var")


def test_bn_relu_cls_equals_k7_plus_stock_convolution_and_is_bit_reproducible():
    """Same mask expression as K7 (`fmaf((z - mean) * invstd, gamma, beta) > 0`): against bn_act + F.conv2d on the SAME tensors no
    ReLU unit can differ, so every output agrees to fp32 summation order -- and twice the same call gives the same bits."""
    import cabinet_amd.functional as Fn

    gen = torch.Generator().manual_seed(3)
    B, C, H, W, K = 4, 256, 48, 40, 8
    bn, cls = _modules(C, K, True, gen)
    bn2, cls2 = _modules(C, K, True, torch.Generator().manual_seed(3))
    z = torch.randn(B, C, H, W, generator=gen)
    g = torch.randn(B, K, H, W, generator=gen).cuda()
    outs = []
    for fused, (b_, c_) in ((True, (bn, cls)), (False, (bn2, cls2)), (True, _modules(C, K, True, torch.Generator().manual_seed(3)))):
        b_, c_ = b_.cuda().train(), c_.cuda()
        Fn.BN_CLS_ENABLED = fused
        try:
            zd = z.cuda().requires_grad_(True)
            y = Fn.bn_relu_cls(zd, b_, c_)
            y.backward(g)
        finally:
            Fn.BN_CLS_ENABLED = True
        outs.append((y.detach(), zd.grad, b_.weight.grad, b_.bias.grad, c_.weight.grad, c_.bias.grad, b_.running_mean, b_.running_var))
    for name, a, b in zip(("y", "dz", "dgamma", "dbeta", "dw", "dbias", "running_mean", "running_var"), outs[0], outs[1]):
        assert rel_err(a, b) < 3e-6, (name, rel_err(a, b))
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,C0,C1,H,W,K", [(2, 64, 64, 16, 24, 8), (8, 256, 0, 128, 128, 8), (2, 256, 0, 256, 128, 19)])
def test_k11_then_k12_statistics_from_the_convolution_epilogue(B, C0, C1, H, W, K):
    """conv3x3 -> bn_relu_cls with the convolution's (mean, M2) partials (no statistics pass over z) against the same chain with
    K12's own statistics pass: same batch statistics to fp32 rounding, so the logits and every gradient agree; the config-3 and
    config-5 grids of conv_out (cabinet.py:160-172) are two of the cases."""
    from cabinet_amd.functional import bn_relu_cls, conv3x3, conv3x3_bn_part

    gen = torch.Generator().manual_seed(5)
    Co = 256 if C0 >= 256 else 64
    x0 = torch.randn(B, C0, H, W, generator=gen).cuda()
    x1 = torch.randn(B, C1, H, W, generator=gen).cuda() if C1 else None
    w3 = (torch.randn(Co, C0 + C1, 3, 3, generator=gen) * (9 * (C0 + C1)) ** -0.5).cuda()
    g = torch.randn(B, K, H, W, generator=gen).cuda()
    res = []
    for with_part in (True, False):
        bn, cls = _modules(Co, K, True, torch.Generator().manual_seed(9))
        bn, cls = bn.cuda().train(), cls.cuda()
        part = conv3x3_bn_part(x0, Co) if with_part else None
        z = conv3x3(x0, w3, x1, part).requires_grad_(True)
        y = bn_relu_cls(z, bn, cls, conv_part=part)
        y.backward(g)
        res.append((y.detach(), z.grad, bn.weight.grad, bn.bias.grad, cls.weight.grad, bn.running_mean, bn.running_var))
    for name, a, b in zip(("y", "dz", "dgamma", "dbeta", "dw", "running_mean", "running_var"), *res):
        assert rel_err(a, b) < 2e-5, (name, rel_err(a, b))   # a handful of units within 1e-7 of zero may flip between the two means


def test_bn_relu_cls_adjoint_identities_at_config3_grid():
    """Size-independent properties at conv_out's production grid (8 x 256 x 128 x 128, 8 classes), eval-mode BatchNorm (then the
    operator is piecewise linear in z with a fixed mask): <y(z) - y(0-activation), g> pairs -- dw and dbias are the exact adjoints of
    the forward in the classifier parameters: <y, g> = <w, dw> + <bias, dbias>; dz is the adjoint of the linearisation:
    <J dz_dir, g> = <dz_dir, dz> for a random direction supported away from the kinks."""
    from cabinet_amd.functional import bn_relu_cls

    gen = torch.Generator().manual_seed(8)
    B, C, H, W, K = 8, 256, 128, 128, 8
    bn, cls = _modules(C, K, True, gen)
    bn, cls = bn.cuda().eval(), cls.cuda()
    z = torch.randn(B, C, H, W, generator=gen).cuda().requires_grad_(True)
    g = torch.randn(B, K, H, W, generator=gen).cuda()
    y = bn_relu_cls(z, bn, cls)
    y.backward(g)
    lhs = float((y.double() * g.double()).sum())
    rhs = float((cls.weight.double() * cls.weight.grad.double()).sum() + (cls.bias.double() * cls.bias.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), float(y.double().norm() * g.double().norm()) * 1e-3), (lhs, rhs)
    # directional derivative in z: finite step small enough that (almost) no unit crosses zero
    d = torch.randn(B, C, H, W, generator=gen).cuda()
    with torch.no_grad():
        eps = 1e-3
        yp, ym = bn_relu_cls(z + eps * d, bn, cls), bn_relu_cls(z - eps * d, bn, cls)
    fd = float(((yp.double() - ym.double()) / (2 * eps) * g.double()).sum())
    an = float((d.double() * z.grad.double()).sum())
    assert abs(fd - an) <= 2e-3 * float(d.double().norm() * z.grad.double().norm()), (fd, an)


def test_bn_relu_cls_falls_back_outside_the_coverage_and_reports_errors():
    import cabinet_amd.functional as Fn
    from cabinet_amd import _lib

    gen = torch.Generator().manual_seed(2)
    for shape, K in (((2, 48, 8, 8), 4), ((2, 64, 5, 5), 4), ((1, 64, 8, 8), 40)):   # C % 64, H*W % 4, K > 32
        bn, cls = _modules(shape[1], K, True, gen)
        bn, cls = bn.cuda().train(), cls.cuda()
        z = torch.randn(*shape, generator=gen).cuda().requires_grad_(True)
        assert not Fn.bn_relu_cls_supported(z, bn, cls)
        y = Fn.bn_relu_cls(z, bn, cls)          # K7 + the stock convolution
        ref = cls(F.relu(F.batch_norm(z, None, None, bn.weight, bn.bias, True, 0.1, 1e-5)))
        assert rel_err(y, ref) < 1e-5
    lib = _lib.load()
    assert lib.cabinet_bn_cls_supported(256, 8, 16384) == 1 and lib.cabinet_bn_cls_supported(250, 8, 16384) == 0
    assert lib.cabinet_bn_cls_table_floats(256, 8) == 256 * 16 and lib.cabinet_bn_cls_table_floats(256, 19) == 256 * 28
    z = torch.randn(1, 64, 4, 4, device="cuda")
    y = torch.empty(1, 4, 4, 4, device="cuda")
    tab = torch.empty(64 * 16, device="cuda")
    v = torch.ones(64, device="cuda")
    w = torch.zeros(4, 64, device="cuda")
    rc = lib.cabinet_bn_cls_fwd(z.data_ptr(), None, v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), w.data_ptr(), None, 1, 64, 4, 4, 4,
                                1, 0.1, 1e-5, y.data_ptr(), tab.data_ptr(), None, 0, None)
    assert rc == -3 and b"workspace" in lib.cabinet_last_error()
    rc = lib.cabinet_bn_cls_fwd(z.data_ptr(), None, v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), w.data_ptr(), None, 1, 60, 4, 4, 4,
                                1, 0.1, 1e-5, y.data_ptr(), tab.data_ptr(), None, 0, None)
    assert rc == -2
