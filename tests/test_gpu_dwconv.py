"""HIP depthwise convolution (K8, through the C ABI) vs the oracle: fp64 F.conv2d with groups == channels, i.e. the
call oracle/model_ref.py::_mobilenet makes for mobilenetv3.py:118-126 -- kernels 3 and 5, strides 1 and 2, odd and
ragged sizes, planes smaller than a tile and planes spanning many tiles."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32; measured ~1e-7


@pytest.mark.parametrize("K,S", [(3, 1), (3, 2), (5, 1), (5, 2)])
@pytest.mark.parametrize("B,C,H,W", [(2, 8, 32, 32), (1, 3, 7, 9), (2, 4, 70, 130), (1, 5, 1, 1), (3, 2, 33, 17),
                                     (1, 16, 128, 128)])
def test_dwconv_vs_oracle(B, C, H, W, K, S):
    conv = torch.nn.Conv2d(C, C, K, S, K // 2, groups=C, bias=False)
    g0 = torch.Generator().manual_seed(K * 10 + S)
    x = torch.randn(B, C, H, W, generator=g0)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, 1, K, K, generator=g0))
    xo = x.double().requires_grad_(True)
    wo = conv.weight.detach().double().requires_grad_(True)
    yo = F.conv2d(xo, wo, None, S, K // 2, 1, C)
    g = torch.randn(yo.shape, generator=g0)
    yo.backward(g.double())

    from cabinet_amd.functional import dwconv, dwconv_supported

    conv = conv.cuda()
    assert dwconv_supported(conv)
    xd = x.cuda().requires_grad_(True)
    y = dwconv(xd, conv)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert y.shape == yo.shape
    assert_close(y, yo, TOL, "y")
    assert_close(xd.grad, xo.grad, TOL, "dx")
    assert_close(conv.weight.grad, wo.grad, TOL, "dw")


def test_dwconv_dispatch_and_determinism():
    from cabinet_amd.functional import dwconv, dwconv_supported

    assert not dwconv_supported(torch.nn.Conv2d(8, 8, 3, 1, 1, groups=4, bias=False))      # grouped, not depthwise
    assert not dwconv_supported(torch.nn.Conv2d(8, 8, 7, 1, 3, groups=8, bias=False))      # kernel 7
    assert not dwconv_supported(torch.nn.Conv2d(8, 8, 3, 1, 1, groups=8, bias=True))       # bias
    assert not dwconv_supported(torch.nn.Conv2d(8, 8, 3, 1, 2, 2, groups=8, bias=False))   # dilation
    conv = torch.nn.Conv2d(24, 24, 5, 2, 2, groups=24, bias=False).cuda()
    x = torch.randn(2, 24, 75, 61, device="cuda", requires_grad=True)
    g = torch.randn(2, 24, 38, 31, device="cuda")
    runs = []
    for _ in range(2):
        conv.zero_grad()
        x.grad = None
        dwconv(x, conv).backward(g)
        runs.append((x.grad.clone(), conv.weight.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("K,S,act", [(3, 1, "relu"), (3, 2, "hardswish"), (5, 1, "hardswish"), (5, 2, "relu")])
@pytest.mark.parametrize("B,C,H,W", [(2, 8, 32, 32), (1, 3, 7, 9), (2, 4, 70, 130), (3, 2, 33, 17)])
def test_bn_act_dwconv_vs_oracle(B, C, H, W, K, S, act, training):
    """conv(act(bn(z))) as one operator (the BatchNorm folded into the convolution's loader, its backward partial sums
    emitted by the convolution's backward) vs fp64 F.batch_norm -> activation -> F.conv2d."""
    from cabinet_amd.functional import bn_act_dwconv
    from oracle.model_ref import _hswish

    g0 = torch.Generator().manual_seed(K * 100 + S * 10 + C)
    conv = torch.nn.Conv2d(C, C, K, S, K // 2, groups=C, bias=False)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, 1, K, K, generator=g0))
        bn.weight.copy_(torch.rand(C, generator=g0) + 0.5)
        bn.bias.copy_(torch.rand(C, generator=g0) - 0.5)
        bn.running_mean.copy_(torch.rand(C, generator=g0) - 0.5)
        bn.running_var.copy_(torch.rand(C, generator=g0) + 0.5)
    z = torch.randn(B, C, H, W, generator=g0) * 1.5 + 0.3
    zo = z.double().requires_grad_(True)
    wo, bo = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    cwo = conv.weight.detach().double().requires_grad_(True)
    rm, rv = bn.running_mean.double().clone(), bn.running_var.double().clone()
    a = {"relu": F.relu, "hardswish": _hswish}[act](F.batch_norm(zo, rm, rv, wo, bo, training, 0.1, 1e-5))
    yo = F.conv2d(a, cwo, None, S, K // 2, 1, C)
    g = torch.randn(yo.shape, generator=g0)
    yo.backward(g.double())

    conv, bn = conv.cuda(), bn.cuda().train(training)
    zd = z.cuda().requires_grad_(True)
    y = bn_act_dwconv(zd, bn, act, conv)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert_close(y, yo, TOL, "y")
    assert_close(zd.grad, zo.grad, TOL, "dz")
    assert_close(bn.weight.grad, wo.grad, TOL, "dbn_weight")
    assert_close(bn.bias.grad, bo.grad, TOL, "dbn_bias")
    assert_close(conv.weight.grad, cwo.grad, TOL, "dconv_weight")
    assert_close(bn.running_mean, rm, 1e-5, "running_mean")
    assert_close(bn.running_var, rv, 1e-5, "running_var")
