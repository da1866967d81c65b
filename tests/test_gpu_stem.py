"""HIP 7x7/2 stem convolution (K9, through the C ABI) vs the oracle: fp64 F.conv2d, the call
oracle/model_ref.py makes for SpatialBranch.conv1 (reference cabinet.py:111) -- odd and ragged image sizes,
sizes smaller than one tile, the config-3 size, and the optional input gradient."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 37, 53), (3, 8, 200), (1, 5, 5), (2, 130, 66), (1, 512, 512)])
def test_stem_conv_vs_oracle(B, H, W):
    from cabinet_amd.functional import stem_conv, stem_conv_supported

    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    g0 = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(B, 3, H, W, generator=g0)
    xo, wo = x.double(), conv.weight.detach().double().requires_grad_(True)
    yo = F.conv2d(xo, wo, None, 2, 3)
    g = torch.randn(yo.shape, generator=g0)
    yo.backward(g.double())
    conv = conv.cuda()
    assert stem_conv_supported(conv)
    y = stem_conv(x.cuda(), conv)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert y.shape == yo.shape
    assert_close(y, yo, TOL, "y")
    assert_close(conv.weight.grad, wo.grad, TOL, "dw")


def test_stem_conv_input_grad_dispatch_and_determinism():
    from cabinet_amd.functional import stem_conv, stem_conv_supported

    assert not stem_conv_supported(torch.nn.Conv2d(3, 64, 7, 2, 3, bias=True))
    assert not stem_conv_supported(torch.nn.Conv2d(3, 32, 7, 2, 3, bias=False))
    assert not stem_conv_supported(torch.nn.Conv2d(3, 64, 3, 2, 1, bias=False))
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    x = torch.randn(2, 3, 96, 80, device="cuda", requires_grad=True)
    g = torch.randn(2, 64, 48, 40, device="cuda")
    stem_conv(x, conv).backward(g)
    xo = x.detach().double().requires_grad_(True)
    F.conv2d(xo, conv.weight.detach().double(), None, 2, 3).backward(g.double())
    assert_close(x.grad, xo.grad, TOL, "dx (ATen backward-data, only when requested)")
    runs = []
    xn = x.detach()
    for _ in range(2):
        conv.zero_grad()
        out = stem_conv(xn, conv)
        out.backward(g)
        runs.append((out.clone(), conv.weight.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
