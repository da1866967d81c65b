"""HIP thin pointwise convolution (K10, through the C ABI) vs the oracle: fp64 F.conv2d with a 1x1 weight, the call
oracle/model_ref.py::_mobilenet makes for mobilenetv3.py:128-131,144-151 -- channel counts that are not multiples of
32, planes that are not multiples of the 64-pixel block, every row-block combination the backbone uses."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 16, 16, 64, 64), (2, 16, 64, 64, 64), (1, 64, 24, 70, 66), (2, 24, 72, 64, 64),
                                         (1, 72, 24, 64, 65), (2, 72, 40, 32, 32), (1, 40, 120, 64, 64),
                                         (1, 120, 40, 64, 64), (1, 8, 8, 3, 5), (3, 104, 56, 17, 19)])
def test_pwconv_vs_oracle(B, Ci, Co, H, W):
    from cabinet_amd import _lib
    from cabinet_amd.functional import pwconv

    assert _lib.load().cabinet_pwconv_supported(Ci, Co, H * W) == 1
    g0 = torch.Generator().manual_seed(Ci * 1000 + Co)
    conv = torch.nn.Conv2d(Ci, Co, 1, bias=False)
    x = torch.randn(B, Ci, H, W, generator=g0)
    g = torch.randn(B, Co, H, W, generator=g0)
    xo, wo = x.double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
    yo = F.conv2d(xo, wo)
    yo.backward(g.double())
    conv = conv.cuda()
    xd = x.cuda().requires_grad_(True)
    y = pwconv(xd, conv)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert_close(y, yo, TOL, "y")
    assert_close(xd.grad, xo.grad, TOL, "dx")
    assert_close(conv.weight.grad, wo.grad, TOL, "dw")


def test_pwconv_dispatch_and_determinism():
    from cabinet_amd import _lib
    from cabinet_amd.functional import pwconv, pwconv_supported

    lib = _lib.load()
    assert lib.cabinet_pwconv_supported(240, 80, 4096) == 0  # wide layers stay with MIOpen
    assert lib.cabinet_pwconv_supported(20, 16, 4096) == 0   # channels not a multiple of 8
    x = torch.randn(1, 16, 256, 256, device="cuda", requires_grad=True)
    assert pwconv_supported(torch.nn.Conv2d(16, 64, 1, bias=False), x)
    assert not pwconv_supported(torch.nn.Conv2d(16, 64, 1, bias=True), x)
    assert not pwconv_supported(torch.nn.Conv2d(16, 64, 1, bias=False), x[:, :, :128, :128])  # small plane: MIOpen
    x = torch.randn(2, 16, 64, 64, device="cuda", requires_grad=True)
    conv = torch.nn.Conv2d(16, 64, 1, bias=False).cuda()
    g = torch.randn(2, 64, 64, 64, device="cuda")
    runs = []
    for _ in range(2):
        conv.zero_grad()
        x.grad = None
        pwconv(x, conv).backward(g)
        runs.append((x.grad.clone(), conv.weight.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
