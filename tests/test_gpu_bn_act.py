"""HIP BatchNorm2d + activation (K7, through the C ABI) vs the oracle (fp64 F.batch_norm + activation, the
restatement used by oracle/model_ref.py::_bn, _hswish) -- ragged planes, planes longer than one chunk, all three
activations, both BatchNorm modes, running-buffer side effects."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32


def _oracle(x, g, bn, act, training):
    from oracle.model_ref import _hswish

    xo = x.detach().cpu().double().requires_grad_(True)
    w = bn.weight.detach().cpu().double().requires_grad_(True)
    b = bn.bias.detach().cpu().double().requires_grad_(True)
    rm, rv = bn.running_mean.detach().cpu().double().clone(), bn.running_var.detach().cpu().double().clone()
    u = F.batch_norm(xo, rm, rv, w, b, training, 0.1, 1e-5)
    y = {"relu": F.relu, "hardswish": _hswish, None: lambda t: t}[act](u)
    y.backward(g.cpu().double())
    return y.detach(), xo.grad, w.grad, b.grad, rm, rv


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act", [None, "relu", "hardswish"])
@pytest.mark.parametrize("shape", [(4, 16, 128, 128), (2, 5, 7, 9), (1, 3, 100, 100), (3, 8, 1, 1), (2, 24, 90, 91)])
def test_bn_act_vs_oracle(shape, act, training):
    from cabinet_amd.functional import bn_act

    g0 = torch.Generator().manual_seed(4)
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g0) + 0.5)
        bn.bias.copy_(torch.rand(C, generator=g0) - 0.5)
        bn.running_mean.copy_(torch.rand(C, generator=g0) - 0.5)
        bn.running_var.copy_(torch.rand(C, generator=g0) + 0.5)
    x = torch.randn(*shape, generator=g0) * 1.7 + 0.6  # non-zero mean: exercises the variance computation
    g = torch.randn(*shape, generator=g0)
    if shape[0] * shape[2] * shape[3] == 1 and training:
        pytest.skip("batch statistics of a single value are undefined")
    ref = _oracle(x, g, bn, act, training)
    bn = bn.cuda().train(training)
    xd = x.cuda().requires_grad_(True)
    y = bn_act(xd, bn, act)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert_close(y, ref[0], TOL, "y")
    assert_close(xd.grad, ref[1], TOL, "dx")
    assert_close(bn.weight.grad, ref[2], TOL, "dweight")
    assert_close(bn.bias.grad, ref[3], TOL, "dbias")
    assert_close(bn.running_mean, ref[4], 1e-5, "running_mean")
    assert_close(bn.running_var, ref[5], 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == int(training)


def test_bn_act_matches_stock_modules_and_is_deterministic():
    """The fused sequential of the backbone equals the stock nn.Sequential on the same device."""
    from cabinet_amd.models.mobilenetv3 import InvertedResidual

    torch.manual_seed(0)
    blk = InvertedResidual(16, 64, 24, 3, 2, True, True).cuda().train()
    x = torch.randn(2, 16, 40, 40, device="cuda", requires_grad=True)
    g = torch.randn(2, 24, 20, 20, device="cuda")
    import copy

    ref = copy.deepcopy(blk)
    out = blk(x)
    out.backward(g)
    gx = x.grad.clone()
    x.grad = None
    out_ref = torch.nn.Sequential.forward(ref.conv, x)  # stock ATen / MIOpen path
    out_ref.backward(g)
    assert_close(out, out_ref, 1e-4, "block output")
    assert_close(gx, x.grad, 1e-3, "block dx")
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert_close(p.grad, q.grad, 2e-3, k, atol=1e-5)
    for (k, p), (_, q) in zip(blk.named_buffers(), ref.named_buffers()):
        assert_close(p.double(), q.double(), 1e-5, k)
    bn = torch.nn.BatchNorm2d(16).cuda().train()
    from cabinet_amd.functional import bn_act

    runs = []
    for _ in range(2):
        bn.zero_grad()
        x.grad = None
        bn_act(x, bn, "hardswish").backward(torch.ones(2, 16, 40, 40, device="cuda"))
        runs.append((x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*runs))  # fixed reduction order, no atomics


@pytest.mark.parametrize("act", [None, "relu", "hardswish"])
@pytest.mark.parametrize("shape", [(2, 12, 16, 16), (1, 5, 7, 9), (2, 3, 100, 100), (3, 8, 1, 1)])
def test_gate_act_vs_oracle(shape, act):
    """act(x * gate) -- SELayer's product (mobilenetv3.py:79-83) + the activation behind it -- vs fp64 autograd."""
    from cabinet_amd.functional import gate_act
    from oracle.model_ref import _hswish

    g0 = torch.Generator().manual_seed(8)
    x = torch.randn(*shape, generator=g0) * 2
    gate = torch.rand(shape[0], shape[1], generator=g0)
    g = torch.randn(*shape, generator=g0)
    xo, go = x.double().requires_grad_(True), gate.double().requires_grad_(True)
    yo = {"relu": F.relu, "hardswish": _hswish, None: lambda t: t}[act](xo * go[:, :, None, None])
    yo.backward(g.double())
    xd, gd = x.cuda().requires_grad_(True), gate.cuda().requires_grad_(True)
    y = gate_act(xd, gd, act)
    y.backward(g.cuda())
    assert_close(y, yo, TOL, "y")
    assert_close(xd.grad, xo.grad, TOL, "dx")
    assert_close(gd.grad, go.grad, TOL, "dgate")


def test_batched_bn_counters_match_per_module_increments():
    from cabinet_amd.functional import batched_bn_counters, bn_act

    bns = [torch.nn.BatchNorm2d(4).cuda().train() for _ in range(3)]
    x = torch.randn(2, 4, 5, 5, device="cuda")
    with batched_bn_counters():
        for bn in bns:
            bn_act(x, bn, "relu")
        bn_act(x, bns[0], None)
        assert int(bns[0].num_batches_tracked) == 0  # deferred
    assert [int(b.num_batches_tracked) for b in bns] == [2, 1, 1]
    bn_act(x, bns[1], None)  # outside the context: immediate
    assert int(bns[1].num_batches_tracked) == 2


@pytest.mark.parametrize("training", [True, False])
def test_bn_act_with_residual_shortcut(training):
    """y = bn(x) + r in one pass (the MBConv identity shortcut, mobilenetv3.py:158) vs fp64 autograd; dr == dy."""
    from cabinet_amd.functional import bn_act

    g0 = torch.Generator().manual_seed(6)
    shape = (2, 24, 33, 31)
    bn = torch.nn.BatchNorm2d(24)
    x, r, g = (torch.randn(*shape, generator=g0) for _ in range(3))
    xo, ro = x.double().requires_grad_(True), r.double().requires_grad_(True)
    w, b = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    yo = F.batch_norm(xo, bn.running_mean.double().clone(), bn.running_var.double().clone(), w, b, training, 0.1, 1e-5) + ro
    yo.backward(g.double())
    bn = bn.cuda().train(training)
    xd, rd = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
    y = bn_act(xd, bn, None, rd)
    y.backward(g.cuda())
    assert_close(y, yo, TOL, "y")
    assert_close(xd.grad, xo.grad, TOL, "dx")
    assert_close(rd.grad, ro.grad, 0.0, "dresidual", atol=0)
    assert_close(bn.weight.grad, w.grad, TOL, "dweight")
