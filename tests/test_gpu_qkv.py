"""HIP q/k/v producers (K6) and the 1x1 output projection (through the C ABI) vs the oracle.

The golden CAB vectors (g2_cab.npz, from the reference) cover the same kernels end to end through
ContextAggregationBlock (test_gpu_model.py::test_cab_module_golden).  Here the producer op is checked on its
own against oracle/model_ref.py (_bn, _psp: the restatement of cab.py:107-123,46-76) in fp64, over shapes where
the pooling bins overlap (H % s != 0), where there are more bins than pixels (s > H), non-square maps, generic
channel counts, and both BatchNorm modes.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32


def _make(C, Kc, Vc, sizes, seed):
    from cabinet_amd.models.cab import GlobalContextAttention

    torch.manual_seed(seed)
    m = GlobalContextAttention(C, Kc, Vc, C, scale=1, psp_sizes=sizes)
    with torch.no_grad():
        torch.nn.init.kaiming_normal_(m.project_out.weight)
        for bn in (m.to_query[1], m.to_key[1]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    return m


def _oracle_qkv(m, x, grads, training, sizes):
    from oracle.model_ref import Weights, _bn, _psp

    w = Weights({k: v.clone() for k, v in m.state_dict().items()}, dtype=torch.float64)
    xo = x.detach().cpu().double().requires_grad_(True)
    b, _, h, wd = xo.shape
    q = F.relu(_bn(w, F.conv2d(xo, w["to_query.0.weight"]), "to_query.1", training))
    k = F.relu(_bn(w, F.conv2d(xo, w["to_key.0.weight"]), "to_key.1", training))
    k = _psp(w, k, "psp_key", sizes)
    v = _psp(w, F.conv2d(xo, w["to_value.weight"]), "psp_value", sizes)
    q, k, v = (t.reshape(b, -1, h * wd) for t in (q, k, v))
    torch.autograd.backward([q, k, v], [g.cpu().double() for g in grads])
    return (q.detach(), k.detach(), v.detach()), xo.grad, w.grads(), w.buffers()


def test_qkv_output_stage_lds_attribute_covers_a_later_larger_pyramid():
    """The one-kernel output stage keeps (ns + 1) weight blocks and the pyramid terms in dynamic LDS; the per-device
    MaxDynamicSharedMemorySize attribute is set once, so a small pyramid first must not pin it (Kc = 128: 34 KB for
    sizes (1,), 101 KB for (1, 3, 6, 8), which crosses the 64 KB default).  First test of the file, so that run on its own
    the small pyramid is the first call of the process."""
    _run_qkv_case(1, 64, 128, 128, 16, 16, (1,), True)
    _run_qkv_case(1, 64, 128, 128, 16, 16, (1, 3, 6, 8), True)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,C,Kc,Vc,H,W,sizes", [
    (2, 256, 128, 128, 16, 16, (1, 3, 6, 8)),   # golden-fixture shape
    (8, 256, 128, 128, 32, 32, (1, 3, 6, 8)),   # config 3
    (1, 64, 32, 48, 5, 7, (1, 3, 6, 8)),        # more bins than pixels, ragged, Kc != Vc
    (2, 32, 16, 16, 8, 20, (2, 5)),             # overlapping bins, non-square, two sizes
    (1, 512, 256, 128, 32, 64, (1, 3, 6, 8)),   # the reference's own test block (test_models.py:49): n = 2048
    (2, 256, 128, 128, 64, 32, (1, 3, 6, 8)),   # config 5's map (H != W): one-kernel output stage, Kc = 128
    (3, 96, 64, 64, 16, 32, (2, 5)),            # one-kernel output stage at Kc = 64, two sizes, odd batch
    (1, 320, 256, 256, 16, 16, (1, 3, 6, 8)),   # ... and at Kc = 256 (pyramid operand fetched in chunks)
])
def test_qkv_vs_oracle(B, C, Kc, Vc, H, W, sizes, training):
    _run_qkv_case(B, C, Kc, Vc, H, W, sizes, training)


def _run_qkv_case(B, C, Kc, Vc, H, W, sizes, training):
    from cabinet_amd.functional import cab_qkv

    m = _make(C, Kc, Vc, sizes, 5).cuda()
    m.train(training)
    ref_m = _make(C, Kc, Vc, sizes, 5)
    g0 = torch.Generator().manual_seed(9)
    x = torch.randn(B, C, H, W, generator=g0).cuda().requires_grad_(True)
    grads = [torch.randn(B, ch, H * W, generator=g0) for ch in (Kc, Kc, Vc)]
    q, k, v = cab_qkv(x, m)
    torch.autograd.backward([q, k, v], [g.cuda() for g in grads])
    torch.cuda.synchronize()
    (oq, ok, ov), o_dx, o_grads, o_buf = _oracle_qkv(ref_m, x, grads, training, sizes)
    assert_close(q, oq, TOL, "q")
    assert_close(k, ok, TOL, "k")
    assert_close(v, ov, TOL, "v")
    assert_close(x.grad, o_dx, TOL, "dx")
    for name, p in m.named_parameters():
        if name.startswith("project_out"):
            continue
        assert_close(p.grad, o_grads[name], TOL, f"grad {name}")
    for name, b in m.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(o_buf[name]), name
        else:
            assert_close(b, o_buf[name], 1e-5, name)


@pytest.mark.parametrize("B,Ci,Co,shape", [(2, 128, 256, (16, 16)), (8, 128, 256, (32, 32)), (1, 48, 16, (5, 7)),
                                           (3, 16, 32, (1, 1)), (2, 24, 72, (16, 16)), (1, 72, 40, (9, 7)),
                                           (2, 200, 80, (8, 8))])  # last three: channel counts not multiples of 16
def test_conv1x1_vs_oracle(B, Ci, Co, shape):
    from cabinet_amd.functional import conv1x1

    g0 = torch.Generator().manual_seed(2)
    x = torch.randn(B, Ci, *shape, generator=g0)
    w = torch.randn(Co, Ci, 1, 1, generator=g0)
    g = torch.randn(B, Co, *shape, generator=g0)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = conv1x1(xd, wd)
    y.backward(g.cuda())
    xo, wo = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yo = F.conv2d(xo, wo)  # reference cab.py:155
    yo.backward(g.double())
    assert_close(y, yo, TOL, "y")
    assert_close(xd.grad, xo.grad, TOL, "dx")
    assert_close(wd.grad, wo.grad, TOL, "dw")


@pytest.mark.parametrize("B,Ci,Co,shape", [(8, 256, 256, (32, 32)), (2, 256, 256, (64, 32)), (1, 48, 16, (5, 7)), (2, 24, 72, (16, 16))])
def test_conv1x1_with_bias_vs_fp64(B, Ci, Co, shape):
    """`AttentionBranch.convb` (reference cabinet.py:65-66, :86: nn.Conv2d(256, 256, 1, bias=True) on the CAB's output): the small-grid MFMA
    product with the bias in its epilogue, dx / dw by cabinet_conv1x1_bwd, dbias by cabinet_channel_sum -- against F.conv2d in fp64;
    the first two shapes are the config-3 and config-5 grids; bit-reproducible."""
    import torch.nn as nn

    from cabinet_amd.functional import conv1x1, conv1x1_bias_supported

    g0 = torch.Generator().manual_seed(6)
    conv = nn.Conv2d(Ci, Co, 1, bias=True)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(Co, generator=g0))
    x = torch.randn(B, Ci, *shape, generator=g0)
    g = torch.randn(B, Co, *shape, generator=g0)
    xo = x.double().requires_grad_(True)
    wo, bo = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    yo = F.conv2d(xo, wo, bo)
    yo.backward(g.double())
    conv = conv.cuda()
    xd = x.cuda().requires_grad_(True)
    assert conv1x1_bias_supported(xd, conv)
    y = conv1x1(xd, conv.weight, conv.bias)
    y.backward(g.cuda())
    assert_close(y, yo, 2e-5, "y")
    assert_close(xd.grad, xo.grad, 2e-5, "dx")
    assert_close(conv.weight.grad, wo.grad, 2e-5, "dw")
    assert_close(conv.bias.grad, bo.grad, 2e-5, "dbias")
    dx1, dw1, db1 = xd.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()
    xd.grad = conv.weight.grad = conv.bias.grad = None
    y2 = conv1x1(xd, conv.weight, conv.bias)
    y2.backward(g.cuda())
    assert torch.equal(y2, y) and torch.equal(xd.grad, dx1) and torch.equal(conv.weight.grad, dw1) and torch.equal(conv.bias.grad, db1)
    big = torch.randn(8, 256, 128, 128, device="cuda")   # a large plane: the bias form does not exist there, the model keeps the stock operator
    assert not conv1x1_bias_supported(big, nn.Conv2d(256, 256, 1).cuda())


def test_global_branch_runs_native_kernels_and_is_deterministic():
    from cabinet_amd import _lib

    m = _make(256, 128, 128, (1, 3, 6, 8), 1).cuda().train()
    x = torch.randn(2, 256, 16, 16, device="cuda", requires_grad=True)
    g = torch.randn_like(x)
    runs = []
    for _ in range(2):
        m.zero_grad()
        x.grad = None
        m(x).backward(g)
        runs.append([x.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    for a, b in zip(*runs):
        assert torch.equal(a, b)  # no atomics on the whole global branch
    lib = _lib.load()
    sizes = (torch.tensor([1, 3, 6, 8], dtype=torch.int32)).numpy()
    assert lib.cabinet_cab_qkv_supported(8, 256, 128, 128, 32, 32, 4, sizes.ctypes.data) == 1
    assert lib.cabinet_cab_qkv_supported(8, 250, 128, 128, 32, 32, 4, sizes.ctypes.data) == 0  # C % 16 != 0
