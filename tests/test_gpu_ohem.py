"""Fused upsample + OHEM-CE kernels (cabinet_ohem_up_fwd/bwd) vs the oracle: OHEM-CE of the
materialised F.interpolate output (reference cabinet.py:240-245 + loss.py:38-80)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.mark.parametrize("B,C,Hl,Wl,H,W,ignore_frac,thresh,n_min", [
    (2, 8, 16, 16, 128, 128, 0.0, 0.7, 2 * 128 * 128 // 16),     # x8, CABiNet's ratio
    (2, 19, 12, 20, 96, 160, 0.2, 0.7, 2 * 96 * 160 // 16),      # 19 classes, non-square, ignored pixels
    (1, 8, 9, 7, 61, 50, 0.1, 0.7, 100),                          # non-integer ratio, odd sizes
    (1, 12, 20, 80, 160, 640, 0.05, 0.7, 160 * 640 // 16),        # two 64-column segments per row, 12 classes
    (1, 27, 6, 6, 48, 48, 0.0, 0.7, 48 * 48 // 16),               # 27 classes (widest class bucket)
    (1, 8, 32, 32, 16, 16, 0.0, 0.7, 16),                         # "upsample" that shrinks (ratio < 1)
    (1, 8, 8, 8, 64, 64, 0.0, 1.5, 64 * 64 // 16),                # higher threshold (fewer selected)
    (1, 8, 8, 8, 64, 64, 0.0, 50.0, 64 * 64 // 16),               # nothing above thresh -> composite top-k branch
    (1, 8, 8, 8, 64, 64, 1.0, 0.7, 10),                           # everything ignored -> 0 with grad
    (1, 8, 16, 24, 64, 96, 0.1, 0.7, 64 * 96 // 16),              # x4: closed-form rows, general column pass
    (2, 5, 10, 6, 20, 48, 0.0, 0.7, 2 * 20 * 48 // 16),           # x2 rows, x8 columns, 5 classes (predicated bucket of 8)
    (1, 19, 16, 136, 128, 1088, 0.1, 0.7, 128 * 1088 // 16),      # x8, three column segments (64 + 64 + 8), 19 classes
    (1, 8, 8, 8, 48, 48, 0.0, 0.7, 48 * 48 // 16),                # x6: an even ratio whose reciprocal is inexact in fp32
    (1, 8, 1, 1, 8, 8, 0.0, 0.7, 4),                              # one source pixel: every weight is 1 (both borders at once)
])
def test_fused_ohem_vs_oracle(B, C, Hl, Wl, H, W, ignore_frac, thresh, n_min):
    from cabinet_amd.loss import OhemCELoss
    from oracle.model_ref import ohem_ce

    g = torch.Generator().manual_seed(H + C)
    low = torch.randn(B, C, Hl, Wl, generator=g) * 2.0
    lab = torch.randint(0, C, (B, H, W), generator=g)
    lab[torch.rand(B, H, W, generator=g) < ignore_frac] = 255
    ref_in = low.double().requires_grad_(True)
    ref = ohem_ce(F.interpolate(ref_in, size=(H, W), mode="bilinear", align_corners=False), lab, thresh, n_min)
    ref.backward()
    crit = OhemCELoss(thresh, n_min, 255).cuda()
    x = low.cuda().requires_grad_(True)
    loss = crit.forward_upsampled(x, lab.cuda(), (H, W))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    if ignore_frac < 1.0:
        assert_close(x.grad, ref_in.grad, TOL, "dlogits_low", atol=1e-9)
    else:
        assert x.grad is None or float(x.grad.abs().sum()) == 0.0


def test_fused_ohem_is_deterministic_and_scales_with_upstream_grad():
    from cabinet_amd.loss import OhemCELoss

    g = torch.Generator().manual_seed(0)
    low = (torch.randn(2, 8, 16, 16, generator=g) * 2).cuda()
    lab = torch.randint(0, 8, (2, 128, 128), generator=g).cuda()
    crit = OhemCELoss(0.7, 2 * 128 * 128 // 16).cuda()
    grads = []
    for scale in (1.0, 1.0, 3.0):
        x = low.clone().requires_grad_(True)
        (crit.forward_upsampled(x, lab) * scale).backward()
        grads.append(x.grad)
    assert torch.equal(grads[0], grads[1])
    assert torch.allclose(grads[2], 3.0 * grads[0], rtol=1e-6, atol=0)


def test_train_step_fused_loss_matches_unfused():
    """Whole step (fwd + 2x OHEM + bwd) with the fused loss vs the materialised-upsample path."""
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

    im, lb = synthetic_batch(2, 256, 256, 8, "cuda", seed=5)
    res = []
    for fused in (False, True):
        net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        step = TrainStep(net, make_criteria(2, 256, 256, "cuda"), fused_loss=fused)
        loss = float(step(im, lb))
        res.append((loss, {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * abs(res[0][0])
    for k, ga in res[0][1].items():
        gb = res[1][1][k]
        err, den = float((ga - gb).norm()), float(ga.norm())
        assert err <= 5e-3 * den + 1e-6 * ga.numel() ** 0.5, (k, err, den)


def test_fused_ohem_rejects_wrong_label_dtype_shape_and_range():
    """The kernels read int64 labels of exactly (B,H,W) in [0,C) or ignore_lb; everything else raises, like
    F.cross_entropy does, instead of reinterpreting bytes or returning logsumexp."""
    from cabinet_amd.loss import OhemCELoss

    crit = OhemCELoss(0.7, 64).cuda()
    low = torch.randn(1, 8, 8, 8, device="cuda")
    lab = torch.randint(0, 8, (1, 64, 64), device="cuda")
    with pytest.raises(RuntimeError, match="int64"):
        crit.forward_upsampled(low, lab.int(), (64, 64))
    with pytest.raises(RuntimeError, match="int64"):
        crit.forward_upsampled(low, lab.to(torch.uint8), (64, 64))
    with pytest.raises(RuntimeError, match="do not match"):
        crit.forward_upsampled(low, lab[:, :32], (64, 64))
    bad = lab.clone()
    bad[0, 5, 7] = 8  # == C, not ignore_lb
    with pytest.raises(RuntimeError, match="out of range"):
        crit.forward_upsampled(low, bad, (64, 64))
    bad[0, 5, 7] = -1
    with pytest.raises(RuntimeError, match="out of range"):
        crit.forward_upsampled(low, bad, (64, 64))
    bad[0, 5, 7] = 255  # ignore_lb is fine
    assert torch.isfinite(crit.forward_upsampled(low, bad, (64, 64)))
