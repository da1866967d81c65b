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
    (1, 8, 4, 64, 32, 512, 0.05, 0.7, 32 * 512 // 16),            # x8 with whole source rows per wave: one interval per lane
    (2, 19, 3, 128, 24, 1024, 0.1, 0.7, 2 * 24 * 1024 // 16),     # ... two intervals per lane (the model's 1024-wide rows), 19 classes
    (1, 5, 2, 256, 16, 2048, 0.0, 0.7, 16 * 2048 // 16),          # ... four intervals per lane, 5 classes (waves 1-3 get one class or none)
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


@pytest.mark.parametrize("paired", [False, True])
def test_fused_ohem_with_a_200_nat_spike_in_one_low_res_column(paired):
    """the x8 forward shifts the log-sum-exp of an interval's eight pixels by ONE bound (maximum over classes and the
    three taps).  A +200 logit in one low-resolution column puts that bound ~288 (log2 units) above every logit of pixels whose own
    convex combination gives the spike almost no weight... and zero weight for the pixels of the NEIGHBOURING interval's far half:
    their shifted sum underflows.  F.cross_entropy is exact for any finite logits: per-pixel loss, loss and gradient must match the
    fp64 oracle, and no pixel may drop out as -inf."""
    from cabinet_amd.functional import ohem_up_fwd_hip
    from cabinet_amd.loss import OhemCELoss, ohem_upsampled_pair
    from oracle.model_ref import ohem_ce

    B, C, Hl, Wl, H, W = 1, 8, 4, 64, 32, 512
    g = torch.Generator().manual_seed(77)
    low = torch.randn(B, C, Hl, Wl, generator=g)
    low[0, 3, 1, 20] += 200.0          # one spike
    low[0, 5, 2, 40:42] -= 150.0       # and a deep trough pair (every OTHER class towers above it)
    lab = torch.randint(0, C, (B, H, W), generator=g)
    n_min = H * W // 16
    ref_in = low.double().requires_grad_(True)
    up = F.interpolate(ref_in, size=(H, W), mode="bilinear", align_corners=False)
    ref = ohem_ce(up, lab, 0.7, n_min)
    ref.backward()
    px_ref = F.cross_entropy(up.detach(), lab, reduction="none")
    x = low.cuda().requires_grad_(True)
    loss_px = ohem_up_fwd_hip(x.detach(), lab.cuda(), (H, W), 0.7, 255)[0]
    assert bool(torch.isfinite(loss_px).all())
    assert_close(loss_px.view(B, H, W), px_ref, 1e-5, "per-pixel loss", atol=1e-5)
    crit = OhemCELoss(0.7, n_min, 255).cuda()
    if paired:
        x2 = low.cuda().requires_grad_(True)
        loss = ohem_upsampled_pair(crit, x, OhemCELoss(0.7, n_min, 255).cuda(), x2, lab.cuda(), (H, W))
        want = 2 * float(ref)
    else:
        loss = crit.forward_upsampled(x, lab.cuda(), (H, W))
        want = float(ref)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - want) <= 1e-5 * abs(want)
    assert_close(x.grad, ref_in.grad, TOL, "dlogits_low", atol=1e-9)


def test_row_kernel_equals_segment_kernel():
    """The round-4 x pass (whole source rows per wave, resize adjoint in registers) against the round-3 segment kernel it
    replaces for W == 8 Wl, Wl % 64 == 0 -- same forward state, both heads, ignored pixels, a border-heavy narrow case."""
    import os

    from cabinet_amd import functional as Fn

    for B, C, Hl, Wl in [(2, 8, 8, 128), (1, 19, 4, 64), (1, 3, 2, 256)]:
        H, W = 8 * Hl, 8 * Wl
        g = torch.Generator().manual_seed(C + Wl)
        la = (torch.randn(B, C, Hl, Wl, generator=g) * 2).cuda()
        lb_ = (torch.randn(B, C, Hl, Wl, generator=g) * 2).cuda()
        lab = torch.randint(0, C, (B, H, W), generator=g)
        lab[torch.rand(B, H, W, generator=g) < 0.1] = 255
        lab = lab.cuda()
        loss_px, _ = Fn.ohem_up_pair_fwd_hip(la, lb_, lab, (H, W), 0.7, 255)
        new = Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37)
        os.environ["CABINET_OHEM_SEGMENT_KERNEL"] = "1"
        try:
            old = Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37)
        finally:
            del os.environ["CABINET_OHEM_SEGMENT_KERNEL"]
        torch.cuda.synchronize()
        assert_close(new, old, 2e-6, f"dlow {B}x{C}x{Hl}x{Wl}", atol=1e-9)
        assert torch.equal(new, Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37))  # bit-reproducible


def test_band_kernel_equals_row_kernel_plus_y_pass():
    """Round 5: the backward WITHOUT the intermediate T (bands of eight output rows keep the y adjoint in registers, two addends per
    source row; opt-in through CABINET_OHEM_BAND=1 -- it measured slower, see ohem.hip) against the default row kernel + y pass: pair and single head, ignored pixels,
    the three lane widths, Hl = 1 (both edge bands touch the only source row), the config-3 and config-5 head grids; the two
    summation orders differ in the last bit, nothing more; bit-reproducible."""
    import os

    from cabinet_amd import functional as Fn

    for B, C, Hl, Wl in [(2, 8, 8, 128), (1, 19, 4, 64), (1, 3, 2, 256), (2, 8, 1, 64), (8, 8, 128, 128), (2, 19, 256, 128)]:
        H, W = 8 * Hl, 8 * Wl
        g = torch.Generator().manual_seed(C + Wl + Hl)
        la = (torch.randn(B, C, Hl, Wl, generator=g) * 2).cuda()
        lb_ = (torch.randn(B, C, Hl, Wl, generator=g) * 2).cuda()
        lab = torch.randint(0, C, (B, H, W), generator=g)
        lab[torch.rand(B, H, W, generator=g) < 0.1] = 255
        lab = lab.cuda()
        loss_px, _ = Fn.ohem_up_pair_fwd_hip(la, lb_, lab, (H, W), 0.7, 255)
        old = Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37)   # default: row kernel + y pass
        os.environ["CABINET_OHEM_BAND"] = "1"
        try:
            new = Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37)
            again = Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37)
            single = Fn.ohem_up_bwd_hip(la, lab, loss_px[0], (H, W), 0.7, 255, 0.37)
        finally:
            del os.environ["CABINET_OHEM_BAND"]
        torch.cuda.synchronize()
        assert_close(new, old, 1e-6, f"dlow {B}x{C}x{Hl}x{Wl}", atol=1e-9)
        assert torch.equal(new, again)            # bit-reproducible
        assert torch.equal(single, new[0])        # one head: the same bits


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


@pytest.mark.parametrize("B,C,Hl,Wl,H,W,ignore_frac", [
    (2, 8, 16, 16, 128, 128, 0.0),      # x8: both heads in one workgroup (shared label tile)
    (8, 8, 128, 128, 1024, 1024, 0.0),  # BASELINE config 3
    (2, 19, 32, 16, 256, 128, 0.1),     # 19 classes, x8, ignored pixels
    (1, 8, 9, 7, 61, 50, 0.1),          # general ratio: one launch per head behind the same entry point
    (1, 19, 16, 136, 128, 1088, 0.1),   # three column segments in the backward
])
def test_paired_heads_equal_two_single_heads_bitwise(B, C, Hl, Wl, H, W, ignore_frac):
    """cabinet_ohem_up_pair_fwd/bwd (both loss heads of reference train.py:435 per launch) against the single-head entry
    points on the same inputs: per-pixel losses, reduced statistics and gradients are bit-identical (the pairing changes
    which workgroup computes a head, not its arithmetic), and the autograd pair equals the sum of two single heads."""
    from cabinet_amd import functional as Fn
    from cabinet_amd.loss import OhemCELoss, ohem_upsampled_pair

    g = torch.Generator().manual_seed(H + C)
    la, lb_ = ((torch.randn(B, C, Hl, Wl, generator=g) * 2.0).cuda() for _ in range(2))
    lab = torch.randint(0, C, (B, H, W), generator=g)
    lab[torch.rand(B, H, W, generator=g) < ignore_frac] = 255
    lab = lab.cuda()
    loss_px, stats = Fn.ohem_up_pair_fwd_hip(la, lb_, lab, (H, W), 0.7, 255)
    singles = [Fn.ohem_up_fwd_hip(x, lab, (H, W), 0.7, 255) for x in (la, lb_)]
    for i, (lp, st) in enumerate(singles):
        assert torch.equal(loss_px[i], lp) and torch.equal(stats[i], st)
    dl = Fn.ohem_up_pair_bwd_hip(la, lb_, lab, loss_px, (H, W), 0.7, 255, 0.37)
    for i, x in enumerate((la, lb_)):
        assert torch.equal(dl[i], Fn.ohem_up_bwd_hip(x, lab, loss_px[i], (H, W), 0.7, 255, 0.37))
    n_min = B * H * W // 16
    ca, cb = OhemCELoss(0.7, n_min).cuda(), OhemCELoss(0.7, n_min).cuda()
    xa, xb = la.clone().requires_grad_(True), lb_.clone().requires_grad_(True)
    pair = ohem_upsampled_pair(ca, xa, cb, xb, lab, (H, W))
    pair.backward()
    ya, yb = la.clone().requires_grad_(True), lb_.clone().requires_grad_(True)
    two = ca.forward_upsampled(ya, lab, (H, W)) + cb.forward_upsampled(yb, lab, (H, W))
    two.backward()
    assert abs(float(pair) - float(two)) <= 1e-6 * abs(float(two))
    assert torch.equal(xa.grad, ya.grad) and torch.equal(xb.grad, yb.grad)


def test_paired_heads_with_one_head_on_the_rare_branch():
    """One head above the threshold everywhere, the other with nothing above it (top-n_min branch, composite path): the pair
    still equals the two single heads; heads that disagree in threshold are launched separately."""
    from cabinet_amd.loss import OhemCELoss, ohem_upsampled_pair

    g = torch.Generator().manual_seed(3)
    la = (torch.randn(1, 8, 8, 8, generator=g) * 2.0).cuda()
    lab = torch.randint(0, 8, (1, 64, 64), generator=g).cuda()
    lb_ = torch.nn.functional.one_hot(torch.nn.functional.interpolate(lab[None].float(), size=(8, 8))[0].long(), 8)
    lb_ = (lb_.permute(0, 3, 1, 2).float() * 30.0).cuda()  # near-perfect logits: every loss far below thresh
    for thr_b in (0.7, 0.9):
        ca, cb = OhemCELoss(0.7, 64 * 64 // 16).cuda(), OhemCELoss(thr_b, 64 * 64 // 16).cuda()
        xa, xb = la.clone().requires_grad_(True), lb_.clone().requires_grad_(True)
        pair = ohem_upsampled_pair(ca, xa, cb, xb, lab, (64, 64))
        pair.backward()
        ya, yb = la.clone().requires_grad_(True), lb_.clone().requires_grad_(True)
        two = ca.forward_upsampled(ya, lab, (64, 64)) + cb.forward_upsampled(yb, lab, (64, 64))
        two.backward()
        assert abs(float(pair) - float(two)) <= 1e-6 * max(1.0, abs(float(two)))
        assert_close(xa.grad, ya.grad, 1e-6, "head a")
        assert_close(xb.grad, yb.grad, 1e-5, "head b", atol=1e-12)


@pytest.mark.parametrize("nheads,nblk", [(1, 1), (2, 8192), (2, 4097), (3, 300)])
def test_ohem_stats_reduces_the_partials(nheads, nblk):
    """cabinet_ohem_stats (ABI v4): [n_valid, n_above, sum_above] per head from the forward's per-workgroup partials -- counts
    exact (64-bit), the sum in double, a block that met an out-of-range label (count -2^30) keeps n_valid negative, and the
    fixed summation order makes two calls bit-identical."""
    import ctypes

    from cabinet_amd import _lib

    lib = _lib.load()
    gen = torch.Generator().manual_seed(nblk)
    blk_sum = (torch.rand(nheads, nblk, generator=gen) * 1e3).cuda()
    blk_cnt = torch.randint(0, 1025, (nheads, nblk, 2), generator=gen, dtype=torch.int32).cuda()
    if nblk > 1:
        blk_cnt[0, nblk // 2, 0] = -(1 << 30)
    outs = []
    for _ in range(2):
        stats = torch.empty((nheads, 3), dtype=torch.float64, device="cuda")
        rc = lib.cabinet_ohem_stats(ctypes.c_void_p(blk_sum.data_ptr()), ctypes.c_void_p(blk_cnt.data_ptr()), nheads, nblk,
                                    ctypes.c_void_p(stats.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "cabinet_ohem_stats")
        outs.append(stats)
    torch.cuda.synchronize()
    ref_cnt = blk_cnt.long().sum(dim=1).double()
    ref_sum = blk_sum.double().sum(dim=1)
    assert torch.equal(outs[0][:, :2], ref_cnt)
    assert torch.allclose(outs[0][:, 2], ref_sum, rtol=1e-13, atol=0.0)
    assert torch.equal(outs[0], outs[1])
    if nblk > 1:
        assert float(outs[0][0, 0]) < 0
