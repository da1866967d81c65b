"""API / shape contract of the nn.Module mirror, restating the reference's own tests
(reference tests/unit/test_models.py:38-54,69-161,177-208, tests/unit/test_loss.py:12-88,
tests/integration/test_training_pipeline.py:23-55,359-372) against cabinet_amd.models.
Runs on CPU tensors (host path of the modules)."""
import copy
import io

import pytest
import torch

from cabinet_amd.models.constants import MOBILENETV3_CFGS
from src.models.cab import ContextAggregationBlock, PSPModule
from src.models.cabinet import AttentionBranch, CABiNet, ConvBNReLU, FeatureFusionModule, SpatialBranch
from src.models.constants import MODEL_CONFIG
from src.models.mobilenetv3 import InvertedResidual, MobileNetV3
from src.utils.loss import OhemCELoss


def _small(num_classes=19, mode="small"):
    return CABiNet(n_classes=num_classes, cfgs=MOBILENETV3_CFGS[mode], mode=mode)


def test_psp_module_forward():
    out = PSPModule(sizes=(1, 2, 4), in_channels=256)(torch.randn(2, 256, 32, 32))
    assert out.shape == (2, 256, 32, 32) and not torch.isnan(out).any()


def test_context_aggregation_block_forward():
    x = torch.randn(2, 512, 16, 24)
    out = ContextAggregationBlock(512, 128)(x)
    assert out.shape == x.shape and not torch.isnan(out).any()


def test_conv_bn_relu_forward():
    out = ConvBNReLU(in_chan=64, out_chan=128, kernel_size=3, stride=2)(torch.randn(2, 64, 64, 64))
    assert out.shape == (2, 128, 32, 32)


def test_attention_branch_forward():
    low, high = AttentionBranch(inplanes=960, interplanes=256, outplanes=256, num_classes=19)(
        torch.randn(2, 960, 16, 16))
    assert low.shape == (2, 256, 16, 16) and high.shape == (2, 19, 16, 16)
    assert not torch.isnan(low).any() and not torch.isnan(high).any()


def test_spatial_branch_forward():
    assert SpatialBranch()(torch.randn(2, 3, 256, 256)).shape == (2, 128, 32, 32)


def test_ffm_forward_host_path():
    out = FeatureFusionModule(384, 256)(torch.randn(2, 128, 16, 16), torch.randn(2, 256, 16, 16))
    assert out.shape == (2, 256, 16, 16)


@pytest.mark.parametrize("mode", ["large", "small"])
def test_cabinet_forward_shape(mode):
    model = _small(19, mode).eval()
    with torch.no_grad():
        out, out16 = model(torch.randn(2, 3, 256, 256))
    assert out.shape == (2, 19, 256, 256) and out16.shape == (2, 19, 256, 256)
    assert not torch.isnan(out).any() and not torch.isnan(out16).any()
    assert model.attention_planes == MODEL_CONFIG[mode]["attention_planes"]


def test_cabinet_get_params_groups():
    model = _small()
    wd, nowd, lr_wd, lr_nowd = model.get_params()
    assert all(len(g) > 0 for g in (wd, nowd, lr_wd, lr_nowd))
    decoder = {id(p) for p in lr_wd + lr_nowd}
    for name, child in model.named_children():
        if name in ("ab", "ffm", "conv_out"):
            assert all(id(p) in decoder for p in child.parameters()), name
    assert not ({id(p) for p in wd + nowd} & decoder)
    # every parameter exactly once
    allp = wd + nowd + lr_wd + lr_nowd
    assert len(allp) == len({id(p) for p in allp}) == len(list(model.parameters()))
    # CAB.gamma is a no-decay x10 parameter; conv weights of the hot path decay (reference cabinet.py:262-280)
    assert any(p is model.ab.a2block.gamma for p in lr_nowd)
    assert any(p is model.ffm.convblk.conv.weight for p in lr_wd)
    assert any(p is model.ab.a2block.global_attn.project_out.weight for p in lr_wd)


def test_state_dict_keys_of_hot_path():
    sd = _small().state_dict()
    want = {
        "ab.a2block.gamma": (1,),
        "ab.a2block.global_attn.to_query.0.weight": (128, 256, 1, 1),
        "ab.a2block.global_attn.to_query.1.running_var": (128,),
        "ab.a2block.global_attn.to_key.1.num_batches_tracked": (),
        "ab.a2block.global_attn.to_value.weight": (128, 256, 1, 1),
        "ab.a2block.global_attn.psp_key.project.weight": (128, 640, 1, 1),
        "ab.a2block.global_attn.psp_value.project.weight": (128, 640, 1, 1),
        "ab.a2block.global_attn.project_out.weight": (256, 128, 1, 1),
        "ab.a2block.local_attn.refine.2.block.0.weight": (256, 1, 3, 3),
        "ab.a2block.local_attn.refine.0.block.1.bias": (256,),
        "ffm.convblk.conv.weight": (256, 384, 1, 1),
        "ffm.convblk.bn.running_mean": (256,),
        "ffm.convblk.bn.num_batches_tracked": (),
        "ffm.conv1.weight": (64, 256, 1, 1),
        "ffm.conv2.weight": (256, 64, 1, 1),
    }
    for k, shape in want.items():
        assert k in sd and tuple(sd[k].shape) == shape, k


def test_standalone_cab_is_zero_init_like_reference():
    cab = ContextAggregationBlock(256, 128)
    assert float(cab.gamma) == 0.0 and float(cab.global_attn.project_out.weight.abs().sum()) == 0.0
    net = _small()
    assert float(net.ab.a2block.global_attn.project_out.weight.abs().sum()) > 0  # re-initialised by init_weight


def test_invalid_arguments_raise_value_error():
    with pytest.raises(ValueError, match="mode must be"):
        MobileNetV3(cfgs=[[3, 1, 16, 1, 0, 2]], mode="xlarge")
    with pytest.raises(ValueError, match="stride must be"):
        InvertedResidual(inp=16, hidden_dim=16, oup=16, kernel_size=3, stride=3, use_se=False, use_hs=False)
    with pytest.raises(ValueError):
        CABiNet(n_classes=8, cfgs=MOBILENETV3_CFGS["small"], mode="huge")
    for mode in ("large", "small"):
        MobileNetV3(cfgs=[[3, 1, 16, 1, 0, 2]], mode=mode)


def test_deepcopy_checkpoint_roundtrip_and_eval_determinism():
    net = _small(8)
    clone = copy.deepcopy(net).eval()  # what ModelEMA does (reference ema.py:44)
    buf = io.BytesIO()
    torch.save(net.state_dict(), buf)
    buf.seek(0)
    fresh = _small(8)
    fresh.load_state_dict(torch.load(buf, weights_only=True))
    x = torch.randn(1, 3, 128, 128)
    with torch.no_grad():
        a, b = clone(x)[0], fresh.eval()(x)[0]
        c = fresh(x)[0]
    assert torch.allclose(a, b, atol=1e-6) and torch.allclose(b, c, atol=1e-6)


def test_single_training_step_finite():
    torch.manual_seed(0)
    net = _small(19).train()
    crit = OhemCELoss(thresh=0.7, n_min=64 * 64 // 16)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    x, y = torch.randn(2, 3, 64, 64), torch.randint(0, 19, (2, 64, 64))
    out, out16 = net(x)
    loss = crit(out, y) + crit(out16, y)
    loss.backward()
    opt.step()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


class TestOhemCELoss:
    def test_forward_scalar_nonneg(self):
        loss = OhemCELoss(thresh=0.7, n_min=100, ignore_lb=255)(torch.randn(4, 19, 32, 32),
                                                                 torch.randint(0, 19, (4, 32, 32)))
        assert loss.ndim == 0 and loss.item() >= 0 and not torch.isnan(loss)

    def test_all_ignored_is_zero_with_grad(self):
        loss = OhemCELoss(thresh=0.7, n_min=100, ignore_lb=255)(torch.randn(1, 19, 16, 16),
                                                                 torch.full((1, 16, 16), 255))
        assert loss.item() == 0.0 and loss.requires_grad

    def test_partial_ignore_and_backward(self):
        logits = torch.randn(2, 19, 16, 16, requires_grad=True)
        labels = torch.randint(0, 19, (2, 16, 16))
        labels[0, :5, :5] = 255
        loss = OhemCELoss(thresh=0.7, n_min=50)(logits, labels)
        loss.backward()
        assert torch.isfinite(logits.grad).all()

    def test_host_tensors_take_the_composite_path_of_the_fused_head(self):
        """forward_upsampled / ohem_upsampled_pair on CPU tensors == interpolate + forward (reference cabinet.py:240-245
        followed by loss.py:38-80); TrainStep(fused_loss=True) on the host therefore equals the reference recipe."""
        from cabinet_amd.loss import ohem_upsampled_pair

        torch.manual_seed(3)
        low_a = torch.randn(2, 5, 8, 8, requires_grad=True)
        low_b = torch.randn(2, 5, 8, 8, requires_grad=True)
        labels = torch.randint(0, 5, (2, 32, 32))
        crit_a, crit_b = OhemCELoss(0.7, 64, 255), OhemCELoss(0.7, 64, 255)
        up = lambda t: torch.nn.functional.interpolate(t, size=(32, 32), mode="bilinear", align_corners=False)  # noqa: E731
        want = crit_a(up(low_a), labels) + crit_b(up(low_b), labels)
        got = ohem_upsampled_pair(crit_a, low_a, crit_b, low_b, labels, (32, 32))
        assert torch.allclose(got, want, atol=1e-6)
        single = crit_a.forward_upsampled(low_a, labels, (32, 32)) + crit_b.forward_upsampled(low_b, labels)
        assert torch.allclose(single, want, atol=1e-6)
        got.backward()
        assert low_a.grad is not None and torch.isfinite(low_a.grad).all()

    def test_class_weights_and_no_criteria_submodule(self):
        w = torch.ones(19)
        w[0] = 2.0
        crit = OhemCELoss(thresh=0.7, n_min=100, ignore_lb=255, weight=w)
        assert not hasattr(crit, "criteria")
        assert crit(torch.randn(2, 19, 16, 16), torch.randint(0, 19, (2, 16, 16))).item() >= 0

    def test_matches_oracle_restatement(self):
        from oracle.model_ref import ohem_ce

        torch.manual_seed(3)
        logits, labels = torch.randn(2, 8, 24, 24) * 3, torch.randint(0, 8, (2, 24, 24))
        labels[1, :4] = 255
        for thresh in (0.7, 2.5, 9.0):          # 9.0: nothing above thresh -> the top-n_min branch
            for n_min in (10, 400, 5000):       # 5000 > #valid -> clamped
                la = logits.clone().requires_grad_(True)
                lb_ = logits.clone().requires_grad_(True)
                a = OhemCELoss(thresh, n_min)(la, labels)
                b = ohem_ce(lb_, labels, thresh, n_min)
                assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (thresh, n_min, float(a), float(b))
                a.backward()
                b.backward()
                assert torch.allclose(la.grad, lb_.grad, rtol=1e-5, atol=1e-9), (thresh, n_min)


class TestSoftmaxFocalLoss:
    """reference tests/unit/test_loss.py:109-158 restated, plus vectors produced by the reference itself
    (tests/golden/g4_focal.npz, written by tests/golden/make_golden_loss.py)."""

    def test_forward_backward_contract(self):
        from src.utils.loss import SoftmaxFocalLoss

        for gamma in (0.0, 1.0, 2.0, 5.0):
            logits = torch.randn(2, 19, 32, 32, requires_grad=True)
            loss = SoftmaxFocalLoss(gamma=gamma, weight=torch.ones(19), ignore_lb=255)(logits, torch.randint(0, 19, (2, 32, 32)))
            loss.backward()
            assert loss.ndim == 0 and float(loss) >= 0 and torch.isfinite(logits.grad).all()

    def test_matches_reference_vectors(self):
        import os

        import numpy as np

        from conftest import GOLDEN
        from src.utils.loss import SoftmaxFocalLoss

        g = np.load(os.path.join(GOLDEN, "g4_focal.npz"))
        logits, labels, weight = (torch.from_numpy(g[k]) for k in ("logits", "labels", "weight"))
        for gi, gamma in enumerate(g["gammas"]):
            for tag, w in (("plain", None), ("weighted", weight)):
                x = logits.clone().requires_grad_(True)
                loss = SoftmaxFocalLoss(float(gamma), weight=w, ignore_lb=255)(x, labels)
                loss.backward()
                assert abs(float(loss) - float(g[f"{tag}.{gi}.loss"])) <= 1e-6 * max(1.0, abs(float(loss)))
                want = torch.from_numpy(g[f"{tag}.{gi}.dlogits"])
                assert float((x.grad - want).norm()) <= 1e-5 * float(want.norm())
