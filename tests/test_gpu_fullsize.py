"""Parity at the sizes BASELINE.json names (north_star: "synthetic 1024x1024 inputs ... logits and grads").

* the full train step (fwd + 2x OHEM-CE + bwd, fused loss -- exactly what bench.py times) for BASELINE config 3
  (Large, 8x3x1024x1024, 8 classes) and config 5 (Large, 2x3x2048x1024, 19 classes) against the functional CPU oracle
  in fp32 AND fp64 (reference cabinet.py:207-247, train.py:429-441): logits and loss at 1e-3, EVERY gradient tensor
  compared, a per-tensor table written to gpurun_out/ (committed copies: profiles/r02_parity_config*.json);
* the same model with BatchNorm in eval mode on populated running statistics (no batch-statistic coupling): every
  gradient tensor at a flat 1e-3;
* every hand-written operator of the timed step at its production grid (the grids the model reaches at config 3 /
  config 5) against its fp64 oracle.

Gradient rule (tests/parity_rules.py, where the measurements behind it are summarised): a tensor passes when it is within
1e-3 of the fp32 reference or of the fp64 oracle; otherwise it must be no further from the fp64 oracle than 3x the fp32
REFERENCE's own distance from fp64 on that very tensor, measured live; otherwise it must be NAMED, with its measured numbers,
in tests/golden/grad_allowlist.json (CAB parameters only: the in-situ tests prove those kernels exact on the model's own
inputs) and is bounded by three single-unit ReLU flips on the CAB grid.  No per-configuration floor, no blanket tolerance; the
per-tensor table of every run -- with the chain of tensors either side of the hot path, the ReLU-mask flip counts and the HIP
CAB run on the fp64 model's own inputs -- is written to gpurun_out/ and committed under profiles/.
"""
import copy

import pytest
import torch

from conftest import assert_close
from insitu import cab_table, instrument, judge_operator_table, operator_table, rel
from parity_rules import TOL, ALLOW_FACTOR, gradient_table, host_memory_gb, judge_gradients, load_allowlist, rel_pair, write_table

TAPS = ("mob", "cab.x", "cab.y", "ab.b1o", "ab.r", "ffm.fsp", "ffm.low", "ffm.y", "head.low", "head16.low")
RELUS = {"ab.conva (CAB input)": "cab.x", "ab.b2 (fusion head)": "ab.r", "ffm.convblk": "ffm.y"}  # post-ReLU taps: mask = value > 0


def _tap_copies(taps):
    """Detached copies of the oracle's tapped tensors and their gradients (the autograd graph can then be freed)."""
    out = {}
    for k, t in taps.items():
        out[k] = t.detach().clone()
        if t.grad is not None:
            out["d." + k] = t.grad.detach().clone()
    return out


def _hip_cab_on(sd, x, g):
    """The HIP CAB alone (K6 -> K1/K2 -> conv1x1 -> K5) on given input / output gradient: -> out, dx, parameter gradients."""
    from cabinet_amd.models.cab import ContextAggregationBlock

    cab = ContextAggregationBlock(256, 128)
    cab.load_state_dict({k[len("ab.a2block."):]: v for k, v in sd.items() if k.startswith("ab.a2block.")})
    cab = cab.cuda().train()
    xd = x.float().cuda().requires_grad_(True)
    y = cab(xd)
    y.backward(g.float().cuda())
    torch.cuda.synchronize()
    return y.detach().cpu(), xd.grad.cpu(), {"ab.a2block." + k: p.grad.cpu() for k, p in cab.named_parameters()}

pytestmark = pytest.mark.gpu


def _full_step(mode, batch, height, width, ncls, tag):
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch
    from oracle import model_ref

    need = 3.8 * batch * height * width / 2 ** 20  # fp64 oracle high-water mark: ~3.6 GB per 1024x1024 image
    have = host_memory_gb()
    if have is not None and have < need + 8:
        pytest.skip(f"fp64 oracle needs ~{need:.0f} GB of host memory, {have:.0f} GB available")
    torch.set_num_threads(model_ref.usable_cpu_threads())
    net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5)
    sd = copy.deepcopy(net.state_dict())
    im, lb = synthetic_batch(batch, height, width, ncls, "cpu", seed=1)

    # the HIP step first (frees the device before the long CPU legs)
    net = net.cuda().train()
    probe = copy.deepcopy(net)
    with torch.no_grad():
        out, out16 = probe(im.cuda())  # materialised full-resolution logits (the step itself never builds them)
    out, out16 = out.cpu(), out16.cpu()
    del probe
    cap = instrument(net)  # what the model feeds the hot path and what comes back, in place (tests/insitu.py)
    step = TrainStep(net, make_criteria(batch, height, width, "cuda"))
    loss = float(step(im.cuda(), lb.cuda()))
    torch.cuda.synchronize()
    cap = {k: v.cpu() for k, v in cap.items()}
    grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    bufs_gpu = {k: v.detach().cpu() for k, v in net.state_dict().items() if "running_" in k}
    net = net.cpu()
    for k, p in net.named_parameters():
        p.grad = grads.get(k)
    torch.cuda.empty_cache()

    w32 = model_ref.Weights(sd)
    taps32 = {}
    out_ref, out16_ref, loss_ref = model_ref.train_step(w32, im, lb, mode, taps=taps32)
    taps32 = _tap_copies(taps32)
    e, d = rel_pair(out, out_ref)
    e16, d16 = rel_pair(out16, out16_ref)
    ref32 = {k: v.clone() for k, v in w32.grads().items()}
    bufs_ref = {k: v.clone() for k, v in w32.buffers().items()}
    del w32, out_ref, out16_ref, out, out16
    # reference vs reference: the SAME fp32 CPU oracle on the same inputs with ONE intra-op thread
    # instead of all cores -- ATen's reductions then sum in another order, a few ReLU units land on the other side of zero,
    # and the reference's own gradients move.  Where that spread exceeds 1e-3 the literal "1e-3 against the PyTorch-CPU
    # reference" clause is not decidable by any implementation, the reference included.
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    w32b = model_ref.Weights(sd)
    model_ref.train_step(w32b, im, lb, mode)
    ref32_1t = {k: v.clone() for k, v in w32b.grads().items()}
    del w32b
    torch.set_num_threads(threads)
    w64 = model_ref.Weights(sd, dtype=torch.float64)
    taps64 = {}
    _, _, loss64 = model_ref.train_step(w64, im.double(), lb, mode, taps=taps64)
    taps64 = _tap_copies(taps64)
    ref64 = w64.grads()
    rows = gradient_table(net, ref32, ref64)
    spread = []
    for k, r in rows.items():
        ediff, _ = rel_pair(ref32_1t[k], ref32[k])
        r["ref32_1thread_vs_ref32_allthreads"] = ediff / max(r["norm"], 1e-300)
        if not r["analytic_zero"]:
            spread.append(r["ref32_1thread_vs_ref32_allthreads"])
    spread.sort()
    ref_vs_ref = dict(threads=[threads, 1], tensors=len(spread), past_1e3=sum(x > TOL for x in spread),
                      median=spread[len(spread) // 2], p90=spread[int(0.9 * len(spread))], max=spread[-1],
                      what="||g_ref32(1 thread) - g_ref32(all threads)|| / ||g_f64|| per gradient tensor: the fp32 PyTorch-CPU "
                           "reference against ITSELF with another summation order")
    del ref32_1t
    cab_insitu = cab_table(net, sd, cap)  # which CAB gradients of THIS run are exact on the model's own tensors
    # the in-situ-backed clause is gated on what ENTERS the CAB's backward in this run: d(cab.y) against the fp64 model's
    failures, listed = judge_gradients(rows, load_allowlist()[tag], cab_insitu,
                                       upstream={"d.cab.y": rel(cap["d.cab.y"], taps64["d.cab.y"])})
    worst = sorted(((r["gpu_vs_f64"], k) for k, r in rows.items() if not r["analytic_zero"]), reverse=True)[:10]
    # Where the gradient noise enters: distance from the fp64 model of every tensor either side of the
    # hot path, for the HIP model and for the fp32 CPU reference ...
    chain = {}
    for name in TAPS:
        for k in (name, "d." + name):
            chain[k] = dict(gpu_vs_f64=rel(cap[k], taps64[k]), ref32_vs_f64=rel(taps32[k], taps64[k]))
    # ReLU units whose mask differs from the fp64 model's, and how many units the map has: ONE flipped unit of a map with N
    # units moves the gradient behind it by ~1/sqrt(N)
    flips = {name: dict(gpu=int(((cap[k] > 0) != (taps64[k] > 0)).sum()), ref32=int(((taps32[k] > 0) != (taps64[k] > 0)).sum()),
                        units=taps64[k].numel()) for name, k in RELUS.items()}
    # ... and the HIP CAB run on the fp64 MODEL's own input and incoming gradient (rounded to fp32): its parameter gradients
    # against the fp64 model's.  Small here + large in `tensors` = the operator is exact and its inputs carry the noise.
    y, dx, gcab = _hip_cab_on(sd, taps64["cab.x"], taps64["d.cab.y"])
    cross = {"cab.out": rel(y, taps64["cab.y"]), "cab.dx": rel(dx, taps64["d.cab.x"])}
    cross.update({k: rel(v, ref64[k]) for k, v in gcab.items() if float(ref64[k].norm()) > 1e-9})
    del taps32, taps64
    write_table(f"parity_{tag}.json", dict(
        config=dict(mode=mode, batch=batch, height=height, width=width, n_classes=ncls, gamma=0.5, model_seed=0,
                    data_seed=1),
        logits_rel=e / d, logits16_rel=e16 / d16, loss_gpu=loss, loss_ref32=float(loss_ref), loss_f64=float(loss64),
        tolerance=TOL, allow_factor=ALLOW_FACTOR, rule=load_allowlist()[tag],
        past_1e3_within_bound=listed, failures=[k for k, _ in failures], worst_vs_f64=worst,
        chain_either_side_of_the_hot_path=chain, relu_mask_flips_vs_f64=flips, hip_cab_on_fp64_model_inputs=cross,
        reference_vs_reference=ref_vs_ref, cab_insitu_vs_f64_replay=cab_insitu, tensors=rows))
    assert e <= TOL * d, f"final_logit rel {e / d:.3e}"
    assert e16 <= TOL * d16, f"high_res_logit_up rel {e16 / d16:.3e}"
    assert abs(loss - float(loss_ref)) <= TOL * abs(float(loss_ref)), (loss, float(loss_ref))
    assert not failures, [(k, {n: f"{v:.2e}" for n, v in r.items() if isinstance(v, float)}) for k, r in failures]
    # BatchNorm running buffers of the whole model, after one step
    for k, want in bufs_ref.items():
        if "running_" in k:
            assert_close(bufs_gpu[k].double(), want.double(), 1e-4, k)


@pytest.mark.timeout(2400)
def test_full_step_config3_large_8x1024x1024():
    """BASELINE config 3 -- the configuration bench.py times."""
    _full_step("large", 8, 1024, 1024, 8, "config3")


@pytest.mark.timeout(1800)
def test_full_step_config5_large_2x2048x1024_19cls():
    """BASELINE config 5 (n = 2048, H' != W', 19 classes)."""
    _full_step("large", 2, 2048, 1024, 19, "config5")


@pytest.mark.parametrize("mode,batch,size,ncls", [("small", 4, 512, 8), ("large", 2, 512, 19)])
def test_model_eval_bn_gradients(mode, batch, size, ncls):
    """BatchNorm in eval mode on populated running statistics: the network keeps every hand-written kernel in the
    loop (eval-mode BN folds, attention, FFM, OHEM) but loses the batch-statistic coupling that makes the train-mode
    gradient ill-conditioned: the fp32 reference is then within 1e-4 of fp64 on the median tensor, and so is the HIP model
    (profiles/r02_parity_eval_*.json).  What is left are ReLU-mask flips, which no fp32 implementation can avoid: on big
    maps ~sqrt(0.8 * forward error) per ReLU layer (sb.*: three layers, 1.4e-3 for the CPU reference itself); on the 16x16
    maps of Large 2x512^2 ONE flipped unit of q (1 of 65,536, found with tools/diag_cab_internal.py: with the fp64 mask the
    GPU's own dq gives d(beta_q) to 3.4e-5) moves d(beta_q) by 2e-3 and dW_q by 4.5e-3 -- those two tensors are the named
    entries of tests/golden/grad_allowlist.json for this case.  Same rule as everywhere (tests/parity_rules.py), plus the
    in-situ operator table of the same step (tests/insitu.py: fp64 oracle replayed on the model's own captured inputs)."""
    from cabinet_amd.train import build_model, make_criteria, synthetic_batch
    from oracle import model_ref

    net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5)
    # populate the running statistics with one calibration batch (momentum 1: running = batch statistics)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    net.train()
    with torch.no_grad():
        net(synthetic_batch(batch, size, size, ncls, "cpu", seed=7)[0])
    net.eval()
    sd = copy.deepcopy(net.state_dict())
    im, lb = synthetic_batch(batch, size, size, ncls, "cpu", seed=1)
    n_min = max(1, batch * size * size // 16)
    refs = {}
    for dt in (torch.float32, torch.float64):
        w = model_ref.Weights(sd, dtype=dt)
        o, o16 = model_ref.cabinet_forward(w, im.to(dt), mode, training=False)
        loss_ref = model_ref.ohem_ce(o, lb, 0.7, n_min) + model_ref.ohem_ce(o16, lb, 0.7, n_min)
        loss_ref.backward()
        refs[dt] = (o.detach(), o16.detach(), float(loss_ref.detach()), w.grads())
    out_ref, out16_ref, loss32, ref32 = refs[torch.float32]
    assert float(out_ref.abs().mean()) > 1e-2  # parity trap 3: eval statistics are populated, activations are O(1)
    net = net.cuda()
    cap = instrument(net)
    crit = make_criteria(batch, size, size, "cuda")
    out, out16 = net(im.cuda())
    loss = crit[0](out, lb.cuda()) + crit[1](out16, lb.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert_close(out, out_ref, TOL, "final_logit")
    assert_close(out16, out16_ref, TOL, "high_res_logit_up")
    assert abs(float(loss) - loss32) <= 1e-4 * abs(loss32)
    tag = f"eval_{mode}_{batch}x{size}"
    insitu, _ = operator_table(net, sd, cap, lb, (size, size), n_min, training=False)
    bad = judge_operator_table(insitu, TOL, training=False)
    assert not bad, {k: {n: (f"{v:.2e}" if isinstance(v, float) else v) for n, v in r.items()} for k, r in bad.items()}
    rows = gradient_table(net, ref32, refs[torch.float64][3])
    assert len(rows) > 150
    failures, listed = judge_gradients(rows, load_allowlist()[tag], insitu)
    write_table(f"parity_{tag}.json", dict(past_1e3_within_bound=listed, failures=[k for k, _ in failures], tensors=rows,
                                           insitu_operators_vs_f64_replay=insitu))
    assert not failures, [(k, {n: f"{v:.2e}" for n, v in r.items() if isinstance(v, float)}) for k, r in failures]
    for k, v in net.state_dict().items():  # eval mode: no buffer moved
        if "running_" in k:
            assert torch.equal(v.cpu(), sd[k]), k


# ------------------------------------------------------------------------------------------------------------------
# operators at the grids the timed step runs them on (config 3: B=8, image 1024^2; config 5: B=2, 2048x1024)


def test_ffm_upsampled_config3_grid():
    """8 x (128 | 256@32^2 -> 256) @ 128^2, training BN (reference cabinet.py:228-236)."""
    from test_gpu_ffm import test_ffm_upsampled_vs_oracle

    test_ffm_upsampled_vs_oracle(8, 128, 256, 256, 64, 128, 128, 32, 32, True)


def test_ffm_upsampled_config5_grid():
    from test_gpu_ffm import test_ffm_upsampled_vs_oracle

    test_ffm_upsampled_vs_oracle(2, 128, 256, 256, 64, 256, 128, 64, 32, True)


def test_ffm_plain_config3_grid():
    """FeatureFusionModule.forward(fsp, fcp) itself at the config-3 grid (the materialised-upsample form)."""
    from test_gpu_ffm import test_ffm_vs_oracle

    test_ffm_vs_oracle(8, 128, 256, 256, 64, 128, 128, True)


@pytest.mark.parametrize("B,C,Hl,Wl,H,W", [(8, 8, 128, 128, 1024, 1024), (2, 19, 256, 128, 2048, 1024)])
def test_ohem_up_production_grid(B, C, Hl, Wl, H, W):
    from test_gpu_ohem import test_fused_ohem_vs_oracle

    test_fused_ohem_vs_oracle(B, C, Hl, Wl, H, W, 0.0, 0.7, B * H * W // 16)


@pytest.mark.parametrize("act", ["relu", "hardswish"])
def test_bn_act_production_grid(act):
    """8 x 64 x 512^2 (sb.conv1's BatchNorm + ReLU: the largest plane of the step) and the backbone stem (16 ch)."""
    from test_gpu_bn_act import test_bn_act_vs_oracle

    test_bn_act_vs_oracle((8, 64 if act == "relu" else 16, 512, 512), act, True)


def test_bn_act_dwconv_production_grid():
    """features.2 of the Large backbone: BN + ReLU + depthwise 3x3 stride 2 on 8 x 64 x 512^2."""
    from test_gpu_dwconv import test_bn_act_dwconv_vs_oracle

    test_bn_act_dwconv_vs_oracle(8, 64, 512, 512, 3, 2, "relu", True)


def test_dwconv_production_grid():
    """features.1 of the Large backbone: depthwise 3x3 stride 1 on 8 x 16 x 512^2."""
    from test_gpu_dwconv import test_dwconv_vs_oracle

    test_dwconv_vs_oracle(8, 16, 512, 512, 3, 1)


@pytest.mark.parametrize("Ci,Co", [(16, 64), (16, 16), (64, 24)])
def test_pwconv_production_grid(Ci, Co):
    """The thin pointwise layers the dispatcher routes to K10 (planes >= 256^2): 16->64 and 16->16 @512^2, 64->24 @256^2."""
    from test_gpu_pwconv import test_pwconv_vs_oracle

    side = 512 if Ci == 16 else 256
    test_pwconv_vs_oracle(8, Ci, Co, side, side)


def test_stem_conv_production_grid():
    """sb.conv1: 7x7 stride 2 on 8 x 3 x 1024^2."""
    from test_gpu_stem import test_stem_conv_vs_oracle

    test_stem_conv_vs_oracle(8, 1024, 1024)


def test_cab_block_config5_grid():
    """ContextAggregationBlock at the config-5 grid (2 x 256 x 64 x 32: n = 2048, H' != W'): K6 + K1/K2 + conv1x1 + K5
    against the functional oracle in fp64 (reference cab.py:192-216)."""
    from cabinet_amd.models.cab import ContextAggregationBlock
    from oracle import model_ref

    torch.manual_seed(3)
    cab = ContextAggregationBlock(256, 128)
    with torch.no_grad():
        cab.gamma.fill_(0.5)
        torch.nn.init.kaiming_normal_(cab.global_attn.project_out.weight, a=1)
    sd = copy.deepcopy(cab.state_dict())
    g0 = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 64, 32, generator=g0)
    g = torch.randn(2, 256, 64, 32, generator=g0)
    w = model_ref.Weights(sd, dtype=torch.float64)
    xo = x.double().requires_grad_(True)
    yo = model_ref.cab_forward(w, xo, True)
    yo.backward(g.double())
    cab = cab.cuda().train()
    xd = x.cuda().requires_grad_(True)
    y = cab(xd)
    y.backward(g.cuda())
    torch.cuda.synchronize()
    assert_close(y, yo, TOL, "out")
    assert_close(xd.grad, xo.grad, TOL, "dx")
    ref = w.grads()
    # gamma of refine.0 / refine.1's BatchNorm sits in front of a depthwise conv + batch-statistics BatchNorm with beta = 0:
    # the output is invariant to its per-channel scale, the gradient is analytically zero and what is computed is the
    # rounding noise of a cancelling sum -- bounded against the same-shaped gradient of refine.2's gamma
    scale_ref = float(ref["local_attn.refine.2.block.1.weight"].norm())
    for k, p in cab.named_parameters():
        if k in ("local_attn.refine.0.block.1.weight", "local_attn.refine.1.block.1.weight"):
            assert float(ref[k].norm()) < 1e-3 * scale_ref and float(p.grad.double().norm()) < 1e-3 * scale_ref, k
            continue
        assert_close(p.grad, ref[k], TOL, k, atol=1e-9)
