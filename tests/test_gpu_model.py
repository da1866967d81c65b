"""Module- and model-level parity on the GPU: nn.Module mirror (HIP CAB/FFM inside) vs
vectors recorded from the reference and vs the CPU oracle on the same weights and inputs."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: logits and grads within 1e-3 relative (||a-b||/||b|| per tensor), fp32


def _npz(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_cab_module_golden(mode):
    """ContextAggregationBlock(256,128) with gamma=0.5 vs the reference's outputs and autograd grads."""
    from cabinet_amd.models.cab import ContextAggregationBlock

    g = _npz("g2_cab.npz")
    cab = ContextAggregationBlock(256, 128)
    cab.load_state_dict({k[5:]: v for k, v in g.items() if k.startswith("init.")})
    cab = cab.cuda().train(mode == "train")
    x = g["x"].cuda().requires_grad_(True)
    y = cab(x)
    y.backward(g["g"].cuda())
    torch.cuda.synchronize()
    assert_close(y, g[f"{mode}.out"], TOL, "out")
    assert_close(x.grad, g[f"{mode}.dx"], TOL, "dx")
    for k, p in cab.named_parameters():
        assert_close(p.grad, g[f"{mode}.grad.{k}"], TOL, k)
    if mode == "train":
        sd = cab.state_dict()
        for k, want in g.items():
            if k.startswith("after."):
                assert_close(sd[k[6:]].double(), want.double(), 1e-5, k)


@pytest.mark.parametrize("mode", ["small", "large"])
def test_model_known_answers_from_reference(mode):
    """Seeded CABiNet on the GPU reproduces the numbers the reference produced on CPU
    (tests/golden/kat_model.json): eval forward, then fwd + 2x OHEM-CE + bwd with gamma = 0.5."""
    from cabinet_amd.train import TrainStep, build_model, make_criteria

    kat = json.load(open(os.path.join(GOLDEN, "kat_model.json")))[mode]
    net = build_model(mode, n_classes=8, seed=kat["model_seed"], freeze_unused=False, device="cuda")
    e = kat["eval"]
    torch.manual_seed(e["data_seed"])
    x = torch.randn(*e["shape"]).cuda()
    net.eval()
    with torch.no_grad():
        out, out16 = net(x)
    assert abs(float(out.abs().mean()) - e["out_abs_mean"]) < 1e-4 * e["out_abs_mean"]
    assert np.allclose([float(t) for t in out[0, :, 0, 0]], e["out_0_c_0_0"], rtol=1e-3, atol=1e-5)
    t = kat["train"]
    with torch.no_grad():
        net.ab.a2block.gamma.fill_(t["gamma"])
    net.train()
    sd0 = copy.deepcopy(net.state_dict())
    torch.manual_seed(t["data_seed"])
    x = torch.randn(*t["shape"])
    lb = torch.randint(0, 8, (t["shape"][0], t["shape"][2], t["shape"][3]))
    step = TrainStep(net, make_criteria(t["shape"][0], t["shape"][2], t["shape"][3], "cuda"))
    loss = step(x.cuda(), lb.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - t["loss"]) < 1e-4 * t["loss"]
    gn = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
    assert sorted(k for k, p in net.named_parameters() if p.grad is None) == t["params_without_grad"]
    bad = {k: (gn[k], w) for k, w in t["grad_norms"].items() if abs(gn[k] - w) > TOL * w + 1e-7}
    if bad:
        # gamma.grad = <dout, global(x)> is one scalar out of 2*256*64 cancelling products: the reference's own
        # fp32 value (6.7095e-4 for "small") is 6.6e-4 away from the fp64 value (6.7139e-4), so the 1e-3 bound
        # around the fp32 number is not decidable for it.  Arbitrate such entries with the fp64 oracle.
        from oracle import model_ref

        w64 = model_ref.Weights(sd0, dtype=torch.float64)
        model_ref.train_step(w64, x.double(), lb, mode)
        g64 = {k: float(v.norm()) for k, v in w64.grads().items()}
        bad = {k: (v[0], v[1], g64[k]) for k, v in bad.items() if abs(v[0] - g64[k]) > TOL * g64[k] + 1e-7}
    assert not bad, bad
    total = float(np.sqrt(sum(v * v for v in gn.values())))
    assert abs(total - t["global_grad_norm"]) < TOL * t["global_grad_norm"]


@pytest.mark.parametrize("mode", ["small", "large"])
def test_step_is_bit_reproducible_in_the_parity_session(mode):
    """The step behind test_model_known_answers_from_reference (KAT3 recipe, reference train.py:429-441), three times from one
    state_dict: every logit and every gradient bit-identical.  The hand-written kernels sum in fixed orders (asserted per
    operator); what made the WHOLE step -- and with it the verdict of every model-level parity test -- depend on the run was a
    stock convolution solver with atomics in the spatial branch's forward (tools/diag_step_determinism.py,
    profiles/r06_step_determinism.txt); tests/conftest.py pins MIOpen to its deterministic solvers for this session."""
    from cabinet_amd.train import TrainStep, build_model, make_criteria

    assert torch.backends.cudnn.deterministic, "tests/conftest.py pins the deterministic solvers for GPU sessions"
    kat = json.load(open(os.path.join(GOLDEN, "kat_model.json")))[mode]
    t = kat["train"]
    net = build_model(mode, n_classes=8, seed=kat["model_seed"], freeze_unused=False, device="cuda")
    with torch.no_grad():
        net.ab.a2block.gamma.fill_(t["gamma"])
    net.train()
    sd0 = copy.deepcopy(net.state_dict())
    torch.manual_seed(t["data_seed"])
    x = torch.randn(*t["shape"]).cuda()
    lb = torch.randint(0, 8, (t["shape"][0], t["shape"][2], t["shape"][3])).cuda()
    runs = []
    for _ in range(3):
        net.load_state_dict(sd0)
        low, low16 = net.forward_lowres(x)
        net.load_state_dict(sd0)
        step = TrainStep(net, make_criteria(t["shape"][0], t["shape"][2], t["shape"][3], "cuda"))
        loss = step(x, lb)
        torch.cuda.synchronize()
        runs.append((low.detach().clone(), low16.detach().clone(), float(loss),
                     {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]), "forward logits differ between repeats"
        assert r[2] == runs[0][2]
        moved = [k for k, g in r[3].items() if not torch.equal(g, runs[0][3][k])]
        assert not moved, f"{len(moved)} of {len(r[3])} gradient tensors differ between repeats: {moved[:6]}"


@pytest.mark.parametrize("mode,batch,size,ncls", [("small", 4, 512, 8),      # BASELINE config 2
                                                 ("large", 2, 512, 19)])
def test_model_vs_oracle_logits_and_grads(mode, batch, size, ncls):
    """HIP-backed model on cuda:0 vs the functional CPU oracle: same random-init weights (gamma=0.5), same synthetic
    input; logits within 1e-3 relative, and gradient tensors under the rule of tests/parity_rules.py: within 1e-3
    of the fp32 reference or of the fp64 oracle, else no further from fp64 than 3x the fp32 reference's own measured
    distance from fp64 / three unit ReLU flips on the CAB grid.  Small 4x512^2 (BASELINE config 2):
    every tensor.  Large 2x512^2: the hot-path modules' parameters (CAB, FFM); its backbone gradients are decided by
    single ReLU flips on 16x16 maps (one flipped unit of 2x256x16x16 moves everything upstream by 3e-3, DESIGN.md
    section 5) and are checked where flips average out: at full size, test_gpu_fullsize.py (configs 3 and 5).
    (Batch >= 2: with B = 1 and batch-statistics BatchNorm every squeeze-excite input is exactly the BN bias, i.e. pure
    rounding noise in front of a ReLU -- the reference's own fp32 and fp64 gradients then differ by 100 %.)"""
    from cabinet_amd.train import build_model, make_criteria, synthetic_batch
    from oracle import model_ref
    from insitu import instrument, judge_operator_table, operator_table
    from parity_rules import gradient_table, judge_gradients, load_allowlist, write_table

    net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False)
    sd = copy.deepcopy(net.state_dict())
    im, lb = synthetic_batch(batch, size, size, ncls, "cpu", seed=1)
    w = model_ref.Weights(sd)
    out_ref, out16_ref, loss_ref = model_ref.train_step(w, im, lb, mode)
    w64 = model_ref.Weights(sd, dtype=torch.float64)  # the truth both fp32 results approximate
    model_ref.train_step(w64, im.double(), lb, mode)
    net = net.cuda().train()
    cap = instrument(net)
    crit = make_criteria(batch, size, size, "cuda")
    out, out16 = net(im.cuda())
    loss = crit[0](out, lb.cuda()) + crit[1](out16, lb.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert_close(out, out_ref, TOL, "final_logit")
    assert_close(out16, out16_ref, TOL, "high_res_logit_up")
    assert abs(float(loss.detach()) - float(loss_ref)) < 1e-4 * float(loss_ref)
    tag = f"{mode}_{batch}x{size}"
    insitu, _ = operator_table(net, sd, cap, lb, (size, size), max(1, batch * size * size // 16))
    bad = judge_operator_table(insitu, TOL)
    assert not bad, {k: {n: (f"{v:.2e}" if isinstance(v, float) else v) for n, v in r.items()} for k, r in bad.items()}
    rows = gradient_table(net, w.grads(), w64.grads())
    if mode == "large":
        rows = {k: r for k, r in rows.items() if k.startswith(("ab.a2block.", "ffm."))}
        assert len(rows) >= 25
    failures, listed = judge_gradients(rows, load_allowlist()[tag], insitu)
    write_table(f"parity_{tag}.json", dict(past_1e3_within_bound=listed, failures=[k for k, _ in failures], tensors=rows,
                                           insitu_operators_vs_f64_replay=insitu))
    assert not failures, [(k, {n: f"{v:.2e}" for n, v in r.items() if isinstance(v, float)}) for k, r in failures]
    # BatchNorm side effects of the hot path match too
    bufs = w.buffers()
    sd_after = net.state_dict()
    for k in ("ffm.convblk.bn.running_mean", "ffm.convblk.bn.running_var", "ffm.convblk.bn.num_batches_tracked",
              "ab.a2block.global_attn.to_query.1.running_var"):
        assert_close(sd_after[k].double(), bufs[k].double(), 1e-4, k)


def test_eval_no_grad_deepcopy_and_odd_sizes():
    """evaluate.py-style use: eval(), no_grad, arbitrary H x W (n not a tile multiple), deep-copied model."""
    from cabinet_amd.train import build_model
    from oracle import model_ref

    net = build_model("small", n_classes=8, seed=0, gamma=0.5).cuda()
    # make eval statistics meaningful: one train-mode pass to move the running stats
    with torch.no_grad():
        net.train()
        net(torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(5)).cuda())
    ema = copy.deepcopy(net).eval()
    x = torch.randn(1, 3, 288, 416, generator=torch.Generator().manual_seed(6))  # H'=9, W'=13 -> n=117
    with torch.no_grad():
        a = ema(x.cuda())[0]
        b = ema(x.cuda())[0]
    assert torch.equal(a, b)  # deterministic
    ref = model_ref.cabinet_forward(model_ref.Weights(ema.state_dict(), requires_grad=False), x, "small", False)[0]
    assert_close(a, ref, TOL, "eval logits")


def test_hot_path_uses_hip_library_not_aten():
    """The forward of CAB attention / FFM on device tensors goes through libcabinet_hip.so."""
    from cabinet_amd import _lib, functional
    from cabinet_amd.models.cabinet import FeatureFusionModule

    calls = []
    real = _lib.load()

    class Spy:
        def __getattr__(self, name):
            fn = getattr(real, name)

            def wrapped(*a):
                calls.append(name)
                return fn(*a)
            return wrapped

    old = _lib._lib
    _lib._lib = Spy()
    try:
        ffm = FeatureFusionModule(384, 256).cuda()
        ffm(torch.randn(1, 128, 8, 8).cuda(), torch.randn(1, 256, 8, 8).cuda()).sum().backward()
        q = torch.randn(1, 128, 64).cuda().requires_grad_(True)
        functional.cab_attention(q, q.detach(), q.detach(), 0.1).sum().backward()
    finally:
        _lib._lib = old
    for name in ("cabinet_ffm_fwd", "cabinet_ffm_bwd", "cabinet_cab_attn_fwd", "cabinet_cab_attn_bwd"):
        assert name in calls, name
    loaded = open("/proc/self/maps").read()
    assert "libcabinet_hip.so" in loaded


def test_reference_style_autocast_step_runs():
    """The reference's train_step wraps forward + loss in autocast (train.py:433).  The custom Functions must
    accept the fp16 activations autocast hands them (custom_fwd casts to fp32), return finite losses and
    gradients, and stay close to the fp32 step."""
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

    im, lb = synthetic_batch(2, 256, 256, 8, "cuda", seed=11)
    losses, gnorms = [], []
    for autocast in (False, True):
        net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        step = TrainStep(net, make_criteria(2, 256, 256, "cuda"), autocast=autocast)
        losses.append(float(step(im, lb)))
        g = [p.grad for p in net.parameters() if p.grad is not None]
        assert all(torch.isfinite(t).all() for t in g)
        assert all(t.dtype == torch.float32 for t in g)
        gnorms.append(float(torch.sqrt(sum(t.double().pow(2).sum() for t in g))))
    assert abs(losses[1] - losses[0]) < 2e-2 * abs(losses[0])
    assert abs(gnorms[1] - gnorms[0]) < 0.1 * gnorms[0]


def test_reference_amp_gradscaler_clip_step():
    """The reference's REAL step (train.py:386,411-441): autocast forward, ``scaler.scale(loss).backward()``,
    ``scaler.unscale_`` + ``clip_grad_norm_``, ``scaler.step`` / ``update`` -- through ``TrainStep(autocast=True, scaler=...)``.
    Checked: (i) autocast really is on (a stock convolution of the spatial branch emits fp16) while EVERY tensor handed to
    the C ABI by the hot-path operators is fp32; (ii) the gradients the clipping callback sees are unscaled (their norm
    equals the fp32 step's, not 65536x it) and point the same way; (iii) loss within 1e-2 of the fp32 step; (iv) the
    scaler did not skip the steps and the weights moved and stayed finite."""
    from cabinet_amd import functional
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

    im, lb = synthetic_batch(2, 256, 256, 8, "cuda", seed=11)
    seen = {}

    def spy(name):
        real = getattr(functional, name)

        def wrapped(*args, **kw):
            seen.setdefault(name, set()).update(a.dtype for a in args if isinstance(a, torch.Tensor) and a.is_floating_point())
            return real(*args, **kw)
        return real, wrapped

    # (round 5: the model's CAB runs K1 with the output projection in its epilogue -- attn_proj_fwd_hip -- at this shape)
    names = ["attn_proj_fwd_hip", "attn_bwd_hip", "ffm_up_fwd_hip", "ffm_up_bwd_hip", "cab_local_fwd_hip", "cab_local_bwd_hip",
             "ohem_up_pair_fwd_hip", "ohem_up_pair_bwd_hip"]
    results = {}
    for amp in (False, True):
        net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        w0 = [p.detach().clone() for p in net.parameters()]
        opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.9)
        norms, conv_dtypes = [], []
        hook = net.sb.conv2.conv.register_forward_hook(lambda m, i, o: conv_dtypes.append(o.dtype))

        flat = []

        def clip():  # what the reference's _optimizer_step does between unscale_ and scaler.step (train.py:414-416)
            flat.append(torch.cat([p.grad.flatten().double() for p in net.parameters() if p.grad is not None]))
            norms.append(float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)))

        scaler = torch.amp.GradScaler(device="cuda") if amp else None
        step = TrainStep(net, make_criteria(2, 256, 256, "cuda"), optimizer=opt, autocast=amp, scaler=scaler,
                         before_optimizer=clip)
        saved = {}
        if amp:
            for n in names:
                saved[n], w = spy(n)
                setattr(functional, n, w)
        try:
            losses = [float(step(im, lb)) for _ in range(3)]
        finally:
            for n, real in saved.items():
                setattr(functional, n, real)
            hook.remove()
        assert all(torch.isfinite(p).all() for p in net.parameters())
        moved = sum(float((p.detach() - w).abs().sum()) for p, w in zip(net.parameters(), w0))
        results[amp] = dict(losses=losses, norms=norms, flat=flat, conv=conv_dtypes, moved=moved,
                            scale=scaler.get_scale() if amp else None)
    fp32, amp = results[False], results[True]
    assert set(fp32["conv"]) == {torch.float32} and set(amp["conv"]) == {torch.float16}   # (i) autocast is active ...
    for n in names:                                                                        # ... and the hot path is fp32
        assert seen.get(n) == {torch.float32}, (n, seen.get(n))
    # (iv) GradScaler skips a step whose fp16 gradients overflowed (early calibration, train.py:421-427) and halves its scale;
    # the weights are then unchanged, so the first step it did NOT skip still starts from the fp32 run's initial weights
    import math

    first = next(i for i, v in enumerate(amp["norms"]) if math.isfinite(v))
    assert first <= 1 and amp["scale"] >= 65536.0 / 2 and amp["moved"] > 0, (amp["norms"], amp["scale"])
    # (ii) what the clipping callback saw: true (unscaled) gradients, same length and direction as the fp32 step's
    assert abs(amp["norms"][first] - fp32["norms"][0]) < 5e-2 * fp32["norms"][0], (amp["norms"], fp32["norms"])
    ga, gf = amp["flat"][first], fp32["flat"][0]
    cos = float(torch.dot(ga, gf) / (ga.norm() * gf.norm()))
    assert cos > 0.98, cos
    # (iii) loss on identical weights: fp16 backbone noise only
    assert abs(amp["losses"][first] - fp32["losses"][0]) < 1e-2 * abs(fp32["losses"][0]), (amp["losses"], fp32["losses"])


def test_non_contiguous_and_channels_last_inputs_are_accepted():
    """Borrowed inputs are made dense NCHW by the binding (the C ABI itself only takes dense pointers)."""
    from cabinet_amd.functional import cab_attention
    from cabinet_amd.models.cabinet import FeatureFusionModule
    from oracle.cab_math import attn_core_fwd

    g = torch.Generator().manual_seed(4)
    q = torch.randn(2, 70, 128, generator=g).cuda().transpose(1, 2)  # (2,128,70) view, non-contiguous
    k = torch.randn(2, 128, 70, generator=g).cuda()
    v = torch.randn(2, 128, 70, generator=g).cuda()
    out = cab_attention(q, k, v, 0.1)
    ref, _ = attn_core_fwd(q.cpu().contiguous(), k.cpu(), v.cpu(), 0.1)
    assert_close(out, ref, TOL, "ctx (non-contiguous q)")
    ffm = FeatureFusionModule(384, 256).cuda().eval()
    fsp = torch.randn(2, 128, 16, 16, generator=g).cuda()
    fcp = torch.randn(2, 256, 16, 16, generator=g).cuda()
    with torch.no_grad():
        a = ffm(fsp, fcp)
        b = ffm(fsp.contiguous(memory_format=torch.channels_last), fcp.contiguous(memory_format=torch.channels_last))
    assert_close(b, a, 1e-6, "channels_last input")


def test_graphed_train_step_equals_eager():
    """GraphedTrainStep (two captured hipGraphs around the one OHEM host read) reproduces TrainStep step for step: same
    losses, same weights after SGD, same BatchNorm buffers; a batch that needs the rare branch (here: every label ignored)
    falls back to the eager step without double-counting the BatchNorm side effects of the replayed forward."""
    from cabinet_amd.train import GraphedTrainStep, TrainStep, build_model, make_criteria, synthetic_batch

    batches = [synthetic_batch(2, 256, 256, 8, "cuda", seed=20 + i) for i in range(4)]
    ign = (batches[0][0], torch.full_like(batches[0][1], 255))
    res = []
    for graphed in (False, True):
        net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-2, momentum=0.9)
        crit = make_criteria(2, 256, 256, "cuda")
        if graphed:
            step = GraphedTrainStep(net, crit, optimizer=opt, warmup=1)
        else:
            step = TrainStep(net, crit, optimizer=opt)
        losses = [float(step(*b)) for b in batches]
        losses.append(float(step(*ign)))       # all-ignored batch: zero loss, eager fallback inside the graphed step
        losses.append(float(step(*batches[1])))
        if graphed:
            assert step.fallbacks == 1 and step.g_bwd is not None
        res.append((losses, {k: v.clone() for k, v in net.state_dict().items()}))
    (la, sa), (lb_, sb) = res
    assert la[4] == 0.0 and lb_[4] == 0.0
    # stock backward kernels with atomics differ run to run (1e-6 on the loss, up to a few 1e-4 on the ill-conditioned first
    # depthwise / stem weights once SGD has carried it along for six steps: seen 2.4e-4 and 2.8e-4 in full-suite runs); a
    # capture bug -- a stale buffer, a missed kernel, a wrong branch -- is an O(1) error
    for x, y in zip(la, lb_):
        assert abs(x - y) <= 1e-4 * max(1.0, abs(x)), (la, lb_)
    for k in sa:
        assert_close(sb[k].double(), sa[k].double(), 2e-3, k, atol=1e-5)


class _WarmupPolyOptimizer:
    """The shape of the reference's optimizer wrapper (src/utils/optimizer.py:141-155): ``step()`` computes the learning
    rate on the HOST from its own step counter (linear warm-up, then polynomial decay), writes it into the param groups and
    only then steps SGD."""

    def __init__(self, params, lr0=2e-2, warmup_steps=2, max_iter=10, power=0.9):
        self.opt = torch.optim.SGD(params, lr=lr0, momentum=0.9, weight_decay=5e-4)
        self.lr0, self.warmup_steps, self.max_iter, self.power, self.it = lr0, warmup_steps, max_iter, power, 0
        self.lrs = []

    def step(self):
        if self.it < self.warmup_steps:
            lr = self.lr0 * (self.it + 1) / (self.warmup_steps + 1)
        else:
            lr = self.lr0 * (1 - (self.it - self.warmup_steps) / (self.max_iter - self.warmup_steps)) ** self.power
        for pg in self.opt.param_groups:
            pg["lr"] = lr
        self.lrs.append(lr)
        self.opt.step()
        self.it += 1


def test_graphed_step_follows_a_host_side_lr_schedule_and_clipping():
    """a captured ``optimizer.step()`` bakes lr / weight decay / momentum into the graph and freezes a
    Python-side schedule at its capture-time value.  The graphed steps therefore run the optimizer (and an optional
    gradient-clipping callback, reference train.py:411-427) EAGERLY after the backward graph: six steps of warm-up + poly decay
    through GraphedTrainStep equal the eager TrainStep, learning rates included; capturing the optimizer is opt-in, refuses a
    wrapper object, and raises when a hyper-parameter has moved since capture."""
    from cabinet_amd.train import GraphedTrainStep, TrainStep, build_model, make_criteria, synthetic_batch

    batches = [synthetic_batch(2, 256, 256, 8, "cuda", seed=40 + i) for i in range(6)]
    res = []
    for graphed in (False, True):
        net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        params = [p for p in net.parameters() if p.requires_grad]
        opt = _WarmupPolyOptimizer(params)
        norms = []

        def clip(params=params, norms=norms):
            norms.append(float(torch.nn.utils.clip_grad_norm_(params, 1.0)))

        crit = make_criteria(2, 256, 256, "cuda")
        step = (GraphedTrainStep(net, crit, optimizer=opt, warmup=2, before_optimizer=clip) if graphed
                else TrainStep(net, crit, optimizer=opt, before_optimizer=clip))
        losses = [float(step(*b)) for b in batches]
        if graphed:
            assert step.g_bwd is not None and step.fallbacks == 0 and step.opt_seg.graph is None
        res.append((losses, opt.lrs, norms, {k: v.clone() for k, v in net.state_dict().items()}))
    (la, lra, na, sa), (lb_, lrb, nb, sb) = res
    assert lra == lrb and len(set(lra)) == 6          # six different learning rates reached the kernels on both paths
    assert len(na) == len(nb) == 6
    # stock backward kernels with atomics differ run to run and SGD carries the difference along (seen 5e-4 on the sixth
    # gradient norm); a schedule bug -- a frozen or shifted learning rate -- moves these by tens of percent
    for x, y in zip(la + na, lb_ + nb):
        assert abs(x - y) <= 3e-3 * max(1.0, abs(x)), (la, lb_, na, nb)
    for k in sa:
        assert_close(sb[k].double(), sa[k].double(), 2e-3, k, atol=1e-5)
    # opt-in capture: plain torch optimizer only, and a changed hyper-parameter raises instead of being ignored
    net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
    with pytest.raises(RuntimeError, match="wrapper"):
        GraphedTrainStep(net, make_criteria(2, 256, 256, "cuda"), optimizer=_WarmupPolyOptimizer(params), capture_optimizer=True)
    sgd = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-2, momentum=0.9)
    step = GraphedTrainStep(net, make_criteria(2, 256, 256, "cuda"), optimizer=sgd, warmup=1, capture_optimizer=True)
    for b in batches[:3]:
        step(*b)
    assert step.opt_seg.graph is not None
    sgd.param_groups[0]["lr"] = 5e-3
    with pytest.raises(RuntimeError, match="hyper-parameters changed"):
        step(*batches[3])
    # a batch of another shape is refused (copy_ into the static input would broadcast a single image over the batch)
    with pytest.raises(RuntimeError, match="does not match the captured"):
        step(batches[0][0][:1], batches[0][1][:1])


def test_one_dispatch_rule_for_shapes_outside_kernel_coverage():
    """Device tensors with shapes no kernel family covers take the composite ATen forward ON THE DEVICE (never an error,
    never the host): ContextAggregationBlock(96, 48) -- attention pair (48,48), producers with 48 channels.  The un-tiled
    UAVid validation frame's CAB grid (1 x 256 x 68 x 128 = 8704 positions, reference train.py:444-456) is INSIDE the coverage
    since round 3 (K5's tiled form): attention, producers and the local branch all run the HIP kernels.  Both agree with the
    functional oracle (reference cab.py:192-216)."""
    from cabinet_amd import _lib
    from cabinet_amd.functional import cab_attention, cab_attention_supported, cab_local_supported
    from cabinet_amd.models.cab import ContextAggregationBlock
    from oracle import model_ref

    assert cab_attention_supported(128, 128) and cab_attention_supported(256, 128) and not cab_attention_supported(48, 48)
    q = torch.randn(1, 48, 50, device="cuda", requires_grad=True)
    out = cab_attention(q, q.detach(), q.detach(), 0.2)  # composite on the device
    out.sum().backward()
    assert out.is_cuda and q.grad is not None

    def run(C, Vc, B, H, W, seed):
        torch.manual_seed(seed)
        cab = ContextAggregationBlock(C, Vc)
        with torch.no_grad():
            cab.gamma.fill_(0.5)
            torch.nn.init.kaiming_normal_(cab.global_attn.project_out.weight, a=1)
        sd = copy.deepcopy(cab.state_dict())
        g0 = torch.Generator().manual_seed(seed)
        x, g = torch.randn(B, C, H, W, generator=g0), torch.randn(B, C, H, W, generator=g0)
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
        assert_close(cab.global_attn.to_value.weight.grad, w.grads()["global_attn.to_value.weight"], TOL, "dW_v")
        return xd

    run(96, 48, 2, 12, 10, 4)        # everything composite (48 channels)
    xd = run(256, 128, 1, 68, 128, 5)  # n = 8704: K5 tiled form, K6 + K1/K2 native
    assert cab_local_supported(xd) and "libcabinet_hip.so" in open("/proc/self/maps").read()
    assert _lib.load().cabinet_cab_attn_supported(128, 128) == 1


def test_c_abi_is_reentrant_two_threads_one_device():
    """the C ABI claims no global mutable state (per-device kernel attributes aside, which are
    idempotent).  Two host threads -- each with its own stream, its own CAB + FFM modules and its own autograd engine
    activity -- hammer the library concurrently on ONE device (ctypes releases the GIL for the duration of every entry
    point, and the backward entry points run on autograd's engine thread): every result must be bit-identical to the same
    work done serially, and cabinet_last_error stays thread-local."""
    import threading

    from cabinet_amd import _lib
    from cabinet_amd.models.cab import ContextAggregationBlock
    from cabinet_amd.models.cabinet import FeatureFusionModule

    def build(seed):
        torch.manual_seed(seed)
        cab, ffm = ContextAggregationBlock(256, 128), FeatureFusionModule(384, 256)
        with torch.no_grad():
            cab.gamma.fill_(0.5)
            torch.nn.init.kaiming_normal_(cab.global_attn.project_out.weight, a=1)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(2, 256, 16 + 2 * seed, 12, generator=g)
        fsp, low = torch.randn(2, 128, 32, 24, generator=g), torch.randn(2, 256, 8, 6, generator=g)
        return cab.cuda().train(), ffm.cuda().train(), x.cuda(), fsp.cuda(), low.cuda()

    def work(objs, reps, out):
        cab, ffm, x, fsp, low = objs
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for _ in range(reps):
                for m in (cab, ffm):
                    m.zero_grad(set_to_none=True)
                xr, fr, lr = (t.clone().requires_grad_(True) for t in (x, fsp, low))
                y = cab(xr)
                z = ffm.forward_upsampled(fr, lr)
                (y.square().mean() + z.square().mean()).backward()
            stream.synchronize()
        out.extend([y.detach().clone(), z.detach().clone(), xr.grad.clone(), fr.grad.clone(), lr.grad.clone()]
                   + [p.grad.clone() for m in (cab, ffm) for p in m.parameters()])

    serial = []
    for seed in (1, 2):
        res = []
        work(build(seed), 1, res)
        serial.append(res)
    objs = [build(1), build(2)]
    results, errors = [[], []], []

    def run(i):
        try:
            work(objs[i], 12, results[i])
            # an argument error on this thread must not leak into the other thread's error slot
            lib = _lib.load()
            assert lib.cabinet_cab_attn_fwd(None, None, None, 1.0, 1, 128, 128, 16, 0, None, None, None, 0, None) == -1
            assert b"null" in lib.cabinet_last_error()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for got, want in zip(results, serial):
        assert len(got) == len(want) >= 30
        for a, b in zip(got, want):
            # BatchNorm buffers advance with every repetition, outputs and gradients (train-mode statistics) do not
            assert torch.equal(a, b)
