#!/usr/bin/env python3
"""Headline benchmark: 1024x1024 images/s, forward + 2x OHEM-CE + backward (+ gradient
all-reduce + SGD step) of CABiNet-MobileNetV3-Large with the hand-written HIP CAB / FFM
kernels, on N MI355X of one node (BASELINE.json metric, config 3 per GPU / config 4 at N=8).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
``roofline`` (dominant hand-written kernel: algorithmic FLOPs / measured launch duration vs the
gfx950 dense fp32-MFMA peak) and ``cpu_baseline`` (the CPU oracle timed on this host, N=1 only),
plus ``kernels`` (every hand-written kernel group with its own roofline numbers).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# MIOpen user find-db / kernel cache recorded on an MI355X for this workload's convolution problems
# (stock PyTorch-ROCm backbone).  Without it the first step on a fresh box spends ~75 s in MIOpen's
# solver search; results are identical either way.  The committed copy is READ-ONLY: every process works on a private
# copy in a fresh temporary directory (MIOpen appends to its user db), removed at exit.
_MIOPEN_DB = os.path.join(ROOT, "cabinet_amd", "miopen_db")
if not os.path.isdir(_MIOPEN_DB) and int(os.environ.get("WORLD_SIZE", "1")) > 1 and "MIOPEN_USER_DB_PATH" not in os.environ:
    # several ranks of one node: every rank gets its own user database and kernel cache (seeded from the user's, if any), so
    # that eight processes do not serialise on one SQLite file while MIOpen compiles / looks up ~150 convolution problems
    import atexit
    import shutil
    import tempfile
    try:
        _tmp = tempfile.mkdtemp(prefix=f"cabinet_miopen_rank{os.environ.get('LOCAL_RANK', '0')}_")
        _src = os.path.expanduser("~/.config/miopen")
        if os.path.isdir(_src):
            shutil.copytree(_src, os.path.join(_tmp, "db"), dirs_exist_ok=True)
        os.makedirs(os.path.join(_tmp, "db"), exist_ok=True)
        atexit.register(shutil.rmtree, _tmp, ignore_errors=True)
        os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(_tmp, "db")
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(_tmp, "cache"))
    except OSError:
        pass
if os.path.isdir(_MIOPEN_DB) and "MIOPEN_USER_DB_PATH" not in os.environ:
    import atexit
    import shutil
    import tempfile
    try:
        _tmp = tempfile.mkdtemp(prefix="cabinet_miopen_")
        _db = os.path.join(_tmp, "db")
        shutil.copytree(_MIOPEN_DB, _db)
        atexit.register(shutil.rmtree, _tmp, ignore_errors=True)
        os.environ["MIOPEN_USER_DB_PATH"] = _db
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(_db, "cache"))
    except OSError:
        pass

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (never the 2:1-sparsity figure)
# a split-bf16 kernel spends 3 (bf16x3) or 6 (bf16x6) bf16 MFMA products on every fp32 product: its ALGORITHMIC rate is priced
# against the bf16 peak divided by that count
MFMA_PEAKS = {"mfma": (PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
              "mfma-bf16x3": (PEAK_BF16_MFMA_TFLOPS / 3, "dense bf16 MFMA 2.5 PFLOP/s / 3 products per fp32 product"),
              "mfma-bf16x6": (PEAK_BF16_MFMA_TFLOPS / 6, "dense bf16 MFMA 2.5 PFLOP/s / 6 products per fp32 product")}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # >= 50 timed steps behind >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE config 3: 8)")
    ap.add_argument("--size", type=int, default=1024, help="square image side (BASELINE config 3: 1024)")
    ap.add_argument("--height", type=int, default=None, help="image height when not square (BASELINE config 5: "
                    "--height 2048 --width 1024 --batch 2 --classes 19)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--mode", default="large", choices=["large", "small"])
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--kernel-iters", type=int, default=30)
    ap.add_argument("--kernels-only", action="store_true", help="only time the hand-written kernels (dev aid)")
    ap.add_argument("--all-kernels", action="store_true", help="also time the kernel groups the model's step does not run "
                    "(plain FFM without the fused upsample, single-head OHEM)")
    ap.add_argument("--no-graph", action="store_true", help="single GPU: enqueue every step eagerly instead of replaying "
                    "the step's two captured hipGraphs")
    ap.add_argument("--eval", action="store_true", help="time the forward-only path evaluate.py consumes (.eval(), no_grad, BatchNorm on "
                    "running statistics, full-resolution logits) instead of the training step; one JSON line of its own")
    ap.add_argument("--no-eval-forward", action="store_true", help="skip the forward-only measurement that the training line carries "
                    "as `eval_forward`")
    args = ap.parse_args()
    args.height = args.height or args.size
    args.width = args.width or args.size
    return args


def time_kernel(fn, iters, warm=3, reps=5):
    """Average duration (ms) of one launch group, with HIP events on the stream the kernels run on (torch's current
    stream is the stream handed to the C ABI).  The group is captured `reps` times into one hipGraph and the graph is
    replayed: several groups are a few launches of 5 .. 30 us each, and timing them through the Python binding measures
    the host (60 us per call), not the kernels.  Falls back to eager launches if a group cannot be captured."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    graph = None
    if os.environ.get("CABINET_BENCH_EAGER_KERNELS") != "1":
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(reps):
                    fn()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] kernel group not capturable ({type(e).__name__}: {e}); timing it eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph is not None:
        n = max(2, iters // reps)
        start.record()
        for _ in range(n):
            graph.replay()
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / (n * reps)
        del graph
        return ms
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters


TRAFFIC_PROFILES = ("r06_pmc_traffic.json", "r06_config5_pmc_traffic.json")  # under profiles/: one per workload shape


def load_traffic(batch, height, width, classes):
    """HBM bytes per launch of every kernel group, from the committed PMC profiles (tools/pmc_traffic.sh: rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE in separate passes, counters only; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950
    FETCH calibration of MI355X_MICROARCH.md).  A profile records the workload shape and the digest of the kernel sources it
    was collected on: it is used ONLY when both equal this run's -- otherwise every `traffic` is null (stale evidence is
    not evidence)."""
    from cabinet_amd import build as _build

    why = "none: no PMC traffic profile under profiles/"
    for name in TRAFFIC_PROFILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        prof = json.load(open(path))
        shape = (prof.get("batch"), prof.get("height", prof.get("size")), prof.get("width", prof.get("size")),
                 prof.get("classes", 8))
        if shape != (batch, height, width, classes):
            continue
        if prof.get("source_digest") != _build.source_digest():
            why = f"none: profiles/{name} was collected on other kernel sources (digest mismatch)"
            continue
        return prof.get("traffic", {}), f"profiles/{name} (rocprofv3 --pmc, FETCH x2 calibrated, digest-checked)"
    return {}, why


def kernel_cases(batch, height, width=None, classes=8, extra=False):
    """The hand-written kernel groups, with the BatchNorm ``num_batches_tracked += 1`` bookkeeping deferred as the model's
    forward defers it (``batched_bn_counters``: ONE multi-tensor add per step for the model's 59 counters): outside that
    context the module-level closures of K5 / K6 carried three / two one-element PyTorch launches per call (6.6 / 7.6 us of
    their 27 / 76 us) that the step never issues there."""
    from cabinet_amd import functional as Fh

    with Fh.batched_bn_counters():
        yield from _kernel_cases(batch, height, width, classes, extra)


def _kernel_cases(batch, height, width=None, classes=8, extra=False):
    """(Backward groups call the autograd Function's ``backward`` on the node their forward built: the same C-ABI calls as
    under autograd, issued from this thread on the current stream -- the autograd engine would run them on the forward's
    stream, outside a hipGraph capture.)

    Generator over the hand-written kernel groups at this workload's shapes: yields
    (name, launch closure, algorithmic FLOPs, algorithmic bytes, bound) one group at a time (operands of a finished group
    are freed before the next is built).  Algorithmic work per launch: SURVEY.md section 8(d) / BASELINE.md section 3.
    Consumers: kernel_rooflines() below (HIP-event timing) and tools/run_kernels.py (the same launches under rocprofv3
    for the PMC / kernel-trace passes).  ``height`` x ``width`` is the image (config 3: 1024 x 1024, config 5: 2048 x 1024),
    ``classes`` the OHEM heads' class count; ``extra`` adds the groups the model's step does not run (plain FFM, single-head
    OHEM).  Groups K7 .. K10 lie outside SURVEY.md section 8 (kernel_rooflines marks them)."""
    from cabinet_amd import functional as Fh

    dev = "cuda"
    H, W = height, (width or height)
    B, Kc, Vc = batch, 128, 128
    hl, wl = H // 32, W // 32
    n = hl * wl
    h, w = H // 8, W // 8
    P = h * w
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, Kc, n, generator=g).relu().to(dev)
    k = torch.randn(B, Kc, n, generator=g).to(dev)
    v = torch.randn(B, Vc, n, generator=g).to(dev)
    dctx = torch.randn(B, Vc, n, generator=g).to(dev)
    scale = Kc ** -0.5
    ctx, lse = Fh.attn_fwd_hip(q, k, v, scale)
    yield ("cab_attn_fwd (K1: affinity+softmax+aggregate)", lambda: Fh.attn_fwd_hip(q, k, v, scale), 2.0 * B * n * n * (Kc + Vc),
          4.0 * B * n * (2 * Kc + 2 * Vc) + 4.0 * B * n, "mfma")
    for code, name in ((Fh.PREC_BF16X6, "bf16x6"), (Fh.PREC_BF16X3, "bf16x3")):
        yield (f"cab_attn_fwd_{name} (K1 on the bf16 matrix pipe: operands split into bf16 pieces, pack pass + attention)",
               lambda c=code: Fh.attn_fwd_hip(q, k, v, scale, c), 2.0 * B * n * n * (Kc + Vc),
               4.0 * B * n * (2 * Kc + 2 * Vc) + 4.0 * B * n, "mfma-" + name)
    wpo = (torch.randn(2 * Vc, Vc, generator=g) * Vc ** -0.5).to(dev)
    if Fh.cab_attention_proj_supported(q, v, wpo):   # the form the model's CAB runs where the forward needs no key split
        qg = q.clone().requires_grad_(True)          # training: ctx is written for the backward
        yield ("cab_attn_proj_fwd (K1 + the CAB's output projection 128 -> 256 applied to the context tile in its epilogue)",
               lambda: Fh.cab_attention_proj(qg, k, v, wpo, scale), 2.0 * B * n * n * (Kc + Vc) + 2.0 * B * n * Vc * 2 * Vc,
               4.0 * B * n * (2 * Kc + 2 * Vc + 2 * Vc) + 4.0 * B * n, "mfma")
    yield ("cab_attn_bwd (K2: dk/dv + stored dS, dq = dS (K - mean K) as a small GEMM)", lambda: Fh.attn_bwd_hip(dctx, q, k, v, ctx, lse, scale), 2.0 * B * n * n * (3 * Kc + 2 * Vc),
          4.0 * B * n * (3 * Kc + 3 * Vc) * 2 + 8.0 * B * n, "mfma")

    Cs, Cc, Co, Cm = 128, 256, 256, 64
    fsp = torch.randn(B, Cs, h, w, generator=g).to(dev)
    fcp = torch.randn(B, Cc, h, w, generator=g).to(dev) if extra else None
    wb = (torch.randn(Co, Cs + Cc, generator=g) * 0.07).to(dev)
    w1 = (torch.randn(Cm, Co, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(Co, Cm, generator=g) * 0.1).to(dev)
    bw, bb = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    dout = torch.randn(B, Co, h, w, generator=g).to(dev)
    if extra:  # the plain FFM (reference signature forward(fsp, fcp)); CABiNet.forward runs the fused-upsample form below
        fwd = lambda: Fh.ffm_fwd_hip(fsp, fcp, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)  # noqa: E731
        o, z, mean, invstd, pooled, gate = fwd()
        # this build's pass structure: read fsp,fcp; write z; read z (stats); read z (pool); read z, write out
        yield ("ffm_fwd (K3: 1x1 GEMM + BN stats, pool, gate)", fwd, 2.0 * B * P * (Cs + Cc) * Co,
               4.0 * B * P * ((Cs + Cc) + 5 * Co), "mfma")
        bwd = lambda: Fh.ffm_bwd_hip(dout, fsp, fcp, wb, bw, bb, w1, w2, z, mean, invstd, pooled, gate, True)  # noqa: E731
        yield ("ffm_bwd (K4: reduce, dz, dX GEMM, dW split-K GEMM)", bwd, 4.0 * B * P * (Cs + Cc) * Co,
               4.0 * B * P * (2 * Co + 2 * Co + Co + Co + (Cs + Cc) + Co + (Cs + Cc)), "mfma")
        del o, z
    del fcp
    # ---- the form CABiNet.forward uses: bilinear upsample of `low` fused into the FFM.
    # conv and resize commute, so the Cc part runs at low resolution: executed GEMM work drops 2.7x and the
    # op becomes HBM-bound; algorithmic bytes = this build's pass structure.
    Pl = hl * wl
    low = torch.randn(B, Cc, hl, wl, generator=g).to(dev)
    upf = lambda: Fh.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)  # noqa: E731
    o, z, mean, invstd, pooled, gate = upf()
    fl_f = 2.0 * B * Co * (P * Cs + Pl * Cc)
    # THIS build's passes: read fsp, write z (the BatchNorm sums ride in the z kernel's epilogue since
    # round 4: no statistics pass), read z (pool), read z + write out (gate); low resolution: read low, write + read y_low
    by_f = 4.0 * B * (P * (Cs + 4 * Co) + Pl * (Cc + 2 * Co))
    yield ("ffm_up_fwd (K3': resize fused, conv commuted to low res)", upf, fl_f, by_f, "hbm")
    # the same operator as evaluate.py runs it (BatchNorm on running statistics, SURVEY 8(d) "eval-mode fwd"): this build keeps the
    # training pass structure (z written, pooled, gated), so the bytes are the training forward's; SURVEY's floor for a BN-folded
    # forward is read fsp + write feat + read feat + write out = 4 B P (Cs + 3 Co) (603 MB at config 3 with the materialised fcp)
    upe = lambda: Fh.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, False, 0.1, 1e-5)  # noqa: E731
    yield ("ffm_up_fwd_eval (K3' in eval mode: BatchNorm on running statistics; the path evaluate.py:77 consumes)", upe, fl_f, by_f, "hbm")
    for code, name in ((Fh.PREC_BF16X6, "bf16x6"), (Fh.PREC_BF16X3, "bf16x3")):
        yield (f"ffm_up_fwd_{name} (K3' with z = W_s fsp + U(W_c low) on the bf16 matrix pipe, operands split while staged)",
               lambda c=code: Fh.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5, c), fl_f, by_f, "hbm")
    upb = lambda: Fh.ffm_up_bwd_hip(dout, fsp, low, wb, bw, bb, w1, w2, z, mean, invstd, pooled, gate, True)  # noqa: E731
    fl_b = 4.0 * B * Co * (P * Cs + Pl * Cc)
    # THIS build's passes: the reduction reads dout and z once and writes the three adjoint fields; the
    # product kernel reads z and dout again (dz is formed while staged, never stored), reads fsp, writes dfsp; low resolution: the
    # three fields read, dz_low written and read, low read, dlow written; one 128 KB dW tile per workgroup (<= 256) written and read
    by_b = 4.0 * B * (P * (2 * Co + 2 * Co + Cs + Cs) + Pl * (3 * Co + 3 * Co + Co + Co + Cc + Cc)) + 2.0 * min(256, B * P // 32) * 4.0 * Co * Cs
    yield ("ffm_up_bwd (K4': reduce + three adjoint fields, dz_low, dfsp + dlow + dW with dz formed while staged, slab sum)", upb, fl_b, by_b, "hbm")
    del fsp, dout, o, z, low
    torch.cuda.empty_cache()

    if extra:  # K7 .. K10: backbone / spatial-branch operators of rounds 1-2, outside SURVEY.md section 8 (--all-kernels)
        yield from _backbone_cases(B, H, W, g, dev)

    # ---- f3: OHEM-CE fused with the final x8 upsample
    ncls = classes
    lowl = torch.randn(B, ncls, h, w, generator=g).to(dev)
    lab = torch.randint(0, ncls, (B, H, W), generator=g).to(dev)
    px = float(B * H * W)
    if extra:  # one head per launch (OhemCELoss.forward_upsampled); the step runs the paired form below
        yield ("ohem_up_fwd (f3: upsample + CE + OHEM partials)", lambda: Fh.ohem_up_fwd_hip(lowl, lab, (H, W), 0.7, 255),
               px * ncls * 12, px * 12 + 4.0 * lowl.numel(), "hbm")
        loss_px = Fh.ohem_up_fwd_hip(lowl, lab, (H, W), 0.7, 255)[0]
        yield ("ohem_up_bwd (f3: U^T[sel * (softmax - onehot)], separable)",
               lambda: Fh.ohem_up_bwd_hip(lowl, lab, loss_px, (H, W), 0.7, 255, 1e-6), px * ncls * 16,
               px * 12 + 4.0 * lowl.numel() + 8.0 * B * ncls * H * w, "hbm")
    lowl2 = torch.randn(B, ncls, h, w, generator=g).to(dev)
    yield ("ohem_up_pair_fwd (f3: BOTH loss heads per launch, label tile shared)",
           lambda: Fh.ohem_up_pair_fwd_hip(lowl, lowl2, lab, (H, W), 0.7, 255),
           2 * px * ncls * 12, px * (8 + 2 * 4) + 8.0 * lowl.numel(), "hbm")
    loss_px2 = Fh.ohem_up_pair_fwd_hip(lowl, lowl2, lab, (H, W), 0.7, 255)[0]
    yield ("ohem_up_pair_bwd (f3: both heads; x pass = whole source rows per wave, resize adjoint in registers; y pass)",
           lambda: Fh.ohem_up_pair_bwd_hip(lowl, lowl2, lab, loss_px2, (H, W), 0.7, 255, 1e-6),
           2 * px * ncls * 16, px * (8 + 2 * 4) + 8.0 * lowl.numel() + 16.0 * B * ncls * H * w, "hbm")
    del lowl2, loss_px2

    # ---- K5 / K6: the rest of the Context Aggregation Block at (B, 256, size/32, size/32)
    from cabinet_amd.models.cab import ContextAggregationBlock

    cab = ContextAggregationBlock(256, 128).to(dev).train()
    xc = torch.randn(B, 256, hl, wl, generator=g).to(dev).requires_grad_(True)
    gc = torch.randn(B, 256, hl, wl, generator=g).to(dev)
    elems = float(xc.numel())
    yl = cab.local_attn(xc)
    yield ("cab_local_fwd (K5: 3x DW3x3+BN+ReLU, gate, one kernel)", lambda: cab.local_attn(xc.detach()), elems * 3 * 22, 8.0 * elems, "hbm")
    yield ("cab_local_bwd (K5: chain recomputed in LDS)", lambda: Fh._CabLocal.backward(yl.grad_fn, gc), elems * 3 * 60, 12.0 * elems, "hbm")
    q3 = Fh.cab_qkv(xc, cab.global_attn)
    gq = [torch.randn_like(t) for t in q3]
    fl_q = 2.0 * B * n * (256 * 384 + 2 * 128 * 128)
    yield ("cab_qkv_fwd (K6: projections + BN + PSP, 3 launches)", lambda: Fh.cab_qkv(xc.detach(), cab.global_attn), fl_q, 4.0 * B * n * (256 + 3 * 128), "mfma")
    yield ("cab_qkv_bwd (K6: adjoint chain, 6 launches)", lambda: Fh._CabQkv.backward(q3[0].grad_fn, *gq), 2.0 * fl_q, 4.0 * B * n * (2 * 256 + 6 * 128), "mfma")
    del cab, xc, gc, yl, q3, gq
    torch.cuda.empty_cache()

    # ---- K11: the decoder's three plain 3x3 convolutions as fused Winograd F(2x2,3x3) kernels:
    # ab.conva (960 -> 256), the fusion head ab.b1 over cat([x, feat]) read through two pointers (960 + 256 -> 256), both at
    # size/32, and conv_out.conv (256 -> 256) at size/8.  `flops` are the EXECUTED matrix FLOPs -- 16 multiplications per
    # (k, c, 2x2 tile) = the direct convolution's 2*9*... / 2.25 -- so that `frac` is a share of the fp32 MFMA peak; the
    # direct-equivalent rate is 2.25x the reported TFLOP/s.  Bytes: inputs + outputs + weights once.
    for tag, C0, C1, hh, ww in (("conva", 960, 0, hl, wl), ("b1", 960, 256, hl, wl), ("out", 256, 0, h, w)):
        Co = 256
        x0 = torch.randn(B, C0, hh, ww, generator=g).to(dev)
        x1 = torch.randn(B, C1, hh, ww, generator=g).to(dev) if C1 else None
        wt = (torch.randn(Co, C0 + C1, 3, 3, generator=g) * 0.02).to(dev)
        dy = torch.randn(B, Co, hh, ww, generator=g).to(dev)
        fl = 2.0 * B * hh * ww * (C0 + C1) * Co * 9 / 2.25
        io = 4.0 * (B * hh * ww * (C0 + C1 + Co) + wt.numel())
        # as the step runs it: the epilogue also leaves the per-block (mean, M2) partials for the training-mode BatchNorm behind each
        # of the three convolutions (round 6: the replayed loop used to time the forward WITHOUT them -- tools/instep_cycles.sh found
        # conv_out's forward a tenth slower inside the step than in this loop for exactly that reason)
        part = Fh.conv3x3_bn_part(x0, Co)
        yield (f"conv3x3_{tag}_fwd (K11: filter transform + fused Winograd forward with the BatchNorm partials in its epilogue, {C0}{'+' + str(C1) if C1 else ''} -> {Co})",
               lambda: Fh.conv3x3_fwd_hip(x0, x1, wt, bn_part=part), fl, io, "mfma")
        yield (f"conv3x3_{tag}_bwd (K11: data gradient + weight gradient, both Winograd, ordered slab sum)",
               lambda: Fh.conv3x3_bwd_hip(dy, x0, x1, wt), 2.0 * fl, 2.0 * io, "mfma")
        del x0, x1, wt, dy, part
        torch.cuda.empty_cache()

    # ---- K12: BatchNorm -> ReLU -> 1x1 classifier as one streaming operator behind K11 (conv_out's tail,
    # cabinet.py:160-172; f2: the fusion head's b2 -> b3 -> b4, cabinet.py:90-92), statistics from K11's epilogue partials as in
    # the step.  Algorithmic bytes = the passes a training-mode BatchNorm cannot avoid when the activation is never written:
    # forward read z + write the logits; backward read z and dy twice + write dz.
    import torch.nn as nn

    for tag, hh, ww, has_bias in (("out", h, w, False), ("head", hl, wl, True)):
        Cz = 256
        x0 = torch.randn(B, Cz, hh, ww, generator=g).to(dev)
        wt = (torch.randn(Cz, Cz, 3, 3, generator=g) * 0.02).to(dev)
        part = Fh.conv3x3_bn_part(x0, Cz)
        zc = Fh.conv3x3(x0, wt, None, part).requires_grad_(True)
        bn = nn.BatchNorm2d(Cz).to(dev).train()
        cls = nn.Conv2d(Cz, ncls, 1, bias=has_bias).to(dev)
        gy = torch.randn(B, ncls, hh, ww, generator=g).to(dev)
        del x0, wt
        nz, ny = 4.0 * zc.numel(), 4.0 * gy.numel()
        yk = Fh.bn_relu_cls(zc, bn, cls, conv_part=part)
        if type(yk.grad_fn).__name__ == "_BnClsBackward":
            yield (f"bn_cls_{tag}_fwd (K12: BatchNorm finalize + table, then ONE pass: relu(bn(z)) in registers, {ncls} logits out)",
                   lambda: Fh.bn_relu_cls(zc.detach(), bn, cls, conv_part=part), 2.0 * zc.numel() * ncls, nz + ny, "hbm")
            yield (f"bn_cls_{tag}_bwd (K12: reduce with da formed on the fly, ordered finalize, dz pass; the activation's gradient never exists)",
                   lambda: Fh._BnCls.backward(yk.grad_fn, gy), 4.0 * zc.numel() * ncls, 3 * nz + 2 * ny, "hbm")
        del zc, yk, gy, part
        torch.cuda.empty_cache()



def _backbone_cases(B, H, W, g, dev):
    """K7 .. K10 (rounds 1-2): operators of the backbone and the spatial branch, outside SURVEY.md section 8; measured only with
    ``--all-kernels`` (the default run's time belongs to the section-8 groups)."""
    from cabinet_amd import functional as Fh

    # ---- K7: BatchNorm + activation at the largest plane of the model (sb.conv1 / features.2: 64 x size/2 x size/2);
    # algorithmic bytes = the passes a training-mode BatchNorm cannot avoid: fwd read x twice + write y,
    # bwd read dy and x twice + write dx
    Cb, hb, wb2 = 64, H // 2, W // 2
    xb = torch.randn(B, Cb, hb, wb2, generator=g).to(dev)
    gb = torch.randn(B, Cb, hb, wb2, generator=g).to(dev)
    bnw, bnb = torch.ones(Cb, device=dev), torch.zeros(Cb, device=dev)
    brm, brv = torch.zeros(Cb, device=dev), torch.ones(Cb, device=dev)
    fwd = lambda: Fh._BnAct.apply(xb, bnw, bnb, brm, brv, 2, True, 0.1, 1e-5)  # noqa: E731
    nbytes = 4.0 * xb.numel()
    yield ("bn_act_fwd (K7: BatchNorm + HardSwish, stats + apply)", fwd, 12.0 * xb.numel(), 3 * nbytes, "hbm")
    xg = xb.clone().requires_grad_(True)
    yb = Fh._BnAct.apply(xg, bnw, bnb, brm, brv, 2, True, 0.1, 1e-5)
    yield ("bn_act_bwd (K7: reduce + dx, pre-activation recomputed)", lambda: Fh._BnAct.backward(yb.grad_fn, gb), 30.0 * xb.numel(), 5 * nbytes, "hbm")
    del xb, gb, xg, yb
    torch.cuda.empty_cache()

    # ---- K8: depthwise 3x3 stride-2 convolution of features.2 (64 ch, size/2 -> size/4) with its BatchNorm + HardSwish
    # folded in; algorithmic bytes: fwd read z twice (statistics, convolution) + write y; bwd read dy, z (convolution),
    # read da, z + write dz (BatchNorm dx) + write da
    import torch.nn as nn

    conv = nn.Conv2d(Cb, Cb, 3, 2, 1, groups=Cb, bias=False).to(dev)
    bn = nn.BatchNorm2d(Cb).to(dev).train()
    zb = torch.randn(B, Cb, hb, wb2, generator=g).to(dev).requires_grad_(True)
    yb = Fh.bn_act_dwconv(zb, bn, "hardswish", conv)
    gy = torch.randn(yb.shape, generator=g).to(dev)
    nz, ny = 4.0 * zb.numel(), 4.0 * yb.numel()
    yield ("bn_dwconv_fwd (K8: BN stats + 3x3/2 depthwise conv with BN+HardSwish folded in)", lambda: Fh.bn_act_dwconv(zb.detach(), bn, "hardswish", conv), 28.0 * yb.numel(),
          2 * nz + ny, "hbm")
    yield ("bn_dwconv_bwd (K8: dx + dw + BN partial sums, then BN dx)", lambda: Fh._BnActDwConv.backward(yb.grad_fn, gy), 60.0 * yb.numel(), ny + 5 * nz, "hbm")
    del zb, yb, gy
    torch.cuda.empty_cache()

    # ---- K9: 7x7/2 stem convolution (3 -> 64) at the image size; 2*B*Ho*Wo*64*147 FLOP each way
    stem = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev)
    img = torch.randn(B, 3, H, W, generator=g).to(dev)
    ys = Fh.stem_conv(img, stem)
    gs = torch.randn(ys.shape, generator=g).to(dev)
    fl_s = 2.0 * ys.numel() * 147
    yield ("stem_conv_fwd (K9: 7x7/2, patch gather from LDS)", lambda: Fh.stem_conv(img, stem), fl_s, 4.0 * (img.numel() + ys.numel()), "mfma")
    yield ("stem_conv_wrw (K9: contraction over pixels, ordered slabs)", lambda: Fh._StemConv.backward(ys.grad_fn, gs), fl_s, 4.0 * (img.numel() + ys.numel()), "mfma")
    del img, ys, gs
    torch.cuda.empty_cache()

    # ---- K10: thin pointwise convolution 16 -> 64 on the size/2 plane (features.2 expansion)
    pw = nn.Conv2d(16, 64, 1, bias=False).to(dev)
    xp = torch.randn(B, 16, hb, wb2, generator=g).to(dev).requires_grad_(True)
    yp = Fh.pwconv(xp, pw)
    gp = torch.randn(yp.shape, generator=g).to(dev)
    nx, nyp = 4.0 * xp.numel(), 4.0 * yp.numel()
    yield ("pwconv_fwd (K10: streaming 1x1 conv 16->64)", lambda: Fh.pwconv(xp.detach(), pw), 2.0 * yp.numel() * 16, nx + nyp, "hbm")
    yield ("pwconv_bwd (K10: dx stream + wgrad slabs)", lambda: Fh._PwConv.backward(yp.grad_fn, gp), 4.0 * yp.numel() * 16, 2 * nyp + 2 * nx, "hbm")
    del xp, yp, gp
    torch.cuda.empty_cache()


_NOTES = {
    "cab_local_fwd": "one workgroup per channel, whole chain in LDS: bound by LDS latency / barriers, not HBM",
    "cab_local_bwd": "one workgroup per channel, chain recomputed in LDS: bound by LDS latency / barriers, not HBM",
    "cab_qkv_fwd": "3 dependent launches (round 5: BatchNorm statistics from the projection GEMM's epilogue) on 8192 positions: "
                   "latency / small-tile MFMA bound",
    "cab_qkv_bwd": "6 dependent launches on 8192 positions: latency / small-tile MFMA bound",
    "cab_attn_bwd": "traffic above the algorithmic bytes is the stored dS (33.5 MB written once, read by the dq product) "
                    ": it replaces recomputing S and dP for dq (4.3 GFLOP)",
    "conv3x3_conva_fwd": "executed FLOPs (Winograd: direct / 2.25); direct-equivalent rate = 2.25 x tflops",
    "conv3x3_conva_bwd": "executed FLOPs (Winograd: direct / 2.25); direct-equivalent rate = 2.25 x tflops",
    "conv3x3_b1_fwd": "executed FLOPs (Winograd: direct / 2.25); the concat of its two inputs is never materialised",
    "conv3x3_b1_bwd": "executed FLOPs (Winograd: direct / 2.25); direct-equivalent rate = 2.25 x tflops",
    "conv3x3_out_fwd": "executed FLOPs (Winograd: direct / 2.25); direct-equivalent rate = 2.25 x tflops",
    "conv3x3_out_bwd": "executed FLOPs (Winograd: direct / 2.25); direct-equivalent rate = 2.25 x tflops",
    "bn_cls_out_fwd": "one pass over z (the 3x3's output); a = relu(bn(z)) lives in registers; per-channel scalars by scalar loads",
    "bn_cls_out_bwd": "two passes over z; da = W^T dy is formed per element (8 FMAs), dW and the BatchNorm sums in the same pass",
    "ohem_up_fwd": "exp/log and VALU bound (8 exps per pixel), not HBM",
    "ohem_up_pair_fwd": "what the step runs: both heads per launch; exp/log and VALU bound (16 exps per pixel, one log-sum-exp "
                        "shift per source interval), not HBM",
    "ohem_up_pair_bwd": "what the step runs: both heads per launch; x pass VALU bound (fma + sub + exp + 2 fma per pixel and "
                        "class), y pass HBM bound (T written once, read once)",
    "ffm_up_bwd": "bound by its two 8.6 GFLOP products on the 1/16-rate fp32 matrix pipe (109 us at peak) behind ONE HBM pass "
                  "(the reduction, which also takes the resize adjoint of three coefficient-free fields); dz is never stored, so "
                  "the measured traffic is below the algorithmic bytes of SURVEY 8(d)'s pass structure",
    "ffm_up_fwd": "persistent z product (fp32 MFMA) with the BatchNorm sums in its epilogue + two HBM passes over z (pool, gate) "
                  "that BatchNorm's batch statistics force",
    "ohem_up_bwd": "exp and VALU bound (softmax recomputed per pixel), not HBM",
    "bn_dwconv_fwd": "the depthwise stencil is VALU bound; the BatchNorm statistics pass is HBM bound",
    "bn_dwconv_bwd": "the depthwise stencil backward is VALU bound; the BatchNorm dx pass is HBM bound",
}


# kernel groups outside SURVEY.md section 8 (widening of earlier rounds: backbone / spatial-branch operators)
_OUTSIDE_S8 = ("bn_act_", "bn_dwconv_", "stem_conv_", "pwconv_")
# groups whose FLOP count is matrix (MFMA) work: priced against the matrix roof AND the HBM roof
_MATRIX_GROUPS = ("cab_attn", "ffm_", "cab_qkv", "conv3x3_", "stem_conv")


def kernel_rooflines(batch, height, width, classes, iters, extra=False):
    """Per hand-written kernel group: launch duration from HIP events on the stream the kernels run on, achieved
    TFLOP/s / GB/s against the gfx950 peaks, and the measured HBM traffic when a PMC profile of THIS build exists."""
    out = []
    traffic, traffic_src = load_traffic(batch, height, width, classes)
    for name, fn, flops, bytes_, bound in kernel_cases(batch, height, width, classes, extra):
        if os.environ.get("CABINET_BENCH_VERBOSE") == "1":
            print(f"[bench] timing {name}", file=sys.stderr, flush=True)
        ms = time_kernel(fn, iters)
        tf = flops / (ms * 1e-3) / 1e12
        gbs = bytes_ / (ms * 1e-3) / 1e9
        key = name.split(" ")[0]
        frac_hbm = gbs / PEAK_HBM_GBS
        if bound in MFMA_PEAKS or key.startswith(_MATRIX_GROUPS):
            # a group that runs matrix products is priced against BOTH roofs on this build's own pass structure: `bound` is the
            # roof that takes longer at its peak (the FFM groups were printed against HBM on SURVEY's
            # five-pass byte count although their longer roof -- backward -- is the fp32 matrix pipe)
            peak, peak_is = MFMA_PEAKS[bound if bound in MFMA_PEAKS else "mfma"]
            frac_mfma = tf / peak
            if frac_mfma >= frac_hbm:
                r = dict(bound="mfma", achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(frac_mfma, 4),
                         peak_is=peak_is)
            else:
                r = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(frac_hbm, 4))
            r.update(frac_mfma=round(frac_mfma, 4), frac_hbm=round(frac_hbm, 4))
        else:
            r = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(frac_hbm, 4))
        r["traffic"] = traffic.get(key)
        r["traffic_source"] = traffic_src
        if r["traffic"] is not None and r["traffic"] < 0.95 * bytes_:
            r["bytes_formula_suspect"] = "measured traffic is below the algorithmic bytes: the byte count above overstates this build's passes"
        if key in _NOTES:
            r["note"] = _NOTES[key]
        r["scope"] = "outside SURVEY section 8 (widening)" if key.startswith(_OUTSIDE_S8) else "SURVEY section 8"
        r.update(kernel=name, ms_per_launch=round(ms, 4), algorithmic_gflop=round(flops / 1e9, 3),
                 algorithmic_mbytes=round(bytes_ / 1e6, 1), tflops=round(tf, 2), gbytes_per_s=round(gbs, 1))
        out.append(r)
    return out


def time_eval_forward(net, im, steps, warmup):
    """The path the reference's evaluator consumes (evaluate.py:77 `model(crop)[0]`; ema.py:44 `deepcopy(model).eval()`): forward only,
    `.eval()` (BatchNorm on running statistics), `torch.no_grad()`, full-resolution logits.  -> dict with the eager time (what the
    unmodified script pays, host dispatch included) and the time of the same forward replayed from ONE hipGraph."""
    import copy

    net_e = copy.deepcopy(net).eval()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    with torch.no_grad():
        for _ in range(max(2, warmup)):
            out = net_e(im)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = net_e(im)[0]
        torch.cuda.synchronize()
        res["eager_ms"] = (time.perf_counter() - t0) / steps * 1e3
        try:
            s_im = im.clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                s_out = net_e(s_im)[0]
            graph.replay()
            torch.cuda.synchronize()
            start.record()
            for _ in range(steps):
                graph.replay()
            stop.record()
            torch.cuda.synchronize()
            res["graph_ms"] = start.elapsed_time(stop) / steps
            # not bitwise: bench.py leaves MIOpen its default solvers, and some of the backbone's convolutions sum with atomics
            res["graph_vs_eager_rel"] = float((s_out.double() - out.double()).norm() / out.double().norm().clamp_min(1e-30))
            del graph
        except Exception as e:  # noqa: BLE001
            res["graph_ms"] = None
            res["graph_error"] = f"{type(e).__name__}: {e}"
    res["finite"] = bool(torch.isfinite(out).all())
    res["out_shape"] = list(out.shape)
    return res


def eval_line(args, net, im, dev):
    """`--eval`: one JSON line for the forward-only path (the `no_grad` / `.eval()` path evaluate.py:77 uses)."""
    H, W = args.height, args.width
    net.train()
    with torch.no_grad():   # running statistics of a freshly initialised model describe nothing: let BatchNorm see this input first
        for _ in range(12):
            net.forward_lowres(im)
    torch.cuda.synchronize()
    r = time_eval_forward(net, im, args.steps, args.warmup)
    ms = r["graph_ms"] if r.get("graph_ms") else r["eager_ms"]
    ks = kernel_rooflines(args.batch, H, W, args.classes, args.kernel_iters, False) if not args.no_kernel_roofline else []
    fe = [k for k in ks if k["kernel"].startswith("ffm_up_fwd_eval")]
    k1 = [k for k in ks if k["kernel"].startswith("cab_attn_fwd ")]
    out = {
        "metric": f"{H}x{W} images/sec forward only, eval mode (CABiNet-MobileNetV3-{args.mode.capitalize()})",
        "value": round(args.batch / (ms * 1e-3), 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"CABiNet-MobileNetV3-{args.mode.capitalize()}, {args.batch}x3x{H}x{W} synthetic, {args.classes} classes, "
                                "forward only: .eval() + torch.no_grad(), BatchNorm on running statistics (populated by 12 training-mode "
                                "forwards of this input), full-resolution logits model(x)[0] as evaluate.py:77 consumes them"),
                   "host_path": "one captured hipGraph per forward" if r.get("graph_ms") else "eager enqueue"},
        "eager": {"ms_per_step": round(r["eager_ms"], 3), "value": round(args.batch / (r["eager_ms"] * 1e-3), 3),
                  "what": "the same forward enqueued eagerly (what the reference's unmodified evaluate.py pays: host dispatch included)"},
        "forward": r,
        "context_only": {"reference_published_inference_fps": {"76 FPS": "Cityscapes, RTX 2080 Ti", "8 FPS": "Jetson Xavier NX",
                                                               "15 FPS": "UAVid, hardware not stated"},
                         "source": "reference .github/CHANGELOG.md:61-63 (other hardware, other input sizes: not a baseline for this line)"},
    }
    if fe:
        out["roofline"] = {k: fe[0][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_mfma", "frac_hbm") if k in fe[0]}
        out["roofline"].update(kernel=fe[0]["kernel"], ms_per_launch=fe[0]["ms_per_launch"], algorithmic_mbytes=fe[0]["algorithmic_mbytes"])
    if k1:
        out["cab_attn_fwd"] = {k: k1[0][k] for k in ("kernel", "ms_per_launch", "bound", "achieved", "peak", "unit", "frac")}
    return out


def main():
    args = parse()
    t_start = time.perf_counter()
    from cabinet_amd import ddp as ddp_mod
    from cabinet_amd.ddp import BucketedGradReducer, init_distributed
    from cabinet_amd.train import GraphedDDPStep, GraphedTrainStep, TrainStep, build_model, make_criteria, synthetic_batch

    # With several ranks, init_distributed() pins the process to its GPU's NUMA node BEFORE the first GPU call so that the
    # runtime's threads inherit the mask: nothing may touch torch.cuda before it (torch.cuda.device_count() falls back to
    # hipGetDeviceCount -- which starts the runtime's threads -- when amdsmi is not importable), so "is there a GPU"
    # is answered from the kernel driver's device node
    if not os.path.exists("/dev/kfd"):
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # the library that gets timed must be the one the sources in this tree describe (and the one the PMC profiles were digest-checked
    # against): cabinet_amd/build.py stamps the library with the digest of csrc/ + include/ at build time (VERDICT r05 hygiene)
    from cabinet_amd import build as _build
    if not _build.is_fresh():
        # never time a stale library: local rank 0 rebuilds it from the sources in this tree, the other ranks wait for its stamp
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            print("[bench] libcabinet_hip.so is stale or missing: rebuilding it from the sources in this tree", file=sys.stderr, flush=True)
            _build.build(verbose=False)   # raises when hipcc is missing or a source does not compile
        else:
            t_wait = time.perf_counter()
            while not _build.is_fresh():
                if time.perf_counter() - t_wait > 1200:
                    raise SystemExit("bench.py: cabinet_amd/libcabinet_hip.so is stale and local rank 0 did not rebuild it")
                time.sleep(2.0)
    if args.kernels_only:
        for r in kernel_rooflines(args.batch, args.height, args.width, args.classes, args.kernel_iters, args.all_kernels):
            print(f"{r['kernel'][:52]:52s} {r['ms_per_launch'] * 1e3:9.1f} us  {r['tflops']:7.2f} TF/s  "
                  f"frac {r['frac']:.3f}  {r['gbytes_per_s']:8.1f} GB/s")
        return
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # NOTE: torch.backends.cudnn.benchmark stays False: MIOpen's exhaustive find mode compiles and times
    # every solver for ~150 conv problems on a fresh box (tens of minutes); immediate mode is used.

    net = build_model(args.mode, n_classes=args.classes, device=dev, seed=0, gamma=0.5).train()
    ddp = world > 1 or os.environ.get("CABINET_FORCE_DDP") == "1"
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9,
                          weight_decay=5e-4)
    graphed = not args.no_graph
    crit = make_criteria(args.batch, args.height, args.width, dev)
    reducer = None
    if graphed and ddp:
        # hipGraph segments with the bucket all-reduces issued between them: the decoder's 23 MB reduce while the encoders
        # back-propagate (cabinet_amd/train.py::GraphedDDPStep); no per-kernel Python on any rank's host
        step = reducer = GraphedDDPStep(net, crit, optimizer=opt, always_reduce=True)
    elif graphed:  # the step replayed from two hipGraphs around its one host read (cabinet_amd/train.py)
        step = GraphedTrainStep(net, crit, optimizer=opt)
    else:
        reducer = BucketedGradReducer(net, always_reduce=True) if ddp else None
        step = TrainStep(net, crit, reducer=reducer, optimizer=opt)
    im, lb = synthetic_batch(args.batch, args.height, args.width, args.classes, dev, seed=1 + rank)
    if args.eval:
        if world != 1:
            raise SystemExit("--eval times one GPU")
        print(json.dumps(eval_line(args, net, im, dev)), flush=True)
        return

    def sync():
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"model on {dev}, world={world}; warm-up {args.warmup} step(s)")
    for i in range(args.warmup):
        step(im, lb)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done (MIOpen kernels compiled / loaded)")
    sync()
    log(f"timing {args.steps} steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(im, lb)
    sync()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if ddp:
        # every rank's own wall time of the K steps (the line's `value` uses the maximum): what a reader needs to see a straggler
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        per_rank_ms = [float(x) / args.steps * 1e3 for x in every]
        dt = max(float(x) for x in every)
    final_loss = float(loss)
    log(f"timed region {dt:.3f}s -> {world * args.batch * args.steps / dt:.2f} images/s")
    # SURVEY.md section 8(d) words the metric as forward + 2x OHEM-CE + backward; `value` above also contains the gradient
    # all-reduce and the SGD step (conservative).  The same K steps without the optimizer, reported next to it:
    def set_optimizer(o):  # the optimizer step is not part of any captured graph: same step, segment skipped
        step.optimizer = o
        if hasattr(step, "opt_seg"):
            step.opt_seg.optimizer = o
        if hasattr(step, "eager"):
            step.eager.optimizer = o

    set_optimizer(None)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(im, lb)
    sync()
    dt_nopt = time.perf_counter() - t0
    set_optimizer(opt)
    if ddp:
        t = torch.tensor([dt_nopt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt_nopt = float(t)

    result = None
    if rank == 0:
        images = world * args.batch * args.steps
        H, W = args.height, args.width
        headline = (H, W, args.mode) == (1024, 1024, "large")
        config5 = (args.batch, H, W, args.classes, args.mode, world) == (2, 2048, 1024, 19, "large", 1)
        which = "configs[4]" if config5 else f"configs[{2 if world == 1 else 3}]"
        result = {
            "metric": "1024x1024 images/sec fwd+bwd (CABiNet-MobileNetV3-Large)" if headline
            else f"{H}x{W} images/sec fwd+bwd (CABiNet-MobileNetV3-{args.mode.capitalize()})",
            "value": round(images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"BASELINE {which}: CABiNet-MobileNetV3-{args.mode.capitalize()}, "
                             f"{args.batch}x3x{H}x{W} synthetic {args.classes}-class per GPU, "
                             "fwd + 2x OhemCE + bwd + grad all-reduce + SGD step, HIP CAB/FFM kernels, fp32, "
                             "random-init (model seed 0, gamma=0.5)"),
                "per_gpu_batch": args.batch, "global_batch": args.batch * world, "image_size": [H, W],
                "n_classes": args.classes, "parallelism": f"dp{world}",
                "grad_buckets_mb": [round(x, 2) for x in reducer.bucket_megabytes] if reducer else None,
                "host_path": ("hipGraph segments, bucket all-reduces between them (GraphedDDPStep)" if graphed and ddp
                              else "two captured hipGraphs per step around the one OHEM read-back (GraphedTrainStep)"
                              if graphed else "eager enqueue (TrainStep)"),
                "dist_backend": torch.distributed.get_backend() if ddp else None,
                "ranks_in_process_group": torch.distributed.get_world_size() if ddp else None,
                "per_rank_ms_per_step": ({"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3),
                                          "all": [round(x, 3) for x in per_rank_ms]} if per_rank_ms else None),
                "ddp_schedule": getattr(step, "schedule", None) if ddp else None,
                "cpu_affinity": ddp_mod.AFFINITY if world > 1 else None,
            },
            "library": {"source_digest": _build.source_digest()[:16], "fresh": _build.is_fresh(),
                        "what": "sha256 of cabinet_amd/csrc/* + include/cabinet_hip.h; the loaded libcabinet_hip.so carries this stamp"},
            "final_loss": round(final_loss, 5),
            "fwd_loss_bwd_only": {"value": round(images / dt_nopt, 3), "unit": "images/s",
                                  "ms_per_step": round(dt_nopt / args.steps * 1e3, 3),
                                  "what": "the same K steps without the SGD step (forward + 2x OHEM-CE + backward"
                                          + (" + gradient all-reduce)" if ddp else ")")},
        }
    if rank == 0 and world == 1 and not args.no_eval_forward:
        # the forward-only path of evaluate.py:77 on the model that was just timed (its BatchNorm running statistics have seen
        # warmup + steps batches): .eval() + no_grad, full-resolution logits; `python bench.py --eval` prints
        # this as a line of its own
        r = time_eval_forward(net, im, max(5, args.steps // 2), 2)
        ms = r["graph_ms"] if r.get("graph_ms") else r["eager_ms"]
        result["eval_forward"] = {"value": round(args.batch / (ms * 1e-3), 3), "unit": "images/s", "ms_per_step": round(ms, 3),
                                  "eager_ms_per_step": round(r["eager_ms"], 3), "graph_vs_eager_rel": r.get("graph_vs_eager_rel"),
                                  "what": "forward only, .eval() + torch.no_grad(), BatchNorm on running statistics, model(x)[0] at "
                                          "full resolution (evaluate.py:77), one hipGraph per forward; eager_ms_per_step = the "
                                          "same forward through the Python dispatch"}
        log(f"eval forward: {ms:.2f} ms/batch graphed, {r['eager_ms']:.2f} eager")
    # ---- per-kernel rooflines and CPU baseline: rank 0, outside the timed region -----------------
    if rank == 0 and not args.no_kernel_roofline:
        del step, opt
        torch.cuda.empty_cache()
        ks = kernel_rooflines(args.batch, args.height, args.width, args.classes, args.kernel_iters, args.all_kernels)
        log("kernel rooflines measured")
        result["kernels"] = ks
        # what the SURVEY section 8 path costs inside the step: the groups the step launches (not the plain K1 where the fused-projection
        # form runs, not the split-bf16 variants, not --all-kernels extras), once each -- next to ms_per_step, so that `value` is not
        # read as a statement about these kernels alone (section 8 is ~18 % of the step; the backbone / spatial branch on
        # stock PyTorch-ROCm operators and K7-K10 are the rest)
        in_step = [r for r in ks if r["scope"] == "SURVEY section 8" and "_bf16x" not in r["kernel"].split(" ")[0]
                   and not r["kernel"].startswith(("ffm_fwd ", "ffm_bwd ", "ohem_up_fwd ", "ohem_up_bwd ", "ffm_up_fwd_eval"))]
        if any(r["kernel"].startswith("cab_attn_proj_fwd") for r in in_step):
            in_step = [r for r in in_step if not r["kernel"].startswith("cab_attn_fwd ")]
        hot = sum(r["ms_per_launch"] for r in in_step)
        result["hot_path_ms_per_step"] = {"value": round(hot, 3), "share_of_step": round(hot / result["ms_per_step"], 3),
                                          "groups": [r["kernel"].split(" ")[0] for r in in_step],
                                          "what": "sum of the SURVEY section 8 kernel groups the step launches (standalone graph-replay "
                                                  "timings, once each); the rest of ms_per_step is the MobileNetV3 backbone and the "
                                                  "spatial branch, which north_star leaves on PyTorch-ROCm"}
        # `roofline` = the CAB affinity+aggregate kernel the north_star sets its MFMA target on; next to it the
        # longest-running hand-written group INSIDE SURVEY section 8 (the hot path's dominant cost) and, for context, the
        # longest group overall (an out-of-scope backbone operator)
        k1 = ks[0]
        s8 = [r for r in ks if r["scope"] == "SURVEY section 8" and "_bf16x" not in r["kernel"].split(" ")[0]
              and not r["kernel"].startswith("ffm_up_fwd_eval")]
        dom8 = max(s8, key=lambda r: r["ms_per_launch"])
        dom = max(ks, key=lambda r: r["ms_per_launch"])
        result["roofline"] = {k: k1[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        result["roofline"].update(kernel=k1["kernel"], ms_per_launch=k1["ms_per_launch"],
                                  algorithmic_bytes=round(k1["algorithmic_mbytes"] * 1e6),
                                  traffic_source=k1["traffic_source"],
                                  peak_is="dense fp32 MFMA (v_mfma_f32_32x32x2_f32), not bf16",
                                  # `frac` of that group = algorithmic_gflop (bound "mfma") or algorithmic_mbytes (bound "hbm") of
                                  # THIS entry over its time and peak: the quantity it divides travels with it
                                  longest_section8_group={k: dom8[k] for k in ("kernel", "ms_per_launch", "bound", "achieved",
                                                                              "peak", "unit", "frac", "traffic",
                                                                              "algorithmic_gflop", "algorithmic_mbytes")},
                                  longest_kernel_group=dom["kernel"], longest_kernel_frac=dom["frac"],
                                  longest_kernel_scope=dom["scope"])
        # round 5: where the forward needs no key split the model's CAB runs K1 with the output projection in its epilogue; that
        # instantiation's own entry (its FLOPs include the projection's 2 B n Vc Co) travels next to the plain kernel's
        kp = [r for r in ks if r["kernel"].startswith("cab_attn_proj_fwd")]
        if kp:
            result["roofline"]["as_run_in_the_step"] = {k: kp[0][k] for k in ("kernel", "ms_per_launch", "bound", "achieved", "peak",
                                                                             "unit", "frac", "traffic", "algorithmic_gflop")}
    if ddp:
        torch.distributed.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import model_ref
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        cb = model_ref.time_cpu_baseline(sd, args.mode, args.cpu_batch, (args.height, args.width), args.classes)
        # the timed HIP model, on the very sample the CPU leg just ran, from the same weights: the model that was timed is
        # checked in the run that timed it
        import copy

        net_chk = copy.deepcopy(net).train()
        xs, ls = model_ref.baseline_sample(args.cpu_batch, (args.height, args.width), args.classes)
        chk = TrainStep(net_chk, make_criteria(args.cpu_batch, args.height, args.width, dev))
        loss_gpu = float(chk(xs.to(dev), ls.to(dev)))
        del net_chk, chk
        result["cpu_baseline"] = {
            "value": round(cb["value"], 4), "unit": "images/s", "cores": cb["cores"], "kind": "port",
            "sample": (f"oracle/model_ref.py (PyTorch-CPU fp32 restatement of the reference, pinned to reference "
                       f"vectors): B={args.cpu_batch} {args.height}x{args.width}, 1 warm-up + {cb['timed_steps']} timed steps "
                       f"({cb['seconds']:.1f} s) of fwd + 2x OhemCE + bwd"),
            "seconds_per_step": round(cb["seconds_per_step"], 3),
            "loss_cpu": round(cb["loss"], 6), "loss_gpu_same_sample": round(loss_gpu, 6),
            "loss_rel_diff": abs(loss_gpu - cb["loss"]) / abs(cb["loss"]),
        }
    # RCCL (NCCL_DEBUG=VERSION on these boxes) writes its version banner to C stdout, which is only flushed at exit:
    # every rank pushes it out now, so that rank 0's JSON line is the LAST line of the job's stdout
    import ctypes

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if ddp:
        torch.distributed.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if ddp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
