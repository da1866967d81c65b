"""Does a model-level step depend on what ran before it in the process?  (round 6: `test_model_vs_oracle_logits_and_grads[large-2-512-19]`
failed behind tests/test_gpu_ffm.py and passed inside the full suite on the same box, MIOpen pinned to deterministic solvers.)

One step of the HIP model (Large, 2x3x512x512, 19 classes by default; the test's recipe) in THIS process after an optional
precondition, with a digest of every captured tensor of tests/insitu.py's hooks and of every gradient:

  --pre none      fresh process
  --pre nan       the caching allocator's free blocks filled with NaN first (reads of never-written workspace show up as NaN)
  --pre junk      ... filled with N(0, 1) values (plausible stale data)
  --pre small     the Small 4x512x512 step of the same test runs first (what precedes it in tests/test_gpu_model.py)
  --pre ffm       a few FFM / conv3x3 calls of other shapes run first

    python tools/diag_order_dependence.py --pre none --out gpurun_out/order_none.json
"""
import argparse
import copy
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import conftest  # noqa: E402,F401


def digest(t):
    return hashlib.sha256(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:10]


def fill_free(kind, gb=24):
    mk = (lambda n: torch.full((n,), float("nan"), device="cuda")) if kind == "nan" else (lambda n: torch.randn(n, device="cuda"))
    blocks = [mk(gb * (1 << 28) // 8) for _ in range(8)]
    blocks += [mk(1 << k) for k in range(8, 25) for _ in range(6)]
    torch.cuda.synchronize()
    del blocks


def step(mode, batch, size, ncls, capture=True):
    from cabinet_amd.train import build_model, make_criteria, synthetic_batch
    from insitu import instrument

    net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False)
    im, lb = synthetic_batch(batch, size, size, ncls, "cpu", seed=1)
    net = net.cuda().train()
    cap = instrument(net) if capture else {}
    if capture:   # the spatial branch layer by layer (stock convolutions behind K9 / K7): where does a difference start?
        for name in ("conv1", "conv2", "conv3", "conv_out"):
            blk = getattr(net.sb, name)
            blk.conv.register_forward_hook(lambda m, a, o, n=name: cap.__setitem__("sb." + n + ".conv", o.detach().clone()))
            blk.register_forward_hook(lambda m, a, o, n=name: cap.__setitem__("sb." + n, o.detach().clone()))
    crit = make_criteria(batch, size, size, "cuda")
    out, out16 = net(im.cuda())
    loss = crit[0](out, lb.cuda()) + crit[1](out16, lb.cuda())
    loss.backward()
    torch.cuda.synchronize()
    res = {"out": digest(out), "out16": digest(out16), "loss": float(loss)}
    for k in sorted(cap):
        if torch.is_tensor(cap[k]):
            res["cap." + k] = digest(cap[k])
    for k, p in net.named_parameters():
        if p.grad is not None:
            res["grad." + k] = digest(p.grad)
    res["nan_grads"] = [k for k, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pre", default="none")
    ap.add_argument("--mode", default="large")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.backends.cudnn.deterministic = True
    if a.pre in ("nan", "junk"):
        fill_free(a.pre)
    elif a.pre == "small":
        step("small", 4, 512, 8)
    elif a.pre == "ffm":
        from cabinet_amd.functional import conv3x3_bwd_hip, conv3x3_fwd_hip, ffm_fused_upsampled
        g = torch.Generator().manual_seed(0)
        for (B, h, w) in ((2, 32, 64), (3, 24, 64), (1, 128, 128)):
            bn = torch.nn.BatchNorm2d(256).cuda().train()
            t = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in
                 ((B, 128, h, w), (B, 256, h // 4, w // 4), (256, 384, 1, 1), (64, 256, 1, 1), (256, 64, 1, 1))]
            ffm_fused_upsampled(*t[:3], bn, *t[3:]).sum().backward()
        x = torch.randn(2, 64, 33, 20, generator=g).cuda()
        wt = torch.randn(128, 64, 3, 3, generator=g).cuda()
        y = conv3x3_fwd_hip(x, None, wt)
        conv3x3_bwd_hip(torch.randn_like(y), x, None, wt)
        torch.cuda.synchronize()
    res = step(a.mode, a.batch, a.size, a.classes)
    res2 = step(a.mode, a.batch, a.size, a.classes)   # and once more in the same process
    res["second_run_differs_in"] = [k for k in res2 if k != "nan_grads" and res2[k] != res.get(k)]
    res["pre"] = a.pre
    print(json.dumps({k: v for k, v in res.items() if not k.startswith(("grad.", "cap."))}, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
