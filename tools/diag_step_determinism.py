"""Run-to-run spread of the step behind tests/test_gpu_model.py::test_model_known_answers_from_reference.

The step (fwd + 2x OHEM-CE + bwd, Large 2x3x256x256, gamma = 0.5: the KAT3 recipe of tests/golden/kat_model.json) is repeated from
ONE state_dict; per repeat a digest of the logits (is the FORWARD reproducible? ReLU masks are decided there) and every gradient
tensor's norm.  Reported: which tensors move between repeats and by how much, against the 1e-3 the test allows around the reference.

  --poison     before each repeat fill the caching allocator's free blocks with NaN (an operator that reads workspace it never
               wrote then produces NaN instead of depending on what ran before)
  --deterministic   torch.backends.cudnn.deterministic = True (MIOpen: MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC on every descriptor)

    python tools/diag_step_determinism.py [--mode large] [--repeats 6] [--poison] [--deterministic]
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
import conftest  # noqa: E402,F401  (the private MIOpen database copy the test session uses)


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


def poison(gb=8):
    """Fill what the caching allocator will hand out next with NaN: one large block and many small ones, then free them."""
    blocks = [torch.full((gb * (1 << 28),), float("nan"), device="cuda")]
    blocks += [torch.full((1 << k,), float("nan"), device="cuda") for k in range(8, 24) for _ in range(4)]
    torch.cuda.synchronize()
    del blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="large")
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--size", type=int, default=0, help="square input size instead of the KAT's 256")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.deterministic:
        torch.backends.cudnn.deterministic = True
    from cabinet_amd.train import TrainStep, build_model, make_criteria

    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_model.json")))[a.mode]
    t = kat["train"]
    shape = list(t["shape"])
    if a.size:
        shape[2] = shape[3] = a.size
    net = build_model(a.mode, n_classes=8, seed=kat["model_seed"], freeze_unused=False, device="cuda")
    with torch.no_grad():
        net.ab.a2block.gamma.fill_(t["gamma"])
    net.train()
    sd0 = copy.deepcopy(net.state_dict())
    torch.manual_seed(t["data_seed"])
    x = torch.randn(*shape).cuda()
    lb = torch.randint(0, 8, (shape[0], shape[2], shape[3])).cuda()
    runs = []
    for r in range(a.repeats):
        net.load_state_dict(sd0)
        if a.poison:
            poison()
        step = TrainStep(net, make_criteria(shape[0], shape[2], shape[3], "cuda"))
        with torch.no_grad():
            net.eval()
            net.train()
        low, low16 = net.forward_lowres(x)
        fw = (digest(low), digest(low16))
        loss = step(x, lb)
        torch.cuda.synchronize()
        g = {k: p.grad.detach().double().clone() for k, p in net.named_parameters() if p.grad is not None}
        runs.append(dict(fwd=fw, loss=float(loss), grads=g, nan=[k for k, v in g.items() if not torch.isfinite(v).all()]))
    base = runs[0]
    res = dict(mode=a.mode, shape=shape, poison=a.poison, deterministic=a.deterministic,
               forward_digests=[r["fwd"] for r in runs], losses=[r["loss"] for r in runs],
               nan_tensors=sorted({k for r in runs for k in r["nan"]}))
    spread = {}
    for k, v in base["grads"].items():
        d = max(float((r["grads"][k] - v).norm() / v.norm().clamp_min(1e-300)) for r in runs[1:])
        nd = max(abs(float(r["grads"][k].norm() - v.norm())) / float(v.norm().clamp_min(1e-300)) for r in runs[1:])
        if d > 0:
            spread[k] = (d, nd)
    res["tensors_that_move"] = len(spread)
    res["tensors"] = len(base["grads"])
    res["worst"] = sorted(((d, nd, k) for k, (d, nd) in spread.items()), reverse=True)[:12]
    if "grad_norms" in t and not a.size:
        gn = {k: float(v.norm()) for k, v in base["grads"].items()}
        res["past_1e-3_of_reference_norm"] = {k: (gn[k], w) for k, w in t["grad_norms"].items() if abs(gn[k] - w) > 1e-3 * w + 1e-7}
    print(json.dumps(res, indent=1, default=str))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1, default=str)


if __name__ == "__main__":
    main()
