"""Time K1/K2 at one shape: python tools/time_attn.py B Kc Vc n"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.functional import attn_bwd_hip, attn_fwd_hip

B, Kc, Vc, n = (int(v) for v in sys.argv[1:5])
g = torch.Generator().manual_seed(0)
q = torch.randn(B, Kc, n, generator=g).relu().cuda()
k = torch.randn(B, Kc, n, generator=g).cuda()
v = torch.randn(B, Vc, n, generator=g).cuda()
d = torch.randn(B, Vc, n, generator=g).cuda()
sc = Kc ** -0.5
ctx, lse = attn_fwd_hip(q, k, v, sc)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tf = timeit(lambda: attn_fwd_hip(q, k, v, sc))
tb = timeit(lambda: attn_bwd_hip(d, q, k, v, ctx, lse, sc))
ff = 2.0 * B * n * n * (Kc + Vc)
fb = 2.0 * B * n * n * (3 * Kc + 2 * Vc)
print(f"B={B} Kc={Kc} Vc={Vc} n={n}: fwd {tf:.1f} us ({ff / tf / 1e6:.1f} TF/s)  bwd {tb:.1f} us ({fb / tb / 1e6:.1f} TF/s)")
from cabinet_amd import _lib  # noqa: E402

for prec, name in ((1, "bf16x3"), (2, "bf16x6")):
    if _lib.load().cabinet_cab_attn_precision_supported(Kc, Vc, prec):
        t = timeit(lambda: attn_fwd_hip(q, k, v, sc, prec))
        c2, _ = attn_fwd_hip(q, k, v, sc, prec)
        print(f"   fwd {name}: {t:.1f} us incl. pack ({ff / t / 1e6:.1f} effective TF/s)  max|ctx - ctx_fp32| / max|ctx| = "
              f"{float((c2 - ctx).abs().max() / ctx.abs().max()):.2e}")
