"""How chaotic the random-init model is: relative change of selected gradient tensors when ONE early layer's output is
perturbed by 1e-6 relative (stock MIOpen path with noise injected after features.1's 1x1 convolution).  Typical output:
1e-3 .. 1e-2 for backbone, attention-branch and even FFM weights -- the reason tests/test_gpu_model.py measures the
sensitivity of each gradient to a one-ulp input perturbation before holding it to 1e-3."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import cabinet_amd.functional as Fh
from cabinet_amd.train import build_model, make_criteria, synthetic_batch
cb = torch.ops.aten.convolution_backward
class Noisy(torch.autograd.Function):
    eps = 0.0
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        y = F.conv2d(x, w)
        if Noisy.eps:
            g = torch.Generator(device="cuda").manual_seed(5)
            y = y * (1 + Noisy.eps * torch.randn(y.shape, device="cuda", generator=g))
        return y
    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        r = cb(g.contiguous(), x, w, None, [1,1],[0,0],[1,1], False,[0,0],1,[True,True,False])
        return r[0], r[1]
Fh.pwconv = lambda x, conv: Noisy.apply(x, conv.weight)
Fh.pwconv_supported = lambda conv, x: conv.kernel_size == (1, 1) and (conv.in_channels, conv.out_channels) == (16, 16)
def grads(mode, batch, size, ncls, dseed, eps):
    Noisy.eps = eps
    n = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False).cuda().train()
    im, lb = synthetic_batch(batch, size, size, ncls, "cuda", seed=dseed)
    crit = make_criteria(batch, size, size, "cuda")
    out, out16 = n(im); (crit[0](out, lb) + crit[1](out16, lb)).backward(); torch.cuda.synchronize()
    return {k: p.grad.double() for k, p in n.named_parameters() if p.grad is not None}
for cfg in [("large", 1, 512, 19, 1), ("large", 2, 512, 19, 1), ("large", 2, 256, 19, 1), ("large", 1, 512, 19, 2), ("large", 1, 512, 19, 3),
            ("large", 2, 384, 19, 1), ("large", 1, 640, 19, 1), ("small", 4, 512, 8, 1)]:
    a = grads(*cfg, 0.0); b = grads(*cfg, 1e-6)
    ks = ["mobile.features.0.0.weight", "ab.conva.0.weight", "ab.a2block.global_attn.to_query.0.weight", "ffm.convblk.conv.weight"]
    print(cfg, ["%.1e" % float((a[k]-b[k]).norm()/a[k].norm()) for k in ks])
