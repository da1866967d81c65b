"""Explicit (autograd-free) CPU formulas for the CAB / FFM hot path.

TEST INFRASTRUCTURE: see ``oracle/__init__.py``.  Every function is dtype
generic (run it in float64 to get a tighter reference than the fp32 result of
either side) and works on NCHW-flattened tensors exactly as the reference lays
them out.

Reference lines restated here
-----------------------------
* attention core        ``/root/reference/src/models/cab.py:149-154``
* BatchNorm2d semantics ``torch.nn.BatchNorm2d`` defaults as used at
                        ``cab.py:33,109,115`` and ``cabinet.py:39``
* FFM                   ``/root/reference/src/models/cabinet.py:142-153``
* PSP                   ``/root/reference/src/models/cab.py:65-76``
* local gate            ``/root/reference/src/models/cab.py:182-184``
"""

from __future__ import annotations

import torch

# ---------------------------------------------------------------------------
# Attention core  (cab.py:149-154)
# ---------------------------------------------------------------------------


def attn_core_fwd(q, k, v, scale):
    """q,k: (B,Kc,n)  v: (B,Vc,n)  ->  ctx (B,Vc,n), lse (B,n).

    S[b,i,j] = scale * sum_c q[b,c,i] k[b,c,j]   (cab.py:149-150)
    P = softmax_j(S)                             (cab.py:151)
    ctx[b,c,i] = sum_j P[b,i,j] v[b,c,j]         (cab.py:153-154)
    """
    s = torch.einsum("bci,bcj->bij", q, k) * scale
    m = s.max(dim=-1, keepdim=True).values
    e = (s - m).exp()
    l = e.sum(dim=-1, keepdim=True)
    p = e / l
    ctx = torch.einsum("bij,bcj->bci", p, v)
    lse = (m + l.log()).squeeze(-1)
    return ctx, lse


def attn_core_bwd(g, q, k, v, ctx, lse, scale):
    """Backward of ``attn_core_fwd`` given g = dL/dctx (B,Vc,n).

    Returns (dq, dk, dv).  P is recomputed from q, k and the saved LSE.
    """
    s = torch.einsum("bci,bcj->bij", q, k) * scale
    p = (s - lse.unsqueeze(-1)).exp()
    delta = (g * ctx).sum(dim=1)  # (B,n)  D_i
    dv = torch.einsum("bij,bci->bcj", p, g)
    dp = torch.einsum("bci,bcj->bij", g, v)
    ds = p * (dp - delta.unsqueeze(-1))
    dq = scale * torch.einsum("bij,bcj->bci", ds, k)
    dk = scale * torch.einsum("bij,bci->bcj", ds, q)
    return dq, dk, dv


# ---------------------------------------------------------------------------
# BatchNorm2d (defaults: eps=1e-5, momentum=0.1, affine, track_running_stats)
# ---------------------------------------------------------------------------


def bn_fwd(z, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """z: (B,C,*).  Returns (y, mean, invstd, new_running_mean, new_running_var).

    Training: biased batch statistics for normalisation, unbiased variance
    folded into the running buffer.  Eval: running statistics.
    """
    dims = [0] + list(range(2, z.dim()))
    shape = [1, -1] + [1] * (z.dim() - 2)
    if training:
        m = z.numel() // z.shape[1]
        mean = z.mean(dim=dims)
        var = ((z - mean.view(shape)) ** 2).mean(dim=dims)
        new_rm = (1 - momentum) * running_mean + momentum * mean
        new_rv = (1 - momentum) * running_var + momentum * var * (m / max(m - 1, 1))
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    invstd = (var + eps).rsqrt()
    y = (z - mean.view(shape)) * (invstd * weight).view(shape) + bias.view(shape)
    return y, mean, invstd, new_rm, new_rv


def bn_bwd(dy, z, weight, mean, invstd, training):
    """Returns (dz, dweight, dbias)."""
    dims = [0] + list(range(2, z.dim()))
    shape = [1, -1] + [1] * (z.dim() - 2)
    xhat = (z - mean.view(shape)) * invstd.view(shape)
    dweight = (dy * xhat).sum(dim=dims)
    dbias = dy.sum(dim=dims)
    if training:
        m = z.numel() // z.shape[1]
        dz = (weight * invstd).view(shape) * (
            dy - (dbias / m).view(shape) - xhat * (dweight / m).view(shape)
        )
    else:
        dz = dy * (weight * invstd).view(shape)
    return dz, dweight, dbias


# ---------------------------------------------------------------------------
# Feature Fusion Module  (cabinet.py:142-153)
# ---------------------------------------------------------------------------


def ffm_fwd(fsp, fcp, w_blk, bn_w, bn_b, run_mean, run_var, w1, w2, training,
            momentum=0.1, eps=1e-5):
    """fsp (B,Cs,h,w), fcp (B,Cc,h,w); w_blk (Co,Cs+Cc); w1 (Co/4,Co); w2 (Co,Co/4).

    Returns dict with out and everything the backward needs.
    """
    b, _, h, w = fsp.shape
    x = torch.cat([fsp, fcp], dim=1).flatten(2)  # (B,Cin,hw)        cabinet.py:143
    z = torch.einsum("oc,bcp->bop", w_blk, x)  # 1x1 conv, no bias  cabinet.py:30-38
    y, mean, invstd, new_rm, new_rv = bn_fwd(z, bn_w, bn_b, run_mean, run_var,
                                             training, momentum, eps)
    feat = y.clamp_min(0)  # ReLU            cabinet.py:44
    pooled = feat.mean(dim=2)  # (B,Co)         cabinet.py:146
    u = pooled @ w1.t()  # (B,Co/4)                         cabinet.py:147
    r = u.clamp_min(0)  # cabinet.py:148
    s = r @ w2.t()  # (B,Co)                                cabinet.py:149
    a = torch.sigmoid(s)  # cabinet.py:150
    out = feat * a.unsqueeze(-1) + feat  # cabinet.py:152-153
    return dict(out=out.view(b, -1, h, w), x=x, z=z, mean=mean, invstd=invstd,
                feat=feat, pooled=pooled, u=u, r=r, a=a,
                new_running_mean=new_rm, new_running_var=new_rv)


def ffm_bwd(g, saved, w_blk, bn_w, w1, w2, training, n_sp_channels):
    """g = dL/dout (B,Co,h,w).  Returns dict of gradients."""
    b, co = g.shape[:2]
    g = g.flatten(2)
    feat, a, r, u, pooled = (saved[k] for k in ("feat", "a", "r", "u", "pooled"))
    hw = feat.shape[2]
    da = (g * feat).sum(dim=2)  # (B,Co)
    ds = da * a * (1 - a)
    dw2 = ds.t() @ r  # (Co,Co/4)
    dr = ds @ w2  # (B,Co/4)
    du = dr * (u > 0).to(dr.dtype)
    dw1 = du.t() @ pooled  # (Co/4,Co)
    dm = du @ w1  # (B,Co)
    dfeat = g * (1 + a).unsqueeze(-1) + (dm / hw).unsqueeze(-1)
    dy = dfeat * (feat > 0).to(dfeat.dtype)
    dz, dbn_w, dbn_b = bn_bwd(dy, saved["z"], bn_w, saved["mean"], saved["invstd"], training)
    dx = torch.einsum("oc,bop->bcp", w_blk, dz)
    dw_blk = torch.einsum("bop,bcp->oc", dz, saved["x"])
    h_w = saved["out"].shape[2:]
    return dict(dfsp=dx[:, :n_sp_channels].reshape(b, n_sp_channels, *h_w),
                dfcp=dx[:, n_sp_channels:].reshape(b, -1, *h_w),
                dw_blk=dw_blk, dbn_w=dbn_w, dbn_b=dbn_b, dw1=dw1, dw2=dw2)


# ---------------------------------------------------------------------------
# PSP operators (cab.py:56, 65-76): adaptive average pooling and bilinear
# resize (align_corners=False) written as explicit 1-D interpolation matrices.
# ---------------------------------------------------------------------------


def adaptive_pool_matrix(n_in, n_out, dtype=torch.float64):
    """(n_out, n_in) matrix A with A @ x == AdaptiveAvgPool1d(n_out)(x).

    Bin r covers floor(r*n_in/n_out) .. ceil((r+1)*n_in/n_out)-1.
    """
    a = torch.zeros(n_out, n_in, dtype=dtype)
    for r in range(n_out):
        lo = (r * n_in) // n_out
        hi = -((-(r + 1) * n_in) // n_out)
        a[r, lo:hi] = 1.0 / (hi - lo)
    return a


def bilinear_matrix(n_in, n_out, dtype=torch.float64):
    """(n_out, n_in) matrix U with U @ x == 1-D linear resize, align_corners=False.

    src = (dst + 0.5) * n_in / n_out - 0.5, clamped below at 0; i1 = min(i0+1, n_in-1).
    """
    u = torch.zeros(n_out, n_in, dtype=dtype)
    scale = n_in / n_out
    for d in range(n_out):
        src = max((d + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(src), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        lam = src - i0
        u[d, i0] += 1.0 - lam
        u[d, i1] += lam
    return u


def psp_fwd(x, w_proj, sizes=(1, 3, 6, 8)):
    """x (B,C,H,W), w_proj (C, C*(len(sizes)+1)) -> (B,C,H,W).   cab.py:65-76"""
    _, _, h, w = x.shape
    priors = [x]
    for s in sizes:
        ah, aw = adaptive_pool_matrix(h, s, x.dtype), adaptive_pool_matrix(w, s, x.dtype)
        uh, uw = bilinear_matrix(s, h, x.dtype), bilinear_matrix(s, w, x.dtype)
        pooled = torch.einsum("rh,bchw,sw->bcrs", ah, x, aw)
        priors.append(torch.einsum("hr,bcrs,ws->bchw", uh, pooled, uw))
    cat = torch.cat(priors, dim=1)
    return torch.einsum("oc,bchw->bohw", w_proj, cat)


# ---------------------------------------------------------------------------
# Local gate (cab.py:182-184):  x + x * sigmoid(refine(x))
# ---------------------------------------------------------------------------


def local_gate_fwd(x, refined):
    return x + x * torch.sigmoid(refined)


def local_gate_bwd(g, x, refined):
    """Returns (dx_direct, drefined) for out = x * (1 + sigmoid(refined))."""
    m = torch.sigmoid(refined)
    return g * (1 + m), g * x * m * (1 - m)
