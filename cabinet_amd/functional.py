"""Autograd operators of the hot path, backed by libcabinet_hip.so on HIP tensors.

``cab_attention``   replaces reference src/models/cab.py:149-154          (K1 / K2)
``cab_qkv``         cab.py:137-146 (projections, BN, PSP)                 (K6)     ``conv1x1``  cab.py:155
``cab_local``       cab.py:182-184 and :213-216                           (K5)
``ffm_fused``       src/models/cabinet.py:142-153                         (K3 / K4; ``ffm_fused_upsampled`` + :228-230)
``bn_act``          cabinet.py:42-44 and every BatchNorm2d(+act) pair     (K7)     ``gate_act``  mobilenetv3.py:79-83
``bn_relu_cls``     cabinet.py:90-92, :161-172 (BN -> ReLU -> 1x1 classifier)  (K12)   ``conv3x3``  cabinet.py:59, :88-89, :160 (K11)
``dwconv`` / ``bn_act_dwconv``   mobilenetv3.py:118-126,135-143           (K8)
``stem_conv``       cabinet.py:111                                        (K9)     ``pwconv``   mobilenetv3.py:128-131
OHEM head           src/utils/loss.py:51-80 + cabinet.py:240-245          (``ohem_up_*``, used by cabinet_amd.loss)

ONE dispatch rule for every operator here:
  * device tensor, shape inside the kernel family's coverage (each family exports ``*_supported``): the hand-written
    HIP kernels run; a missing / mismatching library or a failing launch raises RuntimeError -- nothing is caught and
    retried on another path;
  * device tensor, shape outside the coverage (attention channel pairs other than (128,128) (256,128) (64,64); producers
    with channel counts that are not multiples of 16; depthwise kernels other than 3 / 5; the local branch only for planes
    wider than its LDS tile, > ~1800 columns -- every B*H*W is served: channel-resident kernel up to 8192 positions per
    channel, tiled multi-workgroup form above): the module's composite ATen forward, on the device -- the reference API accepts any
    channel count, so must this one;
  * host (CPU) tensors: the composite ATen forward, so modules stay usable for checkpoint surgery, EMA copies and CPU
    unit tests, exactly like any nn.Module.
"""

from __future__ import annotations

import ctypes as _ct

import torch
import torch.nn.functional as F

from . import _lib


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _stream_handle(device):
    return torch.cuda.current_stream(device).cuda_stream


def _workspace(nbytes, device):
    if nbytes == 0:
        return None, 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws, nbytes


def _aligned(t):
    """A dense tensor of any dtype at a 16-byte aligned address (labels, internal slices such as ``loss_px[1]``)."""
    t = t.contiguous()
    return t.clone() if t.data_ptr() & 15 else t


def _f32c(t):
    """Borrowed inputs must be dense fp32 and 16-byte aligned (the rules the C ABI documents): a contiguous VIEW that starts
    4 or 8 bytes into its allocation (``x[1:]``, ``x.flatten()[1:]``) is copied."""
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.data_ptr() & 15:
        t = t.clone()
    return t


# --------------------------------------------------------------------------- attention core


PREC_FP32, PREC_BF16X3, PREC_BF16X6 = 0, 1, 2  # include/cabinet_hip.h: CABINET_PREC_*
_PRECISIONS = {"fp32": PREC_FP32, "bf16x3": PREC_BF16X3, "bf16x6": PREC_BF16X6}


def attn_fwd_hip(q, k, v, scale, precision=PREC_FP32):
    """q,k (B,Kc,n), v (B,Vc,n) fp32 device tensors -> ctx (B,Vc,n), lse (B,n)."""
    lib = _lib.load()
    B, Kc, n = q.shape
    Vc = v.shape[1]
    ctx = torch.empty((B, Vc, n), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, n), dtype=torch.float32, device=q.device)
    ws, nbytes = _workspace(lib.cabinet_cab_attn_fwd_workspace_bytes(B, Kc, Vc, n, int(precision)), q.device)
    with torch.cuda.device(q.device):
        rc = lib.cabinet_cab_attn_fwd(_ptr(q), _ptr(k), _ptr(v), float(scale), B, Kc, Vc, n, int(precision), _ptr(ctx),
                                      _ptr(lse), _ptr(ws), nbytes, _stream_handle(q.device))
    _lib.check(rc, "cabinet_cab_attn_fwd")
    return ctx, lse


def attn_bwd_hip(g, q, k, v, ctx, lse, scale):
    lib = _lib.load()
    B, Kc, n = q.shape
    Vc = v.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws, nbytes = _workspace(lib.cabinet_cab_attn_bwd_workspace_bytes(B, Kc, Vc, n), q.device)
    with torch.cuda.device(q.device):
        rc = lib.cabinet_cab_attn_bwd(_ptr(g), _ptr(q), _ptr(k), _ptr(v), _ptr(ctx), _ptr(lse), float(scale),
                                      B, Kc, Vc, n, _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), nbytes,
                                      _stream_handle(q.device))
    _lib.check(rc, "cabinet_cab_attn_bwd")
    return dq, dk, dv


class _CabAttention(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, q, k, v, scale, precision):
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        out, lse = attn_fwd_hip(q, k, v, scale, precision)
        fn_ctx.save_for_backward(q, k, v, out, lse)
        fn_ctx.scale = scale
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        q, k, v, out, lse = fn_ctx.saved_tensors
        dq, dk, dv = attn_bwd_hip(_f32c(g), q, k, v, out, lse, fn_ctx.scale)
        return dq, dk, dv, None, None


def cab_attention_supported(Kc, Vc):
    """True for the (key, value) channel pairs K1 / K2 are instantiated for."""
    return bool(_lib.load().cabinet_cab_attn_supported(int(Kc), int(Vc)))


# Matrix arithmetic of K1's two contractions (include/cabinet_hip.h, CABINET_PREC_*): "fp32" = exact fp32 MFMA (default),
# "bf16x6" = fp32 operands as three bf16 pieces each, six bf16 MFMA products per fp32 product (fp32-level accuracy, 2.7x the
# matrix rate), "bf16x3" = two pieces, three products (~1e-5 per tensor; a measured variant).  Set per call, or process-wide
# through this module attribute / the environment variable CABINET_ATTN_PRECISION; channel pairs without a split-bf16
# instantiation always run fp32.
import os as _os  # noqa: E402

ATTN_PRECISION = _os.environ.get("CABINET_ATTN_PRECISION", "fp32")
# the same switch for the FFM's big forward product z = W_s . fsp + U(W_c . low) (cabinet_ffm_up_fwd); shapes without a
# split-bf16 instantiation run fp32 whatever is set
FFM_PRECISION = _os.environ.get("CABINET_FFM_PRECISION", "fp32")


def _ffm_precision():
    """``FFM_PRECISION`` -> CABINET_PREC_*; an unknown name raises (a typo such as ``bf16_6`` must not silently measure
    fp32), exactly as ``_resolve_precision`` does for the attention switch."""
    if FFM_PRECISION not in _PRECISIONS:
        raise RuntimeError(f"ffm_fused_upsampled: unknown CABINET_FFM_PRECISION {FFM_PRECISION!r} "
                           f"(one of {sorted(_PRECISIONS)})")
    return _PRECISIONS[FFM_PRECISION]


def _resolve_precision(precision, Kc, Vc):
    name = ATTN_PRECISION if precision is None else precision
    if name not in _PRECISIONS:
        raise RuntimeError(f"cab_attention: unknown precision {name!r} (one of {sorted(_PRECISIONS)})")
    code = _PRECISIONS[name]
    if code != PREC_FP32 and not _lib.load().cabinet_cab_attn_precision_supported(int(Kc), int(Vc), code):
        if precision is not None:
            raise RuntimeError(f"cab_attention: precision {name!r} is not built for (Kc={Kc}, Vc={Vc})")
        code = PREC_FP32
    return code


def cab_attention(q, k, v, scale, precision=None):
    """ctx[b,c,i] = sum_j softmax_j(scale * <q[b,:,i], k[b,:,j]>) v[b,c,j].

    q,k: (B,Kc,n)  v: (B,Vc,n)  ->  (B,Vc,n).   Reference: cab.py:149-154.
    ``precision``: None (module default ``ATTN_PRECISION``) | "fp32" | "bf16x6" | "bf16x3" -- forward contractions only; the
    backward (K2) is fp32 MFMA in every case.
    """
    if q.dim() != 3 or k.shape != q.shape or v.dim() != 3 or v.shape[0] != q.shape[0] or v.shape[2] != q.shape[2]:
        raise RuntimeError(f"cab_attention: bad shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if q.is_cuda and cab_attention_supported(q.shape[1], v.shape[1]):
        return _CabAttention.apply(q, k, v, float(scale), _resolve_precision(precision, q.shape[1], v.shape[1]))
    # host tensors, and channel pairs without a gfx950 instantiation: composite ATen ops (module docstring: one rule)
    attn = torch.bmm(q.transpose(1, 2), k) * scale
    attn = F.softmax(attn, dim=-1)
    return torch.bmm(v, attn.transpose(1, 2))


def attn_proj_fwd_hip(q, k, v, w2, scale, need_ctx=True):
    """q,k (B,Kc,n), v (B,Vc,n), w2 (Co,Vc) fp32 device tensors -> glob (B,Co,n), ctx (B,Vc,n) or None, lse (B,n): K1 with the output
    projection in its epilogue (cabinet_cab_attn_proj_fwd)."""
    lib = _lib.load()
    B, Kc, n = q.shape
    Vc, Co = v.shape[1], w2.shape[0]
    ctx = torch.empty((B, Vc, n), dtype=torch.float32, device=q.device) if need_ctx else None
    glob = torch.empty((B, Co, n), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, n), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        rc = lib.cabinet_cab_attn_proj_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(w2), float(scale), B, Kc, Vc, Co, n, _ptr(ctx),
                                           _ptr(glob), _ptr(lse), _stream_handle(q.device))
    _lib.check(rc, "cabinet_cab_attn_proj_fwd")
    return glob, ctx, lse


class _CabAttentionProj(torch.autograd.Function):
    """K1 with the CAB's output projection in its epilogue: ``project_out(attention(q, k, v))`` in ONE launch
    (cabinet_cab_attn_proj_fwd).  Backward = the projection's backward (cabinet_conv1x1_bwd: dctx, dw) + K2."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, q, k, v, w_out, scale):
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        w2 = _f32c(w_out.detach()).reshape(w_out.shape[0], -1)
        need_ctx = any(fn_ctx.needs_input_grad[:4])
        glob, ctx, lse = attn_proj_fwd_hip(q, k, v, w2, scale, need_ctx)
        if need_ctx:
            fn_ctx.save_for_backward(q, k, v, ctx, lse, w2)
        fn_ctx.scale, fn_ctx.w_shape = scale, w_out.shape
        return glob

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        q, k, v, ctx, lse, w2 = fn_ctx.saved_tensors
        g = _f32c(g)
        B, Vc, n = ctx.shape
        Co = w2.shape[0]
        need_attn, need_w = any(fn_ctx.needs_input_grad[:3]), fn_ctx.needs_input_grad[3]
        dctx = torch.empty_like(ctx) if need_attn else None
        dw = torch.empty_like(w2) if need_w else None
        ws, nbytes = _workspace(lib.cabinet_conv1x1_bwd_workspace_bytes(B, Vc, Co, n), g.device)
        with torch.cuda.device(g.device):
            rc = lib.cabinet_conv1x1_bwd(_ptr(g), _ptr(ctx), _ptr(w2), B, Vc, Co, n, _ptr(dctx), _ptr(dw), _ptr(ws), nbytes,
                                         _stream_handle(g.device))
        _lib.check(rc, "cabinet_conv1x1_bwd")
        dq = dk = dv = None
        if need_attn:
            dq, dk, dv = attn_bwd_hip(dctx, q, k, v, ctx, lse, fn_ctx.scale)
        return dq, dk, dv, (dw.reshape(fn_ctx.w_shape) if need_w else None), None


def cab_attention_proj_supported(q, v, w_out):
    """True when ``project_out(cab_attention(q, k, v))`` runs as ONE kernel (K1 with the projection in its epilogue): device
    tensors, the 8-wave kernel's channel pairs, n % 4 == 0, Co % 32 == 0, a batch the forward runs without a key split, fp32 MFMA
    (the split-bf16 precisions keep the two-launch form)."""
    if not (q.is_cuda and q.dim() == 3 and v.dim() == 3 and w_out.dim() in (2, 4)) or ATTN_PRECISION != "fp32" or not PROJ_FUSED:
        return False
    if w_out.dim() == 4 and tuple(w_out.shape[2:]) != (1, 1):
        return False
    B, Kc, n = q.shape
    return bool(_lib.load().cabinet_cab_attn_proj_supported(int(B), int(Kc), int(v.shape[1]), int(w_out.shape[0]), int(n)))


# CABINET_ATTN_PROJ_FUSED=0 keeps the round-4 pair of launches (K1, then the projection as a small GEMM): A/B timing
PROJ_FUSED = _os.environ.get("CABINET_ATTN_PROJ_FUSED", "1") != "0"


def cab_attention_proj(q, k, v, w_out, scale):
    """``conv1x1(cab_attention(q, k, v, scale), w_out)`` -> (B, Co, n).  Reference: cab.py:149-155.  ONE launch where
    :func:`cab_attention_proj_supported`, the two operators otherwise."""
    if cab_attention_proj_supported(q, v, w_out) and k.shape == q.shape and v.shape[0] == q.shape[0] and v.shape[2] == q.shape[2] \
            and w_out.shape[1] == v.shape[1]:
        return _CabAttentionProj.apply(q, k, v, w_out, float(scale))
    ctx = cab_attention(q, k, v, scale)
    B, Vc, n = ctx.shape
    return conv1x1(ctx.reshape(B, Vc, n, 1), w_out).reshape(B, -1, n)


# --------------------------------------------------------------------------- FFM


def _ffm_dims(fsp, fcp, w_blk, w1):
    B, Cs, H, W = fsp.shape
    Cc = fcp.shape[1]
    Co = w_blk.shape[0]
    Cm = w1.shape[0]
    return B, Cs, Cc, Co, Cm, H, W


def ffm_fwd_hip(fsp, fcp, w_blk, bn_w, bn_b, run_mean, run_var, w1, w2, training, momentum, eps):
    lib = _lib.load()
    dims = _ffm_dims(fsp, fcp, w_blk, w1)
    B, Cs, Cc, Co, Cm, H, W = dims
    dev = fsp.device
    out = torch.empty((B, Co, H, W), dtype=torch.float32, device=dev)
    z = torch.empty((B, Co, H, W), dtype=torch.float32, device=dev)
    save_mean = torch.empty(Co, dtype=torch.float32, device=dev)
    save_invstd = torch.empty(Co, dtype=torch.float32, device=dev)
    pooled = torch.empty((B, Co), dtype=torch.float32, device=dev)
    gate = torch.empty((B, Co), dtype=torch.float32, device=dev)
    ws, nbytes = _workspace(lib.cabinet_ffm_fwd_workspace_bytes(*dims), dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ffm_fwd(_ptr(fsp), _ptr(fcp), _ptr(w_blk), _ptr(bn_w), _ptr(bn_b), _ptr(run_mean),
                                 _ptr(run_var), _ptr(w1), _ptr(w2), *dims, int(training), float(momentum),
                                 float(eps), _ptr(out), _ptr(z), _ptr(save_mean), _ptr(save_invstd),
                                 _ptr(pooled), _ptr(gate), _ptr(ws), nbytes, _stream_handle(dev))
    _lib.check(rc, "cabinet_ffm_fwd")
    return out, z, save_mean, save_invstd, pooled, gate


def ffm_bwd_hip(g, fsp, fcp, w_blk, bn_w, bn_b, w1, w2, z, save_mean, save_invstd, pooled, gate, training):
    lib = _lib.load()
    dims = _ffm_dims(fsp, fcp, w_blk, w1)
    dev = fsp.device
    dfsp, dfcp = torch.empty_like(fsp), torch.empty_like(fcp)
    dw_blk = torch.empty_like(w_blk)
    dbn_w, dbn_b = torch.empty_like(bn_w), torch.empty_like(bn_b)
    dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
    ws, nbytes = _workspace(lib.cabinet_ffm_bwd_workspace_bytes(*dims), dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ffm_bwd(_ptr(g), _ptr(fsp), _ptr(fcp), _ptr(w_blk), _ptr(bn_w), _ptr(bn_b), _ptr(w1),
                                 _ptr(w2), _ptr(z), _ptr(save_mean), _ptr(save_invstd), _ptr(pooled),
                                 _ptr(gate), *dims, int(training), _ptr(dfsp), _ptr(dfcp), _ptr(dw_blk),
                                 _ptr(dbn_w), _ptr(dbn_b), _ptr(dw1), _ptr(dw2), _ptr(ws), nbytes,
                                 _stream_handle(dev))
    _lib.check(rc, "cabinet_ffm_bwd")
    return dfsp, dfcp, dw_blk, dbn_w, dbn_b, dw1, dw2


class _FfmFused(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, fsp, fcp, w_blk, bn_w, bn_b, w1, w2, run_mean, run_var, training, momentum, eps):
        fsp, fcp = _f32c(fsp), _f32c(fcp)
        w_blk2 = _f32c(w_blk).view(w_blk.shape[0], -1)
        w1_2 = _f32c(w1).view(w1.shape[0], -1)
        w2_2 = _f32c(w2).view(w2.shape[0], -1)
        bn_w, bn_b = _f32c(bn_w), _f32c(bn_b)
        out, z, mean, invstd, pooled, gate = ffm_fwd_hip(fsp, fcp, w_blk2, bn_w, bn_b, run_mean, run_var,
                                                         w1_2, w2_2, training, momentum, eps)
        fn_ctx.save_for_backward(fsp, fcp, w_blk2, bn_w, bn_b, w1_2, w2_2, z, mean, invstd, pooled, gate)
        fn_ctx.training = training
        fn_ctx.w_shapes = (w_blk.shape, w1.shape, w2.shape)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        fsp, fcp, w_blk, bn_w, bn_b, w1, w2, z, mean, invstd, pooled, gate = fn_ctx.saved_tensors
        dfsp, dfcp, dw_blk, dbn_w, dbn_b, dw1, dw2 = ffm_bwd_hip(
            _f32c(g), fsp, fcp, w_blk, bn_w, bn_b, w1, w2, z, mean, invstd, pooled, gate, fn_ctx.training)
        s_blk, s1, s2 = fn_ctx.w_shapes
        return (dfsp, dfcp, dw_blk.view(s_blk), dbn_w, dbn_b, dw1.view(s1), dw2.view(s2),
                None, None, None, None, None)


def ffm_up_fwd_hip(fsp, low, w_blk, bn_w, bn_b, run_mean, run_var, w1, w2, training, momentum, eps, precision=0):
    lib = _lib.load()
    B, Cs, H, W = fsp.shape
    Cc, Hl, Wl = low.shape[1:]
    Co, Cm = w_blk.shape[0], w1.shape[0]
    dims = (B, Cs, Cc, Co, Cm, H, W, Hl, Wl)
    dev = fsp.device
    out = torch.empty((B, Co, H, W), dtype=torch.float32, device=dev)
    z = torch.empty((B, Co, H, W), dtype=torch.float32, device=dev)
    save_mean = torch.empty(Co, dtype=torch.float32, device=dev)
    save_invstd = torch.empty(Co, dtype=torch.float32, device=dev)
    pooled = torch.empty((B, Co), dtype=torch.float32, device=dev)
    gate = torch.empty((B, Co), dtype=torch.float32, device=dev)
    ws, nbytes = _workspace(lib.cabinet_ffm_up_fwd_workspace_bytes(*dims), dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ffm_up_fwd(_ptr(fsp), _ptr(low), _ptr(w_blk), _ptr(bn_w), _ptr(bn_b), _ptr(run_mean),
                                    _ptr(run_var), _ptr(w1), _ptr(w2), *dims, int(training), float(momentum),
                                    float(eps), int(precision), _ptr(out), _ptr(z), _ptr(save_mean), _ptr(save_invstd),
                                    _ptr(pooled), _ptr(gate), _ptr(ws), nbytes, _stream_handle(dev))
    _lib.check(rc, "cabinet_ffm_up_fwd")
    return out, z, save_mean, save_invstd, pooled, gate


def ffm_up_bwd_hip(g, fsp, low, w_blk, bn_w, bn_b, w1, w2, z, save_mean, save_invstd, pooled, gate, training):
    lib = _lib.load()
    B, Cs, H, W = fsp.shape
    Cc, Hl, Wl = low.shape[1:]
    dims = (B, Cs, Cc, w_blk.shape[0], w1.shape[0], H, W, Hl, Wl)
    dev = fsp.device
    dfsp, dlow = torch.empty_like(fsp), torch.empty_like(low)
    dw_blk = torch.empty_like(w_blk)
    dbn_w, dbn_b = torch.empty_like(bn_w), torch.empty_like(bn_b)
    dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
    ws, nbytes = _workspace(lib.cabinet_ffm_up_bwd_workspace_bytes(*dims), dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ffm_up_bwd(_ptr(g), _ptr(fsp), _ptr(low), _ptr(w_blk), _ptr(bn_w), _ptr(bn_b), _ptr(w1),
                                    _ptr(w2), _ptr(z), _ptr(save_mean), _ptr(save_invstd), _ptr(pooled),
                                    _ptr(gate), *dims, int(training), _ptr(dfsp), _ptr(dlow), _ptr(dw_blk),
                                    _ptr(dbn_w), _ptr(dbn_b), _ptr(dw1), _ptr(dw2), _ptr(ws), nbytes,
                                    _stream_handle(dev))
    _lib.check(rc, "cabinet_ffm_up_bwd")
    return dfsp, dlow, dw_blk, dbn_w, dbn_b, dw1, dw2


class _FfmUpFused(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, fsp, low, w_blk, bn_w, bn_b, w1, w2, run_mean, run_var, training, momentum, eps):
        fsp, low = _f32c(fsp), _f32c(low)
        w_blk2 = _f32c(w_blk).view(w_blk.shape[0], -1)
        w1_2 = _f32c(w1).view(w1.shape[0], -1)
        w2_2 = _f32c(w2).view(w2.shape[0], -1)
        bn_w, bn_b = _f32c(bn_w), _f32c(bn_b)
        out, z, mean, invstd, pooled, gate = ffm_up_fwd_hip(fsp, low, w_blk2, bn_w, bn_b, run_mean, run_var,
                                                            w1_2, w2_2, training, momentum, eps,
                                                            _ffm_precision())
        fn_ctx.save_for_backward(fsp, low, w_blk2, bn_w, bn_b, w1_2, w2_2, z, mean, invstd, pooled, gate)
        fn_ctx.training = training
        fn_ctx.w_shapes = (w_blk.shape, w1.shape, w2.shape)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        fsp, low, w_blk, bn_w, bn_b, w1, w2, z, mean, invstd, pooled, gate = fn_ctx.saved_tensors
        dfsp, dlow, dw_blk, dbn_w, dbn_b, dw1, dw2 = ffm_up_bwd_hip(
            _f32c(g), fsp, low, w_blk, bn_w, bn_b, w1, w2, z, mean, invstd, pooled, gate, fn_ctx.training)
        s_blk, s1, s2 = fn_ctx.w_shapes
        return (dfsp, dlow, dw_blk.view(s_blk), dbn_w, dbn_b, dw1.view(s1), dw2.view(s2),
                None, None, None, None, None)


_NBT_PENDING = None  # list of num_batches_tracked buffers while batched_bn_counters() is open


class batched_bn_counters:
    """Within this context the ``num_batches_tracked += 1`` of every BatchNorm2d served by the HIP ops is deferred
    and applied as ONE multi-tensor add on exit (the model has 59 BatchNorms: 59 one-element kernels per step
    otherwise).  Values after the context are identical; nesting is a no-op."""

    def __enter__(self):
        global _NBT_PENDING
        self.owner = _NBT_PENDING is None
        if self.owner:
            _NBT_PENDING = []
        return self

    def __exit__(self, *exc):
        global _NBT_PENDING
        if self.owner:
            pending, _NBT_PENDING = _NBT_PENDING, None
            if pending:
                # a module applied twice inside the context appears twice: a multi-tensor add with a repeated tensor is a
                # read-modify-write race, so each distinct counter is added its own count
                counts = {}
                for t in pending:
                    counts.setdefault(id(t), [t, 0])[1] += 1
                by_count = {}
                for t, c in counts.values():
                    by_count.setdefault(c, []).append(t)
                for c, ts in by_count.items():
                    torch._foreach_add_(ts, c)
        return False


def _bn_step(bn):
    if not (bn.affine and bn.track_running_stats):
        raise RuntimeError("ffm_fused: BatchNorm2d must be affine with running statistics")
    if bn.training:
        if _NBT_PENDING is not None and bn.momentum is not None:
            _NBT_PENDING.append(bn.num_batches_tracked)
            return True, bn.momentum
        bn.num_batches_tracked.add_(1)
        return True, (bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked))
    return False, 0.0


def ffm_fused_upsampled(fsp, low, conv_w, bn, w1, w2):
    """FFM(fsp, bilinear_upsample(low -> fsp's H x W)) with the upsample fused into the kernels
    (reference cabinet.py:228-230 + :236).  ``low`` is the attention branch's (B,Cc,H/32,W/32) output."""
    if fsp.dim() != 4 or low.dim() != 4 or fsp.shape[0] != low.shape[0]:
        raise RuntimeError(f"ffm_fused_upsampled: bad shapes fsp{tuple(fsp.shape)} low{tuple(low.shape)}")
    training, momentum = _bn_step(bn)
    return _FfmUpFused.apply(fsp, low, conv_w, bn.weight, bn.bias, w1, w2, bn.running_mean, bn.running_var,
                             training, momentum, bn.eps)


def ffm_fused(fsp, fcp, conv_w, bn, w1, w2):
    """FeatureFusionModule.forward on HIP tensors (reference cabinet.py:142-153).

    ``bn`` is the nn.BatchNorm2d that owns the affine parameters and running buffers;
    its buffers are updated in place in training mode like nn.BatchNorm2d would.
    """
    training, momentum = _bn_step(bn)
    return _FfmFused.apply(fsp, fcp, conv_w, bn.weight, bn.bias, w1, w2, bn.running_mean, bn.running_var,
                           training, momentum, bn.eps)


# --------------------------------------------------------------------------- OHEM-CE fused with the final upsample


def ohem_up_fwd_hip(logits_low, labels, size, thresh, ignore_lb):
    """Per-pixel CE of bilinear_upsample(logits_low -> size) vs labels, never materialising the upsample.
    Returns loss_px (B,H,W) and the three reduced statistics as ONE device tensor [n_valid, n_above, sum_above]."""
    lib = _lib.load()
    logits_low, labels = _f32c(logits_low), _aligned(labels)
    B, C, Hl, Wl = logits_low.shape
    H, W = size
    dev = logits_low.device
    nblk = lib.cabinet_ohem_up_blocks(B, H, W)
    loss_px = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    blk_sum = torch.empty(nblk, dtype=torch.float32, device=dev)
    blk_cnt = torch.empty((nblk, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ohem_up_fwd(_ptr(logits_low), _ptr(labels), B, C, Hl, Wl, H, W, float(thresh), int(ignore_lb),
                                     _ptr(loss_px), _ptr(blk_sum), _ptr(blk_cnt), _stream_handle(dev))
        _lib.check(rc, "cabinet_ohem_up_fwd")
        stats = torch.empty(3, dtype=torch.float64, device=dev)
        rc = lib.cabinet_ohem_stats(_ptr(blk_sum), _ptr(blk_cnt), 1, nblk, _ptr(stats), _stream_handle(dev))
    _lib.check(rc, "cabinet_ohem_stats")
    return loss_px, stats


def ohem_up_bwd_hip(logits_low, labels, loss_px, size, thresh, ignore_lb, coef):
    lib = _lib.load()
    logits_low, labels, loss_px = _f32c(logits_low), _aligned(labels), _f32c(loss_px)
    B, C, Hl, Wl = logits_low.shape
    H, W = size
    dev = logits_low.device
    dlow = torch.empty_like(logits_low)
    ws, nbytes = _workspace(lib.cabinet_ohem_up_bwd_workspace_bytes(B, C, Hl, Wl, H, W), dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ohem_up_bwd(_ptr(logits_low), _ptr(labels), _ptr(loss_px), B, C, Hl, Wl, H, W, float(thresh),
                                     int(ignore_lb), float(coef), _ptr(dlow), _ptr(ws), nbytes, _stream_handle(dev))
    _lib.check(rc, "cabinet_ohem_up_bwd")
    return dlow


class _OhemUpSelected(torch.autograd.Function):
    """loss = sum_above / n_above for the 'at least n_min pixels above thresh' branch (reference loss.py:74-75);
    forward statistics were produced by ohem_up_fwd_hip, backward runs the two adjoint kernels.  ``n_above`` is a DEVICE
    scalar: nothing of the step's data-dependent state is baked into a kernel argument, so the op can be replayed from a
    captured hipGraph (cabinet_amd.train.GraphedTrainStep); the host only reads it to pick the branch."""

    @staticmethod
    def forward(fn_ctx, logits_low, labels, loss_px, sum_above, n_above, size, thresh, ignore_lb):
        fn_ctx.save_for_backward(logits_low, labels, loss_px, n_above)
        fn_ctx.meta = (size, thresh, ignore_lb)
        return (sum_above / n_above).to(torch.float32)

    @staticmethod
    def backward(fn_ctx, g):
        logits_low, labels, loss_px, n_above = fn_ctx.saved_tensors
        size, thresh, ignore_lb = fn_ctx.meta
        # the kernels produce U^T[sel * (softmax - onehot)]; upstream gradient and 1/n_above are device scalars folded in after
        dlow = ohem_up_bwd_hip(logits_low, labels, loss_px, size, thresh, ignore_lb, 1.0)
        return dlow * (g / n_above).to(torch.float32), None, None, None, None, None, None, None


def ohem_up_pair_fwd_hip(low_a, low_b, labels, size, thresh, ignore_lb):
    """Both loss heads over the same labels in ONE launch (reference train.py:435 on the outputs of cabinet.py:240-245).
    Returns loss_px (2,B,H,W) and stats (2,3) = per head [n_valid, n_above, sum_above] as one device tensor."""
    lib = _lib.load()
    low_a, low_b, labels = _f32c(low_a), _f32c(low_b), _aligned(labels)
    B, C, Hl, Wl = low_a.shape
    H, W = size
    dev = low_a.device
    nblk = lib.cabinet_ohem_up_blocks(B, H, W)
    loss_px = torch.empty((2, B, H, W), dtype=torch.float32, device=dev)
    blk_sum = torch.empty((2, nblk), dtype=torch.float32, device=dev)
    blk_cnt = torch.empty((2, nblk, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ohem_up_pair_fwd(_ptr(low_a), _ptr(low_b), _ptr(labels), B, C, Hl, Wl, H, W, float(thresh),
                                          int(ignore_lb), _ptr(loss_px), _ptr(blk_sum), _ptr(blk_cnt), _stream_handle(dev))
        _lib.check(rc, "cabinet_ohem_up_pair_fwd")
        stats = torch.empty((2, 3), dtype=torch.float64, device=dev)
        rc = lib.cabinet_ohem_stats(_ptr(blk_sum), _ptr(blk_cnt), 2, nblk, _ptr(stats), _stream_handle(dev))
    _lib.check(rc, "cabinet_ohem_stats")
    return loss_px, stats


def ohem_up_pair_bwd_hip(low_a, low_b, labels, loss_px, size, thresh, ignore_lb, coef):
    lib = _lib.load()
    low_a, low_b, labels, loss_px = _f32c(low_a), _f32c(low_b), _aligned(labels), _f32c(loss_px)
    B, C, Hl, Wl = low_a.shape
    H, W = size
    dev = low_a.device
    dlow = torch.empty((2, B, C, Hl, Wl), dtype=torch.float32, device=dev)
    ws, nbytes = _workspace(lib.cabinet_ohem_up_pair_bwd_workspace_bytes(B, C, Hl, Wl, H, W), dev)
    with torch.cuda.device(dev):
        rc = lib.cabinet_ohem_up_pair_bwd(_ptr(low_a), _ptr(low_b), _ptr(labels), _ptr(loss_px), B, C, Hl, Wl, H, W,
                                          float(thresh), int(ignore_lb), float(coef), _ptr(dlow), _ptr(ws), nbytes,
                                          _stream_handle(dev))
    _lib.check(rc, "cabinet_ohem_up_pair_bwd")
    return dlow


class _OhemUpSelectedPair(torch.autograd.Function):
    """loss = sum_above_a / n_above_a + sum_above_b / n_above_b: the two heads' 'at least n_min pixels above thresh' branches
    (reference loss.py:74-75) with ONE backward launch pair for both (see _OhemUpSelected for the single head)."""

    @staticmethod
    def forward(fn_ctx, low_a, low_b, labels, loss_px, stats, size, thresh, ignore_lb):
        fn_ctx.save_for_backward(low_a, low_b, labels, loss_px, stats)
        fn_ctx.meta = (size, thresh, ignore_lb)
        return (stats[0, 2] / stats[0, 1] + stats[1, 2] / stats[1, 1]).to(torch.float32)

    @staticmethod
    def backward(fn_ctx, g):
        low_a, low_b, labels, loss_px, stats = fn_ctx.saved_tensors
        size, thresh, ignore_lb = fn_ctx.meta
        dlow = ohem_up_pair_bwd_hip(low_a, low_b, labels, loss_px, size, thresh, ignore_lb, 1.0)
        scale = (g / stats[:, 1]).to(torch.float32)  # upstream gradient and 1 / n_above per head: device scalars
        dlow = dlow * scale.view(2, 1, 1, 1, 1)
        return dlow[0], dlow[1], None, None, None, None, None, None


# --------------------------------------------------------------------------- CAB local branch + block output (K5)


def _ptr3(ts):
    return (_ct.c_void_p * 3)(*[t.data_ptr() for t in ts])


def cab_local_supported(x):
    return x.dim() == 4 and bool(_lib.load().cabinet_cab_local_supported(*x.shape))


def cab_local_fwd_hip(x, glob, gamma, dw_w, bn_w, bn_b, run_mean, run_var, training, momentum, eps):
    """out = [gamma*glob +] x*(1 + sigmoid(refine(x))); returns out and the (3,C) saved mean / invstd."""
    lib = _lib.load()
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    mean = torch.empty((3, C), dtype=torch.float32, device=x.device)
    invstd = torch.empty((3, C), dtype=torch.float32, device=x.device)
    ws, nbytes = _workspace(lib.cabinet_cab_local_fwd_workspace_bytes(B, C, H, W), x.device)  # 0 for the resident form
    with torch.cuda.device(x.device):
        rc = lib.cabinet_cab_local_fwd(_ptr(x), _ptr(glob), _ptr(gamma), _ptr3(dw_w), _ptr3(bn_w), _ptr3(bn_b),
                                       _ptr3(run_mean), _ptr3(run_var), B, C, H, W, int(training), float(momentum),
                                       float(eps), _ptr(out), _ptr(mean), _ptr(invstd), _ptr(ws), nbytes,
                                       _stream_handle(x.device))
    _lib.check(rc, "cabinet_cab_local_fwd")
    return out, mean, invstd


def cab_local_bwd_hip(g, x, glob, gamma, dw_w, bn_w, bn_b, mean, invstd, training):
    lib = _lib.load()
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    dglob = torch.empty_like(x) if glob is not None else None
    dgamma_part = torch.empty(C, dtype=torch.float32, device=x.device) if glob is not None else None
    ddw = [torch.empty_like(w) for w in dw_w]
    dbw = [torch.empty_like(w) for w in bn_w]
    dbb = [torch.empty_like(w) for w in bn_b]
    ws, nbytes = _workspace(lib.cabinet_cab_local_bwd_workspace_bytes(B, C, H, W), x.device)
    with torch.cuda.device(x.device):
        rc = lib.cabinet_cab_local_bwd(_ptr(g), _ptr(x), _ptr(glob), _ptr(gamma), _ptr3(dw_w), _ptr3(bn_w),
                                       _ptr3(bn_b), _ptr(mean), _ptr(invstd), B, C, H, W, int(training), _ptr(dx),
                                       _ptr(dglob), _ptr(dgamma_part), _ptr3(ddw), _ptr3(dbw), _ptr3(dbb),
                                       _ptr(ws), nbytes, _stream_handle(x.device))
    _lib.check(rc, "cabinet_cab_local_bwd")
    return dx, dglob, dgamma_part, ddw, dbw, dbb


class _CabLocal(torch.autograd.Function):
    """args: x, glob|None, gamma|None, 3 dw weights, 3 bn weights, 3 bn biases, 3 running means, 3 running vars,
    training, momentum, eps."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, glob, gamma, *rest):
        dw_w, bn_w, bn_b = [_f32c(t) for t in rest[0:3]], [_f32c(t) for t in rest[3:6]], [_f32c(t) for t in rest[6:9]]
        run_mean, run_var = list(rest[9:12]), list(rest[12:15])
        training, momentum, eps = rest[15:18]
        x = _f32c(x)
        glob = _f32c(glob) if glob is not None else None
        gamma = _f32c(gamma) if glob is not None else None
        out, mean, invstd = cab_local_fwd_hip(x, glob, gamma, dw_w, bn_w, bn_b, run_mean, run_var, training,
                                              momentum, eps)
        fn_ctx.save_for_backward(x, glob, gamma, mean, invstd, *dw_w, *bn_w, *bn_b)
        fn_ctx.training = bool(training)
        fn_ctx.w_shapes = [t.shape for t in rest[0:3]]
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        x, glob, gamma, mean, invstd, *params = fn_ctx.saved_tensors
        dw_w, bn_w, bn_b = params[0:3], params[3:6], params[6:9]
        dx, dglob, dgamma_part, ddw, dbw, dbb = cab_local_bwd_hip(_f32c(g), x, glob, gamma, dw_w, bn_w, bn_b, mean,
                                                                  invstd, fn_ctx.training)
        dgamma = dgamma_part.sum().reshape(gamma.shape) if glob is not None else None
        ddw = [d.view(s) for d, s in zip(ddw, fn_ctx.w_shapes)]
        return (dx, dglob, dgamma, *ddw, *dbw, *dbb) + (None,) * 9


def cab_local(x, refine, glob=None, gamma=None):
    """LocalAttention.forward (reference cab.py:182-184), plus ``gamma * glob +`` (cab.py:213-216) when given.

    ``refine`` is the nn.Sequential of three DWConv blocks (conv, BatchNorm2d, ReLU) that owns the parameters
    and running buffers; the buffers are updated in place in training mode like nn.BatchNorm2d would.
    """
    if not x.is_cuda:
        raise RuntimeError("cab_local: device tensors only (host tensors take the composite ATen path)")
    convs = [blk.block[0] for blk in refine]
    bns = [blk.block[1] for blk in refine]
    if len(convs) != 3 or any(c.kernel_size != (3, 3) or c.stride != (1, 1) or c.padding != (1, 1)
                              or c.groups != x.shape[1] or c.bias is not None for c in convs):
        raise RuntimeError("cab_local: expects three depthwise 3x3, stride 1, pad 1, bias-free convolutions")
    if len({(bn.eps, bn.momentum) for bn in bns}) != 1:
        raise RuntimeError("cab_local: the three BatchNorm2d must share eps and momentum")
    steps = [_bn_step(bn) for bn in bns]
    training, momentum = steps[0]
    if any(s[0] != training for s in steps):
        raise RuntimeError("cab_local: BatchNorm2d layers disagree on training mode")
    return _CabLocal.apply(x, glob, gamma, *[c.weight for c in convs], *[bn.weight for bn in bns],
                           *[bn.bias for bn in bns], *[bn.running_mean for bn in bns],
                           *[bn.running_var for bn in bns], training, momentum, bns[0].eps)


# --------------------------------------------------------------------------- q/k/v producers (K6) and 1x1 convolution


def _sizes_arg(sizes):
    return (_ct.c_int * len(sizes))(*[int(s) for s in sizes])


def cab_qkv_supported(x, Kc, Vc, sizes):
    if x.dim() != 4 or not 1 <= len(sizes) <= 4:
        return False
    B, C, H, W = x.shape
    return bool(_lib.load().cabinet_cab_qkv_supported(B, C, Kc, Vc, H, W, len(sizes), _sizes_arg(sizes)))


class _CabQkv(torch.autograd.Function):
    """args: x, wq, wk, wv, bnq_w, bnq_b, bnk_w, bnk_b, wpk, wpv, bnq_rm, bnq_rv, bnk_rm, bnk_rv, sizes, training,
    momentum, eps  ->  q (B,Kc,n), k (B,Kc,n), v (B,Vc,n)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, wq, wk, wv, bnq_w, bnq_b, bnk_w, bnk_b, wpk, wpv, bnq_rm, bnq_rv, bnk_rm, bnk_rv, sizes,
                training, momentum, eps):
        lib = _lib.load()
        x = _f32c(x)
        B, C, H, W = x.shape
        Kc, Vc, n, ns = wq.shape[0], wv.shape[0], H * W, len(sizes)
        shapes = (wq.shape, wk.shape, wv.shape, wpk.shape, wpv.shape)
        wq2, wk2, wv2 = (_f32c(t).view(t.shape[0], C) for t in (wq, wk, wv))
        wpk2, wpv2 = _f32c(wpk).view(Kc, (ns + 1) * Kc), _f32c(wpv).view(Vc, (ns + 1) * Vc)
        bnq_w, bnq_b, bnk_w, bnk_b = _f32c(bnq_w), _f32c(bnq_b), _f32c(bnk_w), _f32c(bnk_b)
        sz = _sizes_arg(sizes)
        dims = (B, C, Kc, Vc, H, W, ns, sz)
        nbp = lib.cabinet_cab_qkv_padded_bins(ns, sz)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)  # noqa: E731
        q, k, v = new(B, Kc, n), new(B, Kc, n), new(B, Vc, n)
        zqk, vv, kk = new(B, 2 * Kc, n), new(B, Vc, n), new(B, Kc, n)
        pooled_k, pooled_v = new(B, ns * Kc, nbp), new(B, ns * Vc, nbp)
        mean, invstd = new(2 * Kc), new(2 * Kc)
        ws, nbytes = _workspace(lib.cabinet_cab_qkv_fwd_workspace_bytes(*dims), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_cab_qkv_fwd(_ptr(x), _ptr(wq2), _ptr(wk2), _ptr(wv2), _ptr(bnq_w), _ptr(bnq_b), _ptr(bnq_rm),
                                         _ptr(bnq_rv), _ptr(bnk_w), _ptr(bnk_b), _ptr(bnk_rm), _ptr(bnk_rv), _ptr(wpk2),
                                         _ptr(wpv2), *dims, int(training), float(momentum), float(eps), _ptr(q),
                                         _ptr(k), _ptr(v), _ptr(zqk), _ptr(vv), _ptr(kk), _ptr(pooled_k),
                                         _ptr(pooled_v), _ptr(mean), _ptr(invstd), _ptr(ws), nbytes,
                                         _stream_handle(x.device))
        _lib.check(rc, "cabinet_cab_qkv_fwd")
        fn_ctx.save_for_backward(x, wq2, wk2, wv2, bnq_w, bnq_b, bnk_w, bnk_b, wpk2, wpv2, zqk, vv, kk, pooled_k,
                                 pooled_v, mean, invstd)
        fn_ctx.sizes, fn_ctx.training, fn_ctx.shapes = tuple(sizes), bool(training), shapes
        return q, k, v

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, dq, dk, dv):
        lib = _lib.load()
        (x, wq, wk, wv, bnq_w, bnq_b, bnk_w, bnk_b, wpk, wpv, zqk, vv, kk, pooled_k, pooled_v, mean,
         invstd) = fn_ctx.saved_tensors
        B, C, H, W = x.shape
        Kc, Vc, ns = wq.shape[0], wv.shape[0], len(fn_ctx.sizes)
        sz = _sizes_arg(fn_ctx.sizes)
        dims = (B, C, Kc, Vc, H, W, ns, sz)
        dq = _f32c(dq) if dq is not None else torch.zeros(B, Kc, H * W, device=x.device)
        dk = _f32c(dk) if dk is not None else torch.zeros(B, Kc, H * W, device=x.device)
        dv = _f32c(dv) if dv is not None else torch.zeros(B, Vc, H * W, device=x.device)
        dx = torch.empty_like(x)
        dwqk, dwv = torch.empty(2 * Kc, C, device=x.device), torch.empty(Vc, C, device=x.device)
        dbn = [torch.empty(Kc, device=x.device) for _ in range(4)]
        dwpk, dwpv = torch.empty_like(wpk), torch.empty_like(wpv)
        ws, nbytes = _workspace(lib.cabinet_cab_qkv_bwd_workspace_bytes(*dims), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_cab_qkv_bwd(_ptr(dq), _ptr(dk), _ptr(dv), _ptr(x), _ptr(wq), _ptr(wk), _ptr(wv),
                                         _ptr(bnq_w), _ptr(bnq_b), _ptr(bnk_w), _ptr(bnk_b), _ptr(wpk), _ptr(wpv),
                                         _ptr(zqk), _ptr(vv), _ptr(kk), _ptr(pooled_k), _ptr(pooled_v), _ptr(mean),
                                         _ptr(invstd), *dims, int(fn_ctx.training), _ptr(dx), _ptr(dwqk), _ptr(dwv),
                                         _ptr(dbn[0]), _ptr(dbn[1]), _ptr(dbn[2]), _ptr(dbn[3]), _ptr(dwpk),
                                         _ptr(dwpv), _ptr(ws), nbytes, _stream_handle(x.device))
        _lib.check(rc, "cabinet_cab_qkv_bwd")
        s_q, s_k, s_v, s_pk, s_pv = fn_ctx.shapes
        return (dx, dwqk[:Kc].view(s_q), dwqk[Kc:].view(s_k), dwv.view(s_v), dbn[0], dbn[1], dbn[2], dbn[3],
                dwpk.view(s_pk), dwpv.view(s_pv)) + (None,) * 8


def cab_qkv(x, gca):
    """q, k, v of GlobalContextAttention (reference cab.py:137-146) from its sub-modules, on device tensors.

    ``gca`` owns to_query / to_key (Conv2d 1x1, BatchNorm2d, ReLU), to_value (Conv2d 1x1), psp_key / psp_value
    (PSPModule); BatchNorm running buffers are updated in place in training mode."""
    bn_q, bn_k = gca.to_query[1], gca.to_key[1]
    sizes = [int(st.output_size[0]) for st in gca.psp_key.stages]
    if sizes != [int(st.output_size[0]) for st in gca.psp_value.stages]:
        raise RuntimeError("cab_qkv: psp_key and psp_value must use the same pyramid sizes")
    if (bn_q.eps, bn_q.momentum, bn_q.training) != (bn_k.eps, bn_k.momentum, bn_k.training):
        raise RuntimeError("cab_qkv: the two BatchNorm2d must share eps, momentum and training mode")
    training, momentum = _bn_step(bn_q)
    _bn_step(bn_k)
    return _CabQkv.apply(x, gca.to_query[0].weight, gca.to_key[0].weight, gca.to_value.weight, bn_q.weight, bn_q.bias,
                         bn_k.weight, bn_k.bias, gca.psp_key.project.weight, gca.psp_value.project.weight,
                         bn_q.running_mean, bn_q.running_var, bn_k.running_mean, bn_k.running_var, sizes, training,
                         momentum, bn_q.eps)


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, weight):
        lib = _lib.load()
        x = _f32c(x)
        Co, Ci = weight.shape[0], weight.shape[1]
        w2 = _f32c(weight).view(Co, Ci)
        B, P = x.shape[0], x[0, 0].numel()
        y = torch.empty((B, Co) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(lib.cabinet_conv1x1_fwd_workspace_bytes(Ci, Co), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_conv1x1_fwd(_ptr(x), _ptr(w2), B, Ci, Co, P, _ptr(y), _ptr(ws), nbytes,
                                         _stream_handle(x.device))
        _lib.check(rc, "cabinet_conv1x1_fwd")
        fn_ctx.save_for_backward(x, w2)
        fn_ctx.w_shape = weight.shape
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, w2 = fn_ctx.saved_tensors
        g = _f32c(g)
        Co, Ci = w2.shape
        B, P = x.shape[0], x[0, 0].numel()
        dx = torch.empty_like(x) if fn_ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w2) if fn_ctx.needs_input_grad[1] else None
        ws, nbytes = _workspace(lib.cabinet_conv1x1_bwd_workspace_bytes(B, Ci, Co, P), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_conv1x1_bwd(_ptr(g), _ptr(x), _ptr(w2), B, Ci, Co, P, _ptr(dx), _ptr(dw), _ptr(ws), nbytes,
                                         _stream_handle(x.device))
        _lib.check(rc, "cabinet_conv1x1_bwd")
        return dx, (dw.view(fn_ctx.w_shape) if dw is not None else None)


class _Conv1x1Bias(torch.autograd.Function):
    """1x1 convolution WITH bias on the small-grid path (``AttentionBranch.convb``, reference cabinet.py:65-66, :86)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, weight, bias):
        lib = _lib.load()
        x = _f32c(x)
        Co, Ci = weight.shape[0], weight.shape[1]
        w2 = _f32c(weight).view(Co, Ci)
        B, P = x.shape[0], x[0, 0].numel()
        y = torch.empty((B, Co) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(lib.cabinet_conv1x1_fwd_workspace_bytes(Ci, Co), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_conv1x1_bias_fwd(_ptr(x), _ptr(w2), _ptr(_f32c(bias)), B, Ci, Co, P, _ptr(y), _ptr(ws), nbytes,
                                              _stream_handle(x.device))
        _lib.check(rc, "cabinet_conv1x1_bias_fwd")
        fn_ctx.save_for_backward(x, w2)
        fn_ctx.w_shape = weight.shape
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, w2 = fn_ctx.saved_tensors
        g = _f32c(g)
        Co, Ci = w2.shape
        B, P = x.shape[0], x[0, 0].numel()
        dx = torch.empty_like(x) if fn_ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w2) if fn_ctx.needs_input_grad[1] else None
        db = torch.empty(Co, dtype=torch.float32, device=x.device) if fn_ctx.needs_input_grad[2] else None
        ws, nbytes = _workspace(lib.cabinet_conv1x1_bwd_workspace_bytes(B, Ci, Co, P), x.device)
        with torch.cuda.device(x.device):
            if dx is not None or dw is not None:
                rc = lib.cabinet_conv1x1_bwd(_ptr(g), _ptr(x), _ptr(w2), B, Ci, Co, P, _ptr(dx), _ptr(dw), _ptr(ws), nbytes,
                                             _stream_handle(x.device))
                _lib.check(rc, "cabinet_conv1x1_bwd")
            if db is not None:
                _lib.check(lib.cabinet_channel_sum(_ptr(g), B, Co, P, _ptr(db), _stream_handle(x.device)), "cabinet_channel_sum")
        return dx, (dw.view(fn_ctx.w_shape) if dw is not None else None), db


def conv1x1(x, weight, bias=None):
    """1x1 convolution of a (B,Ci,...) device tensor with a (Co,Ci[,1,1]) weight (reference cab.py:155; with ``bias``:
    cabinet.py:65-66, ``convb``).  The bias form exists on the small-grid path only (``conv1x1_bias_supported``)."""
    if not x.is_cuda:
        raise RuntimeError("conv1x1: device tensors only")
    if bias is not None:
        return _Conv1x1Bias.apply(x, weight, bias)
    return _Conv1x1.apply(x, weight)


def conv1x1_bias_supported(x, conv):
    """True when ``conv(x)`` -- a plain 1x1 nn.Conv2d with bias -- runs through the small-grid MFMA path (the CAB's resolution)."""
    if not (x.is_cuda and x.dim() == 4 and conv.bias is not None and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels == x.shape[1]):
        return False
    if conv.in_channels % 4 or conv.out_channels % 4:
        return False
    return bool(_lib.load().cabinet_conv1x1_bias_supported(int(x.shape[0]), int(conv.in_channels), int(conv.out_channels),
                                                           int(x.shape[2] * x.shape[3])))


# --------------------------------------------------------------------------- dense 3x3 convolution (K11, Winograd on the fp32 MFMA)


# CABINET_CONV3X3=0: the model's plain 3x3 convolutions go back to the stock (MIOpen) operator -- same-box A/B timing only
CONV3X3_ENABLED = _os.environ.get("CABINET_CONV3X3", "1") != "0"


def conv3x3_supported(C0, C1, Co, H=None, W=None):
    """True when K11 takes ``nn.Conv2d(C0 + C1, Co, 3, padding=1, bias=False)`` over inputs of C0 (+ C1) channels -- and, when the
    spatial size is given, over images of H x W: zero padding rides on the buffer range check, which keeps one image's tensors below
    1 GiB (conv3x3_wino.hip: WN_OOB).  A layer past that limit takes the stock convolution instead of raising."""
    if not _lib.load().cabinet_conv3x3_supported(int(C0), int(C1), int(Co)):
        return False
    if H is None or W is None:
        return True
    return 4 * max(int(C0), int(C1), int(Co)) * int(H) * int(W) < (1 << 30)


def conv3x3_fwd_hip(x0, x1, weight, bn_part=None):
    lib = _lib.load()
    B, C0, H, W = x0.shape
    C1 = x1.shape[1] if x1 is not None else 0
    Co = weight.shape[0]
    y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x0.device)
    ws, nbytes = _workspace(lib.cabinet_conv3x3_fwd_workspace_bytes(B, C0, C1, Co, H, W), x0.device)
    with torch.cuda.device(x0.device):
        rc = lib.cabinet_conv3x3_fwd(_ptr(x0), _ptr(x1), _ptr(weight), B, C0, C1, Co, H, W, _ptr(y), _ptr(bn_part), _ptr(ws), nbytes,
                                     _stream_handle(x0.device))
    _lib.check(rc, "cabinet_conv3x3_fwd")
    return y


def conv3x3_bwd_hip(g, x0, x1, weight, need_dx=True, need_dw=True):
    lib = _lib.load()
    B, C0, H, W = x0.shape
    C1 = x1.shape[1] if x1 is not None else 0
    Co = weight.shape[0]
    dx0 = torch.empty_like(x0) if need_dx else None
    dx1 = torch.empty_like(x1) if need_dx and x1 is not None else None
    dw = torch.empty_like(weight) if need_dw else None
    ws, nbytes = _workspace(lib.cabinet_conv3x3_bwd_workspace_bytes(B, C0, C1, Co, H, W), x0.device)
    with torch.cuda.device(x0.device):
        rc = lib.cabinet_conv3x3_bwd(_ptr(g), _ptr(x0), _ptr(x1), _ptr(weight), B, C0, C1, Co, H, W, _ptr(dx0), _ptr(dx1), _ptr(dw),
                                     _ptr(ws), nbytes, _stream_handle(x0.device))
    _lib.check(rc, "cabinet_conv3x3_bwd")
    return dx0, dx1, dw


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x0, x1, weight, bn_part=None):
        x0, weight = _f32c(x0), _f32c(weight)
        x1 = _f32c(x1) if x1 is not None else None
        y = conv3x3_fwd_hip(x0, x1, weight, bn_part)
        fn_ctx.save_for_backward(x0, x1, weight)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        x0, x1, weight = fn_ctx.saved_tensors
        need = fn_ctx.needs_input_grad
        dx0, dx1, dw = conv3x3_bwd_hip(_f32c(g), x0, x1, weight, need_dx=need[0] or (x1 is not None and need[1]), need_dw=need[2])
        return (dx0, dx1, dw) + (None,) * (len(need) - 3)


def conv3x3_bn_part(x, out_channels):
    """The (2, Co, blocks) buffer K11's epilogue fills with per-block (mean, M2) of its output for ``bn_act(..., conv_part=)``."""
    B, _, H, W = x.shape
    return torch.empty((2, out_channels, _lib.load().cabinet_conv3x3_tile_blocks(B, H, W)), dtype=torch.float32, device=x.device)


def conv3x3(x, weight, x1=None, bn_part=None):
    """``F.conv2d(cat([x, x1], 1), weight, padding=1)`` (``x1`` optional) for a bias-free 3x3 stride-1 convolution of device
    tensors, without the concat: reference cabinet.py:59, :68 + :88-89, :160.  ``bn_part`` (from :func:`conv3x3_bn_part`) receives
    the statistics partials of the output for the training-mode BatchNorm that follows."""
    if not x.is_cuda:
        raise RuntimeError("conv3x3: device tensors only")
    if bn_part is not None:
        return _Conv3x3.apply(x, x1, weight, bn_part)
    return _Conv3x3.apply(x, x1, weight)


# --------------------------------------------------------------------------- BatchNorm2d + activation (K7)

_ACT_CODES = {None: 0, "none": 0, "relu": 1, "hardswish": 2}


class _BnAct(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, weight, bias, run_mean, run_var, act, training, momentum, eps, residual=None, conv_part=None):
        lib = _lib.load()
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        residual = _f32c(residual) if residual is not None else None
        B, C = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        if conv_part is not None and x.dim() == 4:
            # x comes straight out of K11 with its per-block (mean, M2) pairs: no statistics pass over x
            with torch.cuda.device(x.device):
                rc = lib.cabinet_bn_act_fwd_part(_ptr(x), _ptr(conv_part), _ptr(weight), _ptr(bias), _ptr(run_mean), _ptr(run_var),
                                                 _ptr(residual), B, C, x.shape[2], x.shape[3], act, int(training), float(momentum),
                                                 float(eps), _ptr(y), _ptr(mean), _ptr(invstd), _stream_handle(x.device))
            _lib.check(rc, "cabinet_bn_act_fwd_part")
        else:
            ws, nbytes = _workspace(lib.cabinet_bn_act_workspace_bytes(B, C, P), x.device)
            with torch.cuda.device(x.device):
                rc = lib.cabinet_bn_act_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(run_mean), _ptr(run_var), _ptr(residual), B, C, P,
                                            act, int(training), float(momentum), float(eps), _ptr(y), _ptr(mean), _ptr(invstd),
                                            _ptr(ws), nbytes, _stream_handle(x.device))
            _lib.check(rc, "cabinet_bn_act_fwd")
        fn_ctx.save_for_backward(x, weight, bias, mean, invstd)
        fn_ctx.act, fn_ctx.training = act, bool(training)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, weight, bias, mean, invstd = fn_ctx.saved_tensors
        g = _f32c(g)
        B, C = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        dx, dw, db = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(bias)
        ws, nbytes = _workspace(lib.cabinet_bn_act_workspace_bytes(B, C, P), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_bn_act_bwd(_ptr(g), _ptr(x), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(invstd), B, C, P,
                                        fn_ctx.act, int(fn_ctx.training), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws),
                                        nbytes, _stream_handle(x.device))
        _lib.check(rc, "cabinet_bn_act_bwd")
        grads = (dx, dw, db, None, None, None, None, None, None)
        if len(fn_ctx.needs_input_grad) > 9:  # called with the optional residual operand: d(residual) = dy
            grads += (g if fn_ctx.needs_input_grad[9] else None,)
        if len(fn_ctx.needs_input_grad) > 10:  # ... and the producer's statistics partials (a buffer, not a variable)
            grads += (None,)
        return grads


def bn_act(x, bn, act=None, residual=None, conv_part=None):
    """act(bn(x)) [+ residual] for a device tensor; ``bn`` is the nn.BatchNorm2d owning parameters and running
    buffers (updated in place in training mode), ``act`` one of None / "relu" / "hardswish"
    (reference cabinet.py:42-44, mobilenetv3.py:86-99); ``residual`` is the MBConv identity shortcut
    (mobilenetv3.py:158), added in the same pass.  ``conv_part``: the (2, C, blocks) statistics partials ``conv3x3`` wrote
    while it produced ``x`` -- the batch statistics then cost no pass over ``x``."""
    if not x.is_cuda:
        raise RuntimeError("bn_act: device tensors only (host tensors take the composite ATen path)")
    if act not in _ACT_CODES:
        raise RuntimeError(f"bn_act: unknown activation {act!r}")
    training, momentum = _bn_step(bn)
    if residual is not None and residual.shape != x.shape:
        raise RuntimeError(f"bn_act: residual {tuple(residual.shape)} does not match x {tuple(x.shape)}")
    if conv_part is not None:
        return _BnAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, _ACT_CODES[act], training, momentum,
                            bn.eps, residual, conv_part)
    return _BnAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, _ACT_CODES[act], training, momentum,
                        bn.eps, residual)


# --------------------------------------------------------------------------- BatchNorm + ReLU + 1x1 classifier (K12)

# CABINET_BN_CLS=0 keeps the round-4 pair (K7 BatchNorm + ReLU, then the stock 1x1 convolution): A/B timing
BN_CLS_ENABLED = _os.environ.get("CABINET_BN_CLS", "1") != "0"


class _BnCls(torch.autograd.Function):
    """``conv1x1(relu(bn(z)))`` as one streaming operator (cabinet_bn_cls_fwd / _bwd): the activation and its gradient are never
    written.  Saved for backward: z and the per-channel table (classifier column, mean, invstd, gamma, beta)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, z, bn_w, bn_b, run_mean, run_var, w_cls, bias, training, momentum, eps, conv_part=None):
        lib = _lib.load()
        z = _f32c(z)
        B, C, H, W = z.shape
        K = w_cls.shape[0]
        w2 = _f32c(w_cls.detach()).reshape(K, C)
        y = torch.empty((B, K, H, W), dtype=torch.float32, device=z.device)
        tab = torch.empty(lib.cabinet_bn_cls_table_floats(C, K), dtype=torch.float32, device=z.device)
        ws, nbytes = _workspace(lib.cabinet_bn_cls_fwd_workspace_bytes(B, C, H * W), z.device)
        with torch.cuda.device(z.device):
            rc = lib.cabinet_bn_cls_fwd(_ptr(z), _ptr(conv_part), _ptr(bn_w), _ptr(bn_b), _ptr(run_mean), _ptr(run_var), _ptr(w2),
                                        _ptr(bias), B, C, K, H, W, int(training), float(momentum), float(eps), _ptr(y), _ptr(tab),
                                        _ptr(ws), nbytes, _stream_handle(z.device))
        _lib.check(rc, "cabinet_bn_cls_fwd")
        fn_ctx.save_for_backward(z, tab)
        fn_ctx.training, fn_ctx.w_shape, fn_ctx.has_bias = bool(training), w_cls.shape, bias is not None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        z, tab = fn_ctx.saved_tensors
        g = _f32c(g)
        B, C, H, W = z.shape
        K = g.shape[1]
        dz = torch.empty_like(z)
        dbn_w = torch.empty(C, dtype=torch.float32, device=z.device)
        dbn_b = torch.empty(C, dtype=torch.float32, device=z.device)
        dw = torch.empty((K, C), dtype=torch.float32, device=z.device)
        db = torch.empty(K, dtype=torch.float32, device=z.device) if fn_ctx.has_bias else None
        ws, nbytes = _workspace(lib.cabinet_bn_cls_bwd_workspace_bytes(B, C, K, H * W), z.device)
        with torch.cuda.device(z.device):
            rc = lib.cabinet_bn_cls_bwd(_ptr(g), _ptr(z), _ptr(tab), B, C, K, H, W, int(fn_ctx.training), _ptr(dz), _ptr(dbn_w),
                                        _ptr(dbn_b), _ptr(dw), _ptr(db), _ptr(ws), nbytes, _stream_handle(z.device))
        _lib.check(rc, "cabinet_bn_cls_bwd")
        grads = (dz, dbn_w, dbn_b, None, None, dw.view(fn_ctx.w_shape), db, None, None, None)
        if len(fn_ctx.needs_input_grad) > 10:  # called with the producer's statistics partials (a buffer, not a variable)
            grads += (None,)
        return grads


def bn_relu_cls_supported(z, bn, cls):
    """True when ``cls(relu(bn(z)))`` runs as K12: a device (B,C,H,W) tensor, ``cls`` a plain 1x1 nn.Conv2d (stride 1, no padding,
    one group; bias optional), C % 64 == 0, at most 32 classes, H*W % 4 == 0."""
    if not (BN_CLS_ENABLED and z.is_cuda and z.dim() == 4):
        return False
    if not (cls.kernel_size == (1, 1) and cls.stride == (1, 1) and cls.padding == (0, 0) and cls.dilation == (1, 1)
            and cls.groups == 1 and cls.in_channels == z.shape[1] and bn.num_features == z.shape[1] and bn.affine
            and bn.track_running_stats):
        return False
    return bool(_lib.load().cabinet_bn_cls_supported(int(z.shape[1]), int(cls.out_channels), int(z.shape[2] * z.shape[3])))


def bn_relu_cls(z, bn, cls, conv_part=None):
    """``cls(relu(bn(z)))`` for a device tensor: ``bn`` the nn.BatchNorm2d (running buffers updated in place in training mode),
    ``cls`` the 1x1 nn.Conv2d classifier behind it (reference cabinet.py:90-92 and :161-172).  ONE operator (K12) where
    :func:`bn_relu_cls_supported`, K7 + the stock convolution otherwise.  ``conv_part``: the statistics partials ``conv3x3`` wrote
    while it produced ``z``."""
    if not bn_relu_cls_supported(z, bn, cls):
        return cls(bn_act(z, bn, "relu", conv_part=conv_part))
    training, momentum = _bn_step(bn)
    args = (z, bn.weight, bn.bias, bn.running_mean, bn.running_var, cls.weight, cls.bias, training, momentum, bn.eps)
    return _BnCls.apply(*args, conv_part) if conv_part is not None else _BnCls.apply(*args)


# --------------------------------------------------------------------------- depthwise convolution (K8)


def dwconv_supported(conv):
    """True for the depthwise nn.Conv2d shapes the HIP kernels cover (reference mobilenetv3.py:118-126)."""
    k, s = conv.kernel_size, conv.stride
    return (conv.groups == conv.in_channels == conv.out_channels and conv.bias is None and k[0] == k[1]
            and s[0] == s[1] and conv.padding == (k[0] // 2, k[0] // 2) and conv.dilation == (1, 1)
            and conv.padding_mode == "zeros" and k[0] in (3, 5) and s[0] in (1, 2))


class _DwConv(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, weight, stride):
        lib = _lib.load()
        x, w = _f32c(x), _f32c(weight)
        B, C, H, W = x.shape
        K = w.shape[-1]
        Ho, Wo = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_dwconv_fwd(_ptr(x), _ptr(w), B, C, H, W, K, stride, _ptr(y), _stream_handle(x.device))
        _lib.check(rc, "cabinet_dwconv_fwd")
        fn_ctx.save_for_backward(x, w)
        fn_ctx.stride = stride
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, w = fn_ctx.saved_tensors
        g = _f32c(g)
        B, C, H, W = x.shape
        K, stride = w.shape[-1], fn_ctx.stride
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        ws, nbytes = _workspace(lib.cabinet_dwconv_bwd_workspace_bytes(B, C, H, W, K, stride), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_dwconv_bwd(_ptr(g), _ptr(x), _ptr(w), B, C, H, W, K, stride, _ptr(dx), _ptr(dw), _ptr(ws),
                                        nbytes, _stream_handle(x.device))
        _lib.check(rc, "cabinet_dwconv_bwd")
        return dx, dw, None


def dwconv(x, conv):
    """Depthwise convolution of a device tensor with the weights of ``conv`` (an nn.Conv2d, see dwconv_supported)."""
    if not x.is_cuda:
        raise RuntimeError("dwconv: device tensors only")
    return _DwConv.apply(x, conv.weight, conv.stride[0])


# --------------------------------------------------------------------------- channel gate + activation (SE tail)


class _GateAct(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, gate, act):
        lib = _lib.load()
        x, gate = _f32c(x), _f32c(gate)
        B, C = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_gate_act_fwd(_ptr(x), _ptr(gate), B, C, P, act, _ptr(y), _stream_handle(x.device))
        _lib.check(rc, "cabinet_gate_act_fwd")
        fn_ctx.save_for_backward(x, gate)
        fn_ctx.act = act
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, gate = fn_ctx.saved_tensors
        g = _f32c(g)
        B, C = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        dx, dgate = torch.empty_like(x), torch.empty_like(gate)
        ws, nbytes = _workspace(lib.cabinet_gate_act_bwd_workspace_bytes(B, C, P), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_gate_act_bwd(_ptr(g), _ptr(x), _ptr(gate), B, C, P, fn_ctx.act, _ptr(dx), _ptr(dgate),
                                          _ptr(ws), nbytes, _stream_handle(x.device))
        _lib.check(rc, "cabinet_gate_act_bwd")
        return dx, dgate, None


def gate_act(x, gate, act=None):
    """act(x * gate[:, :, None, None]) for a device tensor x (B,C,H,W) and a (B,C) gate: the tail of SELayer.forward
    (reference mobilenetv3.py:79-83) fused with the activation that follows it in the MBConv block."""
    if not x.is_cuda:
        raise RuntimeError("gate_act: device tensors only")
    if act not in _ACT_CODES:
        raise RuntimeError(f"gate_act: unknown activation {act!r}")
    if gate.shape != x.shape[:2]:
        raise RuntimeError(f"gate_act: gate {tuple(gate.shape)} does not match x {tuple(x.shape)}")
    return _GateAct.apply(x, gate, _ACT_CODES[act])


# --------------------------------------------------------------------------- BatchNorm (+act) -> depthwise conv, fused


class _BnActDwConv(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, z, bn_w, bn_b, run_mean, run_var, conv_w, act, stride, training, momentum, eps):
        lib = _lib.load()
        z, bn_w, bn_b, cw = _f32c(z), _f32c(bn_w), _f32c(bn_b), _f32c(conv_w)
        B, C, H, W = z.shape
        K = cw.shape[-1]
        Ho, Wo = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=z.device)
        mean = torch.empty(C, dtype=torch.float32, device=z.device)
        invstd = torch.empty(C, dtype=torch.float32, device=z.device)
        ws, nbytes = _workspace(lib.cabinet_bn_dwconv_fwd_workspace_bytes(B, C, H, W), z.device)
        with torch.cuda.device(z.device):
            rc = lib.cabinet_bn_dwconv_fwd(_ptr(z), _ptr(bn_w), _ptr(bn_b), _ptr(run_mean), _ptr(run_var), _ptr(cw), B, C,
                                           H, W, K, stride, act, int(training), float(momentum), float(eps), _ptr(y),
                                           _ptr(mean), _ptr(invstd), _ptr(ws), nbytes, _stream_handle(z.device))
        _lib.check(rc, "cabinet_bn_dwconv_fwd")
        fn_ctx.save_for_backward(z, bn_w, bn_b, cw, mean, invstd)
        fn_ctx.meta = (act, stride, bool(training))
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        z, bn_w, bn_b, cw, mean, invstd = fn_ctx.saved_tensors
        act, stride, training = fn_ctx.meta
        g = _f32c(g)
        B, C, H, W = z.shape
        K = cw.shape[-1]
        dz, dbw, dbb, dcw = torch.empty_like(z), torch.empty_like(bn_w), torch.empty_like(bn_b), torch.empty_like(cw)
        ws, nbytes = _workspace(lib.cabinet_bn_dwconv_bwd_workspace_bytes(B, C, H, W, K, stride), z.device)
        with torch.cuda.device(z.device):
            rc = lib.cabinet_bn_dwconv_bwd(_ptr(g), _ptr(z), _ptr(bn_w), _ptr(bn_b), _ptr(mean), _ptr(invstd), _ptr(cw), B,
                                           C, H, W, K, stride, act, int(training), _ptr(dz), _ptr(dbw), _ptr(dbb),
                                           _ptr(dcw), _ptr(ws), nbytes, _stream_handle(z.device))
        _lib.check(rc, "cabinet_bn_dwconv_bwd")
        return dz, dbw, dbb, None, None, dcw, None, None, None, None, None


def bn_act_dwconv(z, bn, act, conv):
    """conv(act(bn(z))) for a depthwise ``conv`` (see dwconv_supported) on a device tensor, without materialising the
    normalised tensor (reference mobilenetv3.py:135-143)."""
    if not z.is_cuda:
        raise RuntimeError("bn_act_dwconv: device tensors only")
    if act not in _ACT_CODES:
        raise RuntimeError(f"bn_act_dwconv: unknown activation {act!r}")
    training, momentum = _bn_step(bn)
    return _BnActDwConv.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.weight, _ACT_CODES[act],
                              conv.stride[0], training, momentum, bn.eps)


# --------------------------------------------------------------------------- 7x7 stride-2 stem convolution (K9)


def stem_conv_supported(conv):
    """True for the spatial branch's first convolution (reference cabinet.py:111): 3 -> 64, 7x7, stride 2, pad 3."""
    return (conv.in_channels == 3 and conv.out_channels == 64 and conv.kernel_size == (7, 7) and conv.stride == (2, 2)
            and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.padding_mode == "zeros")


class _StemConv(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, weight):
        lib = _lib.load()
        x, w = _f32c(x), _f32c(weight)
        B, _, H, W = x.shape
        y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_stem_conv_fwd(_ptr(x), _ptr(w), B, H, W, _ptr(y), _stream_handle(x.device))
        _lib.check(rc, "cabinet_stem_conv_fwd")
        fn_ctx.save_for_backward(x, w)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, w = fn_ctx.saved_tensors
        g = _f32c(g)
        B, _, H, W = x.shape
        dx = None
        if fn_ctx.needs_input_grad[0]:  # not the training step (the input is the image); served by ATen when asked for
            dx = torch.ops.aten.convolution_backward(g, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        dw = torch.empty_like(w)
        ws, nbytes = _workspace(lib.cabinet_stem_conv_wrw_workspace_bytes(B, H, W), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_stem_conv_wrw(_ptr(g), _ptr(x), B, H, W, _ptr(dw), _ptr(ws), nbytes,
                                           _stream_handle(x.device))
        _lib.check(rc, "cabinet_stem_conv_wrw")
        return dx, dw


def stem_conv(x, conv):
    """The spatial branch's 7x7/2 stem convolution on a device tensor (see stem_conv_supported)."""
    if not x.is_cuda:
        raise RuntimeError("stem_conv: device tensors only")
    return _StemConv.apply(x, conv.weight)


# --------------------------------------------------------------------------- thin pointwise convolution (K10)


def pwconv_supported(conv, x):
    """True for the bias-free stride-1 1x1 convolutions the streaming kernels cover (thin channel counts on large
    planes, reference mobilenetv3.py:128-131,144-151); others stay with MIOpen, which is as fast there."""
    if not (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.bias is None and conv.dilation == (1, 1)):
        return False
    P = x.shape[2] * x.shape[3]
    # measured on MI355X (tools/time_pwconv*.py): the streaming kernels win where the layer is HBM-bound -- planes of
    # >= 256 x 256 pixels with <= 96 channels either side (2.5x forward / input gradient, 1.3x weight gradient)
    return (P >= 65536 and conv.in_channels <= 96 and conv.out_channels <= 96
            and bool(_lib.load().cabinet_pwconv_supported(conv.in_channels, conv.out_channels, P)))


class _PwConv(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(fn_ctx, x, weight):
        lib = _lib.load()
        x = _f32c(x)
        Co, Ci = weight.shape[0], weight.shape[1]
        w2 = _f32c(weight).view(Co, Ci)
        B, P = x.shape[0], x.shape[2] * x.shape[3]
        y = torch.empty((B, Co) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_pwconv_fwd(_ptr(x), _ptr(w2), B, Ci, Co, P, _ptr(y), _stream_handle(x.device))
        _lib.check(rc, "cabinet_pwconv_fwd")
        fn_ctx.save_for_backward(x, w2)
        fn_ctx.w_shape = weight.shape
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(fn_ctx, g):
        lib = _lib.load()
        x, w2 = fn_ctx.saved_tensors
        g = _f32c(g)
        Co, Ci = w2.shape
        B, P = x.shape[0], x.shape[2] * x.shape[3]
        dx = torch.empty_like(x) if fn_ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w2) if fn_ctx.needs_input_grad[1] else None
        ws, nbytes = _workspace(lib.cabinet_pwconv_bwd_workspace_bytes(B, Ci, Co, P), x.device)
        with torch.cuda.device(x.device):
            rc = lib.cabinet_pwconv_bwd(_ptr(g), _ptr(x), _ptr(w2), B, Ci, Co, P, _ptr(dx), _ptr(dw), _ptr(ws), nbytes,
                                        _stream_handle(x.device))
        _lib.check(rc, "cabinet_pwconv_bwd")
        return dx, (dw.view(fn_ctx.w_shape) if dw is not None else None)


def pwconv(x, conv):
    """Thin pointwise convolution of a device tensor with the weights of ``conv`` (see pwconv_supported)."""
    if not x.is_cuda:
        raise RuntimeError("pwconv: device tensors only")
    return _PwConv.apply(x, conv.weight)
