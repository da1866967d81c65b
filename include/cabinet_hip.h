/*
 * cabinet_hip.h -- C ABI of libcabinet_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the CABiNet hot path.  The reference (dronefreak/CABiNet)
 * is pure Python/PyTorch and has no native interface; each entry point below
 * replaces a span of ATen calls inside a reference nn.Module.forward (cited
 * per function, paths relative to the reference repo) plus the autograd
 * backward of that span.  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add at those lines.
 *
 * Conventions
 *  - All tensors are dense fp32, NCHW-contiguous, resident on the current HIP
 *    device.  Pointers are borrowed; nothing is retained after the call returns.
 *  - Tensor pointers must be 16-byte aligned: the kernels move rows with 128-bit
 *    loads and stores.  (Any allocator's base pointer is; a view that starts 4 or
 *    8 bytes into an allocation is not -- copy it first.)  The attention, OHEM, FFM,
 *    q/k/v producer, 1x1 / 3x3 convolution and BatchNorm entry points check the
 *    activation and weight pointers they move that way and return
 *    CABINET_ERR_INVALID_ARG for a misaligned one.
 *  - Every call is asynchronous on `stream` (a hipStream_t; NULL = the default
 *    stream).  No call synchronises, allocates device memory or uses a private
 *    stream, so calls are hipGraph-capturable and re-entrant.
 *  - Scratch memory is caller-provided: query the size with the matching
 *    *_workspace_bytes() and pass a buffer at least that large (256-byte
 *    aligned).  A too-small buffer is an error, never an overflow.
 *  - Return value: CABINET_OK (0) or a negative CABINET_ERR_* code; the
 *    message for the calling thread is available from cabinet_last_error().
 *    Mirrors how the ATen ops they replace raise RuntimeError on bad
 *    shape/dtype/device.
 */
#ifndef CABINET_HIP_H_
#define CABINET_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CABINET_ABI_VERSION 7

#define CABINET_OK 0
#define CABINET_ERR_INVALID_ARG (-1) /* null pointer, non-positive dim            */
#define CABINET_ERR_UNSUPPORTED (-2) /* shape outside what the kernels implement  */
#define CABINET_ERR_WORKSPACE (-3)   /* workspace missing or too small            */
#define CABINET_ERR_HIP (-4)         /* a HIP runtime call failed                 */

typedef void* cabinet_stream_t; /* hipStream_t */

int cabinet_abi_version(void);
const char* cabinet_last_error(void);

/* ------------------------------------------------------------------------- *
 * CAB attention core: affinity matmul -> softmax over keys -> aggregation.
 * Replaces src/models/cab.py:149-154
 *     attn = torch.bmm(query, key); attn = attn * (key.shape[1] ** -0.5)
 *     attn = F.softmax(attn, dim=-1); context = torch.bmm(attn, value)
 *     context = context.transpose(1, 2).view(B, -1, Hd, Wd)
 * with q,k: (B,Kc,n) and v: (B,Vc,n) being the NCHW-flattened outputs of
 * to_query / psp_key / psp_value (cab.py:137-146) -- no transposes needed.
 *   ctx[b,c,i] = sum_j softmax_j(scale * sum_c' q[b,c',i] k[b,c',j]) v[b,c,j]
 *   lse[b,i]   = log sum_j exp(scale * S[b,i,j])       (saved for backward)
 * The n x n affinity matrix is never written to memory.
 * Instantiated for (Kc,Vc) in {(128,128), (256,128), (64,64)}; n >= 1 arbitrary.  cabinet_cab_attn_supported()
 * answers 1 / 0 for a channel pair; callers route other pairs to their composite path (the fwd / bwd entry points
 * return CABINET_ERR_UNSUPPORTED for them).
 *
 * `precision` selects the matrix arithmetic of the two contractions (same kernel structure, same operand layout trick):
 *   CABINET_PREC_FP32   (0)  v_mfma_f32_32x32x2_f32: exact fp32 products and sums (bit-wise an fma chain).  Default.
 *   CABINET_PREC_BF16X3 (1)  every fp32 operand split into 2 bf16 pieces, 3 bf16 MFMA products per fp32 product
 *                            (~2^-17 relative per product, ~1e-5 per tensor): 5.3x the fp32 matrix rate.  Measured variant.
 *   CABINET_PREC_BF16X6 (2)  3 pieces (fp32's 24-bit significand exactly), 6 products: fp32-level accuracy at 2.7x the rate.
 * The split forms exist for (Kc,Vc) in {(128,128), (64,64)} (cabinet_cab_attn_precision_supported); their workspace
 * also holds the operands re-laid out as bf16 pieces in MFMA operand order (one pack pass per call).
 * ------------------------------------------------------------------------- */
#define CABINET_PREC_FP32 0
#define CABINET_PREC_BF16X3 1
#define CABINET_PREC_BF16X6 2
int cabinet_cab_attn_supported(int Kc, int Vc);
int cabinet_cab_attn_precision_supported(int Kc, int Vc, int precision);
size_t cabinet_cab_attn_fwd_workspace_bytes(int B, int Kc, int Vc, int n, int precision);
int cabinet_cab_attn_fwd(const float* q, const float* k, const float* v, float scale,
                         int B, int Kc, int Vc, int n, int precision,
                         float* ctx /* (B,Vc,n) */, float* lse /* (B,n) */,
                         void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* K1 with the CAB's output projection in its epilogue (round 5).  Replaces src/models/cab.py:149-155, i.e. the span above
 * plus   context = self.project_out(context)   (a bias-free 1x1 convolution, w_out: (Co,Vc) row-major):
 *   glob[b,o,i] = sum_c w_out[o,c] ctx[b,c,i]
 * applied to each 32-query context tile while the kernel still holds it in LDS: one launch instead of two, and ctx only
 * reaches memory when the caller asks for it (ctx != NULL: training saves it for cabinet_cab_attn_bwd and the projection's
 * weight gradient; pass NULL for inference).  fp32 MFMA only.  cabinet_cab_attn_proj_supported() answers 1 for the shapes
 * the fused form takes -- (Kc,Vc) in {(128,128), (64,64)}, n % 4 == 0, Co % 32 == 0, and a (B, n) the forward runs
 * without a key split (every BASELINE configuration with B * n/32 >= ~200 query tiles) -- callers use
 * cabinet_cab_attn_fwd + cabinet_conv1x1_fwd otherwise.  Backward: cabinet_conv1x1_bwd (dctx = w_out^T dglob,
 * dw_out = dglob ctx^T) followed by cabinet_cab_attn_bwd; no new entry point. */
int cabinet_cab_attn_proj_supported(int B, int Kc, int Vc, int Co, int n);
int cabinet_cab_attn_proj_fwd(const float* q, const float* k, const float* v, const float* w_out /* (Co,Vc) */,
                              float scale, int B, int Kc, int Vc, int Co, int n,
                              float* ctx /* (B,Vc,n) or NULL */, float* glob /* (B,Co,n) */, float* lse /* (B,n) */,
                              cabinet_stream_t stream);

/* Backward of the span above (what autograd derives for cab.py:149-154).
 * Recomputes the affinity tiles from q, k and lse; dctx is dL/dctx (B,Vc,n). */
size_t cabinet_cab_attn_bwd_workspace_bytes(int B, int Kc, int Vc, int n);
int cabinet_cab_attn_bwd(const float* dctx, const float* q, const float* k, const float* v,
                         const float* ctx, const float* lse, float scale,
                         int B, int Kc, int Vc, int n,
                         float* dq /* (B,Kc,n) */, float* dk /* (B,Kc,n) */, float* dv /* (B,Vc,n) */,
                         void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Feature Fusion Module.
 * Replaces src/models/cabinet.py:142-153 (FeatureFusionModule.forward) incl.
 * its ConvBNReLU (cabinet.py:42-44):
 *     fcat = cat([fsp, fcp], 1); feat = relu(bn(conv1x1(fcat)))
 *     atten = sigmoid(conv2(relu(conv1(avg_pool(feat))))); return feat*atten + feat
 * fsp: (B,Cs,H,W)  fcp: (B,Cc,H,W)  w_blk: (Co,Cs+Cc)  w1: (Cm,Co)  w2: (Co,Cm)
 * The concat is never materialised.  training != 0: BatchNorm uses batch
 * statistics and updates running_mean / running_var in place (momentum, unbiased
 * variance) exactly like nn.BatchNorm2d; training == 0: running statistics.
 * Saved for backward: z = pre-BN conv output (B,Co,H,W), save_mean / save_invstd
 * (Co), pooled (B,Co) = spatial mean of feat, gate (B,Co) = sigmoid output.
 * Requires Cs, Cc, Co multiples of 32, Cm <= 256.
 * ------------------------------------------------------------------------- */
size_t cabinet_ffm_fwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W);
int cabinet_ffm_fwd(const float* fsp, const float* fcp, const float* w_blk,
                    const float* bn_weight, const float* bn_bias,
                    float* running_mean, float* running_var,
                    const float* w1, const float* w2,
                    int B, int Cs, int Cc, int Co, int Cm, int H, int W,
                    int training, float momentum, float eps,
                    float* out, float* z, float* save_mean, float* save_invstd,
                    float* pooled, float* gate,
                    void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

size_t cabinet_ffm_bwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W);
int cabinet_ffm_bwd(const float* dout, const float* fsp, const float* fcp, const float* w_blk,
                    const float* bn_weight, const float* bn_bias,
                    const float* w1, const float* w2,
                    const float* z, const float* save_mean, const float* save_invstd,
                    const float* pooled, const float* gate,
                    int B, int Cs, int Cc, int Co, int Cm, int H, int W, int training,
                    float* dfsp, float* dfcp, float* dw_blk, float* dbn_weight, float* dbn_bias,
                    float* dw1, float* dw2,
                    void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Feature Fusion Module with the bilinear upsample of its context input fused in.
 * Replaces src/models/cabinet.py:228-230 + :236 inside CABiNet.forward
 *     low_res_logit_up = F.interpolate(low_res_logit, size=feat_sb.shape[2:], mode="bilinear",
 *                                      align_corners=False)
 *     feat_fuse = self.ffm(feat_sb, low_res_logit_up)
 * fsp: (B,Cs,H,W)   low: (B,Cc,Hl,Wl)  ->  out (B,Co,H,W); the (B,Cc,H,W) upsampled tensor is never
 * materialised.  The 1x1 conv and the resize commute (both linear, different indices), so the Cc part of
 * the conv runs at (Hl,Wl) and is added bilinearly in the GEMM epilogue; backward likewise returns dlow at
 * (Hl,Wl).  Everything else (BN, gate, saved tensors, argument meaning) is as cabinet_ffm_fwd/bwd.
 * `precision` (CABINET_PREC_*, see the attention section) selects the matrix arithmetic of the big product z = W_s . fsp +
 * U(W_c . low): the split-bf16 forms exist for Co % 256 == 0, Cs == 128, (H*W) % 128 == 0, W == 128, Wl == 32 (the model's
 * grid at 1024 x 1024); any other shape runs the exact fp32 MFMA product whatever is asked.
 * ------------------------------------------------------------------------- */
size_t cabinet_ffm_up_fwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl);
int cabinet_ffm_up_fwd(const float* fsp, const float* low, const float* w_blk,
                       const float* bn_weight, const float* bn_bias,
                       float* running_mean, float* running_var,
                       const float* w1, const float* w2,
                       int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl,
                       int training, float momentum, float eps, int precision,
                       float* out, float* z, float* save_mean, float* save_invstd,
                       float* pooled, float* gate,
                       void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

size_t cabinet_ffm_up_bwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl);
int cabinet_ffm_up_bwd(const float* dout, const float* fsp, const float* low, const float* w_blk,
                       const float* bn_weight, const float* bn_bias,
                       const float* w1, const float* w2,
                       const float* z, const float* save_mean, const float* save_invstd,
                       const float* pooled, const float* gate,
                       int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl, int training,
                       float* dfsp, float* dlow /* (B,Cc,Hl,Wl) */, float* dw_blk, float* dbn_weight,
                       float* dbn_bias, float* dw1, float* dw2,
                       void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * OHEM cross-entropy fused with the final bilinear upsample (one call per head).
 * Replaces src/models/cabinet.py:240-245 (F.interpolate of the (B,C,Hl,Wl) logits to (H,W), bilinear,
 * align_corners=False) + src/utils/loss.py:51-80 (per-pixel CE with ignore_index, OHEM selection, mean)
 * for the selection branch "at least n_min pixels have loss > thresh" (loss.py:74-75).  The other branch
 * (top-n_min) needs an order statistic; callers take the unfused path for it.
 *   fwd : loss_px (B,H,W) per-pixel CE (0 at ignored pixels); per-workgroup partials
 *         blk_cnt[2*i] = #valid, blk_cnt[2*i+1] = #(loss > thresh), blk_sum[i] = sum of those losses,
 *         i < cabinet_ohem_up_blocks(B,H,W); the caller reduces them (and decides the branch)
 *   bwd : dlogits_low (B,C,Hl,Wl) = coef * U^T[ sel * (softmax - onehot) ],  sel = valid & (loss_px > thresh),
 *         coef = upstream_grad / #selected.  Deterministic (no atomics).   C <= 32.
 * labels are int64 (torch.long), (B,H,W), each either ignore_lb or in [0, C).  A label outside that set is the
 * caller's error (F.cross_entropy asserts on it); the forward kernel reports it by writing blk_cnt[2*i] = -2^30 for
 * every workgroup that met one, so the reduced #valid is negative (valid as long as B*H*W < 2^30, W < 2^20).
 * ------------------------------------------------------------------------- */
int cabinet_ohem_up_blocks(int B, int H, int W);
int cabinet_ohem_up_fwd(const float* logits_low, const long long* labels,
                        int B, int C, int Hl, int Wl, int H, int W, float thresh, int ignore_lb,
                        float* loss_px, float* blk_sum, int* blk_cnt, cabinet_stream_t stream);
size_t cabinet_ohem_up_bwd_workspace_bytes(int B, int C, int Hl, int Wl, int H, int W);
/* The forward's partials reduced on the device (ABI v4): stats (nheads,3) double = per head [#valid, #(loss > thresh), sum of
 * those losses] from blk_sum (nheads,nblk) and blk_cnt (nheads,nblk,2), nblk = cabinet_ohem_up_blocks(B,H,W).  One launch in
 * place of the caller's own reductions (loss.py:66-75 takes these three numbers to pick its branch and form the mean); a
 * negative #valid reports an out-of-range label (see above).  Fixed summation order: bit-reproducible.                      */
int cabinet_ohem_stats(const float* blk_sum, const int* blk_cnt, int nheads, int nblk, double* stats, cabinet_stream_t stream);
int cabinet_ohem_up_bwd(const float* logits_low, const long long* labels, const float* loss_px,
                        int B, int C, int Hl, int Wl, int H, int W, float thresh, int ignore_lb, float coef,
                        float* dlogits_low, void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
/* BOTH loss heads of the step in one launch each way -- src/scripts/train.py:435 `criteria_p(out, lb) + criteria_16(out16, lb)`
 * on the two outputs of cabinet.py:240-245: same labels, same (B,C,Hl,Wl) logits shape, same thresh / ignore_lb.
 *   fwd : loss_px (2,B,H,W), blk_sum (2,nblk), blk_cnt (2,nblk,2): head 0 = logits_low_a, head 1 = logits_low_b; the label tile
 *         is read once and every thread runs the two heads' exp / log chains side by side
 *   bwd : dlogits_low (2,B,C,Hl,Wl); workspace = 2 x the single-head workspace                                              */
int cabinet_ohem_up_pair_fwd(const float* logits_low_a, const float* logits_low_b, const long long* labels,
                             int B, int C, int Hl, int Wl, int H, int W, float thresh, int ignore_lb,
                             float* loss_px, float* blk_sum, int* blk_cnt, cabinet_stream_t stream);
size_t cabinet_ohem_up_pair_bwd_workspace_bytes(int B, int C, int Hl, int Wl, int H, int W);
int cabinet_ohem_up_pair_bwd(const float* logits_low_a, const float* logits_low_b, const long long* labels,
                             const float* loss_px, int B, int C, int Hl, int Wl, int H, int W, float thresh, int ignore_lb,
                             float coef, float* dlogits_low, void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * CAB local branch + block output.
 * Replaces src/models/cab.py:175-184 (LocalAttention.forward: three DWConv = depthwise 3x3 conv (cab.py:36-45)
 * + BatchNorm2d + ReLU, sigmoid gate, x + x*mask) and, when `glob` is given, cab.py:213-216
 * (ContextAggregationBlock.forward: gamma * global + local).
 *   x, glob, out, dout, dx, dglob : (B,C,H,W).  Two forms behind the same entry points:
 *     B*H*W <= 8192  : one kernel each way, one channel lives in the LDS of one CU, no workspace (..._workspace_bytes = 0)
 *     anything larger: (e.g. B = 16 at 1024^2, or the un-tiled 4096x2160 validation frame of src/scripts/train.py:444-456,
 *                      n = 8704) tiled form, B * ceil(H / rows) workgroups per channel, BatchNorm batch statistics as
 *                      two-phase ordered reductions: 4 launches forward, 8 backward, workspace from ..._workspace_bytes
 *   dw_w[s] (C,9) depthwise weights (C,1,3,3), bn_weight[s] / bn_bias[s] / running_mean[s] / running_var[s] (C),
 *   s = 0..2: HOST arrays of three DEVICE pointers (the three DWConv stages own separate parameter tensors)
 *   gamma : device scalar (cab.py:208); glob == NULL  ->  out = x * (1 + sigmoid(mask)) only (gamma, dglob and
 *           dgamma_part are then ignored and may be NULL)
 *   training != 0 : batch statistics per channel (biased variance for normalisation, unbiased into running_var,
 *           running = (1-momentum)*running + momentum*stat), save_mean / save_invstd (3,C) hold them;
 *           training == 0 : running statistics are used (and copied into save_*).
 *   bwd : recomputes the chain from x (nothing else is saved); writes dx, dglob = gamma*dout,
 *         dgamma_part (C) = per-channel <dout, glob> (caller sums it: d gamma), ddw_w[s] (C,9),
 *         dbn_weight[s], dbn_bias[s] (C).  Deterministic (no atomics).
 * ------------------------------------------------------------------------- */
int cabinet_cab_local_supported(int B, int C, int H, int W);   /* 1 if one of the two forms serves the shape, else 0 */
size_t cabinet_cab_local_fwd_workspace_bytes(int B, int C, int H, int W);   /* 0 for the channel-resident form */
size_t cabinet_cab_local_bwd_workspace_bytes(int B, int C, int H, int W);
int cabinet_cab_local_fwd(const float* x, const float* glob, const float* gamma,
                          const float* const* dw_w, const float* const* bn_weight, const float* const* bn_bias,
                          float* const* running_mean, float* const* running_var,
                          int B, int C, int H, int W, int training, float momentum, float eps,
                          float* out, float* save_mean, float* save_invstd,
                          void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
int cabinet_cab_local_bwd(const float* dout, const float* x, const float* glob, const float* gamma,
                          const float* const* dw_w, const float* const* bn_weight, const float* const* bn_bias,
                          const float* save_mean, const float* save_invstd,
                          int B, int C, int H, int W, int training,
                          float* dx, float* dglob, float* dgamma_part,
                          float* const* ddw_w, float* const* dbn_weight, float* const* dbn_bias,
                          void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * q/k/v producers of the CAB global branch (1x1 projections + BatchNorm + ReLU + pyramid pooling).
 * Replaces src/models/cab.py:137,141,145 with the modules of cab.py:107-123 and PSPModule.forward cab.py:65-76:
 *   q = relu(bn_q(W_q x));  k = PSP_k(relu(bn_k(W_k x)));  v = PSP_v(W_v x)
 *   PSP(u) = W_p . cat[u, U(A_s u) for s in sizes]   (A_s = AdaptiveAvgPool2d((s,s)), U = bilinear resize to
 *   (H,W), align_corners=False); evaluated as W_p[:, :Kc] u + sum_s U(W_p[:, block s] A_s u), no concat.
 *   x (B,C,H,W); wq, wk (Kc,C); wv (Vc,C)  [Conv2d 1x1 weights, no bias]; bn*_ (Kc);
 *   wpk (Kc,(n_sizes+1)*Kc), wpv (Vc,(n_sizes+1)*Vc)  [PSP project weights, identity block first]
 *   sizes: HOST array of n_sizes (1..4) pyramid sizes (each 1..16); C, Kc, Vc multiples of 16; H*W <= ~8192
 *   outputs q, k (B,Kc,H*W), v (B,Vc,H*W) -- the NCHW-flattened operands cabinet_cab_attn_fwd takes
 *   saved for backward (caller-allocated): zqk (B,2Kc,H*W), vv (B,Vc,H*W), kk (B,Kc,H*W),
 *   pooled_k (B,n_sizes*Kc,NBp), pooled_v (B,n_sizes*Vc,NBp) with NBp = cabinet_cab_qkv_padded_bins(),
 *   save_mean / save_invstd (2Kc) = [bn_q | bn_k]
 *   BatchNorm semantics and running-stat updates as in cabinet_ffm_fwd (training flag, momentum, eps).
 *   bwd: dq, dk (B,Kc,H*W), dv (B,Vc,H*W) -> dx, dwqk (2Kc,C) = [dW_q; dW_k], dwv, dbn*, dwpk, dwpv.
 *   Deterministic (no atomics).
 * ------------------------------------------------------------------------- */
int cabinet_cab_qkv_supported(int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes);
int cabinet_cab_qkv_padded_bins(int n_sizes, const int* sizes);
size_t cabinet_cab_qkv_fwd_workspace_bytes(int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes);
int cabinet_cab_qkv_fwd(const float* x, const float* wq, const float* wk, const float* wv,
                        const float* bnq_weight, const float* bnq_bias, float* bnq_running_mean, float* bnq_running_var,
                        const float* bnk_weight, const float* bnk_bias, float* bnk_running_mean, float* bnk_running_var,
                        const float* wpk, const float* wpv,
                        int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes,
                        int training, float momentum, float eps,
                        float* q, float* k, float* v,
                        float* zqk, float* vv, float* kk, float* pooled_k, float* pooled_v,
                        float* save_mean, float* save_invstd,
                        void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
size_t cabinet_cab_qkv_bwd_workspace_bytes(int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes);
int cabinet_cab_qkv_bwd(const float* dq, const float* dk, const float* dv, const float* x,
                        const float* wq, const float* wk, const float* wv,
                        const float* bnq_weight, const float* bnq_bias, const float* bnk_weight, const float* bnk_bias,
                        const float* wpk, const float* wpv,
                        const float* zqk, const float* vv, const float* kk, const float* pooled_k,
                        const float* pooled_v, const float* save_mean, const float* save_invstd,
                        int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes, int training,
                        float* dx, float* dwqk, float* dwv,
                        float* dbnq_weight, float* dbnq_bias, float* dbnk_weight, float* dbnk_bias,
                        float* dwpk, float* dwpv,
                        void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Bias-free 1x1 convolution, y (B,Co,P) = W (Co,Ci) . x (B,Ci,P), on the exact-fp32 MFMA GEMMs.
 * Replaces src/models/cab.py:155 (project_out of the attention context).  Ci, Co multiples of 4.
 * bwd: dx = W^T dy (skipped if dx == NULL), dw = sum_{b,p} dy (x) x (skipped if dw == NULL).
 * ------------------------------------------------------------------------- */
size_t cabinet_conv1x1_fwd_workspace_bytes(int Ci, int Co);
int cabinet_conv1x1_fwd(const float* x, const float* w, int B, int Ci, int Co, int P, float* y,
                        void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
/* The same with an output bias (`AttentionBranch.convb`, src/models/cabinet.py:65-66, 86: a 1x1 convolution WITH bias on the CAB's
 * output) on the small-grid path (cabinet_conv1x1_bias_supported: the CAB's resolution; larger planes use the stock operator).
 * Backward: cabinet_conv1x1_bwd for dx / dw, cabinet_channel_sum(dy, B, Co, P, dbias) for the bias gradient
 * (dbias[c] = sum over images and positions, one workgroup per channel, fixed order). */
int cabinet_conv1x1_bias_supported(int B, int Ci, int Co, int P);
int cabinet_conv1x1_bias_fwd(const float* x, const float* w, const float* bias /* (Co) */, int B, int Ci, int Co, int P, float* y,
                             void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
int cabinet_channel_sum(const float* d /* (B,C,P) */, int B, int C, int P, float* out /* (C) */, cabinet_stream_t stream);
size_t cabinet_conv1x1_bwd_workspace_bytes(int B, int Ci, int Co, int P);
int cabinet_conv1x1_bwd(const float* dy, const float* x, const float* w, int B, int Ci, int Co, int P,
                        float* dx, float* dw, void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * BatchNorm2d fused with the activation that follows it (NCHW fp32, P = H*W).
 * Replaces the bn -> relu tail of ConvBNReLU.forward, src/models/cabinet.py:42-44, and the same
 * BatchNorm2d -> ReLU / HardSwish pairs at cabinet.py:59-63,67-68 and src/models/mobilenetv3.py:86-99,118-152
 * (HardSwish: x * relu6(x + 3) / 6, mobilenetv3.py:48-50,63-65).
 *   act: 0 = none, 1 = ReLU, 2 = HardSwish.   y = act(weight * xhat + bias) [+ residual], xhat = (x - mean) * invstd
 *   (residual: the identity shortcut `x + self.conv(x)` of an MBConv block, mobilenetv3.py:158, added in the same
 *   pass; its gradient is dy itself, so backward is unchanged)
 *   training != 0: batch statistics (biased variance; unbiased into running_var; running buffers updated with
 *   `momentum`), save_mean / save_invstd (C) receive them; training == 0: running statistics.
 *   bwd needs only x, save_mean, save_invstd (the pre-activation is recomputed):
 *     du = dy * act'(u);  dweight = sum du * xhat;  dbias = sum du;
 *     dx = weight * invstd * (du - mean(du) - xhat * mean(du * xhat))   (training)   or  weight * invstd * du  (eval)
 *   Deterministic (no atomics).  One workspace size serves both directions.
 * ------------------------------------------------------------------------- */
size_t cabinet_bn_act_workspace_bytes(int B, int C, int P);
int cabinet_bn_act_fwd(const float* x, const float* weight, const float* bias,
                       float* running_mean, float* running_var, const float* residual /* nullable, (B,C,P) */,
                       int B, int C, int P, int act, int training, float momentum, float eps,
                       float* y, float* save_mean, float* save_invstd,
                       void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
/* The same forward when x (B,C,H,W) was just produced by cabinet_conv3x3_fwd with `bn_part`: the batch statistics come from
 * those per-block (mean, M2) pairs (Chan-merged in double, as above) and the statistics pass over x is skipped.
 * conv_part: [2][C][cabinet_conv3x3_tile_blocks(B,H,W)] floats; ignored when training == 0. */
int cabinet_bn_act_fwd_part(const float* x, const float* conv_part, const float* weight, const float* bias,
                            float* running_mean, float* running_var, const float* residual /* nullable */,
                            int B, int C, int H, int W, int act, int training, float momentum, float eps,
                            float* y, float* save_mean, float* save_invstd, cabinet_stream_t stream);
int cabinet_bn_act_bwd(const float* dy, const float* x, const float* weight, const float* bias,
                       const float* save_mean, const float* save_invstd,
                       int B, int C, int P, int act, int training,
                       float* dx, float* dweight, float* dbias,
                       void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * BatchNorm2d -> ReLU -> 1x1 classifier convolution as ONE streaming operator (round 5, K12).
 * Replaces the tail of src/models/cabinet.py:156-172 (CABiNetOutput.forward: `self.conv_out(relu(bn(conv(x))))`, behind
 * cabinet_conv3x3_fwd) and of the fusion head, cabinet.py:90-92 (`self.b4(self.b3(self.b2(.)))`):
 *     y[b,k,p] = bias[k] + sum_c w_cls[k,c] * relu(bn_weight[c] * (z[b,c,p] - mean[c]) * invstd[c] + bn_bias[c])
 * z (B,C,H,W): the convolution output in front of the BatchNorm;  w_cls (K,C): the 1x1 classifier;  bias (K) or NULL.
 * The (B,C,H,W) activation and, in backward, its gradient are never written: forward reads z once and writes the K
 * logits; backward reads z and dy twice and writes dz.  training != 0: batch statistics -- from conv_part when given (the
 * (mean, M2) pairs cabinet_conv3x3_fwd left: [2][C][cabinet_conv3x3_tile_blocks(B,H,W)], no statistics pass), else from a
 * pass over z -- and running_mean / running_var updated like nn.BatchNorm2d; training == 0: running statistics.
 * `table` (cabinet_bn_cls_table_floats(C,K) floats) is written by fwd and read by bwd: per channel the classifier column
 * and [mean, invstd, bn_weight, bn_bias, bn_weight * invstd].  Covered: C % 64 == 0, K <= 32, (H*W) % 4 == 0
 * (cabinet_bn_cls_supported).  Deterministic (ordered partial sums, no atomics).
 *   bwd: du = (pre > 0) * sum_k w_cls[k,c] dy[k];  dbn_weight = sum du * xhat;  dbn_bias = sum du;
 *        dz = bn_weight * invstd * (du - mean(du) - xhat * mean(du * xhat))  (training; eval: bn_weight * invstd * du);
 *        dw_cls[k,c] = sum dy[k] * relu(pre);  dbias[k] = sum dy[k]  (dbias may be NULL).
 * ------------------------------------------------------------------------- */
int cabinet_bn_cls_supported(int C, int K, int P);
int cabinet_bn_cls_table_floats(int C, int K);
size_t cabinet_bn_cls_fwd_workspace_bytes(int B, int C, int P);
int cabinet_bn_cls_fwd(const float* z, const float* conv_part /* nullable */, const float* bn_weight, const float* bn_bias,
                       float* running_mean, float* running_var, const float* w_cls /* (K,C) */, const float* bias /* nullable */,
                       int B, int C, int K, int H, int W, int training, float momentum, float eps,
                       float* y /* (B,K,H,W) */, float* table,
                       void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
size_t cabinet_bn_cls_bwd_workspace_bytes(int B, int C, int K, int P);
int cabinet_bn_cls_bwd(const float* dy /* (B,K,H,W) */, const float* z, const float* table,
                       int B, int C, int K, int H, int W, int training,
                       float* dz, float* dbn_weight, float* dbn_bias, float* dw_cls, float* dbias /* nullable */,
                       void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Depthwise KxK convolution (groups == channels, no bias, dilation 1, padding K/2), NCHW fp32.
 * Replaces the depthwise nn.Conv2d of the MBConv blocks, src/models/mobilenetv3.py:118-126,135-143
 * (K in {3,5}, stride in {1,2}); weight is the Conv2d weight (C,1,K,K).
 *   fwd : y (B,C,Ho,Wo), Ho = (H + 2*(K/2) - K)/stride + 1
 *   bwd : dx (B,C,H,W) and dw (C,1,K,K) from dy, x, weight in one pass over dy; deterministic (no atomics)
 * ------------------------------------------------------------------------- */
int cabinet_dwconv_supported(int K, int stride);
int cabinet_dwconv_fwd(const float* x, const float* weight, int B, int C, int H, int W, int K, int stride,
                       float* y, cabinet_stream_t stream);
size_t cabinet_dwconv_bwd_workspace_bytes(int B, int C, int H, int W, int K, int stride);
int cabinet_dwconv_bwd(const float* dy, const float* x, const float* weight, int B, int C, int H, int W, int K,
                       int stride, float* dx, float* dw, void* workspace, size_t workspace_bytes,
                       cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Channel gate + activation: y = act(x * gate[b,c]) for x (B,C,P), gate (B,C); act as in cabinet_bn_act.
 * Replaces the `x * y.view(n, c, 1, 1)` of SELayer.forward, src/models/mobilenetv3.py:79-83, fused with the
 * ReLU / HardSwish that follows it in the MBConv block (mobilenetv3.py:121,141).
 *   bwd: dx = dy * act'(x*gate) * gate,  dgate[b,c] = sum_p dy * act'(x*gate) * x   (x is the only saved tensor)
 * ------------------------------------------------------------------------- */
int cabinet_gate_act_fwd(const float* x, const float* gate, int B, int C, int P, int act, float* y,
                         cabinet_stream_t stream);
size_t cabinet_gate_act_bwd_workspace_bytes(int B, int C, int P);
int cabinet_gate_act_bwd(const float* dy, const float* x, const float* gate, int B, int C, int P, int act,
                         float* dx, float* dgate, void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * BatchNorm2d (+activation) followed by a depthwise convolution, as one operator.
 * Replaces `nn.BatchNorm2d(hidden), act, depthwise nn.Conv2d` of the MBConv block, src/models/mobilenetv3.py:135-143:
 * the normalised, activated (B,C,H,W) tensor is neither written nor re-read -- the convolution normalises while it
 * stages its input tile, and its backward emits the BatchNorm-backward partial sums.
 *   z: the BatchNorm input (B,C,H,W); bn_* / running_* / save_* / act / training / momentum / eps as cabinet_bn_act;
 *   conv_weight (C,1,K,K), K, stride as cabinet_dwconv.   y (B,C,Ho,Wo).
 *   bwd: dy (B,C,Ho,Wo) -> dz (B,C,H,W), dbn_weight, dbn_bias (C), dconv_weight (C,1,K,K).  Deterministic.
 * ------------------------------------------------------------------------- */
size_t cabinet_bn_dwconv_fwd_workspace_bytes(int B, int C, int H, int W);
int cabinet_bn_dwconv_fwd(const float* z, const float* bn_weight, const float* bn_bias,
                          float* running_mean, float* running_var, const float* conv_weight,
                          int B, int C, int H, int W, int K, int stride, int act, int training,
                          float momentum, float eps, float* y, float* save_mean, float* save_invstd,
                          void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
size_t cabinet_bn_dwconv_bwd_workspace_bytes(int B, int C, int H, int W, int K, int stride);
int cabinet_bn_dwconv_bwd(const float* dy, const float* z, const float* bn_weight, const float* bn_bias,
                          const float* save_mean, const float* save_invstd, const float* conv_weight,
                          int B, int C, int H, int W, int K, int stride, int act, int training,
                          float* dz, float* dbn_weight, float* dbn_bias, float* dconv_weight,
                          void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * The 7x7 stride-2 padding-3 stem convolution of the spatial branch: x (B,3,H,W) -> y (B,64,Ho,Wo), no bias.
 * Replaces the nn.Conv2d inside ConvBNReLU(3, 64, kernel_size=7, stride=2, padding=3), src/models/cabinet.py:111
 * (forward cabinet.py:42).  weight (64,3,7,7).  The input is the image: only the weight gradient exists.
 *   wrw : dw (64,3,7,7) = sum over images and pixels of dy (x) patches; ordered slab sum, deterministic.
 * ------------------------------------------------------------------------- */
int cabinet_stem_conv_fwd(const float* x, const float* weight, int B, int H, int W, float* y, cabinet_stream_t stream);
size_t cabinet_stem_conv_wrw_workspace_bytes(int B, int H, int W);
int cabinet_stem_conv_wrw(const float* dy, const float* x, int B, int H, int W, float* dw,
                          void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Thin pointwise (1x1, bias-free, stride 1) convolution on large planes, streaming form:
 * y (B,Co,P) = W (Co,Ci) . x (B,Ci,P).  Replaces the nn.Conv2d(kernel_size=1) of the first MBConv blocks,
 * src/models/mobilenetv3.py:128-131,144-151, where the product is HBM-bound (Ci, Co <= 120).
 * Supported: Ci, Co multiples of 8, <= 120, ceil(Ci/32)*ceil(Co/32) <= 8 (cabinet_pwconv_supported).
 *   bwd: dx = W^T dy (skipped if NULL), dw = sum_{b,p} dy (x) x (skipped if NULL); ordered slab sum, deterministic.
 * ------------------------------------------------------------------------- */
int cabinet_pwconv_supported(int Ci, int Co, int P);
int cabinet_pwconv_fwd(const float* x, const float* w, int B, int Ci, int Co, int P, float* y, cabinet_stream_t stream);
size_t cabinet_pwconv_bwd_workspace_bytes(int B, int Ci, int Co, int P);
int cabinet_pwconv_bwd(const float* dy, const float* x, const float* w, int B, int Ci, int Co, int P,
                       float* dx, float* dw, void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Dense 3x3 convolution, stride 1, padding 1, no bias (nn.Conv2d(Ci, Co, 3, padding=1, bias=False)) as Winograd
 * F(2x2,3x3) on the exact-fp32 MFMA; input transform, the 16 products and the output transform in ONE launch.
 * Replaces src/models/cabinet.py:59 (`conva[0]`), :68 + :88-89 (`b1(torch.cat([x, feat], dim=1))`: the two inputs are read
 * through two pointers, the concat is never materialised) and :160 (`conv_out.conv.conv`), plus their autograd backward.
 *   x0 (B,C0,H,W), x1 (B,C1,H,W) (C1 = 0, x1 = NULL: a plain convolution);  w (Co, C0+C1, 3, 3);  y (B,Co,H,W)
 *   bn_part (fwd, nullable): [2][Co][cabinet_conv3x3_tile_blocks(B,H,W)] floats -- per output channel and tile block
 *     (4 x 32 output pixels) the mean and the sum of squared deviations of the block's valid outputs, for the training-mode
 *     BatchNorm2d that follows (cabinet.py:60, :90): cabinet_bn_act_fwd_stats consumes them instead of a pass over y.
 *   bwd: dx0 (B,C0,H,W) and dx1 (B,C1,H,W) = data gradient (either pair skipped when dx0 == NULL), dw (Co,C0+C1,3,3) =
 *     weight gradient (skipped when NULL); both Winograd too (the weight gradient contracts over tiles: F(3x3,2x2));
 *     ordered slab sums, no atomics: deterministic.
 * Supported (cabinet_conv3x3_supported): C0, C1 multiples of 16, Co and C0+C1 multiples of 64, C0 a multiple of 64 when C1 > 0;
 * any B, H, W (odd sizes masked) with one image's tensors below 1 GiB (zero padding rides on the buffer range check).
 * ------------------------------------------------------------------------- */
int cabinet_conv3x3_supported(int C0, int C1, int Co);
int cabinet_conv3x3_tile_blocks(int B, int H, int W);
size_t cabinet_conv3x3_fwd_workspace_bytes(int B, int C0, int C1, int Co, int H, int W);
int cabinet_conv3x3_fwd(const float* x0, const float* x1, const float* w, int B, int C0, int C1, int Co, int H, int W,
                        float* y, float* bn_part, void* workspace, size_t workspace_bytes, cabinet_stream_t stream);
size_t cabinet_conv3x3_bwd_workspace_bytes(int B, int C0, int C1, int Co, int H, int W);
int cabinet_conv3x3_bwd(const float* dy, const float* x0, const float* x1, const float* w, int B, int C0, int C1, int Co,
                        int H, int W, float* dx0, float* dx1, float* dw,
                        void* workspace, size_t workspace_bytes, cabinet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CABINET_HIP_H_ */
