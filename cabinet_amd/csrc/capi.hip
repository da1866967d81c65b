// C ABI of libcabinet_hip.so: argument checking, workspace carving, dispatch.
// Declarations and the reference lines each entry point replaces: include/cabinet_hip.h
#include "../../include/cabinet_hip.h"

#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "blocks.hpp"
#include "cab_local.hpp"
#include "cab_qkv.hpp"
#include "common.hpp"

namespace cabinet {
// cab_attn_fwd.hip
int attn_fwd_kvsplit(int B, int n);
bool attn_shape_supported(int Kc, int Vc);
hipError_t attn_fwd_dispatch(const float* q, const float* k, const float* v, float scale, int B, int Kc,
                             int Vc, int n, float* ctx, float* lse, float* part_ctx, float* part_lse,
                             int kvsplit, hipStream_t stream);
bool attn_fwd_proj_supported(int B, int Kc, int Vc, int Co, int n);
hipError_t attn_fwd_proj_dispatch(const float* q, const float* k, const float* v, const float* w_out, float scale, int B, int Kc,
                                  int Vc, int Co, int n, float* ctx, float* glob, float* lse, hipStream_t stream);
// cab_attn_bf16.hip
bool attn_bf16_supported(int Kc, int Vc);
size_t attn_bf16_pack_bytes(int B, int Kc, int Vc, int n, int precision);
hipError_t attn_fwd_bf16_dispatch(const float* q, const float* k, const float* v, float scale, int B, int Kc, int Vc, int n,
                                  float* ctx, float* lse, float* part_ctx, float* part_lse, int kvsplit, void* pack, int precision,
                                  hipStream_t stream);
// cab_attn_bwd.hip
size_t attn_bwd_workspace(int B, int Kc, int Vc, int n);
hipError_t attn_bwd_dispatch(const float* dctx, const float* q, const float* k, const float* v,
                             const float* ctx, const float* lse, float scale, int B, int Kc, int Vc, int n,
                             float* dq, float* dk, float* dv, void* ws, hipStream_t stream);
// ffm.hip
struct FfmShape {
    int B, Cs, Cc, Co, Cm, H, W;
};
size_t ffm_fwd_workspace(const FfmShape& s);
size_t ffm_bwd_workspace(const FfmShape& s);
hipError_t ffm_fwd_run(const FfmShape& s, const float* fsp, const float* fcp, const float* w_blk,
                       const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                       const float* w1, const float* w2, int training, float momentum, float eps, float* out,
                       float* z, float* save_mean, float* save_invstd, float* pooled, float* gate, void* ws,
                       hipStream_t stream);
size_t ffm_up_fwd_workspace(const FfmShape& s, int Hl, int Wl);
size_t ffm_up_bwd_workspace(const FfmShape& s, int Hl, int Wl);
hipError_t ffm_up_fwd_run(const FfmShape& s, int Hl, int Wl, const float* fsp, const float* low,
                          const float* w_blk, const float* bn_w, const float* bn_b, float* run_mean,
                          float* run_var, const float* w1, const float* w2, int training, float momentum,
                          float eps, float* out, float* z, float* save_mean, float* save_invstd, float* pooled,
                          float* gate, void* ws, int precision, hipStream_t stream);
hipError_t ffm_up_bwd_run(const FfmShape& s, int Hl, int Wl, const float* dout, const float* fsp,
                          const float* low, const float* w_blk, const float* bn_w, const float* bn_b,
                          const float* w1, const float* w2, const float* z, const float* save_mean,
                          const float* save_invstd, const float* pooled, const float* gate, int training,
                          float* dfsp, float* dlow, float* dw_blk, float* dbn_w, float* dbn_b, float* dw1,
                          float* dw2, void* ws, hipStream_t stream);
hipError_t ffm_bwd_run(const FfmShape& s, const float* dout, const float* fsp, const float* fcp,
                       const float* w_blk, const float* bn_w, const float* bn_b, const float* w1,
                       const float* w2, const float* z, const float* save_mean, const float* save_invstd,
                       const float* pooled, const float* gate, int training, float* dfsp, float* dfcp,
                       float* dw_blk, float* dbn_w, float* dbn_b, float* dw1, float* dw2, void* ws,
                       hipStream_t stream);
// bn_act.hip
size_t bn_act_workspace(int B, int C, int P);
hipError_t bn_act_fwd_run(const float* x, const float* weight, const float* bias, float* running_mean,
                          float* running_var, const float* residual, int B, int C, int P, int act, int training,
                          float momentum, float eps, float* y, float* save_mean, float* save_invstd, void* ws,
                          hipStream_t stream);
hipError_t bn_act_bwd_run(const float* dy, const float* x, const float* weight, const float* bias,
                          const float* save_mean, const float* save_invstd, int B, int C, int P, int act, int training,
                          float* dx, float* dweight, float* dbias, void* ws, hipStream_t stream);
hipError_t bn_act_fwd_part_run(const float* x, const float* conv_part, int H, int W, const float* weight, const float* bias,
                               float* running_mean, float* running_var, const float* residual, int B, int C, int act, int training,
                               float momentum, float eps, float* y, float* save_mean, float* save_invstd, hipStream_t stream);
size_t gate_act_workspace(int rows, int P);
hipError_t gate_act_fwd_run(const float* x, const float* gate, int rows, int P, int act, float* y, hipStream_t stream);
hipError_t gate_act_bwd_run(const float* dy, const float* x, const float* gate, int rows, int P, int act, float* dx,
                            float* dgate, void* ws, hipStream_t stream);
// bn_cls.hip
bool bn_cls_supported(int C, int K, int P);
int bn_cls_table_floats(int C, int K);
size_t bn_cls_fwd_workspace(int B, int C, int P);
size_t bn_cls_bwd_workspace(int B, int C, int K, int P);
hipError_t bn_cls_fwd_run(const float* z, const float* conv_part, const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                          const float* w_cls, const float* bias, int B, int C, int K, int H, int W, int training, float momentum,
                          float eps, float* y, float* tab, void* ws, hipStream_t stream);
hipError_t bn_cls_bwd_run(const float* dy, const float* z, const float* tab, int B, int C, int K, int H, int W, int training, float* dz,
                          float* dbn_w, float* dbn_b, float* dw_cls, float* dbias, void* ws, hipStream_t stream);
// dwconv.hip
bool dwconv_supported(int K, int S);
size_t dwconv_bwd_workspace(int B, int C, int H, int W, int K);
hipError_t dwconv_fwd_run(const float* x, const float* w, int B, int C, int H, int W, int K, int S, float* y,
                          hipStream_t stream);
hipError_t dwconv_bwd_run(const float* dy, const float* x, const float* w, int B, int C, int H, int W, int K, int S,
                          float* dx, float* dw, void* ws, hipStream_t stream);
size_t bn_dwconv_fwd_workspace(int B, int C, int H, int W);
size_t bn_dwconv_bwd_workspace(int B, int C, int H, int W, int K);
hipError_t bn_dwconv_fwd_run(const float* z, const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                             const float* w, int B, int C, int H, int W, int K, int S, int act, int training,
                             float momentum, float eps, float* y, float* save_mean, float* save_invstd, void* ws,
                             hipStream_t stream);
hipError_t bn_dwconv_bwd_run(const float* dy, const float* z, const float* bn_w, const float* bn_b,
                             const float* save_mean, const float* save_invstd, const float* w, int B, int C, int H,
                             int W, int K, int S, int act, int training, float* dz, float* dbn_w, float* dbn_b,
                             float* dw, void* ws, hipStream_t stream);
// stem_conv.hip
size_t stem_conv_wrw_workspace(int B, int H, int W);
hipError_t stem_conv_fwd_run(const float* x, const float* w, int B, int H, int W, float* y, hipStream_t stream);
hipError_t stem_conv_wrw_run(const float* dy, const float* x, int B, int H, int W, float* dw, void* ws,
                             hipStream_t stream);
// pwconv.hip
bool pwconv_supported(int Ci, int Co, int P);
size_t pwconv_bwd_workspace(int B, int Ci, int Co, int P);
hipError_t pwconv_fwd_run(const float* x, const float* w, int B, int Ci, int Co, int P, float* y, hipStream_t stream);
hipError_t pwconv_bwd_run(const float* dy, const float* x, const float* w, int B, int Ci, int Co, int P, float* dx,
                          float* dw, void* ws, hipStream_t stream);
// ohem.hip
int ohem_blocks(int B, int H, int W);
hipError_t ohem_up_fwd_run(int nh, const float* const* low, const long long* labels, int B, int C, int Hl, int Wl, int H, int W,
                           float thresh, int ignore_lb, float* const* loss_px, float* const* blk_sum, int* const* blk_cnt,
                           hipStream_t stream);
hipError_t ohem_stats_run(const float* blk_sum, const int* blk_cnt, int nheads, int nblk, double* stats, hipStream_t stream);
size_t ohem_up_bwd_workspace(int B, int C, int H, int Wl);
bool ohem_up_supported(int C, int Wl, int W);
hipError_t ohem_up_bwd_run(int nh, const float* const* low, const long long* labels, const float* const* loss_px, int B, int C,
                           int Hl, int Wl, int H, int W, float thresh, int ignore_lb, float coef, float* dlow, void* ws,
                           hipStream_t stream);
// conv3x3_wino.hip
struct WinoShape {
    int B, C0, C1, K, H, W;
};
bool conv3x3_supported(int C0, int C1, int K);
bool conv3x3_shape_ok(const WinoShape& s);
int conv3x3_tile_blocks(int B, int H, int W);
size_t conv3x3_fwd_workspace(const WinoShape& s);
size_t conv3x3_bwd_workspace(const WinoShape& s);
hipError_t conv3x3_fwd_run(const WinoShape& s, const float* x0, const float* x1, const float* w, float* y, float* stat_part, void* ws,
                           hipStream_t stream);
hipError_t conv3x3_bwd_run(const WinoShape& s, const float* dy, const float* x0, const float* x1, const float* w, float* dx0,
                           float* dx1, float* dw, void* ws, hipStream_t stream);
}  // namespace cabinet

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// 128-bit loads / stores on activation rows: every tensor pointer named must be 16-byte aligned (include/cabinet_hip.h)
template <typename... P>
static bool aligned16(P... p) {
    return ((reinterpret_cast<uintptr_t>(p) | ...) & 15) == 0;
}
#define CABINET_REQUIRE_ALIGNED(what, ...)                                                                            \
    if (!aligned16(__VA_ARGS__))                                                                                      \
    return fail(CABINET_ERR_INVALID_ARG, what ": tensor pointers must be 16-byte aligned (128-bit loads / stores)")

static int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return CABINET_OK;
    return fail(CABINET_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

extern "C" {

int cabinet_abi_version(void) { return CABINET_ABI_VERSION; }
const char* cabinet_last_error(void) { return g_err; }

// ------------------------------------------------------------------ attention
static int check_attn_shape(int B, int Kc, int Vc, int n) {
    if (B <= 0 || Kc <= 0 || Vc <= 0 || n <= 0)
        return fail(CABINET_ERR_INVALID_ARG, "cab_attn: non-positive dimension B=%d Kc=%d Vc=%d n=%d", B, Kc,
                    Vc, n);
    if (!cabinet::attn_shape_supported(Kc, Vc))
        return fail(CABINET_ERR_UNSUPPORTED,
                    "cab_attn: (Kc=%d, Vc=%d) has no gfx950 instantiation; built: (128,128) (256,128) (64,64)",
                    Kc, Vc);
    if (B > 65535) return fail(CABINET_ERR_UNSUPPORTED, "cab_attn: B=%d exceeds grid.z", B);
    return CABINET_OK;
}

int cabinet_cab_attn_supported(int Kc, int Vc) { return cabinet::attn_shape_supported(Kc, Vc) ? 1 : 0; }

int cabinet_cab_attn_precision_supported(int Kc, int Vc, int precision) {
    if (precision == CABINET_PREC_FP32) return cabinet_cab_attn_supported(Kc, Vc);
    if (precision == CABINET_PREC_BF16X3 || precision == CABINET_PREC_BF16X6) return cabinet::attn_bf16_supported(Kc, Vc) ? 1 : 0;
    return 0;
}

static size_t attn_fwd_split_bytes(int B, int Vc, int n, int split) {
    if (split == 1) return 0;
    return align_up((size_t)split * B * Vc * n * sizeof(float), 256) + align_up((size_t)split * B * n * sizeof(float), 256);
}

size_t cabinet_cab_attn_fwd_workspace_bytes(int B, int Kc, int Vc, int n, int precision) {
    if (B <= 0 || Kc <= 0 || Vc <= 0 || n <= 0 || !cabinet_cab_attn_precision_supported(Kc, Vc, precision)) return 0;
    const size_t pack = precision == CABINET_PREC_FP32 ? 0 : cabinet::attn_bf16_pack_bytes(B, Kc, Vc, n, precision);
    return attn_fwd_split_bytes(B, Vc, n, cabinet::attn_fwd_kvsplit(B, n)) + pack;
}

int cabinet_cab_attn_fwd(const float* q, const float* k, const float* v, float scale, int B, int Kc, int Vc,
                         int n, int precision, float* ctx, float* lse, void* workspace, size_t workspace_bytes,
                         cabinet_stream_t stream) {
    if (int rc = check_attn_shape(B, Kc, Vc, n)) return rc;
    if (!cabinet_cab_attn_precision_supported(Kc, Vc, precision))
        return fail(CABINET_ERR_UNSUPPORTED, "cab_attn_fwd: precision %d is not built for (Kc=%d, Vc=%d)", precision, Kc, Vc);
    if (!q || !k || !v || !ctx || !lse) return fail(CABINET_ERR_INVALID_ARG, "cab_attn_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("cab_attn_fwd", q, k, v, ctx, lse);
    const size_t need = cabinet_cab_attn_fwd_workspace_bytes(B, Kc, Vc, n, precision);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "cab_attn_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    const int split = cabinet::attn_fwd_kvsplit(B, n);
    float* part_ctx = static_cast<float*>(workspace);
    float* part_lse = split > 1 ? reinterpret_cast<float*>(static_cast<char*>(workspace) +
                                                           align_up((size_t)split * B * Vc * n * sizeof(float), 256))
                                : nullptr;
    if (precision != CABINET_PREC_FP32) {
        void* pack = static_cast<char*>(workspace) + attn_fwd_split_bytes(B, Vc, n, split);
        return hip_status(cabinet::attn_fwd_bf16_dispatch(q, k, v, scale, B, Kc, Vc, n, ctx, lse, part_ctx, part_lse, split, pack,
                                                          precision, static_cast<hipStream_t>(stream)),
                          "cab_attn_fwd (split bf16) launch");
    }
    return hip_status(cabinet::attn_fwd_dispatch(q, k, v, scale, B, Kc, Vc, n, ctx, lse, part_ctx, part_lse,
                                                 split, static_cast<hipStream_t>(stream)),
                      "cab_attn_fwd launch");
}

int cabinet_cab_attn_proj_supported(int B, int Kc, int Vc, int Co, int n) {
    return cabinet::attn_fwd_proj_supported(B, Kc, Vc, Co, n) ? 1 : 0;
}

int cabinet_cab_attn_proj_fwd(const float* q, const float* k, const float* v, const float* w_out, float scale, int B, int Kc,
                              int Vc, int Co, int n, float* ctx, float* glob, float* lse, cabinet_stream_t stream) {
    if (int rc = check_attn_shape(B, Kc, Vc, n)) return rc;
    if (Co <= 0) return fail(CABINET_ERR_INVALID_ARG, "cab_attn_proj_fwd: non-positive Co=%d", Co);
    if (!cabinet::attn_fwd_proj_supported(B, Kc, Vc, Co, n))
        return fail(CABINET_ERR_UNSUPPORTED,
                    "cab_attn_proj_fwd: (B=%d, Kc=%d, Vc=%d, Co=%d, n=%d) is outside the fused form (see "
                    "cabinet_cab_attn_proj_supported); use cabinet_cab_attn_fwd + cabinet_conv1x1_fwd",
                    B, Kc, Vc, Co, n);
    if (!q || !k || !v || !w_out || !glob || !lse) return fail(CABINET_ERR_INVALID_ARG, "cab_attn_proj_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("cab_attn_proj_fwd", q, k, v, w_out, ctx, glob, lse);
    return hip_status(cabinet::attn_fwd_proj_dispatch(q, k, v, w_out, scale, B, Kc, Vc, Co, n, ctx, glob, lse,
                                                      static_cast<hipStream_t>(stream)),
                      "cab_attn_proj_fwd launch");
}

size_t cabinet_cab_attn_bwd_workspace_bytes(int B, int Kc, int Vc, int n) {
    if (B <= 0 || Kc <= 0 || Vc <= 0 || n <= 0) return 0;
    return cabinet::attn_bwd_workspace(B, Kc, Vc, n);
}

int cabinet_cab_attn_bwd(const float* dctx, const float* q, const float* k, const float* v, const float* ctx,
                         const float* lse, float scale, int B, int Kc, int Vc, int n, float* dq, float* dk,
                         float* dv, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_attn_shape(B, Kc, Vc, n)) return rc;
    if (!dctx || !q || !k || !v || !ctx || !lse || !dq || !dk || !dv)
        return fail(CABINET_ERR_INVALID_ARG, "cab_attn_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("cab_attn_bwd", dctx, q, k, v, ctx, lse, dq, dk, dv);
    const size_t need = cabinet_cab_attn_bwd_workspace_bytes(B, Kc, Vc, n);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "cab_attn_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::attn_bwd_dispatch(dctx, q, k, v, ctx, lse, scale, B, Kc, Vc, n, dq, dk, dv,
                                                 workspace, static_cast<hipStream_t>(stream)),
                      "cab_attn_bwd launch");
}

// ------------------------------------------------------------------------ FFM
static int check_ffm_shape(int B, int Cs, int Cc, int Co, int Cm, int H, int W) {
    if (B <= 0 || Cs <= 0 || Cc <= 0 || Co <= 0 || Cm <= 0 || H <= 0 || W <= 0)
        return fail(CABINET_ERR_INVALID_ARG, "ffm: non-positive dimension");
    if (Cs % 32 || Cc % 32 || Co % 32)
        return fail(CABINET_ERR_UNSUPPORTED, "ffm: Cs=%d Cc=%d Co=%d must be multiples of 32", Cs, Cc, Co);
    if (Cm > 256 || Co > 1024)
        return fail(CABINET_ERR_UNSUPPORTED, "ffm: Cm=%d (max 256) / Co=%d (max 1024) too large", Cm, Co);
    if (B > 65535) return fail(CABINET_ERR_UNSUPPORTED, "ffm: B=%d exceeds grid limits", B);
    return CABINET_OK;
}

size_t cabinet_ffm_fwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W) {
    if (B <= 0 || Cs <= 0 || Cc <= 0 || Co <= 0 || Cm <= 0 || H <= 0 || W <= 0) return 0;
    return cabinet::ffm_fwd_workspace({B, Cs, Cc, Co, Cm, H, W});
}

int cabinet_ffm_fwd(const float* fsp, const float* fcp, const float* w_blk, const float* bn_weight,
                    const float* bn_bias, float* running_mean, float* running_var, const float* w1,
                    const float* w2, int B, int Cs, int Cc, int Co, int Cm, int H, int W, int training,
                    float momentum, float eps, float* out, float* z, float* save_mean, float* save_invstd,
                    float* pooled, float* gate, void* workspace, size_t workspace_bytes,
                    cabinet_stream_t stream) {
    if (int rc = check_ffm_shape(B, Cs, Cc, Co, Cm, H, W)) return rc;
    if (!fsp || !fcp || !w_blk || !bn_weight || !bn_bias || !running_mean || !running_var || !w1 || !w2 ||
        !out || !z || !save_mean || !save_invstd || !pooled || !gate)
        return fail(CABINET_ERR_INVALID_ARG, "ffm_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ffm_fwd", fsp, fcp, w_blk, out, z);
    const size_t need = cabinet_ffm_fwd_workspace_bytes(B, Cs, Cc, Co, Cm, H, W);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "ffm_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::ffm_fwd_run({B, Cs, Cc, Co, Cm, H, W}, fsp, fcp, w_blk, bn_weight, bn_bias,
                                           running_mean, running_var, w1, w2, training, momentum, eps, out, z,
                                           save_mean, save_invstd, pooled, gate, workspace,
                                           static_cast<hipStream_t>(stream)),
                      "ffm_fwd launch");
}

size_t cabinet_ffm_bwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W) {
    if (B <= 0 || Cs <= 0 || Cc <= 0 || Co <= 0 || Cm <= 0 || H <= 0 || W <= 0) return 0;
    return cabinet::ffm_bwd_workspace({B, Cs, Cc, Co, Cm, H, W});
}

int cabinet_ffm_bwd(const float* dout, const float* fsp, const float* fcp, const float* w_blk,
                    const float* bn_weight, const float* bn_bias, const float* w1, const float* w2,
                    const float* z, const float* save_mean, const float* save_invstd, const float* pooled,
                    const float* gate, int B, int Cs, int Cc, int Co, int Cm, int H, int W, int training,
                    float* dfsp, float* dfcp, float* dw_blk, float* dbn_weight, float* dbn_bias, float* dw1,
                    float* dw2, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_ffm_shape(B, Cs, Cc, Co, Cm, H, W)) return rc;
    if (!dout || !fsp || !fcp || !w_blk || !bn_weight || !bn_bias || !w1 || !w2 || !z || !save_mean ||
        !save_invstd || !pooled || !gate || !dfsp || !dfcp || !dw_blk || !dbn_weight || !dbn_bias || !dw1 ||
        !dw2)
        return fail(CABINET_ERR_INVALID_ARG, "ffm_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ffm_bwd", dout, fsp, fcp, w_blk, z, dfsp, dfcp, dw_blk);
    const size_t need = cabinet_ffm_bwd_workspace_bytes(B, Cs, Cc, Co, Cm, H, W);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "ffm_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::ffm_bwd_run({B, Cs, Cc, Co, Cm, H, W}, dout, fsp, fcp, w_blk, bn_weight, bn_bias,
                                           w1, w2, z, save_mean, save_invstd, pooled, gate, training, dfsp,
                                           dfcp, dw_blk, dbn_weight, dbn_bias, dw1, dw2, workspace,
                                           static_cast<hipStream_t>(stream)),
                      "ffm_bwd launch");
}

// ------------------------------------------------------------- FFM + fused upsample
static int check_low(int Hl, int Wl) {
    if (Hl <= 0 || Wl <= 0) return fail(CABINET_ERR_INVALID_ARG, "ffm_up: non-positive low-resolution size");
    return CABINET_OK;
}

size_t cabinet_ffm_up_fwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl) {
    if (B <= 0 || Cs <= 0 || Cc <= 0 || Co <= 0 || Cm <= 0 || H <= 0 || W <= 0 || Hl <= 0 || Wl <= 0) return 0;
    return cabinet::ffm_up_fwd_workspace({B, Cs, Cc, Co, Cm, H, W}, Hl, Wl);
}

int cabinet_ffm_up_fwd(const float* fsp, const float* low, const float* w_blk, const float* bn_weight,
                       const float* bn_bias, float* running_mean, float* running_var, const float* w1,
                       const float* w2, int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl,
                       int training, float momentum, float eps, int precision, float* out, float* z, float* save_mean,
                       float* save_invstd, float* pooled, float* gate, void* workspace, size_t workspace_bytes,
                       cabinet_stream_t stream) {
    if (int rc = check_ffm_shape(B, Cs, Cc, Co, Cm, H, W)) return rc;
    if (int rc = check_low(Hl, Wl)) return rc;
    if (precision < CABINET_PREC_FP32 || precision > CABINET_PREC_BF16X6)
        return fail(CABINET_ERR_INVALID_ARG, "ffm_up_fwd: unknown precision %d", precision);
    if (!fsp || !low || !w_blk || !bn_weight || !bn_bias || !running_mean || !running_var || !w1 || !w2 || !out ||
        !z || !save_mean || !save_invstd || !pooled || !gate)
        return fail(CABINET_ERR_INVALID_ARG, "ffm_up_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ffm_up_fwd", fsp, low, w_blk, out, z);
    const size_t need = cabinet_ffm_up_fwd_workspace_bytes(B, Cs, Cc, Co, Cm, H, W, Hl, Wl);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "ffm_up_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::ffm_up_fwd_run({B, Cs, Cc, Co, Cm, H, W}, Hl, Wl, fsp, low, w_blk, bn_weight, bn_bias,
                                              running_mean, running_var, w1, w2, training, momentum, eps, out, z,
                                              save_mean, save_invstd, pooled, gate, workspace, precision,
                                              static_cast<hipStream_t>(stream)),
                      "ffm_up_fwd launch");
}

size_t cabinet_ffm_up_bwd_workspace_bytes(int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl) {
    if (B <= 0 || Cs <= 0 || Cc <= 0 || Co <= 0 || Cm <= 0 || H <= 0 || W <= 0 || Hl <= 0 || Wl <= 0) return 0;
    return cabinet::ffm_up_bwd_workspace({B, Cs, Cc, Co, Cm, H, W}, Hl, Wl);
}

int cabinet_ffm_up_bwd(const float* dout, const float* fsp, const float* low, const float* w_blk,
                       const float* bn_weight, const float* bn_bias, const float* w1, const float* w2,
                       const float* z, const float* save_mean, const float* save_invstd, const float* pooled,
                       const float* gate, int B, int Cs, int Cc, int Co, int Cm, int H, int W, int Hl, int Wl,
                       int training, float* dfsp, float* dlow, float* dw_blk, float* dbn_weight, float* dbn_bias,
                       float* dw1, float* dw2, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_ffm_shape(B, Cs, Cc, Co, Cm, H, W)) return rc;
    if (int rc = check_low(Hl, Wl)) return rc;
    if (!dout || !fsp || !low || !w_blk || !bn_weight || !bn_bias || !w1 || !w2 || !z || !save_mean ||
        !save_invstd || !pooled || !gate || !dfsp || !dlow || !dw_blk || !dbn_weight || !dbn_bias || !dw1 || !dw2)
        return fail(CABINET_ERR_INVALID_ARG, "ffm_up_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ffm_up_bwd", dout, fsp, low, w_blk, z, dfsp, dlow, dw_blk);
    const size_t need = cabinet_ffm_up_bwd_workspace_bytes(B, Cs, Cc, Co, Cm, H, W, Hl, Wl);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "ffm_up_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::ffm_up_bwd_run({B, Cs, Cc, Co, Cm, H, W}, Hl, Wl, dout, fsp, low, w_blk, bn_weight,
                                              bn_bias, w1, w2, z, save_mean, save_invstd, pooled, gate, training,
                                              dfsp, dlow, dw_blk, dbn_weight, dbn_bias, dw1, dw2, workspace,
                                              static_cast<hipStream_t>(stream)),
                      "ffm_up_bwd launch");
}

// ------------------------------------------------------ OHEM-CE + fused upsample
static int check_ohem(int B, int C, int Hl, int Wl, int H, int W) {
    if (B <= 0 || C <= 0 || Hl <= 0 || Wl <= 0 || H <= 0 || W <= 0)
        return fail(CABINET_ERR_INVALID_ARG, "ohem_up: non-positive dimension");
    if (C > 32) return fail(CABINET_ERR_UNSUPPORTED, "ohem_up: C=%d classes (max 32)", C);
    if (B > 65535 || H > 65535) return fail(CABINET_ERR_UNSUPPORTED, "ohem_up: B or H exceeds grid limits");
    if (!cabinet::ohem_up_supported(C, Wl, W))
        return fail(CABINET_ERR_UNSUPPORTED, "ohem_up: C=%d x Wl=%d (or the resize ratio %d/%d) exceeds the LDS row buffers",
                    C, Wl, W, Wl);
    return CABINET_OK;
}

int cabinet_ohem_up_blocks(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return cabinet::ohem_blocks(B, H, W);
}

int cabinet_ohem_up_fwd(const float* logits_low, const long long* labels, int B, int C, int Hl, int Wl, int H, int W,
                        float thresh, int ignore_lb, float* loss_px, float* blk_sum, int* blk_cnt,
                        cabinet_stream_t stream) {
    if (int rc = check_ohem(B, C, Hl, Wl, H, W)) return rc;
    if (!logits_low || !labels || !loss_px || !blk_sum || !blk_cnt)
        return fail(CABINET_ERR_INVALID_ARG, "ohem_up_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ohem_up_fwd", logits_low, labels, loss_px);
    return hip_status(cabinet::ohem_up_fwd_run(1, &logits_low, labels, B, C, Hl, Wl, H, W, thresh, ignore_lb, &loss_px, &blk_sum,
                                               &blk_cnt, static_cast<hipStream_t>(stream)),
                      "ohem_up_fwd launch");
}

int cabinet_ohem_stats(const float* blk_sum, const int* blk_cnt, int nheads, int nblk, double* stats, cabinet_stream_t stream) {
    if (nheads <= 0 || nheads > 65535 || nblk <= 0) return fail(CABINET_ERR_INVALID_ARG, "ohem_stats: nheads=%d nblk=%d", nheads, nblk);
    if (!blk_sum || !blk_cnt || !stats) return fail(CABINET_ERR_INVALID_ARG, "ohem_stats: null tensor pointer");
    return hip_status(cabinet::ohem_stats_run(blk_sum, blk_cnt, nheads, nblk, stats, static_cast<hipStream_t>(stream)),
                      "ohem_stats launch");
}

size_t cabinet_ohem_up_bwd_workspace_bytes(int B, int C, int Hl, int Wl, int H, int W) {
    if (B <= 0 || C <= 0 || Hl <= 0 || Wl <= 0 || H <= 0 || W <= 0) return 0;
    return cabinet::ohem_up_bwd_workspace(B, C, H, Wl);
}

int cabinet_ohem_up_bwd(const float* logits_low, const long long* labels, const float* loss_px, int B, int C, int Hl,
                        int Wl, int H, int W, float thresh, int ignore_lb, float coef, float* dlogits_low,
                        void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_ohem(B, C, Hl, Wl, H, W)) return rc;
    if (!logits_low || !labels || !loss_px || !dlogits_low)
        return fail(CABINET_ERR_INVALID_ARG, "ohem_up_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ohem_up_bwd", logits_low, labels, loss_px, dlogits_low);
    const size_t need = cabinet_ohem_up_bwd_workspace_bytes(B, C, Hl, Wl, H, W);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "ohem_up_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::ohem_up_bwd_run(1, &logits_low, labels, &loss_px, B, C, Hl, Wl, H, W, thresh, ignore_lb, coef,
                                               dlogits_low, workspace, static_cast<hipStream_t>(stream)),
                      "ohem_up_bwd launch");
}

// both loss heads of the step (same labels, same shapes) per launch
int cabinet_ohem_up_pair_fwd(const float* logits_low_a, const float* logits_low_b, const long long* labels, int B, int C, int Hl,
                             int Wl, int H, int W, float thresh, int ignore_lb, float* loss_px, float* blk_sum, int* blk_cnt,
                             cabinet_stream_t stream) {
    if (int rc = check_ohem(B, C, Hl, Wl, H, W)) return rc;
    if (!logits_low_a || !logits_low_b || !labels || !loss_px || !blk_sum || !blk_cnt)
        return fail(CABINET_ERR_INVALID_ARG, "ohem_up_pair_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ohem_up_pair_fwd", logits_low_a, logits_low_b, labels, loss_px);
    const int nblk = cabinet::ohem_blocks(B, H, W);
    const float* low[2] = {logits_low_a, logits_low_b};
    float* lp[2] = {loss_px, loss_px + (size_t)B * H * W};
    float* bs[2] = {blk_sum, blk_sum + nblk};
    int* bc[2] = {blk_cnt, blk_cnt + 2 * nblk};
    return hip_status(cabinet::ohem_up_fwd_run(2, low, labels, B, C, Hl, Wl, H, W, thresh, ignore_lb, lp, bs, bc,
                                               static_cast<hipStream_t>(stream)),
                      "ohem_up_pair_fwd launch");
}

size_t cabinet_ohem_up_pair_bwd_workspace_bytes(int B, int C, int Hl, int Wl, int H, int W) {
    return 2 * cabinet_ohem_up_bwd_workspace_bytes(B, C, Hl, Wl, H, W);
}

int cabinet_ohem_up_pair_bwd(const float* logits_low_a, const float* logits_low_b, const long long* labels, const float* loss_px,
                             int B, int C, int Hl, int Wl, int H, int W, float thresh, int ignore_lb, float coef,
                             float* dlogits_low, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_ohem(B, C, Hl, Wl, H, W)) return rc;
    if (!logits_low_a || !logits_low_b || !labels || !loss_px || !dlogits_low)
        return fail(CABINET_ERR_INVALID_ARG, "ohem_up_pair_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("ohem_up_pair_bwd", logits_low_a, logits_low_b, labels, loss_px, dlogits_low);
    const size_t need = cabinet_ohem_up_pair_bwd_workspace_bytes(B, C, Hl, Wl, H, W);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "ohem_up_pair_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    const float* low[2] = {logits_low_a, logits_low_b};
    const float* lp[2] = {loss_px, loss_px + (size_t)B * H * W};
    return hip_status(cabinet::ohem_up_bwd_run(2, low, labels, lp, B, C, Hl, Wl, H, W, thresh, ignore_lb, coef, dlogits_low,
                                               workspace, static_cast<hipStream_t>(stream)),
                      "ohem_up_pair_bwd launch");
}

// ------------------------------------------------------ CAB local branch + block output
// two forms behind one entry point: the channel-resident kernel (B*H*W <= 8192, no workspace) and the tiled
// multi-workgroup-per-channel form for everything larger (cab_local_tiled.hip)
static int check_local(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(CABINET_ERR_INVALID_ARG, "cab_local: non-positive dimension");
    if (C > 65535) return fail(CABINET_ERR_UNSUPPORTED, "cab_local: C = %d exceeds the grid's 65535 channels", C);
    if (!cabinet::local_shape_supported(B, H, W) && !cabinet::local_tiled_supported(B, H, W))
        return fail(CABINET_ERR_UNSUPPORTED, "cab_local: %d x %d x %d planes fit neither the resident nor the tiled form", B, H, W);
    return CABINET_OK;
}

int cabinet_cab_local_supported(int B, int C, int H, int W) {
    return B > 0 && C > 0 && C <= 65535 && H > 0 && W > 0 &&
                   (cabinet::local_shape_supported(B, H, W) || cabinet::local_tiled_supported(B, H, W))
               ? 1
               : 0;
}

size_t cabinet_cab_local_fwd_workspace_bytes(int B, int C, int H, int W) {
    if (!cabinet_cab_local_supported(B, C, H, W) || cabinet::local_shape_supported(B, H, W)) return 0;
    return cabinet::local_tiled_fwd_workspace(B, C, H, W);
}

size_t cabinet_cab_local_bwd_workspace_bytes(int B, int C, int H, int W) {
    if (!cabinet_cab_local_supported(B, C, H, W) || cabinet::local_shape_supported(B, H, W)) return 0;
    return cabinet::local_tiled_bwd_workspace(B, C, H, W);
}

int cabinet_cab_local_fwd(const float* x, const float* glob, const float* gamma, const float* const* dw_w,
                          const float* const* bn_weight, const float* const* bn_bias, float* const* running_mean,
                          float* const* running_var, int B, int C, int H, int W, int training, float momentum,
                          float eps, float* out, float* save_mean, float* save_invstd, void* workspace,
                          size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_local(B, C, H, W)) return rc;
    if (!x || !out || !save_mean || !save_invstd || !dw_w || !bn_weight || !bn_bias || !running_mean || !running_var)
        return fail(CABINET_ERR_INVALID_ARG, "cab_local_fwd: null tensor pointer");
    if (glob && !gamma) return fail(CABINET_ERR_INVALID_ARG, "cab_local_fwd: glob given without gamma");
    const size_t need = cabinet_cab_local_fwd_workspace_bytes(B, C, H, W);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "cab_local_fwd: workspace %zu < %zu", workspace_bytes, need);
    cabinet::LocalArgs a{};
    a.x = x, a.glob = glob, a.gamma = gamma;
    for (int s = 0; s < 3; ++s) {
        if (!dw_w[s] || !bn_weight[s] || !bn_bias[s] || !running_mean[s] || !running_var[s])
            return fail(CABINET_ERR_INVALID_ARG, "cab_local_fwd: null parameter pointer (stage %d)", s);
        a.st[s].w = dw_w[s], a.st[s].bn_w = bn_weight[s], a.st[s].bn_b = bn_bias[s];
        a.st[s].run_mean = running_mean[s], a.st[s].run_var = running_var[s];
    }
    a.B = B, a.C = C, a.H = H, a.W = W, a.training = training, a.momentum = momentum, a.eps = eps;
    a.out = out, a.save_mean = save_mean, a.save_invstd = save_invstd;
    if (need) return hip_status(cabinet::cab_local_tiled_fwd_run(a, workspace, static_cast<hipStream_t>(stream)), "cab_local_fwd launch");
    return hip_status(cabinet::cab_local_fwd_run(a, static_cast<hipStream_t>(stream)), "cab_local_fwd launch");
}

int cabinet_cab_local_bwd(const float* dout, const float* x, const float* glob, const float* gamma,
                          const float* const* dw_w, const float* const* bn_weight, const float* const* bn_bias,
                          const float* save_mean, const float* save_invstd, int B, int C, int H, int W, int training,
                          float* dx, float* dglob, float* dgamma_part, float* const* ddw_w, float* const* dbn_weight,
                          float* const* dbn_bias, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_local(B, C, H, W)) return rc;
    if (!dout || !x || !dx || !save_mean || !save_invstd || !dw_w || !bn_weight || !bn_bias || !ddw_w || !dbn_weight ||
        !dbn_bias)
        return fail(CABINET_ERR_INVALID_ARG, "cab_local_bwd: null tensor pointer");
    if (glob && (!gamma || !dglob || !dgamma_part))
        return fail(CABINET_ERR_INVALID_ARG, "cab_local_bwd: glob given without gamma / dglob / dgamma_part");
    const size_t need = cabinet_cab_local_bwd_workspace_bytes(B, C, H, W);
    if (need && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "cab_local_bwd: workspace %zu < %zu", workspace_bytes, need);
    cabinet::LocalArgs a{};
    a.x = x, a.glob = glob, a.gamma = gamma, a.dout = dout;
    for (int s = 0; s < 3; ++s) {
        if (!dw_w[s] || !bn_weight[s] || !bn_bias[s] || !ddw_w[s] || !dbn_weight[s] || !dbn_bias[s])
            return fail(CABINET_ERR_INVALID_ARG, "cab_local_bwd: null parameter pointer (stage %d)", s);
        a.st[s].w = dw_w[s], a.st[s].bn_w = bn_weight[s], a.st[s].bn_b = bn_bias[s];
        a.st[s].dw = ddw_w[s], a.st[s].dbn_w = dbn_weight[s], a.st[s].dbn_b = dbn_bias[s];
    }
    a.B = B, a.C = C, a.H = H, a.W = W, a.training = training;
    a.save_mean = const_cast<float*>(save_mean), a.save_invstd = const_cast<float*>(save_invstd);
    a.dx = dx, a.dglob = dglob, a.dgamma_part = dgamma_part;
    if (need) return hip_status(cabinet::cab_local_tiled_bwd_run(a, workspace, static_cast<hipStream_t>(stream)), "cab_local_bwd launch");
    return hip_status(cabinet::cab_local_bwd_run(a, static_cast<hipStream_t>(stream)), "cab_local_bwd launch");
}

// ------------------------------------------------------ q/k/v producers
static int qkv_shape(int B, int C, int Kc, int Vc, int H, int W, int ns, const int* sizes, cabinet::QkvShape& s,
                     const char* who) {
    if (B <= 0 || C <= 0 || Kc <= 0 || Vc <= 0 || H <= 0 || W <= 0 || !sizes)
        return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension or null sizes", who);
    if (ns < 1 || ns > 4) return fail(CABINET_ERR_UNSUPPORTED, "%s: n_sizes=%d (1..4)", who, ns);
    s = cabinet::QkvShape{B, C, Kc, Vc, H, W, ns, {0, 0, 0, 0}};
    for (int i = 0; i < ns; ++i) s.sizes[i] = sizes[i];
    if (const char* need = cabinet::qkv_unsupported(s)) return fail(CABINET_ERR_UNSUPPORTED, "%s: needs %s", who, need);
    if ((long long)B * (2 * Kc + Vc) > 2147483647LL / 2) return fail(CABINET_ERR_UNSUPPORTED, "%s: grid too large", who);
    return CABINET_OK;
}

int cabinet_cab_qkv_supported(int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes) {
    cabinet::QkvShape s;
    return qkv_shape(B, C, Kc, Vc, H, W, n_sizes, sizes, s, "cab_qkv") == CABINET_OK ? 1 : 0;
}

int cabinet_cab_qkv_padded_bins(int n_sizes, const int* sizes) {
    if (n_sizes < 1 || n_sizes > 4 || !sizes) return 0;
    cabinet::QkvShape s{1, 16, 16, 16, 1, 1, n_sizes, {0, 0, 0, 0}};
    for (int i = 0; i < n_sizes; ++i) s.sizes[i] = sizes[i];
    return cabinet::qkv_padded_bins(s);
}

size_t cabinet_cab_qkv_fwd_workspace_bytes(int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes) {
    cabinet::QkvShape s;
    if (qkv_shape(B, C, Kc, Vc, H, W, n_sizes, sizes, s, "cab_qkv_fwd")) return 0;
    return cabinet::qkv_fwd_workspace(s);
}

int cabinet_cab_qkv_fwd(const float* x, const float* wq, const float* wk, const float* wv, const float* bnq_weight,
                        const float* bnq_bias, float* bnq_running_mean, float* bnq_running_var,
                        const float* bnk_weight, const float* bnk_bias, float* bnk_running_mean,
                        float* bnk_running_var, const float* wpk, const float* wpv, int B, int C, int Kc, int Vc, int H,
                        int W, int n_sizes, const int* sizes, int training, float momentum, float eps, float* q,
                        float* k, float* v, float* zqk, float* vv, float* kk, float* pooled_k, float* pooled_v,
                        float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                        cabinet_stream_t stream) {
    cabinet::QkvShape s;
    if (int rc = qkv_shape(B, C, Kc, Vc, H, W, n_sizes, sizes, s, "cab_qkv_fwd")) return rc;
    if (!x || !wq || !wk || !wv || !bnq_weight || !bnq_bias || !bnq_running_mean || !bnq_running_var || !bnk_weight ||
        !bnk_bias || !bnk_running_mean || !bnk_running_var || !wpk || !wpv || !q || !k || !v || !zqk || !vv || !kk ||
        !pooled_k || !pooled_v || !save_mean || !save_invstd)
        return fail(CABINET_ERR_INVALID_ARG, "cab_qkv_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("cab_qkv_fwd", x, q, k, v, zqk, vv, kk);
    const size_t need = cabinet::qkv_fwd_workspace(s);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "cab_qkv_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    const cabinet::QkvParams w{wq, wk, wv, bnq_weight, bnq_bias, bnk_weight, bnk_bias, bnq_running_mean,
                               bnq_running_var, bnk_running_mean, bnk_running_var, wpk, wpv};
    const cabinet::QkvSaved sv{zqk, vv, kk, pooled_k, pooled_v, save_mean, save_invstd};
    return hip_status(cabinet::qkv_fwd_run(s, w, x, training, momentum, eps, sv, q, k, v, workspace,
                                           static_cast<hipStream_t>(stream)),
                      "cab_qkv_fwd launch");
}

size_t cabinet_cab_qkv_bwd_workspace_bytes(int B, int C, int Kc, int Vc, int H, int W, int n_sizes, const int* sizes) {
    cabinet::QkvShape s;
    if (qkv_shape(B, C, Kc, Vc, H, W, n_sizes, sizes, s, "cab_qkv_bwd")) return 0;
    return cabinet::qkv_bwd_workspace(s);
}

int cabinet_cab_qkv_bwd(const float* dq, const float* dk, const float* dv, const float* x, const float* wq,
                        const float* wk, const float* wv, const float* bnq_weight, const float* bnq_bias,
                        const float* bnk_weight, const float* bnk_bias, const float* wpk, const float* wpv,
                        const float* zqk, const float* vv, const float* kk, const float* pooled_k,
                        const float* pooled_v, const float* save_mean, const float* save_invstd, int B, int C, int Kc,
                        int Vc, int H, int W, int n_sizes, const int* sizes, int training, float* dx, float* dwqk,
                        float* dwv, float* dbnq_weight, float* dbnq_bias, float* dbnk_weight, float* dbnk_bias,
                        float* dwpk, float* dwpv, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    cabinet::QkvShape s;
    if (int rc = qkv_shape(B, C, Kc, Vc, H, W, n_sizes, sizes, s, "cab_qkv_bwd")) return rc;
    if (!dq || !dk || !dv || !x || !wq || !wk || !wv || !bnq_weight || !bnq_bias || !bnk_weight || !bnk_bias || !wpk ||
        !wpv || !zqk || !vv || !kk || !pooled_k || !pooled_v || !save_mean || !save_invstd || !dx || !dwqk || !dwv ||
        !dbnq_weight || !dbnq_bias || !dbnk_weight || !dbnk_bias || !dwpk || !dwpv)
        return fail(CABINET_ERR_INVALID_ARG, "cab_qkv_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("cab_qkv_bwd", dq, dk, dv, x, zqk, vv, kk, dx);
    const size_t need = cabinet::qkv_bwd_workspace(s);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "cab_qkv_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    const cabinet::QkvParams w{wq, wk, wv, bnq_weight, bnq_bias, bnk_weight, bnk_bias, nullptr, nullptr, nullptr,
                               nullptr, wpk, wpv};
    const cabinet::QkvSaved sv{const_cast<float*>(zqk), const_cast<float*>(vv), const_cast<float*>(kk),
                               const_cast<float*>(pooled_k), const_cast<float*>(pooled_v),
                               const_cast<float*>(save_mean), const_cast<float*>(save_invstd)};
    const cabinet::QkvGrads gr{dx, dwqk, dwv, dbnq_weight, dbnq_bias, dbnk_weight, dbnk_bias, dwpk, dwpv};
    return hip_status(cabinet::qkv_bwd_run(s, w, dq, dk, dv, x, training, sv, gr, workspace,
                                           static_cast<hipStream_t>(stream)),
                      "cab_qkv_bwd launch");
}

// ------------------------------------------------------ bias-free 1x1 convolution
static int check_conv1x1(int B, int Ci, int Co, int P, const char* who) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || P <= 0) return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if ((Ci % 4) || (Co % 4))
        return fail(CABINET_ERR_UNSUPPORTED, "%s: Ci=%d, Co=%d must be multiples of 4", who, Ci, Co);
    if (B > 65535) return fail(CABINET_ERR_UNSUPPORTED, "%s: B exceeds grid limits", who);
    return CABINET_OK;
}

size_t cabinet_conv1x1_fwd_workspace_bytes(int Ci, int Co) {
    return Ci > 0 && Co > 0 ? cabinet::conv1x1_fwd_workspace(Ci, Co) : 0;
}

int cabinet_conv1x1_fwd(const float* x, const float* w, int B, int Ci, int Co, int P, float* y, void* workspace,
                        size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_conv1x1(B, Ci, Co, P, "conv1x1_fwd")) return rc;
    if (!x || !w || !y) return fail(CABINET_ERR_INVALID_ARG, "conv1x1_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("conv1x1_fwd", x, w, y);
    const size_t need = cabinet::conv1x1_fwd_workspace(Ci, Co);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "conv1x1_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::conv1x1_fwd_run(x, w, B, Ci, Co, P, y, workspace, static_cast<hipStream_t>(stream)),
                      "conv1x1_fwd launch");
}

int cabinet_conv1x1_bias_supported(int B, int Ci, int Co, int P) {
    return (B > 0 && Ci > 0 && Co > 0 && P > 0 && cabinet::conv1x1_bias_supported(B, Ci, Co, P)) ? 1 : 0;
}

int cabinet_conv1x1_bias_fwd(const float* x, const float* w, const float* bias, int B, int Ci, int Co, int P, float* y, void* workspace,
                             size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_conv1x1(B, Ci, Co, P, "conv1x1_bias_fwd")) return rc;
    if (!x || !w || !bias || !y) return fail(CABINET_ERR_INVALID_ARG, "conv1x1_bias_fwd: null tensor pointer");
    if (!cabinet::conv1x1_bias_supported(B, Ci, Co, P))
        return fail(CABINET_ERR_UNSUPPORTED, "conv1x1_bias_fwd: (B=%d, Ci=%d, Co=%d, P=%d) is outside the small-grid path "
                    "(cabinet_conv1x1_bias_supported)", B, Ci, Co, P);
    CABINET_REQUIRE_ALIGNED("conv1x1_bias_fwd", x, w, y);
    const size_t need = cabinet::conv1x1_fwd_workspace(Ci, Co);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "conv1x1_bias_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::conv1x1_fwd_run(x, w, B, Ci, Co, P, y, workspace, static_cast<hipStream_t>(stream), bias),
                      "conv1x1_bias_fwd launch");
}

int cabinet_channel_sum(const float* d, int B, int C, int P, float* out, cabinet_stream_t stream) {
    if (B <= 0 || C <= 0 || P <= 0) return fail(CABINET_ERR_INVALID_ARG, "channel_sum: non-positive dimension");
    if (!d || !out) return fail(CABINET_ERR_INVALID_ARG, "channel_sum: null tensor pointer");
    return hip_status(cabinet::channel_sum_run(d, B, C, P, out, static_cast<hipStream_t>(stream)), "channel_sum launch");
}

size_t cabinet_conv1x1_bwd_workspace_bytes(int B, int Ci, int Co, int P) {
    return B > 0 && Ci > 0 && Co > 0 && P > 0 ? cabinet::conv1x1_bwd_workspace(B, Ci, Co, P) : 0;
}

int cabinet_conv1x1_bwd(const float* dy, const float* x, const float* w, int B, int Ci, int Co, int P, float* dx,
                        float* dw, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_conv1x1(B, Ci, Co, P, "conv1x1_bwd")) return rc;
    if (!dy || !x || !w) return fail(CABINET_ERR_INVALID_ARG, "conv1x1_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("conv1x1_bwd", dy, x, w, dx, dw);
    const size_t need = cabinet::conv1x1_bwd_workspace(B, Ci, Co, P);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "conv1x1_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::conv1x1_bwd_run(dy, x, w, B, Ci, Co, P, dx, dw, workspace, static_cast<hipStream_t>(stream)),
                      "conv1x1_bwd launch");
}

// ------------------------------------------------------ BatchNorm2d + activation
static int check_bn_act(int B, int C, int P, int act, const char* who) {
    if (B <= 0 || C <= 0 || P <= 0) return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if (act < 0 || act > 2) return fail(CABINET_ERR_INVALID_ARG, "%s: act=%d (0 none, 1 relu, 2 hardswish)", who, act);
    if ((long long)B * C * ((P + 8191) / 8192) > 2147483647LL)
        return fail(CABINET_ERR_UNSUPPORTED, "%s: grid too large", who);
    return CABINET_OK;
}

size_t cabinet_bn_act_workspace_bytes(int B, int C, int P) {
    return B > 0 && C > 0 && P > 0 ? cabinet::bn_act_workspace(B, C, P) : 0;
}

int cabinet_bn_act_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                       const float* residual, int B, int C, int P, int act, int training, float momentum, float eps, float* y,
                       float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                       cabinet_stream_t stream) {
    if (int rc = check_bn_act(B, C, P, act, "bn_act_fwd")) return rc;
    if (!x || !weight || !bias || !running_mean || !running_var || !y || !save_mean || !save_invstd)
        return fail(CABINET_ERR_INVALID_ARG, "bn_act_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("bn_act_fwd", x, residual, y);
    const size_t need = cabinet::bn_act_workspace(B, C, P);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "bn_act_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::bn_act_fwd_run(x, weight, bias, running_mean, running_var, residual, B, C, P, act, training,
                                              momentum, eps, y, save_mean, save_invstd, workspace,
                                              static_cast<hipStream_t>(stream)),
                      "bn_act_fwd launch");
}

int cabinet_bn_act_fwd_part(const float* x, const float* conv_part, const float* weight, const float* bias, float* running_mean,
                            float* running_var, const float* residual, int B, int C, int H, int W, int act, int training,
                            float momentum, float eps, float* y, float* save_mean, float* save_invstd, cabinet_stream_t stream) {
    if (H <= 0 || W <= 0) return fail(CABINET_ERR_INVALID_ARG, "bn_act_fwd_part: non-positive dimension");
    if (int rc = check_bn_act(B, C, H * W, act, "bn_act_fwd_part")) return rc;
    if (!x || !weight || !bias || !running_mean || !running_var || !y || !save_mean || !save_invstd || (training && !conv_part))
        return fail(CABINET_ERR_INVALID_ARG, "bn_act_fwd_part: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("bn_act_fwd_part", x, residual, y);
    return hip_status(cabinet::bn_act_fwd_part_run(x, conv_part, H, W, weight, bias, running_mean, running_var, residual, B, C, act,
                                                   training, momentum, eps, y, save_mean, save_invstd,
                                                   static_cast<hipStream_t>(stream)),
                      "bn_act_fwd_part launch");
}

int cabinet_bn_act_bwd(const float* dy, const float* x, const float* weight, const float* bias, const float* save_mean,
                       const float* save_invstd, int B, int C, int P, int act, int training, float* dx, float* dweight,
                       float* dbias, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_bn_act(B, C, P, act, "bn_act_bwd")) return rc;
    if (!dy || !x || !weight || !bias || !save_mean || !save_invstd || !dx || !dweight || !dbias)
        return fail(CABINET_ERR_INVALID_ARG, "bn_act_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("bn_act_bwd", dy, x, dx);
    const size_t need = cabinet::bn_act_workspace(B, C, P);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "bn_act_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::bn_act_bwd_run(dy, x, weight, bias, save_mean, save_invstd, B, C, P, act, training, dx,
                                              dweight, dbias, workspace, static_cast<hipStream_t>(stream)),
                      "bn_act_bwd launch");
}

// ------------------------------------------------------ BatchNorm + ReLU + 1x1 classifier (K12)
int cabinet_bn_cls_supported(int C, int K, int P) { return cabinet::bn_cls_supported(C, K, P) ? 1 : 0; }
int cabinet_bn_cls_table_floats(int C, int K) { return (C > 0 && K > 0 && K <= 32) ? cabinet::bn_cls_table_floats(C, K) : 0; }

static int check_bn_cls(int B, int C, int K, int H, int W, const char* who) {
    if (B <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if (!cabinet::bn_cls_supported(C, K, H * W))
        return fail(CABINET_ERR_UNSUPPORTED, "%s: (C=%d, K=%d, H*W=%d) outside C %% 64 == 0, K <= 32, (H*W) %% 4 == 0", who, C, K, H * W);
    if ((long long)B * ((H * W + 63) / 64) > 2147483647LL || (long long)B * C * H * W >= (1LL << 40))
        return fail(CABINET_ERR_UNSUPPORTED, "%s: grid too large", who);
    return CABINET_OK;
}

size_t cabinet_bn_cls_fwd_workspace_bytes(int B, int C, int P) {
    return (B > 0 && C > 0 && P > 0) ? cabinet::bn_cls_fwd_workspace(B, C, P) : 0;
}

int cabinet_bn_cls_fwd(const float* z, const float* conv_part, const float* bn_weight, const float* bn_bias, float* running_mean,
                       float* running_var, const float* w_cls, const float* bias, int B, int C, int K, int H, int W, int training,
                       float momentum, float eps, float* y, float* table, void* workspace, size_t workspace_bytes,
                       cabinet_stream_t stream) {
    if (int rc = check_bn_cls(B, C, K, H, W, "bn_cls_fwd")) return rc;
    if (!z || !bn_weight || !bn_bias || !running_mean || !running_var || !w_cls || !y || !table)
        return fail(CABINET_ERR_INVALID_ARG, "bn_cls_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("bn_cls_fwd", z, y, table);
    const size_t need = cabinet::bn_cls_fwd_workspace(B, C, H * W);
    if (!workspace || workspace_bytes < need) return fail(CABINET_ERR_WORKSPACE, "bn_cls_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::bn_cls_fwd_run(z, conv_part, bn_weight, bn_bias, running_mean, running_var, w_cls, bias, B, C, K, H, W,
                                              training, momentum, eps, y, table, workspace, static_cast<hipStream_t>(stream)),
                      "bn_cls_fwd launch");
}

size_t cabinet_bn_cls_bwd_workspace_bytes(int B, int C, int K, int P) {
    return (B > 0 && C > 0 && K > 0 && K <= 32 && P > 0) ? cabinet::bn_cls_bwd_workspace(B, C, K, P) : 0;
}

int cabinet_bn_cls_bwd(const float* dy, const float* z, const float* table, int B, int C, int K, int H, int W, int training, float* dz,
                       float* dbn_weight, float* dbn_bias, float* dw_cls, float* dbias, void* workspace, size_t workspace_bytes,
                       cabinet_stream_t stream) {
    if (int rc = check_bn_cls(B, C, K, H, W, "bn_cls_bwd")) return rc;
    if (!dy || !z || !table || !dz || !dbn_weight || !dbn_bias || !dw_cls) return fail(CABINET_ERR_INVALID_ARG, "bn_cls_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("bn_cls_bwd", dy, z, table, dz);
    const size_t need = cabinet::bn_cls_bwd_workspace(B, C, K, H * W);
    if (!workspace || workspace_bytes < need) return fail(CABINET_ERR_WORKSPACE, "bn_cls_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::bn_cls_bwd_run(dy, z, table, B, C, K, H, W, training, dz, dbn_weight, dbn_bias, dw_cls, dbias, workspace,
                                              static_cast<hipStream_t>(stream)),
                      "bn_cls_bwd launch");
}

// ------------------------------------------------------ depthwise convolution
static int check_dwconv(int B, int C, int H, int W, int K, int S, const char* who) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if (!cabinet::dwconv_supported(K, S))
        return fail(CABINET_ERR_UNSUPPORTED, "%s: kernel %d stride %d (kernel 3|5, stride 1|2)", who, K, S);
    if ((long long)B * C * ((H + 15) / 16) * ((W + 7) / 8) > 2147483647LL)
        return fail(CABINET_ERR_UNSUPPORTED, "%s: grid too large", who);
    return CABINET_OK;
}

int cabinet_dwconv_supported(int K, int stride) { return cabinet::dwconv_supported(K, stride) ? 1 : 0; }

int cabinet_dwconv_fwd(const float* x, const float* weight, int B, int C, int H, int W, int K, int stride, float* y,
                       cabinet_stream_t stream) {
    if (int rc = check_dwconv(B, C, H, W, K, stride, "dwconv_fwd")) return rc;
    if (!x || !weight || !y) return fail(CABINET_ERR_INVALID_ARG, "dwconv_fwd: null tensor pointer");
    return hip_status(cabinet::dwconv_fwd_run(x, weight, B, C, H, W, K, stride, y, static_cast<hipStream_t>(stream)),
                      "dwconv_fwd launch");
}

size_t cabinet_dwconv_bwd_workspace_bytes(int B, int C, int H, int W, int K, int stride) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !cabinet::dwconv_supported(K, stride)) return 0;
    return cabinet::dwconv_bwd_workspace(B, C, H, W, K);
}

int cabinet_dwconv_bwd(const float* dy, const float* x, const float* weight, int B, int C, int H, int W, int K,
                       int stride, float* dx, float* dw, void* workspace, size_t workspace_bytes,
                       cabinet_stream_t stream) {
    if (int rc = check_dwconv(B, C, H, W, K, stride, "dwconv_bwd")) return rc;
    if (!dy || !x || !weight || !dx || !dw) return fail(CABINET_ERR_INVALID_ARG, "dwconv_bwd: null tensor pointer");
    const size_t need = cabinet::dwconv_bwd_workspace(B, C, H, W, K);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "dwconv_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::dwconv_bwd_run(dy, x, weight, B, C, H, W, K, stride, dx, dw, workspace,
                                              static_cast<hipStream_t>(stream)),
                      "dwconv_bwd launch");
}

// ------------------------------------------------------ channel gate + activation
int cabinet_gate_act_fwd(const float* x, const float* gate, int B, int C, int P, int act, float* y,
                         cabinet_stream_t stream) {
    if (int rc = check_bn_act(B, C, P, act, "gate_act_fwd")) return rc;
    if (!x || !gate || !y) return fail(CABINET_ERR_INVALID_ARG, "gate_act_fwd: null tensor pointer");
    return hip_status(cabinet::gate_act_fwd_run(x, gate, B * C, P, act, y, static_cast<hipStream_t>(stream)),
                      "gate_act_fwd launch");
}

size_t cabinet_gate_act_bwd_workspace_bytes(int B, int C, int P) {
    return B > 0 && C > 0 && P > 0 ? cabinet::gate_act_workspace(B * C, P) : 0;
}

int cabinet_gate_act_bwd(const float* dy, const float* x, const float* gate, int B, int C, int P, int act, float* dx,
                         float* dgate, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_bn_act(B, C, P, act, "gate_act_bwd")) return rc;
    if (!dy || !x || !gate || !dx || !dgate) return fail(CABINET_ERR_INVALID_ARG, "gate_act_bwd: null tensor pointer");
    const size_t need = cabinet::gate_act_workspace(B * C, P);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "gate_act_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::gate_act_bwd_run(dy, x, gate, B * C, P, act, dx, dgate, workspace,
                                                static_cast<hipStream_t>(stream)),
                      "gate_act_bwd launch");
}

// ------------------------------------------------------ BatchNorm (+act) -> depthwise convolution
size_t cabinet_bn_dwconv_fwd_workspace_bytes(int B, int C, int H, int W) {
    return B > 0 && C > 0 && H > 0 && W > 0 ? cabinet::bn_dwconv_fwd_workspace(B, C, H, W) : 0;
}

int cabinet_bn_dwconv_fwd(const float* z, const float* bn_weight, const float* bn_bias, float* running_mean,
                          float* running_var, const float* conv_weight, int B, int C, int H, int W, int K, int stride,
                          int act, int training, float momentum, float eps, float* y, float* save_mean,
                          float* save_invstd, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_dwconv(B, C, H, W, K, stride, "bn_dwconv_fwd")) return rc;
    if (int rc = check_bn_act(B, C, H * W, act, "bn_dwconv_fwd")) return rc;
    if (!z || !bn_weight || !bn_bias || !running_mean || !running_var || !conv_weight || !y || !save_mean || !save_invstd)
        return fail(CABINET_ERR_INVALID_ARG, "bn_dwconv_fwd: null tensor pointer");
    const size_t need = cabinet::bn_dwconv_fwd_workspace(B, C, H, W);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "bn_dwconv_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::bn_dwconv_fwd_run(z, bn_weight, bn_bias, running_mean, running_var, conv_weight, B, C, H, W,
                                                 K, stride, act, training, momentum, eps, y, save_mean, save_invstd,
                                                 workspace, static_cast<hipStream_t>(stream)),
                      "bn_dwconv_fwd launch");
}

size_t cabinet_bn_dwconv_bwd_workspace_bytes(int B, int C, int H, int W, int K, int stride) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !cabinet::dwconv_supported(K, stride)) return 0;
    return cabinet::bn_dwconv_bwd_workspace(B, C, H, W, K);
}

int cabinet_bn_dwconv_bwd(const float* dy, const float* z, const float* bn_weight, const float* bn_bias,
                          const float* save_mean, const float* save_invstd, const float* conv_weight, int B, int C,
                          int H, int W, int K, int stride, int act, int training, float* dz, float* dbn_weight,
                          float* dbn_bias, float* dconv_weight, void* workspace, size_t workspace_bytes,
                          cabinet_stream_t stream) {
    if (int rc = check_dwconv(B, C, H, W, K, stride, "bn_dwconv_bwd")) return rc;
    if (int rc = check_bn_act(B, C, H * W, act, "bn_dwconv_bwd")) return rc;
    if (!dy || !z || !bn_weight || !bn_bias || !save_mean || !save_invstd || !conv_weight || !dz || !dbn_weight ||
        !dbn_bias || !dconv_weight)
        return fail(CABINET_ERR_INVALID_ARG, "bn_dwconv_bwd: null tensor pointer");
    const size_t need = cabinet::bn_dwconv_bwd_workspace(B, C, H, W, K);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "bn_dwconv_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::bn_dwconv_bwd_run(dy, z, bn_weight, bn_bias, save_mean, save_invstd, conv_weight, B, C, H,
                                                 W, K, stride, act, training, dz, dbn_weight, dbn_bias, dconv_weight,
                                                 workspace, static_cast<hipStream_t>(stream)),
                      "bn_dwconv_bwd launch");
}

// ------------------------------------------------------ 7x7 stride-2 stem convolution
static int check_stem(int B, int H, int W, const char* who) {
    if (B <= 0 || H <= 0 || W <= 0) return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if ((long long)B * ((H + 1) / 2) > 2147483647LL / 4) return fail(CABINET_ERR_UNSUPPORTED, "%s: grid too large", who);
    return CABINET_OK;
}

int cabinet_stem_conv_fwd(const float* x, const float* weight, int B, int H, int W, float* y, cabinet_stream_t stream) {
    if (int rc = check_stem(B, H, W, "stem_conv_fwd")) return rc;
    if (!x || !weight || !y) return fail(CABINET_ERR_INVALID_ARG, "stem_conv_fwd: null tensor pointer");
    return hip_status(cabinet::stem_conv_fwd_run(x, weight, B, H, W, y, static_cast<hipStream_t>(stream)),
                      "stem_conv_fwd launch");
}

size_t cabinet_stem_conv_wrw_workspace_bytes(int B, int H, int W) {
    return B > 0 && H > 0 && W > 0 ? cabinet::stem_conv_wrw_workspace(B, H, W) : 0;
}

int cabinet_stem_conv_wrw(const float* dy, const float* x, int B, int H, int W, float* dw, void* workspace,
                          size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_stem(B, H, W, "stem_conv_wrw")) return rc;
    if (!dy || !x || !dw) return fail(CABINET_ERR_INVALID_ARG, "stem_conv_wrw: null tensor pointer");
    const size_t need = cabinet::stem_conv_wrw_workspace(B, H, W);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "stem_conv_wrw: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::stem_conv_wrw_run(dy, x, B, H, W, dw, workspace, static_cast<hipStream_t>(stream)),
                      "stem_conv_wrw launch");
}

// ------------------------------------------------------ thin pointwise convolution
static int check_pwconv(int B, int Ci, int Co, int P, const char* who) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || P <= 0) return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if (!cabinet::pwconv_supported(Ci, Co, P))
        return fail(CABINET_ERR_UNSUPPORTED, "%s: Ci=%d, Co=%d (multiples of 8, <= 120, block product <= 8)", who, Ci, Co);
    return CABINET_OK;
}

int cabinet_pwconv_supported(int Ci, int Co, int P) {
    return Ci > 0 && Co > 0 && P > 0 && cabinet::pwconv_supported(Ci, Co, P) ? 1 : 0;
}

int cabinet_pwconv_fwd(const float* x, const float* w, int B, int Ci, int Co, int P, float* y, cabinet_stream_t stream) {
    if (int rc = check_pwconv(B, Ci, Co, P, "pwconv_fwd")) return rc;
    if (!x || !w || !y) return fail(CABINET_ERR_INVALID_ARG, "pwconv_fwd: null tensor pointer");
    return hip_status(cabinet::pwconv_fwd_run(x, w, B, Ci, Co, P, y, static_cast<hipStream_t>(stream)),
                      "pwconv_fwd launch");
}

size_t cabinet_pwconv_bwd_workspace_bytes(int B, int Ci, int Co, int P) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || P <= 0 || !cabinet::pwconv_supported(Ci, Co, P)) return 0;
    return cabinet::pwconv_bwd_workspace(B, Ci, Co, P);
}

int cabinet_pwconv_bwd(const float* dy, const float* x, const float* w, int B, int Ci, int Co, int P, float* dx,
                       float* dw, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    if (int rc = check_pwconv(B, Ci, Co, P, "pwconv_bwd")) return rc;
    if (!dy || !x || !w) return fail(CABINET_ERR_INVALID_ARG, "pwconv_bwd: null tensor pointer");
    const size_t need = cabinet::pwconv_bwd_workspace(B, Ci, Co, P);
    if (dw && (!workspace || workspace_bytes < need))
        return fail(CABINET_ERR_WORKSPACE, "pwconv_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::pwconv_bwd_run(dy, x, w, B, Ci, Co, P, dx, dw, workspace, static_cast<hipStream_t>(stream)),
                      "pwconv_bwd launch");
}

// ------------------------------------------------------ dense 3x3 convolution (Winograd F(2x2,3x3), fp32 MFMA)
static int check_conv3x3(const cabinet::WinoShape& s, const char* who) {
    if (s.B <= 0 || s.C0 <= 0 || s.C1 < 0 || s.K <= 0 || s.H <= 0 || s.W <= 0)
        return fail(CABINET_ERR_INVALID_ARG, "%s: non-positive dimension", who);
    if (!cabinet::conv3x3_shape_ok(s))
        return fail(CABINET_ERR_UNSUPPORTED,
                    "%s: C0=%d C1=%d Co=%d H=%d W=%d (C0, C1 multiples of 16; Co, C0+C1 multiples of 64; C0 %% 64 == 0 when C1 > 0; "
                    "one image's tensors < 1 GiB)", who, s.C0, s.C1, s.K, s.H, s.W);
    return CABINET_OK;
}

int cabinet_conv3x3_supported(int C0, int C1, int Co) { return cabinet::conv3x3_supported(C0, C1, Co) ? 1 : 0; }

int cabinet_conv3x3_tile_blocks(int B, int H, int W) {
    return B > 0 && H > 0 && W > 0 ? cabinet::conv3x3_tile_blocks(B, H, W) : 0;
}

size_t cabinet_conv3x3_fwd_workspace_bytes(int B, int C0, int C1, int Co, int H, int W) {
    const cabinet::WinoShape s{B, C0, C1, Co, H, W};
    return B > 0 && H > 0 && W > 0 && cabinet::conv3x3_shape_ok(s) ? cabinet::conv3x3_fwd_workspace(s) : 0;
}

int cabinet_conv3x3_fwd(const float* x0, const float* x1, const float* w, int B, int C0, int C1, int Co, int H, int W, float* y,
                        float* bn_part, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    const cabinet::WinoShape s{B, C0, C1, Co, H, W};
    if (int rc = check_conv3x3(s, "conv3x3_fwd")) return rc;
    if (!x0 || !w || !y || (C1 > 0 && !x1)) return fail(CABINET_ERR_INVALID_ARG, "conv3x3_fwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("conv3x3_fwd", x0, x1, w, y, bn_part);
    const size_t need = cabinet::conv3x3_fwd_workspace(s);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "conv3x3_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::conv3x3_fwd_run(s, x0, C1 > 0 ? x1 : nullptr, w, y, bn_part, workspace, static_cast<hipStream_t>(stream)),
                      "conv3x3_fwd launch");
}

size_t cabinet_conv3x3_bwd_workspace_bytes(int B, int C0, int C1, int Co, int H, int W) {
    const cabinet::WinoShape s{B, C0, C1, Co, H, W};
    return B > 0 && H > 0 && W > 0 && cabinet::conv3x3_shape_ok(s) ? cabinet::conv3x3_bwd_workspace(s) : 0;
}

int cabinet_conv3x3_bwd(const float* dy, const float* x0, const float* x1, const float* w, int B, int C0, int C1, int Co, int H,
                        int W, float* dx0, float* dx1, float* dw, void* workspace, size_t workspace_bytes, cabinet_stream_t stream) {
    const cabinet::WinoShape s{B, C0, C1, Co, H, W};
    if (int rc = check_conv3x3(s, "conv3x3_bwd")) return rc;
    if (!dy || !w || (dw && (!x0 || (C1 > 0 && !x1))) || (dx0 && C1 > 0 && !dx1))
        return fail(CABINET_ERR_INVALID_ARG, "conv3x3_bwd: null tensor pointer");
    CABINET_REQUIRE_ALIGNED("conv3x3_bwd", dy, x0, x1, w, dx0, dx1, dw);
    const size_t need = cabinet::conv3x3_bwd_workspace(s);
    if (!workspace || workspace_bytes < need)
        return fail(CABINET_ERR_WORKSPACE, "conv3x3_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return hip_status(cabinet::conv3x3_bwd_run(s, dy, x0, C1 > 0 ? x1 : nullptr, w, dx0, C1 > 0 ? dx1 : nullptr, dw, workspace,
                                               static_cast<hipStream_t>(stream)),
                      "conv3x3_bwd launch");
}

}  // extern "C"
