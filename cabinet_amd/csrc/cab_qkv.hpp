// Argument blocks of the K6 q/k/v producer (cab_qkv.hip), shared with the C-ABI layer (capi.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cabinet {

struct QkvShape {
    int B, C, Kc, Vc, H, W;
    int ns, sizes[4];  // PSP pyramid sizes (reference default 1,3,6,8)
};

struct QkvParams {
    const float *wq, *wk, *wv;                    // (Kc,C) (Kc,C) (Vc,C)
    const float *bnq_w, *bnq_b, *bnk_w, *bnk_b;   // (Kc) each
    float *bnq_rm, *bnq_rv, *bnk_rm, *bnk_rv;     // running statistics, updated in training mode (fwd only)
    const float *wpk, *wpv;                       // (Kc,(ns+1)Kc) (Vc,(ns+1)Vc) PSP projections
};

struct QkvSaved {          // written by fwd, read by bwd
    float* zqk;            // (B,2Kc,P) pre-BN projections [zq | zk]
    float* vv;             // (B,Vc,P)  W_v x
    float* kk;             // (B,Kc,P)  relu(bn(zk))
    float* pooled_k;       // (B,ns*Kc,NBp) block-expanded pooled bins of kk
    float* pooled_v;       // (B,ns*Vc,NBp) ... of vv
    float* mean;           // (2Kc) [q | k]
    float* invstd;         // (2Kc)
};

struct QkvGrads {
    float* dx;             // (B,C,P)
    float* dwqk;           // (2Kc,C) = [dW_q; dW_k]
    float* dwv;            // (Vc,C)
    float *dbnq_w, *dbnq_b, *dbnk_w, *dbnk_b;  // (Kc) each
    float *dwpk, *dwpv;    // like wpk, wpv
};

const char* qkv_unsupported(const QkvShape& s);  // nullptr if supported, else what is required
int qkv_padded_bins(const QkvShape& s);          // NBp: pyramid bins (sum s^2) rounded up to a multiple of 4
size_t qkv_fwd_workspace(const QkvShape& s);
size_t qkv_bwd_workspace(const QkvShape& s);
hipError_t qkv_fwd_run(const QkvShape& s, const QkvParams& w, const float* x, int training, float momentum, float eps,
                       const QkvSaved& sv, float* q, float* k, float* v, void* ws, hipStream_t stream);
hipError_t qkv_bwd_run(const QkvShape& s, const QkvParams& w, const float* dq, const float* dk, const float* dv,
                       const float* x, int training, const QkvSaved& sv, const QkvGrads& gr, void* ws,
                       hipStream_t stream);

// output stage of the forward as one kernel (cab_qkv_fused.hip): pyramid terms per workgroup, W_0 product, bilinear gathers
bool qkv_fused_fwd_supported(const QkvShape& s);
hipError_t qkv_fused_out(const QkvShape& s, const QkvParams& w, const QkvSaved& sv, float* k, float* v, hipStream_t stream);
// dx of the three projections as one LDS-free kernel (cab_qkv_fused.hip)
bool qkv_dx_supported(const QkvShape& s);
hipError_t qkv_dx_run(const QkvShape& s, const QkvParams& w, const float* dzqk, const float* dvv, float* dx, hipStream_t stream);

size_t conv1x1_fwd_workspace(int Ci, int Co);
size_t conv1x1_bwd_workspace(int B, int Ci, int Co, int P);
hipError_t conv1x1_fwd_run(const float* x, const float* wgt, int B, int Ci, int Co, int P, float* y, void* ws,
                           hipStream_t stream, const float* bias = nullptr);
bool conv1x1_bias_supported(int B, int Ci, int Co, int P);   // the small-grid path (CAB resolution) takes an output bias
hipError_t conv1x1_bwd_run(const float* dy, const float* x, const float* wgt, int B, int Ci, int Co, int P, float* dx,
                           float* dw, void* ws, hipStream_t stream);

}  // namespace cabinet
