// Activations that follow a BatchNorm2d in the model, shared by bn_act.hip and dwconv.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace cabinet {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2 };

__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == ACT_RELU) return fmaxf(u, 0.f);
    // mobilenetv3.py:48-50,63-65.  The division by six is a multiplication by the rounded reciprocal (<= 1 ulp from the
    // quotient): an IEEE fp32 division is ~10 instructions, and it made the BatchNorm+HardSwish apply pass spend 25 VALU
    // instructions per element (VALU pipes 50 % busy on a pass that should be waiting for HBM only)
    if (act == ACT_HSWISH) return u * fminf(fmaxf(u + 3.f, 0.f), 6.f) * (1.f / 6.f);
    return u;
}
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (act == ACT_HSWISH) {
        // d/du [u * relu6(u+3)/6] with relu6' = 1 on the open interval (0,6), as ATen's hardtanh backward
        const float t = u + 3.f;
        const float inner = (t > 0.f && t < 6.f) ? u * (1.f / 6.f) : 0.f;
        return fminf(fmaxf(t, 0.f), 6.f) * (1.f / 6.f) + inner;
    }
    return 1.f;
}

// A BatchNorm2d (+activation) folded into the loader of the kernel that consumes its output: the consumer reads the
// raw pre-normalisation tensor z and evaluates a = act(gamma * (z - mean) * invstd + beta) per element.
struct BnFold {
    const float *mean, *invstd, *weight, *bias;  // per channel; mean == nullptr: no fold, the input is used as is
    int act;
};

// BatchNorm pieces of bn_act.hip that a fused producer/consumer chain reuses
size_t bn_act_workspace(int B, int C, int P);
hipError_t bn_stats_run(const float* x, float* running_mean, float* running_var, int B, int C, int P, int training,
                        float momentum, float eps, float* save_mean, float* save_invstd, void* ws, hipStream_t stream);
// part: [2][C][nt] partial sums (sum du, sum du*xhat) already written by the caller's kernel
hipError_t bn_bwd_tail_run(const float* part, int nt, const float* dy, const float* x, const float* weight,
                           const float* bias, const float* save_mean, const float* save_invstd, int B, int C, int P,
                           int act, int training, float* dx, float* dweight, float* dbias, float* coef,
                           hipStream_t stream);

}  // namespace cabinet
