// Activations that follow a BatchNorm2d in the model, shared by bn_act.hip and dwconv.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace cabinet {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2 };

// x / 6, correctly rounded, in three instructions: q = RN(x * RN(1/6)), r = x - 6 q (exact under FMA), q + r * RN(1/6)
// (Markstein's refinement: equal to the IEEE quotient bit for bit -- checked on 22 M values).  The compiler's expansion of a
// division is ~10 instructions and made the BatchNorm+HardSwish apply pass spend 25 VALU instructions per element; the
// plain product x * (1/6) is 1 ulp off for a third of the inputs, and that systematic perturbation was enough to move the
// full-model gradient parity of the CAB's small maps from 1e-4 to 1e-2 (ReLU mask flips), so it is not used.
__device__ __forceinline__ float div6(float x) {
    const float c = 1.f / 6.f;
    const float q = x * c;
    return fmaf(fmaf(-6.f, q, x), c, q);
}

__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == ACT_RELU) return fmaxf(u, 0.f);
    if (act == ACT_HSWISH) return div6(u * fminf(fmaxf(u + 3.f, 0.f), 6.f));  // mobilenetv3.py:48-50,63-65
    return u;
}
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (act == ACT_HSWISH) {
        // d/du [u * relu6(u+3)/6] with relu6' = 1 on the open interval (0,6), as ATen's hardtanh backward
        const float t = u + 3.f;
        const float inner = (t > 0.f && t < 6.f) ? div6(u) : 0.f;
        return div6(fminf(fmaxf(t, 0.f), 6.f)) + inner;
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
