// BatchNorm finalize shared by K7 (bn_act.hip) and K12 (bn_cls.hip): one 256-thread workgroup per channel merges per-chunk (mean, M2)
// partials with Chan's formula in double, in a fixed order, and updates the running buffers exactly like nn.BatchNorm2d.
#pragma once
#include "common.hpp"

namespace cabinet {

constexpr int BA_T = 256;
constexpr int BA_V = 8;                     // float4 per thread and chunk
constexpr int BA_CHUNK = BA_T * 4 * BA_V;   // 8192 elements

// part[0][c][tile] = chunk mean, part[1][c][tile] = chunk M2 (sum of squared deviations), tile = b * chunks + chunk
// conv_h > 0: the partials come from the epilogue of the 3x3 convolution that produced x (conv3x3_wino.hip: one (mean, M2) pair per
// channel and tile block of 4 x 32 output pixels, blocks ordered image, block row, block column) instead of bn_act_stats_kernel.
// Thread 0 writes save_mean[c] / save_invstd[c] (eval mode: from the running buffers).  Call from all BA_T threads of the block.
__device__ __forceinline__ void bn_finalize_channel(const float* __restrict__ part, int c, int B, int C, int P, int chunks, int conv_h,
                                                    int conv_w, int training, float momentum, float eps,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ double dred[4];
    __shared__ double s_mean;
    if (!training) {
        if (threadIdx.x == 0) {
            save_mean[c] = running_mean[c];
            save_invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
        }
        return;
    }
    const int nbx = conv_h > 0 ? (conv_w + 31) / 32 : 1, nbi = conv_h > 0 ? ((conv_h + 3) / 4) * nbx : chunks;
    const int nt = B * nbi;
    const float* pm = part + (size_t)c * nt;
    const float* p2 = part + ((size_t)C + c) * nt;
    auto block_sum_d = [&](double v) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = v;
        __syncthreads();
        const double t = (dred[0] + dred[1]) + (dred[2] + dred[3]);
        __syncthreads();
        return t;
    };
    auto count_of = [&](int t) {
        const int r = t % nbi;
        if (conv_h > 0) return (double)(min(4, conv_h - 4 * (r / nbx)) * min(32, conv_w - 32 * (r % nbx)));
        return (double)(min((r + 1) * BA_CHUNK, P) - r * BA_CHUNK);
    };
    const double N = (double)B * (double)P;
    double s = 0.0;
    for (int t = threadIdx.x; t < nt; t += BA_T) s += count_of(t) * (double)pm[t];
    s = block_sum_d(s);
    if (threadIdx.x == 0) s_mean = s / N;
    __syncthreads();
    const double mean = s_mean;
    double m2 = 0.0;
    for (int t = threadIdx.x; t < nt; t += BA_T) {
        const double d = (double)pm[t] - mean;
        m2 += (double)p2[t] + count_of(t) * d * d;
    }
    m2 = block_sum_d(m2);
    if (threadIdx.x == 0) {
        const double var = m2 / N;
        save_mean[c] = (float)mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = N > 1.0 ? m2 / (N - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

}  // namespace cabinet
