// K7 -- BatchNorm2d fused with the activation that follows it, forward and backward (NCHW fp32).
//
// Replaces the `bn -> relu` tail of reference ConvBNReLU.forward (src/models/cabinet.py:42-44; SURVEY.md section 8
// row a6 lists ConvBNReLU with the FFM) and the same BatchNorm2d -> ReLU / HardSwish pairs elsewhere in the model
// (cabinet.py:59-63,67-68; mobilenetv3.py:86-99,118-152 with HardSwish of mobilenetv3.py:53-65).  On stock
// PyTorch-ROCm each pair is a MIOpen batch-norm launch plus one (ReLU) to four (HardSwish = add, clamp, div, mul)
// elementwise launches, each a full HBM round trip, and the mirror image in backward; at config 3 the model
// normalises 4.0 GB of activations per step and these launches are ~30 % of the step.
//
// This is HBM-bound streaming work; the plan is the minimum number of passes a training-mode BatchNorm allows:
//   fwd  : stats  (read x)            per-chunk (mean, M2), merged per channel with Chan's formula in double
//          apply  (read x, write y)   y = act(gamma * xhat + beta)
//   bwd  : reduce (read dy, x)        sum du, sum du*xhat with du = dy * act'(u), u recomputed from x
//          dx     (read dy, x, write) dx = gamma*invstd*(du - mean(du) - xhat*mean(du*xhat))
// Nothing but x, mean and invstd is kept for backward (no pre-activation or mask tensor).  One workgroup streams
// one 8192-element chunk of one (b,c) plane with 128-bit loads; partial results are combined in a fixed order
// (no atomics): bitwise reproducible.
#include "act.hpp"
#include "bn_finalize.hpp"
#include "common.hpp"

namespace cabinet {


// chunk of a plane -> registers (zeros past the end); returns the number of valid elements of the chunk
__device__ __forceinline__ int ba_load(const float* __restrict__ row, int P, int lo, f32x4 (&v)[BA_V]) {
    const int hi = min(lo + BA_CHUNK, P);
    if ((P & 3) == 0) {
#pragma unroll
        for (int i = 0; i < BA_V; ++i) {
            const int p = lo + (i * BA_T + threadIdx.x) * 4;
            v[i] = p < hi ? *reinterpret_cast<const f32x4*>(row + p) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
#pragma unroll
        for (int i = 0; i < BA_V; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = lo + (i * BA_T + threadIdx.x) * 4 + e;
                v[i][e] = p < hi ? row[p] : 0.f;
            }
    }
    return hi - lo;
}
__device__ __forceinline__ bool ba_valid(int P, int lo, int i, int e) {
    return lo + (i * BA_T + (int)threadIdx.x) * 4 + e < min(lo + BA_CHUNK, P);
}
__device__ __forceinline__ void ba_store(float* __restrict__ row, int P, int lo, const f32x4 (&v)[BA_V]) {
    const int hi = min(lo + BA_CHUNK, P);
    if ((P & 3) == 0) {
#pragma unroll
        for (int i = 0; i < BA_V; ++i) {
            const int p = lo + (i * BA_T + threadIdx.x) * 4;
            if (p < hi) *reinterpret_cast<f32x4*>(row + p) = v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < BA_V; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = lo + (i * BA_T + threadIdx.x) * 4 + e;
                if (p < hi) row[p] = v[i][e];
            }
    }
}

// part[0][c][tile] = chunk mean, part[1][c][tile] = chunk M2 (sum of squared deviations), tile = b*chunks + chunk
__global__ __launch_bounds__(BA_T) void bn_act_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int B,
                                                             int C, int P, int chunks) {
    __shared__ float red[4];
    const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks, b = row / C, c = row - b * C;
    f32x4 v[BA_V];
    const int cnt = ba_load(x + (size_t)row * P, P, ch * BA_CHUNK, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BA_V; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = block_sum_256(s, red) / (float)cnt;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < BA_V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ba_valid(P, ch * BA_CHUNK, i, e)) m2 += (v[i][e] - mean) * (v[i][e] - mean);
    m2 = block_sum_256(m2, red);
    if (threadIdx.x == 0) {
        const int nt = B * chunks, tile = b * chunks + ch;
        part[(size_t)c * nt + tile] = mean;
        part[((size_t)C + c) * nt + tile] = m2;
    }
}

// one workgroup per channel: merge the chunk statistics (Chan et al.), update the running buffers (bn_finalize.hpp: shared with K12)
__global__ __launch_bounds__(BA_T) void bn_act_finalize_kernel(const float* __restrict__ part, int B, int C, int P,
                                                                int chunks, int conv_h, int conv_w, int training, float momentum, float eps,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var,
                                                                float* __restrict__ save_mean,
                                                                float* __restrict__ save_invstd) {
    bn_finalize_channel(part, blockIdx.x, B, C, P, chunks, conv_h, conv_w, training, momentum, eps, running_mean, running_var, save_mean,
                        save_invstd);
}

__global__ __launch_bounds__(BA_T) void bn_act_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ weight,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ residual, int C, int P,
                                                             int chunks, int act, float* __restrict__ y) {
    const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks, c = row % C;
    const float mu = mean[c], inv = invstd[c], gam = weight[c], bet = bias[c];
    f32x4 v[BA_V];
    ba_load(x + (size_t)row * P, P, ch * BA_CHUNK, v);
#pragma unroll
    for (int i = 0; i < BA_V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = act_fwd(fmaf((v[i][e] - mu) * inv, gam, bet), act);
    if (residual) {  // the identity shortcut of an MBConv block (mobilenetv3.py:158), added in the same pass
        f32x4 r[BA_V];
        ba_load(residual + (size_t)row * P, P, ch * BA_CHUNK, r);
#pragma unroll
        for (int i = 0; i < BA_V; ++i) v[i] += r[i];
    }
    ba_store(y + (size_t)row * P, P, ch * BA_CHUNK, v);
}

// part[0][c][tile] = sum du, part[1][c][tile] = sum du * xhat
__global__ __launch_bounds__(BA_T) void bn_act_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias, int B, int C, int P,
                                                                  int chunks, int act, float* __restrict__ part) {
    __shared__ float red[4];
    const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks, b = row / C, c = row - b * C;
    const float mu = mean[c], inv = invstd[c], gam = weight[c], bet = bias[c];
    f32x4 vx[BA_V], vg[BA_V];
    ba_load(x + (size_t)row * P, P, ch * BA_CHUNK, vx);
    ba_load(dy + (size_t)row * P, P, ch * BA_CHUNK, vg);  // zeros past the end: no contribution
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < BA_V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (vx[i][e] - mu) * inv;
            const float du = vg[i][e] * act_grad(fmaf(xh, gam, bet), act);
            s1 += du, s2 += du * xh;
        }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        const int nt = B * chunks, tile = b * chunks + ch;
        part[(size_t)c * nt + tile] = s1;
        part[((size_t)C + c) * nt + tile] = s2;
    }
}

// per channel: dbias = sum du, dweight = sum du*xhat, coef = (mean(du), mean(du*xhat)) (zeros in eval mode)
__global__ __launch_bounds__(BA_T) void bn_act_bwd_finalize_kernel(const float* __restrict__ part, int nt, int C,
                                                                    double count, int training,
                                                                    float* __restrict__ dweight,
                                                                    float* __restrict__ dbias, float* __restrict__ coef) {
    __shared__ double dred[2][4];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int t = threadIdx.x; t < nt; t += BA_T) {
        s1 += (double)part[(size_t)c * nt + t];
        s2 += (double)part[((size_t)C + c) * nt + t];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) dred[0][threadIdx.x >> 6] = s1, dred[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = (dred[0][0] + dred[0][1]) + (dred[0][2] + dred[0][3]);
        s2 = (dred[1][0] + dred[1][1]) + (dred[1][2] + dred[1][3]);
        dbias[c] = (float)s1;
        dweight[c] = (float)s2;
        coef[c] = training ? (float)(s1 / count) : 0.f;
        coef[C + c] = training ? (float)(s2 / count) : 0.f;
    }
}

__global__ __launch_bounds__(BA_T) void bn_act_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ coef, int C, int P, int chunks,
                                                              int act, float* __restrict__ dx) {
    const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks, c = row % C;
    const float mu = mean[c], inv = invstd[c], gam = weight[c], bet = bias[c];
    const float m1 = coef[c], m2 = coef[C + c], gi = gam * inv;
    f32x4 vx[BA_V], vg[BA_V];
    ba_load(x + (size_t)row * P, P, ch * BA_CHUNK, vx);
    ba_load(dy + (size_t)row * P, P, ch * BA_CHUNK, vg);
#pragma unroll
    for (int i = 0; i < BA_V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (vx[i][e] - mu) * inv;
            const float du = vg[i][e] * act_grad(fmaf(xh, gam, bet), act);
            vg[i][e] = gi * (du - m1 - xh * m2);
        }
    ba_store(dx + (size_t)row * P, P, ch * BA_CHUNK, vg);
}

// ---- channel gate + activation (squeeze-excite tail, reference mobilenetv3.py:79-83 followed by :121/:141) ----
// y = act(x * gate[b,c]);  backward: du = dy * act'(x*gate), dx = du * gate, dgate[b,c] = sum_p du * x
__global__ __launch_bounds__(BA_T) void gate_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                             int P, int chunks, int act, float* __restrict__ y) {
    const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks;
    const float gt = gate[row];
    f32x4 v[BA_V];
    ba_load(x + (size_t)row * P, P, ch * BA_CHUNK, v);
#pragma unroll
    for (int i = 0; i < BA_V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = act_fwd(v[i][e] * gt, act);
    ba_store(y + (size_t)row * P, P, ch * BA_CHUNK, v);
}

__global__ __launch_bounds__(BA_T) void gate_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ gate, int P, int chunks, int act,
                                                             float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4];
    const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks;
    const float gt = gate[row];
    f32x4 vx[BA_V], vg[BA_V];
    ba_load(x + (size_t)row * P, P, ch * BA_CHUNK, vx);
    ba_load(dy + (size_t)row * P, P, ch * BA_CHUNK, vg);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BA_V; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float du = vg[i][e] * act_grad(vx[i][e] * gt, act);
            s = fmaf(du, vx[i][e], s);
            vg[i][e] = du * gt;
        }
    ba_store(dx + (size_t)row * P, P, ch * BA_CHUNK, vg);
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) part[(size_t)row * chunks + ch] = s;
}

__global__ void gate_act_dgate_kernel(const float* __restrict__ part, int rows, int chunks, float* __restrict__ dgate) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)row * chunks + c];
    dgate[row] = s;
}

static int ba_chunks(int P) { return ceil_div(P, BA_CHUNK); }

size_t bn_act_workspace(int B, int C, int P) {
    return align_up((size_t)2 * C * B * ba_chunks(P) * sizeof(float), 256) + align_up((size_t)2 * C * sizeof(float), 256);
}

hipError_t bn_stats_run(const float* x, float* running_mean, float* running_var, int B, int C, int P, int training,
                        float momentum, float eps, float* save_mean, float* save_invstd, void* ws, hipStream_t stream) {
    const int chunks = ba_chunks(P), grid = B * C * chunks;
    float* part = static_cast<float*>(ws);
    if (training)
        hipLaunchKernelGGL(bn_act_stats_kernel, dim3(grid), dim3(BA_T), 0, stream, x, part, B, C, P, chunks);
    hipLaunchKernelGGL(bn_act_finalize_kernel, dim3(C), dim3(BA_T), 0, stream, part, B, C, P, chunks, 0, 0, training, momentum,
                       eps, running_mean, running_var, save_mean, save_invstd);
    return hipGetLastError();
}

// the same forward with the statistics pass replaced by partials a producer already holds (K11's epilogue): finalize + apply only
hipError_t bn_act_fwd_part_run(const float* x, const float* conv_part, int H, int W, const float* weight, const float* bias,
                               float* running_mean, float* running_var, const float* residual, int B, int C, int act, int training,
                               float momentum, float eps, float* y, float* save_mean, float* save_invstd, hipStream_t stream) {
    const int P = H * W, chunks = ba_chunks(P), grid = B * C * chunks;
    hipLaunchKernelGGL(bn_act_finalize_kernel, dim3(C), dim3(BA_T), 0, stream, conv_part, B, C, P, chunks, H, W, training, momentum,
                       eps, running_mean, running_var, save_mean, save_invstd);
    hipLaunchKernelGGL(bn_act_apply_kernel, dim3(grid), dim3(BA_T), 0, stream, x, save_mean, save_invstd, weight, bias,
                       residual, C, P, chunks, act, y);
    return hipGetLastError();
}

hipError_t bn_act_fwd_run(const float* x, const float* weight, const float* bias, float* running_mean,
                          float* running_var, const float* residual, int B, int C, int P, int act, int training,
                          float momentum, float eps, float* y, float* save_mean, float* save_invstd, void* ws,
                          hipStream_t stream) {
    const int chunks = ba_chunks(P), grid = B * C * chunks;
    (void)bn_stats_run(x, running_mean, running_var, B, C, P, training, momentum, eps, save_mean, save_invstd, ws, stream);
    hipLaunchKernelGGL(bn_act_apply_kernel, dim3(grid), dim3(BA_T), 0, stream, x, save_mean, save_invstd, weight, bias,
                       residual, C, P, chunks, act, y);
    return hipGetLastError();
}

hipError_t bn_bwd_tail_run(const float* part, int nt, const float* dy, const float* x, const float* weight,
                           const float* bias, const float* save_mean, const float* save_invstd, int B, int C, int P,
                           int act, int training, float* dx, float* dweight, float* dbias, float* coef,
                           hipStream_t stream) {
    const int chunks = ba_chunks(P), grid = B * C * chunks;
    hipLaunchKernelGGL(bn_act_bwd_finalize_kernel, dim3(C), dim3(BA_T), 0, stream, part, nt, C, (double)B * (double)P,
                       training, dweight, dbias, coef);
    hipLaunchKernelGGL(bn_act_bwd_dx_kernel, dim3(grid), dim3(BA_T), 0, stream, dy, x, save_mean, save_invstd, weight,
                       bias, coef, C, P, chunks, act, dx);
    return hipGetLastError();
}

hipError_t bn_act_bwd_run(const float* dy, const float* x, const float* weight, const float* bias,
                          const float* save_mean, const float* save_invstd, int B, int C, int P, int act, int training,
                          float* dx, float* dweight, float* dbias, void* ws, hipStream_t stream) {
    const int chunks = ba_chunks(P), grid = B * C * chunks, nt = B * chunks;
    float* part = static_cast<float*>(ws);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)2 * C * nt * sizeof(float), 256));
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(grid), dim3(BA_T), 0, stream, dy, x, save_mean, save_invstd, weight,
                       bias, B, C, P, chunks, act, part);
    return bn_bwd_tail_run(part, nt, dy, x, weight, bias, save_mean, save_invstd, B, C, P, act, training, dx, dweight,
                           dbias, coef, stream);
}

size_t gate_act_workspace(int rows, int P) { return align_up((size_t)rows * ba_chunks(P) * sizeof(float), 256); }

hipError_t gate_act_fwd_run(const float* x, const float* gate, int rows, int P, int act, float* y, hipStream_t stream) {
    const int chunks = ba_chunks(P);
    hipLaunchKernelGGL(gate_act_fwd_kernel, dim3(rows * chunks), dim3(BA_T), 0, stream, x, gate, P, chunks, act, y);
    return hipGetLastError();
}

hipError_t gate_act_bwd_run(const float* dy, const float* x, const float* gate, int rows, int P, int act, float* dx,
                            float* dgate, void* ws, hipStream_t stream) {
    const int chunks = ba_chunks(P);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(gate_act_bwd_kernel, dim3(rows * chunks), dim3(BA_T), 0, stream, dy, x, gate, P, chunks, act, dx,
                       part);
    hipLaunchKernelGGL(gate_act_dgate_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, part, rows, chunks, dgate);
    return hipGetLastError();
}

}  // namespace cabinet
