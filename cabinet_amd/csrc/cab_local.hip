// K5 -- CAB local branch + block output, forward and backward.
//
// Replaces reference src/models/cab.py:175-184 (LocalAttention: three DWConv = depthwise 3x3 + BatchNorm +
// ReLU, sigmoid gate, x + x*mask) and cab.py:213-216 (gamma * global + local) -- 13 ATen launches forward and
// ~30 backward on (B,256,H/32,W/32) tensors -- with ONE kernel each way.
//
// MI355X-first observation: a depthwise convolution and a per-channel BatchNorm never mix channels, so the whole
// three-stage chain of ONE channel, over all B images, is an independent problem of B*H*W elements (8192 floats
// = 32 KB at config 3).  It fits the 160 KB LDS of a CU several times over, so one workgroup per channel runs
// the entire chain -- batch statistics included, as in-workgroup reductions -- with no grid-wide barrier and no
// intermediate tensor in HBM; C = 256 workgroups are exactly one per CU.  Backward recomputes the chain in LDS
// (nothing but x and the per-stage mean / invstd is saved) and back-propagates through it in the same kernel:
// LDS holds x, y1, y2 (stencil inputs of the weight gradients) and the current dz (input of the transposed
// stencil); everything element-wise lives in registers.
#include "cab_local.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int LOC_T = 512;     // threads per workgroup
constexpr int LOC_EPT = 16;    // elements per thread  ->  B*H*W <= 8192
constexpr int LOC_MAXN = LOC_T * LOC_EPT;

// sum over the workgroup, result in every thread (two barriers)
__device__ __forceinline__ float wg_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < LOC_T / 64; ++w) t += red[w];
    __syncthreads();
    return t;
}

// z[y][x] = sum_{ky,kx} w[ky][kx] * in[y+ky-1][x+kx-1]        (zero padding, one image plane)
__device__ __forceinline__ float stencil_fwd(const float* buf, int e, int y, int x, int H, int W, const float (&w)[9]) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if (xx < 0 || xx >= W) continue;
            acc += w[ky * 3 + kx] * buf[e + (ky - 1) * W + (kx - 1)];
        }
    }
    return acc;
}
// din[y][x] = sum_{ky,kx} w[ky][kx] * dz[y-ky+1][x-kx+1]
__device__ __forceinline__ float stencil_bwd(const float* buf, int e, int y, int x, int H, int W, const float (&w)[9]) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y - ky + 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x - kx + 1;
            if (xx < 0 || xx >= W) continue;
            acc += w[ky * 3 + kx] * buf[e - (ky - 1) * W - (kx - 1)];
        }
    }
    return acc;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(LOC_T) void cab_local_fwd_kernel(LocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.H * a.W, N = a.B * n, c = blockIdx.x, tid = threadIdx.x;
    float* bufA = smem;
    float* bufB = smem + N;
    float* red = smem + 2 * N;
    float xr[LOC_EPT], v[LOC_EPT];
    int yx[LOC_EPT];
#pragma unroll
    for (int k = 0; k < LOC_EPT; ++k) {
        const int e = k * LOC_T + tid;
        xr[k] = 0.f;
        yx[k] = 0;
        if (e < N) {
            const int b = e / n, p = e - b * n, y = p / a.W;
            yx[k] = (y << 16) | (p - y * a.W);
            xr[k] = a.x[((size_t)b * a.C + c) * n + p];
            bufA[e] = xr[k];
        }
    }
    __syncthreads();
    float* in = bufA;
    float* ob = bufB;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float w[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) w[j] = a.st[s].w[c * 9 + j];
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const int e = k * LOC_T + tid;
            v[k] = (e < N) ? stencil_fwd(in, e, yx[k] >> 16, yx[k] & 0xffff, a.H, a.W, w) : 0.f;
        }
        float mean, invstd;
        if (a.training) {  // batch statistics of this channel, two-pass
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) s1 += v[k];
            mean = wg_sum(s1, red) / (float)N;
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const int e = k * LOC_T + tid;
                if (e < N) s2 += (v[k] - mean) * (v[k] - mean);
            }
            const float var = wg_sum(s2, red) / (float)N;
            invstd = 1.0f / sqrtf(var + a.eps);
            if (tid == 0) {
                const float unbiased = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
                a.st[s].run_mean[c] = (1.f - a.momentum) * a.st[s].run_mean[c] + a.momentum * mean;
                a.st[s].run_var[c] = (1.f - a.momentum) * a.st[s].run_var[c] + a.momentum * unbiased;
            }
        } else {
            mean = a.st[s].run_mean[c];
            invstd = 1.0f / sqrtf(a.st[s].run_var[c] + a.eps);
        }
        if (tid == 0) {
            a.save_mean[s * a.C + c] = mean;
            a.save_invstd[s * a.C + c] = invstd;
        }
        const float sc = a.st[s].bn_w[c] * invstd, sh = a.st[s].bn_b[c] - mean * sc;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) v[k] = fmaxf(fmaf(v[k], sc, sh), 0.f);
        if (s < 2) {
            __syncthreads();  // every stencil read of `in` is done (wg_sum syncs only in training mode)
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const int e = k * LOC_T + tid;
                if (e < N) ob[e] = v[k];
            }
            __syncthreads();
            float* t = in;
            in = ob;
            ob = t;
        }
    }
    const float gam = a.glob ? a.gamma[0] : 0.f;
#pragma unroll
    for (int k = 0; k < LOC_EPT; ++k) {
        const int e = k * LOC_T + tid;
        if (e < N) {
            const int b = e / n, p = e - b * n;
            const size_t gi = ((size_t)b * a.C + c) * n + p;
            float o = xr[k] * (1.f + sigmoidf(v[k]));
            if (a.glob) o = fmaf(gam, a.glob[gi], o);
            a.out[gi] = o;
        }
    }
}

// ----------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(LOC_T) void cab_local_bwd_kernel(LocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.H * a.W, N = a.B * n, c = blockIdx.x, tid = threadIdx.x;
    float* bufX = smem;            // x            (stencil input of dW1)
    float* bufY1 = smem + N;       // y1           (stencil input of dW2 and of z2)
    float* bufY2 = smem + 2 * N;   // y2           (stencil input of dW3 and of z3)
    float* bufG = smem + 3 * N;    // current dz   (input of the transposed stencil)
    float* red = smem + 4 * N;     // [LOC_T/64] + [9][LOC_T/64]
    float* red9 = red + LOC_T / 64;

    float xr[LOC_EPT], xh[3][LOC_EPT], yv[LOC_EPT], g[LOC_EPT];
    int yx[LOC_EPT];
    float w[3][9], mean[3], invstd[3], bw[3], bb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int j = 0; j < 9; ++j) w[s][j] = a.st[s].w[c * 9 + j];
        mean[s] = a.save_mean[s * a.C + c];
        invstd[s] = a.save_invstd[s * a.C + c];
        bw[s] = a.st[s].bn_w[c];
        bb[s] = a.st[s].bn_b[c];
    }
#pragma unroll
    for (int k = 0; k < LOC_EPT; ++k) {
        const int e = k * LOC_T + tid;
        xr[k] = 0.f, yx[k] = 0, g[k] = 0.f;
        if (e < N) {
            const int b = e / n, p = e - b * n, y = p / a.W;
            yx[k] = (y << 16) | (p - y * a.W);
            const size_t gi = ((size_t)b * a.C + c) * n + p;
            xr[k] = a.x[gi];
            g[k] = a.dout[gi];
            bufX[e] = xr[k];
        }
    }
    __syncthreads();
    // ---- recompute the forward chain: xhat_s in registers, y1 / y2 in LDS, y3 in registers ----
    {
        const float* in = bufX;
        float* outs[2] = {bufY1, bufY2};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const int e = k * LOC_T + tid;
                const float z = (e < N) ? stencil_fwd(in, e, yx[k] >> 16, yx[k] & 0xffff, a.H, a.W, w[s]) : 0.f;
                xh[s][k] = (z - mean[s]) * invstd[s];
                yv[k] = fmaxf(fmaf(xh[s][k], bw[s], bb[s]), 0.f);
            }
            if (s < 2) {
#pragma unroll
                for (int k = 0; k < LOC_EPT; ++k) {
                    const int e = k * LOC_T + tid;
                    if (e < N) outs[s][e] = yv[k];
                }
                __syncthreads();
                in = outs[s];
            }
        }
    }
    // ---- block output: out = gamma*glob + x*(1 + sigmoid(y3)) ----
    float dxd[LOC_EPT], dy[LOC_EPT];
    {
        const float gam = a.glob ? a.gamma[0] : 0.f;
        float dg = 0.f;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const int e = k * LOC_T + tid;
            const float sg = sigmoidf(yv[k]);
            dxd[k] = g[k] * (1.f + sg);
            dy[k] = (yv[k] > 0.f) ? g[k] * xr[k] * sg * (1.f - sg) : 0.f;  // through the ReLU of stage 3
            if (a.glob && e < N) {
                const int b = e / n, p = e - b * n;
                const size_t gi = ((size_t)b * a.C + c) * n + p;
                dg += g[k] * a.glob[gi];
                a.dglob[gi] = gam * g[k];
            }
        }
        if (a.glob) {
            dg = wg_sum(dg, red);
            if (tid == 0) a.dgamma_part[c] = dg;
        }
    }
    // ---- stages 3, 2, 1 ----
    const float inv_n = 1.f / (float)N;
#pragma unroll
    for (int s = 2; s >= 0; --s) {
        // dy[] holds dL/dy_s already masked by the ReLU of stage s
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) s1 += dy[k], s2 += dy[k] * xh[s][k];
        s1 = wg_sum(s1, red);
        s2 = wg_sum(s2, red);
        if (tid == 0) {
            a.st[s].dbn_b[c] = s1;
            a.st[s].dbn_w[c] = s2;
        }
        const float gi_ = bw[s] * invstd[s];
        const float m1 = a.training ? s1 * inv_n : 0.f, m2 = a.training ? s2 * inv_n : 0.f;
        float dz[LOC_EPT];
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const int e = k * LOC_T + tid;
            dz[k] = (e < N) ? gi_ * (dy[k] - m1 - xh[s][k] * m2) : 0.f;
            if (e < N) bufG[e] = dz[k];
        }
        __syncthreads();
        // weight gradient: dW[ky][kx] = sum_e dz[e] * in_s[e shifted], in_s = x, y1, y2
        const float* ins = (s == 0) ? bufX : (s == 1) ? bufY1 : bufY2;
        float pw[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) pw[j] = 0.f;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const int e = k * LOC_T + tid;
            if (e < N) {
                const int y = yx[k] >> 16, x = yx[k] & 0xffff;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= a.H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= a.W) continue;
                        pw[ky * 3 + kx] += dz[k] * ins[e + (ky - 1) * a.W + (kx - 1)];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float t = wave_sum(pw[j]);
            if ((tid & 63) == 0) red9[j * (LOC_T / 64) + (tid >> 6)] = t;
        }
        __syncthreads();
        if (tid < 9) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < LOC_T / 64; ++wv) t += red9[tid * (LOC_T / 64) + wv];
            a.st[s].dw[c * 9 + tid] = t;
        }
        // input gradient through the transposed stencil, then through the ReLU of the previous stage
        float din[LOC_EPT];
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const int e = k * LOC_T + tid;
            din[k] = (e < N) ? stencil_bwd(bufG, e, yx[k] >> 16, yx[k] & 0xffff, a.H, a.W, w[s]) : 0.f;
        }
        __syncthreads();  // bufG and red9 are free again
        if (s > 0) {
            const float* yprev = (s == 1) ? bufY1 : bufY2;
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const int e = k * LOC_T + tid;
                dy[k] = (e < N && yprev[e] > 0.f) ? din[k] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const int e = k * LOC_T + tid;
                if (e < N) {
                    const int b = e / n, p = e - b * n;
                    a.dx[((size_t)b * a.C + c) * n + p] = dxd[k] + din[k];
                }
            }
        }
    }
}

bool local_shape_supported(int B, int H, int W) { return (long long)B * H * W <= LOC_MAXN && H < 65536 && W < 65536; }

hipError_t cab_local_fwd_run(const LocalArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)2 * a.B * a.H * a.W + 64) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cab_local_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipLaunchKernelGGL(cab_local_fwd_kernel, dim3(a.C), dim3(LOC_T), lds, stream, a);
    return hipGetLastError();
}

hipError_t cab_local_bwd_run(const LocalArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)4 * a.B * a.H * a.W + 10 * (LOC_T / 64) + 16) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cab_local_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipLaunchKernelGGL(cab_local_bwd_kernel, dim3(a.C), dim3(LOC_T), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cabinet
