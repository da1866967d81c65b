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

constexpr int LOC_T = 1024;    // threads per workgroup: 16 waves = 4 per SIMD; the kernel is latency-bound (one workgroup per
                               // CU, a dozen barriers per stage), so waves per SIMD are what hides LDS / barrier latency
constexpr int LOC_EPT = 8;     // elements per thread  ->  B*H*W <= 8192
constexpr int LOC_MAXN = LOC_T * LOC_EPT;

// sum over the workgroup, result in every thread (two barriers).  The per-thread partial (<= 8 terms) is fp32, everything
// across threads is DOUBLE: these are the BatchNorm sums, and ATen's CPU kernels accumulate them in double too -- the
// depthwise-weight gradients downstream are cancelling sums that amplify a 1e-7 error of the statistics to 1e-3.
__device__ __forceinline__ double wg_sum(float v, double* red) {
    double d = (double)v;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < LOC_T / 64; ++w) t += red[w];
    __syncthreads();
    return t;
}

// LDS planes are stored ZERO-PADDED, (H+2) x (W+2) per image, so the stencils need no border tests: element (b,y,x) sits at
// pe = b*(H+2)*(W+2) + (y+1)*(W+2) + (x+1) and its neighbours at pe +- (W+2) +- 1 (the pad cells are written once and stay 0).
// z[y][x] = sum_{ky,kx} w[ky][kx] * in[y+ky-1][x+kx-1]
__device__ __forceinline__ float stencil_fwd(const float* buf, int pe, int WP, const float (&w)[9]) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc += w[ky * 3 + kx] * buf[pe + (ky - 1) * WP + (kx - 1)];
    return acc;
}
// din[y][x] = sum_{ky,kx} w[ky][kx] * dz[y-ky+1][x-kx+1]
__device__ __forceinline__ float stencil_bwd(const float* buf, int pe, int WP, const float (&w)[9]) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc += w[ky * 3 + kx] * buf[pe - (ky - 1) * WP - (kx - 1)];
    return acc;
}

// two sums at the price of one (same two barriers)
__device__ __forceinline__ void wg_sum2d(double a, double b, double* red, double& A, double& B) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64), b += __shfl_xor(b, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a, red[LOC_T / 64 + (threadIdx.x >> 6)] = b;
    __syncthreads();
    A = 0.0, B = 0.0;
#pragma unroll
    for (int w = 0; w < LOC_T / 64; ++w) A += red[w], B += red[LOC_T / 64 + w];
    __syncthreads();
}

// hardware exp2 and reciprocal (1 ulp each): the libm expansion + IEEE division were ~25 instructions per value
__device__ __forceinline__ float sigmoidf(float v) { return __builtin_amdgcn_rcpf(1.f + fast_exp2(-v * LOG2E_F)); }

// ------------------------------------------------------------------------------------------ forward
template <int NPL>  // see cab_local_bwd_kernel
__global__ __launch_bounds__(LOC_T) void cab_local_fwd_kernel(LocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.H * a.W, N = a.B * n, c = blockIdx.x, tid = threadIdx.x;
    const int WP = a.W + 2, PP = (a.H + 2) * WP, NP = a.B * PP;
    float* bufA = smem;
    float* bufB = smem + NP;
    double* red = reinterpret_cast<double*>(smem + ((2 * NP + 1) & ~1));
    float xr[LOC_EPT], v[LOC_EPT];
    constexpr bool ONEPLANE = NPL > 0;
    int pe[ONEPLANE ? 1 : LOC_EPT], pe0[ONEPLANE ? NPL : 1];
    if (ONEPLANE) {
#pragma unroll
        for (int j = 0; j < (ONEPLANE ? NPL : 1); ++j) {
            const int p = j * LOC_T + tid, y = p / a.W;
            pe0[j] = (y + 1) * WP + (p - y * a.W) + 1;
        }
    }
    auto PE = [&](int kk) { return ONEPLANE ? (kk / (ONEPLANE ? NPL : 1)) * PP + pe0[kk % (ONEPLANE ? NPL : 1)] : pe[kk]; };
    for (int i = tid; i < 2 * NP; i += LOC_T) smem[i] = 0.f;  // pad cells stay zero for the whole kernel
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LOC_EPT; ++k) {
        const int e = k * LOC_T + tid;
        xr[k] = 0.f;
        if (!ONEPLANE) pe[k] = WP + 1;
        if (e < N) {
            const int b = ONEPLANE ? k / (ONEPLANE ? NPL : 1) : e / n, p = ONEPLANE ? (k % (ONEPLANE ? NPL : 1)) * LOC_T + tid : e - b * n;
            if (!ONEPLANE) {
                const int y = p / a.W;
                pe[k] = b * PP + (y + 1) * WP + (p - y * a.W) + 1;
            }
            xr[k] = a.x[((size_t)b * a.C + c) * n + p];
            bufA[PE(k)] = xr[k];
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
        for (int k = 0; k < LOC_EPT; ++k) v[k] = (k * LOC_T + tid < N) ? stencil_fwd(in, PE(k), WP, w) : 0.f;
        float mean, invstd;
        if (a.training) {  // batch statistics of this channel: one pass, sums in double (see wg_sum)
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) s1 += (double)v[k], s2 = fma((double)v[k], (double)v[k], s2);
            double S1, S2;
            wg_sum2d(s1, s2, red, S1, S2);
            const double mu = S1 / (double)N;
            double var = S2 / (double)N - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            invstd = (float)(1.0 / sqrt(var + (double)a.eps));
            if (tid == 0) {
                const double unbiased = N > 1 ? var * ((double)N / (double)(N - 1)) : var;
                a.st[s].run_mean[c] = (float)((1.0 - (double)a.momentum) * (double)a.st[s].run_mean[c] + (double)a.momentum * mu);
                a.st[s].run_var[c] = (float)((1.0 - (double)a.momentum) * (double)a.st[s].run_var[c] + (double)a.momentum * unbiased);
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
            __syncthreads();  // every stencil read of `in` is done (wg_sum2d syncs only in training mode)
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k)
                if (k * LOC_T + tid < N) ob[PE(k)] = v[k];
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
            const int b = ONEPLANE ? k / (ONEPLANE ? NPL : 1) : e / n, p = ONEPLANE ? (k % (ONEPLANE ? NPL : 1)) * LOC_T + tid : e - b * n;
            const size_t gi = ((size_t)b * a.C + c) * n + p;
            float o = xr[k] * (1.f + sigmoidf(v[k]));
            if (a.glob) o = fmaf(gam, a.glob[gi], o);
            a.out[gi] = o;
        }
    }
}

// ----------------------------------------------------------------------------------------- backward
// NPL > 0: H * W == NPL * LOC_T (NPL = 1: the model's 32 x 32 map; 2: the 64 x 32 map of BASELINE config 5): element k of a
// thread is pixel (k % NPL) * LOC_T + tid of image k / NPL, so the padded LDS index of every element is one of NPL per-thread
// constants plus (k / NPL) * PP -- the eight-entry index array leaves the register file, and so does every per-element
// division.  NPL == 0 is the general shape.
template <int NPL>
__global__ __launch_bounds__(LOC_T) void cab_local_bwd_kernel(LocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.H * a.W, N = a.B * n, c = blockIdx.x, tid = threadIdx.x;
    const int WP = a.W + 2, PP = (a.H + 2) * WP, NP = a.B * PP;
    float* bufX = smem;             // x            (stencil input of dW1)
    float* bufY1 = smem + NP;       // y1           (stencil input of dW2 and of z2)
    float* bufY2 = smem + 2 * NP;   // y2           (stencil input of dW3 and of z3)
    float* bufG = smem + 3 * NP;    // current dz   (input of the transposed stencil)
    double* red = reinterpret_cast<double*>(smem + ((4 * NP + 1) & ~1));  // [2][LOC_T/64] doubles, then [9][LOC_T/64] floats
    float* red9 = reinterpret_cast<float*>(red + 2 * (LOC_T / 64));

    float xh[3][LOC_EPT], yv[LOC_EPT], g[LOC_EPT];  // x itself stays in bufX (LDS) only: 8 registers less per lane
    constexpr bool ONEPLANE = NPL > 0;
    int pe[ONEPLANE ? 1 : LOC_EPT], pe0[ONEPLANE ? NPL : 1];
    if (ONEPLANE) {
#pragma unroll
        for (int j = 0; j < (ONEPLANE ? NPL : 1); ++j) {
            const int p = j * LOC_T + tid, y = p / a.W;
            pe0[j] = (y + 1) * WP + (p - y * a.W) + 1;
        }
    }
    auto PE = [&](int kk) { return ONEPLANE ? (kk / (ONEPLANE ? NPL : 1)) * PP + pe0[kk % (ONEPLANE ? NPL : 1)] : pe[kk]; };
    for (int i = tid; i < 4 * NP; i += LOC_T) smem[i] = 0.f;  // pad cells stay zero for the whole kernel
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LOC_EPT; ++k) {
        const int e = k * LOC_T + tid;
        if (!ONEPLANE) pe[k] = WP + 1;
        g[k] = 0.f;
        if (e < N) {
            const int b = ONEPLANE ? k / (ONEPLANE ? NPL : 1) : e / n, p = ONEPLANE ? (k % (ONEPLANE ? NPL : 1)) * LOC_T + tid : e - b * n, y = p / a.W;
            if (!ONEPLANE) pe[k] = b * PP + (y + 1) * WP + (p - y * a.W) + 1;
            const size_t gi = ((size_t)b * a.C + c) * n + p;
            g[k] = a.dout[gi];
            bufX[PE(k)] = a.x[gi];
        }
    }
    __syncthreads();
    // ---- recompute the forward chain: xhat_s in registers, y1 / y2 in LDS, y3 in registers ----
    {
        const float* in = bufX;
        float* outs[2] = {bufY1, bufY2};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            // per-stage constants are (re)loaded where they are used: they are wave-uniform (SGPRs), and holding all three
            // stages' 27 weights + 12 BatchNorm constants for the whole kernel ran the scalar file out and spilled
            float w9[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) w9[j] = a.st[s].w[c * 9 + j];
            const float mean_s = a.save_mean[s * a.C + c], invstd_s = a.save_invstd[s * a.C + c];
            const float bw_s = a.st[s].bn_w[c], bb_s = a.st[s].bn_b[c];
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const float z = (k * LOC_T + tid < N) ? stencil_fwd(in, PE(k), WP, w9) : 0.f;
                xh[s][k] = (z - mean_s) * invstd_s;
                yv[k] = fmaxf(fmaf(xh[s][k], bw_s, bb_s), 0.f);
                if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // 36 LDS reads in flight, not 72 (register budget)
            }
            if (s < 2) {
#pragma unroll
                for (int k = 0; k < LOC_EPT; ++k)
                    if (k * LOC_T + tid < N) outs[s][PE(k)] = yv[k];
                __syncthreads();
                in = outs[s];
            }
        }
    }
    // ---- block output: out = gamma*glob + x*(1 + sigmoid(y3)) ----
    // the direct term of dx, g (1 + sigmoid(y3)), goes to a.dx now and the stencil term is added to it at the very end by the
    // same thread: eight more registers that do not have to live through the three backward stages
    float dy[LOC_EPT];
    {
        const float gam = a.glob ? a.gamma[0] : 0.f;
        float dg = 0.f;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const int e = k * LOC_T + tid;
            const float sg = sigmoidf(yv[k]);
            dy[k] = (yv[k] > 0.f) ? g[k] * bufX[PE(k)] * sg * (1.f - sg) : 0.f;  // through the ReLU of stage 3
            if (e < N) {
                const int b = ONEPLANE ? k / (ONEPLANE ? NPL : 1) : e / n, p = ONEPLANE ? (k % (ONEPLANE ? NPL : 1)) * LOC_T + tid : e - b * n;
                const size_t gi = ((size_t)b * a.C + c) * n + p;
                a.dx[gi] = g[k] * (1.f + sg);
                if (a.glob) {
                    dg += g[k] * a.glob[gi];
                    a.dglob[gi] = gam * g[k];
                }
            }
        }
        if (a.glob) {
            const double dgs = wg_sum(dg, red);
            if (tid == 0) a.dgamma_part[c] = (float)dgs;
        }
    }
    // ---- stages 3, 2, 1 ----
    const double inv_n = 1.0 / (double)N;
#pragma unroll
    for (int s = 2; s >= 0; --s) {
        // dy[] holds dL/dy_s already masked by the ReLU of stage s
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) s1 += (double)dy[k], s2 = fma((double)dy[k], (double)xh[s][k], s2);
        double S1, S2;
        wg_sum2d(s1, s2, red, S1, S2);
        if (tid == 0) {
            a.st[s].dbn_b[c] = (float)S1;
            a.st[s].dbn_w[c] = (float)S2;
        }
        const float gi_ = a.st[s].bn_w[c] * a.save_invstd[s * a.C + c];
        const float m1 = a.training ? (float)(S1 * inv_n) : 0.f, m2 = a.training ? (float)(S2 * inv_n) : 0.f;
        float dz[LOC_EPT];
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            const bool ok = k * LOC_T + tid < N;
            dz[k] = ok ? gi_ * (dy[k] - m1 - xh[s][k] * m2) : 0.f;
            if (ok) bufG[PE(k)] = dz[k];
        }
        __syncthreads();
        // weight gradient: dW[ky][kx] = sum_e dz[e] * in_s[e shifted], in_s = x, y1, y2 (zero-padded: no border tests)
        const float* ins = (s == 0) ? bufX : (s == 1) ? bufY1 : bufY2;
        float pw[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) pw[j] = 0.f;
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) pw[ky * 3 + kx] += dz[k] * ins[PE(k) + (ky - 1) * WP + (kx - 1)];
            if (k & 1) __builtin_amdgcn_sched_barrier(0);  // two elements' 18 LDS reads in flight, not all 72 (register budget)
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float t = wave_sum(pw[j]);
            if ((tid & 63) == 0) red9[j * (LOC_T / 64) + (tid >> 6)] = t;
        }
        __syncthreads();
        if (tid < 9) {
            double t = 0.0;
#pragma unroll
            for (int wv = 0; wv < LOC_T / 64; ++wv) t += (double)red9[tid * (LOC_T / 64) + wv];
            a.st[s].dw[c * 9 + tid] = (float)t;
        }
        // input gradient through the transposed stencil, then through the ReLU of the previous stage
        float din[LOC_EPT], w9[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) w9[j] = a.st[s].w[c * 9 + j];
#pragma unroll
        for (int k = 0; k < LOC_EPT; ++k) {
            din[k] = (k * LOC_T + tid < N) ? stencil_bwd(bufG, PE(k), WP, w9) : 0.f;
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // bufG and red9 are free again
        if (s > 0) {
            const float* yprev = (s == 1) ? bufY1 : bufY2;
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) dy[k] = (k * LOC_T + tid < N && yprev[PE(k)] > 0.f) ? din[k] : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < LOC_EPT; ++k) {
                const int e = k * LOC_T + tid;
                if (e < N) {
                    const int b = ONEPLANE ? k / (ONEPLANE ? NPL : 1) : e / n, p = ONEPLANE ? (k % (ONEPLANE ? NPL : 1)) * LOC_T + tid : e - b * n;
                    a.dx[((size_t)b * a.C + c) * n + p] += din[k];
                }
            }
        }
    }
}

static size_t local_lds_fwd(int B, int H, int W) { return ((size_t)2 * B * (H + 2) * (W + 2) + 2 + 4 * (LOC_T / 64)) * sizeof(float); }
static size_t local_lds_bwd(int B, int H, int W) {
    return ((size_t)4 * B * (H + 2) * (W + 2) + 2 + 4 * (LOC_T / 64) + 9 * (LOC_T / 64) + 16) * sizeof(float);
}

bool local_shape_supported(int B, int H, int W) {
    return (long long)B * H * W <= LOC_MAXN && H < 65536 && W < 65536 && local_lds_bwd(B, H, W) <= 160 * 1024;
}

hipError_t cab_local_fwd_run(const LocalArgs& a, hipStream_t stream) {
    const size_t lds = local_lds_fwd(a.B, a.H, a.W);
    // the attribute is per device and a smaller later value would shrink it: always ask for the kernel family's maximum
    static lds_attr_mask attr_mask[3] = {{0}, {0}, {0}};
    const void* fns[3] = {reinterpret_cast<const void*>(cab_local_fwd_kernel<0>), reinterpret_cast<const void*>(cab_local_fwd_kernel<1>),
                          reinterpret_cast<const void*>(cab_local_fwd_kernel<2>)};
    for (int i = 0; i < 3; ++i)
        if (hipError_t e = ensure_dynamic_lds(fns[i], 160 * 1024, attr_mask[i]); e != hipSuccess) return e;
    const int n = a.H * a.W;
    auto kernel = n == LOC_T ? cab_local_fwd_kernel<1> : n == 2 * LOC_T ? cab_local_fwd_kernel<2> : cab_local_fwd_kernel<0>;
    hipLaunchKernelGGL(kernel, dim3(a.C), dim3(LOC_T), lds, stream, a);
    return hipGetLastError();
}

hipError_t cab_local_bwd_run(const LocalArgs& a, hipStream_t stream) {
    const size_t lds = local_lds_bwd(a.B, a.H, a.W);
    // the attribute is per device and a smaller later value would shrink it: always ask for the kernel family's maximum
    static lds_attr_mask attr_mask[3] = {{0}, {0}, {0}};
    const void* fns[3] = {reinterpret_cast<const void*>(cab_local_bwd_kernel<0>), reinterpret_cast<const void*>(cab_local_bwd_kernel<1>),
                          reinterpret_cast<const void*>(cab_local_bwd_kernel<2>)};
    for (int i = 0; i < 3; ++i)
        if (hipError_t e = ensure_dynamic_lds(fns[i], 160 * 1024, attr_mask[i]); e != hipSuccess) return e;
    const int n = a.H * a.W;
    auto kernel = n == LOC_T ? cab_local_bwd_kernel<1> : n == 2 * LOC_T ? cab_local_bwd_kernel<2> : cab_local_bwd_kernel<0>;
    hipLaunchKernelGGL(kernel, dim3(a.C), dim3(LOC_T), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cabinet
