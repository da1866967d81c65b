// K12 (round 5) -- BatchNorm2d -> ReLU -> 1x1 classifier convolution as ONE streaming operator, forward and backward.
//
// Reference spans: src/models/cabinet.py:156-172 (CABiNetOutput: `conv_out(relu(bn(conv(x))))`) behind
// K11's 3x3, and the tail of the fusion head, cabinet.py:90-92 (`b4(b3(b2(.)))`, row f2).  Until now: K7 (BatchNorm + ReLU apply,
// reads z, writes the activation a) + a 256 -> n_classes 1x1 convolution on MIOpen / hipBLASLt (reads a) forward, and
// dgrad (writes da) + NHWC transposes + wgrad (reads a) + K7's two backward passes (read da and z twice, write dz) backward --
// at BASELINE config 3 the activation is 8 x 256 x 128 x 128 = 134 MB, touched 9 times: 390 us of the step
// (tools/cls_tail_probe.py).  The classifier has K = 8 (UAVid) or 19 (Cityscapes) outputs: per position it is K dot products of
// length C, nothing for a matrix pipe -- this is HBM-bound streaming work, and the plan is the minimum number of passes a
// training-mode BatchNorm allows with the activation NEVER materialised:
//   fwd  : table  (BatchNorm finalize from K11's epilogue partials + per-channel row [w(:,c) | mean invstd gamma beta])
//          main   (read z once; a = relu(bn(z)) in registers; y[k] += w[k][c] a; write the K logits)        134 MB -> 4 MB
//   bwd  : reduce (read z, dy; da = sum_k w[k][c] dy[k] on the fly; per channel  sum du, sum du xhat, dw[k][c] = sum dy[k] a)
//          final  (ordered sums of the partials: dgamma, dbeta, batch means, dw, dbias)
//          dx     (read z, dy; write dz = gamma invstd (du - mean(du) - xhat mean(du xhat)))
// da, the 134 MB gradient of the activation, never exists either.  Per-channel scalars (the K weights, mean, invstd, gamma, beta)
// sit in one table row that a wave fetches with scalar loads (the channel index is wave-uniform): the vector unit only sees
// z, dy and the FMAs.  All partial sums are combined in a fixed order (no atomics): bit-reproducible.
#include "bn_finalize.hpp"
#include "common.hpp"

namespace cabinet {

// bn_act.hip
hipError_t bn_stats_run(const float* x, float* running_mean, float* running_var, int B, int C, int P, int training, float momentum,
                        float eps, float* save_mean, float* save_invstd, void* ws, hipStream_t stream);
size_t bn_act_workspace(int B, int C, int P);

constexpr int BC_X = 8;          // table row = KT weights + [mean, invstd, gamma, beta, gamma * invstd, 0, 0, 0]
#ifndef BC_TPW_VALUE
#define BC_TPW_VALUE 2048
#endif
constexpr int BC_TPW = BC_TPW_VALUE;   // positions one wave of the reduce kernel sweeps (64 lanes x 4 per step)

static inline int bc_kt(int K) { return K <= 8 ? 8 : (K <= 20 ? 20 : 32); }

// ---- table: one workgroup per channel -- finalize (training: Chan merge of the partials; eval: running buffers), then the row ----
// mode 0: statistics are already in save_mean / save_invstd (bn_stats_run ran in front);  1: finalize here from `part`
__global__ __launch_bounds__(BA_T) void bn_cls_table_kernel(const float* __restrict__ part, int mode, int B, int C, int P, int conv_h,
                                                             int conv_w, int training, float momentum, float eps,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                             const float* __restrict__ w_cls, int K, int KT, float* __restrict__ tab) {
    const int c = blockIdx.x;
    if (mode == 1) bn_finalize_channel(part, c, B, C, P, 0, conv_h, conv_w, training, momentum, eps, running_mean, running_var, save_mean, save_invstd);
    float* row = tab + (size_t)c * (KT + BC_X);
    if ((int)threadIdx.x < KT) row[threadIdx.x] = (int)threadIdx.x < K ? w_cls[(size_t)threadIdx.x * C + c] : 0.f;
    if (threadIdx.x == 0) {   // the thread that wrote save_mean / save_invstd in mode 1
        const float mu = save_mean[c], inv = save_invstd[c], gam = bn_w[c];
        row[KT + 0] = mu, row[KT + 1] = inv, row[KT + 2] = gam, row[KT + 3] = bn_b[c], row[KT + 4] = gam * inv;
        row[KT + 5] = 0.f, row[KT + 6] = 0.f, row[KT + 7] = 0.f;
    }
}

// ---- forward: workgroup = 64 positions of one image, wave w = channel quarter [w C/4, (w+1) C/4); the four partial logit vectors
//      meet in LDS.  A lane is one position: z rows are read as coalesced 256-byte pieces, 16 channels in flight per lane. ----
template <int KT>
__global__ __launch_bounds__(256) void bn_cls_fwd_kernel(const float* __restrict__ z, const float* __restrict__ tab,
                                                          const float* __restrict__ bias, int C, int P, int K, int tiles,
                                                          float* __restrict__ y) {
    constexpr int TS = KT + BC_X, U = 16;
    __shared__ float red[4][KT][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int p = tile * 64 + lane, pc = min(p, P - 1);
    const int cq = C >> 2, c0 = wave * cq;
    const float* zp = z + ((size_t)b * C + c0) * P + pc;
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    for (int cc = 0; cc < cq; cc += U) {
        float zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zv[u] = zp[(size_t)(cc + u) * P];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* row = tab + (size_t)(c0 + cc + u) * TS;   // wave-uniform: scalar loads
            const float a = fmaxf(fmaf((zv[u] - row[KT]) * row[KT + 1], row[KT + 2], row[KT + 3]), 0.f);
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[k] = fmaf(row[k], a, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) red[wave][k][lane] = acc[k];
    __syncthreads();
    for (int k = wave; k < K; k += 4) {
        const float v = ((red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane])) + (bias ? bias[k] : 0.f);
        if (p < P) y[((size_t)b * K + k) * P + p] = v;
    }
}

// sum over the 64 lanes, result in lane 63 (DPP: quad swaps, half-row / row mirrors, row broadcasts; fixed order)
__device__ __forceinline__ float wave_reduce63(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
    return x;
}

// ---- backward, pass 1: a wave owns CBR channels and sweeps BC_TPW positions of one image, four at a time per lane; the K
//      gradient rows of a position quad are loaded once and serve all CBR channels.  Per channel: sum du, sum du xhat, dw[k].
//      part[(c * ntiles + tile) * (KT + 2) + v];  the wave of channel block 0 also sums dy per class: bpart[tile * KT + k] ----
template <int KT, int CBR>
__global__ __launch_bounds__(256) void bn_cls_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                             const float* __restrict__ tab, int C, int P, int K, int tiles_img,
                                                             float* __restrict__ part, float* __restrict__ bpart) {
    constexpr int TS = KT + BC_X, NV = KT + 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = blockIdx.x, b = tile / tiles_img, t0 = (tile - b * tiles_img) * BC_TPW;
    const int c0 = (blockIdx.y * 4 + wave) * CBR;
    if (c0 >= C) return;
    const bool first = c0 == 0;
    float acc[CBR][NV], bs[KT];
#pragma unroll
    for (int j = 0; j < CBR; ++j)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[j][v] = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) bs[k] = 0.f;
    const float* dyb = dy + (size_t)b * K * P;
    const float* zb = z + ((size_t)b * C + c0) * P;
    for (int i = 0; i < BC_TPW / 256; ++i) {
        const int pq = t0 + 4 * (lane + 64 * i);
        if (t0 + 256 * i >= P) break;                     // wave-uniform: the tile ends here
        const bool ok = pq < P;                           // P % 4 == 0: a quad is inside or outside as a whole
        const int pl = ok ? pq : 0;
        f32x4 dq[KT], zq[CBR];
#pragma unroll
        for (int j = 0; j < CBR; ++j) zq[j] = *reinterpret_cast<const f32x4*>(zb + (size_t)j * P + pl);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            dq[k] = (k < K) ? *reinterpret_cast<const f32x4*>(dyb + (size_t)(k < K ? k : 0) * P + pl) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (!ok) dq[k] = f32x4{0.f, 0.f, 0.f, 0.f};   // zero gradient: nothing below contributes
        }
        if (first) {
#pragma unroll
            for (int k = 0; k < KT; ++k) bs[k] += (dq[k][0] + dq[k][1]) + (dq[k][2] + dq[k][3]);
        }
#pragma unroll
        for (int j = 0; j < CBR; ++j) {
            const float* row = tab + (size_t)(c0 + j) * TS;   // wave-uniform: scalar loads
            const float mu = row[KT], inv = row[KT + 1], gam = row[KT + 2], bet = row[KT + 3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (zq[j][e] - mu) * inv, pre = fmaf(xh, gam, bet), a = fmaxf(pre, 0.f);
                float da = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k) da = fmaf(row[k], dq[k][e], da);
                const float du = pre > 0.f ? da : 0.f;
                acc[j][KT] += du;
                acc[j][KT + 1] = fmaf(du, xh, acc[j][KT + 1]);
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[j][k] = fmaf(dq[k][e], a, acc[j][k]);
            }
        }
    }
    const int ntiles = gridDim.x;
#pragma unroll
    for (int j = 0; j < CBR; ++j)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float s = wave_reduce63(acc[j][v]);
            if (lane == 63) part[((size_t)(c0 + j) * ntiles + tile) * NV + v] = s;
        }
    if (first) {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const float s = wave_reduce63(bs[k]);
            if (lane == 63) bpart[(size_t)tile * KT + k] = s;
        }
    }
}

// ---- backward, finalize: block c < C: ordered sums of channel c's partials (double) -> dgamma, dbeta, batch means, dw[:, c];
//      block C: dbias.  256 threads: lane v of wave q adds the tiles q, q + 4, ... of value v (four loads in flight), the four wave
//      sums are added in a fixed order ----
__global__ __launch_bounds__(256) void bn_cls_bwd_final_kernel(const float* __restrict__ part, const float* __restrict__ bpart, int ntiles,
                                                               int C, int K, int KT, double count, int training,
                                                               float* __restrict__ dbn_w, float* __restrict__ dbn_b,
                                                               float* __restrict__ dw_cls, float* __restrict__ dbias,
                                                               float* __restrict__ coef) {
    __shared__ double red[4][64];
    const int c = blockIdx.x, v = threadIdx.x & 63, q = threadIdx.x >> 6, NV = KT + 2;
    const bool bias_blk = c == C;
    const int nv = bias_blk ? K : NV;
    double s = 0.0;
    if (v < nv) {
        const float* src = bias_blk ? bpart + v : part + (size_t)c * ntiles * NV + v;
        const size_t stride = bias_blk ? (size_t)KT : (size_t)NV;
        int t = q;
        for (; t + 12 < ntiles; t += 16) {
            const float x0 = src[(size_t)t * stride], x1 = src[(size_t)(t + 4) * stride], x2 = src[(size_t)(t + 8) * stride],
                        x3 = src[(size_t)(t + 12) * stride];
            s += (double)x0, s += (double)x1, s += (double)x2, s += (double)x3;
        }
        for (; t < ntiles; t += 4) s += (double)src[(size_t)t * stride];
    }
    red[q][v] = s;
    __syncthreads();
    if (q != 0 || v >= nv) return;
    s = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
    if (bias_blk) {
        if (dbias != nullptr) dbias[v] = (float)s;
        return;
    }
    if (v < KT) {
        if (v < K) dw_cls[(size_t)v * C + c] = (float)s;
    } else if (v == KT) {
        dbn_b[c] = (float)s;
        coef[2 * c] = training ? (float)(s / count) : 0.f;
    } else {
        dbn_w[c] = (float)s;
        coef[2 * c + 1] = training ? (float)(s / count) : 0.f;
    }
}

// ---- backward, pass 2: same shape as the forward (64 positions x channel quarter per wave); no reduction ----
template <int KT>
__global__ __launch_bounds__(256) void bn_cls_dx_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                         const float* __restrict__ tab, const float* __restrict__ coef, int C, int P, int K,
                                                         int tiles, float* __restrict__ dz) {
    constexpr int TS = KT + BC_X, U = 8;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int p = tile * 64 + lane, pc = min(p, P - 1);
    const int cq = C >> 2, c0 = wave * cq;
    float g[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) g[k] = k < K ? dy[((size_t)b * K + (k < K ? k : 0)) * P + pc] : 0.f;
    const float* zp = z + ((size_t)b * C + c0) * P + pc;
    float* op = dz + ((size_t)b * C + c0) * P + pc;
    for (int cc = 0; cc < cq; cc += U) {
        float zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zv[u] = zp[(size_t)(cc + u) * P];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* row = tab + (size_t)(c0 + cc + u) * TS;   // wave-uniform: scalar loads
            const float xh = (zv[u] - row[KT]) * row[KT + 1], pre = fmaf(xh, row[KT + 2], row[KT + 3]);
            float da = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) da = fmaf(row[k], g[k], da);
            const float du = pre > 0.f ? da : 0.f;
            const float m1 = coef[2 * (c0 + cc + u)], m2 = coef[2 * (c0 + cc + u) + 1];
            if (p < P) op[(size_t)(cc + u) * P] = row[KT + 4] * (du - m1 - xh * m2);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
bool bn_cls_supported(int C, int K, int P) { return C > 0 && (C % 64) == 0 && K > 0 && K <= 32 && P > 0 && (P & 3) == 0; }
int bn_cls_table_floats(int C, int K) { return C * (bc_kt(K) + BC_X); }

static size_t fbytes(size_t n) { return align_up(n * sizeof(float), 256); }
static int bc_ntiles(int B, int P) { return B * ceil_div(P, BC_TPW); }

size_t bn_cls_fwd_workspace(int B, int C, int P) { return fbytes(2 * (size_t)C) + bn_act_workspace(B, C, P); }
size_t bn_cls_bwd_workspace(int B, int C, int K, int P) {
    const int KT = bc_kt(K);
    return fbytes((size_t)C * bc_ntiles(B, P) * (KT + 2)) + fbytes((size_t)bc_ntiles(B, P) * KT) + fbytes(2 * (size_t)C);
}

hipError_t bn_cls_fwd_run(const float* z, const float* conv_part, const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                          const float* w_cls, const float* bias, int B, int C, int K, int H, int W, int training, float momentum,
                          float eps, float* y, float* tab, void* ws, hipStream_t stream) {
    const int P = H * W, KT = bc_kt(K), tiles = ceil_div(P, 64);
    float* save_mean = static_cast<float*>(ws);
    float* save_invstd = save_mean + C;
    int mode = 1;
    if (training && conv_part == nullptr) {   // no producer statistics: K7's statistics pass + finalize in front
        void* ws2 = static_cast<char*>(ws) + fbytes(2 * (size_t)C);
        if (hipError_t e = bn_stats_run(z, run_mean, run_var, B, C, P, 1, momentum, eps, save_mean, save_invstd, ws2, stream); e != hipSuccess)
            return e;
        mode = 0;
    }
    hipLaunchKernelGGL(bn_cls_table_kernel, dim3(C), dim3(BA_T), 0, stream, conv_part, mode, B, C, P, H, W, training, momentum, eps,
                       run_mean, run_var, save_mean, save_invstd, bn_w, bn_b, w_cls, K, KT, tab);
    const dim3 grid(B * tiles), block(256);
    switch (KT) {
        case 8: hipLaunchKernelGGL(bn_cls_fwd_kernel<8>, grid, block, 0, stream, z, tab, bias, C, P, K, tiles, y); break;
        case 20: hipLaunchKernelGGL(bn_cls_fwd_kernel<20>, grid, block, 0, stream, z, tab, bias, C, P, K, tiles, y); break;
        default: hipLaunchKernelGGL(bn_cls_fwd_kernel<32>, grid, block, 0, stream, z, tab, bias, C, P, K, tiles, y); break;
    }
    return hipGetLastError();
}

hipError_t bn_cls_bwd_run(const float* dy, const float* z, const float* tab, int B, int C, int K, int H, int W, int training, float* dz,
                          float* dbn_w, float* dbn_b, float* dw_cls, float* dbias, void* ws, hipStream_t stream) {
    const int P = H * W, KT = bc_kt(K), tiles_img = ceil_div(P, BC_TPW), ntiles = B * tiles_img, tiles = ceil_div(P, 64);
    float* part = static_cast<float*>(ws);
    float* bpart = reinterpret_cast<float*>(static_cast<char*>(ws) + fbytes((size_t)C * ntiles * (KT + 2)));
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(bpart) + fbytes((size_t)ntiles * KT));
    const dim3 block(256);
    switch (KT) {
        case 8: hipLaunchKernelGGL((bn_cls_reduce_kernel<8, 8>), dim3(ntiles, ceil_div(C, 32)), block, 0, stream, dy, z, tab, C, P, K, tiles_img, part, bpart); break;
        case 20: hipLaunchKernelGGL((bn_cls_reduce_kernel<20, 4>), dim3(ntiles, ceil_div(C, 16)), block, 0, stream, dy, z, tab, C, P, K, tiles_img, part, bpart); break;
        default: hipLaunchKernelGGL((bn_cls_reduce_kernel<32, 2>), dim3(ntiles, ceil_div(C, 8)), block, 0, stream, dy, z, tab, C, P, K, tiles_img, part, bpart); break;
    }
    hipLaunchKernelGGL(bn_cls_bwd_final_kernel, dim3(C + 1), dim3(256), 0, stream, part, bpart, ntiles, C, K, KT, (double)B * (double)P, training,
                       dbn_w, dbn_b, dw_cls, dbias, coef);
    const dim3 grid(B * tiles);
    switch (KT) {
        case 8: hipLaunchKernelGGL(bn_cls_dx_kernel<8>, grid, block, 0, stream, dy, z, tab, coef, C, P, K, tiles, dz); break;
        case 20: hipLaunchKernelGGL(bn_cls_dx_kernel<20>, grid, block, 0, stream, dy, z, tab, coef, C, P, K, tiles, dz); break;
        default: hipLaunchKernelGGL(bn_cls_dx_kernel<32>, grid, block, 0, stream, dy, z, tab, coef, C, P, K, tiles, dz); break;
    }
    return hipGetLastError();
}

}  // namespace cabinet
