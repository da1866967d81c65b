// K8 -- depthwise KxK convolution (groups == channels, no bias), forward and backward, NCHW fp32.
//
// Replaces the depthwise nn.Conv2d of the reference's MBConv blocks (src/models/mobilenetv3.py:118-126,135-143:
// kernel 3 or 5, stride 1 or 2, padding k//2) for which MIOpen on gfx950 falls back to its `naive_conv_*` solvers
// (fp32, groups == channels): 12 % of the config-3 step, plus an im2col + 16x16-tile GEMM weight gradient.
//
// A depthwise convolution never mixes channels: every (b,c) plane is an independent KxK stencil with its own
// filter -- HBM-bound (read x, write y), no matrix product to feed.  One workgroup computes a 1024-output tile of
// one plane from an LDS copy of the input tile (+halo); a thread owns 4 vertically adjacent outputs so lanes of a
// wave read consecutive LDS words and each loaded row feeds up to K of its outputs.
//   fwd : y[oy][ox] = sum w[ky][kx] * x[oy*S - p + ky][ox*S - p + kx]
//   bwd : ONE kernel per input-space tile, sharing the staged dy tile between both gradients:
//           dx[y][x]   = sum w[ky][kx] * dy[(y+p-ky)/S][(x+p-kx)/S]      (taps where the division is exact)
//           dw[ky][kx] = sum_{b,y,x} x[y][x] * dy[(y+p-ky)/S][(x+p-kx)/S]  -> per-tile partials, ordered final sum
//         stride 1 is the forward stencil with the flipped filter; stride 2 works on 2x2 input quads so that every
//         tap is used exactly once per quad with compile-time offsets (these kernels are VALU-bound, not HBM-bound:
//         the first, generic form spent ~100 instructions per pixel on parity masks and clamps)
// No atomics anywhere: bitwise reproducible (MIOpen's naive backward is, too; its wrw GEMM path is not).
#include "act.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int DW_T = 256;     // threads
constexpr int DW_OUT = 1024;  // outputs (fwd) / input pixels (bwd) per tile: TH x TW with TW in {64,32,16,8}

struct DwShape {
    int B, C, H, W, Ho, Wo;
};

static int dw_tile_w(int w) { return w > 32 ? 64 : w > 16 ? 32 : w > 8 ? 16 : 8; }

template <int K, int S>
__global__ __launch_bounds__(DW_T) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                           DwShape s, int TW, int tiles_x, int tiles_y, BnFold f,
                                                           float* __restrict__ y) {
    constexpr int PAD = K / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TH = DW_OUT / TW;
    const int tile = blockIdx.x % (tiles_x * tiles_y), plane = blockIdx.x / (tiles_x * tiles_y);
    const int c = plane % s.C;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int in_h = (TH - 1) * S + K, in_w = (TW - 1) * S + K;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const float* xp = x + (size_t)plane * s.H * s.W;
    if (f.mean) {  // the input is act(bn(z)): evaluated while staging; the zero padding is that of the activated map
        const float mu = f.mean[c], inv = f.invstd[c], gam = f.weight[c], bet = f.bias[c];
        for (int i = threadIdx.x; i < in_h * in_w; i += DW_T) {
            const int r = i / in_w, q = i - r * in_w, iy = iy0 + r, ix = ix0 + q;
            smem[i] = (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W)
                          ? act_fwd(fmaf((xp[(size_t)iy * s.W + ix] - mu) * inv, gam, bet), f.act)
                          : 0.f;
        }
    } else {
        for (int i = threadIdx.x; i < in_h * in_w; i += DW_T) {
            const int r = i / in_w, q = i - r * in_w, iy = iy0 + r, ix = ix0 + q;
            smem[i] = (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) ? xp[(size_t)iy * s.W + ix] : 0.f;
        }
    }
    float w[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) w[ky][kx] = wgt[(c * K + ky) * K + kx];
    __syncthreads();
    const int tx = threadIdx.x & (TW - 1), tq = threadIdx.x / TW;  // 4 outputs: rows tq*4 .. tq*4+3, column tx
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* base = smem + (tq * 4 * S) * in_w + tx * S;
#pragma unroll
    for (int r = 0; r < 3 * S + K; ++r) {
        float v[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) v[kx] = base[r * in_w + kx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ky = r - j * S;  // compile-time after unrolling
            if (ky >= 0 && ky < K) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[j] = fmaf(w[ky][kx], v[kx], acc[j]);
            }
        }
    }
    float* yp = y + (size_t)plane * s.Ho * s.Wo;
    const int ox = ox0 + tx;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + tq * 4 + j;
        if (oy < s.Ho && ox < s.Wo) yp[(size_t)oy * s.Wo + ox] = acc[j];
    }
}

// ---- backward: dx and the per-tile weight-gradient partials part[c][b*tiles + tile][K*K], one input-space tile ----
// workgroup reduction of the K*K per-thread partial sums (fixed order); red is [KK][DW_T + 8] (row of 32 padded to 33)
template <int KK>
__device__ __forceinline__ void dw_reduce_partials(const float (&pw)[KK], float* red, float* __restrict__ out) {
    constexpr int LD = DW_T + 8;
    const int slot = threadIdx.x + (threadIdx.x >> 5);
#pragma unroll
    for (int t = 0; t < KK; ++t) red[t * LD + slot] = pw[t];
    __syncthreads();
    if (threadIdx.x < KK * 8) {
        const int tap = threadIdx.x >> 3, seg = threadIdx.x & 7;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[tap * LD + seg * 33 + i];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        if (seg == 0) out[tap] = t;
    }
}

// With a folded BatchNorm (BnFold, act.hpp) the backward kernels also (i) rebuild their x operand a = act(bn(z)) from z
// and (ii) turn the input gradient da they produce into the BatchNorm-backward partial sums of this tile,
// sum du and sum du * xhat with du = da * act'(u): the BatchNorm's own reduce pass over (da, z) disappears.

// stride 1: with the flipped filter wf[a][b] = w[K-1-a][K-1-b] the input gradient is the SAME stencil as forward,
// applied to dy (pad K/2), and dw[K-1-a][K-1-b] = sum x[y][x] * dy[y-p+a][x-p+b]: one staged dy tile (+halo),
// a thread owns 4 vertically adjacent pixels, every LDS row it loads feeds up to K of them -- no masks.
template <int K>
__global__ __launch_bounds__(DW_T) void dwconv_bwd_s1_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ wgt, DwShape s, int TW, int tiles_x,
                                                              int tiles_y, BnFold f, float* __restrict__ dx,
                                                              float* __restrict__ part, float* __restrict__ bnpart) {
    constexpr int PAD = K / 2, KK = K * K;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red4[4];
    const int TH = DW_OUT / TW;
    const int ntile = tiles_x * tiles_y;
    const int tile = blockIdx.x % ntile, plane = blockIdx.x / ntile;
    const int b = plane / s.C, c = plane - b * s.C;
    const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
    const int in_h = TH + K - 1, in_w = TW + K - 1;
    float* tile_dy = smem;               // [in_h][in_w]
    float* red = smem + in_h * in_w;     // [KK][DW_T + 8]
    const float* dyp = dy + (size_t)plane * s.H * s.W;  // stride 1: dy has the input's size
    for (int i = threadIdx.x; i < in_h * in_w; i += DW_T) {
        const int r = i / in_w, q = i - r * in_w, oy = y0 - PAD + r, ox = x0 - PAD + q;
        tile_dy[i] = (oy >= 0 && oy < s.H && ox >= 0 && ox < s.W) ? dyp[(size_t)oy * s.W + ox] : 0.f;
    }
    float wf[K][K];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int bb = 0; bb < K; ++bb) wf[a][bb] = wgt[(c * K + (K - 1 - a)) * K + (K - 1 - bb)];
    __syncthreads();
    const int tx = threadIdx.x & (TW - 1), tq = threadIdx.x / TW;
    const int xx = x0 + tx;
    const float* xp = x + (size_t)plane * s.H * s.W;
    float xv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
    float xh[4], dact[4];  // folded BatchNorm: xhat and act'(u) of the thread's pixels
    float mu = 0.f, inv = 1.f, gam = 1.f, bet = 0.f;
    if (f.mean) mu = f.mean[c], inv = f.invstd[c], gam = f.weight[c], bet = f.bias[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = y0 + tq * 4 + j;
        const bool live = yy < s.H && xx < s.W;
        xv[j] = live ? xp[(size_t)yy * s.W + xx] : 0.f;
        xh[j] = 0.f, dact[j] = 0.f;
        if (f.mean) {
            xh[j] = (xv[j] - mu) * inv;
            const float u = fmaf(xh[j], gam, bet);
            xv[j] = live ? act_fwd(u, f.act) : 0.f;
            dact[j] = live ? act_grad(u, f.act) : 0.f;
        }
    }
    float pw[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) pw[t] = 0.f;
    const float* base = tile_dy + (tq * 4) * in_w + tx;
#pragma unroll
    for (int r = 0; r < 3 + K; ++r) {
        float v[K];
#pragma unroll
        for (int bb = 0; bb < K; ++bb) v[bb] = base[r * in_w + bb];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = r - j;  // compile-time after unrolling
            if (a >= 0 && a < K) {
#pragma unroll
                for (int bb = 0; bb < K; ++bb) {
                    acc[j] = fmaf(wf[a][bb], v[bb], acc[j]);
                    pw[(K - 1 - a) * K + (K - 1 - bb)] = fmaf(xv[j], v[bb], pw[(K - 1 - a) * K + (K - 1 - bb)]);
                }
            }
        }
    }
    float* dxp = dx + (size_t)plane * s.H * s.W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = y0 + tq * 4 + j;
        if (yy < s.H && xx < s.W) dxp[(size_t)yy * s.W + xx] = acc[j];
    }
    if (f.mean) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float du = acc[j] * dact[j];
            s1 += du, s2 = fmaf(du, xh[j], s2);
        }
        s1 = block_sum_256(s1, red4);
        s2 = block_sum_256(s2, red4);
        if (threadIdx.x == 0) {
            const int nt = s.B * ntile;
            bnpart[(size_t)c * nt + b * ntile + tile] = s1;
            bnpart[((size_t)s.C + c) * nt + b * ntile + tile] = s2;
        }
    }
    dw_reduce_partials<KK>(pw, red, part + ((size_t)c * (s.B * ntile) + (size_t)b * ntile + tile) * KK);
}

// stride 2: a thread owns the 2x2 input quad (2n+py, 2m+px).  A tap (ky,kx) reaches pixel parity
// (py,px) = ((ky+p)&1, (kx+p)&1) only, from dy[n + (py+p-ky)/2][m + (px+p-kx)/2]: every tap is used exactly once per
// quad, all offsets are compile-time, and the <= 3x3 dy neighbourhood is read from LDS once.
template <int K>
__global__ __launch_bounds__(DW_T) void dwconv_bwd_s2_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ wgt, DwShape s, int TW, int tiles_x,
                                                              int tiles_y, BnFold f, float* __restrict__ dx,
                                                              float* __restrict__ part, float* __restrict__ bnpart) {
    constexpr int PAD = K / 2, KK = K * K;
    __shared__ float red4[4];
    // smallest / largest dy offset (py + PAD - ky) / 2 over the exact divisions: K=3 -> 0..1, K=5 -> -1..1
    constexpr int OLO = (K == 3) ? 0 : -1, OHI = 1, NO = OHI - OLO + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TH = DW_OUT / TW, TWm = TW / 2, THn = TH / 2;
    const int ntile = tiles_x * tiles_y;
    const int tile = blockIdx.x % ntile, plane = blockIdx.x / ntile;
    const int b = plane / s.C, c = plane - b * s.C;
    const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;  // even
    const int n0 = y0 / 2, m0 = x0 / 2;
    const int th = THn + NO - 1, tw = TWm + NO - 1;
    float* tile_dy = smem;          // [th][tw], origin (n0 + OLO, m0 + OLO)
    float* red = smem + th * tw;    // [KK][DW_T + 8]
    const float* dyp = dy + (size_t)plane * s.Ho * s.Wo;
    for (int i = threadIdx.x; i < th * tw; i += DW_T) {
        const int r = i / tw, q = i - r * tw, oy = n0 + OLO + r, ox = m0 + OLO + q;
        tile_dy[i] = (oy >= 0 && oy < s.Ho && ox >= 0 && ox < s.Wo) ? dyp[(size_t)oy * s.Wo + ox] : 0.f;
    }
    float w[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) w[ky][kx] = wgt[(c * K + ky) * K + kx];
    __syncthreads();
    const int tm = threadIdx.x & (TWm - 1), tn = threadIdx.x / TWm;
    float g[NO][NO];
#pragma unroll
    for (int a = 0; a < NO; ++a)
#pragma unroll
        for (int bb = 0; bb < NO; ++bb) g[a][bb] = tile_dy[(tn + a) * tw + tm + bb];
    const float* xp = x + (size_t)plane * s.H * s.W;
    float* dxp = dx + (size_t)plane * s.H * s.W;
    float pw[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) pw[t] = 0.f;
    float mu = 0.f, inv = 1.f, gam = 1.f, bet = 0.f, s1 = 0.f, s2 = 0.f;
    if (f.mean) mu = f.mean[c], inv = f.invstd[c], gam = f.weight[c], bet = f.bias[c];
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const int yy = y0 + 2 * tn + py;
        float xv[2], xh[2] = {0.f, 0.f}, dact[2] = {0.f, 0.f}, acc[2] = {0.f, 0.f};
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int xx = x0 + 2 * tm + px;
            const bool live = yy < s.H && xx < s.W;
            xv[px] = live ? xp[(size_t)yy * s.W + xx] : 0.f;
            if (f.mean) {
                xh[px] = (xv[px] - mu) * inv;
                const float u = fmaf(xh[px], gam, bet);
                xv[px] = live ? act_fwd(u, f.act) : 0.f;
                dact[px] = live ? act_grad(u, f.act) : 0.f;
            }
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            if (((py + PAD - ky) & 1) != 0) continue;           // compile-time
            const int a = (py + PAD - ky) / 2 - OLO;            // row of g
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    if (((px + PAD - kx) & 1) != 0) continue;
                    const int bb = (px + PAD - kx) / 2 - OLO;
                    acc[px] = fmaf(w[ky][kx], g[a][bb], acc[px]);
                    pw[ky * K + kx] = fmaf(xv[px], g[a][bb], pw[ky * K + kx]);
                }
        }
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int xx = x0 + 2 * tm + px;
            if (yy < s.H && xx < s.W) dxp[(size_t)yy * s.W + xx] = acc[px];
            const float du = acc[px] * dact[px];
            s1 += du, s2 = fmaf(du, xh[px], s2);
        }
    }
    if (f.mean) {
        s1 = block_sum_256(s1, red4);
        s2 = block_sum_256(s2, red4);
        if (threadIdx.x == 0) {
            const int nt = s.B * ntile;
            bnpart[(size_t)c * nt + b * ntile + tile] = s1;
            bnpart[((size_t)s.C + c) * nt + b * ntile + tile] = s2;
        }
    }
    dw_reduce_partials<KK>(pw, red, part + ((size_t)c * (s.B * ntile) + (size_t)b * ntile + tile) * KK);
}

// dw[c][tap] = sum over the B*ntile partials (double accumulation, fixed order)
__global__ __launch_bounds__(256) void dwconv_dw_finalize_kernel(const float* __restrict__ part, int nparts, int KK,
                                                                  float* __restrict__ dw) {
    __shared__ double dred[4];
    const int c = blockIdx.x / KK, tap = blockIdx.x - c * KK;
    double t = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) t += (double)part[((size_t)c * nparts + i) * KK + tap];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) dw[blockIdx.x] = (float)((dred[0] + dred[1]) + (dred[2] + dred[3]));
}

bool dwconv_supported(int K, int S) { return (K == 3 || K == 5) && (S == 1 || S == 2); }

static void out_size(int H, int W, int K, int S, int& Ho, int& Wo) {
    Ho = (H + 2 * (K / 2) - K) / S + 1;
    Wo = (W + 2 * (K / 2) - K) / S + 1;
}

size_t dwconv_bwd_workspace(int B, int C, int H, int W, int K) {
    const int TW = dw_tile_w(W), TH = DW_OUT / TW;
    return align_up((size_t)C * B * ceil_div(W, TW) * ceil_div(H, TH) * K * K * sizeof(float), 256);
}

static int bwd_tiles(int H, int W) {
    const int TW = dw_tile_w(W), TH = DW_OUT / TW;
    return ceil_div(W, TW) * ceil_div(H, TH);
}

template <int K, int S>
static hipError_t fwd_launch(const float* x, const float* w, const DwShape& s, const BnFold& f, float* y,
                             hipStream_t stream) {
    const int TW = dw_tile_w(s.Wo), TH = DW_OUT / TW;
    const int tiles_x = ceil_div(s.Wo, TW), tiles_y = ceil_div(s.Ho, TH);
    const size_t lds = (size_t)((TH - 1) * S + K) * ((TW - 1) * S + K) * sizeof(float);
    hipLaunchKernelGGL((dwconv_fwd_kernel<K, S>), dim3((unsigned)((size_t)s.B * s.C * tiles_x * tiles_y)), dim3(DW_T), lds,
                       stream, x, w, s, TW, tiles_x, tiles_y, f, y);
    return hipGetLastError();
}

template <int K, int S>
static hipError_t bwd_launch(const float* dy, const float* x, const float* w, const DwShape& s, const BnFold& f,
                             float* dx, float* dw, float* part, float* bnpart, hipStream_t stream) {
    const int TW = dw_tile_w(s.W), TH = DW_OUT / TW;
    const int tiles_x = ceil_div(s.W, TW), tiles_y = ceil_div(s.H, TH);
    const size_t tile_floats = S == 1 ? (size_t)(TH + K - 1) * (TW + K - 1) : (size_t)(TH / 2 + 2) * (TW / 2 + 2);
    const size_t lds = (tile_floats + (size_t)K * K * (DW_T + 8)) * sizeof(float);
    const dim3 grid((unsigned)((size_t)s.B * s.C * tiles_x * tiles_y));
    if (S == 1)
        hipLaunchKernelGGL((dwconv_bwd_s1_kernel<K>), grid, dim3(DW_T), lds, stream, dy, x, w, s, TW, tiles_x, tiles_y, f,
                           dx, part, bnpart);
    else
        hipLaunchKernelGGL((dwconv_bwd_s2_kernel<K>), grid, dim3(DW_T), lds, stream, dy, x, w, s, TW, tiles_x, tiles_y, f,
                           dx, part, bnpart);
    hipLaunchKernelGGL(dwconv_dw_finalize_kernel, dim3(s.C * K * K), dim3(256), 0, stream, part, s.B * tiles_x * tiles_y,
                       K * K, dw);
    return hipGetLastError();
}

static hipError_t fwd_dispatch(const float* x, const float* w, const DwShape& s, int K, int S, const BnFold& f, float* y,
                               hipStream_t stream) {
    if (K == 3 && S == 1) return fwd_launch<3, 1>(x, w, s, f, y, stream);
    if (K == 3 && S == 2) return fwd_launch<3, 2>(x, w, s, f, y, stream);
    if (K == 5 && S == 1) return fwd_launch<5, 1>(x, w, s, f, y, stream);
    return fwd_launch<5, 2>(x, w, s, f, y, stream);
}

static hipError_t bwd_dispatch(const float* dy, const float* x, const float* w, const DwShape& s, int K, int S,
                               const BnFold& f, float* dx, float* dw, float* part, float* bnpart, hipStream_t stream) {
    if (K == 3 && S == 1) return bwd_launch<3, 1>(dy, x, w, s, f, dx, dw, part, bnpart, stream);
    if (K == 3 && S == 2) return bwd_launch<3, 2>(dy, x, w, s, f, dx, dw, part, bnpart, stream);
    if (K == 5 && S == 1) return bwd_launch<5, 1>(dy, x, w, s, f, dx, dw, part, bnpart, stream);
    return bwd_launch<5, 2>(dy, x, w, s, f, dx, dw, part, bnpart, stream);
}

hipError_t dwconv_fwd_run(const float* x, const float* w, int B, int C, int H, int W, int K, int S, float* y,
                          hipStream_t stream) {
    DwShape s{B, C, H, W, 0, 0};
    out_size(H, W, K, S, s.Ho, s.Wo);
    return fwd_dispatch(x, w, s, K, S, BnFold{nullptr, nullptr, nullptr, nullptr, 0}, y, stream);
}

hipError_t dwconv_bwd_run(const float* dy, const float* x, const float* w, int B, int C, int H, int W, int K, int S,
                          float* dx, float* dw, void* ws, hipStream_t stream) {
    DwShape s{B, C, H, W, 0, 0};
    out_size(H, W, K, S, s.Ho, s.Wo);
    return bwd_dispatch(dy, x, w, s, K, S, BnFold{nullptr, nullptr, nullptr, nullptr, 0}, dx, dw, static_cast<float*>(ws),
                        nullptr, stream);
}

// ---- BatchNorm2d (+activation) -> depthwise convolution as one operator (reference mobilenetv3.py:135-143:
// `BatchNorm2d(hidden), act, depthwise Conv2d`): the normalised, activated tensor is never written or re-read.
//   fwd : BN statistics of z (one read), then the convolution reads z again and normalises while staging
//   bwd : the convolution's backward rebuilds a = act(bn(z)) from z, writes da and the BN-backward partial sums;
//         the BN-backward dx pass (da, z -> dz) finishes.  6 passes over the (B,C,H,W) tensor instead of 10.
struct BnDwWs {
    size_t bn, part, bnpart, coef, da, total;
};
static BnDwWs bn_dw_layout(int B, int C, int H, int W, int K) {
    BnDwWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const int nt = B * bwd_tiles(H, W);
    w.bn = take(bn_act_workspace(B, C, H * W));
    w.part = take((size_t)C * nt * K * K * sizeof(float));
    w.bnpart = take((size_t)2 * C * nt * sizeof(float));
    w.coef = take((size_t)2 * C * sizeof(float));
    w.da = take((size_t)B * C * H * W * sizeof(float));
    w.total = off;
    return w;
}
size_t bn_dwconv_fwd_workspace(int B, int C, int H, int W) { return bn_act_workspace(B, C, H * W); }
size_t bn_dwconv_bwd_workspace(int B, int C, int H, int W, int K) { return bn_dw_layout(B, C, H, W, K).total; }

hipError_t bn_dwconv_fwd_run(const float* z, const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                             const float* w, int B, int C, int H, int W, int K, int S, int act, int training,
                             float momentum, float eps, float* y, float* save_mean, float* save_invstd, void* ws,
                             hipStream_t stream) {
    hipError_t e = bn_stats_run(z, run_mean, run_var, B, C, H * W, training, momentum, eps, save_mean, save_invstd, ws,
                                stream);
    if (e != hipSuccess) return e;
    DwShape s{B, C, H, W, 0, 0};
    out_size(H, W, K, S, s.Ho, s.Wo);
    return fwd_dispatch(z, w, s, K, S, BnFold{save_mean, save_invstd, bn_w, bn_b, act}, y, stream);
}

hipError_t bn_dwconv_bwd_run(const float* dy, const float* z, const float* bn_w, const float* bn_b,
                             const float* save_mean, const float* save_invstd, const float* w, int B, int C, int H,
                             int W, int K, int S, int act, int training, float* dz, float* dbn_w, float* dbn_b,
                             float* dw, void* ws, hipStream_t stream) {
    const BnDwWs L = bn_dw_layout(B, C, H, W, K);
    char* base = static_cast<char*>(ws);
    auto at = [&](size_t o) { return reinterpret_cast<float*>(base + o); };
    DwShape s{B, C, H, W, 0, 0};
    out_size(H, W, K, S, s.Ho, s.Wo);
    hipError_t e = bwd_dispatch(dy, z, w, s, K, S, BnFold{save_mean, save_invstd, bn_w, bn_b, act}, at(L.da), dw,
                                at(L.part), at(L.bnpart), stream);
    if (e != hipSuccess) return e;
    return bn_bwd_tail_run(at(L.bnpart), B * bwd_tiles(H, W), at(L.da), z, bn_w, bn_b, save_mean, save_invstd, B, C,
                           H * W, act, training, dz, dbn_w, dbn_b, at(L.coef), stream);
}

}  // namespace cabinet
