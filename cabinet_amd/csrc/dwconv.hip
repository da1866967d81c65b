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
// No atomics anywhere: bitwise reproducible (MIOpen's naive backward is, too; its wrw GEMM path is not).
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
                                                           DwShape s, int TW, int tiles_x, int tiles_y,
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
    for (int i = threadIdx.x; i < in_h * in_w; i += DW_T) {
        const int r = i / in_w, q = i - r * in_w, iy = iy0 + r, ix = ix0 + q;
        smem[i] = (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) ? xp[(size_t)iy * s.W + ix] : 0.f;
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

// input-space tile: dx and the per-tile weight-gradient partials part[c][b*tiles + tile][K*K]
template <int K, int S>
__global__ __launch_bounds__(DW_T) void dwconv_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ wgt, DwShape s, int TW, int tiles_x,
                                                           int tiles_y, float* __restrict__ dx,
                                                           float* __restrict__ part) {
    constexpr int PAD = K / 2, KK = K * K;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TH = DW_OUT / TW;
    const int ntile = tiles_x * tiles_y;
    const int tile = blockIdx.x % ntile, plane = blockIdx.x / ntile;
    const int b = plane / s.C, c = plane - b * s.C;
    const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
    // dy rows/cols any pixel of the tile can touch: o = (i + PAD - k) / S, k in [0,K)
    const int oy_lo = (y0 + PAD - (K - 1) + S - 1 + S * 1024) / S - 1024;  // ceil(./S), numerator may be negative
    const int ox_lo = (x0 + PAD - (K - 1) + S - 1 + S * 1024) / S - 1024;
    const int th = (y0 + TH - 1 + PAD) / S - oy_lo + 1, tw = (x0 + TW - 1 + PAD) / S - ox_lo + 1;
    float* tile_dy = smem;            // [th][tw]
    float* red = smem + th * tw;      // [KK][DW_T]
    const float* dyp = dy + (size_t)plane * s.Ho * s.Wo;
    for (int i = threadIdx.x; i < th * tw; i += DW_T) {
        const int r = i / tw, q = i - r * tw, oy = oy_lo + r, ox = ox_lo + q;
        tile_dy[i] = (oy >= 0 && oy < s.Ho && ox >= 0 && ox < s.Wo) ? dyp[(size_t)oy * s.Wo + ox] : 0.f;
    }
    float w[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) w[ky][kx] = wgt[(c * K + ky) * K + kx];
    __syncthreads();
    const int tx = threadIdx.x & (TW - 1), tq = threadIdx.x / TW;
    const int xx = x0 + tx;
    const float* xp = x + (size_t)plane * s.H * s.W;
    float* dxp = dx + (size_t)plane * s.H * s.W;
    float pw[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) pw[ky][kx] = 0.f;
    // column taps of this lane: ox = (xx + PAD - kx) / S where exact; mask and LDS column per kx
    int colq[K];
    float colm[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const int t = xx + PAD - kx;
        const bool ok = S == 1 || (t % S) == 0;  // t >= 0 inside the staged range whenever it matters
        colq[kx] = ok ? (t >= 0 ? t / S : -1) - ox_lo : 0;
        colm[kx] = (ok && colq[kx] >= 0 && colq[kx] < tw) ? 1.f : 0.f;
        colq[kx] = min(max(colq[kx], 0), tw - 1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = y0 + tq * 4 + j;
        const bool live = yy < s.H && xx < s.W;
        const float xv = live ? xp[(size_t)yy * s.W + xx] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int t = yy + PAD - ky;
            if (S > 1 && (t % S) != 0) continue;  // uniform per wave row group: yy is the same for a whole row
            const int r = (t >= 0 ? t / S : -1) - oy_lo;
            if (r < 0 || r >= th) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float g = tile_dy[r * tw + colq[kx]] * colm[kx];
                acc = fmaf(w[ky][kx], g, acc);
                pw[ky][kx] = fmaf(xv, g, pw[ky][kx]);
            }
        }
        if (live) dxp[(size_t)yy * s.W + xx] = acc;
    }
    // workgroup reduction of the K*K partial sums, fixed order
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) red[(ky * K + kx) * DW_T + threadIdx.x] = pw[ky][kx];
    __syncthreads();
    if (threadIdx.x < KK * 8) {
        const int tap = threadIdx.x >> 3, seg = threadIdx.x & 7;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[tap * DW_T + seg * 32 + i];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        if (seg == 0) part[((size_t)c * (s.B * ntile) + (size_t)b * ntile + tile) * KK + tap] = t;
    }
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

template <int K, int S>
static hipError_t fwd_launch(const float* x, const float* w, const DwShape& s, float* y, hipStream_t stream) {
    const int TW = dw_tile_w(s.Wo), TH = DW_OUT / TW;
    const int tiles_x = ceil_div(s.Wo, TW), tiles_y = ceil_div(s.Ho, TH);
    const size_t lds = (size_t)((TH - 1) * S + K) * ((TW - 1) * S + K) * sizeof(float);
    hipLaunchKernelGGL((dwconv_fwd_kernel<K, S>), dim3((unsigned)((size_t)s.B * s.C * tiles_x * tiles_y)), dim3(DW_T), lds,
                       stream, x, w, s, TW, tiles_x, tiles_y, y);
    return hipGetLastError();
}

template <int K, int S>
static hipError_t bwd_launch(const float* dy, const float* x, const float* w, const DwShape& s, float* dx, float* dw,
                             void* ws, hipStream_t stream) {
    const int TW = dw_tile_w(s.W), TH = DW_OUT / TW;
    const int tiles_x = ceil_div(s.W, TW), tiles_y = ceil_div(s.H, TH);
    const int th = (TH + K - 1) / S + 2, tw = (TW + K - 1) / S + 2;  // upper bound of the staged dy tile
    const size_t lds = ((size_t)th * tw + (size_t)K * K * DW_T) * sizeof(float);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL((dwconv_bwd_kernel<K, S>), dim3((unsigned)((size_t)s.B * s.C * tiles_x * tiles_y)), dim3(DW_T), lds,
                       stream, dy, x, w, s, TW, tiles_x, tiles_y, dx, part);
    hipLaunchKernelGGL(dwconv_dw_finalize_kernel, dim3(s.C * K * K), dim3(256), 0, stream, part, s.B * tiles_x * tiles_y,
                       K * K, dw);
    return hipGetLastError();
}

hipError_t dwconv_fwd_run(const float* x, const float* w, int B, int C, int H, int W, int K, int S, float* y,
                          hipStream_t stream) {
    DwShape s{B, C, H, W, 0, 0};
    out_size(H, W, K, S, s.Ho, s.Wo);
    if (K == 3 && S == 1) return fwd_launch<3, 1>(x, w, s, y, stream);
    if (K == 3 && S == 2) return fwd_launch<3, 2>(x, w, s, y, stream);
    if (K == 5 && S == 1) return fwd_launch<5, 1>(x, w, s, y, stream);
    return fwd_launch<5, 2>(x, w, s, y, stream);
}

hipError_t dwconv_bwd_run(const float* dy, const float* x, const float* w, int B, int C, int H, int W, int K, int S,
                          float* dx, float* dw, void* ws, hipStream_t stream) {
    DwShape s{B, C, H, W, 0, 0};
    out_size(H, W, K, S, s.Ho, s.Wo);
    if (K == 3 && S == 1) return bwd_launch<3, 1>(dy, x, w, s, dx, dw, ws, stream);
    if (K == 3 && S == 2) return bwd_launch<3, 2>(dy, x, w, s, dx, dw, ws, stream);
    if (K == 5 && S == 1) return bwd_launch<5, 1>(dy, x, w, s, dx, dw, ws, stream);
    return bwd_launch<5, 2>(dy, x, w, s, dx, dw, ws, stream);
}

}  // namespace cabinet
