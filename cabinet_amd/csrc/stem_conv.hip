// K9 -- the 7x7 stride-2 stem convolution of the spatial branch (3 -> 64 channels, no bias), forward and weight
// gradient, NCHW fp32.
//
// Replaces the nn.Conv2d of `ConvBNReLU(3, 64, kernel_size=7, stride=2, padding=3)`, reference
// src/models/cabinet.py:111 (SpatialBranch.conv1; ConvBNReLU is SURVEY.md section 8 row a6's operator).  Its input is
// the image, so no input gradient exists.  MIOpen serves it with NHWC implicit-GEMM kernels plus layout transposes
// of the image, of the 537 MB output and of its gradient: ~3.2 ms of the config-3 step.
//
// As a GEMM: M = 64 output channels, N = pixels, K = 3*7*7 = 147 (padded to 148).  With only 3 input channels the
// im2col matrix is never worth building: a workgroup stages the input patch of its output tile in LDS (17 KB) and
// every lane gathers its B operand from it with a precomputed (ci,ky,kx) offset -- the "pixel on the lane" layout of
// v_mfma_f32_32x32x2_f32, so outputs leave as 128-byte rows of the NCHW tensor.  Weights live in LDS for the whole
// workgroup, which walks a strip of tiles (persistent over 16 tiles: one staging of the 38 KB weight image).
//   fwd : y[co][px]  = sum_k W[co][k] * patch[k][px]            A = W (LDS, stride 149), B = patch gather
//   wrw : dW[co][k]  = sum_px dy[co][px] * patch[k][px]         A = dy tile (LDS, stride 129), B = patch gather,
//         contraction over pixels; each wave accumulates the full 64 x 160 tile over its pixels, one ordered slab per
//         workgroup, final ordered slab sum (no atomics).
#include "common.hpp"

namespace cabinet {

constexpr int ST_K = 7, ST_S = 2, ST_PAD = 3, ST_CI = 3, ST_CO = 64;
constexpr int ST_KK = ST_CI * ST_K * ST_K;  // 147
constexpr int ST_KP = 148;                  // padded to a multiple of the MFMA k-step (2)
constexpr int ST_WLD = 149;                 // LDS row stride of the weight image (odd: conflict-free column reads)

// patch offset of contraction index k = (ci, ky, kx) in a [3][IH][IW] LDS patch (k >= 147: the zero-weight pad)
__host__ __device__ constexpr int stem_koff(int k, int IH, int IW) {
    return k < ST_KK ? ((k / (ST_K * ST_K)) * IH + (k % (ST_K * ST_K)) / ST_K) * IW + (k % ST_K) : 0;
}

struct StemShape {
    int B, H, W, Ho, Wo;
};

// ------------------------------------------------------------------------------------------------ forward
constexpr int SF_TH = 8, SF_TW = 32;                         // output tile of a workgroup step
constexpr int SF_IH = (SF_TH - 1) * ST_S + ST_K;             // 21
constexpr int SF_IW = (SF_TW - 1) * ST_S + ST_K;             // 69
constexpr int SF_PATCH = ST_CI * SF_IH * SF_IW;              // 4347 floats

__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                             StemShape s, int tiles_x, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                          // [64][149]
    float* xs = wl + ST_CO * ST_WLD;           // [3][21][69]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int strips_y = (s.Ho + SF_TH - 1) / SF_TH;
    const int b = blockIdx.x / strips_y, oy0 = (blockIdx.x - b * strips_y) * SF_TH;
    for (int i = tid; i < ST_CO * ST_KP; i += 256) {
        const int co = i / ST_KP, k = i - co * ST_KP;
        wl[co * ST_WLD + k] = k < ST_KK ? wgt[co * ST_KK + k] : 0.f;
    }
    const float* xb = x + (size_t)b * ST_CI * s.H * s.W;
    const int iy0 = oy0 * ST_S - ST_PAD;
    // the two output rows of this wave, pixel li of each: patch offsets of their top-left taps
    const int poff0 = (ST_S * (2 * wave)) * SF_IW + ST_S * li, poff1 = poff0 + ST_S * SF_IW;
    for (int t = 0; t < tiles_x; ++t) {
        const int ox0 = t * SF_TW, ix0 = ox0 * ST_S - ST_PAD;
        __syncthreads();  // previous tile's readers are done with xs (and, first time, wl / kt are being written)
        for (int i = tid; i < SF_PATCH; i += 256) {
            const int ci = i / (SF_IH * SF_IW), r = i - ci * SF_IH * SF_IW, yy = r / SF_IW, xx = r - yy * SF_IW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            xs[i] = (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) ? xb[((size_t)ci * s.H + iy) * s.W + ix] : 0.f;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* wrow = wl + li * ST_WLD + h;
        const float* xp0 = xs + poff0;
        const float* xp1 = xs + poff1;
#pragma unroll  // fully unrolled: the (ci,ky,kx) patch offsets of both lane halves are compile-time constants
        for (int k0 = 0; k0 < ST_KP; k0 += 2) {
            const int ko = h ? stem_koff(k0 + 1, SF_IH, SF_IW) : stem_koff(k0, SF_IH, SF_IW);
            const float a0 = wrow[k0], a1 = wrow[32 * ST_WLD + k0];
            const float b0 = xp0[ko], b1 = xp1[ko];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
        const int ox = ox0 + li;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = oy0 + 2 * wave + j;
            if (oy < s.Ho && ox < s.Wo) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = 32 * i + acc_row(r) + 4 * h;
                        y[(((size_t)b * ST_CO + co) * s.Ho + oy) * s.Wo + ox] = acc[i][j][r];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int SW_TH = 4, SW_TW = 32;                         // 128 pixels per step, 32 per wave
constexpr int SW_IH = (SW_TH - 1) * ST_S + ST_K;             // 13
constexpr int SW_IW = (SW_TW - 1) * ST_S + ST_K;             // 69
constexpr int SW_PATCH = ST_CI * SW_IH * SW_IW;              // 2691 floats
constexpr int SW_DLD = SW_TH * SW_TW + 1;                    // 129: row stride of the dy tile
constexpr int SW_KB = 5;                                     // 5 x 32 = 160 >= 147 columns

__global__ __launch_bounds__(256, 2) void stem_conv_wrw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             StemShape s, int tiles_x, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;                          // [64][129]
    float* xs = dys + ST_CO * SW_DLD;           // [3][13][69]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int strips_y = (s.Ho + SW_TH - 1) / SW_TH;
    const int b = blockIdx.x / strips_y, oy0 = (blockIdx.x - b * strips_y) * SW_TH;
    int ko[SW_KB];  // patch offset of this lane's column k = 32*kb + li (columns >= 147 read offset 0 and are dropped)
#pragma unroll
    for (int kb = 0; kb < SW_KB; ++kb) {
        const int k = 32 * kb + li, kk = k < ST_KK ? k : 0;
        const int ci = kk / (ST_K * ST_K), r = kk - ci * ST_K * ST_K, ky = r / ST_K, kx = r - ky * ST_K;
        ko[kb] = (ci * SW_IH + ky) * SW_IW + kx;
    }
    f32x16 acc[2][SW_KB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < SW_KB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* xb = x + (size_t)b * ST_CI * s.H * s.W;
    const float* dyb = dy + (size_t)b * ST_CO * s.Ho * s.Wo;
    const int iy0 = oy0 * ST_S - ST_PAD;
    for (int t = 0; t < tiles_x; ++t) {
        const int ox0 = t * SW_TW, ix0 = ox0 * ST_S - ST_PAD;
        __syncthreads();
        for (int i = tid; i < SW_PATCH; i += 256) {
            const int ci = i / (SW_IH * SW_IW), r = i - ci * SW_IH * SW_IW, yy = r / SW_IW, xx = r - yy * SW_IW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            xs[i] = (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) ? xb[((size_t)ci * s.H + iy) * s.W + ix] : 0.f;
        }
        if ((s.Wo & 3) == 0 && ox0 + SW_TW <= s.Wo) {  // full-width tile: 128-bit row loads
            for (int i = tid; i < ST_CO * SW_TH * (SW_TW / 4); i += 256) {
                const int co = i / (SW_TH * (SW_TW / 4)), q = i - co * (SW_TH * (SW_TW / 4)), r = q / (SW_TW / 4),
                          c = (q - r * (SW_TW / 4)) * 4;
                const int oy = oy0 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (oy < s.Ho) v = *reinterpret_cast<const f32x4*>(dyb + ((size_t)co * s.Ho + oy) * s.Wo + ox0 + c);
                float* d = dys + co * SW_DLD + r * SW_TW + c;
                d[0] = v[0], d[1] = v[1], d[2] = v[2], d[3] = v[3];
            }
        } else {
            for (int i = tid; i < ST_CO * SW_TH * SW_TW; i += 256) {
                const int co = i / (SW_TH * SW_TW), p = i - co * (SW_TH * SW_TW), r = p / SW_TW, c = p - r * SW_TW;
                const int oy = oy0 + r, ox = ox0 + c;
                dys[co * SW_DLD + p] = (oy < s.Ho && ox < s.Wo) ? dyb[((size_t)co * s.Ho + oy) * s.Wo + ox] : 0.f;
            }
        }
        __syncthreads();
        // this wave contracts over pixels p = 32*wave .. +31 (output row `wave` of the tile), two per MFMA
        const float* drow = dys + li * SW_DLD + 32 * wave + h;
        const int prow = (ST_S * wave) * SW_IW;
#pragma unroll 2
        for (int c0 = 0; c0 < 32; c0 += 2) {
            const float a0 = drow[c0], a1 = drow[32 * SW_DLD + c0];
            const int poff = prow + ST_S * (c0 + h);
#pragma unroll
            for (int kb = 0; kb < SW_KB; ++kb) {
                const float bv = xs[ko[kb] + poff];
                acc[0][kb] = mfma32(a0, bv, acc[0][kb]);
                acc[1][kb] = mfma32(a1, bv, acc[1][kb]);
            }
        }
    }
    // ordered cross-wave sum through LDS, then this workgroup's slab [64][147]
    __syncthreads();
    float* red = smem;  // [64][161] reused (fits: 64*161 <= 64*129 + 2691)
    constexpr int RLD = 161;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kb = 0; kb < SW_KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = 32 * i + acc_row(r) + 4 * h, k = 32 * kb + li;
                        float* p = red + co * RLD + k;
                        *p = (w == 0) ? acc[i][kb][r] : *p + acc[i][kb][r];
                    }
        }
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * ST_CO * ST_KK;
    for (int i = tid; i < ST_CO * ST_KK; i += 256) {
        const int co = i / ST_KK, k = i - co * ST_KK;
        slab[i] = red[co * RLD + k];
    }
}

__global__ void stem_conv_wrw_reduce_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ST_CO * ST_KK) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nslab; k += 4) {
        s0 += slabs[(size_t)k * ST_CO * ST_KK + i];
        s1 += slabs[(size_t)(k + 1) * ST_CO * ST_KK + i];
        s2 += slabs[(size_t)(k + 2) * ST_CO * ST_KK + i];
        s3 += slabs[(size_t)(k + 3) * ST_CO * ST_KK + i];
    }
    for (; k < nslab; ++k) s0 += slabs[(size_t)k * ST_CO * ST_KK + i];
    dw[i] = (s0 + s1) + (s2 + s3);
}

static StemShape stem_shape(int B, int H, int W) {
    return StemShape{B, H, W, (H + 2 * ST_PAD - ST_K) / ST_S + 1, (W + 2 * ST_PAD - ST_K) / ST_S + 1};
}

size_t stem_conv_wrw_workspace(int B, int H, int W) {
    const StemShape s = stem_shape(B, H, W);
    return align_up((size_t)B * ceil_div(s.Ho, SW_TH) * ST_CO * ST_KK * sizeof(float), 256);
}

hipError_t stem_conv_fwd_run(const float* x, const float* w, int B, int H, int W, float* y, hipStream_t stream) {
    const StemShape s = stem_shape(B, H, W);
    const size_t lds = ((size_t)ST_CO * ST_WLD + SF_PATCH) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(B * ceil_div(s.Ho, SF_TH)), dim3(256), lds, stream, x, w, s,
                       ceil_div(s.Wo, SF_TW), y);
    return hipGetLastError();
}

hipError_t stem_conv_wrw_run(const float* dy, const float* x, int B, int H, int W, float* dw, void* ws,
                             hipStream_t stream) {
    const StemShape s = stem_shape(B, H, W);
    const int nslab = B * ceil_div(s.Ho, SW_TH);
    const size_t lds = ((size_t)ST_CO * SW_DLD + SW_PATCH) * sizeof(float);
    float* slabs = static_cast<float*>(ws);
    hipLaunchKernelGGL(stem_conv_wrw_kernel, dim3(nslab), dim3(256), lds, stream, dy, x, s, ceil_div(s.Wo, SW_TW), slabs);
    hipLaunchKernelGGL(stem_conv_wrw_reduce_kernel, dim3(ceil_div(ST_CO * ST_KK, 256)), dim3(256), 0, stream, slabs, nslab,
                       dw);
    return hipGetLastError();
}

}  // namespace cabinet
