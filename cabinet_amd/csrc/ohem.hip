// f3 -- OHEM cross-entropy fused with the final bilinear upsample.
//
// Replaces, per head, reference src/models/cabinet.py:240-245 (F.interpolate of the H/8 logits to H x W,
// bilinear, align_corners=False) followed by src/utils/loss.py:51-80 (per-pixel CE, OHEM selection, mean)
// for the selection branch that training actually takes (at least n_min pixels above the threshold).
// The (B,C,H,W) full-resolution logits (268 MB per head at config 3), their log-softmax and the per-pixel
// gradient tensor are never materialised:
//   fwd   one thread per output pixel: sample the C logits from the four source taps, log-sum-exp,
//         loss = lse - x[label]; writes loss_px (B,H,W) and per-workgroup partials
//         (#valid, #(loss > thresh), sum of those losses)            -- ordered, deterministic
//   bwd   dlow = U^T G with G = coef * sel * (softmax - onehot), separable and in gather form:
//           T[b][c][oy][xs] = sum_ox wx(ox,xs) G[b][c][oy][ox]    (G recomputed on the fly, never stored)
//           dlow[b][c][ys][xs] = sum_oy wy(oy,ys) T[b][c][oy][xs]
//         no atomics -> bitwise reproducible (PyTorch's own upsample backward uses atomicAdd).
#include "common.hpp"

namespace cabinet {

constexpr int OHEM_MAXC = 32;

__device__ __forceinline__ void lin_taps(int dst, float scale, int in_size, int& i0, int& i1, float& lam) {
    const float src = fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
    i0 = min((int)src, in_size - 1);
    i1 = min(i0 + 1, in_size - 1);
    lam = src - (float)i0;
}

// logits of one output pixel: x[c] = bilinear sample of low[b][c] ; returns lse
__device__ __forceinline__ float sample_logits(const float* __restrict__ low_b, int C, size_t plane, int o00, int o01,
                                               int o10, int o11, float w00, float w01, float w10, float w11,
                                               float (&x)[OHEM_MAXC]) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < OHEM_MAXC; ++c) {
        if (c < C) {
            const float* p = low_b + (size_t)c * plane;
            x[c] = w00 * p[o00] + w01 * p[o01] + w10 * p[o10] + w11 * p[o11];
            mx = fmaxf(mx, x[c]);
        }
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < OHEM_MAXC; ++c)
        if (c < C) se += expf(x[c] - mx);
    return mx + logf(se);
}

__global__ __launch_bounds__(256) void ohem_up_fwd_kernel(const float* __restrict__ low, const long long* __restrict__ labels,
                                                           int C, int Hl, int Wl, int H, int W, float rh, float rw,
                                                           float thresh, int ignore_lb, float* __restrict__ loss_px,
                                                           float* __restrict__ blk_sum, int* __restrict__ blk_cnt) {
    __shared__ float s_f[4];
    __shared__ int s_i[2][4];
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x, P = H * W;
    float my_sum = 0.f;
    int my_valid = 0, my_above = 0;
    if (pix < P) {
        const int oy = pix / W, ox = pix - oy * W;
        const long long lb = labels[(size_t)b * P + pix];
        float loss = 0.f;
        if (lb != (long long)ignore_lb) {
            int y0, y1, x0, x1;
            float ly, lx;
            lin_taps(oy, rh, Hl, y0, y1, ly);
            lin_taps(ox, rw, Wl, x0, x1, lx);
            float x[OHEM_MAXC];
            const size_t plane = (size_t)Hl * Wl;
            const float lse = sample_logits(low + (size_t)b * C * plane, C, plane, y0 * Wl + x0, y0 * Wl + x1,
                                            y1 * Wl + x0, y1 * Wl + x1, (1.f - ly) * (1.f - lx), (1.f - ly) * lx,
                                            ly * (1.f - lx), ly * lx, x);
            float xl = 0.f;
#pragma unroll
            for (int c = 0; c < OHEM_MAXC; ++c)
                if (c < C && c == (int)lb) xl = x[c];
            loss = lse - xl;
            my_valid = 1;
            if (loss > thresh) {
                my_above = 1;
                my_sum = loss;
            }
        }
        loss_px[(size_t)b * P + pix] = loss;
    }
    // ordered block reduction
    my_sum = wave_sum(my_sum);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        my_valid += __shfl_xor(my_valid, o, 64);
        my_above += __shfl_xor(my_above, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_f[threadIdx.x >> 6] = my_sum;
        s_i[0][threadIdx.x >> 6] = my_valid;
        s_i[1][threadIdx.x >> 6] = my_above;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        blk_sum[blk] = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
        blk_cnt[2 * blk] = s_i[0][0] + s_i[0][1] + s_i[0][2] + s_i[0][3];
        blk_cnt[2 * blk + 1] = s_i[1][0] + s_i[1][1] + s_i[1][2] + s_i[1][3];
    }
}

// T[b][c][oy][xs] = sum_ox wx(ox, xs) * G_c(oy, ox),   G = coef * sel * (softmax - onehot)
__global__ __launch_bounds__(128) void ohem_up_bwd_x_kernel(const float* __restrict__ low, const long long* __restrict__ labels,
                                                             const float* __restrict__ loss_px, int C, int Hl, int Wl,
                                                             int H, int W, float rh, float rw, float thresh,
                                                             int ignore_lb, float coef, float* __restrict__ T) {
    const int b = blockIdx.z, oy = blockIdx.y;
    const int xs = blockIdx.x * 128 + threadIdx.x;
    if (xs >= Wl) return;
    const int P = H * W;
    const size_t plane = (size_t)Hl * Wl;
    const float* low_b = low + (size_t)b * C * plane;
    int y0, y1;
    float ly;
    lin_taps(oy, rh, Hl, y0, y1, ly);
    const int ox_lo = max(0, (int)floorf(((float)xs - 0.5f) / rw - 0.5f) - 1);
    const int ox_hi = min(W - 1, (int)ceilf(((float)xs + 1.5f) / rw - 0.5f) + 1);
    float acc[OHEM_MAXC];
#pragma unroll
    for (int c = 0; c < OHEM_MAXC; ++c) acc[c] = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        lin_taps(ox, rw, Wl, x0, x1, lx);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        if (wx == 0.f) continue;
        const size_t pix = (size_t)b * P + (size_t)oy * W + ox;
        const long long lb = labels[pix];
        if (lb == (long long)ignore_lb || !(loss_px[pix] > thresh)) continue;  // not selected
        float x[OHEM_MAXC];
        const float lse = sample_logits(low_b, C, plane, y0 * Wl + x0, y0 * Wl + x1, y1 * Wl + x0, y1 * Wl + x1,
                                        (1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx, x);
        const float wc = wx * coef;
#pragma unroll
        for (int c = 0; c < OHEM_MAXC; ++c)
            if (c < C) acc[c] += wc * (expf(x[c] - lse) - (c == (int)lb ? 1.f : 0.f));
    }
#pragma unroll
    for (int c = 0; c < OHEM_MAXC; ++c)
        if (c < C) T[(((size_t)b * C + c) * H + oy) * Wl + xs] = acc[c];
}

// dlow[b][c][ys][xs] = sum_oy wy(oy, ys) * T[b][c][oy][xs]
__global__ __launch_bounds__(256) void ohem_up_bwd_y_kernel(const float* __restrict__ T, int planes, int Hl, int Wl, int H,
                                                             float rh, float* __restrict__ dlow) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Hl * Wl) return;
    const int xs = idx % Wl, ys = (idx / Wl) % Hl, pl = idx / (Wl * Hl);
    const int oy_lo = max(0, (int)floorf(((float)ys - 0.5f) / rh - 0.5f) - 1);
    const int oy_hi = min(H - 1, (int)ceilf(((float)ys + 1.5f) / rh - 0.5f) + 1);
    const float* src = T + (size_t)pl * H * Wl + xs;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly;
        lin_taps(oy, rh, Hl, y0, y1, ly);
        acc += ((y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f)) * src[(size_t)oy * Wl];
    }
    dlow[idx] = acc;
}

int ohem_blocks(int B, int H, int W) { return B * ceil_div(H * W, 256); }

hipError_t ohem_up_fwd_run(const float* low, const long long* labels, int B, int C, int Hl, int Wl, int H, int W,
                           float thresh, int ignore_lb, float* loss_px, float* blk_sum, int* blk_cnt,
                           hipStream_t stream) {
    hipLaunchKernelGGL(ohem_up_fwd_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, stream, low, labels, C, Hl, Wl,
                       H, W, (float)Hl / (float)H, (float)Wl / (float)W, thresh, ignore_lb, loss_px, blk_sum, blk_cnt);
    return hipGetLastError();
}

size_t ohem_up_bwd_workspace(int B, int C, int H, int Wl) { return align_up((size_t)B * C * H * Wl * sizeof(float), 256); }

hipError_t ohem_up_bwd_run(const float* low, const long long* labels, const float* loss_px, int B, int C, int Hl,
                           int Wl, int H, int W, float thresh, int ignore_lb, float coef, float* dlow, void* ws,
                           hipStream_t stream) {
    float* T = static_cast<float*>(ws);
    hipLaunchKernelGGL(ohem_up_bwd_x_kernel, dim3(ceil_div(Wl, 128), H, B), dim3(128), 0, stream, low, labels, loss_px, C,
                       Hl, Wl, H, W, (float)Hl / (float)H, (float)Wl / (float)W, thresh, ignore_lb, coef, T);
    hipLaunchKernelGGL(ohem_up_bwd_y_kernel, dim3(ceil_div(B * C * Hl * Wl, 256)), dim3(256), 0, stream, T, B * C, Hl, Wl,
                       H, (float)Hl / (float)H, dlow);
    return hipGetLastError();
}

}  // namespace cabinet
