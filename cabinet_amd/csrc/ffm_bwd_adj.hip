// K4 (round 4) -- the adjoint of the x4 bilinear resize taken BEFORE the BatchNorm backward coefficients exist.
//
// Reference span: the autograd backward of F.interpolate(..., mode="bilinear", align_corners=False) at
// src/models/cabinet.py:228-230 (feat_cp8 -> feat_sp's size) followed by ConvBNReLU (cabinet.py:143-144) and the gate of
// FeatureFusionModule.forward (cabinet.py:150-160), with the resize commuted behind the 1x1 convolution:
// the low-resolution operand of the backward is  dz_low = U^T dz,  U = the resize,
//     dz = gi (mask (g a1 + a2) - mdy - xhat mdyx),   gi = gamma invstd,  mask = [BN(z) > 0],  xhat = (z - mean) invstd,
// where a1, a2 (per image and channel: the gate and the squeeze-excite branch) and mdy, mdyx (per channel: the batch means of
// dy and dy xhat) are only known after a full pass over g = dout and z.  Rounds 2-4 therefore read both tensors a second time
// (upsample_adjoint_kernel<true>: 268 MB in, dz 134 MB out, 92 us at 4.4 TB/s -- HBM-bound, nothing left in it).
// U^T is linear and the coefficients are constant over a plane, so
//     U^T dz = gi (a1 U^T(mask g) + a2 U^T(mask) - mdy U^T(1) - mdyx U^T(xhat)):
// the three adjoint fields T1 = U^T(mask g), T2 = U^T(mask), T3 = U^T(xhat) need no coefficient and are taken in the SAME pass
// that forms the five BatchNorm sums (ffm_bwd_reduce_adj_kernel: 268 MB in, 3 x 8 MB out), U^T(1) is a product of two 1-D
// weight sums (= 16), and dz_low is an elementwise pass over 8 MB (ffm_bwd_dzl_kernel).  The full-resolution dz is never written:
// ffm_bwd_fused.hip forms it from g and z while it stages a chunk.
// xhat, not z, goes through U^T: the combination mdyx (T3) would otherwise cancel (mean / std) ulps.
//
// One workgroup per (image, channel) plane, bands of 32 output rows:
//   phase A  a thread takes four consecutive pixels = the footprint of ONE source column m: centre sum C (weights 5/8 7/8 7/8
//            5/8) and the halos L (3/8 1/8 -> column m - 1), R (1/8 3/8 -> m + 1); neighbours swap halos with a lane
//            shuffle, a halo that would leave the row is folded back (= the clamp of align_corners=False: weights sum to 1);
//            the horizontally reduced row goes to LDS;
//   phase B  a thread owns (source row k, column) and adds the <= 8 output rows of the band with weight on k; a source row
//            that straddles two bands is carried through LDS (double-buffered by band parity: one barrier per band).
// The pixel -> thread assignment and order of the five sums are those of ffm_bwd_reduce_kernel (for Wl >= 8 and H a multiple of
// 32): the sums are bit-identical.
#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int RA_ROWS = 32;   // output rows per band = 8 source rows

template <int LW>   // log2(Wl); W = 4 Wl
__global__ __launch_bounds__(256) void ffm_bwd_reduce_adj_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ bn_w,
                                                                  const float* __restrict__ bn_b, float* __restrict__ sums,
                                                                  float* __restrict__ t1, float* __restrict__ t2,
                                                                  float* __restrict__ t3, int C, int H, int Hl) {
    constexpr int Wl = 1 << LW, W = 4 * Wl;
    constexpr int IT = (RA_ROWS * Wl + 255) / 256;   // pixel quads per thread and band
    __shared__ float s_red[4];
    __shared__ float hrow[2][3][RA_ROWS][Wl];
    __shared__ float carry[2][3][2][Wl];
    const int row = blockIdx.x, c = row % C, tid = threadIdx.x;
    const float mu = mean[c], is = invstd[c], gw = bn_w[c], gb = bn_b[c];
    const float* zr = z + (size_t)row * H * W;
    const float* gr = g + (size_t)row * H * W;
    const float rh = (float)Hl / (float)H;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f;
    f32x4 zq[IT], gq[IT];
    auto load_band = [&](int r0) {
        const int nq = min(RA_ROWS, H - r0) * Wl;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int qi = tid + 256 * it;
            if (qi < nq) {
                zq[it] = *reinterpret_cast<const f32x4*>(zr + (size_t)r0 * W + 4 * qi);
                gq[it] = *reinterpret_cast<const f32x4*>(gr + (size_t)r0 * W + 4 * qi);
            }
        }
    };
    const int nbands = (H + RA_ROWS - 1) / RA_ROWS;
    load_band(0);
    for (int j = 0; j < nbands; ++j) {
        const int r0 = j * RA_ROWS, rows = min(RA_ROWS, H - r0), r1 = r0 + rows, buf = j & 1;
        // ---- phase A: the five sums and the horizontal adjoint of the three fields
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int qi = tid + 256 * it, rr = qi >> LW, m = qi & (Wl - 1);
            const bool valid = qi < rows * Wl;
            float f1[4], f2[4], f3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f1[e] = f2[e] = f3[e] = 0.f;
                if (valid) {
                    const float zv = zq[it][e], gv = gq[it][e];
                    const float xh = (zv - mu) * is;
                    const float y = fmaf(xh, gw, gb);
                    f3[e] = xh;
                    if (y > 0.f) {
                        s1 += gv * y;
                        s2 += gv;
                        s3 += gv * xh;
                        s4 += 1.f;
                        s5 += xh;
                        f1[e] = gv, f2[e] = 1.f;
                    }
                }
            }
            auto horizontal = [&](const float* d) {
                const float ctr = (0.625f * d[0] + 0.875f * d[1]) + (0.875f * d[2] + 0.625f * d[3]);
                const float lft = 0.375f * d[0] + 0.125f * d[1], rgt = 0.125f * d[2] + 0.375f * d[3];
                const float from_right = __shfl_down(lft, 1, 64), from_left = __shfl_up(rgt, 1, 64);
                return ctr + (m == 0 ? lft : from_left) + (m == Wl - 1 ? rgt : from_right);
            };
            const float h1 = horizontal(f1), h2 = horizontal(f2), h3 = horizontal(f3);
            if (rr < RA_ROWS) hrow[buf][0][rr][m] = h1, hrow[buf][1][rr][m] = h2, hrow[buf][2][rr][m] = h3;
        }
        if (j + 1 < nbands) load_band(r0 + RA_ROWS);   // in flight under the vertical pass
        __syncthreads();
        // ---- phase B: the vertical adjoint; source rows 8 j - 1 .. 8 j + 8 receive from this band
        for (int o = tid; o < 10 * Wl; o += 256) {
            const int s = o >> LW, col = o & (Wl - 1), k = 8 * j - 1 + s;
            if (k < 0 || k >= Hl) continue;
            const int ra = max(r0, 4 * k - 2), rb = min(r1 - 1, 4 * k + 5);
            float p1 = 0.f, p2 = 0.f, p3 = 0.f;
            for (int r = ra; r <= rb; ++r) {
                int y0, y1;
                float ly;
                bilinear_taps(r, rh, Hl, y0, y1, ly);
                const float wgt = (y0 == k ? 1.f - ly : 0.f) + (y1 == k ? ly : 0.f);
                p1 = fmaf(wgt, hrow[buf][0][r - r0][col], p1);
                p2 = fmaf(wgt, hrow[buf][1][r - r0][col], p2);
                p3 = fmaf(wgt, hrow[buf][2][r - r0][col], p3);
            }
            if (max(4 * k - 2, 0) < r0) {   // rows of the previous band contributed (s is 0 or 1)
                p1 += carry[buf ^ 1][0][s][col], p2 += carry[buf ^ 1][1][s][col], p3 += carry[buf ^ 1][2][s][col];
            }
            if (min(4 * k + 5, H - 1) < r1) {
                const size_t o_lo = ((size_t)row * Hl + k) * Wl + col;
                t1[o_lo] = p1, t2[o_lo] = p2, t3[o_lo] = p3;
            } else {   // the next band finishes it (k is 8 j + 7 or 8 j + 8 = its slots 0 and 1)
                const int ci = k - (8 * j + 7);
                carry[buf][0][ci][col] = p1, carry[buf][1][ci][col] = p2, carry[buf][2][ci][col] = p3;
            }
        }
    }
    s1 = block_sum_256(s1, s_red);
    s2 = block_sum_256(s2, s_red);
    s3 = block_sum_256(s3, s_red);
    s4 = block_sum_256(s4, s_red);
    s5 = block_sum_256(s5, s_red);
    if (tid == 0) {
        float* o = sums + (size_t)row * 5;
        o[0] = s1, o[1] = s2, o[2] = s3, o[3] = s4, o[4] = s5;
    }
}

// dz_low = gi (a1 T1 + a2 T2 - mdy U^T(1) - mdyx T3),  U^T(1) = 16 everywhere: every output pixel spreads weight 1 over its
// taps and the fold at the borders keeps it in the plane, so along one axis source index k receives
// 1/8 + 3/8 + 5/8 + 7/8 + 7/8 + 5/8 + 3/8 + 1/8 = 4 (at k = 0: 1 + 1 + 7/8 + 5/8 + 3/8 + 1/8), exactly, in any order of summation.
// One workgroup per quarter-KB of a plane; four positions per thread.
__global__ __launch_bounds__(256) void ffm_bwd_dzl_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                                           const float* __restrict__ t3, const float* __restrict__ invstd,
                                                           const float* __restrict__ bn_w, const float* __restrict__ coef_a1,
                                                           const float* __restrict__ coef_a2,
                                                           const float* __restrict__ mean_dy,
                                                           const float* __restrict__ mean_dyx, float* __restrict__ dzl, int C,
                                                           int Pl) {
    const int row = blockIdx.y, c = row % C;
    const float gi = bn_w[c] * invstd[c], a1 = coef_a1[row], a2 = coef_a2[row], m16 = mean_dy[c] * 16.f, mx = mean_dyx[c];
    const size_t base = (size_t)row * Pl;
    for (int q = (blockIdx.x * 256 + threadIdx.x) * 4; q < Pl; q += gridDim.x * 1024) {
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(t1 + base + q), v2 = *reinterpret_cast<const f32x4*>(t2 + base + q);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(t3 + base + q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gi * (fmaf(a1, v1[e], a2 * v2[e]) - m16 - mx * v3[e]);
        *reinterpret_cast<f32x4*>(dzl + base + q) = o;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
bool ffm_bwd_adj_supported(int H, int W, int Hl, int Wl) {
    return Hl > 0 && Wl >= 4 && Wl <= 64 && (Wl & (Wl - 1)) == 0 && H == 4 * Hl && W == 4 * Wl;
}

// the reduction pass of the FFM backward (five sums per plane) with the three adjoint fields; t: 3 x planes x Hl x Wl floats
hipError_t ffm_bwd_reduce_adj_run(const float* g, const float* z, const float* mean, const float* invstd, const float* bn_w,
                                  const float* bn_b, float* sums, float* t, int planes, int C, int H, int Hl, int Wl,
                                  hipStream_t stream) {
    const size_t n = (size_t)planes * Hl * Wl;
    float *t1 = t, *t2 = t + n, *t3 = t + 2 * n;
#define CABINET_RA(LW_)                                                                                                      \
    hipLaunchKernelGGL(ffm_bwd_reduce_adj_kernel<LW_>, dim3(planes), dim3(256), 0, stream, g, z, mean, invstd, bn_w, bn_b,    \
                       sums, t1, t2, t3, C, H, Hl)
    switch (Wl) {
        case 4: CABINET_RA(2); break;
        case 8: CABINET_RA(3); break;
        case 16: CABINET_RA(4); break;
        case 32: CABINET_RA(5); break;
        case 64: CABINET_RA(6); break;
        default: return hipErrorInvalidValue;
    }
#undef CABINET_RA
    return hipGetLastError();
}

hipError_t ffm_bwd_dzl_run(const float* t, const float* invstd, const float* bn_w, const float* a1, const float* a2,
                           const float* mdy, const float* mdyx, float* dzl, int planes, int C, int Hl, int Wl, hipStream_t stream) {
    const int Pl = Hl * Wl;   // a multiple of 4 (Wl is)
    const size_t n = (size_t)planes * Pl;
    hipLaunchKernelGGL(ffm_bwd_dzl_kernel, dim3(min(ceil_div(Pl, 1024), 64), planes), dim3(256), 0, stream, t, t + n, t + 2 * n,
                       invstd, bn_w, a1, a2, mdy, mdyx, dzl, C, Pl);
    return hipGetLastError();
}

}  // namespace cabinet
