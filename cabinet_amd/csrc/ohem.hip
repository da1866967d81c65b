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
#include <stdint.h>
#include <stdlib.h>

#include "common.hpp"

namespace cabinet {



// One workgroup per output row (b, oy).  The two source rows are lerped vertically into LDS once (coalesced
// reads), so a pixel's C logits cost 2 LDS reads each instead of 4 scattered global loads.
// CMAX: classes rounded up, the per-class loops are fully unrolled.  EXACT (C == CMAX: 8 and 19 classes have their own
// instances) drops the `c < C` predicates: with a run-time C the compiler turns every class into its own uniform branch
// (LDS read -> wait -> arithmetic, one class at a time); without them the 2 C LDS reads of a pixel are in flight together.
template <int CMAX, bool EXACT>
__global__ __launch_bounds__(256) void ohem_up_fwd_kernel(const float* __restrict__ low, const long long* __restrict__ labels,
                                                           int C, int Hl, int Wl, int H, int W, float rh, float rw,
                                                           float thresh, int ignore_lb, float* __restrict__ loss_px,
                                                           float* __restrict__ blk_sum, int* __restrict__ blk_cnt) {
    extern __shared__ __attribute__((aligned(16))) float v[];  // [C][Wl]
    __shared__ float s_f[4];
    __shared__ int s_i[3][4];
    const int b = blockIdx.y, oy = blockIdx.x, P = H * W;
    const size_t plane = (size_t)Hl * Wl;
    const float* low_b = low + (size_t)b * C * plane;
    int y0, y1;
    float ly;
    bilinear_taps(oy, rh, Hl, y0, y1, ly);
    for (int i = threadIdx.x; i < C * Wl; i += 256) {
        const int c = i / Wl, xs = i - c * Wl;
        const float* p = low_b + (size_t)c * plane;
        v[i] = (1.f - ly) * p[y0 * Wl + xs] + ly * p[y1 * Wl + xs];
    }
    __syncthreads();
    float my_sum = 0.f;
    int my_valid = 0, my_above = 0, my_bad = 0;
    for (int ox = threadIdx.x; ox < W; ox += 256) {
        const size_t pix = (size_t)b * P + (size_t)oy * W + ox;
        const long long lb = labels[pix];
        float loss = 0.f;
        if (lb != (long long)ignore_lb) {
            int x0, x1;
            float lx;
            bilinear_taps(ox, rw, Wl, x0, x1, lx);
            float x[CMAX], mx = -INFINITY, xl = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) {
                    x[c] = (1.f - lx) * v[c * Wl + x0] + lx * v[c * Wl + x1];
                    mx = fmaxf(mx, x[c]);
                    if (c == (int)lb) xl = x[c];
                }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) se += fast_exp2((x[c] - mx) * LOG2E_F);  // v_exp_f32: arguments <= 0, 1 ulp
            loss = mx + fast_log2(se) * LN2_F - xl;
            // a label outside [0, C) that is not ignore_lb is an error (F.cross_entropy asserts on it): a separate flag,
            // OR-reduced over the block, poisons the block's valid count so the caller's one host read sees it (an
            // arithmetic sentinel summed over a wide row could wrap)
            my_valid += 1;
            my_bad |= (lb < 0 || lb >= (long long)C) ? 1 : 0;
            if (loss > thresh) {
                my_above += 1;
                my_sum += loss;
            }
        }
        loss_px[pix] = loss;
    }
    // ordered block reduction
    my_sum = wave_sum(my_sum);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        my_valid += __shfl_xor(my_valid, o, 64);
        my_above += __shfl_xor(my_above, o, 64);
        my_bad |= __shfl_xor(my_bad, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_f[threadIdx.x >> 6] = my_sum;
        s_i[0][threadIdx.x >> 6] = my_valid;
        s_i[1][threadIdx.x >> 6] = my_above;
        s_i[2][threadIdx.x >> 6] = my_bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        blk_sum[blk] = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
        const int nv = s_i[0][0] + s_i[0][1] + s_i[0][2] + s_i[0][3];
        const int bad = s_i[2][0] | s_i[2][1] | s_i[2][2] | s_i[2][3];
        blk_cnt[2 * blk] = bad ? -(1 << 30) : nv;  // < 0: the block saw an out-of-range label
        blk_cnt[2 * blk + 1] = s_i[1][0] + s_i[1][1] + s_i[1][2] + s_i[1][3];
    }
}

// The model's x8 resize (W == 8 Wl): a thread owns the eight output pixels [8g, 8g+8) of its row.  They read only the source
// columns g-1, g, g+1 (clamped at the borders, which reproduces the clamped source index of align_corners=False), so the
// 3 C staged values are fetched once for eight pixels instead of 2 C per pixel, the interpolation weights are the
// constants (j + 4.5)/8 and (j - 3.5)/8, and labels / losses move as 16-byte vectors.
// NH = 2: BOTH loss heads of the step (reference train.py:435: criteria_p(out, lb) + criteria_16(out16, lb)) in one launch.
// The heads share the label tile (read once) and every thread carries two independent exp / log dependency chains, which is
// what this VALU-latency-bound kernel lacks with one head (~1000 VALU instructions per wave, v_exp_f32 / v_log_f32 chains).
template <int NH>
struct OhemFwdHeads {
    const float* low[NH];   // (B,C,Hl,Wl) each
    float* loss_px[NH];     // (B,H,W)
    float* blk_sum[NH];     // (nblk)
    int* blk_cnt[NH];       // (nblk,2)
};

template <int CMAX, bool EXACT, int NH>
__global__ __launch_bounds__(256) void ohem_up_fwd_x8_kernel(OhemFwdHeads<NH> hd, const long long* __restrict__ labels, int C,
                                                              int Hl, int Wl, int H, int W, float rh, float thresh,
                                                              int ignore_lb) {
    extern __shared__ __attribute__((aligned(16))) float v[];  // [NH][C][Wl]
    __shared__ float s_f[NH][4];
    __shared__ int s_i[NH][3][4];
    // Grid (H, B) decoded XCD-aware (round 5): the dispatcher deals blocks round-robin over the 8 XCDs, so with
    // oy = blockIdx.x the eight output rows that lerp the SAME two source rows ran on eight different XCDs and every private L2
    // fetched those rows for itself (FETCH_SIZE x2 = 134 MB for 75 MB of inputs at config 3).  Each XCD now walks a contiguous
    // chunk of the (image, output row) list: the source-row pair of a run of 8 rows is fetched into ONE L2.  `tile` = b H + oy
    // also indexes the per-block partials, so cabinet_ohem_stats adds them in the order it always did.  Speed only.
    const int tile = xcd_chunked_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = tile / H, oy = tile - b * H, P = H * W, nt = blockDim.x, CW = C * Wl;
    const size_t plane = (size_t)Hl * Wl;
    int y0, y1;
    float ly;
    bilinear_taps(oy, rh, Hl, y0, y1, ly);
    if ((Wl & 3) == 0) {
        // the rows of v are contiguous in the source (class-major planes, Wl floats per row): all 16-byte loads of a thread are
        // independent and issued before the first lerp (a scalar loop made this 16 dependent L2 round trips per thread)
        const int q4 = CW >> 2;  // float4 slots per head
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const float* low_b = hd.low[hh] + (size_t)b * C * plane;
            for (int i0 = threadIdx.x; i0 < q4; i0 += 4 * nt) {
                f32x4 a0[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(i0 + u * nt, q4 - 1), c = (4 * i) / Wl, xs = 4 * i - c * Wl;
                    const float* p = low_b + (size_t)c * plane + xs;
                    a0[u] = *reinterpret_cast<const f32x4*>(p + y0 * Wl);
                    a1[u] = *reinterpret_cast<const f32x4*>(p + y1 * Wl);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * nt < q4) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = ((1.f - ly) * a0[u][e] + ly * a1[u][e]) * LOG2E_F;
                        *reinterpret_cast<f32x4*>(v + hh * CW + 4 * (i0 + u * nt)) = o;
                    }
            }
        }
    } else {
        const float inv_wl = 1.f / (float)Wl;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const float* low_b = hd.low[hh] + (size_t)b * C * plane;
            for (int i = threadIdx.x; i < CW; i += nt) {
                const int c = (int)(((float)i + 0.5f) * inv_wl), xs = i - c * Wl;
                const float* p = low_b + (size_t)c * plane;
                v[hh * CW + i] = ((1.f - ly) * p[y0 * Wl + xs] + ly * p[y1 * Wl + xs]) * LOG2E_F;
            }
        }
    }
    __syncthreads();
    float my_sum[NH];
    int my_valid = 0, my_bad = 0, my_above[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) my_sum[hh] = 0.f, my_above[hh] = 0;
    for (int g = threadIdx.x; g < Wl; g += nt) {
        const size_t pix = (size_t)b * P + (size_t)oy * W + 8 * g;  // 64-byte aligned labels, 32-byte aligned losses
        long long lb[8];
        {
            const longlong2* lp = reinterpret_cast<const longlong2*>(labels + pix);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const longlong2 t = lp[u];
                lb[2 * u] = t.x, lb[2 * u + 1] = t.y;
            }
        }
        // v holds the vertically interpolated logits times log2(e).  This kernel is VALU-issue bound (1823 VALU instructions
        // per wave measured for two heads: 81 % of the SIMD cycles), so the per-pixel chain is kept short: the horizontal
        // lerp is one fma on a difference formed once per source interval, exp2 needs no scaling multiply, and the label's
        // logit is gathered from LDS (rows are Wl floats apart and lanes walk consecutive columns: conflict-free whatever the
        // labels) instead of a compare + select per class.
        // Round 4: the log-sum-exp of all eight pixels is shifted by ONE bound per interval instead of a maximum per pixel:
        // every upsampled logit is a convex combination of the interval's taps, so M = max over classes and taps bounds them
        // all (3 C maxima per interval instead of 8 C), and with xc - M staged once the per-pixel-and-class work is
        // fma + exp2 + add (it was fma + max + sub + exp2 + add).  M exceeds a pixel's own maximum by at most the spread of
        // neighbouring source columns: normally exp2 of a few tens below zero; when that spread passes ~100 the shifted sum
        // underflows and the pixel is redone with its own maximum (the `se < 1e-30f` branch below).
        // (Measured and not kept, here and in the backward's x pass: two pixels per step as packed 2-vectors -- v_pk_fma_f32 /
        // v_pk_add_f32 halve the non-transcendental issue slots, and the time did not move: 75.7 vs 74.5 us, 72.2 vs 68.8 us.
        // These kernels are bound by v_exp_f32 (quarter rate) and by the load -> barrier start of their short workgroups.)
        float xc[NH][CMAX], dm[NH][CMAX], dp[NH][CMAX], big[NH];
        const int gm = max(g - 1, 0), gp = min(g + 1, Wl - 1);
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            float m3 = -INFINITY;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) {
                    const float* vc = v + hh * CW + c * Wl;
                    const float a = vc[gm], bq = vc[g], cq = vc[gp];
                    xc[hh][c] = bq, dm[hh][c] = bq - a, dp[hh][c] = cq - bq;
                    m3 = fmaxf(m3, fmaxf(a, fmaxf(bq, cq)));
                }
            big[hh] = m3;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) xc[hh][c] -= m3;
        }
        float out[NH][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // pixel j of the interval: x = xm + t (xc - xm) = xc - (1 - t)(xc - xm), t = (j + 4.5)/8, for j < 4;
            //                          x = xc + t (xp - xc),                          t = (j - 3.5)/8, for j >= 4
            const float t = j < 4 ? -(1.f - ((float)j + 4.5f) * 0.125f) : ((float)j - 3.5f) * 0.125f;
            const bool valid = lb[j] != (long long)ignore_lb;
            const bool inrange = lb[j] >= 0 && lb[j] < (long long)C;
            // a label outside [0, C) that is not ignore_lb is an error (F.cross_entropy asserts on it): a separate flag,
            // OR-reduced over the block, poisons the block's valid count so the caller's one host read sees it
            my_valid += valid ? 1 : 0;
            my_bad |= (valid && !inrange) ? 1 : 0;
            const int lrow = inrange ? (int)lb[j] * Wl : 0;
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                float loss = 0.f;
                if (valid) {
                    const float* vl = v + hh * CW + lrow;
                    const float xl = j < 4 ? fmaf(t, vl[g] - vl[gm], vl[g]) : fmaf(t, vl[gp] - vl[g], vl[g]);
                    float se = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (EXACT || c < C) se += fast_exp2(fmaf(t, j < 4 ? dm[hh][c] : dp[hh][c], xc[hh][c]));
                    float lse = fast_log2(se);
                    if (se < 1e-30f) {
                        // The interval bound M towers more than ~100 (log2 units: 70 nats) above every logit of THIS pixel -- a
                        // spike in a neighbouring source column -- and the shifted sum has underflowed (se == 0 would make the
                        // loss -inf and drop the pixel from the OHEM statistics; F.cross_entropy is exact for any finite logits).
                        // Rare and divergent: redo the pixel with its own maximum.  
                        float pm = -INFINITY;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (EXACT || c < C) pm = fmaxf(pm, fmaf(t, j < 4 ? dm[hh][c] : dp[hh][c], xc[hh][c]));
                        float s2 = 0.f;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (EXACT || c < C) s2 += fast_exp2(fmaf(t, j < 4 ? dm[hh][c] : dp[hh][c], xc[hh][c]) - pm);
                        lse = pm + fast_log2(s2);
                    }
                    loss = ((big[hh] - xl) + lse) * LN2_F;
                    if (loss > thresh) {
                        my_above[hh] += 1;
                        my_sum[hh] += loss;
                    }
                }
                out[hh][j] = loss;
            }
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            f32x4* op = reinterpret_cast<f32x4*>(hd.loss_px[hh] + pix);
            op[0] = f32x4{out[hh][0], out[hh][1], out[hh][2], out[hh][3]};
            op[1] = f32x4{out[hh][4], out[hh][5], out[hh][6], out[hh][7]};
        }
    }
    // ordered block reduction
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) my_sum[hh] = wave_sum(my_sum[hh]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        my_valid += __shfl_xor(my_valid, o, 64);
        my_bad |= __shfl_xor(my_bad, o, 64);
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) my_above[hh] += __shfl_xor(my_above[hh], o, 64);
    }
    const int nw = (nt + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            s_f[hh][threadIdx.x >> 6] = my_sum[hh];
            s_i[hh][0][threadIdx.x >> 6] = my_valid;
            s_i[hh][1][threadIdx.x >> 6] = my_above[hh];
            s_i[hh][2][threadIdx.x >> 6] = my_bad;
        }
    }
    __syncthreads();
    if (threadIdx.x < NH) {
        const int hh = threadIdx.x, blk = tile;
        float fs = 0.f;
        int nv = 0, na = 0, bad = 0;
        for (int w = 0; w < nw; ++w) fs += s_f[hh][w], nv += s_i[hh][0][w], na += s_i[hh][1][w], bad |= s_i[hh][2][w];
        hd.blk_sum[hh][blk] = fs;
        hd.blk_cnt[hh][2 * blk] = bad ? -(1 << 30) : nv;  // < 0: the block saw an out-of-range label
        hd.blk_cnt[hh][2 * blk + 1] = na;
    }
}

// T[b][c][oy][xs] = sum_ox wx(ox, xs) * G_c(oy, ox),   G = coef * sel * (softmax - onehot)
// One workgroup per (segment of SX source columns, output row): G of the output pixels the segment touches is
// computed ONCE into LDS (one pixel per thread and pass), then every (c, xs) gathers its <= ~2/rw+2 terms.
// threads per workgroup of the x pass.  A 64-column segment touches ~540 output pixels and owns 512 (class, column) outputs,
// so 256 threads leave a nearly empty third G pass -- yet 256 measured best (128: 189, 192: 162, 256: 147, 320: 163, 384: 174,
// 512: 160, 576: 173 us): the kernel is bound by the latency of its dependent phases, and residency (7 workgroups per CU by
// LDS) hides more of it than lane utilisation wins back.
constexpr int OBX_T = 256;
// FR > 0: W == FR * Wl with FR even and a power of two (the model's x8), known at compile time.  Then
//   * the staged pixel window starts at ox_lo = FR xs0 - 3 FR / 2 (pixels outside the image are zeros in G), so the 2 FR
//     pixels that column xs receives from are exactly the de-interleaved rows (phase d, group rel + 1) and (d, rel + 2):
//     16 LDS reads at fixed phases and 16 FMAs with the closed-form triangle (d + 1/2)/FR, 1 - (d + 1/2)/FR (weight 1 where
//     the clamped source index folds the window of the first / last column) -- no tap evaluation, no phase bookkeeping;
//   * every i / FR, i % FR is a shift or a mask.
// FR == 0 is the general resize ratio (taps evaluated per term).
// 1-D grid, decoded XCD-aware: the dispatcher deals blocks round-robin over the 8 XCDs, so block n runs on XCD n % 8.  Each XCD
// walks a CONTIGUOUS chunk of the (image, output row, segment) list -- it then touches one eighth of the `low` planes instead
// of all of them (every XCD's private L2 used to fetch all of both heads' logits: +64 MB per launch at config 3) -- and the
// NH heads of one item are consecutive slots of the SAME XCD (blocks n and n + 8), so the label / validity tile the heads
// share is fetched from HBM once (round 3 ran head 1 as a second z-half of the grid, 16384 blocks later: its label reads
// missed L2, +67 MB; measured pair traffic 436 MB against 277 MB algorithmic).  Speed only, never correctness.
struct OhemBwdHeads {
    const float* low[2];
    const float* loss_px[2];
    float* T[2];
};

template <int CMAX, bool EXACT, int FR>
__global__ __launch_bounds__(OBX_T) void ohem_up_bwd_x_kernel(OhemBwdHeads hd, const long long* __restrict__ labels, int NH, int B,
                                                             int C, int Hl, int Wl, int H, int W, float rh, float rw, float thresh,
                                                             int ignore_lb, float coef, int SX, int nox_max, int R_,
                                                             int gplane) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = FR ? FR : R_;
    const int nseg = (Wl + SX - 1) / SX, items = nseg * H * B, per_xcd = (items + 7) >> 3;
    const int slot = blockIdx.x >> 3, head = slot % NH, item = (blockIdx.x & 7) * per_xcd + slot / NH;
    if (item >= items) return;  // the grid is rounded up to 8 x NH x per_xcd blocks (uniform per workgroup)
    const float* __restrict__ low = hd.low[head];
    const float* __restrict__ loss_px = hd.loss_px[head];
    float* __restrict__ T = hd.T[head];
    const int seg = item % nseg, row = item / nseg, oy = row % H, b = row / H, xs0 = seg * SX, P = H * W;
    const int nxs = min(SX, Wl - xs0);
    // output pixels whose taps can touch [xs0, xs0 + nxs)
    const int ox_lo = FR ? FR * xs0 - 3 * (FR / 2) : max(0, (int)floorf(((float)xs0 - 0.5f) / rw - 0.5f) - 1);
    const int ox_hi = min(W - 1, (int)ceilf(((float)(xs0 + nxs - 1) + 1.5f) / rw - 0.5f) + 1);
    const int nox = FR ? min(FR * (nxs + 3), nox_max) : min(ox_hi - ox_lo + 1, nox_max);
    const int vx0 = max(xs0 - 2, 0), nvx = min(xs0 + nxs + 2, Wl) - vx0;  // staged source columns
    float* v = smem;                      // [C][SX + 4]
    // G[c] is stored de-interleaved by R = round(W / Wl): pixel i sits at (i % R) * gplane + i / R.  The gather
    // below walks i = lo(xs) + iter with lo(xs) advancing by R per lane, so in every iteration all lanes of a wave
    // read the same phase at consecutive words (a plain [c][i] layout is a 16-way bank conflict there).
    float* G = v + C * (SX + 4);          // [C][R * gplane]
    const int gsize = R * gplane;
    const size_t plane = (size_t)Hl * Wl;
    const float* low_b = low + (size_t)b * C * plane;
    int y0, y1;
    float ly;
    bilinear_taps(oy, rh, Hl, y0, y1, ly);
    {
        const float inv_nvx = 1.f / (float)nvx;  // i / nvx for i < C * nvx <= a few thousand: exact via the reciprocal
        for (int i = threadIdx.x; i < C * nvx; i += OBX_T) {
            int c = (int)(((float)i + 0.5f) * inv_nvx);
            const int q = i - c * nvx;
            const float* p = low_b + (size_t)c * plane;
            v[c * (SX + 4) + q] = ((1.f - ly) * p[y0 * Wl + vx0 + q] + ly * p[y1 * Wl + vx0 + q]) * LOG2E_F;  // exp2 domain
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nox; i += OBX_T) {
        const int ox = ox_lo + i;
        const int gi = FR ? (i & (FR - 1)) * gplane + i / (FR ? FR : 1) : (i % R) * gplane + i / R;
        const bool inimg = !FR || (ox >= 0 && ox < W);
        int x0 = 0, x1 = 0;
        float lx = 0.f, loss = 0.f;
        bool sel = false;
        long long lb = 0;
        if (inimg) {
            bilinear_taps(ox, rw, Wl, x0, x1, lx);
            const size_t pix = (size_t)b * P + (size_t)oy * W + ox;
            lb = labels[pix];
            loss = loss_px[pix];
            sel = lb != (long long)ignore_lb && loss > thresh;
        }
        // taps outside the staged window belong to a pixel no column of this segment receives from
        const bool inwin = x0 >= vx0 && x1 < vx0 + nvx;
        if (sel && inwin) {
            // softmax_c = exp(x_c - lse) with lse = loss + x_label: the forward's per-pixel loss IS lse - x_label, so neither
            // the running max nor the sum of exponentials (nor its reciprocal) is recomputed -- this pass is VALU-issue bound
            // (612 VALU instructions per wave measured, 100 % of the SIMD cycles).  v is in the exp2 domain (x log2 e).
            const int lrow = (lb >= 0 && lb < (long long)C) ? (int)lb * (SX + 4) : 0;
            const float xl = (1.f - lx) * v[lrow + x0 - vx0] + lx * v[lrow + x1 - vx0];
            const float lse2 = fmaf(loss, LOG2E_F, xl);
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) {
                    const float xcv = (1.f - lx) * v[c * (SX + 4) + x0 - vx0] + lx * v[c * (SX + 4) + x1 - vx0];
                    G[c * gsize + gi] = coef * fast_exp2(xcv - lse2) - (c == (int)lb ? coef : 0.f);
                }
        } else {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) G[c * gsize + gi] = 0.f;
        }
    }
    __syncthreads();
    const float inv_nxs = 1.f / (float)nxs;
    for (int it = threadIdx.x; it < C * nxs; it += OBX_T) {
        const int c = (int)(((float)it + 0.5f) * inv_nxs), rel = it - c * nxs, xs = xs0 + rel;
        float acc = 0.f;
        const float* Gc = G + c * gsize;
        if (FR) {
            const float* g1 = Gc + rel + 1;  // phase d of the column's first FR pixels; its last FR pixels are one group on
            const bool first = xs == 0, last = xs == Wl - 1;
#pragma unroll
            for (int d = 0; d < (FR ? FR : 1); ++d) {
                const float t = ((float)d + 0.5f) / (float)(FR ? FR : 1);
                const float w1 = first ? 1.f : t, w2 = last ? 1.f : 1.f - t;
                acc = fmaf(w1, g1[d * gplane], acc);
                acc = fmaf(w2, g1[d * gplane + 1], acc);
            }
        } else {
            const int lo = max(ox_lo, (int)floorf(((float)xs - 0.5f) / rw - 0.5f) - 1) - ox_lo;
            const int hi = min(ox_hi, (int)ceilf(((float)xs + 1.5f) / rw - 0.5f) + 1) - ox_lo;
            int ph = lo % R, q = lo / R;  // de-interleaved position of pixel i, advanced incrementally
            for (int i = lo; i <= min(hi, nox - 1); ++i) {
                int x0, x1;
                float lx;
                bilinear_taps(ox_lo + i, rw, Wl, x0, x1, lx);
                // (x1 == x0 at the clamped right edge: both taps are the same column, weight 1 in total)
                const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
                acc = fmaf(wx, Gc[ph * gplane + q], acc);
                if (++ph == R) ph = 0, ++q;
            }
        }
        T[(((size_t)b * C + c) * H + oy) * Wl + xs] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// x pass for the model's x8 resize with whole source rows per wave (W == 8 Wl, Wl == 64 IPL, IPL in {1, 2, 4}) -- round 4.
//
// The segment kernel above is VALU-issue bound (592 vector instructions per wave and 512 output pixels; the counters of
// profiles/r03_pmc_counters.json put the VALU at ~100 % of the SIMD cycles), and most of that is not the softmax: per pixel
// and class it evaluates the horizontal taps twice (2 LDS reads + 2 multiply-adds), stores G, and gathers it again with 16
// more LDS reads per (class, column).  Here G never exists:
//   * a thread of the class phase owns IPL CONSECUTIVE source intervals g of one class row (a wave = one whole class row), and
//     for each interval the eight output pixels ox = 8 g + j.  Those read only the staged columns g-1, g, g+1, so the class'
//     eight upsampled logits are  x_j = fma(t_j, d, v_g)  from ONE difference per side (exactly the forward kernel's expression,
//     so that lse = loss + x_label reconstructs the forward's own log-sum-exp bit for bit);
//   * the adjoint of the resize is applied in registers: pixel j < 4 sends (1 - lam_j) G_j to column g - 1 and lam_j G_j to
//     column g, pixel j >= 4 sends (1 - lam_j) to g and lam_j to g + 1 (clamped taps at the two borders fold into the border
//     column), i.e. three weighted sums L_g, M_g, R_g per interval and  T[xs] = R_{xs-1} + M_xs + L_{xs+1}  with the neighbour
//     terms of the lane's first / last interval fetched by one wave shuffle each -- fixed summation order, no atomics;
//   * everything that does not depend on the class is done ONCE per pixel by the per-interval threads of the first phase:
//     lse_j = loss_j log2 e + x_label (or +inf for a pixel that is not selected: exp2(x - inf) = 0 switches it off), kept in
//     LDS and then in 8 IPL registers of the class-phase lane for all classes; and the one-hot part of (softmax - onehot),
//     which only touches the label's class: its three weighted sums go to oh{L,M,R}[label][g], owned by the interval's thread.
// Per pixel and class this leaves fma + sub + v_exp_f32 + ~2 fma.
template <int IPL>
__global__ __launch_bounds__(256) void ohem_up_bwd_x8row_kernel(OhemBwdHeads hd, const long long* __restrict__ labels, int NH,
                                                                int B, int C, int Hl, int H, float rh, float thresh,
                                                                int ignore_lb, float coef) {
    constexpr int Wl = 64 * IPL, W = 8 * Wl;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* v = smem;                 // [C][Wl]  vertically interpolated logits of this output row, exp2 domain
    float* lse = v + C * Wl;         // [2 IPL][64][4]  per pixel: loss log2 e + x_label, +inf when the pixel is not selected
    float* ohL = lse + W;            // [C][Wl]  one-hot part of L / M / R (sums of tap weights of the selected pixels of class c)
    float* ohM = ohL + C * Wl;
    float* ohR = ohM + C * Wl;
    const int items = H * B, per_xcd = (items + 7) >> 3;
    const int slot = blockIdx.x >> 3, head = slot % NH, item = (blockIdx.x & 7) * per_xcd + slot / NH;
    if (item >= items) return;  // grid rounded up to 8 x NH x per_xcd blocks
    const int oy = item % H, b = item / H, tid = threadIdx.x, CW = C * Wl;
    const float* __restrict__ low = hd.low[head];
    const float* __restrict__ loss_px = hd.loss_px[head];
    float* __restrict__ T = hd.T[head];
    const size_t plane = (size_t)Hl * Wl;
    int y0, y1;
    float ly;
    bilinear_taps(oy, rh, Hl, y0, y1, ly);
    {   // stage v: the rows of v are contiguous in the source (class-major planes, Wl floats per row): independent 16-byte loads
        const float* low_b = low + (size_t)b * C * plane;
        const int q4 = CW >> 2;
        for (int i0 = tid; i0 < q4; i0 += 4 * 256) {
            f32x4 a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * 256, q4 - 1), c = (4 * i) / Wl, xs = 4 * i - c * Wl;
                const float* p = low_b + (size_t)c * plane + xs;
                a0[u] = *reinterpret_cast<const f32x4*>(p + y0 * Wl);
                a1[u] = *reinterpret_cast<const f32x4*>(p + y1 * Wl);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < q4) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ((1.f - ly) * a0[u][e] + ly * a1[u][e]) * LOG2E_F;
                    *reinterpret_cast<f32x4*>(v + 4 * (i0 + u * 256)) = o;
                }
        }
    }
    __syncthreads();
    // ---- phase 1: one thread per source interval g (eight pixels): lse and the one-hot sums
    for (int g = tid; g < Wl; g += 256) {
        const size_t pix = ((size_t)b * H + oy) * W + 8 * g;
        long long lb[8];
        float ls[8];
        {
            const longlong2* lp = reinterpret_cast<const longlong2*>(labels + pix);
            const f32x4* sp = reinterpret_cast<const f32x4*>(loss_px + pix);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const longlong2 t = lp[u];
                lb[2 * u] = t.x, lb[2 * u + 1] = t.y;
            }
            const f32x4 s0 = sp[0], s1 = sp[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) ls[e] = s0[e], ls[4 + e] = s1[e];
        }
        for (int c = 0; c < C; ++c) ohL[c * Wl + g] = 0.f, ohM[c * Wl + g] = 0.f, ohR[c * Wl + g] = 0.f;
        const int gm = max(g - 1, 0), gp = min(g + 1, Wl - 1);
        float out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lam = j < 4 ? ((float)j + 4.5f) * 0.125f : ((float)j - 3.5f) * 0.125f;
            const float t = j < 4 ? -(1.f - lam) : lam;
            const bool sel = lb[j] != (long long)ignore_lb && ls[j] > thresh;
            const bool inrange = lb[j] >= 0 && lb[j] < (long long)C;
            const int lrow = inrange ? (int)lb[j] * Wl : 0;
            const float* vl = v + lrow;
            const float xl = j < 4 ? fmaf(t, vl[g] - vl[gm], vl[g]) : fmaf(t, vl[gp] - vl[g], vl[g]);  // the forward's x_label
            out[j] = sel ? fmaf(ls[j], LOG2E_F, xl) : INFINITY;
            if (sel && inrange) {  // one-hot part: this thread owns column g of every class row (j ascending: a fixed order)
                if (j < 4) {
                    ohL[lrow + g] += 1.f - lam;
                    ohM[lrow + g] += lam;
                } else {
                    ohM[lrow + g] += 1.f - lam;
                    ohR[lrow + g] += lam;
                }
            }
        }
        // stored per (quarter of a lane's pixels, lane): the class phase reads its 8 IPL values as 2 IPL 16-byte loads with a
        // lane stride of 16 bytes (conflict-free; [pixel]-major rows made that a 4-way conflict at IPL = 2)
        const int gl = g / IPL, gu = g - gl * IPL;
        *reinterpret_cast<f32x4*>(lse + ((2 * gu) * 64 + gl) * 4) = f32x4{out[0], out[1], out[2], out[3]};
        *reinterpret_cast<f32x4*>(lse + ((2 * gu + 1) * 64 + gl) * 4) = f32x4{out[4], out[5], out[6], out[7]};
    }
    __syncthreads();
    // ---- phase 2: a wave per class row, a lane per IPL consecutive intervals
    const int wave = tid >> 6, lane = tid & 63, g0 = lane * IPL;
    float lr[8 * IPL];
#pragma unroll
    for (int u = 0; u < 2 * IPL; ++u) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(lse + (u * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) lr[4 * u + e] = t[e];
    }
    for (int c = wave; c < C; c += 4) {
        const float* vc = v + c * Wl;
        float Lg[IPL], Mg[IPL], Rg[IPL];
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int g = g0 + u;
            const float xc = vc[g], dm = xc - vc[max(g - 1, 0)], dp = vc[min(g + 1, Wl - 1)] - xc;
            float L = 0.f, M = 0.f, R = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lam = j < 4 ? ((float)j + 4.5f) * 0.125f : ((float)j - 3.5f) * 0.125f;
                const float e = fast_exp2(fmaf(j < 4 ? -(1.f - lam) : lam, j < 4 ? dm : dp, xc) - lr[8 * u + j]);
                if (j < 4) {
                    L = fmaf(1.f - lam, e, L);
                    M = fmaf(lam, e, M);
                } else {
                    M = fmaf(1.f - lam, e, M);
                    R = fmaf(lam, e, R);
                }
            }
            Lg[u] = coef * (L - ohL[c * Wl + g]);
            Mg[u] = coef * (M - ohM[c * Wl + g]);
            Rg[u] = coef * (R - ohR[c * Wl + g]);
        }
        // T[xs] = R_{xs-1} + M_xs + L_{xs+1}; the clamped taps of the first / last column fold L_0 / R_{Wl-1} into that column
        float rprev = __shfl_up(Rg[IPL - 1], 1, 64), lnext = __shfl_down(Lg[0], 1, 64);
        if (lane == 0) rprev = Lg[0];
        if (lane == 63) lnext = Rg[IPL - 1];
        float o[IPL];
#pragma unroll
        for (int u = 0; u < IPL; ++u) o[u] = ((u > 0 ? Rg[u - 1] : rprev) + Mg[u]) + (u + 1 < IPL ? Lg[u + 1] : lnext);
        float* dst = T + (((size_t)b * C + c) * H + oy) * Wl + g0;
        if (IPL == 1)
            dst[0] = o[0];
        else if (IPL == 2)
            *reinterpret_cast<float2*>(dst) = float2{o[0], o[IPL > 1 ? 1 : 0]};
        else
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[IPL > 1 ? 1 : 0], o[IPL > 2 ? 2 : 0], o[IPL > 3 ? 3 : 0]};
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5 (second half): the same x pass WITHOUT the intermediate T (opt-in: see ohem_up_bwd_run for what it measured).  T (B x C x H x Wl per head: 67 MB for both
// heads at config 3) was written by the row kernel and read back by the y pass -- 134 of the backward's 293 MB.  The y adjoint of
// an x8 resize is local: the eight output rows 8k+4 .. 8k+11 (a BAND) lerp the same two source rows k and k+1, so a workgroup that
// walks a band row by row can keep  sum (1 - ly) T_row  (for source row k) and  sum ly T_row  (for k+1) in registers -- a lane's
// IPL columns of the wave's classes -- and write them once: P0[k] and P1[k+1].  Every source row then gets exactly two addends,
// dlow[ys] = P1[ys] + P0[ys] (ohem_up_bwd_comb_kernel: 8 MB in, 4 MB out), bands -1 (rows 0..3, all weight on source row 0) and
// Hl-1 (rows H-4..H-1, all weight on the last row) included: fixed order, no atomics.  The two source rows are loaded ONCE per band
// (the row kernel fetched them once per output row) and re-lerped per row with the forward's own expression, so lse = loss +
// x_label still reconstructs the forward's log-sum-exp bit for bit.  Phases 1 and 2 are the row kernel's, per row.
template <int IPL>
__global__ __launch_bounds__(256) void ohem_up_bwd_x8band_kernel(OhemBwdHeads hd, const long long* __restrict__ labels, int NH,
                                                                 int B, int C, int Hl, int H, float rh, float thresh,
                                                                 int ignore_lb, float coef) {
    constexpr int Wl = 64 * IPL, W = 8 * Wl, NCI = 8;   // NCI: classes per wave (C <= 32)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CW = C * Wl;
    float* a0 = smem;                // [C][Wl]  source row k   (raw logits)
    float* a1 = a0 + CW;             // [C][Wl]  source row k+1
    float* v = a1 + CW;              // [C][Wl]  vertically interpolated logits of the current output row, exp2 domain
    float* lse = v + CW;             // [2 IPL][64][4]
    float* ohL = lse + W;            // [C][Wl]
    float* ohM = ohL + CW;
    float* ohR = ohM + CW;
    const int nb = Hl + 1, items = nb * B, per_xcd = (items + 7) >> 3;
    const int slot = blockIdx.x >> 3, head = slot % NH, item = (blockIdx.x & 7) * per_xcd + slot / NH;
    if (item >= items) return;
    const int k = item % nb - 1, b = item / nb, tid = threadIdx.x;
    const float* __restrict__ low = hd.low[head];
    const float* __restrict__ loss_px = hd.loss_px[head];
    float* __restrict__ P = hd.T[head];   // [2][B][C][Hl][Wl]
    const size_t plane = (size_t)Hl * Wl, half = (size_t)B * C * plane;
    const int oy_lo = max(8 * k + 4, 0), oy_hi = min(8 * k + 12, H);
    int ya, yb;   // the two source rows every output row of the band interpolates (the taps of its first row)
    {
        float l0;
        bilinear_taps(oy_lo, rh, Hl, ya, yb, l0);
    }
    {
        const float* low_b = low + (size_t)b * C * plane;
        const int q4 = CW >> 2;
        for (int i = tid; i < q4; i += 256) {
            const int c = (4 * i) / Wl, xs = 4 * i - c * Wl;
            const float* p = low_b + (size_t)c * plane + xs;
            *reinterpret_cast<f32x4*>(a0 + 4 * i) = *reinterpret_cast<const f32x4*>(p + ya * Wl);
            *reinterpret_cast<f32x4*>(a1 + 4 * i) = *reinterpret_cast<const f32x4*>(p + yb * Wl);
        }
    }
    const int wave = tid >> 6, lane = tid & 63, g0 = lane * IPL;
    float accA[NCI][IPL], accB[NCI][IPL];
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int u = 0; u < IPL; ++u) accA[ci][u] = 0.f, accB[ci][u] = 0.f;
    __syncthreads();
    for (int oy = oy_lo; oy < oy_hi; ++oy) {
        int y0, y1;
        float ly;
        bilinear_taps(oy, rh, Hl, y0, y1, ly);   // the forward's weights; (y0, y1) == (ya, yb) for every row of the band
        const float dd = (float)(oy - (8 * k + 4));
        const float wb = k < 0 ? 1.f : (dd + 0.5f) * 0.125f;                               // the y pass's closed-form triangle,
        const float wa = k == Hl - 1 ? 1.f : 1.f - (dd + 0.5f) * 0.125f;                   // weight 1 where the clamp folds the window
        for (int i = tid; i < (CW >> 2); i += 256) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(a0 + 4 * i), r1 = *reinterpret_cast<const f32x4*>(a1 + 4 * i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((1.f - ly) * r0[e] + ly * r1[e]) * LOG2E_F;
            *reinterpret_cast<f32x4*>(v + 4 * i) = o;
        }
        __syncthreads();
        // ---- phase 1 (the row kernel's): one thread per source interval g: lse and the one-hot sums
        for (int g = tid; g < Wl; g += 256) {
            const size_t pix = ((size_t)b * H + oy) * W + 8 * g;
            long long lb[8];
            float ls[8];
            {
                const longlong2* lp = reinterpret_cast<const longlong2*>(labels + pix);
                const f32x4* sp = reinterpret_cast<const f32x4*>(loss_px + pix);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const longlong2 t = lp[u];
                    lb[2 * u] = t.x, lb[2 * u + 1] = t.y;
                }
                const f32x4 s0 = sp[0], s1 = sp[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) ls[e] = s0[e], ls[4 + e] = s1[e];
            }
            for (int c = 0; c < C; ++c) ohL[c * Wl + g] = 0.f, ohM[c * Wl + g] = 0.f, ohR[c * Wl + g] = 0.f;
            const int gm = max(g - 1, 0), gp = min(g + 1, Wl - 1);
            float out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lam = j < 4 ? ((float)j + 4.5f) * 0.125f : ((float)j - 3.5f) * 0.125f;
                const float t = j < 4 ? -(1.f - lam) : lam;
                const bool sel = lb[j] != (long long)ignore_lb && ls[j] > thresh;
                const bool inrange = lb[j] >= 0 && lb[j] < (long long)C;
                const int lrow = inrange ? (int)lb[j] * Wl : 0;
                const float* vl = v + lrow;
                const float xl = j < 4 ? fmaf(t, vl[g] - vl[gm], vl[g]) : fmaf(t, vl[gp] - vl[g], vl[g]);
                out[j] = sel ? fmaf(ls[j], LOG2E_F, xl) : INFINITY;
                if (sel && inrange) {
                    if (j < 4) {
                        ohL[lrow + g] += 1.f - lam;
                        ohM[lrow + g] += lam;
                    } else {
                        ohM[lrow + g] += 1.f - lam;
                        ohR[lrow + g] += lam;
                    }
                }
            }
            const int gl = g / IPL, gu = g - gl * IPL;
            *reinterpret_cast<f32x4*>(lse + ((2 * gu) * 64 + gl) * 4) = f32x4{out[0], out[1], out[2], out[3]};
            *reinterpret_cast<f32x4*>(lse + ((2 * gu + 1) * 64 + gl) * 4) = f32x4{out[4], out[5], out[6], out[7]};
        }
        __syncthreads();
        // ---- phase 2 (the row kernel's): a wave per class row, a lane per IPL consecutive intervals; T stays in registers
        float lr[8 * IPL];
#pragma unroll
        for (int u = 0; u < 2 * IPL; ++u) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(lse + (u * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) lr[4 * u + e] = t[e];
        }
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) {
            const int c = wave + 4 * ci;
            if (c < C) {
                const float* vc = v + c * Wl;
                float Lg[IPL], Mg[IPL], Rg[IPL];
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const int g = g0 + u;
                    const float xc = vc[g], dm = xc - vc[max(g - 1, 0)], dp = vc[min(g + 1, Wl - 1)] - xc;
                    float L = 0.f, M = 0.f, R = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float lam = j < 4 ? ((float)j + 4.5f) * 0.125f : ((float)j - 3.5f) * 0.125f;
                        const float e = fast_exp2(fmaf(j < 4 ? -(1.f - lam) : lam, j < 4 ? dm : dp, xc) - lr[8 * u + j]);
                        if (j < 4) {
                            L = fmaf(1.f - lam, e, L);
                            M = fmaf(lam, e, M);
                        } else {
                            M = fmaf(1.f - lam, e, M);
                            R = fmaf(lam, e, R);
                        }
                    }
                    Lg[u] = coef * (L - ohL[c * Wl + g]);
                    Mg[u] = coef * (M - ohM[c * Wl + g]);
                    Rg[u] = coef * (R - ohR[c * Wl + g]);
                }
                float rprev = __shfl_up(Rg[IPL - 1], 1, 64), lnext = __shfl_down(Lg[0], 1, 64);
                if (lane == 0) rprev = Lg[0];
                if (lane == 63) lnext = Rg[IPL - 1];
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const float o = ((u > 0 ? Rg[u - 1] : rprev) + Mg[u]) + (u + 1 < IPL ? Lg[u + 1] : lnext);
                    accA[ci][u] = fmaf(wa, o, accA[ci][u]);
                    accB[ci][u] = fmaf(wb, o, accB[ci][u]);
                }
            }
        }
        __syncthreads();   // the next row overwrites v, lse and the one-hot sums
    }
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci) {
        const int c = wave + 4 * ci;
        if (c < C) {
#pragma unroll
            for (int u = 0; u < IPL; ++u) {
                if (k >= 0) P[(((size_t)b * C + c) * Hl + k) * Wl + g0 + u] = accA[ci][u];
                if (k + 1 < Hl) P[half + (((size_t)b * C + c) * Hl + k + 1) * Wl + g0 + u] = accB[ci][u];
            }
        }
    }
}

// dlow[i] = P1[i] + P0[i]: the contribution of the band above a source row, then of the band below (ascending output rows)
__global__ __launch_bounds__(256) void ohem_up_bwd_comb_kernel(OhemBwdHeads hd, int NH, size_t half4, float* __restrict__ dlow) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= half4 * NH) return;
    const int head = (int)(i / half4);
    const size_t j = i - (size_t)head * half4;
    const f32x4* P = reinterpret_cast<const f32x4*>(hd.T[head]);
    const f32x4 p0 = P[j], p1 = P[half4 + j];
    reinterpret_cast<f32x4*>(dlow)[i] = f32x4{p1[0] + p0[0], p1[1] + p0[1], p1[2] + p0[2], p1[3] + p0[3]};
}

// dlow[b][c][ys][xs] = sum_oy wy(oy, ys) * T[b][c][oy][xs]
__global__ __launch_bounds__(256) void ohem_up_bwd_y_kernel(const float* __restrict__ T, int planes, int Hl, int Wl, int H,
                                                             float rh, int fastR, float* __restrict__ dlow) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Hl * Wl) return;
    const int xs = idx % Wl, ys = (idx / Wl) % Hl, pl = idx / (Wl * Hl);
    if (fastR) {  // H == fastR * Hl, fastR even: the same triangle as in the x pass
        const int w0 = fastR * ys - (fastR >> 1), lo = max(w0, 0), hi = min(w0 + 2 * fastR - 1, H - 1);
        const float invR = 1.f / (float)fastR;
        const float* src = T + (size_t)pl * H * Wl + xs;
        float acc = 0.f;
        for (int oy = lo; oy <= hi; ++oy) {
            const int d = oy - w0;
            float wy = d < fastR ? ((float)d + 0.5f) * invR : 1.f - ((float)(d - fastR) + 0.5f) * invR;
            if ((ys == 0 && d < fastR) || (ys == Hl - 1 && d >= fastR)) wy = 1.f;
            acc = fmaf(wy, src[(size_t)oy * Wl], acc);
        }
        dlow[idx] = acc;
        return;
    }
    const int oy_lo = max(0, (int)floorf(((float)ys - 0.5f) / rh - 0.5f) - 1);
    const int oy_hi = min(H - 1, (int)ceilf(((float)ys + 1.5f) / rh - 0.5f) + 1);
    const float* src = T + (size_t)pl * H * Wl + xs;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly;
        bilinear_taps(oy, rh, Hl, y0, y1, ly);
        acc += ((y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f)) * src[(size_t)oy * Wl];
    }
    dlow[idx] = acc;
}

// The model's x8 rows (H == 8 Hl, Wl % 4 == 0): a thread owns four consecutive columns of one (plane, ys) and ALL 16 rows of
// its window are requested before the first is used (compile-time trip count, row index clamped, weight zero outside the
// image) -- the general kernel above walks a run-time row range, one dependent 4-byte load per step, and ran at 2.3 TB/s on
// a pass that only reads T once.
__global__ __launch_bounds__(256) void ohem_up_bwd_y8_kernel(const float* __restrict__ T, int planes, int Hl, int Wl, int H,
                                                              float* __restrict__ dlow) {
    const int W4 = Wl >> 2, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Hl * W4) return;
    const int x4 = idx % W4, ys = (idx / W4) % Hl, pl = idx / (W4 * Hl);
    const int w0 = 8 * ys - 4;
    const float* src = T + (size_t)pl * H * Wl + 4 * x4;
    f32x4 r[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) r[d] = *reinterpret_cast<const f32x4*>(src + (size_t)min(max(w0 + d, 0), H - 1) * Wl);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 16; ++d) {  // ascending rows: the summation order of the general kernel
        const int oy = w0 + d;
        float wy = d < 8 ? ((float)d + 0.5f) * 0.125f : 1.f - ((float)(d - 8) + 0.5f) * 0.125f;
        if ((ys == 0 && d < 8) || (ys == Hl - 1 && d >= 8)) wy = 1.f;
        if (oy < 0 || oy >= H) wy = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(wy, r[d][e], acc[e]);
    }
    *reinterpret_cast<f32x4*>(dlow + ((size_t)pl * Hl + ys) * Wl + 4 * x4) = acc;
}

int ohem_blocks(int B, int H, int W) { (void)W; return B * H; }  // one partial per output row

// nh = 1 | 2 heads over the same labels; head i: low[i] -> loss_px[i], blk_sum[i], blk_cnt[i]
hipError_t ohem_up_fwd_run(int nh, const float* const* low, const long long* labels, int B, int C, int Hl, int Wl, int H, int W,
                           float thresh, int ignore_lb, float* const* loss_px, float* const* blk_sum, int* const* blk_cnt,
                           hipStream_t stream) {
    // the x8 form needs 16-byte aligned label / loss rows (W % 8 == 0 holds) and one thread per source column
    bool x8 = Wl > 0 && W == 8 * Wl && (reinterpret_cast<uintptr_t>(labels) & 15) == 0;
    for (int i = 0; i < nh; ++i) x8 = x8 && (reinterpret_cast<uintptr_t>(loss_px[i]) & 15) == 0;
    const int nt8 = Wl >= 256 ? 256 : ((Wl + 63) / 64) * 64;
    // both heads in one workgroup (shared label tile, two independent exp / log chains per thread) when their row buffers fit
    const bool pair = nh == 2 && x8 && (size_t)2 * C * Wl * sizeof(float) <= 60 * 1024;
#define OHEM_FWD(CM, EX)                                                                                                \
    do {                                                                                                                \
        if (pair) {                                                                                                     \
            OhemFwdHeads<2> hd{{low[0], low[1]}, {loss_px[0], loss_px[1]}, {blk_sum[0], blk_sum[1]}, {blk_cnt[0], blk_cnt[1]}}; \
            hipLaunchKernelGGL((ohem_up_fwd_x8_kernel<CM, EX, 2>), dim3(H, B), dim3(nt8), (size_t)2 * C * Wl * sizeof(float), \
                               stream, hd, labels, C, Hl, Wl, H, W, (float)Hl / (float)H, thresh, ignore_lb);           \
        } else {                                                                                                        \
            for (int i = 0; i < nh; ++i) {                                                                              \
                if (x8) {                                                                                               \
                    OhemFwdHeads<1> hd{{low[i]}, {loss_px[i]}, {blk_sum[i]}, {blk_cnt[i]}};                             \
                    hipLaunchKernelGGL((ohem_up_fwd_x8_kernel<CM, EX, 1>), dim3(H, B), dim3(nt8),                       \
                                       (size_t)C * Wl * sizeof(float), stream, hd, labels, C, Hl, Wl, H, W,              \
                                       (float)Hl / (float)H, thresh, ignore_lb);                                         \
                } else {                                                                                                \
                    hipLaunchKernelGGL((ohem_up_fwd_kernel<CM, EX>), dim3(H, B), dim3(256), (size_t)C * Wl * sizeof(float), \
                                       stream, low[i], labels, C, Hl, Wl, H, W, (float)Hl / (float)H, (float)Wl / (float)W, \
                                       thresh, ignore_lb, loss_px[i], blk_sum[i], blk_cnt[i]);                          \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    if (C == 8) OHEM_FWD(8, true);
    else if (C == 19) OHEM_FWD(19, true);
    else if (C < 8) OHEM_FWD(8, false);
    else if (C <= 16) OHEM_FWD(16, false);
    else if (C <= 20) OHEM_FWD(20, false);
    else OHEM_FWD(32, false);
#undef OHEM_FWD
    return hipGetLastError();
}

size_t ohem_up_bwd_workspace(int B, int C, int H, int Wl) { return align_up((size_t)B * C * H * Wl * sizeof(float), 256); }

// segment width (source columns per workgroup) and the bound on output pixels a segment touches
static void ohem_segment(int C, int Wl, int W, int& SX, int& nox_max, int& R, int& gplane, size_t& lds) {
    const float rw = (float)Wl / (float)W;
    R = (W + Wl / 2) / Wl;
    if (R < 1) R = 1;
    // (a segment as wide as the whole row -- no halo pixels, exactly filled passes -- measured 164 vs 157 us for both heads:
    // residency, 7 workgroups per CU by LDS, hides more of the dependent phases than the halo costs)
    for (SX = 64; SX >= 1; SX >>= 1) {
        nox_max = (int)((float)(SX + 2) / rw) + 8;
        gplane = ceil_div(nox_max, R);
        // plane stride = 32 / R (mod 32) banks for a power-of-two R <= 32: the 32 lanes ds_write_b32 serves per LDS cycle are
        // 32 consecutive pixels = R phases x 32 / R consecutive groups, i.e. bank = phase * 32 / R + group: all 32 distinct.
        // (Round 3 used 8 (mod 32) for every R: two-way on the x8 writes -- harmless in time, a 2-way store hides under the
        // instruction's own 4 issue cycles, but it is what SQ_LDS_BANK_CONFLICT counted, 150 cycles per workgroup.)
        const int want = (R <= 32 && (R & (R - 1)) == 0) ? 32 / R : 8;
        gplane += (want - gplane % 32 + 32) % 32;
        lds = ((size_t)C * (SX + 4) + (size_t)C * R * gplane) * sizeof(float);
        if (lds <= 60 * 1024) return;
    }
    SX = 0;
}

bool ohem_up_supported(int C, int Wl, int W) {
    int SX, nox, R, gplane;
    size_t lds;
    ohem_segment(C, Wl, W, SX, nox, R, gplane, lds);
    return SX > 0 && (size_t)C * Wl * sizeof(float) <= 60 * 1024;
}

// CABINET_OHEM_SEGMENT_KERNEL=1 forces the round-3 segment kernel for the x pass (A/B timing, tests of the general form)
static bool os_row_kernel_enabled() {
    const char* e = getenv("CABINET_OHEM_SEGMENT_KERNEL");  // read per call: a test toggles it inside one process
    return !(e && e[0] == '1');
}

// nh heads: dlow is (nh, B, C, Hl, Wl) contiguous, the workspace holds nh T slabs
hipError_t ohem_up_bwd_run(int nh, const float* const* low, const long long* labels, const float* const* loss_px, int B, int C,
                           int Hl, int Wl, int H, int W, float thresh, int ignore_lb, float coef, float* dlow, void* ws,
                           hipStream_t stream) {
    float* T = static_cast<float*>(ws);
    const size_t slab = ohem_up_bwd_workspace(B, C, H, Wl) / sizeof(float);
    OhemBwdHeads hd{};
    for (int i = 0; i < nh; ++i) hd.low[i] = low[i], hd.loss_px[i] = loss_px[i], hd.T[i] = T + i * slab;
    int SX, nox_max, R, gplane;
    size_t lds;
    ohem_segment(C, Wl, W, SX, nox_max, R, gplane, lds);
    const int fast_y = (Hl > 0 && H % Hl == 0 && ((H / Hl) & 1) == 0) ? H / Hl : 0;  // integer, even ratio: closed-form weights
    const bool x8 = Wl > 0 && W == 8 * Wl && 8 * (SX + 3) <= nox_max && SX + 3 <= gplane;  // the model's x8 upsample
    const int xgrid = 8 * nh * ceil_div(ceil_div(Wl, SX) * H * B, 8);  // XCD-chunked, the heads of an item adjacent per XCD
    // whole source rows per wave (see ohem_up_bwd_x8row_kernel): the model's geometry at every BASELINE configuration
    const int ipl = (Wl > 0 && W == 8 * Wl && Wl % 64 == 0) ? Wl / 64 : 0;
    const size_t lds_row = ((size_t)4 * C * Wl + (size_t)W) * sizeof(float);
    bool aligned = (reinterpret_cast<uintptr_t>(labels) & 15) == 0;
    for (int i = 0; i < nh; ++i)
        aligned = aligned && ((reinterpret_cast<uintptr_t>(low[i]) | reinterpret_cast<uintptr_t>(loss_px[i]) |
                               reinterpret_cast<uintptr_t>(hd.T[i])) & 15) == 0;
    // round 5: bands of eight output rows, no T (see ohem_up_bwd_x8band_kernel) -- measured and NOT the default: 76 + 6 us against
    // 58 + 13 us for the row kernel + y pass at config 3 (profiles/r05_ohem_band_ab.txt).  The band kernel moves 134 MB less, but a
    // band is eight rows walked one after the other behind three barriers each, in an eighth of the workgroups: the backward is
    // bound by the latency of its dependent phases (label / loss loads -> lse -> exp chain), and 16384 row workgroups hide that
    // better than 2064 band workgroups.  CABINET_OHEM_BAND=1 selects it (A/B timing, test_band_kernel_equals_row_kernel_plus_y_pass).
    const size_t lds_band = ((size_t)6 * C * Wl + (size_t)W) * sizeof(float);
    static const auto band_enabled = [] { const char* e = getenv("CABINET_OHEM_BAND"); return e && e[0] == '1'; };
    if ((ipl == 1 || ipl == 2 || ipl == 4) && aligned && lds_band <= 64 * 1024 && H == 8 * Hl && C <= 32 && os_row_kernel_enabled() &&
        band_enabled() && (reinterpret_cast<uintptr_t>(dlow) & 15) == 0) {
        const int bgrid = 8 * nh * ceil_div((Hl + 1) * B, 8);
        static lds_attr_mask b1{0}, b2{0}, b4{0};
#define OHEM_BAND(I, M)                                                                                                  \
        do {                                                                                                             \
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ohem_up_bwd_x8band_kernel<I>), 64 * 1024, M); \
                e != hipSuccess)                                                                                         \
                return e;                                                                                                \
            hipLaunchKernelGGL((ohem_up_bwd_x8band_kernel<I>), dim3(bgrid), dim3(256), lds_band, stream, hd, labels, nh, B, C, Hl, \
                               H, (float)Hl / (float)H, thresh, ignore_lb, coef);                                        \
        } while (0)
        if (ipl == 1) OHEM_BAND(1, b1);
        else if (ipl == 2) OHEM_BAND(2, b2);
        else OHEM_BAND(4, b4);
#undef OHEM_BAND
        const size_t half4 = (size_t)B * C * Hl * Wl / 4;
        hipLaunchKernelGGL(ohem_up_bwd_comb_kernel, dim3((unsigned)((half4 * nh + 255) / 256)), dim3(256), 0, stream, hd, nh, half4, dlow);
        return hipGetLastError();
    }
    if ((ipl == 1 || ipl == 2 || ipl == 4) && aligned && lds_row <= 64 * 1024 && os_row_kernel_enabled()) {
        const int rgrid = 8 * nh * ceil_div(H * B, 8);
        static lds_attr_mask m1{0}, m2{0}, m4{0};
#define OHEM_ROW(I, M)                                                                                                   \
        do {                                                                                                             \
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ohem_up_bwd_x8row_kernel<I>), 64 * 1024, M);  \
                e != hipSuccess)                                                                                         \
                return e;                                                                                                \
            hipLaunchKernelGGL((ohem_up_bwd_x8row_kernel<I>), dim3(rgrid), dim3(256), lds_row, stream, hd, labels, nh, B, C, Hl, \
                               H, (float)Hl / (float)H, thresh, ignore_lb, coef);                                        \
        } while (0)
        if (ipl == 1) OHEM_ROW(1, m1);
        else if (ipl == 2) OHEM_ROW(2, m2);
        else OHEM_ROW(4, m4);
#undef OHEM_ROW
    } else {
#define OHEM_BWD(CM, EX)                                                                                                 \
    do {                                                                                                                 \
        if (x8)                                                                                                          \
            hipLaunchKernelGGL((ohem_up_bwd_x_kernel<CM, EX, 8>), dim3(xgrid), dim3(OBX_T), lds, stream,                 \
                               hd, labels, nh, B, C, Hl, Wl, H, W, (float)Hl / (float)H, (float)Wl / (float)W, thresh,   \
                               ignore_lb, coef, SX, nox_max, R, gplane);                                                  \
        else                                                                                                             \
            hipLaunchKernelGGL((ohem_up_bwd_x_kernel<CM, EX, 0>), dim3(xgrid), dim3(OBX_T), lds, stream,                 \
                               hd, labels, nh, B, C, Hl, Wl, H, W, (float)Hl / (float)H, (float)Wl / (float)W, thresh,   \
                               ignore_lb, coef, SX, nox_max, R, gplane);                                                  \
    } while (0)
    if (C == 8) OHEM_BWD(8, true);
    else if (C == 19) OHEM_BWD(19, true);
    else if (C < 8) OHEM_BWD(8, false);
    else if (C <= 16) OHEM_BWD(16, false);
    else if (C <= 20) OHEM_BWD(20, false);
    else OHEM_BWD(32, false);
#undef OHEM_BWD
    }
    // the T slabs are contiguous only when slab == B*C*H*Wl exactly; run the y pass per head otherwise
    const size_t plane_floats = (size_t)B * C * H * Wl;
    const bool y8 = fast_y == 8 && (Wl & 3) == 0 && ((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(dlow)) & 15) == 0;
    if (y8 && (nh == 1 || slab == plane_floats)) {
        hipLaunchKernelGGL(ohem_up_bwd_y8_kernel, dim3(ceil_div(nh * B * C * Hl * (Wl >> 2), 256)), dim3(256), 0, stream, T,
                           nh * B * C, Hl, Wl, H, dlow);
    } else if (nh == 1 || slab == plane_floats) {
        hipLaunchKernelGGL(ohem_up_bwd_y_kernel, dim3(ceil_div(nh * B * C * Hl * Wl, 256)), dim3(256), 0, stream, T, nh * B * C, Hl,
                           Wl, H, (float)Hl / (float)H, fast_y, dlow);
    } else {
        for (int i = 0; i < nh; ++i)
            hipLaunchKernelGGL(ohem_up_bwd_y_kernel, dim3(ceil_div(B * C * Hl * Wl, 256)), dim3(256), 0, stream, T + i * slab, B * C,
                               Hl, Wl, H, (float)Hl / (float)H, fast_y, dlow + (size_t)i * B * C * Hl * Wl);
    }
    return hipGetLastError();
}

// The forward's per-workgroup partials -> per head [n_valid, n_above, sum_above] in double, one workgroup per head: a thread
// adds every 256th block in ascending order, then a fixed tree through LDS (bit-reproducible).  The binding did this with
// five small PyTorch launches (int sum, two casts, float sum, cat): a third of the forward group's 74 us.
__global__ __launch_bounds__(256) void ohem_stats_kernel(const float* __restrict__ blk_sum, const int* __restrict__ blk_cnt,
                                                          int nblk, double* __restrict__ stats) {
    __shared__ long long s_v[256], s_a[256];
    __shared__ double s_s[256];
    const int hh = blockIdx.x, t = threadIdx.x;
    const float* bs = blk_sum + (size_t)hh * nblk;
    const int* bc = blk_cnt + (size_t)hh * nblk * 2;
    long long nv = 0, na = 0;
    double sm = 0.0;
    int i = t;
    for (; i + 7 * 256 < nblk; i += 8 * 256) {   // eight blocks' loads in flight, added in ascending order
        int2 c[8];
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = *reinterpret_cast<const int2*>(bc + 2 * (i + 256 * u)), f[u] = bs[i + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) nv += c[u].x, na += c[u].y, sm += (double)f[u];
    }
    for (; i < nblk; i += 256) nv += bc[2 * i], na += bc[2 * i + 1], sm += (double)bs[i];
    s_v[t] = nv, s_a[t] = na, s_s[t] = sm;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (t < o) s_v[t] += s_v[t + o], s_a[t] += s_a[t + o], s_s[t] += s_s[t + o];
        __syncthreads();
    }
    if (t == 0) stats[3 * hh] = (double)s_v[0], stats[3 * hh + 1] = (double)s_a[0], stats[3 * hh + 2] = s_s[0];
}

hipError_t ohem_stats_run(const float* blk_sum, const int* blk_cnt, int nheads, int nblk, double* stats, hipStream_t stream) {
    hipLaunchKernelGGL(ohem_stats_kernel, dim3(nheads), dim3(256), 0, stream, blk_sum, blk_cnt, nblk, stats);
    return hipGetLastError();
}

}  // namespace cabinet
