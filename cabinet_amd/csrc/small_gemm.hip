// Job-batched small GEMMs for the CAB's producers and the 1x1 output projection (reference cab.py:107-123,137-146,155).
//
// At the CAB's resolution the whole batch has only B*H'*W' = 8192 positions (config 3), and every product is
// (128..512) x (128..512) over those positions: 0.03 .. 1.6 GFLOP.  A 128x128-tile GEMM grid has 16..192 workgroups for
// 256 CUs, so each launch costs ~20 us (33 us for the pyramid-bin products) however little it computes, and the K6
// chain issued 12 of them plus 7 split-K weight-gradient products with their slab sums.  Here:
//   sg_gemm   D[m][p] = sum_k A[k][m] B[k][p] per image, 64x64 tiles on 256 threads (4 waves, one 32x32 MFMA block each),
//             up to 12 independent products ("jobs") per launch, each with up to 3 K-segments (so dx = Wq^T dzq + Wk^T dzk +
//             Wv^T dvv is one job) and with A taken either K-major (a weight used transposed, as in backward) or
//             M-major (a row-major (M,K) weight as PyTorch stores it: no staging/transposition kernel in front);
//   sg_dw     dW[m][n] = sum_{b,p} A[b][m][p] X[b][n][p], 64x64 tiles, split over position ranges, up to 8 products per
//             launch, ONE ordered slab-sum launch for all of them (deterministic, no atomics).
// v_mfma_f32_32x32x2_f32 throughout (exact fp32).  Operands with the contracted index contiguous go through stride-33
// LDS images (conflict-free column reads); the others are read row-wise from stride-68 chunks.
#include "blocks.hpp"
#include "common.hpp"

#include <stdint.h>

namespace cabinet {

constexpr int SG_T = 64;     // output tile edge
constexpr int SG_BK = 32;    // contraction chunk
constexpr int SG_STR = 68;   // LDS row stride of a [k][64] chunk (16-byte aligned rows)
constexpr int SD_STR = 33;   // LDS row stride of a [row][32 positions] image

// MB: 64-row blocks of the output tile per workgroup (tile = 64*MB x 64).  MB = 2 halves the B-operand traffic of a
// product whose B is the big operand (dq = dS K^T: dS is 33.5 MB and was read once per 64-row tile of the 128 channels).
// FAST: every job of the launch has full tiles (M % (64 MB) == 0, P % 64 == 0), whole chunks (every K-segment % 32 == 0) and
// 16-byte aligned rows.  Then the loader has no bounds logic, its per-thread offsets and the key-mean bias are computed once,
// and the segment descriptor (kernarg scalar loads) is only touched when a K-segment ends.  The generic loader spent ~134
// VALU + 94 SALU instructions and 7 dependent kernarg loads per 32-deep chunk (16 MFMAs): MFMA utilisation 34 %.
template <int MB, bool FAST>
__global__ __launch_bounds__(256) void sg_gemm_kernel(const SgJobs jobs) {
    constexpr int MT = SG_T * MB, ASTR = MT + 4, ASTR_T = MT + 1;
    __shared__ __attribute__((aligned(16))) float As[2][SG_BK * ASTR];
    __shared__ __attribute__((aligned(16))) float Bs[2][SG_BK * SG_STR];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].tile_base) ++ji;
    const SgJob& J = jobs.j[ji];
    int t = blockIdx.x - J.tile_base;
    const int nt = t % J.tiles_n;
    t /= J.tiles_n;
    const int mt = t % J.tiles_m, b = t / J.tiles_m;
    const int m0 = mt * MT, p0 = nt * SG_T, M = J.M, P = J.P, lda = J.lda, ldp = J.ldp;
    const bool vec_p = J.vec != 0;
    // LDS row strides: a multiple of 4 keeps float4 stores aligned for operands stored as read; an operand transposed on
    // the way in is written with scalar stores down a column, where an odd stride gives 2-way bank conflicts (free for
    // ds_write_b32) instead of 4-way
    const int astr = J.a_mmajor ? ASTR_T : ASTR, bstr = J.b_pmajor ? 65 : SG_STR;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    f32x4 ra[2 * MB], rb[2];
    // ---- FAST loader state: per-thread element offsets (chunk 0 of the current segment), scalar segment cursors
    int offA[2 * MB], offB[2];
    float biasv[2 * MB];
    const float* apc = nullptr;
    const float* bpc = nullptr;
    int seg_cur = 0, seg_left = 0;
    const int stepA = J.a_mmajor ? SG_BK : SG_BK * lda, stepB = J.b_pmajor ? SG_BK : SG_BK * ldp;
    auto open_segment = [&](int sgi) {
        seg_cur = sgi, seg_left = J.nck[sgi];
        apc = J.seg[sgi].a + (size_t)b * J.a_img_stride;
        bpc = J.seg[sgi].b + (size_t)b * J.seg[sgi].b_rows * (J.b_pmajor ? J.ldb : ldp);
    };
    if (FAST) {
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            const int idx = tid + i * 256;
            offA[i] = J.a_mmajor ? (m0 + (idx >> 3)) * lda + (idx & 7) * 4 : (idx / (MT / 4)) * lda + m0 + (idx % (MT / 4)) * 4;
            biasv[i] = (J.a_mmajor && J.a_bias) ? J.a_bias[(size_t)b * M + m0 + (idx >> 3)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            offB[i] = J.b_pmajor ? (p0 + (idx >> 3)) * J.ldb + (idx & 7) * 4 : (idx >> 4) * ldp + p0 + (idx & 15) * 4;
        }
        open_segment(0);
    }
    auto load_next = [&]() {  // FAST: the chunks are fetched strictly in order
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(apc + offA[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[i][e] -= biasv[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bpc + offB[i]);
        apc += stepA, bpc += stepB;
        if (--seg_left == 0 && seg_cur + 1 < J.nseg) open_segment(seg_cur + 1);
    };
    auto load_chunk = [&](int ci) {
        int s = 0, c = ci;
        while (c >= J.nck[s]) c -= J.nck[s], ++s;
        const int k0 = c * SG_BK, kseg = J.seg[s].k;
        const float* __restrict__ ap = J.seg[s].a + (size_t)b * J.a_img_stride;
        const float* __restrict__ bp = J.seg[s].b + (size_t)b * J.seg[s].b_rows * ldp;
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            const int idx = tid + i * 256;
            if (J.a_mmajor) {  // A stored (M, K) row-major: 8 threads read one row's 32-float run
                const int mrow = idx >> 3, k = k0 + (idx & 7) * 4, m = m0 + mrow;
                if (m < M && k < kseg) {
                    ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)m * lda + k);
                    if (J.a_bias) {
                        const float bias = J.a_bias[(size_t)b * M + m];
#pragma unroll
                        for (int e = 0; e < 4; ++e) ra[i][e] -= bias;
                    }
                } else {
                    ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            } else {           // A stored (K, M): 16*MB threads read one k-row's MT-float run
                const int kk = idx / (MT / 4), m = m0 + (idx % (MT / 4)) * 4;
                ra[i] = (k0 + kk < kseg && m < M) ? *reinterpret_cast<const f32x4*>(ap + (size_t)(k0 + kk) * lda + m)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            if (J.b_pmajor) {  // B stored (P, K): 8 threads read one position's 32-float run, transposed on the way to LDS
                const int prow = idx >> 3, k = k0 + (idx & 7) * 4, pp = p0 + prow;
                rb[i] = (pp < P && k < kseg) ? *reinterpret_cast<const f32x4*>(J.seg[s].b + ((size_t)b * J.seg[s].b_rows + pp) * J.ldb + k)
                                             : f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int kk = idx >> 4, p = p0 + (idx & 15) * 4;
            const float* row = bp + (size_t)min(k0 + kk, kseg - 1) * ldp;  // rows past the segment meet zero A rows
            if (vec_p && p + 3 < P) {
                rb[i] = *reinterpret_cast<const f32x4*>(row + p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[i][e] = (p + e < P) ? row[p + e] : 0.f;
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            const int idx = tid + i * 256;
            if (J.a_mmajor) {
                const int mrow = idx >> 3, k4 = (idx & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][(k4 + e) * astr + mrow] = ra[i][e];
            } else {
                *reinterpret_cast<f32x4*>(&As[buf][(idx / (MT / 4)) * ASTR + (idx % (MT / 4)) * 4]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            if (J.b_pmajor) {
                const int prow = idx >> 3, k4 = (idx & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[buf][(k4 + e) * bstr + prow] = rb[i][e];
            } else {
                *reinterpret_cast<f32x4*>(&Bs[buf][(idx >> 4) * SG_STR + (idx & 15) * 4]) = rb[i];
            }
        }
    };

    const int nchunks = J.nck[0] + J.nck[1] + J.nck[2];
    if (FAST) load_next(); else load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ci = 0; ci < nchunks; ++ci) {
        const int buf = ci & 1;
        if (ci + 1 < nchunks) {
            if (FAST) load_next(); else load_chunk(ci + 1);
        }
        const float* Ab = &As[buf][wm * 32 + li];
        const float* Bb = &Bs[buf][wn * 32 + li];
#pragma unroll
        for (int kk = 0; kk < SG_BK; kk += 2) {
            const float bv = Bb[(kk + h) * bstr];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma32(Ab[(kk + h) * astr + mb * 64], bv, acc[mb]);
        }
        if (ci + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* dst = J.dst + (size_t)b * J.dst_rows * ldp;
    const int p = p0 + wn * 32 + li;
    const float alpha = J.alpha != 0.f ? J.alpha : 1.f;
    const float* ob = J.out_bias;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mb * 64 + wm * 32 + acc_row(r) + 4 * h;
            if (FAST || (m < M && p < P)) dst[(size_t)m * ldp + p] = ob ? fmaf(acc[mb][r], alpha, ob[m]) : acc[mb][r] * alpha;
        }
    // Optional statistics epilogue (K6: the q / k projections feed a training-mode BatchNorm): the tile meets in LDS (the operand
    // ring is dead behind the loop's last barrier), four threads walk one row -- 16 values each, sums of (d - pivot) and
    // (d - pivot)^2 with the row's first value as pivot, so nothing cancels -- and two quad exchanges combine them.  Replaces the
    // statistics launch (5 us + a launch boundary on a chain of dependent launches); measured in round 3 as 160 cross-lane
    // shuffles per wave on the accumulator registers it cost 5 us, through LDS it is 16 stores + 16 loads per thread.
    if constexpr (FAST) {
        if (J.stat != nullptr) {
            constexpr int TP = 65;   // pitch: the 16 rows x 4 segments a wave reads in one step fall in 64 different banks
            float* T = &As[0][0];
            static_assert(2 * SG_BK * (SG_T * MB + 4) >= SG_T * MB * TP, "the tile must fit the A ring");
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[(mb * 64 + wm * 32 + acc_row(r) + 4 * h) * TP + wn * 32 + li] = acc[mb][r] * alpha;
            __syncthreads();
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = mb * 64 + (tid >> 2), seg = tid & 3;
                const float* tr = T + row * TP;
                const float pivot = tr[0];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float d = tr[seg * 16 + j] - pivot;
                    s1 += d, s2 = fmaf(d, d, s2);
                }
                s1 += __shfl_xor(s1, 1, 64), s2 += __shfl_xor(s2, 1, 64);
                s1 += __shfl_xor(s1, 2, 64), s2 += __shfl_xor(s2, 2, 64);
                if (seg == 0) {
                    const float mean_d = s1 * (1.f / 64.f);
                    f32x2 o = {pivot + mean_d, fmaxf(s2 - s1 * mean_d, 0.f)};
                    *reinterpret_cast<f32x2*>(J.stat + (((size_t)(m0 + row) * jobs.B + b) * J.tiles_n + nt) * 2) = o;
                }
            }
        }
    }
}

// ---- round 6: the FAST form with a quarter / half of the LDS operand reads ------------------------------------
// sg_gemm_kernel<MB, true> above reads TWO dwords from LDS per MFMA (one per operand and k-step: 32 ds_read_b32 for the 16 MFMAs of
// a chunk at MB = 1), and on gfx950 those issue slots add up with the fp32 MFMAs of the three or four waves that share a SIMD
// (20.8 us for the 1.6 GFLOP of the q / k / v projections: 0.49 of the matrix peak; five launches of this kernel per step).
// Which two k an MFMA step contracts is free as long as A and B agree, so step s takes k = s from lanes 0-31 and k = 16 + s from
// lanes 32-63: the 16 values a lane needs of an operand are then
//   * 16 CONSECUTIVE floats of its row when the operand is stored with the contracted index contiguous (a weight as PyTorch
//     stores it: M-major A): image [row][32 k], pitch 36 (8-lane groups of a ds_read_b128 / ds_write_b128
//     cover all 32 banks once) -- four 16-byte reads per chunk, written with 16-byte stores exactly as loaded (the old form
//     transposed such an operand with scalar stores);
//   * 16 rows 64 (or 128) floats apart when the operand is stored k-major (an activation (K, P), a weight used transposed):
//     image [k][row], pitch = the tile width, read in pairs by ds_read2st64_b32 -- eight reads per chunk.
// 8 .. 16 LDS reads per chunk and operand pair instead of 32; same loaders, same epilogues, same tile -> workgroup mapping.
constexpr int SG_KP = 36;

template <int MB>
__global__ __launch_bounds__(256) void sg_gemm_fast_kernel(const SgJobs jobs) {
    constexpr int MT = SG_T * MB;
    constexpr int A_FLOATS = MT * SG_KP, B_FLOATS = SG_BK * SG_T;   // SG_KP >= SG_BK: either A image fits
    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_FLOATS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].tile_base) ++ji;
    const SgJob& J = jobs.j[ji];
    int t = blockIdx.x - J.tile_base;
    const int nt = t % J.tiles_n;
    t /= J.tiles_n;
    const int mt = t % J.tiles_m, b = t / J.tiles_m;
    const int m0 = mt * MT, p0 = nt * SG_T, M = J.M, lda = J.lda, ldp = J.ldp;
    const bool am = J.a_mmajor != 0;   // workgroup-uniform (B is k-major here: sg_gemm keeps position-major B jobs on the round-3 kernel)

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    f32x4 ra[2 * MB], rb[2];
    int offA[2 * MB], offB[2];
    float biasv[2 * MB];
    const float* apc = nullptr;
    const float* bpc = nullptr;
    int seg_cur = 0, seg_left = 0;
    const int stepA = am ? SG_BK : SG_BK * lda, stepB = SG_BK * ldp;
    auto open_segment = [&](int sgi) {
        seg_cur = sgi, seg_left = J.nck[sgi];
        apc = J.seg[sgi].a + (size_t)b * J.a_img_stride;
        bpc = J.seg[sgi].b + (size_t)b * J.seg[sgi].b_rows * ldp;
    };
#pragma unroll
    for (int i = 0; i < 2 * MB; ++i) {
        const int idx = tid + i * 256;
        offA[i] = am ? (m0 + (idx >> 3)) * lda + (idx & 7) * 4 : (idx / (MT / 4)) * lda + m0 + (idx % (MT / 4)) * 4;
        biasv[i] = (am && J.a_bias) ? J.a_bias[(size_t)b * M + m0 + (idx >> 3)] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        offB[i] = (idx >> 4) * ldp + p0 + (idx & 15) * 4;
    }
    open_segment(0);
    auto load_next = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(apc + offA[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[i][e] -= biasv[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bpc + offB[i]);
        apc += stepA, bpc += stepB;
        if (--seg_left == 0 && seg_cur + 1 < J.nseg) open_segment(seg_cur + 1);
    };
    auto store_chunk = [&](int buf) {   // every operand is written as it was loaded: 16-byte stores, no transposition
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            const int idx = tid + i * 256;
            float* d = am ? &As[buf][(idx >> 3) * SG_KP + (idx & 7) * 4] : &As[buf][(idx / (MT / 4)) * MT + (idx % (MT / 4)) * 4];
            *reinterpret_cast<f32x4*>(d) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;
            float* d = &Bs[buf][(idx >> 4) * SG_T + (idx & 15) * 4];
            *reinterpret_cast<f32x4*>(d) = rb[i];
        }
    };
    const int nchunks = J.nck[0] + J.nck[1] + J.nck[2];
    load_next();
    store_chunk(0);
    __syncthreads();
    for (int ci = 0; ci < nchunks; ++ci) {
        const int buf = ci & 1;
        if (ci + 1 < nchunks) load_next();
        float av[MB][16], bv[16];
        if (am) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const float* r = &As[buf][(mb * 64 + wm * 32 + li) * SG_KP + 16 * h];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(r + 4 * q);
                    av[mb][4 * q] = v[0], av[mb][4 * q + 1] = v[1], av[mb][4 * q + 2] = v[2], av[mb][4 * q + 3] = v[3];
                }
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const float* r = &As[buf][16 * h * MT + mb * 64 + wm * 32 + li];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) av[mb][s2] = r[s2 * MT];
            }
        }
        {
            const float* r = &Bs[buf][16 * h * SG_T + wn * 32 + li];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) bv[s2] = r[s2 * SG_T];
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma32(av[mb][s2], bv[s2], acc[mb]);
        if (ci + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* dst = J.dst + (size_t)b * J.dst_rows * ldp;
    const int p = p0 + wn * 32 + li;
    const float alpha = J.alpha != 0.f ? J.alpha : 1.f;
    const float* ob = J.out_bias;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mb * 64 + wm * 32 + acc_row(r) + 4 * h;
            dst[(size_t)m * ldp + p] = ob ? fmaf(acc[mb][r], alpha, ob[m]) : acc[mb][r] * alpha;
        }
    // statistics epilogue: as sg_gemm_kernel<MB, true> (the operand ring is dead behind the loop's last barrier)
    if (J.stat != nullptr) {
        constexpr int TP = 65;
        float* T = &As[0][0];
        static_assert(2 * A_FLOATS >= MT * TP, "the tile must fit the A ring");
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[(mb * 64 + wm * 32 + acc_row(r) + 4 * h) * TP + wn * 32 + li] = acc[mb][r] * alpha;
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = mb * 64 + (tid >> 2), seg = tid & 3;
            const float* tr = T + row * TP;
            const float pivot = tr[0];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float d = tr[seg * 16 + j] - pivot;
                s1 += d, s2 = fmaf(d, d, s2);
            }
            s1 += __shfl_xor(s1, 1, 64), s2 += __shfl_xor(s2, 1, 64);
            s1 += __shfl_xor(s1, 2, 64), s2 += __shfl_xor(s2, 2, 64);
            if (seg == 0) {
                const float mean_d = s1 * (1.f / 64.f);
                f32x2 o = {pivot + mean_d, fmaxf(s2 - s1 * mean_d, 0.f)};
                *reinterpret_cast<f32x2*>(J.stat + (((size_t)(m0 + row) * jobs.B + b) * J.tiles_n + nt) * 2) = o;
            }
        }
    }
}

const char* sg_gemm_unsupported(const SgJob& j) {
    if (j.lda & 3) return "lda % 4 == 0";
    for (int s = 0; s < j.nseg; ++s)
        if (j.seg[s].k & 3) return "K % 4 == 0";
    if (!j.a_mmajor && (j.M & 3)) return "M % 4 == 0 for a K-major A operand";
    return nullptr;
}

bool sg_gemm(SgJobs& jobs, int B, hipStream_t stream, int mb) {
    const int MT = SG_T * (mb == 2 ? 2 : 1);
    int base = 0;
    jobs.B = B;
    bool fast = true;
    for (int i = 0; i < jobs.n; ++i) {
        SgJob& j = jobs.j[i];
        if (j.ldp == 0) j.ldp = j.P;
        j.vec = (j.ldp & 3) == 0;
        for (int s = 0; s < 3; ++s) {
            j.nck[s] = s < j.nseg ? ceil_div(j.seg[s].k, SG_BK) : 0;
            if (s < j.nseg && (reinterpret_cast<uintptr_t>(j.seg[s].b) & 15)) j.vec = 0;  // column window off a 16-byte boundary
        }
        j.tiles_m = ceil_div(j.M, MT), j.tiles_n = ceil_div(j.P, SG_T), j.tile_base = base;
        base += j.tiles_m * j.tiles_n * B;
        bool f = j.vec && (j.M % MT) == 0 && (j.P % SG_T) == 0 && (j.lda & 3) == 0 && (!j.b_pmajor || (j.ldb & 3) == 0);
        for (int s = 0; s < j.nseg; ++s)
            f = f && (j.seg[s].k % SG_BK) == 0 && (reinterpret_cast<uintptr_t>(j.seg[s].a) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(j.seg[s].b) & 15) == 0;
        if (j.a_img_stride & 3) f = false;
        fast = fast && f;
    }
    // CABINET_SG_KCONTIG=0: the round-3 FAST kernel (two LDS dwords per MFMA) for A/B timing
    static const bool kc_env = [] { const char* e = getenv("CABINET_SG_KCONTIG"); return !(e && e[0] == '0'); }();
    bool kcontig = kc_env;
    for (int i = 0; i < jobs.n; ++i) kcontig = kcontig && !jobs.j[i].b_pmajor;   // no caller stages a position-major B today
    auto kernel = mb == 2 ? (fast ? (kcontig ? sg_gemm_fast_kernel<2> : sg_gemm_kernel<2, true>) : sg_gemm_kernel<2, false>)
                          : (fast ? (kcontig ? sg_gemm_fast_kernel<1> : sg_gemm_kernel<1, true>) : sg_gemm_kernel<1, false>);
    hipLaunchKernelGGL(kernel, dim3(base), dim3(256), 0, stream, jobs);
    return fast;
}

// ------------------------------------------------------------------------------------------------ weight gradients
__global__ __launch_bounds__(256) void sd_dw_kernel(const SdJobs jobs, float* __restrict__ part, int B) {
    // round 6: both operands have the contracted index (the position) contiguous, so both images are [row][32 p] at pitch SG_KP,
    // written with the 16-byte pieces as loaded and read as four ds_read_b128 per operand and chunk (step s contracts p = s in
    // lanes 0-31 and p = 16 + s in lanes 32-63; see sg_gemm_fast_kernel): 8 LDS reads per 16 MFMAs where stride-33 images took 32
    __shared__ __attribute__((aligned(16))) float Az[2][SG_T * SG_KP];
    __shared__ __attribute__((aligned(16))) float Bx[2][SG_T * SG_KP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].tile_base) ++ji;
    const SdJob& J = jobs.j[ji];
    int t = blockIdx.x - J.tile_base;
    const int nt = t % J.tiles_n;
    t /= J.tiles_n;
    const int mt = t % J.tiles_m, split = t / J.tiles_m;  // tiles of one position range are neighbours: operands from L2
    const int m0 = mt * SG_T, n0 = nt * SG_T, M = J.M, N = J.N, P = J.P;
    const int chunk_lo = split * J.cps, chunk_hi = min(chunk_lo + J.cps, B * J.cpi);
    const bool vec_p = (P & 3) == 0;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 rz[2], rx[2];
    // chunks are fetched in order: (image, position) advance incrementally -- a division per chunk is ~40 emulated instructions
    int nb = chunk_lo / J.cpi, np0 = (chunk_lo - nb * J.cpi) * SG_BK;
    const int prow = (tid >> 3), pcol = (tid & 7) * 4;
    auto load_chunk = [&](int) {
        const int b = nb, p0 = np0;
        np0 += SG_BK;
        if (np0 >= J.cpi * SG_BK) np0 = 0, ++nb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = prow + i * 32, p = p0 + pcol;
            const float* zr = J.a + ((size_t)b * J.a_rows + m0 + row) * P;
            const float* xr = J.x + ((size_t)b * J.x_rows + n0 + row) * P;
            const bool zok = m0 + row < M, xok = n0 + row < N;
            if (vec_p && p + 3 < P) {
                rz[i] = zok ? *reinterpret_cast<const f32x4*>(zr + p) : f32x4{0.f, 0.f, 0.f, 0.f};
                rx[i] = xok ? *reinterpret_cast<const f32x4*>(xr + p) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rz[i][e] = (zok && p + e < P) ? zr[p + e] : 0.f;
                    rx[i][e] = (xok && p + e < P) ? xr[p + e] : 0.f;
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, col = (idx & 7) * 4;
            *reinterpret_cast<f32x4*>(&Az[buf][row * SG_KP + col]) = rz[i];
            *reinterpret_cast<f32x4*>(&Bx[buf][row * SG_KP + col]) = rx[i];
        }
    };
    if (chunk_lo < chunk_hi) {
        load_chunk(chunk_lo);
        store_chunk(0);
    }
    __syncthreads();
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
        const int buf = (chunk - chunk_lo) & 1;
        if (chunk + 1 < chunk_hi) load_chunk(chunk + 1);
        const float* Ab = &Az[buf][(wm * 32 + li) * SG_KP + 16 * h];
        const float* Bb = &Bx[buf][(wn * 32 + li) * SG_KP + 16 * h];
        f32x4 av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(Ab + 4 * q), bv[q] = *reinterpret_cast<const f32x4*>(Bb + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(av[q][e], bv[q][e], acc);
        if (chunk + 1 < chunk_hi) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* slab = part + J.slab_off + (size_t)split * M * N;
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + acc_row(r) + 4 * h;
        if (m < M && n < N) slab[(size_t)m * N + n] = acc[r];
    }
}

// out[m * ldo + col_off + n] = sum_split slab[split][m][n], all jobs in one launch, fixed order
__global__ __launch_bounds__(256) void sd_reduce_kernel(const SdJobs jobs, const float* __restrict__ part) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    int ji = 0;
    while (ji + 1 < jobs.n && gid >= jobs.j[ji + 1].elem_base) ++ji;
    const SdJob& J = jobs.j[ji];
    const int i = gid - J.elem_base, count = J.M * J.N;
    if (i >= count) return;
    const float* p = part + J.slab_off + i;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < J.nsplit; k += 4) {
        s0 += p[(size_t)k * count];
        s1 += p[(size_t)(k + 1) * count];
        s2 += p[(size_t)(k + 2) * count];
        s3 += p[(size_t)(k + 3) * count];
    }
    for (; k < J.nsplit; ++k) s0 += p[(size_t)k * count];
    J.out[(size_t)(i / J.N) * J.ldo + J.col_off + (i % J.N)] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ d, int B, int C, int P, float* __restrict__ out) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* row = d + ((size_t)b * C + c) * P;
        float sb = 0.f;
        for (int p = threadIdx.x; p < P; p += 256) sb += row[p];
        s += (double)sb;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
hipError_t channel_sum_run(const float* d, int B, int C, int P, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, stream, d, B, C, P, out);
    return hipGetLastError();
}

// fills the launch geometry of every job; returns the slab floats needed
size_t sd_plan(SdJobs& jobs, int B) {
    int tiles_total = 0;
    for (int i = 0; i < jobs.n; ++i) tiles_total += ceil_div(jobs.j[i].M, SG_T) * ceil_div(jobs.j[i].N, SG_T);
    size_t off = 0;
    int tbase = 0, ebase = 0;
    for (int i = 0; i < jobs.n; ++i) {
        SdJob& j = jobs.j[i];
        j.tiles_m = ceil_div(j.M, SG_T), j.tiles_n = ceil_div(j.N, SG_T);
        j.cpi = ceil_div(j.P, SG_BK);
        const int total = B * j.cpi;
        // ~8 chunks (128 MFMAs per wave) per workgroup, at most 64 splits, and enough workgroups over all jobs (~1000)
        int nsplit = ceil_div(total, 8);
        const int want = ceil_div(1024, tiles_total);
        if (nsplit > want) nsplit = want;
        if (nsplit > 64) nsplit = 64;
        if (nsplit < 1) nsplit = 1;
        j.cps = ceil_div(total, nsplit);
        j.nsplit = ceil_div(total, j.cps);
        j.slab_off = off;
        off += (size_t)j.nsplit * j.M * j.N;
        off = align_up(off, 64);
        j.tile_base = tbase;
        tbase += j.tiles_m * j.tiles_n * j.nsplit;
        j.elem_base = ebase;
        ebase += (int)align_up((size_t)j.M * j.N, 256);
    }
    jobs.total_tiles = tbase;
    jobs.total_elems = ebase;
    return off;
}

hipError_t sd_run(SdJobs& jobs, int B, float* part, hipStream_t stream) {
    sd_plan(jobs, B);
    hipLaunchKernelGGL(sd_dw_kernel, dim3(jobs.total_tiles), dim3(256), 0, stream, jobs, part, B);
    hipLaunchKernelGGL(sd_reduce_kernel, dim3(jobs.total_elems / 256), dim3(256), 0, stream, jobs, part);
    return hipGetLastError();
}

}  // namespace cabinet
