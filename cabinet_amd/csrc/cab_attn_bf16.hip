// K1, split-bf16 variants -- the fused CAB attention forward (reference src/models/cab.py:149-154) on the bf16 matrix pipe.
//
// gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32, what cab_attn_fwd.hip uses) runs at 1/16 of the bf16 rate and there is no
// TF32.  A product of two fp32 numbers can instead be assembled from bf16 pieces: x = x0 + x1 + x2 with x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) -- three 8-bit pieces cover fp32's 24-bit significand EXACTLY -- and
//     x*y = sum over (i, j) of x_i * y_j ,   every x_i * y_j exact in the MFMA's fp32 accumulator.
//   precision 2 ("bf16x6", NS = 3 pieces, 6 products: i + j <= 2): drops only terms below 2^-26 |x y| -- the result is as
//       accurate as the fp32 fma chain (measured against fp64 next to it), at 16/6 = 2.7x the fp32 MFMA rate;
//   precision 1 ("bf16x3", NS = 2 pieces, 3 products: i + j <= 1): ~2^-17 relative per product (~1e-5 per tensor), at
//       16/3 = 5.3x.  Not enough for the 1e-3 GRADIENT contract of the full model (a 1e-5 forward error flips ~1e-5 of the
//       ReLU units behind it), so it is a measured variant, never a default.
//
// Splitting costs ~7 VALU instructions per element, and in attention every query tile would re-split the whole K / V panel
// of its image.  So the operands are split ONCE by a pack pass (attn_pack_*) into bf16 images laid out in MFMA operand order --
// a lane's 8 contraction values are 16 contiguous bytes, a wave's operand load is 1 KB contiguous -- and the attention kernel
// itself converts nothing but P.  Layouts (chunk = 8 bf16 = 16 B; n32 = n rounded up to 32, zero padded):
//   Qp, Kp : [b][piece][s = channel block of 16][position i][h][8]      element e = x[b][16 s + 8 h + e][i]
//   Vp     : [b][piece][kb = key block of 16][channel c][h][8]          element e = v[b][c][16 kb + kappa(h, e)],
//            kappa(h, e) = (e & 3) + 8 (e >> 2) + 4 h  -- the key that accumulator register e (of 8) of lane-half h holds,
//            so that P^T (keys in accumulator rows, query on the lane) is the B operand of O^T += V P^T as it stands.
// The kernel keeps cab_attn_fwd.hip's structure: query on the lane, 4 waves split the key tiles (no barrier in the loop), one
// wave per SIMD with Q / K / V / O register-resident, next tile's S chain and softmax software-pipelined against the current
// tile's PV chain, deferred rescaling, 4-wave merge through LDS, kvsplit for small grids.
#include <stdint.h>

#include <type_traits>

#include "common.hpp"

namespace cabinet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 bload128(buf_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}

// x -> NS bf16 pieces (x0 + x1 [+ x2] == x up to 2^-17 |x| for NS = 2, exactly for NS = 3: every remainder is exact in fp32)
template <int NS>
__device__ __forceinline__ void split_bf16(float x, __bf16 (&p)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = (__bf16)x;
        x -= (float)p[i];
    }
}

// products x_i * y_j kept for NS pieces, smallest first (the accumulator gains the small terms before the large one)
template <int NS>
struct SplitTerms;
template <>
struct SplitTerms<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0};
    static constexpr int B[3] = {0, 1, 0};
};
template <>
struct SplitTerms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int B[6] = {0, 2, 1, 0, 1, 0};
};

// ------------------------------------------------------------------------------------------ pack pass
// ONE launch, three roles by block index (three launches of 4-6 us each cost more than the attention kernel saved):
//   q, k (B, KC, n) -> [b][piece][s][i][h][8]     blocks [0, 2 nbqk): 256 threads = 128 positions x 2 halves of one channel block
//   v (B, VC, n)    -> [b][piece][kb][c][h][8]    the rest: a workgroup transposes a 32-channel x 64-key block through LDS so
//                                                 that both the read (along keys) and the write (along channels) are contiguous
template <int NS, bool VEC>  // VEC: n % 4 == 0 -- 16-byte loads, a thread packs FOUR consecutive positions (q, k) / eight keys (v)
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, u32x4* __restrict__ qp, u32x4* __restrict__ kp,
                                                        u32x4* __restrict__ vp, int KC, int VC, int n, int n32, int B) {
    __shared__ float tile[32][65];
    constexpr int PPT = VEC ? 4 : 1, PB = 128 * PPT;  // positions per thread / per block of the q, k roles
    const int pb = (n32 + PB - 1) / PB, S = KC >> 4, nbqk = pb * S * B;
    int blk = blockIdx.x;
    if (blk < 2 * nbqk) {
        const float* src = blk < nbqk ? q : k;
        u32x4* dst = blk < nbqk ? qp : kp;
        if (blk >= nbqk) blk -= nbqk;
        const int b = blk / (pb * S), r = blk - b * pb * S, s = r / pb;
        const int i = (r - s * pb) * PB + (threadIdx.x >> 1) * PPT, h = threadIdx.x & 1;
        if (i >= n32) return;
        const float* p = src + ((size_t)b * KC + 16 * s + 8 * h) * n + i;
        float x[PPT][8];
        if (VEC) {  // all eight 16-byte loads in flight before any arithmetic (n % 4 == 0 and i % 4 == 0: whole vectors in or out)
            f32x4 t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = i < n ? *reinterpret_cast<const f32x4*>(p + (size_t)e * n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < PPT; ++j) x[j][e] = t[e][j];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[0][e] = i < n ? p[(size_t)e * n] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            bf16x8 out[NS];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 pc[NS];
                split_bf16<NS>(x[j][e], pc);
#pragma unroll
                for (int c = 0; c < NS; ++c) out[c][e] = pc[c];
            }
            if (i + j < n32) {
#pragma unroll
                for (int c = 0; c < NS; ++c)
                    dst[((((size_t)b * NS + c) * S + s) * n32 + i + j) * 2 + h] = __builtin_bit_cast(u32x4, out[c]);
            }
        }
        return;
    }
    blk -= 2 * nbqk;
    const int jb = (n32 + 63) >> 6, cbk = VC >> 5, KB = n32 >> 4;
    const int b = blk / (jb * cbk), r = blk - b * jb * cbk, c0 = (r / jb) * 32, j0 = (r - (r / jb) * jb) * 64;
    {
        const int c = threadIdx.x >> 3, col0 = (threadIdx.x & 7) * 8;
        const float* p = v + ((size_t)b * VC + c0 + c) * n;
        if (VEC) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + col0 + 4 * u;
                const f32x4 t = j < n ? *reinterpret_cast<const f32x4*>(p + j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[c][col0 + 4 * u + e] = t[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[c][col0 + e] = (j0 + col0 + e < n) ? p[j0 + col0 + e] : 0.f;
        }
    }
    __syncthreads();
    const int kbl = threadIdx.x >> 6, c = (threadIdx.x >> 1) & 31, h = threadIdx.x & 1, kb = (j0 >> 4) + kbl;
    if (kb >= KB) return;
    bf16x8 out[NS];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 pc[NS];
        split_bf16<NS>(tile[c][16 * kbl + (e & 3) + 8 * (e >> 2) + 4 * h], pc);
#pragma unroll
        for (int cc = 0; cc < NS; ++cc) out[cc][e] = pc[cc];
    }
#pragma unroll
    for (int cc = 0; cc < NS; ++cc)
        vp[((((size_t)b * NS + cc) * KB + kb) * VC + c0 + c) * 2 + h] = __builtin_bit_cast(u32x4, out[cc]);
}

// ------------------------------------------------------------------------------------------ attention forward
constexpr float kRescaleThresholdBf = 12.0f;  // log2 units, as in cab_attn_fwd.hip

template <int KC, int VC, int NS>
__global__ __launch_bounds__(256) void cab_attn_fwd_bf16_kernel(const u32x4* __restrict__ qp, const u32x4* __restrict__ kp,
                                                                const u32x4* __restrict__ vp, float* __restrict__ ctx,
                                                                float* __restrict__ lse, int n, int n32, float qscale,
                                                                int kvsplit, int B) {
    constexpr int KS = KC / 16, VB = VC / 32, VSTR = 33;
    using T = SplitTerms<NS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    float* s_m = smem + 4 * VC * VSTR;  // [4][32] running max per wave / query
    float* s_l = s_m + 128;             // [4][32] running sum
    float* s_f = s_l + 128;             // [4][32] merge factors

    const int nqt = n32 >> 5, per_img = nqt * kvsplit;
    const int tile = xcd_chunked_tile(blockIdx.x, per_img * B);
    const int b = tile / per_img, rem = tile - b * per_img, split = rem / nqt, i0 = (rem - split * nqt) * 32;
    const int KB = n32 >> 4;

    const unsigned qk_img = (unsigned)NS * KS * n32 * 32, v_img = (unsigned)NS * KB * VC * 32;  // bytes per image
    const buf_rsrc q_rs = make_rsrc(reinterpret_cast<const char*>(qp) + (size_t)b * qk_img, qk_img);
    const buf_rsrc k_rs = make_rsrc(reinterpret_cast<const char*>(kp) + (size_t)b * qk_img, qk_img);
    const buf_rsrc v_rs = make_rsrc(reinterpret_cast<const char*>(vp) + (size_t)b * v_img, v_img);
    const int qk_step = n32 * 32;        // bytes between consecutive (piece, s) panels of Qp / Kp
    const int v_piece = KB * VC * 32;    // bytes between consecutive pieces of Vp

    bf16x8 qr[NS][KS], kr[NS][KS], vr[NS][2][VB];
    {
        const int voff = ((i0 + li) * 2 + h) * 16;
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int s = 0; s < KS; ++s) qr[p][s] = bload128(q_rs, voff, (p * KS + s) * qk_step);
    }
    f32x16 o[VB];
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int NT = n32 >> 5, tstep = 4 * kvsplit;
    int t = split * 4 + wave;

    auto k_voff = [&](int tile_idx) { return ((min(tile_idx, NT - 1) * 32 + li) * 2 + h) * 16; };
    auto v_voff = [&](int tile_idx) { return ((min(tile_idx, NT - 1) * 2 * VC + li) * 2 + h) * 16; };
    auto load_k_step = [&](int s, int voff) {
#pragma unroll
        for (int p = 0; p < NS; ++p) kr[p][s] = bload128(k_rs, voff, (p * KS + s) * qk_step);
    };
    auto load_v_blk = [&](int u, int cb, int voff) {
#pragma unroll
        for (int p = 0; p < NS; ++p) vr[p][u][cb] = bload128(v_rs, voff, p * v_piece + (u * VC + cb * 32) * 32);
    };
    // S^T = K^T Q for one key tile; with REFILL the K registers of each channel block are reloaded for `next_voff` right
    // behind the MFMAs that consumed them (a full tile time ahead of their next use)
    auto s_chain = [&](f32x16& s, int next_voff) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
#pragma unroll
            for (int e = 0; e < T::N; ++e) s = mfma_bf16(kr[T::A[e]][c], qr[T::B[e]][c], s);
            load_k_step(c, next_voff);
        }
    };
    // online softmax of one S^T tile (statistics per lane = per query), then P split into its bf16 pieces in B-operand order
    auto softmax_split = [&](f32x16& s, int j0, bool mask, float& alpha, bf16x8 (&pp)[NS][2]) {
        if (mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (j0 + acc_row(r) + 4 * h >= n) ? -INFINITY : s[r];
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, swap_half(mt)) * qscale;
        const float mn = (mt > m + kRescaleThresholdBf) ? mt : m;
        alpha = fast_exp2(m - mn);
        m = mn;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = fast_exp2(fmaf(s[r], qscale, -mn));
            rs += pv;
            __bf16 pc[NS];
            split_bf16<NS>(pv, pc);
#pragma unroll
            for (int p = 0; p < NS; ++p) pp[p][r >> 3][r & 7] = pc[p];
        }
        l = l * alpha + rs;
    };
    // O^T += V P^T for one key tile (two key blocks of 16); V registers refilled for `next_voff` behind their use
    auto pv_chain = [&](const bf16x8 (&pp)[NS][2], int next_voff) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int cb = 0; cb < VB; ++cb) {
#pragma unroll
                for (int e = 0; e < T::N; ++e) o[cb] = mfma_bf16(vr[T::A[e]][u][cb], pp[T::B[e]][u], o[cb]);
                load_v_blk(u, cb, next_voff);
            }
    };

    if (t < NT) {
        f32x16 sc, sn;
        bf16x8 pp[NS][2], pn[NS][2];
        float alpha;
        {
            const int kv = k_voff(t), vv = v_voff(t);
#pragma unroll
            for (int c = 0; c < KS; ++c) load_k_step(c, kv);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int cb = 0; cb < VB; ++cb) load_v_blk(u, cb, vv);
        }
        s_chain(sc, k_voff(t + tstep));
        softmax_split(sc, t * 32, t * 32 + 32 > n, alpha, pp);  // O is still zero: nothing to rescale
        for (; t + tstep < NT; t += tstep) {
            // ---- phase A: S^T of the next tile, one fenced slot per MFMA; the K registers of a channel block move on to
            // the tile after next right behind the MFMAs that consumed them ----
            {
                const int nv = k_voff(t + 2 * tstep);
#pragma unroll
                for (int r = 0; r < 16; ++r) sn[r] = 0.f;
#pragma unroll
                for (int c = 0; c < KS; ++c) {
#pragma unroll
                    for (int e = 0; e < T::N; ++e) {
                        sn = mfma_bf16(kr[T::A[e]][c], qr[T::B[e]][c], sn);
                        if (e == T::N - 1) load_k_step(c, nv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- phase B: PV of this tile, one fenced slot per MFMA, each followed by its share of the next tile's softmax
            // and P split (the wave issues in order: VALU work placed behind a group of MFMAs would run with the matrix pipe
            // idle, and left to itself the compiler emits the softmax as one block) and by the V refill loads ----
            auto phase_b = [&](auto mask_tag) {
                constexpr bool MASK = decltype(mask_tag)::value;  // only the last key tile can be ragged
                constexpr int SLOTS = 2 * VB * T::N;
                // micro-steps of the softmax: 8 x (max of a register pair) | row statistics | 8 pairs x (exp, split...) ;
                constexpr int PER_PAIR = NS;              // exp + first piece | remaining pieces, one step each
                constexpr int NSTEP = 8 + 1 + 8 * PER_PAIR;
                const int nv = v_voff(t + tstep), jn = (t + tstep) * 32;
                float mt = -INFINITY, mn = 0.f, rs = 0.f;
                float res[16];
                auto step = [&](int st) {
                    if (st < 8) {
#pragma unroll
                        for (int r = 2 * st; r < 2 * st + 2; ++r) {
                            if (MASK) sn[r] = (jn + acc_row(r) + 4 * h >= n) ? -INFINITY : sn[r];
                            mt = fmaxf(mt, sn[r]);
                        }
                    } else if (st == 8) {
                        mt = fmaxf(mt, swap_half(mt)) * qscale;
                        mn = (mt > m + kRescaleThresholdBf) ? mt : m;
                        alpha = fast_exp2(m - mn);
                        m = mn;
                    } else {
                        const int j = (st - 9) / PER_PAIR, piece = (st - 9) - j * PER_PAIR, r0 = 2 * j;
                        if (piece == 0) {
#pragma unroll
                            for (int r = r0; r < r0 + 2; ++r) {
                                res[r] = fast_exp2(fmaf(sn[r], qscale, -mn));
                                rs += res[r];
                            }
                        }
#pragma unroll
                        for (int r = r0; r < r0 + 2; ++r) {
                            const __bf16 pc = (__bf16)res[r];
                            pn[piece][r >> 3][r & 7] = pc;
                            res[r] -= (float)pc;
                        }
                    }
                };
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
                        for (int e = 0; e < T::N; ++e) {
                            const int slot = (u * VB + cb) * T::N + e;
                            o[cb] = mfma_bf16(vr[T::A[e]][u][cb], pp[T::B[e]][u], o[cb]);
                            if (e == T::N - 1) load_v_blk(u, cb, nv);
#pragma unroll
                            for (int st = slot * NSTEP / SLOTS; st < (slot + 1) * NSTEP / SLOTS; ++st) step(st);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                l = l * alpha + rs;
            };
            if ((t + tstep) * 32 + 32 > n)
                phase_b(std::true_type{});
            else
                phase_b(std::false_type{});
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int cb = 0; cb < VB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
            }
#pragma unroll
            for (int p = 0; p < NS; ++p) pp[p][0] = pn[p][0], pp[p][1] = pn[p][1];
        }
        pv_chain(pp, v_voff(t));  // last tile of this wave (the refill loads are harmless re-reads)
    }

    // ---- merge the 4 waves (disjoint key subsets), as in cab_attn_fwd.hip ----
    l += swap_half(l);
    if (h == 0) {
        s_m[wave * 32 + li] = m;
        s_l[wave * 32 + li] = l;
    }
    float* vs = smem + wave * (VC * VSTR);
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) vs[(cb * 32 + acc_row(r) + 4 * h) * 32 + li] = o[cb][r];
    __syncthreads();
    float ms = -INFINITY, lt = 0.f;
    if (threadIdx.x < 128) {
        const int i = threadIdx.x & 31, w = threadIdx.x >> 5;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) ms = fmaxf(ms, s_m[ww * 32 + i]);
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) lt += s_l[ww * 32 + i] * fast_exp2(s_m[ww * 32 + i] - ms);
        s_f[w * 32 + i] = (lt > 0.f) ? fast_exp2(s_m[w * 32 + i] - ms) / lt : 0.f;
    }
    __syncthreads();
    const size_t out_base = ((size_t)split * B + b) * VC * n;
    if ((n & 3) == 0) {
        for (int idx = threadIdx.x; idx < VC * 8; idx += 256) {
            const int c = idx >> 3, i = (idx & 7) * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 ov = *reinterpret_cast<const f32x4*>(smem + w * (VC * VSTR) + c * 32 + i);
                const f32x4 fv = *reinterpret_cast<const f32x4*>(s_f + w * 32 + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += ov[e] * fv[e];
            }
            if (i0 + i < n) *reinterpret_cast<f32x4*>(ctx + out_base + (size_t)c * n + i0 + i) = acc;
        }
    } else {
        for (int idx = threadIdx.x; idx < VC * 32; idx += 256) {
            const int c = idx >> 5, i = idx & 31;
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc += smem[w * (VC * VSTR) + c * 32 + i] * s_f[w * 32 + i];
            if (i0 + i < n) ctx[out_base + (size_t)c * n + i0 + i] = acc;
        }
    }
    if (threadIdx.x < 32 && i0 + threadIdx.x < n)
        lse[((size_t)split * B + b) * n + i0 + threadIdx.x] = (lt > 0.f) ? (ms + fast_log2(lt)) * LN2_F : -INFINITY;
}

// defined in cab_attn_fwd.hip
void launch_attn_merge(const float* part_ctx, const float* part_lse, float* ctx, float* lse, int B, int VC, int n, int kvsplit,
                       hipStream_t stream);

static int round32(int n) { return (n + 31) / 32 * 32; }

// bytes of the packed operand images (Qp, Kp, Vp) for `precision` 1 (bf16x3) or 2 (bf16x6)
size_t attn_bf16_pack_bytes(int B, int Kc, int Vc, int n, int precision) {
    const size_t ns = precision == 2 ? 3 : 2, n32 = round32(n);
    return align_up((size_t)B * ns * (2 * (size_t)Kc + Vc) * n32 * 2, 256);
}

bool attn_bf16_supported(int Kc, int Vc) { return (Kc == 128 && Vc == 128) || (Kc == 64 && Vc == 64); }

template <int KC, int VC, int NS>
static hipError_t launch_fwd_bf16(const float* q, const float* k, const float* v, float scale, int B, int n, float* ctx, float* lse,
                                  float* part_ctx, float* part_lse, int kvsplit, void* pack, hipStream_t stream) {
    const int n32 = round32(n);
    const size_t qk_chunks = (size_t)B * NS * KC * n32 / 8;  // 16-byte chunks of one packed q / k image set
    u32x4* qp = static_cast<u32x4*>(pack);
    u32x4* kp = qp + qk_chunks;
    u32x4* vp = kp + qk_chunks;
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int nb_pack = 2 * ceil_div(n32, vec ? 512 : 128) * (KC / 16) * B + ceil_div(n32, 64) * (VC / 32) * B;
    if (vec)
        hipLaunchKernelGGL((attn_pack_kernel<NS, true>), dim3(nb_pack), dim3(256), 0, stream, q, k, v, qp, kp, vp, KC, VC, n, n32, B);
    else
        hipLaunchKernelGGL((attn_pack_kernel<NS, false>), dim3(nb_pack), dim3(256), 0, stream, q, k, v, qp, kp, vp, KC, VC, n, n32, B);
    const size_t lds = (size_t)(4 * VC * 33 + 3 * 128) * sizeof(float);
    auto kern = cab_attn_fwd_bf16_kernel<KC, VC, NS>;
    static lds_attr_mask attr_mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_mask); e != hipSuccess) return e;
    const dim3 grid((n32 / 32) * kvsplit * B);
    if (kvsplit == 1) {
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, qp, kp, vp, ctx, lse, n, n32, scale * LOG2E_F, 1, B);
    } else {
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, qp, kp, vp, part_ctx, part_lse, n, n32, scale * LOG2E_F, kvsplit, B);
        launch_attn_merge(part_ctx, part_lse, ctx, lse, B, VC, n, kvsplit, stream);
    }
    return hipGetLastError();
}

hipError_t attn_fwd_bf16_dispatch(const float* q, const float* k, const float* v, float scale, int B, int Kc, int Vc, int n,
                                  float* ctx, float* lse, float* part_ctx, float* part_lse, int kvsplit, void* pack, int precision,
                                  hipStream_t stream) {
#define CAB_BF16_CASE(KC_, VC_)                                                                                              \
    if (Kc == KC_ && Vc == VC_)                                                                                              \
        return precision == 2                                                                                                \
                   ? launch_fwd_bf16<KC_, VC_, 3>(q, k, v, scale, B, n, ctx, lse, part_ctx, part_lse, kvsplit, pack, stream) \
                   : launch_fwd_bf16<KC_, VC_, 2>(q, k, v, scale, B, n, ctx, lse, part_ctx, part_lse, kvsplit, pack, stream);
    CAB_BF16_CASE(128, 128)
    CAB_BF16_CASE(64, 64)
#undef CAB_BF16_CASE
    return hipErrorInvalidValue;
}

}  // namespace cabinet
