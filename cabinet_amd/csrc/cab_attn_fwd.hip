// K1 -- fused CAB attention forward: affinity -> online softmax -> aggregation.
//
// Replaces reference src/models/cab.py:149-154 (bmm, scale, softmax(dim=-1), bmm,
// transpose+view).  Inputs stay in the NCHW-flattened layout the producers emit:
//   q, k : (B, KC, n)   v : (B, VC, n)   ->   ctx : (B, VC, n),  lse : (B, n)
//
// Design (MI355X, exact fp32 via v_mfma_f32_32x32x2_f32):
//  * "query on the lane" throughout.  One workgroup = 32 queries; its waves split the KEYS (wave w takes key tiles
//    t == w mod #waves), so there is no barrier in the main loop and a 8x32x32 image still fills 256 CUs.
//  * S^T tile (32 keys x 32 queries) = K^T Q:  A = K[c][j] (key on lane),
//    B = Q[c][i] (query on lane): both are contiguous 128-B reads of NCHW rows.
//    The accumulator then has the query on the lane and 16 keys in registers,
//    which is exactly the B operand of O^T += V P^T, so P never leaves registers
//    and the softmax row statistics are per-lane scalars (one cross-half swap).
//  * Rescaling of O is deferred (only when the running max grows by > 2^12), so
//    the common tile is MFMA + 16 exp2 per lane.
//  * The per-wave partial results are merged through LDS at the end; with
//    kvsplit > 1 (small batches) partials go to a workspace and a tiny second
//    kernel merges them, so the grid always covers the chip.
// Two kernels share this design:
//    cab_attn_fwd_w8_kernel  (default: n % 4 == 0, Kc <= 128)  8 waves = two per SIMD, operands in 16-register batches, the V
//                            operand (channel on the lane) as 16-byte loads straight from global memory, no LDS in the loop;
//    cab_attn_fwd_kernel     (ragged n, Kc = 256)  4 waves = one per SIMD with the 512-VGPR budget (Q, O and the streamed
//                            K tile in registers), V transposed through a wave-private padded LDS image, tile t + 1's
//                            softmax interleaved into tile t's PV slots.
#include <type_traits>

#include "common.hpp"

namespace cabinet {

constexpr float kRescaleThreshold = 12.0f;  // log2 units

// Diagnostic build only (tools/attn_stamps.hip defines CAB_ATTN_STAMPS): s_memtime marks of wave 0 of every
// workgroup, written to a buffer nothing else reads.  Never compiled into libcabinet_hip.so.
#ifdef CAB_ATTN_STAMPS
__device__ unsigned long long cab_stamps[4096][8];
#define K1_MARK(var)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");     \
    __builtin_amdgcn_sched_barrier(0);
#define K1_DECL unsigned long long k1_t0 = 0, k1_t1 = 0, k1_t2 = 0, k1_t3 = 0, k1_t4 = 0, k1_a = 0, k1_b = 0, k1_x = 0, k1_y = 0;
#else
#define K1_MARK(var)
#define K1_DECL
#endif

template <int KC, int VC>
__global__ __launch_bounds__(256) void cab_attn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ ctx, float* __restrict__ lse, int n, float qscale, int kvsplit, int B) {
    constexpr int VB = VC / 32;
    constexpr int VSTR = 33;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    K1_DECL
    K1_MARK(k1_t0)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    float* vs = smem + wave * (VC * VSTR);
    float* s_m = smem + 4 * VC * VSTR;  // [4][32] running max per wave / query
    float* s_l = s_m + 128;             // [4][32] running sum
    float* s_f = s_l + 128;             // [4][32] merge factors

    // tile list is image-major: (b, split, query tile); each XCD takes a contiguous chunk of it so the
    // (Kc+Vc) x n key/value panel of an image is fetched into ONE L2 instead of all eight
    const int nqt = (n + 31) >> 5, per_img = nqt * kvsplit;
    const int tile = xcd_chunked_tile(blockIdx.x, per_img * B);
    const int b = tile / per_img, rem = tile - b * per_img, split = rem / nqt, i0 = (rem - split * nqt) * 32;

    const int row_bytes = n * 4;  // one NCHW row (channel) of this image
    const buf_rsrc q_rs = make_rsrc(q + (size_t)b * KC * n, (unsigned)KC * row_bytes);
    const buf_rsrc k_rs = make_rsrc(k + (size_t)b * KC * n, (unsigned)KC * row_bytes);
    const buf_rsrc v_rs = make_rsrc(v + (size_t)b * VC * n, (unsigned)VC * row_bytes);

    // Q operand: lane (li,h) holds the RAW q[2s+h][i0+li] for s = 0..KC/2-1 (no arithmetic on the loaded
    // values, so the 64 loads stay in flight together); scale*log2(e) is applied inside the softmax.
    float qreg[KC / 2];
    {
        const int voff = (h * n + min(i0 + li, n - 1)) * 4;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) qreg[s] = bload(q_rs, voff, s * 2 * row_bytes);
    }

    f32x16 o[VB];
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int NT = (n + 31) >> 5, tstep = 4 * kvsplit;
    int t = split * 4 + wave;

    // ---- software-pipelined main loop -------------------------------------------------------------
    // Iteration i overlaps three tiles so the matrix pipe always has a ready MFMA:
    //   phase A:  S^T(i+1) chain (MFMA)      ||  V(i) registers -> LDS image          (LDS idle otherwise)
    //   phase B:  O^T += V(i) P(i)^T (MFMA)  ||  softmax of S^T(i+1) on the VALU, loads K(i+2), V(i+1)
    // K and V tiles are fetched one full phase before use; K loads are issued before V loads so the
    // 6-bit in-order vmcnt can express "K landed" (<= 63 outstanding) without draining V.
    float kA[KC / 2], vA[VC / 2];
    auto load_k = [&](int tile_idx) {
        const int voff = (h * n + min(min(tile_idx, NT - 1) * 32 + li, n - 1)) * 4;
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) kA[c2] = bload(k_rs, voff, c2 * 2 * row_bytes);
    };
    auto load_v = [&](int tile_idx) {
        const int voff = (h * n + min(min(tile_idx, NT - 1) * 32 + li, n - 1)) * 4;
#pragma unroll
        for (int c2 = 0; c2 < VC / 2; ++c2) vA[c2] = bload(v_rs, voff, c2 * 2 * row_bytes);
    };
    auto s_chain = [&](f32x16& s) {  // S^T = K^T Q: keys in accumulator rows, query on the lane
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) s = mfma32(kA[c2], qreg[c2], s);
    };
    auto v_to_lds = [&]() {  // channel-on-lane reads of V need the transpose: padded wave-private image
#pragma unroll
        for (int c2 = 0; c2 < VC / 2; ++c2) vs[(2 * c2 + h) * VSTR + li] = vA[c2];
    };
    // online softmax of one S^T tile, branch-free; statistics are per lane (= per query).
    // Rescaling is deferred: the running max moves only when it grows by > 2^kRescaleThreshold.
    auto softmax = [&](f32x16& s, int j0, float& alpha) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (j0 + acc_row(r) + 4 * h >= n) ? -INFINITY : s[r];
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, swap_half(mt)) * qscale;  // qscale > 0: max commutes with the scaling
        const float mn = (mt > m + kRescaleThreshold) ? mt : m;
        alpha = fast_exp2(m - mn);  // 1 when the max stays; 0 on the first tile (m == -inf)
        m = mn;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(fmaf(s[r], qscale, -mn));
            rs += s[r];
        }
        l = l * alpha + rs;
    };
    auto pv = [&](const f32x16& p) {  // O^T += V P^T : A = V[c][key] from LDS, B = P^T registers
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = acc_row(r) + 4 * h;
#pragma unroll
            for (int cb = 0; cb < VB; ++cb) o[cb] = mfma32(vs[(cb * 32 + li) * VSTR + key], p[r], o[cb]);
        }
    };

    if (t < NT) {
        f32x16 p, sn;
        float alpha;
        load_k(t);
        load_v(t);
        s_chain(p);
        load_k(t + tstep);
        softmax(p, t * 32, alpha);  // O is still zero: nothing to rescale
        K1_MARK(k1_t1)
        for (; t + tstep < NT; t += tstep) {
            K1_MARK(k1_x)
            // ---- phase A: S^T(next) chain; behind each MFMA one row pair of V(t) goes registers -> LDS and
            // the same registers are refilled with V(next) (needed one iteration from now) ----
            {
                const int voff_v = (h * n + min(min(t + tstep, NT - 1) * 32 + li, n - 1)) * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r) sn[r] = 0.f;
#pragma unroll
                for (int c2 = 0; c2 < KC / 2; ++c2) {
                    sn = mfma32(kA[c2], qreg[c2], sn);
#pragma unroll
                    for (int u = c2 * (VC / 2) / (KC / 2); u < (c2 + 1) * (VC / 2) / (KC / 2); ++u) {
                        vs[(2 * u + h) * VSTR + li] = vA[u];
                        vA[u] = bload(v_rs, voff_v, u * 2 * row_bytes);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            K1_MARK(k1_y)
#ifdef CAB_ATTN_STAMPS
            k1_a += k1_y - k1_x;
#endif
            // ---- phase B: 64 fenced slots, one PV MFMA each, with that slot's share of everything else issued
            // right behind it (a K prefetch load, the LDS read of the operand four slots ahead, a slice of the
            // next tile's softmax).  The wave issues in order: work placed after a group of MFMAs runs with
            // the matrix pipe idle, and left to itself the scheduler emitted the softmax as one block. ----
            auto phase_b = [&](auto mask_tag) {
                constexpr bool MASK = decltype(mask_tag)::value;  // only the last key tile can be ragged
                const int voff_k = (h * n + min(min(t + 2 * tstep, NT - 1) * 32 + li, n - 1)) * 4;
                const int jn = (t + tstep) * 32;
                float mt = -INFINITY, mn = 0.f, rs = 0.f;
                float va[VB], vb[VB];
#pragma unroll
                for (int cb = 0; cb < VB; ++cb) va[cb] = vs[(cb * 32 + li) * VSTR + acc_row(0) + 4 * h];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int cb = 0; cb < VB; ++cb) {
                        constexpr int SLOTS = 16 * VB;
                        const int slot = r * VB + cb;
                        o[cb] = mfma32(va[cb], p[r], o[cb]);
                        if (r + 1 < 16) vb[cb] = vs[(cb * 32 + li) * VSTR + acc_row(r + 1) + 4 * h];
                        // prefetch loads: the KC/2 K rows of tile t + 2*tstep spread over the slots
#pragma unroll
                        for (int c2 = slot * (KC / 2) / SLOTS; c2 < (slot + 1) * (KC / 2) / SLOTS; ++c2)
                            kA[c2] = bload(k_rs, voff_k, c2 * 2 * row_bytes);
                        // softmax of S^T(next): first half of the slots = masked running max, middle slot =
                        // row statistics, second half = exponentials
                        if (slot < SLOTS / 2) {
#pragma unroll
                            for (int e = slot * 32 / SLOTS; e < (slot + 1) * 32 / SLOTS; ++e) {
                                if (MASK) sn[e] = (jn + acc_row(e) + 4 * h >= n) ? -INFINITY : sn[e];
                                mt = fmaxf(mt, sn[e]);
                            }
                        }
                        if (slot == SLOTS / 2) {
                            mt = fmaxf(mt, swap_half(mt)) * qscale;
                            mn = (mt > m + kRescaleThreshold) ? mt : m;
                            alpha = fast_exp2(m - mn);
                            m = mn;
                        }
                        if (slot >= SLOTS / 2) {
#pragma unroll
                            for (int e = (slot - SLOTS / 2) * 32 / SLOTS; e < (slot + 1 - SLOTS / 2) * 32 / SLOTS; ++e) {
                                sn[e] = fast_exp2(fmaf(sn[e], qscale, -mn));
                                rs += sn[e];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int cb = 0; cb < VB; ++cb) va[cb] = vb[cb];
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
            p = sn;
            K1_MARK(k1_x)
#ifdef CAB_ATTN_STAMPS
            k1_b += k1_x - k1_y;
#endif
        }
        K1_MARK(k1_t2)
        // ---- last tile of this wave: nothing left to overlap ----
        v_to_lds();
        pv(p);
        K1_MARK(k1_t3)
    }

    // ---- merge the 4 waves (disjoint key subsets) ----
    l += swap_half(l);
    if (h == 0) {
        s_m[wave * 32 + li] = m;
        s_l[wave * 32 + li] = l;
    }
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
        // a (split of a) query row with no key at all cannot happen (host keeps
        // kvsplit <= NT/4), but stay finite if it does
        s_f[w * 32 + i] = (lt > 0.f) ? fast_exp2(s_m[w * 32 + i] - ms) / lt : 0.f;
    }
    __syncthreads();
    const size_t out_base = ((size_t)split * B + b) * VC * n;
    if ((n & 3) == 0) {
        // four consecutive queries per thread: 16-byte LDS reads and 16-byte coalesced stores
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
        lse[((size_t)split * B + b) * n + i0 + threadIdx.x] =
            (lt > 0.f) ? (ms + fast_log2(lt)) * LN2_F : -INFINITY;
#ifdef CAB_ATTN_STAMPS
    K1_MARK(k1_t4)
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned long long* d = cab_stamps[blockIdx.x];
        d[0] = k1_t0, d[1] = k1_t1, d[2] = k1_t2, d[3] = k1_t3, d[4] = k1_t4, d[5] = k1_a, d[6] = k1_b;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Two waves per SIMD (n % 4 == 0).  The kernel above keeps Q, O, a whole K tile and a whole V tile of one wave in registers
// (one wave per SIMD) and hides the softmax by interleaving tile t + 1's statistics into tile t's PV slots; the matrix
// pipes are busy 58 % of its time.  This form gives the second wave of every SIMD that job:
//   * 8 waves split the keys of the workgroup's 32 queries; Q (64 registers) and O (64) stay resident;
//   * K and V travel in BATCHES of 16 registers, one per 16 MFMAs, double-buffered (batch b + 1 is requested in the MFMA
//     slots of batch b; the last batch of a tile requests the first one of the wave's next tile);
//   * the V operand of O^T += V P^T needs the channel on the lane.  At PV step r the B operand p[r] of lane (query, h) is key
//     acc_row(r) + 4h of the tile, so lane (channel, h) needs v[channel][j0 + 8 (r / 4) + 4h + r % 4]: four consecutive keys
//     per r / 4, i.e. ONE 16-byte load serves four steps -- no LDS transpose image, no LDS at all in the loop;
//   * per tile: S chain, softmax (plain, in the gap the other wave fills), O rescale if the maximum moved, PV.
// LDS is used once, for the merge of the eight partial (m, l, O) results.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// an MFMA has no side effect: without this the instruction selector may sink a chain below the loads that share its slots
// ("+v": a 512-thread kernel keeps its accumulators in the VGPR half)
__device__ __forceinline__ void pin(f32x16& acc) { asm volatile("" : "+v"(acc)); }
__device__ __forceinline__ f32x4 bload4(buf_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}

// PROJ (round 5): the CAB's output projection (reference cab.py:155, project_out: a bias-free 1x1 convolution
// Co x VC) is applied to the merged 32-query context tile while it is in LDS -- glob[co][i] = sum_c W[co][c] ctx[c][i], wave w owns
// output channels 32w .. 32w+31 (+256 per further block), A = W rows as 16-byte quads straight from L2 (requested before the merge
// barriers), B = the tile from LDS (row pitch PROJ_STR: the two k-halves of a step land in different bank halves).  The separate
// small-GEMM launch (18 us at config 3, an 8 MB round trip of ctx) becomes 64 MFMAs per wave behind the merge; ctx itself is only
// written when the caller wants it for the backward (ctx != nullptr).
constexpr int PROJ_STR = 40;
template <int KC, int VC, bool PROJ = false>
__global__ __launch_bounds__(512) void cab_attn_fwd_w8_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ ctx, float* __restrict__ lse, int n, float qscale, int kvsplit, int B,
    const float* __restrict__ w_out, float* __restrict__ glob, int Co) {
    constexpr int KB = KC / 32, VB = VC / 32, NW = 8, NBATCH = KB + VB;
    static_assert(NBATCH % 2 == 0, "the buffer parity of batch 0 must repeat from tile to tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    float* s_m = smem + NW * VC * 32;  // [8][32] running max per wave / query
    float* s_l = s_m + NW * 32;        // [8][32] running sum
    float* s_f = s_l + NW * 32;        // [8][32] merge factors

    const int nqt = (n + 31) >> 5, per_img = nqt * kvsplit;
    const int tile = xcd_chunked_tile(blockIdx.x, per_img * B);
    const int b = tile / per_img, rem = tile - b * per_img, split = rem / nqt, i0 = (rem - split * nqt) * 32;
    const int row_bytes = n * 4;
    const buf_rsrc q_rs = make_rsrc(q + (size_t)b * KC * n, (unsigned)KC * row_bytes);
    const buf_rsrc k_rs = make_rsrc(k + (size_t)b * KC * n, (unsigned)KC * row_bytes);
    const buf_rsrc v_rs = make_rsrc(v + (size_t)b * VC * n, (unsigned)VC * row_bytes);
    const int NT = (n + 31) >> 5, tstep = NW * kvsplit, t0 = split * NW + wave;
    const int lin4 = li * n * 4;

    f32x4 buf[2][4];
    int voff_k, voff_kn, cq[4];
    auto chain_off = [&](int t) { return (h * n + min(min(t, NT - 1) * 32 + li, n - 1)) * 4; };
    auto prefetch = [&](auto kn_tag, auto u_tag) {
        constexpr int KN = decltype(kn_tag)::value, U = decltype(u_tag)::value, P = KN & 1;
        if constexpr (KN < KB) {
            buf[P][U >> 2][U & 3] = bload(k_rs, voff_k, (32 * KN + 2 * U) * row_bytes);
        } else if constexpr (KN < NBATCH) {
            if constexpr (U < 4) buf[P][U] = bload4(v_rs, lin4 + cq[U] * 4, (KN - KB) * 32 * row_bytes);
        } else {
            buf[P][U >> 2][U & 3] = bload(k_rs, voff_kn, 2 * U * row_bytes);
        }
    };
    voff_k = chain_off(t0);
    static_for<0, 16>([&](auto u) { prefetch(std::integral_constant<int, 0>{}, u); });
    float qreg[KC / 2];  // raw q[2s + h][i0 + li]; scale * log2(e) is applied inside the softmax
    {
        const int voff = (h * n + min(i0 + li, n - 1)) * 4;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) qreg[s] = bload(q_rs, voff, s * 2 * row_bytes);
    }
    f32x16 o[VB];
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int t = t0; t < NT; t += tstep) {
        const int j0 = t * 32;
        voff_k = chain_off(t), voff_kn = chain_off(t + tstep);
#pragma unroll
        for (int a = 0; a < 4; ++a) cq[a] = min(j0 + 8 * a + 4 * h, n - 4);
        f32x16 sn;
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[r] = 0.f;
        static_for<0, KC / 2>([&](auto c2_tag) {  // S^T = K^T Q: keys in accumulator rows, query on the lane
            constexpr int c2 = decltype(c2_tag)::value, K = c2 / 16, U = c2 % 16;
            sn = mfma32(buf[K & 1][U >> 2][U & 3], qreg[c2], sn);
            pin(sn);
            prefetch(std::integral_constant<int, K + 1>{}, std::integral_constant<int, U>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // online softmax, statistics per lane (= per query); rescaling deferred until the maximum grows by > 2^12
        if (j0 + 32 > n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[r] = (j0 + acc_row(r) + 4 * h >= n) ? -INFINITY : sn[r];
        }
        float mt = sn[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sn[r]);
        mt = fmaxf(mt, swap_half(mt)) * qscale;
        const float mn = (mt > m + kRescaleThreshold) ? mt : m;
        const float alpha = fast_exp2(m - mn);
        m = mn;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sn[r] = fast_exp2(fmaf(sn[r], qscale, -mn));
            rs += sn[r];
        }
        l = l * alpha + rs;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int cb = 0; cb < VB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
        }
        static_for<0, VB>([&](auto cb_tag) {  // O^T += V P^T: A = the transposed batch (channel on lane), B = P^T registers
            constexpr int cb = decltype(cb_tag)::value, K = KB + cb;
            static_for<0, 16>([&](auto r_tag) {
                constexpr int r = decltype(r_tag)::value;
                o[cb] = mfma32(buf[K & 1][r >> 2][r & 3], sn[r], o[cb]);
                pin(o[cb]);
                prefetch(std::integral_constant<int, K + 1>{}, r_tag);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }

    // ---- merge the 8 waves (disjoint key subsets) ----
    l += swap_half(l);
    if (h == 0) {
        s_m[wave * 32 + li] = m;
        s_l[wave * 32 + li] = l;
    }
    float* vs = smem + wave * (VC * 32);
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) vs[(cb * 32 + acc_row(r) + 4 * h) * 32 + li] = o[cb][r];
    // PROJ: the weight rows of this wave's first output block travel while the merge runs (MFMA step (j, e) contracts
    // channel 8j + 4h + e: a lane's operands are 16-byte quads of ITS row)
    f32x4 wq[PROJ ? VC / 8 : 1];
    if constexpr (PROJ) {
        const float* wrow = w_out + (size_t)(wave * 32 + li) * VC + 4 * h;
#pragma unroll
        for (int j = 0; j < VC / 8; ++j) wq[j] = *reinterpret_cast<const f32x4*>(wrow + 8 * j);
    }
    __syncthreads();
    float ms = -INFINITY, lt = 0.f;
    if (threadIdx.x < NW * 32) {
        const int i = threadIdx.x & 31, w = threadIdx.x >> 5;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) ms = fmaxf(ms, s_m[ww * 32 + i]);
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float mw = s_m[ww * 32 + i];
            lt += (mw == -INFINITY) ? 0.f : s_l[ww * 32 + i] * fast_exp2(mw - ms);  // a wave without a key tile: m = -inf, l = 0
        }
        const float mw = s_m[w * 32 + i];
        s_f[w * 32 + i] = (lt > 0.f && mw != -INFINITY) ? fast_exp2(mw - ms) / lt : 0.f;
    }
    __syncthreads();
    const size_t out_base = ((size_t)split * B + b) * VC * n;
    float* mg = s_f + NW * 32;  // PROJ: [VC][PROJ_STR] merged context tile
    for (int idx = threadIdx.x; idx < VC * 8; idx += 512) {  // four consecutive queries per thread (n % 4 == 0)
        const int c = idx >> 3, i = (idx & 7) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x4 ov = *reinterpret_cast<const f32x4*>(smem + w * (VC * 32) + c * 32 + i);
            const f32x4 fv = *reinterpret_cast<const f32x4*>(s_f + w * 32 + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += ov[e] * fv[e];
        }
        if constexpr (PROJ) {
            *reinterpret_cast<f32x4*>(mg + c * PROJ_STR + i) = acc;
            if (ctx != nullptr && i0 + i < n) *reinterpret_cast<f32x4*>(ctx + out_base + (size_t)c * n + i0 + i) = acc;
        } else {
            if (i0 + i < n) *reinterpret_cast<f32x4*>(ctx + out_base + (size_t)c * n + i0 + i) = acc;
        }
    }
    if (threadIdx.x < 32 && i0 + threadIdx.x < n)
        lse[((size_t)split * B + b) * n + i0 + threadIdx.x] = (lt > 0.f) ? (ms + fast_log2(lt)) * LN2_F : -INFINITY;
    if constexpr (PROJ) {
        __syncthreads();  // the merged tile is complete
        const float* bt = mg + 4 * h * PROJ_STR + li;
        for (int cob = wave; cob * 32 < Co; cob += NW) {
            if (cob != wave) {
                const float* wrow = w_out + (size_t)(cob * 32 + li) * VC + 4 * h;
#pragma unroll
                for (int j = 0; j < VC / 8; ++j) wq[j] = *reinterpret_cast<const f32x4*>(wrow + 8 * j);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < VC / 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(wq[j][e], bt[(8 * j + e) * PROJ_STR], acc);
            if (i0 + li < n) {
                float* dst = glob + ((size_t)b * Co + cob * 32 + 4 * h) * n + i0 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)acc_row(r) * n] = acc[r];
            }
        }
    }
}

// merge kvsplit partial results: lse = logsumexp_s(lse_s), ctx = sum_s exp(lse_s - lse) ctx_s
__global__ void cab_attn_fwd_merge_kernel(const float* __restrict__ part_ctx,
                                          const float* __restrict__ part_lse,
                                          float* __restrict__ ctx, float* __restrict__ lse,
                                          int B, int VC, int n, int kvsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
    if (i >= n) return;
    float mx = -INFINITY;
    for (int s = 0; s < kvsplit; ++s) mx = fmaxf(mx, part_lse[((size_t)s * B + b) * n + i]);
    float sum = 0.f;
    for (int s = 0; s < kvsplit; ++s) sum += expf(part_lse[((size_t)s * B + b) * n + i] - mx);
    const float total = mx + logf(sum);
    if (blockIdx.y == 0) lse[(size_t)b * n + i] = total;
    for (int c = blockIdx.y; c < VC; c += gridDim.y) {
        float acc = 0.f;
        for (int s = 0; s < kvsplit; ++s)
            acc += part_ctx[(((size_t)s * B + b) * VC + c) * n + i] *
                   expf(part_lse[((size_t)s * B + b) * n + i] - total);
        ctx[((size_t)b * VC + c) * n + i] = acc;
    }
}

// shared with the split-bf16 variants (cab_attn_bf16.hip)
void launch_attn_merge(const float* part_ctx, const float* part_lse, float* ctx, float* lse, int B, int VC, int n, int kvsplit,
                       hipStream_t stream) {
    const dim3 mgrid((n + 255) / 256, VC < 32 ? VC : 32, B);
    hipLaunchKernelGGL(cab_attn_fwd_merge_kernel, mgrid, dim3(256), 0, stream, part_ctx, part_lse, ctx, lse, B, VC, n, kvsplit);
}

template <int KC, int VC>
static hipError_t launch_fwd(const float* q, const float* k, const float* v, float scale, int B, int n,
                             float* ctx, float* lse, float* part_ctx, float* part_lse, int kvsplit,
                             hipStream_t stream) {
    const size_t lds = (size_t)(4 * VC * 33 + 3 * 128) * sizeof(float);
    auto kern = cab_attn_fwd_kernel<KC, VC>;
    const dim3 block(256);
    static lds_attr_mask attr_mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_mask); e != hipSuccess) return e;
    dim3 grid(((n + 31) / 32) * kvsplit * B);
    float* out_ctx = kvsplit == 1 ? ctx : part_ctx;
    float* out_lse = kvsplit == 1 ? lse : part_lse;
    if constexpr (KC <= 128) {  // (KC = 256: the resident Q alone is 128 registers of the 256 a wave has at two per SIMD)
      if ((n & 3) == 0) {  // two waves per SIMD: needs 16-byte aligned key quads
        auto kern8 = cab_attn_fwd_w8_kernel<KC, VC>;
        const size_t lds8 = (size_t)(8 * VC * 32 + 3 * 8 * 32) * sizeof(float);
        static lds_attr_mask mask8{0};
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern8), lds8, mask8); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern8, grid, dim3(512), lds8, stream, q, k, v, out_ctx, out_lse, n, scale * LOG2E_F, kvsplit, B,
                           static_cast<const float*>(nullptr), static_cast<float*>(nullptr), 0);
        if (kvsplit > 1) launch_attn_merge(part_ctx, part_lse, ctx, lse, B, VC, n, kvsplit, stream);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, q, k, v, out_ctx, out_lse, n, scale * LOG2E_F, kvsplit, B);
    if (kvsplit > 1) launch_attn_merge(part_ctx, part_lse, ctx, lse, B, VC, n, kvsplit, stream);
    return hipGetLastError();
}

// Workgroups per query tile (each takes 1/split of the keys; a merge pass combines the partial results).  One workgroup per CU
// runs at a time, so a launch lasts ceil(workgroups / 256) rounds: 272 query tiles (the un-tiled 8704-position validation frame)
// are TWO rounds of full-length workgroups with 240 CUs idle in the second; split 4 makes them five rounds of quarter length
// (554 -> ~400 us).  The choice minimises rounds / split plus a small charge per partial result the merge has to read.
int attn_fwd_kvsplit(int B, int n) {
    const int nt = (n + 31) / 32, wgs = nt * B;
    int best = 1;
    double best_cost = 1e30;
    double cost1 = 0.0;
    for (int s = 1; s <= 8; s *= 2) {
        if (s > 1 && s * 8 > nt) break;  // every wave of the 8-wave kernel keeps at least one key tile
        const double cost = (double)((wgs * s + 255) / 256) / (double)s + (s > 1 ? 0.04 * s : 0.0);
        if (s == 1) cost1 = cost;
        if (cost < best_cost - 1e-9) best = s, best_cost = cost;
    }
    // a split pays a partial-ctx round trip and a merge launch that this round model only charges roughly: take it only for a
    // clear win (>= 15 % fewer rounds).  B = 8 frames of n = 8704 would otherwise split 2 for 9 -> 8.58 rounds (5 %) at the
    // price of 71 MB of partial results -- a case nobody measured.
    if (best > 1 && best_cost > 0.85 * cost1) best = 1;
    return best;
}

// ---- K1 with the output projection in its epilogue (PROJ above): one workgroup must own a query tile for ALL keys (no key
// split: the projection needs the merged tile), the 8-wave kernel's shapes only
bool attn_fwd_proj_supported(int B, int Kc, int Vc, int Co, int n) {
    if (!((Kc == 128 && Vc == 128) || (Kc == 64 && Vc == 64))) return false;
    if (B <= 0 || n <= 0 || (n & 3) || Co < 32 || (Co & 31)) return false;
    return attn_fwd_kvsplit(B, n) == 1;
}

template <int KC, int VC>
static hipError_t launch_fwd_proj(const float* q, const float* k, const float* v, const float* w_out, float scale, int B, int Co,
                                  int n, float* ctx, float* glob, float* lse, hipStream_t stream) {
    auto kern = cab_attn_fwd_w8_kernel<KC, VC, true>;
    const size_t lds = (size_t)(8 * VC * 32 + 3 * 8 * 32 + VC * PROJ_STR) * sizeof(float);
    static lds_attr_mask mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, mask); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(((n + 31) / 32) * B), dim3(512), lds, stream, q, k, v, ctx, lse, n, scale * LOG2E_F, 1, B, w_out,
                       glob, Co);
    return hipGetLastError();
}

hipError_t attn_fwd_proj_dispatch(const float* q, const float* k, const float* v, const float* w_out, float scale, int B, int Kc,
                                  int Vc, int Co, int n, float* ctx, float* glob, float* lse, hipStream_t stream) {
    if (Kc == 128 && Vc == 128) return launch_fwd_proj<128, 128>(q, k, v, w_out, scale, B, Co, n, ctx, glob, lse, stream);
    if (Kc == 64 && Vc == 64) return launch_fwd_proj<64, 64>(q, k, v, w_out, scale, B, Co, n, ctx, glob, lse, stream);
    return hipErrorInvalidValue;
}

bool attn_shape_supported(int Kc, int Vc) {
    return (Kc == 128 && Vc == 128) || (Kc == 256 && Vc == 128) || (Kc == 64 && Vc == 64);
}

hipError_t attn_fwd_dispatch(const float* q, const float* k, const float* v, float scale, int B, int Kc,
                             int Vc, int n, float* ctx, float* lse, float* part_ctx, float* part_lse,
                             int kvsplit, hipStream_t stream) {
    if (Kc == 128 && Vc == 128)
        return launch_fwd<128, 128>(q, k, v, scale, B, n, ctx, lse, part_ctx, part_lse, kvsplit, stream);
    if (Kc == 256 && Vc == 128)
        return launch_fwd<256, 128>(q, k, v, scale, B, n, ctx, lse, part_ctx, part_lse, kvsplit, stream);
    if (Kc == 64 && Vc == 64)
        return launch_fwd<64, 64>(q, k, v, scale, B, n, ctx, lse, part_ctx, part_lse, kvsplit, stream);
    return hipErrorInvalidValue;
}

}  // namespace cabinet
