// K2 -- backward of the fused CAB attention core (reference src/models/cab.py:149-154
// as differentiated by autograd).  Given g = dL/dctx, with P recomputed from q, k, lse:
//   D_i   = sum_c g[c][i] ctx[c][i]
//   dv    = P^T-weighted sum of g          dv[c][j] = sum_i P[i][j] g[c][i]
//   dP    = g^T v                          dS[i][j] = P[i][j] (dP[i][j] - D_i)
//   dq    = scale * dS k                   dk[c][j] = scale * sum_i dS[i][j] q[c][i]
//
// Conditioning of dq.  sum_j dS[i][j] = 0 exactly, so dq_i = scale * sum_j dS[i][j] (k_j - c) for ANY vector c.  With the
// flash-style D_i (from g and the stored ctx) the rounding error of D_i enters dq as -dD_i * sum_j P[i][j] k_j ~ -dD_i * mean(k):
// small per element (~5e-6 relative) but COHERENT over queries and channels, so it adds up in everything contracted
// over positions downstream (dW_q, d(beta_q): 4e-3 relative in a model whose keys have a large common component and
// near-uniform attention, while the elementwise dq error was 3e-6).  K2a therefore feeds the dq product with CENTRED keys
// k_j - mean_j(k_j) (subtracted where the K tile is laid down as the product's LDS image; S keeps the raw keys, so P is
// bit-identical): the same algebraic value with the coherent term and the large common-mode partial sums removed
// (dq error vs fp64 5e-6 -> 4e-7, below the autograd formulation's 1e-6).  mean_j(k_j) comes from a tiny pre-kernel.
//
// Default form (n % 4 == 0 and B*n*n*4 <= DS_MAX_BYTES), no atomics, bitwise deterministic -- S and dP are computed ONCE:
//   prep      D_i = sum_c g ctx, and the mean key                                            (attn_bwd_prep_kernel)
//   dk, dv    workgroup = 32 keys, 8 waves (two per SIMD) split the queries; S = Q^T K, dP = G^T V, dV += G P, dK += Q dS;
//             it also STORES dS (B*n*n floats, rows of 128 contiguous bytes)                  (cab_attn_bwd_dkdv_w8_kernel)
//   dq        = scale * dS (K - mean K)^T, one kernel over the stored dS, no partial slabs   (cab_attn_bwd_dq_ds_kernel)
// executed = algorithmic FLOPs, 2 n^2 (3Kc + 2Vc) per image.  Fallback forms that recompute S and dP for dq instead of
// storing dS (2 n^2 (4Kc + 3Vc) executed): ragged n and dS above DS_MAX_BYTES (cab_attn_bwd_dq_fast_kernel with the
// one-wave-per-SIMD cab_attn_bwd_dkdv_fast_kernel), and the generic chunk-staged pair cab_attn_bwd_dq_kernel /
// cab_attn_bwd_dkdv_kernel for (Kc, Vc) = (256, 128) in those cases.
//
// Layout trick shared with the forward kernel: the tile index that is NOT contracted
// sits on the lane, so every accumulator is directly the B operand of the next product
// and only the operand that must be read "channel on lane" crosses LDS, in 32-channel
// chunks through a small wave-private double buffer (no workgroup barrier in the loop).
#include <type_traits>

#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int TSTR = 33;            // padded row stride of a transposed chunk
constexpr int TCHUNK = 32 * TSTR;   // one 32-channel x 32-position chunk

// Stage a 32-channel chunk src[c0 .. c0+31][pos] (lane: pos = li, channel parity h) into
// a wave-private LDS chunk laid out [channel][pos] with stride 33.
__device__ __forceinline__ void stage_chunk(float* tb, const float* __restrict__ src, size_t row_stride,
                                            int li, int h) {
#pragma unroll
    for (int s = 0; s < 16; ++s) tb[(2 * s + h) * TSTR + li] = src[(size_t)(2 * s) * row_stride];
}

// Pre-pass of the stored-dS form, ONE launch with two roles:
//   blocks [0, nb_delta)  D_i = sum_c g[b][c][i] ctx[b][c][i] for 64 queries per workgroup (8 waves split the channels)
//   the rest              kmean[b][c] = mean_j k[b][c][j], one wave per row (any vector near the mean serves the identity
//                         of the header; fp32 is ample)
__global__ __launch_bounds__(512) void attn_bwd_prep_kernel(const float* __restrict__ g, const float* __restrict__ ctx,
                                                             const float* __restrict__ k, float* __restrict__ delta,
                                                             float* __restrict__ kmean, int VC, int rows, int n, int nb_delta) {
    __shared__ float part[8][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x < nb_delta) {
        const int tiles = (n + 63) >> 6, b = blockIdx.x / tiles, i = (blockIdx.x - b * tiles) * 64 + lane;
        float acc = 0.f;
        if (i < n) {
            const float* gp = g + (size_t)b * VC * n + i;
            const float* cp = ctx + (size_t)b * VC * n + i;
#pragma unroll 16
            for (int c = wave; c < VC; c += 8) acc += gp[(size_t)c * n] * cp[(size_t)c * n];
        }
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && i < n)
            delta[(size_t)b * n + i] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
                                       ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
        return;
    }
    const int row = (blockIdx.x - nb_delta) * 8 + wave;
    if (row >= rows) return;
    const float* p = k + (size_t)row * n;
    float a = 0.f;
#pragma unroll 4
    for (int j = lane; j < n; j += 64) a += p[j];
    a = wave_sum(a);
    if (lane == 0) kmean[row] = a / (float)n;
}

// kmean only (the recompute form, whose dq kernel forms D_i in its own prologue)
__global__ __launch_bounds__(256) void attn_key_mean_kernel(const float* __restrict__ k, float* __restrict__ kmean, int rows, int n) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = k + (size_t)row * n;
    float a = 0.f;
    for (int j = lane; j < n; j += 64) a += p[j];
    a = wave_sum(a);
    if (lane == 0) kmean[row] = a / (float)n;
}

// stage_chunk with a per-channel constant subtracted (the mean key of the chunk's 32 channels, in LDS)
__device__ __forceinline__ void stage_chunk_centred(float* tb, const float* __restrict__ src, size_t row_stride, int li,
                                                    int h, const float* __restrict__ cmean) {
#pragma unroll
    for (int s = 0; s < 16; ++s) tb[(2 * s + h) * TSTR + li] = src[(size_t)(2 * s) * row_stride] - cmean[2 * s + h];
}

// ------------------------------------------------------------------------------------ K2a: dq
template <int KC, int VC>
__global__ __launch_bounds__(256) void cab_attn_bwd_dq_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
    const float* __restrict__ v, const float* __restrict__ ctx, const float* __restrict__ lse,
    const float* __restrict__ kmean, float* __restrict__ dq, float* __restrict__ delta, int n, float scale) {
    constexpr int KB = KC / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qt = smem;                 // [KC][32]  q * scale*log2e   (B operand of S^T)
    float* gt = qt + KC * 32;         // [VC][32]  g                 (B operand of dP^T)
    float* tbuf = gt + VC * 32;       // [4 waves][2][TCHUNK]
    float* red = tbuf + 8 * TCHUNK;   // [KC][32]  cross-wave reduction of dq
    float* s_part = red + KC * 32;    // [8][32]   partial D_i
    float* s_delta = s_part + 256;    // [32]
    float* s_lse = s_delta + 32;      // [32]
    float* s_kbar = s_lse + 32;       // [KC]  mean key

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, i0 = blockIdx.x * 32;
    const size_t qk_base = (size_t)b * KC * n, v_base = (size_t)b * VC * n;
    const float qscale = scale * LOG2E_F;

    // ---- prologue: stage the query-side operands, compute D_i ----
    {
        const int i = threadIdx.x & 31, part = threadIdx.x >> 5;  // 8 parts over channels
        const int ig = min(i0 + i, n - 1);
        for (int c = part; c < KC; c += 8) qt[c * 32 + i] = q[qk_base + (size_t)c * n + ig] * qscale;
        float acc = 0.f;
        for (int c = part; c < VC; c += 8) {
            const float gv = g[v_base + (size_t)c * n + ig];
            gt[c * 32 + i] = gv;
            acc += gv * ctx[v_base + (size_t)c * n + ig];
        }
        s_part[part * 32 + i] = acc;
        for (int c = threadIdx.x; c < KC; c += 256) s_kbar[c] = kmean[b * KC + c];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float d = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) d += s_part[p * 32 + threadIdx.x];
        s_delta[threadIdx.x] = d;
        const int ig = i0 + threadIdx.x;
        if (ig < n) delta[(size_t)b * n + ig] = d;
        s_lse[threadIdx.x] = lse[(size_t)b * n + min(ig, n - 1)] * LOG2E_F;
    }
    __syncthreads();
    const float my_delta = s_delta[li], my_lse2 = s_lse[li];

    f32x16 acc[KB];
#pragma unroll
    for (int cb = 0; cb < KB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    float* tb = tbuf + wave * 2 * TCHUNK;

    const int NT = (n + 31) >> 5;
    for (int t = wave; t < NT; t += 4) {
        const int j0 = t * 32;
        const int jk = min(j0 + li, n - 1);
        const float* kp = k + qk_base + (size_t)h * n + jk;
        const float* vp = v + v_base + (size_t)h * n + jk;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) s = mfma32(kp[(size_t)(2 * c2) * n], qt[(2 * c2 + h) * 32 + li], s);
#pragma unroll
        for (int c2 = 0; c2 < VC / 2; ++c2) dp = mfma32(vp[(size_t)(2 * c2) * n], gt[(2 * c2 + h) * 32 + li], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool valid = j0 + acc_row(r) + 4 * h < n;
            const float p = valid ? fast_exp2(s[r] - my_lse2) : 0.f;
            s[r] = p * (dp[r] - my_delta);  // dS^T[key][query]
        }
        // dq^T[c][i] += sum_j k[c][j] dS^T[j][i] : A = k chunk (channel on lane) via LDS
#pragma unroll
        for (int cb = 0; cb < KB; ++cb) {
            float* tc = tb + (cb & 1) * TCHUNK;
            stage_chunk_centred(tc, kp + (size_t)(cb * 32) * n, n, li, h, s_kbar + cb * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[cb] = mfma32(tc[li * TSTR + acc_row(r) + 4 * h], s[r], acc[cb]);
        }
    }

    // ---- reduce the 4 waves' partial dq through LDS (ordered -> deterministic) ----
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int cb = 0; cb < KB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = (cb * 32 + acc_row(r) + 4 * h) * 32 + li;
                    red[idx] = (w == 0) ? acc[cb][r] : red[idx] + acc[cb][r];
                }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < KC * 32; idx += 256) {
        const int c = idx >> 5, i = idx & 31;
        if (i0 + i < n) dq[qk_base + (size_t)c * n + i0 + i] = red[idx] * scale;
    }
}

// ------------------------------------------------------------------------------- K2b: dk, dv
template <int KC, int VC>
__global__ __launch_bounds__(256) void cab_attn_bwd_dkdv_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
    const float* __restrict__ v, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ ds, int n, float scale) {
    constexpr int KB = KC / 32, VB = VC / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kt = smem;                  // [KC][32] raw k tile   (B operand of S)
    float* vt = kt + KC * 32;          // [VC][32] raw v tile   (B operand of dP)
    float* tbuf = vt + VC * 32;        // [4 waves][2][TCHUNK]
    float* red = tbuf + 8 * TCHUNK;    // [(KC+VC)][32]

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, j0 = blockIdx.x * 32;
    const size_t qk_base = (size_t)b * KC * n, v_base = (size_t)b * VC * n;
    const float qscale = scale * LOG2E_F;
    {
        const int j = threadIdx.x & 31, part = threadIdx.x >> 5;
        const int jg = min(j0 + j, n - 1);
        for (int c = part; c < KC; c += 8) kt[c * 32 + j] = k[qk_base + (size_t)c * n + jg];
        for (int c = part; c < VC; c += 8) vt[c * 32 + j] = v[v_base + (size_t)c * n + jg];
    }
    __syncthreads();

    f32x16 dka[KB], dva[VB];
#pragma unroll
    for (int cb = 0; cb < KB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dka[cb][r] = 0.f;
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dva[cb][r] = 0.f;
    float* tb = tbuf + wave * 2 * TCHUNK;

    const int NT = (n + 31) >> 5;
    for (int t = wave; t < NT; t += 4) {
        const int i0 = t * 32;
        const int ii = min(i0 + li, n - 1);
        const float* qp = q + qk_base + (size_t)h * n + ii;
        const float* gp = g + v_base + (size_t)h * n + ii;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
        // S[i][j] = q^T k : A = q (query on lane), B = k tile (key on lane) -> key on the lane
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) s = mfma32(qp[(size_t)(2 * c2) * n], kt[(2 * c2 + h) * 32 + li], s);
#pragma unroll
        for (int c2 = 0; c2 < VC / 2; ++c2) dp = mfma32(gp[(size_t)(2 * c2) * n], vt[(2 * c2 + h) * 32 + li], dp);
        f32x16 p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int irow = i0 + acc_row(r) + 4 * h;
            const bool valid = irow < n;
            const size_t si = (size_t)b * n + min(irow, n - 1);
            const float l2 = lse[si] * LOG2E_F, dl = delta[si];
            p[r] = valid ? fast_exp2(s[r] * qscale - l2) : 0.f;  // P[query][key]
            s[r] = p[r] * (dp[r] - dl);                           // dS[query][key]
            if (ds && valid && j0 + li < n) ds[((size_t)b * n + irow) * n + j0 + li] = s[r];  // rows of 128 contiguous bytes
        }
        // dv[c][j] += sum_i g[c][i] P[i][j] ; dk[c][j] += sum_i q[c][i] dS[i][j]
#pragma unroll
        for (int cb = 0; cb < VB; ++cb) {
            float* tc = tb + (cb & 1) * TCHUNK;
            stage_chunk(tc, gp + (size_t)(cb * 32) * n, n, li, h);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dva[cb] = mfma32(tc[li * TSTR + acc_row(r) + 4 * h], p[r], dva[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < KB; ++cb) {
            float* tc = tb + ((cb + VB) & 1) * TCHUNK;
            stage_chunk(tc, qp + (size_t)(cb * 32) * n, n, li, h);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dka[cb] = mfma32(tc[li * TSTR + acc_row(r) + 4 * h], s[r], dka[cb]);
        }
    }

    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int cb = 0; cb < KB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = (cb * 32 + acc_row(r) + 4 * h) * 32 + li;
                    red[idx] = (w == 0) ? dka[cb][r] : red[idx] + dka[cb][r];
                }
#pragma unroll
            for (int cb = 0; cb < VB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = (KC + cb * 32 + acc_row(r) + 4 * h) * 32 + li;
                    red[idx] = (w == 0) ? dva[cb][r] : red[idx] + dva[cb][r];
                }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < KC * 32; idx += 256) {
        const int c = idx >> 5, j = idx & 31;
        if (j0 + j < n) dk[qk_base + (size_t)c * n + j0 + j] = red[idx] * scale;
    }
    for (int idx = threadIdx.x; idx < VC * 32; idx += 256) {
        const int c = idx >> 5, j = idx & 31;
        if (j0 + j < n) dv[v_base + (size_t)c * n + j0 + j] = red[KC * 32 + idx];
    }
}

// =====================================================================================================
// Fast path (Kc <= 128 and Kc + Vc <= 256, i.e. what CABiNet itself uses): whole-tile transposed images
// in LDS (XOR-swizzled, unpadded: element (c,pos) at c*32 + (pos ^ (c&31)), conflict-free for the
// position-on-lane write and the channel-on-lane read), operands prefetched one tile ahead into
// registers with scalar-offset buffer loads, XCD-chunked tile order.  No global value is read twice by a
// wave and no load latency sits between MFMA chains.
// =====================================================================================================
__device__ __forceinline__ constexpr int simg(int c, int pos) { return c * 32 + (pos ^ (c & 31)); }

// The wave issues in order: anything placed after a group of MFMAs runs with the matrix pipe idle.  Both
// helpers therefore emit ONE MFMA per fenced slot, with the LDS operand of a later slot and the caller's
// `extra(slot)` work (LDS writes, prefetch loads) issued right behind it.
//   acc += sum_c2 a[c2] (registers) * bread(c2) (LDS, read four slots ahead)
template <int N, typename BRead, typename Extra>
__device__ __forceinline__ void chain_regA_ldsB(f32x16& acc, const float* a, BRead bread, Extra extra) {
    float bq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bq[u] = bread(u);
#pragma unroll
    for (int c2 = 0; c2 < N; ++c2) {
        const float bcur = bq[c2 & 3];
        if (c2 + 4 < N) bq[c2 & 3] = bread(c2 + 4);
        acc = mfma32(a[c2], bcur, acc);
        extra(c2);
        __builtin_amdgcn_sched_barrier(0);
    }
}
//   accs[cb] += aread(cb, r) (LDS image, read one r-step ahead) * bvec[r]      for r < 16, cb < NB
template <int NB, typename ARead, typename Extra>
__device__ __forceinline__ void product_ldsA(f32x16* accs, ARead aread, const f32x16& bvec, Extra extra) {
    float va[NB], vb[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) va[cb] = aread(cb, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            accs[cb] = mfma32(va[cb], bvec[r], accs[cb]);
            if (r + 1 < 16) vb[cb] = aread(cb, r + 1);
            extra(r * NB + cb);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) va[cb] = vb[cb];
    }
}

template <int KC, int VC>
__global__ __launch_bounds__(256) void cab_attn_bwd_dq_fast_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
    const float* __restrict__ v, const float* __restrict__ ctx, const float* __restrict__ lse,
    const float* __restrict__ kmean, float* __restrict__ dq, float* __restrict__ delta, int n, float scale, int B,
    int nsplit) {
    constexpr int KB = KC / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qt = smem;                   // [KC][32]  q * scale*log2e   (B operand of S^T)
    float* gt = qt + KC * 32;           // [VC][32]  g                 (B operand of dP^T)
    float* kimg = gt + VC * 32;         // [4 waves][KC*32] swizzled transposed K tile
    float* red = kimg + 4 * KC * 32;    // [KC][32]
    float* s_part = red + KC * 32;      // [8][32]
    float* s_delta = s_part + 256;      // [32]
    float* s_lse = s_delta + 32;        // [32]
    float* s_kbar = s_lse + 32;         // [KC]  mean key

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    // small grids (few query tiles) split the KEY range over nsplit workgroups per query tile; partial dq
    // slabs are summed by sum_parts_kernel.  Tile order: image-major, then split, then query tile.
    const int nqt = (n + 31) >> 5, per_img = nqt * nsplit;
    const int tile = xcd_chunked_tile(blockIdx.x, per_img * B);
    const int b = tile / per_img, rem = tile - b * per_img, split = rem / nqt, i0 = (rem - split * nqt) * 32;
    const size_t qk_base = (size_t)b * KC * n, v_base = (size_t)b * VC * n;
    const float qscale = scale * LOG2E_F;
    const int row_bytes = n * 4;
    const buf_rsrc k_rs = make_rsrc(k + qk_base, (unsigned)KC * row_bytes);
    const buf_rsrc v_rs = make_rsrc(v + v_base, (unsigned)VC * row_bytes);

    const int NT = (n + 31) >> 5;
    float kv[KC / 2], vv[VC / 2];
    auto load_kv = [&](int t) {
        const int voff = (h * n + min(min(t, NT - 1) * 32 + li, n - 1)) * 4;
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) kv[c2] = bload(k_rs, voff, c2 * 2 * row_bytes);
#pragma unroll
        for (int c2 = 0; c2 < VC / 2; ++c2) vv[c2] = bload(v_rs, voff, c2 * 2 * row_bytes);
    };
    const int tstep = 4 * nsplit, t0 = split * 4 + wave;
    load_kv(t0);  // in flight during the prologue

    {
        const int i = threadIdx.x & 31, part = threadIdx.x >> 5;
        const int ig = min(i0 + i, n - 1);
        for (int c = part; c < KC; c += 8) qt[c * 32 + i] = q[qk_base + (size_t)c * n + ig] * qscale;
        float acc = 0.f;
        for (int c = part; c < VC; c += 8) {
            const float gv = g[v_base + (size_t)c * n + ig];
            gt[c * 32 + i] = gv;
            acc += gv * ctx[v_base + (size_t)c * n + ig];
        }
        s_part[part * 32 + i] = acc;
        for (int c = threadIdx.x; c < KC; c += 256) s_kbar[c] = kmean[b * KC + c];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float d = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) d += s_part[p * 32 + threadIdx.x];
        s_delta[threadIdx.x] = d;
        const int ig = i0 + threadIdx.x;
        if (ig < n && split == 0) delta[(size_t)b * n + ig] = d;
        s_lse[threadIdx.x] = lse[(size_t)b * n + min(ig, n - 1)] * LOG2E_F;
    }
    __syncthreads();
    const float my_delta = s_delta[li], my_lse2 = s_lse[li];

    f32x16 acc[KB];
#pragma unroll
    for (int cb = 0; cb < KB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    float* kim = kimg + wave * (KC * 32);

    for (int t = t0; t < NT; t += tstep) {
        const int j0 = t * 32;
        const int voff_n = (h * n + min(min(t + tstep, NT - 1) * 32 + li, n - 1)) * 4;  // next tile
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
        // S^T chain; behind each MFMA the K row pair it used is laid down in the transposed image
        chain_regA_ldsB<KC / 2>(s, kv, [&](int c2) { return qt[(2 * c2 + h) * 32 + li]; },
                                [&](int c2) { kim[simg(2 * c2 + h, li)] = kv[c2] - s_kbar[2 * c2 + h]; });
        // dP^T chain; the K registers are free now: refill them with the next tile behind the MFMAs
        chain_regA_ldsB<VC / 2>(dp, vv, [&](int c2) { return gt[(2 * c2 + h) * 32 + li]; }, [&](int c2) {
#pragma unroll
            for (int u = c2 * (KC / 2) / (VC / 2); u < (c2 + 1) * (KC / 2) / (VC / 2); ++u)
                kv[u] = bload(k_rs, voff_n, u * 2 * row_bytes);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool valid = j0 + acc_row(r) + 4 * h < n;
            const float p = valid ? fast_exp2(s[r] - my_lse2) : 0.f;
            s[r] = p * (dp[r] - my_delta);  // dS^T[key][query]
        }
        // dq^T += K dS^T from the image; the V registers are refilled with the next tile meanwhile
        product_ldsA<KB>(acc, [&](int cb, int r) { return kim[simg(cb * 32 + li, acc_row(r) + 4 * h)]; }, s,
                         [&](int slot) {
#pragma unroll
                             for (int u = slot * (VC / 2) / (16 * KB); u < (slot + 1) * (VC / 2) / (16 * KB); ++u)
                                 vv[u] = bload(v_rs, voff_n, u * 2 * row_bytes);
                         });
    }

    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int cb = 0; cb < KB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = (cb * 32 + acc_row(r) + 4 * h) * 32 + li;
                    red[idx] = (w == 0) ? acc[cb][r] : red[idx] + acc[cb][r];
                }
        }
        __syncthreads();
    }
    // nsplit == 1: dq is the final tensor; else a partial slab [split][B][KC][n] (scaled already)
    float* out = dq + (size_t)split * B * KC * n;
    for (int idx = threadIdx.x; idx < KC * 32; idx += 256) {
        const int c = idx >> 5, i = idx & 31;
        if (i0 + i < n) out[qk_base + (size_t)c * n + i0 + i] = red[idx] * scale;
    }
}

// out[i] = sum_s part[s][i]  (ordered -> deterministic)
__global__ void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, size_t count, int nsplit) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * count + i];
    out[i] = s;
}

// DK = false: dk is NOT accumulated here (no q image, no dk registers) -- for Kc = 256 the 128 extra accumulators and the
// 128 KB of q images do not fit; dk = scale * q dS is then a second product over the stored dS, like dq (launch_bwd_fast).
template <int KC, int VC, bool DK>
__global__ __launch_bounds__(256) void cab_attn_bwd_dkdv_fast_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
    const float* __restrict__ v, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ ds, int n, float scale, int B, int nsplit) {
    constexpr int KB = KC / 32, VB = VC / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kt = smem;                    // [KC][32] raw k tile   (B operand of S)
    float* vt = kt + KC * 32;            // [VC][32] raw v tile   (B operand of dP)
    float* img = vt + VC * 32;           // [4 waves][(KC+VC)*32] swizzled transposed q | g tiles (DK; else g only)
    float* red = img;                    // reduction scratch aliases the images after the main loop
    constexpr int IMGW = (DK ? KC + VC : VC) * 32, VOFF = DK ? KC : 0;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int nkt = (n + 31) >> 5, per_img = nkt * nsplit;
    const int tile = xcd_chunked_tile(blockIdx.x, per_img * B);
    const int b = tile / per_img, rem = tile - b * per_img, split = rem / nkt, j0 = (rem - split * nkt) * 32;
    const int tstep = 4 * nsplit, t0 = split * 4 + wave;
    const size_t qk_base = (size_t)b * KC * n, v_base = (size_t)b * VC * n;
    const float qscale = scale * LOG2E_F;
    const int row_bytes = n * 4;
    const buf_rsrc q_rs = make_rsrc(q + qk_base, (unsigned)KC * row_bytes);
    const buf_rsrc g_rs = make_rsrc(g + v_base, (unsigned)VC * row_bytes);

    const int NT = (n + 31) >> 5;
    float qv[KC / 2], gv[VC / 2];
    auto load_qg = [&](int t) {
        const int voff = (h * n + min(min(t, NT - 1) * 32 + li, n - 1)) * 4;
#pragma unroll
        for (int c2 = 0; c2 < KC / 2; ++c2) qv[c2] = bload(q_rs, voff, c2 * 2 * row_bytes);
#pragma unroll
        for (int c2 = 0; c2 < VC / 2; ++c2) gv[c2] = bload(g_rs, voff, c2 * 2 * row_bytes);
    };
    load_qg(t0);
    {
        const int j = threadIdx.x & 31, part = threadIdx.x >> 5;
        const int jg = min(j0 + j, n - 1);
        for (int c = part; c < KC; c += 8) kt[c * 32 + j] = k[qk_base + (size_t)c * n + jg];
        for (int c = part; c < VC; c += 8) vt[c * 32 + j] = v[v_base + (size_t)c * n + jg];
    }
    __syncthreads();

    f32x16 dka[DK ? KB : 1], dva[VB];
#pragma unroll
    for (int cb = 0; cb < (DK ? KB : 1); ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dka[cb][r] = 0.f;
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dva[cb][r] = 0.f;
    float* qim = img + wave * IMGW;
    float* gim = qim + (DK ? KC * 32 : 0);

    for (int t = t0; t < NT; t += tstep) {
        const int i0 = t * 32;
        const int voff_n = (h * n + min(min(t + tstep, NT - 1) * 32 + li, n - 1)) * 4;  // next query tile
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
        // row constants of this query tile (16 rows per lane half), issued early
        float l2[16], dl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t si = (size_t)b * n + min(i0 + acc_row(r) + 4 * h, n - 1);
            l2[r] = lse[si];
            dl[r] = delta[si];
        }
        // S chain (A = q registers, B = k tile in LDS); q row pairs go to the transposed image behind it
        chain_regA_ldsB<KC / 2>(s, qv, [&](int c2) { return kt[(2 * c2 + h) * 32 + li]; },
                                [&](int c2) {
                                    if (DK) qim[simg(2 * c2 + h, li)] = qv[c2];
                                });
        // dP chain (A = g registers); g rows to its image, q registers refilled with the next tile
        chain_regA_ldsB<VC / 2>(dp, gv, [&](int c2) { return vt[(2 * c2 + h) * 32 + li]; }, [&](int c2) {
            gim[simg(2 * c2 + h, li)] = gv[c2];
#pragma unroll
            for (int u = c2 * (KC / 2) / (VC / 2); u < (c2 + 1) * (KC / 2) / (VC / 2); ++u)
                qv[u] = bload(q_rs, voff_n, u * 2 * row_bytes);
        });
        f32x16 p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool valid = i0 + acc_row(r) + 4 * h < n;
            p[r] = valid ? fast_exp2(fmaf(s[r], qscale, -l2[r] * LOG2E_F)) : 0.f;  // P[query][key]
            s[r] = p[r] * (dp[r] - dl[r]);                                          // dS[query][key]
        }
        if (ds && j0 + li < n) {  // dS leaves the kernel once, as rows of 128 contiguous bytes: dq = dS (K - mean K) is then ONE
            float* dsp = ds + ((size_t)b * n + i0 + 4 * h) * n + j0 + li;  // product instead of a second S / dP recomputation
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (i0 + acc_row(r) + 4 * h < n) dsp[(size_t)acc_row(r) * n] = s[r];
        }
        // dv += g P (g registers refilled with the next tile meanwhile), dk += q dS
        product_ldsA<VB>(dva, [&](int cb, int r) { return gim[simg(cb * 32 + li, acc_row(r) + 4 * h)]; }, p,
                         [&](int slot) {
#pragma unroll
                             for (int u = slot * (VC / 2) / (16 * VB); u < (slot + 1) * (VC / 2) / (16 * VB); ++u)
                                 gv[u] = bload(g_rs, voff_n, u * 2 * row_bytes);
                         });
        if constexpr (DK)
            product_ldsA<KB>(dka, [&](int cb, int r) { return qim[simg(cb * 32 + li, acc_row(r) + 4 * h)]; }, s,
                             [&](int) {});
    }

    __syncthreads();  // every wave is done with its images: reuse them as reduction scratch
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            if constexpr (DK) {
#pragma unroll
                for (int cb = 0; cb < KB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int idx = (cb * 32 + acc_row(r) + 4 * h) * 32 + li;
                        red[idx] = (w == 0) ? dka[cb][r] : red[idx] + dka[cb][r];
                    }
            }
#pragma unroll
            for (int cb = 0; cb < VB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = (VOFF + cb * 32 + acc_row(r) + 4 * h) * 32 + li;
                    red[idx] = (w == 0) ? dva[cb][r] : red[idx] + dva[cb][r];
                }
        }
        __syncthreads();
    }
    float* dk_out = dk + (size_t)split * B * KC * n;
    float* dv_out = dv + (size_t)split * B * VC * n;
    if constexpr (DK) {
        for (int idx = threadIdx.x; idx < KC * 32; idx += 256) {
            const int c = idx >> 5, j = idx & 31;
            if (j0 + j < n) dk_out[qk_base + (size_t)c * n + j0 + j] = red[idx] * scale;
        }
    }
    for (int idx = threadIdx.x; idx < VC * 32; idx += 256) {
        const int c = idx >> 5, j = idx & 31;
        if (j0 + j < n) dv_out[v_base + (size_t)c * n + j0 + j] = red[VOFF * 32 + idx];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two waves per SIMD.  The kernel above holds a whole query tile of q and g in 128 staging registers (416 in all, one wave
// per SIMD): its softmax, row-constant loads and dS stores run with the matrix pipe idle (59 % busy).  This form needs no
// transposed LDS images and 256 registers:
//   * the "channel on lane" operand of the dv / dk products is read STRAIGHT from global memory.  At product step r the
//     B operand p[r] of lane (key, h) is row acc_row(r) + 4h of the tile, so lane (channel, h) needs
//     g[channel][i0 + 8 (r / 4) + 4h + r % 4]: four consecutive floats per r / 4 -- one 16-byte buffer load serves four
//     product steps, and the four loads of a 32-channel block use every byte of the 32 cache lines they touch;
//   * every operand travels in BATCHES of 16 registers, one batch per 16 MFMAs (q and g chain batches, then the transposed
//     g and q blocks), double-buffered: batch K + 1 is loaded in the MFMA slots of batch K, the last batch of a tile loads
//     the first one of the wave's next tile.  q and g of an image (1 MB at config 3) stay in the XCD's L2;
//   * 8 waves split the queries of the workgroup's 32 keys; the second wave of each SIMD covers the other one's softmax.
// LDS holds only the raw k | v tiles; after the loop it is reused for a three-round tree sum of the eight partial results.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ f32x4 bload4(buf_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}

// An MFMA has no side effect, so instruction selection is free to sink a whole chain below the loads that share its slots
// (it did: every S / dP MFMA landed behind the last prefetch and 72 registers spilled).  An empty volatile asm that "modifies"
// the accumulator is ordered against the slot fences and emits nothing.
__device__ __forceinline__ void pin(f32x16& acc) { asm volatile("" : "+v"(acc)); }

template <int KC, int VC, bool DK>
__global__ __launch_bounds__(512) void cab_attn_bwd_dkdv_w8_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
    const float* __restrict__ v, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ ds, int n, float scale, int B, int nsplit) {
    constexpr int KB = KC / 32, VB = VC / 32, NW = 8;
    constexpr int QT0 = KB + 2 * VB;                 // first transposed q batch
    constexpr int NBATCH = QT0 + (DK ? KB : 0);      // q chain | g chain | g transposed | q transposed
    static_assert(NBATCH % 2 == 0, "the buffer parity of batch 0 must repeat from tile to tile");
    constexpr int IMG = (DK ? KC + VC : VC) * 32, VOFF = DK ? KC : 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kt = smem;            // [KC][32] raw k tile   (B operand of S)
    float* vt = kt + KC * 32;    // [VC][32] raw v tile   (B operand of dP)
    float* s_lse = vt + VC * 32; // [n4] log2(e) * lse of the image's queries, then [n4] D_i (n4 = n rounded up to 4)
    float* red = smem;           // [4][IMG] reduction scratch, aliases all of the above after the main loop

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int nkt = (n + 31) >> 5, per_img = nkt * nsplit;
    const int tile = xcd_chunked_tile(blockIdx.x, per_img * B);
    const int b = tile / per_img, rem = tile - b * per_img, split = rem / nkt, j0 = (rem - split * nkt) * 32;
    const int tstep = NW * nsplit, t0 = split * NW + wave;
    const size_t qk_base = (size_t)b * KC * n, v_base = (size_t)b * VC * n;
    const float qscale = scale * LOG2E_F;
    const int row_bytes = n * 4;
    const buf_rsrc q_rs = make_rsrc(q + qk_base, (unsigned)KC * row_bytes);
    const buf_rsrc g_rs = make_rsrc(g + v_base, (unsigned)VC * row_bytes);
    const int NT = (n + 31) >> 5, n4 = (n + 3) & ~3;
    float* s_delta = s_lse + n4;

    f32x4 buf[2][4];   // the two operand batches in flight
    int voff_c, voff_cn, cq[4];  // chain offsets of this / the next tile; first query of each transposed quad of this tile
    const int lin4 = li * n * 4;
    auto chain_off = [&](int t) { return (h * n + min(min(t, NT - 1) * 32 + li, n - 1)) * 4; };

    // element U of batch KN (KN == NBATCH: batch 0 of the wave's next tile)
    auto prefetch = [&](auto kn_tag, auto u_tag) {
        constexpr int KN = decltype(kn_tag)::value, U = decltype(u_tag)::value, P = KN & 1;
        if constexpr (KN < KB) {
            buf[P][U >> 2][U & 3] = bload(q_rs, voff_c, (32 * KN + 2 * U) * row_bytes);
        } else if constexpr (KN < KB + VB) {
            buf[P][U >> 2][U & 3] = bload(g_rs, voff_c, (32 * (KN - KB) + 2 * U) * row_bytes);
        } else if constexpr (KN < QT0) {
            if constexpr (U < 4) buf[P][U] = bload4(g_rs, lin4 + cq[U] * 4, (KN - KB - VB) * 32 * row_bytes);
        } else if constexpr (KN < NBATCH) {
            if constexpr (U < 4) buf[P][U] = bload4(q_rs, lin4 + cq[U] * 4, (KN - QT0) * 32 * row_bytes);
        } else {
            buf[P][U >> 2][U & 3] = bload(q_rs, voff_cn, 2 * U * row_bytes);
        }
    };

    voff_c = chain_off(t0);
    static_for<0, 16>([&](auto u) { prefetch(std::integral_constant<int, 0>{}, u); });
    {
        const int j = threadIdx.x & 31, part = threadIdx.x >> 5;
        const int jg = min(j0 + j, n - 1);
        for (int c = part; c < KC; c += 16) kt[c * 32 + j] = k[qk_base + (size_t)c * n + jg];
        for (int c = part; c < VC; c += 16) vt[c * 32 + j] = v[v_base + (size_t)c * n + jg];
        for (int i = threadIdx.x; i < n; i += 512) {  // row constants of every query of the image (8 n bytes)
            s_lse[i] = lse[(size_t)b * n + i] * LOG2E_F;
            s_delta[i] = delta[(size_t)b * n + i];
        }
    }
    __syncthreads();

    f32x16 dka[DK ? KB : 1], dva[VB];
#pragma unroll
    for (int cb = 0; cb < (DK ? KB : 1); ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dka[cb][r] = 0.f;
#pragma unroll
    for (int cb = 0; cb < VB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dva[cb][r] = 0.f;

    for (int t = t0; t < NT; t += tstep) {
        const int i0 = t * 32;
        voff_c = chain_off(t), voff_cn = chain_off(t + tstep);
#pragma unroll
        for (int a = 0; a < 4; ++a) cq[a] = min(i0 + 8 * a + 4 * h, n - 4);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
        // S = q^T k and dP = g^T v: A from the batch registers, B from the LDS tiles (read four slots ahead)
        {
            float bq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bq[u] = kt[(2 * u + h) * 32 + li];
            static_for<0, KC / 2>([&](auto c2_tag) {
                constexpr int c2 = decltype(c2_tag)::value, K = c2 / 16, U = c2 % 16;
                const float bcur = bq[c2 & 3];
                if constexpr (c2 + 4 < KC / 2) bq[c2 & 3] = kt[(2 * (c2 + 4) + h) * 32 + li];
                s = mfma32(buf[K & 1][U >> 2][U & 3], bcur, s);
                pin(s);
                prefetch(std::integral_constant<int, K + 1>{}, std::integral_constant<int, U>{});
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int u = 0; u < 4; ++u) bq[u] = vt[(2 * u + h) * 32 + li];
            static_for<0, VC / 2>([&](auto c2_tag) {
                constexpr int c2 = decltype(c2_tag)::value, K = KB + c2 / 16, U = c2 % 16;
                const float bcur = bq[c2 & 3];
                if constexpr (c2 + 4 < VC / 2) bq[c2 & 3] = vt[(2 * (c2 + 4) + h) * 32 + li];
                dp = mfma32(buf[K & 1][U >> 2][U & 3], bcur, dp);
                pin(dp);
                prefetch(std::integral_constant<int, K + 1>{}, std::integral_constant<int, U>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        f32x16 p;
#pragma unroll
        for (int a = 0; a < 4; ++a) {  // the four rows of quad a: one broadcast 16-byte LDS read per constant
            const f32x4 l2 = *reinterpret_cast<const f32x4*>(s_lse + cq[a]);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(s_delta + cq[a]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * a + e;
                const bool valid = i0 + acc_row(r) + 4 * h < n;
                p[r] = valid ? fast_exp2(fmaf(s[r], qscale, -l2[e])) : 0.f;  // P[query][key]
                s[r] = p[r] * (dp[r] - dl[e]);                               // dS[query][key]
            }
        }
        if (ds && j0 + li < n) {
            float* dsp = ds + ((size_t)b * n + i0 + 4 * h) * n + j0 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (i0 + acc_row(r) + 4 * h < n) dsp[(size_t)acc_row(r) * n] = s[r];
        }
        // dv += g P, dk += q dS: A = the transposed batch (channel on lane), B = p / dS registers
        static_for<0, VB>([&](auto cb_tag) {
            constexpr int cb = decltype(cb_tag)::value, K = KB + VB + cb;
            static_for<0, 16>([&](auto r_tag) {
                constexpr int r = decltype(r_tag)::value;
                dva[cb] = mfma32(buf[K & 1][r >> 2][r & 3], p[r], dva[cb]);
                pin(dva[cb]);
                prefetch(std::integral_constant<int, K + 1>{}, r_tag);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr (DK) {
            static_for<0, KB>([&](auto cb_tag) {
                constexpr int cb = decltype(cb_tag)::value, K = QT0 + cb;
                static_for<0, 16>([&](auto r_tag) {
                    constexpr int r = decltype(r_tag)::value;
                    dka[cb] = mfma32(buf[K & 1][r >> 2][r & 3], s[r], dka[cb]);
                    pin(dka[cb]);
                    prefetch(std::integral_constant<int, K + 1>{}, r_tag);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
    }

    // ---- tree sum of the eight waves' partial results: 4..7 -> 0..3, 2..3 -> 0..1, 1 -> 0 (fixed order: deterministic) ----
    auto put = [&](float* dst) {
        if constexpr (DK) {
#pragma unroll
            for (int cb = 0; cb < KB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(cb * 32 + acc_row(r) + 4 * h) * 32 + li] = dka[cb][r];
        }
#pragma unroll
        for (int cb = 0; cb < VB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(VOFF + cb * 32 + acc_row(r) + 4 * h) * 32 + li] = dva[cb][r];
    };
    auto add = [&](const float* src) {
        if constexpr (DK) {
#pragma unroll
            for (int cb = 0; cb < KB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dka[cb][r] += src[(cb * 32 + acc_row(r) + 4 * h) * 32 + li];
        }
#pragma unroll
        for (int cb = 0; cb < VB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dva[cb][r] += src[(VOFF + cb * 32 + acc_row(r) + 4 * h) * 32 + li];
    };
    __syncthreads();  // every wave is done with the k | v tiles
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) put(red + (wave - half) * IMG);
        __syncthreads();
        if (wave < half) add(red + wave * IMG);
        __syncthreads();
    }
    if (wave == 0) put(red);
    __syncthreads();
    float* dk_out = dk + (size_t)split * B * KC * n;
    float* dv_out = dv + (size_t)split * B * VC * n;
    if constexpr (DK) {
        for (int idx = threadIdx.x; idx < KC * 32; idx += 512) {
            const int c = idx >> 5, j = idx & 31;
            if (j0 + j < n) dk_out[qk_base + (size_t)c * n + j0 + j] = red[idx] * scale;
        }
    }
    for (int idx = threadIdx.x; idx < VC * 32; idx += 512) {
        const int c = idx >> 5, j = idx & 31;
        if (j0 + j < n) dv_out[v_base + (size_t)c * n + j0 + j] = red[VOFF * 32 + idx];
    }
}

// workgroups per key tile (each takes 1/split of the queries; ordered partial slabs): minimises the number of ROUNDS of
// workgroups the chip needs, in units of a full-length workgroup (see attn_fwd_kvsplit), plus a charge for the slab traffic
static int bwd_nsplit(int B, int n) {
    const int nt = (n + 31) / 32, wgs = nt * B;
    int best = 1;
    double best_cost = 1e30;
    double cost1 = 0.0;
    for (int s = 1; s <= 8; s *= 2) {
        if (s > 1 && s * 8 > nt) break;  // every wave keeps at least one query tile
        const double cost = (double)((wgs * s + 255) / 256) / (double)s + (s > 1 ? 0.05 * s : 0.0);
        if (s == 1) cost1 = cost;
        if (cost < best_cost - 1e-9) best = s, best_cost = cost;
    }
    if (best > 1 && best_cost > 0.85 * cost1) best = 1;  // only for a clear win: the slabs and their ordered sum are not free
    return best;
}

// ---------------------------------------------------------------------------------------------------------------------
// dq from the stored dS, dq[b][c][i] = scale * sum_j (k[b][c][j] - mean_j k[b][c][.]) dS[b][i][j], as ONE kernel without LDS
// operands and without partial slabs.  Both operands have the contracted index j contiguous in memory (k rows, dS rows), so
// each lane loads four consecutive j with one 16-byte buffer load: lane (c, h) of A and lane (i, h) of B both take
// j8 + 4h .. + 3 of an eight-key octet, and MFMA step e of the octet contracts the key pair (j8 + e, j8 + 4 + e) -- any
// pairing serves as long as A and B agree.  Workgroup = 32 queries x NCB blocks of 32 channels; its 8 waves (two per SIMD)
// split the key range in groups of 32 keys (one cache line per row) prefetched a group ahead in registers, and their partial
// tiles meet in a three-round tree sum through LDS.  Replaces a key-split small-GEMM launch (31 us at
// config 3: 64 x 64 tiles, operands staged through LDS) plus the ordered sum of its four partial slabs (5 us).
template <int NCB>
__global__ __launch_bounds__(512) void cab_attn_bwd_dq_ds_kernel(const float* __restrict__ k, const float* __restrict__ kmean,
                                                                  const float* __restrict__ ds, float* __restrict__ dq, int KC,
                                                                  int n, float scale, int B) {
    constexpr int IMG = NCB * 32 * 32, NQ = NCB * 4, RS = 36, KIMG = NCB * 32 * RS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;  // [4][IMG] after the main loop; before it: [8 waves][KIMG] wave-private k images
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    float* kimg = smem + wave * KIMG;  // [NCB * 32 channels][32 keys], row stride 36 words (16-byte reads of 16 rows: no conflict)
    const int nqt = (n + 31) >> 5;
    const int tile = xcd_chunked_tile(blockIdx.x, nqt * B);
    const int b = tile / nqt, i0 = (tile - b * nqt) * 32, cb0 = blockIdx.y * NCB;
    const int row_bytes = n * 4;
    const int span = ((n + 255) >> 8) << 5;  // keys per wave: a multiple of the 32-key group
    const int jb = wave * span, je = min(n, jb + span);
    const int ng = je > jb ? (je - jb + 31) >> 5 : 0;
    const buf_rsrc k_rs = make_rsrc(k + ((size_t)b * KC + cb0 * 32) * n, (unsigned)(NCB * 32) * row_bytes);
    const int rows = min(32, n - i0);
    const buf_rsrc ds_rs = make_rsrc(ds + ((size_t)b * n + i0) * n, (unsigned)rows * row_bytes);
    const int b_off = min(li, rows - 1) * row_bytes;
    float kbar[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) kbar[cb] = kmean[b * KC + (cb0 + cb) * 32 + li];

    // A group = 32 keys = one 128-byte line of every row.
    //   k : COALESCED 16-byte loads (an instruction = 8 rows x 128 bytes) -> wave-private LDS image -> one 16-byte LDS read per
    //       (channel row, octet half).  Read per lane straight from global the same quads are a 32-line gather per instruction:
    //       that form measured 31 us, 25 us with three of four k gathers removed -- the L1's line rate, not latency, bound it.
    //   dS: four gathers per group (its rows are private to the workgroup and read once), a group ahead.
    f32x4 S[NQ], a[2][4], Bq[2][4];  // staged k quads of the NEXT group; k operand quads of this / the next step; dS quads
    const int srow = lane >> 3, squad = lane & 7;
    auto stage_load = [&](int g) {
        const int voff = (srow * n + min(jb + g * 32 + 4 * squad, n - 4)) * 4;  // clamped: masked at use where out of range
#pragma unroll
        for (int t = 0; t < NQ; ++t) S[t] = bload4(k_rs, voff, 8 * t * row_bytes);
    };
    auto stage_write = [&](auto t_tag) {
        constexpr int t = decltype(t_tag)::value;
        *reinterpret_cast<f32x4*>(kimg + (8 * t + srow) * RS + 4 * squad) = S[t];
    };
    auto read_a = [&](auto par_tag, auto cb_tag, auto o_tag) {
        constexpr int par = decltype(par_tag)::value, cb = decltype(cb_tag)::value, o = decltype(o_tag)::value;
        a[par][o] = *reinterpret_cast<const f32x4*>(kimg + (cb * 32 + li) * RS + (2 * o + h) * 4);
    };
    auto load_b = [&](auto p_tag, auto o_tag, int g) {
        constexpr int P = decltype(p_tag)::value, o = decltype(o_tag)::value;
        Bq[P][o] = bload4(ds_rs, b_off + min(jb + g * 32 + 8 * o + 4 * h, n - 4) * 4, 0);
    };
    f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

    // group g (dS buffer P): NCB steps of 16 MFMAs, one accumulator each.  Slots of a step: the next step's four k quads are
    // read from LDS; the LAST step of a group lays down the staged group g + 1 (slots 0-7), reads the first step's quads of
    // it (8-11) and requests group g + 2 from global memory (12-15) -- a whole group of lead.
    auto group = [&](auto p_tag, int g) {
        constexpr int P = decltype(p_tag)::value;
        static_for<0, NCB>([&](auto cb_tag) {
            constexpr int cb = decltype(cb_tag)::value, par = (P * NCB + cb) & 1;
            constexpr bool LAST = cb == NCB - 1;
            static_for<0, 4>([&](auto o_tag) {
                constexpr int o = decltype(o_tag)::value;
                const bool valid = jb + g * 32 + 8 * o + 4 * h < je;
                f32x4 av;
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = valid ? a[par][o][e] - kbar[cb] : 0.f;
                static_for<0, 4>([&](auto e_tag) {
                    constexpr int e = decltype(e_tag)::value, slot = 4 * o + e;
                    acc[cb] = mfma32(av[e], Bq[P][o][e], acc[cb]);
                    pin(acc[cb]);
                    if constexpr (cb == 0 && e == 0) load_b(std::integral_constant<int, 1 - P>{}, o_tag, g + 1);
                    if constexpr (!LAST && slot >= 4 && slot < 8)
                        read_a(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, LAST ? 0 : cb + 1>{},
                               std::integral_constant<int, slot - 4>{});
                    if constexpr (LAST) {
                        if constexpr (slot < 8)
                            static_for<slot * NQ / 8, (slot + 1) * NQ / 8>([&](auto t) { stage_write(t); });
                        else if constexpr (slot < 12)
                            read_a(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, 0>{},
                                   std::integral_constant<int, slot - 8>{});
                        else if constexpr (slot == 12)
                            stage_load(g + 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
    };
    stage_load(0);
    static_for<0, 4>([&](auto o) { load_b(std::integral_constant<int, 0>{}, o, 0); });
    static_for<0, NQ>([&](auto t) { stage_write(t); });
    stage_load(1);
    static_for<0, 4>([&](auto o) { read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, o); });
    for (int g = 0; g < ng; g += 2) {
        group(std::integral_constant<int, 0>{}, g);
        group(std::integral_constant<int, 1>{}, g + 1);  // all-masked when ng is odd
    }

    auto put = [&](float* dst) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(cb * 32 + acc_row(r) + 4 * h) * 32 + li] = acc[cb][r];
    };
    __syncthreads();  // the partial-result images alias the k images
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) put(red + (wave - half) * IMG);
        __syncthreads();
        if (wave < half) {
            const float* src = red + wave * IMG;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] += src[(cb * 32 + acc_row(r) + 4 * h) * 32 + li];
        }
        __syncthreads();
    }
    if (wave == 0) put(red);
    __syncthreads();
    float* out = dq + ((size_t)b * KC + cb0 * 32) * n + i0;
    for (int idx = threadIdx.x; idx < IMG; idx += 512) {
        const int c = idx >> 5, i = idx & 31;
        if (i0 + i < n) out[(size_t)c * n + i] = red[idx] * scale;
    }
}

template <int NCB>
static hipError_t launch_dq_ds_kernel(const float* k, const float* kmean, const float* ds, float scale, int B, int KC, int n,
                                      float* dq, hipStream_t stream) {
    auto fn = cab_attn_bwd_dq_ds_kernel<NCB>;
    const size_t lds = (size_t)8 * NCB * 32 * 36 * sizeof(float);  // eight wave-private k images (>= the four result images)
    static lds_attr_mask mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(fn), lds, mask); e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(((n + 31) / 32) * B, KC / 32 / NCB), dim3(512), lds, stream, k, kmean, ds, dq, KC, n, scale, B);
    return hipSuccess;
}
// channel blocks per workgroup: as many as still leave the chip covered (every workgroup re-reads its dS rows from L2)
static hipError_t launch_dq_direct(const float* k, const float* kmean, const float* ds, float scale, int B, int KC, int n,
                                   float* dq, hipStream_t stream) {
    const int tiles = ((n + 31) / 32) * B, blocks = KC / 32;
    if (blocks % 4 == 0 && tiles * (blocks / 4) >= 200) return launch_dq_ds_kernel<4>(k, kmean, ds, scale, B, KC, n, dq, stream);
    if (blocks % 2 == 0 && tiles * (blocks / 2) >= 200) return launch_dq_ds_kernel<2>(k, kmean, ds, scale, B, KC, n, dq, stream);
    return launch_dq_ds_kernel<1>(k, kmean, ds, scale, B, KC, n, dq, stream);
}

// segments of the contracted range of a product over the stored dS (one small-GEMM job each; partial slabs summed in order):
// a single job had one workgroup per CU walking 32 dependent chunks, latency-bound
static int dq_ksplit(int n) {
    int s = n / 256;
    s = s < 1 ? 1 : (s > 8 ? 8 : s);
    while (s > 1 && (n / s) % 4) --s;  // every segment keeps 16-byte aligned rows
    return s;
}
// dk from the stored dS: dk[b][c][j] = scale * sum_i q[b][c][i] dS[b][i][j]  (A = Q as stored, B = dS row-major), the query
// range cut into segments like the key range above.  Used where the dk accumulators do not fit the dk/dv kernel (Kc = 256).
static void launch_dk_from_ds(const float* q, const float* ds, float scale, int B, int KC, int n, float* dk, float* slabs,
                              hipStream_t stream) {
    const int ks = dq_ksplit(n), seg = (n / ks + 3) & ~3;
    SgJobs jobs{};
    jobs.n = ks;
    for (int i = 0; i < ks; ++i) {
        const int i0 = i * seg, il = (i == ks - 1) ? n - i0 : seg;
        SgJob& j = jobs.j[i];
        j.seg[0] = {q + i0, ds + (size_t)i0 * n, il, n};
        j.nseg = 1, j.lda = n, j.a_mmajor = 1, j.M = KC, j.P = n, j.dst = ks == 1 ? dk : slabs + (size_t)i * B * KC * n,
        j.dst_rows = KC;
        j.alpha = scale;
        j.a_img_stride = (size_t)KC * n;  // A = the queries of image b
    }
    sg_gemm(jobs, B, stream);
    if (ks > 1) {
        const size_t cq = (size_t)B * KC * n;
        hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((cq + 255) / 256)), dim3(256), 0, stream, slabs, dk, cq, ks);
    }
}

// the stored-dS form needs 16-byte aligned rows of K and dS for the small GEMM's vector loads -- and B*n*n floats of
// workspace: above DS_MAX_BYTES (n = 8704, the un-tiled UAVid validation frame, is 303 MB per image; n = 16k would be 1 GB
// per image) the recompute kernels are used instead, whose workspace is O(B*n) as in round 1
static constexpr size_t DS_MAX_BYTES = (size_t)512 << 20;
static bool use_ds_path(int B, int n) { return (n & 3) == 0 && (size_t)B * n * n * sizeof(float) <= DS_MAX_BYTES; }

template <int KC, int VC, bool DK = true>  // DK = false requires the stored-dS form (n % 4 == 0)
static hipError_t launch_bwd_fast(const float* g, const float* q, const float* k, const float* v,
                                  const float* ctx, const float* lse, float scale, int B, int n, float* dq,
                                  float* dk, float* dv, float* delta, hipStream_t stream) {
    float* kmean = delta + align_up((size_t)B * n, 64);
    const bool dsp = use_ds_path(B, n);
    if (!DK && !dsp) return hipErrorInvalidValue;
    if (dsp) {
        const int nb_delta = B * ((n + 63) / 64);
        hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(nb_delta + (B * KC + 7) / 8), dim3(512), 0, stream, g, ctx, k, delta,
                           kmean, VC, B * KC, n, nb_delta);
    } else {
        hipLaunchKernelGGL(attn_key_mean_kernel, dim3((B * KC + 3) / 4), dim3(256), 0, stream, k, kmean, B * KC, n);
    }
    const size_t lds_dq = (size_t)((KC + VC) * 32 + 4 * KC * 32 + KC * 32 + 256 + 64 + KC) * sizeof(float);
    const size_t lds_kv = (size_t)((KC + VC) * 32 + 4 * (DK ? KC + VC : VC) * 32) * sizeof(float);
    auto k_kv = cab_attn_bwd_dkdv_fast_kernel<KC, VC, DK>;
    static lds_attr_mask mask_kv{0}, mask_dq{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_kv), lds_kv, mask_kv); e != hipSuccess) return e;
    if constexpr (DK) {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(cab_attn_bwd_dq_fast_kernel<KC, VC>), lds_dq, mask_dq);
            e != hipSuccess)
            return e;
    }
    const int nsplit = bwd_nsplit(B, n);
    float* part_k = kmean + align_up((size_t)B * KC, 64);
    float* part_v = part_k + (size_t)(nsplit > 1 ? nsplit : 0) * B * KC * n;
    float* ds = dsp ? part_v + (size_t)(nsplit > 1 ? nsplit : 0) * B * VC * n : nullptr;
    dim3 grid(((n + 31) / 32) * B * nsplit);
    const size_t cq = (size_t)B * KC * n, cv = (size_t)B * VC * n;
    if constexpr (DK) {
        if (!dsp) {
            auto k_dq = cab_attn_bwd_dq_fast_kernel<KC, VC>;
            hipLaunchKernelGGL(k_dq, grid, dim3(256), lds_dq, stream, g, q, k, v, ctx, lse, kmean, nsplit == 1 ? dq : part_k,
                               delta, n, scale, B, nsplit);
            if (nsplit > 1)
                hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((cq + 255) / 256)), dim3(256), 0, stream, part_k, dq,
                                   cq, nsplit);
        }
    }
    // stored-dS form: the two-waves-per-SIMD kernel (16-byte transposed loads need n % 4 == 0, which that form implies)
    auto k_w8 = cab_attn_bwd_dkdv_w8_kernel<KC, VC, DK>;
    // k | v tiles + the image's row constants, reused as four partial-result images by the final tree sum
    const size_t w8_loop = ((size_t)(KC + VC) * 32 + 2 * (size_t)((n + 3) & ~3)) * sizeof(float);
    const size_t w8_tree = (size_t)4 * (DK ? KC + VC : VC) * 32 * sizeof(float);
    const size_t lds_w8 = w8_loop > w8_tree ? w8_loop : w8_tree;
    static lds_attr_mask mask_w8{0};
    if (dsp) {
        // the size depends on n: the attribute is set once per device, to the most the kernel can ever ask for
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_w8), 160 * 1024, mask_w8); e != hipSuccess) return e;
        if (lds_w8 > 160 * 1024) return hipErrorInvalidValue;
    }
    auto launch_kv = [&](float* dk_dst, float* dv_dst, int ns) {
        if (dsp)
            hipLaunchKernelGGL(k_w8, grid, dim3(512), lds_w8, stream, g, q, k, v, lse, delta, dk_dst, dv_dst, ds, n, scale, B, ns);
        else
            hipLaunchKernelGGL(k_kv, grid, dim3(256), lds_kv, stream, g, q, k, v, lse, delta, dk_dst, dv_dst, ds, n, scale, B, ns);
    };
    if (nsplit == 1) {
        launch_kv(dk, dv, 1);
    } else {
        // partial slabs live behind D_i and the mean key in the workspace: [nsplit][B][KC][n] then [nsplit][B][VC][n]
        launch_kv(part_k, part_v, nsplit);
        if (DK)
            hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((cq + 255) / 256)), dim3(256), 0, stream, part_k, dk, cq, nsplit);
        hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((cv + 255) / 256)), dim3(256), 0, stream, part_v, dv, cv, nsplit);
    }
    if (dsp) {
        if (hipError_t e = launch_dq_direct(k, kmean, ds, scale, B, KC, n, dq, stream); e != hipSuccess) return e;
        if (!DK) launch_dk_from_ds(q, ds, scale, B, KC, n, dk, ds + (size_t)B * n * n, stream);
    }
    return hipGetLastError();
}

// generic chunk-staged pair, recompute form only: it serves (Kc, Vc) = (256, 128) where the stored-dS form does not apply
// (ragged n, or dS above DS_MAX_BYTES); the dispatcher sends every other case to launch_bwd_fast
template <int KC, int VC>
static hipError_t launch_bwd(const float* g, const float* q, const float* k, const float* v, const float* ctx,
                             const float* lse, float scale, int B, int n, float* dq, float* dk, float* dv,
                             float* delta, hipStream_t stream) {
    float* kmean = delta + align_up((size_t)B * n, 64);
    hipLaunchKernelGGL(attn_key_mean_kernel, dim3((B * KC + 3) / 4), dim3(256), 0, stream, k, kmean, B * KC, n);
    const size_t lds_dq = (size_t)((KC + VC) * 32 + 8 * TCHUNK + KC * 32 + 256 + 64 + KC) * sizeof(float);
    const size_t lds_kv = (size_t)((KC + VC) * 32 + 8 * TCHUNK + (KC + VC) * 32) * sizeof(float);
    auto k_dq = cab_attn_bwd_dq_kernel<KC, VC>;
    auto k_kv = cab_attn_bwd_dkdv_kernel<KC, VC>;
    static lds_attr_mask mask_dq{0}, mask_kv{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_dq), lds_dq, mask_dq); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_kv), lds_kv, mask_kv); e != hipSuccess) return e;
    dim3 grid((n + 31) / 32, 1, B);
    hipLaunchKernelGGL(k_dq, grid, dim3(256), lds_dq, stream, g, q, k, v, ctx, lse, kmean, dq, delta, n, scale);
    hipLaunchKernelGGL(k_kv, grid, dim3(256), lds_kv, stream, g, q, k, v, lse, delta, dk, dv, (float*)nullptr, n, scale);
    return hipGetLastError();
}

size_t attn_bwd_workspace(int B, int Kc, int Vc, int n) {
    size_t bytes = (align_up((size_t)B * n, 64) + align_up((size_t)B * Kc, 64)) * sizeof(float);  // D_i, mean key
    const bool fast = (Kc <= 128 && Kc + Vc <= 256) || (Kc == 256 && Vc == 128 && use_ds_path(B, n));
    const int nsplit = fast ? bwd_nsplit(B, n) : 1;
    if (nsplit > 1) bytes += (size_t)nsplit * B * (Kc + Vc) * n * sizeof(float);  // partial dq|dk and dv slabs
    if (use_ds_path(B, n))  // dS, then the query-range slabs of dk for Kc = 256 (dq needs none: cab_attn_bwd_dq_ds_kernel)
        bytes += ((size_t)B * n * n + (size_t)(Kc == 256 ? 1 : 0) * (dq_ksplit(n) > 1 ? dq_ksplit(n) : 0) * B * Kc * n) *
                 sizeof(float);
    return align_up(bytes, 256);
}

hipError_t attn_bwd_dispatch(const float* dctx, const float* q, const float* k, const float* v,
                             const float* ctx, const float* lse, float scale, int B, int Kc, int Vc, int n,
                             float* dq, float* dk, float* dv, void* ws, hipStream_t stream) {
    float* delta = static_cast<float*>(ws);
    if (Kc == 128 && Vc == 128)
        return launch_bwd_fast<128, 128>(dctx, q, k, v, ctx, lse, scale, B, n, dq, dk, dv, delta, stream);
    if (Kc == 256 && Vc == 128 && use_ds_path(B, n))  // dk and dq both as products over the stored dS
        return launch_bwd_fast<256, 128, false>(dctx, q, k, v, ctx, lse, scale, B, n, dq, dk, dv, delta, stream);
    if (Kc == 256 && Vc == 128)
        return launch_bwd<256, 128>(dctx, q, k, v, ctx, lse, scale, B, n, dq, dk, dv, delta, stream);
    if (Kc == 64 && Vc == 64)
        return launch_bwd_fast<64, 64>(dctx, q, k, v, ctx, lse, scale, B, n, dq, dk, dv, delta, stream);
    return hipErrorInvalidValue;
}

}  // namespace cabinet
