// Split-bf16 form of the FFM's forward product  z = W_s . fsp + U(W_c . low)   (reference src/models/cabinet.py:143-144 with
// the x4 bilinear upsample of cabinet.py:228-230 commuted through the 1x1 convolution).
//
// The fp32 form (gemm_kmajor_kernel, ffm.hip) runs on v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 matrix rate.  Here every fp32
// operand is split into NS bf16 pieces (cab_attn_bf16.hip explains the arithmetic: NS = 3 / six products per fp32 product is
// exact to 2^-26, NS = 2 / three products to 2^-17) and the product runs on v_mfma_f32_32x32x16_bf16.
//   * A = W_s (Co x Cs, a weight): packed ONCE per call into bf16 pieces in MFMA operand order (gemm_pack_weight_kernel); every
//     wave keeps the A fragments of its 32 output channels for the WHOLE contraction in registers (K = Cs = 128: 8 k-steps x
//     NS pieces x 16 B per lane) and the workgroup is PERSISTENT over pixel tiles, so A is fetched once per workgroup.
//   * B = fsp (Cs x P per image, fp32 in HBM, read exactly once): a tile of 128 pixels is split while it is staged, half a
//     contraction (64 channels) at a time, into an LDS image [piece][k-step][h][pixel] of 16-byte chunks that every wave reads
//     conflict-free as its MFMA B operand.  Splitting costs ~7 VALU instructions per element but every staged element feeds
//     256 output channels, so it amortises (unlike attention, where it needed a pack pass).
//   * tile = 256 channels x 128 pixels = ONE output row of the model's 128 x 128 map: the upsample term of the row is the
//     vertical lerp of two source rows (staged cooperatively into LDS one half-tile ahead) and two LDS reads per output.
//   * double-buffered half-tiles, one barrier each; 8 waves = 2 per SIMD (the other wave's MFMAs cover a wave's staging).
#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NS>
struct GemmSplitTerms;
template <>
struct GemmSplitTerms<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0};
    static constexpr int B[3] = {0, 1, 0};
};
template <>
struct GemmSplitTerms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int B[6] = {0, 2, 1, 0, 1, 0};
};

template <int NS>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&out)[NS]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = x[e];
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            const __bf16 pc = (__bf16)r;
            out[p][e] = pc;
            r -= (float)pc;
        }
    }
}

// W (M, ldw) row-major, columns [0, K) -> Wp[piece][ks][m][h][8]: element e = W[m][16 ks + 8 h + e]
template <int NS>
__global__ __launch_bounds__(256) void gemm_pack_weight_kernel(const float* __restrict__ w, int ldw, int M, int K,
                                                               u32x4* __restrict__ wp) {
    const int id = blockIdx.x * 256 + threadIdx.x, KS = K >> 4;
    if (id >= M * KS * 2) return;
    const int h = id & 1, m = (id >> 1) % M, ks = (id >> 1) / M;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = w[(size_t)m * ldw + 16 * ks + 8 * h + e];
    bf16x8 out[NS];
    split8<NS>(x, out);
#pragma unroll
    for (int p = 0; p < NS; ++p) wp[(((size_t)p * KS + ks) * M + m) * 2 + h] = __builtin_bit_cast(u32x4, out[p]);
}

struct GemmBf16Args {
    const u32x4* wp;    // packed A: [NS][KS][M][2] chunks
    const float* src;   // (B, K, P) fp32
    float* dst;         // (B, M, P)
    int M, P, B;
    const float* up_src;  // (B, M, Hl, 32): + bilinear upsample, tile = one output row (W == 128, Wl == 32)
    int Hl, W;
    float rh, rw;
};

// grid (workgroups per M block of 256, M / 256); 512 threads; KS = K / 16 (even)
template <int KS, int NS>
__global__ __launch_bounds__(512) void gemm_bf16_rowtile_kernel(GemmBf16Args a) {
    constexpr int KH = KS / 2, NT = 128;
    using T = GemmSplitTerms<NS>;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    // [2 buffers][NS][KH][2 (h)][128 pixels] chunks, then vrow [256][32] floats
    constexpr int BUF = NS * KH * 2 * NT;
    float* vrow = reinterpret_cast<float*>(lds + 2 * BUF);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int M = a.M, P = a.P, mblk = blockIdx.y * 256, m0 = mblk + wave * 32;

    // the two leading pieces of A stay in registers for the whole kernel; the third (NS = 3: used by ONE of the six products,
    // A2 . B0) is re-read per k-step from the 64 KB L2-resident image -- holding it cost 32 more registers and spilled
    constexpr int NSR = NS < 2 ? NS : 2;
    bf16x8 A[NSR][KS];
#pragma unroll
    for (int p = 0; p < NSR; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[p][ks] = __builtin_bit_cast(bf16x8, a.wp[(((size_t)p * KS + ks) * M + m0 + li) * 2 + h]);
    const u32x4* a2p = a.wp + (((size_t)2 * KS) * M + m0 + li) * 2 + h;  // piece 2, k-step 0 (stride 2 M chunks per k-step)

    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

    const int tiles_img = P / NT, ntiles = a.B * tiles_img;
    const int first = blockIdx.x, stride = gridDim.x;
    const int my_tiles = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
    const int Q = 2 * my_tiles;  // half-tiles

    // ---- B staging: two (pixel, h, k-step) items per thread and half-tile ----
    float sx[2][8];
    auto tile_of = [&](int q) { return first + (q >> 1) * stride; };
    auto stage_load = [&](int q) {
        const int t = tile_of(q), b = t / tiles_img, p0 = (t - b * tiles_img) * NT, kh = q & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = tid + 512 * j, p = id & 127, hh = (id >> 7) & 1, ksl = id >> 8;
            const float* s = a.src + ((size_t)b * (KS * 16) + 16 * (kh * KH + ksl) + 8 * hh) * P + p0 + p;
#pragma unroll
            for (int e = 0; e < 8; ++e) sx[j][e] = s[(size_t)e * P];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = tid + 512 * j, p = id & 127, hh = (id >> 7) & 1, ksl = id >> 8;
            bf16x8 out[NS];
            split8<NS>(sx[j], out);
#pragma unroll
            for (int pc = 0; pc < NS; ++pc)
                lds[buf * BUF + ((pc * KH + ksl) * 2 + hh) * NT + p] = __builtin_bit_cast(u32x4, out[pc]);
        }
    };
    // ---- upsample term: vertical lerp of the two source rows of this tile's output row, [256 channels][32 columns] ----
    f32x4 v0[4], v1[4];
    float vly = 0.f;
    auto vrow_load = [&](int q) {
        const int t = tile_of(q), b = t / tiles_img, oy = ((t - b * tiles_img) * NT) / a.W;
        int y0, y1;
        bilinear_taps(oy, a.rh, a.Hl, y0, y1, vly);
        const int m = tid >> 1, x0 = (tid & 1) * 16;
        const float* s = a.up_src + ((size_t)b * M + mblk + m) * ((size_t)a.Hl * 32) + x0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v0[i] = *reinterpret_cast<const f32x4*>(s + y0 * 32 + 4 * i);
            v1[i] = *reinterpret_cast<const f32x4*>(s + y1 * 32 + 4 * i);
        }
    };
    auto vrow_store = [&]() {
        const int m = tid >> 1, x0 = (tid & 1) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (1.f - vly) * v0[i][e] + vly * v1[i][e];
            *reinterpret_cast<f32x4*>(vrow + m * 32 + x0 + 4 * i) = o;
        }
    };

    if (Q > 0) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();
    for (int q = 0; q < Q; ++q) {
        const int buf = q & 1, kh = q & 1;
        if (q + 1 < Q) stage_load(q + 1);
        if (a.up_src && kh == 0) vrow_load(q);
        // ---- MFMAs of this half-tile ----
        bf16x8 a2[KH];
        if (NS == 3) {
#pragma unroll
            for (int ksl = 0; ksl < KH; ++ksl) a2[ksl] = __builtin_bit_cast(bf16x8, a2p[(size_t)(kh * KH + ksl) * M * 2]);
        }
#pragma unroll
        for (int ksl = 0; ksl < KH; ++ksl)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                bf16x8 bf[NS];
#pragma unroll
                for (int pc = 0; pc < NS; ++pc)
                    bf[pc] = __builtin_bit_cast(bf16x8, lds[buf * BUF + ((pc * KH + ksl) * 2 + h) * NT + cb * 32 + li]);
#pragma unroll
                for (int e = 0; e < T::N; ++e) {
                    const int ap = T::A[e];
                    bf16x8 av;
                    if (ap == 2)
                        av = a2[ksl];
                    else if (kh == 0)
                        av = A[ap < NSR ? ap : 0][ksl];
                    else
                        av = A[ap < NSR ? ap : 0][KH + ksl];
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bf[T::B[e]], acc[cb], 0, 0, 0);
                }
            }
        if (q + 1 < Q) stage_store(buf ^ 1);
        if (a.up_src && kh == 0) vrow_store();
        __syncthreads();
        if (kh == 1) {  // tile complete: store (and zero) the accumulators
            const int t = tile_of(q), b = t / tiles_img, p0 = (t - b * tiles_img) * NT;
            float* drow0 = a.dst + ((size_t)b * M + m0 + 4 * h) * P + p0 + li;
            if (a.up_src) {
                int x0[4], x1[4];
                float lx[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bilinear_taps(cb * 32 + li, a.rw, 32, x0[cb], x1[cb], lx[cb]);
                const float* vb = vrow + (wave * 32 + 4 * h) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* vr = vb + acc_row(r) * 32;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        __builtin_nontemporal_store(acc[cb][r] + ((1.f - lx[cb]) * vr[x0[cb]] + lx[cb] * vr[x1[cb]]),
                                                    drow0 + (size_t)acc_row(r) * P + cb * 32);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) __builtin_nontemporal_store(acc[cb][r], drow0 + (size_t)acc_row(r) * P + cb * 32);
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
            // vrow is rewritten during the NEXT tile's first half, after this barrier-free epilogue: the writers reach
            // vrow_store only after their own epilogue, but other waves may still be reading -> fence the tile boundary
            if (a.up_src) __syncthreads();
        }
    }
}

// true when the split-bf16 kernel serves the product D (M x P per image) = A (M x K) . src (K x P) [+ row-form upsample]
bool gemm_bf16_supported(int M, int K, int P, int W, int Wl, bool up) {
    return (M % 256) == 0 && K == 128 && (P % 128) == 0 && (!up || (W == 128 && Wl == 32));
}
size_t gemm_bf16_pack_bytes(int M, int K, int precision) { return align_up((size_t)(precision == 2 ? 3 : 2) * M * K * 2, 256); }

template <int NS>
static hipError_t gemm_bf16_launch(const float* w, int ldw, const float* src, float* dst, int B, int M, int K, int P,
                                   const float* up_src, int Hl, int W, float rh, float rw, void* pack, hipStream_t stream) {
    u32x4* wp = static_cast<u32x4*>(pack);
    hipLaunchKernelGGL((gemm_pack_weight_kernel<NS>), dim3(ceil_div(M * (K / 16) * 2, 256)), dim3(256), 0, stream, w, ldw, M, K, wp);
    GemmBf16Args a{wp, src, dst, M, P, B, up_src, Hl, W, rh, rw};
    const size_t lds = (size_t)2 * NS * 4 * 2 * 128 * 16 + (size_t)256 * 32 * 4;
    auto kern = gemm_bf16_rowtile_kernel<8, NS>;
    static lds_attr_mask attr_mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_mask); e != hipSuccess) return e;
    const int ntiles = B * (P / 128), mblocks = M / 256;
    int wgs = 256 / mblocks;  // one workgroup per CU (LDS: 128 KB for three pieces), persistent over the tiles
    if (wgs > ntiles) wgs = ntiles;
    hipLaunchKernelGGL(kern, dim3(wgs, mblocks), dim3(512), lds, stream, a);
    return hipGetLastError();
}

hipError_t gemm_bf16_run(int precision, const float* w, int ldw, const float* src, float* dst, int B, int M, int K, int P,
                         const float* up_src, int Hl, int W, float rh, float rw, void* pack, hipStream_t stream) {
    return precision == 2 ? gemm_bf16_launch<3>(w, ldw, src, dst, B, M, K, P, up_src, Hl, W, rh, rw, pack, stream)
                          : gemm_bf16_launch<2>(w, ldw, src, dst, B, M, K, P, up_src, Hl, W, rh, rw, pack, stream);
}

}  // namespace cabinet
