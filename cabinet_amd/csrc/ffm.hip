// K3 / K4 -- Feature Fusion Module forward and backward on gfx950.
//
// Replaces reference src/models/cabinet.py:142-153 (+ ConvBNReLU, cabinet.py:42-44):
//   z = W_blk . cat(fsp, fcp)      1x1 conv == GEMM  (Co x Cin) x (Cin x B*H*W)
//   feat = relu(BN(z));  m = mean_hw(feat);  a = sigmoid(W2 relu(W1 m));  out = feat * (1 + a)
//
// HBM plan (fp32, P = H*W pixels per image; the concat is never materialised and feat is
// never written -- z is the one saved activation, feat is recomputed where needed):
//   fwd  : G1 GEMM      read fsp,fcp (Cin*P)   write z (Co*P)   + BN partial sums in epilogue
//          pool         read z                                   (needs global BN stats first)
//          gate         read z                  write out
//   bwd  : reduce       read dout, z            5 sums per (b,c)
//          small        SE-MLP backward, BN coefficient algebra  (one workgroup)
//          dz           read dout, z            write dz
//          G2 GEMM      read dz                 write dfsp, dfcp  (dX = W^T dz)
//          G3 GEMM      read dz, fsp, fcp       split-K partials -> dW_blk
// GEMMs run on v_mfma_f32_32x32x2_f32 (exact fp32).  Every operand is read with the
// contiguous (pixel or output-channel) index on the lane: G1/G2 are "K-major" products
// (A_t[k][m], B[k][p]) that need no transposition at all; G3 contracts over pixels, so both
// operands go through padded (stride 33) LDS images.
#include <stdlib.h>

#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

struct FfmShape {
    int B, Cs, Cc, Co, Cm, H, W;
};

// =====================================================================================
// G1 / G2:  D[m][p] = sum_k At[k][m] * Bm[k][p]   per image
// =====================================================================================


constexpr int GK_BK = 16;   // k-chunk
constexpr int GK_NT = 128;  // pixels per tile (2 waves x 2 blocks x 32)

// INTERIOR: every tile is full (P % 128 == 0, M % tile == 0, 16-byte aligned rows): the staging loads
// carry no guards, so nothing forces the compiler to wait for them before the MFMA block of the
// current chunk (the guarded form branches per load and drains vmcnt at every join).
// UP: 0 = no epilogue term, 1 = + bilinear upsample with the tile inside ONE output row (two source rows lerped into LDS),
// 2 = + bilinear upsample by four gathers per output (any geometry).  A template parameter, not a runtime branch: with the
// gather epilogue compiled into every instance the WM = 2 kernel needed 256 VGPRs and spilled 324 bytes per lane.
template <int WM, bool INTERIOR, int UP>  // 32-row blocks per wave along m; tile = 4 waves x WM x 32 rows
__global__ __launch_bounds__(512) void gemm_kmajor_kernel(GemmKArgs a) {
    constexpr int MT = 128 * WM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BK][MT]
    float* Bs = smem + 2 * GK_BK * MT;   // [2][BK][NT]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z, p0 = blockIdx.x * GK_NT, m0 = blockIdx.y * MT;
    const int P = a.P, M = a.M, K = a.K;
    const bool vec_ok = (P & 3) == 0;

    f32x16 acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging registers: WM float4 of A and one float4 of B per thread and chunk
    f32x4 ra[WM], rb;
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int idx = tid + i * 512, kk = idx / (MT / 4), c4 = idx % (MT / 4);
            const int m = m0 + c4 * 4;
            if (INTERIOR || m < M)
                ra[i] = *reinterpret_cast<const f32x4*>(a.at + (size_t)(k0 + kk) * a.lda + m);
            else
                ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        {
            const int kk = tid >> 5, c4 = tid & 31, k = min(k0 + kk, K - 1), p = p0 + c4 * 4;  // K tail: A rows are 0
            const float* row = (k < a.K0) ? a.src0 + ((size_t)b * a.K0 + k) * P
                                          : a.src1 + ((size_t)b * (K - a.K0) + (k - a.K0)) * P;
            if (INTERIOR || (vec_ok && p + 3 < P)) {
                rb = *reinterpret_cast<const f32x4*>(row + p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = (p + e < P) ? row[p + e] : 0.f;
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int idx = tid + i * 512;
            *reinterpret_cast<f32x4*>(As + (size_t)buf * GK_BK * MT + idx * 4) = ra[i];
        }
        *reinterpret_cast<f32x4*>(Bs + (size_t)buf * GK_BK * GK_NT + tid * 4) = rb;
    };

    const int nchunks = (K + GK_BK - 1) / GK_BK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_chunk((kc + 1) * GK_BK);
        const float* Ab = As + (size_t)buf * GK_BK * MT + wm * (WM * 32) + li;
        const float* Bb = Bs + (size_t)buf * GK_BK * GK_NT + wn * 64 + li;
#pragma unroll
        for (int kk = 0; kk < GK_BK; kk += 2) {
            float av[WM], bv[2];
#pragma unroll
            for (int i = 0; i < WM; ++i) av[i] = Ab[(kk + h) * MT + i * 32];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bb[(kk + h) * GK_NT + j * 32];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
        if (kc + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- optional epilogue term: + bilinear upsample of a low-resolution (B, M, Hl, Wl) map ----
    if (UP != 0) {
        const size_t plane = (size_t)a.Hl * a.Wl;
        if (UP == 1) {  // host guarantees: INTERIOR, W % 128 == 0, Wl == 32 (vrow = MT x 32 floats fits the staging LDS)
            // the tile is a 128-pixel segment of ONE output row: interpolate the two source rows vertically
            // into LDS once (coalesced), then every output needs two LDS reads instead of four L2 gathers
            float* vrow = smem;  // [MT][Wl], aliases the staging buffers (all waves are past the last barrier)
            const int oy = p0 / a.W, ox0 = p0 - oy * a.W;
            int y0, y1;
            float ly;
            bilinear_taps(oy, a.rh, a.Hl, y0, y1, ly);
#pragma unroll 2
            for (int idx = tid; idx < MT * 32; idx += 512) {  // Wl == 32 (see above)
                const int ml = idx >> 5, xs = idx & 31;
                const float* src = a.up_src + ((size_t)b * M + m0 + ml) * plane;
                vrow[idx] = (1.f - ly) * src[y0 * a.Wl + xs] + ly * src[y1 * a.Wl + xs];
            }
            __syncthreads();
            int x0[2], x1[2];
            float lx[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bilinear_taps(ox0 + wn * 64 + j * 32 + li, a.rw, a.Wl, x0[j], x1[j], lx[j]);
            // the accumulators live in the AGPR half of the register file: adding the upsample term IN PLACE made the
            // compiler copy all 64 of them to VGPRs (206 registers, one workgroup per CU); the term is therefore added
            // in the store loop, one value at a time (117 registers, two workgroups per CU)
            const float* v0[2] = {vrow + (wm * (WM * 32) + 4 * h) * 32 + x0[0], vrow + (wm * (WM * 32) + 4 * h) * 32 + x0[1]};
            const float* v1[2] = {vrow + (wm * (WM * 32) + 4 * h) * 32 + x1[0], vrow + (wm * (WM * 32) + 4 * h) * 32 + x1[1]};
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (WM * 32) + i * 32 + acc_row(r) + 4 * h;
                    float* drow = (m < a.M0) ? a.dst0 + ((size_t)b * a.M0 + m) * P
                                             : a.dst1 + ((size_t)b * (M - a.M0) + (m - a.M0)) * P;
                    const int off = (i * 32 + acc_row(r)) * 32;  // compile-time: an immediate LDS offset
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float val = acc[i][j][r] + ((1.f - lx[j]) * v0[j][off] + lx[j] * v1[j][off]);
                        float* dptr = drow + p0 + wn * 64 + j * 32 + li;
                        if (a.wt_store)
                            store_wt(dptr, val);  // write-through: no dirty line in the way of the operand stream (common.hpp)
                        else
                            __builtin_nontemporal_store(val, dptr);
                    }
                }
            }
            return;  // INTERIOR: every row and column of the tile exists, nothing left to store
        }
    }
    // general geometry (UP == 2): four gathers per output from the L2-resident low-resolution map.  Like the row form the
    // term is added in the store loop below, one value at a time: adding it to the accumulators in place made the compiler
    // copy all of them out of the AGPR half of the file (256 VGPRs and 492 .. 1200 bytes of scratch per lane).
    int o00[2] = {0, 0}, o01[2] = {0, 0}, o10[2] = {0, 0}, o11[2] = {0, 0};
    float w00[2] = {0.f, 0.f}, w01[2] = {0.f, 0.f}, w10[2] = {0.f, 0.f}, w11[2] = {0.f, 0.f};
    if (UP == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = min(p0 + wn * 64 + j * 32 + li, P - 1);
            const int oy = p / a.W, ox = p - oy * a.W;
            int y0, y1, x0, x1;
            float ly, lx;
            bilinear_taps(oy, a.rh, a.Hl, y0, y1, ly);
            bilinear_taps(ox, a.rw, a.Wl, x0, x1, lx);
            o00[j] = y0 * a.Wl + x0, o01[j] = y0 * a.Wl + x1, o10[j] = y1 * a.Wl + x0, o11[j] = y1 * a.Wl + x1;
            w00[j] = (1.f - ly) * (1.f - lx), w01[j] = (1.f - ly) * lx, w10[j] = ly * (1.f - lx), w11[j] = ly * lx;
        }
    }

    // ---- epilogue: store D (rows are 128-byte contiguous segments per lane half) ----
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (WM * 32) + i * 32 + acc_row(r) + 4 * h;
            if (m >= M) continue;
            float* drow = (m < a.M0) ? a.dst0 + ((size_t)b * a.M0 + m) * P
                                     : a.dst1 + ((size_t)b * (M - a.M0) + (m - a.M0)) * P;
            const float* src = UP == 2 ? a.up_src + ((size_t)b * M + m) * ((size_t)a.Hl * a.Wl) : nullptr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = p0 + wn * 64 + j * 32 + li;
                float val = acc[i][j][r];
                if (UP == 2)
                    val += w00[j] * src[o00[j]] + w01[j] * src[o01[j]] + w10[j] * src[o10[j]] + w11[j] * src[o11[j]];
                if (p < P) {
                    if (a.wt_store)
                        store_wt(drow + p, val);
                    else
                        __builtin_nontemporal_store(val, drow + p);
                }
            }
        }
    }
}

template <int WM, bool INTERIOR, int UP>
static hipError_t launch_gemm_k(const GemmKArgs& a, int B, hipStream_t stream) {
    constexpr int MT = 128 * WM;
    const size_t lds = (size_t)(2 * GK_BK * MT + 2 * GK_BK * GK_NT) * sizeof(float);
    static lds_attr_mask attr_mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_kmajor_kernel<WM, INTERIOR, UP>), lds, attr_mask);
        e != hipSuccess)
        return e;  // wrong device / LDS above the limit: report THIS, not the unrelated launch failure that would follow
    dim3 grid(ceil_div(a.P, GK_NT), ceil_div(a.M, MT), B);
    hipLaunchKernelGGL((gemm_kmajor_kernel<WM, INTERIOR, UP>), grid, dim3(512), lds, stream, a);
    return hipSuccess;
}

template <int WM>
static hipError_t launch_gemm_k_wm(const GemmKArgs& a, int B, bool interior, hipStream_t stream) {
    // the row form of the upsample epilogue needs full tiles that are segments of one output row, and room for vrow
    const bool row_up = a.up_src && interior && (a.W % GK_NT) == 0 && a.Wl == 32;
    if (!a.up_src) return interior ? launch_gemm_k<WM, true, 0>(a, B, stream) : launch_gemm_k<WM, false, 0>(a, B, stream);
    if (row_up) return launch_gemm_k<WM, true, 1>(a, B, stream);
    return interior ? launch_gemm_k<WM, true, 2>(a, B, stream) : launch_gemm_k<WM, false, 2>(a, B, stream);
}

hipError_t gemm_kmajor(const GemmKArgs& a_in, int B, hipStream_t stream) {
    GemmKArgs a = a_in;
    a.wt_store = stream_wt();
    int wm = a.M <= 128 ? 1 : (a.M > 256 && a.M <= 384) ? 3 : 2;
    if (wm > 1 && (a.M % 128) == 0 && ceil_div(a.P, GK_NT) * B * ceil_div(a.M, 128 * wm) < 200) wm = 1;  // small grid
    const bool interior = (a.P % GK_NT) == 0 && (a.M % (128 * wm)) == 0 && (a.lda % 4) == 0;
    if (wm == 1) return launch_gemm_k_wm<1>(a, B, interior, stream);
    if (wm == 3) return launch_gemm_k_wm<3>(a, B, interior, stream);
    return launch_gemm_k_wm<2>(a, B, interior, stream);
}

// =====================================================================================
// G3:  dW[o][c] = sum_{b,p} dz[b][o][p] * X[b][c][p]      (contraction over pixels, split-K)
// =====================================================================================
constexpr int G3_T = 128;    // output tile (o and c)
constexpr int G3_BK = 32;    // pixels per chunk
constexpr int G3_STR = 33;

template <bool INTERIOR>  // INTERIOR: P % 32 == 0, Co % 128 == 0, Cs % 128 == 0, Cc % 128 == 0 (see G1/G2)
__global__ __launch_bounds__(256) void gemm_dw_kernel(const float* __restrict__ dz, const float* __restrict__ fsp,
                                                       const float* __restrict__ fcp, float* __restrict__ part,
                                                       int B, int Co, int Cs, int Cc, int P, int chunks_per_img,
                                                       int chunks_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Az = smem;                         // [2][128][33]   dz rows (o) x pixels
    float* Bx = smem + 2 * G3_T * G3_STR;     // [2][128][33]   X rows (c) x pixels
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wo = wave >> 1, wc = wave & 1;
    // 1-D grid, XCD-chunked: the (o-tile, c-tile) blocks of ONE pixel range run on the same XCD, so the
    // dz rows re-read by the c-tiles and the X rows re-read by the o-tiles come out of that XCD's L2
    const int Cin = Cs + Cc;
    const int ntx = (Co + G3_T - 1) / G3_T, nty = (Cin + G3_T - 1) / G3_T;
    const int vt = xcd_chunked_tile(blockIdx.x, gridDim.x);
    const int split = vt / (ntx * nty), xy = vt - split * (ntx * nty);
    const int o0 = (xy % ntx) * G3_T, c0 = (xy / ntx) * G3_T;
    const int chunk_lo = split * chunks_per_split;
    const int chunk_hi = min(chunk_lo + chunks_per_split, B * chunks_per_img);
    const bool vec_ok = (P & 3) == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: each operand chunk is 128 rows x 32 px = 1024 float4 -> 4 per thread
    f32x4 rz[4], rx[4];
    auto load_chunk = [&](int chunk) {
        const int b = chunk / chunks_per_img, p0 = (chunk % chunks_per_img) * G3_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, p = p0 + (idx & 7) * 4;
            const int o = o0 + row, c = c0 + row;
            const float* zr = (o < Co) ? dz + ((size_t)b * Co + o) * P : nullptr;
            const float* xr = (c < Cs)    ? fsp + ((size_t)b * Cs + c) * P
                              : (c < Cin) ? fcp + ((size_t)b * Cc + (c - Cs)) * P
                                          : nullptr;
            if (INTERIOR) {
                rz[i] = *reinterpret_cast<const f32x4*>(dz + ((size_t)b * Co + o) * P + p);
                rx[i] = *reinterpret_cast<const f32x4*>(
                    (c0 < Cs ? fsp + ((size_t)b * Cs + c) * P : fcp + ((size_t)b * Cc + (c - Cs)) * P) + p);
            } else if (vec_ok && p + 3 < P) {
                rz[i] = zr ? *reinterpret_cast<const f32x4*>(zr + p) : f32x4{0.f, 0.f, 0.f, 0.f};
                rx[i] = xr ? *reinterpret_cast<const f32x4*>(xr + p) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rz[i][e] = (zr && p + e < P) ? zr[p + e] : 0.f;
                    rx[i][e] = (xr && p + e < P) ? xr[p + e] : 0.f;
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, col = (idx & 7) * 4;
            float* za = Az + (size_t)buf * G3_T * G3_STR + row * G3_STR + col;
            float* xa = Bx + (size_t)buf * G3_T * G3_STR + row * G3_STR + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                za[e] = rz[i][e];
                xa[e] = rx[i][e];
            }
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
        const float* Ab = Az + (size_t)buf * G3_T * G3_STR + (wo * 64 + li) * G3_STR + h;
        const float* Bb = Bx + (size_t)buf * G3_T * G3_STR + (wc * 64 + li) * G3_STR + h;
#pragma unroll
        for (int kk = 0; kk < G3_BK; kk += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = Ab[i * 32 * G3_STR + kk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bb[j * 32 * G3_STR + kk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
        if (chunk + 1 < chunk_hi) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // partial slab: part[split][Co][Cin]
    float* slab = part + (size_t)split * Co * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * 64 + i * 32 + acc_row(r) + 4 * h;
                const int c = c0 + wc * 64 + j * 32 + li;
                if (o < Co && c < Cin) slab[(size_t)o * Cin + c] = acc[i][j][r];
            }
}

// out[o * ldo + col_off + c] = sum_k part[k][o][c]   (slab sum into a column block of dW_blk)
__global__ void reduce_slabs_strided_kernel(const float* __restrict__ part, float* __restrict__ out, int Co, int Cx,
                                            int ldo, int col_off, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, count = Co * Cx;
    if (i >= count) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {
        s0 += part[(size_t)k * count + i];
        s1 += part[(size_t)(k + 1) * count + i];
        s2 += part[(size_t)(k + 2) * count + i];
        s3 += part[(size_t)(k + 3) * count + i];
    }
    for (; k < nsplit; ++k) s0 += part[(size_t)k * count + i];
    out[(size_t)(i / Cx) * ldo + col_off + (i % Cx)] = (s0 + s1) + (s2 + s3);
}

// adjoint of the bilinear upsample (gather form, deterministic): lo[pl][ys][xs] = sum over the output
// pixels whose taps touch (ys,xs) of weight * hi[pl][oy][ox].  One workgroup per (plane, band of 8 source
// rows): the <= 8*ratio+6 output rows the band touches are staged in LDS with coalesced, independent loads,
// then each thread produces source pixels from LDS in a fixed summation order.
constexpr int ADJ_BAND = 8;
// taps one source column can receive from: ceil(2 / rw) + slack
static inline int adj_xtaps(int W, int Wl) { return (int)(2.0f * (float)W / (float)Wl) + 4; }

// The bilinear weights depend on (output row, source row) and (output column, source column) only, so they are tabulated
// once per workgroup; the two gather passes are then one LDS weight (broadcast) + one LDS value + one FMA per term.  (The
// first version recomputed the taps inside both inner loops and ran at 1.8 TB/s on a pass that only reads dz once.)
//
// FUSED form (DzArgs given): the staged rows are not read from a materialised dz but computed on the fly from dout and z
// (dz = gamma*invstd*(dy - mean_dy - xhat*mean_dyx), dy = 1[bn(z) > 0]*(dout*a1 + a2), exactly ffm_dz_kernel's expression);
// the band also WRITES the dz rows it owns (rows [ceil(ys_lo*H/Hl), ceil((ys_hi+1)*H/Hl)): a partition of the plane), so
// the separate dz pass and its 134 MB re-read by the adjoint disappear; the 6 halo rows per band are recomputed.
struct DzArgs {
    const float *g, *z, *mean, *invstd, *bn_w, *bn_b, *coef_a1, *coef_a2, *mean_dy, *mean_dyx;
    float* dz;
    int C;
};
// LDS layout of the horizontal pass (round 4): the vertical sums of a row are stored DE-INTERLEAVED by the resize ratio
// R = W / Wl (when it is an integer; else R = 1): output column ox sits at (ox % R) * plane + ox / R.  Source column xs
// receives from the taps ox = b_lo(xs) + t with b_lo advancing by R per lane, so in tap step t every lane of a wave reads
// the same phase at consecutive words; the tap weights and these permuted indices are tabulated tap-major ([XT][Wl]: a lane
// = a source column reads consecutive words).  Round 3 read tcol[b_lo + t] (lane stride R = 4 words) and wxw[xs * XT + t]
// (lane stride XT = 12 words): both 4-way bank conflicts, 576 extra LDS cycles per workgroup = the 4.7 M
// SQ_LDS_BANK_CONFLICT cycles per launch of profiles/r03_pmc_counters.json.  plane = 32 / R (mod 32) keeps the phase-major
// WRITES of the vertical pass conflict-free as well.
template <bool FUSED>
__global__ __launch_bounds__(256) void upsample_adjoint_kernel(const float* __restrict__ hi, float* __restrict__ lo,
                                                                int H, int W, int Hl, int Wl, float rh, float rw,
                                                                int max_rows, int XT, int R, int plane, DzArgs d) {
    extern __shared__ __attribute__((aligned(16))) float rows[];  // [max_rows][W] staged output rows
    const int TW = R * plane;
    float* tcol = rows + (size_t)max_rows * W;   // [ADJ_BAND][R][plane] vertical sums, de-interleaved by R
    float* wyt = tcol + ADJ_BAND * TW;           // [ADJ_BAND][max_rows] vertical weights
    float* wxw = wyt + ADJ_BAND * max_rows;      // [XT][Wl] horizontal weights of the taps ox = b_lo(xs) + t
    int* xidx = reinterpret_cast<int*>(wxw + Wl * XT);  // [XT][Wl] position of that tap's vertical sum inside a tcol row
    int* yr = xidx + Wl * XT;                    // [ADJ_BAND] r_lo | r_hi << 16 (staged-row range with non-zero weight)
    const int bands = (Hl + ADJ_BAND - 1) / ADJ_BAND, pl = blockIdx.x / bands;
    const int ys_lo = (blockIdx.x % bands) * ADJ_BAND, ys_hi = min(ys_lo + ADJ_BAND, Hl) - 1;
    const int oy_lo = max(0, (int)floorf(((float)ys_lo - 0.5f) / rh - 0.5f) - 1);
    const int oy_hi = min(min(H - 1, (int)ceilf(((float)ys_hi + 1.5f) / rh - 0.5f) + 1), oy_lo + max_rows - 1);
    const int nrows = oy_hi - oy_lo + 1, n_in = nrows * W, nys = ys_hi - ys_lo + 1, tid = threadIdx.x;
    // every index split below is a division by a run-time width (~40 emulated instructions each; they were half of this
    // kernel's 1035 VALU instructions per wave, on a pass that should be HBM-bound): exact reciprocal forms instead
    const float inv_W = 1.f / (float)W, inv_Wl = 1.f / (float)Wl, inv_mr = 1.f / (float)max_rows;
    if (FUSED) {
        const int c = pl % d.C;
        const float mu = d.mean[c], is = d.invstd[c], gw = d.bn_w[c], gb = d.bn_b[c];
        const float a1 = d.coef_a1[pl], a2 = d.coef_a2[pl], mdy = d.mean_dy[c], mdyx = d.mean_dyx[c], gi = gw * is;
        const size_t base = ((size_t)pl * H + oy_lo) * W;
        const float* zs = d.z + base;
        const float* gs = d.g + base;
        float* dzs = d.dz + base;
        // rows this band owns (written to dz): a partition of [0, H) over the bands
        const int own_lo = (ys_lo * H + Hl - 1) / Hl - oy_lo, own_hi = ((ys_hi + 1) * H + Hl - 1) / Hl - 1 - oy_lo;
        auto body = [&](float zv, float gv) {
            const float xh = (zv - mu) * is;
            const float y = fmaf(xh, gw, gb);
            const float dy = y > 0.f ? fmaf(gv, a1, a2) : 0.f;
            return gi * (dy - mdy - xh * mdyx);
        };
        if ((W & 3) == 0) {
            for (int i = tid * 4; i < n_in; i += 1024) {
                const f32x4 zv = *reinterpret_cast<const f32x4*>(zs + i);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gs + i);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = body(zv[e], gv[e]);
                *reinterpret_cast<f32x4*>(rows + i) = o;
                const int r = idiv_small(i, inv_W);
                if (r >= own_lo && r <= own_hi) *reinterpret_cast<f32x4*>(dzs + i) = o;
            }
        } else {
            for (int i = tid; i < n_in; i += 256) {
                const float o = body(zs[i], gs[i]);
                rows[i] = o;
                const int r = idiv_small(i, inv_W);
                if (r >= own_lo && r <= own_hi) dzs[i] = o;
            }
        }
    } else {
        const float* src = hi + ((size_t)pl * H + oy_lo) * W;
        if ((W & 3) == 0) {
            for (int i = tid * 4; i < n_in; i += 1024)
                *reinterpret_cast<f32x4*>(rows + i) = *reinterpret_cast<const f32x4*>(src + i);
        } else {
            for (int i = tid; i < n_in; i += 256) rows[i] = src[i];
        }
    }
    for (int i = tid; i < nys * max_rows; i += 256) {
        const int yi = idiv_small(i, inv_mr), r = i - yi * max_rows, ys = ys_lo + yi;
        float wgt = 0.f;
        if (r < nrows) {
            int y0, y1;
            float ly;
            bilinear_taps(oy_lo + r, rh, Hl, y0, y1, ly);
            wgt = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
        }
        wyt[i] = wgt;
    }
    const float inv_R = 1.f / (float)R;
    for (int i = tid; i < Wl * XT; i += 256) {
        const int t = idiv_small(i, inv_Wl), xs = i - t * Wl;
        const int b_lo = max(0, (int)floorf(((float)xs - 0.5f) / rw - 0.5f) - 1), ox = b_lo + t;
        float wgt = 0.f;
        if (ox < W) {
            int x0, x1;
            float lx;
            bilinear_taps(ox, rw, Wl, x0, x1, lx);
            wgt = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        }
        wxw[i] = wgt;
        const int oc = min(ox, W - 1), q = idiv_small(oc, inv_R);  // weights past the row end are zero
        xidx[i] = (oc - q * R) * plane + q;
    }
    if (tid < nys) {
        const int ys = ys_lo + tid;
        const int a_lo = max(oy_lo, (int)floorf(((float)ys - 0.5f) / rh - 0.5f) - 1);
        const int a_hi = min(oy_hi, (int)ceilf(((float)ys + 1.5f) / rh - 0.5f) + 1);
        yr[tid] = (a_lo - oy_lo) | ((a_hi - oy_lo) << 16);
    }
    __syncthreads();
    // separable: vertical taps first (weights depend on the row only), then horizontal
    for (int o = tid; o < nys * W; o += 256) {
        const int yi = idiv_small(o, inv_W), ox = o - yi * W, rr = yr[yi];
        const float* wv = wyt + yi * max_rows;
        float acc = 0.f;
        for (int r = rr & 0xffff; r <= (rr >> 16); ++r) acc += wv[r] * rows[r * W + ox];
        const int q = idiv_small(ox, inv_R);
        tcol[yi * TW + (ox - q * R) * plane + q] = acc;
    }
    __syncthreads();
    for (int o = tid; o < nys * Wl; o += 256) {
        const int yi = idiv_small(o, inv_Wl), xs = o - yi * Wl;
        const float* tr = tcol + yi * TW;
        float acc = 0.f;
        for (int t = 0; t < XT; ++t) acc += wxw[t * Wl + xs] * tr[xidx[t * Wl + xs]];
        lo[((size_t)pl * Hl + ys_lo + yi) * Wl + xs] = acc;
    }
}

// enough pixel-range splits that (output tiles x splits) covers the chip ~3x
// NEVER more than 128: the workspace holds 128 slabs (bwd_layout, conv1x1's part).  Round 3 returned total_chunks itself
// whenever it was below `want`, i.e. up to 383 slabs for a 128-column product (B*P/32 between 129 and 383 chunks, e.g.
// 3 x 32 x 96 pixels): the slabs ran over the end of `part` into the low-resolution dz behind it -- found in round 4 by the
// fused-vs-chain test (tests/test_gpu_ffm.py::test_ffm_bwd_fused_equals_chain); no BASELINE configuration is in that range.
static int dw_nsplit(int total_chunks, int tiles) {
    int n = ceil_div(768, tiles);
    if (n > total_chunks) n = total_chunks;
    return n < 128 ? n : 128;
}

// =====================================================================================
// small kernels
// =====================================================================================
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int x = blockIdx.x * 32 + threadIdx.x, y0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8)
        if (x < cols && y0 + j < rows) tile[j][threadIdx.x] = in[(size_t)(y0 + j) * cols + x];
    __syncthreads();
    const int ox = blockIdx.y * 32 + threadIdx.x, oy0 = blockIdx.x * 32;
    for (int j = threadIdx.y; j < 32; j += 8)
        if (ox < rows && oy0 + j < cols) out[(size_t)(oy0 + j) * rows + ox] = tile[threadIdx.x][j];
}


// per (b,c) row of z: sum and sum of squares over the P pixels -> stat_part[2][C][B] (double)
// (a streaming pass over z, which is still Infinity-Cache resident right after the GEMM wrote it; this
// replaced an in-epilogue cross-lane reduction that cost ~60 us per launch in wave shuffles).
// Accumulated in DOUBLE end to end: var = E[z^2] - mean^2 cancels by (mean/std)^2, so fp32 row sums lost 1e-7 * that
// ratio (1e-3 of the variance for a channel with mean = 100 std); with double sums the one-pass form is exact to fp32
// output precision for any ratio a float tensor can hold.  The pass stays HBM / Infinity-Cache bound (3 f64 ops per value).
__global__ __launch_bounds__(256) void bn_rowstats_kernel(const float* __restrict__ z, double* __restrict__ stat_part,
                                                           int B, int C, int P) {
    __shared__ double s_red[2][4];
    const int row = blockIdx.x, b = row / C, c = row - b * C;
    const float* zr = z + (size_t)row * P;
    double s1 = 0.0, s2 = 0.0;
    if ((P & 3) == 0) {
        for (int p = threadIdx.x * 4; p < P; p += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(zr + p);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)v[e];
                s1 += d, s2 = fma(d, d, s2);
            }
        }
    } else {
        for (int p = threadIdx.x; p < P; p += 256) {
            const double d = (double)zr[p];
            s1 += d, s2 = fma(d, d, s2);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = s1;
        s_red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        stat_part[(size_t)c * B + b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        stat_part[((size_t)C + c) * B + b] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    }
}

// BN statistics, one workgroup per channel: training -> reduce the per-tile partials (double),
// update the running buffers; eval -> running statistics.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stat_part, int ntiles, int C,
                                                           long long count, int training, float momentum, float eps,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd) {
    __shared__ double s_red[2][4];
    const int c = blockIdx.x;
    if (!training) {
        if (threadIdx.x == 0) {
            save_mean[c] = running_mean[c];
            save_invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
        }
        return;
    }
    double s1 = 0.0, s2 = 0.0;
    const double* p1 = stat_part + (size_t)c * ntiles;
    const double* p2 = stat_part + ((size_t)C + c) * ntiles;
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        s1 += p1[t];
        s2 += p2[t];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = s1;
        s_red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        s2 = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        const double mean = s1 / (double)count;
        double var = s2 / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        save_mean[c] = (float)mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

void bn_rowstats(const float* z, double* stat_part, int B, int C, int P, hipStream_t stream) {
    hipLaunchKernelGGL(bn_rowstats_kernel, dim3(B * C), dim3(256), 0, stream, z, stat_part, B, C, P);
}
void bn_finalize(const double* stat_part, int ntiles, int C, int nch, long long count, int training, float momentum, float eps,
                 float* running_mean, float* running_var, float* save_mean, float* save_invstd, hipStream_t stream) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(nch), dim3(256), 0, stream, stat_part, ntiles, C, count, training,
                       momentum, eps, running_mean, running_var, save_mean, save_invstd);
}

// pooled[b][c] = mean_p relu(bn(z[b][c][p]))        one workgroup per (b,c) row.
// The BatchNorm finalize is folded in: every row reduces its channel's B double partials itself (2 B scalar loads), the
// image-0 row also writes save_mean / save_invstd and updates the running buffers -- it was a 5 us launch of its own.
//
// Round 6 -- ReLU decisions of borderline units in double.  tools/diag_ffm_flips.py on the model's own
// tensors at configs 3 and 5 (profiles/r06_ffm_flips_config{3,5}.json): every gradient of the FFM equals the fp64 replay with the
// kernels' OWN ReLU mask to 2e-7, and ALL of the in-situ distance from the fp64 oracle (dfsp 7.2e-4 / 4.4e-4) is 4 of 33.5 M (3 of
// 16.8 M) units whose pre-activation lies within the fp32 product's rounding error (1.9e-7 abs on |z| ~ 0.8) of zero and lands on
// the other side -- the fp32 CPU reference flips 3 (1) others.  A unit that close to the boundary is decided by rounding noise in
// ANY fp32 implementation, and each one toggles a whole gradient column; so this pass, the first that knows the batch mean,
// re-decides exactly those units: where |pre| < 2^-17 ((|mean| invstd + 1) |gamma| + |beta|) (about 200 of 33.5 M elements at config 3) the wave
// recomputes z = W [fsp; U(low)] for that one unit in double from the operator's inputs (24 loads per lane, a wave sum), takes
// the decision from (z64 - mean64) invstd64 gamma + beta in double, and stores the fp32 z nearest to z64 for which the kernels'
// own expression fmaf((z - mean) invstd, gamma, beta) > 0 -- the one every FFM kernel, forward and backward, evaluates -- gives
// that decision (at most a few ulps from z64; the BatchNorm sums were taken over the uncorrected z: a 1e-14 relative change).
// At most FFM_EXACT_CAP units per row are re-decided (degenerate inputs -- a constant plane -- would flag every element).
struct BnFin {
    const double* stat_part;  // [2][C][B] sums and sums of squares per (channel, image); unused in eval mode
    int B, training;
    long long count;
    float momentum, eps;
    float *run_mean, *run_var, *save_mean, *save_invstd;
};
struct FfmExact {      // the operator's inputs, for the double-precision re-decision; fsp == nullptr: off
    const float* fsp;  // (B, Cs, H, W)
    const float* xc;   // plain form: fcp (B, Cc, H, W); fused-upsample form: low (B, Cc, Hl, Wl)
    const float* w;    // (Co, Cs + Cc) as stored
    int Cs, Cc, H, W, Hl, Wl;   // Hl == 0: plain form
};
constexpr int FFM_EXACT_CAP = 16;

// z[b][co][p] in double, by one whole wave (every lane returns the sum)
__device__ double ffm_exact_z(const FfmExact& e, int b, int co, int p, int lane) {
    const int P = e.H * e.W, Cin = e.Cs + e.Cc;
    const float* wr = e.w + (size_t)co * Cin;
    double acc = 0.0;
    for (int c = lane; c < e.Cs; c += 64) acc += (double)wr[c] * (double)e.fsp[((size_t)b * e.Cs + c) * P + p];
    if (e.Hl == 0) {
        for (int c = lane; c < e.Cc; c += 64) acc += (double)wr[e.Cs + c] * (double)e.xc[((size_t)b * e.Cc + c) * P + p];
    } else {
        // F.interpolate(mode="bilinear", align_corners=False) as ATen evaluates it for a double tensor (cabinet.py:228-230):
        // src = (dst + 0.5) in / out - 0.5 clamped at 0, i1 = min(i0 + 1, in - 1)
        const int oy = p / e.W, ox = p - oy * e.W, Pl = e.Hl * e.Wl;
        auto taps = [](int d, int in, int out, int& i0, int& i1, double& l) {
            double src = ((double)d + 0.5) * ((double)in / (double)out) - 0.5;
            if (src < 0.0) src = 0.0;
            i0 = min((int)src, in - 1), i1 = min(i0 + 1, in - 1), l = src - (double)i0;
        };
        int y0, y1, x0, x1;
        double ly, lx;
        taps(oy, e.Hl, e.H, y0, y1, ly);
        taps(ox, e.Wl, e.W, x0, x1, lx);
        for (int c = lane; c < e.Cc; c += 64) {
            const float* lp = e.xc + ((size_t)b * e.Cc + c) * Pl;
            const double v = (1.0 - ly) * ((1.0 - lx) * (double)lp[y0 * e.Wl + x0] + lx * (double)lp[y0 * e.Wl + x1]) +
                             ly * ((1.0 - lx) * (double)lp[y1 * e.Wl + x0] + lx * (double)lp[y1 * e.Wl + x1]);
            acc += (double)wr[e.Cs + c] * v;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    return acc;
}

// the fp32 z nearest to z64 whose fp32 pre-activation has the sign of the double one
__device__ float ffm_decided_z(double z64, double mean64, double inv64, float mu, float inv, float gw, float gb) {
    const bool want = (z64 - mean64) * inv64 * (double)gw + (double)gb > 0.0;
    float zc = (float)z64;
    const bool up = (gw > 0.f) == want;   // which way z has to move to change the fp32 decision towards `want`
    for (int it = 0; it < 16; ++it) {
        if ((fmaf((zc - mu) * inv, gw, gb) > 0.f) == want) break;
        zc = nextafterf(zc, up ? INFINITY : -INFINITY);
    }
    return zc;
}

__global__ __launch_bounds__(256) void ffm_pool_kernel(float* __restrict__ z, BnFin fin, FfmExact ex,
                                                        const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                        float* __restrict__ pooled, int C, int P) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, c = row % C, b = row / C, lane = threadIdx.x & 63;
    float mu, inv;
    double mean64, inv64;
    if (!fin.training) {
        mu = fin.run_mean[c], inv = 1.0f / sqrtf(fin.run_var[c] + fin.eps);
        mean64 = (double)mu, inv64 = 1.0 / sqrt((double)fin.run_var[c] + (double)fin.eps);
        if (b == 0 && threadIdx.x == 0) fin.save_mean[c] = mu, fin.save_invstd[c] = inv;
    } else {
        double s1 = 0.0, s2 = 0.0;
        if (fin.B <= 16) {   // one partial per image: every thread sums them itself
            for (int bb = 0; bb < fin.B; ++bb) {
                s1 += fin.stat_part[(size_t)c * fin.B + bb];
                s2 += fin.stat_part[((size_t)C + c) * fin.B + bb];
            }
        } else {   // one partial per workgroup of the fused forward: a fixed-order tree over the 256 threads
            __shared__ double s_d[2][256];
            double p1 = 0.0, p2 = 0.0;
            for (int bb = threadIdx.x; bb < fin.B; bb += 256) {
                p1 += fin.stat_part[(size_t)c * fin.B + bb];
                p2 += fin.stat_part[((size_t)C + c) * fin.B + bb];
            }
            s_d[0][threadIdx.x] = p1, s_d[1][threadIdx.x] = p2;
            __syncthreads();
            for (int o = 128; o >= 1; o >>= 1) {
                if ((int)threadIdx.x < o) {
                    s_d[0][threadIdx.x] += s_d[0][threadIdx.x + o];
                    s_d[1][threadIdx.x] += s_d[1][threadIdx.x + o];
                }
                __syncthreads();
            }
            s1 = s_d[0][0], s2 = s_d[1][0];
        }
        const double mean = s1 / (double)fin.count;
        double var = s2 / (double)fin.count - mean * mean;
        if (var < 0.0) var = 0.0;
        mean64 = mean, inv64 = 1.0 / sqrt(var + (double)fin.eps);
        mu = (float)mean, inv = (float)inv64;
        if (b == 0 && threadIdx.x == 0) {
            fin.save_mean[c] = mu, fin.save_invstd[c] = inv;
            const double unbiased = fin.count > 1 ? var * ((double)fin.count / (double)(fin.count - 1)) : var;
            fin.run_mean[c] = (float)((1.0 - (double)fin.momentum) * (double)fin.run_mean[c] + (double)fin.momentum * mean);
            fin.run_var[c] = (float)((1.0 - (double)fin.momentum) * (double)fin.run_var[c] + (double)fin.momentum * unbiased);
        }
    }
    // relu(bn(z)) as EVERY FFM kernel evaluates it (ffm_gate_kernel, the backward's reduction, adjoint and product kernels):
    // pre = fmaf((z - mean) invstd, gamma, beta) -- no cancellation against a folded shift, and one expression for one mask
    const float gw = bn_w[c], gb = bn_b[c];
    // the band in which a unit counts as borderline, in units of the pre-activation: the fp32 product's error is relative to |z|, i.e.
    // to |mean| + a few sigma, while the decision is taken in sigma units -- a channel whose mean is many sigma wide of zero needs a
    // band that much wider (round 6: Large 2x512^2 flipped a unit of such a channel at 1e-5 sigma with a band of 7.6e-6 sigma)
    const float thr = ex.fsp ? 0x1p-17f * ((fabsf(mu) * inv + 1.f) * fabsf(gw) + fabsf(gb)) : -1.f;
    float* zr = z + (size_t)row * P;
    float acc = 0.f;
    int budget = FFM_EXACT_CAP;   // per wave: wave-uniform
    // a borderline element of lane `src` (wave-uniform branch: every lane takes part in the double-precision product)
    auto redecide = [&](int src, int p) {
        const double z64 = ffm_exact_z(ex, b, c, p, lane);
        const float zc = ffm_decided_z(z64, mean64, inv64, mu, inv, gw, gb);
        if (lane == src) zr[p] = zc;
        return zc;
    };
    if ((P & 3) == 0) {
        // four iterations' 16-byte pieces are requested BEFORE the first is tested: the borderline test ends in a wave-uniform
        // branch on loaded data, and behind such a branch the next load is only issued once this one has returned -- one memory
        // latency per iteration (the first round-6 version: 23 -> 41 us in the step, profiles/r06_bench_n1_summary.md)
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        for (int p0 = 0; p0 < P; p0 += 4096) {   // wave-uniform trip count; four 16-byte requests in flight per thread
            f32x4 vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + 1024 * u + threadIdx.x * 4;
                vv[u] = p < P ? *reinterpret_cast<const f32x4*>(zr + p) : zero4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + 1024 * u + threadIdx.x * 4;
                float y[4];
                bool flag = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = fmaf((vv[u][e] - mu) * inv, gw, gb);
                    flag |= fabsf(y[e]) < thr;
                }
                unsigned long long m = __ballot(flag && p < P);
                while (m && budget > 0) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int ps = __shfl(p, src, 64);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ye = __shfl(y[e], src, 64);
                        if (fabsf(ye) < thr && budget > 0) {   // wave-uniform
                            const float zc = redecide(src, ps + e);
                            if (lane == src) y[e] = fmaf((zc - mu) * inv, gw, gb);
                            --budget;
                        }
                    }
                }
                if (p < P) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += fmaxf(y[e], 0.f);
                }
            }
        }
    } else {
        for (int p0 = 0; p0 < P; p0 += 256) {
            const int p = p0 + threadIdx.x;
            const bool in = p < P;
            float y = in ? fmaf((zr[p] - mu) * inv, gw, gb) : 1.f;
            unsigned long long m = __ballot(in && fabsf(y) < thr);
            while (m && budget > 0) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float zc = redecide(src, __shfl(p, src, 64));
                if (lane == src) y = fmaf((zc - mu) * inv, gw, gb);
                --budget;
            }
            if (in) acc += fmaxf(y, 0.f);
        }
    }
    acc = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) pooled[row] = acc / (float)P;
}

// gate[b][c] = sigmoid(W2 relu(W1 pooled[b]))                 one workgroup per image
// SE-MLP kernels: one workgroup of SE_T threads per image.  Every product walks its weight matrix along the contiguous
// dimension (a wave per row, lanes along the row; column sums as per-wave partial rows reduced through LDS), so each phase
// is one round of independent coalesced loads.  Strided per-thread walks made these kernels chains of dependent ~1 us L2
// round trips (10 us forward, 19 us backward on eight workgroups).
constexpr int SE_T = 1024, SE_W = SE_T / 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out[j] = sum_c w[j][c] * x[c]; LR lanes share a row (64 / LR rows per wave at a time), four row groups per pass so that
// all of a pass's loads are in flight together; f(j, sum) consumes the result on the row's first lane
template <int LR, typename F>
__device__ __forceinline__ void se_rows(const float* __restrict__ w, const float* x, int rows, int cols, int wave,
                                        int lane, F f) {
    constexpr int RPW = 64 / LR, STEP = SE_W * RPW;  // rows per wave at a time, rows per workgroup at a time
    const int g = lane / LR, q = lane % LR;
    for (int j = wave * RPW + g; j < rows; j += 4 * STEP) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* wr[4];  // row index clamped, not branched on: a load under its own branch waits for the previous one
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[e] = w + (size_t)min(j + e * STEP, rows - 1) * cols;
        for (int c = q; c < cols; c += LR) {
            const float xv = x[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += wr[e][c] * xv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e];
#pragma unroll
            for (int o = LR / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (q == 0 && j + e * STEP < rows) f(j + e * STEP, v);
        }
    }
}

__global__ __launch_bounds__(SE_T) void ffm_se_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, float* __restrict__ gate, int Co,
                                                       int Cm) {
    extern __shared__ float sm[];
    float* m = sm;        // [Co]
    float* r = sm + Co;   // [Cm]
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int c = tid; c < Co; c += SE_T) m[c] = pooled[(size_t)b * Co + c];
    __syncthreads();
    se_rows<64>(w1, m, Cm, Co, wave, lane, [&](int j, float v) { r[j] = fmaxf(v, 0.f); });
    __syncthreads();
    auto out = [&](int c, float v) { gate[(size_t)b * Co + c] = 1.f / (1.f + expf(-v)); };
    if (Cm <= 64)  // short rows: sixteen lanes per row, the whole second product is one round of loads
        se_rows<16>(w2, r, Co, Cm, wave, lane, out);
    else
        se_rows<64>(w2, r, Co, Cm, wave, lane, out);
}

// out = relu(bn(z)) * (1 + gate)                                 rows of P pixels
__global__ __launch_bounds__(256) void ffm_gate_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                        const float* __restrict__ gate, float* __restrict__ out, int C,
                                                        int P, int chunks_per_row) {
    const int row = blockIdx.x / chunks_per_row, chunk = blockIdx.x % chunks_per_row, c = row % C;
    const float mu = mean[c], is = invstd[c], gw = bn_w[c], gb = bn_b[c], ga = 1.f + gate[row];   // the mask expression of ffm_pool_kernel
    const float* zr = z + (size_t)row * P;
    float* orow = out + (size_t)row * P;
    const int lo = chunk * 4096, hi = min(lo + 4096, P);
    if ((P & 3) == 0) {
        for (int p = lo + threadIdx.x * 4; p < hi; p += 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(zr + p);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf((v[e] - mu) * is, gw, gb), 0.f) * ga;
            *reinterpret_cast<f32x4*>(orow + p) = v;
        }
    } else {
        for (int p = lo + threadIdx.x; p < hi; p += 256) orow[p] = fmaxf(fmaf((zr[p] - mu) * is, gw, gb), 0.f) * ga;
    }
}

// backward reduction: per (b,c) row five sums over pixels
//   S1 = sum g*feat   S2 = sum g*mask   S3 = sum g*mask*xhat   S4 = sum mask   S5 = sum mask*xhat
__global__ __launch_bounds__(256) void ffm_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ bn_w,
                                                              const float* __restrict__ bn_b, float* __restrict__ sums,
                                                              int C, int P) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, c = row % C;
    const float mu = mean[c], is = invstd[c], gw = bn_w[c], gb = bn_b[c];
    const float* zr = z + (size_t)row * P;
    const float* gr = g + (size_t)row * P;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f;
    auto body = [&](float zv, float gv) {
        const float xh = (zv - mu) * is;
        const float y = fmaf(xh, gw, gb);
        if (y > 0.f) {
            s1 += gv * y;
            s2 += gv;
            s3 += gv * xh;
            s4 += 1.f;
            s5 += xh;
        }
    };
    if ((P & 3) == 0) {
        for (int p = threadIdx.x * 4; p < P; p += 1024) {
            const f32x4 zv = *reinterpret_cast<const f32x4*>(zr + p);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gr + p);
#pragma unroll
            for (int e = 0; e < 4; ++e) body(zv[e], gv[e]);
        }
    } else {
        for (int p = threadIdx.x; p < P; p += 256) body(zr[p], gr[p]);
    }
    s1 = block_sum_256(s1, s_red);
    s2 = block_sum_256(s2, s_red);
    s3 = block_sum_256(s3, s_red);
    s4 = block_sum_256(s4, s_red);
    s5 = block_sum_256(s5, s_red);
    if (threadIdx.x == 0) {
        float* o = sums + (size_t)row * 5;
        o[0] = s1, o[1] = s2, o[2] = s3, o[3] = s4, o[4] = s5;
    }
}

// SE-MLP backward + BN coefficient algebra, stage 1: one workgroup per image.  Writes the dz
// coefficients a1 = 1 + gate, a2 = dm / P and this image's contributions to dw1, dw2, dbn_w, dbn_b.
__global__ __launch_bounds__(SE_T) void ffm_bwd_image_kernel(
    const float* __restrict__ sums, const float* __restrict__ pooled, const float* __restrict__ gate,
    const float* __restrict__ w1, const float* __restrict__ w2, int Co, int Cm, int P,
    float* __restrict__ dw1_part, float* __restrict__ dw2_part, float* __restrict__ dbn_part,
    float* __restrict__ coef_a1, float* __restrict__ coef_a2) {
    extern __shared__ float sm[];
    float* m = sm;            // [Co] pooled
    float* a = m + Co;        // [Co] gate
    float* ds = a + Co;       // [Co]
    float* u = ds + Co;       // [Cm]
    float* du = u + Cm;       // [Cm]
    float* red = du + Cm;     // [SE_W][max(Co, Cm)] per-wave partial rows of the two column sums
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ldr = Co > Cm ? Co : Cm;
    for (int c = tid; c < Co; c += SE_T) {
        const float av = gate[(size_t)b * Co + c];
        m[c] = pooled[(size_t)b * Co + c];
        a[c] = av;
        ds[c] = sums[((size_t)b * Co + c) * 5 + 0] * av * (1.f - av);
    }
    __syncthreads();
    se_rows<64>(w1, m, Cm, Co, wave, lane, [&](int j, float v) { u[j] = v; });  // u = W1 m
    for (int j0 = 0; j0 < Cm; j0 += 64) {  // dr = W2^T ds: this wave's rows c of W2, lanes along j
        const int j = j0 + lane;
        float p0 = 0.f, p1 = 0.f;
        if (j < Cm) {
            int c = wave;
#pragma unroll 8
            for (; c + SE_W < Co; c += 2 * SE_W) {
                p0 += w2[(size_t)c * Cm + j] * ds[c];
                p1 += w2[(size_t)(c + SE_W) * Cm + j] * ds[c + SE_W];
            }
            if (c < Co) p0 += w2[(size_t)c * Cm + j] * ds[c];
            red[wave * ldr + j] = p0 + p1;
        }
    }
    __syncthreads();
    for (int j = tid; j < Cm; j += SE_T) {
        float dr = 0.f;
#pragma unroll
        for (int w = 0; w < SE_W; ++w) dr += red[w * ldr + j];
        du[j] = u[j] > 0.f ? dr : 0.f;
    }
    __syncthreads();
    float* d1 = dw1_part + (size_t)b * Co * Cm;
    float* d2 = dw2_part + (size_t)b * Co * Cm;
    for (int c0 = 0; c0 < Co; c0 += 64) {  // dm = W1^T du: this wave's rows j of W1, lanes along c
        const int c = c0 + lane;
        float p = 0.f;
        if (c < Co) {
#pragma unroll 4
            for (int j = wave; j < Cm; j += SE_W) p += w1[(size_t)j * Co + c] * du[j];
            red[wave * ldr + c] = p;
        }
    }
    {   // dw2[c][j] = ds[c] relu(u[j]),  dw1[j][c] = du[j] m[c]: (row, col) advanced incrementally, no integer division
        int r2 = tid / Cm, c2 = tid - r2 * Cm, r1 = tid / Co, c1 = tid - r1 * Co;
        const int s2r = SE_T / Cm, s2c = SE_T - s2r * Cm, s1r = SE_T / Co, s1c = SE_T - s1r * Co;
        for (int i = tid; i < Co * Cm; i += SE_T) {
            d2[i] = ds[r2] * fmaxf(u[c2], 0.f);
            d1[i] = du[r1] * m[c1];
            r2 += s2r, c2 += s2c;
            if (c2 >= Cm) c2 -= Cm, ++r2;
            r1 += s1r, c1 += s1c;
            if (c1 >= Co) c1 -= Co, ++r1;
        }
    }
    float s14[4] = {0.f, 0.f, 0.f, 0.f};  // this thread's first channel: its sums are fetched before the barrier
    if (tid < Co) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s14[e] = sums[((size_t)b * Co + tid) * 5 + 1 + e];
    }
    __syncthreads();
    const float inv_p = 1.f / (float)P;
    for (int c = tid; c < Co; c += SE_T) {
        float dm = 0.f;
#pragma unroll
        for (int w = 0; w < SE_W; ++w) dm += red[w * ldr + c];
        const float a1 = 1.f + a[c], a2 = dm * inv_p;
        if (c != tid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s14[e] = sums[((size_t)b * Co + c) * 5 + 1 + e];
        }
        coef_a1[(size_t)b * Co + c] = a1;
        coef_a2[(size_t)b * Co + c] = a2;
        dbn_part[((size_t)b * 2 + 0) * Co + c] = a1 * s14[0] + a2 * s14[2];  // sum_p dy
        dbn_part[((size_t)b * 2 + 1) * Co + c] = a1 * s14[1] + a2 * s14[3];  // sum_p dy * xhat
    }
}

// stage 2: ordered sums over images (deterministic)
__global__ void ffm_bwd_combine_kernel(const float* __restrict__ dw1_part, const float* __restrict__ dw2_part,
                                       const float* __restrict__ dbn_part, int B, int Co, int Cm, int P, int training,
                                       float* __restrict__ dw1, float* __restrict__ dw2, float* __restrict__ dbn_w,
                                       float* __restrict__ dbn_b, float* __restrict__ mean_dy,
                                       float* __restrict__ mean_dyx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = Co * Cm;
    if (i < n) {
        float s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b) {
            s1 += dw1_part[(size_t)b * n + i];
            s2 += dw2_part[(size_t)b * n + i];
        }
        dw1[i] = s1;
        dw2[i] = s2;
    }
    if (i < Co) {
        float sb = 0.f, sw = 0.f;
        for (int b = 0; b < B; ++b) {
            sb += dbn_part[((size_t)b * 2 + 0) * Co + i];
            sw += dbn_part[((size_t)b * 2 + 1) * Co + i];
        }
        dbn_b[i] = sb;
        dbn_w[i] = sw;
        const float inv_count = 1.f / ((float)B * (float)P);
        mean_dy[i] = training ? sb * inv_count : 0.f;
        mean_dyx[i] = training ? sw * inv_count : 0.f;
    }
}

// dz = gamma*invstd * (dy - mean_dy - xhat*mean_dyx),  dy = mask * (g*a1 + a2)
__global__ __launch_bounds__(256) void ffm_dz_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                      const float* __restrict__ coef_a1,
                                                      const float* __restrict__ coef_a2,
                                                      const float* __restrict__ mean_dy,
                                                      const float* __restrict__ mean_dyx, float* __restrict__ dz, int C,
                                                      int P, int chunks_per_row) {
    const int row = blockIdx.x / chunks_per_row, chunk = blockIdx.x % chunks_per_row, c = row % C;
    const float mu = mean[c], is = invstd[c], gw = bn_w[c], gb = bn_b[c];
    const float a1 = coef_a1[row], a2 = coef_a2[row], mdy = mean_dy[c], mdyx = mean_dyx[c], gi = gw * is;
    const float* zr = z + (size_t)row * P;
    const float* gr = g + (size_t)row * P;
    float* dr = dz + (size_t)row * P;
    const int lo = chunk * 4096, hi = min(lo + 4096, P);
    auto body = [&](float zv, float gv) {
        const float xh = (zv - mu) * is;
        const float y = fmaf(xh, gw, gb);
        const float dy = y > 0.f ? fmaf(gv, a1, a2) : 0.f;
        return gi * (dy - mdy - xh * mdyx);
    };
    if ((P & 3) == 0) {
        for (int p = lo + threadIdx.x * 4; p < hi; p += 1024) {
            const f32x4 zv = *reinterpret_cast<const f32x4*>(zr + p);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gr + p);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = body(zv[e], gv[e]);
            *reinterpret_cast<f32x4*>(dr + p) = o;
        }
    } else {
        for (int p = lo + threadIdx.x; p < hi; p += 256) dr[p] = body(zr[p], gr[p]);
    }
}

// =====================================================================================
// host drivers
// =====================================================================================
// Two entry forms share every kernel:
//   plain : FFM(fsp, fcp)                        -- reference cabinet.py:142-153 as written
//   up    : FFM(fsp, bilinear_up(low -> H x W))  -- cabinet.py:228-230 + :236 fused.
// The "up" form never materialises fcp.  The 1x1 conv and the bilinear resize are both linear and act on
// different indices, so they commute:  W_c . U(low) == U(W_c . low).  The Cc-channel part of the GEMM
// therefore runs at LOW resolution (16x fewer pixels at the x4 upsample of CABiNet) and its result is added,
// bilinearly sampled, in the epilogue of the Cs-channel GEMM; in backward dlow = W_c^T . U^T(dz) and
// dW_c = U^T(dz) . low^T, again at low resolution.  GEMM FLOPs drop from 3 x 2*B*P*Cin*Co to
// 3 x 2*B*P*Cs*Co + 3 x 2*B*Pl*Cc*Co  (25.8 -> 9.7 GFLOP per product at config 3).
// ffm_fwd_fused.hip: z = W_s fsp + U(y_low) as a persistent kernel with the BatchNorm sums in its epilogue
bool ffm_fwd_fused_supported(int B, int Cs, int Co, int H, int W, int Hl, int Wl);
int ffm_fwd_fused_nwg(int B, int P);
hipError_t ffm_fwd_fused_run(const float* fsp, const float* w, int ldw, const float* ylow, int B, int H, int W, int Hl, int Wl,
                             float* z, double* stat_part, hipStream_t stream);
// CABINET_FFM_FWD_UNFUSED=1 keeps round 3's gemm_kmajor + bn_rowstats pair (A/B timing; the tests run both forms)
static bool ffm_fwd_fused_enabled() {
    const char* e = getenv("CABINET_FFM_FWD_UNFUSED");
    return !(e && e[0] == '1');
}

static size_t wt_bytes(const FfmShape& s) { return align_up((size_t)(s.Cs + s.Cc) * s.Co * sizeof(float), 256); }
// BatchNorm partial sums [2][Co][n]: one per image (bn_rowstats) or one per workgroup of the fused forward (<= 256)
static size_t stat_bytes(const FfmShape& s) { return align_up((size_t)(s.B > 256 ? s.B : 256) * 2 * s.Co * sizeof(double), 256); }
static size_t low_bytes(const FfmShape& s, int Hl, int Wl) {
    return align_up((size_t)s.B * s.Co * Hl * Wl * sizeof(float), 256);
}

size_t ffm_fwd_workspace(const FfmShape& s) { return wt_bytes(s) + stat_bytes(s); }
// gemm_bf16.hip
bool gemm_bf16_supported(int M, int K, int P, int W, int Wl, bool up);
size_t gemm_bf16_pack_bytes(int M, int K, int precision);
hipError_t gemm_bf16_run(int precision, const float* w, int ldw, const float* src, float* dst, int B, int M, int K, int P,
                         const float* up_src, int Hl, int W, float rh, float rw, void* pack, hipStream_t stream);

size_t ffm_up_fwd_workspace(const FfmShape& s, int Hl, int Wl) {
    return wt_bytes(s) + stat_bytes(s) + low_bytes(s, Hl, Wl) + gemm_bf16_pack_bytes(s.Co, s.Cs, 2);
}

// everything after z exists: BN statistics, pooling, SE gate, gated output
static bool ffm_exact_mask_enabled() {   // CABINET_FFM_EXACT_MASK=0: fp32 decisions everywhere (A/B: tools/diag_ffm_flips.py)
    static const bool on = [] { const char* e = getenv("CABINET_FFM_EXACT_MASK"); return !(e && e[0] == '0'); }();
    return on;
}
static void ffm_fwd_tail(const FfmShape& s, double* stat_part, const float* bn_w, const float* bn_b,
                         float* run_mean, float* run_var, const float* w1, const float* w2, int training,
                         float momentum, float eps, float* out, float* z, float* save_mean,
                         float* save_invstd, float* pooled, float* gate, hipStream_t stream, FfmExact ex, int nparts = 0) {
    // nparts > 0: the producer of z already left nparts (sum, sum of squares) pairs per channel in stat_part (fused forward)
    const int P = s.H * s.W;
    if (training && nparts == 0)
        hipLaunchKernelGGL(bn_rowstats_kernel, dim3(s.B * s.Co), dim3(256), 0, stream, z, stat_part, s.B, s.Co, P);
    const BnFin fin{stat_part, nparts > 0 ? nparts : s.B, training, (long long)s.B * P, momentum, eps, run_mean, run_var,
                    save_mean, save_invstd};
    if (!ffm_exact_mask_enabled()) ex.fsp = nullptr;
    hipLaunchKernelGGL(ffm_pool_kernel, dim3(s.B * s.Co), dim3(256), 0, stream, z, fin, ex, bn_w, bn_b, pooled, s.Co, P);
    hipLaunchKernelGGL(ffm_se_kernel, dim3(s.B), dim3(SE_T), (size_t)(s.Co + s.Cm) * sizeof(float), stream, pooled, w1,
                       w2, gate, s.Co, s.Cm);
    const int cpr = ceil_div(P, 4096);
    hipLaunchKernelGGL(ffm_gate_kernel, dim3(s.B * s.Co * cpr), dim3(256), 0, stream, z, save_mean, save_invstd, bn_w,
                       bn_b, gate, out, s.Co, P, cpr);
}

hipError_t ffm_fwd_run(const FfmShape& s, const float* fsp, const float* fcp, const float* w_blk,
                       const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                       const float* w1, const float* w2, int training, float momentum, float eps, float* out,
                       float* z, float* save_mean, float* save_invstd, float* pooled, float* gate, void* ws,
                       hipStream_t stream) {
    const int P = s.H * s.W, Cin = s.Cs + s.Cc;
    float* wt = static_cast<float*>(ws);
    double* stat_part = reinterpret_cast<double*>(static_cast<char*>(ws) + wt_bytes(s));
    // W_blk (Co x Cin) -> Wt (Cin x Co): the K-major A operand of G1
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(Cin, 32), ceil_div(s.Co, 32)), dim3(32, 8), 0, stream, w_blk, wt,
                       s.Co, Cin);
    GemmKArgs a{};
    a.at = wt, a.lda = s.Co, a.M = s.Co, a.K = Cin;
    a.src0 = fsp, a.src1 = fcp, a.K0 = s.Cs;
    a.dst0 = z, a.dst1 = z, a.M0 = s.Co;
    a.P = P;
    if (hipError_t ge = gemm_kmajor(a, s.B, stream); ge != hipSuccess) return ge;
    ffm_fwd_tail(s, stat_part, bn_w, bn_b, run_mean, run_var, w1, w2, training, momentum, eps, out, z, save_mean,
                 save_invstd, pooled, gate, stream, FfmExact{fsp, fcp, w_blk, s.Cs, s.Cc, s.H, s.W, 0, 0});
    return hipGetLastError();
}

hipError_t ffm_up_fwd_run(const FfmShape& s, int Hl, int Wl, const float* fsp, const float* low,
                          const float* w_blk, const float* bn_w, const float* bn_b, float* run_mean,
                          float* run_var, const float* w1, const float* w2, int training, float momentum,
                          float eps, float* out, float* z, float* save_mean, float* save_invstd, float* pooled,
                          float* gate, void* ws, int precision, hipStream_t stream) {
    const int P = s.H * s.W, Pl = Hl * Wl, Cin = s.Cs + s.Cc;
    char* base = static_cast<char*>(ws);
    float* wt = reinterpret_cast<float*>(base);
    double* stat_part = reinterpret_cast<double*>(base + wt_bytes(s));
    float* ylow = reinterpret_cast<float*>(base + wt_bytes(s) + stat_bytes(s));
    void* wpack = base + wt_bytes(s) + stat_bytes(s) + low_bytes(s, Hl, Wl);
    const FfmExact ex{fsp, low, w_blk, s.Cs, s.Cc, s.H, s.W, Hl, Wl};
    const bool small_low = small_grid(s.B, s.Co, Pl) && (Cin % 4) == 0 && (s.Cc % 4) == 0;
    // the persistent form of the big product (fp32; ffm_fwd_fused.hip) reads W as stored and leaves the BatchNorm partial sums
    const bool fused_z = precision == 0 && ffm_fwd_fused_enabled() && (Cin % 4) == 0 &&
                         ffm_fwd_fused_supported(s.B, s.Cs, s.Co, s.H, s.W, Hl, Wl);
    if (!(small_low && fused_z))   // W_blk (Co x Cin) -> Wt (Cin x Co): the K-major A operand of gemm_kmajor
        hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(Cin, 32), ceil_div(s.Co, 32)), dim3(32, 8), 0, stream, w_blk, wt,
                           s.Co, Cin);
    if (small_low) {
        // y_low = W_c . low at (Hl x Wl): a few thousand positions -> 64x64 tiles, the weight read as stored
        SgJobs jobs{};
        jobs.n = 1;
        jobs.j[0] = sg_make(w_blk + s.Cs, Cin, 1, low, s.Cc, s.Cc, s.Co, Pl, ylow, s.Co);
        sg_gemm(jobs, s.B, stream);
    } else {   // y_low = W_c . low   at (Hl x Wl)
        GemmKArgs a{};
        a.at = wt + (size_t)s.Cs * s.Co, a.lda = s.Co, a.M = s.Co, a.K = s.Cc;
        a.src0 = low, a.src1 = low, a.K0 = s.Cc;
        a.dst0 = ylow, a.dst1 = ylow, a.M0 = s.Co;
        a.P = Pl;
        if (hipError_t ge = gemm_kmajor(a, s.B, stream); ge != hipSuccess) return ge;
    }
    if (fused_z) {
        // z = W_s . fsp + U(y_low) and the per-channel sums of z in one persistent launch
        if (hipError_t e = ffm_fwd_fused_run(fsp, w_blk, Cin, ylow, s.B, s.H, s.W, Hl, Wl, z, training ? stat_part : nullptr, stream);
            e != hipSuccess)
            return e;
        ffm_fwd_tail(s, stat_part, bn_w, bn_b, run_mean, run_var, w1, w2, training, momentum, eps, out, z, save_mean,
                     save_invstd, pooled, gate, stream, ex, ffm_fwd_fused_nwg(s.B, P));
        return hipGetLastError();
    }
    if (precision != 0 && gemm_bf16_supported(s.Co, s.Cs, P, s.W, Wl, true)) {
        // z = W_s . fsp + U(y_low) on the bf16 matrix pipe (operands split into bf16 pieces, gemm_bf16.hip)
        if (hipError_t e = gemm_bf16_run(precision, w_blk, Cin, fsp, z, s.B, s.Co, s.Cs, P, ylow, Hl, s.W, (float)Hl / (float)s.H,
                                         (float)Wl / (float)s.W, wpack, stream);
            e != hipSuccess)
            return e;
    } else {   // z = W_s . fsp + U(y_low), exact fp32 MFMA
        GemmKArgs a{};
        a.at = wt, a.lda = s.Co, a.M = s.Co, a.K = s.Cs;
        a.src0 = fsp, a.src1 = fsp, a.K0 = s.Cs;
        a.dst0 = z, a.dst1 = z, a.M0 = s.Co;
        a.P = P;
        a.up_src = ylow, a.Hl = Hl, a.Wl = Wl, a.W = s.W;
        a.rh = (float)Hl / (float)s.H, a.rw = (float)Wl / (float)s.W;
        if (hipError_t ge = gemm_kmajor(a, s.B, stream); ge != hipSuccess) return ge;
    }
    ffm_fwd_tail(s, stat_part, bn_w, bn_b, run_mean, run_var, w1, w2, training, momentum, eps, out, z, save_mean,
                 save_invstd, pooled, gate, stream, ex);
    return hipGetLastError();
}

// ffm_bwd_fused.hip: dfsp, dlow and dW from one staged dz tile (one persistent launch + an ordered slab sum)
bool ffm_bwd_fused_supported(int B, int Cs, int Cc, int Co, int P, int Pl);
size_t ffm_bwd_fused_slab_floats(int B, int Cs, int Cc, int P, int Pl);
hipError_t ffm_bwd_fused_run(const float* dz, const float* dzl, const float* fsp, const float* low, const float* w, int B,
                             int Cs, int Cc, int P, int Pl, float* dfsp, float* dlow, float* dw, float* slabs,
                             hipStream_t stream, const XwDzCoef* dz_coef);
// CABINET_FFM_BWD_UNFUSED=1 keeps the round-3 chain of five launches (A/B timing; the tests run both forms)
// ffm_bwd_adj.hip: U^T of three coefficient-free fields inside the reduction pass, dz_low from them (no second pass over dout, z)
bool ffm_bwd_adj_supported(int H, int W, int Hl, int Wl);
hipError_t ffm_bwd_reduce_adj_run(const float* g, const float* z, const float* mean, const float* invstd, const float* bn_w,
                                  const float* bn_b, float* sums, float* t, int planes, int C, int H, int Hl, int Wl,
                                  hipStream_t stream);
hipError_t ffm_bwd_dzl_run(const float* t, const float* invstd, const float* bn_w, const float* a1, const float* a2,
                           const float* mdy, const float* mdyx, float* dzl, int planes, int C, int Hl, int Wl, hipStream_t stream);
// CABINET_FFM_BWD_TWO_PASS=1 keeps the second pass over dout and z (upsample_adjoint_kernel<true>; A/B timing, tests run both)
static bool ffm_bwd_adj_enabled() {
    const char* e = getenv("CABINET_FFM_BWD_TWO_PASS");
    return !(e && e[0] == '1');
}
static bool ffm_bwd_fused_enabled() {
    const char* e = getenv("CABINET_FFM_BWD_UNFUSED");
    return !(e && e[0] == '1');
}

struct BwdWs {
    size_t sums, a1, a2, mdy, mdyx, dw1p, dw2p, dbnp, dz, part, dzl, total;
};
static BwdWs bwd_layout(const FfmShape& s, int Hl, int Wl) {  // Hl == 0: plain form
    const int P = s.H * s.W;
    const int total_chunks = s.B * ceil_div(P, G3_BK);
    BwdWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    w.sums = take((size_t)s.B * s.Co * 5 * sizeof(float));
    w.a1 = take((size_t)s.B * s.Co * sizeof(float));
    w.a2 = take((size_t)s.B * s.Co * sizeof(float));
    w.mdy = take((size_t)s.Co * sizeof(float));
    w.mdyx = take((size_t)s.Co * sizeof(float));
    w.dw1p = take((size_t)s.B * s.Co * s.Cm * sizeof(float));
    w.dw2p = take((size_t)s.B * s.Co * s.Cm * sizeof(float));
    w.dbnp = take((size_t)s.B * 2 * s.Co * sizeof(float));
    w.dz = take((size_t)s.B * s.Co * P * sizeof(float));
    (void)total_chunks;
    size_t part_floats = (size_t)128 * s.Co * (s.Cs > s.Cc ? s.Cs : s.Cc);  // worst-case slabs of one dw_product
    if (Hl && ffm_bwd_fused_supported(s.B, s.Cs, s.Cc, s.Co, P, Hl * Wl)) {
        const size_t f = ffm_bwd_fused_slab_floats(s.B, s.Cs, s.Cc, P, Hl * Wl);  // one dW tile per workgroup and segment
        part_floats = f > part_floats ? f : part_floats;
    }
    w.part = take(part_floats * sizeof(float));
    w.dzl = take(Hl ? (size_t)s.B * s.Co * Hl * Wl * sizeof(float) : 0);
    w.total = off;
    return w;
}

size_t ffm_bwd_workspace(const FfmShape& s) { return bwd_layout(s, 0, 0).total; }
size_t ffm_up_bwd_workspace(const FfmShape& s, int Hl, int Wl) { return bwd_layout(s, Hl, Wl).total; }

// dout -> dz (and the small gradients dw1, dw2, dbn_w, dbn_b): shared by both forms
static hipError_t ffm_bwd_head(const FfmShape& s, const BwdWs& L, char* base, const float* dout, const float* z,
                         const float* save_mean, const float* save_invstd, const float* bn_w, const float* bn_b,
                         const float* w1, const float* w2, const float* pooled, const float* gate, int training,
                         float* dbn_w, float* dbn_b, float* dw1, float* dw2, hipStream_t stream, bool dz_pass = true,
                         int adj_Hl = 0, int adj_Wl = 0) {   // adj_Hl > 0: the adjoint fields ride on the reduction pass
    const int P = s.H * s.W;
    float* sums = reinterpret_cast<float*>(base + L.sums);
    float* a1 = reinterpret_cast<float*>(base + L.a1);
    float* a2 = reinterpret_cast<float*>(base + L.a2);
    float* mdy = reinterpret_cast<float*>(base + L.mdy);
    float* mdyx = reinterpret_cast<float*>(base + L.mdyx);
    float* dz = reinterpret_cast<float*>(base + L.dz);
    float* dw1p = reinterpret_cast<float*>(base + L.dw1p);
    float* dw2p = reinterpret_cast<float*>(base + L.dw2p);
    float* dbnp = reinterpret_cast<float*>(base + L.dbnp);
    // T1..T3 (3 x B Co Hl Wl floats) borrow the head of the dz buffer: read by ffm_bwd_dzl before anything writes dz
    if (adj_Hl > 0) {
        if (hipError_t e = ffm_bwd_reduce_adj_run(dout, z, save_mean, save_invstd, bn_w, bn_b, sums, dz, s.B * s.Co, s.Co, s.H,
                                                  adj_Hl, adj_Wl, stream);
            e != hipSuccess)
            return e;
    } else
        hipLaunchKernelGGL(ffm_bwd_reduce_kernel, dim3(s.B * s.Co), dim3(256), 0, stream, dout, z, save_mean, save_invstd,
                           bn_w, bn_b, sums, s.Co, P);
    hipLaunchKernelGGL(ffm_bwd_image_kernel, dim3(s.B), dim3(SE_T),
                       (size_t)(3 * s.Co + 2 * s.Cm + SE_W * (s.Co > s.Cm ? s.Co : s.Cm)) * sizeof(float), stream, sums,
                       pooled, gate, w1, w2, s.Co, s.Cm, P, dw1p, dw2p, dbnp, a1, a2);
    hipLaunchKernelGGL(ffm_bwd_combine_kernel, dim3(ceil_div(s.Co * s.Cm, 256)), dim3(256), 0, stream, dw1p, dw2p, dbnp,
                       s.B, s.Co, s.Cm, P, training, dw1, dw2, dbn_w, dbn_b, mdy, mdyx);
    if (adj_Hl > 0) {
        if (hipError_t e = ffm_bwd_dzl_run(dz, save_invstd, bn_w, a1, a2, mdy, mdyx, reinterpret_cast<float*>(base + L.dzl),
                                           s.B * s.Co, s.Co, adj_Hl, adj_Wl, stream);
            e != hipSuccess)
            return e;
    }
    if (!dz_pass) return hipGetLastError();  // the fused-upsample forms compute dz inside the upsample-adjoint kernel / while staging it
    const int cpr = ceil_div(P, 4096);
    hipLaunchKernelGGL(ffm_dz_kernel, dim3(s.B * s.Co * cpr), dim3(256), 0, stream, dout, z, save_mean, save_invstd,
                       bn_w, bn_b, a1, a2, mdy, mdyx, dz, s.Co, P, cpr);
    return hipGetLastError();
}

size_t dw_part_floats(int B, int Co, int Cx, int P) {
    const int total_chunks = B * ceil_div(P, G3_BK);
    return (size_t)dw_nsplit(total_chunks, ceil_div(Co, G3_T) * ceil_div(Cx, G3_T)) * Co * Cx;
}

// dW[:, col_off : col_off+Cx] = sum over images and pixels of dzv (B,Co,P) x xs (B,Cx,P)^T
hipError_t dw_product(const float* dzv, const float* xs, int B, int Co, int Cx, int P, float* part,
                             float* dw_blk, int ldo, int col_off, hipStream_t stream) {
    const int chunks_per_img = ceil_div(P, G3_BK), total_chunks = B * chunks_per_img;
    const int nsplit = dw_nsplit(total_chunks, ceil_div(Co, G3_T) * ceil_div(Cx, G3_T));
    const int cps = ceil_div(total_chunks, nsplit);
    const size_t lds = (size_t)(4 * G3_T * G3_STR) * sizeof(float);
    static lds_attr_mask mask_int{0}, mask_gen{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_dw_kernel<true>), lds, mask_int); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_dw_kernel<false>), lds, mask_gen); e != hipSuccess) return e;
    const bool interior = (P % G3_BK) == 0 && (Co % G3_T) == 0 && (Cx % G3_T) == 0;
    const dim3 grid(ceil_div(Co, G3_T) * ceil_div(Cx, G3_T) * nsplit);
    if (interior)
        hipLaunchKernelGGL(gemm_dw_kernel<true>, grid, dim3(256), lds, stream, dzv, xs, (const float*)nullptr, part, B,
                           Co, Cx, 0, P, chunks_per_img, cps);
    else
        hipLaunchKernelGGL(gemm_dw_kernel<false>, grid, dim3(256), lds, stream, dzv, xs, (const float*)nullptr, part, B,
                           Co, Cx, 0, P, chunks_per_img, cps);
    hipLaunchKernelGGL(reduce_slabs_strided_kernel, dim3(ceil_div(Co * Cx, 256)), dim3(256), 0, stream, part, dw_blk, Co,
                       Cx, ldo, col_off, nsplit);
    return hipGetLastError();
}

hipError_t ffm_bwd_run(const FfmShape& s, const float* dout, const float* fsp, const float* fcp,
                       const float* w_blk, const float* bn_w, const float* bn_b, const float* w1,
                       const float* w2, const float* z, const float* save_mean, const float* save_invstd,
                       const float* pooled, const float* gate, int training, float* dfsp, float* dfcp,
                       float* dw_blk, float* dbn_w, float* dbn_b, float* dw1, float* dw2, void* ws,
                       hipStream_t stream) {
    const int P = s.H * s.W, Cin = s.Cs + s.Cc;
    const BwdWs L = bwd_layout(s, 0, 0);
    char* base = static_cast<char*>(ws);
    float* dz = reinterpret_cast<float*>(base + L.dz);
    float* part = reinterpret_cast<float*>(base + L.part);
    if (hipError_t e = ffm_bwd_head(s, L, base, dout, z, save_mean, save_invstd, bn_w, bn_b, w1, w2, pooled, gate, training,
                                    dbn_w, dbn_b, dw1, dw2, stream);
        e != hipSuccess)
        return e;
    // G2: dX[c][p] = sum_o W[o][c] dz[o][p]   (A_t = W_blk as stored: [o][c], c contiguous)
    GemmKArgs a{};
    a.at = w_blk, a.lda = Cin, a.M = Cin, a.K = s.Co;
    a.src0 = dz, a.src1 = dz, a.K0 = s.Co;
    a.dst0 = dfsp, a.dst1 = dfcp, a.M0 = s.Cs;
    a.P = P;
    if (hipError_t ge = gemm_kmajor(a, s.B, stream); ge != hipSuccess) return ge;
    // G3: dW = dz X^T per source tensor, split over pixel chunks, ordered slab reduction
    hipError_t e = dw_product(dz, fsp, s.B, s.Co, s.Cs, P, part, dw_blk, Cin, 0, stream);
    if (e != hipSuccess) return e;
    return dw_product(dz, fcp, s.B, s.Co, s.Cc, P, part, dw_blk, Cin, s.Cs, stream);
}

hipError_t ffm_up_bwd_run(const FfmShape& s, int Hl, int Wl, const float* dout, const float* fsp,
                          const float* low, const float* w_blk, const float* bn_w, const float* bn_b,
                          const float* w1, const float* w2, const float* z, const float* save_mean,
                          const float* save_invstd, const float* pooled, const float* gate, int training,
                          float* dfsp, float* dlow, float* dw_blk, float* dbn_w, float* dbn_b, float* dw1,
                          float* dw2, void* ws, hipStream_t stream) {
    const int P = s.H * s.W, Pl = Hl * Wl, Cin = s.Cs + s.Cc;
    const BwdWs L = bwd_layout(s, Hl, Wl);
    char* base = static_cast<char*>(ws);
    float* dz = reinterpret_cast<float*>(base + L.dz);
    float* part = reinterpret_cast<float*>(base + L.part);
    float* dzl = reinterpret_cast<float*>(base + L.dzl);
    // the band decomposition must cover every output row exactly once and fit the staging buffer
    const int max_rows = (int)((ADJ_BAND + 2) * ((float)s.H / (float)Hl)) + 6, XT = adj_xtaps(s.W, Wl);
    const bool fuse_dz = s.H >= Hl;  // an upsample (the model's x4); shrinking resizes keep the two-pass form
    const bool lin_adj = fuse_dz && ffm_bwd_adj_enabled() && ffm_bwd_adj_supported(s.H, s.W, Hl, Wl);
    const bool fused_xw = fuse_dz && ffm_bwd_fused_enabled() && ffm_bwd_fused_supported(s.B, s.Cs, s.Cc, s.Co, P, Pl);
    const bool dz_in_xw = lin_adj && fused_xw;   // nothing but the fused kernel reads dz: it forms it from (dout, z) itself
    // lin_adj: dz_low from adjoint fields taken in the reduction pass (ffm_bwd_adj.hip)
    if (hipError_t e = ffm_bwd_head(s, L, base, dout, z, save_mean, save_invstd, bn_w, bn_b, w1, w2, pooled, gate, training,
                                    dbn_w, dbn_b, dw1, dw2, stream, lin_adj ? !dz_in_xw : !fuse_dz, lin_adj ? Hl : 0, lin_adj ? Wl : 0);
        e != hipSuccess)
        return e;
    // dz_low = U^T dz  (adjoint of the bilinear upsample), then everything on the Cc side is low resolution
    if (!lin_adj) {
        // de-interleave factor of the horizontal pass: the integer resize ratio (the model's x4), else 1 (plain layout)
        const int R = (Wl > 0 && s.W % Wl == 0 && s.W / Wl <= 32) ? s.W / Wl : 1;
        int plane = ceil_div(s.W, R);
        const int want = (R & (R - 1)) == 0 ? 32 / R : 1;  // phase stride in banks (see the kernel header)
        plane += (want - plane % 32 + 32) % 32;
        const size_t lds = ((size_t)max_rows * s.W + (size_t)ADJ_BAND * R * plane + (size_t)ADJ_BAND * max_rows +
                            (size_t)2 * Wl * XT + ADJ_BAND) * sizeof(float);
        // the staged band grows with the resize ratio: the attribute is set to the device maximum once (per device), so
        // that a small first call cannot pin it below what a later, larger geometry needs
        static lds_attr_mask mask_fused{0}, mask_plain{0};
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(upsample_adjoint_kernel<true>), 160 * 1024, mask_fused);
            e != hipSuccess)
            return e;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(upsample_adjoint_kernel<false>), 160 * 1024, mask_plain);
            e != hipSuccess)
            return e;
        DzArgs da{};
        const dim3 grid(s.B * s.Co * ((Hl + ADJ_BAND - 1) / ADJ_BAND));
        if (fuse_dz) {
            da.g = dout, da.z = z, da.mean = save_mean, da.invstd = save_invstd, da.bn_w = bn_w, da.bn_b = bn_b;
            da.coef_a1 = reinterpret_cast<float*>(base + L.a1), da.coef_a2 = reinterpret_cast<float*>(base + L.a2);
            da.mean_dy = reinterpret_cast<float*>(base + L.mdy), da.mean_dyx = reinterpret_cast<float*>(base + L.mdyx);
            da.dz = dz, da.C = s.Co;
            hipLaunchKernelGGL(upsample_adjoint_kernel<true>, grid, dim3(256), lds, stream, dz, dzl, s.H, s.W, Hl, Wl,
                               (float)Hl / (float)s.H, (float)Wl / (float)s.W, max_rows, XT, R, plane, da);
        } else {
            hipLaunchKernelGGL(upsample_adjoint_kernel<false>, grid, dim3(256), lds, stream, dz, dzl, s.H, s.W, Hl, Wl,
                               (float)Hl / (float)s.H, (float)Wl / (float)s.W, max_rows, XT, R, plane, da);
        }
    }
    if (fused_xw) {
        // dfsp = W_s^T dz, dlow = W_c^T dz_low, dW = [dz fsp^T | dz_low low^T]: both products of every dz tile from ONE
        // staging of it, the low-resolution side as further segments of the same persistent launch
        XwDzCoef dc{};
        dc.g = dout, dc.mean = save_mean, dc.invstd = save_invstd, dc.bn_w = bn_w, dc.bn_b = bn_b;
        dc.coef_a1 = reinterpret_cast<float*>(base + L.a1), dc.coef_a2 = reinterpret_cast<float*>(base + L.a2);
        dc.mean_dy = reinterpret_cast<float*>(base + L.mdy), dc.mean_dyx = reinterpret_cast<float*>(base + L.mdyx);
        return ffm_bwd_fused_run(dz_in_xw ? z : dz, dzl, fsp, low, w_blk, s.B, s.Cs, s.Cc, P, Pl, dfsp, dlow, dw_blk, part, stream,
                                 dz_in_xw ? &dc : nullptr);
    }
    {   // dfsp = W_s^T dz
        GemmKArgs a{};
        a.at = w_blk, a.lda = Cin, a.M = s.Cs, a.K = s.Co;
        a.src0 = dz, a.src1 = dz, a.K0 = s.Co;
        a.dst0 = dfsp, a.dst1 = dfsp, a.M0 = s.Cs;
        a.P = P;
        if (hipError_t ge = gemm_kmajor(a, s.B, stream); ge != hipSuccess) return ge;
    }
    const bool small = small_grid(s.B, s.Cc, Pl) && (Cin % 4) == 0 && (s.Cc % 4) == 0 && (s.Co % 4) == 0;
    if (small) {  // dlow = W_c^T dz_low: the (Co x Cin) weight is the K-major A operand as stored
        SgJobs jobs{};
        jobs.n = 1;
        jobs.j[0] = sg_make(w_blk + s.Cs, Cin, 0, dzl, s.Co, s.Co, s.Cc, Pl, dlow, s.Cc);
        sg_gemm(jobs, s.B, stream);
    } else {   // dlow = W_c^T dz_low
        GemmKArgs a{};
        a.at = w_blk + s.Cs, a.lda = Cin, a.M = s.Cc, a.K = s.Co;
        a.src0 = dzl, a.src1 = dzl, a.K0 = s.Co;
        a.dst0 = dlow, a.dst1 = dlow, a.M0 = s.Cc;
        a.P = Pl;
        if (hipError_t ge = gemm_kmajor(a, s.B, stream); ge != hipSuccess) return ge;
    }
    hipError_t e = dw_product(dz, fsp, s.B, s.Co, s.Cs, P, part, dw_blk, Cin, 0, stream);
    if (e != hipSuccess) return e;
    if (small) {  // dW_c = dz_low low^T over the low-resolution positions
        SdJobs jobs{};
        jobs.n = 1;
        jobs.j[0] = sd_make(dzl, s.Co, low, s.Cc, s.Co, s.Cc, Pl, dw_blk, Cin, s.Cs);
        return sd_run(jobs, s.B, part, stream);
    }
    return dw_product(dzl, low, s.B, s.Co, s.Cc, Pl, part, dw_blk, Cin, s.Cs, stream);
}

}  // namespace cabinet
