// K10 -- thin pointwise (1x1, bias-free) convolutions on large planes: streaming MFMA kernels, NCHW fp32.
//
// Replaces the nn.Conv2d(kernel_size=1) of the first MBConv blocks (reference src/models/mobilenetv3.py:128-131,
// 133-134,144-151: channel counts 16..120 on 128^2..512^2 planes).  There the product is HBM-bound (a few FLOP per
// byte) and MIOpen's NHWC implicit-GEMM path pays three layout transposes per call; the square-tile GEMM of ffm.hip
// is no better (its 128 x 128 tiles are mostly padding).  One wave per workgroup, no barriers:
//   stream : Y[b][m][p] = sum_k A[m][k] X[b][k][p]     forward (A = W) and input gradient (A = W^T)
//            A (<= 32 KB) is staged once in LDS; a wave walks 64-pixel blocks, loading its MFMA B operand straight
//            from global memory in the "pixel on the lane" layout (two 128-byte row segments per load) and storing
//            accumulator rows as 128-byte segments -- no transposition anywhere.
//   wgrad  : dW[co][ci] = sum_{b,p} dY[b][co][p] X[b][ci][p]
//            contraction over pixels: a wave stages 64 pixels of every dY and X row in its own LDS tile (coalesced
//            rows in, conflict-free columns out), keeps the whole Co x Ci tile in accumulators over its share of the
//            pixels, writes one slab; ordered slab sum (no atomics).
#include "common.hpp"

namespace cabinet {

constexpr int PW_MAXB = 4;  // row blocks of 32

// ------------------------------------------------------------------------------------------------ stream
// A[m][k] = a[m * sm + k * sk]  (forward: W (M=Co, K=Ci): sm = Ci, sk = 1;  input gradient: W^T: sm = 1, sk = Ci)
template <int MB>
__global__ __launch_bounds__(64) void pw_stream_kernel(const float* __restrict__ a, int sm, int sk, int M, int K,
                                                        const float* __restrict__ x, int P, int nblocks,
                                                        int blocks_per_img, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float al[];  // [32*MB][K + 1], zero rows past M
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5, ld = K + 1;
    // batches of 8 loads in flight (written as a plain loop the compiler waits for every load before its LDS store)
    for (int i0 = 0; i0 < 32 * MB * K; i0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 64 + lane, m = i / K, k = i - m * K;
            v[j] = (i < 32 * MB * K && m < M) ? a[(size_t)m * sm + (size_t)k * sk] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 64 + lane, m = i / K, k = i - m * K;
            if (i < 32 * MB * K) al[m * ld + k] = v[j];
        }
    }
    __syncthreads();
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int b = blk / blocks_per_img, p0 = (blk - b * blocks_per_img) * 64;
        const float* xb = x + (size_t)b * K * P;
        const int pa = min(p0 + li, P - 1), pb = min(p0 + 32 + li, P - 1);  // clamped: masked at the store
        f32x16 acc[MB][2];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 8) {  // K % 8 == 0
            float b0[4], b1[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* row = xb + (size_t)(k0 + 2 * s + h) * P;
                b0[s] = row[pa];
                b1[s] = row[pb];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const float av = al[(32 * i + li) * ld + k0 + 2 * s + h];
                    acc[i][0] = mfma32(av, b0[s], acc[i][0]);
                    acc[i][1] = mfma32(av, b1[s], acc[i][1]);
                }
            }
        }
        float* yb = y + (size_t)b * M * P;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * i + acc_row(r) + 4 * h;
                if (m < M) {
                    if (p0 + li < P) yb[(size_t)m * P + p0 + li] = acc[i][0][r];
                    if (p0 + 32 + li < P) yb[(size_t)m * P + p0 + 32 + li] = acc[i][1][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
template <int COB, int CIB>
__global__ __launch_bounds__(64) void pw_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int Co,
                                                       int Ci, int P, int nchunks, int chunks_per_img,
                                                       float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float tl[];  // [32*COB + 32*CIB][65]
    float* ty = tl;
    float* tx = tl + 32 * COB * 65;
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
    for (int i = lane; i < (32 * COB + 32 * CIB) * 65; i += 64) tl[i] = 0.f;  // rows past Co / Ci stay zero
    f32x16 acc[COB][CIB];
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int j = 0; j < CIB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int b = ch / chunks_per_img, p0 = (ch - b * chunks_per_img) * 64;
        const bool live = p0 + lane < P;
        const float* dyb = dy + (size_t)b * Co * P + p0 + lane;
        const float* xb = x + (size_t)b * Ci * P + p0 + lane;
        __syncthreads();  // the previous chunk's column reads are done
        // rows in batches of 8 (Co, Ci are multiples of 8), two batches in flight: up to 16 independent 256-byte
        // loads per lane before the first LDS store (the loop is latency-bound, not bandwidth-bound, per wave)
        auto stage = [&](const float* src, float* dst, int rows) {
            for (int r0 = 0; r0 < rows; r0 += 16) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = (live && r0 + j < rows) ? src[(size_t)(r0 + j) * P] : 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (r0 + j < rows) dst[(r0 + j) * 65 + lane] = v[j];
            }
        };
        stage(dyb, ty, Co);
        stage(xb, tx, Ci);
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
            float av[COB], bv[CIB];
#pragma unroll
            for (int i = 0; i < COB; ++i) av[i] = ty[(32 * i + li) * 65 + 2 * s + h];
#pragma unroll
            for (int j = 0; j < CIB; ++j) bv[j] = tx[(32 * j + li) * 65 + 2 * s + h];
#pragma unroll
            for (int i = 0; i < COB; ++i)
#pragma unroll
                for (int j = 0; j < CIB; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * Co * Ci;
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int j = 0; j < CIB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * i + acc_row(r) + 4 * h, ci = 32 * j + li;
                if (co < Co && ci < Ci) slab[(size_t)co * Ci + ci] = acc[i][j][r];
            }
}

// dw[i] = sum over the slabs, one wave per element: lane l adds slabs l, l+64, ... then an ordered wave reduction
// (a thread per element walking 2048 slabs serially cost more than the gradient kernel itself)
__global__ __launch_bounds__(256) void pw_slab_sum_kernel(const float* __restrict__ slabs, int nslab, int count,
                                                           float* __restrict__ dw) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= count) return;
    float s0 = 0.f, s1 = 0.f;
    int k = lane;
    for (; k + 64 < nslab; k += 128) {
        s0 += slabs[(size_t)k * count + i];
        s1 += slabs[(size_t)(k + 64) * count + i];
    }
    if (k < nslab) s0 += slabs[(size_t)k * count + i];
    const float t = wave_sum(s0 + s1);
    if (lane == 0) dw[i] = t;
}

// ------------------------------------------------------------------------------------------------ host side
static int blocks_of(int n) { return ceil_div(n, 32); }
constexpr int PW_GRID = 256 * 8;  // one wave per workgroup: 8 resident per CU

bool pwconv_supported(int Ci, int Co, int P) {
    (void)P;
    return Ci % 8 == 0 && Co % 8 == 0 && Ci <= 32 * PW_MAXB && Co <= 32 * PW_MAXB &&
           blocks_of(Ci) * blocks_of(Co) <= 8 && 32 * PW_MAXB * (Ci > Co ? Ci + 1 : Co + 1) * 4 <= 64 * 1024;
}

static int wgrad_slabs(int B, int P) {
    const int nchunks = B * ceil_div(P, 64);
    return nchunks < PW_GRID ? nchunks : PW_GRID;
}
size_t pwconv_bwd_workspace(int B, int Ci, int Co, int P) {
    return align_up((size_t)wgrad_slabs(B, P) * Co * Ci * sizeof(float), 256);
}

static hipError_t stream_launch(const float* a, int sm, int sk, int M, int K, const float* x, int B, int P, float* y,
                                hipStream_t stream) {
    const int bpi = ceil_div(P, 64), nblocks = B * bpi, grid = nblocks < PW_GRID ? nblocks : PW_GRID;
    const int mb = blocks_of(M);
    const size_t lds = (size_t)32 * mb * (K + 1) * sizeof(float);
#define PW_STREAM(MBV)                                                                                               \
    hipLaunchKernelGGL(pw_stream_kernel<MBV>, dim3(grid), dim3(64), lds, stream, a, sm, sk, M, K, x, P, nblocks, bpi, y)
    if (mb == 1) PW_STREAM(1);
    else if (mb == 2) PW_STREAM(2);
    else if (mb == 3) PW_STREAM(3);
    else PW_STREAM(4);
#undef PW_STREAM
    return hipGetLastError();
}

hipError_t pwconv_fwd_run(const float* x, const float* w, int B, int Ci, int Co, int P, float* y, hipStream_t stream) {
    return stream_launch(w, Ci, 1, Co, Ci, x, B, P, y, stream);
}

template <int COB>
static void wgrad_launch_ci(int cib, int grid, size_t lds, hipStream_t stream, const float* dy, const float* x, int Co,
                            int Ci, int P, int nchunks, int cpi, float* slabs) {
#define PW_WG(CIBV)                                                                                                    \
    hipLaunchKernelGGL((pw_wgrad_kernel<COB, CIBV>), dim3(grid), dim3(64), lds, stream, dy, x, Co, Ci, P, nchunks, cpi, \
                       slabs)
    if (cib == 1) PW_WG(1);
    else if (cib == 2) PW_WG(2);
    else if (cib == 3) { if constexpr (COB <= 2) PW_WG(3); }
    else { if constexpr (COB <= 2) PW_WG(4); }
#undef PW_WG
}

hipError_t pwconv_bwd_run(const float* dy, const float* x, const float* w, int B, int Ci, int Co, int P, float* dx,
                          float* dw, void* ws, hipStream_t stream) {
    if (dx) {
        hipError_t e = stream_launch(w, 1, Ci, Ci, Co, dy, B, P, dx, stream);  // dx = W^T dy
        if (e != hipSuccess) return e;
    }
    if (dw) {
        const int cpi = ceil_div(P, 64), nchunks = B * cpi, grid = wgrad_slabs(B, P);
        const int cob = blocks_of(Co), cib = blocks_of(Ci);
        const size_t lds = (size_t)(32 * cob + 32 * cib) * 65 * sizeof(float);
        float* slabs = static_cast<float*>(ws);
        if (cob == 1) wgrad_launch_ci<1>(cib, grid, lds, stream, dy, x, Co, Ci, P, nchunks, cpi, slabs);
        else if (cob == 2) wgrad_launch_ci<2>(cib, grid, lds, stream, dy, x, Co, Ci, P, nchunks, cpi, slabs);
        else if (cob == 3) wgrad_launch_ci<3>(cib, grid, lds, stream, dy, x, Co, Ci, P, nchunks, cpi, slabs);
        else wgrad_launch_ci<4>(cib, grid, lds, stream, dy, x, Co, Ci, P, nchunks, cpi, slabs);
        hipLaunchKernelGGL(pw_slab_sum_kernel, dim3(ceil_div(Co * Ci, 4)), dim3(256), 0, stream, slabs, grid, Co * Ci, dw);
    }
    return hipGetLastError();
}

}  // namespace cabinet
