// Building blocks defined in ffm.hip and reused by the q/k/v producer (cab_qkv.hip) and the 1x1 convolution op.
#pragma once
#include <hip/hip_runtime.h>

namespace cabinet {

// D[m][p] = sum_k At[k][m] * Bm[k][p]   per image   ("K-major" GEMM == 1x1 convolution in NCHW)
struct GemmKArgs {
    const float* at;   // [K][lda], M (<= lda) contiguous entries per row (shared by all images)
    int lda;
    int M, K;
    const float* src0; // B-operand rows k <  K0 : (B, K0, P)
    const float* src1; // B-operand rows k >= K0 : (B, K-K0, P)
    int K0;
    float* dst0;       // output rows m <  M0 : (B, M0, P)
    float* dst1;       // output rows m >= M0 : (B, M-M0, P)
    int M0;
    int P;
    // optional epilogue term: D[m][p] += bilinear_upsample(up_src[b][m])(p), align_corners=False semantics
    // of F.interpolate (reference cabinet.py:228-230); up_src: (B, M, Hl, Wl), output pixels p = oy*W + ox
    const float* up_src;
    int Hl, Wl, W;
    float rh, rw;      // Hl / H, Wl / W
};

// tails in M and P are masked; if K % 16 != 0 the A operand must hold align16(K) rows, the extra ones zero
// (the B row index is clamped to K-1, so the tail products vanish)
void gemm_kmajor(const GemmKArgs& a, int B, hipStream_t stream);

// dW[:, col_off : col_off+Cx] = sum over images and pixels of dzv (B,Co,P) x xs (B,Cx,P)^T, written with row
// stride ldo; `part` holds the split-K slabs: dw_part_floats(...) floats
hipError_t dw_product(const float* dzv, const float* xs, int B, int Co, int Cx, int P, float* part, float* dw_blk,
                      int ldo, int col_off, hipStream_t stream);
size_t dw_part_floats(int B, int Co, int Cx, int P);

// per (b,c) row of z (B,C,P): sum and sum of squares -> stat_part[2][C][B] (double)
void bn_rowstats(const float* z, double* stat_part, int B, int C, int P, hipStream_t stream);
// one workgroup per channel, nch channels starting at the given pointers (C = channel count of the stat_part
// layout): training -> mean / invstd from the partials (+ running-stat update), eval -> running statistics
void bn_finalize(const double* stat_part, int ntiles, int C, int nch, long long count, int training, float momentum, float eps,
                 float* running_mean, float* running_var, float* save_mean, float* save_invstd, hipStream_t stream);

}  // namespace cabinet
