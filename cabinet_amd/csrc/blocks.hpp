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
    int wt_store;      // 1: output stored write-through (sc1) instead of non-temporal (set by gemm_kmajor from the environment)
};

// tails in M and P are masked; if K % 16 != 0 the A operand must hold align16(K) rows, the extra ones zero
// (the B row index is clamped to K-1, so the tail products vanish)
hipError_t gemm_kmajor(const GemmKArgs& a, int B, hipStream_t stream);

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

// ---- job-batched small GEMMs (small_gemm.hip) ----
struct SgSeg {
    const float* a;   // A rows of this K-segment: K-major (k, lda) or, with a_mmajor, the (M, lda) row-major matrix + k offset
    const float* b;   // B operand (B, b_rows, P); rows [0, k) of each image are contracted
    int k, b_rows;
};
struct SgJob {
    SgSeg seg[3];
    int nseg, lda, a_mmajor;
    int M, P;         // P positions are computed ...
    int ldp;          // ... out of rows that are ldp floats long in B and dst (0: ldp = P); lets a job work on a column window
    float* dst;       // (B, dst_rows, ldp); rows [0, M) of each image are written
    int dst_rows;
    int b_pmajor;     // B stored position-major: (B, P, ldb) with the contracted index contiguous (a transposed operand)
    int ldb;
    const float* a_bias;  // optional (B, M): subtracted from every A row (m) of image b while staging (M-major A only)
    float alpha;      // scale applied to the result (0 means 1)
    size_t a_img_stride;  // floats between the A operands of consecutive images (0: one A shared by all images, a weight)
    // optional epilogue (FAST launches only: sg_gemm's return value says whether it ran): per output row m, image b and 64-column
    // tile nt the pair (mean, M2 = sum (d - mean)^2) of the tile's 64 values, stat[((m * B + b) * tiles_n + nt) * 2 + {0, 1}] --
    // BatchNorm batch statistics of the product without a pass over it (merged by the consumer: equal counts, fixed order)
    float* stat;
    const float* out_bias;  // optional (M): added to every output row m (a 1x1 convolution's bias)
    int tiles_m, tiles_n, tile_base, nck[3], vec;  // filled by sg_gemm
};
struct SgJobs {
    SgJob j[12];
    int n;
    int B;   // images (filled by sg_gemm)
};
const char* sg_gemm_unsupported(const SgJob& j);  // nullptr when the small path can take the job
// mb: 64-row blocks per workgroup tile (1 | 2).  Returns true when the launch took the FAST kernel (every job: full tiles, whole
// chunks, aligned rows) -- only then are the jobs' `stat` partials written.
bool sg_gemm(SgJobs& jobs, int B, hipStream_t stream, int mb = 1);

struct SdJob {
    const float* a;   // (B, a_rows, P), rows [0, M)
    const float* x;   // (B, x_rows, P), rows [0, N)
    int a_rows, x_rows, M, N, P;
    float* out;       // out[m * ldo + col_off + n]
    int ldo, col_off;
    int tiles_m, tiles_n, cpi, cps, nsplit, tile_base, elem_base;  // filled by sd_plan
    size_t slab_off;
};
struct SdJobs {
    SdJob j[8];
    int n, total_tiles, total_elems;
};
// out[c] = sum over images and positions of d (B, C, P): the bias gradient of a 1x1 convolution; one workgroup per channel, fixed order
hipError_t channel_sum_run(const float* d, int B, int C, int P, float* out, hipStream_t stream);
size_t sd_plan(SdJobs& jobs, int B);  // slab floats needed for `part`
hipError_t sd_run(SdJobs& jobs, int B, float* part, hipStream_t stream);

inline SgJob sg_make(const float* a, int lda, int a_mmajor, const float* b, int k, int b_rows, int M, int P, float* dst,
                     int dst_rows) {
    SgJob j{};
    j.seg[0] = {a, b, k, b_rows};
    j.nseg = 1, j.lda = lda, j.a_mmajor = a_mmajor, j.M = M, j.P = P, j.dst = dst, j.dst_rows = dst_rows;
    return j;
}
inline SdJob sd_make(const float* a, int a_rows, const float* x, int x_rows, int M, int N, int P, float* out, int ldo,
                     int col_off) {
    SdJob j{};
    j.a = a, j.x = x, j.a_rows = a_rows, j.x_rows = x_rows, j.M = M, j.N = N, j.P = P, j.out = out, j.ldo = ldo,
    j.col_off = col_off;
    return j;
}
// true when a product over P positions per image is too small to fill the chip with the 128-wide tiles of gemm_kmajor
inline bool small_grid(int B, int M, int P) { return (long long)((P + 127) / 128) * B * ((M + 127) / 128) < 400; }

// What turns (dout, z) into dz inside ffm_bwd_fused.hip (ffm.hip::ffm_dz_kernel has the expression): the incoming gradient,
// the saved BatchNorm statistics and affine, the batch means of dy and dy xhat, and a1, a2 per (image, channel).
struct XwDzCoef {
    const float *g, *mean, *invstd, *bn_w, *bn_b, *mean_dy, *mean_dyx, *coef_a1, *coef_a2;
};

}  // namespace cabinet
