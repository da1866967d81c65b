// K5, tiled form -- the CAB local branch + block output for shapes whose channel does NOT fit one CU's LDS
// (B*H*W > 8192: B = 16 at 1024^2, 8x3x2048x1024, the un-tiled 4096x2160 UAVid validation frame of reference
// src/scripts/train.py:444-456 -> n = 128*68 = 8704).  Same operator as cab_local.hip (reference src/models/cab.py:175-184
// and :213-216), same saved state (x, per-stage mean / invstd), same results; what changes is that one channel is now
// spread over B * nT workgroups, so every BatchNorm batch statistic is a TWO-PHASE reduction: the producing launch writes one
// (sum, sum of squares) pair of doubles per workgroup, the consuming launch sums the pairs of its channel in a fixed order
// (every workgroup of the channel computes the identical value; no atomics, no flags, bit-reproducible).
//
//   forward  (4 launches)   z0 = DW0(x) | z1 = DW1(relu(bn0(z0))) | z2 = DW2(relu(bn1(z1))) | out = gamma*glob + x*(1+sigmoid(relu(bn2(z2))))
//   backward (8 launches)   recompute z0, z1, z2 from x and the saved statistics (3) | head: dy2, direct dx, dglob, sums (1) |
//                           stage 2, 1, 0: dz_s from (dy_s, sums), dW_s partials, transposed stencil, dy_{s-1} + its sums (3) |
//                           ordered sum of the dW / dgamma partials (1)
// A workgroup owns TR full rows of one (image, channel) plane: the stencil input (tile + one halo row either side, zero pad
// columns) sits in LDS; global accesses are contiguous row segments.  HBM-bound: every launch is one read + one write pass.
#include "cab_local.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int LT_T = 256;       // threads per workgroup
constexpr int LT_TILE = 4096;   // target elements per tile (TR = LT_TILE / W rows)
constexpr int LT_NPART = 11;    // doubles per workgroup of a backward stage: sum dy, sum dy*xhat, 9 weight-gradient taps

struct TiledGeom {
    int TR, nT, nblk;
    size_t tile_floats;  // one staged array: (TR + 2) * (W + 2)
};
static TiledGeom tiled_geom(int B, int H, int W) {
    TiledGeom g;
    g.TR = LT_TILE / W;
    if (g.TR < 1) g.TR = 1;
    if (g.TR > H) g.TR = H;
    g.nT = ceil_div(H, g.TR);
    g.nblk = B * g.nT;
    g.tile_floats = (size_t)(g.TR + 2) * (W + 2);
    return g;
}

bool local_tiled_supported(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || H >= 65536 || W >= 65536) return false;
    const TiledGeom g = tiled_geom(B, H, W);
    return 2 * g.tile_floats * sizeof(float) <= 60 * 1024 && (long long)g.nblk <= 65535LL * 16 && (long long)H * W < (1 << 21);
}

// where a kernel takes (mean, invstd) of one stage from
struct StatSrc {
    int mode;              // 0: per-workgroup partial sums of this forward (training), 1: saved arrays (backward), 2: running (eval)
    const double* part;    // mode 0: [C][nblk][2]
    const float* mean;     // mode 1: (C)
    const float* invstd;   // mode 1
    float* run_mean;       // mode 0: updated by the channel's first workgroup; mode 2: read
    float* run_var;
    float* save_mean;      // modes 0, 2: written by the channel's first workgroup (nullable)
    float* save_invstd;
    float momentum, eps;
};

// (mean, invstd) of channel c; with `first` (one workgroup per channel) also the side effects nn.BatchNorm2d has
__device__ __forceinline__ void stage_stats(const StatSrc& s, int c, int nblk, double N, bool first, float& mean, float& invstd) {
    if (s.mode == 1) {
        mean = s.mean[c];
        invstd = s.invstd[c];
        return;
    }
    if (s.mode == 2) {
        mean = s.run_mean[c];
        invstd = 1.0f / sqrtf(s.run_var[c] + s.eps);
    } else {
        double S1 = 0.0, S2 = 0.0;
        const double* p = s.part + (size_t)c * nblk * 2;
        for (int i = 0; i < nblk; ++i) S1 += p[2 * i], S2 += p[2 * i + 1];  // same order in every workgroup of the channel
        const double mu = S1 / N;
        double var = S2 / N - mu * mu;
        if (var < 0.0) var = 0.0;
        mean = (float)mu;
        invstd = (float)(1.0 / sqrt(var + (double)s.eps));
        if (first && threadIdx.x == 0) {
            const double unbiased = N > 1.0 ? var * (N / (N - 1.0)) : var;
            s.run_mean[c] = (float)((1.0 - (double)s.momentum) * (double)s.run_mean[c] + (double)s.momentum * mu);
            s.run_var[c] = (float)((1.0 - (double)s.momentum) * (double)s.run_var[c] + (double)s.momentum * unbiased);
        }
    }
    if (first && threadIdx.x == 0 && s.save_mean) {
        s.save_mean[c] = mean;
        s.save_invstd[c] = invstd;
    }
}

// ordered sum of NV doubles per thread over the 256-thread workgroup; thread j < NV returns value j (others: garbage)
template <int NV>
__device__ __forceinline__ double block_sum_d(double (&v)[NV], double* red /* [NV][4] */) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
    }
    __syncthreads();  // red may still be read from a previous use
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) red[j * 4 + (threadIdx.x >> 6)] = v[j];
    }
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x < NV) t = (red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1]) + (red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3]);
    return t;
}

__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + fast_exp2(-v * LOG2E_F)); }

struct TileId {
    int c, blk, b, r0, rows;
};
__device__ __forceinline__ TileId tile_id(int H, int TR, int nT) {
    TileId t;
    t.c = blockIdx.y, t.blk = blockIdx.x;
    t.b = t.blk / nT;
    t.r0 = (t.blk - t.b * nT) * TR;
    t.rows = min(TR, H - t.r0);
    return t;
}

// ------------------------------------------------------------------------------------------ forward stage
struct FwdStageArgs {
    const float* in;     // x (first stage) or z_{s-1}
    const float* w;      // (C,9) depthwise weights of this stage
    float* z;            // this stage's convolution output
    double* part;        // (C, nblk, 2) sums of z, or nullptr
    int first;           // the input is x: no activation
    StatSrc st;          // statistics of stage s-1 (the input's BatchNorm)
    const float* bn_w;   // affine of stage s-1
    const float* bn_b;
    int B, C, H, W, TR, nT;
};

__global__ __launch_bounds__(LT_T) void local_tiled_fwd_stage_kernel(FwdStageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double red[2 * 4];
    const TileId t = tile_id(a.H, a.TR, a.nT);
    const int tid = threadIdx.x, WP = a.W + 2, n = a.H * a.W, nblk = a.B * a.nT;
    float sc = 1.f, sh = 0.f;
    if (!a.first) {
        float mean, invstd;
        stage_stats(a.st, t.c, nblk, (double)a.B * n, t.blk == 0, mean, invstd);
        sc = a.bn_w[t.c] * invstd;
        sh = a.bn_b[t.c] - mean * sc;
    }
    const size_t plane = ((size_t)t.b * a.C + t.c) * n;
    const float inv_wp = 1.f / (float)WP, inv_w = 1.f / (float)a.W;
    for (int i = tid; i < (t.rows + 2) * WP; i += LT_T) {
        const int ry = idiv_small(i, inv_wp), x = i - ry * WP - 1, y = t.r0 - 1 + ry;
        float v = 0.f;
        if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
            v = a.in[plane + (size_t)y * a.W + x];
            if (!a.first) v = fmaxf(fmaf(v, sc, sh), 0.f);
        }
        smem[i] = v;
    }
    float w[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) w[j] = a.w[t.c * 9 + j];
    __syncthreads();
    double s[2] = {0.0, 0.0};
    for (int e = tid; e < t.rows * a.W; e += LT_T) {
        const int ty = idiv_small(e, inv_w), pe = (ty + 1) * WP + (e - ty * a.W) + 1;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc += w[ky * 3 + kx] * smem[pe + (ky - 1) * WP + (kx - 1)];
        a.z[plane + (size_t)t.r0 * a.W + e] = acc;
        s[0] += (double)acc;
        s[1] = fma((double)acc, (double)acc, s[1]);
    }
    if (a.part) {
        const double r = block_sum_d<2>(s, red);
        if (tid < 2) a.part[((size_t)t.c * nblk + t.blk) * 2 + tid] = r;
    }
}

// ------------------------------------------------------------------------------------------ forward output
struct FwdOutArgs {
    const float* x;
    const float* z2;
    const float* glob;   // nullable
    const float* gamma;
    float* out;
    StatSrc st;          // statistics of stage 2
    const float* bn_w;
    const float* bn_b;
    int B, C, H, W, TR, nT;
};

__global__ __launch_bounds__(LT_T) void local_tiled_fwd_out_kernel(FwdOutArgs a) {
    const TileId t = tile_id(a.H, a.TR, a.nT);
    const int n = a.H * a.W, nblk = a.B * a.nT;
    float mean, invstd;
    stage_stats(a.st, t.c, nblk, (double)a.B * n, t.blk == 0, mean, invstd);
    const float sc = a.bn_w[t.c] * invstd, sh = a.bn_b[t.c] - mean * sc;
    const float gam = a.glob ? a.gamma[0] : 0.f;
    const size_t base = ((size_t)t.b * a.C + t.c) * n + (size_t)t.r0 * a.W;
    for (int e = threadIdx.x; e < t.rows * a.W; e += LT_T) {
        const float y = fmaxf(fmaf(a.z2[base + e], sc, sh), 0.f);
        float o = a.x[base + e] * (1.f + sigmoid_fast(y));
        if (a.glob) o = fmaf(gam, a.glob[base + e], o);
        a.out[base + e] = o;
    }
}

// ------------------------------------------------------------------------------------------ backward head
struct BwdHeadArgs {
    const float* g;      // dout
    const float* x;
    const float* z2;
    const float* glob;   // nullable
    const float* gamma;
    const float* mean;   // saved statistics / affine of stage 2 (C)
    const float* invstd;
    const float* bn_w;
    const float* bn_b;
    float* dy;           // dL/dy2 already masked by the ReLU of stage 2
    float* dx;           // direct term g (1 + sigmoid(y2)); the stencil term is added by the last stage kernel
    float* dglob;        // nullable
    double* part;        // (C, nblk, 3): sum dy, sum dy*xhat2, <g, glob>
    int B, C, H, W, TR, nT;
};

__global__ __launch_bounds__(LT_T) void local_tiled_bwd_head_kernel(BwdHeadArgs a) {
    __shared__ double red[3 * 4];
    const TileId t = tile_id(a.H, a.TR, a.nT);
    const int n = a.H * a.W, nblk = a.B * a.nT;
    const float mean = a.mean[t.c], invstd = a.invstd[t.c], bw = a.bn_w[t.c], bb = a.bn_b[t.c];
    const float gam = a.glob ? a.gamma[0] : 0.f;
    const size_t base = ((size_t)t.b * a.C + t.c) * n + (size_t)t.r0 * a.W;
    double s[3] = {0.0, 0.0, 0.0};
    for (int e = threadIdx.x; e < t.rows * a.W; e += LT_T) {
        const float g = a.g[base + e], xh = (a.z2[base + e] - mean) * invstd, y = fmaxf(fmaf(xh, bw, bb), 0.f);
        const float sg = sigmoid_fast(y);
        const float dy = (y > 0.f) ? g * a.x[base + e] * sg * (1.f - sg) : 0.f;
        a.dy[base + e] = dy;
        a.dx[base + e] = g * (1.f + sg);
        if (a.glob) {
            s[2] = fma((double)g, (double)a.glob[base + e], s[2]);
            a.dglob[base + e] = gam * g;
        }
        s[0] += (double)dy;
        s[1] = fma((double)dy, (double)xh, s[1]);
    }
    const double r = block_sum_d<3>(s, red);
    if (threadIdx.x < 3) a.part[((size_t)t.c * nblk + t.blk) * 3 + threadIdx.x] = r;
}

// ------------------------------------------------------------------------------------------ backward stage
struct BwdStageArgs {
    const float* dy;       // dL/dy_s, masked by the ReLU of stage s
    const float* z;        // z_s
    const float* mean;     // saved statistics / affine scale of stage s (C)
    const float* invstd;
    const float* bn_w;
    const double* part_in; // per-workgroup sums (sum dy_s, sum dy_s * xhat_s) at [C][nblk][stride_in]
    int stride_in;
    int training;
    int first;             // s == 0: the stencil input is x, the input gradient is added to dx
    const float* in_prev;  // x (s == 0) or z_{s-1}
    const float* pmean;    // statistics / affine of stage s-1 (s > 0)
    const float* pinvstd;
    const float* pbn_w;
    const float* pbn_b;
    const float* w;        // (C,9) of stage s
    float* dy_prev;        // s > 0: dL/dy_{s-1} masked
    float* dx;             // s == 0
    double* part_out;      // (C, nblk, LT_NPART)
    float* dbn_w;          // (C) of stage s, written by the channel's first workgroup
    float* dbn_b;
    int B, C, H, W, TR, nT;
};

__global__ __launch_bounds__(LT_T) void local_tiled_bwd_stage_kernel(BwdStageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double red[LT_NPART * 4];
    const TileId t = tile_id(a.H, a.TR, a.nT);
    const int tid = threadIdx.x, WP = a.W + 2, n = a.H * a.W, nblk = a.B * a.nT;
    float* A = smem;                          // dz_s, tile + halo
    float* Bv = smem + (a.TR + 2) * WP;       // stencil input of stage s (x or y_{s-1}), tile + halo
    double S1 = 0.0, S2 = 0.0;
    {
        const double* p = a.part_in + (size_t)t.c * nblk * a.stride_in;
        for (int i = 0; i < nblk; ++i) S1 += p[(size_t)i * a.stride_in], S2 += p[(size_t)i * a.stride_in + 1];
    }
    if (t.blk == 0 && tid == 0) {
        a.dbn_b[t.c] = (float)S1;
        a.dbn_w[t.c] = (float)S2;
    }
    const double inv_n = 1.0 / ((double)a.B * n);
    const float m1 = a.training ? (float)(S1 * inv_n) : 0.f, m2 = a.training ? (float)(S2 * inv_n) : 0.f;
    const float mean = a.mean[t.c], invstd = a.invstd[t.c], gi = a.bn_w[t.c] * invstd;
    float pmean = 0.f, pinvstd = 1.f, pbw = 1.f, pbb = 0.f;
    if (!a.first) pmean = a.pmean[t.c], pinvstd = a.pinvstd[t.c], pbw = a.pbn_w[t.c], pbb = a.pbn_b[t.c];
    const size_t plane = ((size_t)t.b * a.C + t.c) * n;
    const float inv_wp = 1.f / (float)WP, inv_w = 1.f / (float)a.W;
    for (int i = tid; i < (t.rows + 2) * WP; i += LT_T) {
        const int ry = idiv_small(i, inv_wp), x = i - ry * WP - 1, y = t.r0 - 1 + ry;
        float dz = 0.f, inp = 0.f;
        if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
            const size_t gi_ = plane + (size_t)y * a.W + x;
            const float xh = (a.z[gi_] - mean) * invstd;
            dz = gi * (a.dy[gi_] - m1 - xh * m2);
            inp = a.in_prev[gi_];
            if (!a.first) inp = fmaxf(fmaf((inp - pmean) * pinvstd, pbw, pbb), 0.f);
        }
        A[i] = dz;
        Bv[i] = inp;
    }
    float w[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) w[j] = a.w[t.c * 9 + j];
    __syncthreads();
    float pw[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) pw[j] = 0.f;
    double s1 = 0.0, s2 = 0.0;
    for (int e = tid; e < t.rows * a.W; e += LT_T) {
        const int ty = idiv_small(e, inv_w), pe = (ty + 1) * WP + (e - ty * a.W) + 1;
        const float dz = A[pe];
        float din = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                pw[ky * 3 + kx] += dz * Bv[pe + (ky - 1) * WP + (kx - 1)];      // dW[ky][kx] = sum dz[e] in[e + (ky-1, kx-1)]
                din += w[ky * 3 + kx] * A[pe - (ky - 1) * WP - (kx - 1)];       // transposed stencil
            }
        const size_t gi_ = plane + (size_t)t.r0 * a.W + e;
        if (a.first) {
            a.dx[gi_] += din;
        } else {
            const float dyp = Bv[pe] > 0.f ? din : 0.f;  // through the ReLU of stage s-1
            a.dy_prev[gi_] = dyp;
            const float xhp = (a.in_prev[gi_] - pmean) * pinvstd;
            s1 += (double)dyp;
            s2 = fma((double)dyp, (double)xhp, s2);
        }
    }
    double v[LT_NPART];
    v[0] = s1, v[1] = s2;
#pragma unroll
    for (int j = 0; j < 9; ++j) v[2 + j] = (double)pw[j];
    const double r = block_sum_d<LT_NPART>(v, red);
    if (tid < LT_NPART) a.part_out[((size_t)t.c * nblk + t.blk) * LT_NPART + tid] = r;
}

// ordered sums of the per-workgroup partials: dw[s] (C,9) for the three stages and dgamma_part (C)
__global__ __launch_bounds__(256) void local_tiled_finalize_kernel(const double* part_stage0, const double* part_stage1,
                                                                   const double* part_stage2, const double* part_head,
                                                                   float* dw0, float* dw1, float* dw2, float* dgamma_part,
                                                                   int C, int nblk) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 3 * C * 9) {
        const int s = idx / (C * 9), r = idx - s * C * 9, c = r / 9, j = r - c * 9;
        const double* p = (s == 0 ? part_stage0 : s == 1 ? part_stage1 : part_stage2) + (size_t)c * nblk * LT_NPART + 2 + j;
        double t = 0.0;
        for (int i = 0; i < nblk; ++i) t += p[(size_t)i * LT_NPART];
        (s == 0 ? dw0 : s == 1 ? dw1 : dw2)[r] = (float)t;
    } else if (idx < 3 * C * 9 + C && dgamma_part) {
        const int c = idx - 3 * C * 9;
        const double* p = part_head + (size_t)c * nblk * 3 + 2;
        double t = 0.0;
        for (int i = 0; i < nblk; ++i) t += p[(size_t)i * 3];
        dgamma_part[c] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------ host side
static size_t tensor_floats(const LocalArgs& a) { return align_up((size_t)a.B * a.C * a.H * a.W, 64); }

size_t local_tiled_fwd_workspace(int B, int C, int H, int W) {
    const TiledGeom g = tiled_geom(B, H, W);
    return align_up(3 * align_up((size_t)B * C * H * W, 64) * sizeof(float) + (size_t)3 * C * g.nblk * 2 * sizeof(double), 256);
}
size_t local_tiled_bwd_workspace(int B, int C, int H, int W) {
    const TiledGeom g = tiled_geom(B, H, W);
    return align_up(5 * align_up((size_t)B * C * H * W, 64) * sizeof(float) +
                        (size_t)C * g.nblk * (3 + 3 * LT_NPART) * sizeof(double), 256);
}

hipError_t cab_local_tiled_fwd_run(const LocalArgs& a, void* ws, hipStream_t stream) {
    const TiledGeom g = tiled_geom(a.B, a.H, a.W);
    const size_t T = tensor_floats(a);
    float* z[3] = {static_cast<float*>(ws), static_cast<float*>(ws) + T, static_cast<float*>(ws) + 2 * T};
    double* part = reinterpret_cast<double*>(static_cast<float*>(ws) + 3 * T);
    const size_t pstride = (size_t)a.C * g.nblk * 2;
    const dim3 grid(g.nblk, a.C);
    const size_t lds = g.tile_floats * sizeof(float);
    auto stat_src = [&](int s) {
        StatSrc st{};
        st.mode = a.training ? 0 : 2;
        st.part = part + s * pstride;
        st.run_mean = a.st[s].run_mean, st.run_var = a.st[s].run_var;
        st.save_mean = a.save_mean + s * a.C, st.save_invstd = a.save_invstd + s * a.C;
        st.momentum = a.momentum, st.eps = a.eps;
        return st;
    };
    for (int s = 0; s < 3; ++s) {
        FwdStageArgs f{};
        f.in = s == 0 ? a.x : z[s - 1], f.w = a.st[s].w, f.z = z[s];
        f.part = a.training ? part + s * pstride : nullptr;
        f.first = s == 0;
        if (s > 0) f.st = stat_src(s - 1), f.bn_w = a.st[s - 1].bn_w, f.bn_b = a.st[s - 1].bn_b;
        f.B = a.B, f.C = a.C, f.H = a.H, f.W = a.W, f.TR = g.TR, f.nT = g.nT;
        hipLaunchKernelGGL(local_tiled_fwd_stage_kernel, grid, dim3(LT_T), lds, stream, f);
    }
    FwdOutArgs o{};
    o.x = a.x, o.z2 = z[2], o.glob = a.glob, o.gamma = a.gamma, o.out = a.out;
    o.st = stat_src(2), o.bn_w = a.st[2].bn_w, o.bn_b = a.st[2].bn_b;
    o.B = a.B, o.C = a.C, o.H = a.H, o.W = a.W, o.TR = g.TR, o.nT = g.nT;
    hipLaunchKernelGGL(local_tiled_fwd_out_kernel, grid, dim3(LT_T), 0, stream, o);
    return hipGetLastError();
}

hipError_t cab_local_tiled_bwd_run(const LocalArgs& a, void* ws, hipStream_t stream) {
    const TiledGeom g = tiled_geom(a.B, a.H, a.W);
    const size_t T = tensor_floats(a);
    float* base = static_cast<float*>(ws);
    float* z[3] = {base, base + T, base + 2 * T};
    float* d[2] = {base + 3 * T, base + 4 * T};
    double* part_head = reinterpret_cast<double*>(base + 5 * T);
    double* part_stage[3];
    for (int s = 0; s < 3; ++s) part_stage[s] = part_head + (size_t)a.C * g.nblk * 3 + (size_t)s * a.C * g.nblk * LT_NPART;
    const dim3 grid(g.nblk, a.C);
    const size_t lds = g.tile_floats * sizeof(float);
    // recompute the forward chain from x and the saved statistics
    for (int s = 0; s < 3; ++s) {
        FwdStageArgs f{};
        f.in = s == 0 ? a.x : z[s - 1], f.w = a.st[s].w, f.z = z[s], f.part = nullptr, f.first = s == 0;
        if (s > 0) {
            f.st.mode = 1, f.st.mean = a.save_mean + (s - 1) * a.C, f.st.invstd = a.save_invstd + (s - 1) * a.C;
            f.bn_w = a.st[s - 1].bn_w, f.bn_b = a.st[s - 1].bn_b;
        }
        f.B = a.B, f.C = a.C, f.H = a.H, f.W = a.W, f.TR = g.TR, f.nT = g.nT;
        hipLaunchKernelGGL(local_tiled_fwd_stage_kernel, grid, dim3(LT_T), lds, stream, f);
    }
    BwdHeadArgs h{};
    h.g = a.dout, h.x = a.x, h.z2 = z[2], h.glob = a.glob, h.gamma = a.gamma;
    h.mean = a.save_mean + 2 * a.C, h.invstd = a.save_invstd + 2 * a.C, h.bn_w = a.st[2].bn_w, h.bn_b = a.st[2].bn_b;
    h.dy = d[0], h.dx = a.dx, h.dglob = a.dglob, h.part = part_head;
    h.B = a.B, h.C = a.C, h.H = a.H, h.W = a.W, h.TR = g.TR, h.nT = g.nT;
    hipLaunchKernelGGL(local_tiled_bwd_head_kernel, grid, dim3(LT_T), 0, stream, h);
    for (int s = 2; s >= 0; --s) {
        BwdStageArgs b{};
        b.dy = d[(2 - s) & 1], b.z = z[s];
        b.mean = a.save_mean + s * a.C, b.invstd = a.save_invstd + s * a.C, b.bn_w = a.st[s].bn_w;
        b.part_in = s == 2 ? part_head : part_stage[s + 1], b.stride_in = s == 2 ? 3 : LT_NPART;
        b.training = a.training, b.first = s == 0;
        b.in_prev = s == 0 ? a.x : z[s - 1];
        if (s > 0) {
            b.pmean = a.save_mean + (s - 1) * a.C, b.pinvstd = a.save_invstd + (s - 1) * a.C;
            b.pbn_w = a.st[s - 1].bn_w, b.pbn_b = a.st[s - 1].bn_b;
        }
        b.w = a.st[s].w, b.dy_prev = d[(3 - s) & 1], b.dx = a.dx, b.part_out = part_stage[s];
        b.dbn_w = a.st[s].dbn_w, b.dbn_b = a.st[s].dbn_b;
        b.B = a.B, b.C = a.C, b.H = a.H, b.W = a.W, b.TR = g.TR, b.nT = g.nT;
        hipLaunchKernelGGL(local_tiled_bwd_stage_kernel, grid, dim3(LT_T), 2 * lds, stream, b);
    }
    const int total = 3 * a.C * 9 + a.C;
    hipLaunchKernelGGL(local_tiled_finalize_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, part_stage[0], part_stage[1],
                       part_stage[2], part_head, a.st[0].dw, a.st[1].dw, a.st[2].dw, a.glob ? a.dgamma_part : nullptr, a.C,
                       g.nblk);
    return hipGetLastError();
}

}  // namespace cabinet
