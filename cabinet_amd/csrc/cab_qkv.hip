// K6 -- q/k/v producers of the CAB global branch, forward and backward.
//
// Replaces reference src/models/cab.py:137-146, i.e. for x (B,C,H,W):
//   q = ReLU(BN_q(W_q x))                         cab.py:107-112, 137
//   k = PSP_k(ReLU(BN_k(W_k x)))                  cab.py:113-118, 122, 141
//   v = PSP_v(W_v x)                              cab.py:119-121, 123, 145
//   PSP(u) = W_p . cat[u, U(A_s1 u), U(A_s2 u), ...]        cab.py:46-76   (A_s adaptive avg pool to s x s,
//                                                            U bilinear resize back, align_corners=False)
// The reference runs this as ~30 ATen launches forward and ~60 backward (4.1 ms of a 73 ms step at config 3,
// most of it in pooling / resize / cat kernels on tiny tensors and atomics-based resize backward).
//
// MI355X-first restructuring: a 1x1 convolution acts on channels, pooling and resizing act on positions, so
// they commute:  W_p . cat[u, U A_s u ...] = W_0 u + sum_s U( W_s (A_s u) ).  The 5*Kc-channel concat and the
// four full-resolution pyramid maps are never formed; the pyramid terms are (Kc x Kc) x (Kc x s^2) products on
// the pooled bins (110 positions for sizes 1,3,6,8), done as ONE GEMM per branch over a block-expanded
// operand, and come back as a 4-tap gather from LDS.  The three projections share one GEMM (stacked weights).
//   fwd : G(x -> zq|zk|vv) as one 3-job launch, BN statistics + finalize (one launch), plane pass (BN+ReLU, q, kk, pooled bins),
//         output stage (cab_qkv_fused.hip: T = W_i . pooled_i per workgroup, W_0 product, bilinear gathers)   = 4 launches
//         (shapes outside that kernel's range: G(pooled -> T_k, T_v) + G(kk -> k, vv -> v) as one launch, pyramid add = 5)
//   bwd : the adjoint chain in 6 launches (dx of the projections: qkv_dx_kernel, cab_qkv_fused.hip), all deterministic (no atomics)
// GEMMs are the job-batched exact-fp32 MFMA kernels of small_gemm.hip (64x64 tiles: the batch has only 8192 positions).
#include "cab_qkv.hpp"

#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

// geometry of the pooling pyramid, passed by value to the plane kernels
struct PyrGeom {
    int ns, s[4], off[4], coff[4];  // sizes, first bin of each size, first column-bin of each size
    int NB, NBp, NCB;               // bins, bins padded to a multiple of 4, column bins (sum of sizes)
    int H, W;
};

static PyrGeom make_geom(const QkvShape& q) {
    PyrGeom g{};
    g.ns = q.ns, g.H = q.H, g.W = q.W;
    int nb = 0, ncb = 0;
    for (int i = 0; i < q.ns; ++i) {
        g.s[i] = q.sizes[i], g.off[i] = nb, g.coff[i] = ncb;
        nb += q.sizes[i] * q.sizes[i], ncb += q.sizes[i];
    }
    g.NB = nb, g.NBp = (nb + 3) & ~3, g.NCB = ncb;
    return g;
}

constexpr int QC_T = 512;  // threads of the channel-resident kernels
constexpr int SG_TILE_P = 64;  // positions per tile of sg_gemm (small_gemm.hip): one statistics partial each
__device__ __forceinline__ int bin_start(int r, int n, int s) { return (r * n) / s; }            // floor(r*n/s)
__device__ __forceinline__ int bin_end(int r, int n, int s) { return ((r + 1) * n + s - 1) / s; }  // ceil((r+1)*n/s)



// ------------------------------------------------------------------------------- weight staging
struct TrJob {
    const float* in;  // rows x cols, row stride ldi
    float* out;       // transpose: out[c * ldo + off + r]      copy: out[(off + r) * ldo + c]
    int ldi, rows, cols, ldo, off, copy;
};
struct TrJobs {
    TrJob j[8];
};

__global__ void stage_weights_kernel(TrJobs jobs) {
    __shared__ float tile[32][33];
    const TrJob& t = jobs.j[blockIdx.z];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (c0 >= t.cols || r0 >= t.rows) return;
    if (t.copy) {
        for (int j = threadIdx.y; j < 32; j += 8) {
            const int r = r0 + j, c = c0 + threadIdx.x;
            if (r < t.rows && c < t.cols) t.out[(size_t)(t.off + r) * t.ldo + c] = t.in[(size_t)r * t.ldi + c];
        }
        return;
    }
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < t.rows && c < t.cols) tile[j][threadIdx.x] = t.in[(size_t)r * t.ldi + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < t.rows && c < t.cols) t.out[(size_t)c * t.ldo + t.off + r] = tile[threadIdx.x][j];
    }
}

static void stage_weights(const TrJobs& jobs, int n, hipStream_t stream) {
    int mr = 0, mc = 0;
    for (int i = 0; i < n; ++i) {
        mr = jobs.j[i].rows > mr ? jobs.j[i].rows : mr;
        mc = jobs.j[i].cols > mc ? jobs.j[i].cols : mc;
    }
    hipLaunchKernelGGL(stage_weights_kernel, dim3(ceil_div(mc, 32), ceil_div(mr, 32), n), dim3(32, 8), 0, stream, jobs);
}

// ------------------------------------------------------------------------------- forward plane pass
// one workgroup per (b, m) plane of the stacked projection [zq | zk | vv]:
//   zq -> q = relu(bn(zq));   zk -> kk = relu(bn(zk)) and its pooled bins;   vv -> its pooled bins.
// pooled is written block-expanded: row (i, c) holds the bins of size i of channel c and zeros elsewhere, so
// that  T[m][bin] = sum_{(i,c)} W_p[m][Kc + i*Kc + c] * pooled[(i,c)][bin]  is a plain GEMM.
__global__ __launch_bounds__(256) void qkv_plane_fwd_kernel(const float* __restrict__ zqk, const float* __restrict__ vv,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ bnq_w, const float* __restrict__ bnq_b,
                                                             const float* __restrict__ bnk_w, const float* __restrict__ bnk_b,
                                                             int Kc, int Vc, PyrGeom g, float* __restrict__ q,
                                                             float* __restrict__ kk, float* __restrict__ pooled_k,
                                                             float* __restrict__ pooled_v) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = g.H * g.W, Mtot = 2 * Kc + Vc;
    const int b = blockIdx.x / Mtot, m = blockIdx.x - b * Mtot, tid = threadIdx.x;
    const int WS = g.W + 1;            // padded row stride: column sums below walk rows in lock step (bank conflicts otherwise)
    float* plane = smem;               // [H][WS]
    float* rowpart = smem + g.H * WS;  // [H][NCB]
    float* binv = rowpart + g.H * g.NCB;  // [NBp]
    float* pooled;
    int ch, nch;
    if (m < 2 * Kc) {
        const float* src = zqk + ((size_t)b * 2 * Kc + m) * P;
        const bool is_q = m < Kc;
        ch = is_q ? m : m - Kc;
        const float gam = is_q ? bnq_w[ch] : bnk_w[ch], bet = is_q ? bnq_b[ch] : bnk_b[ch];
        const float mu = mean[m], inv = invstd[m];
        float* dst = (is_q ? q : kk) + ((size_t)b * Kc + ch) * P;
        if (is_q) {
            for (int p = tid; p < P; p += 256) dst[p] = fmaxf(fmaf((src[p] - mu) * inv, gam, bet), 0.f);
            return;
        }
        for (int p = tid; p < P; p += 256) {
            const float v = fmaxf(fmaf((src[p] - mu) * inv, gam, bet), 0.f);  // same expression as the backward mask
            const int y = p / g.W;
            dst[p] = v;
            plane[y * WS + (p - y * g.W)] = v;
        }
        pooled = pooled_k, nch = Kc;
    } else {
        ch = m - 2 * Kc, nch = Vc, pooled = pooled_v;
        const float* src = vv + ((size_t)b * Vc + ch) * P;
        for (int p = tid; p < P; p += 256) {
            const int y = p / g.W;
            plane[y * WS + (p - y * g.W)] = src[p];
        }
    }
    __syncthreads();
    // separable adaptive average pooling: column bins per row, then row bins
    for (int it = tid; it < g.H * g.NCB; it += 256) {
        const int y = it / g.NCB, cb = it - y * g.NCB;
        int i = 0;
        while (i + 1 < g.ns && cb >= g.coff[i + 1]) ++i;
        const int c = cb - g.coff[i], xs = bin_start(c, g.W, g.s[i]), xe = bin_end(c, g.W, g.s[i]);
        float acc = 0.f;
        for (int x = xs; x < xe; ++x) acc += plane[y * WS + x];
        rowpart[it] = acc;
    }
    __syncthreads();
    for (int t = tid; t < g.NBp; t += 256) {
        float val = 0.f;
        if (t < g.NB) {
            int i = 0;
            while (i + 1 < g.ns && t >= g.off[i + 1]) ++i;
            const int s = g.s[i], r = (t - g.off[i]) / s, c = (t - g.off[i]) - r * s;
            const int ys = bin_start(r, g.H, s), ye = bin_end(r, g.H, s);
            const int xs = bin_start(c, g.W, s), xe = bin_end(c, g.W, s);
            float acc = 0.f;
            for (int y = ys; y < ye; ++y) acc += rowpart[y * g.NCB + g.coff[i] + c];
            val = acc / (float)((ye - ys) * (xe - xs));
        }
        binv[t] = val;
    }
    __syncthreads();
    for (int it = tid; it < g.ns * g.NBp; it += 256) {
        const int i = it / g.NBp, t = it - i * g.NBp;
        const bool mine = t >= g.off[i] && t < g.off[i] + g.s[i] * g.s[i];
        pooled[(((size_t)b * g.ns + i) * nch + ch) * g.NBp + t] = mine ? binv[t] : 0.f;
    }
}

// ---- round 5: the same pass with the BatchNorm finalize in its prologue -----------------------------------------------------------
// With `part` (training): the (mean, M2) partials the projection GEMM's epilogue left per (channel, image, 64-column tile) are merged
// here -- equal counts, fixed order, double -- by wave 0 of every workgroup of the channel (128 pairs at config 3: two per lane) while
// the other waves' plane loads are in flight; the b == 0 workgroup writes save_mean / save_invstd and the running buffers: the
// statistics launch is gone (4 -> 3 launches).  Without `part` the workgroup reads save_mean / save_invstd (eval mode, or shapes the
// GEMM epilogue does not take: qk_bn_stats_kernel ran in front).
// Measured and not kept: ONE WAVE per plane (four 16-byte loads per lane, the phases ordered by the wave's own in-order LDS queue, no
// s_barrier, four planes per workgroup): K6 forward 77 vs 72 us at config 3, 83 vs 60 us at config 5 -- a plane's four dependent
// phases are a latency chain, and 256 threads per plane walk it with a quarter of the work per thread.
__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
struct PlaneWArgs {
    const float *zqk, *vv;
    const float* part;            // [2Kc][npairs][2] (mean, M2) per 64-value tile, or null
    int npairs;                   // pairs per channel (B * tiles of 64 positions)
    float momentum, eps;
    float *q_rm, *q_rv, *k_rm, *k_rv, *save_mean, *save_invstd;
    const float *bnq_w, *bnq_b, *bnk_w, *bnk_b;
    int B, Kc, Vc;
    float *q, *kk, *pooled_k, *pooled_v;
};
__global__ __launch_bounds__(256) void qkv_plane_fwd_w_kernel(PlaneWArgs a, PyrGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float s_stat[2];
    const int P = g.H * g.W, Mtot = 2 * a.Kc + a.Vc, tid = threadIdx.x;
    const int b = blockIdx.x / Mtot, m = blockIdx.x - b * Mtot;
    const int WS = g.W + 1;
    float* plane = smem;                       // [H][WS]
    float* rowpart = plane + g.H * WS;         // [H][NCB]
    float* binv = rowpart + g.H * g.NCB;       // [NBp]
    const bool is_q = m < a.Kc, is_v = m >= 2 * a.Kc;
    const int ch = is_q ? m : (is_v ? m - 2 * a.Kc : m - a.Kc);
    const float* src = is_v ? a.vv + ((size_t)b * a.Vc + ch) * P : a.zqk + ((size_t)b * 2 * a.Kc + m) * P;
    // the first values of the plane are requested before the statistics are touched
    constexpr int PRE = 4;
    float pre[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) pre[u] = src[min(tid + 256 * u, P - 1)];
    float mu = 0.f, inv = 1.f, gam = 1.f, bet = 0.f;
    if (!is_v) {
        gam = is_q ? a.bnq_w[ch] : a.bnk_w[ch], bet = is_q ? a.bnq_b[ch] : a.bnk_b[ch];
        if (a.part != nullptr) {
            if (tid < 64) {
                const float* pp = a.part + (size_t)m * a.npairs * 2;
                double sm = 0.0;
                for (int i = tid; i < a.npairs; i += 64) sm += (double)pp[2 * i];
                const double mean = wave_sum_d(sm) / (double)a.npairs;
                double m2 = 0.0;
                for (int i = tid; i < a.npairs; i += 64) {
                    const double d = (double)pp[2 * i] - mean;
                    m2 += (double)pp[2 * i + 1] + 64.0 * d * d;
                }
                const double count = 64.0 * (double)a.npairs;
                const double var = wave_sum_d(m2) / count;
                if (tid == 0) {
                    const float muf = (float)mean, invf = (float)(1.0 / sqrt(var + (double)a.eps));
                    s_stat[0] = muf, s_stat[1] = invf;
                    if (b == 0) {
                        a.save_mean[m] = muf, a.save_invstd[m] = invf;
                        float* rm = is_q ? a.q_rm + ch : a.k_rm + ch;
                        float* rv = is_q ? a.q_rv + ch : a.k_rv + ch;
                        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
                        *rm = (float)((1.0 - (double)a.momentum) * (double)*rm + (double)a.momentum * mean);
                        *rv = (float)((1.0 - (double)a.momentum) * (double)*rv + (double)a.momentum * unbiased);
                    }
                }
            }
            __syncthreads();
            mu = s_stat[0], inv = s_stat[1];
        } else {
            mu = a.save_mean[m], inv = a.save_invstd[m];
        }
    }
    float* dst = is_q ? a.q + ((size_t)b * a.Kc + ch) * P : a.kk + ((size_t)b * a.Kc + ch) * P;
    if (is_q) {
        auto put = [&](int p, float v) { dst[p] = fmaxf(fmaf((v - mu) * inv, gam, bet), 0.f); };
#pragma unroll
        for (int u = 0; u < PRE; ++u)
            if (tid + 256 * u < P) put(tid + 256 * u, pre[u]);
        for (int p = tid + 256 * PRE; p < P; p += 256) put(p, src[p]);
        return;
    }
    {
        const float inv_w = 1.f / (float)g.W;
        auto put = [&](int p, float v) {
            if (!is_v) {
                v = fmaxf(fmaf((v - mu) * inv, gam, bet), 0.f);  // same expression as the backward mask
                dst[p] = v;
            }
            const int y = idiv_small(p, inv_w);
            plane[y * WS + (p - y * g.W)] = v;
        };
#pragma unroll
        for (int u = 0; u < PRE; ++u)
            if (tid + 256 * u < P) put(tid + 256 * u, pre[u]);
        for (int p = tid + 256 * PRE; p < P; p += 256) put(p, src[p]);
    }
    __syncthreads();
    // separable adaptive average pooling: column bins per row, then row bins (the arithmetic of qkv_plane_fwd_kernel, same order)
    for (int it = tid; it < g.H * g.NCB; it += 256) {
        const int y = it / g.NCB, cb = it - y * g.NCB;
        int i = 0;
        while (i + 1 < g.ns && cb >= g.coff[i + 1]) ++i;
        const int c = cb - g.coff[i], xs = bin_start(c, g.W, g.s[i]), xe = bin_end(c, g.W, g.s[i]);
        float acc = 0.f;
        for (int x = xs; x < xe; ++x) acc += plane[y * WS + x];
        rowpart[it] = acc;
    }
    __syncthreads();
    for (int t = tid; t < g.NBp; t += 256) {
        float val = 0.f;
        if (t < g.NB) {
            int i = 0;
            while (i + 1 < g.ns && t >= g.off[i + 1]) ++i;
            const int s = g.s[i], r = (t - g.off[i]) / s, c = (t - g.off[i]) - r * s;
            const int ys = bin_start(r, g.H, s), ye = bin_end(r, g.H, s);
            const int xs = bin_start(c, g.W, s), xe = bin_end(c, g.W, s);
            float acc = 0.f;
            for (int y = ys; y < ye; ++y) acc += rowpart[y * g.NCB + g.coff[i] + c];
            val = acc / (float)((ye - ys) * (xe - xs));
        }
        binv[t] = val;
    }
    __syncthreads();
    float* pooled = is_v ? a.pooled_v : a.pooled_k;
    const int nch = is_v ? a.Vc : a.Kc;
    for (int it = tid; it < g.ns * g.NBp; it += 256) {
        const int i = it / g.NBp, t = it - i * g.NBp;
        const bool mine = t >= g.off[i] && t < g.off[i] + g.s[i] * g.s[i];
        pooled[(((size_t)b * g.ns + i) * nch + ch) * g.NBp + t] = mine ? binv[t] : 0.f;
    }
}

// k[b][m][p] += sum_i U_i(T[b][m][bins of size i])(p), same for v; one workgroup per plane.
// The bilinear taps of every (size, row) and (size, column) are tabulated once per workgroup (ns * (H + W) entries); per
// output the loop is two table reads, four T reads and the lerp.  Computing the taps per output (8 float->int tap
// evaluations and a division per position) made this elementwise pass 17 us.
__global__ __launch_bounds__(256) void pyramid_add_kernel(const float* __restrict__ Tk, const float* __restrict__ Tv, int Kc,
                                                           int Vc, PyrGeom g, float* __restrict__ k,
                                                           float* __restrict__ v) {
    extern __shared__ float sm[];
    float* t = sm;                                          // [NBp] this plane's pyramid terms
    float* lam = t + g.NBp;                                 // [ns][H + W] interpolation weight of the upper tap
    int* tap = reinterpret_cast<int*>(lam + g.ns * (g.H + g.W));  // [ns][H + W] lower tap | upper tap << 16
    const int P = g.H * g.W, Mtot = Kc + Vc, HW = g.H + g.W;
    const int b = blockIdx.x / Mtot, m = blockIdx.x - b * Mtot;
    const bool is_k = m < Kc;
    const float* src = is_k ? Tk + ((size_t)b * Kc + m) * g.NBp : Tv + ((size_t)b * Vc + (m - Kc)) * g.NBp;
    float* dst = is_k ? k + ((size_t)b * Kc + m) * P : v + ((size_t)b * Vc + (m - Kc)) * P;
    for (int i = threadIdx.x; i < g.NBp; i += 256) t[i] = src[i];
    for (int e = threadIdx.x; e < g.ns * HW; e += 256) {
        const int i = e / HW, r = e - i * HW, sz = g.s[i];
        int i0, i1;
        float l;
        if (r < g.H)
            bilinear_taps(r, (float)sz / (float)g.H, sz, i0, i1, l);
        else
            bilinear_taps(r - g.H, (float)sz / (float)g.W, sz, i0, i1, l);
        lam[e] = l;
        tap[e] = i0 | (i1 << 16);
    }
    __syncthreads();
    int oy = threadIdx.x / g.W, ox = threadIdx.x - oy * g.W;
    const int dy = 256 / g.W, dx = 256 - dy * g.W;
    for (int p = threadIdx.x; p < P; p += 256) {
        const float cur = dst[p];
        float acc = 0.f;
        for (int i = 0; i < g.ns; ++i) {
            const int sz = g.s[i], ty = tap[i * HW + oy], tx = tap[i * HW + g.H + ox];
            const float ly = lam[i * HW + oy], lx = lam[i * HW + g.H + ox];
            const float* r0 = t + g.off[i] + (ty & 0xffff) * sz;
            const float* r1 = t + g.off[i] + (ty >> 16) * sz;
            const int x0 = tx & 0xffff, x1 = tx >> 16;
            acc += (1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1]);
        }
        dst[p] = cur + acc;
        oy += dy, ox += dx;
        if (ox >= g.W) ox -= g.W, ++oy;
    }
}

// ------------------------------------------------------------------------------- backward plane passes
// dT[b][m][bin] = sum_p U_i[p][bin] * d[b][m][p]   (adjoint of the pyramid resize), d = dk | dv; separable:
// colpart[oy][(i,xs)] = sum_ox wx_i(ox,xs) d[oy][ox], then dT[(i,ys,xs)] = sum_oy wy_i(oy,ys) colpart[oy][(i,xs)]
__global__ __launch_bounds__(256) void pyramid_adjoint_kernel(const float* __restrict__ dk, const float* __restrict__ dv,
                                                               int Kc, int Vc, PyrGeom g, float* __restrict__ dTk,
                                                               float* __restrict__ dTv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = g.H * g.W, Mtot = Kc + Vc, tid = threadIdx.x;
    const int b = blockIdx.x / Mtot, m = blockIdx.x - b * Mtot;
    const bool is_k = m < Kc;
    const float* src = is_k ? dk + ((size_t)b * Kc + m) * P : dv + ((size_t)b * Vc + (m - Kc)) * P;
    float* dst = is_k ? dTk + ((size_t)b * Kc + m) * g.NBp : dTv + ((size_t)b * Vc + (m - Kc)) * g.NBp;
    // row strides W+1 / H+1: with W = 32 the unpadded tables put a whole column in one LDS bank (32-way conflicts on every
    // read of the dot products below; the kernel took 49 us for 2048 planes)
    const int WS = g.W + 1, HS = g.H + 1;
    float* plane = smem;                     // [H][WS]
    float* wx = plane + g.H * WS;            // [NCB][WS]
    float* wy = wx + g.NCB * WS;             // [NCB][HS]
    float* colpart = wy + g.NCB * HS;        // [H][NCB]
    for (int p = tid; p < P; p += 256) {
        const int oy = p / g.W;
        plane[oy * WS + (p - oy * g.W)] = src[p];
    }
    for (int it = tid; it < g.NCB * g.W; it += 256) {
        const int cb = it / g.W, ox = it - cb * g.W;
        int i = 0;
        while (i + 1 < g.ns && cb >= g.coff[i + 1]) ++i;
        const int xs = cb - g.coff[i];
        int x0, x1;
        float lx;
        bilinear_taps(ox, (float)g.s[i] / (float)g.W, g.s[i], x0, x1, lx);
        wx[cb * WS + ox] = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
    }
    for (int it = tid; it < g.NCB * g.H; it += 256) {
        const int cb = it / g.H, oy = it - cb * g.H;
        int i = 0;
        while (i + 1 < g.ns && cb >= g.coff[i + 1]) ++i;
        const int ys = cb - g.coff[i];
        int y0, y1;
        float ly;
        bilinear_taps(oy, (float)g.s[i] / (float)g.H, g.s[i], y0, y1, ly);
        wy[cb * HS + oy] = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
    }
    __syncthreads();
    for (int it = tid; it < g.H * g.NCB; it += 256) {
        const int oy = it / g.NCB, cb = it - oy * g.NCB;
        const float* w = wx + cb * WS;
        const float* row = plane + oy * WS;
        float acc = 0.f;
        for (int ox = 0; ox < g.W; ++ox) acc += w[ox] * row[ox];
        colpart[it] = acc;
    }
    __syncthreads();
    for (int t = tid; t < g.NBp; t += 256) {
        float acc = 0.f;
        if (t < g.NB) {
            int i = 0;
            while (i + 1 < g.ns && t >= g.off[i + 1]) ++i;
            const int s = g.s[i], ys = (t - g.off[i]) / s, xs = (t - g.off[i]) - ys * s;
            const float* w = wy + (g.coff[i] + ys) * HS;
            for (int oy = 0; oy < g.H; ++oy) acc += w[oy] * colpart[oy * g.NCB + g.coff[i] + xs];
        }
        dst[t] = acc;
    }
}

// one workgroup per (b, m) plane of [q | k | v]:
//   q : dy = dq * 1[bn(zq) > 0]                                      -> dzqk rows [0,Kc),  BN-backward partial sums
//   k : dy = (lin_k + A^T dpooled_k) * 1[bn(zk) > 0]                 -> dzqk rows [Kc,2Kc), BN-backward partial sums
//   v : dvv += A^T dpooled_v                                          (in place)
// A^T = adjoint of the adaptive average pools; dpooled comes block-expanded (row (i,c) holds the bins of size i)
__global__ __launch_bounds__(256) void qkv_plane_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ lin_k,
                                                             const float* __restrict__ zqk, const float* __restrict__ dpe_k,
                                                             const float* __restrict__ dpe_v, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ bnq_w, const float* __restrict__ bnq_b,
                                                             const float* __restrict__ bnk_w, const float* __restrict__ bnk_b,
                                                             int B, int Kc, int Vc, PyrGeom g, float* __restrict__ dzqk,
                                                             float* __restrict__ dvv, float* __restrict__ bnpart) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4];
    const int P = g.H * g.W, Mtot = 2 * Kc + Vc, tid = threadIdx.x;
    const int b = blockIdx.x / Mtot, m = blockIdx.x - b * Mtot;
    float* dp = smem;            // [NBp]   dpooled / bin size
    float* E = smem + g.NBp;     // [H][NCB] row-expanded
    int* xr = reinterpret_cast<int*>(E + g.H * g.NCB);  // [ns][W]  lo | hi << 8 : column bins of size i that contain x
    int* yr = xr + g.ns * g.W;                            // [ns][H]  row bins that contain y
    const bool is_q = m < Kc, is_v = m >= 2 * Kc;
    if (!is_q) {
        // which bins cover a coordinate is a contiguous range; tabulate it once (the per-pixel form spent ~24 emulated
        // integer divisions per pixel and size)
        for (int it = tid; it < g.ns * (g.W + g.H); it += 256) {
            const bool isx = it < g.ns * g.W;
            const int j = isx ? it : it - g.ns * g.W, n = isx ? g.W : g.H;
            const int i = j / n, c = j - i * n, s = g.s[i];
            int lo = s, hi = -1;
            for (int r = 0; r < s; ++r)
                if (bin_start(r, n, s) <= c && c < bin_end(r, n, s)) lo = min(lo, r), hi = max(hi, r);
            (isx ? xr : yr)[j] = lo | (hi << 8);
        }
        const int ch = is_v ? m - 2 * Kc : m - Kc, nch = is_v ? Vc : Kc;
        const float* dpe = is_v ? dpe_v : dpe_k;
        for (int t = tid; t < g.NBp; t += 256) {
            float val = 0.f;
            if (t < g.NB) {
                int i = 0;
                while (i + 1 < g.ns && t >= g.off[i + 1]) ++i;
                const int s = g.s[i], r = (t - g.off[i]) / s, c = (t - g.off[i]) - r * s;
                const int cnt = (bin_end(r, g.H, s) - bin_start(r, g.H, s)) * (bin_end(c, g.W, s) - bin_start(c, g.W, s));
                val = dpe[(((size_t)b * g.ns + i) * nch + ch) * g.NBp + t] / (float)cnt;
            }
            dp[t] = val;
        }
        __syncthreads();
        for (int it = tid; it < g.H * g.NCB; it += 256) {
            const int y = it / g.NCB, cb = it - y * g.NCB;
            int i = 0;
            while (i + 1 < g.ns && cb >= g.coff[i + 1]) ++i;
            const int s = g.s[i], c = cb - g.coff[i], rr = yr[i * g.H + y];
            float acc = 0.f;
            for (int r = rr & 255; r <= (rr >> 8); ++r) acc += dp[g.off[i] + r * s + c];
            E[it] = acc;
        }
        __syncthreads();
    }
    auto pool_adjoint = [&](int p) {
        const int y = p / g.W, x = p - y * g.W;
        const float* Ey = E + y * g.NCB;
        float acc = 0.f;
        for (int i = 0; i < g.ns; ++i) {
            const int cr = xr[i * g.W + x];
            for (int c = cr & 255; c <= (cr >> 8); ++c) acc += Ey[g.coff[i] + c];
        }
        return acc;
    };
    if (is_v) {
        float* d = dvv + ((size_t)b * Vc + (m - 2 * Kc)) * P;
        for (int p = tid; p < P; p += 256) d[p] += pool_adjoint(p);
        return;
    }
    const int ch = is_q ? m : m - Kc;
    const float mu = mean[m], inv = invstd[m];
    const float gam = is_q ? bnq_w[ch] : bnk_w[ch], bet = is_q ? bnq_b[ch] : bnk_b[ch];
    const float* z = zqk + ((size_t)b * 2 * Kc + m) * P;
    const float* din = is_q ? dq + ((size_t)b * Kc + ch) * P : lin_k + ((size_t)b * Kc + ch) * P;
    float* dout = dzqk + ((size_t)b * 2 * Kc + m) * P;
    float s1 = 0.f, s2 = 0.f;
    for (int p = tid; p < P; p += 256) {
        const float xh = (z[p] - mu) * inv;
        float d = din[p];
        if (!is_q) d += pool_adjoint(p);
        const float dy = fmaf(xh, gam, bet) > 0.f ? d : 0.f;
        dout[p] = dy;
        s1 += dy, s2 += dy * xh;
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (tid == 0) {
        bnpart[(size_t)m * B + b] = s1;
        bnpart[((size_t)2 * Kc + m) * B + b] = s2;
    }
}

// BN backward, in place on dzqk (B, 2Kc, P):  dz = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat))  (training)
// or gamma*invstd*dy (eval); planes with b == 0 also write dgamma = sum dy*xhat, dbeta = sum dy
__global__ __launch_bounds__(256) void qkv_bn_bwd_kernel(const float* __restrict__ zqk, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ bnq_w,
                                                          const float* __restrict__ bnk_w, const float* __restrict__ bnpart,
                                                          int B, int Kc, int P, int training, float* __restrict__ dzqk,
                                                          float* __restrict__ dbnq_w, float* __restrict__ dbnq_b,
                                                          float* __restrict__ dbnk_w, float* __restrict__ dbnk_b) {
    const int b = blockIdx.x / (2 * Kc), m = blockIdx.x - b * 2 * Kc;
    float S1 = 0.f, S2 = 0.f;
    for (int i = 0; i < B; ++i) {
        S1 += bnpart[(size_t)m * B + i];
        S2 += bnpart[((size_t)2 * Kc + m) * B + i];
    }
    const bool is_q = m < Kc;
    const int ch = is_q ? m : m - Kc;
    if (b == 0 && threadIdx.x == 0) {
        (is_q ? dbnq_w : dbnk_w)[ch] = S2;
        (is_q ? dbnq_b : dbnk_b)[ch] = S1;
    }
    const float inv_n = 1.f / ((float)B * (float)P);
    const float m1 = training ? S1 * inv_n : 0.f, m2 = training ? S2 * inv_n : 0.f;
    const float mu = mean[m], inv = invstd[m], gi = (is_q ? bnq_w[ch] : bnk_w[ch]) * inv;
    const float* z = zqk + ((size_t)b * 2 * Kc + m) * P;
    float* d = dzqk + ((size_t)b * 2 * Kc + m) * P;
    for (int p = threadIdx.x; p < P; p += 256) d[p] = gi * (d[p] - m1 - (z[p] - mu) * inv * m2);
}

// Backward plane pass + BatchNorm backward of ONE stacked channel over all B images in one workgroup -- the idea of K5
// (cab_local.hip): a per-channel BatchNorm never mixes channels, so the B planes of a channel are an independent problem with an
// IN-WORKGROUP reduction.  Replaces qkv_plane_bwd_kernel + qkv_bn_bwd_kernel and their bnpart round trip (7 -> 6 launches):
//   q : dy = dq 1[bn(zq) > 0]                        -> sums -> dz_q                  (dzqk rows [0, Kc))
//   k : dy = (lin_k + A^T dpooled_k) 1[bn(zk) > 0]   -> sums -> dz_k                  (dzqk rows [Kc, 2Kc))
//   v : dvv += A^T dpooled_v                                                          (in place)
// dy of the B planes waits in LDS for the channel's two sums (double), then dz = gamma invstd (dy - mean(dy) - xhat mean(dy xhat)).
static size_t lds_channel_bwd(const PyrGeom& g, int B) {
    return ((size_t)B * g.NBp + (size_t)B * g.H * g.NCB + (size_t)g.ns * (g.W + g.H) + (size_t)B * g.H * g.W) * sizeof(float);
}

__global__ __launch_bounds__(QC_T) void qkv_channel_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ lin_k,
                                                               const float* __restrict__ zqk, const float* __restrict__ dpe_k,
                                                               const float* __restrict__ dpe_v, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ bnq_w,
                                                               const float* __restrict__ bnq_b, const float* __restrict__ bnk_w,
                                                               const float* __restrict__ bnk_b, int B, int Kc, int Vc, PyrGeom g,
                                                               int training, float* __restrict__ dzqk, float* __restrict__ dvv,
                                                               float* __restrict__ dbnq_w, float* __restrict__ dbnq_b,
                                                               float* __restrict__ dbnk_w, float* __restrict__ dbnk_b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double s_red[2][QC_T / 64];
    __shared__ float s_m[2];
    const int P = g.H * g.W, m = blockIdx.x, tid = threadIdx.x, C2 = 2 * Kc, nrp = g.H * g.NCB;
    const bool is_q = m < Kc, is_v = m >= C2;
    float* dp = smem;                                   // [B][NBp]   dpooled / bin size
    float* E = dp + (size_t)B * g.NBp;                  // [B][H][NCB] row-expanded
    int* xr = reinterpret_cast<int*>(E + (size_t)B * nrp);  // [ns][W]  lo | hi << 8 : column bins of size i that contain x
    int* yr = xr + g.ns * g.W;                              // [ns][H]
    float* dyb = reinterpret_cast<float*>(yr + g.ns * g.H);  // [B][P]   dy of this channel
    const float inv_w = 1.f / (float)g.W;
    if (!is_q) {
        for (int it = tid; it < g.ns * (g.W + g.H); it += QC_T) {
            const bool isx = it < g.ns * g.W;
            const int j = isx ? it : it - g.ns * g.W, n = isx ? g.W : g.H;
            const int i = j / n, c = j - i * n, s = g.s[i];
            int lo = s, hi = -1;
            for (int r = 0; r < s; ++r)
                if (bin_start(r, n, s) <= c && c < bin_end(r, n, s)) lo = min(lo, r), hi = max(hi, r);
            (isx ? xr : yr)[j] = lo | (hi << 8);
        }
        const int ch = is_v ? m - C2 : m - Kc, nch = is_v ? Vc : Kc;
        const float* dpe = is_v ? dpe_v : dpe_k;
        for (int it = tid; it < B * g.NBp; it += QC_T) {
            const int b = it / g.NBp, t = it - b * g.NBp;
            float val = 0.f;
            if (t < g.NB) {
                int i = 0;
                while (i + 1 < g.ns && t >= g.off[i + 1]) ++i;
                const int s = g.s[i], r = (t - g.off[i]) / s, c = (t - g.off[i]) - r * s;
                const int cnt = (bin_end(r, g.H, s) - bin_start(r, g.H, s)) * (bin_end(c, g.W, s) - bin_start(c, g.W, s));
                val = dpe[(((size_t)b * g.ns + i) * nch + ch) * g.NBp + t] / (float)cnt;
            }
            dp[it] = val;
        }
        __syncthreads();
        for (int it = tid; it < B * nrp; it += QC_T) {
            const int b = it / nrp, r0 = it - b * nrp, y = r0 / g.NCB, cb = r0 - y * g.NCB;
            int i = 0;
            while (i + 1 < g.ns && cb >= g.coff[i + 1]) ++i;
            const int s = g.s[i], c = cb - g.coff[i], rr = yr[i * g.H + y];
            float acc = 0.f;
            for (int r = rr & 255; r <= (rr >> 8); ++r) acc += dp[b * g.NBp + g.off[i] + r * s + c];
            E[it] = acc;
        }
        __syncthreads();
    }
    auto pool_adjoint = [&](int b, int p) {
        const int y = idiv_small(p, inv_w), x = p - y * g.W;
        const float* Ey = E + (size_t)b * nrp + y * g.NCB;
        float acc = 0.f;
        for (int i = 0; i < g.ns; ++i) {
            const int cr = xr[i * g.W + x];
            for (int c = cr & 255; c <= (cr >> 8); ++c) acc += Ey[g.coff[i] + c];
        }
        return acc;
    };
    if (is_v) {
        for (int b = 0; b < B; ++b) {
            float* d = dvv + ((size_t)b * Vc + (m - C2)) * P;
            for (int p = tid; p < P; p += QC_T) d[p] += pool_adjoint(b, p);
        }
        return;
    }
    const int ch = is_q ? m : m - Kc;
    const float mu = mean[m], inv = invstd[m];
    const float gam = is_q ? bnq_w[ch] : bnk_w[ch], bet = is_q ? bnq_b[ch] : bnk_b[ch];
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* z = zqk + ((size_t)b * C2 + m) * P;
        const float* din = is_q ? dq + ((size_t)b * Kc + ch) * P : lin_k + ((size_t)b * Kc + ch) * P;
        for (int p = tid; p < P; p += QC_T) {
            const float xh = (z[p] - mu) * inv;
            float d = din[p];
            if (!is_q) d += pool_adjoint(b, p);
            const float dy = fmaf(xh, gam, bet) > 0.f ? d : 0.f;
            dyb[(size_t)b * P + p] = dy;
            s1 += (double)dy, s2 = fma((double)dy, (double)xh, s2);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o, 64), s2 += __shfl_xor(s2, o, 64);
    if ((tid & 63) == 0) s_red[0][tid >> 6] = s1, s_red[1][tid >> 6] = s2;
    __syncthreads();
    if (tid == 0) {
        s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < QC_T / 64; ++w) s1 += s_red[0][w], s2 += s_red[1][w];
        (is_q ? dbnq_w : dbnk_w)[ch] = (float)s2;
        (is_q ? dbnq_b : dbnk_b)[ch] = (float)s1;
        const double inv_n = 1.0 / ((double)B * (double)P);
        s_m[0] = training ? (float)(s1 * inv_n) : 0.f;
        s_m[1] = training ? (float)(s2 * inv_n) : 0.f;
    }
    __syncthreads();
    const float m1 = s_m[0], m2 = s_m[1], gi = gam * inv;
    for (int b = 0; b < B; ++b) {
        const float* z = zqk + ((size_t)b * C2 + m) * P;
        float* d = dzqk + ((size_t)b * C2 + m) * P;
        for (int p = tid; p < P; p += QC_T) d[p] = gi * (dyb[(size_t)b * P + p] - m1 - (z[p] - mu) * inv * m2);
    }
}

// ------------------------------------------------------------------------------- host side
static size_t fbytes(size_t n) { return align_up(n * sizeof(float), 256); }

static size_t lds_fwd_plane(const PyrGeom& g) { return ((size_t)g.H * (g.W + 1) + (size_t)g.H * g.NCB + g.NBp) * sizeof(float); }
static size_t lds_adjoint(const PyrGeom& g) {
    return ((size_t)g.H * (g.W + 1) + (size_t)g.NCB * (g.W + 1 + g.H + 1) + (size_t)g.H * g.NCB) * sizeof(float);
}
static size_t lds_bwd_plane(const PyrGeom& g) {
    return ((size_t)g.NBp + (size_t)g.H * g.NCB + (size_t)g.ns * (g.W + g.H)) * sizeof(float);
}

const char* qkv_unsupported(const QkvShape& s) {
    if (s.ns < 1 || s.ns > 4) return "1..4 pyramid sizes";
    for (int i = 0; i < s.ns; ++i)
        if (s.sizes[i] < 1 || s.sizes[i] > 16) return "pyramid sizes in 1..16";
    const PyrGeom g = make_geom(s);
    if (g.NBp > 256) return "at most 256 pyramid bins";
    if ((s.C % 16) || (s.Kc % 16) || (s.Vc % 16)) return "channel counts that are multiples of 16";
    if (lds_fwd_plane(g) > 64 * 1024 || lds_adjoint(g) > 64 * 1024) return "H*W small enough for one plane in LDS";
    return nullptr;
}

// BatchNorm statistics AND finalize of both projections in ONE launch: a workgroup per channel (channels [0,Kc) -> query BN
// buffers, [Kc,2Kc) -> key BN buffers) walks the channel's B planes with QS_T threads (every thread's loads are independent:
// one round trip for B*P = 8192 values), accumulates sum and sum of squares in double, and its first thread writes
// save_mean / save_invstd and updates the running buffers.  It replaces the (b, c)-row pass + per-channel finalize pair:
// one launch boundary and one dependent small kernel less on a chain whose every link is latency-bound.
constexpr int QS_T = 512;
__global__ __launch_bounds__(QS_T) void qk_bn_stats_kernel(const float* __restrict__ zqk, int B, int Kc, int P, int training,
                                                           float momentum, float eps, float* __restrict__ q_rm,
                                                           float* __restrict__ q_rv, float* __restrict__ k_rm,
                                                           float* __restrict__ k_rv, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd) {
    __shared__ double s_red[2][QS_T / 64];
    const int c = blockIdx.x, C2 = 2 * Kc, tid = threadIdx.x;
    float* rm = c < Kc ? q_rm + c : k_rm + (c - Kc);
    float* rv = c < Kc ? q_rv + c : k_rv + (c - Kc);
    if (!training) {
        if (tid == 0) {
            save_mean[c] = *rm;
            save_invstd[c] = 1.0f / sqrtf(*rv + eps);
        }
        return;
    }
    double s1 = 0.0, s2 = 0.0;
    if ((P & 3) == 0) {
        const int P4 = P >> 2, total = B * P4;  // float4 slots of the channel over all images
        for (int i0 = tid; i0 < total; i0 += 4 * QS_T) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // four independent 16-byte loads in flight per thread
                const int i = min(i0 + u * QS_T, total - 1), b = i / P4, p4 = i - b * P4;
                v[u] = *reinterpret_cast<const f32x4*>(zqk + ((size_t)b * C2 + c) * P + (size_t)p4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * QS_T < total) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double d = (double)v[u][e];
                        s1 += d, s2 = fma(d, d, s2);
                    }
                }
        }
    } else {
        for (int b = 0; b < B; ++b) {
            const float* row = zqk + ((size_t)b * C2 + c) * P;
            for (int p = tid; p < P; p += QS_T) {
                const double d = (double)row[p];
                s1 += d, s2 = fma(d, d, s2);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((tid & 63) == 0) s_red[0][tid >> 6] = s1, s_red[1][tid >> 6] = s2;
    __syncthreads();
    if (tid == 0) {
        s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < QS_T / 64; ++w) s1 += s_red[0][w], s2 += s_red[1][w];
        const double count = (double)B * (double)P;
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        save_mean[c] = (float)mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        *rm = (float)((1.0 - (double)momentum) * (double)*rm + (double)momentum * mean);
        *rv = (float)((1.0 - (double)momentum) * (double)*rv + (double)momentum * unbiased);
    }
}

struct FwdWs {
    size_t stat, tk, tv, total;
};
static FwdWs fwd_layout(const QkvShape& s) {
    const PyrGeom g = make_geom(s);
    FwdWs w{};
    size_t off = 0;
    auto take = [&](size_t floats) {
        const size_t o = off;
        off += fbytes(floats);
        return o;
    };
    w.stat = take((size_t)2 * 2 * s.Kc * s.B * ceil_div(s.H * s.W, SG_TILE_P));  // (mean, M2) per (q | k channel, image, 64-position tile)
    w.tk = take((size_t)s.B * s.Kc * g.NBp);
    w.tv = take((size_t)s.B * s.Vc * g.NBp);
    w.total = off;
    return w;
}
size_t qkv_fwd_workspace(const QkvShape& s) { return fwd_layout(s).total; }
int qkv_padded_bins(const QkvShape& s) { return make_geom(s).NBp; }

static SgJob sg_job(const float* a, int lda, int a_mmajor, const float* b, int k, int b_rows, int M, int P, float* dst,
                    int dst_rows) {
    return sg_make(a, lda, a_mmajor, b, k, b_rows, M, P, dst, dst_rows);
}

// forward: 4 launches where cab_qkv_fused.hip applies (the model's shapes), else 5: [zq|zk|vv] GEMMs (one launch, weights read as stored), BN statistics + finalize,
// plane pass (BN + ReLU, q, kk, pooled bins), [T_k, T_v, W0 kk, W0 vv] GEMMs (one launch), pyramid add
hipError_t qkv_fwd_run(const QkvShape& s, const QkvParams& w, const float* x, int training, float momentum, float eps,
                       const QkvSaved& sv, float* q, float* k, float* v, void* ws, hipStream_t stream) {
    const PyrGeom g = make_geom(s);
    const int P = s.H * s.W, Mtot = 2 * s.Kc + s.Vc, ldk = (s.ns + 1) * s.Kc, ldv = (s.ns + 1) * s.Vc;
    const FwdWs L = fwd_layout(s);
    char* base = static_cast<char*>(ws);
    auto at = [&](size_t o) { return reinterpret_cast<float*>(base + o); };
    const bool fused = qkv_fused_fwd_supported(s);
    // round 5: BatchNorm statistics from the projection GEMM's epilogue, finalize in the plane pass's prologue (3 launches instead of
    // 4 where the output stage is fused too); CABINET_QKV_STATS_FUSED=0 keeps the round-4 launches (A/B timing)
    static const bool plane_wave_on = [] { const char* e = getenv("CABINET_QKV_STATS_FUSED"); return !(e && e[0] == '0'); }();
    const bool wplane = plane_wave_on && P < (1 << 21);
    const int tiles_p = ceil_div(P, SG_TILE_P);
    bool stats_in_gemm = false;
    {
        SgJobs jobs{};
        jobs.n = 3;
        jobs.j[0] = sg_job(w.wq, s.C, 1, x, s.C, s.C, s.Kc, P, sv.zqk, 2 * s.Kc);
        jobs.j[1] = sg_job(w.wk, s.C, 1, x, s.C, s.C, s.Kc, P, sv.zqk + (size_t)s.Kc * P, 2 * s.Kc);
        jobs.j[2] = sg_job(w.wv, s.C, 1, x, s.C, s.C, s.Vc, P, sv.vv, s.Vc);
        if (wplane && training) {
            jobs.j[0].stat = at(L.stat);
            jobs.j[1].stat = at(L.stat) + (size_t)s.Kc * s.B * tiles_p * 2;
        }
        stats_in_gemm = sg_gemm(jobs, s.B, stream) && wplane && training;
    }
    // (a channel-resident form of these two launches -- statistics + plane pass of one stacked channel over all B images in one
    // workgroup, as the backward uses -- measured 36 us against 5 + 19: 384 workgroups leave the pooling phases latency-bound)
    if (!stats_in_gemm)
        hipLaunchKernelGGL(qk_bn_stats_kernel, dim3(2 * s.Kc), dim3(QS_T), 0, stream, sv.zqk, s.B, s.Kc, P, training, momentum, eps,
                           w.bnq_rm, w.bnq_rv, w.bnk_rm, w.bnk_rv, sv.mean, sv.invstd);
    if (wplane) {
        PlaneWArgs pa{};
        pa.zqk = sv.zqk, pa.vv = sv.vv, pa.part = stats_in_gemm ? at(L.stat) : nullptr, pa.npairs = s.B * tiles_p;
        pa.momentum = momentum, pa.eps = eps;
        pa.q_rm = w.bnq_rm, pa.q_rv = w.bnq_rv, pa.k_rm = w.bnk_rm, pa.k_rv = w.bnk_rv, pa.save_mean = sv.mean, pa.save_invstd = sv.invstd;
        pa.bnq_w = w.bnq_w, pa.bnq_b = w.bnq_b, pa.bnk_w = w.bnk_w, pa.bnk_b = w.bnk_b;
        pa.B = s.B, pa.Kc = s.Kc, pa.Vc = s.Vc, pa.q = q, pa.kk = sv.kk, pa.pooled_k = sv.pooled_k, pa.pooled_v = sv.pooled_v;
        hipLaunchKernelGGL(qkv_plane_fwd_w_kernel, dim3(s.B * Mtot), dim3(256), lds_fwd_plane(g), stream, pa, g);
    } else {
        hipLaunchKernelGGL(qkv_plane_fwd_kernel, dim3(s.B * Mtot), dim3(256), lds_fwd_plane(g), stream, sv.zqk, sv.vv, sv.mean,
                           sv.invstd, w.bnq_w, w.bnq_b, w.bnk_w, w.bnk_b, s.Kc, s.Vc, g, q, sv.kk, sv.pooled_k, sv.pooled_v);
    }
    if (fused) {  // the output products with the pyramid terms formed per workgroup (one launch instead of two)
        if (hipError_t e = qkv_fused_out(s, w, sv, k, v, stream); e != hipSuccess) return e;
        return hipGetLastError();
    }
    {   // k, v = W_p[:, :Kc] . kk | vv (pyramid terms added below);  T_i = W_p[:, block i] . pooled_i on the bins of size i:
        // the block-expanded operand is zero outside its own bins, so each size is its own K = Kc product on a column
        // window of T (one K = ns*Kc product per branch had 32 workgroups walking 16 dependent chunks: 29 us)
        SgJobs jobs{};
        jobs.n = 2;
        jobs.j[0] = sg_job(w.wpk, ldk, 1, sv.kk, s.Kc, s.Kc, s.Kc, P, k, s.Kc);
        jobs.j[1] = sg_job(w.wpv, ldv, 1, sv.vv, s.Vc, s.Vc, s.Vc, P, v, s.Vc);
        for (int br = 0; br < 2; ++br)
            for (int i = 0; i < s.ns; ++i) {
                const int Kch = br ? s.Vc : s.Kc, ld = br ? ldv : ldk;
                const float* pooled = (br ? sv.pooled_v : sv.pooled_k) + (size_t)i * Kch * g.NBp + g.off[i];
                SgJob j = sg_job((br ? w.wpv : w.wpk) + (i + 1) * Kch, ld, 1, pooled, Kch, s.ns * Kch, Kch, g.s[i] * g.s[i],
                                 at(br ? L.tv : L.tk) + g.off[i], Kch);
                j.ldp = g.NBp;
                jobs.j[jobs.n++] = j;
            }
        sg_gemm(jobs, s.B, stream);
    }
    hipLaunchKernelGGL(pyramid_add_kernel, dim3(s.B * (s.Kc + s.Vc)), dim3(256),
                       (size_t)(g.NBp + 2 * g.ns * (g.H + g.W)) * sizeof(float), stream, at(L.tk), at(L.tv), s.Kc, s.Vc, g, k, v);
    return hipGetLastError();
}

struct BwdWs {
    size_t dtk, dtv, dpek, dpev, link, dzqk, dvv, bnpart, part, total;
};
static SdJob sd_job(const float* a, int a_rows, const float* x, int x_rows, int M, int N, int P, float* out, int ldo,
                    int col_off) {
    return sd_make(a, a_rows, x, x_rows, M, N, P, out, ldo, col_off);
}
static SdJobs qkv_dw_jobs(const QkvShape& s, const PyrGeom& g, const float* x, const QkvSaved& sv, const float* dk,
                          const float* dv, const float* dtk, const float* dtv, const float* dzqk, const float* dvv,
                          const QkvGrads& gr) {
    const int P = s.H * s.W, ldk = (s.ns + 1) * s.Kc, ldv = (s.ns + 1) * s.Vc;
    SdJobs jobs{};
    jobs.n = 6;
    jobs.j[0] = sd_job(dzqk, 2 * s.Kc, x, s.C, 2 * s.Kc, s.C, P, gr.dwqk, s.C, 0);             // [dW_q; dW_k]
    jobs.j[1] = sd_job(dvv, s.Vc, x, s.C, s.Vc, s.C, P, gr.dwv, s.C, 0);                       // dW_v
    jobs.j[2] = sd_job(dk, s.Kc, sv.kk, s.Kc, s.Kc, s.Kc, P, gr.dwpk, ldk, 0);                 // dW_p[:, :Kc] (key)
    jobs.j[3] = sd_job(dv, s.Vc, sv.vv, s.Vc, s.Vc, s.Vc, P, gr.dwpv, ldv, 0);                 // (value)
    jobs.j[4] = sd_job(dtk, s.Kc, sv.pooled_k, s.ns * s.Kc, s.Kc, s.ns * s.Kc, g.NBp, gr.dwpk, ldk, s.Kc);  // pyramid
    jobs.j[5] = sd_job(dtv, s.Vc, sv.pooled_v, s.ns * s.Vc, s.Vc, s.ns * s.Vc, g.NBp, gr.dwpv, ldv, s.Vc);
    return jobs;
}
static BwdWs bwd_layout(const QkvShape& s) {
    const PyrGeom g = make_geom(s);
    const int P = s.H * s.W;
    BwdWs w{};
    size_t off = 0;
    auto take = [&](size_t floats) {
        const size_t o = off;
        off += fbytes(floats);
        return o;
    };
    w.dtk = take((size_t)s.B * s.Kc * g.NBp);
    w.dtv = take((size_t)s.B * s.Vc * g.NBp);
    w.dpek = take((size_t)s.B * s.ns * s.Kc * g.NBp);
    w.dpev = take((size_t)s.B * s.ns * s.Vc * g.NBp);
    w.link = take((size_t)s.B * s.Kc * P);
    w.dzqk = take((size_t)s.B * 2 * s.Kc * P);
    w.dvv = take((size_t)s.B * s.Vc * P);
    w.bnpart = take((size_t)2 * 2 * s.Kc * s.B);
    QkvSaved sv{};
    QkvGrads gr{};
    SdJobs jobs = qkv_dw_jobs(s, g, nullptr, sv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gr);
    w.part = take(sd_plan(jobs, s.B));
    w.total = off;
    return w;
}
size_t qkv_bwd_workspace(const QkvShape& s) { return bwd_layout(s).total; }

// backward: 7 launches (was ~21): pyramid adjoint, [dpooled_k, dpooled_v, W0^T dk, W0^T dv] GEMMs (one launch), plane
// pass (pool adjoint, ReLU mask, BN partial sums), BN backward, dx GEMM (one job, three K-segments), all six weight
// gradients (one split launch + one ordered slab sum)
hipError_t qkv_bwd_run(const QkvShape& s, const QkvParams& w, const float* dq, const float* dk, const float* dv,
                       const float* x, int training, const QkvSaved& sv, const QkvGrads& gr, void* ws,
                       hipStream_t stream) {
    const PyrGeom g = make_geom(s);
    const int P = s.H * s.W, Mtot = 2 * s.Kc + s.Vc, ldk = (s.ns + 1) * s.Kc, ldv = (s.ns + 1) * s.Vc;
    const BwdWs L = bwd_layout(s);
    char* base = static_cast<char*>(ws);
    auto at = [&](size_t o) { return reinterpret_cast<float*>(base + o); };
    // pyramid terms: dT = U^T d
    hipLaunchKernelGGL(pyramid_adjoint_kernel, dim3(s.B * (s.Kc + s.Vc)), dim3(256), lds_adjoint(g), stream, dk, dv, s.Kc,
                       s.Vc, g, at(L.dtk), at(L.dtv));
    {   // dpooled[(i,c)][bin] = sum_m W_p[m][Kc + (i,c)] dT[m][bin];  identity branch: W_0^T d
        SgJobs jobs{};
        jobs.n = 4;
        jobs.j[0] = sg_job(w.wpk + s.Kc, ldk, 0, at(L.dtk), s.Kc, s.Kc, s.ns * s.Kc, g.NBp, at(L.dpek), s.ns * s.Kc);
        jobs.j[1] = sg_job(w.wpv + s.Vc, ldv, 0, at(L.dtv), s.Vc, s.Vc, s.ns * s.Vc, g.NBp, at(L.dpev), s.ns * s.Vc);
        jobs.j[2] = sg_job(w.wpk, ldk, 0, dk, s.Kc, s.Kc, s.Kc, P, at(L.link), s.Kc);
        jobs.j[3] = sg_job(w.wpv, ldv, 0, dv, s.Vc, s.Vc, s.Vc, P, at(L.dvv), s.Vc);
        sg_gemm(jobs, s.B, stream);
    }
    const size_t lds_ch = lds_channel_bwd(g, s.B);
    if (lds_ch <= 150 * 1024 && P < (1 << 21)) {  // plane pass + BatchNorm backward of a channel in ONE workgroup
        static lds_attr_mask attr_mask{0};
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(qkv_channel_bwd_kernel), 150 * 1024, attr_mask);
            e != hipSuccess)
            return e;
        hipLaunchKernelGGL(qkv_channel_bwd_kernel, dim3(Mtot), dim3(QC_T), lds_ch, stream, dq, at(L.link), sv.zqk, at(L.dpek),
                           at(L.dpev), sv.mean, sv.invstd, w.bnq_w, w.bnq_b, w.bnk_w, w.bnk_b, s.B, s.Kc, s.Vc, g, training,
                           at(L.dzqk), at(L.dvv), gr.dbnq_w, gr.dbnq_b, gr.dbnk_w, gr.dbnk_b);
    } else {
        hipLaunchKernelGGL(qkv_plane_bwd_kernel, dim3(s.B * Mtot), dim3(256), lds_bwd_plane(g), stream, dq, at(L.link), sv.zqk,
                           at(L.dpek), at(L.dpev), sv.mean, sv.invstd, w.bnq_w, w.bnq_b, w.bnk_w, w.bnk_b, s.B, s.Kc, s.Vc, g,
                           at(L.dzqk), at(L.dvv), at(L.bnpart));
        hipLaunchKernelGGL(qkv_bn_bwd_kernel, dim3(s.B * 2 * s.Kc), dim3(256), 0, stream, sv.zqk, sv.mean, sv.invstd, w.bnq_w,
                           w.bnk_w, at(L.bnpart), s.B, s.Kc, P, training, at(L.dzqk), gr.dbnq_w, gr.dbnq_b, gr.dbnk_w,
                           gr.dbnk_b);
    }
    if (qkv_dx_supported(s)) {  // both operands K-major as stored: an LDS-free kernel (cab_qkv_fused.hip)
        if (hipError_t e = qkv_dx_run(s, w, at(L.dzqk), at(L.dvv), gr.dx, stream); e != hipSuccess) return e;
    } else {  // dx = W_q^T dzq + W_k^T dzk + W_v^T dvv : the row-major weights ARE the K-major A operands
        SgJobs jobs{};
        jobs.n = 1;
        SgJob& j = jobs.j[0];
        j.seg[0] = {w.wq, at(L.dzqk), s.Kc, 2 * s.Kc};
        j.seg[1] = {w.wk, at(L.dzqk) + (size_t)s.Kc * P, s.Kc, 2 * s.Kc};
        j.seg[2] = {w.wv, at(L.dvv), s.Vc, s.Vc};
        j.nseg = 3, j.lda = s.C, j.a_mmajor = 0, j.M = s.C, j.P = P, j.dst = gr.dx, j.dst_rows = s.C;
        sg_gemm(jobs, s.B, stream);
    }
    SdJobs dw = qkv_dw_jobs(s, g, x, sv, dk, dv, at(L.dtk), at(L.dtv), at(L.dzqk), at(L.dvv), gr);
    return sd_run(dw, s.B, at(L.part), stream);
}

// ------------------------------------------------------------------------------- plain 1x1 convolution (no bias)
static int k16(int k) { return (k + 15) & ~15; }
// the small-tile path serves the CAB's grids (few thousand positions); big planes keep the 128-wide tiles of ffm.hip
static bool conv1x1_small(int B, int Ci, int Co, int P) {
    return (Ci % 4) == 0 && (Co % 4) == 0 && (long long)ceil_div(P, 128) * B * ceil_div(Co > Ci ? Co : Ci, 128) < 400;
}

size_t conv1x1_fwd_workspace(int Ci, int Co) { return fbytes((size_t)k16(Ci) * Co); }
size_t conv1x1_bwd_workspace(int B, int Ci, int Co, int P) {
    SdJobs jobs{};
    jobs.n = 1;
    jobs.j[0] = sd_job(nullptr, Co, nullptr, Ci, Co, Ci, P, nullptr, Ci, 0);
    const size_t small = fbytes(sd_plan(jobs, B));
    const size_t big = fbytes(dw_part_floats(B, Co, Ci, P)) + ((Co & 15) ? fbytes((size_t)k16(Co) * Ci) : 0);
    return small > big ? small : big;
}

bool conv1x1_bias_supported(int B, int Ci, int Co, int P) { return conv1x1_small(B, Ci, Co, P); }

hipError_t conv1x1_fwd_run(const float* x, const float* wgt, int B, int Ci, int Co, int P, float* y, void* ws,
                           hipStream_t stream, const float* bias) {
    if (conv1x1_small(B, Ci, Co, P)) {
        SgJobs jobs{};
        jobs.n = 1;
        jobs.j[0] = sg_job(wgt, Ci, 1, x, Ci, Ci, Co, P, y, Co);
        jobs.j[0].out_bias = bias;
        sg_gemm(jobs, B, stream);
        return hipGetLastError();
    }
    if (bias != nullptr) return hipErrorInvalidValue;   // the big-plane path has no bias term (capi.hip asks conv1x1_bias_supported first)
    float* wt = static_cast<float*>(ws);  // W^T, [align16(Ci)][Co], zero rows past Ci (K tail of the GEMM)
    if (Ci & 15) {
        hipError_t e = hipMemsetAsync(wt + (size_t)Ci * Co, 0, (size_t)(k16(Ci) - Ci) * Co * sizeof(float), stream);
        if (e != hipSuccess) return e;
    }
    TrJobs jobs{};
    jobs.j[0] = {wgt, wt, Ci, Co, Ci, Co, 0, 0};
    stage_weights(jobs, 1, stream);
    GemmKArgs a{};
    a.at = wt, a.lda = Co, a.M = Co, a.K = Ci;
    a.src0 = a.src1 = x, a.K0 = Ci;
    a.dst0 = a.dst1 = y, a.M0 = Co;
    a.P = P;
    if (hipError_t ge = gemm_kmajor(a, B, stream); ge != hipSuccess) return ge;
    return hipGetLastError();
}

hipError_t conv1x1_bwd_run(const float* dy, const float* x, const float* wgt, int B, int Ci, int Co, int P, float* dx,
                           float* dw, void* ws, hipStream_t stream) {
    float* part = static_cast<float*>(ws);
    if (conv1x1_small(B, Ci, Co, P)) {
        if (dx) {  // dx = W^T dy: the (Co x Ci) weight is the K-major A operand as stored
            SgJobs jobs{};
            jobs.n = 1;
            jobs.j[0] = sg_job(wgt, Ci, 0, dy, Co, Co, Ci, P, dx, Ci);
            sg_gemm(jobs, B, stream);
        }
        if (dw) {
            SdJobs jobs{};
            jobs.n = 1;
            jobs.j[0] = sd_job(dy, Co, x, Ci, Co, Ci, P, dw, Ci, 0);
            return sd_run(jobs, B, part, stream);
        }
        return hipGetLastError();
    }
    if (dx) {  // dx = W^T dy: the (Co x Ci) weight is already the K-major A operand (padded copy if Co % 16 != 0)
        const float* at = wgt;
        if (Co & 15) {
            float* wpad = reinterpret_cast<float*>(static_cast<char*>(ws) + fbytes(dw_part_floats(B, Co, Ci, P)));
            hipError_t e = hipMemsetAsync(wpad + (size_t)Co * Ci, 0, (size_t)(k16(Co) - Co) * Ci * sizeof(float), stream);
            if (e != hipSuccess) return e;
            TrJobs jobs{};
            jobs.j[0] = {wgt, wpad, Ci, Co, Ci, Ci, 0, 1};
            stage_weights(jobs, 1, stream);
            at = wpad;
        }
        GemmKArgs a{};
        a.at = at, a.lda = Ci, a.M = Ci, a.K = Co;
        a.src0 = a.src1 = dy, a.K0 = Co;
        a.dst0 = a.dst1 = dx, a.M0 = Ci;
        a.P = P;
        if (hipError_t ge = gemm_kmajor(a, B, stream); ge != hipSuccess) return ge;
    }
    if (dw) return dw_product(dy, x, B, Co, Ci, P, part, dw, Ci, 0, stream);
    return hipGetLastError();
}

}  // namespace cabinet
