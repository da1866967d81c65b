// K6 forward, output stage: k = W_0 kk + sum_i U_i(W_i A_i kk) and v likewise (reference src/models/cab.py:46-76 PSP module,
// cab.py:122-123, 141, 145; see cab_qkv.hip for the algebra) in ONE kernel.
//
//   qkv_psp_out_kernel   a workgroup (32 output channels x 256 positions of one image and branch) first forms the pyramid
//                        terms T = W_i . pooled_i of ITS channels (110 bins: six 32-bin products over the four sizes, one per
//                        wave), keeps them in LDS bin-major, runs the W_0 product and adds the four bilinear gathers in the
//                        epilogue (a lane is one position: its taps are computed once for its 16 channels, and four
//                        channels come with one 16-byte LDS read per tap).
// It replaces a 10-job small-GEMM launch (17 us) plus the pyramid-add launch (11 us) with one launch of 21.5 us.
//
// Operand feeding (the lesson of the K2 kernels): the weight blocks of a workgroup are staged ONCE into LDS row-major and read
// back as 16-byte quads (lane = output channel, four consecutive input channels: MFMA step e of a group of eight contracts
// channels (8g + e, 8g + 4 + e)); the activation operand has the position contiguous, so lane = position reads it straight
// from global memory with coalesced dword loads, requested before anything else, and never touches LDS.
//
// Measured and NOT kept: the projection [zq | zk | vv] = [W_q; W_k; W_v] x as a kernel of the same build with the BatchNorm
// partial statistics in its epilogue (and the finalize folded into the plane pass), which would have made the forward
// 3 launches: 27.7 us (MFMA loop 13.2, ramp + weight staging 6.9, stores 2.5, statistics 5.0 as 160 cross-lane shuffles per
// wave; 32 us with a 34-shuffle halving butterfly) + 5 us more in the plane pass, against 20 + 5 + a launch boundary for the
// small-GEMM launch and the statistics kernel it would replace.  One more finding of that experiment is kept as a rule:
// an empty `asm volatile("" : "+v"(acc))` that pins an MFMA in its issue slot must name the register class the compiler
// keeps accumulators in -- "+a" in a 256-thread kernel; "+v" there cost 16 v_accvgpr_write + 16 v_accvgpr_read + s_nop 15
// around EVERY MFMA (45 us).
#include <type_traits>

#include "cab_qkv.hpp"
#include "common.hpp"

namespace cabinet {

namespace {

struct PspArgs {
    const float* wp[2];      // W_p of the key / value branch: (Kch, (ns + 1) * Kch) row-major
    const float* src[2];     // kk (B,Kc,P), vv (B,Vc,P)
    const float* pooled[2];  // block-expanded pooled bins (B, ns * Kch, NBp)
    float* out[2];           // k (B,Kc,P), v (B,Vc,P)
    int kch[2];
    int B, P, H, W, ns, NBp;
    int s[4], off[4];
};

// workgroup = 32 output channels x 256 positions of one image and branch, 8 waves (one 32-position block each)
template <int KCH>
__global__ __launch_bounds__(512) void qkv_psp_out_kernel(PspArgs a) {
    constexpr int RS = KCH + 4, NG = KCH / 8, TS = 36;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W0 = sm;                      // [32][RS]            W_p[:, 0:Kch) rows of this block
    float* Wi = W0 + 32 * RS;            // [ns][32][RS]        W_p[:, (i+1) Kch : (i+2) Kch)
    float* T = Wi + a.ns * 32 * RS;      // [NBp][TS]           pyramid terms, bin-major: the epilogue reads four channels per 16-byte read
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int PT = a.P / 256, mb0 = a.kch[0] / 32, MBT = mb0 + a.kch[1] / 32;
    const int t = xcd_chunked_tile(blockIdx.x, MBT * PT * a.B);
    const int mblk_all = t % MBT, pt = t / MBT, b = pt / PT, p0 = (pt - b * PT) * 256 + wave * 32;
    const int br = mblk_all >= mb0, m0 = (br ? mblk_all - mb0 : mblk_all) * 32;
    const int P = a.P, NBp = a.NBp, ld = (a.ns + 1) * KCH, row_bytes = P * 4;
    const float* wp = a.wp[br] + (size_t)m0 * ld;

    // the position operand of the W_0 product: all KCH / 2 values of this lane, requested before anything else
    const buf_rsrc s_rs = make_rsrc(a.src[br] + (size_t)b * KCH * P, (unsigned)KCH * row_bytes);
    const int voff = (4 * h * P + p0 + li) * 4;
    f32x4 Bv[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) Bv[g][e] = bload(s_rs, voff, (8 * g + e) * row_bytes);

    // ---- pyramid terms: job = (size i, block of 32 bins), T[m][off_i + n] = sum_c W_i[m][c] pooled_i[c][n]; at most 8 jobs
    //      (sizes^2 <= 64), so every wave has at most ONE: its operand is requested here, before the weights are staged ----
    int ji = -1, jn0 = 0;
    {
        int j = 0;
        for (int i = 0; i < a.ns; ++i)
            for (int n0 = 0; n0 < a.s[i] * a.s[i]; n0 += 32, ++j)
                if (j == wave) ji = i, jn0 = n0;
    }
    const bool has_job = ji >= 0;
    const int jsz = has_job ? a.s[ji] * a.s[ji] : 0, joff = has_job ? a.off[ji] + jn0 : 0;
    const bool jvalid = jn0 + li < jsz;
    // rows (i, c) of the block-expanded operand hold the bins of size i at columns off_i ..
    const buf_rsrc p_rs = make_rsrc(a.pooled[br], (unsigned)((size_t)a.B * a.ns * KCH * NBp * 4));
    const int pvoff = ((((b * a.ns + (has_job ? ji : 0)) * KCH + 4 * h) * NBp) + joff + li) * 4;  // one address register
    constexpr bool EARLY = NG <= 16;  // KCH = 256: 128 more registers would not fit, the operand is fetched in chunks later
    f32x4 bv[EARLY ? NG : 1];
    if (EARLY && has_job) {
#pragma unroll
        for (int g = 0; g < (EARLY ? NG : 1); ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[g][e] = jvalid ? bload(p_rs, pvoff, (8 * g + e) * NBp * 4) : 0.f;
    }

    {   // stage the (ns + 1) weight blocks: rows of ld floats, KCH of them per block
        const int q4 = KCH / 4, per_row = (a.ns + 1) * q4;
        for (int i = tid; i < 32 * per_row; i += 512) {
            const int r = i / per_row, rem = i - r * per_row, blk = rem / q4, c4 = rem - blk * q4;
            *reinterpret_cast<f32x4*>(sm + (blk * 32 + r) * RS + 4 * c4) =
                *reinterpret_cast<const f32x4*>(wp + (size_t)r * ld + blk * KCH + 4 * c4);
        }
    }
    __syncthreads();

    if (has_job) {
        const float* wa = Wi + (ji * 32 + li) * RS + 4 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (EARLY) {
            f32x4 av[NG];  // all weight quads first: one LDS round trip for the chain instead of one per group
#pragma unroll
            for (int g = 0; g < NG; ++g) av[g] = *reinterpret_cast<const f32x4*>(wa + 8 * g);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(av[g][e], bv[g][e], acc);
        } else {
            constexpr int GU = 8;
            for (int g0 = 0; g0 < NG; g0 += GU) {
                f32x4 cv[GU];
#pragma unroll
                for (int g = 0; g < GU; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cv[g][e] = jvalid ? bload(p_rs, pvoff, (8 * (g0 + g) + e) * NBp * 4) : 0.f;
#pragma unroll
                for (int g = 0; g < GU; ++g) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(wa + 8 * (g0 + g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma32(av[e], cv[g][e], acc);
                }
            }
        }
        if (jvalid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(joff + li) * TS + acc_row(r) + 4 * h] = acc[r];
        }
    }

    // ---- W_0 product: 32 channels x this wave's 32 positions ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const float* wa = W0 + li * RS + 4 * h;
        f32x4 av = *reinterpret_cast<const f32x4*>(wa);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 cur = av;
            if (g + 1 < NG) av = *reinterpret_cast<const f32x4*>(wa + 8 * (g + 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(cur[e], Bv[g][e], acc);
        }
    }
    __syncthreads();  // T complete

    // ---- epilogue: + the four bilinear gathers; a lane is ONE position, so its taps are computed once for 16 channels ----
    const int p = p0 + li, oy = p / a.W, ox = p - oy * a.W;
    float* dst = a.out[br] + ((size_t)b * KCH + m0 + 4 * h) * P + p;
    const float* Tr = T + 4 * h;
    for (int i = 0; i < a.ns; ++i) {
        const int sz = a.s[i];
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_taps(oy, (float)sz / (float)a.H, sz, y0, y1, ly);
        bilinear_taps(ox, (float)sz / (float)a.W, sz, x0, x1, lx);
        const float* t00 = Tr + (a.off[i] + y0 * sz + x0) * TS;
        const float* t01 = Tr + (a.off[i] + y0 * sz + x1) * TS;
        const float* t10 = Tr + (a.off[i] + y1 * sz + x0) * TS;
        const float* t11 = Tr + (a.off[i] + y1 * sz + x1) * TS;
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {  // registers 4 rq .. 4 rq + 3 are channels 8 rq + 4 h + (0..3): one quad per tap
            const f32x4 q00 = *reinterpret_cast<const f32x4*>(t00 + 8 * rq), q01 = *reinterpret_cast<const f32x4*>(t01 + 8 * rq);
            const f32x4 q10 = *reinterpret_cast<const f32x4*>(t10 + 8 * rq), q11 = *reinterpret_cast<const f32x4*>(t11 + 8 * rq);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * rq + e] += (w00 * q00[e] + w01 * q01[e]) + (w10 * q10[e] + w11 * q11[e]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)acc_row(r) * P] = acc[r];
}

// ---------------------------------------------------------------------------------------------------------------------
// dx = W_q^T dzq + W_k^T dzk + W_v^T dvv  (backward of the three projections, cab.py:107-121): D[c][p] = sum_m W[m][c] dz[m][p].
// Both operands are "K-major" as stored -- the weight rows have the input channel c contiguous, the gradient rows the position
// p -- so lane = c reads A and lane = p reads B straight from global memory with coalesced dword loads: no LDS, no staging
// kernel, no barrier.  Workgroup = 32 input channels x 256 positions of one image, 8 waves (one 32-position block each);
// the contraction runs over the 2 Kc + Vc stacked output channels in batches of 32 (16 MFMAs), double-buffered in registers
// (the K2 batch scheme).  Replaces a one-job, three-segment small-GEMM launch: 24 -> ~12 us at config 3.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// an MFMA has no side effect: pinned in its slot ("+v": a 512-thread kernel keeps accumulators in the VGPR half)
__device__ __forceinline__ void pin(f32x16& acc) { asm volatile("" : "+v"(acc)); }

struct DxArgs {
    const float* w[3];   // W_q (Kc,C), W_k (Kc,C), W_v (Vc,C) row-major
    const float* dz[3];  // dzq = dzqk rows [0,Kc), dzk = dzqk rows [Kc,2Kc) (image stride 2Kc*P), dvv (image stride Vc*P)
    int rows[3];         // Kc, Kc, Vc
    int img_rows[3];     // rows per image of the tensor each dz pointer walks: 2Kc, 2Kc, Vc
    float* dx;           // (B,C,P)
    int B, C, P;
};

__global__ __launch_bounds__(512) void qkv_dx_kernel(DxArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int C = a.C, P = a.P, CB = C / 32, PT = P / 256;
    const int t = xcd_chunked_tile(blockIdx.x, CB * PT * a.B);
    const int cblk = t % CB, pt = t / CB, b = pt / PT, p0 = (pt - b * PT) * 256 + wave * 32, c0 = cblk * 32;
    const int a_voff = (h * C + c0 + li) * 4, b_voff = (h * P + p0 + li) * 4;
    buf_rsrc w_rs[3], z_rs[3];
    int nb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        w_rs[s] = make_rsrc(a.w[s], (unsigned)a.rows[s] * C * 4);
        z_rs[s] = make_rsrc(a.dz[s] + (size_t)b * a.img_rows[s] * P, (unsigned)a.rows[s] * P * 4);
        nb[s] = a.rows[s] / 32;
    }
    const int nbat = nb[0] + nb[1] + nb[2];
    f32x4 A[2][4], Bq[2][4];
    // element U (m = 32 * local batch + 2U + h) of global batch bt; the segment is wave-uniform
    auto load = [&](auto p_tag, auto u_tag, int bt) {
        constexpr int PB = decltype(p_tag)::value, U = decltype(u_tag)::value;
        const int s = bt < nb[0] ? 0 : (bt < nb[0] + nb[1] ? 1 : 2), lb = bt - (s > 0 ? nb[0] : 0) - (s > 1 ? nb[1] : 0);
        const int m = 32 * lb + 2 * U;
        A[PB][U >> 2][U & 3] = bload(w_rs[s], a_voff, m * C * 4);
        Bq[PB][U >> 2][U & 3] = bload(z_rs[s], b_voff, m * P * 4);
    };
    static_for<0, 16>([&](auto u) { load(std::integral_constant<int, 0>{}, u, 0); });
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto batch = [&](auto p_tag, int bt) {
        constexpr int PB = decltype(p_tag)::value;
        const int nxt = min(bt + 1, nbat - 1);  // unconditional prefetch (a branch around a load drains vmcnt at the join)
        static_for<0, 16>([&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            acc = mfma32(A[PB][U >> 2][U & 3], Bq[PB][U >> 2][U & 3], acc);
            pin(acc);
            load(std::integral_constant<int, 1 - PB>{}, u_tag, nxt);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int bt = 0; bt < nbat; bt += 2) {
        batch(std::integral_constant<int, 0>{}, bt);
        if (bt + 1 < nbat) batch(std::integral_constant<int, 1>{}, bt + 1);
    }
    float* dst = a.dx + ((size_t)b * C + c0 + 4 * h) * P + p0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)acc_row(r) * P] = acc[r];
}

size_t psp_lds(const QkvShape& s, int kch, int NBp) { return ((size_t)(s.ns + 1) * 32 * (kch + 4) + (size_t)36 * NBp) * sizeof(float); }

}  // namespace

bool qkv_fused_fwd_supported(const QkvShape& s) {
    const int P = s.H * s.W;
    if ((s.Kc % 32) || (s.Vc % 32) || (P % 256)) return false;
    if (s.Kc != s.Vc || (s.Kc != 64 && s.Kc != 128 && s.Kc != 256)) return false;  // instantiated branch widths
    for (int i = 0; i < s.ns; ++i)
        if (s.sizes[i] * s.sizes[i] > 64) return false;  // at most two 32-bin blocks per size (8 jobs)
    return psp_lds(s, s.Kc, qkv_padded_bins(s)) <= 150 * 1024;
}

template <int KCH>
static hipError_t launch_psp(const PspArgs& a, const QkvShape& s, hipStream_t stream) {
    auto fn = qkv_psp_out_kernel<KCH>;
    const size_t lds = psp_lds(s, KCH, a.NBp);
    static lds_attr_mask mask{0};
    // the size depends on the pyramid: the attribute is set once per device, to the most the kernel can ever ask for
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(fn), 160 * 1024, mask); e != hipSuccess) return e;
    const int blocks = ((a.kch[0] + a.kch[1]) / 32) * (a.P / 256) * a.B;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds, stream, a);
    return hipSuccess;
}

bool qkv_dx_supported(const QkvShape& s) {
    const int P = s.H * s.W;
    return (s.C % 32) == 0 && (s.Kc % 32) == 0 && (s.Vc % 32) == 0 && (P % 256) == 0 && (size_t)s.B * (2 * s.Kc + s.Vc) * P < (1u << 29);
}

hipError_t qkv_dx_run(const QkvShape& s, const QkvParams& w, const float* dzqk, const float* dvv, float* dx, hipStream_t stream) {
    const int P = s.H * s.W;
    DxArgs a{};
    a.w[0] = w.wq, a.w[1] = w.wk, a.w[2] = w.wv;
    a.dz[0] = dzqk, a.dz[1] = dzqk + (size_t)s.Kc * P, a.dz[2] = dvv;
    a.rows[0] = s.Kc, a.rows[1] = s.Kc, a.rows[2] = s.Vc;
    a.img_rows[0] = 2 * s.Kc, a.img_rows[1] = 2 * s.Kc, a.img_rows[2] = s.Vc;
    a.dx = dx, a.B = s.B, a.C = s.C, a.P = P;
    hipLaunchKernelGGL(qkv_dx_kernel, dim3((s.C / 32) * (P / 256) * s.B), dim3(512), 0, stream, a);
    return hipGetLastError();
}

hipError_t qkv_fused_out(const QkvShape& s, const QkvParams& w, const QkvSaved& sv, float* k, float* v, hipStream_t stream) {
    PspArgs a{};
    a.wp[0] = w.wpk, a.wp[1] = w.wpv, a.src[0] = sv.kk, a.src[1] = sv.vv, a.pooled[0] = sv.pooled_k, a.pooled[1] = sv.pooled_v;
    a.out[0] = k, a.out[1] = v, a.kch[0] = s.Kc, a.kch[1] = s.Vc;
    a.B = s.B, a.P = s.H * s.W, a.H = s.H, a.W = s.W, a.ns = s.ns, a.NBp = qkv_padded_bins(s);
    int nb = 0;
    for (int i = 0; i < s.ns; ++i) a.s[i] = s.sizes[i], a.off[i] = nb, nb += s.sizes[i] * s.sizes[i];
    switch (s.Kc) {
        case 64: return launch_psp<64>(a, s, stream);
        case 128: return launch_psp<128>(a, s, stream);
        case 256: return launch_psp<256>(a, s, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace cabinet
