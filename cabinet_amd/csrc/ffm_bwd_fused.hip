// K4 (round 4) -- the two big products of the FFM backward from ONE staged tile of dz.
//
// Reference span: the autograd backward of the 1x1 convolution of src/models/cabinet.py:143-144 (ConvBNReLU over
// cat(fsp, fcp), with the x4 bilinear resize of cabinet.py:228-230 commuted behind the convolution):
//     dX[c][p]  = sum_o W[o][c] dz[o][p]          (input gradient:  dfsp at full resolution, dlow at low resolution)
//     dW[o][c] += sum_p dz[o][p] X[c][p]          (weight gradient: the fsp columns, the two halves of the low columns)
// Round 3 ran them as five launches that each re-read their dz operand from HBM: gemm_kmajor (dfsp, 87 us), gemm_dw + slab
// sum (dW_s, 88 + 10 us), and three small-grid launches for the low-resolution side (24 + 20 + 5 us, latency-bound on 8192
// positions).  Here ONE persistent kernel walks a list of 32-pixel chunks; a chunk of dz (256 x 32) and of X (128 x 32) is
// staged in LDS once and feeds BOTH products:
//   * dW: wave w owns the 32 output rows o = 32 w .. 32 w + 31 and all 128 columns c; its 256 x 128 tile of dW lives in 64
//     accumulator registers for the whole run of chunks of one "segment" (a segment = one (dz, X, W column block) triple: the
//     full-resolution fsp side, then the two 128-channel halves of the low-resolution side); the contraction index is the pixel;
//   * dX: wave w owns output channels 32 (w & 3) .. + 31 and HALF of the contraction (o in 128 (w >> 2) .. + 127); its A
//     operand W[o][c] (64 values per lane) is loaded once per segment and stays in registers; the two halves are added
//     through LDS one chunk later (double-buffered: no extra barrier).
// Both operands of the pixel contraction need "row on the lane, pixel along the instruction's k": the staged rows are
// stored DE-INTERLEAVED (even pixels, then odd pixels), so the 16 values a lane needs for the 16 k-steps of a chunk are 16
// consecutive floats = four 16-byte LDS reads (row pitch 36 floats: conflict-free for ds_read_b128).  The dX product reads
// the same rows with the pixel slot on the lane (consecutive words: conflict-free); its output columns are the slots, mapped
// back to pixels in the store.
// Work is dealt in equal contiguous runs of chunks over <= 256 workgroups (one per CU); a workgroup writes its dW tile as
// one slab per segment it touched, and an ordered slab sum finishes dW: no atomics, bit-reproducible.
// Exact fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout.
#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int XW_CO = 256;     // output channels of the 1x1 convolution (rows of dz and of W)
constexpr int XW_CX = 128;     // input channels per segment (columns of W / rows of X handled at a time)
constexpr int XW_PX = 32;      // pixels per chunk
constexpr int XW_PITCH = 36;   // LDS row pitch in floats (32 + 4: 16-byte aligned rows, conflict-free ds_read_b128)
constexpr int XW_MAXSEG = 4;

struct XwSeg {
    const float* dz;   // (B, XW_CO, P)
    const float* x;    // (B, x_rows, P): channels [x_c0, x_c0 + 128) are this segment's
    float* dx;         // (B, x_rows, P): gradient of the same channels
    int P, x_rows, x_c0, w_c0;
    int chunks_per_img, unit_lo;   // unit_lo: index of the segment's first chunk in the global chunk list
};
struct XwArgs {
    XwSeg seg[XW_MAXSEG];
    int nseg, total_units, units_per_wg;
    const float* w;    // (XW_CO, ldw) row-major weight
    int ldw;
    float* slabs;      // [(workgroup + segment)][XW_CO][XW_CX] partial dW tiles
};

__device__ __forceinline__ int xw_slot_to_px(int slot) { return slot < 16 ? 2 * slot : 2 * (slot - 16) + 1; }

__global__ __launch_bounds__(512) void ffm_bwd_xw_kernel(XwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzs = smem;                                   // [2][256][36]
    float* xs = dzs + 2 * XW_CO * XW_PITCH;              // [2][128][36]
    float* red = xs + 2 * XW_CX * XW_PITCH;              // [2][4 waves][16 regs][64 lanes]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int ci = wave & 3, kh = wave >> 2;             // dX: channel block, contraction half
    const int u_lo = blockIdx.x * a.units_per_wg, u_hi = min(u_lo + a.units_per_wg, a.total_units);
    if (u_lo >= u_hi) return;

    // staging assignment: dz chunk = 256 rows x 32 px, two threads per row (16 px each); X chunk = 128 rows, four threads per row
    const int zr = tid >> 1, zh = tid & 1, xr = tid >> 2, xq = tid & 3;
    f32x4 rz[4], rx[2];
    auto seg_of = [&](int u) {
        int k = 0;
        while (k + 1 < a.nseg && u >= a.seg[k + 1].unit_lo) ++k;
        return k;
    };
    auto load_chunk = [&](int u) {
        const int k = seg_of(u);
        const XwSeg& s = a.seg[k];
        const int cs = u - s.unit_lo, b = cs / s.chunks_per_img, p0 = (cs - b * s.chunks_per_img) * XW_PX;
        const float* zp = s.dz + ((size_t)b * XW_CO + zr) * s.P + p0 + zh * 16;
        const float* xp = s.x + ((size_t)b * s.x_rows + s.x_c0 + xr) * s.P + p0 + xq * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) rz[q] = *reinterpret_cast<const f32x4*>(zp + 4 * q);
#pragma unroll
        for (int q = 0; q < 2; ++q) rx[q] = *reinterpret_cast<const f32x4*>(xp + 4 * q);
    };
    auto store_chunk = [&](int buf) {
        // de-interleave: pixel p of the row -> slot (p & 1) * 16 + (p >> 1)
        float* zd = dzs + ((size_t)buf * XW_CO + zr) * XW_PITCH + zh * 8;
        *reinterpret_cast<f32x4*>(zd) = f32x4{rz[0][0], rz[0][2], rz[1][0], rz[1][2]};
        *reinterpret_cast<f32x4*>(zd + 4) = f32x4{rz[2][0], rz[2][2], rz[3][0], rz[3][2]};
        *reinterpret_cast<f32x4*>(zd + 16) = f32x4{rz[0][1], rz[0][3], rz[1][1], rz[1][3]};
        *reinterpret_cast<f32x4*>(zd + 20) = f32x4{rz[2][1], rz[2][3], rz[3][1], rz[3][3]};
        float* xd = xs + ((size_t)buf * XW_CX + xr) * XW_PITCH + xq * 4;
        *reinterpret_cast<f32x4*>(xd) = f32x4{rx[0][0], rx[0][2], rx[1][0], rx[1][2]};
        *reinterpret_cast<f32x4*>(xd + 16) = f32x4{rx[0][1], rx[0][3], rx[1][1], rx[1][3]};
    };

    f32x16 dw[4];      // dW tile rows 32 wave .. +31, columns 32 j .. +31
    f32x16 dxa;        // dX partial: channels 32 ci .. +31, the chunk's 32 pixel slots, contraction half kh
    float wf[64];      // W[128 kh + 2 s + h][w_c0 + 32 ci + li], s = 0 .. 63
    auto zero_dw = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[j][r] = 0.f;
    };
    auto load_w = [&](int k) {
        const float* wp = a.w + (size_t)(128 * kh + h) * a.ldw + a.seg[k].w_c0 + 32 * ci + li;
#pragma unroll
        for (int s = 0; s < 64; ++s) wf[s] = wp[(size_t)(2 * s) * a.ldw];
    };
    auto flush_dw = [&](int k) {
        float* slab = a.slabs + (size_t)(blockIdx.x + k) * XW_CO * XW_CX;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(size_t)(32 * wave + acc_row(r) + 4 * h) * XW_CX + 32 * j + li] = dw[j][r];
    };
    // finish the dX tile of a chunk: waves 0-3 add the upper half's partial (LDS) to their own and store the rows
    auto store_dx = [&](int u, int rbuf) {
        const int k = seg_of(u);
        const XwSeg& s = a.seg[k];
        const int cs = u - s.unit_lo, b = cs / s.chunks_per_img, p0 = (cs - b * s.chunks_per_img) * XW_PX;
        const float* rp = red + ((size_t)(rbuf * 4 + ci) * 16) * 64 + lane;
        float* dp = s.dx + ((size_t)b * s.x_rows + s.x_c0 + 32 * ci + 4 * h) * s.P + p0 + xw_slot_to_px(li);
        // WRITE-THROUGH stores (sc1: the line is not kept dirty in the XCD's L2).  With plain or non-temporal stores this
        // kernel took 228 us instead of 188: every 128-byte row piece stays in L2 as a dirty line, and the streaming reads of dz
        // and X (268 MB through a 4 MB L2) then pay a write-back on eviction IN THEIR OWN MISS PATH -- the loads of the next
        // chunk came back later than one whole chunk of MFMAs (measured with the stand-alone probe: chunk loads served from
        // cache 190 us, no dX stores 192 us, both as they are 241 us, sc1 stores 201 us; all + 13 us slab sum).
#pragma unroll
        for (int r = 0; r < 16; ++r)
            store_wt(dp + (size_t)acc_row(r) * s.P, dxa[r] + rp[r * 64]);
    };

    int k_cur = seg_of(u_lo);
    zero_dw();
    load_w(k_cur);
    load_chunk(u_lo);
    store_chunk(0);
    __syncthreads();
    for (int u = u_lo; u < u_hi; ++u) {
        const int buf = (u - u_lo) & 1;
        if (u + 1 < u_hi) load_chunk(u + 1);
        if (u > u_lo && kh == 0) store_dx(u - 1, buf ^ 1);   // the previous chunk's tile: its upper half landed before the barrier
        const int k = seg_of(u);
        if (k != k_cur) {   // wave-uniform: a new segment starts with this chunk
            flush_dw(k_cur);
            zero_dw();
            load_w(k);
            k_cur = k;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[r] = 0.f;
        // Eight blocks of 16 MFMAs per chunk: blocks 0-3 = dW column blocks j (operands: 16 consecutive floats of the wave's dz
        // row and of X row 32 j + li), blocks 4-7 = quarters of the dX half-contraction (operand: one float per k-step from the
        // dz rows, pixel slot on the lane).  The operands of block n + 1 are requested BEFORE the MFMAs of block n are issued
        // and the order is pinned with scheduling barriers: left to itself the compiler emitted  read -> wait -> 2 MFMAs  for
        // the dX part (every LDS round trip exposed, both waves of a SIMD stalling in lockstep) and the fused kernel was no
        // faster than the five launches it replaces.
        const float* zrow = dzs + ((size_t)buf * XW_CO + 32 * wave + li) * XW_PITCH + 16 * h;
        const float* xrow = xs + ((size_t)buf * XW_CX + li) * XW_PITCH + 16 * h;
        const float* zcol = dzs + ((size_t)buf * XW_CO + 128 * kh + h) * XW_PITCH + li;
        float af[16], bf[2][16];
        auto read16 = [&](const float* p, float* d) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
                d[4 * q] = t[0], d[4 * q + 1] = t[1], d[4 * q + 2] = t[2], d[4 * q + 3] = t[3];
            }
        };
        auto read_col = [&](int blk, float* d) {   // k-steps 16 blk .. 16 blk + 15 of the dX contraction
#pragma unroll
            for (int s = 0; s < 16; ++s) d[s] = zcol[(size_t)(2 * (16 * blk + s)) * XW_PITCH];
        };
        read16(zrow, af);
        read16(xrow, bf[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3)
                read16(xrow + (size_t)32 * (j + 1) * XW_PITCH, bf[(j + 1) & 1]);
            else
                read_col(0, bf[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) dw[j] = mfma32(af[s], bf[j & 1][s], dw[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (blk < 3) read_col(blk + 1, bf[(blk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) dxa = mfma32(wf[16 * blk + s], bf[blk & 1][s], dxa);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kh == 1) {   // upper contraction half: hand the partial tile to the lower half's wave
            float* rp = red + ((size_t)(buf * 4 + ci) * 16) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) rp[r * 64] = dxa[r];
        }
        if (u + 1 < u_hi) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (kh == 0) store_dx(u_hi - 1, (u_hi - 1 - u_lo) & 1);
    flush_dw(k_cur);
}

// out[o * ldo + col_off[k] + c] = sum over the workgroups w of segment k's run, ascending, of slab[w + k][o][c]
struct XwSumArgs {
    const float* slabs;
    float* out;
    int ldo, nseg;
    int col_off[XW_MAXSEG], wg_lo[XW_MAXSEG], wg_hi[XW_MAXSEG];
};
__global__ __launch_bounds__(256) void ffm_bwd_xw_sum_kernel(XwSumArgs a) {
    const int k = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;   // i over XW_CO * XW_CX
    if (k >= a.nseg || i >= XW_CO * XW_CX) return;
    const size_t stride = (size_t)XW_CO * XW_CX;
    const float* p = a.slabs + (size_t)(a.wg_lo[k] + k) * stride + i;
    const int n = a.wg_hi[k] - a.wg_lo[k] + 1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int w = 0;
    for (; w + 7 < n; w += 8) {
        s0 += p[(size_t)w * stride], s1 += p[(size_t)(w + 1) * stride], s2 += p[(size_t)(w + 2) * stride];
        s3 += p[(size_t)(w + 3) * stride], s4 += p[(size_t)(w + 4) * stride], s5 += p[(size_t)(w + 5) * stride];
        s6 += p[(size_t)(w + 6) * stride], s7 += p[(size_t)(w + 7) * stride];
    }
    for (; w < n; ++w) s0 += p[(size_t)w * stride];
    a.out[(size_t)(i / XW_CX) * a.ldo + a.col_off[k] + (i % XW_CX)] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int xw_units(int B, int P, int Pl, int Cc) { return B * (P / XW_PX) + (Cc / XW_CX) * B * (Pl / XW_PX); }
static int xw_units_per_wg(int total) { return ceil_div(total, 256); }

bool ffm_bwd_fused_supported(int B, int Cs, int Cc, int Co, int P, int Pl) {
    return Co == XW_CO && Cs == XW_CX && Cc > 0 && (Cc % XW_CX) == 0 && Cc / XW_CX + 1 <= XW_MAXSEG && (P % XW_PX) == 0 &&
           (Pl % XW_PX) == 0 && B > 0 && xw_units_per_wg(xw_units(B, P, Pl, Cc)) <= B * (Pl / XW_PX);
}

size_t ffm_bwd_fused_slab_floats(int B, int Cs, int Cc, int P, int Pl) {
    (void)Cs;
    const int total = xw_units(B, P, Pl, Cc), nwg = ceil_div(total, xw_units_per_wg(total));
    return (size_t)(nwg + XW_MAXSEG) * XW_CO * XW_CX;
}

// dfsp = W_s^T dz, dlow = W_c^T dzl, dW = [dz fsp^T | dzl low^T]   (w: (Co, Cs + Cc) row-major, dw likewise)
hipError_t ffm_bwd_fused_run(const float* dz, const float* dzl, const float* fsp, const float* low, const float* w, int B,
                             int Cs, int Cc, int P, int Pl, float* dfsp, float* dlow, float* dw, float* slabs,
                             hipStream_t stream) {
    XwArgs a{};
    XwSumArgs sa{};
    const int ldw = Cs + Cc;
    int unit = 0, k = 0;
    a.seg[k] = XwSeg{dz, fsp, dfsp, P, Cs, 0, 0, P / XW_PX, unit};
    sa.col_off[k] = 0;
    unit += B * (P / XW_PX);
    for (int c0 = 0; c0 < Cc; c0 += XW_CX) {
        ++k;
        a.seg[k] = XwSeg{dzl, low, dlow, Pl, Cc, c0, Cs + c0, Pl / XW_PX, unit};
        sa.col_off[k] = Cs + c0;
        unit += B * (Pl / XW_PX);
    }
    a.nseg = sa.nseg = k + 1;
    a.total_units = unit;
    a.units_per_wg = xw_units_per_wg(unit);
    a.w = w, a.ldw = ldw, a.slabs = slabs;
    const int nwg = ceil_div(unit, a.units_per_wg);
    for (int s = 0; s < a.nseg; ++s) {
        const int lo = a.seg[s].unit_lo, hi = (s + 1 < a.nseg ? a.seg[s + 1].unit_lo : unit) - 1;
        sa.wg_lo[s] = lo / a.units_per_wg, sa.wg_hi[s] = hi / a.units_per_wg;
    }
    sa.slabs = slabs, sa.out = dw, sa.ldo = ldw;
    const size_t lds = ((size_t)2 * (XW_CO + XW_CX) * XW_PITCH + 2 * 4 * 16 * 64) * sizeof(float);
    static lds_attr_mask mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ffm_bwd_xw_kernel), lds, mask); e != hipSuccess) return e;
    hipLaunchKernelGGL(ffm_bwd_xw_kernel, dim3(nwg), dim3(512), lds, stream, a);
    hipLaunchKernelGGL(ffm_bwd_xw_sum_kernel, dim3(XW_CO * XW_CX / 256, a.nseg), dim3(256), 0, stream, sa);
    return hipGetLastError();
}

}  // namespace cabinet
