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
// Both operands of the pixel contraction need "row on the lane, pixel along the instruction's k".  Which two pixels an MFMA
// step contracts is free as long as A and B agree, so k-step s takes pixel s from the lanes 0-31 and pixel s + 16 from the
// lanes 32-63: the 16 values a lane needs for the 16 k-steps of a chunk are then 16 CONSECUTIVE floats of its row in plain
// pixel order = four 16-byte LDS reads (row pitch 36 floats: conflict-free for ds_read_b128), and the staged rows need no
// re-ordering at all.  The dX product reads the same rows with the pixel on the lane (consecutive words: conflict-free).
// The full-resolution dz is not read but FORMED here: the segment's "dz" pointer is z, the incoming gradient g = dout rides
// along, and dz = gi (mask (g a1 + a2) - mdy - xhat mdyx) (ffm.hip::ffm_dz_kernel's expression, operation for operation) is
// evaluated on the staged registers between the MFMAs of the dX product, per-channel coefficients from an LDS table -- the
// low-resolution operand dz_low = U^T dz comes from ffm_bwd_adj.hip by linearity, so nothing else needs dz and its 134 MB
// are neither written nor read back.  The next chunk's loads and the previous chunk's dX stores are dealt over the half-blocks
// of the dW product: issued in one piece at the top of a chunk they kept every wave at its next load for 2000-5000 of the
// chunk's 23000 cycles with no MFMA in flight (tools/xw_trace.py: cycle stamps at the phase boundaries, -DXW_TRACE).
// Work is dealt in equal contiguous runs of chunks over <= 256 workgroups (one per CU); a workgroup writes its dW tile as
// one slab per segment it touched, and an ordered slab sum finishes dW: no atomics, bit-reproducible.
// (Measured and not kept: walking the full-resolution segment TRANSPOSED -- step j of workgroup i = chunk 228 j + i, so that
// the workgroups read adjacent 128-byte pieces of every row at the same time instead of pieces 2 KB apart: 221 us against 219.)
// (Also measured and not kept: eight lanes per staged row instead of two, i.e. whole 128-byte lines per load instruction and a
// quarter of the line touches: 294 us for the operator against 288.)
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
    const float* dz;   // (B, XW_CO, P): dz -- or, when g is set, z: dz is then formed from (g, z) while the chunk is staged
    const float* g;    // (B, XW_CO, P) the operator's incoming gradient dout, or null
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
    // BatchNorm backward coefficients of a segment with g set (ffm.hip::ffm_dz_kernel has the formula): per channel ...
    const float *mean, *invstd, *bn_w, *bn_b, *mean_dy, *mean_dyx;
    const float *coef_a1, *coef_a2;   // ... and per (image, channel)
};

// -DXW_TRACE (tools/xw_trace.py): cycle stamps of waves 0 and 4 of one workgroup at the phase boundaries of every chunk
#ifdef XW_TRACE
__device__ unsigned long long xw_trace_buf[2 * 32 * 16];
#define XW_T(i)                                                                                                            \
    if (blockIdx.x == 100 && (tid & 255) == 0 && u - u_lo < 32)                                                             \
    xw_trace_buf[((tid >> 8) * 32 + (u - u_lo)) * 16 + (i)] = __builtin_readcyclecounter()
#else
#define XW_T(i)
#endif

__global__ __launch_bounds__(512) void ffm_bwd_xw_kernel(XwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzs = smem;                                   // [2][256][36]
    float* xs = dzs + 2 * XW_CO * XW_PITCH;              // [2][128][36]
    float* red = xs + 2 * XW_CX * XW_PITCH;              // [2][4 channel blocks][2 halves][8 regs][64 lanes] / [16 rows][32 px]
    float* coef = red + 2 * 4 * 16 * 64;                 // [256 channels][8]: mean invstd gamma beta gi mdy mdyx -
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int ci = wave & 3, kh = wave >> 2;             // dX: channel block, contraction half
    const int u_lo = blockIdx.x * a.units_per_wg, u_hi = min(u_lo + a.units_per_wg, a.total_units);
    if (u_lo >= u_hi) return;

    // staging assignment: dz chunk = 256 rows x 32 px, two threads per row (16 px each); X chunk = 128 rows, four threads per row
    const int zr = tid >> 1, zh = tid & 1, xr = tid >> 2, xq = tid & 3;
    f32x4 rz[4], rx[2], rg[4] = {};
    float ca1 = 0.f, ca2 = 0.f;   // a1, a2 of the staged chunk's (image, channel zr)
    // Where a unit (chunk) lives: wave-uniform base pointers of its image and segment + the pixel offset, advanced
    // INCREMENTALLY from chunk to chunk.  The first version located every chunk from scratch -- a scalar-load loop over the
    // segment table and an integer division, three times per iteration (next chunk's loads, previous chunk's stores, this
    // chunk's segment): ~9 dependent scalar-memory round trips right behind the barrier, during which none of the eight waves
    // issues an MFMA (the kernel ran at 0.65 MFMA busy where tools/mfma_probe2.hip shows 0.9 for the bare block structure).
    struct UnitRef {
        const float* z;   // dz rows of the unit's image, at the chunk's first pixel
        const float* x;   // X rows (the segment's 128 channels) of the image, at the chunk's first pixel
        float* dx;        // the same 128 channels of the output gradient
        int P, k, left;   // row pitch; segment; chunks left in this image after this one
        const float* g;   // dout rows (meaningful when comp)
        int cb;           // image * XW_CO: row of the per-(image, channel) coefficients
        bool comp;        // dz = f(g, z) is formed at staging time
    };
    auto make_ref = [&](int u) {
        int k = 0;
        while (k + 1 < a.nseg && u >= a.seg[k + 1].unit_lo) ++k;
        const XwSeg& s = a.seg[k];
        const int cs = u - s.unit_lo, b = cs / s.chunks_per_img, ci_ = cs - b * s.chunks_per_img, p0 = ci_ * XW_PX;
        UnitRef r;
        r.z = s.dz + (size_t)b * XW_CO * s.P + p0;
        r.x = s.x + ((size_t)b * s.x_rows + s.x_c0) * s.P + p0;
        r.dx = s.dx + ((size_t)b * s.x_rows + s.x_c0) * s.P + p0;
        r.P = s.P, r.k = k, r.left = s.chunks_per_img - 1 - ci_;
        r.comp = s.g != nullptr, r.cb = b * XW_CO;
        r.g = (r.comp ? s.g : s.dz) + (size_t)b * XW_CO * s.P + p0;
        return r;
    };
    auto advance = [&](UnitRef& r, int u_next) {   // r describes unit u_next - 1
        if (r.left > 0) {
            r.z += XW_PX, r.x += XW_PX, r.dx += XW_PX, r.g += XW_PX, r.left -= 1;
        } else {
            r = make_ref(u_next);   // next image or next segment: rare (once per >= 32 chunks)
        }
    };
    auto load_chunk = [&](const UnitRef& r) {
        const float* zp = r.z + (size_t)zr * r.P + zh * 16;
        const float* xp = r.x + (size_t)xr * r.P + xq * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) rz[q] = *reinterpret_cast<const f32x4*>(zp + 4 * q);
#pragma unroll
        for (int q = 0; q < 2; ++q) rx[q] = *reinterpret_cast<const f32x4*>(xp + 4 * q);
        if (r.comp) {   // wave-uniform
            const float* gp = r.g + (size_t)zr * r.P + zh * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) rg[q] = *reinterpret_cast<const f32x4*>(gp + 4 * q);
            ca1 = a.coef_a1[r.cb + zr], ca2 = a.coef_a2[r.cb + zr];
        }
    };
    // The same loads dealt over the first five half-blocks of the dW product.  Issued in one piece at the top of the chunk, the
    // 8 x 14 load instructions of a workgroup (two lanes per row: 32 cache lines per instruction) kept the texture-address path
    // busy for ~3000 cycles during which every wave sat at its next load and no MFMA was issued (tools/xw_trace.py: 2000-5000
    // of a chunk's 23000 cycles in "advance + load").
    auto load_part = [&](const UnitRef& r, int t) {
        if (t < 2) {
            const float* zp = r.z + (size_t)zr * r.P + zh * 16 + 8 * t;
            rz[2 * t] = *reinterpret_cast<const f32x4*>(zp), rz[2 * t + 1] = *reinterpret_cast<const f32x4*>(zp + 4);
        } else if (t < 4) {
            if (r.comp) {
                const float* gp = r.g + (size_t)zr * r.P + zh * 16 + 8 * (t - 2);
                rg[2 * (t - 2)] = *reinterpret_cast<const f32x4*>(gp), rg[2 * (t - 2) + 1] = *reinterpret_cast<const f32x4*>(gp + 4);
            }
        } else if (t == 4) {
            const float* xp = r.x + (size_t)xr * r.P + xq * 8;
            rx[0] = *reinterpret_cast<const f32x4*>(xp), rx[1] = *reinterpret_cast<const f32x4*>(xp + 4);
            if (r.comp) ca1 = a.coef_a1[r.cb + zr], ca2 = a.coef_a2[r.cb + zr];
        }
    };
    // dz = gi (mask (g a1 + a2) - mdy - xhat mdyx): ffm_dz_kernel's expression, operation for operation, on the sixteen staged
    // values of the NEXT chunk, a quarter at a time between the MFMAs of the dX product's last four half-blocks (the VALU is
    // idle under a chain of dependent MFMAs; done in one piece in front of the barrier it cost 10 us per launch)
    f32x4 cf0, cf1;
    auto dz_coef = [&]() {
        cf0 = *reinterpret_cast<const f32x4*>(coef + 8 * zr), cf1 = *reinterpret_cast<const f32x4*>(coef + 8 * zr + 4);
    };
    auto dz_quarter = [&](int q, bool on) {   // branch-free (a select on the wave-uniform `on`): the VALU work has to sit in the
        const float mu = cf0[0], is = cf0[1], gw = cf0[2], gb = cf0[3], gi = cf1[0], mdy = cf1[1], mdyx = cf1[2];   // MFMAs' block
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (rz[q][e] - mu) * is;
            const float y = fmaf(xh, gw, gb);
            const float dy = y > 0.f ? fmaf(rg[q][e], ca1, ca2) : 0.f;
            const float v = gi * (dy - mdy - xh * mdyx);
            rz[q][e] = on ? v : rz[q][e];
        }
        // keeps the arithmetic HERE: left alone, instruction sinking moves it down to its use, the LDS store behind the MFMAs
        asm volatile("" : "+v"(rz[q]));
    };
    auto store_chunk = [&](int buf) {   // rows keep their pixel order (see the header: k-step s pairs pixels s and s + 16)
        float* zd = dzs + ((size_t)buf * XW_CO + zr) * XW_PITCH + zh * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(zd + 4 * q) = rz[q];
        float* xd = xs + ((size_t)buf * XW_CX + xr) * XW_PITCH + xq * 8;
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(xd + 4 * q) = rx[q];
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
    // load_w / flush_dw run once per segment.  Their per-lane addresses are recomputed INSIDE them from a lane index made
    // opaque by an empty asm: left visible, the compiler hoisted the 64-bit pointers out of the chunk loop as loop invariants,
    // ran out of registers, spilled them, and put the scratch reloads -- each followed by s_waitcnt vmcnt(0), i.e. a wait for
    // the chunk loads issued a few instructions earlier -- on the loop's common path.
    auto load_w = [&](int k) {
        int ln = lane, wv = wave;
        asm volatile("" : "+v"(ln), "+v"(wv));
        const float* wp = a.w + (size_t)(128 * (wv >> 2) + (ln >> 5)) * a.ldw + a.seg[k].w_c0 + 32 * (wv & 3) + (ln & 31);
#pragma unroll
        for (int s = 0; s < 64; ++s) wf[s] = wp[(size_t)(2 * s) * a.ldw];
    };
    auto flush_dw = [&](int k) {
        int ln = lane, wv = wave;
        asm volatile("" : "+v"(ln), "+v"(wv));
        float* slab = a.slabs + (size_t)(blockIdx.x + k) * XW_CO * XW_CX + (size_t)(32 * wv + 4 * (ln >> 5)) * XW_CX + (ln & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(size_t)acc_row(r) * XW_CX + 32 * j] = dw[j][r];
    };
    // Finish the dX tile of a chunk (32 channels x 32 pixels, two contraction halves): the upper-half wave hands its partial
    // tile to the lower-half wave of its pair through LDS at the end of the chunk (double-buffered: no extra barrier); the
    // lower-half wave adds and stores the rows at the start of the next iteration.
    // WRITE-THROUGH stores (sc1: the line is not kept dirty in the XCD's L2).  With plain or non-temporal stores this kernel
    // took 228 us instead of 188: every row piece stays in L2 as a dirty line, and the streaming reads of dz and X (268 MB
    // through a 4 MB L2) then pay a write-back on eviction IN THEIR OWN MISS PATH -- the loads of the next chunk came back
    // later than one whole chunk of MFMAs (stand-alone probe: chunk loads served from cache 190 us, no dX stores 192 us,
    // both as they are 241 us, sc1 stores 201 us; all + 13 us slab sum; profiles/r04_ffm_bwd_probe.txt).
    // Buffer stores with a SCALAR row offset: one SALU multiply per row instead of 64-bit vector address arithmetic.
    // (Measured and not kept: BOTH waves of a pair finalising half a tile each, the sums transposed through LDS so that the
    // tile leaves as two 16-byte stores per wave: 217 us against 185 -- three dependent LDS round trips in all eight waves
    // right behind the barrier, and 16-byte sc1 stores that touch eight rows each.)
    auto hand_over = [&](int rbuf) {
        if (kh == 1) {
            float* rp = red + ((size_t)(rbuf * 4 + ci) * 16) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) rp[q * 64] = dxa[q];
        }
    };
    auto store_dx = [&](const UnitRef& r, int rbuf, bool valid, int q_lo, int q_hi) {   // rows q_lo .. q_hi - 1 of the tile
        if (!valid || kh != 0) return;
        const float* rp = red + ((size_t)(rbuf * 4 + ci) * 16) * 64 + lane;
        const buf_rsrc rs = make_rsrc(r.dx, 0x7fffffffu);
        const int voff = ((32 * ci + 4 * h) * r.P + li) * 4;
#pragma unroll
        for (int q = q_lo; q < q_hi; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dxa[q] + rp[q * 64]), rs, voff, acc_row(q) * r.P * 4, 16);
    };

    if (a.seg[0].g != nullptr && tid < XW_CO) {   // the per-channel coefficient table (read back at staging time: no registers)
        const float is = a.invstd[tid], gw = a.bn_w[tid];
        f32x4 c0, c1;
        c0[0] = a.mean[tid], c0[1] = is, c0[2] = gw, c0[3] = a.bn_b[tid];
        c1[0] = gw * is, c1[1] = a.mean_dy[tid], c1[2] = a.mean_dyx[tid], c1[3] = 0.f;
        *reinterpret_cast<f32x4*>(coef + 8 * tid) = c0;
        *reinterpret_cast<f32x4*>(coef + 8 * tid + 4) = c1;
    }
    __syncthreads();
    UnitRef nxt = make_ref(u_lo), cur = nxt, prv = nxt;
    int k_cur = cur.k;
    zero_dw();
    load_w(k_cur);
    load_chunk(nxt);
    if (nxt.comp) {
        dz_coef();
#pragma unroll
        for (int q = 0; q < 4; ++q) dz_quarter(q, true);
    }
    store_chunk(0);
    __syncthreads();
    for (int u = u_lo; u < u_hi; ++u) {
        const int buf = (u - u_lo) & 1;
        XW_T(0);
        const bool has_next = u + 1 < u_hi;
        if (has_next) advance(nxt, u + 1);
        XW_T(1);
        XW_T(2);
        if (cur.k != k_cur) {   // wave-uniform: a new segment starts with this chunk
            flush_dw(k_cur);
            zero_dw();
            load_w(cur.k);
            k_cur = cur.k;
        }
        // Sixteen half-blocks of 8 MFMAs per chunk: half-blocks 0-7 = dW column block j = t / 2, k-steps 8 (t & 1) .. + 7
        // (operands: 8 consecutive floats of the wave's dz row and of X row 32 j + li), half-blocks 8-15 = eighths of the dX
        // half-contraction (operand: one float per k-step from the dz rows, pixel on the lane).  The operands of half-block
        // t + 1 are requested BEFORE the MFMAs of t are issued and the order is pinned with scheduling barriers: left to itself
        // the compiler emitted  read -> wait -> 2 MFMAs  for the dX part (every LDS round trip exposed, both waves of a SIMD
        // stalling in lockstep).  Eight MFMAs = 512 cycles cover an LDS round trip; 16-deep blocks cost 16 more registers and
        // spilled once the balanced dX finalisation was in.
        const float* zrow = dzs + ((size_t)buf * XW_CO + 32 * wave + li) * XW_PITCH + 16 * h;
        const float* xrow = xs + ((size_t)buf * XW_CX + li) * XW_PITCH + 16 * h;
        const float* zcol = dzs + ((size_t)buf * XW_CO + 128 * kh + h) * XW_PITCH + li;
        float af[8], bf[2][8];
        auto read8 = [&](const float* p, float* d) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
                d[4 * q] = t[0], d[4 * q + 1] = t[1], d[4 * q + 2] = t[2], d[4 * q + 3] = t[3];
            }
        };
        auto read_col = [&](int e, float* d) {   // k-steps 8 e .. 8 e + 7 of the dX contraction
#pragma unroll
            for (int s = 0; s < 8; ++s) d[s] = zcol[(size_t)(2 * (8 * e + s)) * XW_PITCH];
        };
        // half-block t of dW: pixel half hf = t >> 2 (k-steps 8 hf .. + 7), column block j = t & 3.  Only ONE half of the wave's
        // dz row (8 registers) is resident: the second half is read once the MFMAs of t = 3 are issued (one exposed LDS round
        // trip per chunk) -- with both halves resident the kernel needed 8 registers more than the 256 a wave has at two per
        // SIMD, and a scratch reload in the loop makes the compiler wait for vmcnt(0), i.e. for the chunk loads just issued.
        XW_T(3);
        read8(zrow, af);
        read8(xrow, bf[0]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = t & 3;
            if (t < 7)
                read8(xrow + (size_t)32 * ((t + 1) & 3) * XW_PITCH + 8 * ((t + 1) >> 2), bf[(t + 1) & 1]);
            else
                read_col(0, bf[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) dw[j] = mfma32(af[s], bf[t & 1][s], dw[j]);
            __builtin_amdgcn_sched_barrier(0);
            if (t == 3) read8(zrow + 8, af);
            // between the half-blocks: two rows of the PREVIOUS chunk's dX tile leave (its two halves' hand-overs landed before the
            // barrier; dxa is not touched by the dW product) and a share of the NEXT chunk's loads is issued
            store_dx(prv, buf ^ 1, u > u_lo, 2 * t, 2 * t + 2);
            if (has_next) load_part(nxt, t);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[r] = 0.f;
        XW_T(4);
        const bool comp_next = u + 1 < u_hi && nxt.comp;   // wave-uniform
        dz_coef();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e < 7) read_col(e + 1, bf[(e + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) dxa = mfma32(wf[8 * e + s], bf[e & 1][s], dxa);
            if (e >= 4) {
                dz_quarter(e - 4, comp_next);
#pragma unroll
                for (int s = 0; s < 8; ++s) {   // program order: one MFMA, five VALU, one MFMA, ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        XW_T(5);
        hand_over(buf);
        XW_T(6);
        // (staging the next chunk here, at the end, or between the two products -- its 16-byte LDS writes draining under the
        // dX MFMAs -- measured the same: 183 against 185 us)
        if (u + 1 < u_hi) store_chunk(buf ^ 1);
        XW_T(7);
        __syncthreads();
        XW_T(8);
        prv = cur, cur = nxt;
    }
    store_dx(prv, (u_hi - 1 - u_lo) & 1, true, 0, 16);
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
    // 64 elements per workgroup; the slab range of the segment in four quarters, one wave each: eight slabs in flight per
    // thread, ascending order in eight interleaved partial sums, then the four quarters in order through LDS -- a fixed
    // association, whatever the placement.  (One element per thread over the whole range was one chain of 228 loads, eight in
    // flight: 14 us for 34 MB; four elements per thread with 16-byte loads left 96 workgroups: 33 us.)
    __shared__ float part[4][64];
    const int k = blockIdx.y, e = threadIdx.x & 63, qt = threadIdx.x >> 6, i = blockIdx.x * 64 + e;   // i over XW_CO * XW_CX
    if (k >= a.nseg) return;
    const size_t stride = (size_t)XW_CO * XW_CX;
    const int n = a.wg_hi[k] - a.wg_lo[k] + 1, per = (n + 3) >> 2, lo = qt * per, hi = min(n, lo + per);
    const float* p = a.slabs + (size_t)(a.wg_lo[k] + k) * stride + i;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int w = lo;
    for (; w + 7 < hi; w += 8) {
        s0 += p[(size_t)w * stride], s1 += p[(size_t)(w + 1) * stride], s2 += p[(size_t)(w + 2) * stride];
        s3 += p[(size_t)(w + 3) * stride], s4 += p[(size_t)(w + 4) * stride], s5 += p[(size_t)(w + 5) * stride];
        s6 += p[(size_t)(w + 6) * stride], s7 += p[(size_t)(w + 7) * stride];
    }
    for (; w < hi; ++w) s0 += p[(size_t)w * stride];
    part[qt][e] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    if (qt == 0) a.out[(size_t)(i / XW_CX) * a.ldo + a.col_off[k] + (i % XW_CX)] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
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
// dz_coef set: `dz` is z, and dz = f(dout, z, coefficients) is formed inside the kernel (the full-resolution dz is never stored)
hipError_t ffm_bwd_fused_run(const float* dz, const float* dzl, const float* fsp, const float* low, const float* w, int B,
                             int Cs, int Cc, int P, int Pl, float* dfsp, float* dlow, float* dw, float* slabs,
                             hipStream_t stream, const XwDzCoef* dz_coef) {
    XwArgs a{};
    XwSumArgs sa{};
    const int ldw = Cs + Cc;
    int unit = 0, k = 0;
    a.seg[k] = XwSeg{dz, dz_coef ? dz_coef->g : nullptr, fsp, dfsp, P, Cs, 0, 0, P / XW_PX, unit};
    if (dz_coef) {
        a.mean = dz_coef->mean, a.invstd = dz_coef->invstd, a.bn_w = dz_coef->bn_w, a.bn_b = dz_coef->bn_b;
        a.mean_dy = dz_coef->mean_dy, a.mean_dyx = dz_coef->mean_dyx, a.coef_a1 = dz_coef->coef_a1, a.coef_a2 = dz_coef->coef_a2;
    }
    sa.col_off[k] = 0;
    unit += B * (P / XW_PX);
    for (int c0 = 0; c0 < Cc; c0 += XW_CX) {
        ++k;
        a.seg[k] = XwSeg{dzl, nullptr, low, dlow, Pl, Cc, c0, Cs + c0, Pl / XW_PX, unit};
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
    const size_t lds = ((size_t)2 * (XW_CO + XW_CX) * XW_PITCH + 2 * 4 * 16 * 64 + XW_CO * 8) * sizeof(float);
    static lds_attr_mask mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ffm_bwd_xw_kernel), lds, mask); e != hipSuccess) return e;
    hipLaunchKernelGGL(ffm_bwd_xw_kernel, dim3(nwg), dim3(512), lds, stream, a);
    hipLaunchKernelGGL(ffm_bwd_xw_sum_kernel, dim3(XW_CO * XW_CX / 64, a.nseg), dim3(256), 0, stream, sa);
    return hipGetLastError();
}

}  // namespace cabinet

#ifdef XW_TRACE
extern "C" int cabinet_debug_xw_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cabinet::xw_trace_buf), (size_t)n * sizeof(unsigned long long));
}
#endif
