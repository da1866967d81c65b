// K3 (round 4) -- the FFM forward's big product as a persistent kernel with the BatchNorm statistics in its epilogue.
//
// Reference span: src/models/cabinet.py:143-144 (the 1x1 convolution of ConvBNReLU over cat(fsp, fcp)) with the x4 bilinear
// resize of cabinet.py:228-230 commuted behind the convolution:
//     z[o][p] = sum_c W_s[o][c] fsp[c][p] + U(y_low)[o][p],      y_low = W_c . low  (computed at low resolution before)
// and the first half of BatchNorm2d in training mode: per-channel sum and sum of squares of z.
// Round 3 ran gemm_kmajor (one 256 x 128-pixel tile per workgroup, 4096 workgroups: every tile pays its own prologue, and its
// 134 MB store phase does not overlap the matrix phase: 94 us at 0.55 MFMA busy) and then read z back for the statistics
// (bn_rowstats, 22 us; an in-epilogue reduction per tile cost 60 us more, DESIGN_HISTORY.md).  Here, as in ffm_bwd_fused.hip:
//   * <= 256 workgroups (one per CU) each walk a contiguous run of 64-pixel chunks; wave w owns output channels 32 w .. + 31
//     and keeps W_s[o][c] (64 values per lane) in registers for the whole run; a chunk of fsp (128 x 64) is staged in LDS once
//     and read with the pixel on the lane (consecutive words: conflict-free), two 32-pixel column blocks per wave;
//   * the PIXEL is the MFMA's row and the channel its column (A = staged fsp, B = W): a lane holds ONE channel and four runs
//     of four consecutive pixels per block, so z leaves in 16-byte stores and the statistics are two scalars per lane;
//   * software pipeline: the epilogue of block n (bilinear(y_low) added, statistics, stores) sits between the MFMAs of block
//     n + 1 (two accumulators, alternating);
//   * the upsample term needs, per output row, two y_low rows for the wave's 32 channels: kept in a wave-private LDS image,
//     requested one chunk ahead when the row pair changes (every fourth output row); a quad of pixels is the footprint of
//     one source column, so its six taps are read once and combined with the x4 constants;
//   * statistics: a lane sums (z - pivot) and (z - pivot)^2 in fp32 over its <= few hundred values, pivot = its first value:
//     var = E[z^2] - mean^2 loses (mean / std)^2 ulps in fp32 (test_ffm_fwd_fused_statistics_with_large_channel_means: mean =
//     130 std), the centred sums do not.  At the end they are widened to double, shifted back (sum z = S1 + n p, sum z^2 =
//     S2 + 2 p S1 + n p^2), the two halves of the wave added, and one pair per workgroup and channel goes to
//     stat_part[2][C][nwg], summed in a fixed order by ffm_pool's finalize.
// What tools/xw_trace.py (cycle stamps at the phase boundaries, -DFZ_TRACE) showed on the way: y_low rows fetched with dword
// loads in dependent batches cost 16000-27000 cycles per reload; with the channel on the MFMA row a block's epilogue (16 dword
// stores, 80 LDS reads, 32 accumulating registers) took 6000 cycles against 4200 for its 64 MFMAs, all of it exposed; a wait
// for staged rows placed behind the next chunk's loads waits for those too (vmcnt counts in order).  What is left: two waves
// per SIMD share the matrix pipe at ~0.77 busy (chunk boundary: LDS write -> barrier -> first operand reads with no MFMA in
// flight), at ~1.9 GHz under this load rather than the 2.4 GHz the 157 TFLOP/s peak assumes.
// Exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
#include "blocks.hpp"
#include "common.hpp"

namespace cabinet {

constexpr int FZ_CO = 256, FZ_CS = 128, FZ_PX = 64, FZ_WL = 32, FZ_VW = FZ_WL + 1;

struct FzArgs {
    const float* fsp;    // (B, 128, P)
    const float* w;      // (256, ldw) row-major; columns [0, 128) = W_s
    const float* ylow;   // (B, 256, Hl, Wl)
    float* z;            // (B, 256, P)
    double* stat_part;   // [2][256][nwg], or null (eval mode: no batch statistics)
    int ldw, B, P, W, Hl, Wl;
    float rh, rw;
    int total_units, units_per_wg, nwg;
};

// -DFZ_TRACE (tools/xw_trace.py fwd): cycle stamps of waves 0 and 4 of one workgroup at the phase boundaries of every chunk
#ifdef FZ_TRACE
__device__ unsigned long long fz_trace_buf[2 * 32 * 16];
#define FZ_T(i)                                                                                                            \
    if (blockIdx.x == 100 && (tid & 255) == 0 && u - u_lo < 32)                                                             \
    fz_trace_buf[((tid >> 8) * 32 + (u - u_lo)) * 16 + (i)] = __builtin_readcyclecounter()
#else
#define FZ_T(i)
#endif

__global__ __launch_bounds__(512) void ffm_fwd_z_kernel(FzArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* fs = smem;                                  // [2][128][64] staged fsp chunks
    constexpr int VW = FZ_VW;                          // odd pitch of a staged y_low row (constant: row offsets are immediates)
    float* yl_all = fs + 2 * FZ_CS * FZ_PX;            // [8 waves][2 source rows][32 channels][VW]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    float* yl = yl_all + (size_t)wave * 2 * 32 * VW;
    const int u_lo = blockIdx.x * a.units_per_wg, u_hi = min(u_lo + a.units_per_wg, a.total_units);
    if (u_lo >= u_hi) return;
    const int cpi = a.P / FZ_PX;                       // chunks per image
    // staging: 128 rows x 64 px, four threads per row (16 px each)
    f32x4 rs[4];
    auto load_chunk = [&](const float* base) {   // base: fsp of the image at the chunk's first pixel
        int t = tid;   // opaque: the per-lane offset is recomputed (two VALU) instead of kept live and spilled -- a scratch
        asm volatile("" : "+v"(t));   // reload here waits for the previous chunk's 32 stores (vmcnt is in order)
        const float* p = base + (size_t)(t >> 2) * a.P + (t & 3) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };
    auto store_chunk = [&](int buf) {
        int t = tid;
        asm volatile("" : "+v"(t));
        float* d = fs + ((size_t)buf * FZ_CS + (t >> 2)) * FZ_PX + (t & 3) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(d + 4 * q) = rs[q];
    };

    // A operand: W_s[32 wave + li][2 s + h], s = 0 .. 63 (loaded once; the address is made opaque so that it is not kept
    // live across the loop -- see ffm_bwd_fused.hip)
    float wf[64];
    {
        int ln = lane, wv = wave;
        asm volatile("" : "+v"(ln), "+v"(wv));
        // a lane reads its whole row (32 x 16 bytes; lanes li and li + 32 share the lines) and keeps the columns of its half
        const float* wp = a.w + (size_t)(32 * wv + (ln & 31)) * a.ldw;
        const bool odd = (ln >> 5) != 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wp + 4 * q);
            wf[2 * q] = odd ? v[1] : v[0];
            wf[2 * q + 1] = odd ? v[3] : v[2];
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four loads in flight, not 32 (registers)
        }
    }
    // The MFMA takes the PIXEL as its row and the channel as its column (A = the staged fsp, B = W): a lane then holds ONE channel
    // (32 wave + li) and, per block, the pixels 8 g + 4 h + {0,1,2,3}, g < 4 -- four runs of four consecutive pixels: the
    // statistics are two scalars per lane and z leaves in 16-byte stores.  (With the channel on the row a lane held 16 channels
    // x 1 pixel: 32 accumulating registers and 16 dword stores per block -- 256 store instructions per chunk and workgroup at
    // ~64 cycles each in the address path = as long as the chunk's MFMAs; tools/xw_trace.py fwd.)
    float s1 = 0.f, s2 = 0.f, pivot = 0.f;   // sums of (z - pivot), (z - pivot)^2 over this lane's pixels; pivot = its first value
    bool pivot_open = true;                  // wave-uniform
    // ---- the epilogue of a finished block (32 pixels x 32 channels per wave), one run of four pixels (quad g) per call --------
    // val = acc + bilinear(y_low) from the wave's two staged source rows; centred statistics; 16-byte store.  The four pixels of
    // a quad start at a multiple of 4 = the footprint of source column m: taps m - 1, m, m + 1 (clamped: the fold at the border
    // is align_corners=False's clamp) with the x4 weights 3/8 5/8 | 1/8 7/8 | 7/8 1/8 | 5/8 3/8.
    struct Pending {
        float ly;            // vertical weight of the chunk's row
        int col0;            // first pixel column of the block (ox0 + 32 j)
        int voff;            // byte offset of (channel 32 wave + li, pixel 32 j + 4 h) from the chunk's z
        const float* z;      // z of the chunk's image at the chunk's first pixel (block-uniform)
    };
    auto make_pending = [&](int b_, int p0_, int oy_, int ox0_, int j_) {
        Pending q;
        int y0_, y1_;
        bilinear_taps(oy_, a.rh, a.Hl, y0_, y1_, q.ly);
        q.col0 = ox0_ + 32 * j_;
        q.voff = ((32 * wave + li) * a.P + 32 * j_ + 4 * h) * 4;
        q.z = a.z + (size_t)b_ * FZ_CO * a.P + p0_;
        return q;
    };
    const float* ylc = yl + li * VW;   // this lane's channel in the two staged rows (second row: + 32 VW)
    float ev[6];
    f32x4 val_even = {0.f, 0.f, 0.f, 0.f};   // the even quad of a pair, held until its odd neighbour exists (epi_compute)
    auto epi_read = [&](const Pending& q, int g) {
        const int m = (q.col0 + 8 * g + 4 * h) >> 2, xm = max(m - 1, 0), xp = min(m + 1, a.Wl - 1);
        ev[0] = ylc[xm], ev[1] = ylc[m], ev[2] = ylc[xp];
        ev[3] = ylc[32 * VW + xm], ev[4] = ylc[32 * VW + m], ev[5] = ylc[32 * VW + xp];
    };
    auto epi_compute = [&](const Pending& q, const f32x16& acc_, int g) {
        const float lyc = 1.f - q.ly;
        const float va = lyc * ev[0] + q.ly * ev[3], vb = lyc * ev[1] + q.ly * ev[4], vc = lyc * ev[2] + q.ly * ev[5];
        f32x4 val;
        val[0] = acc_[4 * g + 0] + (0.375f * va + 0.625f * vb);
        val[1] = acc_[4 * g + 1] + (0.125f * va + 0.875f * vb);
        val[2] = acc_[4 * g + 2] + (0.875f * vb + 0.125f * vc);
        val[3] = acc_[4 * g + 3] + (0.625f * vb + 0.375f * vc);
        pivot = pivot_open ? val[0] : pivot;
        pivot_open = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = val[e] - pivot;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
        // Round 5: a lane's quad is 16 bytes of ITS channel row, so a store instruction used to write 32-byte
        // pieces (h = 0, 1) of 32 different rows -- WRITE_SIZE 212 MB for a 134 MB tensor, partial sectors.  Quads 2m and 2m + 1
        // of a row are neighbours in memory: the lanes of a channel pair (li, li ^ 1) swap one quad each (quad_perm [1,0,3,2],
        // folded into the select), after which the even row's two quads sit in the pair's two lanes and leave in ONE instruction
        // as a 64-byte piece (then the odd row's): whole sectors, half as many rows per instruction.  The statistics above use the
        // lane's own values, before the swap.
        if ((g & 1) == 0) {
            val_even = val;   // quad 2m of this lane's row: waits for quad 2m + 1
            return;
        }
        const bool odd_lane = (li & 1) != 0;
        f32x4 lo, hi;   // lo: row li & ~1, hi: row li | 1; the even lane carries quad 2m, the odd lane quad 2m + 1
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (the element goes through a scalar first: __builtin_bit_cast applied to a vector ELEMENT expression takes element 0 for
            // every e with this compiler -- hipcc 7.2 -- which made all four swapped values the first one)
            const float mine = val[e], mine_even = val_even[e];
            const float from_odd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0xB1, 0xF, 0xF, true));
            const float from_even = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_even), 0xB1, 0xF, 0xF, true));
            lo[e] = odd_lane ? from_odd : val_even[e];    // odd lane: the even partner's quad 2m + 1 (it reads `val` of lane li - 1)
            hi[e] = odd_lane ? val[e] : from_even;        // even lane: the odd partner's quad 2m
        }
        // block-uniform descriptor (a wave-dependent base makes every store a readfirstlane waterfall loop); the channel row is
        // part of the lane offset
        const buf_rsrc zr = make_rsrc(q.z, 0x7fffffffu);
        const int quad = (g - 1) + (odd_lane ? 1 : 0), row_shift = odd_lane ? -a.P * 4 : 0;   // offset of row li & ~1 from row li
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), zr, q.voff + row_shift + 32 * quad, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), zr, q.voff + row_shift + a.P * 4 + 32 * quad, 0, 0);
    };
    auto epi_full = [&](const Pending& q, const f32x16& acc_) {   // not overlapped: row changes, end of the run
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            epi_read(q, g);
            epi_compute(q, acc_, g);
        }
    };

    // staging of two y_low rows (32 channels x Wl floats each) as 16-byte pieces: all (<= 8) loads in flight together, kept in
    // registers until the chunk that needs them starts.  (Dword loads in four dependent batches at the point of use:
    // 16000-27000 cycles per reload in tools/xw_trace.py, a fifth of the kernel; 16-byte loads at the point of use: 8000-9500.)
    f32x4 yt0[4], yt1[4];
    auto yl_issue = [&](int b_, int y0_, int y1_) {
        const float* src = a.ylow + ((size_t)b_ * FZ_CO + 32 * wave) * a.Hl * a.Wl;
        const int ppr = a.Wl >> 2, np = 32 * ppr;   // pieces per row; per source row over the 32 channels (<= 256)
        const float inv_ppr = 1.f / (float)ppr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pcl = min(lane + 64 * k, np - 1);
            const int ml = idiv_small(pcl, inv_ppr), piece = pcl - ml * ppr;
            const float* sp = src + (size_t)ml * a.Hl * a.Wl + 4 * piece;
            if (64 * k < np) {   // wave-uniform
                yt0[k] = *reinterpret_cast<const f32x4*>(sp + y0_ * a.Wl);
                yt1[k] = *reinterpret_cast<const f32x4*>(sp + y1_ * a.Wl);
            }
        }
    };
    auto yl_commit = [&]() {
        const int ppr = a.Wl >> 2, np = 32 * ppr;
        const float inv_ppr = 1.f / (float)ppr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pc = lane + 64 * k;
            if (pc < np) {
                const int ml = idiv_small(pc, inv_ppr), off = ml * VW + 4 * (pc - ml * ppr);
#pragma unroll
                for (int e = 0; e < 4; ++e) yl[off + e] = yt0[k][e], yl[32 * VW + off + e] = yt1[k][e];
            }
        }
    };

    // position of unit u: image b, first pixel p0 = (row oy, column ox0)
    int b = u_lo / cpi, p0 = (u_lo - b * cpi) * FZ_PX;
    int oy = p0 / a.W, ox0 = p0 - oy * a.W, yl_y0 = -1, yl_y1 = -1, yl_b = -1;
    const float* fbase = a.fsp + (size_t)b * FZ_CS * a.P + p0;
    load_chunk(fbase);
    {
        int y0, y1;
        float ly;
        bilinear_taps(oy, a.rh, a.Hl, y0, y1, ly);
        yl_issue(b, y0, y1);   // committed by the first iteration
    }
    store_chunk(0);
    __syncthreads();
    // Software pipeline over the blocks (chunk u, column block j): the MFMAs of a block run with the epilogue of the PREVIOUS
    // block between them (two accumulators, alternating).  Done one after the other, MFMA phase and epilogue took 4200 and 6000
    // cycles per block (tools/xw_trace.py fwd): the matrix pipe idle for more than half of the kernel.
    f32x16 acc2[2];
    Pending pend{};
    bool have_pending = false;   // wave-uniform
    for (int u = u_lo; u < u_hi; ++u) {
        const int buf = (u - u_lo) & 1;
        FZ_T(0);
        // next unit's position (wave-uniform, incremental)
        int nb = b, np0 = p0 + FZ_PX, noy = oy, nox0 = ox0 + FZ_PX;
        if (nox0 == a.W) nox0 = 0, noy += 1;
        if (np0 == a.P) np0 = 0, nb += 1, noy = 0, nox0 = 0;
        FZ_T(1);
        // the two y_low rows under this output row for the wave's 32 channels (wave-private: no barrier); they change every
        // fourth output row, so a run of a few rows reloads them once or twice.  The pending epilogue still reads the OLD rows:
        // it is finished first (not overlapped); the new rows were requested one chunk ahead (yl_issue below) and only have to
        // be written to LDS here.
        int y0, y1;
        float ly;
        bilinear_taps(oy, a.rh, a.Hl, y0, y1, ly);
        if (y0 != yl_y0 || y1 != yl_y1 || b != yl_b) {
            if (have_pending) epi_full(pend, acc2[1]);   // the pending block is always a j = 1 block here
            have_pending = false;
            yl_commit();
            yl_y0 = y0, yl_y1 = y1, yl_b = b;
        }
        // (the next chunk's loads go out BEHIND the commit: in front of it, the commit's wait for the staged rows -- vmcnt counts
        // in order -- became a wait for these loads too: 6000-12000 cycles per reload)
        if (u + 1 < u_hi) {   // does the next chunk need other rows?  Request them now: a chunk's worth of MFMAs hides the trip
            load_chunk(a.fsp + (size_t)nb * FZ_CS * a.P + np0);
            int ny0, ny1;
            float nly;
            bilinear_taps(noy, a.rh, a.Hl, ny0, ny1, nly);
            if (ny0 != yl_y0 || ny1 != yl_y1 || nb != yl_b) yl_issue(nb, ny0, ny1);
        }
        FZ_T(2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16& acc = acc2[j];
            const f32x16& accp = acc2[j ^ 1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // eight half-blocks of 8 k-steps; the B operand of half-block t + 1 (and the LDS operands of the pending epilogue's rows
            // 2 t, 2 t + 1) are requested before the 8 MFMAs of t are issued
            const float* col = fs + ((size_t)buf * FZ_CS + h) * FZ_PX + 32 * j + li;
            float bq[2][8];
            auto read_half = [&](int t, float* d) {
#pragma unroll
                for (int s = 0; s < 8; ++s) d[s] = col[(size_t)(2 * (8 * t + s)) * FZ_PX];
            };
            read_half(0, bq[0]);
            if (have_pending) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t < 7) read_half(t + 1, bq[(t + 1) & 1]);
                    if ((t & 1) == 0) epi_read(pend, t >> 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = mfma32(bq[t & 1][s], wf[8 * t + s], acc);
                    if ((t & 1) == 0) {
                        epi_compute(pend, accp, t >> 1);
#pragma unroll
                        for (int s = 0; s < 8; ++s) {   // program order: one MFMA, four VALU, one MFMA, ...
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t < 7) read_half(t + 1, bq[(t + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = mfma32(bq[t & 1][s], wf[8 * t + s], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FZ_T(3 + 2 * j);
            pend = make_pending(b, p0, oy, ox0, j);
            have_pending = true;
            FZ_T(4 + 2 * j);
        }
        if (u + 1 < u_hi) store_chunk(buf ^ 1);
        FZ_T(7);
        __syncthreads();
        FZ_T(8);
        b = nb, p0 = np0, oy = noy, ox0 = nox0;
    }
    if (have_pending) epi_full(pend, acc2[1]);
    if (a.stat_part) {   // the two halves of the wave hold the same channel: shift back, add (in double), one pair per workgroup
        const double n = 32.0 * (double)(u_hi - u_lo);   // values per lane over the run: 16 per block
        const double pv = (double)pivot, d1 = (double)s1, d2 = (double)s2;
        double t1 = d1 + n * pv, t2 = d2 + 2.0 * pv * d1 + n * pv * pv;
        t1 += __shfl_xor(t1, 32, 64);
        t2 += __shfl_xor(t2, 32, 64);
        if (h == 0) {
            const int c = 32 * wave + li;
            a.stat_part[(size_t)c * a.nwg + blockIdx.x] = t1;
            a.stat_part[((size_t)FZ_CO + c) * a.nwg + blockIdx.x] = t2;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
bool ffm_fwd_fused_supported(int B, int Cs, int Co, int H, int W, int Hl, int Wl) {
    const long long P = (long long)H * W;
    return Co == FZ_CO && Cs == FZ_CS && B > 0 && (W % FZ_PX) == 0 && Wl >= 4 && Wl <= FZ_WL && (Wl & 3) == 0 && W == 4 * Wl && Hl >= 1 &&
           (size_t)FZ_CO * P * sizeof(float) < 0x7fffffffull;   // one image's z rows inside a buffer resource
}
int ffm_fwd_fused_nwg(int B, int P) {
    const int units = B * (P / FZ_PX), upw = ceil_div(units, 256);
    return ceil_div(units, upw);
}

hipError_t ffm_fwd_fused_run(const float* fsp, const float* w, int ldw, const float* ylow, int B, int H, int W, int Hl, int Wl,
                             float* z, double* stat_part, hipStream_t stream) {
    FzArgs a{};
    a.fsp = fsp, a.w = w, a.ylow = ylow, a.z = z, a.stat_part = stat_part;
    a.ldw = ldw, a.B = B, a.P = H * W, a.W = W, a.Hl = Hl, a.Wl = Wl;
    a.rh = (float)Hl / (float)H, a.rw = (float)Wl / (float)W;
    a.total_units = B * (a.P / FZ_PX);
    a.units_per_wg = ceil_div(a.total_units, 256);
    a.nwg = ceil_div(a.total_units, a.units_per_wg);
    const size_t lds = ((size_t)2 * FZ_CS * FZ_PX + (size_t)8 * 2 * 32 * FZ_VW) * sizeof(float);
    static lds_attr_mask mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ffm_fwd_z_kernel), 160 * 1024, mask); e != hipSuccess) return e;
    hipLaunchKernelGGL(ffm_fwd_z_kernel, dim3(a.nwg), dim3(512), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cabinet

#ifdef FZ_TRACE
extern "C" int cabinet_debug_fz_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cabinet::fz_trace_buf), (size_t)n * sizeof(unsigned long long));
}
#endif
