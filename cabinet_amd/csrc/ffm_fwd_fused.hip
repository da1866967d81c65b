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
//     and keeps its A operand W_s[o][c] (64 values per lane) in registers for the whole run; a chunk of fsp (128 x 64) is
//     staged in LDS once and read with the pixel on the lane (consecutive words: conflict-free), two column blocks per wave;
//   * the upsample term needs, per output row, two y_low rows for the wave's 32 channels: kept in a wave-private LDS image
//     (reloaded when the row pair changes: every fourth output row), the four taps are combined in the epilogue;
//   * the statistics are accumulated PER LANE over the whole run and reduced across lanes ONCE at the end (a persistent
//     kernel pays the cross-lane reduction per workgroup, not per tile).  A lane sums at most a few dozen values per channel,
//     so it does so in fp32 -- but of z MINUS A PIVOT (the first value the workgroup sees of that channel, kept in LDS):
//     var = E[z^2] - mean^2 loses (mean / std)^2 ulps in fp32 (test_ffm_bn_statistics_with_large_channel_means: mean =
//     100 std), the centred sums do not; 32 double accumulators per lane did not fit the register file beside W and the
//     tile.  The lane sums are widened to double, reduced, shifted back (sum z = S1 + n p, sum z^2 = S2 + 2 p S1 + n p^2) and
//     one pair per workgroup and channel goes to stat_part[2][C][nwg], summed in a fixed order by ffm_pool's finalize;
//   * z leaves with write-through stores (common.hpp::store_wt: no dirty L2 line in the way of the fsp stream).
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
    float s1[16], s2[16];   // per lane: sums of (z - pivot) and (z - pivot)^2 over its pixel columns, rows acc_row(r) + 4 h
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = 0.f, s2[r] = 0.f;
    float* pivot = yl_all + (size_t)8 * 2 * 32 * VW + wave * 32;   // [2 halves][16 rows] per wave

    // position of unit u: image b, first pixel p0 = (row oy, column ox0)
    int b = u_lo / cpi, p0 = (u_lo - b * cpi) * FZ_PX;
    int oy = p0 / a.W, ox0 = p0 - oy * a.W, yl_y0 = -1, yl_y1 = -1, yl_b = -1;
    const float* fbase = a.fsp + (size_t)b * FZ_CS * a.P + p0;
    load_chunk(fbase);
    store_chunk(0);
    __syncthreads();
    for (int u = u_lo; u < u_hi; ++u) {
        const int buf = (u - u_lo) & 1;
        // next unit's position (wave-uniform, incremental)
        int nb = b, np0 = p0 + FZ_PX, noy = oy, nox0 = ox0 + FZ_PX;
        if (nox0 == a.W) nox0 = 0, noy += 1;
        if (np0 == a.P) np0 = 0, nb += 1, noy = 0, nox0 = 0;
        if (u + 1 < u_hi) load_chunk(a.fsp + (size_t)nb * FZ_CS * a.P + np0);
        // the two y_low rows under this output row for the wave's 32 channels (wave-private: no barrier); they change every
        // fourth output row, so a run of a few rows reloads them once or twice.  All loads of a batch are in flight together
        // (a loop of dependent load -> store iterations cost one L2 round trip each: 8 us per two chunks, measured)
        int y0, y1;
        float ly;
        bilinear_taps(oy, a.rh, a.Hl, y0, y1, ly);
        if (y0 != yl_y0 || y1 != yl_y1 || b != yl_b) {
            const float* src = a.ylow + ((size_t)b * FZ_CO + 32 * wave) * a.Hl * a.Wl;
            const float inv_wl = 1.f / (float)a.Wl;
            const int n = 32 * a.Wl;
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {
                float t0[4], t1[4];
                int off[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * (4 * bt + k);
                    const int ml = idiv_small(min(i, n - 1), inv_wl), xs = min(i, n - 1) - ml * a.Wl;
                    const float* sp = src + (size_t)ml * a.Hl * a.Wl + xs;
                    t0[k] = sp[y0 * a.Wl], t1[k] = sp[y1 * a.Wl];
                    off[k] = i < n ? ml * VW + xs : -1;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (off[k] >= 0) yl[off[k]] = t0[k], yl[32 * VW + off[k]] = t1[k];
                __builtin_amdgcn_sched_barrier(0);
            }
            yl_y0 = y0, yl_y1 = y1, yl_b = b;
        }
        // block-uniform descriptor (a wave-dependent base makes every store a readfirstlane waterfall loop); the wave's rows are
        // part of the lane offset
        const buf_rsrc zr = make_rsrc(a.z + (size_t)b * FZ_CO * a.P + p0, 0x7fffffffu);
        const float* vr = yl + 4 * h * VW;
        const float lyc = 1.f - ly;
        // the two 32-pixel column blocks of the chunk one after the other (one 16-register accumulator at a time)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // eight half-blocks of 8 k-steps; the B operand of half-block t + 1 is requested before the 8 MFMAs of t are issued
            const float* col = fs + ((size_t)buf * FZ_CS + h) * FZ_PX + 32 * j + li;
            float bq[2][8];
            auto read_half = [&](int t, float* d) {
#pragma unroll
                for (int s = 0; s < 8; ++s) d[s] = col[(size_t)(2 * (8 * t + s)) * FZ_PX];
            };
            read_half(0, bq[0]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t < 7) read_half(t + 1, bq[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = mfma32(wf[8 * t + s], bq[t & 1][s], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: + bilinear(y_low), centred statistics, write-through stores (rows of 128 contiguous bytes per lane half)
            int x0, x1;
            float lx;
            bilinear_taps(ox0 + 32 * j + li, a.rw, a.Wl, x0, x1, lx);
            const int voff = ((32 * wave + 4 * h) * a.P + 32 * j + li) * 4;
            if (u == u_lo && j == 0) {   // the pivots: the first value of each of the wave's rows (lane 0 of each half)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* v = vr + acc_row(r) * VW;
                    const float v0 = lyc * v[x0] + ly * v[32 * VW + x0], v1 = lyc * v[x1] + ly * v[32 * VW + x1];
                    if (li == 0) pivot[16 * h + r] = acc[r] + ((1.f - lx) * v0 + lx * v1);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* v = vr + acc_row(r) * VW;
                const float v0 = lyc * v[x0] + ly * v[32 * VW + x0], v1 = lyc * v[x1] + ly * v[32 * VW + x1];
                const float val = acc[r] + ((1.f - lx) * v0 + lx * v1);
                const float d = val - pivot[16 * h + r];
                s1[r] += d;
                s2[r] = fmaf(d, d, s2[r]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), zr, voff, acc_row(r) * a.P * 4, 16);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // 16 LDS reads in flight, not 64 (registers)
            }
        }
        if (u + 1 < u_hi) store_chunk(buf ^ 1);
        __syncthreads();
        b = nb, p0 = np0, oy = noy, ox0 = nox0;
    }
    if (a.stat_part) {   // one cross-lane reduction per workgroup: the 32 lanes of a half hold the same 16 rows
        const double n = 64.0 * (double)(u_hi - u_lo);   // values per channel in this workgroup's run
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double d1 = (double)s1[r], d2 = (double)s2[r];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {
                d1 += __shfl_xor(d1, o, 64);
                d2 += __shfl_xor(d2, o, 64);
            }
            if (li == 0) {
                const double pv = (double)pivot[16 * h + r];
                const int c = 32 * wave + acc_row(r) + 4 * h;
                a.stat_part[(size_t)c * a.nwg + blockIdx.x] = d1 + n * pv;
                a.stat_part[((size_t)FZ_CO + c) * a.nwg + blockIdx.x] = d2 + 2.0 * pv * d1 + n * pv * pv;
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
bool ffm_fwd_fused_supported(int B, int Cs, int Co, int H, int W, int Hl, int Wl) {
    const long long P = (long long)H * W;
    return Co == FZ_CO && Cs == FZ_CS && B > 0 && (W % FZ_PX) == 0 && Wl >= 1 && Wl <= FZ_WL && Hl >= 1 &&
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
    const size_t lds = ((size_t)2 * FZ_CS * FZ_PX + (size_t)8 * 2 * 32 * FZ_VW + 8 * 32) * sizeof(float);
    static lds_attr_mask mask{0};
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ffm_fwd_z_kernel), 160 * 1024, mask); e != hipSuccess) return e;
    hipLaunchKernelGGL(ffm_fwd_z_kernel, dim3(a.nwg), dim3(512), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cabinet
