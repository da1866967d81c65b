// K11 (round 5) -- dense 3x3 convolution (stride 1, padding 1, no bias) as Winograd F(2x2, 3x3) on the exact-fp32 MFMA.
//
// Reference spans (src/models/cabinet.py): :59 `conva[0]` (Ci -> 256), :68 + :88-89 `b1(torch.cat([x, feat], 1))`
// (Ci + 256 -> 256; SURVEY.md section 8 row f2: "the 3x3 of north_star's phrase"), :160 `conv_out.conv.conv` (256 -> 256, row
// f4), and the autograd backward of each.  Until round 4 these ran on MIOpen (Winograd assembly for forward / data gradient,
// NHWC implicit GEMM + layout transposes for conv_out and the weight gradients): 2.4 + 4.0 ms of a 27.7 ms step.
//
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 2x2 output tile and (k, c) pair: 16 multiplications instead of 36, i.e. 16
// independent GEMMs  M_xi (K x tiles) = U_xi (K x C) . V_xi (C x tiles), xi = 0..15, at 1 / 2.25 of the direct FLOPs.  The
// matrix pipe is the 1/16-rate fp32 one (gfx950 has no xf32 and bf16 cannot hold the 1e-3 contract through BatchNorm's
// batch statistics), so the 2.25x is worth more here than anywhere: a direct implicit GEMM at the 0.62 of peak this repo's
// best fp32 kernels reach would LOSE to MIOpen's Winograd (97 vs 114 TFLOP/s effective).
//
// One launch does the input transform, the 16 products and the output transform; V (16/4 = 4x the input) and M never
// touch HBM:
//   * a workgroup (512 threads) owns 64 output channels x 32 tiles (2 tile rows x 16 tile columns = 4 x 32 output pixels of
//     one image) for ALL 16 xi; wave w owns xi = 2w, 2w+1: 2 xi x 2 (32-channel blocks) = 4 accumulator tiles = 64 registers;
//   * per chunk of 16 input channels every thread loads ONE 4x4 input patch (tile = lane & 31, channel = 2 wave + (lane >> 5)),
//     transforms it in registers (32 additions) and writes its 16 values to LDS as V[xi][channel][tile] (64 consecutive floats
//     per wave and xi: conflict-free); the B operand of wave w's MFMAs is then one ds_read_b32 per k-step from its two xi
//     planes (64 consecutive floats again);
//   * the A operand (the transformed filters U) is written by wino_filter_kernel in MFMA operand order: one 16-byte load per
//     lane = four k-steps, straight from L2 into registers (a 64-channel slice of U is 16 x 64 x C x 4 B = 3.9 MB at C = 960:
//     the XCD-chunked grid gives each XCD ONE slice, so it stays in that XCD's 4 MB L2);
//   * two-pointer input: the chunk loop walks C0 channels of x0, then C1 channels of x1 -- `torch.cat([x, feat], 1)` (8 x 1216 x
//     32 x 32 x 4 B = 40 MB written and read back at config 3) never exists;
//   * epilogue: the 16 accumulator planes meet in LDS (128 KB), a thread applies A^T . A to its (channel, tile) pairs and writes
//     2 x 2 outputs as two 8-byte stores (16 lanes = one 128-byte row segment); optional per-channel (mean, M2) partials of the
//     block for the BatchNorm that follows (Chan-merged later, as bn_act.hip does).
// The data gradient is the same kernel on filters transformed with the spatial flip and the channel roles swapped
// (dx = conv(dy, rot180(w)^T)); the weight gradient is its own kernel below (contraction over tiles).
// Exact fp32 MFMA (v_mfma_f32_32x32x2_f32); Winograd's own rounding (+-1, 1/2 coefficients) stays at the 1e-6 level.
#include <type_traits>

#include "common.hpp"

namespace cabinet {

constexpr int WN_KB = 64;   // output channels per workgroup
constexpr int WN_TB = 32;   // tiles per workgroup: 2 tile rows x 16 tile columns
constexpr int WN_CC = 16;   // input channels per chunk
constexpr int WN_VBUF = 16 * WN_CC * WN_TB;   // floats of one staged V chunk (32 KB)
constexpr int WN_OOB = 0x40000000;           // byte offset of a tap outside the image: beyond every buffer resource (< 1 GB), no 32-bit wrap when two add up

// -DWN_TRACE (tools/wino_trace.py): cycle stamps of waves 0 and 4 of one workgroup (the two waves of one SIMD) at the quarter
// boundaries of the first 32 chunks
#ifdef WN_TRACE
__device__ unsigned long long wn_trace_buf[2 * 32 * 8];
#define WN_T(i)                                                                                                            \
    if (blockIdx.x == 100 && (tid & 255) == 0 && n < 32)                                                                    \
    wn_trace_buf[((tid >> 8) * 32 + n) * 8 + (i)] = __builtin_readcyclecounter()
#else
#define WN_T(i)
#endif

struct WinoArgs {
    const float* x0;   // (B, C0, H, W)
    const float* x1;   // (B, C1, H, W) or null
    const float* u;    // transformed filters, operand order: [16 xi][K/32][C/8][64 lanes][4]
    float* y0;         // output channels [0, K0): (B, K0, H, W)
    float* y1;         // output channels [K0, K): (B, K - K0, H, W) (null when K0 == K)
    float* stat_part;  // optional [2][K][ntb]: (mean, M2) of the block's valid outputs per channel; null: none
    int C0, C1, K, K0;
    int B, H, W;
    int nby, nbx, ntb, nkb;   // tile blocks per image (rows, columns), tile blocks in all, 64-channel blocks
    int accumulate;           // 1: y += result (the second data gradient into a shared input)
};

// ---- filter transform: U = G g G^T, written in MFMA operand order -----------------------------------------------------------
// `rows` output channels (the MFMA row), `q` contracted channels.  dgrad == 0: g = w[row][q] (w: (rows, q, 3, 3));
// dgrad == 1: g = rot180(w[q][row]) (w: (q, rows, 3, 3)) -- the data gradient's filters.
// thread <-> (row block of 32, group of 8 contracted channels, lane): row = 32 rb + (lane & 31), channels 8 g + 2 s + (lane >> 5)
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, int rows, int q, int dgrad,
                                                           float* __restrict__ u) {
    const int gid = blockIdx.x * 256 + threadIdx.x, lane = gid & 63, unit = gid >> 6;
    const int ng = q >> 3, nrb = rows >> 5;
    if (unit >= nrb * ng) return;
    const int rb = unit / ng, g = unit - rb * ng, row = 32 * rb + (lane & 31), hh = lane >> 5;
    f32x4 out[16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 8 * g + 2 * s + hh;
        float f[9];
        if (dgrad) {
            const float* p = w + ((size_t)c * rows + row) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) f[e] = p[8 - e];
        } else {
            const float* p = w + ((size_t)row * q + c) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) f[e] = p[e];
        }
        float gg[4][3];   // G g
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gg[0][j] = f[j];
            gg[1][j] = 0.5f * ((f[j] + f[6 + j]) + f[3 + j]);
            gg[2][j] = 0.5f * ((f[j] + f[6 + j]) - f[3 + j]);
            gg[3][j] = f[6 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[4 * i + 0][s] = gg[i][0];
            out[4 * i + 1][s] = 0.5f * ((gg[i][0] + gg[i][2]) + gg[i][1]);
            out[4 * i + 2][s] = 0.5f * ((gg[i][0] + gg[i][2]) - gg[i][1]);
            out[4 * i + 3][s] = gg[i][2];
        }
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
        *reinterpret_cast<f32x4*>(u + ((((size_t)xi * nrb + rb) * ng + g) * 64 + lane) * 4) = out[xi];
}

// ---- input transform of one 4x4 patch: V = B^T d B ---------------------------------------------------------------------------
__device__ __forceinline__ void wino_bt_d_b(const float (&d)[16], float (&v)[16]) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0 + j] = d[0 + j] - d[8 + j];
        t[4 + j] = d[4 + j] + d[8 + j];
        t[8 + j] = d[8 + j] - d[4 + j];
        t[12 + j] = d[4 + j] - d[12 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[4 * i + 0] = t[4 * i + 0] - t[4 * i + 2];
        v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
        v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
        v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
    }
}

__device__ __forceinline__ f32x4 bload4(buf_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}

// Sum over the 32 lanes of this lane's half (= the 32 tiles of one channel in the epilogues below), result in every lane: four DPP
// steps inside the 16-lane rows (quad swaps, half-row mirror, row mirror: VALU, ~8 cycles each) and ONE cross-row exchange.
// Round 6: the statistics epilogue took three butterfly reductions of five ds_bpermute each per channel -- a dependent chain of
// ~15 LDS-crossbar round trips, 16 channels per thread: tools/instep_cycles.sh counts 1.52-1.54 M cycles for conv_out's forward
// inside the step (where the epilogue leaves the BatchNorm partials) against 1.38 M for the same kernel without them (the data
// gradient; the replayed loop) -- a tenth of the kernel.  (The count of valid outputs needs no reduction at all: it is geometry.)
__device__ __forceinline__ float wino_half_sum(float x) {
    auto dpp = [](float v, auto ctrl_tag) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_tag)::value, 0xF, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    x += dpp(x, std::integral_constant<int, 0x141>{});   // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});   // row_mirror
    return x + __shfl_xor(x, 16, 64);                    // the other row of the half
}

// PAIR (W even): the 4x4 patch of a lane is assembled from ONE 8-byte load per row (columns 2 tx, 2 tx + 1: the tile's own two
// columns) and its neighbours' registers -- column 2 tx - 1 is the left neighbour's second value, column 2 tx + 2 the right
// neighbour's first (DPP row_shr:1 / row_shl:1: a DPP row of 16 lanes IS one tile row of the block and one channel) -- plus one
// dword per row that only the row's first and last lane use (the halo columns 32 bx - 1 and 32 bx + 32; every other lane's offset
// is out of range: no memory request).  8 vector-memory instructions per chunk and thread instead of 16 dword gathers; zero padding
// still costs nothing (a tile right of the image loads zeros, and that is exactly what its left neighbour needs in column W).
// NKB: 32-channel blocks of output channels per workgroup.  2 (64 channels) is the default; 1 is for grids that would not fill the
// chip (config 5's conva / b1 forward: 2 images x 16 tile blocks x 4 channel blocks = 128 workgroups for 256 CUs): twice the
// workgroups with half the MFMAs per chunk each -- the patch transform is done twice as often, but on CUs that were idle.
template <bool PAIR, int NKB>
__global__ __launch_bounds__(512) void wino_conv_kernel(WinoArgs a) {
    constexpr int KBW = 32 * NKB;   // output channels of this workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];   // V: [2][16 xi][16 ch][32 tiles]; epilogue: M [16][64][32]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: a scalar, or every load offset built from it becomes a waterfall loop
    // The channel block is the SLOW index of the XCD-chunked tile list: an XCD works on one 64-channel slice of U (3.9 MB at C = 960)
    // and its share of the tile blocks.  (Measured and not kept: the channel block as the FAST index, so that an XCD fetches each
    // input row once instead of once per channel block -- conv_out's 680 MB of fabric reads are Infinity-Cache hits, 683 vs 690 us;
    // s_setprio 1 for waves 4-7: 185 vs 185, 223 vs 220, 688 vs 694 us; one extra dword load per lane and chunk that pulls the
    // wave's 64 lines of U for chunk n + 3 into L2 ahead of the real loads: 184 vs 179, 232 vs 211, 687 vs 643 us -- the filter
    // stream is not what the MFMAs wait for.)
    const int tau = xcd_chunked_tile(blockIdx.x, a.ntb * a.nkb);
    const int kblk = tau / a.ntb, tb = tau - kblk * a.ntb;
    const int per_img = a.nby * a.nbx, b = tb / per_img, rem = tb - b * per_img, by = rem / a.nbx, bx = rem - by * a.nbx;
    const int HW = a.H * a.W, C = a.C0 + a.C1, nch = C / WN_CC, nch0 = a.C0 / WN_CC;

    // ---- this thread's patch: tile li of the block, channel 2 wave + h of the chunk --------------------------------------
    // Zero padding costs nothing: a raw buffer load whose offset (per-lane + scalar) lies outside the resource returns 0 -- measured,
    // tools/buf_oob_probe.hip: the range check covers voffset + soffset -- so a tap outside the image gets an offset of WN_OOB
    // (>= every image tensor, which conv3x3_shape_ok keeps below 1 GB) instead of a clamped address and a select behind the load
    // (round 5's first version: 16 v_cndmask per chunk and thread).
    const int ty = 2 * by + (li >> 4), tx = 16 * bx + (li & 15);
    int poff[PAIR ? 5 : 16];   // PAIR: [row] pair offset, [4] halo offset
    if constexpr (PAIR) {
        // ONE halo load per chunk: lane t = 0..3 of a tile row fetches the LEFT halo (column 32 bx - 1) of patch row t, lane 12 + t the
        // RIGHT halo (column 32 bx + 32) of patch row t; a quad broadcast (DPP quad_perm [t,t,t,t]) then hands row t's halo to the
        // row's first lane (from its own quad) and to its last lane (from the last quad) in one move
        const int t = li & 15, hr = t & 3, r = 2 * ty - 1 + hr;
        const int hc = t < 4 ? 32 * bx - 1 : (t >= 12 ? 32 * bx + 32 : -1);
        poff[4] = ((unsigned)r < (unsigned)a.H && (unsigned)hc < (unsigned)a.W) ? ((2 * wave + h) * HW + r * a.W + hc) * 4 : WN_OOB;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 2 * ty - 1 + i;
        if constexpr (PAIR) {
            const bool rin = (unsigned)r < (unsigned)a.H;
            const int base = ((2 * wave + h) * HW + r * a.W) * 4;
            poff[i] = (rin && 2 * tx < a.W) ? base + 2 * tx * 4 : WN_OOB;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 2 * tx - 1 + j;
                poff[4 * i + j] = ((unsigned)r < (unsigned)a.H && (unsigned)c < (unsigned)a.W) ? ((2 * wave + h) * HW + r * a.W + c) * 4 : WN_OOB;
            }
        }
    }
    const buf_rsrc rx0 = make_rsrc(a.x0 + (size_t)b * a.C0 * HW, (unsigned)((size_t)a.C0 * HW * 4));
    const buf_rsrc rx1 = make_rsrc(a.x1 ? a.x1 + (size_t)b * a.C1 * HW : a.x0, (unsigned)((size_t)(a.x1 ? a.C1 : a.C0) * HW * 4));
    float pd[16];   // PAIR: pd[4 i + 1], pd[4 i + 2] = the pair of row i, pd[0] = this lane's halo value; the rest is filled by patch_ready()
    auto load_patch = [&](int n) {   // chunk n -> pd, zero padding included
        const bool first = n < nch0;   // wave-uniform
        const int soff = (first ? n : n - nch0) * WN_CC * HW * 4;
        if constexpr (PAIR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 pr = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(first ? rx0 : rx1, poff[i], soff, 0));
                pd[4 * i + 1] = pr[0], pd[4 * i + 2] = pr[1];
            }
            pd[0] = bload(first ? rx0 : rx1, poff[4], soff);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) pd[e] = bload(first ? rx0 : rx1, poff[e], soff);
        }
    };
    // PAIR: columns 2 tx - 1 and 2 tx + 2 from the neighbours (the row's first / last lane keeps its halo value: bound_ctrl off)
    auto patch_ready = [&]() {
        if constexpr (PAIR) {
            const int hv = __builtin_bit_cast(int, pd[0]);
            const int hq[4] = {__builtin_amdgcn_update_dpp(0, hv, 0x00, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0x55, 0xF, 0xF, true),
                               __builtin_amdgcn_update_dpp(0, hv, 0xAA, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0xFF, 0xF, 0xF, true)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int halo = hq[i];   // quad_perm [i,i,i,i]: row i's halo, in the row's first quad the left one, in its last the right one
                const int left = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, pd[4 * i + 2]), 0x111, 0xF, 0xF, false);   // row_shr:1
                const int right = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, pd[4 * i + 1]), 0x101, 0xF, 0xF, false);  // row_shl:1
                pd[4 * i] = __builtin_bit_cast(float, left), pd[4 * i + 3] = __builtin_bit_cast(float, right);
            }
        }
    };
    auto transform_store = [&](int buf) {
        float v[16];
        patch_ready();
        wino_bt_d_b(pd, v);
        float* dst = smem + buf * WN_VBUF + (2 * wave + h) * WN_TB + li;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) dst[xi * WN_CC * WN_TB] = v[xi];
    };

    // ---- A operand: U rows of the wave's two xi, two 32-channel blocks; one 16-byte load = four k-steps -------------------
    const int ng = C >> 3, nrb = a.K >> 5;
    const buf_rsrc ru = make_rsrc(a.u, (unsigned)((size_t)16 * a.K * C * 4));
    f32x4 ua[2][NKB][2], ub[2][NKB][2];   // [xi][channel block][half chunk]
    auto load_u = [&](f32x4 (&dst)[2][NKB][2], int n) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int hc = 0; hc < 2; ++hc) {
                    const int soff = ((((2 * wave + x) * nrb + NKB * kblk + kb) * ng + 2 * n + hc) * 64) * 16;
                    dst[x][kb][hc] = bload4(ru, lane * 16, soff);
                }
    };

    f32x16 acc[2][NKB];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][kb][r] = 0.f;

    // ---- the chunk loop as ONE basic block per chunk, software-pipelined --------------------------------------------------------
    // A chunk is 32 MFMAs per wave (2 xi x 2 channel blocks x 8 k-steps) = 2048 cycles of the SIMD's matrix pipe; everything else a
    // wave does for a chunk -- 8 filter loads, 16 patch loads, the input transform of the NEXT chunk (32 additions),
    // 8 LDS stores, 8 LDS operand reads -- is ~100 instructions that must sit BETWEEN its MFMAs, or the two waves of a SIMD (in
    // lockstep through the barrier) leave the matrix pipe idle while they both transform: the first version, with the transform in
    // its own conditional block behind the 32 MFMAs, ran at 0.51 MFMA busy (profiles/r05_conv3x3_first_counters.json).  So the tail
    // is handled by clamping the chunk index (the last chunk transforms a chunk nobody reads) instead of branching, and four fenced
    // quarters of 8 MFMAs carry:
    //   Q0  the filter loads of chunk n+1, the LDS operand reads of this chunk's second half
    //   Q1  the 32 additions of B^T d B on patch n+1, then the 16 loads of patch n+2 into the registers they free
    //   Q2  the 16 LDS stores of V[n+1]            -- barrier: V[n+1] is complete --
    //   Q3  the LDS operand reads of chunk n+1's first half (so no MFMA waits for LDS after the barrier)
    // All reads of V[n] are issued by the end of Q0 of chunk n, i.e. in front of barrier n; V[n+2] is written behind barrier n+1:
    // two buffers suffice.
    // What the cycle stamps (-DWN_TRACE, tools/wino_trace.py, profiles/r05_wino_phase_stamps.txt) say about the remaining third: a
    // chunk takes 5600-5900 cycles for 2 x 32 MFMAs = 4096 cycles of the SIMD's pipe, and the difference is the other ~70 instructions
    // of each wave, SERIALLY: on gfx950 the fp32 MFMA runs at the vector rate and nothing of the partner wave that touches the VGPR
    // file executes beside it.  Three rebuilds of this loop measured that: (i) waves 0-3 running the chunk's 32 MFMAs back to back out
    // of registers (2080 cycles: the pipe CAN be kept full) while waves 4-7 do the chunk's loads / transform / LDS traffic, then roles
    // swapped at a barrier -- the working wave's first wait for returning data (a vmcnt wait, or with the loads removed the first LDS
    // read) lasted exactly as long as the partner's matrix segment, every chunk, so the period was 2 x (2100 + 700) = 5600, no
    // better; inserting s_nop gaps between the dense MFMAs changed nothing; (ii) the input region staged once per workgroup through
    // LDS (3 wide loads per thread instead of 16 dword loads: the dword gathers cost ~23 cycles each in the texture path) -- 650
    // vs 643 us, the LDS round trip costs what the gathers did; (iii) the barrier replaced by an LDS arrival counter with four MFMAs
    // of slack: 686 vs 643 us (the older wave of a SIMD wins every arbitration, runs ahead and waits; the skew is systematic).
    // So time = MFMA time + everything else, and the lever left is less "everything else" per MFMA (a 128-channel block per
    // workgroup needs 64 more accumulator registers than two waves per SIMD have).
    float bva[2][4], bvb[2][4];   // B operands (V) of half 0 / half 1 of the current chunk: [xi][k-step]
    float vout[16];
    auto read_v = [&](float (&dst)[2][4], int buf, int hc) {
        const float* vb = smem + buf * WN_VBUF + (2 * wave) * WN_CC * WN_TB + h * WN_TB + li;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[x][s] = vb[x * WN_CC * WN_TB + (2 * (4 * hc + s)) * WN_TB];
    };
    auto mfma_quarter = [&](const f32x4 (&uu)[2][NKB][2], const float (&bv)[2][4], int hc, int s0) {
#pragma unroll
        for (int s = s0; s < s0 + 2; ++s)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) acc[x][kb] = mfma32(uu[x][kb][hc][s], bv[x][s], acc[x][kb]);
    };
    // group-barrier recipes: `per` instructions of class `mask` behind each of `n` MFMAs (0x002 VALU, 0x020 VMEM read, 0x100 DS read,
    // 0x200 DS write)
#define WN_PIN(n, mask, per)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                    \
    }
    auto body = [&](const f32x4 (&ucur)[2][NKB][2], f32x4 (&unext)[2][NKB][2], int n, int buf) {
        const int n1 = min(n + 1, nch - 1), n2 = min(n + 2, nch - 1);
        // ---- Q0
        __builtin_amdgcn_sched_barrier(0);
        WN_T(0);
        load_u(unext, n1);
        read_v(bvb, buf, 1);
        mfma_quarter(ucur, bva, 0, 0);
        WN_PIN(2 * NKB, 0x020, 2)    // 4 NKB filter loads
        WN_PIN(2 * NKB, 0x100, 4 / NKB)    // LDS operand reads (ds_read2: 4 instructions; the recipe tolerates fewer)
        __builtin_amdgcn_sched_barrier(0);
        WN_T(1);
        // ---- Q1
        patch_ready();
        wino_bt_d_b(pd, vout);
        load_patch(n2);
        mfma_quarter(ucur, bva, 0, 2);
        WN_PIN(4 * NKB, 0x002, (PAIR ? 12 : 8) / NKB)    // 32 additions (+ 12 DPP moves)
        __builtin_amdgcn_sched_barrier(0);
        WN_T(2);
        // ---- Q2
        {
            float* dst = smem + (buf ^ 1) * WN_VBUF + (2 * wave + h) * WN_TB + li;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) dst[xi * WN_CC * WN_TB] = vout[xi];
        }
        mfma_quarter(ucur, bvb, 1, 0);
        WN_PIN(4 * NKB, 0x200, 2 / NKB)    // 8 LDS stores (ds_write2)
        __builtin_amdgcn_sched_barrier(0);
        WN_T(3);
        __syncthreads();
        WN_T(4);
        // ---- Q3
        read_v(bva, buf ^ 1, 0);
        mfma_quarter(ucur, bvb, 1, 2);
        WN_PIN(4, 0x100, 1)
        __builtin_amdgcn_sched_barrier(0);
        WN_T(5);
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------
    load_patch(0);
    load_u(ua, 0);
    transform_store(0);
    load_patch(min(1, nch - 1));
    __syncthreads();
    read_v(bva, 0, 0);
    for (int n = 0; n < nch; n += 2) {
        body(ua, ub, n, 0);
        if (n + 1 < nch) body(ub, ua, n + 1, 1);
    }
    __syncthreads();   // every wave is past its last LDS operand read: the epilogue's planes may overwrite the V buffers
#undef WN_PIN

    // ---- epilogue: the 16 planes meet in LDS, output transform Y = A^T M A ---------------------------------------------------
    // (the loop's last barrier has passed: V is dead)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                smem[((2 * wave + x) * KBW + 32 * kb + acc_row(r) + 4 * h) * WN_TB + li] = acc[x][kb][r];
    __syncthreads();
    const int k0 = kblk * KBW;
    float* ybase;
    int krow0, kimg;   // first channel of the block inside its output tensor, channels of that tensor
    if (k0 < a.K0) ybase = a.y0, krow0 = k0, kimg = a.K0;
    else ybase = a.y1, krow0 = k0 - a.K0, kimg = a.K - a.K0;
    const int t = tid & 31, oty = 2 * by + (t >> 4), otx = 16 * bx + (t & 15);
    const int oy = 2 * oty, ox = 2 * otx;
    const bool even_w = (a.W & 1) == 0;
#pragma unroll
    for (int i = 0; i < 2 * NKB; ++i) {
        const int kk = (tid >> 5) + 16 * i;   // channel of the block
        float m[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi] = smem[(xi * KBW + kk) * WN_TB + t];
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = (m[j] + m[4 + j]) + m[8 + j];
            s1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
        }
        float o[2][2];
        o[0][0] = (s0[0] + s0[1]) + s0[2], o[0][1] = (s0[1] - s0[2]) - s0[3];
        o[1][0] = (s1[0] + s1[1]) + s1[2], o[1][1] = (s1[1] - s1[2]) - s1[3];
        float* yp = ybase + ((size_t)b * kimg + krow0 + kk) * HW;
        const bool vx0 = ox < a.W, vx1 = ox + 1 < a.W;
        float cnt = 0.f, sum = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool vy = oy + r < a.H;
            float* p = yp + (size_t)(oy + r) * a.W + ox;
            if (vy && vx1 && even_w) {
                f32x2 val = {o[r][0], o[r][1]};
                if (a.accumulate) {
                    const f32x2 old = *reinterpret_cast<const f32x2*>(p);
                    val[0] += old[0], val[1] += old[1], o[r][0] = val[0], o[r][1] = val[1];
                }
                *reinterpret_cast<f32x2*>(p) = val;
            } else if (vy) {
                if (vx0) {
                    if (a.accumulate) o[r][0] += p[0];
                    p[0] = o[r][0];
                }
                if (vx1) {
                    if (a.accumulate) o[r][1] += p[1];
                    p[1] = o[r][1];
                }
            }
            if (vy && vx0) cnt += 1.f, sum += o[r][0];
            if (vy && vx1) cnt += 1.f, sum += o[r][1];
        }
        if (a.stat_part) {   // (mean, M2) of the block's valid outputs of channel kk: two half-wave reductions, fixed order
            // valid outputs of the block: geometry (4 rows x 32 columns clipped at the image), the same for every channel
            const float n_blk = (float)(max(0, min(4, a.H - 4 * by)) * max(0, min(32, a.W - 32 * bx)));
            (void)cnt;
            const float mean = wino_half_sum(sum) / fmaxf(n_blk, 1.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool vy = oy + r < a.H;
                if (vy && vx0) m2 += (o[r][0] - mean) * (o[r][0] - mean);
                if (vy && vx1) m2 += (o[r][1] - mean) * (o[r][1] - mean);
            }
            m2 = wino_half_sum(m2);
            if (t == 0) {
                a.stat_part[(size_t)(k0 + kk) * a.ntb + tb] = mean;
                a.stat_part[((size_t)a.K + k0 + kk) * a.ntb + tb] = m2;
            }
        }
    }
}

// ---- round 6: 128 output channels per workgroup at ONE wave per SIMD ---------------------------------------
// The 64-channel kernel above stops at 0.66-0.72 of the fp32 MFMA peak because, per chunk, every wave issues ~70 other instructions
// (patch loads, B^T d B, LDS stores / operand reads, filter loads) for 32 MFMAs, and on gfx950 those add up serially with the fp32
// MFMAs (see the loop comment there).  With 256 threads a wave may hold 512 registers: 16 accumulator tiles (256 registers, the
// AGPR half) per wave, so a workgroup owns 128 output channels x 32 tiles for all 16 xi and, per chunk and wave, issues 128 MFMAs
// beside 2 patches (10 loads, 88 VALU, 8 LDS stores), 32 filter loads and 32 LDS operand reads: transform, V traffic and x traffic
// per MFMA halve (x is fetched once per 128-channel block: 2x instead of 4x for K = 256).
//   * A wave owns 32 CHANNELS for ALL 16 xi (not 4 xi for all channels): a lane then holds, for its tile and 16 channel rows, every
//     xi of the product -- the output transform Y = A^T M A runs in registers, the epilogue needs no LDS and no barrier, and the V
//     ring survives it.  V is staged as [channel][tile][16 xi] (pitch 20 floats: the 8-lane groups of a 16-byte LDS access cover
//     all 32 banks once): a thread stores its patch's 16 values as four ds_write_b128, a wave reads the four xi of a group and one
//     k-step as ONE ds_read_b128.
//   * Persistent over tile blocks: a workgroup walks a contiguous range of the (channel block, tile block) list, and the software
//     pipeline runs THROUGH the tile boundary -- the last chunk of a tile transforms and stores the first patches of the next
//     tile, requests its second chunk and its first filters -- so only the first tile of a workgroup pays the load latency of a
//     prologue (one wave per SIMD: there is no partner to cover it).
//   * per chunk, four groups of 4 xi (32 MFMAs each); everything a group needs is requested one group ahead:
//       group g:  filter loads of group g+1, LDS operand reads of group g+1, 32 MFMAs, and
//       g = 0: B^T d B of patch A of the next chunk + its 4 LDS stores     g = 1: the same for patch B
//       g = 2: the loads of patch A two chunks ahead, then the chunk's one barrier (V of the next chunk complete)
//       g = 3: the loads of patch B two chunks ahead; its operand reads are the first of the next chunk
// Contracts the channels in the order of the 64-channel kernel (one fma chain per output, chunk by chunk, k-step by k-step) and
// evaluates the same output-transform expressions: outputs and BatchNorm partials are bit-identical to that kernel's (tested).
// Used for grids of at least two rounds of workgroups (conv_out at configs 3 and 5); the attention branch's convolutions (64 or
// 128 tile blocks) keep the 64- / 32-channel kernel.  W even and 8-byte aligned inputs only (the paired patch loads).
constexpr int WN_KB2 = 128;
constexpr int WN_VP = 20;                              // floats per (channel, tile) of the staged V: 16 xi + 4 of padding
constexpr int WN_VBUF2 = WN_CC * WN_TB * WN_VP;        // one staged chunk: 40 KB

__global__ __launch_bounds__(256) void wino_conv128p_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // V ring: [2][16 ch][32 tiles][20]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3, a scalar
    const int HW = a.H * a.W, C = a.C0 + a.C1, nch = C / WN_CC, nch0 = a.C0 / WN_CC;
    const int per_img = a.nby * a.nbx, total = a.ntb * a.nkb;   // a.nkb = K / 128
    const int wg = xcd_chunked_tile(blockIdx.x, gridDim.x);
    const int t_lo = (int)((long long)wg * total / gridDim.x), t_hi = (int)((long long)(wg + 1) * total / gridDim.x);
    if (t_lo >= t_hi) return;
    const int cw = 2 * wave + h;   // channel of patch A inside a chunk; patch B = cw + 8 (scalar offset)

    // ---- a tile of the list: the channel block is the slow index (an XCD works on one slice of U at a time) ----------------------
    struct Tile {
        int kblk, tb, b, by, bx;
        int poff[5];   // per-lane byte offsets of the paired patch loads (rows 0..3, halo), zero padding = WN_OOB
    };
    auto make_tile = [&](int tau) {
        Tile t;
        t.kblk = tau / a.ntb, t.tb = tau - t.kblk * a.ntb;
        t.b = t.tb / per_img;
        const int rem = t.tb - t.b * per_img;
        t.by = rem / a.nbx, t.bx = rem - t.by * a.nbx;
        const int ty = 2 * t.by + (li >> 4), tx = 16 * t.bx + (li & 15);
        const int tt = li & 15, hr = tt & 3, r = 2 * ty - 1 + hr;
        const int hc = tt < 4 ? 32 * t.bx - 1 : (tt >= 12 ? 32 * t.bx + 32 : -1);
        t.poff[4] = ((unsigned)r < (unsigned)a.H && (unsigned)hc < (unsigned)a.W) ? (cw * HW + r * a.W + hc) * 4 : WN_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = 2 * ty - 1 + i;
            t.poff[i] = ((unsigned)rr < (unsigned)a.H && 2 * tx < a.W) ? (cw * HW + rr * a.W + 2 * tx) * 4 : WN_OOB;
        }
        return t;
    };
    Tile tc = make_tile(t_lo), tn = make_tile(min(t_lo + 1, t_hi - 1));   // this tile, the next one (the last tile prefetches itself)

    float pdA[16], pdB[16];   // pd[4 i + 1], pd[4 i + 2] = the pair of row i, pd[0] = this lane's halo value; patch_ready() fills the rest
    // chunk n of tile t (n in [0, nch)): 16 channels of x0 or x1 of image t.b
    auto load_patch = [&](float (&pd)[16], const Tile& t, int n, int which) {
        const bool first = n < nch0;   // wave-uniform
        const float* base = first ? a.x0 + (size_t)t.b * a.C0 * HW : a.x1 + (size_t)t.b * a.C1 * HW;
        const buf_rsrc rx = make_rsrc(base, (unsigned)((size_t)(first ? a.C0 : a.C1) * HW * 4));
        const int soff = ((first ? n : n - nch0) * WN_CC + 8 * which) * HW * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 pr = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, t.poff[i], soff, 0));
            pd[4 * i + 1] = pr[0], pd[4 * i + 2] = pr[1];
        }
        pd[0] = bload(rx, t.poff[4], soff);
    };
    auto patch_ready = [&](float (&pd)[16]) {   // see wino_conv_kernel<PAIR>
        const int hv = __builtin_bit_cast(int, pd[0]);
        const int hq[4] = {__builtin_amdgcn_update_dpp(0, hv, 0x00, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0x55, 0xF, 0xF, true),
                           __builtin_amdgcn_update_dpp(0, hv, 0xAA, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0xFF, 0xF, 0xF, true)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int halo = hq[i];
            const int left = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, pd[4 * i + 2]), 0x111, 0xF, 0xF, false);   // row_shr:1
            const int right = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, pd[4 * i + 1]), 0x101, 0xF, 0xF, false);  // row_shl:1
            pd[4 * i] = __builtin_bit_cast(float, left), pd[4 * i + 3] = __builtin_bit_cast(float, right);
        }
    };
    auto transform_store = [&](float (&pd)[16], int buf, int which) {
        float v[16];
        patch_ready(pd);
        wino_bt_d_b(pd, v);
        float* dst = smem + buf * WN_VBUF2 + ((cw + 8 * which) * WN_TB + li) * WN_VP;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    };

    // ---- A operand: U rows of the wave's 32 channels, the 4 xi of group g, two half chunks; one 16-byte load = four k-steps -------
    const int ng = C >> 3, nrb = a.K >> 5;
    const buf_rsrc ru = make_rsrc(a.u, (unsigned)((size_t)16 * a.K * C * 4));
    auto load_u = [&](f32x4 (&dst)[4][2], int kblk, int g, int n) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int base = (((4 * g + x) * nrb + 4 * kblk + wave) * ng + 2 * n) * 1024;
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) dst[x][hc] = bload4(ru, lane * 16, base + hc * 1024);
        }
    };
    // B operand of group g: k-step s contracts channels 2 s + h; one 16-byte read = the group's four xi
    auto read_v = [&](f32x4 (&dst)[8], int buf, int g) {
        const float* vb = smem + buf * WN_VBUF2 + (h * WN_TB + li) * WN_VP + 4 * g;
#pragma unroll
        for (int s = 0; s < 8; ++s) dst[s] = *reinterpret_cast<const f32x4*>(vb + 2 * s * WN_TB * WN_VP);
    };
    f32x16 acc[16];   // [xi]
    auto zero_acc = [&]() {
#pragma unroll
        for (int x = 0; x < 16; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    };
    zero_acc();
    f32x4 u0[4][2], u1[4][2], v0[8], v1[8];
#define WN_MFMA_GROUP(G, UU, VV)                                                                                       \
    _Pragma("unroll") for (int hc = 0; hc < 2; ++hc) _Pragma("unroll") for (int s = 0; s < 4; ++s)                     \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) acc[4 * (G) + x] = mfma32(UU[x][hc][s], VV[4 * hc + s][x], acc[4 * (G) + x]);
#define WN_PIN(n, mask, per)                                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                        \
    }
    // chunk n of the current tile; positions past its last chunk continue in the next tile (selects, no branches: one basic block)
    auto chunk = [&](int n, int buf) {
        const bool w1 = n + 1 >= nch, w2 = n + 2 >= nch;   // scalars
        const int n1 = w1 ? 0 : n + 1, n2 = w2 ? n + 2 - nch : n + 2;
        const int kb1 = w1 ? tn.kblk : tc.kblk;
        Tile t2;   // the tile of the patches requested in this chunk
        t2.b = w2 ? tn.b : tc.b;
#pragma unroll
        for (int i = 0; i < 5; ++i) t2.poff[i] = w2 ? tn.poff[i] : tc.poff[i];
        // ---- group 0
        __builtin_amdgcn_sched_barrier(0);
        load_u(u1, tc.kblk, 1, n);
        read_v(v1, buf, 1);
        transform_store(pdA, buf ^ 1, 0);
        WN_MFMA_GROUP(0, u0, v0)
        WN_PIN(8, 0x020, 1)     // 8 filter loads
        WN_PIN(8, 0x100, 1)     // 8 LDS operand reads
        WN_PIN(12, 0x002, 4)    // 12 DPP moves + 32 additions
        WN_PIN(4, 0x200, 1)     // 4 LDS stores
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1
        load_u(u0, tc.kblk, 2, n);
        read_v(v0, buf, 2);
        transform_store(pdB, buf ^ 1, 1);
        WN_MFMA_GROUP(1, u1, v1)
        WN_PIN(8, 0x020, 1)
        WN_PIN(8, 0x100, 1)
        WN_PIN(12, 0x002, 4)
        WN_PIN(4, 0x200, 1)
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 2: patch A two chunks ahead; the chunk's barrier
        load_u(u1, tc.kblk, 3, n);
        read_v(v1, buf, 3);
        load_patch(pdA, t2, n2, 0);
        WN_MFMA_GROUP(2, u0, v0)
        WN_PIN(13, 0x020, 1)
        WN_PIN(8, 0x100, 1)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // V of the next chunk is complete; every read of this chunk's V has returned
        // ---- group 3: first operands of the next chunk, patch B two chunks ahead
        load_u(u0, kb1, 0, n1);
        read_v(v0, buf ^ 1, 0);
        load_patch(pdB, t2, n2, 1);
        WN_MFMA_GROUP(3, u1, v1)
        WN_PIN(13, 0x020, 1)
        WN_PIN(8, 0x100, 1)
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- prologue of the workgroup's first tile
    load_patch(pdA, tc, 0, 0);
    load_patch(pdB, tc, 0, 1);
    load_u(u0, tc.kblk, 0, 0);
    transform_store(pdA, 0, 0);
    transform_store(pdB, 0, 1);
    load_patch(pdA, tc, 1, 0);   // nch >= 4 (C % 64 == 0)
    load_patch(pdB, tc, 1, 1);
    __syncthreads();
    read_v(v0, 0, 0);

    const int t = li, oty_in = li >> 4, otx_in = li & 15;
    (void)t;
    for (int tau = t_lo; tau < t_hi; ++tau) {
        // ONE chunk per iteration, the V buffer chosen by a scalar (nch is a multiple of 4: the parity restarts with every tile): a
        // second, conditional copy of the body puts 256 accumulator registers through a phi, and with all 256 AGPRs taken the copies
        // go through scratch (181 spilled registers; this form: none)
        for (int n = 0; n < nch; ++n) chunk(n, n & 1);
        // ---- epilogue in registers: Y = A^T M A for this lane's tile and its 16 channel rows ------------------------------------
        {
            const int k0 = tc.kblk * WN_KB2 + 32 * wave;   // first channel of the wave's rows
            float* ybase;
            int krow0, kimg;
            if (k0 < a.K0) ybase = a.y0, krow0 = k0, kimg = a.K0;   // K0 % 64 == 0: a wave's 32 channels lie on one side
            else ybase = a.y1, krow0 = k0 - a.K0, kimg = a.K - a.K0;
            const int oy = 2 * (2 * tc.by + oty_in), ox = 2 * (16 * tc.bx + otx_in);
            const bool vx = ox < a.W;   // W is even: both columns of the tile are inside, or none
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = acc_row(r) + 4 * h;   // channel row of the wave's block
                float m[16];
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) m[xi] = acc[xi][r];
                float s0[4], s1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0[j] = (m[j] + m[4 + j]) + m[8 + j];
                    s1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
                }
                float o[2][2];
                o[0][0] = (s0[0] + s0[1]) + s0[2], o[0][1] = (s0[1] - s0[2]) - s0[3];
                o[1][0] = (s1[0] + s1[1]) + s1[2], o[1][1] = (s1[1] - s1[2]) - s1[3];
                float* yp = ybase + ((size_t)tc.b * kimg + krow0 + kk) * HW;
                float cnt = 0.f, sum = 0.f;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    if (oy + rr < a.H && vx) {
                        float* p = yp + (size_t)(oy + rr) * a.W + ox;
                        f32x2 val = {o[rr][0], o[rr][1]};
                        if (a.accumulate) {
                            const f32x2 old = *reinterpret_cast<const f32x2*>(p);
                            val[0] += old[0], val[1] += old[1], o[rr][0] = val[0], o[rr][1] = val[1];
                        }
                        *reinterpret_cast<f32x2*>(p) = val;
                        cnt += 1.f, sum += o[rr][0];
                        cnt += 1.f, sum += o[rr][1];
                    }
                }
                if (a.stat_part) {   // (mean, M2) of the block's valid outputs of this channel: the 64-channel kernel's order, same bits
                    const float n_blk = (float)(max(0, min(4, a.H - 4 * tc.by)) * max(0, min(32, a.W - 32 * tc.bx)));
                    (void)cnt;
                    const float mean = wino_half_sum(sum) / fmaxf(n_blk, 1.f);
                    float m2 = 0.f;
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
                        if (oy + rr < a.H && vx) {
                            m2 += (o[rr][0] - mean) * (o[rr][0] - mean);
                            m2 += (o[rr][1] - mean) * (o[rr][1] - mean);
                        }
                    m2 = wino_half_sum(m2);
                    if (li == 0) {
                        a.stat_part[(size_t)(k0 + kk) * a.ntb + tc.tb] = mean;
                        a.stat_part[((size_t)a.K + k0 + kk) * a.ntb + tc.tb] = m2;
                    }
                }
            }
        }
        zero_acc();
        tc = tn;
        tn = make_tile(min(tau + 2, t_hi - 1));
    }
#undef WN_PIN
#undef WN_MFMA_GROUP
}

// ---- the form that ships (round 6): 128 channels per workgroup, one wave per SIMD, wave w owns xi = 4 w .. 4 w + 3 for all four
// 32-channel blocks; V staged as in the 64-channel kernel ([xi][channel][tile]); one tile block per workgroup; the epilogue meets in
// LDS in two halves of 64 channels.  Measured on one box, conv_out of config 3 (tools/time_conv3x3.py): 581 / 579 us forward / data
// gradient (0.752 / 0.755 of the fp32 MFMA peak) against 607 / 596 us for the 64-channel kernel; the persistent all-xi-per-wave form
// above (wino_conv128p_kernel: register epilogue, pipeline through the tile boundary): 629 / 621 us -- every wave reads the WHOLE
// staged V there (4x the LDS read volume) and the register epilogue of 16 channel rows per lane runs with no other wave beside
// it; kept behind CABINET_WINO_128=2 with its bit-equality test.
__global__ __launch_bounds__(256) void wino_conv128_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // V: [2][16 xi][16 ch][32 tiles]; epilogue: M [16][64][32], twice
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3, a scalar
    const int tau = xcd_chunked_tile(blockIdx.x, a.ntb * a.nkb);  // a.nkb = K / 128
    const int kblk = tau / a.ntb, tb = tau - kblk * a.ntb;
    const int per_img = a.nby * a.nbx, b = tb / per_img, rem = tb - b * per_img, by = rem / a.nbx, bx = rem - by * a.nbx;
    const int HW = a.H * a.W, C = a.C0 + a.C1, nch = C / WN_CC, nch0 = a.C0 / WN_CC;

    // ---- the thread's two patches per chunk: tile li, channels cw = 2 wave + h (patch A) and cw + 8 (patch B: scalar offset) ----
    const int ty = 2 * by + (li >> 4), tx = 16 * bx + (li & 15), cw = 2 * wave + h;
    int poff[5];
    {
        const int t = li & 15, hr = t & 3, r = 2 * ty - 1 + hr;
        const int hc = t < 4 ? 32 * bx - 1 : (t >= 12 ? 32 * bx + 32 : -1);
        poff[4] = ((unsigned)r < (unsigned)a.H && (unsigned)hc < (unsigned)a.W) ? (cw * HW + r * a.W + hc) * 4 : WN_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = 2 * ty - 1 + i;
            poff[i] = ((unsigned)rr < (unsigned)a.H && 2 * tx < a.W) ? (cw * HW + rr * a.W + 2 * tx) * 4 : WN_OOB;
        }
    }
    const buf_rsrc rx0 = make_rsrc(a.x0 + (size_t)b * a.C0 * HW, (unsigned)((size_t)a.C0 * HW * 4));
    const buf_rsrc rx1 = make_rsrc(a.x1 ? a.x1 + (size_t)b * a.C1 * HW : a.x0, (unsigned)((size_t)(a.x1 ? a.C1 : a.C0) * HW * 4));
    float pdA[16], pdB[16];   // pd[4 i + 1], pd[4 i + 2] = the pair of row i, pd[0] = this lane's halo value; patch_ready() fills the rest
    auto load_patch = [&](float (&pd)[16], int n, int which) {
        const bool first = n < nch0;   // wave-uniform
        const int soff = ((first ? n : n - nch0) * WN_CC + 8 * which) * HW * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 pr = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(first ? rx0 : rx1, poff[i], soff, 0));
            pd[4 * i + 1] = pr[0], pd[4 * i + 2] = pr[1];
        }
        pd[0] = bload(first ? rx0 : rx1, poff[4], soff);
    };
    auto patch_ready = [&](float (&pd)[16]) {   // see wino_conv_kernel<PAIR>
        const int hv = __builtin_bit_cast(int, pd[0]);
        const int hq[4] = {__builtin_amdgcn_update_dpp(0, hv, 0x00, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0x55, 0xF, 0xF, true),
                           __builtin_amdgcn_update_dpp(0, hv, 0xAA, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0xFF, 0xF, 0xF, true)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int halo = hq[i];
            const int left = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, pd[4 * i + 2]), 0x111, 0xF, 0xF, false);   // row_shr:1
            const int right = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, pd[4 * i + 1]), 0x101, 0xF, 0xF, false);  // row_shl:1
            pd[4 * i] = __builtin_bit_cast(float, left), pd[4 * i + 3] = __builtin_bit_cast(float, right);
        }
    };
    auto transform_store = [&](float (&pd)[16], int buf, int which) {
        float v[16];
        patch_ready(pd);
        wino_bt_d_b(pd, v);
        float* dst = smem + buf * WN_VBUF + (cw + 8 * which) * WN_TB + li;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) dst[xi * WN_CC * WN_TB] = v[xi];
    };

    // ---- A operand: U rows of block x (xi = 4 wave + x), four 32-channel blocks, two half chunks; one 16-byte load = four k-steps
    const int ng = C >> 3, nrb = a.K >> 5;
    const buf_rsrc ru = make_rsrc(a.u, (unsigned)((size_t)16 * a.K * C * 4));
    auto load_u = [&](f32x4 (&dst)[4][2], int x, int n) {
        const int base = (((4 * wave + x) * nrb + 4 * kblk) * ng + 2 * n) * 1024;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) dst[kb][hc] = bload4(ru, lane * 16, base + (kb * ng + hc) * 1024);
    };
    auto read_v = [&](float (&dst)[8], int buf, int x) {
        const float* vb = smem + buf * WN_VBUF + (4 * wave + x) * WN_CC * WN_TB + h * WN_TB + li;
#pragma unroll
        for (int s = 0; s < 8; ++s) dst[s] = vb[2 * s * WN_TB];
    };
    f32x16 acc[4][4];   // [x][channel block]
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][kb][r] = 0.f;
    f32x4 u0[4][2], u1[4][2];
    float v0[8], v1[8];
#define WN_MFMA_BLOCK(X, UU, VV)                                                                                       \
    _Pragma("unroll") for (int hc = 0; hc < 2; ++hc) _Pragma("unroll") for (int s = 0; s < 4; ++s)                     \
        _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) acc[X][kb] = mfma32(UU[kb][hc][s], VV[4 * hc + s], acc[X][kb]);
#define WN_PIN(n, mask, per)                                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                        \
    }
    // per chunk, four blocks of 32 MFMAs (block x: xi = 4 wave + x); everything a block needs is requested one block ahead:
    //   block 0: U / V of block 1, B^T d B of patch A of chunk n+1 and its 16 LDS stores     block 1: the same for patch B
    //   block 2: the loads of patch A of chunk n+2, then the chunk's one barrier (V[n+1] complete, every read of V[n] done)
    //   block 3: the loads of patch B of chunk n+2; its operand reads are the first of chunk n+1
    auto chunk = [&](int n, int buf) {
        const int n1 = min(n + 1, nch - 1), n2 = min(n + 2, nch - 1);
        __builtin_amdgcn_sched_barrier(0);
        load_u(u1, 1, n);
        read_v(v1, buf, 1);
        transform_store(pdA, buf ^ 1, 0);
        WN_MFMA_BLOCK(0, u0, v0)
        WN_PIN(8, 0x020, 1)     // 8 filter loads
        WN_PIN(4, 0x100, 2)     // 8 LDS operand reads (ds_read2st64: 4)
        WN_PIN(16, 0x002, 3)    // 12 DPP moves + 32 additions
        WN_PIN(4, 0x200, 4)     // 16 LDS stores (ds_write2st64: 8)
        __builtin_amdgcn_sched_barrier(0);
        load_u(u0, 2, n);
        read_v(v0, buf, 2);
        transform_store(pdB, buf ^ 1, 1);
        WN_MFMA_BLOCK(1, u1, v1)
        WN_PIN(8, 0x020, 1)
        WN_PIN(4, 0x100, 2)
        WN_PIN(16, 0x002, 3)
        WN_PIN(4, 0x200, 4)
        __builtin_amdgcn_sched_barrier(0);
        load_u(u1, 3, n);
        read_v(v1, buf, 3);
        load_patch(pdA, n2, 0);
        WN_MFMA_BLOCK(2, u0, v0)
        WN_PIN(13, 0x020, 1)
        WN_PIN(4, 0x100, 2)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // V[n+1] is complete; every read of V[n] has returned
        load_u(u0, 0, n1);
        read_v(v0, buf ^ 1, 0);
        load_patch(pdB, n2, 1);
        WN_MFMA_BLOCK(3, u1, v1)
        WN_PIN(13, 0x020, 1)
        WN_PIN(4, 0x100, 2)
        __builtin_amdgcn_sched_barrier(0);
    };
    load_patch(pdA, 0, 0);
    load_patch(pdB, 0, 1);
    load_u(u0, 0, 0);
    transform_store(pdA, 0, 0);
    transform_store(pdB, 0, 1);
    load_patch(pdA, min(1, nch - 1), 0);
    load_patch(pdB, min(1, nch - 1), 1);
    __syncthreads();
    read_v(v0, 0, 0);
    // ONE chunk per iteration, the V buffer chosen by a scalar: a second, conditional copy of the body (`if (n + 1 < nch)`) puts 256
    // accumulator registers through a phi at its merge point, and with all 256 AGPRs taken the copies go through scratch
    // (181 spilled registers; this form: none)
    for (int n = 0; n < nch; ++n) chunk(n, n & 1);
#undef WN_PIN
#undef WN_MFMA_BLOCK

    // ---- epilogue: two halves of 64 channels through LDS (16 planes x 64 x 32 = 128 KB each), Y = A^T M A ------------------------
    const int t = tid & 31, oty = 2 * by + (t >> 4), otx = 16 * bx + (t & 15);
    const int oy = 2 * oty, ox = 2 * otx;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();   // half 0: every wave is past its last V read; half 1: the first half's planes have been read
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[((4 * wave + x) * 64 + 32 * kq + acc_row(r) + 4 * h) * WN_TB + li] = acc[x][2 * half + kq][r];
        __syncthreads();
        const int k0 = kblk * WN_KB2 + 64 * half;
        float* ybase;
        int krow0, kimg;
        if (k0 < a.K0) ybase = a.y0, krow0 = k0, kimg = a.K0;
        else ybase = a.y1, krow0 = k0 - a.K0, kimg = a.K - a.K0;
#pragma unroll 4
        for (int i = 0; i < 8; ++i) {
            const int kk = (tid >> 5) + 8 * i;   // channel of the half block
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = smem[(xi * 64 + kk) * WN_TB + t];
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = (m[j] + m[4 + j]) + m[8 + j];
                s1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
            }
            float o[2][2];
            o[0][0] = (s0[0] + s0[1]) + s0[2], o[0][1] = (s0[1] - s0[2]) - s0[3];
            o[1][0] = (s1[0] + s1[1]) + s1[2], o[1][1] = (s1[1] - s1[2]) - s1[3];
            float* yp = ybase + ((size_t)b * kimg + krow0 + kk) * HW;
            const bool vx = ox < a.W;   // W is even: both columns of the tile are inside, or none
            float cnt = 0.f, sum = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool vy = oy + r < a.H;
                if (vy && vx) {
                    float* p = yp + (size_t)(oy + r) * a.W + ox;
                    f32x2 val = {o[r][0], o[r][1]};
                    if (a.accumulate) {
                        const f32x2 old = *reinterpret_cast<const f32x2*>(p);
                        val[0] += old[0], val[1] += old[1], o[r][0] = val[0], o[r][1] = val[1];
                    }
                    *reinterpret_cast<f32x2*>(p) = val;
                    cnt += 1.f, sum += o[r][0];
                    cnt += 1.f, sum += o[r][1];
                }
            }
            if (a.stat_part) {   // (mean, M2) of the block's valid outputs of channel kk (as wino_conv_kernel: same order, same bits)
                const float n_blk = (float)(max(0, min(4, a.H - 4 * by)) * max(0, min(32, a.W - 32 * bx)));
                (void)cnt;
                const float mean = wino_half_sum(sum) / fmaxf(n_blk, 1.f);
                float m2 = 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (oy + r < a.H && vx) {
                        m2 += (o[r][0] - mean) * (o[r][0] - mean);
                        m2 += (o[r][1] - mean) * (o[r][1] - mean);
                    }
                m2 = wino_half_sum(m2);
                if (t == 0) {
                    a.stat_part[(size_t)(k0 + kk) * a.ntb + tb] = mean;
                    a.stat_part[((size_t)a.K + k0 + kk) * a.ntb + tb] = m2;
                }
            }
        }
    }
}

// ---- weight gradient: dU_xi (K x C) = sum over tiles of dM_xi (K x tiles) . V_xi^T (tiles x C), dw = G^T dU G ---------------------
// dM = A dY A^T (the 2x2 output-gradient tile spread to 4x4), V = B^T d B as in forward: F(3x3, 2x2), the exact adjoint of the
// forward's bilinear form, 16 instead of 36 multiplications per (k, c, tile).  The contraction runs over tiles, so BOTH operands are
// transformed on the fly and meet in LDS:
//   * a workgroup owns 64 output channels x 64 input channels for all 16 xi over a contiguous range of tile chunks (8 tiles of one
//     tile row per chunk); wave w owns xi = 2w, 2w+1: 2 x 2 x 2 accumulator tiles = 128 registers;
//   * per chunk a thread (tile = tid & 7, channel = tid >> 3) loads one 2x2 dy tile and one 4x4 x patch, transforms both and writes
//     2 x 16 values: planes [xi][tile][channel], pitch 68 (a wave's 8 tiles x 8 channels land in 4 tile + channel: conflict-free; 72 was two-way);
//   * every range writes its G^T dU G (9 values per (k, c), via LDS in two halves) to a slab; wino_wgrad_sum_kernel adds the slabs in
//     a fixed order: no atomics, bit-reproducible.
constexpr int WG_KB = 64, WG_CB = 64, WG_TT = 8;
// Round 6: operand planes [xi][channel][8 tiles] (the TILE -- the contracted index -- fastest, no padding), the two 4-tile halves
// of a row swapped where bit 2 of the channel is set.  Which two tiles an MFMA step contracts is free as long as both operands
// agree, so step s takes tile s from lanes 0-31 and tile 4 + s from lanes 32-63: the four values a lane needs of one (xi, channel
// block) are 16 consecutive bytes -- ONE ds_read_b128 per chunk where the [xi][tile][channel] planes of round 5 took four
// ds_read_b32 (8 LDS reads per chunk and wave instead of 32; the swap makes the 8-lane groups of a 16-byte read cover all 32 banks
// once).  A wave's stores are 64 consecutive floats per plane, permuted inside rows: two lanes per bank, as any 64-lane dword store.
constexpr int WG_PLANE = 512;                 // floats of one xi plane of one operand: 64 channels x 8 tiles
constexpr int WG_OPBUF = 16 * WG_PLANE;      // one operand of one chunk: 32 KB
constexpr int WG_BUF = 2 * WG_OPBUF;         // dM + V of one chunk

struct WinoWgArgs {
    const float* dy;   // (B, K, H, W)
    const float* x0;   // (B, C0, H, W)
    const float* x1;   // (B, C1, H, W) or null
    float* slab;       // [nsplit][K][C0 + C1][9]
    int C0, C1, K, B, H, W;
    int TH, ncx, nchunks, nsplit, nkb, ncb;   // tile rows, chunks per tile row, chunks in all, ranges, 64-channel blocks of K and C
};

__device__ __forceinline__ f32x2 bload2(buf_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff_bytes, soff_bytes, 0));
}

template <bool EVEN_W>
__global__ __launch_bounds__(512) void wino_wgrad_kernel(WinoWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][dM | V][16 xi][64 channels][8 tiles]; epilogue: [16][64][32]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: a scalar, or every load offset built from it becomes a waterfall loop
    const int tt = tid & 7, ch = tid >> 3;
    const int wg_slot = ch * 8 + ((((tt >> 2) ^ (ch >> 2)) & 1) << 2) + (tt & 3);   // where (channel ch, tile tt) sits in a plane
    const int unit = xcd_chunked_tile(blockIdx.x, a.nsplit * a.nkb * a.ncb);
    const int per_split = a.nkb * a.ncb, split = unit / per_split, rem = unit - split * per_split;
    const int kblk = rem / a.ncb, cblk = rem - kblk * a.ncb;
    const int q_lo = (int)((long long)split * a.nchunks / a.nsplit), q_hi = (int)((long long)(split + 1) * a.nchunks / a.nsplit);
    const int HW = a.H * a.W, C = a.C0 + a.C1;
    // the 64 input channels of the block lie in x0 or in x1 (C0 is a multiple of 64 when there are two inputs)
    const bool in0 = cblk * WG_CB < a.C0;
    const float* xsrc = in0 ? a.x0 : a.x1;
    const int Cx = in0 ? a.C0 : a.C1, cfirst = in0 ? cblk * WG_CB : cblk * WG_CB - a.C0;

    // position of chunk q: image b, tile row ty, chunk column cx (wave-uniform; advanced with selects, not branches: the chunk
    // loop below is one basic block per chunk)
    struct Pos {
        int b, ty, cx;
    };
    auto pos_of = [&](int q) {
        Pos p;
        const int per_img = a.TH * a.ncx;
        p.b = q / per_img;
        const int r = q - p.b * per_img;
        p.ty = r / a.ncx, p.cx = r - p.ty * a.ncx;
        return p;
    };
    auto advance = [&](Pos& p, bool go) {   // go == false: stay (the clamped tail re-loads the last chunk)
        const int cx1 = p.cx + 1;
        const bool wx = cx1 == a.ncx;
        const int ty1 = p.ty + (wx ? 1 : 0);
        const bool wy = ty1 == a.TH;
        p.cx = go ? (wx ? 0 : cx1) : p.cx;
        p.ty = go ? (wy ? 0 : ty1) : p.ty;
        p.b = go ? p.b + (wy ? 1 : 0) : p.b;
    };

    float px[16];
    f32x2 pg[2];
    // zero padding and masked tiles through the buffer range check (see wino_conv_kernel): a tap outside the image -- row term in the
    // scalar offset, column term in the per-lane offset -- is WN_OOB away and loads as 0; no clamps, no selects
    // EVEN_W: the x patch as one 8-byte load per row (the tile's own two columns) + one halo dword that only the first / last tile of
    // the 8-tile chunk uses; columns 2 tx - 1 and 2 tx + 2 come from the neighbour lanes (patch_ready: DPP row_shr:1 / row_shl:1, the
    // halo selected at the chunk's ends -- a DPP row holds two channels' tile groups) -- see wino_conv_kernel<PAIR>
    auto load_chunk = [&](const Pos& p) {
        const int tx = 8 * p.cx + tt;
        const buf_rsrc rx = make_rsrc(xsrc + (size_t)p.b * Cx * HW, (unsigned)((size_t)Cx * HW * 4));
        if constexpr (EVEN_W) {
            const int cpair = 2 * tx < a.W ? (2 * tx + (cfirst + ch) * HW) * 4 : WN_OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * p.ty - 1 + i;
                const int so = (unsigned)r < (unsigned)a.H ? r * a.W * 4 : WN_OOB;
                const f32x2 pr = bload2(rx, cpair, so);
                px[4 * i + 1] = pr[0], px[4 * i + 2] = pr[1];
            }
            {   // ONE halo load: tile lane tt = 0..3 fetches the LEFT halo (column 16 cx - 1) of patch row tt, lane 4 + t the RIGHT halo
                // (column 16 cx + 16) of row t; quad broadcasts hand them to the chunk's first / last tile (see wino_conv_kernel<PAIR>)
                const int hr = tt & 3, r = 2 * p.ty - 1 + hr, hc = tt < 4 ? 16 * p.cx - 1 : 16 * p.cx + 16;
                const int off = ((unsigned)r < (unsigned)a.H && (unsigned)hc < (unsigned)a.W) ? (hc + (cfirst + ch) * HW + r * a.W) * 4 : WN_OOB;
                px[0] = bload(rx, off, 0);
            }
        } else {
            int cof[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 2 * tx - 1 + j;
                cof[j] = (unsigned)c < (unsigned)a.W ? (c + (cfirst + ch) * HW) * 4 : WN_OOB;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * p.ty - 1 + i;
                const int so = (unsigned)r < (unsigned)a.H ? r * a.W * 4 : WN_OOB;
#pragma unroll
                for (int j = 0; j < 4; ++j) px[4 * i + j] = bload(rx, cof[j], so);
            }
        }
        // dy tile: rows 2 ty, 2 ty + 1, columns 2 tx, 2 tx + 1
        const buf_rsrc rg = make_rsrc(a.dy + (size_t)p.b * a.K * HW, (unsigned)((size_t)a.K * HW * 4));
        const int gbase = (kblk * WG_KB + ch) * HW + 2 * tx;
        const int g0 = 2 * tx < a.W ? gbase * 4 : WN_OOB, g1 = 2 * tx + 1 < a.W ? (gbase + 1) * 4 : WN_OOB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int so = 2 * p.ty + r < a.H ? (2 * p.ty + r) * a.W * 4 : WN_OOB;
            if constexpr (EVEN_W) pg[r] = bload2(rg, g0, so);   // W even: column 2 tx + 1 is inside whenever 2 tx is
            else pg[r][0] = bload(rg, g0, so), pg[r][1] = bload(rg, g1, so);
        }
    };
    // The transforms are cut in pieces so that no transformed value lives across a quarter of the pipeline (with all of them kept
    // from the additions to the stores the kernel needed 146 registers more than the 256 a wave has at two per SIMD):
    //   stage M: dM = A g A^T, A = [[1,0],[1,1],[1,-1],[0,-1]], stored at once
    //   stage A: t = B^T d (the loaded patch is dead afterwards: its registers take the next loads)
    //   stage B: V = t B, stored row by row
    float tsel[16];
    auto stage_m = [&](int buf) {
        float rr[4][2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
            rr[0][e] = pg[0][e], rr[1][e] = pg[0][e] + pg[1][e], rr[2][e] = pg[0][e] - pg[1][e], rr[3][e] = -pg[1][e];
        float* dm = smem + buf * WG_BUF + wg_slot;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dm[(4 * i + 0) * WG_PLANE] = rr[i][0];
            dm[(4 * i + 1) * WG_PLANE] = rr[i][0] + rr[i][1];
            dm[(4 * i + 2) * WG_PLANE] = rr[i][0] - rr[i][1];
            dm[(4 * i + 3) * WG_PLANE] = -rr[i][1];
        }
    };
    auto stage_a = [&]() {
        if constexpr (EVEN_W) {   // patch columns 2 tx - 1 and 2 tx + 2 from the neighbours (the halo at the chunk's ends)
            const int hv = __builtin_bit_cast(int, px[0]);
            const int hq[4] = {__builtin_amdgcn_update_dpp(0, hv, 0x00, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0x55, 0xF, 0xF, true),
                               __builtin_amdgcn_update_dpp(0, hv, 0xAA, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, hv, 0xFF, 0xF, 0xF, true)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int halo = hq[i];   // quad_perm [i,i,i,i]: in the chunk's first quad the left halo of row i, in its second the right one
                const int left = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, px[4 * i + 2]), 0x111, 0xF, 0xF, false);   // row_shr:1
                const int right = __builtin_amdgcn_update_dpp(halo, __builtin_bit_cast(int, px[4 * i + 1]), 0x101, 0xF, 0xF, false);  // row_shl:1
                px[4 * i] = __builtin_bit_cast(float, tt == 0 ? halo : left);
                px[4 * i + 3] = __builtin_bit_cast(float, tt == 7 ? halo : right);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tsel[j] = px[j] - px[8 + j], tsel[4 + j] = px[4 + j] + px[8 + j];
            tsel[8 + j] = px[8 + j] - px[4 + j], tsel[12 + j] = px[4 + j] - px[12 + j];
        }
    };
    auto stage_b = [&](int buf) {
        float* dv = smem + buf * WG_BUF + WG_OPBUF + wg_slot;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dv[(4 * i + 0) * WG_PLANE] = tsel[4 * i + 0] - tsel[4 * i + 2];
            dv[(4 * i + 1) * WG_PLANE] = tsel[4 * i + 1] + tsel[4 * i + 2];
            dv[(4 * i + 2) * WG_PLANE] = tsel[4 * i + 2] - tsel[4 * i + 1];
            dv[(4 * i + 3) * WG_PLANE] = tsel[4 * i + 1] - tsel[4 * i + 3];
        }
    };

    f32x16 acc[2][2][2];   // [xi][k block][c block]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][kb][cb][r] = 0.f;
    // operands of ONE chunk (four k-steps of two tiles): av[xi][k block], bv[xi][c block], component s = k-step s
    struct Ops {
        f32x4 av[2][2], bv[2][2];
    };
    auto read_ops = [&](Ops& o, int buf) {
        const float* ab = smem + buf * WG_BUF + (2 * wave) * WG_PLANE + li * 8 + (((h ^ (li >> 2)) & 1) << 2);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                o.av[x][e] = *reinterpret_cast<const f32x4*>(ab + x * WG_PLANE + 32 * 8 * e);
                o.bv[x][e] = *reinterpret_cast<const f32x4*>(ab + WG_OPBUF + x * WG_PLANE + 32 * 8 * e);
            }
    };
    auto mfma_step = [&](const Ops& o, int s) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[x][kb][cb] = mfma32(o.av[x][kb][s], o.bv[x][cb][s], acc[x][kb][cb]);
    };
#define WG_PIN(n, mask, per)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier((mask), (per), 0);                                                    \
    }

    // Software pipeline, one basic block per chunk (see wino_conv_kernel): a chunk is four k-steps of 8 MFMAs; between them sit
    //   Q0  stage M of chunk q+1 (dM stored)
    //   Q1  stage A, then the loads of chunk q+2 (a whole chunk of MFMAs ahead of their use)
    //   Q2  stage B (V stored)                                                          -- barrier --
    //   Q3  the 8 operand reads (ds_read_b128) of chunk q+1 -- all four k-steps of both operands at once (round 6; round 5 read
    //       8 dwords in every quarter)
    // (Measured and not kept: a second register set for the loaded patches, i.e. requests two chunks ahead of their use instead of
    // one: 820 vs 804, 233 vs 228, 281 vs 286 us -- latency is not what this loop waits for; see wino_conv_kernel on what it is.)
    // (every range holds at least one chunk: nsplit <= nchunks; a wave-uniform condition inside the loop body -- a clamp on the tile
    // row, a `valid ? x : 0` -- becomes a scalar BRANCH that cuts the chunk's basic block: positions advance with selects and the
    // tail re-loads the last chunk instead of branching)
    Pos p_ld = pos_of(q_lo);
    load_chunk(p_ld);
    stage_m(0);
    stage_a();
    stage_b(0);
    advance(p_ld, q_lo + 1 < q_hi);
    load_chunk(p_ld);
    __syncthreads();
    Ops oa, ob;
    read_ops(oa, 0);
    // Q0 stage M of chunk q+1 | Q1 stage A, the loads of chunk q+2 | Q2 stage B -- barrier -- | Q3 the 8 operand reads of chunk q+1
    auto body = [&](int q, int buf, const Ops& oc, Ops& on) {
        __builtin_amdgcn_sched_barrier(0);
        // ---- Q0
        stage_m(buf ^ 1);
        mfma_step(oc, 0);
        WG_PIN(8, 0x002, 2)
        __builtin_amdgcn_sched_barrier(0);
        // ---- Q1
        stage_a();
        advance(p_ld, q + 2 < q_hi);
        load_chunk(p_ld);
        mfma_step(oc, 1);
        WG_PIN(8, 0x002, EVEN_W ? 6 : 3)
        __builtin_amdgcn_sched_barrier(0);
        // ---- Q2
        stage_b(buf ^ 1);
        mfma_step(oc, 2);
        WG_PIN(8, 0x002, 2)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // ---- Q3
        read_ops(on, buf ^ 1);
        mfma_step(oc, 3);
        WG_PIN(8, 0x100, 1)
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int q = q_lo; q < q_hi; q += 2) {
        body(q, 0, oa, ob);
        if (q + 1 < q_hi) body(q + 1, 1, ob, oa);
    }
    __syncthreads();   // every wave is past its last operand read: the epilogue's planes may overwrite the ring
#undef WG_PIN
    // ---- epilogue: dw contribution G^T dU G of this range, two halves of 32 input channels through LDS --------------------------
    float* slab = a.slab + ((size_t)split * a.K + kblk * WG_KB) * C * 9;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (cb) __syncthreads();
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[((2 * wave + x) * WG_KB + 32 * kb + acc_row(r) + 4 * h) * 32 + li] = acc[x][kb][cb][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = (tid >> 5) + 16 * i, cc = tid & 31;
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = smem[(xi * WG_KB + kk) * 32 + cc];
            // G^T = [[1, 1/2, 1/2, 0], [0, 1/2, -1/2, 0], [0, 1/2, 1/2, 1]]
            float t[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = m[j] + 0.5f * (m[4 + j] + m[8 + j]);
                t[1][j] = 0.5f * (m[4 + j] - m[8 + j]);
                t[2][j] = m[12 + j] + 0.5f * (m[4 + j] + m[8 + j]);
            }
            float* o = slab + ((size_t)kk * C + cblk * WG_CB + 32 * cb + cc) * 9;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                o[3 * r + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
                o[3 * r + 1] = 0.5f * (t[r][1] - t[r][2]);
                o[3 * r + 2] = t[r][3] + 0.5f * (t[r][1] + t[r][2]);
            }
        }
    }
}

// dw[i] = sum of the nsplit slabs in ascending order (fixed order: deterministic)
__global__ __launch_bounds__(256) void wino_wgrad_sum_kernel(const float* __restrict__ slab, int n4, int nsplit, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(slab)[i];
    for (int k = 1; k < nsplit; ++k) {
        const f32x4 v = reinterpret_cast<const f32x4*>(slab)[(size_t)k * n4 + i];
        s[0] += v[0], s[1] += v[1], s[2] += v[2], s[3] += v[3];
    }
    reinterpret_cast<f32x4*>(dw)[i] = s;
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
struct WinoShape {
    int B, C0, C1, K, H, W;
};
static inline int wn_th(int H) { return (H + 1) / 2; }
bool conv3x3_supported(int C0, int C1, int K) {
    return C0 > 0 && C1 >= 0 && K > 0 && (C0 % WN_CC) == 0 && (C1 % WN_CC) == 0 && (K % WN_KB) == 0 &&
           // the data gradient runs the same kernel with the roles swapped: its output blocks must not straddle x0 | x1
           (K % WN_CC) == 0 && ((C0 + C1) % WN_KB) == 0 && (C1 == 0 || (C0 % WN_KB) == 0);
}
bool conv3x3_shape_ok(const WinoShape& s) {
    const long long img_in = (long long)(s.C0 > s.C1 ? s.C0 : s.C1) * s.H * s.W * 4, img_out = (long long)s.K * s.H * s.W * 4;
    return conv3x3_supported(s.C0, s.C1, s.K) && s.B > 0 && s.H > 0 && s.W > 0 && img_in < WN_OOB && img_out < WN_OOB &&
           (long long)16 * s.K * (s.C0 + s.C1) * 4 < 0x7fffffffLL;
}
int conv3x3_tile_blocks(int B, int H, int W) { return B * ((wn_th(H) + 1) / 2) * ((wn_th(W) + 15) / 16); }
size_t conv3x3_filter_bytes(int C, int K) { return align_up((size_t)16 * K * C * sizeof(float), 256); }

static hipError_t wino_filter_run(const float* w, int rows, int q, int dgrad, float* u, hipStream_t stream) {
    const int threads = (rows / 32) * (q / 8) * 64;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(ceil_div(threads, 256)), dim3(256), 0, stream, w, rows, q, dgrad, u);
    return hipGetLastError();
}

static hipError_t wino_conv_run(const float* x0, const float* x1, const float* u, int B, int C0, int C1, int K, int K0, int H,
                                int W, float* y0, float* y1, float* stat_part, int accumulate, hipStream_t stream) {
    WinoArgs a{};
    a.x0 = x0, a.x1 = x1, a.u = u, a.y0 = y0, a.y1 = y1, a.stat_part = stat_part;
    a.C0 = C0, a.C1 = C1, a.K = K, a.K0 = K0, a.B = B, a.H = H, a.W = W;
    a.nby = (wn_th(H) + 1) / 2, a.nbx = (wn_th(W) + 15) / 16, a.ntb = B * a.nby * a.nbx, a.nkb = K / WN_KB;
    a.accumulate = accumulate;
    // round 6: 128-channel blocks at one wave per SIMD where the grid still fills at least two rounds of workgroups
    // (CABINET_WINO_128=0: off, =1: wherever the shape allows -- A/B timing)
    const char* k128_s = getenv("CABINET_WINO_128");   // read per call: the tests flip it in-process
    const int k128_env = k128_s ? atoi(k128_s) : -1;
    {
        const bool aligned8 = ((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(x1)) & 7) == 0;
        const bool can = (W & 1) == 0 && aligned8 && (K % WN_KB2) == 0 && (K0 % 64) == 0;
        const int wgs = a.ntb * (K / WN_KB2);
        if (can && k128_env != 0 && (k128_env == 1 || k128_env == 2 || wgs >= 512)) {
            a.nkb = K / WN_KB2;
            if (k128_env == 2) {   // the persistent all-xi-per-wave form (measured slower; kept for A/B and its test)
                static lds_attr_mask mask128p{0};
                if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(wino_conv128p_kernel), 160 * 1024, mask128p); e != hipSuccess)
                    return e;
                int cus = 256;
                {
                    int dev = 0;
                    hipDeviceProp_t prop;
                    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                        cus = prop.multiProcessorCount;
                }
                hipLaunchKernelGGL(wino_conv128p_kernel, dim3(wgs < cus ? wgs : cus), dim3(256), (size_t)2 * WN_VBUF2 * sizeof(float), stream, a);
                return hipGetLastError();
            }
            static lds_attr_mask mask128{0};
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(wino_conv128_kernel), 160 * 1024, mask128); e != hipSuccess)
                return e;
            hipLaunchKernelGGL(wino_conv128_kernel, dim3(wgs), dim3(256), (size_t)16 * 64 * WN_TB * sizeof(float), stream, a);
            return hipGetLastError();
        }
    }
    // 32-channel blocks where 64-channel ones leave CUs idle (fewer than ~0.8 of one round of workgroups); CABINET_WINO_NKB=2 keeps 64
    static const int nkb_env = [] { const char* e = getenv("CABINET_WINO_NKB"); return e ? atoi(e) : 0; }();   // 1 / 2: force (A/B timing)
    const bool small = nkb_env == 1 || (nkb_env != 2 && a.ntb * a.nkb <= 200);
    if (small) a.nkb = K / 32;
    const size_t lds = (size_t)16 * (small ? 32 : WN_KB) * WN_TB * sizeof(float);   // the epilogue's planes (128 KB / 64 KB); the V ring needs 64 KB
    static lds_attr_mask mask[4] = {{0}, {0}, {0}, {0}};
    // CABINET_WINO_PAIR=0: the round-5a patch loads (16 dword gathers) also for even W (A/B timing)
    static const bool pair_on = [] { const char* e = getenv("CABINET_WINO_PAIR"); return !(e && e[0] == '0'); }();
    const bool aligned8 = ((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(x1)) & 7) == 0;
    const bool pair = pair_on && (W & 1) == 0 && aligned8;
#define WN_LAUNCH(P, N, M)                                                                                                       \
    do {                                                                                                                         \
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(wino_conv_kernel<P, N>), 160 * 1024, mask[M]); e != hipSuccess) \
            return e;                                                                                                            \
        hipLaunchKernelGGL((wino_conv_kernel<P, N>), dim3(a.ntb * a.nkb), dim3(512), lds, stream, a);                            \
    } while (0)
    if (pair && small) WN_LAUNCH(true, 1, 0);
    else if (pair) WN_LAUNCH(true, 2, 1);
    else if (small) WN_LAUNCH(false, 1, 2);
    else WN_LAUNCH(false, 2, 3);
#undef WN_LAUNCH
    return hipGetLastError();
}

// forward: ws = transformed filters
size_t conv3x3_fwd_workspace(const WinoShape& s) { return conv3x3_filter_bytes(s.C0 + s.C1, s.K); }
hipError_t conv3x3_fwd_run(const WinoShape& s, const float* x0, const float* x1, const float* w, float* y, float* stat_part, void* ws,
                           hipStream_t stream) {
    float* u = static_cast<float*>(ws);
    if (hipError_t e = wino_filter_run(w, s.K, s.C0 + s.C1, 0, u, stream); e != hipSuccess) return e;
    return wino_conv_run(x0, x1, u, s.B, s.C0, s.C1, s.K, s.K, s.H, s.W, y, nullptr, stat_part, 0, stream);
}

// data gradient: dx0 (B,C0,H,W), dx1 (B,C1,H,W) from dy (B,K,H,W)
size_t conv3x3_dgrad_workspace(const WinoShape& s) { return conv3x3_filter_bytes(s.C0 + s.C1, s.K); }
hipError_t conv3x3_dgrad_run(const WinoShape& s, const float* dy, const float* w, float* dx0, float* dx1, int accumulate0, void* ws,
                             hipStream_t stream) {
    float* u = static_cast<float*>(ws);
    const int C = s.C0 + s.C1;
    if (hipError_t e = wino_filter_run(w, C, s.K, 1, u, stream); e != hipSuccess) return e;
    // accumulate applies to dx0 only (the input two convolutions share); dx1 blocks are plain stores: handled by two launches
    // only when both are asked for and differ -- the common case (accumulate0 == 0) is one launch
    if (!accumulate0 || s.C1 == 0) return wino_conv_run(dy, nullptr, u, s.B, s.K, 0, C, s.C0, s.H, s.W, dx0, dx1, nullptr, accumulate0, stream);
    return hipErrorInvalidValue;   // accumulate into dx0 with a second output: not offered
}


// weight gradient: ranges of the chunk list so that the grid fills whole rounds of 256 workgroups
static int wino_wgrad_nsplit(const WinoShape& s, int nchunks) {
    const int units = (s.K / WG_KB) * ((s.C0 + s.C1) / WG_CB);
    int rounds = ceil_div(units, 256), ns = (256 * rounds) / units;
    if (ns < 1) ns = 1;
    return ns > nchunks ? nchunks : ns;
}
static int wino_wgrad_nchunks(const WinoShape& s) { return s.B * wn_th(s.H) * ((wn_th(s.W) + WG_TT - 1) / WG_TT); }
static size_t wino_wgrad_slab_bytes(const WinoShape& s) {
    return align_up((size_t)wino_wgrad_nsplit(s, wino_wgrad_nchunks(s)) * s.K * (s.C0 + s.C1) * 9 * sizeof(float), 256);
}
static hipError_t wino_wgrad_run(const WinoShape& s, const float* dy, const float* x0, const float* x1, float* dw, float* slab,
                                 hipStream_t stream) {
    WinoWgArgs a{};
    a.dy = dy, a.x0 = x0, a.x1 = x1, a.slab = slab;
    a.C0 = s.C0, a.C1 = s.C1, a.K = s.K, a.B = s.B, a.H = s.H, a.W = s.W;
    a.TH = wn_th(s.H), a.ncx = (wn_th(s.W) + WG_TT - 1) / WG_TT, a.nchunks = wino_wgrad_nchunks(s);
    a.nsplit = wino_wgrad_nsplit(s, a.nchunks), a.nkb = s.K / WG_KB, a.ncb = (s.C0 + s.C1) / WG_CB;
    const size_t lds = (size_t)2 * WG_BUF * sizeof(float);   // 131,072 B (the epilogue's 128 KB of planes alias it)
    static lds_attr_mask mask{0};
    static lds_attr_mask mask_odd{0};
    if ((s.W & 1) == 0) {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(wino_wgrad_kernel<true>), 160 * 1024, mask); e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_wgrad_kernel<true>, dim3(a.nsplit * a.nkb * a.ncb), dim3(512), lds, stream, a);
    } else {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(wino_wgrad_kernel<false>), 160 * 1024, mask_odd); e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_wgrad_kernel<false>, dim3(a.nsplit * a.nkb * a.ncb), dim3(512), lds, stream, a);
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    const int n4 = s.K * (s.C0 + s.C1) * 9 / 4;   // K % 64 == 0: divisible
    hipLaunchKernelGGL(wino_wgrad_sum_kernel, dim3(ceil_div(n4, 256)), dim3(256), 0, stream, slab, n4, a.nsplit, dw);
    return hipGetLastError();
}

// backward: ws = transformed (flipped, transposed) filters | weight-gradient slabs
size_t conv3x3_bwd_workspace(const WinoShape& s) { return conv3x3_filter_bytes(s.C0 + s.C1, s.K) + wino_wgrad_slab_bytes(s); }
hipError_t conv3x3_bwd_run(const WinoShape& s, const float* dy, const float* x0, const float* x1, const float* w, float* dx0,
                           float* dx1, float* dw, void* ws, hipStream_t stream) {
    if (dx0)
        if (hipError_t e = conv3x3_dgrad_run(s, dy, w, dx0, dx1, 0, ws, stream); e != hipSuccess) return e;
    if (dw) {
        float* slab = reinterpret_cast<float*>(static_cast<char*>(ws) + conv3x3_filter_bytes(s.C0 + s.C1, s.K));
        return wino_wgrad_run(s, dy, x0, x1, dw, slab, stream);
    }
    return hipSuccess;
}

}  // namespace cabinet

#ifdef WN_TRACE
extern "C" int cabinet_debug_wn_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cabinet::wn_trace_buf), (size_t)n * sizeof(unsigned long long));
}
#endif
