// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libcabinet_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stddef.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: exact-fp32 matrix FMA, D(32x32) += A(32x2) * B(2x32).
//   lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31];
//   accumulator register r of lane l is D[row = acc_row(r) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// exchange with the other 32-lane half of the wave
__device__ __forceinline__ float swap_half(float x) { return __shfl_xor(x, 32, 64); }

// sum / max over the 32 lanes of this lane's half (result in every lane of the half)
__device__ __forceinline__ float half_sum(float x) {
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// sum of the 16 accumulator registers of one lane (pairwise)
__device__ __forceinline__ float row_sum16(const f32x16& s) {
    const float a = (s[0] + s[1]) + (s[2] + s[3]), b = (s[4] + s[5]) + (s[6] + s[7]);
    const float c = (s[8] + s[9]) + (s[10] + s[11]), d = (s[12] + s[13]) + (s[14] + s[15]);
    return (a + b) + (c + d);
}

// sum over a 256-thread workgroup (4 waves), result in every thread; fixed order; red: 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

// one axis of F.interpolate(mode="bilinear", align_corners=False): source taps and the weight of the upper tap
// (src = (dst + 0.5) * in/out - 0.5 clamped at 0, i1 = min(i0 + 1, in - 1))
__device__ __forceinline__ void bilinear_taps(int dst, float scale, int in_size, int& i0, int& i1, float& lam) {
    const float src = fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
    i0 = min((int)src, in_size - 1);
    i1 = min(i0 + 1, in_size - 1);
    lam = src - (float)i0;
}

// ---- buffer loads: wave-uniform descriptor + per-lane byte offset + SCALAR byte offset, so a strided
// walk over NCHW rows costs one SALU multiply per load and no vector address arithmetic ----
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(buf_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}

// ---- write-through ("sc1") stores for STREAMED outputs -------------------------------------------------------------------
// A plain (or non-temporal) store leaves its line dirty in the XCD's 4 MB L2; a kernel that streams hundreds of MB of reads
// through that L2 at the same time then pays the write-back of a dirty victim IN THE MISS PATH of its own loads.  Measured in
// round 4 on the fused FFM backward (ffm_bwd_fused.hip): 228 us with plain stores of its 67 MB output, 188 us write-through.
// sc1 stores do not keep the line (MI355X_MICROARCH.md, "stores of each flavour"); use them only for outputs the SAME kernel
// never re-reads, 4 or 16 bytes per lane with whole 128-byte lines written per instruction.  It pays where loads are on a
// short leash (an MFMA kernel whose next operand tile must arrive within one tile of arithmetic: the fused FFM backward, the
// epilogue of gemm_kmajor: -6 us on the FFM forward); on pure streaming passes it measured nothing (BatchNorm apply, FFM gate,
// the upsample adjoint's dz rows) or -2 % (BatchNorm backward dx): those keep plain stores.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword ... sc1
}
// row: wave-uniform base of a dense row; byte_off: this lane's 16-byte aligned offset into it (< 2 GB)
__device__ __forceinline__ void store_wt4(float* row, int byte_off, f32x4 v) {
    const buf_rsrc r = __builtin_amdgcn_make_buffer_rsrc(row, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);  // buffer_store_dwordx4 ... sc1
}

// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  Map the linear
// block id to a tile index so that each XCD walks a CONTIGUOUS chunk of the tile list (bijective for
// any total): tiles that share an operand panel then share one L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_chunked_tile(int block, int total) {
    const int xcd = block & 7, slot = block >> 3, q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

// floor(i / d) for 0 <= i < 2^21 and d >= 1, given inv = 1.f / d.  Exact: (i + 1/2) / d is at least 1/(2d) away from an integer,
// the fp32 error of the product is below (i / d) * 2^-22.  An integer division by a run-time value is ~40 emulated instructions.
__device__ __forceinline__ int idiv_small(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process-global "done" flag leaves a
// second GPU of the same process without it.  One bit per device ordinal; the call is idempotent, so two threads racing
// through the first launch on a device both set it and nothing is lost (no lock, no process-global bool).

typedef std::atomic<unsigned long long> lds_attr_mask;
static inline hipError_t ensure_dynamic_lds(const void* fn, size_t bytes, lds_attr_mask& mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((mask.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && tracked) mask.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}

// host side: are streamed outputs stored write-through?  CABINET_WT_STREAM=0 restores plain stores (A/B timing)
#include <stdlib.h>
static inline int stream_wt() {
    const char* e = getenv("CABINET_WT_STREAM");
    return (e && e[0] == '0') ? 0 : 1;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
