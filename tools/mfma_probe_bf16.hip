// Micro-probe: what does v_mfma_f32_32x32x16_bf16 sustain on this box, on random operands, in the issue patterns the
// split-bf16 kernels use (one wave per SIMD, 256 workgroups)?  ns per MFMA for short (~20 us) and long launches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const u32x4* in, int iters) {
    __shared__ u32x4 lds[48 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) lds[i] = in[i & 1023];
    bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x]), b = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256]);
    bf16x8 a2 = __builtin_bit_cast(bf16x8, in[threadIdx.x + 512]);
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // one dependent chain, register operands
#pragma unroll
            for (int u = 0; u < 48; ++u) c0 = MF((u & 1) ? a : a2, b, c0);
        } else if (MODE == 1) {  // 4 accumulators
#pragma unroll
            for (int u = 0; u < 12; ++u) { c0 = MF(a, b, c0); c1 = MF(a2, b, c1); c2 = MF(a, b, c2); c3 = MF(a2, b, c3); }
        } else if (MODE == 2) {  // chain, A operand re-read from LDS every second MFMA (ds_read_b128, lane-linear)
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const bf16x8 x = __builtin_bit_cast(bf16x8, lds[u * 64 + lane]);
                c0 = MF(x, b, c0);
                c0 = MF(x, __builtin_bit_cast(bf16x8, in[0]), c0);
            }
        } else {  // 16x16x32, 4 accumulators (same FLOP per instruction pair)
#pragma unroll
            for (int u = 0; u < 24; ++u) { d0 = MF16(a, b, d0); d1 = MF16(a2, b, d1); d2 = MF16(a, b, d2); d3 = MF16(a2, b, d3); }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int r = 0; r < 4; ++r) s += d0[r] + d1[r] + d2[r] + d3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; u32x4* in;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&in, 1024 * 16);
    unsigned short h[8192];
    unsigned x = 12345;
    for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3f00 + ((x >> 16) & 0xff) + ((x >> 31) << 15)); }  // random bf16 in +-[0.5, 1)
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"32x32x16 one chain, reg operands", "32x32x16 4 accumulators", "32x32x16 chain, A from LDS b128", "16x16x32 4 accumulators"};
    for (int iters : {8, 1000}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) probe<0><<<256, 256>>>(out, in, iters);
                if (mode == 1) probe<1><<<256, 256>>>(out, in, iters);
                if (mode == 2) probe<2><<<256, 256>>>(out, in, iters);
                if (mode == 3) probe<3><<<256, 256>>>(out, in, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double n_mfma = (mode == 3 ? 96.0 : 48.0) * iters, flop = (mode == 3 ? 16384.0 : 32768.0);
            printf("iters %4d  %-34s %9.1f us  %6.1f ns/MFMA  %7.1f TF/s\n", iters, names[mode], best * 1e3, best * 1e6 / n_mfma,
                   256.0 * 4 * n_mfma * flop / best / 1e9);
        }
    }
    return 0;
}
