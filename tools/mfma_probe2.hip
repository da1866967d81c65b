// Micro-probe (round 4): v_mfma_f32_32x32x2_f32 at TWO waves per SIMD (512-thread workgroups, one per CU) in the issue
// patterns of ffm_bwd_xw_kernel: dependent chains, a workgroup barrier every 128 MFMAs, operands from LDS a block ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
template <int MODE, int T>
__global__ __launch_bounds__(T) void probe(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += T) lds[i] = in[i & 1023];
    float a = in[threadIdx.x], b = in[threadIdx.x + 512];
    f32x16 c0 = {0}, c1 = {0};
    __syncthreads();
    const float* row = lds + (wave * 64 + lane) * 20;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 128 dependent MFMAs, register operands
#pragma unroll
            for (int u = 0; u < 128; ++u) c0 = MF(a, b, c0);
        } else if (MODE == 1) {   // + one workgroup barrier per 128
#pragma unroll
            for (int u = 0; u < 128; ++u) c0 = MF(a, b, c0);
            __syncthreads();
        } else if (MODE == 2) {   // 8 blocks of 16, operands of block n+1 read from LDS (4 x b128) before block n's MFMAs
            float f[2][16];
#pragma unroll
            for (int q = 0; q < 4; ++q) { f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * q); f[0][4*q]=t[0]; f[0][4*q+1]=t[1]; f[0][4*q+2]=t[2]; f[0][4*q+3]=t[3]; }
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { f32x4 t = *reinterpret_cast<const f32x4*>(row + 1280 * ((blk + 1) & 3) + 4 * q); f[(blk+1)&1][4*q]=t[0]; f[(blk+1)&1][4*q+1]=t[1]; f[(blk+1)&1][4*q+2]=t[2]; f[(blk+1)&1][4*q+3]=t[3]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 16; ++s) c0 = MF(f[blk & 1][s], b, c0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        } else {                  // two independent accumulators interleaved, barrier per 128
#pragma unroll
            for (int u = 0; u < 64; ++u) { c0 = MF(a, b, c0); c1 = MF(b, a, c1); }
            __syncthreads();
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    out[blockIdx.x * T + threadIdx.x] = s;
}
template <int MODE, int T>
void run(const char* name, float* out, float* in) {
    const int iters = 100, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, T>), dim3(blocks), dim3(T), 65536, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * (T / 64) * iters * 128 * 4096.0;
    printf("%-64s %3d thr %8.1f us  %7.1f TF/s\n", name, T, ms * 1e3, flops / ms / 1e9);
}
int main() {
    float *out, *in;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&in, 4096 * 4);
    float hin[4096]; for (int i = 0; i < 4096; ++i) hin[i] = (float)((i * 7919) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    run<0, 256>("dependent chain, 1 wave/SIMD", out, in);
    run<0, 512>("dependent chain, 2 waves/SIMD", out, in);
    run<1, 512>("dependent chain + barrier per 128, 2 waves/SIMD", out, in);
    run<2, 512>("8 x 16 blocks, LDS operands a block ahead + barrier, 2 waves/SIMD", out, in);
    run<2, 256>("8 x 16 blocks, LDS operands a block ahead + barrier, 1 wave/SIMD", out, in);
    run<3, 512>("two interleaved accumulators + barrier, 2 waves/SIMD", out, in);
    return 0;
}
