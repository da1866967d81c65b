// Micro-probe: what does v_mfma_f32_32x32x2_f32 sustain on this box in the issue patterns K1 uses?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters, unsigned long long* cyc) {
    __shared__ float lds[4 * 128 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, h = lane >> 5;
    float* my = lds + wave * 128 * 33;
    for (int i = lane; i < 128 * 33; i += 64) my[i] = in[i & 1023];
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 4 independent accumulators, register operands
#pragma unroll
            for (int u = 0; u < 16; ++u) { c0 = MF(a, b, c0); c1 = MF(a, b, c1); c2 = MF(a, b, c2); c3 = MF(a, b, c3); }
        } else if (MODE == 1) {  // one dependent chain
#pragma unroll
            for (int u = 0; u < 64; ++u) c0 = MF(a, b, c0);
        } else if (MODE == 2) {  // 4 accumulators, A operand from LDS (stride-33 column read) each MFMA
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
                c0 = MF(my[(0 * 32 + li) * 33 + key], b, c0);
                c1 = MF(my[(1 * 32 + li) * 33 + key], b, c1);
                c2 = MF(my[(2 * 32 + li) * 33 + key], b, c2);
                c3 = MF(my[(3 * 32 + li) * 33 + key], b, c3);
            }
        } else {  // dependent chain + one global (buffer) load per MFMA, results consumed next iteration
#pragma unroll
            for (int u = 0; u < 64; ++u) c0 = MF(a, b, c0);
            a += in[(it * 64 + lane) & 1023];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float *out, *in; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&cyc, 64);
    hipMemset(in, 0, 4096 * 4);
    float hin[4096]; for (int i = 0; i < 4096; ++i) hin[i] = (float)((i * 7919) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    const int iters = 200, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"4 indep acc, reg operands", "1 dependent chain", "4 acc, A from LDS stride-33", "chain + global load/iter"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) probe<0><<<blocks, 256>>>(out, in, iters, cyc);
            if (mode == 1) probe<1><<<blocks, 256>>>(out, in, iters, cyc);
            if (mode == 2) probe<2><<<blocks, 256>>>(out, in, iters, cyc);
            if (mode == 3) probe<3><<<blocks, 256>>>(out, in, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[4]; hipMemcpy(hc, cyc, 32, hipMemcpyDeviceToHost);
            double flops = (double)blocks * 4 * iters * 64 * 4096.0;
            if (rep == 2) printf("%-32s %8.1f us  %7.1f TF/s  %6.1f memtime-ticks/MFMA\n", names[mode], ms * 1e3, flops / ms / 1e9,
                                 (double)hc[mode] / (iters * 64.0));
        }
    }
    return 0;
}
