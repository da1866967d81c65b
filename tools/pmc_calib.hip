// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the hot kernels use
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads 1/2 for 16 B/lane streams; other widths uncalibrated).
// Each kernel moves exactly N*4 bytes of a buffer larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_dword(const float* __restrict__ p, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read_dwordx4(const float4* __restrict__ p, float* out, size_t n4) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void write_dword(float* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}
__global__ void write_dwordx4(float4* __restrict__ p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t n = (size_t)1 << 28;  // 1 GiB of floats
    float *a, *o;
    hipMalloc(&a, n * 4);
    hipMalloc(&o, 4);
    hipMemset(a, 0, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        read_dword<<<2048, 256>>>(a, o, n);
        read_dwordx4<<<2048, 256>>>((const float4*)a, o, n / 4);
        write_dword<<<2048, 256>>>(a, n);
        write_dwordx4<<<2048, 256>>>((float4*)a, n / 4);
    }
    hipDeviceSynchronize();
    printf("calib done: each kernel moved %zu bytes\n", n * 4);
    return 0;
}
