// What does a raw buffer load return when (a) voffset alone exceeds num_records, (b) soffset alone does, (c) voffset + soffset does
// but each is in range?  (gfx950; decides how conv3x3_wino.hip expresses zero padding.)   hipcc --offload-arch=gfx950 -O2 tools/buf_oob_probe.hip -o tools/bin/buf_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__global__ void probe(const float* p, unsigned bytes, float* out) {
    const buf_rsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
    auto ld = [&](int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); };
    if (threadIdx.x == 0) {
        out[0] = ld(16, 0);                    // in range: p[4]
        out[1] = ld((int)bytes + 16, 0);       // (a) voffset out of range
        out[2] = ld(16, (int)bytes);           // (b) soffset out of range, voffset in range
        out[3] = ld((int)bytes - 64, 128);     // (c) sum out of range
        out[4] = ld(0x7fffff00, 0);            // huge voffset
        out[5] = ld(-4, 0);                    // negative voffset (0xfffffffc)
        out[6] = ld(-4, 64);                   // negative voffset + soffset that brings the ADDRESS back in range
    }
}
int main() {
    float *d, *o, h[1024], ho[8];
    for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
    hipMalloc(&d, 4096); hipMalloc(&o, 64);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 2048, o);   // the resource covers the first 512 floats of a 1024-float allocation
    hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
    printf("in-range %.0f | voff>=n %.0f | soff>=n %.0f (p[516]=616 if unchecked) | sum>=n %.0f (p[528]=628 if unchecked) | huge %.0f | neg %.0f | neg+soff %.0f (p[15]=115 if address wraps)\n",
           ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6]);
    return 0;
}
