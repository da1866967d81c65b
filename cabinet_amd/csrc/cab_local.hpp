// Argument block of the K5 kernels (cab_local.hip), shared with the C-ABI layer (capi.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cabinet {

struct LocalStage {
    const float* w;      // (C,9) depthwise 3x3 weights
    const float* bn_w;   // (C)
    const float* bn_b;   // (C)
    float* run_mean;     // (C)   forward only (updated in training mode)
    float* run_var;      // (C)
    float* dw;           // (C,9) backward only
    float* dbn_w;        // (C)
    float* dbn_b;        // (C)
};

struct LocalArgs {
    const float* x;        // (B,C,H,W)
    const float* glob;     // (B,C,H,W) or nullptr
    const float* gamma;    // device scalar or nullptr
    LocalStage st[3];
    int B, C, H, W;
    int training;
    float momentum, eps;
    float* out;            // fwd: (B,C,H,W)
    float* save_mean;      // (3,C)   fwd: written, bwd: read
    float* save_invstd;    // (3,C)
    const float* dout;     // bwd
    float* dx;             // bwd
    float* dglob;          // bwd, nullable
    float* dgamma_part;    // bwd, (C) partial sums of <dout, glob>, nullable
};

// resident form (cab_local.hip): one workgroup per channel, B*H*W <= 8192
bool local_shape_supported(int B, int H, int W);
hipError_t cab_local_fwd_run(const LocalArgs& a, hipStream_t stream);
hipError_t cab_local_bwd_run(const LocalArgs& a, hipStream_t stream);
// tiled form (cab_local_tiled.hip): any B*H*W, B*nT workgroups per channel, two-phase BatchNorm reductions, workspace
bool local_tiled_supported(int B, int H, int W);
size_t local_tiled_fwd_workspace(int B, int C, int H, int W);
size_t local_tiled_bwd_workspace(int B, int C, int H, int W);
hipError_t cab_local_tiled_fwd_run(const LocalArgs& a, void* ws, hipStream_t stream);
hipError_t cab_local_tiled_bwd_run(const LocalArgs& a, void* ws, hipStream_t stream);

}  // namespace cabinet
