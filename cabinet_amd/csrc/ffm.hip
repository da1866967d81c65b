// placeholder until K3/K4 land
#include "common.hpp"
namespace cabinet {
struct FfmShape { int B, Cs, Cc, Co, Cm, H, W; };
size_t ffm_fwd_workspace(const FfmShape&) { return 0; }
size_t ffm_bwd_workspace(const FfmShape&) { return 0; }
hipError_t ffm_fwd_run(const FfmShape&, const float*, const float*, const float*, const float*, const float*, float*, float*, const float*, const float*, int, float, float, float*, float*, float*, float*, float*, float*, void*, hipStream_t) { return hipErrorNotSupported; }
hipError_t ffm_bwd_run(const FfmShape&, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, float*, float*, float*, float*, float*, float*, float*, void*, hipStream_t) { return hipErrorNotSupported; }
}  // namespace cabinet
