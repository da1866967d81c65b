"""cabinet_amd -- MI355X-native CAB / FFM hot path for CABiNet.

* ``cabinet_amd.csrc``       hand-written HIP kernels (gfx950) + the C ABI (include/cabinet_hip.h)
* ``cabinet_amd.functional`` autograd operators bound to the C ABI via ctypes
* ``cabinet_amd.models``     mirror of the reference's ``src.models`` nn.Module API
* ``cabinet_amd.loss``       OHEM cross-entropy used by the train step
* ``cabinet_amd.ddp``        bucketed RCCL gradient all-reduce overlapped with backward
* ``cabinet_amd.train``      the reference's train step (train.py:429-441) as a harness
"""
import os as _os

__version__ = "0.1.0"

# Stock-library workaround (outside the hot path): MIOpen's igemm_bwd_gtcx35_nhwc_fp32 backward-data kernels
# read past the end of an operand for some of the backbone's convolutions; when that allocation ends a mapped
# segment the GPU raises a VM fault and ROCr aborts the process (rocgdb trace: profiles/r01_miopen_igemm_bwd_fault.txt).
# Whether it strikes depends on the caching allocator's layout.  MIOpen reads this switch on first use and falls
# back to its next solver; measured cost on the config-3 step: none (115.9 vs 116.2 images/s, within noise).
_os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")
