"""cabinet_amd -- MI355X-native CAB / FFM hot path for CABiNet.

* ``cabinet_amd.csrc``       hand-written HIP kernels (gfx950) + the C ABI (include/cabinet_hip.h)
* ``cabinet_amd.functional`` autograd operators bound to the C ABI via ctypes
* ``cabinet_amd.models``     mirror of the reference's ``src.models`` nn.Module API
* ``cabinet_amd.loss``       OHEM cross-entropy used by the train step
* ``cabinet_amd.ddp``        bucketed RCCL gradient all-reduce overlapped with backward
* ``cabinet_amd.train``      the reference's train step (train.py:429-441) as a harness
"""
__version__ = "0.1.0"
