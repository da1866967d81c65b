"""CPU oracle for the CABiNet CAB/FFM hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU (PyTorch fp32 / fp64) restatement of the reference's
algorithm for the hot path named in BASELINE.json.  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the checker / the timed
CPU baseline.  ``cabinet_amd`` never imports it.

Pinning
-------
The reference (dronefreak/CABiNet) holds no golden vectors for this path
(its tests assert shapes / no-NaN only).  The oracle is
therefore pinned against outputs of the reference itself, generated in the
build container by importing ``/root/reference/src`` (a Python reference cannot
travel to the GPU box): ``tests/golden/make_golden.py`` is the generating
script, ``tests/golden/*.npz|*.json`` are the committed vectors, and
``tests/test_oracle_golden.py`` checks every oracle function against them.

Modules
-------
``cab_math``   explicit forward AND hand-derived backward formulas (no autograd)
               for the attention core, BatchNorm, the FFM, PSP and the local
               gate.  Follows reference ``src/models/cab.py`` and
               ``src/models/cabinet.py:132-153``.
``model_ref``  functional (state_dict-driven) restatement of the whole CABiNet
               forward, the OHEM loss and the train step; uses autograd.
               Follows ``src/models/cabinet.py``, ``src/models/mobilenetv3.py``,
               ``src/utils/loss.py`` and ``src/scripts/train.py:429-441``.
"""
