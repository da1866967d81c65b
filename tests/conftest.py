import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# MIOpen picks convolution solvers from its per-user find database; on a box that ran profiled or timed workloads before the
# tests, that database holds whatever those runs recorded (a full-suite run right after the rocprofv3 / PMC passes of
# tools/collect_round.sh failed two model-level parity cases that pass on a fresh box, and config 5 ran at 23.0 instead of
# 17.1 ms/step).  The session therefore works on a private copy of the committed database, exactly as bench.py does, so the
# backbone's solver choices -- and with them the summation order the gradient-parity tables see -- do not depend on box history.
_MIOPEN_DB = os.path.join(ROOT, "cabinet_amd", "miopen_db")
if os.path.isdir(_MIOPEN_DB) and "MIOPEN_USER_DB_PATH" not in os.environ:
    import atexit
    import shutil
    import tempfile

    try:
        _tmp = tempfile.mkdtemp(prefix="cabinet_miopen_tests_")
        _db = os.path.join(_tmp, "db")
        shutil.copytree(_MIOPEN_DB, _db)
        atexit.register(shutil.rmtree, _tmp, ignore_errors=True)
        os.environ["MIOPEN_USER_DB_PATH"] = _db
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(_db, "cache"))
    except OSError:
        pass


# Round 6: a convolution whose shape is NOT in that database (every test shape that is not a bench shape) made MIOpen's default find
# mode time the applicable solvers on the spot and keep the fastest -- so which solver ran, and with it the bits of the spatial
# branch's forward, depended on what the process (or the box) had done before: tools/diag_order_dependence.py gives exactly two
# sets of digests for one step, {fresh process, after a Small step} and {after large allocations, after other FFM / K11 calls}, and
# `test_model_vs_oracle_logits_and_grads[large-2-512-19]` passed in one test order and failed in another on the same box
# (profiles/r06_order_dependence.txt).  FAST find mode takes a database hit when there is one and MIOpen's static heuristics when
# there is none: no timing, no history.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        # Round 6: the GPU parity session pins MIOpen to its deterministic solvers.  tools/diag_step_determinism.py
        # (profiles/r06_step_determinism.txt): with the default solver choice the FORWARD of the spatial branch is not reproducible
        # run to run (six repeats of one step from one state_dict: six different digests of the final logits, the attention
        # branch's logits identical) -- a split-K convolution solver with atomics --, so which ReLU units flip, and with them every
        # gradient upstream (216 of 221 tensors move; gamma.grad lands up to 1.9e-3 from the reference's value), depends on the
        # run.  With torch.backends.cudnn.deterministic (MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC on every descriptor) six repeats are
        # bit-identical in every logit and every gradient: the verdict of the parity suite no longer depends on box history.
        # (bench.py keeps MIOpen's default choice: the timed step is what a user runs.)
        torch.backends.cudnn.deterministic = True
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def rel_err(a, b):
    """||a-b|| / ||b|| per tensor (trap 5)."""
    import torch

    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = b.norm()
    return float((a - b).norm() / den) if den > 0 else float((a - b).norm())


def assert_close(a, b, tol, name="", atol=1e-6):
    """||a-b|| <= tol*||b|| + atol*sqrt(numel): relative per tensor, with an absolute floor
    so that exactly-zero references (e.g. dq when n == 1) stay well posed."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err, den = float((a - b).norm()), float(b.norm())
    assert err <= tol * den + atol * (b.numel() ** 0.5), f"{name}: |a-b|={err:.3e} |b|={den:.3e} rel={err / max(den, 1e-300):.3e} tol={tol}"


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
