"""Shared gradient-parity rule of the full-model GPU tests.

What fp32 can and cannot promise for the gradients of this network (measured):

* every hand-written operator reproduces its fp64 oracle to 1e-7 .. 5e-6 on the tensors the model feeds it
  (tools/diag_*.py, the operator tests);
* the gradient of a ReLU network is a DISCONTINUOUS function of the forward rounding: a pre-activation within rounding
  distance of zero lands on the other side and the gradient through that unit toggles.  With a forward relative error d the
  fraction of flipped units is ~0.8 d, so on big maps the gradient moves by ~sqrt(0.8 d) per ReLU layer (4.5e-4 for the CPU
  reference's d = 2.5e-7, 9e-4 for MIOpen's convolutions at d = 1e-6), and on the CAB's small maps ONE flipped unit of N
  positions x 256 channels moves everything upstream by ~1/sqrt(256 N) (2e-3 at N = 1024, found unit by unit with
  tools/diag_cab_internal.py).  The fp32 CPU reference itself is therefore 1e-3 .. 4e-3 from the fp64 oracle on most
  gradient tensors of the full-size configurations (profiles/r02_parity_config*.json).

Rule, per gradient tensor (no blanket bound):
  pass  if within TOL = 1e-3 of the fp32 reference, or of the fp64 oracle;
  else  the distance from the fp64 oracle must not exceed
            max( ALLOW_FACTOR x the fp32 reference's own distance from fp64 on this tensor (measured live),
                 ALLOW_FACTOR x the committed 90th percentile of that distance over the configuration's tensors,
                 FLIPS / sqrt(256 x CAB positions)   -- three single-unit ReLU flips on the CAB grid )
  i.e. "as close to the truth as the reference's own fp32 arithmetic is".  The two committed numbers per configuration
  live in tests/golden/grad_allowlist.json (measured; tests/test_oracle_golden.py limits what may be written there), and
  every run writes the full per-tensor table (gpurun_out/parity_<tag>.json)."""
import json
import os

from conftest import GOLDEN, ROOT

TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32
ALLOW_FACTOR = 3.0
FLIPS = 3.0            # single-unit ReLU flips on the CAB grid the bound makes room for
MAX_REF_P90 = 5e-3     # a configuration whose reference is further than this from fp64 is not a parity test
MAX_BOUND = 1.5e-2     # no tensor is ever allowed further than this from the fp64 oracle


def rel_pair(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm()), float(b.norm())


def host_memory_gb():
    """Memory this process may use: min(MemAvailable, cgroup limit - cgroup usage)."""
    avail = None
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            avail = int(line.split()[1]) / 2 ** 20
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, (int(lim) - used) / 2 ** 30)
    except (OSError, ValueError):
        pass
    return avail


def load_allowlist():
    with open(os.path.join(GOLDEN, "grad_allowlist.json")) as f:
        return json.load(f)


def write_table(name, payload):
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def gradient_table(net, ref32, ref64):
    """Per tensor: relative distance GPU-fp32ref, GPU-fp64, fp32ref-fp64, and ||g||."""
    rows = {}
    for k, p in net.named_parameters():
        if k not in ref64:
            assert p.grad is None or not p.requires_grad, k
            continue
        e32, d32 = rel_pair(p.grad, ref32[k])
        e64, d64 = rel_pair(p.grad, ref64[k])
        ec, _ = rel_pair(ref32[k], ref64[k])
        floor = 1e-7 * p.numel() ** 0.5  # analytically-zero gradients (a bias in front of a batch-statistics BN)
        rows[k] = dict(numel=p.numel(), norm=d64, gpu_vs_ref32=e32 / max(d32, 1e-300), gpu_vs_f64=e64 / max(d64, 1e-300),
                       ref32_vs_f64=ec / max(d64, 1e-300), analytic_zero=bool(e64 <= floor and d64 <= floor))
    return rows


def tensor_bound(row, cfg):
    """Largest admissible distance of the GPU gradient from the fp64 oracle for one tensor (module docstring)."""
    floor = cfg.get("ref32_vs_f64_p90", 0.0)
    flip = FLIPS / (256.0 * cfg["cab_positions"]) ** 0.5 if cfg.get("cab_positions") else 0.0
    return min(max(TOL, ALLOW_FACTOR * max(row["ref32_vs_f64"], floor), flip), MAX_BOUND)


def judge_gradients(rows, cfg):
    """Apply the rule of the module docstring; returns (failures, tensors that needed more than TOL)."""
    failures, listed = [], []
    for k, r in rows.items():
        if r["analytic_zero"] or min(r["gpu_vs_ref32"], r["gpu_vs_f64"]) <= TOL:
            continue
        if r["gpu_vs_f64"] <= tensor_bound(r, cfg):
            listed.append(k)
            continue
        failures.append((k, dict(r, bound=tensor_bound(r, cfg))))
    return failures, listed
