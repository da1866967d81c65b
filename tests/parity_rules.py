"""Shared gradient-parity rule of the full-model GPU tests (see tests/test_gpu_fullsize.py for the statement)."""
import json
import os

from conftest import GOLDEN, ROOT

TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32
ALLOW_FACTOR = 3.0  # a listed tensor may be this many times further from fp64 than the fp32 reference was measured to be
ALLOW_MIN_REF_ERR = 2.5e-4  # a tensor whose fp32 reference is closer to fp64 than this has no business on the list
EXPLICIT_BOUND_CAP = 1e-2  # entries with an explicit "bound" (documented single-ReLU-flip events etc.) never exceed this
MAX_EXPLICIT_ENTRIES = 10


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


def judge_gradients(rows, allow):
    """Apply the rule of the module docstring; returns (failures, tensors that needed the allow-list)."""
    failures, listed = [], []
    for k, r in rows.items():
        if r["analytic_zero"] or min(r["gpu_vs_ref32"], r["gpu_vs_f64"]) <= TOL:
            continue
        entry = allow.get(k)
        if entry is not None and r["gpu_vs_f64"] <= entry.get("bound", ALLOW_FACTOR * entry["ref32_vs_f64"]):
            listed.append(k)
            continue
        failures.append((k, r))
    return failures, listed


