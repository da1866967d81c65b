"""Shared gradient-parity rule of the full-model GPU tests.

What fp32 can and cannot promise for the gradients of this network (measured):

* every hand-written operator reproduces its fp64 oracle to 1e-7 .. 5e-6 ON THE TENSORS THE MODEL FEEDS IT -- asserted in the
  test tier itself since round 3: tests/test_gpu_insitu.py captures input and incoming gradient of the CAB, the fused FFM
  and both OHEM heads inside the real step at BASELINE configs 3 and 5 and replays the fp64 oracle on them;
* the gradient of a ReLU network is a DISCONTINUOUS function of the forward rounding: a pre-activation within rounding
  distance of zero lands on the other side and the gradient through that unit toggles.  With a forward relative error d the
  fraction of flipped units is ~0.8 d, so on big maps the gradient moves by ~sqrt(0.8 d) per ReLU layer (4.5e-4 for the CPU
  reference's d = 2.5e-7, 9e-4 for MIOpen's convolutions at d = 1e-6), and on the CAB's small maps ONE flipped unit of N
  positions x 256 channels moves everything behind it by ~1/sqrt(256 N).  The fp32 CPU reference itself is therefore
  1e-3 .. 4e-3 from the fp64 oracle on most gradient tensors of the full-size configurations (profiles/r03_parity_config*.json,
  column ref32_vs_f64; `relu_mask_flips_vs_f64` counts the flipped units either side of the CAB for both implementations).

Rule, per gradient tensor -- no per-configuration floor, no blanket bound:
  pass  if within TOL = 1e-3 of the fp32 reference, or of the fp64 oracle;
  else  its distance from the fp64 oracle must not exceed ALLOW_FACTOR x the fp32 reference's OWN distance from fp64 ON THIS
        TENSOR, measured live ("as close to the truth as the reference's own fp32 arithmetic is on this tensor");
  else  it must be a parameter INSIDE the CAB whose gradient the SAME run proves exact in situ -- the fp64 oracle replayed on
        the tensors the model itself fed the CAB reproduces the model's own gradient of this parameter to EXACT_IN_SITU = 2e-5
        (tests/insitu.py; measured 6e-8 .. 6e-6) -- or a tensor NAMED in tests/golden/grad_allowlist.json[tag]["tensors"]
        with its measured numbers (the cases seen so far, kept as the record; only CAB parameters may be listed,
        tests/test_oracle_golden.py enforces names, count and sizes).  Such a tensor is bounded by the effect of FLIPS
        single-unit ReLU flips on the CAB grid, FLIPS / sqrt(256 x CAB positions): the operator being exact on its own inputs,
        what remains is the incoming gradient's flip noise, which differs from run to run (MIOpen's backward kernels use
        atomics: the same test, same seeds, gave 4.0e-4 on one box and 2.1e-3 on the next for refine.1's BatchNorm bias at
        Small 4x512^2, with the in-situ row at 2.5e-7 both times; config 5: d(cab.y) is 2.1e-3 from fp64 on the GPU and 5.9e-4
        on the CPU at ~1 flipped unit per million either way -- and the reverse, 4.9e-4 vs 2.1e-3, at config 3).
The in-situ-backed widening is GATED: in-situ exactness proves the CAB kernels correct on their own inputs only, so
the same run must also show that what ENTERS the CAB's backward is sound -- every in-situ row of the operators between the loss and
the CAB (ffm.*, head*.dlow and, since round 5, the K11 convolutions conv_out.conv.* / ab.b1.* / ab.conva.*) within TOL of its fp64
replay, or the incoming gradient d(cab.y) itself within the flip bound of the fp64 model's.  A 1 % error injected upstream by a
fused FFM backward or an OHEM kernel then fails the gate instead of hiding behind "the CAB is exact in situ".

Distribution rule (round 5): on top of the per-tensor rule, over all gradient tensors of a full-model test the COUNT of
tensors further than TOL from fp64 must not exceed COUNT_SLACK x the fp32 reference's own count (+ COUNT_FLOOR), and the MEDIAN
distance from fp64 must not exceed MEDIAN_SLACK x the reference's median: "3x the reference on each tensor" cannot turn into
"3x the reference on every tensor".  The reference's own spread these yardsticks rest on is pinned to the imported reference by
tests/golden/reference_spread_large_2x1024.json (tests/test_oracle_golden.py).
Every run writes the full per-tensor table (gpurun_out/parity_<tag>.json; committed copies under profiles/)."""
import json
import os

from conftest import GOLDEN, ROOT

TOL = 1e-3  # north_star: 1e-3 relative (||a-b||/||b|| per tensor), fp32
ALLOW_FACTOR = 3.0
FLIPS = 3.0            # single-unit ReLU flips on the CAB grid the bound makes room for
MAX_NAMED = 8          # named exceptions per configuration
MAX_BOUND = 8e-3       # no tensor is ever allowed further than this from the fp64 oracle (round 5: was 1.5e-2; the largest
                       # distance ever measured on MI355X is 6.9e-3, profiles/r0*_parity_config*.json)
COUNT_SLACK = 1.10     # distribution rule: tensors past TOL from fp64 <= COUNT_SLACK x the fp32 reference's count (+ COUNT_FLOOR)
COUNT_FLOOR = 2
MEDIAN_SLACK = 1.5     # ... and the median distance from fp64 <= MEDIAN_SLACK x the fp32 reference's median
DISTRIBUTION_MIN_TENSORS = 100
EXACT_IN_SITU = 2e-5   # a CAB gradient this close to the fp64 replay on the model's own tensors is "exact in situ"


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


def flip_bound(cfg):
    """FLIPS single-unit ReLU flips on the CAB grid of this configuration."""
    return FLIPS / (256.0 * cfg["cab_positions"]) ** 0.5


GATE_PREFIXES = ("ffm.", "head.", "head16.", "conv_out.conv.", "ab.b1.", "ab.conva.",
                 "conv_out.bn.", "conv_out.conv_out.", "ab.b2.", "ab.b4.")   # round 5: + K12's rows (the classifier tails sit between the loss and the CAB too)


def upstream_gate(insitu_rows, cfg=None, upstream=None):
    """Is what enters the CAB's backward sound in THIS run?  (module docstring)  -> (bool, reason)"""
    rows = {k: r for k, r in (insitu_rows or {}).items() if k.startswith(GATE_PREFIXES) and not r.get("analytic_zero")}
    if rows:
        bad = {k: r["gpu_vs_f64"] for k, r in rows.items() if not r["gpu_vs_f64"] <= TOL}
        return (not bad), (f"in-situ rows past {TOL:g}: {bad}" if bad else f"{len(rows)} upstream in-situ rows within {TOL:g}")
    if upstream is not None and cfg is not None and "d.cab.y" in upstream:
        bound = min(flip_bound(cfg), MAX_BOUND)
        ok = upstream["d.cab.y"] <= bound
        return ok, f"d(cab.y) {upstream['d.cab.y']:.2e} vs fp64 model, flip bound {bound:.2e}"
    return False, "no evidence about the gradient entering the CAB in this run"


def exact_in_situ(insitu_rows, cfg=None, upstream=None):
    """Names of the CAB parameters whose in-place gradient equals the fp64 replay on the model's own tensors (rows of
    tests/insitu.py::operator_table / cab_table of the SAME run) to EXACT_IN_SITU -- provided the run also passes the upstream
    gate (empty set otherwise)."""
    if not upstream_gate(insitu_rows, cfg, upstream)[0]:
        return set()
    return {k for k, r in (insitu_rows or {}).items()
            if k.startswith("ab.a2block.") and not r.get("analytic_zero") and r["gpu_vs_f64"] <= EXACT_IN_SITU}


def tensor_bound(name, row, cfg, exact=()):
    """Largest admissible distance of the GPU gradient from the fp64 oracle for one tensor (module docstring)."""
    flips = flip_bound(cfg) if (name in cfg.get("tensors", {}) or name in exact) else 0.0
    return min(max(TOL, ALLOW_FACTOR * row["ref32_vs_f64"], flips), MAX_BOUND)


def judge_distribution(rows):
    """The distribution rule of the module docstring -> list of violated clauses (empty: pass; fewer than
    DISTRIBUTION_MIN_TENSORS live tensors: not applicable)."""
    live = [r for r in rows.values() if not r["analytic_zero"]]
    if len(live) < DISTRIBUTION_MIN_TENSORS:
        return []
    g, f = sorted(r["gpu_vs_f64"] for r in live), sorted(r["ref32_vs_f64"] for r in live)
    n_g, n_f = sum(x > TOL for x in g), sum(x > TOL for x in f)
    out = []
    if n_g > COUNT_SLACK * n_f + COUNT_FLOOR:
        out.append(f"{n_g} tensors past {TOL:g} from fp64; the fp32 reference has {n_f} (limit {COUNT_SLACK:g} x + {COUNT_FLOOR})")
    if g[len(g) // 2] > MEDIAN_SLACK * f[len(f) // 2] and g[len(g) // 2] > TOL:
        out.append(f"median distance from fp64 {g[len(g) // 2]:.2e} > {MEDIAN_SLACK:g} x the reference's {f[len(f) // 2]:.2e}")
    return out


def judge_gradients(rows, cfg, insitu_rows=None, upstream=None):
    """Apply the rule of the module docstring; returns (failures, tensors that needed more than TOL).  ``insitu_rows``: the
    in-situ operator table of the same run (which CAB gradients are exact on the model's own tensors, and the upstream gate);
    ``upstream``: {"d.cab.y": distance of the incoming CAB gradient from the fp64 model's} where the caller has the fp64 taps.
    Distribution failures are appended as ("<distribution>", {...}) entries."""
    failures, listed = [], []
    exact = exact_in_situ(insitu_rows, cfg, upstream)
    for k, r in rows.items():
        if r["analytic_zero"] or min(r["gpu_vs_ref32"], r["gpu_vs_f64"]) <= TOL:
            continue
        bound = tensor_bound(k, r, cfg, exact)
        if k in exact:
            r["exact_in_situ"] = insitu_rows[k]["gpu_vs_f64"]
        if r["gpu_vs_f64"] <= bound:
            listed.append(k)
            continue
        failures.append((k, dict(r, bound=bound)))
    for clause in judge_distribution(rows):
        failures.append(("<distribution>", dict(clause=clause)))
    return failures, listed
