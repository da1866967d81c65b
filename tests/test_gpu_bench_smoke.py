"""bench.py's per-kernel roofline loop at a tiny size: every kernel group must launch and report finite numbers
(the default `python bench.py` runs this loop at config 3 after the timed region)."""
import importlib.util
import json
import math
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_kernel_rooflines_run_at_a_small_size():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = bench.kernel_rooflines(2, 256, 256, 8, 2, extra=True)
    names = [r["kernel"].split(" ")[0] for r in rows]
    for want in ("cab_attn_fwd", "cab_attn_bwd", "ffm_up_fwd", "ffm_up_bwd", "bn_act_fwd", "bn_act_bwd", "bn_dwconv_fwd",
                 "bn_dwconv_bwd", "stem_conv_fwd", "stem_conv_wrw", "pwconv_fwd", "pwconv_bwd", "ohem_up_fwd",
                 "ohem_up_bwd", "cab_local_fwd", "cab_local_bwd", "cab_qkv_fwd", "cab_qkv_bwd", "conv3x3_conva_fwd",
                 "conv3x3_conva_bwd", "conv3x3_b1_fwd", "conv3x3_b1_bwd", "conv3x3_out_fwd", "conv3x3_out_bwd"):
        assert want in names, want
    for r in rows:
        assert r["bound"] in ("hbm", "mfma") and r["ms_per_launch"] > 0
        assert all(math.isfinite(r[k]) for k in ("achieved", "peak", "frac", "tflops", "gbytes_per_s"))
        assert r["traffic"] is None  # PMC traffic is only attached at the shape it was measured at (config 3 / config 5)
        assert r["scope"].startswith("SURVEY section 8") != r["kernel"].startswith(("bn_act", "bn_dwconv", "stem_conv", "pwconv"))


def test_kernel_rooflines_rectangular_19_classes():
    """BASELINE config 5's geometry (H != W, H' x W' = 2:1, 19 classes) at a small size: default group set (what the step runs)."""
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = bench.kernel_rooflines(2, 512, 256, 19, 2)
    names = [r["kernel"].split(" ")[0] for r in rows]
    assert "ffm_fwd" not in names and "ohem_up_fwd" not in names  # only behind --all-kernels
    for want in ("cab_attn_fwd", "cab_attn_bwd", "ffm_up_fwd", "ffm_up_bwd", "ohem_up_pair_fwd", "ohem_up_pair_bwd",
                 "cab_local_fwd", "cab_local_bwd", "cab_qkv_fwd", "cab_qkv_bwd", "conv3x3_conva_fwd", "conv3x3_out_bwd"):
        assert want in names, want
    assert not any(n.startswith(("bn_act_", "bn_dwconv_", "stem_conv_", "pwconv_")) for n in names)  # K7-K10: --all-kernels only
    assert all(r["ms_per_launch"] > 0 and math.isfinite(r["frac"]) for r in rows)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(1500)
def test_bench_main_with_eight_ranks_on_one_gpu():
    """The driver's 8-GPU launch line rehearsed on ONE GPU: WORLD_SIZE = 8 under torchrun, every rank on device
    0 over gloo, reduced size.  Eight processes go through rendezvous, per-rank MIOpen database copies, NUMA affinity (all ranks
    take device 0's node), GraphedDDPStep capture and replay with the same bucket order on every rank, and the max-over-ranks
    timing -- no hang, exit 0 on all ranks, one JSON line, global batch 8."""
    env = dict(os.environ, CABINET_DIST_BACKEND="gloo", CABINET_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.pop("MIOPEN_USER_DB_PATH", None)
    env.pop("MIOPEN_CUSTOM_CACHE_DIR", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "2",
           "--batch", "1", "--size", "256", "--mode", "small", "--no-cpu-baseline", "--no-kernel-roofline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1400, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    r = json.loads(lines[-1])
    assert sum(ln.lstrip().startswith("{") for ln in lines) == 1
    assert r["n_gpus"] == 8 and r["steps"] == 2 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp8" and r["config"]["dist_backend"] == "gloo"
    assert "GraphedDDPStep" in r["config"]["host_path"] and len(r["config"]["grad_buckets_mb"]) >= 3
    assert r["value"] > 0 and math.isfinite(r["final_loss"])
    # round 6: what a reader needs to trust an N > 1 line without a profiler -- ranks of the process group, every rank's own time
    # (the line's value uses the maximum), the replay schedule GraphedDDPStep chose and why
    c = r["config"]
    assert c["ranks_in_process_group"] == 8 and len(c["per_rank_ms_per_step"]["all"]) == 8
    assert abs(c["per_rank_ms_per_step"]["max"] - r["ms_per_step"]) < 1e-2 * r["ms_per_step"]
    assert c["ddp_schedule"]["world"] == 8 and c["ddp_schedule"]["decided_by"] == "model" and c["ddp_schedule"]["one_event"] is True
    assert r["library"]["fresh"] is True and len(r["library"]["source_digest"]) == 16


@pytest.mark.timeout(1200)
def test_bench_main_with_two_ranks_on_one_gpu():
    """bench.py's own main() under torchrun with WORLD_SIZE = 2 -- the launch line the driver uses for N > 1 -- rehearsed on a
    single-GPU box: both ranks on device 0, gloo instead of RCCL (RCCL refuses two ranks per device).  What it proves is
    everything around the transport: the torchrun environment, the rank != 0 branch, per-rank MIOpen database copies,
    GraphedDDPStep's segments and buckets, max-over-ranks timing, ONE JSON line as the last line of stdout."""
    env = dict(os.environ, CABINET_DIST_BACKEND="gloo", CABINET_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MIOPEN_USER_DB_PATH", None)  # let bench.py make its per-process copies
    env.pop("MIOPEN_CUSTOM_CACHE_DIR", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "2", "--size", "256", "--no-cpu-baseline", "--no-kernel-roofline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    r = json.loads(lines[-1])  # the JSON line is the LAST line of the job's stdout
    assert sum(ln.lstrip().startswith("{") for ln in lines) == 1
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["dist_backend"] == "gloo" and r["config"]["grad_buckets_mb"]
    assert "GraphedDDPStep" in r["config"]["host_path"]
    assert r["value"] > 0 and math.isfinite(r["final_loss"]) and r["ms_per_step"] > 0
    assert abs(r["value"] - 4 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-2 * r["value"]  # whole-job images / max-over-ranks time
    assert r["config"]["ranks_in_process_group"] == 2 and len(r["config"]["per_rank_ms_per_step"]["all"]) == 2
    assert r["config"]["ddp_schedule"]["world"] == 2 and "exposed_ms_model" in r["config"]["ddp_schedule"]


@pytest.mark.timeout(900)
def test_bench_eval_line_and_training_line_fields():
    """`bench.py --eval` (round 6): the forward-only path evaluate.py:77 consumes -- .eval(), no_grad, BatchNorm on
    running statistics, full-resolution logits -- as a JSON line of its own; and the training line's round-6 fields at a small size:
    `eval_forward`, `hot_path_ms_per_step` (the section-8 groups the step launches, once each), `library` (digest of the sources the
    loaded library was built from), both roofs per matrix group."""
    env = dict(os.environ)
    env.pop("MIOPEN_USER_DB_PATH", None)
    env.pop("MIOPEN_CUSTOM_CACHE_DIR", None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "2", "--size", "256",
            "--no-cpu-baseline", "--kernel-iters", "4"]
    out = subprocess.run(base + ["--eval"], env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    e = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert "forward only" in e["metric"] and e["unit"] == "images/s" and e["value"] > 0 and e["n_gpus"] == 1
    assert e["forward"]["finite"] and e["forward"]["out_shape"] == [2, 8, 256, 256]
    assert e["forward"]["graph_ms"] is None or e["forward"]["graph_vs_eager_rel"] < 1e-4
    assert e["roofline"]["kernel"].startswith("ffm_up_fwd_eval") and "evaluate.py:77" in e["config"]["workload"]
    out = subprocess.run(base, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert r["eval_forward"]["value"] > 0 and r["eval_forward"]["ms_per_step"] > 0
    hp = r["hot_path_ms_per_step"]
    assert 0 < hp["value"] and 0 < hp["share_of_step"] and "conv3x3_out_fwd" in hp["groups"] and "cab_attn_fwd_bf16x6" not in hp["groups"]
    assert not any(g.startswith("ffm_up_fwd_eval") for g in hp["groups"])
    assert r["library"]["fresh"] is True
    ffm = [k for k in r["kernels"] if k["kernel"].startswith(("ffm_up_fwd ", "ffm_up_bwd "))]
    assert len(ffm) == 2 and all("frac_mfma" in k and "frac_hbm" in k and k["frac"] == max(k["frac_mfma"], k["frac_hbm"]) for k in ffm)
