"""north_star: "keeping the src/models nn.Module API surface so the existing train.py/evaluate.py drive it unchanged".

Build-container only (the reference's Python never travels to the GPU box: skipped when /root/reference is absent).
A child process with ``PYTHONPATH=<repo>:<repo>/tests/stubs:<reference>`` -- INTEGRATION.md path A, plus stand-ins for the
three third-party packages this image lacks -- runs ``tests/reference_driver.py``: the reference's UNMODIFIED
``src/scripts/train.py::train_and_evaluate`` (data loaders, ``Optimizer`` warm-up + poly schedule, ``ModelEMA``,
``EarlyStopping``, two ``OhemCELoss`` with class weights, the AMP + GradScaler + clip step, validation, per-epoch
``MscEvalV0`` on the EMA copy, ``_save_checkpoint``; then a resumed run through ``_load_checkpoint``) and
``src/scripts/evaluate.py::evaluate_checkpoint`` on this repo's ``CABiNet``."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
REF = Path(os.environ.get("CABINET_REFERENCE_ROOT", "/root/reference"))

pytestmark = pytest.mark.skipif(not (REF / "src" / "scripts" / "train.py").is_file(),
                                reason="reference checkout not present (GPU box): its Python never travels")


def _env():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([str(REPO), str(REPO / "tests" / "stubs"), str(REF)])
    env["PYTHONDONTWRITEBYTECODE"] = "1"  # never write into the read-only reference tree
    env["CABINET_REFERENCE_ROOT"] = str(REF)
    env["CABINET_DRIVER_DEVICE"] = "cpu"
    return env


def test_shim_does_not_shadow_the_reference_package():
    """Every module reference train.py:18-31 imports resolves: the model + loss to this repo, the rest to the reference."""
    code = (
        "import json, src, src.models.cabinet, src.models.cab, src.models.constants, src.models.mobilenetv3, src.utils.loss\n"
        "import src.utils.optimizer, src.utils.ema, src.utils.early_stopping, src.utils.exceptions, src.utils.logger\n"
        "import src.utils.class_weights, src.datasets.registry, src.scripts.evaluate, src.scripts.train, src.models.layers\n"
        "import sys\n"
        "print(json.dumps({m: sys.modules[m].__file__ for m in sorted(sys.modules) if m.startswith('src.') and "
        "getattr(sys.modules[m], '__file__', None)}))\n")
    out = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-3000:]
    files = json.loads(out.stdout.strip().splitlines()[-1])
    mine = {"src.models.cabinet", "src.models.cab", "src.models.constants", "src.models.mobilenetv3", "src.utils.loss",
            "src.models", "src.utils"}
    for mod, f in files.items():
        root = REPO if mod in mine else REF
        assert Path(f).is_relative_to(root), f"{mod} resolved to {f}, expected under {root}"
    for mod in ("src.utils.optimizer", "src.utils.ema", "src.scripts.train", "src.scripts.evaluate", "src.datasets.registry",
                "src.models.layers"):
        assert mod in files


def test_reference_train_and_evaluate_drive_this_model(tmp_path):
    out = subprocess.run([sys.executable, str(REPO / "tests" / "reference_driver.py"), str(tmp_path)], env=_env(),
                         capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("REPORT ")][-1]
    r = json.loads(line[len("REPORT "):])
    assert r["ok"]
    # who ran: the reference's scripts / utilities, this repo's model and loss
    assert r["src_path"][0] == str(REPO / "src") and r["src_path"][1] == str(REF / "src")
    assert Path(r["train_module_file"]).is_relative_to(REF) and Path(r["eval_module_file"]).is_relative_to(REF)
    assert Path(r["optimizer_module_file"]).is_relative_to(REF) and Path(r["ema_module_file"]).is_relative_to(REF)
    assert Path(r["model_module_file"]).is_relative_to(REPO)
    assert r["model_is_repo"] and r["eval_model_is_repo"] and r["loss_is_repo"]
    # 2 epochs x 2 batches, accum 1: four optimizer steps (none skipped by the scaler), four EMA folds; checkpoint layout of
    # reference train.py:70-85; weights_only=True round trip
    assert r["checkpoint_keys"] == sorted(["epoch", "model_state", "optimizer_state", "optimizer_it", "scaler_state",
                                           "best_miou", "best_loss", "ema_state", "ema_updates",
                                           "early_stop_best_fitness", "early_stop_best_epoch"])
    assert r["optimizer_it_after_run1"] == 4 and r["ema_updates_after_run1"] == 4 and r["epoch_after_run1"] == 1
    assert r["final_state_dict_loads_strict"] and r["ckpt_model_keys_equal_state_dict"]
    assert r["weights_moved"] and r["all_finite"]
    assert r["gamma_after_run1"] != 0.0  # the CAB's scale received a gradient through the attention path
    # resume (train.py:87-123): continues at epoch 2 with the restored step / EMA counters
    assert r["epoch_after_resume"] == 2 and r["optimizer_it_after_resume"] == 6 and r["ema_updates_after_resume"] == 6
    # MscEvalV0 (evaluate.py:74-148,195-253) sliding window on 96x80 frames with a 64 crop: every pixel counted once
    assert r["const_eval_pixels"] == 2 * 96 * 80 and r["const_eval_label_column_only"]
    assert r["const_eval_accuracy_matches_hist"]
