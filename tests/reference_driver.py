"""Runs the REFERENCE's own, unmodified entry points against this repo's ``src.models`` -- north_star's "the existing
train.py/evaluate.py drive it unchanged" (reference src/scripts/train.py:18-31,258-640, src/scripts/evaluate.py:261-369).

Started by ``tests/test_reference_drives_it.py`` as a child process with
``PYTHONPATH=<repo>:<repo>/tests/stubs:<reference>``:  ``src.models.*`` / ``src.utils.loss`` resolve to this repo,
``src.scripts.*``, ``src.datasets.*``, ``src.utils.{optimizer,ema,early_stopping,class_weights,logger,exceptions}``
to the reference (``src/__init__.py`` extends the package path), and ``hydra`` / ``omegaconf`` / ``torchvision`` to the
stand-ins under ``tests/stubs`` (absent from this image).  The configuration is the reference's own
``configs/train.yaml`` + ``configs/model/mobilenetv3_small.yaml`` with sizes cut down to a toy data set registered in the
reference's ``DATASET_REGISTRY``; nothing of the reference is edited or copied.  Prints one JSON line."""
import json
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch
import yaml
from torch.utils.data import Dataset

REF = Path(os.environ["CABINET_REFERENCE_ROOT"])
OUT = Path(sys.argv[1])
DEVICE = os.environ.get("CABINET_DRIVER_DEVICE", "cpu")

import src  # noqa: E402
import src.models.cabinet as model_mod  # noqa: E402
import src.utils.loss as loss_mod  # noqa: E402
import src.scripts.train as ref_train  # noqa: E402  (the reference's file)
import src.scripts.evaluate as ref_eval  # noqa: E402  (the reference's file)
from omegaconf import OmegaConf  # noqa: E402
from src.datasets.registry import DATASET_KWARGS_BUILDERS, DATASET_REGISTRY  # noqa: E402

import cabinet_amd.models.cabinet  # noqa: E402
import cabinet_amd.loss  # noqa: E402

report = {
    "src_path": [str(p) for p in src.__path__],
    "model_module_file": model_mod.__file__,
    "train_module_file": ref_train.__file__,
    "eval_module_file": ref_eval.__file__,
    "optimizer_module_file": sys.modules["src.utils.optimizer"].__file__,
    "ema_module_file": sys.modules["src.utils.ema"].__file__,
    "model_is_repo": ref_train.CABiNet is cabinet_amd.models.cabinet.CABiNet,
    "eval_model_is_repo": ref_eval.CABiNet is cabinet_amd.models.cabinet.CABiNet,
    "loss_is_repo": ref_train.OhemCELoss is cabinet_amd.loss.OhemCELoss and loss_mod.OhemCELoss is cabinet_amd.loss.OhemCELoss,
}

N_CLASSES = 4
CROP = [64, 64]


class ToySeg(Dataset):
    """Images whose label is a function of the pixel's quadrant colour: train = random crops of CROP, val = frames
    larger than the crop (the evaluator's sliding window then takes several chips, evaluate.py:117-136)."""

    def __init__(self, ignore_lb, rootpth, cropsize, augmentation=None, mode="train"):
        self.mode = mode
        self.n = 4 if mode == "train" else 2
        self.size = tuple(cropsize) if mode == "train" else (96, 80)
        self.ignore_lb = ignore_lb

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(1000 * (self.mode == "train") + int(i))
        h, w = self.size
        lb = torch.zeros(h, w, dtype=torch.int64)
        lb[: h // 2, w // 2:] = 1
        lb[h // 2:, : w // 2] = 2
        lb[h // 2:, w // 2:] = 3
        lb = torch.roll(lb, shifts=(int(i) * 7, int(i) * 5), dims=(0, 1))
        im = torch.randn(3, h, w, generator=g) * 0.3
        for c in range(3):
            im[c] += (lb == c).float() * 1.5
        lb[0, :3] = self.ignore_lb  # a few ignored pixels
        return im, lb  # (H, W) int64 class ids, as reference src/datasets/uavid.py:248-251 yields them


DATASET_REGISTRY["toy"] = ToySeg
DATASET_KWARGS_BUILDERS["toy"] = lambda cfg, ignore_idx, cropsize: dict(
    ignore_lb=ignore_idx, rootpth=cfg.dataset.dataset_path, cropsize=cropsize, augmentation=cfg.dataset.get("augmentation"))

train_yaml = yaml.safe_load((REF / "configs" / "train.yaml").read_text())
model_yaml = yaml.safe_load((REF / "configs" / "model" / "mobilenetv3_small.yaml").read_text())
tc = dict(train_yaml["training_config"])
tc.update(batch_size=2, num_workers=0, epochs=2, accum_steps=1, warmup_steps=2, optimizer_lr_start=1e-2,
          experiments_path=str(OUT / "exp"), patience=5, ema_tau=4, resume=False, max_iterations=None)
vc = dict(train_yaml["validation_config"])
vc.update(batch_size=1, num_workers=0, eval_scales=[0.75, 1.0], flip=True, results_path=str(OUT / "exp" / "results"))
cfg = OmegaConf.create({
    "model": model_yaml,
    "dataset": {"name": "toy", "num_classes": N_CLASSES, "cropsize": CROP, "dataset_path": "", "ignore_idx": 255, "seed": 0},
    "training_config": tc,
    "validation_config": vc,
})

if DEVICE == "cpu":
    torch.cuda.is_available = lambda: False  # the reference picks its device from this (train.py:294)

# ---- 1. the reference's whole training entry point: loaders, model, EMA, OHEM x2, Optimizer (warm-up + poly), AMP +
# GradScaler + clip step, validation, per-epoch MscEvalV0 on the EMA copy, checkpoints, final multi-scale evaluation
ref_train.train_and_evaluate(cfg)

exp = OUT / "exp"
ckpt = torch.load(exp / "checkpoint_last.pth", map_location="cpu", weights_only=True)
final_sd = torch.load(exp / tc["model_save_name"], map_location="cpu", weights_only=True)
report["checkpoint_keys"] = sorted(ckpt.keys())
report["optimizer_it_after_run1"] = int(ckpt["optimizer_it"])
report["ema_updates_after_run1"] = int(ckpt["ema_updates"])
report["epoch_after_run1"] = int(ckpt["epoch"])
fresh = model_mod.CABiNet(n_classes=N_CLASSES, cfgs=model_yaml["cfgs"], mode="small")
report["final_state_dict_loads_strict"] = not any(fresh.load_state_dict(final_sd, strict=True))
report["ckpt_model_keys_equal_state_dict"] = sorted(ckpt["model_state"].keys()) == sorted(fresh.state_dict().keys())
report["weights_moved"] = float((ckpt["model_state"]["ffm.convblk.conv.weight"]
                                 - model_mod.CABiNet(n_classes=N_CLASSES, cfgs=model_yaml["cfgs"], mode="small")
                                 .state_dict()["ffm.convblk.conv.weight"]).abs().max()) > 0
report["all_finite"] = all(bool(torch.isfinite(v).all()) for v in ckpt["model_state"].values() if v.dtype.is_floating_point)
report["gamma_after_run1"] = float(ckpt["model_state"]["ab.a2block.gamma"])

# ---- 2. resume: the reference's _load_checkpoint path (train.py:87-123,393-404), one more epoch
cfg.training_config.resume = True
cfg.training_config.epochs = 3
ref_train.train_and_evaluate(cfg)
ckpt2 = torch.load(exp / "checkpoint_last.pth", map_location="cpu", weights_only=True)
report["epoch_after_resume"] = int(ckpt2["epoch"])
report["optimizer_it_after_resume"] = int(ckpt2["optimizer_it"])
report["ema_updates_after_resume"] = int(ckpt2["ema_updates"])

# ---- 3. the reference's evaluation entry point on the saved EMA weights (evaluate.py:261-369)
ecfg = OmegaConf.create({
    "model": model_yaml,
    "dataset": dict(cfg.dataset),
    "checkpoint_path": str(exp / "checkpoint_last.pth"),
    "split": "val",
    "validation_config": {"batch_size": 1, "num_workers": 0, "eval_scales": [1.0], "flip": False},
    "device": DEVICE,
})
ref_eval.evaluate_checkpoint(ecfg)

# ---- 4. the same evaluator object on a constant-label set: the metric is known in closed form
net = model_mod.CABiNet(n_classes=N_CLASSES, cfgs=model_yaml["cfgs"], mode="small")
net.load_state_dict(ckpt2["model_state"])
net.to(DEVICE)


class Const(Dataset):
    def __len__(self):
        return 2

    def __getitem__(self, i):
        return torch.randn(3, 96, 80, generator=torch.Generator().manual_seed(i)), torch.full((96, 80), 1, dtype=torch.int64)


res = ref_eval.MscEvalV0(net, torch.utils.data.DataLoader(Const(), batch_size=1), N_CLASSES, cropsize=64,
                         device=torch.device(DEVICE))()
hist = np.asarray(res["confusion_matrix"])
report["const_eval_pixels"] = float(hist.sum())
report["const_eval_label_column_only"] = bool(hist.sum() == hist[:, 1].sum())
report["const_eval_accuracy_matches_hist"] = bool(math.isclose(float(res["accuracy"]), float(hist[1, 1] / hist.sum())))
report["ok"] = True
print("REPORT " + json.dumps(report))
