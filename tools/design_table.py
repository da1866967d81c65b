#!/usr/bin/env python3
"""The per-group table of DESIGN.md section 4 from the committed bench lines (profiles/<tag>_bench_n1.json and
profiles/<tag>_config5_bench_n1.json):  python tools/design_table.py [r06]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
b3 = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_n1.json")))
b5 = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_config5_bench_n1.json")))


def group(b, name):
    for k in b["kernels"]:
        if k["kernel"].split(" ")[0] == name:
            return k
    return None


def cell(k):
    if k is None:
        return "—"
    t = k.get("traffic")
    s = f"{k['ms_per_launch'] * 1e3:.1f} µs; {k['bound']} **{k['frac']:.3f}**"
    if k.get("frac_mfma") is not None:
        s += f" (mfma {k['frac_mfma']:.3f} / hbm {k['frac_hbm']:.3f})"
    if t:
        s += f"; {t / 1e6:.0f} MB = {t / 1e6 / k['algorithmic_mbytes']:.2f}×"
    return s


NAMES = [("cab_attn_fwd", "K1 attention forward"), ("cab_attn_proj_fwd", "K1 + `project_out` in its epilogue (what the block runs where the keys are not split)"),
         ("cab_attn_bwd", "K2 attention backward"), ("ffm_up_fwd", "FFM forward (fused upsample, training)"),
         ("ffm_up_fwd_eval", "FFM forward, eval mode"), ("ffm_up_bwd", "FFM backward"),
         ("ohem_up_pair_fwd", "OHEM heads forward"), ("ohem_up_pair_bwd", "OHEM heads backward"),
         ("cab_local_fwd", "K5 local branch forward"), ("cab_local_bwd", "K5 backward"), ("cab_qkv_fwd", "K6 producers forward (3 launches)"),
         ("cab_qkv_bwd", "K6 backward (6 launches)"), ("conv3x3_conva_fwd", "K11 `conva` forward"), ("conv3x3_conva_bwd", "K11 `conva` backward (data + weight gradient)"),
         ("conv3x3_b1_fwd", "K11 `b1` forward (two pointers)"), ("conv3x3_b1_bwd", "K11 `b1` backward"), ("conv3x3_out_fwd", "K11 `conv_out.conv` forward (128-channel kernel)"),
         ("conv3x3_out_bwd", "K11 `conv_out.conv` backward"), ("bn_cls_out_fwd", "K12 `conv_out` tail forward"), ("bn_cls_out_bwd", "K12 `conv_out` tail backward"),
         ("bn_cls_head_fwd", "K12 `ab` head forward"), ("bn_cls_head_bwd", "K12 `ab` head backward")]
print("| group (`bench.py` name) | config 3 (8×3×1024², 8 classes) | config 5 (2×3×2048×1024, 19 classes) |")
print("|---|---|---|")
print(f"| step: fwd + 2×OHEM-CE + bwd + SGD | **{b3['value']:.1f} images/s, {b3['ms_per_step']:.2f} ms** (without SGD: {b3['fwd_loss_bwd_only']['value']:.1f}) | "
      f"**{b5['value']:.1f} images/s, {b5['ms_per_step']:.2f} ms** (without SGD: {b5['fwd_loss_bwd_only']['value']:.1f}) |")
for b, lab in ((b3, "config 3"), (b5, "config 5")):
    pass
print(f"| forward only, eval mode (`eval_forward`) | {b3['eval_forward']['value']:.0f} images/s, {b3['eval_forward']['ms_per_step']:.2f} ms per batch (eager: {b3['eval_forward']['eager_ms_per_step']:.2f}) | "
      f"{b5['eval_forward']['value']:.0f} images/s, {b5['eval_forward']['ms_per_step']:.2f} ms (eager: {b5['eval_forward']['eager_ms_per_step']:.2f}) |")
print(f"| section-8 groups the step launches, once each (`hot_path_ms_per_step`) | {b3['hot_path_ms_per_step']['value']:.2f} ms = {b3['hot_path_ms_per_step']['share_of_step']:.3f} of the step | "
      f"{b5['hot_path_ms_per_step']['value']:.2f} ms = {b5['hot_path_ms_per_step']['share_of_step']:.3f} |")
for name, label in NAMES:
    print(f"| {label} (`{name}`) | {cell(group(b3, name))} | {cell(group(b5, name))} |")
print(f"| CPU baseline (oracle restatement, {b3['cpu_baseline']['cores']} host cores, B = 2) | {b3['cpu_baseline']['value']:.2f} images/s | {b5['cpu_baseline']['value']:.2f} images/s |")
