#!/usr/bin/env python3
"""Where do the gradient all-reduces of the data-parallel step run?  From a `rocprofv3 --kernel-trace --output-format csv`
run of `CABINET_FORCE_DDP=1 python3 bench.py ...` (world size 1, RCCL forced; tools/collect_round.sh), take the LAST timed
step and list every RCCL kernel with its interval, the compute kernels that ran during it, and the landmarks of the three
backward segments of cabinet_amd.train.GraphedDDPStep:
    B1 decoder  ... ends with the CAB's backward (cabinet::sd_reduce_kernel of K6 is its last hand-written kernel)
    B2 backbone ... the span of cabinet::dwconv_bwd / bn_dwconv kernels (only `mobile` has depthwise convolutions)
    B3 spatial branch ... ends with cabinet::stem_conv_wrw_kernel (sb.conv1's weight gradient, the last kernel of backward)
usage: python tools/ddp_overlap.py <rocprof dir> <out prefix>   ->  <out>.csv (all kernels of the step), <out>.md (summary)"""
import csv
import glob
import os
import sys


def main(run_dir, out):
    if run_dir.endswith(".csv"):  # a step list written by an earlier run of this tool: rebuild the summary from it
        path = run_dir
        rows = [dict(Kernel_Name=r["kernel"], Start_Timestamp=str(int(round(float(r["start_us"]) * 1e3))),
                     End_Timestamp=str(int(round(float(r["end_us"]) * 1e3))), Queue_Id=r["queue"]) for r in csv.DictReader(open(path))]
        rows.append(dict(Kernel_Name="ohem_up_fwd sentinel", Start_Timestamp=rows[-1]["End_Timestamp"], End_Timestamp=rows[-1]["End_Timestamp"], Queue_Id="-"))
    else:
        path = glob.glob(os.path.join(run_dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"]
    # (at world size 1 RCCL runs its all-reduce as `oneRankReduce<FuncPreMulSum<float>>`: the AVG pre-multiplication, one kernel per bucket)
    is_rccl = lambda r: any(k in name(r) for k in ("nccl", "Nccl", "rccl", "oneRankReduce"))
    steps = [i for i, r in enumerate(rows) if "ohem_up_fwd" in name(r)]          # one forward OHEM launch per step
    first_fwd = [i for i, r in enumerate(rows) if "stem_conv_fwd" in name(r)]    # sb.conv1 forward: first kernel of a step
    lo = max(i for i in first_fwd if i < steps[-1])
    hi = len(rows)
    step = [r for r in rows[lo:hi] if "sentinel" not in name(r)]
    t0 = int(step[0]["Start_Timestamp"])
    us = lambda t: (int(t) - t0) / 1e3
    if not run_dir.endswith(".csv"):
      with open(out + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["start_us", "end_us", "dur_us", "queue", "rccl", "kernel"])
        for r in step:
            w.writerow([f"{us(r['Start_Timestamp']):.1f}", f"{us(r['End_Timestamp']):.1f}",
                        f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}", r.get("Queue_Id", "?"),
                        int(is_rccl(r)), name(r)[:120]])
    comp = [r for r in step if not is_rccl(r)]
    span = lambda rs: (us(min(int(r["Start_Timestamp"]) for r in rs)), us(max(int(r["End_Timestamp"]) for r in rs))) if rs else None
    dw = [r for r in comp if "dwconv_bwd" in name(r) or "bn_dwconv_bwd" in name(r)]
    wrw = [r for r in comp if "stem_conv_wrw" in name(r)]
    k6 = [r for r in comp if "sd_reduce_kernel" in name(r)]
    ohem_b = [r for r in comp if "ohem_up_bwd" in name(r)]
    lines = [f"# RCCL kernels inside the data-parallel step (world size 1, collectives forced), last step of the run\n",
             f"source: `{os.path.basename(path)}`; times in microseconds from the step's first kernel; the full kernel list of the step is `{os.path.basename(out)}.csv`\n",
             "## landmarks of the backward segments\n",
             f"* backward starts (first OHEM backward kernel): {span(ohem_b)[0]:.0f} us" if ohem_b else "* no OHEM backward kernel found",
             f"* B1 decoder: its last hand-written kernel (K6 weight-gradient slab sum) ends at {span(k6)[1]:.0f} us" if k6 else "",
             f"* B2 backbone (`mobile`): depthwise-convolution backward kernels span {span(dw)[0]:.0f} .. {span(dw)[1]:.0f} us" if dw else "",
             f"* B3 spatial branch (`sb`): ends with `stem_conv_wrw_kernel` at {span(wrw)[1]:.0f} us" if wrw else "",
             f"* last compute kernel of the step (optimizer) ends at {us(max(int(r['End_Timestamp']) for r in comp)):.0f} us\n",
             "## RCCL kernels\n",
             "| # | start | end | duration | compute kernels running meanwhile (time inside the interval) | previous compute kernel ended | next compute kernel starts | position in the schedule |",
             "|---|---|---|---|---|---|---|---|"]
    for i, r in enumerate([r for r in step if is_rccl(r)]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy = sum(max(0, min(e, int(c["End_Timestamp"])) - max(s, int(c["Start_Timestamp"]))) for c in comp)
        n_over = sum(1 for c in comp if int(c["End_Timestamp"]) > s and int(c["Start_Timestamp"]) < e)
        seg = "-"
        if dw and us(e) <= span(dw)[0]:
            seg = "B1 | B2 boundary: decoder bucket, issued behind graph B1, graph B2 (backbone backward) enqueued beside it"
        if dw and us(s) < span(dw)[1] and us(e) > span(dw)[0]:
            seg = "inside B2 (backbone backward)"
        if wrw and dw and us(s) >= span(dw)[1] and us(s) < span(wrw)[1]:
            seg = "B2 | B3 boundary / inside B3: backbone bucket beside the spatial branch's backward"
        if wrw and us(s) >= span(wrw)[1]:
            seg = "after the last backward kernel: the spatial branch's own bucket (the only exposed collective)"
        prev_end = max((int(c["End_Timestamp"]) for c in comp if int(c["End_Timestamp"]) <= s), default=s)
        next_start = min((int(c["Start_Timestamp"]) for c in comp if int(c["Start_Timestamp"]) >= e), default=e)
        lines.append(f"| {i} | {us(s):.0f} | {us(e):.0f} | {(e - s) / 1e3:.0f} | {n_over} kernels, {busy / 1e3:.0f} us | "
                     f"{(s - prev_end) / 1e3:.0f} us earlier | {(next_start - e) / 1e3:.0f} us later | {seg} |")
    lines.append("\nAt world size 1 RCCL's all-reduce is a `oneRankReduce` kernel of a few microseconds per bucket (the AVG pre-multiplication): "
                 "this trace shows WHERE in the schedule each bucket's collective is issued -- on RCCL's own stream, behind the graph "
                 "segment that produced the bucket and beside the next segment -- not how long a real 8-rank all-reduce lasts "
                 "(no 8-GPU node was available to the builder; the driver's SCALE run measures that).")
    open(out + ".md", "w").write("\n".join(l for l in lines if l != "") + "\n")
    print("\n".join(lines[-12:]))
    print("wrote", out + ".csv", out + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
