#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run into the files kept
under profiles/: the per-kernel stats table (names shortened) and the hand-written kernels' rows."""
import csv
import glob
import os
import sys


def main(run_dir, out_prefix, title):
    stats = glob.glob(os.path.join(run_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out_prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    ours = [r for r in rows if "cabinet::" in r["Name"]]
    with open(out_prefix + "_summary.md", "w") as f:
        f.write(f"# {title}\n\nsource: `{os.path.basename(stats)}` (rocprofv3 --kernel-trace --stats), "
                f"{len(rows)} distinct kernels, {total / 1e6:.1f} ms of kernel time in the run\n\n")
        f.write("## hand-written kernels (namespace `cabinet::`)\n\n| kernel | calls | avg us | min us | max us | % of run |\n|---|---|---|---|---|---|\n")
        for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
            f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        f.write(f"\nhand-written total: {sum(float(r['TotalDurationNs']) for r in ours) / total * 100:.2f} % of kernel time\n\n")
        f.write("## top 25 kernels overall\n\n| kernel | calls | avg us | % of run |\n|---|---|---|---|\n")
        for r in rows[:25]:
            f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    print("wrote", out_prefix + "_kernel_stats.csv", out_prefix + "_summary.md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 summary")
