#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection.csv values per kernel and counter."""
import collections
import csv
import glob
import os
import sys


def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main(root):
    merged = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(d):
            continue
        for kern, cs in load(d).items():
            for c, vals in cs.items():
                merged[kern][c] = (sum(vals) / len(vals), len(vals))
    for kern in sorted(merged):
        short = kern.split("(")[0][-70:]
        print(short)
        for c, (v, n) in sorted(merged[kern].items()):
            print(f"    {c:34s} {v:16.1f}  (n={n})")
    return merged


if __name__ == "__main__":
    m = main(sys.argv[1])
    if len(sys.argv) > 2:  # also write / update a JSON file: {kernel name up to "(": {counter: per-launch average}}
        import json

        out = json.load(open(sys.argv[2])) if os.path.exists(sys.argv[2]) else {}
        for kern, cs in m.items():
            out.setdefault(kern.split("(")[0].replace("void ", ""), {}).update({c: v for c, (v, n) in cs.items()})
        json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
        print("updated", sys.argv[2])
