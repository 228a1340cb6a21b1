#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel over the LAST `--last` dispatches of each kernel.
Usage: python tools/pmc_summary.py <counter_collection.csv> [--last 20] [--match bin_fill,bin_reduce] [--out file]"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last", type=int, default=20)
    ap.add_argument("--match", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    pats = [p for p in args.match.split(",") if p]
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> [(dispatch, value)]
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if pats and not any(p in name for p in pats):
                continue
            per[name[:70]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    lines = []
    for k, ctrs in sorted(per.items()):
        for c, vals in sorted(ctrs.items()):
            # a counter may be reported once per dimension (XCC/SE): sum within a dispatch, then average dispatches
            by_disp = collections.defaultdict(float)
            for d, v in vals:
                by_disp[d] += v
            last = [by_disp[d] for d in sorted(by_disp)[-args.last:]]
            lines.append(f"{k},{c},{sum(last) / len(last):.1f},{len(last)}")
    text = "kernel,counter,avg_per_dispatch,dispatches\n" + "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
