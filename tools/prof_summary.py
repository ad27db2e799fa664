#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats and --pmc runs) into a small text summary.

    python tools/prof_summary.py <rocprof output dir> [<label>]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name[:100]


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/"))
    print(f"# rocprofv3 summary: {label}")
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    for f in stats:
        rows = list(csv.DictReader(open(f)))
        print(f"\n## kernel stats ({os.path.relpath(f, d)})")
        print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>11s} {'pct':>6s}")
        for r in rows[:25]:
            print(f"{short(r['Name']):100s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:11.3f} "
                  f"{float(r['AverageNs'])/1e3:11.2f} {float(r['Percentage']):6.2f}")
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if traces and not stats:
        agg = defaultdict(lambda: [0, 0.0])
        for f in traces:
            for r in csv.DictReader(open(f)):
                a = agg[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        print("\n## kernel trace aggregate")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"{short(k):100s} {c:7d} {t/1e6:11.3f} {t/c/1e3:11.2f}")
    counters = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in counters:
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print(f"\n## counters ({os.path.relpath(f, d)}): mean per dispatch")
        for k in sorted(agg, key=lambda kk: -max(v[1] for v in agg[kk].values()))[:12]:
            items = ", ".join(f"{c}={v[1]/v[0]:.4g} (n={v[0]})" for c, v in sorted(agg[k].items()))
            print(f"{short(k):100s} {items}")


if __name__ == "__main__":
    main()
