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


GEMM_SOURCES = ("gemm_tile.h", "gemm_f64.hip", "gemm_persist.hip", "gemm_persist.h", "hdm_common.h")


def kernel_source_sha(root=None):
    """digest of the GEMM kernels' sources: a traffic file is only valid for the kernel generation it was measured on"""
    import hashlib
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in GEMM_SOURCES:
        with open(os.path.join(root, "hdsdp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic(dirs):
    """--traffic <FETCH_SIZE run dir> <WRITE_SIZE run dir>: HBM bytes per launch of every kernel as JSON, the way
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (separate --pmc passes; the counters are in KiB;
    FETCH_SIZE x 2 for the wide-read correction; `--nm NxM` first names the bench size the passes ran at): {kernel name: {fetch_kib, write_kib, launches, bytes_per_launch}}.
    bench.py reads the newest profiles/*_traffic.json and matches on the kernel name, so a renamed or re-templated
    kernel shows up as `traffic: null` instead of a stale number."""
    import json
    nm = "2000x2000"
    if dirs and dirs[0] == "--nm":
        nm, dirs = dirs[1], dirs[2:]
    agg = defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    a = agg[r["Kernel_Name"]][r["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    out = {}
    for k, v in agg.items():
        nf, nw = v["FETCH_SIZE"][0], v["WRITE_SIZE"][0]
        if nf == 0 or nw == 0:
            continue
        fk, wk = v["FETCH_SIZE"][1] / nf, v["WRITE_SIZE"][1] / nw
        out[k.replace("void ", "")] = {"fetch_kib": fk, "write_kib": wk, "launches": nf,
                                       "bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    import time
    print(json.dumps({"unit": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: L2-miss (fabric) bytes, "
                              "Infinity-Cache hits included -- an upper bound on HBM bytes",
                      "n": int(nm.split("x")[0]), "m": int(nm.split("x")[1]),
                      "kernel_source_sha": kernel_source_sha(), "persist": os.environ.get("HDM_PERSIST", "1") != "0",
                      "taken_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "kernels": out}, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        return traffic(sys.argv[2:])
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
