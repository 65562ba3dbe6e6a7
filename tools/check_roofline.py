#!/usr/bin/env python3
"""Recompute bench.py's roofline fractions from the COMMITTED rocprofv3 --stats CSV: for every kernel symbol of the bench line's
table, algorithmic bytes per launch (from the JSON) / average duration (from the CSV) / 8 TB/s, next to the figure the bench
measured in situ with hipEvents.

usage: python tools/check_roofline.py <bench.json> [<kernel_stats.csv>]"""
import csv
import json
import sys

d = json.load(open(sys.argv[1]))
rf = d["roofline"]
path = sys.argv[2] if len(sys.argv) > 2 else rf.get("rocprof_stats_source")
avg = {}
for row in csv.DictReader(open(path)):
    n = row["Name"]
    n = n[5:] if n.startswith("void ") else n
    n = n[6:] if n.startswith("tsgo::") else n
    avg[n.split("(")[0]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
print("%-44s %9s %9s %9s %8s %8s %9s" % ("kernel symbol", "MB/launch", "us bench", "us rocprof", "frac", "frac(csv)", "us model"))
for name, k in sorted(rf["kernels"].items(), key=lambda kv: -kv[1]["us_per_step"]):
    us_csv = avg.get(name, (None, 0))[0]
    b = k["algorithmic_bytes_per_launch"]
    print("%-44s %9.2f %9.2f %9s %8.3f %8s %9s" % (name, b / 1e6, k["us_in_situ"], "%.2f" % us_csv if us_csv else "-", b / (k["us_in_situ"] * 1e-6) / 8e12,
                                                 "%.3f" % (b / (us_csv * 1e-6) / 8e12) if us_csv else "-", "%.2f" % k["us_latency_model"] if k.get("us_latency_model") else "-"))
print("dominant:", rf["kernel"], "frac", round(rf["frac"], 3))
if rf.get("latency_model"):
    print("one PCG iteration: %.1f us measured, %.1f us by the model (us model = 3.7 us per launch + memory-side bytes / 6 TB/s; %d launches)"
          % (rf["us_per_pcg_iteration"], rf["latency_model"]["us_per_pcg_iteration"], rf["latency_model"]["launches_per_pcg_iteration"]))
