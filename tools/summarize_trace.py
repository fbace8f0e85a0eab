#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace/--stats CSV output into a small text summary
(per-kernel count / total / avg / min / max in microseconds) for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


def main(d):
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not traces:
        print("no kernel_trace.csv under", d)
        return
    agg = defaultdict(list)
    t_min, t_max = None, None
    for row in csv.DictReader(open(traces[0])):
        s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        agg[short(row["Kernel_Name"])].append((e - s) / 1e3)
        t_min = s if t_min is None else min(t_min, s)
        t_max = e if t_max is None else max(t_max, e)
    total = sum(sum(v) for v in agg.values())
    print("# source: %s" % os.path.basename(traces[0]))
    print("# GPU busy %.1f us over a %.1f us span" % (total, (t_max - t_min) / 1e3))
    print("%-92s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-92s %7d %12.1f %10.2f %10.2f %10.2f %6.1f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                          100 * sum(v) / total))


if __name__ == "__main__":
    main(sys.argv[1])
