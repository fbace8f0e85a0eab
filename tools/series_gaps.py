#!/usr/bin/env python3
"""Per-dispatch durations and inter-dispatch gaps of the last `reps` dispatches of a kernel in a rocprofv3 kernel trace
(tools/launch_series.py).  usage: tools/series_gaps.py <rocprof dir> <kernel name substring> [reps]"""
import csv
import glob
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))]
rows.sort()
sel = [(s, e) for s, e, n in rows if pat in n][-reps:]
dur = [(e - s) / 1e3 for s, e in sel]
gap = [(sel[i + 1][0] - sel[i][1]) / 1e3 for i in range(len(sel) - 1)]
sd, sg = sorted(dur), sorted(gap)
print("%d dispatches of *%s*: duration median %.1f us (min %.1f, max %.1f); gap to the next dispatch median %.1f us (min %.1f, max %.1f)"
      % (len(sel), pat, sd[len(sd) // 2], sd[0], sd[-1], sg[len(sg) // 2], sg[0], sg[-1]))
print("first 12 durations:", " ".join("%.1f" % x for x in dur[:12]))
print("last 12 durations: ", " ".join("%.1f" % x for x in dur[-12:]))
