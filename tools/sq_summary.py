#!/usr/bin/env python3
"""Per-kernel averages of the SQ/GRBM counters collected by tools/profile_sq.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if not name.startswith("k_"):
                continue
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    print(name)
    c = {k: sum(v) / len(v) for k, v in agg[name].items()}
    for k in sorted(c):
        print("   %-26s %16.0f" % (k, c[k]))
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in c:
                print("   %-26s %15.1f%% of wave cycles" % (k, 100 * c[k] / wc))
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        print("   VALU insts per wave        %16.1f" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"]))
