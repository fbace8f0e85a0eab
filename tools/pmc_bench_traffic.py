#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py -> HBM bytes per launch of every dcts kernel in
the run, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE reads half
the bytes of a coalesced streaming read on gfx950; the factor is taken from k_calib_read, a read of known
size (bench.py --pmc-calib) in the same run and access width. WRITE_SIZE is taken as is.
usage: tools/pmc_bench_traffic.py <dir_fetch_pass> <dir_write_pass> > profiles/pmc_traffic_bench.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CALIB_BYTES = (1 << 28) * 4  # bench.py --pmc-calib


def load(d, counter):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[name].append(float(row["Counter_Value"]))
    return agg


def bench_line(d):
    for l in open(os.path.join(d, "bench.json")):
        if l.startswith("{"):
            return json.loads(l)
    return {}


def main(dfetch, dwrite):
    fetch, write = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    line = bench_line(dfetch)
    out = {"units": "bytes per launch; FETCH_SIZE / WRITE_SIZE are KiB per dispatch in the CSVs",
           "bench": {k: line.get(k) for k in ("metric", "value", "ms_per_step", "config")},
           "bench_roofline": line.get("roofline")}
    calib = fetch.get("k_calib_read")
    factor = None
    if calib:
        raw = sum(calib) / len(calib) * 1024
        factor = CALIB_BYTES / raw
        out["calibration"] = {"known_bytes": CALIB_BYTES, "FETCH_SIZE_bytes": raw, "correction": factor}
    kernels = {}
    for name, vals in fetch.items():
        if not name.startswith("k_") or name == "k_calib_read":
            continue
        # steady state: drop the first launch of a kernel (cold caches / first touch)
        v = vals[1:] if len(vals) > 2 else vals
        raw = sum(v) / len(v) * 1024
        w = write.get(name)
        w = (w[1:] if w and len(w) > 2 else w) or []
        wb = sum(w) / len(w) * 1024 if w else 0.0
        kernels[name] = {"launches": len(vals), "fetch_raw_bytes": raw, "fetch_corrected_bytes": raw * factor if factor else None,
                         "write_bytes": wb, "hbm_bytes_per_launch": (raw * factor if factor else raw) + wb}
    rl = line.get("roofline") or {}
    dom = rl.get("kernel", "")
    want = dom.split(" (")[0].replace(" ", "").rstrip(">")  # "k_energy_codelet_multi<56,56" / "k_tile2d" / ...
    for name, k in kernels.items():
        if want and name.replace(" ", "").startswith(want):
            k["alg_bytes_per_launch"] = rl.get("alg_bytes_per_launch")
            if k["alg_bytes_per_launch"]:
                k["hbm_over_alg"] = k["hbm_bytes_per_launch"] / k["alg_bytes_per_launch"]
    out["kernels"] = kernels
    # the record is valid for ONE binary: bench.py compares this with the library it runs and reports traffic: null otherwise
    import hashlib
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dct_pruning_amd", "csrc", "libdctscore.so")
    out["libdctscore_sha256"] = hashlib.sha256(open(so, "rb").read()).hexdigest() if os.path.isfile(so) else None
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
