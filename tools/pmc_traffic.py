#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs into per-kernel HBM bytes per launch,
corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the factor for
the read side comes from the calibration kernel of known size in the same access width.
usage: tools/pmc_traffic.py <dir_with_fetch_pass> <dir_with_write_pass> > profiles/pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[name].append(float(row["Counter_Value"]))
    return agg


PROBE_SHAPES = {56: 65536, 28: 262144, 14: 1048576, 7: 4194304, 32: 196608}  # tools/pmc_probe.py
PROBE_FUSED = {(14, 4): (224, 4096), (16, 3): (128, 12288)}  # k_split_fused<M, L> -> (edge, maps)


def main(dfetch, dwrite):
    fetch, write = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    out = {"units": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch (rocprofv3 derived counters)"}
    calib = fetch.get("k_calib_read")
    known = (1 << 28) * 4
    factor = None
    if calib:
        raw = sum(calib) / len(calib) * 1024
        factor = known / raw
        out["calibration_dword_per_lane"] = {"known_bytes": known, "FETCH_SIZE_bytes": raw, "correction": factor}
    for name, vals in fetch.items():
        if not name.startswith("k_"):
            if "copy" in name.lower() or "elementwise" in name.lower():
                out.setdefault("other", {})[name[:60]] = sum(vals) / len(vals) * 1024
            continue
        key = name.replace("<", "_").replace(">", "").replace(", ", "_")
        raw = sum(vals) / len(vals) * 1024
        w = write.get(name)
        wb = sum(w) / len(w) * 1024 if w else None
        alg = None
        for edge, nmaps in PROBE_SHAPES.items():
            if key.startswith("k_energy_codelet_%d_%d_" % (edge, edge)):
                alg = nmaps * (4 * edge * edge + 4)
        for (mm, ll), (edge, nmaps) in PROBE_FUSED.items():
            if key in ("k_split_fused_%d_%d" % (mm, ll), "k_split_pipe_%d_%d" % (mm, ll)):
                alg = nmaps * (4 * edge * edge + 4)
        out[key] = {"launches": len(vals), "alg_bytes_per_launch": alg,
                    "hbm_over_alg": (((raw * factor if factor else raw) + (wb or 0.0)) / alg) if alg else None,
                    "fetch_raw_bytes_per_launch": raw,
                    "fetch_corrected_bytes_per_launch": raw * factor if factor else None,
                    "write_bytes_per_launch": wb,
                    "hbm_bytes_per_launch": (raw * factor if factor else raw) + (wb or 0.0)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
