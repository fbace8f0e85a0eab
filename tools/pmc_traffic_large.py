#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe_large.py -> HBM bytes per launch of every large-tile
kernel against its algorithmic bytes, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes (the read
side by the factor the known-size calibration read shows in the same run).
usage: tools/pmc_traffic_large.py <dir_fetch_pass> <dir_write_pass> > profiles/pmc_traffic_large.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_probe_large import SHAPES  # noqa: E402

MAPS = dict(SHAPES)


def load(d, counter):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                agg[name].append(float(row["Counter_Value"]))
    return agg


def edge_of(name):
    a = [int(v) for v in re.findall(r"-?\d+", name.split("<", 1)[1])] if "<" in name else []
    if name.startswith("k_tile2g"):
        return a[1] << a[0]          # <L, M, G, STORE>
    if name.startswith("k_tile2d"):
        return 8 * a[0]              # <M, STORE>
    if name.startswith(("k_split_fused2", "k_split_fused", "k_split_pipe")):
        return a[0] << a[1]          # <M, L>
    return None


def main(dfetch, dwrite):
    fetch, write = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    calib = fetch["k_calib_read"]
    raw = sum(calib) / len(calib) * 1024
    factor = ((1 << 28) * 4) / raw
    out = {"units": "bytes per launch; FETCH_SIZE / WRITE_SIZE are KiB per dispatch in the CSVs",
           "calibration": {"known_bytes": (1 << 28) * 4, "FETCH_SIZE_bytes": raw, "correction": factor}, "kernels": {}}
    for name, vals in sorted(fetch.items()):
        e = edge_of(name) if name.startswith("k_") else None
        if not e or e not in MAPS:
            continue
        v = vals[1:] if len(vals) > 2 else vals
        r = sum(v) / len(v) * 1024
        w = write.get(name) or []
        w = w[1:] if len(w) > 2 else w
        wb = sum(w) / len(w) * 1024 if w else 0.0
        alg = MAPS[e] * (4 * e * e + 4)
        out["kernels"][name] = {"edge": e, "maps": MAPS[e], "launches": len(vals), "fetch_raw_bytes": r, "fetch_corrected_bytes": r * factor,
                                "write_bytes": wb, "hbm_bytes_per_launch": r * factor + wb, "alg_bytes_per_launch": alg,
                                "hbm_over_alg": (r * factor + wb) / alg}
    import hashlib
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dct_pruning_amd", "csrc", "libdctscore.so")
    out["libdctscore_sha256"] = hashlib.sha256(open(so, "rb").read()).hexdigest() if os.path.isfile(so) else None
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
