#!/usr/bin/env python3
"""Score and prune-mask parity on IDENTICAL activations, at full configuration.

TEST INFRASTRUCTURE (imports oracle/). Used two ways:
  * `python tests/mask_parity.py --net resnet_50 --batch_size 256 --limit 5 --out profiles/...`
    on the GPU box: BASELINE.json's target configuration (importance_generation.py for ResNet-50 /
    224x224 / limit 5), one sweep with every hook point registered;
  * tests/test_mask_parity_gpu.py imports run() at reduced batch for all seven nets.

Every time a hook fires, the hooked tensor is scored twice: by the product path
(dct_pruning_amd.ops.energy_nc on the GPU + the reference-exact host accumulation) and, after a
device-to-host copy of the very same tensor, by the CPU oracle (oracle.energy_nc_batched + the same
running-mean rule, utils/common.py:271-277). The forward pass runs once, so MIOpen's run-to-run
differences cannot enter the comparison. Reported per score file: max relative score difference
(bar: 1e-4, BASELINE.json), dead channels exactly +0.0 on both sides, and whether the prune masks
of the consumer rule np.argsort(imp)[O-K:]; sort() (utils/load_models.py:40-41) are IDENTICAL for
the README's compress rates (kept widths from the reference's model constructors,
dct_pruning_amd.transplant) and for a sweep of generic rates."""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dct_pruning_amd import harness, masks, nets, ops, schedules, transplant  # noqa: E402
from dct_pruning_amd.accumulate import HostAccumulator  # noqa: E402
from dct_pruning_amd.data import load_data  # noqa: E402
from oracle import dct_oracle as orc  # noqa: E402

README_RATES = {  # /root/reference/README.md:90, :114, :138, :162, :186, :211
    "vgg_16_bn": [0.5] * 7 + [0.95] * 5,
    "resnet_56": [0.0] + [0.18] * 29,
    "resnet_110": [0.0] + [0.2] * 2 + [0.3] * 18 + [0.4] * 18 + [0.39] * 19,
    "densenet_40": [0.0] + [0.2] * 12 + [0.0] + [0.2] * 12 + [0.0] + [0.2] * 12,
    "googlenet": [0.4] + [0.85] * 2 + [0.9] * 5 + [0.9] * 2,
    "resnet_50": [0.0] + [0.1] * 3 + [0.4] * 7 + [0.4] * 9,
}
GENERIC_RATES = (0.1, 0.3, 0.5, 0.7, 0.95)


def readme_kept(net):
    """{file stem: kept width} under the README's compress_rate, where the consumer prunes that file."""
    if net == "resnet_50":
        return {s: k for s, o, k in transplant.resnet_50_kept(README_RATES[net]) if k != o}
    if net == "vgg_16_bn":
        return {s: k for s, o, k in transplant.vgg_16_bn_kept(README_RATES[net]) if k != o}
    if net in ("resnet_56", "resnet_110"):
        return {s: k for s, o, k in transplant.resnet_cifar_kept(README_RATES[net], int(net.split("_")[1])) if k != o}
    if net == "googlenet":
        return {s: k for s, o, k in transplant.googlenet_kept(README_RATES[net]) if k != o}
    if net == "densenet_40":
        return {s: k for s, o, k in transplant.densenet_40_kept(README_RATES[net]) if k != o}
    return {}


def run(net_name, batch_size, limit, seed=0, input_size=None, dataset=None, device="cuda", log=None):
    dataset = dataset or {"resnet_50": "imagenet", "u2netp": "DUTS"}.get(net_name, "cifar10")
    args = types.SimpleNamespace(net=net_name, dataset=dataset, synthetic=True, batch_size=batch_size, limit=limit,
                                 seed=seed, input_size=input_size)
    torch.manual_seed(seed)
    net = nets.get_network(net_name).to(device).eval()
    loader, _ = load_data(args)
    pts = harness._schedule_for(net, net_name)
    acc_gpu = [HostAccumulator() for _ in pts]
    acc_orc = [orc.HookState() for _ in pts]
    stats = {"maps": 0, "bytes": 0, "t_gpu": 0.0, "t_orc": 0.0}

    def make_hook(i, pt):
        def hook(module, inputs, output):
            x = inputs[0] if pt.kind == "input" else output
            cb, cc, pad = schedules.scored_shape(pt._replace(C=x.shape[1]))
            t0 = time.perf_counter()
            e = ops.energy_nc(x, cb, cc, pad)
            acc_gpu[i].update(e)  # device -> host, then the reference's own torch CPU ops
            t1 = time.perf_counter()
            xc = x.detach().to("cpu")
            eo = orc.energy_nc_batched(xc, cb, cc, pad)
            acc_orc[i].update(eo.view(xc.shape[0], -1).sum(0), xc.shape[0])
            t2 = time.perf_counter()
            stats["maps"] += xc.shape[0] * cc
            stats["bytes"] += xc.shape[0] * cc * xc.shape[2] * xc.shape[3] * 4
            stats["t_gpu"] += t1 - t0
            stats["t_orc"] += t2 - t1
        return hook

    handles = [harness._resolve(net, pt.module).register_forward_hook(make_hook(i, pt)) for i, pt in enumerate(pts)]
    sweep = harness.u2netp_inference if net_name == "u2netp" else harness.inference
    t0 = time.perf_counter()
    sweep(net, loader, limit)
    wall = time.perf_counter() - t0
    for h in handles:
        h.remove()

    kept = readme_kept(net_name)
    files = []
    for i, pt in enumerate(pts):
        g_all = np.ascontiguousarray(acc_gpu[i].scores(), dtype=np.float32)
        o_all = np.ascontiguousarray(acc_orc[i].feature_result.numpy(), dtype=np.float32)
        for stem, lo, hi in pt.files:
            g, o = (g_all, o_all) if lo is None else (g_all[lo:hi], o_all[lo:hi])
            nz = o != 0
            rel = float(np.max(np.abs(g[nz].astype(np.float64) - o[nz]) / np.abs(o[nz]))) if nz.any() else 0.0
            dead_ok = bool(np.all(g[~nz] == 0) and not np.any(np.signbit(g[~nz])))
            rec = {"file": stem, "C": int(g.size), "H": pt.H, "max_rel": rel, "dead": int((~nz).sum()), "dead_plus_zero": dead_ok,
                   "scores_bitwise_equal": bool(np.array_equal(g, o))}
            checks = {}
            if stem in kept:
                checks["readme(K=%d)" % kept[stem]] = bool(np.array_equal(masks.select_index(g, g.size, kept[stem]),
                                                                          masks.select_index(o, o.size, kept[stem])))
            for r in GENERIC_RATES:
                k = int(g.size * (1 - r))
                checks["rate %.2f" % r] = bool(np.array_equal(masks.select_index(g, g.size, k), masks.select_index(o, o.size, k)))
            rec["masks_equal"] = checks
            files.append(rec)
            if log:
                log("%-44s C=%5d %3dx%-3d rel %.2e dead %4d masks %s" % (
                    stem, g.size, pt.H, pt.W, rel, rec["dead"], "ok" if all(checks.values()) else "DIFFER " + str(checks)))
    summary = {
        "net": net_name, "batch_size": batch_size, "limit": limit, "seed": seed, "input_size": input_size,
        "files": len(files), "hook_points": len(pts), "maps_scored": stats["maps"], "activation_bytes": stats["bytes"],
        "max_rel": max(f["max_rel"] for f in files), "all_dead_plus_zero": all(f["dead_plus_zero"] for f in files),
        "all_masks_equal": all(all(f["masks_equal"].values()) for f in files),
        "readme_masks_checked": sum(1 for f in files for k in f["masks_equal"] if k.startswith("readme")),
        "readme_masks_equal": all(v for f in files for k, v in f["masks_equal"].items() if k.startswith("readme")),
        "seconds": {"sweep_wall": wall, "gpu_scoring_incl_host_update": stats["t_gpu"], "oracle_incl_d2h": stats["t_orc"]},
        "oracle": "oracle/dct_oracle.py energy_nc_batched (parity unpinned against torch_dct / cv2 round-off: neither is installed)",
    }
    return summary, files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="resnet_50")
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--limit", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--input_size", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    summary, files = run(a.net, a.batch_size, a.limit, a.seed, a.input_size, log=log)
    log(json.dumps(summary))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    sys.exit(0 if (summary["all_masks_equal"] and summary["max_rel"] <= 1e-4 and summary["all_dead_plus_zero"]) else 1)


if __name__ == "__main__":
    main()
