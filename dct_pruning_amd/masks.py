"""Prune masks from score files — the step right after the path.

The reference derives, inside its weight-transplant loops, for every conv whose width shrank
    select_index = np.argsort(imp)[orifilter_num - currentfilter_num:]; select_index.sort()
(utils/load_models.py:40-41 and its 8 siblings :103-104, :266-267, :314-315, :408-409, :470-471,
:522-523, :630-631, :746-747). That sorted index list IS the prune mask. This module computes it
from a score directory so "masks identical" is a tool run, not a training run:

    python -m dct_pruning_amd.masks --imp_score importance_score/vgg_16_bn_limit5 \\
        --compress_rate '[0.50]*7+[0.95]*5' --out masks.npz [--compare other_score_dir]

compress_rate uses the reference's mini-DSL (utils/common.py:164-181: '+'-joined terms, each with
one decimal rate and an optional *count). Rates are applied to the score files in natural order
(imp_conv1, imp_conv2, ...); kept = int(C * (1 - rate)) as the model constructors do
(e.g. models/cifar10/vgg.py:37). The consumer-side architecture tables (which file feeds which
conv) stay in the reference's load_models.py.
"""
import argparse
import os
import re
import sys

import numpy as np


def parse_compress_rate(text):
    """utils/common.py:164-181 semantics."""
    rates = []
    for term in text.split("+"):
        counts = re.findall(r"\*\d+", term)
        if len(counts) > 1:
            raise ValueError("more than one *count in %r" % term)
        num = int(counts[0][1:]) if counts else 1
        found = re.findall(r"\d+\.\d*", term)
        if len(found) != 1:
            raise ValueError("each term needs exactly one decimal rate: %r" % term)
        rates += [float(found[0])] * num
    return rates


def select_index(imp, orifilter_num, currentfilter_num):
    """utils/load_models.py:40-41."""
    sel = np.argsort(imp)[orifilter_num - currentfilter_num:]
    sel.sort()
    return sel


def _natural(name):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def score_files(score_dir):
    return sorted((f for f in os.listdir(score_dir) if f.endswith(".npy")), key=_natural)


def masks_for_dir(score_dir, rates):
    """{file stem: kept indices (int64, sorted)}; `rates` is a float or a list in natural file order."""
    files = score_files(score_dir)
    if isinstance(rates, (int, float)):
        rates = [float(rates)] * len(files)
    if len(rates) < len(files):
        raise ValueError("%d rates for %d score files" % (len(rates), len(files)))
    out = {}
    for f, r in zip(files, rates):
        imp = np.load(os.path.join(score_dir, f), allow_pickle=False)
        c = imp.shape[0]
        out[f[:-4]] = select_index(imp, c, int(c * (1 - r)))
    return out


def compare(a, b):
    """Names whose masks differ between two {stem: indices} dicts."""
    bad = [k for k in sorted(set(a) | set(b), key=_natural)
           if k not in a or k not in b or not np.array_equal(a[k], b[k])]
    return bad


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--imp_score", required=True, help="directory of imp_*.npy files")
    ap.add_argument("--compress_rate", default="[0.5]*200", help="reference DSL, e.g. '[0.50]*7+[0.95]*5'")
    ap.add_argument("--out", default=None, help="write the masks to this .npz")
    ap.add_argument("--compare", default=None, help="second score directory: report whether the masks match")
    args = ap.parse_args(argv)
    rates = parse_compress_rate(args.compress_rate)
    masks = masks_for_dir(args.imp_score, rates)
    for k, v in masks.items():
        print("%s: keep %d" % (k, v.size))
    if args.out:
        np.savez(args.out, **masks)
    if args.compare:
        bad = compare(masks, masks_for_dir(args.compare, rates))
        print("masks identical" if not bad else "masks differ in: " + ", ".join(bad))
        return 1 if bad else 0
    return 0


if __name__ == "__main__":
    sys.exit(main())
