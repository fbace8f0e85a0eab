#!/usr/bin/env python3
"""Generate tests/golden/transplant_<net>.json by running the REFERENCE's own weight-transplant
loaders (utils/load_models.py:17-772: load_vgg_model, load_resnet_model, load_google_model,
load_densenet_model, load_resnet_imagenet_model, load_u2netp_model) in this build container, on the
reference's own model classes at FULL width, with seeded weights (tests/helpers.det_tensor, keyed by
state-dict name) and seeded score files (helpers.det_scores, keyed by file stem; ties and dead
channels included). Never runs on the GPU box: the reference does not travel, only these fixtures do.

A fixture holds no tensor data: the (key, shape) lists of the full and the pruned network, the rate
list, the score-file stems and lengths, and a 80-bit SHA-256 digest of every tensor of the pruned
model's state dict AFTER the reference's loader ran. tests/test_transplant_goldens.py rebuilds the same
inputs from the seeds, runs dct_pruning_amd.transplant and the oracle restatement, and compares digests.

What this pins: oracle/transplant_oracle.py and dct_pruning_amd/transplant.py against the reference's
real loops and the reference constructors' widths. torchvision is absent from this image; an inert
stand-in module is registered before the import (models/DUTS/u2net.py imports it and never uses it).

usage: python tests/golden/make_transplant_goldens.py [net ...]
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

from helpers import det_scores, det_tensor, tensor_digest  # noqa: E402
from dct_pruning_amd import schedules  # noqa: E402

CASES = {
    # net: rate list the reference's README / argparse default gives (README.md:90, :114, :138, :162, :186, :211; prune_u2netp.py:99)
    "vgg_16_bn": [0.5] * 7 + [0.95] * 5,
    "resnet_56": [0.0] + [0.18] * 29,
    "resnet_110": [0.0] + [0.2] * 2 + [0.3] * 18 + [0.4] * 18 + [0.39] * 19,
    "densenet_40": [0.0] + [0.2] * 12 + [0.0] + [0.2] * 12 + [0.0] + [0.2] * 12,
    "googlenet": [0.4] + [0.85] * 2 + [0.9] * 5 + [0.9] * 2,
    "u2netp": [0.40] * 40,
    "resnet_50": [0.0] + [0.1] * 3 + [0.4] * 7 + [0.4] * 9,
}


def fill(net, salt):
    sd = net.state_dict()
    with torch.no_grad():
        for k, t in sd.items():
            t.copy_(det_tensor(k, t.shape, salt))
    return net


def main(nets):
    from make_harness_goldens import _install_stand_ins
    _install_stand_ins()
    sys.path.insert(0, REF)
    import utils.common as rc
    import utils.load_models as lm
    for name in nets:
        rates = CASES[name]
        args = types.SimpleNamespace(net=name)
        full = fill(rc.get_network(args, [0.0] * 100), "")
        slim = fill(rc.get_network(args, rates), "slim:")
        stems = []
        for pt in schedules.SCHEDULES[name]():
            for stem, lo, hi in pt.files:
                stems.append([stem, schedules.scored_shape(pt)[1] if lo is None else hi - lo])
        t0 = time.time()
        with tempfile.TemporaryDirectory() as tmp:
            for stem, c in stems:
                np.save(os.path.join(tmp, stem + ".npy"), det_scores(stem, c))
            args.imp_score = tmp
            ori = full.state_dict()
            with contextlib.redirect_stdout(io.StringIO()):
                if name == "vgg_16_bn":
                    lm.load_vgg_model(slim, ori, args)
                elif name in ("resnet_56", "resnet_110"):
                    lm.load_resnet_model(slim, ori, int(name.split("_")[1]), args)
                elif name == "densenet_40":
                    lm.load_densenet_model(slim, ori, args)
                elif name == "googlenet":
                    lm.load_google_model(slim, ori, args)
                elif name == "resnet_50":
                    lm.load_resnet_imagenet_model(slim, ori, args)
                else:
                    lm.load_u2netp_model(slim, ori, args)
        out = slim.state_dict()
        fx = {"net": name, "rates": rates, "stems": stems,
              "ori": [[k, list(v.shape)] for k, v in full.state_dict().items()],
              "slim": [[k, list(v.shape)] for k, v in out.items()],
              "digest": {k: tensor_digest(v) for k, v in out.items()}}
        json.dump(fx, open(os.path.join(HERE, "transplant_%s.json" % name), "w"))
        print("%s: %d tensors, reference loader %.1f s" % (name, len(out), time.time() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
