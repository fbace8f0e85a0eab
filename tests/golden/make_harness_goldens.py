#!/usr/bin/env python3
"""Generate tests/golden/harness_<net>.npz + state_dict_keys.json by running the REFERENCE's own
imp_score (utils/common.py:367-977) in this build container. Never runs on the GPU box: the
reference does not travel, only these fixtures (inputs are re-derived from seeds) do.

What this pins: the harness — hook order, channel slicing, odd-pad path selection, running
mean, file names, write order, stdout lines — and the architectures' state_dict layout.
What it does NOT pin: torch_dct / cv2 round-off. Both packages (and torchvision, skimage)
are absent from this image, so inert stand-in modules are registered before the import and
the DCT arithmetic inside this run is the oracle's restatement.

usage: python tests/golden/make_harness_goldens.py [net ...]
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import HARNESS_CASES, deterministic_init  # noqa: E402
from oracle import dct_oracle as orc  # noqa: E402
from dct_pruning_amd.data import SyntheticLoader  # noqa: E402


def _install_stand_ins():
    from scipy.fft import dctn

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", **{n: _Inert for n in (
        "Compose", "RandomCrop", "RandomHorizontalFlip", "ToTensor", "Normalize", "RandomResizedCrop", "Resize",
        "CenterCrop")})
    tv.datasets = mod("torchvision.datasets", CIFAR10=_Inert, ImageFolder=_Inert)
    tv.models = mod("torchvision.models")
    tv.utils = mod("torchvision.utils")
    sk = mod("skimage")
    sk.io = mod("skimage.io")
    sk.transform = mod("skimage.transform")
    sk.color = mod("skimage.color")
    mod("cv2", dct=lambda t: dctn(t, type=2, norm="ortho").astype(np.float32))
    mod("torch_dct", dct_2d=orc.dct_2d, dct=orc.dct_1d)


def main(nets):
    _install_stand_ins()
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self  # u2netp_inference calls .cuda() unconditionally
    import utils.common as rc  # the reference harness

    keys_path = os.path.join(HERE, "state_dict_keys.json")
    keys = json.load(open(keys_path)) if os.path.isfile(keys_path) else {}
    for name in nets:
        bs, limit, size, as_dict = HARNESS_CASES[name]
        args = types.SimpleNamespace(net=name, limit=limit, dataset="synthetic", batch_size=bs, data_dir=".")
        net = rc.get_network(args)
        deterministic_init(net)
        keys[name] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        loader = SyntheticLoader((3, size, size), bs, limit + 1, seed=7, as_dict=as_dict)
        rc.load_data = lambda a, _l=loader: (_l, None)
        cwd = os.getcwd()
        buf = io.StringIO()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                with contextlib.redirect_stdout(buf):
                    rc.imp_score(net, args)
            finally:
                os.chdir(cwd)
            d = os.path.join(tmp, "importance_score", "%s_limit%d" % (name, limit))
            files = sorted(os.listdir(d), key=lambda f: (os.path.getmtime(os.path.join(d, f)), f))
            arrays = {f[:-4]: np.load(os.path.join(d, f)) for f in files}
            raw = {f: open(os.path.join(d, f), "rb").read()[:128] for f in files[:1]}
        meta = {"files": sorted(arrays), "stdout": buf.getvalue().splitlines(),
                "case": {"batch_size": bs, "limit": limit, "size": size, "seed": 7},
                "first_header_hex": {k: v.hex() for k, v in raw.items()}}
        np.savez_compressed(os.path.join(HERE, "harness_%s.npz" % name), **arrays)
        json.dump(meta, open(os.path.join(HERE, "harness_%s.json" % name), "w"), indent=0)
        print(name, len(arrays), "files;", sum(a.size for a in arrays.values()), "floats;",
              "zeros:", sum(int((a == 0).sum()) for a in arrays.values()))
    json.dump(keys, open(keys_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1:] or list(HARNESS_CASES))
