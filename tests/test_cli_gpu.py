"""End to end on the GPU: importance_generation.py (the reference's CLI surface) -> score files ->
prune masks, for BASELINE.json config 2 (VGG-16-bn / CIFAR shapes, limit 5) at a reduced batch and
a ResNet-50 / 224x224 run at batch 2: file list, format and scores (1e-4 against the CPU oracle on a
second forward pass). Strict mask equality on identical activations: tests/test_mask_parity_gpu.py."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from dct_pruning_amd import harness, nets, schedules
from dct_pruning_amd.data import load_data
from oracle import dct_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_cli(tmp_path, *argv):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "importance_generation.py"), *argv], cwd=tmp_path,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    return p.stdout


def oracle_scores(name, args, points):
    """Same seeds as the CLI: same net, same synthetic batches; energies by the CPU oracle."""
    torch.manual_seed(args.seed)
    net = nets.get_network(name).cuda().eval()
    loader, _ = load_data(args)
    acts = {}
    handles = [harness._resolve(net, p.module).register_forward_hook(
        lambda m, i, o, _p=p: acts.setdefault(_p.module, []).append(
            (i[0] if _p.kind == "input" else o).detach().cpu())) for p in points]
    harness.inference(net, loader, args.limit)
    for h in handles:
        h.remove()
    out = {}
    for p in points:
        st = orc.HookState()
        for a in acts[p.module]:
            cb, cc, pad = schedules.scored_shape(p._replace(C=a.shape[1]))
            st.update(orc.energy_nc_batched(a, cb, cc, pad).sum(0), a.shape[0])
        out[p.files[0][0]] = st.feature_result.numpy()
    return out


@pytest.mark.parametrize("name,bs,limit,extra", [("vgg_16_bn", 32, 5, []), ("resnet_50", 2, 2, []),
                                                  ("vgg_16_bn", 32, 5, ["--single_sweep", "--device_accumulate"])])
def test_cli_scores(tmp_path, name, bs, limit, extra):
    dataset = "imagenet" if name == "resnet_50" else "cifar10"
    out = run_cli(tmp_path, "--net", name, "--dataset", dataset, "--synthetic", "--pretrain_dir", "",
                  "--batch_size", str(bs), "--limit", str(limit), *extra)
    assert "The importance score generation has been completed!" in out
    d = tmp_path / "importance_score" / ("%s_limit%d" % (name, limit))
    pts = schedules.SCHEDULES[name]()
    assert sorted(os.listdir(d)) == sorted(s + ".npy" for p in pts for s, _, _ in p.files)
    args = types.SimpleNamespace(net=name, dataset=dataset, synthetic=True,
                                 batch_size=bs, limit=limit, seed=0, input_size=None)
    some = pts[:4] + pts[-3:]
    ref = oracle_scores(name, args, some)
    for stem, r in ref.items():
        got = np.load(d / (stem + ".npy"))
        assert got.dtype == np.float32 and got.shape == r.shape
        # the forward pass re-runs here (MIOpen is not bit-reproducible run to run): 1e-4, not bitwise
        np.testing.assert_allclose(got, r, rtol=1e-4, atol=1e-6 * float(r.max()))
        # masks are compared strictly - no near-tie allowance - where the activations are identical on both
        # sides: tests/test_mask_parity_gpu.py (here the oracle's forward pass is a second MIOpen run)


def test_two_rank_cli_equals_single_rank(tmp_path):
    """The layer-sharded path end to end (two ranks rehearsed on the one GPU, gloo group): rank 0
    writes the same files as a single-rank run (1e-4: the forward pass re-runs)."""
    common = ["--net", "resnet_56", "--dataset", "cifar10", "--synthetic", "--pretrain_dir", "", "--batch_size", "16",
              "--limit", "2"]
    (tmp_path / "one").mkdir()
    (tmp_path / "two").mkdir()
    run_cli(tmp_path / "one", *common)
    env = dict(os.environ, PYTHONPATH=ROOT, DCTS_REHEARSE="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "importance_generation.py"), *common],
                       cwd=tmp_path / "two", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout
    d1 = tmp_path / "one" / "importance_score" / "resnet_56_limit2"
    d2 = tmp_path / "two" / "importance_score" / "resnet_56_limit2"
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2)) and len(os.listdir(d2)) == 55
    for f in os.listdir(d1):
        a, b = np.load(d1 / f), np.load(d2 / f)
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-6 * float(a.max()), err_msg=f)


def test_four_rank_single_sweep_cli_with_channel_range_units(tmp_path):
    """The single-sweep modes shard CHANNEL RANGES of wide hook points (VGG-16-bn: 12 hook points, four ranks):
    rank 0 writes the same 12 files as a single-rank run, every channel exactly once."""
    common = ["--net", "vgg_16_bn", "--dataset", "cifar10", "--synthetic", "--pretrain_dir", "", "--batch_size", "16",
              "--limit", "2", "--deferred"]
    (tmp_path / "one").mkdir()
    (tmp_path / "four").mkdir()
    run_cli(tmp_path / "one", *common)
    env = dict(os.environ, PYTHONPATH=ROOT, DCTS_REHEARSE="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "importance_generation.py"), *common],
                       cwd=tmp_path / "four", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout
    assert "work units over 4 ranks" in p.stdout
    d1 = tmp_path / "one" / "importance_score" / "vgg_16_bn_limit2"
    d4 = tmp_path / "four" / "importance_score" / "vgg_16_bn_limit2"
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d4)) and len(os.listdir(d4)) == 12
    for f in os.listdir(d1):
        a, b = np.load(d1 / f), np.load(d4 / f)
        assert a.shape == b.shape
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-6 * float(a.max()), err_msg=f)
