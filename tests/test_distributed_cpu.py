"""The N>1 path on CPU: world_size-2 gloo processes run the layer-sharded harness and rank 0's
files equal the single-process result byte for byte. The energy operator is the
CPU oracle here (host logic only); the collective is the same code path as RCCL, backend aside."""
import os
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from dct_pruning_amd import sharding


def test_lpt_assignment_and_layout():
    chans = [64, 64, 256, 128, 512]
    cost = [100.0, 100.0, 100.0, 25.0, 25.0]
    units = sharding.make_units(chans, cost)
    owner, load = sharding.assign(units, 2)
    assert len(owner) == 5 and abs(load[0] - load[1]) <= 0.2 * sum(load)
    off, seg = sharding.layout(units, owner, 2)
    assert seg == max(sum(c for c, o in zip(chans, owner) if o == r) for r in (0, 1))
    # splitting wide layers improves balance and keeps every channel exactly once
    units2 = sharding.make_units(chans, cost, max_unit_cost=4000.0)
    cover = {}
    for u in units2:
        cover.setdefault(u.layer, []).append((u.c_lo, u.c_hi))
    for layer, spans in cover.items():
        spans.sort()
        assert spans[0][0] == 0 and spans[-1][1] == chans[layer]
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # world 1: identity
    g = sharding.all_gather_scores(torch.arange(4.0), 1)
    assert g.shape == (1, 4)


def _worker(rank, world, port, name, out_root):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dct_pruning_amd import harness
    from oracle import dct_oracle as orc
    from test_harness_cpu import run_harness
    harness._energy_nc = orc.energy_nc_batched
    d = os.path.join(out_root, "rank%d" % rank)
    os.makedirs(d)
    run_harness(name, d)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name", ["vgg_16_bn", "googlenet"])
def test_two_rank_gloo_equals_single_process(name, tmp_path, monkeypatch):
    from dct_pruning_amd import harness
    from oracle import dct_oracle as orc
    from test_harness_cpu import run_harness
    monkeypatch.setattr(harness, "_energy_nc", orc.energy_nc_batched)
    single, _, _ = run_harness(name, tmp_path / "single")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    from helpers import HARNESS_CASES
    limit = HARNESS_CASES[name][1]
    d0 = tmp_path / "rank0" / "importance_score" / ("%s_limit%d" % (name, limit))
    got = {f[:-4]: np.load(d0 / f) for f in os.listdir(d0)}
    assert sorted(got) == sorted(single)
    for k in single:
        assert got[k].tobytes() == single[k].tobytes(), k
    # only rank 0 writes
    assert not (tmp_path / "rank1" / "importance_score").exists()
