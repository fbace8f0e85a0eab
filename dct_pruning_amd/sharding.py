"""Multi-GPU partition of the score pass. One process per GPU.

Work units are hook points (layers); optionally a wide layer is cut into channel ranges —
per-channel scores are independent of which rank computes them, so any partition gives
results byte-identical to a single GPU. Units go to ranks by LPT (largest first onto the
least-loaded rank) on bytes per sweep. The only collective is ONE all-gather of a flat,
equally padded fp32 buffer at the end of the pass (RCCL over xGMI with backend "nccl";
"gloo" on CPU for the tests). The reference has no distributed path at all
(utils/common.py:52-53: .cuda() = device 0); this is new capability BASELINE.json asks for.
"""
from collections import namedtuple

import torch

Unit = namedtuple("Unit", "layer c_lo c_hi cost")  # channels [c_lo, c_hi) of hook point `layer`


def init_process_group(backend, rank=None, world_size=None, device=None, timeout_s=None, what="dct_pruning_amd"):
    """torch.distributed.init_process_group that cannot hang silently: first contact with RCCL on a new
    node is where a multi-GPU run dies (IPC mode, a missing link, a rank that never arrives), and a stuck
    bootstrap would otherwise sit there until the caller's own limit. The rendezvous and every later
    collective get `timeout_s` (env DCTS_DIST_TIMEOUT_S, default 180); a watchdog thread covers the part
    of communicator creation that ignores it; one 1-element all-reduce is run and waited for right away
    so that a broken fabric fails HERE, before any timing. On failure every rank prints one JSON line
    {"error": ..., "stage": ..., "rank": ...} (rank 0 on stdout, where the drivers read the bench line;
    the others on stderr) and the process exits with code 3."""
    import datetime
    import json
    import os
    import sys
    import threading
    import torch.distributed as dist

    if timeout_s is None:
        timeout_s = float(os.environ.get("DCTS_DIST_TIMEOUT_S", "180"))
    rk = int(os.environ.get("RANK", "0")) if rank is None else rank
    stage = ["rendezvous"]

    def fail(msg):
        line = json.dumps({"error": msg, "stage": stage[0], "rank": rk, "backend": backend, "what": what,
                           "timeout_s": timeout_s})
        print(line, file=sys.stdout if rk == 0 else sys.stderr, flush=True)
        if rk == 0:
            print(line, file=sys.stderr, flush=True)
        os._exit(3)

    dog = threading.Timer(timeout_s + 15.0, fail, args=("no progress within the time limit (hung bring-up)",))
    dog.daemon = True
    dog.start()
    try:
        kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
        if rank is not None:
            kw.update(rank=rank, world_size=world_size)
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
        stage[0] = "first collective"
        probe = torch.ones(1, dtype=torch.float32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if probe.is_cuda:
            torch.cuda.synchronize(probe.device)
        if int(probe.item()) != dist.get_world_size():
            fail("first all-reduce returned %r, expected %d" % (probe.item(), dist.get_world_size()))
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001 - whatever it was, it must end the rank loudly
        fail("%s: %s" % (type(exc).__name__, exc))
    finally:
        dog.cancel()


def make_units(channel_counts, costs_per_channel, max_unit_cost=None):
    """One unit per layer; layers costlier than max_unit_cost are cut into equal channel ranges."""
    units = []
    for layer, (c, cpc) in enumerate(zip(channel_counts, costs_per_channel)):
        parts = 1
        if max_unit_cost and c * cpc > max_unit_cost:
            parts = min(c, -(-int(c * cpc) // int(max_unit_cost)))
        step = -(-c // parts)
        lo = 0
        while lo < c:
            hi = min(c, lo + step)
            units.append(Unit(layer, lo, hi, (hi - lo) * cpc))
            lo = hi
    return units


def assign(units, world_size):
    """LPT: returns owner[i] for every unit (deterministic: ties by unit index)."""
    order = sorted(range(len(units)), key=lambda i: (-units[i].cost, i))
    load = [0.0] * world_size
    owner = [0] * len(units)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += units[i].cost
    return owner, load


def layout(units, owner, world_size):
    """Offsets of every unit inside its owner's segment of the flat buffer, and the padded
    per-rank segment length L = max_r sum of channels."""
    off = [0] * len(units)
    fill = [0] * world_size
    for i, u in enumerate(units):
        off[i] = fill[owner[i]]
        fill[owner[i]] += u.c_hi - u.c_lo
    return off, max(fill) if fill else 0


def all_gather_scores(local_flat, world_size, group=None):
    """local_flat: [L] fp32 on this rank's device -> [world_size, L] on every rank."""
    if world_size == 1:
        return local_flat[None, :]
    import torch.distributed as dist
    out = torch.empty(world_size * local_flat.numel(), dtype=local_flat.dtype, device=local_flat.device)
    dist.all_gather_into_tensor(out, local_flat.contiguous(), group=group)
    return out.view(world_size, local_flat.numel())


def unpack(gathered, units, owner, off, channel_counts):
    """[world, L] -> list of per-layer [C] score tensors (on gathered's device)."""
    res = [torch.empty(c, dtype=gathered.dtype, device=gathered.device) for c in channel_counts]
    for i, u in enumerate(units):
        n = u.c_hi - u.c_lo
        res[u.layer][u.c_lo:u.c_hi] = gathered[owner[i], off[i]:off[i] + n]
    return res
