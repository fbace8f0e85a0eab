"""Host-side mirror of the reference's hook / imp_score interface (utils/common.py:258-977).

Same entry points, same argument meaning, same side effects:
  get_feature_hook / get_feature_hook_densenet / get_feature_hook_u2net_input
      forward-hook callables `(module, input, output) -> None` updating the module-level
      accumulator (the reference's globals feature_result / total, utils/common.py:258-259);
  inference / u2netp_inference   (utils/common.py:312-332);
  imp_score(net, args)           (utils/common.py:367-977): creates
      importance_score/<net>_limit<L>/ under the CWD, runs one forward sweep of `limit`
      batches per hook point, np.save()s the (C,) fp32 vectors under the reference's file
      names and prints the reference's progress lines.
What changed underneath: the per-map Python loop over dct_2d + .item() is ONE
dcts_energy_f32 launch per hooked tensor (ops.energy_nc), and the hooked tensor never
leaves the GPU.

Additions (opt-in, results identical on fixed batches):
  single_sweep=True   all hook points registered at once, one sweep instead of 12..118
                      (SURVEY.md §8 f1);
  accumulate="device" running mean kept on the GPU (dcts_running_mean_update_f32);
  deferred=True       single sweep where the hooks only keep references; at the end of each batch
                      all tensors of one tile shape are scored in ONE launch
                      (dcts_energy_multi_f32) and all running means in one more;
  world_size > 1      hook points LPT-sharded over ranks, one all-gather at the end,
                      rank 0 writes the files.
"""
import os

import numpy as np
import torch

from . import ops, schedules, sharding
from .accumulate import DeviceAccumulator, DeviceBatchAccumulator, HostAccumulator

# tests swap this for the oracle to exercise the host logic without a GPU
_energy_nc = ops.energy_nc

# the reference's module globals (utils/common.py:258-259)
_acc = HostAccumulator()


def _scored_tensor(kind, inputs, output):
    return inputs[0] if kind == "input" else output


def _hook_energy(kind, x):
    b = x.shape[1]
    if kind == "last12":
        return _energy_nc(x, c_begin=b - 12, c_count=12, pad_front_if_odd=True)
    if kind == "input":
        return _energy_nc(x, pad_front_if_odd=True)
    return _energy_nc(x)


def get_feature_hook(self, input, output):
    """utils/common.py:262-277."""
    _acc.update(_hook_energy("full", output))


def get_feature_hook_densenet(self, input, output):
    """utils/common.py:280-293: channels [b-12, b), cv2 path."""
    _acc.update(_hook_energy("last12", output))


def get_feature_hook_u2net_input(self, input, output):
    """utils/common.py:296-309: scores input[0], cv2 path."""
    _acc.update(_hook_energy("input", input[0]))


def make_weighted_feature_hook(weights_for):
    """Score variant in the coefficient domain (the reference only hints at variants,
    utils/common.py:268-269): a forward hook with get_feature_hook's signature and accumulation that scores
    sum_{u,v} w[u,v] * coeff[u,v]^2 instead of sum coeff^2. `weights_for(H, W)` returns the [H, W] weights
    (any array-like); they are cached per shape on the hooked tensor's device."""
    cache = {}

    def hook(self, input, output):
        key = (output.shape[2], output.shape[3], output.device)
        if key not in cache:
            cache[key] = torch.as_tensor(weights_for(output.shape[2], output.shape[3]), dtype=torch.float32).to(output.device)
        _acc.update(ops.weighted_energy_nc(output, cache[key]))

    return hook


_HOOKS = {"full": get_feature_hook, "last12": get_feature_hook_densenet, "input": get_feature_hook_u2net_input}


def _net_device(net):
    for p in net.parameters():
        return p.device
    return torch.device("cpu")


def inference(net, train_loader, limit):
    """utils/common.py:312-320 (data goes to the net's device instead of an unconditional .cuda())."""
    net.eval()
    dev = _net_device(net)
    for batch_idx, (data, _) in enumerate(train_loader):
        if batch_idx >= limit:
            break
        data = data.to(dev)
        with torch.no_grad():
            net(data)


def u2netp_inference(net, train_loader, limit):
    """utils/common.py:323-332: dict batches with key 'image', cast to float."""
    net.eval()
    dev = _net_device(net)
    with torch.no_grad():
        for batch_idx, data in enumerate(train_loader):
            if batch_idx >= limit:
                break
            inputs = data["image"].type(torch.FloatTensor)
            net(inputs.to(dev))


def _resolve(net, path):
    """'features.2' -> net.features[2], 'layer1.0.relu1' -> net.layer1[0].relu1 (what the
    reference's eval('net.' + name) / net.features[idx] expressions reach)."""
    mod = net
    for atom in path.split("."):
        mod = mod[int(atom)] if atom.isdigit() else getattr(mod, atom)
    return mod


def _schedule_for(net, name):
    if name == "googlenet":
        return schedules.googlenet(getattr(net, "filters_p", None))
    if name == "resnet_50":
        return schedules.resnet_50(tuple(getattr(net, "num_blocks", (3, 4, 6, 3))))
    if name not in schedules.SCHEDULES:
        raise ValueError("imp_score: unknown net %r" % (name,))
    return schedules.SCHEDULES[name]()


def _done_line(net_name, idx, stem):
    """The progress line the reference prints after saving (e.g. utils/common.py:395, :510, :515)."""
    if net_name == "u2netp":
        return None
    if net_name == "googlenet":
        if stem.endswith("_"):
            return "/" + stem[:-1] + ":done!"
        head, tp = stem.split("_", 2)[0:2], stem.split("_", 2)[2]
        return "/" + "_".join(head) + tp + ":done!"
    return "/" + stem + ":done!"


def _save(out_dir, net_name, pt, scores):
    for stem, lo, hi in pt.files:
        arr = scores if lo is None else scores[lo:hi]
        np.save(os.path.join(out_dir, stem + ".npy"), arr)
        line = _done_line(net_name, 0, stem)
        if line:
            print(line)


class _PointHook:
    """A hook with its own accumulator (single-sweep / device modes). With `batch` set, the
    device-side update is deferred and fused across hook points (DeviceBatchAccumulator).

    `ranges` (multi-GPU, single-sweep modes): the channel ranges [(key, lo, hi), ...] of this hook point's
    scored channels that THIS rank owns (sharding.make_units cuts wide layers so that eight ranks balance);
    None = the whole hook point under `key`. Per-channel scores do not depend on which call computes them,
    so the pieces concatenate to the unsplit result bit for bit."""

    def __init__(self, kind, accumulate, device, batch=None, key=None, deferred=False, ranges=None, nominal_c=None):
        self.kind, self.accumulate, self.device, self.acc = kind, accumulate, device, None
        self.batch, self.key, self.deferred = batch, key, deferred
        self.ranges, self.nominal_c, self.accs = ranges, nominal_c, {}

    def _pieces(self, x):
        """(key, c_begin, c_count, pad_front_if_odd) of every operator call this hook makes on x."""
        b = x.shape[1]
        base, count = (b - 12, 12) if self.kind == "last12" else (0, b)
        pad = self.kind != "full"
        if self.ranges is None:
            return [(self.key, base, count, pad)]
        if count != self.nominal_c:
            raise RuntimeError(
                "channel-range sharding cut this hook point by the schedule's channel count (%d) but the hooked "
                "tensor has %d: run pruned / non-standard nets without --single_sweep under torch.distributed"
                % (self.nominal_c, count))
        return [(k, base + lo, hi - lo, pad) for k, lo, hi in self.ranges]

    def __call__(self, module, inputs, output):
        x = _scored_tensor(self.kind, inputs, output)
        if self.ranges is None and not (self.deferred and self.batch is not None):
            pieces = [(self.key, None, None, None)]  # the reference's own three calls, argument for argument
        else:
            pieces = self._pieces(x)
        for key, cb, cc, pad in pieces:
            if self.deferred and self.batch is not None:
                self.batch.add_tensor(key, x, cb, cc, pad)
                continue
            e = _hook_energy(self.kind, x) if cb is None else _energy_nc(x, c_begin=cb, c_count=cc, pad_front_if_odd=pad)
            if self.batch is not None:
                self.batch.add(key, e)
                continue
            acc = self.accs.get(key)
            if acc is None:
                acc = self.accs[key] = (DeviceAccumulator(e.shape[1], e.device) if self.accumulate == "device"
                                        else HostAccumulator())
            acc.update(e)

    def scores(self, key=None):
        key = self.key if key is None else key
        if self.batch is not None:
            return np.ascontiguousarray(self.batch.scores(key), dtype=np.float32)
        return np.ascontiguousarray(self.accs[key].scores(), dtype=np.float32)


def imp_score(net, args, train_loader=None, single_sweep=False, accumulate="host", group=None, deferred=False):
    """Counterpart of utils/common.py:367-977. `args` needs .net, .limit (and whatever
    load_data reads when train_loader is None)."""
    global _acc
    if not hasattr(args, "limit"):
        # utils/load_models.py:819 calls imp_score from prune_*.py whose parsers define no --limit
        # (AttributeError in the reference as shipped); fall back to importance_generation.py's default
        args.limit = 5
    out_dir = "importance_score/" + args.net + "_limit" + str(args.limit)
    world, rank = 1, 0
    if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
        world = torch.distributed.get_world_size(group)
        rank = torch.distributed.get_rank(group)
    if rank == 0:
        if not os.path.isdir("importance_score"):
            os.mkdir("importance_score")
        if not os.path.isdir(out_dir):
            os.mkdir(out_dir)

    print("==> Loading data of {}..".format(getattr(args, "dataset", "synthetic")))
    if train_loader is None:
        from .data import load_data
        train_loader, _ = load_data(args)

    print("==> Generating importance score..")
    print("Importance Score is located at ./" + out_dir)
    _acc = HostAccumulator()

    pts = _schedule_for(net, args.net)
    sweep = u2netp_inference if args.net == "u2netp" else inference
    dev = _net_device(net)

    if deferred:
        single_sweep, accumulate = True, "device"

    # Work units and their owners. A unit is a hook point, or - in the single-sweep modes, where every rank
    # runs the one forward sweep anyway and only the scoring is divisible - a channel range of a wide hook
    # point (sharding.make_units): VGG-16-bn has 12 hook points and GoogLeNet 10, fewer than LPT needs to
    # balance eight ranks. In the reference's one-sweep-per-hook-point schedule a hook point stays whole:
    # cutting it would repeat its forward sweep on another rank.
    scored = [schedules.scored_shape(p) for p in pts]
    chans = [sc[1] for sc in scored]
    cost_pc = [float(p.H * p.W) for p in pts]
    if world > 1:
        total_cost = sum(c * k for c, k in zip(chans, cost_pc))
        cut = total_cost / (8.0 * world) if single_sweep else None  # G = 8: every net within 6 % of balance
        units = sharding.make_units(chans, cost_pc, max_unit_cost=cut)
        owner, load = sharding.assign(units, world)
        if rank == 0 and max(load) > 0:
            print("==> %d work units over %d ranks, load imbalance %.3f" % (len(units), world, max(load) * world / sum(load)))
    else:
        units = sharding.make_units(chans, cost_pc)
        owner = [0] * len(units)
    per_layer = {}
    for u in units:
        per_layer[u.layer] = per_layer.get(u.layer, 0) + 1
    mine = [k for k in range(len(units)) if owner[k] == rank]
    results = {}  # unit index -> (channels of the unit,) fp32

    if single_sweep:
        hooks, handles = {}, []
        batch = DeviceBatchAccumulator(dev) if (accumulate == "device" and dev.type == "cuda") else None
        by_layer = {}
        for k in mine:
            by_layer.setdefault(units[k].layer, []).append(k)
        for i, ks in by_layer.items():
            whole = per_layer[i] == 1
            hooks[i] = _PointHook(pts[i].kind, accumulate, dev, batch=batch, key=ks[0], deferred=deferred,
                                  ranges=None if whole else [(k, units[k].c_lo, units[k].c_hi) for k in ks],
                                  nominal_c=chans[i])
            handles.append(_resolve(net, pts[i].module).register_forward_hook(hooks[i]))
        sweep(net, train_loader, args.limit)
        for h in handles:
            h.remove()
        for i, ks in by_layer.items():
            for k in ks:
                results[k] = hooks[i].scores(k)
    else:
        for k in mine:  # one unit per hook point here
            i = units[k].layer
            pt = pts[i]
            if args.net == "u2netp" and world == 1:
                print("current layer:", "net." + pt.module)
            layer = _resolve(net, pt.module)
            if accumulate == "device":
                hook = _PointHook(pt.kind, accumulate, dev, key=k)
                handler = layer.register_forward_hook(hook)
                sweep(net, train_loader, args.limit)
                handler.remove()
                results[k] = hook.scores()
            else:
                handler = layer.register_forward_hook(_HOOKS[pt.kind])
                sweep(net, train_loader, args.limit)
                handler.remove()
                results[k] = np.ascontiguousarray(_acc.feature_result.numpy(), dtype=np.float32)
                _acc.reset()
            if world == 1:
                _save(out_dir, args.net, pt, results[k])
        if world == 1:
            print("The importance score generation has been completed!")  # utils/common.py:977
            return

    if world > 1:
        layer_scores = _gather_results(results, units, len(pts), owner, world, rank, dev, group)
    else:
        layer_scores = {units[k].layer: results[k] for k in mine}
    if rank == 0:
        for i, pt in enumerate(pts):
            if args.net == "u2netp":
                print("current layer:", "net." + pt.module)
            _save(out_dir, args.net, pt, layer_scores[i])
        print("The importance score generation has been completed!")
    if world > 1:
        torch.distributed.barrier(group)


def _gather_results(local, units, n_layers, owner, world, rank, dev, group):
    """One all-gather of the flat, equally padded score buffer (plus a tiny all-reduce that tells every rank
    the channel counts of the units, which only their owners know for certain: imp_score also runs on
    already-pruned nets whose widths differ from the schedule's). Returns {layer: (C,) scores}."""
    import torch.distributed as dist
    backend = dist.get_backend(group)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    counts = torch.zeros(len(units), dtype=torch.int64, device=cdev)
    for k, v in local.items():
        counts[k] = v.shape[0]
    dist.all_reduce(counts, group=group)
    counts = counts.cpu().tolist()
    # actual units: a whole hook point spans [0, actual C); a channel range keeps its bounds
    per_layer = [0] * n_layers
    for u in units:
        per_layer[u.layer] += 1
    real = []
    for k, u in enumerate(units):
        if per_layer[u.layer] == 1:
            real.append(sharding.Unit(u.layer, 0, counts[k], float(counts[k])))
        else:
            if counts[k] != u.c_hi - u.c_lo:
                raise RuntimeError("unit %d of hook point %d came back with %d channels, expected %d"
                                   % (k, u.layer, counts[k], u.c_hi - u.c_lo))
            real.append(u)
    chans = [0] * n_layers
    for u in real:
        chans[u.layer] = max(chans[u.layer], u.c_hi)
    off, seg = sharding.layout(real, owner, world)
    flat = torch.zeros(seg, dtype=torch.float32, device=cdev)
    for k, v in local.items():
        flat[off[k]:off[k] + counts[k]] = torch.from_numpy(v).to(cdev)
    gathered = sharding.all_gather_scores(flat, world, group)
    res = sharding.unpack(gathered, real, owner, off, chans)
    return {i: r.cpu().numpy() for i, r in enumerate(res)}
