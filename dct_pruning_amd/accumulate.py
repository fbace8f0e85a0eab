"""Score accumulation: the reference's module globals feature_result / total
(utils/common.py:258-259) and their update rule (:271-277), as objects.

HostAccumulator   reference-exact: per-map energies come back to the host and the SAME torch
                  CPU ops as the reference run on them (view/sum(0), three-rounding mean).
DeviceAccumulator the fused device form (dcts_running_mean_update_f32): nothing leaves the
                  GPU until the scores are read; same rounding sequence, batch sum in
                  ascending n instead of torch's pairwise sum(0).
DeviceBatchAccumulator  many hook points at once (single-sweep harness): the per-layer updates of
                  a batch are deferred and issued as ONE launch
                  (dcts_running_mean_update_multi_f32) when the next batch starts or the
                  scores are read.
"""
import torch

from . import _lib


class HostAccumulator:
    def __init__(self):
        self.reset()

    def reset(self):
        # utils/common.py:380-381
        self.feature_result = torch.tensor(0.)
        self.total = torch.tensor(0.)

    def update(self, energy_nc):
        """energy_nc: [a, b] fp32 (any device). utils/common.py:271-277."""
        a = energy_nc.shape[0]
        c = energy_nc.detach().to("cpu", torch.float32)
        c = c.view(a, -1)
        c = c.sum(0)
        self.feature_result = self.feature_result * self.total + c
        self.total = self.total + a
        self.feature_result = self.feature_result / self.total

    def scores(self):
        return self.feature_result.numpy()


class DeviceAccumulator:
    def __init__(self, c_count, device):
        self.device = torch.device(device)
        self.feature_result = torch.zeros(c_count, dtype=torch.float32, device=self.device)
        self.total = 0.0

    def reset(self):
        self.feature_result.zero_()
        self.total = 0.0

    def update(self, energy_nc):
        if energy_nc.dim() != 2 or energy_nc.shape[1] != self.feature_result.numel():
            raise ValueError("expected [N, %d] energies" % self.feature_result.numel())
        e = energy_nc.contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().dcts_running_mean_update_f32(
                e.data_ptr(), e.shape[0], e.shape[1], self.feature_result.data_ptr(), float(self.total), stream))
        self.total += e.shape[0]

    def scores(self):
        return self.feature_result.cpu().numpy()


class DeviceBatchAccumulator:
    def __init__(self, device):
        self.device = torch.device(device)
        self.state = {}      # key -> [feature_result tensor, total]
        self.pending = {}    # key -> energy tensor of the current batch
        self.pending_x = {}  # key -> (hooked tensor, c_begin, c_count, pad) awaiting deferred scoring

    def add(self, key, energy_nc):
        if key in self.pending:  # the same hook fired again: a new batch has started
            self.flush()
        e = energy_nc.contiguous()
        if key not in self.state:
            self.state[key] = [torch.zeros(e.shape[1], dtype=torch.float32, device=self.device), 0.0]
        self.pending[key] = e

    def add_tensor(self, key, x, c_begin, c_count, pad):
        """Deferred scoring: keep a reference to the hooked tensor; its energy is computed at flush time
        together with every other pending tensor of the same tile shape (ops.energy_multi). Nothing may
        overwrite the tensor before the flush (true for the reference's nets: hooked tensors are ReLU /
        pool / concat outputs that later layers only read). That is checked, not assumed: the tensor's
        autograd version counter is recorded here and compared at flush time; a net with an in-place op
        downstream of a hooked module (inplace ReLU, `out += ...`) raises instead of scoring overwritten
        data - use the per-hook mode (deferred=False) for such nets."""
        if key in self.pending or key in self.pending_x:
            self.flush()
        self.pending_x[key] = (x, c_begin, c_count, pad, x._version)

    def _score_pending_tensors(self):
        from . import ops
        keys = list(self.pending_x)
        for key in keys:
            x, cb, cc, pad, version = self.pending_x[key]
            if x._version != version:
                self.pending_x.clear()
                raise RuntimeError(
                    "deferred scoring: the tensor hooked at %r was modified in place after its hook fired "
                    "(version %d -> %d); scores would be those of the overwritten data. Run without "
                    "--deferred / deferred=False for this network." % (key, version, x._version))
        # every pending tensor in ONE call: small tiles of all shapes share a launch (dcts_energy_mixed_f32)
        outs = ops.energy_mixed([self.pending_x[k][:4] for k in keys])
        for k, e in zip(keys, outs):
            if k not in self.state:
                self.state[k] = [torch.zeros(e.shape[1], dtype=torch.float32, device=self.device), 0.0]
            self.pending[k] = e
        self.pending_x.clear()

    def flush(self):
        if self.pending_x:
            self._score_pending_tensors()
        if not self.pending:
            return
        keys = list(self.pending)
        descs = (_lib.UpdateDesc * len(keys))()
        for i, k in enumerate(keys):
            e, (fr, total) = self.pending[k], self.state[k]
            descs[i].energy_nc, descs[i].feature_result = e.data_ptr(), fr.data_ptr()
            descs[i].N, descs[i].C_count, descs[i].total_before = e.shape[0], e.shape[1], float(total)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().dcts_running_mean_update_multi_f32(descs, len(keys), stream))
        for k in keys:
            self.state[k][1] += self.pending[k].shape[0]
        # the launch is enqueued on the stream the energies were produced on: dropping the references
        # now is safe for the caching allocator (same-stream reuse is ordered after the kernel)
        self.pending.clear()

    def scores(self, key):
        self.flush()
        return self.state[key][0].cpu().numpy()
