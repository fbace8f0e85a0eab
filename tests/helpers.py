"""Shared test helpers (no reference code, no oracle dependency)."""
import math
import zlib

import torch


def deterministic_init(net):
    """Fill every parameter/buffer from a generator keyed by the tensor's state_dict NAME, so two
    implementations of the same architecture (same keys/shapes) get bit-identical weights without
    sharing constructor RNG order. Kaiming-scaled convs keep activations alive through deep nets."""
    sd = net.state_dict()
    with torch.no_grad():
        for key, t in sd.items():
            g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
            if key.endswith("num_batches_tracked"):
                t.zero_()
            elif key.endswith("running_var"):
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            elif key.endswith("running_mean"):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif t.dim() == 1:
                base = 1.0 if key.endswith("weight") else 0.0
                t.copy_(base + 0.1 * torch.randn(t.shape, generator=g))
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan_in))
    return net


HARNESS_CASES = {
    # net: (batch_size, limit, input H=W, dict batches)
    "vgg_16_bn": (4, 2, 32, False),
    "resnet_56": (2, 1, 32, False),
    "resnet_110": (1, 1, 32, False),
    "densenet_40": (2, 1, 32, False),
    "googlenet": (2, 1, 32, False),
    "resnet_50": (1, 1, 64, False),
    "u2netp": (1, 1, 72, True),
}


def synth(n, c, h, w, seed, dead=True):
    """SURVEY.md §8(d) synthetic maps: relu(randn) * per-channel scale, every c % 8 == 5 dead."""
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, c, h, w, generator=g))
    s = torch.exp(0.5 * torch.randn(c, generator=g))
    if dead:
        s[torch.arange(c) % 8 == 5] = 0
    return x * s[None, :, None, None]


def det_tensor(key, shape, salt=""):
    """The tensor deterministic_init gives `key` (same rule), from (key, shape) alone; `salt` makes a
    second, different network with the same keys (the slim model of a transplant)."""
    t = torch.empty(tuple(shape), dtype=torch.int64 if key.endswith("num_batches_tracked") else torch.float32)
    g = torch.Generator().manual_seed(zlib.crc32((salt + key).encode()))
    if key.endswith("num_batches_tracked"):
        t.zero_()
        if salt:
            t += 3
    elif key.endswith("running_var"):
        t.copy_(0.5 + torch.rand(t.shape, generator=g))
    elif key.endswith("running_mean"):
        t.copy_(0.1 * torch.randn(t.shape, generator=g))
    elif t.dim() == 1:
        base = 1.0 if key.endswith("weight") else 0.0
        t.copy_(base + 0.1 * torch.randn(t.shape, generator=g))
    else:
        fan_in = t[0].numel()
        t.copy_(torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan_in))
    return t


def det_scores(stem, c):
    """Seeded score vector for file `stem`: exact ties and dead channels included."""
    import numpy as np
    g = torch.Generator().manual_seed(zlib.crc32(("score:" + stem).encode()))
    s = torch.rand(c, generator=g)
    s[torch.arange(c) % 5 == 3] = 0.0
    if c > 4:
        s[1] = s[2]
    return s.numpy().astype(np.float32)


def tensor_digest(t):
    import hashlib
    t = t.detach().cpu().contiguous()
    return hashlib.sha256(str(tuple(t.shape)).encode() + str(t.dtype).encode() + t.numpy().tobytes()).hexdigest()[:20]
