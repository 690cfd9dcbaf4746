"""Helpers shared by the tests and by tests/golden/make_golden.py."""
import numpy as np
import torch


def seeded_state_dict(module, seed, last_scale=None):
    """Deterministic weights reproducible on any side: tensor i of ``state_dict()`` (in order) is drawn
    from RandomState(seed + i); conv/linear weights ~ N(0, 1/fan_in), BN gamma ~ U(0.5,1.5),
    beta / bias ~ N(0, 0.05), running_mean ~ N(0, 0.1), running_var ~ U(0.5, 1.5)."""
    new = {}
    for i, (k, v) in enumerate(module.state_dict().items()):
        rng = np.random.RandomState(seed + i)
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            new[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            new[k] = torch.tensor(rng.uniform(0.5, 1.5, shp), dtype=torch.float32)
        elif k.endswith("running_mean"):
            new[k] = torch.tensor(rng.normal(0, 0.1, shp), dtype=torch.float32)
        elif v.dim() >= 2:
            fan_in = int(np.prod(shp[1:]))
            new[k] = torch.tensor(rng.normal(0, 1.0 / np.sqrt(fan_in), shp), dtype=torch.float32)
        elif "bn" in k or "downsample.1" in k:
            new[k] = torch.tensor(rng.uniform(0.5, 1.5, shp) if k.endswith("weight") else rng.normal(0, 0.05, shp),
                                  dtype=torch.float32)
        else:
            new[k] = torch.tensor(rng.normal(0, 0.05, shp), dtype=torch.float32)
    if last_scale is not None:  # shrink the output layer (small residual updates, like the trained MLP heads)
        last = [k for k in new if k.endswith('weight')][-1]
        new[last] = new[last] * last_scale
        new[last.replace('weight', 'bias')] = new[last.replace('weight', 'bias')] * last_scale
    return new
