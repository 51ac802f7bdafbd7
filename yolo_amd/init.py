"""Seeded, platform-independent weight initialisers (numpy Mersenne Twister, not torch's RNG), so the
same ``state_dict`` can be regenerated on any machine from a seed instead of shipping checkpoints.

``mode="spread"``: variance-preserving conv weights, randomised BatchNorm affine/statistics and
N(0,1)-scale head biases: activations keep a healthy range through the whole graph, head logits are
well spread (the default torch initialisation collapses every score of a random-init detector to
~0.25).  ``mode="torch_default"`` reproduces the *distribution* of torch's default module
initialisation (kaiming-uniform(a=sqrt(5)) convs, identity BatchNorm), i.e. what ``YoloV5(param)``
gives in the reference, with this module's RNG.
"""
import math
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch


def seeded_state_dict(model: torch.nn.Module, seed: int = 0, mode: str = "spread", scale: Dict[str, float] = None
                      ) -> "OrderedDict[str, torch.Tensor]":
    """``scale`` maps a key fragment to a factor applied to the generated tensor of every key containing it
    (e.g. damping the last conv of every residual branch so that a deep residual stack keeps its activation
    range)."""
    rng = np.random.RandomState(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    bn_prefixes = {n + "." for n, m in model.named_modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)}
    for name, ref in model.state_dict().items():
        shape = tuple(ref.shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros((), dtype=torch.long)
            continue
        is_bn = name[:len(name) - len(leaf)] in bn_prefixes
        if len(shape) == 4:                                        # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            if mode == "spread":
                v = rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)
            else:
                bound = 1.0 / math.sqrt(fan_in)
                v = rng.uniform(-bound, bound, shape)
        elif is_bn:
            if mode == "spread":
                if leaf == "weight":
                    v = rng.uniform(0.8, 1.2, shape)
                elif leaf == "running_var":
                    v = rng.uniform(0.7, 1.3, shape)
                else:                                              # bias, running_mean
                    v = rng.standard_normal(shape) * 0.1
            else:
                v = np.ones(shape) if leaf in ("weight", "running_var") else np.zeros(shape)
        elif len(shape) == 1 and leaf == "bias":                   # head conv bias
            if mode == "spread":
                v = rng.standard_normal(shape)
            else:
                w = model.state_dict()[name[:-4] + "weight"]
                bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
                v = rng.uniform(-bound, bound, shape)
        else:
            v = ref.detach().cpu().double().numpy()                # constants (e.g. DFL projection)
        for fragment, factor in (scale or {}).items():
            if fragment in name:
                v = np.asarray(v) * factor
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return out


def seeded_images(batch: int, size: int, seed: int = 0, start: int = 0) -> np.ndarray:
    """uniform[0,1) float32 NCHW images; image i depends only on (seed, start+i)."""
    imgs = np.empty((batch, 3, size, size), dtype=np.float32)
    for i in range(batch):
        imgs[i] = np.random.RandomState((seed * 1000003 + start + i) % (2 ** 31 - 1)).random_sample((3, size, size))
    return imgs
