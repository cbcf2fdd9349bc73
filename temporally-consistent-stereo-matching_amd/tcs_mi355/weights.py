"""Key-seeded synthetic weights (SURVEY.md §8c fixture plan).

No checkpoint ships with the reference (README.md:82 is a Dropbox link), so every parity run and
every bench uses the same deterministic, name-seeded weights: for each state-dict key a
`numpy Philox(key=crc32(name))` normal draw scaled by sqrt(2/fan_in), biases zero, and the three
regression heads damped so that the 32-iteration loop stays contractive (its own fp32 noise floor is
then ~1e-4 EPE instead of ~8e-4).  The same function feeds the reference model (via
`load_state_dict`), the CPU oracle and the HIP model, on any machine, with no weight file.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import numpy as np
import torch

# update.py:137 (flow head), :193-195 (gradient residual head), :348-350 (mono disparity head)
DAMPED_HEADS = ("update_block.flow_head.conv2.weight", "disp_grad_refine.residual_head.2.weight",
                "disp_completor.disp_head.2.weight")
DAMP = 0.05


def synth_tensor(name: str, shape: Sequence[int]) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith("running_mean"):
        return torch.zeros(shape)
    if name.endswith("running_var"):
        return torch.ones(shape)
    if len(shape) <= 1:
        # conv biases -> 0; affine norm weights -> 1 (extractor.py:256-260)
        if name.endswith(".weight"):
            return torch.ones(shape)
        return torch.zeros(shape)
    gen = np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode())))
    fan_in = int(np.prod(shape[1:]))
    w = gen.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
    if name in DAMPED_HEADS:
        w = w * np.float32(DAMP)
    return torch.from_numpy(np.ascontiguousarray(w))


def synth_state_dict(key_shapes: Mapping[str, Sequence[int]]) -> Dict[str, torch.Tensor]:
    """key_shapes: state-dict key -> shape (e.g. tests/golden/state_dict_keys.json, or
    {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: synth_tensor(k, s) for k, s in key_shapes.items()}


def load_synth_weights(model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    """Fill `model` (reference-compatible key names) with the key-seeded weights, strict."""
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd, strict=True)
    return sd
