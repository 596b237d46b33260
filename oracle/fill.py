"""Deterministic closed-form parameter fill shared by `tools/gen_goldens.py` (applied to the
reference model's `state_dict`) and by the tests (applied to the oracle and to the HIP
model), so that golden fixtures need not carry the 59 MB of weights.  TEST INFRASTRUCTURE.

Each tensor is filled from a 64-bit integer hash of (crc32(name), flat index) -> uniform
[-1, 1), scaled by the tensor's role so that activations stay O(1) through the network.
"""
import zlib

import numpy as np
import torch


def _uniform(name, numel):
    idx = np.arange(numel, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = idx * np.uint64(6364136223846793005) + np.uint64(zlib.crc32(name.encode()) * 2654435761 + 1442695040888963407)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0)  # 24 bits -> [-1, 1)


def fill_tensor(name, shape):
    shape = tuple(shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    u = _uniform(name, numel).reshape(shape)
    if name.endswith('post_multipliers'):
        v = 0.5 * u
    elif 'embedding.weight' in name and len(shape) == 2:      # symbols / speaker embeddings
        v = 0.3 * u
    elif len(shape) >= 2:                                       # conv / linear / in_proj weights
        fan_in = int(np.prod(shape[1:]))
        v = u * (1.8 / np.sqrt(fan_in))
    elif name.endswith('.weight'):                              # LayerNorm gains
        v = 1.0 + 0.2 * u
    else:                                                       # biases
        v = 0.1 * u
    return torch.from_numpy(np.ascontiguousarray(v)).float()


def fill_params(shapes, inference_friendly=True):
    ''' shapes: ordered {state_dict name: shape}.  With `inference_friendly`, the duration
        head of the prosody predictor is re-centred so that predicted durations are ~60 ms
        (a few fall under the 23 ms threshold), which makes `inference` fixtures meaningful. '''
    P = {name: fill_tensor(name, shape) for name, shape in shapes.items()}
    if inference_friendly:
        w, b = 'prosody_predictor.projection.linear_layer.weight', 'prosody_predictor.projection.linear_layer.bias'
        P[w] = P[w] * torch.tensor([[0.1], [1.0], [1.0]])
        P[b] = torch.tensor([0.2, 0.1, -0.2])
    return P
