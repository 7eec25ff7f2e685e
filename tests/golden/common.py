"""Shared between the golden-vector generator (runs the REAL reference, build container only) and
the tests (run everywhere).  Keeps fixtures tiny: a fixture stores the parameter *shapes*, a seed,
the integer inputs and the reference's outputs; the fp32 parameter values are re-synthesised
deterministically on both sides from (shapes, seed) with the CPU generator.
"""
from __future__ import annotations

import math
import re

import torch

GOLDEN_DIR = __import__('os').path.dirname(__import__('os').path.abspath(__file__))

_HC_GAMMA = re.compile(r'transformer\.layers\.\d+\.[02]\.norm\.gamma$')


def synth_state_dict(shapes: dict, seed: int, *, streams: int = 4, dtype=torch.float32) -> dict:
    """Deterministic, non-degenerate parameter values for every key of a reference state_dict.

    Values are chosen so that every code path carries signal (hyper-connection dynamic weights,
    RMSNorm gamma, LayerNorm gamma, biases are all non-trivial) -- the reference's default init
    leaves several of them at exactly zero.
    """
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    sd = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        last = key.split('.')[-1]
        if key.endswith('_codebook.initted'):
            v = torch.tensor([True])
        elif key.endswith('_codebook.cluster_size'):
            v = torch.ones(shape)
        elif key.endswith('_codebook.embed_avg') or key.endswith('_codebook.embed'):
            v = r * 0.5
        elif last == 'beta':                                  # LayerNorm zero buffer (audiolm_pytorch.py:195)
            v = torch.zeros(shape)
        elif _HC_GAMMA.search(key):                           # hyper-connection RMSNorm gamma (init 0)
            v = 0.1 * r
        elif last == 'gamma':                                 # LayerNorm gamma (init 1)
            v = 1.0 + 0.1 * r
        elif last == 'static_alpha':
            s = shape[0]
            base = torch.cat([torch.zeros(s, 1), torch.eye(s)], dim=1)
            base[0, 0] = 1.0
            v = base + 0.1 * r
        elif last == 'static_beta':
            v = 1.0 + 0.1 * r
        elif last in ('dynamic_alpha_fn', 'dynamic_beta_fn'):
            # width-scaled (round 3): the pre-activation `normed @ fn` sums shape[0] = dim features of unit RMS, so a fixed 0.05 * randn gives a
            # std of 0.05 * sqrt(dim) -- 0.4 at dim 64 (every small golden: unchanged) but 1.6 at dim 1024, where tanh saturates and every
            # rounding difference is amplified until no bound on the 4-stream gradients discriminates (round-2 VERDICT).  Keep it at 0.4.
            v = 0.05 * math.sqrt(64.0 / max(64, shape[0])) * r
        elif last in ('dynamic_alpha_scale', 'dynamic_beta_scale'):
            v = 0.1 + 0.02 * r
        elif last == 'bias':
            v = 0.02 * r
        elif key.endswith('logit_weights'):
            v = r * (2.0 / math.sqrt(shape[-1]))
        elif 'embedding' in key:
            v = 0.5 * r
        elif key.endswith('start_token') or key in ('cross_attn_bias', 'null_pos_bias'):
            v = 0.5 * r
        elif last == 'weight' and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            v = r / math.sqrt(fan_in)
        else:
            v = 0.1 * r
        sd[key] = v.to(dtype) if v.dtype.is_floating_point else v
    return sd


def grad_digest(grads: dict, full: bool) -> dict:
    """Per-parameter gradient summary stored in a fixture: L2 norm + a strided sample (+ full tensor
    for small models)."""
    out = {}
    for k, gr in grads.items():
        if gr is None:
            out[k] = None
            continue
        flat = gr.detach().float().reshape(-1)
        stride = max(1, flat.numel() // 257)
        out[k] = dict(norm=float(flat.norm()), sample=flat[::stride].clone(), stride=stride,
                      full=(gr.detach().float().clone() if full else None))
    return out
