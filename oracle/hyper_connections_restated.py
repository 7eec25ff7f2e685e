"""TEST INFRASTRUCTURE ONLY (oracle).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path never does.

Restatement of the third-party package `hyper-connections` (pinned `>=0.1.8` by
/root/reference/setup.py:29, source NOT vendored under /root/reference, no lockfile), as used
by the reference at audiolm_pytorch/audiolm_pytorch.py:24, 446, 452-454, 524, 551.

PARITY UNPINNED: there is no upstream source, test or golden vector for this component in
/root/reference; the arithmetic below restates the published algorithm ("Hyper-Connections",
Zhu et al. 2024, and the lucidrains implementation at v0.1.x) from knowledge:

  residuals R : ((b s), n, d)  -- batch-major, stream-minor
  width:  normed = F.normalize(R, dim=-1) * sqrt(d) * (gamma + 1)
          alpha  = tanh(normed @ dynamic_alpha_fn) * dynamic_alpha_scale + static_alpha   (.., s, s+1)
          beta   = tanh(normed @ dynamic_beta_fn)  * dynamic_beta_scale  + static_beta    (.., s)
          mix    = einsum('... s t, ... s d -> ... t d', alpha, R)
          branch_input = mix[..., 0, :],  R' = mix[..., 1:, :]
  depth:  R'' = branch_out[..., None, :] * beta[..., :, None] + R'
  num_residual_streams == 1 -> plain residual wrapper (`branch(x) + x`) with the same `.branch.` key prefix.

It is used (a) as the `hyper_connections` stub when the real reference is imported to
generate golden vectors (tests/golden/make_golden.py), and (b) as documentation of the
arithmetic that oracle/audiolm_oracle.py restates functionally.
"""
from __future__ import annotations

from functools import partial
from random import randrange

import torch
import torch.nn.functional as F
from torch import nn
from torch.utils._pytree import tree_flatten, tree_unflatten


def _exists(v):
    return v is not None


class RMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return F.normalize(x, dim=-1) * self.scale * (self.gamma + 1)


class Residual(nn.Module):
    """num_residual_streams == 1 / disable=True wrapper."""

    def __init__(self, *args, branch=None, **kwargs):
        super().__init__()
        self.branch = branch

    def width_connection(self, residuals):
        return residuals, residuals, dict()

    def depth_connection(self, branch_output, residuals):
        return branch_output + residuals

    def forward(self, residuals, *branch_args, **branch_kwargs):
        branch_input, residuals, residual_kwargs = self.width_connection(residuals)

        def add_residual_fn(branch_out):
            (branch_out, *rest), tree_spec = tree_flatten(branch_out)
            branch_out = self.depth_connection(branch_out, residuals, **residual_kwargs)
            return tree_unflatten((branch_out, *rest), tree_spec)

        if not _exists(self.branch):
            return branch_input, add_residual_fn

        branch_output = self.branch(branch_input, *branch_args, **branch_kwargs)
        return add_residual_fn(branch_output)


class HyperConnections(nn.Module):
    def __init__(self, num_residual_streams, *, dim, branch=None, layer_index=None, tanh=True, **kwargs):
        super().__init__()
        self.branch = branch
        self.act = nn.Tanh() if tanh else nn.Identity()
        self.norm = RMSNorm(dim)
        self.num_residual_streams = num_residual_streams
        s = num_residual_streams

        init_residual_index = (layer_index if _exists(layer_index) else randrange(s)) % s

        self.static_beta = nn.Parameter(torch.ones(s))

        init_alpha0 = torch.zeros((s, 1))
        init_alpha0[init_residual_index, 0] = 1.0
        self.static_alpha = nn.Parameter(torch.cat([init_alpha0, torch.eye(s)], dim=1))

        self.dynamic_alpha_fn = nn.Parameter(torch.zeros(dim, s + 1))
        self.dynamic_alpha_scale = nn.Parameter(torch.ones(()) * 1e-2)
        self.dynamic_beta_fn = nn.Parameter(torch.zeros(dim))
        self.dynamic_beta_scale = nn.Parameter(torch.ones(()) * 1e-2)

    def width_connection(self, residuals):
        s = self.num_residual_streams
        bs = residuals.shape[0]
        # '(b s) ... d -> b ... s d'
        r = residuals.reshape(bs // s, s, *residuals.shape[1:])
        r = r.movedim(1, -2)

        normed = self.norm(r)

        wc_weight = self.act(normed @ self.dynamic_alpha_fn)
        alpha = wc_weight * self.dynamic_alpha_scale + self.static_alpha

        dc_weight = self.act(normed @ self.dynamic_beta_fn)
        beta = dc_weight * self.dynamic_beta_scale + self.static_beta

        mix_h = torch.einsum('...st,...sd->...td', alpha, r)
        branch_input, r = mix_h[..., 0, :], mix_h[..., 1:, :]
        return branch_input, r, dict(beta=beta)

    def depth_connection(self, branch_output, residuals, *, beta):
        out = branch_output.unsqueeze(-2) * beta.unsqueeze(-1)   # 'b ... d, b ... s -> b ... s d'
        r = residuals + out
        # 'b ... s d -> (b s) ... d'
        r = r.movedim(-2, 1)
        return r.reshape(r.shape[0] * r.shape[1], *r.shape[2:])

    def forward(self, residuals, *branch_args, **branch_kwargs):
        branch_input, residuals, residual_kwargs = self.width_connection(residuals)

        def add_residual_fn(branch_out):
            (branch_out, *rest), tree_spec = tree_flatten(branch_out)
            branch_out = self.depth_connection(branch_out, residuals, **residual_kwargs)
            return tree_unflatten((branch_out, *rest), tree_spec)

        if not _exists(self.branch):
            return branch_input, add_residual_fn

        branch_output = self.branch(branch_input, *branch_args, **branch_kwargs)
        return add_residual_fn(branch_output)


class _Expand(nn.Module):
    def __init__(self, s):
        super().__init__()
        self.s = s

    def forward(self, x):   # 'b ... -> (b s) ...'
        return x.repeat_interleave(self.s, dim=0)


class _Reduce(nn.Module):
    def __init__(self, s):
        super().__init__()
        self.s = s

    def forward(self, x):   # '(b s) ... -> b ...' sum
        return x.reshape(x.shape[0] // self.s, self.s, *x.shape[1:]).sum(dim=1)


def get_expand_reduce_stream_functions(num_streams, disable=False):
    if num_streams == 1 or disable:
        return nn.Identity(), nn.Identity()
    return _Expand(num_streams), _Reduce(num_streams)


def get_init_and_expand_reduce_stream_functions(num_streams, disable=False):
    klass = HyperConnections if not disable else Residual
    init_fn = partial(klass, num_streams)
    return (init_fn, *get_expand_reduce_stream_functions(num_streams, disable=disable))
