"""TEST INFRASTRUCTURE ONLY (oracle).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path never does.

Restatement of `vector-quantize-pytorch` (`>=1.19.3`, /root/reference/setup.py:38; source NOT
vendored, no lockfile) `GroupedResidualVQ` **eval-mode forward**, as constructed by the
reference at audiolm_pytorch/soundstream.py:592-607 and called at soundstream.py:840, and `get_output_from_indices` (soundstream.py:697).

PARITY UNPINNED: no upstream source / tests / golden vectors under /root/reference.  Restated
from the published algorithm (SoundStream residual VQ, Zeghidour et al. 2021; lucidrains
EuclideanCodebook) -- SURVEY.md §8(a) A17:

  per group g (features split in `groups` chunks along the channel dim):
    residual = x_g ; out = 0
    for q in 0..Q-1:
        d(x,e)  = sqrt(clamp(|x|^2 + |e|^2 - 2 x.e, min=0))
        idx_q   = argmax(-d)              (first maximum on ties == first minimum distance)
        quant   = E_q[idx_q]
        residual = residual - quant ; out = out + quant
  eval mode => no EMA update, no code expiry, no quantize-dropout, no commitment loss,
  no rotation trick, no stochastic sampling.  Codebooks must be `initted` (kmeans init on the
  first batch is NOT restated: parity configs set the codebooks explicitly).

state_dict names mirror the upstream module tree so real checkpoints load:
  rvqs.{g}.layers.{q}._codebook.{initted, cluster_size, embed_avg, embed(1, C, d)}
"""
from __future__ import annotations

import torch
from torch import nn


class EuclideanCodebook(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        self.register_buffer('initted', torch.tensor([False]))
        self.register_buffer('cluster_size', torch.ones(1, codebook_size))
        self.register_buffer('embed_avg', torch.zeros(1, codebook_size, dim))
        self.register_buffer('embed', torch.zeros(1, codebook_size, dim))

    def forward(self, x):
        assert bool(self.initted.item()), 'restated RVQ needs explicitly initialised codebooks'
        shape = x.shape
        flat = x.reshape(1, -1, shape[-1]).float()
        embed = self.embed.float()
        x2 = (flat ** 2).sum(dim=-1)
        y2 = (embed ** 2).sum(dim=-1)
        xy = torch.einsum('bid,bjd->bij', flat, embed) * -2
        dist = -(x2.unsqueeze(-1) + y2.unsqueeze(-2) + xy).clamp(min=0).sqrt()
        ind = dist.argmax(dim=-1)                                  # (1, B*T)
        quant = embed[0][ind[0]]
        return quant.reshape(shape), ind.reshape(shape[:-1])


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        self._codebook = EuclideanCodebook(dim, codebook_size)

    def forward(self, x):
        return self._codebook(x)


class ResidualVQ(nn.Module):
    def __init__(self, *, dim, num_quantizers, codebook_size, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantize(dim, codebook_size) for _ in range(num_quantizers)])

    def forward(self, x):
        residual, out, inds = x, 0., []
        for layer in self.layers:
            quant, ind = layer(residual)
            residual = residual - quant
            out = out + quant
            inds.append(ind)
        return out, torch.stack(inds, dim=-1), torch.zeros(1, len(self.layers), device=x.device)

    def get_output_from_indices(self, indices):
        """(b, n, q) int -> (b, n, d): sum over the quantizers of the selected code vectors; index -1 (quantize-dropout / padding) selects
        nothing (upstream get_codes_from_indices masks those positions to 0).  project_out is the identity here (codebook_dim == dim)."""
        out = 0.
        for q, layer in enumerate(self.layers):
            idx = indices[..., q]
            codes = layer._codebook.embed[0][idx.clamp(min=0)]
            out = out + codes.masked_fill((idx < 0).unsqueeze(-1), 0.)
        return out


class GroupedResidualVQ(nn.Module):
    def __init__(self, *, dim, groups=1, num_quantizers, codebook_size, **kwargs):
        super().__init__()
        assert dim % groups == 0
        self.dim, self.groups = dim, groups
        self.rvqs = nn.ModuleList([
            ResidualVQ(dim=dim // groups, num_quantizers=num_quantizers, codebook_size=codebook_size)
            for _ in range(groups)])

    def forward(self, x):
        assert not self.training, 'restated RVQ implements the eval-mode forward only'
        chunks = x.chunk(self.groups, dim=-1)
        outs = [rvq(c) for rvq, c in zip(self.rvqs, chunks)]
        quantized = torch.cat([o[0] for o in outs], dim=-1)
        indices = torch.stack([o[1] for o in outs])                # (g, b, n, q)
        losses = torch.stack([o[2] for o in outs])
        return quantized, indices, losses

    def get_output_from_indices(self, indices):
        """(g, b, n, q) -> (b, n, dim): per-group outputs concatenated along the feature dim (soundstream.py:697)."""
        return torch.cat([rvq.get_output_from_indices(ix) for rvq, ix in zip(self.rvqs, indices)], dim=-1)


class _Unsupported(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError('not restated (out of scope: SURVEY.md §2 row 6)')


GroupedResidualLFQ = GroupedResidualFSQ = _Unsupported
