"""Logit heads + cross-entropy as ONE autograd node (reference audiolm_pytorch.py:719, :957-983, :1319-1361 for the logits and
:1561-1565, :1826-1854, :2112-2137 for the losses).

The reference computes 'q c d, b n q d -> b n q c' on a (b, n, q, d) view plus a remainder einsum with W[:r].  Position i of a
head's token range always uses W[i mod Q] (the remainder rows continue the same pattern), so here the final hidden states are
REGROUPED per quantizer ('b (n q) d -> q (b n) d', a row gather driven by an int32 index built from integer bookkeeping) and every
head group becomes one batched MFMA GEMM over q with fp32 logits, followed by an online-softmax cross-entropy.  Rows that do not
exist for a quantizer (the ragged tail) are padded with index -1 / label -1 and contribute nothing.
"""
from __future__ import annotations

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


class HeadGroup:
    """One logit head group: weights [G, C, D] (G quantizers; G = 1 for the semantic nn.Linear head, which also has a bias)."""

    def __init__(self, name, weight, bias, idx, labels):
        self.name = name                               # weight-cache key
        self.weight, self.bias = weight, bias          # fp32 parameters
        self.idx = idx                                 # int32 [G, Rg] rows of hn (or -1)
        self.labels = labels                           # int64 [G, Rg] (-1 = ignore)  | None for logits-only use


def _pack_head(w):
    """fp32 [G, C, D] -> (bf16 [G, C, D], bf16 [G, D, Cpad])."""
    G, C, D = w.shape
    Cp = (C + 7) // 8 * 8
    Wb = torch.empty((G, Cp, D), dtype=BF16, device=w.device)
    WT = torch.empty((G, D, Cp), dtype=BF16, device=w.device)
    for g in range(G):
        ops.pack_weight(w[g], Wb[g], WT[g], rows_pad=Cp, cols_pad=D)
    return Wb[:, :C], WT


def head_logits(hn, w3, bias, idx, cache, key):
    """-> (hg bf16 [G*Rg, D], logits fp32 [G*Rg, Cpad] (first C columns valid))."""
    G, C, D = w3.shape
    Rg = idx.shape[1]
    Cp = (C + 7) // 8 * 8
    Wb, _ = cache.get(key, w3, _pack_head)
    hg = ops.gather_rows(hn, idx.reshape(-1))
    logits = torch.empty((G, Rg, Cp), dtype=F32, device=hn.device)
    ops.gemm_nt(hg.view(G, Rg, D), Wb, logits[:, :, :C], bias=bias)
    return hg, logits.view(G * Rg, Cp)


class HeadsLossFn(torch.autograd.Function):
    """(hn, groups) -> one CE loss SUM per group (the wrappers turn sums into the reference's weighted means)."""

    @staticmethod
    def forward(ctx, hn, groups, cache, *params):
        saved, outs = [], []
        pi = 0
        for gi, g in enumerate(groups):
            w = params[pi].detach(); pi += 1
            b = None
            if g.bias is not None:
                b = params[pi].detach(); pi += 1
            w3 = w if w.dim() == 3 else w.unsqueeze(0)
            C = w3.shape[1]
            hg, logits = head_logits(hn, w3, b, g.idx, cache, ('head', g.name))
            labels = g.labels.reshape(-1).contiguous()
            loss_rows, lse = ops.cross_entropy_fwd(logits, labels, C)
            outs.append(ops.reduce_sum(loss_rows))
            saved.append((w3, b is not None, hg, logits, lse, labels))
        ctx.saved, ctx.groups, ctx.cache, ctx.params, ctx.hn_shape = saved, groups, cache, params, hn.shape
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        M, D = ctx.hn_shape
        dev = gouts[0].device if gouts[0] is not None else ctx.saved[0][2].device
        dhn = torch.zeros((M, D), dtype=BF16, device=dev)
        grads = []
        for gi, (g, (w3, has_bias, hg, logits, lse, labels), go) in enumerate(zip(ctx.groups, ctx.saved, gouts)):
            G, C, _ = w3.shape
            Rg = g.idx.shape[1]
            Cp = logits.shape[1]
            if go is None:
                grads.append(None)
                if has_bias:
                    grads.append(None)
                continue
            gs = go.detach().to(F32).contiguous()
            dl = ops.cross_entropy_bwd(logits, labels, lse, gs, C, Cp)            # bf16 [G*Rg, Cp], pad columns zero
            _, WT = ctx.cache.get(('head', g.name), w3, _pack_head)
            dhg = torch.empty((G, Rg, D), dtype=BF16, device=dev)
            ops.gemm_nt(dl.view(G, Rg, Cp), WT, dhg)                              # dgrad: dlogits @ W
            ops.scatter_rows(dhg.view(G * Rg, D), g.idx.reshape(-1), dhn)
            dW = torch.empty((G, C, D), dtype=F32, device=dev)
            for q in range(G):                                                    # wgrad: dlogits_q^T @ hidden_q
                ops.gemm_tn_splitk(dl[q * Rg:(q + 1) * Rg, :C], hg[q * Rg:(q + 1) * Rg], dW[q])
            grads.append(dW.reshape(ctx.params[len(grads)].shape))
            if has_bias:
                grads.append(ops.colsum(dl[:, :C]))
        ctx.saved = None
        return (dhn, None, None, *grads)
