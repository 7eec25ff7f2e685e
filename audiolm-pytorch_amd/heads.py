"""Logit heads + cross-entropy as ONE autograd node (reference audiolm_pytorch.py:719, :957-983, :1319-1361 for the logits and
:1561-1565, :1826-1854, :2112-2137 for the losses).

The reference computes 'q c d, b n q d -> b n q c' on a (b, n, q, d) view plus a remainder einsum with W[:r].  Position i of a
head's token range always uses W[i mod Q] (the remainder rows continue the same pattern), so here the final hidden states are
REGROUPED per quantizer ('b (n q) d -> q (b n) d', a row gather driven by an int32 index built from integer bookkeeping) and every
head group becomes one batched MFMA GEMM over q with fp32 logits, followed by an online-softmax cross-entropy.  Rows that do not
exist for a quantizer (the ragged tail) are padded with index -1 / label -1 and contribute nothing.

Precision: the logits are where the loss is read off, so the heads do NOT round their operands to bf16.  The final hidden states arrive in
fp32 (the final LayerNorm writes fp32) and the fp32 master weights are used: both are split into bf16 pairs x = hi + lo
(alm_gather_split_bf16, ~16 mantissa bits) and logits = hi.Whi + hi.Wlo + lo.Whi runs as three accumulating bf16 MFMA GEMMs with fp32
accumulation (the dropped lo.Wlo term is ~2^-16 relative).  The heads are < 2 % of the model FLOPs.  Backward (dgrad / wgrad) uses the
bf16 halves hi / Whi, like every other GEMM of the path.
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


_PAD_IDX = {}


def _pad_index(G, C, Cp, device):
    """int32 [G * Cp]: row g * Cp + c of the padded head image <- row g * C + c of the flat [G * C, D] weight, -1 (zero row) for c >= C (shape-only: cached)"""
    key = (G, C, Cp, device.type, device.index)
    if key not in _PAD_IDX:
        c = torch.arange(Cp, device=device)
        idx = torch.where(c[None, :] < C, torch.arange(G, device=device)[:, None] * C + c[None, :], torch.full((1, 1), -1, device=device))
        _PAD_IDX[key] = idx.reshape(-1).to(torch.int32).contiguous()
    return _PAD_IDX[key]


# Round 5: the three accumulating GEMMs of the split-bf16 logits (hi.Whi + hi.Wlo + lo.Whi) are ONE GEMM over a K-concatenated pair of operands
# ([hi | hi | lo] . [Whi | Wlo | Whi]^T, K = 3 D): the same products summed in one fp32 accumulator chain, one fast (non-accumulating) epilogue instead of
# two read-modify-write passes over the fp32 logits, one launch instead of three.  ALM_HEAD_KCAT=0 restores the three launches (A/B switch).
import os as _os
HEAD_KCAT = _os.environ.get('ALM_HEAD_KCAT', '1') != '0'


def _pack_head(w):
    """fp32 [G, C, D] -> (Wcat bf16 [G, C, 3 D] = [Whi | Wlo | Whi] (w ~= Whi + Wlo; rows of one quantizer Cpad apart), WT bf16 [G, D, Cpad] = Whi^T zero
    padded).  Whi / Wlo are the column blocks [0, D) / [D, 2 D) of Wcat.  Three launches for all G quantizers: one split-gather straight into the column
    blocks of the row-padded [G, Cpad, 3 D] image, one block copy (the second Whi), one multi-weight transpose pack."""
    G, C, D = w.shape
    Cp = (C + 7) // 8 * 8
    w = w.contiguous()
    capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
    idx = _pad_index(G, C, Cp, w.device) if not capturing else None
    Wcat = torch.empty((G, Cp, 3 * D), dtype=BF16, device=w.device)
    flat = Wcat.view(G * Cp, 3 * D)
    if idx is not None:
        ops.gather_split(w.view(G * C, D), idx, out=(flat[:, :D], flat[:, D:2 * D]))      # rows C .. Cp-1 of every quantizer are zero
    else:                                                                      # (index tensors are not created inside a hipGraph capture)
        for g in range(G):
            ops.gather_split(w[g], None, rows_out=Cp, out=(Wcat[g, :, :D], Wcat[g, :, D:2 * D]))
    flat[:, 2 * D:].copy_(flat[:, :D])
    WT = torch.empty((G, D, Cp), dtype=BF16, device=w.device)
    for g0 in range(0, G, 8):
        ops.pack_weights_multi([(w[g], None, WT[g], Cp, D) for g in range(g0, min(G, g0 + 8))])
    return Wcat[:, :C], WT


def _full_rows(Wcat, Cp):
    """the [G, Cpad, 3 D] image behind the [G, C, 3 D] view _pack_head returns (rows C .. Cpad-1 are zero)"""
    G, C, K3 = Wcat.shape
    return Wcat.as_strided((G, Cp, K3), (Wcat.stride(0), Wcat.stride(1), 1))


def head_logits(hn, w3, bias, idx, cache, key):
    """hn fp32 [M, D] (bf16 accepted: then there is no low half) -> (hg bf16 [G*Rg, D] (row stride may exceed D) = high halves of the gathered rows,
    logits fp32 [G*Rg, Cpad] (first C columns valid))."""
    G, C, D = w3.shape
    Rg = idx.shape[1]
    Cp = (C + 7) // 8 * 8
    Wcat, _ = cache.get(key, w3, _pack_head)
    Whi, Wlo = Wcat[:, :, :D], Wcat[:, :, D:2 * D]
    logits = torch.empty((G, Rg, Cp), dtype=F32, device=hn.device)
    out = logits[:, :, :C]
    if hn.dtype == F32 and HEAD_KCAT:
        Acat = torch.empty((G * Rg, 3 * D), dtype=BF16, device=hn.device)
        hg, _ = ops.gather_split(hn, idx.reshape(-1), out=(Acat[:, :D], Acat[:, 2 * D:]))
        Acat[:, D:2 * D].copy_(hg)
        if bias is None:                # all Cpad columns (the pad rows of Wcat are zero): N % 4 == 0 -> the GEMM's 16-byte row-segment epilogue
            ops.gemm_nt(Acat.view(G, Rg, 3 * D), _full_rows(Wcat, Cp), logits)
        else:
            ops.gemm_nt(Acat.view(G, Rg, 3 * D), Wcat, out, bias=bias)
    elif hn.dtype == F32:
        hg, hl = ops.gather_split(hn, idx.reshape(-1))
        ops.gemm_nt(hg.view(G, Rg, D), Whi, out, bias=bias)
        ops.gemm_nt(hg.view(G, Rg, D), Wlo, out, accumulate=True)
        ops.gemm_nt(hl.view(G, Rg, D), Whi, out, accumulate=True)
    else:
        hg = ops.gather_rows(hn, idx.reshape(-1))
        ops.gemm_nt(hg.view(G, Rg, D), Whi, out, bias=bias)
        ops.gemm_nt(hg.view(G, Rg, D), Wlo, out, accumulate=True)
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
        ctx.saved, ctx.groups, ctx.cache, ctx.params, ctx.hn_shape, ctx.hn_dtype = saved, groups, cache, params, hn.shape, hn.dtype
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        M, D = ctx.hn_shape
        dev = gouts[0].device if gouts[0] is not None else ctx.saved[0][2].device
        f32 = ctx.hn_dtype == F32                  # fp32 hidden states in -> fp32 gradient out (what autograd expects; no conversion passes)
        dhn = torch.zeros((M, D), dtype=ctx.hn_dtype, device=dev)
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
            dhg = torch.empty((G, Rg, D), dtype=ctx.hn_dtype, device=dev)
            ops.gemm_nt(dl.view(G, Rg, Cp), WT, dhg)                              # dgrad: dlogits @ W
            if f32:                                                               # row copies are type-blind: fp32 rows = bf16 rows of twice the width
                ops.scatter_rows(dhg.view(G * Rg, D).view(BF16), g.idx.reshape(-1), dhn.view(BF16))
            else:
                ops.scatter_rows(dhg.view(G * Rg, D), g.idx.reshape(-1), dhn)
            dW = torch.empty((G, C, D), dtype=F32, device=dev)
            # wgrad: dlogits_q^T @ hidden_q for all G quantizers in ONE batched launch (round 5; one split-K launch + reduce per quantizer before)
            ops.gemm_tn_batched(dl.view(1, G, Rg, Cp)[..., :C], hg.as_strided((1, G, Rg, D), (0, Rg * hg.stride(0), hg.stride(0), 1)), dW.view(1, G, C, D))
            grads.append(dW.reshape(ctx.params[len(grads)].shape))
            if has_bias:
                grads.append(ops.colsum(dl[:, :C]))
        ctx.saved = None
        return (dhn, None, None, *grads)


class CombineLossFn(torch.autograd.Function):
    """(per-group cross-entropy SUMS, their label tensors, weights) -> loss = sum_g w_g * sum_g / max(#valid labels_g, 1): the means of F.cross_entropy
    (ignore_index) per head and the wrappers' weighted combination (audiolm_pytorch.py:1561-1565, :1826-1854, :2112-2137) as one kernel, one more in
    the backward -- instead of ne / sum / clamp / div per group and mul / add / div on top, each with its autograd twin."""

    @staticmethod
    def forward(ctx, labels, weights, ignore_index, *sums):
        loss, scales = ops.loss_combine([s.detach() for s in sums], labels, weights, ignore_index)
        ctx.save_for_backward(scales)
        ctx.G = len(sums)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        (scales,) = ctx.saved_tensors
        d = scales * go.to(F32)
        return (None, None, None, *[d[g] for g in range(ctx.G)])


def combine_losses(sums, labels, weights, ignore_index=-1):
    return CombineLossFn.apply(list(labels), tuple(float(w) for w in weights), ignore_index, *sums)
