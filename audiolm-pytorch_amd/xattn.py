"""Attention over a short, always-visible key set -- the conditioning paths of the reference's Attention.forward
(audiolm_pytorch.py:307-406): cross-attention layers ([null_kv | to_kv(context_norm(text embeds))], non-causal, :372-388) and
`cond_as_self_attn_prefix` (text embeds prepended to the causal self-attention keys, :330-345).

The key set is tens to a few hundred positions and shared by all heads (MQA), so its scores are small dense matrices: the three contractions
run on the bf16 MFMA GEMM (ops.gemm_nt / the TN split-K form), the row-wise softmax pieces are csrc/xattn.hip.  A causal self-attention part
(flash kernels) is merged through its log-sum-exp; with the JOINT lse / output, both parts' backward formulas stay exact (see xattn.hip).

The same pieces are the reference's MATH path for an arbitrary dense `attn_bias` tensor (attend.py:98-146): keys = the sequence itself, bias fp32
[H, N, Me] added to the scaled scores, causal rule e <= n + (Me - N) -- O(N^2) memory like the reference's math path (relpos.DenseBias).

Layouts: q / o / do bf16 [B*N, H*dh] (row (b n), column (h d)) == [B, N*H, dh]; ke / ve bf16 [B, Me, dh]; statistics fp32 [B, H, N].
"""
from __future__ import annotations

import torch

from . import _lib, ops

BF16, F32 = torch.bfloat16, torch.float32


def _pad8(n):
    return (n + 7) // 8 * 8


def _transpose_pad(t, Mp):
    """bf16 [B, Me, dh] -> [B, dh, Mp] (zero-filled pad columns): the K-contiguous operand of `P @ V` / `dS @ K`"""
    B, Me, dh = t.shape
    out = torch.empty((B, dh, Mp), dtype=BF16, device=t.device)
    assert t.stride(2) == 1
    _lib.call('alm_transpose_bf16_batched', t.data_ptr(), out.data_ptr(), Me, dh, t.stride(1), Mp, Mp, B, t.stride(0), dh * Mp, ops._st())   # one launch for the batch
    return out


NOT_CAUSAL = 0x7fffffff


def _keep_mask(shape, p, device):
    """0 / 1 keep mask (bf16) of the attention dropout on this key set's probabilities (attend.py:140).  Module-level: the parity test pins it."""
    return torch.empty(shape, dtype=BF16, device=device).bernoulli_(1. - p)


def extra_attn_fwd(q, ke, ve, emask, B, N, H, dh, scale, o_self=None, lse_self=None, bias=None, causal=False, dropout_p=0.):
    """-> (o bf16 [B*N, H*dh], lse_tot fp32 [B, H, N], saved) ; emask uint8 [B, Me] (1 = attend) | None; bias fp32 [H, N, >= Me] contiguous | None
    (dense attn_bias, added to the scaled scores); causal: key e visible to query n iff e <= n + (Me - N) (attend.py:131-134).
    dropout_p > 0 (training): the probabilities of this key set are multiplied by a drawn 0 / 1 mask, the 1 / (1 - p) factor rides on the fp32 alpha
    of the value GEMM (and of its two backward GEMMs); the softmax normalisation (lse) is untouched, as in the reference (dropout AFTER softmax)."""
    dev = q.device
    Me = ke.shape[1]
    Mp = _pad8(Me)
    ke, ve = ke.contiguous(), ve.contiguous()
    S = torch.empty((B, N * H, Mp), dtype=F32, device=dev)
    ops.gemm_nt(q.view(B, N * H, dh), ke, S[:, :, :Me])
    P = torch.empty((B * N * H, Mp), dtype=BF16, device=dev)
    lse = torch.empty((B, H, N), dtype=F32, device=dev)
    fself = torch.empty(B * N * H, dtype=F32, device=dev) if o_self is not None else None
    if bias is not None:
        assert bias.dtype == F32 and bias.is_contiguous() and bias.dim() == 3 and bias.shape[0] == H and bias.shape[1] == N and bias.shape[2] >= Me, tuple(bias.shape)
    _lib.call('alm_xattn_softmax_fwd', S.data_ptr(), Mp, ops._p(emask), ops._p(lse_self), float(scale), P.data_ptr(), Mp, lse.data_ptr(), ops._p(fself),
              ops._p(bias), bias.shape[2] if bias is not None else 0, (Me - N) if causal else NOT_CAUSAL, B, N, H, Me, ops._st())
    veT = _transpose_pad(ve, Mp)
    Oe = torch.empty((B, N * H, dh), dtype=F32, device=dev)
    keep, da, Pd = None, 1.0, P
    if dropout_p > 0.:
        keep, da = _keep_mask(P.shape, dropout_p, dev), 1. / (1. - dropout_p)
        Pd = P * keep
    ops.gemm_nt(Pd.view(B, N * H, Mp), veT, Oe, alpha=da)
    o = torch.empty((B * N, H * dh), dtype=BF16, device=dev)
    _lib.call('alm_xattn_combine', ops._p(o_self), o_self.stride(0) if o_self is not None else 0, ops._p(fself), Oe.data_ptr(), o.data_ptr(), H * dh,
              B * N, H, dh, ops._st())
    return o, lse, dict(P=P, ke=ke, ve=ve, Me=Me, Mp=Mp, keep=keep, da=da, Pd=Pd)


def extra_attn_bwd(q, dout, saved, ndelta, B, N, H, dh, scale, dq=None, dbias=None):
    """dout bf16 [B*N, H*dh]; ndelta fp32 [B, H, N] = -rowsum(dO o O_joint).  dq given (the flash backward's bf16 dQ of the self part): the
    extra part is ACCUMULATED into it; else a new dq is returned.  dbias fp32 [H, N, Me] (contiguous) given: receives the gradient of a dense
    attn_bias (sum over the batch of the un-scaled dS).  -> (dq bf16, dke fp32 [B, Me, dh], dve fp32 [B, Me, dh])."""
    dev = q.device
    P, ke, ve, Me, Mp = saved['P'], saved['ke'], saved['ve'], saved['Me'], saved['Mp']
    keep, da, Pd = saved.get('keep'), saved.get('da', 1.0), saved.get('Pd', P)
    dP = torch.empty((B, N * H, Mp), dtype=F32, device=dev)
    ops.gemm_nt(dout.view(B, N * H, dh), ve, dP[:, :, :Me], alpha=da)
    if keep is not None:
        dP.mul_(keep.view(B, N * H, Mp))                              # d(loss)/dP through the dropout: keep / (1 - p)
    dS = torch.empty((B * N * H, Mp), dtype=BF16, device=dev)
    _lib.call('alm_xattn_softmax_bwd', P.data_ptr(), Mp, dP.data_ptr(), Mp, ndelta.data_ptr(), float(scale), dS.data_ptr(), Mp, Me, B, N, H, ops._st())
    if dbias is not None:
        assert dbias.dtype == F32 and dbias.is_contiguous() and tuple(dbias.shape) == (H, N, Me)
        _lib.call('alm_xattn_dbias', P.data_ptr(), Mp, dP.data_ptr(), Mp, ndelta.data_ptr(), dbias.data_ptr(), Me, Me, B, N, H, ops._st())
    keT = _transpose_pad(ke, Mp)
    acc = dq is not None
    if dq is None:
        dq = torch.empty((B * N, H * dh), dtype=BF16, device=dev)
    ops.gemm_nt(dS.view(B, N * H, Mp), keT, dq.view(B, N * H, dh), accumulate=acc)
    K = N * H
    dke = torch.empty((B, Mp, dh), dtype=F32, device=dev)
    dve = torch.empty((B, Mp, dh), dtype=F32, device=dev)
    ops._splitk('alm_gemm_bf16_tn_splitk', dS.view(B, K, Mp), q.view(B, K, dh), dke, Mp, dh, K, B, K * Mp, K * dh, Mp * dh, 1.0, False)
    ops._splitk('alm_gemm_bf16_tn_splitk', Pd.view(B, K, Mp), dout.view(B, K, dh), dve, Mp, dh, K, B, K * Mp, K * dh, Mp * dh, da, False)
    return dq, dke[:, :Me], dve[:, :Me]


def attn_delta(o, dout, B, N, H, dh):
    """ndelta fp32 [B, H, N] = -rowsum(dO o O) (pure cross-attention: no flash backward fills it)"""
    nd = torch.empty((B, H, N), dtype=F32, device=o.device)
    _lib.call('alm_xattn_delta', o.data_ptr(), o.stride(0), dout.data_ptr(), dout.stride(0), nd.data_ptr(), B, N, H, dh, ops._st())
    return nd
