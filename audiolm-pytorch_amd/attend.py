"""Mirror of the reference's `audiolm_pytorch/attend.py` (Attend, attend.py:35-146) on the MI355X flash-MQA kernels.

forward(q (b h n d), k (b n d), v (b n d), mask (b n) bool | None, attn_bias) -> (b h n d); causal multi-query attention with
d == 64.  The math path's attn_bias (attend.py:118-121) in its STRUCTURED form (relpos.AttnBias: per-head table + index vectors, what
RelativePositionBias / the Coarse and Fine transformers build) runs on the flash kernels: no (b, h, n, n) tensor is ever materialised.
An arbitrary DENSE attn_bias tensor ((h, i, j) or broadcastable to it), and non-causal attention, take the reference's math path as three MFMA
GEMMs + the row-softmax kernels of csrc/xattn.hip (AttendMathFn: O(n^2) memory, like the reference).  dropout > 0 (training mode, attend.py:92 / :140):
the flash kernels decide keep / drop per (batch, head, query, key) with a stateless hash of a per-call seed (the same decision in forward, dQ and
dK/dV); the math path multiplies its probabilities with a drawn 0 / 1 mask.  No CPU fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops, relpos, xattn


class AttendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask_u8, bias, tbl, dropout_p=0., seed=0):
        # bias: relpos.AttnBias | None; tbl = bias.tbl passed separately so that autograd routes the table gradient
        b, h, n, d = q.shape
        q2 = q.detach().permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        k2 = k.detach().reshape(b * n, d).to(torch.bfloat16).contiguous()
        v2 = v.detach().reshape(b * n, d).to(torch.bfloat16).contiguous()
        bias = bias.detached() if bias is not None else None
        o, lse = ops.mqa_attn_fwd(q2, k2, v2, mask_u8, b, n, h, d, bias=bias, dropout_p=dropout_p, seed=seed)
        ctx.drop = (float(dropout_p), int(seed))
        ctx.save_for_backward(q2, k2, v2, o, lse)
        ctx.mask, ctx.shape, ctx.dtypes, ctx.bias = mask_u8, (b, h, n, d), (q.dtype, k.dtype, v.dtype), bias
        return o.view(b, n, h, d).permute(0, 2, 1, 3).to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q2, k2, v2, o, lse = ctx.saved_tensors
        b, h, n, d = ctx.shape
        do = dout.permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        part = ops.attn_bias_part(b, n, h, ctx.bias.tbl.shape[1], do.device) if ctx.bias is not None else None
        dq, dkv = ops.mqa_attn_bwd(q2, k2, v2, ctx.mask, o, lse, do, b, n, h, d, bias=ctx.bias, dtbl_part=part, dropout_p=ctx.drop[0], seed=ctx.drop[1])
        dtbl = ops.attn_bias_grad_reduce(part, b, n, h, d) if ctx.bias is not None else None
        dkv = dkv.sum(0)                                   # per-head-group partials (summed by alm_kv_grad_pack on the fused path)
        dq = dq.view(b, n, h, d).permute(0, 2, 1, 3).to(ctx.dtypes[0])
        dk = dkv[:, :d].reshape(b, n, d).to(ctx.dtypes[1])
        dv = dkv[:, d:].reshape(b, n, d).to(ctx.dtypes[2])
        return dq, dk, dv, None, None, dtbl, None, None


class AttendMathFn(torch.autograd.Function):
    """attend.py:98-146 (flash=False): sim = q k^T * scale (+ attn_bias), key mask, causal triu(j - i + 1), softmax, attn v -- multi-query
    (k / v (b j d) shared by the heads).  q (b h i d), dense bias fp32 (h, i, j) | None."""

    @staticmethod
    def forward(ctx, q, k, v, mask_u8, bias, causal, dropout_p=0.):
        b, h, n, d = q.shape
        m = k.shape[1]
        q2 = q.detach().permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        k2 = k.detach().to(torch.bfloat16).contiguous()
        v2 = v.detach().to(torch.bfloat16).contiguous()
        bd = None if bias is None else bias.detach().to(torch.float32).expand(h, n, m).contiguous()
        o, lse, saved = xattn.extra_attn_fwd(q2, k2, v2, mask_u8, b, n, h, d, float(d) ** -0.5, bias=bd, causal=causal, dropout_p=float(dropout_p))
        ctx.save_for_backward(q2, o)
        ctx.saved, ctx.shape, ctx.dtypes, ctx.m = saved, (b, h, n, d), (q.dtype, k.dtype, v.dtype), m
        ctx.bias_meta = None if bias is None else (bias.shape, bias.dtype, bias.requires_grad)
        return o.view(b, n, h, d).permute(0, 2, 1, 3).to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q2, o = ctx.saved_tensors
        b, h, n, d = ctx.shape
        do = dout.permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        nd = xattn.attn_delta(o, do, b, n, h, d)
        dbias = None
        if ctx.bias_meta is not None and ctx.bias_meta[2]:
            dbias = torch.empty((h, n, ctx.m), dtype=torch.float32, device=do.device)
        dq, dke, dve = xattn.extra_attn_bwd(q2, do, ctx.saved, nd, b, n, h, d, float(d) ** -0.5, dbias=dbias)
        if dbias is not None:
            shape, dtype, _ = ctx.bias_meta
            dbias = dbias.sum_to_size(shape) if tuple(shape) != (h, n, ctx.m) else dbias
            dbias = dbias.to(dtype)
        return dq.view(b, n, h, d).permute(0, 2, 1, 3).to(ctx.dtypes[0]), dke.to(ctx.dtypes[1]), dve.to(ctx.dtypes[2]), None, dbias, None, None


class Attend(nn.Module):
    def __init__(self, dropout=0., causal=False, flash=False):
        super().__init__()
        self.dropout = dropout
        self.attn_dropout = nn.Dropout(dropout)
        self.causal = causal
        self.register_buffer('mask', None, persistent=False)
        self.flash = flash

    def forward(self, q, k, v, mask=None, attn_bias=None):
        pd = float(self.dropout) if self.training else 0.
        if not q.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd.Attend runs on the MI355X only (no CPU fallback)')
        mask_u8 = None if mask is None else mask.to(torch.bool).contiguous().view(torch.uint8)
        dense = attn_bias is not None and not isinstance(attn_bias, relpos.AttnBias)
        if dense or not self.causal or k.shape[1] != q.shape[2]:
            if isinstance(attn_bias, relpos.AttnBias):
                raise NotImplementedError('the structured relpos.AttnBias belongs to causal self-attention (the flash kernels)')
            if dense:
                while attn_bias.dim() > 3:                          # (1, h, i, j)
                    assert attn_bias.shape[0] == 1, 'one (h, i, j) bias for every sample (attend.py:118-121)'
                    attn_bias = attn_bias[0]
            return AttendMathFn.apply(q, k, v, mask_u8, attn_bias if dense else None, self.causal, pd)
        from . import core
        return AttendFn.apply(q, k, v, mask_u8, attn_bias, attn_bias.tbl if attn_bias is not None else None, pd, core._attn_seed() if pd > 0. else 0)
