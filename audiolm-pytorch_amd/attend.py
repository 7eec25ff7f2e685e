"""Mirror of the reference's `audiolm_pytorch/attend.py` (Attend, attend.py:35-146) on the MI355X flash-MQA kernels.

forward(q (b h n d), k (b n d), v (b n d), mask (b n) bool | None, attn_bias) -> (b h n d); causal multi-query attention with
d == 64.  The math path's attn_bias (attend.py:118-121) is accepted in its STRUCTURED form (relpos.AttnBias: per-head table + index
vectors, what RelativePositionBias / the Coarse and Fine transformers build); an arbitrary dense (h, n, n) tensor is refused; dropout must
be 0.  No (b, h, n, n) tensor is ever materialised and there is no CPU fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops, relpos


class AttendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask_u8, bias, tbl):
        # bias: relpos.AttnBias | None; tbl = bias.tbl passed separately so that autograd routes the table gradient
        b, h, n, d = q.shape
        q2 = q.detach().permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        k2 = k.detach().reshape(b * n, d).to(torch.bfloat16).contiguous()
        v2 = v.detach().reshape(b * n, d).to(torch.bfloat16).contiguous()
        bias = bias.detached() if bias is not None else None
        o, lse = ops.mqa_attn_fwd(q2, k2, v2, mask_u8, b, n, h, d, bias=bias)
        ctx.save_for_backward(q2, k2, v2, o, lse)
        ctx.mask, ctx.shape, ctx.dtypes, ctx.bias = mask_u8, (b, h, n, d), (q.dtype, k.dtype, v.dtype), bias
        return o.view(b, n, h, d).permute(0, 2, 1, 3).to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q2, k2, v2, o, lse = ctx.saved_tensors
        b, h, n, d = ctx.shape
        do = dout.permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        part = ops.attn_bias_part(b, n, h, ctx.bias.tbl.shape[1], do.device) if ctx.bias is not None else None
        dq, dkv = ops.mqa_attn_bwd(q2, k2, v2, ctx.mask, o, lse, do, b, n, h, d, bias=ctx.bias, dtbl_part=part)
        dtbl = ops.attn_bias_grad_reduce(part, b, n, h, d) if ctx.bias is not None else None
        dkv = dkv.sum(0)                                   # per-head-group partials (summed by alm_kv_grad_pack on the fused path)
        dq = dq.view(b, n, h, d).permute(0, 2, 1, 3).to(ctx.dtypes[0])
        dk = dkv[:, :d].reshape(b, n, d).to(ctx.dtypes[1])
        dv = dkv[:, d:].reshape(b, n, d).to(ctx.dtypes[2])
        return dq, dk, dv, None, None, dtbl


class Attend(nn.Module):
    def __init__(self, dropout=0., causal=False, flash=False):
        super().__init__()
        self.dropout = dropout
        self.attn_dropout = nn.Dropout(dropout)
        self.causal = causal
        self.register_buffer('mask', None, persistent=False)
        self.flash = flash

    def forward(self, q, k, v, mask=None, attn_bias=None):
        if attn_bias is not None and not isinstance(attn_bias, relpos.AttnBias):
            raise NotImplementedError('a dense (h, n, n) attn_bias tensor is not supported: pass the structured relpos.AttnBias')
        if not self.causal:
            raise NotImplementedError('only causal attention is on the hot path (audiolm_pytorch.py:452)')
        if self.dropout != 0. and self.training:
            raise NotImplementedError('attention dropout > 0 is not implemented (reference default 0.)')
        if not q.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd.Attend runs on the MI355X only (no CPU fallback)')
        mask_u8 = None if mask is None else mask.to(torch.bool).contiguous().view(torch.uint8)
        return AttendFn.apply(q, k, v, mask_u8, attn_bias, attn_bias.tbl if attn_bias is not None else None)
