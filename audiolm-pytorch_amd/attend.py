"""Mirror of the reference's `audiolm_pytorch/attend.py` (Attend, attend.py:35-146) on the MI355X flash-MQA kernels.

forward(q (b h n d), k (b n d), v (b n d), mask (b n) bool | None, attn_bias) -> (b h n d); causal multi-query attention with
d == 64.  The math path's attn_bias (attend.py:121-122) is SURVEY.md §8(f) item 1 and is refused; dropout must be 0.
No (b, h, n, n) tensor is ever materialised and there is no CPU fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class AttendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask_u8):
        b, h, n, d = q.shape
        q2 = q.detach().permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        k2 = k.detach().reshape(b * n, d).to(torch.bfloat16).contiguous()
        v2 = v.detach().reshape(b * n, d).to(torch.bfloat16).contiguous()
        o, lse = ops.mqa_attn_fwd(q2, k2, v2, mask_u8, b, n, h, d)
        ctx.save_for_backward(q2, k2, v2, o, lse)
        ctx.mask, ctx.shape, ctx.dtypes = mask_u8, (b, h, n, d), (q.dtype, k.dtype, v.dtype)
        return o.view(b, n, h, d).permute(0, 2, 1, 3).to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q2, k2, v2, o, lse = ctx.saved_tensors
        b, h, n, d = ctx.shape
        do = dout.permute(0, 2, 1, 3).reshape(b * n, h * d).to(torch.bfloat16).contiguous()
        dq, dkv = ops.mqa_attn_bwd(q2, k2, v2, ctx.mask, o, lse, do, b, n, h, d)
        dkv = dkv.sum(0)                                   # per-head-group partials (summed by alm_kv_grad_pack on the fused path)
        dq = dq.view(b, n, h, d).permute(0, 2, 1, 3).to(ctx.dtypes[0])
        dk = dkv[:, :d].reshape(b, n, d).to(ctx.dtypes[1])
        dv = dkv[:, d:].reshape(b, n, d).to(ctx.dtypes[2])
        return dq, dk, dv, None


class Attend(nn.Module):
    def __init__(self, dropout=0., causal=False, flash=False):
        super().__init__()
        self.dropout = dropout
        self.attn_dropout = nn.Dropout(dropout)
        self.causal = causal
        self.register_buffer('mask', None, persistent=False)
        self.flash = flash

    def forward(self, q, k, v, mask=None, attn_bias=None):
        if attn_bias is not None:
            raise NotImplementedError('attention bias is SURVEY.md §8(f) item 1 (not fused yet)')
        if not self.causal:
            raise NotImplementedError('only causal attention is on the hot path (audiolm_pytorch.py:452)')
        if self.dropout != 0. and self.training:
            raise NotImplementedError('attention dropout > 0 is not implemented (reference default 0.)')
        if not q.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd.Attend runs on the MI355X only (no CPU fallback)')
        mask_u8 = None if mask is None else mask.to(torch.bool).contiguous().view(torch.uint8)
        return AttendFn.apply(q, k, v, mask_u8)
