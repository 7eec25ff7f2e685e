"""Structured attention bias of the `flash_attn=False` models, host side.

The reference builds a dense (heads, n, n) fp32 tensor by gathering rows of a small MLP output and adds it to the attention scores
(RelativePositionBias audiolm_pytorch.py:202-242; CoarseTransformerWrapper-side override :924-936; FineTransformer :1227-1298).  Here the
MLP output stays a per-head TABLE and the attention kernels index it in place (csrc/attention.hip):

    bias(h, i, j) = (qattr[i] & kattr[j]) ? tbl[h][0] : tbl[h][(qkey4[i] - kkey4[j]) / 4]

`AttnBias` carries the table (an autograd tensor produced by `PosTableFn`) and the four int32 index vectors; `Transformer.forward`
accepts it as `attn_bias`.  The table gradient comes back from the fused stack (core.TransformerStackFn) and flows through
`PosTableFn.backward` into the MLP parameters and the special-pair parameter (cross_attn_bias / null_pos_bias).
"""
from __future__ import annotations

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


class AttnBias:
    """tbl fp32 [H, LT] (raw-score units, slot 0 = special pairs); qkey4 / kkey4 / qattr / kattr int32 [N]."""

    def __init__(self, tbl, qkey4, kkey4, qattr, kattr):
        self.tbl, self.qkey4, self.kkey4, self.qattr, self.kattr = tbl, qkey4, kkey4, qattr, kattr

    def detached(self):
        out = AttnBias(self.tbl.detach(), self.qkey4, self.kkey4, self.qattr, self.kattr)
        out._host = getattr(self, '_host', None)
        return out

    def sliced(self, n):
        """the same bias restricted to the first n positions (prefix forward of a sampling run: slots do not depend on the length)"""
        return AttnBias(self.tbl, self.qkey4[:n], self.kkey4[:n], self.qattr[:n], self.kattr[:n])

    def _host_index(self):
        if getattr(self, '_host', None) is None:                       # one device -> host copy per sampling run, not per step
            self._host = (self.qkey4.cpu().tolist(), self.qattr.cpu().tolist())
        return self._host

    @property
    def qkey4_host(self):
        return self._host_index()[0]

    @property
    def qattr_host(self):
        return self._host_index()[1]


class DenseBias:
    """An arbitrary dense attn_bias of the reference's math path (attend.py:118-121): tbl fp32 [H, N, N] contiguous, added to the scaled scores.
    Attention then runs the O(N^2)-memory GEMM path of xattn.py instead of the flash kernels (the model family's own biases never need this:
    they stay structured, class AttnBias)."""

    def __init__(self, tbl):
        assert tbl.dim() == 3 and tbl.dtype == F32, (tbl.shape, tbl.dtype)
        self.tbl = tbl

    @staticmethod
    def wrap(attn_bias, H, N, Mk=None):
        """any tensor broadcastable to (H, N, Mk) -> DenseBias (keeps the autograd link of `attn_bias`)"""
        Mk = N if Mk is None else Mk
        t = attn_bias.to(F32)
        while t.dim() < 3:
            t = t.unsqueeze(0)
        if t.dim() == 4:                                               # (1, h, i, j)
            assert t.shape[0] == 1, 'a per-sample attn_bias is not supported (the reference adds one (h, i, j) bias to every sample)'
            t = t[0]
        return DenseBias(t.expand(H, N, Mk).contiguous())

    def detached(self):
        return DenseBias(self.tbl.detach())


class PosTableFn(torch.autograd.Function):
    """x fp32 [L, in] (in = 1 | 2), special [H] | None, weights (W0, b0, W1, b1, ..., Wout, bout) -> tbl fp32 [H, L + 1] = MLP(x)^T / scale
    with slot 0 = special / scale.  First / last layers and SiLU: csrc/relpos.hip; the C x C layers: bf16 MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, special, inv_scale, *weights):
        x = x.detach().contiguous().to(F32)
        ws = [w.detach().to(F32).contiguous() for w in weights]
        nhid = len(ws) // 2 - 2
        pres, acts, packed = [], [], []
        pre, act = ops.posmlp_in_fwd(x, ws[0], ws[1])
        pres.append(pre), acts.append(act)
        for k in range(nhid):
            W, b = ws[2 + 2 * k], ws[3 + 2 * k]
            C = W.shape[0]
            Wb = torch.empty((C, C), dtype=BF16, device=x.device)
            WbT = torch.empty((C, C), dtype=BF16, device=x.device)
            ops.pack_weight(W, Wb, WbT, rows_pad=C, cols_pad=C)
            packed.append(WbT)
            pre = torch.empty((x.shape[0], C), dtype=F32, device=x.device)
            ops.gemm_nt(act, Wb, pre, bias=b)
            act = ops.silu_fwd(pre)
            pres.append(pre), acts.append(act)
        sp = None if special is None else special.detach().reshape(-1).to(F32).contiguous()
        tbl = ops.posmlp_out_fwd(act, ws[-2], ws[-1], sp, inv_scale)
        ctx.x, ctx.ws, ctx.pres, ctx.acts, ctx.packed, ctx.inv_scale = x, ws, pres, acts, packed, inv_scale
        ctx.special_shape = None if special is None else special.shape
        ctx.wshapes = [w.shape for w in weights]
        return tbl

    @staticmethod
    def backward(ctx, dtbl):
        x, ws, pres, acts = ctx.x, ctx.ws, ctx.pres, ctx.acts
        nhid = len(ws) // 2 - 2
        H = ws[-2].shape[0]
        dev = x.device
        grads = [None] * len(ws)
        g, dpre, dsp = ops.posmlp_out_bwd(dtbl.contiguous().to(F32), ws[-2], pres[-1], ctx.inv_scale)
        dWout = torch.empty((g.shape[1], ws[-2].shape[1]), dtype=F32, device=dev)
        ops.gemm_tn_splitk(g, acts[-1], dWout)
        grads[-2], grads[-1] = dWout[:H], ops.colsum(g)[:H]
        for k in reversed(range(nhid)):
            C = ws[2 + 2 * k].shape[0]
            dW = torch.empty((C, C), dtype=F32, device=dev)
            ops.gemm_tn_splitk(dpre, acts[k], dW)                              # dW[out][in] = dpre^T @ act_below
            grads[2 + 2 * k], grads[3 + 2 * k] = dW, ops.colsum(dpre)
            dact = torch.empty((x.shape[0], C), dtype=F32, device=dev)
            ops.gemm_nt(dpre, ctx.packed[k], dact)                             # dact = dpre @ W
            dpre = ops.silu_bwd(dact, pres[k])
        grads[0], grads[1] = ops.posmlp_in_bwd(dpre, x)
        dspecial = None if ctx.special_shape is None else dsp.reshape(ctx.special_shape)
        return (None, dspecial, None, *[gr.reshape(s) for gr, s in zip(grads, ctx.wshapes)])


def _i32(t):
    return t.to(torch.int32).contiguous()


def toeplitz_index(n, device, num_leading=None):
    """Index vectors of RelativePositionBias(n, n) (audiolm_pytorch.py:229-241): table row i - j + n - 1 -> slot 1 + row.
    num_leading (Coarse, :929-936): pairs with exactly one position among the first `num_leading` are special; under the causal mask
    only (i >= num_leading, j < num_leading) can be attended, which is what the attribute bits encode."""
    pos = torch.arange(n, device=device)
    qkey4 = _i32(4 * (pos + n))
    kkey4 = _i32(4 * pos)
    if num_leading is None:
        qattr = kattr = torch.zeros(n, dtype=torch.int32, device=device)
    else:
        qattr = _i32(pos >= num_leading)
        kattr = _i32(pos < num_leading)
    return qkey4, kkey4, qattr, kattr


def fine_index(coarse_length, fine_length, num_coarse_quantizers, num_fine_quantizers, device):
    """Index vectors + MLP input grid of FineTransformer's (relative frame, relative quantizer) bias (audiolm_pytorch.py:1229-1298).
    Sequence = [coarse start][coarse tokens][fine start][fine tokens]; a pair involving a start token is special (null_pos_bias)."""
    Qc, Qf = num_coarse_quantizers, num_fine_quantizers
    cdiv = lambda a, b: -(-a // b)
    cs, fs = cdiv(coarse_length, Qc), cdiv(fine_length, Qf)
    M = max(cs, fs)
    Qt = Qc + Qf
    R = 2 * Qt - 1
    c_idx = torch.arange(coarse_length, device=device)
    f_idx = torch.arange(fine_length, device=device)
    c_key = (c_idx // Qc) * R + (c_idx % Qc)
    f_key = (f_idx // Qf) * R + (f_idx % Qf) + Qc
    # start tokens carry a neighbour's key: their pairs are special anyway, and a far-away key would widen the kernels' table windows
    c0 = c_key[:1] if coarse_length > 0 else torch.zeros(1, dtype=torch.long, device=device)
    f0 = f_key[:1] if fine_length > 0 else (c_key[-1:] if coarse_length > 0 else c0)
    key = torch.cat((c0, c_key, f0, f_key))
    is_start = torch.zeros(key.shape[0], dtype=torch.bool, device=device)
    is_start[0] = True
    is_start[coarse_length + 1] = True
    off = (M - 1) * R + (Qt - 1)
    qkey4 = _i32(4 * (key + off + 1))
    kkey4 = _i32(4 * key)
    qattr = _i32(is_start.to(torch.int32) | 2)                    # bit 0: the query is a start token; bit 1 pairs with "key is a start token"
    kattr = _i32(1 | (is_start.to(torch.int32) << 1))
    rel_seq = torch.arange(2 * M - 1, device=device).repeat_interleave(R)
    rel_off = torch.arange(R, device=device).repeat(2 * M - 1)
    grid = torch.stack((rel_seq, rel_off), dim=-1).float()        # :1262-1271
    return grid, (qkey4, kkey4, qattr, kattr)
