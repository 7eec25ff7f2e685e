"""TEST INFRASTRUCTURE ONLY (oracle).  Rounding-matched mode of the CPU restatement (round 3).

`oracle/audiolm_oracle.py` is the fp32 restatement of the reference.  The HIP path is NOT an fp32 program: like the reference under
`accelerator.autocast()` (trainer.py:1241) it feeds bf16 operands to its dense contractions and stores activations in bf16.  Against
the fp32 oracle its logits therefore sit at the bf16 noise floor (~1e-2), which cannot tell rounding from a small algorithmic error.  This module
restates the SAME algorithm (every function below calls or mirrors its fp32 twin and cites the same reference lines) but rounds to bf16 at exactly
the points where the HIP path stores or consumes bf16 -- fp32 accumulation everywhere, fp32 statistics -- so that the only differences left are
summation order and 1-ulp transcendental approximations.  The GPU tests then assert north_star's number against it: logits <= 1e-3 rel-Frobenius.

Rounding points (forward; `rst` = round-to-nearest-even bf16 whose GRADIENT is rounded too, because the HIP path stores the gradient of every bf16
activation in bf16 as well; `rf` = forward-only rounding):
  weights            rf(W)                 the packed bf16 copies W / W^T (core.layer_weights); weight gradients are fp32 GEMM outputs
  branch inputs      XN = rst(LN(x) g)     fp32 statistics on the fp32 branch input;  X = rst(x): the un-normalised copy feeding to_kv (:325)
  attention          Q, KV = rst(...)      v mix rf(0.5 (v + v0)) (:357-358);  scores S = K Q^T on bf16 operands, fp32;  ONLINE softmax over 64-key
                                           tiles with the kernel's deferred rescale (running maximum raised only when a tile exceeds it by 2^8, decided
                                           per 32-query wave), P = exp2(S c - m) rounded to bf16 as the operand of P V, row sums from the unrounded P,
                                           O / l rounded to bf16;  backward: P = exp2(S c - lse), dV = rf(P)^T dO, dS = P (dP - delta),
                                           dQ = scale rf(dS) K, dK = scale rf(dS)^T Q   (csrc/attention.hip)
  projections        Y = rst(AO rf(Wo)^T), U = rst(XN rf(W1)^T), HN = rst(LN(x gelu(gate)) g3), Y = rst(HN rf(W2)^T)
  residual           1 stream: fp32 (R + Y);  4 streams: fp32 arithmetic, stored fp32 or bf16 (`residual_bf16`: R = rst(R') after every depth
                     connection except the last, which feeds the fused stream sum + final LayerNorm unrounded)
  final LayerNorm    fp32 output;  logit heads: split-bf16 operands hi + lo (hi.Whi + hi.Wlo + lo.Whi), backward on bf16(dlogits) with the high halves
Not restated here (the fp32 oracle covers them): structured attention bias, conditioning.
"""
from __future__ import annotations

import contextlib
import math

import torch
import torch.nn.functional as F

import audiolm_oracle as O

LOG2E = 1.4426950408889634
RESCALE_THR = 8.0


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundST(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


rst = _RoundST.apply
rf = _RoundFwd.apply


def _c2(dh):
    return float(torch.tensor(float(dh) ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32))     # the kernel's fp32 constant


def flash_fwd_emul(q, k, v, mask):
    """csrc/attention.hip mqa_fwd_kernel, arithmetic only (reference attend.py:98-146).  q (b h n d), k / v (b n d): fp32 tensors holding bf16 values;
    mask bool (b n) | None -> (o (b h n d) fp32 = un-rounded O / l, lse (b h n))."""
    B, H, N, dh = q.shape
    c2 = _c2(dh)
    ninf = float('-inf')
    m = torch.full((B, H, N), ninf)
    l = torch.zeros((B, H, N))
    o = torch.zeros((B, H, N, dh))
    qi = torch.arange(N)
    for t in range((N + 63) // 64):
        k0, k1 = t * 64, min(N, t * 64 + 64)
        qs = k0                                                     # queries before the tile see none of its keys (and their waves skip it)
        s = torch.einsum('bhid,bjd->bhij', q[:, :, qs:], k[:, k0:k1])
        dead = (torch.arange(k0, k1)[None, :] > qi[qs:, None])[None, None]
        if mask is not None:
            dead = dead | ~mask[:, None, None, k0:k1]
        s = s.masked_fill(dead, ninf)
        tm = s.amax(dim=-1) * c2
        mc, lc, oc = m[:, :, qs:], l[:, :, qs:], o[:, :, qs:]
        nq = N - qs
        pad = (-nq) % 32
        need = F.pad(tm > mc + RESCALE_THR, (0, pad)).reshape(B, H, -1, 32).any(dim=-1, keepdim=True)
        need = need.expand(-1, -1, -1, 32).reshape(B, H, -1)[:, :, :nq]                                   # __any over the 32-query wave
        mn = torch.where(need, torch.maximum(mc, tm), mc)
        alpha = torch.where(mn == ninf, torch.ones_like(mn), torch.exp2(mc - mn))
        alpha = torch.where(need, alpha, torch.ones_like(alpha))
        ms = torch.where(mn == ninf, torch.zeros_like(mn), mn)
        p = torch.exp2((s.double() * c2 - ms.double()[..., None]).float())                                # fma(s, c2, -m) in the kernel
        m[:, :, qs:] = mn
        l[:, :, qs:] = lc * alpha + p.sum(dim=-1)
        o[:, :, qs:] = oc * alpha[..., None] + torch.einsum('bhij,bjd->bhid', _bf(p), v[:, k0:k1])
    inv = torch.where(l > 0, 1.0 / l, torch.zeros_like(l))
    lse = torch.where(l > 0, m / LOG2E + torch.log(l), torch.full_like(l, ninf))
    return o * inv[..., None], lse


def flash_bwd_emul(q, k, v, o, lse, do, mask):
    """attn_delta_kernel + mqa_bwd_dq_kernel + mqa_bwd_dkv_kernel, arithmetic only.  o: the bf16 forward output (as fp32), do: bf16 dO (as fp32)
    -> (dq (b h n d), dk (b n d), dv (b n d)), fp32, un-rounded"""
    B, H, N, dh = q.shape
    scale = float(dh) ** -0.5
    c2 = _c2(dh)
    delta = (do * o).sum(dim=-1)                                      # attn_delta_kernel: from the bf16 O and dO
    dq = torch.empty_like(q)
    dk = torch.zeros_like(k)
    dv = torch.zeros_like(v)
    ii = torch.arange(N)
    causal = (ii[None, :] > ii[:, None])[None]
    for b in range(B):
        s = torch.einsum('hid,jd->hij', q[b], k[b])
        dead = causal if mask is None else (causal | ~mask[b][None, None, :])
        lse2 = torch.where(lse[b] == float('-inf'), torch.full_like(lse[b], float('inf')), lse[b]).double() * LOG2E
        p = torch.exp2((s.double() * c2 - lse2[..., None]).float()).masked_fill(dead, 0.)
        dp = torch.einsum('hid,jd->hij', do[b], v[b])
        ds = p * (dp - delta[b][..., None])
        dv[b] = torch.einsum('hij,hid->jd', _bf(p), do[b])
        dsr = _bf(ds)
        dk[b] = scale * torch.einsum('hij,hid->jd', dsr, q[b])
        dq[b] = scale * torch.einsum('hij,jd->hid', dsr, k[b])
    return dq, dk, dv


class _FlashEmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask):
        out, lse = flash_fwd_emul(q, k, v, mask)
        ctx.save_for_backward(q, k, v, _bf(out), lse)
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = flash_bwd_emul(q, k, v, o, lse, _bf(do), ctx.mask)       # dAO is a bf16 GEMM output
        return dq, dk, dv, None


def attention(sd, p, x, xn_of, heads, mask=None, value_residual=None):
    """oracle.attention (audiolm_pytorch.py:307-406) at the HIP path's rounding points.  x: fp32 branch input; xn_of(x) -> the pre-LayerNorm'd
    input (the caller owns that LayerNorm because the 4-stream path fuses it with the width connection)."""
    b, n, _ = x.shape
    XN = rst(xn_of(x))
    X = rst(x)
    q = rst(F.linear(XN, rf(sd[p + 'to_q.weight'])))
    kv = rst(F.linear(X, rf(sd[p + 'to_kv.weight'])))
    k, v = kv.chunk(2, dim=-1)
    orig_v = v
    if value_residual is not None:
        v = rf(0.5 * (v + value_residual))
    q = q.reshape(b, n, heads, -1).transpose(1, 2)
    out = rst(_FlashEmul.apply(q, k, v, mask))
    out = out.transpose(1, 2).reshape(b, n, -1)
    return rst(F.linear(out, rf(sd[p + 'to_out.0.weight']))), orig_v


def feedforward(sd, p, x):
    """oracle.feedforward (audiolm_pytorch.py:246-260) at the HIP path's rounding points"""
    XN = rst(O.layer_norm(x, sd[p + '0.gamma']))
    u = rst(F.linear(XN, rf(sd[p + '1.weight'])))
    xh, gate = u.chunk(2, dim=-1)
    h = F.gelu(gate) * xh
    hn = rst(O.layer_norm(h, sd[p + '3.gamma']))
    return rst(F.linear(hn, rf(sd[p + '5.weight'])))


def make_transformer(residual_bf16):
    """-> a drop-in for oracle.transformer (audiolm_pytorch.py:461-560), flash / unconditioned models only"""

    def transformer(sd, p, x, *, depth, heads, streams=4, self_attn_mask=None, attn_bias=None, grad_shrink_alpha=0.1, add_value_residual=True,
                    context=None, context_mask=None, cond_as_self_attn_prefix=False, ff_keep=None, ff_dropout=0.):
        if attn_bias is not None or context is not None or ff_keep is not None or (p + 'rel_pos_bias.net.0.0.weight') in sd:
            raise NotImplementedError('rounding-matched oracle: flash / unconditioned models only')
        x = x * grad_shrink_alpha + x.detach() * (1 - grad_shrink_alpha)
        value_residual = None
        if streams > 1:
            x = x.repeat_interleave(streams, dim=0)
        nbranch, done = 2 * depth, 0

        def branch(pp, fn):
            nonlocal x, done
            done += 1
            if streams > 1:
                bi, Rp, beta = O.hc_width(sd, pp, x, streams)
                out, extra = fn(bi)
                x = O.hc_depth(out, Rp, beta)
                if residual_bf16 and done < nbranch:
                    x = rst(x)
            else:
                out, extra = fn(x)
                x = out + x
            return extra

        for l in range(depth):
            pa, pf = f'{p}layers.{l}.0.', f'{p}layers.{l}.2.'
            values = branch(pa, lambda t: attention(sd, pa + 'branch.', t, lambda u: O.layer_norm(u, sd[pa + 'branch.norm.gamma']), heads,
                                                    mask=self_attn_mask, value_residual=value_residual))
            if add_value_residual and value_residual is None:
                value_residual = values
            branch(pf, lambda t: (feedforward(sd, pf + 'branch.', t), None))
        if streams > 1:
            x = x.reshape(x.shape[0] // streams, streams, *x.shape[1:]).sum(dim=1)
        return O.layer_norm(x, sd[p + 'norm.gamma'])

    return transformer


class _HeadMatmul(torch.autograd.Function):
    """heads.head_logits / HeadsLossFn.backward: x (..., d) @ w (c, d)^T on split-bf16 operands; backward on bf16(dlogits) and the high halves"""

    @staticmethod
    def forward(ctx, x, w):
        hi, whi = _bf(x), _bf(w)
        lo, wlo = _bf(x - hi), _bf(w - whi)
        ctx.save_for_backward(hi, whi)
        return hi @ whi.t() + hi @ wlo.t() + lo @ whi.t()

    @staticmethod
    def backward(ctx, g):
        hi, whi = ctx.saved_tensors
        g = _bf(g)
        return g @ whi, g.reshape(-1, g.shape[-1]).t() @ hi.reshape(-1, hi.shape[-1])


def head_linear(x, w, b=None):
    out = _HeadMatmul.apply(x, w)
    return out if b is None else out + b


def grouped_logits(weights, pred, Q):
    """oracle._grouped_logits (Coarse :965-983, Fine-fine :1343-1361): position i uses W[i mod Q]"""
    n = pred.shape[1]
    outs = []
    for q in range(Q):
        if q < n:
            outs.append(_HeadMatmul.apply(pred[:, q::Q], weights[q]))
    lg = torch.zeros((pred.shape[0], n, weights.shape[1]))
    for q, t in enumerate(outs):
        lg[:, q::Q] = t
    return lg


@contextlib.contextmanager
def rounding_matched(residual_bf16=False):
    """inside: oracle.semantic_forward / coarse_forward / fine_forward (and the wrapper losses built on them) run at the HIP path's rounding points"""
    saved = (O.transformer, O._grouped_logits, O._padded_logits, O.head_linear)
    O.transformer, O._grouped_logits, O._padded_logits, O.head_linear = make_transformer(residual_bf16), grouped_logits, grouped_logits, head_linear
    try:
        yield
    finally:
        O.transformer, O._grouped_logits, O._padded_logits, O.head_linear = saved
