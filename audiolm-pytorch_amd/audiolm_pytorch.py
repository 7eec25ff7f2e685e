"""Host-side mirror of the reference's `audiolm_pytorch/audiolm_pytorch.py` for the training hot path.

Same class names, constructor / forward signatures, attribute names and `state_dict` key names & shapes as the reference
(SURVEY.md §8(b)), so that these classes drop in under the reference's trainer.py.  The modules are
parameter containers + integer bookkeeping; every floating-point op of the path runs in libaudiolm_hip.so on the MI355X:

  Transformer.forward                       -> core.TransformerStackFn   (one autograd node, explicit HIP launch sequences)
  *Transformer.forward embeddings           -> EmbedAssembleFn           (gather + quantizer-position add + start tokens + concat)
  logit heads + F.cross_entropy             -> heads.HeadsLossFn         (regrouped batched MFMA GEMM + online-softmax CE)

  RelativePositionBias / pos_bias_mlp       -> relpos.PosTableFn         (`flash_attn=False` models: the bias stays a per-head table that
                                                                          the attention kernels index in place; no (h, n, n) tensor)

  *TransformerWrapper.generate()            -> sample_logits(): prefix once through the training forward path, then one single-position pass per
                                               token over a per-layer k/v cache (core.DecodeCache, alm_mqa_decode_attn); sampling helpers in torch

Conditioning (`has_condition=True`, reference :325-375, :450-455, :640-668, :818-855, :1097-1134): cross-attention layers with a null key / value,
`cond_as_self_attn_prefix`, per-sample condition dropping and classifier-free guidance all run natively from pre-computed `text_embeds`
(xattn.py / csrc/xattn.hip); the T5 text encoder is out of scope (`text=` raises).
The reference's stacked kv_cache= / embed_cache= TENSOR protocol is accepted on forward() / forward_with_cond_scale()
(Transformer.forward_kv_protocol: the one-new-token step runs the same single-position kernels); generate() drives the native cache directly.
An arbitrary dense `attn_bias` tensor takes the reference's O(n^2) math path (relpos.DenseBias, xattn.py) instead of the flash kernels.
attn_dropout > 0 and ff_dropout > 0 are supported (training mode; in-kernel keep decisions for the flash attention, drawn 0 / 1 masks elsewhere).
Attention / FeedForward / GEGLU also run STANDALONE (outside Transformer) on the same kernels, un-fused.
Waveform reconstruction (SoundStream decoder) is native: soundstream.py.
There is NO CPU or eager-PyTorch fallback for the hot path: CPU tensors are refused.
"""
from __future__ import annotations

import dataclasses
import functools
import os
from functools import partial
from pathlib import Path

import torch
import torch.nn.functional as F
from torch import nn

from . import core, heads, ops, relpos
from .attend import Attend
from .version import __version__

DEFAULT_T5_NAME = 'google/t5-v1_1-base'
FUSED_PREPARE = os.environ.get('ALM_FUSED_PREPARE', '1') != '0'      # Coarse wrapper's training step / FineTransformer's assembly: id bookkeeping as one kernel (A/B switch, tests)
DECODE_GRAPH = os.environ.get('ALM_DECODE_GRAPH', '1') != '0'        # capture the single-position sampling step into a hipGraph
_T5_DIMS = {'google/t5-v1_1-small': 512, 'google/t5-v1_1-base': 768, 'google/t5-v1_1-large': 1024,
            'google/t5-v1_1-xl': 2048, 'google/t5-v1_1-xxl': 4096, 't5-small': 512, 't5-base': 768, 't5-large': 1024}


def get_encoded_dim(name):
    """reference t5.py:get_encoded_dim needs the HF hub; text conditioning is out of scope, only the width is needed so that
    `proj_text_embed` keeps the reference's shape in the state_dict (audiolm_pytorch.py:604-605)."""
    return _T5_DIMS.get(name, 768)


# ---------------------------------------------------------------------------------------------- helpers (audiolm_pytorch.py:40-186)

def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else d


def ceil_div(numer, denom):
    return (numer + denom - 1) // denom


def round_down_nearest_multiple(val, mult):
    return (val // mult) * mult


def _forgetful_drop_(keep, mask_prob):
    """keep (bool [..., n]) &= forgetful mask: per row, the int(n * mask_prob) keys (at most n - 1) with the largest Gaussian draw are dropped; key 0
    (the start token) never.  ONE randn(shape) draw on keep's device, like the reference (audiolm_pytorch.py:82-89), so a seeded run masks the same
    keys.  On the GPU the selection is one kernel (alm_forgetful_mask) instead of ATen's topk + sort + scatter + fills."""
    n = keep.shape[-1]
    drop_count = min(int(n * mask_prob), n - 1)
    score = torch.randn(keep.shape, device=keep.device)
    if drop_count <= 0:
        return keep
    if keep.is_cuda and keep.dim() == 2 and keep.is_contiguous() and n <= ops.FORGETFUL_MAX_N:
        return ops.forgetful_mask_(keep, score, drop_count)
    score[..., 0] = -torch.finfo(score.dtype).max                # never among the top draws
    return keep.scatter_(-1, score.topk(drop_count, dim=-1).indices, False)


def _forgetful_and_(mask, mask_prob):
    """mask & generate_mask_with_prob(mask.shape, mask_prob, mask.device) without the ones / and passes (tests swap generate_mask_with_prob for a
    recorded mask: honoured)"""
    if generate_mask_with_prob is not _GENERATE_MASK:
        return mask & generate_mask_with_prob(mask.shape, mask_prob, mask.device)
    return _forgetful_drop_(mask, mask_prob)


def generate_mask_with_prob(shape, mask_prob, device):
    """Forgetful causal mask (reference audiolm_pytorch.py:82-89): True = key kept."""
    return _forgetful_drop_(torch.ones(shape, dtype=torch.bool, device=device), mask_prob)


_GENERATE_MASK = generate_mask_with_prob


def grad_shrink(t, alpha=0.1):
    """forward identity, gradient scaled by alpha (audiolm_pytorch.py:93-94).  API compatibility only: the fused stack applies the factor to
    the input gradient itself (core.stack_backward)."""
    return torch.lerp(t.detach(), t, alpha)


def append_eos_id(ids, eos_id):                               # semantics of audiolm_pytorch.py:155-160
    """(b, n) int64 -> (b, n + 1): one eos column on the right (one cat with a cached column: F.pad is a fill + a copy)"""
    if ids.dim() == 2 and ids.dtype == torch.int64:
        return torch.cat((ids, _const_ids(int(eos_id), ids.shape[0], ids.device)), dim=1)
    return F.pad(ids, (0, 1), value=eos_id)


def batch_unique_consecutive(t, pad_value=0.):                # semantics of audiolm_pytorch.py:162-164
    """per row: collapse runs of equal ids; rows are ragged afterwards and right-padded with pad_value to the longest one
    (data-dependent width -> a host loop over the batch, as in the reference)"""
    if t.is_cuda and t.dtype == torch.int64 and t.dim() == 2 and t.numel() > 0:
        # one launch + ONE host read of the row lengths (ops.unique_consecutive) instead of a synchronising torch.unique_consecutive per row
        out, lengths = ops.unique_consecutive(t if t.stride(1) == 1 else t.contiguous(), None, int(pad_value))
        return out[:, :int(lengths.max())].contiguous()
    rows = [torch.unique_consecutive(row) for row in t]
    out = t.new_full((len(rows), max(r.numel() for r in rows)), pad_value)
    for i, r in enumerate(rows):
        out[i, :r.numel()] = r
    return out


# sampling helpers (semantics of audiolm_pytorch.py:96-130): integer / tiny (batch, vocab) bookkeeping of the generate() loops, in torch ops

def log(t, eps=1e-20):
    """log with a floor"""
    return (t + eps).log()


def gumbel_noise(t):
    """-log(-log u), u ~ U(0, 1): one in-place uniform draw of t's shape (the reference's RNG consumption)"""
    u = torch.empty_like(t).uniform_(0, 1)
    return log(log(u).neg()).neg()


def gumbel_sample(t, temperature=1., dim=-1):
    """Gumbel-max draw from softmax(t / temperature)"""
    return (t / temperature + gumbel_noise(t)).argmax(dim=dim)


def top_k(logits, thres=0.5):
    """each row keeps exactly k = max(int((1 - thres) * n), 1) largest logits, the rest become -inf"""
    k = max(int((1 - thres) * logits.shape[-1]), 1)
    vals, idx = logits.topk(k, dim=-1)
    return torch.full_like(logits, float('-inf')).scatter(-1, idx, vals)


def mask_out_after_eos_id(t, eos_id, mask_value=-1, keep_eos=True):
    """everything after the first eos of a row becomes mask_value -- the eos itself too unless keep_eos"""
    from_eos = (t == eos_id).cumsum(dim=-1) > 0                # at or after the first eos
    if keep_eos:
        from_eos = F.pad(from_eos[..., :-1], (1, 0), value=False)   # strictly after it
    return t.masked_fill(from_eos, mask_value)


def all_rows_have_eos_id(t, eos_id):
    return (t == eos_id).any(dim=-1).all()


def eval_decorator(fn):                                        # semantics of audiolm_pytorch.py:71-78
    """run a module method in eval mode and put the previous training flag back afterwards (also when it raises)"""
    @functools.wraps(fn)
    def run_in_eval(model, *args, **kwargs):
        mode = model.training
        model.eval()
        try:
            return fn(model, *args, **kwargs)
        finally:
            model.train(mode)
    return run_in_eval


def get_embeds(embeddings: nn.Embedding, codes: torch.Tensor, pad_id=-1, return_mask=False, mask_pad_pos_to=0):
    """Pad-aware embedding lookup with the semantics of audiolm_pytorch.py:168-186: positions holding pad_id read row 0 and are then overwritten
    with mask_pad_pos_to (None: left as row 0); optionally also returns the not-pad mask.  Exported for API compatibility only -- the
    transformers never call it (their lookup is alm_embed_assemble, which applies the same rule inside the kernel)."""
    is_pad = codes.eq(pad_id)
    rows = embeddings(torch.where(is_pad, torch.zeros_like(codes), codes))
    if mask_pad_pos_to is not None:
        rows = torch.where(is_pad[..., None], torch.full_like(rows, mask_pad_pos_to), rows)
    return (rows, ~is_pad) if return_mask else rows


def _flatten_ids(t):
    return t.reshape(t.shape[0], -1)


def prob_mask_like(shape, prob, device):                      # semantics of audiolm_pytorch.py:144-150 (same RNG consumption)
    """bool mask, True with probability `prob`; the degenerate probabilities draw nothing"""
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    if prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


class LinearNoBiasFn(torch.autograd.Function):
    """y = x W^T on the bf16 MFMA GEMM (fp32 in / out): `proj_text_embed` (audiolm_pytorch.py:605 / :777 / :1043) -- the one dense layer of the
    conditioning path outside the fused stack."""

    @staticmethod
    def forward(ctx, x, w, cache, key):
        shape = x.shape
        x2 = x.detach().reshape(-1, shape[-1]).to(torch.bfloat16).contiguous()
        W, WT = cache.get(key, w.detach(), core._pack_plain)
        y = torch.empty((x2.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        ops.gemm_nt(x2, W, y)
        ctx.x2, ctx.WT, ctx.shape, ctx.wshape, ctx.need_dx = x2, WT, shape, w.shape, x.requires_grad
        return y.view(*shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).contiguous()
        dw = torch.empty(ctx.wshape, dtype=torch.float32, device=dy.device)
        ops.gemm_tn_splitk(dy2, ctx.x2, dw)
        dx = None
        if ctx.need_dx:
            dx = torch.empty((dy2.shape[0], ctx.wshape[1]), dtype=torch.float32, device=dy.device)
            ops.gemm_nt(dy2, ctx.WT, dx)
            dx = dx.view(ctx.shape)
        return dx, dw, None, None


# ---------------------------------------------------------------------------------------------- parameter containers

class LayerNorm(nn.Module):                                   # audiolm_pytorch.py:191-198
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))

    def forward(self, x):
        shape = x.shape
        y, _, _, _ = LayerNormFn.apply(x.reshape(-1, shape[-1]), self.gamma)
        return y.reshape(shape)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma):
        xc = x.detach().contiguous()
        if xc.dtype not in (torch.float32, torch.bfloat16):
            xc = xc.float()
        y, _, mean, rstd = ops.layernorm_fwd(xc, gamma.detach())
        ctx.save_for_backward(xc, mean, rstd, gamma)
        ctx.mark_non_differentiable(mean, rstd)
        return y, None, mean, rstd

    @staticmethod
    def backward(ctx, dy, *_):
        x, mean, rstd, gamma = ctx.saved_tensors
        dx, dg = ops.layernorm_bwd(dy.contiguous().to(torch.bfloat16), x, mean, rstd, gamma.detach(), dx_dtype=torch.float32)
        return dx.to(x.dtype), dg


class RelativePositionBias(nn.Module):                        # audiolm_pytorch.py:202-242
    def __init__(self, *, dim, heads, layers=3):
        super().__init__()
        self.net = nn.ModuleList([])
        self.net.append(nn.Sequential(nn.Linear(1, dim), nn.SiLU()))
        for _ in range(layers - 1):
            self.net.append(nn.Sequential(nn.Linear(dim, dim), nn.SiLU()))
        self.net.append(nn.Linear(dim, heads))

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, i, j, special=None, num_leading=None):
        """-> relpos.AttnBias: the (2j - 1)-row MLP table (:234-238) kept as a table + the index vectors that replace the gather
        `x[rel_pos]` (:229-241); the (h, i, j) tensor is never built.  special / num_leading: the Coarse cross-attention override (:929-936)."""
        if i != j:
            raise NotImplementedError('RelativePositionBias(i, j) with i < j (cached decoding) is not used here: sampling builds the table for the full length once')
        dev = self.device
        x = torch.arange(-j + 1, j, device=dev).float().unsqueeze(-1)                             # :234-235
        ws = []
        for layer in self.net:
            lin = layer[0] if isinstance(layer, nn.Sequential) else layer
            ws += [lin.weight, lin.bias]
        heads_ = self.net[-1].weight.shape[0]
        tbl = relpos.PosTableFn.apply(x, special, 64 ** 0.5, *ws)                                 # scores are scaled by dim_head^-0.5 = 1 / 8
        assert tbl.shape[0] == heads_
        return relpos.AttnBias(tbl, *relpos.toeplitz_index(j, dev, num_leading))


class GEGLUFn(torch.autograd.Function):
    """x, gate = chunk(2, dim=-1); gelu(gate) * x  (audiolm_pytorch.py:246-249), fp32 in / out, exact-erf GELU: csrc/norm_act.hip alm_geglu_fwd / bwd"""

    @staticmethod
    def forward(ctx, x):
        shape = x.shape
        x2 = x.detach().reshape(-1, shape[-1]).float().contiguous()
        ctx.save_for_backward(x2)
        ctx.shape, ctx.dtype = shape, x.dtype
        return ops.geglu_fwd(x2).view(*shape[:-1], shape[-1] // 2).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x2, = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).float().contiguous()
        return ops.geglu_bwd(dy2, x2).view(ctx.shape).to(ctx.dtype)


class GEGLU(nn.Module):                                       # audiolm_pytorch.py:246-249 (inside Transformer it is fused into alm_geglu_ln_*)
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd runs on the MI355X only (no CPU fallback)')
        return GEGLUFn.apply(x)


class FeedForwardFn(torch.autograd.Function):
    """FeedForward.forward OUTSIDE the fused stack (audiolm_pytorch.py:251-260): LayerNorm -> W1 -> GEGLU + LayerNorm -> Dropout -> W2 on the same
    kernels and launch sequence the stack uses (core._run_ff / core.ff_backward), fp32 in / out."""

    @staticmethod
    def forward(ctx, x, seq, p_drop, g0, w1, g3, w2):
        shape = x.shape
        D = shape[-1]
        x2 = x.detach().reshape(-1, D).float().contiguous()
        I = w2.shape[1]
        cfg = core.StackCfg(dim=D, depth=1, heads=1, dim_head=64, streams=1, inner=I, add_value_residual=False, grad_shrink_alpha=1.)
        Ip = cfg.inner_pad
        W = dict(w1=seq._cache.get('w1', w1, lambda w: core._pack_w1(w, I, Ip)), w2=seq._cache.get('w2', w2, lambda w: core._pack_w2(w, I, Ip)))
        XN, _, mean, rstd = ops.layernorm_fwd(x2, g0.detach())
        prm = dict(ln3=g3.detach())
        Y, sv = core._run_ff(cfg, W, prm, XN, x2.shape[0], float(p_drop))
        ctx.cfg, ctx.W, ctx.prm, ctx.sv = cfg, W, prm, dict(sv, XN=XN)
        ctx.save_for_backward(x2, mean, rstd, g0)
        ctx.shape, ctx.dtype = shape, x.dtype
        return Y.float().view(shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, g0 = ctx.saved_tensors
        dY = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).contiguous()
        dXN, dW1, dg3, dW2 = core.ff_backward(ctx.cfg, ctx.W, ctx.prm, ctx.sv, dY, None)
        dx, dg0 = ops.layernorm_bwd(dXN, x2, mean, rstd, g0.detach())
        return dx.view(ctx.shape).to(ctx.dtype), None, None, dg0, dW1, dg3, dW2


class FeedForwardSeq(nn.Sequential):
    """the reference's nn.Sequential(LayerNorm, Linear, GEGLU, LayerNorm, Dropout, Linear) -- same children, same state_dict keys -- whose forward
    runs the fused kernels instead of walking the children through ATen"""

    def __init__(self, *mods):
        super().__init__(*mods)
        self._cache = core.WeightCache()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd runs on the MI355X only (no CPU fallback)')
        p = self[4].p if self.training else 0.
        return FeedForwardFn.apply(x, self, p, self[0].gamma, self[1].weight, self[3].gamma, self[5].weight)


def FeedForward(dim, mult=4, dropout=0.1):                    # audiolm_pytorch.py:251-260
    inner_dim = int(dim * 2 * mult / 3)
    return FeedForwardSeq(
        LayerNorm(dim),
        nn.Linear(dim, inner_dim * 2, bias=False),
        GEGLU(),
        LayerNorm(inner_dim),
        nn.Dropout(dropout),
        nn.Linear(inner_dim, dim, bias=False)
    )


class _MaskMulFn(torch.autograd.Function):
    """y = x * keep / (1 - p): nn.Dropout with the mask drawn by core._dropout_keep (so that tests can pin it)"""

    @staticmethod
    def forward(ctx, x, p):
        keep = core._dropout_keep(x.shape, p, x.device).to(x.dtype)
        ctx.save_for_backward(keep)
        ctx.s = 1. / (1. - p)
        return x * keep * ctx.s

    @staticmethod
    def backward(ctx, dy):
        keep, = ctx.saved_tensors
        return dy * keep * ctx.s, None


class Attention(nn.Module):                                   # audiolm_pytorch.py:264-406
    """Parameter container of the fused stack (Transformer reads to_q / to_kv / to_out / norm from here) AND a working module of its own: forward()
    below is the reference's Attention.forward (:307-406) composed, un-fused, from the same kernels -- LayerNorm (alm_layernorm_*), the bias-free
    projections (bf16 MFMA GEMMs, LinearNoBiasFn), Attend (flash-MQA kernels / the math path of csrc/xattn.hip) -- with torch only gluing views."""

    def __init__(self, dim, causal=False, dim_head=64, dim_context=None, heads=8, norm_context=False, num_null_kv=0,
                 dropout=0.1, scale=8, flash=False):
        super().__init__()
        self.heads = heads
        self.dim_head = dim_head
        self.causal = causal
        inner_dim = dim_head * heads
        dim_context = default(dim_context, dim)
        self.norm = LayerNorm(dim)
        self.context_norm = LayerNorm(dim_context) if norm_context else nn.Identity()
        self.attn_dropout = nn.Dropout(dropout)
        self.num_null_kv = num_null_kv
        self.null_kv = nn.Parameter(torch.randn(2, num_null_kv, dim_head)) if num_null_kv > 0 else None
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim_context, dim_head * 2, bias=False)
        self.attend = Attend(flash=flash, dropout=dropout, causal=causal)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim, bias=False), nn.Dropout(dropout))
        self._cache = core.WeightCache()

    def forward(self, x, context=None, mask=None, attn_bias=None, prefix_context=None, prefix_context_mask=None, return_kv_cache=False,
                return_values=False, value_residual=None, kv_cache=None):
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd runs on the MI355X only (no CPU fallback)')
        b, n, _ = x.shape
        device = x.device
        if exists(context):
            context = self.context_norm(context)                                                   # :322-323
        kv_input = default(context, x)                                                             # :325 (bound BEFORE the pre-norm)
        if exists(prefix_context):                                                                 # :330-345
            kv_input = torch.cat((prefix_context, kv_input), dim=-2)
            m = prefix_context.shape[-2]
            if not exists(mask):
                mask = torch.ones((b, n), device=device, dtype=torch.bool)
            mask = torch.cat((prefix_context_mask, mask), dim=-1) if exists(prefix_context_mask) else F.pad(mask, (m, 0), value=True)
            if exists(attn_bias):
                attn_bias = F.pad(attn_bias, (m, 0), value=0.)
        xn = self.norm(x)                                                                          # :347
        q = LinearNoBiasFn.apply(xn, self.to_q.weight, self._cache, 'to_q')                        # :351
        k, v = LinearNoBiasFn.apply(kv_input, self.to_kv.weight, self._cache, 'to_kv').chunk(2, dim=-1)
        orig_v = v
        if exists(value_residual):                                                                 # :357-358
            v = 0.5 * (v + value_residual)
        if exists(kv_cache):                                                                       # :362-366
            ck, cv = kv_cache
            k, v = torch.cat((ck, k), dim=-2), torch.cat((cv, v), dim=-2)
        if return_kv_cache:
            kv_cache = torch.stack((k, v))                                                         # :370
        if self.num_null_kv > 0:                                                                   # :372-376
            nk, nv = self.null_kv[0], self.null_kv[1]
            k = torch.cat((nk.expand(b, -1, -1).to(k.dtype), k), dim=-2)
            v = torch.cat((nv.expand(b, -1, -1).to(v.dtype), v), dim=-2)
        q = q.reshape(b, n, self.heads, -1).transpose(1, 2)                                        # 'b n (h d) -> b h n d'
        if exists(mask):
            mask = F.pad(mask, (self.num_null_kv, 0), value=True)                                  # :384-385
        out = self.attend(q, k, v, attn_bias=attn_bias, mask=mask)                                 # :389
        out = out.transpose(1, 2).reshape(b, n, -1)                                                # 'b h n d -> b n (h d)'
        out = LinearNoBiasFn.apply(out, self.to_out[0].weight, self._cache, 'to_out')
        pd = self.to_out[1].p
        if self.training and pd > 0.:
            out = _MaskMulFn.apply(out, pd)                                                        # the nn.Dropout behind to_out (:304)
        if not return_kv_cache and not return_values:
            return out
        if return_kv_cache and not return_values:
            return out, kv_cache
        if return_values and not return_kv_cache:
            return out, orig_v
        return out, (kv_cache, orig_v)


class RMSNorm(nn.Module):                                     # hyper-connections stream norm (gamma init 0)
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.zeros(dim))


class HyperConnections(nn.Module):
    """Parameter container with the third-party module's state_dict names."""

    def __init__(self, num_residual_streams, *, dim, branch=None, layer_index=None):
        super().__init__()
        from random import randrange
        self.branch = branch
        self.norm = RMSNorm(dim)
        self.num_residual_streams = s = num_residual_streams
        init_residual_index = default(layer_index, randrange(s)) % s
        self.static_beta = nn.Parameter(torch.ones(s))
        init_alpha0 = torch.zeros((s, 1))
        init_alpha0[init_residual_index, 0] = 1.
        self.static_alpha = nn.Parameter(torch.cat([init_alpha0, torch.eye(s)], dim=1))
        self.dynamic_alpha_fn = nn.Parameter(torch.zeros(dim, s + 1))
        self.dynamic_alpha_scale = nn.Parameter(torch.ones(()) * 1e-2)
        self.dynamic_beta_fn = nn.Parameter(torch.zeros(dim))
        self.dynamic_beta_scale = nn.Parameter(torch.ones(()) * 1e-2)

    def hc_params(self):   # order == core.HC_KEYS
        return [self.static_beta, self.static_alpha, self.dynamic_alpha_fn, self.dynamic_alpha_scale, self.dynamic_beta_fn,
                self.dynamic_beta_scale, self.norm.gamma]


class Residual(nn.Module):                                    # num_residual_streams == 1 wrapper (same `.branch.` prefix)
    def __init__(self, *args, branch=None, **kwargs):
        super().__init__()
        self.branch = branch

    def hc_params(self):
        return []


# ---------------------------------------------------------------------------------------------- Transformer (audiolm_pytorch.py:410-560)

class Transformer(nn.Module):
    def __init__(self, *, dim, depth, heads, dim_context=None, cross_attend=False, attn_dropout=0., ff_dropout=0.,
                 grad_shrink_alpha=0.1, cond_as_self_attn_prefix=False, rel_pos_bias=True, flash_attn=False,
                 add_value_residual=True, num_residual_streams=4, residual_dtype=None, **kwargs):
        # residual_dtype (extension): HBM storage of the hyper-connection residual streams and their gradients, torch.float32 | torch.bfloat16 | None.
        # bf16 is what trainer.py:1241's autocast gives the reference (its streams are bf16 tensors from the first width connection on) and halves the
        # traffic of the HBM-bound hyper-connection kernels; arithmetic is fp32 either way.  None (default) = ALM_RESIDUAL_DTYPE, whose default `auto`
        # follows the caller like the reference does: bf16 inside torch.autocast(bfloat16), fp32 outside (core.default_residual_bf16).
        super().__init__()
        rel_pos_bias = rel_pos_bias and not flash_attn
        assert not (cross_attend and cond_as_self_attn_prefix)
        assert 0. <= ff_dropout < 1. and 0. <= attn_dropout < 1.
        self.ff_dropout = float(ff_dropout)
        self.attn_dropout = float(attn_dropout)          # Attention(dropout=): attention probabilities (attend.py:92 / :140) + the Dropout behind to_out (:304)
        self.dim = dim
        self.depth = depth
        self.heads = heads
        self.dim_context = default(dim_context, dim)
        self.cond_as_self_attn_prefix = cond_as_self_attn_prefix
        self.grad_shrink_alpha = grad_shrink_alpha
        self.grad_shrink = partial(grad_shrink, alpha=grad_shrink_alpha)
        self.num_residual_streams = num_residual_streams
        self.layers = nn.ModuleList([])
        self.rel_pos_bias = RelativePositionBias(dim=dim // 2, heads=heads) if rel_pos_bias else None
        hc = partial(HyperConnections, num_residual_streams) if num_residual_streams > 1 else partial(Residual, num_residual_streams)
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                hc(dim=dim, branch=Attention(dim=dim, heads=heads, dropout=attn_dropout, flash=flash_attn, causal=True, **kwargs)),
                hc(dim=dim, branch=Attention(dim=dim, heads=heads, dropout=attn_dropout, dim_context=dim_context, flash=flash_attn, num_null_kv=1,
                                             norm_context=True, **kwargs)) if cross_attend else None,          # :450
                hc(dim=dim, branch=FeedForward(dim=dim, dropout=ff_dropout))
            ]))
        self.norm = LayerNorm(dim)
        self.add_value_residual = add_value_residual
        attn0 = self.layers[0][0].branch
        self.cfg = core.StackCfg(dim=dim, depth=depth, heads=heads, dim_head=attn0.dim_head, streams=num_residual_streams,
                                 inner=int(dim * 2 * 4 / 3), add_value_residual=add_value_residual,
                                 grad_shrink_alpha=grad_shrink_alpha,
                                 residual_bf16=bool(core.default_residual_bf16() if residual_dtype is None else residual_dtype == torch.bfloat16),
                                 cross_attend=cross_attend, prefix=cond_as_self_attn_prefix, dim_context=self.dim_context)
        self._residual_auto = residual_dtype is None and core.default_residual_bf16() is None      # storage follows the caller's autocast state
        self.cross_attend = cross_attend
        if cond_as_self_attn_prefix:
            assert self.dim_context == dim, 'cond_as_self_attn_prefix feeds the context through the self-attention to_kv: dim_context must equal dim'
        assert residual_dtype in (None, torch.float32, torch.bfloat16), residual_dtype
        self._cache = core.WeightCache()
        self._layer_grad_hook = None          # set by parallel.DataParallelEngine: called as each layer's grads become final
        self.micro_batches = 1                # 2: two half-batches on two HIP streams inside the stack (core.TransformerStackFn; graphed.GraphedTrainStep sets it)

    def flat_params(self):
        """Parameter order consumed by core.stack_forward / stack_backward."""
        out = []
        for attn, cross, ff in self.layers:
            a, f = attn.branch, ff.branch
            out += attn.hc_params() + [a.norm.gamma, a.to_q.weight, a.to_kv.weight, a.to_out[0].weight]
            if cross is not None:
                c = cross.branch
                out += cross.hc_params() + [c.norm.gamma, c.context_norm.gamma, c.null_kv, c.to_q.weight, c.to_kv.weight, c.to_out[0].weight]
            out += ff.hc_params() + [f[0].gamma, f[1].weight, f[3].gamma, f[5].weight]
        out.append(self.norm.gamma)
        return out

    def prepack_weights(self):
        """(re)builds the bf16 packed copies of every dense weight of the stack now (they are otherwise built lazily, layer by layer, by the first
        forward that finds a master weight changed).  graphed.GraphedTrainStep calls this ahead of forking its two half-batch streams."""
        flat = [t.detach() for t in self.flat_params()]
        cfg = self.cfg
        if core.PACK_ALL:                                          # the forward's own form: every stale layer in one launch
            core.pack_stack_weights(self._cache, flat, cfg)
            return
        ppl = core.params_per_layer(cfg.streams, cfg.cross_attend)
        for l in range(cfg.depth):
            core.layer_weights(self._cache, l, core._split_layer(flat[l * ppl:(l + 1) * ppl], cfg.streams, cfg.cross_attend), cfg.inner, cfg.inner_pad)

    def forward(self, x, self_attn_mask=None, context=None, context_mask=None, attn_bias=None, return_kv_cache=False, kv_cache=None,
                return_flat_hidden=False):
        if exists(kv_cache) or return_kv_cache:                 # the reference's cache protocol: inference only (the outputs carry no autograd graph)
            out, new_cache = self.forward_kv_protocol(x, self_attn_mask=self_attn_mask, context=context, context_mask=context_mask,
                                                      attn_bias=attn_bias, kv_cache=kv_cache)
            return (out, new_cache) if return_kv_cache else out
        assert not (self.cond_as_self_attn_prefix and not exists(context))                          # :471
        assert exists(context) == (self.cross_attend or self.cond_as_self_attn_prefix), 'a conditioning context needs (and is needed by) a conditioned Transformer'
        assert not (exists(context) and context.shape[-1] != self.dim_context), \
            f'you had specified a conditioning dimension of {self.dim_context}, yet what was received by the transformer has dimension of {context.shape[-1]}'
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd runs on the MI355X only: move the model and its inputs to cuda (no CPU fallback)')
        b, n, d = x.shape
        if exists(attn_bias) and not isinstance(attn_bias, (relpos.AttnBias, relpos.DenseBias)):
            # an arbitrary dense bias (reference :500-503 takes any tensor): the O(n^2) math path (xattn.py); the model family's own biases are
            # structured (RelativePositionBias.forward / FineTransformer build relpos.AttnBias) and stay on the flash kernels
            attn_bias = relpos.DenseBias.wrap(attn_bias, self.cfg.heads, n)
        if not exists(attn_bias) and exists(self.rel_pos_bias):
            attn_bias = self.rel_pos_bias(n, n)                                                   # :500-503
        mask_u8 = None
        if exists(self_attn_mask):
            mask_u8 = self_attn_mask.to(torch.bool).contiguous().view(torch.uint8)
        opts = dict(hook=self._layer_grad_hook, grad=torch.is_grad_enabled(), micro=self.micro_batches, context_mask=context_mask,
                    ff_dropout=self.ff_dropout if self.training else 0., attn_dropout=self.attn_dropout if self.training else 0.)
        cfg = self.cfg
        if self._residual_auto and cfg.streams > 1 and cfg.residual_bf16 != core.autocast_bf16():
            cfg = dataclasses.replace(cfg, residual_bf16=not cfg.residual_bf16)
        hn = core.TransformerStackFn.apply(x, mask_u8, cfg, self._cache, opts, attn_bias,
                                           attn_bias.tbl if exists(attn_bias) else None, context, *self.flat_params())
        if return_flat_hidden:
            return hn                                           # bf16 [b*n, d] (feeds heads.HeadsLossFn)
        return hn.view(b, n, d)

    def forward_kv_protocol(self, x, self_attn_mask=None, context=None, context_mask=None, attn_bias=None, kv_cache=None):
        """see _forward_kv_protocol.  The outputs carry NO autograd graph, whereas the reference's Coarse / Fine forward requests the cache internally
        and stays differentiable: code that passes return_kv_cache / return_cache while TRAINING would silently get detached logits, so a call in
        training mode with autograd on is refused, and an eval-mode call with autograd on (inference without no_grad) warns once."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self.training:
                raise RuntimeError('audiolm_pytorch_amd: the kv_cache / return_kv_cache (return_cache) tensor protocol is inference-only -- its outputs are '
                                   'detached from autograd.  Train without the cache arguments, or call it under torch.no_grad() / .eval()')
            if not Transformer._warned_protocol_grad:
                Transformer._warned_protocol_grad = True
                import warnings
                warnings.warn('audiolm_pytorch_amd: kv_cache / return_kv_cache outputs are detached from autograd (inference-only protocol); wrap '
                              'sampling code in torch.no_grad() to silence this', stacklevel=3)
        return self._forward_kv_protocol(x, self_attn_mask=self_attn_mask, context=context, context_mask=context_mask, attn_bias=attn_bias,
                                         kv_cache=kv_cache)

    _warned_protocol_grad = False

    @torch.no_grad()
    def _forward_kv_protocol(self, x, self_attn_mask=None, context=None, context_mask=None, attn_bias=None, kv_cache=None):
        """The reference's kv-cache TENSOR protocol (audiolm_pytorch.py:360-370 store / concat, :487-496 `x = x[:, cache_len:]`, :560 stack):
        x fp32 [b, n, d] = embeddings of the WHOLE sequence, kv_cache [depth, 2 (k | v), b, cache_len, dim_head] | None
        -> (final-norm hidden states of the positions cache_len .. n-1, fp32 [b, n - cache_len, d];  new kv_cache fp32 [depth, 2, b, n, dim_head]).
        The cached v is the value-residual-mixed one, as in the reference (:357-366).  Inference only (no autograd graph).
        One new position (the sampling case): the tensor is loaded into a core.DecodeCache and the single-position kernels run
        (alm_mqa_decode_attn).  No cache, or more than one new position: the whole sequence is recomputed -- the same numbers, the reference's
        incremental saving only exists for the one-token step.  `cond_as_self_attn_prefix` ignores an incoming cache like the reference (:481)."""
        assert not (self.cond_as_self_attn_prefix and not exists(context))
        assert exists(context) == (self.cross_attend or self.cond_as_self_attn_prefix), 'a conditioning context needs (and is needed by) a conditioned Transformer'
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd runs on the MI355X only: move the model and its inputs to cuda (no CPU fallback)')
        b, n, d = x.shape
        cfg, dh = self.cfg, self.cfg.dim_head
        cache_len = 0
        if exists(kv_cache) and not self.cond_as_self_attn_prefix:
            assert tuple(kv_cache.shape[:3]) == (cfg.depth, 2, b) and kv_cache.shape[-1] == dh, \
                f'kv_cache must be [depth={cfg.depth}, 2, b={b}, cached positions, dim_head={dh}], got {tuple(kv_cache.shape)}'
            cache_len = kv_cache.shape[-2]
            assert cache_len <= n
        dense = exists(attn_bias) and not isinstance(attn_bias, relpos.AttnBias)
        if dense:
            # an arbitrary dense attn_bias (h, n, n): the math path has no single-position kernel, so the sequence is recomputed from scratch on
            # every call (`cache_len` only selects the positions that are returned) -- the same numbers the reference's cached step gives
            attn_bias = attn_bias if isinstance(attn_bias, relpos.DenseBias) else relpos.DenseBias.wrap(attn_bias, cfg.heads, n)
        if not exists(attn_bias) and exists(self.rel_pos_bias):
            attn_bias = self.rel_pos_bias(n, n)
        tbl = attn_bias.tbl if exists(attn_bias) else None
        mask_u8 = self_attn_mask.to(torch.bool).contiguous().view(torch.uint8) if exists(self_attn_mask) else None
        state = core.DecodeCache(cfg, b, n, x.device)
        flat = self.flat_params()
        if cache_len > 0 and n - cache_len == 1 and not dense:
            for l in range(cfg.depth):
                state.kv[l][:, :cache_len, :dh] = kv_cache[l, 0]
                state.kv[l][:, :cache_len, dh:] = kv_cache[l, 1]
            state.length = cache_len
            h = core.TransformerStackFn.apply(x[:, -1:].contiguous(), mask_u8, cfg, self._cache, dict(grad=False, decode=state, context_mask=context_mask),
                                              attn_bias, tbl, context, *flat)
            hidden = h.view(b, 1, d)
        else:
            hn = core.TransformerStackFn.apply(x, mask_u8, cfg, self._cache, dict(grad=False, kv_out=state, context_mask=context_mask), attn_bias, tbl,
                                               context, *flat)
            hidden = hn.view(b, n, d)[:, cache_len:]
        new_cache = torch.stack([torch.stack((kv[:, :n, :dh], kv[:, :n, dh:])) for kv in state.kv]).float()
        return hidden.float(), new_cache


    def _sample(self, tokens, self_attn_mask, state, context=None, context_mask=None):
        """One step of an autoregressive sampling run (kv cache, reference :360-394 / :560): tokens fp32 [b, n, d] = embeddings of the WHOLE
        sequence so far, state = core.DecodeCache (state.bias: AttnBias laid out for state.nmax positions, or None).  First call: ordinary
        forward over the n positions that also fills the cache; later calls: only the last position runs (n == state.length + 1), its
        attention reads the cache.  The single-position step is ~150 launches for ~0.3 ms of GPU work, i.e. bound by the host: after one
        eager step it is captured into a hipGraph (torch.cuda.CUDAGraph; the position index lives on the device) and every further step is
        ONE graph launch (ALM_DECODE_GRAPH=0 disables).  -> hidden state of the last position, bf16 [b, d]."""
        b, n, d = tokens.shape
        mask_u8 = None if self_attn_mask is None else self_attn_mask.to(torch.bool).contiguous().view(torch.uint8)
        bias = getattr(state, 'bias', None)
        flat = self.flat_params()
        if state.length == 0:
            pb = bias.sliced(n) if exists(bias) else None
            hn = core.TransformerStackFn.apply(tokens, mask_u8, self.cfg, self._cache, dict(grad=False, kv_out=state, context_mask=context_mask), pb,
                                               pb.tbl if exists(pb) else None, context, *flat)
            return hn.view(b, n, d)[:, -1]
        assert n == state.length + 1 and n <= state.nmax, (n, state.length, state.nmax)
        x = tokens[:, -1:].contiguous()
        tbl = bias.tbl if exists(bias) else None
        if not DECODE_GRAPH:
            return core.TransformerStackFn.apply(x, mask_u8, self.cfg, self._cache, dict(grad=False, decode=state, context_mask=context_mask), bias, tbl,
                                                 context, *flat)
        if state.pos_dev is None:
            # first single-position step: eager, already in the device-position form (also warms every lazily initialised launch path up)
            state.pos_dev = torch.full((1,), state.length, dtype=torch.int32, device=x.device)
            if exists(mask_u8):
                state.mask_in = torch.ones((b, state.nmax), dtype=torch.uint8, device=x.device)
                state.mask_in[:, :n] = mask_u8
            h = core.TransformerStackFn.apply(x, state.mask_in, self.cfg, self._cache, dict(grad=False, decode=state, context_mask=context_mask), bias, tbl,
                                              context, *flat)
            state.pos_dev += 1
            return h
        if exists(state.mask_in):
            state.mask_in[:, :n] = mask_u8
        if state.graph is None:
            state.x_in = x.clone()
            graph = torch.cuda.CUDAGraph()
            state.frozen = True
            try:
                with torch.cuda.graph(graph):
                    state.h_out = core.TransformerStackFn.apply(state.x_in, state.mask_in, self.cfg, self._cache,
                                                                dict(grad=False, decode=state, context_mask=context_mask), bias, tbl, context, *flat)
            finally:
                state.frozen = False
            state.graph = graph
        else:
            state.x_in.copy_(x)
        state.graph.replay()
        state.length += 1
        state.pos_dev += 1
        return state.h_out


# ---------------------------------------------------------------------------------------------- embedding assembly

_EMBED_SCATTER_OWNED = os.environ.get('ALM_EMBED_SCATTER', 'owned') != 'atomic'
_SEMANTIC_PREPARE = os.environ.get('ALM_SEMANTIC_PREPARE', '1') != '0'        # A/B + test switch: 0 = the ATen id bookkeeping in SemanticTransformerWrapper.forward


class EmbedAssembleFn(torch.autograd.Function):
    """tokens[r] = table_a[row_a] (+ table_b[row_b]); src codes are (table_id << 24 | row), -1 = zero vector."""

    @staticmethod
    def forward(ctx, src_a, src_b, rows, dim, *tables):
        flat = [t.detach().reshape(-1, dim) for t in tables]
        out = ops.embed_assemble(flat, src_a, src_b, rows, dim)
        ctx.src, ctx.tables, ctx.dim, ctx.rows = (src_a, src_b), tables, dim, rows
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous().to(torch.float32)
        sizes = [(t.numel() + 3) // 4 * 4 for t in ctx.tables]                                    # every table's slice starts on a 16-byte boundary
        # destination-owned scatter (default): one owner per table row, fixed summation order -> bitwise run-to-run deterministic embedding gradients, and it
        # writes every row itself (no zero fill).  ALM_EMBED_SCATTER=atomic: the round-2 kernel (fp32 L2 atomics, arrival order) for A/B runs.
        owned = _EMBED_SCATTER_OWNED and ctx.dim % 4 == 0
        flat = (torch.empty if owned else torch.zeros)(sum(sizes), dtype=torch.float32, device=dout.device)
        grads, o = [], 0
        for t, n in zip(ctx.tables, sizes):
            grads.append(flat[o:o + t.numel()].view(t.shape))
            o += n
        fn = ops.embed_scatter_owned if owned else ops.embed_scatter_add
        fn([g.view(-1, ctx.dim) for g in grads], ctx.src[0], ctx.src[1], dout.view(-1, ctx.dim), 1.0, ctx.rows, ctx.dim)
        return (None, None, None, None, *grads)


def _code(table_id, rows):
    return (rows.to(torch.int32) + (table_id << 24)).to(torch.int32)


# Index tensors that depend only on shapes (never on token values) are built once per (shape, device) and re-used: every training step used to
# launch ~30 aranges / fills / compares / wheres for them.  Callers treat them as read-only.  While a stream is being captured into a hipGraph
# nothing is cached (a tensor born inside a capture has no contents until the graph is replayed) and nothing cached is created.
def _shape_cached(fn):
    cached = functools.lru_cache(maxsize=256)(fn)

    @functools.wraps(fn)
    def wrapper(*args):
        device = args[-1]
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return fn(*args)
        with torch.inference_mode(False), torch.no_grad():
            return cached(*args)
    wrapper.cache_clear = cached.cache_clear
    return wrapper


@_shape_cached
def _const_code(table_id, b, device):
    return torch.full((b, 1), table_id << 24, dtype=torch.int32, device=device)


@_shape_cached
def _const_ids(value, b, device):
    return torch.full((b, 1), value, dtype=torch.int64, device=device)


@_shape_cached
def _neg(b, n, device):
    return torch.full((b, n), -1, dtype=torch.int32, device=device)


@_shape_cached
def _quantizer_rows(n, Q, device):
    return (torch.arange(n, device=device) % Q).to(torch.int32)


@_shape_cached
def _quantizer_row_offsets(n, Q, stride, device):
    """[1, n] int32: stride * (i mod Q) -- the per-quantizer offset of position i into a stacked codebook table"""
    return (stride * _quantizer_rows(n, Q, device))[None].contiguous()


@_shape_cached
def _quantizer_codes(table_id, b, lead, n, Q, device):
    """src_b of the assembly, [b, lead + n] int32: -1 for the `lead` leading positions, then the code of row (i mod Q) of table `table_id`"""
    return torch.cat((_neg(b, lead, device), _code(table_id, _quantizer_rows(n, Q, device))[None].expand(b, -1)), dim=1).contiguous()


@_shape_cached
def _fine_quantizer_codes(b, n, nf, Qc, Qf, device):
    """src_b of FineTransformer's assembly, [b, n + nf + 2] int32: -1 at the two start positions, the quantizer-embedding codes (tables 2 / 3) elsewhere"""
    qc, qf = _quantizer_rows(n, Qc, device), _quantizer_rows(nf, Qf, device)
    return torch.cat((_neg(b, 1, device), _code(2, qc)[None].expand(b, -1), _neg(b, 1, device), _code(3, qf)[None].expand(b, -1)), dim=1).contiguous()


@_shape_cached
def _group_index(B, N, start, n, Q, device):
    """Rows of the flat hidden states [B*N, D] regrouped per quantizer: position i of the range uses head i mod Q
    -> (idx int32 [Q, B*J] (-1 = pad), i_grid [Q, J], valid [Q, J], i_clamped [Q*J])."""
    J = ceil_div(n, Q)
    i_grid = torch.arange(J, device=device)[None, :] * Q + torch.arange(Q, device=device)[:, None]     # [Q, J]
    valid = i_grid < n
    rows = torch.arange(B, device=device)[None, :, None] * N + start + i_grid[:, None, :]              # [Q, B, J]
    idx = torch.where(valid[:, None, :], rows, torch.full_like(rows, -1)).reshape(Q, B * J).to(torch.int32)
    return idx.contiguous(), i_grid, valid


@_shape_cached
def _group_cols(n, Q, device):
    """[Q * J] int64: label column read by slot (q, j) = min(j * Q + q, n - 1) (the clamp only touches slots that do not exist)"""
    J = ceil_div(n, Q)
    i_grid = torch.arange(J, device=device)[None, :] * Q + torch.arange(Q, device=device)[:, None]
    return i_grid.clamp(max=n - 1).reshape(-1).contiguous()


def _group_labels(labels, i_grid, valid, n):
    """labels [B, n] int64 -> [Q, B*J] (-1 where the slot does not exist)."""
    B = labels.shape[0]
    Q, J = i_grid.shape
    g = labels[:, _group_cols(n, Q, labels.device)].reshape(B, Q, J).permute(1, 0, 2)                 # [Q, B, J]
    if J * Q != n:                                                                                    # ragged tail: mark the missing slots
        g = g.masked_fill(~valid[:, None, :], -1)
    return g.reshape(Q, B * J).contiguous()


def _ungroup_logits(logits_g, B, n, Q, C):
    """logits_g fp32 [Q*B*J, Cpad] -> reference layout [B, n, C]."""
    J = ceil_div(n, Q)
    lg = logits_g.view(Q, B, J, -1)[..., :C].permute(1, 2, 0, 3).reshape(B, J * Q, C)                # b (j q) c
    return lg[:, :n].contiguous()


class _TransformerBase(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    def load(self, path):                                     # audiolm_pytorch.py:627-638 / :805-816 / :1084-1095
        from packaging import version
        device = self.device
        path = Path(path)
        assert path.exists()
        pkg = torch.load(str(path), map_location=device)
        if 'version' in pkg and version.parse(pkg['version']) < version.parse(__version__):
            print(f'model was trained on older version {pkg["version"]} of audiolm-pytorch')
        self.load_state_dict(pkg['model'])
        return pkg

    def _last_logits(self, h, weight3, bias, key, q):
        """logits of head q for the last-position hidden states h bf16 [b, d] (sampling): every head of the group runs on the b rows (one
        batched GEMM against the cached bf16 weights of the training path), head q is returned -> fp32 [b, C]"""
        b, G = h.shape[0], weight3.shape[0]
        idx = torch.arange(b, device=h.device, dtype=torch.int32)[None].expand(G, b).contiguous()
        _, lg = heads.head_logits(h, weight3.detach(), None if bias is None else bias.detach(), idx, self._heads_cache(), ('head', key))
        return lg.view(G, b, -1)[q, :, :weight3.shape[1]]

    def _condition(self, b, device, text, text_embeds, cond_drop_prob, mask_from_embeds):
        """Conditioning of one forward (audiolm_pytorch.py:685-704 / :873-892 / :1150-1169): -> (context [b, m, dim] | None, context_mask bool
        [b, m] | None).  `text_embeds` are the pre-computed T5 / MuLaN embeddings (the text encoder itself is out of scope); positions whose
        embedding is all zero are padding.  Quirk kept: Semantic- and FineTransformer derive the mask only when they ran the text encoder
        themselves (:692-695, :1156-1160), so with pre-computed embeddings they have no mask and `cond_drop_prob` has nothing to drop
        (mask_from_embeds=False); CoarseTransformer takes the mask from the embeddings (:882-883)."""
        has_text = exists(text) or exists(text_embeds)
        assert not (self.has_condition ^ has_text)
        if not has_text:
            return None, None
        text_mask = None
        if not exists(text_embeds):
            with torch.inference_mode():
                text_embeds = self.embed_text(text, output_device=device)
            if not mask_from_embeds:
                text_mask = torch.any(text_embeds != 0, dim=-1)
        if mask_from_embeds:
            text_mask = torch.any(text_embeds != 0, dim=-1)
        if isinstance(self.proj_text_embed, nn.Linear):
            text_embeds = LinearNoBiasFn.apply(text_embeds.to(device).float(), self.proj_text_embed.weight, self._heads_cache(), ('proj_text_embed',))
        cond_drop_prob = default(cond_drop_prob, self.cond_drop_prob)
        if exists(text_mask) and cond_drop_prob > 0:
            keep_mask = prob_mask_like((b,), 1 - cond_drop_prob, device=device)
            text_mask = keep_mask[:, None] & text_mask
        return text_embeds, text_mask

    def forward_with_cond_scale(self, *args, cond_scale=3, kv_cache=None, embed_cache=None, return_kv_cache=False, **kwargs):
        """Classifier-free guidance (audiolm_pytorch.py:640-667 / :818-855 / :1097-1134): logits with the conditioning kept (cond_drop_prob = 0) and,
        for conditioned models at cond_scale != 1, with every text position masked out (cond_drop_prob = 1: cross-attention then sees only its
        null key / value); result = null + (cond - null) * cond_scale.
        kv_cache / embed_cache: the reference's stacked cache tensors -- one entry per pass ([cond] or [cond, null]), each a
        Transformer.forward_kv_protocol cache [depth, 2, b, n, dim_head] / hidden states [b, n, dim]; with return_kv_cache the new stacks are
        returned the same way (SemanticTransformer: logits, kv; Coarse / Fine: logits, (kv, embed)).  The wrappers' generate() do not go through
        this tensor protocol: they drive the native core.DecodeCache (sample_logits), which also replays the step as a hipGraph."""
        two = not isinstance(self, SemanticTransformer)
        if not (exists(kv_cache) or exists(embed_cache) or return_kv_cache):
            out = self.forward(*args, cond_drop_prob=0., **kwargs)
            if cond_scale != 1 and self.has_condition:
                null = self.forward(*args, cond_drop_prob=1., **kwargs)
                mix = lambda c, n: None if c is None else n + (c - n) * cond_scale                        # noqa: E731
                out = tuple(mix(c, n) for c, n in zip(out, null)) if isinstance(out, tuple) else mix(out, null)
            return out
        it_kv, it_em = iter(default(kv_cache, [])), iter(default(embed_cache, []))
        new_kv, new_em = [], []

        def one(cdp):
            if two:
                lg, (kv, em) = self.forward(*args, cond_drop_prob=cdp, return_cache=True, kv_cache=next(it_kv, None), embed_cache=next(it_em, None), **kwargs)
                new_em.append(em)
            else:
                lg, kv = self.forward(*args, cond_drop_prob=cdp, return_kv_cache=True, kv_cache=next(it_kv, None), **kwargs)
            new_kv.append(kv)
            return lg
        out = one(0.)
        if cond_scale != 1 and self.has_condition:
            null = one(1.)
            mix = lambda c, n: None if c is None else n + (c - n) * cond_scale                            # noqa: E731
            out = tuple(mix(c, n) for c, n in zip(out, null)) if two else mix(out, null)
        if not return_kv_cache:
            return out
        return out, ((torch.stack(new_kv), torch.stack(new_em)) if two else torch.stack(new_kv))

    def _protocol_hidden(self, tokens, self_attn_mask, attn_bias, context, context_mask, kv_cache, embed_cache):
        """hidden states of ALL positions through the reference's cache protocol: the transformer runs on the positions the kv cache does not
        cover, the rest comes from `embed_cache` (audiolm_pytorch.py:950-953 / :1312-1315) -> (flat fp32 [b*n, d], new kv cache, new embed cache)"""
        h, new_kv = self.transformer.forward_kv_protocol(tokens, self_attn_mask=self_attn_mask, context=context, context_mask=context_mask,
                                                         attn_bias=attn_bias, kv_cache=kv_cache)
        if exists(embed_cache):
            h = torch.cat((embed_cache.to(h.dtype), h), dim=-2)
        b, n, d = h.shape
        assert n == tokens.shape[1], f'kv_cache / embed_cache cover {n} of {tokens.shape[1]} positions: pass both caches of the same call'
        return h.reshape(b * n, d).contiguous(), new_kv, h

    def _state_condition(self, state, b, device, text_embeds, cond_drop_prob, mask_from_embeds):
        """conditioning context of a cached sampling run: projected ONCE and kept in the sampling state -- the text does not change during a run,
        and the hipGraph replay of the single-position step reads these very tensors (their addresses are baked into the captured graph)"""
        if state is not None and getattr(state, 'ctx', None) is not None:
            return state.ctx
        return self._condition(b, device, None, text_embeds, cond_drop_prob, mask_from_embeds)

    def _sample_guided(self, one, state, cond_scale):
        """cached sampling step of a conditioned model: `one(state, cond_drop_prob)` -> (logits, state).  cond_scale == 1: one pass with the
        conditioning kept; otherwise the guided pair (conditioned, unconditioned) on two caches, mixed like forward_with_cond_scale."""
        st = state if isinstance(state, list) else [state, None]
        lg, st[0] = one(st[0], 0.)
        if cond_scale != 1:
            null, st[1] = one(st[1], 1.)
            lg = null + (lg - null) * cond_scale
        return lg, st

    def _init_common(self, *, t5_name, has_condition, cond_dim, audio_text_condition, cond_drop_prob, dim):
        if audio_text_condition:
            has_condition = True
            cond_dim = default(cond_dim, dim)
        self.has_condition = has_condition
        self.cond_drop_prob = cond_drop_prob
        text_dim = default(cond_dim, get_encoded_dim(t5_name))
        self.proj_text_embed = nn.Linear(text_dim, dim, bias=False) if text_dim != dim else nn.Identity()

    def embed_text(self, *a, **k):
        raise NotImplementedError('the T5 text encoder is out of scope (SURVEY.md §2): pass pre-computed `text_embeds` (b, m, cond_dim) instead of `text`')

    def _heads_cache(self):
        return self.transformer._cache

    def _head_weights(self):
        """(cache key, fp32 weight [G, C, D]) of every logit head of this model"""
        out = []
        for key, attr in (('semantic', 'to_logits'), ('semantic', 'to_semantic_logits')):
            lin = getattr(self, attr, None)
            if lin is not None:
                out.append((key, lin.weight.unsqueeze(0)))
        for key, attr in (('coarse', 'coarse_logit_weights'), ('fine', 'fine_logit_weights')):
            w = getattr(self, attr, None)
            if w is not None:
                out.append((key, w))
        return out

    def prepack_weights(self):
        """packs every bf16 weight copy of the model (stack + logit heads) now instead of lazily inside the next forward"""
        self.transformer.prepack_weights()
        for key, w in self._head_weights():
            self._heads_cache().get(('head', key), w.detach(), heads._pack_head)


# ---------------------------------------------------------------------------------------------- SemanticTransformer (:564-724)

class SemanticTransformer(_TransformerBase):
    def __init__(self, *, dim, depth, num_semantic_tokens, heads=8, attn_dropout=0., ff_dropout=0., t5_name=DEFAULT_T5_NAME,
                 cond_dim=None, has_condition=False, audio_text_condition=False, cond_as_self_attn_prefix=False, cond_drop_prob=0.5,
                 grad_shrink_alpha=0.1, rel_pos_bias=True, flash_attn=False, **kwargs):
        super().__init__()
        rel_pos_bias = rel_pos_bias and not flash_attn
        self.num_semantic_tokens = num_semantic_tokens
        self._init_common(t5_name=t5_name, has_condition=has_condition, cond_dim=cond_dim, audio_text_condition=audio_text_condition,
                          cond_drop_prob=cond_drop_prob, dim=dim)
        self.start_token = nn.Parameter(torch.randn(dim))
        self.semantic_embedding = nn.Embedding(num_semantic_tokens + 1, dim)
        self.eos_id = num_semantic_tokens
        self.transformer = Transformer(dim=dim, depth=depth, heads=heads, attn_dropout=attn_dropout, ff_dropout=ff_dropout,
                                       cross_attend=self.has_condition and not cond_as_self_attn_prefix,
                                       cond_as_self_attn_prefix=cond_as_self_attn_prefix, grad_shrink_alpha=grad_shrink_alpha,
                                       rel_pos_bias=rel_pos_bias, flash_attn=flash_attn, **kwargs)
        self.to_logits = nn.Linear(dim, num_semantic_tokens + 1)
        self.dim = dim

    def _tokens(self, ids, self_attn_mask, src_a=None):
        b, n = ids.shape
        dev = ids.device
        if src_a is None:                                                                      # (else: from ops.semantic_prepare -- the same codes, one kernel)
            sem = ids.to(torch.int32)                                                          # table 0; pad (-1) -> zero vector (:176-181)
            src_a = torch.cat((_const_code(1, b, dev), sem), dim=1).contiguous()
        src_b = _neg(b, n + 1, dev)
        tokens = EmbedAssembleFn.apply(src_a.reshape(-1), src_b.reshape(-1), b * (n + 1), self.dim,
                                       self.semantic_embedding.weight, self.start_token).view(b, n + 1, self.dim)
        if exists(self_attn_mask):
            self_attn_mask = F.pad(self_attn_mask, (1, 0), value=True)                       # :716
        return tokens, self_attn_mask

    def _hidden(self, ids, self_attn_mask, context=None, context_mask=None, src_a=None):
        tokens, self_attn_mask = self._tokens(ids, self_attn_mask, src_a)
        b, n1 = tokens.shape[:2]
        return self.transformer(tokens, self_attn_mask=self_attn_mask, context=context, context_mask=context_mask, return_flat_hidden=True), b, n1

    @torch.no_grad()
    def sample_logits(self, ids, state, nmax, text_embeds=None, cond_scale=1.):
        """Sampling step with a kv cache: ids (b, n) = everything sampled so far -> (next-token logits fp32 (b, C + 1), state).  state None:
        prefix forward that creates the cache for up to `nmax` positions (incl. the start token); afterwards only the last id is new.
        Conditioned models (cross-attention): `text_embeds`; cond_scale != 1 runs the guided pair of passes on two caches (:640-667)."""
        if self.has_condition:
            return self._sample_guided(lambda st, cdp: self._sample_one(ids, st, nmax, text_embeds, cdp), state, cond_scale)
        return self._sample_one(ids, state, nmax, None, None)

    def _sample_one(self, ids, state, nmax, text_embeds, cond_drop_prob):
        b, n = ids.shape
        context, cmask = self._state_condition(state, b, ids.device, text_embeds, cond_drop_prob, mask_from_embeds=False)
        dev = ids.device
        src_a = torch.cat((_const_code(1, b, dev), ids.to(torch.int32)), dim=1).contiguous()
        tokens = EmbedAssembleFn.apply(src_a.reshape(-1), _neg(b, n + 1, dev).reshape(-1), b * (n + 1), self.dim,
                                       self.semantic_embedding.weight, self.start_token).view(b, n + 1, self.dim)
        if state is None:
            state = core.DecodeCache(self.transformer.cfg, b, nmax, dev)
            state.ctx = (context, cmask)
            state.bias = self.transformer.rel_pos_bias(nmax, nmax) if exists(self.transformer.rel_pos_bias) else None
        h = self.transformer._sample(tokens, None, state, context, cmask)
        return self._last_logits(h, self.to_logits.weight.unsqueeze(0), self.to_logits.bias, 'semantic', 0), state

    def forward(self, *, ids=None, return_loss=False, text=None, text_embeds=None, self_attn_mask=None, cond_drop_prob=None,
                unique_consecutive=None, kv_cache=None, return_kv_cache=False, labels=None, _src_a=None):
        context, context_mask = self._condition(ids.shape[0], ids.device, text, text_embeds, cond_drop_prob, mask_from_embeds=False)
        if return_loss:
            ids = ids[:, :-1]                                                                # :706-707 (the reference drops the labels)
        new_kv = None
        if exists(kv_cache) or return_kv_cache:                                              # the reference's cache protocol (:719): inference only
            assert not exists(labels)
            tokens, mask = self._tokens(ids, self_attn_mask)
            h, new_kv = self.transformer.forward_kv_protocol(tokens, self_attn_mask=mask, context=context, context_mask=context_mask, kv_cache=kv_cache)
            b, N = h.shape[:2]                                                               # only the positions the cache did not cover
            hn = h.reshape(b * N, self.dim).contiguous()
        else:
            hn, b, N = self._hidden(ids, self_attn_mask, context, context_mask, src_a=None if return_loss else _src_a)
        idx, i_grid, valid = _group_index(b, N, 0, N, 1, ids.device)
        if exists(labels):                                                                   # fused loss path (wrapper)
            grp = heads.HeadGroup('semantic', self.to_logits.weight, self.to_logits.bias, idx, _group_labels(labels, i_grid, valid, N))
            (loss_sum,) = heads.HeadsLossFn.apply(hn, [grp], self._heads_cache(), self.to_logits.weight, self.to_logits.bias)
            return heads.combine_losses([loss_sum], [labels], [1.0])                               # CE mean over the non-pad labels
        C = self.num_semantic_tokens + 1
        _, lg = heads.head_logits(hn, self.to_logits.weight.detach().unsqueeze(0), self.to_logits.bias.detach(), idx,
                                  self._heads_cache(), ('head', 'semantic'))
        logits = _ungroup_logits(lg, b, N, 1, C)
        if not return_kv_cache:
            return logits
        return logits, new_kv


# ---------------------------------------------------------------------------------------------- CoarseTransformer (:726-990)

class CoarseTransformer(_TransformerBase):
    def __init__(self, *, codebook_size, num_coarse_quantizers, dim, depth, num_semantic_tokens, heads=8, attn_dropout=0.,
                 ff_dropout=0., t5_name=DEFAULT_T5_NAME, has_condition=False, cond_dim=None, audio_text_condition=False,
                 cond_as_self_attn_prefix=False, cond_drop_prob=0.5, grad_shrink_alpha=0.1, project_semantic_logits=True,
                 rel_pos_bias=True, flash_attn=False, **kwargs):
        super().__init__()
        rel_pos_bias = rel_pos_bias and not flash_attn
        self.num_semantic_tokens = num_semantic_tokens
        self._init_common(t5_name=t5_name, has_condition=has_condition, cond_dim=cond_dim, audio_text_condition=audio_text_condition,
                          cond_drop_prob=cond_drop_prob, dim=dim)
        self.semantic_start_token = nn.Parameter(torch.randn(dim))
        self.coarse_start_token = nn.Parameter(torch.randn(dim))
        self.semantic_eos_id = num_semantic_tokens
        self.semantic_embedding = nn.Embedding(num_semantic_tokens + 1, dim)
        self.coarse_eos_id = codebook_size
        codebook_size_with_eos = codebook_size + 1
        self.coarse_embedding = nn.Embedding(num_coarse_quantizers * codebook_size_with_eos, dim)
        self.coarse_quantize_embedding = nn.Embedding(num_coarse_quantizers, dim)
        self.cross_attn_bias = nn.Parameter(torch.zeros(heads, 1, 1)) if rel_pos_bias else None
        self.transformer = Transformer(dim=dim, depth=depth, heads=heads, attn_dropout=attn_dropout, ff_dropout=ff_dropout,
                                       cross_attend=self.has_condition and not cond_as_self_attn_prefix,
                                       cond_as_self_attn_prefix=cond_as_self_attn_prefix, grad_shrink_alpha=grad_shrink_alpha,
                                       rel_pos_bias=rel_pos_bias, flash_attn=flash_attn, **kwargs)
        self.codebook_size = codebook_size
        self.num_coarse_quantizers = num_coarse_quantizers
        self.to_semantic_logits = nn.Linear(dim, num_semantic_tokens + 1) if project_semantic_logits else None
        self.coarse_logit_weights = nn.Parameter(torch.randn(num_coarse_quantizers, codebook_size_with_eos, dim))
        self.dim = dim

    def _assemble(self, semantic_token_ids, coarse_token_ids, prepared=None):
        b, dev = semantic_token_ids.shape[0], semantic_token_ids.device
        Q, C = self.num_coarse_quantizers, self.codebook_size
        if prepared is not None:                                                                 # (src_a, ns, nc) from ops.coarse_prepare: the codes below, one kernel
            src_a, ns, nc = prepared
        else:
            coarse, sem = _flatten_ids(coarse_token_ids), _flatten_ids(semantic_token_ids)       # :894
            ns, nc = sem.shape[1], coarse.shape[1]
            coarse_rows = coarse.to(torch.int32) + _quantizer_row_offsets(nc, Q, C, dev)         # :896-899 (stride C: eos aliasing kept)
            sem_code = sem.to(torch.int32).clamp(min=-1)                                         # table 0; pad (any negative id) -> zero (:901)
            src_a = torch.cat((_const_code(3, b, dev), sem_code, _const_code(4, b, dev), _code(1, coarse_rows)), dim=1).contiguous()
        src_b = _quantizer_codes(2, b, ns + 2, nc, Q, dev)                                       # :904-906
        N = ns + nc + 2
        tokens = EmbedAssembleFn.apply(src_a.reshape(-1), src_b.reshape(-1), b * N, self.dim, self.semantic_embedding.weight,
                                       self.coarse_embedding.weight, self.coarse_quantize_embedding.weight,
                                       self.semantic_start_token, self.coarse_start_token).view(b, N, self.dim)   # :913-918
        return tokens, b, N, ns, nc

    @torch.no_grad()
    def sample_logits(self, semantic_token_ids, coarse_token_ids, state, nmax, text_embeds=None, cond_scale=1.):
        """Sampling step with a kv cache -> (logits of the NEXT coarse token fp32 (b, C + 1), state); see SemanticTransformer.sample_logits.
        nmax = semantic length + 2 + the largest number of coarse tokens the run will hold."""
        if self.has_condition:
            return self._sample_guided(lambda st, cdp: self._sample_one(semantic_token_ids, coarse_token_ids, st, nmax, text_embeds, cdp), state, cond_scale)
        return self._sample_one(semantic_token_ids, coarse_token_ids, state, nmax, None, None)

    def _sample_one(self, semantic_token_ids, coarse_token_ids, state, nmax, text_embeds, cond_drop_prob):
        tokens, b, N, ns, nc = self._assemble(semantic_token_ids, coarse_token_ids)
        context, cmask = self._state_condition(state, b, tokens.device, text_embeds, cond_drop_prob, mask_from_embeds=True)
        if state is None:
            state = core.DecodeCache(self.transformer.cfg, b, nmax, tokens.device)
            state.ctx = (context, cmask)
            state.bias = None
            if exists(self.transformer.rel_pos_bias):
                state.bias = self.transformer.rel_pos_bias(nmax, nmax, special=self.cross_attn_bias, num_leading=ns + 1)
        h = self.transformer._sample(tokens, None, state, context, cmask)
        return self._last_logits(h, self.coarse_logit_weights, None, 'coarse', nc % self.num_coarse_quantizers), state

    def _hidden(self, semantic_token_ids, coarse_token_ids, self_attn_mask, context=None, context_mask=None, prepared=None):
        tokens, b, N, ns, nc = self._assemble(semantic_token_ids, coarse_token_ids, prepared)
        attn_bias = None
        if exists(self.transformer.rel_pos_bias):
            # :924-936 -- relative positions everywhere except across the semantic / coarse boundary, where every pair gets the learned
            # per-head cross_attn_bias (is_semantic = arange(N) < ns + 1)
            attn_bias = self.transformer.rel_pos_bias(N, N, special=self.cross_attn_bias, num_leading=ns + 1)
        hn = self.transformer(tokens, self_attn_mask=self_attn_mask, attn_bias=attn_bias, context=context, context_mask=context_mask,
                              return_flat_hidden=True)
        return hn, b, N, ns, nc

    def _groups(self, b, N, ns, nc, dev, semantic_labels=None, coarse_labels=None, only_coarse=False):
        Q = self.num_coarse_quantizers
        groups, params = [], []
        n_coarse = nc + 1                                                                         # tokens[:, ns+1:]  (:957)
        if exists(self.to_semantic_logits) and not only_coarse:
            idx, ig, va = _group_index(b, N, 0, ns, 1, dev)                                       # tokens[:, :ns]
            lab = _group_labels(semantic_labels, ig, va, ns) if exists(semantic_labels) else None
            groups.append(heads.HeadGroup('semantic', self.to_semantic_logits.weight, self.to_semantic_logits.bias, idx, lab))
            params += [self.to_semantic_logits.weight, self.to_semantic_logits.bias]
        idx, ig, va = _group_index(b, N, ns + 1, n_coarse, Q, dev)
        lab = _group_labels(coarse_labels, ig, va, n_coarse) if exists(coarse_labels) else None
        groups.append(heads.HeadGroup('coarse', self.coarse_logit_weights, None, idx, lab))
        params.append(self.coarse_logit_weights)
        return groups, params

    def forward(self, *, semantic_token_ids, coarse_token_ids, self_attn_mask=None, text=None, text_embeds=None, cond_drop_prob=None,
                return_only_coarse_logits=False, return_cache=False, kv_cache=None, embed_cache=None, labels=None, loss_weights=None, _prepared=None):
        context, context_mask = self._condition(semantic_token_ids.shape[0], semantic_token_ids.device, text, text_embeds, cond_drop_prob, mask_from_embeds=True)
        caches = (None, None)
        if exists(kv_cache) or exists(embed_cache) or return_cache:                               # the reference's cache protocol (:938-953): inference only
            assert not exists(labels)
            tokens, b, N, ns, nc = self._assemble(semantic_token_ids, coarse_token_ids)
            attn_bias = self.transformer.rel_pos_bias(N, N, special=self.cross_attn_bias, num_leading=ns + 1) if exists(self.transformer.rel_pos_bias) else None
            hn, new_kv, new_em = self._protocol_hidden(tokens, self_attn_mask, attn_bias, context, context_mask, kv_cache, embed_cache)
            caches = (new_kv, new_em)
        else:
            hn, b, N, ns, nc = self._hidden(semantic_token_ids, coarse_token_ids, self_attn_mask, context, context_mask, _prepared)
        dev = hn.device
        if exists(labels):                                                                        # fused loss path: (sem_labels, coarse_labels)
            sem_labels, coarse_labels = labels
            groups, params = self._groups(b, N, ns, nc, dev, sem_labels, coarse_labels)
            sums = heads.HeadsLossFn.apply(hn, groups, self._heads_cache(), *params)
            labs = ([sem_labels] if len(sums) == 2 else []) + [coarse_labels]
            if exists(loss_weights):                                                              # the wrapper's weighted combination, fused with the means
                return heads.combine_losses(sums, labs, loss_weights[-len(sums):])
            out = [heads.combine_losses([s], [lab], [1.0]) for s, lab in zip(sums, labs)]
            return (out[0] if len(out) == 2 else None), out[-1]
        groups, _ = self._groups(b, N, ns, nc, dev, only_coarse=return_only_coarse_logits)
        outs = []
        for gi, g in enumerate(groups):
            w3 = g.weight.detach() if g.weight.dim() == 3 else g.weight.detach().unsqueeze(0)
            _, lg = heads.head_logits(hn, w3, None if g.bias is None else g.bias.detach(), g.idx, self._heads_cache(), ('head', g.name))
            outs.append(lg)
        coarse_logits = _ungroup_logits(outs[-1], b, nc + 1, self.num_coarse_quantizers, self.codebook_size + 1)
        semantic_logits = _ungroup_logits(outs[0], b, ns, 1, self.num_semantic_tokens + 1) if len(outs) == 2 else None
        logits = (semantic_logits, coarse_logits)
        if not return_cache:
            return logits
        return logits, caches


# ---------------------------------------------------------------------------------------------- FineTransformer (:992-1368)

class FineTransformer(_TransformerBase):
    def __init__(self, *, num_coarse_quantizers, num_fine_quantizers, codebook_size, dim, depth, heads=8, attn_dropout=0.,
                 ff_dropout=0., t5_name=DEFAULT_T5_NAME, has_condition=False, cond_dim=None, audio_text_condition=False,
                 cond_as_self_attn_prefix=False, cond_drop_prob=0.5, grad_shrink_alpha=0.1, project_coarse_logits=True, pad_id=-1,
                 rel_pos_bias=True, flash_attn=False, **kwargs):
        super().__init__()
        rel_pos_bias = rel_pos_bias and not flash_attn
        self._init_common(t5_name=t5_name, has_condition=has_condition, cond_dim=cond_dim, audio_text_condition=audio_text_condition,
                          cond_drop_prob=cond_drop_prob, dim=dim)
        self.num_coarse_quantizers = num_coarse_quantizers
        self.coarse_start_token = nn.Parameter(torch.randn(dim))
        self.fine_start_token = nn.Parameter(torch.randn(dim))
        self.coarse_embedding = nn.Embedding(num_coarse_quantizers * codebook_size, dim)
        self.fine_embedding = nn.Embedding(num_fine_quantizers * codebook_size, dim)
        self.coarse_quantize_embedding = nn.Embedding(num_coarse_quantizers, dim)
        self.fine_quantize_embedding = nn.Embedding(num_fine_quantizers, dim)
        self.pad_id = pad_id
        self.eos_id = codebook_size
        self.transformer = Transformer(dim=dim, depth=depth, heads=heads, attn_dropout=attn_dropout, ff_dropout=ff_dropout,
                                       cross_attend=self.has_condition and not cond_as_self_attn_prefix,
                                       cond_as_self_attn_prefix=cond_as_self_attn_prefix, rel_pos_bias=False,
                                       grad_shrink_alpha=grad_shrink_alpha, flash_attn=flash_attn, **kwargs)
        self.null_pos_bias = nn.Parameter(torch.randn(heads, 1, 1)) if rel_pos_bias else None
        pos_bias_mlp_dim = dim // 2
        self.pos_bias_mlp = nn.Sequential(
            nn.Linear(2, pos_bias_mlp_dim), nn.SiLU(), nn.Linear(pos_bias_mlp_dim, pos_bias_mlp_dim), nn.SiLU(),
            nn.Linear(pos_bias_mlp_dim, heads)) if rel_pos_bias else None
        self.codebook_size = codebook_size
        self.num_fine_quantizers = num_fine_quantizers
        self.coarse_logit_weights = nn.Parameter(torch.randn(num_coarse_quantizers, codebook_size, dim)) if project_coarse_logits else None
        self.fine_logit_weights = nn.Parameter(torch.randn(num_fine_quantizers, codebook_size, dim))
        self.dim = dim

    def _assemble(self, coarse_token_ids, fine_token_ids, self_attn_mask):
        b, dev = coarse_token_ids.shape[0], coarse_token_ids.device
        Qc, Qf, C = self.num_coarse_quantizers, self.num_fine_quantizers, self.codebook_size
        coarse, fine = _flatten_ids(coarse_token_ids), _flatten_ids(fine_token_ids)               # :1171
        n, nf = coarse.shape[1], fine.shape[1]
        if (FUSED_PREPARE and coarse.is_cuda and coarse.dtype == torch.int64 and fine.dtype == torch.int64 and coarse.stride(1) == 1 and fine.stride(1) == 1
                and max(Qc, Qf) * C < (1 << 24)):
            # the id bookkeeping below as ONE kernel (ops.fine_prepare; round 4: the Coarse wrapper got its own in round 3): ~12 ATen launches fewer
            src_a, coarse_mask = ops.fine_prepare(coarse, fine, nf, self.pad_id, self.eos_id, Qc, Qf, C)
        else:
            coarse_mask = (coarse != self.pad_id) & (coarse != self.eos_id)                       # :1175
            coarse = coarse.masked_fill(~coarse_mask, 0)
            coarse_mask = F.pad(coarse_mask, (1, nf + 1), value=True)                             # :1179
            qc, qf = _quantizer_rows(n, Qc, dev), _quantizer_rows(nf, Qf, dev)
            coarse_rows = coarse.to(torch.int32) + qc[None] * C                                   # :1195
            fine_rows = fine.to(torch.int32) + qf[None] * C                                       # :1202
            src_a = torch.cat((_const_code(4, b, dev), _code(0, coarse_rows), _const_code(5, b, dev), _code(1, fine_rows)), dim=1).contiguous()
        if exists(self_attn_mask):
            self_attn_mask &= coarse_mask                                                         # in place, like the reference (:1182)
        else:
            self_attn_mask = coarse_mask
        src_b = _fine_quantizer_codes(b, n, nf, Qc, Qf, dev)
        N = n + nf + 2
        tokens = EmbedAssembleFn.apply(src_a.reshape(-1), src_b.reshape(-1), b * N, self.dim, self.coarse_embedding.weight,
                                       self.fine_embedding.weight, self.coarse_quantize_embedding.weight,
                                       self.fine_quantize_embedding.weight, self.coarse_start_token, self.fine_start_token).view(b, N, self.dim)
        return tokens, self_attn_mask, b, n, nf, N

    def _attn_bias(self, n, nf, dev):
        if not exists(self.pos_bias_mlp):
            return None
        # :1229-1298 -- the MLP over the (relative frame, relative quantizer) grid stays a table; start-token pairs read null_pos_bias
        grid, index = relpos.fine_index(n, nf, self.num_coarse_quantizers, self.num_fine_quantizers, dev)
        mlp = self.pos_bias_mlp
        tbl = relpos.PosTableFn.apply(grid, self.null_pos_bias, 64 ** 0.5, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias,
                                      mlp[4].weight, mlp[4].bias)
        return relpos.AttnBias(tbl, *index)

    @torch.no_grad()
    def sample_logits(self, coarse_token_ids, fine_token_ids, state, max_fine_length, text_embeds=None, cond_scale=1.):
        """Sampling step with a kv cache -> (logits of the NEXT fine token fp32 (b, C), state); see SemanticTransformer.sample_logits.  The
        bias table / index vectors are laid out once for `max_fine_length` fine tokens (the reference's table does not change while the
        number of fine frames stays <= the number of coarse frames, :1230)."""
        if self.has_condition:
            return self._sample_guided(lambda st, cdp: self._sample_one(coarse_token_ids, fine_token_ids, st, max_fine_length, text_embeds, cdp), state, cond_scale)
        return self._sample_one(coarse_token_ids, fine_token_ids, state, max_fine_length, None, None)

    def _sample_one(self, coarse_token_ids, fine_token_ids, state, max_fine_length, text_embeds, cond_drop_prob):
        tokens, mask, b, n, nf, N = self._assemble(coarse_token_ids, fine_token_ids, None)
        context, cmask = self._state_condition(state, b, tokens.device, text_embeds, cond_drop_prob, mask_from_embeds=False)
        if state is None:
            state = core.DecodeCache(self.transformer.cfg, b, n + max_fine_length + 2, tokens.device)
            state.ctx = (context, cmask)
            state.bias = self._attn_bias(n, max_fine_length, tokens.device)
        h = self.transformer._sample(tokens, mask, state, context, cmask)
        return self._last_logits(h, self.fine_logit_weights, None, 'fine', nf % self.num_fine_quantizers), state

    def forward(self, coarse_token_ids, fine_token_ids, text=None, text_embeds=None, cond_drop_prob=None, self_attn_mask=None,
                kv_cache=None, embed_cache=None, return_cache=False, return_only_fine_logits=False, labels=None, loss_weights=None):
        context, context_mask = self._condition(coarse_token_ids.shape[0], coarse_token_ids.device, text, text_embeds, cond_drop_prob, mask_from_embeds=False)
        tokens, self_attn_mask, b, n, nf, N = self._assemble(coarse_token_ids, fine_token_ids, self_attn_mask)
        dev = tokens.device
        Qc, Qf, C = self.num_coarse_quantizers, self.num_fine_quantizers, self.codebook_size
        attn_bias = self._attn_bias(n, nf, dev)
        caches = (None, None)
        if exists(kv_cache) or exists(embed_cache) or return_cache:                               # the reference's cache protocol (:1300-1315): inference only
            assert not exists(labels)
            hn, new_kv, new_em = self._protocol_hidden(tokens, self_attn_mask, attn_bias, context, context_mask, kv_cache, embed_cache)
            caches = (new_kv, new_em)
        else:
            hn = self.transformer(tokens, self_attn_mask=self_attn_mask, attn_bias=attn_bias, context=context, context_mask=context_mask,
                                  return_flat_hidden=True)

        n_fine = nf + 1                                                                           # tokens[:, n+1:]  (:1319)
        want_coarse = exists(self.coarse_logit_weights) and not return_only_fine_logits
        groups, params = [], []
        if want_coarse:
            idx, ig, va = _group_index(b, N, 0, n, Qc, dev)                                       # tokens[:, :n] (zero-pad + slice == ragged tail)
            lab = _group_labels(labels[0], ig, va, n) if exists(labels) else None
            groups.append(heads.HeadGroup('coarse', self.coarse_logit_weights, None, idx, lab)); params.append(self.coarse_logit_weights)
        idx, ig, va = _group_index(b, N, n + 1, n_fine, Qf, dev)
        lab = _group_labels(labels[1], ig, va, n_fine) if exists(labels) else None
        groups.append(heads.HeadGroup('fine', self.fine_logit_weights, None, idx, lab)); params.append(self.fine_logit_weights)

        if exists(labels):
            sums = heads.HeadsLossFn.apply(hn, groups, self._heads_cache(), *params)
            labs = ([labels[0]] if want_coarse else []) + [labels[1]]
            if exists(loss_weights):                                                              # the wrapper's weighted combination, fused with the means
                return heads.combine_losses(sums, labs, loss_weights[-len(sums):])
            out = [heads.combine_losses([s], [l], [1.0]) for s, l in zip(sums, labs)]
            return (out[0] if want_coarse else None), out[-1]
        outs = []
        for gi, g in enumerate(groups):
            _, lg = heads.head_logits(hn, g.weight.detach(), None, g.idx, self._heads_cache(), ('head', g.name))
            outs.append(lg)
        fine_logits = _ungroup_logits(outs[-1], b, n_fine, Qf, C)
        coarse_logits = _ungroup_logits(outs[0], b, n, Qc, C) if want_coarse else None
        logits = (coarse_logits, fine_logits)
        if not return_cache:
            return logits
        return logits, caches


# ---------------------------------------------------------------------------------------------- training wrappers

class _WrapperBase(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    def _reconstruct(self, ids, per_position_any_pad=True):
        """codec.decode_from_codebook_indices on (b, n, q) ids; rows with padded (-1) positions are decoded one by one without them, like
        the reference (audiolm_pytorch.py:1712-1740 / :2011-2039) -> (b, samples) tensor, or a list of 1-D waves (None: nothing to decode)."""
        assert exists(self.codec)
        pad = (ids == -1).any(dim=-1)                                        # (b, n)
        if not bool(pad.any()):
            return self.codec.decode_from_codebook_indices(ids).squeeze(1)   # 'b 1 n -> b n'
        wavs = []
        for sample, has_padding in zip(ids, pad):
            if bool(has_padding.all()):
                wavs.append(None)
                continue
            wavs.append(self.codec.decode_from_codebook_indices(sample[~has_padding].unsqueeze(0)).reshape(-1))
        return wavs

    def embed_text(self, text):
        return self.transformer.embed_text(text, output_device=self.device)

    def _resolve_text(self, text, text_embeds, wave, namespace):
        """conditioning inputs of generate() / forward() (audiolm_pytorch.py:1446-1458, :1525-1528 and the Coarse / Fine twins): joint audio-text
        embeddings from the audio conditioner (any callable: MuLaN wrapper) when a wave is there, else `text` through the text encoder,
        else the pre-computed `text_embeds`"""
        if exists(self.audio_conditioner) and exists(wave):
            assert not exists(text) and not exists(text_embeds)
            text_embeds = self.audio_conditioner(wavs=wave, namespace=namespace)
        has_text = exists(text) or exists(text_embeds)
        assert not (self.transformer.has_condition ^ has_text)
        if not exists(text_embeds) and exists(text):
            with torch.inference_mode():
                text_embeds = self.transformer.embed_text(text, output_device=self.device)
        return text_embeds


class SemanticTransformerWrapper(_WrapperBase):               # audiolm_pytorch.py:1372-1567
    def __init__(self, *, transformer: SemanticTransformer, wav2vec=None, audio_conditioner=None, pad_id=-1, unique_consecutive=True,
                 mask_prob=0.15):
        super().__init__()
        self.wav2vec = wav2vec
        self.transformer = transformer
        self.to(transformer.device)
        self.audio_conditioner = audio_conditioner
        assert not (exists(audio_conditioner) and not transformer.has_condition)
        assert not exists(self.wav2vec) or self.wav2vec.codebook_size == transformer.num_semantic_tokens
        self.unique_consecutive = unique_consecutive
        self.pad_id = pad_id
        self.eos_id = transformer.eos_id
        self.mask_prob = mask_prob

    @eval_decorator
    @torch.inference_mode()
    def generate(self, *, max_length, text=None, text_embeds=None, prime_wave=None, prime_wave_input_sample_hz=None, prime_ids=None,
                 batch_size=1, cond_scale=3, filter_thres=0.9, temperature=1., use_kv_cache=True, include_eos_in_output=True, **kwargs):
        """audiolm_pytorch.py:1406-1511.  Same sampling semantics (top-k filter, gumbel-max, eos stop, mask after eos).  use_kv_cache=True:
        the prefix runs once through the training forward path, every further token costs one single-position pass whose attention reads
        the per-layer key / value cache (alm_mqa_decode_attn); False: the whole prefix is recomputed each step (same logits, O(n) more work)."""
        device = self.device
        if exists(prime_wave):
            assert not exists(prime_ids)
            assert exists(self.wav2vec)
            ids = self.wav2vec(prime_wave, flatten=False, input_sample_hz=prime_wave_input_sample_hz)
        elif exists(prime_ids):
            ids = prime_ids
        else:
            ids = torch.empty((batch_size, 0), dtype=torch.long, device=device)
        if self.unique_consecutive:
            ids = batch_unique_consecutive(ids, pad_value=self.pad_id)
        text_embeds = self._resolve_text(text, text_embeds, prime_wave, 'semantic')
        batch = ids.shape[0]
        start_length = ids.shape[-1]
        sample_semantic_ids = ids.clone()
        last_logit_indices = (ids != self.pad_id).sum(dim=-1).long()
        # kv cache (native: core.DecodeCache through SemanticTransformer.sample_logits): usable when no prime row is padded, i.e. every row's
        # next-token logits sit at the last position; ragged primes take the recompute path, which reproduces the reference's gather
        # (prefix conditioning: the reference turns the kv cache off, audiolm_pytorch.py:481-482)
        use_cache = use_kv_cache and not kwargs and bool((ids != self.pad_id).all()) and not self.transformer.transformer.cond_as_self_attn_prefix
        state = None
        for ind in range(start_length, max_length):
            if use_cache:
                last_logits, state = self.transformer.sample_logits(sample_semantic_ids, state, max_length + 1, text_embeds=text_embeds, cond_scale=cond_scale)
            else:
                logits = self.transformer.forward_with_cond_scale(ids=sample_semantic_ids, text_embeds=text_embeds, cond_scale=cond_scale, **kwargs)
                last_logits = logits.gather(1, last_logit_indices.view(batch, 1, 1).expand(batch, 1, logits.shape[-1])).squeeze(1)
            filtered_logits = top_k(last_logits, thres=filter_thres)
            sampled = gumbel_sample(filtered_logits, temperature=temperature, dim=-1)
            sample_semantic_ids = torch.cat((sample_semantic_ids, sampled.unsqueeze(-1)), dim=-1)
            if all_rows_have_eos_id(sample_semantic_ids, self.eos_id):
                break
            last_logit_indices += 1
        return mask_out_after_eos_id(sample_semantic_ids, self.eos_id, keep_eos=False)

    def forward(self, *, semantic_token_ids=None, raw_wave=None, text=None, text_embeds=None, return_loss=False, **kwargs):
        assert exists(raw_wave) or exists(semantic_token_ids)
        if exists(self.audio_conditioner):                                                          # :1525-1528
            assert exists(raw_wave)
            assert not exists(text) and not exists(text_embeds)
            text_embeds = self.audio_conditioner(wavs=raw_wave, namespace='semantic')
        if not exists(semantic_token_ids):
            assert exists(self.wav2vec), 'VQWav2Vec must be be provided if given raw wave for training'
            semantic_token_ids = self.wav2vec(raw_wave, flatten=False)
        semantic_token_ids = _flatten_ids(semantic_token_ids)
        if (self.training and return_loss and semantic_token_ids.is_cuda and semantic_token_ids.dtype == torch.int64 and semantic_token_ids.numel() > 0
                and not kwargs and _SEMANTIC_PREPARE):
            # the training step's id bookkeeping as ONE kernel (ops.semantic_prepare: eos appended = the labels, [start | ids] embedding codes), as the
            # Coarse / Fine wrappers have it; every other call pattern takes the ATen formulation below (equality-tested against it)
            sem = semantic_token_ids if semantic_token_ids.stride(1) == 1 else semantic_token_ids.contiguous()
            rows = self.transformer.semantic_embedding.weight.shape[0]
            if self.unique_consecutive:
                # :1536-1539 -- eos appended, runs collapsed, rows right-padded to the longest: one launch + one host read (the width is data-dependent)
                full, lengths = ops.unique_consecutive(sem, self.transformer.eos_id, self.pad_id)
                self.transformer.transformer.prepack_weights()                                        # (width-independent work before the host read)
                full = full[:, :int(lengths.max())]                                                   # [ids | eos | pad ...]
                labels, src_a = ops.semantic_prepare(full, self.transformer.eos_id, rows, has_eos=True)
                sem = full[:, :-1]                                                                     # the input ids (shape only from here on)
            else:
                labels, src_a = ops.semantic_prepare(sem, self.transformer.eos_id, rows)
            self_attn_mask = generate_mask_with_prob(sem.shape, self.mask_prob, sem.device) if self.mask_prob > 0. else None
            return self.transformer(ids=sem, text=text, text_embeds=text_embeds, self_attn_mask=self_attn_mask, labels=labels, _src_a=src_a)
        if self.training:
            semantic_token_ids = append_eos_id(semantic_token_ids, self.transformer.eos_id)        # :1536-1537
        if self.unique_consecutive:
            semantic_token_ids = batch_unique_consecutive(semantic_token_ids, pad_value=self.pad_id)
        input_ids = semantic_token_ids
        if return_loss:
            input_ids = semantic_token_ids[:, :-1]
        self_attn_mask = None
        if self.mask_prob > 0. and self.training:
            self_attn_mask = generate_mask_with_prob(input_ids.shape, self.mask_prob, input_ids.device)
        if not return_loss:
            return self.transformer(ids=input_ids, text=text, text_embeds=text_embeds, self_attn_mask=self_attn_mask, **kwargs)
        # reference: logits = transformer(ids = input_ids); CE(logits 'b c n', semantic_token_ids, ignore_index = pad_id)  (:1550-1565)
        return self.transformer(ids=input_ids, text=text, text_embeds=text_embeds, self_attn_mask=self_attn_mask,
                                labels=semantic_token_ids, **kwargs)


class CoarseTransformerWrapper(_WrapperBase):                 # audiolm_pytorch.py:1569-1854
    def __init__(self, *, transformer: CoarseTransformer, codec=None, wav2vec=None, audio_conditioner=None, pad_id=-1,
                 unique_consecutive=True, semantic_cross_entropy_loss_weight=1., mask_prob=0.15):
        super().__init__()
        self.codec = codec
        self.wav2vec = wav2vec
        self.transformer = transformer
        self.to(transformer.device)
        self.audio_conditioner = audio_conditioner
        assert not (exists(audio_conditioner) and not transformer.has_condition)
        self.unique_consecutive = unique_consecutive
        self.pad_id = pad_id
        self.semantic_cross_entropy_loss_weight = semantic_cross_entropy_loss_weight
        self.num_coarse_quantizers = transformer.num_coarse_quantizers * codec.rq_groups        # :1598 (codec=None crashes, like the reference)
        self.semantic_eos_id = transformer.semantic_eos_id
        self.coarse_eos_id = transformer.coarse_eos_id
        self.mask_prob = mask_prob

    @eval_decorator
    @torch.inference_mode()
    def generate(self, *, semantic_token_ids, prime_wave=None, prime_wave_input_sample_hz=None, prime_coarse_token_ids=None, text=None,
                 text_embeds=None, max_time_steps=512, cond_scale=3., filter_thres=0.9, temperature=1., reconstruct_wave=False,
                 use_kv_cache=True, **kwargs):
        """audiolm_pytorch.py:1608-1740 (prefix recomputed each step; see SemanticTransformerWrapper.generate)."""
        batch, device = semantic_token_ids.shape[0], self.device
        semantic_token_ids = semantic_token_ids.to(device)
        text_embeds = self._resolve_text(text, text_embeds, prime_wave, 'coarse')
        assert not (exists(prime_wave) and exists(prime_coarse_token_ids)), 'you can either pass in the prime as a raw wave (codec required) or as preprocessed acoustic token ids'
        if exists(prime_coarse_token_ids):
            coarse_token_ids = prime_coarse_token_ids
        elif exists(prime_wave):
            assert exists(self.codec)
            self.codec.eval()
            _, indices, _ = self.codec(prime_wave, return_encoded=True, input_sample_hz=prime_wave_input_sample_hz)
            coarse_token_ids = _flatten_ids(indices[..., :self.num_coarse_quantizers])
        else:
            coarse_token_ids = torch.empty((batch, 0), device=device, dtype=torch.long)
        if self.unique_consecutive:
            semantic_token_ids = batch_unique_consecutive(semantic_token_ids, pad_value=self.pad_id)
        sampled_coarse_token_ids = coarse_token_ids.clone()
        use_cache, state = use_kv_cache and not kwargs and not self.transformer.transformer.cond_as_self_attn_prefix, None
        nmax = semantic_token_ids.shape[1] + 2 + coarse_token_ids.shape[1] + max_time_steps * self.num_coarse_quantizers
        for time_step in range(0, max_time_steps):
            for ind in range(self.num_coarse_quantizers):
                just_finished_quantizer_step = (ind == 0 and time_step > 0)
                if use_cache:
                    last_coarse_logits, state = self.transformer.sample_logits(semantic_token_ids, sampled_coarse_token_ids, state, nmax,
                                                                               text_embeds=text_embeds, cond_scale=cond_scale)
                    last_coarse_logits = last_coarse_logits.clone()
                else:
                    _, coarse_logits = self.transformer.forward_with_cond_scale(coarse_token_ids=sampled_coarse_token_ids,
                                                                                semantic_token_ids=semantic_token_ids, text_embeds=text_embeds,
                                                                                cond_scale=cond_scale, return_only_coarse_logits=True, **kwargs)
                    last_coarse_logits = coarse_logits[:, -1].clone()
                if not just_finished_quantizer_step:
                    last_coarse_logits[:, -1] = float('-inf')          # prevent from eos in the middle of a time step
                filtered_logits = top_k(last_coarse_logits, thres=filter_thres)
                sampled = gumbel_sample(filtered_logits, temperature=temperature, dim=-1)
                sampled_coarse_token_ids = torch.cat((sampled_coarse_token_ids, sampled.unsqueeze(-1)), dim=-1)
        sampled_coarse_token_ids = mask_out_after_eos_id(sampled_coarse_token_ids, self.coarse_eos_id, keep_eos=False)
        sampled_coarse_token_ids = sampled_coarse_token_ids.reshape(batch, -1, self.num_coarse_quantizers)      # 'b (n q) -> b n q'
        if not reconstruct_wave:
            return sampled_coarse_token_ids
        return self._reconstruct(sampled_coarse_token_ids)

    def forward(self, *, semantic_token_ids=None, raw_wave=None, raw_wave_for_codec=None, text=None, text_embeds=None,
                coarse_token_ids=None, return_loss=False, **kwargs):
        assert exists(raw_wave) or exists(semantic_token_ids)
        raw_wave_for_codec = default(raw_wave_for_codec, raw_wave)
        assert exists(raw_wave_for_codec) or exists(coarse_token_ids)
        assert not all(map(exists, (raw_wave, raw_wave_for_codec, semantic_token_ids, coarse_token_ids)))
        if exists(self.audio_conditioner):                                                          # :1761-1764
            assert exists(raw_wave)
            assert not exists(text) and not exists(text_embeds)
            text_embeds = self.audio_conditioner(wavs=raw_wave, namespace='coarse')
        if not exists(semantic_token_ids):
            assert exists(self.wav2vec), 'VQWav2Vec must be be provided if given raw wave for training'
            semantic_token_ids = self.wav2vec(raw_wave, flatten=False)
        if not exists(coarse_token_ids):
            assert exists(self.codec), 'Codec must be provided if given raw wave for training'
            with torch.inference_mode():                                                          # :1773-1783
                self.codec.eval()
                _, indices, _ = self.codec(raw_wave_for_codec, return_encoded=True)
                batch, num_timesteps = raw_wave_for_codec.shape
                num_frames = int(num_timesteps / self.codec.seq_len_multiple_of)
                assert indices.shape[0] == batch and indices.shape[1] == num_frames
                coarse_token_ids = indices[..., :self.num_coarse_quantizers]
            coarse_token_ids = coarse_token_ids.clone()
        semantic_token_ids = _flatten_ids(semantic_token_ids)
        coarse_token_ids = _flatten_ids(coarse_token_ids)
        if (FUSED_PREPARE and self.training and return_loss and semantic_token_ids.is_cuda and semantic_token_ids.dtype == torch.int64
                and coarse_token_ids.dtype == torch.int64 and 'kv_cache' not in kwargs and 'embed_cache' not in kwargs and 'return_cache' not in kwargs):
            return self._forward_train_fused(semantic_token_ids, coarse_token_ids, text, text_embeds, kwargs)
        if self.training:                                                                         # :1788-1790
            semantic_token_ids = append_eos_id(semantic_token_ids, self.transformer.semantic_eos_id)
            coarse_token_ids = append_eos_id(coarse_token_ids, self.transformer.coarse_eos_id)
        if self.unique_consecutive:
            semantic_token_ids = batch_unique_consecutive(semantic_token_ids, pad_value=self.pad_id)
        if return_loss:
            semantic_labels, coarse_labels = semantic_token_ids, coarse_token_ids                  # (nothing below writes into the ids: no copy)
            coarse_token_ids = coarse_token_ids[:, :-1]
        self_attn_mask = (semantic_token_ids != self.pad_id) & (semantic_token_ids != self.semantic_eos_id)   # :1801
        semantic_token_ids = semantic_token_ids.masked_fill(~self_attn_mask, 0)
        coarse_token_len = coarse_token_ids.shape[-1]
        self_attn_mask = F.pad(self_attn_mask, (1, coarse_token_len + 1), value=True)             # :1805
        if self.mask_prob > 0 and self.training:                                                  # forgetful causal mask, :1809-1810
            self_attn_mask = _forgetful_and_(self_attn_mask, self.mask_prob)
        if not return_loss:
            return self.transformer(semantic_token_ids=semantic_token_ids, coarse_token_ids=coarse_token_ids,
                                    self_attn_mask=self_attn_mask, text=text, text_embeds=text_embeds, **kwargs)
        use_sem = self.semantic_cross_entropy_loss_weight > 0 and exists(self.transformer.to_semantic_logits)
        if not self.unique_consecutive:
            # the logit counts are plain integers here: the reference's weighted combination (:1826-1854) rides in the same kernel as the CE means
            n_c, n_s = coarse_labels.shape[-1], (semantic_labels.shape[-1] if use_sem else 0)
            w = (n_s * self.semantic_cross_entropy_loss_weight / (n_s + n_c), n_c / (n_s + n_c))
            return self.transformer(semantic_token_ids=semantic_token_ids, coarse_token_ids=coarse_token_ids, self_attn_mask=self_attn_mask, text=text,
                                    text_embeds=text_embeds, labels=(semantic_labels, coarse_labels), loss_weights=w, **kwargs)
        semantic_loss, coarse_loss = self.transformer(semantic_token_ids=semantic_token_ids, coarse_token_ids=coarse_token_ids,
                                                      self_attn_mask=self_attn_mask, text=text, text_embeds=text_embeds,
                                                      labels=(semantic_labels, coarse_labels), **kwargs)
        if self.unique_consecutive:                                                               # :1828-1831
            num_coarse_logits, _num_semantic_logits = coarse_labels.numel(), (semantic_labels != self.pad_id).sum()
        else:
            num_coarse_logits, _num_semantic_logits = coarse_labels.shape[-1], semantic_labels.shape[-1]
        num_semantic_logits = 0
        if not (use_sem and exists(semantic_loss)):
            semantic_loss = 0.
        else:
            num_semantic_logits = _num_semantic_logits
        return (semantic_loss * num_semantic_logits * self.semantic_cross_entropy_loss_weight +
                coarse_loss * num_coarse_logits) / (num_semantic_logits + num_coarse_logits)        # :1851-1854


    def _forward_train_fused(self, sem, coarse, text, text_embeds, kwargs):
        """The training step of forward() (:1785-1854) with its id bookkeeping as ONE kernel (ops.coarse_prepare: eos appended, key mask, zeroed masked ids,
        padded mask, embedding source codes, labels) instead of ~18 small ATen launches; same arithmetic, same RNG consumption (one randn for the
        forgetful mask).  Taken when the ids are given and the wrapper is in training mode; everything else runs forward().  unique_consecutive (the
        reference's default): one more launch collapses the runs of the [ids | eos] rows and ONE host read of the row lengths fixes the data-dependent
        width and the logit counts of the loss weights -- the reference formulation synchronises once per row."""
        tr = self.transformer
        sem = sem if sem.stride(1) == 1 else sem.contiguous()
        coarse = coarse if coarse.stride(1) == 1 else coarse.contiguous()
        use_sem = self.semantic_cross_entropy_loss_weight > 0 and exists(tr.to_semantic_logits)
        if self.unique_consecutive and sem.numel() > 0:
            full, lengths = ops.unique_consecutive(sem, tr.semantic_eos_id, self.pad_id)                  # :1788-1795
            tr.transformer.prepack_weights()             # width-independent work of the step goes out BEFORE the host read: the GPU idles for whatever the host
            lens = lengths.tolist()                      # still has to do between the read and the first big launch
            sem = full[:, :max(lens)]                                                                     # [ids | eos | pad ...]
            sem_labels, coarse_labels, src_a, keep = ops.coarse_prepare(sem, coarse, self.pad_id, tr.semantic_eos_id, tr.coarse_eos_id, tr.num_coarse_quantizers,
                                                                        tr.codebook_size, sem_has_eos=True)
            ns = sem.shape[1]
            n_c, n_s = coarse_labels.numel(), (sum(lens) if use_sem else 0)                               # :1828-1831 (numel / the non-pad semantic labels)
        else:
            sem_labels, coarse_labels, src_a, keep = ops.coarse_prepare(sem, coarse, self.pad_id, tr.semantic_eos_id, tr.coarse_eos_id, tr.num_coarse_quantizers,
                                                                        tr.codebook_size)
            ns = sem.shape[1] + 1
            if self.unique_consecutive:                                                                   # (an empty semantic prompt: nothing to collapse)
                n_c, n_s = coarse_labels.numel(), (sem_labels.numel() if use_sem else 0)
            else:
                n_c, n_s = coarse_labels.shape[-1], (sem_labels.shape[-1] if use_sem else 0)
        if self.mask_prob > 0:                                                                    # forgetful causal mask, :1809-1810
            keep = _forgetful_and_(keep, self.mask_prob)
        w = (n_s * self.semantic_cross_entropy_loss_weight / (n_s + n_c), n_c / (n_s + n_c))    # :1826-1854 with integer logit counts
        return tr(semantic_token_ids=sem, coarse_token_ids=coarse, self_attn_mask=keep, text=text, text_embeds=text_embeds,
                  labels=(sem_labels, coarse_labels), loss_weights=w, _prepared=(src_a, ns, coarse.shape[1]), **kwargs)


class FineTransformerWrapper(_WrapperBase):                   # audiolm_pytorch.py:1856-2137
    def __init__(self, *, transformer: FineTransformer, codec=None, audio_conditioner=None, coarse_cross_entropy_loss_weight=1.,
                 pad_id=-1, mask_prob=0.15):
        super().__init__()
        self.codec = codec
        self.transformer = transformer
        self.to(transformer.device)
        self.audio_conditioner = audio_conditioner
        assert not (exists(audio_conditioner) and not transformer.has_condition)
        self.num_fine_quantizers = transformer.num_fine_quantizers * codec.rq_groups
        self.num_coarse_quantizers = transformer.num_coarse_quantizers * codec.rq_groups
        if exists(codec):
            assert (self.num_fine_quantizers + self.num_coarse_quantizers) == (codec.num_quantizers * codec.rq_groups)
        self.eos_id = transformer.eos_id
        assert self.num_coarse_quantizers > 0
        self.pad_id = pad_id
        self.coarse_cross_entropy_loss_weight = coarse_cross_entropy_loss_weight
        self.mask_prob = mask_prob

    @eval_decorator
    @torch.inference_mode()
    def generate(self, *, coarse_token_ids, prime_wave=None, prime_wave_input_sample_hz=None, prime_fine_token_ids=None, text=None,
                 text_embeds=None, cond_scale=3., filter_thres=0.9, temperature=1., reconstruct_wave=False, use_kv_cache=True,
                 mask_out_generated_fine_tokens=False, **kwargs):
        """audiolm_pytorch.py:1896-2039 (prefix recomputed each step; see SemanticTransformerWrapper.generate)."""
        coarse_token_ids = _flatten_ids(coarse_token_ids)
        batch, device = coarse_token_ids.shape[0], self.device
        coarse_token_ids = coarse_token_ids.to(device)
        text_embeds = self._resolve_text(text, text_embeds, prime_wave, 'fine')
        assert not (exists(prime_wave) and exists(prime_fine_token_ids)), 'you can either pass in the prime as a raw wave (codec required) or as preprocessed acoustic token ids'
        if exists(prime_fine_token_ids):
            fine_token_ids = prime_fine_token_ids
        elif exists(prime_wave):
            assert exists(self.codec)
            self.codec.eval()
            _, token_ids, _ = self.codec(prime_wave, return_encoded=True, input_sample_hz=prime_wave_input_sample_hz)
            fine_token_ids = _flatten_ids(token_ids[..., self.num_coarse_quantizers:])
        else:
            fine_token_ids = torch.empty((batch, 0), device=device, dtype=torch.long)
        init_fine_time_step = fine_token_ids.shape[-1] // self.num_fine_quantizers
        max_time_steps = coarse_token_ids.shape[1] // self.num_coarse_quantizers
        sampled_fine_token_ids = fine_token_ids.clone()
        use_cache, state = use_kv_cache and not kwargs and not self.transformer.transformer.cond_as_self_attn_prefix, None
        max_fine_length = fine_token_ids.shape[-1] + max(0, max_time_steps - init_fine_time_step) * self.num_fine_quantizers
        for time_step in range(init_fine_time_step, max_time_steps):
            for ind in range(self.num_fine_quantizers):
                just_finished_quantizer_step = (ind == 0 and time_step > 0)
                if use_cache:
                    last_fine_logits, state = self.transformer.sample_logits(coarse_token_ids, sampled_fine_token_ids, state, max_fine_length,
                                                                             text_embeds=text_embeds, cond_scale=cond_scale)
                    last_fine_logits = last_fine_logits.clone()
                else:
                    _, fine_logits = self.transformer.forward_with_cond_scale(coarse_token_ids=coarse_token_ids,
                                                                              fine_token_ids=sampled_fine_token_ids, text_embeds=text_embeds,
                                                                              cond_scale=cond_scale, return_only_fine_logits=True, **kwargs)
                    last_fine_logits = fine_logits[:, -1].clone()
                if not just_finished_quantizer_step:
                    last_fine_logits[:, -1] = float('-inf')            # prevent from eos in the middle of a time step
                filtered_logits = top_k(last_fine_logits, thres=filter_thres)
                sampled = gumbel_sample(filtered_logits, temperature=temperature, dim=-1)
                sampled_fine_token_ids = torch.cat((sampled_fine_token_ids, sampled.unsqueeze(-1)), dim=-1)
        sampled_fine_token_ids = mask_out_after_eos_id(sampled_fine_token_ids, self.eos_id, keep_eos=False)
        sampled_fine_token_ids = sampled_fine_token_ids.reshape(batch, -1, self.num_fine_quantizers)            # 'b (n q) -> b n q'
        coarse_token_ids = coarse_token_ids.reshape(batch, -1, self.num_coarse_quantizers)
        if mask_out_generated_fine_tokens:
            pos_is_all_padding = (coarse_token_ids == self.pad_id).all(dim=-1, keepdim=True)
            sampled_fine_token_ids = sampled_fine_token_ids.masked_fill(pos_is_all_padding, self.pad_id)
        if not reconstruct_wave:
            return sampled_fine_token_ids
        return self._reconstruct(torch.cat((coarse_token_ids, sampled_fine_token_ids), dim=-1))

    def forward(self, *, raw_wave=None, text=None, text_embeds=None, token_ids=None, coarse_token_ids=None, fine_token_ids=None,
                return_loss=False, **kwargs):
        assert exists(raw_wave) ^ (exists(token_ids) ^ (exists(coarse_token_ids) and exists(fine_token_ids)))
        if exists(self.audio_conditioner):                                                          # :2055-2058
            assert exists(raw_wave)
            assert not exists(text) and not exists(text_embeds)
            text_embeds = self.audio_conditioner(wavs=raw_wave, namespace='fine')
        if exists(raw_wave):
            assert exists(self.codec), 'Codec must be provided if given raw wave for training'
            with torch.inference_mode():                                                          # :2063-2071
                self.codec.eval()
                _, token_ids, _ = self.codec(raw_wave, return_encoded=True)
                batch, num_timesteps = raw_wave.shape
                num_frames = int(num_timesteps / self.codec.seq_len_multiple_of)
                assert token_ids.shape == torch.Size((batch, num_frames, self.num_coarse_quantizers + self.num_fine_quantizers))
            token_ids = token_ids.clone()
        if exists(token_ids):
            coarse_token_ids, fine_token_ids = token_ids[..., :self.num_coarse_quantizers], token_ids[..., self.num_coarse_quantizers:]
        coarse_token_ids = _flatten_ids(coarse_token_ids)
        fine_token_ids = _flatten_ids(fine_token_ids)
        if return_loss:
            coarse_labels, fine_labels = coarse_token_ids, fine_token_ids
            fine_token_ids = fine_token_ids[:, :-1]
        self_attn_mask = None
        if self.mask_prob > 0 and self.training:
            mask_shape = (coarse_token_ids.shape[0], coarse_token_ids.shape[-1] + fine_token_ids.shape[-1] + 2)
            self_attn_mask = generate_mask_with_prob(mask_shape, self.mask_prob, device=self.device)
        if not return_loss:
            return self.transformer(coarse_token_ids=coarse_token_ids, fine_token_ids=fine_token_ids, self_attn_mask=self_attn_mask,
                                    text=text, text_embeds=text_embeds, **kwargs)
        use_coarse = self.coarse_cross_entropy_loss_weight > 0 and exists(self.transformer.coarse_logit_weights)
        # (coarse_loss * n_coarse * weight + fine_loss * n_fine) / (n_coarse + n_fine)  (:2112-2137): integers, folded into the kernel that takes the CE means
        n_f, n_c = fine_labels.shape[-1], (coarse_labels.shape[-1] if use_coarse else 0)
        w = (n_c * self.coarse_cross_entropy_loss_weight / (n_c + n_f), n_f / (n_c + n_f))
        return self.transformer(coarse_token_ids=coarse_token_ids, fine_token_ids=fine_token_ids, self_attn_mask=self_attn_mask, text=text,
                                text_embeds=text_embeds, labels=(coarse_labels, fine_labels), return_only_fine_logits=not use_coarse, loss_weights=w, **kwargs)


class AudioLM(nn.Module):                                     # audiolm_pytorch.py:2141-2254
    """Hierarchical sampling: semantic -> coarse -> fine -> waveform, every stage on the native path (kv-cache sampling, SoundStream decoder).
    Conditioned transformers take pre-computed `text_embeds` (or an audio conditioner callable); the T5 text encoder itself is out of scope."""

    def __init__(self, *, wav2vec, codec, semantic_transformer: SemanticTransformer, coarse_transformer: CoarseTransformer,
                 fine_transformer: FineTransformer, audio_conditioner=None, unique_consecutive=True):
        super().__init__()
        self.audio_conditioner = audio_conditioner
        assert semantic_transformer.num_semantic_tokens == coarse_transformer.num_semantic_tokens
        assert coarse_transformer.codebook_size == fine_transformer.codebook_size
        assert coarse_transformer.num_coarse_quantizers == fine_transformer.num_coarse_quantizers
        assert (fine_transformer.num_coarse_quantizers + fine_transformer.num_fine_quantizers) == codec.num_quantizers
        self.semantic_has_condition = semantic_transformer.has_condition
        self.coarse_has_condition = coarse_transformer.has_condition
        self.fine_has_condition = fine_transformer.has_condition
        self.needs_text = any([self.semantic_has_condition, self.coarse_has_condition, self.fine_has_condition])
        self.semantic = SemanticTransformerWrapper(wav2vec=wav2vec, transformer=semantic_transformer, audio_conditioner=audio_conditioner,
                                                   unique_consecutive=unique_consecutive)
        self.coarse = CoarseTransformerWrapper(wav2vec=wav2vec, codec=codec, transformer=coarse_transformer, audio_conditioner=audio_conditioner,
                                               unique_consecutive=unique_consecutive)
        self.fine = FineTransformerWrapper(codec=codec, transformer=fine_transformer, audio_conditioner=audio_conditioner)

    @property
    def device(self):
        return next(self.parameters()).device

    @eval_decorator
    @torch.inference_mode()
    def forward(self, *, batch_size=1, text=None, text_embeds=None, prime_wave=None, prime_wave_input_sample_hz=None, prime_wave_path=None,
                max_length=2048, return_coarse_generated_wave=False, mask_out_generated_fine_tokens=False):
        assert not (self.needs_text and (not exists(text) and not exists(text_embeds))), 'text needs to be passed in if one of the transformer requires conditioning'
        if self.needs_text and exists(text):
            text_embeds = self.semantic.embed_text(text)
        assert not (exists(prime_wave) and exists(prime_wave_path)), 'prompt audio must be given as either `prime_wave: Tensor` or `prime_wave_path: str`'
        if exists(prime_wave):
            assert exists(prime_wave_input_sample_hz), 'the input sample frequency for the prompt audio must be given as `prime_wave_input_sample_hz: int`'
            prime_wave = prime_wave.to(self.device)
        elif exists(prime_wave_path):
            raise NotImplementedError('loading audio files needs torchaudio (not part of this package): pass `prime_wave` as a tensor')
        semantic_token_ids = self.semantic.generate(text_embeds=text_embeds if self.semantic_has_condition else None, batch_size=batch_size,
                                                    prime_wave=prime_wave, prime_wave_input_sample_hz=prime_wave_input_sample_hz, max_length=max_length)
        coarse_token_ids_or_recon_wave = self.coarse.generate(text_embeds=text_embeds if self.coarse_has_condition else None,
                                                              semantic_token_ids=semantic_token_ids, prime_wave=prime_wave,
                                                              prime_wave_input_sample_hz=prime_wave_input_sample_hz,
                                                              reconstruct_wave=return_coarse_generated_wave)
        if return_coarse_generated_wave:
            return coarse_token_ids_or_recon_wave
        return self.fine.generate(text_embeds=text_embeds if self.fine_has_condition else None, coarse_token_ids=coarse_token_ids_or_recon_wave,
                                  prime_wave=prime_wave, prime_wave_input_sample_hz=prime_wave_input_sample_hz, reconstruct_wave=True,
                                  mask_out_generated_fine_tokens=mask_out_generated_fine_tokens)
