"""TEST INFRASTRUCTURE ONLY (oracle).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path never does.

Restatement of the third-party package `local-attention` (pinned `>=1.9.0` by /root/reference/setup.py, source NOT vendored under
/root/reference, no lockfile), as used by the reference's SoundStream at audiolm_pytorch/soundstream.py:26-27, 397-440:

    LocalTransformer.layers[i] = (LocalMHA(dim, heads, dim_head, qk_rmsnorm=True, window_size, use_rotary_pos_emb=True,
                                           gate_values_per_head=True, use_xpos=True, xpos_scale_base, prenorm=True, causal=True),
                                  FeedForward(dim))
    x = attn(x, attn_bias=None) + x ;  x = ff(x) + x

PARITY UNPINNED: there is no upstream source, test or golden vector for this component in /root/reference; the classes below restate the
published implementation (lucidrains/local-attention v1.9.x: local_attention.py, rotary.py, transformer.py) from knowledge, keeping its
structure literally -- bucketing into windows, `look_around`, rotary + xpos on the concatenated (look-back | own) window, the causal /
exact-window / pad masks -- so that the HIP kernel (csrc/local_attn.hip), which is written in a different, direct form (key j visible to
query i iff 0 <= i - j <= window), is checked against an independent formulation.  What is restated, in the order of the data flow:

  LocalMHA.forward:   x = LayerNorm(x) (prenorm);  q, k, v = to_qkv(x).chunk(3) as 'b n (h d) -> b h n d';
                      q, k = l2norm(q) * q_scale, l2norm(k) * k_scale (qk_rmsnorm; attention scale = qk_scale = 8 instead of d^-0.5);
                      out = LocalAttention(q, k, v);  out *= sigmoid(to_v_gate(x)) per head (gates from the NORMED x);  to_out(out)
  LocalAttention:     autopad to a multiple of the window (zeros, cut off again at the end), buckets 'b (w n) d -> b w n d', bq *= scale,
                      bk / bv = look_around(backward=1, forward=0, pad_value=-1), rotary(+xpos) with positions 0 .. 2W-1 for the keys and the
                      LAST W of them for the queries, sim = bq bk^T, masks (key after query | query - key > W * look_backward | padded
                      bucket), softmax, attn bv
  SinusoidalEmbeddings (xpos):  inv_freq = theta^(-arange(0, d, 2) / d), freqs = cat(t inv_freq, t inv_freq);
                      scale = ((arange(0, d, 2) + 0.4 d) / (1.4 d)) ** ((t - L // 2) / scale_base), scale_base = default(xpos_scale_base, W // 2);
                      q = (q cos + rotate_half(q) sin) * scale[-W:],  k = (k cos + rotate_half(k) sin) / scale;  rotate_half: (x1, x2) -> (-x2, x1)
  FeedForward(dim, mult=4): LayerNorm, Linear(dim, 2 * int(dim * mult * 2 / 3), bias=False), GEGLU (x * gelu(gate)), Dropout(0), Linear(inner, dim, bias=False)

It is used (a) as the `local_attention` stub when the real reference is imported to generate golden vectors
(tests/golden/make_golden.py soundstream_local_attn), and (b) directly by tests/test_gpu_codec.py against the HIP path.
`DynamicPositionBias` (SoundStream(attn_dynamic_pos_bias=True), not the default) is not restated.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import einsum, nn


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def l2norm(t):
    return F.normalize(t, dim=-1)


def max_neg_value(t):
    return -torch.finfo(t.dtype).max


def pad_to_multiple(t, multiple, dim=-1, value=0):
    seqlen = t.shape[dim]
    m = seqlen / multiple
    if m.is_integer():
        return False, t
    remainder = -(-seqlen // multiple) * multiple - seqlen
    pad_offset = (0,) * (-1 - dim) * 2
    return True, F.pad(t, (*pad_offset, 0, remainder), value=value)


def look_around(x, backward=1, forward=0, pad_value=-1, dim=2):
    t = x.shape[1]
    dims = (len(x.shape) - dim) * (0, 0)
    padded_x = F.pad(x, (*dims, backward, forward), value=pad_value)
    tensors = [padded_x[:, ind:(ind + t), ...] for ind in range(forward + backward + 1)]
    return torch.cat(tensors, dim=dim)


# ---- rotary.py ----------------------------------------------------------------------------------------------------------------------------

class SinusoidalEmbeddings(nn.Module):
    def __init__(self, dim, scale_base=None, use_xpos=False, theta=10000):
        super().__init__()
        inv_freq = 1. / (theta ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer('inv_freq', inv_freq)
        self.use_xpos = use_xpos
        self.scale_base = scale_base
        assert not (use_xpos and not exists(scale_base)), 'scale base must be defined if using xpos'
        scale = (torch.arange(0, dim, 2) + 0.4 * dim) / (1.4 * dim)
        self.register_buffer('scale', scale, persistent=False)

    def forward(self, x):
        seq_len, device = x.shape[-2], x.device
        t = torch.arange(seq_len, device=device).type_as(self.inv_freq)
        freqs = torch.einsum('i , j -> i j', t, self.inv_freq)
        freqs = torch.cat((freqs, freqs), dim=-1)
        if not self.use_xpos:
            return freqs, torch.ones(1, device=device)
        power = (t - (seq_len // 2)) / self.scale_base
        scale = self.scale ** power[:, None]
        scale = torch.cat((scale, scale), dim=-1)
        return freqs, scale


def rotate_half(x):
    x1, x2 = x.reshape(*x.shape[:-1], 2, x.shape[-1] // 2).unbind(dim=-2)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(q, k, freqs, scale=1):
    q_len = q.shape[-2]
    q_freqs = freqs[..., -q_len:, :]
    inv_scale = scale ** -1
    if scale.ndim == 2:
        scale = scale[-q_len:, :]
    q = (q * q_freqs.cos() * scale) + (rotate_half(q) * q_freqs.sin() * scale)
    k = (k * freqs.cos() * inv_scale) + (rotate_half(k) * freqs.sin() * inv_scale)
    return q, k


# ---- local_attention.py -------------------------------------------------------------------------------------------------------------------

class LocalAttention(nn.Module):
    def __init__(self, window_size, causal=False, look_backward=1, look_forward=None, dropout=0., shared_qk=False, rel_pos_emb_config=None, dim=None,
                 autopad=False, exact_windowsize=False, scale=None, use_rotary_pos_emb=True, use_xpos=False, xpos_scale_base=None):
        super().__init__()
        look_forward = default(look_forward, 0 if causal else 1)
        assert not (causal and look_forward > 0), 'you cannot look forward if causal'
        assert not shared_qk and dropout == 0., 'not restated (unused by the reference)'
        self.scale = scale
        self.window_size = window_size
        self.autopad = autopad
        self.exact_windowsize = exact_windowsize
        self.causal = causal
        self.look_backward = look_backward
        self.look_forward = look_forward
        self.rel_pos = None
        self.use_xpos = use_xpos
        if use_rotary_pos_emb and (exists(rel_pos_emb_config) or exists(dim)):
            if exists(rel_pos_emb_config):
                dim = rel_pos_emb_config[0]
            self.rel_pos = SinusoidalEmbeddings(dim, use_xpos=use_xpos, scale_base=default(xpos_scale_base, window_size // 2))

    def forward(self, q, k, v, mask=None, input_mask=None, attn_bias=None, window_size=None):
        assert mask is None and input_mask is None and attn_bias is None and window_size is None, 'not restated (unused by the reference defaults)'
        autopad, pad_value, window_size, causal, look_backward, look_forward = self.autopad, -1, self.window_size, self.causal, self.look_backward, self.look_forward
        lead = q.shape[:-2]
        q, k, v = (t.reshape(-1, *t.shape[-2:]) for t in (q, k, v))                      # pack([t], '* n d')
        if autopad:
            orig_seq_len = q.shape[1]
            (_, q), (_, k), (_, v) = (pad_to_multiple(t, self.window_size, dim=-2) for t in (q, k, v))
        b, n, dim_head = q.shape
        device = q.device
        scale = default(self.scale, dim_head ** -0.5)
        assert (n % window_size) == 0, f'sequence length {n} must be divisible by window size {window_size} for local attention'
        windows = n // window_size
        seq = torch.arange(n, device=device)
        b_t = seq.reshape(1, windows, window_size)
        bq, bk, bv = (t.reshape(b, windows, window_size, dim_head) for t in (q, k, v))
        bq = bq * scale
        look_around_kwargs = dict(backward=look_backward, forward=look_forward, pad_value=pad_value)
        bk = look_around(bk, **look_around_kwargs)
        bv = look_around(bv, **look_around_kwargs)
        if exists(self.rel_pos):
            pos_emb, xpos_scale = self.rel_pos(bk)
            bq, bk = apply_rotary_pos_emb(bq, bk, pos_emb, scale=xpos_scale)
        bq_t = b_t
        bq_k = look_around(b_t, **look_around_kwargs)
        bq_t = bq_t[..., :, None]
        bq_k = bq_k[..., None, :]
        pad_mask = bq_k == pad_value
        sim = einsum('b h i e, b h j e -> b h i j', bq, bk)
        mask_value = max_neg_value(sim)
        if causal:
            causal_mask = bq_t < bq_k
            if self.exact_windowsize:
                max_causal_window_size = self.window_size * self.look_backward
                causal_mask = causal_mask | (bq_t > (bq_k + max_causal_window_size))
            sim = sim.masked_fill(causal_mask, mask_value)
        if not causal and self.exact_windowsize:
            max_backward_window_size = self.window_size * self.look_backward
            max_forward_window_size = self.window_size * self.look_forward
            window_mask = ((bq_k - max_forward_window_size) > bq_t) | (bq_t > (bq_k + max_backward_window_size)) | pad_mask
            sim = sim.masked_fill(window_mask, mask_value)
        else:
            sim = sim.masked_fill(pad_mask, mask_value)
        attn = sim.softmax(dim=-1)
        out = einsum('b h i j, b h j e -> b h i e', attn, bv)
        out = out.reshape(b, windows * window_size, dim_head)
        if autopad:
            out = out[:, :orig_seq_len, :]
        return out.reshape(*lead, *out.shape[-2:])


# ---- transformer.py -----------------------------------------------------------------------------------------------------------------------

class LocalMHA(nn.Module):
    def __init__(self, *, dim, window_size, dim_head=64, heads=8, dropout=0., causal=False, prenorm=False, qk_rmsnorm=False, qk_scale=8,
                 use_xpos=False, xpos_scale_base=None, exact_windowsize=None, gate_values_per_head=False, **kwargs):
        super().__init__()
        inner_dim = dim_head * heads
        self.norm = nn.LayerNorm(dim) if prenorm else None
        self.heads = heads
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.qk_rmsnorm = qk_rmsnorm
        if qk_rmsnorm:
            self.q_scale = nn.Parameter(torch.ones(dim_head))
            self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.causal = causal
        self.window_size = window_size
        self.exact_windowsize = default(exact_windowsize, True)
        self.attn_fn = LocalAttention(dim=dim_head, window_size=window_size, causal=causal, autopad=True,
                                      scale=(qk_scale if qk_rmsnorm else None), exact_windowsize=self.exact_windowsize, use_xpos=use_xpos,
                                      xpos_scale_base=xpos_scale_base, **kwargs)
        self.to_v_gate = None
        if gate_values_per_head:
            self.to_v_gate = nn.Sequential(nn.Linear(dim, heads))
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, x, mask=None, attn_bias=None, cache=None, return_cache=False):
        assert mask is None and attn_bias is None and cache is None and not return_cache, 'not restated (unused by the reference defaults)'
        if exists(self.norm):
            x = self.norm(x)
        b, n, _ = x.shape
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        q, k, v = (t.reshape(b, n, self.heads, -1).transpose(1, 2) for t in (q, k, v))           # 'b n (h d) -> b h n d'
        if self.qk_rmsnorm:
            q, k = map(l2norm, (q, k))
            q = q * self.q_scale
            k = k * self.k_scale
        out = self.attn_fn(q, k, v)
        if exists(self.to_v_gate):
            gates = self.to_v_gate(x)
            gates = gates.transpose(1, 2)[..., None]                                            # 'b n h -> b h n 1'
            out = out * gates.sigmoid()
        out = out.transpose(1, 2).reshape(b, n, -1)                                              # 'b h n d -> b n (h d)'
        return self.to_out(out)


class GEGLU(nn.Module):
    def forward(self, x):
        x, gate = x.chunk(2, dim=-1)
        return x * F.gelu(gate)


def FeedForward(dim, mult=4, dropout=0.):
    inner_dim = int(dim * mult * 2 / 3)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner_dim * 2, bias=False), GEGLU(), nn.Dropout(dropout), nn.Linear(inner_dim, dim, bias=False))


class DynamicPositionBias(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError('DynamicPositionBias (SoundStream(attn_dynamic_pos_bias=True), not the default) is not restated')
