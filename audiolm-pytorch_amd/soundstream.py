"""Mirror of the reference's `audiolm_pytorch/soundstream.py` for the TOKENIZE path and the DECODE path
(§8(f) item 3, incl. the LocalTransformer of the reference-default `use_local_attn=True`): `SoundStream.tokenize(audio)`, `SoundStream.forward(x, return_encoded=True |
return_codes_only=True)` -- the causal-conv encoder (soundstream.py:332-380, 519-531) and the eval-mode forward of the grouped residual VQ
(soundstream.py:592-607, :840) -- and `decode_from_codebook_indices` / `decode` (soundstream.py:691-709: code lookup, transposed-conv
decoder :347-360, 382-395, 615-627) run on the MI355X kernels of csrc/codec.hip (exact-fp32 MFMA).  Everything else the reference class
does (discriminators, losses, training of the codec, LFQ / FSQ quantizers) is out of scope and raises.
`encoder_attn` / `decoder_attn` (soundstream.py:397-440, 545, 613): local-attention's LocalMHA + FeedForward (third-party source, not vendored:
restated, parity unpinned -- oracle/local_attention_restated.py) run in the codec's [B, C, T] layout: LayerNorm / windowed causal attention with
qk-l2norm, rotary + xpos and per-head value gates / GEGLU are csrc/local_attn.hip, the Linear layers are k = 1 convs on the exact-fp32 MFMA kernel.

The module tree keeps the reference's parameter / buffer NAMES for the parts it has, so `state_dict()` entries `encoder.*` and
`rq.*` of a reference checkpoint load with `load_state_dict(..., strict=False)`:
    encoder.0.conv.{weight,bias}                        CausalConv1d(input_channels -> channels, 7)
    encoder.{b}.{r}.fn.{0,2}.conv.{weight,bias}         ResidualUnit r of EncoderBlock b (k7 dilated conv, ELU, k1 conv, ELU, + x)
    encoder.{b}.3.conv.{weight,bias}                    strided down-sampling conv (k = 2 * stride)
    encoder.{last}.conv.{weight,bias}                   CausalConv1d(-> codebook_dim, 3)
    {encoder,decoder}_attn.layers.{i}.0.{norm.{weight,bias}, to_qkv.weight, q_scale, k_scale, attn_fn.rel_pos.inv_freq, to_v_gate.0.{weight,bias},
                                         to_out.weight}   LocalMHA;   .layers.{i}.1.{0.{weight,bias}, 1.weight, 4.weight}   FeedForward
    decoder.0.conv / decoder.{b}.0.conv (ConvTranspose1d) / decoder.{b}.{1,2,3}.fn.{0,2}.conv / decoder.{last}.conv
    rq.rvqs.{g}.layers.{q}._codebook.{initted, cluster_size, embed_avg, embed (1, C, d)}
"""
from __future__ import annotations

import functools
import os
from itertools import cycle

import torch
from torch import nn

from . import core, ops

F32 = torch.float32
FUSE_RESUNIT = os.environ.get('ALM_FUSE_RESUNIT', '1') != '0'      # A/B switch: 0 = the two alm_conv1d_causal launches per ResidualUnit


class CausalConv1d(nn.Module):                                   # soundstream.py:332-345
    def __init__(self, chan_in, chan_out, kernel_size, pad_mode='reflect', **kwargs):
        super().__init__()
        if pad_mode != 'reflect':
            raise NotImplementedError('only the reference default pad_mode="reflect" is implemented')
        self.dilation = kwargs.get('dilation', 1)
        self.stride = kwargs.get('stride', 1)
        self.kernel_size = kernel_size
        self.pad_mode = pad_mode
        self.causal_padding = self.dilation * (kernel_size - 1) + (1 - self.stride)
        self.conv = nn.Conv1d(chan_in, chan_out, kernel_size, **kwargs)
        self._packed = None

    def packed(self):
        w = self.conv.weight
        ver = (w.data_ptr(), core.tensor_version(w))
        if self._packed is None or self._packed[0] != ver:
            self._packed = (ver, ops.conv1d_pack(w.detach().to(F32)))
        return self._packed[1]

    def run(self, x, *, elu=False, residual=None):
        return ops.conv1d_causal(x, self.packed(), self.conv.bias.detach(), self.conv.out_channels, self.kernel_size, stride=self.stride,
                                 dilation=self.dilation, elu=elu, residual=residual)

    def forward(self, x):
        return self.run(x)


class CausalConvTranspose1d(nn.Module):                          # soundstream.py:347-360
    """ConvTranspose1d(k = 2 * stride, stride) cut to n * stride outputs.  Output t = q * stride + r depends on input frames q and q - 1
    only (taps r and r + stride), so it runs as the k = 2, zero-left-padded causal conv over `stride` phase-major copies of the output
    channels (alm_conv1d_causal, exact-fp32 MFMA), followed by the phase interleave."""

    def __init__(self, chan_in, chan_out, kernel_size, stride, **kwargs):
        super().__init__()
        if kernel_size != 2 * stride or kwargs:
            raise NotImplementedError('only the reference decoder form is implemented: kernel_size = 2 * stride, default ConvTranspose1d options')
        self.upsample_factor = stride
        self.padding = kernel_size - 1
        self.conv = nn.ConvTranspose1d(chan_in, chan_out, kernel_size, stride)
        self._packed = None

    def packed(self):
        w, b = self.conv.weight, self.conv.bias
        ver = (w.data_ptr(), core.tensor_version(w), core.tensor_version(b))
        if self._packed is None or self._packed[0] != ver:
            s = self.upsample_factor
            cin, cout, _ = w.shape
            wd = w.detach().to(F32)                                                  # [Cin, Cout, 2 s]
            # W2[(r, co), ci, tap]: tap 0 <- x[q - 1] uses w[ci, co, r + s]; tap 1 <- x[q] uses w[ci, co, r]
            w2 = torch.stack((wd[:, :, s:], wd[:, :, :s]), dim=-1)                   # [Cin, Cout, s(r), 2(tap)]
            w2 = w2.permute(2, 1, 0, 3).reshape(s * cout, cin, 2).contiguous()
            self._packed = (ver, ops.conv1d_pack(w2), b.detach().to(F32).repeat(s).contiguous())
        return self._packed[1], self._packed[2]

    def forward(self, x):
        wp, b2 = self.packed()
        s, cout = self.upsample_factor, self.conv.out_channels
        y = ops.conv1d_causal(x, wp, b2, s * cout, 2, zero_pad=True)
        return ops.phase_interleave(y, cout, s)


class _ResidualFn(nn.Module):
    """holder with the reference's `.fn` Sequential naming: fn.0 = dilated k7 conv, fn.2 = k1 conv (fn.1 / fn.3 are ELUs)."""

    def __init__(self, chan, dilation, kernel_size, pad_mode):
        super().__init__()
        self.fn = nn.Sequential(CausalConv1d(chan, chan, kernel_size, dilation=dilation, pad_mode=pad_mode), nn.ELU(),
                                CausalConv1d(chan, chan, 1, pad_mode=pad_mode), nn.ELU())

    def forward(self, x):                                        # soundstream.py:362-369: ELU(conv1(ELU(conv7(x)))) + x
        c7, c1 = self.fn[0], self.fn[2]
        if FUSE_RESUNIT and ops.resunit_supported(x.shape[1]) and c7.dilation * (c7.kernel_size - 1) < x.shape[2]:
            # one launch, the intermediate in registers (alm_resunit_causal: bitwise equal to the two launches below)
            return ops.resunit_causal(x, c7.packed(), c7.conv.bias.detach(), c1.packed(), c1.conv.bias.detach(), c7.kernel_size, c7.dilation)
        h = c7.run(x, elu=True)
        return c1.run(h, elu=True, residual=x)


def ResidualUnit(chan_in, chan_out, dilation, kernel_size=7, squeeze_excite=False, pad_mode='reflect'):
    if squeeze_excite:
        raise NotImplementedError('squeeze_excite is not on the tokenize hot path (reference default False)')
    assert chan_in == chan_out
    return _ResidualFn(chan_in, dilation, kernel_size, pad_mode)


def EncoderBlock(chan_in, chan_out, stride, cycle_dilations=(1, 3, 9), squeeze_excite=False, pad_mode='reflect'):   # soundstream.py:371-380
    it = cycle(cycle_dilations)
    return nn.Sequential(ResidualUnit(chan_in, chan_in, next(it), squeeze_excite=squeeze_excite, pad_mode=pad_mode),
                         ResidualUnit(chan_in, chan_in, next(it), squeeze_excite=squeeze_excite, pad_mode=pad_mode),
                         ResidualUnit(chan_in, chan_in, next(it), squeeze_excite=squeeze_excite, pad_mode=pad_mode),
                         CausalConv1d(chan_in, chan_out, 2 * stride, stride=stride, pad_mode=pad_mode))


def DecoderBlock(chan_in, chan_out, stride, cycle_dilations=(1, 3, 9), squeeze_excite=False, pad_mode='reflect'):   # soundstream.py:382-395
    it = cycle(cycle_dilations)
    return nn.Sequential(CausalConvTranspose1d(chan_in, chan_out, 2 * stride, stride=stride),
                         ResidualUnit(chan_out, chan_out, next(it), squeeze_excite=squeeze_excite, pad_mode=pad_mode),
                         ResidualUnit(chan_out, chan_out, next(it), squeeze_excite=squeeze_excite, pad_mode=pad_mode),
                         ResidualUnit(chan_out, chan_out, next(it), squeeze_excite=squeeze_excite, pad_mode=pad_mode))


class _EuclideanCodebook(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        self.register_buffer('initted', torch.tensor([False]))
        self.register_buffer('cluster_size', torch.ones(1, codebook_size))
        self.register_buffer('embed_avg', torch.zeros(1, codebook_size, dim))
        self.register_buffer('embed', torch.zeros(1, codebook_size, dim))


class _VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        self._codebook = _EuclideanCodebook(dim, codebook_size)


class _ResidualVQ(nn.Module):
    def __init__(self, dim, num_quantizers, codebook_size):
        super().__init__()
        self.layers = nn.ModuleList([_VectorQuantize(dim, codebook_size) for _ in range(num_quantizers)])


class GroupedResidualVQ(nn.Module):
    """eval-mode forward of vector-quantize-pytorch's GroupedResidualVQ as the reference builds it (soundstream.py:592-607)."""

    def __init__(self, *, dim, groups=1, num_quantizers, codebook_size, **unused):
        super().__init__()
        assert dim % groups == 0
        self.dim, self.groups, self.num_quantizers, self.codebook_size = dim, groups, num_quantizers, codebook_size
        self.rvqs = nn.ModuleList([_ResidualVQ(dim // groups, num_quantizers, codebook_size) for _ in range(groups)])
        self._packed = None

    def _pack(self):
        embeds = [[l._codebook.embed for l in r.layers] for r in self.rvqs]
        ver = tuple((e.data_ptr(), core.tensor_version(e)) for row in embeds for e in row)
        if self._packed is None or self._packed[0] != ver:
            for r in self.rvqs:
                for l in r.layers:
                    if not bool(l._codebook.initted.item()):
                        raise RuntimeError('codebooks are not initialised (`initted` is False): the k-means initialisation of the first training '
                                           'batch (soundstream.py:600) is not part of the tokenize path -- load a trained codec or set them')
            packs = []
            for row in embeds:
                E = torch.stack([e[0].detach().to(F32) for e in row]).contiguous()       # [Q, C, d]
                packs.append((E,) + ops.rvq_pack(E))
            self._packed = (ver, packs)
        return self._packed[1]

    def forward(self, x):
        """x fp32 (b, n, dim) -> (quantized (b, n, dim), indices (g, b, n, q) int64, commit_loss zeros (g, q))."""
        if self.training:
            raise NotImplementedError('only the eval-mode forward (tokenize) is implemented')
        b, n, dim = x.shape
        x2 = x.reshape(b * n, dim).to(F32).contiguous()
        dg = dim // self.groups
        quant = torch.empty_like(x2)
        idx = torch.empty((self.groups, b * n, self.num_quantizers), dtype=torch.int64, device=x.device)
        for g, (E, Et, e2) in enumerate(self._pack()):
            ops.rvq_encode(x2[:, g * dg:(g + 1) * dg], E, Et, e2, idx_out=idx[g], quant_out=quant[:, g * dg:(g + 1) * dg])
        return quant.view(b, n, dim), idx.view(self.groups, b, n, self.num_quantizers), torch.zeros((self.groups, self.num_quantizers), device=x.device)


    @torch.no_grad()
    def get_output_from_indices(self, indices):
        """indices int (g, b, n, q) (-1 = no code) -> (b, n, dim): per group the sum of the selected code vectors (alm_rvq_decode)."""
        g, b, n, q = indices.shape
        assert g == self.groups and q <= self.num_quantizers
        dg = self.dim // self.groups
        out = torch.empty((b * n, self.dim), dtype=F32, device=indices.device)
        idx = indices.to(torch.int64).reshape(g, b * n, q).contiguous()
        for gi, (E, _, _) in enumerate(self._pack()):
            ops.rvq_decode(idx[gi], E[:q].contiguous(), out[:, gi * dg:(gi + 1) * dg])
        return out.view(b, n, self.dim)


# ---------------------------------------------------------------------------------------------- LocalTransformer (soundstream.py:397-440)

class _Linear1x1:
    """an nn.Linear applied along the channel axis of [B, C, T] = a k = 1 conv on the exact-fp32 MFMA kernel; packed weight cached per version"""

    def __init__(self):
        self._packed = None

    def __call__(self, lin, x, residual=None):
        w, b = lin.weight, lin.bias
        ver = (w.data_ptr(), core.tensor_version(w), None if b is None else core.tensor_version(b))
        if self._packed is None or self._packed[0] != ver:
            bias = b.detach().to(F32).contiguous() if b is not None else torch.zeros(w.shape[0], dtype=F32, device=w.device)
            self._packed = (ver, ops.conv1d_pack(w.detach().to(F32).unsqueeze(-1).contiguous()), bias)
        return ops.conv1d_causal(x, self._packed[1], self._packed[2], w.shape[0], 1, residual=residual)


class _SinusoidalEmbeddings(nn.Module):                          # local-attention rotary.py (holder of the `inv_freq` buffer)
    def __init__(self, dim, scale_base, theta=10000):
        super().__init__()
        self.register_buffer('inv_freq', 1. / (theta ** (torch.arange(0, dim, 2).float() / dim)))
        self.dim, self.scale_base = dim, scale_base

    def tables(self, slots, device):
        """rotary angle cos / sin and the xpos scale for slots 0 .. slots-1 of the (look-back | own) window pair, fp32 [slots, dim] each,
        computed on the host exactly like the library does (t * inv_freq, scale ** ((t - slots // 2) / scale_base))"""
        inv = self.inv_freq.detach().float().cpu()
        t = torch.arange(slots).float()
        freqs = torch.einsum('i,j->ij', t, inv)
        freqs = torch.cat((freqs, freqs), dim=-1)
        base = (torch.arange(0, self.dim, 2) + 0.4 * self.dim) / (1.4 * self.dim)
        scale = base ** ((t - (slots // 2)) / self.scale_base)[:, None]
        scale = torch.cat((scale, scale), dim=-1)
        return tuple(x.float().contiguous().to(device) for x in (freqs.cos(), freqs.sin(), scale))


class _LocalAttention(nn.Module):
    def __init__(self, dim_head, window_size, xpos_scale_base):
        super().__init__()
        self.rel_pos = _SinusoidalEmbeddings(dim_head, scale_base=xpos_scale_base if xpos_scale_base is not None else window_size // 2)


class LocalMHA(nn.Module):
    """local-attention's LocalMHA in the one configuration the reference builds (soundstream.py:418-427): prenorm LayerNorm, causal, look-back of
    one window with the exact window size, qk_rmsnorm (attention scale 8), rotary + xpos, per-head sigmoid value gates."""

    def __init__(self, *, dim, window_size, dim_head=64, heads=8, xpos_scale_base=None, qk_scale=8):
        super().__init__()
        inner = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.heads, self.dim_head, self.window_size, self.qk_scale = heads, dim_head, window_size, qk_scale
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.attn_fn = _LocalAttention(dim_head, window_size, xpos_scale_base)
        self.to_v_gate = nn.Sequential(nn.Linear(dim, heads))
        self.to_out = nn.Linear(inner, dim, bias=False)
        self._lin = [_Linear1x1() for _ in range(3)]
        self._tables = None

    def run(self, x, add_residual=True):
        """x fp32 [B, dim, T] -> attn(x) (+ x), same layout"""
        inv = self.attn_fn.rel_pos.inv_freq
        key = (inv.data_ptr(), core.tensor_version(inv), x.device)
        if self._tables is None or self._tables[0] != key:
            self._tables = (key, self.attn_fn.rel_pos.tables(2 * self.window_size, x.device))
        cos_t, sin_t, xpos_t = self._tables[1]
        xn = ops.layernorm_bct(x, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps)
        qkv = self._lin[0](self.to_qkv, xn)
        gates = self._lin[1](self.to_v_gate[0], xn)                      # from the NORMED input (LocalMHA.forward re-binds x)
        o = ops.local_attn(qkv, self.q_scale.detach(), self.k_scale.detach(), cos_t, sin_t, xpos_t, gates, self.heads, self.dim_head,
                           self.window_size, self.qk_scale)
        return self._lin[2](self.to_out, o, residual=x if add_residual else None)

    def forward(self, x):
        """reference layout: x (b, n, dim) -> attention output WITHOUT the residual (the caller adds it, soundstream.py:437)"""
        return ops.bct_to_btc(self.run(ops.bct_to_btc(x.to(F32).contiguous()), add_residual=False))      # (n, c) -> (c, n) and back


class _GEGLU(nn.Module):
    def forward(self, x):
        raise NotImplementedError('runs fused inside LocalTransformer (csrc/local_attn.hip alm_geglu_bct)')


def _local_feed_forward(dim, mult=4):                                # local_attention.transformer.FeedForward: same Sequential indices
    inner = int(dim * mult * 2 / 3)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), _GEGLU(), nn.Dropout(0.), nn.Linear(inner, dim, bias=False))


class LocalTransformer(nn.Module):
    def __init__(self, *, dim, depth, heads, window_size, dynamic_pos_bias=False, **kwargs):
        super().__init__()
        if dynamic_pos_bias:
            raise NotImplementedError('attn_dynamic_pos_bias=True (DynamicPositionBias instead of rotary) is not implemented (reference default False)')
        kwargs.pop('prenorm', None), kwargs.pop('causal', None)           # always True in the reference (soundstream.py:541-542)
        self.window_size = window_size
        self.pos_bias = None
        self.layers = nn.ModuleList([nn.ModuleList([LocalMHA(dim=dim, heads=heads, window_size=window_size, **kwargs), _local_feed_forward(dim)])
                                     for _ in range(depth)])
        self._ff_lin = [[_Linear1x1(), _Linear1x1()] for _ in range(depth)]

    def run_bct(self, x):
        """x fp32 [B, dim, T] (codec layout) -> same: x = attn(x) + x; x = ff(x) + x per layer (soundstream.py:436-438)"""
        for (attn, ff), lins in zip(self.layers, self._ff_lin):
            x = attn.run(x)
            h = ops.layernorm_bct(x, ff[0].weight.detach(), ff[0].bias.detach(), ff[0].eps)
            h = ops.geglu_bct(lins[0](ff[1], h))
            x = lins[1](ff[4], h, residual=x)
        return x

    @torch.no_grad()
    def forward(self, x):
        """x (b, n, dim) -> (b, n, dim)"""
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd.SoundStream runs on the MI355X only (no CPU fallback)')
        return ops.bct_to_btc(self.run_bct(ops.bct_to_btc(x.to(F32).contiguous())))


def curtail_to_multiple(t, mult, from_left=False):               # soundstream.py:86-90
    data_len = t.shape[-1]
    rounded = (data_len // mult) * mult
    return t[..., :rounded] if not from_left else t[..., -rounded:]


class SoundStream(nn.Module):
    """Constructor keywords and defaults follow the reference (soundstream.py:451-510); options outside the tokenize path raise."""

    def __init__(self, *, channels=32, strides=(2, 4, 5, 8), channel_mults=(2, 4, 8, 16), codebook_dim=512, codebook_size=None,
                 finite_scalar_quantizer_levels=None, rq_num_quantizers=8, rq_commitment_weight=1., rq_ema_decay=0.95,
                 rq_quantize_dropout_multiple_of=1, rq_groups=1, rq_stochastic_sample_codes=False, rq_rotation_trick=True, rq_kwargs: dict = {},
                 use_lookup_free_quantizer=False, use_finite_scalar_quantizer=False, input_channels=1, enc_cycle_dilations=(1, 3, 9),
                 target_sample_hz=16000, use_local_attn=True, attn_window_size=128, attn_dim_head=64, attn_heads=8, attn_depth=1,
                 attn_xpos_scale_base=None, attn_dynamic_pos_bias=False, use_gate_loop_layers=False, squeeze_excite=False, pad_mode='reflect', **kwargs):
        super().__init__()
        if use_lookup_free_quantizer or use_finite_scalar_quantizer or finite_scalar_quantizer_levels is not None:
            raise NotImplementedError('LFQ / FSQ quantizers are out of scope (SURVEY.md §2)')
        if use_gate_loop_layers or squeeze_excite or rq_stochastic_sample_codes:
            raise NotImplementedError('gate-loop layers / squeeze-excite / stochastic code sampling are not on the tokenize hot path')
        assert codebook_size is not None, '`codebook_size` must be set'
        self.target_sample_hz = target_sample_hz
        self.single_channel = input_channels == 1
        self.strides = strides
        layer_channels = (channels, *[m * channels for m in channel_mults])
        pairs = tuple(zip(layer_channels[:-1], layer_channels[1:]))
        blocks = [EncoderBlock(ci, co, s, enc_cycle_dilations, squeeze_excite, pad_mode) for (ci, co), s in zip(pairs, strides)]
        self.encoder = nn.Sequential(CausalConv1d(input_channels, channels, 7, pad_mode=pad_mode), *blocks,
                                     CausalConv1d(layer_channels[-1], codebook_dim, 3, pad_mode=pad_mode))
        attn_kwargs = dict(dim=codebook_dim, dim_head=attn_dim_head, heads=attn_heads, depth=attn_depth, window_size=attn_window_size,
                           xpos_scale_base=attn_xpos_scale_base, dynamic_pos_bias=attn_dynamic_pos_bias, prenorm=True, causal=True)   # soundstream.py:533-543
        self.encoder_attn = LocalTransformer(**attn_kwargs) if use_local_attn else None
        self.decoder_attn = LocalTransformer(**attn_kwargs) if use_local_attn else None
        dec_cycle_dilations = kwargs.pop('dec_cycle_dilations', (1, 3, 9))          # remaining kwargs: attention / discriminator / loss options of
                                                                                    # the parts that are not built here (ignored, like before)
        dblocks = [DecoderBlock(co, ci, s, dec_cycle_dilations, squeeze_excite, pad_mode) for (ci, co), s in reversed(tuple(zip(pairs, strides)))]
        self.decoder = nn.Sequential(CausalConv1d(codebook_dim, layer_channels[-1], 7, pad_mode=pad_mode), *dblocks,
                                     CausalConv1d(channels, input_channels, 7, pad_mode=pad_mode))             # soundstream.py:615-627
        self.num_quantizers = rq_num_quantizers
        self.codebook_dim = codebook_dim
        self.rq_groups = rq_groups
        self.codebook_size = codebook_size
        self.rq = GroupedResidualVQ(dim=codebook_dim, num_quantizers=rq_num_quantizers, codebook_size=codebook_size, groups=rq_groups)
        self.eval()

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def seq_len_multiple_of(self):                               # soundstream.py:772-774
        return functools.reduce(lambda x, y: x * y, self.strides)

    @property
    def downsample_factor(self):
        return self.seq_len_multiple_of

    def process_input(self, x, input_sample_hz=None, curtail_from_left=False):                   # soundstream.py:779-795
        if input_sample_hz is not None and input_sample_hz != self.target_sample_hz:
            raise NotImplementedError('on-the-fly resampling (torchaudio) is outside the hot path: resample before tokenizing')
        lead = x.shape[:-1]
        x = x.reshape(-1, x.shape[-1])                           # pack([x], '* n')
        x = curtail_to_multiple(x, self.seq_len_multiple_of, from_left=curtail_from_left)
        return x.unsqueeze(1), lead

    def encode(self, x):
        """(b, 1, n) fp32 -> (b, n / prod(strides), codebook_dim): the encoder stack + 'b c n -> b n c'."""
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd.SoundStream runs on the MI355X only (no CPU fallback)')
        h = x.to(F32).contiguous()
        for layer in self.encoder:
            if isinstance(layer, CausalConv1d):
                h = layer.run(h)
            else:
                for sub in layer:
                    h = sub(h) if isinstance(sub, _ResidualFn) else sub.run(h)
        if self.encoder_attn is not None:                        # :830-833 ('b c n -> b n c' first there; here the layout is kept)
            h = self.encoder_attn.run_bct(h)
        return ops.bct_to_btc(h)

    @torch.no_grad()
    def tokenize(self, audio):                                   # soundstream.py:797-800
        self.eval()
        return self.forward(audio, return_codes_only=True)

    @torch.no_grad()
    def forward(self, x, target=None, is_denoising=None, return_encoded=False, return_codes_only=False, return_discr_loss=False,
                return_discr_losses_separately=False, return_loss_breakdown=False, return_recons_only=False, input_sample_hz=None,
                apply_grad_penalty=False, curtail_from_left=False):
        """The eval-mode branches of soundstream.py:802-862 (same positional order): return_codes_only -> indices (g, b, n, q);
        return_encoded -> (quantized, indices 'b n (g q)', commit_loss); return_recons_only -> the reconstructed wave.  The training
        branches (discriminators, losses) are out of scope."""
        if target is not None or is_denoising is not None or return_discr_loss or return_discr_losses_separately or return_loss_breakdown \
                or apply_grad_penalty or not (return_encoded or return_codes_only or return_recons_only):
            raise NotImplementedError('only the eval-mode branches forward(..., return_codes_only=True | return_encoded=True | '
                                      'return_recons_only=True) are implemented (SoundStream training is out of scope)')
        x, lead = self.process_input(x, input_sample_hz=input_sample_hz, curtail_from_left=curtail_from_left)
        feats = self.encode(x)
        quantized, indices, commit_loss = self.rq(feats)
        if return_codes_only:
            return indices                                       # (g, b, n, q), soundstream.py:847-848
        b, n = indices.shape[1], indices.shape[2]
        if return_encoded:
            return quantized, indices.permute(1, 2, 0, 3).reshape(b, n, -1), commit_loss          # 'g b n q -> b n (g q)', :851
        recon = self.decode(quantized)                           # :857-866, unpack(recon_x, ps, '* c n')
        return recon.reshape(*lead, recon.shape[-2], recon.shape[-1])

    @torch.no_grad()
    def decode_from_codebook_indices(self, quantized_indices):               # soundstream.py:691-699
        assert quantized_indices.dtype in (torch.long, torch.int32)
        if quantized_indices.ndim == 3:
            b, n, gq = quantized_indices.shape
            quantized_indices = quantized_indices.reshape(b, n, self.rq_groups, gq // self.rq_groups).permute(2, 0, 1, 3)   # 'b n (g q) -> g b n q'
        return self.decode(self.rq.get_output_from_indices(quantized_indices))

    @torch.no_grad()
    def decode(self, x, quantize=False):                                      # soundstream.py:701-709
        """x fp32 (b, n, codebook_dim) -> wave (b, input_channels, n * prod(strides)): 'b n c -> b c n', then the causal transposed-conv decoder."""
        if quantize:
            x, *_ = self.rq(x)
        if not x.is_cuda:
            raise RuntimeError('audiolm_pytorch_amd.SoundStream runs on the MI355X only (no CPU fallback)')
        h = ops.bct_to_btc(x.to(F32).contiguous())                           # the same per-batch 2-D transpose, applied to (n, c) -> (c, n)
        if self.decoder_attn is not None:                                    # :705-706
            h = self.decoder_attn.run_bct(h)
        for layer in self.decoder:
            if isinstance(layer, CausalConv1d):
                h = layer.run(h)
            else:
                for sub in layer:
                    h = sub(h)
        return h
