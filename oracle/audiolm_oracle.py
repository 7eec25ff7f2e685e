"""TEST INFRASTRUCTURE ONLY (oracle).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file,
and only as the *checker* -- never as the thing that is measured or shipped.  The product
path (audiolm-pytorch_amd/) never imports anything under oracle/.

What this is: a CPU, fp32, plain-PyTorch *functional restatement* of the reference's training
hot path.  Every function takes the reference's `state_dict`
(same key names / shapes as /root/reference's modules) plus inputs and cites the reference
file:line it follows.  It is pinned against golden vectors produced by running the REAL
reference in the build container (tests/golden/make_golden.py -> tests/golden/*.pt;
tests/test_oracle_golden.py).

Parity status:
  * num_residual_streams == 1 : every arithmetic op is first-party reference code -> PINNED.
  * num_residual_streams  > 1 : hyper-connections are a third-party package whose source is not
    vendored (oracle/hyper_connections_restated.py) -> "parity unpinned / restated-oracle".
  * SoundStream RVQ: third-party vector-quantize-pytorch -> restated (oracle/rvq_restated.py),
    "parity unpinned"; the causal-conv encoder is first-party -> PINNED.

All file:line citations are relative to /root/reference/audiolm_pytorch/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence


# ----------------------------------------------------------------------------------------------
# integer / bookkeeping helpers (bit-exact)
# ----------------------------------------------------------------------------------------------

def ceil_div(numer, denom):                      # audiolm_pytorch.py:62-63
    return (numer + denom - 1) // denom


def round_down_nearest_multiple(val, mult):      # audiolm_pytorch.py:68-69
    return (val // mult) * mult


def append_eos_id(ids, eos_id):                  # audiolm_pytorch.py:155-160
    b = ids.shape[0]
    eos = torch.full((b, 1), eos_id, dtype=torch.long, device=ids.device)
    return torch.cat((ids, eos), dim=-1)


def batch_unique_consecutive(t, pad_value=0.):   # audiolm_pytorch.py:162-164
    unique_arr = [torch.unique_consecutive(el) for el in t.unbind(dim=0)]
    return pad_sequence(unique_arr, batch_first=True, padding_value=pad_value)


def generate_mask_with_prob(shape, mask_prob, device, generator=None):   # audiolm_pytorch.py:82-89
    seq = shape[-1]
    rand = torch.randn(shape, device=device, generator=generator)
    rand[:, 0] = -torch.finfo(rand.dtype).max
    num_mask = min(int(seq * mask_prob), seq - 1)
    indices = rand.topk(num_mask, dim=-1).indices
    return ~torch.zeros(shape, device=device).scatter(1, indices, 1.).bool()


def get_embeds(weight, codes, pad_id=-1):        # audiolm_pytorch.py:168-186
    pad_mask = codes == pad_id
    codes_without_pad = codes.masked_fill(pad_mask, 0)
    embeds = F.embedding(codes_without_pad, weight)
    return embeds.masked_fill(pad_mask.unsqueeze(-1), 0.)


# ----------------------------------------------------------------------------------------------
# A1/A2  Attend   (attend.py:98-146; the flash path attend.py:69-96 is the same maths)
# ----------------------------------------------------------------------------------------------

def attend(q, k, v, mask=None, attn_bias=None, causal=True, keep=None, p_drop=0.):
    """q (b h i d), k/v (b j d) single shared head (MQA).  attend.py:115-146.  keep / p_drop: the nn.Dropout on the attention probabilities
    (attend.py:140; the flash path's dropout_p, :92) with its Bernoulli keep mask SUPPLIED (0 / 1, (b h i j)): attn * keep / (1 - p) is what
    F.dropout computes once the mask is drawn."""
    scale = q.shape[-1] ** -0.5
    sim = torch.einsum('bhid,bjd->bhij', q, k) * scale
    if attn_bias is not None:
        sim = sim + attn_bias
    neg = -torch.finfo(sim.dtype).max
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None, None, :], neg)
    if causal:
        i, j = sim.shape[-2:]
        causal_mask = torch.ones((i, j), device=sim.device, dtype=torch.bool).triu(j - i + 1)
        sim = sim.masked_fill(causal_mask, neg)
    attn = sim.softmax(dim=-1)
    if keep is not None:
        attn = attn * keep / (1. - p_drop)
    return torch.einsum('bhij,bjd->bhid', attn, v)


# ----------------------------------------------------------------------------------------------
# A4 LayerNorm / GEGLU / FeedForward ; A3 Attention
# ----------------------------------------------------------------------------------------------

def layer_norm(x, gamma):                        # audiolm_pytorch.py:191-198 (beta is a zero buffer)
    return F.layer_norm(x, x.shape[-1:], gamma, torch.zeros_like(gamma))


def feedforward(sd, p, x, keep=None, p_drop=0.):  # audiolm_pytorch.py:246-260
    """keep / p_drop: the nn.Dropout(ff_dropout) of :258 with its Bernoulli keep mask SUPPLIED (0 / 1, shape of the inner activation): x * keep / (1 - p)
    is what F.dropout computes once the mask is drawn -- the parity test hands the masks of the device run to this restatement."""
    x = layer_norm(x, sd[p + '0.gamma'])
    x = F.linear(x, sd[p + '1.weight'])
    x, gate = x.chunk(2, dim=-1)                 # GEGLU: gate is the SECOND half (:246-249)
    x = F.gelu(gate) * x
    x = layer_norm(x, sd[p + '3.gamma'])
    if keep is not None:
        x = x * keep / (1. - p_drop)
    return F.linear(x, sd[p + '5.weight'])


def attention(sd, p, x, heads, mask=None, attn_bias=None, value_residual=None, context=None, prefix_context=None,
              prefix_context_mask=None, causal=True, attn_keep=None, out_keep=None, p_drop=0.):
    """Attention.forward, audiolm_pytorch.py:307-406 (no kv-cache).  Self-attention: context None.  Cross-attention (:450: dim_context,
    num_null_kv=1, norm_context=True, causal=False): `context` (b m dc), `mask` = context mask.  cond_as_self_attn_prefix: `prefix_context`
    (b m d) is prepended to the self-attention key / value inputs (:330-345).  Returns (out, orig_v).
    attn_keep (b h i j) / out_keep (b n dim) / p_drop: Attention(dropout = p) in training mode with both Bernoulli masks supplied -- the dropout on the
    attention probabilities (Attend, :299-303) and the nn.Dropout behind to_out (:304)."""
    b, n, _ = x.shape
    if context is not None and (p + 'context_norm.gamma') in sd:
        context = layer_norm(context, sd[p + 'context_norm.gamma'])      # :322-323
    kv_input = x if context is None else context                          # :325  -- bound BEFORE the pre-norm (reference quirk)
    if prefix_context is not None:                                        # :330-345
        kv_input = torch.cat((prefix_context, kv_input), dim=-2)
        m = prefix_context.shape[-2]
        if mask is None:
            mask = torch.ones((b, n), device=x.device, dtype=torch.bool)
        if prefix_context_mask is not None:
            mask = torch.cat((prefix_context_mask, mask), dim=-1)
        else:
            mask = F.pad(mask, (m, 0), value=True)
        if attn_bias is not None:
            attn_bias = F.pad(attn_bias, (m, 0), value=0.)
    xn = layer_norm(x, sd[p + 'norm.gamma'])      # :347
    q = F.linear(xn, sd[p + 'to_q.weight'])       # :351
    k, v = F.linear(kv_input, sd[p + 'to_kv.weight']).chunk(2, dim=-1)
    orig_v = v
    if value_residual is not None:                # :357-358
        v = 0.5 * (v + value_residual)
    if (p + 'null_kv') in sd:                     # :372-376 (after the value residual: the null value is never mixed)
        nk, nv = sd[p + 'null_kv'][0], sd[p + 'null_kv'][1]
        k = torch.cat((nk.expand(b, -1, -1), k), dim=-2)
        v = torch.cat((nv.expand(b, -1, -1), v), dim=-2)
        if mask is not None:                      # :384-385
            mask = F.pad(mask, (nk.shape[0], 0), value=True)
    q = q.reshape(b, n, heads, -1).transpose(1, 2)            # 'b n (h d) -> b h n d'
    out = attend(q, k, v, mask=mask, attn_bias=attn_bias, causal=causal, keep=attn_keep, p_drop=p_drop)
    out = out.transpose(1, 2).reshape(b, n, -1)               # 'b h n d -> b n (h d)'
    out = F.linear(out, sd[p + 'to_out.0.weight'])
    if out_keep is not None:
        out = out * out_keep / (1. - p_drop)
    return out, orig_v


# ----------------------------------------------------------------------------------------------
# A6 hyper-connections (restated third-party; see oracle/hyper_connections_restated.py)
# ----------------------------------------------------------------------------------------------

def hc_width(sd, p, R, S):
    """R ((b s) n d) -> branch_input (b n d), R' (b n s d), beta (b n s)."""
    bs, n, d = R.shape
    r = R.reshape(bs // S, S, n, d).permute(0, 2, 1, 3)      # b n s d
    normed = F.normalize(r, dim=-1) * (d ** 0.5) * (sd[p + 'norm.gamma'] + 1)
    alpha = torch.tanh(normed @ sd[p + 'dynamic_alpha_fn']) * sd[p + 'dynamic_alpha_scale'] + sd[p + 'static_alpha']
    beta = torch.tanh(normed @ sd[p + 'dynamic_beta_fn']) * sd[p + 'dynamic_beta_scale'] + sd[p + 'static_beta']
    mix = torch.einsum('bnst,bnsd->bntd', alpha, r)
    return mix[..., 0, :], mix[..., 1:, :], beta


def hc_depth(branch_out, Rp, beta):
    r = Rp + branch_out.unsqueeze(-2) * beta.unsqueeze(-1)   # b n s d
    b, n, s, d = r.shape
    return r.permute(0, 2, 1, 3).reshape(b * s, n, d)        # 'b n s d -> (b s) n d'


# ----------------------------------------------------------------------------------------------
# A7 relative position bias ; A5 Transformer
# ----------------------------------------------------------------------------------------------

def rel_pos_bias(sd, p, i, j):                    # audiolm_pytorch.py:202-242
    dev = sd[p + 'net.0.0.weight'].device
    i_pos = torch.arange(i, device=dev) + (j - i)
    j_pos = torch.arange(j, device=dev)
    rel_pos = i_pos[:, None] - j_pos[None, :] + (j - 1)
    x = torch.arange(-j + 1, j, device=dev).float()[:, None]
    li = 0
    while (p + f'net.{li}.0.weight') in sd:
        x = F.silu(F.linear(x, sd[p + f'net.{li}.0.weight'], sd[p + f'net.{li}.0.bias']))
        li += 1
    x = F.linear(x, sd[p + f'net.{li}.weight'], sd[p + f'net.{li}.bias'])
    return x[rel_pos].permute(2, 0, 1)            # 'i j h -> h i j'


def transformer(sd, p, x, *, depth, heads, streams=4, self_attn_mask=None, attn_bias=None,
                grad_shrink_alpha=0.1, add_value_residual=True, context=None, context_mask=None, cond_as_self_attn_prefix=False,
                ff_keep=None, ff_dropout=0., attn_keep=None, out_keep=None, attn_dropout=0.):
    """audiolm_pytorch.py:461-560 (no kv-cache).  `p` is e.g. 'transformer.'.  A conditioning `context` goes to the cross-attention layers
    (present in `sd` as layers.{l}.1.*) or, with cond_as_self_attn_prefix, in front of the self-attention keys."""
    n = x.shape[1]
    x = x * grad_shrink_alpha + x.detach() * (1 - grad_shrink_alpha)     # :93-94, :478
    if attn_bias is None and (p + 'rel_pos_bias.net.0.0.weight') in sd:  # :500-503
        attn_bias = rel_pos_bias(sd, p + 'rel_pos_bias.', n, n)
    self_kw = dict(prefix_context=context, prefix_context_mask=context_mask) if cond_as_self_attn_prefix else {}     # :510-515
    value_residual = cross_value_residual = None
    if streams > 1:
        x = x.repeat_interleave(streams, dim=0)                            # :524  'b ... -> (b s) ...'

    def branch(pp, fn):
        nonlocal x
        if streams > 1:
            bi, Rp, beta = hc_width(sd, pp, x, streams)
            out = fn(bi)
            out, extra = out if isinstance(out, tuple) else (out, None)
            x = hc_depth(out, Rp, beta)
        else:
            out = fn(x)
            out, extra = out if isinstance(out, tuple) else (out, None)
            x = out + x
        return extra

    for l in range(depth):
        pa, pc, pf = f'{p}layers.{l}.0.', f'{p}layers.{l}.1.', f'{p}layers.{l}.2.'
        akw = {} if attn_keep is None else dict(attn_keep=attn_keep[l], out_keep=out_keep[l], p_drop=attn_dropout)     # one mask pair per layer
        values = branch(pa, lambda t: attention(sd, pa + 'branch.', t, heads, mask=self_attn_mask, attn_bias=attn_bias,
                                                value_residual=value_residual, **self_kw, **akw))
        if add_value_residual and value_residual is None:                  # :534-535
            value_residual = values
        if (pc + 'branch.to_q.weight') in sd:                              # :539-544 cross attention
            assert context is not None
            cvalues = branch(pc, lambda t: attention(sd, pc + 'branch.', t, heads, mask=context_mask, context=context,
                                                     value_residual=cross_value_residual, causal=False))
            if add_value_residual and cross_value_residual is None:
                cross_value_residual = cvalues
        branch(pf, lambda t: feedforward(sd, pf + 'branch.', t, None if ff_keep is None else ff_keep[l], ff_dropout))   # ff_keep: one mask per layer
    if streams > 1:
        x = x.reshape(x.shape[0] // streams, streams, *x.shape[1:]).sum(dim=1)   # :551
    return layer_norm(x, sd[p + 'norm.gamma'])                             # :555


def condition(sd, cfg, b, text_embeds, cond_drop_keep=None, mask_from_embeds=True):
    """audiolm_pytorch.py:685-704 / :873-892 / :1150-1169 with pre-computed text embeddings: -> (context, context_mask).  `cond_drop_keep`:
    bool (b,) = the `prob_mask_like((b,), 1 - cond_drop_prob)` draw (None: cond_drop_prob == 0).  Semantic- and FineTransformer derive the text
    mask only when they ran the text encoder themselves (:692-695, :1156-1160): with pre-computed embeddings they have NO mask, so nothing is
    padded out and the condition is never dropped (mask_from_embeds=False); CoarseTransformer takes it from the embeddings (:882-883)."""
    if text_embeds is None:
        return None, None
    text_mask = torch.any(text_embeds != 0, dim=-1) if mask_from_embeds else None
    if 'proj_text_embed.weight' in sd:
        text_embeds = F.linear(text_embeds, sd['proj_text_embed.weight'])
    if text_mask is not None and cond_drop_keep is not None:
        text_mask = cond_drop_keep[:, None] & text_mask
    return text_embeds, text_mask


# ----------------------------------------------------------------------------------------------
# model configs
# ----------------------------------------------------------------------------------------------

@dataclass
class Cfg:
    dim: int
    depth: int
    heads: int = 8
    streams: int = 4
    num_semantic_tokens: int = 0
    codebook_size: int = 0
    num_coarse_quantizers: int = 0
    num_fine_quantizers: int = 0
    grad_shrink_alpha: float = 0.1
    add_value_residual: bool = True
    pad_id: int = -1
    cond_as_self_attn_prefix: bool = False


# ----------------------------------------------------------------------------------------------
# A8 SemanticTransformer.forward  (audiolm_pytorch.py:671-724)
# ----------------------------------------------------------------------------------------------

def semantic_forward(sd, cfg: Cfg, ids, self_attn_mask=None, text_embeds=None, cond_drop_keep=None):
    tokens = get_embeds(sd['semantic_embedding.weight'], ids)                    # :709
    b = ids.shape[0]
    start = sd['start_token'].expand(b, 1, -1)                                  # :711
    tokens = torch.cat((start, tokens), dim=1)
    if self_attn_mask is not None:
        self_attn_mask = F.pad(self_attn_mask, (1, 0), value=True)              # :716
    context, context_mask = condition(sd, cfg, b, text_embeds, cond_drop_keep, mask_from_embeds=False)
    tokens = transformer(sd, 'transformer.', tokens, depth=cfg.depth, heads=cfg.heads, streams=cfg.streams,
                         self_attn_mask=self_attn_mask, grad_shrink_alpha=cfg.grad_shrink_alpha,
                         add_value_residual=cfg.add_value_residual, context=context, context_mask=context_mask,
                         cond_as_self_attn_prefix=cfg.cond_as_self_attn_prefix)
    return head_linear(tokens, sd['to_logits.weight'], sd['to_logits.bias'])     # :719


# ----------------------------------------------------------------------------------------------
# A9 CoarseTransformer.forward  (audiolm_pytorch.py:858-990)
# ----------------------------------------------------------------------------------------------

def head_linear(x, w, b=None):
    """the nn.Linear logit heads (:621 / :798).  A module-level name on purpose: oracle/rounding_matched.py swaps the three head functions (this one,
    _grouped_logits, _padded_logits) for split-bf16 restatements of the same contractions."""
    return F.linear(x, w, b)


def _padded_logits(weights, pred, Q):
    """Fine-coarse logits: zero-pad to a multiple of Q, group, slice (:1325-1339)"""
    b, n = pred.shape[:2]
    padding = ceil_div(n, Q) * Q - n
    pc = F.pad(pred, (0, 0, 0, padding), value=0.) if padding else pred
    pc = pc.reshape(b, -1, Q, pc.shape[-1])
    return torch.einsum('qcd,bnqd->bnqc', weights, pc).reshape(b, -1, weights.shape[1])[:, :n]


def _grouped_logits(weights, pred, Q):
    """groupable part + remainder with W[:r]  (Coarse :965-983, Fine-fine :1343-1361)."""
    n = pred.shape[1]
    nq = round_down_nearest_multiple(n, Q)
    g, r = pred[:, :nq], pred[:, nq:]
    g = g.reshape(g.shape[0], nq // Q, Q, pred.shape[-1])                        # explicit width: nq may be 0 (first sampling step)
    lg = torch.einsum('qcd,bnqd->bnqc', weights, g).reshape(g.shape[0], nq, weights.shape[1])
    if r.shape[1] > 0:
        lr = torch.einsum('qcd,bqd->bqc', weights[:r.shape[1]], r)
        lg = torch.cat((lg, lr), dim=1)
    return lg


def coarse_forward(sd, cfg: Cfg, semantic_token_ids, coarse_token_ids, self_attn_mask=None, text_embeds=None, cond_drop_keep=None):
    b = semantic_token_ids.shape[0]
    context, context_mask = condition(sd, cfg, b, text_embeds, cond_drop_keep)
    dev = semantic_token_ids.device
    Q, C = cfg.num_coarse_quantizers, cfg.codebook_size
    coarse_token_ids = coarse_token_ids.reshape(b, -1)
    semantic_token_ids = semantic_token_ids.reshape(b, -1)                       # :894
    nc = coarse_token_ids.shape[-1]
    offsets = (C * torch.arange(Q, device=dev)).repeat(ceil_div(nc, Q))[:nc]   # :896-898 (stride C, table Q*(C+1))
    coarse_token_ids = coarse_token_ids + offsets[None]
    semantic_tokens = get_embeds(sd['semantic_embedding.weight'], semantic_token_ids)   # :901
    coarse_tokens = F.embedding(coarse_token_ids, sd['coarse_embedding.weight'])       # :902
    qpos = sd['coarse_quantize_embedding.weight'].repeat(ceil_div(nc, Q), 1)[:nc]      # :904-906
    coarse_tokens = coarse_tokens + qpos
    semantic_seq_len = semantic_tokens.shape[1]
    tokens = torch.cat((sd['semantic_start_token'].expand(b, 1, -1), semantic_tokens,
                        sd['coarse_start_token'].expand(b, 1, -1), coarse_tokens), dim=1)   # :913-918
    seq_len = tokens.shape[-2]
    attn_bias = None
    if 'transformer.rel_pos_bias.net.0.0.weight' in sd:                          # :926-936
        attn_bias = rel_pos_bias(sd, 'transformer.rel_pos_bias.', seq_len, seq_len)
        is_sem = torch.arange(seq_len, device=dev) < (semantic_seq_len + 1)
        is_cross = is_sem[:, None] ^ is_sem[None, :]
        attn_bias = torch.where(is_cross, sd['cross_attn_bias'], attn_bias)
    tokens = transformer(sd, 'transformer.', tokens, depth=cfg.depth, heads=cfg.heads, streams=cfg.streams,
                         self_attn_mask=self_attn_mask, attn_bias=attn_bias,
                         grad_shrink_alpha=cfg.grad_shrink_alpha, add_value_residual=cfg.add_value_residual,
                         context=context, context_mask=context_mask, cond_as_self_attn_prefix=cfg.cond_as_self_attn_prefix)
    pred_sem, pred_coarse = tokens[:, :semantic_seq_len], tokens[:, semantic_seq_len + 1:]   # :957
    semantic_logits = None
    if 'to_semantic_logits.weight' in sd:                                        # :961
        semantic_logits = head_linear(pred_sem, sd['to_semantic_logits.weight'], sd['to_semantic_logits.bias'])
    coarse_logits = _grouped_logits(sd['coarse_logit_weights'], pred_coarse, Q)  # :965-983
    return semantic_logits, coarse_logits


# ----------------------------------------------------------------------------------------------
# A10 FineTransformer.forward  (audiolm_pytorch.py:1136-1368)
# ----------------------------------------------------------------------------------------------

def fine_attn_bias(sd, cfg: Cfg, coarse_length, fine_length, dev):
    """audiolm_pytorch.py:1229-1298."""
    Qc, Qf = cfg.num_coarse_quantizers, cfg.num_fine_quantizers
    coarse_seq_length, fine_seq_length = ceil_div(coarse_length, Qc), ceil_div(fine_length, Qf)
    coarse_offsets = torch.arange(Qc, device=dev).repeat(coarse_seq_length)[:coarse_length]
    fine_offsets = torch.arange(Qf, device=dev).repeat(fine_seq_length)[:fine_length]
    max_seq_len = max(coarse_seq_length, fine_seq_length)
    coarse_pos = torch.arange(coarse_seq_length, device=dev).repeat_interleave(Qc)[:coarse_length]
    fine_pos = torch.arange(fine_seq_length, device=dev).repeat_interleave(Qf)[:fine_length]
    coarse_pos = F.pad(coarse_pos, (1, 0), value=-1)
    fine_pos = F.pad(fine_pos, (1, 0), value=-1)
    seq_positions = torch.cat((coarse_pos, fine_pos), dim=-1)
    coarse_offsets = F.pad(coarse_offsets, (1, 0), value=0)
    fine_offsets = F.pad(fine_offsets + Qc, (1, 0), value=0)
    seq_offsets = torch.cat((coarse_offsets, fine_offsets), dim=-1)
    pos_mlp_input = torch.stack((seq_positions.clamp(min=0), seq_offsets), dim=-1)
    num_offsets = Qf + Qc
    rel_seq_len, rel_offsets = 2 * max_seq_len - 1, 2 * num_offsets - 1
    rel_dist = pos_mlp_input[:, None, :] - pos_mlp_input[None, :, :]
    rel_seq_len_range = torch.arange(rel_seq_len, device=dev).repeat_interleave(rel_offsets)
    rel_offset_range = torch.arange(rel_offsets, device=dev).repeat(rel_seq_len)
    mlp_inputs = torch.stack((rel_seq_len_range, rel_offset_range), dim=-1).float()
    h = F.silu(F.linear(mlp_inputs, sd['pos_bias_mlp.0.weight'], sd['pos_bias_mlp.0.bias']))
    h = F.silu(F.linear(h, sd['pos_bias_mlp.2.weight'], sd['pos_bias_mlp.2.bias']))
    table = F.linear(h, sd['pos_bias_mlp.4.weight'], sd['pos_bias_mlp.4.bias'])
    idx = (rel_dist[..., 0] + max_seq_len - 1) * rel_offsets + (rel_dist[..., 1] + num_offsets - 1)
    bias = table[idx].permute(2, 0, 1)
    is_start = seq_positions == -1
    start_mask = is_start[:, None] | is_start[None, :]
    return torch.where(start_mask, sd['null_pos_bias'], bias)


def fine_forward(sd, cfg: Cfg, coarse_token_ids, fine_token_ids, self_attn_mask=None, text_embeds=None, cond_drop_keep=None):
    b = coarse_token_ids.shape[0]
    context, context_mask = condition(sd, cfg, b, text_embeds, cond_drop_keep, mask_from_embeds=False)       # :1155-1169: like Semantic, no mask from pre-computed embeds
    dev = coarse_token_ids.device
    Qc, Qf, C = cfg.num_coarse_quantizers, cfg.num_fine_quantizers, cfg.codebook_size
    eos_id = C                                                                    # :1040
    coarse_token_ids = coarse_token_ids.reshape(b, -1)
    fine_token_ids = fine_token_ids.reshape(b, -1)                               # :1171
    coarse_mask = (coarse_token_ids != cfg.pad_id) & (coarse_token_ids != eos_id)   # :1175
    coarse_token_ids = coarse_token_ids.masked_fill(~coarse_mask, 0)
    nf = fine_token_ids.shape[-1]
    coarse_mask = F.pad(coarse_mask, (1, nf + 1), value=True)                    # :1179
    self_attn_mask = coarse_mask if self_attn_mask is None else (self_attn_mask & coarse_mask)   # :1181-1184
    n = coarse_token_ids.shape[-1]
    coarse_offsets = torch.arange(Qc, device=dev).repeat(ceil_div(n, Qc))[:n]
    fine_offsets = torch.arange(Qf, device=dev).repeat(ceil_div(nf, Qf))[:nf]
    coarse_token_ids = coarse_token_ids + coarse_offsets[None] * C               # :1195
    fine_token_ids = fine_token_ids + fine_offsets[None] * C                     # :1202
    coarse_tokens = F.embedding(coarse_token_ids, sd['coarse_embedding.weight'])
    fine_tokens = F.embedding(fine_token_ids, sd['fine_embedding.weight'])
    coarse_tokens = coarse_tokens + sd['coarse_quantize_embedding.weight'].repeat(ceil_div(n, Qc), 1)[:n]
    fine_tokens = fine_tokens + sd['fine_quantize_embedding.weight'].repeat(ceil_div(nf, Qf), 1)[:nf]
    tokens = torch.cat((sd['coarse_start_token'].expand(b, 1, -1), coarse_tokens,
                        sd['fine_start_token'].expand(b, 1, -1), fine_tokens), dim=1)     # :1218-1223
    attn_bias = None
    if 'pos_bias_mlp.0.weight' in sd:
        attn_bias = fine_attn_bias(sd, cfg, n, nf, dev)
    tokens = transformer(sd, 'transformer.', tokens, depth=cfg.depth, heads=cfg.heads, streams=cfg.streams,
                         self_attn_mask=self_attn_mask, attn_bias=attn_bias,
                         grad_shrink_alpha=cfg.grad_shrink_alpha, add_value_residual=cfg.add_value_residual,
                         context=context, context_mask=context_mask, cond_as_self_attn_prefix=cfg.cond_as_self_attn_prefix)
    pred_coarse, pred_fine = tokens[:, :n], tokens[:, n + 1:]                    # :1319
    coarse_logits = None
    if 'coarse_logit_weights' in sd:                                             # :1325-1339 (zero-pad then slice)
        coarse_logits = _padded_logits(sd['coarse_logit_weights'], pred_coarse, Qc)
    fine_logits = _grouped_logits(sd['fine_logit_weights'], pred_fine, Qf)       # :1343-1361
    return coarse_logits, fine_logits


# ----------------------------------------------------------------------------------------------
# A11-A13 training wrappers: bookkeeping + loss
# `forgetful_mask`: the reference draws it from the device RNG (generate_mask_with_prob); parity
# runs inject it explicitly (None == mask_prob 0 / eval).
# ----------------------------------------------------------------------------------------------

def semantic_wrapper_bookkeeping(semantic_token_ids, eos_id, *, training=True, unique_consecutive=True, pad_id=-1):
    """audiolm_pytorch.py:1534-1544 -> (input_ids, labels)."""
    ids = semantic_token_ids.reshape(semantic_token_ids.shape[0], -1)
    if training:
        ids = append_eos_id(ids, eos_id)
    if unique_consecutive:
        ids = batch_unique_consecutive(ids, pad_value=pad_id)
    return ids[:, :-1], ids


def semantic_wrapper_loss(sd, cfg: Cfg, semantic_token_ids, *, training=True, unique_consecutive=True,
                          forgetful_mask=None, text_embeds=None, cond_drop_keep=None):
    """audiolm_pytorch.py:1513-1567 with return_loss=True."""
    input_ids, labels = semantic_wrapper_bookkeeping(semantic_token_ids, cfg.num_semantic_tokens, training=training,
                                                     unique_consecutive=unique_consecutive, pad_id=cfg.pad_id)
    logits = semantic_forward(sd, cfg, input_ids, self_attn_mask=forgetful_mask, text_embeds=text_embeds, cond_drop_keep=cond_drop_keep)
    return F.cross_entropy(logits.transpose(1, 2), labels, ignore_index=cfg.pad_id)


def coarse_wrapper_bookkeeping(semantic_token_ids, coarse_token_ids, semantic_eos_id, coarse_eos_id, *,
                               training=True, unique_consecutive=True, pad_id=-1):
    """audiolm_pytorch.py:1785-1805 -> (semantic_in, coarse_in, semantic_labels, coarse_labels, key_mask)."""
    b = semantic_token_ids.shape[0]
    sem = semantic_token_ids.reshape(b, -1)
    coarse = coarse_token_ids.reshape(b, -1)
    if training:
        sem = append_eos_id(sem, semantic_eos_id)
        coarse = append_eos_id(coarse, coarse_eos_id)
    if unique_consecutive:
        sem = batch_unique_consecutive(sem, pad_value=pad_id)
    sem_labels, coarse_labels = sem, coarse.clone()
    coarse_in = coarse[:, :-1]
    mask = (sem != pad_id) & (sem != semantic_eos_id)
    sem_in = sem.masked_fill(~mask, 0)
    mask = F.pad(mask, (1, coarse_in.shape[-1] + 1), value=True)
    return sem_in, coarse_in, sem_labels, coarse_labels, mask


def coarse_wrapper_loss(sd, cfg: Cfg, semantic_token_ids, coarse_token_ids, *, training=True, unique_consecutive=True,
                        forgetful_mask=None, semantic_ce_weight=1., text_embeds=None, cond_drop_keep=None):
    """audiolm_pytorch.py:1742-1854 with return_loss=True (ids supplied, no codec/wav2vec)."""
    sem_in, coarse_in, sem_labels, coarse_labels, mask = coarse_wrapper_bookkeeping(
        semantic_token_ids, coarse_token_ids, cfg.num_semantic_tokens, cfg.codebook_size,
        training=training, unique_consecutive=unique_consecutive, pad_id=cfg.pad_id)
    if forgetful_mask is not None:
        mask = mask & forgetful_mask                                             # :1809-1810
    sem_logits, coarse_logits = coarse_forward(sd, cfg, sem_in, coarse_in, self_attn_mask=mask, text_embeds=text_embeds, cond_drop_keep=cond_drop_keep)
    if unique_consecutive:                                                       # :1828-1831
        num_coarse, num_sem = coarse_labels.numel(), (sem_labels != cfg.pad_id).sum()
    else:
        num_coarse, num_sem = coarse_logits.shape[1], sem_logits.shape[1]
    sem_loss, n_sem = 0., 0
    if semantic_ce_weight > 0 and sem_logits is not None:
        n_sem = num_sem
        sem_loss = F.cross_entropy(sem_logits.transpose(1, 2), sem_labels, ignore_index=cfg.pad_id)
    coarse_loss = F.cross_entropy(coarse_logits.transpose(1, 2), coarse_labels, ignore_index=cfg.pad_id)
    return (sem_loss * n_sem * semantic_ce_weight + coarse_loss * num_coarse) / (n_sem + num_coarse)


def fine_wrapper_loss(sd, cfg: Cfg, coarse_token_ids, fine_token_ids, *, forgetful_mask=None, coarse_ce_weight=1., text_embeds=None,
                      cond_drop_keep=None):
    """audiolm_pytorch.py:2041-2137 with return_loss=True (ids supplied)."""
    b = coarse_token_ids.shape[0]
    coarse = coarse_token_ids.reshape(b, -1)
    fine = fine_token_ids.reshape(b, -1)
    coarse_labels, fine_labels = coarse, fine
    fine_in = fine[:, :-1]
    mask = None if forgetful_mask is None else forgetful_mask.clone()            # (b, nc + nf_in + 2)  :2090-2096
    coarse_logits, fine_logits = fine_forward(sd, cfg, coarse, fine_in, self_attn_mask=mask, text_embeds=text_embeds, cond_drop_keep=cond_drop_keep)
    n_fine = fine_logits.shape[1]
    n_coarse, coarse_loss = 0, 0.
    if coarse_ce_weight > 0 and coarse_logits is not None:
        n_coarse = coarse_logits.shape[1]
        coarse_loss = F.cross_entropy(coarse_logits.transpose(1, 2), coarse_labels, ignore_index=cfg.pad_id)
    fine_loss = F.cross_entropy(fine_logits.transpose(1, 2), fine_labels, ignore_index=cfg.pad_id)
    return (coarse_loss * n_coarse * coarse_ce_weight + fine_loss * n_fine) / (n_coarse + n_fine)


# ----------------------------------------------------------------------------------------------
# A15-A17 SoundStream encode  (soundstream.py:332-380, 519-531, 779-852)
# ----------------------------------------------------------------------------------------------

def causal_conv1d(x, w, b, *, dilation=1, stride=1):        # soundstream.py:332-345 (reflect left pad)
    k = w.shape[-1]
    pad = dilation * (k - 1) + (1 - stride)
    if pad > 0:
        x = F.pad(x, (pad, 0), mode='reflect')
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def residual_unit(sd, p, x, dilation):                      # soundstream.py:362-369
    h = causal_conv1d(x, sd[p + 'fn.0.conv.weight'], sd[p + 'fn.0.conv.bias'], dilation=dilation)
    h = F.elu(h)
    h = causal_conv1d(h, sd[p + 'fn.2.conv.weight'], sd[p + 'fn.2.conv.bias'])
    h = F.elu(h)
    return h + x


def soundstream_encoder(sd, x, strides=(2, 4, 5, 8), dilations=(1, 3, 9), p='encoder.'):
    """soundstream.py:519-531: conv(k7) -> 4 x [3 x ResidualUnit, strided conv(k=2s)] -> conv(k3).  x (b 1 n)."""
    x = causal_conv1d(x, sd[p + '0.conv.weight'], sd[p + '0.conv.bias'])
    for bi, s in enumerate(strides):
        bp = f'{p}{bi + 1}.'
        for ri, d in enumerate(dilations):
            x = residual_unit(sd, f'{bp}{ri}.', x, d)
        x = causal_conv1d(x, sd[bp + '3.conv.weight'], sd[bp + '3.conv.bias'], stride=s)
    last = len(strides) + 1
    return causal_conv1d(x, sd[f'{p}{last}.conv.weight'], sd[f'{p}{last}.conv.bias'])


def causal_conv_transpose1d(x, w, b, stride):               # soundstream.py:347-360: ConvTranspose1d(k, stride), output cut to n * stride
    n = x.shape[-1]
    return F.conv_transpose1d(x, w, b, stride=stride)[..., :n * stride]


def soundstream_decoder(sd, x, strides=(2, 4, 5, 8), dilations=(1, 3, 9), p='decoder.'):
    """soundstream.py:615-627: conv(k7) -> for the strides in REVERSE order [transposed conv(k = 2s, stride s), 3 x ResidualUnit] -> conv(k7).
    x (b c n)."""
    x = causal_conv1d(x, sd[p + '0.conv.weight'], sd[p + '0.conv.bias'])
    for bi, s in enumerate(reversed(strides)):
        bp = f'{p}{bi + 1}.'
        x = causal_conv_transpose1d(x, sd[bp + '0.conv.weight'], sd[bp + '0.conv.bias'], s)
        for ri, d in enumerate(dilations):
            x = residual_unit(sd, f'{bp}{ri + 1}.', x, d)
    last = len(strides) + 1
    return causal_conv1d(x, sd[f'{p}{last}.conv.weight'], sd[f'{p}{last}.conv.bias'])


def rvq_decode(indices, codebooks):
    """(b, n, q) int -> (b, n, d): sum of the selected code vectors, -1 selects nothing (get_output_from_indices, soundstream.py:697)."""
    out = 0.
    for q, E in enumerate(codebooks):
        idx = indices[..., q]
        out = out + E[idx.clamp(min=0)].masked_fill((idx < 0).unsqueeze(-1), 0.)
    return out


def soundstream_decode_from_indices(sd, indices, *, strides=(2, 4, 5, 8), num_quantizers=8, groups=1):
    """SoundStream.decode_from_codebook_indices (soundstream.py:691-709, use_local_attn=False): indices (b, n, (g q)) -> wave (b, 1, n * prod)."""
    b, n, _ = indices.shape
    ix = indices.reshape(b, n, groups, num_quantizers).permute(2, 0, 1, 3)          # 'b n (g q) -> g b n q'
    outs = []
    for g in range(groups):
        cbs = [sd[f'rq.rvqs.{g}.layers.{q}._codebook.embed'][0] for q in range(num_quantizers)]
        outs.append(rvq_decode(ix[g], cbs))
    x = torch.cat(outs, dim=-1).transpose(1, 2)                                       # 'b n c -> b c n'
    return soundstream_decoder(sd, x, strides=strides)


def rvq_encode(x, codebooks):
    """x (b n d), codebooks (Q, C, d) -> indices (b n Q) int64.  See oracle/rvq_restated.py."""
    shape = x.shape
    residual = x.reshape(-1, shape[-1]).float()
    inds = []
    for E in codebooks:
        E = E.float()
        d2 = (residual ** 2).sum(-1, keepdim=True) + (E ** 2).sum(-1)[None] - 2 * residual @ E.t()
        dist = -d2.clamp(min=0).sqrt()
        idx = dist.argmax(dim=-1)
        residual = residual - E[idx]
        inds.append(idx)
    return torch.stack(inds, dim=-1).reshape(*shape[:-1], len(codebooks))


def soundstream_tokenize(sd, wave, *, strides=(2, 4, 5, 8), num_quantizers=8, groups=1):
    """SoundStream.forward(return_encoded=True) indices, soundstream.py:802-852 with use_local_attn=False:
    wave (b n) -> indices (b T (g q))."""
    mult = math.prod(strides)
    n = (wave.shape[-1] // mult) * mult                    # curtail_to_multiple, :789
    x = wave[..., :n].unsqueeze(1)
    x = soundstream_encoder(sd, x, strides=strides).transpose(1, 2)     # 'b c n -> b n c'
    outs = []
    for g, xg in enumerate(x.chunk(groups, dim=-1)):
        cbs = [sd[f'rq.rvqs.{g}.layers.{q}._codebook.embed'][0] for q in range(num_quantizers)]
        outs.append(rvq_encode(xg, cbs))
    return torch.cat(outs, dim=-1)                         # 'g b n q -> b n (g q)'
