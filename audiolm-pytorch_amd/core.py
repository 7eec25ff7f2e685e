"""Host-side driver of the transformer stack: the depth loop of reference audiolm_pytorch.py:461-560 (training path) written
as ONE autograd node whose forward and backward are explicit sequences of HIP launches (ops.py -> C ABI).  There is no
per-op autograd graph, no (b, h, n, n) tensors, no kv-cache stacking (audiolm_pytorch.py:532, :560 are semantics-free in
training -- SURVEY.md Appendix A.7).

Data layout in HBM (B sequences, N tokens, M = B*N rows, D model width, S residual streams):
  residual streams  R    fp32 | bf16 [B][S][N][D]     (reference '(b s) n d'); one tensor per branch boundary is kept for backward.  bf16
                                                      (StackCfg.residual_bf16) is what trainer.py:1241's autocast gives the reference: its streams
                                                      are bf16 from the first width connection on; arithmetic stays fp32 in registers
  branch input      X/XN bf16 [M][D]                  (XN = pre-LayerNorm output, X = un-normalised input feeding to_kv -- quirk A3)
  q / kv / attn out      bf16 [M][H*64] / [M][128] / [M][H*64]
  FFN               U    bf16 [M][2*Ipad] (x | gate halves, Ipad = inner rounded up to 8), HN bf16 [M][Ipad]
  weights                fp32 masters (nn.Parameter) + bf16 packed copies W and W^T refreshed when the master's version changes
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass

import torch

from . import launchlist, ops, relpos, xattn

BF16, F32 = torch.bfloat16, torch.float32


@dataclass
class StackCfg:
    dim: int
    depth: int
    heads: int
    dim_head: int
    streams: int
    inner: int                    # FFN inner width int(dim * 8 / 3)
    add_value_residual: bool
    grad_shrink_alpha: float
    residual_bf16: bool = False   # storage type of the 4-stream residual tensors and their gradients (num_residual_streams > 1 only)
    cross_attend: bool = False    # a cross-attention branch between attention and feed-forward (reference Transformer(cross_attend=True), :450)
    prefix: bool = False          # cond_as_self_attn_prefix: the conditioning context is prepended to the self-attention keys (:330-345, :510-515)
    dim_context: int = 0

    @property
    def inner_pad(self):
        return (self.inner + 7) // 8 * 8


HC_KEYS = ('Bb', 'Aa', 'Wa', 'sa', 'wb', 'sb', 'gamma')   # static_beta, static_alpha, dynamic_alpha_fn, dynamic_alpha_scale, dynamic_beta_fn, dynamic_beta_scale, norm.gamma
BRANCH_KEYS = dict(attn=('ln', 'wq', 'wkv', 'wo'), cross=('ln', 'ctx_ln', 'null_kv', 'wq', 'wkv', 'wo'), ff=('ln', 'w1', 'ln3', 'w2'))


def branch_kinds(cfg_or_cross):
    cross = cfg_or_cross if isinstance(cfg_or_cross, bool) else cfg_or_cross.cross_attend
    return ('attn', 'cross', 'ff') if cross else ('attn', 'ff')


def params_per_layer(S, cross=False):
    hc = len(HC_KEYS) if S > 1 else 0
    return sum(hc + len(BRANCH_KEYS[k]) for k in branch_kinds(bool(cross)))


def _split_layer(flat, S, cross=False):
    """flat per-layer parameter list -> [(kind, params dict, first index within the layer)] in execution order (see Transformer.flat_params)."""
    i, out = 0, []
    for kind in branch_kinds(bool(cross)):
        d, first = {}, i
        if S > 1:
            d['hc'] = dict(zip(HC_KEYS, flat[i:i + 7])); i += 7
        keys = BRANCH_KEYS[kind]
        d.update(zip(keys, flat[i:i + len(keys)])); i += len(keys)
        out.append((kind, d, first))
    return out


def tensor_version(t):
    """in-place-update counter used as the cache key of packed weights; inference tensors (created under torch.inference_mode) have none
    and cannot be updated in place: constant 0"""
    try:
        return t._version
    except RuntimeError:
        return 0


# ---- optimiser <-> packed weights (round 4).  FusedAdam (optimizer.py) can write the bf16 images of a dense weight while it updates the fp32 master
# (alm_opt_adam_pack_step): layer_weights() registers, per master weight, where its current images live; the optimiser asks pack_target(p) and, after
# the fused update, calls stamp_packed() so that the next forward finds the cache entry current and skips the re-pack.  Keyed by the master's data
# pointer and validated by its (pointer, version, shape) stamp: an entry whose stamp no longer matches the tensor (any other in-place update, a cleared
# or re-packed cache, another tensor at a recycled address) is simply not used.
FUSED_ADAM_PACK = os.environ.get('ALM_FUSED_ADAM_PACK', '1') != '0'
_PACK_REGISTRY = {}      # data_ptr -> dict(stamp, cache (weakref), key, out (the cache entry's packed pair), jobs [(row0, rows, cols, dst, dstT, rows_pad, cols_pad)])


_PACK_CACHES_SEEN = set()          # id(cache) of every WeightCache that has a finaliser installed
_PACK_GEN = [0]                    # bumped whenever an entry is (re)registered or dropped: optimizer.FusedAdam keys its prepared tables on it


def _drop_cache_entries(cid):
    _PACK_CACHES_SEEN.discard(cid)
    for k in [k for k, e in _PACK_REGISTRY.items() if e['cid'] == cid]:
        del _PACK_REGISTRY[k]
    _PACK_GEN[0] += 1


def _register_pack(w, stamp, cache, key, out, jobs):
    """An entry holds strong references to the packed images (`out` and the dst / dstT views in `jobs`: ~4 bytes per dense parameter) -- the same tensors
    the cache's own store holds.  They must not outlive the cache: a finaliser on the WeightCache removes its entries the moment the model that owns it is
    collected (until round 4 they were pruned only above 4096 keys, so rebuilding a dim-1024 model in a notebook or a test session pinned
    several GB of dead images in HBM)."""
    import weakref
    cid = id(cache)
    if cid not in _PACK_CACHES_SEEN:
        _PACK_CACHES_SEEN.add(cid)
        weakref.finalize(cache, _drop_cache_entries, cid)
    _PACK_REGISTRY[w.data_ptr()] = dict(stamp=stamp, cache=weakref.ref(cache), cid=cid, key=key, out=out, jobs=jobs)
    _PACK_GEN[0] += 1


def pack_target(p):
    """-> the registry entry of master weight `p` when its packed images are CURRENT (then updating p and the images together keeps them so), else None"""
    if not FUSED_ADAM_PACK:
        return None
    e = _PACK_REGISTRY.get(p.data_ptr())
    if e is None or e['stamp'] != (p.data_ptr(), tensor_version(p), tuple(p.shape)):
        return None
    cache = e['cache']()
    hit = cache.store.get(e['key']) if cache is not None else None
    if hit is None or hit[1] is not e['out'] or hit[0] != e['stamp']:
        return None
    return e


def stamp_packed(p, e):
    """after alm_opt_adam_pack_step wrote p and its images (and p's version counter advanced): the cache entry now describes the NEW value"""
    stamp = (p.data_ptr(), tensor_version(p), tuple(p.shape))
    e['stamp'] = stamp
    cache = e['cache']()
    if cache is not None:
        cache.store[e['key']] = (stamp, e['out'])


class WeightCache:
    """bf16 packed copies (W and W^T, zero padded) of the fp32 master weights; refreshed when a master's version changes
    (i.e. once per optimiser step) -- this is what `accelerator.autocast()` (trainer.py:1241) does per call, amortised."""

    def __init__(self):
        self.store = {}

    def get(self, key, w, builder):
        ver = (w.data_ptr(), tensor_version(w), tuple(w.shape))
        hit = self.store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            packed = builder(w.detach())
        self.store[key] = (ver, packed)
        return packed


def _pack_plain(w):
    rows, cols = w.shape
    rp, cp = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
    W = torch.empty((rp, cp), dtype=BF16, device=w.device)
    WT = torch.empty((cp, rp), dtype=BF16, device=w.device)
    ops.pack_weight(w, W, WT, rows_pad=rp, cols_pad=cp)
    return W[:rows], WT[:cols]


def _pack_w1(w, I, Ip):
    D = w.shape[1]
    W = torch.empty((2 * Ip, D), dtype=BF16, device=w.device)
    WT = torch.empty((D, 2 * Ip), dtype=BF16, device=w.device)
    ops.pack_weight(w[:I], W[:Ip], WT[:, :Ip], rows_pad=Ip, cols_pad=D)
    ops.pack_weight(w[I:], W[Ip:], WT[:, Ip:], rows_pad=Ip, cols_pad=D)
    return W, WT


def _pack_w2(w, I, Ip):
    D = w.shape[0]
    W = torch.empty((D, Ip), dtype=BF16, device=w.device)
    WT = torch.empty((Ip, D), dtype=BF16, device=w.device)
    ops.pack_weight(w, W, WT, rows_pad=D, cols_pad=Ip)
    return W, WT


def _layer_pack_plan(cache: WeightCache, l, branches, I, Ip):
    """-> None when every packed weight of layer `l` is current, else (ws, vers, out, jobs, targets): the fresh (uninitialised) packed buffers of ALL of
    the layer's dense weights and the pack jobs that fill them"""
    ws = []
    for kind, d, _ in branches:
        for name in ('wq', 'wkv', 'wo', 'w1', 'w2'):
            if name in d:
                ws.append(((kind, name), d[name]))
    vers = {k: (w.data_ptr(), tensor_version(w), tuple(w.shape)) for k, w in ws}
    hits = {k: cache.store.get((l,) + k) for k, _ in ws}
    if all(h is not None and h[0] == vers[k] for k, h in hits.items()):
        return None
    out, jobs, targets = {}, [], {}
    for k, w in ws:
        w = w.detach()
        if k[1] == 'w1':
            D = w.shape[1]
            W1 = torch.empty((2 * Ip, D), dtype=BF16, device=w.device)
            W1T = torch.empty((D, 2 * Ip), dtype=BF16, device=w.device)
            jobs.append((w[:I], W1[:Ip], W1T[:, :Ip], Ip, D))
            jobs.append((w[I:], W1[Ip:], W1T[:, Ip:], Ip, D))
            out[k] = (W1, W1T)
            targets[k] = [(0, I, D, W1[:Ip], W1T[:, :Ip], Ip, D), (I, w.shape[0] - I, D, W1[Ip:], W1T[:, Ip:], Ip, D)]
        elif k[1] == 'w2':
            D = w.shape[0]
            W2 = torch.empty((D, Ip), dtype=BF16, device=w.device)
            W2T = torch.empty((Ip, D), dtype=BF16, device=w.device)
            jobs.append((w, W2, W2T, D, Ip))
            out[k] = (W2, W2T)
            targets[k] = [(0, D, w.shape[1], W2, W2T, D, Ip)]
        else:
            rows, cols = w.shape
            rp, cp = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
            W = torch.empty((rp, cp), dtype=BF16, device=w.device)
            WT = torch.empty((cp, rp), dtype=BF16, device=w.device)
            jobs.append((w, W, WT, rp, cp))
            out[k] = (W[:rows], WT[:cols])
            targets[k] = [(0, rows, cols, W, WT, rp, cp)]
    return ws, vers, out, jobs, targets


def _commit_pack_plan(cache, l, plan):
    ws, vers, out, _, targets = plan
    for k, w in ws:
        cache.store[(l,) + k] = (vers[k], out[k])
        if FUSED_ADAM_PACK and w.is_contiguous():
            _register_pack(w, vers[k], cache, (l,) + k, out[k], targets[k])


# all stale layers of the stack packed by ONE launch at the top of the forward (round 5: alm_pack_weights_multi takes 40 jobs; six launches of 27-35 us
# before).  ALM_PACK_ALL=0: layer by layer as the forward reaches them (A/B switch)
PACK_ALL = os.environ.get('ALM_PACK_ALL', '1') != '0'


def pack_stack_weights(cache: WeightCache, flat, cfg):
    """re-pack the bf16 (W, W^T) images of every layer whose masters changed since they were last packed -- one launch for the whole stack"""
    S, ppl = cfg.streams, params_per_layer(cfg.streams, cfg.cross_attend)
    plans, jobs, split = [], [], []
    with torch.no_grad():
        for l in range(cfg.depth):
            split.append(_split_layer(flat[l * ppl:(l + 1) * ppl], S, cfg.cross_attend))
            plan = _layer_pack_plan(cache, l, split[-1], cfg.inner, cfg.inner_pad)
            if plan is not None:
                plans.append((l, plan))
                jobs += plan[3]
        if jobs:
            ops.pack_weights_multi(jobs)
    for l, plan in plans:
        _commit_pack_plan(cache, l, plan)
    return split                       # per layer: its branches -- every layer's images are current now (layer_weights(..., current=True) skips its own scan)


def layer_weights(cache: WeightCache, l, branches, I, Ip, current=False):
    """bf16 packed (W, W^T) pairs of one layer's dense weights -> {kind: {name: (W, WT)}}; when any master changed, ALL of the layer are re-packed
    (one launch).  branches: _split_layer() of the layer's detached parameters.  current: pack_stack_weights() has just confirmed (or refreshed) this
    layer's images in this call -- no second version scan (it doubled the per-token cache-check cost of a decode step)."""
    if not current:
        with torch.no_grad():
            plan = _layer_pack_plan(cache, l, branches, I, Ip)
            if plan is not None:
                ops.pack_weights_multi(plan[3])
        if plan is not None:
            _commit_pack_plan(cache, l, plan)
    res = {}
    for kind, d, _ in branches:
        for name in ('wq', 'wkv', 'wo', 'w1', 'w2'):
            if name in d:
                res.setdefault(kind, {})[name] = cache.store[(l, kind, name)][1]
    return res


def default_residual_bf16():
    """ALM_RESIDUAL_DTYPE = auto | fp32 | bf16: storage of the hyper-connection residual streams (see StackCfg.residual_bf16) of a Transformer built
    without `residual_dtype=`.  -> True / False, or None for `auto` (the default since round 4): decided per forward by `autocast_bf16()` -- bf16 when
    the call runs under `torch.autocast(bfloat16)`, which is how the reference trainer runs the model (trainer.py:1241 `accelerator.autocast()`: the
    reference's own streams are bf16 tensors there), fp32 otherwise (a plain fp32 call of the reference keeps fp32 streams).  A drop-in user of trainer.py
    therefore gets the storage bench.py measures without setting anything."""
    v = os.environ.get('ALM_RESIDUAL_DTYPE', 'auto').lower()
    if v not in ('auto', 'fp32', 'float32', 'bf16', 'bfloat16'):
        raise ValueError(f'ALM_RESIDUAL_DTYPE={v!r}: expected auto, fp32 or bf16')
    return None if v == 'auto' else v in ('bf16', 'bfloat16')


def autocast_bf16():
    """is the caller inside torch.autocast(device_type='cuda', dtype=torch.bfloat16)?  (the HIP kernels themselves always feed bf16 operands to the
    matrix cores; this only selects the residual-stream storage of `residual_dtype=None` models)"""
    try:
        return bool(torch.is_autocast_enabled('cuda')) and torch.get_autocast_dtype('cuda') == torch.bfloat16
    except TypeError:                                                # older torch: no device argument
        return bool(torch.is_autocast_enabled()) and torch.get_autocast_gpu_dtype() == torch.bfloat16


def _empty(shape, dtype, dev):
    return ops._new(shape, dtype, dev)                   # (torch.empty, or the arena of a stack pass that launchlist.py is sizing / recording)


class DecodeCache:
    """Per-layer key / value cache of an autoregressive sampling run (reference kv_cache, audiolm_pytorch.py:360-394 / :560): bf16
    [B, nmax, 2 * dim_head] per layer holding k | v, v already value-residual mixed (so a sampling step never re-mixes old positions).
    `length` = number of cached positions."""

    def __init__(self, cfg, B, nmax, device):
        self.kv = [torch.zeros((B, nmax, 2 * cfg.dim_head), dtype=BF16, device=device) for _ in range(cfg.depth)]
        self.B, self.nmax, self.length = B, nmax, 0
        # hipGraph replay of the single-position step (Transformer._sample): the position then lives on the device (`pos_dev` == length)
        self.pos_dev = None
        self.graph = self.x_in = self.h_out = self.mask_in = None
        self.ctx = None                 # (context, context_mask) of a conditioned run, projected once (audiolm_pytorch._state_condition)
        self.frozen = False             # True while a step is being CAPTURED (recorded, not executed): host bookkeeping must not advance


class Context:
    """Conditioning context of one forward (reference Transformer.forward(context=, context_mask=), audiolm_pytorch.py:461-470): text embeddings
    already projected to the model / context width.  x bf16 [B*m, Dc] (row (b m)), mask uint8 [B, m] | None."""

    def __init__(self, x, mask_u8, B, m):
        self.x, self.mask, self.B, self.m = x, mask_u8, B, m


def _attn_seed():
    """64-bit seed of one forward's attention-dropout streams, drawn from torch's CPU generator (torch.manual_seed reproduces it; no device sync).
    A module-level function on purpose: the GPU parity test pins it and restates the keep masks for the oracle."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


_GRAPH_SEEDS = {}


def graph_seed_state(dev, create=False):
    """Device-side counter (int64 [1]) of the attention-dropout mask streams of CAPTURED steps (graphed.GraphedTrainStep).  A host seed passed by
    value is baked into a captured launch, so every replay of the graph would repeat one and the same keep mask; inside a capture the flash kernels
    therefore take `seed + counter[0]`, read when they run, and the captured step advances the counter (an in-place add that is itself part of the
    graph): each replay draws fresh masks, forward and backward of one replay see the same value.  The tensor must exist BEFORE the capture starts
    (memory allocated inside a capture belongs to the graph's pool and has no contents until a replay): GraphedTrainStep creates it (create=True),
    initialised from torch's CPU generator like the eager seeds (`torch.manual_seed` reproduces a run)."""
    key = (dev.type, dev.index)
    st = _GRAPH_SEEDS.get(key)
    if st is None:
        if not create or (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
            raise RuntimeError('attention dropout inside a hipGraph capture needs core.graph_seed_state(device, create=True) BEFORE the capture '
                               '(graphed.GraphedTrainStep does it): the mask-stream counter must be a live device tensor')
        st = _GRAPH_SEEDS[key] = torch.tensor([_attn_seed()], dtype=torch.int64, device=dev)
    return st


def _run_attn(cfg, W, prm, X, XN, mask_u8, B, N, st, bias, kv_out, decode, l, ctx, drop=(0., 0, None), ao_out=None):
    """self-attention branch (audiolm_pytorch.py:307-406): -> (Y bf16 [M, D], saved dict).  drop = (p, seed): Attention(dropout=p) in training mode --
    dropout on the attention probabilities (attend.py:92 / :140; in-kernel for the flash part, a drawn 0 / 1 mask for a prefix / dense-bias part) and
    the nn.Dropout(p) behind to_out (:304): a 0 / 1 mask on the bf16 output with the 1 / (1 - p) factor on the fp32 alpha of the GEMMs."""
    pd, seed, seed_dev = drop
    M, D, H, dh, dev = B * N, cfg.dim, cfg.heads, cfg.dim_head, X.device
    (Wq, _), (Wkv, _), (Wo, _) = W['wq'], W['wkv'], W['wo']
    Q = _empty((M, H * dh), BF16, dev)
    KV = _empty((M, 2 * dh), BF16, dev)
    fside = st.get('fside')
    if fside is not None:
        # to_kv (N = 128: a quarter of the chip for ~17 us) runs on a side stream UNDER to_q: both only read the branch input
        fside.run(lambda: ops.gemm_nt(X, Wkv, KV), X, KV)
        ops.gemm_nt(XN, Wq, Q)
        fside.join()
    else:
        # to_q || to_kv in one launch (round 5): k / v from the UN-normalised branch input (:325 vs :347); alone, the N = 128 projection is 128
        # workgroups waiting on an HBM round trip per K-step on a quarter of the CU slots
        if QKV_GROUP == 1 or (QKV_GROUP < 0 and XN.shape[0] <= QKV_GROUP_MAX_M):
            ops.gemm_nt_group2(XN, Wq, Q, X, Wkv, KV)
        else:
            ops.gemm_nt(XN, Wq, Q)
            ops.gemm_nt(X, Wkv, KV)                      # (128 tiles: takes the 4-stage DMA-ring form of the 128 x 128 tile, alm tile 16)
    K, Vown = KV[:, :dh], KV[:, dh:]
    if cfg.add_value_residual and st['kv0'] is not None:
        V = ops.value_residual_mix(Vown, st['kv0'][:, dh:])  # :357-358
    else:
        V = Vown
    if st['kv0'] is None:
        st['kv0'] = KV                                   # :534-535 (layer-0 values, pre-mix)
    sv = dict(Q=Q, KV=KV, V=V, mixed=V is not Vown)
    pre = None
    if cfg.prefix:
        # cond_as_self_attn_prefix (:330-345): keys / values of the prepended context come from the SAME to_kv, un-normalised like x; its values
        # take part in the value residual (orig_v covers the prefix rows too, :353-358)
        m = ctx.m
        KVp = _empty((B * m, 2 * dh), BF16, dev)
        ops.gemm_nt(ctx.x, Wkv, KVp)
        Vp, pmixed = KVp[:, dh:], False
        if cfg.add_value_residual and st['kvp0'] is not None:
            Vp, pmixed = ops.value_residual_mix(Vp, st['kvp0'][:, dh:]), True
        if st['kvp0'] is None:
            st['kvp0'] = KVp
        pre = dict(KVp=KVp, ke=KVp[:, :dh].reshape(B, m, dh), ve=Vp.reshape(B, m, dh), pmixed=pmixed)
    dense = isinstance(bias, relpos.DenseBias)
    if dense and decode is not None:
        raise NotImplementedError('a dense attn_bias tensor with the native kv cache is not implemented (the structured relpos.AttnBias is; the tensor '
                                  'protocol recomputes dense-bias models without a cache)')
    if decode is not None:
        assert pre is None, 'the reference turns the kv cache off for prefix conditioning (audiolm_pytorch.py:481-482)'
        kv_new = KV if V is Vown else torch.cat((K, V), dim=1)       # k | value-residual-mixed v of the new position
        AO, LSE = ops.mqa_decode_attn(Q, decode.kv[l], kv_new, decode.length, mask_u8, H, dh, bias=bias, pos_dev=decode.pos_dev), None
    elif dense:
        # an arbitrary dense attn_bias (reference math path, attend.py:98-146): GEMM scores + bias + causal / key mask + softmax + GEMM values.
        # With a conditioning prefix the key set is [prefix | sequence] (audiolm_pytorch.py:330-345): the bias gets zero columns for the prefix keys
        # (F.pad(attn_bias, (m, 0)), :345), the key mask the prefix mask in front, and "causal" means key e <= query n + m (attend.py:131-134)
        kd, vd, md, bd = K.reshape(B, N, dh), V.reshape(B, N, dh), mask_u8, bias.tbl
        if pre is not None:
            m = ctx.m
            kd, vd = torch.cat((pre['ke'], kd), dim=1), torch.cat((pre['ve'], vd), dim=1)
            ones = lambda n_: torch.ones((B, n_), dtype=torch.uint8, device=dev)
            md = None if (mask_u8 is None and ctx.mask is None) else torch.cat((ones(m) if ctx.mask is None else ctx.mask, ones(N) if mask_u8 is None else mask_u8), dim=1).contiguous()
            bd = torch.cat((torch.zeros((H, N, m), dtype=F32, device=dev), bd), dim=2).contiguous()
        AO, LSE, sv['dense'] = xattn.extra_attn_fwd(Q, kd, vd, md, B, N, H, dh, float(dh) ** -0.5, bias=bd, causal=True, dropout_p=pd)
    else:
        AO, LSE = ops.mqa_attn_fwd(Q, K, V, mask_u8, B, N, H, dh, bias=bias, dropout_p=pd, seed=seed, o=ao_out if pre is None else None, seed_dev=seed_dev)
        if pre is not None:
            AO, LSE, xs = xattn.extra_attn_fwd(Q, pre['ke'], pre['ve'], ctx.mask, B, N, H, dh, float(dh) ** -0.5, o_self=AO, lse_self=LSE, dropout_p=pd)
            pre['xs'] = xs
    if kv_out is not None and decode is None:                # a (prefix) forward of a sampling run / of the cache protocol: keep every layer's k | v
        kv_out.kv[l][:, :N, :dh] = K.reshape(B, N, dh)
        kv_out.kv[l][:, :N, dh:] = V.reshape(B, N, dh)
    Y = _empty((M, D), BF16, dev)
    okeep, oalpha = None, 1.0
    if pd > 0.:
        okeep, oalpha = _dropout_keep((M, D), pd, dev), 1. / (1. - pd)
    ops.gemm_nt(AO, Wo, Y, alpha=oalpha)
    if okeep is not None:
        Y = Y * okeep
    sv.update(AO=AO, LSE=LSE, pre=pre, drop=(pd, seed, seed_dev), okeep=okeep, oalpha=oalpha)
    return Y, sv


def _run_cross(cfg, W, prm, XN, B, N, st, ctx, pd=0.):
    """cross-attention branch (Attention(dim_context, num_null_kv=1, norm_context=True, causal=False), :450; forward :307-406 with `context`):
    q from the pre-LayerNorm'd stream, k / v from context_norm(context), value residual across the cross-attention layers (:541-544), one
    learned null key / value in front (:372-376) which the context mask never hides (:384-385)."""
    M, D, H, dh, dev = B * N, cfg.dim, cfg.heads, cfg.dim_head, XN.device
    (Wq, _), (Wkv, _), (Wo, _) = W['wq'], W['wkv'], W['wo']
    m = ctx.m
    Q = _empty((M, H * dh), BF16, dev)
    ops.gemm_nt(XN, Wq, Q)
    CN, _, cmean, crstd = ops.layernorm_fwd(ctx.x, prm['ctx_ln'])              # context_norm (:323)
    KVc = _empty((B * m, 2 * dh), BF16, dev)
    ops.gemm_nt(CN, Wkv, KVc)
    Vc, cmixed = KVc[:, dh:], False
    if cfg.add_value_residual and st['kvc0'] is not None:
        Vc, cmixed = ops.value_residual_mix(Vc, st['kvc0'][:, dh:]), True
    if st['kvc0'] is None:
        st['kvc0'] = KVc
    nkv = prm['null_kv'].to(BF16)                                              # [2, 1, dh]
    ke = torch.cat((nkv[0].expand(B, 1, dh), KVc[:, :dh].reshape(B, m, dh)), dim=1)
    ve = torch.cat((nkv[1].expand(B, 1, dh), Vc.reshape(B, m, dh)), dim=1)
    emask = None
    if ctx.mask is not None:
        emask = torch.cat((torch.ones((B, 1), dtype=torch.uint8, device=dev), ctx.mask), dim=1).contiguous()
    AO, LSE, xs = xattn.extra_attn_fwd(Q, ke, ve, emask, B, N, H, dh, float(dh) ** -0.5, dropout_p=pd)
    Y = _empty((M, D), BF16, dev)
    okeep, oalpha = None, 1.0
    if pd > 0.:
        okeep, oalpha = _dropout_keep((M, D), pd, dev), 1. / (1. - pd)
    ops.gemm_nt(AO, Wo, Y, alpha=oalpha)
    if okeep is not None:
        Y = Y * okeep
    return Y, dict(Q=Q, CN=CN, cmean=cmean, crstd=crstd, KVc=KVc, cmixed=cmixed, AO=AO, xs=xs, okeep=okeep, oalpha=oalpha)


def _dropout_keep(shape, p, device):
    """0 / 1 keep mask (bf16) of an nn.Dropout(p).  A module-level function on purpose: the GPU parity test swaps it for a seeded generator and hands
    the same masks to the oracle."""
    return torch.empty(shape, dtype=BF16, device=device).bernoulli_(1. - p)


def _run_ff(cfg, W, prm, XN, M, p_drop=0., hn_out=None):
    """feed-forward branch (audiolm_pytorch.py:246-260).  p_drop > 0 (training only): the nn.Dropout(ff_dropout) between the inner LayerNorm and
    the output projection (:258) -- the 0 / 1 mask is applied to HN in bf16 (exact), the 1 / (1 - p) factor rides on the fp32 alpha of the W2 GEMM
    (and of its two backward GEMMs), so no rounded scale ever touches the activations."""
    I, Ip, dev = cfg.inner, cfg.inner_pad, XN.device
    (W1, _), (W2, _) = W['w1'], W['w2']
    U = _empty((M, 2 * Ip), BF16, dev)
    ops.gemm_nt(XN, W1, U)
    HN, mean3, rstd3 = ops.geglu_ln_fwd(U, prm['ln3'], I, Ip, out=hn_out)
    keep, alpha = None, 1.0
    if p_drop > 0.:
        keep = _dropout_keep((M, Ip), p_drop, dev)
        HN = HN * keep
        alpha = 1. / (1. - p_drop)
    Y = _empty((M, cfg.dim), BF16, dev)
    ops.gemm_nt(HN, W2, Y, alpha=alpha)
    return Y, dict(U=U, HN=HN, mean3=mean3, rstd3=rstd3, keep=keep, alpha=alpha)


def stack_forward(x, mask_u8, flat, cfg: StackCfg, cache: WeightCache, need_grad: bool, bias=None, kv_out=None, decode=None, ctx=None, ff_dropout=0.,
                  attn_dropout=0., defer_wgrad=False, drop_seed=None):
    """x fp32 [B, N, D] -> (hn fp32 [B*N, D], saved-for-backward | None).  bias: relpos.AttnBias (structured score bias shared by every layer,
    audiolm_pytorch.py:500-506 / :532) or None.  ctx: Context (cross-attention layers / self-attention prefix) or None.  Sampling: `kv_out`
    (DecodeCache) is filled with every layer's k / v of this (prefix) forward; `decode` (DecodeCache) means x holds ONE new position per
    sequence (N == 1) at index decode.length: its attention runs over the cache (alm_mqa_decode_attn, which also appends the new k / v),
    everything else is the same launch sequence on B rows.

    Every layer is a sequence of BRANCHES (attention, [cross-attention], feed-forward), each wrapped by a hyper-connection (S > 1: the depth
    connection of the previous branch is fused with the width connection + pre-LayerNorm of the next one: one pass over the residual
    streams per branch) or by a plain residual (S == 1)."""
    B, N, D = x.shape
    M, S = B * N, cfg.streams
    I, Ip = cfg.inner, cfg.inner_pad
    ppl = params_per_layer(S, cfg.cross_attend)
    assert (ctx is not None) == (cfg.cross_attend or cfg.prefix), 'conditioned stack <-> conditioning context'
    saved = dict(branches=[], B=B, N=N) if need_grad else None

    R = x.reshape(M, D)                  # S > 1: the stream expansion (:524) is never materialised -- the first branch reads x for every stream
    rb = S > 1
    rdt = BF16 if (cfg.residual_bf16 and S > 1) else F32
    st = dict(kv0=None, kvp0=None, kvc0=None)
    stk = None
    if (defer_wgrad and need_grad and ctx is None and decode is None and kv_out is None and ff_dropout == 0. and attn_dropout == 0.
            and not isinstance(bias, relpos.DenseBias) and deferred_bytes(cfg, M) <= DEFER_MAX_BYTES):
        L, dev = cfg.depth, x.device                     # operands of the deferred weight-gradient GEMMs, stacked over the layers (see DEFER_WGRAD)
        stk = dict(XNat=_empty((L, M, D), BF16, dev), Xat=_empty((L, M, D), BF16, dev), AO=_empty((L, M, cfg.heads * cfg.dim_head), BF16, dev),
                   XNff=_empty((L, M, D), BF16, dev), HN=_empty((L, M, Ip), BF16, dev))
    if ASYNC_KV and decode is None and kv_out is None and M >= 4096 and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        st['fside'] = _SideStream(x.device, True)
    # attention-dropout mask streams: one host draw per forward, layer l's self-attention uses stream seed + l; inside a hipGraph capture the
    # caller hands (offset, device counter) instead (TransformerStackFn.forward / graph_seed_state: a by-value seed would freeze the masks of every replay)
    drop_dev = None
    if drop_seed is not None:
        drop_seed, drop_dev = drop_seed
    else:
        drop_seed = _attn_seed() if attn_dropout > 0. else 0
    pend_y = pend_coef = None            # S > 1: branch output + coefficient record whose depth connection is still to be applied
    split = pack_stack_weights(cache, flat, cfg) if PACK_ALL else None
    for l in range(cfg.depth):
        branches = split[l] if split is not None else _split_layer(flat[l * ppl:(l + 1) * ppl], S, cfg.cross_attend)
        LW = layer_weights(cache, l, branches, I, Ip, current=split is not None)
        for kind, prm, first in branches:
            want_x = kind == 'attn'                  # only to_kv reads the un-normalised branch input
            xo = xno = None
            if stk is not None:
                xo, xno = (stk['Xat'][l], stk['XNat'][l]) if kind == 'attn' else (None, stk['XNff'][l])
            if S > 1:
                h = ops.hc_fwd(R, B, S, N, D, y_prev=pend_y, coef_prev=pend_coef, hc=prm['hc'], ln_gamma=prm['ln'], rin_bcast=rb, r_dtype=rdt,
                               want_x=want_x, x_out=xo, xn_out=xno)
                r_bcast = pend_y is None                 # R is still the un-expanded x (first branch only)
                R, X, XN, mean, rstd, coef = h['R'], h['x'], h['xn'], h['mean'], h['rstd'], h['coef']
                rb = rb and pend_y is None               # still the un-expanded x after a width-only call
            else:
                XN, X, mean, rstd = ops.layernorm_fwd(R, prm['ln'], want_copy=want_x, y_out=xno, xc_out=xo)
                coef, r_bcast = None, False
            if kind == 'attn':
                Y, sv = _run_attn(cfg, LW['attn'], prm, X, XN, mask_u8, B, N, st, bias, kv_out, decode, l, ctx, (float(attn_dropout), drop_seed + l, drop_dev),
                                  ao_out=stk['AO'][l] if stk is not None else None)
            elif kind == 'cross':
                Y, sv = _run_cross(cfg, LW['cross'], prm, XN, B, N, st, ctx, float(attn_dropout))
            else:
                Y, sv = _run_ff(cfg, LW['ff'], prm, XN, M, ff_dropout, hn_out=stk['HN'][l] if stk is not None else None)
            if need_grad:
                sv.update(kind=kind, layer=l, first=l * ppl + first, R=R, X=X, XN=XN, mean=mean, rstd=rstd, coef=coef, Y=Y, r_bcast=r_bcast)
                saved['branches'].append(sv)
            if S > 1:
                pend_y, pend_coef = Y, coef
            else:
                R = ops.residual_add(R, Y)

    if S > 1:
        # last depth connection + stream sum (:551) + final LayerNorm (:555) in one pass; the final residual streams are never stored
        h = ops.hc_fwd(R, B, S, N, D, y_prev=pend_y, coef_prev=pend_coef, ln_gamma=flat[-1], final=True, r_dtype=rdt, final_f32=True)
        xs, hn, fmean, frstd = h['xs'], h['xn'], h['mean'], h['rstd']
    else:
        xs = R
        hn, _, fmean, frstd = ops.layernorm_fwd(xs, flat[-1], out_f32=True)  # :555 (fp32 out: F.layer_norm autocasts to fp32; the logit heads split it)
    if need_grad:
        saved.update(xs=xs, fmean=fmean, frstd=frstd, st=st, ctx=ctx, stk=stk)
    if decode is not None and not decode.frozen:
        decode.length += 1
    if kv_out is not None:
        kv_out.length = N
    return hn, saved


class _SideStream:
    """Weight-gradient GEMMs are off the critical path of the backward chain (nothing downstream reads dW before the optimiser / the
    gradient all-reduce), so they are issued on side HIP streams and fill CUs the critical-path kernels leave idle (tails of the
    causal attention grids, HBM-bound hyper-connection / LayerNorm kernels, launch gaps).  Inputs are pinned with record_stream so the
    caching allocator cannot recycle them while a side stream still reads them.  With more than one side stream (ALM_SIDE_STREAMS) the
    GEMMs are dealt round-robin, so the partial last wave of one weight gradient also overlaps the next one."""
    _streams = {}

    def __init__(self, dev, enabled=True):
        self.enabled = enabled and dev.type == 'cuda'
        if self.enabled:
            self.main = torch.cuda.current_stream(dev)
            # one set of side streams per MAIN stream: two half-batches running on two streams (graphed.GraphedTrainStep) must not serialise
            # their weight gradients behind each other
            key = (dev.type, dev.index, SIDE_STREAMS, self.main.cuda_stream)
            if key not in _SideStream._streams:
                _SideStream._streams[key] = [torch.cuda.Stream(device=dev) for _ in range(SIDE_STREAMS)]
            self.streams = _SideStream._streams[key]
            self.turn = 0

    def _issue(self, stream, fn, tensors):
        ev = torch.cuda.Event()
        ev.record(self.main)
        for t in tensors:
            t.record_stream(stream)
        with torch.cuda.stream(stream):
            stream.wait_event(ev)
            return fn()

    def run(self, fn, *tensors):
        if not self.enabled:
            return fn()
        stream = self.streams[self.turn % len(self.streams)]
        self.turn += 1
        return self._issue(stream, fn, tensors)

    def run_after_all(self, fn, *tensors):
        """fn on side stream 0, ordered after everything issued on every side stream so far (the per-layer gradient hand-off)"""
        if not self.enabled:
            return fn()
        s0 = self.streams[0]
        for s in self.streams[1:]:
            ev = torch.cuda.Event()
            ev.record(s)
            s0.wait_event(ev)
        return self._issue(s0, fn, tensors)

    def join(self):
        """main stream waits for everything issued on the side streams so far"""
        if self.enabled:
            for s in self.streams:
                ev = torch.cuda.Event()
                ev.record(s)
                self.main.wait_event(ev)


ASYNC_WGRAD = os.environ.get('ALM_ASYNC_WGRAD', '1') != '0'          # switch (ALM_ASYNC_WGRAD=0 turns the side stream off: A/B runs)
# forward to_q / to_kv: two launches, to_kv (128 tiles) on the 4-stage DMA-ring form of the 128 x 128 tile (default; measured -0.03 ms/step against the
# grouped launch, profiles/r5t_ab_qkv_ring.log); 1: to_q || to_kv as ONE grouped launch (alm_gemm_bf16_nt_group2)
# Round 6: per shape class -- `auto` (default) groups them when the batch has at most QKV_GROUP_MAX_M token rows: at M = 8 192 (configs[1]) the grouped launch
# is the faster one (-0.03 ... -0.05 ms/step, profiles/r5v_coarse1024_switches.log, r6u_qkv_group_ab.log), at M = 16 384 the ring tile is.
QKV_GROUP = {'0': 0, '1': 1}.get(os.environ.get('ALM_QKV_GROUP', 'auto'), -1)
QKV_GROUP_MAX_M = 8192
ASYNC_KV = os.environ.get('ALM_ASYNC_KV', '0') != '0'                # forward: to_kv on a side stream under to_q (A/B switch; measured +-0: off)
# Deferred weight gradients (round 3): instead of one split-K GEMM + reduce per weight and layer as the backward walks down, the activations the
# weight gradients need (dU, dY, dQ, dKV; XN, X, AO, HN from the forward) are written into buffers STACKED over the layers, and each weight kind is
# computed for all layers at once at the end of the backward pass (ops.gemm_tn_batched: 5 launches instead of 30 + 30 reduces, long K slices, the
# chip full).  Used when nothing needs a layer's gradients early (no data-parallel hook) and no dropout mask sits on the operands.
# ALM_DEFER_GROUPS: the layers can be cut into groups whose GEMMs start as soon as the backward has passed them (2 groups measured 0.04 ms better than 1
# with the uniform split-K plan; with the hybrid plan of alm_gemm_bf16_tn_batched all 6 layers in one group are 0.27 ms / step better: 528 and 264 tiles
# are whole waves of the chip plus a small tail, 3 layers of dW2 are 132 tiles and fall back to the uniform plan).
# Tried and dropped: re-packing every layer's bf16 weight copies on a side stream at the start of the step (so that the 35-us packs of layers 1..5 run
# under layer 0's kernels): +0.26 ms / step on three interleaved runs -- the HBM-bound packs slow the critical kernels more than they hide.
DEFER_WGRAD = os.environ.get('ALM_DEFER_WGRAD', '1') != '0'
# Memory cost of the deferred mode: the gradient-side operands (dY of both branches, dU, dQ, dKV) of EVERY layer stay alive until the batched launches
# at the end of the backward pass, where the per-layer path frees them layer by layer -- L * M * (2 D + 2 Ipad + H dh + 2 dh) * 2 bytes more at the peak
# (1.6 GB at the headline shape: nothing against 288 GB, but it grows with batch x sequence).  Above ALM_DEFER_MAX_GB (default 24) the stack falls back to
# the per-layer split-K path instead of failing an allocation.
DEFER_MAX_BYTES = int(float(os.environ.get('ALM_DEFER_MAX_GB', '24')) * (1 << 30))


def deferred_bytes(cfg, M):
    """extra peak bytes of the deferred weight-gradient mode for M = B * N rows (see DEFER_MAX_BYTES)"""
    return cfg.depth * M * (2 * cfg.dim + 2 * cfg.inner_pad + cfg.heads * cfg.dim_head + 2 * cfg.dim_head) * 2
DEFER_GROUPS = max(1, int(os.environ.get('ALM_DEFER_GROUPS', '1')))
# Data-parallel runs (a per-layer gradient hook is attached: parallel.DataParallelEngine) take the SAME deferred path since round 4, cut into
# ALM_DP_DEFER_GROUPS layer groups (default 2): each group's batched weight-gradient launches go to the side stream as soon as the backward has passed the
# group, followed by ONE gradient bucket for the whole group (fewer, larger collectives: a ring over xGMI is per-link bound), so the upper group's
# all-reduce runs under the lower layers' backward and only the last group's is exposed.  Until round 3 the hook forced the per-layer split-K path
# (30 + 30 launches, +0.8 ms/step per GPU against the single-GPU step).  0 = that per-layer path (one bucket per layer, maximal overlap, slower GEMMs).
DP_DEFER_GROUPS = max(0, int(os.environ.get('ALM_DP_DEFER_GROUPS', '2')))
# Round 6: UNEVEN layer groups for the data-parallel path, in backward order (top of the stack first), e.g. ALM_DP_GROUP_SIZES=4,2 at depth 6: the LAST group's
# bucket is the one whose all-reduce nothing hides (the backward is over when it starts), so it should be the small one.  Sizes that do not sum to the depth
# are rejected at the first backward.  DEFAULT (variable unset) with 2 groups: (L - 1, 1) -- the bottom layer alone closes the backward.  Measured at N = 1
# (profiles/r6n_groups_ab.log, depth 6, 3 interleaved rounds): one group 11.08 ms/step, (3,3) +0.33, (4,2) +0.48, (5,1) +0.22 -- the cheapest cut is also the
# one with the smallest exposed bucket (62 MB: one layer's dW1 / dW2 + the small kinds of all layers, against 129 MB for (3,3)).  ALM_DP_GROUP_SIZES=even:
# equal groups (rounds 4-5).
_DPGS = os.environ.get('ALM_DP_GROUP_SIZES', '').replace(' ', '').lower()
DP_GROUP_SIZES = None if _DPGS == '' else (() if _DPGS == 'even' else tuple(int(v) for v in _DPGS.split(',') if v))


def dp_group_sizes(L, ngroups):
    """group sizes (backward order) of the data-parallel step's weight-gradient cut: the explicit ALM_DP_GROUP_SIZES, else (L - 1, 1) for two groups"""
    if DP_GROUP_SIZES is not None:
        return DP_GROUP_SIZES
    return (L - 1, 1) if (ngroups == 2 and L >= 3) else ()
# the same cut without a gradient hook (single GPU): what the grouping itself costs at N = 1 is measured with this (bench.py --gpus 1)
DEFER_GROUP_SIZES = tuple(int(v) for v in os.environ.get('ALM_DEFER_GROUP_SIZES', '').replace(' ', '').split(',') if v)


def group_starts(L, ngroups, sizes=()):
    """first layer of every weight-gradient group of an L-layer stack, ascending.  sizes: group sizes in BACKWARD order (top group first); else `ngroups`
    equal groups (the last one may be short)"""
    if sizes:
        if sum(sizes) != L or min(sizes) < 1:
            raise ValueError(f'ALM_DP_GROUP_SIZES={sizes}: the group sizes must be positive and sum to the depth {L}')
        starts, top = [], L
        for n in sizes:
            top -= n
            starts.append(top)
        return sorted(starts)
    gsz = (L + ngroups - 1) // ngroups
    return list(range(0, L, gsz))
# deferred mode: the hyper-connection parameter gradients of all branches of a layer group finished together in two launches (ops.hc_param_grads_batched)
# instead of two small launches behind every hc_bwd (A/B switch)
HC_BATCH_FINISH = os.environ.get('ALM_HC_BATCH_FINISH', '1') != '0'
# with several layer groups: the small weight kinds (dWq / dWkv / dWo) of ALL layers in one batched launch each, together with the last group (see stack_backward)
DEFER_SMALL_LATE = os.environ.get('ALM_DEFER_SMALL_LATE', '1') != '0'
# Inside a hipGraph capture ALWAYS one group (see stack_backward): more than one corrupts the replays on ROCm 7.2, not root-caused.  The reproducer
# (scripts/debug/graph_defer_groups.py) sets core._DEBUG_DEFER_GROUPS_CAPTURE itself; there is deliberately no environment switch for it.
_DEBUG_DEFER_GROUPS_CAPTURE = 1
MICRO_ASYNC_WGRAD = os.environ.get('ALM_MICRO_ASYNC_WGRAD', '0') != '0'   # weight-gradient side streams inside the two-half-batch schedule
SIDE_STREAMS = max(1, int(os.environ.get('ALM_SIDE_STREAMS', '1')))    # number of side streams the weight-gradient GEMMs are dealt over


# Test hook (tests/test_gpu_opwise.py): a list that stack_backward appends one record per branch to -- the gradient tensors flowing between its
# launches (references, nothing is copied), so that every backward op can be checked on its own against the oracle given the SAME inputs.
TRACE = None


def _vgrad_mode(acc, mixed):
    """alm_kv_grad_pack mode: 0 no value residual, 1 this layer's v was mixed (half of dv goes to layer 0's accumulator), 2 layer 0 (receives it)"""
    return 0 if acc is None else (1 if mixed else 2)


def ff_backward(cfg, W, prm, sv, dY, side, want_trace=False, du_out=None):
    """backward of the feed-forward branch (the launches of _run_ff in reverse): dY bf16 [M, D] -> (dXN bf16 [M, D], dW1 fp32 [2 I, D], d gamma3 [I],
    dW2 fp32 [D, I]).  The two weight gradients go to `side` (a _SideStream) when one is given."""
    M, D = dY.shape
    I, Ip, dev = cfg.inner, cfg.inner_pad, dY.device
    (_, W1T), (_, W2T) = W['w1'], W['w2']
    run = side.run if side is not None else (lambda fn, *t: fn())
    dHN = _empty((M, Ip), BF16, dev)
    fa = sv['alpha']                                                      # 1 / (1 - ff_dropout) (1.0 without dropout)
    ops.gemm_nt(dY, W2T, dHN, alpha=fa)                                   # dHN = dY @ W2
    if sv['keep'] is not None:
        dHN.mul_(sv['keep'])                                              # through the dropout mask (HN below is the masked HN)
    deferred = du_out is not None                                         # the two weight gradients are computed for all layers at the end
    dW2 = None
    if not deferred:
        dW2 = _empty((D, I), F32, dev)
        HNs, dYs = sv['HN'], dY
        run(lambda: ops.gemm_tn_splitk(dYs, HNs[:, :I], dW2, alpha=fa), dYs, HNs, dW2)      # dW2 = dY^T @ HN
    dU, dg3 = ops.geglu_ln_bwd(dHN, sv['U'], prm['ln3'], sv['mean3'], sv['rstd3'], I, Ip, du_out=du_out)
    dXN = _empty((M, D), BF16, dev)
    ops.gemm_nt(dU, W1T, dXN)                                             # dXN = dU @ W1
    dW1 = None
    if not deferred:
        dW1 = _empty((2 * I, D), F32, dev)
        XNs = sv['XN']
        run(lambda: ops.gemm_tn_splitk(dU.view(M, 2, Ip).permute(1, 0, 2)[:, :, :I], XNs, dW1.view(2, I, D)), dU, XNs, dW1)   # dW1 = dU^T @ XN
    return (dXN, dW1, dg3, dW2, dHN, dU) if want_trace else (dXN, dW1, dg3, dW2)


def stack_backward(dhn, mask_u8, flat, cfg: StackCfg, cache: WeightCache, saved, on_layer_grads=None, bias=None, async_wgrad=True, dx_scale=1.0):
    """dhn fp32 | bf16 [M, D] -> (dx fp32 [B, N, D] x dx_scale, list of parameter grads aligned with `flat`, d(loss)/d(bias.tbl) | None,
    d(loss)/d(context) fp32 [B*m, Dc] | None).  dx_scale: grad_shrink's alpha (audiolm_pytorch.py:93-94) -- applied by the kernel that produces dx (the
    first branch's stream-sum / residual add) instead of a separate pass over the [B, N, D] gradient."""
    B, N = saved['B'], saved['N']
    D, S, H, dh = cfg.dim, cfg.streams, cfg.heads, cfg.dim_head
    M = B * N
    I, Ip = cfg.inner, cfg.inner_pad
    dev = dhn.device
    ppl = params_per_layer(S, cfg.cross_attend)
    grads = [None] * len(flat)
    side = _SideStream(dev, ASYNC_WGRAD and async_wgrad)
    rdt = BF16 if (cfg.residual_bf16 and S > 1) else F32
    ctx = saved['ctx']
    brs = saved['branches']
    scale = float(dh) ** -0.5

    dxs, dgam = ops.layernorm_bwd(dhn, saved['xs'], saved['fmean'], saved['frstd'], flat[-1])
    grads[-1] = dgam
    multi = cfg.add_value_residual and cfg.depth > 1
    acc_v0 = ops._new_zeros((M, dh), dtype=F32, device=dev) if multi else None
    acc_vp0 = ops._new_zeros((B * ctx.m, dh), dtype=F32, device=dev) if (multi and cfg.prefix) else None
    acc_vc0 = ops._new_zeros((B * ctx.m, dh), dtype=F32, device=dev) if (multi and cfg.cross_attend) else None
    dctx = None                                                                # fp32 [B*m, Dc], summed over layers
    dense = isinstance(bias, relpos.DenseBias)
    dtbl_part = ops.attn_bias_part(B, N, H, bias.tbl.shape[1], dev) if (bias is not None and not dense) else None
    ddense = torch.zeros_like(bias.tbl) if dense else None                     # gradient of a dense attn_bias, summed over the layers
    # S > 1: dR = gradient wrt the residual streams after the current branch; right after the final stream sum it is dxs for every
    # stream (`bcast`).  dY / dbeta (depth-connection backward of the branch about to be processed) are produced one step ahead by the
    # fused kernels.
    dR, bcast = dxs, S > 1
    dY = dbeta = None
    stk = saved.get('stk')                                                     # deferred weight gradients (see DEFER_WGRAD): operands stacked over the layers
    bst = None
    if stk is not None:
        L = cfg.depth
        bst = dict(dYat=_empty((L, M, D), BF16, dev), dYff=_empty((L, M, D), BF16, dev), dU=_empty((L, M, 2 * Ip), BF16, dev),
                   dQ=_empty((L, M, H * dh), BF16, dev), dKV=_empty((L, M, 2 * dh), BF16, dev))

    def dy_slot(br):
        """where the gradient wrt branch `br`'s output goes: its slot of the stacked buffers in deferred mode"""
        return None if bst is None else bst['dYat' if br['kind'] == 'attn' else 'dYff'][br['layer']]
    if S > 1:
        h = ops.hc_bwd(dxs, B, S, N, D, bcast=True, y_prev=brs[-1]['Y'], coef_prev=brs[-1]['coef'], r_dtype=rdt, dy_out=dy_slot(brs[-1]))
        dY, dbeta = h['dy'], h['dbeta']

    wg = None
    if bst is not None:
        # the weight gradients of a GROUP of layers, one launch per weight kind (ops.gemm_tn_batched: C[l] = At[l]^T @ Bt[l], K = all tokens), issued on
        # the side stream as soon as the backward has passed the group's lowest layer: the upper groups run under the remaining layers' kernels,
        # only the last group is exposed.  ALM_DEFER_GROUPS = number of groups (1: everything at the end).
        # Inside a hipGraph capture ONE group at the end: with the first group forked in the middle of the backward pass (any kernel node on the side
        # branch while the capturing stream goes on, even one that only zeroes an unrelated buffer) the replays of the captured step came out corrupted on
        # ROCm 7.2 (NaN gradients, sometimes a wrong loss; scripts/debug/graph_defer_groups.py reproduces; a join right after
        # the group, or no side stream, cures it; the same fork-twice pattern in pure torch, scripts/ubench/capture_fork_twice.py, replays correctly).
        # Not root-caused.  A replay has no host issue to hide, so the early start is worth nothing there anyway.
        L = cfg.depth
        ngroups = _DEBUG_DEFER_GROUPS_CAPTURE if (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()) else (
            DEFER_GROUPS if on_layer_grads is None else max(1, DP_DEFER_GROUPS))
        ngroups = min(ngroups, L)
        capturing = dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        starts = group_starts(L, ngroups, () if capturing else (dp_group_sizes(L, ngroups) if on_layer_grads is not None else DEFER_GROUP_SIZES))
        ngroups = len(starts)
        on_group = getattr(on_layer_grads, 'on_group', None)                   # parallel.DataParallelEngine: one bucket per layer GROUP
        group_buffer = getattr(on_layer_grads, 'group_buffer', None)           # ... whose flat bucket the weight-gradient GEMMs write straight into
        wg = True
        wshape = dict(dW1=(2, I, D), dW2=(1, D, I), dWo=(1, D, H * dh), dWq=(1, H * dh, D), dWkv=(1, 2 * dh, D))
        hc_n_ = 7 if S > 1 else 0
        wslot = dict(dWq=hc_n_ + 1, dWkv=hc_n_ + 2, dWo=hc_n_ + 3, dW1=(hc_n_ + 4) + hc_n_ + 1, dW2=(hc_n_ + 4) + hc_n_ + 3)   # index of the weight within a layer's flat parameters

        # With more than one group the SMALL weight kinds (dWq, dWkv, dWo: 9 % of a layer's gradient bytes) still run once, for all layers, with the last
        # group: cut into groups their few tiles fall back to inefficient split-K plans (measured at N = 1: 2 groups +0.45 ms/step, 3 groups +0.72, almost
        # all of it these three kinds); only dW1 / dW2 (91 % of the bytes: what the early all-reduce is for) follow the groups.  ALM_DEFER_SMALL_LATE=0: A/B
        big_kinds, small_kinds = ('dW1', 'dW2'), ('dWo', 'dWq', 'dWkv')
        late = DEFER_SMALL_LATE and ngroups > 1
        handed = set()                                                         # indices into `grads` already handed to the gradient hook

        def group_outputs(l0, l1, kinds, late_kinds):
            """fp32 outputs of the group's batched weight-gradient launches, per weight kind ([l1 - l0, ...]; `late_kinds`: [L, ...], all layers): views into
            the data-parallel engine's persistent flat bucket of this group when it offers one (the all-reduce then runs in place: no bucket copy, no copy
            back), else fresh"""
            views = None
            if group_buffer is not None:
                views = group_buffer(l0, l1, [wslot[k] for k in kinds], [wslot[k] for k in late_kinds],
                                     [wslot[k] for k in small_kinds] if (late and not late_kinds) else [])
            if views is not None:
                return {k: v.view(-1, *wshape[k]) for k, v in zip(tuple(kinds) + tuple(late_kinds), views)}
            out = {k: _empty((l1 - l0,) + wshape[k], F32, dev) for k in kinds}
            out.update({k: _empty((L,) + wshape[k], F32, dev) for k in late_kinds})
            return out

        def launch_group(l0):
            l1 = ([s_ for s_ in starts if s_ > l0] + [L])[0]                   # the group runs up to the next group's first layer
            last = l0 == 0                                                     # the groups are launched from the top of the stack down
            kinds = big_kinds if late else big_kinds + small_kinds
            late_kinds = small_kinds if (late and last) else ()
            wgg = group_outputs(l0, l1, kinds, late_kinds)
            operands = dict(dW1=lambda a, b: (bst['dU'][a:b].view(b - a, M, 2, Ip).permute(0, 2, 1, 3)[..., :I], stk['XNff'][a:b].unsqueeze(1)),   # dW1 = dU^T @ XN (x | gate)
                            dW2=lambda a, b: (bst['dYff'][a:b].unsqueeze(1), stk['HN'][a:b][..., :I].unsqueeze(1)),                                # dW2 = dY^T @ HN
                            dWo=lambda a, b: (bst['dYat'][a:b].unsqueeze(1), stk['AO'][a:b].unsqueeze(1)),                                         # dWo = dY^T @ AO
                            dWq=lambda a, b: (bst['dQ'][a:b].unsqueeze(1), stk['XNat'][a:b].unsqueeze(1)),                                         # dWq = dQ^T @ XN
                            dWkv=lambda a, b: (bst['dKV'][a:b].unsqueeze(1), stk['Xat'][a:b].unsqueeze(1)))                                        # dWkv = dKV^T @ X
            jobs = [operands[k](l0, l1) + (wgg[k],) for k in kinds] + [operands[k](0, L) + (wgg[k],) for k in late_kinds]
            for At, Bt, C in jobs:
                if ngroups > 1 or on_layer_grads is not None:  # (with a gradient hook even a single group goes to the side stream: its bucket follows it there)
                    side.run(lambda At=At, Bt=Bt, C=C: ops.gemm_tn_batched(At, Bt, C), At, Bt, C)
                else:                                          # one group at the end: nothing left to run beside it -- the main stream, no fork / join
                    ops.gemm_tn_batched(At, Bt, C)             # (interleaved A/B: 13.11 -> 12.97 ms/step)
            for k in kinds:
                for l in range(l0, l1):
                    g_ = wgg[k][l - l0]
                    grads[l * ppl + wslot[k]] = g_.view(2 * I, D) if k == 'dW1' else g_[0]
            for k in late_kinds:
                for l in range(L):
                    grads[l * ppl + wslot[k]] = wgg[k][l, 0]
            if on_layer_grads is not None:
                # the gradient hand-off of the GROUP, ordered after its weight-gradient launches ON THE SIDE STREAM (the critical path never waits):
                # one call for the whole group when the hook takes groups (one bucket, one collective), else layer by layer in backward order.
                # Every gradient is handed over exactly once: with the small kinds late, the last call also carries those of the earlier layers.
                layers = list(range((L if late_kinds else l1) - 1, l0 - 1, -1))
                per_layer = []
                for l in layers:
                    gl = [g if (l * ppl + j) not in handed else None for j, g in enumerate(grads[l * ppl:(l + 1) * ppl])]
                    handed.update(l * ppl + j for j, g in enumerate(gl) if g is not None)
                    per_layer.append(gl)
                live = [g for gl in per_layer for g in gl if g is not None]
                if on_group is not None:
                    side.run_after_all(lambda: on_group(layers, per_layer), *live)
                else:
                    def hand_off():
                        for l, gl in zip(layers, per_layer):
                            on_layer_grads(l, gl)
                    side.run_after_all(hand_off, *live)

    hc_pending = []                                    # deferred mode: (first, ip, partial rows, trace record) of the branches whose parameter gradients wait

    def finish_hc():
        """the hyper-connection parameter gradients of every branch processed since the last call, in two launches (ops.hc_param_grads_batched)"""
        if not hc_pending:
            return
        for (first_, ip_, _, rec_), g_ in zip(hc_pending, ops.hc_param_grads_batched([t[2] for t in hc_pending], S, D)):
            for j, k in enumerate(HC_KEYS):
                grads[first_ + j] = g_[k]
            grads[ip_] = g_['ln']
            if rec_ is not None:
                rec_['hc_grads'] = dict(g_)
        hc_pending.clear()

    def add_ctx(g):
        nonlocal dctx
        dctx = g if dctx is None else ops.add_f32(dctx, g)

    LW, LW_layer = None, -1
    for bi in reversed(range(len(brs))):
        sv = brs[bi]
        kind, l, first = sv['kind'], sv['layer'], sv['first']
        if l != LW_layer:
            branches = _split_layer(flat[l * ppl:(l + 1) * ppl], S, cfg.cross_attend)
            LW, LW_layer = layer_weights(cache, l, branches, I, Ip), l
            prms = {k: (d, f) for k, d, f in branches}
        prm = prms[kind][0]
        hc_n = 7 if S > 1 else 0
        ip = first + hc_n                                                      # index of this branch's first non-hyper-connection parameter
        if S == 1:
            dY = ops.f32_to_bf16(dR, out=dy_slot(sv))
        extra = None
        W = LW[kind]
        if kind == 'ff':
            dXN, dW1, dg3, dW2, dHN, dU = ff_backward(cfg, W, prm, sv, dY, side, want_trace=True, du_out=bst['dU'][l] if bst is not None else None)
            grads[ip + 1], grads[ip + 2], grads[ip + 3] = dW1, dg3, dW2
            rec = dict(dY=dY, dHN=dHN, dU=dU, dXN=dXN, dW1=dW1, dg3=dg3, dW2=dW2) if TRACE is not None else None
        elif kind == 'cross':
            (_, WqT), (_, WkvT), (_, WoT) = W['wq'], W['wkv'], W['wo']
            m = ctx.m
            dAO = _empty((M, H * dh), BF16, dev)
            oa = sv['oalpha']
            dYs = dY if sv['okeep'] is None else dY * sv['okeep']                  # through the to_out dropout mask (:304)
            ops.gemm_nt(dYs, WoT, dAO, alpha=oa)
            dWo = _empty((D, H * dh), F32, dev)
            AOs = sv['AO']
            side.run(lambda: ops.gemm_tn_splitk(dYs, AOs, dWo, alpha=oa), dYs, AOs, dWo)
            nd = xattn.attn_delta(sv['AO'], dAO, B, N, H, dh)
            dQ, dke, dve = xattn.extra_attn_bwd(sv['Q'], dAO, sv['xs'], nd, B, N, H, dh, scale)
            dnull = torch.stack((dke[:, 0].sum(0), dve[:, 0].sum(0))).reshape(2, 1, dh)           # null_kv is shared by the batch (:373)
            dkv32 = torch.cat((dke[:, 1:].reshape(B * m, dh), dve[:, 1:].reshape(B * m, dh)), dim=1).contiguous()
            dKVc = ops.kv_grad_pack(dkv32, acc_vc0, _vgrad_mode(acc_vc0, sv['cmixed']), dh)
            dXN = _empty((M, D), BF16, dev)
            ops.gemm_nt(dQ, WqT, dXN)
            dWq = _empty((H * dh, D), F32, dev)
            XNs = sv['XN']
            side.run(lambda: ops.gemm_tn_splitk(dQ, XNs, dWq), dQ, XNs, dWq)
            dCN = _empty((B * m, cfg.dim_context), BF16, dev)
            ops.gemm_nt(dKVc, WkvT, dCN)
            dWkv = _empty((2 * dh, cfg.dim_context), F32, dev)
            CNs = sv['CN']
            side.run(lambda: ops.gemm_tn_splitk(dKVc, CNs, dWkv), dKVc, CNs, dWkv)
            dc, dgc = ops.layernorm_bwd(dCN, ctx.x, sv['cmean'], sv['crstd'], prm['ctx_ln'])      # context_norm backward -> d(context)
            add_ctx(dc)
            grads[ip + 1], grads[ip + 2], grads[ip + 3], grads[ip + 4], grads[ip + 5] = dgc, dnull, dWq, dWkv, dWo
            rec = dict(dY=dY) if TRACE is not None else None
        else:
            (_, WqT), (_, WkvT), (_, WoT) = W['wq'], W['wkv'], W['wo']
            dAO = _empty((M, H * dh), BF16, dev)
            oa = sv['oalpha']
            dYs = dY if sv['okeep'] is None else dY * sv['okeep']                  # through the to_out dropout mask (:304)
            ops.gemm_nt(dYs, WoT, dAO, alpha=oa)
            dWo = dWq = dWkv = None
            if bst is None:
                dWo = _empty((D, H * dh), F32, dev)
                AOs = sv['AO']
                side.run(lambda: ops.gemm_tn_splitk(dYs, AOs, dWo, alpha=oa), dYs, AOs, dWo)
            pd, dseed, dseed_dev = sv['drop']
            KV = sv['KV']
            # with a prefix the joint softmax statistics (LSE) and the joint output (AO) make the flash backward exact for the sequence's own keys
            dense_pre = None
            if dense:
                nd = xattn.attn_delta(sv['AO'], dAO, B, N, H, dh)
                mpre = ctx.m if sv['pre'] is not None else 0
                dbl = torch.empty((H, N, mpre + N), dtype=F32, device=dev)
                dQ, dke, dve = xattn.extra_attn_bwd(sv['Q'], dAO, sv['dense'], nd, B, N, H, dh, scale, dbias=dbl)
                ddense += dbl[:, :, mpre:]                                        # the prefix columns of the padded bias are constants (:345)
                if mpre:
                    dense_pre = (dke[:, :mpre], dve[:, :mpre])
                    dke, dve = dke[:, mpre:], dve[:, mpre:]
                dkv32 = torch.cat((dke.reshape(M, dh), dve.reshape(M, dh)), dim=1).contiguous()
            else:
                dQ, dkv32 = ops.mqa_attn_bwd(sv['Q'], KV[:, :dh], sv['V'], mask_u8, sv['AO'], sv['LSE'], dAO, B, N, H, dh, bias=bias, dtbl_part=dtbl_part,
                                             dropout_p=pd, seed=dseed, dq_out=bst['dQ'][l] if bst is not None else None, seed_dev=dseed_dev)
            dKV = ops.kv_grad_pack(dkv32, acc_v0, _vgrad_mode(acc_v0, sv['mixed']), dh, out=bst['dKV'][l] if bst is not None else None)
            dKVp = None
            pre = sv['pre']
            if pre is not None:
                m = ctx.m
                if dense_pre is not None:
                    dke, dve = dense_pre                                          # the dense path already covered the prefix keys
                else:
                    nd = xattn.attn_delta(sv['AO'], dAO, B, N, H, dh)
                    dQ, dke, dve = xattn.extra_attn_bwd(sv['Q'], dAO, pre['xs'], nd, B, N, H, dh, scale, dq=dQ)
                dkvp32 = torch.cat((dke.reshape(B * m, dh), dve.reshape(B * m, dh)), dim=1).contiguous()
                dKVp = ops.kv_grad_pack(dkvp32, acc_vp0, _vgrad_mode(acc_vp0, pre['pmixed']), dh)
                dcp = _empty((B * m, D), F32, dev)
                ops.gemm_nt(dKVp, WkvT, dcp)                                      # the prefix enters to_kv un-normalised: d(context) directly
                add_ctx(dcp)
            dXN = _empty((M, D), BF16, dev)
            extra = _empty((M, D), BF16, dev)
            ops.gemm_nt_group2(dQ, WqT, dXN, dKV, WkvT, extra)                    # (extra: the K/V-path gradient reaches the un-normalised branch input directly)
            if bst is None:
                dWq = _empty((H * dh, D), F32, dev)
                XNs = sv['XN']
                side.run(lambda: ops.gemm_tn_splitk(dQ, XNs, dWq), dQ, XNs, dWq)
                dWkv = _empty((2 * dh, D), F32, dev)
                Xs, cx = sv['X'], (ctx.x if dKVp is not None else None)

                def wkv_grad():
                    ops.gemm_tn_splitk(dKV, Xs, dWkv)
                    if dKVp is not None:
                        ops.gemm_tn_splitk(dKVp, cx, dWkv, accumulate=True)
                side.run(wkv_grad, *[t for t in (dKV, Xs, dWkv, dKVp, cx) if t is not None])
            grads[ip + 1], grads[ip + 2], grads[ip + 3] = dWq, dWkv, dWo
            rec = dict(dY=dY, dAO=dAO, dQ=dQ, dkv32=dkv32, dKV=dKV, dXN=dXN, extra=extra, dWq=dWq, dWkv=dWkv, dWo=dWo) if TRACE is not None else None

        # ---- this branch's pre-LayerNorm backward + width-connection backward + the depth-connection backward of the previous branch: ONE pass
        prev = brs[bi - 1] if bi > 0 else None
        if S > 1:
            py, pc = (prev['Y'], prev['coef']) if prev is not None else (None, None)
            h = ops.hc_bwd(dR, B, S, N, D, bcast=bcast, dxn=dXN, extra=extra, mean=sv['mean'], rstd=sv['rstd'], ln_gamma=prm['ln'], R=sv['R'],
                           coef=sv['coef'], dbeta=dbeta, hc=prm['hc'], y_prev=py, coef_prev=pc, r_bcast=sv['r_bcast'], sum_only=prev is None,
                           r_dtype=rdt, dsum_scale=dx_scale if prev is None else 1.0, dy_out=dy_slot(prev) if prev is not None else None,
                           defer_grads=wg is not None and HC_BATCH_FINISH)
            if rec is not None:
                rec.update(kind=kind, layer=l, dR_in=dR, dR_in_bcast=bcast, dbeta_in=dbeta, dR_out=h['dsum'] if prev is None else h['dR'], sum_only=prev is None,
                           dY_prev=h['dy'], dbeta_prev=h['dbeta'], hc_grads=dict(h['grads']) if h['grads'] is not None else None)
                TRACE.append(rec)
            dR, dY, dbeta, bcast = (h['dsum'] if prev is None else h['dR']), h['dy'], h['dbeta'], False     # first branch: summed over the streams (:524)
            if h['grads'] is None:
                hc_pending.append((first, ip, h['part'], rec))     # deferred mode: finished with the other branches (finish_hc)
            else:
                for j, k in enumerate(HC_KEYS):
                    grads[first + j] = h['grads'][k]
                grads[ip] = h['grads']['ln']
        else:
            dX, dgl = ops.layernorm_bwd(dXN, sv['R'], sv['mean'], sv['rstd'], prm['ln'], extra=extra)
            if rec is not None:
                rec.update(kind=kind, layer=l, dR_in=dR, dX=dX, dln=dgl)
                TRACE.append(rec)
            dR = ops.add_f32(dR, dX, dx_scale if prev is None else 1.0)
            grads[ip] = dgl
        sv.clear()
        if wg is not None and (prev is None or prev['layer'] != l) and l in starts:
            finish_hc()                                        # (before the group's hand-off: a gradient hook wants the whole group's gradients)
            launch_group(l)                                    # this layer closes a group: its operands (and the layers' above it) are complete
        if on_layer_grads is not None and wg is None and (prev is None or prev['layer'] != l):
            # (per-layer path) the bucket copy + all-reduce launch of this layer is ordered after its weight gradients ON THE SIDE STREAM: the
            # critical path never waits for them
            base = l * ppl
            side.run_after_all(lambda: on_layer_grads(l, grads[base:base + ppl]), *[g for g in grads[base:base + ppl] if g is not None])

    finish_hc()
    dx = dR.view(B, N, D)
    dtbl = ddense if dense else (ops.attn_bias_grad_reduce(dtbl_part, B, N, H) if bias is not None else None)
    side.join()                                    # autograd hands the gradients to consumers on the main stream
    return dx, grads, dtbl, dctx


_MICRO_STREAMS = {}


def _micro_stream(dev):
    """second HIP stream of the two-half-batch schedule (one per device)"""
    key = (dev.type, dev.index)
    if key not in _MICRO_STREAMS:
        _MICRO_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _MICRO_STREAMS[key]


def _halves(t, h):
    return (None, None) if t is None else (t[:h], t[h:])


class TransformerStackFn(torch.autograd.Function):
    """x fp32 [B,N,D] , key mask -> final-LayerNorm'd hidden states fp32 [B*N, D].

    opts['micro'] == 2 (even B, training / plain forward): the batch is processed as TWO independent half-batches on two HIP streams, forward
    and backward -- every kernel of the stack is per-sequence, so this changes no arithmetic.  The point is co-residency: while one half runs an
    MFMA-bound GEMM (HBM idle) the other half's HBM-bound row kernels (hyper-connections, LayerNorm, GEGLU: matrix cores idle) share the CUs.
    The host issues twice as many launches, so this schedule is meant to be captured into a hipGraph (graphed.GraphedTrainStep); autograd sees
    ONE node on ONE stream -- the fork / join is done here with events."""

    @staticmethod
    def forward(ctx, x, mask_u8, cfg, cache, opts, bias, tbl, context, *flat):
        # bias: relpos.AttnBias | None; tbl = bias.tbl passed separately so that autograd routes its gradient.
        # context: conditioning embeddings [B, m, dim_context] (cross-attention layers / self-attention prefix) | None; its key mask (bool / uint8
        #          [B, m]) travels in opts['context_mask'].
        # opts: dict(hook = per-layer gradient callback | None, grad = torch.is_grad_enabled() AT THE CALL SITE (it is always off in here),
        #            kv_out / decode = DecodeCache | None: sampling, micro = 1 | 2)
        hooks = opts.get('hook')
        need = bool(opts.get('grad', True)) and (any(t.requires_grad for t in flat) or x.requires_grad or (tbl is not None and tbl.requires_grad)
                                                 or (context is not None and context.requires_grad))
        xin = x.detach().contiguous().to(F32)
        bias = bias.detached() if bias is not None else None
        flat_d = flat                                            # (autograd is off inside Function.forward: the launches only take data pointers -- no per-parameter detach)
        B = xin.shape[0]
        cx = None
        if context is not None:
            m = context.shape[1]
            cm = opts.get('context_mask')
            cm = None if cm is None else cm.to(torch.bool).contiguous().view(torch.uint8)
            cx = Context(context.detach().to(BF16).reshape(B * m, -1).contiguous(), cm, B, m)
        ctx.ctx_meta = None if context is None else (context.shape, context.dtype)
        micro = int(opts.get('micro', 1))
        if micro == 2 and (B % 2 or B < 2 or opts.get('kv_out') is not None or opts.get('decode') is not None or hooks is not None):
            micro = 1
        ctx.micro = micro
        dseed = None
        if float(opts.get('attn_dropout', 0.)) > 0. and xin.is_cuda and torch.cuda.is_current_stream_capturing():
            seed_state = graph_seed_state(xin.device)
            seed_state.add_(2 * cfg.depth)                            # part of the capture: every replay moves on to unused mask streams
            # THIS call's snapshot (the copy is captured and re-run on every replay): a second stack forward inside the same capture -- two transformers
            # in one module, one transformer called twice, in-capture gradient accumulation -- advances the shared counter again before this call's
            # backward runs, and the backward must see the value its own forward used (dQ / dK / dV with other keep masks otherwise)
            dseed = (0, seed_state.clone())                           # (second half-batch: offset cfg.depth)
        if micro == 1:
            defer = DEFER_WGRAD and (hooks is None or DP_DEFER_GROUPS > 0)
            plan = None
            if (launchlist.ENABLED and hooks is None and cx is None and opts.get('kv_out') is None and opts.get('decode') is None and TRACE is None
                    and float(opts.get('ff_dropout', 0.)) == 0. and float(opts.get('attn_dropout', 0.)) == 0. and not isinstance(bias, relpos.DenseBias)
                    and xin.is_cuda and not torch.cuda.is_current_stream_capturing()):
                # plain training / evaluation call: the launch sequence is a function of (configuration, shape) alone -- sized, recorded, then re-issued
                # from one C call per pass (launchlist.py)
                plan = launchlist.plan_for(sys.modules[__name__], xin, mask_u8, cfg, need, bias, len(flat_d), defer, cfg.grad_shrink_alpha)
            if plan is not None:
                hn, saved = launchlist.forward(sys.modules[__name__], plan, xin, mask_u8, flat_d, cfg, cache, need, bias, defer)
            else:
                hn, saved = stack_forward(xin, mask_u8, flat_d, cfg, cache, need, bias, kv_out=opts.get('kv_out'), decode=opts.get('decode'), ctx=cx,
                                          ff_dropout=float(opts.get('ff_dropout', 0.)), attn_dropout=float(opts.get('attn_dropout', 0.)),
                                          defer_wgrad=defer, drop_seed=dseed)
        else:
            S, ppl, h = cfg.streams, params_per_layer(cfg.streams, cfg.cross_attend), B // 2
            for l in range(cfg.depth):                                   # pack the bf16 weight copies once, ahead of the fork
                layer_weights(cache, l, _split_layer(flat_d[l * ppl:(l + 1) * ppl], S, cfg.cross_attend), cfg.inner, cfg.inner_pad)
            cur, s2 = torch.cuda.current_stream(xin.device), _micro_stream(xin.device)
            (xa, xb), (ma, mb) = _halves(xin, h), _halves(mask_u8, h)
            ca = cb = None
            if cx is not None:
                (cma, cmb) = _halves(cx.mask, h)
                ca, cb = Context(cx.x[:h * cx.m], cma, h, cx.m), Context(cx.x[h * cx.m:], cmb, B - h, cx.m)
            s2.wait_stream(cur)
            adp = float(opts.get('attn_dropout', 0.))
            hna, sva = stack_forward(xa, ma, flat_d, cfg, cache, need, bias, ctx=ca, ff_dropout=float(opts.get('ff_dropout', 0.)), attn_dropout=adp, drop_seed=dseed)
            xb.record_stream(s2)
            with torch.cuda.stream(s2):
                hnb, svb = stack_forward(xb, mb, flat_d, cfg, cache, need, bias, ctx=cb, ff_dropout=float(opts.get('ff_dropout', 0.)), attn_dropout=adp,
                                         drop_seed=None if dseed is None else (cfg.depth, dseed[1]))
            cur.wait_stream(s2)
            hnb.record_stream(cur)
            hn = torch.cat((hna, hnb), dim=0)
            saved = (sva, svb)
        ctx.saved, ctx.cfg, ctx.cache, ctx.mask, ctx.hooks, ctx.bias = saved, cfg, cache, mask_u8, hooks, bias
        ctx.flat = flat
        return hn

    @staticmethod
    def backward(ctx, dhn):
        cfg = ctx.cfg
        flat = ctx.flat                                          # (grad mode is off in backward unless create_graph: no per-parameter detach)
        dhn = dhn.contiguous()
        if dhn.dtype not in (BF16, F32):
            dhn = dhn.to(F32)
        if ctx.micro == 1 and (isinstance(ctx.saved, launchlist.Replay) or '_ll' in ctx.saved):
            dx, grads, dtbl, dctx = launchlist.backward(sys.modules[__name__], dhn, ctx.mask, flat, cfg, ctx.cache, ctx.saved, ctx.bias, cfg.grad_shrink_alpha)
        elif ctx.micro == 1:
            dx, grads, dtbl, dctx = stack_backward(dhn, ctx.mask, flat, cfg, ctx.cache, ctx.saved, ctx.hooks, ctx.bias, dx_scale=cfg.grad_shrink_alpha)
        else:
            sva, svb = ctx.saved
            h = sva['B']
            rows = h * sva['N']
            (ma, mb) = _halves(ctx.mask, h)
            cur, s2 = torch.cuda.current_stream(dhn.device), _micro_stream(dhn.device)
            da, db = dhn[:rows], dhn[rows:]
            s2.wait_stream(cur)
            # no weight-gradient side streams here: the other half already fills the idle CUs, and nested stream forks do not survive
            # hipStreamEndCapture on ROCm 7.0 (segmentation fault)
            dxa, ga, ta, ca = stack_backward(da, ma, flat, cfg, ctx.cache, sva, None, ctx.bias, async_wgrad=MICRO_ASYNC_WGRAD, dx_scale=cfg.grad_shrink_alpha)
            db.record_stream(s2)
            with torch.cuda.stream(s2):
                dxb, gb, tb, cb = stack_backward(db, mb, flat, cfg, ctx.cache, svb, None, ctx.bias, async_wgrad=MICRO_ASYNC_WGRAD, dx_scale=cfg.grad_shrink_alpha)
            cur.wait_stream(s2)
            for t in [dxb, tb, cb] + gb:
                if t is not None:
                    t.record_stream(cur)
            dx = torch.cat((dxa, dxb), dim=0)
            pa = [a for a, b in zip(ga, gb) if a is not None and b is not None]
            pb = [b for a, b in zip(ga, gb) if a is not None and b is not None]
            if pa:
                torch._foreach_add_(pa, pb)
            grads = [a if a is not None else b for a, b in zip(ga, gb)]
            dtbl = None if ta is None else ta + tb
            dctx = None if ca is None else torch.cat((ca, cb), dim=0)
        ctx.saved = None                                                      # (grad_shrink, audiolm_pytorch.py:93-94, :478: applied inside stack_backward)
        out = []
        for p, g in zip(ctx.flat, grads):
            out.append((g if g.shape == p.shape else g.reshape(p.shape)) if (g is not None and p.requires_grad) else None)
        dcontext = None
        if dctx is not None and ctx.ctx_meta is not None:
            dcontext = dctx.reshape(ctx.ctx_meta[0]).to(ctx.ctx_meta[1])
        return (dx, None, None, None, None, None, dtbl, dcontext, *out)
