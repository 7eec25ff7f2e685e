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
from dataclasses import dataclass

import torch

from . import ops

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

    @property
    def inner_pad(self):
        return (self.inner + 7) // 8 * 8


HC_KEYS = ('Bb', 'Aa', 'Wa', 'sa', 'wb', 'sb', 'gamma')   # static_beta, static_alpha, dynamic_alpha_fn, dynamic_alpha_scale, dynamic_beta_fn, dynamic_beta_scale, norm.gamma


def params_per_layer(S):
    hc = len(HC_KEYS) if S > 1 else 0
    return (hc + 4) + (hc + 4)


def _split_layer(flat, S):
    """flat per-layer parameter list -> (attn dict, ff dict) (see Transformer.flat_params for the order)."""
    i = 0
    a, f = {}, {}
    if S > 1:
        a['hc'] = dict(zip(HC_KEYS, flat[i:i + 7])); i += 7
    a['ln'], a['wq'], a['wkv'], a['wo'] = flat[i:i + 4]; i += 4
    if S > 1:
        f['hc'] = dict(zip(HC_KEYS, flat[i:i + 7])); i += 7
    f['ln'], f['w1'], f['ln3'], f['w2'] = flat[i:i + 4]
    return a, f


def tensor_version(t):
    """in-place-update counter used as the cache key of packed weights; inference tensors (created under torch.inference_mode) have none
    and cannot be updated in place: constant 0"""
    try:
        return t._version
    except RuntimeError:
        return 0


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


def layer_weights(cache: WeightCache, l, pa, pf, I, Ip):
    """bf16 packed (W, W^T) pairs of one layer's five dense weights; when any master changed, ALL are re-packed in one launch."""
    ws = (('wq', pa['wq']), ('wkv', pa['wkv']), ('wo', pa['wo']), ('w1', pf['w1']), ('w2', pf['w2']))
    vers = {k: (w.data_ptr(), tensor_version(w), tuple(w.shape)) for k, w in ws}
    hits = {k: cache.store.get((l, k)) for k, _ in ws}
    if all(h is not None and h[0] == vers[k] for k, h in hits.items()):
        return tuple(hits[k][1] for k, _ in ws)
    out, jobs = {}, []
    with torch.no_grad():
        for k, w in ws[:3]:
            w = w.detach()
            rows, cols = w.shape
            rp, cp = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
            W = torch.empty((rp, cp), dtype=BF16, device=w.device)
            WT = torch.empty((cp, rp), dtype=BF16, device=w.device)
            jobs.append((w, W, WT, rp, cp))
            out[k] = (W[:rows], WT[:cols])
        w1 = pf['w1'].detach()
        D = w1.shape[1]
        W1 = torch.empty((2 * Ip, D), dtype=BF16, device=w1.device)
        W1T = torch.empty((D, 2 * Ip), dtype=BF16, device=w1.device)
        jobs.append((w1[:I], W1[:Ip], W1T[:, :Ip], Ip, D))
        jobs.append((w1[I:], W1[Ip:], W1T[:, Ip:], Ip, D))
        out['w1'] = (W1, W1T)
        w2 = pf['w2'].detach()
        W2 = torch.empty((D, Ip), dtype=BF16, device=w2.device)
        W2T = torch.empty((Ip, D), dtype=BF16, device=w2.device)
        jobs.append((w2, W2, W2T, D, Ip))
        out['w2'] = (W2, W2T)
        ops.pack_weights_multi(jobs)
    for k, _ in ws:
        cache.store[(l, k)] = (vers[k], out[k])
    return tuple(out[k] for k, _ in ws)


def default_residual_bf16():
    """ALM_RESIDUAL_DTYPE = fp32 | bf16: storage of the hyper-connection residual streams (see StackCfg.residual_bf16)"""
    v = os.environ.get('ALM_RESIDUAL_DTYPE', 'fp32').lower()
    if v not in ('fp32', 'float32', 'bf16', 'bfloat16'):
        raise ValueError(f'ALM_RESIDUAL_DTYPE={v!r}: expected fp32 or bf16')
    return v in ('bf16', 'bfloat16')


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


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
        self.frozen = False             # True while a step is being CAPTURED (recorded, not executed): host bookkeeping must not advance


def stack_forward(x, mask_u8, flat, cfg: StackCfg, cache: WeightCache, need_grad: bool, bias=None, kv_out=None, decode=None):
    """x fp32 [B, N, D] -> (hn fp32 [B*N, D], saved-for-backward | None).  bias: relpos.AttnBias (structured score bias shared by every layer,
    audiolm_pytorch.py:500-506 / :532) or None.  Sampling: `kv_out` (DecodeCache) is filled with every layer's k / v of this (prefix)
    forward; `decode` (DecodeCache) means x holds ONE new position per sequence (N == 1) at index decode.length: its attention runs over
    the cache (alm_mqa_decode_attn, which also appends the new k / v), everything else is the same launch sequence on B rows."""
    B, N, D = x.shape
    M, S, H, dh = B * N, cfg.streams, cfg.heads, cfg.dim_head
    I, Ip = cfg.inner, cfg.inner_pad
    dev = x.device
    ppl = params_per_layer(S)
    saved = dict(layers=[], B=B, N=N) if need_grad else None

    R = x.reshape(M, D)                  # S > 1: the stream expansion (:524) is never materialised -- the first branch reads x for every stream
    rb = S > 1
    rdt = BF16 if (cfg.residual_bf16 and S > 1) else F32
    kv0 = None
    pend_y = pend_coef = None            # S > 1: branch output + coefficient record whose depth connection is still to be applied
    for l in range(cfg.depth):
        pa, pf = _split_layer(flat[l * ppl:(l + 1) * ppl], S)
        (Wq, WqT), (Wkv, WkvT), (Wo, WoT), (W1, W1T), (W2, W2T) = layer_weights(cache, l, pa, pf, I, Ip)

        # ---------------- attention branch (audiolm_pytorch.py:307-406) ----------------
        if S > 1:
            # depth connection of the previous branch fused with this branch's width connection + pre-LayerNorm (one pass over R)
            h = ops.hc_fwd(R, B, S, N, D, y_prev=pend_y, coef_prev=pend_coef, hc=pa['hc'], ln_gamma=pa['ln'], rin_bcast=rb, r_dtype=rdt)
            R, X, XN, mean, rstd, coef = h['R'], h['x'], h['xn'], h['mean'], h['rstd'], h['coef']
            rb = rb and pend_y is None                 # still the un-expanded x after a width-only call
        else:
            XN, X, mean, rstd = ops.layernorm_fwd(R, pa['ln'], want_copy=True)
            coef = None
        Q = _empty((M, H * dh), BF16, dev)
        ops.gemm_nt(XN, Wq, Q)
        KV = _empty((M, 2 * dh), BF16, dev)
        ops.gemm_nt(X, Wkv, KV)                          # k / v from the UN-normalised branch input (:325 vs :347)
        K, Vown = KV[:, :dh], KV[:, dh:]
        if cfg.add_value_residual and kv0 is not None:
            V = ops.value_residual_mix(Vown, kv0[:, dh:])  # :357-358
        else:
            V = Vown
        if kv0 is None:
            kv0 = KV                                      # :534-535 (layer-0 values, pre-mix)
        if decode is not None:
            kv_new = KV if V is Vown else torch.cat((K, V), dim=1)       # k | value-residual-mixed v of the new position
            AO, LSE = ops.mqa_decode_attn(Q, decode.kv[l], kv_new, decode.length, mask_u8, H, dh, bias=bias, pos_dev=decode.pos_dev), None
        else:
            AO, LSE = ops.mqa_attn_fwd(Q, K, V, mask_u8, B, N, H, dh, bias=bias)
            if kv_out is not None:
                kv_out.kv[l][:, :N, :dh] = K.reshape(B, N, dh)
                kv_out.kv[l][:, :N, dh:] = V.reshape(B, N, dh)
        Y = _empty((M, D), BF16, dev)
        ops.gemm_nt(AO, Wo, Y)

        # ---------------- feed-forward branch (audiolm_pytorch.py:246-260) ----------------
        if S > 1:
            # (the un-normalised branch input X2 is only read by the un-fused LayerNorm backward)
            h = ops.hc_fwd(R, B, S, N, D, y_prev=Y, coef_prev=coef, hc=pf['hc'], ln_gamma=pf['ln'], rin_bcast=rb, r_dtype=rdt,
                           want_x=need_grad and not FUSE_LN_BWD)
            R1, X2, XN2, mean2, rstd2, coef2 = h['R'], h['x'], h['xn'], h['mean'], h['rstd'], h['coef']
            rb = False
        else:
            R1 = ops.residual_add(R, Y)
            XN2, X2, mean2, rstd2 = ops.layernorm_fwd(R1, pf['ln'], want_copy=False)
            coef2 = None
        U = _empty((M, 2 * Ip), BF16, dev)
        ops.gemm_nt(XN2, W1, U)
        HN, mean3, rstd3 = ops.geglu_ln_fwd(U, pf['ln3'], I, Ip)
        Y2 = _empty((M, D), BF16, dev)
        ops.gemm_nt(HN, W2, Y2)

        if need_grad:
            saved['layers'].append(dict(R=R, X=X, XN=XN, mean=mean, rstd=rstd, coef=coef, Q=Q, KV=KV, V=V, AO=AO, LSE=LSE, Y=Y,
                                        R1=R1, X2=X2, XN2=XN2, mean2=mean2, rstd2=rstd2, coef2=coef2, U=U, HN=HN, mean3=mean3,
                                        rstd3=rstd3, Y2=Y2, mixed=V is not Vown, r_bcast=(S > 1 and l == 0)))
        if S > 1:
            R, pend_y, pend_coef = R1, Y2, coef2
        else:
            R = ops.residual_add(R1, Y2)

    if S > 1:
        # last depth connection + stream sum (:551) + final LayerNorm (:555) in one pass; the final residual streams are never stored
        h = ops.hc_fwd(R, B, S, N, D, y_prev=pend_y, coef_prev=pend_coef, ln_gamma=flat[-1], final=True, r_dtype=rdt, final_f32=True)
        xs, hn, fmean, frstd = h['xs'], h['xn'], h['mean'], h['rstd']
    else:
        xs = R
        hn, _, fmean, frstd = ops.layernorm_fwd(xs, flat[-1], out_f32=True)  # :555 (fp32 out: F.layer_norm autocasts to fp32; the logit heads split it)
    if need_grad:
        saved.update(xs=xs, fmean=fmean, frstd=frstd, kv0=kv0)
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


FUSE_LN_BWD = os.environ.get('ALM_FUSE_LN_BWD', '1') != '0'             # switch: pre-LayerNorm backward inside the hyper-connection kernel
ASYNC_WGRAD = os.environ.get('ALM_ASYNC_WGRAD', '1') != '0'          # switch (ALM_ASYNC_WGRAD=0 turns the side stream off: A/B runs)
MICRO_ASYNC_WGRAD = os.environ.get('ALM_MICRO_ASYNC_WGRAD', '0') != '0'   # weight-gradient side streams inside the two-half-batch schedule
SIDE_STREAMS = max(1, int(os.environ.get('ALM_SIDE_STREAMS', '1')))    # number of side streams the weight-gradient GEMMs are dealt over


def stack_backward(dhn, mask_u8, flat, cfg: StackCfg, cache: WeightCache, saved, on_layer_grads=None, bias=None, async_wgrad=True):
    """dhn fp32 | bf16 [M, D] -> (dx fp32 [B, N, D], list of parameter grads aligned with `flat`, d(loss)/d(bias.tbl) | None)."""
    B, N = saved['B'], saved['N']
    D, S, H, dh = cfg.dim, cfg.streams, cfg.heads, cfg.dim_head
    M = B * N
    I, Ip = cfg.inner, cfg.inner_pad
    dev = dhn.device
    ppl = params_per_layer(S)
    grads = [None] * len(flat)
    side = _SideStream(dev, ASYNC_WGRAD and async_wgrad)
    rdt = BF16 if (cfg.residual_bf16 and S > 1) else F32

    dxs, dgam = ops.layernorm_bwd(dhn, saved['xs'], saved['fmean'], saved['frstd'], flat[-1])
    grads[-1] = dgam
    acc_v0 = torch.zeros((M, dh), dtype=F32, device=dev) if cfg.add_value_residual and cfg.depth > 1 else None
    # S > 1: dR = gradient wrt the residual streams after the current branch; right after the final stream sum it is dxs for every
    # stream (`bcast`).  dY2 / dbeta2 (depth-connection backward of the FF branch) are produced one step ahead by the fused kernels.
    dtbl_part = ops.attn_bias_part(B, N, H, bias.tbl.shape[1], dev) if bias is not None else None
    dR, bcast = dxs, S > 1
    dY2 = dbeta2 = None
    if S > 1:
        last = saved['layers'][-1]
        h = ops.hc_bwd(dxs, B, S, N, D, bcast=True, y_prev=last['Y2'], coef_prev=last['coef2'], r_dtype=rdt)
        dY2, dbeta2 = h['dy'], h['dbeta']

    for l in reversed(range(cfg.depth)):
        sv = saved['layers'][l]
        base = l * ppl
        pa, pf = _split_layer(flat[base:base + ppl], S)
        (Wq, WqT), (Wkv, WkvT), (Wo, WoT), (W1, W1T), (W2, W2T) = layer_weights(cache, l, pa, pf, I, Ip)
        hc_n = 7 if S > 1 else 0
        ia = base + hc_n                      # index of attn ln gamma
        iff = base + hc_n + 4 + hc_n          # index of ff ln gamma

        # ================= feed-forward branch =================
        if S == 1:
            dY2 = ops.f32_to_bf16(dR)
        dHN = _empty((M, Ip), BF16, dev)
        ops.gemm_nt(dY2, W2T, dHN)                                            # dHN = dY2 @ W2
        dW2 = _empty((D, I), F32, dev)
        HNs = sv['HN']
        side.run(lambda: ops.gemm_tn_splitk(dY2, HNs[:, :I], dW2), dY2, HNs, dW2)      # dW2 = dY2^T @ HN
        dU, dg3 = ops.geglu_ln_bwd(dHN, sv['U'], pf['ln3'], sv['mean3'], sv['rstd3'], I, Ip)
        dXN2 = _empty((M, D), BF16, dev)
        ops.gemm_nt(dU, W1T, dXN2)                                            # dXN2 = dU @ W1
        dW1 = _empty((2 * I, D), F32, dev)
        XN2s = sv['XN2']
        side.run(lambda: ops.gemm_tn_splitk(dU.view(M, 2, Ip).permute(1, 0, 2)[:, :, :I], XN2s, dW1.view(2, I, D)), dU, XN2s, dW1)   # dW1 = dU^T @ XN2
        if S > 1:
            # the FF branch's pre-LayerNorm backward + its width-connection backward + the depth-connection backward of this layer's
            # attention branch: ONE pass over the residual streams
            if FUSE_LN_BWD:
                h = ops.hc_bwd(dR, B, S, N, D, bcast=bcast, dxn=dXN2, mean=sv['mean2'], rstd=sv['rstd2'], ln_gamma=pf['ln'], R=sv['R1'],
                               coef=sv['coef2'], dbeta=dbeta2, hc=pf['hc'], y_prev=sv['Y'], coef_prev=sv['coef'], r_dtype=rdt)
                dgl = h['grads']['ln']
            else:
                dX2, dgl = ops.layernorm_bwd(dXN2, sv['X2'], sv['mean2'], sv['rstd2'], pf['ln'])
                h = ops.hc_bwd(dR, B, S, N, D, bcast=bcast, dx=dX2, R=sv['R1'], coef=sv['coef2'], dbeta=dbeta2, hc=pf['hc'],
                               y_prev=sv['Y'], coef_prev=sv['coef'], r_dtype=rdt)
            dR1, dY, dbeta, bcast = h['dR'], h['dy'], h['dbeta'], False
            for j, k in enumerate(HC_KEYS):
                grads[base + hc_n + 4 + j] = h['grads'][k]
        else:
            dX2, dgl = ops.layernorm_bwd(dXN2, sv['R1'], sv['mean2'], sv['rstd2'], pf['ln'])
            dR1 = ops.add_f32(dR, dX2)
            dY = ops.f32_to_bf16(dR1)
        grads[iff], grads[iff + 1], grads[iff + 2], grads[iff + 3] = dgl, dW1, dg3, dW2

        # ================= attention branch =================
        dAO = _empty((M, H * dh), BF16, dev)
        ops.gemm_nt(dY, WoT, dAO)
        dWo = _empty((D, H * dh), F32, dev)
        AOs = sv['AO']
        side.run(lambda: ops.gemm_tn_splitk(dY, AOs, dWo), dY, AOs, dWo)
        KV = sv['KV']
        dQ, dkv32 = ops.mqa_attn_bwd(sv['Q'], KV[:, :dh], sv['V'], mask_u8, sv['AO'], sv['LSE'], dAO, B, N, H, dh, bias=bias, dtbl_part=dtbl_part)
        if acc_v0 is None:
            mode = 0
        elif sv['mixed']:
            mode = 1
        else:
            mode = 2                                                          # layer 0: receives every later layer's 0.5 * dV
        dKV = ops.kv_grad_pack(dkv32, acc_v0, mode, dh)
        dXN = _empty((M, D), BF16, dev)
        ops.gemm_nt(dQ, WqT, dXN)
        dWq = _empty((H * dh, D), F32, dev)
        XNs = sv['XN']
        side.run(lambda: ops.gemm_tn_splitk(dQ, XNs, dWq), dQ, XNs, dWq)
        dXkv = _empty((M, D), BF16, dev)
        ops.gemm_nt(dKV, WkvT, dXkv)
        dWkv = _empty((2 * dh, D), F32, dev)
        Xs = sv['X']
        side.run(lambda: ops.gemm_tn_splitk(dKV, Xs, dWkv), dKV, Xs, dWkv)
        if S > 1:
            # pre-LayerNorm backward (+ the K/V-path gradient dXkv, which reaches the un-normalised branch input directly) + width-connection
            # backward of the attention branch (+ depth-connection backward of the previous layer's FF branch)
            prev = saved['layers'][l - 1] if l > 0 else None
            py, pc = (prev['Y2'], prev['coef2']) if prev else (None, None)
            if FUSE_LN_BWD:
                h = ops.hc_bwd(dR1, B, S, N, D, dxn=dXN, extra=dXkv, mean=sv['mean'], rstd=sv['rstd'], ln_gamma=pa['ln'], R=sv['R'],
                               coef=sv['coef'], dbeta=dbeta, hc=pa['hc'], y_prev=py, coef_prev=pc, r_bcast=sv['r_bcast'], sum_only=l == 0, r_dtype=rdt)
                dgla = h['grads']['ln']
            else:
                dX, dgla = ops.layernorm_bwd(dXN, sv['X'], sv['mean'], sv['rstd'], pa['ln'], extra=dXkv)
                h = ops.hc_bwd(dR1, B, S, N, D, dx=dX, R=sv['R'], coef=sv['coef'], dbeta=dbeta, hc=pa['hc'], y_prev=py, coef_prev=pc,
                               r_bcast=sv['r_bcast'], sum_only=l == 0, r_dtype=rdt)
            dR, dY2, dbeta2 = (h['dsum'] if l == 0 else h['dR']), h['dy'], h['dbeta']      # layer 0: already summed over the streams (:524)
            for j, k in enumerate(HC_KEYS):
                grads[base + j] = h['grads'][k]
        else:
            dX, dgla = ops.layernorm_bwd(dXN, sv['R'], sv['mean'], sv['rstd'], pa['ln'], extra=dXkv)
            dR = ops.add_f32(dR1, dX)
        grads[ia], grads[ia + 1], grads[ia + 2], grads[ia + 3] = dgla, dWq, dWkv, dWo
        sv.clear()
        if on_layer_grads is not None:
            # the bucket copy + all-reduce launch of this layer is ordered after its weight gradients ON THE SIDE STREAM: the critical
            # path never waits for them
            side.run_after_all(lambda: on_layer_grads(l, grads[base:base + ppl]), *[g for g in grads[base:base + ppl] if g is not None])

    dx = dR.view(B, N, D)
    dtbl = ops.attn_bias_grad_reduce(dtbl_part, B, N, H) if bias is not None else None
    side.join()                                    # autograd hands the gradients to consumers on the main stream
    return dx, grads, dtbl


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
    def forward(ctx, x, mask_u8, cfg, cache, opts, bias, tbl, *flat):
        # bias: relpos.AttnBias | None; tbl = bias.tbl passed separately so that autograd routes its gradient.
        # opts: dict(hook = per-layer gradient callback | None, grad = torch.is_grad_enabled() AT THE CALL SITE (it is always off in here),
        #            kv_out / decode = DecodeCache | None: sampling, micro = 1 | 2)
        hooks = opts.get('hook')
        need = bool(opts.get('grad', True)) and (any(t.requires_grad for t in flat) or x.requires_grad or (tbl is not None and tbl.requires_grad))
        xin = x.detach().contiguous().to(F32)
        bias = bias.detached() if bias is not None else None
        flat_d = [t.detach() for t in flat]
        B = xin.shape[0]
        micro = int(opts.get('micro', 1))
        if micro == 2 and (B % 2 or B < 2 or opts.get('kv_out') is not None or opts.get('decode') is not None or hooks is not None):
            micro = 1
        ctx.micro = micro
        if micro == 1:
            hn, saved = stack_forward(xin, mask_u8, flat_d, cfg, cache, need, bias, kv_out=opts.get('kv_out'), decode=opts.get('decode'))
        else:
            S, ppl, h = cfg.streams, params_per_layer(cfg.streams), B // 2
            for l in range(cfg.depth):                                   # pack the bf16 weight copies once, ahead of the fork
                pa, pf = _split_layer(flat_d[l * ppl:(l + 1) * ppl], S)
                layer_weights(cache, l, pa, pf, cfg.inner, cfg.inner_pad)
            cur, s2 = torch.cuda.current_stream(xin.device), _micro_stream(xin.device)
            (xa, xb), (ma, mb) = _halves(xin, h), _halves(mask_u8, h)
            s2.wait_stream(cur)
            hna, sva = stack_forward(xa, ma, flat_d, cfg, cache, need, bias)
            xb.record_stream(s2)
            with torch.cuda.stream(s2):
                hnb, svb = stack_forward(xb, mb, flat_d, cfg, cache, need, bias)
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
        flat = [t.detach() for t in ctx.flat]
        dhn = dhn.contiguous()
        if dhn.dtype not in (BF16, F32):
            dhn = dhn.to(F32)
        if ctx.micro == 1:
            dx, grads, dtbl = stack_backward(dhn, ctx.mask, flat, cfg, ctx.cache, ctx.saved, ctx.hooks, ctx.bias)
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
            dxa, ga, ta = stack_backward(da, ma, flat, cfg, ctx.cache, sva, None, ctx.bias, async_wgrad=MICRO_ASYNC_WGRAD)
            db.record_stream(s2)
            with torch.cuda.stream(s2):
                dxb, gb, tb = stack_backward(db, mb, flat, cfg, ctx.cache, svb, None, ctx.bias, async_wgrad=MICRO_ASYNC_WGRAD)
            cur.wait_stream(s2)
            for t in [dxb, tb] + gb:
                if t is not None:
                    t.record_stream(cur)
            dx = torch.cat((dxa, dxb), dim=0)
            pa = [a for a, b in zip(ga, gb) if a is not None and b is not None]
            pb = [b for a, b in zip(ga, gb) if a is not None and b is not None]
            if pa:
                torch._foreach_add_(pa, pb)
            grads = [a if a is not None else b for a, b in zip(ga, gb)]
            dtbl = None if ta is None else ta + tb
        ctx.saved = None
        dx = dx * cfg.grad_shrink_alpha                                       # grad_shrink, audiolm_pytorch.py:93-94, :478
        out = []
        for p, g in zip(ctx.flat, grads):
            out.append(g.reshape(p.shape) if (g is not None and p.requires_grad) else None)
        return (dx, None, None, None, None, None, dtbl, *out)
