"""Attention dropout (round 3: the last refused constructor argument) and the standalone Attention / FeedForward / GEGLU forwards, on a real MI355X.

Reference: Attend(dropout) attend.py:36-46, :92 (SDPA dropout_p), :140 (attn_dropout(attn)); Attention(dropout) audiolm_pytorch.py:288-304 (the same
p drives the probability dropout inside Attend AND the nn.Dropout behind to_out); Attention.forward :307-406; FeedForward :251-260; GEGLU :246-249.

The flash kernels decide keep / drop per (batch, head, query, key) with a stateless integer hash of a per-call seed (csrc/attention.hip `drop_keep`);
this file restates that hash in numpy, hands the resulting masks to the oracle (`oracle.attend(keep=..., p_drop=...)`), and compares outputs and every
gradient -- like the ff_dropout test, which does the same for drawn masks."""
import numpy as np
import pytest
import torch

import audiolm_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dev())


def fro(a, w):
    a, w = a.detach().cpu().double(), w.detach().cpu().double()
    return float((a - w).norm() / w.norm().clamp(min=1e-30))


def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def keep_mask(seed, B, H, N, p):
    """restatement of csrc/attention.hip drop_salt / drop_keep -> bool (B, H, N, N): pair (i, j) of (b, h) is kept iff mix32((i N + j) ^ salt) >= p 2^32"""
    lo, hi = np.uint32(seed & 0xffffffff), np.uint32((seed >> 32) & 0xffffffff)
    thr = np.uint32(min(int(p * 4294967296.0), 4294967295))
    out = np.empty((B, H, N, N), dtype=bool)
    idx = (np.arange(N, dtype=np.uint64)[:, None] * np.uint64(N) + np.arange(N, dtype=np.uint64)[None, :]).astype(np.uint32)
    with np.errstate(over='ignore'):
        for b in range(B):
            for h in range(H):
                bh = np.array([(b * H + h)], dtype=np.uint64)
                inner = _mix32(((np.uint64(hi) + bh * np.uint64(0x9E3779B9)) & np.uint64(0xffffffff)).astype(np.uint32))
                salt = _mix32(np.array([lo], dtype=np.uint32) ^ inner)[0]
                out[b, h] = _mix32(idx ^ salt) >= thr
    return torch.from_numpy(out)


@pytest.mark.parametrize('N,H', [(200, 8), (64, 4), (333, 6)])
def test_flash_attention_dropout_matches_oracle_with_the_restated_masks(N, H, monkeypatch):
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import core
    B, dh, p, seed = 2, 64, 0.2, 0x1234_5678_9abc_def
    monkeypatch.setattr(core, '_attn_seed', lambda: seed)
    att = A.attend.Attend(dropout=p, causal=True, flash=True).to(dev())
    att.train()
    q = rnd(B, H, N, dh, seed=1, scale=0.7).requires_grad_(True)
    k = rnd(B, N, dh, seed=2, scale=0.7).requires_grad_(True)
    v = rnd(B, N, dh, seed=3).requires_grad_(True)
    mask = torch.rand(B, N, generator=torch.Generator().manual_seed(4)) > 0.15
    mask[:, 0] = True
    out = att(q, k, v, mask=mask.to(dev()))
    go = rnd(B, H, N, dh, seed=5)
    out.backward(go)
    keep = keep_mask(seed, B, H, N, p)
    tri = torch.ones(N, N, dtype=torch.bool).tril()
    frac = float(keep[:, :, tri].float().mean())
    assert abs(frac - (1 - p)) < 0.01, frac                                            # the hash really is ~Bernoulli(1 - p) ...
    assert float((keep[0, 0] == keep[0, 1]).float().mean()) < 0.75                     # ... and differs between heads
    qr, kr, vr = (t.detach().cpu().to(torch.bfloat16).float().requires_grad_(True) for t in (q, k, v))
    ref = O.attend(qr, kr, vr, mask=mask, keep=keep.float(), p_drop=p)
    ref.backward(go.cpu())
    assert fro(out, ref) <= 6e-3, fro(out, ref)
    assert fro(q.grad, qr.grad) <= 1.5e-2 and fro(k.grad, kr.grad) <= 1.5e-2 and fro(v.grad, vr.grad) <= 1.5e-2, (fro(q.grad, qr.grad), fro(k.grad, kr.grad), fro(v.grad, vr.grad))
    ref0 = O.attend(qr.detach(), kr.detach(), vr.detach(), mask=mask)
    assert fro(out, ref0) > 0.1                                                         # without the masks the result is clearly different
    att.eval()                                                                          # eval(): no dropout (attend.py:92 `if self.training`)
    with torch.no_grad():
        oe = att(q.detach(), k.detach(), v.detach(), mask=mask.to(dev()))
    assert fro(oe, ref0) <= 6e-3


def test_math_path_attention_dropout_matches_oracle_with_the_drawn_masks(monkeypatch):
    """non-causal / dense-bias Attend (the O(n^2) math path, csrc/xattn.hip): the keep mask is drawn by xattn._keep_mask; pinned and handed to the oracle"""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import xattn
    B, H, N, M, dh, p = 2, 4, 50, 37, 64, 0.3
    gm = torch.Generator().manual_seed(11)
    drawn = []

    def seeded(shape, p_, device):
        kp = (torch.rand(shape, generator=gm) >= p_).to(torch.bfloat16)
        drawn.append(kp)
        return kp.to(device)
    monkeypatch.setattr(xattn, '_keep_mask', seeded)
    att = A.attend.Attend(dropout=p, causal=False, flash=False).to(dev())
    att.train()
    q = rnd(B, H, N, dh, seed=21, scale=0.7).requires_grad_(True)
    k = rnd(B, M, dh, seed=22, scale=0.7).requires_grad_(True)
    v = rnd(B, M, dh, seed=23).requires_grad_(True)
    bias = rnd(H, N, M, seed=24, scale=0.5)
    out = att(q, k, v, attn_bias=bias)
    go = rnd(B, H, N, dh, seed=25)
    out.backward(go)
    assert len(drawn) == 1
    keep = drawn[0].float().reshape(B, N, H, -1)[..., :M].permute(0, 2, 1, 3)           # rows of P are (b, n, h): -> (b h n m)
    qr, kr, vr = (t.detach().cpu().to(torch.bfloat16).float().requires_grad_(True) for t in (q, k, v))
    ref = O.attend(qr, kr, vr, attn_bias=bias.cpu(), causal=False, keep=keep, p_drop=p)
    ref.backward(go.cpu())
    assert fro(out, ref) <= 8e-3, fro(out, ref)
    assert fro(q.grad, qr.grad) <= 2e-2 and fro(k.grad, kr.grad) <= 2e-2 and fro(v.grad, vr.grad) <= 2e-2


@pytest.mark.parametrize('streams', [1, 4])
def test_transformer_attn_dropout_vs_oracle_with_the_same_masks(streams, monkeypatch):
    """Transformer(attn_dropout = p), training mode: per layer the flash kernels' hashed probability masks (stream seed + layer) and the drawn to_out
    mask; eval() is the dropout-free model"""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import core
    from common import synth_state_dict
    from test_gpu_parity import HC_SCALARS
    dim, depth, heads, n, b, p, seed = 128, 2, 4, 70, 2, 0.2, 777_000_111_222
    torch.manual_seed(0)
    tr = A.audiolm_pytorch.Transformer(dim=dim, depth=depth, heads=heads, num_residual_streams=streams, rel_pos_bias=False, flash_attn=True, attn_dropout=p)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in tr.state_dict().items()}, 191)
    tr.load_state_dict(sd)
    tr.to(dev())
    x = rnd(b, n, dim, seed=192).requires_grad_(True)
    gm = torch.Generator().manual_seed(193)
    mask = (torch.rand(b, n, generator=gm) > 0.15)
    mask[:, 0] = True
    omasks = []

    def seeded_keep(shape, p_, device):
        kp = (torch.rand(shape, generator=gm) >= p_).to(torch.bfloat16)
        omasks.append(kp)
        return kp.to(device)
    monkeypatch.setattr(core, '_dropout_keep', seeded_keep)
    monkeypatch.setattr(core, '_attn_seed', lambda: seed)
    tr.train()
    out = tr(x, self_attn_mask=mask.to(dev()))
    go = rnd(b, n, dim, seed=194)
    out.backward(go)
    assert len(omasks) == depth
    akeep = [keep_mask(seed + l, b, heads, n, p).float() for l in range(depth)]
    okeep = [m.float().reshape(b, n, dim) for m in omasks]
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    ref = O.transformer(sdr, '', xr, depth=depth, heads=heads, streams=streams, self_attn_mask=mask, attn_keep=akeep, out_keep=okeep, attn_dropout=p)
    ref.backward(go.cpu())
    assert fro(out, ref) <= 1.5e-2, fro(out, ref)
    assert fro(x.grad, xr.grad) <= 5e-2, fro(x.grad, xr.grad)
    worst = max((fro(q.grad, sdr[k].grad), k) for k, q in tr.named_parameters()
                if sdr[k].grad is not None and float(sdr[k].grad.norm()) > 1e-7 and not k.endswith(HC_SCALARS))
    assert worst[0] <= 8e-2, worst
    ref0 = O.transformer({k: v.detach() for k, v in sd.items()}, '', x.detach().cpu(), depth=depth, heads=heads, streams=streams, self_attn_mask=mask)
    assert fro(out, ref0) > 5e-2
    tr.eval()
    with torch.no_grad():
        oe = tr(x.detach(), self_attn_mask=mask.to(dev()))
    assert len(omasks) == depth and fro(oe, ref0) <= 1.5e-2


def test_standalone_attention_feedforward_geglu_match_the_oracle():
    """Attention.forward / FeedForward / GEGLU called as modules OUTSIDE Transformer (SURVEY §8(b) lists both as boundary constructors): the reference's
    forward signatures and return conventions, on the same kernels, un-fused"""
    import audiolm_pytorch_amd as A
    from common import synth_state_dict
    AP = A.audiolm_pytorch
    b, n, dim, heads = 2, 90, 128, 4
    # --- causal self-attention with a key mask, value residual in and out (reference :307-406)
    att = AP.Attention(dim=dim, heads=heads, causal=True, dropout=0., flash=True)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in att.state_dict().items()}, 301)
    att.load_state_dict(sd)
    att.to(dev()).eval()
    x = rnd(b, n, dim, seed=302).requires_grad_(True)
    vres = rnd(b, n, 64, seed=303)
    mask = torch.rand(b, n, generator=torch.Generator().manual_seed(304)) > 0.2
    mask[:, 0] = True
    out, orig_v = att(x, mask=mask.to(dev()), value_residual=vres, return_values=True)
    go = rnd(b, n, dim, seed=305)
    out.backward(go)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    ref, ref_v = O.attention(sdr, '', xr, heads, mask=mask, value_residual=vres.cpu())
    ref.backward(go.cpu())
    assert fro(out, ref) <= 1.5e-2 and fro(orig_v, ref_v) <= 1e-2, (fro(out, ref), fro(orig_v, ref_v))
    assert fro(x.grad, xr.grad) <= 3e-2
    for k_, q_ in att.named_parameters():
        if sdr[k_].grad is not None and float(sdr[k_].grad.norm()) > 1e-7:
            assert fro(q_.grad, sdr[k_].grad) <= 3e-2, k_
    # --- cross-attention: context + context_norm + one null key / value, not causal (:450), and the kv-cache return convention (:370, :399-406)
    ca = AP.Attention(dim=dim, heads=heads, dim_context=96, num_null_kv=1, norm_context=True, dropout=0.)
    sdc = synth_state_dict({k: tuple(v.shape) for k, v in ca.state_dict().items()}, 311)
    ca.load_state_dict(sdc)
    ca.to(dev()).eval()
    ctxt = rnd(b, 13, 96, seed=312)
    cmask = torch.rand(b, 13, generator=torch.Generator().manual_seed(313)) > 0.3
    with torch.no_grad():
        oc, kvc = ca(x.detach(), context=ctxt, mask=cmask.to(dev()), return_kv_cache=True)
        rc, _ = O.attention(sdc, '', x.detach().cpu(), heads, mask=cmask, context=ctxt.cpu(), causal=False)
    assert tuple(kvc.shape) == (2, b, 13, 64) and fro(oc, rc) <= 1.5e-2, (tuple(kvc.shape), fro(oc, rc))
    # --- FeedForward as a module (:251-260) and GEGLU alone (:246-249)
    ff = AP.FeedForward(dim=dim, dropout=0.)
    sdf = synth_state_dict({k: tuple(v.shape) for k, v in ff.state_dict().items()}, 321)
    ff.load_state_dict(sdf)
    ff.to(dev())
    x2 = rnd(b, n, dim, seed=322).requires_grad_(True)
    of = ff(x2)
    of.backward(go)
    sfr = {k: v.clone().requires_grad_(True) for k, v in sdf.items()}
    x2r = x2.detach().cpu().clone().requires_grad_(True)
    rf = O.feedforward(sfr, '', x2r)
    rf.backward(go.cpu())
    assert fro(of, rf) <= 1.5e-2 and fro(x2.grad, x2r.grad) <= 3e-2, (fro(of, rf), fro(x2.grad, x2r.grad))
    for k_, q_ in ff.named_parameters():
        assert fro(q_.grad, sfr[k_].grad) <= 3e-2, k_
    g = rnd(b, n, 2 * 170, seed=331).requires_grad_(True)
    og = AP.GEGLU()(g)
    og.backward(rnd(b, n, 170, seed=332))
    gr = g.detach().cpu().clone().requires_grad_(True)
    xh, gate = gr.chunk(2, dim=-1)
    rg = torch.nn.functional.gelu(gate) * xh
    rg.backward(rnd(b, n, 170, seed=332).cpu())
    assert fro(og, rg) <= 1e-6 and fro(g.grad, gr.grad) <= 1e-5
