"""Structured attention bias (`flash_attn=False` models) on a real MI355X, through the C ABI:
  * alm_mqa_attn_bias_fwd / _bwd / alm_attn_bias_grad_reduce vs the fp32 oracle attention fed the DENSE (h, n, n) bias tensor the
    reference would build (oracle/audiolm_oracle.py: attend, rel_pos_bias, coarse / fine bias), incl. the table gradient;
  * the table MLP (alm_posmlp_* + GEMMs, relpos.PosTableFn) vs the same MLP in fp32 PyTorch autograd;
  * index-vector builders vs the oracle's dense bias (exact slot agreement).
Tolerances: attention outputs / dq / dk / dv as in test_gpu_kernels (bf16 operands: 1.2e-2 / 2e-2 rel-max); table gradient 1e-2 rel-max
(fp32 sums of dS = P (dP - delta), P and dP from bf16 operands); MLP table 1e-2 rel-max, MLP parameter gradients 5e-2 rel-Frobenius
(bf16 GEMM operands, the precision the reference's autocast gives these Linears; first-layer bias / weight gradients pass through three bf16
roundings of the chain: 5e-2, cf. the reference's own bf16 noise of 3-9 % on these tensors in tests/golden/bf16_noise.pt).
"""
import os

import pytest
import torch
import torch.nn.functional as F

import audiolm_oracle as O
from common import GOLDEN_DIR

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32
SCALE = 64 ** -0.5


@pytest.fixture(scope='module')
def ops():
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import ops as o
    return o


def dev():
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0, dtype=F32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


def relmax(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def dense_bias(tbl, index):
    """(h, n, n) score bias (already x scale, i.e. what the reference adds to sim) from a raw-score-unit table + index vectors."""
    qkey4, kkey4, qattr, kattr = [t.long() for t in index]
    LT = tbl.shape[1]
    slot = (qkey4[:, None] - kkey4[None, :]) // 4
    ok = (slot >= 0) & (slot < LT)
    slot = torch.where((qattr[:, None] & kattr[None, :]) != 0, torch.zeros_like(slot), slot)
    vals = tbl[:, slot.clamp(0, LT - 1)]
    vals = torch.where(ok[None] | ((qattr[:, None] & kattr[None, :]) != 0)[None], vals, torch.zeros_like(vals))
    return vals * SCALE


def make_index(kind, N):
    from audiolm_pytorch_amd import relpos
    if kind == 'toeplitz':
        return relpos.toeplitz_index(N, dev()), 2 * N
    if kind == 'coarse':
        return relpos.toeplitz_index(N, dev(), num_leading=N // 3 + 1), 2 * N
    Qc, Qf = 3, 5
    nf = (N - 2) * 5 // 8
    n = N - 2 - nf
    grid, index = relpos.fine_index(n, nf, Qc, Qf, dev())
    return index, grid.shape[0] + 1


@pytest.mark.parametrize('kind', ['toeplitz', 'coarse', 'fine'])
@pytest.mark.parametrize('B,N,H,use_mask', [(2, 37, 8, False), (2, 200, 8, True), (1, 512, 8, False), (3, 130, 6, True), (1, 1030, 8, True), (2, 96, 3, False)])
def test_biased_attention_fwd_bwd(ops, kind, B, N, H, use_mask):
    from audiolm_pytorch_amd import relpos
    d = 64
    q = rnd(B * N, H * d, seed=32, dtype=BF16)
    kv = rnd(B * N, 2 * d, seed=33, dtype=BF16)
    k, v = kv[:, :d], kv[:, d:]
    index, LT = make_index(kind, N)
    tbl = rnd(H, LT, seed=34, scale=6.0)                              # raw-score units: bias = tbl / 8 ~ N(0, 0.75^2)
    mask = None
    if use_mask:
        g = torch.Generator().manual_seed(35)
        mask = (torch.rand(B, N, generator=g) > 0.25)
        mask[:, 0] = True
        mask = mask.to(dev())
    mu8 = None if mask is None else mask.contiguous().view(torch.uint8)
    bias = relpos.AttnBias(tbl, *index)
    o, lse = ops.mqa_attn_fwd(q, k, v, mu8, B, N, H, d, bias=bias)

    qf = q.float().clone().requires_grad_(True)
    kf = k.float().clone().requires_grad_(True)
    vf = v.float().clone().requires_grad_(True)
    tf = tbl.clone().requires_grad_(True)
    ref = O.attend(qf.view(B, N, H, d).permute(0, 2, 1, 3), kf.view(B, N, d), vf.view(B, N, d), mask=mask, attn_bias=dense_bias(tf, index), causal=True)
    ref2 = ref.permute(0, 2, 1, 3).reshape(B * N, H * d)
    e = relmax(o, ref2)
    assert e <= 1.2e-2, f'{kind}: biased attention fwd rel-max err {e}'
    do = rnd(B * N, H * d, seed=36, dtype=BF16)
    ref2.backward(do.float())
    part = ops.attn_bias_part(B, N, H, LT, dev())
    dq, dkv_parts = ops.mqa_attn_bwd(q, k, v, mu8, o, lse, do, B, N, H, d, bias=bias, dtbl_part=part)
    dkv = dkv_parts.sum(0)
    for name, got, want in (('dq', dq, qf.grad), ('dk', dkv[:, :d], kf.grad), ('dv', dkv[:, d:], vf.grad)):
        e = relmax(got, want)
        assert e <= 2e-2, f'{kind}: biased attention bwd {name} rel-max err {e}'
    dtbl = ops.attn_bias_grad_reduce(part, B, N, H)
    want = tf.grad
    e = relmax(dtbl, want)
    assert e <= 1e-2, f'{kind}: table gradient rel-max err {e}'
    if kind != 'toeplitz':
        e0 = float((dtbl[:, 0] - want[:, 0]).abs().max() / want[:, 0].abs().max().clamp(min=1e-30))
        assert e0 <= 3e-2, f'{kind}: special-slot gradient rel err {e0}'          # one heavily cancelling sum over every special pair
    # a second layer accumulates into the same partials: reduce(2 x) == 2 x reduce
    ops.mqa_attn_bwd(q, k, v, mu8, o, lse, do, B, N, H, d, bias=bias, dtbl_part=part)
    assert relmax(ops.attn_bias_grad_reduce(part, B, N, H), 2 * dtbl) <= 1e-5
    # the un-biased entry points are untouched by a zero table
    zb = relpos.AttnBias(torch.zeros_like(tbl), *index)
    o0, lse0 = ops.mqa_attn_fwd(q, k, v, mu8, B, N, H, d, bias=zb)
    o1, lse1 = ops.mqa_attn_fwd(q, k, v, mu8, B, N, H, d)
    assert torch.equal(o0, o1) and torch.equal(lse0, lse1)


def test_index_vectors_match_oracle_dense_bias():
    """relpos.toeplitz_index / fine_index + a table == the dense tensors oracle.rel_pos_bias / coarse override / fine_attn_bias gather."""
    from audiolm_pytorch_amd import relpos
    g = torch.Generator().manual_seed(41)
    H, n = 4, 23
    # Semantic / Coarse: table rows = MLP output on -(n-1) .. n-1
    T = torch.randn(2 * n - 1, H, generator=g)
    tbl = torch.cat((torch.full((H, 1), 7.5), T.t() / SCALE), dim=1).to(dev())
    i_pos = torch.arange(n)
    rel = i_pos[:, None] - i_pos[None, :] + n - 1
    want = T[rel].permute(2, 0, 1)
    got = dense_bias(tbl, relpos.toeplitz_index(n, dev())).cpu()
    causal = torch.ones(n, n, dtype=torch.bool).tril()
    assert torch.allclose(got[:, causal], want[:, causal], atol=1e-5)
    ns1 = 9
    is_sem = torch.arange(n) < ns1
    cross = is_sem[:, None] ^ is_sem[None, :]
    want_c = torch.where(cross, torch.tensor(7.5 * SCALE), want)
    got_c = dense_bias(tbl, relpos.toeplitz_index(n, dev(), num_leading=ns1)).cpu()
    assert torch.allclose(got_c[:, causal], want_c[:, causal], atol=1e-5)
    # Fine: oracle.fine_attn_bias with an identity-like "MLP" replaced by a random table -> compare slots through the oracle's own index maths
    Qc, Qf = 3, 5
    for (nc, nf) in ((12, 20), (11, 17), (3, 1), (30, 4)):
        grid, index = relpos.fine_index(nc, nf, Qc, Qf, dev())
        L = grid.shape[0]
        Tf = torch.randn(L, H, generator=g)
        sd = {'null_pos_bias': torch.full((H, 1, 1), -3.25)}
        # restate oracle.fine_attn_bias' gather with the random table in place of the MLP output
        cs, fs = -(-nc // Qc), -(-nf // Qf)
        M, Qt = max(cs, fs), Qc + Qf
        R = 2 * Qt - 1
        c_pos = F.pad(torch.arange(cs).repeat_interleave(Qc)[:nc], (1, 0), value=-1)
        f_pos = F.pad(torch.arange(fs).repeat_interleave(Qf)[:nf], (1, 0), value=-1)
        c_off = F.pad(torch.arange(Qc).repeat(cs)[:nc], (1, 0), value=0)
        f_off = F.pad(torch.arange(Qf).repeat(fs)[:nf] + Qc, (1, 0), value=0)
        pos, off = torch.cat((c_pos, f_pos)), torch.cat((c_off, f_off))
        pin = torch.stack((pos.clamp(min=0), off), dim=-1)
        rd = pin[:, None, :] - pin[None, :, :]
        idx = (rd[..., 0] + M - 1) * R + (rd[..., 1] + Qt - 1)
        want_f = Tf[idx].permute(2, 0, 1)
        st = pos == -1
        want_f = torch.where(st[:, None] | st[None, :], sd['null_pos_bias'], want_f)
        # the MLP input grid must enumerate (rel_seq, rel_off) in table-row order
        assert torch.equal(grid.cpu()[:, 0].long() * R + grid.cpu()[:, 1].long(), torch.arange(L))
        tblf = torch.cat((torch.full((H, 1), -3.25 / SCALE), Tf.t() / SCALE), dim=1).to(dev())
        got_f = dense_bias(tblf, index).cpu()
        N = nc + nf + 2
        causal = torch.ones(N, N, dtype=torch.bool).tril()
        assert torch.allclose(got_f[:, causal], want_f[:, causal], atol=1e-5), (nc, nf)


@pytest.mark.parametrize('in_dim,nhid,L,C,H', [(1, 2, 399, 32, 8), (2, 1, 1005, 64, 8), (1, 2, 4095, 512, 8), (2, 1, 300, 32, 3)])
def test_pos_table_mlp(ops, in_dim, nhid, L, C, H):
    from audiolm_pytorch_amd import relpos
    g = torch.Generator().manual_seed(50)
    if in_dim == 1:
        x = torch.arange(-(L // 2), L - L // 2).float().unsqueeze(-1)
    else:
        x = torch.stack((torch.arange(L) // 15, torch.arange(L) % 15), dim=-1).float()
    dims = [in_dim] + [C] * (nhid + 1) + [H]
    ws = []
    for a, b in zip(dims[:-1], dims[1:]):
        bound = a ** -0.5
        ws += [(torch.rand(b, a, generator=g) * 2 - 1) * bound, (torch.rand(b, generator=g) * 2 - 1) * bound]
    special = torch.randn(H, 1, 1, generator=g)
    ws_g = [w.to(dev()).requires_grad_(True) for w in ws]
    sp_g = special.to(dev()).requires_grad_(True)
    tbl = relpos.PosTableFn.apply(x.to(dev()), sp_g, 8.0, *ws_g)
    # fp32 reference
    ws_r = [w.clone().requires_grad_(True) for w in ws]
    sp_r = special.clone().requires_grad_(True)
    h = x
    for li in range(nhid + 1):
        h = F.silu(F.linear(h, ws_r[2 * li], ws_r[2 * li + 1]))
    out = F.linear(h, ws_r[-2], ws_r[-1])
    ref = torch.cat((sp_r.reshape(H, 1), out.t()), dim=1) * 8.0
    e = relmax(tbl.cpu(), ref)
    assert e <= 1e-2, f'table rel-max err {e}'
    dt = torch.randn(H, L + 1, generator=g)
    tbl.backward(dt.to(dev()))
    ref.backward(dt)
    for i, (a, b) in enumerate(zip(ws_g, ws_r)):
        err = float((a.grad.cpu().double() - b.grad.double()).norm() / b.grad.double().norm().clamp(min=1e-30))
        assert err <= 5e-2, f'MLP parameter {i} grad rel-frob {err}'
    assert relmax(sp_g.grad.cpu(), sp_r.grad) <= 1e-5


def test_attend_module_with_structured_bias():
    """the Attend mirror (reference attend.py:98-146 signature: q (b h n d), k / v (b n d), mask, attn_bias) with a structured bias: output and
    the gradients of q, k, v AND the bias table vs the oracle's math path fed the dense tensor"""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import relpos
    B, H, N, d = 2, 8, 150, 64
    q = rnd(B, H, N, d, seed=60).requires_grad_(True)
    k = rnd(B, N, d, seed=61).requires_grad_(True)
    v = rnd(B, N, d, seed=62).requires_grad_(True)
    g = torch.Generator().manual_seed(63)
    mask = (torch.rand(B, N, generator=g) > 0.2)
    mask[:, 0] = True
    mask = mask.to(dev())
    index = relpos.toeplitz_index(N, dev(), num_leading=40)
    tbl = rnd(H, 2 * N, seed=64, scale=6.0).requires_grad_(True)
    out = A.Attend(causal=True)(q, k, v, mask=mask, attn_bias=relpos.AttnBias(tbl, *index))
    go = rnd(B, H, N, d, seed=65)
    out.backward(go)
    qr, kr, vr, tr = [t.detach().clone().requires_grad_(True) for t in (q, k, v, tbl)]
    # the kernels see bf16 operands: give the reference the same rounded q / k / v
    ref = O.attend(qr.bfloat16().float(), kr.bfloat16().float(), vr.bfloat16().float(), mask=mask, attn_bias=dense_bias(tr, index), causal=True)
    ref.backward(go.bfloat16().float())
    assert relmax(out, ref) <= 1.2e-2
    for name, got, want in (('dq', q.grad, qr.grad), ('dk', k.grad, kr.grad), ('dv', v.grad, vr.grad), ('dtbl', tbl.grad, tr.grad)):
        assert relmax(got, want) <= 2e-2, (name, relmax(got, want))


# ---------------------------------------------------------------------------------------------- dense attn_bias: the reference's math path

def test_attend_dense_bias_matches_real_reference_golden():
    """tests/golden/attend.pt: outputs of the REAL reference Attend (attend.py:98-146) incl. `math_mask_bias` = key mask + an arbitrary dense
    (h, n, n) attn_bias tensor.  Our Attend takes the same tensor (math path on the MFMA GEMMs + csrc/xattn.hip) -- head dim 16, n = 37."""
    import audiolm_pytorch_amd as A
    fx = torch.load(os.path.join(GOLDEN_DIR, 'attend.pt'), weights_only=False)
    i, o = fx['inputs'], fx['outputs']
    q, k, v, mask, bias = (i[n].to(dev()) for n in ('q', 'k', 'v', 'mask', 'bias'))
    att = A.Attend(causal=True)
    assert relmax(att(q, k, v, mask=mask, attn_bias=bias).cpu(), o['math_mask_bias']) <= 1.2e-2
    assert relmax(att(q, k, v, mask=mask, attn_bias=torch.zeros_like(bias)).cpu(), o['math_mask']) <= 1.2e-2
    assert relmax(att(q, k, v, attn_bias=torch.zeros(1, 1, 1, device=dev())).cpu(), o['math_plain']) <= 1.2e-2      # broadcastable bias


@pytest.mark.parametrize('B,H,N,M,d,causal', [(2, 8, 150, 150, 64, True), (1, 4, 64, 200, 64, False), (3, 2, 33, 33, 32, True), (2, 4, 100, 37, 64, False)])
def test_attend_math_path_dense_bias_forward_backward(B, H, N, M, d, causal):
    """the math path vs the oracle's attend: output and the gradients of q, k, v and of the DENSE bias; causal and non-causal, key counts
    different from the query count (cross-attention shapes), a broadcast (h, 1, j) bias"""
    import audiolm_pytorch_amd as A
    q = rnd(B, H, N, d, seed=70).requires_grad_(True)
    k = rnd(B, M, d, seed=71).requires_grad_(True)
    v = rnd(B, M, d, seed=72).requires_grad_(True)
    g = torch.Generator().manual_seed(73)
    mask = (torch.rand(B, M, generator=g) > 0.2)
    mask[:, 0] = True
    mask = mask.to(dev())
    full = N == M
    bias = (rnd(H, N, M, seed=74, scale=2.0) if full else rnd(H, 1, M, seed=74, scale=2.0)).requires_grad_(True)
    out = A.Attend(causal=causal)(q, k, v, mask=mask, attn_bias=bias)
    go = rnd(B, H, N, d, seed=75)
    out.backward(go)
    qr, kr, vr, br = [t.detach().clone().requires_grad_(True) for t in (q, k, v, bias)]
    ref = O.attend(qr.bfloat16().float(), kr.bfloat16().float(), vr.bfloat16().float(), mask=mask, attn_bias=br, causal=causal)
    ref.backward(go.bfloat16().float())
    assert relmax(out, ref) <= 1.2e-2
    for name, got, want in (('dq', q.grad, qr.grad), ('dk', k.grad, kr.grad), ('dv', v.grad, vr.grad), ('dbias', bias.grad, br.grad)):
        assert got.shape == want.shape and relmax(got, want) <= 2e-2, (name, relmax(got, want))


@pytest.mark.parametrize('streams', [1, 4])
def test_transformer_with_dense_attn_bias_vs_oracle(streams):
    """Transformer.forward(attn_bias=<any tensor>) (reference audiolm_pytorch.py:500-503 takes whatever it is given): the fused stack routes the
    self-attention through the math path; output, parameter gradients and the gradient of the bias tensor vs the fp32 oracle."""
    import audiolm_pytorch_amd as A
    from common import synth_state_dict
    dim, depth, heads, n, b = 128, 2, 4, 70, 2
    torch.manual_seed(0)
    tr = A.audiolm_pytorch.Transformer(dim=dim, depth=depth, heads=heads, num_residual_streams=streams, rel_pos_bias=False)
    shapes = {k: tuple(v.shape) for k, v in tr.state_dict().items()}
    sd = synth_state_dict(shapes, 77)
    tr.load_state_dict(sd)
    tr.to(dev())
    x = rnd(b, n, dim, seed=78).requires_grad_(True)
    bias = rnd(heads, n, n, seed=79, scale=1.5).requires_grad_(True)
    g = torch.Generator().manual_seed(80)
    mask = (torch.rand(b, n, generator=g) > 0.15)
    mask[:, 0] = True
    out = tr(x, self_attn_mask=mask.to(dev()), attn_bias=bias)
    go = rnd(b, n, dim, seed=81)
    out.backward(go)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, br = x.detach().cpu().clone().requires_grad_(True), bias.detach().cpu().clone().requires_grad_(True)
    ref = O.transformer(sdr, '', xr, depth=depth, heads=heads, streams=streams, self_attn_mask=mask, attn_bias=br)
    ref.backward(go.cpu())
    fro = lambda a, w: float((a.detach().cpu().double() - w.double()).norm() / w.double().norm().clamp(min=1e-30))
    assert fro(out, ref) <= 1.5e-2, fro(out, ref)
    assert fro(bias.grad, br.grad) <= 5e-2, fro(bias.grad, br.grad)
    assert fro(x.grad, xr.grad) <= 5e-2
    # the hyper-connection scalar statistics (cancelling sums) are compared through their per-feature siblings: see tests/test_gpu_parity.py
    from test_gpu_parity import HC_SCALARS
    worst = max((fro(p.grad, sdr[k].grad), k) for k, p in tr.named_parameters()
                if sdr[k].grad is not None and float(sdr[k].grad.norm()) > 1e-7 and not k.endswith(HC_SCALARS))
    assert worst[0] <= 8e-2, worst


def test_transformer_dense_attn_bias_with_a_conditioning_prefix_vs_oracle():
    """attn_bias=<dense tensor> together with cond_as_self_attn_prefix (reference audiolm_pytorch.py:330-345: the prefix keys get zero bias columns,
    F.pad(attn_bias, (m, 0)); :510-515): the math path runs over the concatenated [prefix | sequence] key set.  Output, d(context), d(bias) and
    the parameter gradients vs the fp32 oracle (round 3: this combination used to be refused)."""
    import audiolm_pytorch_amd as A
    from common import synth_state_dict
    from test_gpu_parity import HC_SCALARS
    dim, depth, heads, n, b, m = 128, 2, 4, 60, 2, 7
    torch.manual_seed(0)
    tr = A.audiolm_pytorch.Transformer(dim=dim, depth=depth, heads=heads, num_residual_streams=4, rel_pos_bias=False, cond_as_self_attn_prefix=True)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in tr.state_dict().items()}, 177)
    tr.load_state_dict(sd)
    tr.to(dev())
    x = rnd(b, n, dim, seed=178).requires_grad_(True)
    ctxt = rnd(b, m, dim, seed=182).requires_grad_(True)
    bias = rnd(heads, n, n, seed=179, scale=1.5).requires_grad_(True)
    g = torch.Generator().manual_seed(180)
    mask = (torch.rand(b, n, generator=g) > 0.15)
    mask[:, 0] = True
    cmask = torch.rand(b, m, generator=g) > 0.3
    cmask[:, 0] = True
    out = tr(x, self_attn_mask=mask.to(dev()), attn_bias=bias, context=ctxt, context_mask=cmask.to(dev()))
    go = rnd(b, n, dim, seed=181)
    out.backward(go)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, br, cr = (t.detach().cpu().clone().requires_grad_(True) for t in (x, bias, ctxt))
    ref = O.transformer(sdr, '', xr, depth=depth, heads=heads, streams=4, self_attn_mask=mask, attn_bias=br, context=cr, context_mask=cmask,
                        cond_as_self_attn_prefix=True)
    ref.backward(go.cpu())
    fro = lambda a, w: float((a.detach().cpu().double() - w.double()).norm() / w.double().norm().clamp(min=1e-30))
    assert fro(out, ref) <= 1.5e-2, fro(out, ref)
    assert fro(bias.grad, br.grad) <= 5e-2 and fro(x.grad, xr.grad) <= 5e-2 and fro(ctxt.grad, cr.grad) <= 5e-2, (fro(bias.grad, br.grad), fro(x.grad, xr.grad), fro(ctxt.grad, cr.grad))
    worst = max((fro(p.grad, sdr[k].grad), k) for k, p in tr.named_parameters()
                if sdr[k].grad is not None and float(sdr[k].grad.norm()) > 1e-7 and not k.endswith(HC_SCALARS))
    assert worst[0] <= 8e-2, worst


@pytest.mark.parametrize('streams', [1, 4])
def test_transformer_ff_dropout_vs_oracle_with_the_same_masks(streams, monkeypatch):
    """Transformer(ff_dropout = p) (reference FeedForward: nn.Dropout between the inner LayerNorm and the output projection, audiolm_pytorch.py:251-260).
    The device run draws its keep masks through core._dropout_keep; the test swaps that for a seeded generator, records the masks and hands them to
    the oracle: output and every gradient must agree as they do without dropout.  eval() must be the p = 0 model exactly."""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import core
    from common import synth_state_dict
    dim, depth, heads, n, b, p = 128, 2, 4, 70, 2, 0.25
    torch.manual_seed(0)
    tr = A.audiolm_pytorch.Transformer(dim=dim, depth=depth, heads=heads, num_residual_streams=streams, rel_pos_bias=False, flash_attn=True, ff_dropout=p)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in tr.state_dict().items()}, 91)
    tr.load_state_dict(sd)
    tr.to(dev())
    x = rnd(b, n, dim, seed=92).requires_grad_(True)
    gm = torch.Generator().manual_seed(93)
    mask = (torch.rand(b, n, generator=gm) > 0.15)
    mask[:, 0] = True
    masks = []

    def seeded_keep(shape, p_, device):
        k = (torch.rand(shape, generator=gm) >= p_).to(torch.bfloat16)
        masks.append(k)
        return k.to(device)
    monkeypatch.setattr(core, '_dropout_keep', seeded_keep)
    tr.train()
    out = tr(x, self_attn_mask=mask.to(dev()))
    go = rnd(b, n, dim, seed=94)
    out.backward(go)
    assert len(masks) == depth and 0.6 < float(torch.stack(masks).float().mean()) < 0.9
    inner = int(dim * 8 / 3)
    keep = [m[:, :inner].float().reshape(b, n, inner) for m in masks]
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    ref = O.transformer(sdr, '', xr, depth=depth, heads=heads, streams=streams, self_attn_mask=mask, ff_keep=keep, ff_dropout=p)
    ref.backward(go.cpu())
    fro = lambda a, w: float((a.detach().cpu().double() - w.double()).norm() / w.double().norm().clamp(min=1e-30))
    assert fro(out, ref) <= 1.5e-2, fro(out, ref)
    assert fro(x.grad, xr.grad) <= 5e-2, fro(x.grad, xr.grad)
    from test_gpu_parity import HC_SCALARS
    worst = max((fro(q.grad, sdr[k].grad), k) for k, q in tr.named_parameters()
                if sdr[k].grad is not None and float(sdr[k].grad.norm()) > 1e-7 and not k.endswith(HC_SCALARS))
    assert worst[0] <= 8e-2, worst
    # a run WITHOUT the masks must be clearly different (the masks really were applied) ...
    ref0 = O.transformer({k: v.detach() for k, v in sd.items()}, '', x.detach().cpu(), depth=depth, heads=heads, streams=streams, self_attn_mask=mask)
    assert fro(out, ref0) > 5e-2
    # ... and eval() is the dropout-free model: no mask is drawn
    tr.eval()
    n_before = len(masks)
    with torch.no_grad():
        oe = tr(x.detach(), self_attn_mask=mask.to(dev()))
    assert len(masks) == n_before and fro(oe, ref0) <= 1.5e-2
