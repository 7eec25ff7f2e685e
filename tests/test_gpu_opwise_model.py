"""Op-by-op ("teacher-forced") parity of everything AROUND the transformer stack at the benchmark size (round 4; tests/test_gpu_opwise.py covers the
stack itself): the embedding assembly and its scatter-add backward (`alm_embed_assemble` / `alm_embed_scatter_add`), the split-bf16 logit heads
(heads.py: logits = hi.Whi + hi.Wlo + lo.Whi on the bf16 MFMA GEMM), the cross-entropy forward / backward kernels with the wrapper's weighted loss
combination, and the FineTransformer's (relative frame, relative quantizer) attention-bias table (relpos.PosTableFn).

Method, as in the stack test: each piece gets THE HIP PATH'S OWN inputs on both sides.  The transformer stack is taken out of both programs -- the
HIP model's `transformer.forward` and the oracle's `O.transformer` are swapped for a stub that records the token embeddings it was handed and returns
ONE AND THE SAME hidden-state tensor (the real stack's output on these inputs) -- so what is compared is exactly: embeddings in, heads + CE +
loss combination out, and their gradients.  The oracle side runs at the HIP path's rounding points (oracle/rounding_matched.py: split-bf16 head
operands, bf16 dlogits) and, for the loss, also in plain fp32.
Bounds (rel-Frobenius unless stated): forward <= 1e-3 (north_star's number; measured figures are printed and land in gpurun_out/r6_opwise_parity.jsonl),
gradients that pass through the bf16 dlogits <= 3e-3, fp32 weight / table gradients <= 1e-3, the loss |d| <= 1e-5 relative.
Reference lines: CoarseTransformer.forward audiolm_pytorch.py:858-990, FineTransformer.forward :1136-1368, wrappers :1742-1854 / :2041-2137.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

import audiolm_oracle as O
import rounding_matched as RM
from common import synth_state_dict
from test_gpu_fullsize import _case
from test_gpu_parity import Codec

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, 'gpurun_out', 'r6_opwise_parity.jsonl')
bf = RM._bf


def _frob(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


class _Stub:
    """stands in for the transformer stack on the HIP side: records the tokens, returns the supplied flat hidden states"""

    def __init__(self):
        self.tokens = None
        self.hn = None

    def __call__(self, x, **kw):
        self.tokens = x
        return self.hn if kw.get('return_flat_hidden') else self.hn.view(x.shape)


@pytest.mark.parametrize('kind,batch', [('coarse', 8), ('fine', 2), ('coarse4096', 1)])
def test_embeddings_heads_and_cross_entropy_match_the_oracle_given_the_same_hidden_states(kind, batch):
    """coarse, B = 8: the benchmarked shape (the coarse head's three 4096 x 1025 x 1024 problems run on the staggered 256 x 256 tile there, the semantic
    head on the 128 x 128 one); fine, B = 2: the zero-padded coarse head + the grouped fine head with its ragged tail (N = 2049); coarse4096, B = 1
    (round 6): BASELINE configs[4]'s model at its own N = 8253 -- the C = 4097-column coarse heads (three 2250 x 4097 x 1024 problems + the ragged
    remainder, audiolm_pytorch.py:965-983), the 3 x 4097-row embedding table with the eos / next-quantizer aliasing (:896-906) and the CE over 4097 classes."""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import audiolm_pytorch as AP
    dev = torch.device('cuda:0')
    N_kind = dict(coarse=2048, fine=2049, coarse4096=8253)[kind]
    kind = 'coarse' if kind == 'coarse4096' else kind
    ctor, inputs, options, N, B = _case(kind, 4, N_kind, batch)
    K = dict(coarse=A.CoarseTransformer, fine=A.FineTransformer)[kind]
    torch.manual_seed(7)
    model = K(**ctor, residual_dtype=torch.bfloat16)
    state = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 4246)
    model.load_state_dict(state)
    model.to(dev)
    if kind == 'coarse':
        w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
        kw = dict(semantic_token_ids=inputs['semantic_token_ids'].to(dev), coarse_token_ids=inputs['coarse_token_ids'].to(dev))
    else:
        w = A.FineTransformerWrapper(transformer=model, codec=Codec(8), mask_prob=0.15)
        kw = dict(coarse_token_ids=inputs['coarse_token_ids'].to(dev), fine_token_ids=inputs['fine_token_ids'].to(dev))
    w.train()
    mask = inputs['forgetful_mask']
    orig_mask = AP.generate_mask_with_prob
    AP.generate_mask_with_prob = lambda shape, prob, device: mask.to(device).clone()
    tr = model.transformer
    real_forward = tr.forward
    rows = []

    def chk(name, got, want, tol):
        rows.append((name, _frob(got, want), tol))

    try:
        # ---- 1. the real stack once (no gradient): realistic hidden states, used as THE input of the heads on both sides
        seen = {}

        def capture(x, **kwargs):
            out = real_forward(x, **kwargs)
            seen['tokens'], seen['hn'] = x.detach(), out.detach()
            return out
        tr.forward = capture
        with torch.no_grad():
            w(**kw, return_loss=True)
        tokens_real, hn_real = seen['tokens'].float().cpu(), seen['hn'].float().reshape(B * N, -1).cpu()
        D = hn_real.shape[1]

        # ---- 2. HIP heads + CE + loss combination on the supplied hidden states (stack stubbed out), forward + backward
        stub = _Stub()
        stub.hn = hn_real.clone().to(dev).requires_grad_(True)
        tr.forward = stub
        for p in model.parameters():
            p.grad = None
        loss_h = w(**kw, return_loss=True)
        loss_h.backward()
        dhn_h = stub.hn.grad.detach().float().cpu()
        head_names = [k for k, _ in model.named_parameters() if 'logit' in k]
        gh = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if k in head_names and p.grad is not None}
        # logits for exactly the ids the loss path saw (no labels: the un-fused head path)
        with torch.no_grad():
            if kind == 'coarse':
                s_in, c_in, _, _, km = O.coarse_wrapper_bookkeeping(inputs['semantic_token_ids'], inputs['coarse_token_ids'], model.semantic_eos_id,
                                                                     model.coarse_eos_id, training=True, unique_consecutive=False)
                logits_h = model(semantic_token_ids=s_in.to(dev), coarse_token_ids=c_in.to(dev), self_attn_mask=km.to(dev))
            else:
                logits_h = model(inputs['coarse_token_ids'].reshape(B, -1).to(dev), inputs['fine_token_ids'].reshape(B, -1)[:, :-1].to(dev))
        logits_h = [t.float().cpu() for t in logits_h if t is not None]

        # ---- 3. HIP embedding assembly + scatter-add backward, given a seeded upstream gradient
        g = torch.Generator().manual_seed(5)
        dT = torch.randn(B, N, D, generator=g) * 1e-2
        for p in model.parameters():
            p.grad = None
        if kind == 'coarse':
            tok_h = model._assemble(s_in.to(dev), c_in.to(dev))[0]
        else:
            tok_h = model._assemble(inputs['coarse_token_ids'].reshape(B, -1).to(dev), inputs['fine_token_ids'].reshape(B, -1)[:, :-1].to(dev), None)[0]
        tok_h.backward(dT.to(dev))
        ge = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
        assert torch.equal(tok_h.detach().float().cpu(), tokens_real), 'the tokens the stack saw are the assembled embeddings'
    finally:
        tr.forward = real_forward
        AP.generate_mask_with_prob = orig_mask

    # ---- the oracle with its stack stubbed out the same way
    sd = {k: v.float() for k, v in state.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith('.beta')}
    full = dict(sd)
    full.update(params)
    if kind == 'coarse':
        cfg = O.Cfg(dim=ctor['dim'], depth=ctor['depth'], streams=4, num_semantic_tokens=500, codebook_size=ctor['codebook_size'], num_coarse_quantizers=3)
    else:
        cfg = O.Cfg(dim=ctor['dim'], depth=ctor['depth'], streams=4, codebook_size=1024, num_coarse_quantizers=3, num_fine_quantizers=5)

    def oracle_pass(rounded):
        cap = {}
        hid = hn_real.clone().view(B, N, D).requires_grad_(True)

        def stub_tr(sd_, p_, x, **kwargs):
            cap['tokens'] = x
            return hid
        fwd_name = 'coarse_forward' if kind == 'coarse' else 'fine_forward'
        orig_fwd = getattr(O, fwd_name)
        setattr(O, fwd_name, lambda *a, **k: cap.setdefault('logits', orig_fwd(*a, **k)))
        for p in params.values():
            p.grad = None
        try:
            import contextlib
            with (RM.rounding_matched() if rounded else contextlib.nullcontext()):
                saved_tr = O.transformer
                O.transformer = stub_tr
                try:
                    if kind == 'coarse':
                        loss = O.coarse_wrapper_loss(full, cfg, inputs['semantic_token_ids'], inputs['coarse_token_ids'], training=True, unique_consecutive=False,
                                                     forgetful_mask=mask)
                    else:
                        loss = O.fine_wrapper_loss(full, cfg, inputs['coarse_token_ids'], inputs['fine_token_ids'], forgetful_mask=mask)
                finally:
                    O.transformer = saved_tr
        finally:
            setattr(O, fwd_name, orig_fwd)
        loss.backward()
        return loss.detach(), cap, hid.grad, {k: (None if p.grad is None else p.grad.clone()) for k, p in params.items()}

    loss_r, cap_r, dhid_r, gp_r = oracle_pass(True)              # rounding-matched heads (split-bf16 operands, bf16 dlogits)
    loss_f, cap_f, _, _ = oracle_pass(False)                     # plain fp32 heads: the reference's arithmetic
    # embeddings: the oracle's own assembly (captured on its way into the stubbed stack), then its backward on the same upstream gradient
    tok_o = cap_f['tokens']
    for p in params.values():
        p.grad = None
    (tok_o * dT).sum().backward()
    ge_o = {k: p.grad for k, p in params.items() if p.grad is not None}

    chk('embedding assembly: tokens (fp32 gather + add)', tokens_real, tok_o.detach(), 1e-6)
    for k, want in ge_o.items():
        if float(want.norm()) > 0:
            assert k in ge, f'no HIP gradient for {k}'
            chk(f'embedding scatter-add: d {k}', ge[k], want, 1e-3)
    lo_r = [t.detach() for t in (cap_r['logits'] if isinstance(cap_r['logits'], (tuple, list)) else (cap_r['logits'],)) if t is not None]
    lo_f = [t.detach() for t in (cap_f['logits'] if isinstance(cap_f['logits'], (tuple, list)) else (cap_f['logits'],)) if t is not None]
    assert len(logits_h) == len(lo_r)
    for i, (got, wr, wf) in enumerate(zip(logits_h, lo_r, lo_f)):
        chk(f'logits[{i}] {tuple(got.shape)} vs split-bf16 restatement', got, wr, 1e-3)
        chk(f'logits[{i}] vs fp32 heads (the reference arithmetic)', got, wf, 1e-3)
    rel_r = abs(float(loss_h) - float(loss_r)) / abs(float(loss_r))
    rel_f = abs(float(loss_h) - float(loss_f)) / abs(float(loss_f))
    rows.append(('loss (CE means + weighted combination) vs rounding-matched heads', rel_r, 1e-5))
    rows.append(('loss vs fp32 heads', rel_f, 1e-4))
    chk('d hidden states (CE backward -> bf16 dlogits -> dgrad GEMM + row scatter)', dhn_h.view(B, N, D), dhid_r, 3e-3)
    for k in head_names:
        if gp_r.get(k) is not None and float(gp_r[k].norm()) > 0:
            chk(f'd {k} (wgrad of the head: bf16 dlogits^T hi)', gh[k], gp_r[k], 3e-3 if k.endswith('bias') else 1e-3)

    bad = [(nm, e, tol) for nm, e, tol in rows if not e <= tol]
    print(f'{kind} B={B} N={N}: {len(rows)} comparisons around the stack, {len(bad)} over their bound')
    for nm, e, tol in rows:
        print(f'   {nm}: {e:.2e} (bound {tol:.0e})')
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, 'a') as fh:
        fh.write(json.dumps(dict(kind=kind + ' (embeddings / heads / CE around the stack)', N=N, B=B, comparisons=len(rows), over_bound=len(bad),
                                 rows=[(nm, float(f'{e:.3e}'), tol) for nm, e, tol in rows])) + '\n')
    assert not bad, '\n'.join(f'{nm}: {e:.3e} > {tol:.0e}' for nm, e, tol in bad)


def test_fine_attention_bias_table_at_the_benchmark_size_vs_oracle():
    """FineTransformer(flash_attn=False) at N = 2049 (768 coarse + 1279 fine positions): the per-head table relpos.PosTableFn builds from pos_bias_mlp +
    null_pos_bias, expanded through the index vectors the attention kernels use, against (a) a restatement of reference :1229-1298 at the HIP path's
    rounding points (the hidden C x C layer runs on the bf16 MFMA GEMM: bf16 activations and weights, fp32 accumulate) -- bound 1e-3 -- and (b) the fp32
    oracle's dense (h, n, n) tensor -- bound 1e-2, the bf16 operand rounding of that one layer; and the table's gradient into the MLP parameters and
    null_pos_bias, given a seeded upstream gradient, against autograd of (a)."""
    import audiolm_pytorch_amd as A
    from test_gpu_bias import dense_bias
    dev = torch.device('cuda:0')
    ctor = dict(dim=1024, depth=1, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024, flash_attn=False)
    torch.manual_seed(3)
    model = A.FineTransformer(**ctor)
    state = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 777)
    model.load_state_dict(state)
    model.to(dev)
    sd = {k: v.float() for k, v in state.items()}
    n, nf = 768, 1279
    N = n + nf + 2
    bias = model._attn_bias(n, nf, dev)
    got = dense_bias(bias.tbl.detach(), (bias.qkey4, bias.kkey4, bias.qattr, bias.kattr)).cpu()          # (h, N, N), what the kernels add to the scores
    cfg = O.Cfg(dim=1024, depth=1, streams=4, codebook_size=1024, num_coarse_quantizers=3, num_fine_quantizers=5)
    want32 = O.fine_attn_bias(sd, cfg, n, nf, 'cpu')
    # (a) the same computation with the hidden layer's operands rounded to bf16 (relpos.PosTableFn: posmlp_in writes bf16 activations, the C x C layer
    # is alm_gemm_bf16_nt on bf16(W), its SiLU output is bf16 again; first and last layers fp32 weights)
    prm = {k: sd[k].clone().requires_grad_(True) for k in sd if k.startswith('pos_bias_mlp.') or k == 'null_pos_bias'}

    def table(p):
        grid, _ = A.relpos.fine_index(n, nf, 3, 5, torch.device('cpu'))
        h = RM.rst(F.silu(F.linear(grid, p['pos_bias_mlp.0.weight'], p['pos_bias_mlp.0.bias'])))
        h = RM.rst(F.silu(F.linear(h, RM.rf(p['pos_bias_mlp.2.weight']), p['pos_bias_mlp.2.bias'])))
        return F.linear(h, p['pos_bias_mlp.4.weight'], p['pos_bias_mlp.4.bias'])                          # [L, H]
    t_rm = table(prm)
    tbl_rm = torch.cat((prm['null_pos_bias'].reshape(-1, 1), t_rm.t()), dim=1) * (64 ** 0.5)              # raw-score units, slot 0 = special
    want_rm = dense_bias(tbl_rm.detach(), tuple(t.cpu() for t in (bias.qkey4, bias.kkey4, bias.qattr, bias.kattr)))
    causal = torch.ones(N, N, dtype=torch.bool).tril()
    e_rm = _frob(got[:, causal], want_rm[:, causal])
    e_32 = _frob(got[:, causal], want32[:, causal])
    e_self = _frob(want_rm[:, causal], want32[:, causal])
    # gradient of the table: a seeded upstream gradient on the table itself
    g = torch.Generator().manual_seed(8)
    dt = torch.randn(bias.tbl.shape, generator=g) * 1e-2
    for p in model.parameters():
        p.grad = None
    bias.tbl.backward(dt.to(dev))
    (tbl_rm * dt).sum().backward()
    rows = [('dense bias over the causal region vs bf16-operand restatement', e_rm, 1e-3), ('... vs the fp32 oracle', e_32, 1e-2)]
    for k, p in prm.items():
        gk = dict(model.named_parameters())[k].grad
        assert gk is not None, k
        rows.append((f'd {k}', _frob(gk.float().cpu(), p.grad), 1e-2))
    print(f'fine bias table N={N}: restatement-vs-fp32 {e_self:.2e}')
    for nm, e, tol in rows:
        print(f'   {nm}: {e:.2e} (bound {tol:.0e})')
    with open(REPORT, 'a') as fh:
        fh.write(json.dumps(dict(kind='fine attention-bias table', N=N, rows=[(nm, float(f'{e:.3e}'), tol) for nm, e, tol in rows])) + '\n')
    bad = [(nm, e, tol) for nm, e, tol in rows if not e <= tol]
    assert not bad, bad
