"""Pins oracle/audiolm_oracle.py (the CPU fp32 restatement) against golden vectors produced by the REAL
reference (tests/golden/make_golden.py, run in the build container).  CPU only.

Tolerances: fp32 vs fp32 on CPU, same op order up to reassociation -> loss 1e-5 rel, logits 2e-4 abs,
gradients 1e-4 of the gradient's max-abs.
"""
import glob
import os

import pytest
import torch

import audiolm_oracle as O
from common import GOLDEN_DIR, synth_state_dict


def _load(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + '.pt'), weights_only=False)


def _cfg(fx):
    c = fx['ctor']
    return O.Cfg(dim=c['dim'], depth=c['depth'], heads=c.get('heads', 8), streams=c.get('num_residual_streams', 4),
                 num_semantic_tokens=c.get('num_semantic_tokens', 0), codebook_size=c.get('codebook_size', 0),
                 num_coarse_quantizers=c.get('num_coarse_quantizers', 0), num_fine_quantizers=c.get('num_fine_quantizers', 0),
                 cond_as_self_attn_prefix=c.get('cond_as_self_attn_prefix', False))


def oracle_run(fx):
    """Runs the oracle on a fixture -> (loss, logits tuple, grads dict)."""
    sd = synth_state_dict(fx['shapes'], fx['seed'])
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith('.beta')}
    full = dict(sd)
    full.update(params)
    cfg, inp, opt = _cfg(fx), fx['inputs'], fx['options']
    ckw = dict(text_embeds=inp.get('text_embeds'), cond_drop_keep=inp.get('cond_keep'))       # conditioning fixtures (has_condition=True)
    captured = {}
    if fx['kind'] == 'semantic':
        orig = O.semantic_forward
        O.semantic_forward = lambda *a, **k: captured.setdefault('l', orig(*a, **k))
        try:
            loss = O.semantic_wrapper_loss(full, cfg, inp['ids'], training=opt['training'],
                                           unique_consecutive=opt['unique_consecutive'], forgetful_mask=inp['forgetful_mask'], **ckw)
        finally:
            O.semantic_forward = orig
        logits = (captured['l'],)
    elif fx['kind'] == 'coarse':
        orig = O.coarse_forward
        O.coarse_forward = lambda *a, **k: captured.setdefault('l', orig(*a, **k))
        try:
            loss = O.coarse_wrapper_loss(full, cfg, inp['semantic_token_ids'], inp['coarse_token_ids'], training=opt['training'],
                                         unique_consecutive=opt['unique_consecutive'], forgetful_mask=inp['forgetful_mask'], **ckw)
        finally:
            O.coarse_forward = orig
        logits = captured['l']
    else:
        orig = O.fine_forward
        O.fine_forward = lambda *a, **k: captured.setdefault('l', orig(*a, **k))
        try:
            loss = O.fine_wrapper_loss(full, cfg, inp['coarse_token_ids'], inp['fine_token_ids'], forgetful_mask=inp['forgetful_mask'], **ckw)
        finally:
            O.fine_forward = orig
        logits = captured['l']
    loss.backward()
    return loss.detach(), logits, {k: p.grad for k, p in params.items()}


MODEL_FIXTURES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.pt'))
                        if os.path.basename(p).split('_')[0] in ('semantic', 'coarse', 'fine'))


def test_fixtures_present():
    assert len(MODEL_FIXTURES) >= 14, MODEL_FIXTURES
    assert sum('_cond_' in n for n in MODEL_FIXTURES) >= 5


def test_oracle_classifier_free_guidance_matches_reference():
    """forward_with_cond_scale (audiolm_pytorch.py:818-855): eval-mode logits with the conditioning kept and with every text position masked
    (cond_drop_prob = 1 -> keep mask all False), mixed as null + (cond - null) * cond_scale"""
    fx = _load('coarse_s4_cond_cross')
    sd = synth_state_dict(fx['shapes'], fx['seed'])
    cfg, inp, out = _cfg(fx), fx['inputs'], fx['outputs']
    b = inp['semantic_token_ids'].shape[0]
    coarse = inp['coarse_token_ids'].reshape(b, -1)
    with torch.no_grad():
        cs, cc = O.coarse_forward(sd, cfg, inp['semantic_token_ids'], coarse, text_embeds=inp['text_embeds'])
        ns, nc = O.coarse_forward(sd, cfg, inp['semantic_token_ids'], coarse, text_embeds=inp['text_embeds'], cond_drop_keep=torch.zeros(b, dtype=torch.bool))
    w = out['cfg_scale']
    assert torch.allclose(ns + (cs - ns) * w, out['cfg_semantic_logits'], atol=5e-4, rtol=1e-4)
    assert torch.allclose(nc + (cc - nc) * w, out['cfg_coarse_logits'], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize('name', MODEL_FIXTURES)
def test_oracle_matches_reference(name):
    fx = _load(name)
    loss, logits, grads = oracle_run(fx)
    ref = fx['outputs']
    assert abs(float(loss) - float(ref['loss'])) <= 1e-5 * max(1.0, abs(float(ref['loss']))), (float(loss), float(ref['loss']))
    if fx['kind'] == 'semantic':
        got = logits[0].detach()
        want = ref['logits']
        if want.shape != got.shape:             # cfg0 stores a strided subsample
            got = got[:, ::16]
        assert torch.allclose(got, want, atol=2e-4, rtol=1e-4), (got - want).abs().max()
    else:
        keys = ('semantic_logits', 'coarse_logits') if fx['kind'] == 'coarse' else ('coarse_logits', 'fine_logits')
        for g, k in zip(logits, keys):
            assert torch.allclose(g.detach(), ref[k], atol=2e-4, rtol=1e-4), (k, (g.detach() - ref[k]).abs().max())
    for k, dg in ref['grads'].items():
        if dg is None:                             # unused parameter in the reference (e.g. proj_text_embed)
            assert grads.get(k) is None or float(grads[k].abs().max()) == 0.0, k
            continue
        g = grads[k]
        assert g is not None, k
        flat = g.reshape(-1)
        scale = max(float(flat.abs().max()), 1e-6)
        assert abs(float(flat.norm()) - dg['norm']) <= 1e-4 * max(dg['norm'], 1e-6) + 1e-7, (k, float(flat.norm()), dg['norm'])
        assert float((flat[::dg['stride']] - dg['sample']).abs().max()) <= 1e-4 * scale + 1e-7, k
        if dg['full'] is not None:
            assert float((g - dg['full']).abs().max()) <= 1e-4 * scale + 1e-7, k


FULL_FIXTURES = ['full_coarse_s1', 'full_coarse_s4', 'full_fine_s4']


def digest_errors(fx, loss, logits, grads):
    """(loss rel error, [logits sample rel-Frobenius error], {param: (norm rel error, sample rel-Frobenius error)}) of a run against a benchmark-size
    digest fixture of the REAL reference (tests/golden/make_golden.py fullsize)"""
    ref = fx['outputs']
    keys = ('semantic_logits', 'coarse_logits') if fx['kind'] == 'coarse' else ('coarse_logits', 'fine_logits')
    lerr = []
    for g, k in zip(logits, keys):
        d = ref[k]
        assert tuple(g.shape) == tuple(d['shape']), (k, tuple(g.shape), d['shape'])
        smp = g.detach().float().cpu().reshape(-1)[::d['stride']]
        lerr.append(float((smp - d['sample']).norm() / d['sample'].norm()))
    gerr = {}
    for k, dg in ref['grads'].items():
        if dg is None:
            assert grads.get(k) is None or float(grads[k].abs().max()) == 0.0, k
            continue
        assert grads.get(k) is not None, f'missing gradient for {k}'
        if dg['norm'] < 1e-9:
            continue
        flat = grads[k].detach().float().cpu().reshape(-1)
        gerr[k] = (abs(float(flat.norm()) - dg['norm']) / dg['norm'], float((flat[::dg['stride']] - dg['sample']).norm() / dg['sample'].norm().clamp(min=1e-30)))
    rl = float(ref['loss'])
    return abs(float(loss) - rl) / abs(rl), lerr, gerr


@pytest.mark.parametrize('name', FULL_FIXTURES)
def test_oracle_matches_reference_at_benchmark_size(name):
    """The oracle is pinned to the REAL reference at the sizes the benchmark runs (round 3): CoarseTransformer dim=1024 depth=6 N=2048 with 1 and 4
    residual streams, FineTransformer N=2049 -- reference audiolm_pytorch.py:858-990 / :1136-1368 under the wrappers :1742-1854 / :2041-2137.  A
    size-dependent divergence (head regrouping remainder at 1537 = 3 * 512 + 1, offset aliasing at codebook 1024, long softmax rows) would show here.
    fp32 vs fp32 on CPU: loss 1e-5, logits 2e-5 rel-Frobenius on the sample; gradients 2e-3 (hyper-connection scalars are cancelling sums over 2048
    tokens: 3e-4 measured), everything else 1e-4."""
    fx = _load(name)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    loss, logits, grads = oracle_run(fx)
    lrel, lerr, gerr = digest_errors(fx, loss, logits, grads)
    assert lrel <= 1e-5, lrel
    assert max(lerr) <= 2e-5, lerr
    for k, (en, es) in gerr.items():
        tol = 2e-3 if k.endswith(('static_alpha', 'static_beta', 'dynamic_alpha_scale', 'dynamic_beta_scale')) else 1e-4
        assert en <= tol and es <= tol, (k, en, es)


def test_rounding_matched_mode_is_a_small_perturbation_of_the_fp32_oracle():
    """oracle/rounding_matched.py (the restatement at the HIP path's bf16 rounding points) on a golden fixture: it must move the fp32 results by bf16
    noise -- not by zero (rounding really applied) and not by more (same algorithm) -- and leave the module-level hooks restored."""
    import rounding_matched as RM
    for name in ('coarse_s1_flash_uc_mask', 'fine_s4_flash'):
        fx = _load(name)
        l0, lg0, g0 = oracle_run(fx)
        saved = (O.transformer, O._grouped_logits, O._padded_logits, O.head_linear)
        for rb in (False, True):
            with RM.rounding_matched(residual_bf16=rb):
                l1, lg1, g1 = oracle_run(fx)
            assert abs(float(l1) - float(l0)) <= 2e-3 * abs(float(l0))
            for a, b in zip(lg1, lg0):
                e = float((a.detach() - b.detach()).norm() / b.detach().norm())
                assert 1e-3 < e < 3e-2, e
            w = 'transformer.layers.1.2.branch.1.weight'
            assert 1e-3 < float((g1[w] - g0[w]).norm() / g0[w].norm()) < 5e-2
        assert saved == (O.transformer, O._grouped_logits, O._padded_logits, O.head_linear)


def test_attend_matches_reference():
    fx = _load('attend')
    i, o = fx['inputs'], fx['outputs']
    q, k, v, mask, bias = i['q'], i['k'], i['v'], i['mask'], i['bias']
    assert torch.allclose(O.attend(q, k, v), o['math_plain'], atol=1e-6)
    assert torch.allclose(O.attend(q, k, v, mask=mask), o['math_mask'], atol=1e-6)
    assert torch.allclose(O.attend(q, k, v, mask=mask, attn_bias=bias), o['math_mask_bias'], atol=1e-6)
    # the reference's flash (SDPA) path computes the same function (attend.py:69-96)
    assert torch.allclose(O.attend(q, k, v, mask=mask), o['flash_mask'], atol=2e-6)
    assert torch.allclose(O.attend(q, k, v), o['flash_plain'], atol=2e-6)


def test_soundstream_encode_matches_reference():
    fx = _load('soundstream_small')
    sd = synth_state_dict(fx['shapes'], fx['seed'])
    wave = fx['inputs']['wave']
    c = fx['ctor']
    n = (wave.shape[-1] // 320) * 320
    enc = O.soundstream_encoder(sd, wave[:, None, :n], strides=c['strides'])
    assert torch.allclose(enc, fx['outputs']['encoder_out'], atol=1e-5, rtol=1e-5)
    idx = O.soundstream_tokenize(sd, wave, strides=c['strides'], num_quantizers=c['rq_num_quantizers'])
    assert idx.dtype == torch.int64
    assert torch.equal(idx, fx['outputs']['indices'])                       # (b n (g q))
    assert torch.equal(idx[None], fx['outputs']['tokenize'])                # tokenize() -> (g b n q), soundstream.py:847-848


def test_bookkeeping_helpers_bit_exact():
    ids = torch.tensor([[1, 1, 2, 2, 2, 3], [4, 4, 4, 4, 4, 4]])
    uc = O.batch_unique_consecutive(ids, pad_value=-1)
    assert torch.equal(uc, torch.tensor([[1, 2, 3], [4, -1, -1]]))
    assert torch.equal(O.append_eos_id(ids, 9)[:, -1], torch.tensor([9, 9]))
    m = O.generate_mask_with_prob((4, 20), 0.15, 'cpu')
    assert m.dtype == torch.bool and bool(m[:, 0].all()) and int((~m).sum()) == 4 * 3


def test_soundstream_decode_matches_reference():
    """oracle.soundstream_decode_from_indices vs the REAL reference decoder (soundstream.py:691-709, 615-627) around the restated code lookup,
    incl. dropped (-1) quantizer indices."""
    fx = _load('soundstream_decode_small')
    c = fx['ctor']
    sd = synth_state_dict(fx['shapes'], fx['seed'])
    wave = O.soundstream_decode_from_indices(sd, fx['inputs']['indices'], strides=c['strides'], num_quantizers=c['rq_num_quantizers'])
    assert wave.shape == fx['outputs']['wave'].shape
    assert torch.allclose(wave, fx['outputs']['wave'], atol=1e-5, rtol=1e-5)
