"""The reference's kv_cache / embed_cache TENSOR protocol (audiolm_pytorch.py:360-370, :487-496, :560, :719, :938-953, :1300-1315) on the MI355X
against tests/golden/cache_protocol.pt -- the call sequences below run on the REAL reference (tests/golden/make_golden.py cache): a prefix call
that returns the caches, then one-token steps that consume them (the single-position decode kernels), plain and guided ([cond, null] stacks).
Compared per call: logits (same bound as the parity tests: bf16 GEMM operands vs the fp32 reference), cache SHAPES exactly; at the end the
cache VALUES (bf16-rounded keys / values, fp32 hidden states)."""
import os

import pytest
import torch

from common import GOLDEN_DIR, synth_state_dict

pytestmark = pytest.mark.gpu

LOGITS_TOL = 1.5e-2         # rel. Frobenius; the reference's own bf16-autocast logits deviate 0.7-1.9e-2 on these toy models (bf16_noise.pt)
CACHE_TOL = 1.5e-2


def _frob(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def _fx():
    return torch.load(os.path.join(GOLDEN_DIR, 'cache_protocol.pt'), weights_only=False)['models']


def _build(K, m):
    dev = torch.device('cuda:0')
    model = K(**m['ctor'])
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v) for k, v in m['shapes'].items()}
    model.load_state_dict(synth_state_dict(m['shapes'], m['seed']))
    return model.to(dev).eval(), dev


def test_semantic_kv_cache_protocol():
    import audiolm_pytorch_amd as A
    m = _fx()['semantic']
    model, dev = _build(A.SemanticTransformer, m)
    ids, kv = m['ids'].to(dev), None
    with torch.no_grad():
        for c in m['calls']:
            lg, kv = model(ids=ids[:, :c['n']], kv_cache=kv, return_kv_cache=True)
            assert tuple(kv.shape) == c['kv_shape'] and tuple(lg.shape) == tuple(c['logits'].shape)      # cached call: the NEW position's logits only
            assert _frob(lg, c['logits']) <= LOGITS_TOL, (c['n'], _frob(lg, c['logits']))
        assert _frob(kv, m['kv']) <= CACHE_TOL
        # more than one new position against a cache, and a cache that already covers everything but one token of a longer prefix
        lg_full = model(ids=ids)
        lg2, kv2 = model(ids=ids, kv_cache=kv[..., :6, :], return_kv_cache=True)
        assert tuple(lg2.shape) == (ids.shape[0], ids.shape[1] + 1 - 6, lg_full.shape[-1]) and tuple(kv2.shape) == tuple(kv.shape)
        assert _frob(lg2, lg_full[:, 6:]) <= 2e-3
        # the uncached forward() is untouched by the protocol
        assert _frob(lg_full[:, -1:], m['calls'][-1]['logits']) <= LOGITS_TOL


def test_coarse_kv_and_embed_cache_protocol():
    import audiolm_pytorch_amd as A
    m = _fx()['coarse']
    model, dev = _build(A.CoarseTransformer, m)
    sem, coarse, kv, em = m['sem'].to(dev), m['coarse'].to(dev), None, None
    with torch.no_grad():
        for c in m['calls']:
            (sl, cl), (kv, em) = model(semantic_token_ids=sem, coarse_token_ids=coarse[:, :c['n']], kv_cache=kv, embed_cache=em, return_cache=True)
            assert tuple(kv.shape) == c['kv_shape'] and tuple(em.shape) == c['embed_shape']
            assert tuple(cl.shape) == tuple(c['coarse_logits'].shape) and tuple(sl.shape) == tuple(c['semantic_logits'].shape)   # ALL positions (embed cache)
            assert _frob(cl, c['coarse_logits']) <= LOGITS_TOL and _frob(sl, c['semantic_logits']) <= LOGITS_TOL, (c['n'], _frob(cl, c['coarse_logits']))
        assert _frob(kv, m['kv']) <= CACHE_TOL and _frob(em, m['embed']) <= CACHE_TOL
        with pytest.raises(AssertionError):
            model(semantic_token_ids=sem, coarse_token_ids=coarse[:, :7], kv_cache=kv[..., :12, :], return_cache=True)        # embed_cache of the same call missing


def test_fine_guided_stacked_cache_protocol():
    import audiolm_pytorch_amd as A
    m = _fx()['fine_guided']
    model, dev = _build(A.FineTransformer, m)
    coarse, fine, te, kv, em = m['coarse'].to(dev), m['fine'].to(dev), m['text_embeds'].to(dev), None, None
    with torch.no_grad():
        for c in m['calls']:
            (cl, fl), (kv, em) = model.forward_with_cond_scale(coarse, fine[:, :c['n']], text_embeds=te, cond_scale=m['cond_scale'], kv_cache=kv,
                                                               embed_cache=em, return_kv_cache=True)
            assert tuple(kv.shape) == c['kv_shape'] and tuple(em.shape) == c['embed_shape']                   # [cond | null] stacks
            assert tuple(fl.shape) == tuple(c['fine_logits'].shape) and tuple(cl.shape) == tuple(c['coarse_logits'].shape)
            # guided logits = null + 3 (cond - null): the difference of two bf16-noisy passes is amplified 3x
            assert _frob(fl, c['fine_logits']) <= 3 * LOGITS_TOL and _frob(cl, c['coarse_logits']) <= 3 * LOGITS_TOL, (c['n'], _frob(fl, c['fine_logits']))
        assert _frob(kv, m['kv']) <= CACHE_TOL and _frob(em, m['embed']) <= CACHE_TOL
        # without caches the guided logits are the same numbers
        cl0, fl0 = model.forward_with_cond_scale(coarse, fine[:, :m['calls'][-1]['n']], text_embeds=te, cond_scale=m['cond_scale'])
        assert _frob(fl0, fl) <= 2e-2 and _frob(cl0, cl) <= 2e-2


def test_dense_attn_bias_through_the_cache_protocol_equals_the_uncached_forward():
    """Transformer.forward(attn_bias=<dense tensor>, kv_cache=..., return_kv_cache=True) (reference :487-506 with any bias tensor): the math path has no
    single-position kernel, so the protocol recomputes the sequence; the positions it returns and the cache it hands back must equal the uncached
    forward / a structured run of the same model (round 3: this combination used to be refused)."""
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    dim, depth, heads, n, b = 128, 2, 4, 40, 2
    torch.manual_seed(0)
    tr = A.audiolm_pytorch.Transformer(dim=dim, depth=depth, heads=heads, num_residual_streams=4, rel_pos_bias=False)
    tr.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in tr.state_dict().items()}, 401))
    tr.to(dev).eval()
    g = torch.Generator().manual_seed(402)
    x = torch.randn(b, n, dim, generator=g).to(dev)
    bias = (torch.randn(heads, n, n, generator=g) * 1.5).to(dev)
    with torch.no_grad():
        full = tr(x, attn_bias=bias)
        h0, kv0 = tr(x[:, :n - 1], attn_bias=bias[:, :n - 1, :n - 1], return_kv_cache=True)
        h1, kv1 = tr(x, attn_bias=bias, kv_cache=kv0, return_kv_cache=True)
    assert tuple(kv0.shape) == (depth, 2, b, n - 1, 64) and tuple(kv1.shape) == (depth, 2, b, n, 64) and tuple(h1.shape) == (b, 1, dim)
    assert _frob(h0, full[:, :n - 1]) <= 2e-3 and _frob(h1, full[:, -1:]) <= 2e-3, (_frob(h0, full[:, :n - 1]), _frob(h1, full[:, -1:]))
    assert _frob(kv1[..., :n - 1, :], kv0) <= 2e-3
