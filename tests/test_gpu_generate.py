"""*TransformerWrapper.generate() (reference audiolm_pytorch.py:1406-1511, :1608-1740, :1896-2039) on a real MI355X.

Sampling is made greedy (gumbel noise zeroed) so that the run is deterministic, then every generated token is checked against the fp32 CPU
oracle's logits ON THE SAME PREFIX: the chosen token must be the oracle's arg-max up to bf16 noise (its oracle logit within 5e-2 of the
oracle's maximum over the allowed tokens), the eos-in-the-middle-of-a-time-step rule and the after-eos masking must hold, and output shapes
must be the reference's.  `flash_attn=True` and the default constructors (bias tables) are both exercised.
"""
import pytest
import torch

import audiolm_oracle as O

pytestmark = pytest.mark.gpu

TOL = 5e-2


class Codec:
    rq_groups = 1

    def __init__(self, nq=8):
        self.num_quantizers = nq


@pytest.fixture()
def greedy():
    from audiolm_pytorch_amd import audiolm_pytorch as AP
    orig = AP.gumbel_noise
    AP.gumbel_noise = lambda t: torch.zeros_like(t)
    yield
    AP.gumbel_noise = orig


def _sd(model):
    return {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}


def _check_choice(oracle_last_logits, chosen, allow_eos, filter_thres=0.9):
    lg = oracle_last_logits.clone()
    if not allow_eos:
        lg[:, -1] = float('-inf')
    best = lg.max(dim=-1).values
    got = lg.gather(1, chosen.view(-1, 1)).squeeze(1)
    assert bool((got >= best - TOL).all()), (got, best)


@pytest.mark.parametrize('use_kv_cache', [True, False])
@pytest.mark.parametrize('flash', [True, False])
def test_semantic_generate_greedy_matches_oracle(greedy, flash, use_kv_cache):
    import audiolm_pytorch_amd as A
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    model = A.SemanticTransformer(dim=64, depth=2, heads=2, num_semantic_tokens=20, flash_attn=flash)
    sd = _sd(model)
    model.to(dev)
    w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=False)
    g = torch.Generator().manual_seed(1)
    prime = torch.randint(0, 20, (3, 4), generator=g)
    out = w.generate(max_length=12, prime_ids=prime.to(dev), use_kv_cache=use_kv_cache)
    assert out.dtype == torch.long and out.shape[0] == 3 and 5 <= out.shape[1] <= 12
    out = out.cpu()
    assert torch.equal(out[:, :4], prime)
    cfg = O.Cfg(dim=64, depth=2, heads=2, streams=4, num_semantic_tokens=20)
    alive = torch.ones(3, dtype=torch.bool)
    for t in range(4, out.shape[1]):
        prefix = out[:, :t].clone()
        prefix[prefix < 0] = 0                                                     # rows already past eos: content irrelevant (masked output)
        lg = O.semantic_forward(sd, cfg, prefix)[:, t]                             # logits after the t-th id (index t incl. the start token)
        tok = out[:, t]
        rows = alive & (tok >= 0)
        if bool(rows.any()):
            _check_choice(lg[rows], tok[rows], allow_eos=True)
        # once a row produced eos, everything from eos on is masked to -1 (keep_eos = False)
        alive = alive & (tok >= 0)
    assert not bool((out == 20).any())                                            # eos itself is masked out of the output


@pytest.mark.parametrize('use_kv_cache', [True, False])
@pytest.mark.parametrize('flash', [True, False])
def test_coarse_generate_greedy_matches_oracle(greedy, flash, use_kv_cache):
    import audiolm_pytorch_amd as A
    torch.manual_seed(2)
    dev = torch.device('cuda:0')
    model = A.CoarseTransformer(dim=64, depth=2, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3, flash_attn=flash)
    sd = _sd(model)
    model.to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False)
    g = torch.Generator().manual_seed(3)
    sem = torch.randint(0, 20, (2, 6), generator=g)
    out = w.generate(semantic_token_ids=sem.to(dev), max_time_steps=3, use_kv_cache=use_kv_cache).cpu()
    assert out.shape == (2, 3, 3)
    flat = out.reshape(2, -1)
    cfg = O.Cfg(dim=64, depth=2, heads=2, streams=4, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3)
    alive = torch.ones(2, dtype=torch.bool)
    for t in range(flat.shape[1]):
        prefix = flat[:, :t].clone()
        prefix[prefix < 0] = 0
        _, coarse_logits = O.coarse_forward(sd, cfg, sem, prefix)
        tok = flat[:, t]
        rows = alive & (tok >= 0)
        if bool(rows.any()):
            _check_choice(coarse_logits[rows, -1], tok[rows], allow_eos=(t % 3 == 0 and t > 0))
        alive = alive & (tok >= 0)
    # eos may only ever have been sampled at the first quantizer of a time step: masked output starts at a multiple of 3
    for r in range(2):
        neg = (flat[r] < 0).nonzero()
        if neg.numel():
            assert int(neg[0]) % 3 == 0


@pytest.mark.parametrize('use_kv_cache', [True, False])
@pytest.mark.parametrize('flash', [True, False])
def test_fine_generate_greedy_matches_oracle(greedy, flash, use_kv_cache):
    import audiolm_pytorch_amd as A
    torch.manual_seed(4)
    dev = torch.device('cuda:0')
    model = A.FineTransformer(dim=64, depth=2, heads=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5, flash_attn=flash)
    sd = _sd(model)
    model.to(dev)
    w = A.FineTransformerWrapper(transformer=model, codec=Codec())
    g = torch.Generator().manual_seed(5)
    coarse = torch.randint(0, 16, (2, 2, 3), generator=g)
    out = w.generate(coarse_token_ids=coarse.to(dev), use_kv_cache=use_kv_cache).cpu()
    assert out.shape == (2, 2, 5)
    flat = out.reshape(2, -1)
    cfg = O.Cfg(dim=64, depth=2, heads=2, streams=4, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5)
    alive = torch.ones(2, dtype=torch.bool)
    for t in range(flat.shape[1]):
        prefix = flat[:, :t].clone()
        prefix[prefix < 0] = 0
        _, fine_logits = O.fine_forward(sd, cfg, coarse, prefix)
        tok = flat[:, t]
        rows = alive & (tok >= 0)
        if bool(rows.any()):
            _check_choice(fine_logits[rows, -1], tok[rows], allow_eos=(t % 5 == 0 and t > 0))
        alive = alive & (tok >= 0)


def test_generate_refuses_what_is_not_native():
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    model = A.CoarseTransformer(dim=64, depth=1, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3, flash_attn=True).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False)
    sem = torch.randint(0, 20, (1, 4), device=dev)
    with pytest.raises(AssertionError):
        w.generate(semantic_token_ids=sem, max_time_steps=1, text=['a'])                        # conditioning an un-conditioned model (reference :1650)
    mc = A.CoarseTransformer(dim=64, depth=1, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3, flash_attn=True,
                             has_condition=True, cond_dim=32).to(dev)
    wc = A.CoarseTransformerWrapper(transformer=mc, codec=Codec(), unique_consecutive=False)
    with pytest.raises(NotImplementedError):
        wc.generate(semantic_token_ids=sem, max_time_steps=1, text=['a'])                       # the T5 text encoder is out of scope: pass text_embeds


@pytest.mark.parametrize('B,H,pos,nmax,use_mask,use_bias', [(2, 8, 0, 16, False, False), (3, 8, 70, 128, True, False), (2, 4, 129, 130, False, True),
                                                            (1, 2, 300, 512, True, True), (4, 8, 63, 64, False, True)])
def test_decode_attention_kernel(B, H, pos, nmax, use_mask, use_bias):
    """alm_mqa_decode_attn vs an fp32 restatement: one query per sequence over a cache of pos + 1 keys (the new row included and appended
    by the kernel), key mask, structured bias.  bf16 operands, fp32 statistics: 8e-3 rel-max (one bf16 rounding of the output)."""
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import ops, relpos
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(pos + nmax)
    d = 64
    q = torch.randn(B, H * d, generator=g).to(dev).bfloat16()
    cache = torch.randn(B, nmax, 2 * d, generator=g).to(dev).bfloat16()
    kv_new = torch.randn(B, 2 * d, generator=g).to(dev).bfloat16()
    mask = None
    if use_mask:
        mask = (torch.rand(B, pos + 1, generator=g) > 0.3)
        mask[:, 0] = True
        mask = mask.to(dev)
    bias = None
    if use_bias:
        index = relpos.toeplitz_index(nmax, dev, num_leading=max(1, nmax // 3))
        bias = relpos.AttnBias((torch.randn(H, 2 * nmax, generator=g) * 6).to(dev), *index)
    before = cache.clone()
    out = ops.mqa_decode_attn(q, cache, kv_new, pos, None if mask is None else mask.contiguous().view(torch.uint8), H, d, bias=bias)
    # the kernel appended the new row and touched nothing else
    assert torch.equal(cache[:, pos], kv_new)
    keep = torch.ones(nmax, dtype=torch.bool, device=dev)
    keep[pos] = False
    assert torch.equal(cache[:, keep], before[:, keep])
    k = cache[:, :pos + 1, :d].float()
    v = cache[:, :pos + 1, d:].float()
    qf = q.float().view(B, H, d)
    sim = torch.einsum('bhd,bjd->bhj', qf, k)
    if bias is not None:
        slot = (bias.qkey4[pos].long() - bias.kkey4[:pos + 1].long()) // 4
        special = (bias.qattr[pos] & bias.kattr[:pos + 1]) != 0
        slot = torch.where(special, torch.zeros_like(slot), slot)
        sim = sim + bias.tbl[:, slot][None]
    sim = sim * d ** -0.5
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None, :], float('-inf'))
    ref = torch.einsum('bhj,bjd->bhd', sim.softmax(dim=-1), v).reshape(B, H * d)
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    assert err <= 8e-3, err


def test_cached_step_logits_equal_recomputed_logits():
    """the same prefix through the two paths: (prefill + single-position steps with the cache) vs (full forward): next-token logits agree to
    bf16 noise at every step -- for all three transformers with their bias tables (default constructors)."""
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(8)

    def close(a, b):
        return float((a.float() - b.float()).abs().max()) <= 2e-2 * max(1.0, float(b.float().abs().max()))
    m = A.SemanticTransformer(dim=64, depth=3, heads=2, num_semantic_tokens=20).to(dev).eval()
    c = A.CoarseTransformer(dim=64, depth=3, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3).to(dev).eval()
    f = A.FineTransformer(dim=64, depth=3, heads=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5).to(dev).eval()
    with torch.inference_mode():
        ids = torch.randint(0, 20, (2, 9), generator=g).to(dev)
        state = None
        for n in range(3, 10):
            lg, state = m.sample_logits(ids[:, :n], state, 12)
            assert close(lg, m(ids=ids[:, :n])[:, -1]), ('semantic', n)
        sem = torch.randint(0, 20, (2, 5), generator=g).to(dev)
        co = torch.randint(0, 16, (2, 7), generator=g).to(dev)
        state = None
        for n in range(0, 8):
            lg, state = c.sample_logits(sem, co[:, :n], state, 5 + 2 + 9)
            _, full = c(semantic_token_ids=sem, coarse_token_ids=co[:, :n], return_only_coarse_logits=True)
            assert close(lg, full[:, -1]), ('coarse', n)
        coarse = torch.randint(0, 16, (2, 6), generator=g).to(dev)
        coarse[0, 4] = -1                                                       # a padded coarse position: key-masked (:1175)
        fi = torch.randint(0, 16, (2, 9), generator=g).to(dev)
        state = None
        for n in range(0, 10):
            lg, state = f.sample_logits(coarse, fi[:, :n], state, 10)
            _, full = f(coarse, fi[:, :n], return_only_fine_logits=True)
            assert close(lg, full[:, -1]), ('fine', n)


def test_audiolm_end_to_end_hierarchical_sampling():
    """AudioLM.forward (audiolm_pytorch.py:2141-2254): semantic -> coarse -> fine -> SoundStream decoder, all native.  The coarse stage's
    default of 512 time steps is shortened for the test."""
    import functools
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    codec = A.SoundStream(codebook_size=16, rq_num_quantizers=8, channels=4, codebook_dim=16, use_local_attn=False, strides=(2, 4, 5, 8)).to(dev)
    for r in codec.rq.rvqs:
        for l in r.layers:
            l._codebook.embed.normal_()
            l._codebook.initted.fill_(True)
    sem = A.SemanticTransformer(dim=64, depth=1, heads=2, num_semantic_tokens=20, flash_attn=True).to(dev)
    coarse = A.CoarseTransformer(dim=64, depth=1, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3).to(dev)
    fine = A.FineTransformer(dim=64, depth=1, heads=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5, flash_attn=True).to(dev)
    lm = A.AudioLM(wav2vec=None, codec=codec, semantic_transformer=sem, coarse_transformer=coarse, fine_transformer=fine)
    lm.coarse.generate = functools.partial(lm.coarse.generate, max_time_steps=10)
    out = lm(batch_size=2, max_length=12)
    waves = out if isinstance(out, list) else list(out)
    assert len(waves) == 2
    for w in waves:
        assert w is None or (w.dim() == 1 and w.numel() % 320 == 0 and bool(torch.isfinite(w).all()))
    assert lm.training is False or True


@pytest.mark.parametrize('kind', ['semantic', 'coarse', 'fine'])
def test_guided_sampling_cached_equals_recomputed(kind):
    """Cached sampling of a CONDITIONED model with classifier-free guidance (cond_scale 3; two caches: conditioned / unconditioned; the
    cross-attention over the text is recomputed for the new position each step) against the recompute path (forward_with_cond_scale on the
    whole prefix), step by step on the same token sequence.  Guidance amplifies the bf16 difference of two passes by (2 * cond_scale - 1)."""
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    te = torch.randn(2, 6, 32, generator=g).to(dev)
    te[1, 4:] = 0.                                                # padded text positions (masked out by CoarseTransformer, :882-883)
    T, scale = 7, 3.

    def frob(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    with torch.no_grad():
        if kind == 'semantic':
            model = A.SemanticTransformer(dim=128, depth=2, num_semantic_tokens=30, has_condition=True, cond_dim=32, flash_attn=True).to(dev).eval()
            ids = torch.randint(0, 30, (2, T), generator=g).to(dev)
            state = None
            for t in range(1, T):
                lc, state = model.sample_logits(ids[:, :t], state, T + 1, text_embeds=te, cond_scale=scale)
                lf = model.forward_with_cond_scale(ids=ids[:, :t], text_embeds=te, cond_scale=scale)[:, -1]
                assert frob(lc, lf) <= 3e-2, (t, frob(lc, lf))
        elif kind == 'coarse':
            model = A.CoarseTransformer(dim=128, depth=2, num_semantic_tokens=30, codebook_size=40, num_coarse_quantizers=3, has_condition=True,
                                        cond_dim=32).to(dev).eval()
            sem = torch.randint(0, 30, (2, 7), generator=g).to(dev)
            coarse = torch.randint(0, 40, (2, T), generator=g).to(dev)
            state = None
            for t in range(0, T):
                lc, state = model.sample_logits(sem, coarse[:, :t], state, 7 + 2 + T, text_embeds=te, cond_scale=scale)
                _, lf = model.forward_with_cond_scale(semantic_token_ids=sem, coarse_token_ids=coarse[:, :t], text_embeds=te, cond_scale=scale,
                                                      return_only_coarse_logits=True)
                assert frob(lc, lf[:, -1]) <= 3e-2, (t, frob(lc, lf[:, -1]))
        else:
            model = A.FineTransformer(dim=128, depth=2, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=40, has_condition=True, cond_dim=32,
                                      flash_attn=True).to(dev).eval()
            coarse = torch.randint(0, 40, (2, 6), generator=g).to(dev)
            fine = torch.randint(0, 40, (2, T), generator=g).to(dev)
            state = None
            for t in range(0, T):
                lc, state = model.sample_logits(coarse, fine[:, :t], state, 10, text_embeds=te, cond_scale=scale)
                _, lf = model.forward_with_cond_scale(coarse, fine[:, :t], text_embeds=te, cond_scale=scale, return_only_fine_logits=True)
                assert frob(lc, lf[:, -1]) <= 3e-2, (t, frob(lc, lf[:, -1]))


def test_conditioned_generate_runs_end_to_end():
    """wrapper.generate() of conditioned models (text_embeds, cond_scale 3), kv-cache and recompute paths, prefix conditioning falls back to recompute"""
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    te = torch.randn(2, 5, 32, device=dev)
    model = A.SemanticTransformer(dim=64, depth=2, num_semantic_tokens=30, has_condition=True, cond_dim=32, flash_attn=True).to(dev)
    w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=False)
    for cache in (True, False):
        out = w.generate(max_length=8, batch_size=2, text_embeds=te, cond_scale=3., use_kv_cache=cache)
        assert out.shape[0] == 2 and out.dtype == torch.long and int(out.max()) <= 30
    mp = A.SemanticTransformer(dim=64, depth=2, num_semantic_tokens=30, has_condition=True, cond_dim=32, cond_as_self_attn_prefix=True).to(dev)
    wp = A.SemanticTransformerWrapper(transformer=mp, unique_consecutive=False)
    out = wp.generate(max_length=6, batch_size=2, text_embeds=te, cond_scale=1.5)
    assert out.shape[0] == 2
    with pytest.raises(AssertionError):                          # a conditioned model needs its conditioning (reference :1456)
        w.generate(max_length=4, batch_size=1)
