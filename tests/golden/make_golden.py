"""Generates tests/golden/*.pt by running the REAL reference (/root/reference, imported under
oracle/ref_shims.py) on CPU in fp32.  Build-container only (the reference cannot travel); the
resulting fixtures are committed.  Re-run:  python tests/golden/make_golden.py

Each fixture: dict(case=<ctor kwargs / options>, shapes={state_dict key: shape}, seed, inputs,
outputs(loss / logits / grad digests)).  Parameter VALUES are re-synthesised from (shapes, seed)
by tests/golden/common.py on both sides.

Cases with num_residual_streams > 1 execute the reference's own code around the RESTATED
hyper-connections module (third-party, not vendored) -> labelled restated=True.
"""
from __future__ import annotations

import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
from common import synth_state_dict, grad_digest  # noqa: E402

warnings.filterwarnings('ignore')
A, S, AT = ref_shims.load_reference()


class _Codec:   # the wrappers dereference codec.rq_groups / num_quantizers unconditionally (audiolm_pytorch.py:1598, 1877-1881)
    rq_groups = 1

    def __init__(self, num_quantizers=8):
        self.num_quantizers = num_quantizers


def _shapes(sd):
    return {k: tuple(v.shape) for k, v in sd.items()}


def _forgetful(shape, prob, seed):
    g = torch.Generator().manual_seed(seed)
    seq = shape[-1]
    rand = torch.randn(shape, generator=g)
    rand[:, 0] = -torch.finfo(rand.dtype).max
    num_mask = min(int(seq * prob), seq - 1)
    idx = rand.topk(num_mask, dim=-1).indices
    return ~torch.zeros(shape).scatter(1, idx, 1.).bool()


def _run(model, wrapper, wrapper_kwargs, training, mask, cond_keep=None):
    """forward(return_loss=True) + backward on the real reference.  `mask` replaces the RNG draw of the forgetful mask, `cond_keep` (bool (b,))
    the `prob_mask_like` draw of the per-sample condition dropping (audiolm_pytorch.py:703 / :890 / :1167)."""
    wrapper.train(training)
    model.zero_grad(set_to_none=True)
    orig = A.generate_mask_with_prob
    orig_pml = A.prob_mask_like
    seen = {}

    def fake_pml(shape, prob, device):
        assert cond_keep is not None and tuple(shape) == tuple(cond_keep.shape), (shape, cond_keep)
        seen['cond_prob'] = prob
        return cond_keep.clone()
    if cond_keep is not None:
        A.prob_mask_like = fake_pml

    def fake(shape, prob, device):
        seen['shape'] = tuple(shape)
        assert mask is not None and tuple(mask.shape) == tuple(shape), (mask is None, shape)
        return mask.clone()

    A.generate_mask_with_prob = fake
    hook = model.register_forward_hook(lambda m, a, out: seen.__setitem__('logits', out))
    try:
        loss = wrapper(**wrapper_kwargs, return_loss=True)
    finally:
        A.generate_mask_with_prob = orig
        A.prob_mask_like = orig_pml
        hook.remove()
    logits = seen['logits']
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    return loss.detach(), logits, grads


def _cond_inputs(cond, b, seed):
    """cond = dict(dim, m, drop (bool: inject a keep mask), zero_rows (padding positions))"""
    if cond is None:
        return {}, None, {}
    g = torch.Generator().manual_seed(seed + 500)
    te = torch.randn(b, cond['m'], cond['dim'], generator=g)
    for (bi, mi) in cond.get('zero_rows', ()):
        te[bi, mi] = 0.                                    # all-zero embedding = padding position (text_mask, :883)
    keep = None
    if cond.get('drop'):
        keep = torch.rand(b, generator=g) < 0.5
        keep[0], keep[-1] = True, False                    # both branches present
    return dict(text_embeds=te), keep, dict(text_embeds=te, cond_keep=keep)


def semantic_case(name, *, ctor, ids, training=True, unique_consecutive=True, mask_prob=0., seed=1, full=True, cond=None):
    torch.manual_seed(0)
    model = A.SemanticTransformer(**ctor)
    shapes = _shapes(model.state_dict())
    model.load_state_dict(synth_state_dict(shapes, seed))
    wrapper = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=unique_consecutive, mask_prob=mask_prob)
    mask = None
    if mask_prob > 0 and training:
        import audiolm_oracle as O
        inp, _ = O.semantic_wrapper_bookkeeping(ids, model.eos_id, training=training, unique_consecutive=unique_consecutive)
        mask = _forgetful(tuple(inp.shape), mask_prob, seed + 100)
    ckw, keep, cin = _cond_inputs(cond, ids.shape[0], seed)
    loss, logits, grads = _run(model, wrapper, dict(semantic_token_ids=ids, **ckw), training, mask, keep)
    return dict(name=name, kind='semantic', ctor=ctor, shapes=shapes, seed=seed, restated=ctor.get('num_residual_streams', 4) > 1,
                options=dict(training=training, unique_consecutive=unique_consecutive, mask_prob=mask_prob),
                inputs=dict(ids=ids, forgetful_mask=mask, **cin),
                outputs=dict(loss=loss, logits=logits.detach() if full else logits.detach()[:, ::16].clone(),
                             grads=grad_digest(grads, full)))


def coarse_case(name, *, ctor, sem, coarse, training=True, unique_consecutive=True, mask_prob=0., seed=2, full=True, cond=None, cfg_scale=None):
    torch.manual_seed(0)
    model = A.CoarseTransformer(**ctor)
    shapes = _shapes(model.state_dict())
    model.load_state_dict(synth_state_dict(shapes, seed))
    wrapper = A.CoarseTransformerWrapper(transformer=model, codec=_Codec(), unique_consecutive=unique_consecutive, mask_prob=mask_prob)
    mask = None
    if mask_prob > 0 and training:
        import audiolm_oracle as O
        *_, km = O.coarse_wrapper_bookkeeping(sem, coarse, model.semantic_eos_id, model.coarse_eos_id, training=training,
                                              unique_consecutive=unique_consecutive)
        mask = _forgetful(tuple(km.shape), mask_prob, seed + 100)
    ckw, keep, cin = _cond_inputs(cond, sem.shape[0], seed)
    loss, logits, grads = _run(model, wrapper, dict(semantic_token_ids=sem, coarse_token_ids=coarse, **ckw), training, mask, keep)
    sl, cl = logits
    out = dict(loss=loss, semantic_logits=sl.detach(), coarse_logits=cl.detach(), grads=grad_digest(grads, full))
    if cfg_scale is not None:                                # classifier-free guidance, eval mode (audiolm_pytorch.py:818-855)
        model.eval()
        with torch.no_grad():
            gs, gc = model.forward_with_cond_scale(semantic_token_ids=sem, coarse_token_ids=coarse.reshape(coarse.shape[0], -1), cond_scale=cfg_scale, **ckw)
        out.update(cfg_scale=cfg_scale, cfg_semantic_logits=gs, cfg_coarse_logits=gc)
    return dict(name=name, kind='coarse', ctor=ctor, shapes=shapes, seed=seed, restated=ctor.get('num_residual_streams', 4) > 1,
                options=dict(training=training, unique_consecutive=unique_consecutive, mask_prob=mask_prob),
                inputs=dict(semantic_token_ids=sem, coarse_token_ids=coarse, forgetful_mask=mask, **cin),
                outputs=out)


def fine_case(name, *, ctor, coarse, fine, training=True, mask_prob=0., seed=3, full=True, cond=None):
    torch.manual_seed(0)
    model = A.FineTransformer(**ctor)
    shapes = _shapes(model.state_dict())
    model.load_state_dict(synth_state_dict(shapes, seed))
    nq = ctor['num_coarse_quantizers'] + ctor['num_fine_quantizers']
    wrapper = A.FineTransformerWrapper(transformer=model, codec=_Codec(nq), mask_prob=mask_prob)
    mask = None
    if mask_prob > 0 and training:
        b = coarse.shape[0]
        mask = _forgetful((b, coarse.reshape(b, -1).shape[1] + fine.reshape(b, -1).shape[1] - 1 + 2), mask_prob, seed + 100)
    ckw, keep, cin = _cond_inputs(cond, coarse.shape[0], seed)
    loss, logits, grads = _run(model, wrapper, dict(coarse_token_ids=coarse, fine_token_ids=fine, **ckw), training, mask, keep)
    cl, fl = logits
    return dict(name=name, kind='fine', ctor=ctor, shapes=shapes, seed=seed, restated=ctor.get('num_residual_streams', 4) > 1,
                options=dict(training=training, mask_prob=mask_prob),
                inputs=dict(coarse_token_ids=coarse, fine_token_ids=fine, forgetful_mask=mask, **cin),
                outputs=dict(loss=loss, coarse_logits=cl.detach(), fine_logits=fl.detach(), grads=grad_digest(grads, full)))


def logits_digest(t, n=32768):
    """strided sample + norm of a big logits tensor (full-size fixtures stay small)"""
    flat = t.detach().float().reshape(-1)
    stride = max(1, flat.numel() // n)
    if stride > 1 and stride % 2 == 0:
        stride += 1                                           # odd stride: the sample walks through every class column
    return dict(shape=tuple(t.shape), norm=float(flat.norm()), stride=stride, sample=flat[::stride].clone())


def fullsize_case(name, kind, streams):
    """BENCHMARK-SIZE fixtures of the REAL reference in digest form (round 3): CoarseTransformer dim=1024 depth=6 at N=2048 (BASELINE headline /
    configs[3] shape, B=1) and FineTransformer 3 + 5 quantizers at N=2049 (configs[2]), reference forward :858-990 / :1136-1368 under the wrappers
    :1742-1854 / :2041-2137, forgetful mask 0.15 injected.  Stored: loss, strided logits sample + norm, per-parameter gradient norm + strided
    sample, and the reference's OWN bf16-autocast deviation (loss / logits / per-tensor gradients) on the same inputs.  The inputs are the ones
    tests/test_gpu_fullsize.py::_case builds (same generator seed)."""
    import audiolm_oracle as O
    g = torch.Generator().manual_seed(1234)
    extra = {} if streams == 4 else dict(num_residual_streams=streams)
    if kind == 'coarse':
        ctor = dict(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True, **extra)
        sem, coarse = torch.randint(0, 500, (1, 509), generator=g), torch.randint(0, 1024, (1, 512, 3), generator=g)
        N = 1 + 510 + 1 + 1536
        mask = O.generate_mask_with_prob((1, N), 0.15, 'cpu', generator=g)
        K, seed = A.CoarseTransformer, 4242 + streams
        inputs = dict(semantic_token_ids=sem, coarse_token_ids=coarse, forgetful_mask=mask)
        options = dict(training=True, unique_consecutive=False, mask_prob=0.15)
        mkw = lambda m: (A.CoarseTransformerWrapper(transformer=m, codec=_Codec(), unique_consecutive=False, mask_prob=0.15),
                         dict(semantic_token_ids=sem, coarse_token_ids=coarse))
        keys = ('semantic_logits', 'coarse_logits')
    else:
        ctor = dict(dim=1024, depth=6, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024, flash_attn=True, **extra)
        grid = torch.randint(0, 1024, (1, 256, 8), generator=g)
        coarse, fine = grid[..., :3].contiguous(), grid[..., 3:].contiguous()
        N = 1 + 768 + 1 + 1279
        mask = O.generate_mask_with_prob((1, N), 0.15, 'cpu', generator=g)
        K, seed = A.FineTransformer, 4242 + streams
        inputs = dict(coarse_token_ids=coarse, fine_token_ids=fine, forgetful_mask=mask)
        options = dict(training=True, mask_prob=0.15)
        mkw = lambda m: (A.FineTransformerWrapper(transformer=m, codec=_Codec(8), mask_prob=0.15), dict(coarse_token_ids=coarse, fine_token_ids=fine))
        keys = ('coarse_logits', 'fine_logits')
    torch.manual_seed(0)
    model = K(**ctor)
    shapes = _shapes(model.state_dict())
    model.load_state_dict(synth_state_dict(shapes, seed))
    wrapper, kw = mkw(model)
    import time
    t0 = time.time()
    loss, logits, grads = _run(model, wrapper, kw, True, mask)
    t_ref = time.time() - t0
    out = dict(loss=loss, grads=grad_digest(grads, False), ref_seconds=t_ref)
    for k, t in zip(keys, logits):
        out[k] = logits_digest(t)
    g32 = {k: (v.detach().float().clone() if v is not None else None) for k, v in grads.items()}
    l32 = [t.detach().float().clone() for t in logits]
    with torch.autocast('cpu', dtype=torch.bfloat16):
        nloss, nlogits, ngrads = _run(model, wrapper, kw, True, mask)
    noise = dict(loss_abs=abs(float(nloss) - float(loss)),
                 logits=[float((a.detach().float() - b).norm() / b.norm()) for a, b in zip(nlogits, l32)],
                 grads={k: float((ngrads[k].float() - v).norm() / v.norm()) for k, v in g32.items() if v is not None and float(v.norm()) >= 1e-9})
    return dict(name=name, kind=kind, ctor=ctor, shapes=shapes, seed=seed, restated=streams > 1, options=options, inputs=inputs, outputs=out,
                noise=noise, N=N)


def attend_case():
    g = torch.Generator().manual_seed(7)
    b, h, n, d = 2, 4, 37, 16
    q = torch.randn(b, h, n, d, generator=g)
    k = torch.randn(b, n, d, generator=g)
    v = torch.randn(b, n, d, generator=g)
    mask = torch.rand(b, n, generator=g) > 0.2
    mask[:, 0] = True
    bias = torch.randn(h, n, n, generator=g)
    math_attn = AT.Attend(causal=True, flash=False)
    flash_attn = AT.Attend(causal=True, flash=True)
    out = dict(
        math_plain=math_attn(q, k, v),
        math_mask=math_attn(q, k, v, mask=mask),
        math_mask_bias=math_attn(q, k, v, mask=mask, attn_bias=bias),
        flash_mask=flash_attn(q, k, v, mask=mask),
        flash_plain=flash_attn(q, k, v),
    )
    return dict(name='attend', kind='attend', inputs=dict(q=q, k=k, v=v, mask=mask, bias=bias), outputs=out)


def soundstream_case():
    torch.manual_seed(0)
    ctor = dict(codebook_size=32, rq_num_quantizers=4, channels=4, codebook_dim=16, use_local_attn=False,
                strides=(2, 4, 5, 8), target_sample_hz=16000)
    ss = S.SoundStream(**ctor)
    full_sd = ss.state_dict()
    keep = {k: v for k, v in full_sd.items() if k.startswith('encoder.') or k.startswith('rq.')}
    shapes = _shapes(keep)
    new = synth_state_dict(shapes, 5)
    full_sd.update(new)
    ss.load_state_dict(full_sd)
    ss.eval()
    g = torch.Generator().manual_seed(11)
    wave = torch.randn(2, 320 * 12 + 77, generator=g) * 0.3
    with torch.no_grad():
        emb, indices, _ = ss(wave, return_encoded=True)
        codes = ss.tokenize(wave)
        x, _ = ss.process_input(wave)
        enc = ss.encoder(x)
    return dict(name='soundstream_small', kind='soundstream', ctor=ctor, shapes=shapes, seed=5, restated=True,
                inputs=dict(wave=wave),
                outputs=dict(indices=indices, tokenize=codes, encoder_out=enc, quantized=emb))


def soundstream_decode_case():
    """SoundStream.decode_from_codebook_indices of the REAL reference decoder (first-party code) around the restated RVQ lookup."""
    torch.manual_seed(0)
    ctor = dict(codebook_size=32, rq_num_quantizers=4, channels=4, codebook_dim=16, use_local_attn=False,
                strides=(2, 4, 5, 8), target_sample_hz=16000)
    ss = S.SoundStream(**ctor)
    full_sd = ss.state_dict()
    keep = {k: v for k, v in full_sd.items() if k.startswith('decoder.') or k.startswith('rq.')}
    shapes = _shapes(keep)
    new = synth_state_dict(shapes, 6)
    full_sd.update(new)
    ss.load_state_dict(full_sd)
    ss.eval()
    g = torch.Generator().manual_seed(12)
    indices = torch.randint(0, 32, (2, 9, 4), generator=g)
    indices[1, -2:, 2:] = -1                                  # dropped quantizers (quantize-dropout / variable-length padding)
    with torch.no_grad():
        wave = ss.decode_from_codebook_indices(indices)
    return dict(name='soundstream_decode_small', kind='soundstream_decode', ctor=ctor, shapes=shapes, seed=6, restated=True,
                inputs=dict(indices=indices), outputs=dict(wave=wave))


def soundstream_local_attn_case():
    """The REAL reference SoundStream in its DEFAULT form (use_local_attn=True, soundstream.py:545, 613, 830-833, 705-706) around the RESTATED
    local-attention modules (oracle/local_attention_restated.py; third-party source not vendored -> restated=True): tokenize + decode.
    150 frames at window 64: two full windows and a ragged third one (autopad); every attention / feed-forward parameter non-trivial."""
    torch.manual_seed(0)
    ctor = dict(codebook_size=32, rq_num_quantizers=4, channels=4, codebook_dim=16, strides=(2, 4, 5, 8), target_sample_hz=16000,
                attn_window_size=64, attn_dim_head=32, attn_heads=2, attn_depth=2)
    ss = S.SoundStream(**ctor)
    full_sd = ss.state_dict()
    keep = {k: v for k, v in full_sd.items() if k.split('.')[0] in ('encoder', 'decoder', 'rq', 'encoder_attn', 'decoder_attn')}
    shapes = _shapes(keep)
    new = synth_state_dict(shapes, 8)
    for k in list(new):
        if k.endswith('rel_pos.inv_freq'):
            new[k] = full_sd[k].clone()                        # the rotary frequencies are a constant buffer, not a parameter
    full_sd.update(new)
    ss.load_state_dict(full_sd)
    ss.eval()
    g = torch.Generator().manual_seed(13)
    wave = torch.randn(2, 320 * 150 + 31, generator=g) * 0.3
    with torch.no_grad():
        emb, indices, _ = ss(wave, return_encoded=True)
        codes = ss.tokenize(wave)
        x, _ = ss.process_input(wave)
        enc = ss.encoder(x).transpose(1, 2)
        enc_attn = ss.encoder_attn(enc)
        mha = ss.encoder_attn.layers[0][0](enc)               # LocalMHA alone (no residual)
        recon = ss.decode_from_codebook_indices(indices)
    return dict(name='soundstream_local_attn_small', kind='soundstream_local_attn', ctor=ctor, shapes=shapes, seed=8, restated=True,
                const_keys=[k for k in shapes if k.endswith('rel_pos.inv_freq')],
                inputs=dict(wave=wave),
                outputs=dict(indices=indices, tokenize=codes, encoder_out=enc, encoder_attn_out=enc_attn, mha0_out=mha, quantized=emb, recon=recon))


def signatures_case():
    """Constructor / forward / generate parameter lists (name, kind, default) of the REAL reference's boundary classes:
    tests/test_host_logic.py holds this package's mirror against them."""
    import inspect
    import json
    from audiolm_pytorch.optimizer import get_optimizer

    def sig(fn):
        out = []
        if fn.__name__ == 'inner' and fn.__closure__:                  # @eval_decorator (audiolm_pytorch.py:140-147) hides the signature
            fn = [c.cell_contents for c in fn.__closure__ if callable(c.cell_contents)][0]
        for name, prm in inspect.signature(fn).parameters.items():
            d = prm.default
            if d is inspect.Parameter.empty:
                d = '<required>'
            elif not isinstance(d, (int, float, bool, str, type(None), tuple)):
                d = '<object>'
            out.append([name, prm.kind.name, list(d) if isinstance(d, tuple) else d])
        return out
    table = {}
    for cls in (A.SemanticTransformer, A.CoarseTransformer, A.FineTransformer, A.Transformer):
        table[cls.__name__ + '.__init__'] = sig(cls.__init__)
        table[cls.__name__ + '.forward'] = sig(cls.forward)
        if hasattr(cls, 'forward_with_cond_scale'):
            table[cls.__name__ + '.forward_with_cond_scale'] = sig(cls.forward_with_cond_scale)
    for cls in (A.SemanticTransformerWrapper, A.CoarseTransformerWrapper, A.FineTransformerWrapper):
        table[cls.__name__ + '.__init__'] = sig(cls.__init__)
        table[cls.__name__ + '.forward'] = sig(cls.forward)
        table[cls.__name__ + '.generate'] = sig(cls.generate)
    table['AudioLM.__init__'] = sig(A.AudioLM.__init__)
    table['AudioLM.forward'] = sig(A.AudioLM.forward)
    table['Attend.__init__'] = sig(AT.Attend.__init__)
    table['Attend.forward'] = sig(AT.Attend.forward)
    table['SoundStream.__init__'] = sig(S.SoundStream.__init__)
    table['SoundStream.forward'] = sig(S.SoundStream.forward)
    table['SoundStream.tokenize'] = sig(S.SoundStream.tokenize)
    table['SoundStream.decode_from_codebook_indices'] = sig(S.SoundStream.decode_from_codebook_indices)
    table['get_optimizer'] = sig(get_optimizer)
    with open(os.path.join(HERE, 'signatures.json'), 'w') as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    print('signatures.json:', len(table), 'callables')


def cond_cases(R):
    """text / audio conditioning from pre-computed embeddings (has_condition=True): cross-attention layers with a null key / value and value
    residual (:450, :539-544), per-sample condition dropping (:888-892), classifier-free guidance (:818-855), cond_as_self_attn_prefix
    (:330-345, :510-515).  cond_dim 24 / 40 -> proj_text_embed is a Linear(cond_dim, 64)."""
    cc = dict(dim=64, depth=2, num_semantic_tokens=6, codebook_size=16, num_coarse_quantizers=3)
    fc = dict(dim=64, depth=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5)
    cases = []
    cases.append(coarse_case('coarse_s4_cond_cross', ctor=dict(cc, flash_attn=True, has_condition=True, cond_dim=24, cond_drop_prob=0.),
                             sem=R(6, (3, 9), 21), coarse=R(16, (3, 5, 3), 22), unique_consecutive=False, mask_prob=0.15, seed=21,
                             cond=dict(dim=24, m=5, zero_rows=((1, 4), (2, 3), (2, 4))), cfg_scale=3.))
    cases.append(coarse_case('coarse_s1_cond_cross_drop_bias', ctor=dict(cc, num_residual_streams=1, has_condition=True, cond_dim=24),
                             sem=R(6, (4, 8), 23), coarse=R(16, (4, 4, 3), 24), unique_consecutive=False, mask_prob=0., seed=23,
                             cond=dict(dim=24, m=6, drop=True, zero_rows=((0, 5),))))
    cases.append(semantic_case('semantic_s4_cond_prefix_bias', ctor=dict(dim=64, depth=2, num_semantic_tokens=20, has_condition=True, cond_dim=40,
                                                                         cond_as_self_attn_prefix=True, cond_drop_prob=0.),
                               ids=R(20, (3, 15), 25), unique_consecutive=False, mask_prob=0.15, seed=25, cond=dict(dim=40, m=4)))
    cases.append(fine_case('fine_s4_cond_prefix_flash_drop', ctor=dict(fc, flash_attn=True, has_condition=True, cond_dim=40, cond_as_self_attn_prefix=True),
                           coarse=R(16, (3, 4, 3), 26), fine=R(16, (3, 4, 5), 27), mask_prob=0.15, seed=26,
                           cond=dict(dim=40, m=5, drop=True, zero_rows=((0, 4), (1, 3)))))
    cases.append(fine_case('fine_s1_cond_cross', ctor=dict(fc, num_residual_streams=1, flash_attn=True, has_condition=True, cond_dim=24, cond_drop_prob=0.),
                           coarse=R(16, (2, 5, 3), 28), fine=R(16, (2, 5, 5), 29), seed=28, cond=dict(dim=24, m=3)))
    return cases


def cache_protocol_case(R):
    """The reference's kv_cache / embed_cache TENSOR protocol (audiolm_pytorch.py:360-370, :487-496, :560, :719, :938-953, :1300-1315) run on
    the REAL reference, eval mode: a prefix call that returns the caches, then one-token steps that consume them; and the guided form
    (forward_with_cond_scale, stacked [cond, null] caches).  Recorded: every call's logits, the cache shapes, the final caches."""
    out = dict(name='cache_protocol', kind='cache_protocol', models={})

    def build(K, ctor, seed):
        torch.manual_seed(0)
        m = K(**ctor)
        shapes = _shapes(m.state_dict())
        m.load_state_dict(synth_state_dict(shapes, seed))
        m.eval()
        return m, shapes

    with torch.no_grad():
        # SemanticTransformer, 1 residual stream (100 % reference code), relative position bias
        ctor = dict(dim=64, depth=2, num_semantic_tokens=20, num_residual_streams=1)
        m, shapes = build(A.SemanticTransformer, ctor, 41)
        ids = R(20, (2, 9), 41)
        calls, kv = [], None
        for n in (6, 7, 8, 9):
            lg, kv = m(ids=ids[:, :n], kv_cache=kv, return_kv_cache=True)
            calls.append(dict(n=n, logits=lg.clone(), kv_shape=tuple(kv.shape)))
        out['models']['semantic'] = dict(ctor=ctor, shapes=shapes, seed=41, ids=ids, calls=calls, kv=kv.clone())

        # CoarseTransformer, 4 streams (restated hyper-connections around the reference's own cache code), bias + cross_attn_bias
        ctor = dict(dim=64, depth=2, num_semantic_tokens=6, codebook_size=16, num_coarse_quantizers=3)
        m, shapes = build(A.CoarseTransformer, ctor, 42)
        sem, coarse = R(6, (2, 7), 42), R(16, (2, 8), 43)
        calls, kv, em = [], None, None
        for n in (4, 5, 6, 7):
            (sl, cl), (kv, em) = m(semantic_token_ids=sem, coarse_token_ids=coarse[:, :n], kv_cache=kv, embed_cache=em, return_cache=True)
            calls.append(dict(n=n, semantic_logits=sl.clone(), coarse_logits=cl.clone(), kv_shape=tuple(kv.shape), embed_shape=tuple(em.shape)))
        out['models']['coarse'] = dict(ctor=ctor, shapes=shapes, seed=42, sem=sem, coarse=coarse, calls=calls, kv=kv.clone(), embed=em.clone())

        # FineTransformer, 1 stream, flash (no bias), conditioned (cross-attention) and GUIDED: stacked [cond, null] caches
        ctor = dict(dim=64, depth=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5, num_residual_streams=1, flash_attn=True,
                    has_condition=True, cond_dim=24)
        m, shapes = build(A.FineTransformer, ctor, 44)
        coarse, fine = R(16, (2, 6), 44), R(16, (2, 9), 45)
        te = torch.randn(2, 4, 24, generator=torch.Generator().manual_seed(46))
        calls, kv, em = [], None, None
        for n in (5, 6, 7):
            (cl, fl), (kv, em) = m.forward_with_cond_scale(coarse, fine[:, :n], text_embeds=te, cond_scale=3., kv_cache=kv, embed_cache=em,
                                                           return_kv_cache=True)
            calls.append(dict(n=n, coarse_logits=cl.clone(), fine_logits=fl.clone(), kv_shape=tuple(kv.shape), embed_shape=tuple(em.shape)))
        out['models']['fine_guided'] = dict(ctor=ctor, shapes=shapes, seed=44, coarse=coarse, fine=fine, text_embeds=te, cond_scale=3., calls=calls,
                                            kv=kv.clone(), embed=em.clone())
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'signatures':
        signatures_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'soundstream_local_attn':  # default-constructor SoundStream (local attention), added in round 2
        c = soundstream_local_attn_case()
        torch.save(c, os.path.join(HERE, c['name'] + '.pt'))
        print(c['name'], {k: tuple(v.shape) for k, v in c['outputs'].items()})
        print(sorted(k for k in c['shapes'] if 'attn' in k)[:14])
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'soundstream_decode':      # add this fixture without touching the others
        c = soundstream_decode_case()
        torch.save(c, os.path.join(HERE, c['name'] + '.pt'))
        print(c['name'], tuple(c['outputs']['wave'].shape))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'fullsize':                # benchmark-size digests of the REAL reference (added in round 3)
        for name, kind, streams in (('full_coarse_s1', 'coarse', 1), ('full_coarse_s4', 'coarse', 4), ('full_fine_s4', 'fine', 4)):
            c = fullsize_case(name, kind, streams)
            path = os.path.join(HERE, c['name'] + '.pt')
            torch.save(c, path)
            nz = c['noise']
            print(f'{name}: loss={float(c["outputs"]["loss"]):.6f} ref fwd+bwd {c["outputs"]["ref_seconds"]:.1f} s; reference bf16-autocast: loss |d| {nz["loss_abs"]:.2e}, '
                  f'logits {["%.2e" % v for v in nz["logits"]]}, worst grads {sorted(((round(v, 3), k) for k, v in nz["grads"].items()), reverse=True)[:4]}; '
                  f'{os.path.getsize(path) / 1024:.0f} KiB')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg0':                    # BASELINE configs[0] alone (its synthetic weights are width-scaled since round 3)
        R0 = lambda hi, shape, seed: torch.randint(0, hi, shape, generator=torch.Generator().manual_seed(seed))
        c = semantic_case('semantic_cfg0', ctor=dict(dim=256, depth=2, num_semantic_tokens=500), ids=R0(500, (8, 255), 0), unique_consecutive=False,
                          mask_prob=0., full=False)
        torch.save(c, os.path.join(HERE, c['name'] + '.pt'))
        print(c['name'], float(c['outputs']['loss']))
        return
    R = lambda hi, shape, seed: torch.randint(0, hi, shape, generator=torch.Generator().manual_seed(seed))
    cases = []
    if len(sys.argv) > 1 and sys.argv[1] == 'cache':                   # the kv / embed cache protocol fixture only (added in round 2)
        c = cache_protocol_case(R)
        path = os.path.join(HERE, c['name'] + '.pt')
        torch.save(c, path)
        for k, v in c['models'].items():
            print(k, [cc.get('kv_shape') for cc in v['calls']], [cc.get('embed_shape') for cc in v['calls']])
        print(f'{c["name"]} {os.path.getsize(path) / 1024:.1f} KiB')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'cond':                    # the conditioning fixtures only (added in round 2)
        cases = cond_cases(R)
        for c in cases:
            path = os.path.join(HERE, c['name'] + '.pt')
            torch.save(c, path)
            print(f'{c["name"]:28s} {os.path.getsize(path) / 1024:8.1f} KiB', 'loss=%s' % float(c['outputs']['loss']))
        return

    # BASELINE.json configs[0]: SemanticTransformer dim=256 depth=2 seq=256 on CPU
    cases.append(semantic_case('semantic_cfg0', ctor=dict(dim=256, depth=2, num_semantic_tokens=500),
                               ids=R(500, (8, 255), 0), unique_consecutive=False, mask_prob=0., full=False))
    ids = R(6, (4, 24), 1)     # tiny vocab -> consecutive repeats -> ragged after unique_consecutive
    cases.append(semantic_case('semantic_s1_uc_mask', ctor=dict(dim=64, depth=2, num_semantic_tokens=6, num_residual_streams=1),
                               ids=ids, unique_consecutive=True, mask_prob=0.15))
    cases.append(semantic_case('semantic_s4_flash', ctor=dict(dim=64, depth=2, num_semantic_tokens=20, flash_attn=True),
                               ids=R(20, (3, 17), 2), unique_consecutive=False, mask_prob=0.15))

    cc = dict(dim=64, depth=2, num_semantic_tokens=6, codebook_size=16, num_coarse_quantizers=3)
    cases.append(coarse_case('coarse_s1_flash_uc_mask', ctor=dict(cc, num_residual_streams=1, flash_attn=True),
                             sem=R(6, (3, 12), 3), coarse=R(16, (3, 7, 3), 4), unique_consecutive=True, mask_prob=0.15))
    cases.append(coarse_case('coarse_s4_bias', ctor=dict(cc), sem=R(6, (3, 12), 5), coarse=R(16, (3, 7, 3), 6),
                             unique_consecutive=False, mask_prob=0.))
    cases.append(coarse_case('coarse_s4_flash_mask', ctor=dict(cc, flash_attn=True), sem=R(6, (2, 9), 7), coarse=R(16, (2, 5, 3), 8),
                             unique_consecutive=False, mask_prob=0.15))
    cases.append(coarse_case('coarse_s1_bias_eval', ctor=dict(cc, num_residual_streams=1), sem=R(6, (2, 9), 9),
                             coarse=R(16, (2, 5, 3), 10), training=False, unique_consecutive=False))

    fc = dict(dim=64, depth=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5)
    coarse = R(16, (3, 6, 3), 11)
    coarse[0, -1, -1] = -1      # pad -> excluded from attention keys and from the CE
    cases.append(fine_case('fine_s1_bias_mask', ctor=dict(fc, num_residual_streams=1), coarse=coarse, fine=R(16, (3, 6, 5), 12),
                           mask_prob=0.15))
    cases.append(fine_case('fine_s4_flash', ctor=dict(fc, flash_attn=True), coarse=R(16, (2, 4, 3), 13), fine=R(16, (2, 4, 5), 14)))

    cases.append(attend_case())
    cases.append(soundstream_case())
    cases.append(soundstream_decode_case())

    for c in cases:
        path = os.path.join(HERE, c['name'] + '.pt')
        torch.save(c, path)
        print(f'{c["name"]:28s} {os.path.getsize(path) / 1024:8.1f} KiB', 'loss=%s' % (float(c['outputs']['loss']) if 'loss' in c['outputs'] else '-'))


if __name__ == '__main__':
    main()
