"""Fused optimiser step (csrc/optim.hip, through the C ABI) vs torch.optim.Adam / AdamW + torch.nn.utils.clip_grad_norm_ on a real MI355X:
same parameters / gradients for several steps.  fp32 elementwise maths in a different association order: parameters agree to 2e-6
relative to their magnitude after 5 steps, the reported gradient norm to 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1024, 1024), (2730,), (3, 1025, 64), (17,), (1,), (128, 1024), (5, 7, 9)]
    return [torch.randn(*s, generator=g).cuda() for s in shapes]


@pytest.mark.parametrize('wd,max_norm', [(0., None), (0., 0.5), (1e-2, 0.5), (1e-2, None), (0., 1e9)])
def test_fused_adam_matches_torch(wd, max_norm):
    import audiolm_pytorch_amd as A
    ours = [torch.nn.Parameter(p.clone()) for p in _params(0)]
    ref = [torch.nn.Parameter(p.clone()) for p in _params(0)]
    o = A.get_optimizer(ours, lr=1e-3, wd=wd)
    if wd > 0:
        r = torch.optim.AdamW([{'params': [p for p in ref if p.ndim >= 2]}, {'params': [p for p in ref if p.ndim < 2], 'weight_decay': 0}],
                              lr=1e-3, weight_decay=wd, betas=(0.9, 0.99), eps=1e-8)
    else:
        r = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    for step in range(5):
        grads = _params(100 + step)
        for p, q, gr in zip(ours, ref, grads):
            p.grad = (gr * (3.0 if step % 2 else 0.01)).clone()
            q.grad = p.grad.clone()
        if max_norm is not None:
            n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            n_ours = o.clip_grad_norm_(max_norm)
            assert abs(float(n_ours) - float(n_ref)) <= 1e-5 * float(n_ref)
        v0 = [p._version for p in ours]
        o.step()
        r.step()
        assert all(p._version > v for p, v in zip(ours, v0)), 'parameter version counters must advance (bf16 weight caches key on them)'
        for p, q in zip(ours, ref):
            err = float((p.detach() - q.detach()).abs().max() / q.detach().abs().max().clamp(min=1e-30))
            assert err <= 2e-6, (step, tuple(p.shape), err)
    so, sr = o.state_dict()['state'], r.state_dict()['state']
    assert set(so[0].keys()) == set(sr[0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    assert float(so[0]['step']) == float(sr[0]['step']) == 5.0


def test_fused_adam_skips_parameters_without_gradient():
    import audiolm_pytorch_amd as A
    ps = [torch.nn.Parameter(p.clone()) for p in _params(1)]
    o = A.get_optimizer(ps, lr=1e-2, wd=0.)
    ps[0].grad = torch.ones_like(ps[0])
    before = [p.detach().clone() for p in ps]
    o.step()
    assert not torch.equal(ps[0].detach(), before[0])
    assert all(torch.equal(p.detach(), b) for p, b in zip(ps[1:], before[1:]))


def test_fused_adam_keeps_a_step_count_per_parameter():
    """torch.optim.Adam tracks `step` per parameter: a parameter whose gradient is None on some steps (a conditional head, a late-unfrozen
    weight) gets its own bias corrections.  Three steps where the second parameter only has a gradient on steps 2 and 3."""
    import audiolm_pytorch_amd as A
    base = _params(3)[:3]
    ours = [torch.nn.Parameter(p.clone()) for p in base]
    ref = [torch.nn.Parameter(p.clone()) for p in base]
    o = A.get_optimizer(ours, lr=3e-3, wd=0.)
    r = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        for i, (p, q) in enumerate(zip(ours, ref)):
            if i == 1 and step == 0:
                p.grad = q.grad = None
                continue
            gr = torch.randn(p.shape, generator=g).to(p.device)
            p.grad, q.grad = gr.clone(), gr.clone()
        o.step()
        r.step()
        for p, q in zip(ours, ref):
            err = float((p.detach() - q.detach()).abs().max() / q.detach().abs().max().clamp(min=1e-30))
            assert err <= 2e-6, (step, tuple(p.shape), err)
    assert [float(o.state[p]['step']) for p in ours] == [float(r.state[q]['step']) for q in ref] == [3.0, 2.0, 3.0]


def _train_model(seed=0):
    import audiolm_pytorch_amd as A
    torch.manual_seed(seed)
    model = A.CoarseTransformer(dim=256, depth=2, num_semantic_tokens=100, codebook_size=64, num_coarse_quantizers=3, flash_attn=True).cuda()

    class Codec:
        rq_groups = 1
        num_quantizers = 8
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.)
    w.train()
    g = torch.Generator().manual_seed(7)
    data = dict(semantic_token_ids=torch.randint(0, 100, (2, 40), generator=g).cuda(), coarse_token_ids=torch.randint(0, 64, (2, 30, 3), generator=g).cuda())
    return model, w, data


@pytest.mark.parametrize('wd', [0., 1e-2])
def test_fused_adam_writes_the_packed_weight_images(monkeypatch, wd):
    """Round 4 (alm_opt_adam_pack_step): the optimiser step of a dense GEMM weight also writes its packed bf16 images, so the forward after a step
    does not re-pack.  Model A trains (forward, backward, fused step); model B -- same initial weights, never run forward, hence no packed images and
    the plain alm_opt_adam_step for every tensor -- receives A's gradients and takes the same steps: parameters bit-identical after every step (the two
    kernels share the arithmetic).  A packs its stack weights in the first forward only; the images the optimiser left in A's cache are bit-identical to
    a fresh pack of the updated masters; with ALM_FUSED_ADAM_PACK off every forward re-packs.  (Losses of two separate training runs are NOT compared:
    the embedding-gradient atomics make gradients differ in the last bit between runs, and Adam's first steps turn that into sign flips.)"""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import core, ops
    real = ops.pack_weights_multi

    def counting(packs):
        # the stack's weights only (jobs with a row-major destination; the logit heads pack transposed images of their own and are not fused)
        return lambda jobs: (packs.append(sum(1 for j in jobs if j[1] is not None)), real(jobs))[1]

    monkeypatch.setattr(core, 'FUSED_ADAM_PACK', True)
    model_a, w_a, data = _train_model()
    model_b, _, _ = _train_model()
    model_b.load_state_dict(model_a.state_dict())                 # (the hyper-connection init draws from Python's `random`: two builds differ)
    opt_a = A.get_optimizer(model_a.parameters(), lr=1e-3, wd=wd)
    opt_b = A.get_optimizer(model_b.parameters(), lr=1e-3, wd=wd)
    packs, per_step = [], []
    monkeypatch.setattr(ops, 'pack_weights_multi', counting(packs))
    for step in range(3):
        n0 = sum(packs)
        loss = w_a(**data, return_loss=True)
        per_step.append(sum(packs) - n0)
        loss.backward()
        if step == 0:
            # one dense weight sits the first update out (a late-unfrozen layer): its step count then differs from its group's, the group's table entries
            # carry per-tensor counts -- the per-tensor bias-correction branch of alm_opt_adam_pack_step
            late = next(p for k, p in model_a.named_parameters() if 'to_q' in k and p.dim() == 2)
            late.grad = None
        for pa, pb in zip(model_a.parameters(), model_b.parameters()):
            pb.grad = None if pa.grad is None else pa.grad.detach().clone()
        na = opt_a.clip_grad_norm_(0.5)
        nb = opt_b.clip_grad_norm_(0.5)
        assert float(na) == float(nb)
        opt_a.step()
        opt_b.step()
        for (k, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
            assert torch.equal(pa, pb), (step, k)
        opt_a.zero_grad(set_to_none=True)
    assert per_step[0] > 0 and per_step[1:] == [0, 0], per_step            # packed once, kept current by the optimiser
    steps = {float(opt_a.state[p]['step']) for p in model_a.parameters() if p in opt_a.state}
    assert steps == {2.0, 3.0}, steps                                      # (the late weight: two updates)
    # the images the fused step left in the cache == a fresh pack of the final masters
    monkeypatch.setattr(ops, 'pack_weights_multi', real)
    checked = 0
    for key, entry in model_a.transformer._cache.store.items():
        if not (isinstance(key, tuple) and len(key) == 3 and key[2] in ('wq', 'wkv', 'wo', 'w1', 'w2')):
            continue
        stamp, (W, WT) = entry
        master = next(p for p in model_a.parameters() if p.data_ptr() == stamp[0])
        assert stamp == (master.data_ptr(), master._version, tuple(master.shape)), key
        if key[2] == 'w1':
            ref_W, ref_WT = core._pack_w1(master.detach(), *_inner(model_a))
        elif key[2] == 'w2':
            ref_W, ref_WT = core._pack_w2(master.detach(), *_inner(model_a))
        else:
            ref_W, ref_WT = core._pack_plain(master.detach())
        assert torch.equal(W, ref_W) and torch.equal(WT, ref_WT), key
        checked += 1
    assert checked >= 10, checked
    # the switch: off -> every forward after a step re-packs
    monkeypatch.setattr(core, 'FUSED_ADAM_PACK', False)
    model_c, w_c, _ = _train_model()
    opt_c = A.get_optimizer(model_c.parameters(), lr=1e-3, wd=wd)
    packs_c, per_step_c = [], []
    monkeypatch.setattr(ops, 'pack_weights_multi', counting(packs_c))
    for step in range(2):
        n0 = sum(packs_c)
        w_c(**data, return_loss=True).backward()
        per_step_c.append(sum(packs_c) - n0)
        opt_c.step()
        opt_c.zero_grad(set_to_none=True)
    assert all(n > 0 for n in per_step_c), per_step_c


def _inner(model):
    cfg = model.transformer.cfg
    return cfg.inner, cfg.inner_pad
