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
