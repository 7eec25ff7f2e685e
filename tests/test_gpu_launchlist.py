"""MI355X: the recorded launch lists of the transformer stack (audiolm-pytorch_amd/launchlist.py, C ABI alm_list_run) re-issue EXACTLY the eager step.

A list is the Python path's own launch sequence written down, so the bar is bitwise: loss, logits and every parameter gradient of a replayed step are
`torch.equal` to the step issued launch by launch from Python (ALM_LAUNCH_LIST off) -- on the same inputs, after the weights changed (the packed weight
images are bases of the list, not baked addresses), with and without a key mask, for all three transformers (structured attention bias included), in
evaluation mode, and when two shapes alternate.  The reference behaviour these steps implement: audiolm_pytorch.py:528-547 (depth loop) and its autograd."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class Codec:
    rq_groups = 1

    def __init__(self, nq=8):
        self.num_quantizers = nq


def _build(kind, dev, flash):
    import audiolm_pytorch_amd as A
    torch.manual_seed(0)
    if kind == 'coarse':
        m = A.CoarseTransformer(dim=256, depth=3, heads=4, num_semantic_tokens=50, codebook_size=64, num_coarse_quantizers=3, flash_attn=flash).to(dev)
        w = A.CoarseTransformerWrapper(transformer=m, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
        g = torch.Generator().manual_seed(3)
        mk = lambda B, n: dict(semantic_token_ids=torch.randint(0, 50, (B, n), generator=g).to(dev), coarse_token_ids=torch.randint(0, 64, (B, n, 3), generator=g).to(dev))
    elif kind == 'fine':
        m = A.FineTransformer(dim=256, depth=2, heads=4, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=64, flash_attn=flash).to(dev)
        w = A.FineTransformerWrapper(transformer=m, codec=Codec(8), mask_prob=0.)
        g = torch.Generator().manual_seed(4)
        mk = lambda B, n: dict(coarse_token_ids=torch.randint(0, 64, (B, n, 3), generator=g).to(dev), fine_token_ids=torch.randint(0, 64, (B, n, 5), generator=g).to(dev))
    else:
        m = A.SemanticTransformer(dim=256, depth=2, heads=4, num_semantic_tokens=50, flash_attn=flash).to(dev)
        w = A.SemanticTransformerWrapper(transformer=m, unique_consecutive=False, mask_prob=0.)
        g = torch.Generator().manual_seed(5)
        mk = lambda B, n: dict(semantic_token_ids=torch.randint(0, 50, (B, n), generator=g).to(dev))
    w.train()
    return m, w, mk


def _step(model, w, kw, amp=True):
    torch.manual_seed(1)                                       # the forgetful mask of every step alike
    for p in model.parameters():
        p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        loss = w(**kw, return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _same(a, b):
    la, ga = a
    lb, gb = b
    assert torch.equal(la, lb), (float(la), float(lb))
    assert ga.keys() == gb.keys() and len(ga) > 20
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not bad, bad[:8]


@pytest.mark.parametrize('kind,flash,amp', [('coarse', True, True), ('coarse', True, False), ('coarse', False, True), ('fine', False, True), ('semantic', False, True),
                                            ('semantic', True, False)])
def test_replayed_step_is_bitwise_the_eager_step(kind, flash, amp, monkeypatch):
    from audiolm_pytorch_amd import launchlist as LL
    dev = torch.device('cuda:0')
    model, w, mk = _build(kind, dev, flash)
    kw = mk(2, 96)
    monkeypatch.setattr(LL, 'ENABLED', False)
    eager = _step(model, w, kw, amp)
    monkeypatch.setattr(LL, 'ENABLED', True)
    LL.PLANS.clear()
    before = dict(LL.STATS)
    sized = _step(model, w, kw, amp)                           # SIZE pass (an eager step with a byte tally)
    recorded = _step(model, w, kw, amp)                        # RECORD pass (an eager step out of one arena, written down)
    assert LL.STATS['refused'] == before['refused'], [p.why for p in LL.PLANS.values()]
    assert LL.STATS['recorded'] - before['recorded'] == 2      # one forward and one backward list
    r1 = _step(model, w, kw, amp)
    r2 = _step(model, w, kw, amp)
    assert LL.STATS['replayed'] - before['replayed'] == 2 and all(p.state == 'ready' for p in LL.PLANS.values())
    for other in (sized, recorded, r1, r2):
        _same(eager, other)
    # the weights move (an optimiser step): the lists follow the re-packed bf16 images
    with torch.no_grad():
        for i, p in enumerate(model.parameters()):
            p.mul_(1.0 + 1e-3 * ((i % 7) - 3))
    r3 = _step(model, w, kw, amp)
    monkeypatch.setattr(LL, 'ENABLED', False)
    e3 = _step(model, w, kw, amp)
    _same(e3, r3)
    assert not torch.equal(e3[0], eager[0])


def test_two_shapes_alternate_and_an_evaluation_forward_replays():
    from audiolm_pytorch_amd import launchlist as LL
    dev = torch.device('cuda:0')
    model, w, mk = _build('coarse', dev, True)
    ka, kb = mk(2, 64), mk(3, 80)
    LL.PLANS.clear()
    ref = {}
    for name, kw in (('a', ka), ('b', kb)):
        ref[name] = _step(model, w, kw)
    for _ in range(3):
        for name, kw in (('a', ka), ('b', kb)):
            _same(ref[name], _step(model, w, kw))
    assert sum(p.state == 'ready' for p in LL.PLANS.values()) == 2
    # no-grad forward: its own key (no saved activations), sized / recorded / replayed -- equal logits every time
    outs = []
    w.eval()
    n0 = LL.STATS['replayed']
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        for _ in range(4):
            torch.manual_seed(1)
            outs.append(w(**ka, return_loss=True).clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert LL.STATS['replayed'] - n0 == 2


def test_headline_shape_replays_bitwise_and_counts_its_launches():
    """the benchmarked step (CoarseTransformer d = 1024, depth 6, B = 8 x N = 2048, mask_prob 0.15, bf16 streams under autocast): replayed == eager, bit for bit;
    the forward list holds every launch of the depth loop (6 layers x (width / depth connection + to_q, to_kv + attention + to_out; connection + W1 + GEGLU-LN + W2))"""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import launchlist as LL
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = A.CoarseTransformer(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
    w.train()
    g = torch.Generator().manual_seed(5)
    kw = dict(semantic_token_ids=torch.randint(0, 500, (8, 509), generator=g).to(dev), coarse_token_ids=torch.randint(0, 1024, (8, 512, 3), generator=g).to(dev))
    LL.PLANS.clear()
    runs = [_step(model, w, kw) for _ in range(4)]
    plans = list(LL.PLANS.values())
    assert len(plans) == 1 and plans[0].state == 'ready', [(p.state, p.why) for p in plans]
    for r in runs[1:]:
        _same(runs[0], r)
    assert 50 <= plans[0].fwd.n <= 80 and 50 <= plans[0].bwd.n <= 120, (plans[0].fwd.n, plans[0].bwd.n)     # measured: 60 and 70 launches
    assert plans[0].fwd.arena_bytes < 8 << 30 and plans[0].bwd.arena_bytes < 8 << 30
