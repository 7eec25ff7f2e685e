"""Schedules of the training step on a real MI355X: the eager launch sequence, the same step captured into a hipGraph, and the two-half-batch
variant (core.TransformerStackFn opts micro = 2: two HIP streams inside the fused stack) must compute the same loss and gradients -- only the
order of the weight-gradient sums over tokens differs between one batch and two half batches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class Codec:
    rq_groups = 1
    num_quantizers = 8


def _frob(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def _setup(residual_dtype):
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = A.CoarseTransformer(dim=256, depth=3, num_semantic_tokens=100, codebook_size=64, num_coarse_quantizers=3, flash_attn=True,
                                residual_dtype=residual_dtype).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.)
    w.train()
    g = torch.Generator().manual_seed(3)
    inputs = dict(semantic_token_ids=torch.randint(0, 100, (4, 50), generator=g).to(dev), coarse_token_ids=torch.randint(0, 64, (4, 40, 3), generator=g).to(dev))
    return model, w, inputs


def _eager(model, w, inputs):
    for p in model.parameters():
        p.grad = None
    loss = w(**inputs, return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('residual', [torch.float32, torch.bfloat16])
def test_two_half_batches_inside_the_stack_match_one_batch(residual):
    model, w, inputs = _setup(residual)
    l1, g1 = _eager(model, w, inputs)
    model.transformer.micro_batches = 2
    try:
        l2, g2 = _eager(model, w, inputs)
    finally:
        model.transformer.micro_batches = 1
    assert abs(l1 - l2) <= 1e-6 * abs(l1), (l1, l2)            # the forward is per-sequence: identical
    assert g1.keys() == g2.keys()
    bad = [(k, _frob(g2[k], g1[k])) for k in g1 if _frob(g2[k], g1[k]) > 2e-3]
    assert not bad, bad[:6]


@pytest.mark.parametrize('micro', [1, 2])
def test_graphed_step_matches_eager(micro):
    from audiolm_pytorch_amd.graphed import GraphedTrainStep
    model, w, inputs = _setup(torch.bfloat16)
    l1, g1 = _eager(model, w, inputs)
    for p in model.parameters():
        p.grad = None
    step = GraphedTrainStep(w, inputs, micro_batches=micro)
    for _ in range(2):                                          # replays are repeatable
        loss = step(**inputs)
        torch.cuda.synchronize()
        assert abs(float(loss) - l1) <= 1e-6 * abs(l1), (float(loss), l1)
        got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        assert got.keys() == g1.keys()
        bad = [(k, _frob(got[k], g1[k])) for k in g1 if _frob(got[k], g1[k]) > 2e-3]
        assert not bad, bad[:6]
    # new inputs / new weights flow through the replay: compare with a fresh eager step
    g = torch.Generator().manual_seed(9)
    inputs2 = dict(semantic_token_ids=torch.randint(0, 100, (4, 50), generator=g).cuda(), coarse_token_ids=torch.randint(0, 64, (4, 40, 3), generator=g).cuda())
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
    loss = step(**inputs2)
    torch.cuda.synchronize()
    got = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    l3, g3 = _eager(model, w, inputs2)
    assert abs(float(loss) - l3) <= 1e-6 * abs(l3)
    bad = [(k, _frob(got[k], g3[k])) for k in g3 if _frob(got[k], g3[k]) > 2e-3]
    assert not bad, bad[:6]


def test_eager_forward_right_after_capture_uses_real_weight_packs():
    """ADVICE (round 2): the capture only RECORDS the weight re-pack kernels, yet keys the packs into the WeightCache -- an eager forward between
    construction and the first replay would read uninitialised bf16 weights.  GraphedTrainStep now replays once before returning."""
    from audiolm_pytorch_amd.graphed import GraphedTrainStep
    model, w, inputs = _setup(torch.bfloat16)
    with torch.no_grad():
        before = float(w(**inputs, return_loss=True))
    step = GraphedTrainStep(w, inputs, micro_batches=1)
    with torch.no_grad():
        after = float(w(**inputs, return_loss=True))               # hits the cache entries the capture created
    assert abs(after - before) <= 1e-6 * abs(before), (before, after)
    assert step.loss is not None and abs(float(step.loss) - before) <= 1e-6 * abs(before)
    assert all(g is None or bool(torch.isfinite(g).all()) for g in step.grads)


def test_graph_replays_redraw_the_attention_dropout_masks(monkeypatch):
    """ADVICE (round 3, medium): the attention-dropout seed was a host integer passed by value -- baked into the captured launches, so every replay of a
    graphed step dropped the SAME (query, key) pairs.  Inside a capture the flash kernels now read `seed + counter` from a device counter that the
    captured step advances (core.graph_seed_state): (1) two replays on identical inputs give different losses; (2) the counter moves by 2 x depth per
    replay; (3) forward and backward of ONE replay use the same masks: the replay's loss and gradients equal an EAGER step whose host seed is set to the
    counter value that replay used (the to_out dropout masks, drawn by torch's generator, are pinned to all-ones on both sides for this comparison)."""
    from audiolm_pytorch_amd import core
    from audiolm_pytorch_amd.graphed import GraphedTrainStep
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = A.CoarseTransformer(dim=256, depth=3, num_semantic_tokens=100, codebook_size=64, num_coarse_quantizers=3, flash_attn=True, attn_dropout=0.25,
                                residual_dtype=torch.bfloat16).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.)
    w.train()
    g = torch.Generator().manual_seed(3)
    inputs = dict(semantic_token_ids=torch.randint(0, 100, (4, 50), generator=g).to(dev), coarse_token_ids=torch.randint(0, 64, (4, 40, 3), generator=g).to(dev))
    monkeypatch.setattr(core, '_dropout_keep', lambda shape, p, device: torch.ones(shape, dtype=torch.bfloat16, device=device))
    step = GraphedTrainStep(w, inputs, micro_batches=1)
    counter = core.graph_seed_state(dev)
    c0 = int(counter.item())
    la = float(step(**inputs))
    torch.cuda.synchronize()
    c1 = int(counter.item())
    ga = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    lb = float(step(**inputs))
    torch.cuda.synchronize()
    c2 = int(counter.item())
    assert c1 - c0 == 2 * model.transformer.cfg.depth and c2 - c1 == 2 * model.transformer.cfg.depth, (c0, c1, c2)
    assert abs(la - lb) > 1e-5 * abs(la), f'two replays dropped the same pairs: {la} == {lb}'
    monkeypatch.setattr(core, '_attn_seed', lambda: c1)            # the eager path: host seed = the counter value the first replay's kernels read
    le, ge = _eager(model, w, inputs)
    assert abs(le - la) <= 1e-6 * abs(la), (le, la)
    bad = [(k, _frob(ga[k], ge[k])) for k in ge if _frob(ga[k], ge[k]) > 2e-3]
    assert not bad, bad[:6]
    # eval(): no dropout, replays of an eval-mode capture are repeatable
    w.eval()
    with torch.no_grad():
        assert float(w(**inputs, return_loss=True)) == float(w(**inputs, return_loss=True))


def test_two_stack_forwards_in_one_capture_keep_their_own_dropout_masks(monkeypatch):
    """ADVICE (round 4, medium): the mask-stream counter is ONE device int64 that every stack forward advances in place; the backward of the first of two
    forwards inside one captured step (the same transformer called twice: in-capture gradient accumulation) therefore read the counter AFTER the second
    forward had advanced it -- dQ / dK / dV with other keep masks than the forward.  Each call now carries its own snapshot of the counter.  Check: the
    replay's loss and gradients equal an eager step whose two forwards are given, as host seeds, the two counter values the replay's forwards used."""
    from audiolm_pytorch_amd import core
    from audiolm_pytorch_amd.graphed import GraphedTrainStep
    import audiolm_pytorch_amd as A
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = A.CoarseTransformer(dim=256, depth=3, num_semantic_tokens=100, codebook_size=64, num_coarse_quantizers=3, flash_attn=True, attn_dropout=0.25,
                                residual_dtype=torch.bfloat16).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.)
    w.train()

    class Twice(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = w

        def forward(self, **kw):
            return self.w(**kw) + self.w(**kw)

    twice = Twice()
    g = torch.Generator().manual_seed(3)
    inputs = dict(semantic_token_ids=torch.randint(0, 100, (4, 50), generator=g).to(dev), coarse_token_ids=torch.randint(0, 64, (4, 40, 3), generator=g).to(dev))
    monkeypatch.setattr(core, '_dropout_keep', lambda shape, p, device: torch.ones(shape, dtype=torch.bfloat16, device=device))
    step = GraphedTrainStep(twice, inputs, micro_batches=1)
    counter = core.graph_seed_state(dev)
    c0 = int(counter.item())
    la = float(step(**inputs))
    torch.cuda.synchronize()
    depth = model.transformer.cfg.depth
    assert int(counter.item()) - c0 == 4 * depth                    # two forwards per replay
    ga = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    seeds = iter([c0 + 2 * depth, c0 + 4 * depth])                  # what the first / second forward of that replay read
    monkeypatch.setattr(core, '_attn_seed', lambda: next(seeds))
    le, ge = _eager(model, twice, inputs)
    assert abs(le - la) <= 1e-6 * abs(la), (le, la)
    bad = [(k, _frob(ga[k], ge[k])) for k in ge if _frob(ga[k], ge[k]) > 2e-3]
    assert not bad, bad[:6]


@pytest.mark.parametrize('groups', [2, 3])
def test_eager_deferred_weight_gradient_groups_match_one_group_and_the_per_layer_path(monkeypatch, groups):
    """ADVICE (round 3): the eager ALM_DEFER_GROUPS > 1 path (the upper layer groups' batched weight-gradient launches forked to the side stream in the
    middle of the backward pass) had no test.  Gradients of 2 / 3 groups vs one group at the end vs the per-layer split-K path: the same sums over
    tokens in a different order (the split-K plan depends on how many problems a launch holds) -- equal to fp32 summation noise."""
    from audiolm_pytorch_amd import core
    model, w, inputs = _setup(torch.bfloat16)
    monkeypatch.setattr(core, 'DEFER_GROUPS', 1)
    l1, g1 = _eager(model, w, inputs)
    monkeypatch.setattr(core, 'DEFER_GROUPS', groups)
    l2, g2 = _eager(model, w, inputs)
    monkeypatch.setattr(core, 'DEFER_WGRAD', False)
    l3, g3 = _eager(model, w, inputs)
    assert l1 == l2 == l3
    assert g1.keys() == g2.keys() == g3.keys()
    bad = [(k, _frob(g2[k], g1[k]), _frob(g3[k], g1[k])) for k in g1 if _frob(g2[k], g1[k]) > 2e-5 or _frob(g3[k], g1[k]) > 2e-5]
    assert not bad, bad[:6]
