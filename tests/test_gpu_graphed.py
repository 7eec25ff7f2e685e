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
