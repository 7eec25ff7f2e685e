"""The shipped defaults are the benchmarked configuration (round-3 VERDICT item 9): a model built WITHOUT `residual_dtype=` stores its hyper-connection
residual streams the way the caller's precision context implies -- bf16 inside `torch.autocast(bfloat16)` (how reference trainer.py:1241 runs the model, and
what bench.py times), fp32 in a plain fp32 call -- with no environment variable involved."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class Codec:
    rq_groups = 1
    num_quantizers = 8


def _step(model, inputs, autocast):
    import audiolm_pytorch_amd as A
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.)
    w.train()
    for p in model.parameters():
        p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        loss = w(**inputs, return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def test_residual_stream_storage_follows_autocast(monkeypatch):
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import core
    monkeypatch.delenv('ALM_RESIDUAL_DTYPE', raising=False)
    assert core.default_residual_bf16() is None                     # `auto`
    dev = torch.device('cuda:0')
    ctor = dict(dim=256, depth=2, num_semantic_tokens=100, codebook_size=64, num_coarse_quantizers=3, flash_attn=True)
    torch.manual_seed(0)
    auto = A.CoarseTransformer(**ctor).to(dev)
    assert auto.transformer._residual_auto
    sd = {k: v.detach().clone() for k, v in auto.state_dict().items()}
    bf = A.CoarseTransformer(**ctor, residual_dtype=torch.bfloat16).to(dev)
    fp = A.CoarseTransformer(**ctor, residual_dtype=torch.float32).to(dev)
    bf.load_state_dict(sd), fp.load_state_dict(sd)
    assert not bf.transformer._residual_auto and bf.transformer.cfg.residual_bf16 and not fp.transformer.cfg.residual_bf16
    g = torch.Generator().manual_seed(3)
    inputs = dict(semantic_token_ids=torch.randint(0, 100, (2, 50), generator=g).to(dev), coarse_token_ids=torch.randint(0, 64, (2, 40, 3), generator=g).to(dev))
    seen = []
    orig = core.stack_forward
    monkeypatch.setattr(core, 'stack_forward', lambda x, m, flat, cfg, *a, **k: (seen.append(cfg.residual_bf16), orig(x, m, flat, cfg, *a, **k))[1])
    l_auto_ac, g_auto_ac = _step(auto, inputs, True)
    l_auto_32, g_auto_32 = _step(auto, inputs, False)
    assert seen == [True, False], seen                              # bf16 streams under autocast, fp32 outside
    l_bf, g_bf = _step(bf, inputs, False)                           # an explicit dtype ignores the context
    l_fp, g_fp = _step(fp, inputs, True)
    assert seen[2:] == [True, False], seen
    assert l_auto_ac == l_bf and l_auto_32 == l_fp                  # the same launch sequences: identical losses
    frob = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))   # noqa: E731  (the embedding scatter adds atomically)
    for k in g_bf:
        assert frob(g_auto_ac[k], g_bf[k]) <= 1e-6 and frob(g_auto_32[k], g_fp[k]) <= 1e-6, k
    assert abs(l_bf - l_fp) > 0                                     # and the two storages really are different programs
