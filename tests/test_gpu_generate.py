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


@pytest.mark.parametrize('flash', [True, False])
def test_semantic_generate_greedy_matches_oracle(greedy, flash):
    import audiolm_pytorch_amd as A
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    model = A.SemanticTransformer(dim=64, depth=2, heads=2, num_semantic_tokens=20, flash_attn=flash)
    sd = _sd(model)
    model.to(dev)
    w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=False)
    g = torch.Generator().manual_seed(1)
    prime = torch.randint(0, 20, (3, 4), generator=g)
    out = w.generate(max_length=12, prime_ids=prime.to(dev))
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


@pytest.mark.parametrize('flash', [True, False])
def test_coarse_generate_greedy_matches_oracle(greedy, flash):
    import audiolm_pytorch_amd as A
    torch.manual_seed(2)
    dev = torch.device('cuda:0')
    model = A.CoarseTransformer(dim=64, depth=2, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3, flash_attn=flash)
    sd = _sd(model)
    model.to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False)
    g = torch.Generator().manual_seed(3)
    sem = torch.randint(0, 20, (2, 6), generator=g)
    out = w.generate(semantic_token_ids=sem.to(dev), max_time_steps=3).cpu()
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


@pytest.mark.parametrize('flash', [True, False])
def test_fine_generate_greedy_matches_oracle(greedy, flash):
    import audiolm_pytorch_amd as A
    torch.manual_seed(4)
    dev = torch.device('cuda:0')
    model = A.FineTransformer(dim=64, depth=2, heads=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5, flash_attn=flash)
    sd = _sd(model)
    model.to(dev)
    w = A.FineTransformerWrapper(transformer=model, codec=Codec())
    g = torch.Generator().manual_seed(5)
    coarse = torch.randint(0, 16, (2, 2, 3), generator=g)
    out = w.generate(coarse_token_ids=coarse.to(dev)).cpu()
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
    with pytest.raises(NotImplementedError):
        w.generate(semantic_token_ids=sem, max_time_steps=1, reconstruct_wave=True)           # needs the SoundStream decoder
    with pytest.raises(NotImplementedError):
        w.generate(semantic_token_ids=sem, max_time_steps=1, text=['a'])                        # conditioning
    with pytest.raises(NotImplementedError):
        A.AudioLM()
