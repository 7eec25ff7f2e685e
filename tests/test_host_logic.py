"""CPU: host-side mirror of the reference interface -- state_dict compatibility, integer bookkeeping (bit-exact vs the
oracle / golden vectors), logit-head regrouping index logic.  No GPU, no HIP compute."""
import glob
import os

import pytest
import torch

import audiolm_oracle as O
import audiolm_pytorch_amd as A
from audiolm_pytorch_amd import audiolm_pytorch as AP
from common import GOLDEN_DIR, synth_state_dict

FIXTURES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.pt'))
                  if os.path.basename(p).split('_')[0] in ('semantic', 'coarse', 'fine'))
KLASS = dict(semantic=A.SemanticTransformer, coarse=A.CoarseTransformer, fine=A.FineTransformer)


class Codec:
    rq_groups = 1

    def __init__(self, nq=8):
        self.num_quantizers = nq


def _load(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + '.pt'), weights_only=False)


@pytest.mark.parametrize('name', FIXTURES)
def test_state_dict_matches_reference(name):
    """Same parameter / buffer names and shapes as the reference modules (checkpoints interchange)."""
    fx = _load(name)
    model = KLASS[fx['kind']](**fx['ctor'])
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ours == {k: tuple(v) for k, v in fx['shapes'].items()}
    model.load_state_dict(synth_state_dict(fx['shapes'], fx['seed']), strict=True)


def _capture(model):
    seen = {}

    def fake(*args, **kwargs):
        seen['args'], seen['kwargs'] = args, kwargs
        z = torch.zeros(())
        return z if isinstance(model, A.SemanticTransformer) else (z, z)
    model.forward = fake
    return seen


def test_coarse_wrapper_bookkeeping_bit_exact():
    fx = _load('coarse_s1_flash_uc_mask')
    model = A.CoarseTransformer(**fx['ctor'])
    seen = _capture(model)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=True, mask_prob=0.15)
    w.train()
    inp = fx['inputs']
    orig = AP.generate_mask_with_prob
    AP.generate_mask_with_prob = lambda shape, prob, device: inp['forgetful_mask'].clone()
    try:
        w(semantic_token_ids=inp['semantic_token_ids'], coarse_token_ids=inp['coarse_token_ids'], return_loss=True)
    finally:
        AP.generate_mask_with_prob = orig
    sem_in, coarse_in, sem_labels, coarse_labels, mask = O.coarse_wrapper_bookkeeping(
        inp['semantic_token_ids'], inp['coarse_token_ids'], model.semantic_eos_id, model.coarse_eos_id, training=True, unique_consecutive=True)
    k = seen['kwargs']
    assert torch.equal(k['semantic_token_ids'], sem_in) and torch.equal(k['coarse_token_ids'], coarse_in)
    assert torch.equal(k['self_attn_mask'], mask & inp['forgetful_mask'])
    assert torch.equal(k['labels'][0], sem_labels) and torch.equal(k['labels'][1], coarse_labels)
    assert k['semantic_token_ids'].dtype == torch.int64


def test_semantic_wrapper_bookkeeping_bit_exact():
    fx = _load('semantic_s1_uc_mask')
    model = A.SemanticTransformer(**fx['ctor'])
    seen = _capture(model)
    w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=True, mask_prob=0.)
    w.train()
    w(semantic_token_ids=fx['inputs']['ids'], return_loss=True)
    inp_ids, labels = O.semantic_wrapper_bookkeeping(fx['inputs']['ids'], model.eos_id, training=True, unique_consecutive=True)
    assert torch.equal(seen['kwargs']['ids'], inp_ids) and torch.equal(seen['kwargs']['labels'], labels)
    w.eval()
    w(semantic_token_ids=fx['inputs']['ids'], return_loss=True)
    inp_ids, labels = O.semantic_wrapper_bookkeeping(fx['inputs']['ids'], model.eos_id, training=False, unique_consecutive=True)
    assert torch.equal(seen['kwargs']['ids'], inp_ids) and torch.equal(seen['kwargs']['labels'], labels)


@pytest.mark.parametrize('B,N,start,n,Q', [(3, 40, 5, 22, 3), (2, 33, 0, 33, 1), (2, 50, 10, 40, 5), (1, 20, 3, 7, 3)])
def test_head_regrouping_equals_reference_einsum(B, N, start, n, Q):
    """'q c d, b n q d -> b n q c' + remainder with W[:r] (audiolm_pytorch.py:965-983)  ==  per-quantizer row gather + matmul."""
    g = torch.Generator().manual_seed(0)
    D, C = 8, 6
    hn = torch.randn(B * N, D, generator=g, dtype=torch.float64)
    W = torch.randn(Q, C, D, generator=g, dtype=torch.float64)
    labels = torch.randint(0, C, (B, n), generator=g)
    idx, ig, va = AP._group_index(B, N, start, n, Q, torch.device('cpu'))
    lab = AP._group_labels(labels, ig, va, n)
    rows = torch.where(idx.reshape(-1)[:, None] >= 0, hn[idx.reshape(-1).clamp(min=0).long()], torch.zeros(1, D, dtype=torch.float64))
    lg = torch.einsum('grd,gcd->grc', rows.view(Q, -1, D), W).reshape(-1, C)
    ours = AP._ungroup_logits(lg, B, n, Q, C)
    ref = O._grouped_logits(W, hn.view(B, N, D)[:, start:start + n], Q)
    assert torch.allclose(ours, ref, atol=1e-12)
    # labels travel with their rows; padded slots are ignored
    flat_lab = lab.reshape(-1)
    valid = idx.reshape(-1) >= 0
    assert bool((flat_lab[~valid] == -1).all())
    b_of = (idx.reshape(-1)[valid].long() // N)
    i_of = (idx.reshape(-1)[valid].long() % N) - start
    assert torch.equal(flat_lab[valid], labels[b_of, i_of])
    assert int(valid.sum()) == B * n


def test_product_refuses_cpu_and_out_of_scope_features():
    m = A.SemanticTransformer(dim=64, depth=1, num_semantic_tokens=10, flash_attn=True)
    with pytest.raises(RuntimeError):
        m(ids=torch.randint(0, 10, (2, 5)))
    with pytest.raises(NotImplementedError):
        A.SemanticTransformer(dim=64, depth=1, num_semantic_tokens=10, has_condition=True)


def test_soundstream_module_tree_matches_reference_state_dict_names():
    """The tokenize-path mirror keeps the reference's parameter / buffer names and shapes for `encoder.*` and `rq.*` (fixture shapes
    come from the REAL reference module), so reference checkpoints load by name; options outside the hot path are refused."""
    import os
    import pytest
    import torch
    from common import GOLDEN_DIR
    import audiolm_pytorch_amd as A
    fx = torch.load(os.path.join(GOLDEN_DIR, 'soundstream_small.pt'), weights_only=False)
    ss = A.SoundStream(**fx['ctor'])
    ours = {k: tuple(v.shape) for k, v in ss.state_dict().items()}
    assert ours == {k: tuple(v) for k, v in fx['shapes'].items()}
    assert ss.seq_len_multiple_of == 320 and ss.rq_groups == 1 and ss.num_quantizers == fx['ctor']['rq_num_quantizers']
    with pytest.raises(NotImplementedError):
        A.SoundStream(codebook_size=32)                              # reference default use_local_attn=True is SURVEY §8(f)-3
    with pytest.raises(RuntimeError):
        ss.tokenize(torch.zeros(1, 640))                              # CPU tensor: no CPU fallback
