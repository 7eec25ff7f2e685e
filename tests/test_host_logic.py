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


def test_fine_wrapper_bookkeeping_bit_exact():
    """FineTransformerWrapper.forward (audiolm_pytorch.py:2041-2137): flatten, labels = (coarse, fine) un-shifted, fine input [:, :-1],
    forgetful mask of width nc + nf_in + 2, no eos appended, ids stay int64."""
    fx = _load('fine_s1_bias_mask')
    model = A.FineTransformer(**fx['ctor'])
    seen = _capture(model)
    w = A.FineTransformerWrapper(transformer=model, codec=Codec(), mask_prob=0.15)
    w.train()
    inp = fx['inputs']
    orig = AP.generate_mask_with_prob
    AP.generate_mask_with_prob = lambda shape, prob, device: inp['forgetful_mask'].clone()
    try:
        w(coarse_token_ids=inp['coarse_token_ids'], fine_token_ids=inp['fine_token_ids'], return_loss=True)
    finally:
        AP.generate_mask_with_prob = orig
    b = inp['coarse_token_ids'].shape[0]
    coarse, fine = inp['coarse_token_ids'].reshape(b, -1), inp['fine_token_ids'].reshape(b, -1)
    k = seen['kwargs']
    assert torch.equal(k['coarse_token_ids'], coarse) and torch.equal(k['fine_token_ids'], fine[:, :-1])
    assert torch.equal(k['labels'][0], coarse) and torch.equal(k['labels'][1], fine)
    assert torch.equal(k['self_attn_mask'], inp['forgetful_mask'])
    assert k['self_attn_mask'].shape == (b, coarse.shape[1] + fine.shape[1] - 1 + 2)
    assert k['coarse_token_ids'].dtype == torch.int64 and k['fine_token_ids'].dtype == torch.int64


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
    md = A.SemanticTransformer(dim=64, depth=1, num_semantic_tokens=10, attn_dropout=0.1)        # round 3: attention dropout is built
    assert md.transformer.attn_dropout == 0.1 and md.transformer.layers[0][0].branch.attend.dropout == 0.1
    with pytest.raises(RuntimeError):                                                            # standalone modules: MI355X only, like everything else
        md.transformer.layers[0][0].branch(torch.zeros(1, 4, 64))
    with pytest.raises(RuntimeError):
        md.transformer.layers[0][2].branch(torch.zeros(1, 4, 64))
    # conditioning is native from pre-computed text embeddings; the T5 text encoder itself is out of scope
    mc = A.SemanticTransformer(dim=64, depth=1, num_semantic_tokens=10, has_condition=True)
    assert 'transformer.layers.0.1.branch.null_kv' in mc.state_dict()
    with pytest.raises(NotImplementedError):
        mc.embed_text(['a dog barking'])
    with pytest.raises(AssertionError):                          # reference :687: a conditioned model needs its conditioning
        mc(ids=torch.randint(0, 10, (2, 5)))


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
    fd = torch.load(os.path.join(GOLDEN_DIR, 'soundstream_decode_small.pt'), weights_only=False)          # decoder.* + rq.* of the REAL reference
    want = {k: tuple(v) for k, v in fx['shapes'].items()}
    want.update({k: tuple(v) for k, v in fd['shapes'].items()})
    assert ours == want, (sorted(set(ours) ^ set(want))[:10])
    assert ss.seq_len_multiple_of == 320 and ss.rq_groups == 1 and ss.num_quantizers == fx['ctor']['rq_num_quantizers']
    dflt = A.SoundStream(codebook_size=32)                           # the reference-default constructor: use_local_attn=True
    fl = torch.load(os.path.join(GOLDEN_DIR, 'soundstream_local_attn_small.pt'), weights_only=False)
    la = A.SoundStream(**fl['ctor'])
    assert {k: tuple(v.shape) for k, v in la.state_dict().items()} == {k: tuple(v) for k, v in fl['shapes'].items()}   # names / shapes of the REAL reference module tree
    assert 'encoder_attn.layers.0.0.attn_fn.rel_pos.inv_freq' in dflt.state_dict() and dflt.encoder_attn.layers[0][0].to_qkv.weight.shape == (1536, 512)
    with pytest.raises(NotImplementedError):
        A.SoundStream(codebook_size=32, attn_dynamic_pos_bias=True)
    with pytest.raises(RuntimeError):
        ss.tokenize(torch.zeros(1, 640))                              # CPU tensor: no CPU fallback


def _dense_from_table(tbl, index, scale):
    """what the attention kernels compute from (table, index vectors): bias(h,i,j) = special ? tbl[h][0] : tbl[h][(qkey4[i]-kkey4[j])/4]"""
    qkey4, kkey4, qattr, kattr = [t.long() for t in index]
    LT = tbl.shape[1]
    slot = (qkey4[:, None] - kkey4[None, :]) // 4
    special = (qattr[:, None] & kattr[None, :]) != 0
    inside = (slot >= 0) & (slot < LT)
    vals = tbl[:, torch.where(special, torch.zeros_like(slot), slot).clamp(0, LT - 1)]
    return torch.where((inside | special)[None], vals, torch.zeros_like(vals)) * scale


def test_structured_bias_index_vectors_reproduce_the_reference_bias():
    """relpos.toeplitz_index / fine_index + the MLP output laid out as a table == the dense (h, n, n) tensors the reference gathers
    (oracle.rel_pos_bias :202-242, the Coarse override :924-936, oracle.fine_attn_bias :1229-1298) on every causally visible pair."""
    from audiolm_pytorch_amd import relpos
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7)
    H, d, scale = 4, 8, 64 ** -0.5
    cpu = torch.device('cpu')

    def mlp_sd(prefix, in_dim, nhid, seq):
        sd, dims = {}, [in_dim] + [d] * (nhid + 1)
        for li in range(nhid + 1):
            name = f'{prefix}{li}.0.' if seq is None else f'{prefix}{2 * li}.'
            sd[name + 'weight'], sd[name + 'bias'] = torch.randn(dims[li + 1], dims[li], generator=g) * 0.3, torch.randn(dims[li + 1], generator=g) * 0.3
        name = f'{prefix}{nhid + 1}.' if seq is None else f'{prefix}{2 * (nhid + 1)}.'
        sd[name + 'weight'], sd[name + 'bias'] = torch.randn(H, d, generator=g) * 0.3, torch.randn(H, generator=g) * 0.3
        return sd

    # ---- Semantic / Coarse
    n = 29
    sd = mlp_sd('rel_pos_bias.net.', 1, 2, None)
    want = O.rel_pos_bias(sd, 'rel_pos_bias.', n, n)                                          # (h, n, n)
    x = torch.arange(-n + 1, n).float()[:, None]
    for li in range(3):
        x = F.silu(F.linear(x, sd[f'rel_pos_bias.net.{li}.0.weight'], sd[f'rel_pos_bias.net.{li}.0.bias']))
    T = F.linear(x, sd['rel_pos_bias.net.3.weight'], sd['rel_pos_bias.net.3.bias'])           # (2n-1, h): PosTableFn's rows
    cross = torch.randn(H, 1, 1, generator=g)
    tbl = torch.cat((cross.reshape(H, 1), T.t()), dim=1) / scale
    causal = torch.ones(n, n, dtype=torch.bool).tril()
    got = _dense_from_table(tbl, relpos.toeplitz_index(n, cpu), scale)
    assert torch.allclose(got[:, causal], want[:, causal], atol=1e-5)
    ns1 = 11                                                                                  # semantic_seq_len + 1 (:929)
    is_sem = torch.arange(n) < ns1
    want_c = torch.where(is_sem[:, None] ^ is_sem[None, :], cross, want)
    got_c = _dense_from_table(tbl, relpos.toeplitz_index(n, cpu, num_leading=ns1), scale)
    assert torch.allclose(got_c[:, causal], want_c[:, causal], atol=1e-5)

    # ---- Fine (ragged coarse / fine lengths included)
    Qc, Qf = 3, 5
    for nc, nf in ((12, 20), (11, 17), (3, 1), (30, 4), (2, 14)):
        sdf = mlp_sd('pos_bias_mlp.', 2, 1, 'seq')
        sdf['null_pos_bias'] = torch.randn(H, 1, 1, generator=g)
        cfg = O.Cfg(dim=2 * d, depth=1, heads=H, streams=1, num_semantic_tokens=0, codebook_size=16, num_coarse_quantizers=Qc, num_fine_quantizers=Qf)
        want_f = O.fine_attn_bias(sdf, cfg, nc, nf, cpu)
        grid, index = relpos.fine_index(nc, nf, Qc, Qf, cpu)
        h = F.silu(F.linear(grid, sdf['pos_bias_mlp.0.weight'], sdf['pos_bias_mlp.0.bias']))
        h = F.silu(F.linear(h, sdf['pos_bias_mlp.2.weight'], sdf['pos_bias_mlp.2.bias']))
        Tf = F.linear(h, sdf['pos_bias_mlp.4.weight'], sdf['pos_bias_mlp.4.bias'])
        tblf = torch.cat((sdf['null_pos_bias'].reshape(H, 1), Tf.t()), dim=1) / scale
        N = nc + nf + 2
        causal = torch.ones(N, N, dtype=torch.bool).tril()
        got_f = _dense_from_table(tblf, index, scale)
        assert torch.allclose(got_f[:, causal], want_f[:, causal], atol=1e-5), (nc, nf)
        for t in index:
            assert t.dtype == torch.int32 and t.shape == (N,)


def test_get_optimizer_grouping_matches_reference_rule():
    """optimizer.py:3-37: wd == 0 -> Adam over the plain parameter list; wd > 0 -> AdamW with weight decay only on ndim >= 2 parameters."""
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3))
    o = A.get_optimizer(m.parameters(), lr=3e-4, wd=0.)
    assert len(o.param_groups) == 1 and o.param_groups[0]['weight_decay'] == 0 and not o.param_groups[0]['decoupled_weight_decay']
    assert o.param_groups[0]['lr'] == 3e-4 and o.param_groups[0]['betas'] == (0.9, 0.99) and o.param_groups[0]['eps'] == 1e-8
    o = A.get_optimizer(m.parameters(), lr=1e-4, wd=1e-2)
    assert len(o.param_groups) == 2
    assert all(p.ndim >= 2 for p in o.param_groups[0]['params']) and o.param_groups[0]['weight_decay'] == 1e-2
    assert all(p.ndim < 2 for p in o.param_groups[1]['params']) and o.param_groups[1]['weight_decay'] == 0
    assert all(g['decoupled_weight_decay'] for g in o.param_groups)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(Exception):                 # CPU parameters are refused at step time (no CPU fallback)
        o.step()


def test_install_as_reference_registers_the_reference_import_paths():
    """after install_as_reference() the reference trainer's own import lines (trainer.py:32-48) resolve to this
    package.  Run in a subprocess: it rewires sys.modules."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import audiolm_pytorch_amd as A\n"
        "A.install_as_reference()\n"
        "from audiolm_pytorch.audiolm_pytorch import SemanticTransformer, CoarseTransformer, FineTransformer, SemanticTransformerWrapper, "
        "CoarseTransformerWrapper, FineTransformerWrapper\n"
        "from audiolm_pytorch.soundstream import SoundStream\n"
        "from audiolm_pytorch.optimizer import get_optimizer\n"
        "from audiolm_pytorch.attend import Attend\n"
        "assert SemanticTransformer is A.SemanticTransformer and SoundStream is A.SoundStream and get_optimizer is A.get_optimizer\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]


def test_sampling_helpers_semantics():
    """audiolm_pytorch.py:96-130: top-k keeps max(int((1 - thres) * n), 1) logits, the rest -inf; everything after the first eos is masked
    (the eos itself too when keep_eos is False); all_rows_have_eos_id."""
    lg = torch.tensor([[0.1, 2.0, -1.0, 0.5, 1.5, 0.0, -2.0, 0.3, 0.2, 0.05]])
    out = AP.top_k(lg, thres=0.9)                                     # k = max(int(0.1 * 10), 1) = 1
    assert int(torch.isfinite(out).sum()) == 1 and float(out[0, 1]) == 2.0
    out = AP.top_k(lg, thres=0.7)                                     # k = 3
    assert sorted(torch.isfinite(out)[0].nonzero().flatten().tolist()) == [1, 3, 4]
    t = torch.tensor([[3, 9, 4, 9, 1], [1, 2, 3, 4, 5], [9, 1, 1, 1, 1]])
    assert AP.mask_out_after_eos_id(t, 9, keep_eos=False).tolist() == [[3, -1, -1, -1, -1], [1, 2, 3, 4, 5], [-1, -1, -1, -1, -1]]
    assert AP.mask_out_after_eos_id(t, 9, keep_eos=True).tolist() == [[3, 9, -1, -1, -1], [1, 2, 3, 4, 5], [9, -1, -1, -1, -1]]
    assert not bool(AP.all_rows_have_eos_id(t, 9)) and bool(AP.all_rows_have_eos_id(t[[0, 2]], 9))
    torch.manual_seed(0)
    s = AP.gumbel_sample(torch.tensor([[0.0, 50.0, 0.0]]), temperature=1.)
    assert int(s) == 1


def test_boundary_signatures_match_reference_snapshot():
    """every constructor / forward / generate of the drop-in classes accepts the reference's parameters -- same names, same
    order, same kinds, same defaults (tests/golden/signatures.json, taken from the REAL reference by make_golden.py signatures).  Extra
    keyword parameters of ours (labels=, return_flat_hidden=, zero_pad-like extensions) must come with defaults."""
    import inspect
    import json
    import audiolm_pytorch_amd.attend as AT
    import audiolm_pytorch_amd.optimizer as OPT
    import audiolm_pytorch_amd.soundstream as SS
    ref = json.load(open(os.path.join(GOLDEN_DIR, 'signatures.json')))
    owners = {'SemanticTransformer': AP, 'CoarseTransformer': AP, 'FineTransformer': AP, 'Transformer': AP, 'SemanticTransformerWrapper': AP,
              'CoarseTransformerWrapper': AP, 'FineTransformerWrapper': AP, 'AudioLM': AP, 'Attend': AT, 'SoundStream': SS}
    problems = []
    for key, want in sorted(ref.items()):
        if key == 'get_optimizer':
            fn = OPT.get_optimizer
        else:
            cls, meth = key.split('.')
            fn = getattr(getattr(owners[cls], cls), meth)
        if fn.__name__ == 'inner' and fn.__closure__:
            fn = [c.cell_contents for c in fn.__closure__ if callable(c.cell_contents)][0]
        have = inspect.signature(fn).parameters
        var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in have.values())
        pos_ref = [n for n, kind, _ in want if kind == 'POSITIONAL_OR_KEYWORD']
        pos_have = [n for n, p in have.items() if p.kind is inspect.Parameter.POSITIONAL_OR_KEYWORD]
        if pos_have[:len(pos_ref)] != pos_ref:
            problems.append(f'{key}: positional parameters {pos_have} vs reference {pos_ref}')
        for name, kind, default in want:
            if kind in ('VAR_POSITIONAL', 'VAR_KEYWORD'):
                continue
            if name not in have:
                if not var_kw:
                    problems.append(f'{key}: parameter {name!r} missing')
                continue
            p = have[name]
            if p.kind.name != kind:
                problems.append(f'{key}: {name} is {p.kind.name}, reference {kind}')
            d = '<required>' if p.default is inspect.Parameter.empty else p.default
            if isinstance(d, tuple):
                d = list(d)
            if default != '<object>' and d != default and not (isinstance(d, float) and isinstance(default, (int, float)) and d == default):
                problems.append(f'{key}: default of {name} is {d!r}, reference {default!r}')
        for name, p in have.items():                       # our extensions never become mandatory
            if name not in {n for n, _, _ in want} and p.kind in (inspect.Parameter.POSITIONAL_OR_KEYWORD, inspect.Parameter.KEYWORD_ONLY):
                if p.default is inspect.Parameter.empty:
                    problems.append(f'{key}: extra parameter {name!r} has no default')
    assert not problems, '\n'.join(problems)


@pytest.mark.parametrize('shape,prob', [((4, 64), 0.15), ((2, 7), 0.9), ((3, 5), 0.0), ((1, 2048), 0.15), ((2, 3), 1.0)])
def test_forgetful_mask_matches_oracle_draw_for_draw(shape, prob):
    """int(n p) keys capped at n - 1, key 0 never masked, and the SAME keys as the reference for the same seed."""
    torch.manual_seed(123)
    ours = AP.generate_mask_with_prob(shape, prob, 'cpu')
    torch.manual_seed(123)
    ref = O.generate_mask_with_prob(shape, prob, 'cpu')
    assert ours.dtype == torch.bool and torch.equal(ours, ref)
    assert bool(ours[:, 0].all())
    assert bool(((~ours).sum(-1) == min(int(shape[-1] * prob), shape[-1] - 1)).all())


def test_integer_helpers_match_oracle_on_random_inputs():
    """append_eos_id / batch_unique_consecutive (ragged -> padded, pad -1) against the oracle's restatement of audiolm_pytorch.py:155-164;
    mask_out_after_eos_id against a position-by-position definition; dtype stays int64."""
    g = torch.Generator().manual_seed(5)
    for _ in range(20):
        b, n, vocab = int(torch.randint(1, 5, (1,), generator=g)), int(torch.randint(1, 30, (1,), generator=g)), int(torch.randint(2, 6, (1,), generator=g))
        ids = torch.randint(0, vocab, (b, n), generator=g)
        assert torch.equal(AP.append_eos_id(ids, vocab), O.append_eos_id(ids, vocab))
        ours, ref = AP.batch_unique_consecutive(ids, pad_value=-1), O.batch_unique_consecutive(ids, pad_value=-1)
        assert ours.dtype == torch.int64 and torch.equal(ours, ref)
        for keep in (True, False):
            m = AP.mask_out_after_eos_id(ids, 1, mask_value=-7, keep_eos=keep)
            for r in range(b):
                row = ids[r].tolist()
                first = row.index(1) if 1 in row else None
                want = list(row)
                if first is not None:
                    for j in range(first + (1 if keep else 0), n):
                        want[j] = -7
                assert m[r].tolist() == want
    lg = torch.randn(6, 40, generator=g)
    for thres in (0.0, 0.5, 0.9, 0.999):
        out = AP.top_k(lg, thres)
        k = max(int((1 - thres) * 40), 1)
        assert bool((torch.isfinite(out).sum(-1) == k).all())
        assert torch.equal(out.max(-1).values, lg.max(-1).values) and bool((out[torch.isfinite(out)] == lg[torch.isfinite(out)]).all())


def test_eval_decorator_restores_mode_even_on_error():
    class M(torch.nn.Module):
        @AP.eval_decorator
        def ok(self):
            return self.training

        @AP.eval_decorator
        def boom(self):
            raise ValueError('x')
    m = M().train()
    assert m.ok() is False and m.training is True
    with pytest.raises(ValueError):
        m.boom()
    assert m.training is True
    m.eval()
    m.ok()
    assert m.training is False


def test_api_compat_helpers_get_embeds_and_grad_shrink():
    """exported for API compatibility (not on the product path): same values as the oracle's restatement of audiolm_pytorch.py:93-94, 168-186"""
    g = torch.Generator().manual_seed(3)
    emb = torch.nn.Embedding(7, 5)
    codes = torch.randint(-1, 7, (3, 11), generator=g)
    ours, mask = AP.get_embeds(emb, codes, pad_id=-1, return_mask=True)
    ref = O.get_embeds(emb.weight, codes, pad_id=-1)
    assert torch.equal(ours, ref) and torch.equal(mask, codes != -1)
    keep_row0 = AP.get_embeds(emb, codes, pad_id=-1, mask_pad_pos_to=None)
    assert torch.equal(keep_row0[codes == -1], emb.weight[0].expand(int((codes == -1).sum()), -1))
    x = torch.randn(4, 3, generator=g, requires_grad=True)
    y = AP.grad_shrink(x, alpha=0.1)
    assert torch.allclose(y, x, atol=1e-7)
    y.sum().backward()
    assert torch.allclose(x.grad, torch.full_like(x, 0.1))


def test_shape_derived_index_tensors_are_cached_and_plain():
    """index tensors that depend only on shapes are built once per (shape, device): same object on the second call, never inference tensors (the first call
    may happen under torch.inference_mode(), the next under autograd), and equal to the straightforward construction"""
    dev = torch.device('cpu')
    for fn in (AP._const_code, AP._neg, AP._quantizer_rows, AP._quantizer_row_offsets, AP._quantizer_codes, AP._group_index, AP._group_cols):
        fn.cache_clear()
    with torch.inference_mode():
        a = AP._quantizer_codes(2, 3, 5, 7, 3, dev)
        idx, ig, va = AP._group_index(2, 20, 4, 10, 3, dev)
    b = AP._quantizer_codes(2, 3, 5, 7, 3, dev)
    assert a is b and not a.is_inference() and not idx.is_inference()
    assert AP._group_index(2, 20, 4, 10, 3, dev)[0] is idx
    rows = torch.arange(7) % 3
    ref = torch.cat((torch.full((3, 5), -1, dtype=torch.int32), ((2 << 24) + rows).to(torch.int32)[None].expand(3, -1)), dim=1)
    assert torch.equal(a, ref) and a.dtype == torch.int32 and a.is_contiguous()
    assert torch.equal(AP._quantizer_row_offsets(7, 3, 1024, dev), (1024 * rows).to(torch.int32)[None])
    assert torch.equal(AP._group_cols(10, 3, dev), (torch.arange(4)[None, :] * 3 + torch.arange(3)[:, None]).clamp(max=9).reshape(-1))
    assert torch.equal(AP._const_code(4, 2, dev), torch.full((2, 1), 4 << 24, dtype=torch.int32))
    # a different shape is a different entry
    assert AP._quantizer_codes(2, 3, 5, 8, 3, dev).shape == (3, 13)


def test_oracle_feedforward_dropout_with_supplied_mask():
    """the oracle's restated nn.Dropout (mask supplied): a ones-mask with p = 0 is the plain feed-forward, and a 0 / 1 mask with p scales the kept
    activations of the inner LayerNorm output by 1 / (1 - p) -- checked against F.dropout's definition on the same tensor"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    d, inner = 24, int(24 * 8 / 3)
    sd = {'f.0.gamma': torch.rand(d, generator=g) + 0.5, 'f.1.weight': torch.randn(2 * inner, d, generator=g) * 0.2,
          'f.3.gamma': torch.rand(inner, generator=g) + 0.5, 'f.5.weight': torch.randn(d, inner, generator=g) * 0.2}
    x = torch.randn(2, 7, d, generator=g)
    plain = O.feedforward(sd, 'f.', x)
    assert torch.equal(O.feedforward(sd, 'f.', x, torch.ones(2, 7, inner), 0.), plain)
    p = 0.3
    keep = (torch.rand(2, 7, inner, generator=g) >= p).float()
    h = O.layer_norm(x, sd['f.0.gamma'])
    a, gate = F.linear(h, sd['f.1.weight']).chunk(2, dim=-1)
    hn = O.layer_norm(F.gelu(gate) * a, sd['f.3.gamma'])
    want = F.linear(hn * keep / (1 - p), sd['f.5.weight'])
    assert torch.allclose(O.feedforward(sd, 'f.', x, keep, p), want, atol=1e-6)
