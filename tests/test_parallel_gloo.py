"""CPU, world_size 2, gloo: the data-parallel gradient engine (audiolm_pytorch_amd.parallel) averages gradients exactly like a
single process seeing both shards -- for loose parameters (hooks) and for the fused-stack per-layer callback path."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeStack(nn.Module):
    """Stands in for Transformer: exposes depth / flat_params() / _layer_grad_hook and calls the hook layer by layer in backward order."""

    def __init__(self, depth=3):
        super().__init__()
        self.depth = depth
        self.ws = nn.ParameterList([nn.Parameter(torch.randn(4, 4)) for _ in range(depth * 2)])
        self.norm = nn.Parameter(torch.ones(4))
        self._layer_grad_hook = None

    def flat_params(self):
        return list(self.ws) + [self.norm]


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Embedding(10, 4)
        self.transformer = FakeStack()
        self.head = nn.Linear(4, 3)
        self.unused = nn.Parameter(torch.zeros(2))          # never receives a gradient (like proj_text_embed)


def _loss(model, ids):
    x = model.embed(ids)
    for w in model.transformer.ws:
        x = torch.tanh(x @ w)
    x = x * model.transformer.norm
    return model.head(x).square().mean()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    torch.manual_seed(1234 + rank)                          # deliberately different init: the engine must broadcast rank 0's weights
    model = Model()
    eng = DataParallelEngine(model, dist)
    ids_all = torch.arange(12).reshape(2, 6) % 10
    ids = ids_all[rank:rank + 1]
    loss = _loss(model, ids)
    loss.backward()
    # emulate the fused stack's per-layer callbacks (reverse layer order), as core.stack_backward does
    flat = model.transformer.flat_params()
    for l in reversed(range(model.transformer.depth)):
        eng._on_layer_grads(l, [p.grad.clone() for p in flat[l * 2:(l + 1) * 2]])
    eng.finish()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    if rank == 0:
        torch.save(dict(sd=sd, grads=grads, ids=ids_all), out)
    dist.barrier()
    dist.destroy_process_group()


def _worker_accum(rank, world, port, out):
    """two micro-steps: the first inside no_sync(), the second synchronising -> mean over ranks of the SUM of both micro-step gradients"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    torch.manual_seed(99)
    model = Model()
    eng = DataParallelEngine(model, dist)
    ids_all = torch.arange(24).reshape(4, 6) % 10                  # rows 0, 1: micro-step 0 of ranks 0, 1; rows 2, 3: micro-step 1
    flat = model.transformer.flat_params()

    def micro(ids):
        before = [None if p.grad is None else p.grad.clone() for p in flat]
        _loss(model, ids).backward()
        for l in reversed(range(model.transformer.depth)):
            fresh = [p.grad - b if b is not None else p.grad.clone() for p, b in zip(flat[l * 2:(l + 1) * 2], before[l * 2:(l + 1) * 2])]
            eng._on_layer_grads(l, fresh)
        eng.finish()
    with eng.no_sync():
        micro(ids_all[rank:rank + 1])
    micro(ids_all[2 + rank:3 + rank])
    assert not eng._inflight and not eng._dirty
    # a following ordinary step must take the overlapped path again
    for p in model.parameters():
        p.grad = None
    micro(ids_all[rank:rank + 1])
    g3 = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    with eng.no_sync():
        micro(ids_all[rank:rank + 1])
    micro(ids_all[2 + rank:3 + rank])
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    if rank == 0:
        torch.save(dict(sd={k: v.detach().clone() for k, v in model.state_dict().items()}, grads=grads, g3=g3, ids=ids_all), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_engine_gradient_accumulation_no_sync(tmp_path):
    out = str(tmp_path / 'acc.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_accum, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    ids = r['ids']
    ((_loss(model, ids[0:1]) + _loss(model, ids[1:2])) / 2 + (_loss(model, ids[2:3]) + _loss(model, ids[3:4])) / 2).backward()
    for k, p in model.named_parameters():
        if k == 'unused':
            assert r['grads'][k] is None
            continue
        assert torch.allclose(r['grads'][k], p.grad, atol=1e-6), k
    model.zero_grad()
    ((_loss(model, ids[0:1]) + _loss(model, ids[1:2])) / 2).backward()
    for k, p in model.named_parameters():
        if k != 'unused':
            assert torch.allclose(r['g3'][k], p.grad, atol=1e-6), k


def test_dp_engine_world2_matches_big_batch(tmp_path):
    out = str(tmp_path / 'r0.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    losses = [_loss(model, r['ids'][i:i + 1]) for i in range(2)]
    (sum(losses) / 2).backward()
    for k, p in model.named_parameters():
        if k == 'unused':
            assert r['grads'][k] is None
            continue
        assert torch.allclose(r['grads'][k], p.grad, atol=1e-6), k


def _worker_bf16_none_grad(rank, world, port, out):
    """bf16 buckets + parameters whose .grad is None when finish() runs (the fused stack hands fresh gradient tensors to the callback BEFORE autograd
    has assigned p.grad): finish() must create fp32 .grad tensors from the bf16 bucket (`p.grad = v.clone()` made them bf16)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    torch.manual_seed(5)
    model = Model()
    eng = DataParallelEngine(model, dist, bucket_dtype=torch.bfloat16)
    ids_all = torch.arange(12).reshape(2, 6) % 10
    _loss(model, ids_all[rank:rank + 1]).backward()
    flat = model.transformer.flat_params()
    fresh = [[p.grad.clone() for p in flat[l * 2:(l + 1) * 2]] for l in range(model.transformer.depth)]
    for p in flat[:-1]:
        p.grad = None                                              # as inside the fused backward: the stack's .grad does not exist yet
    for l in reversed(range(model.transformer.depth)):
        eng._on_layer_grads(l, fresh[l])
    eng.finish()
    st = eng.last_stats
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    dtypes = {k: (str(p.grad.dtype) if p.grad is not None else None) for k, p in model.named_parameters()}
    if rank == 0:
        torch.save(dict(sd={k: v.detach().clone() for k, v in model.state_dict().items()}, grads=grads, dtypes=dtypes, ids=ids_all, stats=st), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_engine_bf16_buckets_with_missing_grads(tmp_path):
    out = str(tmp_path / 'bf.pt')
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_worker_bf16_none_grad, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    (sum(_loss(model, r['ids'][i:i + 1]) for i in range(2)) / 2).backward()
    for k, p in model.named_parameters():
        if k == 'unused':
            assert r['grads'][k] is None
            continue
        assert r['dtypes'][k] == 'torch.float32', (k, r['dtypes'][k])
        assert torch.allclose(r['grads'][k], p.grad, atol=2e-2 * float(p.grad.abs().max()) + 1e-6), k          # bf16 wire format
    assert r['stats']['buckets'] >= 3 and r['stats']['bytes'] > 0 and r['stats']['tail_ms'] >= 0.0, r['stats']


def _worker_groups(rank, world, port, out):
    """round 4: the fused stack hands over whole LAYER GROUPS (core.stack_backward's deferred weight-gradient mode: `on_group(layers, grads per layer)`,
    layers in backward order) -- one bucket per group; the hook object the engine installs is callable per layer too (`engine(layer, grads)`)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    torch.manual_seed(77 + rank)
    model = Model()
    eng = DataParallelEngine(model, dist)
    hook = model.transformer._layer_grad_hook
    assert hook is eng and callable(hook) and callable(getattr(hook, 'on_group', None))
    ids_all = torch.arange(12).reshape(2, 6) % 10
    _loss(model, ids_all[rank:rank + 1]).backward()
    flat = model.transformer.flat_params()
    fresh = [[p.grad.clone() for p in flat[l * 2:(l + 1) * 2]] for l in range(model.transformer.depth)]
    for p in flat[:-1]:
        p.grad = None                                              # as inside the fused backward: the stack's .grad does not exist yet
    hook.on_group([2], [fresh[2]])                                  # depth 3, two groups: {2}, then {1, 0}
    hook.on_group([1, 0], [fresh[1], fresh[0]])
    eng.finish()
    st = eng.last_stats
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    # a second step through the per-layer entry of the same hook object
    for p in model.parameters():
        p.grad = None
    _loss(model, ids_all[rank:rank + 1]).backward()
    fresh = [[p.grad.clone() for p in flat[l * 2:(l + 1) * 2]] for l in range(model.transformer.depth)]
    for p in flat[:-1]:
        p.grad = None
    for l in reversed(range(model.transformer.depth)):
        hook(l, fresh[l])
    eng.finish()
    grads2 = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    if rank == 0:
        torch.save(dict(sd={k: v.detach().clone() for k, v in model.state_dict().items()}, grads=grads, grads2=grads2, ids=ids_all, stats=st), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_engine_layer_group_buckets_match_big_batch(tmp_path):
    out = str(tmp_path / 'grp.pt')
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_worker_groups, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    (sum(_loss(model, r['ids'][i:i + 1]) for i in range(2)) / 2).backward()
    for k, p in model.named_parameters():
        if k == 'unused':
            assert r['grads'][k] is None and r['grads2'][k] is None
            continue
        assert torch.allclose(r['grads'][k], p.grad, atol=1e-6), k
        assert torch.allclose(r['grads2'][k], p.grad, atol=1e-6), k
    assert r['stats']['buckets'] == 1 + 2, r['stats']               # [every loose parameter: here the whole backward ran before the hand-off] + 2 layer groups


def _worker_flat_groups(rank, world, port, out):
    """round 5: in-place group buckets.  core.stack_backward asks `group_buffer(l0, l1, dense)` where the batched weight-gradient GEMMs should write;
    with the views the bucket is produced in place: `on_group` only copies the group's remaining (small) gradients into the tail and all-reduces the
    flat buffer; `finish()` hands out VIEWS of the bucket as .grad (no torch.cat staging, no copy back).  Emulated here exactly as core does it: layer
    = (dense weight w_a, small parameter w_b); the dense gradients are written into the views, the small ones are handed over as separate tensors."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    torch.manual_seed(31 + rank)
    model = Model()
    eng = DataParallelEngine(model, dist)
    flat = model.transformer.flat_params()
    ids_all = torch.arange(12).reshape(2, 6) % 10
    res = []
    for step in range(2):                                           # the persistent buffers are re-used by the second step
        for p in model.parameters():
            p.grad = None
        _loss(model, ids_all[rank:rank + 1] if step == 0 else (ids_all[rank:rank + 1] + 3) % 10).backward()
        fresh = [[p.grad.clone() for p in flat[l * 2:(l + 1) * 2]] for l in range(model.transformer.depth)]
        for p in flat[:-1]:
            p.grad = None                                           # as inside the fused backward: the stack's .grad does not exist yet
        for (l0, l1) in ((2, 3), (0, 2)):                           # depth 3, two groups in backward order: {2}, then {1, 0}
            views = eng.group_buffer(l0, l1, [0])                   # one dense kind: slot 0 = the first weight of every layer
            assert views is not None and views[0].shape == (l1 - l0, 4, 4)
            for l in range(l0, l1):
                views[0][l - l0].copy_(fresh[l][0])                 # "the GEMM writes its output"
            layers = list(range(l1 - 1, l0 - 1, -1))
            eng.on_group(layers, [[views[0][l - l0], fresh[l][1]] for l in layers])
        eng.finish()
        st = eng.last_stats
        # the handed-out gradients are views of the persistent buckets
        aliased = all(any(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * 4 for f, _ in eng._flat_groups.values()) for p in flat[:-1])
        res.append(dict(grads={k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}, stats=st, aliased=aliased))
    # while gradients accumulate the engine must NOT hand out the persistent buffer (a .grad may alias it)
    with eng.no_sync():
        refused = eng.group_buffer(2, 3, [0]) is None
    eng._bw_started = False
    if rank == 0:
        torch.save(dict(sd={k: v.detach().clone() for k, v in model.state_dict().items()}, res=res, ids=ids_all, refused=refused), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_engine_in_place_group_buckets(tmp_path):
    out = str(tmp_path / 'flat.pt')
    port = 23500 + (os.getpid() % 2000)
    mp.spawn(_worker_flat_groups, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    for step, ids in enumerate((r['ids'], (r['ids'] + 3) % 10)):
        model.zero_grad()
        (sum(_loss(model, ids[i:i + 1]) for i in range(2)) / 2).backward()
        rs = r['res'][step]
        for k, p in model.named_parameters():
            if k == 'unused':
                assert rs['grads'][k] is None
                continue
            assert torch.allclose(rs['grads'][k], p.grad, atol=1e-6), (step, k)
        assert rs['stats']['buckets'] == 1 + 2 and rs['stats'].get('direct_buckets') == 2, rs['stats']
        assert rs['aliased']
    assert r['refused']


def _worker_flat_groups_late(rank, world, port, out):
    """in-place group buckets with the SMALL weight kinds late (core.stack_backward, DEFER_SMALL_LATE): the early group's bucket holds only its layers' big
    kind (slot 0; slot 1 is skipped), the last group's bucket holds its own big kind plus slot 1 of ALL layers; every gradient is handed over exactly once."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    torch.manual_seed(41 + rank)
    model = Model()
    eng = DataParallelEngine(model, dist)
    flat = model.transformer.flat_params()
    L = model.transformer.depth
    ids_all = torch.arange(12).reshape(2, 6) % 10
    _loss(model, ids_all[rank:rank + 1]).backward()
    fresh = [[p.grad.clone() for p in flat[l * 2:(l + 1) * 2]] for l in range(L)]
    for p in flat[:-1]:
        p.grad = None
    # early group {2}: big kind only
    v = eng.group_buffer(2, 3, [0], [], [1])
    assert v is not None and v[0].shape == (1, 4, 4)
    v[0][0].copy_(fresh[2][0])
    eng.on_group([2], [[v[0][0], None]])
    # last group {1, 0}: its big kind + the late kind of all three layers
    v = eng.group_buffer(0, 2, [0], [1], [])
    assert v is not None and v[0].shape == (2, 4, 4) and v[1].shape == (3, 4, 4)
    for l in (0, 1):
        v[0][l].copy_(fresh[l][0])
    for l in range(L):
        v[1][l].copy_(fresh[l][1])
    eng.on_group([2, 1, 0], [[None, v[1][2]], [v[0][1], v[1][1]], [v[0][0], v[1][0]]])
    eng.finish()
    st = eng.last_stats
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    if rank == 0:
        torch.save(dict(sd={k: v_.detach().clone() for k, v_ in model.state_dict().items()}, grads=grads, ids=ids_all, stats=st), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_engine_in_place_group_buckets_with_late_small_kinds(tmp_path):
    out = str(tmp_path / 'late.pt')
    port = 21500 + (os.getpid() % 2000)
    mp.spawn(_worker_flat_groups_late, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    (sum(_loss(model, r['ids'][i:i + 1]) for i in range(2)) / 2).backward()
    for k, p in model.named_parameters():
        if k == 'unused':
            assert r['grads'][k] is None
            continue
        assert torch.allclose(r['grads'][k], p.grad, atol=1e-6), k
    assert r['stats']['buckets'] == 1 + 2 and r['stats'].get('direct_buckets') == 2, r['stats']


def _worker_uneven_groups(rank, world, port, out):
    """round 6: UNEVEN layer groups (core.group_starts with ALM_DP_GROUP_SIZES, backward order): depth 3 cut as (2, 1) -- the top two layers form the
    first bucket, the LAST (exposed) bucket is the single bottom layer.  Same in-place protocol as _worker_flat_groups; afterwards the .grad-is-a-view
    contract: a gradient tensor kept across steps makes the next synchronising backward raise."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    from audiolm_pytorch_amd.core import group_starts
    torch.manual_seed(53 + rank)
    model = Model()
    eng = DataParallelEngine(model, dist)
    flat = model.transformer.flat_params()
    L = model.transformer.depth
    starts = group_starts(L, 2, (2, 1))
    assert starts == [0, 1], starts
    ranges = [(s, ([t for t in starts if t > s] + [L])[0]) for s in reversed(starts)]          # backward order: (1, 3), then (0, 1)
    assert ranges == [(1, 3), (0, 1)], ranges
    ids_all = torch.arange(12).reshape(2, 6) % 10
    res = []
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        _loss(model, ids_all[rank:rank + 1] if step == 0 else (ids_all[rank:rank + 1] + 5) % 10).backward()
        fresh = [[p.grad.clone() for p in flat[l * 2:(l + 1) * 2]] for l in range(L)]
        for p in flat[:-1]:
            p.grad = None
        for (l0, l1) in ranges:
            views = eng.group_buffer(l0, l1, [0])
            assert views is not None and views[0].shape == (l1 - l0, 4, 4)
            for l in range(l0, l1):
                views[0][l - l0].copy_(fresh[l][0])
            layers = list(range(l1 - 1, l0 - 1, -1))
            eng.on_group(layers, [[views[0][l - l0], fresh[l][1]] for l in layers])
            del views
        eng.finish()
        res.append(dict(grads={k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}, stats=eng.last_stats))
    # the contract: keep one handed-out gradient across the step boundary -> the next synchronising backward refuses to overwrite it
    kept = flat[0].grad
    for p in model.parameters():
        p.grad = None
    raised = False
    try:
        _loss(model, ids_all[rank:rank + 1]).backward()
    except RuntimeError as e:
        raised = 'VIEW of a persistent all-reduce bucket' in str(e)
    del kept
    eng._bw_started = False
    # ... and with nothing kept the same backward goes through
    for p in model.parameters():
        p.grad = None
    _loss(model, ids_all[rank:rank + 1]).backward()
    eng.finish()
    if rank == 0:
        torch.save(dict(sd={k: v.detach().clone() for k, v in model.state_dict().items()}, res=res, ids=ids_all, raised=raised), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_engine_uneven_layer_groups_and_view_contract(tmp_path):
    out = str(tmp_path / 'uneven.pt')
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_worker_uneven_groups, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    model = Model()
    model.load_state_dict(r['sd'])
    for step, ids in enumerate((r['ids'], (r['ids'] + 5) % 10)):
        model.zero_grad()
        (sum(_loss(model, ids[i:i + 1]) for i in range(2)) / 2).backward()
        rs = r['res'][step]
        for k, p in model.named_parameters():
            if k == 'unused':
                assert rs['grads'][k] is None
                continue
            assert torch.allclose(rs['grads'][k], p.grad, atol=1e-6), (step, k)
        st = rs['stats']
        assert st['buckets'] == 1 + 2 and st.get('direct_buckets') == 2, st
        # bucket order on the wire: [loose (handed over BEFORE the first group's GEMMs), top group of 2 layers, bottom group of 1 layer]; the exposed
        # tail is the small last bucket alone, and the model prices exactly its bytes
        bb = st['bucket_bytes']
        assert len(bb) == 3 and bb[1] > bb[2], bb
        assert st['exposed_bucket_bytes'] == [bb[2]], st
        tm = st['tail_model_ms']
        assert tm['bytes'] == bb[2] and tm['world'] == 8 and tm['ring'] >= tm['direct'] >= 0, tm
    assert r['raised']


def test_group_starts_uneven_and_equal():
    sys.path.insert(0, ROOT)
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd.core import group_starts
    from audiolm_pytorch_amd.parallel import tail_model
    assert group_starts(6, 2) == [0, 3] and group_starts(6, 1) == [0] and group_starts(6, 6) == list(range(6)) and group_starts(5, 2) == [0, 3]
    assert group_starts(6, 2, (4, 2)) == [0, 2]                    # backward order: layers 5..2 first, the exposed last bucket = layers 1..0
    assert group_starts(6, 2, (3, 2, 1)) == [0, 1, 3]
    for bad in ((4, 3), (6, 0), (2, 2)):
        try:
            group_starts(6, 2, bad)
        except ValueError:
            continue
        raise AssertionError(bad)
    # SURVEY.md section 5's figures: 262 MB over a ring of 8 -> 3.0 ms, direct over 7 links -> 0.43 ms
    tm = tail_model(262e6)
    assert abs(tm['ring'] - 3.0) < 0.05 and abs(tm['direct'] - 0.43) < 0.01, tm
