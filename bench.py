"""Headline benchmark: audio-tokens/sec, forward + backward of CoarseTransformer (dim 1024, depth 6, 3 coarse quantizers, 4 residual
streams, flash_attn=True) over synthetic SoundStream-RVQ token sequences with transformer length N = 2048 (BASELINE.json metric,
configs[1]/[3]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; per-rank batch B = 8 (weak scaling); a "step" = CoarseTransformerWrapper.forward(return_loss=True) +
backward (+ RCCL gradient all-reduce over xGMI for N > 1), inputs resident in HBM, the bf16 weight copies are re-packed from
the fp32 masters every step (as after an optimiser step).  The optimiser itself is outside the metric ("fwd+bwd"); a second
timed loop that includes the clip + Adam step is reported as `with_optimizer` for reference.

--config selects the other BASELINE.json configurations / SURVEY §8(d) rows (same JSON contract, each with its own roofline and CPU
baseline): coarse2048 (default, the metric's configuration), coarse1024 (configs[1]), fine2049 (configs[2]), fine_t2048_q8 (the other
reading of "(B=8, Q=8, seq=2048)": a 2048-frame x 8-quantizer grid, N = 16385), e2e_config5 (configs[4]: SoundStream tokenize of
8 x 30 s @ 24 kHz + CoarseTransformer codebook 4096, N = 8253).

--schedule: eager = every launch issued from Python each step; graph = the step captured once into a hipGraph (graphed.GraphedTrainStep)
and replayed; graph2 = the same with the batch split into two half-batches on two HIP streams (HBM-bound row kernels of one half under
the MFMA-bound GEMMs of the other); eager2 = that split without the graph.  auto (default) = eager: measured, the captured schedules are
SLOWER here (see main()); N > 1 ranks always run eager (the gradient all-reduce is launched from backward callbacks).  The arithmetic is
the same in all of them; the JSON line says which ran.

Extra objects on the JSON line:
  roofline     dominant kernel = the bf16 MFMA NT GEMM on the big tiles (gemm_stag_kernel<NT> 256x256x64 / gemm_kernel<384,256,...,NT>); achieved = algorithmic GEMM FLOPs per launch / average
               launch duration, measured live with HIP events on the launch stream over one instrumented EAGER step; peak = 2500 TFLOP/s.
  cpu_baseline the oracle (CPU fp32 restatement of the reference, "port") timed on this box's host cores on a bounded sample
               (B = 1, same architecture), rank 0 at N = 1 only, with 4 residual streams and with 1 (pure reference code).
  parity       the HIP path's loss on that same sample (same weights, ids, forgetful mask) vs the oracle's: asserted within 1e-3.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

B_PER_GPU = 8
PEAK_BF16_TFLOPS = 2500.0                          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
METRIC = 'audio-tokens/sec fwd+bwd, CoarseTransformer d=1024 seq=2048'
COARSE = dict(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True)
FINE = dict(dim=1024, depth=6, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024, flash_attn=True)


class Codec:
    rq_groups = 1
    num_quantizers = 8


def flops_per_token(kind, N, cb=1024):
    """algorithmic fwd+bwd FLOPs per token: dense GEMMs + causal attention + logit heads + hyper-connections, x2 FLOP/MAC, x3"""
    dense = 6 * 9566208
    attn = 6 * 2 * 8 * 64 * (N + 1) / 2
    heads = 0.92e6 * (cb + 1) / 1025 if kind == 'coarse' else 1.05e6
    return (dense + attn + heads + 0.59e6) * 2 * 3


def build(config, dev, rank, residual_dtype):
    """-> dict(kind, model, wrapper, inputs (device tensors, batch-first), N, metric, workload, sample = (kind, ctor, N_sample builder))"""
    import audiolm_pytorch_amd as A
    g = torch.Generator().manual_seed(1000 + rank)              # a different synthetic shard per rank
    torch.manual_seed(0)                                        # identical initial weights on every rank
    B = B_PER_GPU
    if config in ('coarse2048', 'coarse1024', 'e2e_config5'):
        ctor = dict(COARSE)
        codec = Codec()
        if config == 'e2e_config5':
            ctor['codebook_size'] = 4096
            ss = A.SoundStream(codebook_size=4096, rq_num_quantizers=8, target_sample_hz=24000, strides=(2, 4, 5, 8), use_local_attn=False)
            with torch.no_grad():
                for r in ss.rq.rvqs:
                    for q, l in enumerate(r.layers):
                        l._codebook.embed.copy_(torch.randn(1, 4096, 512) * (0.5 ** q))
                        l._codebook.initted.fill_(True)
            codec = ss.to(dev)
        model = A.CoarseTransformer(**ctor, residual_dtype=residual_dtype).to(dev)
        wrapper = A.CoarseTransformerWrapper(transformer=model, codec=codec, unique_consecutive=False, mask_prob=0.15)
        if config == 'e2e_config5':
            n_sem, N = 1500, 1 + 1501 + 1 + 6750
            inputs = dict(semantic_token_ids=torch.randint(0, 500, (B, n_sem), generator=g).to(dev),
                          raw_wave=(torch.randn(B, 720000, generator=g) * 0.1).to(dev))
            metric = 'audio-tokens/sec, SoundStream tokenize + CoarseTransformer fwd+bwd end to end (BASELINE configs[4])'
            sample_kw = dict(n_sem=1500, n_fr=2250)              # the bench-internal parity / cpu_baseline sample at this configuration's own N
            work = (f'SoundStream(codebook 4096, 8 quantizers, 24 kHz, strides 2-4-5-8) tokenize of {B} x 30 s synthetic audio + CoarseTransformer dim=1024 '
                    f'depth=6 codebook=4096 fwd+bwd; N = 1 + 1501 + 1 + 6750 = {N}; mask_prob=0.15')
        else:
            n_sem, n_fr = (509, 512) if config == 'coarse2048' else (253, 256)
            N = 1 + (n_sem + 1) + 1 + n_fr * 3
            sample_kw = dict(n_sem=n_sem, n_fr=n_fr)              # the parity / cpu_baseline sample at this configuration's own N
            inputs = dict(semantic_token_ids=torch.randint(0, 500, (B, n_sem), generator=g).to(dev),
                          coarse_token_ids=torch.randint(0, 1024, (B, n_fr, 3), generator=g).to(dev))
            metric = METRIC if config == 'coarse2048' else 'audio-tokens/sec fwd+bwd, CoarseTransformer d=1024 seq=1024 (BASELINE configs[1])'
            work = (f'CoarseTransformer dim=1024 depth=6 heads=8 (MQA) num_coarse_quantizers=3 codebook=1024 semantic=500 residual_streams=4 flash_attn=True; '
                    f'per-GPU B={B}, N={N} ({n_sem} semantic + {n_fr}x3 coarse ids + eos/start); mask_prob=0.15')
        kind = 'coarse'
    elif config in ('fine2049', 'fine_t2048_q8'):
        T = 256 if config == 'fine2049' else 2048
        model = A.FineTransformer(**FINE, residual_dtype=residual_dtype).to(dev)
        wrapper = A.FineTransformerWrapper(transformer=model, codec=Codec(), mask_prob=0.15)
        grid = torch.randint(0, 1024, (B, T, 8), generator=g).to(dev)
        inputs = dict(coarse_token_ids=grid[..., :3].contiguous(), fine_token_ids=grid[..., 3:].contiguous())
        N = 1 + 3 * T + 1 + 5 * T - 1
        metric = f'audio-tokens/sec fwd+bwd, FineTransformer d=1024 seq={N}' + (' (BASELINE configs[2])' if T == 256 else ' (2048 frames x 8 quantizers)')
        work = (f'FineTransformer dim=1024 depth=6 heads=8 (MQA) 3 coarse + 5 fine quantizers codebook=1024 residual_streams=4 flash_attn=True; per-GPU B={B}, '
                f'{T} frames x 8 quantizers -> N={N}; mask_prob=0.15')
        kind, ctor = 'fine', dict(FINE)
    else:
        raise SystemExit(f'unknown --config {config}')
    wrapper.train()
    return dict(kind=kind, ctor=ctor, model=model, wrapper=wrapper, inputs=inputs, N=N, B=B, metric=metric, workload=work, sample_kw=locals().get('sample_kw'))


def oracle_sample(kind, ctor, sd, streams, seed=0, n_sem=509, n_fr=512):
    """One B = 1 sample of the same architecture through the CPU oracle, by default at the metric's sequence length class (N = 2048 Coarse / 2049 Fine),
    for `--config e2e_config5` at that configuration's OWN length (n_sem = 1500, n_fr = 2250: N = 8253):
    -> (loss tensor with graph, params dict, inputs for the HIP side)"""
    import audiolm_oracle as O
    g = torch.Generator().manual_seed(seed)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith('.beta')}
    full = dict(sd)
    full.update(params)
    if kind == 'coarse':
        cfg = O.Cfg(dim=ctor['dim'], depth=ctor['depth'], streams=streams, num_semantic_tokens=500, codebook_size=ctor['codebook_size'], num_coarse_quantizers=3)
        sem = torch.randint(0, 500, (1, n_sem), generator=g)
        coarse = torch.randint(0, ctor['codebook_size'], (1, n_fr, 3), generator=g)
        Ns = 1 + (n_sem + 1) + 1 + 3 * n_fr
        mask = O.generate_mask_with_prob((1, Ns), 0.15, 'cpu', generator=g)
        fn = lambda: O.coarse_wrapper_loss(full, cfg, sem, coarse, training=True, unique_consecutive=False, forgetful_mask=mask)   # noqa: E731
        return fn, params, dict(semantic_token_ids=sem, coarse_token_ids=coarse), mask, Ns
    cfg = O.Cfg(dim=ctor['dim'], depth=ctor['depth'], streams=streams, codebook_size=1024, num_coarse_quantizers=3, num_fine_quantizers=5)
    grid = torch.randint(0, 1024, (1, 256, 8), generator=g)
    c, f = grid[..., :3].contiguous(), grid[..., 3:].contiguous()
    mask = O.generate_mask_with_prob((1, 2049), 0.15, 'cpu', generator=g)
    fn = lambda: O.fine_wrapper_loss(full, cfg, c, f, forgetful_mask=mask)                                                         # noqa: E731
    return fn, params, dict(coarse_token_ids=c, fine_token_ids=f), mask, 2049


def _port_ratio():
    """oracle tokens/s over real-reference tokens/s measured back to back in the build container (profiles/r5_cpu_reference_vs_oracle.json, 4 streams)"""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r5_cpu_reference_vs_oracle.json')) as fh:
            d = json.load(fh)
        r = next(x for x in d['runs'] if x['streams'] == 4)
        return dict(ratio=r['oracle_over_reference'], source='profiles/r5_cpu_reference_vs_oracle.json (scripts/cpu_reference_time.py, build container, same cores)')
    except Exception:
        return None


def cpu_baseline_and_parity(W, dev, max_seconds=30.0):
    """Times the CPU oracle (fp32) on a bounded sample of the same architecture (B = 1, N = 2048 / 2049, fwd + bwd; 4 streams and 1 stream) and
    checks the HIP path's loss on EXACTLY that sample (same weights, ids and forgetful mask) against it."""
    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import audiolm_pytorch as AP
    cores = min(os.cpu_count() or 1, 32)          # one GPU's share of the host (256 cores / 8 GPUs); more threads oversubscribe torch's CPU ops
    torch.set_num_threads(cores)
    kind, ctor, model = W['kind'], W['ctor'], W['model']
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    out = {}
    skw = dict(W.get('sample_kw') or {})       # e2e_config5: the oracle sample has the configuration's own N = 8253 (one fwd + bwd, about a minute of CPU)
    fn, params, ids, mask, Ns = oracle_sample(kind, ctor, sd, 4, **skw)
    times, t_start, loss4 = [], time.time(), None
    for _ in range(3):
        t0 = time.time()
        loss = fn()
        loss.backward()
        times.append(time.time() - t0)
        loss4 = float(loss)
        for p in params.values():
            p.grad = None
        if time.time() - t_start > max_seconds:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    # kind "port": /root/reference cannot travel to the GPU box.  scripts/cpu_reference_time.py (build container) times the REAL shimmed reference and this
    # oracle back to back on the same cores: profiles/r5_cpu_reference_vs_oracle.json -- the oracle runs the 4-stream step 1.06x as fast as the reference
    # (identical loss bits), the 1-stream step 0.76x
    out['cpu_baseline'] = dict(value=round(Ns / best, 1), unit='audio-tokens/s', cores=cores, kind='port',
                               port_over_reference=_port_ratio(),
                               sample=f'oracle (CPU fp32 restatement of the reference, 4 residual streams) fwd+bwd, B=1 x N={Ns}, best of {max(1, len(times) - 1)} after 1 warm-up')
    del params
    # the HIP path on the same sample
    orig = AP.generate_mask_with_prob
    AP.generate_mask_with_prob = lambda shape, prob, device: mask.to(device).clone()
    try:
        with torch.no_grad(), W['amp']():
            lh = float(W['wrapper'](**{k: v.to(dev) for k, v in ids.items()}, return_loss=True))
    finally:
        AP.generate_mask_with_prob = orig
    rel = abs(lh - loss4) / abs(loss4)
    out['parity'] = dict(loss_hip=round(lh, 6), loss_oracle=round(loss4, 6), rel=float(f'{rel:.3e}'), bound=1e-3, ok=bool(rel <= 1e-3), sample_N=Ns,
                         # the bounds the test suite ENFORCES (north_star's "<= 1e-3 for bf16 tensors" holds op by op, not for end-to-end logits of a
                         # 6-layer bf16 stack: DESIGN section 5): what is asserted where
                         enforced_bounds=dict(loss_end_to_end=1e-3, opwise_forward=1e-3, opwise_backward_bf16_streams=4e-3,
                                              logits_end_to_end="<= the reference's own bf16-autocast deviation on the same inputs (0.9-4.7e-2)",
                                              token_id_bookkeeping='bit-exact', tests='tests/test_gpu_fullsize.py, test_gpu_opwise.py, test_gpu_opwise_model.py, test_host_logic.py'),
                         sample=f'same weights / ids / forgetful mask as the cpu_baseline sample (B=1, N={Ns}' +
                                ('' if Ns == W['N'] else f'; the timed configuration runs N={W["N"]}: the flash kernels at that length are covered by '
                                                         f'tests/test_gpu_kernels.py::test_mqa_attention_long_sequences_vs_chunked_fp64') + ')')
    if Ns > 4096:                                                # one pass of the long sample is already past the time bound: no second architecture
        return out
    # 1 residual stream = every op is first-party reference code (no restated hyper-connections): timing only, fresh weights of that architecture
    try:
        K = A.CoarseTransformer if kind == 'coarse' else A.FineTransformer
        torch.manual_seed(0)
        sd1 = {k: v.detach().clone() for k, v in K(**ctor, num_residual_streams=1).state_dict().items()}
        fn1, params1, _, _, _ = oracle_sample(kind, ctor, sd1, 1)
        t1 = []
        for _ in range(2):
            t0 = time.time()
            fn1().backward()
            t1.append(time.time() - t0)
        out['cpu_baseline']['streams1'] = dict(value=round(Ns / min(t1), 1), unit='audio-tokens/s',
                                               sample=f'oracle with num_residual_streams=1 (pure reference code), B=1 x N={Ns}, best of 2')
    except Exception as e:                                       # the baseline is a reported number, never a reason to lose the bench line
        out['cpu_baseline']['streams1'] = dict(error=str(e)[:200])
    return out


def csrc_digest():
    """digest of the kernel sources + the C-ABI header (audiolm-pytorch_amd/build.py: what decides whether the library is rebuilt)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_alm_build', os.path.join(ROOT, 'audiolm-pytorch_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod._digest()[:16]


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x 2 per the MI355X guide's gfx950
    correction + WRITE_SIZE) -- read from profiles/, NOT measured in this run (counters cannot be collected inside this process;
    scripts/pmc.sh regenerates them).  -> (bytes | None, file name | None, provenance text).  Since round 4 the summary records the commit and the
    digest of the kernel sources it was measured on; a summary whose digest differs from the sources of THIS run is refused (traffic: null) --
    the numbers would describe other kernels."""
    for name in ('r6_pmc_summary.json', 'r5_pmc_summary.json', 'r4_pmc_summary.json', 'r3_pmc_summary.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as fh:
                d = json.load(fh)
        except Exception:
            continue
        dig, commit = d.get('csrc_digest'), d.get('commit')
        if dig is None:
            return None, name, f'profiles/{name} carries no source digest (generated before round 4): refused, traffic not reported'
        if dig != csrc_digest():
            return None, name, (f'profiles/{name} was measured at commit {commit} on kernel sources with digest {dig}; this run\'s sources have digest '
                                f'{csrc_digest()}: refused (regenerate: bash scripts/gpu_final.sh <tag> <commit>)')
        return d['hbm_bytes_per_launch'], name, (f'profiles/{name}: rocprofv3 PMC passes of this workload at commit {commit} (kernel-source digest {dig} == this '
                                                 f'run\'s), FETCH_SIZE x 2 + WRITE_SIZE per the MI355X guide, one counter group per run; not re-measured in this run')
    return None, None, 'no PMC summary under profiles/'



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='coarse2048', choices=['coarse2048', 'coarse1024', 'fine2049', 'fine_t2048_q8', 'e2e_config5'])
    ap.add_argument('--schedule', default='auto', choices=['auto', 'eager', 'eager2', 'graph', 'graph2'])
    ap.add_argument('--residual', default='auto', choices=['auto', 'bf16', 'fp32'],
                    help='HBM storage of the 4 hyper-connection residual streams.  auto (default) = the SHIPPED default: the model is built without '
                         'residual_dtype and the forward runs inside torch.autocast(bfloat16) like reference trainer.py:1241 -> bf16 streams (what autocast '
                         'gives the reference); bf16 / fp32 = built with that storage explicitly, no autocast context')
    ap.add_argument('--bucket-dtype', default='auto', choices=['auto', 'bf16', 'fp32'],
                    help='wire format of the gradient all-reduce buckets for N > 1.  auto = fp32: what the reference reduces (DDP / accelerate all-reduce fp32 '
                         'gradients), so the scaling numbers are like for like; bf16 halves the bytes on the xGMI ring (131 instead of 262 MB/step) at the price of a '
                         'bf16 reduction across ranks -- an explicitly labelled variant (config.dp.bucket_dtype)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-optimizer-leg', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('ALM_BENCH_LAUNCH_CHECK') == '1':
        # launcher self-test (tests/test_bench_launch.py, runs without a GPU): every rank joins a gloo group, one all-reduce proves the rendezvous the
        # self-launch set up works, rank 0 prints ONE JSON line.  Nothing is measured.
        import datetime
        import torch.distributed as dist
        assert world == args.gpus, (world, args.gpus)
        if world > 1:
            dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t) == world
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'launch_check': True, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup}), flush=True)
        return
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    # test hook (scripts/gpu_check.sh dp2): ALM_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo, to exercise the multi-rank control flow
    # (callbacks, bucket order, barriers) on a 1-GPU box; its numbers mean nothing
    share = os.environ.get('ALM_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        print(f'[bench rank {rank}/{world}] init_process_group on cuda:{local_rank}', file=sys.stderr, flush=True)
        if share:
            dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=300))      # backend "nccl" IS RCCL on ROCm
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started {world} ranks (python bench.py --gpus N starts its own N ranks when no launcher did)'

    from audiolm_pytorch_amd import core, graphed, launchlist, ops, parallel
    import audiolm_pytorch_amd as A

    W = build(args.config, dev, rank, {'bf16': torch.bfloat16, 'fp32': torch.float32, 'auto': None}[args.residual])
    # --residual auto: the training call runs under autocast exactly as trainer.py:1241 (`with self.accelerator.autocast(): loss = train_wrapper(...)`);
    # it changes nothing in the HIP launches (they take raw pointers) except that the default-constructed model then stores bf16 residual streams
    import contextlib
    amp = (lambda: torch.autocast('cuda', dtype=torch.bfloat16)) if args.residual == 'auto' else contextlib.nullcontext
    W['amp'] = amp
    model, wrapper, inputs, N = W['model'], W['wrapper'], W['inputs'], W['N']
    bucket_dtype = torch.bfloat16 if args.bucket_dtype == 'bf16' else torch.float32
    engine = parallel.DataParallelEngine(model, dist, bucket_dtype=bucket_dtype) if world > 1 else None
    cache = model.transformer._cache
    params = list(model.parameters())                           # (as optimizer.zero_grad(set_to_none=True) holds them: no module-tree walk per step)

    def eager_step(opt=None, repack=True):
        if repack:
            cache.store.clear()                                 # weights "changed": re-pack bf16 copies like after an optimiser step
        for p in params:
            p.grad = None
        with amp():
            loss = wrapper(**inputs, return_loss=True)
        loss.backward()
        if engine is not None:
            engine.finish()                                     # waits for the overlapped RCCL all-reduces, grads averaged in place
        if opt is not None:
            opt.step()
        return loss.detach()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- schedule
    schedule, note, gstep = 'eager', None, None
    want = args.schedule
    probe = None
    auto = want == 'auto'
    if auto:
        # measured on MI355X (profiles/r2_schedules.log) at the headline shape: eager 15.5 ms, graph 16.2 ms, graph2 17.1 ms per step -- there the
        # GPU, not the host, is the limit (host issue ~11 ms < 14 ms) and a replayed graph only adds a per-node gap to ~900 kernels.  The smaller
        # configurations are the other way round (coarse1024: 8.4 ms of GPU work against 8-9 ms of host issue), so `auto` PROBES: a few eager steps
        # and a few replays of the captured step, and the timed region below runs whichever was faster (both probe figures are reported).
        want = 'eager'
    if want == 'eager2':                                        # diagnostic: the two-half-batch schedule without a graph (host-bound)
        model.transformer.micro_batches = 2
        schedule = 'eager2'
    for _ in range(3):                                          # lazy initialisation before any capture
        eager_step()
    if want in ('graph', 'graph2') and world == 1:
        try:
            with amp():
                gstep = graphed.GraphedTrainStep(wrapper, inputs, micro_batches=2 if want == 'graph2' else 1)
            torch.cuda.synchronize()
            schedule = want
        except Exception as e:                                  # capture is an optimisation: report why it was not used, run eager
            note = f'{want} capture failed ({type(e).__name__}: {str(e)[:160]}); ran eager'
            print('[bench] ' + note, file=sys.stderr, flush=True)
            gstep = None
            torch.cuda.synchronize()
    elif want not in ('eager', 'eager2'):
        note = f'{want} is a single-GPU schedule (the gradient all-reduce is issued from backward callbacks); ran eager'

    def step(opt=None, repack=True):
        if gstep is not None:
            loss = gstep(**inputs)
            if opt is not None:
                opt.step()
            return loss
        return eager_step(opt, repack)

    def timed(nsteps, fn):
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    if auto and world == 1 and N <= 4096:
        for _ in range(8):
            eager_step()
        te = timed(6, eager_step) / 6 * 1e3
        probe = {'eager_ms': round(te, 3)}
        try:
            with amp():
                cand = graphed.GraphedTrainStep(wrapper, inputs, micro_batches=1)
            torch.cuda.synchronize()
            for _ in range(3):
                cand(**inputs)
            tg = timed(6, lambda: cand(**inputs)) / 6 * 1e3
            probe['graph_ms'] = round(tg, 3)
            if tg < 0.97 * te:                                  # a clear win only: the eager path keeps the weight-gradient side stream
                gstep, schedule = cand, 'graph'
            else:
                del cand
        except Exception as e:
            probe['graph_error'] = f'{type(e).__name__}: {str(e)[:120]}'
            torch.cuda.synchronize()

    # untimed priming in addition to --warmup: the first steps grow the caching allocator's pools (main + side stream) and run at ramping
    # clocks; measured on MI355X the step time only settles after ~10 steps (17.6 -> 16.4 ms).  The timed region below is exactly --steps steps.
    big = N > 4096
    for _ in range(3 if big else 10):
        loss = step()
    for _ in range(args.warmup):
        loss = step()
    dt = timed(args.steps, step)
    ms = dt / args.steps * 1e3
    tokens_per_s = world * W['B'] * N / (dt / args.steps)
    dp_stats = None
    if engine is not None:
        # one line per rank on stderr, so that a driver SCALE run is diagnosable from its tail: what went on the wire and how much of the exchange
        # did NOT hide under backward (host time from the launch of the last bucket to the return of finish())
        dp_stats = dict(engine.last_stats or {}, bucket_dtype=str(bucket_dtype).replace('torch.', ''), ms_per_step=round(ms, 3))
        # GPU-clock time of the LAST step between "backward's own kernels done" and "every bucket reduced and handed out": the part of the gradient
        # exchange that did not hide under backward (events recorded in finish(); read here, outside the timed region)
        dp_stats['exposed_tail_ms'] = engine.exposed_tail_ms()
        print(f'[bench rank {rank}/{world}] dp: {dp_stats["buckets"]} buckets, {dp_stats["bytes"] / 1e6:.1f} MB/step on the wire ({dp_stats["bucket_dtype"]}), '
              f'{dp_stats.get("direct_buckets", 0)} of them reduced in place, exposed tail {dp_stats["exposed_tail_ms"]} ms on the GPU clock (host: {dp_stats["tail_ms"]} ms after the last bucket launch), step {ms:.3f} ms, collectives issued from a side stream: '
              f'{any(sid != int(torch.cuda.current_stream(dev).cuda_stream) for sid in dp_stats.get("launch_streams", []))}', file=sys.stderr, flush=True)

    # host time to ISSUE one step (no synchronisation inside): eager launches vs one graph replay
    host = {}
    issue = []
    for _ in range(5):                                          # median of 5 single steps (one sample swings by +-1.5 ms with the host's other work)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eager_step()
        issue.append((time.perf_counter() - t0) * 1e3)
    host['eager_issue_ms'] = round(sorted(issue)[2], 3)
    host['eager_issue_ms_samples'] = [round(v, 3) for v in issue]
    host['launch_list'] = dict(launchlist.STATS, enabled=launchlist.ENABLED)       # stack passes sized / recorded / replayed so far (replayed > 0: the timed steps ran from the lists)
    torch.cuda.synchronize()
    if gstep is not None:
        t0 = time.perf_counter()
        gstep(**inputs)
        host['graph_replay_issue_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
        torch.cuda.synchronize()
        dte = timed(max(3, args.steps // 2), eager_step)
        host['eager_ms_per_step'] = round(dte / max(3, args.steps // 2) * 1e3, 3)

    # ---- roofline of the dominant kernel: one instrumented EAGER step, HIP events (torch.cuda.Event on the launch stream = torch's current
    # stream, which is the stream every alm_* launch uses) around every MFMA GEMM launch.  The dominant kernel is the NT big-tile GEMM (8 waves: gemm_stag_kernel
    # <false,*> 256x256x64 with staggered wave rows, or gemm_kernel<384,256,2,4,false,*> where pick_tile() prefers the coarser tiling): its
    # launches are singled out; all GEMM launches are reported alongside.
    # EVERY rank runs the instrumented step (with N > 1 it contains the gradient all-reduces: a step on rank 0 alone would dead-lock the
    # collectives); only rank 0's numbers are reported.
    roof = None
    events = []                                                  # (start event, end event, algorithmic work, unit, kernel key)
    timed_names = ('gemm_nt', 'gemm_nt_group2', 'gemm_tn_splitk', 'gemm_tn_batched', 'mqa_attn_fwd', 'mqa_attn_bwd', 'hc_fwd', 'hc_bwd', 'geglu_ln_fwd', 'geglu_ln_bwd', 'layernorm_fwd',
                   'layernorm_bwd', 'conv1d_causal', 'resunit_causal', 'rvq_encode')       # (the last three: the SoundStream tokenize kernels of --config e2e_config5)
    originals = {n: getattr(ops, n) for n in timed_names}

    def big_tile(M_, N_, nb):               # mirrors pick_tile() in csrc/gemm.hip
        return M_ >= 256 and N_ >= 256 and ((M_ + 255) // 256) * ((N_ + 255) // 256) * nb >= 192

    import ctypes as _ct
    from audiolm_pytorch_amd import _lib as _L
    _plan = (_ct.c_int * 4)()
    model_dim = int(W['ctor']['dim'])

    def nt_kind(M_, N_, K_, nb):
        """kernel key of an ops.gemm_nt launch, from the library's own plan query (alm_gemm_nt_plan): nt256 = a big tile on a launch that fills the chip (the
        dominant kernel), nt256sk = the staggered 256 x 256 tile on an UNDER-FILLED launch (round 6: in-launch split-K, or unsplit), nt128 = the 128 x 128 tile"""
        _L.query('alm_gemm_nt_plan', M_, N_, K_, nb, int(ops.NT_WS and M_ >= 256 and N_ >= 256), _ct.cast(_plan, _ct.c_void_p))
        if _plan[0] in (11, 13, 17):                     # 17 = 320 x 256 (round 6: ragged token counts)
            return 'nt256' if big_tile(M_, N_, nb) else 'nt256sk'
        return 'nt128'

    def esz(t):
        return 0 if t is None else t.element_size()

    def work_of(name, a, kw, out):
        """(algorithmic work of ONE launch, unit, kernel key): FLOPs for the MFMA kernels, HBM bytes (every operand read once, every result written
        once: DESIGN.md section 4) for the row kernels"""
        if name == 'gemm_nt':
            Am, Bm = a[0], a[1]
            nb = 1
            for d in Am.shape[:-2]:
                nb *= d
            Mm, Nn, Kk = Am.shape[-2], Bm.shape[-2], Am.shape[-1]
            # the split-bf16 logit heads run ONE K-concatenated GEMM [hi | hi | lo] x [Whi | Wlo | Whi] (heads.py): 3 K executed, K algorithmic
            k_alg = Kk // 3 if (a[2].dtype == torch.float32 and Kk == 3 * model_dim and Am.dim() == 3) else Kk
            return 2.0 * nb * Mm * Nn * k_alg, 'flop', nt_kind(Mm, Nn, Kk, nb)
        if name == 'gemm_nt_group2':                                             # two un-batched problems in one launch (to_q || to_kv, their two dgrads)
            (M0, K0), N0, (M1, K1), N1 = a[0].shape, a[1].shape[0], a[3].shape, a[4].shape[0]
            t0_, t1_ = ((M0 + 255) // 256) * ((N0 + 255) // 256), ((M1 + 255) // 256) * ((N1 + 255) // 256)
            grp_big = os.environ.get('ALM_GEMM_GROUP2_BIG', '0') != '0' and min(M0, N0, M1, N1) >= 256 and 192 <= t0_ + t1_ <= 256 and t0_ % 8 == 0   # mirrors alm_gemm_bf16_nt_group2
            return 2.0 * (M0 * N0 * K0 + M1 * N1 * K1), 'flop', ('nt256' if big_tile(M0, N0, 1) else 'nt256sk' if grp_big else 'nt128')
        if name == 'gemm_tn_splitk':
            At, Bt = a[0], a[1]
            nb = At.shape[0] if At.dim() == 3 else 1
            fl = 2.0 * nb * At.shape[-1] * Bt.shape[-1] * At.shape[-2]
            from audiolm_pytorch_amd import _lib as _L
            return fl, 'flop', ('tn128' if _L.query('alm_gemm_splitk_tile', At.shape[-1], Bt.shape[-1], At.shape[-2], nb) == 1 else 'tn256')
        if name == 'gemm_tn_batched':                                            # every layer's gradient of one weight kind in one launch
            At, Bt = a[0], a[1]
            n1, n2, Kk, Mm = At.shape
            from audiolm_pytorch_amd import _lib as _L
            return 2.0 * n1 * n2 * Mm * Bt.shape[-1] * Kk, 'flop', ('tn128' if _L.query('alm_gemm_splitk_tile', Mm, Bt.shape[-1], Kk, n1 * n2) == 1 else 'tn256')
        if name == 'conv1d_causal':                                              # implicit GEMM on the exact-fp32 MFMA: 2 B Cout Cin k Tout
            x = a[0]
            Bc, Cin, T = x.shape
            st_ = kw.get('stride', 1)
            return 2.0 * Bc * a[3] * Cin * a[4] * ((T - st_) // st_ + 1), 'flop32', 'conv1d_causal'
        if name == 'resunit_causal':                                             # fused ResidualUnit: k-tap dilated conv + k = 1 conv, both C -> C: 2 B T C C (k + 1)
            x = a[0]
            Bc, Cc, T = x.shape
            return 2.0 * Bc * T * Cc * Cc * (a[5] + 1), 'flop32', 'conv1d_causal'
        if name == 'rvq_encode':                                                 # distance GEMM of every quantizer stage: 2 T C d Q
            x, E = a[0], a[1]
            return 2.0 * x.shape[0] * E.shape[1] * E.shape[2] * E.shape[0], 'flop32', 'rvq_encode'
        if name in ('mqa_attn_fwd', 'mqa_attn_bwd'):
            Bq, Nq, Hq, dh = (a[4], a[5], a[6], a[7] if len(a) > 7 else 64) if name == 'mqa_attn_fwd' else (a[7], a[8], a[9], a[10] if len(a) > 10 else 64)
            fwd = 4.0 * Hq * dh * Nq * (Nq + 1) / 2 * Bq                            # causal: QK^T + PV over the lower triangle
            return (fwd, 'flop', 'mqa_fwd') if name == 'mqa_attn_fwd' else (2.5 * fwd, 'flop', 'mqa_bwd')
        if name == 'hc_fwd':
            R_in, Bq, Sq, Nq, Dq = a[0], a[1], a[2], a[3], a[4]
            M_ = Bq * Nq
            rsz = 2 if kw.get('r_dtype') == torch.bfloat16 else 4
            by = (4 * Dq if kw.get('rin_bcast') else Sq * Dq * rsz)
            if kw.get('y_prev') is not None:
                by += 2 * Dq + (0 if kw.get('final') else Sq * Dq * rsz)
            if kw.get('hc') is not None:
                by += 2 * Dq + (2 * Dq if kw.get('want_x', True) else 0) + 4 * 52
            if kw.get('final'):
                by += 4 * Dq + (4 * Dq if kw.get('final_f32') else 2 * Dq)
            return float(by) * M_, 'byte', 'hc_fwd'
        if name == 'hc_bwd':
            Bq, Sq, Nq, Dq = a[1], a[2], a[3], a[4]
            M_ = Bq * Nq
            rsz = 2 if kw.get('r_dtype') == torch.bfloat16 else 4
            by = 4 * Dq if kw.get('bcast') else Sq * Dq * rsz
            if kw.get('hc') is not None:
                by += (4 * Dq if kw.get('r_bcast') else Sq * Dq * rsz) + 2 * Dq + (2 * Dq if kw.get('extra') is not None else 0)
                by += 4 * Dq if kw.get('sum_only') else Sq * Dq * rsz
            if kw.get('y_prev') is not None:
                by += 2 * Dq + 2 * Dq
            return float(by) * M_, 'byte', 'hc_bwd'
        if name == 'geglu_ln_fwd':
            return 6.0 * a[3] * a[0].shape[0], 'byte', 'geglu_ln_fwd'              # read U (x | gate), write HN
        if name == 'geglu_ln_bwd':
            return 10.0 * a[6] * a[1].shape[0], 'byte', 'geglu_ln_bwd'             # read dHN, U; write dU
        if name == 'layernorm_fwd':
            x = a[0]
            return float(x.numel()) * (esz(x) + (4 if kw.get('out_f32') else 2) + (2 if kw.get('want_copy') else 0)), 'byte', 'layernorm_fwd'
        x = a[1]
        return float(x.numel()) * (esz(a[0]) + esz(x) + (2 if kw.get('extra') is not None else 0) + (2 if kw.get('dx_dtype') == torch.bfloat16 else 4)), 'byte', 'layernorm_bwd'

    def make_timed(name):
        fn = originals[name]

        def timed_fn(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            events.append((e0, e1) + work_of(name, a, kw, out))
            return out
        return timed_fn
    for n in timed_names:
        setattr(ops, n, make_timed(n))
    was_async, core.ASYNC_WGRAD = core.ASYNC_WGRAD, False      # per-kernel durations: no concurrent side-stream GEMMs in this step
    was_ll, launchlist.ENABLED = launchlist.ENABLED, False     # ... and every launch issued from Python through the wrappers above (a replayed launch list by-passes them)
    try:
        eager_step()
        torch.cuda.synchronize()
    finally:
        for n in timed_names:
            setattr(ops, n, originals[n])
        core.ASYNC_WGRAD = was_async
        launchlist.ENABLED = was_ll
    agg = {}
    for e0, e1, wk, unit, key in events:
        a = agg.setdefault(key, [0.0, 0.0, 0, unit])
        a[0] += e0.elapsed_time(e1)
        a[1] += wk
        a[2] += 1
    PEAK_F32_MFMA_TFLOPS = 157.0                       # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32; MI355X_MICROARCH.md)
    gem = {k: v for k, v in agg.items() if v[3] == 'flop' and k[:2] in ('nt', 'tn')}
    tot_ms = sum(a[0] for a in gem.values())
    tot_fl = sum(a[1] for a in gem.values())
    d_ms, d_fl, d_n, _ = agg.get('nt256', [0.0, 0.0, 0, 'flop'])
    PEAK_HBM_GBS = 8000.0
    desc = {'nt256': 'gemm_stag_kernel<NT> 256x256 / gemm_kernel<384,256,NT>: forward + dgrad GEMMs on the big tiles (dominant)',
            'nt128': 'NT GEMMs on the 128x128 tile or grouped launches (attention projections, logit heads)',
            'nt256sk': 'gemm_stag_inl_kernel / gemm_stag(_group2)_kernel<NT> on launches of < 192 tiles: in-launch split-K (last-arriver reduce) or unsplit',
            'tn256': 'gemm_stag_kernel<TN>: weight gradients on the big tile, all layers of a weight kind per launch (deferred mode) or split-K per layer', 'tn128': 'weight gradients on the 128x128 tile (split-K)',
            'mqa_fwd': 'mqa_fwd_kernel (causal flash attention forward)', 'mqa_bwd': 'attn_delta + mqa_bwd_dq + mqa_bwd_dkv (flash attention backward)',
            'hc_fwd': 'hc_fwd_kernel (depth + width connection + pre-LayerNorm, fused)', 'hc_bwd': 'hc_bwd_kernel (+ its colsum / param-grad launches)',
            'geglu_ln_fwd': 'geglu_ln_fwd_kernel', 'geglu_ln_bwd': 'geglu_ln_bwd_kernel (+ colsum)', 'layernorm_fwd': 'ln_fwd_kernel', 'layernorm_bwd': 'ln_bwd_kernel (+ colsum)',
            'conv1d_causal': 'resunit_kernel + conv1d_causal_kernel (SoundStream encoder: fused ResidualUnits (k7 dilated conv + ELU + k1 conv + ELU + residual, one launch) and the strided / in / out causal convs as implicit GEMMs on the exact-fp32 MFMA)',
            'rvq_encode': 'rvq_encode_kernel (residual VQ: fp32-MFMA distance GEMM + first-index argmin over 8 quantizers)'}
    kernels = []
    for key, (k_ms, k_w, k_n, unit) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        if k_ms <= 0:
            continue
        mf = unit in ('flop', 'flop32')
        ach = k_w / (k_ms * 1e-3) / (1e12 if mf else 1e9)
        peak = PEAK_BF16_TFLOPS if unit == 'flop' else PEAK_F32_MFMA_TFLOPS if unit == 'flop32' else PEAK_HBM_GBS
        kernels.append({'kernel': key, 'what': desc.get(key, key), 'bound': 'mfma' if mf else 'hbm', 'launches_per_step': k_n,
                        **({'mfma_dtype': 'fp32'} if unit == 'flop32' else {}),
                        'avg_launch_us': round(k_ms * 1e3 / k_n, 2), 'ms_per_step': round(k_ms, 3),
                        ('flop_per_launch' if mf else 'bytes_per_launch'): round(k_w / k_n, 0), 'achieved': round(ach, 1), 'peak': peak,
                        'unit': 'TFLOP/s' if mf else 'GB/s', 'frac': round(ach / peak, 4)})
    if d_n:
        ach = d_fl / (d_ms * 1e-3) / 1e12
        traffic, traffic_src, traffic_note = pmc_traffic()
        roof = {'bound': 'mfma', 'kernel': 'gemm_stag_kernel<NT> 256x256x64 + gemm_kernel<384,256,2,4,NT> (8-wave bf16 MFMA 32x32x16 big tiles; FFN / projection forward + dgrad GEMMs)',
                'achieved': round(ach, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
                'traffic': traffic if args.config == 'coarse2048' else None,
                'traffic_source': traffic_note if args.config == 'coarse2048' else None,
                'launches_per_step': d_n, 'avg_launch_us': round(d_ms * 1e3 / d_n, 2),
                'flop_per_launch_avg': round(d_fl / d_n, 0),
                'measured_in': 'one instrumented eager step, weight-gradient side stream off (kernels do not overlap); HIP events on the launch stream bracket each '
                               'host-level op (its launch gap and, for the row kernels, their small second-stage reductions included)',
                'all_gemm_launches': {'launches_per_step': sum(v[2] for v in gem.values()), 'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                                      'by_kind_ms': {k: round(v[0], 3) for k, v in gem.items()}},
                'kernels': kernels,
                'instrumented_ms_per_step': round(sum(v[0] for v in agg.values()), 3),
                'model_flops_frac': round(tokens_per_s / world * flops_per_token(W['kind'], N, W['ctor'].get('codebook_size', 1024)) / (PEAK_BF16_TFLOPS * 1e12), 4)}
    if dist is not None:
        dist.barrier()

    opt_leg = None
    if not args.no_optimizer_leg:
        # full training step as the reference trainer runs it (trainer.py:1241-1255): fwd + bwd (+ gradient all-reduce) + clip_grad_norm_(0.5)
        # + Adam (optimizer.py:get_optimizer, wd = 0 -> Adam).  Ours: FusedAdam (two HIP launches, norm kept on the device); beside it the
        # same step with torch.optim.Adam + torch.nn.utils.clip_grad_norm_ as the stock baseline.
        nst = max(3, args.steps // 2)

        opt_events = []

        def opt_step(opt, clip):
            # no forced re-pack here: a real optimiser step follows, the packed bf16 weight images go stale through the parameters' version counters
            # exactly as in training -- torch.optim.Adam: every dense weight is re-packed by the next forward; FusedAdam (round 4) writes the images
            # itself (alm_opt_adam_pack_step) and the forward finds them current
            step(repack=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            clip(opt)
            opt.step()
            e1.record()
            opt_events.append((e0, e1))

        def timed_opt(opt, clip):
            """-> (seconds for nst steps, diagnostics): GPU time of the clip + update kernels alone (HIP events) and the caching allocator's device-level
            traffic during the timed steps -- the `with_optimizer` figure has been bimodal between processes on one box (12.96 vs 14.75 ms for the same
            library in one visit, profiles/r4_*): these numbers say whether the optimiser kernels or the rest of the step move"""
            # untimed priming, like the main loop's: the first optimiser steps allocate the Adam state and re-shape the caching allocator's pools (the
            # step's working set now competes with 0.8 GB of state + the saved parameters for cached blocks).  With ONE priming step the timed region
            # still saw a device-level malloc per step (`diagnostics.device_mallocs`: 10 in 10 steps, each a millisecond-class synchronising call) in
            # whichever optimiser leg ran FIRST -- the "slow mode" of this figure in rounds 2-3 (+1.9 instead of +0.5 ms); the second leg, on settled
            # pools, never did.  Four priming steps settle them.
            for _ in range(4):
                opt_step(opt, clip)
            opt_events.clear()
            trace = os.environ.get('ALM_BENCH_ALLOC_TRACE') == '1'      # debugging aid: which Python frames make the caching allocator go to the device
            if trace:
                torch.cuda.memory._record_memory_history(max_entries=200000)
            ms0 = torch.cuda.memory_stats(dev)
            dt_ = timed(nst, lambda: opt_step(opt, clip))
            ms1 = torch.cuda.memory_stats(dev)
            if trace:
                snap = torch.cuda.memory._snapshot()
                torch.cuda.memory._record_memory_history(enabled=None)
                for tr_ in snap.get('device_traces', []):
                    for ev in tr_:
                        if ev.get('action') in ('segment_alloc', 'segment_free', 'segment_map', 'segment_unmap', 'oom'):
                            fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get('frames', []) if 'site-packages' not in f['filename']][:8]
                            print('[alloc-trace]', type(opt).__name__, ev['action'], ev.get('size'), fr, file=sys.stderr, flush=True)
            gpu_ms = sorted(a.elapsed_time(b) for a, b in opt_events)
            diag = dict(optimizer_gpu_ms_median=round(gpu_ms[len(gpu_ms) // 2], 3), optimizer_gpu_ms_max=round(gpu_ms[-1], 3),
                        device_mallocs=int(ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)),
                        device_frees=int(ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0)),
                        alloc_retries=int(ms1.get('num_alloc_retries', 0) - ms0.get('num_alloc_retries', 0)),
                        reserved_gb=round(ms1.get('reserved_bytes.all.current', 0) / 2 ** 30, 2))
            return dt_, diag
        state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        fused = A.get_optimizer(model.parameters(), lr=1e-5, wd=0.)
        dto, diag_f = timed_opt(fused, lambda o: o.clip_grad_norm_(0.5))
        del fused
        model.load_state_dict(state0)
        stock = torch.optim.Adam(model.parameters(), lr=1e-5, betas=(0.9, 0.99))
        dts, diag_s = timed_opt(stock, lambda o: torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5))
        del stock
        model.load_state_dict(state0)
        opt_leg = dict(ms_per_step=round(dto / nst * 1e3, 3), value=round(world * W['B'] * N / (dto / nst), 1),
                       optimizer='FusedAdam (alm_opt_grad_sumsq + alm_opt_adam_step + alm_opt_adam_pack_step: the dense weights\' bf16 images written by the update' + ('' if core.FUSED_ADAM_PACK else ' -- OFF') + '): clip_grad_norm_(0.5) + Adam',
                       torch_adam_ms_per_step=round(dts / nst * 1e3, 3), diagnostics=dict(fused=diag_f, torch_adam=diag_s, steps=nst))

    if rank == 0:
        out = {
            'metric': W['metric'],
            'value': round(tokens_per_s, 1),
            'unit': 'audio-tokens/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'bf16',
            'data': 'synthetic (uniform random semantic / acoustic RVQ token ids, random-init weights)',
            'config': {'workload': W['workload'], 'global_batch': world * W['B'], 'seq_len': N, 'parallelism': f'dp{world}',
                       'schedule': {'eager': ('eager: every kernel launched each step, no hipGraph -- the transformer stack\'s forward / backward launch sequences recorded once per '
                                              'shape and re-issued from one C call each (alm_list_run: same entry points, same order, live addresses; bit-identical to the '
                                              'Python-issued step), everything around the stack launched from Python') if (launchlist.ENABLED and launchlist.STATS['replayed']) else
                                             'eager (every kernel launched from Python each step)',
                                    'eager2': 'eager, two half-batches on two HIP streams inside the stack (diagnostic: host-bound)',
                                    'graph': 'one hipGraph replay per step (captured fwd + bwd)',
                                    'graph2': 'one hipGraph replay per step: two half-batches of 4 sequences on two HIP streams (row kernels of one half under the GEMMs '
                                              'of the other), gradients summed'}[schedule],
                       'residual_stream_storage': ('bf16 (shipped default: model built without residual_dtype, forward inside torch.autocast(bfloat16) like trainer.py:1241)'
                                                   if args.residual == 'auto' else 'bf16 (what autocast gives the reference)' if model.transformer.cfg.residual_bf16 else 'fp32'),
                       # which weight-gradient path the timed step took: the N = 1 line and the N > 1 lines of a scaling run are comparable because both
                       # are the deferred, layer-batched path (N > 1: cut into layer groups, one gradient bucket per group)
                       'wgrad_path': ('per-layer' if (not core.DEFER_WGRAD or (world > 1 and core.DP_DEFER_GROUPS == 0)) else
                                      f'deferred-g{core.DEFER_GROUPS if world == 1 else core.DP_DEFER_GROUPS}' +
                                      (lambda gs: ('[' + ','.join(map(str, gs)) + ']') if gs else '')(core.DEFER_GROUP_SIZES if world == 1 else core.dp_group_sizes(model.transformer.depth, core.DP_DEFER_GROUPS)))},
            'loss': round(float(loss), 4),
            'host': host,
        }
        if note:
            out['config']['schedule_note'] = note
        if probe:
            out['config']['schedule_probe'] = probe              # --schedule auto: ms/step of a few eager steps / graph replays; the faster one ran
        if roof:
            out['roofline'] = roof
            out['model_flops_frac'] = roof['model_flops_frac']     # whole step: algorithmic model FLOPs / s over the dense bf16 MFMA peak
        if dp_stats:
            out['config']['dp'] = dp_stats
        if opt_leg:
            out['with_optimizer'] = opt_leg
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline_and_parity(W, dev))
            if 'parity' in out:
                out['config']['parity_sample_N'] = out['parity']['sample_N']    # the sequence length the in-bench loss parity check ran at (== seq_len unless stated)
        print(json.dumps(out), flush=True)
        par = out.get('parity')
        if par is not None and not par['ok']:
            print(f'[bench] PARITY FAILURE: HIP loss {par["loss_hip"]} vs oracle {par["loss_oracle"]} (rel {par["rel"]} > 1e-3)', file=sys.stderr, flush=True)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _supervised(cmd, env=None, retries=1):
    """Runs `cmd` (the measuring process) and, if it is killed by a signal (a core dump at GPU start-up was seen once in a few hundred launches on the
    shared boxes: same tree, same flags, the next launch ran), runs it again -- at most `retries` times, saying so on stderr.  A Python exception or a
    failed assertion (ordinary non-zero exit) is never retried.  The child measures itself; nothing about the numbers changes."""
    import signal
    import subprocess
    rc = 0
    for attempt in range(retries + 1):
        rc = subprocess.run(cmd, env=env).returncode
        killed = rc < 0 or rc in (128 + signal.SIGSEGV, 128 + signal.SIGABRT, 128 + signal.SIGBUS)
        if not killed or attempt == retries:
            break
        print(f'[bench.py] measuring process died with {"signal " + str(-rc) if rc < 0 else "exit code " + str(rc)}; running it once more', file=sys.stderr, flush=True)
    return rc


def _requested_gpus(argv):
    """--gpus N / --gpus=N from the raw command line (1 when absent): read before argparse so that the launcher decision costs no torch import"""
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith('--gpus='):
            return int(a.split('=', 1)[1])
    return 1


def _self_launch(ngpus):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (reference: `accelerate launch train.py`, README.md:344-358 -- the user
    never types the per-rank command): start the N ranks ourselves, exactly as the driver's own command line does -- `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` -- one rank per GPU over RCCL; rank 0
    prints the ONE JSON line on the stdout this process passes through."""
    import socket
    import subprocess
    with socket.socket() as s:                                  # a port that is free NOW (the rendezvous store binds it a moment later)
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))   # dmabuf IPC: RCCL needs it on these hosts
    env.pop('ALM_BENCH_CHILD', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ngpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    print(f'[bench.py] --gpus {ngpus} without WORLD_SIZE: launching {ngpus} ranks myself: {" ".join(cmd[1:8])} bench.py ...', file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


if __name__ == '__main__':
    import faulthandler
    faulthandler.enable()                 # a SIGSEGV / SIGABRT inside a native call prints the Python stack of every thread: WHICH launch died
    # single-process runs are measured in a child process that is re-launched once if it is killed by a signal; under torch.distributed.run
    # (one rank per GPU, rendezvous owned by the launcher) every rank measures in place; `--gpus N` typed without a launcher starts one itself
    _world_env = int(os.environ.get('WORLD_SIZE', '1'))
    _want = _requested_gpus(sys.argv[1:])
    if _want > 1 and _world_env == 1 and 'RANK' not in os.environ:
        rc = _self_launch(_want)
        sys.exit(rc if rc >= 0 else 128 - rc)
    if os.environ.get('ALM_BENCH_CHILD') == '1' or _world_env > 1 or os.environ.get('ALM_BENCH_SUPERVISE', '1') == '0':
        main()
    else:
        rc = _supervised([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=dict(os.environ, ALM_BENCH_CHILD='1'))
        sys.exit(rc if rc >= 0 else 128 - rc)
