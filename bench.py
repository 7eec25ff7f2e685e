"""Headline benchmark: audio-tokens/sec, forward + backward of CoarseTransformer (dim 1024, depth 6, 3 coarse quantizers, 4 residual
streams, flash_attn=True) over synthetic SoundStream-RVQ token sequences with transformer length N = 2048 (BASELINE.json metric,
configs[1]/[3]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; per-rank batch B = 8 (weak scaling); a "step" = CoarseTransformerWrapper.forward(return_loss=True) +
backward (+ RCCL gradient all-reduce over xGMI for N > 1), inputs resident in HBM, the bf16 weight copies are re-packed from
the fp32 masters every step (as after an optimiser step).  The optimiser itself is outside the metric ("fwd+bwd"); a second
timed loop that includes torch's Adam step is reported as `with_optimizer` for reference.

Extra objects on the JSON line:
  roofline     dominant kernel = the bf16 MFMA GEMM (gemm_nt_kernel); achieved = algorithmic GEMM FLOPs per launch / average launch
               duration, measured live with HIP events on the launch stream over one instrumented step; peak = 2500 TFLOP/s dense bf16.
  cpu_baseline the oracle (CPU fp32 restatement of the reference, "port") timed on this box's host cores on a bounded sample
               (B = 1, N = 2048, same architecture), rank 0 at N = 1 only.
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

MODEL = dict(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True)
B_PER_GPU, N_SEM, N_FRAMES = 8, 509, 512          # N = 1 + (509 + 1) + 1 + 512 * 3 = 2048
SEQ = 1 + (N_SEM + 1) + 1 + N_FRAMES * 3
PEAK_BF16_TFLOPS = 2500.0                          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


class Codec:
    rq_groups = 1
    num_quantizers = 8


def cpu_baseline(max_seconds=30.0):
    """Times the CPU oracle (fp32) on a bounded sample of the same workload: B = 1, N = 2048, fwd + bwd."""
    import audiolm_oracle as O
    import audiolm_pytorch_amd as A
    cores = min(os.cpu_count() or 1, 32)          # one GPU's share of the host (256 cores / 8 GPUs); more threads oversubscribe torch's CPU ops
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = A.CoarseTransformer(**MODEL)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith('.beta')}
    full = dict(sd)
    full.update(params)
    cfg = O.Cfg(dim=1024, depth=6, streams=4, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3)
    g = torch.Generator().manual_seed(0)
    sem = torch.randint(0, 500, (1, N_SEM), generator=g)
    coarse = torch.randint(0, 1024, (1, N_FRAMES, 3), generator=g)
    mask = O.generate_mask_with_prob((1, SEQ), 0.15, 'cpu', generator=g)
    times = []
    t_start = time.time()
    for it in range(3):
        t0 = time.time()
        loss = O.coarse_wrapper_loss(full, cfg, sem, coarse, training=True, unique_consecutive=False, forgetful_mask=mask)
        loss.backward()
        times.append(time.time() - t0)
        for p in params.values():
            p.grad = None
        if time.time() - t_start > max_seconds:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return dict(value=round(SEQ / best, 1), unit='audio-tokens/s', cores=cores, kind='port',
                sample=f'oracle (CPU fp32 restatement of the reference, 4 residual streams) fwd+bwd, B=1 x N={SEQ}, best of {max(1, len(times) - 1)} after 1 warm-up')


PRIME_STEPS = 10


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r1_pmc_summary.json: FETCH_SIZE x 2
    per the MI355X guide's gfx950 correction + WRITE_SIZE), or None when the summary is absent.  Counters cannot be collected inside this
    process; scripts/pmc.sh regenerates them."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r1_pmc_summary.json')) as fh:
            return json.load(fh)['hbm_bytes_per_launch']
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-optimizer-leg', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
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
        import torch.distributed as dist
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)      # backend "nccl" IS RCCL on ROCm
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'

    import audiolm_pytorch_amd as A
    from audiolm_pytorch_amd import ops, parallel

    torch.manual_seed(0)                                        # identical initial weights on every rank
    model = A.CoarseTransformer(**MODEL).to(dev)
    wrapper = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
    wrapper.train()
    engine = parallel.DataParallelEngine(model, dist) if world > 1 else None

    g = torch.Generator().manual_seed(1000 + rank)              # a different synthetic shard per rank
    sem = torch.randint(0, 500, (B_PER_GPU, N_SEM), generator=g).to(dev)
    coarse = torch.randint(0, 1024, (B_PER_GPU, N_FRAMES, 3), generator=g).to(dev)
    cache = model.transformer._cache

    def step(opt=None):
        cache.store.clear()                                     # weights "changed": re-pack bf16 copies like after an optimiser step
        for p in model.parameters():
            p.grad = None
        loss = wrapper(semantic_token_ids=sem, coarse_token_ids=coarse, return_loss=True)
        loss.backward()
        if engine is not None:
            engine.finish()                                     # waits for the overlapped RCCL all-reduces, grads averaged in place
        if opt is not None:
            opt.step()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, opt=None):
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(opt)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    # untimed priming in addition to --warmup: the first steps grow the caching allocator's pools (main + side stream) and run at ramping
    # clocks; measured on MI355X the step time only settles after ~10 steps (17.6 -> 16.4 ms).  The timed region below is exactly --steps steps.
    for _ in range(PRIME_STEPS):
        loss = step()
    for _ in range(args.warmup):
        loss = step()
    dt = timed(args.steps)
    ms = dt / args.steps * 1e3
    tokens_per_s = world * B_PER_GPU * SEQ / (dt / args.steps)

    # ---- roofline of the dominant kernel: one instrumented step, HIP events (torch.cuda.Event on the launch stream = torch's current
    # stream, which is the stream every alm_* launch uses) around every MFMA GEMM launch.  The dominant kernel is the NT 256x256x64
    # 8-wave tile (gemm_kernel<256,256,2,4,false,*>): its launches are singled out; all GEMM launches are reported alongside.
    # EVERY rank runs the instrumented step (with N > 1 it contains the gradient all-reduces: a step on rank 0 alone would dead-lock the
    # collectives); only rank 0's numbers are reported.
    roof = None
    if True:
        events = []
        orig_nt, orig_tn = ops.gemm_nt, ops.gemm_tn_splitk

        def big_tile(M_, N_, nb):               # mirrors pick_tile() in csrc/gemm.hip
            return M_ >= 256 and N_ >= 256 and ((M_ + 255) // 256) * ((N_ + 255) // 256) * nb >= 192

        def timed_nt(Am, Bm, Cm, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_nt(Am, Bm, Cm, **kw)
            e1.record()
            nb = 1
            for d in Am.shape[:-2]:
                nb *= d
            Mm, Nn, Kk = Am.shape[-2], Bm.shape[-2], Am.shape[-1]
            events.append((e0, e1, 2.0 * nb * Mm * Nn * Kk, 'nt256' if big_tile(Mm, Nn, nb) else 'nt128'))
            return out

        def timed_tn(At, Bt, Cm, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_tn(At, Bt, Cm, **kw)
            e1.record()
            nb = At.shape[0] if At.dim() == 3 else 1
            events.append((e0, e1, 2.0 * nb * At.shape[-1] * Bt.shape[-1] * At.shape[-2], 'tn'))
            return out
        ops.gemm_nt, ops.gemm_tn_splitk = timed_nt, timed_tn
        import audiolm_pytorch_amd.core as core_mod
        was_async, core_mod.ASYNC_WGRAD = core_mod.ASYNC_WGRAD, False      # per-kernel durations: no concurrent side-stream GEMMs in this step
        try:
            step()
            torch.cuda.synchronize()
        finally:
            ops.gemm_nt, ops.gemm_tn_splitk = orig_nt, orig_tn
            core_mod.ASYNC_WGRAD = was_async
        agg = {}
        for e0, e1, fl, kind in events:
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += fl
            a[2] += 1
        tot_ms = sum(a[0] for a in agg.values())
        tot_fl = sum(a[1] for a in agg.values())
        d_ms, d_fl, d_n = agg.get('nt256', [0.0, 0.0, 0])
        if d_n:
            ach = d_fl / (d_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'gemm_kernel<256,256,2,4,NT> (bf16 MFMA 32x32x16; FFN / projection forward + dgrad GEMMs)',
                    'achieved': round(ach, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
                    'traffic': pmc_traffic(), 'launches_per_step': d_n, 'avg_launch_us': round(d_ms * 1e3 / d_n, 2),
                    'flop_per_launch_avg': round(d_fl / d_n, 0),
                    'share_of_step': round(d_ms / ms, 3),
                    'all_gemm_launches': {'launches_per_step': len(events), 'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                                          'share_of_step': round(tot_ms / ms, 3),
                                          'by_kind_ms': {k: round(v[0], 3) for k, v in agg.items()}},
                    'model_flops_frac': round(tokens_per_s / world * 391e6 / (PEAK_BF16_TFLOPS * 1e12), 4)}
    if dist is not None:
        dist.barrier()

    opt_leg = None
    if not args.no_optimizer_leg:
        # full training step as the reference trainer runs it (trainer.py:1241-1255): fwd + bwd (+ gradient all-reduce) + clip_grad_norm_(0.5)
        # + Adam (optimizer.py:get_optimizer, wd = 0 -> Adam).  Ours: FusedAdam (two HIP launches, norm kept on the device); beside it the
        # same step with torch.optim.Adam + torch.nn.utils.clip_grad_norm_ as the stock baseline.
        nst = max(3, args.steps // 2)

        def opt_step(opt, clip):
            cache.store.clear()
            for p in model.parameters():
                p.grad = None
            wrapper(semantic_token_ids=sem, coarse_token_ids=coarse, return_loss=True).backward()
            if engine is not None:
                engine.finish()
            clip(opt)
            opt.step()

        def timed_opt(opt, clip):
            opt_step(opt, clip)
            barrier()
            t0 = time.perf_counter()
            for _ in range(nst):
                opt_step(opt, clip)
            barrier()
            dt_ = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt_], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t)
            return dt_
        state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        fused = A.get_optimizer(model.parameters(), lr=1e-5, wd=0.)
        dto = timed_opt(fused, lambda o: o.clip_grad_norm_(0.5))
        del fused
        model.load_state_dict(state0)
        stock = torch.optim.Adam(model.parameters(), lr=1e-5, betas=(0.9, 0.99))
        dts = timed_opt(stock, lambda o: torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5))
        opt_leg = dict(ms_per_step=round(dto / nst * 1e3, 3), value=round(world * B_PER_GPU * SEQ / (dto / nst), 1),
                       optimizer='FusedAdam (alm_opt_grad_sumsq + alm_opt_adam_step): clip_grad_norm_(0.5) + Adam',
                       torch_adam_ms_per_step=round(dts / nst * 1e3, 3))

    if rank == 0:
        out = {
            'metric': 'audio-tokens/sec fwd+bwd, CoarseTransformer d=1024 seq=2048',
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
            'data': 'synthetic (uniform random semantic / coarse RVQ token ids, random-init weights)',
            'config': {'workload': 'CoarseTransformer dim=1024 depth=6 heads=8 (MQA) num_coarse_quantizers=3 codebook=1024 semantic=500 '
                                   'residual_streams=4 flash_attn=True; per-GPU B=8, N=2048 (509 semantic + 512x3 coarse ids + eos/start); mask_prob=0.15',
                       'global_batch': world * B_PER_GPU, 'seq_len': SEQ, 'parallelism': f'dp{world}'},
            'loss': round(float(loss), 4),
        }
        if roof:
            out['roofline'] = roof
        if opt_leg:
            out['with_optimizer'] = opt_leg
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
