"""Per-shape GEMM time INSIDE the headline training step (side stream off, HIP events around every launch), next to the same shape timed
back-to-back stand-alone: shows which launches lose time to their neighbours (cold operands, clock, launch gaps) rather than to the kernel.
Usage: python scripts/gemm_in_step.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd as A  # noqa: E402
import audiolm_pytorch_amd.core as core_mod  # noqa: E402
from audiolm_pytorch_amd import ops  # noqa: E402
from bench import B_PER_GPU, MODEL, N_FRAMES, N_SEM, Codec  # noqa: E402

dev = torch.device('cuda')


def main():
    torch.manual_seed(0)
    model = A.CoarseTransformer(**MODEL).to(dev)
    wrapper = A.CoarseTransformerWrapper(transformer=model, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
    wrapper.train()
    g = torch.Generator().manual_seed(1000)
    sem = torch.randint(0, 500, (B_PER_GPU, N_SEM), generator=g).to(dev)
    coarse = torch.randint(0, 1024, (B_PER_GPU, N_FRAMES, 3), generator=g).to(dev)
    cache = model.transformer._cache

    def step():
        cache.store.clear()
        for p in model.parameters():
            p.grad = None
        wrapper(semantic_token_ids=sem, coarse_token_ids=coarse, return_loss=True).backward()

    core_mod.ASYNC_WGRAD = False
    for _ in range(12):
        step()
    events = []
    orig_nt, orig_tn = ops.gemm_nt, ops.gemm_tn_splitk

    def timed_nt(Am, Bm, Cm, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_nt(Am, Bm, Cm, **kw)
        e1.record()
        nb = 1
        for d in Am.shape[:-2]:
            nb *= d
        events.append((e0, e1, ('nt', nb, Am.shape[-2], Bm.shape[-2], Am.shape[-1], str(Cm.dtype)[6:], 'bias' if kw.get('bias') is not None else '')))
        return out

    def timed_tn(At, Bt, Cm, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_tn(At, Bt, Cm, **kw)
        e1.record()
        nb = At.shape[0] if At.dim() == 3 else 1
        events.append((e0, e1, ('tn', nb, At.shape[-1], Bt.shape[-1], At.shape[-2], str(Cm.dtype)[6:], '')))
        return out
    ops.gemm_nt, ops.gemm_tn_splitk = timed_nt, timed_tn
    reps = 5
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ops.gemm_nt, ops.gemm_tn_splitk = orig_nt, orig_tn
    agg = {}
    for e0, e1, key in events:
        a = agg.setdefault(key, [0.0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += 1
    print(f'{"kind nb M N K out":44s} launches/step   in-step us   stand-alone us   TF in-step')
    tot = 0.0
    for key, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        kind, nb, M, N, K, dt, bias = key
        us = ms / n * 1e3
        tot += ms / reps
        # stand-alone: same shape, fresh random operands, back-to-back
        sa = float('nan')
        if kind == 'nt' and nb == 1:
            Am = torch.randn(M, K, device=dev).to(torch.bfloat16)
            Bm = torch.randn(N, K, device=dev).to(torch.bfloat16)
            Cm = torch.empty(M, N, dtype=torch.float32 if dt == 'float32' else torch.bfloat16, device=dev)
            for _ in range(3):
                orig_nt(Am, Bm, Cm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                orig_nt(Am, Bm, Cm)
            e1.record()
            torch.cuda.synchronize()
            sa = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * nb * M * N * K
        print(f'{kind} nb={nb:<2d} {M:6d} x {N:5d} x {K:6d} {dt:8s} {bias:4s}   {n / reps:5.1f}        {us:8.1f}      {sa:8.1f}        {fl / us / 1e6:7.0f}')
    print(f'total GEMM time per step {tot:.3f} ms')


if __name__ == '__main__':
    main()
