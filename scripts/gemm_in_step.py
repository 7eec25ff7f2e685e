"""Per-shape GEMM time INSIDE the headline training step (side stream off, HIP events around every launch), next to the same shape timed
back-to-back stand-alone: shows which launches lose time to their neighbours (cold operands, clock, launch gaps) rather than to the kernel.
With --vendor a second pass substitutes torch.mm (hipBLASLt / rocBLAS) for every plain 2-D NT launch (no bias, alpha 1, no accumulation) INSIDE the
same step -- a yardstick only: is the vendor kernel faster than ours in the step, or does it lose the same 10-15 % to the step's conditions?
Usage: python scripts/gemm_in_step.py [--vendor]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd as A  # noqa: E402
import audiolm_pytorch_amd.core as core_mod  # noqa: E402
from audiolm_pytorch_amd import ops  # noqa: E402
from bench import build  # noqa: E402

dev = torch.device('cuda')


def fl_of(nb, M, N, K):
    return 2.0 * nb * M * N * K


def main():
    W = build('coarse2048', dev, 0, torch.bfloat16)
    model, wrapper = W['model'], W['wrapper']
    sem, coarse = W['inputs']['semantic_token_ids'], W['inputs']['coarse_token_ids']
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
    vend = {}
    if '--vendor' in sys.argv:
        vev = []

        def vendor_nt(Am, Bm, Cm, **kw):
            plain = Am.dim() == 2 and kw.get('bias') is None and kw.get('alpha', 1.0) == 1.0 and not kw.get('accumulate', False) and Cm.is_contiguous() and Am.is_contiguous() and Bm.is_contiguous()
            if not plain:
                return orig_nt(Am, Bm, Cm, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if Cm.dtype == torch.bfloat16:
                torch.mm(Am, Bm.t(), out=Cm)
            else:
                Cm.copy_(torch.mm(Am, Bm.t()))                               # fp32 result: the vendor path pays a cast (shown, not hidden)
            e1.record()
            vev.append((e0, e1, ('nt', 1, Am.shape[0], Bm.shape[0], Am.shape[1], str(Cm.dtype)[6:], '')))
            return Cm
        ops.gemm_nt = vendor_nt
        for _ in range(3):
            step()
        vev.clear()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        ops.gemm_nt = orig_nt
        for e0, e1, key in vev:
            a = vend.setdefault(key, [0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
    agg = {}
    for e0, e1, key in events:
        a = agg.setdefault(key, [0.0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += 1
    print(f'{"kind nb M N K out":44s} launches/step   in-step us   stand-alone us   TF in-step   vendor in-step us   vendor stand-alone us')
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
        vtxt = ''
        if key in vend:
            vus = vend[key][0] / vend[key][1] * 1e3
            Cv = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            for _ in range(3):
                torch.mm(Am, Bm.t(), out=Cv)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                torch.mm(Am, Bm.t(), out=Cv)
            e1.record()
            torch.cuda.synchronize()
            vtxt = f'      {vus:8.1f} ({fl_of(nb, M, N, K) / vus / 1e6:5.0f} TF)   {e0.elapsed_time(e1) / 20 * 1e3:8.1f}'
        fl = 2.0 * nb * M * N * K
        print(f'{kind} nb={nb:<2d} {M:6d} x {N:5d} x {K:6d} {dt:8s} {bias:4s}   {n / reps:5.1f}        {us:8.1f}      {sa:8.1f}        {fl / us / 1e6:7.0f}{vtxt}')
    print(f'total GEMM time per step {tot:.3f} ms')


if __name__ == '__main__':
    main()
