"""Cost-model measurements for the in-launch split-K NT launches (round 6, csrc/gemm.hip nt_plan): for every under-filled NT shape of the timed steps
(M = 8192: configs[1]; M = 16384: the headline; the long-sequence configurations) the time of
   * the workspace-free choice of alm_gemm_bf16_nt (the 128 x 128 tile / its DMA-ring form),
   * the staggered 256 x 256 tile without a split (slices 1) and with 2 / 3 / 4 / 8 in-launch K slices,
interleaved on ONE box, every launch on a DIFFERENT operand / output buffer set (ring of `NSETS`: nothing is warm in the L2 from the previous launch of the
same variant -- inside the step the operands come from the previous kernel, not from a repeat), plus what alm_gemm_nt_plan picks.
usage: python scripts/ab_nt_inl.py [M ...]          -> one line per shape and variant; `fit` lines give least-squares constants of the model
    t(256^2, S) = ceil(ksteps / S) * us_kstep + us_fixed + (S > 1) * (us_publish + us_reduce * slabs read)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import _lib, ops  # noqa: E402

dev, BF16 = torch.device('cuda'), torch.bfloat16
Ms = [int(a) for a in sys.argv[1:]] or [8192, 16384]
NSETS = 6
SHAPES = [('to_q / dAO', 512, 1024), ('to_out', 1024, 512), ('W2 fwd', 1024, 2736), ('dXN_ff = dU W1', 1024, 5472), ('to_kv-like N=256', 256, 1024), ('head 1025', 1025, 1024)]


def timed(fn, sets, reps=4, inner=2):
    best = 1e9
    for _ in range(reps):
        for s in sets:
            fn(*s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            for s in sets:
                fn(*s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (inner * len(sets)))
    return best * 1e3          # us


rows = []
for M in Ms:
    for name, N, K in SHAPES:
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        if t256 >= 192:
            continue
        sets = [(torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16), torch.empty(M, N, dtype=BF16, device=dev)) for _ in range(NSETS)]
        ksteps = (K + 63) // 64
        res = {}
        res['nt(no ws)'] = timed(lambda A, B, C: ops.gemm_nt_tile(A, B, C, 0), sets)
        ref = sets[0][2].clone()
        res['128ring'] = timed(lambda A, B, C: ops.gemm_nt_tile(A, B, C, 16), sets)
        res['256x128'] = timed(lambda A, B, C: ops.gemm_nt_tile(A, B, C, 15), sets)
        for S in (1, 2, 3, 4, 8):
            if t256 * S > 256 or (S > 1 and (ksteps + S - 1) // S < 4):
                continue
            res[f'S={S}'] = timed(lambda A, B, C, S=S: ops.gemm_nt_inl(A, B, C, S), sets)
            if not torch.allclose(sets[0][2].float(), ref.float(), rtol=2e-2, atol=2e-2 * float(ref.float().abs().max())):
                print(f'  MISMATCH {name} M={M} S={S}')
            rows.append((t256, ksteps, S, res[f'S={S}']))
        plan = (ctypes.c_int * 4)()
        _lib.query('alm_gemm_nt_plan', M, N, K, 1, 1, ctypes.cast(plan, ctypes.c_void_p))
        res['auto(ws)'] = timed(lambda A, B, C: ops.gemm_nt(A, B, C), sets)
        fl = 2.0 * M * N * K
        best = min(res, key=res.get)
        print(f'M={M:6d} {name:18s} N={N:5d} K={K:5d} tiles256={t256:4d} | ' + ' | '.join(f'{k}: {v:6.1f} us {fl / v / 1e6:5.0f} TF' for k, v in res.items())
              + f' | plan: tile {plan[0]} S={plan[1]} | best: {best}', flush=True)
        del sets

# least-squares fit of the model constants over the measured split variants
if rows:
    import numpy as np
    A_ = np.array([[-(-ks // S), 1.0, 1.0 if S > 1 else 0.0, (0 if S == 1 else (1 if S == 2 else S))] for _, ks, S, _ in rows], dtype=float)
    y = np.array([t for *_, t in rows])
    coef, *_ = np.linalg.lstsq(A_, y, rcond=None)
    print('fit: us_kstep %.3f us_fixed %.2f us_publish %.2f us_reduce %.2f  (rms residual %.2f us over %d points)'
          % (*coef, float(np.sqrt(np.mean((A_ @ coef - y) ** 2))), len(rows)))
