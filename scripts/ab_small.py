"""A/B of tile ids on the SMALL projection shapes of the step (attention Wq / Wkv / Wo and their dgrads): usage: python scripts/ab_small.py [tile ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

dev, BF16 = torch.device('cuda'), torch.bfloat16
tiles = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 13]
T = 16384
shapes = [('Wq fwd / dAO dgrad', T, 512, 1024), ('Wo fwd / dQ dgrad', T, 1024, 512), ('Wkv fwd', T, 128, 1024), ('dKV dgrad', T, 1024, 128), ('coarse head', 5464, 1025, 1024)]
for name, M, N, K in shapes:
    A, B = torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16)
    C = torch.empty(M, N, dtype=BF16, device=dev)
    best = {t: 1e9 for t in tiles}
    ref = None
    for t in tiles:
        C.zero_()
        ops.gemm_nt_tile(A, B, C, t)
        torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        elif not torch.equal(C, ref):
            print(f'  MISMATCH {name} tile {t}: {float((C.float() - ref.float()).abs().max()):.3e}')
    for rnd in range(5):
        for t in tiles:
            for _ in range(3):
                ops.gemm_nt_tile(A, B, C, t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt_tile(A, B, C, t)
            e1.record()
            torch.cuda.synchronize()
            best[t] = min(best[t], e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    print(f'{name:20s} ' + ' | '.join(f'tile {t}: {v * 1e3:6.1f} us {fl / v / 1e9:5.0f} TF' for t, v in best.items()), flush=True)
