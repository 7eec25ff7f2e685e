"""A/B of the product library's GEMM tile ids on the model shapes (round-robin on the same buffers).  usage: python scripts/ab_tiles.py [tile ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

dev, BF16 = torch.device('cuda'), torch.bfloat16
tiles = [int(a) for a in sys.argv[1:]] or [13, 14]
T = 16384
if os.environ.get('ALM_AB_SMALL'):
    shapes_small = [('Wq fwd / dAO', T, 512, 1024), ('Wo fwd / dXN', T, 1024, 512), ('Wkv fwd', T, 128, 1024), ('dKV.WkvT', T, 1024, 128), ('N=768', T, 768, 1024)]
shapes = [('W1 fwd', T, 5472, 1024), ('W2 fwd', T, 1024, 2736), ('dHN dgrad', T, 2736, 1024), ('dXN2 dgrad', T, 1024, 5472), ('Wo fwd', T, 1024, 512), ('square 8192', 8192, 8192, 8192)]
if os.environ.get('ALM_AB_SMALL'):
    shapes = shapes_small
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
    for rnd in range(6):
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
    print(f'{name:12s} ' + ' | '.join(f'tile {t}: {v * 1e3:7.1f} us {fl / v / 1e9:6.0f} TF' for t, v in best.items()), flush=True)
