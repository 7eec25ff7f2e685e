"""dXN_q || dX_kv (and to_q || to_kv) as ONE grouped launch: per-launch time at the timed batch shapes; run once per ALM_GEMM_GROUP2_BIG / ALM_GEMM_GROUP2 setting
(the switches are read once per process).  usage: python scripts/ab_group2.py [M ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

dev, BF16 = torch.device('cuda'), torch.bfloat16
for M in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    for name, (N0, K0), (N1, K1) in (('dXN_q || dX_kv', (1024, 512), (1024, 128)), ('to_q || to_kv', (512, 1024), (128, 1024))):
        sets = [tuple(torch.randn(*sh, device=dev).to(BF16) for sh in ((M, K0), (N0, K0), (M, N0), (M, K1), (N1, K1), (M, N1))) for _ in range(6)]
        best = 1e9
        for _ in range(5):
            for s in sets:
                ops.gemm_nt_group2(*s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for s in sets:
                    ops.gemm_nt_group2(*s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 18)
        fl = 2.0 * M * (N0 * K0 + N1 * K1)
        print(f'M={M} {name}: {best * 1e3:.1f} us {fl / best / 1e9:.0f} TF  [GROUP2_BIG={os.environ.get("ALM_GEMM_GROUP2_BIG", "1")} GROUP2={os.environ.get("ALM_GEMM_GROUP2", "1")}]', flush=True)
