"""Times the flash MQA attention launches at the timed shapes (B=8, H=8, dh=64, key mask on: the training call): N = 2048 (headline), 1024 (configs[1]),
2049 (configs[2]), 8253 (configs[4]), 16385 (fine_t2048_q8).   usage: python scripts/attn_bench.py [N ...]
A/B of two builds on ONE box:  ALM_LIB_PATH=/path/to/other/libaudiolm_hip.so python scripts/attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


def run(N):
    B, H, d = 8, 8, 64
    q = torch.randn(B * N, H * d, device=dev).to(BF16)
    kv = torch.randn(B * N, 2 * d, device=dev).to(BF16)
    k, v = kv[:, :d], kv[:, d:]
    mask = (torch.rand(B, N, device=dev) > 0.15).to(torch.uint8)
    mask[:, 0] = 1
    do = torch.randn(B * N, H * d, device=dev).to(BF16)
    o, lse = ops.mqa_attn_fwd(q, k, v, mask, B, N, H, d)
    it = 30 if N <= 4096 else 5
    tf = timeit(lambda: ops.mqa_attn_fwd(q, k, v, mask, B, N, H, d), iters=it, warm=2)
    tb = timeit(lambda: ops.mqa_attn_bwd(q, k, v, mask, o, lse, do, B, N, H, d), iters=it, warm=2)
    fl = 4.0 * B * H * N * (N + 1) * d / 2
    print(f'{os.environ.get("ALM_LIB_PATH", "default")} B={B} N={N}: fwd {tf:7.1f} us ({fl / tf / 1e6:5.0f} TF)   bwd (delta + dQ + dK/dV) {tb:7.1f} us ({2.5 * fl / tb / 1e6:5.0f} TF)', flush=True)


if __name__ == '__main__':
    for N_ in [int(a) for a in sys.argv[1:]] or [1024, 2048, 2049, 8253, 16385]:
        run(N_)
