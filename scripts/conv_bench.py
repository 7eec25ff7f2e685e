"""SoundStream encoder launches at the configs[4] shape (8 x 30 s @ 24 kHz): per-stage time of the ResidualUnits fused (alm_resunit_causal) vs as two
alm_conv1d_causal launches, and of the whole tokenize.   usage: python scripts/conv_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd as A  # noqa: E402
from audiolm_pytorch_amd import ops  # noqa: E402

dev = torch.device('cuda')


def timed(fn, iters=5):
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
    return best


B = 8
for C, T in ((32, 720000), (64, 360000), (128, 90000), (256, 18000)):
    x = torch.randn(B, C, T, device=dev)
    w7p = ops.conv1d_pack(torch.randn(C, C, 7, device=dev) * (7 * C) ** -0.5)
    w1p = ops.conv1d_pack(torch.randn(C, C, 1, device=dev) * C ** -0.5)
    b7, b1 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    for dil in (1, 9):
        def two():
            h = ops.conv1d_causal(x, w7p, b7, C, 7, dilation=dil, elu=True)
            return ops.conv1d_causal(h, w1p, b1, C, 1, elu=True, residual=x)
        t2 = timed(two)
        t1 = timed(lambda: ops.resunit_causal(x, w7p, b7, w1p, b1, 7, dil))
        fl = 2.0 * B * T * C * C * 8
        print(f'C={C:4d} T={T:7d} dil={dil}: two launches {t2 * 1e3:8.1f} us ({fl / t2 / 1e9:5.1f} TF)   fused {t1 * 1e3:8.1f} us ({fl / t1 / 1e9:5.1f} TF)', flush=True)
    del x

ss = A.SoundStream(codebook_size=4096, rq_num_quantizers=8, target_sample_hz=24000, strides=(2, 4, 5, 8), use_local_attn=False).to(dev)
with torch.no_grad():
    for r in ss.rq.rvqs:
        for q, l in enumerate(r.layers):
            l._codebook.embed.copy_(torch.randn(1, 4096, 512) * (0.5 ** q))
            l._codebook.initted.fill_(True)
wave = torch.randn(B, 720000, device=dev) * 0.1
with torch.no_grad():
    print(f'tokenize 8 x 30 s: {timed(lambda: ss.tokenize(wave), iters=3):.2f} ms  [ALM_FUSE_RESUNIT={os.environ.get("ALM_FUSE_RESUNIT", "1")}]')
