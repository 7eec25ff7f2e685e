"""Times the two production hyper-connection launches at the headline shape (B=8, N=2048, D=1024, S=4, bf16 stream storage: the fused
[depth k + width k+1] forward and the fused [pre-LayerNorm backward + width k+1 + depth k] backward).  A/B of two builds on ONE box:
    ALM_LIB_PATH=/path/to/other/libaudiolm_hip.so python scripts/hc_bench.py     (the default library otherwise)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

dev = torch.device('cuda')
BF16, F32 = torch.bfloat16, torch.float32


def rnd(*shape, scale=1.0, dtype=F32):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


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


def main():
    B, S, N, D = 8, 4, 2048, 1024
    M = B * N
    hc = dict(gamma=0.1 * rnd(D), Wa=0.05 * rnd(D, S + 1), sa=torch.tensor(0.1, device=dev), Aa=torch.cat([torch.zeros(S, 1, device=dev), torch.eye(S, device=dev)], 1),
              wb=0.05 * rnd(D), sb=torch.tensor(0.1, device=dev), Bb=torch.ones(S, device=dev))
    lng = 1 + 0.1 * rnd(D)
    R = rnd(B, S, N, D, dtype=BF16)
    y = rnd(M, D, dtype=BF16)
    w0 = ops.hc_fwd(R, B, S, N, D, hc=hc, ln_gamma=lng, r_dtype=BF16)                                   # width only: produces a coefficient record
    f = lambda: ops.hc_fwd(R, B, S, N, D, y_prev=y, coef_prev=w0['coef'], hc=hc, ln_gamma=lng, r_dtype=BF16)
    w1 = f()
    t_f = timeit(f)
    dRn = rnd(B, S, N, D, dtype=BF16)
    dxn = rnd(M, D, dtype=BF16)
    extra = rnd(M, D, dtype=BF16)
    dbeta = rnd(M, S)
    g = lambda: ops.hc_bwd(dRn, B, S, N, D, dxn=dxn, extra=extra, mean=w1['mean'], rstd=w1['rstd'], ln_gamma=lng, R=w1['R'], coef=w1['coef'], dbeta=dbeta,
                           hc=hc, y_prev=y, coef_prev=w0['coef'], r_dtype=BF16)
    g()
    t_b = timeit(g)
    print(f'{os.environ.get("ALM_LIB_PATH", "default")}: hc_fwd fused {t_f:7.1f} us   hc_bwd fused {t_b:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
