"""Bench-only: where do the cycles of a token go in hc_bwd's LDS-DMA variant?  Builds csrc/hyper.hip with -DALM_HC_CYC (s_memtime stamps between the phases of
a token, per wave) next to the shipped objects and prints mean cycles per token and phase at the headline shape.
usage: git apply scripts/experiments/hc_bwd_cycle_probe_and_six_dma.patch (the stamps are not in the shipped source) ; scripts/build_variant.sh cyc hyper.hip
-DALM_HC_CYC=1  (CPU)  ;  timeout 120 python scripts/hc_cyc_probe.py  (GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'scripts', 'ubench', 'bin', 'libaudiolm_hip_cyc.so')
os.environ['ALM_LIB_PATH'] = LIB
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

SEG = ['DMA issue (addresses + 6 global_load_lds)', 'read-back, LayerNorm 1st half, dots', 'butterfly + cross-wave combine (barrier)', 'post-reduction scalars (LN 2nd half, tanh, dpre)',
       'element loop + dR stores', 'depth tail (dy, <dR,y> reduce + barrier, dbeta)']
dev, BF16, F32 = torch.device('cuda'), torch.bfloat16, torch.float32


def rnd(*shape, scale=1.0, dtype=F32):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


B, S, N, D = 8, 4, 2048, 1024
M = B * N
hc = dict(gamma=0.1 * rnd(D), Wa=0.05 * rnd(D, S + 1), sa=torch.tensor(0.1, device=dev), Aa=torch.cat([torch.zeros(S, 1, device=dev), torch.eye(S, device=dev)], 1),
          wb=0.05 * rnd(D), sb=torch.tensor(0.1, device=dev), Bb=torch.ones(S, device=dev))
lng = 1 + 0.1 * rnd(D)
R = rnd(B, S, N, D, dtype=BF16)
y = rnd(M, D, dtype=BF16)
w0 = ops.hc_fwd(R, B, S, N, D, hc=hc, ln_gamma=lng, r_dtype=BF16)
w1 = ops.hc_fwd(R, B, S, N, D, y_prev=y, coef_prev=w0['coef'], hc=hc, ln_gamma=lng, r_dtype=BF16)
dRn, dxn, extra, dbeta = rnd(B, S, N, D, dtype=BF16), rnd(M, D, dtype=BF16), rnd(M, D, dtype=BF16), rnd(M, S)
for _ in range(5):
    ops.hc_bwd(dRn, B, S, N, D, dxn=dxn, extra=extra, mean=w1['mean'], rstd=w1['rstd'], ln_gamma=lng, R=w1['R'], coef=w1['coef'], dbeta=dbeta, hc=hc, y_prev=y,
               coef_prev=w0['coef'], r_dtype=BF16)
torch.cuda.synchronize()
lib = ctypes.CDLL(LIB)
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
assert lib.alm_hc_cyc_read(buf, n) == 0
a = np.array(buf, dtype=np.float64).reshape(4096, 8)
a = a[a[:, 6] > 0]
tok = a[:, 6].sum()
print(f'hc_bwd (LDS-DMA variant): {len(a)} waves, {tok / 4:.0f} tokens; mean cycles per token and wave by phase:')
tot = 0.0
for i, name in enumerate(SEG + ['counted wait for the current buffer (vmcnt)']):
    i = 7 if i == 6 else i
    c = a[:, i].sum() / tok
    tot += c
    print(f'    {name:52s} {c:8.0f}')
print(f'    {"sum":52s} {tot:8.0f}   (per-wave total / tokens: min {np.min((a[:, :6].sum(1) + a[:, 7]) / a[:, 6]):.0f}  max {np.max((a[:, :6].sum(1) + a[:, 7]) / a[:, 6]):.0f})')
