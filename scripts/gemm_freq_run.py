"""helper of scripts/gemm_freq_pmc.sh: 6 launches of the 8192^3 NT GEMM (tile 2) per library, production first, then scripts/ubench/bin/libgemm_v*.so"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import _lib  # noqa: E402

dev = torch.device('cuda')
libs = [('prod', _lib.load())]
for p in sorted(glob.glob(os.path.join(ROOT, 'scripts/ubench/bin/libgemm_v*.so'))):
    lib = ctypes.CDLL(p)
    lib.alm_gemm_bf16_nt_tile.argtypes = _lib.SIGNATURES['alm_gemm_bf16_nt_tile']
    lib.alm_gemm_bf16_nt_tile.restype = ctypes.c_int
    libs.append((os.path.basename(p)[3:-3], lib))
M = N = K = 8192
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = torch.randn(N, K, device=dev).to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for name, lib in libs:
    for _ in range(6):
        lib.alm_gemm_bf16_nt_tile(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, K, K, N, 1.0, 0, 0, 2, st)
    torch.cuda.synchronize()
print('order:', [n for n, _ in libs])
