"""rocprofv3 target: biased attention fwd + bwd (Toeplitz table, B=8 N=2048 H=8), 20 iterations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiolm_pytorch_amd as A  # noqa: E402,F401
from audiolm_pytorch_amd import ops, relpos  # noqa: E402

dev = torch.device('cuda:0')
B, N, H, dh = 8, 2048, 8, 64
kind = sys.argv[1] if len(sys.argv) > 1 else 'toeplitz'
M = B * N
Q, KV = torch.randn(M, H * dh, device=dev).bfloat16(), torch.randn(M, 2 * dh, device=dev).bfloat16()
K_, V_ = KV[:, :dh], KV[:, dh:]
dAO = torch.randn(M, H * dh, device=dev).bfloat16()
if kind == 'fine':
    grid, index = relpos.fine_index(765, 1281, 3, 5, dev)
    LT = grid.shape[0] + 1
else:
    index, LT = relpos.toeplitz_index(N, dev, num_leading=1024 if kind == 'coarse' else None), 2 * N
bias = relpos.AttnBias(torch.randn(H, LT, device=dev), *index)
part = ops.attn_bias_part(B, N, H, LT, dev)
for _ in range(20):
    AO, LSE = ops.mqa_attn_fwd(Q, K_, V_, None, B, N, H, dh, bias=bias)
    ops.mqa_attn_bwd(Q, K_, V_, None, AO, LSE, dAO, B, N, H, dh, bias=bias, dtbl_part=part)
    AO0, LSE0 = ops.mqa_attn_fwd(Q, K_, V_, None, B, N, H, dh)
    ops.mqa_attn_bwd(Q, K_, V_, None, AO0, LSE0, dAO, B, N, H, dh)
torch.cuda.synchronize()
