#!/bin/bash
tag=${1:-r5s}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x -k "gemm_nt_tile_configs and 16" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_tests.log | cut -c1-300
# race screen: the ring kernel many times on the to_kv shape against the two-stage kernel (bit-identical tiles expected)
python - <<'PY' 2>&1 | tail -n 6
import torch, sys
sys.path.insert(0, '.')
import audiolm_pytorch_amd
from audiolm_pytorch_amd import ops
torch.manual_seed(0)
bad = 0
for M, N, K in ((16384, 128, 1024), (4096, 512, 1024), (16384, 128, 192), (2048, 128, 4096)):
    A = torch.randn(M, K, device='cuda').bfloat16(); B = torch.randn(N, K, device='cuda').bfloat16()
    ref = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); ops.gemm_nt_tile(A, B, ref, 1)
    for it in range(200):
        C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        ops.gemm_nt_tile(A, B, C, 16)
        if not torch.equal(C, ref): bad += 1
    import time
    for tile in (1, 16):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.gemm_nt_tile(A, B, C, tile)
        e1.record(); torch.cuda.synchronize()
        print(M, N, K, 'tile', tile, round(e0.elapsed_time(e1) / 50 * 1e3, 1), 'us')
print('mismatching runs:', bad)
PY
STEPS=30 bash scripts/ab_env2.sh 3 "ALM_GEMM_RING=1" "ALM_GEMM_RING=0" > gpurun_out/${tag}_ab.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab.log | cut -c1-200
