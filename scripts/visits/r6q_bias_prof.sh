#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, audiolm_pytorch_amd as A
kind = sys.argv[1]
dev = torch.device('cuda'); g = torch.Generator().manual_seed(1); B = 8
torch.manual_seed(0)
if kind == 'fine':
    m = A.FineTransformer(**dict(bench.FINE, flash_attn=False)).to(dev)
    w = A.FineTransformerWrapper(transformer=m, codec=bench.Codec(), mask_prob=0.15).train()
    grid = torch.randint(0, 1024, (B, 256, 8), generator=g).to(dev)
    inp = dict(coarse_token_ids=grid[..., :3].contiguous(), fine_token_ids=grid[..., 3:].contiguous())
else:
    m = A.CoarseTransformer(**dict(bench.COARSE, flash_attn=False)).to(dev)
    w = A.CoarseTransformerWrapper(transformer=m, codec=bench.Codec(), unique_consecutive=False, mask_prob=0.15).train()
    inp = dict(semantic_token_ids=torch.randint(0, 500, (B, 509), generator=g).to(dev), coarse_token_ids=torch.randint(0, 1024, (B, 512, 3), generator=g).to(dev))
ps = [p for p in w.parameters()]
for _ in range(12):
    for p in ps: p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = w(**inp, return_loss=True)
    loss.backward()
torch.cuda.synchronize()
PY
for k in fine coarse; do
  rm -rf /tmp/prof_$k
  ALM_ASYNC_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$k -o p -- python /tmp/one.py $k > gpurun_out/r6q_prof_$k.log 2>&1
  db=$(find /tmp/prof_$k -name "*.db" | head -1)
  python scripts/prof_summary.py "$db" gpurun_out/r6q_bias_kernels_$k.csv "$k flash_attn=False (reference default), B=8, 12 steps, ALM_ASYNC_WGRAD=0" | tail -1
done
