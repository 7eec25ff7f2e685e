"""tokens/s of the sampling loops (kv-cache decode path) at the benchmark model size: SemanticTransformerWrapper.generate and CoarseTransformerWrapper.generate,
batch 1 and 8.  usage: python scripts/generate_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audiolm_pytorch_amd as A  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
sem = A.SemanticTransformer(dim=1024, depth=6, num_semantic_tokens=500, flash_attn=True).to(dev)
sw = A.SemanticTransformerWrapper(transformer=sem, unique_consecutive=False).eval()
for B in (1, 8):
    for n in (64, 256):
        sw.generate(max_length=8, batch_size=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sw.generate(max_length=n, batch_size=B)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'semantic generate B={B} max_length={n}: {dt * 1e3:.1f} ms = {dt / n * 1e3:.3f} ms/token-step, {B * n / dt:.0f} tokens/s', flush=True)
coarse = A.CoarseTransformer(**bench.COARSE).to(dev)
cw = A.CoarseTransformerWrapper(transformer=coarse, codec=bench.Codec(), unique_consecutive=False).eval()
g = torch.Generator().manual_seed(1)
for B in (1, 8):
    s_ids = torch.randint(0, 500, (B, 250), generator=g).to(dev)
    for steps in (8, 64):
        cw.generate(semantic_token_ids=s_ids, max_time_steps=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cw.generate(semantic_token_ids=s_ids, max_time_steps=steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'coarse generate B={B} time_steps={steps} (x3 quantizers): {dt * 1e3:.1f} ms = {dt / (steps * 3) * 1e3:.3f} ms/token-step, {B * steps * 3 / dt:.0f} tokens/s', flush=True)
