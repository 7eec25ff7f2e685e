"""ms/step of the three wrappers with the REFERENCE'S DEFAULT constructor arguments (flash_attn = False: relative position bias in every attention layer,
audiolm_pytorch.py:502 / :924-936 / :1261-1285; unique_consecutive = True) next to the benchmark's settings, same sizes.  usage: python scripts/defaults_bench.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audiolm_pytorch_amd as A  # noqa: E402

dev = torch.device('cuda')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(wrapper, inputs, tag):
    params = [p for p in wrapper.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = wrapper(**inputs, return_loss=True)
        loss.backward()
        return loss

    for _ in range(6):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    print(f'{tag}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step  loss {float(loss.detach()):.5f}', flush=True)


g = torch.Generator().manual_seed(1)
B = 8
sem = torch.randint(0, 500, (B, 509), generator=g).to(dev)
coarse = torch.randint(0, 1024, (B, 512, 3), generator=g).to(dev)
grid = torch.randint(0, 1024, (B, 256, 8), generator=g).to(dev)
for r in range(2):
    for flash in (True, False):
        torch.manual_seed(0)
        m = A.CoarseTransformer(**dict(bench.COARSE, flash_attn=flash)).to(dev)
        w = A.CoarseTransformerWrapper(transformer=m, codec=bench.Codec(), unique_consecutive=False, mask_prob=0.15).train()
        run(w, dict(semantic_token_ids=sem, coarse_token_ids=coarse), f'round {r} coarse N=2048 flash_attn={flash}')
        del m, w
        torch.manual_seed(0)
        m = A.FineTransformer(**dict(bench.FINE, flash_attn=flash)).to(dev)
        w = A.FineTransformerWrapper(transformer=m, codec=bench.Codec(), mask_prob=0.15).train()
        run(w, dict(coarse_token_ids=grid[..., :3].contiguous(), fine_token_ids=grid[..., 3:].contiguous()), f'round {r} fine N=2049 flash_attn={flash}')
        del m, w
        torch.manual_seed(0)
        m = A.SemanticTransformer(dim=1024, depth=6, num_semantic_tokens=500, flash_attn=flash).to(dev)
        w = A.SemanticTransformerWrapper(transformer=m, unique_consecutive=False, mask_prob=0.15).train()
        run(w, dict(semantic_token_ids=torch.randint(0, 500, (B, 2047), generator=g).to(dev)), f'round {r} semantic N=2048 flash_attn={flash}')
        del m, w
