"""Which torch (ATen) kernels does ONE eager training step launch besides the library's own, and from which line of the package?  The round-2 kernel
profile shows ~110 such launches per step (fills, aranges, copies, small elementwise ops): 0.5 ms of kernel time plus their launch gaps.
usage: python scripts/glue_profile.py [config]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'coarse2048'
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
W = bench.build(cfg, dev, 0, torch.bfloat16)
model, wrapper, inputs = W['model'], W['wrapper'], W['inputs']


def step():
    for p in model.parameters():
        p.grad = None
    loss = wrapper(**inputs, return_loss=True)
    loss.backward()


torch.autograd.set_multithreading_enabled(False)
for _ in range(6):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    where = next((f for f in ev.stack if 'audiolm' in f or 'bench.py' in f), ev.stack[0] if ev.stack else '?')
    where = where.replace(ROOT + '/', '')
    a = agg[(ev.name, where)]
    a[0] += 1
    a[1] += ev.device_time_total
tot_n, tot_t = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
print(f'leaf ATen ops with device time in one step: {tot_n} ops, {tot_t:.0f} us of kernels')
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f'{n:4d} x {t / n:7.1f} us  {name:28s} {where[:150]}')
