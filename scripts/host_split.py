"""Host (Python / ctypes) ISSUE time of one eager training step, split: the transformer stack's forward / backward launch sequences (core.stack_forward /
stack_backward) vs everything around them (wrapper bookkeeping, embeddings, heads, autograd engine).  No profiler: perf_counter around the calls, no device
synchronisation inside the step.   usage: python scripts/host_split.py [config]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from audiolm_pytorch_amd import core  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'coarse2048'
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
W = bench.build(cfg, dev, 0, torch.bfloat16)
model, wrapper, inputs = W['model'], W['wrapper'], W['inputs']
params = [p for p in model.parameters()]
acc = {'sf': 0.0, 'sb': 0.0}
osf, osb = core.stack_forward, core.stack_backward


def sf(*a, **k):
    t0 = time.perf_counter()
    r = osf(*a, **k)
    acc['sf'] += time.perf_counter() - t0
    return r


def sb(*a, **k):
    t0 = time.perf_counter()
    r = osb(*a, **k)
    acc['sb'] += time.perf_counter() - t0
    return r


core.stack_forward, core.stack_backward = sf, sb
from audiolm_pytorch_amd import launchlist  # noqa: E402
if launchlist.ENABLED:                       # replayed launch lists by-pass core.stack_*: time the list calls instead (the sized / recorded passes fall in the warm-up)
    olf, olb = launchlist.forward, launchlist.backward

    def lf(*a, **k):
        t0 = time.perf_counter()
        r = olf(*a, **k)
        acc['sf'] = acc['sf'] + time.perf_counter() - t0 if launchlist.STATS['replayed'] else acc['sf']
        return r

    def lb(*a, **k):
        t0 = time.perf_counter()
        r = olb(*a, **k)
        acc['sb'] = acc['sb'] + time.perf_counter() - t0 if launchlist.STATS['replayed'] else acc['sb']
        return r
    launchlist.forward, launchlist.backward = lf, lb


def step():
    for p in params:
        p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = wrapper(**inputs, return_loss=True)
    t1 = time.perf_counter()
    loss.backward()
    return t1


for _ in range(8):
    step()
torch.cuda.synchronize()
n = 10
acc['sf'] = acc['sb'] = 0.0
tot = fwd = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t1 = step()
    t2 = time.perf_counter()
    tot += t2 - t0
    fwd += t1 - t0
torch.cuda.synchronize()
print(f'launch lists: {dict(launchlist.STATS, enabled=launchlist.ENABLED)}')
print(f'{cfg}: issue {tot / n * 1e3:.2f} ms/step = forward {fwd / n * 1e3:.2f} (stack {acc["sf"] / n * 1e3:.2f}) + backward {(tot - fwd) / n * 1e3:.2f} (stack {acc["sb"] / n * 1e3:.2f})')
