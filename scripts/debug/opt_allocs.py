"""Which allocation makes the caching allocator go to the device (hipMalloc) once per step in the `with_optimizer` leg of bench.py?  (round 4: the leg's
optimiser kernels take 0.4 ms of GPU time, the step still gets 1.5-2 ms slower in its "slow mode", and torch.cuda.memory_stats shows one device malloc per
step there.)  Records the allocator history over a few fused-optimiser steps and prints every segment_alloc with the Python frames that asked for it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audiolm_pytorch_amd as A  # noqa: E402

dev = torch.device('cuda:0')
W = bench.build('coarse2048', dev, 0, None)
model, wrapper, inputs = W['model'], W['wrapper'], W['inputs']
cache = model.transformer._cache


def step(opt=None):
    cache.store.clear()
    for p in model.parameters():
        p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = wrapper(**inputs, return_loss=True)
    loss.backward()
    if opt is not None:
        opt.clip_grad_norm_(0.5)
        opt.step()
    return loss


for _ in range(12):
    step()
torch.cuda.synchronize()
opt = A.get_optimizer(model.parameters(), lr=1e-5, wd=0.)
for _ in range(3):
    step(opt)
torch.cuda.synchronize()
s0 = torch.cuda.memory_stats(dev)
torch.cuda.memory._record_memory_history(max_entries=200000)
for _ in range(6):
    step(opt)
torch.cuda.synchronize()
s1 = torch.cuda.memory_stats(dev)
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
print('device mallocs in 6 steps:', s1['num_device_alloc'] - s0['num_device_alloc'], ' frees:', s1['num_device_free'] - s0['num_device_free'],
      ' reserved GB', round(s1['reserved_bytes.all.current'] / 2 ** 30, 2), ' allocated peak GB', round(s1['allocated_bytes.all.peak'] / 2 ** 30, 2))
n = 0
for tr in snap.get('device_traces', []):
    for ev in tr:
        if ev.get('action') in ('segment_alloc', 'segment_free', 'oom'):
            n += 1
            frames = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get('frames', []) if 'site-packages' not in f['filename']][:8]
            print(ev['action'], ev.get('size'), 'stream', ev.get('stream'), frames)
print(n, 'segment events')
