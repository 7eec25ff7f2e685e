"""cProfile of the Python / ctypes host side of ONE eager training step (headline shape): where the ~10 ms of issue time per step go.
usage: python scripts/host_profile.py [config]"""
import cProfile
import os
import pstats
import sys

import torch

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


torch.autograd.set_multithreading_enabled(False)          # backward nodes run in this thread: cProfile sees them
for _ in range(8):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(70)
