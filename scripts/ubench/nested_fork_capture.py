"""Minimal repro attempt for the `graph2` core dump of round 2: NESTED stream forks inside a hipGraph capture, pure torch -- no
kernel of this package is involved.  The capture stream forks a second stream (the two-half-batch schedule), and each of the two forks its own side
stream (the weight-gradient streams), all joined again before the capture ends.  If THIS dumps core in hipStreamEndCapture the defect is in the
ROCm runtime's capture of nested forks, not in libaudiolm_hip.so.   usage: python scripts/ubench/nested_fork_capture.py [nested|flat]"""
import faulthandler
import sys

import torch

faulthandler.enable()
mode = sys.argv[1] if len(sys.argv) > 1 else 'nested'
dev = torch.device('cuda:0')
a = torch.randn(1024, 1024, device=dev)
s2, side_a, side_b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def fork(src, dst):
    ev = torch.cuda.Event()
    ev.record(src)
    dst.wait_event(ev)


def step():
    cur = torch.cuda.current_stream()
    fork(cur, s2)
    y1 = a @ a
    if mode == 'nested':
        fork(cur, side_a)
        with torch.cuda.stream(side_a):
            z1 = y1 @ a
    else:
        z1 = y1 @ a
    with torch.cuda.stream(s2):
        y2 = a @ a.t()
        if mode == 'nested':
            fork(s2, side_b)
            with torch.cuda.stream(side_b):
                z2 = y2 @ a
            fork(side_b, s2)
        else:
            z2 = y2 @ a
    if mode == 'nested':
        fork(side_a, cur)
    fork(s2, cur)
    return z1 + z2


cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    for _ in range(3):
        ref = step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print('capture ended', flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print(mode, 'ok: max |replay - eager| =', float((out - ref).abs().max()), flush=True)
