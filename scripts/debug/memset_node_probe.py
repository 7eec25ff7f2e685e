"""Does a hipMemsetAsync captured into a hipGraph (a memset NODE) clear its buffer on every replay, in order with the kernel nodes around it?
buffer <- 7 (kernel), clear(buffer), buffer += 1 (kernel); expected after every replay: all ones.  clear = hipMemsetAsync (through ctypes: the runtime's own
entry point) or alm_memset_zero (this library's kernel).  Measured on ROCm 7.0 / torch 2.10 (profiles/r6u_memset_node_probe.log): the memset node of 16 B and of
10 KB leaves garbage from the second replay on, 1 MB is fine; the kernel is fine at every size -- which is why no product path issues hipMemsetAsync."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import audiolm_pytorch_amd  # noqa: F401
from audiolm_pytorch_amd import ops

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device('cuda')


def runtime_memset(buf):
    rc = hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for name, clear in (('hipMemsetAsync', runtime_memset), ('alm_memset_zero', ops.memset_zero)):
    for n in (16, 2565 * 4, 1 << 20):
        buf = torch.empty(n // 4, dtype=torch.int32, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                buf.fill_(7); clear(buf); buf.add_(1)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            buf.fill_(7)
            clear(buf)
            buf.add_(1)
        res = []
        for _ in range(3):
            g.replay()
            torch.cuda.synchronize()
            res.append((int(buf.min()), int(buf.max())))
        print(f'{name} bytes {n}: (min, max) after each replay = {res}  expected (1, 1)', flush=True)
