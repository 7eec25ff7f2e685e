"""debug: GPU time and host time of FusedAdam.clip_grad_norm_ + step at the benchmark size, and of the pieces of _prepare"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd as A
from bench import build
dev = torch.device('cuda')
W = build('coarse2048', dev, 0, torch.bfloat16)
model, wrapper, inputs = W['model'], W['wrapper'], W['inputs']
opt = A.get_optimizer(model.parameters(), lr=1e-5, wd=0.)
def fb():
    for p in model.parameters():
        p.grad = None
    wrapper(**inputs, return_loss=True).backward()
for _ in range(5):
    fb(); opt.clip_grad_norm_(0.5); opt.step()
torch.cuda.synchronize()
for rep in range(3):
    fb()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    opt.clip_grad_norm_(0.5)
    t1 = time.perf_counter()
    opt.step()
    t2 = time.perf_counter(); e1.record()
    torch.cuda.synchronize()
    print(f'rep {rep}: GPU {e0.elapsed_time(e1):.3f} ms   host clip {1e3 * (t1 - t0):.3f} ms  host step {1e3 * (t2 - t1):.3f} ms', flush=True)
# pipelined: N steps back to back with and without the optimizer
def timed(n, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def full():
    fb(); opt.clip_grad_norm_(0.5); opt.step()
print(f'fwd+bwd {timed(10, fb):.3f} ms   with FusedAdam {timed(10, full):.3f} ms', flush=True)
