import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import audiolm_pytorch_amd
from audiolm_pytorch_amd import ops
dev = torch.device('cuda'); BF16 = torch.bfloat16
def timeit(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(512, 1024), (128, 1024), (1024, 512), (5460, 1024), (1024, 2730)]
jobs = []
for r, c in shapes:
    w = torch.randn(r, c, device=dev)
    rp, cp = (r + 7) // 8 * 8, (c + 7) // 8 * 8
    jobs.append((w, torch.empty((rp, cp), dtype=BF16, device=dev), torch.empty((cp, rp), dtype=BF16, device=dev), rp, cp))
t = timeit(lambda: ops.pack_weights_multi(jobs))
n = sum(r * c for r, c in shapes)
print(f'{os.environ.get("ALM_LIB_PATH", "default")}: pack one layer {t:.1f} us  ({n * 8 / t / 1e6:.2f} TB/s)')
