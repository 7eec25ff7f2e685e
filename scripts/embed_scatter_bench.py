"""Times the destination-owned embedding scatter (alm_embed_scatter_owned) at the headline shape (16 384 tokens, tables 501 / 3 075 / 3 / 1 / 1 rows) and at the
codebook-4096 shape of BASELINE configs[4] (66 024 tokens, tables 501 / 12 291 / 3 / 1 / 1).   usage: python scripts/embed_scatter_bench.py
A/B of two builds on one box: ALM_LIB_PATH=/path/to/other/libaudiolm_hip.so python scripts/embed_scatter_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import ops  # noqa: E402

dev = torch.device('cuda')


def run(tokens, nsem, ncoarse, D=1024, hot=0.0):
    g = torch.Generator().manual_seed(0)
    tables = [torch.empty(nsem, D, device=dev), torch.empty(ncoarse, D, device=dev), torch.empty(3, D, device=dev), torch.empty(1, D, device=dev), torch.empty(1, D, device=dev)]
    ns = tokens // 4
    a = torch.cat([torch.randint(0, nsem, (ns,), generator=g), torch.randint(0, ncoarse, (tokens - ns,), generator=g) + (1 << 24)]).to(torch.int32)
    b = torch.cat([torch.full((ns,), -1), torch.randint(0, 3, (tokens - ns,), generator=g) + (2 << 24)]).to(torch.int32)
    if hot > 0:                                     # a fraction `hot` of the coarse tokens on 4 rows (skewed ids: silence, a collapsed codebook)
        m = torch.rand(tokens, generator=g) < hot
        m[:ns] = False
        a[m] = (torch.randint(0, 4, (int(m.sum()),), generator=g) * 7 + (1 << 24)).to(torch.int32)
    a, b = a.to(dev), b.to(dev)
    dout = torch.randn(tokens, D, device=dev)
    for _ in range(3):
        ops.embed_scatter_owned(tables, a, b, dout, 1.0, tokens, D)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.embed_scatter_owned(tables, a, b, dout, 1.0, tokens, D)
    e1.record()
    torch.cuda.synchronize()
    print(f'{os.path.basename(os.environ.get("ALM_LIB_PATH", "default"))}: tokens {tokens} rows {nsem}+{ncoarse} hot {hot}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us (small + owned launches)', flush=True)


if os.environ.get('SCATTER_CASES') == 'headline':
    run(16384, 501, 3075)
    sys.exit(0)
run(16384, 501, 3075)
run(66024, 501, 12291)
run(66024, 501, 3075)
run(16384, 501, 12291)
run(66024, 501, 12291, hot=0.5)
run(16384, 501, 3075, hot=0.5)
run(16384, 501, 3075, hot=0.1)
run(66024, 501, 12291, hot=0.1)
