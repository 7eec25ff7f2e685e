"""Bench-only: where do the cycles of the forward attention tile loop go?  Builds csrc/attention.hip with -DALM_ATTN_PROBE (s_memtime stamps between the
segments of a tile step, per wave) into scripts/ubench/bin/libalm_probe.so next to the product objects, runs the headline shape and prints cycles per
tile step and segment.  usage:  python scripts/attn_probe.py build   (CPU, cross-compiles)   |   python scripts/attn_probe.py run   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'audiolm-pytorch_amd')
KERNEL = os.environ.get('ALM_PROBE_KERNEL', 'fwd')                   # 'fwd' or 'dkv'
VARIANT = os.environ.get('ALM_PROBE_VARIANT', '')                  # '' or 'NOSTAGE' (leave the tile DMA out: wrong results, timing only)
LIB = os.path.join(ROOT, 'scripts', 'ubench', 'bin', f'libalm_probe{KERNEL}{VARIANT}.so')
SEG = ['stage (DMA issue, key side)', 'score init + K reads + S MFMA issue', 'causal mask + row max (+ rescale)', 'exp2 + row sums', 'pack + V^T reads + PV MFMA issue',
       'barrier (vmcnt 0)', 'epilogue']


def build():
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value']
    obj = '/tmp/attention_probe.o'
    subprocess.run(['/opt/rocm/bin/hipcc', *flags, '-DALM_DKV_PROBE' if KERNEL == 'dkv' else '-DALM_ATTN_PROBE', *([f'-DALM_PROBE_{VARIANT}'] if VARIANT else []), '-c', '-o', obj, os.path.join(PKG, 'csrc', 'attention.hip')], check=True)
    objs = [os.path.join(PKG, 'build', f) for f in sorted(os.listdir(os.path.join(PKG, 'build'))) if f.endswith('.o') and not f.startswith('attention')]
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs, obj], check=True)
    print('built', LIB)


def run():
    os.environ['ALM_LIB_PATH'] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import ops
    dev, BF16 = torch.device('cuda'), torch.bfloat16
    B, N, H, d = 8, 2048, 8, 64
    q = torch.randn(B * N, H * d, device=dev).to(BF16)
    kv = torch.randn(B * N, 2 * d, device=dev).to(BF16)
    k, v = kv[:, :d], kv[:, d:]
    mask = (torch.rand(B, N, device=dev) > 0.15).to(torch.uint8)
    mask[:, 0] = 1
    o, lse = ops.mqa_attn_fwd(q, k, v, mask, B, N, H, d)
    do = torch.randn(B * N, H * d, device=dev).to(BF16)
    for _ in range(5):
        if KERNEL == 'dkv':
            ops.mqa_attn_bwd(q, k, v, mask, o, lse, do, B, N, H, d)
        else:
            ops.mqa_attn_fwd(q, k, v, mask, B, N, H, d)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(LIB)
    n = 4096 * 8
    buf = (ctypes.c_ulonglong * n)()
    rc = lib.alm_attn_probe_read(buf, n)
    assert rc == 0, rc
    import numpy as np
    if KERNEL == 'dkv':
        a = np.array(buf, dtype=np.float64).reshape(4096, 8)
        steps = a[:, 7].sum()
        print(f'dK/dV kernel: {len(a)} waves, {a[:, 7].sum() / 8:.0f} workgroup steps; mean cycles per step and segment (weighted over all waves):')
        for i, name in enumerate(['stage (8 + 1 DMA pieces)', 'row terms + Q / dO reads + S, dP MFMA issue', 'exp2, mask, dS', 'pack + Q^T / dO^T reads + dV, dK MFMA issue',
                                  'barrier (vmcnt 0)']):
            print(f'    {name:46s} {a[:, i].sum() / steps:9.0f}')
        heavy = a[a[:, 7] == a[:, 7].max()]
        print(f'  heaviest workgroups ({int(a[:, 7].max())} steps): total cycles per wave {heavy[:, :5].sum(1).mean():.0f}')
        return
    qb, nwg, steps = 1, 32 * 2 * B, 33
    a = np.array(buf, dtype=np.float64).reshape(4096, 8)[:nwg * 4]
    tot = a[:, :7].sum(1)
    print(f'QB={qb}: {nwg} workgroups; per-wave total cycles: min {tot.min():.0f}  mean {tot.mean():.0f}  max {tot.max():.0f}')
    print(f'  every workgroup runs {steps} tile steps; mean cycles per step and segment (all waves):')
    for i, name in enumerate(SEG):
        print(f'    {name:42s} {a[:, i].mean() / (steps if i < 6 else 1):9.0f}' + ('  (per pass pair)' if i == 6 else ''))


if __name__ == '__main__':
    (build if sys.argv[1:] == ['build'] else run)()
