"""Bench-only: where do the cycles of the staggered 256x256x64 GEMM go?  Builds csrc/gemm.hip with -DALM_GEMM_PROBE (s_memtime stamps around the DMA issue,
the two barriers and the MFMA part of every slot pair, per wave) into scripts/ubench/bin/libalm_gprobe.so, runs model-shape GEMMs and prints cycles per
slot pair.  usage:  python scripts/gemm_probe.py build   (CPU)   |   python scripts/gemm_probe.py run   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'audiolm-pytorch_amd')
LIB = os.path.join(ROOT, 'scripts', 'ubench', 'bin', 'libalm_gprobe.so')
SEG = ['prologue', 'DMA issue (8 pieces)', 'raw barrier', 'fragment reads + 32 MFMA', 'barrier (vmcnt 0, lgkmcnt 0)', 'tail barriers', 'epilogue']
SEG14 = ['prologue (1.5 stages + first fragments)', 'S0: 16 MFMA || 8 reads || 8 DMA pieces', 'S1 + S2: 32 MFMA || 16 reads', 'vmcnt(0) lgkmcnt(0) + barrier', 'S3: 16 MFMA || 8 reads || 8 DMA pieces', '-', 'epilogue']
TILE = int(os.environ.get('ALM_PROBE_TILE', '13'))


def build():
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value']
    obj = '/tmp/gemm_probe.o'
    subprocess.run(['/opt/rocm/bin/hipcc', *flags, '-DALM_GEMM_PROBE', '-c', '-o', obj, os.path.join(PKG, 'csrc', 'gemm.hip')], check=True)
    objs = [os.path.join(PKG, 'build', f) for f in sorted(os.listdir(os.path.join(PKG, 'build'))) if f.endswith('.o') and not f.startswith('gemm')]
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs, obj], check=True)
    print('built', LIB)


def run():
    os.environ['ALM_LIB_PATH'] = LIB
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import ops
    dev, BF16 = torch.device('cuda'), torch.bfloat16
    lib = ctypes.CDLL(LIB)
    T = 16384
    shapes = [('W1 fwd  NT', T, 5472, 1024, False), ('dXN2 dgrad NT', T, 1024, 5472, False), ('square 8192 NT', 8192, 8192, 8192, False),
              ('dW1 wgrad TN (split-K)', 5472, 1024, T, True)]
    if os.environ.get('ALM_PROBE_SHAPE'):
        shapes = [sh for sh in shapes if os.environ['ALM_PROBE_SHAPE'] in sh[0]]
    for name, M, N, K, tn in shapes:
        if tn:
            At, Bt = torch.randn(K, M, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
            C = torch.empty(M, N, dtype=torch.float32, device=dev)
            fn = lambda: ops.gemm_tn_splitk(At, Bt, C)
        else:
            A, B = torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16)
            C = torch.empty(M, N, dtype=BF16, device=dev)
            fn = lambda: ops.gemm_nt_tile(A, B, C, TILE)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        n = 16384 * 8
        buf = (ctypes.c_ulonglong * n)()
        assert lib.alm_gemm_probe_read(buf, n) == 0
        a = np.array(buf, dtype=np.float64).reshape(16384, 8)
        a = a[a[:, :7].sum(1) > 0]          # NOTE: a launch with fewer waves than an earlier one leaves the earlier one's slots in place (the means of the
        #                                     later shapes then include stale rows): the first shape printed is the clean one
        print(f'{name}: M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1e3:.1f} us (probed build), {len(a)} waves recorded; mean cycles per wave:')
        tot = a[:, :7].sum(1).mean()
        for i, s in enumerate(SEG14 if TILE == 14 else SEG):
            print(f'    {s:34s} {a[:, i].mean():9.0f}  ({100 * a[:, i].mean() / tot:4.1f} %)')


if __name__ == '__main__':
    (build if sys.argv[1:] == ['build'] else run)()
