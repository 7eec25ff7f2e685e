"""A/B of GEMM main-loop variants in ONE process: the bench-only library (csrc/lab/gemm_lab.hip: every tile id, incl. the ones that were
not adopted) and any alternative build of it (`scripts/ubench/bin/libgemm_v*.so`, e.g. `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared
-DALM_GEMM_WHATIF=8 -o scripts/ubench/bin/libgemm_v1.so audiolm-pytorch_amd/csrc/lab/gemm_lab.hip`), for every tile id given.  The libraries are
timed round-robin on the same buffers, so clock ramps and box-to-box differences cancel.
Usage: python scripts/ab_gemm.py [tile ...]   (default tiles: 2 13 11 7 10 12)
"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import _lib  # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16


import gemm_lab  # noqa: E402


def main():
    tiles = [int(a) for a in sys.argv[1:]] or [2, 13, 11, 7, 10, 12]
    libs = [('lab', gemm_lab.bind())] + [(os.path.basename(p)[3:-3], gemm_lab.bind(p)) for p in sorted(glob.glob(os.path.join(ROOT, 'scripts/ubench/bin/libgemm_v*.so')))]
    T = 16384
    shapes = [('W1 fwd', T, 5472, 1024), ('W2 fwd', T, 1024, 2736), ('dHN dgrad', T, 2736, 1024), ('dXN2 dgrad', T, 1024, 5472), ('square 8192', 8192, 8192, 8192)]
    st = torch.cuda.current_stream().cuda_stream
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device=dev).to(BF16)
        B = torch.randn(N, K, device=dev).to(BF16)
        C = torch.empty(M, N, dtype=BF16, device=dev)
        ref = None
        for tile in tiles:
            def run(lib):
                rc = lib.almlab_gemm_bf16_nt_tile(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, A.stride(0), B.stride(0), C.stride(0), 1.0, 0, 0, tile, st)
                assert rc == 0, rc
            best = {n: 1e9 for n, _ in libs}
            for n, lib in libs:                               # correctness of every variant against the first library
                C.zero_()
                run(lib)
                torch.cuda.synchronize()
                if ref is None:
                    ref = C.clone()
                if not torch.equal(C, ref):
                    print(f'  MISMATCH {name} tile {tile} {n}: max |d| = {float((C.float() - ref.float()).abs().max()):.3e}', flush=True)
            for rnd in range(6):
                for n, lib in libs:
                    for _ in range(3):
                        run(lib)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        run(lib)
                    e1.record()
                    torch.cuda.synchronize()
                    best[n] = min(best[n], e0.elapsed_time(e1) / 20)
            fl = 2.0 * M * N * K
            print(f'{name:12s} tile {tile}: ' + ' | '.join(f'{n} {t * 1e3:7.1f} us {fl / t / 1e9:6.0f} TF' for n, t in best.items()), flush=True)


if __name__ == '__main__':
    main()
