"""A/B of GEMM tile variants with COLD operands: every launch works on a different (A, B, C) set, enough sets that the 256 MB Infinity Cache never holds
the one that is used next -- the condition of the training step (operands arrive from HBM), unlike scripts/ab_gemm.py's back-to-back loop on ONE set
(warm: A + B + C of most model shapes fit the Infinity Cache and the figures flatter every kernel).  Bench-only library (csrc/lab/gemm_lab.hip).
Usage: python scripts/ab_gemm_cold.py [tile ...]   (default: 13 11 10 12 4 2)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import audiolm_pytorch_amd  # noqa: E402,F401
import gemm_lab  # noqa: E402

dev, BF16 = torch.device('cuda'), torch.bfloat16


def main():
    tiles = [int(a) for a in sys.argv[1:]] or [13, 11, 10, 12, 4, 2]
    lib = gemm_lab.bind()
    T = 16384
    shapes = [('W1 fwd', T, 5472, 1024), ('dHN dgrad', T, 2736, 1024), ('W2 fwd', T, 1024, 2736), ('dXN2 dgrad', T, 1024, 5472), ('Wo fwd', T, 1024, 512)]
    st = torch.cuda.current_stream().cuda_stream
    for name, M, N, K in shapes:
        per_set = 2 * (M * K + N * K + M * N)
        nsets = max(4, int(1.5e9 // per_set))
        sets = [(torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16), torch.empty(M, N, dtype=BF16, device=dev)) for _ in range(nsets)]

        def run(tile, s):
            A, B, C = sets[s]
            rc = lib.almlab_gemm_bf16_nt_tile(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, A.stride(0), B.stride(0), C.stride(0), 1.0, 0, 0, tile, st)
            assert rc == 0, (rc, tile)
        cold, warm = {}, {}
        for rnd in range(4):
            for t in tiles:
                for s in range(nsets):
                    run(t, s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for rep in range(2):
                    for s in range(nsets):
                        run(t, s)
                e1.record()
                torch.cuda.synchronize()
                cold[t] = min(cold.get(t, 1e9), e0.elapsed_time(e1) / (2 * nsets))
                for _ in range(3):
                    run(t, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(t, 0)
                e1.record()
                torch.cuda.synchronize()
                warm[t] = min(warm.get(t, 1e9), e0.elapsed_time(e1) / 20)
        fl = 2.0 * M * N * K
        print(f'{name:11s} ({nsets} sets) ' + ' | '.join(f'tile {t}: cold {cold[t] * 1e3:6.1f} us {fl / cold[t] / 1e9:5.0f} TF (warm {warm[t] * 1e3:6.1f})' for t in tiles), flush=True)
        del sets


if __name__ == '__main__':
    main()
