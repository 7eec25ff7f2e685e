"""CPU: the kernel every dense launch of the BENCHMARKED step takes, read from the library's own host-side plan queries (no GPU, no launch).

Why this exists (round-3 VERDICT, "the benchmarked configuration is not the tested configuration"): every model-level parity case used to run B = 1 or 2
(M = 2048 rows), where `pick_tile` sends every NT GEMM to the 128 x 128 kernel, while bench.py times B = 8 (M = 16 384), where the FFN / projection GEMMs run on
`gemm_stag_kernel<NT>` (tile 13) / `gemm_kernel<384,256>` (tile 11) and the layer-batched weight gradients on the hybrid plan (`gemm_w4_kernel<TN>` at full K + a
split-K tail).  tests/test_gpu_benchshape.py and the B = 8 cases of tests/test_gpu_fullsize.py / test_gpu_opwise.py put those kernels inside oracle comparisons;
this file pins WHICH kernels they are, so that a change of the selection rules cannot silently move the benchmark (or the tests) onto other kernels.
Shapes: CoarseTransformer dim 1024, depth 6, heads 8 x 64, FFN inner 2730 (padded 2736), N = 2048 (reference audiolm_pytorch.py:255-259, :351, :395).
"""
import ctypes

import audiolm_pytorch_amd  # noqa: F401
from audiolm_pytorch_amd import _lib

D, HD, DH2, I, IP, L = 1024, 512, 128, 2730, 2736, 6


def nt(M, N, nb=1):
    return _lib.query('alm_gemm_nt_tile_choice', M, N, nb)


def tn(M, N, K, nb):
    plan = (ctypes.c_int * 4)()
    kind = _lib.query('alm_gemm_tn_batched_plan', M, N, K, nb, ctypes.cast(plan, ctypes.c_void_p))
    return kind, list(plan)


def test_nt_tiles_of_the_benchmarked_step():
    M = 8 * 2048
    # forward: to_q / to_kv on the 128 x 128 tile (N = 512 / 128: too few 256 x 256 tiles for 256 CUs), to_out + W2 on the staggered 256 x 256 tile,
    # W1 on 384 x 256 (946 tiles = 4 rounds instead of 6)
    assert nt(M, HD) == 1 and nt(M, DH2) == 1
    assert nt(M, D) == 13                      # to_out (K = 512), W2 (K = 2736), dXN of both branches, the K/V-path gradient (K = 128)
    assert nt(M, 2 * IP) == 11                 # W1 forward: U = XN W1^T
    assert nt(M, IP) == 11                     # dHN = dY W2 (473 tiles of 384 x 256 = 2 rounds instead of 3)
    # logit heads: the per-quantizer coarse head (3 problems of 4096 x 1025) on the big tile, the semantic head (4072 x 501) on the small one
    assert nt(8 * 512, 1025, 3) == 13
    assert nt(8 * 509, 501, 1) == 1


def test_nt_tiles_at_the_old_test_batch_are_all_small():
    """B = 1 (what every full-size parity case ran before round 4): not one launch of the step reaches a big tile -- the reason the B = 8 cases exist"""
    M = 2048
    for N in (HD, DH2, D, 2 * IP, IP):
        assert nt(M, N) == 1, N


def test_batched_weight_gradient_plans_of_the_benchmarked_step():
    K = 8 * 2048
    # dW1: 12 problems (6 layers x the x / gate halves) of 2730 x 1024 = 528 tiles: 512 at full K (128 panels of 4), tail = the last 4 row blocks of the last problem
    kind, plan = tn(I, D, K, 2 * L)
    assert kind == 2 and plan[0] == 128 and plan[3] == 1 and plan[2] == (11 - 4) * 256 and plan[1] >= 8, (kind, plan)
    # dW2: 6 problems of 1024 x 2730 = 264 tiles: 256 at full K (64 panels of 4 along N), tail = the last 2 column blocks
    kind, plan = tn(D, I, K, L)
    assert kind == 2 and plan[0] == 64 and plan[3] == 0 and plan[2] == (11 - 2) * 256 and plan[1] >= 8, (kind, plan)
    # dWo / dWq / dWkv: 48 / 48 / 24 tiles of 256 x 256 -- uniform split-K
    for Mw, Nw in ((D, HD), (HD, D), (DH2, D)):
        kind, plan = tn(Mw, Nw, K, L)
        assert kind == 1 and plan[1] > 1, (Mw, Nw, kind, plan)
    # B = 1: K = 2048 tokens is below the hybrid plan's 4096 floor
    assert tn(I, D, 2048, 2 * L)[0] != 2
