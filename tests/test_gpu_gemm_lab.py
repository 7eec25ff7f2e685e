"""The bench-only GEMM variants (csrc/lab/gemm_lab.hip -> libaudiolm_gemm_lab.so; never loaded by the package): every main loop that was
built, measured and NOT adopted stays correct, so the negative results of DESIGN.md section 8.1 remain reproducible."""
import ctypes
import os

import pytest
import torch

from test_gpu_kernels import BF16, F32, dev, relmax, rnd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_I, _L, _F, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p


@pytest.fixture(scope='module')
def lab():
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import build
    lib = ctypes.CDLL(build.build_lab())
    lib.almlab_gemm_bf16_nt_tile.argtypes = [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _F, _I, _I, _I, _P]
    sk = [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _L, _L, _L, _F, _I, _P]
    lib.almlab_gemm_bf16_nt_splitk.argtypes = sk
    lib.almlab_gemm_bf16_tn_splitk.argtypes = sk
    lib.almlab_gemm_splitk_ws_floats.argtypes = [_I, _I, _I, _I]
    lib.almlab_debug_stream.argtypes = [_I]
    return lib


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize('tile', [2, 3, 4, 6, 7, 8, 9, 10, 12])
@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (512, 768, 1024), (300, 520, 200), (1000, 2736, 1024), (257, 300, 2736), (512, 256, 128), (640, 384, 192), (1536, 5472, 128), (4096, 2736, 192)])
def test_lab_tile_configs(lab, M, N, K, tile):
    """256x128x64 with the 3-stage counted-vmcnt DMA ring (1, 2, 3 and more K-steps: prologue / steady state / drain), the persistent kernel
    (more tiles than CUs: cross-tile prefetch), the software-pipelined loops, B-from-registers, the 32-deep rings."""
    A, B = rnd(M, K, seed=11, dtype=BF16), rnd(N, K, seed=12, dtype=BF16)
    for dt, tol in ((F32, 2e-5), (BF16, 4e-3)):
        C = torch.full((M + 3, N + 5), float('nan'), dtype=dt, device=dev())
        Cv = C[:M, :N]
        rc = lab.almlab_gemm_bf16_nt_tile(A.data_ptr(), B.data_ptr(), Cv.data_ptr(), None, M, N, K, A.stride(0), B.stride(0), Cv.stride(0), 1.0,
                                          int(dt == F32), 0, tile, _st())
        assert rc == 0, rc
        err = relmax(Cv, A.double() @ B.double().t())
        assert err <= tol, f'lab gemm tile={tile} {M}x{N}x{K} {dt}: rel-max err {err}'
        assert bool(torch.isnan(C[M:]).all()) and bool(torch.isnan(C[:, N:]).all()), 'GEMM wrote outside its tile'


def _splitk(lab, name, A, B, C, M, N, K, nb, sA, sB, sC, alpha, accumulate):
    nws = lab.almlab_gemm_splitk_ws_floats(M, N, K, nb)
    assert nws >= 0
    ws = torch.empty(max(nws, 1), dtype=F32, device=dev())
    lda = A.stride(-2)
    ldb = B.stride(-2)
    rc = getattr(lab, name)(A.data_ptr(), B.data_ptr(), C.data_ptr(), ws.data_ptr(), M, N, K, lda, ldb, C.stride(-2), nb, sA, sB, sC, alpha, int(accumulate), _st())
    assert rc == 0, rc
    torch.cuda.synchronize()


@pytest.mark.parametrize('M,N,K,nb', [(512, 1024, 4096, 1), (2730, 1024, 2048, 1), (1025, 300, 1100, 1), (1024, 2730, 16384, 1), (530, 512, 3000, 2), (256, 256, 1024, 1)])
def test_lab_balanced_split(lab, M, N, K, nb):
    """the balanced split (one K-step range per CU, partial tiles in the workspace, second-stage sum) forced on, TN and NT, vs fp64;
    and bit-identical results from two runs (no atomics, fixed summation order)."""
    lab.almlab_debug_stream(2)
    try:
        Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8
        At, Bt = rnd(nb, K, Mp, seed=40, dtype=BF16), rnd(K, Np, seed=41, dtype=BF16)
        C = torch.full((nb, M, N), float('nan'), dtype=F32, device=dev())
        run = lambda out, alpha=1.0, acc=False: _splitk(lab, 'almlab_gemm_bf16_tn_splitk', At, Bt, out, M, N, K, nb, K * Mp, 0, M * N, alpha, acc)
        run(C)
        ref = torch.einsum('bkm,kn->bmn', At[:, :, :M].double(), Bt[:, :N].double())
        assert relmax(C, ref) <= 3e-5
        C2 = torch.empty_like(C)
        run(C2)
        assert torch.equal(C, C2)
        C0 = rnd(*C.shape, seed=42)
        C1 = C0.clone()
        run(C1, 0.5, True)
        assert relmax(C1, C0.double() + 0.5 * ref) <= 3e-5
        if nb == 1:
            Kp = (K + 7) // 8 * 8
            A, B = rnd(M, Kp, seed=43, dtype=BF16), rnd(N, Kp, seed=44, dtype=BF16)
            Cn = torch.full((M, N), float('nan'), dtype=F32, device=dev())
            _splitk(lab, 'almlab_gemm_bf16_nt_splitk', A, B, Cn, M, N, Kp, 1, 0, 0, 0, 1.0, False)
            assert relmax(Cn, A.double() @ B.double().t()) <= 3e-5
    finally:
        lab.almlab_debug_stream(0)
