"""ctypes binding of the bench-only GEMM library (csrc/lab/gemm_lab.hip -> libaudiolm_gemm_lab.so): the main-loop / split variants that were
measured and not adopted.  Used by scripts/ab_gemm.py and scripts/kbench.py; the package itself never loads it."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_I, _L, _F, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p
NT_TILE = [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _F, _I, _I, _I, _P]
SPLITK = [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _L, _L, _L, _F, _I, _P]


def bind(path=None):
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import build
    lib = ctypes.CDLL(path or build.build_lab())
    lib.almlab_gemm_bf16_nt_tile.argtypes = NT_TILE
    lib.almlab_gemm_bf16_nt_splitk.argtypes = SPLITK
    lib.almlab_gemm_bf16_tn_splitk.argtypes = SPLITK
    lib.almlab_gemm_splitk_ws_floats.argtypes = [_I, _I, _I, _I]
    lib.almlab_gemm_splitk_slices.argtypes = [_I, _I, _I, _I]
    lib.almlab_debug_stream.argtypes = [_I]
    lib.almlab_debug_splitk.argtypes = [_I, _I, _I]
    return lib
