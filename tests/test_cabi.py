"""CPU: the C-ABI shared library builds for gfx950, loads, and exports exactly the symbols include/audiolm_hip.h declares
(and the ctypes binding table covers every one of them).  No compute calls: there is no GPU here."""
import os
import re

import audiolm_pytorch_amd  # noqa: F401
from audiolm_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'audiolm_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(alm_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    names = set(_declared())
    table = set(_lib.SIGNATURES)
    assert names == table, (sorted(names - table), sorted(table - names))


def test_header_argument_counts_match_binding():
    src = open(os.path.join(ROOT, 'include', 'audiolm_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    for m in re.finditer(r'\bint\s+(alm_\w+)\s*\((.*?)\)\s*;', src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ('', 'void') else len(args.split(','))
        assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))


def test_size_queries_run_on_host():
    assert _lib.query('alm_hc_coef_width', 4) == 52
    assert _lib.query('alm_hc_partial_width', 4, 1024) == 1024 * 7 + 26
    assert _lib.query('alm_hc_grads_width', 4, 1024) == 1024 * 8 + 26
    assert _lib.query('alm_ln_partial_blocks', 16384) == 512


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from audiolm_pytorch_amd import ops
    with pytest.raises(_lib.AlmError):
        ops.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8))


def test_import_fails_loudly_without_the_library(tmp_path):
    """No silent fallback: with the shared library unreachable the package import itself raises (nothing on the product path can run)."""
    import subprocess
    import sys
    env = dict(os.environ, ALM_LIB_PATH=str(tmp_path / 'nowhere.so'), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-c', 'import audiolm_pytorch_amd'], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'ImportError' in r.stderr and 'nowhere.so' in r.stderr, r.stderr[-500:]


def test_product_modules_do_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import it"""
    pkg = os.path.join(ROOT, 'audiolm-pytorch_amd')
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), fn


def test_splitk_workspace_covers_the_hybrid_tail():
    """alm_gemm_bf16_tn_batched's hybrid plan (whole waves of 256 tiles at full K + the last problem's remaining blocks as one deep split-K launch)
    writes slices x M2 x N2 partial floats: alm_gemm_splitk_ws_floats, which sizes the caller's workspace, must cover them for every shape the plan
    accepts (host arithmetic only: no GPU).  The plan is restated here from its definition (csrc/gemm.hip hybrid_plan)."""
    import importlib
    L = importlib.import_module('audiolm_pytorch_amd._lib')

    def tail_floats(M, N, K, nb):
        if M < 256 or N < 256 or K < 4096:
            return 0
        tm, tn = -(-M // 256), -(-N // 256)
        m_major = tm >= tn
        tmaj, Q = (tm, tn) if m_major else (tn, tm)
        P, total = tmaj * nb, tmaj * nb * Q
        if total <= 256:
            return 0
        pa = (total // 256) * 256 // Q
        rem = P - pa
        if pa * Q % 256 or rem <= 0 or rem >= tmaj or rem * Q > 64:
            return 0
        off = (tmaj - rem) * 256
        s = 256 // (rem * Q)
        ksteps = -(-K // 64)
        while s > 1 and ksteps // s < 8:
            s -= 1
        if s < 2:
            return 0
        m2, n2 = (M - off, N) if m_major else (M, N - off)
        return s * m2 * n2
    hit = 0
    for M, N in ((2730, 1024), (1024, 2730), (2736, 1024), (4096, 1024), (1024, 512), (3000, 768), (700, 2050)):
        for K in (4096, 8192, 16384, 131072):
            for nb in (1, 2, 3, 6, 12, 24):
                need = tail_floats(M, N, K, nb)
                got = L.query('alm_gemm_splitk_ws_floats', M, N, K, nb)
                assert got == -1 or got >= need, (M, N, K, nb, got, need)
                hit += need > 0
    assert hit >= 10                                            # the benchmark's dW1 / dW2 shapes (6 and 12 problems) are among them
