"""Builds libaudiolm_hip.so (every HIP kernel + the C ABI, include/audiolm_hip.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
Each translation unit is compiled to its own object (in parallel, re-used while its source digest is unchanged), then linked.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')                       # git-ignored (objects + per-object digests)
LIB = os.path.join(HERE, 'libaudiolm_hip.so')
STAMP = os.path.join(HERE, '.libaudiolm_hip.stamp')
SOURCES = ['gemm.hip', 'norm_act.hip', 'attention.hip', 'hyper.hip', 'embed_ce.hip', 'codec.hip', 'relpos.hip', 'optim.hip', 'decode.hip',
           'xattn.hip', 'local_attn.hip', 'launchlist.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value']


def _headers() -> bytes:
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hpp', '.h'))] + [os.path.join(HERE, '..', 'include', 'audiolm_hip.h')]
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.digest()


def _src_digest(name: str, hdr: bytes) -> str:
    h = hashlib.sha256(hdr + ' '.join(FLAGS).encode())
    with open(os.path.join(CSRC, name), 'rb') as fh:
        h.update(name.encode() + b'\0' + fh.read())            # content only: the tree is copied to another path on the GPU box
    return h.hexdigest()


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _digest() -> str:
    hdr = _headers()
    h = hashlib.sha256()
    for s in _sources():
        h.update(_src_digest(s, hdr).encode())
    return h.hexdigest()


def _fresh(dig: str) -> bool:
    return os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and _fresh(dig):
        return LIB
    # one builder at a time (N ranks of a torch.distributed.run launch import the package concurrently): exclusive lock, re-check, link
    # into a temporary file and rename it into place, so that no process can ever dlopen a half-written library
    import fcntl
    with open(os.path.join(HERE, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh(dig):
                return LIB
            hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
            os.makedirs(OBJ, exist_ok=True)
            hdr = _headers()
            if verbose:
                print('[audiolm_pytorch_amd] building', os.path.basename(LIB), file=sys.stderr)

            def compile_one(name):
                obj = os.path.join(OBJ, name + '.o')
                tag = os.path.join(OBJ, name + '.digest')
                d = _src_digest(name, hdr)
                if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read().strip() == d:
                    return obj
                tmp = obj + f'.tmp{os.getpid()}'
                try:
                    subprocess.run([hipcc, *FLAGS, '-c', '-o', tmp, os.path.join(CSRC, name)], check=True)
                    os.replace(tmp, obj)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                with open(tag, 'w') as fh:
                    fh.write(d)
                return obj

            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
                objs = list(pool.map(compile_one, _sources()))
            tmp = LIB + f'.tmp{os.getpid()}'
            try:
                subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp, *objs], check=True)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            with open(STAMP, 'w') as fh:
                fh.write(dig)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


LAB_SRC = os.path.join(CSRC, 'lab', 'gemm_lab.hip')
LAB_LIB = os.path.join(HERE, 'libaudiolm_gemm_lab.so')


def build_lab(force: bool = False) -> str:
    """bench-only library of the GEMM variants that were measured and not adopted (csrc/lab/gemm_lab.hip).  The package never loads it:
    scripts/ab_gemm.py, scripts/kbench.py and tests/test_gpu_gemm_lab.py do."""
    import fcntl
    h = hashlib.sha256(_headers() + ' '.join(FLAGS).encode())
    for f in (LAB_SRC, os.path.join(CSRC, 'lab', 'gemm_lab.h')):
        with open(f, 'rb') as fh:
            h.update(fh.read())
    dig, stamp = h.hexdigest(), LAB_LIB + '.stamp'
    fresh = lambda: os.path.exists(LAB_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig
    if not force and fresh():
        return LAB_LIB
    with open(os.path.join(HERE, '.build_lab.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():
                return LAB_LIB
            tmp = LAB_LIB + f'.tmp{os.getpid()}'
            try:
                subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), *FLAGS, '-Wno-unused-function', '-shared', '-o', tmp, LAB_SRC], check=True)
                os.replace(tmp, LAB_LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            with open(stamp, 'w') as fh:
                fh.write(dig)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LAB_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    if '--lab' in sys.argv:
        print(build_lab(force='--force' in sys.argv))
