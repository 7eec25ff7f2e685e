"""Builds libaudiolm_hip.so (every HIP kernel + the C ABI, include/audiolm_hip.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libaudiolm_hip.so')
STAMP = os.path.join(HERE, '.libaudiolm_hip.stamp')
SOURCES = ['gemm.hip', 'norm_act.hip', 'attention.hip', 'hyper.hip', 'embed_ce.hip', 'codec.hip', 'relpos.hip', 'optim.hip', 'decode.hip']


def _digest() -> str:
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.hpp', '.h'))] + [os.path.join(HERE, '..', 'include', 'audiolm_hip.h')]
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())      # content only: the tree is copied to another path on the GPU box
    return h.hexdigest()


def _fresh(dig: str) -> bool:
    return os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and _fresh(dig):
        return LIB
    # one builder at a time (N ranks of a torch.distributed.run launch import the package concurrently): exclusive lock, re-check, compile
    # into a temporary file and rename it into place, so that no process can ever dlopen a half-written library
    import fcntl
    with open(os.path.join(HERE, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh(dig):
                return LIB
            hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
            srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
            tmp = LIB + f'.tmp{os.getpid()}'
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value', '-o', tmp] + srcs
            if verbose:
                print('[audiolm_pytorch_amd] building', os.path.basename(LIB), file=sys.stderr)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            with open(STAMP, 'w') as fh:
                fh.write(dig)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
