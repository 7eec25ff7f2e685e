"""pytest configuration: registers the `gpu` marker and puts the repo root / oracle / golden helpers on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host-side logic, C-ABI symbol checks, gloo world_size-2 tests (CPU only).
`-m gpu`       : parity tests proper -- the HIP path (through the C-ABI) vs the oracle, on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
