"""Loader: makes the hyphenated package directory `audiolm-pytorch_amd/` importable as `audiolm_pytorch_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'audiolm-pytorch_amd')
_spec = importlib.util.spec_from_file_location('audiolm_pytorch_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['audiolm_pytorch_amd'] = _mod
_spec.loader.exec_module(_mod)
