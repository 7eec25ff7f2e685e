"""TEST INFRASTRUCTURE ONLY (oracle).  Imports the REAL reference (/root/reference) under
sys.modules shims so that its own code can be executed on CPU to generate golden vectors
(tests/golden/make_golden.py).  /root/reference does not exist on the GPU box: nothing in the
`-m gpu` tests, smoke() or bench.py calls this module.

Recipe = SURVEY.md Appendix B (verified in the build container):
 1. import transformers BEFORE stubbing (it probes torchaudio.__spec__)
 2. sys.modules stubs for the third-party packages that are not installed
    (beartype, torchaudio, fairseq, encodec, local_attention, gateloop_transformer,
     vector_quantize_pytorch -> oracle/rvq_restated.py, hyper_connections -> oracle/hyper_connections_restated.py,
     local_attention -> oracle/local_attention_restated.py)
 3. bypass audiolm_pytorch/__init__.py (it imports trainer.py -> wandb, ema_pytorch, ...)
 4. get_encoded_dim -> 768 (constructor otherwise hits the HF hub; audiolm_pytorch.py:604/776/1042)
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'audiolm_pytorch'))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns (audiolm_pytorch.audiolm_pytorch, audiolm_pytorch.soundstream, audiolm_pytorch.attend)
    modules of the REAL reference."""
    if 'audiolm_pytorch.audiolm_pytorch' in sys.modules and getattr(sys.modules['audiolm_pytorch'], '_is_reference_shim', False):
        return (sys.modules['audiolm_pytorch.audiolm_pytorch'], sys.modules['audiolm_pytorch.soundstream'],
                sys.modules['audiolm_pytorch.attend'])
    assert reference_available(), '/root/reference is not present (GPU box?) -- golden vectors are committed under tests/golden/'
    assert 'audiolm_pytorch' not in sys.modules, 'another audiolm_pytorch is already imported in this process'

    import typing
    import torch
    from torch import nn
    try:
        import transformers  # noqa: F401  (must precede the torchaudio stub)
        from transformers import T5Tokenizer, T5EncoderModel, T5Config  # noqa: F401
    except Exception:
        pass

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import hyper_connections_restated
    import rvq_restated
    import local_attention_restated

    ident = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)
    _mod('beartype', beartype=ident)
    _mod('beartype.typing', Tuple=typing.Tuple, Union=typing.Union, Optional=typing.Optional, List=typing.List,
         Type=typing.Type, Dict=typing.Dict, Callable=typing.Callable, Sequence=typing.Sequence)
    _mod('beartype.door', is_bearable=lambda *a, **k: True)
    _mod('beartype.vale', Is=type('Is', (), {'__class_getitem__': classmethod(lambda cls, item: typing.Any)}))

    class _TorchaudioTransform(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            raise NotImplementedError('torchaudio stub')

    def _resample(x, a, b):
        raise NotImplementedError('torchaudio stub')

    ta = _mod('torchaudio')
    ta.functional = _mod('torchaudio.functional', resample=_resample)
    ta.transforms = _mod('torchaudio.transforms', MelSpectrogram=_TorchaudioTransform, Spectrogram=_TorchaudioTransform)
    _mod('fairseq')
    enc = _mod('encodec', EncodecModel=type('EncodecModel', (), {}))
    enc.utils = _mod('encodec.utils', _linear_overlap_add=None)

    class _Absent(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError('third-party module not installed and not restated')

    la = _mod('local_attention', LocalMHA=local_attention_restated.LocalMHA)
    la.transformer = _mod('local_attention.transformer', FeedForward=local_attention_restated.FeedForward,
                          DynamicPositionBias=local_attention_restated.DynamicPositionBias)
    _mod('gateloop_transformer', SimpleGateLoopLayer=_Absent)
    _mod('vector_quantize_pytorch', GroupedResidualVQ=rvq_restated.GroupedResidualVQ,
         GroupedResidualLFQ=rvq_restated.GroupedResidualLFQ, GroupedResidualFSQ=rvq_restated.GroupedResidualFSQ,
         ResidualVQ=rvq_restated.ResidualVQ)
    _mod('hyper_connections',
         get_init_and_expand_reduce_stream_functions=hyper_connections_restated.get_init_and_expand_reduce_stream_functions)
    for n in ('pytorch_warmup', 'wandb'):
        _mod(n)
    _mod('ema_pytorch', EMA=_Absent)

    pkg = types.ModuleType('audiolm_pytorch')
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'audiolm_pytorch')]
    pkg._is_reference_shim = True
    sys.modules['audiolm_pytorch'] = pkg

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import audiolm_pytorch.audiolm_pytorch as A
        import audiolm_pytorch.soundstream as S
        import audiolm_pytorch.attend as AT
    A.get_encoded_dim = lambda name: 768
    return A, S, AT
