"""audiolm_pytorch_amd -- MI355X (gfx950) native implementation of the audiolm-pytorch token-transformer training hot path.

The directory is called `audiolm-pytorch_amd/` (repo contract); import it as `audiolm_pytorch_amd` through the loader module
`audiolm_pytorch_amd.py` at the repo root.  `install_as_reference()` additionally registers these modules under the reference's
import paths (`audiolm_pytorch.audiolm_pytorch`, `.attend`, `.soundstream`) so that an unmodified reference trainer.py picks
them up.
"""
from __future__ import annotations

import sys
import types

from . import _lib

_lib.load()            # no fallback path exists: fail at import, loudly, if libaudiolm_hip.so can neither be found nor built

from .attend import Attend  # noqa: E402
from .audiolm_pytorch import (AudioLM, CoarseTransformer, CoarseTransformerWrapper, FineTransformer, FineTransformerWrapper,
                              SemanticTransformer, SemanticTransformerWrapper, Transformer, get_embeds)
from .optimizer import FusedAdam, get_optimizer
from .soundstream import SoundStream
from .version import __version__


def install_as_reference():
    """Makes `from audiolm_pytorch.audiolm_pytorch import CoarseTransformer, ...` (reference trainer.py:36-48) resolve to this package."""
    from . import attend, audiolm_pytorch
    pkg = types.ModuleType('audiolm_pytorch')
    pkg.__path__ = []
    for name in ('SemanticTransformer', 'CoarseTransformer', 'FineTransformer', 'SemanticTransformerWrapper', 'CoarseTransformerWrapper',
                 'FineTransformerWrapper', 'AudioLM', 'get_embeds'):
        setattr(pkg, name, getattr(audiolm_pytorch, name))
    sys.modules['audiolm_pytorch'] = pkg
    sys.modules['audiolm_pytorch.audiolm_pytorch'] = audiolm_pytorch
    sys.modules['audiolm_pytorch.attend'] = attend
    from . import optimizer
    sys.modules['audiolm_pytorch.optimizer'] = optimizer          # trainer.py:32 `from audiolm_pytorch.optimizer import get_optimizer`
    try:
        from . import soundstream
        sys.modules['audiolm_pytorch.soundstream'] = soundstream
        pkg.SoundStream = soundstream.SoundStream
    except ImportError:
        pass
    return pkg
