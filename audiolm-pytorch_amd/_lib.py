"""ctypes binding of libaudiolm_hip.so (C ABI declared in include/audiolm_hip.h).

This is the reference-side binding a maintainer would add.  There is NO fallback: if the shared
library is missing / cannot be built the import fails loudly, and every op refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ALM_LIB_PATH') or os.path.join(_HERE, 'libaudiolm_hip.so')      # ALM_LIB_PATH: A/B runs of an alternative build

_P, _I, _L, _F, _U = c_void_p, c_int, c_longlong, c_float, c_ulonglong

# name -> argtypes (all return int).  Kept in sync with include/audiolm_hip.h (tests/test_cabi.py checks both directions).
SIGNATURES = {
    'alm_gemm_bf16_nt': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _L, _L, _L, _L, _L, _L, _F, _I, _I, _P],
    'alm_gemm_bf16_nt_ws': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _L, _L, _L, _L, _L, _L, _F, _I, _I, _P, _L, _P],
    'alm_gemm_bf16_nt_inl': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _F, _I, _I, _I, _P, _L, _P],
    'alm_gemm_nt_plan': [_I, _I, _I, _I, _I, _P],
    'alm_gemm_nt_ws_bytes': [],
    'alm_gemm_splitk_slices': [_I, _I, _I, _I],
    'alm_gemm_splitk_ws_floats': [_I, _I, _I, _I],
    'alm_gemm_splitk_tile': [_I, _I, _I, _I],
    'alm_gemm_nt_tile_choice': [_I, _I, _I],
    'alm_gemm_tn_batched_plan': [_I, _I, _I, _I, _P],
    'alm_gemm_bf16_nt_splitk': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _L, _L, _L, _F, _I, _P],
    'alm_gemm_bf16_tn_splitk': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _L, _L, _L, _F, _I, _P],
    'alm_gemm_bf16_tn_batched': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _L, _L, _L, _L, _L, _L, _F, _I, _P],
    'alm_gemm_bf16_nt_group2': [_P, _P, _P, _I, _I, _I, _L, _L, _L, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P],
    'alm_gemm_bf16_nt_tile': [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _F, _I, _I, _I, _P],
    'alm_transpose_bf16': [_P, _P, _I, _I, _L, _L, _I, _P],
    'alm_transpose_bf16_batched': [_P, _P, _I, _I, _L, _L, _I, _I, _L, _L, _P],
    'alm_pack_weights_multi': [_P, _I, _P],
    'alm_pack_weight': [_P, _I, _I, _L, _P, _L, _I, _I, _P, _L, _P],
    'alm_ln_partial_blocks': [_I],
    'alm_layernorm_fwd': [_P, _I, _L, _P, _P, _I, _L, _P, _L, _P, _P, _I, _I, _P],
    'alm_layernorm_bwd': [_P, _I, _L, _P, _I, _L, _P, _P, _P, _P, _L, _P, _I, _L, _P, _I, _I, _P],
    'alm_colsum': [_P, _I, _L, _I, _I, _P, _F, _I, _P, _P],
    'alm_colsum_chunks': [_I],
    'alm_colsum_partial': [_P, _L, _I, _I, _P, _P],
    'alm_geglu_fwd': [_P, _P, _L, _I, _P],
    'alm_geglu_bwd': [_P, _P, _P, _L, _I, _P],
    'alm_geglu_partial_blocks': [_I],
    'alm_geglu_ln_fwd': [_P, _L, _I, _P, _P, _L, _P, _P, _I, _I, _I, _P],
    'alm_geglu_ln_bwd': [_P, _L, _P, _L, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'alm_mqa_attn_fwd': [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _F, _F, _U, _P, _P],
    'alm_mqa_attn_bwd': [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _P, _L, _P, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _F, _F, _U, _P, _P],
    'alm_mqa_attn_bias_fwd': [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _F, _U, _P, _P],
    'alm_mqa_attn_bias_bwd': [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _P, _L, _P, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _F,
                              _P, _I, _P, _P, _P, _P, _P, _F, _U, _P, _P],
    'alm_attn_bias_part_rows': [_I, _I, _I],
    'alm_attn_bias_grad_reduce': [_P, _P, _I, _I, _I, _I, _F, _P],
    'alm_posmlp_in_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    'alm_posmlp_in_bwd_chunks': [_I],
    'alm_posmlp_in_bwd': [_P, _P, _P, _I, _I, _I, _P],
    'alm_silu_fwd': [_P, _P, _L, _P],
    'alm_silu_bwd': [_P, _P, _P, _L, _P],
    'alm_posmlp_out_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    'alm_posmlp_out_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    'alm_xattn_softmax_fwd': [_P, _L, _P, _P, _F, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    'alm_xattn_dbias': [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P],
    'alm_xattn_combine': [_P, _L, _P, _P, _P, _L, _L, _I, _I, _P],
    'alm_xattn_softmax_bwd': [_P, _L, _P, _L, _P, _F, _P, _L, _I, _I, _I, _I, _P],
    'alm_xattn_delta': [_P, _L, _P, _L, _P, _I, _I, _I, _I, _P],
    'alm_value_residual_mix': [_P, _L, _P, _L, _P, _L, _L, _I, _P],
    'alm_kv_grad_pack': [_P, _P, _L, _I, _L, _P, _P, _L, _L, _I, _I, _P],
    'alm_forgetful_mask': [_P, _L, _P, _L, _I, _I, _I, _P],
    'alm_coarse_prepare': [_P, _L, _P, _L, _I, _I, _I, _L, _L, _L, _I, _I, _P, _P, _P, _P, _I, _P],
    'alm_semantic_prepare': [_P, _L, _I, _I, _L, _L, _P, _P, _I, _P],
    'alm_unique_consecutive_i64': [_P, _L, _I, _I, _I, _L, _L, _P, _L, _P, _P],
    'alm_fine_prepare': [_P, _L, _P, _L, _I, _I, _I, _L, _L, _I, _I, _I, _P, _P, _P],
    'alm_loss_combine': [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _F, _F, _F, _F, _I, _L, _P, _P, _P],
    'alm_mqa_head_groups': [_I],
    'alm_mqa_bwd_parts': [_I, _I, _I],
    'alm_hc_coef_width': [_I],
    'alm_hc_partial_width': [_I, _I],
    'alm_hc_grads_width': [_I, _I],
    'alm_hc_partial_rows': [_I, _I, _I, _I, _L, _I],
    'alm_hc_fwd': [_P, _I, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'alm_hc_bwd': [_P, _I, _I, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    'alm_hc_param_grads': [_P, _I, _P, _P, _P, _P, _I, _I, _P],
    'alm_hc_param_grads_batched': [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    'alm_streams_expand': [_P, _P, _I, _I, _L, _P],
    'alm_streams_reduce': [_P, _P, _I, _I, _L, _P],
    'alm_residual_add': [_P, _P, _L, _P, _L, _I, _P],
    'alm_f32_to_bf16': [_P, _P, _P, _L, _L, _I, _P],
    'alm_add_f32': [_P, _P, _P, _L, _F, _P],
    'alm_embed_assemble': [_P, _P, _I, _P, _P, _P, _L, _I, _P, _P],
    'alm_embed_scatter_add': [_P, _P, _I, _P, _P, _P, _F, _L, _I, _P],
    'alm_embed_scatter_ws_floats': [_P, _I, _L, _I],
    'alm_embed_scatter_owned': [_P, _P, _I, _P, _P, _P, _F, _L, _I, _P, _P],
    'alm_gather_split_bf16': [_P, _L, _L, _P, _P, _P, _L, _L, _I, _P],
    'alm_gather_rows_bf16': [_P, _L, _P, _P, _L, _L, _I, _P],
    'alm_scatter_rows_bf16': [_P, _L, _P, _P, _L, _L, _I, _P],
    'alm_cross_entropy_fwd': [_P, _L, _P, _P, _P, _L, _I, _I, _P],
    'alm_cross_entropy_bwd': [_P, _L, _P, _P, _P, _P, _L, _L, _I, _I, _I, _P],
    'alm_reduce_sum': [_P, _L, _P, _F, _P],
    'alm_mqa_decode_attn': [_P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    'alm_opt_chunk_elems': [],
    'alm_opt_grad_sumsq': [_P, _I, _P, _I, _P, _P],
    'alm_opt_adam_step': [_P, _I, _P, _I, _F, _F, _F, _F, _I, _I, _P, _F, _P],
    'alm_opt_adam_pack_step': [_P, _I, _F, _F, _F, _F, _I, _I, _P, _F, _P],
    'alm_conv1d_packed_floats': [_I, _I, _I],
    'alm_conv1d_pack': [_P, _P, _I, _I, _I, _P],
    'alm_conv1d_causal': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'alm_resunit_causal': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'alm_phase_interleave': [_P, _P, _I, _I, _I, _I, _P],
    'alm_rvq_decode': [_P, _L, _P, _P, _L, _I, _I, _I, _I, _P],
    'alm_rvq_padded_codes': [_I],
    'alm_rvq_padded_dim': [_I],
    'alm_rvq_pack': [_P, _P, _P, _I, _I, _I, _P],
    'alm_rvq_encode': [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    'alm_bct_to_btc': [_P, _P, _I, _I, _I, _P],
    'alm_layernorm_bct': [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    'alm_geglu_bct': [_P, _P, _I, _I, _I, _P],
    'alm_local_attn': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    'alm_memset_zero': [_P, _L, _P],
    'alm_list_op_id': [ctypes.c_char_p],
    'alm_list_op_nargs': [_I],
    'alm_list_run': [_P, _I, _P, _P, _I, _P, _I, _P, _P],
}

_lib = None


class AlmError(RuntimeError):
    pass


def load(build_if_missing: bool = True):
    """Loads (building in-tree with hipcc if necessary) the shared library.  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and not os.environ.get('ALM_LIB_PATH'):
        from . import build as _build
        try:
            _build.build(verbose=False)
        except Exception as e:  # no hipcc on the box: fine as long as the prebuilt .so that travelled with the tree matches the sources
            if not os.path.exists(LIB_PATH):
                raise ImportError(f'libaudiolm_hip.so is missing and could not be built: {e}') from e
            if not _build._fresh(_build._digest()):
                raise ImportError(f'libaudiolm_hip.so is stale (csrc/ or include/ changed since it was built) and could not be rebuilt: {e}') from e
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"`')
    # torch FIRST: its wheel bundles its own libamdhip64 / libhsa-runtime64, and whichever copy enters the process first serves both torch
    # and this library (same SONAME).  Loaded the other way round, torch ends up on the system runtime it was not built against and every
    # launch fails with hipErrorNoDevice (seen on MI355X when the package was imported before torch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f'{LIB_PATH} does not export {name} (declared in include/audiolm_hip.h): stale or foreign library')
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


class AlmOptTensor(ctypes.Structure):
    """mirror of AlmOptTensor in include/audiolm_hip.h"""
    _fields_ = [('p', c_void_p), ('g', c_void_p), ('m', c_void_p), ('v', c_void_p), ('n', c_longlong), ('wd', c_float), ('step', c_int)]


class AlmOptPackJob(ctypes.Structure):
    """mirror of AlmOptPackJob in include/audiolm_hip.h"""
    _fields_ = [('p', c_void_p), ('g', c_void_p), ('m', c_void_p), ('v', c_void_p), ('rows', c_int), ('cols', c_int), ('ld', c_longlong),
                ('dst', c_void_p), ('ld_dst', c_longlong), ('rows_pad', c_int), ('cols_pad', c_int), ('dstT', c_void_p), ('ld_dstT', c_longlong),
                ('wd', c_float), ('step', c_int)]


class AlmListEntry(ctypes.Structure):
    """mirror of AlmListEntry in include/audiolm_hip.h"""
    _fields_ = [('op', c_int), ('nargs', c_int), ('first', c_int), ('reserved', c_int)]


class AlmPackJob(ctypes.Structure):
    """mirror of AlmPackJob in include/audiolm_hip.h"""
    _fields_ = [('src', c_void_p), ('rows', c_int), ('cols', c_int), ('ld_src', c_longlong), ('dst', c_void_p), ('ld_dst', c_longlong),
                ('rows_pad', c_int), ('cols_pad', c_int), ('dstT', c_void_p), ('ld_dstT', c_longlong)]


_BOUND = {}          # name -> bound ctypes function (one dict lookup per call instead of load() + getattr on the CDLL: ~150 calls per training step)
RECORDER = None      # launchlist.Recorder while a stack pass is being recorded (the launches still run: recording is an ordinary step that is also written down)


def call(name: str, *args):
    fn = _BOUND.get(name)
    if fn is None:
        fn = _BOUND[name] = getattr(load(), name)
    if RECORDER is not None:
        RECORDER.note(name, args)
    rc = fn(*args)
    if rc != 0:
        raise AlmError(f'{name} failed with code {rc}' + (' (ALM_ERR_BAD_ARG)' if rc == 10001 else ' (ALM_ERR_UNSUPPORTED)' if rc == 10002 else ' (hipError_t)'))
    return rc


def query(name: str, *args) -> int:
    """For the int-returning size queries (alm_*_blocks / alm_*_width)."""
    fn = _BOUND.get(name)
    if fn is None:
        fn = _BOUND[name] = getattr(load(), name)
    return fn(*args)
