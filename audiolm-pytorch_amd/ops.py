"""Thin tensor-level wrappers over the C ABI (one Python function per `alm_*` entry).  PyTorch only supplies device
memory and the stream; all arithmetic happens in libaudiolm_hip.so.  No CPU / eager fallback exists on purpose.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib

BF16, F32 = torch.bfloat16, torch.float32


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
# current device as ONE C call (torch.cuda.current_device() is a Python wrapper with a lazy-init check: ~0.9 us, and _chk / _st run ~600 times per step)
_cur_dev = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device


def _st():
    """hipStream_t of torch's CURRENT stream on the current device (every alm_* launch goes there).  The raw-handle query is one C call; the
    public torch.cuda.current_stream() builds a Stream object per call (~8 us: ~1.5 ms of host time per training step at ~180 launches)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=None):
    if not t.is_cuda:
        raise _lib.AlmError('audiolm_pytorch_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback')
    if dtype is not None and t.dtype != dtype:
        raise _lib.AlmError(f'expected {dtype}, got {t.dtype}')
    if t.device.index != _cur_dev():
        # every launch goes to the CURRENT device's stream (_st): a tensor of another device would be dereferenced there
        raise _lib.AlmError(f'tensor on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: call '
                            'torch.cuda.set_device(...) (one process per GPU) before using audiolm_pytorch_amd')
    return t


def _p(t):
    return None if t is None else t.data_ptr()


# Every device buffer an op wrapper creates goes through _new / _new_zeros / _new_like.  ALLOC is None in ordinary use (torch's caching allocator); while
# launchlist.py sizes or records a stack pass it is an allocator object (a byte tally / a bump allocator over the pass's arena), so that every temporary of
# the pass sits at a fixed offset of ONE buffer and the recorded launch list can be re-issued against a fresh arena.
ALLOC = None


def _new(shape, dtype=None, device=None):
    if ALLOC is not None:
        return ALLOC.empty(shape, dtype, device)
    return torch.empty(shape, dtype=dtype, device=device)


def _new_zeros(shape, dtype=None, device=None):
    if ALLOC is not None:
        return ALLOC.zeros(shape, dtype, device)
    return torch.zeros(shape, dtype=dtype, device=device)


def _new_like(t):
    if ALLOC is not None and t.is_contiguous():
        return ALLOC.empty(t.shape, t.dtype, t.device)
    return torch.empty_like(t)


def memset_zero(t):
    """zero-fill of a contiguous tensor on the current stream by a kernel (C ABI: alm_memset_zero; not hipMemsetAsync: its small graph nodes replay wrongly)"""
    assert t.is_contiguous()
    _lib.call('alm_memset_zero', t.data_ptr(), t.numel() * t.element_size(), _st())
    return t


def persistent_buffers():
    """device buffers that live at a fixed address for the life of the process (a recorded launch list may carry their addresses as literals)"""
    return list(_NT_WS.values())


def _rows_ld(t):
    """2-D row-major view -> (rows, cols, ld)."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.shape[0], t.shape[1], t.stride(0)


# ------------------------------------------------------------------------------------------------ dense contractions

# Workspace of the in-launch split-K NT launches (alm_gemm_bf16_nt_ws): one persistent buffer per (device, stream) -- a launch owns its slabs and ticket
# counters until it retires, and launches on ONE stream retire in order.  Zeroed once (the counters; every launch leaves them zero).  64 MB + 4 KB.
_NT_WS = {}
NT_WS = os.environ.get('ALM_GEMM_NT_WS', '1') != '0'           # A/B switch: 0 = never hand a workspace over (the round-5 launches)


def _nt_ws(dev, stream):
    key = (dev.index, stream)
    ws = _NT_WS.get(key)
    if ws is None:
        ws = _NT_WS[key] = torch.zeros(_lib.query('alm_gemm_nt_ws_bytes'), dtype=torch.uint8, device=dev)
    return ws


def gemm_nt(A, B, C, *, bias=None, alpha=1.0, accumulate=False):
    """C[..., M, N] (+)= alpha * A[..., M, K] @ B[..., N, K]^T (+ bias).  Up to two leading batch dims (broadcast via stride 0)."""
    _chk(A, BF16), _chk(B, BF16), _chk(C)
    assert A.stride(-1) == 1 and B.stride(-1) == 1 and C.stride(-1) == 1
    nb = A.dim() - 2
    assert nb in (0, 1, 2) and B.dim() == A.dim() and C.dim() == A.dim()
    M, K = A.shape[-2:]
    N = B.shape[-2]
    assert B.shape[-1] == K and C.shape[-2] == M and C.shape[-1] == N, (A.shape, B.shape, C.shape)
    bs = [1, 1]
    sa, sb, sc = [0, 0], [0, 0], [0, 0]
    for i in range(nb):
        j = 2 - nb + i
        bs[j] = A.shape[i]
        sa[j], sb[j], sc[j] = A.stride(i), B.stride(i) if B.shape[i] != 1 else 0, C.stride(i)
    st = _st()
    if NT_WS and M >= 256 and N >= 256:
        ws = _nt_ws(A.device, st)
        _lib.call('alm_gemm_bf16_nt_ws', A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(bias), M, N, K, A.stride(-2), B.stride(-2), C.stride(-2),
                  bs[0], bs[1], sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], float(alpha), int(C.dtype == F32), int(accumulate), ws.data_ptr(), ws.numel(), st)
    else:
        _lib.call('alm_gemm_bf16_nt', A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(bias), M, N, K, A.stride(-2), B.stride(-2), C.stride(-2),
                  bs[0], bs[1], sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], float(alpha), int(C.dtype == F32), int(accumulate), st)
    return C


def gemm_nt_inl(A, B, C, slices, *, bias=None, alpha=1.0, accumulate=False):
    """un-batched gemm_nt on the staggered 256 x 256 tile with a GIVEN number of in-launch K slices (alm_gemm_bf16_nt_inl): tests / benchmarks"""
    _chk(A, BF16), _chk(B, BF16), _chk(C)
    M, K = A.shape
    N = B.shape[0]
    st = _st()
    ws = _nt_ws(A.device, st)
    _lib.call('alm_gemm_bf16_nt_inl', A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(bias), M, N, K, A.stride(0), B.stride(0), C.stride(0),
              float(alpha), int(C.dtype == F32), int(accumulate), int(slices), ws.data_ptr(), ws.numel(), st)
    return C


def gemm_nt_group2(A0, B0, C0, A1, B1, C1):
    """C0 = A0 @ B0^T and C1 = A1 @ B1^T (2-D, last dim contiguous, both outputs of one dtype) in ONE launch when the two problems pick the same block
    tile (alm_gemm_bf16_nt_group2; two launches otherwise -- identical results)."""
    for t in (A0, B0, A1, B1):
        _chk(t, BF16)
        assert t.dim() == 2 and t.stride(1) == 1
    _chk(C0), _chk(C1)
    assert C0.dtype == C1.dtype and C0.stride(1) == 1 and C1.stride(1) == 1
    (M0, K0), (M1, K1), N0, N1 = A0.shape, A1.shape, B0.shape[0], B1.shape[0]
    assert B0.shape[1] == K0 and B1.shape[1] == K1 and tuple(C0.shape) == (M0, N0) and tuple(C1.shape) == (M1, N1)
    _lib.call('alm_gemm_bf16_nt_group2', A0.data_ptr(), B0.data_ptr(), C0.data_ptr(), M0, N0, K0, A0.stride(0), B0.stride(0), C0.stride(0),
              A1.data_ptr(), B1.data_ptr(), C1.data_ptr(), M1, N1, K1, A1.stride(0), B1.stride(0), C1.stride(0), int(C0.dtype == F32), _st())
    return C0, C1


def _splitk(name, A, B, C, M, N, K, nb, sA, sB, sC, alpha, accumulate):
    nws = _lib.query('alm_gemm_splitk_ws_floats', M, N, K, nb)
    if nws < 0:
        raise _lib.AlmError('split-K workspace exceeds 2^31 floats')
    ws = _new(nws, dtype=F32, device=A.device) if nws > 0 else None
    _lib.call(name, A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(ws), M, N, K, A.stride(-2), B.stride(-2), C.stride(-2), nb, sA, sB, sC,
              float(alpha), int(accumulate), _st())
    return C


def gemm_nt_splitk(A, B, C, *, alpha=1.0, accumulate=False):
    """fp32 C[M, N] (+)= alpha * A[M, K] @ B[N, K]^T with K split over workgroups (long-K, few-tile contractions)."""
    _chk(A, BF16), _chk(B, BF16), _chk(C, F32)
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and C.shape == (M, N) and A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1
    return _splitk('alm_gemm_bf16_nt_splitk', A, B, C, M, N, K, 1, 0, 0, 0, alpha, accumulate)


def gemm_tn_splitk(At, Bt, C, *, alpha=1.0, accumulate=False):
    """fp32 C[(nb,) M, N] (+)= alpha * At[K, (nb,) M]^T @ Bt[K, N]: the weight-gradient contraction over K = tokens, read straight from
    the row-major activations (LDS transpose reads inside the kernel; no transposed copies).  A 3-D `At` of shape [nb, K, M] (a strided
    view, e.g. the x / gate halves of dU) with a 3-D C [nb, M, N] runs the nb problems in one launch against the same Bt."""
    _chk(At, BF16), _chk(Bt, BF16), _chk(C, F32)
    nb, sA, sC = 1, 0, 0
    if At.dim() == 3:
        nb, sA, sC = At.shape[0], At.stride(0), C.stride(0)
        assert C.dim() == 3 and C.shape[0] == nb
    K, M = At.shape[-2:]
    N = Bt.shape[1]
    assert Bt.dim() == 2 and Bt.shape[0] == K and C.shape[-2:] == (M, N) and At.stride(-1) == 1 and Bt.stride(1) == 1 and C.stride(-1) == 1
    return _splitk('alm_gemm_bf16_tn_splitk', At, Bt, C, M, N, K, nb, sA, 0, sC, alpha, accumulate)


def gemm_tn_batched(At, Bt, C, *, alpha=1.0, accumulate=False):
    """fp32 C[n1, n2, M, N] (+)= alpha * At[n1, n2, K, M]^T @ Bt[n1, n2, K, N]: 4-D strided views (last dim contiguous; a size-1 / stride-0 leading dim of Bt
    broadcasts), e.g. every layer's gradient of one weight kind from stacked activation buffers in ONE launch."""
    _chk(At, BF16), _chk(Bt, BF16), _chk(C, F32)
    assert At.dim() == 4 and Bt.dim() == 4 and C.dim() == 4 and At.stride(-1) == 1 and Bt.stride(-1) == 1 and C.stride(-1) == 1
    n1, n2, K, M = At.shape
    N = Bt.shape[-1]
    assert Bt.shape[-2] == K and tuple(C.shape) == (n1, n2, M, N), (At.shape, Bt.shape, C.shape)
    sb = [Bt.stride(0) if Bt.shape[0] != 1 else 0, Bt.stride(1) if Bt.shape[1] != 1 else 0]
    nws = _lib.query('alm_gemm_splitk_ws_floats', M, N, K, n1 * n2)
    if nws < 0:
        raise _lib.AlmError('split-K workspace exceeds 2^31 floats')
    ws = _new(nws, dtype=F32, device=At.device) if nws > 0 else None
    _lib.call('alm_gemm_bf16_tn_batched', At.data_ptr(), Bt.data_ptr(), C.data_ptr(), _p(ws), M, N, K, At.stride(-2), Bt.stride(-2), C.stride(-2), n1, n2,
              At.stride(0), At.stride(1), sb[0], sb[1], C.stride(0), C.stride(1), float(alpha), int(accumulate), _st())
    return C


def gemm_nt_tile(A, B, C, tile, *, bias=None, alpha=1.0, accumulate=False):
    """un-batched gemm_nt with an explicit tile configuration (0 auto, 1 = 128x128, 2 = 256x256): tuning / benchmarks."""
    _chk(A, BF16), _chk(B, BF16), _chk(C)
    M, K = A.shape
    N = B.shape[0]
    _lib.call('alm_gemm_bf16_nt_tile', A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(bias), M, N, K, A.stride(0), B.stride(0), C.stride(0),
              float(alpha), int(C.dtype == F32), int(accumulate), int(tile), _st())
    return C


def transpose(src, rows_pad=None):
    """bf16 [rows, cols] -> new bf16 [cols, rows_pad] with zero-filled pad columns (rows_pad = rows rounded up to 8)."""
    _chk(src, BF16)
    rows, cols, ld = _rows_ld(src)
    rp = rows_pad or ((rows + 7) // 8 * 8)
    dst = _new((cols, rp), dtype=BF16, device=src.device)
    _lib.call('alm_transpose_bf16', src.data_ptr(), dst.data_ptr(), rows, cols, ld, rp, rp, _st())
    return dst


def pack_weight(w, dst=None, dstT=None, rows_pad=None, cols_pad=None):
    """fp32 [rows, cols] -> bf16 dst[rows_pad, >= cols_pad] and / or bf16 dstT[cols_pad, >= rows_pad], zero padded."""
    _chk(w, F32)
    rows, cols, ld = _rows_ld(w)
    rp, cp = rows_pad or rows, cols_pad or cols
    _lib.call('alm_pack_weight', w.data_ptr(), rows, cols, ld, _p(dst), dst.stride(0) if dst is not None else 0, rp, cp,
              _p(dstT), dstT.stride(0) if dstT is not None else 0, _st())


def pack_weights_multi(jobs):
    """jobs: list of (w fp32 [rows, cols] view, dst | None, dstT | None, rows_pad, cols_pad): any number of weights, 40 per launch (8 on the narrow fallback kernels)."""
    arr = (_lib.AlmPackJob * len(jobs))()
    for i, (w, dst, dstT, rp, cp) in enumerate(jobs):
        _chk(w, F32)
        rows, cols, ld = _rows_ld(w)
        arr[i] = _lib.AlmPackJob(w.data_ptr(), rows, cols, ld, _p(dst), dst.stride(0) if dst is not None else 0, rp, cp,
                                 _p(dstT), dstT.stride(0) if dstT is not None else 0)
    _lib.call('alm_pack_weights_multi', ctypes.cast(arr, ctypes.c_void_p), len(jobs), _st())


# ------------------------------------------------------------------------------------------------ LayerNorm / GEGLU

def layernorm_fwd(x, gamma, *, want_copy=False, out_f32=False, y_out=None, xc_out=None):
    """x [rows, D] fp32|bf16 -> (y bf16 (fp32 with out_f32: the final LayerNorm feeding the logit heads), xcopy bf16|None, mean, rstd)."""
    _chk(x)
    rows, D, ld = _rows_ld(x)
    y = _new((rows, D), dtype=F32 if out_f32 else BF16, device=x.device) if y_out is None else y_out
    xc = (_new((rows, D), dtype=BF16, device=x.device) if xc_out is None else xc_out) if want_copy else None
    assert y.shape == (rows, D) and y.is_contiguous() and (xc is None or (xc.shape == (rows, D) and xc.is_contiguous()))
    mean = _new(rows, dtype=F32, device=x.device)
    rstd = _new(rows, dtype=F32, device=x.device)
    _lib.call('alm_layernorm_fwd', x.data_ptr(), int(x.dtype == BF16), ld, gamma.data_ptr(), y.data_ptr(), int(out_f32), D, _p(xc), D, mean.data_ptr(),
              rstd.data_ptr(), rows, D, _st())
    return y, xc, mean, rstd


def colsum(inp, out=None, scale=1.0, accumulate=False):
    _chk(inp)
    rows, cols, ld = _rows_ld(inp)
    if out is None:
        out = _new(cols, dtype=F32, device=inp.device)
    chunks = _lib.query('alm_colsum_chunks', rows)
    ws = _new((chunks, cols), dtype=F32, device=inp.device) if chunks > 1 else None
    _lib.call('alm_colsum', inp.data_ptr(), int(inp.dtype == BF16), ld, rows, cols, out.data_ptr(), float(scale), int(accumulate), _p(ws), _st())
    return out


def layernorm_bwd(dy, x, mean, rstd, gamma, *, extra=None, dx_dtype=F32, want_dgamma=True):
    """dy bf16 (or fp32: the gradient arriving from the logit heads) -> (dx [rows, D] dx_dtype, dgamma [D] fp32 | None)."""
    _chk(dy), _chk(x)
    assert dy.dtype in (BF16, F32)
    rows, D, lddy = _rows_ld(dy)
    dx = _new((rows, D), dtype=dx_dtype, device=dy.device)
    part = None
    if want_dgamma:
        nblk = _lib.query('alm_ln_partial_blocks', rows)
        part = _new((nblk, D), dtype=F32, device=dy.device)
    _lib.call('alm_layernorm_bwd', dy.data_ptr(), int(dy.dtype == F32), lddy, x.data_ptr(), int(x.dtype == BF16), x.stride(0), mean.data_ptr(), rstd.data_ptr(),
              gamma.data_ptr(), _p(extra), extra.stride(0) if extra is not None else 0, dx.data_ptr(), int(dx_dtype == BF16), D, _p(part),
              rows, D, _st())
    return dx, (colsum(part) if want_dgamma else None)


def geglu_fwd(x):
    """fp32 [rows, 2 I] -> [rows, I]: x_half * gelu(gate_half) (the standalone GEGLU module)"""
    _chk(x, F32)
    rows, two_i, ld = _rows_ld(x)
    assert ld == two_i and two_i % 2 == 0
    y = _new((rows, two_i // 2), dtype=F32, device=x.device)
    _lib.call('alm_geglu_fwd', x.data_ptr(), y.data_ptr(), rows, two_i // 2, _st())
    return y


def geglu_bwd(dy, x):
    _chk(dy, F32), _chk(x, F32)
    rows, two_i, ld = _rows_ld(x)
    assert ld == two_i and dy.shape == (rows, two_i // 2) and dy.is_contiguous()
    dx = _new_like(x)
    _lib.call('alm_geglu_bwd', dy.data_ptr(), x.data_ptr(), dx.data_ptr(), rows, two_i // 2, _st())
    return dx


def geglu_ln_fwd(u, gamma, inner, inner_pad, out=None):
    """u bf16 [rows, 2 * inner_pad] (x | gate halves) -> (hn bf16 [rows, inner_pad] (written into `out` when given), mean, rstd)."""
    _chk(u, BF16)
    rows, _, ldu = _rows_ld(u)
    if out is None:
        out = _new((rows, inner_pad), dtype=BF16, device=u.device)
    assert out.shape == (rows, inner_pad) and out.dtype == BF16 and out.stride(1) == 1 and out.stride(0) == inner_pad
    mean = _new(rows, dtype=F32, device=u.device)
    rstd = _new(rows, dtype=F32, device=u.device)
    _lib.call('alm_geglu_ln_fwd', u.data_ptr(), ldu, inner_pad, gamma.data_ptr(), out.data_ptr(), inner_pad, mean.data_ptr(), rstd.data_ptr(),
              rows, inner, inner_pad, _st())
    return out, mean, rstd


def geglu_ln_bwd(dhn, u, gamma, mean, rstd, inner, inner_pad, du_out=None):
    """-> (du bf16 [rows, 2 * inner_pad] (written into `du_out` when given: same shape and row stride as u), dgamma [inner])."""
    rows, _, ldu = _rows_ld(u)
    du = _new_like(u) if du_out is None else du_out
    assert du.shape == u.shape and du.stride(0) == ldu and du.stride(1) == 1 and du.dtype == BF16
    nblk = _lib.query('alm_geglu_partial_blocks', rows)
    part = _new((nblk, inner), dtype=F32, device=u.device)
    _lib.call('alm_geglu_ln_bwd', dhn.data_ptr(), dhn.stride(0), u.data_ptr(), ldu, inner_pad, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
              du.data_ptr(), part.data_ptr(), rows, inner, inner_pad, _st())
    return du, colsum(part)


# ------------------------------------------------------------------------------------------------ attention

def _bias_args(bias, N, H):
    """bias: object with tbl fp32 [H, LT] and int32 [N] qkey4 / kkey4 / qattr / kattr (relpos.AttnBias)."""
    tbl = _chk(bias.tbl, F32)
    assert tbl.dim() == 2 and tbl.shape[0] == H and tbl.is_contiguous(), tbl.shape
    for t in (bias.qkey4, bias.kkey4, bias.qattr, bias.kattr):
        _chk(t, torch.int32)
        assert t.shape == (N,) and t.is_contiguous(), (t.shape, N)
    return (tbl.data_ptr(), tbl.shape[1], bias.qkey4.data_ptr(), bias.kkey4.data_ptr(), bias.qattr.data_ptr(), bias.kattr.data_ptr())


def attn_bias_part(B, N, H, LT, device):
    """zeroed per-workgroup partial tables the biased attention backward accumulates the table gradient into"""
    return _new_zeros((_lib.query('alm_attn_bias_part_rows', B, N, H), LT), dtype=F32, device=device)


def attn_bias_grad_reduce(part, B, N, H, dim_head=64):
    LT = part.shape[1]
    dtbl = _new((H, LT), dtype=F32, device=part.device)
    _lib.call('alm_attn_bias_grad_reduce', part.data_ptr(), dtbl.data_ptr(), B, N, H, LT, float(dim_head) ** -0.5, _st())
    return dtbl


def mqa_attn_fwd(q, k, v, mask, B, N, H, dim_head=64, bias=None, dropout_p=0., seed=0, o=None, seed_dev=None):
    """q bf16 [B*N, H*dh]; k, v bf16 [B*N, dh] views (row stride arbitrary); mask uint8 [B, N] | None -> (o, lse).
    bias: structured score bias (see alm_mqa_attn_bias_fwd) or None.  dropout_p > 0: attention dropout with the mask stream `seed` (the
    backward must get the same pair); seed_dev (int64 [1] device tensor | None): the stream is seed + seed_dev[0], read when the kernel runs."""
    _chk(q, BF16), _chk(k, BF16), _chk(v, BF16)
    if o is None:
        o = _new((B * N, H * dim_head), dtype=BF16, device=q.device)
    assert o.shape == (B * N, H * dim_head) and o.dtype == BF16 and o.stride(1) == 1
    lse = _new((B, H, N), dtype=F32, device=q.device)
    if bias is not None:
        _lib.call('alm_mqa_attn_bias_fwd', q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), _p(mask), o.data_ptr(),
                  o.stride(0), lse.data_ptr(), B, N, H, dim_head, float(dim_head) ** -0.5, *_bias_args(bias, N, H), float(dropout_p), int(seed), _p(seed_dev), _st())
        return o, lse
    _lib.call('alm_mqa_attn_fwd', q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), _p(mask), o.data_ptr(),
              o.stride(0), lse.data_ptr(), B, N, H, dim_head, float(dim_head) ** -0.5, float(dropout_p), int(seed), _p(seed_dev), _st())
    return o, lse


def mqa_attn_bwd(q, k, v, mask, o, lse, dout, B, N, H, dim_head=64, bias=None, dtbl_part=None, dropout_p=0., seed=0, dq_out=None, seed_dev=None):
    """-> (dq bf16 [B*N, H*dh], dkv fp32 [HG, B*N, 2*dh] = per-head-group partials of (dk | dv); kv_grad_pack sums them).
    With `bias`, the table gradient is accumulated into dtbl_part (attn_bias_part)."""
    _chk(dout, BF16)
    dq = _new_like(q) if dq_out is None else dq_out
    assert dq.shape == q.shape and dq.dtype == BF16 and dq.stride(1) == 1
    hg = _lib.query('alm_mqa_bwd_parts', B, N, H)                  # dk / dv partial sets the dK/dV kernel writes for this shape (4 or 2 heads per workgroup)
    dkv = _new((hg, B * N, 2 * dim_head), dtype=F32, device=q.device)
    delta = _new((2, B, H, N), dtype=F32, device=q.device)
    if bias is not None:
        assert dtbl_part is not None and dtbl_part.shape[1] == bias.tbl.shape[1] and dtbl_part.is_contiguous()
        _lib.call('alm_mqa_attn_bias_bwd', q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), _p(mask),
                  o.data_ptr(), o.stride(0), lse.data_ptr(), dout.data_ptr(), dout.stride(0), dq.data_ptr(), dq.stride(0), dkv.data_ptr(),
                  dkv.data_ptr() + 4 * dim_head, dkv.stride(1), dkv.stride(0), delta.data_ptr(), B, N, H, dim_head, float(dim_head) ** -0.5,
                  *_bias_args(bias, N, H), dtbl_part.data_ptr(), float(dropout_p), int(seed), _p(seed_dev), _st())
        return dq, dkv
    _lib.call('alm_mqa_attn_bwd', q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), _p(mask), o.data_ptr(),
              o.stride(0), lse.data_ptr(), dout.data_ptr(), dout.stride(0), dq.data_ptr(), dq.stride(0), dkv.data_ptr(),
              dkv.data_ptr() + 4 * dim_head, dkv.stride(1), dkv.stride(0), delta.data_ptr(), B, N, H, dim_head, float(dim_head) ** -0.5,
              float(dropout_p), int(seed), _p(seed_dev), _st())
    return dq, dkv


def mqa_decode_attn(q, cache, kv_new, pos, mask, H, dim_head=64, bias=None, pos_dev=None):
    """One new position per sequence.  q bf16 [B, H*dh]; cache bf16 [B, Nmax, 2*dh] (k | v); kv_new bf16 [B, 2*dh] (appended at `pos` by the
    kernel); mask uint8 [B, >= pos+1] | None; bias: relpos.AttnBias whose index vectors cover position `pos` -> out bf16 [B, H*dh].
    pos_dev (int32 device scalar): the kernel reads the position from it instead (hipGraph replay)."""
    _chk(q, BF16), _chk(cache, BF16), _chk(kv_new, BF16)
    B, Nmax = cache.shape[0], cache.shape[1]
    assert cache.is_contiguous() and cache.shape[2] == 2 * dim_head and kv_new.shape == (B, 2 * dim_head) and kv_new.stride(1) == 1
    out = _new((B, H * dim_head), dtype=BF16, device=q.device)
    qk = qa = 0
    tb = (None, 0)
    vecs = (None, None, None, None)
    if bias is not None:
        tbl = _chk(bias.tbl, F32)
        assert tbl.is_contiguous() and tbl.shape[0] == H and bias.kkey4.shape[0] >= Nmax
        tb = (tbl.data_ptr(), tbl.shape[1])
        if pos_dev is None:
            qk, qa = int(bias.qkey4_host[pos]), int(bias.qattr_host[pos])
        vecs = (bias.kkey4.data_ptr(), bias.kattr.data_ptr(), bias.qkey4.data_ptr(), bias.qattr.data_ptr())
    _lib.call('alm_mqa_decode_attn', q.data_ptr(), q.stride(0), cache.data_ptr(), cache.stride(0), kv_new.data_ptr(), kv_new.stride(0), _p(mask),
              mask.stride(0) if mask is not None else 0, out.data_ptr(), out.stride(0), B, H, dim_head, int(pos), Nmax, float(dim_head) ** -0.5,
              tb[0], tb[1], qk, qa, vecs[0], vecs[1], _p(pos_dev), vecs[2] if pos_dev is not None else None,
              vecs[3] if pos_dev is not None else None, _st())
    return out


def value_residual_mix(v, v0):
    rows, dh, _ = _rows_ld(v)
    out = _new((rows, dh), dtype=BF16, device=v.device)
    _lib.call('alm_value_residual_mix', v.data_ptr(), v.stride(0), v0.data_ptr(), v0.stride(0), out.data_ptr(), dh, rows, dh, _st())
    return out


def loss_combine(sums, labels, weights, ignore_index=-1):
    """sums: G <= 4 fp32 scalars (per-group cross-entropy SUMS), labels: G int64 tensors, weights: G floats ->
    (loss fp32 [1] = sum_g w_g * sum_g / max(#(labels_g != ignore), 1), scales fp32 [G] = w_g / max(count_g, 1))."""
    G = len(sums)
    assert 1 <= G <= 4 and len(labels) == G and len(weights) == G
    dev = sums[0].device
    labels = [l if l.is_contiguous() else l.contiguous() for l in labels]
    for t, l in zip(sums, labels):
        _chk(t, F32)
        _chk(l, torch.int64)
    loss = _new(1, dtype=F32, device=dev)
    scales = _new(G, dtype=F32, device=dev)
    pad = [None] * (4 - G)
    _lib.call('alm_loss_combine', *[t.data_ptr() for t in sums], *pad, *[l.data_ptr() for l in labels], *pad, *[l.numel() for l in labels], *([0] * (4 - G)),
              *[float(w) for w in weights], *([0.0] * (4 - G)), G, int(ignore_index), loss.data_ptr(), scales.data_ptr(), _st())
    return loss, scales


def coarse_prepare(sem, coarse, pad_id, sem_eos, coarse_eos, Q, C, sem_has_eos=False):
    """sem int64 [B, ns0], coarse int64 [B, nc0] (flattened (n q)) -> (sem_labels [B, ns], coarse_labels [B, nc0+1], src_a int32 [B, N],
    keep bool [B, N]) with ns = ns0 + 1, N = ns + nc0 + 2: CoarseTransformerWrapper's training-step id bookkeeping (C ABI: alm_coarse_prepare).
    sem_has_eos: the rows of `sem` already are [ids | eos | pad ...] (unique_consecutive): ns = ns0."""
    _chk(sem, torch.int64), _chk(coarse, torch.int64)
    assert sem.dim() == 2 and coarse.dim() == 2 and sem.stride(1) == 1 and coarse.stride(1) == 1 and sem.shape[0] == coarse.shape[0]
    B, ns0 = sem.shape
    nc0 = coarse.shape[1]
    ns = ns0 + (0 if sem_has_eos else 1)
    N, dev = ns + nc0 + 2, sem.device
    sl = _new((B, ns), dtype=torch.int64, device=dev)
    cl = _new((B, nc0 + 1), dtype=torch.int64, device=dev)
    src_a = _new((B, N), dtype=torch.int32, device=dev)
    keep = _new((B, N), dtype=torch.bool, device=dev)
    _lib.call('alm_coarse_prepare', sem.data_ptr(), sem.stride(0), coarse.data_ptr(), coarse.stride(0), B, ns0, nc0, int(pad_id), int(sem_eos), int(coarse_eos),
              int(Q), int(C), sl.data_ptr(), cl.data_ptr(), src_a.data_ptr(), keep.data_ptr(), 1 if sem_has_eos else 0, _st())
    return sl, cl, src_a, keep


def semantic_prepare(sem, eos_id, num_rows, has_eos=False):
    """sem int64 [B, n0] -> (labels int64 [B, n0 + 1] = [ids | eos], src_a int32 [B, n0 + 1] = [start token | ids]): SemanticTransformerWrapper's training-step
    id bookkeeping + SemanticTransformer's embedding source codes (C ABI: alm_semantic_prepare).  has_eos: the rows already are [ids | eos | pad ...]
    (unique_consecutive): labels [B, n0] = the rows, src_a [B, n0] = [start token | rows without their last column]."""
    _chk(sem, torch.int64)
    assert sem.dim() == 2 and (sem.shape[1] == 0 or sem.stride(1) == 1)
    B, n0 = sem.shape
    W = n0 + (0 if has_eos else 1)
    labels = _new((B, W), dtype=torch.int64, device=sem.device)
    src_a = _new((B, W), dtype=torch.int32, device=sem.device)
    _lib.call('alm_semantic_prepare', sem.data_ptr(), sem.stride(0), B, n0, int(eos_id), int(num_rows), labels.data_ptr(), src_a.data_ptr(), 1 if has_eos else 0, _st())
    return labels, src_a


def unique_consecutive(ids, eos_id=None, pad=-1):
    """ids int64 [B, n] -> (out int64 [B, n + e], lengths int32 [B]): every row with its runs of equal ids collapsed, followed by `pad`; e = 1 and the rows
    are [ids | eos_id] when eos_id is given (C ABI: alm_unique_consecutive_i64; reference batch_unique_consecutive, audiolm_pytorch.py:162-164, after
    append_eos_id).  The caller reads `lengths` once and slices out[:, :max]."""
    _chk(ids, torch.int64)
    assert ids.dim() == 2 and (ids.shape[1] == 0 or ids.stride(1) == 1)
    B, n = ids.shape
    W = n + (0 if eos_id is None else 1)
    out = _new((B, W), dtype=torch.int64, device=ids.device)
    lengths = _new((B,), dtype=torch.int32, device=ids.device)
    if B:
        _lib.call('alm_unique_consecutive_i64', ids.data_ptr(), ids.stride(0), B, n, 0 if eos_id is None else 1, 0 if eos_id is None else int(eos_id), int(pad),
                  out.data_ptr(), W, lengths.data_ptr(), _st())
    return out, lengths


def fine_prepare(coarse, fine, nf, pad_id, eos_id, Qc, Qf, C):
    """coarse int64 [B, n], fine int64 [B, >= nf] (flattened (n q)) -> (src_a int32 [B, N], keep bool [B, N]), N = n + nf + 2: FineTransformer.forward's
    id bookkeeping (C ABI: alm_fine_prepare)."""
    _chk(coarse, torch.int64), _chk(fine, torch.int64)
    assert coarse.dim() == 2 and fine.dim() == 2 and coarse.stride(1) == 1 and fine.stride(1) == 1 and coarse.shape[0] == fine.shape[0] and fine.shape[1] >= nf
    B, n = coarse.shape
    N, dev = n + nf + 2, coarse.device
    src_a = _new((B, N), dtype=torch.int32, device=dev)
    keep = _new((B, N), dtype=torch.bool, device=dev)
    _lib.call('alm_fine_prepare', coarse.data_ptr(), coarse.stride(0), fine.data_ptr(), fine.stride(0), B, n, nf, int(pad_id), int(eos_id), int(Qc), int(Qf),
              int(C), src_a.data_ptr(), keep.data_ptr(), _st())
    return src_a, keep


FORGETFUL_MAX_N = 16384


def forgetful_mask_(keep, score, drop):
    """keep (bool [B, N]) &= NOT(one of the `drop` largest entries of its row of score (fp32 [B, N])); column 0 is never dropped (audiolm_pytorch.py:82-89)."""
    _chk(score, F32)
    assert keep.dtype == torch.bool and keep.dim() == 2 and score.shape == keep.shape and keep.stride(1) == 1 and score.stride(1) == 1
    B, N = keep.shape
    _lib.call('alm_forgetful_mask', score.data_ptr(), score.stride(0), keep.data_ptr(), keep.stride(0), B, N, int(drop), _st())
    return keep


def kv_grad_pack(dkv_f32, acc_v0, mode, dim_head=64, out=None):
    """dkv_f32: [rows, 2*dh] or per-head-group partials [HG, rows, 2*dh] (summed here) -> bf16 [rows, 2*dh]."""
    if dkv_f32.dim() == 2:
        dkv_f32 = dkv_f32.unsqueeze(0)
    nparts, rows = dkv_f32.shape[0], dkv_f32.shape[1]
    if out is None:
        out = _new((rows, 2 * dim_head), dtype=BF16, device=dkv_f32.device)
    assert out.shape == (rows, 2 * dim_head) and out.dtype == BF16 and out.stride(1) == 1
    _lib.call('alm_kv_grad_pack', dkv_f32.data_ptr(), dkv_f32.data_ptr() + 4 * dim_head, dkv_f32.stride(1), nparts, dkv_f32.stride(0),
              _p(acc_v0), out.data_ptr(), out.stride(0), rows, dim_head, mode, _st())
    return out


# ------------------------------------------------------------------------------------------------ position-bias table MLPs

def posmlp_in_fwd(x, W, b):
    """x fp32 [L, in]; W fp32 [C, in]; b [C] -> (pre fp32 [L, C], act bf16 [L, C] = silu(pre))."""
    _chk(x, F32), _chk(W, F32), _chk(b, F32)
    L, ind = x.shape
    C = W.shape[0]
    assert W.shape == (C, ind) and x.is_contiguous() and W.is_contiguous()
    pre = _new((L, C), dtype=F32, device=x.device)
    act = _new((L, C), dtype=BF16, device=x.device)
    _lib.call('alm_posmlp_in_fwd', x.data_ptr(), W.data_ptr(), b.data_ptr(), pre.data_ptr(), act.data_ptr(), L, ind, C, _st())
    return pre, act


def posmlp_in_bwd(dpre, x):
    """-> (dW fp32 [C, in], db [C])"""
    _chk(dpre, BF16), _chk(x, F32)
    L, C = dpre.shape
    ind = x.shape[1]
    chunks = _lib.query('alm_posmlp_in_bwd_chunks', L)
    part = _new((chunks, (1 + ind) * C), dtype=F32, device=x.device)
    _lib.call('alm_posmlp_in_bwd', dpre.data_ptr(), x.data_ptr(), part.data_ptr(), L, ind, C, _st())
    s = colsum(part)
    return s[C:].view(ind, C).t().contiguous(), s[:C]


def silu_fwd(pre):
    _chk(pre, F32)
    act = _new(pre.shape, dtype=BF16, device=pre.device)
    _lib.call('alm_silu_fwd', pre.data_ptr(), act.data_ptr(), pre.numel(), _st())
    return act


def silu_bwd(dact, pre):
    _chk(dact, F32), _chk(pre, F32)
    dpre = _new(pre.shape, dtype=BF16, device=pre.device)
    _lib.call('alm_silu_bwd', dact.data_ptr(), pre.data_ptr(), dpre.data_ptr(), pre.numel(), _st())
    return dpre


def posmlp_out_fwd(act, W, b, special, inv_scale):
    """act bf16 [L, C]; W fp32 [H, C]; b [H]; special fp32 [H] | None -> tbl fp32 [H, L + 1] (x inv_scale, slot 0 = special)."""
    _chk(act, BF16), _chk(W, F32), _chk(b, F32)
    L, C = act.shape
    H = W.shape[0]
    tbl = _new((H, L + 1), dtype=F32, device=act.device)
    _lib.call('alm_posmlp_out_fwd', act.data_ptr(), W.data_ptr(), b.data_ptr(), _p(special), tbl.data_ptr(), L, C, H, float(inv_scale), _st())
    return tbl


def posmlp_out_bwd(dtbl, W, pre, inv_scale):
    """-> (g bf16 [L, Hp] = d(loss)/d(last-layer output), dpre bf16 [L, C] for the layer below, dspecial fp32 [H])."""
    _chk(dtbl, F32), _chk(W, F32), _chk(pre, F32)
    H, LT = dtbl.shape
    L, C = pre.shape
    assert LT == L + 1 and dtbl.is_contiguous()
    Hp = (H + 7) // 8 * 8
    g = _new((L, Hp), dtype=BF16, device=pre.device)
    dpre = _new((L, C), dtype=BF16, device=pre.device)
    dsp = _new(H, dtype=F32, device=pre.device)
    _lib.call('alm_posmlp_out_bwd', dtbl.data_ptr(), W.data_ptr(), pre.data_ptr(), g.data_ptr(), dpre.data_ptr(), dsp.data_ptr(), L, C, H, Hp,
              float(inv_scale), _st())
    return g, dpre, dsp


# ------------------------------------------------------------------------------------------------ hyper-connections

_HC_KEYS7 = ('gamma', 'Wa', 'sa', 'Aa', 'wb', 'sb', 'Bb')


def hc_fwd(R_in, B, S, N, D, *, y_prev=None, coef_prev=None, hc=None, ln_gamma=None, final=False, want_x=True, rin_bcast=False, r_dtype=F32,
           final_f32=False, x_out=None, xn_out=None):
    """Hyper-connection forward pass over the residual streams R_in [B, S, N, D] (C ABI: alm_hc_fwd).
      y_prev / coef_prev given : depth connection of the previous branch (R = mix(R_in) + beta * y_prev)
      hc given                 : width connection + pre-LayerNorm of the next branch on that R
      final                    : depth connection + stream sum + final LayerNorm (ln_gamma); final_f32: its output in fp32 (logit heads)
      rin_bcast                : R_in is ONE fp32 [B*N, D] tensor every stream equals (right after the stream expansion)
      r_dtype                  : storage type of the stream tensors (fp32 | bf16; arithmetic is fp32 either way)
    -> dict(R=..., x, xn, mean, rstd, coef | xs, hn, mean, rstd)."""
    _chk(R_in, F32 if rin_bcast else r_dtype)
    dev, M = R_in.device, B * N
    depth, width = y_prev is not None, hc is not None
    mode = (1 if depth else 0) | (2 if width else 0) | (4 if final else 0)
    out = {}
    R_out = _new((B, S, N, D), dtype=r_dtype, device=dev) if (depth and not final) else None
    x = xn = xn32 = mean = rstd = coef = xs = None
    if width or final:
        if final and final_f32:
            xn32 = _new((M, D), dtype=F32, device=dev)
        else:
            xn = _new((M, D), dtype=BF16, device=dev) if xn_out is None else xn_out
            assert xn.shape == (M, D) and xn.dtype == BF16 and xn.is_contiguous()
        mean = _new(M, dtype=F32, device=dev)
        rstd = _new(M, dtype=F32, device=dev)
    if width:
        x = (_new((M, D), dtype=BF16, device=dev) if x_out is None else x_out) if want_x else None
        assert x is None or (x.shape == (M, D) and x.dtype == BF16 and x.is_contiguous())
        coef = _new((M, _lib.query('alm_hc_coef_width', S)), dtype=F32, device=dev)
    if final:
        xs = _new((M, D), dtype=F32, device=dev)
    hp = [hc[k].data_ptr() for k in _HC_KEYS7] if width else [None] * 7
    _lib.call('alm_hc_fwd', R_in.data_ptr(), int(rin_bcast), int(r_dtype == BF16), _p(y_prev), y_prev.stride(0) if depth else 0, _p(coef_prev), _p(R_out),
              *hp, _p(ln_gamma), _p(x), D, _p(xn), D, _p(xn32), _p(mean), _p(rstd), _p(coef), _p(xs), mode, B, S, N, D, _st())
    out.update(R=R_out if depth else R_in, x=x, xn=xn if xn32 is None else xn32, mean=mean, rstd=rstd, coef=coef, xs=xs)
    return out


def hc_bwd(dRn, B, S, N, D, *, bcast=False, dx=None, dxn=None, extra=None, mean=None, rstd=None, ln_gamma=None, R=None, coef=None, dbeta=None,
           hc=None, y_prev=None, coef_prev=None, r_bcast=False, sum_only=False, r_dtype=F32, dsum_scale=1.0, dy_out=None, defer_grads=False):
    """Hyper-connection backward (C ABI: alm_hc_bwd).  dRn: gradient wrt the residual output of a width connection, [B, S, N, D] (r_dtype), or with
    bcast fp32 [B*N, D] shared by all streams.  hc / R / coef / dbeta given: width-connection backward -> dR + the 7 parameter gradients; the
    gradient wrt the branch input is either `dx` (fp32, LayerNorm backward already applied) or `dxn` (bf16, wrt the LayerNorm output) +
    optional `extra` (bf16, added to dx) + mean / rstd / ln_gamma: then the LayerNorm backward is fused (grads['ln'] = its weight gradient).
    y_prev / coef_prev given: depth-connection backward of the previous branch on that dR (or on dRn) -> dy (bf16), dbeta_prev.
    r_bcast: R is one fp32 [B*N, D] tensor for all streams; sum_only: return dsum fp32 [B*N, D] = sum over streams of dR instead of dR.
    r_dtype: storage type of dRn / R / dR (the non-bcast forms).  dsum_scale: factor on dsum (grad_shrink's alpha rides here).
    defer_grads: the parameter gradients are NOT finished here -- `grads` is None and `part` = (partial rows, row count, hc) goes to
    hc_param_grads_batched together with the other branches' (two launches for all of them instead of two per branch).
    -> dict(dR, dsum, grads, dy, dbeta, part)."""
    width, depth = hc is not None, y_prev is not None
    mode = (2 if width else 0) | (1 if depth else 0)
    dev, M = dRn.device, B * N
    _chk(dRn, F32 if bcast else r_dtype)
    if R is not None:
        _chk(R, F32 if r_bcast else r_dtype)
    rbf = int(r_dtype == BF16)
    dR = dsum = part = dy = dbo = None
    if width:
        assert (dx is None) != (dxn is None)
        if sum_only:
            dsum = _new((M, D), dtype=F32, device=dev)
        else:
            dR = _new((B, S, N, D), dtype=r_dtype, device=dev)
        rows = _lib.query('alm_hc_partial_rows', mode, int(dxn is not None), rbf, S, M, D)
        P = _lib.query('alm_hc_partial_width', S, D)
        part = _new((rows, P), dtype=F32, device=dev)
    if depth:
        dy = _new((M, D), dtype=BF16, device=dev) if dy_out is None else dy_out
        assert dy.shape == (M, D) and dy.dtype == BF16 and dy.is_contiguous()
        dbo = _new((M, S), dtype=F32, device=dev)
    hp = [hc[k].data_ptr() for k in ('gamma', 'Wa', 'sa', 'wb', 'sb')] if width else [None] * 5
    _lib.call('alm_hc_bwd', dRn.data_ptr(), int(bcast), rbf, _p(dx), dx.stride(0) if dx is not None else 0, _p(dxn),
              dxn.stride(0) if dxn is not None else 0, _p(extra), extra.stride(0) if extra is not None else 0, _p(mean), _p(rstd), _p(ln_gamma),
              _p(R), int(r_bcast), _p(coef), _p(dbeta), *hp, _p(dR), _p(dsum), float(dsum_scale), _p(part), _p(y_prev), y_prev.stride(0) if depth else 0, _p(coef_prev), _p(dy), D, _p(dbo),
              mode, B, S, N, D, _st())
    grads = None
    if width and defer_grads:
        return dict(dR=dR, dsum=dsum, grads=None, dy=dy, dbeta=dbo, part=(part, rows, hc))
    if width:
        chunks = _lib.query('alm_colsum_chunks', rows)
        if chunks > 1:                                       # stage 1 of the column sums; the chunk rows are summed inside alm_hc_param_grads
            ws = _new((chunks, P), dtype=F32, device=dev)
            _lib.call('alm_colsum_partial', part.data_ptr(), P, rows, P, ws.data_ptr(), _st())
        else:
            ws = colsum(part)
        g = _new(_lib.query('alm_hc_grads_width', S, D), dtype=F32, device=dev)
        _lib.call('alm_hc_param_grads', ws.data_ptr(), chunks, hc['gamma'].data_ptr(), hc['Wa'].data_ptr(), hc['wb'].data_ptr(), g.data_ptr(), S, D, _st())
        o = 0
        grads = {}
        for name, shape in (('Wa', (D, S + 1)), ('wb', (D,)), ('gamma', (D,)), ('Aa', (S, S + 1)), ('Bb', (S,)), ('sa', ()), ('sb', ()), ('ln', (D,))):
            n = 1
            for d in shape:
                n *= d
            grads[name] = g[o:o + n].view(shape)
            o += n
    return dict(dR=dR, dsum=dsum, grads=grads, dy=dy, dbeta=dbo, part=None)


def _hc_grad_views(g, S, D):
    o, grads = 0, {}
    for name, shape in (('Wa', (D, S + 1)), ('wb', (D,)), ('gamma', (D,)), ('Aa', (S, S + 1)), ('Bb', (S,)), ('sa', ()), ('sb', ()), ('ln', (D,))):
        n = 1
        for d in shape:
            n *= d
        grads[name] = g[o:o + n].view(shape)
        o += n
    return grads


def hc_param_grads_batched(parts, S, D):
    """parts: list of hc_bwd(..., defer_grads=True)['part'] = (partial rows, row count, hc) -> list of gradient dicts (keys as hc_bwd's `grads`), finished
    in two launches per 16 branches (C ABI: alm_hc_param_grads_batched)."""
    out = []
    for i0 in range(0, len(parts), 16):
        grp = parts[i0:i0 + 16]
        nb, dev = len(grp), grp[0][0].device
        P, GW = _lib.query('alm_hc_partial_width', S, D), _lib.query('alm_hc_grads_width', S, D)
        g = _new((nb, GW), dtype=F32, device=dev)
        ws = _new((nb, 16, P), dtype=F32, device=dev)
        arr_p = (ctypes.c_void_p * nb)(*[t[0].data_ptr() for t in grp])
        arr_r = (ctypes.c_int * nb)(*[int(t[1]) for t in grp])
        arr_g = (ctypes.c_void_p * nb)(*[t[2]['gamma'].data_ptr() for t in grp])
        arr_wa = (ctypes.c_void_p * nb)(*[t[2]['Wa'].data_ptr() for t in grp])
        arr_wb = (ctypes.c_void_p * nb)(*[t[2]['wb'].data_ptr() for t in grp])
        arr_o = (ctypes.c_void_p * nb)(*[g[z].data_ptr() for z in range(nb)])
        _lib.call('alm_hc_param_grads_batched', arr_p, arr_r, nb, arr_g, arr_wa, arr_wb, ws.data_ptr(), arr_o, S, D, _st())
        out += [_hc_grad_views(g[z], S, D) for z in range(nb)]
    return out


# un-fused single-connection forms (kernel tests; the product path uses the fused modes)
def hc_width_fwd(R, hc, ln_gamma, B, S, N, D, want_x=True):
    r = hc_fwd(R, B, S, N, D, hc=hc, ln_gamma=ln_gamma, want_x=want_x, r_dtype=R.dtype)
    return r['x'], r['xn'], r['mean'], r['rstd'], r['coef']


def hc_depth_fwd(R, y, coef, B, S, N, D):
    return hc_fwd(R, B, S, N, D, y_prev=y, coef_prev=coef, r_dtype=R.dtype)['R']


def hc_depth_bwd(dRn, y, coef, B, S, N, D, bcast=False):
    r = hc_bwd(dRn, B, S, N, D, bcast=bcast, y_prev=y, coef_prev=coef, r_dtype=F32 if bcast else dRn.dtype)
    return r['dy'], r['dbeta']


def hc_width_bwd(dRn, dx, R, coef, dbeta, hc, B, S, N, D):
    r = hc_bwd(dRn, B, S, N, D, dx=dx, R=R, coef=coef, dbeta=dbeta, hc=hc, r_dtype=R.dtype)
    return r['dR'], r['grads']


def streams_expand(x, B, S):
    _chk(x, F32)
    nd = x.numel() // B
    R = _new((B, S) + tuple(x.shape[1:]), dtype=F32, device=x.device)
    _lib.call('alm_streams_expand', x.data_ptr(), R.data_ptr(), B, S, nd, _st())
    return R


def streams_reduce(R, B, S):
    nd = R.numel() // (B * S)
    x = _new((B,) + tuple(R.shape[2:]), dtype=F32, device=R.device)
    _lib.call('alm_streams_reduce', R.data_ptr(), x.data_ptr(), B, S, nd, _st())
    return x


def residual_add(x, y):
    """fp32 x [rows, D] + bf16 y [rows, D] -> new fp32."""
    rows, D = y.shape
    out = _new((rows, D), dtype=F32, device=y.device)
    _lib.call('alm_residual_add', x.data_ptr(), y.data_ptr(), y.stride(0), out.data_ptr(), rows, D, _st())
    return out


def f32_to_bf16(a, b=None, out=None):
    rows, D = a.shape
    if out is None:
        out = _new((rows, D), dtype=BF16, device=a.device)
    assert out.shape == (rows, D) and out.dtype == BF16 and out.is_contiguous()
    _lib.call('alm_f32_to_bf16', a.data_ptr(), _p(b), out.data_ptr(), D, rows, D, _st())
    return out


def add_f32(a, b, scale=1.0):
    """(a + b) * scale, fp32"""
    out = _new_like(a)
    _lib.call('alm_add_f32', a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), float(scale), _st())
    return out


# ------------------------------------------------------------------------------------------------ token-id side

def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


_err_flags = {}


def device_error_flag(device):
    """int32 device word the id-consuming kernels raise when they meet an id outside its table (the reference's nn.Embedding would raise
    IndexError; the kernels never dereference such an id).  Poll it with check_device_errors() at a point where a sync is acceptable."""
    key = (device.type, device.index)
    if key not in _err_flags:
        _err_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _err_flags[key]


def check_device_errors(device=None):
    """Raises IndexError if any embedding lookup since the last check saw an out-of-range token id (synchronises)."""
    for key, flag in list(_err_flags.items()):
        if device is not None and (device.type, device.index) != key:
            continue
        if int(flag.item()) != 0:
            flag.zero_()
            _err_poll.pop(key, None)
            raise IndexError('token id out of range of its embedding table (audiolm_pytorch_amd embed_assemble): the offending positions were '
                             'embedded as zero vectors')


_err_poll = {}
CHECK_IDS = __import__('os').environ.get('ALM_CHECK_IDS', '1') != '0'


def poll_device_errors(device):
    """Non-blocking form of check_device_errors(), called by every embedding lookup (ALM_CHECK_IDS=0 turns it off): the error word is copied to
    pinned host memory asynchronously and the copy issued by an EARLIER call is examined once it has landed -- an out-of-range token id therefore
    raises IndexError (what the reference's nn.Embedding does at once) one or two lookups late, without a host-device synchronisation anywhere.
    Skipped while a hipGraph is being captured (events cannot be queried there)."""
    if not CHECK_IDS or device.type != 'cuda' or torch.cuda.is_current_stream_capturing():
        return
    key = (device.type, device.index)
    flag = _err_flags.get(key)
    if flag is None:
        return
    st = _err_poll.get(key)
    if st is None:
        st = _err_poll[key] = [torch.zeros(1, dtype=torch.int32).pin_memory(), None]
    host, ev = st
    if ev is not None:
        if not ev.query():
            return                                            # the previous copy is still in flight: look again at the next lookup
        st[1] = None
        if int(host[0]) != 0:
            flag.zero_()
            host.zero_()
            raise IndexError('token id out of range of its embedding table (audiolm_pytorch_amd embed_assemble, an earlier step): the offending '
                             'positions were embedded as zero vectors')
    host.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st[1] = ev


def _table_rows(tables):
    return (ctypes.c_int * len(tables))(*[t.shape[0] for t in tables])


def embed_assemble(tables, src_a, src_b, rows, D):
    """tables: fp32 [rows_t, D] each.  Ids outside a table never touch memory: zero vector + device error flag (see device_error_flag)."""
    out = _new((rows, D), dtype=F32, device=src_a.device)
    arr = _ptr_array(tables)
    poll_device_errors(src_a.device)
    _lib.call('alm_embed_assemble', ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(_table_rows(tables), ctypes.c_void_p), len(tables), src_a.data_ptr(),
              src_b.data_ptr(), out.data_ptr(), rows, D, device_error_flag(src_a.device).data_ptr(), _st())
    return out


def embed_scatter_add(grad_tables, src_a, src_b, dout, alpha, rows, D):
    arr = _ptr_array(grad_tables)
    _lib.call('alm_embed_scatter_add', ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(_table_rows(grad_tables), ctypes.c_void_p), len(grad_tables),
              src_a.data_ptr(), src_b.data_ptr(), dout.data_ptr(), float(alpha), rows, D, _st())


# fp32 elements of chunk partials the owned scatter may allocate per backward (default 2^28 = 1 GiB); above it the atomic kernel runs (ALM_EMBED_SCATTER_WS_CAP)
EMBED_SCATTER_WS_CAP = int(os.environ.get('ALM_EMBED_SCATTER_WS_CAP', str(1 << 28)))


def embed_scatter_owned(grad_tables, src_a, src_b, dout, alpha, rows, D):
    """deterministic form of embed_scatter_add (one owner per destination row, fixed summation order, no atomics): WRITES every row of every table --
    `grad_tables` may be uninitialised memory (alm_embed_scatter_owned).  -> True; False when the size is outside its range and the atomic kernel ran instead"""
    arr, tr = _ptr_array(grad_tables), _table_rows(grad_tables)
    nws = _lib.query('alm_embed_scatter_ws_floats', ctypes.cast(tr, ctypes.c_void_p), len(grad_tables), rows, D)
    if nws < 0 or nws > EMBED_SCATTER_WS_CAP:
        # outside the owned kernel's range (rows >= 2^28, or a chunk-partial workspace that grows with B * N beyond the cap: ~0.5 GB at 128 k tokens):
        # the atomic kernel takes over on zeroed tables -- same sums, arrival-order rounding (the caller is told through the return value)
        for g in grad_tables:
            g.zero_()
        embed_scatter_add(grad_tables, src_a, src_b, dout, alpha, rows, D)
        return False
    ws = _new(max(nws, 1), dtype=F32, device=dout.device)
    _lib.call('alm_embed_scatter_owned', ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(tr, ctypes.c_void_p), len(grad_tables),
              src_a.data_ptr(), src_b.data_ptr(), dout.data_ptr(), float(alpha), rows, D, ws.data_ptr(), _st())
    return True


def gather_split(inp, idx=None, rows_out=None, out=None):
    """fp32 inp [rows_in, D] (row stride arbitrary) -> (hi, lo) bf16 [rows_out, D] with hi + lo ~= inp[idx] to ~16 mantissa bits: the split-bf16
    operands of the logit heads.  idx int32 [rows_out] (-1 = zero row) or None (identity; rows past rows_in are zero = row padding).
    out = (hi, lo): bf16 [rows_out, D] views of ONE row stride (column blocks of a wider buffer: the K-concatenated head operands)."""
    _chk(inp, F32)
    rows_in, D, ld = _rows_ld(inp)
    if idx is not None:
        rows_out = idx.numel()
    elif rows_out is None:
        rows_out = rows_in
    if out is None:
        hi = _new((rows_out, D), dtype=BF16, device=inp.device)
        lo = _new((rows_out, D), dtype=BF16, device=inp.device)
    else:
        hi, lo = out
        assert hi.shape == lo.shape == (rows_out, D) and hi.dtype == lo.dtype == BF16 and hi.stride(1) == lo.stride(1) == 1 and hi.stride(0) == lo.stride(0)
    _lib.call('alm_gather_split_bf16', inp.data_ptr(), ld, rows_in, _p(idx), hi.data_ptr(), lo.data_ptr(), hi.stride(0), rows_out, D, _st())
    return hi, lo


def gather_rows(inp, idx, out=None):
    rows = idx.numel()
    D = inp.shape[1]
    if out is None:
        out = _new((rows, D), dtype=BF16, device=inp.device)
    _lib.call('alm_gather_rows_bf16', inp.data_ptr(), inp.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0), rows, D, _st())
    return out


def scatter_rows(inp, idx, out):
    rows = idx.numel()
    _lib.call('alm_scatter_rows_bf16', inp.data_ptr(), inp.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0), rows, inp.shape[1], _st())
    return out


def cross_entropy_fwd(logits, labels, C, ignore_index=-1):
    """logits fp32 [rows, >= C]; labels int64 [rows] -> (loss_rows fp32, lse fp32)."""
    _chk(logits, F32), _chk(labels, torch.int64)
    rows = labels.numel()
    loss = _new(rows, dtype=F32, device=logits.device)
    lse = _new(rows, dtype=F32, device=logits.device)
    _lib.call('alm_cross_entropy_fwd', logits.data_ptr(), logits.stride(0), labels.data_ptr(), loss.data_ptr(), lse.data_ptr(), rows, C, ignore_index, _st())
    return loss, lse


def cross_entropy_bwd(logits, labels, lse, gscale, C, Cpad, ignore_index=-1):
    rows = labels.numel()
    d = _new((rows, Cpad), dtype=BF16, device=logits.device)
    _lib.call('alm_cross_entropy_bwd', logits.data_ptr(), logits.stride(0), labels.data_ptr(), lse.data_ptr(), gscale.data_ptr(), d.data_ptr(), Cpad,
              rows, C, Cpad, ignore_index, _st())
    return d


def reduce_sum(x, scale=1.0):
    out = _new((), dtype=F32, device=x.device)
    _lib.call('alm_reduce_sum', x.data_ptr(), x.numel(), out.data_ptr(), float(scale), _st())
    return out


# ------------------------------------------------------------------------------------------------ SoundStream tokenize path

def conv1d_pack(w):
    """nn.Conv1d weight fp32 [Cout, Cin, k] -> packed [k][Cin_pad][Cout_pad] fp32 (once per weight update)."""
    _chk(w, F32)
    Cout, Cin, ks = w.shape
    wp = _new(_lib.query('alm_conv1d_packed_floats', Cout, Cin, ks), dtype=F32, device=w.device)
    _lib.call('alm_conv1d_pack', w.contiguous().data_ptr(), wp.data_ptr(), Cout, Cin, ks, _st())
    return wp


def conv1d_causal(x, wp, bias, Cout, ksize, *, stride=1, dilation=1, elu=False, residual=None, zero_pad=False):
    """x fp32 [B, Cin, T] -> fp32 [B, Cout, T // stride]: CausalConv1d (reflect left pad; zero_pad: zeros) + bias (+ ELU) (+ residual)."""
    _chk(x, F32)
    B, Cin, T = x.shape
    assert x.is_contiguous()
    Tout = (T - stride) // stride + 1
    out = _new((B, Cout, Tout), dtype=F32, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    _lib.call('alm_conv1d_causal', x.data_ptr(), wp.data_ptr(), bias.data_ptr(), _p(residual), out.data_ptr(), B, Cin, Cout, T, ksize, stride,
              dilation, int(elu), int(zero_pad), _st())
    return out


def resunit_supported(C):
    return C % 32 == 0 and C <= 256


def resunit_causal(x, w7p, b7, w1p, b1, ksize, dilation):
    """x fp32 [B, C, T] -> x + ELU(conv_k1(ELU(conv_k,dilation(x)))) in one launch (alm_resunit_causal; C % 32 == 0, C <= 256)"""
    _chk(x, F32)
    B, C, T = x.shape
    assert x.is_contiguous()
    out = _new_like(x)
    _lib.call('alm_resunit_causal', x.data_ptr(), w7p.data_ptr(), b7.data_ptr(), w1p.data_ptr(), b1.data_ptr(), out.data_ptr(), B, C, T, ksize, dilation, _st())
    return out


def phase_interleave(y, Cout, s):
    """y fp32 [B, s * Cout, n] (phase-major channels) -> [B, Cout, n * s]: out[b, co, q * s + r] = y[b, r * Cout + co, q]."""
    _chk(y, F32)
    B, SC, n = y.shape
    assert SC == s * Cout and y.is_contiguous()
    out = _new((B, Cout, n * s), dtype=F32, device=y.device)
    _lib.call('alm_phase_interleave', y.data_ptr(), out.data_ptr(), B, Cout, s, n, _st())
    return out


def rvq_decode(idx, E, out):
    """idx int64 [T, Q] (row stride arbitrary, -1 = no code), E fp32 [Q, C, d] -> out fp32 [T, d] view (row stride arbitrary): summed code vectors."""
    _chk(E, F32), _chk(out, F32)
    assert idx.dtype == torch.int64 and idx.stride(1) == 1 and out.stride(1) == 1 and E.is_contiguous()
    T, Q = idx.shape
    _lib.call('alm_rvq_decode', idx.data_ptr(), idx.stride(0), E.data_ptr(), out.data_ptr(), out.stride(0), T, E.shape[2], E.shape[1], Q, _st())
    return out


def rvq_pack(E):
    """codebooks fp32 [Q, C, d] -> (Et [Q, dpad, Cpad] MFMA-ordered image, e2 [Q, Cpad])."""
    _chk(E, F32)
    Q, C, d = E.shape
    CP = _lib.query('alm_rvq_padded_codes', C)
    Et = _new((Q, _lib.query('alm_rvq_padded_dim', d), CP), dtype=F32, device=E.device)
    e2 = _new((Q, CP), dtype=F32, device=E.device)
    _lib.call('alm_rvq_pack', E.contiguous().data_ptr(), Et.data_ptr(), e2.data_ptr(), Q, C, d, _st())
    return Et, e2


def rvq_encode(x, E, Et, e2, idx_out=None, quant_out=None):
    """x fp32 [T, d] (row stride arbitrary) against codebooks E [Q, C, d] -> int64 indices [T, Q] (written into idx_out view if given)."""
    _chk(x, F32)
    T, d = x.shape
    Q, C, _ = E.shape
    assert x.stride(1) == 1
    if idx_out is None:
        idx_out = _new((T, Q), dtype=torch.int64, device=x.device)
    assert idx_out.shape == (T, Q) and idx_out.stride(1) == 1
    _lib.call('alm_rvq_encode', x.data_ptr(), x.stride(0), E.data_ptr(), Et.data_ptr(), e2.data_ptr(), idx_out.data_ptr(), idx_out.stride(0),
              _p(quant_out), quant_out.stride(0) if quant_out is not None else 0, T, d, C, Q, _st())
    return idx_out


def bct_to_btc(x):
    B, C, T = x.shape
    out = _new((B, T, C), dtype=F32, device=x.device)
    _lib.call('alm_bct_to_btc', x.data_ptr(), out.data_ptr(), B, C, T, _st())
    return out


# ---- SoundStream LocalTransformer (csrc/local_attn.hip), fp32, codec layout [B, C, T]

def layernorm_bct(x, gamma, beta, eps=1e-5):
    _chk(x, F32)
    B, C, T = x.shape
    assert x.is_contiguous()
    out = _new_like(x)
    _lib.call('alm_layernorm_bct', x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), B, C, T, float(eps), _st())
    return out


def geglu_bct(x):
    """x fp32 [B, 2 I, T] -> [B, I, T]: x[:, :I] * gelu(x[:, I:])"""
    _chk(x, F32)
    B, C2, T = x.shape
    assert x.is_contiguous() and C2 % 2 == 0
    out = _new((B, C2 // 2, T), dtype=F32, device=x.device)
    _lib.call('alm_geglu_bct', x.data_ptr(), out.data_ptr(), B, C2 // 2, T, _st())
    return out


def local_attn(qkv, q_scale, k_scale, cos_t, sin_t, xpos_t, gates, heads, dim_head, window, scale):
    """qkv fp32 [B, 3 H dh, T], gates fp32 [B, H, T] | None, slot tables [2 window, dh] -> [B, H dh, T]"""
    _chk(qkv, F32)
    B, C3, T = qkv.shape
    assert qkv.is_contiguous() and C3 == 3 * heads * dim_head
    for t in (cos_t, sin_t, xpos_t):
        assert t.dtype == F32 and t.is_contiguous() and tuple(t.shape) == (2 * window, dim_head) and t.device == qkv.device
    out = _new((B, heads * dim_head, T), dtype=F32, device=qkv.device)
    _lib.call('alm_local_attn', qkv.data_ptr(), q_scale.data_ptr(), k_scale.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), xpos_t.data_ptr(), _p(gates),
              out.data_ptr(), B, heads, dim_head, T, window, float(scale), _st())
    return out
