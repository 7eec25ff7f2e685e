"""Launch lists: the transformer stack's forward / backward launch sequences recorded once per shape and re-issued from ONE C call (C ABI: alm_list_run,
csrc/launchlist.hip; the depth loop of reference audiolm_pytorch.py:528-547 and its autograd).

The Python path (core.stack_forward / core.stack_backward) stays the single description of the launch sequence.  For an eligible call (plain training /
evaluation forward: no conditioning context, no dropout, no sampling cache, no data-parallel gradient hook, not inside a hipGraph capture) the first
calls of a (configuration, batch shape) key run that path THREE ways:

  1. SIZE    -- an ordinary step whose device allocations (ops._new: every temporary of the pass) are tallied,
  2. RECORD  -- an ordinary step whose temporaries are bump-allocated from ONE arena of that size and whose `alm_*` launches (`_lib.call`) are written down:
                every pointer argument is classified as (base, byte offset) against the table of bases of the pass -- [arena, (forward arena), input, key
                mask, structured-bias tensors, every parameter, every packed weight image] -- or refused (a pointer into no known buffer turns the list off
                for this key; the step itself is unaffected, it IS the eager step),
  3. REPLAY  -- one fresh arena per pass, the table of bases refilled from the live tensors, one alm_list_run call.  Same entry points, same order, same
                scalars: the outputs and every gradient are bit-identical to the eager step (tests/test_gpu_launchlist.py).

Host time of the stack (headline shape): ~3.4 ms of Python per step -> two C calls.  ALM_LAUNCH_LIST=0 turns it off (A/B)."""
from __future__ import annotations

import bisect
import ctypes
import dataclasses
import math
import os
import struct
import sys

import torch

from . import _lib, ops

ENABLED = os.environ.get('ALM_LAUNCH_LIST', '1') != '0'
MAX_ARENA_BYTES = int(float(os.environ.get('ALM_LAUNCH_LIST_MAX_GB', '24')) * (1 << 30))    # per pass; larger steps keep the eager path (frees as it goes)
MAX_PLANS = max(1, int(os.environ.get('ALM_LAUNCH_LIST_MAX_PLANS', '64')))                   # recorded (configuration, shape) keys kept; least recently used dropped
ALIGN = 256
LITERAL, STREAM, HOST_PTRS, HOST_INTS = 0, 0xFFFF, 0xFFFE, 0xFFFD
_P, _I, _L, _F, _U = _lib._P, _lib._I, _lib._L, _lib._F, _lib._U
_M64 = (1 << 64) - 1

STATS = dict(sized=0, recorded=0, replayed=0, refused=0)         # counters (tests / bench report them)


_ITEM = {torch.float32: 4, torch.bfloat16: 2, torch.uint8: 1, torch.int32: 4, torch.int64: 8, torch.float16: 2, torch.float64: 8, torch.bool: 1}


def _nbytes(shape, dtype):
    n = shape if isinstance(shape, int) else math.prod(shape)
    return n * (_ITEM.get(dtype) or torch.empty((), dtype=dtype).element_size())


def _round(n):
    return -(-n // ALIGN) * ALIGN


class _Tally:
    """SIZE pass: torch's allocator, bytes counted the way the arena will lay them out"""

    def __init__(self):
        self.bytes = 0

    def empty(self, shape, dtype, device):
        self.bytes += _round(_nbytes(shape, dtype))
        return torch.empty(shape, dtype=dtype, device=device)

    def zeros(self, shape, dtype, device):
        self.bytes += _round(_nbytes(shape, dtype))
        return torch.zeros(shape, dtype=dtype, device=device)


class _Arena:
    """RECORD pass: bump allocator over one uint8 buffer (every allocation ALIGN-aligned); an allocation that does not fit falls back to torch and
    marks the recording failed (the SIZE and RECORD passes then disagreed about the allocation sequence)"""

    def __init__(self, nbytes, device, recorder=None):
        self.cap = _round(max(int(nbytes), ALIGN))
        self.buf = torch.empty(self.cap, dtype=torch.uint8, device=device)
        self.base = self.buf.data_ptr()
        self.off = 0
        self.recorder = recorder

    def empty(self, shape, dtype, device):
        n = _nbytes(shape, dtype)
        off, end = self.off, self.off + n
        if end > self.cap:
            if self.recorder is not None:
                self.recorder.fail(f'arena overflow ({end} > {self.cap} bytes)')
            return torch.empty(shape, dtype=dtype, device=device)
        self.off = _round(end)
        return self.buf[off:end].view(dtype).view(shape)

    def zeros(self, shape, dtype, device):
        t = self.empty(shape, dtype, device)
        if t.numel():
            ops.memset_zero(t)                           # (recorded like any other launch)
        return t


def _extent(t):
    """bytes from t.data_ptr() to the end of the last element a view reaches"""
    if t.numel() == 0:
        return 0
    return (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size()


class Recorded:
    """one pass's launch list in the C layout + how to rebuild its outputs from an arena"""

    def __init__(self, entries, slots, reloc, nbases, arena_bytes):
        n, ns = len(entries), len(slots)
        self.n, self.nslots, self.nbases, self.arena_bytes = n, ns, nbases, arena_bytes
        self.entries = (_lib.AlmListEntry * max(n, 1))(*[_lib.AlmListEntry(op, na, first, 0) for op, na, first in entries])
        self.slots = (ctypes.c_ulonglong * max(ns, 1))(*slots)
        self.reloc = (ctypes.c_ushort * max(ns, 1))(*reloc)
        self.bases = (ctypes.c_ulonglong * max(nbases, 1))()
        self.failed_at = ctypes.c_int(-1)
        self.names = None
        self.outputs = None

    def run(self, stream):
        fn = _lib._BOUND.get('alm_list_run')
        if fn is None:
            fn = _lib._BOUND['alm_list_run'] = getattr(_lib.load(), 'alm_list_run')
        rc = fn(ctypes.addressof(self.entries), self.n, ctypes.addressof(self.slots), ctypes.addressof(self.reloc), self.nslots, ctypes.addressof(self.bases),
                self.nbases, stream, ctypes.addressof(self.failed_at))
        if rc != 0:
            at = self.failed_at.value
            raise _lib.AlmError(f'alm_list_run: launch {at} ({self.names[at] if self.names and 0 <= at < len(self.names) else "?"}) failed with code {rc}')


class Recorder:
    """_lib.RECORDER of a RECORD pass.  bases: list of (tensor | (address, nbytes) | None) -- base id = index + 1."""

    _OPIDS = {}

    def __init__(self, bases, stream):
        self.stream = int(stream or 0)
        self.entries, self.slots, self.reloc, self.names = [], [], [], []
        self.failed = None
        rng = []
        for i, b in enumerate(bases):
            if b is None:
                continue
            start, n = (b.data_ptr(), _extent(b)) if isinstance(b, torch.Tensor) else b
            if n > 0:
                rng.append((start, start + n, i + 1))
        rng.sort()
        self.starts = [r[0] for r in rng]
        self.rng = rng
        self.nbases = len(bases)
        self.literal = [(t.data_ptr(), t.data_ptr() + _extent(t)) for t in ops.persistent_buffers()]

    def fail(self, why):
        if self.failed is None:
            self.failed = why

    def _classify(self, p, where):
        i = bisect.bisect_right(self.starts, p) - 1
        k = i
        while k >= 0 and k > i - 8:
            s, e, bid = self.rng[k]
            if s <= p < e:
                return bid, p - s
            k -= 1
        for attempt in range(2):
            for s, e in self.literal:
                if s <= p < e:
                    return LITERAL, p
            self.literal = [(t.data_ptr(), t.data_ptr() + _extent(t)) for t in ops.persistent_buffers()]      # (one may have been created inside this pass)
        self.fail(f'{where}: device pointer {p:#x} lies in no buffer of this pass')
        return LITERAL, p

    def _opid(self, name):
        op = Recorder._OPIDS.get(name)
        if op is None:
            op = Recorder._OPIDS[name] = _lib.query('alm_list_op_id', name.encode())
        return op

    def note(self, name, args):
        op = self._opid(name)
        sig = _lib.SIGNATURES[name]
        if op < 0 or len(args) != len(sig):
            self.fail(f'{name}: not a launch-list entry point')
            return
        first, last = len(self.slots), len(sig) - 1
        arrays = []
        for i, (a, t) in enumerate(zip(args, sig)):
            if t is _P:
                if i == last:
                    if int(a or 0) != self.stream:
                        self.fail(f'{name}: launched on another stream')
                    self.slots.append(0), self.reloc.append(STREAM)
                elif a is None:
                    self.slots.append(0), self.reloc.append(LITERAL)
                elif isinstance(a, ctypes.Array):
                    arrays.append((len(self.slots), a))
                    self.slots.append(0), self.reloc.append(LITERAL)           # patched below
                elif isinstance(a, int):
                    bid, off = self._classify(a, f'{name} argument {i}')
                    self.slots.append(off), self.reloc.append(bid)
                else:
                    self.fail(f'{name} argument {i}: {type(a).__name__} is not a device pointer')
                    self.slots.append(0), self.reloc.append(LITERAL)
            elif t is _F:
                self.slots.append(struct.unpack('<I', struct.pack('<f', float(a)))[0]), self.reloc.append(LITERAL)
            else:
                self.slots.append(int(a) & _M64), self.reloc.append(LITERAL)
        for at, arr in arrays:                              # host arrays: their elements follow the entry's own slots
            here = len(self.slots)
            if arr._type_ is ctypes.c_void_p:
                for j, v in enumerate(arr):
                    if v is None:
                        self.slots.append(0), self.reloc.append(LITERAL)
                    else:
                        bid, off = self._classify(int(v), f'{name} host array element {j}')
                        self.slots.append(off), self.reloc.append(bid)
                self.slots[at], self.reloc[at] = here, HOST_PTRS
            elif arr._type_ is ctypes.c_int:
                vals = [int(v) & 0xFFFFFFFF for v in arr] + [0]
                for j in range(0, len(vals) - 1, 2):
                    self.slots.append(vals[j] | (vals[j + 1] << 32)), self.reloc.append(LITERAL)
                self.slots[at], self.reloc[at] = here, HOST_INTS
            else:
                self.fail(f'{name}: host array of {arr._type_.__name__}')
        self.entries.append((op, len(sig), first))
        self.names.append(name)

    def finish(self, arena_bytes):
        rec = Recorded(self.entries, self.slots, self.reloc, self.nbases, arena_bytes)
        rec.names = self.names
        return rec


def _spec(t, arena, rec):
    """how to rebuild output tensor `t` (a view into the pass's arena) from another arena: (element offset, dtype, shape, stride)"""
    if t is None:
        return None
    off = t.data_ptr() - arena.base
    if not (0 <= off and off + _extent(t) <= arena.cap) or off % t.element_size():
        rec.fail('an output of the pass does not live in its arena')
        return None
    return (off // t.element_size(), t.dtype, tuple(t.shape), tuple(t.stride()))


def _rebuild(spec, typed):
    if spec is None:
        return None
    off, dtype, shape, stride = spec
    return typed[dtype].as_strided(shape, stride, off)


def _typed(buf):
    return {torch.float32: buf.view(torch.float32), torch.bfloat16: buf.view(torch.bfloat16)}


# ------------------------------------------------------------------------------------------------ the stack's two passes

class Plan:
    def __init__(self, key):
        self.key = key
        self.state = 'size'                 # size -> record -> ready | off
        self.fwd_bytes = self.bwd_bytes = None
        self.fwd = self.bwd = None
        self.why = None
        self.dhn_dtype = None               # dtype of the gradient the backward list was recorded with (a scalar argument of its first launch depends on it)
        self.orphans = 0                    # forwards recorded since the last recorded backward (a caller that never runs backward must not keep recording)


class Replay:
    """what a replayed forward leaves for its backward (TransformerStackFn keeps it in ctx.saved)"""

    def __init__(self, plan, arena, xin):
        self.plan, self.arena, self.xin = plan, arena, xin          # (xin: the first branch's residual input IS the stack input -- the backward reads it)


PLANS = {}


def _switches(core):
    return (core.QKV_GROUP, core.QKV_GROUP_MAX_M, core.ASYNC_KV, core.DEFER_WGRAD, core.DEFER_GROUPS, core.DEFER_GROUP_SIZES, core.DEFER_MAX_BYTES, core.HC_BATCH_FINISH,
            core.PACK_ALL, core.ASYNC_WGRAD, core.SIDE_STREAMS, ops.NT_WS)


def plan_for(core, x, mask_u8, cfg, need, bias, nflat, defer, dx_scale):
    """the Plan of this call (created in state 'size' on first sight), or None when launch lists are off"""
    if not ENABLED:
        return None
    dev = x.device
    key = (dev.index, ops._st(), tuple(x.shape), need, mask_u8 is None, None if bias is None else int(bias.tbl.shape[1]), dataclasses.astuple(cfg), nflat,
           bool(defer), float(dx_scale), _switches(core))
    plan = PLANS.pop(key, None)
    if plan is None:
        plan = Plan(key)
        while len(PLANS) >= MAX_PLANS:                   # least recently used first (a dict keeps insertion order; a hit is re-inserted below): data-dependent
            PLANS.pop(next(iter(PLANS)))                 # sequence lengths (unique_consecutive, ragged batches) would otherwise grow the table without bound
    PLANS[key] = plan
    return None if plan.state == 'off' else plan


def _weight_images(core, cache, cfg):
    """every packed (W, W^T) image of the stack in a fixed order (they are re-created whenever a master weight changes: bases, not literals)"""
    out = []
    for l in range(cfg.depth):
        for kind in core.branch_kinds(cfg):
            for name in ('wq', 'wkv', 'wo', 'w1', 'w2'):
                hit = cache.store.get((l, kind, name))
                if hit is not None:
                    out += [hit[1][0], hit[1][1]]
    return out


def _bias_tensors(bias):
    return [] if bias is None else [bias.tbl, bias.qkey4, bias.kkey4, bias.qattr, bias.kattr]


def _turn_off(plan, why):
    plan.state, plan.why, plan.fwd, plan.bwd = 'off', why, None, None
    STATS['refused'] += 1
    print(f'[audiolm_pytorch_amd] launch list refused for a stack shape (the eager path stays in use): {why}', file=sys.stderr, flush=True)


def _fill(rec, first, tensors):
    b = rec.bases
    for i, t in enumerate(tensors, first):
        b[i] = 0 if t is None else t.data_ptr()


def forward(core, plan, xin, mask_u8, flat, cfg, cache, need, bias, defer):
    """stack_forward under the plan's current state -> (hn, saved)"""
    kw = dict(defer_wgrad=defer)
    if plan.state == 'ready' and (plan.bwd is not None or not need):
        core.pack_stack_weights(cache, flat, cfg)
        rec = plan.fwd
        arena = torch.empty(rec.arena_bytes, dtype=torch.uint8, device=xin.device)
        rec.bases[0] = arena.data_ptr()
        _fill(rec, 1, [xin, mask_u8] + _bias_tensors(bias) + list(flat) + _weight_images(core, cache, cfg))
        rec.run(ops._st())
        STATS['replayed'] += 1
        hn = _rebuild(rec.outputs, _typed(arena))
        return hn, (Replay(plan, arena, xin) if need else None)
    core.pack_stack_weights(cache, flat, cfg)            # (outside the tally / recording: the packed images are not temporaries of the pass)
    if plan.state == 'size':
        tally = _Tally()
        ops.ALLOC = tally
        try:
            hn, saved = core.stack_forward(xin, mask_u8, flat, cfg, cache, need, bias, **kw)
        finally:
            ops.ALLOC = None
        plan.fwd_bytes = tally.bytes
        STATS['sized'] += 1
        if tally.bytes > MAX_ARENA_BYTES:
            _turn_off(plan, f'forward arena of {tally.bytes / 2 ** 30:.1f} GB exceeds ALM_LAUNCH_LIST_MAX_GB')
        elif not need:
            plan.state = 'record'
        if saved is not None:
            saved['_ll'] = (plan, 'size', None, xin)
        return hn, saved
    # record (also: a forward that arrives while the backward list does not exist yet)
    plan.orphans += 1
    if plan.orphans > 8:
        _turn_off(plan, 'forwards with gradients enabled whose backward never ran')
        return core.stack_forward(xin, mask_u8, flat, cfg, cache, need, bias, **kw)
    bases = [None, xin, mask_u8] + _bias_tensors(bias) + list(flat) + _weight_images(core, cache, cfg)
    arena = _Arena(plan.fwd_bytes, xin.device)
    bases[0] = (arena.base, arena.cap)
    rec = Recorder(bases, ops._st())
    arena.recorder = rec
    ops.ALLOC, _lib.RECORDER = arena, rec
    try:
        hn, saved = core.stack_forward(xin, mask_u8, flat, cfg, cache, need, bias, **kw)
    finally:
        ops.ALLOC, _lib.RECORDER = None, None
    out = rec.finish(arena.cap)
    out.outputs = _spec(hn, arena, rec)
    if rec.failed:
        _turn_off(plan, 'forward: ' + rec.failed)
    else:
        plan.fwd = out
        STATS['recorded'] += 1
        if not need:
            plan.state = 'ready'
    if saved is not None:
        saved['_ll'] = (plan, 'record', arena, xin)
    return hn, saved


def backward(core, dhn, mask_u8, flat, cfg, cache, saved, bias, dx_scale):
    """stack_backward under the plan's current state -> (dx, grads, dtbl, dctx)"""
    if isinstance(saved, Replay):
        plan, rec, farena = saved.plan, saved.plan.bwd, saved.arena
        if dhn.dtype != plan.dhn_dtype:
            if plan.dhn_dtype != torch.float32:
                raise _lib.AlmError(f'the backward launch list of this stack was recorded with a {plan.dhn_dtype} output gradient and now receives {dhn.dtype} '
                                    '(ALM_LAUNCH_LIST=0 keeps the launch-by-launch path)')
            dhn = dhn.to(torch.float32)                   # (exact: the kernel widens bf16 gradients to fp32 itself)
        core.pack_stack_weights(cache, flat, cfg)
        arena = torch.empty(rec.arena_bytes, dtype=torch.uint8, device=dhn.device)
        rec.bases[0], rec.bases[1] = arena.data_ptr(), farena.data_ptr()
        _fill(rec, 2, [dhn, mask_u8, saved.xin] + _bias_tensors(bias) + list(flat) + _weight_images(core, cache, cfg))
        rec.run(ops._st())
        typed = _typed(arena)
        dx_s, grad_s, dtbl_s = rec.outputs
        return _rebuild(dx_s, typed), [_rebuild(s, typed) for s in grad_s], _rebuild(dtbl_s, typed), None
    plan, mode, farena, xin = saved['_ll']
    if plan.state == 'off' or mode == 'size':
        tally = _Tally() if plan.state == 'size' else None
        ops.ALLOC = tally
        try:
            res = core.stack_backward(dhn, mask_u8, flat, cfg, cache, saved, None, bias, dx_scale=dx_scale)
        finally:
            ops.ALLOC = None
        if tally is not None and plan.state == 'size':
            plan.bwd_bytes = tally.bytes
            if tally.bytes > MAX_ARENA_BYTES:
                _turn_off(plan, f'backward arena of {tally.bytes / 2 ** 30:.1f} GB exceeds ALM_LAUNCH_LIST_MAX_GB')
            else:
                plan.state = 'record'
        return res
    core.pack_stack_weights(cache, flat, cfg)
    bases = [None, (farena.base, farena.cap), dhn, mask_u8, xin] + _bias_tensors(bias) + list(flat) + _weight_images(core, cache, cfg)
    arena = _Arena(plan.bwd_bytes, dhn.device)
    bases[0] = (arena.base, arena.cap)
    rec = Recorder(bases, ops._st())
    arena.recorder = rec
    ops.ALLOC, _lib.RECORDER = arena, rec
    try:
        dx, grads, dtbl, dctx = core.stack_backward(dhn, mask_u8, flat, cfg, cache, saved, None, bias, dx_scale=dx_scale)
    finally:
        ops.ALLOC, _lib.RECORDER = None, None
    out = rec.finish(arena.cap)
    out.outputs = (_spec(dx, arena, rec), [_spec(g, arena, rec) for g in grads], _spec(dtbl, arena, rec))
    if dctx is not None:
        rec.fail('a context gradient')
    if rec.failed:
        if plan.state != 'off':
            _turn_off(plan, 'backward: ' + rec.failed)
    elif plan.state != 'off' and plan.fwd is not None:
        plan.bwd, plan.dhn_dtype, plan.orphans = out, dhn.dtype, 0
        plan.state = 'ready'
        STATS['recorded'] += 1
    return dx, grads, dtbl, dctx
