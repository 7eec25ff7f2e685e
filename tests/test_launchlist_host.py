"""CPU: the host side of the launch lists (audiolm-pytorch_amd/launchlist.py + alm_list_run's slot resolution) -- no kernel runs here.

The recorder must classify every pointer argument against the bases of a pass (nearest base below the address, inside its extent), refuse what it cannot
place, and lay host arrays out behind the entry; alm_list_run must resolve literals / base + offset / the stream / host arrays and stop at the first
failing entry.  alm_memset_zero with zero bytes is the one entry point that is safe to call without a GPU."""
import ctypes

import torch

import audiolm_pytorch_amd  # noqa: F401
from audiolm_pytorch_amd import _lib
from audiolm_pytorch_amd import launchlist as LL


def test_op_table_matches_the_binding_table():
    n = 0
    for name, sig in _lib.SIGNATURES.items():
        op = _lib.query('alm_list_op_id', name.encode())
        if op >= 0:
            n += 1
            assert _lib.query('alm_list_op_nargs', op) == len(sig), name
            assert sig[-1] is _lib._P, name                     # the stream is the last argument of every list entry point
    assert n >= 25
    assert _lib.query('alm_list_op_id', b'alm_pack_weights_multi') == -1          # host-struct arguments: not relocatable, not in the table
    assert _lib.query('alm_list_op_id', b'nonsense') == -1


def test_recorder_classifies_pointers_against_the_nearest_base():
    arena = torch.empty(4096, dtype=torch.uint8)
    x = torch.empty((4, 8), dtype=torch.float32)
    w = torch.empty((16, 8), dtype=torch.bfloat16)
    rec = LL.Recorder([(arena.data_ptr(), 4096), x, None, w[4:]], stream=0)
    rec.note('alm_memset_zero', (arena.data_ptr() + 256, 64, 0))
    rec.note('alm_memset_zero', (x.data_ptr() + 16, 8, None))
    rec.note('alm_memset_zero', (w.data_ptr() + 4 * 8 * 2 + 6, 2, 0))
    assert rec.failed is None
    assert rec.reloc == [1, 0, LL.STREAM, 2, 0, LL.STREAM, 4, 0, LL.STREAM]
    assert rec.slots[0] == 256 and rec.slots[3] == 16 and rec.slots[6] == 6 and rec.slots[1] == 64
    rec.note('alm_memset_zero', (w.data_ptr(), 2, 0))           # rows 0..3 of w are NOT part of the registered view
    assert rec.failed is not None and 'no buffer' in rec.failed


def test_recorder_refuses_other_streams_and_foreign_entry_points():
    arena = torch.empty(1024, dtype=torch.uint8)
    rec = LL.Recorder([(arena.data_ptr(), 1024)], stream=0)
    rec.note('alm_memset_zero', (arena.data_ptr(), 4, 12345))
    assert 'another stream' in rec.failed
    rec = LL.Recorder([(arena.data_ptr(), 1024)], stream=0)
    rec.note('alm_pack_weights_multi', (arena.data_ptr(), 1, 0))
    assert 'not a launch-list entry point' in rec.failed


def test_host_arrays_follow_the_entry():
    arena = torch.empty(8192, dtype=torch.uint8)
    base = arena.data_ptr()
    rec = LL.Recorder([(base, 8192)], stream=0)
    nb = 3
    ptrs = (ctypes.c_void_p * nb)(base + 0, base + 512, base + 1024)
    rows = (ctypes.c_int * nb)(7, 8, 9)
    sig = _lib.SIGNATURES['alm_hc_param_grads_batched']
    args = [ptrs, rows, nb, ptrs, ptrs, ptrs, base + 2048, ptrs, 4, 256, 0]
    assert len(args) == len(sig)
    rec.note('alm_hc_param_grads_batched', tuple(args))
    assert rec.failed is None
    first = 0
    assert rec.reloc[first] == LL.HOST_PTRS and rec.reloc[first + 1] == LL.HOST_INTS
    at = rec.slots[first]
    assert at == len(sig) and rec.reloc[at:at + 3] == [1, 1, 1] and rec.slots[at:at + 3] == [0, 512, 1024]
    ints = rec.slots[rec.slots[first + 1]:rec.slots[first + 1] + 2]
    assert ints[0] == (7 | (8 << 32)) and ints[1] == 9


def _run(entries, slots, reloc, bases, stream=0):
    E = (_lib.AlmListEntry * len(entries))(*[_lib.AlmListEntry(*e, 0) for e in entries])
    S = (ctypes.c_ulonglong * len(slots))(*slots)
    R = (ctypes.c_ushort * len(reloc))(*reloc)
    B = (ctypes.c_ulonglong * max(len(bases), 1))(*bases)
    failed = ctypes.c_int(-7)
    rc = _lib.query('alm_list_run', ctypes.addressof(E), len(entries), ctypes.addressof(S), ctypes.addressof(R), len(slots), ctypes.addressof(B), len(bases), stream,
                    ctypes.addressof(failed))
    return rc, failed.value


def test_list_run_resolves_slots_and_stops_at_the_first_failure():
    op = _lib.query('alm_list_op_id', b'alm_memset_zero')
    # (ptr = bases[0] + 64, 0 bytes, stream): fine without a GPU -- nothing is launched for zero bytes
    ok = [(op, 3, 0)], [64, 0, 0], [1, LL.LITERAL, LL.STREAM], [4096]
    assert _run(*ok) == (0, -1)
    # second entry: NULL + 8 bytes -> ALM_ERR_BAD_ARG from the entry point itself, reported with its index
    rc, at = _run([(op, 3, 0), (op, 3, 3)], [64, 0, 0, 0, 8, 0], [1, 0, LL.STREAM, 0, 0, LL.STREAM], [4096])
    assert (rc, at) == (10001, 1)
    # negative byte count passes through sign extension
    rc, at = _run([(op, 3, 0)], [64, (1 << 64) - 5, 0], [1, 0, LL.STREAM], [4096])
    assert (rc, at) == (10001, 0)
    # malformed lists are refused before anything is issued
    assert _run([(op, 2, 0)], [0, 0, 0], [0, 0, LL.STREAM], [])[0] == 10001            # wrong argument count
    assert _run([(op, 3, 1)], [0, 0, 0], [0, 0, LL.STREAM], [])[0] == 10001            # slots out of range
    assert _run([(op, 3, 0)], [0, 0, 0], [5, 0, LL.STREAM], [4096])[0] == 10001        # base id beyond the table
    assert _run([(99999, 3, 0)], [0, 0, 0], [0, 0, LL.STREAM], [])[0] == 10001


def test_arena_layout_matches_the_tally():
    t = LL._Tally()
    shapes = [((3, 5), torch.float32), (7, torch.bfloat16), ((2, 2, 2), torch.uint8), ((0,), torch.float32), ((129,), torch.float32)]
    for s, d in shapes:
        t.empty(s, d, 'cpu')
    a = LL._Arena(t.bytes, 'cpu')
    outs = [a.empty(s, d, 'cpu') for s, d in shapes]
    assert a.off == t.bytes <= a.cap
    for o, (s, d) in zip(outs, shapes):
        assert o.dtype == d and tuple(o.shape) == ((s,) if isinstance(s, int) else tuple(s))
        assert (o.data_ptr() - a.base) % LL.ALIGN == 0 or o.numel() == 0
    spec = LL._spec(outs[0], a, LL.Recorder([], 0))
    again = LL._rebuild(spec, {torch.float32: a.buf.view(torch.float32)})
    assert again.data_ptr() == outs[0].data_ptr() and again.shape == outs[0].shape


def test_plan_table_is_bounded_least_recently_used(monkeypatch):
    """data-dependent sequence lengths (unique_consecutive, ragged batches) meet a new (configuration, shape) key almost every step: the table keeps the
    MAX_PLANS most recently used plans; a plan that is still referenced by a forward in flight survives its eviction as an object"""
    import dataclasses
    import types

    from audiolm_pytorch_amd import ops

    @dataclasses.dataclass
    class Cfg:
        depth: int = 2

    core = types.SimpleNamespace(**{k: 0 for k in ('QKV_GROUP', 'QKV_GROUP_MAX_M', 'ASYNC_KV', 'DEFER_WGRAD', 'DEFER_GROUPS', 'DEFER_GROUP_SIZES', 'DEFER_MAX_BYTES',
                                                   'HC_BATCH_FINISH', 'PACK_ALL', 'ASYNC_WGRAD', 'SIDE_STREAMS')})
    monkeypatch.setattr(ops, '_st', lambda: 0)
    monkeypatch.setattr(LL, 'PLANS', {})
    monkeypatch.setattr(LL, 'MAX_PLANS', 4)
    monkeypatch.setattr(LL, 'ENABLED', True)
    plans = [LL.plan_for(core, torch.empty((1, n, 8)), None, Cfg(), True, None, 10, False, 1.0) for n in range(1, 7)]
    assert len(LL.PLANS) == 4 and all(p is not None for p in plans)
    assert [k[2][1] for k in LL.PLANS] == [3, 4, 5, 6]                       # the two oldest shapes were dropped
    again = LL.plan_for(core, torch.empty((1, 3, 8)), None, Cfg(), True, None, 10, False, 1.0)
    assert again is plans[2] and [k[2][1] for k in LL.PLANS] == [4, 5, 6, 3]    # a hit becomes the most recent
    LL.plan_for(core, torch.empty((1, 9, 8)), None, Cfg(), True, None, 10, False, 1.0)
    assert [k[2][1] for k in LL.PLANS] == [5, 6, 3, 9]
    fresh = LL.plan_for(core, torch.empty((1, 1, 8)), None, Cfg(), True, None, 10, False, 1.0)
    assert fresh is not plans[0] and fresh.state == 'size'                     # an evicted key starts over
