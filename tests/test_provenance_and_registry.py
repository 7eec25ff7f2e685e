"""CPU tests of two pieces of round-4 host logic that no GPU is needed for.

1. bench.pmc_traffic(): `roofline.traffic` is read from a COMMITTED PMC summary, so the summary must describe the kernels of this run.  It carries the digest
   of csrc/ + the C-ABI header it was measured on; a missing or different digest -> traffic None with the reason (never a stale number).
2. core.pack_target / stamp_packed: the hand-over between FusedAdam (which can write a dense weight's packed bf16 images while it updates the master,
   alm_opt_adam_pack_step) and the weight cache.  The optimiser may only take the fused path while the registered images are CURRENT: same tensor, same
   version, same shape, the cache still holding exactly those image tensors."""
import importlib.util
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('_alm_bench', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_summary_is_refused_unless_its_source_digest_matches(tmp_path, monkeypatch):
    bench = _bench()
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'csrc_digest', lambda: 'aaaa000011112222')
    b, name, note = bench.pmc_traffic()
    assert b is None and name is None and 'no PMC summary' in note
    (prof / 'r3_pmc_summary.json').write_text(json.dumps(dict(hbm_bytes_per_launch=123)))             # pre-round-4 file: no digest
    b, name, note = bench.pmc_traffic()
    assert b is None and name == 'r3_pmc_summary.json' and 'no source digest' in note
    (prof / 'r4_pmc_summary.json').write_text(json.dumps(dict(hbm_bytes_per_launch=456, commit='deadbee', csrc_digest='bbbb000011112222')))
    b, name, note = bench.pmc_traffic()
    assert b is None and name == 'r4_pmc_summary.json' and 'refused' in note and 'deadbee' in note
    (prof / 'r4_pmc_summary.json').write_text(json.dumps(dict(hbm_bytes_per_launch=456, commit='deadbee', csrc_digest='aaaa000011112222')))
    b, name, note = bench.pmc_traffic()
    assert b == 456 and 'deadbee' in note and 'aaaa000011112222' in note


def test_committed_pmc_summary_describes_the_committed_kernel_sources():
    """the evidence chain of the tree itself: the newest profiles/r*_pmc_summary.json was measured on exactly the csrc/ this checkout holds"""
    bench = _bench()
    name = 'r6_pmc_summary.json' if os.path.exists(os.path.join(ROOT, 'profiles', 'r6_pmc_summary.json')) else 'r5_pmc_summary.json'
    with open(os.path.join(ROOT, 'profiles', name)) as fh:
        d = json.load(fh)
    assert d.get('commit') and d.get('csrc_digest')
    if d['csrc_digest'] != bench.csrc_digest():
        # not an error of the code: the kernels were edited after the last PMC visit -- bench.py then reports `traffic: null` with the reason, and the next
        # round-end visit (scripts/gpu_final.sh) refreshes the summary
        pytest.skip(f"profiles/{name} is stale (measured at {d['commit']}): bench.py will report roofline.traffic = null until it is regenerated")


def test_pack_registry_hands_over_only_current_images():
    sys.path.insert(0, ROOT)
    from audiolm_pytorch_amd import core
    cache = core.WeightCache()
    w = torch.nn.Parameter(torch.randn(16, 8))
    out = (torch.zeros(16, 8, dtype=torch.bfloat16), torch.zeros(8, 16, dtype=torch.bfloat16))
    key = (0, 'attn', 'wq')
    stamp = (w.data_ptr(), core.tensor_version(w), tuple(w.shape))
    cache.store[key] = (stamp, out)
    core._register_pack(w.detach(), stamp, cache, key, out, [(0, 16, 8, out[0], out[1], 16, 8)])
    old = core.FUSED_ADAM_PACK
    try:
        core.FUSED_ADAM_PACK = True
        e = core.pack_target(w)
        assert e is not None and e['jobs'][0][:3] == (0, 16, 8)
        # another in-place update of the master (anything but the fused step): the images are stale
        with torch.no_grad():
            w.add_(1.0)
        assert core.pack_target(w) is None
        # the fused step's bookkeeping: re-stamp with the advanced version -> current again, and the cache entry follows
        core.stamp_packed(w, e)
        assert core.pack_target(w) is e and cache.store[key][0] == (w.data_ptr(), core.tensor_version(w), tuple(w.shape))
        # the cache re-packed into other tensors (or was cleared): the registered images are not the ones the forward will read
        cache.store[key] = (cache.store[key][0], (out[0].clone(), out[1].clone()))
        assert core.pack_target(w) is None
        cache.store.clear()
        assert core.pack_target(w) is None
        # switch off
        cache.store[key] = (e['stamp'], out)
        assert core.pack_target(w) is e
        core.FUSED_ADAM_PACK = False
        assert core.pack_target(w) is None
        # a dead cache
        core.FUSED_ADAM_PACK = True
        del cache
        import gc
        gc.collect()
        assert core.pack_target(w) is None
    finally:
        core.FUSED_ADAM_PACK = old


def test_pack_registry_entries_die_with_their_cache():
    """_PACK_REGISTRY holds strong references to the packed bf16 images; they were pruned only above 4096 keys, so every rebuilt model pinned its
    dead images.  Now a finaliser on the WeightCache drops its entries as soon as the cache (the model) is collected; other caches' entries stay."""
    import gc

    import torch

    from audiolm_pytorch_amd import core
    before = dict(core._PACK_REGISTRY)
    c1, c2 = core.WeightCache(), core.WeightCache()
    w1, w2 = torch.zeros(4, 4), torch.zeros(4, 4)
    core._register_pack(w1, (w1.data_ptr(), 0, (4, 4)), c1, (0, 'attn', 'wq'), (w1, w1), [])
    core._register_pack(w2, (w2.data_ptr(), 0, (4, 4)), c2, (0, 'attn', 'wq'), (w2, w2), [])
    assert w1.data_ptr() in core._PACK_REGISTRY and w2.data_ptr() in core._PACK_REGISTRY
    del c1
    gc.collect()
    assert w1.data_ptr() not in core._PACK_REGISTRY and w2.data_ptr() in core._PACK_REGISTRY
    del c2
    gc.collect()
    assert dict(core._PACK_REGISTRY).keys() == before.keys()
