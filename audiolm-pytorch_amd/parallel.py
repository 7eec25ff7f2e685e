"""Data-parallel gradient exchange for the token-transformer training path: one process per GPU, RCCL over xGMI.

Re-creates what the reference gets implicitly from HF accelerate -> torch DDP -> NCCL (trainer.py:56-57, 75, 1132-1140, 1247;
SURVEY.md §2 row 8 / §8(e)) for THIS path only: pure data parallelism, one gradient all-reduce (mean) per step, overlapped with
backward.  It is not a general DDP replacement.

  * the path shards over batch rows (independent sequences); the only exchange step is the gradient mean
  * buckets follow the backward order of the fused stack: [logit heads] -> layer groups from the top down -> [embeddings / start tokens].
    Round 4: the stack computes its weight gradients per GROUP of layers (core.DP_DEFER_GROUPS, default 2 groups of 3 layers: the layer-batched TN
    launches of the single-GPU step) and hands the whole group over at once (`on_group`): ONE bucket and ONE collective per group (115 MB fp32 at
    dim 1024 depth 6 -- a ring over xGMI is per-link bound, few large collectives beat many small ones), issued from the side stream right behind the
    group's GEMMs, so the upper group's RCCL traffic runs under the lower layers' backward and only the last group's is exposed.
    ALM_DP_DEFER_GROUPS=0 restores the round-3 behaviour: per-layer split-K weight gradients and one 38 MB bucket per layer
  * parameters that never receive a gradient (proj_text_embed -- the reason the reference needs find_unused_parameters=True,
    trainer.py:75) are simply never bucketed; DDP's per-forward buffer broadcast is skipped (only constant zero `beta` buffers exist)
  * `finish()` must be called after backward(): it waits for the collectives and writes the averaged gradients back into p.grad
  * gradient accumulation (trainer.py:1237 `no_sync`): `with engine.no_sync():` micro-steps only accumulate locally; the next synchronising
    step then reduces the ACCUMULATED p.grad in `finish()` (bucketed per layer, not overlapped: the fresh per-layer gradients handed to the
    callback are only that micro-step's share)
  * DDP semantics for gradients that are already there: if a synchronising backward starts while parameters still hold a `.grad` (accumulation
    without no_sync(), or `zero_grad(set_to_none=False)`), autograd ADDS this backward's share to it -- the overlapped buckets only carry the
    fresh share, so that backward is exchanged like an accumulated one (p.grad itself is reduced in `finish()`, per layer, not overlapped).
    `zero_grad()` with torch's default set_to_none=True keeps the overlap.
  * `bucket_dtype=torch.bfloat16` halves the bytes on the xGMI links (the reduction then runs in bf16; fp32 is the default, like DDP;
    bench.py --bucket-dtype bf16 selects it as a labelled variant: a ring over 7 x ~153 GB/s xGMI links is per-link bound, 131 instead of
    262 MB per step -- the default scaling run reduces fp32 like the reference)
  * `force_collectives=True` issues the collectives even in a 1-rank group (they are the identity there): the way the RCCL hand-off -- bucket copy
    and asynchronous all-reduce issued from the backward's side stream, `work.wait()` ordering the main stream in `finish()` -- is executed on a
    1-GPU box (tests/test_gpu_dp.py)
  * `stats` (per step, reset by `finish()`; the finished step's copy is `last_stats`): bucket count, bytes put on the wire, the HIP streams the
    collectives were issued from, host time from the last bucket launch to the return of `finish()`
    (what the step still waited for: the tail that did not overlap)

  * round 6 -- the exposed tail: (i) uneven layer groups (ALM_DP_GROUP_SIZES=4,2, core.group_starts): the last group's bucket, which starts when the
    backward is over, is the small one; (ii) the bucket of the gradients that exist BEFORE the stack's backward (logit heads, final norm) is handed over
    when the first group's GEMMs are about to be launched (`group_buffer`), not behind them; (iii) `last_stats['tail_model_ms']`: bytes of the buckets
    launched after the last layer group's GEMMs over the ring (one xGMI link, ~153 GB/s) and direct (7 links) all-reduce rates of SURVEY.md section 5,
    for an 8-GPU node -- a MODEL (no N > 1 hardware run has been made), next to the measured single-GPU `exposed_tail_ms`
  * THE .grad-IS-A-VIEW CONTRACT of the in-place buckets: `finish()` hands out `p.grad` tensors that are VIEWS of the engine's persistent per-group flat
    buffers; the next synchronising backward's weight-gradient GEMMs overwrite those buffers.  `optimizer.step()` + `zero_grad()` (set_to_none or not)
    is fine; a caller that KEEPS a gradient tensor across steps (gradient statistics, an EMA of gradients, a held reference after
    zero_grad(set_to_none=True)) would see it silently rewritten -- so the engine RAISES at the start of the next backward when a handed-out view is
    still referenced by anything but `p.grad` (clone what you want to keep; ALM_DP_DIRECT=0 stages every bucket instead and hands out copies)

Works with any torch.distributed backend (`nccl` == RCCL on ROCm; `gloo` for the CPU tests).
"""
from __future__ import annotations

import contextlib
import os
import sys
import time

import torch


class _Bucket:
    __slots__ = ('params', 'grads', 'flat', 'work', 'direct')

    def __init__(self):
        self.params, self.grads, self.flat, self.work, self.direct = [], [], None, None, None


XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7          # 7 point-to-point links of ~153 GB/s per GPU on an 8 x MI355X node


def tail_model(nbytes, world=8):
    """MODEL of the all-reduce time of `nbytes` that nothing hides, on one `world`-GPU xGMI node: a ring moves 2 (G - 1) / G of the bytes over ONE link per
    GPU, a direct (one-shot reduce-scatter + all-gather) exchange spreads them over all 7 links.  SURVEY.md section 5 quotes 3.0 / 0.43 ms for 262 MB."""
    vol = 2.0 * (world - 1) / world * nbytes
    return dict(world=world, bytes=int(nbytes), ring=round(vol / (XGMI_LINK_GBS * 1e9) * 1e3, 3), direct=round(vol / (XGMI_LINK_GBS * XGMI_LINKS * 1e9) * 1e3, 3))


class DataParallelEngine:
    def __init__(self, model, dist, process_group=None, broadcast_parameters=True, bucket_dtype=torch.float32, force_collectives=False):
        assert bucket_dtype in (torch.float32, torch.bfloat16), bucket_dtype
        self.model, self.dist, self.pg, self.bucket_dtype = model, dist, process_group, bucket_dtype
        self.world = dist.get_world_size(process_group)
        self.force = bool(force_collectives)
        self.stats = dict(buckets=0, bytes=0, tail_ms=0.0, launch_streams=[], bucket_bytes=[])
        self._t_last_launch = None
        self._avg_ok = str(dist.get_backend(process_group)).lower() == 'nccl'
        self.params = [p for p in model.parameters() if p.requires_grad]
        if broadcast_parameters and self.world > 1:
            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p.data, src=0, group=process_group)
        self._flat_cache = {}
        self._flat_groups = {}                # (l0, l1) -> (flat fp32 buffer, {id(param): (offset, numel)}): see group_buffer()
        self.direct_buckets = os.environ.get('ALM_DP_DIRECT', '1') != '0'     # A/B switch: 0 = every bucket staged through torch.cat + copied back
        self._ev_bw_done = self._ev_all_done = None
        self._pending_flat = None
        self._inflight = []
        self._handed = []                     # (param, view) pairs handed out as .grad by the last finish() (in-place buckets): see _check_handed_views
        self._exposed_from = None             # index into stats['bucket_bytes'] of the first bucket launched after the last layer group's GEMMs
        self._loose = _Bucket()               # parameters outside the fused stack (heads first, embeddings last)
        self._stack_param_ids = set()
        tr = getattr(model, 'transformer', None)
        self._stack = tr if (tr is not None and hasattr(tr, '_layer_grad_hook')) else None
        if self._stack is not None:
            self._stack._layer_grad_hook = self                      # callable per layer (__call__) and per layer group (on_group): see core.stack_backward
            self._stack_flat = self._stack.flat_params()
            self._stack_param_ids = {id(p) for p in self._stack_flat}
            self._ppl = (len(self._stack_flat) - 1) // self._stack.depth
        self._hooks = []
        for p in self.params:
            if id(p) not in self._stack_param_ids or (self._stack is not None and p is self._stack_flat[-1]):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_loose_grad))
        self._n_loose_heads = None
        self._sync = True                     # False inside no_sync()
        self._dirty = False                   # p.grad holds local, not yet reduced micro-step gradients
        self._bw_started = False              # a backward is in progress (first gradient callback seen, finish() not yet called)
        self._stale = False                   # that backward started with gradients already present: exchange p.grad itself afterwards
        self.last_stats = None                # stats of the last synchronising step (see the module docstring)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient-accumulation micro-steps: backward() + finish() inside this context exchange nothing."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def _begin_backward(self, skip=None):
        """first gradient callback of a backward: do parameters (other than the one whose hook fired) still hold a gradient from before?"""
        if not self._bw_started:
            self._bw_started = True
            self._stale = any(q.grad is not None for q in self.params if q is not skip)
            self._check_handed_views()

    def _check_handed_views(self):
        """the .grad-is-a-view contract (module docstring): the in-place bucket views handed out by the last finish() are about to be overwritten by this
        backward's GEMMs.  A view that is still `p.grad` is handled (`_stale`: this backward takes the staged path); a view that ANYTHING ELSE still
        references would be silently rewritten -> raise."""
        handed, self._handed = self._handed, []
        if not self._sync:
            self._handed = handed                                  # (an accumulation micro-step writes no bucket: check again at the next synchronising backward)
            return
        for i in range(len(handed)):
            p, v = handed[i]
            handed[i] = None
            extra = sys.getrefcount(v) - 2 - (1 if p.grad is v else 0)      # 2 = the local name `v` + getrefcount's own argument
            if extra > 0:
                raise RuntimeError('DataParallelEngine: a gradient tensor handed out as `.grad` by the previous finish() is still referenced elsewhere. '
                                   'It is a VIEW of a persistent all-reduce bucket that this backward overwrites in place -- clone gradients you keep across '
                                   'steps, or set ALM_DP_DIRECT=0 (staged buckets, .grad are copies).')

    def _overlapped(self):
        return self._sync and not self._dirty and not self._stale

    # ---- bucket launch -------------------------------------------------------------------------------------------
    def _launch(self, key, params, grads):
        if (self.world == 1 and not self.force) or not params:
            return
        n = sum(g.numel() for g in grads)
        flat = self._flat_cache.get(key)
        if flat is None or flat.numel() != n or flat.device != grads[0].device:
            flat = torch.empty(n, dtype=self.bucket_dtype, device=grads[0].device)
            self._flat_cache[key] = flat
        torch.cat([g.reshape(-1).to(self.bucket_dtype) for g in grads], out=flat)
        op = self.dist.ReduceOp.AVG if self._avg_ok else self.dist.ReduceOp.SUM       # gloo has no AVG: sum, divide in finish()
        work = self.dist.all_reduce(flat, op=op, group=self.pg, async_op=True)
        self.stats['buckets'] += 1
        self.stats['bytes'] += flat.numel() * flat.element_size()
        self.stats['bucket_bytes'].append(flat.numel() * flat.element_size())
        if flat.is_cuda:
            sid = int(torch.cuda.current_stream(flat.device).cuda_stream)
            if sid not in self.stats['launch_streams']:
                self.stats['launch_streams'].append(sid)
        self._t_last_launch = time.perf_counter()
        b = _Bucket()
        b.params, b.flat, b.work = list(params), flat, work
        b.grads = [tuple(g.shape) for g in grads]
        self._inflight.append((b, op))

    def _on_layer_grads(self, layer, grads):
        """core.stack_backward callback: `grads` are the fresh gradients of layer `layer` (order == Transformer.flat_params)."""
        self._begin_backward()
        if not self._overlapped():
            return
        self._flush_loose(('loose', 'pre', layer))       # whatever accumulated so far (logit heads, final norm) goes first
        params = self._stack_flat[layer * self._ppl:(layer + 1) * self._ppl]
        pg = [(p, g) for p, g in zip(params, grads) if g is not None and p.requires_grad]
        self._launch(('layer', layer), [p for p, _ in pg], [g for _, g in pg])

    __call__ = _on_layer_grads

    def group_buffer(self, l0, l1, dense_slots, late_slots=(), skip_slots=()):
        """core.stack_backward (deferred mode) asks, right before it launches the batched weight-gradient GEMMs of layers [l0, l1): where should they
        write?  `dense_slots`: per weight kind computed for THESE layers, the index of that weight within a layer's flat parameter list; `late_slots`:
        kinds computed with this (the last) group for ALL layers of the stack; `skip_slots`: kinds of these layers that are NOT part of this group's
        bucket (they follow with the last group).  -> per kind (dense, then late) one fp32 view [layers, *weight.shape] into this group's PERSISTENT
        flat bucket, or None.  With the views the GEMMs produce the bucket in place: `on_group` only adds the group's few small gradients
        (hyper-connection parameters, norm gains) to the tail and all-reduces the flat buffer, `finish()` hands out views as `.grad` -- no torch.cat
        staging pass and no copy back (2 x 262 MB of HBM traffic per step at dim 1024 depth 6).  Only for the overlapped fp32 path: while gradients
        accumulate (no_sync / stale .grad) a `.grad` may alias the bucket the next backward would overwrite -> None, the staged path runs."""
        self._begin_backward()
        self._pending_flat = None
        if self._overlapped():
            # round 6: whatever exists BEFORE the stack's backward (logit heads, final norm) goes on the wire now -- ahead of this group's GEMMs, not behind them
            self._flush_loose(('loose', 'pre', l1))
        if not (self.direct_buckets and self.bucket_dtype == torch.float32 and self._overlapped() and (self.world > 1 or self.force)):
            return None
        L, ppl = self._stack.depth, self._ppl
        dense = [[self._stack_flat[l * ppl + slot] for l in range(l0, l1)] for slot in dense_slots]      # (the engine's own parameter objects)
        dense += [[self._stack_flat[l * ppl + slot] for l in range(L)] for slot in late_slots]
        key = (l0, l1, tuple(dense_slots), tuple(late_slots), tuple(skip_slots))
        ent = self._flat_groups.get(key)
        dev = dense[0][0].device
        if ent is None or ent[0].device != dev:
            offs, o = {}, 0
            for plist in dense:                                   # dense kinds first, each kind's layers contiguous: one stacked GEMM output per kind
                for p in plist:
                    offs[id(p)] = (o, p.numel())
                    o += p.numel()
                o = (o + 3) // 4 * 4                              # 16-byte aligned starts (the GEMM epilogues store 16-byte vectors)
            skip = {id(self._stack_flat[layer * ppl + slot]) for layer in range(l0, l1) for slot in skip_slots}
            for layer in range(l0, l1):                           # then every other trainable parameter of the group's layers
                for p in self._stack_flat[layer * ppl:(layer + 1) * ppl]:
                    if p.requires_grad and id(p) not in offs and id(p) not in skip:
                        offs[id(p)] = (o, p.numel())
                        o += p.numel()
            ent = (torch.zeros(o, dtype=torch.float32, device=dev), offs)      # (zeros: the alignment gaps between kinds take part in the all-reduce)
            self._flat_groups = {k: v for k, v in self._flat_groups.items() if k[:2] != (l0, l1)}        # one layout per layer range
            self._flat_groups[key] = ent
        flat, offs = ent
        views = []
        for plist in dense:
            o0 = offs[id(plist[0])][0]
            n = plist[0].numel()
            if any(offs[id(p)] != (o0 + i * n, n) or not p.requires_grad or p.shape != plist[0].shape for i, p in enumerate(plist)):
                return None                                       # (a frozen dense weight in the group: staged path)
            views.append(flat[o0:o0 + n * len(plist)].view(len(plist), *plist[0].shape))
        self._pending_flat = ent                                  # the on_group() call that follows is this group's
        return views

    def on_group(self, layers, grads_per_layer):
        """core.stack_backward callback of the deferred (layer-batched) weight-gradient mode: the fresh gradients of a GROUP of layers, `layers` in
        backward order -- one bucket, one collective for the whole group."""
        self._begin_backward()
        if not self._overlapped():
            return
        self._flush_loose(('loose', 'pre', layers[0]))
        if 0 in layers:                                           # the LAST group: everything launched from here on starts after the backward's last kernel
            self._exposed_from = len(self.stats['bucket_bytes'])
        params, grads = [], []
        for layer, gl in zip(layers, grads_per_layer):
            for p, g in zip(self._stack_flat[layer * self._ppl:(layer + 1) * self._ppl], gl):
                if g is not None and p.requires_grad:
                    params.append(p)
                    grads.append(g)
        ent, self._pending_flat = self._pending_flat, None
        if ent is not None and params and (self.world > 1 or self.force):
            flat, offs = ent
            base, esz = flat.data_ptr(), flat.element_size()
            inplace = [id(p) in offs and g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() == base + offs[id(p)][0] * esz for p, g in zip(params, grads)]
            covered = sum(offs[id(p)][1] for p in params if id(p) in offs)
            if any(inplace) and all(id(p) in offs for p in params) and covered == sum(n for _, n in offs.values()):
                # the dense gradients already sit in the bucket (written there by the GEMMs); the rest is a handful of small tensors: one fused copy
                rest = [(flat[offs[id(p)][0]:offs[id(p)][0] + offs[id(p)][1]].view(g.shape), g) for p, g, here in zip(params, grads, inplace) if not here]
                if rest:
                    torch._foreach_copy_([d for d, _ in rest], [g for _, g in rest])
                self._launch_flat(flat, params, [tuple(g.shape) for g in grads], direct={id(p): offs[id(p)] for p in params})
                return
        self._launch(('group', layers[0], len(layers)), params, grads)

    def _launch_flat(self, flat, params, shapes, direct):
        op = self.dist.ReduceOp.AVG if self._avg_ok else self.dist.ReduceOp.SUM
        work = self.dist.all_reduce(flat, op=op, group=self.pg, async_op=True)
        self.stats['buckets'] += 1
        self.stats['bytes'] += flat.numel() * flat.element_size()
        self.stats['bucket_bytes'].append(flat.numel() * flat.element_size())
        self.stats['direct_buckets'] = self.stats.get('direct_buckets', 0) + 1
        if flat.is_cuda:
            sid = int(torch.cuda.current_stream(flat.device).cuda_stream)
            if sid not in self.stats['launch_streams']:
                self.stats['launch_streams'].append(sid)
        self._t_last_launch = time.perf_counter()
        b = _Bucket()
        b.params, b.flat, b.work, b.grads, b.direct = list(params), flat, work, list(shapes), direct
        self._inflight.append((b, op))

    def _on_loose_grad(self, p):
        self._begin_backward(skip=p)
        if self._overlapped():
            self._loose.params.append(p)

    def _flush_loose(self, key):
        if self._loose.params:
            ps = self._loose.params
            self._loose = _Bucket()
            self._launch(key, ps, [p.grad for p in ps])

    # ---- end of backward ------------------------------------------------------------------------------------------
    def finish(self):
        """Call after loss.backward(): waits for every in-flight all-reduce and stores the mean gradients in p.grad."""
        stale, self._bw_started, self._stale = self._stale, False, False
        if not self._sync:
            self._dirty = True
            return
        if self._dirty or stale:                                      # accumulated gradients: reduce p.grad itself, layer by layer
            self._dirty = False
            groups = []
            if self._stack is not None:
                loose = [p for p in self.params if id(p) not in self._stack_param_ids or p is self._stack_flat[-1]]
                groups.append(('acc', 'loose', loose))
                for layer in reversed(range(self._stack.depth)):
                    groups.append(('acc', layer, self._stack_flat[layer * self._ppl:(layer + 1) * self._ppl]))
            else:
                groups.append(('acc', 'all', self.params))
            for *key, params in groups:
                ps = [p for p in params if p.requires_grad and p.grad is not None]
                self._launch(tuple(key), ps, [p.grad for p in ps])
        self._flush_loose(('loose', 'post'))
        cuda = bool(self._inflight) and self._inflight[0][0].flat.is_cuda
        if cuda:
            # exposed tail on the GPU clock: from "backward's own kernels are done" (everything queued on the current stream so far) to "every bucket is
            # reduced and handed out"; read lazily in `last_stats` (an event query needs a sync: never inside the step)
            self._ev_bw_done = torch.cuda.Event(enable_timing=True)
            self._ev_all_done = torch.cuda.Event(enable_timing=True)
            self._ev_bw_done.record()
        for b, op in self._inflight:
            b.work.wait()
            if op == self.dist.ReduceOp.SUM:
                b.flat.div_(self.world)
            if b.direct is not None:                                   # in-place bucket: the reduced gradients ARE the bucket -- hand out views
                for p, shape in zip(b.params, b.grads):
                    o, n = b.direct[id(p)]
                    v = b.flat[o:o + n].view(shape)
                    p.grad = v
                    self._handed.append((p, v))
                    del v
                continue
            views, o = [], 0
            for p, shape in zip(b.params, b.grads):
                n = 1
                for d in shape:
                    n *= d
                views.append(b.flat[o:o + n].view(shape))
                o += n
            have = [(p, v) for p, v in zip(b.params, views) if p.grad is not None]
            if have:                                                   # one fused launch per bucket instead of one copy per parameter
                torch._foreach_copy_([p.grad for p, _ in have], [v for _, v in have])
            for p, v in zip(b.params, views):
                if p.grad is None:
                    p.grad = v.to(p.dtype, copy=True)                  # a bf16 bucket must not become the .grad of an fp32 parameter
        if cuda:
            self._ev_all_done.record()
        if self._inflight and self._t_last_launch is not None:
            self.last_stats = dict(self.stats, tail_ms=round((time.perf_counter() - self._t_last_launch) * 1e3, 3))
            bb = self.stats['bucket_bytes']
            ef = self._exposed_from if self._exposed_from is not None else max(0, len(bb) - 1)
            self.last_stats['tail_model_ms'] = tail_model(sum(bb[ef:]))
            self.last_stats['exposed_bucket_bytes'] = bb[ef:]
        self._exposed_from = None
        self.stats = dict(buckets=0, bytes=0, tail_ms=0.0, launch_streams=[], bucket_bytes=[])
        self._t_last_launch = None
        self._inflight = []

    def exposed_tail_ms(self):
        """GPU time of the last synchronising step between the end of backward's own work and the last bucket being reduced and handed out -- the part
        of the gradient exchange that did NOT hide under backward.  Synchronises on the recorded events: call outside the timed region."""
        if self._ev_bw_done is None or self._ev_all_done is None:
            return None
        self._ev_all_done.synchronize()
        return round(self._ev_bw_done.elapsed_time(self._ev_all_done), 3)

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self._stack is not None:
            self._stack._layer_grad_hook = None
