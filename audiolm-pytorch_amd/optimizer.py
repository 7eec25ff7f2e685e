"""Mirror of the reference's `audiolm_pytorch/optimizer.py` (get_optimizer, optimizer.py:11-37) on a fused MI355X optimiser step.

`get_optimizer(params, lr, wd, betas, eps, filter_by_requires_grad, group_wd_params)` keeps the reference's signature and grouping rule
(weight decay only for parameters with ndim >= 2, optimizer.py:3-8 / :25-33; wd == 0 -> Adam, else AdamW) and returns a `FusedAdam`:
a torch.optim.Optimizer whose `step()` is two HIP launches over EVERY parameter (csrc/optim.hip) and whose state_dict has torch.optim.Adam's
layout ('step', 'exp_avg', 'exp_avg_sq'), so checkpoints interchange with the reference.  Round 4: the dense GEMM weights of the transformer stack
are updated by `alm_opt_adam_pack_step`, which also writes their packed bf16 images (W, W^T) -- the forward after a step no longer re-packs them
(core.pack_target / core.stamp_packed; `ALM_FUSED_ADAM_PACK=0` restores update + separate re-pack; both give the same bits).

The trainers' `accelerator.clip_grad_norm_(transformer.parameters(), max_grad_norm)` (trainer.py:953-954, :1251-1255) maps to
`optim.clip_grad_norm_(max_norm)`: it computes the global gradient norm ON THE DEVICE (returned as a 0-d tensor, no host sync) and the next
`step()` applies the clip coefficient to the gradients on the fly -- `p.grad` itself is left unscaled (nothing reads it after the step).
No CPU fallback: CPU parameters are refused.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib, ops

F32 = torch.float32


def separate_weight_decayable_params(params):                  # optimizer.py:3-8
    wd_params, no_wd_params = [], []
    for param in params:
        (no_wd_params if param.ndim < 2 else wd_params).append(param)
    return wd_params, no_wd_params


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0., decoupled_weight_decay=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled_weight_decay=decoupled_weight_decay)
        super().__init__(params, defaults)
        self._chunks = {}               # tag ('all' | 'plain') -> (key, [per group: device chunk table], [first chunk per group], chunk count)
        self._partial = None            # fp32 [chunks]: per-chunk sums of squares of the gradients
        self._prepared = None           # (key of gradient pointers, [per group: dict of tables], the host arrays' owner)
        self._pending_clip = None       # (max_norm, device scalar sum of squares)

    # ------------------------------------------------------------------------------------------------------------------ tables
    def _group_tensors(self, group):
        ps = [p for p in group['params'] if p.grad is not None]
        for p in ps:
            if not p.is_cuda or p.dtype != F32 or not p.is_contiguous():
                raise _lib.AlmError('FusedAdam updates contiguous fp32 parameters on the MI355X only (no CPU fallback)')
            if p.grad.dtype != F32 or not p.grad.is_contiguous():
                p.grad = p.grad.to(F32).contiguous()
            st = self.state[p]
            if len(st) != 0 and st['step'].is_cuda:              # a checkpoint loaded with map_location='cuda': keep the counter on the host
                st['step'] = st['step'].cpu()
            if len(st) == 0:
                st['step'] = torch.tensor(0., dtype=F32)         # torch.optim.Adam layout (host step counter, capturable=False)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return ps

    def _chunk_tables(self, tag, groups, dev):
        """cached device chunk tables (tensor index, chunk index) of a list of tensor groups -> ([per group: table | None], [first chunk of each group], chunks)"""
        ckey = tuple(tuple(p.numel() for p in ps) for ps in groups)
        hit = self._chunks.get(tag)
        if hit is None or hit[0] != ckey:
            ch = _lib.query('alm_opt_chunk_elems')
            tabs, offs, off = [], [], 0
            for ps in groups:
                pairs = [(i, c) for i, p in enumerate(ps) for c in range((p.numel() + ch - 1) // ch)]
                tabs.append(torch.tensor(pairs, dtype=torch.int32).reshape(-1, 2).to(dev) if pairs else None)
                offs.append(off)
                off += len(pairs)
            hit = (ckey, tabs, offs, off)
            self._chunks[tag] = hit
        return hit[1], hit[2], hit[3]

    def _table(self, groups, mixed):
        """HOST AlmOptTensor array over the tensors of `groups` -> (array, [per group: (address, count)])"""
        rec = ctypes.sizeof(_lib.AlmOptTensor)
        arr = (_lib.AlmOptTensor * max(sum(len(ps) for ps in groups), 1))()
        out, i = [], 0
        for group, ps, mx in zip(self.param_groups, groups, mixed):
            start = i
            for p in ps:
                st = self.state[p]
                # step: THIS parameter's count for the coming update (torch.optim.Adam tracks it per parameter: one whose gradient first appears
                # later, or is None on some steps, has its own bias corrections); 0 = the launch's `step` argument (every tensor of the group at
                # the same count: the usual case -- bias corrections computed once on the host, in double)
                arr[i] = _lib.AlmOptTensor(p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel(),
                                           float(group['weight_decay']), int(st['step']) + 1 if mx else 0)
                i += 1
            out.append((ctypes.addressof(arr) + start * rec, len(ps)))
        return arr, out

    def _pack_jobs_ok(self, p, e):
        """alm_opt_adam_pack_step's requirements (8-byte fp32 pairs, 4-byte bf16 pairs); a weight that does not meet them takes the plain path"""
        if p.dim() != 2 or not p.is_contiguous():
            return False
        cols = p.shape[1]
        for row0, rows, c, dst, dstT, rp, cp in e['jobs']:
            if c != cols or (cols | rp | cp | dst.stride(0) | dstT.stride(0)) & 1 or (dst.data_ptr() | dstT.data_ptr()) & 3:
                return False
        st = self.state[p]
        return (p.data_ptr() | p.grad.data_ptr() | st['exp_avg'].data_ptr() | st['exp_avg_sq'].data_ptr()) % 8 == 0

    def _prepare(self):
        """Per iteration: HOST tables of AlmOptTensor (the C side hands them to the kernels in their arguments, 64 tensors per launch: gradient
        storage changes every step, and a staged host-to-device copy per step is exactly what could stall the host behind the stream) + the cached
        device chunk tables.  Two tables: every tensor (the gradient norm) and the tensors updated by alm_opt_adam_step; the dense GEMM weights whose
        packed bf16 images are current (core.pack_target) are updated by alm_opt_adam_pack_step instead, which writes the images too.
        clip_grad_norm_() and step() of the same iteration share the result."""
        groups = [self._group_tensors(g) for g in self.param_groups]
        flat = [p for ps in groups for p in ps]
        if not flat:
            return None
        from . import core
        # the tables are shared by clip_grad_norm_() and step() of ONE iteration.  With the in-place data-parallel buckets the gradient pointers are the
        # same every step, so they alone do not identify the iteration (a clip_grad_norm_() without a following step() left tables with stale
        # step counts / pack targets for the next one): the key also carries every parameter's version counter (advanced by each update), the step counts
        # and the generation of the packed-weight registry.
        gkey = (tuple(p.grad.data_ptr() for p in flat), tuple(core.tensor_version(p) for p in flat),
                tuple(int(self.state[ps[0]]['step']) if ps else -1 for ps in groups), core._PACK_GEN[0])
        if self._prepared is not None and self._prepared[0] == gkey:
            return self._prepared[1]
        dev = flat[0].device
        mixed = [len({int(self.state[p]['step']) for p in ps}) > 1 for ps in groups]
        targets = [[core.pack_target(p) for p in ps] for ps in groups]
        targets = [[e if e is not None and self._pack_jobs_ok(p, e) else None for p, e in zip(ps, es)] for ps, es in zip(groups, targets)]
        plain = [[p for p, e in zip(ps, es) if e is None] for ps, es in zip(groups, targets)]
        any_packed = any(e is not None for es in targets for e in es)
        arr_all, tab_all = self._table(groups, mixed)
        ch_all, off_all, n_all = self._chunk_tables('all', groups, dev)
        if any_packed:
            arr_plain, tab_plain = self._table(plain, mixed)
            ch_plain, _, _ = self._chunk_tables('plain', plain, dev)
        else:
            arr_plain, tab_plain, ch_plain = arr_all, tab_all, ch_all
        if self._partial is None or self._partial.numel() < max(n_all, 1) or self._partial.device != dev:
            self._partial = torch.empty(max(n_all, 1), dtype=F32, device=dev)
        out, keep = [], [arr_all, arr_plain]
        for gi, (group, ps) in enumerate(zip(self.param_groups, groups)):
            jobs, packed = None, [(p, e) for p, e in zip(ps, targets[gi]) if e is not None]
            if packed:
                n = sum(len(e['jobs']) for _, e in packed)
                jarr = (_lib.AlmOptPackJob * n)()
                j = 0
                for p, e in packed:
                    st = self.state[p]
                    cols = p.shape[1]
                    for row0, rows, _, dst, dstT, rp, cp in e['jobs']:
                        o = row0 * cols * 4
                        jarr[j] = _lib.AlmOptPackJob(p.data_ptr() + o, p.grad.data_ptr() + o, st['exp_avg'].data_ptr() + o, st['exp_avg_sq'].data_ptr() + o,
                                                     rows, cols, cols, dst.data_ptr(), dst.stride(0), rp, cp, dstT.data_ptr(), dstT.stride(0),
                                                     float(group['weight_decay']), int(st['step']) + 1 if mixed[gi] else 0)
                        j += 1
                keep.append(jarr)
                jobs = (ctypes.addressof(jarr), n)
            out.append(dict(ps=ps, all=tab_all[gi], all_chunks=ch_all[gi], off=off_all[gi], plain=tab_plain[gi], plain_chunks=ch_plain[gi],
                            has_plain=len(plain[gi]) > 0, jobs=jobs, packed=packed))
        self._prepared = (gkey, out, keep)                        # `keep` owns the memory the addresses above point into
        return out

    # ------------------------------------------------------------------------------------------------------------------ API
    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """Global L2 norm of every gradient this optimiser owns -> 0-d device tensor (no host sync); the next step() clips by it
        (torch.nn.utils.clip_grad_norm_ semantics: coef = min(1, max_norm / (norm + 1e-6)))."""
        prep = self._prepare()
        if prep is None:
            return torch.zeros((), dtype=F32)
        partial = self._partial
        n = 0
        for g in prep:
            chunks = g['all_chunks']
            if chunks is None:
                continue
            _lib.call('alm_opt_grad_sumsq', g['all'][0], g['all'][1], chunks.data_ptr(), chunks.shape[0], partial.data_ptr() + 4 * g['off'], ops._st())
            n = g['off'] + chunks.shape[0]
        total = ops.reduce_sum(partial[:n])
        self._pending_clip = (float(max_norm), total)
        return total.sqrt()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip = self._pending_clip
        self._pending_clip = None
        prep = self._prepare()
        keep = self._prepared                                     # (holds the host tables alive until the launches below have copied them)
        self._prepared = None
        if prep is None:
            return loss
        from . import core
        for group, g in zip(self.param_groups, prep):
            ps = g['ps']
            if not ps:
                continue
            step = int(self.state[ps[0]]['step']) + 1            # used by the tensors whose table entry carries no step of its own
            b1, b2 = group['betas']
            hyper = (float(group['lr']), float(b1), float(b2), float(group['eps']), step, int(bool(group['decoupled_weight_decay'])),
                     clip[1].data_ptr() if clip else None, clip[0] if clip else 0.0, ops._st())
            # the fused pack step FIRST: it is the launch whose argument validation can refuse (ALM_ERR_BAD_ARG) -- if it raises, nothing of this group
            # has been applied yet (no step count bumped, no plain tensor updated: ADVICE r4).  The table entries carry step counts that _prepare()
            # computed as state + 1, so the counters are advanced only after both launches were accepted.
            if g['jobs'] is not None:
                _lib.call('alm_opt_adam_pack_step', g['jobs'][0], g['jobs'][1], *hyper)
            if g['has_plain']:
                chunks = g['plain_chunks']
                _lib.call('alm_opt_adam_step', g['plain'][0], g['plain'][1], chunks.data_ptr(), chunks.shape[0], *hyper)
            for p in ps:
                self.state[p]['step'] += 1
            _mark_updated(ps)
            for p, e in g['packed']:
                core.stamp_packed(p, e)
        del keep
        return loss


def _mark_updated(ps):
    """The kernel updated the parameters in place behind autograd's back: advance their version counters (the bf16 weight caches of
    core.layer_weights / heads key on tensor._version, exactly as they would notice torch.optim.Adam's in-place update)."""
    try:
        torch._C._autograd._unsafe_set_version_counter(tuple(ps), tuple(p._version + 1 for p in ps))
    except (AttributeError, TypeError):                          # older torch: a no-op in-place op does the same (one extra pass over memory)
        torch._foreach_mul_(list(ps), 1.0)


def get_optimizer(params, lr=1e-4, wd=1e-2, betas=(0.9, 0.99), eps=1e-8, filter_by_requires_grad=False, group_wd_params=True, use_lion=False,
                  **kwargs):
    """optimizer.py:11-37 (same signature).  `use_lion` is accepted and ignored exactly like the reference (it never reads it)."""
    has_wd = wd > 0
    params = list(params)
    if filter_by_requires_grad:
        params = [t for t in params if t.requires_grad]
    if group_wd_params and has_wd:
        wd_params, no_wd_params = separate_weight_decayable_params(params)
        params = [{'params': wd_params}, {'params': no_wd_params, 'weight_decay': 0}]
    if not has_wd:
        return FusedAdam(params, lr=lr, betas=betas, eps=eps)                                        # torch.optim.Adam
    return FusedAdam(params, lr=lr, weight_decay=wd, betas=betas, eps=eps, decoupled_weight_decay=True)   # torch.optim.AdamW
