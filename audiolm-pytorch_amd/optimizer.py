"""Mirror of the reference's `audiolm_pytorch/optimizer.py` (get_optimizer, optimizer.py:11-37) on a fused MI355X optimiser step.

`get_optimizer(params, lr, wd, betas, eps, filter_by_requires_grad, group_wd_params)` keeps the reference's signature and grouping rule
(weight decay only for parameters with ndim >= 2, optimizer.py:3-8 / :25-33; wd == 0 -> Adam, else AdamW) and returns a `FusedAdam`:
a torch.optim.Optimizer whose `step()` is two HIP launches over EVERY parameter (csrc/optim.hip) and whose state_dict has torch.optim.Adam's
layout ('step', 'exp_avg', 'exp_avg_sq'), so checkpoints interchange with the reference.

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
        self._chunks = None             # (key, [per group: device chunk table], partial buffer, [offsets])
        self._prepared = None           # (key of gradient pointers, [per group: (ps, host AlmOptTensor array, ...)], the arrays' owner)
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

    def _prepare(self):
        """Per iteration: one HOST table of AlmOptTensor per group (the C side hands it to the kernels in their arguments, 64 tensors per launch:
        gradient storage changes every step, and a staged host-to-device copy per step is exactly what could stall the host behind the stream)
        + the cached device chunk tables.  clip_grad_norm_() and step() of the same iteration share it."""
        groups = [self._group_tensors(g) for g in self.param_groups]
        flat = [p for ps in groups for p in ps]
        if not flat:
            return None
        gkey = tuple(p.grad.data_ptr() for p in flat)
        if self._prepared is not None and self._prepared[0] == gkey:
            return self._prepared[1]
        dev = flat[0].device
        rec = ctypes.sizeof(_lib.AlmOptTensor)
        arr = (_lib.AlmOptTensor * len(flat))()
        i = 0
        for group, ps in zip(self.param_groups, groups):
            # the usual case -- every tensor of the launch at the same step -- takes the launch's `step` argument (bias corrections computed once
            # on the host, in double); only a group with mixed counts stores them per tensor (the kernel then derives its own corrections)
            steps = [int(self.state[p]['step']) for p in ps]
            mixed = len(set(steps)) > 1
            for p, st_count in zip(ps, steps):
                st = self.state[p]
                # step: THIS parameter's count for the coming update (torch.optim.Adam tracks it per parameter: one whose gradient first
                # appears later, or is None on some steps, has its own bias corrections)
                arr[i] = _lib.AlmOptTensor(p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel(),
                                           float(group['weight_decay']), st_count + 1 if mixed else 0)
                i += 1
        ckey = tuple(tuple(p.numel() for p in ps) for ps in groups)
        if self._chunks is None or self._chunks[0] != ckey:
            ch = _lib.query('alm_opt_chunk_elems')
            tabs, offs, off = [], [], 0
            for ps in groups:
                pairs = [(i, c) for i, p in enumerate(ps) for c in range((p.numel() + ch - 1) // ch)]
                tabs.append(torch.tensor(pairs, dtype=torch.int32).reshape(-1, 2).to(dev) if pairs else None)
                offs.append(off)
                off += len(pairs)
            self._chunks = (ckey, tabs, torch.empty(max(off, 1), dtype=F32, device=dev), offs)
        out, start = [], 0
        for gi, ps in enumerate(groups):
            out.append((ps, (ctypes.addressof(arr) + start * rec, len(ps)), self._chunks[1][gi], self._chunks[3][gi]))
            start += len(ps)
        self._prepared = (gkey, out, arr)                         # `arr` owns the memory the addresses above point into
        return out

    # ------------------------------------------------------------------------------------------------------------------ API
    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """Global L2 norm of every gradient this optimiser owns -> 0-d device tensor (no host sync); the next step() clips by it
        (torch.nn.utils.clip_grad_norm_ semantics: coef = min(1, max_norm / (norm + 1e-6)))."""
        prep = self._prepare()
        if prep is None:
            return torch.zeros((), dtype=F32)
        partial = self._chunks[2]
        n = 0
        for ps, table, chunks, off in prep:
            if chunks is None:
                continue
            _lib.call('alm_opt_grad_sumsq', table[0], table[1], chunks.data_ptr(), chunks.shape[0], partial.data_ptr() + 4 * off, ops._st())
            n = off + chunks.shape[0]
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
        keep = self._prepared                                     # (holds the host table alive until the launches below have copied it)
        self._prepared = None
        if prep is None:
            return loss
        for group, (ps, table, chunks, _) in zip(self.param_groups, prep):
            if chunks is None:
                continue
            for p in ps:
                self.state[p]['step'] += 1
            step = int(self.state[ps[0]]['step'])                # used by the tensors whose table entry carries no step of its own
            b1, b2 = group['betas']
            _lib.call('alm_opt_adam_step', table[0], table[1], chunks.data_ptr(), chunks.shape[0], float(group['lr']), float(b1), float(b2),
                      float(group['eps']), step, int(bool(group['decoupled_weight_decay'])), clip[1].data_ptr() if clip else None,
                      clip[0] if clip else 0.0, ops._st())
            _mark_updated(ps)
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
