"""Training step as ONE hipGraph launch, optionally as two half-batches on two HIP streams.

Why: one forward + backward of the fused stack is ~600 kernel launches issued from Python / ctypes (~9.6 ms of host time per step at
dim 1024 depth 6, measured: DESIGN.md section 4).  As long as the GPU needs longer than that per step the host hides behind it, but (a) every
GPU-side gain moves the step towards being host-bound and (b) the schedule that overlaps the HBM-bound kernels (hyper-connections,
LayerNorm, GEGLU: the matrix cores idle) with the MFMA-bound GEMMs (HBM idle) needs twice as many launches: the batch is split into two
micro-batches that run the whole forward + backward independently on two streams, so that one half's GEMM is co-resident with the other
half's row kernels.  Captured once (torch.cuda.CUDAGraph: every alm_* launch goes to torch's current stream, which is the capturing stream)
and replayed with one launch per step, the host cost disappears.

The split lives INSIDE the fused stack (core.TransformerStackFn, opts micro = 2): autograd sees one node on one stream, the stack forks /
joins its second stream with events, embeddings / logit heads / loss run on the full batch.  Every kernel of the stack is per-sequence, so
the arithmetic is unchanged; the weight gradients of the two halves are summed.

    step = GraphedTrainStep(wrapper, dict(semantic_token_ids=sem, coarse_token_ids=coarse), micro_batches=2)
    loss = step(semantic_token_ids=sem, coarse_token_ids=coarse)      # p.grad of every parameter holds this step's gradient
    optimizer.step()

Dropout inside a captured step: the feed-forward / to_out masks are drawn by torch's graph-safe generator (re-drawn per replay); the in-kernel
attention dropout of the flash kernels reads its mask-stream counter from the device (core.graph_seed_state), advanced by the captured step, so every
replay draws new masks as well (tests/test_gpu_graphed.py::test_graph_replays_redraw_the_attention_dropout_masks).

Restrictions (checked or documented): fixed shapes; no data-dependent host control flow inside the step (`unique_consecutive=True` is one:
its output length depends on the data); the gradient exchange of parallel.DataParallelEngine is not captured -- call the engine's
`reduce_grads()` style hooks outside, or use the eager path for multi-GPU runs.
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, module, example_inputs: dict, *, micro_batches: int = 1, warmup: int = 3, call_kwargs=None, params=None):
        assert micro_batches in (1, 2), 'one batch or two half-batches'
        self.module = module
        self.kw = dict(call_kwargs or dict(return_loss=True))
        self.params = [p for p in (params if params is not None else module.parameters()) if p.requires_grad]
        self.nmb = micro_batches
        self.static_in = {k: v.clone() for k, v in example_inputs.items()}
        b = next(iter(self.static_in.values())).shape[0]
        assert all(v.shape[0] == b for v in self.static_in.values()), 'inputs are batch-first tensors of one batch size'
        assert b % micro_batches == 0, (b, micro_batches)
        self.dev = next(iter(self.static_in.values())).device
        self._caches = [m._cache for m in module.modules() if hasattr(m, '_cache') and hasattr(m._cache, 'store')]
        self.graph = None
        self.loss = None
        self.grads = None
        self._stacks = [m for m in module.modules() if hasattr(m, 'micro_batches') and hasattr(m, 'flat_params')]
        if any(getattr(m, 'attn_dropout', 0.) > 0. for m in self._stacks):
            # attention dropout inside a captured step: the mask-stream counter lives on the device and is advanced by the captured step itself
            # (core.graph_seed_state) -- a host seed would be baked into the graph and every replay would drop the same pairs.  It has to exist
            # before the capture begins.
            from . import core
            core.graph_seed_state(self.dev, create=True)
        self._capture(warmup)

    # the work of one step, issued on the current stream (with micro_batches == 2 the fused stack forks its second stream itself)
    def _issue(self):
        for c in self._caches:
            c.store.clear()                                  # the bf16 weight copies are re-packed inside the step: weights change between replays
        for m in self._stacks:
            m.micro_batches = self.nmb
        try:
            loss = self.module(**self.static_in, **self.kw)
            grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        finally:
            for m in self._stacks:
                m.micro_batches = 1
        return loss.detach(), list(grads)

    def _capture(self, warmup):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):                   # lazy initialisation (occupancy queries, allocator pools, side streams) outside the capture
                self._issue()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss, self.grads = self._issue()
        self.graph = g
        # The capture only RECORDED the step: the bf16 weight packs it re-created were keyed into the WeightCaches with the masters' current
        # (data_ptr, version), but their contents -- like self.loss and self.grads -- do not exist until the graph has run once.  An eager forward
        # (eval, generate, a parity check) between construction and the first replay would hit those cache entries and read uninitialised packs.
        # One replay makes every captured buffer real.
        g.replay()
        torch.cuda.synchronize(self.dev)

    def __call__(self, **inputs):
        for k, v in inputs.items():
            self.static_in[k].copy_(v, non_blocking=True)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return self.loss
