"""Repro: the captured training step with the deferred weight gradients forked in more than one group (`python scripts/debug/graph_defer_groups.py 2`)
replays with NaN / wrong gradients; with one group at the end of the backward pass (argument 1, what the product always does inside a capture) every
gradient matches the eager step.  See core.stack_backward.  The group count is set HERE (core._DEBUG_DEFER_GROUPS_CAPTURE): the product has no
environment switch for the known-corrupting configuration."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import audiolm_pytorch_amd
from audiolm_pytorch_amd import core
from audiolm_pytorch_amd.graphed import GraphedTrainStep
from test_gpu_graphed import _setup, _eager
core._DEBUG_DEFER_GROUPS_CAPTURE = max(1, int(sys.argv[1])) if len(sys.argv) > 1 else 2
print('env', {k: v for k, v in os.environ.items() if k.startswith('ALM_')}, 'groups in capture', core._DEBUG_DEFER_GROUPS_CAPTURE, 'side streams', core.SIDE_STREAMS, core.ASYNC_WGRAD)
model, w, inputs = _setup(torch.bfloat16)
l1, g1 = _eager(model, w, inputs)
for p in model.parameters():
    p.grad = None
step = GraphedTrainStep(w, inputs, micro_batches=1)
names = [k for k, p in model.named_parameters() if p.requires_grad]
ok = [k for k, gr in zip(names, step.grads) if gr is not None and bool(torch.isfinite(gr).all())]
bad = [k for k, gr in zip(names, step.grads) if gr is not None and not bool(torch.isfinite(gr).all())]
print('loss', float(step.loss), l1, 'finite', len(ok), 'nan', len(bad))
print('finite:', ok)
def frob(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))
print('nan:', bad)
print('wrong (finite, > 2e-3 from eager):', [(k, round(frob(gr, g1[k]), 4)) for k, gr in zip(names, step.grads) if gr is not None and k in ok and frob(gr, g1[k]) > 2e-3])
import sys; sys.exit(0)
for k, gr in zip(names, step.grads):
    if gr is not None and k in bad and gr.dim() == 2:
        fin = torch.isfinite(gr)
        print(k, tuple(gr.shape), 'finite frac', float(fin.float().mean()))
