"""ms/step of the Coarse / Semantic wrappers with unique_consecutive ON (the reference's default, audiolm_pytorch.py:1412 / :1619: a data-dependent width) against
OFF (the benchmark's setting), same model and inputs.  usage: python scripts/uc_bench.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audiolm_pytorch_amd as A  # noqa: E402

dev = torch.device('cuda')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(wrapper, inputs, tag):
    params = [p for p in wrapper.parameters() if p.requires_grad]
    losses = []

    def step():
        for p in params:
            p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = wrapper(**inputs, return_loss=True)
        loss.backward()
        return loss

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(step())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f'{tag}: {ms:.3f} ms/step  loss {float(losses[-1]):.5f}', flush=True)


b = bench.build('coarse2048', dev, 0, None)
model, inputs = b['model'], b['inputs']
g = torch.Generator().manual_seed(5)
# semantic ids with RUNS (what unique_consecutive is for: ~40 % of the positions repeat their predecessor) next to the benchmark's uniform ids
rep = inputs['semantic_token_ids'].clone()
m = (torch.rand(rep.shape, generator=g) < 0.4).to(dev)
for _ in range(3):
    rep[:, 1:] = torch.where(m[:, 1:], rep[:, :-1], rep[:, 1:])
inputs_rep = dict(inputs, semantic_token_ids=rep)
# ids WITHOUT any run: the collapse changes nothing, N stays 2048 -- what is left against unique_consecutive=False is the cost of the data-dependent width itself
# (one launch + one host read per step)
nr = inputs['semantic_token_ids'].clone()
for i in range(1, nr.shape[1]):
    nr[:, i] = torch.where(nr[:, i] == nr[:, i - 1], (nr[:, i] + 1) % 500, nr[:, i])
inputs_norun = dict(inputs, semantic_token_ids=nr)
for r in range(2):
    for uc in (False, True):
        w = A.CoarseTransformerWrapper(transformer=model, codec=bench.Codec(), unique_consecutive=uc, mask_prob=0.15).train()
        run(w, inputs, f'round {r} coarse unique_consecutive={uc} uniform ids')
        run(w, inputs_norun, f'round {r} coarse unique_consecutive={uc} ids without runs (N = 2048 either way)')
        if uc:
            run(w, inputs_rep, f'round {r} coarse unique_consecutive={uc} ids with runs')
