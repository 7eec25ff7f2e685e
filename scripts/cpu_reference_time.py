"""BUILD-CONTAINER ONLY (needs /root/reference): times the REAL reference (imported under oracle/ref_shims.py) and the CPU oracle (oracle/audiolm_oracle.py, the
thing bench.py's `cpu_baseline` leg times on the GPU box as kind "port") BACK TO BACK on the same cores, same architecture, same inputs -- so that the
reported port figure is backed by a measured reference / port ratio.

    python scripts/cpu_reference_time.py [--reps 3] > profiles/r5_cpu_reference_vs_oracle.json

Workload = bench.py's cpu_baseline sample: CoarseTransformer dim=1024 depth=6 heads=8 (MQA), 3 coarse quantizers, codebook 1024, B=1, 509 semantic +
512 x 3 coarse ids -> N = 2048, forgetful mask 0.15 injected (same draw on both sides), training mode, fp32, forward + backward
(reference: CoarseTransformerWrapper.forward, audiolm_pytorch.py:1742-1854).  Both the shipped 4-stream model (hyper-connections restated on BOTH sides:
the third-party package is absent) and num_residual_streams=1 (every op first-party reference code)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402

import ref_shims  # noqa: E402

A, _, _ = ref_shims.load_reference()
import audiolm_oracle as O  # noqa: E402


class _Codec:
    rq_groups = 1
    num_quantizers = 8


def run(streams, reps):
    torch.manual_seed(0)
    extra = {} if streams == 4 else dict(num_residual_streams=streams)
    model = A.CoarseTransformer(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True, **extra)
    wrapper = A.CoarseTransformerWrapper(transformer=model, codec=_Codec(), unique_consecutive=False, mask_prob=0.15)
    wrapper.train()
    g = torch.Generator().manual_seed(0)
    sem, coarse = torch.randint(0, 500, (1, 509), generator=g), torch.randint(0, 1024, (1, 512, 3), generator=g)
    N = 1 + 510 + 1 + 1536
    mask = O.generate_mask_with_prob((1, N), 0.15, 'cpu', generator=g)
    orig = A.generate_mask_with_prob
    A.generate_mask_with_prob = lambda shape, prob, device: mask.clone()
    t_ref, loss_ref = [], None
    try:
        for _ in range(reps):
            model.zero_grad(set_to_none=True)
            t0 = time.time()
            loss = wrapper(semantic_token_ids=sem, coarse_token_ids=coarse, return_loss=True)
            loss.backward()
            t_ref.append(time.time() - t0)
            loss_ref = float(loss)
    finally:
        A.generate_mask_with_prob = orig
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith('.beta')}
    full = dict(sd)
    full.update(params)
    cfg = O.Cfg(dim=1024, depth=6, streams=streams, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3)
    t_or, loss_or = [], None
    for _ in range(reps):
        for p in params.values():
            p.grad = None
        t0 = time.time()
        loss = O.coarse_wrapper_loss(full, cfg, sem, coarse, training=True, unique_consecutive=False, forgetful_mask=mask)
        loss.backward()
        t_or.append(time.time() - t0)
        loss_or = float(loss)
    best = lambda t: min(t[1:]) if len(t) > 1 else t[0]     # noqa: E731
    return dict(streams=streams, N=N, reference=dict(seconds=[round(x, 3) for x in t_ref], tokens_per_s=round(N / best(t_ref), 1), loss=loss_ref),
                oracle=dict(seconds=[round(x, 3) for x in t_or], tokens_per_s=round(N / best(t_or), 1), loss=loss_or),
                oracle_over_reference=round(best(t_ref) / best(t_or), 3), loss_rel=abs(loss_or - loss_ref) / abs(loss_ref))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    out = dict(what='real reference (shimmed, /root/reference) vs CPU oracle, fwd+bwd fp32, B=1 x N=2048, CoarseTransformer d=1024 depth=6, best of reps-1 after 1 warm-up',
               host=dict(cores=args.threads, machine='build container (no GPU): absolute tokens/s are NOT the GPU box\'s; the RATIO is what transfers'),
               torch=torch.__version__, runs=[run(4, args.reps), run(1, args.reps)])
    print(json.dumps(out, indent=1))
