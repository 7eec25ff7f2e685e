"""Measures the REAL reference's own bf16-autocast noise: for every model-level golden fixture, runs the reference wrapper under
torch.autocast('cpu', dtype=bfloat16) (what trainer.py:1241 `accelerator.autocast()` does) and records how far its loss / logits /
gradients move from its own fp32 run.  The GPU parity tests bound the HIP path's deviation by max(fixed tolerance, 2 x this noise).
Build-container only.  Output: tests/golden/bf16_noise.pt
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle'))
import make_golden as MG  # noqa: E402  (loads the reference under shims)
from common import synth_state_dict  # noqa: E402

A = MG.A


def run(fx, autocast):
    kind, ctor, opt, inp = fx['kind'], fx['ctor'], fx['options'], fx['inputs']
    K = dict(semantic=A.SemanticTransformer, coarse=A.CoarseTransformer, fine=A.FineTransformer)[kind]
    model = K(**ctor)
    model.load_state_dict(synth_state_dict(fx['shapes'], fx['seed']))
    if kind == 'semantic':
        w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=opt['unique_consecutive'], mask_prob=opt['mask_prob'])
        kw = dict(semantic_token_ids=inp['ids'])
    elif kind == 'coarse':
        w = A.CoarseTransformerWrapper(transformer=model, codec=MG._Codec(), unique_consecutive=opt['unique_consecutive'], mask_prob=opt['mask_prob'])
        kw = dict(semantic_token_ids=inp['semantic_token_ids'], coarse_token_ids=inp['coarse_token_ids'])
    else:
        nq = ctor['num_coarse_quantizers'] + ctor['num_fine_quantizers']
        w = A.FineTransformerWrapper(transformer=model, codec=MG._Codec(nq), mask_prob=opt['mask_prob'])
        kw = dict(coarse_token_ids=inp['coarse_token_ids'], fine_token_ids=inp['fine_token_ids'])
    if inp.get('text_embeds') is not None:
        kw['text_embeds'] = inp['text_embeds']
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
        loss, logits, grads = MG._run(model, w, kw, opt.get('training', True), inp.get('forgetful_mask'), inp.get('cond_keep'))
    lg = logits if isinstance(logits, (tuple, list)) else (logits,)
    return float(loss), {k: (g.float().clone() if g is not None else None) for k, g in grads.items()}, [t.detach().float().clone() for t in lg if t is not None]


def main():
    out = {}
    for name in ('semantic_s4_flash', 'coarse_s1_flash_uc_mask', 'coarse_s4_flash_mask', 'fine_s4_flash',
                 'coarse_s4_bias', 'coarse_s1_bias_eval', 'fine_s1_bias_mask',
                 'coarse_s4_cond_cross', 'coarse_s1_cond_cross_drop_bias', 'semantic_s4_cond_prefix_bias', 'fine_s4_cond_prefix_flash_drop', 'fine_s1_cond_cross'):
        fx = torch.load(os.path.join(HERE, name + '.pt'), weights_only=False)
        l32, g32, lg32 = run(fx, False)
        l16, g16, lg16 = run(fx, True)
        # the reference's own logits move by this much (relative Frobenius norm) when it runs under bf16 autocast
        lgn = [float((a - b).norm() / b.norm()) for a, b in zip(lg16, lg32)]
        gn = {}
        for k in g32:
            if g32[k] is None or float(g32[k].norm()) < 1e-9:
                continue
            gn[k] = float((g16[k] - g32[k]).norm() / g32[k].norm())
        out[name] = dict(loss_fp32=l32, loss_bf16=l16, loss_abs=abs(l16 - l32), grads=gn, logits=lgn)
        worst = sorted(gn.items(), key=lambda kv: -kv[1])[:4]
        print(f'{name}: loss fp32={l32:.6f} bf16={l16:.6f} |d|={abs(l16 - l32):.2e} rel={abs(l16 - l32) / abs(l32):.2e}; logits rel-frob', ['%.2e' % v for v in lgn],
              '; worst grad noise', [(k, round(v, 3)) for k, v in worst])
    torch.save(out, os.path.join(HERE, 'bf16_noise.pt'))


if __name__ == '__main__':
    main()
