"""Parity AT THE BENCHMARK SIZES on a real MI355X: the HIP path vs the fp32 CPU oracle for BASELINE.json configs[1] (CoarseTransformer
dim=1024 depth=6, N=1024), the headline (same model, N=2048) and configs[2] (FineTransformer dim=1024 depth=6, 3 + 5 quantizers, N=2049),
each with 4 residual streams (restated hyper-connections) and with 1 stream (every op first-party reference code).

WHICH kernels run depends on the batch (tests/test_plan_queries.py pins the selection on the CPU):
  * B = 1 / 2 (M = 2048 rows; the round 2-3 cases): attention across many 64-key tiles, the 2730 -> 2736 padded FFN width, split-K weight gradients
    over K = B*N tokens, the multi-token-per-workgroup hyper-connection loops -- but every NT GEMM of the step is on the 128 x 128 tile (too few 256 x 256
    tiles for 256 CUs) and the weight gradients on the uniform split-K plan;
  * B = 8 (M = 16 384 rows = EXACTLY what bench.py times; round 4): `gemm_kernel<384,256>` (W1 forward, dHN), `gemm_stag_kernel<NT>` (to_out, W2, the dXN
    dgrads, the coarse logit head), the layer-batched weight gradients on the hybrid plan (`gemm_w4_kernel<TN>` at full K = 16 384 + split-K tail), the
    stacked `out=` buffers of the deferred mode.  The oracle runs the 8 sequences in one CPU pass (~8x the B = 1 time).
Parameters are the
non-degenerate synthetic values of tests/golden/common.py (every hyper-connection / LayerNorm / bias path carries signal); inputs are
seeded uniform token ids; the forgetful mask is drawn once on the CPU and injected on both sides.

Compared END TO END: the loss, EVERY logit, and the gradient of EVERY parameter, against two oracles:
  (B) the fp32 oracle (pinned to the real reference at these sizes: tests/test_oracle_golden.py::test_oracle_matches_reference_at_benchmark_size):
        loss     |d| <= 1e-3 * |loss|                      (north_star: loss within 1e-3; no noise clause)
        logits   rel Frobenius error <= max(1e-2, the oracle's own bf16-autocast deviation on the same inputs)  -- both numbers are reported
        grads    per tensor rel Frobenius error <= max(3e-2, 2 x the oracle's bf16-autocast deviation of that tensor); hyper-connection scalar
                 statistics pooled (see tests/test_gpu_parity.py).  HONEST SCOPE: for the hyper-connection SCALARS (static_alpha / static_beta /
                 dynamic_*_scale: sums over all tokens of terms that cancel) this end-to-end bound does not discriminate -- e.g.
                 layers.0.0.dynamic_beta_scale passes at a rel error of 5.8 because the reference's own bf16-autocast run moves it by 6.1
                 (profiles/r3_runZ_fullsize_parity.jsonl); those gradients are checked with teeth only op by op (tests/test_gpu_opwise.py: 1e-2
                 given the same inputs, measured <= 1e-4 with fp32 streams)
  (A) the ROUNDING-MATCHED oracle (oracle/rounding_matched.py: bf16 rounding at the HIP path's storage / operand points).  MEASURED (round 3,
      profiles/r3_runC_fullsize_parity.jsonl, profiles/r3_runA_golden_parity.log): end to end it is NO closer than (B) -- logits 0.7-3e-2 -- although the very same oracle reproduces the
      small goldens to 4e-7 when no rounding flips and every single op of THIS stack to <= 7e-5 given the same inputs (tests/test_gpu_opwise.py).  A
      6-layer dim-1024 stack amplifies the ~1e-3 of bf16 roundings that a different fp32 summation order flips until the two runs' rounding errors are
      independent draws: an end-to-end 1e-3 does not exist for this model at this size, for any implementation.  (A) is therefore reported and held
      to the same bound as (B); the bound WITH TEETH at these sizes is the op-by-op one in tests/test_gpu_opwise.py (303 comparisons per case:
      forward <= 1e-3, measured <= 7e-5; backward <= 3e-3; cancelling hyper-connection scalar gradients <= 1e-2).
Synthetic hyper-connection weights are width-scaled (tests/golden/common.py): the dynamic pre-activations keep a std of ~0.4 at dim 1024.
test_full_size_matches_real_reference_digest compares the HIP path DIRECTLY with digests of the REAL reference at these sizes (tests/golden/full_*.pt).
Every run appends its numbers to gpurun_out/r6_fullsize_parity.jsonl (copied to profiles/ for the record).
Round 6: the TIMED batch shapes of the other BASELINE configurations -- configs[1] at B = 8 x N = 1024 (M = 8192: every NT output of width 1024 / 512 is
an under-filled launch there), configs[2] at B = 8 x N = 2049 (M = 16 392: a ragged last tile row in every GEMM and hyper-connection launch), and
configs[4]'s MODEL (codebook 4096: C = 4097-column coarse heads over 3 x 4097-row tables) at its own N = 8253, B = 1.
"""
import json
import os
import time

import pytest
import torch

import audiolm_oracle as O
from common import synth_state_dict
from test_gpu_parity import HC_SCALARS, _frob, grad_report, ours_run
from test_oracle_golden import FULL_FIXTURES, _load, digest_errors, oracle_run
import rounding_matched as RM

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, 'gpurun_out', 'r6_fullsize_parity.jsonl')


def _case(kind, streams, N_kind, batch=None):
    g = torch.Generator().manual_seed(1234)
    extra = {} if streams == 4 else dict(num_residual_streams=streams)
    if kind == 'coarse':
        cb = 4096 if N_kind == 8253 else 1024                                  # N = 8253: BASELINE configs[4]'s model
        ctor = dict(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=cb, num_coarse_quantizers=3, flash_attn=True, **extra)
        ns, nf = {1024: (253, 256), 2048: (509, 512), 8253: (1500, 2250)}[N_kind]
        B = batch or (2 if N_kind == 1024 else 1)
        inputs = dict(semantic_token_ids=torch.randint(0, 500, (B, ns), generator=g), coarse_token_ids=torch.randint(0, cb, (B, nf, 3), generator=g))
        N = 1 + (ns + 1) + 1 + nf * 3
        inputs['forgetful_mask'] = O.generate_mask_with_prob((B, N), 0.15, 'cpu', generator=g)
        options = dict(training=True, unique_consecutive=False, mask_prob=0.15)
    else:
        ctor = dict(dim=1024, depth=6, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024, flash_attn=True, **extra)
        B = batch or 1
        grid = torch.randint(0, 1024, (B, 256, 8), generator=g)
        inputs = dict(coarse_token_ids=grid[..., :3].contiguous(), fine_token_ids=grid[..., 3:].contiguous())
        N = 1 + 768 + 1 + 1279
        inputs['forgetful_mask'] = O.generate_mask_with_prob((B, N), 0.15, 'cpu', generator=g)
        options = dict(training=True, mask_prob=0.15)
    return ctor, inputs, options, N, B


@pytest.mark.parametrize('kind,streams,N_kind,residual,batch',
                         [('coarse', 4, 2048, 'fp32', None), ('coarse', 4, 2048, 'bf16', None), ('coarse', 1, 2048, 'fp32', None),
                          ('coarse', 4, 1024, 'fp32', None), ('coarse', 4, 1024, 'bf16', None), ('coarse', 1, 1024, 'fp32', None),
                          ('fine', 4, 2049, 'fp32', None), ('fine', 4, 2049, 'bf16', None), ('fine', 1, 2049, 'fp32', None),
                          ('coarse-default-init', 4, 2048, 'bf16', None),
                          # round 4: the BENCHMARKED shape itself -- B = 8 x N = 2048, bf16 streams, synthetic and default (= bench.py's) initialisation
                          ('coarse', 4, 2048, 'bf16', 8), ('coarse-default-init', 4, 2048, 'bf16', 8),
                          # round 6: the other configurations AT THEIR TIMED SHAPES -- configs[1] B = 8 x N = 1024, configs[2] B = 8 x N = 2049 (ragged
                          # M = 16 392), configs[4]'s codebook-4096 model at N = 8253 (reference audiolm_pytorch.py:896-906, 965-983)
                          ('coarse', 4, 1024, 'bf16', 8), ('fine', 4, 2049, 'bf16', 8), ('coarse', 4, 8253, 'bf16', 1)])
def test_full_size_matches_oracle(kind, streams, N_kind, residual, batch):
    """residual: HBM storage of the 4 residual streams (bf16 = the benchmark's setting = what autocast gives the reference).
    batch = 8: M = 16 384 rows, the shape bench.py times -- the big-tile NT GEMMs, the hybrid-plan batched weight gradients and the stacked buffers of the
    deferred mode run inside this comparison (asserted below through the library's plan queries).  The rounding-matched oracle pass is skipped there
    (it would triple the CPU time and is reported-only: see (A) in the module docstring); every gradient tensor listed is held to max(3e-2, 2 x the
    oracle's own bf16-autocast deviation), the hyper-connection SCALAR gradients pooled -- their per-tensor errors are dominated by cancellation noise
    (a scalar passes at rel error > 1 when the reference's own bf16 run moves it as much): the discriminating bound for them is the op-wise one (1e-2).
    'coarse-default-init': the reference's DEFAULT initialisation instead of the synthetic values (hyper-connection dynamic weights zero,
    randn logit weights: the weights bench.py times) -- the synthetic hyper-connection weights (0.05 randn over 1024 features: saturating tanh
    gates) amplify every rounding difference, the reference's own bf16 run moves its logits by 7-19 % there."""
    import audiolm_pytorch_amd as A
    default_init = kind.endswith('-default-init')
    kind = kind.split('-')[0]
    ctor, inputs, options, N, B = _case(kind, streams, N_kind, batch)
    assert N == N_kind
    if batch == 8:                                                             # the kernels the roofline is quoted on really run in this case
        import ctypes
        from audiolm_pytorch_amd import _lib
        plan = (ctypes.c_int * 4)()
        if N_kind == 1024:
            # configs[1]'s timed shape, M = 8192: the two wide FFN GEMMs are on the big tiles; every D- / 512-wide output is an under-filled launch (128 / 64
            # tiles of 256 x 256) -- the long contraction (dXN = dU W1, K = 5472) takes the in-launch split-K form of the staggered tile (round 6: the one shape
            # where it measured faster), W2 / to_out stay on the 128 x 128 tile, to_q / to_kv on its 4-stage DMA-ring form
            assert _lib.query('alm_gemm_nt_tile_choice', B * N, 2 * 2736, 1) in (11, 13)
            q = lambda n_, k_: (_lib.query('alm_gemm_nt_plan', B * N, n_, k_, 1, 1, ctypes.cast(plan, ctypes.c_void_p)), plan[0], plan[1])[1:]
            assert q(1024, 2 * 2736) == (13, 2) and q(1024, 2736)[1] == 1 and q(1024, 512) == (1, 1) and q(512, 1024) == (16, 1) and q(128, 1024) == (16, 1)
        else:
            assert _lib.query('alm_gemm_nt_tile_choice', B * N, 2 * 2736, 1) == 11
            assert _lib.query('alm_gemm_nt_tile_choice', B * N, 1024, 1) == (13 if B * N == 16384 else 17)      # M = 16 392 (fine, ragged): 52 x 4 tiles of 320 x 256 (round 6)
            assert _lib.query('alm_gemm_tn_batched_plan', 2730, 1024, B * N, 12, ctypes.cast(plan, ctypes.c_void_p)) == 2
    K = dict(coarse=A.CoarseTransformer, fine=A.FineTransformer)[kind]
    torch.manual_seed(7)
    m0 = K(**ctor)
    shapes = {k: tuple(v.shape) for k, v in m0.state_dict().items()}
    seed = 4242 + streams
    fx = dict(kind=kind, ctor=ctor, shapes=shapes, seed=seed, options=options, inputs=inputs)
    state = {k: v.detach().clone() for k, v in m0.state_dict().items()} if default_init else synth_state_dict(shapes, seed)
    del m0
    import test_oracle_golden as TG
    orig_synth = TG.synth_state_dict
    TG.synth_state_dict = lambda shapes_, seed_: {k: v.clone() for k, v in state.items()}      # oracle_run() re-synthesises from (shapes, seed)

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    oloss, ologits, ograds = oracle_run(fx)                                   # fp32 CPU oracle, same synthetic parameters (shapes, seed)
    t_oracle = time.time() - t0
    ologits = [t.detach() for t in ologits if t is not None]
    with torch.autocast('cpu', dtype=torch.bfloat16):                          # the oracle's own bf16-autocast deviation on these inputs
        nloss, nlogits, ngrads = oracle_run(fx)
    t0 = time.time()
    rloss = rlogits = rgrads = None
    if batch is None and N_kind != 8253:                                       # (one oracle pass each for the round-6 cases)
        with RM.rounding_matched(residual_bf16=(residual == 'bf16' and streams > 1)):    # (A) the rounding-matched oracle, same storage type of the streams
            rloss, rlogits, rgrads = oracle_run(fx)
        rlogits = [t.detach() for t in rlogits if t is not None]
    t_rm = time.time() - t0
    TG.synth_state_dict = orig_synth
    nlogits = [t.detach().float() for t in nlogits if t is not None]
    noise = dict(loss_rel=abs(float(nloss) - float(oloss)) / abs(float(oloss)),
                 logits=[_frob(a, b) for a, b in zip(nlogits, ologits)],
                 grads={k: _frob(ngrads[k].float(), g) for k, g in ograds.items() if g is not None and float(g.norm()) >= 1e-7})
    del nlogits, ngrads

    loss, logits, grads = ours_run(fx, want_logits=True, state=state, residual_dtype=torch.bfloat16 if residual == 'bf16' else torch.float32)
    logits = [t for t in (logits if isinstance(logits, (tuple, list)) else (logits,)) if t is not None]

    rel = abs(loss - float(oloss)) / abs(float(oloss))
    rep = [f'{kind}{" (default init)" if default_init else ""} S={streams} N={N} B={B} residual streams {residual}: loss ours={loss:.6f} oracle={float(oloss):.6f} rel |d|={rel:.2e} (bound 1e-3; oracle bf16-autocast {noise["loss_rel"]:.2e}); '
           f'oracle fwd+bwd {t_oracle:.1f} s']
    ok = rel <= 1e-3
    lerr = []
    assert len(logits) == len(ologits)
    for got, want, nz in zip(logits, ologits, noise['logits']):
        assert got.shape == want.shape, (got.shape, want.shape)
        e = _frob(got, want)
        lerr.append(e)
        rep.append(f'  logits {tuple(want.shape)}: rel-frob {e:.2e} (bound max(1e-2, oracle bf16-autocast {nz:.2e}))')
        ok &= e <= max(1e-2, nz)
    items = []
    for k, g in ograds.items():
        if g is None or float(g.norm()) < 1e-7:
            continue
        assert grads.get(k) is not None, f'missing gradient for {k}'
        items.append((k, _frob(grads[k], g), float(g.norm()), noise['grads'].get(k, 0.0)))
    ok &= grad_report(items, rep)
    # (A) against the rounding-matched oracle, end to end: reported, held to the same bound as (B) (see the module docstring: a deep stack decorrelates
    # the rounding errors of ANY two runs; the tight per-op bounds are in tests/test_gpu_opwise.py)
    rm_l, worst_t, worst_s, rm_loss = None, (None, '-'), (None, '-'), None
    if rloss is not None:
        rm_l = [_frob(a, b) for a, b in zip(logits, rlogits)]
        rm_loss = abs(loss - float(rloss)) / abs(float(rloss))
        rep.append(f'  vs ROUNDING-MATCHED oracle ({t_rm:.1f} s): loss rel |d| {rm_loss:.2e}; logits rel-frob {["%.2e" % e for e in rm_l]}')
        ok &= all(e <= max(1e-2, nz) for e, nz in zip(rm_l, noise['logits']))
        rm_g = {k: _frob(grads[k], g) for k, g in rgrads.items() if g is not None and float(g.norm()) >= 1e-7}
        worst_t = max((e, k) for k, e in rm_g.items() if not k.endswith(HC_SCALARS))
        hc = [(e, k) for k, e in rm_g.items() if k.endswith(HC_SCALARS)]
        worst_s = max(hc) if hc else (0.0, '-')
        rep.append(f'     grads: worst tensor {worst_t[0]:.2e} ({worst_t[1]}), worst hyper-connection scalar {worst_s[0]:.2e} ({worst_s[1]})')
    print('\n'.join(rep))
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, 'a') as fh:
        fh.write(json.dumps(dict(kind=kind, init='default' if default_init else 'synthetic', streams=streams, N=N, B=B, residual_streams=residual, loss_ours=loss, loss_oracle=float(oloss),
                                 loss_rel=rel, loss_rel_oracle_bf16=noise['loss_rel'], logits_rel_frob=lerr, logits_rel_frob_oracle_bf16=noise['logits'],
                                 worst_grad_tensor_rel_frob_vs_rounding_matched=worst_t[0], worst_grad_hc_scalar_rel_frob_vs_rounding_matched=worst_s[0],
                                 logits_rel_frob_vs_rounding_matched=rm_l, loss_rel_vs_rounding_matched=rm_loss,
                                 worst_grad_rel_frob_vs_fp32_oracle=max(e for _, e, _, _ in items), n_grad_tensors=len(items),
                                 grads_over_3e2=sorted([(k, round(e, 4), round(nz, 4)) for k, e, _, nz in items if e > 3e-2], key=lambda t: -t[1])[:12],
                                 oracle_seconds=round(t_oracle, 1), ok=bool(ok))) + '\n')
    assert ok, '\n'.join(rep)


@pytest.mark.parametrize('name,residual', [(n, 'fp32') for n in FULL_FIXTURES] + [('full_coarse_s4', 'bf16'), ('full_fine_s4', 'bf16')])
def test_full_size_matches_real_reference_digest(name, residual):
    """The HIP path against the REAL reference at the benchmark sizes -- no oracle in between (round 3; fixtures: tests/golden/make_golden.py
    fullsize, run in the build container on reference audiolm_pytorch.py:858-990 / :1136-1368 / :1742-1854 / :2041-2137).  The digests hold the loss, a
    strided sample of every logits tensor, norm + strided sample of every parameter gradient and the reference's own bf16-autocast deviation.
    Bounds: loss 1e-3; logits sample rel-Frobenius <= max(1e-2, the reference's bf16 deviation); gradients (sample and norm) per tensor
    <= max(3e-2, 2 x the reference's bf16 deviation), hyper-connection scalars pooled."""
    fx = _load(name)
    loss, logits, grads = ours_run(fx, want_logits=True, residual_dtype=torch.bfloat16 if residual == 'bf16' else torch.float32)
    logits = [t for t in (logits if isinstance(logits, (tuple, list)) else (logits,)) if t is not None]
    lrel, lerr, gerr = digest_errors(fx, loss, logits, grads)
    nz = fx['noise']
    rep = [f'{name} [residual streams {residual}] vs the REAL reference: loss rel |d| {lrel:.2e} (bound 1e-3; reference bf16-autocast {nz["loss_abs"] / abs(float(fx["outputs"]["loss"])):.2e})']
    ok = lrel <= 1e-3
    for e, n_ in zip(lerr, nz['logits']):
        rep.append(f'  logits sample rel-frob {e:.2e} (bound max(1e-2, reference bf16-autocast {n_:.2e}))')
        ok &= e <= max(1e-2, n_)
    items = [(k, max(en, es), fx['outputs']['grads'][k]['norm'], nz['grads'].get(k, 0.0)) for k, (en, es) in gerr.items()]
    ok &= grad_report(items, rep)
    print('\n'.join(rep))
    with open(REPORT, 'a') as fh:
        fh.write(json.dumps(dict(kind='real-reference digest ' + name, residual_streams=residual, loss_rel=lrel, logits_sample_rel_frob=lerr,
                                 logits_rel_frob_reference_bf16=nz['logits'], worst_grad_sample_rel_frob=max(e for _, e, _, _ in items), ok=bool(ok))) + '\n')
    assert ok, '\n'.join(rep)
