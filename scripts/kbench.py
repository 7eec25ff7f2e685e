#!/usr/bin/env python
"""Kernel micro-benchmarks on the MI355X for the hot-path shapes (B=8 x N=2048 tokens, D=1024): every GEMM variant vs the vendor
BLAS yardstick (torch.matmul; NOT part of the product), attention fwd/bwd, hyper-connection kernels.  Prints one line per case:
name, ms, TFLOP/s or GB/s.   usage: python scripts/kbench.py [gemm] [attn] [hc] [misc]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd as A  # noqa: E402,F401
from audiolm_pytorch_amd import ops, _lib  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*shape, dtype=BF16):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(dtype)


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def bench_gemm():
    T = 16384
    shapes = [('W1 fwd', T, 5472, 1024), ('W2 fwd', T, 1024, 2736), ('dHN dgrad', T, 2736, 1024), ('dXN2 dgrad', T, 1024, 5472),
              ('Wq fwd', T, 512, 1024), ('Wo fwd', T, 1024, 512), ('dAO dgrad', T, 512, 1024), ('dXN dgrad', T, 1024, 512), ('Wkv fwd', T, 128, 1024),
              ('dXkv dgrad', T, 1024, 128), ('square 4096', 4096, 4096, 4096)]
    print('--- NT GEMMs  C[M,N] = A[M,K] B[N,K]^T (bf16 out)')
    for name, M, N, K in shapes:
        Am, Bm = rnd(M, K), rnd(N, K)
        C = torch.empty((M, N), dtype=BF16, device=dev)
        fl = 2.0 * M * N * K
        ref = None
        line = f'{name:14s} {M}x{N}x{K}:'
        t = timeit(lambda: torch.matmul(Am, Bm.t(), out=C))
        ref = C.clone()
        line += f'  blas {t:.3f} ms {fl / t / 1e9:7.0f} TF |'
        for tile in (1, 2, 13, 11):
            if tile >= 2 and (M < 256 or N < 256):
                continue
            C.zero_()
            t = timeit(lambda: ops.gemm_nt_tile(Am, Bm, C, tile))
            line += f'  tile{tile} {t:.3f} ms {fl / t / 1e9:7.0f} TF (err {relerr(C, ref):.1e}) |'
        print(line, flush=True)
    print('--- TN split-K wgrads  C[M,N] = At[K,M]^T Bt[K,N] (fp32 out)')
    for name, M, N, K in [('dW1 half', 2730, 1024, T), ('dW2', 1024, 2730, T), ('dWq', 512, 1024, T), ('dWo', 1024, 512, T), ('dWkv', 128, 1024, T),
                          ('dWhead', 1025, 1024, 4096)]:
        At, Bt = rnd(K, (M + 7) // 8 * 8)[:, :M], rnd(K, (N + 7) // 8 * 8)[:, :N]
        C = torch.empty((M, N), dtype=F32, device=dev)
        fl = 2.0 * M * N * K
        t0 = timeit(lambda: torch.matmul(At.t(), Bt))
        ref = torch.matmul(At.t().float(), Bt.float())
        t = timeit(lambda: ops.gemm_tn_splitk(At, Bt, C))
        print(f'{name:14s} {M}x{N}x{K}:  blas {t0:.3f} ms {fl / t0 / 1e9:7.0f} TF |  tn-splitk {t:.3f} ms {fl / t / 1e9:7.0f} TF (err {relerr(C, ref):.1e}, '
              f'slices {_lib.query("alm_gemm_splitk_slices", M, N, K, 1)})', flush=True)


def bench_tn():
    """split-K plan / rasterisation sweep for the two big weight-gradient shapes, on the bench-only library (tuning hook almlab_debug_splitk)."""
    import gemm_lab
    lab = gemm_lab.bind()
    T, I, Ip, D = 16384, 2730, 2736, 1024
    dU, XN2 = rnd(T, 2 * Ip), rnd(T, D)
    dW1 = torch.empty((2, I, D), dtype=F32, device=dev)
    dY2, HN = rnd(T, D), rnd(T, Ip)
    dW2 = torch.empty((D, I), dtype=F32, device=dev)
    ws = torch.empty(1 << 28, dtype=F32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def f1():
        assert lab.almlab_gemm_bf16_tn_splitk(dU.data_ptr(), XN2.data_ptr(), dW1.data_ptr(), ws.data_ptr(), I, D, T, 2 * Ip, D, D, 2, Ip, 0, I * D, 1.0, 0, st) == 0

    def f2():
        assert lab.almlab_gemm_bf16_tn_splitk(dY2.data_ptr(), HN.data_ptr(), dW2.data_ptr(), ws.data_ptr(), D, I, T, D, Ip, I, 1, 0, 0, 0, 1.0, 0, st) == 0
    fl1, fl2 = 2.0 * 2 * I * D * T, 2.0 * D * I * T
    for raster in (0, 1):
        for tile, sl in [(0, 0), (2, 2), (2, 5), (3, 1), (3, 2), (3, 3), (3, 4), (3, 5), (3, 6)]:
            lab.almlab_debug_splitk(tile, sl, raster)
            assert lab.almlab_gemm_splitk_ws_floats(I, D, T, 2) <= ws.numel() and lab.almlab_gemm_splitk_ws_floats(D, I, T, 1) <= ws.numel()
            t1, t2 = timeit(f1, iters=10), timeit(f2, iters=10)
            print(f'raster {raster} tile {tile} slices {sl}:  dW1(batched x2) {t1:.3f} ms {fl1 / t1 / 1e9:5.0f} TF | dW2 {t2:.3f} ms {fl2 / t2 / 1e9:5.0f} TF', flush=True)
    lab.almlab_debug_splitk(0, 0, 1)


def bench_wgrad():
    """every weight-gradient contraction of one layer (+ the coarse head) with the automatic split-K plan"""
    T, I, Ip, D, H, dh = 16384, 2730, 2736, 1024, 8, 64
    dU, XN2, dY, HN, AO, dQ, XN, dKV, X = rnd(T, 2 * Ip), rnd(T, D), rnd(T, D), rnd(T, Ip), rnd(T, H * dh), rnd(T, H * dh), rnd(T, D), rnd(T, 2 * dh), rnd(T, D)
    cases = [
        ('dW1 (x | gate batched)', lambda C: ops.gemm_tn_splitk(dU.view(T, 2, Ip).permute(1, 0, 2)[:, :, :I], XN2, C), (2, I, D), 2 * I, D, 2),
        ('dW2', lambda C: ops.gemm_tn_splitk(dY, HN[:, :I], C), (D, I), D, I, 1),
        ('dWo', lambda C: ops.gemm_tn_splitk(dY, AO, C), (D, H * dh), D, H * dh, 1),
        ('dWq', lambda C: ops.gemm_tn_splitk(dQ, XN, C), (H * dh, D), H * dh, D, 1),
        ('dWkv', lambda C: ops.gemm_tn_splitk(dKV, X, C), (2 * dh, D), 2 * dh, D, 1),
    ]
    tot = 0.0
    for name, fn, shape, M, N, nb in cases:
        C = torch.empty(shape, dtype=F32, device=dev)
        t = timeit(lambda: fn(C), iters=20)
        sl = _lib.query('alm_gemm_splitk_slices', M // nb, N, T, nb)
        fl = 2.0 * M * N * T
        tot += t
        print(f'{name:24s} [{M}x{N}] K={T}: slices {sl:2d}  {t * 1e3:7.1f} us  {fl / t / 1e9:5.0f} TF')
    print(f'one layer: {tot * 1e3:.1f} us -> x6 = {tot * 6:.2f} ms/step; at the NT-256 rate (866 TF) it would be {313e9 / 866e12 * 6e3:.2f} ms')


def bench_attn():
    B, N, H, dh = 8, 2048, 8, 64
    M = B * N
    Q, KV = rnd(M, H * dh), rnd(M, 2 * dh)
    K_, V_ = KV[:, :dh], KV[:, dh:]
    mask = (torch.rand(B, N, device=dev) > 0.15).to(torch.uint8)
    mask[:, 0] = 1
    fl_f = 4.0 * B * H * dh * N * (N + 1) / 2
    AO, LSE = ops.mqa_attn_fwd(Q, K_, V_, mask, B, N, H, dh)
    t = timeit(lambda: ops.mqa_attn_fwd(Q, K_, V_, mask, B, N, H, dh))
    print(f'attn fwd  B{B} N{N} H{H}: {t:.3f} ms  {fl_f / t / 1e9:.0f} TF (causal-exact flops)')
    dAO = rnd(M, H * dh)
    t = timeit(lambda: ops.mqa_attn_bwd(Q, K_, V_, mask, AO, LSE, dAO, B, N, H, dh))
    print(f'attn bwd  B{B} N{N} H{H}: {t:.3f} ms  {2.5 * fl_f / t / 1e9:.0f} TF (2.5 x fwd flops)')


def bench_hc_old():
    from audiolm_pytorch_amd import core
    B, S, N, D = 8, 4, 2048, 1024
    M = B * N
    R = torch.randn(B, S, N, D, device=dev)
    hc = dict(Bb=torch.ones(S, device=dev), Aa=torch.randn(S, S + 1, device=dev), Wa=torch.randn(D, S + 1, device=dev) * 0.02,
              sa=torch.tensor(0.01, device=dev), wb=torch.randn(D, device=dev) * 0.02, sb=torch.tensor(0.01, device=dev),
              gamma=torch.zeros(D, device=dev))
    lng = torch.ones(D, device=dev)
    Rb = R.numel() * 4
    X, XN, mean, rstd, coef = ops.hc_width_fwd(R, hc, lng, B, S, N, D, want_x=True)
    t = timeit(lambda: ops.hc_width_fwd(R, hc, lng, B, S, N, D, want_x=True))
    print(f'hc_width_fwd: {t:.3f} ms  {(Rb + 2 * M * D * 2) / t / 1e6:.0f} GB/s (alg bytes: read R, write X, XN)')
    Y = rnd(M, D)
    t = timeit(lambda: ops.hc_depth_fwd(R, Y, coef, B, S, N, D))
    print(f'hc_depth_fwd: {t:.3f} ms  {(2 * Rb + M * D * 2) / t / 1e6:.0f} GB/s (read R, Y; write R)')
    dR = torch.randn(B, S, N, D, device=dev)
    t = timeit(lambda: ops.hc_depth_bwd(dR, Y, coef, B, S, N, D))
    dY, dbeta = ops.hc_depth_bwd(dR, Y, coef, B, S, N, D)
    print(f'hc_depth_bwd: {t:.3f} ms  {(Rb + 2 * M * D * 2) / t / 1e6:.0f} GB/s (read dR, Y; write dY)')
    dX = torch.randn(M, D, device=dev)
    t = timeit(lambda: ops.hc_width_bwd(dR, dX, R, coef, dbeta, hc, B, S, N, D))
    print(f'hc_width_bwd: {t:.3f} ms  {(3 * Rb + M * D * 4) / t / 1e6:.0f} GB/s (read dR, R, dX; write dR)')


def bench_hc():
    B, S, N, D = 8, 4, 2048, 1024
    M = B * N
    R = torch.randn(B, S, N, D, device=dev)
    hc = dict(Bb=torch.ones(S, device=dev), Aa=torch.randn(S, S + 1, device=dev), Wa=torch.randn(D, S + 1, device=dev) * 0.02,
              sa=torch.tensor(0.01, device=dev), wb=torch.randn(D, device=dev) * 0.02, sb=torch.tensor(0.01, device=dev),
              gamma=torch.zeros(D, device=dev))
    lng = torch.ones(D, device=dev)
    Rb, Ab = R.numel() * 4, M * D * 2
    w = ops.hc_fwd(R, B, S, N, D, hc=hc, ln_gamma=lng)
    coef = w['coef']
    Y = rnd(M, D)
    for name, fn, byt in [
        ('hc fwd width only', lambda: ops.hc_fwd(R, B, S, N, D, hc=hc, ln_gamma=lng), Rb + 2 * Ab),
        ('hc fwd depth only', lambda: ops.hc_fwd(R, B, S, N, D, y_prev=Y, coef_prev=coef), 2 * Rb + Ab),
        ('hc fwd depth+width fused', lambda: ops.hc_fwd(R, B, S, N, D, y_prev=Y, coef_prev=coef, hc=hc, ln_gamma=lng), 2 * Rb + 3 * Ab),
        ('hc fwd final (depth+sum+LN)', lambda: ops.hc_fwd(R, B, S, N, D, y_prev=Y, coef_prev=coef, ln_gamma=lng, final=True), Rb + 2 * Ab + M * D * 4),
    ]:
        t = timeit(fn)
        print(f'{name:30s}: {t:.3f} ms  {byt / t / 1e6:.0f} GB/s algorithmic')
    dR = torch.randn(B, S, N, D, device=dev)
    dX = torch.randn(M, D, device=dev)
    dbeta = ops.hc_bwd(dR, B, S, N, D, y_prev=Y, coef_prev=coef)['dbeta']
    for name, fn, byt in [
        ('hc bwd depth only', lambda: ops.hc_bwd(dR, B, S, N, D, y_prev=Y, coef_prev=coef), Rb + 2 * Ab),
        ('hc bwd width only', lambda: ops.hc_bwd(dR, B, S, N, D, dx=dX, R=R, coef=coef, dbeta=dbeta, hc=hc), 3 * Rb + M * D * 4),
        ('hc bwd width+depth fused', lambda: ops.hc_bwd(dR, B, S, N, D, dx=dX, R=R, coef=coef, dbeta=dbeta, hc=hc, y_prev=Y, coef_prev=coef), 3 * Rb + M * D * 4 + 2 * Ab),
    ]:
        t = timeit(fn)
        print(f'{name:30s}: {t:.3f} ms  {byt / t / 1e6:.0f} GB/s algorithmic')


def bench_codec():
    """BASELINE configs[4] codec: SoundStream(codebook 4096, 8 quantizers, 24 kHz, strides 2*4*5*8) on 8 x 30 s of synthetic audio."""
    torch.manual_seed(0)
    ss = A.SoundStream(codebook_size=4096, rq_num_quantizers=8, target_sample_hz=24000, strides=(2, 4, 5, 8), use_local_attn=False)
    for r in ss.rq.rvqs:
        for q, l in enumerate(r.layers):
            l._codebook.embed.copy_(torch.randn(1, 4096, 512) * (0.5 ** q))
            l._codebook.initted.fill_(True)
    ss.to(dev)
    wave = torch.randn(8, 720000, device=dev) * 0.1
    x, _ = ss.process_input(wave)
    feats = ss.encode(x)
    t_enc = timeit(lambda: ss.encode(x), iters=3, warm=1)
    t_rvq = timeit(lambda: ss.rq(feats), iters=3, warm=1)
    t_all = timeit(lambda: ss.tokenize(wave), iters=3, warm=1)
    frames = feats.shape[0] * feats.shape[1]
    macs_enc = 8 * 720000 * 190.6e3
    macs_rvq = frames * 8 * 4096 * 512
    print(f'codec encoder   8 x 30 s @ 24 kHz: {t_enc:.2f} ms  {2 * macs_enc / t_enc / 1e9:.1f} TF fp32')
    print(f'codec rvq       {frames} frames x 8 x 4096 codes: {t_rvq:.2f} ms  {2 * macs_rvq / t_rvq / 1e9:.1f} TF fp32')
    print(f'codec tokenize  total {t_all:.2f} ms -> {frames * 8 / t_all * 1e3:.0f} codes/s, {8 * 30 / t_all * 1e3:.0f} x real time')


def bench_convs():
    """per-layer timing of the SoundStream encoder (8 x 30 s @ 24 kHz): every causal conv launch with its fp32 MFMA rate / HBM rate."""
    torch.manual_seed(0)
    ss = A.SoundStream(codebook_size=4096, rq_num_quantizers=8, target_sample_hz=24000, strides=(2, 4, 5, 8), use_local_attn=False).to(dev)
    from audiolm_pytorch_amd import soundstream as SS
    h = torch.randn(8, 1, 720000, device=dev) * 0.1
    rows = []

    def run(layer, x, **kw):
        y = layer.run(x, **kw)
        t = timeit(lambda: layer.run(x, **kw), iters=3, warm=1)
        cin, cout, k = layer.conv.in_channels, layer.conv.out_channels, layer.kernel_size
        fl = 2.0 * y.numel() * cin * k
        by = 4.0 * (x.numel() + y.numel() * (2 if kw.get('residual') is not None else 1))
        rows.append((f'{cin}->{cout} k{k} s{layer.stride} d{layer.dilation} T={x.shape[-1]}', t, fl / t / 1e9, by / t / 1e6))
        return y
    for layer in ss.encoder:
        if isinstance(layer, SS.CausalConv1d):
            h = run(layer, h)
        else:
            for sub in layer:
                if isinstance(sub, SS._ResidualFn):
                    a1 = run(sub.fn[0], h, elu=True)
                    h = run(sub.fn[2], a1, elu=True, residual=h)
                else:
                    h = run(sub, h)
    tot = sum(r[1] for r in rows)
    for name, t, tf, gb in rows:
        print(f'{name:40s} {t:7.3f} ms  {tf:6.1f} TF fp32  {gb:7.0f} GB/s')
    print(f'total {tot:.2f} ms')


def bench_e2e():
    """BASELINE configs[4]: SoundStream(codebook 4096, 8 quantizers, 24 kHz) tokenize + CoarseTransformer(codebook 4096) step on 8 x 30 s of
    synthetic audio per GPU: N = 1 + 1501 + 1 + 6750 = 8253 tokens per sequence."""
    torch.manual_seed(0)
    ss = A.SoundStream(codebook_size=4096, rq_num_quantizers=8, target_sample_hz=24000, strides=(2, 4, 5, 8), use_local_attn=False)
    for r in ss.rq.rvqs:
        for q, l in enumerate(r.layers):
            l._codebook.embed.copy_(torch.randn(1, 4096, 512) * (0.5 ** q))
            l._codebook.initted.fill_(True)
    ss.to(dev)
    model = A.CoarseTransformer(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=4096, num_coarse_quantizers=3, flash_attn=True).to(dev)
    w = A.CoarseTransformerWrapper(transformer=model, codec=ss, unique_consecutive=False, mask_prob=0.15)
    w.train()
    wave = torch.randn(8, 720000, device=dev) * 0.1
    sem = torch.randint(0, 500, (8, 1500), device=dev)

    def step():
        for p in model.parameters():
            p.grad = None
        loss = w(semantic_token_ids=sem, raw_wave=wave, return_loss=True)
        loss.backward()
        return loss
    t_tok = timeit(lambda: ss.tokenize(wave), iters=3, warm=1)
    t_all = timeit(step, iters=3, warm=1)
    ntok = 8 * 8253
    print(f'config-5 end to end (B=8, 30 s @ 24 kHz, N=8253): tokenize {t_tok:.1f} ms + transformer fwd+bwd {t_all - t_tok:.1f} ms = {t_all:.1f} ms/step '
          f'-> {ntok / t_all * 1e3:.0f} audio-tokens/s/GPU end to end, {ntok / (t_all - t_tok) * 1e3:.0f} transformer only')


def bench_models():
    """the other BASELINE configs on one GPU: configs[1] CoarseTransformer seq=1024, configs[2] FineTransformer seq=2049 (B=8, fwd+bwd)."""
    class Codec:
        rq_groups = 1
        num_quantizers = 8
    torch.manual_seed(0)
    cm = A.CoarseTransformer(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True).to(dev)
    cw = A.CoarseTransformerWrapper(transformer=cm, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
    cw.train()
    sem, coarse = torch.randint(0, 500, (8, 253), device=dev), torch.randint(0, 1024, (8, 256, 3), device=dev)

    def cstep():
        for p in cm.parameters():
            p.grad = None
        cw(semantic_token_ids=sem, coarse_token_ids=coarse, return_loss=True).backward()
    t = timeit(cstep, iters=10, warm=3)
    print(f'configs[1] CoarseTransformer d=1024 depth=6 N=1024 B=8: {t:.2f} ms/step -> {8 * 1024 / t * 1e3:.0f} audio-tokens/s')
    del cm, cw
    fm = A.FineTransformer(dim=1024, depth=6, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024, flash_attn=True).to(dev)
    fw = A.FineTransformerWrapper(transformer=fm, codec=Codec(), mask_prob=0.15)
    fw.train()
    grid = torch.randint(0, 1024, (8, 256, 8), device=dev)

    def fstep():
        for p in fm.parameters():
            p.grad = None
        fw(coarse_token_ids=grid[..., :3], fine_token_ids=grid[..., 3:], return_loss=True).backward()
    t = timeit(fstep, iters=10, warm=3)
    print(f'configs[2] FineTransformer d=1024 depth=6 N=2049 B=8: {t:.2f} ms/step -> {8 * 2049 / t * 1e3:.0f} audio-tokens/s')


def bench_bias():
    """`flash_attn=False` models: biased attention kernels vs the plain ones, the table MLP, and default-constructor model steps."""
    from audiolm_pytorch_amd import relpos
    B, N, H, dh = 8, 2048, 8, 64
    M = B * N
    Q, KV = rnd(M, H * dh), rnd(M, 2 * dh)
    K_, V_ = KV[:, :dh], KV[:, dh:]
    mask = (torch.rand(B, N, device=dev) > 0.15).to(torch.uint8)
    mask[:, 0] = 1
    dAO = rnd(M, H * dh)
    AO, LSE = ops.mqa_attn_fwd(Q, K_, V_, mask, B, N, H, dh)
    t0f = timeit(lambda: ops.mqa_attn_fwd(Q, K_, V_, mask, B, N, H, dh))
    t0b = timeit(lambda: ops.mqa_attn_bwd(Q, K_, V_, mask, AO, LSE, dAO, B, N, H, dh))
    print(f'plain attention  fwd {t0f:.3f} ms  bwd {t0b:.3f} ms')
    grid, findex = relpos.fine_index(765, 1281, 3, 5, dev)
    for name, index, LT in (('toeplitz', relpos.toeplitz_index(N, dev), 2 * N), ('coarse', relpos.toeplitz_index(N, dev, num_leading=1024), 2 * N),
                            ('fine', findex, grid.shape[0] + 1)):
        tbl = torch.randn(H, LT, device=dev)
        bias = relpos.AttnBias(tbl, *index)
        AOb, LSEb = ops.mqa_attn_fwd(Q, K_, V_, mask, B, N, H, dh, bias=bias)
        tf = timeit(lambda: ops.mqa_attn_fwd(Q, K_, V_, mask, B, N, H, dh, bias=bias))
        part = ops.attn_bias_part(B, N, H, LT, dev)
        tb = timeit(lambda: ops.mqa_attn_bwd(Q, K_, V_, mask, AOb, LSEb, dAO, B, N, H, dh, bias=bias, dtbl_part=part))
        tr = timeit(lambda: ops.attn_bias_grad_reduce(part, B, N, H))
        print(f'biased attention [{name}, LT={LT}]  fwd {tf:.3f} ms ({tf / t0f:.2f}x)  bwd {tb:.3f} ms ({tb / t0b:.2f}x)  table-grad reduce {tr:.3f} ms '
              f'(partials {part.numel() * 4 / 2**20:.0f} MiB)')
    rp = A.audiolm_pytorch.RelativePositionBias(dim=512, heads=8).to(dev)

    def table():
        for p in rp.parameters():
            p.grad = None
        b = rp(N, N)
        b.tbl.backward(torch.ones_like(b.tbl))
    t = timeit(table, iters=10, warm=3)
    print(f'RelativePositionBias table MLP (4095 rows, d=512) fwd+bwd: {t:.3f} ms')
    torch.manual_seed(0)
    for flash in (True, False):
        m = A.SemanticTransformer(dim=1024, depth=6, num_semantic_tokens=500, flash_attn=flash).to(dev)
        w = A.SemanticTransformerWrapper(transformer=m, unique_consecutive=False, mask_prob=0.15)
        w.train()
        ids = torch.randint(0, 500, (8, 2047), device=dev)

        def step():
            for p in m.parameters():
                p.grad = None
            w(semantic_token_ids=ids, return_loss=True).backward()
        t = timeit(step, iters=10, warm=3)
        print(f'SemanticTransformer d=1024 depth=6 N=2048 B=8 flash_attn={flash}: {t:.2f} ms/step -> {8 * 2048 / t * 1e3:.0f} tokens/s')
        del m, w


def bench_host():
    """is the headline step host-bound?  Python/ctypes time to ISSUE one step (no synchronisation) vs the GPU time of the step."""
    class Codec:
        rq_groups = 1
        num_quantizers = 8
    torch.manual_seed(0)
    m = A.CoarseTransformer(dim=1024, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3, flash_attn=True).to(dev)
    w = A.CoarseTransformerWrapper(transformer=m, codec=Codec(), unique_consecutive=False, mask_prob=0.15)
    w.train()
    sem, coarse = torch.randint(0, 500, (8, 509), device=dev), torch.randint(0, 1024, (8, 512, 3), device=dev)

    def step():
        m.transformer._cache.store.clear()
        for p in m.parameters():
            p.grad = None
        w(semantic_token_ids=sem, coarse_token_ids=coarse, return_loss=True).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    issue = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        step()
        issue.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'headline step: host issue time {sum(issue) / n * 1e3:.2f} ms/step (min {min(issue) * 1e3:.2f}), wall {(t2 - t0) / n * 1e3:.2f} ms/step, '
          f'GPU tail after the last issue {(t2 - t1) * 1e3:.2f} ms')
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime')
    st.print_stats(14)


def bench_gen():
    """autoregressive sampling throughput: SemanticTransformerWrapper.generate (d=1024, depth=6, B=8) with and without the kv cache"""
    torch.manual_seed(0)
    for flash in (True, False):
        m = A.SemanticTransformer(dim=1024, depth=6, num_semantic_tokens=500, flash_attn=flash).to(dev)
        w = A.SemanticTransformerWrapper(transformer=m, unique_consecutive=False)
        for use_cache, L in ((True, 512), (False, 128)):
            w.generate(max_length=8, batch_size=8, use_kv_cache=use_cache)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = w.generate(max_length=L, batch_size=8, use_kv_cache=use_cache)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'generate semantic d=1024 depth=6 B=8 flash_attn={flash} kv_cache={use_cache}: {out.shape[1]} steps in {dt:.2f} s -> '
                  f'{dt / out.shape[1] * 1e3:.2f} ms/step, {8 * out.shape[1] / dt:.0f} tokens/s')
        del m, w


def bench_gen_notebook():
    """the three sampling rates the reference's demo notebook prints (audiolm_pytorch_demo.ipynb:603-605: Tesla T4, v0.7.5, no kv cache;
    semantic dim=1024, coarse / fine dim=512, depth 6, batch 1), on this path with the kv cache"""
    class Codec:
        rq_groups = 1
        num_quantizers = 8
    torch.manual_seed(0)
    sem = A.SemanticTransformer(dim=1024, depth=6, num_semantic_tokens=500).to(dev)
    sw = A.SemanticTransformerWrapper(transformer=sem, unique_consecutive=False)
    sw.generate(max_length=8, batch_size=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ids = sw.generate(max_length=256, batch_size=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'semantic sampling (d=1024, depth 6, B=1, default ctor): {ids.shape[1] / dt:.1f} it/s   [notebook: 78.55 it/s on a T4, no kv cache]')
    coarse = A.CoarseTransformer(dim=512, depth=6, num_semantic_tokens=500, codebook_size=1024, num_coarse_quantizers=3).to(dev)
    cw = A.CoarseTransformerWrapper(transformer=coarse, codec=Codec(), unique_consecutive=False)
    s_ids = torch.randint(0, 500, (1, 256), device=dev)
    cw.generate(semantic_token_ids=s_ids, max_time_steps=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = cw.generate(semantic_token_ids=s_ids, max_time_steps=128)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'coarse sampling (d=512, depth 6, 3 quantizers, B=1): {128 / dt:.1f} it/s (1 it = 3 tokens)   [notebook: 34.83 it/s]')
    fine = A.FineTransformer(dim=512, depth=6, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024).to(dev)
    fw = A.FineTransformerWrapper(transformer=fine, codec=Codec())
    c_ids = torch.randint(0, 1024, (1, 128, 3), device=dev)
    fw.generate(coarse_token_ids=c_ids[:, :8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = fw.generate(coarse_token_ids=c_ids)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'fine sampling (d=512, depth 6, 3 + 5 quantizers, B=1): {128 / dt:.1f} it/s (1 it = 5 tokens)   [notebook: 2.91 it/s]')


def bench_misc():
    M, D, I, Ip = 16384, 1024, 2730, 2736
    U = rnd(M, 2 * Ip)
    g3 = torch.ones(I, device=dev)
    HN, mean3, rstd3 = ops.geglu_ln_fwd(U, g3, I, Ip)
    t = timeit(lambda: ops.geglu_ln_fwd(U, g3, I, Ip))
    print(f'geglu_ln_fwd: {t:.3f} ms  {(M * 3 * Ip * 2) / t / 1e6:.0f} GB/s')
    dHN = rnd(M, Ip)
    t = timeit(lambda: ops.geglu_ln_bwd(dHN, U, g3, mean3, rstd3, I, Ip))
    print(f'geglu_ln_bwd: {t:.3f} ms  {(M * 5 * Ip * 2) / t / 1e6:.0f} GB/s')
    x = rnd(M, D)
    g = torch.ones(D, device=dev)
    y, _, mean, rstd = ops.layernorm_fwd(x, g)
    t = timeit(lambda: ops.layernorm_fwd(x, g))
    print(f'layernorm_fwd bf16: {t:.3f} ms  {(M * D * 4) / t / 1e6:.0f} GB/s')
    dy = rnd(M, D)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g))
    print(f'layernorm_bwd: {t:.3f} ms  {(M * D * 8) / t / 1e6:.0f} GB/s')


if __name__ == '__main__':
    what = sys.argv[1:] or ['gemm', 'attn', 'hc', 'misc']
    print(torch.cuda.get_device_name(0), 'CUs', torch.cuda.get_device_properties(0).multi_processor_count)
    for w in what:
        globals()['bench_' + w]()
