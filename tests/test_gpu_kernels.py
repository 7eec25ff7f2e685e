"""Per-kernel parity: every C-ABI entry (through audiolm_pytorch_amd.ops -> ctypes -> libaudiolm_hip.so) against a plain fp32
PyTorch restatement of the same op / the oracle function, on a real MI355X.  Tolerances are written next to each check:
  * fp32 outputs of bf16-input MFMA contractions: <= 2e-5 of the output's max-abs (products exact, fp32 accumulate)
  * bf16 outputs: one bf16 rounding of an fp32 result -> <= 2^-8 relative to max-abs (4e-3)
  * integer / index work: bit exact
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def ops():
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import ops as o
    return o


def dev():
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0, dtype=F32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


def relmax(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def relfrob(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


# ------------------------------------------------------------------------------------------------ GEMM / transpose / pack

@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 384, 128), (300, 200, 72), (1000, 520, 1024), (130, 1025, 64), (257, 129, 2736)])
@pytest.mark.parametrize('out_f32', [True, False])
def test_gemm_nt(ops, M, N, K, out_f32):
    A, B = rnd(M, K, seed=1, dtype=BF16), rnd(N, K, seed=2, dtype=BF16)
    C = torch.full((M, N), float('nan'), dtype=F32 if out_f32 else BF16, device=dev())
    ops.gemm_nt(A, B, C)
    ref = A.double() @ B.double().t()
    err = relmax(C, ref)
    assert err <= (2e-5 if out_f32 else 4e-3), f'gemm {M}x{N}x{K} out_f32={out_f32}: rel-max err {err}'


@pytest.mark.parametrize('tile', [1, 2, 11, 13, 14, 15, 16, 17])
@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (512, 768, 1024), (300, 520, 200), (1000, 2736, 1024), (257, 300, 2736), (512, 256, 128), (640, 384, 192), (1536, 5472, 128), (4096, 2736, 192)])
def test_gemm_nt_tile_configs(ops, M, N, K, tile):
    """every block-tile configuration of the product library (1 = 128x128x64 / 4 waves, 16 = the same tile with a 4-stage DMA ring and counted waits --
    1, 2, 3, 4 and more K-steps exercise its fill / steady state / drain --, 2 = 256x256x64 / 8 waves lock-step, 13 = 256x256x64
    with staggered wave rows -- 1, 2, 3 and more K-steps exercise its slot pipeline's fill / steady state / drain --, 11 = 384x256x64) on
    full, ragged and K-tail shapes; an asymmetric B catches operand / output transposes.  (The variants that were measured and not adopted
    are tested in tests/test_gpu_gemm_lab.py against the bench-only library.)"""
    A, B = rnd(M, K, seed=11, dtype=BF16), rnd(N, K, seed=12, dtype=BF16)
    for dt, tol in ((F32, 2e-5), (BF16, 4e-3)):
        C = torch.full((M + 3, N + 5), float('nan'), dtype=dt, device=dev())
        ops.gemm_nt_tile(A, B, C[:M, :N], tile)
        ref = A.double() @ B.double().t()
        err = relmax(C[:M, :N], ref)
        assert err <= tol, f'gemm tile={tile} {M}x{N}x{K} {dt}: rel-max err {err}'
        assert bool(torch.isnan(C[M:]).all()) and bool(torch.isnan(C[:, N:]).all()), 'GEMM wrote outside its tile'


@pytest.mark.parametrize('nb,M,N,K', [(1, 2048, 128, 1024), (1, 100, 512, 520), (3, 300, 384, 264), (2, 1024, 1024, 456), (6, 64, 520, 1000)])
def test_gemm_nt_ring_tile_automatic_route_every_epilogue(ops, nb, M, N, K):
    """the 4-stage DMA-ring form of the 128 x 128 tile (tile 16) is taken AUTOMATICALLY by every NT launch of <= 256 such tiles with K >= 256 --
    batched launches, bias / alpha / accumulate epilogues, fp32 and bf16 output, M < 128 and K % 64 != 0 included, not only the un-batched `to_kv` it was
    measured on.  Each of those through the automatic entry point against fp64, with the plan query confirming that the ring kernel is what ran."""
    import ctypes
    from audiolm_pytorch_amd import _lib
    plan = (ctypes.c_int * 4)()
    _lib.query('alm_gemm_nt_plan', M, N, K, nb, 0, ctypes.cast(plan, ctypes.c_void_p))
    assert plan[0] == 16, list(plan)
    A = rnd(nb, M, K, seed=21, dtype=BF16) if nb > 1 else rnd(M, K, seed=21, dtype=BF16)
    B = rnd(nb, N, K, seed=22, dtype=BF16) if nb > 1 else rnd(N, K, seed=22, dtype=BF16)
    bias = rnd(N, seed=23)
    ref0 = A.double() @ B.double().transpose(-1, -2)
    for dt, tol in ((F32, 3e-5), (BF16, 6e-3)):
        shape = (nb, M + 3, N + 8) if nb > 1 else (M + 3, N + 8)
        for kw, ref in ((dict(), ref0), (dict(bias=bias), ref0 + bias.double()), (dict(alpha=0.37), 0.37 * ref0)):
            C = torch.full(shape, float('nan'), dtype=dt, device=dev())
            ops.gemm_nt(A, B, C[..., :M, :N], **kw)
            assert relmax(C[..., :M, :N], ref) <= tol, (dt, kw.keys())
            assert bool(torch.isnan(C[..., M:, :]).all()) and bool(torch.isnan(C[..., :, N:]).all()), 'GEMM wrote outside its tile'
        C0 = rnd(*shape, seed=24).to(dt)
        C = C0.clone()
        ops.gemm_nt(A, B, C[..., :M, :N], alpha=0.5, accumulate=True)
        assert relmax(C[..., :M, :N], C0[..., :M, :N].double() + 0.5 * ref0) <= tol
        assert torch.equal(C[..., M:, :], C0[..., M:, :]) and torch.equal(C[..., :, N:], C0[..., :, N:])


@pytest.mark.parametrize('M,N,K,slices', [(8192, 1024, 2736, 2), (8192, 1024, 512, 2), (8192, 512, 1024, 4), (16384, 512, 1024, 2), (8192, 1024, 5472, 2),
                                          (1000, 520, 1000, 3), (300, 256, 4096, 8), (515, 700, 264, 2), (2049, 1024, 2736, 4), (256, 256, 128, 2),
                                          (8192, 1024, 2736, 1)])
def test_gemm_nt_inlaunch_splitk(ops, M, N, K, slices):
    """in-launch split-K of the staggered 256 x 256 NT tile (round 6, alm_gemm_bf16_nt_inl): every K slice publishes its fp32 accumulators, the tile's last
    arriver reduces them in slice order and runs the epilogue.  Checked: values vs fp64 (full, ragged M / N, K tails, slices that the K range cannot fill),
    both output types, the bias / alpha / accumulate epilogue, nothing written outside the output, BITWISE equality of repeated launches (whichever slice
    arrives last) and of launches issued while another stream keeps part of the chip busy (uneven arrival order), and that the ticket counters are left
    zero (the next launch on the same workspace is correct)."""
    A, B = rnd(M, K, seed=31, dtype=BF16), rnd(N, K, seed=32, dtype=BF16)
    ref = A.double() @ B.double().t()
    for dt, tol in ((F32, 2e-5), (BF16, 4e-3)):
        C = torch.full((M + 3, N + 8), float('nan'), dtype=dt, device=dev())
        ops.gemm_nt_inl(A, B, C[:M, :N], slices)
        assert relmax(C[:M, :N], ref) <= tol, f'in-launch split-K {M}x{N}x{K} / {slices} {dt}'
        assert bool(torch.isnan(C[M:]).all()) and bool(torch.isnan(C[:, N:]).all()), 'wrote outside its tile'
        first = C[:M, :N].clone()
        side = torch.cuda.Stream()
        X = rnd(4096, 4096, seed=33)
        for rep in range(6):
            if rep >= 3:                                      # a second stream occupies CUs: the slices of a tile start at different times
                with torch.cuda.stream(side):
                    for _ in range(3):
                        X = torch.tanh(X)
            C2 = torch.full((M, N), float('nan'), dtype=dt, device=dev())
            ops.gemm_nt_inl(A, B, C2, slices)
            assert torch.equal(C2, first), f'in-launch split-K is not bitwise reproducible (repeat {rep})'
        torch.cuda.synchronize()
    ws = ops._nt_ws(A.device, ops._st())
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0, 'ticket counters not left zero'
    bias = rnd(N, seed=34)
    C0 = rnd(M, N, seed=35)
    C1 = C0.clone()
    ops.gemm_nt_inl(A, B, C1, slices, bias=bias, alpha=0.5, accumulate=True)
    assert relmax(C1, C0.double() + 0.5 * ref + bias.double()) <= 3e-5


@pytest.mark.parametrize('M,N,K,nb', [(8192, 1024, 2736, 1), (8192, 512, 1024, 1), (16384, 512, 1024, 1), (8192, 1024, 5472, 1), (4096, 1025, 1024, 3),
                                      (8192, 128, 1024, 1), (16384, 1024, 2736, 1)])
def test_gemm_nt_with_workspace_follows_its_plan(ops, M, N, K, nb):
    """ops.gemm_nt (alm_gemm_bf16_nt_ws) on the shapes of the timed steps: the result is right whatever the plan is, equals the forced-slices launch bit for bit
    when the plan splits (same kernel, same slices), and the batched form (per-quantizer heads: nb problems) is covered by the same ticket / slab arithmetic"""
    import ctypes
    from audiolm_pytorch_amd import _lib
    plan = (ctypes.c_int * 4)()
    _lib.query('alm_gemm_nt_plan', M, N, K, nb, 1, ctypes.cast(plan, ctypes.c_void_p))
    A, B = rnd(nb, M, K, seed=41, dtype=BF16), rnd(nb, N, K, seed=42, dtype=BF16)
    C = torch.full((nb, M, N), float('nan'), dtype=BF16, device=dev())
    ops.gemm_nt(A if nb > 1 else A[0], B if nb > 1 else B[0], C if nb > 1 else C[0])
    for z in range(nb):
        assert relmax(C[z], A[z].double() @ B[z].double().t()) <= 4e-3, (M, N, K, nb, list(plan))
    if plan[1] > 1 and nb == 1:
        C2 = torch.empty((M, N), dtype=BF16, device=dev())
        ops.gemm_nt_inl(A[0], B[0], C2, plan[1])
        assert torch.equal(C2, C[0])
    print(f'alm_gemm_nt_plan({M}, {N}, {K}, nb={nb}) -> tile {plan[0]}, {plan[1]} K slices, {plan[2]} workspace bytes, {plan[3]} workgroups')


def test_gemm_nt_group2_small_tile_then_big_tile_in_one_process(ops):
    """round-5 advisor finding: the grouped launcher kept ONE dynamic-LDS attribute flag for its four kernels, so only the first variant launched in a process
    got hipFuncAttributeMaxDynamicSharedMemorySize.  A 128 x 128 pair (64 KB of LDS) followed by a staggered 256 x 256 pair (160 KB) in the same process, both
    output types, must both launch and be right."""
    for odt in (BF16, F32):
        for (M0, N0, K0), (M1, N1, K1) in (((2048, 512, 1024), (2048, 128, 1024)), ((16384, 1024, 512), (16384, 1024, 128)), ((8192, 1024, 512), (8192, 1024, 128))):
            A0, B0 = rnd(M0, K0, seed=51, dtype=BF16), rnd(N0, K0, seed=52, dtype=BF16)
            A1, B1 = rnd(M1, K1, seed=53, dtype=BF16), rnd(N1, K1, seed=54, dtype=BF16)
            C0, C1 = torch.full((M0, N0), float('nan'), dtype=odt, device=dev()), torch.full((M1, N1), float('nan'), dtype=odt, device=dev())
            ops.gemm_nt_group2(A0, B0, C0, A1, B1, C1)
            torch.cuda.synchronize()
            tol = 2e-5 if odt == F32 else 4e-3
            assert relmax(C0, A0.double() @ B0.double().t()) <= tol and relmax(C1, A1.double() @ B1.double().t()) <= tol


@pytest.mark.parametrize('shapes', [((2048, 512, 1024), (2048, 128, 1024)),        # to_q || to_kv: both on the 128 x 128 tile, 64 + 16 tiles
                                    ((16384, 1024, 512), (16384, 1024, 128)),      # dXN_q || dX_kv: both on the staggered 256 x 256 tile, 256 + 256 tiles
                                    ((1000, 520, 200), (300, 96, 72)),             # ragged, K tails, tiles0 = 40 (multiple of 8)
                                    ((1100, 520, 200), (300, 96, 72)),             # tiles0 = 45: not a multiple of 8 -> two launches
                                    ((16384, 1024, 512), (2048, 128, 1024))])      # different tiles -> two launches
@pytest.mark.parametrize('odt', [BF16, F32])
def test_gemm_nt_group2_equals_two_launches(ops, shapes, odt):
    """alm_gemm_bf16_nt_group2: two independent NT problems in one launch == the two problems launched separately, bit for bit (every tile is computed
    by the same code), nothing written outside either output"""
    (M0, N0, K0), (M1, N1, K1) = shapes
    A0, B0 = rnd(M0, K0, seed=21, dtype=BF16), rnd(N0, K0, seed=22, dtype=BF16)
    A1, B1 = rnd(M1, K1, seed=23, dtype=BF16), rnd(N1, K1, seed=24, dtype=BF16)
    C0 = torch.full((M0 + 2, N0 + 8), float('nan'), dtype=odt, device=dev())
    C1 = torch.full((M1 + 2, N1 + 8), float('nan'), dtype=odt, device=dev())
    ops.gemm_nt_group2(A0, B0, C0[:M0, :N0], A1, B1, C1[:M1, :N1])
    R0, R1 = torch.empty((M0, N0), dtype=odt, device=dev()), torch.empty((M1, N1), dtype=odt, device=dev())
    ops.gemm_nt(A0, B0, R0)
    ops.gemm_nt(A1, B1, R1)
    assert torch.equal(C0[:M0, :N0], R0) and torch.equal(C1[:M1, :N1], R1)
    assert relmax(R0, A0.double() @ B0.double().t()) <= (2e-5 if odt == F32 else 4e-3)
    for C, M, N in ((C0, M0, N0), (C1, M1, N1)):
        assert bool(torch.isnan(C[M:]).all()) and bool(torch.isnan(C[:, N:]).all()), 'wrote outside its output'


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (512, 1024, 4096), (130, 200, 100), (1025, 256, 333), (2730, 1024, 2048), (128, 1024, 16384),
                                   (1024, 2730, 777), (8, 512, 4095), (32, 32, 29)])
def test_gemm_tn_splitk(ops, M, N, K):
    """weight-gradient contraction over the ROW index of both operands (LDS transpose reads), incl. strided views, ragged M / N / K."""
    Abig, Bbig = rnd(K, (M + 23) // 8 * 8, seed=13, dtype=BF16), rnd(K, (N + 15) // 8 * 8, seed=14, dtype=BF16)
    At, Bt = Abig[:, 8:8 + M], Bbig[:, :N]
    C = torch.full((M, N), float('nan'), dtype=F32, device=dev())
    ops.gemm_tn_splitk(At, Bt, C)
    ref = At.double().t() @ Bt.double()
    err = relmax(C, ref)
    assert err <= 3e-5, f'tn gemm {M}x{N}x{K}: rel-max err {err}'
    C0 = rnd(M, N, seed=15)
    C1 = C0.clone()
    ops.gemm_tn_splitk(At, Bt, C1, alpha=0.25, accumulate=True)
    assert relmax(C1, C0.double() + 0.25 * ref) <= 3e-5


def test_gemm_tn_splitk_batched_halves(ops):
    """the FFN W1 weight gradient: both (x | gate) halves of dU against the same XN2 in ONE launch (batch stride = Ipad columns)."""
    T, I, Ip, D = 1000, 170, 176, 128
    dU = rnd(T, 2 * Ip, seed=16, dtype=BF16)
    XN = rnd(T, D, seed=17, dtype=BF16)
    dW = torch.full((2 * I, D), float('nan'), dtype=F32, device=dev())
    ops.gemm_tn_splitk(dU.view(T, 2, Ip).permute(1, 0, 2)[:, :, :I], XN, dW.view(2, I, D))
    ref = torch.cat([dU[:, :I].double().t() @ XN.double(), dU[:, Ip:Ip + I].double().t() @ XN.double()])
    assert relmax(dW, ref) <= 3e-5


def test_gemm_bias_alpha_accumulate_strided(ops):
    M, N, K = 200, 136, 192
    Abig, Bbig = rnd(M, K + 64, seed=3, dtype=BF16), rnd(N, K + 8, seed=4, dtype=BF16)
    A, B = Abig[:, :K], Bbig[:, :K]                       # lda != K
    bias = rnd(N, seed=5)
    Cbig = rnd(M, N + 24, seed=6)
    C0 = Cbig.clone()
    ops.gemm_nt(A, B, Cbig[:, :N], bias=bias, alpha=0.5, accumulate=True)
    ref = C0[:, :N].double() + 0.5 * (A.double() @ B.double().t()) + bias.double()
    assert relmax(Cbig[:, :N], ref) <= 2e-5
    assert torch.equal(Cbig[:, N:], C0[:, N:]), 'GEMM wrote outside its N columns'


def test_gemm_batched_two_level(ops):
    G, Bn, R, C, D = 3, 2, 70, 33, 64
    A = rnd(G, R, D, seed=7, dtype=BF16)
    W = rnd(G, C, D, seed=8, dtype=BF16)
    out = torch.zeros((G, R, 40), dtype=F32, device=dev())
    ops.gemm_nt(A, W, out[:, :, :C])
    ref = torch.einsum('grd,gcd->grc', A.double(), W.double())
    assert relmax(out[:, :, :C], ref) <= 2e-5
    assert float(out[:, :, C:].abs().max()) == 0.0
    A4 = rnd(Bn, G, R, D, seed=9, dtype=BF16)
    out4 = torch.zeros((Bn, G, R, C), dtype=BF16, device=dev())
    ops.gemm_nt(A4, W.unsqueeze(0).expand(Bn, -1, -1, -1), out4)
    ref4 = torch.einsum('bgrd,gcd->bgrc', A4.double(), W.double())
    assert relmax(out4, ref4) <= 4e-3


def test_transpose_and_pack(ops):
    src = rnd(150, 70, seed=10, dtype=BF16)
    t = ops.transpose(src)
    assert t.shape == (70, 152)
    assert torch.equal(t[:, :150], src.t()) and float(t[:, 150:].float().abs().max()) == 0.0
    w = rnd(170, 64, seed=11)
    W = torch.full((176, 64), 7.0, dtype=BF16, device=dev())
    WT = torch.full((64, 176), 7.0, dtype=BF16, device=dev())
    ops.pack_weight(w, W, WT, rows_pad=176, cols_pad=64)
    assert torch.equal(W[:170], w.to(BF16)) and float(W[170:].float().abs().max()) == 0.0
    assert torch.equal(WT[:, :170], w.to(BF16).t()) and float(WT[:, 170:].float().abs().max()) == 0.0


def test_pack_weights_multi(ops):
    """several weights per launch == one alm_pack_weight launch each (incl. row-sliced sources and column-sliced destinations)"""
    ws = [rnd(70, 130, seed=90), rnd(128, 64, seed=91), rnd(341, 96, seed=92), rnd(33, 129, seed=94), rnd(5460, 1024, seed=95)]   # odd rows / cols, a real W1
    jobs, singles = [], []
    for w in ws:
        rows, cols = w.shape
        rp, cp = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
        d1, t1 = torch.zeros((rp, cp), dtype=BF16, device=dev()), torch.zeros((cp, rp), dtype=BF16, device=dev())
        d2, t2 = torch.zeros_like(d1), torch.zeros_like(t1)
        jobs.append((w, d1, t1, rp, cp))
        ops.pack_weight(w, d2, t2, rows_pad=rp, cols_pad=cp)
        singles.append((d2, t2))
    big = rnd(200, 64, seed=93)                                     # two row-halves into column-halves of one transposed buffer (the W1 layout)
    WT = torch.zeros((64, 208), dtype=BF16, device=dev())
    jobs.append((big[:100], None, WT[:, :104], 104, 64))
    jobs.append((big[100:], None, WT[:, 104:], 104, 64))
    ops.pack_weights_multi(jobs)
    for (w, d1, t1, rp, cp), (d2, t2) in zip(jobs[:5], singles):
        assert torch.equal(d1, d2) and torch.equal(t1, t2)
    assert torch.equal(WT[:, :100], big[:100].to(BF16).t()) and torch.equal(WT[:, 104:204], big[100:].to(BF16).t())
    assert float(WT[:, 100:104].float().abs().max()) == 0.0


def test_pack_weights_multi_many_jobs_one_call(ops):
    """round 5: all layers of a stack in one alm_pack_weights_multi call (40 jobs per launch of the wide kernel: 45 jobs = two launches); shapes of the
    real stack (dim 1024: Wq 512 x 1024, Wkv 128 x 1024, Wo 1024 x 512, the two W1 halves 2730 x 1024 into 2736-row slots, W2 1024 x 2730 into 2736 columns)
    plus ragged ones; every image equals the single-weight kernel's, pad rows / columns are zero"""
    shapes = [(512, 1024), (128, 1024), (1024, 512), (2730, 1024), (2730, 1024), (1024, 2730), (70, 130), (33, 129), (9, 4)] * 5
    jobs, refs = [], []
    for i, (rows, cols) in enumerate(shapes):
        w = rnd(rows, cols, seed=300 + i)
        rp, cp = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
        d1, t1 = torch.full((rp, cp), 7.0, dtype=BF16, device=dev()), torch.full((cp, rp), 7.0, dtype=BF16, device=dev())
        d2, t2 = torch.zeros_like(d1), torch.zeros_like(t1)
        jobs.append((w, d1, t1, rp, cp))
        ops.pack_weight(w, d2, t2, rows_pad=rp, cols_pad=cp)
        refs.append((d2, t2))
    ops.pack_weights_multi(jobs)
    for (w, d1, t1, rp, cp), (d2, t2) in zip(jobs, refs):
        assert torch.equal(d1, d2) and torch.equal(t1, t2), tuple(w.shape)


# ------------------------------------------------------------------------------------------------ LayerNorm / GEGLU

@pytest.mark.parametrize('D', [64, 256, 1024, 2048])
@pytest.mark.parametrize('xdt', [F32, BF16])
def test_layernorm(ops, D, xdt):
    rows = 37
    x = rnd(rows, D, seed=12, scale=2.0).add_(0.3).to(xdt)
    gamma = (1 + 0.1 * rnd(D, seed=13)).contiguous()
    y, xc, mean, rstd = ops.layernorm_fwd(x, gamma, want_copy=True)
    xr = x.float().clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, torch.zeros_like(gr))
    assert relmax(y, ref) <= 4e-3, relmax(y, ref)
    assert torch.equal(xc, x.to(BF16))
    assert torch.allclose(mean, x.float().mean(-1), atol=1e-5) and relmax(rstd, (x.float().var(-1, unbiased=False) + 1e-5).rsqrt()) <= 1e-5
    dy = rnd(rows, D, seed=14, dtype=BF16)
    extra = rnd(rows, D, seed=15, dtype=BF16)
    ref.backward(dy.float())
    dx, dg = ops.layernorm_bwd(dy, x, mean, rstd, gamma, extra=extra, dx_dtype=F32)
    assert relmax(dx, xr.grad + extra.float()) <= 2e-5, relmax(dx, xr.grad + extra.float())
    assert relmax(dg, gr.grad) <= 2e-5, relmax(dg, gr.grad)
    dxb, _ = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dx_dtype=BF16, want_dgamma=False)
    assert relmax(dxb, xr.grad) <= 4e-3


@pytest.mark.parametrize('I', [170, 682, 2730])
def test_geglu_ln(ops, I):
    rows, Ip = 19, (I + 7) // 8 * 8
    u = torch.zeros((rows, 2 * Ip), dtype=BF16, device=dev())
    xh, gh = rnd(rows, I, seed=16, dtype=BF16), rnd(rows, I, seed=17, dtype=BF16)
    u[:, :I], u[:, Ip:Ip + I] = xh, gh
    gamma = (1 + 0.1 * rnd(I, seed=18)).contiguous()
    hn, mean, rstd = ops.geglu_ln_fwd(u, gamma, I, Ip)
    xr, gr_, gam = xh.float().requires_grad_(True), gh.float().requires_grad_(True), gamma.clone().requires_grad_(True)
    h = F.gelu(gr_) * xr
    ref = F.layer_norm(h, (I,), gam, torch.zeros_like(gam))
    assert relmax(hn[:, :I], ref) <= 8e-3, relmax(hn[:, :I], ref)       # one bf16 ulp at the largest output
    assert float(hn[:, I:].float().abs().max()) == 0.0
    dhn = torch.zeros((rows, Ip), dtype=BF16, device=dev())
    dhn[:, :I] = rnd(rows, I, seed=19, dtype=BF16)
    ref.backward(dhn[:, :I].float())
    du, dg = ops.geglu_ln_bwd(dhn, u, gamma, mean, rstd, I, Ip)
    assert relmax(du[:, :I], xr.grad) <= 4e-3 and relmax(du[:, Ip:Ip + I], gr_.grad) <= 4e-3
    assert float(du[:, I:Ip].float().abs().max()) == 0.0 and float(du[:, Ip + I:].float().abs().max()) == 0.0
    assert relmax(dg, gam.grad) <= 2e-5, relmax(dg, gam.grad)


def test_colsum_reduce(ops):
    x = rnd(300, 130, seed=20)
    assert relmax(ops.colsum(x), x.double().sum(0)) <= 1e-6
    assert relmax(ops.colsum(x.to(BF16)), x.to(BF16).double().sum(0)) <= 1e-6
    # short (single stage), tall (two stages), strided rows, scale + accumulate
    for rows, cols in ((1, 7), (64, 64), (128, 7232), (129, 100), (768, 7232), (1024, 2730), (3000, 65)):
        big = rnd(rows, cols + 9, seed=rows + cols)
        xv = big[:, :cols]
        want = xv.double().sum(0)
        assert relmax(ops.colsum(xv), want) <= 2e-6, (rows, cols)
        o = rnd(cols, seed=3)
        o0 = o.clone()
        ops.colsum(xv, out=o, scale=0.25, accumulate=True)
        assert relmax(o, o0.double() + 0.25 * want) <= 2e-6, (rows, cols)
        assert relmax(ops.colsum(xv.to(BF16)), xv.to(BF16).double().sum(0)) <= 2e-6
    v = rnd(5000, seed=21)
    assert abs(float(ops.reduce_sum(v, 0.5)) - 0.5 * float(v.double().sum())) <= 1e-3


# ------------------------------------------------------------------------------------------------ attention

def _attn_ref(q2, k2, v2, mask, B, N, H, d):
    import audiolm_oracle as O
    q = q2.float().view(B, N, H, d).permute(0, 2, 1, 3)
    return O.attend(q, k2.float().view(B, N, d), v2.float().view(B, N, d), mask=mask, causal=True)


@pytest.mark.parametrize('B,N,H', [(2, 37, 8), (1, 64, 8), (2, 200, 8), (1, 512, 8), (2, 96, 4), (1, 33, 2), (3, 130, 6), (1, 1024, 8),
                                   (4, 300, 8), (8, 641, 8),      # (batch x head group) % 8 == 0 with an ODD block count (5, 11): the XCD block map's odd-count branch (round 6)
                                   (3, 2900, 8)])      # 6 (batch, head group) combos x 46 query blocks = 276 workgroups: the non-XCD block map past workgroup 256
@pytest.mark.parametrize('use_mask', [False, True])
def test_mqa_attention_fwd_bwd(ops, B, N, H, use_mask):
    d = 64
    q = rnd(B * N, H * d, seed=22, dtype=BF16)
    kv = rnd(B * N, 2 * d, seed=23, dtype=BF16)                     # k | v interleaved buffer: row stride 128
    k, v = kv[:, :d], kv[:, d:]
    mask = None
    if use_mask:
        g = torch.Generator().manual_seed(24)
        mask = (torch.rand(B, N, generator=g) > 0.25)
        mask[:, 0] = True
        mask = mask.to(dev())
    mu8 = None if mask is None else mask.contiguous().view(torch.uint8)
    o, lse = ops.mqa_attn_fwd(q, k, v, mu8, B, N, H, d)

    qf = q.float().clone().requires_grad_(True)
    kf = k.float().clone().requires_grad_(True)
    vf = v.float().clone().requires_grad_(True)
    ref = _attn_ref(qf, kf, vf, mask, B, N, H, d)                    # b h n d
    ref2 = ref.permute(0, 2, 1, 3).reshape(B * N, H * d)
    e = relmax(o, ref2)
    ef = relfrob(o, ref2)
    # P and O are rounded to bf16 once each (2^-9 relative per element): measured on MI355X rel-max 1.8-2.5e-3, rel-Frobenius 1.8-2.1e-3 over
    # these shapes; the bounds leave a factor ~1.5-2 (round 1 allowed 1.2e-2 / 2e-2 rel-max)
    assert e <= 5e-3 and ef <= 3e-3, f'attention fwd rel-max err {e}, rel-frob {ef}'
    do = rnd(B * N, H * d, seed=25, dtype=BF16)
    ref2.backward(do.float())
    dq, dkv_parts = ops.mqa_attn_bwd(q, k, v, mu8, o, lse, do, B, N, H, d)
    dkv = dkv_parts.sum(0)                                           # per-head-group partials (alm_kv_grad_pack adds them in the product path)

    for name, got, want in (('dq', dq, qf.grad), ('dk', dkv[:, :d], kf.grad), ('dv', dkv[:, d:], vf.grad)):
        e = relmax(got, want)
        ef = relfrob(got, want)                                     # measured: rel-max 0.8-5.4e-3, rel-Frobenius 1.3-2.9e-3
        assert e <= 1e-2 and ef <= 4e-3, f'attention bwd {name} rel-max err {e}, rel-frob {ef}'
    # log-sum-exp statistic (fp32): lse = logsumexp(scale * q k^T over allowed keys)
    sim = torch.einsum('bhid,bjd->bhij', qf.detach().view(B, N, H, d).permute(0, 2, 1, 3), kf.detach().view(B, N, d)) * d ** -0.5
    allow = torch.ones(N, N, dtype=torch.bool, device=dev()).tril()
    if mask is not None:
        allow = allow[None] & mask[:, None, :]
        allow = allow[:, None]
    sim = sim.masked_fill(~allow, float('-inf'))
    assert torch.allclose(lse, torch.logsumexp(sim, dim=-1), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize('N', [8253, 16385])
def test_mqa_attention_long_sequences_vs_chunked_fp64(ops, N):
    """The flash kernels at the sequence lengths bench.py's other configurations time (`--config e2e_config5`: N = 8253, `fine_t2048_q8`: N = 16 385;
    round-3 VERDICT: the kernel tests stopped at N = 1024): 129 / 257 key tiles per query block, 33 M / 134 M score pairs per head.  Reference: the math
    path of attend.py:98-146 (scores -> key mask + causal mask -> softmax -> values) in fp64 on the device, computed in query chunks of 1024 rows so
    that no (h, n, n) tensor is ever held (autograd accumulates dK / dV over the chunks).  A forgetful-style key mask (15 % of the keys dropped,
    key 0 kept) is on, as in training.  Same bounds as the short-sequence test above."""
    B, H, d = 1, 8, 64
    q = rnd(B * N, H * d, seed=122, dtype=BF16)
    kv = rnd(B * N, 2 * d, seed=123, dtype=BF16)
    k, v = kv[:, :d], kv[:, d:]
    g = torch.Generator().manual_seed(124)
    mask = torch.rand(B, N, generator=g) > 0.15
    mask[:, 0] = True
    mask = mask.to(dev())
    mu8 = mask.contiguous().view(torch.uint8)
    o, lse = ops.mqa_attn_fwd(q, k, v, mu8, B, N, H, d)
    do = rnd(B * N, H * d, seed=125, dtype=BF16)
    dq, dkv_parts = ops.mqa_attn_bwd(q, k, v, mu8, o, lse, do, B, N, H, d)
    dkv = dkv_parts.sum(0)
    torch.cuda.synchronize()

    qd = q.double().view(N, H, d).permute(1, 0, 2).contiguous().requires_grad_(True)      # (h, n, d), B = 1
    kd = k.double().contiguous().requires_grad_(True)
    vd = v.double().contiguous().requires_grad_(True)
    dod = do.double().view(N, H, d).permute(1, 0, 2)
    o_ref = torch.empty((H, N, d), dtype=torch.float64, device=dev())
    lse_ref = torch.empty((H, N), dtype=torch.float64, device=dev())
    CH = 1024
    for i0 in range(0, N, CH):
        i1 = min(N, i0 + CH)
        sim = torch.einsum('hid,jd->hij', qd[:, i0:i1], kd[:i1]) * d ** -0.5            # keys beyond the chunk's last query are causally dead
        dead = (torch.arange(i1, device=dev())[None, :] > torch.arange(i0, i1, device=dev())[:, None]) | ~mask[0, :i1][None, :]
        sim = sim.masked_fill(dead[None], float('-inf'))
        oc = torch.einsum('hij,jd->hid', sim.softmax(dim=-1), vd[:i1])
        oc.backward(dod[:, i0:i1])                                                      # accumulates into kd.grad / vd.grad; qd.grad rows i0..i1
        o_ref[:, i0:i1] = oc.detach()
        lse_ref[:, i0:i1] = torch.logsumexp(sim.detach(), dim=-1)
        del sim, oc
    ref2 = o_ref.permute(1, 0, 2).reshape(N, H * d)
    e, ef = relmax(o, ref2), relfrob(o, ref2)
    assert e <= 5e-3 and ef <= 3e-3, f'attention fwd N={N}: rel-max err {e}, rel-frob {ef}'
    assert torch.allclose(lse.double().view(H, N), lse_ref, atol=2e-3, rtol=1e-4)
    for name, got, want in (('dq', dq, qd.grad.permute(1, 0, 2).reshape(N, H * d)), ('dk', dkv[:, :d], kd.grad), ('dv', dkv[:, d:], vd.grad)):
        e, ef = relmax(got, want), relfrob(got, want)
        assert e <= 1e-2 and ef <= 4e-3, f'attention bwd {name} N={N}: rel-max err {e}, rel-frob {ef}'


def test_value_residual_and_kv_grad(ops):
    rows, d = 77, 64
    kv, kv0 = rnd(rows, 2 * d, seed=26, dtype=BF16), rnd(rows, 2 * d, seed=27, dtype=BF16)
    mix = ops.value_residual_mix(kv[:, d:], kv0[:, d:])
    assert torch.equal(mix, (0.5 * (kv[:, d:].float() + kv0[:, d:].float())).to(BF16))
    dkv = rnd(rows, 2 * d, seed=28)
    acc = torch.zeros((rows, d), dtype=F32, device=dev())
    p1 = ops.kv_grad_pack(dkv, acc, 1, d)
    assert torch.equal(p1[:, :d], dkv[:, :d].to(BF16)) and torch.equal(p1[:, d:], (0.5 * dkv[:, d:]).to(BF16))
    assert torch.equal(acc, 0.5 * dkv[:, d:])
    p2 = ops.kv_grad_pack(dkv, acc, 2, d)
    assert torch.equal(p2[:, d:], (dkv[:, d:] + acc).to(BF16))
    p0 = ops.kv_grad_pack(dkv, None, 0, d)
    assert torch.equal(p0, dkv.to(BF16))


# ------------------------------------------------------------------------------------------------ hyper-connections

def _hc_params(S, D, seed):
    return dict(gamma=0.1 * rnd(D, seed=seed), Wa=0.05 * rnd(D, S + 1, seed=seed + 1), sa=(0.1 + 0.02 * rnd(1, seed=seed + 2)).reshape(()),
                Aa=torch.cat([torch.zeros(S, 1, device=dev()), torch.eye(S, device=dev())], 1) + 0.1 * rnd(S, S + 1, seed=seed + 3),
                wb=0.05 * rnd(D, seed=seed + 4), sb=(0.1 + 0.02 * rnd(1, seed=seed + 5)).reshape(()), Bb=1 + 0.1 * rnd(S, seed=seed + 6))


def _hc_sd(hc):
    return {'norm.gamma': hc['gamma'], 'dynamic_alpha_fn': hc['Wa'], 'dynamic_alpha_scale': hc['sa'], 'static_alpha': hc['Aa'],
            'dynamic_beta_fn': hc['wb'], 'dynamic_beta_scale': hc['sb'], 'static_beta': hc['Bb']}


@pytest.mark.parametrize('S,D', [(4, 64), (4, 1024), (2, 256), (4, 512), (3, 128), (3, 1024)])
def test_hyper_connections(ops, S, D):
    import audiolm_oracle as O
    B, N = 2, 9
    R = rnd(B, S, N, D, seed=30)
    hc = {k: v.contiguous() for k, v in _hc_params(S, D, 31).items()}
    ln_gamma = (1 + 0.1 * rnd(D, seed=40)).contiguous()
    x, xn, mean, rstd, coef = ops.hc_width_fwd(R, hc, ln_gamma, B, S, N, D)

    Rr = R.clone().requires_grad_(True)
    hcr = {k: v.clone().requires_grad_(True) for k, v in hc.items()}
    gl = ln_gamma.clone().requires_grad_(True)
    bi, Rp, beta = O.hc_width(_hc_sd(hcr), '', Rr.reshape(B * S, N, D), S)
    xn_ref = O.layer_norm(bi, gl)
    assert relmax(x, bi.reshape(B * N, D)) <= 4e-3
    assert relmax(xn, xn_ref.reshape(B * N, D)) <= 4e-3
    assert relmax(coef[:, S * (S + 1):S * (S + 1) + S], beta.reshape(B * N, S)) <= 1e-5

    y = rnd(B * N, D, seed=41, dtype=BF16)
    Rn = ops.hc_depth_fwd(R, y, coef, B, S, N, D)
    yr = y.float().clone().requires_grad_(True)
    Rn_ref = O.hc_depth(yr.view(B, N, D), Rp, beta).reshape(B, S, N, D)
    assert relmax(Rn, Rn_ref) <= 1e-5, relmax(Rn, Rn_ref)

    # backward: L = <Rn, G> + <xn, Gx>
    G = rnd(B, S, N, D, seed=42)
    Gx = rnd(B * N, D, seed=43, dtype=BF16)
    ((Rn_ref * G).sum() + (xn_ref.reshape(B * N, D) * Gx.float()).sum()).backward()
    dy, dbeta = ops.hc_depth_bwd(G, y, coef, B, S, N, D)
    assert relmax(dy, yr.grad) <= 4e-3
    dx, dgl = ops.layernorm_bwd(Gx, x, mean, rstd, ln_gamma)           # x is the bf16 copy written by the width kernel
    dR, hg = ops.hc_width_bwd(G, dx, R, coef, dbeta, hc, B, S, N, D)
    e = relmax(dR, Rr.grad)
    assert e <= 6e-3, f'hc dR rel-max err {e}'                          # LN backward re-derives xhat from the bf16 copy of x
    assert relmax(dgl, gl.grad) <= 6e-3
    for k in ('Wa', 'wb', 'gamma', 'Aa', 'Bb', 'sa', 'sb'):
        e = relmax(hg[k], hcr[k].grad)
        assert e <= 1e-2, f'hc grad {k} rel-max err {e}'


@pytest.mark.parametrize('S,D,N', [(4, 1024, 19), (4, 256, 37), (2, 512, 10), (3, 256, 21)])
def test_hyper_connections_fused_modes(ops, S, D, N):
    """the fused product-path modes (depth of branch k + width of branch k+1 in one pass; final depth + stream sum + LayerNorm; width
    backward of branch k+1 + depth backward of branch k; stream-broadcast gradients) against the single-connection kernels that
    test_hyper_connections pins to the oracle."""
    B = 2
    M = B * N
    R = rnd(B, S, N, D, seed=50)
    hc1 = {k: v.contiguous() for k, v in _hc_params(S, D, 51).items()}
    hc2 = {k: v.contiguous() for k, v in _hc_params(S, D, 61).items()}
    g1, g2 = (1 + 0.1 * rnd(D, seed=52)).contiguous(), (1 + 0.1 * rnd(D, seed=53)).contiguous()
    _, _, _, _, coef1 = ops.hc_width_fwd(R, hc1, g1, B, S, N, D)
    y = rnd(M, D, seed=54, dtype=BF16)
    # forward: fused vs two-step
    R1 = ops.hc_depth_fwd(R, y, coef1, B, S, N, D)
    x2, xn2, mean2, rstd2, coef2 = ops.hc_width_fwd(R1, hc2, g2, B, S, N, D)
    f = ops.hc_fwd(R, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2)
    assert relmax(f['R'], R1) <= 1e-6
    assert relmax(f['coef'], coef2) <= 1e-6 and relmax(f['mean'], mean2) <= 1e-5 and relmax(f['rstd'], rstd2) <= 1e-5
    assert relmax(f['x'], x2) <= 4e-3 and relmax(f['xn'], xn2) <= 8e-3
    # final: depth + stream sum + LayerNorm
    fin = ops.hc_fwd(R, B, S, N, D, y_prev=y, coef_prev=coef1, ln_gamma=g2, final=True)
    xs_ref = ops.streams_reduce(R1, B, S).reshape(M, D)
    hn_ref, _, fm, fr = ops.layernorm_fwd(xs_ref, g2)
    assert relmax(fin['xs'], xs_ref) <= 1e-6 and relmax(fin['mean'], fm) <= 1e-5 and relmax(fin['rstd'], fr) <= 1e-5
    assert relmax(fin['xn'], hn_ref) <= 8e-3
    # backward: fused vs two-step
    G = rnd(B, S, N, D, seed=55)
    dx2 = rnd(M, D, seed=56)
    y2 = rnd(M, D, seed=57, dtype=BF16)
    _, dbeta2 = ops.hc_depth_bwd(G, y2, coef2, B, S, N, D)
    dR1, hg = ops.hc_width_bwd(G, dx2, R1, coef2, dbeta2, hc2, B, S, N, D)
    dy1, dbeta1 = ops.hc_depth_bwd(dR1, y, coef1, B, S, N, D)
    fb = ops.hc_bwd(G, B, S, N, D, dx=dx2, R=R1, coef=coef2, dbeta=dbeta2, hc=hc2, y_prev=y, coef_prev=coef1)
    assert relmax(fb['dR'], dR1) <= 1e-6 and relmax(fb['dbeta'], dbeta1) <= 1e-5 and relmax(fb['dy'], dy1) <= 8e-3
    for k in hg:
        assert relmax(fb['grads'][k], hg[k]) <= 1e-5, k
    # fused pre-LayerNorm backward: (dxn, extra, mean, rstd, ln gamma) in place of a pre-computed fp32 dx
    dxn = rnd(M, D, seed=59, dtype=BF16)
    ex = rnd(M, D, seed=60, dtype=BF16)
    for extra in (None, ex):
        dx_ref, dg_ref = ops.layernorm_bwd(dxn, x2, mean2, rstd2, g2, extra=extra)         # two-kernel path (LN backward on the bf16 copy of x)
        ref = ops.hc_bwd(G, B, S, N, D, dx=dx_ref, R=R1, coef=coef2, dbeta=dbeta2, hc=hc2, y_prev=y, coef_prev=coef1)
        fl = ops.hc_bwd(G, B, S, N, D, dxn=dxn, extra=extra, mean=mean2, rstd=rstd2, ln_gamma=g2, R=R1, coef=coef2, dbeta=dbeta2, hc=hc2,
                        y_prev=y, coef_prev=coef1)
        assert relmax(fl['dR'], ref['dR']) <= 8e-3, relmax(fl['dR'], ref['dR'])       # the fused path recomputes x / xhat in fp32 instead of bf16
        assert relmax(fl['grads']['ln'], dg_ref) <= 8e-3
        assert relmax(fl['dy'], ref['dy']) <= 1.5e-2 and relmax(fl['dbeta'], ref['dbeta']) <= 8e-3
        for k in hg:
            if k != 'ln':
                assert relmax(fl['grads'][k], ref['grads'][k]) <= 1e-2, k
    # un-expanded residual input (first branch): one [M, D] tensor for all streams; summed-over-streams gradient output
    xb = rnd(M, D, seed=70)
    Rb = ops.streams_expand(xb.view(B, N, D), B, S)
    fe = ops.hc_fwd(Rb, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2)
    fbc = ops.hc_fwd(xb, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2, rin_bcast=True)
    assert torch.equal(fe['R'], fbc['R']) and torch.equal(fe['xn'], fbc['xn']) and torch.equal(fe['coef'], fbc['coef'])
    _, _, _, _, coefb = ops.hc_width_fwd(Rb, hc2, g2, B, S, N, D)
    be = ops.hc_bwd(G, B, S, N, D, dx=dx2, R=Rb, coef=coefb, dbeta=dbeta2, hc=hc2)
    bb = ops.hc_bwd(G, B, S, N, D, dx=dx2, R=xb, coef=coefb, dbeta=dbeta2, hc=hc2, r_bcast=True, sum_only=True)
    assert relmax(bb['dsum'], ops.streams_reduce(be['dR'], B, S).reshape(M, D)) <= 1e-6
    for k in hg:
        assert relmax(bb['grads'][k], be['grads'][k]) <= 1e-6, k
    # stream-broadcast gradient (right after the final stream sum)
    gb = rnd(M, D, seed=58)
    Gb = ops.streams_expand(gb.view(B, N, D), B, S)
    dyb, dbb = ops.hc_depth_bwd(gb, y2, coef2, B, S, N, D, bcast=True)
    dye, dbe = ops.hc_depth_bwd(Gb, y2, coef2, B, S, N, D)
    assert torch.equal(dyb, dye) and relmax(dbb, dbe) <= 1e-6
    dRb = ops.hc_bwd(gb, B, S, N, D, bcast=True, dx=dx2, R=R1, coef=coef2, dbeta=dbb, hc=hc2)['dR']
    dRe, _ = ops.hc_width_bwd(Gb, dx2, R1, coef2, dbe, hc2, B, S, N, D)
    assert relmax(dRb, dRe) <= 1e-6


@pytest.mark.parametrize('S,D,N', [(4, 1024, 19), (4, 1024, 700), (4, 256, 1501), (2, 512, 330), (3, 128, 45)])
def test_hyper_connections_bf16_streams(ops, S, D, N):
    """bf16 storage of the residual streams and their gradients (the dtype trainer.py:1241's autocast gives the reference's streams): the same
    kernels with a bf16 HBM image (software-prefetching variants where no broadcast operand is involved; N = 700 / 1501 give every workgroup
    several tokens, odd counts, a ragged tail) against the fp32-stream kernels fed the SAME (bf16-representable) values.  What may differ:
    the rounding of a stored bf16 output (one bf16 ulp: the two template instantiations contract their fp32 FMAs differently, which flips
    round-to-nearest ties), and in the fused forward the width connection reading the rounded streams."""
    def ulp1(got_bf16, want_f32):                                   # stored bf16 image within one bf16 ulp of the fp32 result
        return relmax(got_bf16.float(), want_f32) <= 4e-3 and float((got_bf16.float() - want_f32).abs().max()) <= float(want_f32.abs().max()) * 2 ** -7

    B = 2
    M = B * N
    Rb = rnd(B, S, N, D, seed=80, dtype=BF16)
    Rf = Rb.float()
    hc1 = {k: v.contiguous() for k, v in _hc_params(S, D, 81).items()}
    hc2 = {k: v.contiguous() for k, v in _hc_params(S, D, 82).items()}
    g1, g2 = (1 + 0.1 * rnd(D, seed=83)).contiguous(), (1 + 0.1 * rnd(D, seed=84)).contiguous()
    y = rnd(M, D, seed=85, dtype=BF16)
    # width only (mode 2)
    wf = ops.hc_fwd(Rf, B, S, N, D, hc=hc1, ln_gamma=g1)
    wb = ops.hc_fwd(Rb, B, S, N, D, hc=hc1, ln_gamma=g1, r_dtype=BF16)
    assert relmax(wb['coef'], wf['coef']) <= 1e-5 and relmax(wb['x'], wf['x']) <= 4e-3 and relmax(wb['xn'], wf['xn']) <= 8e-3
    coef1 = wf['coef']
    # depth + width fused (mode 3): R_out is rounded to bf16, the next width connection reads the rounded streams
    ff = ops.hc_fwd(Rf, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2)
    fb = ops.hc_fwd(Rb, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2, r_dtype=BF16)
    assert fb['R'].dtype == BF16 and ulp1(fb['R'], ff['R'])
    w2 = ops.hc_fwd(fb['R'].float(), B, S, N, D, hc=hc2, ln_gamma=g2)            # fp32 kernel on the rounded streams = what the fused bf16 pass computes
    assert relmax(fb['coef'], w2['coef']) <= 1e-5 and relmax(fb['xn'], w2['xn']) <= 8e-3 and relmax(fb['mean'], w2['mean']) <= 1e-4
    # final: depth + stream sum + LayerNorm, fp32 hidden states out
    nf = ops.hc_fwd(Rf, B, S, N, D, y_prev=y, coef_prev=coef1, ln_gamma=g2, final=True, final_f32=True)
    nb = ops.hc_fwd(Rb, B, S, N, D, y_prev=y, coef_prev=coef1, ln_gamma=g2, final=True, final_f32=True, r_dtype=BF16)
    assert nb['xn'].dtype == F32 and relmax(nb['xn'], nf['xn']) <= 1e-5 and relmax(nb['xs'], nf['xs']) <= 1e-5
    hn_bf = ops.hc_fwd(Rf, B, S, N, D, y_prev=y, coef_prev=coef1, ln_gamma=g2, final=True)['xn']
    assert ulp1(hn_bf, nf['xn'])
    # backward (fused LayerNorm backward + width backward + depth backward of the previous branch): same inputs, bf16 vs fp32 images
    R1b, coef2, mean2, rstd2 = fb['R'], fb['coef'], fb['mean'], fb['rstd']
    Gb = rnd(B, S, N, D, seed=86, dtype=BF16)
    dxn = rnd(M, D, seed=87, dtype=BF16)
    ex = rnd(M, D, seed=88, dtype=BF16)
    dbeta2 = rnd(M, S, seed=89)
    for extra in (None, ex):
        rf = ops.hc_bwd(Gb.float(), B, S, N, D, dxn=dxn, extra=extra, mean=mean2, rstd=rstd2, ln_gamma=g2, R=R1b.float(), coef=coef2, dbeta=dbeta2,
                        hc=hc2, y_prev=y, coef_prev=coef1)
        rb = ops.hc_bwd(Gb, B, S, N, D, dxn=dxn, extra=extra, mean=mean2, rstd=rstd2, ln_gamma=g2, R=R1b, coef=coef2, dbeta=dbeta2,
                        hc=hc2, y_prev=y, coef_prev=coef1, r_dtype=BF16)
        assert rb['dR'].dtype == BF16 and ulp1(rb['dR'], rf['dR'])
        assert relmax(rb['dbeta'], rf['dbeta']) <= 1e-5 and relmax(rb['dy'], rf['dy']) <= 8e-3
        for k in rf['grads']:
            assert relmax(rb['grads'][k], rf['grads'][k]) <= 1e-4, k
        # the same call with dxn / extra / y as views that start 8 bytes into a wider buffer: rows no longer begin on 16-byte boundaries, so the LDS-DMA
        # variant of hc_bwd (round 4; 16-byte pieces) must hand over to the register-prefetch kernel -- same results
        def off8(t):
            wide = torch.zeros((t.shape[0], t.shape[1] + 8), dtype=t.dtype, device=t.device)
            wide[:, 4:4 + t.shape[1]] = t
            v = wide[:, 4:4 + t.shape[1]]
            assert v.data_ptr() % 16 == 8
            return v
        rv = ops.hc_bwd(Gb, B, S, N, D, dxn=off8(dxn), extra=None if extra is None else off8(extra), mean=mean2, rstd=rstd2, ln_gamma=g2, R=R1b, coef=coef2,
                        dbeta=dbeta2, hc=hc2, y_prev=off8(y), coef_prev=coef1, r_dtype=BF16)
        assert relmax(rv['dR'].float(), rb['dR'].float()) <= 4e-3 and relmax(rv['dy'].float(), rb['dy'].float()) <= 8e-3 and relmax(rv['dbeta'], rb['dbeta']) <= 1e-5
        for k in rb['grads']:
            assert relmax(rv['grads'][k], rb['grads'][k]) <= 1e-5, k
    # width-only backward (mode 2), depth-only backward (mode 1)
    dx2 = rnd(M, D, seed=90)
    rf = ops.hc_bwd(Gb.float(), B, S, N, D, dx=dx2, R=R1b.float(), coef=coef2, dbeta=dbeta2, hc=hc2)
    rb = ops.hc_bwd(Gb, B, S, N, D, dx=dx2, R=R1b, coef=coef2, dbeta=dbeta2, hc=hc2, r_dtype=BF16)
    assert ulp1(rb['dR'], rf['dR'])
    for k in rf['grads']:
        assert relmax(rb['grads'][k], rf['grads'][k]) <= 1e-4, k
    dyf, dbf = ops.hc_depth_bwd(Gb.float(), y, coef2, B, S, N, D)
    dyb, dbb = ops.hc_depth_bwd(Gb, y, coef2, B, S, N, D)
    assert relmax(dyb, dyf) <= 8e-3 and relmax(dbb, dbf) <= 1e-5
    # broadcast operands stay fp32 next to bf16 streams: first branch (x for every stream) and the gradient of the final stream sum
    xb = rnd(M, D, seed=91)
    f1 = ops.hc_fwd(xb, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2, rin_bcast=True, r_dtype=BF16)
    f0 = ops.hc_fwd(xb, B, S, N, D, y_prev=y, coef_prev=coef1, hc=hc2, ln_gamma=g2, rin_bcast=True)
    assert ulp1(f1['R'], f0['R'])
    gb = rnd(M, D, seed=92)
    b1 = ops.hc_bwd(gb, B, S, N, D, bcast=True, dx=dx2, R=R1b, coef=coef2, dbeta=dbeta2, hc=hc2, r_dtype=BF16)
    b0 = ops.hc_bwd(gb, B, S, N, D, bcast=True, dx=dx2, R=R1b.float(), coef=coef2, dbeta=dbeta2, hc=hc2)
    assert ulp1(b1['dR'], b0['dR'])
    _, _, _, _, coefx = ops.hc_width_fwd(ops.streams_expand(xb.view(B, N, D), B, S), hc2, g2, B, S, N, D)
    s1 = ops.hc_bwd(Gb, B, S, N, D, dx=dx2, R=xb, coef=coefx, dbeta=dbeta2, hc=hc2, r_bcast=True, sum_only=True, r_dtype=BF16)
    s0 = ops.hc_bwd(Gb.float(), B, S, N, D, dx=dx2, R=xb, coef=coefx, dbeta=dbeta2, hc=hc2, r_bcast=True, sum_only=True)
    assert relmax(s1['dsum'], s0['dsum']) <= 1e-6


@pytest.mark.parametrize('S,D,N', [(4, 1024, 700), (2, 512, 333)])
def test_hyper_connection_param_grads_batched_finish(ops, S, D, N):
    """round 4: the parameter-gradient finish of several width connections in two launches (alm_hc_param_grads_batched: column sums of every branch's
    partial rows + the gradient kernel over (element block, branch)) against the per-branch finish (alm_colsum_partial + alm_hc_param_grads): the same
    sums in another chunking -- equal to fp32 summation noise.  Three branches with different parameters and different row counts (first-branch
    broadcast variant, plain, fused-LayerNorm + depth)."""
    B = 2
    M = B * N
    hcs = [{k: v.contiguous() for k, v in _hc_params(S, D, 300 + 10 * i).items()} for i in range(3)]
    gs = [(1 + 0.1 * rnd(D, seed=340 + i)).contiguous() for i in range(3)]
    Rb = rnd(B, S, N, D, seed=350, dtype=BF16)
    y = rnd(M, D, seed=351, dtype=BF16)
    w = [ops.hc_fwd(Rb, B, S, N, D, hc=hcs[i], ln_gamma=gs[i], r_dtype=BF16) for i in range(3)]
    Gb = rnd(B, S, N, D, seed=352, dtype=BF16)
    dxn = rnd(M, D, seed=353, dtype=BF16)
    dx = rnd(M, D, seed=354)
    dbeta = rnd(M, S, seed=355)
    xb = rnd(M, D, seed=356)
    _, _, _, _, coefx = ops.hc_width_fwd(ops.streams_expand(xb.view(B, N, D), B, S), hcs[2], gs[2], B, S, N, D)
    calls = [dict(dRn=Gb, kw=dict(dxn=dxn, mean=w[0]['mean'], rstd=w[0]['rstd'], ln_gamma=gs[0], R=Rb, coef=w[0]['coef'], dbeta=dbeta, hc=hcs[0], y_prev=y,
                                  coef_prev=w[1]['coef'], r_dtype=BF16)),
             dict(dRn=Gb, kw=dict(dx=dx, R=Rb, coef=w[1]['coef'], dbeta=dbeta, hc=hcs[1], r_dtype=BF16)),
             dict(dRn=Gb, kw=dict(dx=dx, R=xb, coef=coefx, dbeta=dbeta, hc=hcs[2], r_bcast=True, sum_only=True, r_dtype=BF16))]
    ref = [ops.hc_bwd(c['dRn'], B, S, N, D, **c['kw']) for c in calls]
    dfr = [ops.hc_bwd(c['dRn'], B, S, N, D, defer_grads=True, **c['kw']) for c in calls]
    assert all(d['grads'] is None and d['part'] is not None for d in dfr)
    got = ops.hc_param_grads_batched([d['part'] for d in dfr], S, D)
    for i, (r, g) in enumerate(zip(ref, got)):
        assert r['grads'].keys() == g.keys()
        for k in g:
            assert relmax(g[k], r['grads'][k]) <= 2e-5, (i, k, relmax(g[k], r['grads'][k]))
        for k in ('dR', 'dsum', 'dy', 'dbeta'):
            assert (r[k] is None) == (dfr[i][k] is None)
            if r[k] is not None:
                assert torch.equal(r[k], dfr[i][k]), (i, k)


def test_layernorm_fp32_output_and_fp32_upstream_gradient(ops):
    """the final LayerNorm of the stack writes fp32 hidden states (logit heads) and receives an fp32 gradient from them"""
    rows, D = 37, 1024
    x = rnd(rows, D, seed=93)
    gamma = (1 + 0.1 * rnd(D, seed=94)).contiguous()
    y32, _, mean, rstd = ops.layernorm_fwd(x, gamma, out_f32=True)
    ref = F.layer_norm(x, (D,), gamma, None, 1e-5)
    assert y32.dtype == F32 and relmax(y32, ref) <= 2e-6
    ybf, _, _, _ = ops.layernorm_fwd(x, gamma)
    assert torch.equal(ybf, y32.to(BF16))
    dy = rnd(rows, D, seed=95)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    (F.layer_norm(xr, (D,), gr, None, 1e-5) * dy).sum().backward()
    dx, dg = ops.layernorm_bwd(dy, x, mean, rstd, gamma)
    assert relmax(dx, xr.grad) <= 1e-5 and relmax(dg, gr.grad) <= 1e-5


@pytest.mark.parametrize('kcat', [True, False])
@pytest.mark.parametrize('G,C,D,R', [(3, 1025, 1024, 300), (1, 501, 256, 77), (5, 64, 128, 40)])
def test_split_bf16_logit_heads(ops, G, C, D, R, kcat, monkeypatch):
    """logit heads on split-bf16 operands (heads.head_logits): fp32 hidden states x fp32 master weights to ~16 mantissa bits, vs an fp64
    contraction.  Plain bf16 operands give ~3e-3; the split form must be >= 100 x better.  kcat: the three products as ONE GEMM over K-concatenated
    operands (round 5, default) or as three accumulating launches."""
    from audiolm_pytorch_amd import core, heads
    monkeypatch.setattr(heads, 'HEAD_KCAT', kcat)
    hn = rnd(R * 2, D, seed=96)
    w = rnd(G, C, D, seed=97, scale=2.0 / math.sqrt(D))
    bias = rnd(C, seed=98, scale=0.1) if G == 1 else None
    g = torch.Generator().manual_seed(99)
    idx = torch.randint(-1, R * 2, (G, R), generator=g).to(torch.int32).to(dev())
    hg, logits = heads.head_logits(hn, w, bias, idx, core.WeightCache(), ('head', 't'))
    rows = torch.where(idx.reshape(-1)[:, None] >= 0, hn[idx.reshape(-1).clamp(min=0).long()], torch.zeros(1, D, device=dev())).view(G, R, D)
    ref = torch.einsum('grd,gcd->grc', rows.double(), w.double())
    if bias is not None:
        ref = ref + bias.double()
    got = logits.view(G, R, -1)[..., :C]
    e = relmax(got, ref)
    assert e <= 3e-5, e
    assert torch.equal(hg.reshape(G, R, D), rows.to(BF16))                 # the saved high halves = bf16 image of the gathered rows
    hi, lo = ops.gather_split(hn)
    assert torch.equal(hi, hn.to(BF16)) and relmax(hi.float() + lo.float(), hn) <= 2 ** -15


def test_embed_assemble_out_of_range_ids_never_touch_memory(ops):
    """an id outside its table (nn.Embedding raises IndexError, reference audiolm_pytorch.py:709 / :901-906): zero vector, skipped in the
    backward, device error flag raised -- never an out-of-bounds access"""
    D = 64
    t0, t1 = rnd(11, D, seed=50), rnd(7, D, seed=51)
    src_a = torch.tensor([3, 11, 0xffffff, 10, (5 << 24) | 1], dtype=torch.int32, device=dev())      # row 11 of an 11-row table, huge row, table 5 of 2
    src_b = torch.tensor([(1 << 24) | 6, (1 << 24) | 7, -1, -1, -1], dtype=torch.int32, device=dev())
    ops.check_device_errors(dev())
    out = ops.embed_assemble([t0, t1], src_a, src_b, 5, D)
    assert torch.equal(out[0], t0[3] + t1[6]) and float(out[1].abs().max()) == 0 and float(out[2].abs().max()) == 0
    assert torch.equal(out[3], t0[10]) and float(out[4].abs().max()) == 0
    with pytest.raises(IndexError):
        ops.check_device_errors(dev())
    ops.check_device_errors(dev())                                           # the flag is cleared by the check
    grads = [torch.zeros_like(t0), torch.zeros_like(t1)]
    ops.embed_scatter_add(grads, src_a, src_b, torch.ones(5, D, device=dev()), 1.0, 5, D)
    assert float(grads[0].sum()) == 2 * D and float(grads[1].sum()) == D


def test_streams_and_elementwise(ops):
    B, S, N, D = 2, 4, 5, 64
    x = rnd(B, N, D, seed=44)
    R = ops.streams_expand(x, B, S)
    assert torch.equal(R, x[:, None].expand(B, S, N, D))
    R2 = rnd(B, S, N, D, seed=45)
    assert torch.allclose(ops.streams_reduce(R2, B, S), R2.sum(1), atol=1e-6)
    y = rnd(B * N, D, seed=46, dtype=BF16)
    xf = x.reshape(B * N, D)
    assert torch.equal(ops.residual_add(xf, y), xf + y.float())
    a, b = rnd(B * N, D, seed=47), rnd(B * N, D, seed=48)
    assert torch.equal(ops.f32_to_bf16(a, b), (a + b).to(BF16)) and torch.equal(ops.f32_to_bf16(a), a.to(BF16))
    assert torch.equal(ops.add_f32(a, b), a + b)


# ------------------------------------------------------------------------------------------------ token-id side

def test_embed_assemble_and_scatter(ops):
    D = 64
    t0, t1, t2 = rnd(11, D, seed=50), rnd(7, D, seed=51), rnd(1, D, seed=52)
    g = torch.Generator().manual_seed(53)
    rows = 40
    ia = torch.randint(-1, 11, (rows,), generator=g)
    ib = torch.randint(-1, 7, (rows,), generator=g)
    src_a = torch.where(ia >= 0, ia, torch.full_like(ia, -1)).to(torch.int32).to(dev())
    src_b = torch.where(ib >= 0, ib + (1 << 24), torch.full_like(ib, -1)).to(torch.int32).to(dev())
    src_a[0] = 2 << 24
    out = ops.embed_assemble([t0, t1, t2], src_a, src_b, rows, D)
    ref = torch.zeros(rows, D, device=dev())
    for r in range(rows):
        a, b = int(src_a[r]), int(src_b[r])
        if a >= 0:
            ref[r] += [t0, t1, t2][a >> 24][a & 0xffffff]
        if b >= 0:
            ref[r] += [t0, t1, t2][b >> 24][b & 0xffffff]
    assert torch.equal(out, ref)
    dout = rnd(rows, D, seed=54)
    grads = [torch.zeros_like(t) for t in (t0, t1, t2)]
    ops.embed_scatter_add(grads, src_a, src_b, dout, 0.1, rows, D)
    refg = [torch.zeros_like(t) for t in (t0, t1, t2)]
    for r in range(rows):
        for c in (int(src_a[r]), int(src_b[r])):
            if c >= 0:
                refg[c >> 24][c & 0xffffff] += 0.1 * dout[r]
    for a, b in zip(grads, refg):
        assert torch.allclose(a, b, atol=1e-5)


def test_embed_scatter_large_and_small_tables(ops):
    """codebook-sized tables take one atomic per (token, element); the few-row tables every token adds to are pre-summed per workgroup in LDS.
    300 rows x 320 columns: more than one 128-row chunk, a ragged last chunk and a ragged 256-column block; includes out-of-range codes."""
    D, rows = 320, 300
    big, small, one = rnd(100, D, seed=60), rnd(3, D, seed=61), rnd(1, D, seed=62)
    g = torch.Generator().manual_seed(63)
    ia = torch.randint(-1, 100, (rows,), generator=g)
    ib = torch.randint(-1, 3, (rows,), generator=g)
    src_a = torch.where(ia >= 0, ia, torch.full_like(ia, -1)).to(torch.int32)
    src_b = torch.where(ib >= 0, ib + (1 << 24), torch.full_like(ib, -1)).to(torch.int32)
    src_a[5] = 2 << 24                       # the one-row table
    src_a[6] = 100                           # out of range in the big table: skipped
    src_b[7] = (1 << 24) | 3                 # out of range in the small table: skipped
    src_a, src_b = src_a.to(dev()), src_b.to(dev())
    dout = rnd(rows, D, seed=64)
    grads = [torch.zeros_like(t) for t in (big, small, one)]
    ops.embed_scatter_add(grads, src_a, src_b, dout, 0.5, rows, D)
    refg = [torch.zeros_like(t, dtype=torch.float64) for t in (big, small, one)]
    for r in range(rows):
        for c in (int(src_a[r]), int(src_b[r])):
            if c >= 0 and (c & 0xffffff) < refg[c >> 24].shape[0]:
                refg[c >> 24][c & 0xffffff] += 0.5 * dout[r].double()
    for a, b in zip(grads, refg):
        assert torch.allclose(a.double(), b, atol=1e-4)


def _scatter_ref(tables, src_a, src_b, dout, alpha):
    refg = [torch.zeros_like(t, dtype=torch.float64) for t in tables]
    for r in range(dout.shape[0]):
        for c in (int(src_a[r]), int(src_b[r])):
            if c >= 0 and (c >> 24) < len(refg) and (c & 0xffffff) < refg[c >> 24].shape[0]:
                refg[c >> 24][c & 0xffffff] += alpha * dout[r].double()
    return refg


@pytest.mark.parametrize('D,rows,nbig', [(320, 300, 100), (1024, 1500, 301), (64, 5, 40), (2048, 700, 77)])
def test_embed_scatter_owned_is_exact_deterministic_and_needs_no_zero_fill(ops, D, rows, nbig):
    """alm_embed_scatter_owned (destination-owned, no atomics): equals the fp64 scatter, writes EVERY row (the buffers start as NaN garbage), two runs are
    bit-identical.  Ragged chunks / column blocks / last row group (nbig % 8 != 0), out-of-range codes, a one-row table, D > 1024 (two column blocks)."""
    big, small, one, big2 = rnd(nbig, D, seed=60), rnd(3, D, seed=61), rnd(1, D, seed=62), rnd(45, D, seed=65)
    g = torch.Generator().manual_seed(63)
    ia = torch.randint(-1, nbig, (rows,), generator=g)
    ib = torch.randint(-1, 3, (rows,), generator=g)
    src_a = torch.where(ia >= 0, ia, torch.full_like(ia, -1)).to(torch.int32)
    src_b = torch.where(ib >= 0, ib + (1 << 24), torch.full_like(ib, -1)).to(torch.int32)
    if rows > 8:
        src_a[5] = 2 << 24                       # the one-row table
        src_a[6] = nbig                          # out of range in the big table: skipped
        src_b[7] = (1 << 24) | 3                 # out of range in the small table: skipped
        src_b[8] = (3 << 24) | 44                # a second large table (45 rows), last row
        src_a[3] = (3 << 24) | 44
    src_a, src_b = src_a.to(dev()), src_b.to(dev())
    dout = rnd(rows, D, seed=64)
    tables = (big, small, one, big2)
    outs = []
    for _ in range(2):
        grads = [torch.full_like(t, float('nan')) for t in tables]
        ops.embed_scatter_owned(grads, src_a, src_b, dout, 0.5, rows, D)
        outs.append(grads)
    for a, b in zip(outs[0], _scatter_ref([t.cpu() for t in tables], src_a.cpu(), src_b.cpu(), dout.cpu(), 0.5)):
        assert bool(torch.isfinite(a).all())
        assert torch.allclose(a.double().cpu(), b, atol=1e-4)
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # and against the atomic form (same sums in another order)
    grads = [torch.zeros_like(t) for t in tables]
    ops.embed_scatter_add(grads, src_a, src_b, dout, 0.5, rows, D)
    assert all(torch.allclose(a, b, atol=1e-4) for a, b in zip(outs[0], grads))


def test_embed_scatter_owned_falls_back_to_the_atomic_kernel_above_the_workspace_cap(ops, monkeypatch):
    """outside the owned kernel's range (rows >= 2^28 / a chunk-partial workspace above ops.EMBED_SCATTER_WS_CAP) the call must not raise
    mid-backward: the atomic kernel runs on zeroed tables (return value False) and gives the same sums"""
    D, rows, nbig = 320, 300, 100
    big, small = rnd(nbig, D, seed=60), rnd(3, D, seed=61)
    g = torch.Generator().manual_seed(63)
    src_a = torch.randint(-1, nbig, (rows,), generator=g).to(torch.int32).to(dev())
    ib = torch.randint(-1, 3, (rows,), generator=g)
    src_b = torch.where(ib >= 0, ib + (1 << 24), torch.full_like(ib, -1)).to(torch.int32).to(dev())
    dout = rnd(rows, D, seed=64)
    ref = [torch.full_like(t, float('nan')) for t in (big, small)]
    assert ops.embed_scatter_owned(ref, src_a, src_b, dout, 0.5, rows, D) is True
    monkeypatch.setattr(ops, 'EMBED_SCATTER_WS_CAP', 0)
    got = [torch.full_like(t, float('nan')) for t in (big, small)]
    assert ops.embed_scatter_owned(got, src_a, src_b, dout, 0.5, rows, D) is False
    assert all(bool(torch.isfinite(a).all()) and torch.allclose(a, b, atol=1e-4) for a, b in zip(got, ref))


def test_embed_scatter_owned_no_tokens(ops):
    """rows == 0: every table gradient is written as zeros (no scan of an empty code array)"""
    D = 64
    big, small = torch.full((40, D), float('nan'), device=dev()), torch.full((2, D), float('nan'), device=dev())
    e = torch.empty(0, dtype=torch.int32, device=dev())
    ops.embed_scatter_owned([big, small], e, e, torch.empty(0, D, device=dev()), 1.0, 0, D)
    assert float(big.abs().sum()) == 0 and float(small.abs().sum()) == 0


def test_embed_scatter_owned_skewed_ids(ops):
    """every token on ONE destination row (the pending list of the owning wave is flushed many times) and a table with more rows than tokens"""
    D, rows = 256, 3000
    big, small = rnd(5000, D, seed=70), rnd(2, D, seed=71)
    src_a = torch.full((rows,), 4321, dtype=torch.int32, device=dev())
    src_b = torch.full((rows,), (1 << 24) | 1, dtype=torch.int32, device=dev())
    dout = rnd(rows, D, seed=72)
    grads = [torch.full_like(big, float('nan')), torch.full_like(small, float('nan'))]
    ops.embed_scatter_owned(grads, src_a, src_b, dout, 1.0, rows, D)
    ref = dout.double().sum(0)
    assert torch.allclose(grads[0][4321].double(), ref, atol=1e-3) and torch.allclose(grads[1][1].double(), ref, atol=1e-3)
    assert float(grads[0].abs().sum() - grads[0][4321].abs().sum()) == 0 and float(grads[1][0].abs().max()) == 0


def _scatter_ref_fast(tables, src_a, src_b, dout, alpha):
    """fp64 scatter by index_add_ (the per-token loop of _scatter_ref is too slow for tens of thousands of tokens)"""
    refg = [torch.zeros(t.shape, dtype=torch.float64) for t in tables]
    for src in (src_a.long(), src_b.long()):
        tb, row = src >> 24, src & 0xffffff
        for i, g in enumerate(refg):
            m = (src >= 0) & (tb == i) & (row < g.shape[0])
            g.index_add_(0, row[m], alpha * dout[m].double())
    return refg


@pytest.mark.parametrize('case', ['four-hot-rows', 'threshold', 'more-hot-rows-than-listed', 'two-column-blocks', 'clustered', 'no-small-table'])
def test_embed_scatter_owned_hot_rows(ops, case):
    """skewed ids (round 6): rows with more than 128 tokens are listed by the histogram's last arriver and summed by (row, token-range) workers + a ticketed
    ordered reduction (csrc/embed_ce.hip, HOT_*).  Against the fp64 scatter and the atomic kernel; written everywhere (NaN garbage before); bitwise run to run."""
    g = torch.Generator().manual_seed(80)
    D, rows, nbig, nbig2 = 1024, 20000, 3000, 700
    if case == 'two-column-blocks':
        D, rows = 2048, 6000
    if case == 'more-hot-rows-than-listed':
        D, rows = 256, 90000
    a = torch.randint(0, nbig, (rows,), generator=g)
    b = torch.randint(0, 3, (rows,), generator=g) + (1 << 24)
    if case in ('four-hot-rows', 'two-column-blocks', 'no-small-table'):
        m = torch.rand(rows, generator=g) < 0.5
        a[m] = torch.randint(0, 4, (int(m.sum()),), generator=g) * 7
        m2 = torch.rand(rows, generator=g) < 0.2                             # the second code array feeds a hot row of a SECOND large table too
        b[m2] = (2 << 24) | 699
    elif case == 'threshold':                                                # exactly 128 tokens (not hot), 129 (hot), and one row that takes a third of the batch
        a[:] = torch.randint(10, nbig, (rows,), generator=g)
        perm = torch.randperm(rows, generator=g)
        a[perm[:128]] = 3
        a[perm[128:257]] = 4
        a[perm[257:257 + rows // 3]] = 5
    elif case == 'more-hot-rows-than-listed':                                # 300 rows x 200 tokens: 256 are listed, 44 stay with their owners
        perm = torch.randperm(rows, generator=g)
        for r in range(300):
            a[perm[r * 200:(r + 1) * 200]] = r * 9 + 1
    elif case == 'clustered':                                                # every hot token inside ONE token range (a long silence): one part does all the work
        a[4000:9000] = 77
    src_a, src_b = a.to(torch.int32).to(dev()), b.to(torch.int32).to(dev())
    dout = rnd(rows, D, seed=81)
    shapes = [(nbig, D), (3, D), (nbig2, D)]
    if case == 'no-small-table':
        shapes = [(nbig, D), (40, D), (nbig2, D)]                            # 40 rows > the 32-row small path: three large tables, no small one
        src_b = torch.where(src_b >> 24 == 1, torch.full_like(src_b, (1 << 24) | 17), src_b).to(torch.int32)
    outs = []
    for _ in range(2):
        grads = [torch.full(sh, float('nan'), device=dev()) for sh in shapes]
        assert ops.embed_scatter_owned(grads, src_a, src_b, dout, 0.5, rows, D) is True
        outs.append(grads)
    ref = _scatter_ref_fast([torch.empty(sh) for sh in shapes], src_a.cpu(), src_b.cpu(), dout.cpu(), 0.5)
    for x, r in zip(outs[0], ref):
        assert bool(torch.isfinite(x).all())
        scale = float(r.abs().max()) + 1e-30
        assert float((x.double().cpu() - r).abs().max()) <= 1e-4 * scale, case        # fp32 sums of up to ~10 k terms
    assert all(torch.equal(x, y) for x, y in zip(*outs))
    grads = [torch.zeros(sh, device=dev()) for sh in shapes]
    ops.embed_scatter_add(grads, src_a, src_b, dout, 0.5, rows, D)
    for x, y in zip(outs[0], grads):
        assert float((x - y).abs().max()) <= 1e-4 * (float(y.abs().max()) + 1e-30)


@pytest.mark.parametrize('nbytes,offset', [(1, 0), (15, 1), (16, 0), (17, 3), (4096, 0), (10260, 4), (1 << 20, 0), ((1 << 20) + 5, 7)])
def test_memset_zero_is_a_kernel_with_ragged_ends_and_survives_graph_replay(ops, nbytes, offset):
    """alm_memset_zero zero-fills exactly [ptr, ptr + bytes) -- unaligned start, ragged tail, neighbours untouched -- and, captured into a hipGraph, clears on
    EVERY replay (a small hipMemsetAsync node does not on ROCm 7.0: scripts/debug/memset_node_probe.py; the owned embedding scatter and hc_bwd rely on this)"""
    buf = torch.full((nbytes + 64,), 0x5a, dtype=torch.uint8, device=dev())
    view = buf[offset:offset + nbytes]
    ops.memset_zero(view)
    assert int(view.max()) == 0 and bool((buf[:offset] == 0x5a).all()) and bool((buf[offset + nbytes:] == 0x5a).all())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        buf.fill_(7); ops.memset_zero(view); view.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.fill_(7)
        ops.memset_zero(view)
        view.add_(1)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert int(view.min()) == 1 and int(view.max()) == 1
        assert bool((buf[:offset] == 7).all()) and bool((buf[offset + nbytes:] == 7).all())


def test_gather_scatter_rows(ops):
    x = rnd(50, 64, seed=55, dtype=BF16)
    idx = torch.tensor([3, -1, 49, 0, 7, -1], dtype=torch.int32, device=dev())
    g = ops.gather_rows(x, idx)
    assert torch.equal(g[0], x[3]) and float(g[1].float().abs().max()) == 0 and torch.equal(g[2], x[49])
    out = torch.zeros_like(x)
    ops.scatter_rows(g, idx, out)
    assert torch.equal(out[3], x[3]) and torch.equal(out[49], x[49]) and float(out[1].float().abs().max()) == 0


@pytest.mark.parametrize('C', [21, 501, 1025])
def test_cross_entropy(ops, C):
    rows, Cp = 45, (C + 7) // 8 * 8
    logits = torch.full((rows, Cp), 1e9, dtype=F32, device=dev())       # poison the pad columns: they must never be read
    logits[:, :C] = rnd(rows, C, seed=56, scale=3.0)
    g = torch.Generator().manual_seed(57)
    labels = torch.randint(0, C, (rows,), generator=g).to(dev())
    labels[::7] = -1
    loss_rows, lse = ops.cross_entropy_fwd(logits, labels, C)
    lr = logits[:, :C].clone().requires_grad_(True)
    ref_rows = F.cross_entropy(lr, labels, ignore_index=-1, reduction='none')
    assert torch.allclose(loss_rows, ref_rows, atol=1e-5, rtol=1e-5)
    gs = torch.tensor(0.37, device=dev())
    (ref_rows.sum() * 0.37).backward()
    d = ops.cross_entropy_bwd(logits, labels, lse, gs, C, Cp)
    assert relmax(d[:, :C], lr.grad) <= 4e-3 and float(d[:, C:].float().abs().max()) == 0.0
    assert float(d[::7].float().abs().max()) == 0.0


def test_package_import_before_torch():
    """Import order must not matter: the package loads torch's bundled HIP runtime before its own library (a process that imported the
    package first used to fail every launch with hipErrorNoDevice)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import audiolm_pytorch_amd as A\nimport torch\nfrom audiolm_pytorch_amd import ops\n"
            "x = torch.randn(64, 1024, device='cuda')\nout = ops.layernorm_fwd(x, torch.ones(1024, device='cuda'))\n"
            "torch.cuda.synchronize()\nassert torch.isfinite(out[0].float()).all()\nprint('import-order-ok')\n")
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'import-order-ok' in r.stdout, (r.stdout[-300:], r.stderr[-800:])


def test_gemm_tn_batched_two_level_strided(ops):
    """alm_gemm_bf16_tn_batched: C[l][h] = At[l][h]^T @ Bt[l][(h)] over strided 4-D views -- the stacked-buffer form of the deferred weight gradients
    (dW1: the x / gate halves of dU against the layer's XN, Bt broadcast over the halves; dW2: padded HN columns sliced off)."""
    L, M, I, Ip, D = 3, 1000, 170, 176, 128
    dU = rnd(L, M, 2 * Ip, seed=501, dtype=BF16)
    XN = rnd(L, M, D, seed=502, dtype=BF16)
    C = torch.empty((L, 2, I, D), dtype=torch.float32, device=dev())
    At = dU.view(L, M, 2, Ip).permute(0, 2, 1, 3)[..., :I]
    ops.gemm_tn_batched(At, XN.unsqueeze(1), C)
    ref = torch.einsum('lhki,lkd->lhid', At.float(), XN.float())
    assert relmax(C, ref) <= 2e-5, relmax(C, ref)
    HN = rnd(L, M, Ip, seed=503, dtype=BF16)
    dY = rnd(L, M, D, seed=504, dtype=BF16)
    C2 = torch.empty((L, 1, D, I), dtype=torch.float32, device=dev())
    ops.gemm_tn_batched(dY.unsqueeze(1), HN[..., :I].unsqueeze(1), C2)
    ref2 = torch.einsum('lkd,lki->ldi', dY.float(), HN[..., :I].float())
    assert relmax(C2[:, 0], ref2) <= 2e-5
    # a long contraction that takes the split-K path, accumulate on top
    K = 9000
    A3, B3 = rnd(2, 1, K, 64, seed=505, dtype=BF16), rnd(2, 1, K, 256, seed=506, dtype=BF16)
    C3 = rnd(2, 1, 64, 256, seed=507)
    base = C3.clone()
    ops.gemm_tn_batched(A3, B3, C3, alpha=0.5, accumulate=True)
    ref3 = base.double() + 0.5 * torch.einsum('lhkm,lhkn->lhmn', A3.double(), B3.double())
    assert relmax(C3, ref3) <= 2e-5


@pytest.mark.parametrize('K', [4160, 8256])                                        # 8256: the full-K launch takes the 4-wave tile (>= 8192 per slice)
@pytest.mark.parametrize('accumulate', [False, True])
def test_gemm_tn_batched_hybrid_plan(ops, accumulate, K):
    """alm_gemm_bf16_tn_batched, hybrid plan: the panels that fill whole waves of the 256 CUs run at full K straight into C, the last problem's
    remaining row (m-major: dW1) or column (n-major: dW2) blocks are a deep split-K launch on the sub-matrix.  Shapes of the benchmark's weight
    gradients (inner width 2730, padded 2736, 3 layers x 2 halves / 6 layers), a shorter contraction."""
    from audiolm_pytorch_amd import _lib
    L, I, Ip, D = 3, 2730, 2736, 1024
    assert _lib.query('alm_gemm_splitk_ws_floats', I, D, K, 2 * L) > 0
    dU = rnd(L, K, 2 * Ip, seed=601, dtype=BF16)
    XN = rnd(L, K, D, seed=602, dtype=BF16)
    At = dU.view(L, K, 2, Ip).permute(0, 2, 1, 3)[..., :I]                     # 2 * 3 problems x 11 x 4 tiles = 264: 64 panels + a 2-panel tail
    C = rnd(L, 2, I, D, seed=603) if accumulate else torch.empty((L, 2, I, D), dtype=torch.float32, device=dev())
    base = C.clone()
    ops.gemm_tn_batched(At, XN.unsqueeze(1), C, alpha=0.5, accumulate=accumulate)
    ref = 0.5 * torch.einsum('lhki,lkd->lhid', At.float(), XN.float()) + (base if accumulate else 0.)
    assert relmax(C, ref) <= 3e-5, relmax(C, ref)
    L2 = 6                                                                     # n-major: 6 problems x 4 x 11 tiles = 264
    HN = rnd(L2, K, Ip, seed=604, dtype=BF16)
    dY = rnd(L2, K, D, seed=605, dtype=BF16)
    C2 = rnd(L2, 1, D, I, seed=606) if accumulate else torch.empty((L2, 1, D, I), dtype=torch.float32, device=dev())
    base2 = C2.clone()
    ops.gemm_tn_batched(dY.unsqueeze(1), HN[..., :I].unsqueeze(1), C2, accumulate=accumulate)
    ref2 = torch.einsum('lkd,lki->ldi', dY.float(), HN[..., :I].float()) + (base2[:, 0] if accumulate else 0.)
    assert relmax(C2[:, 0], ref2) <= 3e-5, relmax(C2[:, 0], ref2)


@pytest.mark.parametrize('N', [2, 7, 100, 2048, 2049, 5000, 16384])
def test_forgetful_mask_kernel_matches_topk(ops, N):
    """alm_forgetful_mask vs the reference's own formulation (audiolm_pytorch.py:82-89: rand[:, 0] = -max; topk; scatter): same keys dropped."""
    B = 5
    g = torch.Generator().manual_seed(700 + N)
    score = torch.randn(B, N, generator=g).to(dev())
    for prob in (0.15, 0.5, 1.0):
        k = min(int(N * prob), N - 1)
        if k <= 0:
            continue
        pre = (torch.rand(B, N, generator=g) > 0.1).to(dev())                   # the key-padding mask it is ANDed into
        keep = pre.clone()
        ops.forgetful_mask_(keep, score, k)
        s2 = score.clone()
        s2[:, 0] = -torch.finfo(s2.dtype).max
        ref = torch.ones(B, N, dtype=torch.bool, device=dev()).scatter_(1, s2.topk(k, dim=-1).indices, False) & pre
        assert torch.equal(keep, ref), (N, prob, int((keep != ref).sum()))


def test_forgetful_mask_kernel_ties_and_seeded_run(ops):
    """equal scores at the threshold: exactly k keys go, none of them column 0, and no kept key beats a dropped one; a seeded run of the host function
    masks the same keys as the ATen formulation of the reference"""
    import audiolm_pytorch_amd.audiolm_pytorch as AP
    B, N, k = 3, 300, 45
    g = torch.Generator().manual_seed(77)
    score = torch.randint(0, 20, (B, N), generator=g).float().to(dev())          # heavy ties
    keep = torch.ones(B, N, dtype=torch.bool, device=dev())
    ops.forgetful_mask_(keep, score, k)
    assert bool(keep[:, 0].all()) and torch.equal((~keep).sum(1), torch.full((B,), k, device=dev()))
    for b in range(B):
        dropped, kept = score[b][~keep[b]], score[b][1:][keep[b][1:]]
        assert float(dropped.min()) >= float(kept.max())
    torch.manual_seed(5)
    m1 = AP.generate_mask_with_prob((4, 777), 0.15, dev())
    torch.manual_seed(5)
    s = torch.randn((4, 777), device=dev())
    s[:, 0] = -torch.finfo(s.dtype).max
    m2 = torch.ones((4, 777), dtype=torch.bool, device=dev()).scatter_(1, s.topk(min(int(777 * 0.15), 776), dim=-1).indices, False)
    assert torch.equal(m1, m2)


def test_loss_combine_kernel_and_its_backward(ops):
    """alm_loss_combine vs the formulation it replaces: per head `sum / (labels != -1).sum().clamp(min=1)`, then the wrappers' weighted sum; the
    autograd Function's backward is the scale per group"""
    from audiolm_pytorch_amd import heads
    g = torch.Generator().manual_seed(9)
    labels = [torch.randint(-1, 50, (7, 33), generator=g).to(dev()), torch.randint(-1, 3, (5, 1000), generator=g).to(dev()),
              torch.full((2, 9), -1, dtype=torch.int64, device=dev())]                                     # the last one: no valid label at all
    sums = [(torch.rand((), generator=g) * 100).to(dev()).requires_grad_() for _ in labels]
    w = (0.3, 1.7, 0.5)
    loss = heads.combine_losses(sums, labels, w)
    ref = sum(wi * s.detach().double() / (l != -1).sum().clamp(min=1) for wi, s, l in zip(w, sums, labels))
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    (loss * 2.5).backward()
    for wi, s, l in zip(w, sums, labels):
        want = 2.5 * wi / max(int((l != -1).sum()), 1)
        assert abs(float(s.grad) - want) <= 1e-6 * want


@pytest.mark.parametrize('B,ns0,nf,Q', [(3, 17, 5, 3), (1, 1, 1, 1), (4, 100, 33, 4), (2, 509, 512, 3)])
def test_coarse_prepare_kernel_matches_the_wrapper_bookkeeping(ops, B, ns0, nf, Q):
    """alm_coarse_prepare vs the ATen formulation it replaces (CoarseTransformerWrapper.forward + CoarseTransformer._assemble): labels with the eos
    appended, key mask, source codes -- with pad ids (-1) and stray eos ids inside the semantic rows"""
    import torch.nn.functional as F
    import audiolm_pytorch_amd.audiolm_pytorch as AP
    g = torch.Generator().manual_seed(31 * B + ns0)
    C, n_sem, pad = 64, 50, -1
    sem_eos, coarse_eos = n_sem, C
    sem = torch.randint(0, n_sem, (B, ns0), generator=g)
    sem[torch.rand(B, ns0, generator=g) < 0.15] = pad
    sem[torch.rand(B, ns0, generator=g) < 0.05] = sem_eos
    coarse = torch.randint(0, C, (B, nf * Q), generator=g)
    sem, coarse = sem.to(dev()), coarse.to(dev())
    sl, cl, src_a, keep = ops.coarse_prepare(sem, coarse, pad, sem_eos, coarse_eos, Q, C)
    sem1, coarse1 = F.pad(sem, (0, 1), value=sem_eos), F.pad(coarse, (0, 1), value=coarse_eos)
    assert torch.equal(sl, sem1) and torch.equal(cl, coarse1)
    m = (sem1 != pad) & (sem1 != sem_eos)
    sem_clean = sem1.masked_fill(~m, 0)
    nc = coarse.shape[1]
    assert torch.equal(keep, F.pad(m, (1, nc + 1), value=True))
    rows = coarse.to(torch.int32) + AP._quantizer_row_offsets(nc, Q, C, coarse.device)
    ref = torch.cat((AP._const_code(3, B, coarse.device), sem_clean.to(torch.int32).clamp(min=-1), AP._const_code(4, B, coarse.device), AP._code(1, rows)), dim=1)
    assert torch.equal(src_a, ref)


@pytest.mark.parametrize('B,n0', [(3, 17), (1, 0), (4, 100), (2, 1021)])
def test_semantic_prepare_kernel_matches_the_wrapper_bookkeeping(ops, B, n0):
    """alm_semantic_prepare (round 6) vs the ATen formulation it replaces (SemanticTransformerWrapper.forward :1536-1548 + SemanticTransformer._tokens): labels
    with the eos appended, [start | ids] source codes, pad ids (-1) as the zero vector; a strided (sliced) id tensor"""
    import audiolm_pytorch_amd.audiolm_pytorch as AP
    g = torch.Generator().manual_seed(17 * B + n0)
    n_sem, pad = 50, -1
    wide = torch.randint(0, n_sem, (B, n0 + 3), generator=g)
    wide[torch.rand(B, n0 + 3, generator=g) < 0.15] = pad
    sem = wide.to(dev())[:, :n0]                                     # row stride n0 + 3
    labels, src_a = ops.semantic_prepare(sem, n_sem, n_sem + 1)
    ref_labels = AP.append_eos_id(sem, n_sem)
    assert torch.equal(labels, ref_labels)
    ref_src = torch.cat((AP._const_code(1, B, sem.device), ref_labels[:, :-1].to(torch.int32)), dim=1)
    assert torch.equal(src_a, ref_src)


def test_semantic_wrapper_training_step_same_loss_and_gradients_with_and_without_the_prepare_kernel(monkeypatch):
    """the fused bookkeeping path of SemanticTransformerWrapper.forward (ops.semantic_prepare) == the ATen path: loss and every gradient bit for bit
    (same kernels downstream, same forgetful-mask draw under the same seed)"""
    import audiolm_pytorch_amd as A
    import audiolm_pytorch_amd.audiolm_pytorch as AP
    torch.manual_seed(0)
    model = A.SemanticTransformer(num_semantic_tokens=50, dim=128, depth=2, flash_attn=True).to(dev())
    w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=False, mask_prob=0.15)
    w.train()
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 50, (3, 40), generator=g)
    ids[0, 30:] = -1
    ids = ids.to(dev())
    res = []
    for on in (True, False):
        monkeypatch.setattr(AP, '_SEMANTIC_PREPARE', on)
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(123)
        loss = w(semantic_token_ids=ids, return_loss=True)
        loss.backward()
        res.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0]), (float(res[0][0]), float(res[1][0]))
    assert res[0][1].keys() == res[1][1].keys() and all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])


@pytest.mark.parametrize('B,n,eos', [(1, 1, None), (3, 40, 50), (4, 255, None), (2, 256, 50), (5, 257, 50), (8, 1500, 50), (2, 3000, None), (3, 0, 50)])
def test_unique_consecutive_kernel_matches_torch_per_row(ops, B, n, eos):
    """alm_unique_consecutive_i64 == the reference's batch_unique_consecutive (audiolm_pytorch.py:162-164: torch.unique_consecutive per row, right-padded)
    applied after append_eos_id: collapsed rows, pad tail, row lengths; strided input rows; runs that cross the 256-id rounds of the kernel"""
    g = torch.Generator().manual_seed(31 * B + n)
    wide = torch.randint(0, 50, (B, n + 2), generator=g)
    for _ in range(3):                                                   # runs: ~40 % of the positions repeat their predecessor (up to 4 long)
        m = torch.rand(B, n + 2, generator=g) < 0.4
        wide[:, 1:] = torch.where(m[:, 1:], wide[:, :-1], wide[:, 1:])
    if n > 300:
        wide[0, 250:262] = 7                                             # one run across a round boundary
        wide[1 % B, :n] = 3                                              # a whole row collapses to one id
    ids = wide.to(dev())[:, :n]                                          # row stride n + 2
    out, lengths = ops.unique_consecutive(ids, eos, -1)
    rows = []
    for r in ids.cpu():
        if eos is not None:
            r = torch.cat((r, torch.tensor([eos])))
        rows.append(torch.unique_consecutive(r))
    assert lengths.cpu().tolist() == [r.numel() for r in rows]
    W = n + (eos is not None)
    assert out.shape == (B, W)
    for b, r in enumerate(rows):
        assert torch.equal(out[b, :r.numel()].cpu(), r)
        assert bool((out[b, r.numel():] == -1).all())


def test_prepare_kernels_on_rows_that_already_hold_their_eos(ops):
    """sem_has_eos / has_eos (unique_consecutive = True): coarse_prepare / semantic_prepare on [ids | eos | pad ...] rows == the ATen bookkeeping of the
    wrappers on those rows (CoarseTransformerWrapper.forward :1797-1806 + _assemble; SemanticTransformerWrapper.forward :1541-1548 + _tokens)"""
    import audiolm_pytorch_amd.audiolm_pytorch as AP
    g = torch.Generator().manual_seed(9)
    B, L, nf, Q, C, n_sem = 3, 17, 5, 3, 64, 50
    sem = torch.randint(0, n_sem, (B, L + 2), generator=g)
    for b, ln in enumerate((17, 9, 1)):                                  # row b: ln - 1 ids, the eos, pads
        sem[b, ln - 1] = n_sem
        sem[b, ln:] = -1
    coarse = torch.randint(0, C, (B, nf * Q), generator=g).to(dev())
    sem = sem.to(dev())[:, :L]                                           # row stride L + 2
    sl, cl, src_a, keep = ops.coarse_prepare(sem, coarse, -1, n_sem, C, Q, C, sem_has_eos=True)
    assert torch.equal(sl, sem) and torch.equal(cl, AP.append_eos_id(coarse, C))
    mask = (sem != -1) & (sem != n_sem)
    assert torch.equal(keep, F.pad(mask, (1, coarse.shape[1] + 1), value=True))
    rows = coarse.to(torch.int32) + AP._quantizer_row_offsets(coarse.shape[1], Q, C, sem.device)
    ref = torch.cat((AP._const_code(3, B, sem.device), sem.masked_fill(~mask, 0).to(torch.int32), AP._const_code(4, B, sem.device), AP._code(1, rows)), dim=1)
    assert torch.equal(src_a, ref)
    labels, src = ops.semantic_prepare(sem, n_sem, n_sem + 1, has_eos=True)
    assert torch.equal(labels, sem)
    assert torch.equal(src, torch.cat((AP._const_code(1, B, sem.device), sem[:, :-1].to(torch.int32).clamp(min=-1)), dim=1))


def test_semantic_wrapper_unique_consecutive_training_step_fused_equals_unfused(monkeypatch):
    """unique_consecutive = True (the reference's default): the fused path (one collapse launch + one host read + ops.semantic_prepare) == the ATen path
    (append_eos_id, batch_unique_consecutive, slicing): loss and every gradient bit for bit"""
    import audiolm_pytorch_amd as A
    import audiolm_pytorch_amd.audiolm_pytorch as AP
    torch.manual_seed(0)
    model = A.SemanticTransformer(num_semantic_tokens=50, dim=128, depth=2, flash_attn=True).to(dev())
    w = A.SemanticTransformerWrapper(transformer=model, unique_consecutive=True, mask_prob=0.15)
    w.train()
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 50, (3, 40), generator=g)
    ids[0, 10:25] = 4                                                    # runs: the rows collapse to different lengths
    ids[1, 1::2] = ids[1, 0::2]
    ids = ids.to(dev())
    res = []
    for on in (True, False):
        monkeypatch.setattr(AP, '_SEMANTIC_PREPARE', on)
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(123)
        loss = w(semantic_token_ids=ids, return_loss=True)
        loss.backward()
        res.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0]), (float(res[0][0]), float(res[1][0]))
    assert res[0][1].keys() == res[1][1].keys() and all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])
    # and the host formulation of the collapse itself (a CPU tensor takes the per-row torch.unique_consecutive loop)
    assert torch.equal(AP.batch_unique_consecutive(ids, pad_value=-1).cpu(), AP.batch_unique_consecutive(ids.cpu(), pad_value=-1))


@pytest.mark.parametrize('nh', [2, 4])
def test_mqa_attention_backward_with_either_dkv_workgroup_shape(nh):
    """round 6: the dK/dV kernel runs 4 heads per workgroup (N < 8192) or 2 (two workgroups per CU, H / 2 partial sets: alm_mqa_bwd_parts); the choice is a
    process-wide policy read once, so each forced value (ALM_ATTN_DKV_NH) runs the attention / bias / dropout kernel tests in its own interpreter"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALM_ATTN_DKV_NH=str(nh))
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_kernels.py', 'tests/test_gpu_bias.py', '-m', 'gpu', '-q', '-x', '-p', 'no:cacheprovider',
                        '-k', '(mqa_attention_fwd_bwd or biased_attention_fwd_bwd or structured_bias) and not either_dkv'], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert ' passed' in r.stdout
