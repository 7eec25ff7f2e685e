"""SoundStream tokenize path on the MI355X (csrc/codec.hip through the C ABI) vs the CPU oracle and the golden fixture produced by the
REAL reference encoder (first-party, pinned) around the restated residual VQ (third-party, parity unpinned -- SURVEY.md §8(c)).

Tolerances: fp32 everywhere (exact-fp32 MFMA, a different summation order than the CPU convolution): activations rel-max 2e-5;
code indices are integers -- bit-exact on the golden fixture; on large random problems an index may differ from the oracle only where
the two candidates' distances are within float rounding of each other (checked explicitly, and < 0.1 % of the frames)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import audiolm_oracle as O
from common import GOLDEN_DIR, synth_state_dict

pytestmark = pytest.mark.gpu
F32 = torch.float32


def dev():
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.fixture(scope='module')
def ops():
    from audiolm_pytorch_amd import ops as _ops
    return _ops


@pytest.mark.parametrize('B,Cin,Cout,T,k,stride,dil', [(2, 1, 32, 1000, 7, 1, 1), (1, 32, 32, 777, 7, 1, 9), (2, 32, 64, 640, 4, 2, 1),
                                                       (1, 64, 128, 512, 8, 4, 1), (1, 128, 256, 400, 10, 5, 1), (1, 4, 8, 333, 7, 1, 3),
                                                       (2, 48, 40, 300, 3, 1, 1), (1, 256, 512, 64, 16, 8, 1)])
def test_causal_conv1d(ops, B, Cin, Cout, T, k, stride, dil):
    x, w, b = rnd(B, Cin, T, seed=1), rnd(Cout, Cin, k, seed=2, scale=(Cin * k) ** -0.5), rnd(Cout, seed=3, scale=0.1)
    ref = O.causal_conv1d(x, w, b, dilation=dil, stride=stride)
    wp = ops.conv1d_pack(w.to(dev()))
    out = ops.conv1d_causal(x.to(dev()), wp, b.to(dev()), Cout, k, stride=stride, dilation=dil)
    assert out.shape == ref.shape
    assert relmax(out, ref) <= 2e-5
    if stride == 1 and Cin == Cout:                                 # the ResidualUnit epilogue: ELU + residual
        out2 = ops.conv1d_causal(x.to(dev()), wp, b.to(dev()), Cout, k, stride=1, dilation=dil, elu=True, residual=x.to(dev()))
        assert relmax(out2, F.elu(ref) + x) <= 2e-5


@pytest.mark.parametrize('B,C,T,dil', [(2, 32, 1000, 1), (1, 32, 777, 9), (2, 64, 641, 3), (1, 128, 515, 9), (1, 256, 300, 3), (1, 96, 200, 1), (1, 160, 130, 1),
                                       (1, 32, 55, 9), (3, 64, 64, 1)])
def test_fused_residual_unit(ops, B, C, T, dil):
    """alm_resunit_causal (round 6; reference soundstream.py:362-369): x + ELU(conv_k1(ELU(conv_k7,dil(x)))) in ONE launch, the intermediate in registers.
    BITWISE equal to the two alm_conv1d_causal launches it replaces (same fma chains on the exact-fp32 matrix core -- the code indices downstream cannot
    move), equal to the fp32 CPU oracle within the conv tolerance, ragged time tails and every channel-block count C / 32 = 1 .. 8."""
    x = rnd(B, C, T, seed=11)
    w7, b7 = rnd(C, C, 7, seed=12, scale=(7 * C) ** -0.5), rnd(C, seed=13, scale=0.1)
    w1, b1 = rnd(C, C, 1, seed=14, scale=C ** -0.5), rnd(C, seed=15, scale=0.1)
    ref = F.elu(O.causal_conv1d(F.elu(O.causal_conv1d(x, w7, b7, dilation=dil)), w1, b1)) + x
    xd = x.to(dev())
    w7p, w1p = ops.conv1d_pack(w7.to(dev())), ops.conv1d_pack(w1.to(dev()))
    h = ops.conv1d_causal(xd, w7p, b7.to(dev()), C, 7, dilation=dil, elu=True)
    two = ops.conv1d_causal(h, w1p, b1.to(dev()), C, 1, elu=True, residual=xd)
    one = ops.resunit_causal(xd, w7p, b7.to(dev()), w1p, b1.to(dev()), 7, dil)
    assert one.shape == two.shape == ref.shape
    assert torch.equal(one, two), f'fused unit differs from the two launches: max |d| {float((one - two).abs().max()):.3e}'
    assert relmax(one, ref) <= 2e-5

def _check_indices(idx, x, cbs):
    """idx (T, Q) vs the oracle on the same fp32 inputs: equal, or a float-rounding tie (replays the oracle's residual path)."""
    ref = O.rvq_encode(x[None], cbs)[0]
    bad = (idx != ref).any(dim=-1)
    nbad = int(bad.sum())
    if nbad == 0:
        return 0
    assert nbad <= max(1, int(1e-3 * idx.shape[0])), f'{nbad} of {idx.shape[0]} frames differ from the oracle'
    for t in torch.nonzero(bad).flatten().tolist():
        r = x[t].clone()
        for q, E in enumerate(cbs):
            d = ((r[None] - E) ** 2).sum(-1).sqrt()
            a, b = int(idx[t, q]), int(ref[t, q])
            if a != b:
                assert abs(float(d[a]) - float(d[b])) <= 1e-4 * max(1.0, float(d[b])), (t, q, float(d[a]), float(d[b]))
                break                                               # residual paths diverge after the first differing stage
            r = r - E[a]
    return nbad


@pytest.mark.parametrize('T,d,C,Q', [(50, 16, 32, 4), (333, 64, 100, 3), (1000, 512, 1024, 8), (129, 24, 37, 2)])
def test_rvq_encode(ops, T, d, C, Q):
    x = rnd(T, d, seed=4)
    E = rnd(Q, C, d, seed=5)
    E[1:] *= 0.6 ** torch.arange(1, Q)[:, None, None]               # later stages quantize smaller residuals
    Ed = E.to(dev())
    Et, e2 = ops.rvq_pack(Ed)
    quant = torch.empty((T, d), dtype=F32, device=dev())
    idx = ops.rvq_encode(x.to(dev()), Ed, Et, e2, quant_out=quant).cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < C
    _check_indices(idx, x, list(E))
    deq = sum(E[q][idx[:, q]] for q in range(Q))                    # quantized output == sum of the selected code vectors
    assert relmax(quant, deq) <= 1e-5


def test_rvq_exact_codes_and_idempotence(ops):
    """known answers: a frame that IS a code vector selects that code with distance 0; re-encoding a dequantized frame reproduces it."""
    d, C, Q = 64, 256, 1
    E = rnd(Q, C, d, seed=6)
    Ed = E.to(dev())
    Et, e2 = ops.rvq_pack(Ed)
    pick = torch.randint(0, C, (500,), generator=torch.Generator().manual_seed(7))
    idx = ops.rvq_encode(E[0][pick].to(dev()).contiguous(), Ed, Et, e2).cpu()
    assert torch.equal(idx[:, 0], pick)


def _load_fixture():
    return torch.load(os.path.join(GOLDEN_DIR, 'soundstream_small.pt'), weights_only=False)


def test_soundstream_matches_reference_golden():
    """tests/golden/soundstream_small.pt: the REAL reference SoundStream (encoder = reference code; RVQ = restated module) on seeded
    weights / audio.  Our module loads the same state_dict by NAME and must reproduce the encoder output and every code index."""
    import audiolm_pytorch_amd as A
    fx = _load_fixture()
    ss = A.SoundStream(**fx['ctor'])
    missing, unexpected = ss.load_state_dict(synth_state_dict(fx['shapes'], fx['seed']), strict=False)
    assert not unexpected and all(k.startswith('decoder.') for k in missing), (missing, unexpected)     # this fixture carries encoder.* + rq.*
    ss.to(dev())
    wave = fx['inputs']['wave'].to(dev())
    x, _ = ss.process_input(wave)
    enc = ss.encode(x)                                              # (b, n, c)
    ref_enc = fx['outputs']['encoder_out'].transpose(1, 2)          # reference encoder output is (b, c, n)
    assert relmax(enc, ref_enc) <= 2e-5
    codes = ss.tokenize(wave)
    assert codes.dtype == torch.int64 and torch.equal(codes.cpu(), fx['outputs']['tokenize'])            # (g, b, n, q)
    emb, indices, _ = ss(wave, return_encoded=True)
    assert torch.equal(indices.cpu(), fx['outputs']['indices'])                                         # (b, n, g*q)
    assert relmax(emb, fx['outputs']['quantized']) <= 2e-5


def test_soundstream_config5_shape_properties():
    """BASELINE configs[4] codec shape (codebook 4096, 8 quantizers, strides 2*4*5*8 = 320, 512-d codes) on 1.5 s of 24 kHz audio:
    size-independent properties -- index range, dequantization consistency, per-stage minimality of the selected code against sampled
    alternatives (torch recomputation from the produced indices), and time-causality of the encoder (changing the LAST 320 samples changes only the last frame)."""
    import audiolm_pytorch_amd as A
    torch.manual_seed(0)
    ss = A.SoundStream(codebook_size=4096, rq_num_quantizers=8, target_sample_hz=24000, strides=(2, 4, 5, 8), use_local_attn=False)
    g = torch.Generator().manual_seed(3)
    for r in ss.rq.rvqs:
        for q, l in enumerate(r.layers):
            l._codebook.embed.copy_(torch.randn(1, 4096, 512, generator=g) * (0.5 ** q))
            l._codebook.initted.fill_(True)
    ss.to(dev())
    wave = (torch.randn(2, 36000, generator=g) * 0.1).to(dev())
    emb, idx, _ = ss(wave, return_encoded=True)
    assert idx.shape == (2, 36000 // 320, 8) and int(idx.min()) >= 0 and int(idx.max()) < 4096
    E = torch.stack([l._codebook.embed[0] for l in ss.rq.rvqs[0].layers])
    feats = ss.encode(ss.process_input(wave)[0])
    res = feats.reshape(-1, 512).clone()
    gen = torch.Generator().manual_seed(9)
    for q in range(8):
        chosen = (res - E[q][idx.reshape(-1, 8)[:, q]]).norm(dim=-1)
        for _ in range(4):                                          # minimality: no sampled alternative code is closer
            alt = torch.randint(0, 4096, (res.shape[0],), generator=gen).to(dev())
            other = (res - E[q][alt]).norm(dim=-1)
            assert bool((chosen <= other * (1 + 1e-5)).all()), f'stage {q}: a sampled code is closer than the selected one'
        res = res - E[q][idx.reshape(-1, 8)[:, q]]
    assert relmax(emb.reshape(-1, 512), feats.reshape(-1, 512) - res) <= 1e-4
    wave2 = wave.clone()
    wave2[:, -320:] += 0.05
    idx2 = ss.tokenize(wave2)[0]
    assert torch.equal(idx2[:, :-1], idx[:, :-1]), 'encoder is not causal'


def test_end_to_end_raw_wave_to_coarse_loss_vs_oracle():
    """BASELINE configs[4] at reduced size: raw audio -> SoundStream.tokenize (HIP conv encoder + RVQ) -> CoarseTransformerWrapper loss,
    against the CPU oracle chain (oracle encoder + restated RVQ + oracle transformer) on the same weights.  The code indices are integers:
    they must be identical, after which the loss comparison is the usual bf16-vs-fp32 one."""
    import audiolm_pytorch_amd as A
    fx = _load_fixture()
    c = fx['ctor']
    ss = A.SoundStream(**c)
    sd_codec = synth_state_dict(fx['shapes'], fx['seed'])
    ss.load_state_dict(sd_codec, strict=False)
    ss.to(dev())
    torch.manual_seed(3)
    ctor = dict(dim=128, depth=2, num_semantic_tokens=50, codebook_size=c['codebook_size'], num_coarse_quantizers=3, flash_attn=True)
    model = A.CoarseTransformer(**ctor)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev())
    w = A.CoarseTransformerWrapper(transformer=model, codec=ss, unique_consecutive=False, mask_prob=0.)
    w.train()
    g = torch.Generator().manual_seed(21)
    wave = torch.randn(2, 320 * 24, generator=g) * 0.3
    sem = torch.randint(0, 50, (2, 30), generator=g)
    loss = w(semantic_token_ids=sem.to(dev()), raw_wave=wave.to(dev()), return_loss=True)
    loss.backward()
    idx = O.soundstream_tokenize(sd_codec, wave, strides=c['strides'], num_quantizers=c['rq_num_quantizers'])     # (b, T, q)
    ours_idx = ss(wave.to(dev()), return_encoded=True)[1].cpu()
    assert torch.equal(ours_idx, idx)
    cfg = O.Cfg(dim=128, depth=2, streams=4, num_semantic_tokens=50, codebook_size=c['codebook_size'], num_coarse_quantizers=3)
    ref = O.coarse_wrapper_loss(sd, cfg, sem, idx[..., :3], training=True, unique_consecutive=False)
    assert abs(float(loss) - float(ref)) <= 2e-3 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)


@pytest.mark.parametrize('B,Cin,Cout,n,stride', [(2, 8, 4, 19, 2), (1, 64, 32, 40, 4), (2, 16, 8, 33, 5), (1, 512, 256, 25, 8), (1, 3, 5, 7, 3)])
def test_causal_conv_transpose1d(ops, B, Cin, Cout, n, stride):
    """CausalConvTranspose1d (soundstream.py:347-360) as the k = 2 zero-padded conv over phase-major channels + interleave, vs
    F.conv_transpose1d cut to n * stride (exact-fp32 MFMA: 2e-5 rel-max)."""
    from audiolm_pytorch_amd import soundstream as SS
    torch.manual_seed(stride)
    m = SS.CausalConvTranspose1d(Cin, Cout, 2 * stride, stride).to(dev())
    x = rnd(B, Cin, n, seed=70 + stride).to(dev())
    got = m(x)
    ref = F.conv_transpose1d(x, m.conv.weight, m.conv.bias, stride=stride)[..., :n * stride]
    assert got.shape == ref.shape
    assert relmax(got, ref) <= 2e-5


def test_rvq_decode(ops):
    g = torch.Generator().manual_seed(3)
    Q, C, d, T = 5, 37, 24, 300
    E = torch.randn(Q, C, d, generator=g).to(dev())
    idx = torch.randint(0, C, (T, Q), generator=g)
    idx[5, 2:] = -1
    idx[17] = -1
    out = torch.full((T, d + 3), float('nan'), device=dev())
    ops.rvq_decode(idx.to(dev()), E, out[:, :d])
    ref = O.rvq_decode(idx[None], [E[q].cpu() for q in range(Q)])[0]
    assert relmax(out[:, :d].cpu(), ref) <= 1e-6 and bool(torch.isnan(out[:, d:]).all())


def test_soundstream_decode_matches_reference_golden():
    """tests/golden/soundstream_decode_small.pt: decode_from_codebook_indices of the REAL reference decoder (incl. dropped -1 codes)."""
    import audiolm_pytorch_amd as A
    fx = torch.load(os.path.join(GOLDEN_DIR, 'soundstream_decode_small.pt'), weights_only=False)
    ss = A.SoundStream(**fx['ctor'])
    missing, unexpected = ss.load_state_dict(synth_state_dict(fx['shapes'], fx['seed']), strict=False)
    assert not unexpected and all(k.startswith('encoder.') for k in missing), (missing, unexpected)
    ss.to(dev())
    wave = ss.decode_from_codebook_indices(fx['inputs']['indices'].to(dev()))
    assert wave.shape == fx['outputs']['wave'].shape
    assert relmax(wave, fx['outputs']['wave']) <= 2e-5


def test_tokenize_decode_round_trip_shapes_and_generate_reconstruct():
    """encode -> codes -> decode keeps the length (multiple of prod(strides)); CoarseTransformerWrapper.generate(reconstruct_wave=True) and
    FineTransformerWrapper.generate(reconstruct_wave=True) end in a waveform produced by the native decoder."""
    import audiolm_pytorch_amd as A
    torch.manual_seed(0)
    codec = A.SoundStream(codebook_size=16, rq_num_quantizers=8, channels=4, codebook_dim=16, use_local_attn=False, strides=(2, 4, 5, 8)).to(dev())
    for r in codec.rq.rvqs:
        for l in r.layers:
            l._codebook.embed.normal_()
            l._codebook.initted.fill_(True)
    wave = rnd(2, 320 * 12 + 11, seed=80, scale=0.3).to(dev())
    codes = codec.tokenize(wave)                                                   # (1, 2, 12, 8)
    rec = codec.decode_from_codebook_indices(codes)
    assert rec.shape == (2, 1, 320 * 12) and bool(torch.isfinite(rec).all())
    # forward(return_recons_only=True) (soundstream.py:857-866) is encode -> quantize -> decode in one call (the quantized sum comes from
    # the encode kernel instead of the code lookup: same values up to fp32 summation order); positional call in the reference's argument
    # order; a 1-D clip comes back as (channels, n)
    rec2 = codec(wave, return_recons_only=True)
    assert rec2.shape == rec.shape and relmax(rec2, rec) <= 1e-5, relmax(rec2, rec)
    rec3 = codec(wave, None, None, False, False, False, False, False, True)
    assert torch.equal(rec3, rec2)
    assert codec(wave[0], return_recons_only=True).shape == (1, 320 * 12)
    with pytest.raises(NotImplementedError):
        codec(wave, return_discr_loss=True)
    coarse = A.CoarseTransformer(dim=64, depth=1, heads=2, num_semantic_tokens=20, codebook_size=16, num_coarse_quantizers=3, flash_attn=True).to(dev())
    cw = A.CoarseTransformerWrapper(transformer=coarse, codec=codec, unique_consecutive=False)
    sem = torch.randint(0, 20, (2, 5), device=dev())
    out = cw.generate(semantic_token_ids=sem, max_time_steps=8, reconstruct_wave=True)      # >= 7 frames: reflect padding of the k = 7 convs
    if isinstance(out, list):
        assert all(w is None or w.dim() == 1 for w in out)
    else:
        assert out.shape == (2, 8 * 320)
    fine = A.FineTransformer(dim=64, depth=1, heads=2, codebook_size=16, num_coarse_quantizers=3, num_fine_quantizers=5, flash_attn=True).to(dev())
    fw = A.FineTransformerWrapper(transformer=fine, codec=codec)
    cids = torch.randint(0, 16, (2, 8, 3), device=dev())
    out = fw.generate(coarse_token_ids=cids, reconstruct_wave=True)
    if isinstance(out, list):
        assert all(w.dim() == 1 for w in out)
    else:
        assert out.shape == (2, 8 * 320)


# ---------------------------------------------------------------------------------------------- LocalTransformer

def _state_for(fx, module):
    sd = synth_state_dict(fx['shapes'], fx['seed'])
    own = module.state_dict()
    for k in fx.get('const_keys', ()):
        sd[k] = own[k].clone()                                      # rotary inv_freq: a constant buffer on both sides
    return sd


@pytest.mark.parametrize('B,T,dim,heads,dh,W,depth', [(2, 300, 512, 8, 64, 128, 1), (1, 129, 64, 2, 32, 64, 2), (3, 64, 48, 3, 32, 64, 1), (1, 1000, 128, 4, 64, 128, 1),
                                                       (2, 50, 32, 2, 32, 16, 1)])
def test_local_transformer_vs_restated_library(B, T, dim, heads, dh, W, depth):
    """soundstream.LocalTransformer (csrc/local_attn.hip: direct form, key j visible iff 0 <= i - j <= window; k = 1 convs for the Linear layers) vs
    the literal restatement of local-attention's bucketed implementation (oracle/local_attention_restated.py: look_around, pad value -1,
    causal / exact-window / pad masks, rotary + xpos on the window pair), fp32 on both sides: full windows, a ragged last window (autopad),
    a sequence shorter than one window, the reference-default geometry (dim 512, 8 heads x 64, window 128)."""
    import audiolm_pytorch_amd  # noqa: F401
    from audiolm_pytorch_amd import soundstream as SS
    import local_attention_restated as LR

    class RefLocalTransformer(torch.nn.Module):                     # = the reference's own LocalTransformer.forward (soundstream.py:397-440)
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([torch.nn.ModuleList([
                LR.LocalMHA(dim=dim, heads=heads, dim_head=dh, qk_rmsnorm=True, window_size=W, use_rotary_pos_emb=True, gate_values_per_head=True,
                            use_xpos=True, prenorm=True, causal=True), LR.FeedForward(dim=dim)]) for _ in range(depth)])

        def forward(self, x):
            for attn, ff in self.layers:
                x = attn(x) + x
                x = ff(x) + x
            return x
    torch.manual_seed(B * 1000 + T)
    ref = RefLocalTransformer()
    for n, p in ref.named_parameters():                             # non-trivial norms / scales / gates
        with torch.no_grad():
            if p.dim() == 1:
                p.copy_(1.0 + 0.3 * torch.randn_like(p) if ('scale' in n or n.endswith('weight')) else 0.2 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * p.shape[1] ** -0.5)
    ours = SS.LocalTransformer(dim=dim, depth=depth, heads=heads, window_size=W, dim_head=dh, prenorm=True, causal=True)
    missing, unexpected = ours.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)    # same parameter / buffer names as the library modules
    ours.to(dev())
    x = rnd(B, T, dim, seed=T)
    with torch.no_grad():
        want = ref(x)
        want_mha = ref.layers[0][0](x)
    got = ours(x.to(dev()))
    assert relmax(got, want) <= 3e-5, relmax(got, want)
    assert relmax(ours.layers[0][0](x.to(dev())), want_mha) <= 3e-5


def test_soundstream_default_ctor_local_attention_golden():
    """tests/golden/soundstream_local_attn_small.pt: the REAL reference SoundStream with its default use_local_attn=True (soundstream.py:545, 613,
    830-833, 705-706) around the restated local-attention modules: encoder + encoder_attn output, every code index, and the decoded wave."""
    import audiolm_pytorch_amd as A
    fx = torch.load(os.path.join(GOLDEN_DIR, 'soundstream_local_attn_small.pt'), weights_only=False)
    ss = A.SoundStream(**fx['ctor'])
    assert ss.encoder_attn is not None and ss.decoder_attn is not None
    assert {k: tuple(v.shape) for k, v in ss.state_dict().items()} == {k: tuple(v) for k, v in fx['shapes'].items()}
    ss.load_state_dict(_state_for(fx, ss))
    ss.to(dev())
    wave = fx['inputs']['wave'].to(dev())
    x, _ = ss.process_input(wave)
    assert relmax(ss.encode(x), fx['outputs']['encoder_attn_out']) <= 3e-5
    enc = fx['outputs']['encoder_out'].to(dev())
    assert relmax(ss.encoder_attn(enc), fx['outputs']['encoder_attn_out']) <= 3e-5
    assert relmax(ss.encoder_attn.layers[0][0](enc), fx['outputs']['mha0_out']) <= 3e-5
    emb, indices, _ = ss(wave, return_encoded=True)
    b, n, q = indices.shape
    cbs = [ss.rq.rvqs[0].layers[i]._codebook.embed[0].detach().cpu() for i in range(q)]
    _check_indices(indices.reshape(b * n, q).cpu(), fx['outputs']['encoder_attn_out'].reshape(b * n, -1), cbs)
    same = (indices.cpu() == fx['outputs']['indices']).all(dim=-1).float().mean()
    assert float(same) >= 0.99, float(same)
    assert torch.equal(ss.tokenize(wave).cpu()[0], indices.cpu())
    recon = ss.decode_from_codebook_indices(fx['outputs']['indices'].to(dev()))
    assert recon.shape == fx['outputs']['recon'].shape and relmax(recon, fx['outputs']['recon']) <= 5e-5
