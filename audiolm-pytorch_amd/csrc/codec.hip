// SoundStream tokenize path for gfx950: causal 1-D convolution encoder + grouped residual vector quantisation (encode only).
//
// Replaces, for `SoundStream.tokenize()` / `forward(return_encoded=True)` (reference soundstream.py:779-852):
//   * CausalConv1d / ResidualUnit / EncoderBlock / encoder stack (soundstream.py:332-345, 362-380, 519-531): left REFLECT pad of
//     dilation*(k-1) + (1-stride), Conv1d with bias, ELU, residual add;
//   * the eval-mode forward of the third-party GroupedResidualVQ the reference instantiates at soundstream.py:592-607 (restated in
//     oracle/rvq_restated.py): per quantizer  idx = argmin_e sqrt(clamp(|r|^2 + |e|^2 - 2 r.e, 0)),  r -= E[idx].
//
// Precision: the code indices are integer outputs of an argmin over float distances, so both kernels compute in EXACT fp32 on the
// matrix core -- v_mfma_f32_32x32x2_f32 is bitwise an fp32 fma chain (157 TFLOP/s peak, 1/16 of bf16) -- never in bf16.
//
//   conv1d : implicit GEMM  out[co][t] = sum_{tap, ci} Wp[tap][ci][co] * x[ci][t*stride + tap*dil - pad]; a wave owns a
//            (32*NA co) x 64 t tile; the time axis is the MFMA column / lane axis, so activation loads and stores are coalesced
//            along t and the reflect padding is an index map on the load; bias + ELU + residual add fused in the epilogue.
//            Weights are pre-packed once as [tap][ci][co] (co contiguous = lane axis of the A operand).
//   rvq    : one workgroup = 32 frames for ALL quantizers of a group; the residual tile lives in LDS ([32][d+1] fp32, conflict-free
//            column reads), each of the 4 waves scans a quarter of the codebook in 32-code MFMA blocks (codebook pre-packed
//            transposed [d][C]), keeps a running (distance, index) minimum per frame with first-index tie-breaking, the waves'
//            candidates are merged through LDS, the winning code vector is subtracted in place and the next quantizer starts --
//            no intermediate tensor ever reaches HBM.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

// ELU(alpha = 1): v > 0 ? v : exp(v) - 1, branch-free.  Near zero exp(v) - 1 cancels, so |v| < 0.35 uses the degree-8 Taylor polynomial
// (truncation < 1e-9 relative); elsewhere exp2 (<= 2 ulp) minus one loses < 2 bits.  Agrees with expm1f to a few ulp.
__device__ __forceinline__ float elu1(float v) {
    const float em = __builtin_amdgcn_exp2f(v * 1.4426950408889634f) - 1.f;
    float p = 2.48015873e-5f;                                                   // 1/8!
    p = fmaf(p, v, 1.98412698e-4f);
    p = fmaf(p, v, 1.38888889e-3f);
    p = fmaf(p, v, 8.33333333e-3f);
    p = fmaf(p, v, 4.16666667e-2f);
    p = fmaf(p, v, 1.66666667e-1f);
    p = fmaf(p, v, 0.5f);
    p = fmaf(p, v, 1.f);
    p *= v;
    const float neg = v > -0.35f ? p : em;
    return v > 0.f ? v : neg;
}

struct ConvArgs {
    const float* x; const float* wp; const float* bias; const float* residual; float* out;
    int B, Cin, CinP, Cout, CoutP, Tin, Tout, ks, stride, dil, pad, elu;
    int zero_pad;       // 0: reflect left pad (CausalConv1d, soundstream.py:343); 1: zeros left of the signal (the k = 2 form of a transposed conv)
};

// grid: (ceil(Tout / 256), CoutP / (32 * NA), B); 4 waves along time, 64 output steps each
template <int NA>
__global__ __launch_bounds__(256, 2) void conv1d_causal_kernel(ConvArgs a) {      // (256, 2): without the occupancy hint the compiler parks copies in AGPRs (184 registers, NA = 2)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * 32 * NA;
    const int t0 = blockIdx.x * 256 + wave * 64;
    if (t0 >= a.Tout) return;
    const float* xb = a.x + (long long)b * a.Cin * a.Tin;

    f32x16 acc[NA][2];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int tin[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) tin[j] = (t0 + j * 32 + lr) * a.stride - a.pad;

    // The (tap, channel-pair) contraction steps are flattened into one sequence and software-pipelined by hand: two register groups of U
    // steps, the loads of group g + 1 are issued before the MFMAs of group g.  All addressing is 32-bit buffer addressing: a per-lane byte
    // offset that only changes with the tap (reflect-padded time index) plus a wave-uniform scalar offset per step -- no 64-bit VALU
    // multiplies in the loop (they, not the loads, were what limited the first version).  Channels >= Cin meet zero-padded weights and
    // read 0 past the end of this batch element's buffer; time steps >= Tout compute garbage that is never stored.
    constexpr int U = 2;
    const int nk = a.CinP >> 1;                                    // channel pairs per tap
    const int nsteps = a.ks * nk, ng = (nsteps + U - 1) / U;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, a.Cin * a.Tin * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.ks * a.CinP * a.CoutP * 4, 0x00020000);
    const int wvo = (lh * a.CoutP + co0 + lr) * 4;
    int ltap = 0, lk = 0;                                          // next step to load (wave-uniform)
    int xvo[2];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int p = tin[j] + tap * a.dil;
            const bool left = p < 0;
            p = left ? -p : p;                                     // reflect (F.pad mode='reflect'): index -i -> i
            xvo[j] = (left && a.zero_pad) ? (int)0x80000000 : (lh * a.Tin + p) * 4;      // zero pad: out-of-range offset reads 0
        }
    };
    set_tap(0);
    float av[2][U][NA], bv[2][U][2];
    auto load_group = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ltap < a.ks) {
                const int wso = ((ltap * a.CinP + 2 * lk) * a.CoutP) * 4, xso = (2 * lk * a.Tin) * 4;
#pragma unroll
                for (int i = 0; i < NA; ++i) av[BUF][u][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, wvo + i * 128, wso, 0));
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[BUF][u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, xvo[j], xso, 0));
                if (++lk == nk) {
                    lk = 0;
                    ++ltap;
                    set_tap(ltap);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) av[BUF][u][i] = 0.f;
                bv[BUF][u][0] = bv[BUF][u][1] = 0.f;
            }
        }
    };
    auto mfma_group = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[BUF][u][i], bv[BUF][u][j], acc[i][j], 0, 0, 0);
    };
    load_group(std::integral_constant<int, 0>{});
    for (int g = 0; g < ng; g += 2) {
        if (g + 1 < ng) load_group(std::integral_constant<int, 1>{});
        mfma_group(std::integral_constant<int, 0>{});
        if (g + 2 < ng) load_group(std::integral_constant<int, 0>{});
        if (g + 1 < ng) mfma_group(std::integral_constant<int, 1>{});
    }
    // D layout: column = lane & 31 (time), row = (r & 3) + 8 * (r >> 2) + 4 * lh (output channel).  Epilogue addressing is 32-bit buffer
    // addressing too: per-lane byte offset (channel half, time) + a wave-uniform row offset; rows >= Cout fall outside the buffer and are
    // dropped by the bounds check, lanes with t >= Tout get an out-of-range offset.
    const long long ob = (long long)b * a.Cout * a.Tout;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + ob, 0, a.Cout * a.Tout * 4, 0x00020000);
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.residual ? a.residual + ob : a.out + ob), 0, a.Cout * a.Tout * 4, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.Cout * 4, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float bias[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(rsB, (co0 + i * 32 + 8 * g + 4 * lh) * 4, 0, 0);      // channels >= Cout read 0
#pragma unroll
            for (int c = 0; c < 4; ++c) bias[4 * g + c] = __uint_as_float(bq[c]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + j * 32 + lr;
            const int vb = t < a.Tout ? ((co0 + i * 32 + 4 * lh) * a.Tout + t) * 4 : (int)0x80000000;
            float v[16], rr[16];
            if (a.residual) {                                      // all 16 row loads in flight before the first use (they were issued and waited for one by one)
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsR, vb, ((r & 3) + 8 * (r >> 2)) * a.Tout * 4, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[i][j][r] + bias[r];
                if (a.elu) v[r] = elu1(v[r]);
            }
            if (a.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rr[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rsO, vb, ((r & 3) + 8 * (r >> 2)) * a.Tout * 4, 0);
        }
    }
}

// ---- strided down-sampling conv of an EncoderBlock (reference soundstream.py:379: CausalConv1d(k = 2 s, stride s), reflect left pad = s) ---------------------
// Round 6.  In conv1d_causal_kernel a lane's activation loads for consecutive output steps are s samples apart (2-8): one dword per lane and tap, 2-8 cache
// lines per wave load -- the four down-sampling convs ran at 0.28 of the fp32-MFMA peak, the stride-1 convs at 0.6-0.7.  Here out[t] reads the 2 s CONTIGUOUS
// samples x[(t - 1) s .. (t + 1) s) of every input channel: a lane fetches them with 16-byte loads (all taps of a channel at once, consecutive lanes
// contiguous), and the contraction runs channel-pair-major (taps inner) instead of tap-major -- still one exact-fp32 fma chain per output, in a fixed order.
//   the reflect pad only concerns output step 0 (samples -s .. -1 mirror s .. 1): that one lane patches its first s values with scalar loads.
template <int NA, int S>
__global__ __launch_bounds__(256, 2) void conv1d_strided_kernel(ConvArgs a) {
    constexpr int K = 2 * S, NV = (K + 3) / 4;                     // taps, 16-byte loads per channel and 32-step block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * 32 * NA;
    const int t0 = blockIdx.x * 256 + wave * 64;
    if (t0 >= a.Tout) return;
    const float* xb = a.x + (long long)b * a.Cin * a.Tin;

    f32x16 acc[NA][2];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.CinP >> 1;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, a.Cin * a.Tin * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.ks * a.CinP * a.CoutP * 4, 0x00020000);
    const int wvo = (lh * a.CoutP + co0 + lr) * 4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    int xvo[2];
    bool first[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = t0 + j * 32 + lr;
        first[j] = t == 0;
        xvo[j] = (lh * a.Tin + max(t - 1, 0) * S) * 4;             // (step 0 loads x[0 .. 2 s) and patches / shifts below)
    }
    float xv[2][2][4 * NV], av[2][K][NA];
    auto load_pair = [&](auto bufc, int lk) {
        constexpr int BUF = decltype(bufc)::value;
        const int xso = (2 * lk * a.Tin) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvo[j] + q * 16, xso, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[BUF][j][4 * q + c] = __uint_as_float(v[c]);
            }
            if (first[j]) {                                         // output step 0: taps 0 .. s-1 are the mirrored samples s .. 1, taps s .. 2s-1 the samples 0 .. s-1
#pragma unroll
                for (int tap = K - 1; tap >= S; --tap) xv[BUF][j][tap] = xv[BUF][j][tap - S];
#pragma unroll
                for (int tap = 0; tap < S; ++tap) xv[BUF][j][tap] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, (lh * a.Tin + (S - tap)) * 4, xso, 0));
            }
        }
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
            const int wso = ((tap * a.CinP + 2 * lk) * a.CoutP) * 4;
#pragma unroll
            for (int i = 0; i < NA; ++i) av[BUF][tap][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, wvo + i * 128, wso, 0));
        }
    };
    auto mfma_pair = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int tap = 0; tap < K; ++tap)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[BUF][tap][i], xv[BUF][j][tap], acc[i][j], 0, 0, 0);
    };
    load_pair(std::integral_constant<int, 0>{}, 0);
    for (int lk = 0; lk < nk; lk += 2) {
        if (lk + 1 < nk) load_pair(std::integral_constant<int, 1>{}, lk + 1);
        mfma_pair(std::integral_constant<int, 0>{});
        if (lk + 2 < nk) load_pair(std::integral_constant<int, 0>{}, lk + 2);
        if (lk + 1 < nk) mfma_pair(std::integral_constant<int, 1>{});
    }
    const long long ob = (long long)b * a.Cout * a.Tout;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + ob, 0, a.Cout * a.Tout * 4, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.Cout * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float bias[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(rsB, (co0 + i * 32 + 8 * g + 4 * lh) * 4, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) bias[4 * g + c] = __uint_as_float(bq[c]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + j * 32 + lr;
            const int vb = t < a.Tout ? ((co0 + i * 32 + 4 * lh) * a.Tout + t) * 4 : (int)0x80000000;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bias[r];
                if (a.elu) v = elu1(v);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vb, ((r & 3) + 8 * (r >> 2)) * a.Tout * 4, 0);
            }
        }
    }
}

// ---- fused ResidualUnit (round 6; reference soundstream.py:362-369: x + ELU(conv_k1(ELU(conv_k7,dil(x))))): ONE launch per unit instead of two, the
// C x (time tile) intermediate never leaves the registers.  Unfused, every unit of the first two encoder stages wrote and re-read a 737 MB fp32
// activation (8 x 30 s at 24 kHz) around an HBM-bound k = 1 conv.
//   * a wave owns ALL C channels of 32 * NJ time steps: the k7 conv accumulators (NA = C / 32 row blocks) are the k1 conv's B operands IN PLACE.  The
//     MFMA's B operand wants lane half lh <-> contraction index 2 kk + lh; its D layout gives lane half lh the rows {0-3, 8-11, ...} + 4 lh.  So the k7
//     conv is computed with its OUTPUT CHANNELS PERMUTED: MFMA row 8 a + 4 h + b of block i holds channel 32 i + 2 (4 a + b) + h, i.e. accumulator
//     register r of lane half lh holds channel 32 i + 2 r + lh -- exactly the k1 conv's step kk = 16 i + r, no data movement.  (A permutation of output
//     rows is a permutation of the weight rows fed as the A operand: every output element is still the same fma chain in the same (tap, channel pair)
//     order as alm_conv1d_causal computes, and the k1 contraction runs over the same channel pairs in the same order -- the fused unit is BITWISE equal to
//     the two launches it replaces; tests/test_gpu_codec.py asserts it.)
//   * bias + ELU of the k7 conv in registers, k1 conv straight from them (weights [ci][co] from L2, double-buffered 16 steps ahead), bias + ELU + residual
//     x + store in the usual D-layout epilogue.
struct ResUnitArgs {
    const float* x; const float* w7; const float* b7; const float* w1; const float* b1; float* out;
    int B, C, T, ks, dil;
};

template <int NA, int NJ>
__global__ __launch_bounds__(256, 2) void resunit_kernel(ResUnitArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z;
    const int t0 = (blockIdx.x * 4 + wave) * 32 * NJ;
    if (t0 >= a.T) return;
    const int C = a.C;
    const float* xb = a.x + (long long)b * C * a.T;
    const int pad = a.dil * (a.ks - 1);

    f32x16 h[NA][NJ];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[i][j][r] = 0.f;

    int tin[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) tin[j] = t0 + j * 32 + lr - pad;

    // ---- k7 dilated conv (the loop of conv1d_causal_kernel, all C output channels, permuted rows)
    constexpr int U = 2;
    const int nk = C >> 1;
    const int nsteps = a.ks * nk, ng = (nsteps + U - 1) / U;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, C * a.T * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w7), 0, a.ks * C * C * 4, 0x00020000);
    const int prow = 2 * (4 * (lr >> 3) + (lr & 3)) + ((lr >> 2) & 1);        // channel (within a 32-block) that MFMA row lr holds
    const int wvo = (lh * C + prow) * 4;
    int ltap = 0, lk = 0;
    int xvo[NJ];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int p = tin[j] + tap * a.dil;
            p = p < 0 ? -p : p;                                      // reflect left pad
            xvo[j] = (lh * a.T + p) * 4;
        }
    };
    set_tap(0);
    float av[2][U][NA], bv[2][U][NJ];
    auto load_group = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ltap < a.ks) {
                const int wso = ((ltap * C + 2 * lk) * C) * 4, xso = (2 * lk * a.T) * 4;
#pragma unroll
                for (int i = 0; i < NA; ++i) av[BUF][u][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, wvo + i * 128, wso, 0));
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[BUF][u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, xvo[j], xso, 0));
                if (++lk == nk) {
                    lk = 0;
                    ++ltap;
                    set_tap(ltap);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) av[BUF][u][i] = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[BUF][u][j] = 0.f;
            }
        }
    };
    auto mfma_group = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) h[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[BUF][u][i], bv[BUF][u][j], h[i][j], 0, 0, 0);
    };
    load_group(std::integral_constant<int, 0>{});
    for (int g = 0; g < ng; g += 2) {
        if (g + 1 < ng) load_group(std::integral_constant<int, 1>{});
        mfma_group(std::integral_constant<int, 0>{});
        if (g + 2 < ng) load_group(std::integral_constant<int, 0>{});
        if (g + 1 < ng) mfma_group(std::integral_constant<int, 1>{});
    }

    // ---- bias + ELU of the k7 conv: register r of lane half lh = channel 32 i + 2 r + lh
    {
        const auto rsB7 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b7), 0, C * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bb = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB7, (32 * i + 2 * r + lh) * 4, 0, 0));
#pragma unroll
                for (int j = 0; j < NJ; ++j) h[i][j][r] = elu1(h[i][j][r] + bb);
            }
    }

    // ---- k1 conv out of the registers + epilogue, one 32-channel output block at a time
    const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1), 0, C * C * 4, 0x00020000);
    const long long ob = (long long)b * C * a.T;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + ob, 0, C * a.T * 4, 0x00020000);
    const auto rsB1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b1), 0, C * 4, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int w1vo = (lh * C + lr) * 4;                              // A operand: W1p[ci = 2 kk + lh][co = 32 io + lr]
    float a1[2][16];
    auto load_a1 = [&](auto bufc, int io, int ii) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[BUF][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW1, w1vo + io * 128, (2 * (16 * ii + r) * C) * 4, 0));
    };
#pragma unroll 1
    for (int io = 0; io < NA; ++io) {
        f32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        load_a1(std::integral_constant<int, 0>{}, io, 0);
        // the residual x and the bias of this output block are fetched NOW, under the k1 MFMAs (fetched in the epilogue, each of the 16 row loads was
        // followed by its own wait: 16 serial L2 round trips per 32 x 32 block -- seen in the ISA, about half of the unit's time)
        int vb[NJ];
        float xr[NJ][16];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = t0 + j * 32 + lr;
            vb[j] = t < a.T ? ((io * 32 + 4 * lh) * a.T + t) * 4 : (int)0x80000000;
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, vb[j], ((r & 3) + 8 * (r >> 2)) * a.T * 4, 0));
        }
        float bias[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(rsB1, (io * 32 + 8 * g + 4 * lh) * 4, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) bias[4 * g + c] = __uint_as_float(bq[c]);
        }
#pragma unroll
        for (int ii = 0; ii < NA; ++ii) {
            if (ii & 1) {
                if (ii + 1 < NA) load_a1(std::integral_constant<int, 0>{}, io, ii + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][r], h[ii][j][r], acc[j], 0, 0, 0);
            } else {
                if (ii + 1 < NA) load_a1(std::integral_constant<int, 1>{}, io, ii + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][r], h[ii][j][r], acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = elu1(acc[j][r] + bias[r]) + xr[j][r];
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rsO, vb[j], ((r & 3) + 8 * (r >> 2)) * a.T * 4, 0);
        }
    }
}

// weight [Cout][Cin][ks] -> packed [ks][CinP][CoutP] (zero padded)
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int ks, int CinP,
                                                        int CoutP) {
    const long long total = (long long)ks * CinP * CoutP;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % CoutP);
        const int ci = (int)((i / CoutP) % CinP);
        const int tap = (int)(i / ((long long)CoutP * CinP));
        wp[i] = (co < Cout && ci < Cin) ? w[((long long)co * Cin + ci) * ks + tap] : 0.f;
    }
}

// codebook [C][d] -> MFMA-ordered image P[j][lh][code][4] = E[code][8 j + 4 lh + 0..3] (zero padded to dP = 8 ceil(d / 8) columns and CP codes):
// lane (code, lh) of the distance GEMM reads ONE float4 per 4 MFMA steps, 512 contiguous bytes per half wave.  + squared norms [CP]
// (+inf for the pad codes: never selected)
__global__ __launch_bounds__(256) void rvq_pack_kernel(const float* __restrict__ E, float* __restrict__ Et, float* __restrict__ e2, int C, int d,
                                                       int CP) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= CP) return;
    const int dP = (d + 7) & ~7;
    float s = 0.f;
    for (int k = 0; k < dP; ++k) {
        const float v = (c < C && k < d) ? E[(long long)c * d + k] : 0.f;
        Et[(((long long)(k >> 3) * 2 + ((k >> 2) & 1)) * CP + c) * 4 + (k & 3)] = v;
        s += v * v;
    }
    e2[c] = c < C ? s : INFINITY;
}

struct RvqArgs {
    const float* x; long long ldx;            // [T][ldx] frames (this group's d columns start at x)
    const float* E;                           // [Q][C][d]   original codebooks (residual update)
    const float* Et;                          // [Q][dP/8][2][CP][4]  MFMA-ordered image (rvq_pack_kernel)
    const float* e2;                          // [Q][CP]
    long long* idx; long long ldi;            // [T][ldi] output indices (this group's Q columns start at idx)
    float* quant; long long ldq;              // optional: sum of the selected code vectors [T][ldq] (the `quantized` output), or NULL
    int T, d, C, CP, Q;
};

// NW = waves per workgroup (4 or 8), each scanning 1 / NW of the codebook.  Round 6: 18 000 frames are 563 tiles of 32; with 4 waves (two workgroups per CU,
// 512 resident) that is ONE full round plus a 10 % tail that takes a second full round -- the launch ran at 0.48 of the fp32-MFMA peak; with 8 waves a tile
// takes half as long and three rounds of 256 cost 1.5 instead of 2 tile-times (alm_rvq_encode picks NW by that round count).
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void rvq_encode_kernel(RvqArgs a) {
    extern __shared__ float sm[];
    const int dP = (a.d + 7) & ~7;
    const int ld = dP + 4;                    // 16-B aligned rows (ds_read_b128), consecutive rows 4 banks apart
    float* res = sm;                          // [32][dP + 4]
    float* x2 = res + 32 * ld;                // [32]
    float* candd = x2 + 32;                   // [NW][32]
    int* candi = reinterpret_cast<int*>(candd + 32 * NW);   // [NW][32]
    int* win = candi + 32 * NW;               // [32]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int f0 = blockIdx.x * 32;

    // residual tile <- x (pad columns zero) ; |x|^2 per frame (thread = (frame t / NP, column phase t % NP), NP = 2 NW)
    constexpr int NP = 2 * NW;
    const int fj = t / NP, ph = t % NP;
    {
        float s = 0.f;
        const bool ok = f0 + fj < a.T;
        for (int e = ph; e < dP; e += NP) {
            const float v = (ok && e < a.d) ? a.x[(long long)(f0 + fj) * a.ldx + e] : 0.f;
            res[fj * ld + e] = v;
            s += v * v;
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (NP == 16) s += __shfl_xor(s, 8, 64);
        if (ph == 0) x2[fj] = s;
    }
    __syncthreads();

    constexpr int PF = 4;                     // 8-column steps per register group; two groups in flight (software pipeline)
    const int nj = dP >> 3, ng = (nj + PF - 1) / PF;
    const int nblk = a.CP / 32;               // 32-code blocks; wave w scans block pairs 2w, 2w + 2 NW, ...
    const float4* bp = reinterpret_cast<const float4*>(res + lr * ld + 4 * lh);          // + 2 j  (float4 units): residual[frame][8 j + 4 lh ..]
    for (int q = 0; q < a.Q; ++q) {
        const float4* Pq = reinterpret_cast<const float4*>(a.Et + (long long)q * dP * a.CP);
        const float* e2 = a.e2 + (long long)q * a.CP;
        const float xn = x2[lr];
        float best = INFINITY;
        int besti = 0x7fffffff;
        // two 32-code blocks per pass share every residual (B) operand read
        for (int cb = wave * 2; cb < nblk; cb += 2 * NW) {
            const bool two = cb + 1 < nblk;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const float4* ap = Pq + (long long)lh * a.CP + cb * 32 + lr;                  // + 2 j CP per step
            const int o1 = two ? 32 : 0;
            float4 A0[2][PF], A1[2][PF], Bv[2][PF];
            auto load_group = [&](auto bufc, int g) {
                constexpr int BUF = decltype(bufc)::value;
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = g * PF + u;
                    if (j < nj) {
                        const float4* pj = ap + (long long)j * 2 * a.CP;
                        A0[BUF][u] = pj[0];
                        A1[BUF][u] = pj[o1];
                        Bv[BUF][u] = bp[2 * j];
                    } else {
                        A0[BUF][u] = A1[BUF][u] = Bv[BUF][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            };
            auto mfma_group = [&](auto bufc) {
                constexpr int BUF = decltype(bufc)::value;
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const float a0v[4] = {A0[BUF][u].x, A0[BUF][u].y, A0[BUF][u].z, A0[BUF][u].w};
                    const float a1v[4] = {A1[BUF][u].x, A1[BUF][u].y, A1[BUF][u].z, A1[BUF][u].w};
                    const float bvv[4] = {Bv[BUF][u].x, Bv[BUF][u].y, Bv[BUF][u].z, Bv[BUF][u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[i], bvv[i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[i], bvv[i], acc1, 0, 0, 0);
                    }
                }
            };
            load_group(std::integral_constant<int, 0>{}, 0);
            for (int g = 0; g < ng; g += 2) {
                if (g + 1 < ng) load_group(std::integral_constant<int, 1>{}, g + 1);
                mfma_group(std::integral_constant<int, 0>{});
                if (g + 2 < ng) load_group(std::integral_constant<int, 0>{}, g + 2);
                if (g + 1 < ng) mfma_group(std::integral_constant<int, 1>{});
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                if (h2 == 1 && !two) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) {                                       // codes in increasing order per lane
                    const int code = (cb + h2) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float d2 = (xn + e2[code]) + (-2.f * (h2 ? acc1[r] : acc0[r]));
                    const float dist = sqrtf(fmaxf(d2, 0.f));
                    if (dist < best) { best = dist; besti = code; }
                }
            }
        }
        {   // merge the two lane halves (same frame, different codes), then the 4 waves: smallest distance, first index on ties
            const float ob = __shfl_xor(best, 32, 64);
            const int oi = __shfl_xor(besti, 32, 64);
            if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
            if (lane < 32) { candd[wave * 32 + lane] = best; candi[wave * 32 + lane] = besti; }
        }
        __syncthreads();
        if (t < 32) {
            float bd = candd[t];
            int bi = candi[t];
            for (int w = 1; w < NW; ++w) {
                const float od = candd[w * 32 + t];
                const int oi = candi[w * 32 + t];
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            win[t] = bi;
            if (f0 + t < a.T) a.idx[(long long)(f0 + t) * a.ldi + q] = bi;
        }
        __syncthreads();
        {   // residual -= E[q][win]; new |r|^2
            const int code = win[fj];
            const float* er = a.E + ((long long)q * a.C + code) * a.d;
            float s = 0.f;
            for (int e = ph; e < a.d; e += NP) {
                const float v = res[fj * ld + e] - er[e];
                res[fj * ld + e] = v;
                s += v * v;
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (NP == 16) s += __shfl_xor(s, 8, 64);
            if (ph == 0) x2[fj] = s;
        }
        __syncthreads();
    }
    if (a.quant) {                            // quantized = x - final residual
        const bool ok = f0 + fj < a.T;
        if (ok)
            for (int e = ph; e < a.d; e += NP) a.quant[(long long)(f0 + fj) * a.ldq + e] = a.x[(long long)(f0 + fj) * a.ldx + e] - res[fj * ld + e];
    }
}

// Transposed conv as a k = 2 causal conv: y[b][r * Cout + co][q] (r = output phase, s phases) -> out[b][co][q * s + r]
__global__ __launch_bounds__(256) void phase_interleave_kernel(const float* __restrict__ y, float* __restrict__ out, int Cout, int s, int n) {
    const long long To = (long long)n * s;
    const long long total = (long long)Cout * To;
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i / To);
        const long long t = i % To;
        const int q = (int)(t / s), r = (int)(t % s);
        out[((long long)b * Cout + co) * To + t] = y[(((long long)b * s + r) * Cout + co) * n + q];
    }
}

// code lookup of a residual VQ: out[t][0..d) = sum_q E[q][idx[t][q]][0..d)   (idx < 0 selects nothing)
__global__ __launch_bounds__(256) void rvq_decode_kernel(const long long* __restrict__ idx, long long ldi, const float* __restrict__ E,
                                                         float* __restrict__ out, long long ldo, int T, int d, int C, int Q) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= T) return;
    for (int e = lane; e < d; e += 64) {
        float acc = 0.f;
        for (int q = 0; q < Q; ++q) {
            const long long c = idx[(long long)t * ldi + q];
            if (c >= 0 && c < C) acc += E[((long long)q * C + c) * d + e];
        }
        out[(long long)t * ldo + e] = acc;
    }
}

// [B][C][T] -> [B][T][C]  ('b c n -> b n c', soundstream.py:823)
__global__ __launch_bounds__(256) void bct_to_btc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, tt = t0 + tx;
        tile[i][tx] = (c < C && tt < T) ? in[((long long)b * C + c) * T + tt] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int tt = t0 + i, c = c0 + tx;
        if (tt < T && c < C) out[((long long)b * T + tt) * C + c] = tile[tx][i];
    }
}

}  // namespace

extern "C" int alm_conv1d_packed_floats(int Cout, int Cin, int ksize) { return ksize * ((Cin + 1) & ~1) * ((Cout + 31) & ~31); }

// w fp32 [Cout][Cin][ksize] (nn.Conv1d layout) -> wp (alm_conv1d_packed_floats floats); once per weight update
extern "C" int alm_conv1d_pack(const float* w, float* wp, int Cout, int Cin, int ksize, void* stream) {
    if (Cout <= 0 || Cin <= 0 || ksize <= 0) return ALM_ERR_BAD_ARG;
    const int CinP = (Cin + 1) & ~1, CoutP = (Cout + 31) & ~31;
    const long long total = (long long)ksize * CinP * CoutP;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, w, wp,
                       Cout, Cin, ksize, CinP, CoutP);
    ALM_LAUNCH_CHECK();
    return 0;
}

// out[b][co][t] = act(bias[co] + sum_{ci,k} w[co][ci][k] * xpad[b][ci][t*stride + k*dilation]) (+ residual[b][co][t]),
// xpad = x left-padded by dilation*(ksize-1) + 1 - stride in 'reflect' mode (soundstream.py:339-345); Tout = (Tin - stride) / stride + 1.
extern "C" int alm_conv1d_causal(const float* x, const float* wp, const float* bias, const float* residual, float* out, int B, int Cin, int Cout,
                                 int Tin, int ksize, int stride, int dilation, int elu, int zero_pad, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || ksize <= 0 || stride <= 0 || dilation <= 0) return ALM_ERR_BAD_ARG;
    const int pad = dilation * (ksize - 1) + 1 - stride;
    if (pad < 0 || pad >= Tin || Tin < stride) return ALM_ERR_UNSUPPORTED;
    if ((long long)(Cout + 32) * Tin * 4 >= 0x7fffffffLL || (long long)(Cin + 2) * Tin * 4 >= 0x7fffffffLL || (long long)ksize * (Cin + 1) * (Cout + 31) * 4 >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    const int Tout = (Tin - stride) / stride + 1;
    ConvArgs a{x, wp, bias, residual, out, B, Cin, (Cin + 1) & ~1, Cout, (Cout + 31) & ~31, Tin, Tout, ksize, stride, dilation, pad, elu, zero_pad};
    const int gx = (Tout + 255) / 256;
    // the down-sampling convs of the encoder (k = 2 s, stride s, no dilation, reflect pad, no residual): 16-byte activation loads, channel-major contraction
    static const int strided_on = [] { const char* e = getenv("ALM_CONV_STRIDED"); return e ? atoi(e) : 1; }();     // A/B switch
    if (strided_on && stride > 1 && ksize == 2 * stride && dilation == 1 && !zero_pad && !residual && a.CoutP % 64 == 0 && Tin >= 2 * stride &&
        (stride == 2 || stride == 4 || stride == 5 || stride == 8)) {
        const dim3 grid(gx, a.CoutP / 64, B);
        hipStream_t st = (hipStream_t)stream;
        if (stride == 2) hipLaunchKernelGGL((conv1d_strided_kernel<2, 2>), grid, dim3(256), 0, st, a);
        else if (stride == 4) hipLaunchKernelGGL((conv1d_strided_kernel<2, 4>), grid, dim3(256), 0, st, a);
        else if (stride == 5) hipLaunchKernelGGL((conv1d_strided_kernel<2, 5>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv1d_strided_kernel<2, 8>), grid, dim3(256), 0, st, a);
        ALM_LAUNCH_CHECK();
        return 0;
    }
    if (a.CoutP % 64 == 0)
        hipLaunchKernelGGL(conv1d_causal_kernel<2>, dim3(gx, a.CoutP / 64, B), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv1d_causal_kernel<1>, dim3(gx, a.CoutP / 32, B), dim3(256), 0, (hipStream_t)stream, a);
    ALM_LAUNCH_CHECK();
    return 0;
}

// One ResidualUnit (reference soundstream.py:362-369) in ONE launch: out = x + ELU(b1 + W1 . ELU(b7 + conv_k,dil(x))) with the reflect left pad of a
// CausalConv1d (stride 1).  w7p / w1p: alm_conv1d_pack images of the two weights ([k][C][C] and [1][C][C]).  C % 32 == 0, C <= 256 (else
// ALM_ERR_UNSUPPORTED: the caller runs the two alm_conv1d_causal launches).  Bitwise equal to those two launches.
extern "C" int alm_resunit_causal(const float* x, const float* w7p, const float* b7, const float* w1p, const float* b1, float* out, int B, int C, int T,
                                  int ksize, int dilation, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0 || ksize <= 0 || dilation <= 0) return ALM_ERR_BAD_ARG;
    if ((C & 31) || C > 256) return ALM_ERR_UNSUPPORTED;
    if (dilation * (ksize - 1) >= T) return ALM_ERR_UNSUPPORTED;
    if ((long long)(C + 32) * T * 4 >= 0x7fffffffLL || (long long)ksize * (C + 1) * (C + 31) * 4 >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    ResUnitArgs a{x, w7p, b7, w1p, b1, out, B, C, T, ksize, dilation};
    hipStream_t st = (hipStream_t)stream;
    const int na = C / 32;
    // 32-step time blocks per wave (NJ): more blocks = more MFMAs per weight (A operand) load, fewer waves.  ALM_RESUNIT_NJ=<1|2|4>: A/B override
    static const int nj_env = [] { const char* e = getenv("ALM_RESUNIT_NJ"); return e ? atoi(e) : 0; }();
    int nj = na == 1 ? 2 : na == 2 ? 2 : 1;
    if (nj_env == 1 || nj_env == 2 || (nj_env == 4 && na <= 2)) nj = nj_env;
    if (na > 4 && nj > 1) nj = 1;
    const dim3 grid((T + 128 * nj - 1) / (128 * nj), 1, B);
#define ALM_RU(NA_, NJ_) hipLaunchKernelGGL((resunit_kernel<NA_, NJ_>), grid, dim3(256), 0, st, a)
    switch (na * 8 + nj) {
        case 1 * 8 + 1: ALM_RU(1, 1); break;  case 1 * 8 + 2: ALM_RU(1, 2); break;  case 1 * 8 + 4: ALM_RU(1, 4); break;
        case 2 * 8 + 1: ALM_RU(2, 1); break;  case 2 * 8 + 2: ALM_RU(2, 2); break;  case 2 * 8 + 4: ALM_RU(2, 4); break;
        case 3 * 8 + 1: ALM_RU(3, 1); break;  case 3 * 8 + 2: ALM_RU(3, 2); break;
        case 4 * 8 + 1: ALM_RU(4, 1); break;  case 4 * 8 + 2: ALM_RU(4, 2); break;
        case 5 * 8 + 1: ALM_RU(5, 1); break;  case 6 * 8 + 1: ALM_RU(6, 1); break;  case 7 * 8 + 1: ALM_RU(7, 1); break;  case 8 * 8 + 1: ALM_RU(8, 1); break;
        default: return ALM_ERR_UNSUPPORTED;
    }
#undef ALM_RU
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_rvq_padded_codes(int C) { return (C + 31) & ~31; }

extern "C" int alm_rvq_padded_dim(int d) { return (d + 7) & ~7; }

// E fp32 [Q][C][d] -> Et [Q][alm_rvq_padded_dim(d)][CP] floats (MFMA-ordered image), e2 [Q][CP]  (CP = alm_rvq_padded_codes(C)); once per
// codebook update
extern "C" int alm_rvq_pack(const float* E, float* Et, float* e2, int Q, int C, int d, void* stream) {
    if (Q <= 0 || C <= 0 || d <= 0) return ALM_ERR_BAD_ARG;
    const int CP = alm_rvq_padded_codes(C);
    for (int q = 0; q < Q; ++q)
        hipLaunchKernelGGL(rvq_pack_kernel, dim3((CP + 255) / 256), dim3(256), 0, (hipStream_t)stream, E + (long long)q * C * d,
                           Et + (long long)q * alm_rvq_padded_dim(d) * CP, e2 + (long long)q * CP, C, d, CP);
    ALM_LAUNCH_CHECK();
    return 0;
}

// residual-VQ encode of T frames x[T][ldx] (d columns) against Q codebooks: idx[T][ldi] (int64, Q columns), optional quantized sum
extern "C" int alm_rvq_encode(const float* x, long long ldx, const float* E, const float* Et, const float* e2, long long* idx, long long ldi,
                              float* quant, long long ldq, int T, int d, int C, int Q, void* stream) {
    if (T <= 0) return 0;
    if (d <= 0 || C <= 0 || Q <= 0) return ALM_ERR_BAD_ARG;
    // waves per workgroup: the round count of the launch decides (see rvq_encode_kernel).  ALM_RVQ_WAVES=4|8: A/B override
    static const int nw_env = [] { const char* e = getenv("ALM_RVQ_WAVES"); return e ? atoi(e) : 0; }();
    const long long tiles = (T + 31) / 32;
    const double cost4 = (double)((tiles + 511) / 512) / 4.0, cost8 = (double)((tiles + 255) / 256) / 8.0;      // rounds x tile time (1 / waves)
    const int nw = nw_env == 4 || nw_env == 8 ? nw_env : (cost8 < cost4 ? 8 : 4);
    const size_t smem = (size_t)(32 * (alm_rvq_padded_dim(d) + 4) + 32 + 32 * nw) * sizeof(float) + (32 * nw + 32) * sizeof(int);
    if (smem > 160 * 1024) return ALM_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rvq_encode_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(rvq_encode_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RvqArgs a{x, ldx, E, Et, e2, idx, ldi, quant, ldq, T, d, C, alm_rvq_padded_codes(C), Q};
    if (nw == 8) hipLaunchKernelGGL(rvq_encode_kernel<8>, dim3((unsigned)tiles), dim3(512), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(rvq_encode_kernel<4>, dim3((unsigned)tiles), dim3(256), smem, (hipStream_t)stream, a);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_bct_to_btc(const float* in, float* out, int B, int C, int T, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    hipLaunchKernelGGL(bct_to_btc_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, in, out, C, T);
    ALM_LAUNCH_CHECK();
    return 0;
}

// y [B][s * Cout][n] (phase-major channels) -> out [B][Cout][n * s]: the output interleave of CausalConvTranspose1d (soundstream.py:347-360)
extern "C" int alm_phase_interleave(const float* y, float* out, int B, int Cout, int s, int n, void* stream) {
    if (B <= 0 || Cout <= 0 || s <= 0 || n <= 0) return ALM_ERR_BAD_ARG;
    const long long total = (long long)Cout * n * s;
    const int gx = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(phase_interleave_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, y, out, Cout, s, n);
    ALM_LAUNCH_CHECK();
    return 0;
}

// idx int64 [T][ldi] (Q columns, -1 = no code), E fp32 [Q][C][d] -> out fp32 [T][ldo] (d columns): GroupedResidualVQ.get_output_from_indices
// for one group (soundstream.py:697)
extern "C" int alm_rvq_decode(const long long* idx, long long ldi, const float* E, float* out, long long ldo, int T, int d, int C, int Q,
                              void* stream) {
    if (T <= 0) return 0;
    if (d <= 0 || C <= 0 || Q <= 0) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(rvq_decode_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, idx, ldi, E, out, ldo, T, d, C, Q);
    ALM_LAUNCH_CHECK();
    return 0;
}
