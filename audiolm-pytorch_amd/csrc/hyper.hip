// Hyper-connection residual-stream mixing around every attention / feed-forward branch (gfx950, HBM-bound).
//
// Replaces the third-party `hyper_connections` modules the reference wraps each block in
// (audiolm_pytorch.py:24, 446, 452-454, 524, 551; arithmetic restated in oracle/hyper_connections_restated.py):
//   width : n_s = normalize(R_s) * sqrt(D) * (g + 1);  alpha = tanh(n Wa) * sa + Aa;  beta = tanh(n wb) * sb + Bb
//           x = sum_s alpha[s][0] R_s   (branch input; the branch pre-LayerNorm audiolm_pytorch.py:347 / :254 is fused here)
//   depth : R'_t = sum_s alpha[s][t+1] R_s + beta[t] * y
// Residual streams R: [B][S][N][D]  (the reference's '(b s) n d'), stored fp32 or -- `r_bf16` -- bf16: under `accelerator.autocast()`
// (trainer.py:1241) the reference's streams ARE bf16 tensors from the first width connection on (its einsums autocast); the arithmetic
// here is fp32 in registers either way, only the HBM image changes (half the traffic of these HBM-bound kernels).  Tensors that stand
// for all streams at once (`*_bcast`: the embedded input x, the gradient of the final stream sum) are always fp32 [B*N][D].
//
// One wave64 owns one token: its S x D residual slab lives in registers (16-B coalesced loads, 4 x float4 per stream for
// D = 1024), every reduction (S norms, S*(S+2) dot products, LayerNorm statistics) is a wave shuffle butterfly: no LDS, no
// block barrier.  Parameter gradients are accumulated in registers over a grid-stride token loop, reduced over the 4 waves
// of a block through LDS and written as per-block partial rows (second stage: alm_colsum).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "../../include/audiolm_hip.h"

// A/B switch of the backward kernel's per-stream element loop (round 4): 1 = the token's per-stream scalars travel through a
// per-wave LDS record as broadcast splat pairs + folded weights (24 % fewer VALU instructions, but 256 VGPRs + scratch spills whose reloads drain the
// vector-memory queue); 0 = v_readlane broadcasts from lane-distributed registers (215 VGPRs, no spill).
#ifndef ALM_HC_LDSREC
#define ALM_HC_LDSREC 0
#endif

// Residual-stream tensor layout (internal to Transformer.forward: only the kernels of this file read or write it).  Shipped: [B][S][N][D] -- the S rows
// of one token lie N * D elements apart.  -DALM_HC_TOKEN_MAJOR=1 (measurement build, scripts/build_variant.sh): [B][N][S][D] -- one contiguous S * D block
// per token.  Round-5 micro-benchmark (scripts/ubench/stream_layout.hip, profiles/r5a_stream_layout.log): +5-10 % on the bare access pattern.
#ifndef ALM_HC_TOKEN_MAJOR
#define ALM_HC_TOKEN_MAJOR 0
#endif
template <typename I> __device__ __forceinline__ I hc_rbase(I b, I n, I S, I N, I D) { return ALM_HC_TOKEN_MAJOR ? (b * N + n) * S * D : (b * S * N + n) * D; }
template <typename I> __device__ __forceinline__ I hc_sstride(I N, I D) { return ALM_HC_TOKEN_MAJOR ? D : N * D; }

namespace {

constexpr bool HC_LDSREC = ALM_HC_LDSREC != 0;
constexpr float LN_EPS = 1e-5f;
constexpr float NORM_EPS = 1e-12f;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4bf(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4bf(bf16_t* p, float4 v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }
// residual-stream element access: RT = float | bf16_t
__device__ __forceinline__ float4 ldR(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldR(const bf16_t* p) { return ld4bf(p); }
__device__ __forceinline__ void stR(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// bf16 stream stores are NON-TEMPORAL: R' / dR' (134 MB per launch) are next read several kernels later, while the X / XN / dY rows written beside them
// (st4bf, cached) feed the GEMM that follows immediately -- hc_fwd 84 -> 80 us stand-alone, -0.05 ms per training step (three interleaved runs)
typedef unsigned int nt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stR(bf16_t* p, float4 v) {
    nt_u2 w;
    w.x = pack_bf2(v.x, v.y);
    w.y = pack_bf2(v.z, v.w);
    __builtin_nontemporal_store(w, reinterpret_cast<nt_u2*>(p));
}
// 4 consecutive residual elements exactly as they sit in HBM (fp32: 4 registers, bf16: 2): what a software-prefetched token keeps in
// registers while the previous token is being processed
template <typename RT> struct Raw4;
template <> struct Raw4<float> { float4 v; };
template <> struct Raw4<bf16_t> { uint2 v; };
__device__ __forceinline__ void ldraw(Raw4<float>& d, const float* p) { d.v = ld4(p); }
__device__ __forceinline__ void ldraw(Raw4<bf16_t>& d, const bf16_t* p) { d.v = *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ void zraw(Raw4<float>& d) { d.v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zraw(Raw4<bf16_t>& d) { d.v = make_uint2(0u, 0u); }
__device__ __forceinline__ float4 unraw(const Raw4<float>& d) { return d.v; }
__device__ __forceinline__ float4 unraw(const uint2& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 unraw(const Raw4<bf16_t>& d) { return unraw(d.v); }

// the value a later kernel will read back from the stored image (bf16 streams: what was computed from must equal what is stored,
// or the backward's recomputation of x from R would differ from the forward's)
template <typename RT> __device__ __forceinline__ float4 as_stored(float4 v) {
    if constexpr (sizeof(RT) == 4) return v;
    else {
        const uint32_t a = pack_bf2(v.x, v.y), b = pack_bf2(v.z, v.w);
        return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u));
    }
}
__device__ __forceinline__ float& f4(float4& v, int c) { return reinterpret_cast<float*>(&v)[c]; }
__device__ __forceinline__ float f4c(const float4& v, int c) { return reinterpret_cast<const float*>(&v)[c]; }

// coef layout per token (floats): alpha[S][S+1] | beta[S] | a_pre[S][S+1] | b_pre[S] | rn[S]
template <int S> struct Coef {
    static constexpr int A = 0, Bt = S * (S + 1), AP = Bt + S, BP = AP + S * (S + 1), RN = BP + S, W = RN + S;
};

struct HcParams {
    const float* hc_gamma;   // [D]   RMSNorm gamma (init 0)
    const float* Wa;         // [D][S+1] dynamic_alpha_fn
    const float* sa;         // []    dynamic_alpha_scale
    const float* Aa;         // [S][S+1] static_alpha
    const float* wb;         // [D]   dynamic_beta_fn
    const float* sb;         // []    dynamic_beta_scale
    const float* Bb;         // [S]   static_beta
};

// ------------------------------------------------------------------------------------------------------------------
// Work decomposition: ONE TOKEN = WPT waves (WPT * 256 >= D), thread -> 4 consecutive elements of every stream; a 256-thread
// block works on 4 / WPT tokens at a time and grid-strides over the rest.  All token-wide reductions (S norms, S*(S+2) dot
// products, S stream sums ...) are done TOGETHER: a multi-value butterfly leaves the wave total of slot (lane & 31) in every lane
// with 31 shuffles instead of 6 per value, one LDS exchange combines the WPT waves, and lane l post-processes slot l
// (one tanh per lane).  Per-element weights and parameter-gradient accumulators live in registers across the token loop.
// ------------------------------------------------------------------------------------------------------------------
// lane-permute primitives (VALU only, no LDS round trip)
__device__ __forceinline__ float dpp_xor1(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float dpp_xor2(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float dpp_ror4(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x124, 0xF, 0xF, true)); }  // row_ror:4
__device__ __forceinline__ float dpp_ror8(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x128, 0xF, 0xF, true)); }  // row_ror:8 (= lane ^ 8)
// v_permlane32_swap: lanes 32..63 of `a` <-> lanes 0..31 of `b`; afterwards a + b = (a[l] + a[l+32]) in the lower half, (b[l-32] + b[l]) in the upper
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_permlane16_swap: odd 16-lane rows of `a` <-> even rows of `b`; afterwards a + b = a-pair sums in even rows, b-pair sums in odd rows
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the 64 lanes, every lane gets it: 4 DPP adds + 2 swap-adds instead of 6 ds_bpermute round trips
__device__ __forceinline__ float wave_sum_fast(float v) {
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_ror4(v);
    v += dpp_ror8(v);
    v = swap16_add(v, v);
    return swap32_add(v, v);
}

// Slot <-> lane maps of the multi-value butterflies below.  NV < 32: slot = lane & (NV - 1).  NV = 32: the halving stages are ordered so that
// the stages with MANY value pairs use the cheapest exchanges -- 16 pairs on v_permlane32_swap (lane bit 5), 8 on v_permlane16_swap (bit 4),
// 4 on DPP quad_perm xor 1 (bit 0), 2 on DPP xor 2 (bit 1), 1 on a shuffle (bit 2), and the last all-reduce over bit 3 is one DPP row_ror:8:
// 71 VALU instructions instead of 31 x (2 selects + ds_bpermute + add) -- so slot bits (0,1,2,3,4) <-> lane bits (5,4,0,1,2), and the two
// lanes l, l ^ 8 hold the same slot (`primary`: the one with bit 3 clear).
template <int NV> struct SlotMap {
    static __device__ __forceinline__ int slot(int lane) { return lane & (NV - 1); }
    static constexpr int lane_of(int s) { return s; }
    static __device__ __forceinline__ int lane_of_dyn(int s) { return s; }
    static __device__ __forceinline__ bool primary(int lane) { return lane < NV; }
};
template <> struct SlotMap<32> {
    static __device__ __forceinline__ int slot(int l) { return ((l >> 5) & 1) | (((l >> 4) & 1) << 1) | ((l & 1) << 2) | (((l >> 1) & 1) << 3) | (((l >> 2) & 1) << 4); }
    static constexpr int lane_of(int s) { return ((s & 1) << 5) | (((s >> 1) & 1) << 4) | ((s >> 2) & 1) | (((s >> 3) & 1) << 1) | (((s >> 4) & 1) << 2); }
    static __device__ __forceinline__ int lane_of_dyn(int s) { return lane_of(s); }
    static __device__ __forceinline__ bool primary(int lane) { return (lane & 8) == 0; }
};

template <int NV>
__device__ __forceinline__ float bfly(float (&v)[NV], int lane) {
    // NV (power of two <= 32) values per lane -> every lane returns the sum over the 64 lanes of slot SlotMap<NV>::slot(lane)
    if constexpr (NV == 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = swap32_add(v[2 * i], v[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = swap16_add(v[2 * i], v[2 * i + 1]);
        const bool u0 = lane & 1, u1 = (lane >> 1) & 1, u2 = (lane >> 2) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ta = v[2 * i] + dpp_xor1(v[2 * i]), tb = v[2 * i + 1] + dpp_xor1(v[2 * i + 1]);
            v[i] = u0 ? tb : ta;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float ta = v[2 * i] + dpp_xor2(v[2 * i]), tb = v[2 * i + 1] + dpp_xor2(v[2 * i + 1]);
            v[i] = u1 ? tb : ta;
        }
        const float keep = u2 ? v[1] : v[0], send = u2 ? v[0] : v[1];
        float r = keep + __shfl_xor(send, 4, 64);
        return r + dpp_ror8(r);
    } else {
        int st = 0;
#pragma unroll
        for (int n = NV / 2; n >= 1; n >>= 1, ++st) {
            const bool up = (lane >> st) & 1;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const float keep = up ? v[2 * i + 1] : v[2 * i];
                const float send = up ? v[2 * i] : v[2 * i + 1];
                v[i] = keep + __shfl_xor(send, 1 << st, 64);
            }
        }
        float r = v[0];
#pragma unroll
        for (int m = NV; m < 64; m <<= 1) r += __shfl_xor(r, m, 64);
        return r;
    }
}

// 4 values per lane -> every lane gets the 64-lane sum of slot (lane >> 5) | ((lane >> 4) & 1) << 1: two swap-adds, then an all-reduce inside the
// 16-lane row (3 DPP adds + ... all VALU)
__device__ __forceinline__ float bfly4(float (&v)[4]) {
    const float a = swap32_add(v[0], v[1]), b = swap32_add(v[2], v[3]);
    float r = swap16_add(a, b);
    r += dpp_xor1(r);
    r += dpp_xor2(r);
    r += dpp_ror4(r);
    return r + dpp_ror8(r);
}
__device__ __forceinline__ int bfly4_slot(int lane) { return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1); }
__device__ __forceinline__ int bfly4_lane_of(int s) { return ((s & 1) << 5) | (((s >> 1) & 1) << 4); }

// tanh(x) = 1 - 2 / (exp(2x) + 1) with the hardware exponential and reciprocal: |error| <= ~2e-7 absolute (the library tanhf costs ~40
// instructions with its range selects; this value feeds alpha = tanh(.) * scale + static, where an absolute 2e-7 is far below the bf16 GEMM noise)
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.f * fminf(fmaxf(x, -15.f), 15.f));
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
// 1 / max(sqrt(ss), NORM_EPS)  (F.normalize's eps clamp) as min(rsqrt(ss), 1 / NORM_EPS): one v_rsq instead of sqrt + IEEE divide
__device__ __forceinline__ float inv_norm(float ss) { return fminf(__builtin_amdgcn_rsqf(ss), 1.f / NORM_EPS); }

__device__ __forceinline__ float lane_bcast(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }

typedef float hc_f2 __attribute__((ext_vector_type(2)));            // element pair: the operand shape of v_pk_fma_f32
typedef float hc_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hc_f2 f2fma(hc_f2 a, hc_f2 b, hc_f2 c) { return __builtin_elementwise_fma(a, b, c); }
// <a, b> of two packed bf16 pairs + c on the raw HBM image (v_dot2_f32_bf16: products of bf16 values are exact in fp32)
typedef __attribute__((ext_vector_type(2))) __bf16 hc_bf2;
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hc_bf2, a), __builtin_bit_cast(hc_bf2, b), c, false);
}

// combine the per-wave totals of the WPT waves of a token through LDS (one block barrier); every lane gets slot (lane & (NV-1))
// RAWBAR: the kernel has LDS-DMA loads in flight (hc_bwd's GL variant).  __syncthreads() is a fence + barrier, and the fence waits for every pending
// vector-memory operation -- the DMA prefetch of the NEXT token included.  The partial sums only need this wave's LDS writes to be done before the barrier:
// s_waitcnt lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier_raw() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int WPT, int NV, bool RAWBAR = false>
__device__ __forceinline__ float token_combine(float tot, float* red, int tok, int wv, int lane) {
    if (WPT == 1) return tot;
    using SM = SlotMap<NV>;
    const int sl = SM::slot(lane);
    if (SM::primary(lane)) red[(tok * WPT + wv) * NV + sl] = tot;
    if constexpr (RAWBAR) lds_barrier_raw();
    else __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < WPT; ++w) r += red[(tok * WPT + w) * NV + sl];
    return r;
}

struct HcFwdArgs {
    const void* R_in; int rin_bcast;                      // residual streams RT [B][S][N][D], or (rin_bcast) ONE fp32 [B*N][D] tensor every stream equals (:524)
    const bf16_t* y; int ldy; const float* coef_prev; void* R_out;           // row strides: int (fewer SGPRs / SALU ops in the token loop)
    HcParams hp; const float* ln_gamma;
    bf16_t* x_out; int ldx; bf16_t* xn_out; int ldxn; float* mean_out; float* rstd_out; float* coef; float* xs_out;
    float* xn32_out;                                      // FINAL: the final LayerNorm output in fp32 [B*N][D] instead of bf16 xn_out (logit heads)
    int B, N, D;
};

// forward: [depth connection of the previous branch] -> [width connection + pre-LayerNorm of the next branch | stream sum + final LN]
//   DEPTH: r_t = sum_s alpha_p[s][t+1] R_in[s] + beta_p[t] y       (written to R_out unless FINAL)
//   WIDTH: coefficients of the next branch from r, x = sum_s alpha[s][0] r_s, xn = LN(x) * ln_gamma
//   FINAL: xs = sum_t r_t (reference audiolm_pytorch.py:551), xn = LN(xs) * ln_gamma (:555)
// PF: software prefetch (bf16 streams, no `rin_bcast`): the NEXT token's loads are issued before this token is processed, so the HBM latency
// hides under the reductions instead of heading every iteration.  The loop is unrolled by two over a pair of register sets (no copies: the
// loads stay in flight until the set is unpacked); loads are unconditional on a clamped token (a skipped token re-reads token 0).
template <typename RT, int S, int WPT, bool DEPTH, bool WIDTH, bool FINAL, bool PF>
__global__ __launch_bounds__(256) void hc_fwd_kernel(HcFwdArgs a) {
    using C = Coef<S>;
    const RT* const Rin = reinterpret_cast<const RT*>(a.R_in);
    RT* const Rout = reinterpret_cast<RT*>(a.R_out);
    constexpr int TPB = 4 / WPT;
    constexpr int NV = (S >= 3) ? 32 : 16;                       // slots: ss[S] | dots[S][S+2] | sums[S]  (S = 4: 32, 3: 21, 2: 12)
    constexpr int O_DOT = S, O_SUM = S + S * (S + 2);
    static_assert(O_SUM + S <= NV, "slot budget");
    __shared__ float red[TPB * WPT * NV];
    __shared__ float red2[2][TPB * WPT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok = wave / WPT, wv = wave % WPT;
    const int e0 = (wv * 64 + lane) * 4;
    const bool eok = e0 < a.D;
    const long long M = (long long)a.B * a.N;
    const float cD = sqrtf((float)a.D), invD = 1.f / (float)a.D;

    float w[S + 2][4];
    float4 lng = make_float4(0.f, 0.f, 0.f, 0.f);
    float sa = 0.f, sb = 0.f;
    if (WIDTH || FINAL) { if (eok) lng = ld4(a.ln_gamma + e0); }
    if (WIDTH) {
        sa = *a.hp.sa; sb = *a.hp.sb;
        // UNCONDITIONAL loads at a clamped index, selects afterwards: written as `eok ? p[e] * g1 : 0` each of the 28 parameter reads became its own
        // exec-masked block with an `s_waitcnt vmcnt(0)` right behind the load -- 28 memory latencies one after the other at the top of every
        // (persistent) workgroup, ~10 % of the kernel.  Now they are all in flight together.
        const int eb = eok ? e0 : 0;
        float gq[4], wa[4][S + 1], wbq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gq[c] = a.hp.hc_gamma[eb + c];
            wbq[c] = a.hp.wb[eb + c];
#pragma unroll
            for (int t = 0; t < S + 1; ++t) wa[c][t] = a.hp.Wa[(long long)(eb + c) * (S + 1) + t];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float g1 = eok ? gq[c] + 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < S + 1; ++t) w[t][c] = eok ? wa[c][t] * g1 : 0.f;
            w[S + 1][c] = eok ? wbq[c] * g1 : 0.f;
        }
    }

    // slot role of this lane in the width connection's reduction (loop-invariant; see SlotMap): ss[S] | dots[S][S+2] | sums[S]
    using SM = SlotMap<NV>;
    const int l = SM::slot(lane);
    const int di = l - O_DOT;
    const bool is_dot = di >= 0 && di < S * (S + 2);
    const int ds = is_dot ? di / (S + 2) : 0, dt = is_dot ? di % (S + 2) : 0;
    const bool is_beta = dt == S + 1;
    float stat = 0.f;                                                        // static part of this lane's coefficient
    if (WIDTH && is_dot) stat = is_beta ? a.hp.Bb[ds] : a.hp.Aa[ds * (S + 1) + dt];

    const int niter = (int)((M + TPB - 1) / TPB);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long sND = hc_sstride<long long>(a.N, a.D);                   // stream stride of R
    // token coordinates (b, n) advance incrementally: no 64-bit divisions in the loop; `valid` is wave-uniform, so the loads sit in ONE
    // uniform branch instead of an exec-masked region each
    unsigned bb, nn;                                                         // coordinates of the token whose loads are issued next
    {
        const unsigned m0 = (unsigned)(blockIdx.x * TPB + tok);
        bb = m0 / (unsigned)a.N;
        nn = m0 % (unsigned)a.N;
    }
    const unsigned tstride = gridDim.x * TPB, sbb = tstride / (unsigned)a.N, snn = tstride % (unsigned)a.N;
    struct Tok { int m; int b, n; bool valid; };                             // m = b * N + n < 2^31 (checked by the launcher)
    auto next_tok = [&](int it_) {
        Tok t;
        t.m = it_ * TPB + tok;
        t.b = (int)bb; t.n = (int)nn;
        t.valid = it_ < niter && t.m < M;
        bb += sbb; nn += snn;
        if (nn >= (unsigned)a.N) { nn -= (unsigned)a.N; ++bb; }
        return t;
    };
    struct In { Raw4<RT> r[S]; float4 rb; uint2 y; float cf; };              // one token's inputs as loaded; cf: lane l = coef_prev[m][l] (DEPTH)
    const int cl = lane < C::W ? lane : 0;                                   // this lane's entry of a token's coefficient record
    auto issue_pf = [&](In& in, const Tok& t) {
        const int m_ = t.valid ? t.m : 0;
        const int b_ = t.valid ? t.b : 0, n_ = t.valid ? t.n : 0;                  // PF: D == WPT * 256, every lane in range
        const RT* Rt = Rin + hc_rbase<long long>(b_, n_, S, a.N, a.D) + e0;
#pragma unroll
        for (int s = 0; s < S; ++s) ldraw(in.r[s], Rt + s * sND);
        if (DEPTH) { in.y = *reinterpret_cast<const uint2*>(a.y + (long long)m_ * a.ldy + e0); in.cf = a.coef_prev[(long long)m_ * C::W + cl]; }
    };
    auto issue_plain = [&](In& in, const Tok& t) {
#pragma unroll
        for (int s = 0; s < S; ++s) zraw(in.r[s]);
        in.rb = z4;
        in.y = make_uint2(0u, 0u);
        in.cf = 0.f;
        if (DEPTH) in.cf = a.coef_prev[(t.valid ? t.m : 0) * C::W + cl];
        if (t.valid && eok) {
            if (a.rin_bcast) {
                in.rb = ld4(reinterpret_cast<const float*>(a.R_in) + (long long)t.m * a.D + e0);
            } else {
                const RT* Rt = Rin + hc_rbase<long long>(t.b, t.n, S, a.N, a.D) + e0;
#pragma unroll
                for (int s = 0; s < S; ++s) ldraw(in.r[s], Rt + s * sND);
            }
            if (DEPTH) in.y = *reinterpret_cast<const uint2*>(a.y + (long long)t.m * a.ldy + e0);
        }
    };
    auto process = [&](const In& in, const Tok& t) {
        const int m = t.m;
        // STRAIGHT: see hc_bwd_kernel -- with one token per workgroup pass the prefetching loop only processes valid tokens, and every store of the
        // token is made unconditional (all waves / lanes write; duplicates carry the same value to the same address), so that the compiler can count
        // the vector-memory operations between a prefetch and its first use instead of draining the queue (s_waitcnt vmcnt(0)) once per token pair
        constexpr bool STRAIGHT = PF && TPB == 1;
        const bool valid = STRAIGHT ? true : t.valid;
        const int b = t.b, n = t.n;
        const bool ld_ok = valid && eok;
        float4 r[S];
        float4 yv = z4;
        if (PF) {
            // every lane owns 4 real elements (the launcher takes this path only for D == WPT * 256) and a skipped token re-reads token 0:
            // whatever is computed from it is never stored, so no zero-selects are needed
#pragma unroll
            for (int s = 0; s < S; ++s) r[s] = unraw(in.r[s]);
            if (DEPTH) yv = unraw(in.y);
        } else {
            if (a.rin_bcast) {
#pragma unroll
                for (int s = 0; s < S; ++s) r[s] = in.rb;
            } else {
#pragma unroll
                for (int s = 0; s < S; ++s) r[s] = unraw(in.r[s]);
            }
            if (DEPTH) yv = unraw(in.y);
        }
        if (DEPTH) {
            // the previous branch's coefficients arrived with the token's other loads (one lane-distributed record): no load sits between the
            // prefetch of the next token and the end of this one, so that prefetch really stays in flight
            float4 o[S];
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const float bt = lane_bcast(in.cf, C::Bt + t);
                o[t] = make_float4(bt * yv.x, bt * yv.y, bt * yv.z, bt * yv.w);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float al = lane_bcast(in.cf, C::A + s * (S + 1) + t + 1);
                    o[t].x += al * r[s].x; o[t].y += al * r[s].y; o[t].z += al * r[s].z; o[t].w += al * r[s].w;
                }
            }
#pragma unroll
            for (int t = 0; t < S; ++t) {
                r[t] = FINAL ? o[t] : as_stored<RT>(o[t]);
                if (!FINAL && ld_ok) stR(Rout + hc_rbase<long long>(b, n, S, a.N, a.D) + t * sND + e0, o[t]);
            }
        }
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        float mean = 0.f;
        if (WIDTH) {
            float v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float rv = f4c(r[s], c);
                    v[s] += rv * rv;
                    v[O_SUM + s] += rv;
#pragma unroll
                    for (int t = 0; t < S + 2; ++t) v[O_DOT + s * (S + 2) + t] += rv * w[t][c];
                }
            float tot = bfly<NV>(v, lane);
            tot = token_combine<WPT, NV>(tot, red, tok, wv, lane);
            // every lane post-processes its slot
            float rn[S];
#pragma unroll
            for (int s = 0; s < S; ++s) rn[s] = inv_norm(lane_bcast(tot, SM::lane_of(s)));
            float rn_l = rn[0];
#pragma unroll
            for (int s = 1; s < S; ++s) rn_l = (ds == s) ? rn[s] : rn_l;
            const float pre = tot * rn_l * cD;
            const float th = tanh_fast(pre);
            const float coefv = th * (is_beta ? sb : sa) + stat;             // alpha[ds][dt] or beta[ds]
            if constexpr (STRAIGHT) {
                // the coefficient record, two stores by EVERY lane of every wave of the token: a (stream, coefficient) slot writes its coefficient and its
                // pre-activation, every other lane 1 / |R_(l mod S)| twice; all four waves hold identical totals (same LDS sums in the same order)
                float* cp = a.coef + (long long)m * C::W;
                float rn_k = rn[0];
#pragma unroll
                for (int s = 1; s < S; ++s) rn_k = (l % S == s) ? rn[s] : rn_k;
                const int i1 = is_dot ? (is_beta ? C::Bt + ds : C::A + ds * (S + 1) + dt) : C::RN + l % S;
                const int i2 = is_dot ? (is_beta ? C::BP + ds : C::AP + ds * (S + 1) + dt) : C::RN + l % S;
                cp[i1] = is_dot ? coefv : rn_k;
                cp[i2] = is_dot ? pre : rn_k;
            } else if (valid && wv == 0 && SM::primary(lane)) {
                float* cp = a.coef + (long long)m * C::W;
                if (is_dot) {
                    if (is_beta) { cp[C::Bt + ds] = coefv; cp[C::BP + ds] = pre; }
                    else { cp[C::A + ds * (S + 1) + dt] = coefv; cp[C::AP + ds * (S + 1) + dt] = pre; }
                } else if (l < S) {
                    cp[C::RN + l] = inv_norm(tot);
                }
            }
            float msum = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float a0 = lane_bcast(coefv, SM::lane_of(O_DOT + s * (S + 2)));
                x.x += a0 * r[s].x; x.y += a0 * r[s].y; x.z += a0 * r[s].z; x.w += a0 * r[s].w;
                msum += a0 * lane_bcast(tot, SM::lane_of(O_SUM + s));
            }
            mean = msum * invD;
        }
        if (FINAL) {
#pragma unroll
            for (int s = 0; s < S; ++s) { x.x += r[s].x; x.y += r[s].y; x.z += r[s].z; x.w += r[s].w; }
            float sm = wave_sum_fast(x.x + x.y + x.z + x.w);
            if (WPT > 1) {
                if (lane == 0) red2[0][tok * WPT + wv] = sm;
                __syncthreads();
                sm = 0.f;
#pragma unroll
                for (int q = 0; q < WPT; ++q) sm += red2[0][tok * WPT + q];
            }
            mean = sm * invD;
            if (ld_ok) *reinterpret_cast<float4*>(a.xs_out + (long long)m * a.D + e0) = x;
        }
        if (WIDTH || FINAL) {
            float q = 0.f;
            if (eok) {
                const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
                q = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
            q = wave_sum_fast(q);
            if (WPT > 1) {
                if (lane == 0) red2[1][tok * WPT + wv] = q;
                __syncthreads();
                q = 0.f;
#pragma unroll
                for (int j = 0; j < WPT; ++j) q += red2[1][tok * WPT + j];
            }
            const float rstd = rsqrtf(q * invD + LN_EPS);
            if (ld_ok) {
                const float4 xnv = make_float4((x.x - mean) * rstd * lng.x, (x.y - mean) * rstd * lng.y, (x.z - mean) * rstd * lng.z,
                                               (x.w - mean) * rstd * lng.w);
                if (FINAL && a.xn32_out) *reinterpret_cast<float4*>(a.xn32_out + (long long)m * a.D + e0) = xnv;
                else st4bf(a.xn_out + (long long)m * a.ldxn + e0, xnv);
                if (WIDTH && a.x_out) st4bf(a.x_out + (long long)m * a.ldx + e0, x);
            }
            if (STRAIGHT || (valid && wv == 0 && lane == 0)) { a.mean_out[m] = mean; a.rstd_out[m] = rstd; }      // (STRAIGHT: every lane, same value)
        }
    };
    if constexpr (PF) {
        In wa, wb2;
        Tok ta = next_tok(blockIdx.x), tb = ta;
        issue_pf(wa, ta);
        for (int it = blockIdx.x; it < niter; it += 2 * (int)gridDim.x) {
            tb = next_tok(it + gridDim.x);
            issue_pf(wb2, tb);
            process(wa, ta);
            if (it + gridDim.x >= niter) break;
            ta = next_tok(it + 2 * (int)gridDim.x);
            issue_pf(wa, ta);
            process(wb2, tb);
        }
    } else {
        for (int it = blockIdx.x; it < niter; it += gridDim.x) {
            const Tok t = next_tok(it);
            In w;
            issue_plain(w, t);
            process(w, t);
        }
    }
}

struct HcBwdArgs {
    const void* dRn; int bcast;                          // gradient wrt the width connection's residual output: RT [B][S][N][D], or (bcast) fp32 [M][D] shared by all streams
    const float* dx; int lddx;                           // gradient wrt the branch input x (fp32)                      [LNF == false]   (row strides: int -- fewer SGPRs / SALU ops in the token loop)
    const bf16_t* dxn; int lddxn;                        // gradient wrt the branch's pre-LayerNorm OUTPUT xn (bf16)     [LNF == true]
    const bf16_t* extra; int ldex;                       //   + gradient arriving at x directly (attention: the K/V path), or NULL
    const float* mean; const float* rstd; const float* ln_gamma;        //   LayerNorm statistics saved by the forward, LN weight
    const void* R; int r_bcast;                          // residual input of the width connection (RT [B][S][N][D], or one fp32 [B*N][D] tensor for all streams)
    const float* coef; const float* dbeta;
    HcParams hp;
    void* dR; float* dsum; float dsum_scale;             // dR RT [B][S][N][D] and / or dsum fp32 [B*N][D] = dsum_scale * sum over streams of dR (gradient of the :524 expand; the scale
                                                         // is grad_shrink's alpha, audiolm_pytorch.py:93-94: folded in here instead of a separate pass over dx)
    float* partial;
    const bf16_t* y; int ldy; const float* coef_prev; bf16_t* dy; int lddy; float* dbeta_out;
    int B, N, D;
};

// backward: [width connection of branch k+1] -> [depth connection of branch k]
//   WIDTH: dR_s = alpha[s][0] dx + sum_t alpha[s][t+1] dRn_t + (dynamic-coefficient / RMS-norm terms); parameter-gradient partial sums
//   DEPTH: dy = sum_t beta_p[t] dR_t (bf16), dbeta_p[t] = <dR_t, y>      (on dRn itself when there is no width part)
//   LNF  : the branch's pre-LayerNorm backward (audiolm_pytorch.py:191-198 autograd) is done HERE: x = sum_s alpha[s][0] R_s and
//          xhat are recomputed in fp32 from the residual streams, dx = rstd (g - mean(g) - xhat mean(g xhat)) + extra with g = dxn * gamma,
//          and <dx, R_s> is assembled from reduction slots (<rstd g + extra, R_s>, sum R_s, <xhat, R_s>, sum g, sum g xhat), so the
//          whole thing still needs ONE token-wide reduction; the LN weight gradient joins the partial rows.  No fp32 dx tensor and no
//          separate LayerNorm-backward launch exist on the 4-stream path.
// partial row (floats): raw_a[S+1][D] | raw_b[D] | dln[D] | dAa[S][S+1] | dBb[S] | dsa | dsb   with raw_* = sum over tokens, streams of
// nhat * (dap | dbp): dWa = (gamma+1) raw_a, dwb = (gamma+1) raw_b, dgamma = sum_t Wa raw_a + wb raw_b  (alm_hc_param_grads).
// PF: see hc_fwd_kernel.  BC (PF only): 0 = stream tensors everywhere, 1 = dRn is the broadcast fp32 [M][D] tensor (`bcast`: the last branch, behind
// the final stream sum), 2 = R is (`r_bcast`: the first branch, right after the stream expansion) -- the two once-per-step launches prefetch too.
// The kernel arguments re-read from the kernarg segment (scalar loads: free for the VALU) instead of held in SGPRs across the token loop: hc_bwd keeps
// ~25 pointers / strides alive next to 2 x 13 per-stream scalars, more than the 102 SGPRs -- the compiler parked the surplus in VGPR lanes and fetched it
// back with v_readlane INSIDE the loop (56 of the ~1190 VALU instructions per token and wave).  The empty asm makes every call a fresh base the loads
// cannot be hoisted over.  HcBwdArgs is the kernel's first and only parameter: it sits at offset 0 of the kernarg segment (static_assert below).
// base + a 32-bit BYTE offset: the form the global_load / global_store "SGPR base + 32-bit VGPR offset" addressing takes (an element index would be
// widened to 64 bits before the scale, i.e. a v_lshl_add_u64 per access)
template <typename T> __device__ __forceinline__ T* at_bytes(T* base, unsigned byte_off) {
    typedef __attribute__((address_space(1))) char gchar;                      // every tensor of the C ABI is device (global) memory
    return (T*)((gchar*)(base) + byte_off);
}
typedef const __attribute__((address_space(4))) HcBwdArgs* HcBwdKArgs;
static_assert(std::is_trivially_copyable<HcBwdArgs>::value && alignof(HcBwdArgs) <= 8, "passed by value in the kernarg segment, read back in place");
__device__ __forceinline__ HcBwdKArgs hc_bwd_kargs() {
    HcBwdKArgs p = (HcBwdKArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// ALM_HC_PROBE (measurement builds only, never the shipped library: scripts/build_variant.sh): where does hc_bwd's time go?
//   1 = memory pattern only (same loads, same stores, trivial arithmetic, no reductions / barriers)     2 = full arithmetic, stores to L2-resident rows
//   3 = full arithmetic, loads from L2-resident rows                                                    4 = both (no HBM traffic: arithmetic + latency chains alone)
//   5 = production arithmetic and traffic, the SECOND block barrier of a token (depth connection's reduction) left out: what one barrier per token would buy
#ifndef ALM_HC_PROBE
#define ALM_HC_PROBE 0
#endif
#ifndef ALM_HC_GLREC
#define ALM_HC_GLREC 0                   // 1: the LDS-DMA variant takes the LDS scalar-record element loop (see hcrec) at two workgroups per CU
#endif
#ifndef ALM_HC_BWD_OCC
#define ALM_HC_BWD_OCC 2                  // workgroups per CU the register allocation is bounded for (3 was tried: see DESIGN.md section 8.9)
#endif
// GL (round 4; PF, BC == 0, bf16 streams, WIDTH + DEPTH + LNF, WPT == 4 only): the token's inputs travel HBM -> LDS by DMA (global_load_lds) instead of
// through a second register set.  Each wave fetches exactly the 5.5 KB it consumes -- its 256-element segment of the 2 x S stream rows and of dxn / extra / y
// (six 16-byte instructions: lanes 0-31 one segment, lanes 32-63 the next, LDS image lane-linear) and the token's scalar records (three 4-byte instructions:
// coefficient record, previous branch's record, [dbeta | mean | rstd]) -- into its own slice of one of two LDS buffers, and reads it back with ds_read_b64 /
// b32 when the token's turn comes: no cross-wave hand-over, hence no barrier for the data.  EVERY load of the loop is on the DMA path (a VGPR-destination
// load next to an LDS-DMA makes the compiler drain vmcnt(0)), the kernel has ONE __shared__ object (a second one does the same), the reduction barriers
// are raw (see token_combine), and the two buffers reach the step lambda as __restrict__ parameters so that the reads of the current buffer wait for THEIR
// DMA only (counted vmcnt) and not for the one just issued.  The 56 registers of the two prefetch sets are gone: three workgroups per CU (<= 168 VGPRs,
// 52 KB of LDS each) with the prefetch intact -- the probes of DESIGN.md section 8.9 (b) say that is what the kernel is short of.
constexpr int GL_WAVE = 11 * 512 + 3 * 256, GL_BUF = 4 * GL_WAVE;        // per wave: 2 S + 3 row segments of 512 B, three 256-byte scalar records (S == 4)
template <typename RT, int S, int WPT, bool WIDTH, bool DEPTH, bool LNF, bool PF, int BC = 0, bool GL = false>
__global__ __launch_bounds__(256, GL ? (ALM_HC_GLREC ? 2 : 3) : ALM_HC_BWD_OCC) void hc_bwd_kernel(HcBwdArgs a) {
    static_assert(!GL || (PF && BC == 0 && WIDTH && DEPTH && LNF && WPT == 4 && sizeof(RT) == 2 && !HC_LDSREC), "GL: the production variant of the inner branches only");
    constexpr bool REC = HC_LDSREC || (GL && ALM_HC_GLREC != 0);        // the element loop reads the token's per-stream scalars from a per-wave LDS record
    using C = Coef<S>;
    const RT* const dRn = reinterpret_cast<const RT*>(a.dRn);
    const RT* const Rsv = reinterpret_cast<const RT*>(a.R);
    RT* const dRo = reinterpret_cast<RT*>(a.dR);
    constexpr int TPB = 4 / WPT;
    constexpr int NV = (S >= 3) ? 32 : 16;                       // width slots: dal[S][S+1] | (LNF) sum R_s [S] | <xhat, R_s> [S] | sum g | sum g xhat
    constexpr int NB = S * (S + 1);
    constexpr int O_SR = NB, O_XR = NB + S, O_C1 = NB + 2 * S, O_C2 = NB + 2 * S + 1;
    static_assert(O_C2 < NV, "slot budget");
    __shared__ float red_[GL ? 1 : 2][GL ? 1 : TPB * WPT * NV];  // parity-double-buffered: possibly the only barrier of an iteration
    __shared__ float redd_[GL ? 1 : 2][GL ? 1 : TPB * WPT * 4];
    constexpr int RED_F = TPB * WPT * NV, REDD_F = TPB * WPT * 4;
    constexpr int GL_REC_OFF = 2 * GL_BUF + (2 * RED_F + 2 * REDD_F) * 4;
    constexpr int GL_REC_BYTES = (GL && REC) ? 4 * ((S * (4 * (S + 2) + ((2 * (S + 1) + 3) / 4) * 4) + 4 + 3) / 4 * 4) * 4 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char glsm[GL ? GL_REC_OFF + GL_REC_BYTES : 16];     // GL: the kernel's ONLY LDS object
    float* const red_base = GL ? reinterpret_cast<float*>(glsm + 2 * GL_BUF) : &red_[0][0];
    float* const redd_base = GL ? reinterpret_cast<float*>(glsm + 2 * GL_BUF) + 2 * RED_F : &redd_[0][0];
    // Per-WAVE record of the token's per-stream scalars (round 4).  The element loop needs 12 token-wide scalars per stream (the five dap, dbp, the
    // normalisation term, five alpha); they used to be fetched one by one with v_readlane from lane-distributed registers -- 74 VALU instructions per
    // token and wave, a fifth of that loop, plus the moves that build the {x, x} register pairs v_pk_fma_f32 wants.  Now the lanes that own a value
    // write it ONCE to LDS as a splat pair and every lane reads the pairs back with broadcast ds_read_b128 (same address in all lanes: conflict-free,
    // issued on the LDS pipe, not the VALU).  Written and read by the same wave: LDS operations of a wave execute in order -- no barrier.
    // Layout per stream (floats): S + 2 slots of [dR, dR, q, q] (dR = dpre / |R_s| for the S + 1 alpha slots and the beta slot, q = dR pre / |R_s|),
    // then the S + 1 alpha pairs, padded to 16 bytes; 4 dump floats at the end take the writes of the lanes that own nothing.
    constexpr int AF = ((2 * (S + 1) + 3) / 4) * 4, RS = 4 * (S + 2) + AF, RECW = S * RS + 4;
    constexpr int RECW4 = (RECW + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float hcrec_[(GL || !REC) ? 1 : 4][(GL || !REC) ? 4 : RECW4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok = wave / WPT, wv = wave % WPT;
    const int e0 = (wv * 64 + lane) * 4;
    const bool eok = PF || e0 < a.D;                                        // PF: D == WPT * 256, every lane in range (no selects in the loop)
    const long long M = (long long)a.B * a.N;
    const float cD = sqrtf((float)a.D);

    float wa[S + 1][4], wbv[4], g1[4];
    float rawa[S + 1][4], rawb[4], dln[4] = {0.f, 0.f, 0.f, 0.f};
    float4 lng = make_float4(0.f, 0.f, 0.f, 0.f);
    float sa = 0.f, sb = 0.f;
    float accA = 0.f, accsa = 0.f, accB = 0.f, accsb = 0.f;      // lane l of the token's wave 0: slot-l scalar statistics
    if (WIDTH) {
        sa = *a.hp.sa; sb = *a.hp.sb;
        if (LNF && eok) lng = ld4(a.ln_gamma + e0);
        const int eb = eok ? e0 : 0;                             // unconditional loads at a clamped index, selects afterwards: see hc_fwd_kernel
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            g1[c] = a.hp.hc_gamma[eb + c];
            wbv[c] = a.hp.wb[eb + c];
#pragma unroll
            for (int t = 0; t < S + 1; ++t) wa[t][c] = a.hp.Wa[(long long)(eb + c) * (S + 1) + t];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // (gamma + 1) sqrt(D) is folded into the weights once: the element loop's dn * g1 * cD becomes part of the dot with dap / dbp (4 registers and 4
            // multiplies per stream less; out-of-range lanes: zero weights)
            g1[c] = eok ? g1[c] + 1.f : 0.f;
            const float gc = g1[c] * cD;
            wbv[c] = wbv[c] * gc;
            rawb[c] = 0.f;
#pragma unroll
            for (int t = 0; t < S + 1; ++t) { wa[t][c] = wa[t][c] * gc; rawa[t][c] = 0.f; }
        }
    }

    const int niter = (int)((M + TPB - 1) / TPB);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long sND = hc_sstride<long long>(a.N, a.D);
    unsigned bb, nn;                                                         // coordinates of the token whose loads are issued next (see hc_fwd_kernel)
    {
        const unsigned m0 = (unsigned)(blockIdx.x * TPB + tok);
        bb = m0 / (unsigned)a.N;
        nn = m0 % (unsigned)a.N;
    }
    const unsigned tstride = gridDim.x * TPB, sbb = tstride / (unsigned)a.N, snn = tstride % (unsigned)a.N;
    // slot roles of this lane (loop-invariant): slots 0 .. NB-1 = alpha pre-activation (s, t), NB .. NB+S-1 = beta of stream s
    using SM = SlotMap<NV>;
    const int slot = SM::slot(lane);
    const bool prim = SM::primary(lane);
    const bool is_a = prim && slot < NB, is_b = prim && slot >= NB && slot < NB + S;
    const int sl = is_a ? slot / (S + 1) : (is_b ? slot - NB : 0);
    const bool is_dx = slot < NB && slot % (S + 1) == 0;                     // <dx, R_s> slots (both lanes of a slot keep them consistent)
    const int sl0 = (slot < NB) ? slot / (S + 1) : 0;
    const int src_sr = SM::lane_of_dyn(O_SR + sl0), src_xr = SM::lane_of_dyn(O_XR + sl0);
    const int cl = lane < C::W ? lane : 0;                                   // this lane's entry of a coefficient record
    const int pre_idx = is_a ? C::AP + slot : C::BP + sl;
    // record addresses of this lane (see hcrec): its slot [dR, dR, q, q], its alpha pair; lanes that own neither write to the dump floats
    float* const recw = (GL && REC) ? reinterpret_cast<float*>(glsm + GL_REC_OFF) + wave * RECW4 : &hcrec_[(GL || !REC) ? 0 : wave][0];
    float* const rec_slot = recw + ((is_a || is_b) ? sl * RS + 4 * (is_a ? slot % (S + 1) : S + 1) : S * RS);
    float* const rec_alpha = recw + (lane < NB ? (lane / (S + 1)) * RS + 4 * (S + 2) + 2 * (lane % (S + 1)) : S * RS);
    auto issue_scalars = [&](auto& w, unsigned m_, const auto& a) {
        w.cf = w.cfp = w.pre = w.upb = w.ms = w.rnl = 0.f;
        if (WIDTH) {
            if constexpr (REC) w.rnl = *at_bytes(a.coef, (m_ * (unsigned)C::W + (unsigned)(C::RN + sl)) * 4u);         // 1 / |R_s| of this lane's stream
            w.cf = *at_bytes(a.coef, (m_ * (unsigned)C::W + (unsigned)cl) * 4u);       // 32-bit byte offsets (the launcher checks the sizes): SGPR base + VGPR offset
            w.pre = *at_bytes(a.coef, (m_ * (unsigned)C::W + (unsigned)pre_idx) * 4u);
            w.upb = *at_bytes(a.dbeta, (m_ * (unsigned)S + (unsigned)sl) * 4u);
            if (LNF) w.ms = (lane & 1) ? *at_bytes(a.rstd, m_ * 4u) : *at_bytes(a.mean, m_ * 4u);
        }
        if (DEPTH) w.cfp = *at_bytes(a.coef_prev, (m_ * (unsigned)C::W + (unsigned)cl) * 4u);
    };
    struct Tok { int m; int b, n; bool valid; };                             // m = b * N + n < 2^31 (checked by the launcher)
    auto next_tok = [&](int it_) {
        Tok t;
        t.m = it_ * TPB + tok;
        t.b = (int)bb; t.n = (int)nn;
        t.valid = it_ < niter && t.m < M;
        bb += sbb; nn += snn;
        if (nn >= (unsigned)a.N) { nn -= (unsigned)a.N; ++bb; }
        return t;
    };
    // one token's inputs as loaded.  cf / cfp: lane l = entry l of the token's coefficient record / of the previous branch's (every per-token
    // scalar is then a v_readlane away); pre / upb: this lane's pre-activation and (beta slots) upstream gradient; ms: lane 0 mean, lane 1 rstd.
    // With them NO load sits between the prefetch of the next token and the end of this one: the prefetch really stays in flight (vmcnt
    // retires in order: a late scalar load would force a wait for everything issued before it, i.e. for the whole prefetch).
    struct In { Raw4<RT> g[S], r[S]; float4 gb, rb, dx; uint2 dxn, ex, y; float cf, cfp, pre, upb, ms, rnl; const unsigned char* lds; };       // lds: GL only (this wave's slice of the current buffer)
    auto issue_pf = [&](In& w, const Tok& t) {
        const auto& a = *hc_bwd_kargs();                                           // (shadows the by-value parameter: see hc_bwd_kargs)
        const RT* const dRn = reinterpret_cast<const RT*>(a.dRn);
        const RT* const Rsv = reinterpret_cast<const RT*>(a.R);
#if ALM_HC_PROBE == 3 || ALM_HC_PROBE == 4
        const unsigned m_ = blockIdx.x, b_ = 0u, n_ = blockIdx.x, el = (unsigned)e0;
#else
        const unsigned m_ = t.valid ? (unsigned)t.m : 0u;
        const unsigned b_ = t.valid ? (unsigned)t.b : 0u, n_ = t.valid ? (unsigned)t.n : 0u, el = (unsigned)e0;         // PF: D == WPT * 256, every lane in range
#endif
        const unsigned uN = (unsigned)a.N, uD = (unsigned)a.D, sND32 = hc_sstride<unsigned>(uN, uD);
        constexpr unsigned RB = sizeof(RT);
        const unsigned tofs = (hc_rbase<unsigned>(b_, n_, (unsigned)S, uN, uD) + el) * RB;       // 32-bit BYTE offsets: the launcher takes this path only when every tensor is < 4 GB
        issue_scalars(w, m_, a);
        if constexpr (BC == 1) {
            w.gb = ld4(at_bytes(reinterpret_cast<const float*>(a.dRn), (m_ * uD + el) * 4u));
        } else {
#pragma unroll
            for (int t2 = 0; t2 < S; ++t2) ldraw(w.g[t2], at_bytes(dRn, tofs + (unsigned)t2 * sND32 * RB));
        }
        if (WIDTH) {
            if constexpr (BC == 2) {
                w.rb = ld4(at_bytes(reinterpret_cast<const float*>(a.R), (m_ * uD + el) * 4u));
            } else {
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) ldraw(w.r[s2], at_bytes(Rsv, tofs + (unsigned)s2 * sND32 * RB));
            }
            if (LNF) {
                w.dxn = *reinterpret_cast<const uint2*>(at_bytes(a.dxn, (m_ * (unsigned)a.lddxn + el) * 2u));
                w.ex = make_uint2(0u, 0u);
                if (a.extra) w.ex = *reinterpret_cast<const uint2*>(at_bytes(a.extra, (m_ * (unsigned)a.ldex + el) * 2u));
            } else {
                w.dx = ld4(at_bytes(a.dx, (m_ * (unsigned)a.lddx + el) * 4u));
            }
        }
        if (DEPTH) w.y = *reinterpret_cast<const uint2*>(at_bytes(a.y, (m_ * (unsigned)a.ldy + el) * 2u));
    };
    auto issue_plain = [&](In& w, const Tok& t) {
#pragma unroll
        for (int t2 = 0; t2 < S; ++t2) { zraw(w.g[t2]); zraw(w.r[t2]); }
        w.gb = w.rb = w.dx = z4;
        w.dxn = w.ex = w.y = make_uint2(0u, 0u);
        issue_scalars(w, t.valid ? (unsigned)t.m : 0u, a);
        if (t.valid && eok) {
            const long long tofs = hc_rbase<long long>(t.b, t.n, S, a.N, a.D) + e0;
            if (a.bcast) {
                w.gb = ld4(reinterpret_cast<const float*>(a.dRn) + (long long)t.m * a.D + e0);
            } else {
#pragma unroll
                for (int t2 = 0; t2 < S; ++t2) ldraw(w.g[t2], dRn + tofs + t2 * sND);
            }
            if (WIDTH) {
                if (a.r_bcast) {
                    w.rb = ld4(reinterpret_cast<const float*>(a.R) + (long long)t.m * a.D + e0);
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) ldraw(w.r[s2], Rsv + tofs + s2 * sND);
                }
                if (LNF) {
                    w.dxn = *reinterpret_cast<const uint2*>(a.dxn + (long long)t.m * a.lddxn + e0);
                    if (a.extra) w.ex = *reinterpret_cast<const uint2*>(a.extra + (long long)t.m * a.ldex + e0);
                } else {
                    w.dx = ld4(a.dx + (long long)t.m * a.lddx + e0);
                }
            }
            if (DEPTH) w.y = *reinterpret_cast<const uint2*>(a.y + (long long)t.m * a.ldy + e0);
        }
    };
    typedef __attribute__((address_space(1))) const char gcchar;
    typedef __attribute__((address_space(3))) void lds_void;
    // GL: DMA of one token's inputs into this wave's slice of an LDS buffer (`dstw`: wave-uniform).  Segment order: dRn_0..3 | R_0..3 | dxn | extra | y | (y)
    auto issue_gl = [&](unsigned char* dstw, const Tok& t) {
        const auto& a = *hc_bwd_kargs();
#if ALM_HC_PROBE == 3 || ALM_HC_PROBE == 4
        const unsigned m_ = blockIdx.x, b_ = 0u, n_ = blockIdx.x;
#else
        const unsigned m_ = t.valid ? (unsigned)t.m : 0u;
        const unsigned b_ = t.valid ? (unsigned)t.b : 0u, n_ = t.valid ? (unsigned)t.n : 0u;
#endif
        const unsigned half = (unsigned)lane >> 5, l32 = (unsigned)lane & 31u;
        const unsigned uN = (unsigned)a.N, uD = (unsigned)a.D, sND32 = hc_sstride<unsigned>(uN, uD);
        const unsigned el = (unsigned)(wv * 256) + l32 * 8u;                                        // 8 bf16 = 16 bytes per lane
        const unsigned tofs = (hc_rbase<unsigned>(b_, n_, (unsigned)S, uN, uD) + el) * 2u + half * sND32 * 2u;    // lanes 32-63: the next stream's row
        gcchar* const gd = (gcchar*)a.dRn;
        gcchar* const gr = (gcchar*)a.R;
#pragma unroll
        for (int j = 0; j < S / 2; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gd + (tofs + (unsigned)(2 * j) * sND32 * 2u)), (lds_void*)(dstw + j * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < S / 2; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gr + (tofs + (unsigned)(2 * j) * sND32 * 2u)), (lds_void*)(dstw + (S / 2 + j) * 1024), 16, 0, 0);
        }
        // every base pointer is fetched (scalar loads) BEFORE the per-lane selects and pinned: left to itself the compiler moves each load into the arm of the
        // select that uses it and branches around it -- with branches between the DMA instructions
        auto pin = [](const void* q) { unsigned long long v = (unsigned long long)q; asm volatile("" : "+s"(v)); return v; };
        {
            const unsigned long long bx = pin(a.dxn), bex = pin(a.extra), by = pin(a.y);
            const unsigned ox = (m_ * (unsigned)a.lddxn + el) * 2u, oe = (m_ * (unsigned)a.ldex + el) * 2u;
            const bool hx = half != 0u && bex != 0ull;
            const unsigned long long px = (hx ? bex : bx) + (unsigned long long)(hx ? oe : ox);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)px, (lds_void*)(dstw + S * 1024), 16, 0, 0);
            // y is the 11th and last 512-byte segment: lanes 0-31 only (an LDS-DMA writes M0 + 16 * lane for the ACTIVE lanes; the upper half would land in
            // the scalar records behind it)
            const unsigned long long py = by + (unsigned long long)((m_ * (unsigned)a.ldy + el) * 2u);
            if (half == 0u) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)py, (lds_void*)(dstw + (S + 1) * 1024), 16, 0, 0);
        }
        {
            const unsigned lc = (unsigned)lane < (unsigned)C::W ? (unsigned)lane : (unsigned)C::W - 1u;
            unsigned char* const sc = dstw + (2 * S + 3) * 512;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((gcchar*)a.coef + (m_ * (unsigned)C::W + lc) * 4u), (lds_void*)sc, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((gcchar*)a.coef_prev + (m_ * (unsigned)C::W + lc) * 4u), (lds_void*)(sc + 256), 4, 0, 0);
            const unsigned long long bdb = pin(a.dbeta), bmu = pin(a.mean), brs = pin(a.rstd);
            const unsigned long long pm = (lane < S ? bdb : (lane == S ? bmu : brs)) + (unsigned long long)(lane < S ? (m_ * (unsigned)S + (unsigned)lane) * 4u : m_ * 4u);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pm, (lds_void*)(sc + 512), 4, 0, 0);
        }
    };
    // ... and its read-back into the In record process() consumes
    auto fetch_gl = [&](In& w, const unsigned char* curw) {
        const unsigned lo = (unsigned)lane * 8u;
        if constexpr (sizeof(RT) == 2) {
#pragma unroll
            for (int t2 = 0; t2 < S; ++t2) {
                w.g[t2].v = *reinterpret_cast<const uint2*>(curw + t2 * 512 + lo);
                w.r[t2].v = *reinterpret_cast<const uint2*>(curw + (S + t2) * 512 + lo);
            }
        }
        w.dxn = *reinterpret_cast<const uint2*>(curw + 2 * S * 512 + lo);
        w.ex = *reinterpret_cast<const uint2*>(curw + (2 * S + 1) * 512 + lo);
        w.y = *reinterpret_cast<const uint2*>(curw + (2 * S + 2) * 512 + lo);
        const float* sc = reinterpret_cast<const float*>(curw + (2 * S + 3) * 512);
        w.cf = sc[cl];
        w.pre = sc[pre_idx];
        w.cfp = sc[64 + cl];
        w.upb = sc[128 + sl];
        w.ms = sc[128 + S + (lane & 1)];
        w.rnl = REC ? sc[C::RN + sl] : 0.f;
        w.gb = w.rb = w.dx = z4;
        w.lds = curw;
    };
    int par = 0;
    auto process_impl = [&](const In& w, const Tok& t, const auto& a) {
        RT* const dRo = reinterpret_cast<RT*>(a.dR);
#if ALM_HC_PROBE == 2 || ALM_HC_PROBE == 4
        const int m = PF ? (int)blockIdx.x : t.m;
#else
        const int m = t.m;
#endif
        // STRAIGHT (round 4): in the prefetching loop with one token per workgroup pass, process() only ever sees valid tokens (skipped ones are
        // prefetched, never processed) and every lane is in range, so every store of the token is UNCONDITIONAL.  That is not cosmetic: a store
        // inside a branch makes the number of vector-memory operations issued since a prefetch unknowable at compile time, the compiler then waits
        // with s_waitcnt vmcnt(0) -- i.e. for every store the wave has just issued -- before the first use of the next token's prefetched data,
        // once per loop iteration: the write latency of HBM in the critical path of every wave (found in the ISA after a 24 % cut of the VALU
        // instructions of this kernel changed nothing: DESIGN.md section 8.9).
        constexpr bool STRAIGHT = PF && TPB == 1;
        const bool valid = STRAIGHT ? true : t.valid;
#if ALM_HC_PROBE == 2 || ALM_HC_PROBE == 4
        const int b = PF ? 0 : t.b, n = PF ? (int)blockIdx.x : t.n;
#else
        const int b = t.b, n = t.n;
#endif
        const bool ld_ok = valid && eok;
        float4 g[S], r_c[S];
        float4 dx_c = z4, yv = z4, ex_c = z4;
        // DOT2: the <dRn_t, R_s> products are taken from the packed bf16 images (v_dot2_f32_bf16); dRn is then unpacked only in front of the element
        // loop that mixes it -- 8 raw registers live through the reduction instead of 16 unpacked ones
        constexpr bool DOT2 = WIDTH && PF && BC == 0 && sizeof(RT) == 2;
        if (DOT2) {
        } else if (PF) {
            // a skipped token re-read token 0 (finite data): its results are never stored, its coefficient gradients are zeroed through `up`
            // below, and the one running sum that takes the loaded data directly (dln) is protected by zeroing dxn
#pragma unroll
            for (int t2 = 0; t2 < S; ++t2) g[t2] = BC == 1 ? w.gb : unraw(w.g[t2]);
        } else if (a.bcast) {
#pragma unroll
            for (int t2 = 0; t2 < S; ++t2) g[t2] = w.gb;
        } else {
#pragma unroll
            for (int t2 = 0; t2 < S; ++t2) g[t2] = unraw(w.g[t2]);
        }
        if (WIDTH) {
            if (PF) {
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) r_c[s2] = BC == 2 ? w.rb : unraw(w.r[s2]);
            } else if (a.r_bcast) {
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) r_c[s2] = w.rb;
            } else {
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) r_c[s2] = unraw(w.r[s2]);
            }
            if (LNF) {
                dx_c = unraw((!PF || valid) ? w.dxn : make_uint2(0u, 0u));       // dxn (bf16) travels in dx_c
                ex_c = unraw((GL && !a.extra) ? make_uint2(0u, 0u) : w.ex);      // (GL fetched dxn in extra's place when there is no extra)
            } else {
                dx_c = w.dx;
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) r_c[s2] = z4;
        }
        if (DEPTH) yv = unraw(w.y);
        // consumers of one stream's gradient dR_s -- the store (stream tensor and / or the stream sum of the first branch) and the depth-connection
        // backward of the previous branch (dy += beta_p[s] dR_s, dbeta_p[s] = <dR_s, y>) -- run as soon as the stream is computed: only ONE stream's
        // output is live at a time (16 registers less than holding all S of them: the difference between two workgroups per CU and one)
        float4 dsum_acc = z4, dy_acc = z4;
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
        auto emit = [&](int s_, const float4& o_) {
            if (WIDTH && ld_ok) {
                if ((STRAIGHT && BC != 2) || a.dR) {                 // (the launcher takes the prefetching kernels with BC != 2 only when dR is given)
                    if constexpr (PF) stR(at_bytes(dRo, (hc_rbase<unsigned>((unsigned)b, (unsigned)n, (unsigned)S, (unsigned)a.N, (unsigned)a.D) + (unsigned)s_ * hc_sstride<unsigned>((unsigned)a.N, (unsigned)a.D) + (unsigned)e0) * (unsigned)sizeof(RT)), o_);
                    else stR(dRo + hc_rbase<long long>(b, n, S, a.N, a.D) + s_ * sND + e0, o_);
                }
            }
            if (WIDTH) { dsum_acc.x += o_.x; dsum_acc.y += o_.y; dsum_acc.z += o_.z; dsum_acc.w += o_.w; }
            if (DEPTH) {
                const float bt = lane_bcast(w.cfp, C::Bt + s_);
                dy_acc.x += bt * o_.x; dy_acc.y += bt * o_.y; dy_acc.z += bt * o_.z; dy_acc.w += bt * o_.w;
                v4[s_] = o_.x * yv.x + o_.y * yv.y + o_.z * yv.z + o_.w * yv.w;
            }
        };
#if ALM_HC_PROBE == 1
        if (PF && WIDTH && DEPTH) {
            const float sc = w.cf + w.cfp + w.pre + w.upb + w.ms;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 gs_ = DOT2 ? unraw(w.g[s]) : g[s];
                const float4 o_ = make_float4(gs_.x + r_c[s].x * sc + dx_c.x, gs_.y + r_c[s].y * sc + ex_c.y, gs_.z + r_c[s].z + yv.z, gs_.w + r_c[s].w);
                stR(at_bytes(dRo, ((((unsigned)b * (unsigned)S + (unsigned)s) * (unsigned)a.N + (unsigned)n) * (unsigned)a.D + (unsigned)e0) * (unsigned)sizeof(RT)), o_);
                dy_acc.x += o_.x; dy_acc.y += o_.y; dy_acc.z += o_.z; dy_acc.w += o_.w;
            }
            st4bf(at_bytes(a.dy, ((unsigned)m * (unsigned)a.lddy + (unsigned)e0) * 2u), dy_acc);
            *at_bytes(a.dbeta_out, ((unsigned)m * 4u + (unsigned)(lane & 3)) * 4u) = sc;
            return;
        }
#endif
        if (WIDTH) {
            float4 r[S];
#pragma unroll
            for (int s = 0; s < S; ++s) r[s] = r_c[s];
            auto cpv = [&](int k) { return lane_bcast(w.cf, k); };             // entry k of the token's coefficient record (k: compile-time)
            float4 dxv = dx_c;
            float4 xh = z4, gg = z4;                                        // LNF: xhat and g = dxn * gamma of this thread's 4 elements
            float rs = 0.f;
            float v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = 0.f;
            if (LNF) {
                const float mu = lane_bcast(w.ms, 0);
                rs = lane_bcast(w.ms, 1);
                float4 x = z4;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float a0 = cpv(C::A + s * (S + 1));
                    x.x += a0 * r[s].x; x.y += a0 * r[s].y; x.z += a0 * r[s].z; x.w += a0 * r[s].w;
                }
                xh = eok ? make_float4((x.x - mu) * rs, (x.y - mu) * rs, (x.z - mu) * rs, (x.w - mu) * rs) : z4;
                gg = make_float4(dx_c.x * lng.x, dx_c.y * lng.y, dx_c.z * lng.z, dx_c.w * lng.w);
                dln[0] += dx_c.x * xh.x; dln[1] += dx_c.y * xh.y; dln[2] += dx_c.z * xh.z; dln[3] += dx_c.w * xh.w;
                dxv = make_float4(rs * gg.x + ex_c.x, rs * gg.y + ex_c.y, rs * gg.z + ex_c.z, rs * gg.w + ex_c.w);      // u = rstd g + extra
                v[O_C1] = gg.x + gg.y + gg.z + gg.w;
                v[O_C2] = gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    v[O_SR + s] = r[s].x + r[s].y + r[s].z + r[s].w;
                    v[O_XR + s] = xh.x * r[s].x + xh.y * r[s].y + xh.z * r[s].z + xh.w * r[s].w;
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                v[s * (S + 1)] = dxv.x * r[s].x + dxv.y * r[s].y + dxv.z * r[s].z + dxv.w * r[s].w;
#pragma unroll
                for (int t = 0; t < S; ++t) {
                    if constexpr (DOT2) {
                        // <dRn_t, R_s> straight from the packed bf16 images: two v_dot2_f32_bf16 instead of a multiply + three FMAs (or their packed forms)
                        v[s * (S + 1) + t + 1] = dot2bf(w.g[t].v.y, w.r[s].v.y, dot2bf(w.g[t].v.x, w.r[s].v.x, 0.f));
                    } else {
                        v[s * (S + 1) + t + 1] = g[t].x * r[s].x + g[t].y * r[s].y + g[t].z * r[s].z + g[t].w * r[s].w;
                    }
                }
            }
            float da = bfly<NV>(v, lane);
            da = token_combine<WPT, NV, GL>(da, red_base + par * RED_F, tok, wv, lane);
            if (LNF) {
                // <dx, R_s> = <u, R_s> - rstd mean(g) sum R_s - rstd mean(g xhat) <xhat, R_s>; then the per-element dx itself
                const float invD = 1.f / (float)a.D;
                const float c1 = lane_bcast(da, SM::lane_of(O_C1)) * invD, c2 = lane_bcast(da, SM::lane_of(O_C2)) * invD;
                const float sr = __shfl(da, src_sr, 64), xr = __shfl(da, src_xr, 64);
                if (is_dx) da -= rs * (c1 * sr + c2 * xr);
                dxv = make_float4(rs * (gg.x - c1 - xh.x * c2) + ex_c.x, rs * (gg.y - c1 - xh.y * c2) + ex_c.y,
                                  rs * (gg.z - c1 - xh.z * c2) + ex_c.z, rs * (gg.w - c1 - xh.w * c2) + ex_c.w);
            }
            // slot (s, t) = (slot / (S+1), slot % (S+1)) for slot < NB; slots NB .. NB+S-1: the beta path of stream slot - NB  (roles: see above the loop)
            float pre = 0.f, up = 0.f;                                   // pre-activation and upstream gradient of this lane's coefficient
            if (valid) {
                if (is_a) { pre = w.pre; up = da; }
                else if (is_b) { pre = w.pre; up = w.upb; }
            }
            const float th = tanh_fast(pre);
            const float dpre = up * (is_a ? sa : sb) * (1.f - th * th);     // dap[s][t] (a slots) | dbp[s] (b slots)
            if (wv == 0) {
                if (is_a) { accA += up; accsa += up * th; }
                if (is_b) { accB += up; accsb += up * th; }
            }
            if constexpr (DOT2) {
                __builtin_amdgcn_sched_barrier(0);                        // (keeps the unpack HERE: see DOT2)
#pragma unroll
                for (int t2 = 0; t2 < S; ++t2) {
                    // GL: the packed image is read again from the LDS buffer -- no register carries dRn across the reduction
                    if constexpr (GL) g[t2] = unraw(*reinterpret_cast<const uint2*>(w.lds + t2 * 512 + lane * 8));
                    else g[t2] = unraw(w.g[t2]);
                }
            }
            if constexpr (REC) {
            // ---- the token's per-stream scalars -> this wave's LDS record (see hcrec), read back as broadcast splat pairs
            {
                const float dR = dpre * w.rnl;                             // dap[s][t] / |R_s|  (a slots) | dbp[s] / |R_s|  (b slots); 0 in lanes that own no slot
                const float q = dR * w.rnl * pre;                          // its share of <g_s, R_s> / |R_s|^2 = sum_t dap apre + dbp bpre, over |R_s|^2
                *reinterpret_cast<hc_f4*>(rec_slot) = hc_f4{dR, dR, q, q};
                *reinterpret_cast<hc_f2*>(rec_alpha) = hc_f2{w.cf, w.cf};   // lane l < NB holds alpha entry l of the coefficient record
            }
            hc_f2 dxv2[2] = {hc_f2{dxv.x, dxv.y}, hc_f2{dxv.z, dxv.w}};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float* rs_ = recw + s * RS;
                hc_f2 dRp[S + 2], qs = hc_f2{0.f, 0.f}, al[S + 1];
#pragma unroll
                for (int k = 0; k < S + 2; ++k) {
                    const hc_f4 t4 = *reinterpret_cast<const hc_f4*>(rs_ + 4 * k);
                    dRp[k] = hc_f2{t4[0], t4[1]};
                    qs += hc_f2{t4[2], t4[3]};
                }
#pragma unroll
                for (int k = 0; k < AF / 4; ++k) {
                    const hc_f4 a4 = *reinterpret_cast<const hc_f4*>(rs_ + 4 * (S + 2) + 4 * k);
                    al[2 * k] = hc_f2{a4[0], a4[1]};
                    if (2 * k + 1 < S + 1) al[2 * k + 1] = hc_f2{a4[2], a4[3]};
                }
                // dR_s = alpha[s][0] dx + (1 / |R_s|) (gs - <gs, R_s> R_s / |R_s|^2) + sum_t alpha[s][t+1] dRn_t   with gs = (sum_t dap W_t + dbp wb) (gamma + 1) sqrt(D):
                // the factors (gamma + 1) sqrt(D) sit in wa / wbv, 1 / |R_s| in dRp, <gs, R_s> / |R_s|^2 = qs (n_s . W = apre |R_s| ... see the forward)
                float4 out_s;
                float4 r_s = r[s];
                if constexpr (GL) r_s = unraw(*reinterpret_cast<const uint2*>(w.lds + (S + s) * 512 + lane * 8));
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const hc_f2 rv = cp ? hc_f2{r_s.z, r_s.w} : hc_f2{r_s.x, r_s.y};
                    hc_f2 o = al[0] * dxv2[cp];
                    o = f2fma(dRp[S + 1], hc_f2{wbv[2 * cp], wbv[2 * cp + 1]}, o);
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) o = f2fma(dRp[t], hc_f2{wa[t][2 * cp], wa[t][2 * cp + 1]}, o);
                    o = f2fma(-qs, rv, o);
#pragma unroll
                    for (int t = 0; t < S; ++t) o = f2fma(al[t + 1], cp ? hc_f2{g[t].z, g[t].w} : hc_f2{g[t].x, g[t].y}, o);
                    f4(out_s, 2 * cp) = o[0];
                    f4(out_s, 2 * cp + 1) = o[1];
                    // parameter-gradient sums: raw_a[t] += nhat dap[t], nhat = R_s sqrt(D) / |R_s| -- the sqrt(D) is applied once, when the partial row is written
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) {
                        const hc_f2 acc = f2fma(rv, dRp[t], hc_f2{rawa[t][2 * cp], rawa[t][2 * cp + 1]});
                        rawa[t][2 * cp] = acc[0];
                        rawa[t][2 * cp + 1] = acc[1];
                    }
                    const hc_f2 accb = f2fma(rv, dRp[S + 1], hc_f2{rawb[2 * cp], rawb[2 * cp + 1]});
                    rawb[2 * cp] = accb[0];
                    rawb[2 * cp + 1] = accb[1];
                }
                emit(s, out_s);
                // one stream's record live at a time: left alone the scheduler hoists all S x 9 broadcast reads to the top (another 100 VGPRs: one
                // workgroup per CU instead of two)
                __builtin_amdgcn_sched_barrier(0);
            }
            } else {
            // stream by stream: the 13 per-token scalars of stream s (alpha[s][.], dap[s][.], dbp, 1/|R_s|, <g_s, R_s>) are fetched right where they
            // are used, so that only one stream's worth of SGPRs is live at a time (all 52 at once spill)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float dap[S + 1], alpha[S + 1];
                const float rn = cpv(C::RN + s);
                const float dbp = lane_bcast(dpre, SM::lane_of(NB + s));
                float gd = dbp * cpv(C::BP + s);
#pragma unroll
                for (int t = 0; t < S + 1; ++t) {
                    dap[t] = lane_bcast(dpre, SM::lane_of(s * (S + 1) + t));
                    alpha[t] = cpv(C::A + s * (S + 1) + t);
                    gd += dap[t] * cpv(C::AP + s * (S + 1) + t);
                }
                // <g_s, R_s> = sum_t dap * apre / rn  (n_s . W = apre => sum_e W[e] (gamma+1) c R_s[e] = apre / rn); 1-ulp reciprocal of the uniform 1/|R_s|
                const float gdot = gd * __builtin_amdgcn_rcpf(rn);
                const float grr = gdot * rn * rn, rnc = rn * cD;
                float4 out_s;
                // GL: R_s comes back from the LDS buffer right here (2 registers, unpacked for this stream only) instead of living unpacked through the token
                float4 r_s = r[s];
                if constexpr (GL) r_s = unraw(*reinterpret_cast<const uint2*>(w.lds + (S + s) * 512 + lane * 8));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float rv = f4c(r_s, c);
                    float dn = dbp * wbv[c];
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) dn += dap[t] * wa[t][c];
                    const float nhat = rv * rnc;                           // n_s / (gamma + 1)
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) rawa[t][c] += nhat * dap[t];
                    rawb[c] += nhat * dbp;
                    const float gs = dn;                                   // ((gamma + 1) sqrt(D) sits in wa / wbv)
                    float o = alpha[0] * f4c(dxv, c) + rn * (gs - grr * rv);
#pragma unroll
                    for (int t = 0; t < S; ++t) o += alpha[t + 1] * f4c(g[t], c);
                    f4(out_s, c) = o;
                }
                emit(s, out_s);
            }
            }
            if (ld_ok && !(STRAIGHT && BC != 2) && a.dsum) {           // (... and no stream sum is asked for)
                const float ds = a.dsum_scale;
                float* dsp = PF ? at_bytes(a.dsum, ((unsigned)m * (unsigned)a.D + (unsigned)e0) * 4u) : a.dsum + (long long)m * a.D + e0;
                *reinterpret_cast<float4*>(dsp) = make_float4(dsum_acc.x * ds, dsum_acc.y * ds, dsum_acc.z * ds, dsum_acc.w * ds);
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) emit(s, g[s]);
        }
        if (DEPTH) {
            const float4 o = dy_acc;
            if (ld_ok) st4bf(PF ? at_bytes(a.dy, ((unsigned)m * (unsigned)a.lddy + (unsigned)e0) * 2u) : a.dy + (long long)m * a.lddy + e0, o);
            float db = bfly4(v4);                                           // every lane: total of slot bfly4_slot(lane)
            if (WPT > 1) {                                                  // parity-double-buffered: this may be the only barrier of the iteration
                float* rd = redd_base + par * REDD_F;
                if ((lane & 15) == 0) rd[(tok * WPT + wv) * 4 + bfly4_slot(lane)] = db;
#if ALM_HC_PROBE != 5                                                    // (probe 5: the depth connection's barrier left out -- timing only, dbeta wrong)
                if constexpr (GL) lds_barrier_raw();
                else __syncthreads();
#endif
                db = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WPT; ++w2) db += rd[(tok * WPT + w2) * 4 + (lane & 3)];
            } else {
                db = __shfl(db, bfly4_lane_of(lane & 3), 64);
            }
            if constexpr (STRAIGHT && S == 4) {
                // every lane of every wave holds the total of slot (lane & 3): all of them write it (same value, same address) -- an unconditional store
                *at_bytes(a.dbeta_out, ((unsigned)m * 4u + (unsigned)(lane & 3)) * 4u) = db;
            } else {
                if (valid && wv == 0 && lane < S) *(PF ? at_bytes(a.dbeta_out, ((unsigned)m * (unsigned)S + (unsigned)lane) * 4u) : a.dbeta_out + ((long long)m * S + lane)) = db;
            }
        }
        par ^= 1;
    };
    auto process = [&](const In& w, const Tok& t) {
        if constexpr (PF) process_impl(w, t, *hc_bwd_kargs());
        else process_impl(w, t, a);
    };
    if constexpr (GL) {
        unsigned char* const b0 = glsm + wave * GL_WAVE;
        unsigned char* const b1 = glsm + GL_BUF + wave * GL_WAVE;
        // one token: DMA of the NEXT token into `dst`, then the current one out of `cur` (distinct buffers; __restrict__ parameters -> alias scopes once inlined)
        // The compiler does not order these ds_reads behind the DMA that filled `cur` (it was issued in the previous call of this lambda), so the wait is
        // explicit and COUNTED: since that DMA group the wave has issued the previous token's 6 stores (S x dR, dy, dbeta: all unconditional, see
        // STRAIGHT) and the 9 DMA instructions of the group just issued = 15 younger operations; vmcnt retires in order, so "at most 15 outstanding" means
        // the older group has landed while the stores and the new prefetch stay in flight.  (More younger operations than counted would only make the wait
        // longer than needed, never too short; the first token is waited for in full before the loop.)
        static_assert(S == 4, "the counted wait below assumes S + 2 stores per token and 9 DMA instructions per group");
        auto gstep = [&](unsigned char* __restrict__ dst, const unsigned char* __restrict__ cur, const Tok& tn, const Tok& tc) {
            issue_gl(dst, tn);
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            In w;
            fetch_gl(w, cur);
            process(w, tc);
        };
        Tok ta = next_tok(blockIdx.x), tb = ta;
        issue_gl(b0, ta);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int it = blockIdx.x; it < niter; it += 2 * (int)gridDim.x) {
            tb = next_tok(it + gridDim.x);
            gstep(b1, b0, tb, ta);
            if (it + gridDim.x >= niter) break;
            ta = next_tok(it + 2 * (int)gridDim.x);
            gstep(b0, b1, ta, tb);
        }
    } else if constexpr (PF) {
        In wa, wb2;
        Tok ta = next_tok(blockIdx.x), tb = ta;
        issue_pf(wa, ta);
        for (int it = blockIdx.x; it < niter; it += 2 * (int)gridDim.x) {
            tb = next_tok(it + gridDim.x);
            issue_pf(wb2, tb);
            process(wa, ta);
            if (it + gridDim.x >= niter) break;
            ta = next_tok(it + 2 * (int)gridDim.x);
            issue_pf(wa, ta);
            process(wb2, tb);
        }
    } else {
        for (int it = blockIdx.x; it < niter; it += gridDim.x) {
            const Tok t = next_tok(it);
            In w;
            issue_plain(w, t);
            process(w, t);
        }
    }

    if (!WIDTH) return;
    // per-(block, token slot) partial rows; the token's WPT waves own disjoint element ranges
    const int P = a.D * (S + 3) + NB + S + 2;
    float* prow = a.partial + ((long long)blockIdx.x * TPB + tok) * P;
    const float rsc = REC ? cD : 1.f;                          // LDSREC accumulates R dap / |R| and applies the sqrt(D) of nhat here
    if (eok) {
#pragma unroll
        for (int t = 0; t < S + 1; ++t) *reinterpret_cast<float4*>(prow + (long long)t * a.D + e0) = make_float4(rawa[t][0] * rsc, rawa[t][1] * rsc, rawa[t][2] * rsc, rawa[t][3] * rsc);
        *reinterpret_cast<float4*>(prow + (long long)(S + 1) * a.D + e0) = make_float4(rawb[0] * rsc, rawb[1] * rsc, rawb[2] * rsc, rawb[3] * rsc);
        *reinterpret_cast<float4*>(prow + (long long)(S + 2) * a.D + e0) = make_float4(dln[0], dln[1], dln[2], dln[3]);
    }
    if (wv == 0) {
        float* q = prow + (long long)a.D * (S + 3);
        if (is_a) q[slot] = accA;
        if (is_b) q[slot] = accB;
        const float tsa = wave_sum(accsa), tsb = wave_sum(accsb);
        if (lane == 0) { q[NB + S] = tsa; q[NB + S + 1] = tsb; }
    }
}

// second stage of the hyper-connection parameter gradients: column sums of the partial rows (alm_colsum) -> the gradients,
// out layout (floats): dWa[D][S+1] | dwb[D] | dgamma[D] | dAa[S][S+1] | dBb[S] | dsa | dsb | dln[D] (LayerNorm weight, fused-LN mode)
template <int S>
__global__ __launch_bounds__(256) void hc_param_grads_kernel(const float* __restrict__ ws, int chunks, HcParams hp, float* __restrict__ out, int D) {
    constexpr int NB = S * (S + 1);
    const long long P = (long long)D * (S + 3) + NB + S + 2;              // floats per chunk row (alm_hc_partial_width)
    // `ws` holds `chunks` partial column sums (stage 1 of alm_colsum): summed here -- the former colsum_finish launch.  A workgroup owns 64 elements;
    // its 4 waves take the chunk rows k = wave, wave + 4, ... of all S + 3 columns (independent loads, all in flight together: the one-thread-per-
    // element form of this kernel was a 4-workgroup chain of dependent loads, 18 us for 0.5 MB) and meet in LDS.  Fixed order: deterministic.
    __shared__ float red[S + 3][4][64];
    const int el = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float part[S + 3];
#pragma unroll
    for (int c = 0; c < S + 3; ++c) part[c] = 0.f;
    if (e < D) {
        for (int k = kq; k < chunks; k += 4) {
#pragma unroll
            for (int c = 0; c < S + 3; ++c) part[c] += ws[k * P + (long long)c * D + e];
        }
    }
#pragma unroll
    for (int c = 0; c < S + 3; ++c) red[c][kq][el] = part[c];
    __syncthreads();
    if (kq == 0 && e < D) {
        float col[S + 3];
#pragma unroll
        for (int c = 0; c < S + 3; ++c) col[c] = (red[c][0][el] + red[c][1][el]) + (red[c][2][el] + red[c][3][el]);
        const float g1 = hp.hc_gamma[e] + 1.f;
        float dg = hp.wb[e] * col[S + 1];
#pragma unroll
        for (int t = 0; t < S + 1; ++t) {
            out[(long long)e * (S + 1) + t] = g1 * col[t];
            dg += hp.Wa[(long long)e * (S + 1) + t] * col[t];
        }
        out[(long long)D * (S + 1) + e] = g1 * col[S + 1];
        out[(long long)D * (S + 2) + e] = dg;
        out[(long long)D * (S + 3) + NB + S + 2 + e] = col[S + 2];
    }
    if (blockIdx.x == 0 && threadIdx.x < NB + S + 2) {
        float v = 0.f;
        for (int k = 0; k < chunks; ++k) v += ws[k * P + (long long)D * (S + 3) + threadIdx.x];
        out[(long long)D * (S + 3) + threadIdx.x] = v;
    }
}

// ---- the same for SEVERAL width connections in two launches (round 4): the fused backward of a 6-layer stack finishes 12 hyper-connection branches,
// each with a column-sum launch + a parameter-gradient launch of ~5 us -- 24 launches and as many dispatch gaps per step for 0.1 ms of work.  The
// deferred mode of core.stack_backward keeps the partial rows of every branch and finishes them together: stage 1 = the column sums of every
// problem's partial rows in HC_CHUNKS row chunks (grid.z = problem), stage 2 = hc_param_grads_kernel over (element block, problem).
constexpr int HC_FINISH_MAX = 16, HC_CHUNKS = 16;
struct HcFinish {
    const float* part[HC_FINISH_MAX];      // partial rows of problem z: [rows[z]][P]
    const float* gamma[HC_FINISH_MAX];
    const float* Wa[HC_FINISH_MAX];
    const float* wb[HC_FINISH_MAX];
    float* out[HC_FINISH_MAX];             // alm_hc_grads_width(S, D) floats each
    int rows[HC_FINISH_MAX];
    int nb;
};
__global__ __launch_bounds__(256) void hc_colsum_batched_kernel(HcFinish f, float* __restrict__ ws, int P) {
    __shared__ float red[4][64];
    const int z = blockIdx.z;
    const float* __restrict__ in = f.part[z];
    const int rows = f.rows[z], rpc = (rows + HC_CHUNKS - 1) / HC_CHUNKS;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < P) {
        int r = r0 + ry;
        for (; r + 12 < r1; r += 16) {
            s0 += in[(long long)r * P + c]; s1 += in[(long long)(r + 4) * P + c]; s2 += in[(long long)(r + 8) * P + c]; s3 += in[(long long)(r + 12) * P + c];
        }
        for (; r < r1; r += 4) s0 += in[(long long)r * P + c];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < P) ws[((long long)z * HC_CHUNKS + blockIdx.y) * P + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}
template <int S>
__global__ __launch_bounds__(256) void hc_param_grads_batched_kernel(HcFinish f, const float* __restrict__ ws, int D) {
    constexpr int NB = S * (S + 1);
    const long long P = (long long)D * (S + 3) + NB + S + 2;
    const int z = blockIdx.y;
    ws += (long long)z * HC_CHUNKS * P;
    float* __restrict__ out = f.out[z];
    __shared__ float red[S + 3][4][64];
    const int el = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float part[S + 3];
#pragma unroll
    for (int c = 0; c < S + 3; ++c) part[c] = 0.f;
    if (e < D) {
        for (int k = kq; k < HC_CHUNKS; k += 4) {
#pragma unroll
            for (int c = 0; c < S + 3; ++c) part[c] += ws[k * P + (long long)c * D + e];
        }
    }
#pragma unroll
    for (int c = 0; c < S + 3; ++c) red[c][kq][el] = part[c];
    __syncthreads();
    if (kq == 0 && e < D) {
        float col[S + 3];
#pragma unroll
        for (int c = 0; c < S + 3; ++c) col[c] = (red[c][0][el] + red[c][1][el]) + (red[c][2][el] + red[c][3][el]);
        const float g1 = f.gamma[z][e] + 1.f;
        float dg = f.wb[z][e] * col[S + 1];
#pragma unroll
        for (int t = 0; t < S + 1; ++t) {
            out[(long long)e * (S + 1) + t] = g1 * col[t];
            dg += f.Wa[z][(long long)e * (S + 1) + t] * col[t];
        }
        out[(long long)D * (S + 1) + e] = g1 * col[S + 1];
        out[(long long)D * (S + 2) + e] = dg;
        out[(long long)D * (S + 3) + NB + S + 2 + e] = col[S + 2];
    }
    if (blockIdx.x == 0 && threadIdx.x < NB + S + 2) {
        float v = 0.f;
        for (int k = 0; k < HC_CHUNKS; ++k) v += ws[k * P + (long long)D * (S + 3) + threadIdx.x];
        out[(long long)D * (S + 3) + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// stream expand / reduce  (reference audiolm_pytorch.py:524 / :551): R[b][s] = x[b]  ;  x[b] = sum_s R[b][s]
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void streams_expand_kernel(const float* __restrict__ x, float* __restrict__ R, int B, int S, long long nd4) {
    const long long total = (long long)B * nd4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / nd4, j = i % nd4;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        for (int s = 0; s < S; ++s) reinterpret_cast<float4*>(R)[(b * S + s) * nd4 + j] = v;
    }
}
__global__ __launch_bounds__(256) void streams_reduce_kernel(const float* __restrict__ R, float* __restrict__ x, int B, int S, long long nd4) {
    const long long total = (long long)B * nd4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / nd4, j = i % nd4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < S; ++s) {
            const float4 v = reinterpret_cast<const float4*>(R)[(b * S + s) * nd4 + j];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(x)[i] = a;
    }
}

// residual add for the single-stream configuration (num_residual_streams == 1): out = x + y (y bf16)
__global__ __launch_bounds__(256) void residual_add_kernel(const float* __restrict__ x, const bf16_t* __restrict__ y, long long ldy,
                                                           float* __restrict__ out, long long rows, int D) {
    const int d4 = D / 4;
    const long long total = rows * d4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d4;
        const int e = (int)(i % d4) * 4;
        float4 v = ld4(x + r * D + e);
        const float4 w = ld4bf(y + r * ldy + e);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        *reinterpret_cast<float4*>(out + r * D + e) = v;
    }
}

// out_bf16 = a (fp32) [+ b (fp32)]   -- gradient hand-off into a bf16 GEMM operand
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ out,
                                                          long long ldo, long long rows, int D) {
    const int d4 = D / 4;
    const long long total = rows * d4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d4;
        const int e = (int)(i % d4) * 4;
        float4 v = ld4(a + r * D + e);
        if (b) { const float4 w = ld4(b + r * D + e); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        st4bf(out + r * ldo + e, v);
    }
}

// out (fp32) = a (fp32) + b (fp32)
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n4,
                                                      float scale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(a)[i], w = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4((v.x + w.x) * scale, (v.y + w.y) * scale, (v.z + w.z) * scale, (v.w + w.w) * scale);
    }
}

// grid-stride kernels with ~10-30 tokens per workgroup: the grid must be EXACTLY the resident workgroup count (CUs x occupancy),
// otherwise the surplus workgroups form a second, partly empty round (measured: 1.5 rounds at 768 blocks vs 512 resident)
template <typename K>
int resident_blocks(K kernel) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess || occ < 1) occ = 2;
#if ALM_HC_PROBE
    if (const char* e = getenv("ALM_HC_PROBE_OCC")) occ = atoi(e) < occ ? atoi(e) : occ;      // (probe builds: fewer resident workgroups than the registers allow)
#endif
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return occ * cus;
}
int hc_grid(long long M, int tpb, int resident) {
    const long long niter = (M + tpb - 1) / tpb;
    return (int)(niter < resident ? niter : resident);
}
int hc_wpt(int D) { return D <= 256 ? 1 : (D <= 512 ? 2 : 4); }

template <typename RT, int S, int WPT, bool DEPTH, bool WIDTH, bool FINAL, bool PF>
void launch_fwd_p(const HcFwdArgs& a, hipStream_t st) {
    static int resident = 0;                                   // idempotent lazy query (same value from every thread)
    auto k = hc_fwd_kernel<RT, S, WPT, DEPTH, WIDTH, FINAL, PF>;
    if (!resident) resident = resident_blocks(k);
    hipLaunchKernelGGL(k, dim3(hc_grid((long long)a.B * a.N, 4 / WPT, resident)), dim3(256), 0, st, a);
}
template <typename RT, int S, int WPT, bool DEPTH, bool WIDTH, bool FINAL>
void launch_fwd_w(const HcFwdArgs& a, hipStream_t st) {
    if constexpr (sizeof(RT) == 2) {
        if (!a.rin_bcast && a.D == WPT * 256) return launch_fwd_p<RT, S, WPT, DEPTH, WIDTH, FINAL, true>(a, st);      // PF assumes no out-of-range lanes
    }
    launch_fwd_p<RT, S, WPT, DEPTH, WIDTH, FINAL, false>(a, st);
}
template <typename RT, int S, bool DEPTH, bool WIDTH, bool FINAL>
int launch_fwd(const HcFwdArgs& a, hipStream_t st) {
    const int wpt = hc_wpt(a.D);
    if (wpt == 1) launch_fwd_w<RT, S, 1, DEPTH, WIDTH, FINAL>(a, st);
    else if (wpt == 2) launch_fwd_w<RT, S, 2, DEPTH, WIDTH, FINAL>(a, st);
    else launch_fwd_w<RT, S, 4, DEPTH, WIDTH, FINAL>(a, st);
    return 0;
}
template <typename RT, int S>
int dispatch_fwd(const HcFwdArgs& a, int mode, hipStream_t st) {
    switch (mode) {
        case 1: return launch_fwd<RT, S, true, false, false>(a, st);
        case 2: return launch_fwd<RT, S, false, true, false>(a, st);
        case 3: return launch_fwd<RT, S, true, true, false>(a, st);
        case 5: return launch_fwd<RT, S, true, false, true>(a, st);
        default: return ALM_ERR_BAD_ARG;
    }
}

// upper bound of the workgroup count of the backward kernel (sizes the partial-row buffer; the launch may use fewer)
int hc_bwd_blocks(long long M, int D) { return hc_grid(M, 4 / hc_wpt(D), 256 * 4); }

// the prefetching variant needs more registers and may be resident in fewer copies: the partial-row buffer is sized for the larger of the
// two grids (a launch may then use fewer rows than alm_hc_partial_rows reported: the surplus rows are zeroed by the launch wrapper)
template <typename RT, int S, int WPT, bool WIDTH, bool DEPTH, bool LNF, bool PF, int BC = 0, bool GL = false>
int bwd_grid_p(long long M, int D) {
    static int resident = 0;
    if (!resident) resident = resident_blocks(hc_bwd_kernel<RT, S, WPT, WIDTH, DEPTH, LNF, PF, BC, GL>);
    const int cap = hc_bwd_blocks(M, D);
    const int grid = hc_grid(M, 4 / WPT, resident);
    return grid > cap ? cap : grid;
}
template <typename RT, int S, int WPT, bool WIDTH, bool DEPTH, bool LNF>
int bwd_grid_w(long long M, int D) {
    int g = bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, false>(M, D);
    if constexpr (sizeof(RT) == 2) {
        g = std::max(g, bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 0>(M, D));
        g = std::max(g, bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 1>(M, D));
        if constexpr (WIDTH) g = std::max(g, bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 2>(M, D));
        if constexpr (WIDTH && DEPTH && LNF && WPT == 4 && S == 4) g = std::max(g, bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 0, true>(M, D));
    }
    return g;
}
// ALM_HC_GL (default: see hc_gl_default): the LDS-DMA variant of the inner-branch backward (hc_bwd_kernel<..., GL = true>)
constexpr int hc_gl_default = 1;
int hc_gl_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ALM_HC_GL");
        v = e ? (atoi(e) != 0) : hc_gl_default;
    }
    return v;
}
template <typename RT, int S, bool WIDTH, bool DEPTH, bool LNF>
int bwd_grid(long long M, int D) {
    const int wpt = hc_wpt(D);
    if (wpt == 1) return bwd_grid_w<RT, S, 1, WIDTH, DEPTH, LNF>(M, D);
    if (wpt == 2) return bwd_grid_w<RT, S, 2, WIDTH, DEPTH, LNF>(M, D);
    return bwd_grid_w<RT, S, 4, WIDTH, DEPTH, LNF>(M, D);
}
template <typename RT, int S, int WPT, bool WIDTH, bool DEPTH, bool LNF>
void launch_bwd_w(const HcBwdArgs& a, hipStream_t st) {
    const long long M = (long long)a.B * a.N;
    const int rows_alloc = (4 / WPT) * bwd_grid_w<RT, S, WPT, WIDTH, DEPTH, LNF>(M, a.D);     // what alm_hc_partial_rows reported
    // prefetching variants (bf16 streams, every lane in range): bc 0 plain, 1 broadcast dRn, 2 broadcast R; both broadcast: the plain kernel
    int bc = -1;
    if constexpr (sizeof(RT) == 2) {
        // the prefetching kernels address with 32-bit byte offsets: every tensor below 4 GB (the largest: fp32 [M][D] / RT [B][S][N][D] / the coefficient records)
        const bool small = (long long)a.B * S * a.N * a.D * 4 < 0xffffffffLL && M * 64 * 4 < 0xffffffffLL && M * (long long)std::max(std::max(std::max(a.lddxn, a.ldex), std::max(a.ldy, a.lddy)), a.lddx) * 4 < 0xffffffffLL;
        if (small && a.D == WPT * 256 && !(a.bcast && a.r_bcast)) bc = a.bcast ? 1 : ((a.r_bcast && WIDTH) ? 2 : (a.r_bcast ? -1 : 0));
        // the prefetching kernels with stream-tensor R store dR unconditionally and never the stream sum (see STRAIGHT): any other request -> plain kernel
        if (WIDTH && (bc == 0 || bc == 1) && (!a.dR || a.dsum)) bc = -1;
        if constexpr (WIDTH && DEPTH && LNF && WPT == 4 && S == 4) {
            // the LDS-DMA variant moves 16-byte pieces: every row it reads must start on a 16-byte boundary (bases and row strides)
            const bool al16 = ((((uintptr_t)a.dRn | (uintptr_t)a.R | (uintptr_t)a.dxn | (uintptr_t)a.extra | (uintptr_t)a.y) & 15) == 0) &&
                              (((a.lddxn | a.ldex | a.ldy) & 7) == 0) && a.coef_prev && a.dbeta;
            if (bc == 0 && al16 && hc_gl_enabled()) bc = 3;
        }
    }
    int grid = bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, false>(M, a.D);
    if constexpr (sizeof(RT) == 2) {
        if (bc == 0) grid = bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 0>(M, a.D);
        if (bc == 1) grid = bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 1>(M, a.D);
        if constexpr (WIDTH) { if (bc == 2) grid = bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 2>(M, a.D); }
        if constexpr (WIDTH && DEPTH && LNF && WPT == 4 && S == 4) { if (bc == 3) grid = bwd_grid_p<RT, S, WPT, WIDTH, DEPTH, LNF, true, 0, true>(M, a.D); }
    }
    if (WIDTH && a.partial) {
        const int rows_used = (4 / WPT) * grid;
        const long long P = (long long)a.D * (S + 3) + S * (S + 1) + S + 2;
        // (alm_memset_zero is a kernel: a small hipMemsetAsync node of a captured hipGraph does not clear its buffer on replay, csrc/launchlist.hip)
        if (rows_used < rows_alloc) (void)alm_memset_zero(a.partial + (long long)rows_used * P, (long long)(rows_alloc - rows_used) * P * (long long)sizeof(float), (void*)st);
    }
    if constexpr (sizeof(RT) == 2) {
        if (bc == 0) { hipLaunchKernelGGL((hc_bwd_kernel<RT, S, WPT, WIDTH, DEPTH, LNF, true, 0>), dim3(grid), dim3(256), 0, st, a); return; }
        if (bc == 1) { hipLaunchKernelGGL((hc_bwd_kernel<RT, S, WPT, WIDTH, DEPTH, LNF, true, 1>), dim3(grid), dim3(256), 0, st, a); return; }
        if constexpr (WIDTH) {
            if (bc == 2) { hipLaunchKernelGGL((hc_bwd_kernel<RT, S, WPT, WIDTH, DEPTH, LNF, true, 2>), dim3(grid), dim3(256), 0, st, a); return; }
        }
        if constexpr (WIDTH && DEPTH && LNF && WPT == 4 && S == 4) {
            if (bc == 3) { hipLaunchKernelGGL((hc_bwd_kernel<RT, S, WPT, WIDTH, DEPTH, LNF, true, 0, true>), dim3(grid), dim3(256), 0, st, a); return; }
        }
    }
    hipLaunchKernelGGL((hc_bwd_kernel<RT, S, WPT, WIDTH, DEPTH, LNF, false>), dim3(grid), dim3(256), 0, st, a);
}
template <typename RT, int S, bool WIDTH, bool DEPTH, bool LNF>
int launch_bwd(const HcBwdArgs& a, hipStream_t st) {
    const int wpt = hc_wpt(a.D);
    if (wpt == 1) launch_bwd_w<RT, S, 1, WIDTH, DEPTH, LNF>(a, st);
    else if (wpt == 2) launch_bwd_w<RT, S, 2, WIDTH, DEPTH, LNF>(a, st);
    else launch_bwd_w<RT, S, 4, WIDTH, DEPTH, LNF>(a, st);
    return 0;
}
template <typename RT, int S>
int bwd_rows(int mode, bool lnf, long long M, int D) {
    const int tpb = 4 / hc_wpt(D);
    if (mode == 2) return tpb * (lnf ? bwd_grid<RT, S, true, false, true>(M, D) : bwd_grid<RT, S, true, false, false>(M, D));
    if (mode == 3) return tpb * (lnf ? bwd_grid<RT, S, true, true, true>(M, D) : bwd_grid<RT, S, true, true, false>(M, D));
    return 0;
}
template <typename RT, int S>
int dispatch_bwd(const HcBwdArgs& a, int mode, bool lnf, hipStream_t st) {
    switch (mode) {
        case 1: return launch_bwd<RT, S, false, true, false>(a, st);
        case 2: return lnf ? launch_bwd<RT, S, true, false, true>(a, st) : launch_bwd<RT, S, true, false, false>(a, st);
        case 3: return lnf ? launch_bwd<RT, S, true, true, true>(a, st) : launch_bwd<RT, S, true, true, false>(a, st);
        default: return ALM_ERR_BAD_ARG;
    }
}
template <typename RT>
int bwd_rows_s(int mode, bool lnf, int S, long long M, int D) {
    if (S == 2) return bwd_rows<RT, 2>(mode, lnf, M, D);
    if (S == 3) return bwd_rows<RT, 3>(mode, lnf, M, D);
    if (S == 4) return bwd_rows<RT, 4>(mode, lnf, M, D);
    return 0;
}
template <typename RT>
int dispatch_fwd_s(const HcFwdArgs& a, int S, int mode, hipStream_t st) {
    if (S == 2) return dispatch_fwd<RT, 2>(a, mode, st);
    if (S == 3) return dispatch_fwd<RT, 3>(a, mode, st);
    if (S == 4) return dispatch_fwd<RT, 4>(a, mode, st);
    return ALM_ERR_UNSUPPORTED;
}
template <typename RT>
int dispatch_bwd_s(const HcBwdArgs& a, int S, int mode, bool lnf, hipStream_t st) {
    if (S == 2) return dispatch_bwd<RT, 2>(a, mode, lnf, st);
    if (S == 3) return dispatch_bwd<RT, 3>(a, mode, lnf, st);
    if (S == 4) return dispatch_bwd<RT, 4>(a, mode, lnf, st);
    return ALM_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int alm_hc_coef_width(int S) { return 2 * S * (S + 1) + 3 * S; }
extern "C" int alm_hc_partial_width(int S, int D) { return D * (S + 3) + S * (S + 1) + S + 2; }
extern "C" int alm_hc_grads_width(int S, int D) { return D * (S + 4) + S * (S + 1) + S + 2; }
/* number of partial rows alm_hc_bwd writes for (mode, fused-LayerNorm or not, stream storage type): one per RESIDENT workgroup and token slot */
extern "C" int alm_hc_partial_rows(int mode, int fused_ln, int r_bf16, int S, long long tokens, int D) {
    return r_bf16 ? bwd_rows_s<bf16_t>(mode, fused_ln != 0, S, tokens, D) : bwd_rows_s<float>(mode, fused_ln != 0, S, tokens, D);
}

// mode: 1 = depth connection only (-> R_out), 2 = width connection only, 3 = depth (previous branch) + width (next branch) fused,
//       5 = depth + stream sum + final LayerNorm (-> xs_out fp32, xn_out bf16 or xn32_out fp32, mean, rstd; R_out is not written)
// r_bf16: R_in (unless rin_bcast) / R_out hold bf16 instead of fp32.
extern "C" int alm_hc_fwd(const void* R_in, int rin_bcast, int r_bf16, const void* y_prev, long long ldy, const float* coef_prev, void* R_out,
                          const float* hc_gamma, const float* Wa, const float* sa, const float* Aa, const float* wb, const float* sb,
                          const float* Bb, const float* ln_gamma, void* x_out, long long ldx, void* xn_out, long long ldxn, float* xn32_out,
                          float* mean, float* rstd, float* coef, float* xs_out, int mode, int B, int S, int N, int D, void* stream) {
    if ((D & 3) || D > 1024 || (ldx & 3) || (ldxn & 3) || (ldy & 3)) return ALM_ERR_BAD_ARG;
    if ((mode & 1) && (!y_prev || !coef_prev || (!(mode & 4) && !R_out))) return ALM_ERR_BAD_ARG;
    if ((mode & 2) && (!hc_gamma || !Wa || !sa || !Aa || !wb || !sb || !Bb || !ln_gamma || !xn_out || !mean || !rstd || !coef)) return ALM_ERR_BAD_ARG;
    if ((mode & 4) && (!ln_gamma || !(xn_out || xn32_out) || !mean || !rstd || !xs_out)) return ALM_ERR_BAD_ARG;
    if (((ldx | ldxn | ldy) >> 31) || (long long)B * N >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;             // row strides / token index travel as int
    HcFwdArgs a{R_in, rin_bcast, (const bf16_t*)y_prev, (int)ldy, coef_prev, R_out, HcParams{hc_gamma, Wa, sa, Aa, wb, sb, Bb}, ln_gamma,
                (bf16_t*)x_out, (int)ldx, (bf16_t*)xn_out, (int)ldxn, mean, rstd, coef, xs_out, xn32_out, B, N, D};
    const int rc = r_bf16 ? dispatch_fwd_s<bf16_t>(a, S, mode, (hipStream_t)stream) : dispatch_fwd_s<float>(a, S, mode, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// mode: 1 = depth-connection backward only (dy, dbeta_out from dRn), 2 = width-connection backward only (dR, partial),
//       3 = width backward of branch k+1 followed by the depth backward of branch k on the freshly computed dR.
// dRn_bcast != 0: dRn is fp32 [B*N][D] and stands for all S streams (the gradient of the final stream sum, audiolm_pytorch.py:551).
// r_bf16: dRn (unless dRn_bcast), R (unless r_bcast) and dR hold bf16 instead of fp32.
// partial: [alm_hc_partial_rows(...)][alm_hc_partial_width(S, D)] floats -> alm_colsum -> alm_hc_param_grads.
extern "C" int alm_hc_bwd(const void* dRn, int dRn_bcast, int r_bf16, const float* dx, long long lddx, const void* dxn, long long lddxn, const void* extra,
                          long long ldex, const float* mean, const float* rstd, const float* ln_gamma, const void* R, int r_bcast,
                          const float* coef, const float* dbeta, const float* hc_gamma, const float* Wa, const float* sa, const float* wb,
                          const float* sb, void* dR, float* dsum, float dsum_scale, float* partial, const void* y_prev, long long ldy, const float* coef_prev, void* dy,
                          long long lddy, float* dbeta_out, int mode, int B, int S, int N, int D, void* stream) {
    if ((D & 3) || D > 1024 || (lddx & 3) || (lddxn & 3) || (ldex & 3) || (ldy & 3) || (lddy & 3) || !dRn) return ALM_ERR_BAD_ARG;
    const bool lnf = dxn != nullptr;
    if ((mode & 2) && (!R || !coef || !dbeta || !hc_gamma || !Wa || !sa || !wb || !sb || !(dR || dsum) || !partial)) return ALM_ERR_BAD_ARG;
    if ((mode & 1) && (mode & 2) && !dR) return ALM_ERR_BAD_ARG;
    if ((mode & 2) && (lnf ? (!mean || !rstd || !ln_gamma || dx != nullptr) : !dx)) return ALM_ERR_BAD_ARG;
    if ((mode & 1) && (!y_prev || !coef_prev || !dy || !dbeta_out)) return ALM_ERR_BAD_ARG;
    if (((lddx | lddxn | ldex | ldy | lddy) >> 31) || (long long)B * N >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;   // row strides / token index travel as int
    HcBwdArgs a{dRn, dRn_bcast, dx, (int)lddx, (const bf16_t*)dxn, (int)lddxn, (const bf16_t*)extra, (int)ldex, mean, rstd, ln_gamma, R, r_bcast, coef, dbeta,
                HcParams{hc_gamma, Wa, sa, nullptr, wb, sb, nullptr}, dR, dsum, dsum_scale, partial, (const bf16_t*)y_prev, (int)ldy, coef_prev, (bf16_t*)dy, (int)lddy, dbeta_out,
                B, N, D};
    const int rc = r_bf16 ? dispatch_bwd_s<bf16_t>(a, S, mode, lnf, (hipStream_t)stream) : dispatch_bwd_s<float>(a, S, mode, lnf, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// sums: column sums (alm_colsum) of the partial rows.  out: alm_hc_grads_width(S, D) floats:
// dWa[D][S+1] | dwb[D] | dgamma[D] | dAa[S][S+1] | dBb[S] | dsa | dsb | dln[D]
extern "C" int alm_hc_param_grads(const float* sums, int chunks, const float* hc_gamma, const float* Wa, const float* wb, float* out, int S, int D,
                                  void* stream) {
    if (chunks < 1) return ALM_ERR_BAD_ARG;
    HcParams hp{hc_gamma, Wa, nullptr, nullptr, wb, nullptr, nullptr};
    const int grid = (D + 63) / 64;
    if (S == 2) hipLaunchKernelGGL(hc_param_grads_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sums, chunks, hp, out, D);
    else if (S == 3) hipLaunchKernelGGL(hc_param_grads_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sums, chunks, hp, out, D);
    else if (S == 4) hipLaunchKernelGGL(hc_param_grads_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sums, chunks, hp, out, D);
    else return ALM_ERR_UNSUPPORTED;
    ALM_LAUNCH_CHECK();
    return 0;
}

/* the finish of nb <= 16 width connections in two launches (see hc_colsum_batched_kernel): parts[z] = the partial rows alm_hc_bwd wrote for problem z
 * ([rows[z]][alm_hc_partial_width]), gammas / Was / wbs its hyper-connection parameters, outs[z] = alm_hc_grads_width(S, D) floats (layout as
 * alm_hc_param_grads); ws: nb * 16 * alm_hc_partial_width(S, D) floats of scratch.  The pointer arrays are HOST arrays (they travel in the kernel arguments). */
extern "C" int alm_hc_param_grads_batched(const float* const* parts, const int* rows, int nb, const float* const* gammas, const float* const* Was,
                                          const float* const* wbs, float* ws, float* const* outs, int S, int D, void* stream) {
    if (nb <= 0) return 0;
    if (nb > HC_FINISH_MAX || !parts || !rows || !gammas || !Was || !wbs || !ws || !outs) return ALM_ERR_BAD_ARG;
    HcFinish f{};
    f.nb = nb;
    for (int z = 0; z < nb; ++z) {
        if (rows[z] < 1 || !parts[z] || !outs[z]) return ALM_ERR_BAD_ARG;
        f.part[z] = parts[z]; f.rows[z] = rows[z]; f.gamma[z] = gammas[z]; f.Wa[z] = Was[z]; f.wb[z] = wbs[z]; f.out[z] = outs[z];
    }
    const int P = alm_hc_partial_width(S, D);
    hipLaunchKernelGGL(hc_colsum_batched_kernel, dim3((P + 63) / 64, HC_CHUNKS, nb), dim3(256), 0, (hipStream_t)stream, f, ws, P);
    const dim3 grid((D + 63) / 64, nb);
    if (S == 2) hipLaunchKernelGGL(hc_param_grads_batched_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, f, (const float*)ws, D);
    else if (S == 3) hipLaunchKernelGGL(hc_param_grads_batched_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, f, (const float*)ws, D);
    else if (S == 4) hipLaunchKernelGGL(hc_param_grads_batched_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, f, (const float*)ws, D);
    else return ALM_ERR_UNSUPPORTED;
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_streams_expand(const float* x, float* R, int B, int S, long long nd, void* stream) {
    if (nd & 3) return ALM_ERR_BAD_ARG;
    const long long total = (long long)B * (nd / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(streams_expand_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, R, B, S, nd / 4);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_streams_reduce(const float* R, float* x, int B, int S, long long nd, void* stream) {
    if (nd & 3) return ALM_ERR_BAD_ARG;
    const long long total = (long long)B * (nd / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(streams_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, R, x, B, S, nd / 4);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_residual_add(const float* x, const void* y, long long ldy, float* out, long long rows, int D, void* stream) {
    if ((D & 3) || (ldy & 3)) return ALM_ERR_BAD_ARG;
    const long long total = rows * (D / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(residual_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)y, ldy, out, rows, D);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_f32_to_bf16(const float* a, const float* b, void* out, long long ldo, long long rows, int D, void* stream) {
    if ((D & 3) || (ldo & 3)) return ALM_ERR_BAD_ARG;
    const long long total = rows * (D / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, (bf16_t*)out, ldo, rows, D);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_add_f32(const float* a, const float* b, float* out, long long n, float scale, void* stream) {
    if (n & 3) return ALM_ERR_BAD_ARG;
    const long long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, n4, scale);
    ALM_LAUNCH_CHECK();
    return 0;
}
