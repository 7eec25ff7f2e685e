// Attention over a SHORT, always-visible key set (gfx950): the conditioning paths of the reference's Attention.forward
// (audiolm_pytorch.py:307-406):
//   * cross-attention layers (Transformer(cross_attend=True), :450): keys / values = [null_kv | to_kv(context_norm(text embeds))], non-causal,
//     key mask = [True | context_mask]  (:325, :372-388; Attend math path attend.py:98-146 with causal=False)
//   * `cond_as_self_attn_prefix` (:330-345): the text embeds are PREPENDED to the causal self-attention keys -- every query sees all of them.
// These key sets are tens to a few hundred positions long and shared by all heads (MQA), so the scores are small dense matrices:
//     S_e [B][N*H][Me] = Q K_e^T      P_e = softmax part      O_e = P_e V_e         (all three contractions run on the bf16 MFMA GEMM, gemm.hip)
// and only the row-wise softmax pieces live here.  A causal self-attention part over the sequence itself (flash kernels, attention.hip)
// is merged through its log-sum-exp:  lse = logaddexp(lse_self, lse_e),  O = exp(lse_self - lse) O_self + P_e V_e  with P_e = exp(S_e - lse).
// Backward: with the JOINT lse and the JOINT output O (delta = rowsum(dO o O)), dS_e = P_e o (dP_e - delta) * scale is exact for the extra
// keys, and the flash backward kernels given the same lse / O are exact for the self keys.
// Row r = (b * N + n) * H + h  <->  statistics index (b * H + h) * N + n  (the flash kernels' [B][H][N] layout).
//   * the reference's MATH path with an arbitrary DENSE `attn_bias` tensor (attend.py:98-146: sim = q k^T * scale + attn_bias, key mask, causal
//     triu(j - i + 1), softmax, attn v): the same three GEMMs over the sequence's own keys with `bias` [H][N][ldbias] added to the scaled scores
//     and the causal rule "key e visible to query n iff e <= n + causal_off" (causal_off = Me - N; INT_MAX: not causal) -- O(N^2) memory like
//     the reference's math path; the structured biases of the model family never take it (attention.hip indexes their table in place).
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

// one wave per row.  S fp32 [rows][ldS] (first Me columns valid), emask uint8 [B][Me] (1 = attend) or NULL, lse_self [B][H][N] or NULL
// -> P bf16 [rows][ldP] (columns >= Me zero), lse_tot [B][H][N], fself [rows] = exp(lse_self - lse_tot) (0 without a self part)
__global__ __launch_bounds__(256) void extra_softmax_fwd_kernel(const float* __restrict__ S, long long ldS, const uint8_t* __restrict__ emask,
                                                                const float* __restrict__ lse_self, float scale, bf16_t* __restrict__ P,
                                                                long long ldP, float* __restrict__ lse_tot, float* __restrict__ fself,
                                                                const float* __restrict__ bias, long long ldbias, int causal_off,
                                                                int B, int N, int H, int Me) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)B * N * H;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const int h = (int)(r % H);
        const long long bn = r / H;
        const int n = (int)(bn % N), b = (int)(bn / N);
        const long long si = ((long long)b * H + h) * N + n;
        const float* sp = S + r * ldS;
        const uint8_t* mp = emask ? emask + (long long)b * Me : nullptr;
        const float* bp = bias ? bias + ((long long)h * N + n) * ldbias : nullptr;
        const int elast = (causal_off == 0x7fffffff) ? Me - 1 : min(Me - 1, n + causal_off);     // last visible key of this query
        float mx = -INFINITY;
        for (int e = lane; e <= elast; e += 64)
            if (!mp || mp[e]) mx = fmaxf(mx, sp[e] * scale + (bp ? bp[e] : 0.f));
        mx = wave_max(mx);
        float sum = 0.f;
        if (mx != -INFINITY)
            for (int e = lane; e <= elast; e += 64)
                if (!mp || mp[e]) sum += __expf(sp[e] * scale + (bp ? bp[e] : 0.f) - mx);
        sum = wave_sum(sum);
        const float lse_e = (mx == -INFINITY) ? -INFINITY : mx + logf(sum);
        const float ls = lse_self ? lse_self[si] : -INFINITY;
        float lt;                                                            // logaddexp(ls, lse_e)
        if (ls == -INFINITY) lt = lse_e;
        else if (lse_e == -INFINITY) lt = ls;
        else { const float hi = fmaxf(ls, lse_e), lo = fminf(ls, lse_e); lt = hi + log1pf(__expf(lo - hi)); }
        bf16_t* pp = P + r * ldP;
        for (int e = lane; e < (int)ldP; e += 64) {
            float v = 0.f;
            if (e <= elast && (!mp || mp[e]) && lt != -INFINITY) v = __expf(sp[e] * scale + (bp ? bp[e] : 0.f) - lt);
            pp[e] = f2bf(v);
        }
        if (lane == 0) {
            lse_tot[si] = lt;
            if (fself) fself[r] = (ls == -INFINITY || lt == -INFINITY) ? 0.f : __expf(ls - lt);
        }
    }
}

// O[(b n)][h * dh + d] = fself[r] * O_self + O_e   (O_self / fself may be NULL: cross-attention has no self part); r = (b n) * H + h
__global__ __launch_bounds__(256) void attn_combine_kernel(const bf16_t* __restrict__ o_self, long long ldos, const float* __restrict__ fself,
                                                           const float* __restrict__ o_e, bf16_t* __restrict__ out, long long ldo, long long tokens,
                                                           int H, int dh) {
    const int d4 = dh / 4;
    const long long total = tokens * H * d4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % d4) * 4;
        const long long r = i / d4;                                          // (token, head)
        const long long tok = r / H;
        const int h = (int)(r % H);
        float4 v = *reinterpret_cast<const float4*>(o_e + r * dh + c);
        if (o_self) {
            const float f = fself[r];
            const uint2 u = *reinterpret_cast<const uint2*>(o_self + tok * ldos + h * dh + c);
            v.x += f * __uint_as_float(u.x << 16); v.y += f * __uint_as_float(u.x & 0xffff0000u);
            v.z += f * __uint_as_float(u.y << 16); v.w += f * __uint_as_float(u.y & 0xffff0000u);
        }
        *reinterpret_cast<uint2*>(out + tok * ldo + h * dh + c) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
}

// dS[r][e] = P[r][e] * (dP[r][e] + ndelta[b][h][n]) * scale    (bf16; pad columns e >= Me -> 0)
__global__ __launch_bounds__(256) void extra_softmax_bwd_kernel(const bf16_t* __restrict__ P, long long ldP, const float* __restrict__ dP, long long lddP,
                                                                const float* __restrict__ ndelta, float scale, bf16_t* __restrict__ dS, long long lddS,
                                                                int Me, int B, int N, int H) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)B * N * H;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const int h = (int)(r % H);
        const long long bn = r / H;
        const int n = (int)(bn % N), b = (int)(bn / N);
        const float nd = ndelta[((long long)b * H + h) * N + n];
        for (int e = lane; e < (int)ldP; e += 64)                            // pad columns: dP there was never written
            dS[r * lddS + e] = (e < Me) ? f2bf(bf2f(P[r * ldP + e]) * (dP[r * lddP + e] + nd) * scale) : (bf16_t)0;
    }
}

// ndelta[b][h][n] = -sum_d dO * O   (the flash backward's own prologue, exported for the pure cross-attention case)
__global__ __launch_bounds__(256) void xattn_delta_kernel(const bf16_t* __restrict__ o, long long ldo, const bf16_t* __restrict__ dout, long long lddo,
                                                          float* __restrict__ ndelta, int B, int N, int H, int dh) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)B * N * H;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const int h = (int)(r % H);
        const long long bn = r / H;
        const int n = (int)(bn % N), b = (int)(bn / N);
        float s = 0.f;
        for (int d = lane; d < dh; d += 64) s += bf2f(o[bn * ldo + h * dh + d]) * bf2f(dout[bn * lddo + h * dh + d]);
        s = wave_sum(s);
        if (lane == 0) ndelta[((long long)b * H + h) * N + n] = -s;
    }
}

// dbias[h][n][e] = sum_b P[r(b, n, h)][e] * (dP[r][e] + ndelta[b][h][n])      (gradient of the dense attn_bias: the un-scaled dS summed over the batch;
// one wave per (n, h), fixed summation order: deterministic)
__global__ __launch_bounds__(256) void xattn_dbias_kernel(const bf16_t* __restrict__ P, long long ldP, const float* __restrict__ dP, long long lddP,
                                                          const float* __restrict__ ndelta, float* __restrict__ dbias, long long lddb, int Me, int B, int N,
                                                          int H) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)N * H;
    for (long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); q < rows; q += (long long)gridDim.x * 4) {
        const int h = (int)(q % H), n = (int)(q / H);
        for (int e = lane; e < Me; e += 64) {
            float acc = 0.f;
            for (int b = 0; b < B; ++b) {
                const long long r = ((long long)b * N + n) * H + h;
                acc += bf2f(P[r * ldP + e]) * (dP[r * lddP + e] + ndelta[((long long)b * H + h) * N + n]);
            }
            dbias[((long long)h * N + n) * lddb + e] = acc;
        }
    }
}

int rows_grid(long long rows) { const long long g = (rows + 3) / 4; return (int)(g < 16384 ? (g < 1 ? 1 : g) : 16384); }

}  // namespace

extern "C" int alm_xattn_softmax_fwd(const float* S, long long ldS, const unsigned char* emask, const float* lse_self, float scale, void* P,
                                     long long ldP, float* lse_tot, float* fself, const float* bias, long long ldbias, int causal_off, int B, int N,
                                     int H, int Me, void* stream) {
    if (Me < 1 || ldS < Me || ldP < Me || !S || !P || !lse_tot || (bias && ldbias < Me)) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(extra_softmax_fwd_kernel, dim3(rows_grid((long long)B * N * H)), dim3(256), 0, (hipStream_t)stream, S, ldS, emask, lse_self,
                       scale, (bf16_t*)P, ldP, lse_tot, fself, bias, ldbias, causal_off, B, N, H, Me);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_xattn_combine(const void* o_self, long long ldos, const float* fself, const float* o_e, void* out, long long ldo, long long tokens,
                                 int H, int dim_head, void* stream) {
    if ((dim_head & 3) || (ldo & 3) || (o_self && ((ldos & 3) || !fself))) return ALM_ERR_BAD_ARG;
    const long long total = tokens * H * (dim_head / 4);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(attn_combine_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o_self, ldos, fself, o_e,
                       (bf16_t*)out, ldo, tokens, H, dim_head);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_xattn_softmax_bwd(const void* P, long long ldP, const float* dP, long long lddP, const float* ndelta, float scale, void* dS,
                                     long long lddS, int Me, int B, int N, int H, void* stream) {
    if (lddP < ldP || lddS < ldP || Me > ldP) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(extra_softmax_bwd_kernel, dim3(rows_grid((long long)B * N * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P, ldP, dP,
                       lddP, ndelta, scale, (bf16_t*)dS, lddS, Me, B, N, H);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_xattn_delta(const void* o, long long ldo, const void* dout, long long lddo, float* ndelta, int B, int N, int H, int dim_head,
                               void* stream) {
    hipLaunchKernelGGL(xattn_delta_kernel, dim3(rows_grid((long long)B * N * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, ldo,
                       (const bf16_t*)dout, lddo, ndelta, B, N, H, dim_head);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_xattn_dbias(const void* P, long long ldP, const float* dP, long long lddP, const float* ndelta, float* dbias, long long lddb,
                               int Me, int B, int N, int H, void* stream) {
    if (Me < 1 || ldP < Me || lddP < Me || lddb < Me) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(xattn_dbias_kernel, dim3(rows_grid((long long)N * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P, ldP, dP, lddP, ndelta,
                       dbias, lddb, Me, B, N, H);
    ALM_LAUNCH_CHECK();
    return 0;
}
