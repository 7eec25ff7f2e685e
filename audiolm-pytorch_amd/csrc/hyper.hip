// Hyper-connection residual-stream mixing around every attention / feed-forward branch (gfx950, HBM-bound).
//
// Replaces the third-party `hyper_connections` modules the reference wraps each block in
// (audiolm_pytorch.py:24, 446, 452-454, 524, 551; arithmetic restated in oracle/hyper_connections_restated.py):
//   width : n_s = normalize(R_s) * sqrt(D) * (g + 1);  alpha = tanh(n Wa) * sa + Aa;  beta = tanh(n wb) * sb + Bb
//           x = sum_s alpha[s][0] R_s   (branch input; the branch pre-LayerNorm audiolm_pytorch.py:347 / :254 is fused here)
//   depth : R'_t = sum_s alpha[s][t+1] R_s + beta[t] * y
// Residual streams R: fp32 [B][S][N][D]  (the reference's '(b s) n d').
//
// One wave64 owns one token: its S x D residual slab lives in registers (16-B coalesced loads, 4 x float4 per stream for
// D = 1024), every reduction (S norms, S*(S+2) dot products, LayerNorm statistics) is a wave shuffle butterfly: no LDS, no
// block barrier.  Parameter gradients are accumulated in registers over a grid-stride token loop, reduced over the 4 waves
// of a block through LDS and written as per-block partial rows (second stage: alm_colsum).
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr float NORM_EPS = 1e-12f;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4bf(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4bf(bf16_t* p, float4 v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }
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
// width forward (+ fused branch pre-LayerNorm)
// ------------------------------------------------------------------------------------------------------------------
template <int S, int NI>
__global__ __launch_bounds__(256) void hc_width_fwd_kernel(const float* __restrict__ R, HcParams hp, const float* __restrict__ ln_gamma,
                                                           bf16_t* __restrict__ x_out, long long ldx, bf16_t* __restrict__ xn_out,
                                                           long long ldxn, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           float* __restrict__ coef, int B, int N, int D) {
    using C = Coef<S>;
    const int lane = threadIdx.x & 63;
    const long long M = (long long)B * N;
    const float cD = sqrtf((float)D);
    const float sa = *hp.sa, sb = *hp.sb;
    for (long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long long)gridDim.x * 4) {
        const int b = (int)(m / N), n = (int)(m % N);
        float4 r[S][NI];
        float ss[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            ss[s] = 0.f;
            const float* rp = R + (((long long)b * S + s) * N + n) * D;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int e = (i * 64 + lane) * 4;
                r[s][i] = (e < D) ? ld4(rp + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                ss[s] += r[s][i].x * r[s][i].x + r[s][i].y * r[s][i].y + r[s][i].z * r[s][i].z + r[s][i].w * r[s][i].w;
            }
        }
        float dots[S][S + 2];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < S + 2; ++t) dots[s][t] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e0 = (i * 64 + lane) * 4;
            if (e0 < D) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int e = e0 + c;
                    const float g1 = hp.hc_gamma[e] + 1.f;
                    float w[S + 2];
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) w[t] = hp.Wa[(long long)e * (S + 1) + t] * g1;
                    w[S + 1] = hp.wb[e] * g1;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float rv = f4c(r[s][i], c);
#pragma unroll
                        for (int t = 0; t < S + 2; ++t) dots[s][t] += rv * w[t];
                    }
                }
            }
        }
        float alpha[S][S + 1], beta[S], apre[S][S + 1], bpre[S], rn[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            rn[s] = 1.f / fmaxf(sqrtf(wave_sum(ss[s])), NORM_EPS);
#pragma unroll
            for (int t = 0; t < S + 1; ++t) {
                apre[s][t] = wave_sum(dots[s][t]) * rn[s] * cD;
                alpha[s][t] = tanhf(apre[s][t]) * sa + hp.Aa[s * (S + 1) + t];
            }
            bpre[s] = wave_sum(dots[s][S + 1]) * rn[s] * cD;
            beta[s] = tanhf(bpre[s]) * sb + hp.Bb[s];
        }
        // branch input + LayerNorm
        float4 x[NI];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                x[i].x += alpha[s][0] * r[s][i].x; x[i].y += alpha[s][0] * r[s][i].y;
                x[i].z += alpha[s][0] * r[s][i].z; x[i].w += alpha[s][0] * r[s][i].w;
            }
            sum += x[i].x + x[i].y + x[i].z + x[i].w;
        }
        const float mean = wave_sum(sum) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < D) {
                const float a = x[i].x - mean, bq = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
                q += a * a + bq * bq + c * c + d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + LN_EPS);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < D) {
                const float4 g = ld4(ln_gamma + e);
                st4bf(xn_out + m * ldxn + e, make_float4((x[i].x - mean) * rstd * g.x, (x[i].y - mean) * rstd * g.y,
                                                         (x[i].z - mean) * rstd * g.z, (x[i].w - mean) * rstd * g.w));
                if (x_out) st4bf(x_out + m * ldx + e, x[i]);
            }
        }
        if (lane == 0) {
            mean_out[m] = mean;
            rstd_out[m] = rstd;
            float* cp = coef + m * C::W;
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int t = 0; t < S + 1; ++t) {
                    cp[C::A + s * (S + 1) + t] = alpha[s][t];
                    cp[C::AP + s * (S + 1) + t] = apre[s][t];
                }
                cp[C::Bt + s] = beta[s];
                cp[C::BP + s] = bpre[s];
                cp[C::RN + s] = rn[s];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// depth forward: Rn_t = sum_s alpha[s][t+1] R_s + beta[t] y
// ------------------------------------------------------------------------------------------------------------------
template <int S, int NI>
__global__ __launch_bounds__(256) void hc_depth_fwd_kernel(const float* __restrict__ R, const bf16_t* __restrict__ y, long long ldy,
                                                           const float* __restrict__ coef, float* __restrict__ Rn, int B, int N, int D) {
    using C = Coef<S>;
    const int lane = threadIdx.x & 63;
    const long long M = (long long)B * N;
    for (long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long long)gridDim.x * 4) {
        const int b = (int)(m / N), n = (int)(m % N);
        const float* cp = coef + m * C::W;
        float alpha[S][S + 1], beta[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int t = 0; t < S + 1; ++t) alpha[s][t] = cp[C::A + s * (S + 1) + t];
            beta[s] = cp[C::Bt + s];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e >= D) continue;
            float4 r[S];
#pragma unroll
            for (int s = 0; s < S; ++s) r[s] = ld4(R + (((long long)b * S + s) * N + n) * D + e);
            const float4 yv = ld4bf(y + m * ldy + e);
#pragma unroll
            for (int t = 0; t < S; ++t) {
                float4 o = make_float4(beta[t] * yv.x, beta[t] * yv.y, beta[t] * yv.z, beta[t] * yv.w);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    o.x += alpha[s][t + 1] * r[s].x; o.y += alpha[s][t + 1] * r[s].y;
                    o.z += alpha[s][t + 1] * r[s].z; o.w += alpha[s][t + 1] * r[s].w;
                }
                *reinterpret_cast<float4*>(Rn + (((long long)b * S + t) * N + n) * D + e) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// depth backward: dy = sum_t beta[t] dRn_t (bf16) ; dbeta[t] = <dRn_t, y>
// ------------------------------------------------------------------------------------------------------------------
template <int S, int NI>
__global__ __launch_bounds__(256) void hc_depth_bwd_kernel(const float* __restrict__ dRn, const bf16_t* __restrict__ y, long long ldy,
                                                           const float* __restrict__ coef, bf16_t* __restrict__ dy, long long lddy,
                                                           float* __restrict__ dbeta, int B, int N, int D) {
    using C = Coef<S>;
    const int lane = threadIdx.x & 63;
    const long long M = (long long)B * N;
    for (long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long long)gridDim.x * 4) {
        const int b = (int)(m / N), n = (int)(m % N);
        const float* cp = coef + m * C::W;
        float beta[S], db[S];
#pragma unroll
        for (int s = 0; s < S; ++s) { beta[s] = cp[C::Bt + s]; db[s] = 0.f; }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e >= D) continue;
            const float4 yv = ld4bf(y + m * ldy + e);
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const float4 g = ld4(dRn + (((long long)b * S + t) * N + n) * D + e);
                o.x += beta[t] * g.x; o.y += beta[t] * g.y; o.z += beta[t] * g.z; o.w += beta[t] * g.w;
                db[t] += g.x * yv.x + g.y * yv.y + g.z * yv.z + g.w * yv.w;
            }
            st4bf(dy + m * lddy + e, o);
        }
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const float v = wave_sum(db[t]);
            if (lane == 0) dbeta[m * S + t] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// width backward.  Inputs: dRn (= dR'), dx (grad wrt branch input x, fp32), R, coef, dbeta.  Output dR + parameter partials.
// partial row layout (floats): dWa[D][S+1] | dwb[D] | dgamma[D] | dAa[S][S+1] | dBb[S] | dsa | dsb
// ------------------------------------------------------------------------------------------------------------------
template <int S, int NI>
__global__ __launch_bounds__(256) void hc_width_bwd_kernel(const float* __restrict__ dRn, const float* __restrict__ dx, long long lddx,
                                                           const float* __restrict__ R, const float* __restrict__ coef,
                                                           const float* __restrict__ dbeta, HcParams hp, float* __restrict__ dR,
                                                           float* __restrict__ partial, int B, int N, int D) {
    using C = Coef<S>;
    extern __shared__ float lds_red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long M = (long long)B * N;
    const float cD = sqrtf((float)D);
    const float sa = *hp.sa, sb = *hp.sb;

    float accWa[NI][4][S + 1], accwb[NI][4], accg[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            accwb[i][c] = 0.f; accg[i][c] = 0.f;
#pragma unroll
            for (int t = 0; t < S + 1; ++t) accWa[i][c][t] = 0.f;
        }
    float accAa[S][S + 1], accBb[S], accsa = 0.f, accsb = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        accBb[s] = 0.f;
#pragma unroll
        for (int t = 0; t < S + 1; ++t) accAa[s][t] = 0.f;
    }

    for (long long m = (long long)blockIdx.x * 4 + wave; m < M; m += (long long)gridDim.x * 4) {
        const int b = (int)(m / N), n = (int)(m % N);
        const float* cp = coef + m * C::W;
        float4 r[S][NI], g[S][NI], dxv[NI];
        float dal[S][S + 1];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < S + 1; ++t) dal[s][t] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            const bool ok = e < D;
            dxv[i] = ok ? ld4(dx + m * lddx + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const long long off = (((long long)b * S + s) * N + n) * D + e;
                r[s][i] = ok ? ld4(R + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                g[s][i] = ok ? ld4(dRn + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                dal[s][0] += dxv[i].x * r[s][i].x + dxv[i].y * r[s][i].y + dxv[i].z * r[s][i].z + dxv[i].w * r[s][i].w;
#pragma unroll
                for (int t = 0; t < S; ++t)
                    dal[s][t + 1] += g[t][i].x * r[s][i].x + g[t][i].y * r[s][i].y + g[t][i].z * r[s][i].z + g[t][i].w * r[s][i].w;
            }
        }
        float alpha[S][S + 1], dap[S][S + 1], dbp[S], rn[S], gdot[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            rn[s] = cp[C::RN + s];
            const float bpre = cp[C::BP + s];
            const float tb = tanhf(bpre);
            const float dbt = dbeta[m * S + s];
            dbp[s] = dbt * sb * (1.f - tb * tb);
            accBb[s] += dbt;
            accsb += dbt * tb;
            gdot[s] = dbp[s] * bpre;
#pragma unroll
            for (int t = 0; t < S + 1; ++t) {
                alpha[s][t] = cp[C::A + s * (S + 1) + t];
                const float apre = cp[C::AP + s * (S + 1) + t];
                const float ta = tanhf(apre);
                const float da = wave_sum(dal[s][t]);
                dap[s][t] = da * sa * (1.f - ta * ta);
                accAa[s][t] += da;
                accsa += da * ta;
                gdot[s] += dap[s][t] * apre;
            }
            // <g_s, R_s> = sum_t dap * apre / rn  (n_s . W = apre  =>  sum_e W[e] (gamma+1) c R_s[e] = apre / rn)
            gdot[s] = gdot[s] / rn[s];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e0 = (i * 64 + lane) * 4;
            if (e0 >= D) continue;
            float4 outv[S];
#pragma unroll
            for (int s = 0; s < S; ++s) outv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + c;
                const float g1 = hp.hc_gamma[e] + 1.f;
                float w[S + 2];
#pragma unroll
                for (int t = 0; t < S + 1; ++t) w[t] = hp.Wa[(long long)e * (S + 1) + t];
                w[S + 1] = hp.wb[e];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float rv = f4c(r[s][i], c);
                    float dn = dbp[s] * w[S + 1];
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) dn += dap[s][t] * w[t];
                    const float nhat = rv * rn[s] * cD;            // n_s / (gamma + 1)
                    const float nv = nhat * g1;
#pragma unroll
                    for (int t = 0; t < S + 1; ++t) accWa[i][c][t] += nv * dap[s][t];
                    accwb[i][c] += nv * dbp[s];
                    accg[i][c] += dn * nhat;
                    const float gs = dn * g1 * cD;
                    float o = alpha[s][0] * f4c(dxv[i], c) + rn[s] * (gs - gdot[s] * rn[s] * rn[s] * rv);
#pragma unroll
                    for (int t = 0; t < S; ++t) o += alpha[s][t + 1] * f4c(g[t][i], c);
                    f4(outv[s], c) = o;
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) *reinterpret_cast<float4*>(dR + (((long long)b * S + s) * N + n) * D + e0) = outv[s];
        }
    }

    // block reduction of the parameter partials over the 4 waves
    const int P = D * (S + 3) + S * (S + 1) + S + 2;
    float* red = lds_red;                      // [4][P]
    float* mine = red + (size_t)wave * P;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = (i * 64 + lane) * 4 + c;
            if (e < D) {
#pragma unroll
                for (int t = 0; t < S + 1; ++t) mine[(long long)e * (S + 1) + t] = accWa[i][c][t];
                mine[D * (S + 1) + e] = accwb[i][c];
                mine[D * (S + 2) + e] = accg[i][c];
            }
        }
    if (lane == 0) {
        float* q = mine + D * (S + 3);
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int t = 0; t < S + 1; ++t) q[s * (S + 1) + t] = accAa[s][t];
            q[S * (S + 1) + s] = accBb[s];
        }
        q[S * (S + 1) + S] = accsa;
        q[S * (S + 1) + S + 1] = accsb;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < P; e += 256)
        partial[(long long)blockIdx.x * P + e] = red[e] + red[P + e] + red[2 * P + e] + red[3 * P + e];
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
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(a)[i], w = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
    }
}

template <int S>
int dispatch_width_fwd(int D, int grid, hipStream_t st, const float* R, HcParams hp, const float* lng, bf16_t* x, long long ldx, bf16_t* xn,
                       long long ldxn, float* mean, float* rstd, float* coef, int B, int N) {
#define ALM_W(NI) hipLaunchKernelGGL((hc_width_fwd_kernel<S, NI>), dim3(grid), dim3(256), 0, st, R, hp, lng, x, ldx, xn, ldxn, mean, rstd, coef, B, N, D)
    if (D <= 256) ALM_W(1); else if (D <= 512) ALM_W(2); else if (D <= 1024) ALM_W(4); else return ALM_ERR_UNSUPPORTED;
#undef ALM_W
    return 0;
}
template <int S>
int dispatch_depth_fwd(int D, int grid, hipStream_t st, const float* R, const bf16_t* y, long long ldy, const float* coef, float* Rn, int B, int N) {
#define ALM_W(NI) hipLaunchKernelGGL((hc_depth_fwd_kernel<S, NI>), dim3(grid), dim3(256), 0, st, R, y, ldy, coef, Rn, B, N, D)
    if (D <= 256) ALM_W(1); else if (D <= 512) ALM_W(2); else if (D <= 1024) ALM_W(4); else return ALM_ERR_UNSUPPORTED;
#undef ALM_W
    return 0;
}
template <int S>
int dispatch_depth_bwd(int D, int grid, hipStream_t st, const float* dRn, const bf16_t* y, long long ldy, const float* coef, bf16_t* dy,
                       long long lddy, float* dbeta, int B, int N) {
#define ALM_W(NI) hipLaunchKernelGGL((hc_depth_bwd_kernel<S, NI>), dim3(grid), dim3(256), 0, st, dRn, y, ldy, coef, dy, lddy, dbeta, B, N, D)
    if (D <= 256) ALM_W(1); else if (D <= 512) ALM_W(2); else if (D <= 1024) ALM_W(4); else return ALM_ERR_UNSUPPORTED;
#undef ALM_W
    return 0;
}
template <int S>
int dispatch_width_bwd(int D, int grid, hipStream_t st, const float* dRn, const float* dx, long long lddx, const float* R, const float* coef,
                       const float* dbeta, HcParams hp, float* dR, float* partial, int B, int N) {
    const int P = D * (S + 3) + S * (S + 1) + S + 2;
    const size_t smem = (size_t)4 * P * sizeof(float);
#define ALM_W(NI)                                                                                                                        \
    do {                                                                                                                                 \
        hipFuncSetAttribute(reinterpret_cast<const void*>(hc_width_bwd_kernel<S, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((hc_width_bwd_kernel<S, NI>), dim3(grid), dim3(256), smem, st, dRn, dx, lddx, R, coef, dbeta, hp, dR, partial, B, N, D); \
    } while (0)
    if (D <= 256) ALM_W(1); else if (D <= 512) ALM_W(2); else if (D <= 1024) ALM_W(4); else return ALM_ERR_UNSUPPORTED;
#undef ALM_W
    return 0;
}

int grid_tokens(long long M, int cap) { return (int)((M + 3) / 4 < cap ? (M + 3) / 4 : cap); }

}  // namespace

#define ALM_S_DISPATCH(S, CALL2, CALL4)          \
    do {                                         \
        if ((S) == 2) rc = CALL2;                \
        else if ((S) == 4) rc = CALL4;           \
        else return ALM_ERR_UNSUPPORTED;         \
    } while (0)

extern "C" int alm_hc_coef_width(int S) { return 2 * S * (S + 1) + 3 * S; }
extern "C" int alm_hc_partial_width(int S, int D) { return D * (S + 3) + S * (S + 1) + S + 2; }
extern "C" int alm_hc_partial_blocks(long long M) { return grid_tokens(M, 512); }

extern "C" int alm_hc_width_fwd(const float* R, const float* hc_gamma, const float* Wa, const float* sa, const float* Aa, const float* wb,
                                const float* sb, const float* Bb, const float* ln_gamma, void* x_out, long long ldx, void* xn_out,
                                long long ldxn, float* mean, float* rstd, float* coef, int B, int S, int N, int D, void* stream) {
    if ((D & 3) || (ldx & 3) || (ldxn & 3)) return ALM_ERR_BAD_ARG;
    HcParams hp{hc_gamma, Wa, sa, Aa, wb, sb, Bb};
    const int grid = grid_tokens((long long)B * N, 8192);
    int rc;
    ALM_S_DISPATCH(S, (dispatch_width_fwd<2>(D, grid, (hipStream_t)stream, R, hp, ln_gamma, (bf16_t*)x_out, ldx, (bf16_t*)xn_out, ldxn, mean, rstd, coef, B, N)),
                   (dispatch_width_fwd<4>(D, grid, (hipStream_t)stream, R, hp, ln_gamma, (bf16_t*)x_out, ldx, (bf16_t*)xn_out, ldxn, mean, rstd, coef, B, N)));
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_hc_depth_fwd(const float* R, const void* y, long long ldy, const float* coef, float* Rn, int B, int S, int N, int D,
                                void* stream) {
    if ((D & 3) || (ldy & 3)) return ALM_ERR_BAD_ARG;
    const int grid = grid_tokens((long long)B * N, 8192);
    int rc;
    ALM_S_DISPATCH(S, (dispatch_depth_fwd<2>(D, grid, (hipStream_t)stream, R, (const bf16_t*)y, ldy, coef, Rn, B, N)),
                   (dispatch_depth_fwd<4>(D, grid, (hipStream_t)stream, R, (const bf16_t*)y, ldy, coef, Rn, B, N)));
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_hc_depth_bwd(const float* dRn, const void* y, long long ldy, const float* coef, void* dy, long long lddy, float* dbeta,
                                int B, int S, int N, int D, void* stream) {
    if ((D & 3) || (ldy & 3) || (lddy & 3)) return ALM_ERR_BAD_ARG;
    const int grid = grid_tokens((long long)B * N, 8192);
    int rc;
    ALM_S_DISPATCH(S, (dispatch_depth_bwd<2>(D, grid, (hipStream_t)stream, dRn, (const bf16_t*)y, ldy, coef, (bf16_t*)dy, lddy, dbeta, B, N)),
                   (dispatch_depth_bwd<4>(D, grid, (hipStream_t)stream, dRn, (const bf16_t*)y, ldy, coef, (bf16_t*)dy, lddy, dbeta, B, N)));
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// partial: [alm_hc_partial_blocks(B*N)][alm_hc_partial_width(S, D)] floats; reduce with alm_colsum.
extern "C" int alm_hc_width_bwd(const float* dRn, const float* dx, long long lddx, const float* R, const float* coef, const float* dbeta,
                                const float* hc_gamma, const float* Wa, const float* sa, const float* wb, const float* sb, float* dR,
                                float* partial, int B, int S, int N, int D, void* stream) {
    if ((D & 3) || (lddx & 3)) return ALM_ERR_BAD_ARG;
    HcParams hp{hc_gamma, Wa, sa, nullptr, wb, sb, nullptr};
    const int grid = alm_hc_partial_blocks((long long)B * N);
    int rc;
    ALM_S_DISPATCH(S, (dispatch_width_bwd<2>(D, grid, (hipStream_t)stream, dRn, dx, lddx, R, coef, dbeta, hp, dR, partial, B, N)),
                   (dispatch_width_bwd<4>(D, grid, (hipStream_t)stream, dRn, dx, lddx, R, coef, dbeta, hp, dR, partial, B, N)));
    if (rc) return rc;
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

extern "C" int alm_add_f32(const float* a, const float* b, float* out, long long n, void* stream) {
    if (n & 3) return ALM_ERR_BAD_ARG;
    const long long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, n4);
    ALM_LAUNCH_CHECK();
    return 0;
}
