// HBM-bound row kernels of the transformer block (gfx950):
//   * bias-less LayerNorm fwd/bwd            -- reference audiolm_pytorch.py:191-198  (F.layer_norm, gamma only, eps 1e-5)
//   * fused GEGLU + inner LayerNorm fwd/bwd  -- reference audiolm_pytorch.py:246-260  (gelu(gate) * x, gate = 2nd half; LN(inner))
//   * column sums (two-stage parameter-gradient reductions, bias grads)
// Statistics are fp32; activations are bf16 (the dtype the reference's autocast feeds its GEMMs), residual input fp32.
// One wave (64 lanes) owns one LayerNorm row of <= 1024 features: 16-B coalesced loads, the row stays in registers,
// reductions are wave-64 shuffles (no LDS, no block barrier).  The 2730-wide GEGLU row is owned by a 256-thread block.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma.  NI = ceil(D / 256) chunks of 4 features per lane.
// ------------------------------------------------------------------------------------------------------------------
template <typename TIN, typename TOUT, int NI>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIN* __restrict__ x, long long ldx, const float* __restrict__ gamma,
                                                     TOUT* __restrict__ y, long long ldy, bf16_t* __restrict__ xcopy, long long ldc,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        float4 v[NI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            v[i] = (e < D) ? load4(x + (long long)row * ldx + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < D) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += a * a + b * b + c * c + d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + LN_EPS);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < D) {
                const float4 g = load4(gamma + e);
                store4(y + (long long)row * ldy + e,
                       make_float4((v[i].x - mean) * rstd * g.x, (v[i].y - mean) * rstd * g.y, (v[i].z - mean) * rstd * g.z,
                                   (v[i].w - mean) * rstd * g.w));
                if (xcopy) store4(xcopy + (long long)row * ldc + e, v[i]);
            }
        }
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward: g = dy * gamma ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ extra) ;
// dgamma partial sums per block -> dgamma_part[gridDim.x][D] (reduced by colsum).
// ------------------------------------------------------------------------------------------------------------------
template <typename TDY, typename TX, typename TDX, int NI>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, long long lddy, const TX* __restrict__ x, long long ldx,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const bf16_t* __restrict__ extra, long long lde,
                                                     TDX* __restrict__ dx, long long lddx, float* __restrict__ dgamma_part, int rows, int D) {
    __shared__ float red[4][NI * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 dg[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float4 xh[NI], g[NI];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < D) {
                const float4 xv = load4(x + (long long)row * ldx + e);
                const float4 d = load4(dy + (long long)row * lddy + e);
                const float4 gm = load4(gamma + e);
                xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                g[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
                dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
                s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
            } else {
                xh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < D) {
                float4 o = make_float4(rstd * (g[i].x - c1 - xh[i].x * c2), rstd * (g[i].y - c1 - xh[i].y * c2),
                                       rstd * (g[i].z - c1 - xh[i].z * c2), rstd * (g[i].w - c1 - xh[i].w * c2));
                if (extra) {
                    const float4 ex = load4(extra + (long long)row * lde + e);
                    o.x += ex.x; o.y += ex.y; o.z += ex.z; o.w += ex.w;
                }
                store4(dx + (long long)row * lddx + e, o);
            }
        }
    }
    if (!dgamma_part) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(&red[wave][(i * 64 + lane) * 4]) = dg[i];
    __syncthreads();
    for (int e = threadIdx.x; e < D; e += 256)
        dgamma_part[(long long)blockIdx.x * D + e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
}

// ------------------------------------------------------------------------------------------------------------------
// column sum: out[c] (+)= scale * sum_r in[r][c]   (fp32 or bf16 input)
// ------------------------------------------------------------------------------------------------------------------
// Two stages for tall inputs (deterministic, no atomics): stage 1 = grid (cols / 64, R row chunks), a block covers 64 columns (256 B per
// row: whole cache lines) x 4 row lanes and writes ws[chunk][c]; stage 2 sums the R chunk rows.  Short inputs run stage 1 only (R = 1).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, long long ld, int rows, int cols, float* __restrict__ out,
                                                     float scale, int accumulate, int rows_per_chunk, int direct) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto ldv = [&](int r) -> float {
        if constexpr (sizeof(T) == 2) return bf2f(in[(long long)r * ld + c]);
        else return in[(long long)r * ld + c];
    };
    if (c < cols) {
        int r = r0 + ry;
        for (; r + 12 < r1; r += 16) {
            s0 += ldv(r); s1 += ldv(r + 4); s2 += ldv(r + 8); s3 += ldv(r + 12);
        }
        for (; r < r1; r += 4) s0 += ldv(r);
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < cols) {
        const float v = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
        if (direct) out[c] = accumulate ? out[c] + v * scale : v * scale;
        else out[(long long)blockIdx.y * cols + c] = v;
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ ws, int chunks, int cols, float* __restrict__ out, float scale,
                                                            int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float v = 0.f;
    for (int k = 0; k < chunks; ++k) v += ws[(long long)k * cols + c];
    v *= scale;
    out[c] = accumulate ? out[c] + v : v;
}

// ------------------------------------------------------------------------------------------------------------------
// fused GEGLU + LayerNorm(inner) forward.  u: [rows][ldu] bf16, x half at cols [0, I), gate half at [goff, goff + I).
// h = gelu(gate) * x ; out = LN(h) * gamma (bf16, cols [I, Ipad) zero-filled).
// ------------------------------------------------------------------------------------------------------------------
// component c of a float4 (indexing through `&v.x` is undefined behaviour and was observed to drop the 4th component)
__device__ __forceinline__ float& f4e(float4& v, int c) { return reinterpret_cast<float*>(&v)[c]; }
__device__ __forceinline__ float f4e(const float4& v, int c) { return reinterpret_cast<const float*>(&v)[c]; }

constexpr int GE_MAX = 3;    // 4-wide vectors per thread: supports padded inner widths up to 3 * 256 * 4 = 3072

// block-wide sum of one or two values with ONE barrier: the scratch is parity-double-buffered by the caller, so the previous
// reduction's readers never race with this one's writers
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ void block_sum256x2(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
}

// 8-byte (4 x bf16) accesses: u rows are 16-B aligned (ldu = 2 * Ipad, Ipad % 8 == 0) and the pad columns [I, Ipad) of both halves
// hold zeros (the packed W1 has zero rows there), so whole vectors can be loaded up to Ipad; statistics only count e < I.
// GELU = v * Phi(v) with the exact-erf definition of F.gelu (reference audiolm_pytorch.py:246-249), Phi from gauss_cdf_pdf.
__global__ __launch_bounds__(256) void geglu_ln_fwd_kernel(const bf16_t* __restrict__ u, long long ldu, int goff, const float* __restrict__ gamma,
                                                           bf16_t* __restrict__ out, long long ldo, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int rows, int I, int Ipad) {
    __shared__ float red[4][4];                                  // 2 reductions per row x parity
    const int t = threadIdx.x;
    float4 gam[GE_MAX];
#pragma unroll
    for (int j = 0; j < GE_MAX; ++j) {
        const int e = (t + 256 * j) * 4;
        gam[j] = make_float4(e < I ? gamma[e] : 0.f, e + 1 < I ? gamma[e + 1] : 0.f, e + 2 < I ? gamma[e + 2] : 0.f, e + 3 < I ? gamma[e + 3] : 0.f);
    }
    // SOFTWARE PREFETCH (round 3): a row costs a load phase, two block reductions and a store phase that depend on each other; with one row in
    // flight per block the kernel was latency-bound (4.5 TB/s, ~7 us per row and block).  The NEXT row's raw bf16 vectors are requested before the
    // current row is processed (two register sets, loop unrolled by two), as in the hyper-connection kernels.
    struct Raw { uint2 x[GE_MAX], g[GE_MAX]; };
    auto issue = [&](Raw& r, int row) {
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            r.x[j] = r.g[j] = make_uint2(0u, 0u);
            if (e < Ipad && row < rows) {
                r.x[j] = *reinterpret_cast<const uint2*>(u + (long long)row * ldu + e);
                r.g[j] = *reinterpret_cast<const uint2*>(u + (long long)row * ldu + goff + e);
            }
        }
    };
    auto cvt = [](uint2 v) {
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    };
    auto process = [&](const Raw& r, int row, int par) {
        float4 h[GE_MAX];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            h[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Ipad) {
                const float4 xv = cvt(r.x[j]);
                const float4 gv = cvt(r.g[j]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float cdf, pdf;
                    gauss_cdf_pdf(f4e(gv, c), cdf, pdf);
                    f4e(h[j], c) = (e + c < I) ? f4e(gv, c) * cdf * f4e(xv, c) : 0.f;
                }
                s += h[j].x + h[j].y + h[j].z + h[j].w;
            }
        }
        const float mean = block_sum256(s, red[par]) / (float)I;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            if (e < I) {
                const float a = h[j].x - mean, b = h[j].y - mean, c = h[j].z - mean, d = h[j].w - mean;
                q += a * a + (e + 1 < I ? b * b : 0.f) + (e + 2 < I ? c * c : 0.f) + (e + 3 < I ? d * d : 0.f);
            }
        }
        const float rstd = rsqrtf(block_sum256(q, red[par + 1]) / (float)I + LN_EPS);
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            if (e < Ipad) {
                float4 o;
                o.x = (e < I) ? (h[j].x - mean) * rstd * gam[j].x : 0.f;
                o.y = (e + 1 < I) ? (h[j].y - mean) * rstd * gam[j].y : 0.f;
                o.z = (e + 2 < I) ? (h[j].z - mean) * rstd * gam[j].z : 0.f;
                o.w = (e + 3 < I) ? (h[j].w - mean) * rstd * gam[j].w : 0.f;
                store4(out + (long long)row * ldo + e, o);
            }
        }
        if (t == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    };
    Raw ra, rb;
    const int stride = gridDim.x;
    issue(ra, blockIdx.x);
    for (int row = blockIdx.x; row < rows; row += 2 * stride) {
        issue(rb, row + stride);
        process(ra, row, 0);
        if (row + stride >= rows) break;
        issue(ra, row + 2 * stride);
        process(rb, row + stride, 2);
    }
}

// backward: dhn = grad wrt LN output (bf16 [rows][lddh]); writes du (bf16 [rows][ldu], both halves, pads zeroed) and
// per-block dgamma partial sums dgamma_part[gridDim.x][I].  One block reduction (two sums, one barrier) per row; gelu and its
// derivative come from one cdf / pdf evaluation per element and are kept in registers between the two passes.
__global__ __launch_bounds__(256) void geglu_ln_bwd_kernel(const bf16_t* __restrict__ dhn, long long lddh, const bf16_t* __restrict__ u,
                                                           long long ldu, int goff, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           bf16_t* __restrict__ du, float* __restrict__ dgamma_part, int rows, int I, int Ipad) {
    __shared__ float red[2][8];
    const int t = threadIdx.x;
    float4 dgam[GE_MAX], gam[GE_MAX];
#pragma unroll
    for (int j = 0; j < GE_MAX; ++j) {
        const int e = (t + 256 * j) * 4;
        dgam[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        gam[j] = make_float4(e < I ? gamma[e] : 0.f, e + 1 < I ? gamma[e + 1] : 0.f, e + 2 < I ? gamma[e + 2] : 0.f, e + 3 < I ? gamma[e + 3] : 0.f);
    }
    // software prefetch of the next row (see geglu_ln_fwd_kernel): its raw vectors and its two LayerNorm statistics travel together
    struct Raw { uint2 x[GE_MAX], g[GE_MAX], d[GE_MAX]; float mean, rstd; };
    auto issue = [&](Raw& r, int row) {
        const int rr = row < rows ? row : 0;
        r.mean = mean_in[rr];
        r.rstd = rstd_in[rr];
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            r.x[j] = r.g[j] = r.d[j] = make_uint2(0u, 0u);
            if (e < Ipad && row < rows) {
                r.x[j] = *reinterpret_cast<const uint2*>(u + (long long)row * ldu + e);
                r.g[j] = *reinterpret_cast<const uint2*>(u + (long long)row * ldu + goff + e);
                r.d[j] = *reinterpret_cast<const uint2*>(dhn + (long long)row * lddh + e);
            }
        }
    };
    auto cvt = [](uint2 v) {
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    };
    auto process = [&](const Raw& r, int row, int par) {
        const float mean = r.mean, rstd = r.rstd;
        float4 ga[GE_MAX], gb[GE_MAX], g[GE_MAX];                    // ga = d h / d x = gelu(gate); gb = d h / d gate = x * gelu'(gate)
                                                                      // (xhat is recomputed in the second pass: 12 registers that decide 4 vs 3 waves per SIMD)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            ga[j] = gb[j] = g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Ipad) {
                const float4 xv = cvt(r.x[j]);
                const float4 gv = cvt(r.g[j]);
                const float4 d = cvt(r.d[j]);
                // BRANCH-FREE over the 4 elements (round 3): as `if (e + c < I) { ... }` every element became its own exec-masked block -- ~60 taken
                // branches per row.  The pad columns [I, Ipad) need no test: gamma is 0 there, so g = 0 and the two row sums see nothing; the
                // dgamma accumulators of pad columns are never stored; the outputs are zeroed by a select at the store.
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gt = f4e(gv, c), xx = f4e(xv, c);
                    float cdf, pdf;
                    gauss_cdf_pdf(gt, cdf, pdf);
                    const float gl = gt * cdf;
                    const float xhat = (gl * xx - mean) * rstd;
                    const float gg = f4e(d, c) * f4e(gam[j], c);
                    f4e(ga[j], c) = gl;
                    f4e(gb[j], c) = xx * fmaf(gt, pdf, cdf);
                    f4e(g[j], c) = gg;
                    f4e(dgam[j], c) += f4e(d, c) * xhat;
                    s1 += gg;
                    s2 += gg * xhat;
                }
            }
        }
        block_sum256x2(s1, s2, red[par]);
        const float c1 = s1 / (float)I, c2 = s2 / (float)I;
#pragma unroll
        for (int j = 0; j < GE_MAX; ++j) {
            const int e = (t + 256 * j) * 4;
            if (e < Ipad) {
                float4 dxh, dgh;
                const float4 xv = cvt(r.x[j]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xhat = (f4e(ga[j], c) * f4e(xv, c) - mean) * rstd;
                    const float dh = rstd * (f4e(g[j], c) - c1 - xhat * c2);
                    f4e(dxh, c) = (e + c < I) ? dh * f4e(ga[j], c) : 0.f;
                    f4e(dgh, c) = (e + c < I) ? dh * f4e(gb[j], c) : 0.f;
                }
                store4(du + (long long)row * ldu + e, dxh);
                store4(du + (long long)row * ldu + goff + e, dgh);
            }
        }
    };
    Raw ra, rb;
    const int stride = gridDim.x;
    issue(ra, blockIdx.x);
    for (int row = blockIdx.x; row < rows; row += 2 * stride) {
        issue(rb, row + stride);
        process(ra, row, 0);
        if (row + stride >= rows) break;
        issue(ra, row + 2 * stride);
        process(rb, row + stride, 1);
    }
    if (!dgamma_part) return;
#pragma unroll
    for (int j = 0; j < GE_MAX; ++j) {
        const int e = (t + 256 * j) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (e + c < I) dgamma_part[(long long)blockIdx.x * I + e + c] = f4e(dgam[j], c);
    }
}

template <typename TIN, typename TOUT>
int launch_ln_fwd(const TIN* x, long long ldx, const float* gamma, TOUT* y, long long ldy, bf16_t* xc, long long ldc, float* mean,
                  float* rstd, int rows, int D, hipStream_t st) {
    const int grid = min((rows + 3) / 4, 4096);
    if (D <= 256) hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, 1>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, y, ldy, xc, ldc, mean, rstd, rows, D);
    else if (D <= 512) hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, 2>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, y, ldy, xc, ldc, mean, rstd, rows, D);
    else if (D <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, 4>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, y, ldy, xc, ldc, mean, rstd, rows, D);
    else if (D <= 2048) hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, 8>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, y, ldy, xc, ldc, mean, rstd, rows, D);
    else return ALM_ERR_UNSUPPORTED;
    return 0;
}

template <typename TDY, typename TX, typename TDX>
int launch_ln_bwd(const TDY* dy, long long lddy, const TX* x, long long ldx, const float* mean, const float* rstd, const float* gamma,
                  const bf16_t* extra, long long lde, TDX* dx, long long lddx, float* part, int grid, int rows, int D, hipStream_t st) {
#define ALM_LNB(NI) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, NI>), dim3(grid), dim3(256), 0, st, dy, lddy, x, ldx, mean, rstd, gamma, extra, lde, dx, lddx, part, rows, D)
    if (D <= 256) ALM_LNB(1);
    else if (D <= 512) ALM_LNB(2);
    else if (D <= 1024) ALM_LNB(4);
    else if (D <= 2048) ALM_LNB(8);
    else return ALM_ERR_UNSUPPORTED;
#undef ALM_LNB
    return 0;
}


// ---- standalone GEGLU (reference audiolm_pytorch.py:246-249: x, gate = chunk(2, dim=-1); gelu(gate) * x) on fp32 rows [rows][2 I]: the module form
// used outside the fused stack (inside it the gate is fused with the inner LayerNorm: geglu_ln_*).  Exact-erf GELU (F.gelu default).
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int I) {
    const long long n = rows * I;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long r = e / I;
        const int c = (int)(e % I);
        const float xv = x[r * 2 * I + c], g = x[r * 2 * I + I + c];
        y[e] = gelu_f(g) * xv;
    }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long long rows,
                                                        int I) {
    const long long n = rows * I;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long r = e / I;
        const int c = (int)(e % I);
        const float xv = x[r * 2 * I + c], g = x[r * 2 * I + I + c], d = dy[e];
        dx[r * 2 * I + c] = d * gelu_f(g);
        dx[r * 2 * I + I + c] = d * xv * gelu_grad_f(g);
    }
}

}  // namespace

extern "C" int alm_ln_partial_blocks(int rows) { return min((rows + 3) / 4, 512); }

extern "C" int alm_layernorm_fwd(const void* x, int x_is_bf16, long long ldx, const float* gamma, void* y, int y_is_f32, long long ldy, void* xcopy,
                                 long long ldc, float* mean, float* rstd, int rows, int D, void* stream) {
    if (rows <= 0) return 0;
    if ((D & 3) || (ldx & 3) || (ldy & 3) || (xcopy && (ldc & 3))) return ALM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (y_is_f32)
        rc = x_is_bf16 ? launch_ln_fwd((const bf16_t*)x, ldx, gamma, (float*)y, ldy, (bf16_t*)xcopy, ldc, mean, rstd, rows, D, st)
                       : launch_ln_fwd((const float*)x, ldx, gamma, (float*)y, ldy, (bf16_t*)xcopy, ldc, mean, rstd, rows, D, st);
    else
        rc = x_is_bf16 ? launch_ln_fwd((const bf16_t*)x, ldx, gamma, (bf16_t*)y, ldy, (bf16_t*)xcopy, ldc, mean, rstd, rows, D, st)
                       : launch_ln_fwd((const float*)x, ldx, gamma, (bf16_t*)y, ldy, (bf16_t*)xcopy, ldc, mean, rstd, rows, D, st);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// dgamma_part must hold alm_ln_partial_blocks(rows) * D floats (or be NULL); reduce it with alm_colsum_f32.
extern "C" int alm_layernorm_bwd(const void* dy, int dy_is_f32, long long lddy, const void* x, int x_is_bf16, long long ldx, const float* mean,
                                 const float* rstd, const float* gamma, const void* extra, long long lde, void* dx, int dx_is_bf16,
                                 long long lddx, float* dgamma_part, int rows, int D, void* stream) {
    if (rows <= 0) return 0;
    if ((D & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) || (extra && (lde & 3))) return ALM_ERR_BAD_ARG;
    const int grid = alm_ln_partial_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dy_is_f32) {
        // fp32 upstream gradient: the final LayerNorm below the logit heads (fp32 x, fp32 dx)
        if (x_is_bf16 || dx_is_bf16) return ALM_ERR_UNSUPPORTED;
        rc = launch_ln_bwd((const float*)dy, lddy, (const float*)x, ldx, mean, rstd, gamma, (const bf16_t*)extra, lde, (float*)dx, lddx, dgamma_part, grid, rows, D, st);
    } else if (x_is_bf16 && dx_is_bf16)
        rc = launch_ln_bwd((const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, rstd, gamma, (const bf16_t*)extra, lde, (bf16_t*)dx, lddx, dgamma_part, grid, rows, D, st);
    else if (x_is_bf16)
        rc = launch_ln_bwd((const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, rstd, gamma, (const bf16_t*)extra, lde, (float*)dx, lddx, dgamma_part, grid, rows, D, st);
    else if (dx_is_bf16)
        rc = launch_ln_bwd((const bf16_t*)dy, lddy, (const float*)x, ldx, mean, rstd, gamma, (const bf16_t*)extra, lde, (bf16_t*)dx, lddx, dgamma_part, grid, rows, D, st);
    else
        rc = launch_ln_bwd((const bf16_t*)dy, lddy, (const float*)x, ldx, mean, rstd, gamma, (const bf16_t*)extra, lde, (float*)dx, lddx, dgamma_part, grid, rows, D, st);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// stage 1 alone: ws[alm_colsum_chunks(rows)][cols] partial column sums (fp32 input); a consumer that sums the chunk rows itself (alm_hc_param_grads)
// saves the finish launch
extern "C" int alm_colsum_chunks(int rows);
extern "C" int alm_colsum_partial(const float* in, long long ld, int rows, int cols, float* ws, void* stream) {
    if (cols <= 0 || rows <= 0 || !ws) return ALM_ERR_BAD_ARG;
    const int chunks = alm_colsum_chunks(rows);
    const int rpc = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((cols + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, in, ld, rows, cols, ws, 1.f, 0, rpc, 0);
    ALM_LAUNCH_CHECK();
    return 0;
}

// number of row chunks stage 1 uses; the caller provides `ws` = alm_colsum_chunks(rows) * cols floats when that is > 1
extern "C" int alm_colsum_chunks(int rows) { return rows <= 128 ? 1 : min(16, (rows + 63) / 64); }

extern "C" int alm_colsum(const void* in, int in_is_bf16, long long ld, int rows, int cols, float* out, float scale, int accumulate, float* ws,
                          void* stream) {
    if (cols <= 0) return 0;
    const int chunks = ws ? alm_colsum_chunks(rows) : 1;
    const int rpc = (rows + chunks - 1) / chunks;
    const int direct = chunks == 1;
    dim3 grid((cols + 63) / 64, chunks);
    float* o1 = direct ? out : ws;
    if (in_is_bf16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld, rows, cols, o1, scale, accumulate, rpc, direct);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, ld, rows, cols, o1, scale, accumulate, rpc, direct);
    if (!direct)
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, chunks, cols, out, scale, accumulate);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_geglu_partial_blocks(int rows) { return min(rows, 1024); }

extern "C" int alm_geglu_ln_fwd(const void* u, long long ldu, int gate_offset, const float* gamma, void* out, long long ldo, float* mean,
                                float* rstd, int rows, int inner, int inner_pad, void* stream) {
    if (rows <= 0) return 0;
    if (inner <= 0 || inner_pad < inner || inner_pad > GE_MAX * 1024 || (inner_pad & 7) || (ldu & 7) || (gate_offset & 7)) return ALM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(geglu_ln_fwd_kernel, dim3(min(rows, 2048)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u, ldu, gate_offset, gamma,
                       (bf16_t*)out, ldo, mean, rstd, rows, inner, inner_pad);
    ALM_LAUNCH_CHECK();
    return 0;
}

// dgamma_part: alm_geglu_partial_blocks(rows) * inner floats (or NULL).
extern "C" int alm_geglu_ln_bwd(const void* dhn, long long lddh, const void* u, long long ldu, int gate_offset, const float* gamma,
                                const float* mean, const float* rstd, void* du, float* dgamma_part, int rows, int inner, int inner_pad,
                                void* stream) {
    if (rows <= 0) return 0;
    if (inner <= 0 || inner_pad < inner || inner_pad > GE_MAX * 1024 || (inner_pad & 7) || (ldu & 7) || (gate_offset & 7)) return ALM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(geglu_ln_bwd_kernel, dim3(alm_geglu_partial_blocks(rows)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dhn, lddh,
                       (const bf16_t*)u, ldu, gate_offset, gamma, mean, rstd, (bf16_t*)du, dgamma_part, rows, inner, inner_pad);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_geglu_fwd(const float* x, float* y, long long rows, int inner, void* stream) {
    if (rows <= 0 || inner <= 0) return 0;
    const long long n = rows * inner;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)min((n + 255) / 256, (long long)8192)), dim3(256), 0, (hipStream_t)stream, x, y, rows, inner);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_geglu_bwd(const float* dy, const float* x, float* dx, long long rows, int inner, void* stream) {
    if (rows <= 0 || inner <= 0) return 0;
    const long long n = rows * inner;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)min((n + 255) / 256, (long long)8192)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, rows, inner);
    ALM_LAUNCH_CHECK();
    return 0;
}
