// SoundStream's LocalTransformer (reference soundstream.py:397-440: local-attention's LocalMHA + FeedForward around the codec's
// encoder / decoder), gfx950, exact fp32 like the rest of the codec path (the encoder output feeds the residual VQ: indices must not move).
// Everything stays in the codec's [B][C][T] layout (time = lane axis, coalesced): the four Linear layers of a block are k = 1 causal
// convolutions on the exact-fp32 MFMA kernel (codec.hip, alm_conv1d_causal, with its fused residual add); this file holds what is not a GEMM:
//   * LayerNorm over the channel axis (prenorm of LocalMHA / first layer of FeedForward: nn.LayerNorm with weight and bias)
//   * the windowed causal attention itself: qk-l2norm + learned per-feature scales (qk_rmsnorm, attention scale 8), rotary + xpos positions
//     relative to the (look-back | own) window pair, keys j visible to query i iff 0 <= i - j <= window (causal + exact window size +
//     look_backward = 1), softmax, value aggregation, per-head sigmoid value gate
//   * GEGLU over the channel halves
// The attention is tens of MFLOP per window and runs once per codec call (18 windows for 30 s of audio against a 30 ms conv encoder):
// a plain fp32 VALU kernel -- one thread per query, keys / values of the window pair in LDS transposed to [feature][slot] (conflict-free:
// the lanes of a wave read consecutive slots) -- not an MFMA tiling.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

// out[b][c][t] = (x[b][c][t] - mean_c) * rstd * gamma[c] + beta[c]      one thread per (b, t): consecutive lanes = consecutive t
__global__ __launch_bounds__(256) void ln_bct_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ out, int C, int T, float eps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const long long base = (long long)blockIdx.y * C * T + t;
    const float* xp = x + base;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < C; c += 4) {
        s0 += xp[(long long)c * T]; s1 += xp[(long long)(c + 1) * T]; s2 += xp[(long long)(c + 2) * T]; s3 += xp[(long long)(c + 3) * T];
    }
    for (; c < C; ++c) s0 += xp[(long long)c * T];
    const float mean = ((s0 + s1) + (s2 + s3)) / (float)C;
    s0 = s1 = s2 = s3 = 0.f;
    for (c = 0; c + 3 < C; c += 4) {
        const float d0 = xp[(long long)c * T] - mean, d1 = xp[(long long)(c + 1) * T] - mean, d2 = xp[(long long)(c + 2) * T] - mean,
                    d3 = xp[(long long)(c + 3) * T] - mean;
        s0 += d0 * d0; s1 += d1 * d1; s2 += d2 * d2; s3 += d3 * d3;
    }
    for (; c < C; ++c) { const float d = xp[(long long)c * T] - mean; s0 += d * d; }
    const float rstd = 1.0f / sqrtf(((s0 + s1) + (s2 + s3)) / (float)C + eps);
    float* op = out + base;
    for (c = 0; c < C; ++c) op[(long long)c * T] = (xp[(long long)c * T] - mean) * rstd * gamma[c] + beta[c];
}

// out[b][i][t] = x[b][i][t] * gelu(x[b][I + i][t])        (local_attention.transformer.GEGLU: x, gate = chunk(2, dim = -1) of 'b n c')
__global__ __launch_bounds__(256) void geglu_bct_kernel(const float* __restrict__ x, float* __restrict__ out, int I, long long T, long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long bi = e / T, t = e - bi * T;
        const long long b = bi / I, i = bi - b * I;
        const float* xb = x + b * 2 * I * T;
        out[e] = xb[i * T + t] * gelu_f(xb[(I + i) * T + t]);
    }
}

// Windowed causal attention of one (window, head, batch): blockDim = window size W; thread = one query of the window.
// qkv [B][3 * H * DH][T] (channels q (h d) | k (h d) | v (h d)), gates [B][H][T] (pre-sigmoid), out [B][H * DH][T].
// cos_t / sin_t / xpos_t [2 W][DH]: rotary angle cosine / sine and the xpos scale of slot s of the (look-back | own) window pair
// (local-attention rotary.py: queries use slots W .. 2W-1 and the scale, keys use all slots and 1 / scale).
template <int DH>
__global__ __launch_bounds__(256) void local_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                         const float* __restrict__ cos_t, const float* __restrict__ sin_t, const float* __restrict__ xpos_t,
                                                         const float* __restrict__ gates, float* __restrict__ out, int H, int T, int W, float scale) {
    extern __shared__ float lds[];
    const int S2 = 2 * W;
    float* Kl = lds;                       // [DH][2W]
    float* Vl = lds + DH * S2;             // [DH][2W]
    const int w = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const long long HD = (long long)H * DH;
    const float* qb = qkv + ((long long)b * 3 * HD + (long long)h * DH) * T;
    const float* kb = qb + HD * T;
    const float* vb = kb + HD * T;
    constexpr int HALF = DH / 2;

    // ---- keys / values of the window pair: slot s <-> position j = (w - 1) W + s
    for (int s = tid; s < S2; s += W) {
        const long long j = (long long)(w - 1) * W + s;
        if (j < 0 || j >= T) {
#pragma unroll
            for (int d = 0; d < DH; ++d) { Kl[d * S2 + s] = 0.f; Vl[d * S2 + s] = 0.f; }     // never visible (before the sequence / after the query)
            continue;
        }
        float kn[DH];
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { kn[d] = kb[(long long)d * T + j]; ss += kn[d] * kn[d]; }
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);                                   // F.normalize(dim = -1, eps = 1e-12)
#pragma unroll
        for (int d = 0; d < DH; ++d) kn[d] = kn[d] * inv * k_scale[d];
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float rot = d < HALF ? -kn[d + HALF] : kn[d - HALF];                       // rotate_half: (x1, x2) -> (-x2, x1)
            const float isc = 1.0f / xpos_t[s * DH + d];
            Kl[d * S2 + s] = kn[d] * cos_t[s * DH + d] * isc + rot * sin_t[s * DH + d] * isc;
            Vl[d * S2 + s] = vb[(long long)d * T + j];
        }
    }
    __syncthreads();

    const long long i = (long long)w * W + tid;
    if (i >= T) return;
    const int sq = W + tid;                                                                  // the query's own slot
    float q[DH];
    {
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { q[d] = qb[(long long)d * T + i]; ss += q[d] * q[d]; }
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        float qn[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) qn[d] = q[d] * inv * q_scale[d] * scale;                // bq = bq * scale before the rotation (linear: same thing)
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float rot = d < HALF ? -qn[d + HALF] : qn[d - HALF];
            const float sc = xpos_t[sq * DH + d];
            q[d] = qn[d] * cos_t[sq * DH + d] * sc + rot * sin_t[sq * DH + d] * sc;
        }
    }
    // visible keys: 0 <= i - j <= W  <=>  slots max(sq - W, sq - i) .. sq
    int s0 = sq - W;
    if ((long long)(sq - s0) > i) s0 = sq - (int)i;
    float m = -INFINITY, l = 0.f;
    float acc[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    for (int s = s0; s <= sq; ++s) {
        float sim = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sim += q[d] * Kl[d * S2 + s];
        const float mn = fmaxf(m, sim);
        const float corr = expf(m - mn), p = expf(sim - mn);
        l = l * corr + p;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = acc[d] * corr + p * Vl[d * S2 + s];
        m = mn;
    }
    const float g = gates ? 1.0f / (1.0f + expf(-gates[((long long)b * H + h) * T + i])) : 1.0f;
    const float nrm = g / l;
    float* ob = out + ((long long)b * HD + (long long)h * DH) * T + i;
#pragma unroll
    for (int d = 0; d < DH; ++d) ob[(long long)d * T] = acc[d] * nrm;
}

template <int DH>
int launch_local(const float* qkv, const float* q_scale, const float* k_scale, const float* cos_t, const float* sin_t, const float* xpos_t,
                 const float* gates, float* out, int B, int H, int T, int W, float scale, hipStream_t st) {
    const int smem = 2 * DH * 2 * W * (int)sizeof(float);
    if (smem > 160 * 1024) return ALM_ERR_UNSUPPORTED;
    static bool attr_done = false;
    auto kfn = local_attn_kernel<DH>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kfn, dim3((T + W - 1) / W, H, B), dim3(W), smem, st, qkv, q_scale, k_scale, cos_t, sin_t, xpos_t, gates, out, H, T, W, scale);
    return 0;
}

}  // namespace

extern "C" int alm_layernorm_bct(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int T, float eps, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(ln_bct_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out, C, T, eps);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_geglu_bct(const float* x, float* out, int B, int I, int T, void* stream) {
    if (B <= 0 || I <= 0 || T <= 0) return ALM_ERR_BAD_ARG;
    const long long total = (long long)B * I * T;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(geglu_bct_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, out, I, (long long)T, total);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_local_attn(const float* qkv, const float* q_scale, const float* k_scale, const float* cos_t, const float* sin_t, const float* xpos_t,
                              const float* gates, float* out, int B, int H, int dim_head, int T, int window, float scale, void* stream) {
    if (B <= 0 || H <= 0 || T <= 0 || window <= 0) return ALM_ERR_BAD_ARG;
    if (window > 256) return ALM_ERR_UNSUPPORTED;                           // one thread per query of a window
    int rc;
    if (dim_head == 64) rc = launch_local<64>(qkv, q_scale, k_scale, cos_t, sin_t, xpos_t, gates, out, B, H, T, window, scale, (hipStream_t)stream);
    else if (dim_head == 32) rc = launch_local<32>(qkv, q_scale, k_scale, cos_t, sin_t, xpos_t, gates, out, B, H, T, window, scale, (hipStream_t)stream);
    else return ALM_ERR_UNSUPPORTED;
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}
