// Position-bias table MLPs of the `flash_attn=False` models, gfx950 (MI355X).
//
// Reference: RelativePositionBias.forward audiolm_pytorch.py:226-242 (Linear(1, d) SiLU, 2 x (Linear(d, d) SiLU), Linear(d, heads) on the
// relative positions -(n-1) .. n-1) and FineTransformer.pos_bias_mlp :1065-1071 / :1271 (Linear(2, d) SiLU, Linear(d, d) SiLU,
// Linear(d, heads) on the (relative frame, relative quantizer) grid).  The reference then GATHERS an (h, n, n) tensor from the MLP output;
// here the MLP output stays a per-head table [heads][1 + rows] (slot 0 = the "special pair" value: cross_attn_bias :779 / null_pos_bias
// :1061), already divided by the attention scale, which the attention kernels index in place (attention.hip).
//
// The d x d layers are plain contractions and run on the MFMA GEMMs (gemm.hip) from the host mirror; this file holds the thin first layer
// (1 or 2 inputs), the thin last layer (heads outputs), SiLU and their backward passes.  All fp32 except the bf16 GEMM operands.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float silu_grad_f(float x) {
    const float s = sigmoid_f(x);
    return s * fmaf(x, 1.f - s, 1.f);
}

// pre[l][c] = b[c] + sum_i x[l][i] W[c][i];  act = silu(pre) (bf16)
__global__ __launch_bounds__(256) void posmlp_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                            float* __restrict__ pre, bf16_t* __restrict__ act, int L, int in_dim, int C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)L * C) return;
    const int l = (int)(e / C), c = (int)(e % C);
    float v = b[c];
    for (int i = 0; i < in_dim; ++i) v = fmaf(x[(long long)l * in_dim + i], W[c * in_dim + i], v);
    pre[e] = v;
    act[e] = f2bf(silu_f(v));
}

// partial[chunk][j * C + c]: j = 0 -> sum_l dpre[l][c] (bias grad), j = 1 + i -> sum_l dpre[l][c] x[l][i] (weight grad column i)
__global__ __launch_bounds__(256) void posmlp_in_bwd_kernel(const bf16_t* __restrict__ dpre, const float* __restrict__ x, float* __restrict__ partial,
                                                            int L, int in_dim, int C, int rows_per_chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int l0 = blockIdx.y * rows_per_chunk, l1 = min(L, l0 + rows_per_chunk);
    float sb = 0.f, s0 = 0.f, s1 = 0.f;
    for (int l = l0; l < l1; ++l) {
        const float g = bf2f(dpre[(long long)l * C + c]);
        sb += g;
        s0 = fmaf(g, x[(long long)l * in_dim], s0);
        if (in_dim > 1) s1 = fmaf(g, x[(long long)l * in_dim + 1], s1);
    }
    float* out = partial + (long long)blockIdx.y * (1 + in_dim) * C;
    out[c] = sb;
    out[C + c] = s0;
    if (in_dim > 1) out[2 * C + c] = s1;
}

__global__ __launch_bounds__(256) void silu_fwd_kernel(const float* __restrict__ pre, bf16_t* __restrict__ act, long long n4) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(pre)[e];
    reinterpret_cast<uint2*>(act)[e] = make_uint2(pack_bf2(silu_f(v.x), silu_f(v.y)), pack_bf2(silu_f(v.z), silu_f(v.w)));
}

// dpre = dact * silu'(pre)  (bf16: the next contraction's operand)
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ dact, const float* __restrict__ pre, bf16_t* __restrict__ dpre,
                                                       long long n4) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    const float4 g = reinterpret_cast<const float4*>(dact)[e];
    const float4 v = reinterpret_cast<const float4*>(pre)[e];
    reinterpret_cast<uint2*>(dpre)[e] = make_uint2(pack_bf2(g.x * silu_grad_f(v.x), g.y * silu_grad_f(v.y)),
                                                   pack_bf2(g.z * silu_grad_f(v.z), g.w * silu_grad_f(v.w)));
}

constexpr int MAXH = 16;

// one wave per table row l: tbl[h][1 + l] = inv_scale * (b[h] + sum_c act[l][c] W[h][c]); workgroup 0 also writes slot 0
__global__ __launch_bounds__(256) void posmlp_out_fwd_kernel(const bf16_t* __restrict__ act, const float* __restrict__ W, const float* __restrict__ b,
                                                             const float* __restrict__ special, float* __restrict__ tbl, int L, int C, int H,
                                                             float inv_scale) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x < H) tbl[(long long)threadIdx.x * (L + 1)] = special ? special[threadIdx.x] * inv_scale : 0.f;
    if (l >= L) return;
    float acc[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) acc[h] = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float a = bf2f(act[(long long)l * C + c]);
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < H) acc[h] = fmaf(a, W[h * C + c], acc[h]);
    }
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
        if (h < H) {
            const float s = wave_sum(acc[h]);
            if (lane == 0) tbl[(long long)h * (L + 1) + 1 + l] = (s + b[h]) * inv_scale;
        }
}

// one wave per table row l: g[h] = inv_scale * dtbl[h][1 + l];  g_bf16[l][0 .. Hp) (zero padded);
// dpre[l][c] = silu'(pre[l][c]) * sum_h g[h] W[h][c];  workgroup 0: dspecial[h] = inv_scale * dtbl[h][0]
__global__ __launch_bounds__(256) void posmlp_out_bwd_kernel(const float* __restrict__ dtbl, const float* __restrict__ W, const float* __restrict__ pre,
                                                             bf16_t* __restrict__ g_out, bf16_t* __restrict__ dpre, float* __restrict__ dspecial,
                                                             int L, int C, int H, int Hp, float inv_scale) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x < H && dspecial) dspecial[threadIdx.x] = dtbl[(long long)threadIdx.x * (L + 1)] * inv_scale;
    if (l >= L) return;
    float g[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) g[h] = (h < H) ? dtbl[(long long)h * (L + 1) + 1 + l] * inv_scale : 0.f;
    if (lane < Hp) {
        float mine = 0.f;
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h == lane) mine = g[h];
        g_out[(long long)l * Hp + lane] = f2bf(mine);
    }
    for (int c = lane; c < C; c += 64) {
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < H) s = fmaf(g[h], W[h * C + c], s);
        dpre[(long long)l * C + c] = f2bf(s * silu_grad_f(pre[(long long)l * C + c]));
    }
}

}  // namespace

extern "C" int alm_posmlp_in_fwd(const float* x, const float* W, const float* b, float* pre, void* act, int L, int in_dim, int C, void* stream) {
    if (L <= 0 || C <= 0 || in_dim < 1 || in_dim > 2) return ALM_ERR_UNSUPPORTED;
    const long long n = (long long)L * C;
    hipLaunchKernelGGL(posmlp_in_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, W, b, pre, (bf16_t*)act, L,
                       in_dim, C);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_posmlp_in_bwd_chunks(int L) { return (L + 63) / 64 < 64 ? (L + 63) / 64 : 64; }

// partial: fp32 [alm_posmlp_in_bwd_chunks(L)][(1 + in_dim) * C]; alm_colsum over it gives [db | dW[:, 0] | dW[:, 1]]
extern "C" int alm_posmlp_in_bwd(const void* dpre, const float* x, float* partial, int L, int in_dim, int C, void* stream) {
    if (L <= 0 || C <= 0 || in_dim < 1 || in_dim > 2) return ALM_ERR_UNSUPPORTED;
    const int chunks = alm_posmlp_in_bwd_chunks(L);
    const int rpc = (L + chunks - 1) / chunks;
    hipLaunchKernelGGL(posmlp_in_bwd_kernel, dim3((C + 255) / 256, chunks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpre, x, partial, L,
                       in_dim, C, rpc);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_silu_fwd(const float* pre, void* act, long long n, void* stream) {
    if (n <= 0 || (n & 3) || ((uintptr_t)pre & 15) || ((uintptr_t)act & 7)) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, (bf16_t*)act, n / 4);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_silu_bwd(const float* dact, const float* pre, void* dpre, long long n, void* stream) {
    if (n <= 0 || (n & 3) || ((uintptr_t)pre & 15) || ((uintptr_t)dact & 15) || ((uintptr_t)dpre & 7)) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dact, pre, (bf16_t*)dpre, n / 4);
    ALM_LAUNCH_CHECK();
    return 0;
}

// tbl: fp32 [H][L + 1]
extern "C" int alm_posmlp_out_fwd(const void* act, const float* W, const float* b, const float* special, float* tbl, int L, int C, int H,
                                  float inv_scale, void* stream) {
    if (L <= 0 || C <= 0 || H <= 0 || H > MAXH) return ALM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(posmlp_out_fwd_kernel, dim3((L + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)act, W, b, special, tbl, L, C, H,
                       inv_scale);
    ALM_LAUNCH_CHECK();
    return 0;
}

// g_out: bf16 [L][Hp] (Hp = H rounded up to 8); dpre: bf16 [L][C]; dspecial: fp32 [H] or NULL
extern "C" int alm_posmlp_out_bwd(const float* dtbl, const float* W, const float* pre, void* g_out, void* dpre, float* dspecial, int L, int C,
                                  int H, int Hp, float inv_scale, void* stream) {
    if (L <= 0 || C <= 0 || H <= 0 || H > MAXH || Hp < H || Hp > 64) return ALM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(posmlp_out_bwd_kernel, dim3((L + 3) / 4), dim3(256), 0, (hipStream_t)stream, dtbl, W, pre, (bf16_t*)g_out, (bf16_t*)dpre,
                       dspecial, L, C, H, Hp, inv_scale);
    ALM_LAUNCH_CHECK();
    return 0;
}
