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
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

struct ConvArgs {
    const float* x; const float* wp; const float* bias; const float* residual; float* out;
    int B, Cin, CinP, Cout, CoutP, Tin, Tout, ks, stride, dil, pad, elu;
};

// grid: (ceil(Tout / 256), CoutP / (32 * NA), B); 4 waves along time, 64 output steps each
template <int NA>
__global__ __launch_bounds__(256) void conv1d_causal_kernel(ConvArgs a) {
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

    for (int tap = 0; tap < a.ks; ++tap) {
        int pos[2];
        bool pok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int p = tin[j] + tap * a.dil;
            p = p < 0 ? -p : p;                                    // reflect (F.pad mode='reflect'): index -i -> i
            pok[j] = p < a.Tin;
            pos[j] = pok[j] ? p : 0;
        }
        const float* wt = a.wp + (long long)tap * a.CinP * a.CoutP + co0 + lr;
#pragma unroll 4
        for (int cc = 0; cc < a.CinP; cc += 2) {                    // unrolled: 4 k-steps of loads in flight ahead of their MFMAs
            const int ci = cc + lh;
            float av[NA], bv[2];
#pragma unroll
            for (int i = 0; i < NA; ++i) av[i] = wt[(long long)ci * a.CoutP + i * 32];
            const bool cok = ci < a.Cin;
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = (cok && pok[j]) ? xb[(long long)ci * a.Tin + pos[j]] : 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout: column = lane & 31 (time), row = (r & 3) + 8 * (r >> 2) + 4 * lh (output channel)
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + j * 32 + lr;
            if (t >= a.Tout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co >= a.Cout) continue;
                float v = acc[i][j][r] + a.bias[co];
                if (a.elu) v = elu1(v);
                const long long o = ((long long)b * a.Cout + co) * a.Tout + t;
                if (a.residual) v += a.residual[o];
                a.out[o] = v;
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

// codebook [C][d] -> transposed [d][CP] (zero padded) + squared norms [CP] (+inf for the pad codes: never selected)
__global__ __launch_bounds__(256) void rvq_pack_kernel(const float* __restrict__ E, float* __restrict__ Et, float* __restrict__ e2, int C, int d,
                                                       int CP) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= CP) return;
    float s = 0.f;
    for (int k = 0; k < d; ++k) {
        const float v = c < C ? E[(long long)c * d + k] : 0.f;
        Et[(long long)k * CP + c] = v;
        s += v * v;
    }
    e2[c] = c < C ? s : INFINITY;
}

struct RvqArgs {
    const float* x; long long ldx;            // [T][ldx] frames (this group's d columns start at x)
    const float* E;                           // [Q][C][d]   original codebooks (residual update)
    const float* Et;                          // [Q][d][CP]  packed transposed
    const float* e2;                          // [Q][CP]
    long long* idx; long long ldi;            // [T][ldi] output indices (this group's Q columns start at idx)
    float* quant; long long ldq;              // optional: sum of the selected code vectors [T][ldq] (the `quantized` output), or NULL
    int T, d, C, CP, Q;
};

__global__ __launch_bounds__(256) void rvq_encode_kernel(RvqArgs a) {
    extern __shared__ float sm[];
    const int ld = a.d + 1;
    float* res = sm;                          // [32][d + 1]
    float* x2 = res + 32 * ld;                // [32]
    float* candd = x2 + 32;                   // [4][32]
    int* candi = reinterpret_cast<int*>(candd + 128);   // [4][32]
    int* win = candi + 128;                   // [32]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int f0 = blockIdx.x * 32;

    // residual tile <- x ; |x|^2 per frame (thread = (frame t / 8, column phase t % 8))
    const int fj = t >> 3, ph = t & 7;
    {
        float s = 0.f;
        const bool ok = f0 + fj < a.T;
        for (int e = ph; e < a.d; e += 8) {
            const float v = ok ? a.x[(long long)(f0 + fj) * a.ldx + e] : 0.f;
            res[fj * ld + e] = v;
            s += v * v;
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (ph == 0) x2[fj] = s;
    }
    __syncthreads();

    const int nblk = a.CP / 32;               // 32-code blocks; wave w scans block pairs 2w, 2w + 8, ...
    for (int q = 0; q < a.Q; ++q) {
        const float* Et = a.Et + (long long)q * a.d * a.CP;
        const float* e2 = a.e2 + (long long)q * a.CP;
        const float xn = x2[lr];
        float best = INFINITY;
        int besti = 0x7fffffff;
        // two 32-code blocks per pass share every residual (B) operand read; the k loop is unrolled so that several steps' loads are
        // in flight ahead of the dependent MFMA chain
        for (int cb = wave * 2; cb < nblk; cb += 8) {
            const bool two = cb + 1 < nblk;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const float* ep = Et + cb * 32 + lr;
            const int o1 = two ? 32 : 0;
#pragma unroll 8
            for (int k = 0; k < a.d; k += 2) {
                const int kk = k + lh;
                const bool kok = kk < a.d;
                const float av0 = kok ? ep[(long long)kk * a.CP] : 0.f;             // A[code][k]
                const float av1 = kok ? ep[(long long)kk * a.CP + o1] : 0.f;
                const float bv = kok ? res[lr * ld + kk] : 0.f;                      // B[k][frame]
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv, acc1, 0, 0, 0);
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
            for (int w = 1; w < 4; ++w) {
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
            for (int e = ph; e < a.d; e += 8) {
                const float v = res[fj * ld + e] - er[e];
                res[fj * ld + e] = v;
                s += v * v;
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (ph == 0) x2[fj] = s;
        }
        __syncthreads();
    }
    if (a.quant) {                            // quantized = x - final residual
        const bool ok = f0 + fj < a.T;
        if (ok)
            for (int e = ph; e < a.d; e += 8) a.quant[(long long)(f0 + fj) * a.ldq + e] = a.x[(long long)(f0 + fj) * a.ldx + e] - res[fj * ld + e];
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
                                 int Tin, int ksize, int stride, int dilation, int elu, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || ksize <= 0 || stride <= 0 || dilation <= 0) return ALM_ERR_BAD_ARG;
    const int pad = dilation * (ksize - 1) + 1 - stride;
    if (pad < 0 || pad >= Tin || Tin < stride) return ALM_ERR_UNSUPPORTED;
    const int Tout = (Tin - stride) / stride + 1;
    ConvArgs a{x, wp, bias, residual, out, B, Cin, (Cin + 1) & ~1, Cout, (Cout + 31) & ~31, Tin, Tout, ksize, stride, dilation, pad, elu};
    const int gx = (Tout + 255) / 256;
    if (a.CoutP % 64 == 0)
        hipLaunchKernelGGL(conv1d_causal_kernel<2>, dim3(gx, a.CoutP / 64, B), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv1d_causal_kernel<1>, dim3(gx, a.CoutP / 32, B), dim3(256), 0, (hipStream_t)stream, a);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_rvq_padded_codes(int C) { return (C + 31) & ~31; }

// E fp32 [Q][C][d] -> Et [Q][d][CP], e2 [Q][CP]  (CP = alm_rvq_padded_codes(C)); once per codebook update
extern "C" int alm_rvq_pack(const float* E, float* Et, float* e2, int Q, int C, int d, void* stream) {
    if (Q <= 0 || C <= 0 || d <= 0) return ALM_ERR_BAD_ARG;
    const int CP = alm_rvq_padded_codes(C);
    for (int q = 0; q < Q; ++q)
        hipLaunchKernelGGL(rvq_pack_kernel, dim3((CP + 255) / 256), dim3(256), 0, (hipStream_t)stream, E + (long long)q * C * d,
                           Et + (long long)q * d * CP, e2 + (long long)q * CP, C, d, CP);
    ALM_LAUNCH_CHECK();
    return 0;
}

// residual-VQ encode of T frames x[T][ldx] (d columns) against Q codebooks: idx[T][ldi] (int64, Q columns), optional quantized sum
extern "C" int alm_rvq_encode(const float* x, long long ldx, const float* E, const float* Et, const float* e2, long long* idx, long long ldi,
                              float* quant, long long ldq, int T, int d, int C, int Q, void* stream) {
    if (T <= 0) return 0;
    if (d <= 0 || C <= 0 || Q <= 0) return ALM_ERR_BAD_ARG;
    const size_t smem = (size_t)(32 * (d + 1) + 32 + 128) * sizeof(float) + (128 + 32) * sizeof(int);
    if (smem > 160 * 1024) return ALM_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rvq_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RvqArgs a{x, ldx, E, Et, e2, idx, ldi, quant, ldq, T, d, C, alm_rvq_padded_codes(C), Q};
    hipLaunchKernelGGL(rvq_encode_kernel, dim3((T + 31) / 32), dim3(256), smem, (hipStream_t)stream, a);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_bct_to_btc(const float* in, float* out, int B, int C, int T, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    hipLaunchKernelGGL(bct_to_btc_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, in, out, C, T);
    ALM_LAUNCH_CHECK();
    return 0;
}
