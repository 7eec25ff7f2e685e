// gfx950 (MI355X / CDNA4) device helpers shared by every kernel in libaudiolm_hip.so.
// wave = 64 lanes, 4 SIMDs / CU, 256 CUs in 8 XCDs.  No CUDA compatibility layer: CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                             // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;            // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;

#define ALM_WAVE 64

// error codes returned through the C ABI (0 == hipSuccess)
#define ALM_ERR_BAD_ARG 10001
#define ALM_ERR_UNSUPPORTED 10002

#define ALM_LAUNCH_CHECK()                       \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even fp32 -> bf16 (same rounding as torch's .to(bfloat16)); NaN stays NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16 pair with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, quiet NaN)
typedef __attribute__((ext_vector_type(2))) __bf16 alm_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    alm_bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// LDS transpose read (ds_read_b64_tr_b16): within each 16-lane group, lane s supplies the address of 4 consecutive bf16 of
// "row" s >> 2, column chunk s & 3 of a [4][16] block; lane c of the group receives column c of that block (4 values, rows 0..3).
__device__ __forceinline__ bf16x4 lds_tr16(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds_ptr));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and derivative -- F.gelu default (reference audiolm_pytorch.py:246-249)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Gaussian cdf / pdf pair for GELU with ONE exponential (Abramowitz & Stegun 7.1.26: |erf error| < 1.5e-7, far below the bf16 / fp32
// statistics tolerances of the callers): cdf = 0.5 (1 + erf(v / sqrt 2)), pdf = exp(-v^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ void gauss_cdf_pdf(float v, float& cdf, float& pdf) {
    const float ax = fabsf(v) * 0.70710678118654752f;
    const float ex = __expf(-ax * ax);                                    // = exp(-v^2 / 2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, ex, 1.f);
    cdf = 0.5f * (1.f + copysignf(erf_abs, v));
    pdf = 0.3989422804014327f * ex;
}

// bijective XCD-aware block remap (block b runs on XCD b % 8): gives every XCD a contiguous chunk of the
// logical grid so that neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    const int xcd = bid % nx, idx = bid / nx;
    const int q = nwg / nx, r = nwg % nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
