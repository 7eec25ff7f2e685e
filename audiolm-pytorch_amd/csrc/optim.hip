// Fused optimiser step for the training path, gfx950 (MI355X): global gradient-norm clip + Adam / AdamW over EVERY parameter in two
// launches (sum of squares; update), instead of the ~10 foreach passes + a host synchronisation the reference's
//     accelerator.clip_grad_norm_(transformer.parameters(), max_grad_norm)      trainer.py:953-954 (and :595, :1251-1255)
//     self.optim.step()   with optimizer.py:get_optimizer -> torch.optim.Adam / AdamW
// spend.  HBM-bound: 16 B read + 12 B written per parameter (p, g, m, v fp32).  SURVEY.md §8(f) item 4.
//
// Multi-tensor layout: a device table of AlmOptTensor (pointers + length + weight decay) and a device table of chunks (tensor index,
// chunk index): workgroup b processes 16 Ki elements of one tensor.  The clip coefficient is derived on the device from the sum of
// squares (no host round trip): coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_).
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int CHUNK = 16384;

__global__ __launch_bounds__(256) void sumsq_kernel(const AlmOptTensor* __restrict__ tensors, const int2* __restrict__ chunks, float* __restrict__ partial) {
    __shared__ float red[4];
    const int2 c = chunks[blockIdx.x];
    const AlmOptTensor t = tensors[c.x];
    const long long beg = (long long)c.y * CHUNK, end = beg + CHUNK < t.n ? beg + CHUNK : t.n;
    const float* g = reinterpret_cast<const float*>(t.g);
    float s = 0.f;
    if (((uintptr_t)g & 15) == 0) {
        const long long e4 = beg + ((end - beg) & ~3LL);
        for (long long i = beg + threadIdx.x * 4; i < e4; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (long long i = e4 + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
    float lr, beta1, beta2, eps, step_size, inv_sqrt_bc2, max_norm;
    int decoupled;                   // 1: AdamW (p *= 1 - lr * wd), 0: Adam with L2 (g += wd * p)
    const float* sumsq;              // device scalar (sum of squares of every gradient) or NULL: no clipping
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float wd, float clip, const AdamArgs& a) {
    g *= clip;
    if (wd != 0.f) {
        if (a.decoupled) p *= 1.f - a.lr * wd;
        else g = fmaf(wd, p, g);
    }
    m = fmaf(a.beta1, m, (1.f - a.beta1) * g);              // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(a.beta2, v, (1.f - a.beta2) * g * g);          //        exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AlmOptTensor* __restrict__ tensors, const int2* __restrict__ chunks, AdamArgs a) {
    const int2 c = chunks[blockIdx.x];
    const AlmOptTensor t = tensors[c.x];
    const long long beg = (long long)c.y * CHUNK, end = beg + CHUNK < t.n ? beg + CHUNK : t.n;
    float* p = reinterpret_cast<float*>(t.p);
    const float* g = reinterpret_cast<const float*>(t.g);
    float* m = reinterpret_cast<float*>(t.m);
    float* v = reinterpret_cast<float*>(t.v);
    float clip = 1.f;
    if (a.sumsq) clip = fminf(1.f, a.max_norm / (sqrtf(*a.sumsq) + 1e-6f));
    if (t.step > 0) {
        // this tensor's own step count (torch.optim.Adam keeps `step` per parameter: tensors whose gradients first appear later, or are None on
        // some steps, have their own bias corrections)
        const double bc1 = 1.0 - exp((double)t.step * log((double)a.beta1)), bc2 = 1.0 - exp((double)t.step * log((double)a.beta2));
        a.step_size = (float)((double)a.lr / bc1);
        a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    long long tail = beg;
    if (vec) {
        const long long e4 = beg + ((end - beg) & ~3LL);
        for (long long i = beg + threadIdx.x * 4; i < e4; i += 1024) {
            float4 pv = *reinterpret_cast<const float4*>(p + i), mv = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            adam_one(pv.x, gv.x, mv.x, vv.x, t.wd, clip, a);
            adam_one(pv.y, gv.y, mv.y, vv.y, t.wd, clip, a);
            adam_one(pv.z, gv.z, mv.z, vv.z, t.wd, clip, a);
            adam_one(pv.w, gv.w, mv.w, vv.w, t.wd, clip, a);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
        tail = e4;
    }
    for (long long i = tail + threadIdx.x; i < end; i += 256) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_one(pv, g[i], mv, vv, t.wd, clip, a);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

}  // namespace

extern "C" int alm_opt_chunk_elems(void) { return CHUNK; }

// partial: fp32 [nchunks] (sum of squares of each chunk's gradients; alm_reduce_sum over it gives the squared global norm)
extern "C" int alm_opt_grad_sumsq(const AlmOptTensor* tensors, const int* chunks, int nchunks, float* partial, void* stream) {
    if (nchunks <= 0) return 0;
    if (!tensors || !chunks || !partial) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, tensors, reinterpret_cast<const int2*>(chunks), partial);
    ALM_LAUNCH_CHECK();
    return 0;
}

// step: 1-based count of THIS step (bias corrections 1 - beta^step) for tensors whose own AlmOptTensor.step is 0.  sumsq: device scalar for the clip (NULL = no clipping).
extern "C" int alm_opt_adam_step(const AlmOptTensor* tensors, const int* chunks, int nchunks, float lr, float beta1, float beta2, float eps, int step,
                                 int decoupled_weight_decay, const float* sumsq, float max_norm, void* stream) {
    if (nchunks <= 0) return 0;
    if (!tensors || !chunks || step < 1) return ALM_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a{lr, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), max_norm, decoupled_weight_decay, sumsq};
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, tensors, reinterpret_cast<const int2*>(chunks), a);
    ALM_LAUNCH_CHECK();
    return 0;
}
