// Fused optimiser step for the training path, gfx950 (MI355X): global gradient-norm clip + Adam / AdamW over EVERY parameter in two
// launches (sum of squares; update), instead of the ~10 foreach passes + a host synchronisation the reference's
//     accelerator.clip_grad_norm_(transformer.parameters(), max_grad_norm)      trainer.py:953-954 (and :595, :1251-1255)
//     self.optim.step()   with optimizer.py:get_optimizer -> torch.optim.Adam / AdamW
// spend.  HBM-bound: 16 B read + 12 B written per parameter (p, g, m, v fp32).  SURVEY.md §8(f) item 4.
//
// Multi-tensor layout: a table of AlmOptTensor (pointers + length + weight decay) and a device table of chunks (tensor index, chunk index):
// workgroup b processes 16 Ki elements of one tensor.  The clip coefficient is derived on the device from the sum of squares (no host round
// trip): coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_).
// Round 4: the tensor table travels IN THE KERNEL ARGUMENTS (64 tensors = 3 KB per launch; the ~150 tensors of a model = 3 launches per kernel), not
// through a pinned staging buffer + an asynchronous host-to-device copy per step.  The gradient storage moves every step (autograd allocates fresh
// .grad tensors), so the table cannot be cached on the device; the staged copy was the one thing in the fused step that could make the host wait for
// the stream (a small H2D copy is not guaranteed to be asynchronous on every box): the suspected cause of the +1.9 ms "slow mode" of the
// `with_optimizer` leg seen on some boxes and not on others (profiles/README.md).  The chunk tables depend on shapes only and stay on the device.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int CHUNK = 16384;

constexpr int OPT_BATCH = 64;                                   // tensors per launch: 64 x 48 B of kernel arguments
struct OptBatch {
    AlmOptTensor t[OPT_BATCH];
    int t0;                                                     // index of t[0] in the caller's table (the chunk table holds global tensor indices)
};

__global__ __launch_bounds__(256) void sumsq_kernel(OptBatch tb, const int2* __restrict__ chunks, float* __restrict__ partial) {
    __shared__ float red[4];
    const int2 c = chunks[blockIdx.x];
    const AlmOptTensor t = tb.t[c.x - tb.t0];
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

__global__ __launch_bounds__(256) void adam_kernel(OptBatch tb, const int2* __restrict__ chunks, AdamArgs a) {
    const int2 c = chunks[blockIdx.x];
    const AlmOptTensor t = tb.t[c.x - tb.t0];
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

// walks the HOST table in batches of OPT_BATCH tensors; the chunk table is tensor-major (all chunks of tensor 0, then tensor 1, ...), so a batch owns a
// contiguous chunk range.  launch(batch, first chunk, chunk count) issues one kernel.
template <typename F>
int for_each_batch(const AlmOptTensor* tensors, int ntensors, int nchunks, F launch) {
    int c0 = 0;
    for (int t0 = 0; t0 < ntensors; t0 += OPT_BATCH) {
        OptBatch tb;
        tb.t0 = t0;
        const int nt = ntensors - t0 < OPT_BATCH ? ntensors - t0 : OPT_BATCH;
        long long nc = 0;
        for (int i = 0; i < nt; ++i) {
            tb.t[i] = tensors[t0 + i];
            if (tb.t[i].n < 0) return ALM_ERR_BAD_ARG;
            nc += (tb.t[i].n + CHUNK - 1) / CHUNK;
        }
        for (int i = nt; i < OPT_BATCH; ++i) tb.t[i] = AlmOptTensor{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0};
        if (c0 + nc > nchunks) return ALM_ERR_BAD_ARG;             // the chunk table does not belong to this tensor table
        if (nc > 0) launch(tb, c0, (int)nc);
        c0 += (int)nc;
    }
    return c0 == nchunks ? 0 : ALM_ERR_BAD_ARG;
}

// ---- Adam + the bf16 re-pack of the GEMM weights in one pass (round 4).  The dense weights of the stack are consumed as packed bf16 copies (W and W^T,
// zero-padded: alm_pack_weights_multi); after every optimiser step they used to be re-packed from the fp32 masters by the next forward: one more read
// of every weight (4 B) + 2 x 2 B written, 30 us per layer.  Here the update kernel walks a weight matrix in the pack kernel's 64 x 128 tiles and writes the
// new fp32 parameter, both moments AND both bf16 images: the re-pack's read disappears and its launches with it.  Same arithmetic as adam_kernel
// (adam_one), same packed bits as pack_weights_multi2_kernel (pack_bf2 of the updated fp32 value).
constexpr int PACK_BATCH = 12;
struct OptPackBatch {
    AlmOptPackJob job[PACK_BATCH];
    int tile_end[PACK_BATCH];                                   // exclusive prefix sums of the per-job 64 x 128 tile counts
    int njobs;
};
__global__ __launch_bounds__(256) void adam_pack_kernel(OptPackBatch pb, AdamArgs a) {
    __shared__ uint32_t tile[64][65];                                   // [row][column pair]: see pack_weights_multi2_kernel
    int j = 0;
    while (j + 1 < pb.njobs && (int)blockIdx.x >= pb.tile_end[j]) ++j;
    const AlmOptPackJob& q = pb.job[j];
    const int local = blockIdx.x - (j ? pb.tile_end[j - 1] : 0);
    const int tcols = (q.cols_pad + 127) / 128;
    const int r0 = (local / tcols) * 64, c0 = (local % tcols) * 128;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float* const P = reinterpret_cast<float*>(q.p);
    const float* const G = reinterpret_cast<const float*>(q.g);
    float* const Mo = reinterpret_cast<float*>(q.m);
    float* const V = reinterpret_cast<float*>(q.v);
    bf16_t* const dst = reinterpret_cast<bf16_t*>(q.dst);
    bf16_t* const dstT = reinterpret_cast<bf16_t*>(q.dstT);
    float clip = 1.f;
    if (a.sumsq) clip = fminf(1.f, a.max_norm / (sqrtf(*a.sumsq) + 1e-6f));
    if (q.step > 0) {
        const double bc1 = 1.0 - exp((double)q.step * log((double)a.beta1)), bc2 = 1.0 - exp((double)q.step * log((double)a.beta2));
        a.step_size = (float)((double)a.lr / bc1);
        a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
    const int c = c0 + 2 * tx;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i;
        float2 pv = make_float2(0.f, 0.f);
        if (r < q.rows) {
            const long long o = (long long)r * q.ld + c;
            if (c + 1 < q.cols) {
                pv = *reinterpret_cast<const float2*>(P + o);
                const float2 gv = *reinterpret_cast<const float2*>(G + o);
                float2 mv = *reinterpret_cast<const float2*>(Mo + o), vv = *reinterpret_cast<const float2*>(V + o);
                adam_one(pv.x, gv.x, mv.x, vv.x, q.wd, clip, a);
                adam_one(pv.y, gv.y, mv.y, vv.y, q.wd, clip, a);
                *reinterpret_cast<float2*>(P + o) = pv;
                *reinterpret_cast<float2*>(Mo + o) = mv;
                *reinterpret_cast<float2*>(V + o) = vv;
            } else if (c < q.cols) {
                float mv = Mo[o], vv = V[o];
                pv.x = P[o];
                adam_one(pv.x, G[o], mv, vv, q.wd, clip, a);
                P[o] = pv.x; Mo[o] = mv; V[o] = vv;
            }
        }
        const uint32_t pk = pack_bf2(pv.x, pv.y);
        tile[i][tx] = pk;
        if (dst && r < q.rows_pad && c < q.cols_pad) *reinterpret_cast<uint32_t*>(dst + (long long)r * q.ld_dst + c) = pk;
    }
    if (!dstT) return;
    __syncthreads();
    const int rp = tx & 31, hw = tx >> 5;
    for (int it = ty; it < 64; it += 4) {
        const int cc = it * 2 + hw;                                     // source column within the tile
        const int oc = c0 + cc, orow = r0 + 2 * rp;
        const uint32_t x = tile[2 * rp][cc >> 1], y = tile[2 * rp + 1][cc >> 1];
        const uint32_t pk = (cc & 1) ? ((x >> 16) | (y & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
        if (oc < q.cols_pad && orow < q.rows_pad) *reinterpret_cast<uint32_t*>(dstT + (long long)oc * q.ld_dstT + orow) = pk;
    }
}

}  // namespace

extern "C" int alm_opt_chunk_elems(void) { return CHUNK; }

// tensors: HOST array of ntensors AlmOptTensor (device pointers inside); chunks: DEVICE int32 pairs (tensor index, chunk index), tensor-major.
// partial: fp32 [nchunks] (sum of squares of each chunk's gradients; alm_reduce_sum over it gives the squared global norm)
extern "C" int alm_opt_grad_sumsq(const AlmOptTensor* tensors, int ntensors, const int* chunks, int nchunks, float* partial, void* stream) {
    if (nchunks <= 0 || ntensors <= 0) return 0;
    if (!tensors || !chunks || !partial) return ALM_ERR_BAD_ARG;
    const int rc = for_each_batch(tensors, ntensors, nchunks, [&](const OptBatch& tb, int c0, int nc) {
        hipLaunchKernelGGL(sumsq_kernel, dim3(nc), dim3(256), 0, (hipStream_t)stream, tb, reinterpret_cast<const int2*>(chunks) + c0, partial + c0);
    });
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// step: 1-based count of THIS step (bias corrections 1 - beta^step) for tensors whose own AlmOptTensor.step is 0.  sumsq: device scalar for the clip (NULL = no clipping).
extern "C" int alm_opt_adam_step(const AlmOptTensor* tensors, int ntensors, const int* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                                 int step, int decoupled_weight_decay, const float* sumsq, float max_norm, void* stream) {
    if (nchunks <= 0 || ntensors <= 0) return 0;
    if (!tensors || !chunks || step < 1) return ALM_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a{lr, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), max_norm, decoupled_weight_decay, sumsq};
    const int rc = for_each_batch(tensors, ntensors, nchunks, [&](const OptBatch& tb, int c0, int nc) {
        hipLaunchKernelGGL(adam_kernel, dim3(nc), dim3(256), 0, (hipStream_t)stream, tb, reinterpret_cast<const int2*>(chunks) + c0, a);
    });
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// jobs: HOST array (device pointers inside), any count (PACK_BATCH per launch).  Each job: one fp32 weight matrix view [rows][cols] (row stride ld) of
// parameter / gradient / both moments + the destinations of alm_pack_weights_multi.  Requirements (ALM_ERR_BAD_ARG otherwise: the caller then takes
// alm_opt_adam_step + a separate pack): even ld / ld_dst / ld_dstT / rows_pad / cols_pad, 8-byte aligned fp32 bases, 4-byte aligned bf16 bases.
extern "C" int alm_opt_adam_pack_step(const AlmOptPackJob* jobs, int njobs, float lr, float beta1, float beta2, float eps, int step, int decoupled_weight_decay,
                                      const float* sumsq, float max_norm, void* stream) {
    if (njobs <= 0) return 0;
    if (!jobs || step < 1) return ALM_ERR_BAD_ARG;
    for (int j = 0; j < njobs; ++j) {
        const AlmOptPackJob& q = jobs[j];
        if (!q.p || !q.g || !q.m || !q.v || q.rows <= 0 || q.cols <= 0 || q.ld < q.cols || q.rows_pad < q.rows || q.cols_pad < q.cols) return ALM_ERR_BAD_ARG;
        if ((q.dst && q.ld_dst < q.cols_pad) || (q.dstT && q.ld_dstT < q.rows_pad)) return ALM_ERR_BAD_ARG;
        if (((q.ld | q.ld_dst | q.ld_dstT | q.rows_pad | q.cols_pad) & 1) || (((uintptr_t)q.p | (uintptr_t)q.g | (uintptr_t)q.m | (uintptr_t)q.v) & 7) ||
            (((uintptr_t)q.dst | (uintptr_t)q.dstT) & 3))
            return ALM_ERR_BAD_ARG;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a{lr, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), max_norm, decoupled_weight_decay, sumsq};
    for (int j0 = 0; j0 < njobs; j0 += PACK_BATCH) {
        OptPackBatch pb{};
        const int nj = njobs - j0 < PACK_BATCH ? njobs - j0 : PACK_BATCH;
        int total = 0;
        for (int j = 0; j < nj; ++j) {
            pb.job[j] = jobs[j0 + j];
            total += ((pb.job[j].cols_pad + 127) / 128) * ((pb.job[j].rows_pad + 63) / 64);
            pb.tile_end[j] = total;
        }
        pb.njobs = nj;
        hipLaunchKernelGGL(adam_pack_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, pb, a);
    }
    ALM_LAUNCH_CHECK();
    return 0;
}
