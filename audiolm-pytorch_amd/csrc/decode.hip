// Single-query multi-query attention over a key/value cache: the attention of ONE new token per sequence during autoregressive sampling
// (reference Attention.forward with kv_cache, audiolm_pytorch.py:360-394, as driven by the generate() loops :1476-1507, :1677-1706,
// :1965-1994).  gfx950 (MI355X).  SURVEY.md §8(f) item 2.
//
//   cache  bf16 [B][Nmax][2 * 64]  (k | v of every earlier position; v already value-residual mixed, audiolm_pytorch.py:357-358)
//   kv_new bf16 [B][2 * 64]        the new position's k | v: appended to the cache at index `pos` by this kernel
//   q      bf16 [B][H * 64]        the new position's queries;  out bf16 [B][H * 64]
// One workgroup per sequence, one wave per head (MQA: the heads share the K / V rows, which therefore stay in L1 / L2).  Per 64-key
// chunk: lane = key computes q . k (+ the structured score bias of the flash_attn=False models, indexed exactly like attention.hip),
// online softmax with wave-uniform statistics, then lane = head dim accumulates sum_j p_j v[j][d] with the probabilities passed through
// LDS.  HBM-bound and tiny (n * 256 B per sequence and layer): the point is that a sampling step costs O(n), not a full forward.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int DH = 64;

struct DecodeArgs {
    const bf16_t* q; long long ldq;
    bf16_t* cache; long long cache_stride;       // elements between sequences (= Nmax * 128)
    const bf16_t* kv_new; long long ldkv;
    const uint8_t* mask; long long ldm;          // [B][>= pos + 1] key mask (1 = attend) or NULL
    bf16_t* out; long long ldo;
    const float* tbl; int LT; int qkey4, qattr;  // structured bias (tbl NULL = none): this query's offset / attribute, per-key vectors below
    const int* kkey4; const int* kattr;
    int B, H, pos;
    float scale;
    const int* pos_dev;                          // non-NULL: the position index lives on the device (hipGraph replay: the launch arguments are frozen)
    const int* qkey4_vec; const int* qattr_vec;  // with pos_dev: per-position query offsets / attributes, indexed by the position
    int nmax;
};

__global__ __launch_bounds__(1024) void mqa_decode_kernel(DecodeArgs a) {
    __shared__ float qs[16][DH];
    __shared__ float ps[16][64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    if (a.pos_dev) {
        a.pos = *a.pos_dev;
        if (a.pos < 0 || a.pos >= a.nmax) return;
        if (a.tbl) { a.qkey4 = a.qkey4_vec[a.pos]; a.qattr = a.qattr_vec[a.pos]; }
    }
    bf16_t* cb = a.cache + (long long)b * a.cache_stride;
    const bf16_t* kvn = a.kv_new + (long long)b * a.ldkv;
    // append the new row (wave 0 of the workgroup; the loop below takes the new row from kv_new directly, so no read-after-write)
    if (h == 0) {
        reinterpret_cast<uint32_t*>(cb + (long long)a.pos * 2 * DH)[lane] = reinterpret_cast<const uint32_t*>(kvn)[lane];
    }
    if (h >= a.H) return;
    qs[h][lane] = bf2f(a.q[(long long)b * a.ldq + h * DH + lane]);
    __builtin_amdgcn_s_waitcnt(0xc07f);                                     // lgkmcnt(0): this wave's own LDS writes (wave-private rows)
    const float* tb = a.tbl ? a.tbl + (long long)h * a.LT : nullptr;
    const uint8_t* mrow = a.mask ? a.mask + (long long)b * a.ldm : nullptr;
    const int n = a.pos + 1;
    float m = -INFINITY, l = 0.f, acc = 0.f;                                // acc: output dim `lane`
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j < n && (!mrow || mrow[j])) {
            const bf16_t* kr = (j == a.pos) ? kvn : cb + (long long)j * 2 * DH;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(qs[h][c * 8 + e], bf2f((bf16_t)kv[e]), d);
            }
            if (tb) {
                const int special = a.qattr & a.kattr[j];
                const unsigned slot = special ? 0u : (unsigned)(a.qkey4 - a.kkey4[j]) >> 2;
                d += slot < (unsigned)a.LT ? tb[slot] : 0.f;
            }
            s = d * a.scale;
        }
        const float cm = wave_max(s);
        const float mn = fmaxf(m, cm);
        const float alpha = (mn == -INFINITY) ? 1.f : __expf(m - mn);      // m == -inf: nothing accumulated yet
        const float p = (s == -INFINITY) ? 0.f : __expf(s - mn);
        l = l * alpha + wave_sum(p);
        acc *= alpha;
        m = mn;
        ps[h][lane] = p;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int cnt = min(64, n - j0);
        for (int jj = 0; jj < cnt; ++jj) {
            const int jg = j0 + jj;
            const bf16_t* vr = (jg == a.pos) ? kvn + DH : cb + (long long)jg * 2 * DH + DH;
            acc = fmaf(ps[h][jj], bf2f(vr[lane]), acc);
        }
    }
    a.out[(long long)b * a.ldo + h * DH + lane] = f2bf(l > 0.f ? acc / l : 0.f);
}

}  // namespace

// One new position per sequence: appends kv_new to the cache at `pos` and attends over positions 0 .. pos.
// tbl / kkey4 / kattr (device) + qkey4 / qattr (the new position's values, by value): structured bias as in alm_mqa_attn_bias_fwd, or tbl NULL.
// pos_dev (device int32) non-NULL: the position is read from it at run time (`pos`, `qkey4`, `qattr` are ignored; qkey4_vec / qattr_vec are
// the per-position device vectors) -- the form a captured hipGraph replays.
extern "C" int alm_mqa_decode_attn(const void* q, long long ldq, void* cache, long long cache_stride, const void* kv_new, long long ldkv,
                                   const unsigned char* mask, long long ldm, void* out, long long ldo, int B, int H, int dim_head, int pos,
                                   int nmax, float scale, const float* tbl, int LT, int qkey4, int qattr, const int* kkey4, const int* kattr,
                                   const int* pos_dev, const int* qkey4_vec, const int* qattr_vec, void* stream) {
    if (dim_head != DH || B <= 0 || H <= 0 || H > 16 || (!pos_dev && (pos < 0 || pos >= nmax))) return ALM_ERR_UNSUPPORTED;
    if (pos_dev && tbl && (!qkey4_vec || !qattr_vec)) return ALM_ERR_BAD_ARG;
    if ((ldkv & 7) || ((uintptr_t)kv_new & 15) || ((uintptr_t)cache & 15) || (cache_stride & 7)) return ALM_ERR_BAD_ARG;
    if (tbl && (!kkey4 || !kattr || LT < 1)) return ALM_ERR_BAD_ARG;
    DecodeArgs a{(const bf16_t*)q, ldq, (bf16_t*)cache, cache_stride, (const bf16_t*)kv_new, ldkv, mask, ldm, (bf16_t*)out, ldo,
                 tbl, LT, qkey4, qattr, kkey4, kattr, B, H, pos, scale, pos_dev, qkey4_vec, qattr_vec, nmax};
    hipLaunchKernelGGL(mqa_decode_kernel, dim3(B), dim3(64 * (H < 1 ? 1 : H)), 0, (hipStream_t)stream, a);
    ALM_LAUNCH_CHECK();
    return 0;
}
