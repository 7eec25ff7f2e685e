// Causal multi-query flash attention, forward + backward, for gfx950 (MI355X).
//
// Replaces reference attend.py:98-146 (einsum -> mask fills -> softmax -> einsum) / attend.py:69-96 (SDPA) as called from
// audiolm_pytorch.py:381-394: q (b, n, h*64), ONE shared k / v head of width 64 (MQA, audiolm_pytorch.py:293-294),
// key-padding mask (b, n) bool, causal, scale = 64^-0.5.  The (b, h, n, n) score / probability / bool-mask tensors the
// reference materialises never exist here: online softmax, fp32 statistics, bf16 MFMA operands.
//
// Mapping (wave64, MFMA 32x32x16 bf16):
//   * forward / dQ: one workgroup = one 32-row query block of one batch element, one WAVE PER HEAD (8 waves for h = 8).
//     The K / V tile (64 keys x 64) is loaded from HBM/L2 ONCE per workgroup and shared by all heads through LDS.
//   * scores are computed transposed, S^T = K Q^T, so that one lane owns one query column: the softmax row statistics
//     are lane-local (one cross-half shuffle), and the fp32 S^T accumulator registers ARE the B operand (P^T) of the
//     second MFMA (O^T = V^T P^T) after an in-register bf16 pack: no LDS round trip, no permutes.  The k-index
//     permutation this implies (a lane half holds keys {4h..4h+3} u {8+4h..8+4h+3} of each 16-key step) is applied
//     consistently to the V^T fragment read.
//   * K tile: LDS row-major [key][64], 16-B chunks XOR-swizzled by ((row>>1)&7)  -> conflict-free ds_read_b128.
//     V tile: LDS transposed [d][64 keys (+4 pad)]                                -> conflict-free ds_read_b64.
//   * dK/dV: one workgroup = 32 keys of one batch element, one wave per head, loops over the query blocks at/after the
//     diagonal; the per-head dK^T / dV^T accumulators are summed over heads through LDS (MQA: k/v are shared).
//   * heavy (long-causal-span) blocks are scheduled first.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int VT_LD = 68;    // transposed tile row stride (64 keys + 4 pad) in bf16
constexpr int QT_LD = 36;    // transposed 32-wide tile row stride in bf16

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

__device__ __forceinline__ bf16x8 ld_frag_rows(const bf16_t* base, int idx) { return *reinterpret_cast<const bf16x8*>(base + idx); }

// two 8-byte reads -> one 8 x bf16 fragment (keys/queries {o..o+3} and {o+8..o+11})
__device__ __forceinline__ bf16x8 ld_frag_t(const bf16_t* row_ptr, int o) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(row_ptr + o);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(row_ptr + o + 8);
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(v[8 * s + j]);
    return r;
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = 0;
    return r;
}

__device__ __forceinline__ bf16x8 gload8(const bf16_t* p, bool ok) {
    if (!ok) return zero8();
    return *reinterpret_cast<const bf16x8*>(p);
}

struct AttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const uint8_t* mask;
    bf16_t* o; float* lse;
    const bf16_t* dout; const float* delta;
    bf16_t* dq; float* dk; float* dv;
    long long ldq, ldk, ldv, ldo, lddo, lddq, lddk;
    int B, N, H;
    float scale;
};

// key index (within a 32-key block) held by accumulator register r of a lane in half lh
__device__ __forceinline__ int drow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mqa_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[2][64 * VT_LD];

    const int nqb = (p.N + 31) / 32;
    const int qb = nqb - 1 - blockIdx.x;          // heavy blocks first
    const int b = blockIdx.y;
    const int q0 = qb * 32;
    const int t = threadIdx.x, nthreads = blockDim.x;
    const int lane = t & 63, h = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const float sl2 = p.scale * LOG2E;

    const bf16_t* kb_ptr = p.k + (long long)b * p.N * p.ldk;
    const bf16_t* vb_ptr = p.v + (long long)b * p.N * p.ldv;
    const uint8_t* mrow = p.mask ? p.mask + (long long)b * p.N : nullptr;

    // Q^T fragments (B operand): lane = query column, k = head-dim
    const int qi = q0 + lr;
    const bool qok = qi < p.N;
    bf16x8 qf[4];
    {
        const bf16_t* qp = p.q + ((long long)b * p.N + qi) * p.ldq + h * DH;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = gload8(qp + ks * 16 + lh * 8, qok);
    }

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int ntiles = min((q0 + 31) / 64 + 1, (p.N + 63) / 64);

    // cooperative K / V tile staging: 512 16-byte chunks per operand per tile
    uint4 rk, rv;
    auto load_kv = [&](int tile) {
        rk = rv = make_uint4(0, 0, 0, 0);
        if (t < 512) {
            const int key = tile * 64 + (t >> 3), c = t & 7;
            if (key < p.N) {
                rk = *reinterpret_cast<const uint4*>(kb_ptr + (long long)key * p.ldk + c * 8);
                rv = *reinterpret_cast<const uint4*>(vb_ptr + (long long)key * p.ldv + c * 8);
            }
        }
    };
    auto store_kv = [&](int buf) {
        if (t < 512) {
            const int key = t >> 3, c = t & 7;
            *reinterpret_cast<uint4*>(&Ks[buf][kswz(key, c)]) = rk;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[buf][(c * 8 + j) * VT_LD + key] = e[j];
        }
    };
    // with fewer than 8 heads (< 512 threads) every thread stages several chunks
    auto stage_small = [&](int tile, int buf) {
        for (int cidx = t; cidx < 512; cidx += nthreads) {
            const int key = cidx >> 3, c = cidx & 7;
            const int gk = tile * 64 + key;
            uint4 a = make_uint4(0, 0, 0, 0), bq = make_uint4(0, 0, 0, 0);
            if (gk < p.N) {
                a = *reinterpret_cast<const uint4*>(kb_ptr + (long long)gk * p.ldk + c * 8);
                bq = *reinterpret_cast<const uint4*>(vb_ptr + (long long)gk * p.ldv + c * 8);
            }
            *reinterpret_cast<uint4*>(&Ks[buf][kswz(key, c)]) = a;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&bq);
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[buf][(c * 8 + j) * VT_LD + key] = e[j];
        }
    };
    const bool big = nthreads >= 512;

    if (big) { load_kv(0); store_kv(0); } else stage_small(0, 0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const int key0 = tile * 64;
        if (big && tile + 1 < ntiles) load_kv(tile + 1);

        // key-validity bits of this tile (bit i <-> key0 + i), wave-uniform
        unsigned long long mbits;
        {
            const int kk = key0 + lane;
            bool ok = kk < p.N;
            if (ok && mrow) ok = mrow[kk] != 0;
            mbits = __ballot(ok);
        }

        // S^T = K Q^T  (keys x queries), two 32-key blocks
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = ld_frag_rows(Ks[buf], kswz(kb * 32 + lr, ks * 2 + lh));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
            }
        }
        // mask + online softmax (lane owns query qi; its partner lane^32 owns the other half of the keys)
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kb * 32 + drow(r, lh);
                const bool ok = ((mbits >> kl) & 1ull) && (key0 + kl <= qi);
                const float s = ok ? st[kb][r] * sl2 : -INFINITY;
                st[kb][r] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = exp2f(m - msafe);          // m == -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(st[kb][r] - msafe);
                st[kb][r] = pv;
                psum += pv;
            }
        l = l * alpha + psum;
        m = mnew;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack8(st[kb], s);
#pragma unroll
                for (int di = 0; di < 2; ++di) {
                    const bf16x8 vf = ld_frag_t(&Vt[buf][(di * 32 + lr) * VT_LD], kb * 32 + s * 16 + lh * 4);
                    o[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[di], 0, 0, 0);
                }
            }

        if (tile + 1 < ntiles) {
            if (big) store_kv(buf ^ 1); else stage_small(tile + 1, buf ^ 1);
        }
        __syncthreads();
    }

    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (qok) {
        bf16_t* op = p.o + ((long long)b * p.N + qi) * p.ldo + h * DH;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = di * 32 + 8 * g + 4 * lh;
                *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf2(o[di][4 * g] * inv, o[di][4 * g + 1] * inv),
                                                               pack_bf2(o[di][4 * g + 2] * inv, o[di][4 * g + 3] * inv));
            }
        if (lh == 0) p.lse[((long long)b * p.H + h) * p.N + qi] = (lt > 0.f) ? (m / LOG2E + logf(lt)) : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// delta[b][h][q] = sum_d dO * O     (one wave per (b, q) row; 8 lanes per head)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, long long ldo, const bf16_t* __restrict__ dout,
                                                         long long lddo, float* __restrict__ delta, int B, int N, int H) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * N) return;
    const int b = (int)(row / N), qi = (int)(row % N);
    for (int c = lane; c < H * 8; c += 64) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(o + row * ldo + c * 8);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + row * lddo + c * 8);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf2f((bf16_t)a[j]) * bf2f((bf16_t)d[j]);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if ((lane & 7) == 0) delta[((long long)b * H + (c >> 3)) * N + qi] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward dQ: same decomposition as forward.  dQ^T = scale * K^T dS^T,  dS^T = P^T o (dP^T - delta),  dP^T = V dO^T.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mqa_bwd_dq_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * 64];     // K natural  (A operand of K Q^T)
    __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * 64];     // V natural  (A operand of V dO^T)
    __shared__ __attribute__((aligned(16))) bf16_t Kt[64 * VT_LD];  // K transposed (A operand of K^T dS^T)

    const int nqb = (p.N + 31) / 32;
    const int qb = nqb - 1 - blockIdx.x;
    const int b = blockIdx.y;
    const int q0 = qb * 32;
    const int t = threadIdx.x, nthreads = blockDim.x;
    const int lane = t & 63, h = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const float sl2 = p.scale * LOG2E;

    const bf16_t* kb_ptr = p.k + (long long)b * p.N * p.ldk;
    const bf16_t* vb_ptr = p.v + (long long)b * p.N * p.ldv;
    const uint8_t* mrow = p.mask ? p.mask + (long long)b * p.N : nullptr;

    const int qi = q0 + lr;
    const bool qok = qi < p.N;
    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qp = p.q + ((long long)b * p.N + qi) * p.ldq + h * DH;
        const bf16_t* dp = p.dout + ((long long)b * p.N + qi) * p.lddo + h * DH;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = gload8(qp + ks * 16 + lh * 8, qok);
            dof[ks] = gload8(dp + ks * 16 + lh * 8, qok);
        }
    }
    const float lse2 = qok ? p.lse[((long long)b * p.H + h) * p.N + qi] * LOG2E : INFINITY;
    const float dlt = qok ? p.delta[((long long)b * p.H + h) * p.N + qi] : 0.f;

    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    const int ntiles = min((q0 + 31) / 64 + 1, (p.N + 63) / 64);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int key0 = tile * 64;
        __syncthreads();
        for (int cidx = t; cidx < 512; cidx += nthreads) {
            const int key = cidx >> 3, c = cidx & 7;
            const int gk = key0 + key;
            uint4 a = make_uint4(0, 0, 0, 0), bq = make_uint4(0, 0, 0, 0);
            if (gk < p.N) {
                a = *reinterpret_cast<const uint4*>(kb_ptr + (long long)gk * p.ldk + c * 8);
                bq = *reinterpret_cast<const uint4*>(vb_ptr + (long long)gk * p.ldv + c * 8);
            }
            *reinterpret_cast<uint4*>(&Ks[kswz(key, c)]) = a;
            *reinterpret_cast<uint4*>(&Vs[kswz(key, c)]) = bq;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&a);
#pragma unroll
            for (int j = 0; j < 8; ++j) Kt[(c * 8 + j) * VT_LD + key] = e[j];
        }
        __syncthreads();

        unsigned long long mbits;
        {
            const int kk = key0 + lane;
            bool ok = kk < p.N;
            if (ok && mrow) ok = mrow[kk] != 0;
            mbits = __ballot(ok);
        }

        f32x16 st[2], dpt[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] = 0.f; dpt[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = ld_frag_rows(Ks, kswz(kb * 32 + lr, ks * 2 + lh));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
                const bf16x8 vf = ld_frag_rows(Vs, kswz(kb * 32 + lr, ks * 2 + lh));
                dpt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dpt[kb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kb * 32 + drow(r, lh);
                const bool ok = ((mbits >> kl) & 1ull) && (key0 + kl <= qi);
                const float pv = ok ? exp2f(st[kb][r] * sl2 - lse2) : 0.f;
                st[kb][r] = pv * (dpt[kb][r] - dlt);      // dS^T (unscaled)
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsf = pack8(st[kb], s);
#pragma unroll
                for (int di = 0; di < 2; ++di) {
                    const bf16x8 ktf = ld_frag_t(&Kt[(di * 32 + lr) * VT_LD], kb * 32 + s * 16 + lh * 4);
                    dq[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[di], 0, 0, 0);
                }
            }
    }

    if (qok) {
        bf16_t* op = p.dq + ((long long)b * p.N + qi) * p.lddq + h * DH;
        const float sc = p.scale;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = di * 32 + 8 * g + 4 * lh;
                *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf2(dq[di][4 * g] * sc, dq[di][4 * g + 1] * sc),
                                                               pack_bf2(dq[di][4 * g + 2] * sc, dq[di][4 * g + 3] * sc));
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward dK / dV: workgroup = 32 keys of one batch element; wave = head; loop over query blocks >= diagonal.
//   S = Q K^T (q x keys),  P = exp2(S*sl2 - lse2[q]),  dP = dO V^T,  dS = P o (dP - delta[q])
//   dV^T (d x keys) += dO^T P ;  dK^T (dh x keys) += scale * Q^T dS ;  then summed over heads through LDS.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mqa_bwd_dkv_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int H = p.H;
    // per wave: Qt[64][QT_LD], dOt[64][QT_LD] bf16 ; after the loop the region is reused as fp32 [H][64*32]
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

    const int kblk = blockIdx.x;
    const int b = blockIdx.y;
    const int key0 = kblk * 32;
    const int t = threadIdx.x;
    const int lane = t & 63, h = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const float sl2 = p.scale * LOG2E;

    bf16_t* Qt = smem + (size_t)h * (2 * 64 * QT_LD);
    bf16_t* dOt = Qt + 64 * QT_LD;

    const int key = key0 + lr;
    bool kvalid = key < p.N;
    if (kvalid && p.mask) kvalid = p.mask[(long long)b * p.N + key] != 0;

    // K^T / V^T fragments (B operands): lane = key column, k = head-dim
    bf16x8 kf[4], vf[4];
    {
        const bf16_t* kp = p.k + ((long long)b * p.N + key) * p.ldk;
        const bf16_t* vp = p.v + ((long long)b * p.N + key) * p.ldv;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = gload8(kp + ks * 16 + lh * 8, key < p.N);
            vf[ks] = gload8(vp + ks * 16 + lh * 8, key < p.N);
        }
    }

    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[i][r] = 0.f; dvt[i][r] = 0.f; }

    const int nqb = (p.N + 31) / 32;
    for (int qb = kblk; qb < nqb; ++qb) {
        const int q0 = qb * 32;
        const int qi = q0 + lr;
        const bool qok = qi < p.N;
        // A operands: Q / dO rows (lane = query row, k = head-dim)
        bf16x8 qa[4], da[4];
        {
            const bf16_t* qp = p.q + ((long long)b * p.N + qi) * p.ldq + h * DH;
            const bf16_t* dp = p.dout + ((long long)b * p.N + qi) * p.lddo + h * DH;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qa[ks] = gload8(qp + ks * 16 + lh * 8, qok);
                da[ks] = gload8(dp + ks * 16 + lh * 8, qok);
            }
        }
        __syncthreads();     // previous iteration's transposed tiles fully consumed
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = ks * 16 + lh * 8 + j;
                Qt[d * QT_LD + lr] = (bf16_t)qa[ks][j];
                dOt[d * QT_LD + lr] = (bf16_t)da[ks][j];
            }
        __syncthreads();

        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[ks], dp, 0, 0, 0);
        }
        // rows of S are queries q0 + drow(r, lh); column = this lane's key
        const float* lsep = p.lse + ((long long)b * H + h) * p.N;
        const float* dltp = p.delta + ((long long)b * H + h) * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + drow(r, lh);
            const bool ok = kvalid && qq < p.N && key <= qq;
            float pv = 0.f, dsv = 0.f;
            if (ok) {
                pv = exp2f(s[r] * sl2 - lsep[qq] * LOG2E);
                dsv = pv * (dp[r] - dltp[qq]);
            }
            s[r] = pv;
            dp[r] = dsv;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 pf = pack8(s, st);
            const bf16x8 dsf = pack8(dp, st);
#pragma unroll
            for (int di = 0; di < 2; ++di) {
                const bf16x8 dotf = ld_frag_t(&dOt[(di * 32 + lr) * QT_LD], st * 16 + lh * 4);
                dvt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dvt[di], 0, 0, 0);
                const bf16x8 qtf = ld_frag_t(&Qt[(di * 32 + lr) * QT_LD], st * 16 + lh * 4);
                dkt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dkt[di], 0, 0, 0);
            }
        }
    }

    // reduce over heads through LDS: red[h][d][key] fp32 (64 x 32 per head)
    float* red = reinterpret_cast<float*>(smem_raw);
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        const f32x16* acc = pass == 0 ? dkt : dvt;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(size_t)h * 2048 + (di * 32 + drow(r, lh)) * 32 + lr] = acc[di][r];
        __syncthreads();
        float* outp = pass == 0 ? p.dk : p.dv;
        const float sc = pass == 0 ? p.scale : 1.f;
        for (int e = t; e < 2048; e += blockDim.x) {
            const int kk = e >> 6, d = e & 63;          // output element (key kk, dim d): coalesced along d
            float sum = 0.f;
            for (int hh = 0; hh < H; ++hh) sum += red[(size_t)hh * 2048 + d * 32 + kk];
            if (key0 + kk < p.N) outp[((long long)b * p.N + key0 + kk) * p.lddk + d] = sum * sc;
        }
    }
}

}  // namespace

static int check_attn(int B, int N, int H, long long ldq, long long ldk, long long ldv, long long ldo) {
    if (B <= 0 || N <= 0 || H <= 0 || H > 8) return ALM_ERR_UNSUPPORTED;
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return ALM_ERR_BAD_ARG;
    return 0;
}

extern "C" int alm_mqa_attn_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                const unsigned char* mask, void* o, long long ldo, float* lse, int B, int N, int H, int dim_head,
                                float scale, void* stream) {
    if (dim_head != DH) return ALM_ERR_UNSUPPORTED;
    int rc = check_attn(B, N, H, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    AttnParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.mask = mask; p.o = (bf16_t*)o; p.lse = lse;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipLaunchKernelGGL(mqa_fwd_kernel, dim3((N + 31) / 32, B), dim3(64 * H), 0, (hipStream_t)stream, p);
    ALM_LAUNCH_CHECK();
    return 0;
}

// dk / dv: fp32 [B*N][lddk] (64 valid columns each).  delta: fp32 workspace [B][H][N].
extern "C" int alm_mqa_attn_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                const unsigned char* mask, const void* o, long long ldo, const float* lse, const void* dout, long long lddo,
                                void* dq, long long lddq, float* dk, float* dv, long long lddk, float* delta, int B, int N, int H,
                                int dim_head, float scale, void* stream) {
    if (dim_head != DH) return ALM_ERR_UNSUPPORTED;
    int rc = check_attn(B, N, H, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    if ((lddo & 7) || (lddq & 3)) return ALM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(((long long)B * N + 3) / 4), dim3(256), 0, st, (const bf16_t*)o, ldo, (const bf16_t*)dout, lddo,
                       delta, B, N, H);
    AttnParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.mask = mask; p.lse = const_cast<float*>(lse);
    p.dout = (const bf16_t*)dout; p.delta = delta; p.dq = (bf16_t*)dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk;
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipLaunchKernelGGL(mqa_bwd_dq_kernel, dim3((N + 31) / 32, B), dim3(64 * H), 0, st, p);
    const size_t per_wave = 2 * 64 * QT_LD * sizeof(bf16_t);
    size_t smem = (size_t)H * per_wave;
    const size_t red = (size_t)H * 2048 * sizeof(float);
    if (red > smem) smem = red;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(mqa_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(mqa_bwd_dkv_kernel, dim3((N + 31) / 32, B), dim3(64 * H), smem, st, p);
    ALM_LAUNCH_CHECK();
    return 0;
}
