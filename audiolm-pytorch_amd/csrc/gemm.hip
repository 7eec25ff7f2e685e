// bf16 MFMA GEMMs for gfx950 (MI355X): the dense-contraction workhorse of the token-transformer hot path.
//
//   NT:  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T (+ bias[N])     both operands K-contiguous: forward and dgrad of every
//        nn.Linear / einsum (to_q / to_kv / to_out, the two FFN projections, the logit heads).
//        Replaces aten::mm / addmm / bmm at reference audiolm_pytorch.py:255-259, :351, :395, :719, :961, :972.
//   TN:  C[M,N] (+)= alpha * At[K,M]^T . Bt[K,N]               both operands contraction-major (K = tokens): every weight
//        gradient dW = dY^T X straight from the row-major activations -- no transposed copies of dY / X are ever made.
//
// Design (wave64 / CDNA4):
//   * block tile 128x128 (4 waves, 2x2), 256x256 (8 waves as 2(M) x 4(N), wave tile 128x64; the production form staggers the two wave
//     rows by half a K-step, gemm_stag_kernel) or 384x256 (wave tile 192x64), K-step 64, MFMA 32x32x16 bf16.
//     The accumulators hold C^T blocks (operands swapped in the MFMA) so that a lane owns ONE output row and 4 consecutive
//     output columns per register quad: the epilogue stores 8-byte (bf16) / 16-byte (fp32) vectors.
//   * operands go HBM/L2 -> LDS by DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write pass.  Out-of-range
//     K chunks / rows are redirected to an out-of-bounds buffer offset, for which the DMA writes zeros.  Two LDS stages,
//     ONE barrier per K-step: the next tile's DMA is issued before the MFMA block of the current tile.
//   * NT LDS image: [row][64 k] (128-B rows); the DMA destination is lane-linear, so the conflict-avoiding XOR swizzle
//     chunk ^= (row >> 1) & 7 is applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads.
//   * TN LDS image: [k/4][i/16][4][16] sub-tiles of 128 B; a 16-lane group of `ds_read_b64_tr_b16` (LDS transpose read)
//     turns one sub-tile into 4 k-consecutive values of 16 rows = half an MFMA A/B fragment.  Every half-wave reads 256
//     contiguous bytes: bank-conflict free, and the DMA still fetches whole 128-B/256-B row segments from HBM.
//   * XCD-aware block remap + grouped rasterisation: the blocks resident on one XCD share A / B panels in that XCD's L2.
//   * split-K (weight gradients: K = B*N tokens, few output tiles): blockIdx.y = K-slice, fp32 partial tiles in a workspace,
//     deterministic second-stage reduction (no atomics).
//   * two-level batch (blockIdx.y -> (z1, z2)) with element strides for the per-quantizer logit heads
//     (einsum 'q c d, b n q d -> b n q c').
// Requirements: NT: K % 8 == 0, lda/ldb % 8 == 0;  TN: lda/ldb % 8 == 0;  A/B 16-byte aligned; every operand view < 2 GiB.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "../../include/audiolm_hip.h"

// Cache policy of the operand DMA (`buffer_load ... lds` aux immediate: 1 = sc0, 2 = nt, 16 = sc1).  0 in the product build; scripts/build_variant.sh
// builds measurement libraries with -DALM_GEMM_AUX_A=2 / -DALM_GEMM_AUX_B=2 (one operand streamed non-temporally past the L2: round 6, DESIGN labbook)
#ifndef ALM_GEMM_AUX_A
#define ALM_GEMM_AUX_A 0
#endif
#ifndef ALM_GEMM_AUX_B
#define ALM_GEMM_AUX_B 0
#endif

namespace {

constexpr int BK = 64;
constexpr unsigned OOB = 0x80000000u;         // buffer offset beyond every num_records: the DMA returns zeros

typedef __attribute__((address_space(3))) void lds_void;

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const float* bias;
    int M, N, K;
    long long lda, ldb, ldc;
    int nb2;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha;
    int accumulate;
    int ksplit;        // > 0: split-K -- blockIdx.z is the K-slice index, slice s covers k in [s*ksplit, min(K, (s+1)*ksplit)) and
                       // writes its fp32 partial tile to C + s * sCk (reduced afterwards by splitk_reduce_kernel)
    long long sCk;
    int raster;        // 1: split-K launches -- 1-D grid, XCD-panel rasterisation (see kernel): panel q on XCD q % 8.  2: the same with panel q on
                       // XCD q / (P / 8) -- consecutive panels are the row blocks of ONE problem, so an XCD then needs the small operand of 2-3
                       // problems instead of every problem's (batched weight gradients: the small operand was fetched by all 8 XCD L2s)
    int nsl;           // number of K slices (raster 1)
    int nbt = 0;       // total number of batched problems nb1 * nb2 (0: nb2 -- the one-level callers); raster 1 enumerates panels over all of them
    int plimit = 0;    // raster 1: > 0 = launch only the first `plimit` panels (alm_gemm_bf16_tn_batched: whole waves at full K, the tail separately)
    int group_m = 8;   // raster 0: the logical grid is walked in groups of `group_m` tile rows (all tile columns of a group before the next group); an XCD
                       // runs a contiguous chunk of that order, ~32 tiles at a time = group_m x (32 / group_m) tiles sharing A / B panels in its L2.
                       // < 0: groups of |group_m| tile COLUMNS instead (the B panel is the resident one).  Chosen per launch by pick_group()
    int nt_store = 0;  // 1: the C tile is written with non-temporal stores (does not displace the operand panels in the L2)
    // in-launch split-K (round 6, gemm_stag_inl_kernel): blockIdx.z = K slice (ksplit elements each); every slice writes its fp32 accumulators to its slab of
    // `inl_ws`, the LAST ARRIVER of a tile (ticket counter inl_cnt[tile]) sums the slabs in slice order and runs the normal epilogue
    float* inl_ws = nullptr;
    unsigned* inl_cnt = nullptr;
    int inl_slices = 0;
};

// logical block id -> (tile row, tile column): groups of g tile rows (g > 0) or of |g| tile columns (g < 0), the other dimension running fastest
// within a group (see GemmParams::group_m)
__device__ __forceinline__ int grouped_tile(int bid, int tiles_m, int tiles_n, int g) {     // -> tm | tn << 16
    const bool rows = g > 0;
    const int ga = rows ? g : -g;
    const int t_in = rows ? tiles_n : tiles_m, t_gr = rows ? tiles_m : tiles_n;        // tiles along the fast / the grouped dimension
    const int per_group = ga * t_in;
    const int group = bid / per_group, rem = bid % per_group;
    const int first = group * ga;
    const int gsz = min(t_gr - first, ga);
    const int a = first + rem % gsz, b = rem / gsz;                                       // a: grouped dimension, b: the other one
    return rows ? (a | (b << 16)) : (b | (a << 16));
}

// ---- epilogue (shared by every GEMM kernel) --------------------------------------------------------------------------------------------
// lane owns row gm of each 32-row block; register quad g holds columns n = 8*g + 4*lh + {0..3} of each 32-wide block.
// Fast path: every wave transposes its 32 x (32*TNB) block through a private, XOR-swizzled LDS slab (`slabs`: NW slabs, free LDS) and
// writes whole 128-B (bf16) / 256-B (fp32) row segments with 16-byte stores.  Returns after the stores were ISSUED (they drain
// asynchronously).
template <int BM, int BN, int WM, int WN, int TM, int TNB, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TNB], unsigned char* slabs, long long coff0, int m0, int n0,
                                              int wave, int wr, int wc, int lane, int lr, int lh) {
    {
        constexpr int ES = OUT_F32 ? 4 : 2;                 // output element size
        constexpr int WCOLS = 32 * TNB;                     // columns of the wave tile
        constexpr int ROWB = WCOLS * ES;                    // bytes per slab row
        constexpr int NCH = ROWB / 16;                      // 16-B chunks per slab row
        constexpr int SLAB = 32 * ROWB;
        unsigned char* Cb = reinterpret_cast<unsigned char*>(p.C) + coff0 * ES;
        const bool fast = !p.accumulate && ((p.ldc * ES) & 15) == 0 && ((uintptr_t)Cb & 15) == 0 && (p.N % (16 / ES)) == 0;
        if (fast) {
            unsigned char* slab = slabs + wave * SLAB;
            const int ncol0 = n0 + wc * (BN / WN);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow0 = m0 + wr * (BM / WM) + i * 32;
#pragma unroll
                for (int j = 0; j < TNB; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = j * 32 + 8 * g + 4 * lh;
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            v[c] = acc[i][j][4 * g + c] * p.alpha;
                            if (p.bias && ncol0 + c0 + c < p.N) v[c] += p.bias[ncol0 + c0 + c];
                        }
                        if (OUT_F32) {
                            const int ch = c0 / 4;
                            *reinterpret_cast<float4*>(slab + lr * ROWB + ((ch ^ (lr & (NCH - 1))) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            const int ch = c0 / 8;
                            *reinterpret_cast<uint2*>(slab + lr * ROWB + ((ch ^ (lr & (NCH - 1))) << 4) + lh * 8) =
                                make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                        }
                    }
                constexpr int LPR = NCH;                     // lanes per row
                constexpr int RPI = 64 / LPR;                // rows per wave-instruction
#pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int row = it * RPI + lane / LPR, ch = lane % LPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * ROWB + ((ch ^ (row & (NCH - 1))) << 4));
                    const int gm = mrow0 + row, gn = ncol0 + ch * (16 / ES);
#if defined(ALM_GEMM_WHATIF_NOSTORE)                                   // diagnostic build (scripts/gemm_probe.py nostore): WRONG results, timing only
                    if (gm < p.M && gn < p.N && val.x == 0x7fc12345u) *reinterpret_cast<uint4*>(Cb + ((long long)gm * p.ldc + gn) * ES) = val;
#else
                    if (gm < p.M && gn < p.N) {
                        uint4* dst = reinterpret_cast<uint4*>(Cb + ((long long)gm * p.ldc + gn) * ES);
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        if (p.nt_store) __builtin_nontemporal_store(u32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(dst)); else *dst = val;
                    }
#endif
                }
            }
            return;
        }
    }
    const long long coff = coff0;
    const bool vec_ok = OUT_F32 ? ((p.ldc & 3) == 0 && (coff & 3) == 0 && ((uintptr_t)p.C & 15) == 0)
                                : ((p.ldc & 3) == 0 && (coff & 3) == 0 && ((uintptr_t)p.C & 7) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int gm = m0 + wr * (BM / WM) + i * 32 + lr;
        if (gm >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TNB; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gn = n0 + wc * (BN / WN) + j * 32 + 8 * g + 4 * lh;
                if (gn >= p.N) continue;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = acc[i][j][4 * g + c] * p.alpha;
                    if (p.bias && gn + c < p.N) v[c] += p.bias[gn + c];
                }
                const long long idx = coff + (long long)gm * p.ldc + gn;
                if (OUT_F32) {
                    float* C = reinterpret_cast<float*>(p.C) + idx;
                    if (vec_ok && gn + 3 < p.N) {
                        float4 o = make_float4(v[0], v[1], v[2], v[3]);
                        if (p.accumulate) { const float4 w = *reinterpret_cast<const float4*>(C); o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w; }
                        *reinterpret_cast<float4*>(C) = o;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (gn + c < p.N) C[c] = p.accumulate ? C[c] + v[c] : v[c];
                    }
                } else {
                    bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + idx;
                    if (vec_ok && gn + 3 < p.N) {
                        if (p.accumulate) {
                            const uint2 w = *reinterpret_cast<const uint2*>(C);
                            v[0] += __uint_as_float(w.x << 16); v[1] += __uint_as_float(w.x & 0xffff0000u);
                            v[2] += __uint_as_float(w.y << 16); v[3] += __uint_as_float(w.y & 0xffff0000u);
                        }
                        *reinterpret_cast<uint2*>(C) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (gn + c < p.N) C[c] = f2bf(p.accumulate ? bf2f(C[c]) + v[c] : v[c]);
                    }
                }
            }
        }
    }
}

// counted wait on the vector-memory queue: at most N operations (here: LDS-DMA pieces, which retire in order) may still be in flight
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 8 || N == 16, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

// Two LDS stages, one __syncthreads per K-step: the next step's DMA is in flight during the current step's MFMAs (prefetch distance 1).
// (Deeper DMA rings, a hand software-pipelined loop, a persistent variant and B-from-registers variants were built and measured slower:
// they live in the bench-only translation unit csrc/lab/gemm_lab.hip, results in DESIGN.md section 8.1.)
template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32, int NS = 2>
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / WM / 32, TNB = BN / WN / 32;
    constexpr int NIA = BM / 8 / NW, NIB = BN / 8 / NW;       // 1-KiB DMA pieces per wave per stage
    static_assert(NIA >= 1 && NIB >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile / wave shape");

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn, zb, zs;
    if (p.raster >= 1) {
        // split-K weight gradients: the operand with MANY tile panels (e.g. dU: 22 panels of 256 columns) is the big one.  Panel q
        // (its K slices and the few tiles along the other dimension) is pinned to XCD q % 8 (workgroup L runs on XCD L % 8 -- observed
        // dispatch order; a wrong guess only costs speed), so each of its K slices is fetched from HBM once and shared through that
        // XCD's L2; only the small operand is fetched by every XCD.
        const int L = bx, xcd = L & 7, j = L >> 3;
        const bool m_major = tiles_m >= tiles_n;
        const int tmaj = m_major ? tiles_m : tiles_n, Q = m_major ? tiles_n : tiles_m;
        const int P = p.plimit > 0 ? p.plimit : tmaj * (p.nbt > 0 ? p.nbt : p.nb2);   // plimit: only the first panels (hybrid full-K + tail launch)
        const int PL = (P + 7) / 8;
        zs = j / (PL * Q);
        const int rem = j % (PL * Q);
        const int panel = p.raster == 2 ? xcd * PL + rem / Q : (rem / Q) * 8 + xcd;       // raster 2: a contiguous block of panels per XCD (see GemmParams)
        const int minor = rem % Q;
        if (panel >= P || zs >= p.nsl) return;
        zb = panel / tmaj;
        const int tmajor = panel % tmaj;
        tm = m_major ? tmajor : minor;
        tn = m_major ? minor : tmajor;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int bid = xcd_remap(bx, nwg);
        const int tt = grouped_tile(bid, tiles_m, tiles_n, p.group_m);
        tm = tt & 0xffff;
        tn = tt >> 16;
        zb = by;
        zs = bz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    int kbeg = 0, Krem = p.K;
    const long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    if (p.ksplit > 0) {
        kbeg = zs * p.ksplit;
        Krem = min(p.K - kbeg, p.ksplit);
    }

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- DMA source descriptors (block-local views) and per-lane source offsets --------------------------------------------
    const bf16_t* Ab;
    const bf16_t* Bb;
    long long extA, extB;
    if (!TNMODE) {
        Ab = p.A + zoffA + (long long)m0 * p.lda + kbeg;
        Bb = p.B + zoffB + (long long)n0 * p.ldb + kbeg;
        extA = ((long long)(min(p.M - m0, BM) - 1) * p.lda + Krem) * 2;
        extB = ((long long)(min(p.N - n0, BN) - 1) * p.ldb + Krem) * 2;
    } else {
        Ab = p.A + zoffA + (long long)kbeg * p.lda + m0;
        Bb = p.B + zoffB + (long long)kbeg * p.ldb + n0;
        extA = ((long long)(Krem - 1) * p.lda + ((min(p.M - m0, BM) + 7) & ~7)) * 2;     // whole 16-B chunks (lda >= roundup8(M))
        extB = ((long long)(Krem - 1) * p.ldb + ((min(p.N - n0, BN) + 7) & ~7)) * 2;
    }
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)min(extA, 0x7fffffffLL), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)min(extB, 0x7fffffffLL), 0x00020000);

    unsigned offA[NIA], offB[NIB];
    int kcA[NIA], kcB[NIB];      // K coordinate (within the stage) of the piece this lane fetches: for the K-tail predicate
    if (!TNMODE) {
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            offA[j] = (unsigned)(row * p.lda * 2 + c * 16);
            kcA[j] = c * 8;
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            offB[j] = (unsigned)(row * p.ldb * 2 + c * 16);
            kcB[j] = c * 8;
        }
    } else {
        const int st = lane >> 3, kin = (lane >> 1) & 3, half = lane & 1;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int q = j * NW + wave;                       // 1-KiB piece index within the stage
            const int kg = q / (BM / 128), part = q % (BM / 128);
            const int k = kg * 4 + kin, i = part * 128 + st * 16 + half * 8;
            offA[j] = (unsigned)(k * p.lda * 2 + i * 2);
            kcA[j] = k;
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const int q = j * NW + wave;
            const int kg = q / (BN / 128), part = q % (BN / 128);
            const int k = kg * 4 + kin, i = part * 128 + st * 16 + half * 8;
            offB[j] = (unsigned)(k * p.ldb * 2 + i * 2);
            kcB[j] = k;
        }
    }
    const unsigned kstepA = TNMODE ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
    const unsigned kstepB = TNMODE ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);

    auto stage = [&](int kt, unsigned char* base) {
        const int kleft = Krem - kt * BK;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const unsigned vo = (kcA[j] < kleft) ? offA[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, vo, kt * kstepA, 0, ALM_GEMM_AUX_A);
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const unsigned vo = (kcB[j] < kleft) ? offB[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (j * NW + wave) * 1024), 16, vo, kt * kstepB, 0, ALM_GEMM_AUX_B);
        }
    };

    // ---- fragment read addresses (bytes, within a stage) -------------------------------------------------------------------
    unsigned fragA[TM], fragB[TNB];
    if (!TNMODE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((wr * (BM / WM) + i * 32 + lr) * 128);
#pragma unroll
        for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)(A_BYTES + (wc * (BN / WN) + j * 32 + lr) * 128);
    } else {
        const int g = lane >> 4, s = lane & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int it = (wr * (BM / WM) + i * 32) / 16 + (g & 1);
            fragA[i] = (unsigned)((((g >> 1) * 2) * (BM / 16) + it) * 128 + s * 8);
        }
#pragma unroll
        for (int j = 0; j < TNB; ++j) {
            const int it = (wc * (BN / WN) + j * 32) / 16 + (g & 1);
            fragB[j] = (unsigned)(A_BYTES + (((g >> 1) * 2) * (BN / 16) + it) * 128 + s * 8);
        }
    }
    const unsigned sw = (unsigned)((lr >> 1) & 7);          // NT read swizzle: (row >> 1) & 7 == (lr >> 1) & 7 (row bases are multiples of 32)

    f32x16 acc[TM][TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (Krem + BK - 1) / BK;
    if constexpr (NS == 2) {
        stage(0, smem);
        __syncthreads();
    }

    // One K-step: issue the next stage's DMA into `dst`, read this stage's fragments from `sb`.  The two are DISTINCT buffers, and they are
    // `__restrict__` parameters on purpose: inlined, that becomes alias-scope metadata, without which the compiler makes every LDS read that has
    // no type-based alias info of its own -- the `ds_read_b64_tr_b16` transpose reads of the TN mode are such -- wait (`s_waitcnt vmcnt(0)`) for
    // ALL pending LDS-DMA: the DMA just issued.  The weight-gradient loops then ran DMA and MFMAs strictly one after the other.
    auto kstep = [&](unsigned char* __restrict__ dst, const unsigned char* __restrict__ sb, int kt) {
        if (kt + NS - 1 < nk) stage(kt + NS - 1, dst);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[TM], b[TNB];
            if (!TNMODE) {
                const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
#pragma unroll
                for (int j = 0; j < TNB; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + fragB[j] + co);
            } else {
                // k-groups ks*4 + (g>>1)*2 + {0, 1}; consecutive k-groups are (Bx/16)*128 bytes apart
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned char* ad = sb + fragA[i] + ks * 4 * (BM / 16) * 128;
                    a[i] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + (BM / 16) * 128), 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TNB; ++j) {
                    const unsigned char* ad = sb + fragB[j] + ks * 4 * (BN / 16) * 128;
                    b[j] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + (BN / 16) * 128), 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);     // C^T block: lane = row m
        }
    };
    if constexpr (NS == 2) {
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            kstep(smem + (buf ^ 1) * STAGE, smem + buf * STAGE, kt);
            __syncthreads();          // reads of `buf` done (lgkmcnt 0), next stage landed (vmcnt 0), all waves agree
            buf ^= 1;
        }
    } else {
        // DMA RING of NS stages (round 5, the launches that cannot fill the chip: to_kv is 128 workgroups, one per CU at best, and with prefetch distance 1
        // every K-step waited for one HBM / Infinity-Cache round trip -- 16 of them in a row): stage kt + NS - 1 is issued at the top of step kt, the wait
        // before the step's barrier is COUNTED -- at most NS - 2 newer stages (8 pieces each per wave) may still be in flight -- so NS - 2 round trips are
        // always hidden behind the current step.  One raw barrier per step orders both hazards: every wave's pieces of stage kt have landed (its counted
        // wait precedes the barrier) and every wave has finished reading stage kt - 1, whose buffer the DMA issued right after the barrier overwrites.
        static_assert(NIA + NIB == 8 && (NS == 3 || NS == 4), "counted waits below: 8 pieces per wave and stage");
#pragma unroll
        for (int s_ = 0; s_ < NS - 1; ++s_)
            if (s_ < nk) stage(s_, smem + s_ * STAGE);
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const int newer = min(nk - 1 - kt, NS - 2);               // stages issued after stage kt so far
            if (newer >= 2) wait_vmcnt<16>(); else if (newer == 1) wait_vmcnt<8>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            kstep(smem + ((kt + NS - 1) % NS) * STAGE, smem + (kt % NS) * STAGE, kt);
        }
        __syncthreads();              // the epilogue slabs reuse the stage buffers
    }

    // ---- epilogue: slabs reuse the (now idle) stage buffers ------------------------------------------------------------------------
    constexpr int ES = OUT_F32 ? 4 : 2;
    static_assert(NW * 32 * (32 * TNB * ES) <= 2 * STAGE, "epilogue slab");
    const long long coff0 = z1 * p.sC1 + z2 * p.sC2 + (p.ksplit > 0 ? zs * p.sCk : 0);
    gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
}

template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmParams p) {
    gemm_body<BM, BN, WM, WN, TNMODE, OUT_F32>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32, int NS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_ring_kernel(GemmParams p) {
    gemm_body<BM, BN, WM, WN, TNMODE, OUT_F32, NS>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// TWO independent un-batched problems in one launch (round 5): workgroups [0, tiles0) run problem 0, the rest problem 1 -- for pairs of launches that sit
// next to each other in the step and each leave most of the chip idle or latency-bound on their own (to_q || to_kv: the N = 128 projection is 128
// workgroups waiting on one HBM round trip per K-step; dXN_q || dX_kv).  The parameter block is selected by a workgroup-uniform pointer select.
template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32>
__global__ __launch_bounds__(WM * WN * 64) void gemm_group2_kernel(GemmParams p0, GemmParams p1, int tiles0) {
    const bool second = (int)blockIdx.x >= tiles0;
    const GemmParams& p = second ? p1 : p0;
    gemm_body<BM, BN, WM, WN, TNMODE, OUT_F32>(p, second ? (int)blockIdx.x - tiles0 : (int)blockIdx.x, 0, 0);
}

// Bench-only cycle probe of the staggered kernel (scripts/gemm_probe.py builds gemm.hip with -DALM_GEMM_PROBE into its own library; the product build
// has none of this): per wave, s_memtime cycles in the four parts of a slot pair + prologue / epilogue.
#ifdef ALM_GEMM_PROBE
__device__ unsigned long long g_gemm_probe[16384 * 8];
#define GPROBE_DECL unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long plast = __builtin_readcyclecounter();
#define GPROBE_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long pt_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); pacc[i] += pt_ - plast; plast = pt_; } while (0)
#define GPROBE_FLUSH(slot) do { if (lane == 0 && (slot) < 16384) for (int i_ = 0; i_ < 8; ++i_) g_gemm_probe[(slot) * 8 + i_] = pacc[i_]; } while (0)
#else
#define GPROBE_DECL
#define GPROBE_T(i)
#define GPROBE_FLUSH(slot)
#endif

// ---- staggered 256 x 256 x 64 kernel (8 waves, NT and TN): the two wave rows (wr = 0 / 1: the two waves co-resident on each SIMD) run
// HALF A K-STEP APART.  Why: in the lock-step kernel above every wave issues its 8 LDS-DMA instructions at the top of a K-step -- at
// 60-185 issue cycles each (MI355X_MICROARCH: "LDS-DMA piece issue cost") that is ~1000 cycles during which neither wave of the SIMD feeds
// the matrix pipe (measured: MFMA-only loop 601 us, DMA-only 716 us, both 886 us at 8192^3: they do not overlap).
// Here the K-loop is a sequence of half-step SLOTS closed by one workgroup barrier each; in a slot one wave row issues the DMA of a later
// stage (LOAD slot) while the other reads fragments and runs its 32 MFMAs (COMPUTE slot), then they swap:
//     wave row g, slot s, q = s - g:   q even  -> LOAD stage q / 2          q odd -> COMPUTE stage (q - 1) / 2 - 1
// so a stage is loaded 3 slots before the loading row computes on it, its DMA is waited for (vmcnt(0)) at the end of the row's next
// COMPUTE slot -- a whole slot of MFMAs later -- and every wave executes exactly one barrier per slot (2 nk + 3 slots: no wave ever waits
// on a barrier the others skip).
// LDS (160 KB): the A operand is PRIVATE to a wave row (row g only reads tile rows g*128 .. +127): 2 buffers x 16 KB per row; the B
// operand is shared by both rows and lives 5 slots (written in slots 2j, 2j+1, read in 2j+3, 2j+4): 3 buffers x 32 KB.  Row g loads its
// own A half and the B pieces [16 g, 16 g + 16): 8 one-KiB DMA pieces per wave and stage, as before.
template <bool TNMODE, bool OUT_F32, bool INL = false>
__device__ __forceinline__ void gemm_stag_body(const GemmParams& p, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    GPROBE_DECL
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8, TM = 4, TNB = 2;
    constexpr int AH_BYTES = 128 * BK * 2, B_BYTES = BN * BK * 2;           // 16 KB, 32 KB
    constexpr int B_BASE = 4 * AH_BYTES;                                    // [A row 0: buf 0, 1][A row 1: buf 0, 1][B: buf 0, 1, 2]
    static_assert(B_BASE + 3 * B_BYTES == 163840, "LDS map");

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn, zb, zs;
    if (p.raster >= 1) {                                                    // split-K weight gradients: XCD-panel rasterisation (see gemm_kernel)
        const int L = bx, xcd = L & 7, j = L >> 3;
        const bool m_major = tiles_m >= tiles_n;
        const int tmaj = m_major ? tiles_m : tiles_n, Q = m_major ? tiles_n : tiles_m;
        const int P = p.plimit > 0 ? p.plimit : tmaj * (p.nbt > 0 ? p.nbt : p.nb2);   // plimit: only the first panels (hybrid full-K + tail launch)
        const int PL = (P + 7) / 8;
        zs = j / (PL * Q);
        const int rem = j % (PL * Q);
        const int panel = p.raster == 2 ? xcd * PL + rem / Q : (rem / Q) * 8 + xcd;       // raster 2: a contiguous block of panels per XCD (see GemmParams)
        const int minor = rem % Q;
        if (panel >= P || zs >= p.nsl) return;                              // whole workgroup: no barrier is skipped by a subset
        zb = panel / tmaj;
        const int tmajor = panel % tmaj;
        tm = m_major ? tmajor : minor;
        tn = m_major ? minor : tmajor;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int bid = xcd_remap(bx, nwg);
        const int tt = grouped_tile(bid, tiles_m, tiles_n, p.group_m);
        tm = tt & 0xffff;
        tn = tt >> 16;
        zb = by;
        zs = bz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    int kbeg = 0, Krem = p.K;
    const long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    if (p.ksplit > 0) {
        kbeg = zs * p.ksplit;
        Krem = min(p.K - kbeg, p.ksplit);
    }

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;                               // wr = wave row = stagger group
    const int lr = lane & 31, lh = lane >> 5;

    const bf16_t* Ab;
    const bf16_t* Bb;
    long long extA, extB;
    if (!TNMODE) {
        Ab = p.A + zoffA + (long long)m0 * p.lda + kbeg;
        Bb = p.B + zoffB + (long long)n0 * p.ldb + kbeg;
        extA = ((long long)(min(p.M - m0, BM) - 1) * p.lda + Krem) * 2;
        extB = ((long long)(min(p.N - n0, BN) - 1) * p.ldb + Krem) * 2;
    } else {
        Ab = p.A + zoffA + (long long)kbeg * p.lda + m0;
        Bb = p.B + zoffB + (long long)kbeg * p.ldb + n0;
        extA = ((long long)(Krem - 1) * p.lda + ((min(p.M - m0, BM) + 7) & ~7)) * 2;
        extB = ((long long)(Krem - 1) * p.ldb + ((min(p.N - n0, BN) + 7) & ~7)) * 2;
    }
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)min(extA, 0x7fffffffLL), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)min(extB, 0x7fffffffLL), 0x00020000);

    // DMA pieces of this wave: 4 of the row's own A half (16 pieces) and 4 of the row's share of B (pieces 16 wr .. 16 wr + 15 of 32)
    unsigned offA[4], offB[4], dstA[4], dstB[4];
    int kcA[4], kcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int qa = j * 4 + wc;                                          // piece within the A half
        const int qb = wr * 16 + j * 4 + wc;                                // piece within the B tile
        dstA[j] = (unsigned)(qa * 1024);
        dstB[j] = (unsigned)(qb * 1024);
        if (!TNMODE) {
            const int rowa = wr * 128 + qa * 8 + (lane >> 3), rowb = qb * 8 + (lane >> 3);
            const int ca = (lane & 7) ^ ((rowa >> 1) & 7), cb = (lane & 7) ^ ((rowb >> 1) & 7);
            offA[j] = (unsigned)(rowa * p.lda * 2 + ca * 16);
            offB[j] = (unsigned)(rowb * p.ldb * 2 + cb * 16);
            kcA[j] = ca * 8;
            kcB[j] = cb * 8;
        } else {
            const int st = lane >> 3, kin = (lane >> 1) & 3, half = lane & 1;
            const int ka = qa * 4 + kin, ia = wr * 128 + st * 16 + half * 8;          // A half image [k/4][8 x 16 cols][4][16]
            const int kgb = qb / 2, partb = qb % 2;                                    // B image [k/4][16 x 16 cols][4][16]
            const int kb = kgb * 4 + kin, ib = partb * 128 + st * 16 + half * 8;
            offA[j] = (unsigned)(ka * p.lda * 2 + ia * 2);
            offB[j] = (unsigned)(kb * p.ldb * 2 + ib * 2);
            kcA[j] = ka;
            kcB[j] = kb;
        }
    }
    const unsigned kstepA = TNMODE ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
    const unsigned kstepB = TNMODE ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);
    unsigned char* const myA = smem + wr * 2 * AH_BYTES;                    // this wave row's two A-half buffers

    auto stage = [&](int j, unsigned char* ad, unsigned char* bd) {
        const int kleft = Krem - j * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = (kcA[i] < kleft) ? offA[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(ad + dstA[i]), 16, vo, j * kstepA, 0, ALM_GEMM_AUX_A);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = (kcB[i] < kleft) ? offB[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(bd + dstB[i]), 16, vo, j * kstepB, 0, ALM_GEMM_AUX_B);
        }
    };

    // fragment read offsets (bytes): A within the row's half image, B within the B image
    unsigned fragA[TM], fragB[TNB];
    if (!TNMODE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((i * 32 + lr) * 128);
#pragma unroll
        for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)((wc * 64 + j * 32 + lr) * 128);
    } else {
        const int g = lane >> 4, s = lane & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((((g >> 1) * 2) * 8 + (i * 32) / 16 + (g & 1)) * 128 + s * 8);
#pragma unroll
        for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)((((g >> 1) * 2) * 16 + (wc * 64 + j * 32) / 16 + (g & 1)) * 128 + s * 8);
    }
    const unsigned sw = (unsigned)((lr >> 1) & 7);

    f32x16 acc[TM][TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The fragments of K sub-step ks + 1 are read while the 8 MFMAs of sub-step ks run (two register sets, explicit schedule groups): left to itself
    // the compiler keeps ONE 16-register fragment set and issues each read right in front of the MFMA that needs it -- ~100 cycles of LDS latency
    // exposed a dozen times per slot (probe: 1450 cycles per COMPUTE slot for 1024 cycles of MFMA; the other wave of the SIMD is in its LOAD slot
    // and cannot fill the gaps).
    auto load_frags = [&](const unsigned char* sa, const unsigned char* sb, int ks, bf16x8 (&a)[TM], bf16x8 (&b)[TNB]) {
        if (!TNMODE) {
            const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + fragA[i] + co);
#pragma unroll
            for (int j = 0; j < TNB; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + fragB[j] + co);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* ad = sa + fragA[i] + ks * 4 * 8 * 128;
                a[i] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + 8 * 128), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < TNB; ++j) {
                const unsigned char* bd = sb + fragB[j] + ks * 4 * 16 * 128;
                b[j] = __builtin_shufflevector(lds_tr16(bd), lds_tr16(bd + 16 * 128), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto mfma_all = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TNB]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    };
    constexpr int NRD = (TNMODE ? 2 : 1) * (TM + TNB);                     // LDS read instructions per sub-step
    auto compute = [&](const unsigned char* sa, const unsigned char* sb) {
        bf16x8 a0[TM], b0[TNB], a1[TM], b1[TNB];
        load_frags(sa, sb, 0, a0, b0);
        load_frags(sa, sb, 1, a1, b1);
        mfma_all(a0, b0);
        load_frags(sa, sb, 2, a0, b0);
        mfma_all(a1, b1);
        load_frags(sa, sb, 3, a1, b1);
        mfma_all(a0, b0);
        mfma_all(a1, b1);
        // schedule: [reads 0][reads 1] | 8 MFMA (0) interleaved with reads 2 | 8 MFMA (1) interleaved with reads 3 | 8 MFMA | 8 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRD, 0);
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int q = 0; q < TM * TNB; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (NRD > TM * TNB) __builtin_amdgcn_sched_group_barrier(0x100, NRD - TM * TNB, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TNB, 0);
    };

    const int nk = (Krem + BK - 1) / BK;
    // slot pair i = slots 2 i, 2 i + 1 (written out per wave row: q = s - wr); nk + 2 pairs cover the 2 nk + 3 slots.
    // The buffers a pair writes by DMA (dA, dB) and the ones it reads (rA, rB) are distinct, and `__restrict__` parameters on purpose: see the
    // lock-step kernel (without the alias scopes the TN mode's transpose reads wait for the DMA issued one slot earlier before they start).
    auto Abuf = [&](int j) { return myA + (j & 1) * AH_BYTES; };
    auto Bbuf = [&](int j) { return smem + B_BASE + (j % 3) * B_BYTES; };
    // Both rows run the SAME slot pair -- LOAD stage i, COMPUTE stage i - 1 -- row 1 simply starts one slot later (it idles through slot 0 and
    // the last slot).  The DMA pending while a pair computes is always the one the pair itself issued: the alias scopes of one (not unrolled)
    // inlined body cover it.
    auto pair = [&](unsigned char* __restrict__ dA, unsigned char* __restrict__ dB, const unsigned char* __restrict__ rA,
                    const unsigned char* __restrict__ rB, int i) {
        if (i < nk) stage(i, dA, dB);
        GPROBE_T(1);
        __builtin_amdgcn_s_barrier();                                       // raw: this slot's DMA stays in flight across it
        GPROBE_T(2);
        if (i >= 1) compute(rA, rB);                                        // stage i - 1: (i - 1) & 1 == (i + 1) & 1, (i - 1) % 3 == (i + 2) % 3
        GPROBE_T(3);
        __syncthreads();                                                    // vmcnt(0): the DMA issued one slot ago has landed; lgkmcnt(0): reads done
        GPROBE_T(4);
    };
    GPROBE_T(0);
    if (wr == 1) __syncthreads();                                           // slot 0 of row 1
#pragma unroll 1
    for (int i = 0; i <= nk; ++i) pair(Abuf(i), Bbuf(i), Abuf(i + 1), Bbuf(i + 2), i);
    if (wr == 0) __syncthreads();                                           // row 0: slots 2 nk + 2, 2 nk + 3;  row 1: slot 2 nk + 3
    __builtin_amdgcn_s_barrier();

    if constexpr (INL) {
        // ---- in-launch split-K: publish this slice's accumulators, the tile's last arriver reduces (cdna_hip_programming.md, "In-launch split-K reduction";
        // section 6 guideline 16).  Slab layout = the accumulators' own: 16-byte chunk c = (i, j, quad) of thread t at byte (c * 512 + t) * 16 -- every wave
        // instruction moves 1 KiB of contiguous bytes.  Stores are WRITE-THROUGH (sc1: visible in memory without an L2 write-back fence), every wave
        // drains them, the barrier collects the waves, ONE lane takes the ticket (relaxed, agent scope).  The reducer reads the slabs with sc1 loads (served
        // by the L2 / fabric, never by this CU's L1).  Deterministic for any arrival order: 2 slices -> own + other (IEEE addition commutes); more ->
        // every slab, the reducer's own included, is re-read and summed in slice order 0, 1, 2, ...
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        constexpr int SLAB_BYTES = BM * BN * 4, NCHUNK = TM * TNB * 4;
        const int S = p.inl_slices;
        const int tile_lin = zb * (tiles_m * tiles_n) + tm * tiles_n + tn;
        unsigned char* const slab0 = reinterpret_cast<unsigned char*>(p.inl_ws) + (long long)tile_lin * S * SLAB_BYTES;
        {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(slab0 + (long long)zs * SLAB_BYTES, 0, SLAB_BYTES, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32x4 v = {__float_as_uint(acc[i][j][4 * g]), __float_as_uint(acc[i][j][4 * g + 1]), __float_as_uint(acc[i][j][4 * g + 2]),
                                         __float_as_uint(acc[i][j][4 * g + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, t * 16, ((i * TNB + j) * 4 + g) * 8192, 16);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // EVERY storing wave drains its write-through stores ...
        __syncthreads();                                                    // ... before the one ticket is taken
        unsigned* const flag = reinterpret_cast<unsigned*>(smem + 163840 - 16);   // beyond every epilogue slab (64 KB): the one LDS array, no second object
        if (t == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.inl_cnt + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(p.inl_cnt + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning: zero again for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;                                             // whole workgroup (uniform)
        // reduce: 4 chunks (16 VGPRs) of loads in flight at a time -- 128 accumulator registers + the loop state are live (8 chunks spilled)
        auto add_slab = [&](int sl, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;                 // FIRST: acc = slab (the ordered sum starts from slab 0), else acc += slab
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(slab0 + (long long)sl * SLAB_BYTES, 0, SLAB_BYTES, 0x00020000);
#pragma unroll
            for (int cb = 0; cb < NCHUNK; cb += 4) {
                u32x4 v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 16, (cb + c) * 8192, 16);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ai = (cb + c) / (4 * TNB), aj = ((cb + c) / 4) % TNB, ar = 4 * ((cb + c) % 4) + e;
                        acc[ai][aj][ar] = FIRST ? __uint_as_float(v[c][e]) : acc[ai][aj][ar] + __uint_as_float(v[c][e]);
                    }
            }
        };
        if (S == 2) {
            add_slab(1 - zs, std::false_type{});
        } else {
            add_slab(0, std::true_type{});
#pragma unroll 1
            for (int sl = 1; sl < S; ++sl) add_slab(sl, std::false_type{});
        }
    }

    const long long coff0 = z1 * p.sC1 + z2 * p.sC2 + (p.ksplit > 0 ? zs * p.sCk : 0);
    GPROBE_T(5);
    gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
    GPROBE_T(6);
    GPROBE_FLUSH((bx + gridDim.x * (by + gridDim.y * bz)) * 8 + wave);
}


template <bool TNMODE, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_stag_kernel(GemmParams p) {
    gemm_stag_body<TNMODE, OUT_F32>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// in-launch split-K form of the staggered NT tile (round 6): grid (tiles, problems, K slices); see the INL block of gemm_stag_body
template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_stag_inl_kernel(GemmParams p) {
    gemm_stag_body<false, OUT_F32, true>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <bool TNMODE, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_stag_group2_kernel(GemmParams p0, GemmParams p1, int tiles0) {        // see gemm_group2_kernel
    const bool second = (int)blockIdx.x >= tiles0;
    const GemmParams& p = second ? p1 : p0;
    gemm_stag_body<TNMODE, OUT_F32>(p, second ? (int)blockIdx.x - tiles0 : (int)blockIdx.x, 0, 0);
}

// ---- 256 x 256 x 64, FOUR waves (one per SIMD), wave tile 128 x 128 (tile id 14; NT and TN) -----------------------------------------------
// Why another main loop (round 3): the 8-wave tiles above read (4 + 2) 16-byte fragments per 8 MFMAs -- 48 ds_read_b128 per CU for every 512
// matrix-pipe cycles = 75 % of the LDS port (128 B / clk / CU), which is why their MFMAs issue every ~45 instead of 32 cycles (slot probe, DESIGN.md
// section 8.1 (e)) -- and while one wave of a SIMD computes, the other one loads or waits, so nothing fills a wave's own gaps.  A 128 x 128 wave tile
// reads (4 + 4) fragments per 16 MFMAs: 50 % of the LDS port, and its 256 accumulator registers (AGPRs) leave one wave per SIMD with the whole
// 512-entry file.  With ONE wave per SIMD every stall is exposed, so the loop is software-pipelined by hand, in the order the vendor's Tensile kernels
// for this chip use (read off their ISA: MT256x256x64, 4 waves, wave tile 128 x 128, direct-to-LDS, prefetch 2, local-read prefetch 1):
//     K-step t (stage buffers cur = t & 1, nxt = the other), sub-steps S0 .. S3 of 16 MFMAs, fragment register sets F0 .. F3 (F0 read during step t - 1):
//       S0 (F0) || read F1 <- cur || DMA of the B half of tile t + 1 -> nxt      (one 1-KiB piece per TWO MFMAs)
//       S1 (F1) || read F2 <- cur          S2 (F2) || read F3 <- cur
//       vmcnt(0) lgkmcnt(0), barrier: cur is consumed by every wave and tile t + 1 has landed in nxt -- the ONLY barrier of the K-step
//       S3 (F3) || read F0 <- nxt (sub-step 0 of tile t + 1) || DMA of the A half of tile t + 2 -> cur
// so 8 fragment reads ride under every 16 MFMAs, the 16 DMA pieces of a wave are spread over two sub-steps, and the operand fetch runs 1 - 1.75
// K-steps ahead out of two 64-KB stages.  How this schedule was found (in-kernel s_memtime probe, scripts/gemm_probe.py, ALM_PROBE_TILE=14): with all
// 16 pieces issued under S2, one per MFMA, S2 took 1316 cycles for 512 cycles of MFMA -- the CU's vector-memory path moves 64 B / clk, so the 4
// waves' 64 one-KiB pieces occupy it for ~1024 cycles however they are issued; they must be spread over at least that many MFMA cycles (spread over
// S2 + S3: 8192^3 1301 -> 1418 TFLOP/s).  The compiler cannot see the DMA -> LDS -> ds_read dependency through `__restrict__` (it must not: it
// would wait for ALL pending DMA, see gemm_kernel); the counted wait + barrier above is what orders them.
template <bool TNMODE, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    GPROBE_DECL
    constexpr int BM = 256, BN = 256, WM = 2, WN = 2, NW = 4, TM = 4, TNB = 4;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;       // 32 KB + 32 KB
    constexpr int NIA = BM / 8 / NW, NIB = BN / 8 / NW;                                          // 8 + 8 one-KiB DMA pieces per wave and stage
    static_assert(NIA + NIB == 16, "the counted waits below assume 16 pieces per wave and stage");

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn, zb, zs;
    if (p.raster >= 1) {                                                    // split-K weight gradients: XCD-panel rasterisation (see gemm_kernel)
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const bool m_major = tiles_m >= tiles_n;
        const int tmaj = m_major ? tiles_m : tiles_n, Q = m_major ? tiles_n : tiles_m;
        const int P = p.plimit > 0 ? p.plimit : tmaj * (p.nbt > 0 ? p.nbt : p.nb2);   // plimit: only the first panels (hybrid full-K + tail launch)
        const int PL = (P + 7) / 8;
        zs = j / (PL * Q);
        const int rem = j % (PL * Q);
        const int panel = p.raster == 2 ? xcd * PL + rem / Q : (rem / Q) * 8 + xcd;       // raster 2: a contiguous block of panels per XCD (see GemmParams)
        const int minor = rem % Q;
        if (panel >= P || zs >= p.nsl) return;
        zb = panel / tmaj;
        const int tmajor = panel % tmaj;
        tm = m_major ? tmajor : minor;
        tn = m_major ? minor : tmajor;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int bid = xcd_remap(blockIdx.x, nwg);
        const int tt = grouped_tile(bid, tiles_m, tiles_n, p.group_m);
        tm = tt & 0xffff;
        tn = tt >> 16;
        zb = blockIdx.y;
        zs = blockIdx.z;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    int kbeg = 0, Krem = p.K;
    const long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    if (p.ksplit > 0) {
        kbeg = zs * p.ksplit;
        Krem = min(p.K - kbeg, p.ksplit);
    }

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    const bf16_t* Ab;
    const bf16_t* Bb;
    long long extA, extB;
    if (!TNMODE) {
        Ab = p.A + zoffA + (long long)m0 * p.lda + kbeg;
        Bb = p.B + zoffB + (long long)n0 * p.ldb + kbeg;
        extA = ((long long)(min(p.M - m0, BM) - 1) * p.lda + Krem) * 2;
        extB = ((long long)(min(p.N - n0, BN) - 1) * p.ldb + Krem) * 2;
    } else {
        Ab = p.A + zoffA + (long long)kbeg * p.lda + m0;
        Bb = p.B + zoffB + (long long)kbeg * p.ldb + n0;
        extA = ((long long)(Krem - 1) * p.lda + ((min(p.M - m0, BM) + 7) & ~7)) * 2;
        extB = ((long long)(Krem - 1) * p.ldb + ((min(p.N - n0, BN) + 7) & ~7)) * 2;
    }
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)min(extA, 0x7fffffffLL), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)min(extB, 0x7fffffffLL), 0x00020000);

    unsigned offA[NIA], offB[NIB];
    int kcA[NIA], kcB[NIB];
    if (!TNMODE) {
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            offA[j] = (unsigned)(row * p.lda * 2 + c * 16);
            kcA[j] = c * 8;
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            offB[j] = (unsigned)(row * p.ldb * 2 + c * 16);
            kcB[j] = c * 8;
        }
    } else {
        const int st = lane >> 3, kin = (lane >> 1) & 3, half = lane & 1;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int q = j * NW + wave;
            const int kg = q / (BM / 128), part = q % (BM / 128);
            const int k = kg * 4 + kin, i = part * 128 + st * 16 + half * 8;
            offA[j] = (unsigned)(k * p.lda * 2 + i * 2);
            kcA[j] = k;
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const int q = j * NW + wave;
            const int kg = q / (BN / 128), part = q % (BN / 128);
            const int k = kg * 4 + kin, i = part * 128 + st * 16 + half * 8;
            offB[j] = (unsigned)(k * p.ldb * 2 + i * 2);
            kcB[j] = k;
        }
    }
    const unsigned kstepA = TNMODE ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
    const unsigned kstepB = TNMODE ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);

    auto stage_a = [&](int kt, unsigned char* base) {
        const int kleft = Krem - kt * BK;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const unsigned vo = (kcA[j] < kleft) ? offA[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, vo, kt * kstepA, 0, ALM_GEMM_AUX_A);
        }
    };
    auto stage_b = [&](int kt, unsigned char* base) {
        const int kleft = Krem - kt * BK;
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const unsigned vo = (kcB[j] < kleft) ? offB[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (j * NW + wave) * 1024), 16, vo, kt * kstepB, 0, ALM_GEMM_AUX_B);
        }
    };
    auto stage = [&](int kt, unsigned char* base) { stage_a(kt, base); stage_b(kt, base); };

    unsigned fragA[TM], fragB[TNB];
    if (!TNMODE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((wr * 128 + i * 32 + lr) * 128);
#pragma unroll
        for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)(A_BYTES + (wc * 128 + j * 32 + lr) * 128);
    } else {
        const int g = lane >> 4, s = lane & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((((g >> 1) * 2) * (BM / 16) + (wr * 128 + i * 32) / 16 + (g & 1)) * 128 + s * 8);
#pragma unroll
        for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)(A_BYTES + (((g >> 1) * 2) * (BN / 16) + (wc * 128 + j * 32) / 16 + (g & 1)) * 128 + s * 8);
    }
    const unsigned sw = (unsigned)((lr >> 1) & 7);

    f32x16 acc[TM][TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Frags { bf16x8 a[TM], b[TNB]; };
    auto load_frags = [&](const unsigned char* sb, int ks, Frags& f) {
        if (!TNMODE) {
            const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
#pragma unroll
            for (int j = 0; j < TNB; ++j) f.b[j] = *reinterpret_cast<const bf16x8*>(sb + fragB[j] + co);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* ad = sb + fragA[i] + ks * 4 * (BM / 16) * 128;
                f.a[i] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + (BM / 16) * 128), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < TNB; ++j) {
                const unsigned char* bd = sb + fragB[j] + ks * 4 * (BN / 16) * 128;
                f.b[j] = __builtin_shufflevector(lds_tr16(bd), lds_tr16(bd + (BN / 16) * 128), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto mfma_all = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j], f.a[i], acc[i][j], 0, 0, 0);
    };
    constexpr int NMF = TM * TNB;                                           // 16 MFMAs per sub-step
    constexpr int NRD = (TNMODE ? 2 : 1) * (TM + TNB);                      // LDS read instructions per fragment set (8 b128 | 16 b64)

    const int nk = (Krem + BK - 1) / BK;
    Frags F0, F1, F2, F3;

    stage(0, smem);
    if (nk > 1) { stage_a(1, smem + STAGE); wait_vmcnt<8>(); } else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_frags(smem, 0, F0);
    GPROBE_T(0);

    // one K-step; `cur` (read, then overwritten by the A half of tile kt + 2) and `nxt` (receives the B half of tile kt + 1, then read: sub-step 0 of
    // tile kt + 1) are distinct buffers.  MORE2 / MORE1 (tile kt + 2 / kt + 1 exists) are COMPILE-TIME: the steady-state body must be straight-line
    // code, or the schedule groups below (which only order instructions inside a basic block) cannot interleave DMA pieces and fragment reads with
    // the MFMAs.
    auto kstep = [&](unsigned char* __restrict__ cur, unsigned char* __restrict__ nxt, int kt, auto m2c, auto m1c) {
        constexpr bool MORE2 = decltype(m2c)::value, MORE1 = decltype(m1c)::value;
        // ---- S0: MFMA(F0) || read F1 <- cur || DMA of the B half of tile kt + 1 -> nxt (consumed one step ago: the barrier below ordered that)
        if (MORE1) stage_b(kt + 1, nxt);
        load_frags(cur, 1, F1);
        mfma_all(F0);
#pragma unroll
        for (int q = 0; q < NMF / 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (MORE1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // ONE DMA piece per two MFMAs (64 cycles): the CU's vector-memory path moves
            __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NMF / 2 - 1) / (NMF / 2), 0);   // 64 B / clk: 4 waves x 1 KiB = 64 cycles per round
        }
        __builtin_amdgcn_sched_barrier(0);
        GPROBE_T(1);
        // ---- S1, S2: MFMA(F1) || read F2 ; MFMA(F2) || read F3
        load_frags(cur, 2, F2);
        mfma_all(F1);
        load_frags(cur, 3, F3);
        mfma_all(F2);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int q = 0; q < NMF / 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NMF / 2 - 1) / (NMF / 2), 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        GPROBE_T(2);
        // ---- ONE barrier per K-step: this wave's reads of `cur` have returned and its pieces of tile kt + 1 have landed -- and everybody else's
        if (MORE1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        GPROBE_T(3);
        // ---- S3: MFMA(F3) || read F0 <- nxt (sub-step 0 of tile kt + 1) || DMA of the A half of tile kt + 2 -> cur
        if (MORE2) stage_a(kt + 2, cur);
        if (MORE1) load_frags(nxt, 0, F0);
        mfma_all(F3);
#pragma unroll
        for (int q = 0; q < NMF / 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (MORE2) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            if (MORE1) __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NMF / 2 - 1) / (NMF / 2), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        GPROBE_T(4);
    };
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
#pragma unroll 1
    for (; kt + 2 < nk; ++kt) kstep(smem + (kt & 1) * STAGE, smem + ((kt & 1) ^ 1) * STAGE, kt, T{}, T{});
    if (kt + 1 < nk) { kstep(smem + (kt & 1) * STAGE, smem + ((kt & 1) ^ 1) * STAGE, kt, F{}, T{}); ++kt; }
    kstep(smem + (kt & 1) * STAGE, smem + ((kt & 1) ^ 1) * STAGE, kt, F{}, F{});
    __syncthreads();                                                        // every wave is done with the stage buffers: the epilogue slabs reuse them

    constexpr int ES = OUT_F32 ? 4 : 2;
    static_assert(NW * 32 * (32 * TNB * ES) <= 2 * STAGE, "epilogue slab");
    const long long coff0 = z1 * p.sC1 + z2 * p.sC2 + (p.ksplit > 0 ? zs * p.sCk : 0);
    gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
    GPROBE_T(6);
    GPROBE_FLUSH((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + wave);
}

// ---- split-K second stage: C[b][m][n] (+)= sum_z ws[z][b][m][n]   (blockIdx.y = b)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long mn, long long slice_stride, int N,
                                                            float* __restrict__ C, long long ldc, long long sC, int accumulate,
                                                            int nb2 = 0x40000000, long long sC1 = 0) {   // defaults: one batch level (index y, stride sC)
    ws += (long long)blockIdx.y * mn;
    C += (long long)(blockIdx.y / nb2) * sC1 + (long long)(blockIdx.y % nb2) * sC;
    const bool vec = (N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && (mn & 3) == 0 && (slice_stride & 3) == 0;
    if (vec) {
        const long long mn4 = mn >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < mn4; i += (long long)gridDim.x * 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int z = 0; z < splits; ++z) {
                const float4 v = *reinterpret_cast<const float4*>(ws + (long long)z * slice_stride + i * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const long long e = i * 4, m = e / N, n = e % N;
            float4* c = reinterpret_cast<float4*>(C + m * ldc + n);
            if (accumulate) { const float4 w = *c; s.x += w.x; s.y += w.y; s.z += w.z; s.w += w.w; }
            *c = s;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < mn; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += ws[(long long)z * slice_stride + i];
        const long long m = i / N, n = i % N;
        float* c = C + m * ldc + n;
        *c = accumulate ? *c + s : s;
    }
}

// ---- 2-D transpose of a bf16 matrix: dst[c][r] = src[r][c]; dst has ld_dst >= rows (pad columns are zero-filled
// up to rows_pad so the transposed matrix can be used as a K-contiguous GEMM operand with K % 8 == 0).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols,
                                                             long long ld_src, long long ld_dst, int rows_pad, long long bs_src,
                                                             long long bs_dst) {
    __shared__ bf16_t tile[64][66];
    src += (long long)blockIdx.z * bs_src;
    dst += (long long)blockIdx.z * bs_dst;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * ld_src + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) dst[(long long)c * ld_dst + r] = tile[tx][i];
    }
}

// ---- fp32 master weight [rows][cols] -> bf16 packed copy dst[rows_pad][ld_dst] (zero padded) and, optionally, its
// transpose dstT[cols_pad][ld_dstT] (zero padded).  One launch per weight per optimiser step.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, int rows, int cols, long long ld_src,
                                                          bf16_t* __restrict__ dst, long long ld_dst, int rows_pad, int cols_pad,
                                                          bf16_t* __restrict__ dstT, long long ld_dstT) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const bf16_t v = (r < rows && c < cols) ? f2bf(src[(long long)r * ld_src + c]) : (bf16_t)0;
        tile[i][tx] = v;
        if (dst && r < rows_pad && c < cols_pad) dst[(long long)r * ld_dst + c] = v;
    }
    if (!dstT) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols_pad && r < rows_pad) dstT[(long long)c * ld_dstT + r] = tile[tx][i];
    }
}

// ---- several weights per launch (one transformer layer = 6 jobs): the per-step bf16 re-pack is launch-bound when done one weight at a time
struct PackJobs {
    AlmPackJob job[8];
    int tile_end[8];          // exclusive prefix sums of the per-job 64x64 tile counts
    int njobs;
};
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(PackJobs pj) {
    __shared__ bf16_t tile[64][66];
    int j = 0;
    while (j + 1 < pj.njobs && (int)blockIdx.x >= pj.tile_end[j]) ++j;
    const AlmPackJob& q = pj.job[j];
    const int local = blockIdx.x - (j ? pj.tile_end[j - 1] : 0);
    const int tcols = (q.cols_pad + 63) / 64;
    const int r0 = (local / tcols) * 64, c0 = (local % tcols) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    bf16_t* dst = reinterpret_cast<bf16_t*>(q.dst);
    bf16_t* dstT = reinterpret_cast<bf16_t*>(q.dstT);
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const bf16_t v = (r < q.rows && c < q.cols) ? f2bf(q.src[(long long)r * q.ld_src + c]) : (bf16_t)0;
        tile[i][tx] = v;
        if (dst && r < q.rows_pad && c < q.cols_pad) dst[(long long)r * q.ld_dst + c] = v;
    }
    if (!dstT) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < q.cols_pad && r < q.rows_pad) dstT[(long long)c * q.ld_dstT + r] = tile[tx][i];
    }
}

// the same with two elements per lane: 64 x 128 tiles, 8-byte fp32 loads, 4-byte bf16-pair stores both ways (needs even leading dimensions and
// 8 / 4-byte aligned bases; the re-pack of a layer's weights is a pure HBM stream: 4 B read + 2 x 2 B written per parameter)
struct PackJobs2 {
    AlmPackJob job[8];
    int tile_end[8];          // exclusive prefix sums of the per-job 64 x 128 tile counts
    int njobs;
};
__global__ __launch_bounds__(256) void pack_weights_multi2_kernel(PackJobs2 pj) {
    __shared__ uint32_t tile[64][65];                                   // [row][column pair], 65: the transposed reads spread over the banks
    int j = 0;
    while (j + 1 < pj.njobs && (int)blockIdx.x >= pj.tile_end[j]) ++j;
    const AlmPackJob& q = pj.job[j];
    const int local = blockIdx.x - (j ? pj.tile_end[j - 1] : 0);
    const int tcols = (q.cols_pad + 127) / 128;
    const int r0 = (local / tcols) * 64, c0 = (local % tcols) * 128;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    bf16_t* dst = reinterpret_cast<bf16_t*>(q.dst);
    bf16_t* dstT = reinterpret_cast<bf16_t*>(q.dstT);
    const int c = c0 + 2 * tx;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i;
        float2 v = make_float2(0.f, 0.f);
        if (r < q.rows) {
            if (c + 1 < q.cols) v = *reinterpret_cast<const float2*>(q.src + (long long)r * q.ld_src + c);
            else if (c < q.cols) v.x = q.src[(long long)r * q.ld_src + c];
        }
        const uint32_t pk = pack_bf2(v.x, v.y);
        tile[i][tx] = pk;
        if (dst && r < q.rows_pad && c < q.cols_pad) *reinterpret_cast<uint32_t*>(dst + (long long)r * q.ld_dst + c) = pk;
    }
    if (!dstT) return;
    __syncthreads();
    // transposed rows: output row = source column cc, 64 source rows = 32 pairs; a half wave writes one 128-byte output row segment
    const int rp = tx & 31, hw = tx >> 5;
    for (int it = ty; it < 64; it += 4) {
        const int cc = it * 2 + hw;                                     // source column within the tile
        const int oc = c0 + cc, orow = r0 + 2 * rp;
        const uint32_t a = tile[2 * rp][cc >> 1], b = tile[2 * rp + 1][cc >> 1];
        const uint32_t pk = (cc & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        if (oc < q.cols_pad && orow < q.rows_pad) *reinterpret_cast<uint32_t*>(dstT + (long long)oc * q.ld_dstT + orow) = pk;
    }
}

// Round 5: the same pack as a wide stream -- 64 x 128 tiles again, but a thread fetches 8 consecutive columns of 4 rows (8 unconditional 16-byte loads in
// flight: addresses clamped into the matrix, out-of-range values zeroed afterwards) and writes 16-byte vectors BOTH ways: 8 columns of a row to dst, and --
// through an LDS image of column pairs -- 8 rows of a column to dstT (lane = row group within a column pair: 128 contiguous bytes per half ... see below).
// The round-2 form above moved 8 / 4 / 4 bytes per lane behind per-element branches: 2.2-2.8 TB/s, 27-35 us per layer.  Up to PACK3_JOBS weights per
// launch (all layers of the stack: one launch instead of six).  Needs: ld_dst, ld_dstT, rows_pad, cols_pad multiples of 8, 16-byte aligned dst / dstT,
// 4-byte aligned src (any ld_src).
constexpr int PACK3_JOBS = 40;
struct PackJobs3 {
    AlmPackJob job[PACK3_JOBS];
    int tile_end[PACK3_JOBS];
    int njobs;
};
static_assert(sizeof(PackJobs3) <= 3800, "kernel-argument segment");
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte load that only assumes dword alignment (ld_src = 2730: rows start 8-byte aligned)

__global__ __launch_bounds__(256) void pack_weights_multi3_kernel(PackJobs3 pj) {
    __shared__ uint32_t tile[64][65];                                       // [row][column pair]
    int j = 0;
    while (j + 1 < pj.njobs && (int)blockIdx.x >= pj.tile_end[j]) ++j;
    const AlmPackJob q = pj.job[j];
    const int local = blockIdx.x - (j ? pj.tile_end[j - 1] : 0);
    const int tcols = (q.cols_pad + 127) / 128;
    const int r0 = (local / tcols) * 64, c0 = (local % tcols) * 128;
    const int t = threadIdx.x;
    const int cg = t & 15, ri = t >> 4;                                     // column group (8 columns), row within a 16-row slab
    const int c = c0 + cg * 8;
    bf16_t* dst = reinterpret_cast<bf16_t*>(q.dst);
    bf16_t* dstT = reinterpret_cast<bf16_t*>(q.dstT);
    f32x4u v[4][2];
    const int cmax = q.cols >= 4 ? q.cols - 4 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = min(r0 + ri + 16 * k, q.rows - 1);
        const float* sp = q.src + (long long)r * q.ld_src;
        v[k][0] = *reinterpret_cast<const f32x4u*>(sp + min(c, cmax));
        v[k][1] = *reinterpret_cast<const f32x4u*>(sp + min(c + 4, cmax));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = ri + 16 * k, r = r0 + i;
        float e[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cb = c + 4 * h, cl = min(cb, cmax);                   // the vector was loaded at column cl <= cb: element x of column cb + x sits at x + cb - cl
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int col = cb + x, idx = col - cl;
                float val = 0.f;
                if (r < q.rows && col < q.cols) val = idx == 0 ? v[k][h].x : idx == 1 ? v[k][h].y : idx == 2 ? v[k][h].z : idx == 3 ? v[k][h].w : 0.f;
                e[4 * h + x] = val;
            }
        }
        const uint4 pk = make_uint4(pack_bf2(e[0], e[1]), pack_bf2(e[2], e[3]), pack_bf2(e[4], e[5]), pack_bf2(e[6], e[7]));
        tile[i][cg * 4 + 0] = pk.x; tile[i][cg * 4 + 1] = pk.y; tile[i][cg * 4 + 2] = pk.z; tile[i][cg * 4 + 3] = pk.w;
        if (dst && r < q.rows_pad && c < q.cols_pad) *reinterpret_cast<uint4*>(dst + (long long)r * q.ld_dst + c) = pk;
    }
    if (!dstT) return;
    __syncthreads();
    // transposed image: a work item = (column pair cp, group of 8 source rows rg) -> two 16-byte stores (output rows 2 cp and 2 cp + 1, columns r0 + 8 rg ..).
    // lane = rg + 8 * (cp % 8): the 8 row groups of an output row are 128 contiguous bytes; LDS word (8 rg + jj) * 65 + cp -> bank 8 rg + cp + jj: no conflict
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int w = it * 256 + t;
        const int rg = w & 7, cp = w >> 3;                                  // 64 column pairs x 8 row groups
        uint32_t a[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a[jj] = tile[rg * 8 + jj][cp];
        const uint4 lo = make_uint4((a[0] & 0xffffu) | (a[1] << 16), (a[2] & 0xffffu) | (a[3] << 16), (a[4] & 0xffffu) | (a[5] << 16), (a[6] & 0xffffu) | (a[7] << 16));
        const uint4 hi = make_uint4((a[0] >> 16) | (a[1] & 0xffff0000u), (a[2] >> 16) | (a[3] & 0xffff0000u), (a[4] >> 16) | (a[5] & 0xffff0000u),
                                    (a[6] >> 16) | (a[7] & 0xffff0000u));
        const int oc = c0 + 2 * cp, orow = r0 + 8 * rg;
        if (orow < q.rows_pad) {
            if (oc < q.cols_pad) *reinterpret_cast<uint4*>(dstT + (long long)oc * q.ld_dstT + orow) = lo;
            if (oc + 1 < q.cols_pad) *reinterpret_cast<uint4*>(dstT + (long long)(oc + 1) * q.ld_dstT + orow) = hi;
        }
    }
}

// ---- launch plumbing ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32>
int launch_cfg(const GemmParams& p, int ny, int nz, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * BK * 2;
    static bool attr_done = false;                 // idempotent; a benign race sets the same value twice
    auto kfn = gemm_kernel<BM, BN, WM, WN, TNMODE, OUT_F32>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    if (p.raster >= 1) {
        const int tmaj = tiles_m >= tiles_n ? tiles_m : tiles_n, Q = tiles_m >= tiles_n ? tiles_n : tiles_m;
        const int PL = ((p.plimit > 0 ? p.plimit : tmaj * ny) + 7) / 8;
        hipLaunchKernelGGL(kfn, dim3(8 * PL * Q * nz), dim3(WM * WN * 64), smem, st, p);
        return 0;
    }
    hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n, ny, nz), dim3(WM * WN * 64), smem, st, p);
    return 0;
}

template <bool OUT_F32>
int launch_ring(const GemmParams& p, int ny, int nz, hipStream_t st) {          // tile 16: 128 x 128, 4 waves, 4-stage DMA ring (128 KB of LDS, NT, raster 0)
    constexpr int smem = 4 * (128 + 128) * BK * 2;
    static bool attr_done = false;
    auto kfn = gemm_ring_kernel<128, 128, 2, 2, false, OUT_F32, 4>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kfn, dim3(((p.M + 127) / 128) * ((p.N + 127) / 128), ny, nz), dim3(256), smem, st, p);
    return 0;
}

template <bool TNMODE, bool OUT_F32>
int launch_stag(const GemmParams& p, int ny, int nz, hipStream_t st) {
    constexpr int smem = 163840;
    static bool attr_done = false;
    auto kfn = gemm_stag_kernel<TNMODE, OUT_F32>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    if (p.raster >= 1) {
        const int tmaj = tiles_m >= tiles_n ? tiles_m : tiles_n, Q = tiles_m >= tiles_n ? tiles_n : tiles_m;
        const int PL = ((p.plimit > 0 ? p.plimit : tmaj * ny) + 7) / 8;
        hipLaunchKernelGGL(kfn, dim3(8 * PL * Q * nz), dim3(512), smem, st, p);
        return 0;
    }
    hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n, ny, nz), dim3(512), smem, st, p);
    return 0;
}


// in-launch split-K launch of the staggered NT tile: p.inl_ws / p.inl_cnt / p.inl_slices / p.ksplit set by the caller (nt_launch_ws); raster 0
template <bool OUT_F32>
int launch_stag_inl(const GemmParams& p, int ny, hipStream_t st) {
    constexpr int smem = 163840;
    static bool attr_done = false;
    auto kfn = gemm_stag_inl_kernel<OUT_F32>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n, ny, p.inl_slices), dim3(512), smem, st, p);
    return 0;
}

template <bool TNMODE, bool OUT_F32>
int launch_w4(const GemmParams& p, int ny, int nz, hipStream_t st) {
    constexpr int smem = 131072;
    static bool attr_done = false;
    auto kfn = gemm_w4_kernel<TNMODE, OUT_F32>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    if (p.raster >= 1) {
        const int tmaj = tiles_m >= tiles_n ? tiles_m : tiles_n, Q = tiles_m >= tiles_n ? tiles_n : tiles_m;
        const int PL = ((p.plimit > 0 ? p.plimit : tmaj * ny) + 7) / 8;
        hipLaunchKernelGGL(kfn, dim3(8 * PL * Q * nz), dim3(256), smem, st, p);
        return 0;
    }
    hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n, ny, nz), dim3(256), smem, st, p);
    return 0;
}

// Tile ids (alm_gemm_bf16_nt_tile): 1 = 128 x 128 (4 waves, two workgroups per CU), 2 = 256 x 256 lock-step (8 waves), 13 = 256 x 256 with
// staggered wave rows (the production big tile), 11 = 384 x 256 (8 waves, wave tile 192 x 64: 17 % fewer L2 -> LDS bytes per flop), 17 = 320 x 256 (round 6).
// Automatic choice (tile 0): small problems -> 1; otherwise 13, except that an NT problem takes 11 when the coarser tiling needs so many
// fewer rounds of 256 resident workgroups that it wins despite its 1.5x longer tile (measured per-tile cost ratio 1.41: W1 forward
// 16384 x 5472: 6 rounds -> 4, +6 %; with N = 1024 it is 172 tiles on 256 CUs, -15 %).
int pick_tile(int M, int N, int ny, int tile, bool tn) {
    if (tile == 1 || tile == 2 || tile == 11 || tile == 13 || tile == 14 || tile == 15 || tile == 16 || tile == 17) return tile;
    if (tile != 0) return -1;
    if (M < 256 || N < 256) return 1;
    const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * ny;
    if (t256 < 192) return 1;               // not enough 256^2 tiles to occupy most of the 256 CUs
    if (!tn) {
        // round model over the three big tiles: rounds of 256 resident workgroups x measured cost of one tile relative to 256 x 256 (384 x 256: 1.41).  Round 6:
        // 320 x 256 (tile 17, the generic 8-wave kernel with a 160 x 64 wave tile) for RAGGED token counts -- FineTransformer N = 2049, B = 8: M = 16 392 is 65 tile
        // rows of 256, so an N = 1024 output is 260 tiles = TWO rounds; 384 x 256 makes it 172 tiles (one round at 67 % of the CUs), 320 x 256 makes it 208
        // (one round at 81 %, shorter tiles).  ALM_GEMM_T320_COST (x 100, default 122; 0 = never) is its relative cost in the model.
        static const double c320 = [] { const char* e = getenv("ALM_GEMM_T320_COST"); return e ? atoi(e) / 100.0 : 1.22; }();
        const long long t384 = (long long)((M + 383) / 384) * ((N + 255) / 256) * ny;
        const long long t320 = (long long)((M + 319) / 320) * ((N + 255) / 256) * ny;
        const double c13 = (double)((t256 + 255) / 256), c11 = (double)((t384 + 255) / 256) * 1.41;
        const double c17 = c320 > 0. ? (double)((t320 + 255) / 256) * c320 : 1e30;
        if (c17 < c13 && c17 < c11) return 17;
        if (c11 < c13) return 11;
    }
    return 13;
}

// Rasterisation group of a raster-0 launch (GemmParams::group_m).  An XCD's L2 (4 MiB) serves the ~32 tiles its CUs run at a time: with groups of g
// tile rows those are g x (32 / g) tiles, and the group's A panel (g x BM rows x K) is what successive tile columns re-read -- it should FIT the L2
// beside the streaming B tiles.  ALM_GEMM_GROUP_M=<g> (A/B runs) overrides; negative = groups of tile columns.
int pick_group(const GemmParams& p, int tl) {
    static const int env_g = [] { const char* e = getenv("ALM_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
    if (env_g != 0) return env_g;
    (void)p; (void)tl;
    return 8;
}

template <bool TNMODE>
int launch_gemm(const GemmParams& p0, int ny, int nz, int out_f32, int tile, hipStream_t st, bool hook = true, bool only256 = false) {
    static const int big_tile = [] { const char* e = getenv("ALM_GEMM_BIG_TILE"); return e ? atoi(e) : 0; }();   // A/B hook: 2 / 11 / 13 for every big launch
    static const int nt_store = [] { const char* e = getenv("ALM_GEMM_NT_STORE"); return e ? atoi(e) : 0; }();    // A/B hook: non-temporal C stores
    GemmParams p = p0;
    int tl = pick_tile(p.M, p.N, ny * nz, tile, TNMODE);
    if (tl < 0) return ALM_ERR_UNSUPPORTED;
    p.group_m = pick_group(p, tl);
    p.nt_store = (nt_store == 1 || (nt_store == 2 && !out_f32)) ? 1 : 0;
    // only256: the caller's panel arithmetic (GemmParams.plimit and the tail offset of the hybrid weight-gradient plan) is in units of 256 x 256 tiles --
    // the hook may swap the 256 x 256 kernels for one another there, never re-tile the launch (a 384 x 256 grid would cover the wrong set of C tiles)
    if (hook && tl != 1 && (big_tile == 2 || big_tile == 13 || big_tile == 14 || (big_tile == 11 && !only256))) tl = big_tile;
    // The 4-wave tile (gemm_w4_kernel, id 14) is OPT-IN: stand-alone its main loop is 4-9 % faster per K-step than the staggered tile on long
    // contractions (K = 1024 -1 %, 2736 +-0, 5472 +4 %, 8192 +6-9 %), but INSIDE the training step it measured slower on alternating runs
    // (ALM_GEMM_W4_MINK = 4096: NT big-tile time 3.94 -> 4.05 ms / step; 2048: 4.13 ms; TN unchanged) -- one wave per SIMD has nothing to cover a
    // late operand fetch with when the inputs come from HBM instead of a warm L2.  ALM_GEMM_W4_MINK=<K> routes the
    // automatic 256 x 256 choices with at least that many K elements per slice to it (A/B runs).
    static const int w4_mink = [] { const char* e = getenv("ALM_GEMM_W4_MINK"); return e ? atoi(e) : 0; }();
    if (hook && w4_mink > 0 && tl == 13 && tile == 0 && (p.ksplit > 0 ? p.ksplit : p.K) >= w4_mink) tl = 14;
    // ... except the full-K weight-gradient launches of the hybrid plan (TN, K = all tokens of the batch, no slices): there the 4-wave tile is ON by
    // default -- its longer uninterrupted main loop is what it was built for: -0.05 ms / step on three interleaved runs (ALM_GEMM_W4_TN_MINK=0: off)
    static const int w4_tn_mink = [] { const char* e = getenv("ALM_GEMM_W4_TN_MINK"); return e ? atoi(e) : 8192; }();
    if (TNMODE && hook && w4_tn_mink > 0 && tl == 13 && (p.ksplit > 0 ? p.ksplit : p.K) >= w4_tn_mink) tl = 14;
    if (tl == 14) return out_f32 ? launch_w4<TNMODE, true>(p, ny, nz, st) : launch_w4<TNMODE, false>(p, ny, nz, st);
    if (tl == 13) return out_f32 ? launch_stag<TNMODE, true>(p, ny, nz, st) : launch_stag<TNMODE, false>(p, ny, nz, st);
    // tile 15 (round 5 experiment): 256 x 128, 8 waves as 4 x 2 (wave tile 64 x 64), one workgroup per CU -- for the N = 512 projections (Wq forward, dAO):
    // 64 x 4 = 256 tiles = ONE round of the chip where the 128 x 128 tile runs 512 workgroups at twice the L2 -> LDS bytes per flop.
    // ALM_GEMM_MID_TILE=1 routes the automatic small-tile choices with N >= 512 and >= 192 such tiles to it (NT only)
    static const int mid_tile = [] { const char* e = getenv("ALM_GEMM_MID_TILE"); return e ? atoi(e) : 0; }();
    if (!TNMODE && hook && mid_tile && tl == 1 && tile == 0 && p.ksplit == 0 && p.M >= 256 && p.N >= 512 &&
        (long long)((p.M + 255) / 256) * ((p.N + 127) / 128) * ny >= 192) tl = 15;
    // tile 16 (round 5): the 128 x 128 tile with a 4-stage DMA ring, for NT launches of so few tiles that every CU holds at most one workgroup anyway
    // (to_kv: 128) -- there the two-stage loop waits for one memory round trip per K-step.  ALM_GEMM_RING=0: off (A/B)
    static const int ring_on = [] { const char* e = getenv("ALM_GEMM_RING"); return e ? atoi(e) : 1; }();
    if (!TNMODE && hook && ring_on && tl == 1 && tile == 0 && p.ksplit == 0 && p.raster == 0 && p.K >= 256 &&
        (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * ny * nz <= 256) tl = 16;
    if (tl == 16) { if (TNMODE || p.raster != 0) return ALM_ERR_UNSUPPORTED; return out_f32 ? launch_ring<true>(p, ny, nz, st) : launch_ring<false>(p, ny, nz, st); }
    if (tl == 15) return out_f32 ? launch_cfg<256, 128, 4, 2, TNMODE, true>(p, ny, nz, st) : launch_cfg<256, 128, 4, 2, TNMODE, false>(p, ny, nz, st);
    if (tl == 11) return out_f32 ? launch_cfg<384, 256, 2, 4, TNMODE, true>(p, ny, nz, st) : launch_cfg<384, 256, 2, 4, TNMODE, false>(p, ny, nz, st);
    if (tl == 17) { if (TNMODE) return ALM_ERR_UNSUPPORTED; return out_f32 ? launch_cfg<320, 256, 2, 4, false, true>(p, ny, nz, st) : launch_cfg<320, 256, 2, 4, false, false>(p, ny, nz, st); }
    if (tl == 2) return out_f32 ? launch_cfg<256, 256, 2, 4, TNMODE, true>(p, ny, nz, st) : launch_cfg<256, 256, 2, 4, TNMODE, false>(p, ny, nz, st);
    return out_f32 ? launch_cfg<128, 128, 2, 2, TNMODE, true>(p, ny, nz, st) : launch_cfg<128, 128, 2, 2, TNMODE, false>(p, ny, nz, st);
}

bool view_too_big(long long rows, long long ld) { return rows * ld * 2 >= 0x7fffffffLL; }

// ---- NT launches that cannot fill the chip with 256 x 256 tiles (round 6): in-launch split-K on the staggered tile ----------------------------------------
// pick_tile() sends every NT launch with < 192 tiles of 256 x 256 to the 128 x 128 tile (twice the L2 -> LDS bytes per flop).  With a caller-owned workspace
// the alternative is the staggered 256 x 256 tile with each tile's K range cut over S workgroups (gemm_stag_inl_kernel): tiles x S <= 256 workgroups = ONE
// resident round of the chip.  MEASURED on MI355X (scripts/ab_nt_inl.py, every launch on a fresh operand set; profiles/r6a_nt_inl_cost.log, r6b_*):
//   * a half-empty chip is NOT half as fast: one K-step of a 256 x 256 workgroup takes 1.2 us with 32 workgroups in flight, 1.6 us with 128, 2.0 us with 256
//     (L2 / fabric contention), so cutting K in two over twice the workgroups saves ~35 %, not 50 %, of the main loop;
//   * the publish + reduce is a BURST: every workgroup ends at the same time and writes its 256-KiB slab through to memory -- 16 us with 128 workgroups (32 MB),
//     24 us with 256 (64 MB) -- and the 128 x 128 tile it competes with runs at 0.25-0.31 of the MFMA peak on these shapes (2 workgroups per CU, all CUs busy).
//   Net: the split wins only on LONG contractions -- M = 8192, N = 1024, K = 5472 (dXN = dU W1): 116 -> 107 us; K = 2736: 66 vs 68 us (tie); K <= 1024: the
//   128 x 128 tile wins by 1.5-2x (to_q at M = 16384: 24.5 vs 45.6 us).  A 256 x 128 tile (one round of 256 workgroups for N = 1024) measured 0-14 % SLOWER than
//   the 128 x 128 tile on every shape.  The model below reproduces those decisions; its constants are the fitted ones:
//     t(256^2, S >= 2) = ceil(ksteps / S) * (1.15 + 0.0033 W) + 7 + (8 + 0.04 W) + 2.6 * slabs read        [us; W = tiles x S workgroups]
//     t(128^2)         = rounds of 512 resident workgroups * (ksteps * 1.31 + 4.5)   |   <= 256 tiles (4-stage ring form): ksteps * 0.72 + 4.5
// Workspace: [1024 ticket counters, zero when handed over -- the kernel leaves them zero][tiles x S slabs of 256 KiB].
constexpr long long INL_CNT_BYTES = 4096, INL_SLAB_BYTES = 256 * 256 * 4;
struct NtPlan { int tile, slices; };

NtPlan nt_plan(int M, int N, int K, int nb) {
    static const int mode = [] { const char* e = getenv("ALM_GEMM_NT_INL"); return e ? atoi(e) : 1; }();     // 0: off (A/B), 1: cost model, >= 2: that many slices wherever they fit
    const int base = pick_tile(M, N, nb, 0, false);
    if (mode == 0 || base != 1 || M < 256 || N < 256) return NtPlan{base, 1};
    const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * nb;
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * nb;
    if (t256 > 128) return NtPlan{base, 1};                                     // (two slices must fit one resident round)
    const int ksteps = (K + BK - 1) / BK;
    double best_t = t128 <= 256 ? ksteps * 0.72 + 4.5 : (double)((t128 + 511) / 512) * (ksteps * 1.31 + 4.5);
    NtPlan best{base, 1};
    for (int S = 2; S <= 8; ++S) {
        const long long W = t256 * S;
        if (W > 256) break;
        const int kc = (ksteps + S - 1) / S;
        if (kc < 4) break;
        if ((ksteps + kc - 1) / kc != S) continue;                              // (no empty slices)
        const double t = kc * (1.15 + 0.0033 * W) + 7.0 + (8.0 + 0.04 * W) + 2.6 * (S == 2 ? 1 : S);
        if (mode >= 2 ? (S == mode || best.slices == 1) : t < best_t) { best_t = t; best = NtPlan{13, S}; }
    }
    return best;
}

// launch of one (possibly batched) NT problem on the plan above; p: as for launch_gemm<false>, raster 0, no split.  ws == nullptr or too small: the
// workspace-free choice of alm_gemm_bf16_nt
int nt_launch_ws(const GemmParams& p0, int ny, int out_f32, int force_slices, void* ws, long long ws_bytes, hipStream_t st) {
    NtPlan pl = nt_plan(p0.M, p0.N, p0.K, ny);
    if (force_slices > 0) {
        const long long t256 = (long long)((p0.M + 255) / 256) * ((p0.N + 255) / 256) * ny;
        if (p0.M < 256 || p0.N < 256 || t256 > 1024) return ALM_ERR_UNSUPPORTED;
        pl = NtPlan{13, force_slices};
    }
    if (force_slices == 0 && (pl.tile != 13 || pl.slices < 2)) return launch_gemm<false>(p0, ny, 1, out_f32, 0, st);      // the workspace-free choice
    const long long tiles = (long long)((p0.M + 255) / 256) * ((p0.N + 255) / 256) * ny;
    if (pl.slices == 1) return launch_gemm<false>(p0, ny, 1, out_f32, 13, st, false);      // (forced: the plain staggered tile)
    const long long need = INL_CNT_BYTES + tiles * pl.slices * INL_SLAB_BYTES;
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 15)) {
        if (force_slices > 0) return ALM_ERR_BAD_ARG;
        return launch_gemm<false>(p0, ny, 1, out_f32, 0, st);
    }
    GemmParams p = p0;
    const int ksteps = (p.K + BK - 1) / BK;
    p.ksplit = (ksteps + pl.slices - 1) / pl.slices * BK;
    p.sCk = 0;
    p.inl_slices = (p.K + p.ksplit - 1) / p.ksplit;
    if (p.inl_slices < 2) return launch_gemm<false>(p0, ny, 1, out_f32, 13, st, false);
    p.inl_cnt = reinterpret_cast<unsigned*>(ws);
    p.inl_ws = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + INL_CNT_BYTES);
    p.group_m = pick_group(p, 13);
    return out_f32 ? launch_stag_inl<true>(p, ny, st) : launch_stag_inl<false>(p, ny, st);
}

// Split-K plan for `nb` same-shape problems: choose (tile, slices) minimising a simple time model --
//   block waves over the chip x K-steps per block x measured time per K-step  +  workspace round trip through HBM.
// (A balanced "stream-K"-like split -- one equally long K-step range per CU -- measured SLOWER: the workgroups of an XCD then walk different
// K ranges and no longer share operand panels through the L2; it lives in csrc/lab/gemm_lab.hip)
struct SplitPlan { int tile, slices; };

// XCD-panel rasterisation pays only when the panels spread evenly over the 8 XCDs (measured: 22 panels +3 %, 11 panels -40 %)
int pick_raster(int M, int N, int nb, int tile) {
    const int bm = tile >= 2 ? 256 : 128;
    const int tmm = (M + bm - 1) / bm, tnn = (N + bm - 1) / bm;
    const int P = (tmm >= tnn ? tmm : tnn) * nb;
    return (P % 8 == 0 || P >= 20) ? 1 : 0;
}
SplitPlan splitk_plan(int M, int N, int K, int nb) {
    SplitPlan best{1, 1};
    double best_t = 1e30;
    for (int tile = 1; tile <= 2; ++tile) {
        if (tile == 2 && (M < 256 || N < 256)) continue;
        const int bm = tile == 2 ? 256 : 128;
        const double tiles = (double)((M + bm - 1) / bm) * ((N + bm - 1) / bm) * nb;
        const double slots = tile == 2 ? 256.0 : 512.0;               // resident blocks on the chip
        const double us_per_kstep = tile == 2 ? 2.0 : 1.25;          // one 64-deep K-step of one block (measured, whole chip busy)
        for (int s = 1; s <= 64; ++s) {
            if (s > 1 && (K + s - 1) / s < 256) break;
            const int kc = ((K + s - 1) / s + BK - 1) / BK * BK;
            const int nsl = (K + kc - 1) / kc;
            if (nsl != s) continue;
            const double waves = ceil(tiles * s / slots);
            double t = waves * ((kc / BK) * us_per_kstep + 3.0);
            if (s > 1) t += (double)s * M * N * nb * 8.0 / 4.0e6 + 3.0;   // fp32 partials: written + read back at ~4 TB/s, + the reduce launch
            if (t < best_t) { best_t = t; best = SplitPlan{tile == 2 ? 13 : 1, s}; }
        }
    }
    return best;
}

// Hybrid plan of the layer-batched weight gradients (alm_gemm_bf16_tn_batched): when the 256 x 256 tiles of all problems fill the chip's 256 CUs a
// whole number of times plus a SMALL remainder (dW1 of 3 layers: 264 tiles), the uniform split-K plan pays for the remainder with partials of
// EVERYTHING (4 slices: 5 waves of K/4 + 270 MB of fp32 partials).  Instead: the first panels_a panels (whole waves) run at full K straight into C,
// and the remaining row (column) blocks -- they all lie in the LAST problem -- are one ordinary split-K launch, sliced deep enough to occupy the chip
// once.  panels_a == 0: not applicable (use the uniform plan).
struct HybridPlan { int panels_a, m_major, off, slices_b; };
HybridPlan hybrid_plan(int M, int N, int K, int nb) {
    static const int off_switch = [] { const char* e = getenv("ALM_GEMM_HYBRID"); return e ? atoi(e) : 1; }();    // A/B switch (0: uniform plan)
    HybridPlan none{0, 0, 0, 0};
    if (!off_switch || M < 256 || N < 256 || K < 4096) return none;
    const int tm = (M + 255) / 256, tn = (N + 255) / 256;
    const int m_major = tm >= tn, tmaj = m_major ? tm : tn, Q = m_major ? tn : tm;
    const int P = tmaj * nb, total = P * Q;
    if (total <= 256) return none;
    const int panels_a = (total / 256) * 256 / Q;                 // whole waves of 256 tiles (Q tiles per panel)
    const int rem = P - panels_a;
    if (panels_a * Q % 256 != 0 || rem <= 0 || rem >= tmaj || rem * Q > 64) return none;    // the tail must be small and inside the last problem
    const int off = (tmaj - rem) * 256;
    int s = 256 / (rem * Q);                                      // tail tiles x slices ~ one wave of the chip
    const int ksteps = (K + BK - 1) / BK;
    while (s > 1 && ksteps / s < 8) --s;                          // at least 8 K-steps per slice
    if (s < 2) return none;
    return HybridPlan{panels_a, m_major, off, s};
}

// one TN split-K problem with a GIVEN number of slices (the tail of the hybrid plan): partials into ws [slices][M][N], then the fixed-order reduce
int splitk_fixed(const bf16_t* At, const bf16_t* Bt, float* C, float* ws, int M, int N, int K, long long lda, long long ldb, long long ldc, int slices,
                 float alpha, int accumulate, hipStream_t st) {
    int kc = (K + slices - 1) / slices;
    kc = (kc + BK - 1) / BK * BK;
    const int nsl = (K + kc - 1) / kc;
    const long long mn = (long long)M * N;
    GemmParams p{At, Bt, ws, nullptr, M, N, K, lda, ldb, (long long)N, 1, 0, 0, 0, 0, 0, mn, alpha, 0, kc, mn, 0, nsl, 1};
    int rc = launch_gemm<true>(p, 1, nsl, 1, 13, st);
    if (rc) return rc;
    const int grid = (int)((mn + 255) / 256 < 2048 ? (mn + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid, 1), dim3(256), 0, st, (const float*)ws, nsl, mn, mn, N, C, ldc, 0LL, accumulate);
    return 0;
}

}  // namespace

extern "C" int alm_gemm_bf16_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                long long ldb, long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                                long long sB2, long long sC1, long long sC2, float alpha, int out_f32, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (((sA1 | sA2 | sB1 | sB2) & 7) != 0) return ALM_ERR_BAD_ARG;
    if (view_too_big(384, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate, 0, 0, 0, 1};
    int rc = launch_gemm<false>(p, nb1 * nb2, 1, out_f32, 0, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// alm_gemm_bf16_nt with a caller-owned workspace (round 6): launches that leave most of the chip idle on 256 x 256 tiles (< 192 of them: every D- / 512-wide
// projection at M = 8192, to_q / dAO at M = 16384) run on the staggered tile with IN-LAUNCH split-K when the measured cost model says so (nt_plan): each K
// slice publishes its fp32 accumulators to `ws`, the tile's last arriver reduces them in fixed slice order and runs the usual epilogue -- bitwise
// deterministic, no second launch, no atomics on data.  ws: >= alm_gemm_nt_ws_bytes() bytes, 16-byte aligned, its first 4096 bytes ZERO when first handed
// over (ticket counters; every launch leaves them zero), used by ONE stream at a time.  ws == NULL (or too small): exactly alm_gemm_bf16_nt.
extern "C" int alm_gemm_bf16_nt_ws(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb, long long ldc,
                                   int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2, float alpha,
                                   int out_f32, int accumulate, void* ws, long long ws_bytes, void* stream) {
    if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (((sA1 | sA2 | sB1 | sB2) & 7) != 0) return ALM_ERR_BAD_ARG;
    if (view_too_big(384, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate, 0, 0, 0, 1};
    int rc = nt_launch_ws(p, nb1 * nb2, out_f32, 0, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// the same with a GIVEN number of K slices on the staggered 256 x 256 tile, un-batched (tests / the cost-model benchmark scripts/ab_nt_inl.py); slices = 1: the
// plain staggered tile.  M, N >= 256.
extern "C" int alm_gemm_bf16_nt_inl(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb, long long ldc,
                                    float alpha, int out_f32, int accumulate, int slices, void* ws, long long ws_bytes, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || slices < 1 || slices > 16) return ALM_ERR_BAD_ARG;
    if (view_too_big(384, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, 1, 0, 0, 0, 0, 0, 0, alpha, accumulate, 0, 0, 0, 1};
    int rc = nt_launch_ws(p, 1, out_f32, slices, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// Host-side plan query of alm_gemm_bf16_nt_ws (no launch, no GPU): plan[0] = block tile (1 = 128 x 128, 16 = its 4-stage DMA-ring form, 13 = 256 x 256 staggered,
// 11 = 384 x 256), plan[1] = K slices of the in-launch split-K form (1: none), plan[2] = workspace bytes that plan needs (0: none), plan[3] = workgroups
extern "C" int alm_gemm_nt_plan(int M, int N, int K, int nb, int with_ws, int* plan) {
    nb = nb < 1 ? 1 : nb;
    NtPlan pl = with_ws ? nt_plan(M, N, K, nb) : NtPlan{pick_tile(M, N, nb, 0, false), 1};
    int tile = pl.tile;
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * nb;
    static const int ring_on = [] { const char* e = getenv("ALM_GEMM_RING"); return e ? atoi(e) : 1; }();
    if (tile == 1 && ring_on && K >= 256 && t128 <= 256) tile = 16;
    const long long bm = tile == 11 ? 384 : (tile == 17 ? 320 : (tile == 13 ? 256 : 128)), bn = tile == 1 || tile == 16 ? 128 : 256;
    const long long wgs = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * nb * pl.slices;
    if (plan) {
        plan[0] = tile; plan[1] = pl.slices;
        plan[2] = pl.slices > 1 ? (int)(INL_CNT_BYTES + (wgs * INL_SLAB_BYTES)) : 0;
        plan[3] = (int)wgs;
    }
    return 0;
}

// workspace bytes that cover EVERY plan alm_gemm_bf16_nt_ws can choose (one resident round of 256 workgroups + the ticket counters)
extern "C" int alm_gemm_nt_ws_bytes(void) { return (int)(INL_CNT_BYTES + 256 * INL_SLAB_BYTES); }

// Two un-batched NT problems in ONE launch (gemm_group2_kernel / gemm_stag_group2_kernel): C0 = A0 . B0^T and C1 = A1 . B1^T, bf16 outputs (or fp32:
// out_f32), no bias, alpha 1, no accumulation.  Both problems must pick the same block tile (128 x 128 or the staggered 256 x 256) and problem 0's
// tile count must be a multiple of 8 (problem 1's workgroups keep the XCD <-> tile-order correspondence the rasterisation assumes); otherwise -- and
// with ALM_GEMM_GROUP2=0 -- the two problems are launched one after the other: same results either way (each tile is computed by the same code).
extern "C" int alm_gemm_bf16_nt_group2(const void* A0, const void* B0, void* C0, int M0, int N0, int K0, long long lda0, long long ldb0, long long ldc0,
                                       const void* A1, const void* B1, void* C1, int M1, int N1, int K1, long long lda1, long long ldb1, long long ldc1,
                                       int out_f32, void* stream) {
    static const int on = [] { const char* e = getenv("ALM_GEMM_GROUP2"); return e ? atoi(e) : 1; }();
    hipStream_t st = (hipStream_t)stream;
    auto bad = [](const void* A, const void* B, int K, long long lda, long long ldb) {
        return K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15);
    };
    if (M0 <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0) {
        int rc = alm_gemm_bf16_nt(A0, B0, C0, nullptr, M0, N0, K0, lda0, ldb0, ldc0, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, out_f32, 0, stream);
        if (rc) return rc;
        return alm_gemm_bf16_nt(A1, B1, C1, nullptr, M1, N1, K1, lda1, ldb1, ldc1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, out_f32, 0, stream);
    }
    if (bad(A0, B0, K0, lda0, ldb0) || bad(A1, B1, K1, lda1, ldb1)) return ALM_ERR_BAD_ARG;
    if (view_too_big(384, lda0) || view_too_big(256, ldb0) || view_too_big(384, lda1) || view_too_big(256, ldb1)) return ALM_ERR_UNSUPPORTED;
    int t0 = pick_tile(M0, N0, 1, 0, false), t1 = pick_tile(M1, N1, 1, 0, false);
    // round 6, measured and NOT adopted (switch kept for A/B runs, default off): two problems that are each too small for the big tile (< 192 tiles of
    // 256 x 256) but TOGETHER fill the chip in one round on the staggered tile (dXN_q || dX_kv at M = 8192: 128 + 128 tiles) -- 23.1 us against 22.3 us on
    // the 128 x 128 tile (profiles/r6b_group2.log): K = 512 / 128 is 8 / 2 K-steps, the staggered tile's fill + drain dominates
    static const int grp_big = [] { const char* e = getenv("ALM_GEMM_GROUP2_BIG"); return e ? atoi(e) : 0; }();
    if (grp_big && t0 == 1 && t1 == 1 && M0 >= 256 && N0 >= 256 && M1 >= 256 && N1 >= 256) {
        const long long a = (long long)((M0 + 255) / 256) * ((N0 + 255) / 256), b = (long long)((M1 + 255) / 256) * ((N1 + 255) / 256);
        if (a + b >= 192 && a + b <= 256 && (a & 7) == 0) t0 = t1 = 13;
    }
    const int bm = t0 == 1 ? 128 : 256;
    const int tiles0 = ((M0 + bm - 1) / bm) * ((N0 + bm - 1) / bm), tiles1 = ((M1 + bm - 1) / bm) * ((N1 + bm - 1) / bm);
    if (!on || t0 != t1 || (t0 != 1 && t0 != 13) || (tiles0 & 7)) {
        int rc = alm_gemm_bf16_nt(A0, B0, C0, nullptr, M0, N0, K0, lda0, ldb0, ldc0, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, out_f32, 0, stream);
        if (rc) return rc;
        return alm_gemm_bf16_nt(A1, B1, C1, nullptr, M1, N1, K1, lda1, ldb1, ldc1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, out_f32, 0, stream);
    }
    GemmParams p0{(const bf16_t*)A0, (const bf16_t*)B0, C0, nullptr, M0, N0, K0, lda0, ldb0, ldc0, 1, 0, 0, 0, 0, 0, 0, 1.f, 0, 0, 0, 0, 1};
    GemmParams p1{(const bf16_t*)A1, (const bf16_t*)B1, C1, nullptr, M1, N1, K1, lda1, ldb1, ldc1, 1, 0, 0, 0, 0, 0, 0, 1.f, 0, 0, 0, 0, 1};
    p0.group_m = pick_group(p0, t0);
    p1.group_m = pick_group(p1, t1);
    // ONE dynamic-LDS attribute flag PER KERNEL (`which`): the four kernels decay to one function-pointer type, so a generic lambda with a local static would
    // share a single flag between them and only the first variant launched in a process would get its attribute (round-5 advisor finding)
    auto launch = [&](auto kfn, int which, int threads, int smem) -> int {
        static bool attr_done[4] = {false, false, false, false};
        if (!attr_done[which]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
            attr_done[which] = true;
        }
        hipLaunchKernelGGL(kfn, dim3(tiles0 + tiles1), dim3(threads), smem, st, p0, p1, tiles0);
        return 0;
    };
    int rc;
    if (t0 == 1) rc = out_f32 ? launch(gemm_group2_kernel<128, 128, 2, 2, false, true>, 0, 256, 2 * (128 + 128) * BK * 2)
                              : launch(gemm_group2_kernel<128, 128, 2, 2, false, false>, 1, 256, 2 * (128 + 128) * BK * 2);
    else rc = out_f32 ? launch(gemm_stag_group2_kernel<false, true>, 2, 512, 163840) : launch(gemm_stag_group2_kernel<false, false>, 3, 512, 163840);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

/* tile-selectable form of the above without batching (tests / benchmarks): see pick_tile for the ids */
extern "C" int alm_gemm_bf16_nt_tile(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                     long long ldb, long long ldc, float alpha, int out_f32, int accumulate, int tile, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (view_too_big(384, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, 1, 0, 0, 0, 0, 0, 0, alpha, accumulate, 0, 0, 0, 1};
    int rc = launch_gemm<false>(p, 1, 1, out_f32, tile, (hipStream_t)stream, false);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// Split-K for long-K / few-tile contractions (weight gradients: K = B*N tokens), `nb` same-shape problems per launch (element
// strides sA / sB / sC between them).  ws: fp32 workspace of alm_gemm_splitk_ws_floats(M, N, K, nb) floats (unused when that is 0).
// Deterministic (no atomics): the slices are reduced in a fixed order by a second kernel.
extern "C" int alm_gemm_splitk_slices(int M, int N, int K, int nb) { return splitk_plan(M, N, K, nb < 1 ? 1 : nb).slices; }

// block tile the split-K plan picks for this problem: 1 = 128 x 128, otherwise a 256 x 256 tile (bench.py files its per-kernel timings by it)
extern "C" int alm_gemm_splitk_tile(int M, int N, int K, int nb) { return splitk_plan(M, N, K, nb < 1 ? 1 : nb).tile; }

// Host-side plan queries (no launch, no GPU): which kernel a given launch WILL take.  Tests assert with them that the shapes of the benchmarked step
// (B = 8: M = 16384) run on the big tiles the roofline is quoted on, bench.py files its per-kernel timings by them.
//   alm_gemm_nt_tile_choice: block tile of alm_gemm_bf16_nt(M, N, nb = nb1 * nb2 problems): 1 = 128 x 128 (launched in its 4-stage DMA-ring form, tile 16,
//   when the whole launch is <= 256 such tiles and K >= 256), 13 = 256 x 256 staggered, 11 = 384 x 256
//   (the ALM_GEMM_BIG_TILE / ALM_GEMM_W4_MINK A/B environment hooks are NOT applied: this is the shipped choice)
extern "C" int alm_gemm_nt_tile_choice(int M, int N, int nb) { return pick_tile(M, N, nb < 1 ? 1 : nb, 0, false); }

//   alm_gemm_tn_batched_plan: alm_gemm_bf16_tn_batched(M, N, K, nb problems) -> 0 = one launch at full K (no partials), 1 = uniform split-K
//   (plan[0] = tile, plan[1] = slices), 2 = hybrid (plan[0] = panels at full K on the 4-wave 256 x 256 tile when K >= 8192, else the staggered
//   tile; plan[1] = slices of the tail; plan[2] = first row / column of the tail inside the last problem; plan[3] = 1 when the tail is cut along M)
extern "C" int alm_gemm_tn_batched_plan(int M, int N, int K, int nb, int* plan) {
    nb = nb < 1 ? 1 : nb;
    const HybridPlan hy = hybrid_plan(M, N, K, nb);
    if (hy.panels_a > 0) {
        if (plan) { plan[0] = hy.panels_a; plan[1] = hy.slices_b; plan[2] = hy.off; plan[3] = hy.m_major; }
        return 2;
    }
    const SplitPlan pl = splitk_plan(M, N, K, nb);
    if (plan) { plan[0] = pl.tile; plan[1] = pl.slices; plan[2] = 0; plan[3] = 0; }
    return pl.slices > 1 ? 1 : 0;
}

// fp32 workspace floats alm_gemm_bf16_{nt,tn}_splitk need for this problem (0: none)
extern "C" int alm_gemm_splitk_ws_floats(int M, int N, int K, int nb) {
    nb = nb < 1 ? 1 : nb;
    const SplitPlan pl = splitk_plan(M, N, K, nb);
    long long fl = pl.slices > 1 ? (long long)pl.slices * nb * M * N : 0;
    const HybridPlan hy = hybrid_plan(M, N, K, nb);               // alm_gemm_bf16_tn_batched: the tail's partials (the larger of the two covers both callers)
    if (hy.panels_a > 0) {
        const long long m2 = hy.m_major ? M - hy.off : M, n2 = hy.m_major ? N : N - hy.off;
        const long long f2 = (long long)hy.slices_b * m2 * n2;
        if (f2 > fl) fl = f2;
    }
    return fl > 0x7fffffffLL ? -1 : (int)fl;
}

static int splitk_common(bool tn, const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                         long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate, hipStream_t st) {
    const SplitPlan pl = splitk_plan(M, N, K, nb);
    if (pl.slices <= 1) {
        GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, M, N, K, lda, ldb, ldc, nb, 0, sA, 0, sB, 0, sC, alpha, accumulate, 0, 0, pick_raster(M, N, nb, pl.tile), 1};
        return tn ? launch_gemm<true>(p, nb, 1, 1, pl.tile, st) : launch_gemm<false>(p, nb, 1, 1, pl.tile, st);
    }
    if (!ws) return ALM_ERR_BAD_ARG;
    int kc = (K + pl.slices - 1) / pl.slices;
    kc = (kc + BK - 1) / BK * BK;
    const int nsl = (K + kc - 1) / kc;
    const long long mn = (long long)M * N;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, ws, nullptr, M, N, K, lda, ldb, (long long)N, nb, 0, sA, 0, sB, 0, mn, alpha, 0, kc, mn * nb, pick_raster(M, N, nb, pl.tile), nsl};
    int rc = tn ? launch_gemm<true>(p, nb, nsl, 1, pl.tile, st) : launch_gemm<false>(p, nb, nsl, 1, pl.tile, st);
    if (rc) return rc;
    const int grid = (int)((mn + 255) / 256 < 2048 ? (mn + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid, nb), dim3(256), 0, st, (const float*)ws, nsl, mn, mn * nb, N, C, ldc, sC, accumulate);
    return 0;
}


extern "C" int alm_gemm_bf16_nt_splitk(const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                                       long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate,
                                       void* stream) {
    if (M <= 0 || N <= 0 || nb <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((sA | sB) & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (view_too_big(256, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    int rc = splitk_common(false, A, B, C, ws, M, N, K, lda, ldb, ldc, nb, sA, sB, sC, alpha, accumulate, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// Weight-gradient form: C[M,N] fp32 (+)= alpha * sum_k At[k][m] * Bt[k][n]  (At: [K][lda], Bt: [K][ldb], row-major activations).
extern "C" int alm_gemm_bf16_tn_splitk(const void* At, const void* Bt, float* C, float* ws, int M, int N, int K, long long lda,
                                       long long ldb, long long ldc, int nb, long long sA, long long sB, long long sC, float alpha,
                                       int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || nb <= 0) return 0;
    if (K <= 0 || (lda & 7) || (ldb & 7) || ((sA | sB) & 7) || ((uintptr_t)At & 15) || ((uintptr_t)Bt & 15)) return ALM_ERR_BAD_ARG;
    if (view_too_big(K, lda) || view_too_big(K, ldb)) return ALM_ERR_UNSUPPORTED;
    int rc = splitk_common(true, At, Bt, C, ws, M, N, K, lda, ldb, ldc, nb, sA, sB, sC, alpha, accumulate, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// Two-level batched weight gradients (round 3): C[z1][z2][M][N] fp32 (+)= alpha * At[z1][z2][K][M]^T . Bt[z1][z2][K][N] for nb1 x nb2 same-shape problems
// with element strides (sA1, sA2) / (sB1, sB2) / (sC1, sC2) -- the form in which ALL LAYERS' gradients of one weight kind are computed in one launch at
// the end of the backward pass from stacked activation buffers (core.stack_backward, deferred mode): enough tiles to fill the chip with long K slices,
// i.e. few or no split-K partials and one reduce launch per kind instead of one per layer.  ws: alm_gemm_splitk_ws_floats(M, N, K, nb1 * nb2) floats.
extern "C" int alm_gemm_bf16_tn_batched(const void* At, const void* Bt, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                                        long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1,
                                        long long sC2, float alpha, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return 0;
    if (K <= 0 || (lda & 7) || (ldb & 7) || ((sA1 | sA2 | sB1 | sB2) & 7) || ((uintptr_t)At & 15) || ((uintptr_t)Bt & 15)) return ALM_ERR_BAD_ARG;
    if (view_too_big(K, lda) || view_too_big(K, ldb)) return ALM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nb = nb1 * nb2;
    const SplitPlan pl = splitk_plan(M, N, K, nb);
    int rc;
    const HybridPlan hy = hybrid_plan(M, N, K, nb);
    if (hy.panels_a > 0) {
        // (a) the panels that fill whole waves of the chip: full K, straight into C -- no partials, no reduce
        static const int raster_a = [] { const char* e = getenv("ALM_GEMM_HYBRID_RASTER"); return e ? atoi(e) : 2; }();    // A/B switch: 1 = round-robin panels
        GemmParams pa{(const bf16_t*)At, (const bf16_t*)Bt, C, nullptr, M, N, K, lda, ldb, ldc, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate, 0, 0,
                      (hy.panels_a % 8 == 0 && raster_a == 2) ? 2 : 1, 1, nb, hy.panels_a};
        rc = launch_gemm<true>(pa, nb, 1, 1, 13, st, true, true);
        if (rc) return rc;
        // (b) the last problem's remaining row (column) blocks: an ordinary split-K problem on the sub-matrix, deep enough to fill the chip once
        const long long zA = (long long)(nb1 - 1) * sA1 + (long long)(nb2 - 1) * sA2, zB = (long long)(nb1 - 1) * sB1 + (long long)(nb2 - 1) * sB2;
        const long long zC = (long long)(nb1 - 1) * sC1 + (long long)(nb2 - 1) * sC2;
        const bf16_t* A2 = (const bf16_t*)At + zA + (hy.m_major ? hy.off : 0);
        const bf16_t* B2 = (const bf16_t*)Bt + zB + (hy.m_major ? 0 : hy.off);
        float* C2 = C + zC + (hy.m_major ? (long long)hy.off * ldc : hy.off);
        const int M2 = hy.m_major ? M - hy.off : M, N2 = hy.m_major ? N : N - hy.off;
        if (!ws) return ALM_ERR_BAD_ARG;
        rc = splitk_fixed(A2, B2, C2, ws, M2, N2, K, lda, ldb, ldc, hy.slices_b, alpha, accumulate, st);
        if (rc) return rc;
        ALM_LAUNCH_CHECK();
        return 0;
    }
    if (pl.slices <= 1) {
        GemmParams p{(const bf16_t*)At, (const bf16_t*)Bt, C, nullptr, M, N, K, lda, ldb, ldc, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate, 0, 0,
                     pick_raster(M, N, nb, pl.tile), 1, nb};
        rc = launch_gemm<true>(p, nb, 1, 1, pl.tile, st);
    } else {
        if (!ws) return ALM_ERR_BAD_ARG;
        int kc = (K + pl.slices - 1) / pl.slices;
        kc = (kc + BK - 1) / BK * BK;
        const int nsl = (K + kc - 1) / kc;
        const long long mn = (long long)M * N;
        GemmParams p{(const bf16_t*)At, (const bf16_t*)Bt, ws, nullptr, M, N, K, lda, ldb, (long long)N, nb2, sA1, sA2, sB1, sB2, mn * nb2, mn, alpha, 0, kc, mn * nb,
                     pick_raster(M, N, nb, pl.tile), nsl, nb};
        rc = launch_gemm<true>(p, nb, nsl, 1, pl.tile, st);
        if (rc) return rc;
        const int grid = (int)((mn + 255) / 256 < 2048 ? (mn + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid, nb), dim3(256), 0, st, (const float*)ws, nsl, mn, mn * nb, N, C, ldc, sC2, accumulate, nb2, sC1);
    }
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_transpose_bf16(const void* src, void* dst, int rows, int cols, long long ld_src, long long ld_dst, int rows_pad,
                                  void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    if (rows_pad < rows || ld_dst < rows_pad) return ALM_ERR_BAD_ARG;
    dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_src,
                       ld_dst, rows_pad, 0LL, 0LL);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_transpose_bf16_batched(const void* src, void* dst, int rows, int cols, long long ld_src, long long ld_dst, int rows_pad, int nb,
                                          long long bs_src, long long bs_dst, void* stream) {
    if (rows <= 0 || cols <= 0 || nb <= 0) return 0;
    if (rows_pad < rows || ld_dst < rows_pad || nb > 65535) return ALM_ERR_BAD_ARG;
    dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64, nb);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_src,
                       ld_dst, rows_pad, bs_src, bs_dst);
    ALM_LAUNCH_CHECK();
    return 0;
}

static int pack3_launch(const AlmPackJob* jobs, int njobs, hipStream_t st) {
    PackJobs3 pj{};                                  // 3 KB, by value: inside the 4 KB kernel-argument segment; indexed with a workgroup-uniform job number
    int total = 0;
    for (int j = 0; j < njobs; ++j) {
        pj.job[j] = jobs[j];
        total += ((jobs[j].cols_pad + 127) / 128) * ((jobs[j].rows_pad + 63) / 64);
        pj.tile_end[j] = total;
    }
    pj.njobs = njobs;
    hipLaunchKernelGGL(pack_weights_multi3_kernel, dim3(total), dim3(256), 0, st, pj);
    return 0;
}

extern "C" int alm_pack_weights_multi(const AlmPackJob* jobs, int njobs, void* stream) {
    if (njobs <= 0) return 0;
    if (!jobs) return ALM_ERR_BAD_ARG;
    static const int wide_on = [] { const char* e = getenv("ALM_PACK_WIDE"); return e ? atoi(e) : 1; }();     // A/B switch: 0 = the round-2 kernels
    bool vec = true, wide = wide_on != 0;
    for (int j = 0; j < njobs; ++j) {
        const AlmPackJob& q = jobs[j];
        if (q.rows <= 0 || q.cols <= 0 || q.rows_pad < q.rows || q.cols_pad < q.cols) return ALM_ERR_BAD_ARG;
        if ((q.dst && q.ld_dst < q.cols_pad) || (q.dstT && q.ld_dstT < q.rows_pad)) return ALM_ERR_BAD_ARG;
        vec = vec && !((q.ld_src | q.ld_dst | q.ld_dstT | q.rows_pad | q.cols_pad) & 1) && !((uintptr_t)q.src & 7) && !((uintptr_t)q.dst & 3) &&
              !((uintptr_t)q.dstT & 3);
        wide = wide && q.cols >= 4 && !((q.ld_dst | q.ld_dstT | q.rows_pad | q.cols_pad) & 7) && !((uintptr_t)q.src & 3) && !((uintptr_t)q.dst & 15) && !((uintptr_t)q.dstT & 15);
    }
    hipStream_t st = (hipStream_t)stream;
    if (wide) {
        for (int j0 = 0; j0 < njobs; j0 += PACK3_JOBS) {
            int rc = pack3_launch(jobs + j0, std::min(PACK3_JOBS, njobs - j0), st);
            if (rc) return rc;
        }
        ALM_LAUNCH_CHECK();
        return 0;
    }
    for (int j0 = 0; j0 < njobs; j0 += 8) {
        const int nj = std::min(8, njobs - j0);
        if (vec) {
            PackJobs2 pj{};
            int total = 0;
            for (int j = 0; j < nj; ++j) {
                pj.job[j] = jobs[j0 + j];
                total += ((jobs[j0 + j].cols_pad + 127) / 128) * ((jobs[j0 + j].rows_pad + 63) / 64);
                pj.tile_end[j] = total;
            }
            pj.njobs = nj;
            hipLaunchKernelGGL(pack_weights_multi2_kernel, dim3(total), dim3(256), 0, st, pj);
        } else {
            PackJobs pj{};
            int total = 0;
            for (int j = 0; j < nj; ++j) {
                const AlmPackJob& q = jobs[j0 + j];
                pj.job[j] = q;
                total += ((q.cols_pad + 63) / 64) * ((q.rows_pad + 63) / 64);
                pj.tile_end[j] = total;
            }
            pj.njobs = nj;
            hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total), dim3(256), 0, st, pj);
        }
    }
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_pack_weight(const float* src, int rows, int cols, long long ld_src, void* dst, long long ld_dst, int rows_pad,
                               int cols_pad, void* dstT, long long ld_dstT, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    if (rows_pad < rows || cols_pad < cols) return ALM_ERR_BAD_ARG;
    if (dst && ld_dst < cols_pad) return ALM_ERR_BAD_ARG;
    if (dstT && ld_dstT < rows_pad) return ALM_ERR_BAD_ARG;
    dim3 grid((cols_pad + 63) / 64, (rows_pad + 63) / 64);
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, rows, cols, ld_src, (bf16_t*)dst, ld_dst, rows_pad,
                       cols_pad, (bf16_t*)dstT, ld_dstT);
    ALM_LAUNCH_CHECK();
    return 0;
}

#ifdef ALM_GEMM_PROBE
extern "C" int alm_gemm_probe_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_probe), sizeof(unsigned long long) * n);
}
#endif
