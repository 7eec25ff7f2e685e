// BENCH-ONLY translation unit (libaudiolm_gemm_lab.so; never loaded by the package): every GEMM main-loop / scheduling variant that was built,
// measured on MI355X and NOT adopted, kept runnable so the negative results in DESIGN.md section 8.1 stay reproducible
// (scripts/ab_gemm.py, scripts/kbench.py wgrad, tests/test_gpu_gemm_lab.py):
//   tile 3  256x128, 3-stage DMA ring with counted vmcnt           tile 4  persistent 256x256 (one workgroup per CU, cross-tile prefetch)
//   tile 6/7  hand software-pipelined fragment reads (4 / 8 waves)  tile 8/9  B operand streamed from L2 into registers
//   tile 10/12  32-deep K-steps with a 4- / 3-stage DMA ring        balanced ("stream-K"-like) split for weight gradients (almlab_debug_stream)
//   -DALM_GEMM_WHATIF=n diagnostic builds (WRONG results: which resource bounds the main loop), -DALM_GEMM_SPREAD
// plus the lock-step 256x256 tile (2), the staggered tile (13) and 384x256 (11) as they were when the experiments ran.  Symbols: almlab_*.
// It is a snapshot of csrc/gemm.hip before the product file was reduced to the adopted kernels.
//
// bf16 MFMA GEMMs for gfx950 (MI355X): the dense-contraction workhorse of the token-transformer hot path.
//
//   NT:  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T (+ bias[N])     both operands K-contiguous: forward and dgrad of every
//        nn.Linear / einsum (to_q / to_kv / to_out, the two FFN projections, the logit heads).
//        Replaces aten::mm / addmm / bmm at reference audiolm_pytorch.py:255-259, :351, :395, :719, :961, :972.
//   TN:  C[M,N] (+)= alpha * At[K,M]^T . Bt[K,N]               both operands contraction-major (K = tokens): every weight
//        gradient dW = dY^T X straight from the row-major activations -- no transposed copies of dY / X are ever made.
//
// Design (wave64 / CDNA4):
//   * block tile 128x128 (4 waves, 2x2) or 256x256 (8 waves as 2(M) x 4(N), wave tile 128x64), K-step 64, MFMA 32x32x16 bf16.
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
#include <array>
#include <map>
#include <vector>

#include "../common.hpp"
#include "../../../include/audiolm_hip.h"      // error codes only
#include "gemm_lab.h"

namespace {

constexpr int BK = 64;
constexpr int GROUP_M = 8;
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
    int raster;        // 1: split-K launches -- 1-D grid, XCD-panel rasterisation (see kernel)
    int nsl;           // number of K slices (raster 1)
    const int* units;  // non-null: BALANCED split -- 1-D grid, workgroup b executes work unit units[4b .. 4b+3] = (linear tile, first K-step,
                       // K-steps, workspace slot) and writes a whole BM x BN fp32 partial tile to C + slot * BM * BN (see build_stream_plan)
};

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
                    if (gm < p.M && gn < p.N) *reinterpret_cast<uint4*>(Cb + ((long long)gm * p.ldc + gn) * ES) = val;
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {           // counted wait: at most N vector-memory operations (here: LDS DMA pieces) still in flight
    static_assert(N == 0 || N == 6 || N == 8 || N == 10 || N == 16, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// STAGES = 2: one __syncthreads per K-step, the next step's DMA is in flight during the current step's MFMAs (prefetch distance 1).
// STAGES = 3: prefetch distance 2 -- the DMA of step k+2 is issued at step k and stays in flight ACROSS the barrier of step k: the barrier
//             is a raw s_barrier behind a COUNTED s_waitcnt vmcnt(pieces of one stage), so only step k+1's data is waited for.
template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32, int STAGES = 2, bool PIPE = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / WM / 32, TNB = BN / WN / 32;
    constexpr int NIA = BM / 8 / NW, NIB = BN / 8 / NW;       // 1-KiB DMA pieces per wave per stage
    static_assert(NIA >= 1 && NIB >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile / wave shape");

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn, zb, zs;
    int ukbeg = 0, uksteps = 0, uslot = 0;
    if (p.units) {
        const int4 u = reinterpret_cast<const int4*>(p.units)[blockIdx.x];
        if (u.x < 0) return;
        const int per = tiles_m * tiles_n, r = u.x % per;
        zb = u.x / per;
        if (tiles_m >= tiles_n) { tm = r / tiles_n; tn = r % tiles_n; } else { tn = r / tiles_m; tm = r % tiles_m; }
        zs = 0;
        ukbeg = u.y; uksteps = u.z; uslot = u.w;
    } else if (p.raster == 1) {
        // split-K weight gradients: the operand with MANY tile panels (e.g. dU: 22 panels of 256 columns) is the big one.  Panel q
        // (its K slices and the few tiles along the other dimension) is pinned to XCD q % 8 (workgroup L runs on XCD L % 8 -- observed
        // dispatch order; a wrong guess only costs speed), so each of its K slices is fetched from HBM once and shared through that
        // XCD's L2; only the small operand is fetched by every XCD.
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const bool m_major = tiles_m >= tiles_n;
        const int tmaj = m_major ? tiles_m : tiles_n, Q = m_major ? tiles_n : tiles_m;
        const int P = tmaj * p.nb2;
        const int PL = (P + 7) / 8;
        zs = j / (PL * Q);
        const int rem = j % (PL * Q);
        const int panel = (rem / Q) * 8 + xcd;
        const int minor = rem % Q;
        if (panel >= P || zs >= p.nsl) return;
        zb = panel / tmaj;
        const int tmajor = panel % tmaj;
        tm = m_major ? tmajor : minor;
        tn = m_major ? minor : tmajor;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int bid = xcd_remap(blockIdx.x, nwg);
        const int per_group = GROUP_M * tiles_n;
        const int group = bid / per_group;
        const int first_m = group * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        tm = first_m + (bid % per_group) % gsz;
        tn = (bid % per_group) / gsz;
        zb = blockIdx.y;
        zs = blockIdx.z;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    int kbeg = 0, Krem = p.K;
    long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    if (p.units) {
        kbeg = ukbeg * BK;
        Krem = min(p.K - kbeg, uksteps * BK);
    } else if (p.ksplit > 0) {
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
#if defined(ALM_GEMM_WHATIF) && (ALM_GEMM_WHATIF == 6 || ALM_GEMM_WHATIF == 7 || ALM_GEMM_WHATIF == 9 || ALM_GEMM_WHATIF == 10)
        Ab = p.A + zoffA + kbeg;
        Bb = p.B + zoffB + kbeg;
#else
        Ab = p.A + zoffA + (long long)m0 * p.lda + kbeg;
        Bb = p.B + zoffB + (long long)n0 * p.ldb + kbeg;
#endif
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

    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * STAGE;
        const int kleft = Krem - kt * BK;
#if defined(ALM_GEMM_WHATIF) && ALM_GEMM_WHATIF == 9
#pragma unroll
        for (int j = 0; j < NIA + NIB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
        return;
#elif defined(ALM_GEMM_WHATIF) && ALM_GEMM_WHATIF == 10
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
#pragma unroll
        for (int j = 0; j < NIA; ++j) { u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[j], kt * kstepA, 0); asm volatile("" ::"v"(v)); }
#pragma unroll
        for (int j = 0; j < NIB; ++j) { u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[j], kt * kstepB, 0); asm volatile("" ::"v"(v)); }
        return;
#endif
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const unsigned vo = (kcA[j] < kleft) ? offA[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, vo, kt * kstepA, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const unsigned vo = (kcB[j] < kleft) ? offB[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (j * NW + wave) * 1024), 16, vo, kt * kstepB, 0, 0);
        }
    };

    // one quarter of a stage (STAGES == 2, NT: NIA == NIB == 4 on the 8-wave tile): piece j of A and piece j of B.  Issued one quarter per
    // sub-step, the DMA instructions queue up behind the memory pipeline while this wave's MFMAs run, instead of blocking the wave's
    // in-order issue for the whole burst at the top of the K-step
    auto stage_part = [&](int kt, int buf, int j0, int j1) {
        unsigned char* base = smem + buf * STAGE;
        const int kleft = Krem - kt * BK;
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            if (j >= j0 && j < j1) {
                const unsigned vo = (kcA[j] < kleft) ? offA[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, vo, kt * kstepA, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < NIB; ++j)
            if (j >= j0 && j < j1) {
                const unsigned vo = (kcB[j] < kleft) ? offB[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (j * NW + wave) * 1024), 16, vo, kt * kstepB, 0, 0);
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
    stage(0, 0);
    if (STAGES == 3) {
        if (nk > 1) { stage(1, 1); wait_vmcnt<NIA + NIB>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    } else {
        __syncthreads();
    }

    // fragment reads of sub-step `ks` of the stage at `sb` into register set `fb`
    bf16x8 a[2][TM], b[2][TNB];
    auto load_frags = [&](const unsigned char* sb, int ks, auto fbc) {
        constexpr int fb = decltype(fbc)::value;
        if (!TNMODE) {
            const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[fb][i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
#pragma unroll
            for (int j = 0; j < TNB; ++j) b[fb][j] = *reinterpret_cast<const bf16x8*>(sb + fragB[j] + co);
        } else {
            // k-groups ks*4 + (g>>1)*2 + {0, 1}; consecutive k-groups are (Bx/16)*128 bytes apart
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* ad = sb + fragA[i] + ks * 4 * (BM / 16) * 128;
                a[fb][i] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + (BM / 16) * 128), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < TNB; ++j) {
                const unsigned char* ad = sb + fragB[j] + ks * 4 * (BN / 16) * 128;
                b[fb][j] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + (BN / 16) * 128), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto mfmas = [&](auto fbc) {
        constexpr int fb = decltype(fbc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[fb][j], a[fb][i], acc[i][j], 0, 0, 0);     // C^T block: lane = row m
    };
    using FB0 = std::integral_constant<int, 0>;
    using FB1 = std::integral_constant<int, 1>;

    int buf = 0;
    if (PIPE) {
        // Software-pipelined main loop (2 stages): the fragment reads of sub-step ks + 1 are issued BEFORE the MFMAs of sub-step ks into the
        // other register set, with scheduling fences so that the compiler keeps that order (left to itself it emits read, wait, 4 MFMAs,
        // read, wait, ...: every LDS latency exposed).  The hand-over barrier of a K-step sits before the LAST sub-step's MFMAs: by then this
        // wave's reads of the stage are complete (its last fragments are in registers), so after the barrier the first fragments of the next
        // stage are fetched underneath those MFMAs.
        static_assert(!PIPE || STAGES == 2, "pipelined loop: 2 stages");
        load_frags(smem, 0, FB0{});
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
            const unsigned char* sb = smem + buf * STAGE;
            load_frags(sb, 1, FB1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(FB0{});
            __builtin_amdgcn_sched_barrier(0);
            load_frags(sb, 2, FB0{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(FB1{});
            __builtin_amdgcn_sched_barrier(0);
            load_frags(sb, 3, FB1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(FB0{});
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                   // reads of `buf` done (lgkmcnt 0), next stage landed (vmcnt 0), all waves agree
            buf ^= 1;
            if (kt + 1 < nk) load_frags(smem + buf * STAGE, 0, FB0{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(FB1{});
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            if (STAGES == 3) {
                if (kt + 2 < nk) stage(kt + 2, buf >= 1 ? buf - 1 : 2);       // (kt + 2) % 3: the buffer read during step kt - 1
            } else {
#if defined(ALM_GEMM_WHATIF) && (ALM_GEMM_WHATIF == 3 || ALM_GEMM_WHATIF == 8)      // no DMA after the first stage
                if (kt + 1 < nk && kt < 0) stage(kt + 1, buf ^ 1);
#elif defined(ALM_GEMM_SPREAD)
                // issued in quarters inside the sub-step loop below
#else
                if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
#endif
            }
            const unsigned char* sb = smem + buf * STAGE;
#ifdef ALM_GEMM_WHATIF      // diagnostic builds (WRONG results; scripts/ab_gemm.py): which resource bounds the main loop
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#if ALM_GEMM_WHATIF == 1        // no B fragment reads
                if (kt == 0 && ks == 0) load_frags(sb, ks, FB0{});
                else {
                    const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
                }
                mfmas(FB0{});
#elif ALM_GEMM_WHATIF == 2 || ALM_GEMM_WHATIF == 8      // no fragment reads at all (8: and no DMA: MFMA issue + barriers only)
                if (kt == 0 && ks == 0) load_frags(sb, ks, FB0{});
                mfmas(FB0{});
#elif ALM_GEMM_WHATIF == 5 || ALM_GEMM_WHATIF == 6 || ALM_GEMM_WHATIF == 7 || ALM_GEMM_WHATIF == 9 || ALM_GEMM_WHATIF == 10     // (9: as 7, DMA sources confined to 1 KB = L1 hits; 10: as 7, plain register loads instead of the LDS DMA)
                // DMA only (6: every workgroup fetches tile (0, 0): all L2 hits; 7: MFMA + DMA, no fragment reads, tile (0, 0))
#if ALM_GEMM_WHATIF == 7 || ALM_GEMM_WHATIF == 9 || ALM_GEMM_WHATIF == 10
                if (kt == 0 && ks == 0) load_frags(sb, ks, FB0{});
                mfmas(FB0{});
#endif
#elif ALM_GEMM_WHATIF == 4      // no MFMAs
                load_frags(sb, ks, FB0{});
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(a[0][i]));
#pragma unroll
                for (int j = 0; j < TNB; ++j) asm volatile("" ::"v"(b[0][j]));
#else
                load_frags(sb, ks, FB0{});
                mfmas(FB0{});
#endif
            }
#elif defined(ALM_GEMM_SPREAD)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                load_frags(sb, ks, FB0{});
                if (STAGES == 2 && kt + 1 < nk) stage_part(kt + 1, buf ^ 1, ks * NIA / 4, (ks + 1) * NIA / 4);
                mfmas(FB0{});
            }
#else
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                load_frags(sb, ks, FB0{});
                mfmas(FB0{});
            }
#endif
            if (STAGES == 3) {
                // this wave's LDS reads of the step are complete (their results fed the MFMAs above); step kt+1's DMA must have landed,
                // step kt+2's (just issued) may stay in flight
                if (kt + 2 < nk) wait_vmcnt<NIA + NIB>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                buf = buf == 2 ? 0 : buf + 1;
            } else {
                __syncthreads();
                buf ^= 1;
            }
        }
    }

    // ---- epilogue: slabs reuse the (now idle) stage buffers ------------------------------------------------------------------------
    {
        constexpr int ES = OUT_F32 ? 4 : 2;
        static_assert(NW * 32 * (32 * TNB * ES) <= STAGES * STAGE, "epilogue slab");
        if (OUT_F32 && p.units) {
            // balanced split: the whole BM x BN partial tile (rows / columns beyond M / N hold exact zeros) goes to its workspace slot
            GemmParams q = p;
            q.C = reinterpret_cast<float*>(p.C) + (long long)uslot * BM * BN;
            q.ldc = BN; q.M = BM; q.N = BN; q.accumulate = 0; q.bias = nullptr;
            gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(q, acc, smem, 0, 0, 0, wave, wr, wc, lane, lr, lh);
            return;
        }
        const long long coff0 = z1 * p.sC1 + z2 * p.sC2 + (p.ksplit > 0 ? zs * p.sCk : 0);
        gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
    }
}

// ---- staggered 256 x 256 x 64 kernel (8 waves, NT and TN): the two wave rows (wr = 0 / 1: the two waves co-resident on each SIMD) run
// HALF A K-STEP APART.  Why: in the lock-step kernel above every wave issues its 8 LDS-DMA instructions at the top of a K-step -- at
// 60-185 issue cycles each (MI355X_MICROARCH: "LDS-DMA piece issue cost") that is ~1000 cycles during which neither wave of the SIMD feeds
// the matrix pipe (measured: MFMA-only loop 601 us, DMA-only 716 us, both 886 us at 8192^3: they do not overlap).
// Here the K-loop is a sequence of half-step SLOTS closed by one workgroup barrier each; in a slot one wave row issues the DMA of a later
// stage (LOAD slot) while the other reads fragments and runs its 32 MFMAs (COMPUTE slot), then they swap:
//     wave row g, slot s, q = s - g:   q even  -> LOAD stage q / 2          q odd -> COMPUTE stage (q - 3) / 2
// so a stage is loaded 3 slots before the loading row computes on it, its DMA is waited for (vmcnt(0)) at the end of the row's next
// COMPUTE slot -- a whole slot of MFMAs later -- and every wave executes exactly one barrier per slot (2 nk + 3 slots: no wave ever waits
// on a barrier the others skip).
// LDS (160 KB): the A operand is PRIVATE to a wave row (row g only reads tile rows g*128 .. +127): 2 buffers x 16 KB per row; the B
// operand is shared by both rows and lives 5 slots (written in slots 2j, 2j+1, read in 2j+3, 2j+4): 3 buffers x 32 KB.  Row g loads its
// own A half and the B pieces [16 g, 16 g + 16): 8 one-KiB DMA pieces per wave and stage, as before.
template <bool TNMODE, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_stag_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8, TM = 4, TNB = 2;
    constexpr int AH_BYTES = 128 * BK * 2, B_BYTES = BN * BK * 2;           // 16 KB, 32 KB
    constexpr int B_BASE = 4 * AH_BYTES;                                    // [A row 0: buf 0, 1][A row 1: buf 0, 1][B: buf 0, 1, 2]
    static_assert(B_BASE + 3 * B_BYTES == 163840, "LDS map");

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn, zb, zs;
    if (p.raster == 1) {                                                    // split-K weight gradients: XCD-panel rasterisation (see gemm_kernel)
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const bool m_major = tiles_m >= tiles_n;
        const int tmaj = m_major ? tiles_m : tiles_n, Q = m_major ? tiles_n : tiles_m;
        const int P = tmaj * p.nb2;
        const int PL = (P + 7) / 8;
        zs = j / (PL * Q);
        const int rem = j % (PL * Q);
        const int panel = (rem / Q) * 8 + xcd;
        const int minor = rem % Q;
        if (panel >= P || zs >= p.nsl) return;                              // whole workgroup: no barrier is skipped by a subset
        zb = panel / tmaj;
        const int tmajor = panel % tmaj;
        tm = m_major ? tmajor : minor;
        tn = m_major ? minor : tmajor;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int bid = xcd_remap(blockIdx.x, nwg);
        const int per_group = GROUP_M * tiles_n;
        const int group = bid / per_group;
        const int first_m = group * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        tm = first_m + (bid % per_group) % gsz;
        tn = (bid % per_group) / gsz;
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

    auto stage = [&](int j) {
        unsigned char* ad = myA + (j & 1) * AH_BYTES;
        unsigned char* bd = smem + B_BASE + (j % 3) * B_BYTES;
        const int kleft = Krem - j * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = (kcA[i] < kleft) ? offA[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(ad + dstA[i]), 16, vo, j * kstepA, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = (kcB[i] < kleft) ? offB[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(bd + dstB[i]), 16, vo, j * kstepB, 0, 0);
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

    auto compute = [&](int c) {
        const unsigned char* sa = myA + (c & 1) * AH_BYTES;
        const unsigned char* sb = smem + B_BASE + (c % 3) * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[TM], b[TNB];
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
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = (Krem + BK - 1) / BK;
    // slot pair i = slots 2 i, 2 i + 1 (written out per wave row: q = s - wr); nk + 2 pairs cover the 2 nk + 3 slots
    if (wr == 0) {
        for (int i = 0; i < nk + 2; ++i) {
            if (i < nk) stage(i);
            __builtin_amdgcn_s_barrier();                                   // raw: this slot's DMA stays in flight across it
            if (i >= 1 && i <= nk) compute(i - 1);
            __syncthreads();                                                // vmcnt(0): the DMA issued one slot ago has landed; lgkmcnt(0): reads done
        }
    } else {
        for (int i = 0; i < nk + 2; ++i) {
            if (i >= 2) compute(i - 2);
            __syncthreads();
            if (i < nk) stage(i);
            __builtin_amdgcn_s_barrier();
        }
    }

    const long long coff0 = z1 * p.sC1 + z2 * p.sC2 + (p.ksplit > 0 ? zs * p.sCk : 0);
    gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
}

// ---- persistent NT kernel (256x256x64, 8 waves): one workgroup per CU walks its share of the output tiles.  The DMA of the NEXT tile's
// first K-step is issued before the epilogue of the current tile (into stage 0; the epilogue's transpose slabs live in stage 1), so the
// cold-start latency of a tile and the drain of its output stores overlap -- this is what the K = 1024 shapes (W1 forward, dHN) lose
// ~30 % of a tile to in the one-tile-per-workgroup kernel.  Same arithmetic, same rasterisation (tile v of workgroup b: v = b + i * grid).
template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
    constexpr int A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;
    constexpr int TM = 4, TNB = 2, NIA = 4, NIB = 4;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int zb = blockIdx.y;
    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    const long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    const long long coff0 = z1 * p.sC1 + z2 * p.sC2;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    unsigned offA[NIA], offB[NIB];
    int kc;                                                     // K coordinate (within a stage) of this lane's pieces
    {
        const int row0 = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row0 >> 1) & 7);           // (row >> 1) & 7 is the same for every piece of a lane (pieces are 64 rows apart)
        kc = c * 8;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            offA[j] = (unsigned)(row * p.lda * 2 + c * 16);
            offB[j] = (unsigned)(row * p.ldb * 2 + c * 16);
        }
    }
    unsigned fragA[TM], fragB[TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((wr * 128 + i * 32 + lr) * 128);
#pragma unroll
    for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)(A_BYTES + (wc * 64 + j * 32 + lr) * 128);
    const unsigned sw = (unsigned)((lr >> 1) & 7);

    auto coords = [&](int v, int& m0, int& n0) {
        const int bid = xcd_remap(v, nwg);
        const int per_group = GROUP_M * tiles_n;
        const int group = bid / per_group;
        const int first_m = group * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (bid % per_group) % gsz) * BM;
        n0 = ((bid % per_group) / gsz) * BN;
    };
    auto stage = [&](int m0, int n0, int kt, int buf) {
        const bf16_t* Ab = p.A + zoffA + (long long)m0 * p.lda;
        const bf16_t* Bb = p.B + zoffB + (long long)n0 * p.ldb;
        const long long extA = ((long long)(min(p.M - m0, BM) - 1) * p.lda + p.K) * 2;
        const long long extB = ((long long)(min(p.N - n0, BN) - 1) * p.ldb + p.K) * 2;
        const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)min(extA, 0x7fffffffLL), 0x00020000);
        const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)min(extB, 0x7fffffffLL), 0x00020000);
        unsigned char* base = smem + buf * STAGE;
        const bool kok = kc < p.K - kt * BK;
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, kok ? offA[j] : OOB, kt * (BK * 2), 0, 0);
#pragma unroll
        for (int j = 0; j < NIB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (j * NW + wave) * 1024), 16, kok ? offB[j] : OOB, kt * (BK * 2), 0, 0);
    };

    const int nk = (p.K + BK - 1) / BK;
    int v = blockIdx.x;
    if (v >= nwg) return;
    int m0, n0;
    coords(v, m0, n0);
    stage(m0, n0, 0, 0);
    for (;;) {
        __syncthreads();            // this tile's first K-step has landed; everyone has left the previous tile's epilogue (slabs in stage 1)
        f32x16 acc[TM][TNB];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage(m0, n0, kt + 1, buf ^ 1);
            const unsigned char* sb = smem + buf * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[TM], b[TNB];
                const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
#pragma unroll
                for (int j = 0; j < TNB; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + fragB[j] + co);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        // both stages are idle now: fetch the next tile's first K-step into stage 0 while this tile drains through stage 1
        const int vn = v + (int)gridDim.x;
        const int cm0 = m0, cn0 = n0;
        const bool more = vn < nwg;
        if (more) {
            coords(vn, m0, n0);
            stage(m0, n0, 0, 0);
        }
        gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem + STAGE, coff0, cm0, cn0, wave, wr, wc, lane, lr, lh);
        if (!more) break;
        v = vn;
    }
}

// ---- NT kernel with the B operand streamed from L2 straight into registers (256x256x64, 8 waves; tiles 8 / 9) ------------------------------
// The plain 256x256 tile keeps the LDS port as busy as the matrix cores (per 64-deep K-step: 192 KB of fragment reads + 64 KB of DMA writes at
// 128 B/clk = 2030 cycles vs 2048 MFMA cycles per SIMD), and re-ordering its issue slots does not help.  Here only A goes
// through the LDS; every lane fetches its own B fragments (row n0 + wc*64 + j*32 + lr, 16-byte k-chunk 2*ks + lh -- exactly the MFMA operand
// layout, so no shuffle is needed) with buffer_load_dwordx4 one K-step ahead into a second register set.  LDS traffic per K-step: 128 KB of A
// fragment reads + 32 KB of DMA writes = 160 KB (-38 %); the price is 64 KB per K-step of L2 -> register traffic per CU (the two waves that
// share a B row block fetch it twice; the second fetch hits the vector L1) and 32 more VGPRs.  Same accumulation order as the plain tile:
// bit-identical results.  PIPE: A fragment reads of sub-step ks + 1 issued before the MFMAs of sub-step ks (second fragment register set).
// NEGATIVE RESULT (MI355X, profiles/r1_run12_ab_gemm.log): 22-30 % SLOWER than the plain tile on every shape (W1 fwd 231 vs 179 us, 8192^3
// 1197 vs 832 us), plain and pipelined loop alike -- so the limit moved from the LDS to the vector-memory path: the MFMA operand layout gives
// a row only 2 lanes, i.e. every buffer_load_dwordx4 touches 32 cache lines for 32 bytes each (~32 tag cycles instead of 16 data cycles),
// 8 loads x 8 waves = 2048 cycles per K-step before the A DMA is counted.  Kept selectable (tiles 8 / 9) and tested; not used.
template <bool OUT_F32, bool PIPE>
__global__ __launch_bounds__(512) void gemm_nt_bdirect_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int TM = 4, TNB = 2, NIA = 4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int m0, n0;
    {
        const int bid = xcd_remap(blockIdx.x, nwg);
        const int per_group = GROUP_M * tiles_n;
        const int group = bid / per_group;
        const int first_m = group * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (bid % per_group) % gsz) * BM;
        n0 = ((bid % per_group) / gsz) * BN;
    }
    const int zb = blockIdx.y;
    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    const long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    const long long coff0 = z1 * p.sC1 + z2 * p.sC2;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    const bf16_t* Ab = p.A + zoffA + (long long)m0 * p.lda;
    const bf16_t* Bb = p.B + zoffB + (long long)n0 * p.ldb;
    const long long extA = ((long long)(min(p.M - m0, BM) - 1) * p.lda + p.K) * 2;
    const long long extB = ((long long)(min(p.N - n0, BN) - 1) * p.ldb + p.K) * 2;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)min(extA, 0x7fffffffLL), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)min(extB, 0x7fffffffLL), 0x00020000);

    unsigned offA[NIA], offBd[TNB];
    int kc;                                                     // K coordinate (within a stage) of this lane's DMA pieces
    {
        const int row0 = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row0 >> 1) & 7);           // (row >> 1) & 7 is the same for every piece of a lane (pieces are 64 rows apart)
        kc = c * 8;
#pragma unroll
        for (int j = 0; j < NIA; ++j) offA[j] = (unsigned)(((j * NW + wave) * 8 + (lane >> 3)) * p.lda * 2 + c * 16);
#pragma unroll
        for (int j = 0; j < TNB; ++j) offBd[j] = (unsigned)((wc * 64 + j * 32 + lr) * p.ldb * 2 + lh * 16);
    }
    unsigned fragA[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((wr * 128 + i * 32 + lr) * 128);
    const unsigned sw = (unsigned)((lr >> 1) & 7);

    auto stage_a = [&](int kt, int buf) {
        unsigned char* base = smem + buf * A_BYTES;
        const bool kok = kc < p.K - kt * BK;
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, kok ? offA[j] : OOB, kt * (BK * 2), 0, 0);
    };
    bf16x8 bq[2][TNB][4];                                       // B fragments of two K-steps: [set][32-column block][sub-step]
    auto load_b = [&](int kt, auto setc) {
        constexpr int st = decltype(setc)::value;
        const int kleft = p.K - kt * BK;
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned vo = ((ks * 2 + lh) * 8 < kleft) ? offBd[j] + ks * 32 : OOB;
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, kt * (BK * 2), 0);
                bq[st][j][ks] = __builtin_bit_cast(bf16x8, raw);
            }
    };
    bf16x8 a[2][TM];
    auto load_a = [&](const unsigned char* sb, int ks, auto fbc) {
        constexpr int fb = decltype(fbc)::value;
        const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[fb][i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
    };
    f32x16 acc[TM][TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfmas = [&](auto fbc, auto setc, auto ksc) {
        constexpr int fb = decltype(fbc)::value, st = decltype(setc)::value, ks = decltype(ksc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st][j][ks], a[fb][i], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    const int nk = (p.K + BK - 1) / BK;
    int buf = 0;
    stage_a(0, 0);
    load_b(0, I0{});
    __syncthreads();
    if (PIPE) load_a(smem, 0, I0{});
    auto kstep = [&](int kt, auto cur, auto nxt) {
        // UNCONDITIONAL prefetch (beyond the last K-step every offset is out of bounds: zeros, no traffic): with an `if` around it the
        // compiler has to place s_waitcnt for the path WITHOUT the new loads, i.e. vmcnt(7) instead of vmcnt(19), which makes the MFMAs of
        // this K-step wait for the loads issued a moment ago
        stage_a(kt + 1, buf ^ 1);
        load_b(kt + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);                     // keep the prefetch at the top of the K-step (the scheduler otherwise sinks it to the end)
        const unsigned char* sb = smem + buf * A_BYTES;
        if (PIPE) {
            load_a(sb, 1, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{}, cur, I0{});
            __builtin_amdgcn_sched_barrier(0);
            load_a(sb, 2, I0{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{}, cur, I1{});
            __builtin_amdgcn_sched_barrier(0);
            load_a(sb, 3, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{}, cur, I2{});
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                   // reads of `buf` done, next stage (DMA + B registers) landed, all waves agree
            buf ^= 1;
            load_a(smem + buf * A_BYTES, 0, I0{});               // (after the last K-step: zeros, unused)
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{}, cur, I3{});
            __builtin_amdgcn_sched_barrier(0);
        } else {
            load_a(sb, 0, I0{});
            mfmas(I0{}, cur, I0{});
            load_a(sb, 1, I0{});
            mfmas(I0{}, cur, I1{});
            load_a(sb, 2, I0{});
            mfmas(I0{}, cur, I2{});
            load_a(sb, 3, I0{});
            mfmas(I0{}, cur, I3{});
            __syncthreads();
            buf ^= 1;
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, I0{}, I1{});
        if (kt + 1 < nk) kstep(kt + 1, I1{}, I0{});
    }
    static_assert(NW * 32 * (32 * TNB * (OUT_F32 ? 4 : 2)) <= 2 * A_BYTES, "epilogue slab");
    gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
}

// ---- NT kernels with a deep DMA ring (32-deep K-steps, NST stages; tile 10: 256x256 / 8 waves / 4 x 32 KB, tile 12: 256x128 / 4 waves /
// 3 x 24 KB with two workgroups per CU) ---------------------------------------------------------------------------------------------------
// Built to test "the 2-stage loop is latency-bound on the operand fetch": NST - 1 stages in flight, a COUNTED s_waitcnt (only the stage about
// to be read), one s_barrier per K-step that both publishes that stage and frees the buffer read in the previous K-step, which is refilled
// at once.  LDS image: [row][32 k] (64-byte rows); the DMA destination is lane-linear (16 rows x 64 B per instruction), so the
// conflict-avoiding swizzle slot ^= (row >> 2) & 3 is applied to the source chunk and again on the ds_read_b128 fragment reads (16
// consecutive rows x one chunk = 16 distinct 16-byte bank groups).  The prefetch is unconditional (beyond K every offset is out of bounds:
// zeros, no traffic) so the vmcnt accounting is uniform.  Same accumulation order as the 2-stage tile: bit-identical results.
// NEGATIVE RESULT (MI355X, profiles/r1_run15_ab_gemm_ring.log): tile 10 is 2-5 % SLOWER than the 2-stage tile (4 and 5 stages alike: prefetch
// depth is not the limit), tile 12 25-30 % slower (1.5x the L2 -> LDS bytes per flop).  Together with the what-if builds this says the
// operand fetch is a THROUGHPUT cost that only partly overlaps the MFMAs -- fewer bytes per flop is the lever, not more stages.
template <bool OUT_F32, int NST, int BN, int WN>
__global__ __launch_bounds__(2 * WN * 64) void gemm_nt_ring_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // the host pass needs the stub only (and drops the stub silently when it instantiates this body)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int BM = 256, WM = 2, NW = WM * WN, KB = 32;
    constexpr int A_BYTES = BM * KB * 2, B_BYTES = BN * KB * 2, STAGE = A_BYTES + B_BYTES;      // 16 KB + 16 KB (BN = 256) / 16 KB + 8 KB (BN = 128)
    constexpr int TM = 4, TNB = 2, NIA = BM * KB * 2 / 1024 / NW, NIB = BN * KB * 2 / 1024 / NW;   // DMA pieces per wave per stage
    constexpr int DIST = NST - 1;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int m0, n0;
    {
        const int bid = xcd_remap(blockIdx.x, nwg);
        const int per_group = GROUP_M * tiles_n;
        const int group = bid / per_group;
        const int first_m = group * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (bid % per_group) % gsz) * BM;
        n0 = ((bid % per_group) / gsz) * BN;
    }
    const int zb = blockIdx.y;
    const int z1 = zb / p.nb2, z2 = zb % p.nb2;
    const long long zoffA = z1 * p.sA1 + z2 * p.sA2, zoffB = z1 * p.sB1 + z2 * p.sB2;
    const long long coff0 = z1 * p.sC1 + z2 * p.sC2;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    const bf16_t* Ab = p.A + zoffA + (long long)m0 * p.lda;
    const bf16_t* Bb = p.B + zoffB + (long long)n0 * p.ldb;
    const long long extA = ((long long)(min(p.M - m0, BM) - 1) * p.lda + p.K) * 2;
    const long long extB = ((long long)(min(p.N - n0, BN) - 1) * p.ldb + p.K) * 2;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)min(extA, 0x7fffffffLL), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)min(extB, 0x7fffffffLL), 0x00020000);

    unsigned offA[NIA], offB[NIB];
    int kc;                                                     // K coordinate (within a stage) of this lane's chunk
    {
        const int r0 = lane >> 2;                               // row within a 16-row piece
        const int c = (lane & 3) ^ ((r0 >> 2) & 3);             // pieces start at multiples of 16 rows: (row >> 2) & 3 == (r0 >> 2) & 3
        kc = c * 8;
#pragma unroll
        for (int j = 0; j < NIA; ++j) offA[j] = (unsigned)(((j * NW + wave) * 16 + r0) * p.lda * 2 + c * 16);
#pragma unroll
        for (int j = 0; j < NIB; ++j) offB[j] = (unsigned)(((j * NW + wave) * 16 + r0) * p.ldb * 2 + c * 16);
    }
    unsigned fragA[TM], fragB[TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i) fragA[i] = (unsigned)((wr * 128 + i * 32 + lr) * 64);
#pragma unroll
    for (int j = 0; j < TNB; ++j) fragB[j] = (unsigned)(A_BYTES + (wc * 64 + j * 32 + lr) * 64);      // wave tile 128 x 64 in both geometries
    const unsigned sw = (unsigned)((lr >> 2) & 3);

    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * STAGE;
        const bool kok = kc < p.K - kt * KB;
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (j * NW + wave) * 1024), 16, kok ? offA[j] : OOB, kt * (KB * 2), 0, 0);
#pragma unroll
        for (int j = 0; j < NIB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (j * NW + wave) * 1024), 16, kok ? offB[j] : OOB, kt * (KB * 2), 0, 0);
    };

    f32x16 acc[TM][TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + KB - 1) / KB;
#pragma unroll
    for (int s0 = 0; s0 < DIST; ++s0) stage(s0, s0);
    int buf = 0, fill = DIST;                                   // fill: the buffer stage kt + DIST goes to
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most (DIST - 1) * 2 * NI of this wave's DMA instructions are still in flight
        constexpr int INFLIGHT = (DIST - 1) * (NIA + NIB);
        static_assert(INFLIGHT == 0 || INFLIGHT == 6 || INFLIGHT == 8 || INFLIGHT == 12 || INFLIGHT == 16 || INFLIGHT == 18, "add the literal");
        if constexpr (INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (INFLIGHT == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if defined(ALM_GEMM_WHATIF) && ALM_GEMM_WHATIF == 8        // MFMA + fragment reads + barriers, no DMA in the loop
        if (kt < 0)
#endif
        stage(kt + DIST, fill);
        fill = fill + 1 == NST ? 0 : fill + 1;
        const unsigned char* sb = smem + buf * STAGE;
        buf = buf + 1 == NST ? 0 : buf + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TM], b[TNB];
            const unsigned co = (((unsigned)(ks * 2 + lh)) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sb + fragA[i] + co);
#pragma unroll
            for (int j = 0; j < TNB; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + fragB[j] + co);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the (zero-filling) prefetches beyond K target the buffers the epilogue reuses
    __syncthreads();
    static_assert(NW * 32 * (32 * TNB * (OUT_F32 ? 4 : 2)) <= NST * STAGE, "epilogue slab");
    gemm_epilogue<BM, BN, WM, WN, TM, TNB, OUT_F32>(p, acc, smem, coff0, m0, n0, wave, wr, wc, lane, lr, lh);
#endif
}

// ---- balanced-split second stage: C tile = sum of its workspace slots [tile_first[t], tile_first[t+1]) (in K order: deterministic).
// grid (tiles, BM / 16): a block sums 16 rows of one tile, 64 threads (float4 each) per row.
template <int BM, int BN>
__global__ __launch_bounds__(256) void stream_reduce_kernel(const float* __restrict__ ws, const int* __restrict__ tile_first, float* __restrict__ C,
                                                            long long ldc, long long sC, int M, int N, int tiles_m, int tiles_n, int accumulate) {
    static_assert(BN == 256, "64 float4 lanes per row");
    const int tile = blockIdx.x;
    const int per = tiles_m * tiles_n, r = tile % per, zb = tile / per;
    int tm, tn;
    if (tiles_m >= tiles_n) { tm = r / tiles_n; tn = r % tiles_n; } else { tn = r / tiles_m; tm = r % tiles_m; }
    const int s0 = tile_first[tile], s1 = tile_first[tile + 1];
    const int c4 = (threadIdx.x & 63) * 4, rsub = threadIdx.x >> 6;
    const int n = tn * BN + c4;
    float* Cz = C + (long long)zb * sC;
    const bool vec = (ldc & 3) == 0 && ((uintptr_t)Cz & 15) == 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = blockIdx.y * 16 + it * 4 + rsub, m = tm * BM + row;
        if (m >= M || n >= N) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sl = s0; sl < s1; ++sl) {
            const float4 v = *reinterpret_cast<const float4*>(ws + ((long long)sl * BM + row) * BN + c4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float* cp = Cz + (long long)m * ldc + n;
        if (vec && n + 3 < N) {
            if (accumulate) { const float4 w = *reinterpret_cast<const float4*>(cp); a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w; }
            *reinterpret_cast<float4*>(cp) = a;
        } else {
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (n + c < N) cp[c] = accumulate ? cp[c] + av[c] : av[c];
        }
    }
}

// ---- split-K second stage: C[b][m][n] (+)= sum_z ws[z][b][m][n]   (blockIdx.y = b)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long mn, long long slice_stride, int N,
                                                            float* __restrict__ C, long long ldc, long long sC, int accumulate) {
    ws += (long long)blockIdx.y * mn;
    C += (long long)blockIdx.y * sC;
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

// ---- launch plumbing ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool TNMODE, bool OUT_F32, int STAGES = 2, bool PIPE = false>
int launch_cfg(const GemmParams& p, int ny, int nz, hipStream_t st) {
    constexpr int smem = STAGES * (BM + BN) * BK * 2;
    static bool attr_done = false;                 // idempotent; a benign race sets the same value twice
    auto kfn = gemm_kernel<BM, BN, WM, WN, TNMODE, OUT_F32, STAGES, PIPE>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    if (p.raster == 1) {
        const int tmaj = tiles_m >= tiles_n ? tiles_m : tiles_n, Q = tiles_m >= tiles_n ? tiles_n : tiles_m;
        const int PL = (tmaj * ny + 7) / 8;
        hipLaunchKernelGGL(kfn, dim3(8 * PL * Q * nz), dim3(WM * WN * 64), smem, st, p);
        return 0;
    }
    hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n, ny, nz), dim3(WM * WN * 64), smem, st, p);
    return 0;
}

// tile: 0 = auto, 1 = 128x128 (4 waves, 2 blocks / CU), 2 = 256x256 (8 waves, 1 block / CU), 3 = 256x128 with a 3-stage DMA ring (8 waves)
int pick_tile(int M, int N, int ny, int tile) {
    if (tile >= 1 && tile <= 13) return tile;
    if (M < 256 || N < 256) return 1;
    const long long big = (long long)((M + 255) / 256) * ((N + 255) / 256) * ny;
    return big >= 192 ? 2 : 1;              // enough 256^2 tiles to occupy most of the 256 CUs
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
    if (p.raster == 1) {
        const int tmaj = tiles_m >= tiles_n ? tiles_m : tiles_n, Q = tiles_m >= tiles_n ? tiles_n : tiles_m;
        const int PL = (tmaj * ny + 7) / 8;
        hipLaunchKernelGGL(kfn, dim3(8 * PL * Q * nz), dim3(512), smem, st, p);
        return 0;
    }
    hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n, ny, nz), dim3(512), smem, st, p);
    return 0;
}

template <bool OUT_F32>
int launch_persist(const GemmParams& p, int ny, hipStream_t st) {
    constexpr int smem = 2 * (256 + 256) * BK * 2;
    static bool attr_done = false;
    auto kfn = gemm_nt_persist_kernel<OUT_F32>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    hipLaunchKernelGGL(kfn, dim3(tiles < 256 ? tiles : 256, ny), dim3(512), smem, st, p);
    return 0;
}

template <bool OUT_F32, bool PIPE>
int launch_bdirect(const GemmParams& p, int ny, hipStream_t st) {
    constexpr int smem = 2 * 256 * BK * 2;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    hipLaunchKernelGGL((gemm_nt_bdirect_kernel<OUT_F32, PIPE>), dim3(tiles, ny), dim3(512), smem, st, p);
    return 0;
}

template <bool OUT_F32, int NST, int BNX, int WNX>
int launch_ring(const GemmParams& p, int ny, hipStream_t st) {
    constexpr int smem = NST * (256 + BNX) * 32 * 2;
    static bool attr_done = false;
    void (*kfn)(GemmParams) = &gemm_nt_ring_kernel<OUT_F32, NST, BNX, WNX>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + BNX - 1) / BNX);
    hipLaunchKernelGGL((gemm_nt_ring_kernel<OUT_F32, NST, BNX, WNX>), dim3(tiles, ny), dim3(2 * WNX * 64), smem, st, p);
    return 0;
}

template <bool TNMODE>
int launch_gemm(const GemmParams& p, int ny, int nz, int out_f32, int tile, hipStream_t st) {
    int tl = pick_tile(p.M, p.N, ny * nz, tile);
    static const int big_tile = [] { const char* e = getenv("ALM_GEMM_BIG_TILE"); return e ? atoi(e) : 2; }();
    if (tl == 2 && (!TNMODE || big_tile == 13 || big_tile == 11) && !p.units) tl = big_tile;     // ALM_GEMM_BIG_TILE=13 / 7: every 256 x 256 launch takes that loop (A/B hook)
    if (tl == 13) {                                // staggered wave rows (NT and TN, split-K, XCD-panel raster; not the balanced-split units)
        if (p.units) return ALM_ERR_UNSUPPORTED;
        return out_f32 ? launch_stag<TNMODE, true>(p, ny, nz, st) : launch_stag<TNMODE, false>(p, ny, nz, st);
    }
    if (tl == 4) {
        if (TNMODE || nz != 1 || p.ksplit > 0 || p.raster != 0) return ALM_ERR_UNSUPPORTED;
        return out_f32 ? launch_persist<true>(p, ny, st) : launch_persist<false>(p, ny, st);
    }
    if (tl == 10 || tl == 12) {                    // deep DMA ring, 32-deep K-steps (NT only): 10 = 256 x 256 on 8 waves, 4 stages;
                                                   // 12 = 256 x 128 on 4 waves, 3 stages, TWO workgroups per CU
        if (TNMODE || nz != 1 || p.ksplit > 0 || p.raster != 0 || p.units) return ALM_ERR_UNSUPPORTED;
        if (tl == 10) return out_f32 ? launch_ring<true, 4, 256, 4>(p, ny, st) : launch_ring<false, 4, 256, 4>(p, ny, st);
        return out_f32 ? launch_ring<true, 3, 128, 2>(p, ny, st) : launch_ring<false, 3, 128, 2>(p, ny, st);
    }
    if (tl == 11) {                                // 384 x 256 on 8 waves (wave tile 192 x 64): 17 % fewer L2 -> LDS bytes per flop than 256 x 256
        if (p.units) return ALM_ERR_UNSUPPORTED;
        return out_f32 ? launch_cfg<384, 256, 2, 4, TNMODE, true>(p, ny, nz, st) : launch_cfg<384, 256, 2, 4, TNMODE, false>(p, ny, nz, st);
    }
    if (tl == 8 || tl == 9) {                      // B operand streamed into registers (NT only): 8 = plain loop, 9 = pipelined A fragment reads
        if (TNMODE || nz != 1 || p.ksplit > 0 || p.raster != 0 || p.units) return ALM_ERR_UNSUPPORTED;
        if (tl == 8) return out_f32 ? launch_bdirect<true, false>(p, ny, st) : launch_bdirect<false, false>(p, ny, st);
        return out_f32 ? launch_bdirect<true, true>(p, ny, st) : launch_bdirect<false, true>(p, ny, st);
    }
    if (tl == 5) return ALM_ERR_UNSUPPORTED;       // 4 waves x (128 x 128) with the plain loop: measured 10-20 % slower than tile 2, superseded by tile 6
    if (tl == 6 || tl == 7) {                      // hand software-pipelined main loop (NT only): 6 = 4 waves x (128 x 128), 7 = 8 waves x (128 x 64)
        if constexpr (TNMODE) return ALM_ERR_UNSUPPORTED;
        else {
            if (tl == 6) return out_f32 ? launch_cfg<256, 256, 2, 2, false, true, 2, true>(p, ny, nz, st) : launch_cfg<256, 256, 2, 2, false, false, 2, true>(p, ny, nz, st);
            return out_f32 ? launch_cfg<256, 256, 2, 4, false, true, 2, true>(p, ny, nz, st) : launch_cfg<256, 256, 2, 4, false, false, 2, true>(p, ny, nz, st);
        }
    }
    if (tl == 3) return out_f32 ? launch_cfg<256, 128, 4, 2, TNMODE, true, 3>(p, ny, nz, st) : launch_cfg<256, 128, 4, 2, TNMODE, false, 3>(p, ny, nz, st);
    if (tl == 2) return out_f32 ? launch_cfg<256, 256, 2, 4, TNMODE, true>(p, ny, nz, st) : launch_cfg<256, 256, 2, 4, TNMODE, false>(p, ny, nz, st);
    return out_f32 ? launch_cfg<128, 128, 2, 2, TNMODE, true>(p, ny, nz, st) : launch_cfg<128, 128, 2, 2, TNMODE, false>(p, ny, nz, st);
}

bool view_too_big(long long rows, long long ld) { return rows * ld * 2 >= 0x7fffffffLL; }

// Split-K plan for `nb` same-shape problems: choose (tile, slices) minimising a simple time model --
//   block waves over the chip x K-steps per block x measured time per K-step  +  workspace round trip through HBM.
struct SplitPlan { int tile, slices; bool stream; };

// ---- balanced split ("stream-K" without fix-up waves) for long-K contractions with too few output tiles to fill the chip ------------------
// The tiles x K-steps iteration space is cut into one contiguous, equally long range per CU: XCD x owns the x-th eighth (its tiles share
// operand panels through that XCD's L2), each of its 32 CUs one range.  A range that crosses a tile boundary is cut there into pieces;
// the first piece of every range is dispatched first, the remaining pieces follow in DECREASING length -- the CU whose first piece was
// shortest frees up first and takes the longest remaining piece (its complement), so every CU ends up with the same number of K-steps:
// no partial last wave, which is what costs the uniform split (e.g. dW1: 88 tiles x 5 slices = 440 workgroups on 256 CUs = 1.72 waves).
// Workgroup b = j * 8 + x is the j-th unit of XCD x (workgroups are dispatched round-robin over the XCDs).  Every piece writes a whole
// fp32 partial tile into its own workspace slot; slots of a tile are consecutive and in K order (stream_reduce_kernel: deterministic).
struct StreamPlan { int nunits = 0, nslots = 0; int* d_units = nullptr; int* d_tile_first = nullptr; };
std::map<std::array<long long, 4>, StreamPlan> g_stream_plans;

struct Piece { int tile, kbeg, ksteps, slot; };

int stream_plan_counts(int M, int N, int K, int nb, int* nslots_out) {
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256) * nb, S = (K + BK - 1) / BK;
    const long long total = (long long)tiles * S;
    const long long per_xcd = (total + 7) / 8;
    int nslots = 0, maxlen = 0;
    for (int x = 0; x < 8; ++x) {
        const long long lo = x * per_xcd, hi = std::min(total, lo + per_xcd);
        if (lo >= hi) continue;
        const long long L = (hi - lo + 31) / 32;
        int len = 0;
        for (long long a = lo; a < hi; a += L) {
            const long long b = std::min(hi, a + L);
            len += (int)((b - 1) / S - a / S) + 1;
        }
        nslots += len;
        maxlen = std::max(maxlen, len);
    }
    if (nslots_out) *nslots_out = nslots;
    return maxlen * 8;
}

const StreamPlan* get_stream_plan(int M, int N, int K, int nb) {
    const std::array<long long, 4> key{M, N, K, nb};
    auto it = g_stream_plans.find(key);
    if (it != g_stream_plans.end()) return &it->second;
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256) * nb, S = (K + BK - 1) / BK;
    const long long total = (long long)tiles * S;
    const long long per_xcd = (total + 7) / 8;
    std::vector<Piece> seq[8];
    std::vector<Piece*> all;
    for (int x = 0; x < 8; ++x) {
        const long long lo = x * per_xcd, hi = std::min(total, lo + per_xcd);
        if (lo >= hi) continue;
        const long long L = (hi - lo + 31) / 32;
        std::vector<Piece> first, rest;
        for (long long a = lo; a < hi; a += L) {
            const long long b = std::min(hi, a + L);
            bool head = true;
            for (long long c = a; c < b;) {
                const long long tile = c / S, e = std::min(b, (tile + 1) * S);
                (head ? first : rest).push_back(Piece{(int)tile, (int)(c - tile * S), (int)(e - c), 0});
                head = false;
                c = e;
            }
        }
        std::stable_sort(rest.begin(), rest.end(), [](const Piece& a, const Piece& b) { return a.ksteps > b.ksteps; });
        seq[x] = first;
        seq[x].insert(seq[x].end(), rest.begin(), rest.end());
    }
    for (int x = 0; x < 8; ++x)
        for (auto& pc : seq[x]) all.push_back(&pc);
    std::stable_sort(all.begin(), all.end(), [](const Piece* a, const Piece* b) { return a->tile != b->tile ? a->tile < b->tile : a->kbeg < b->kbeg; });
    std::vector<int> tile_first(tiles + 1, 0);
    for (size_t i = 0; i < all.size(); ++i) {
        all[i]->slot = (int)i;
        tile_first[all[i]->tile + 1] = (int)i + 1;
    }
    for (int t = 1; t <= tiles; ++t) tile_first[t] = std::max(tile_first[t], tile_first[t - 1]);
    size_t maxlen = 0;
    for (int x = 0; x < 8; ++x) maxlen = std::max(maxlen, seq[x].size());
    std::vector<int> units(maxlen * 8 * 4, -1);
    for (int x = 0; x < 8; ++x)
        for (size_t j = 0; j < seq[x].size(); ++j) {
            int* u = &units[(j * 8 + x) * 4];
            u[0] = seq[x][j].tile; u[1] = seq[x][j].kbeg; u[2] = seq[x][j].ksteps; u[3] = seq[x][j].slot;
        }
    StreamPlan pl;
    pl.nunits = (int)maxlen * 8;
    pl.nslots = (int)all.size();
    if (hipMalloc(&pl.d_units, units.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMalloc(&pl.d_tile_first, tile_first.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(pl.d_units, units.data(), units.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (hipMemcpy(pl.d_tile_first, tile_first.data(), tile_first.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return &(g_stream_plans[key] = pl);
}
int g_dbg_tile = 0, g_dbg_slices = 0, g_dbg_raster = 1;     // tuning hook (almlab_debug_splitk): 0 = automatic
int g_dbg_stream = 0;                                        // balanced split: 0 never (default), 1 by the cost model, 2 always (when applicable)
// NEGATIVE RESULT (MI355X, dW1 5460 x 1024 x 16384: 310 us vs 294 us uniform 5-slice split; dW2 198 vs 160 us): the balanced split removes
// the partial last wave but the 32 workgroups of an XCD then walk DIFFERENT K ranges of their tiles at any moment, so operand panels are no
// longer shared through the L2 (256 workgroups x 2 x 2.9 MB = 1.5 GB of operand traffic per GEMM, ~4.8 TB/s: traffic-bound), whereas the
// uniform split + XCD-panel rasterisation fetches each (panel, slice) once per XCD.  Kept selectable (almlab_debug_stream) and tested.

// XCD-panel rasterisation pays only when the panels spread evenly over the 8 XCDs (measured: 22 panels +3 %, 11 panels -40 %)
int pick_raster(int M, int N, int nb, int tile) {
    if (!g_dbg_raster) return 0;
    const int bm = tile >= 2 ? 256 : 128, bn = tile == 2 ? 256 : 128;
    const int tmm = (M + bm - 1) / bm, tnn = (N + bn - 1) / bn;
    const int P = (tmm >= tnn ? tmm : tnn) * nb;
    return (P % 8 == 0 || P >= 20) ? 1 : 0;
}
SplitPlan splitk_plan(int M, int N, int K, int nb) {
    if (g_dbg_tile > 0 && g_dbg_slices > 0) {
        const int kc = ((K + g_dbg_slices - 1) / g_dbg_slices + BK - 1) / BK * BK;
        return SplitPlan{(g_dbg_tile >= 2 && M >= 256 && N >= 256) ? g_dbg_tile : 1, (K + kc - 1) / kc, false};
    }
    SplitPlan best{1, 1, false};
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
            if (t < best_t) { best_t = t; best = SplitPlan{tile, s, false}; }
        }
    }
    if (g_dbg_stream != 0 && M >= 256 && N >= 256 && K >= 1024) {
        // balanced split over the 256 CUs (256 x 256 tiles): every CU runs ceil(total K-steps / 256) steps; partial tiles go through HBM
        const double tiles = (double)((M + 255) / 256) * ((N + 255) / 256) * nb, S = (K + BK - 1) / BK;
        int nslots = 0;
        stream_plan_counts(M, N, K, nb, &nslots);
        const double t = ceil(tiles * S / 256.0) * 2.0 + 6.0 + (double)nslots * 65536.0 * 8.0 / 4.0e6 + 3.0;
        if ((tiles >= 16 && t < best_t) || g_dbg_stream == 2) best = SplitPlan{2, 0, true};
    }
    return best;
}

}  // namespace

extern "C" int almlab_gemm_bf16_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                long long ldb, long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                                long long sB2, long long sC1, long long sC2, float alpha, int out_f32, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (((sA1 | sA2 | sB1 | sB2) & 7) != 0) return ALM_ERR_BAD_ARG;
    if (view_too_big(256, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate, 0, 0, 0, 1};
    int rc = launch_gemm<false>(p, nb1 * nb2, 1, out_f32, 0, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

/* tile-selectable form of the above without batching (benchmarks / tuning): tile 0 = auto, 1 = 128x128, 2 = 256x256 */
extern "C" int almlab_gemm_bf16_nt_tile(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                     long long ldb, long long ldc, float alpha, int out_f32, int accumulate, int tile, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (view_too_big(256, lda) || view_too_big(256, ldb)) return ALM_ERR_UNSUPPORTED;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, 1, 0, 0, 0, 0, 0, 0, alpha, accumulate, 0, 0, 0, 1};
    int rc = launch_gemm<false>(p, 1, 1, out_f32, tile, (hipStream_t)stream);
    if (rc) return rc;
    ALM_LAUNCH_CHECK();
    return 0;
}

// Split-K for long-K / few-tile contractions (weight gradients: K = B*N tokens), `nb` same-shape problems per launch (element
// strides sA / sB / sC between them).  ws: fp32 workspace of almlab_gemm_splitk_slices(M, N, K, nb) * nb * M * N floats (unused when
// that is 1).  Deterministic (no atomics): the slices are reduced in a fixed order by a second kernel.
// tuning hook for benchmarks (process-global, not thread-safe; 0 = automatic): force the split-K tile (1 = 128x128, 2 = 256x256) and
// slice count, and select the rasterisation of split-K launches (0 = plain grid, 1 = XCD-panel)
extern "C" int almlab_debug_splitk(int tile, int slices, int raster) {
    g_dbg_tile = tile;
    g_dbg_slices = slices;
    g_dbg_raster = raster ? 1 : 0;
    return 0;
}

extern "C" int almlab_debug_stream(int mode) {          // balanced split: 0 never (default), 1 by the cost model, 2 whenever applicable
    g_dbg_stream = mode;
    return 0;
}

extern "C" int almlab_gemm_splitk_slices(int M, int N, int K, int nb) { return splitk_plan(M, N, K, nb < 1 ? 1 : nb).slices; }

// fp32 workspace floats alm_gemm_bf16_{nt,tn}_splitk need for this problem (0: none)
extern "C" int almlab_gemm_splitk_ws_floats(int M, int N, int K, int nb) {
    nb = nb < 1 ? 1 : nb;
    const SplitPlan pl = splitk_plan(M, N, K, nb);
    long long fl = 0;
    if (pl.stream) {
        int nslots = 0;
        stream_plan_counts(M, N, K, nb, &nslots);
        fl = (long long)nslots * 65536;
    } else if (pl.slices > 1) {
        fl = (long long)pl.slices * nb * M * N;
    }
    return fl > 0x7fffffffLL ? -1 : (int)fl;
}

static int splitk_common(bool tn, const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                         long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate, hipStream_t st) {
    const SplitPlan pl = splitk_plan(M, N, K, nb);
    if (pl.stream) {
        if (!ws) return ALM_ERR_BAD_ARG;
        const StreamPlan* sp = get_stream_plan(M, N, K, nb);
        if (!sp) return ALM_ERR_UNSUPPORTED;
        GemmParams p{(const bf16_t*)A, (const bf16_t*)B, ws, nullptr, M, N, K, lda, ldb, 256, nb, 0, sA, 0, sB, 0, 0, alpha, 0, 0, 0, 0, 1, sp->d_units};
        static bool attr_done[2] = {false, false};
        auto kfn = tn ? gemm_kernel<256, 256, 2, 4, true, true> : gemm_kernel<256, 256, 2, 4, false, true>;
        constexpr int smem = 2 * (256 + 256) * BK * 2;
        if (!attr_done[tn]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
            attr_done[tn] = true;
        }
        hipLaunchKernelGGL(kfn, dim3(sp->nunits), dim3(512), smem, st, p);
        const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
        hipLaunchKernelGGL((stream_reduce_kernel<256, 256>), dim3(tiles_m * tiles_n * nb, 16), dim3(256), 0, st, (const float*)ws, sp->d_tile_first, C, ldc,
                           sC, M, N, tiles_m, tiles_n, accumulate);
        return 0;
    }
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

extern "C" int almlab_gemm_bf16_nt_splitk(const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
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
extern "C" int almlab_gemm_bf16_tn_splitk(const void* At, const void* Bt, float* C, float* ws, int M, int N, int K, long long lda,
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
