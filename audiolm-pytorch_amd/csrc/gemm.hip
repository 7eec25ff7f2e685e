// bf16 MFMA GEMM for gfx950:  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T (+ bias[N])      ("NT": both operands K-contiguous)
//
// This is the dense-contraction workhorse of the token-transformer hot path: to_q / to_kv / to_out, the two FFN
// projections, the logit heads, and -- with explicitly transposed operands -- every dgrad / wgrad of those.
// Replaces aten::mm / addmm / bmm at reference audiolm_pytorch.py:255-259, :351, :395, :719, :961, :972.
//
// Design (wave64 / CDNA4, not a warp-32 tiling):
//   * 128x128x64 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64 output tile
//     = 2x2 MFMA 32x32x16 bf16 blocks -> 64 fp32 accumulator registers / lane.
//   * operands are staged global -> registers -> LDS (16-B loads, 8 lanes cover one 128-B row segment), LDS is
//     double buffered (2 x 32 KiB) so one __syncthreads per K-tile; the next tile's global loads are issued before
//     the MFMA block of the current tile and written to the other buffer afterwards.
//   * LDS rows are 128 B (64 bf16, no padding) with the 16-B chunk index XOR-swizzled by ((row >> 1) & 7): the
//     ds_read_b128 of an MFMA fragment (32 rows x one chunk per half-wave) then touches 16 distinct 16-B slots per
//     16-lane service group (conflict-free), and ds_write_b128 of one row (8 lanes) is conflict-free too.
//   * XCD-aware block remap + grouped (GROUP_M = 8) rasterisation so that the ~64 blocks resident on one XCD share
//     A / B panels in that XCD's 4 MiB L2.
//   * two-level batch (blockIdx.y -> (z1, z2)) with element strides: used for the per-quantizer logit heads
//     (einsum 'q c d, b n q d -> b n q c').
// Requirements: K % 8 == 0, lda % 8 == 0, ldb % 8 == 0, A/B 16-byte aligned (vector loads); M, N, ldc arbitrary.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GROUP_M = 8;

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
    int ksplit;        // > 0: split-K mode -- blockIdx.y is the K-slice index, slice z covers k in [z*ksplit, min(K, (z+1)*ksplit)) and
                       // writes its fp32 partial tile to C + z * sC2 (reduced afterwards by splitk_reduce_kernel)
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row * 64) + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][BM * BK];   // [buffer][A|B][row*64 + swizzled chunk]

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int per_group = GROUP_M * tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z1 = blockIdx.y / p.nb2, z2 = blockIdx.y % p.nb2;
    const bf16_t* __restrict__ A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const bf16_t* __restrict__ B = p.B + z1 * p.sB1 + z2 * p.sB2;
    int Kend = p.K;
    if (p.ksplit > 0) {                     // sA2 == sB2 == ksplit: the slice starts ksplit elements further along every row
        Kend = min(p.K - z2 * p.ksplit, p.ksplit);
    }

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // staging assignment: 4 x (row, chunk) per operand
    const int srow = t >> 3, schunk = t & 7;

    uint4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
        const int kk = k0 + schunk * 8;
        const bool kok = kk < Kend;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int gm = m0 + row, gn = n0 + row;
            ra[i] = (kok && gm < p.M) ? *reinterpret_cast<const uint4*>(A + (long long)gm * p.lda + kk) : make_uint4(0, 0, 0, 0);
            rb[i] = (kok && gn < p.N) ? *reinterpret_cast<const uint4*>(B + (long long)gn * p.ldb + kk) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            *reinterpret_cast<uint4*>(&lds[buf][0][swz(row, schunk)]) = ra[i];
            *reinterpret_cast<uint4*>(&lds[buf][1][swz(row, schunk)]) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (Kend + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const bf16_t* As = lds[buf][0];
        const bf16_t* Bs = lds[buf][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(&As[swz(wr * 64 + i * 32 + lr, ks * 2 + lh)]);
                b[i] = *reinterpret_cast<const bf16x8*>(&Bs[swz(wc * 64 + i * 32 + lr, ks * 2 + lh)]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const long long coff = z1 * p.sC1 + z2 * p.sC2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wc * 64 + j * 32 + lr;
            if (gn >= p.N) continue;
            const float bv = p.bias ? p.bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (gm >= p.M) continue;
                float v = acc[i][j][r] * p.alpha + bv;
                const long long idx = coff + (long long)gm * p.ldc + gn;
                if (OUT_F32) {
                    float* C = reinterpret_cast<float*>(p.C);
                    if (p.accumulate) v += C[idx];
                    C[idx] = v;
                } else {
                    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
                    if (p.accumulate) v += bf2f(C[idx]);
                    C[idx] = f2bf(v);
                }
            }
        }
}

// ---- split-K second stage: C[m][n] (+)= sum_z ws[z][m][n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long mn, int N, float* __restrict__ C,
                                                            long long ldc, int accumulate) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < mn; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += ws[(long long)z * mn + i];
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

}  // namespace

extern "C" int alm_gemm_bf16_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                long long ldb, long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                                long long sB2, long long sC1, long long sC2, float alpha, int out_f32, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    if (((sA1 | sA2 | sB1 | sB2) & 7) != 0) return ALM_ERR_BAD_ARG;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, lda, ldb, ldc, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate, 0};
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    dim3 grid(tiles, nb1 * nb2);
    if (out_f32)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    ALM_LAUNCH_CHECK();
    return 0;
}

// Split-K variant for long-K / few-tile contractions (weight gradients: K = B*N tokens).  ws: fp32 workspace of
// alm_gemm_splitk_slices(M, N, K) * M * N floats.  Deterministic (no atomics): slices are reduced in a fixed order.
extern "C" int alm_gemm_splitk_slices(int M, int N, int K) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int s = (1024 + tiles - 1) / tiles;                 // aim at ~2 resident blocks on each of the 256 CUs, twice over
    const int maxs = (K + 511) / 512;                   // keep >= 512 of K per slice
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

extern "C" int alm_gemm_bf16_nt_splitk(const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                                       long long ldc, float alpha, int accumulate, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ALM_ERR_BAD_ARG;
    const int splits = alm_gemm_splitk_slices(M, N, K);
    int kc = (K + splits - 1) / splits;
    kc = (kc + BK - 1) / BK * BK;
    const int nsl = (K + kc - 1) / kc;
    GemmParams p{(const bf16_t*)A, (const bf16_t*)B, ws, nullptr, M, N, K, lda, ldb, (long long)N, nsl, 0, kc, 0, kc, 0, (long long)M * N, alpha, 0, kc};
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_nt_kernel<true>, dim3(tiles, nsl), dim3(256), 0, (hipStream_t)stream, p);
    const long long mn = (long long)M * N;
    const int grid = (int)((mn + 255) / 256 < 4096 ? (mn + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nsl, mn, N, C, ldc, accumulate);
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

// batched form used inside the attention launchers: dst[z][c][r] = src[z][r][c]
int alm_transpose_bf16_batched_internal(const void* src, void* dst, int rows, int cols, long long ld_src, long long ld_dst, int rows_pad,
                                        int batch, long long bs_src, long long bs_dst, hipStream_t stream) {
    dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64, batch);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_src, ld_dst, rows_pad,
                       bs_src, bs_dst);
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
