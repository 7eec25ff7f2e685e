// Causal multi-query flash attention, forward + backward, for gfx950 (MI355X).
//
// Replaces reference attend.py:98-146 (einsum -> mask fills -> softmax -> einsum) / attend.py:69-96 (SDPA) as called from
// audiolm_pytorch.py:381-394: q (b, n, h*64), ONE shared k / v head of width 64 (MQA, audiolm_pytorch.py:293-294),
// key-padding mask (b, n) bool, causal, scale = 64^-0.5.  The (b, h, n, n) score / probability / bool-mask tensors the
// reference materialises never exist here: online softmax, fp32 statistics, bf16 MFMA operands.
//
// Mapping (wave64, MFMA 32x32x16 bf16):
//   * forward / dQ: workgroup = one query block of one batch element x 4 heads, ONE WAVE PER HEAD; dQ: 64 queries, a 64 x 64 (queries x keys)
//     score tile per wave and step; forward: 32 queries (32 x 64 tile), and a workgroup takes TWO blocks (idx, last - idx) one after the other so
//     that every workgroup of the launch walks the same number of key tiles (see mqa_fwd_kernel).  The 64-key K / V tile is fetched ONCE per
//     workgroup by DMA (`buffer_load ... lds`, double buffered, one barrier per tile) and shared by the heads -- MQA makes K / V traffic ~free.
//   * scores are computed transposed, S^T = K Q^T, so that one lane owns one query column: softmax statistics are lane-local
//     (one cross-half exchange), and the fp32 S^T accumulator registers ARE the B operand (P^T) of the second MFMA
//     (O^T = V^T P^T) after an in-register bf16 pack: no LDS round trip, no permutes.  The key-index permutation this implies
//     (a lane half holds keys {4h..4h+3} u {8+4h..8+4h+3} of each 16-key step) is applied to the V^T fragment reads.
//   * the key-padding mask costs nothing: the S^T accumulators are INITIALISED with a per-key bias (0 / -inf) instead of zero.
//   * every LDS tile is a [64 rows][64 dims] image with 128-B rows whose 16-B chunks are XOR-swizzled by
//     f(row) = bit1<<2 | bit3<<1 | bit2, chosen so that BOTH access patterns are bank-conflict free: row fragments
//     (ds_read_b128, lane = row) and transposed fragments (ds_read_b64_tr_b16, lane = dim, 4 consecutive rows per read).
//     One image therefore serves K (S^T) and K^T (dQ), V and V^T, Q and Q^T, dO and dO^T.
//   * deferred rescaling: the running maximum is only raised (and O rescaled) when a tile's maximum exceeds it by 2^8.
//   * dK/dV: workgroup = 64 keys x 4 heads (8 waves = 4 heads x 2 key halves), loops over the 64-query tiles at/after the
//     diagonal (Q / dO tiles by DMA, double buffered); per-head dK^T / dV^T accumulators are summed over the 4 heads through
//     LDS; the two head groups write separate fp32 partials that alm_kv_grad_pack adds (deterministic, no atomics).
//   * heavy (long-causal-span) workgroups are scheduled first and (dQ) paired with light ones on a CU; the DMA destination and the image being read
//     are distinct `__restrict__` parameters of one inlined tile step in all three kernels (no compiler-inserted DMA wait before transposed reads).
//   * STRUCTURED ATTENTION BIAS (the `flash_attn=False` models: reference RelativePositionBias audiolm_pytorch.py:202-242, the Coarse
//     cross-attention override :924-936 and the Fine (frame, quantizer) table :1227-1298).  The reference gathers a (h, n, n) fp32
//     tensor from a small per-head table and adds it to the scores; here the table is indexed INSIDE the kernels:
//         bias(h, i, j) = (qattr[i] & kattr[j]) ? tbl[h][0] : tbl[h][(qkey4[i] - kkey4[j]) / 4]
//     (tbl in raw-score units, i.e. already divided by `scale`; slot 0 is the "special pair" value: cross_attn_bias / null_pos_bias).
//     The gathered value is one more term of the S accumulators' INITIAL value.  The table gradient (dS summed over every pair that
//     reads a slot) is accumulated by the dQ kernel: per wave, an LDS window covering the slots one (32 q x 64 key) pass can touch
//     takes ds_add_f32 updates and is flushed into that workgroup's own partial table (plain read-modify-write: deterministic); a
//     pass whose window would not fit falls back to global atomics on the same partial.  alm_attn_bias_grad_reduce sums the partials.
#include <type_traits>

#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 8.0f;                 // log2 units
constexpr unsigned OOB = 0x80000000u;
constexpr int HPB = 4;                              // heads per workgroup

typedef __attribute__((address_space(3))) void lds_void;

// chunk swizzle of the 128-B-row LDS images
__device__ __forceinline__ int fsw(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }

// 16-byte LDS loads inside the DMA loops go through EXT-VECTOR types, never through HIP's float4 / int4 structs.  Why (found in the ISA, reproduced
// in a 20-line kernel): after a `buffer_load ... lds` the compiler puts `s_waitcnt vmcnt(0)` in front of the next LDS load made through one of the
// HIP_vector_type structs (an aggregate access it cannot tell apart from the DMA's destination), but not in front of ext_vector or scalar loads.
// A single `*(const float4*)(kbias + ...)` at the top of a tile step therefore waited for the WHOLE DMA of the next tile that had just been issued:
// the prefetch never overlapped the MFMAs in any of the three kernels.  The barrier at the end of every step is what orders DMA writes and reads.
typedef __attribute__((ext_vector_type(4))) float lds_f4v;
typedef __attribute__((ext_vector_type(4))) int lds_i4v;
__device__ __forceinline__ float4 lds_ld_f4(const float* p) {
    const lds_f4v v = *reinterpret_cast<const lds_f4v*>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ int4 lds_ld_i4(const int* p) {
    const lds_i4v v = *reinterpret_cast<const lds_i4v*>(p);
    return make_int4(v[0], v[1], v[2], v[3]);
}

// row fragment: lane (row, lh) gets dims ks*16 + lh*8 .. +7 of `row`
__device__ __forceinline__ bf16x8 nat_frag(const unsigned char* tile, int row, int frow, int ks, int lh) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((((ks << 1) | lh) ^ frow) << 4));
}

// transposed fragment for the 16-row step starting at row `k16`: lane (dim = db*32 + (lane & 31), half lh) gets rows
// k16 + 4*lh + {0..3} and k16 + 8 + 4*lh + {0..3}  -- exactly the key set a lane half holds in the S^T accumulators.
struct TrOff { unsigned o[2][2]; };                 // [db][r] byte offsets within a 16-row step
__device__ __forceinline__ TrOff make_troff(int lane) {
    TrOff t;
    const int g = lane >> 4, s = lane & 15;
    const int rr = 4 * (g >> 1) + (s >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 8 * r + rr;
            const int ch = db * 4 + (g & 1) * 2 + ((s & 3) >> 1);
            t.o[db][r] = (unsigned)(row * 128 + ((ch ^ fsw(row)) << 4) + (s & 1) * 8);
        }
    return t;
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int k16, const TrOff& t, int db) {
    const unsigned char* base = tile + k16 * 128;
    return __builtin_shufflevector(lds_tr16(base + t.o[db][0]), lds_tr16(base + t.o[db][1]), 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
    f32x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = v[8 * s + j];
    return __builtin_bit_cast(bf16x8, __builtin_convertvector(f, hbf16x8));      // 4 x v_cvt_pk_bf16_f32 (round-to-nearest-even)
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
    const bf16_t* dout; const float* ndelta; const float* nlse;
    bf16_t* dq; float* dk; float* dv;
    long long ldq, ldk, ldv, ldo, lddo, lddq, lddk, part_stride;
    int B, N, H, HG;
    int dkv_split;                          // dK/dV kernel: 1, or 2 = SPLIT-Q (see mqa_bwd_dkv_kernel)
    float scale;
    // structured bias (null tbl = none)
    const float* tbl; const int* qkey4; const int* kkey4; const int* qattr; const int* kattr; float* dtbl_part;
    int LT;
    // attention dropout (DROP instantiations only): keep iff hash >= drop_thr (= p * 2^32), kept probabilities x drop_scale = 1 / (1 - p)
    unsigned drop_thr; float drop_scale; unsigned seed_lo, seed_hi;
    const unsigned long long* seed_dev;     // non-NULL: the mask stream is seed + *seed_dev, read at RUN time (a hipGraph replay re-draws: see set_dropout)
};

// ---- attention dropout (reference attend.py:92 `dropout_p` / :140 `attn_dropout(attn)`, training only) ----------------------------------------
// The keep decision of pair (query i, key j) of (batch element b, head h) must be the SAME in the forward, the dQ and the dK/dV kernels, which hold
// the pair in different lanes / registers: it is a stateless function of (seed, b, h, i, j) -- a 32-bit integer finaliser (two multiply-xorshift
// rounds, "lowbias32") over the pair index i * N + j, salted per (seed, b, h).  tests/test_gpu_dropout.py restates it in numpy and hands the
// resulting masks to the oracle.  Dropped probabilities still count in the softmax normalisation (the reference drops AFTER the softmax).
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_salt(const AttnParams& p, int b, int head) {
    unsigned lo = p.seed_lo, hi = p.seed_hi;
    if (p.seed_dev) {                                                    // wave-uniform scalar load; the three kernels of a layer read the same value
        const unsigned long long s = (((unsigned long long)hi << 32) | lo) + *p.seed_dev;
        lo = (unsigned)(s & 0xffffffffu);
        hi = (unsigned)(s >> 32);
    }
    return mix32(lo ^ mix32(hi + (unsigned)(b * p.H + head) * 0x9E3779B9u));
}
__device__ __forceinline__ bool drop_keep(unsigned salt, int qi, int kj, int N, unsigned thr) {
    return mix32(((unsigned)qi * (unsigned)N + (unsigned)kj) ^ salt) >= thr;
}

constexpr int WCAP = 1024;                          // floats per wave in the dQ kernel's table-gradient window (slot 0 = special pairs)

// SP = false: the caller knows (wave-uniformly) that no pair of the tile is special, and skips the attribute test
template <bool SP>
__device__ __forceinline__ float bias_at(const __amdgpu_buffer_rsrc_t& rsT, int kq4, int kk4, int aq, int ak) {
    int voff = kq4 - kk4;                                                // out-of-table offsets read 0 (buffer bounds check)
    if (SP) voff = (aq & ak) ? 0 : voff;
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsT, voff, 0, 0));
}

__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return __builtin_amdgcn_readfirstlane(v);
}

// log-sum-exp -> log2 domain; a fully masked query row (lse = -inf, the forward wrote zeros) gets +inf so that every P is 0
__device__ __forceinline__ float lse_log2(float l) { return l == -INFINITY ? INFINITY : l * LOG2E; }

// row (within a 32-row block) held by accumulator register r of a lane in half lh
__device__ __forceinline__ int drow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// workgroup id -> (light/heavy-paired block index, head group, batch element)
struct BlockId { int blk, hg, b; };
__device__ __forceinline__ BlockId decode_block(int L, int nblk, int HG, int B, bool heavy_is_high, bool pair_on_cu) {
    BlockId r;
    const int combos = HG * B;
    if (pair_on_cu) {
        // Kernels with TWO co-resident workgroups per CU (dQ).  Round 6: one ordering for every block count.  The list of an XCD (all blocks of the
        // combos / 8 (batch element, head group) pairs pinned to it: their K / V tiles are fetched from HBM once and then hit in that XCD's L2; workgroup L runs on
        // XCD L % 8 -- observed dispatch order, a wrong guess only costs speed) is walked in GLOBAL weight order g = 0, 1, ... (heaviest first: longest-
        // processing-time-first list scheduling for everything dispatched after the first wave of residents), except that the first 2 x 32 positions -- the
        // workgroups resident together at the start, position p and p + 32 sharing a CU -- are arranged as complementary (heavy, light) pairs.
        // Until round 6 the list was [combo 0 heavy -> light][combo 1 light -> heavy]: exact for N = 2048 (64 workgroups = 64 slots, every pair sums to 33 key
        // tiles) but with more workgroups than slots the HEAVIEST blocks of every odd combo were dispatched LAST and ran alone -- N = 2049 (33 blocks): dQ + dK/dV
        // 201 -> 172 us with the list fixed (profiles/r6v_odd_blocks_ab.log); at N = 8 253 / 16 385 the tail was 129 / 257 key tiles long.
        int p, ncl, half, xcd;
        if ((combos & 7) == 0) { xcd = L & 7; p = L >> 3; ncl = combos >> 3; half = 32; }
        else { xcd = -1; p = L; ncl = combos; half = 256; }       // (no XCD pinning: positions p and p + 256 share a CU)
        const int W = ncl * nblk, M2 = min(W, 2 * half);
        const int g = (p < half || p >= M2) ? p : M2 - 1 - (p - half);
        const int rank = g / ncl, cl = g - rank * ncl;             // rank 0 = heaviest block of its combo
        const int combo = xcd >= 0 ? cl * 8 + xcd : cl;
        r.hg = combo % HG;
        r.b = combo / HG;
        r.blk = heavy_is_high ? nblk - 1 - rank : rank;
        return r;
    }
    int idx, combo;
    if ((combos & 7) == 0) {
        // XCD-aware: all key / query blocks of one (batch element, head group) go to ONE XCD (see above); each XCD serves combos / 8 groups one after the
        // other, heaviest block first
        const int xcd = L & 7, j = L >> 3;
        const int cl = j / nblk;
        idx = j % nblk;
        combo = cl * 8 + xcd;
    } else {
        idx = L % nblk;
        combo = L / nblk;
    }
    r.hg = combo % HG;
    r.b = combo / HG;
    // one workgroup per CU, round-based execution: plain heaviest-first order within a combo (longest-processing-time-first list scheduling)
    r.blk = heavy_is_high ? nblk - 1 - idx : idx;
    return r;
}

// DMA of one [64 rows][64 dims] tile (rows row0 .. row0 + 63 of a row-major matrix with `ld_bytes` row pitch, dims at byte
// offset col_bytes) into a swizzled LDS image: 8 pieces of 8 rows; this wave issues the NP pieces first, first + step, ...
// STRAIGHT-LINE on purpose (compile-time piece count, no bounds predicate): as a run-time loop with a `row < nrows ? offset : OOB` test the
// compiler emitted an exec-masked branch per piece, and every taken branch stalls the wave on an instruction fetch -- 150 cycles per piece in
// isolation, ~260 in the kernels (8 waves fetching), against 25-35 for the unrolled form (scripts/ubench/dma_issue.hip; the forward kernel spent
// ~1000 of its ~3800 cycles per tile step here).  Rows >= nrows need no test: their offsets lie beyond the buffer resource's num_records
// (callers size it to end inside row nrows - 1), so the hardware bounds check returns zeros for them.
template <int NP>
__device__ __forceinline__ void dma_tile(const __amdgpu_buffer_rsrc_t& rs, unsigned char* img, int first, int step, int lane, int row0, unsigned ld_bytes,
                                         unsigned col_bytes) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = first + i * step;
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ fsw(row);
        const unsigned vo = (unsigned)(row0 + row) * ld_bytes + col_bytes + (unsigned)c * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + piece * 1024), 16, vo, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <bool BIAS> constexpr int FWD_LDS = 2 * 16384 + 512 + (BIAS ? 1024 + 64 : 0);
template <bool BIAS> constexpr int DQ_LDS = 2 * 16384 + 512 + (BIAS ? 1024 + 64 + HPB * WCAP * 4 : 0);

// Bench-only cycle probe (scripts/attn_probe.py builds attention.hip with -DALM_ATTN_PROBE into its own library; the product build has none of this):
// per wave, s_memtime cycles spent in each segment of the forward tile loop.
#ifdef ALM_ATTN_PROBE
__device__ unsigned long long g_attn_probe[4096 * 8];
#define PROBE_DECL unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long plast = __builtin_readcyclecounter();
#define PROBE_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long pt_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); pacc[i] += pt_ - plast; plast = pt_; } while (0)
#define PROBE_FLUSH(slot) do { if (lane == 0 && (slot) < 4096) for (int i_ = 0; i_ < 8; ++i_) g_attn_probe[(slot) * 8 + i_] = pacc[i_]; } while (0)
#else
#define PROBE_DECL
#define PROBE_T(i)
#define PROBE_FLUSH(slot)
#endif
#ifdef ALM_DKV_PROBE                                    // the same for the dK/dV kernel (slot 7 = number of query-tile steps of the workgroup)
__device__ unsigned long long g_attn_probe[4096 * 8];
#define DKV_PROBE_DECL unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long plast = __builtin_readcyclecounter();
#define DKV_PROBE_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long pt_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); pacc[i] += pt_ - plast; plast = pt_; } while (0)
#define DKV_PROBE_FLUSH(slot, steps) do { pacc[7] = (steps); if (lane == 0 && (slot) < 4096) for (int i_ = 0; i_ < 8; ++i_) g_attn_probe[(slot) * 8 + i_] = pacc[i_]; } while (0)
#else
#define DKV_PROBE_DECL
#define DKV_PROBE_T(i)
#define DKV_PROBE_FLUSH(slot, steps)
#endif

// QB = 32-query sub-blocks per wave.  QB == 1 (the launcher's choice): a workgroup takes query blocks idx and nqb - 1 - idx back to back, so every
// workgroup of the launch runs the SAME number of key tiles (nqb / 2 + 1).  With one 64-query block per workgroup and heavy / light blocks paired on
// a CU (QB == 2 + decode_block's pairing, what this kernel did before) the CU that got blocks (31, 0) ran its heavy workgroup ALONE -- one wave per
// SIMD, nothing to overlap the softmax VALU work with the other's MFMAs -- for the whole launch and was the long pole: 74 us, 23 % MFMA-busy, while a
// (16, 15) CU was done long before (scripts/ubench/wg_placement.hip shows the placement).
template <bool BIAS, int QB, bool DROP = false>
__global__ __launch_bounds__(256, 2) void mqa_fwd_kernel(AttnParams p) {
    constexpr int QR = 32 * QB;                                       // query rows per block
    // DYNAMIC LDS on purpose: for a static __shared__ array the compiler knows the object every ds_read touches and, having no alias scopes for the
    // LDS-DMA writes, waits (s_waitcnt vmcnt(0)) for ALL outstanding DMA before each batch of fragment reads -- the next tile's prefetch could never
    // overlap this tile's MFMAs.  With extern __shared__ it leaves the ordering to the barriers below (as in gemm.hip and the dK/dV kernel).
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];       // FWD_LDS<BIAS> bytes
    float* kbias = reinterpret_cast<float*>(smem + 32768);          // [2][64]: 0 for attendable keys, -inf otherwise
    int* kk4s = reinterpret_cast<int*>(smem + 32768 + 512);         // BIAS: [2][64] key-side table offsets, [2][64] key-side attributes
    int* kas = kk4s + 128;
    int* kor = kas + 128;                                           //       [2] OR of the staged tile's key attributes

    const int nqb = (p.N + QR - 1) / QR;
    const BlockId id = decode_block(blockIdx.x, QB == 1 ? (nqb + 1) / 2 : nqb, p.HG, p.B, true, QB != 1);
    const int b = id.b;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int head = id.hg * HPB + wave;
    const bool active = head < p.H;
    const int lr = lane & 31, lh = lane >> 5;
    const float c2 = p.scale * LOG2E;
    const unsigned dsalt = DROP ? drop_salt(p, b, active ? head : 0) : 0u;

    const bf16_t* kbase = p.k + (long long)b * p.N * p.ldk;
    const bf16_t* vbase = p.v + (long long)b * p.N * p.ldv;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0, (int)(((long long)(p.N - 1) * p.ldk + DH) * 2), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0, (int)(((long long)(p.N - 1) * p.ldv + DH) * 2), 0x00020000);
    const uint8_t* mrow = p.mask ? p.mask + (long long)b * p.N : nullptr;

    // The key-mask byte of a tile is fetched ONE TILE AHEAD into a register (wave 0) and written to LDS when the tile's K / V DMA is issued:
    // vector-memory results retire in order, so a load issued after the DMA and needed before the end of the step would make wave 0 wait for the
    // whole DMA it has just started.  The register holds the RAW byte: with the `!= 0` test done at load time the compiler put the test -- and an
    // `s_waitcnt vmcnt(0)` that EVERY wave executes, exec mask or not -- right behind the load, i.e. right behind the DMA issue (seen in the ISA,
    // ~1300 cycles per step in scripts/attn_probe.py).
    struct KeySide { int raw; };
    auto load_side = [&](int tile) {
        KeySide ks{1};
        if (t < 64) {
            const int key = tile * 64 + t;
            ks.raw = key < p.N;
            if (ks.raw && mrow) ks.raw = mrow[key];
        }
        return ks;
    };
    auto stage = [&](unsigned char* __restrict__ img, int tile, int buf, const KeySide& ks) {
        dma_tile<2>(rsK, img, wave, 4, lane, tile * 64, (unsigned)(p.ldk * 2), 0);
        dma_tile<2>(rsV, img + 8192, wave, 4, lane, tile * 64, (unsigned)(p.ldv * 2), 0);
        if (t < 64) {
            int raw = ks.raw;
            asm volatile("" : "+v"(raw));                              // the test stays HERE (one step after the load)
            kbias[buf * 64 + t] = raw ? 0.f : -INFINITY;
            if (BIAS) {                                                // (the biased variants are at their register limit: these stay at stage time)
                const int kc = min(tile * 64 + t, p.N - 1);
                kk4s[buf * 64 + t] = p.kkey4[kc];
                const int ka = p.kattr[kc];
                kas[buf * 64 + t] = ka;
                const int ko = wave_or(ka);
                if (t == 0) kor[buf] = ko;
            }
        }
    };

    PROBE_DECL
    const int npass = (QB == 1 && nqb - 1 - id.blk != id.blk) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    const int qblk = pass ? nqb - 1 - id.blk : id.blk;
    const int q0 = qblk * QR;
    // Q^T fragments (B operand): lane = query column, k = head dim
    bf16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 32 + lr;
        const bf16_t* qp = p.q + ((long long)b * p.N + qi) * p.ldq + head * DH;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = gload8(qp + ks * 16 + lh * 8, active && qi < p.N);
    }

    int kq4[QB], aq[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { kq4[qb] = 0; aq[qb] = 0; }
    __amdgpu_buffer_rsrc_t rsT = rsK;
    if (BIAS) {
        rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.tbl + (long long)(active ? head : 0) * p.LT), 0, p.LT * 4, 0x00020000);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qc = min(q0 + qb * 32 + lr, p.N - 1);
            kq4[qb] = p.qkey4[qc];
            aq[qb] = p.qattr[qc];
        }
    }
    int aq_all = 0;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) aq_all |= aq[qb];
    const int aq_or = BIAS ? wave_or(aq_all) : 0;

    f32x16 o[2][QB];                   // [db][qb]: O^T blocks (rows = head dim, cols = queries)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][j][r] = 0.f;
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m[qb] = -INFINITY; l[qb] = 0.f; }

    const int frow = fsw(lr);
    const TrOff troff = make_troff(lane);
    const int ntiles = (min(q0 + QR, p.N) - 1) / 64 + 1;

    KeySide side = load_side(0);
    stage(smem, 0, 0, side);
    side = load_side(1);
    __syncthreads();

    // One tile step.  The DMA destination (next tile's image) and the image being read are distinct __restrict__ parameters of ONE inlined body:
    // that gives the compiler the alias scopes it needs to let the ds_read_b64_tr_b16 V^T reads go ahead of the in-flight DMA -- without them it
    // put `s_waitcnt vmcnt(0)` in front of the first transposed read of every step (the builtin carries no alias information of its own).
    auto step = [&](unsigned char* __restrict__ nimg, const unsigned char* __restrict__ Kt, int tile) {
        const int buf = tile & 1;
#ifndef ALM_PROBE_NOSTAGE                                             // (probe builds can leave the DMA out: timing only, wrong results)
        if (tile + 1 < ntiles) stage(nimg, tile + 1, buf ^ 1, side);
#endif
        side = load_side(tile + 2);                                   // lands during this step; first needed by the next step's stage()
        PROBE_T(0);
        if (active) {
            const unsigned char* Vt = Kt + 8192;
            const float* kbs = kbias + buf * 64;

            // S^T = K Q^T (+ key bias): [kb][qb] 32x32 blocks
            f32x16 st[2][QB];
            auto init_scores = [&](auto spc) {
                constexpr bool SP = decltype(spc)::value;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bv = lds_ld_f4(kbs + kb * 32 + 8 * g + 4 * lh);
                        const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
                        if (BIAS) {
                            const int4 kk = lds_ld_i4(kk4s + buf * 64 + kb * 32 + 8 * g + 4 * lh);
                            int4 ka = make_int4(0, 0, 0, 0);
                            if (SP) ka = lds_ld_i4(kas + buf * 64 + kb * 32 + 8 * g + 4 * lh);
                            const int kkv[4] = {kk.x, kk.y, kk.z, kk.w}, kav[4] = {ka.x, ka.y, ka.z, ka.w};
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                                for (int c = 0; c < 4; ++c) st[kb][qb][4 * g + c] = bvv[c] + bias_at<SP>(rsT, kq4[qb], kkv[c], aq[qb], kav[c]);
                        } else {
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                                for (int c = 0; c < 4; ++c) st[kb][qb][4 * g + c] = bvv[c];
                        }
                    }
            };
            if (BIAS && (aq_or & __builtin_amdgcn_readfirstlane(kor[buf])) != 0) init_scores(std::true_type{});
            else init_scores(std::false_type{});
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8 kf = nat_frag(Kt, kb * 32 + lr, frow, ks, lh);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) st[kb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], st[kb][qb], 0, 0, 0);
                }
            PROBE_T(1);
            if (tile == ntiles - 1) {                              // diagonal tile: causal mask (key index > query index)
                const int doff = tile * 64 - q0;                   // 0, or -32 for an odd 32-query block
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (doff + kb * 32 + drow(r, lh) > qb * 32 + lr) st[kb][qb][r] = -INFINITY;
            }
            // online softmax, one lane = one query (its partner lane ^ 32 owns the other half of the keys)
            float tm[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float a = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a = fmaxf(a, st[kb][qb][r]);
                a = fmaxf(a, __shfl_xor(a, 32, 64));
                tm[qb] = a * c2;
            }
            bool need = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) need = need || (tm[qb] > m[qb] + RESCALE_THR);
            if (__any(need)) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const float mn = fmaxf(m[qb], tm[qb]);
                    const float alpha = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m[qb] - mn);
                    m[qb] = mn;
                    l[qb] *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][qb][r] *= alpha;
                }
            }
            PROBE_T(2);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float ms = (m[qb] == -INFINITY) ? 0.f : m[qb];
                float ps = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][qb][r], c2, -ms));
                        ps += pv;                                  // the row sum counts dropped pairs too: dropout acts on the softmax OUTPUT
                        st[kb][qb][r] = (!DROP || drop_keep(dsalt, q0 + qb * 32 + lr, tile * 64 + kb * 32 + drow(r, lh), p.N, p.drop_thr)) ? pv : 0.f;
                    }
                l[qb] += ps;
            }
            PROBE_T(3);
            // O^T += V^T P^T
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 pf[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) pf[qb] = pack8(st[kb][qb], s);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 vf = tr_frag(Vt, kb * 32 + s * 16, troff, db);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[db][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], o[db][qb], 0, 0, 0);
                    }
                }
            PROBE_T(4);
        }
    };
#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
        step(smem + ((tile & 1) ^ 1) * 16384, smem + (tile & 1) * 16384, tile);
        __syncthreads();
        PROBE_T(5);
    }

    if (active)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 32 + lr;
        const float lt = l[qb] + __shfl_xor(l[qb], 32, 64);
        const float inv = (lt > 0.f ? 1.f / lt : 0.f) * (DROP ? p.drop_scale : 1.f);
        if (qi < p.N) {
            bf16_t* op = p.o + ((long long)b * p.N + qi) * p.ldo + head * DH;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * lh;
                    *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf2(o[db][qb][4 * g] * inv, o[db][qb][4 * g + 1] * inv),
                                                                   pack_bf2(o[db][qb][4 * g + 2] * inv, o[db][qb][4 * g + 3] * inv));
                }
            if (lh == 0) p.lse[((long long)b * p.H + head) * p.N + qi] = (lt > 0.f) ? (m[qb] / LOG2E + logf(lt)) : -INFINITY;
        }
    }
    PROBE_T(6);
    }
    PROBE_FLUSH(blockIdx.x * 4 + wave);
}

// ------------------------------------------------------------------------------------------------------------------
// backward prologue (one wave per (b, q) row; 8 lanes per head):
//   ndelta[b][h][q] = -sum_d dO * O          nlse[b][h][q] = -lse / scale   (-inf for a fully masked query row)
// Both are accumulator INITIAL VALUES of the backward MFMAs: dP - delta and (S - lse / scale) come out of the matrix core.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, long long ldo, const bf16_t* __restrict__ dout,
                                                         long long lddo, const float* __restrict__ lse, float* __restrict__ ndelta,
                                                         float* __restrict__ nlse, float inv_scale, int B, int N, int H) {
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
        if ((lane & 7) == 0) {
            const long long idx = ((long long)b * H + (c >> 3)) * N + qi;
            ndelta[idx] = -s;
            const float l = lse[idx];
            nlse[idx] = (l == -INFINITY) ? -INFINITY : -l * inv_scale;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward dQ: same decomposition as forward.  dQ^T = scale * K^T dS^T,  dS^T = P^T o (dP^T - delta),  dP^T = V dO^T.
// The two 32-query blocks of a wave are processed one after the other inside a tile (register budget).
// ------------------------------------------------------------------------------------------------------------------
template <bool BIAS, bool DROP = false>
__global__ __launch_bounds__(256, 2) void mqa_bwd_dq_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];       // DQ_LDS<BIAS> bytes; dynamic LDS: see mqa_fwd_kernel
    float* kbias = reinterpret_cast<float*>(smem + 32768);
    int* kk4s = reinterpret_cast<int*>(smem + 32768 + 512);         // BIAS: [2][64] key-side table offsets, [2][64] attributes,
    int* kas = kk4s + 128;                                          //       [2][2] min / max offset of the staged key tile,
    int* kmm = kas + 128;                                           //       (+ OR of its attributes), [HPB][WCAP] per-wave table-gradient windows
    float* wins = reinterpret_cast<float*>(smem + 32768 + 512 + 1024 + 64);

    const int nqb = (p.N + 63) / 64;
    const BlockId id = decode_block(blockIdx.x, nqb, p.HG, p.B, true, true);
    const int qblk = id.blk, b = id.b;
    const int q0 = qblk * 64;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int head = id.hg * HPB + wave;
    const bool active = head < p.H;
    const int lr = lane & 31, lh = lane >> 5;
    const float c2 = p.scale * LOG2E;
    const unsigned dsalt = DROP ? drop_salt(p, b, active ? head : 0) : 0u;

    const bf16_t* kbase = p.k + (long long)b * p.N * p.ldk;
    const bf16_t* vbase = p.v + (long long)b * p.N * p.ldv;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0, (int)(((long long)(p.N - 1) * p.ldk + DH) * 2), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0, (int)(((long long)(p.N - 1) * p.ldv + DH) * 2), 0x00020000);
    const uint8_t* mrow = p.mask ? p.mask + (long long)b * p.N : nullptr;

    struct KeySide { int raw; };                                     // the RAW mask byte, fetched one tile ahead and tested one step later: see mqa_fwd_kernel
    auto load_side = [&](int tile) {
        KeySide ks{1};
        if (t < 64) {
            const int key = tile * 64 + t;
            ks.raw = key < p.N;
            if (ks.raw && mrow) ks.raw = mrow[key];
        }
        return ks;
    };
    auto stage = [&](unsigned char* __restrict__ img, int tile, int buf, const KeySide& ks) {
        dma_tile<2>(rsK, img, wave, 4, lane, tile * 64, (unsigned)(p.ldk * 2), 0);
        dma_tile<2>(rsV, img + 8192, wave, 4, lane, tile * 64, (unsigned)(p.ldv * 2), 0);
        if (t < 64) {
            int raw = ks.raw;
            asm volatile("" : "+v"(raw));
            kbias[buf * 64 + t] = raw ? 0.f : -INFINITY;
            if (BIAS) {                                                // (the biased variants are at their register limit: these stay at stage time)
                const int kc = min(tile * 64 + t, p.N - 1);
                const int kv = p.kkey4[kc];
                kk4s[buf * 64 + t] = kv;
                const int ka = p.kattr[kc];
                kas[buf * 64 + t] = ka;
                const int ko = wave_or(ka);
                int mn = kv, mx = kv;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mn = min(mn, __shfl_xor(mn, o, 64));
                    mx = max(mx, __shfl_xor(mx, o, 64));
                }
                if (t == 0) { kmm[buf * 4] = mn; kmm[buf * 4 + 1] = mx; kmm[buf * 4 + 2] = ko; }
            }
        }
    };

    bf16x8 qf[2][4], dof[2][4];
    float lse2[2], dlt[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 32 + lr;
        const bool ok = active && qi < p.N;
        const bf16_t* qp = p.q + ((long long)b * p.N + qi) * p.ldq + head * DH;
        const bf16_t* dp = p.dout + ((long long)b * p.N + qi) * p.lddo + head * DH;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[qb][ks] = gload8(qp + ks * 16 + lh * 8, ok);
            dof[qb][ks] = gload8(dp + ks * 16 + lh * 8, ok);
        }
        lse2[qb] = ok ? lse_log2(p.lse[((long long)b * p.H + head) * p.N + qi]) : INFINITY;
        dlt[qb] = ok ? p.ndelta[((long long)b * p.H + head) * p.N + qi] : 0.f;          // = -delta
    }

    int kq4[2] = {0, 0}, aq[2] = {0, 0}, qmin4[2] = {0, 0}, qmax4[2] = {0, 0};
    __amdgpu_buffer_rsrc_t rsT = rsK;
    float* win = wins + wave * WCAP;
    float* part = nullptr;
    if (BIAS) {
        rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.tbl + (long long)(active ? head : 0) * p.LT), 0, p.LT * 4, 0x00020000);
        part = p.dtbl_part + ((((long long)b * p.HG + id.hg) * nqb + qblk) * HPB + wave) * p.LT;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qc = min(q0 + qb * 32 + lr, p.N - 1);
            kq4[qb] = p.qkey4[qc];
            aq[qb] = p.qattr[qc];
            int mn = kq4[qb], mx = kq4[qb];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                mn = min(mn, __shfl_xor(mn, o, 64));
                mx = max(mx, __shfl_xor(mx, o, 64));
            }
            qmin4[qb] = __builtin_amdgcn_readfirstlane(mn);
            qmax4[qb] = __builtin_amdgcn_readfirstlane(mx);
        }
        for (int e = lane; e < WCAP; e += 64) win[e] = 0.f;
    }
    const int aq_or = BIAS ? wave_or(aq[0] | aq[1]) : 0;

    f32x16 dq[2][2];                   // [db][qb]: dQ^T blocks
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[i][j][r] = 0.f;

    const int frow = fsw(lr);
    const TrOff troff = make_troff(lane);
    const int ntiles = qblk + 1;

    KeySide side = load_side(0);
    stage(smem, 0, 0, side);
    side = load_side(1);
    __syncthreads();

    // one tile step; DMA destination and the image being read are distinct __restrict__ parameters of one inlined body, so that the transposed
    // K^T reads do not wait for the in-flight DMA (see mqa_fwd_kernel)
    auto step = [&](unsigned char* __restrict__ nimg, const unsigned char* __restrict__ Kt, int tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) stage(nimg, tile + 1, buf ^ 1, side);
        side = load_side(tile + 2);
        if (active) {
            const unsigned char* Vt = Kt + 8192;
            const float* kbs = kbias + buf * 64;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 st[2], dpt[2];
                const bool sp_tile = BIAS && (aq_or & __builtin_amdgcn_readfirstlane(kmm[buf * 4 + 2])) != 0;       // wave-uniform
                auto init_scores = [&](auto spc) {
                    constexpr bool SP = decltype(spc)::value;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 bv = lds_ld_f4(kbs + kb * 32 + 8 * g + 4 * lh);
                            const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
                            if (BIAS) {
                                const int4 kk = lds_ld_i4(kk4s + buf * 64 + kb * 32 + 8 * g + 4 * lh);
                                int4 ka = make_int4(0, 0, 0, 0);
                                if (SP) ka = lds_ld_i4(kas + buf * 64 + kb * 32 + 8 * g + 4 * lh);
                                const int kkv[4] = {kk.x, kk.y, kk.z, kk.w}, kav[4] = {ka.x, ka.y, ka.z, ka.w};
#pragma unroll
                                for (int c = 0; c < 4; ++c) st[kb][4 * g + c] = bvv[c] + bias_at<SP>(rsT, kq4[qb], kkv[c], aq[qb], kav[c]);
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c) st[kb][4 * g + c] = bvv[c];
                            }
#pragma unroll
                            for (int c = 0; c < 4; ++c) dpt[kb][4 * g + c] = 0.f;
                        }
                };
                if (sp_tile) init_scores(std::true_type{});
                else init_scores(std::false_type{});
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const bf16x8 kf = nat_frag(Kt, kb * 32 + lr, frow, ks, lh);
                        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], st[kb], 0, 0, 0);
                        const bf16x8 vf = nat_frag(Vt, kb * 32 + lr, frow, ks, lh);
                        dpt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[qb][ks], dpt[kb], 0, 0, 0);
                    }
                const bool diag = tile == qblk;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c2, -lse2[qb]));      // masked keys: exp2(-inf) = 0
                        if (diag && kb * 32 + drow(r, lh) > qb * 32 + lr) pv = 0.f;
                        float dpe = dpt[kb][r];
                        if (DROP) dpe = drop_keep(dsalt, q0 + qb * 32 + lr, tile * 64 + kb * 32 + drow(r, lh), p.N, p.drop_thr) ? dpe * p.drop_scale : 0.f;
                        st[kb][r] = pv * (dpe + dlt[qb]);                                                  // dS^T (unscaled)
                    }
                if (BIAS) {
                    // table gradient: every pair adds its dS to the slot it read.  wlo .. whi = byte offsets this pass can touch.
                    const int wlo = qmin4[qb] - __builtin_amdgcn_readfirstlane(kmm[buf * 4 + 1]), whi = qmax4[qb] - __builtin_amdgcn_readfirstlane(kmm[buf * 4]);
                    const bool windowed = (whi - wlo) <= (WCAP - 66) * 4;                                        // the last 64 entries are per-lane dump slots
                    const int kq4w = kq4[qb] - wlo + 4;                                                          // -> byte offset into this wave's window
                    // The window accumulates BLOCK-SCALED INTEGERS: on gfx950 ds_add_f32 is serialised (~190 clk per wave instruction, ~3 clk per
                    // lane: scripts/ubench/lds_atomics.hip) while ds_add_u32 runs at LDS rate (~6 clk).  Per pass: m = max |dS| over the wave,
                    // values are scaled by the power of two that puts m just below 2^19 (<= 2048 addends per slot cannot overflow int32), rounded
                    // to nearest, added as integers (associative: the sum does not depend on the order) and scaled back at the flush.
                    float pmax = 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) pmax = fmaxf(pmax, fabsf(st[kb][r]));
                    pmax = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(wave_max(pmax))));
                    const int pex = (int)((__float_as_uint(pmax) >> 23) & 0xffu) - 127;                          // pmax < 2^(pex + 1)
                    const int psh = min(100, max(-100, 18 - pex));
                    const float pscale = __uint_as_float((unsigned)(127 + psh) << 23), pinv = __uint_as_float((unsigned)(127 - psh) << 23);
                    int* iwin = reinterpret_cast<int*>(win);
                    int spacc = 0;                                                                               // special pairs: summed in a register
                    auto accumulate = [&](auto spc, auto wc) {
                        constexpr bool SP = decltype(spc)::value, WIN = decltype(wc)::value;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int4 kk = lds_ld_i4(kk4s + buf * 64 + kb * 32 + 8 * g + 4 * lh);
                                int4 ka = make_int4(0, 0, 0, 0);
                                if (SP) ka = lds_ld_i4(kas + buf * 64 + kb * 32 + 8 * g + 4 * lh);
                                const int kkv[4] = {kk.x, kk.y, kk.z, kk.w}, kav[4] = {ka.x, ka.y, ka.z, ka.w};
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float ds = st[kb][4 * g + c];
                                    const bool sp = SP && (aq[qb] & kav[c]) != 0;
                                    if (WIN) {
                                        // a special pair's lane adds into its private dump slot instead (64 lanes on slot 0 would serialise)
                                        const int iv = __float2int_rn(ds * pscale);
                                        int la = kq4w - kkv[c];
                                        if (SP) {
                                            la = sp ? (WCAP - 64 + lane) * 4 : la;
                                            spacc += sp ? iv : 0;
                                        }
                                        __hip_atomic_fetch_add(reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(iwin) + la), iv, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WAVEFRONT);
                                    } else if (ds != 0.f) {
                                        const unsigned gi = sp ? 0u : (unsigned)(kq4[qb] - kkv[c]) >> 2;
                                        if (gi < (unsigned)p.LT) unsafeAtomicAdd(part + gi, ds);
                                    }
                                }
                            }
                    };
                    if (pmax > 0.f) {
                        if (windowed) {
                            if (sp_tile) accumulate(std::true_type{}, std::true_type{});
                            else accumulate(std::false_type{}, std::true_type{});
                            // flush into this workgroup's own partial table.  Every update of `part` is an L2 atomic issued by this one wave
                            // (program order => deterministic), so the windowed and the fallback path can be mixed freely.
                            const int n = ((whi - wlo) >> 2) + 2, g0 = (wlo >> 2) - 1;                           // window entry e >= 1 <-> table slot g0 + e
                            for (int e = 1 + lane; e < n; e += 64) {
                                const int iv = iwin[e];
                                const int gi = g0 + e;
                                if (iv != 0) {
                                    iwin[e] = 0;
                                    if (gi >= 1 && gi < p.LT) unsafeAtomicAdd(part + gi, (float)iv * pinv);
                                }
                            }
                            if (sp_tile) {
#pragma unroll
                                for (int o = 32; o > 0; o >>= 1) spacc += __shfl_xor(spacc, o, 64);
                                if (lane == 0 && spacc != 0) unsafeAtomicAdd(part, (float)spacc * pinv);
                            }
                        } else {
                            accumulate(std::true_type{}, std::false_type{});
                        }
                    }
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 dsf = pack8(st[kb], s);
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const bf16x8 ktf = tr_frag(Kt, kb * 32 + s * 16, troff, db);
                            dq[db][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[db][qb], 0, 0, 0);
                        }
                    }
            }
        }
    };
#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
        step(smem + ((tile & 1) ^ 1) * 16384, smem + (tile & 1) * 16384, tile);
        __syncthreads();
    }

    if (!active) return;
    const float sc = p.scale;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 32 + lr;
        if (qi >= p.N) continue;
        bf16_t* op = p.dq + ((long long)b * p.N + qi) * p.lddq + head * DH;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * lh;
                *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf2(dq[db][qb][4 * g] * sc, dq[db][qb][4 * g + 1] * sc),
                                                               pack_bf2(dq[db][qb][4 * g + 2] * sc, dq[db][qb][4 * g + 3] * sc));
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward dK / dV: workgroup = 64 keys x 4 heads of one batch element, 8 waves = (head, 32-key half); loop over the
// 64-query tiles at/after the diagonal.
//   S = Q K^T (q x keys),  P = exp2(S*c2 - lse2[q]),  dP = dO V^T,  dS = P o (dP - delta[q])
//   dV^T (d x keys) += dO^T P ;  dK^T (d x keys) += Q^T dS  (x scale at the end); then summed over the heads through LDS.
// LDS stage: per head Q tile [64 q][64 d] + dO tile (16 KB) -> 64 KB / stage, 2 stages.
// ------------------------------------------------------------------------------------------------------------------
// NH (round 6): heads per workgroup.  4 = one 512-thread workgroup per CU (132 KiB of LDS).  2 = 256 threads and 66 KiB: TWO workgroups per CU with independent
// barriers -- one's barrier / DMA wait runs under the other's MFMAs -- and twice as many workgroups for short sequences (B = 8 x N = 1024: 512 instead of 256 for
// 256 CUs); dK / dV partials per head group: H / NH of them (alm_mqa_bwd_parts tells the caller how many to allocate).
template <bool BIAS, bool DROP = false, int NH = HPB>
__global__ __launch_bounds__(128 * NH, 2) void mqa_bwd_dkv_kernel(AttnParams p) {
    constexpr int NT = 128 * NH, STG = NH * 16384, RTB = NH * 512;             // threads, bytes of one stage of Q / dO tiles, bytes of one buffer of row terms
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];     // 2 x NH x 16 KiB of Q / dO tiles + 2 x NH x 512 B of row terms
    __shared__ int qor_s[BIAS ? 256 : 1];                                      // BIAS: OR of the query attributes of every 64-query tile

    const int nkb = (p.N + 63) / 64;
    // SPLIT-Q (round 6, p.dkv_split = 2; short sequences: fewer key blocks than CUs): the query-tile loop [kblk, nqt) of a key block is cut in two halves run by
    // two workgroups, each writing its own dK / dV partial set (alm_kv_grad_pack sums them in index order: deterministic).  Taken while the doubled launch is
    // still one round of the chip (dkv_split(): B = 8 x N = 512, -10 %); at N = 1024 (256 key-block workgroups of 16 ... 1 steps for 256 CUs) it measured +11 %
    const int SP = p.dkv_split;
    const BlockId id = decode_block(blockIdx.x, nkb, p.HG * SP, p.B, false, false);      // low key blocks are the heavy ones
    const int kblk = id.blk, b = id.b;
    const int hgq = id.hg / SP, part = id.hg - hgq * SP;                       // head group, half of the query range; id.hg = index of the partial set
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hl = wave >> 1, kh = wave & 1;                                   // head within the group, key half
    const int head = hgq * NH + hl;
    const bool active = head < p.H;
    const int lr = lane & 31, lh = lane >> 5;
    const float c2 = p.scale * LOG2E;
    const unsigned dsalt = DROP ? drop_salt(p, b, active ? head : 0) : 0u;
    const int key = kblk * 64 + kh * 32 + lr;

    bool kvalid = key < p.N;
    if (kvalid && p.mask) kvalid = p.mask[(long long)b * p.N + key] != 0;

    // K^T / V^T fragments (B operands): lane = key column, k = head dim
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

    // DMA descriptors over this batch element's Q / dO rows
    const bf16_t* qbase = p.q + (long long)b * p.N * p.ldq;
    const bf16_t* dobase = p.dout + (long long)b * p.N * p.lddo;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qbase), 0, (int)(((long long)(p.N - 1) * p.ldq + p.H * DH) * 2), 0x00020000);
    const auto rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dobase), 0, (int)(((long long)(p.N - 1) * p.lddo + p.H * DH) * 2), 0x00020000);
    const int hclamp = active ? head : 0;
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.nlse + ((long long)b * p.H + hclamp) * p.N), 0, p.N * 4, 0x00020000);
    const auto rsDl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ndelta + ((long long)b * p.H + hclamp) * p.N), 0, p.N * 4, 0x00020000);
    constexpr int ROWT = 2 * STG;                                            // row terms [2 buffers][4 heads][-lse/scale | -delta][64 queries] fp32
    auto stage = [&](unsigned char* __restrict__ tiles, int qt, int buf) {                      // tiles = smem + buf * STG
        // wave (hl, kh) fetches head hl's Q tile (kh == 0) or dO tile (kh == 1): 8 pieces each -- and, through the SAME DMA queue, the tile's 64
        // row terms (-lse / scale for kh == 0, -delta for kh == 1: the accumulators' initial values).  They used to be 16 register loads per
        // wave at the top of every step, issued right after the next tile's DMA: vector-memory results retire in order, so waiting for them
        // meant waiting for that whole DMA -- the prefetch never overlapped the MFMAs.
        if (!active) return;
        unsigned char* img = tiles + hl * 16384 + kh * 8192;
        if (kh == 0) dma_tile<8>(rsQ, img, 0, 1, lane, qt * 64, (unsigned)(p.ldq * 2), (unsigned)(head * DH * 2));
        else dma_tile<8>(rsD, img, 0, 1, lane, qt * 64, (unsigned)(p.lddo * 2), (unsigned)(head * DH * 2));
        const int qi = qt * 64 + lane;
        const unsigned vo = qi < p.N ? (unsigned)qi * 4u : OOB;               // rows >= N read 0 (harmless: their Q / dO rows are zero)
        unsigned char* rt = smem + ROWT + buf * RTB + hl * 512 + kh * 256;
        if (kh == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (lds_void*)rt, 4, vo, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsDl, (lds_void*)rt, 4, vo, 0, 0, 0);
    };

    // the same, BRANCH-FREE (the software-pipelined step issues these pieces between its MFMAs: one basic block, explicit schedule groups): the
    // descriptor / pitch of the wave's operand (Q for kh == 0, dO for kh == 1) are scalar selects; an inactive wave fetches head 0's tile into its own slot
    const auto rsX = kh ? rsD : rsQ;
    const auto rsR = kh ? rsDl : rsL;
    const unsigned ldx_bytes = (unsigned)((kh ? p.lddo : p.ldq) * 2), colx_bytes = (unsigned)(hclamp * DH * 2);
    auto stage_piece = [&](unsigned char* __restrict__ tiles, int qt, int buf, int piece) {        // piece 0..7: tile rows 8 piece .. + 7; piece 8: the row terms
        unsigned char* img = tiles + hl * 16384 + kh * 8192;
        if (piece < 8) {
            const int row = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ fsw(row);
            const unsigned vo = (unsigned)(qt * 64 + row) * ldx_bytes + colx_bytes + (unsigned)c * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(img + piece * 1024), 16, vo, 0, 0, 0);
        } else {
            const int qi = qt * 64 + lane;
            const unsigned vo = qi < p.N ? (unsigned)qi * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (lds_void*)(smem + ROWT + buf * RTB + hl * 512 + kh * 256), 4, vo, 0, 0, 0);
        }
    };

    f32x16 dkt[2], dvt[2];             // [db]: (d x 32 keys) blocks
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[i][r] = 0.f; dvt[i][r] = 0.f; }

    const int frow = fsw(lr);
    const TrOff troff = make_troff(lane);
    const int nqt = (p.N + 63) / 64;
    const int qhalf = (nqt - kblk + 1) / 2;                              // SPLIT-Q: part 0 = the diagonal tile and the first half, part 1 = the rest (may be empty)
    const int q_lo = part ? kblk + qhalf : kblk, q_hi = (SP > 1 && !part) ? kblk + qhalf : nqt;
    const int key_eff = kvalid ? key : 0x7fffffff;                       // a masked / out-of-range key "follows" every query
    int kk4l = 0, kal = 0;
    __amdgpu_buffer_rsrc_t rsT = rsL, rsKQ = rsL, rsAQ = rsL;
    if (BIAS) {
        rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.tbl + (long long)hclamp * p.LT), 0, p.LT * 4, 0x00020000);
        rsKQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.qkey4), 0, p.N * 4, 0x00020000);
        rsAQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.qattr), 0, p.N * 4, 0x00020000);
        kk4l = p.kkey4[min(key, p.N - 1)];
        kal = p.kattr[min(key, p.N - 1)];
        for (int i = t; i < 256; i += NT) qor_s[i] = 0;
        __syncthreads();
        for (int i = t; i < p.N; i += NT) atomicOr(&qor_s[min(i >> 6, 255)], p.qattr[i]);
    }
    const int kal_or = BIAS ? wave_or(kal) : 0;

    if (q_lo < q_hi) stage(smem, q_lo, 0);                               // (workgroup-uniform)
    __syncthreads();
    DKV_PROBE_DECL

    // one query-tile step; DMA destination and the images being read are distinct __restrict__ parameters of one inlined body, so that the transposed
    // Q^T / dO^T reads do not wait for the in-flight DMA (see mqa_fwd_kernel)
    // DIAG (compile time): the diagonal query tile (qt == kblk, the first step) is the only one where the causal rule can hide a key from a query.  Round 6:
    // off the diagonal NO per-element mask is applied at all -- every key precedes every query there, and an invalid key (key-padding mask, or beyond N) only
    // ever produces garbage in ITS OWN column of dK^T / dV^T (the MFMA's B operand column = the key), which the epilogue replaces by zero.  That removes an
    // index build + compare + select per score (96 of 176 VALU instructions of the step) from a loop whose VALU, MFMA and LDS phases barely overlap.
    auto step = [&](unsigned char* __restrict__ ntiles_img, const unsigned char* __restrict__ ctiles, int qt, auto diag_c, auto more_c) {
        constexpr bool DIAG = decltype(diag_c)::value;
        constexpr bool MORE = decltype(more_c)::value;                      // compile time on the pipelined path: tile qt + 1 exists (its DMA is part of the schedule)
        constexpr bool PIPE = !DIAG && !DROP && !BIAS;
        const int buf = (qt - q_lo) & 1;
        if (!PIPE && qt + 1 < q_hi) stage(ntiles_img, qt + 1, buf ^ 1);
        DKV_PROBE_T(0);
        if (active) {
            const unsigned char* Qt = ctiles + hl * 16384;
            const unsigned char* Dt = Qt + 8192;
            const int q0 = qt * 64;
            // S' = Q K^T - lse / scale, dP' = dO V^T - delta : [qb] (32 queries x 32 keys) blocks; lane = key column, registers = query
            // rows.  The row terms are the accumulators' initial values (bounds-checked loads: rows >= N read 0 -- harmless, their
            // Q / dO rows are zero so P only ever meets zeros).
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            f32x16 s[2], dp[2];
            const float* rowt = reinterpret_cast<const float*>(smem + ROWT + buf * RTB + hl * 512);      // [64] -lse/scale, [64] -delta (staged by DMA)
            if constexpr (PIPE) {
                // ---- off-diagonal step, SOFTWARE-PIPELINED over the two 32-query blocks (round 6).  Measured before (scripts/attn_probe.py dkv): the step spent
                // 1095 cycles in the exp2 / dS block during which the SIMD's matrix pipe idled -- both waves of a SIMD run the same phase at the same time
                // (one barrier per step), so nothing else filled it.  Order now:   M1(0) | M1(1) || E(0) | M2(0) || E(1) | M2(1)
                //   M1(qb) = S', dP' MFMAs of query block qb (8), E(qb) = exp2 + dS + bf16 packs (VALU), M2(qb) = dV, dK MFMAs (8)
                // with the VALU of one block issued BETWEEN the MFMAs of the other (explicit schedule groups: one MFMA, its fragment reads, ~7 VALU).
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                auto init_rows = [&](int qb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ql = qb * 32 + 8 * g + 4 * lh;
                        const float4 lraw = lds_ld_f4(rowt + ql);
                        const float4 draw = lds_ld_f4(rowt + 64 + ql);
                        s[qb][4 * g] = lraw.x; s[qb][4 * g + 1] = lraw.y; s[qb][4 * g + 2] = lraw.z; s[qb][4 * g + 3] = lraw.w;
                        dp[qb][4 * g] = draw.x; dp[qb][4 * g + 1] = draw.y; dp[qb][4 * g + 2] = draw.z; dp[qb][4 * g + 3] = draw.w;
                    }
                };
                auto m1 = [&](int qb) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 qa = nat_frag(Qt, qb * 32 + lr, frow, ks, lh);
                        s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s[qb], 0, 0, 0);
                        const bf16x8 da = nat_frag(Dt, qb * 32 + lr, frow, ks, lh);
                        dp[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp[qb], 0, 0, 0);
                    }
                };
                bf16x8 pf[2][2], dsf[2][2];                                             // [qb][16-query step]: P and dS as bf16 B operands
                auto e = [&](int qb) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 sv = f32x2{s[qb][r], s[qb][r + 1]} * f32x2{c2, c2};
                        const f32x2 pv = {__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])};
                        const f32x2 dv = f32x2{dp[qb][r], dp[qb][r + 1]} * pv;
                        s[qb][r] = pv[0]; s[qb][r + 1] = pv[1];
                        dp[qb][r] = dv[0]; dp[qb][r + 1] = dv[1];
                    }
#pragma unroll
                    for (int st = 0; st < 2; ++st) { pf[qb][st] = pack8(s[qb], st); dsf[qb][st] = pack8(dp[qb], st); }
                };
                auto m2 = [&](int qb) {
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const bf16x8 dotf = tr_frag(Dt, qb * 32 + st * 16, troff, db);
                            dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf[qb][st], dvt[db], 0, 0, 0);
                            const bf16x8 qtf = tr_frag(Qt, qb * 32 + st * 16, troff, db);
                            dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[qb][st], dkt[db], 0, 0, 0);
                        }
                };
                // The 8 + 1 DMA pieces of the NEXT tile ride in the same schedule -- one piece per two MFMAs (the probe: 830 cycles of DMA issue per step at
                // the top of the step, every wave of the workgroup at once and the matrix pipe idle; an LDS-DMA piece costs ~60 cycles beside MFMAs).
                init_rows(0);
                init_rows(1);
                m1(0);
                __builtin_amdgcn_sched_barrier(0);
                if (MORE) {
#pragma unroll
                    for (int pc = 0; pc < 4; ++pc) stage_piece(ntiles_img, qt + 1, buf ^ 1, pc);
                }
                m1(1);
                e(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {                                            // M1(1) || E(0): per MFMA its fragment read + a slice of the VALU block
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (MORE && (i & 1)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                DKV_PROBE_T(1);
                if (MORE) {
#pragma unroll
                    for (int pc = 4; pc < 9; ++pc) stage_piece(ntiles_img, qt + 1, buf ^ 1, pc);
                }
                m2(0);
                e(1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {                                            // M2(0) || E(1)
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (MORE && i < 5) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                DKV_PROBE_T(2);
                m2(1);
            } else {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql = qb * 32 + 8 * g + 4 * lh;                            // 4 consecutive query rows of the tile
                    const float4 lraw = lds_ld_f4(rowt + ql);
                    const float4 draw = lds_ld_f4(rowt + 64 + ql);
                    s[qb][4 * g] = lraw.x; s[qb][4 * g + 1] = lraw.y; s[qb][4 * g + 2] = lraw.z; s[qb][4 * g + 3] = lraw.w;
                    if (DROP) {                                                             // -delta joins AFTER the keep mask (below): dS = P (keep dP / (1 - p) - delta)
                        dp[qb][4 * g] = 0.f; dp[qb][4 * g + 1] = 0.f; dp[qb][4 * g + 2] = 0.f; dp[qb][4 * g + 3] = 0.f;
                    } else {
                        dp[qb][4 * g] = draw.x; dp[qb][4 * g + 1] = draw.y; dp[qb][4 * g + 2] = draw.z; dp[qb][4 * g + 3] = draw.w;
                    }
                }
            if (BIAS) {                                                                 // rows >= N read offset 0 -> an out-of-table gather -> 0
                auto add_bias = [&](auto spc) {
                    constexpr bool SP = decltype(spc)::value;
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int qq0 = q0 + qb * 32 + 8 * g + 4 * lh;
                            const u32x4 kq = __builtin_amdgcn_raw_buffer_load_b128(rsKQ, qq0 * 4, 0, 0);
                            u32x4 aqv = {0u, 0u, 0u, 0u};
                            if (SP) aqv = __builtin_amdgcn_raw_buffer_load_b128(rsAQ, qq0 * 4, 0, 0);
#pragma unroll
                            for (int c = 0; c < 4; ++c) s[qb][4 * g + c] += bias_at<SP>(rsT, (int)kq[c], kk4l, (int)aqv[c], kal);
                        }
                };
                if ((__builtin_amdgcn_readfirstlane(qor_s[min(qt, 255)]) & kal_or) != 0) add_bias(std::true_type{});
                else add_bias(std::false_type{});
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const bf16x8 qa = nat_frag(Qt, qb * 32 + lr, frow, ks, lh);
                    s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s[qb], 0, 0, 0);
                    const bf16x8 da = nat_frag(Dt, qb * 32 + lr, frow, ks, lh);
                    dp[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp[qb], 0, 0, 0);
                }
            DKV_PROBE_T(1);
            if constexpr (!DIAG && !DROP) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 c2v = {c2, c2};
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {                               // packed fp32 math (v_pk_mul_f32: two scores per instruction)
                        const f32x2 sv = f32x2{s[qb][r], s[qb][r + 1]} * c2v;
                        const f32x2 pv = {__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])};
                        const f32x2 dv = f32x2{dp[qb][r], dp[qb][r + 1]} * pv;
                        s[qb][r] = pv[0]; s[qb][r + 1] = pv[1];
                        dp[qb][r] = dv[0]; dp[qb][r + 1] = dv[1];
                    }
            } else {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = q0 + qb * 32 + drow(r, lh);
                    float pv = __builtin_amdgcn_exp2f(s[qb][r] * c2);
                    if (DIAG ? key_eff > qq : !kvalid) pv = 0.f;                    // masked key | causal (diagonal tile only)
                    if (DROP) {
                        const bool kp = drop_keep(dsalt, qq, key, p.N, p.drop_thr);
                        const float nd = rowt[64 + qb * 32 + drow(r, lh)];           // -delta of this query row
                        dp[qb][r] = pv * ((kp ? dp[qb][r] * p.drop_scale : 0.f) + nd);
                        s[qb][r] = kp ? pv : 0.f;                                   // dV takes the dropped probabilities (x 1 / (1 - p) at the end)
                    } else {
                        s[qb][r] = pv;
                        dp[qb][r] *= pv;
                    }
                }
            }
            DKV_PROBE_T(2);
            // dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 64 queries of the tile: 4 steps of 16)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const bf16x8 pf = pack8(s[qb], st);
                    const bf16x8 dsf = pack8(dp[qb], st);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 dotf = tr_frag(Dt, qb * 32 + st * 16, troff, db);
                        dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dvt[db], 0, 0, 0);
                        const bf16x8 qtf = tr_frag(Qt, qb * 32 + st * 16, troff, db);
                        dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dkt[db], 0, 0, 0);
                    }
                }
            }
            DKV_PROBE_T(3);
        }
    };
    int qfirst = q_lo;                                                       // first off-diagonal tile of this workgroup
    if (part == 0) {
        step(smem + STG, smem, kblk, std::true_type{}, std::true_type{});    // the diagonal tile
        __syncthreads();
        qfirst = kblk + 1;
    }
    DKV_PROBE_T(4);
    constexpr bool PEEL_LAST = !DROP && !BIAS;                               // (only the pipelined step bakes "a next tile exists" into its schedule)
#pragma unroll 1
    for (int qt = qfirst; qt + (PEEL_LAST ? 1 : 0) < q_hi; ++qt) {
        const int buf = (qt - q_lo) & 1;
        step(smem + (buf ^ 1) * STG, smem + buf * STG, qt, std::false_type{}, std::true_type{});
        __syncthreads();
        DKV_PROBE_T(4);
    }
    if (PEEL_LAST && q_hi - 1 >= qfirst) {                                   // the last query tile: nothing left to fetch
        const int buf = (q_hi - 1 - q_lo) & 1;
        step(smem + (buf ^ 1) * STG, smem + buf * STG, q_hi - 1, std::false_type{}, std::false_type{});
        __syncthreads();
        DKV_PROBE_T(4);
    }
    DKV_PROBE_FLUSH(blockIdx.x * 8 + wave, q_hi - q_lo);

    // reduce over the 4 heads through LDS: red[wave][d][32 keys (+1 pad)] fp32.  The pad matters: the accumulators are written key-major (lane = key)
    // and read dim-major (lane = d, for coalesced global stores) -- with a 32-float row every lane of the read hit ONE bank (32-way conflict,
    // SQ_LDS_BANK_CONFLICT = 54 % of this kernel's LDS cycles, ~17 us per workgroup); 33 makes both directions conflict-free.
    constexpr int RS = 33, RW = 64 * RS;
    float* red = reinterpret_cast<float*>(smem);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        const f32x16* acc = pass == 0 ? dkt : dvt;
        if (active) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave * RW + (db * 32 + drow(r, lh)) * RS + lr] = acc[db][r];
        }
        __syncthreads();
        float* outp = (pass == 0 ? p.dk : p.dv) + (long long)id.hg * p.part_stride;
        const float sc = pass == 0 ? p.scale : (DROP ? p.drop_scale : 1.f);
        const int nh = min(NH, p.H - hgq * NH);
        for (int e = t; e < 64 * 64; e += NT) {
            const int kk = e >> 6, d = e & 63;          // output element (key kk of the block, dim d): coalesced along d
            const int half = kk >> 5, kl = kk & 31;
            float sum = 0.f;
            for (int hh = 0; hh < nh; ++hh) sum += red[(hh * 2 + half) * RW + d * RS + kl];
            if (kblk * 64 + kk < p.N) {
                // a key the padding mask hides receives no gradient: its column may hold anything (the off-diagonal steps do not mask) -- SELECT zero
                const bool kv_ok = !p.mask || p.mask[(long long)b * p.N + kblk * 64 + kk] != 0;
                outp[((long long)b * p.N + kblk * 64 + kk) * p.lddk + d] = kv_ok ? sum * sc : 0.f;
            }
        }
    }
}

// dtbl[h][t] = sum over (batch element, query block) of the dQ kernel's per-workgroup partial tables
__global__ __launch_bounds__(256) void bias_grad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dtbl, int B, int HG, int nqb,
                                                               int H, int LT, float scale) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    if (t >= LT) return;
    const int hg = h / HPB, w = h % HPB;
    float s0 = 0.f, s1 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* pp = part + ((((long long)b * HG + hg) * nqb) * HPB + w) * LT + t;
        int qb = 0;
        for (; qb + 1 < nqb; qb += 2) {
            s0 += pp[(long long)qb * HPB * LT];
            s1 += pp[(long long)(qb + 1) * HPB * LT];
        }
        if (qb < nqb) s0 += pp[(long long)qb * HPB * LT];
    }
    dtbl[(long long)h * LT + t] = (s0 + s1) * scale;               // the partials hold sums of dS; sim = scale * (q.k + tbl)
}

}  // namespace

static int check_attn(int B, int N, int H, long long ldq, long long ldk, long long ldv, long long ldo) {
    if (B <= 0 || N <= 0 || H <= 0 || H > 64) return ALM_ERR_UNSUPPORTED;
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return ALM_ERR_BAD_ARG;
    if ((long long)N * ldq * 2 >= 0x7fffffffLL || (long long)N * ldk * 2 >= 0x7fffffffLL || (long long)N * ldv * 2 >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;
    // the tile DMA relies on the buffer bounds check for rows N .. N + 63 of the last tile: their 32-bit byte offsets must not wrap
    const long long ldmax = ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv);
    if ((long long)(N + 64) * (ldmax > ldo ? ldmax : ldo) * 2 >= 0xffffffffLL) return ALM_ERR_UNSUPPORTED;
    return 0;
}

extern "C" int alm_mqa_head_groups(int H) { return (H + HPB - 1) / HPB; }

// heads per workgroup of the dK / dV kernel for a shape (see mqa_bwd_dkv_kernel's NH).  ALM_ATTN_DKV_NH = 4 | 2 forces one (A/B runs).
static int dkv_heads_per_block(int B, int N, int H) {
    static const int env_nh = [] { const char* e = getenv("ALM_ATTN_DKV_NH"); return e ? atoi(e) : 0; }();
    if (env_nh == 2 || env_nh == 4) return env_nh;
    (void)B; (void)H;
    // measured (profiles/r6t_dkv_nh_ab.log, B = 8, H = 8, one box, two interleaved rounds): 2 heads per workgroup is +9 % / +8 % SLOWER at N = 1024 / 2048 (twice
    // the workgroups, each with its own K / V load, diagonal step and head-reduce epilogue), -0.6 % at N = 8 253 and -1.3 % at N = 16 385 (the second workgroup
    // of a CU runs under the first one's barrier / DMA waits; the per-workgroup overhead is amortised over >= 65 query-tile steps)
    return N >= 8192 ? 2 : HPB;
}
// SPLIT-Q of the dK / dV kernel (see mqa_bwd_dkv_kernel): 2 when the launch would otherwise fill at most HALF the CUs.  ALM_ATTN_DKV_SPLIT = 1 | 2 forces.
static int dkv_split(int B, int N, int H, int nh) {
    static const int env_sp = [] { const char* e = getenv("ALM_ATTN_DKV_SPLIT"); return e ? atoi(e) : 0; }();
    if (env_sp == 1 || env_sp == 2) return env_sp;
    // measured (profiles/r6z_dkv_split_ab.log, B = 8, H = 8): -10 % at N = 512 (128 -> 256 workgroups: still ONE round of the 256 CUs, the heaviest block halves),
    // but +11 % at N = 1024 and +14 % at N = 1536 -- the kernel holds 132 KiB of LDS, one workgroup per CU: 512 workgroups are two rounds (8 + 4 steps instead of
    // 16) and every workgroup pays its own K / V load, head-reduce epilogue and partial set.  So: only while the doubled launch still fits one round.
    const long long wgs = (long long)((N + 63) / 64) * ((H + nh - 1) / nh) * B;
    return (N > 64 && 2 * wgs <= 256) ? 2 : 1;
}
extern "C" int alm_mqa_bwd_parts(int B, int N, int H) {
    const int nh = dkv_heads_per_block(B, N, H);
    return ((H + nh - 1) / nh) * dkv_split(B, N, H, nh);
}

struct BiasArgs { const float* tbl; int LT; const int* qkey4; const int* kkey4; const int* qattr; const int* kattr; float* dtbl_part; };

static int check_bias(const BiasArgs& ba, bool bwd) {
    if (!ba.tbl) return 0;
    if (ba.LT < 2 || ba.LT > (1 << 28) || !ba.qkey4 || !ba.kkey4 || !ba.qattr || !ba.kattr || (bwd && !ba.dtbl_part)) return ALM_ERR_BAD_ARG;
    if (((uintptr_t)ba.qkey4 & 15) || ((uintptr_t)ba.qattr & 15)) return ALM_ERR_BAD_ARG;                // read 4 rows at a time (dK/dV kernel)
    return 0;
}

// dropout_p in [0, 1): > 0 selects the DROP instantiations (training-mode attention dropout), `seed` picks the mask stream.  seed_dev (device
// uint64, may be NULL): the stream is seed + *seed_dev, read by the kernels when they RUN -- a by-value seed is baked into a captured hipGraph and every
// replay would repeat the same keep mask; with the counter on the device a small captured kernel advances it and each replay draws a new mask.
static bool set_dropout(AttnParams& p, float dropout_p, unsigned long long seed, const void* seed_dev) {
    p.seed_dev = nullptr;
    if (!(dropout_p > 0.f)) return false;
    p.seed_dev = reinterpret_cast<const unsigned long long*>(seed_dev);
    const double thr = (double)dropout_p * 4294967296.0;
    p.drop_thr = thr >= 4294967295.0 ? 4294967295u : (unsigned)thr;
    p.drop_scale = 1.f / (1.f - dropout_p);
    p.seed_lo = (unsigned)(seed & 0xffffffffu);
    p.seed_hi = (unsigned)(seed >> 32);
    return true;
}

static int attn_fwd_impl(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                         const unsigned char* mask, void* o, long long ldo, float* lse, int B, int N, int H, int dim_head,
                         float scale, const BiasArgs& ba, float dropout_p, unsigned long long seed, const void* seed_dev, void* stream) {
    if (dropout_p < 0.f || dropout_p >= 1.f) return ALM_ERR_BAD_ARG;
    if (dim_head != DH) return ALM_ERR_UNSUPPORTED;
    int rc = check_attn(B, N, H, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 7)) return ALM_ERR_BAD_ARG;
    AttnParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.mask = mask; p.o = (bf16_t*)o; p.lse = lse;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.B = B; p.N = N; p.H = H; p.HG = alm_mqa_head_groups(H); p.scale = scale;
    rc = check_bias(ba, false);
    if (rc) return rc;
    p.tbl = ba.tbl; p.LT = ba.LT; p.qkey4 = ba.qkey4; p.kkey4 = ba.kkey4; p.qattr = ba.qattr; p.kattr = ba.kattr;
    const int npair = ((N + 31) / 32 + 1) / 2;                           // 32-query blocks, taken two (idx, last - idx) per workgroup
    const bool drop = set_dropout(p, dropout_p, seed, seed_dev);
    // QB = 2 (one 64-query block per workgroup and wave: half the K / V fragment reads per MFMA, heavy / light blocks paired on a CU) is what the kernel
    // did before the equal-length pairing of QB = 1 fixed the long pole at N = 2048.  ALM_ATTN_FWD_QB2_MINN=<N>: sequences of at least N keys take it (A/B).
    static const int qb2_minn = [] { const char* e = getenv("ALM_ATTN_FWD_QB2_MINN"); return e ? atoi(e) : 0; }();
    if (qb2_minn > 0 && N >= qb2_minn && !drop && !p.tbl) {
        hipLaunchKernelGGL((mqa_fwd_kernel<false, 2>), dim3(((N + 63) / 64) * p.HG * B), dim3(256), FWD_LDS<false>, (hipStream_t)stream, p);
        ALM_LAUNCH_CHECK();
        return 0;
    }
    if (drop) {
        if (p.tbl) hipLaunchKernelGGL((mqa_fwd_kernel<true, 1, true>), dim3(npair * p.HG * B), dim3(256), FWD_LDS<true>, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((mqa_fwd_kernel<false, 1, true>), dim3(npair * p.HG * B), dim3(256), FWD_LDS<false>, (hipStream_t)stream, p);
    } else if (p.tbl) hipLaunchKernelGGL((mqa_fwd_kernel<true, 1>), dim3(npair * p.HG * B), dim3(256), FWD_LDS<true>, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((mqa_fwd_kernel<false, 1>), dim3(npair * p.HG * B), dim3(256), FWD_LDS<false>, (hipStream_t)stream, p);
    ALM_LAUNCH_CHECK();
    return 0;
}

// dk / dv: fp32 partials [alm_mqa_bwd_parts(B, N, H)][B*N][lddk] (64 valid columns each; partial g at + g * part_stride floats);
// the caller (alm_kv_grad_pack) adds the partials.  delta: fp32 workspace [2][B][H][N].
static int attn_bwd_impl(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                         const unsigned char* mask, const void* o, long long ldo, const float* lse, const void* dout, long long lddo,
                         void* dq, long long lddq, float* dk, float* dv, long long lddk, long long part_stride, float* delta, int B,
                         int N, int H, int dim_head, float scale, const BiasArgs& ba, float dropout_p, unsigned long long seed, const void* seed_dev, void* stream) {
    if (dim_head != DH) return ALM_ERR_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return ALM_ERR_BAD_ARG;
    int rc = check_attn(B, N, H, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    if ((lddo & 7) || (lddq & 3) || (long long)N * lddo * 2 >= 0x7fffffffLL) return ALM_ERR_BAD_ARG;
    if ((long long)(N + 64) * lddo * 2 >= 0xffffffffLL) return ALM_ERR_UNSUPPORTED;          // dO tile DMA: see check_attn
    rc = check_bias(ba, true);
    if (rc) return rc;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 7)) return ALM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* ndelta = delta;
    float* nlse = delta + (long long)B * H * N;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(((long long)B * N + 3) / 4), dim3(256), 0, st, (const bf16_t*)o, ldo, (const bf16_t*)dout, lddo, lse,
                       ndelta, nlse, 1.0f / scale, B, N, H);
    AttnParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.mask = mask; p.lse = const_cast<float*>(lse);
    p.dout = (const bf16_t*)dout; p.ndelta = ndelta; p.nlse = nlse; p.dq = (bf16_t*)dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.part_stride = part_stride;
    p.B = B; p.N = N; p.H = H; p.HG = alm_mqa_head_groups(H); p.scale = scale;
    p.tbl = ba.tbl; p.LT = ba.LT; p.qkey4 = ba.qkey4; p.kkey4 = ba.kkey4; p.qattr = ba.qattr; p.kattr = ba.kattr; p.dtbl_part = ba.dtbl_part;
    const int nqb = (N + 63) / 64;
    const bool drop = set_dropout(p, dropout_p, seed, seed_dev);
    if (drop) {
        if (p.tbl) hipLaunchKernelGGL((mqa_bwd_dq_kernel<true, true>), dim3(nqb * p.HG * B), dim3(256), DQ_LDS<true>, st, p);
        else hipLaunchKernelGGL((mqa_bwd_dq_kernel<false, true>), dim3(nqb * p.HG * B), dim3(256), DQ_LDS<false>, st, p);
    } else if (p.tbl) hipLaunchKernelGGL(mqa_bwd_dq_kernel<true>, dim3(nqb * p.HG * B), dim3(256), DQ_LDS<true>, st, p);
    else hipLaunchKernelGGL(mqa_bwd_dq_kernel<false>, dim3(nqb * p.HG * B), dim3(256), DQ_LDS<false>, st, p);
    // dK / dV: 4 heads per workgroup (one 512-thread workgroup per CU) or 2 (two 256-thread workgroups per CU): dkv_heads_per_block; the caller sized the
    // partial buffers with alm_mqa_bwd_parts
    const int nh = dkv_heads_per_block(B, N, H);
    p.HG = (H + nh - 1) / nh;
    p.dkv_split = dkv_split(B, N, H, nh);
    auto launch_dkv = [&](auto kern, int threads, int lds) -> hipError_t {
        // the LDS attribute is set once per KERNEL: the eight instantiations below are eight function-pointer VALUES of one type (one operator() of this
        // lambda, one static): keyed on the value (ADVICE r5 found a per-type flag shared by four kernels in the grouped GEMM launcher)
        static const void* done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        const void* fn = reinterpret_cast<const void*>(kern);
        bool seen = false;
        for (int i = 0; i < 8; ++i) seen = seen || done[i] == fn;
        if (!seen) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            for (int i = 0; i < 8; ++i) if (done[i] == nullptr) { done[i] = fn; break; }
        }
        hipLaunchKernelGGL(kern, dim3(nqb * p.HG * p.dkv_split * B), dim3(threads), lds, st, p);
        return hipSuccess;
    };
    hipError_t le;
    if (nh == 2) {
        constexpr int LDS2 = 2 * 2 * 16384 + 2 * 2 * 512;
        if (drop) le = p.tbl ? launch_dkv(mqa_bwd_dkv_kernel<true, true, 2>, 256, LDS2) : launch_dkv(mqa_bwd_dkv_kernel<false, true, 2>, 256, LDS2);
        else le = p.tbl ? launch_dkv(mqa_bwd_dkv_kernel<true, false, 2>, 256, LDS2) : launch_dkv(mqa_bwd_dkv_kernel<false, false, 2>, 256, LDS2);
    } else {
        constexpr int LDS4 = 131072 + 4096;
        if (drop) le = p.tbl ? launch_dkv(mqa_bwd_dkv_kernel<true, true, 4>, 512, LDS4) : launch_dkv(mqa_bwd_dkv_kernel<false, true, 4>, 512, LDS4);
        else le = p.tbl ? launch_dkv(mqa_bwd_dkv_kernel<true, false, 4>, 512, LDS4) : launch_dkv(mqa_bwd_dkv_kernel<false, false, 4>, 512, LDS4);
    }
    if (le != hipSuccess) return (int)le;
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_mqa_attn_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                const unsigned char* mask, void* o, long long ldo, float* lse, int B, int N, int H, int dim_head,
                                float scale, float dropout_p, unsigned long long seed, const void* seed_dev, void* stream) {
    return attn_fwd_impl(q, ldq, k, ldk, v, ldv, mask, o, ldo, lse, B, N, H, dim_head, scale, BiasArgs{}, dropout_p, seed, seed_dev, stream);
}

extern "C" int alm_mqa_attn_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                const unsigned char* mask, const void* o, long long ldo, const float* lse, const void* dout, long long lddo,
                                void* dq, long long lddq, float* dk, float* dv, long long lddk, long long part_stride, float* delta, int B,
                                int N, int H, int dim_head, float scale, float dropout_p, unsigned long long seed, const void* seed_dev, void* stream) {
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, mask, o, ldo, lse, dout, lddo, dq, lddq, dk, dv, lddk, part_stride, delta, B, N, H, dim_head,
                         scale, BiasArgs{}, dropout_p, seed, seed_dev, stream);
}

// Same contractions with the structured score bias described at the top of this file.  tbl: fp32 [H][LT] in raw-score units (bias /
// scale), slot 0 = the value of "special" pairs; qkey4 / kkey4 / qattr / kattr: int32 [N] (16-B aligned), shared by the batch.
extern "C" int alm_mqa_attn_bias_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                     const unsigned char* mask, void* o, long long ldo, float* lse, int B, int N, int H, int dim_head,
                                     float scale, const float* tbl, int LT, const int* qkey4, const int* kkey4, const int* qattr,
                                     const int* kattr, float dropout_p, unsigned long long seed, const void* seed_dev, void* stream) {
    if (!tbl) return ALM_ERR_BAD_ARG;
    return attn_fwd_impl(q, ldq, k, ldk, v, ldv, mask, o, ldo, lse, B, N, H, dim_head, scale, BiasArgs{tbl, LT, qkey4, kkey4, qattr, kattr, nullptr},
                         dropout_p, seed, seed_dev, stream);
}

// dtbl_part: fp32 [alm_attn_bias_part_rows(B, N, H)][LT] per-workgroup partial table gradients, ACCUMULATED into (zero it before the
// first layer's backward, call alm_attn_bias_grad_reduce after the last: every layer of a stack shares one bias table).
extern "C" int alm_mqa_attn_bias_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                     const unsigned char* mask, const void* o, long long ldo, const float* lse, const void* dout,
                                     long long lddo, void* dq, long long lddq, float* dk, float* dv, long long lddk, long long part_stride,
                                     float* delta, int B, int N, int H, int dim_head, float scale, const float* tbl, int LT, const int* qkey4,
                                     const int* kkey4, const int* qattr, const int* kattr, float* dtbl_part, float dropout_p,
                                     unsigned long long seed, const void* seed_dev, void* stream) {
    if (!tbl) return ALM_ERR_BAD_ARG;
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, mask, o, ldo, lse, dout, lddo, dq, lddq, dk, dv, lddk, part_stride, delta, B, N, H, dim_head,
                         scale, BiasArgs{tbl, LT, qkey4, kkey4, qattr, kattr, dtbl_part}, dropout_p, seed, seed_dev, stream);
}

extern "C" int alm_attn_bias_part_rows(int B, int N, int H) { return B * alm_mqa_head_groups(H) * ((N + 63) / 64) * HPB; }

// dtbl fp32 [H][LT] = scale * sum of the partial tables (= d loss / d tbl; the partials accumulate the un-scaled dS)
extern "C" int alm_attn_bias_grad_reduce(const float* dtbl_part, float* dtbl, int B, int N, int H, int LT, float scale, void* stream) {
    if (B <= 0 || N <= 0 || H <= 0 || LT <= 0) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3((LT + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, dtbl_part, dtbl, B,
                       alm_mqa_head_groups(H), (N + 63) / 64, H, LT, scale);
    ALM_LAUNCH_CHECK();
    return 0;
}

#if defined(ALM_ATTN_PROBE) || defined(ALM_DKV_PROBE)
extern "C" int alm_attn_probe_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_probe), sizeof(unsigned long long) * n);
}
#endif
