// Token-id side of the hot path (gfx950, HBM-bound byte/index work -- deliberately NOT reshaped into GEMMs):
//   * embedding assembly: per-quantizer offset embedding + quantizer-position embedding + start tokens + concat
//     (reference audiolm_pytorch.py:709-713, :894-918, :1186-1223) and its scatter-add backward (x grad_shrink alpha, :93-94, :478)
//   * row gather / scatter used to regroup the final hidden states per logit head ('b (n q) d -> q (b n) d', :965-983, :1325-1361)
//   * cross-entropy forward (online log-sum-exp, fp32) and backward (softmax - onehot, bf16 for the dgrad/wgrad GEMMs)
//     (reference F.cross_entropy(ignore_index = -1), audiolm_pytorch.py:1561-1565, :1839-1849, :2122-2132)
//   * value-residual mixing v <- (v + v_layer0) / 2 (:357-358) and the matching gradient fan-in.
#include "common.hpp"
#include "../../include/audiolm_hip.h"

namespace {

constexpr int MAX_TABLES = 8;
struct Tables { const float* p[MAX_TABLES]; int rows[MAX_TABLES]; int n; };

// source code: (table_id << 24) | row ; -1 = none (zero vector: padded position, get_embeds :176-181).  A code naming a table or a row that
// does not exist (nn.Embedding would raise IndexError) is never dereferenced: it reads as zero / is skipped and raises the error flag.
template <typename T>
__device__ __forceinline__ bool code_ok(const T& t, int code) {
    const int tb = code >> 24, row = code & 0xffffff;
    return tb < t.n && row < t.rows[tb];
}
__device__ __forceinline__ const float* src_row(const Tables& t, int code, int D) {
    return t.p[code >> 24] + (long long)(code & 0xffffff) * D;
}

__global__ __launch_bounds__(256) void embed_assemble_kernel(Tables tabs, const int* __restrict__ src_a, const int* __restrict__ src_b,
                                                             float* __restrict__ out, long long rows, int D, int* __restrict__ err) {
    const int d4 = D / 4;
    const long long total = rows * d4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d4;
        const int e = (int)(i % d4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int a = src_a[r], b = src_b[r];
        const bool a_ok = a >= 0 && code_ok(tabs, a), b_ok = b >= 0 && code_ok(tabs, b);
        if (a_ok) v = *reinterpret_cast<const float4*>(src_row(tabs, a, D) + e);
        if (b_ok) {
            const float4 w = *reinterpret_cast<const float4*>(src_row(tabs, b, D) + e);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (err && e == 0 && ((a >= 0 && !a_ok) || (b >= 0 && !b_ok))) *err = 1;
        *reinterpret_cast<float4*>(out + r * D + e) = v;
    }
}

// Scatter-add backward of the assembly.  Workgroup = 128 token rows x 256 columns; thread = one column.
//   * rows of LARGE tables (codebooks: thousands of destinations, a handful of tokens each): one L2 atomic per (token, element), as before;
//   * rows of SMALL tables (<= SMALL_ROWS destinations in total: quantizer-position / start-token embeddings, 3-5 rows that EVERY token of the
//     batch adds to): summed per workgroup in LDS first (a column belongs to one thread: plain read-modify-write), then one atomic per
//     (workgroup, destination, element).  With one atomic per token the ~16k updates of each of those few floats were serialised in the L2
//     (all of them on ~100 cache lines): 158 us for 67 MB of input; this form is bound by reading dout.
constexpr int SMALL_ROWS = 32;          // LDS: 32 x 256 floats
struct ScatterTabs { float* p[MAX_TABLES]; int rows[MAX_TABLES]; int small_base[MAX_TABLES]; int n; int nsmall; };

__global__ __launch_bounds__(256) void embed_scatter_kernel(ScatterTabs tabs, const int* __restrict__ src_a, const int* __restrict__ src_b,
                                                            const float* __restrict__ dout, float alpha, long long rows, int D) {
    __shared__ float acc[SMALL_ROWS][256];
    __shared__ int used[SMALL_ROWS];
    const int t = threadIdx.x;
    const int c = blockIdx.y * 256 + t;
    const long long r0 = (long long)blockIdx.x * 128;
    const int nr = (int)min(128LL, rows - r0);
    for (int g = 0; g < tabs.nsmall; ++g) acc[g][t] = 0.f;
    if (t < SMALL_ROWS) used[t] = 0;
    __syncthreads();
    const bool cok = c < D;
    for (int i0 = 0; i0 < nr; i0 += 8) {
        float gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) gv[u] = (cok && i0 + u < nr) ? dout[(r0 + i0 + u) * D + c] * alpha : 0.f;      // 8 loads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u >= nr) break;
            const long long r = r0 + i0 + u;
            const float g = gv[u];
            const int codes[2] = {src_a[r], src_b[r]};                   // wave-uniform
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int code = codes[k];
                if (code < 0 || !code_ok(tabs, code)) continue;
                const int tb = code >> 24, row = code & 0xffffff;
                const int sb = tabs.small_base[tb];
                if (sb >= 0) {
                    acc[sb + row][t] += g;
                    if (t == 0) used[sb + row] = 1;
                } else if (cok) {
                    atomicAdd(tabs.p[tb] + (long long)row * D + c, g);
                }
            }
        }
    }
    __syncthreads();
    if (!cok) return;
    for (int tb = 0; tb < tabs.n; ++tb) {
        const int sb = tabs.small_base[tb];
        if (sb < 0) continue;
        for (int row = 0; row < tabs.rows[tb]; ++row)
            if (used[sb + row]) atomicAdd(tabs.p[tb] + (long long)row * D + c, acc[sb + row][t]);
    }
}

// ---- the same scatter WITHOUT atomics: every destination row has ONE owner, sums are formed in a fixed order -> bitwise run-to-run deterministic
// gradients (the atomic form above adds in arrival order).  Two launches:
//   (1) embed_scatter_small_kernel: the few-row tables, per OWN_CH-token chunk, summed in LDS in token order; each chunk WRITES its sums to
//       ws[chunk][small row][D];
//   (2) embed_scatter_owned_kernel: workgroup g < ngroups OWNS OWN_G consecutive rows of one large table.  Each of its 4 waves owns a 256-column
//       slice, scans the code arrays (src_a then src_b: 4 B per token and array, L2-resident, read by every workgroup) in token order, collects the
//       tokens whose code falls in the group (ballot + prefix count: in-order compaction into a small per-wave pending list), and adds their dout
//       slices to its accumulators in that order, 8 row loads in flight.  Workgroups g >= ngroups own one (small-table row, 256-column slice) each
//       and sum the chunk partials of (1) in a fixed order.  Every row of every table is WRITTEN (zero when nothing maps to it): the caller need not clear them.
// A code array is read once per owning workgroup (~450 x 128 KB from the L2 at the headline shape) -- cheaper than serialising 16 k fp32 atomics per
// hot cache line, and there is no data-dependent sort on the host side.
constexpr int OWN_G = 8;                 // destination rows per workgroup
constexpr int OWN_PEND = 1024;           // pending-list entries per wave (flushed when fewer than one scan step's 512 slots are left)
// HOT ROWS (round 6): a destination row that very many tokens map to (skewed ids: the codes of silence, a collapsed codebook, a dominant semantic unit) made
// its ONE owner add thousands of 4 KB rows one after the other while the rest of the chip idled (half the tokens on 4 rows: 1.65 ms at 66 k tokens against
// 0.2 ms with uniform ids).  Now the first launch also HISTOGRAMS the codes of the large tables (integer atomics: the counts are exact, hence deterministic),
// its last-arriving workgroup lists the rows with more than HOT_T tokens in row order and gives each `parts` = ceil(count / HOT_PT) (2 ... HOT_PMAX) equal
// TOKEN RANGES; the second launch's first HOT_NSPLIT workgroups take (hot row, part) items: scan only their token range, sum the matching dout rows as an
// owner would, publish the partial with write-through stores, and the row's last arriver (ticket) adds the partials in part order 0, 1, 2, ... and writes the
// row.  Owners skip the rows that were listed.  Every sum is still a fixed function of the code arrays: bitwise run-to-run deterministic, no float atomics.
constexpr int HOT_T = 128;               // a row is hot above this many tokens
constexpr int HOT_PT = 256;              // target tokens per part
constexpr int HOT_PMAX = 64;             // parts per hot row at most
constexpr int HOT_MAX = 256;             // listed hot rows at most (further ones stay with their owners)
constexpr int HOT_NSPLIT = 256;          // workgroups of the second launch that take (hot row, part) items
constexpr int HOT_NY = 8;                // column blocks (of 1024) the ticket table covers: D <= 8192, wider rows keep the plain owner path
// int workspace (zeroed by the caller's memset): [0] arrival ticket of the histogram, [1] nhot, [2] total parts, [4, 4 + HOT_MAX) hot codes,
// [.., + HOT_MAX + 1) part offsets, [.., + HOT_MAX * HOT_NY) row tickets, then one count per large-table row (after listing: -(hot index + 1) for listed rows)
constexpr int IW_HOT = 4, IW_POFF = IW_HOT + HOT_MAX, IW_TICK = IW_POFF + HOT_MAX + 1, IW_CNT = (IW_TICK + HOT_MAX * HOT_NY + 3) & ~3;     // (counts: 16-byte aligned)
struct OwnTabs { float* p[MAX_TABLES]; int rows[MAX_TABLES]; int small_base[MAX_TABLES]; int grp_base[MAX_TABLES + 1]; int lrow_base[MAX_TABLES]; int n; int nsmall;
                 int nchunks; int ltot; };
__device__ __forceinline__ int hot_parts(int c) { return min(HOT_PMAX, max(2, (c + HOT_PT - 1) / HOT_PT)); }

// chunk = OWN_CH token rows x 1024 columns per workgroup (thread = 4 columns).  The chunk's codes are fetched once into LDS (the first version read them
// with two dependent scalar loads per token: a latency chain of 2 x 128 L2 round trips per workgroup, 91 us for 67 MB); the dout rows stream with 8
// float4 loads per thread in flight; accumulators: one float4 per (small row, thread) in LDS, touched by their own thread only.
// iw != NULL: the workgroups of column block 0 also count the codes of the large tables, and the last of them to arrive lists the hot rows (see above).
constexpr int OWN_CH = 32;
__global__ __launch_bounds__(256) void embed_scatter_small_kernel(OwnTabs tabs, const int* __restrict__ src_a, const int* __restrict__ src_b,
                                                                  const float* __restrict__ dout, float alpha, long long rows, int D, float* __restrict__ ws,
                                                                  int* __restrict__ iw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char own_smem[];
    int* codes = reinterpret_cast<int*>(own_smem);                          // [2][OWN_CH]: slot of the small row (>= 0) or -1
    float4* acc = reinterpret_cast<float4*>(own_smem + 2 * OWN_CH * sizeof(int));   // [nsmall][256]
    __shared__ int hs_hot[256], hs_parts[256], hs_flag;
    const int t = threadIdx.x;
    const int col = blockIdx.y * 1024 + t * 4;
    const bool cok = col < D;
    const int colc = cok ? col : 0;
    const long long r0 = (long long)blockIdx.x * OWN_CH;
    const int nr = (int)min((long long)OWN_CH, rows - r0);
    const bool hist = iw != nullptr && blockIdx.y == 0;
    if (t < 2 * OWN_CH) {
        const int k = t / OWN_CH, i = t % OWN_CH;
        int slot = -1;
        if (i < nr) {
            const int code = (k == 0 ? src_a : src_b)[r0 + i];
            if (code >= 0 && code_ok(tabs, code)) {
                const int sb = tabs.small_base[code >> 24];
                if (sb >= 0) slot = sb + (code & 0xffffff);
                else if (hist) __hip_atomic_fetch_add(iw + IW_CNT + tabs.lrow_base[code >> 24] + (code & 0xffffff), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        codes[t] = slot;
    }
    for (int g = 0; g < tabs.nsmall; ++g) acc[g * 256 + t] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tabs.nsmall > 0) {
        const float* dp = dout + r0 * D + colc;
#pragma unroll 1
        for (int i0 = 0; i0 < nr; i0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(dp + (long long)min(i0 + u, nr - 1) * D);
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));      // all 8 in flight (see the owned kernel)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u >= nr) break;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int slot = codes[k * OWN_CH + i0 + u];               // broadcast read, workgroup-uniform
                    if (slot < 0) continue;
                    float4 a = acc[slot * 256 + t];
                    a.x += v[u].x * alpha; a.y += v[u].y * alpha; a.z += v[u].z * alpha; a.w += v[u].w * alpha;
                    acc[slot * 256 + t] = a;
                }
            }
        }
        if (cok)
            for (int g = 0; g < tabs.nsmall; ++g) *reinterpret_cast<float4*>(ws + ((long long)blockIdx.x * tabs.nsmall + g) * D + col) = acc[g * 256 + t];
    }
    if (!hist) return;                                                      // (workgroup-uniform)
    // ---- the histogram's last arriver lists the hot rows.  The counting atomics above are agent-scope read-modify-writes; every wave drains its own, the
    // barrier collects the waves, one lane takes the ticket (the in-launch split-K idiom of gemm.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) hs_flag = __hip_atomic_fetch_add(iw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!hs_flag) return;
    // counts: KC coalesced 16-byte loads per thread in flight (sc1: served by the fabric, never by a stale L2 line), thread t holds rows kb * 1024 * KC +
    // u * 1024 + 4 t + {0..3}.  List order = (thread, batch, load, element): a fixed function of the counts, which is all determinism needs.  (The first
    // version walked 48 consecutive rows per thread with dependent atomic loads: +36 us on the 12 291-row table.)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int KC = 8;
    int* const cnt = iw + IW_CNT;
    const int lpad = (tabs.ltot + 3) & ~3;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(cnt, 0, lpad * 4, 0x00020000);        // (past the end: zeros = not hot)
    const int nb = (lpad + 1024 * KC - 1) / (1024 * KC);
    int nh = 0, npt = 0;
#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
        u32x4 v[KC];
#pragma unroll
        for (int u = 0; u < KC; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((kb * KC + u) * 1024 + t * 4) * 4, 0, 16);
#pragma unroll
        for (int u = 0; u < KC; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (int)v[u][e];
                if (c > HOT_T) { ++nh; npt += hot_parts(c); }
            }
    }
    hs_hot[t] = nh; hs_parts[t] = npt;
    __syncthreads();
    if (t < 64) {                                                           // exclusive prefix over the 256 threads' (rows, parts): wave 0, 4 entries per lane
        int x[4], y[4], sx = 0, sy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = hs_hot[4 * t + k]; y[k] = hs_parts[4 * t + k]; sx += x[k]; sy += y[k]; }
        int ix = sx, iy = sy;                                               // inclusive scan of the lane sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ux = __shfl_up(ix, d), uy = __shfl_up(iy, d);
            if (t >= d) { ix += ux; iy += uy; }
        }
        int a = ix - sx, b = iy - sy;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hs_hot[4 * t + k] = a; hs_parts[4 * t + k] = b; a += x[k]; b += y[k]; }
        if (t == 63) {                                                      // (a, b) = the totals
            const int listed = min(a, HOT_MAX);
            iw[1] = listed;
            if (a <= HOT_MAX) { iw[2] = b; iw[IW_POFF + listed] = b; }       // (a > HOT_MAX: the thread that lists entry HOT_MAX - 1 writes the totals)
        }
    }
    __syncthreads();
    int h = hs_hot[t], po = hs_parts[t];
    if (nh == 0 || h >= HOT_MAX) return;                                    // (no barrier below)
#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
        u32x4 v[KC];
#pragma unroll
        for (int u = 0; u < KC; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((kb * KC + u) * 1024 + t * 4) * 4, 0, 16);
#pragma unroll
        for (int u = 0; u < KC; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                   // (unrolled: v[] stays in registers; the body runs for hot rows only)
                const int c = (int)v[u][e];
                if (c <= HOT_T || h >= HOT_MAX) continue;
                const int i = (kb * KC + u) * 1024 + t * 4 + e;
                int tb = 0;
                while (tb + 1 < tabs.n && !(tabs.lrow_base[tb] >= 0 && i >= tabs.lrow_base[tb] && i < tabs.lrow_base[tb] + tabs.rows[tb])) ++tb;
                iw[IW_HOT + h] = (tb << 24) | (i - tabs.lrow_base[tb]);
                iw[IW_POFF + h] = po;
                cnt[i] = -(h + 1);                                          // listed: its owner skips it (read by the NEXT launch)
                po += hot_parts(c);
                if (++h == HOT_MAX) { iw[2] = po; iw[IW_POFF + HOT_MAX] = po; }
            }
    }
}

__global__ __launch_bounds__(256) void embed_scatter_owned_kernel(OwnTabs tabs, const int* __restrict__ src_a, const int* __restrict__ src_b,
                                                                  const float* __restrict__ dout, float alpha, long long rows, int D,
                                                                  const float* __restrict__ ws, int ngroups, int nsplit, int* __restrict__ iw,
                                                                  float* __restrict__ hp) {
    __shared__ int pend[4][OWN_PEND];
    __shared__ int pcnt[4];
    __shared__ int hflag;
    __shared__ float4 accs[4][OWN_G][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = (blockIdx.y * 4 + wave) * 256 + lane * 4;
    const bool cok = col < D;                                              // D % 4 == 0: a lane's 4 columns are in or out together
    const int colc = cok ? col : 0;
    const int g = (int)blockIdx.x - nsplit;                                 // < 0: a (hot row, part) worker; < ngroups: an owner; else a small-row reducer
    if (g >= ngroups) {
        // ---- one small-table row x one 256-column slice: the 4 waves take every 4th chunk partial each (16 loads in flight), then their sums are
        // combined in wave order -- a fixed association, like everything else here
        const int s = (g - ngroups) >> 2, slice = (g - ngroups) & 3;
        int tb = 0;
        while (tb + 1 < tabs.n && !(tabs.small_base[tb] >= 0 && s >= tabs.small_base[tb] && s < tabs.small_base[tb] + tabs.rows[tb])) ++tb;
        const int scol = (blockIdx.y * 4 + slice) * 256 + lane * 4;
        const bool sok = scol < D;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wp = ws + (long long)s * D + (sok ? scol : 0);
        const long long cs = (long long)tabs.nsmall * D;
        const int nch = tabs.nchunks;
#pragma unroll 1
        for (int ch = wave; ch < nch; ch += 64) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(wp + (long long)min(ch + 4 * u, nch - 1) * cs);
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (ch + 4 * u < nch) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        accs[wave][0][lane] = a;
        __syncthreads();
        if (wave == 0 && sok) {
            const float4 b1 = accs[1][0][lane], b2 = accs[2][0][lane], b3 = accs[3][0][lane];
            a.x = ((a.x + b1.x) + b2.x) + b3.x; a.y = ((a.y + b1.y) + b2.y) + b3.y; a.z = ((a.z + b1.z) + b2.z) + b3.z; a.w = ((a.w + b1.w) + b2.w) + b3.w;
            *reinterpret_cast<float4*>(tabs.p[tb] + (long long)(s - tabs.small_base[tb]) * D + scol) = a;
        }
        return;
    }
    // accumulators: one float4 per (destination row, lane) in LDS, a lane only ever touches its own slots (no conflicts, no barrier); register
    // accumulators would need a select chain per (entry, row) -- 17 k lines of ISA in the first version, which ran out of the instruction cache
    float4* myacc = &accs[wave][0][lane];
    // Round 6: the SCAN is shared by the workgroup.  Until then every one of the 4 waves (one per 256-column slice) scanned both code arrays itself: 4 x the
    // scan work, and the scan is what the kernel costs on a large table -- codebook 4096 (12 291 rows = 1 537 workgroups) x 66 k tokens: 3.0 ms per step of
    // `e2e_config5`, against 44 us at the headline shape.  Now wave w scans every 4th 512-code block, collects ITS hits (ballot + prefix count: in-order
    // compaction) in its own list pend[w]; after each round of 4 blocks a barrier, and when some list could overflow in the next round (or at the end) ALL waves
    // add the entries of list 0, 1, 2, 3 -- in that fixed order -- to their own column slice, 8 row loads in flight.  The order of the additions is a fixed
    // function of the code arrays (not token order any more, but the same in every run): bitwise deterministic as before.
    // AU row loads in flight per lane (16: a hot destination row -- skewed token ids, e.g. the codes of silence -- makes ONE workgroup add thousands of 4 KB
    // rows one after the other; the memory-level parallelism of that chain is AU x 16 B per lane)
    constexpr int AU = 16;
    auto add_list = [&](const int* __restrict__ pl, int np) {
#pragma unroll 1
        for (int i0 = 0; i0 < np; i0 += AU) {
            float4 v[AU];
            int d[AU];
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int e = pl[min(i0 + u, np - 1)];                     // broadcast LDS read: (local row << 28) | token row
                d[u] = i0 + u < np ? (e >> 28) & 7 : -1;
                const long long r = e & 0x0fffffff;
                v[u] = *reinterpret_cast<const float4*>(dout + r * D + colc);   // unconditional: AU row loads in flight (colc: clamped column)
            }
            // pin the loaded vectors here: without a use at this point the compiler sinks each load into "its" iteration of the loop below
            // (load, s_waitcnt vmcnt(0), add -- one HBM round trip per entry)
#pragma unroll
            for (int u = 0; u < AU; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                if (d[u] < 0) break;                                       // wave-uniform
                float4 a = myacc[d[u] * 64];
                a.x += v[u].x * alpha; a.y += v[u].y * alpha; a.z += v[u].z * alpha; a.w += v[u].w * alpha;
                myacc[d[u] * 64] = a;
            }
        }
    };
    // the scan of token rows [t0, t1) for codes code_lo + [0, nrow) whose bit in `skip` is clear: 8 coalesced dword loads per lane in flight (512 codes per
    // block), the next block's loads issued before this one is processed.  The loads are UNCONDITIONAL (index clamped to the last row, validity applied to the
    // value afterwards): a per-lane `i < rows ? load : -1` compiles to an exec-masked branch per load with a full s_waitcnt behind each (one L2 round trip per
    // 64 codes: ~200 us for 32 k codes)
    constexpr int SU = 8, BLK = SU * 64;
    const int last = (int)rows - 1;
    auto scan_add = [&](const int code_lo, const int nrow, const unsigned skip, const int t0, const int t1) {
        int* pl = pend[wave];
        int np = 0;                                                        // wave-uniform: entries in this wave's list
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int* __restrict__ arr = pass == 0 ? src_a : src_b;
            auto fetch = [&](int base, int (&c)[SU]) {
#pragma unroll
                for (int u = 0; u < SU; ++u) c[u] = arr[min(base + u * 64 + lane, last)];
            };
            int cur[SU], nxt[SU];
            fetch(t0 + wave * BLK, cur);
#pragma unroll 1
            for (int base0 = t0; base0 < t1; base0 += 4 * BLK) {            // one round: blocks base0 + {0, 1, 2, 3} * BLK, one per wave
                const int base = base0 + wave * BLK;
                fetch(base + 4 * BLK, nxt);                                 // past the end: the clamped last code, discarded below
                bool any = false;
#pragma unroll
                for (int u = 0; u < SU; ++u) any |= (unsigned)(cur[u] - code_lo) < (unsigned)nrow;    // (a negative code gives a huge unsigned offset)
                if (__ballot(any)) {                                        // most blocks hold no token of this group: one ballot instead of eight
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int i = base + u * 64 + lane;
                        const unsigned off = (unsigned)(cur[u] - code_lo);
                        const bool hit = i < t1 && cur[u] >= 0 && off < (unsigned)nrow && !((skip >> (off & 7)) & 1u);
                        const unsigned long long m = __ballot(hit);
                        const int pos = np + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        if (hit) pl[pos] = (int)((off << 28) | (unsigned)i);
                        np += __popcll(m);
                    }
                }
                const bool final_round = pass == 1 && base0 + 4 * BLK >= t1;
                if (lane == 0) pcnt[wave] = np;
                __syncthreads();                                            // the four lists and their lengths are visible
                const int n0 = pcnt[0], n1 = pcnt[1], n2 = pcnt[2], n3 = pcnt[3];
                const bool flush = final_round || max(max(n0, n1), max(n2, n3)) > OWN_PEND - BLK;       // (a round adds at most BLK entries per list); uniform
                if (flush) {
                    add_list(pend[0], n0); add_list(pend[1], n1); add_list(pend[2], n2); add_list(pend[3], n3);
                    np = 0;
                }
                __syncthreads();                                            // nobody overwrites a list (or a count) that is still being read
#pragma unroll
                for (int u = 0; u < SU; ++u) cur[u] = nxt[u];
            }
        }
    };
    if (g < 0) {
        // ---- (hot row, part) items: item = poff[h] + part.  The partial goes out with write-through stores (sc1), every wave drains them, one lane takes the
        // row's ticket; the last arriver re-reads ALL partials of the row with sc1 loads and adds them in part order
        const int nparts = iw[2], nhot = iw[1];
#pragma unroll 1
        for (int item = (int)blockIdx.x; item < nparts; item += nsplit) {
            int lo = 0, hi = nhot - 1;                                      // largest h with poff[h] <= item
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (iw[IW_POFF + mid] <= item) lo = mid; else hi = mid - 1; }
            const int h = lo, p0 = iw[IW_POFF + h], P = iw[IW_POFF + h + 1] - p0, part = item - p0;
            const int code = iw[IW_HOT + h];
            const int t0 = (int)((long long)rows * part / P), t1 = (int)((long long)rows * (part + 1) / P);
            myacc[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            scan_add(code, 1, 0u, t0, t1);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(hp + (long long)p0 * D, 0, (int)((long long)P * D * 4), 0x00020000);
            if (cok) {
                const float4 a = myacc[0];
                const u32x4 v = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (part * D + col) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                hflag = __hip_atomic_fetch_add(iw + IW_TICK + h * HOT_NY + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P - 1;
            __syncthreads();
            if (hflag && cok) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
                for (int q0 = 0; q0 < P; q0 += 16) {
                    u32x4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (min(q0 + u, P - 1) * D + col) * 4, 0, 16);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (q0 + u < P) { a.x += __uint_as_float(v[u].x); a.y += __uint_as_float(v[u].y); a.z += __uint_as_float(v[u].z); a.w += __uint_as_float(v[u].w); }
                }
                *reinterpret_cast<float4*>(tabs.p[code >> 24] + (long long)(code & 0xffffff) * D + col) = a;
            }
            __syncthreads();                                                // hflag is rewritten by the next item
        }
        return;
    }
    // ---- OWN_G rows of a large table
    int tb = 0;
    while (tb + 1 < tabs.n && g >= tabs.grp_base[tb + 1]) ++tb;
    const int row0 = (g - tabs.grp_base[tb]) * OWN_G;
    const int nrow = min(OWN_G, tabs.rows[tb] - row0);
    const int code_lo = (tb << 24) | row0;
    unsigned skip = 0;                                                     // rows of this group that (hot row, part) workers take
    if (iw != nullptr)
        for (int k = 0; k < nrow; ++k) skip |= (iw[IW_CNT + tabs.lrow_base[tb] + row0 + k] < 0 ? 1u : 0u) << k;
    if (skip == (1u << nrow) - 1u) return;                                  // (uniform) nothing left to own
#pragma unroll
    for (int k = 0; k < OWN_G; ++k) myacc[k * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
    scan_add(code_lo, nrow, skip, 0, (int)rows);
    if (!cok) return;
    for (int k = 0; k < nrow; ++k)
        if (!((skip >> k) & 1u)) *reinterpret_cast<float4*>(tabs.p[tb] + (long long)(row0 + k) * D + col) = myacc[k * 64];
}

// out[r] = in[idx[r]] (idx < 0 -> zero row)           bf16 rows, D % 8 == 0
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ in, long long ld_in, const int* __restrict__ idx,
                                                          bf16_t* __restrict__ out, long long ld_out, long long rows, int D) {
    const int d8 = D / 8;
    const long long total = rows * d8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d8;
        const int e = (int)(i % d8) * 8;
        const int s = idx[r];
        uint4 v = make_uint4(0, 0, 0, 0);
        if (s >= 0) v = *reinterpret_cast<const uint4*>(in + (long long)s * ld_in + e);
        *reinterpret_cast<uint4*>(out + r * ld_out + e) = v;
    }
}
// hi[r] = bf16(x), lo[r] = bf16(x - hi) for x = in[idx ? idx[r] : r]; source rows outside [0, rows_in) give zero rows.  D % 4 == 0
__global__ __launch_bounds__(256) void gather_split_kernel(const float* __restrict__ in, long long ld_in, long long rows_in, const int* __restrict__ idx,
                                                           bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long long ld_out, long long rows, int D) {
    const int d4 = D / 4;
    const long long total = rows * d4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d4;
        const int e = (int)(i % d4) * 4;
        const long long s = idx ? (long long)idx[r] : r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s >= 0 && s < rows_in) v = *reinterpret_cast<const float4*>(in + s * ld_in + e);
        const uint32_t h0 = pack_bf2(v.x, v.y), h1 = pack_bf2(v.z, v.w);
        const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
        const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
        *reinterpret_cast<uint2*>(hi + r * ld_out + e) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lo + r * ld_out + e) = make_uint2(pack_bf2(r0, r1), pack_bf2(r2, r3));
    }
}
// out[idx[r]] = in[r] (idx < 0 skipped; idx must be injective; untouched rows keep their previous contents)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ in, long long ld_in, const int* __restrict__ idx,
                                                           bf16_t* __restrict__ out, long long ld_out, long long rows, int D) {
    const int d8 = D / 8;
    const long long total = rows * d8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d8;
        const int e = (int)(i % d8) * 8;
        const int s = idx[r];
        if (s >= 0) *reinterpret_cast<uint4*>(out + (long long)s * ld_out + e) = *reinterpret_cast<const uint4*>(in + r * ld_in + e);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// cross entropy: one wave per row
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long long ld, const long long* __restrict__ labels,
                                                     float* __restrict__ loss, float* __restrict__ lse_out, long long rows, int C, int ignore) {
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const float* lp = logits + r * ld;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lp[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += __expf(lp[c] - mx);
        s = wave_sum(s);
        const float lse = mx + logf(s);
        if (lane == 0) {
            const long long lab = labels[r];
            lse_out[r] = lse;
            loss[r] = (lab == ignore || lab < 0 || lab >= C) ? 0.f : (lse - lp[lab]);
        }
    }
}

// dlogits[r][c] = (exp(logit - lse) - [c == label]) * gscale[0]   (ignored rows -> 0; pad columns [C, Cpad) -> 0)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long long ld, const long long* __restrict__ labels,
                                                     const float* __restrict__ lse_in, const float* __restrict__ gscale,
                                                     bf16_t* __restrict__ dlogits, long long ldd, long long rows, int C, int Cpad, int ignore) {
    const int lane = threadIdx.x & 63;
    const float gs = *gscale;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const long long lab = labels[r];
        const bool ign = (lab == ignore || lab < 0 || lab >= C);
        const float lse = lse_in[r];
        const float* lp = logits + r * ld;
        bf16_t* dp = dlogits + r * ldd;
        for (int c = lane; c < Cpad; c += 64) {
            float v = 0.f;
            if (!ign && c < C) v = (__expf(lp[c] - lse) - (c == lab ? 1.f : 0.f)) * gs;
            dp[c] = f2bf(v);
        }
    }
}

__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ in, long long n, float* __restrict__ out, float scale) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) s += in[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = t * scale;
    }
}

// vmix[m][0:64] = 0.5 * (v[m] + v0[m])     (bf16, 8 lanes per row)
__global__ __launch_bounds__(256) void vmix_kernel(const bf16_t* __restrict__ v, long long ldv, const bf16_t* __restrict__ v0, long long ldv0,
                                                   bf16_t* __restrict__ out, long long ldo, long long rows, int dh) {
    const int c8 = dh / 8;
    const long long total = rows * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c8;
        const int e = (int)(i % c8) * 8;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(v + r * ldv + e);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(v0 + r * ldv0 + e);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(0.5f * (bf2f((bf16_t)a[j]) + bf2f((bf16_t)b[j])));
        *reinterpret_cast<bf16x8*>(out + r * ldo + e) = o;
    }
}

// gradient fan-in of the value residual.  dk, dv: `nparts` fp32 partials [rows][ld] (part_stride floats apart; one per head group)
// from the attention backward, summed here (dv is the grad wrt the MIXED v).
//   mode 0 (no value residual) : dkv = (dk, dv)
//   mode 1 (layer >= 1)        : dkv = (dk, 0.5 dv)          ; acc_v0 += 0.5 dv
//   mode 2 (layer 0)           : dkv = (dk, dv + acc_v0)
__global__ __launch_bounds__(256) void kv_grad_pack_kernel(const float* __restrict__ dk, const float* __restrict__ dv, long long ld, int nparts,
                                                           long long part_stride, float* __restrict__ acc_v0, bf16_t* __restrict__ dkv,
                                                           long long ldo, long long rows, int dh, int mode) {
    const long long total = rows * dh;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / dh;
        const int e = (int)(i % dh);
        float gk = 0.f, gv = 0.f;
        for (int pt = 0; pt < nparts; ++pt) {
            gk += dk[pt * part_stride + r * ld + e];
            gv += dv[pt * part_stride + r * ld + e];
        }
        if (mode == 1) {
            gv *= 0.5f;
            acc_v0[i] += gv;
        } else if (mode == 2) {
            gv += acc_v0[i];
        }
        dkv[r * ldo + e] = f2bf(gk);
        dkv[r * ldo + dh + e] = f2bf(gv);
    }
}

int grid_for(long long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

// ------------------------------------------------------------------------------------------------------------------
// forgetful causal mask (reference audiolm_pytorch.py:82-89: `rand[:, 0] = -max; idx = rand.topk(k).indices; mask = ~zeros.scatter(1, idx, 1)`):
// per row, the k keys with the largest Gaussian draw are dropped, key 0 never.  ONE workgroup per row replaces ATen's topk (radix select + sort) +
// scatter + fills + and: the k-th largest score is found by bisection over the order-preserving integer image of the floats (32 rounds of
// count-and-compare; a thread owns a contiguous run of the row), equal scores at the threshold go in index order.  keep[b][i] &= !dropped.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FM_MAX_PER_THREAD = 64;                                          // rows up to 256 * 64 = 16384 keys

__device__ __forceinline__ uint32_t fm_key(float f) {                          // monotone float -> uint32 (larger float <=> larger key); NaN sorts high
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int PER>                                                             // keys per thread (compile-time: the row lives in registers)
__global__ __launch_bounds__(256) void forgetful_mask_kernel(const float* __restrict__ score, long long ld_score, unsigned char* __restrict__ keep,
                                                             long long ld_keep, int N, int drop) {
    __shared__ int wsum[2][4];
    __shared__ int tcnt[256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = t * PER;
    const float* row = score + (long long)blockIdx.x * ld_score;
    unsigned char* krow = keep + (long long)blockIdx.x * ld_keep;
    uint32_t key[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = i0 + j;
        key[j] = (i < N && i > 0) ? fm_key(row[i]) : 0u;                       // key 0 = "never dropped" (below every real score): column 0 and the padding
    }
    int par = 0;
    auto block_count = [&](int c) {
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) wsum[par][wave] = c;
        __syncthreads();
        const int tot = wsum[par][0] + wsum[par][1] + wsum[par][2] + wsum[par][3];
        par ^= 1;                                                              // double-buffered: one barrier per count
        return tot;
    };
    uint32_t prefix = 0u;                                                      // largest value v with  #(key >= v) >= drop  = the drop-th largest key
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) c += key[j] >= cand;
        if (block_count(c) >= drop) prefix = cand;
    }
    if (prefix == 0u) return;                                                  // fewer real keys than `drop` cannot happen (drop <= N - 1); defensive
    // everything above the threshold goes; of the keys EQUAL to it, the first (drop - #above) in index order (one key unless two draws coincide)
    int above = 0, equal = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { above += key[j] > prefix; equal += key[j] == prefix; }
    const int n_above = block_count(above), n_equal = block_count(equal);
    int quota = equal;
    if (n_above + n_equal > drop) {                                            // a tie at the threshold: a thread owns a contiguous run -> thread order = index order
        tcnt[t] = equal;
        __syncthreads();
        int before = 0;
        for (int u = 0; u < t; ++u) before += tcnt[u];
        quota = drop - n_above - before;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = i0 + j;
        if (i < N) {
            bool gone = key[j] > prefix;
            if (key[j] == prefix && quota > 0) { gone = true; --quota; }
            if (gone) krow[i] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// loss = sum_g w_g * sum_g / max(#(labels_g != ignore), 1): the cross-entropy means of up to 4 head groups and the wrappers' weighted combination
// (audiolm_pytorch.py:1561-1565, :1826-1854, :2112-2137) in ONE launch instead of ne / sum / clamp / div per group + mul / add / div.
// scale_g = w_g / max(count_g, 1) is kept for the backward (d sum_g = d loss * scale_g).
// ------------------------------------------------------------------------------------------------------------------
struct LossGroups { const float* sum[4]; const long long* labels[4]; long long n[4]; float w[4]; int G; long long ignore; };

__global__ __launch_bounds__(1024) void loss_combine_kernel(LossGroups g, float* __restrict__ loss, float* __restrict__ scales) {
    __shared__ int red[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < g.G; ++k) {
        int c = 0;
        for (long long i = threadIdx.x; i < g.n[k]; i += 1024) c += g.labels[k][i] != g.ignore;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) red[k][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int k = 0; k < g.G; ++k) {
            int c = 0;
            for (int w2 = 0; w2 < 16; ++w2) c += red[k][w2];
            const float sc = g.w[k] / (float)(c > 1 ? c : 1);
            scales[k] = sc;
            tot += sc * g.sum[k][0];
        }
        loss[0] = tot;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// batch_unique_consecutive (reference audiolm_pytorch.py:162-164: per row torch.unique_consecutive, rows right-padded to the longest) as ONE launch and ONE
// host read instead of a host loop over the batch with a device synchronisation per row.  Workgroup = one row: keep[i] = (i == 0 || id[i] != id[i - 1]),
// in-order compaction (ballot + prefix count per wave, wave totals through LDS), the tail of the row filled with pad, lengths[b] = kept ids.  append_eos:
// the row is [ids | eos_id] (append_eos_id :155-160 happens BEFORE the collapse in the wrappers, :1536-1539 / :1788-1795).  out int64 [B][n + append_eos].
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unique_consecutive_kernel(const long long* __restrict__ ids, long long ld, int n, int append_eos, long long eos_id,
                                                                 long long pad, long long* __restrict__ out, long long ld_out, int* __restrict__ lengths) {
    __shared__ int wsum[2][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int W = n + (append_eos ? 1 : 0);
    const long long* row = ids + (long long)blockIdx.x * ld;
    long long* orow = out + (long long)blockIdx.x * ld_out;
    auto at = [&](int i) -> long long { return i < n ? row[i] : eos_id; };
    int base = 0, buf = 0;
    for (int i0 = 0; i0 < W; i0 += 256, buf ^= 1) {
        const int i = i0 + t;
        const long long v = i < W ? at(i) : 0;
        const bool keep = i < W && (i == 0 || v != at(i - 1));
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[buf][wave] = __popcll(m);
        __syncthreads();                                                    // (two buffers: the next round's totals do not overwrite the ones still being read)
        int off = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        for (int w = 0; w < wave; ++w) off += wsum[buf][w];
        if (keep) orow[off] = v;
        base += wsum[buf][0] + wsum[buf][1] + wsum[buf][2] + wsum[buf][3];
    }
    for (int i = base + t; i < W; i += 256) orow[i] = pad;
    if (t == 0) lengths[blockIdx.x] = base;
}

// ------------------------------------------------------------------------------------------------------------------
// CoarseTransformerWrapper.forward's id bookkeeping for a training step (audiolm_pytorch.py:1785-1810 + the code arithmetic of :894-918) in ONE launch:
// append the eos ids, key mask of the semantic ids (pad / eos keys are masked and their ids zeroed), padded mask over the whole sequence, embedding
// source codes (start tokens, semantic ids, per-quantizer offset coarse rows) and the two label tensors -- ~18 ATen launches of <= 5 us otherwise.
//   N = 1 + (ns0 + 1) + 1 + nc0:  [start | semantic ids + eos | coarse start | coarse ids]      (the appended coarse eos is a label only)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coarse_prepare_kernel(const long long* __restrict__ sem, long long ld_sem, const long long* __restrict__ coarse,
                                                             long long ld_coarse, int B, int ns0, int nc0, long long pad_id, long long sem_eos,
                                                             long long coarse_eos, int Q, int C, long long* __restrict__ sem_labels,
                                                             long long* __restrict__ coarse_labels, int* __restrict__ src_a,
                                                             unsigned char* __restrict__ keep, int sem_has_eos) {
    // sem_has_eos: the semantic rows already are [ids | eos | pad ...] (alm_unique_consecutive_i64 with the eos appended): nothing is appended here
    const int ns = ns0 + (sem_has_eos ? 0 : 1), N = ns + nc0 + 2, W = max(N, nc0 + 1);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)B * W; t += (long long)gridDim.x * 256) {
        const int b = (int)(t / W), i = (int)(t % W);
        if (i < N) {
            int code;
            bool k = true;
            if (i == 0) {
                code = 3 << 24;                                                // semantic_start_token (table 3)
            } else if (i <= ns) {
                const int j = i - 1;
                const long long v = j < ns0 ? sem[(long long)b * ld_sem + j] : sem_eos;
                sem_labels[(long long)b * ns + j] = v;
                k = v != pad_id && v != sem_eos;                               // :1801
                const long long c = k ? v : 0;                                 // masked_fill(~mask, 0)
                code = c < -1 ? -1 : (int)c;                                   // table 0; a negative id is the zero vector (:901)
            } else if (i == ns + 1) {
                code = 4 << 24;                                                // coarse_start_token (table 4)
            } else {
                const int kx = i - (ns + 2);
                code = (int)coarse[(long long)b * ld_coarse + kx] + (kx % Q) * C + (1 << 24);      // :896-899 (stride C: eos aliasing kept)
            }
            src_a[(long long)b * N + i] = code;
            keep[(long long)b * N + i] = k ? 1 : 0;
        }
        if (i <= nc0) coarse_labels[(long long)b * (nc0 + 1) + i] = i < nc0 ? coarse[(long long)b * ld_coarse + i] : coarse_eos;
    }
}

// SemanticTransformerWrapper.forward's id bookkeeping of a training step (audiolm_pytorch.py:1536-1548: eos appended, the input ids = the labels without their
// last position) + the embedding source codes of SemanticTransformer.forward (:709-714: [start token | ids], a negative id = the zero vector, :176-181) in ONE
// launch -- the cat / ones / cast / cat chain of ~6 ATen launches otherwise.  labels int64 [B][n0 + 1], src_a int32 [B][n0 + 1].
__global__ __launch_bounds__(256) void semantic_prepare_kernel(const long long* __restrict__ sem, long long ld_sem, int B, int n0, long long eos_id,
                                                               long long* __restrict__ labels, int* __restrict__ src_a, int has_eos) {
    const int W = n0 + (has_eos ? 0 : 1);                                      // has_eos: the rows already are [ids | eos | pad ...]: labels = the rows
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)B * W; t += (long long)gridDim.x * 256) {
        const int b = (int)(t / W), i = (int)(t % W);
        labels[t] = i < n0 ? sem[(long long)b * ld_sem + i] : eos_id;
        int code = 1 << 24;                                                    // start_token (table 1)
        if (i > 0) {
            const long long v = sem[(long long)b * ld_sem + i - 1];            // input position i = id i - 1 (the appended eos is a label only)
            code = v < 0 ? -1 : (int)v;                                        // table 0
        }
        src_a[t] = code;
    }
}

// FineTransformer.forward's id bookkeeping (audiolm_pytorch.py:1171-1223) in ONE launch: key mask of the coarse ids (pad / eos keys are masked and their ids
// zeroed, :1175-1177), the mask padded over [coarse start | coarse | fine start | fine] (:1179) and the embedding source codes (start tokens, per-quantizer
// offset rows id + (i mod Q) * codebook_size of the two tables, :1186-1223) -- ~12 ATen launches otherwise.  N = 1 + n + 1 + nf.
__global__ __launch_bounds__(256) void fine_prepare_kernel(const long long* __restrict__ coarse, long long ld_coarse, const long long* __restrict__ fine,
                                                           long long ld_fine, int B, int n, int nf, long long pad_id, long long eos_id, int Qc, int Qf, int C,
                                                           int* __restrict__ src_a, unsigned char* __restrict__ keep) {
    const int N = n + nf + 2;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)B * N; t += (long long)gridDim.x * 256) {
        const int b = (int)(t / N), i = (int)(t % N);
        int code;
        bool k = true;
        if (i == 0) {
            code = 4 << 24;                                                    // coarse_start_token (table 4)
        } else if (i <= n) {
            const int j = i - 1;
            const long long v = coarse[(long long)b * ld_coarse + j];
            k = v != pad_id && v != eos_id;                                    // :1175
            code = (int)(k ? v : 0) + (j % Qc) * C;                            // masked_fill(~mask, 0) + quantizer offset, table 0
        } else if (i == n + 1) {
            code = 5 << 24;                                                    // fine_start_token (table 5)
        } else {
            const int j = i - (n + 2);
            code = (int)fine[(long long)b * ld_fine + j] + (j % Qf) * C + (1 << 24);       // table 1
        }
        src_a[t] = code;
        keep[t] = k ? 1 : 0;
    }
}

}  // namespace

extern "C" int alm_embed_assemble(const float* const* tables, const int* table_rows, int ntables, const int* src_a, const int* src_b, float* out,
                                  long long rows, int D, int* err_flag, void* stream) {
    if (ntables > MAX_TABLES || (D & 3) || !table_rows) return ALM_ERR_BAD_ARG;
    Tables t{};
    t.n = ntables;
    for (int i = 0; i < ntables; ++i) { t.p[i] = tables[i]; t.rows[i] = table_rows[i]; }
    hipLaunchKernelGGL(embed_assemble_kernel, dim3(grid_for(rows * (D / 4))), dim3(256), 0, (hipStream_t)stream, t, src_a, src_b, out, rows, D, err_flag);
    ALM_LAUNCH_CHECK();
    return 0;
}

// grad tables must be zero-initialised (or hold a running gradient); dout fp32 [rows][D]; alpha = grad_shrink factor
extern "C" int alm_embed_scatter_add(float* const* grad_tables, const int* table_rows, int ntables, const int* src_a, const int* src_b,
                                     const float* dout, float alpha, long long rows, int D, void* stream) {
    if (ntables > MAX_TABLES || !table_rows) return ALM_ERR_BAD_ARG;
    ScatterTabs t{};
    t.n = ntables;
    for (int i = 0; i < ntables; ++i) {
        t.p[i] = grad_tables[i];
        t.rows[i] = table_rows[i];
        t.small_base[i] = -1;
        if (table_rows[i] > 0 && t.nsmall + table_rows[i] <= SMALL_ROWS) { t.small_base[i] = t.nsmall; t.nsmall += table_rows[i]; }
    }
    if (rows <= 0 || D <= 0) return 0;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)((rows + 127) / 128), (unsigned)((D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, src_a, src_b, dout,
                       alpha, rows, D);
    ALM_LAUNCH_CHECK();
    return 0;
}

// Deterministic form of the scatter (no float atomics; every row of every table is written).  ws: alm_embed_scatter_ws_floats(...) fp32 floats:
// [chunk partials of the few-row tables][partials of the (hot row, part) items][the int workspace of the hot-row listing].
// Limits: rows < 2^28 (the pending entries pack a 3-bit local row + a 28-bit token row), D % 4 == 0, 16-byte aligned rows.
__global__ __launch_bounds__(256) void zero_ints_kernel(int4* __restrict__ p, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_int4(0, 0, 0, 0);
}

struct OwnPlan { long long small_fl, hot_fl, int_fl; bool hot; };
static int own_tabs(OwnTabs& t, OwnPlan& pl, float* const* grad_tables, const int* table_rows, int ntables, long long rows, int D) {
    if (ntables > MAX_TABLES || !table_rows) return ALM_ERR_BAD_ARG;
    t = OwnTabs{};
    t.n = ntables;
    int ng = 0;
    for (int i = 0; i < ntables; ++i) {
        t.p[i] = grad_tables ? grad_tables[i] : nullptr;
        t.rows[i] = table_rows[i];
        t.small_base[i] = -1;
        t.lrow_base[i] = -1;
        t.grp_base[i] = ng;
        if (table_rows[i] > 0 && t.nsmall + table_rows[i] <= SMALL_ROWS) { t.small_base[i] = t.nsmall; t.nsmall += table_rows[i]; }
        else if (table_rows[i] > 0) { ng += (table_rows[i] + OWN_G - 1) / OWN_G; t.lrow_base[i] = t.ltot; t.ltot += table_rows[i]; }
    }
    t.grp_base[ntables] = ng;
    t.nchunks = (int)((rows + OWN_CH - 1) / OWN_CH);
    static const int hot_env = [] { const char* e = getenv("ALM_EMBED_SCATTER_HOT"); return e ? atoi(e) : 1; }();     // 0: the plain owner path (A/B)
    pl.small_fl = (long long)t.nchunks * t.nsmall * D;
    pl.hot = hot_env != 0 && ng > 0 && D > 0 && (D + 1023) / 1024 <= HOT_NY && 2 * rows > HOT_T;
    pl.hot_fl = pl.hot ? (2 * rows / HOT_PT + 2 * HOT_MAX + 1) * (long long)D : 0;
    pl.int_fl = pl.hot ? (long long)IW_CNT + ((t.ltot + 3) & ~3) : 0;
    return 0;
}

extern "C" int alm_embed_scatter_ws_floats(const int* table_rows, int ntables, long long rows, int D) {
    OwnTabs t;
    OwnPlan pl;
    if (rows < 0 || rows >= (1LL << 28) || own_tabs(t, pl, nullptr, table_rows, ntables, rows, D)) return -1;
    const long long fl = pl.small_fl + pl.hot_fl + pl.int_fl;
    return fl > 0x7fffffffLL ? -1 : (int)fl;
}

extern "C" int alm_embed_scatter_owned(float* const* grad_tables, const int* table_rows, int ntables, const int* src_a, const int* src_b, const float* dout,
                                       float alpha, long long rows, int D, float* ws, void* stream) {
    OwnTabs t;
    OwnPlan pl;
    int rc = own_tabs(t, pl, grad_tables, table_rows, ntables, rows, D);
    if (rc) return rc;
    if (D <= 0 || (D & 3) || rows < 0 || rows >= (1LL << 28) || ((uintptr_t)dout & 15) || ((uintptr_t)ws & 15)) return ALM_ERR_BAD_ARG;
    for (int i = 0; i < ntables; ++i)
        if ((uintptr_t)t.p[i] & 15) return ALM_ERR_BAD_ARG;
    const int ngroups = t.grp_base[ntables];
    if (rows == 0) {                                   // nothing maps anywhere: every table's gradient is zero (the scan below assumes at least one code)
        for (int i = 0; i < ntables; ++i)
            if (t.rows[i] > 0) {
                const int e = alm_memset_zero(t.p[i], (long long)t.rows[i] * D * (long long)sizeof(float), stream);
                if (e != 0) return e;
            }
        return 0;
    }
    if ((pl.small_fl > 0 || pl.hot) && !ws) return ALM_ERR_BAD_ARG;
    float* const hp = pl.hot ? ws + pl.small_fl : nullptr;
    int* const iw = pl.hot ? reinterpret_cast<int*>(ws + pl.small_fl + pl.hot_fl) : nullptr;
    // tickets, counts: zero (the kernels do not clean up after themselves: a fresh workspace per call).  A KERNEL, not hipMemsetAsync: captured into a hipGraph
    // (graphed.GraphedTrainStep) the memset node left the workspace uncleared on replay -- the listing then read the previous replay's leftovers and the step
    // faulted (tests/test_gpu_graphed.py found it); a kernel node replays like every other launch of the step
    if (pl.hot)
        hipLaunchKernelGGL(zero_ints_kernel, dim3((unsigned)((pl.int_fl / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<int4*>(iw),
                           (int)(pl.int_fl / 4));
    if ((t.nsmall > 0 || pl.hot) && t.nchunks > 0) {
        const size_t smem = 2 * OWN_CH * sizeof(int) + (size_t)t.nsmall * 256 * sizeof(float4);       // <= 256 B + 32 x 4 KB
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embed_scatter_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(2 * OWN_CH * sizeof(int) + SMALL_ROWS * 256 * sizeof(float4)));
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
        hipLaunchKernelGGL(embed_scatter_small_kernel, dim3((unsigned)t.nchunks, (unsigned)((D + 1023) / 1024)), dim3(256), smem, (hipStream_t)stream, t, src_a,
                           src_b, dout, alpha, rows, D, ws, iw);
    }
    const int nsplit = pl.hot ? HOT_NSPLIT : 0;
    if (ngroups + t.nsmall > 0)
        hipLaunchKernelGGL(embed_scatter_owned_kernel, dim3((unsigned)(nsplit + ngroups + 4 * t.nsmall), (unsigned)((D + 1023) / 1024)), dim3(256), 0,
                           (hipStream_t)stream, t, src_a, src_b, dout, alpha, rows, D, (const float*)ws, ngroups, nsplit, iw, hp);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_gather_rows_bf16(const void* in, long long ld_in, const int* idx, void* out, long long ld_out, long long rows, int D,
                                    void* stream) {
    if ((D & 7) || (ld_in & 7) || (ld_out & 7)) return ALM_ERR_BAD_ARG;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, idx,
                       (bf16_t*)out, ld_out, rows, D);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_gather_split_bf16(const float* in, long long ld_in, long long rows_in, const int* idx, void* hi, void* lo, long long ld_out,
                                     long long rows_out, int D, void* stream) {
    if ((D & 3) || (ld_in & 3) || (ld_out & 3)) return ALM_ERR_BAD_ARG;
    if (rows_out <= 0) return 0;
    hipLaunchKernelGGL(gather_split_kernel, dim3(grid_for(rows_out * (D / 4))), dim3(256), 0, (hipStream_t)stream, in, ld_in, rows_in, idx,
                       (bf16_t*)hi, (bf16_t*)lo, ld_out, rows_out, D);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_scatter_rows_bf16(const void* in, long long ld_in, const int* idx, void* out, long long ld_out, long long rows, int D,
                                     void* stream) {
    if ((D & 7) || (ld_in & 7) || (ld_out & 7)) return ALM_ERR_BAD_ARG;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(rows * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, idx,
                       (bf16_t*)out, ld_out, rows, D);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_cross_entropy_fwd(const float* logits, long long ld, const long long* labels, float* loss_rows, float* lse, long long rows,
                                     int C, int ignore_index, void* stream) {
    if (rows <= 0) return 0;
    const int grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, loss_rows, lse, rows, C, ignore_index);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_cross_entropy_bwd(const float* logits, long long ld, const long long* labels, const float* lse, const float* gscale,
                                     void* dlogits, long long ldd, long long rows, int C, int Cpad, int ignore_index, void* stream) {
    if (rows <= 0) return 0;
    if (Cpad < C || ldd < Cpad) return ALM_ERR_BAD_ARG;
    const int grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, lse, gscale, (bf16_t*)dlogits, ldd, rows,
                       C, Cpad, ignore_index);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_reduce_sum(const float* in, long long n, float* out, float scale, void* stream) {
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, n, out, scale);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_value_residual_mix(const void* v, long long ldv, const void* v0, long long ldv0, void* out, long long ldo, long long rows,
                                      int dim_head, void* stream) {
    if ((dim_head & 7) || (ldv & 7) || (ldv0 & 7) || (ldo & 7)) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(vmix_kernel, dim3(grid_for(rows * (dim_head / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v, ldv,
                       (const bf16_t*)v0, ldv0, (bf16_t*)out, ldo, rows, dim_head);
    ALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int alm_kv_grad_pack(const float* dk, const float* dv, long long ld, int nparts, long long part_stride, float* acc_v0, void* dkv,
                                long long ldo, long long rows, int dim_head, int mode, void* stream) {
    if (mode < 0 || mode > 2 || (mode && !acc_v0) || nparts < 1) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(kv_grad_pack_kernel, dim3(grid_for(rows * dim_head)), dim3(256), 0, (hipStream_t)stream, dk, dv, ld, nparts, part_stride,
                       acc_v0, (bf16_t*)dkv, ldo, rows, dim_head, mode);
    ALM_LAUNCH_CHECK();
    return 0;
}

// keep [B][N] (bytes: torch.bool storage) &= forgetful mask of score [B][N]: see forgetful_mask_kernel.  drop = min(int(N * mask_prob), N - 1).
extern "C" int alm_forgetful_mask(const float* score, long long ld_score, void* keep, long long ld_keep, int B, int N, int drop, void* stream) {
    if (B <= 0 || N <= 0 || drop <= 0) return 0;
    if (drop > N - 1 || N > 256 * FM_MAX_PER_THREAD) return ALM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* kp = (unsigned char*)keep;
    if (N <= 256 * 8) hipLaunchKernelGGL(forgetful_mask_kernel<8>, dim3(B), dim3(256), 0, st, score, ld_score, kp, ld_keep, N, drop);
    else if (N <= 256 * 16) hipLaunchKernelGGL(forgetful_mask_kernel<16>, dim3(B), dim3(256), 0, st, score, ld_score, kp, ld_keep, N, drop);
    else if (N <= 256 * 32) hipLaunchKernelGGL(forgetful_mask_kernel<32>, dim3(B), dim3(256), 0, st, score, ld_score, kp, ld_keep, N, drop);
    else hipLaunchKernelGGL(forgetful_mask_kernel<64>, dim3(B), dim3(256), 0, st, score, ld_score, kp, ld_keep, N, drop);
    ALM_LAUNCH_CHECK();
    return 0;
}

// loss[0] = sum_g w_g * sum_g[0] / max(#(labels_g != ignore), 1), scales[g] = w_g / max(count_g, 1); G <= 4 groups (unused slots: NULL / 0)
extern "C" int alm_loss_combine(const float* s0, const float* s1, const float* s2, const float* s3, const long long* l0, const long long* l1,
                                const long long* l2, const long long* l3, long long n0, long long n1, long long n2, long long n3, float w0, float w1,
                                float w2, float w3, int G, long long ignore_index, float* loss, float* scales, void* stream) {
    if (G < 1 || G > 4 || !loss || !scales) return ALM_ERR_BAD_ARG;
    LossGroups g{{s0, s1, s2, s3}, {l0, l1, l2, l3}, {n0, n1, n2, n3}, {w0, w1, w2, w3}, G, ignore_index};
    for (int k = 0; k < G; ++k)
        if (!g.sum[k] || (g.n[k] > 0 && !g.labels[k]) || g.n[k] < 0) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, loss, scales);
    ALM_LAUNCH_CHECK();
    return 0;
}

// sem [B][ns0] / coarse [B][nc0] int64 ids (row strides ld_*) -> sem_labels [B][ns0 + 1], coarse_labels [B][nc0 + 1] (eos appended), src_a int32 [B][N]
// (embedding source codes, table << 24 | row), keep bytes [B][N] (torch.bool storage; the forgetful mask is ANDed in afterwards), N = ns0 + nc0 + 3.
extern "C" int alm_coarse_prepare(const long long* sem, long long ld_sem, const long long* coarse, long long ld_coarse, int B, int ns0, int nc0,
                                  long long pad_id, long long sem_eos, long long coarse_eos, int Q, int C, long long* sem_labels,
                                  long long* coarse_labels, int* src_a, void* keep, int sem_has_eos, void* stream) {
    if (B <= 0 || ns0 < 0 || nc0 < 0 || Q < 1 || (sem_has_eos && ns0 < 1)) return ALM_ERR_BAD_ARG;
    const long long N = (long long)ns0 + nc0 + 3 - (sem_has_eos ? 1 : 0), W = N > nc0 + 1 ? N : nc0 + 1;
    if ((long long)(Q - 1) * C + C + 1 >= (1 << 24) || B * W >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(coarse_prepare_kernel, dim3(grid_for(B * W)), dim3(256), 0, (hipStream_t)stream, sem, ld_sem, coarse, ld_coarse, B, ns0, nc0, pad_id,
                       sem_eos, coarse_eos, Q, C, sem_labels, coarse_labels, src_a, (unsigned char*)keep, sem_has_eos);
    ALM_LAUNCH_CHECK();
    return 0;
}

/* see semantic_prepare_kernel.  sem int64 [B][n0] (row stride ld_sem) */
extern "C" int alm_semantic_prepare(const long long* sem, long long ld_sem, int B, int n0, long long eos_id, long long num_rows, long long* labels, int* src_a,
                                    int has_eos, void* stream) {
    if (B <= 0 || n0 < 0 || !labels || !src_a || (n0 > 0 && !sem) || (has_eos && n0 < 1)) return ALM_ERR_BAD_ARG;
    if (num_rows >= (1 << 24) || (long long)B * (n0 + 1) >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;      // (the table id lives in bits 24+ of a source code)
    hipLaunchKernelGGL(semantic_prepare_kernel, dim3(grid_for((long long)B * (n0 + 1))), dim3(256), 0, (hipStream_t)stream, sem, ld_sem, B, n0, eos_id, labels, src_a,
                       has_eos);
    ALM_LAUNCH_CHECK();
    return 0;
}

/* see unique_consecutive_kernel.  ids int64 [B][n] (row stride ld), out int64 [B][n + append_eos] (row stride ld_out), lengths int32 [B] */
extern "C" int alm_unique_consecutive_i64(const long long* ids, long long ld, int B, int n, int append_eos, long long eos_id, long long pad, long long* out,
                                          long long ld_out, int* lengths, void* stream) {
    if (B < 0 || n < 0 || !lengths || (n > 0 && !ids) || ld_out < n + (append_eos ? 1 : 0)) return ALM_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (n + (append_eos ? 1 : 0) > 0 && !out) return ALM_ERR_BAD_ARG;
    hipLaunchKernelGGL(unique_consecutive_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, ids, ld, n, append_eos ? 1 : 0, eos_id, pad, out, ld_out, lengths);
    ALM_LAUNCH_CHECK();
    return 0;
}

/* see fine_prepare_kernel.  coarse int64 [B][n], fine int64 [B][>= nf] (the first nf ids of each row are used: the wrapper drops the last fine id). */
extern "C" int alm_fine_prepare(const long long* coarse, long long ld_coarse, const long long* fine, long long ld_fine, int B, int n, int nf, long long pad_id,
                                long long eos_id, int Qc, int Qf, int C, int* src_a, void* keep, void* stream) {
    if (B <= 0 || n < 0 || nf < 0 || Qc < 1 || Qf < 1) return ALM_ERR_BAD_ARG;
    const long long N = (long long)n + nf + 2;
    if ((long long)(Qc > Qf ? Qc : Qf) * C >= (1 << 24) || B * N >= 0x7fffffffLL) return ALM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fine_prepare_kernel, dim3(grid_for(B * N)), dim3(256), 0, (hipStream_t)stream, coarse, ld_coarse, fine, ld_fine, B, n, nf, pad_id, eos_id,
                       Qc, Qf, C, src_a, (unsigned char*)keep);
    ALM_LAUNCH_CHECK();
    return 0;
}
