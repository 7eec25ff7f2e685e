// micro-benchmark (round 5): does the residual-stream LAYOUT limit the hyper-connection kernels' memory pattern?
//   layout 0 = [B][S][N][D] (shipped: the S rows of one token are N*D*2 bytes = 4 MiB apart at the headline shape)
//   layout 1 = [B][N][S][D] (token-major: the S rows of one token are one contiguous 8 KB block)
//   layout 2 = token-major read as ONE linear 8 KB piece per token (16 B per lane, 2 loads per lane): the ceiling of a token-major DMA
//   layout 3 = plain linear copy of the same bytes (grid-stride, 16 B per lane): the chip's copy ceiling
// bf16 streams (2 KB rows, D = 1024), S = 4, one workgroup (256 threads x 8 B) per token and pass, like hc_fwd / hc_bwd; traffic per token shaped like
// hc_fwd (read S rows + 1 row y, write S rows + 2 rows x / xn) or a pure S -> S copy.  OCC = resident workgroups per CU (capped through dynamic LDS),
// PF = software prefetch of the next token's rows (two register sets), as the shipped kernels do at 2-3 workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int S = 4, D = 1024;

__device__ __forceinline__ long long row_off(int layout, int B, int N, int b, int n, int s) {       // in elements (bf16)
    return layout == 0 ? (((long long)b * S + s) * N + n) * D : (((long long)b * N + n) * S + s) * D;
}

template <int LAYOUT, bool PF, bool HCF>
__global__ __launch_bounds__(256) void k(const unsigned short* __restrict__ R, unsigned short* __restrict__ O, const unsigned short* __restrict__ Y,
                                         unsigned short* __restrict__ X, unsigned short* __restrict__ XN, int B, int N) {
    extern __shared__ unsigned char lds[];
    const int tokens = B * N;
    const int t0 = threadIdx.x;
    if (LAYOUT == 3) {
        const long long n16 = (long long)tokens * S * D / 8;
        const uint4* src = reinterpret_cast<const uint4*>(R);
        uint4* dst = reinterpret_cast<uint4*>(O);
        for (long long i = (long long)blockIdx.x * 256 + t0; i < n16; i += (long long)gridDim.x * 256 * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long j = i + (long long)u * gridDim.x * 256; v[u] = j < n16 ? src[j] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long j = i + (long long)u * gridDim.x * 256; if (j < n16) dst[j] = v[u]; }
        }
        return;
    }
    if (LAYOUT == 2) {
        for (int t = blockIdx.x; t < tokens; t += gridDim.x) {
            const uint4* src = reinterpret_cast<const uint4*>(R + (long long)t * S * D);
            uint4* dst = reinterpret_cast<uint4*>(O + (long long)t * S * D);
            const uint4 a = src[t0], b = src[256 + t0];
            uint2 y = make_uint2(0, 0);
            if (HCF) y = *reinterpret_cast<const uint2*>(Y + (long long)t * D + t0 * 4);
            dst[t0] = a; dst[256 + t0] = b;
            if (HCF) {
                *reinterpret_cast<uint2*>(X + (long long)t * D + t0 * 4) = make_uint2(a.x ^ y.x, a.y ^ y.y);
                *reinterpret_cast<uint2*>(XN + (long long)t * D + t0 * 4) = make_uint2(b.x ^ y.x, b.y ^ y.y);
            }
        }
        return;
    }
    const int e0 = t0 * 4;
    auto load = [&](int t, uint2 (&v)[S], uint2& y) {
        const int b = t / N, n = t % N;
#pragma unroll
        for (int s = 0; s < S; ++s) v[s] = *reinterpret_cast<const uint2*>(R + row_off(LAYOUT, B, N, b, n, s) + e0);
        if (HCF) y = *reinterpret_cast<const uint2*>(Y + (long long)t * D + e0);
    };
    auto store = [&](int t, const uint2 (&v)[S], const uint2& y) {
        const int b = t / N, n = t % N;
#pragma unroll
        for (int s = 0; s < S; ++s) *reinterpret_cast<uint2*>(O + row_off(LAYOUT, B, N, b, n, s) + e0) = v[s];
        if (HCF) {
            *reinterpret_cast<uint2*>(X + (long long)t * D + e0) = make_uint2(v[0].x ^ y.x, v[1].y ^ y.y);
            *reinterpret_cast<uint2*>(XN + (long long)t * D + e0) = make_uint2(v[2].x ^ y.x, v[3].y ^ y.y);
        }
    };
    if (!PF) {
        for (int t = blockIdx.x; t < tokens; t += gridDim.x) {
            uint2 v[S], y = make_uint2(0, 0);
            load(t, v, y);
            store(t, v, y);
        }
        return;
    }
    uint2 va[S], vb[S], ya = make_uint2(0, 0), yb = make_uint2(0, 0);
    int t = blockIdx.x;
    if (t < tokens) load(t, va, ya);
    for (; t < tokens; t += 2 * gridDim.x) {
        const int t1 = t + gridDim.x, t2 = t + 2 * gridDim.x;
        if (t1 < tokens) load(t1, vb, yb);
        store(t, va, ya);
        if (t2 < tokens) load(t2, va, ya);
        if (t1 < tokens) store(t1, vb, yb);
    }
}

template <int LAYOUT, bool PF, bool HCF>
void run(const char* name, int occ) {
    const int B = 8, N = 2048;
    const size_t n = (size_t)B * S * N * D, m = (size_t)B * N * D;
    unsigned short *R, *O, *Y, *X, *XN;
    hipMalloc(&R, n * 2); hipMalloc(&O, n * 2); hipMalloc(&Y, m * 2); hipMalloc(&X, m * 2); hipMalloc(&XN, m * 2);
    hipMemset(R, 1, n * 2); hipMemset(Y, 2, m * 2);
    auto fn = k<LAYOUT, PF, HCF>;
    const int lds = 160 * 1024 / occ - 1024 > 65536 ? 160 * 1024 / occ - 1024 : (occ >= 8 ? 0 : 160 * 1024 / occ - 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * occ;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, R, O, Y, X, XN, B, N);
    hipEventRecord(a);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, R, O, Y, X, XN, B, N);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= iters;
    const double bytes = (LAYOUT == 3) ? 2.0 * n * 2 : (HCF ? (2.0 * n + 3.0 * m) * 2 : 2.0 * n * 2);
    printf("%-44s occ %d %s %s: %7.1f us  %5.2f TB/s\n", name, occ, PF ? "prefetch" : "plain   ", HCF ? "hc_fwd-shaped" : "copy S->S     ", ms * 1e3,
           bytes / (ms * 1e-3) / 1e12);
    hipFree(R); hipFree(O); hipFree(Y); hipFree(X); hipFree(XN);
}

int main() {
    if (hipSetDevice(0) != hipSuccess) { printf("no device\n"); return 1; }
    for (int rep = 0; rep < 2; ++rep) {
        printf("---- round %d\n", rep);
        for (int occ : {2, 3, 8}) {
            run<0, true, true>("[B][S][N][D] (shipped)", occ);
            run<1, true, true>("[B][N][S][D] token-major, 8 B lanes", occ);
            run<0, true, false>("[B][S][N][D] (shipped)", occ);
            run<1, true, false>("[B][N][S][D] token-major, 8 B lanes", occ);
        }
        for (int occ : {2, 8}) {
            run<0, false, true>("[B][S][N][D] (shipped)", occ);
            run<1, false, true>("[B][N][S][D] token-major, 8 B lanes", occ);
            run<2, false, true>("token-major, linear 8 KB per token, 16 B lanes", occ);
            run<2, false, false>("token-major, linear 8 KB per token, 16 B lanes", occ);
            run<3, false, false>("plain linear copy", occ);
        }
    }
    return 0;
}
