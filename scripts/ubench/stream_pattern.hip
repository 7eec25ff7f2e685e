// micro-benchmark: what HBM bandwidth does the hyper-connection kernels' ACCESS PATTERN reach without any arithmetic?
// layout R[B][S][N][D] fp32 (D = 1024), one workgroup (256 threads x float4) per token: reads the S stream rows of the token (4 KB each,
// N*D*4 bytes apart), writes S rows.  Variants: copy (read S, write S), read-only reduction, two tokens in flight per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int S = 4, D = 1024;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ R, float4* __restrict__ O, float* __restrict__ red, int B, int N) {
    const int tokens = B * N;
    const int d4 = threadIdx.x;
    float acc = 0.f;
    if (MODE == 2) {
        for (int t = blockIdx.x * 2; t < tokens; t += gridDim.x * 2) {
            float4 v[2][S];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = t + u < tokens ? t + u : t;
                const int b = tt / N, n = tt % N;
#pragma unroll
                for (int s = 0; s < S; ++s) v[u][s] = R[(((long long)b * S + s) * N + n) * (D / 4) + d4];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (t + u >= tokens) break;
                const int b = (t + u) / N, n = (t + u) % N;
#pragma unroll
                for (int s = 0; s < S; ++s) O[(((long long)b * S + s) * N + n) * (D / 4) + d4] = v[u][s];
            }
        }
        return;
    }
    for (int t = blockIdx.x; t < tokens; t += gridDim.x) {
        const int b = t / N, n = t % N;
        float4 v[S];
#pragma unroll
        for (int s = 0; s < S; ++s) v[s] = R[(((long long)b * S + s) * N + n) * (D / 4) + d4];
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) O[(((long long)b * S + s) * N + n) * (D / 4) + d4] = v[s];
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) acc += v[s].x + v[s].y + v[s].z + v[s].w;
        }
    }
    if (MODE == 1) red[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, int grid, double bytes) {
    const int B = 8, N = 2048;
    const size_t n = (size_t)B * S * N * D;
    float4 *R, *O; float* red;
    hipMalloc(&R, n * 4); hipMalloc(&O, n * 4); hipMalloc(&red, (size_t)16384 * 256 * 4);
    hipMemset(R, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, R, O, red, B, N);
    hipEventRecord(a);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, R, O, red, B, N);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-34s grid %5d: %7.1f us  %5.2f TB/s\n", name, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    hipFree(R); hipFree(O); hipFree(red);
}

int main() {
    const double sz = 8.0 * S * 2048 * D * 4;
    for (int grid : {512, 768, 1024, 2048, 4096, 16384}) run<0>("copy (read S rows, write S rows)", grid, 2 * sz);
    for (int grid : {768, 2048, 16384}) run<1>("read only", grid, sz);
    for (int grid : {384, 768, 2048, 8192}) run<2>("copy, 2 tokens in flight", grid, 2 * sz);
    return 0;
}
