// micro-benchmark: GPU-side cost of back-to-back dependent kernel launches in one stream (queue kept full by the host), plain launches vs
// a captured hipGraph.  Tells how much of a ~400-kernel training step can be launch latency.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void medium(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }

int main() {
    float* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 4000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < N; ++i) {
            if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
            else hipLaunchKernelGGL(medium, dim3(4096), dim3(256), 0, st, d, 1 << 20);        // 8 MB of traffic: ~3 us of work
        }
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s kernels, plain launches : %.2f us per kernel\n", mode ? "medium (1 Mi elements)" : "empty", ms * 1e3 / N);
    }
    // graph of 400 kernels
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 400; ++i) {
            if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
            else hipLaunchKernelGGL(medium, dim3(4096), dim3(256), 0, st, d, 1 << 20);
        }
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s kernels, hipGraph of 400: %.2f us per kernel\n", mode ? "medium (1 Mi elements)" : "empty", ms * 1e3 / 4000);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
