// micro-benchmark: LDS atomic / plain update throughput on gfx950 with the Toeplitz-like addressing of the table-gradient window
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomics scripts/ubench/lds_atomics.hip && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float w[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* win = w + wave * 1024;
    for (int e = lane; e < 1024; e += 64) win[e] = 0.f;
    const int col = lane & 31, lh = lane >> 5;
    float v = 1.0f + lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int slot = 64 + col - row + (it & 7) * 64;
            if (MODE == 0) __hip_atomic_fetch_add(win + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<int*>(win) + slot, (int)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(win) + (slot >> 1), (unsigned long long)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (MODE == 3) win[slot + lh * 512 - (it & 7) * 32] += v;          // plain read-modify-write, halves separated (no same-address pairs)
            if (MODE == 4) __hip_atomic_fetch_add(win + slot + lh * 512 - (it & 7) * 32, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // f32 atomic, no same-address pairs
            if (MODE == 5) __hip_atomic_fetch_max(reinterpret_cast<int*>(win) + slot, (int)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            v += 1e-4f;
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int e = lane; e < 1024; e += 64) s += win[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr = 512.0 * 4 * iters * 16;                     // wave-instructions
    // 256 CUs, 2 workgroups (8 waves) per CU
    printf("%-44s %8.3f ms  %6.1f clk per wave-instruction per CU (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (instr / 256.0));
    hipFree(out);
}

int main() {
    run<0>("ds_add_f32 (2 lanes per address)");
    run<4>("ds_add_f32 (distinct addresses)");
    run<1>("ds_add_u32 (2 lanes per address)");
    run<5>("ds_max_i32 (2 lanes per address)");
    run<2>("ds_add_u64 (4 lanes per address)");
    run<3>("plain ds_read + add + ds_write (distinct)");
    return 0;
}
