// micro-benchmark: issue cost (cycles per wave64 instruction) of the instruction kinds the hyper-connection kernels are made of, on one wave per SIMD
// and on two / three -- and whether the f32 4x4x1 MFMA runs BESIDE them.  The hc_bwd probes (profiles/r4l_hc_probe.log) say the kernel's dependency chain
// costs the same at two and at three workgroups per CU: something is throughput-bound per SIMD.  Candidates: v_pk_fma_f32 issuing at half rate (then a packed
// FMA buys nothing over two scalar ones), v_readlane_b32 / DPP, and whether v_mfma_f32_4x4x1_16b_f32 (exact f32, 16 outer products per instruction) can take
// the rank-1 updates of the element loop off the VALU.
//   kinds: 0 v_fma_f32 (independent)   1 v_pk_fma_f32 (independent)   2 v_readlane_b32 + use   3 v_mfma_f32_4x4x1 (4 independent accumulators)
//          4 v_mfma 4x4x1 dependent chain (1 accumulator)   5 interleaved 1 MFMA : 2 v_pk_fma   6 interleaved 1 MFMA : 4 v_fma   7 v_dot2c_f32_bf16   8 v_exp_f32
// usage: valu_mfma_rate            (prints cycles per instruction for 1, 2, 3 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) float f4;
constexpr int N = 256;            // instructions of the kind per loop body (unrolled)
constexpr int REPS = 64;

template <int KIND>
__global__ __launch_bounds__(256) void rate(float* sink, unsigned long long* cyc, float seed) {
    const int lane = threadIdx.x & 63;
    float a[16];
    f2 p[8];
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{seed + i, seed - i};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float x = seed, y = seed * 0.5f;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < N / 16; ++i) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
            } else if constexpr (KIND == 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]), "v"(p[(k + 2) & 7]));
            } else if constexpr (KIND == 2) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float s;
                    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(a[k]), "n"(5));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(k + 8) & 15]) : "s"(s), "v"(y));
                }
            } else if constexpr (KIND == 3) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], a[(k + 1) & 15], acc[k & 3], 0, 0, 0);
            } else if constexpr (KIND == 4) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], a[(k + 1) & 15], acc[0], 0, 0, 0);
            } else if constexpr (KIND == 5) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k % 3 == 0) acc[(k / 3) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], a[(k + 1) & 15], acc[(k / 3) & 3], 0, 0, 0);
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]), "v"(p[(k + 2) & 7]));
                }
            } else if constexpr (KIND == 6) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k % 5 == 0) acc[(k / 5) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], a[(k + 1) & 15], acc[(k / 5) & 3], 0, 0, 0);
                    else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                }
            } else if constexpr (KIND == 7) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, %1" : "=v"(a[k]) : "v"(x));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1234.5f) sink[0] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* sink, unsigned long long* dcyc) {
    printf("%-44s", name);
    for (int wps = 1; wps <= 3; ++wps) {                         // waves per SIMD = workgroups (4 waves) per CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, sink, dcyc, 1.0f);
        hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, sink, dcyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0;
        for (auto v : h) m += (double)v;
        m /= h.size();
        // cycles of SIMD time per instruction = wave cycles / instructions / ... (wps waves share the SIMD: per-wave cycles already include the sharing)
        printf("  %d w/SIMD: %6.2f cyc/instr/wave (%5.2f per SIMD)", wps, m / (REPS * N), m / (REPS * N) / wps);
    }
    printf("\n");
}

int main() {
    float* sink; unsigned long long* dcyc;
    hipMalloc(&sink, 64); hipMalloc(&dcyc, 8 * 4096);
    run<0>("v_fma_f32 (independent)", sink, dcyc);
    run<1>("v_pk_fma_f32 (independent)", sink, dcyc);
    run<2>("v_readlane_b32 + v_fma with the SGPR (pair)", sink, dcyc);
    run<3>("v_mfma_f32_4x4x1f32, 4 accumulators", sink, dcyc);
    run<4>("v_mfma_f32_4x4x1f32, dependent chain", sink, dcyc);
    run<5>("1 MFMA 4x4x1 : 2 v_pk_fma_f32 (per instr)", sink, dcyc);
    run<6>("1 MFMA 4x4x1 : 4 v_fma_f32 (per instr)", sink, dcyc);
    run<7>("v_dot2c_f32_bf16", sink, dcyc);
    run<8>("v_exp_f32", sink, dcyc);
    return 0;
}
