// micro-benchmark: WHERE do the workgroups of an attention-shaped launch run?  The attention kernels pair a heavy and a light causal block on
// every CU by workgroup index (decode_block in csrc/attention.hip assumes "workgroup L -> XCD L % 8, co-resident pairs (j, j + 32) inside the
// XCD").  This launches the same grid shape (512 workgroups x 256 threads, 33 KB of dynamic LDS, >= 200 VGPRs: two workgroups per CU), lets
// every workgroup spin for a time proportional to its causal weight and records (XCC, SE, CU, start, end).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

struct Rec { unsigned hw, xcc, weight, pad; unsigned long long t0, t1; };

__global__ __launch_bounds__(256, 2) void probe(Rec* out, int nblk, int combos, int unit, int pair) {
    extern __shared__ unsigned char smem[];
    const int L = blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    const int cl = j / nblk, idx = j % nblk;
    const bool flip = pair && (cl & 1);
    const int blk = flip ? idx : nblk - 1 - idx;
    const int weight = blk + 1;
    const unsigned long long t0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < weight * unit; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = __builtin_fmaf(a, b, 1e-7f);
    }
    asm volatile("" ::: "v200");                                     // >= 201 VGPRs: at most two workgroups per CU, as for the real kernel
    if (a == 12345.f) smem[threadIdx.x] = 1;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        Rec r;
        r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);           // HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);         // XCC_ID
        r.weight = weight; r.pad = (unsigned)(cl * 8 + xcd);
        r.t0 = t0; r.t1 = t1;
        out[L] = r;
    }
}

int main(int argc, char** argv) {
    const int nblk = 32, combos = 16, G = nblk * combos;
    Rec* d; hipMalloc(&d, G * sizeof(Rec));
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 33280);
    for (int pair = 1; pair >= 0; --pair) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 33280, 0, d, nblk, combos, 200, pair);
        hipDeviceSynchronize();
        std::vector<Rec> h(G);
        hipMemcpy(h.data(), d, G * sizeof(Rec), hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu;                     // (xcc, se, sh, cu) -> workgroups
        unsigned long long tmin = ~0ull, tmax = 0;
        int xcd_guess_ok = 0;
        for (int L = 0; L < G; ++L) {
            const Rec& r = h[L];
            const unsigned key = ((r.xcc & 15) << 16) | (((r.hw >> 13) & 7) << 12) | (((r.hw >> 12) & 1) << 8) | ((r.hw >> 8) & 15);
            cu[key].push_back(L);
            tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1);
            xcd_guess_ok += ((r.xcc & 15) == (unsigned)(L & 7));
        }
        int hist[100] = {0}, nres[8] = {0};
        for (auto& kv : cu) {
            int sum = 0;
            for (int L : kv.second) sum += h[L].weight;
            hist[std::min(sum, 99)]++;
            nres[std::min<size_t>(kv.second.size(), 7)]++;
        }
        printf("pair_on_cu=%d: %zu distinct CUs used; XCD == L %% 8 for %d / %d workgroups; span %.1f us (100 MHz ticks: %llu)\n", pair, cu.size(), xcd_guess_ok, G,
               (tmax - tmin) / 100.0, tmax - tmin);
        printf("  workgroups per CU histogram:");
        for (int i = 0; i < 8; ++i) if (nres[i]) printf("  %d wg: %d CUs", i, nres[i]);
        printf("\n  summed causal weight per CU (ideal: every CU = 33):");
        for (int i = 0; i < 100; ++i) if (hist[i]) printf("  %d:%d", i, hist[i]);
        printf("\n  first CUs:\n");
        int shown = 0;
        for (auto& kv : cu) {
            if (shown++ >= 12) break;
            printf("    xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 12) & 7, (kv.first >> 8) & 1, kv.first & 15);
            for (int L : kv.second) printf("  L=%d (w %u, %.0f..%.0f us)", L, h[L].weight, (h[L].t0 - tmin) / 100.0, (h[L].t1 - tmin) / 100.0);
            printf("\n");
        }
    }
    return 0;
}
