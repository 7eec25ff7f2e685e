// micro-benchmark: what does it cost a wave to ISSUE LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction)?
// The forward attention kernel's probe (scripts/attn_probe.py) shows ~1300 cycles per tile step in the segment that issues 4 pieces per wave.
// Launch shape of that kernel: 512 workgroups x 256 threads, 33 KB LDS, two workgroups per CU; every step: issue NP pieces, ~WORK cycles of
// dependent VALU work, s_waitcnt vmcnt(0), barrier.  Variants:
//   0: one M0 value per piece (what dma_tile compiles to)        1: one M0 for all pieces, destination through the instruction's immediate offset
//   2: plain global_load_dwordx4 into registers (no LDS)          3: variant 0 with the pieces spread over the step (one every WORK / NP cycles)
//   4 / 5 / 6: variant 0 with the step's work = (ds_read_b128 + MFMA) / MFMA only / ds_read_b128 only instead of dependent FMAs -- what the OTHER
//   workgroup of the CU is doing while this one issues its pieces
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// the attention kernels' tile DMA, verbatim (csrc/attention.hip): [64 rows][64 dims] bf16 tile -> swizzled LDS image, 8 pieces of 8 rows
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ int fsw(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }
template <int VAR = 0>                               // VAR bit 0: no chunk swizzle   bit 1: no bounds predicate   bit 2: fully unrolled (first, step known)
__device__ __forceinline__ void dma_tile(const __amdgpu_buffer_rsrc_t& rs, unsigned char* img, int first, int step, int lane, int row0, int nrows,
                                         unsigned ld_bytes, unsigned col_bytes) {
#pragma unroll
    for (int i = 0; i < ((VAR & 4) ? 2 : 0); ++i) {
        const int piece = first + 4 * i;
        const int row = piece * 8 + (lane >> 3);
        const int c = (VAR & 1) ? (lane & 7) : ((lane & 7) ^ fsw(row));
        const unsigned vo = ((VAR & 2) || row0 + row < nrows) ? (unsigned)(row0 + row) * ld_bytes + col_bytes + (unsigned)c * 16u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + piece * 1024), 16, vo, 0, 0, 0);
    }
    if (VAR & 4) return;
    for (int piece = first; piece < 8; piece += step) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (VAR & 1) ? (lane & 7) : ((lane & 7) ^ fsw(row));
        const unsigned vo = ((VAR & 2) || row0 + row < nrows) ? (unsigned)(row0 + row) * ld_bytes + col_bytes + (unsigned)c * 16u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + piece * 1024), 16, vo, 0, 0, 0);
    }
}

struct Out { unsigned long long issue, work, wait, bar; };

template <int MODE, int NP>
__global__ __launch_bounds__(256, 2) void probe(const unsigned char* src, Out* out, int steps, int work, float* sink, int nrows) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 1 << 21, 0x00020000);
    unsigned long long a_issue = 0, a_work = 0, a_wait = 0, a_bar = 0;
    float x = lane * 1e-3f;
    u32x4 keep = {0, 0, 0, 0};
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int s = 0; s < steps; ++s) {
        unsigned char* img = smem + (((s & 1) * 16384 + wave * NP * 1024) & 0x7fff) / (NP * 1024) * (NP * 1024) % 32768;
        const unsigned base = (unsigned)((((blockIdx.x & 7) * 37 + s) & 127) * 16384 + wave * NP * 1024 + lane * 16);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 7) {
            unsigned char* timg = smem + (s & 1) * 16384;
            const int row0 = (((blockIdx.x & 7) * 37 + s) & 63) * 64;
            dma_tile<MODE - 7>(rs, timg, wave, 4, lane, row0, nrows, 256, 0);
            dma_tile<MODE - 7>(rs, timg + 8192, wave, 4, lane, row0, nrows, 256, 128);
        } else if (MODE == 0 || MODE >= 4) {
#pragma unroll
            for (int i = 0; i < NP; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + i * 1024), 16, base + i * 1024, 0, 0, 0);
        } else if (MODE == 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)img, 16, base, 0, 0, 0);
            if (NP > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)img, 16, base, 0, 1024, 0);
            if (NP > 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)img, 16, base, 0, 2048, 0);
            if (NP > 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)img, 16, base, 0, 3072, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, base + i * 1024, 0, 0);
                keep ^= v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                for (int k = 0; k < work / NP; ++k) x = __builtin_fmaf(x, 1.0001f, 1e-7f);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + i * 1024), 16, base + i * 1024, 0, 0, 0);
            }
        } else if (MODE >= 4 && MODE < 7) {
            const unsigned char* cur = smem + ((s & 1) ^ 1) * 16384;
            for (int k = 0; k < work / 16; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b2 = {1, 1, 1, 1, 1, 1, 1, 1};
                    if (MODE != 5) {
                        a = *reinterpret_cast<const bf16x8*>(cur + ((lane * 16 + j * 1024 + k * 4096) & 16383));
                        b2 = *reinterpret_cast<const bf16x8*>(cur + ((lane * 16 + j * 1024 + 512 + k * 4096) & 16383));
                    }
                    if (MODE != 6) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc[j], 0, 0, 0);
                    else x += (float)a[0] + (float)b2[1];
                }
            }
        } else {
            for (int k = 0; k < work; ++k) x = __builtin_fmaf(x, 1.0001f, 1e-7f);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        const unsigned long long t4 = __builtin_readcyclecounter();
        a_issue += t1 - t0; a_work += t2 - t1; a_wait += t3 - t2; a_bar += t4 - t3;
    }
    if (x == 12345.f || keep[0] == 0x12345678u || acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) sink[threadIdx.x] = x + smem[lane];
    if (lane == 0) out[blockIdx.x * 4 + wave] = Out{a_issue, a_work, a_wait, a_bar};
}

template <int MODE, int NP>
void run(const char* name, const unsigned char* src, Out* d, float* sink, int work) {
    const int G = 512, steps = 32;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, 33280);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<MODE, NP>), dim3(G), dim3(256), 33280, 0, src, d, steps, work, sink, 8192);
    hipDeviceSynchronize();
    std::vector<Out> h(G * 4);
    hipMemcpy(h.data(), d, sizeof(Out) * G * 4, hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0, e = 0;
    for (auto& o : h) { a += o.issue; b += o.work; c += o.wait; e += o.bar; }
    const double n = (double)h.size() * steps;
    printf("%-58s NP=%d work=%4d: issue %7.0f (%.0f / piece)  work %7.0f  vmcnt(0) %6.0f  barrier %6.0f   cycles per step\n", name, NP, work, a / n, a / n / NP, b / n, c / n,
           e / n);
}

int main() {
    unsigned char* src; hipMalloc(&src, 1 << 21); hipMemset(src, 1, 1 << 21);
    Out* d; hipMalloc(&d, sizeof(Out) * 4096);
    float* sink; hipMalloc(&sink, 4096);
    for (int work : {100}) {
        run<0, 4>("0: M0 per piece", src, d, sink, work);
        run<1, 4>("1: one M0, immediate offsets", src, d, sink, work);
        run<2, 4>("2: global_load_dwordx4 -> registers", src, d, sink, work);
        run<3, 4>("3: M0 per piece, pieces spread over the work", src, d, sink, work);
        run<0, 2>("0: M0 per piece", src, d, sink, work);
        run<0, 8>("0: M0 per piece", src, d, sink, work);
        run<1, 2>("1: one M0, immediate offsets", src, d, sink, work);
    }
    run<7, 4>("7: attention's dma_tile (K + V pieces, swizzled rows)", src, d, sink, 100);
    run<8, 4>("8: dma_tile, no chunk swizzle", src, d, sink, 100);
    run<9, 4>("9: dma_tile, no bounds predicate", src, d, sink, 100);
    run<11, 4>("11: dma_tile, unrolled pieces", src, d, sink, 100);
    run<13, 4>("13: dma_tile, unrolled, no predicate", src, d, sink, 100);
    run<14, 4>("14: dma_tile, unrolled, no swizzle, no predicate", src, d, sink, 100);
    for (int work : {256}) {
        run<4, 4>("4: M0 per piece; work = ds_read_b128 + MFMA", src, d, sink, work);
        run<5, 4>("5: M0 per piece; work = MFMA", src, d, sink, work);
        run<6, 4>("6: M0 per piece; work = ds_read_b128", src, d, sink, work);
    }
    return 0;
}
