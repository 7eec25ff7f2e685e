// Launch lists: a recorded sequence of C-ABI launches re-issued by ONE host call (include/audiolm_hip.h, "launch lists").
//
// Why: the transformer stack of one training step is ~190 launches; issued one by one from Python (tensor allocation + argument marshalling + a ctypes
// call each) they cost ~25 us of host time apiece.  The launch SEQUENCE and every scalar argument of a step are a pure function of the model configuration
// and the batch shape; only the buffer addresses move from step to step.  The host side (audiolm-pytorch_amd/launchlist.py) therefore records the sequence
// once per shape while the ordinary Python path runs, with every pointer argument classified as `base + offset` against a small table of bases (the step's
// activation arena, the inputs, every parameter, every packed weight image), and from then on hands that table and the recorded list to alm_list_run,
// which resolves the pointers and calls the SAME entry points in the same order -- the kernels, their launch geometry and their results are bit-identical
// to the Python-issued step (tests/test_gpu_launchlist.py), and unlike a hipGraph replay nothing is baked: streams, events and addresses are live.
//
// Host code only (no kernel in this file); compiled with hipcc like the rest of the library for the runtime headers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <utility>
#include <vector>

#include "../../include/audiolm_hip.h"

namespace {

// one 64-bit slot -> one C argument
template <class T> struct Slot;
template <> struct Slot<int> { static int get(uint64_t v) { return (int)(int64_t)v; } };
template <> struct Slot<long long> { static long long get(uint64_t v) { return (long long)v; } };
template <> struct Slot<unsigned long long> { static unsigned long long get(uint64_t v) { return (unsigned long long)v; } };
template <> struct Slot<float> {
    static float get(uint64_t v) {                   // the fp32 bit pattern rides in the low word
        uint32_t u = (uint32_t)v;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
};
template <class T> struct Slot<T*> { static T* get(uint64_t v) { return (T*)(uintptr_t)v; } };

template <class... A, size_t... I> int invoke(int (*f)(A...), const uint64_t* a, std::index_sequence<I...>) { return f(Slot<A>::get(a[I])...); }
template <class... A> int call_packed(int (*f)(A...), const uint64_t* a) { return invoke(f, a, std::index_sequence_for<A...>{}); }
template <class... A> constexpr int arity(int (*)(A...)) { return (int)sizeof...(A); }

struct Op {
    const char* name;
    int nargs;
    int (*thunk)(const uint64_t*);
};

#define ALM_OP(fn) {#fn, arity(fn), [](const uint64_t* a) -> int { return call_packed(fn, a); }}

// every entry: all pointer arguments are DEVICE pointers (or host arrays the recorder knows how to rebuild: alm_hc_param_grads_batched), the LAST argument is the stream
// zero [p, p + n): 16-byte stores over the aligned middle (grid-stride), single bytes at the two ragged ends
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* __restrict__ p, long long n) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const long long head = (long long)((16 - (a & 15)) & 15) < n ? (long long)((16 - (a & 15)) & 15) : n;     // bytes before the first 16-byte boundary
    const long long nvec = (n - head) / 16;
    uint4* v = reinterpret_cast<uint4*>(p + head);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) v[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0) {
        if ((long long)threadIdx.x < head) p[threadIdx.x] = 0;
        const long long tail0 = head + nvec * 16;
        if (tail0 + threadIdx.x < n && threadIdx.x < 16) p[tail0 + threadIdx.x] = 0;
    }
}

const Op OPS[] = {
    ALM_OP(alm_memset_zero),
    ALM_OP(alm_gemm_bf16_nt),
    ALM_OP(alm_gemm_bf16_nt_ws),
    ALM_OP(alm_gemm_bf16_nt_group2),
    ALM_OP(alm_gemm_bf16_nt_splitk),
    ALM_OP(alm_gemm_bf16_tn_splitk),
    ALM_OP(alm_gemm_bf16_tn_batched),
    ALM_OP(alm_layernorm_fwd),
    ALM_OP(alm_layernorm_bwd),
    ALM_OP(alm_colsum),
    ALM_OP(alm_colsum_partial),
    ALM_OP(alm_geglu_ln_fwd),
    ALM_OP(alm_geglu_ln_bwd),
    ALM_OP(alm_mqa_attn_fwd),
    ALM_OP(alm_mqa_attn_bwd),
    ALM_OP(alm_mqa_attn_bias_fwd),
    ALM_OP(alm_mqa_attn_bias_bwd),
    ALM_OP(alm_attn_bias_grad_reduce),
    ALM_OP(alm_value_residual_mix),
    ALM_OP(alm_kv_grad_pack),
    ALM_OP(alm_hc_fwd),
    ALM_OP(alm_hc_bwd),
    ALM_OP(alm_hc_param_grads),
    ALM_OP(alm_hc_param_grads_batched),
    ALM_OP(alm_streams_expand),
    ALM_OP(alm_streams_reduce),
    ALM_OP(alm_residual_add),
    ALM_OP(alm_f32_to_bf16),
    ALM_OP(alm_add_f32),
};
constexpr int NOPS = (int)(sizeof(OPS) / sizeof(OPS[0]));

}  // namespace

extern "C" {

// A KERNEL, not hipMemsetAsync: captured into a hipGraph, a small memset node does not clear its buffer on the second and later replays on this ROCm (16 B and
// 10 KB: garbage; 1 MB: fine -- scripts/debug/memset_node_probe.py, profiles/r6u_memset_node_probe.log); a kernel node replays like every other launch.
int alm_memset_zero(void* ptr, long long bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && ptr == nullptr)) return ALM_ERR_BAD_ARG;
    if (bytes == 0) return 0;
    const long long chunks = (bytes + 15) / 16 + 1;                       // (+1: an unaligned start spills into one more 16-byte chunk)
    const long long blocks = (chunks + 255) / 256;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(blocks < 65535 * 16 ? blocks : 65535 * 16)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned char*>(ptr), bytes);
    return (int)hipGetLastError();
}

int alm_list_op_id(const char* name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < NOPS; ++i)
        if (strcmp(OPS[i].name, name) == 0) return i;
    return -1;
}

int alm_list_op_nargs(int op) { return (op >= 0 && op < NOPS) ? OPS[op].nargs : -1; }

int alm_list_run(const AlmListEntry* entries, int n, const unsigned long long* slots, const unsigned short* reloc, int nslots, const unsigned long long* bases,
                 int nbases, void* stream, int* failed_at) {
    if (failed_at) *failed_at = -1;
    if (n < 0 || nslots < 0 || nbases < 0 || (n > 0 && entries == nullptr) || (nslots > 0 && (slots == nullptr || reloc == nullptr)) || (nbases > 0 && bases == nullptr))
        return ALM_ERR_BAD_ARG;
    // resolve every slot once: literals, base + offset pointers, the stream; host arrays point INTO the resolved (pointer arrays) or the recorded (int arrays) slots
    static thread_local std::vector<uint64_t> r;
    r.resize((size_t)nslots);
    for (int i = 0; i < nslots; ++i) {
        const unsigned k = reloc[i];
        if (k == ALM_LIST_LITERAL) r[i] = slots[i];
        else if (k == ALM_LIST_STREAM) r[i] = (uint64_t)(uintptr_t)stream;
        else if (k == ALM_LIST_HOST_PTRS || k == ALM_LIST_HOST_INTS) r[i] = 0;       // second pass
        else if ((int)k <= nbases) r[i] = bases[k - 1] + slots[i];
        else return ALM_ERR_BAD_ARG;
    }
    for (int i = 0; i < nslots; ++i) {
        const unsigned k = reloc[i];
        if (k != ALM_LIST_HOST_PTRS && k != ALM_LIST_HOST_INTS) continue;
        const uint64_t at = slots[i];
        if (at >= (uint64_t)nslots) return ALM_ERR_BAD_ARG;
        r[i] = (uint64_t)(uintptr_t)(k == ALM_LIST_HOST_PTRS ? (const void*)&r[at] : (const void*)&slots[at]);
    }
    for (int e = 0; e < n; ++e) {
        const AlmListEntry& en = entries[e];
        if (en.op < 0 || en.op >= NOPS || en.nargs != OPS[en.op].nargs || en.first < 0 || en.first + en.nargs > nslots) {
            if (failed_at) *failed_at = e;
            return ALM_ERR_BAD_ARG;
        }
        const int rc = OPS[en.op].thunk(r.data() + en.first);
        if (rc != 0) {
            if (failed_at) *failed_at = e;
            return rc;
        }
    }
    return 0;
}

}  // extern "C"
