/* Bench-only GEMM variants (libaudiolm_gemm_lab.so): NOT part of the product ABI (include/audiolm_hip.h). See gemm_lab.hip. */
#ifndef AUDIOLM_GEMM_LAB_H
#define AUDIOLM_GEMM_LAB_H
#ifdef __cplusplus
extern "C" {
#endif
int almlab_gemm_bf16_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb,
                        long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1,
                        long long sC2, float alpha, int out_f32, int accumulate, void* stream);
/* tile: 0 auto, 1 128x128, 2 256x256 lock-step, 3 256x128 3-stage ring, 4 persistent, 6/7 software-pipelined, 8/9 B from registers,
 * 10/12 deep ring with 32-deep K-steps, 11 384x256, 13 staggered wave rows */
int almlab_gemm_bf16_nt_tile(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb,
                             long long ldc, float alpha, int out_f32, int accumulate, int tile, void* stream);
int almlab_debug_splitk(int tile, int slices, int raster);   /* force the split-K tile / slice count / rasterisation (0 = automatic) */
int almlab_debug_stream(int mode);                           /* balanced split: 0 never, 1 by the cost model, 2 whenever applicable */
int almlab_gemm_splitk_slices(int M, int N, int K, int nb);
int almlab_gemm_splitk_ws_floats(int M, int N, int K, int nb);
int almlab_gemm_bf16_nt_splitk(const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                               long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate, void* stream);
int almlab_gemm_bf16_tn_splitk(const void* At, const void* Bt, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                               long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate, void* stream);
#ifdef __cplusplus
}
#endif
#endif
