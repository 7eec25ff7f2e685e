/* libaudiolm_hip.so -- C ABI of the MI355X (gfx950) kernels behind the audiolm token-transformer training hot path.
 *
 * The reference (lucidrains/audiolm-pytorch) has no FFI of its own: its hot path is stock PyTorch ops.  This header is the
 * drop-in boundary a maintainer binds instead of those ops (ctypes stub: INTEGRATION.md; the in-tree binding is
 * audiolm-pytorch_amd/_lib.py).  Each entry cites the reference code it replaces (paths relative to
 * /root/reference/audiolm_pytorch/).
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller (PyTorch allocator), incl. workspaces
 *   - bf16 tensors are raw uint16 bit patterns (void*); fp32 statistics / master weights / gradients are float*
 *   - `ld*` = leading dimension (row stride) in ELEMENTS; row-major everywhere
 *   - kernels are stateless, re-entrant, asynchronous on `stream` (a hipStream_t passed as void*); callable from any thread
 *   - return 0 on success, a hipError_t value (> 0) or ALM_ERR_* (>= 10001) otherwise; nothing is printed
 */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

#define ALM_ERR_BAD_ARG 10001
#define ALM_ERR_UNSUPPORTED 10002

/* ---- dense contractions (MFMA) -------------------------------------------------------------------------------------------
 * C[z1][z2][M,N] (+)= alpha * A[z1][z2][M,K] . B[z1][z2][N,K]^T (+ bias[N]);  bf16 operands, fp32 accumulate, C bf16 or fp32.
 * Replaces aten::mm / addmm / bmm behind nn.Linear and einsum at audiolm_pytorch.py:255-259 (FFN), :351 (to_q, to_kv),
 * :395 (to_out), :719 / :961 (semantic logits, bias), :972 / :1335 / :1350 (per-quantizer logit heads, batched over z2 = q),
 * and all their dgrad / wgrad forms (operands transposed with alm_transpose_bf16 / alm_pack_weight).
 * K % 8 == 0, lda/ldb % 8 == 0, batch strides % 8 == 0. */
int alm_gemm_bf16_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb,
                     long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1,
                     long long sC2, float alpha, int out_f32, int accumulate, void* stream);
/* alm_gemm_bf16_nt with a caller-owned workspace (round 6): NT launches that leave most of the chip idle on 256 x 256 tiles (every D- / 512-wide projection
 * of audiolm_pytorch.py:255-259, :351, :395 at M = 8192 rows; to_q / dAO at M = 16384) run on the staggered 256 x 256 tile with IN-LAUNCH split-K when the
 * measured cost model picks it (alm_gemm_nt_plan): each K slice publishes its fp32 accumulators to `ws`, the tile's LAST ARRIVER (per-tile ticket counter,
 * agent scope) sums them in fixed slice order and runs the usual epilogue (alpha, bias, accumulate, bf16 / fp32) -- bitwise deterministic, one launch.
 * ws: alm_gemm_nt_ws_bytes() bytes, 16-byte aligned, used by one stream at a time; its first 4096 bytes (the counters) must be ZERO when first handed over,
 * every launch leaves them zero.  ws == NULL or too small: exactly alm_gemm_bf16_nt. */
int alm_gemm_bf16_nt_ws(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb, long long ldc,
                        int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2, float alpha,
                        int out_f32, int accumulate, void* ws, long long ws_bytes, void* stream);
/* un-batched, with a GIVEN number of K slices on the staggered 256 x 256 tile (tests / scripts/ab_nt_inl.py); slices 1 = the plain staggered tile; M, N >= 256 */
int alm_gemm_bf16_nt_inl(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb, long long ldc,
                         float alpha, int out_f32, int accumulate, int slices, void* ws, long long ws_bytes, void* stream);
/* plan of alm_gemm_bf16_nt_ws (with_ws = 1) / alm_gemm_bf16_nt (0) for nb problems, no launch: plan[0] = block tile (1 = 128 x 128, 16 = its 4-stage DMA-ring
 * form, 13 = 256 x 256 staggered, 11 = 384 x 256), plan[1] = K slices (1: no split), plan[2] = workspace bytes that plan uses, plan[3] = workgroups */
int alm_gemm_nt_plan(int M, int N, int K, int nb, int with_ws, int* plan);
int alm_gemm_nt_ws_bytes(void);
/* two independent un-batched NT problems (bf16 or fp32 outputs, no bias, alpha 1) in ONE launch: pairs of projections that sit next to each other in
 * the step and each leave most of the chip idle on their own -- to_q || to_kv (audiolm_pytorch.py:351 / :347: x_norm . Wq^T beside x . Wkv^T) and their two
 * dgrads.  Falls back to two launches when the problems do not pick the same tile; results are identical either way. */
int alm_gemm_bf16_nt_group2(const void* A0, const void* B0, void* C0, int M0, int N0, int K0, long long lda0, long long ldb0, long long ldc0,
                            const void* A1, const void* B1, void* C1, int M1, int N1, int K1, long long lda1, long long ldb1, long long ldc1,
                            int out_f32, void* stream);
/* tile-selectable, un-batched form of alm_gemm_bf16_nt (tests / benchmarks): tile 0 = auto, 1 = 128x128x64 (4 waves), 2 = 256x256x64
 * (8 waves, lock-step), 13 = 256x256x64 with staggered wave rows (the production big tile), 11 = 384x256x64 (8 waves); other ids ->
 * ALM_ERR_UNSUPPORTED (the variants that were measured and not adopted live in the bench-only csrc/lab/gemm_lab.hip). */
int alm_gemm_bf16_nt_tile(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb,
                          long long ldc, float alpha, int out_f32, int accumulate, int tile, void* stream);
/* split-K forms for long-K / few-tile contractions (weight gradients: K = B*N tokens): fp32 C (+)= alpha * op(A) . op(B),
 * deterministic two-stage reduction through `ws` (alm_gemm_splitk_ws_floats(M,N,K,nb) floats; may be NULL when that
 * query returns 0).  `nb` same-shape problems per launch with element strides sA / sB / sC between them (sA, sB % 8 == 0).
 *   _nt_: A[M][K], B[N][K] (K-contiguous operands)
 *   _tn_: At[K][M], Bt[K][N] (contraction-major operands = the row-major activations themselves):
 *         dW[out][in] = sum_tokens dY[token][out] * X[token][in], the wgrad of every nn.Linear on the path
 *         (autograd of audiolm_pytorch.py:255-259, :351, :395, :961, :972) with no transposed copies; lda/ldb % 8 == 0. */
int alm_gemm_splitk_slices(int M, int N, int K, int nb);
int alm_gemm_splitk_ws_floats(int M, int N, int K, int nb);
/* block tile of the split-K plan for this problem: 1 = 128 x 128 (4 waves), otherwise a 256 x 256 tile */
int alm_gemm_splitk_tile(int M, int N, int K, int nb);
/* host-side plan queries (no launch): the kernel a launch WILL take.
 *   alm_gemm_nt_tile_choice: block tile of alm_gemm_bf16_nt(M, N) with nb = nb1 * nb2 problems: 1 = 128 x 128 (4 waves), 13 = 256 x 256 with staggered
 *     wave rows, 11 = 384 x 256 (the shipped choice; A/B environment hooks not applied)
 *   alm_gemm_tn_batched_plan: alm_gemm_bf16_tn_batched(M, N, K) over nb problems -> 0 = one full-K launch, 1 = uniform split-K (plan = {tile, slices, 0, 0}),
 *     2 = hybrid: plan = {panels at full K, slices of the tail, first row / column of the tail in the last problem, tail cut along M}; plan: int[4] | NULL */
int alm_gemm_nt_tile_choice(int M, int N, int nb);
int alm_gemm_tn_batched_plan(int M, int N, int K, int nb, int* plan);
int alm_gemm_bf16_nt_splitk(const void* A, const void* B, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                            long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate, void* stream);
int alm_gemm_bf16_tn_splitk(const void* At, const void* Bt, float* C, float* ws, int M, int N, int K, long long lda, long long ldb,
                            long long ldc, int nb, long long sA, long long sB, long long sC, float alpha, int accumulate, void* stream);
/* two-level batched _tn_ form: C[z1][z2] (+)= alpha * At[z1][z2]^T . Bt[z1][z2] for nb1 x nb2 same-shape problems (element strides per level): every
 * layer's gradient of one weight kind in ONE launch from stacked activation buffers (the deferred weight-gradient mode of the fused backward).
 * ws: alm_gemm_splitk_ws_floats(M, N, K, nb1 * nb2) floats (NULL when that is 0). */
int alm_gemm_bf16_tn_batched(const void* At, const void* Bt, float* C, float* ws, int M, int N, int K, long long lda, long long ldb, long long ldc,
                             int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2, float alpha,
                             int accumulate, void* stream);
/* dst[c][r] = src[r][c]; columns [rows, rows_pad) of dst are zero-filled (K-padding of a transposed GEMM operand). */
int alm_transpose_bf16(const void* src, void* dst, int rows, int cols, long long ld_src, long long ld_dst, int rows_pad, void* stream);
/* nb matrices (element strides bs_src / bs_dst) in ONE launch: the per-sequence key / value sets of a conditioning context (xattn.py). */
int alm_transpose_bf16_batched(const void* src, void* dst, int rows, int cols, long long ld_src, long long ld_dst, int rows_pad, int nb,
                               long long bs_src, long long bs_dst, void* stream);
/* fp32 master weight -> zero-padded bf16 copy (dst, may be NULL) and zero-padded bf16 transpose (dstT, may be NULL).
 * This is the autocast weight cast of trainer.py:1241 (accelerator.autocast), done once per optimiser step. */
int alm_pack_weight(const float* src, int rows, int cols, long long ld_src, void* dst, long long ld_dst, int rows_pad, int cols_pad,
                    void* dstT, long long ld_dstT, void* stream);

/* the same for up to 8 weights in ONE launch (a transformer layer's q / kv / out / W1 x-half / W1 gate-half / W2) */
typedef struct {
    const float* src; int rows, cols; long long ld_src;
    void* dst; long long ld_dst; int rows_pad, cols_pad;
    void* dstT; long long ld_dstT;
} AlmPackJob;
int alm_pack_weights_multi(const AlmPackJob* jobs, int njobs, void* stream);

/* ---- LayerNorm (gamma only, eps 1e-5): audiolm_pytorch.py:191-198 ---------------------------------------------------------- */
int alm_ln_partial_blocks(int rows);
int alm_layernorm_fwd(const void* x, int x_is_bf16, long long ldx, const float* gamma, void* y, int y_is_f32, long long ldy, void* xcopy_bf16,
                      long long ldc, float* mean, float* rstd, int rows, int D, void* stream);   /* y bf16, or fp32 (final LayerNorm feeding the logit heads) */
int alm_layernorm_bwd(const void* dy, int dy_is_f32, long long lddy, const void* x, int x_is_bf16, long long ldx, const float* mean,
                      const float* rstd, const float* gamma, const void* extra_bf16, long long lde, void* dx, int dx_is_bf16,
                      long long lddx, float* dgamma_part, int rows, int D, void* stream);
/* out[c] (+)= scale * sum_r in[r][c]  (second stage of every parameter-gradient reduction; bias gradients) */
int alm_colsum(const void* in, int in_is_bf16, long long ld, int rows, int cols, float* out, float scale, int accumulate, float* ws,
               void* stream);          /* ws: alm_colsum_chunks(rows) * cols floats (two-stage reduction of tall inputs) or NULL */
int alm_colsum_chunks(int rows);
/* stage 1 alone: ws[alm_colsum_chunks(rows)][cols] partial column sums of an fp32 matrix (the consumer sums the chunk rows: alm_hc_param_grads) */
int alm_colsum_partial(const float* in, long long ld, int rows, int cols, float* ws, void* stream);

/* ---- GEGLU + inner LayerNorm: audiolm_pytorch.py:246-260 (gate = second half, exact-erf GELU, LN over int(dim*8/3)) -------- */
/* standalone GEGLU module (audiolm_pytorch.py:246-249) on fp32 rows [rows][2 inner] -> [rows][inner]: y = x * gelu(gate), gate = the second half */
int alm_geglu_fwd(const float* x, float* y, long long rows, int inner, void* stream);
int alm_geglu_bwd(const float* dy, const float* x, float* dx, long long rows, int inner, void* stream);
int alm_geglu_partial_blocks(int rows);
int alm_geglu_ln_fwd(const void* u_bf16, long long ldu, int gate_offset, const float* gamma, void* out_bf16, long long ldo, float* mean,
                     float* rstd, int rows, int inner, int inner_pad, void* stream);
int alm_geglu_ln_bwd(const void* dhn_bf16, long long lddh, const void* u_bf16, long long ldu, int gate_offset, const float* gamma,
                     const float* mean, const float* rstd, void* du_bf16, float* dgamma_part, int rows, int inner, int inner_pad,
                     void* stream);

/* ---- causal multi-query flash attention: attend.py:69-146 as called from audiolm_pytorch.py:381-394 ------------------------
 * q (B,N,H*64) bf16, k/v (B,N,64) bf16 single shared head, mask (B,N) uint8 (1 = attend) or NULL, scale = 64^-0.5.
 * lse fp32 [B][H][N].  Backward: dq bf16; dk/dv fp32 partials [alm_mqa_bwd_parts(B,N,H)][B*N][lddk] (64 columns each, `part_stride`
 * floats between partials; summed by alm_kv_grad_pack: deterministic, no atomics); delta = fp32 workspace [2][B][H][N].
 * dropout_p in [0, 1) / seed: training-mode attention dropout (attend.py:92 `dropout_p`, :140 `attn_dropout(attn)`): the softmax OUTPUT of pair
 * (b, h, i, j) is kept iff hash(seed, b, h, i * N + j) >= dropout_p * 2^32 (a stateless 32-bit finaliser, the same in forward, dQ and dK/dV) and
 * scaled by 1 / (1 - dropout_p); 0 = off (the default path, no cost).  Pass the forward's (dropout_p, seed, seed_dev) to the backward.
 * seed_dev (device uint64 | NULL): when given, the stream is seed + *seed_dev with the counter read when the kernel RUNS -- a hipGraph replay then
 * draws a new mask each time (a by-value seed is baked into the captured launch); the caller advances the counter on the device between steps. */
int alm_mqa_attn_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv, const unsigned char* mask,
                     void* o, long long ldo, float* lse, int B, int N, int H, int dim_head, float scale, float dropout_p,
                     unsigned long long seed, const void* seed_dev, void* stream);
int alm_mqa_attn_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv, const unsigned char* mask,
                     const void* o, long long ldo, const float* lse, const void* dout, long long lddo, void* dq, long long lddq, float* dk,
                     float* dv, long long lddk, long long part_stride, float* delta, int B, int N, int H, int dim_head, float scale,
                     float dropout_p, unsigned long long seed, const void* seed_dev, void* stream);
/* number of head groups (4 heads each) of the forward / dQ kernels (the table-gradient partial rows of the bias variants count in these) */
int alm_mqa_head_groups(int H);
/* number of dk / dv partial sets alm_mqa_attn_bwd / alm_mqa_attn_bias_bwd WRITE for this shape (the dK/dV kernel runs 4 or 2 heads per workgroup: H / 4 or
 * H / 2 partials, part_stride floats apart): the caller allocates this many and hands them all to alm_kv_grad_pack(nparts) */
int alm_mqa_bwd_parts(int B, int N, int H);
/* The same attention with the STRUCTURED SCORE BIAS of the `flash_attn=False` models -- Attend.forward's `sim + attn_bias` (attend.py:118-
 * 121) with the bias tensors of RelativePositionBias (audiolm_pytorch.py:202-242), the Coarse cross-attention override (:924-936) and the
 * Fine (frame, quantizer) table (:1227-1298) -- without ever materialising the (h, n, n) tensor:
 *     bias(h, i, j) = (qattr[i] & kattr[j]) ? tbl[h][0] : tbl[h][(qkey4[i] - kkey4[j]) / 4]
 * tbl fp32 [H][LT] in raw-score units (bias / scale; slot 0 = cross_attn_bias / null_pos_bias); qkey4 / kkey4 / qattr / kattr int32 [N]
 * (16-byte aligned, shared by the batch; qkey4 - kkey4 is a BYTE offset into a table row, offsets outside [0, 4 LT) read 0).
 * Backward additionally accumulates d(loss)/d(tbl) into dtbl_part [alm_attn_bias_part_rows(B, N, H)][LT] (per-workgroup partial tables:
 * zero before the first layer, alm_attn_bias_grad_reduce after the last -- all layers of a stack share one table). */
int alm_mqa_attn_bias_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv, const unsigned char* mask,
                          void* o, long long ldo, float* lse, int B, int N, int H, int dim_head, float scale, const float* tbl, int LT,
                          const int* qkey4, const int* kkey4, const int* qattr, const int* kattr, float dropout_p, unsigned long long seed,
                          const void* seed_dev, void* stream);
int alm_mqa_attn_bias_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv, const unsigned char* mask,
                          const void* o, long long ldo, const float* lse, const void* dout, long long lddo, void* dq, long long lddq,
                          float* dk, float* dv, long long lddk, long long part_stride, float* delta, int B, int N, int H, int dim_head,
                          float scale, const float* tbl, int LT, const int* qkey4, const int* kkey4, const int* qattr, const int* kattr,
                          float* dtbl_part, float dropout_p, unsigned long long seed, const void* seed_dev, void* stream);
int alm_attn_bias_part_rows(int B, int N, int H);
int alm_attn_bias_grad_reduce(const float* dtbl_part, float* dtbl, int B, int N, int H, int LT, float scale, void* stream);
/* The small MLPs that produce `tbl` (RelativePositionBias.net audiolm_pytorch.py:214-221, FineTransformer.pos_bias_mlp :1065-1071): first
 * layer (in_dim 1 or 2 -> C) + SiLU, SiLU forward / backward for the C x C layers (which run on alm_gemm_*), last layer (C -> H) written
 * straight into the table layout [H][L + 1] (x inv_scale, slot 0 = special * inv_scale), and their backward passes. */
int alm_posmlp_in_fwd(const float* x, const float* W, const float* b, float* pre, void* act_bf16, int L, int in_dim, int C, void* stream);
int alm_posmlp_in_bwd_chunks(int L);
int alm_posmlp_in_bwd(const void* dpre_bf16, const float* x, float* partial, int L, int in_dim, int C, void* stream);
int alm_silu_fwd(const float* pre, void* act_bf16, long long n, void* stream);
int alm_silu_bwd(const float* dact, const float* pre, void* dpre_bf16, long long n, void* stream);
int alm_posmlp_out_fwd(const void* act_bf16, const float* W, const float* b, const float* special, float* tbl, int L, int C, int H,
                       float inv_scale, void* stream);
int alm_posmlp_out_bwd(const float* dtbl, const float* W, const float* pre, void* g_bf16, void* dpre_bf16, float* dspecial, int L, int C, int H,
                       int Hp, float inv_scale, void* stream);
/* ---- attention over a short, always-visible key set: the conditioning paths of Attention.forward (audiolm_pytorch.py:307-406) ----------------
 * cross-attention layers (Transformer(cross_attend=True), :450: keys = [null_kv | to_kv(context_norm(text embeds))], non-causal, :372-388) and
 * `cond_as_self_attn_prefix` (:330-345: the text embeds prepended to the causal self-attention keys).  The three small contractions
 * S_e = Q K_e^T [B][N*H][Me], O_e = P_e V_e, and their gradients run on alm_gemm_bf16_*; these entries are the row-wise softmax pieces.
 * Row r = (b*N + n)*H + h; statistics (lse, ndelta) use the flash kernels' [B][H][N] layout.
 *   softmax_fwd: P = exp(S*scale - lse_tot) on unmasked keys (emask uint8 [B][Me], NULL = all), lse_tot = logaddexp(lse_self, lse_e)
 *                (lse_self NULL: no causal self part), fself[r] = exp(lse_self - lse_tot): the weight of the self part's output.
 *   combine:     O[(b n)][h*dh + d] = fself[r] * O_self + O_e[r][d]        (O_self / fself NULL: cross-attention)
 *   softmax_bwd: dS = P o (dP + ndelta) * scale, ndelta[b][h][n] = -sum_d dO*O of the JOINT output (alm_xattn_delta, or the workspace
 *                alm_mqa_attn_bwd filled when a self part exists).
 * The same pieces serve the reference's MATH path with an arbitrary dense attn_bias (attend.py:98-146: sim = q k^T * scale + attn_bias, key mask,
 * causal triu(j - i + 1)): `bias` fp32 [H][N][ldbias] (NULL: none) is added to the scaled scores, `causal_off` makes key e visible to query n iff
 * e <= n + causal_off (Me - N for the reference's rule; 0x7fffffff: not causal); alm_xattn_dbias: dbias[h][n][e] = sum_b P o (dP + ndelta). */
int alm_xattn_softmax_fwd(const float* S, long long ldS, const unsigned char* emask, const float* lse_self, float scale, void* P_bf16, long long ldP,
                          float* lse_tot, float* fself, const float* bias, long long ldbias, int causal_off, int B, int N, int H, int Me, void* stream);
int alm_xattn_dbias(const void* P_bf16, long long ldP, const float* dP, long long lddP, const float* ndelta, float* dbias, long long lddb, int Me, int B,
                    int N, int H, void* stream);
int alm_xattn_combine(const void* o_self_bf16, long long ldos, const float* fself, const float* o_e, void* out_bf16, long long ldo, long long tokens,
                      int H, int dim_head, void* stream);
int alm_xattn_softmax_bwd(const void* P_bf16, long long ldP, const float* dP, long long lddP, const float* ndelta, float scale, void* dS_bf16,
                          long long lddS, int Me, int B, int N, int H, void* stream);
int alm_xattn_delta(const void* o_bf16, long long ldo, const void* dout_bf16, long long lddo, float* ndelta, int B, int N, int H, int dim_head,
                    void* stream);
/* value residual, audiolm_pytorch.py:353-358 / :534-535 */
int alm_value_residual_mix(const void* v, long long ldv, const void* v0, long long ldv0, void* out, long long ldo, long long rows,
                           int dim_head, void* stream);
int alm_kv_grad_pack(const float* dk, const float* dv, long long ld, int nparts, long long part_stride, float* acc_v0, void* dkv_bf16,
                     long long ldo, long long rows, int dim_head, int mode, void* stream);

/* cross-entropy means + the wrappers' weighted combination (audiolm_pytorch.py:1561-1565, :1826-1854, :2112-2137: F.cross_entropy(..., ignore_index) per
 * head, then (loss_a * n_a * w + loss_b * n_b) / (n_a + n_b)) from the per-group loss SUMS: loss[0] = sum_g w_g * sum_g[0] / max(#(labels_g != ignore), 1);
 * scales[g] = w_g / max(count_g, 1) (the backward's factor).  G <= 4 groups; unused slots NULL / 0. */
int alm_loss_combine(const float* s0, const float* s1, const float* s2, const float* s3, const long long* l0, const long long* l1, const long long* l2,
                     const long long* l3, long long n0, long long n1, long long n2, long long n3, float w0, float w1, float w2, float w3, int G,
                     long long ignore_index, float* loss, float* scales, void* stream);

/* CoarseTransformerWrapper.forward's id bookkeeping of a training step, audiolm_pytorch.py:1785-1810 (append eos, key mask of pad / eos semantic keys,
 * their ids zeroed, mask padded over [start | semantic | coarse start | coarse]) + the embedding source codes of :894-918 (start tokens, semantic ids,
 * coarse rows id + (i mod Q) * codebook_size) + the label tensors (ids with the eos appended), in one launch.  N = ns0 + nc0 + 3.
 * sem_has_eos != 0: the semantic rows already are [ids | eos | pad ...] (alm_unique_consecutive_i64, unique_consecutive = True, :1794-1795): nothing is
 * appended to them, sem_labels is [B][ns0] and N = ns0 + nc0 + 2. */
int alm_coarse_prepare(const long long* sem, long long ld_sem, const long long* coarse, long long ld_coarse, int B, int ns0, int nc0, long long pad_id,
                       long long sem_eos, long long coarse_eos, int Q, int C, long long* sem_labels, long long* coarse_labels, int* src_a, void* keep,
                       int sem_has_eos, void* stream);
/* SemanticTransformerWrapper.forward's id bookkeeping of a training step, audiolm_pytorch.py:1536-1548 (`append_eos_id`, input ids = ids[:, :-1]) + the
 * embedding source codes of SemanticTransformer.forward :709-714 ([start token | ids]; a negative id is the zero vector, :176-181) in one launch.
 * sem int64 [B][n0]; labels int64 [B][n0 + 1] = [ids | eos]; src_a int32 [B][n0 + 1] = [start | ids]; num_rows = rows of the embedding table (< 2^24).
 * has_eos != 0: the rows already are [ids | eos | pad ...]: labels [B][n0] = the rows, src_a [B][n0] = [start | rows without their last column]. */
int alm_semantic_prepare(const long long* sem, long long ld_sem, int B, int n0, long long eos_id, long long num_rows, long long* labels, int* src_a,
                         int has_eos, void* stream);
/* batch_unique_consecutive, audiolm_pytorch.py:162-164 (per row torch.unique_consecutive, right-padded with `pad` to the longest row), as one launch:
 * out int64 [B][n + append_eos] holds every collapsed row followed by pad, lengths int32 [B] the kept ids per row -- the caller reads the lengths ONCE and
 * slices out[:, :max(lengths)] (the reference's host loop synchronises per row).  append_eos != 0: the rows are [ids | eos_id] (append_eos_id :155-160, which
 * the wrappers apply before the collapse, :1536-1539 / :1788-1795). */
int alm_unique_consecutive_i64(const long long* ids, long long ld, int B, int n, int append_eos, long long eos_id, long long pad, long long* out,
                               long long ld_out, int* lengths, void* stream);
/* FineTransformer.forward's id bookkeeping, audiolm_pytorch.py:1171-1223 (key mask of pad / eos coarse keys, their ids zeroed, mask padded over
 * [coarse start | coarse | fine start | fine], embedding source codes id + (i mod Q) * codebook_size per table) in one launch.  fine: the first nf ids of
 * each row (the training wrapper drops the last one, :2086).  src_a int32 [B][n + nf + 2], keep bool [B][n + nf + 2]. */
int alm_fine_prepare(const long long* coarse, long long ld_coarse, const long long* fine, long long ld_fine, int B, int n, int nf, long long pad_id,
                     long long eos_id, int Qc, int Qf, int C, int* src_a, void* keep, void* stream);

/* forgetful causal mask, audiolm_pytorch.py:82-89 (`rand[:, 0] = -max; mask = ~zeros.scatter(1, rand.topk(k).indices, 1)`): keep [B][N] bytes (torch.bool
 * storage) &= NOT(one of the `drop` largest scores of its row); column 0 is never dropped; equal scores at the threshold go in index order.
 * drop <= N - 1, N <= 16384. */
int alm_forgetful_mask(const float* score, long long ld_score, void* keep, long long ld_keep, int B, int N, int drop, void* stream);

/* ---- hyper-connection residual streams: third-party `hyper_connections` used at audiolm_pytorch.py:24, 446-454, 524, 551 ----
 * R [B][S][N][D], stored fp32 or (r_bf16) bf16 -- under trainer.py:1241's autocast the reference's streams are bf16 tensors; the arithmetic is
 * fp32 in registers either way.  Tensors standing for all streams at once (the `*_bcast` forms) are always fp32 [B*N][D].
 * coef: per-token fp32 record of alm_hc_coef_width(S) floats (alpha | beta | pre-activations | 1/norm). */
int alm_hc_coef_width(int S);
int alm_hc_partial_width(int S, int D);
int alm_hc_grads_width(int S, int D);
int alm_hc_partial_rows(int mode, int fused_ln, int r_bf16, int S, long long tokens, int D);
/* forward.  mode 1: depth connection only, R_out[t] = sum_s alpha[s][t+1] R_in[s] + beta[t] y_prev (coef_prev = that branch's record);
 * mode 2: width connection of a branch (its 7 parameters) + the branch's pre-LayerNorm: x, xn = LN(x) ln_gamma, mean, rstd, coef;
 * mode 3: mode 1 of the previous branch fused with mode 2 of the next one on the freshly computed residual (one pass over R);
 * mode 5: depth connection + stream sum (:551) + final LayerNorm (:555): xs_out fp32 [B*N][D], mean, rstd and the LayerNorm output either as
 *         xn_out bf16 or -- xn32_out != NULL -- as fp32 [B*N][D] (what the logit heads read).
 * rin_bcast: R_in is ONE fp32 [B*N][D] tensor that every stream equals (the state right after the stream expansion, :524). */
int alm_hc_fwd(const void* R_in, int rin_bcast, int r_bf16, const void* y_prev_bf16, long long ldy, const float* coef_prev, void* R_out,
               const float* hc_gamma, const float* Wa, const float* sa, const float* Aa, const float* wb, const float* sb, const float* Bb,
               const float* ln_gamma, void* x_out_bf16, long long ldx, void* xn_out_bf16, long long ldxn, float* xn32_out, float* mean,
               float* rstd, float* coef, float* xs_out, int mode, int B, int S, int N, int D, void* stream);
/* backward.  mode 2: width-connection backward (dR, parameter-gradient partial rows); mode 1: depth-connection backward
 * (dy = sum_t beta[t] dRn[t], dbeta_out[t] = <dRn[t], y>); mode 3: mode 2 of branch k+1 fused with mode 1 of branch k on the
 * freshly computed dR.  dRn_bcast: dRn is fp32 [B*N][D] and stands for all S streams (gradient of the final stream sum).
 * r_bcast: R is one fp32 [B*N][D] tensor for all streams (first branch); dsum (optional): fp32 [B*N][D] sum over streams of dR = the gradient of
 * the stream expansion (:524); dR may then be NULL.  r_bf16: dRn / R / dR (the non-bcast forms) hold bf16.
 * The gradient wrt the branch input comes either as dx (fp32 [B*N][lddx], already through the branch's LayerNorm backward) or -- fused
 * mode, dx == NULL -- as dxn (bf16, gradient wrt the LayerNorm OUTPUT) + optional extra (bf16, added to dx directly: the K/V path of
 * the attention branch) + the LayerNorm statistics / weight: the LayerNorm backward (audiolm_pytorch.py:191-198 autograd) then happens
 * inside this kernel and its weight gradient joins the outputs.
 * partial: [alm_hc_partial_rows(mode, dx == NULL, r_bf16, S, B*N, D)][alm_hc_partial_width(S, D)] floats -> alm_colsum_partial (`chunks` partial sums) -> alm_hc_param_grads(sums, chunks, ...), whose
 * output is dWa[D][S+1] | dwb[D] | dgamma[D] | dAa[S][S+1] | dBb[S] | dsa | dsb | dln[D]  (alm_hc_grads_width(S, D) floats). */
int alm_hc_bwd(const void* dRn, int dRn_bcast, int r_bf16, const float* dx, long long lddx, const void* dxn_bf16, long long lddxn,
               const void* extra_bf16, long long ldex, const float* mean, const float* rstd, const float* ln_gamma, const void* R, int r_bcast,
               const float* coef, const float* dbeta, const float* hc_gamma, const float* Wa, const float* sa, const float* wb, const float* sb,
               void* dR, float* dsum, float dsum_scale, float* partial, const void* y_prev_bf16, long long ldy, const float* coef_prev, void* dy_bf16, long long lddy,
               float* dbeta_out, int mode, int B, int S, int N, int D, void* stream);
int alm_hc_param_grads(const float* sums, int chunks, const float* hc_gamma, const float* Wa, const float* wb, float* out, int S, int D, void* stream);
/* the same finish for nb <= 16 width connections in two launches (column sums of every problem's partial rows, then the parameter gradients): the deferred
 * mode of the fused backward keeps the partial rows of all 12 branches of a 6-layer stack and finishes them together (24 small launches -> 2).  HOST pointer
 * arrays; ws: nb * 16 * alm_hc_partial_width(S, D) floats; outs[z]: alm_hc_grads_width(S, D) floats. */
int alm_hc_param_grads_batched(const float* const* parts, const int* rows, int nb, const float* const* gammas, const float* const* Was,
                               const float* const* wbs, float* ws, float* const* outs, int S, int D, void* stream);
int alm_streams_expand(const float* x, float* R, int B, int S, long long nd, void* stream);   /* :524 */
int alm_streams_reduce(const float* R, float* x, int B, int S, long long nd, void* stream);   /* :551 */
int alm_residual_add(const float* x, const void* y_bf16, long long ldy, float* out, long long rows, int D, void* stream);
int alm_f32_to_bf16(const float* a, const float* b_or_null, void* out_bf16, long long ldo, long long rows, int D, void* stream);
int alm_add_f32(const float* a, const float* b, float* out, long long n, float scale, void* stream);   /* out = (a + b) * scale */

/* ---- token-id side: embedding assembly (:709-713, :894-918, :1186-1223), logit-head regrouping (:965-983, :1325-1361),
 *      cross-entropy (:1561-1565, :1839-1849, :2122-2132) -------------------------------------------------------------------- */
/* source codes: (table << 24) | row, -1 = zero vector.  table_rows[t] = number of rows of table t: a code whose table or row is out of range
 * (nn.Embedding raises IndexError for it, audiolm_pytorch.py:709 / :901-906 / :1204-1213) reads as a zero vector, is skipped by the scatter, and
 * sets *err_flag (device int32, may be NULL) to 1 -- never an out-of-bounds access. */
int alm_embed_assemble(const float* const* tables, const int* table_rows, int ntables, const int* src_a, const int* src_b, float* out,
                       long long rows, int D, int* err_flag, void* stream);
int alm_embed_scatter_add(float* const* grad_tables, const int* table_rows, int ntables, const int* src_a, const int* src_b, const float* dout,
                          float alpha, long long rows, int D, void* stream);
/* the same scatter with ONE OWNER per destination row and fixed summation order: no atomics, bitwise run-to-run deterministic gradients (the
 * reference's embedding backward, aten::embedding_dense_backward under audiolm_pytorch.py:709 / :901-918, is not; SURVEY section 5 asks for
 * reproducible runs).  Every row of every table is WRITTEN (zero where no token maps to it): the gradient buffers need no clearing.
 * Skewed ids: destination rows with more than 128 tokens are listed by a histogram (integer atomics: exact) and summed by (row, token-range) workers whose
 * partials the row's last arriver adds in range order -- the sums stay a fixed function of the code arrays.  ALM_EMBED_SCATTER_HOT=0: owners only (A/B).
 * ws: alm_embed_scatter_ws_floats(...) floats (chunk partials of the few-row tables + hot-row partials + the listing's int workspace; the call zeroes what
 * it needs).  rows < 2^28, D % 4 == 0, ws 16-byte aligned. */
int alm_embed_scatter_ws_floats(const int* table_rows, int ntables, long long rows, int D);
int alm_embed_scatter_owned(float* const* grad_tables, const int* table_rows, int ntables, const int* src_a, const int* src_b, const float* dout,
                            float alpha, long long rows, int D, float* ws, void* stream);
int alm_gather_rows_bf16(const void* in, long long ld_in, const int* idx, void* out, long long ld_out, long long rows, int D, void* stream);
int alm_scatter_rows_bf16(const void* in, long long ld_in, const int* idx, void* out, long long ld_out, long long rows, int D, void* stream);
/* split-bf16 operands of the logit-head contraction (the heads read the fp32 final hidden states / fp32 master weights at ~16 mantissa bits:
 * x = hi + lo with hi = bf16(x), lo = bf16(x - hi); logits = hi.Whi + hi.Wlo + lo.Whi on the bf16 MFMA, fp32 accumulate):
 * hi[r] / lo[r] = split(in[idx ? idx[r] : r]) for r < rows_out; source rows < 0 or >= rows_in give zero rows (ragged head groups, row padding). */
int alm_gather_split_bf16(const float* in, long long ld_in, long long rows_in, const int* idx, void* hi, void* lo, long long ld_out,
                          long long rows_out, int D, void* stream);
int alm_cross_entropy_fwd(const float* logits, long long ld, const long long* labels, float* loss_rows, float* lse, long long rows, int C,
                          int ignore_index, void* stream);
int alm_cross_entropy_bwd(const float* logits, long long ld, const long long* labels, const float* lse, const float* gscale, void* dlogits_bf16,
                          long long ldd, long long rows, int C, int Cpad, int ignore_index, void* stream);
int alm_reduce_sum(const float* in, long long n, float* out, float scale, void* stream);

/* ---- autoregressive sampling: attention of ONE new position per sequence over a key / value cache (Attention.forward with kv_cache,
 * audiolm_pytorch.py:360-394, driven by the generate() loops :1476-1507, :1677-1706, :1965-1994).  cache bf16 [B][nmax][128] (k | v, v already
 * value-residual mixed); kv_new bf16 [B][ldkv] is appended at index `pos` by the call; keys 0 .. pos are attended (mask uint8 [B][ldm], 1 =
 * attend, or NULL).  Structured score bias as in alm_mqa_attn_bias_fwd: tbl [H][LT] + per-key kkey4 / kattr (device) and the new position's
 * qkey4 / qattr BY VALUE; tbl NULL = none.  pos_dev (device int32) non-NULL: the position is read from the device at run time and the
 * query-side values from qkey4_vec / qattr_vec [nmax] -- the launch a captured hipGraph replays for every sampling step. */
int alm_mqa_decode_attn(const void* q, long long ldq, void* cache, long long cache_stride, const void* kv_new, long long ldkv,
                        const unsigned char* mask, long long ldm, void* out, long long ldo, int B, int H, int dim_head, int pos, int nmax,
                        float scale, const float* tbl, int LT, int qkey4, int qattr, const int* kkey4, const int* kattr, const int* pos_dev,
                        const int* qkey4_vec, const int* qattr_vec, void* stream);

/* ---- fused optimiser step: global-norm clip (trainer.py:953-954 accelerator.clip_grad_norm_) + Adam / AdamW (optimizer.py:get_optimizer) over
 * every parameter in two kernels.  `tensors`: HOST array of `ntensors` AlmOptTensor (device pointers inside): it travels in the kernel arguments, 64
 * tensors per launch -- no staged host-to-device copy per step (the gradient storage moves every step, so the table cannot live on the device);
 * `chunks`: DEVICE int32 pairs (tensor index, chunk index), tensor-major, one per alm_opt_chunk_elems() elements of each tensor (shape-dependent: cached).  alm_opt_grad_sumsq writes one partial sum of squares per chunk (alm_reduce_sum over it =
 * the squared global gradient norm, kept on the device); alm_opt_adam_step applies coef = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) to the
 * gradients on the fly (sumsq NULL: no clipping), then torch.optim.Adam's update (decoupled_weight_decay 1: AdamW).  All fp32. */
typedef struct AlmOptTensor {
    void* p; const void* g; void* m; void* v;   /* parameter, gradient, exp_avg, exp_avg_sq */
    long long n;                                /* elements */
    float wd; int step;                         /* weight decay of this tensor; its own 1-based step count of THIS update (bias corrections), 0: the launch's `step` */
} AlmOptTensor;
int alm_opt_chunk_elems(void);
int alm_opt_grad_sumsq(const AlmOptTensor* tensors, int ntensors, const int* chunks, int nchunks, float* partial, void* stream);
int alm_opt_adam_step(const AlmOptTensor* tensors, int ntensors, const int* chunks, int nchunks, float lr, float beta1, float beta2, float eps, int step,
                      int decoupled_weight_decay, const float* sumsq, float max_norm, void* stream);

/* The same update for the dense GEMM weights, fused with their bf16 re-pack (round 4): walks each weight matrix in alm_pack_weights_multi's tiles and
 * writes the new fp32 parameter, both moments and both packed images (dst / dstT as in AlmPackJob) -- the forward after an optimiser step then finds its
 * packed copies current (core.layer_weights) instead of re-reading every master weight.  `jobs`: HOST array.  Bit-identical to alm_opt_adam_step
 * followed by alm_pack_weights_multi.  Needs even leading dimensions / paddings and 8-byte (fp32) / 4-byte (bf16) aligned bases: ALM_ERR_BAD_ARG otherwise. */
typedef struct AlmOptPackJob {
    void* p; const void* g; void* m; void* v;   /* fp32 [rows][cols] views, row stride ld: parameter, gradient, exp_avg, exp_avg_sq */
    int rows, cols; long long ld;
    void* dst; long long ld_dst; int rows_pad, cols_pad;
    void* dstT; long long ld_dstT;
    float wd; int step;                         /* as AlmOptTensor */
} AlmOptPackJob;
int alm_opt_adam_pack_step(const AlmOptPackJob* jobs, int njobs, float lr, float beta1, float beta2, float eps, int step, int decoupled_weight_decay,
                           const float* sumsq, float max_norm, void* stream);

/* ---- SoundStream tokenize path (encode only): soundstream.py:332-345, 362-380, 519-531 (causal conv encoder), :592-607 / :840 ----
 * (eval-mode GroupedResidualVQ of vector-quantize-pytorch, restated in oracle/rvq_restated.py).  All fp32: the code indices are an
 * argmin over float distances, so both kernels run on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), never bf16.
 * conv: x [B][Cin][Tin] -> out [B][Cout][Tout], Tout = (Tin - stride) / stride + 1; left reflect pad dilation*(ksize-1) + 1 - stride;
 *       out = act(conv + bias) (+ residual), act = ELU when `elu`.  wp = weights packed once by alm_conv1d_pack. */
int alm_conv1d_packed_floats(int Cout, int Cin, int ksize);
int alm_conv1d_pack(const float* w, float* wp, int Cout, int Cin, int ksize, void* stream);
int alm_conv1d_causal(const float* x, const float* wp, const float* bias, const float* residual, float* out, int B, int Cin, int Cout, int Tin,
                      int ksize, int stride, int dilation, int elu, int zero_pad, void* stream);
/* one ResidualUnit (soundstream.py:362-369: x + ELU(conv_k1(ELU(conv_k,dilation(x)))), CausalConv1d reflect left pad, stride 1) in ONE launch; the
 * intermediate stays in registers.  w7p / w1p: alm_conv1d_pack images of the two conv weights; x, out fp32 [B][C][T].  C % 32 == 0 and C <= 256, else
 * ALM_ERR_UNSUPPORTED (run the two alm_conv1d_causal launches).  Bitwise equal to those two launches (same fma chains on the exact-fp32 matrix core). */
int alm_resunit_causal(const float* x, const float* w7p, const float* b7, const float* w1p, const float* b1, float* out, int B, int C, int T,
                       int ksize, int dilation, void* stream);
/* decoder side (soundstream.py:347-360, 382-395, 615-627, 691-709).  zero_pad = 1 above pads with zeros instead of reflecting: a
 * CausalConvTranspose1d(k = 2 s, stride s) is that conv with k = 2 over s * Cout phase-major output channels (weights re-indexed on the host),
 * followed by alm_phase_interleave: y [B][s * Cout][n] -> out [B][Cout][n * s].  alm_rvq_decode: codes -> summed code vectors. */
int alm_phase_interleave(const float* y, float* out, int B, int Cout, int s, int n, void* stream);
int alm_rvq_decode(const long long* idx, long long ldi, const float* E, float* out, long long ldo, int T, int d, int C, int Q, void* stream);
/* rvq: frames x [T][ldx] (d columns of one group), codebooks E [Q][C][d]; Et [Q][alm_rvq_padded_dim(d)][CP] floats (MFMA-ordered image)
 * / e2 [Q][CP] packed once by alm_rvq_pack (CP = alm_rvq_padded_codes(C)); idx int64 [T][ldi] (Q columns): per quantizer argmin_e sqrt(clamp(|r|^2 + |e|^2 - 2 r.e, 0)) with
 * first-index tie-breaking, r -= E[idx]; quant (optional) = sum of the selected code vectors. */
int alm_rvq_padded_codes(int C);
int alm_rvq_padded_dim(int d);
int alm_rvq_pack(const float* E, float* Et, float* e2, int Q, int C, int d, void* stream);
int alm_rvq_encode(const float* x, long long ldx, const float* E, const float* Et, const float* e2, long long* idx, long long ldi, float* quant,
                   long long ldq, int T, int d, int C, int Q, void* stream);
int alm_bct_to_btc(const float* in, float* out, int B, int C, int T, void* stream);   /* 'b c n -> b n c', soundstream.py:823 */
/* SoundStream LocalTransformer (soundstream.py:397-440 = local-attention's LocalMHA + FeedForward; third-party, restated), fp32, in the codec's
 * [B][C][T] layout; the Linear layers are k = 1 alm_conv1d_causal calls.
 *   alm_layernorm_bct : nn.LayerNorm over the channel axis (weight gamma, bias beta)
 *   alm_geglu_bct     : out[b][i][t] = x[b][i][t] * gelu(x[b][I + i][t])                          (x [B][2 I][T])
 *   alm_local_attn    : qkv [B][3 H dh][T] (q | k | v channel blocks) -> out [B][H dh][T]: l2-normalised q / k times q_scale / k_scale [dh]
 *                       (qk_rmsnorm) and `scale`, rotary + xpos from the slot tables cos_t / sin_t / xpos_t [2 window][dh] (slot s of the
 *                       (look-back | own) window pair; queries sit in slots window .. 2 window - 1), key j visible to query i iff
 *                       0 <= i - j <= window, softmax, values, times sigmoid(gates [B][H][T]) (gates may be NULL).  dh in {32, 64}, window <= 256. */
int alm_layernorm_bct(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int T, float eps, void* stream);
int alm_geglu_bct(const float* x, float* out, int B, int I, int T, void* stream);
int alm_local_attn(const float* qkv, const float* q_scale, const float* k_scale, const float* cos_t, const float* sin_t, const float* xpos_t,
                   const float* gates, float* out, int B, int H, int dim_head, int T, int window, float scale, void* stream);

/* ---- launch lists (round 6): a recorded sequence of the launches above re-issued by ONE host call ---------------------------------------------
 * The depth loop of audiolm_pytorch.py:528-547 (Transformer.forward) and its backward are ~190 launches per training step whose ORDER and scalar arguments
 * depend only on the model configuration and the batch shape; only buffer addresses change between steps.  The host records the sequence once per shape
 * (audiolm-pytorch_amd/launchlist.py: every pointer argument classified as base + byte offset against a table of bases -- the step's activation arena, the
 * inputs, every parameter, every packed weight image) and alm_list_run re-issues it: the same entry points, in the same order, on `stream` -- the results are
 * bit-identical to issuing them one by one.  Nothing is baked (unlike a hipGraph replay): addresses, the stream and the weight images are live.
 *   entries[e] = {op (alm_list_op_id of the entry point), nargs (must equal alm_list_op_nargs), first (index of its first slot)}
 *   slots[i]   = one 64-bit value per C argument (int / long long sign-extended, float = its bit pattern in the low word, pointer = byte offset)
 *   reloc[i]   = ALM_LIST_LITERAL: slots[i] as is;  1 .. nbases: bases[reloc - 1] + slots[i];  ALM_LIST_STREAM: the `stream` argument;
 *                ALM_LIST_HOST_PTRS: a HOST array of pointers = the resolved slots starting at index slots[i];  ALM_LIST_HOST_INTS: a HOST array of ints
 *                packed into the slots starting at index slots[i] (alm_hc_param_grads_batched takes both)
 * Returns 0, or the first failing launch's code with its index in *failed_at (launches before it were issued).  alm_memset_zero: zero-fill by a kernel (a small hipMemsetAsync node of a captured hipGraph is not replayed correctly on ROCm 7.0: scripts/debug/memset_node_probe.py). */
typedef struct { int op; int nargs; int first; int reserved; } AlmListEntry;
#define ALM_LIST_LITERAL 0
#define ALM_LIST_STREAM 0xFFFF
#define ALM_LIST_HOST_PTRS 0xFFFE
#define ALM_LIST_HOST_INTS 0xFFFD
int alm_memset_zero(void* ptr, long long bytes, void* stream);
int alm_list_op_id(const char* name);
int alm_list_op_nargs(int op);
int alm_list_run(const AlmListEntry* entries, int n, const unsigned long long* slots, const unsigned short* reloc, int nslots,
                 const unsigned long long* bases, int nbases, void* stream, int* failed_at);

#ifdef __cplusplus
}
#endif
