/*
 * qfx.h -- C ABI of libqfx.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * LoRA-training hot path of the Qwen-Image-Edit / FLUX-Kontext DiT.
 *
 * The reference (tsiendragon/qwen-image-finetune) has no native code and no FFI: its hot path is
 * Python calling torch/diffusers/peft ops.  Each entry point below names the
 * reference call sites whose device work it replaces (file:line relative to /root/reference).
 * Callers are the torch.autograd.Function wrappers in qwen-image-finetune_amd/qflux_amd/ops.py.
 *
 * Conventions
 *   - extern "C", plain device pointers + explicit sizes/strides (in ELEMENTS unless stated).
 *   - bf16 tensors are passed as `const uint16_t*` (raw bits), fp32 as `const float*`.
 *   - caller allocates every output and workspace; kernels are launched on `stream`
 *     (a hipStream_t passed as void*); no global state; thread-safe for distinct streams.
 *   - return 0 on success, a negative QFX_E* code on a rejected argument; HIP launch errors are
 *     returned as -(1000 + hipError_t).  No exceptions cross the ABI.
 */
#ifndef QFX_H
#define QFX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QFX_ABI_VERSION 7

#define QFX_OK 0
#define QFX_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define QFX_EUNSUPPORTED (-2)

/* ---- GEMM with fused LoRA / epilogues ------------------------------------------------------
 * C[M,N] = A1[M,K1] * B1[N,K1]^T (+ A2[M,K2] * B2[N,K2]^T) (+ bias[N]) , then epilogue.
 * Replaces: nn.Linear + peft lora.Linear side branch + following elementwise ops
 *   (src/qflux/models/transformer_qwenimage.py:286-293,348,352 ; FeedForward at :479,485 ;
 *    gated residuals at :473-474,480,486 ; LoRA call site src/qflux/trainer/base_trainer.py:929-941)
 * and, with transposed weights, the dX GEMMs autograd runs for them.
 * K1, K2 multiples of 64; all row strides multiples of 8 elements; pointers 16-byte aligned.
 * Rounding points follow the reference's bf16 eager graph: if K2>0 the accumulator is rounded to
 * bf16 after (A1*B1^T + bias) -- the base nn.Linear output -- before the LoRA segment is added.
 */
enum {
  QFX_EPI_NONE = 0,      /* C = bf16(acc) */
  QFX_EPI_GELU = 1,      /* C = bf16(acc) (pre-activation), C2 = bf16(gelu_tanh(C)) */
  QFX_EPI_GATE_RES = 2,  /* C = bf16(aux + bf16(gate[b] * bf16(acc)))  (x + gate*y); if C2 != NULL also C2[m] = bf16(acc) (pre-gate y, rows unmapped, for d(gate)) */
  QFX_EPI_DGELU = 3      /* C = bf16(bf16(acc) * gelu_tanh'(aux))      (backward through GELU) */
};

typedef struct qfx_gemm_args {
  const uint16_t* A1; const uint16_t* B1; int64_t lda1; int64_t ldb1; int32_t K1;
  const uint16_t* A2; const uint16_t* B2; int64_t lda2; int64_t ldb2; int32_t K2;
  int32_t M; int32_t N;
  const uint16_t* bias;            /* [N] bf16 or NULL */
  uint16_t* C; int64_t ldc;
  uint16_t* C2; int64_t ldc2;      /* EPI_GELU second output */
  const uint16_t* aux; int64_t ldaux;   /* residual (GATE_RES) / pre-activation (DGELU); indexed like C */
  const uint16_t* gate; int64_t gate_bstride; /* gate[b*gate_bstride + n], b = m / rows_per_batch */
  int32_t rows_per_batch;          /* rows of A per batch sample (M if unused) */
  /* row remaps (joint [text|image] buffers): row(m) = (m / rows_per_batch) * X_batch_rows + X_row_off + m % rows_per_batch */
  int32_t a_batch_rows; int32_t a_row_off;   /* applied to A1 rows; a_batch_rows==0 => identity */
  int32_t c_batch_rows; int32_t c_row_off;   /* applied to C/C2/aux rows; c_batch_rows==0 => identity */
  int32_t epi;
  const float* row_mask;           /* optional [M] (compact row m): rows with mask 0 are written as exact zeros (multi-resolution
                                      padding, transformer_flux_custom.py:427-442,648-660,724-733); NULL = no masking */
  int32_t aux_unmapped;            /* 1: aux rows are indexed by m (compact) even when C uses the c_* row remap */
  int32_t seg2_plain;              /* 1: the second K segment is an ordinary continuation of the contraction (FLUX single block:
                                      [attn | mlp] @ W_out as two segments) -- no bf16 mid-rounding, bias added at the end */
} qfx_gemm_args;

int qfx_gemm_bf16(const qfx_gemm_args* args, void* stream);

/* Grouped launch: n <= QFX_MAX_GROUPS independent problems with the same epilogue kind in ONE grid
 * (image + text stream, q/k/v projections): the small text-stream GEMMs share the machine with the
 * image-stream ones instead of running latency-bound on a few dozen tiles. `groups` is a HOST array. */
#define QFX_MAX_GROUPS 6
int qfx_gemm_grouped(const qfx_gemm_args* groups, int32_t n, void* stream);

/* ---- low-precision trunk: MX-FP8 base GEMM (reference: src/qflux/models/quantize.py -- TE fp8 / bnb int8 / NF4 --
 * whose MI355X analogue is the block-scaled FP8 MFMA, v_mfma_scale_f32_16x16x128_f8f6f4, at twice the bf16 matrix rate).
 * OCP MX format: elements fp8 e4m3 (OCP e4m3fn), one shared E8M0 scale per 32 consecutive K elements,
 *   scale exponent = floor(log2(max|v|)) - 8 (e4m3 emax), elements = RNE(v / 2^e) saturated to +-448.
 * qfx_quant_mxfp8: X[M,K] bf16 (row stride ldx, optional joint-buffer row remap) -> Q[M,K] fp8 bytes (row stride ldq) and the
 *   E8M0 scale bytes S in TILE-MAJOR order [K/128][M][4]: the scale of row m, 32-element block kb sits at byte
 *   ((kb/4)*M + m)*4 + kb%4, so that the 16 rows of an MFMA fragment read one contiguous 64-byte line per K tile.  K % 128 == 0.
 *   (`lds` / `ldsa` / `ldsb` are unused, kept for ABI stability.)
 * qfx_gemm_mxfp8: same contract as qfx_gemm_bf16 (bias, bf16 mid-rounding of the base output, bf16 LoRA K-extension segment
 *   A2/B2, all four epilogues, row maps of C / aux) except that A1 / B1 are MX-FP8: g.A1 / g.B1 point at fp8 BYTES, lda1 / ldb1
 *   are byte strides, K1 % 128 == 0, A1 rows are NOT remapped; sa / sb are the tile-major scale arrays of A1 ([K1/128][M][4]) and
 *   B1 ([K1/128][N][4]).
 *   The product of two e4m3 values and a power-of-two scale is exact in fp32; accumulation is fp32 as in the bf16 kernel. */
typedef struct qfx_quant_args {
  const uint16_t* X; int64_t ldx; int32_t M; int32_t K;
  uint8_t* Q; int64_t ldq; uint8_t* S; int64_t lds;
  int32_t rows_per_batch; int32_t x_batch_rows; int32_t x_row_off;
} qfx_quant_args;
int qfx_quant_mxfp8(const qfx_quant_args* a, void* stream);
typedef struct qfx_gemm_fp8_args {
  qfx_gemm_args g;
  const uint8_t* sa; int64_t ldsa;
  const uint8_t* sb; int64_t ldsb;
  /* ABI 2: quantising epilogue (persistent kernel only: problems of >= 160 256x128 tiles, N % 128 == 0, not GATE_RES).  When
   * cq != NULL the output the NEXT GEMM contracts over -- C2 = gelu(h) for EPI_GELU, C otherwise -- also leaves the epilogue as
   * MX-FP8: bytes at cq + row*ldcq + n (rows indexed like C), E8M0 scales tile-major in cs ([N/128][cq_rows][4], cq_rows >= the
   * row extent of C); bit-identical to qfx_quant_mxfp8 of the bf16 tensor.  cq_only != 0: the bf16 copy of that output is not
   * written at all (the consumer is an MX-FP8 GEMM and nothing else reads it). */
  uint8_t* cq; uint8_t* cs; int64_t ldcq; int32_t cq_rows; int32_t cq_only;
} qfx_gemm_fp8_args;
int qfx_gemm_mxfp8(const qfx_gemm_fp8_args* a, void* stream);
/* n <= QFX_MAX_GROUPS problems with the same epilogue in ONE persistent grid (image + text stream, q/k/v), as qfx_gemm_grouped;
 * no seg2_plain.  qfx_gemm_mxfp8 routes large single problems here and keeps the 128x128 kernel for small ones. */
int qfx_gemm_mxfp8_grouped(const qfx_gemm_fp8_args* list, int32_t n, void* stream);

/* ---- LoRA rank-r down projection ("skinny" GEMM, HBM-bound) ----------------------------------
 * U[M,R] (fp32) = X[M,K] (bf16) * (W_hi + W_lo)[R,K]^T   (W = fp32 LoRA weight split in two bf16)
 * and its packed bf16 image EXT[M, 3R] = [U_hi | U_lo | U_hi] written at ext + m*ld_ext
 * (consumed as the A2 operand of qfx_gemm_bf16 against B2 = [V_hi | V_hi | V_lo]).
 * Replaces peft lora_A(x.float()) forward and the dY*B product of its backward. R in {16,32,48,64,96}.
 */
typedef struct qfx_lora_down_args {
  const uint16_t* X; int64_t ldx; int32_t M; int32_t K;
  const uint16_t* W_hi; const uint16_t* W_lo; int64_t ldw; int32_t R;
  float* U; int64_t ldu;                 /* may be NULL */
  uint16_t* ext; int64_t ld_ext;         /* may be NULL */
  uint16_t* Ut_hi; uint16_t* Ut_lo; int64_t ld_ut; /* may be NULL: transposed split image Ut[R][ld_ut] (column = row index m),
                                            the V operand of qfx_lora_grad; columns >= M must be pre-zeroed by the caller */
  int32_t group_R; int32_t group_stride; /* ext column of U column j: (j/group_R)*group_stride + j%group_R + {0,group_R,2*group_R}
                                            (several LoRA targets sharing X are fused in one call; group_R = R for one) */
  int32_t rows_per_batch; int32_t x_batch_rows; int32_t x_row_off; /* X row remap as in gemm */
  /* ABI 2, MX-FP8 trunk: xq != NULL -- the kernel reads every element of X anyway; it also writes X's MX-FP8 image for the GEMM that
   * contracts over it: bytes at xq + m*ldxq + k (m = compact row), E8M0 scales tile-major as qfx_quant_mxfp8 writes them, the
   * 32-column block ks of this call being block xq_kb0 + ks of an operand with xs_rows rows (several calls fill the column
   * sections of one operand: q / k / v sections of dqkv).  K % 128 == 0, xq_kb0 % 4 == 0. */
  uint8_t* xq; uint8_t* xs; int64_t ldxq; int32_t xs_rows; int32_t xq_kb0;
} qfx_lora_down_args;

int qfx_lora_down(const qfx_lora_down_args* args, void* stream);

/* ABI 6: second half of a down projection fused into an attention epilogue (qfx_head_lora): U[m, j] = sum_h part[h][row(m)][j] in
 * head order (deterministic), then exactly qfx_lora_down's outputs -- EXT[m, ...] = [U_hi | U_lo | U_hi] per group and the transposed
 * split image Ut -- for the M compact rows of one stream (row(m) = the joint row, X-remap fields as in qfx_lora_down).  R = total
 * columns (e.g. 3 * Rp for q | k | v with group_R = Rp).  Up to 2 problems (image + text stream) per launch. */
typedef struct qfx_lora_head_reduce_args {
  const float* part; int64_t part_hstride; int32_t ld_part; int32_t H;
  int32_t M; int32_t R;
  uint16_t* ext; int64_t ld_ext;
  uint16_t* Ut_hi; uint16_t* Ut_lo; int64_t ld_ut;
  int32_t group_R; int32_t group_stride;
  int32_t rows_per_batch; int32_t x_batch_rows; int32_t x_row_off; int32_t reserved;
} qfx_lora_head_reduce_args;
int qfx_lora_head_reduce(const qfx_lora_head_reduce_args* list, int32_t n, void* stream);
/* n <= QFX_MAX_BATCH independent problems of the same R in ONE launch (e.g. the q, k and v down projections of a block's
 * backward): every separate launch of these one-round-trip kernels costs a dispatch gap plus its own latency floor. */
#define QFX_MAX_BATCH 8
int qfx_lora_down_batch(const qfx_lora_down_args* list, int32_t n, void* stream);

/* ---- LoRA weight gradients (contraction over tokens, MFMA + LDS transpose reads) -------------
 * G[j,k] += out_scale * sum_m (Vt_hi+Vt_lo)[j,m] * X[m,k]   j<R, k<K ; Vt = transposed bf16 split of the fp32
 * rank-r activations written by qfx_lora_down (rows zero-padded to a multiple of 32 tokens), X bf16 [M,K];
 * G fp32, atomically accumulated at G[j*g_sr + k*g_sc] (dA [r,K] uses (K,1), dB [N,r] uses (1,r)).
 * Up to three fused targets: rank j belongs to group j/group_R and goes to G, G1 or G2 (rank j%group_R,
 * written only if < r_valid).  Replaces autograd's dW for lora_A / lora_B (frozen base weights never get a dW).
 */
typedef struct qfx_lora_grad_args {
  const uint16_t* Vt_hi; const uint16_t* Vt_lo; int64_t ldvt; int32_t R; int32_t r_valid; int32_t group_R;
  const uint16_t* X; int64_t ldx; int32_t M; int32_t K;
  float* G; float* G1; float* G2; int64_t g_sr; int64_t g_sc;
  int32_t rows_per_batch; int32_t x_batch_rows; int32_t x_row_off;
  float out_scale;                       /* lora_alpha/r for dB, 1 for dA */
  /* ABI 7 (optional; NULL = the round-1..5 behaviour: the partial sums of the token chunks meet in G by fp32 atomics, order-dependent
   * in the last bit).  ws: fp32 scratch of ws_floats >= qfx_lora_grad_ws_floats(M, K, R) elements, ws_count: int32[(K + 127) / 128], ZERO before the
   * first launch (every launch leaves it zero).  With both set the chunk partials go to ws and are added up in CHUNK ORDER before G is
   * updated with plain stores: same inputs -> same bits.  The library does that in a second launch on the same stream (the kernel
   * boundary is the hand-off; ws_count is then not touched); a build with -DQFX_GRAD_HANDOFF=0 lets the last block to arrive at a
   * 128-column strip do it inside the one launch (tickets in ws_count) -- slower: profiles/r06_grad_handoff.json. */
  float* ws; int32_t* ws_count; int64_t ws_floats;      /* ws_floats = elements allocated behind ws: launches that need more are refused (QFX_EINVAL) */
} qfx_lora_grad_args;

int qfx_lora_grad(const qfx_lora_grad_args* args, void* stream);
int qfx_lora_grad_batch(const qfx_lora_grad_args* list, int32_t n, void* stream);   /* same R for all; see qfx_lora_down_batch */
int64_t qfx_lora_grad_ws_floats(int32_t M, int32_t K, int32_t R);   /* fp32 elements of qfx_lora_grad_args.ws for one problem (0: a single token chunk, no scratch needed) */

/* ---- LoRA operand packing (after every optimizer step) ---------------------------------------
 * From fp32 A[r,K], B[N,r] and scale s = lora_alpha/r build (Rp = r rounded up to 16):
 *   A_hi/A_lo [Rp,K] bf16, Bt_hi/Bt_lo [Rp,N] bf16 (= split of s*B^T),
 *   We  [N, Kext] = [sB_hi | sB_hi | sB_lo | 0]   (forward B2 operand)
 *   WeT [K, Kext] = [A_hi^T | A_hi^T | A_lo^T | 0] (dX B2 operand),  Kext = roundup(3*Rp, 64).
 */
typedef struct qfx_lora_pack_args {
  const float* A; const float* B; int32_t r; int32_t K; int32_t N; float scale;
  uint16_t* A_hi; uint16_t* A_lo; int64_t ld_a;      /* [Rp,K] rows at stride ld_a */
  uint16_t* Bt_hi; uint16_t* Bt_lo; int64_t ld_bt;   /* [Rp,N] rows at stride ld_bt */
  uint16_t* We; int64_t ld_we;                       /* [N,Kext] */
  uint16_t* WeT; int64_t ld_wet;                     /* [K,Kext] */
  int32_t Rp; int32_t Kext;
  /* ABI 6 (optional, NULL = off): head-fragment images for qfx_head_lora.  A_hl: (A_hi, A_lo) of an adapter whose INPUT is a
   * [*, H*hl_dh] attention output; Bt_hl: (Bt_hi, Bt_lo) of an adapter whose OUTPUT is a q / k / v row.  Element (row j, column
   * h*hl_dh + 32 ks + 16 db + 4 g + r) of the hi (sel = 0) / lo (sel = 1) split sits at
   *   ((((h * Rp/16 + j/16) * hl_dh/32 + ks) * 2 + sel) * 64 + 16 g + j%16) * 8 + 4 db + r        (bf16 elements). */
  uint16_t* A_hl; uint16_t* Bt_hl; int32_t hl_dh; int32_t reserved;
  /* ABI 6 (optional, NULL = off): the A rows of this adapter inside the MFMA-FRAGMENT image of a row group that a fused
   * LayerNorm + down projection reads (qfx_ln_down_args.W_fr): fr_nf = 16-row fragments of the whole group (q, k, v adapters
   * of one stream: 3 Rp / 16), fr_row0 = first group row of this adapter.  Element (group row j, column k) of the hi split sits at
   *   ((k/32 * fr_nf + j/16) * 64 + 16 * ((k%32)/8) + j%16) * 8 + k%8      (bf16 elements),
   * the lo split fr_nf * 16 * K elements further: a fragment of 16 rows x 32 columns is ONE lane-linear 1 KiB piece. */
  uint16_t* A_fr; int32_t fr_row0; int32_t fr_nf;
} qfx_lora_pack_args;

/* descs: DEVICE array of n descriptors (one per LoRA target); one launch packs them all. */
int qfx_lora_pack(const qfx_lora_pack_args* descs, int32_t n, int32_t max_dim, void* stream);

/* ---- LayerNorm (no affine, eps) + modulation ------------------------------------------------
 * y = bf16(bf16(bf16(LN(x)) * bf16(1+scale[b])) + shift[b])    (transformer_qwenimage.py:420-423,443-448,477-484;
 * AdaLayerNormContinuous at :662).  x,y [rows,D]; shift/scale [B,*] with batch stride mod_bstride.
 */
int qfx_ln_modulate_fwd(const uint16_t* x, const uint16_t* shift, const uint16_t* scale, int64_t mod_bstride,
                        uint16_t* y, int32_t rows, int32_t D, int32_t rows_per_batch, float eps, void* stream);
/* dx = bf16(dres + bf16(LN_bwd(dy * bf16(1+scale)))) ; optional dyg = bf16(gate[b] * dx) (input of the
 * previous gated-residual GEMM's backward). dres/gate/dyg may be NULL. */
int qfx_ln_modulate_bwd(const uint16_t* dy, const uint16_t* x, const uint16_t* scale, int64_t mod_bstride,
                        const uint16_t* dres, const uint16_t* gate, int64_t gate_bstride,
                        uint16_t* dx, uint16_t* dyg, int32_t rows, int32_t D, int32_t rows_per_batch,
                        float eps, const float* row_mask, void* stream);
/* row_mask (optional, [rows]): rows with mask 0 get dx = dyg = 0 (backward of the padded-token zeroing). */

/* Batched forms: n <= QFX_MAX_LN_BATCH problems in ONE launch (the image and the text stream of a block: the 384-row text
 * problem otherwise pays a dispatch gap and a memory round trip of its own).  Every problem but the last needs rows % 4 == 0. */
#define QFX_MAX_LN_BATCH 4
/* ABI 2, MX-FP8 trunk: yq / dygq != NULL -- the output (y resp. dyg) ALSO leaves the kernel as MX-FP8 (bytes at yq + row*ldyq,
 * E8M0 scales tile-major [D/128][ys_rows][4] as qfx_quant_mxfp8 writes them, bit-identical to quantising the bf16 output in a
 * separate pass; D % 128 == 0): the GEMM that consumes it skips its quantisation pass. */
typedef struct qfx_ln_fwd_args {
  const uint16_t* x; const uint16_t* shift; const uint16_t* scale; int64_t mod_bstride; uint16_t* y;
  int32_t rows; int32_t D; int32_t rows_per_batch; float eps;
  uint8_t* yq; uint8_t* ys; int64_t ldyq; int32_t ys_rows; int32_t pad_;
} qfx_ln_fwd_args;
typedef struct qfx_ln_bwd_args {
  const uint16_t* dy; const uint16_t* x; const uint16_t* scale; int64_t mod_bstride;
  const uint16_t* dres; const uint16_t* gate; int64_t gate_bstride; uint16_t* dx; uint16_t* dyg;
  const float* row_mask; int32_t rows; int32_t D; int32_t rows_per_batch; float eps;
  uint8_t* dygq; uint8_t* dygs; int64_t lddygq; int32_t dygs_rows; int32_t pad_;
} qfx_ln_bwd_args;
int qfx_ln_modulate_fwd_batch(const qfx_ln_fwd_args* list, int32_t n, void* stream);
int qfx_ln_modulate_bwd_batch(const qfx_ln_bwd_args* list, int32_t n, void* stream);
/* ---- fused LayerNorm+modulate forward AND the LoRA down projection of its output ("AdaLN fused into the projection
 * prologue", BASELINE north_star): y = bf16(LN(x) * bf16(1 + scale[b])) + shift[b] as qfx_ln_modulate_fwd, and in the same pass over
 * the row block u = y (W_hi + W_lo)^T with the K-extension image `ext` and the transposed split image `Ut` exactly as
 * qfx_lora_down writes them (same rounding points and MFMA order; the fp32 row statistics are summed in a different order than
 * in the one-wave-per-row kernel, so y may differ from it in the last bf16 ulp on rare elements).  Problems with
 * W_hi == NULL are plain LayerNorm+modulate rows (the un-adapted stream of a block rides in the same launch).
 * D % 256 == 0, D <= 3072, R in {16, 32, 48} and equal for all adapted problems of one call; n <= 2. */
typedef struct qfx_ln_down_args {
  qfx_ln_fwd_args ln;
  const uint16_t* W_hi; const uint16_t* W_lo; int64_t ldw; int32_t R;
  uint16_t* ext; int64_t ld_ext; uint16_t* Ut_hi; uint16_t* Ut_lo; int64_t ld_ut;
  int32_t group_R; int32_t group_stride;
  /* ABI 6 (optional): the same weights in MFMA-fragment order (qfx_lora_pack_args.A_fr: hi image, lo image R * D elements further).
   * Row-major weights are read as 16-byte pieces of 16 different rows per 16 lanes (64 line visits per wave load); the fragment image
   * as 8 whole lines: 37.4 -> 30.3 us per launch at the headline shape.  NULL: W_hi / W_lo are read. */
  const uint16_t* W_fr;
} qfx_ln_down_args;
int qfx_ln_down_fwd(const qfx_ln_down_args* list, int32_t n, void* stream);

/* ---- gradients of the AdaLN modulation vectors (needed only when the modulation linears carry adapters:
 * `img_mod.1` / `norm1.linear` ... in target_modules, configs/face_seg_flux_kontext_fp16.yaml:11, "all-linear").
 * For xm = bf16(LN(x) * bf16(1 + scale[b])) + shift[b]   (transformer_qwenimage.py:443-448; AdaLayerNormZero)
 * and  xo = x_res + bf16(gate[b] * y)                     (transformer_qwenimage.py:479-485)
 *   dshift[b] += sum_rows dy,   dscale[b] += sum_rows dy * bf16(LN(x)),   dgate[b] += sum_rows dxo * y
 * (fp32 accumulation, atomics into zero-initialised [B, out_bstride] buffers; rows of sample b = rows_per_batch consecutive
 * rows).  dxo / y / dgate may be NULL together (AdaLayerNormContinuous has no gate).  Row strides are explicit. */
typedef struct qfx_mod_grad_args {
  const uint16_t* dy; int64_t ld_dy; const uint16_t* x; int64_t ld_x;
  const uint16_t* dxo; int64_t ld_dxo; const uint16_t* y; int64_t ld_y;
  float* dshift; float* dscale; float* dgate; int64_t out_bstride;
  const float* row_mask; int32_t rows; int32_t D; int32_t rows_per_batch; float eps;
} qfx_mod_grad_args;
int qfx_mod_grad(const qfx_mod_grad_args* a, void* stream);
/* ABI 4: n <= QFX_MAX_LN_BATCH problems of one width class (ceil(D / 512) equal) in ONE launch -- the image and the text stream of a
 * block (the 384-row text problem otherwise pays a dispatch gap and a latency chain of its own for 48 blocks of work). */
int qfx_mod_grad_batch(const qfx_mod_grad_args* list, int32_t n, void* stream);

/* dyg = bf16(gate[b] * dx) only (used where no LayerNorm precedes). */
int qfx_gate_mul(const uint16_t* dx, const uint16_t* gate, int64_t gate_bstride, uint16_t* dyg,
                 int32_t rows, int32_t D, int32_t rows_per_batch, void* stream);

/* ---- RMSNorm over the last dim with learned weight (txt_norm, transformer_qwenimage.py:625) */
int qfx_rmsnorm_fwd(const uint16_t* x, const uint16_t* w, uint16_t* y, int32_t rows, int32_t D, float eps, void* stream);

/* ---- modulation GEMV: out[i][b][n] = bf16( sum_k bf16(silu(temb[b][k])) * W_i[n][k] + bias_i[n] )
 * for a list of nmat weight matrices (all [N,K]) -- every block's img_mod/txt_mod in one launch
 * (transformer_qwenimage.py:389-392,411-414,435-436; HBM-bound). */
int qfx_mod_gemv(const uint16_t* temb, int32_t B, int32_t K, const uint16_t* const* W, const uint16_t* const* bias,
                 int32_t nmat, int32_t N, int32_t apply_silu, uint16_t* out, void* stream);
/* W, bias: DEVICE arrays of nmat device pointers. apply_silu=0 gives a plain small-batch Linear
 * (timestep_embedder.linear_1). B <= 8. */
/* Backward of the same frozen linears w.r.t. their (shared) input, needed when the conditioning head carries adapters
 * (target_modules "all-linear", configs/example_with_sampling.yaml:9): out[b, k] (fp32 [B, K], zeroed by the caller) +=
 * sum_mat sum_n dy[mat, b, n] * W_mat[n, k], dy bf16 [nmat, B, N] -- what autograd computes as dy @ W per module and sums.
 * One streaming pass over the weights (two passes per pair of samples beyond B = 2).  K <= 3072, K % 8 == 0. */
int qfx_mod_gemv_t(const uint16_t* dy, int32_t B, int32_t N, int32_t K, const uint16_t* const* W, int32_t nmat, float* out,
                   void* stream);

/* ---- adapters on the conditioning head (target_modules "all-linear", configs/example_with_sampling.yaml:9; the
 * (norm|norm1|norm1_context).linear alternatives of configs/face_seg_flux_kontext_fp16.yaml:11): the rank-r side terms of linears
 * that see M = batch rows -- timestep / guidance / pooled-text embedders (transformer_qwenimage.py:143-156,
 * transformer_flux.py:634-639), AdaLN modulation linears (:389-392,411-414 / transformer_flux.py:391,446-447), norm_out.linear
 * (:565) -- for a BANK of `na` adapters of one rank that share the input x [B,K] (bf16).  peft lora.Linear semantics
 * (call site base_trainer.py:929-941):
 *   qfx_cond_lora_fwd:  u_a = A_a act(x) (fp32, kept in `u`);  y_a[b][n] = bf16(float(y_a[b][n]) + scale_a * sum_j B_a[n][j] u_a[b][j])
 *                       in place on the base outputs qfx_mod_gemv wrote (y: device array of na row pointers, row b at + b*ldy).
 *   qfx_cond_lora_bwd:  g_a = bf16 gradient rows of the outputs (device array of na pointers, row b at + b*ldg);
 *                       dB_a += (scale_a g_a)^T u_a;  du_a = (scale_a g_a) B_a;  dA_a += du_a^T act(x);  dx[b][k] += sum_a du_a A_a
 *                       (dx fp32 [B,K] = gradient w.r.t. act(x), may be NULL; du [na,B,r] fp32 scratch ZEROED by the caller;
 *                       dA / dB: device arrays of pointers into the flat LoRA gradient buffer, accumulated).
 * act = bf16(silu(.)) when apply_silu (the eager graph's F.silu on bf16), identity otherwise.  A_a [r,K], B_a [N,r] fp32
 * (device arrays of pointers to the adapter weights), scale: device [na].  B <= 8, r <= 64. */
typedef struct qfx_cond_lora_args {
  const uint16_t* x; int32_t B; int32_t K; int32_t apply_silu; int32_t na; int32_t r; int32_t N;
  const float* const* A; const float* const* Bm; const float* scale;
  float* u;
  uint16_t* const* y; int64_t ldy;
  const uint16_t* const* g; int64_t ldg;
  float* const* dA; float* const* dB;
  float* du; float* dx;
} qfx_cond_lora_args;
int qfx_cond_lora_fwd(const qfx_cond_lora_args* a, void* stream);
int qfx_cond_lora_bwd(const qfx_cond_lora_args* a, void* stream);
/* out = bf16(in): the fp32 column sums of the HIP backward (qfx_mod_grad) handed to the head's backward as the bf16 gradients
 * autograd would see */
int qfx_cast_f32_bf16(const float* in, uint16_t* out, int64_t n, void* stream);
/* dx = bf16(bf16(ds) * silu'(x)): silu_backward of the eager graph on the conditioning vectors (temb, the embedders' hidden layers) */
int qfx_silu_bwd(const float* ds, const uint16_t* x, uint16_t* dx, int64_t n, void* stream);

/* ---- sinusoidal timestep projection (diffusers Timesteps(dim, flip_sin_to_cos=True, shift 0, scale);
 * transformer_qwenimage.py:147,151-152,623-624): t is first rounded to bf16 (timestep.to(bf16)),
 * out[b] = bf16([cos(t*scale*f_i) | sin(t*scale*f_i)]), f_i = exp(-ln(1e4) * i / (dim/2)). */
int qfx_timestep_embed(const float* t, int32_t B, int32_t dim, float scale, float pre_scale, uint16_t* out, void* stream);
/* pre_scale != 1: the value embedded is bf16(bf16(t) * pre_scale) (FLUX: `timestep.to(dtype) * 1000`, transformer_flux.py:729-730,
 * with scale = 1); Qwen passes pre_scale = 1, scale = 1000. */

/* out = bf16(bf16(a + b) + c)  (c may be NULL): sum of the time / guidance / pooled-text embeddings
 * (diffusers CombinedTimestepGuidanceTextProjEmbeddings, transformer_flux.py:731-735) */
int qfx_add3_bf16(const uint16_t* a, const uint16_t* b, const uint16_t* c, uint16_t* out, int64_t n, void* stream);

/* ---- QK RMSNorm + RoPE on the joint [text|image] qkv buffer ---------------------------------
 * qkv [B,S,3*H*dh] (q|k|v sections), in place on the q and k sections:
 *   t = bf16(bf16(x * rsqrt(mean(x^2)+eps)) * w) ; out = bf16(complex(t) * rope[s])   pairs (2j,2j+1)
 * w = w_txt_* for rows s < T, w_img_* otherwise. rope [S, dh/2, 2] fp32 (cos,sin), joint order.
 * (transformer_qwenimage.py:305-320, apply_rotary_emb_qwen :134-140). Pre-norm q,k are first copied to
 * `saved` [B,S,2*H*dh] (needed by the backward) when saved != NULL.
 */
/* rope_bstride: elements between consecutive samples' tables (per-sample RoPE of the multi-resolution path,
 * transformer_flux_custom.py:537-560); 0 = one table shared by the batch. */
int qfx_qk_norm_rope_fwd(uint16_t* qkv, uint16_t* saved, const float* rope,
                         const uint16_t* wq_txt, const uint16_t* wk_txt, const uint16_t* wq_img, const uint16_t* wk_img,
                         int32_t B, int32_t S, int32_t T, int32_t H, int32_t dh, float eps, int32_t flags, int64_t rope_bstride,
                         void* stream);
/* flags bit0: torch.nn.RMSNorm rounding (FLUX, transformer_flux.py:342-343: one rounding after x*rstd*w) instead of the
 * diffusers RMSNorm double rounding (Qwen).
 * flags bit1 (forward only): OUT OF PLACE -- the pre-norm q,k are read from `saved` ([B,S,2*H*dh], where the q/k projections
 * wrote them) and the normalised, rotated q,k go to the q,k sections of qkv: no copy pass (one 2*B*S*H*dh*2-byte write less). */
/* in place on the q,k sections of dqkv (v section untouched): un-rotate, RMSNorm backward. */
int qfx_qk_norm_rope_bwd(uint16_t* dqkv, const uint16_t* saved, const float* rope,
                         const uint16_t* wq_txt, const uint16_t* wk_txt, const uint16_t* wq_img, const uint16_t* wk_img,
                         int32_t B, int32_t S, int32_t T, int32_t H, int32_t dh, float eps, int32_t flags, int64_t rope_bstride,
                         void* stream);

/* ---- [B,S,H,dh] (row stride ld_in, column offset applied by caller) -> [B,H,dh,S_pad] with zero pad */
int qfx_transpose_heads(const uint16_t* in, int64_t ld_in, uint16_t* out, int32_t B, int32_t S, int32_t S_pad,
                        int32_t H, int32_t dh, void* stream);

/* ---- joint attention (non-causal flash attention, optional additive key mask) ---------------
 * Replaces torch.cat + F.scaled_dot_product_attention (transformer_qwenimage.py:324-337) and its backward.
 * Q,K,V: token-major [B,S,H,dh] views with row stride ld (elements).  Kt,Vt,Qt,dOt ([B,H,dh,S_pad]) are RESERVED: the
 * kernels take their transposed MFMA operands from the row-major LDS tiles with ds_read_b64_tr_b16, so no transposed
 * copy is read any more; the fields keep the struct layout stable and may be NULL.
 * O [B,S,H*dh] (row stride ldo). lse2 [B,H,S_pad] fp32 = log2-domain logsumexp. key_mask [B,S] fp32 additive or NULL.
 */
/* ABI 6: a rank-r LoRA down projection fused into an attention epilogue (north_star: "LoRA side branches ... inside the kernels").
 * The kernel that holds a row of X = O / d(pre-norm q) / d(pre-norm k) / dV in registers also emits, per head h,
 *     part[h * part_hstride + row * ld_part + c0 + j] = sum_{n < dh} X[row, h*dh + n] * (W_hi + W_lo)[j, h*dh + n]      (j < R, fp32)
 * with row = the JOINT row b*S + s; qfx_lora_head_reduce adds the H slabs in a fixed order and writes the bf16 images
 * qfx_lora_down would have written (peft: lora_A(x) in the forward, dY * B in the backward; base_trainer.py:929-941).
 * w_pk = the adapter weight in HEAD-FRAGMENT order (qfx_lora_pack's A_hl / Bt_hl images: the MFMA operand of lane l for head h, 16-row
 * group nf, 32-column step ks is 16 contiguous bytes, hi and lo split interleaved per step -- fully coalesced 1 KiB wave loads; the
 * row-major split images cost 7 us per launch in 8-byte pieces at row stride); index [0] applies to rows s >= T (image stream), [1]
 * to rows s < T (text stream); NULL = that stream has no adapter on this linear.  Needs T % 16 == 0, R in {16, 32}, part 16-byte
 * aligned with ld_part % 4 == 0 and c0 % 4 == 0.  part == NULL: off. */
typedef struct qfx_head_lora {
  const uint16_t* w_pk[2];
  float* part; int64_t part_hstride; int32_t ld_part; int32_t c0; int32_t R; int32_t reserved;
} qfx_head_lora;

typedef struct qfx_attn_args {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; int64_t ldq; int64_t ldk; int64_t ldv;
  const uint16_t* Qt; const uint16_t* Kt; const uint16_t* Vt;   /* reserved (unused), may be NULL */
  uint16_t* O; int64_t ldo;
  float* lse2; float* dsum;                                      /* [B,H,S_pad] */
  const uint16_t* dO; int64_t lddo; const uint16_t* dOt;         /* dOt reserved (unused) */
  uint16_t* dQ; uint16_t* dK; uint16_t* dV; int64_t lddq; int64_t lddk; int64_t lddv;
  const float* key_mask;
  int32_t B; int32_t S; int32_t S_pad; int32_t H; int32_t dh; float scale;
  /* ABI 3: backward of the QK RMSNorm + RoPE (transformer_qwenimage.py:305-320) fused into the EPILOGUES of qfx_attn_bwd_dq / _dkv
   * when qk_saved != NULL: the kernels then write d(pre-norm q) / d(pre-norm k) instead of d(q) / d(k) -- the arithmetic of
   * qfx_qk_norm_rope_bwd on the bf16-rounded attention gradient, same rounding points (norm_flags as there), without the extra
   * pass over dqkv.  qk_saved: the pre-norm q | k copy [B,S,2*H*dh] qfx_qk_norm_rope_fwd kept (row stride ld_saved, q at +0, k at
   * +H*dh); rope [S, dh/2, 2] fp32 (+ b * rope_bstride); weights [dh] bf16, *_txt for rows s < T. */
  const uint16_t* qk_saved; int64_t ld_saved; const float* rope; int64_t rope_bstride;
  const uint16_t* wq_txt; const uint16_t* wk_txt; const uint16_t* wq_img; const uint16_t* wk_img;
  int32_t T; int32_t norm_flags; float norm_eps;
  /* ABI 6 (zero = off): hl[0] X = O in qfx_attn_fwd; hl[1] X = d(pre-norm q) in qfx_attn_bwd_dq; hl[2] X = d(pre-norm k), hl[3] X = dV
   * in qfx_attn_bwd_dkv (the backward slots need qk_saved != NULL). */
  qfx_head_lora hl[4];
  /* ABI 7: workspace of the ONE-PASS backward qfx_attn_bwd_fused (sizes from qfx_attn_bwd_fused_workspace; ignored by every other entry
   * point): dq_acc = fp32 dQ tiles accumulated across the key blocks of a head, dq_turn = per-(batch, head, query tile) turn counters,
   * ZERO before the first launch (every launch leaves them zero again). */
  float* dq_acc; int32_t* dq_turn;
} qfx_attn_args;

int qfx_attn_fwd(const qfx_attn_args* a, void* stream);       /* needs Q,K,V -> O,lse2 */
int qfx_attn_bwd_prep(const qfx_attn_args* a, void* stream);  /* dsum = rowsum(dO*O); optional: qfx_attn_bwd_dq computes and writes it too */
int qfx_attn_bwd_dq(const qfx_attn_args* a, void* stream);    /* needs Q,K,V,O,dO,lse2 -> dQ and dsum (= rowsum(dO*O), consumed by qfx_attn_bwd_dkv: launch dq first) */
int qfx_attn_bwd_dkv(const qfx_attn_args* a, void* stream);   /* needs Q,K,V,dO,lse2,dsum -> dK,dV */
/* ABI 7: the whole backward of the joint SDPA (transformer_qwenimage.py:329-337 under autograd) in ONE pass over the score tiles
 * (csrc/qfx_attn_bwd1.hip): needs Q,K,V,O,dO,lse2 and the workspace dq_acc / dq_turn -> dQ,dK,dV (and dsum); every optional block of
 * qfx_attn_args (key_mask, qk_saved..., hl[1..3]) means what it means for qfx_attn_bwd_dq + qfx_attn_bwd_dkv, results agree with that
 * pair to fp32 summation order (dQ is summed over 256-key blocks in a fixed order: bit-reproducible).  dh = 128 only; the launch is a
 * persistent grid of whole heads (<= 256 blocks) whose blocks wait for each other: it must not share the device with another kernel
 * that never terminates.  qfx_attn_bwd_fused_workspace: bytes the two workspaces need for this shape; returns QFX_EUNSUPPORTED (sizes 0)
 * where the one-pass form does not exist (dh != 128, S < 64) -- the caller then launches the two-pass pair. */
int qfx_attn_bwd_fused(const qfx_attn_args* a, void* stream);
int qfx_attn_bwd_fused_workspace(const qfx_attn_args* a, int64_t* acc_bytes, int64_t* turn_bytes);
/* ABI 7: kernel-selection policy of the attention entry points (process-wide, read from the environment QFX_ATTN_FWD64 /
 * QFX_ATTN_DQ64 / QFX_ATTN_FWD_WAVES once, at the first launch; this call overrides it): comma list of key=value with
 * fwd64 = 0 | 1 | 1p | auto (32-query kernels | 64-query kernel | its pipelined form | by shape), dq64 = 0 | 1 | auto, fwd_waves = 0 | 4 | 8.
 * NULL / "" = keep.  Returns QFX_EINVAL (nothing changed) on an unknown key or value. */
int qfx_attn_tune(const char* spec);

/* ---- flow-matching MSE criterion (src/qflux/losses/mse_loss.py:66-83 with weighting=1;
 * caller math of src/qflux/trainer/qwen_image_edit_trainer.py:839-847) --------------------------
 * pred [B, S_all, C] bf16 (only rows < S_t per sample enter), target [B,S_t,C] bf16.
 * loss (fp32 scalar, must be zeroed by caller) += mean_b mean_{s,c} (pred-target)^2 ;
 * dpred [B,S_all,C] bf16 = bf16(2*(pred-target)/(B*S_t*C) * gscale), zero for rows >= S_t. */
int qfx_mse_loss_fwd_bwd(const uint16_t* pred, const uint16_t* target, float* loss, uint16_t* dpred,
                         int32_t B, int32_t S_all, int32_t S_t, int32_t C, float gscale, void* stream);

/* token-weighted variant (src/qflux/losses/attention_mask_loss.py:146-226, reduction="mean"): token_w [B,S_t] fp32 =
 * attention_mask * edit weight; loss += sum_{b,s} token_w * mean_c (pred-target)^2 * inv_denom, inv_denom = 1/(num_valid+eps)
 * (the caller knows the valid-token count on the host: no device sync). dpred as above with the same weights. */
int qfx_mse_token_weighted_fwd_bwd(const uint16_t* pred, const uint16_t* target, const float* token_w, float* loss, uint16_t* dpred,
                                   int32_t B, int32_t S_all, int32_t S_t, int32_t C, float inv_denom, float gscale, void* stream);

/* ---- flow-matching input preparation (qwen_image_edit_trainer.py:811-812,841), bf16 eager rounding:
 * packed[b] = cat(bf16(bf16(bf16(1-sigma)*x0) + bf16(sigma*noise)), ctrl) ; target = bf16(noise - x0) */
int qfx_flowmatch_prepare(const uint16_t* x0, const uint16_t* noise, const uint16_t* ctrl, const uint16_t* sigma,
                          uint16_t* packed, uint16_t* target, int32_t B, int32_t S_t, int32_t S_c, int32_t C, int32_t mode,
                          void* stream);
/* mode 0 (Qwen): everything bf16 as documented above. mode 1 (FLUX, flux_kontext_trainer.py:522-525,567-568): x0 holds FP16
 * bits (the cache dtype), packed = bf16(fp32(bf16(1-t))*x0 + fp32(bf16(t*noise))), target = bf16(noise - bf16(x0)). */

/* ---- fused global-norm clip + AdamW over the flat LoRA parameter buffer ----------------------
 * (base_trainer.py:449-455 clip_grad_norm_ ; optimizer.step :531 with torch.optim.AdamW semantics) */
int qfx_sumsq(const float* g, int64_t n, float* out /* zeroed by caller */, void* stream);
/* Deterministic form (ABI 3): block partial sums into `partials` (fp32[nslots], caller workspace, e.g. 1024), folded by one block in
 * a fixed order into out[0] (overwritten).  Same inputs -> same bits: data-parallel replicas (identical gradients after the
 * all-reduce) compute the SAME clip coefficient; qfx_sumsq's fp32 atomics let them drift apart by ~1e-10 per step. */
int qfx_sumsq_det(const float* g, int64_t n, float* out, float* partials, int32_t nslots, void* stream);
int qfx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float bias_corr1, float bias_corr2,
                   const float* gnorm_sq /* may be NULL */, float max_norm, float grad_scale, void* stream);

/* ---- Prodigy, the reference's parameter-free optimizer choice (third party prodigyopt.Prodigy, requirements.txt:33; selected by
 * configs/face_seg_flux_kontext_fp16_prodigy.yaml:41-47 and the optimizer section of every tests/test_configs/test_example_*.yaml;
 * instantiated generically at base_trainer.py:884-898, stepped at :531 after clip_gradients :449-455).  One fused step over the
 * flat fp32 LoRA buffers: EMA update + the two global reductions (<g, p0 - p>, |s|_1) -> on-device update of the distance estimate
 * d -> parameter update.  The package's Python-float scalars live in `state` (device double[QFX_PRODIGY_STATE]: d, d_max,
 * d_numerator, d_denom, d_hat, k, then scratch), so nothing synchronises.  lr == 0 is the package's early return (nothing changes,
 * k does not advance).  beta3 <= 0 means sqrt(beta2).  Coupled weight decay (decouple = 0 with weight_decay != 0) is unsupported. */
#define QFX_PRODIGY_STATE 12
typedef struct qfx_prodigy_args {
  float* p;              /* [n] parameters, updated in place */
  const float* g;        /* [n] gradient (summed over ranks / micro-steps; scaled by grad_scale and the clip factor on the fly) */
  float* exp_avg;        /* [n] */
  float* exp_avg_sq;     /* [n] */
  float* s;              /* [n] */
  const float* p0;       /* [n] parameters at the first step() call */
  int64_t n;
  double* state;         /* device double[QFX_PRODIGY_STATE], initialised by qfx_prodigy_init_state */
  float lr, beta1, beta2, beta3, eps, weight_decay, d0, d_coef, growth_rate;
  int32_t use_bias_correction, safeguard_warmup, decouple;
  const float* gnorm_sq; /* may be NULL: sum of squares of g (qfx_sumsq) for the global-norm clip */
  float max_norm, grad_scale;
} qfx_prodigy_args;
int qfx_prodigy_init_state(double* state, double d0, void* stream);   /* d = d_max = d_hat = d0, everything else 0; synchronises */
int qfx_prodigy_step(const qfx_prodigy_args* a, void* stream);

/* ---- runtime: a HIP stream confined to the first `n_cus` bits of the driver's CU mask (consecutive bits walk the 8 XCDs first, so
 * 16 = two CUs per XCD).  The persistent GEMM grids occupy 240 of the 256 CUs; leaf work of the backward (the LoRA weight-gradient
 * launches, which the reference's autograd also schedules off the dX critical path) runs here without ever taking a CU a GEMM block
 * is waiting for.  Streams are plain hipStream_t handles; destroy with qfx_stream_destroy. ---- */
#define QFX_NUM_CU_TOTAL 256
int qfx_stream_create_cu_masked(int32_t n_cus, void** stream_out);
int qfx_stream_destroy(void* stream);
/* debug: out[2*b] = HW_ID register, out[2*b+1] = XCC_ID register of the CU block b ran on (blocks spin ~0.1 ms) */
int qfx_debug_where(uint32_t* out, int32_t n_blocks, void* stream);

/* ---- tuning: tile-geometry policy of the persistent GEMM (qfx_gemm_grouped / large qfx_gemm_bf16; ABI 5).  Every launch picks
 * its tile from rounds-over-256-CUs x relative tile time among the enabled geometries "256x128", "256x256" (rounds 1-3) and
 * "160x192" (round 4: M = 2048 image + 384 text rows tile into 13 + 3 M-tiles of 160 x 16 N-tiles = one whole round of 256 tiles).
 * tiles: NULL / "" = keep, "all", "legacy" (the two 256-row tiles), or a comma list of names; eff: NULL = keep, or three
 * comma-separated per-flop efficiencies relative to 256x128 in the order above.  Process-wide; the defaults come from the
 * environment (QFX_GEMM_TILES / QFX_GEMM_EFF) at the first launch.  Results do not depend on the geometry (same K order per
 * output element); only speed does.  Returns QFX_EINVAL for an unparsable argument.
 * Round 5: `tiles` also takes ONE policy token of the same-XCD split-K lever instead of a geometry list -- "splitk=0|1" (default 0),
 * "splitk_mink=<smallest base K taken, >= 2048>", "splitk_bias=<0..16 K tiles>" (environment: QFX_GEMM_SPLITK, _MINK, _BIAS).  With the
 * lever on, a launch whose problems all have N % 256 == 0 and K1 >= mink and whose 256x256 tiles make 218..256 work items in pairs
 * runs as two work items per tile (fp32 partial tiles through a per-stream workspace the library allocates on first use; never while the
 * stream is capturing a graph).  The fp32 summation order changes (results stay within the bf16 tolerance of the tests, run-to-run
 * bit-identical); measured slower than the unsplit launch on MI355X (profiles/r05_gemm_splitk.json): an A/B lever, not a default. ---- */
int qfx_gemm_tune(const char* tiles, const char* eff);

/* ---- debug: lane mapping of ds_read_b64_tr_b16 (64 lanes x 4 bf16 in, same out) ---- */
int qfx_debug_tr_read(const uint16_t* in, uint16_t* out, void* stream);

/* ---- misc ---- */
int qfx_abi_version(void);
const char* qfx_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* QFX_H */
