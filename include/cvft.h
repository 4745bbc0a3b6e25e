/*
 * cvft.h -- C ABI of libcvft.so: the MI355X (gfx950) compute kernels behind the
 * joint LLM+Flow LoRA fine-tuning hot path of CosyVoice-300M.
 *
 * The reference (leeoisaboy/cosyvoice-lora-finetune-framework) has NO native / FFI
 * layer: its hot path is plain torch ops inside Python modules.
 * Each entry point below therefore cites the reference *torch-op sequence* it replaces
 * (paths relative to /root/reference/cosyvoice_flow_finetune/).  A ctypes binding is in
 * cosyvoice_lora_finetune_framework_amd/hipops/binding.py; the reference-side stub a
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer supplied by the caller (torch owns all memory);
 *    the library never allocates, frees or retains pointers; no global mutable state
 *    except a thread-local error string.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises the host (safe under hipGraph capture).
 *  - activations are row-major, channel-last: [rows][channels]; rows = batch*time.
 *  - `dtype`: CVFT_F32 (0) = fp32 storage + exact-fp32 MFMA (parity path),
 *             CVFT_BF16 (1) = bf16 storage + bf16 MFMA with fp32 accumulation.
 *    Biases, norm affine params, statistics, losses and LoRA master grads are fp32.
 *  - return 0 on success, negative on bad argument / launch failure; message via
 *    cvft_last_error().
 */
#ifndef CVFT_H
#define CVFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CVFT_F32 = 0, CVFT_BF16 = 1 };
enum { CVFT_ACT_NONE = 0, CVFT_ACT_RELU = 1, CVFT_ACT_SILU = 2, CVFT_ACT_GELU_ERF = 3,
       CVFT_ACT_GELU_TANH = 4, CVFT_ACT_MISH = 5 };

int cvft_version(void);
const char* cvft_last_error(void);
/* Execution hint (performance only, never results): the number of independent kernel chains the caller keeps in flight on
 * separate streams (llm_flow_model.JointLLMFlowModel: LLM + Flow sub-batches).  With >= 3, kernels that would own a whole CU
 * give way to tile shapes that share it.  Returns the previous value. */
int cvft_set_concurrent_chains(int n);
int cvft_concurrent_chains(void);

/* ---------------------------------------------------------------------------------
 * Tap-GEMM with fused rank-r LoRA side path and epilogue.
 *
 *   C[orow(m), n] = epi( alpha * ( sum_{j<ntaps} A[irow_j(m), :K] . W[n, j*K:(j+1)*K]
 *                                 + U[m, :R] . Bl[n, :R] ) + bias[n] )
 *   (U either given, or -- fused form -- computed in the same launch as  lora_scale * A . La^T)
 *
 * m = b*Tm + t (t < Tm):   irow_j(m) = b*Tin + (t*in_stride + tap_off[j])   (zero row if the
 * time index is outside [0,Tin) or >= in_len[b]);   orow(m) = b*Tout + t*out_stride + out_off
 * (skipped if >= Tout; written as 0 if >= out_len[b]).
 * epi(v): if preact: preact[orow,n] = v;  v = act(v);  if dact_src: v *= act'_{dact}(dact_src[orow,n]);
 *         if residual: v += residual[orow,n].
 *
 * Replaces: nn.Linear / lora.LoRALinear.forward (lora.py:64-76: 3x F.linear + scale + add),
 * nn.Conv1d k=1/k=3 stride 1/2 and nn.ConvTranspose1d(k4,s2,p1) of the U-Net estimator
 * (matcha/models/components/decoder.py:35-158 == modules.py:60-120) with their
 * `x * mask` pre/post multiplies, activation (positionwise_feed_forward.py:55,
 * diffusers GELU) and residual adds; and each of their input-gradient passes.
 * ------------------------------------------------------------------------------- */
typedef struct {
    int dtype;
    int M, N, K;
    int Tm, Tin, Tout, in_stride, out_stride, out_off;
    int ntaps;
    int tap_off[4];
    const int32_t* in_len;    /* [M/Tm] or NULL */
    const int32_t* out_len;   /* [M/Tm] or NULL */
    const void* A;  int lda;
    const void* W;  int ldw;
    const void* U;  int ldu; int R;      /* LoRA: U = s * x A^T (rows indexed by m) or NULL */
    const void* Bl; int ldbl;            /* LoRA B [N][R] */
    const float* bias;                   /* [N] or NULL */
    float alpha;
    int act;
    void* preact; int ldp;               /* or NULL */
    const void* dact_src; int ldd; int dact;   /* or NULL */
    const void* residual; int ldr;       /* or NULL */
    void* C; int ldc;
    /* fused side path (optional; R <= 16 -- bf16 with K % 64 == 0: R <= 48 --, identity row geometry, 16-byte aligned operands): when La != NULL the
     * launch itself computes  Uf = lora_scale * A . La^T  (La [R][ldla]) from the A tiles it streams anyway, uses it
     * in place of U for the rank-R extension, and writes it to Uout [M][ldu] (may be NULL).  `U` is ignored. */
    const void* La; int ldla; float lora_scale; void* Uout;
    /* masked rank extension (dgrad of a LoRA layer with lora_dropout, lora.py:70-73): when xdrop_p > 0 the U . Bl^T term is
     * added per 16-wide rank tile t as  mask_t[m, n] / (1 - p) * (U_t . Bl_t^T)[m, n],  mask_t = the counter-based keep mask
     * of site xdrop_sites[t] over the OUTPUT elements (index m * N + n: the mask the forward applied to that layer's input).
     * bf16, LDS-DMA kernels with the register epilogue only (identity rows, K % 64 == 0, N % 4 == 0, R % 16 == 0, R <= 64);
     * any other launch is an error. */
    float xdrop_p; const int64_t* xdrop_seed; unsigned xdrop_sites[4];
    /* output dropout (x = residual + dropout(linear(.)), encoder_layer.py:95 / 104): when odrop_p > 0 the epilogue result is
     * multiplied by keep[m * N + n] / (1 - p) BEFORE the residual is added -- the counter-based mask of (xdrop_seed, odrop_site),
     * the one cvft_dropout_add(site) applies to a contiguous [M, N] tensor (its backward).  Needs N % 4 == 0, ldc == N,
     * identity row geometry and a seed; every GEMM kernel honours it. */
    float odrop_p; unsigned odrop_site;
} cvft_gemm_args;

int cvft_gemm(const cvft_gemm_args* a, void* stream);

/* fp8 path for the frozen-W GEMMs (BASELINE configs[4]) -- OCP e4m3 operands, fp32 accumulation:
 *   cvft_quant_fp8_rows : q[m][k] = e4m3(x[m][k] / s[m]),  s[m] = max_k |x[m][k]| / 448   (bf16 in; activation rows = per-token
 *                         scales, weight rows = per-output-channel scales; W is frozen under LoRA, so its copy is made once)
 *   cvft_gemm_fp8       : C = epilogue(alpha * (a_scale[m] * w_scale[n] * (A8 . W8^T) + U . Bl^T)) -- `a` as for cvft_gemm
 *                         (dtype bf16: U / Bl / bias / residual / C stay bf16 / fp32; a->A, a->W, lda, ldw are ignored),
 *                         identity row geometry, K % 128 == 0, N % 4 == 0.  Block-scaled MFMA (16x16x128) with unit block
 *                         scales: twice the bf16 matrix rate and half the L2 -> LDS bytes per FLOP of cvft_gemm. */
int cvft_quant_fp8_rows(int M, int K, const void* x, int ldx, void* q, int ldq, float* scale, void* stream);
int cvft_gemm_fp8(const cvft_gemm_args* a, const void* A8, int lda8, const float* a_scale, const void* W8, int ldw8,
                  const float* w_scale, void* stream);
/* Name of the kernel the calling thread's most recent cvft_gemm launched, e.g. "gemm_glds_kernel<bf16,128,64,4,2>"
 * (profiling label only: bench.py groups its HIP-event timings by it). */
const char* cvft_gemm_last_kernel(void);

/* LoRA adapter gradients (lora.py:71-76 backward; W frozen => no wgrad):
 *   dA[r,k] += sum_m V[m,r] * X[m,k]      (V = s * dY B,  already scaled)
 *   dB[n,r] += sum_m dY[m,n] * U[m,r]     (U = s * X A^T, already scaled)
 * Generic form:  G[p,q] += sum_m P[m,p] * Q[m,q]  with fp32 atomics into G (fp32). */
int cvft_tn_accum(int dtype, int M, int P, int Q, const void* Pm, int ldp, const void* Qm, int ldq,
                  float* G, int ldg, void* stream);

/* Same gradients with the rank side r (multiple of 4, <= 64) on a VALU kernel that reads the wide operand
 * once:  transpose_out = 0: out[r][C] += Rk^T Wd   (dA = V^T X);   transpose_out = 1: out[C][r] += Wd^T Rk
 * (dB = dY^T U).  Wd [M][C], Rk [M][r]; out fp32 (atomics). Other ranks fall back to cvft_tn_accum. */
int cvft_lora_rank_accum(int dtype, int M, int C, int r, const void* Wd, int ldw, const void* Rk, int ldr,
                         float* out, int ldo, int transpose_out, void* stream);
/* Deterministic two-stage form of the same gradients (no atomics): stage 1 writes one fp32 slab per row block
 * (part[s][r][C] or part[s][C][r], s < ceil(M / rows_per_block), rows_per_block 64, 128 or k*256; operands 16-byte
 * aligned, r % 16 == 0; the bf16 matrix-core path takes r in {16,32,48,64} and any rows_per_block % 32 == 0);
 * stage 2 is ONE launch for all adapters: tasks int64[ntasks][8] = {slab ptr, grad ptr, rows, cols, slab_pitch,
 * slab_stride, nsplit, 0}:  grad[i*cols + j] += sum_s slab[s*slab_stride + i*slab_pitch + j]  in fixed order (a task
 * may address a sub-block of a wider slab: the stacked q|k|v adapters share one slab). */
int cvft_lora_rank_partial(int dtype, int M, int C, int r, const void* Wd, int ldw, const void* Rk, int ldr,
                           float* part, int transpose_out, int rows_per_block, void* stream);
/* bf16 matrix-core form, dA and dB of one LoRA layer in ONE launch:  partA[s][r][K] = V^T X,  partB[s][N][r] = dY^T U
 * (row blocks of rpbA / rpbB rows, multiples of 32; operands 16-byte aligned, K % 8 == N % 8 == 0, r in {16,32,48,64}). */
int cvft_lora_rank_partial_pair(int M, int r, int K, const void* X, int ldx, const void* V, int ldv, float* partA, int rpbA,
                                int N, const void* dY, int ldy, const void* U, int ldu, float* partB, int rpbB, void* stream);
typedef struct {
    int C; const void* Wd; int ldw;      /* wide operand [M][C] (x or dY) */
    const void* Rk; int ldr;            /* rank operand [M][r] (V or U) */
    float* part; int transpose_out;     /* slabs [s][r][C] (0) or [s][C][r] (1) */
    int rows_per_block;                 /* multiple of 32 */
} cvft_rank_prob;
/* 1..4 such products (same M, same r) in one launch */
int cvft_lora_rank_partial_multi(int M, int r, int n, const cvft_rank_prob* probs, void* stream);
typedef struct {
    int M; int C; const void* Wd; int ldw;   /* wide operand [M][C] (x or dY) */
    const void* Rk; int ldr;                 /* rank operand [M][r] (V or U) */
    float* part; int transpose_out;          /* slabs [s][r][C] (0) or [s][C][r] (1) */
    int rows_per_block;                      /* multiple of 32 */
} cvft_rank_prob_m;
/* n such products of rank r, each with its OWN row count (host array; one problem per blockIdx.z, 64 per launch, passed
 * in the kernel arguments so the launches are hipGraph-capturable).  Used by a backward pass that defers the small
 * per-layer adapter-gradient products (autograd of lora.py:71-76) to its end. */
int cvft_lora_rank_partial_batch(int r, int n, const cvft_rank_prob_m* probs, void* stream);
int cvft_lora_grad_reduce(int ntasks, const void* tasks, int max_blocks_x, void* stream);
/* One launch per optimiser step: bf16 copy and transposed bf16 copy of every LoRA master in the flat fp32 buffer.
 * tiles: int64[ntiles][6] = {src offset (elements of flat_p), dst ptr, dst_t ptr, rows | cols << 32,
 * tile_row | tile_col << 32, dst_pitch | dst_t_pitch << 32} (32x32 tiles):
 *   dst[i*dst_pitch + j] = bf16(src[i*cols + j]),  dst_t[j*dst_t_pitch + i] = same.
 * Destinations are free-form so that the same launch also fills the stacked / block-diagonal q|k|v operands. */
int cvft_lora_shadow(int ntiles, const void* tiles, const float* flat_p, void* stream);

/* ---------------------------------------------------------------------------------
 * LayerNorm over the channel axis (+ optional ReLU, + optional post-scale).
 * Replaces nn.LayerNorm (+ReLU, + x*sqrt(d)) in encoder_layer.py:90-106,
 * subsampling.py:69-113/338-383 + embedding.py:267-270, matcha transformer.py:255-316.
 * Backward returns dX only (affine params are frozen under LoRA: lora.py:214-216).
 * ------------------------------------------------------------------------------- */
int cvft_layernorm_fwd(int dtype, int rows, int C, const void* x, const float* gamma, const float* beta,
                       float eps, int relu, float post_scale, void* y, float* mean, float* rstd, void* stream);
int cvft_layernorm_bwd(int dtype, int rows, int C, const void* x, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, int relu, float post_scale,
                       const void* dy, const void* dres, void* dx, void* stream);
/* dres (or NULL): a second gradient of x (the residual branch of a pre-norm block), added into dx by the same launch */
/* cvft_layernorm_bwd (relu 0, post_scale 1) that ALSO writes dxm = keep(seed, site) / (1 - p) * dx with the mask of
 * cvft_dropout_add over the flat [rows][C] index: when x = residual + dropout(linear(.)) (encoder_layer.py:95 / 104) the
 * linear's backward takes dxm as its incoming gradient instead of running its own mask pass over dx.  Vector path only
 * (C a multiple of 16 bytes of elements, 16-byte aligned pointers), else CVFT_EINVAL. */
int cvft_layernorm_bwd_mask(int dtype, int rows, int C, const void* x, const float* gamma, const float* beta,
                            const float* mean, const float* rstd, const void* dy, const void* dres, void* dx,
                            float p, const int64_t* seed, unsigned site, void* dxm, void* stream);
/* cvft_layernorm_bwd_mask (bf16) for a linear that carries a rank-16 LoRA adapter (lora.py:64-76): the same launch also writes
 * V = alpha * dxm Bt^T  ([rows][16]; Bt = lora_B^T in the compute dtype, [16][C] row-major), the product that adapter's
 * backward starts with (dA = V^T drop(x), dx += V A) -- the wave that writes a row of dxm still holds it in registers.
 * R must be 16, C a multiple of 128 and at most 1536; vector path only as above. */
int cvft_layernorm_bwd_mask_side(int rows, int C, const void* x, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, const void* dy, const void* dres, void* dx,
                                 float p, const int64_t* seed, unsigned site, void* dxm,
                                 const void* Bt, int R, float alpha, void* V, void* stream);

/* ---------------------------------------------------------------------------------
 * GroupNorm(G) + Mish (+ length mask, + per-(batch,channel) additive term), channel-last.
 * x [B][T][C]; statistics over (T, C/G) per (b, g) exactly as nn.GroupNorm on (B,C,T).
 *   y[b,t,c] = mish(gn(x)[b,t,c]) * (t < len[b]) + add[b,c]
 * Replaces Block1D (matcha decoder.py:35-47 == modules.py:60-73), the ResnetBlock1D
 * time-embedding add (modules.py:91) and InterpolateRegulator's Conv+GroupNorm(1)+Mish
 * stack (length_regulator.py:34-41).  apply_mish=0 gives plain GroupNorm.
 * ------------------------------------------------------------------------------- */
#define CVFT_GN_SPLIT 8   /* frame chunks of the backward statistics pass (partials in ws, summed in fixed order) */
int cvft_groupnorm_mish_fwd(int dtype, int B, int T, int C, int G, const void* x, const float* gamma,
                            const float* beta, float eps, const int32_t* len, const void* add /*[B][C] dtype or NULL*/,
                            int apply_mish, void* y, float* mean /*[B*G]*/, float* rstd /*[B*G]*/,
                            const int32_t* t_eff /*device int or NULL*/, void* stream);
int cvft_groupnorm_mish_bwd(int dtype, int B, int T, int C, int G, const void* x, const float* gamma,
                            const float* beta, const float* mean, const float* rstd, const int32_t* len,
                            int apply_mish, const void* dy, void* dx, float* ws /*[B*G*CVFT_GN_SPLIT*2] scratch*/,
                            const int32_t* t_eff /*device int or NULL*/, void* stream);
/* t_eff (both): the reference normalises over the PADDED batch's frames -- every t < T_max, utterance or padding
 * (modules.py:60-73 on a [B, C, T_max] tensor).  When the trainer pads T_max up to a shape bucket so that one captured step
 * serves many batches, *t_eff (<= T) carries the exact T_max: statistics and their backward run over t < *t_eff only, frames
 * t >= *t_eff are written as zeros (forward output and dx).  NULL = T. */

/* ---------------------------------------------------------------------------------
 * Fused attention, head_dim 64, additive key-padding bias -1e10 (NOT -inf):
 *   O = softmax(Q K^T * scale + bias(key >= klen[b])) V
 * Q,K,V,O: [B*T][ld] with head h at columns [h*64, h*64+64).  lse [B][H][T] fp32.
 * Replaces diffusers Attention / modules.Attention.forward (modules.py:253-293) and the
 * (B,T,T) mask_to_bias tensor (decoder.py:238-240, utils.py:103-109).
 * ------------------------------------------------------------------------------- */
int cvft_attn_bias_fwd(int dtype, int B, int H, int T, const void* q, const void* k, const void* v, int ld,
                       const int32_t* klen, float scale, int iso_len, void* o, int ldo, float* lse,
                       void* o_lo /* bf16 only, [B*T][ldo] or NULL: bf16(O - bf16(O)), the part of the fp32 output the bf16
                                     store drops; hand it to the backward entry point (its delta = rowsum(dO (O + O_lo))) */,
                       void* stream);
/* iso_len > 0: prompt-isolation mask (modules.py:844-879): frames [0, iso_len) and [iso_len, T) attend only within
 * their own segment (block-diagonal -inf bias on top of the key-padding bias). */
int cvft_attn_bias_bwd(int dtype, int B, int H, int T, const void* q, const void* k, const void* v, int ld,
                       const int32_t* klen, float scale, int iso_len, const void* o, const void* d_o, int ldo,
                       const float* lse, const void* o_lo /* the forward's, or NULL */, float* delta /*[B][H][T] ws*/,
                       void* dq, void* dk, void* dv, int ldg, void* stream);
/* (o == NULL, bf16: delta is an INPUT -- rowsum over each head's columns of dO . (O + O_lo), formed by the producer of dO
 * (cvft_block_tail_bwd / cvft_block_link_bwd, `delta`) -- and neither backward role reads the forward's output.) */

/* ---------------------------------------------------------------------------------
 * Fused relative-position attention (Transformer-XL / ESPnet), head_dim 64:
 *   S[i,j] = ((q_i+u_h).k_j + (q_i+v_h).p[L-1-i+j]) * scale ; masked (j>=len[b] or, if causal,
 *   j>i) -> -inf ; O = softmax(S) V  (post-softmax zero fill is implied).
 * p [2L-1][ldp] = linear_pos(pos_emb) (batch-independent); u,v [H][64] fp32.
 * Replaces RelPositionMultiHeadedAttention.forward + rel_shift + forward_attention
 * (cosyvoice/transformer/attention.py:200-330, 82-127) and the (B,L,L) chunk mask
 * (cosyvoice/utils/mask.py:161-236).
 * ------------------------------------------------------------------------------- */
int cvft_attn_relpos_fwd(int dtype, int B, int H, int L, const void* q, const void* k, const void* v, int ld,
                         const void* p, int ldp, const float* bias_u, const float* bias_v,
                         const int32_t* len, int causal, float scale, void* o, int ldo, float* lse,
                         void* o_lo /* as cvft_attn_bias_fwd */,
                         float drop_p, const int64_t* drop_seed, unsigned drop_site, void* stream);
/* drop_p > 0: attention-probability dropout (attention.py:118): the PV operand is masked / scaled by 1/(1-p), the softmax
 * denominator is not; the mask is a function of (*drop_seed (device int64), drop_site, b, h, i, j) and is re-derived by
 * the backward entry point given the same three values. */
int cvft_attn_relpos_bwd(int dtype, int B, int H, int L, const void* q, const void* k, const void* v, int ld,
                         const void* p, int ldp, const float* bias_u, const float* bias_v,
                         const int32_t* len, int causal, float scale, const void* o, const void* d_o, int ldo,
                         const float* lse, const void* o_lo, float* delta, void* dq, void* dk, void* dv, int ldg,
                         float* dp /*[2L-1][H*64] fp32 accum or NULL*/,
                         float drop_p, const int64_t* drop_seed, unsigned drop_site, void* stream);

/* ---------------------------------------------------------------------------------
 * Small fused ops.
 * ------------------------------------------------------------------------------- */
/* out[b,l,:] = table[max(tok[b,l],0), :] * (l < len[b])      (flow.py:92-93, llm_flow_model.py:206-208) */
int cvft_embed_gather(int dtype, int B, int L, int D, const int64_t* tok, const int32_t* len /*or NULL*/,
                      const void* table, void* out, void* stream);
/* ragged gather of rows: out[i,:] = (idx[i] >= 0) ? src[idx[i],:] : fill   (llm.py:88-95 pad_unpad_sequence) */
int cvft_gather_rows(int dtype, int n, int D, const int32_t* idx, const void* src, float fill, void* out, void* stream);
/* dsrc[idx[i],:] += dout[i,:] for idx>=0 (each source row referenced at most once) */
int cvft_scatter_rows(int dtype, int n, int D, const int32_t* idx, const void* dout, void* dsrc, void* stream);
/* y = x / max(||x||_2, 1e-12) per row (F.normalize, llm_flow_model.py:146,202) */
int cvft_l2norm_rows(int dtype, int rows, int D, const float* x, void* y, void* stream);
/* SinusoidalPosEmb(dim)(t, scale) -> [B][dim] = [sin(scale*t*f_k) | cos(scale*t*f_k)]; freqs [dim/2] fp32 is the
 * constant table exp(-k*ln(1e4)/(dim/2-1)) built once on the host  (matcha decoder.py:14-32 == modules.py:20-42) */
int cvft_time_embed(int dtype, int B, int dim, const float* t, const float* freqs, float scale, void* out, void* stream);
/* y = act(x) elementwise */
int cvft_act_fwd(int dtype, int64_t n, int act, const void* x, void* y, void* stream);
/* dz = dy * act'(z) elementwise */
int cvft_act_bwd(int dtype, int64_t n, int act, const void* z, const void* dy, void* dz, void* stream);
/* Inverted dropout (+ optional residual add), nn.Dropout call sites of the encoders (subsampling.py:84,
 * embedding.py:285-288, encoder_layer.py:95-104 / 205-234, positionwise_feed_forward.py:54):
 *   y[i] = residual[i] + (keep(i) ? x[i] / (1 - p) : 0),  keep = f(*seed (device int64), site, i) -- counter-based, so the
 * backward pass calls the same entry point on dy with the same (seed, site) instead of storing a mask. */
int cvft_dropout_add(int dtype, int64_t n, const void* x, const void* residual, void* y, float p,
                     const int64_t* seed, unsigned site, void* stream);
/* h = dropout(act(z)) (positionwise_feed_forward.py:54: w_2(dropout(activation(w_1 x)))) in one pass, and (dh != NULL) its
 * backward  y = keep/(1-p) * dh * act'(z)  in one pass; same mask generator and (seed, site) convention. */
int cvft_act_dropout(int dtype, int64_t n, int act, const void* z, const void* dh, void* y, float p,
                     const int64_t* seed, unsigned site, void* stream);
/* LoRA side path under lora_dropout (lora.py:70-73), bf16, masks shared with cvft_dropout_add (element index m*K + k):
 *   cvft_skinny_dropout : U[M,R] = alpha/(1-p) * sum_k keep_t(m,k) X[m,k] A[16t+j][k]   (t = rank tile, sites[t], R/16 entries;
 *                         R in {16, 32, 48, 64}: all sites equal = ONE adapter of rank R; distinct sites only for R = 48, the
 *                         stacked q|k|v adapters); xd (NULL or 3 pointers, one per DISTINCT site, entries may be NULL): also
 *                         writes drop_t(X) = keep_t X / (1-p), [M][K] each, for the backward's dA_t = V_t^T drop_t(X)
 *   cvft_lora_side_dgrad: out[m,k] = dx[m,k] + sum_t keep_t(m,k)/(1-p) * sum_j V[m,16t+j] A[16t+j][k] */
/* Dropout masks: keep(i) is a pure function of (*seed, site, element index i) -- one 64-bit draw (two 32-bit murmur3 finalisers of the group index under the halves of the SplitMix64 site key) per group of 4 consecutive
 * elements, a 16-bit field each, kept when field >= rint(p * 65536); kept values are scaled by the nominal 1 / (1 - p).  The
 * rate is therefore quantised to 1/65536 (0.05 -> 0.050003); every entry point that takes a rate accepts p == 0 (off, where the
 * entry allows it) or 2^-16 <= p <= 1 - 2^-16 and rejects anything else (a smaller p would drop nothing yet still scale, a
 * larger one would keep 1 element in 65536 at a huge scale). */
int cvft_skinny_dropout(int M, int K, int R, const void* X, int ldx, const void* A, int lda, float alpha, void* C, int ldc,
                        float p, const int64_t* seed, const unsigned* sites, void* const* xd, void* stream);
/* LayerNorm + that product in ONE launch, for a pre-norm block whose normalised input feeds a LoRA adapter with lora_dropout
 * (encoder_layer.py:90-104 norm -> linear_q|k|v / w_1; matcha transformer.py:255-316 norm1 -> to_q|k|v):
 *   Y = LN(X) = (X - mean) * rstd * gamma + beta   (bf16, two-pass fp32 statistics; what the main GEMM then reads)
 *   mean, rstd [M] fp32                              (saved for cvft_layernorm_bwd)
 *   U = alpha / (1 - p) * drop_t(Y) A_t^T, xd[t] = drop_t(Y)   exactly as cvft_skinny_dropout(Y, ...)
 * X, Y contiguous [M][K] bf16; K % 32 == 0, K <= 1024; R = 16 (one site) or 48 (stacked q|k|v, three sites). */
int cvft_ln_skinny_dropout(int M, int K, int R, const void* X, const float* gamma, const float* beta, float eps, void* Y,
                           float* mean, float* rstd, const void* A, int lda, float alpha, void* U, int ldu, float p,
                           const int64_t* seed, const unsigned* sites, void* const* xd /*[3] or NULL*/, void* stream);
int cvft_lora_side_dgrad(int M, int K, int R, const void* V, int ldv, const void* A, int lda, const void* dx, int ldi,
                         void* out, int ldo, float p, const int64_t* seed, const unsigned* sites, void* stream);
/* CFM prepare (flow_matching.py:173-186 == flow_model.py:143-161), channel-last:
 *   feat raw log-mel [B][T][80] fp32, z [B][T][80] fp32, t_raw [B], cfg_keep [B] (0/1), mu [B][T][80],
 *   spk [B][80], cond (or NULL = zeros)  ->  xin [B][T][320] = [y | mu*keep | spk*keep | cond*keep],  u [B][T][80] (fp32), t [B].
 *   t_cosine: t_scheduler == 'cosine' (t = 1 - cos(t_raw pi/2), the CosyVoice-300M setting); 0 = t_raw as drawn. */
int cvft_cfm_prepare(int dtype, int B, int T, const float* feat, const float* z, const float* t_raw,
                     const float* cfg_keep, const void* mu, const void* spk, const void* cond /*[B][T][80] or NULL*/,
                     float mel_mean, float mel_std, float sigma_min, int t_cosine, void* xin, float* u, float* t, void* stream);
/* masked MSE (flow_matching.py:192): loss_sum[0] += sum(((pred-u)*m)^2) (caller divides by sum(mask)*80);
 * backward: dpred = gscale[0] * 2 * (pred-u) * m  with gscale a DEVICE scalar (= upstream grad / denominator). */
/* w (or NULL): per-frame loss weights [B*T] fp32 -- loss_sum += sum(((pred-u) * w)^2) over frames t < len[b]
 * (prompt region 0, boundary frames > 1: flow_model.py:179-202; note the weight enters squared, as in the reference) */
int cvft_masked_mse_fwd(int dtype, int B, int T, int C, const void* pred, const float* u, const int32_t* len,
                        const float* w, float* loss_sum, void* stream);
int cvft_masked_mse_bwd(int dtype, int B, int T, int C, const void* pred, const float* u, const int32_t* len,
                        const float* w, const float* gscale, void* dpred, void* stream);
/* linear interpolation along time (F.interpolate mode='linear', align_corners=False), channel-last.
 * x [B][Lin][C] -> y [B][Lout][C]   (length_regulator.py:47)
 * eff (device int32[2] or NULL): the exact-shape batch's (Lin, Lout) when x / y are padded to shape buckets: the scale
 * Lin/Lout and the source clamp come from eff, output frames >= eff[1] are zeros, input frames >= eff[0] get no gradient. */
int cvft_interp_linear_fwd(int dtype, int B, int Lin, int Lout, int C, const void* x, void* y, const int32_t* eff, void* stream);
int cvft_interp_linear_bwd(int dtype, int B, int Lin, int Lout, int C, const void* dy, void* dx, const int32_t* eff, void* stream);
/* token-mean label-smoothed cross entropy with ignore index + argmax accuracy
 * (LabelSmoothingLoss.forward, label_smoothing_loss.py:68-96 ; th_accuracy, common.py:78-97).
 * logits [n][V]; target [n] int32 (-1 ignore); smoothing in [0, 1] (0: plain NLL, the CosyVoice-300M setting):
 * true_dist = 1 - smoothing at the target, smoothing / (V - 1) elsewhere.
 * out[0] += sum over valid rows of sum_c true_dist_c (log true_dist_c - log_softmax_c), out[1] += #valid, out[2] += #correct.
 * dlogits = gscale_ptr[0] * (softmax - true_dist) for valid rows, 0 otherwise. */
int cvft_ce_fwd(int dtype, int n, int V, const void* logits, int ld, const int32_t* target, float* out3,
                float* row_lse /*[n]*/, float smoothing, void* stream);
int cvft_ce_bwd(int dtype, int n, int V, const void* logits, int ld, const int32_t* target, const float* row_lse,
                const float* gscale /*device [1]*/, void* dlogits, int ldd, float smoothing, void* stream);
/* depthwise Conv1d (groups=C), channel-last, zero padding, optional length mask on the input
 * (cosyvoice/transformer/convolution.py:62-70,118; not executed by the 300M config). */
int cvft_dwconv1d_fwd(int dtype, int B, int T, int C, int Kw, int pad_left, const void* x, const float* w /*[C][Kw]*/,
                      const float* bias, const int32_t* len, void* y, void* stream);
int cvft_dwconv1d_bwd(int dtype, int B, int T, int C, int Kw, int pad_left, const void* dy, const float* w,
                      const int32_t* len, void* dx, void* stream);

/* flat-buffer optimiser pieces (train_joint.py:198-226, 349-360): */
/* out[0] += sum(g^2) */
int cvft_sumsq(int64_t n, const float* g, float* out, void* stream);
/* out[0] = sum(g^2), summed in a fixed order (bitwise reproducible; what torch.nn.utils.clip_grad_norm_ feeds the
 * clip at train_joint.py:349-360 needs to be identical on every data-parallel replica); partials: CVFT_SUMSQ_PARTS floats */
#define CVFT_SUMSQ_PARTS 1024
int cvft_sumsq_ordered(int64_t n, const float* g, float* partials, float* out, void* stream);
/* AdamW on a flat fp32 buffer; clip coefficient = min(1, max_norm/(sqrt(gnorm_sq[0])*inv_scale + 1e-6)),
 * lr read from device lr[0]; step count from device step[0] (float, already incremented). */
int cvft_adamw_flat(int64_t n, float* p, const float* g, float* m, float* v, const float* lr, float beta1, float beta2,
                    float eps, float wd, const float* step, const float* gnorm_sq, float max_norm, float grad_scale,
                    void* stream);
/* fp32 -> bf16 cast of a flat buffer */
int cvft_cast_f32_to_bf16(int64_t n, const float* src, void* dst, void* stream);

/* ---------------------------------------------------------------------------------
 * Estimator transformer block as row-tile chain kernels (bf16, residual width d = 256).
 * Replaces, per BasicTransformerBlock (matcha/models/components/transformer.py:243-316 == modules.py:296-375; diffusers
 * Attention.to_out / FeedForward / GELU), the torch-op sequence
 *     x1  = x0 + to_out(o)                       (attention output projection + residual)
 *     out = x1 + ff.net[2](gelu(ff.net[0].proj(norm3(x1))))
 * and its backward.  The [M, F] hidden activations and LN(x1) stay on chip; the pre-activations go to an opaque workspace z.
 *
 * Packed weights (made once per frozen weight by hipops/blockpack.py; element type bf16).  A fragment = what one
 * v_mfma_f32_32x32x16_bf16 takes as its A operand = 64 lanes x 8 elements = 1 KB, lane l = 32 h + r, element j:
 *   natural(Wm, rt, ks)[l][j]    = Wm[32 rt + r][16 ks + 8 h + j]
 *   chained(Wm, rt, kt, s)[l][j] = Wm[32 rt + r][32 kt + 16 s + 8 (j>>2) + 4 h + (j&3)]
 *               (k order of a bf16-packed 32x32 accumulator tile used as the next product's B operand)
 * W_fwd / W_bwd are the four waves' weight STREAMS, [4 waves][wave_frags] fragments, each wave's fragments in the order the
 * wave consumes them (n = F / 128 hidden tiles per wave, ht = n w + t; q = DI / 64 k-steps per wave in the projection):
 *   W_fwd, wave w:  natural(Wo [256][DI], ct, q w + ks)            for ks < q, ct < 8                (absent when DI == 0)
 *                   natural(W1 [F][256], ht(0), ks)                for ks < 16
 *                   for t < n: [ natural(W1, ht(t+1), ks) for ks < 16   (t + 1 < n) ],  chained(W2 [256][F], ct, ht(t), s) for s < 2, ct < 8
 *                   wave_frags = DI / 8 + F / 4
 *   W_bwd, wave w:  natural(W2^T [F][256], ht(0), ks)              for ks < 16
 *                   for t < n: [ natural(W2^T, ht(t+1), ks) for ks < 16 (t + 1 < n) ],  chained(W1^T [256][F], dt, ht(t), s) for s < 2, dt < 8
 *                   natural(Wo^T [DI][256], (DI/128) w + 2 r + f2, ks)  for r < DI/256, ks < 16, f2 < 2   (absent when DI == 0)
 *                   wave_frags = F / 4 + DI / 8
 * z: [ceil(M/32)][F/32][64 lanes][16] bf16 pre-activations (accumulator order of the producing wave; only
 *    cvft_block_tail_bwd reads it).
 * M rows, any M > 0; F % 128 == 0, F <= 2048; DI in {0, 256, 512} (the pack's: o != NULL iff DI > 0);
 * act = CVFT_ACT_GELU_ERF | CVFT_ACT_GELU_TANH; all pointers 16-byte aligned.
 * ------------------------------------------------------------------------------- */
typedef struct {
    int M;
    const void* o; int ldo; int DI;      /* attention output [M][DI] (row pitch ldo); NULL iff DI == 0: then x1 is an INPUT */
    const void* x0;                      /* [M][256] residual in front of the attention (DI > 0) */
    const void* W_fwd;                   /* weight streams (above) */
    const float* bo;                     /* to_out bias [256] (DI > 0) */
    void* x1;                            /* [M][256]: written when DI > 0 (saved for backward), read otherwise */
    const float* gamma; const float* beta; float eps;      /* norm3 */
    const float* b1; int F;
    const float* b2;
    int act;
    void* z;                             /* backward workspace or NULL (inference): gelu'(pre-activation), bf16, in the producing wave's
                                            accumulator order [row tile][F / 32][64 lanes][16] */
    float* mean; float* rstd;            /* [M] LayerNorm statistics of x1 (saved for backward) */
    void* out;                           /* [M][256] */
    int lean;                            /* 0: W_fwd as above, the workgroup owns its CU (512 registers per lane, 155 KB LDS);
                                            1: the CU-sharing form (block_lean.hip: <= 256 registers, 82 KB) -- W_fwd then holds the
                                            "lean" streams: wave w = fragments of ITS 64 output features: natural(Wo, 2 w + c2, ks) in
                                            order [ks][c2]; natural(W1, w, ks); per round r < F / 128: [natural(W1, 4 (r+1) + w, ks)],
                                            natural(W2, 2 w + c2, 8 r + k') in order [k'][c2]  (same wave_frags);
                                            2: 64 rows per workgroup (block_wide.hip; F >= 256): the lean groups with the second product
                                            lagging one round -- G1(0), G1(1), { G1(r + 1), G2(r - 1) : r = 1 .. nr - 2 }, G2(nr - 2),
                                            G2(nr - 1) with G1(r) = natural(W1, 4 r + w, ks), G2(r) = natural(W2, 2 w + c2, 8 r + k')
                                            [k'][c2].  In this form z is mandatory, holds whole 64-row groups (ceil(M / 64) * 64 * F
                                            elements);
                                            4: the eight-wave 64-row forward (block_wide8.hip; F % 256 == 0, 512 <= F <= 1024; W_fwd = wave w of 8:
                                            natural(Wo, w, ks), then G1(r) = natural(W1, 8 r + w, ks), G2(r) = natural(W2, w, 16 r + k') in the same
                                            lagged order).  Every form stores the same z, so the directions may use different forms */
} cvft_block_tail_args;
int cvft_block_tail_fwd(const cvft_block_tail_args* a, void* stream);
typedef struct {
    int M;
    const void* x1; const void* dy;      /* [M][256] saved x1; gradient at the block output */
    const float* gamma; const float* mean; const float* rstd;
    const void* z;
    const void* W_bwd; int F; int DI; int act;
    void* dx1;                           /* [M][256] gradient at x1 (= gradient of x0 through the residual) */
    void* dout; int lddo;                /* dout [M][DI] = dx1 . to_out.weight, or NULL (not wanted / DI == 0) */
    /* optional (forms 0 and 2, with dout): delta[(b * DI/64 + h) * T + t] = sum over head h's 64 columns of dout[b T + t] . (attn_o +
       attn_o_lo)[b T + t] -- what the attention backward that consumes dout needs of its own output (cvft_attn_bias_bwd with o ==
       NULL); attn_o [M][ldao] is the attention output the forward's `o` was, attn_o_lo its rounding residual or NULL; M == B * T */
    const void* attn_o; const void* attn_o_lo; int ldao; float* delta; int T;
    int lean;                            /* as in cvft_block_tail_args; lean W_bwd, wave w: natural(W2^T, w, ks); per round r:
                                            [natural(W2^T, 4 (r+1) + w, ks)], natural(W1^T, 2 w + c2, 8 r + k') in order [k'][c2];
                                            natural(Wo^T, (DI/128) w + f, ks) in order [ks][f];  2: the same groups in the lagged
                                            order of the forward's form 2, then the Wo^T groups; lddo % 8 == 0 */
} cvft_block_tail_bwd_args;
int cvft_block_tail_bwd(const cvft_block_tail_bwd_args* a, void* stream);

/* First half of the same block (matcha transformer.py:255-289 == modules.py:349-361: norm1, the three LoRALinear projections
 * to_q / to_k / to_v of lora.py:64-76 with their lora_dropout, stacked), forward and backward, as row-tile chain kernels:
 *   fwd:  y = LN(x);   U[m, 16 t + j] = alpha / (1 - p) * sum_k keep_t[m, k] y[m, k] A[16 t + j, k]   (t = q, k, v);
 *         Y[m, n] = sum_k y[m, k] Wqkv[n, k] + bias[n] + sum_r U[m, r] Bb[n, r];   xd[t] = keep_t y / (1 - p)   (p > 0)
 *   bwd:  V = alpha * dY Bb;   dy[m, c] = sum_n dY[m, n] Wqkv[n, c] + sum_t keep_t[m, c] / (1 - p) * sum_j V[m, 16 t + j] A[16 t + j, c];
 *         dx = dres + LN'(dy)
 * keep_t = the counter-based mask of (seed, sites[t]) over the [M, 256] elements (index m * 256 + k), the one cvft_skinny_dropout
 * and cvft_gemm's masked rank extension draw -- so this form and the launch-per-stage form agree mask for mask; p == 0: no masks.
 * Wqkv [3N][256] = rows of to_q | to_k | to_v (3N = 1536); A [48][lda] = the adapters' lora_A stacked, At its transpose [256][48];
 * Bb [3N][48] block-diagonal (rows of adapter t hold its lora_B in columns 16 t .. 16 t + 15, zeros elsewhere), Bbt its transpose
 * -- the optimiser's bf16 shadows (optim.FlatAdamW.stack_for), read in place.
 * W_fwd: per wave w the fragments natural(Wqkv, 12 w + i, ks) for i < 12, ks < 16;  W_bwd: natural(Wqkv^T [256][3N], ct, 24 w + k)
 * for k < 24, ct < 8 (order [k][ct]);  each [4 waves][192 fragments] + 32 fragments of padding (hipops/blockpack.py).
 * Adapter gradients stay with cvft_lora_rank_partial*: dA_t = V_t^T xd[t] (or y_out when p == 0), dB = dY^T U. */
typedef struct {
    int M;
    const void* x;                       /* [M][256] block input */
    const float* gamma; const float* beta; float eps;      /* norm1 */
    float* mean; float* rstd;            /* [M] out (saved for backward) */
    const void* W_fwd; const float* bias; int N3;          /* bias [3N] or NULL */
    const void* A; int lda; const void* Bb; int ldb;
    float alpha; float p; const int64_t* seed; unsigned sites[3];
    void* U; int ldu;                    /* [M][48] out */
    void* xd[3];                         /* p > 0: dropped copies [M][256] (entries may be NULL) */
    void* y_out;                         /* p == 0: LN(x) [M][256] or NULL */
    void* Y; int ldy;                    /* [M][3N] out */
    int wide;                            /* 0: 32 rows per workgroup (block_qkv.hip); 1: 64 rows per workgroup (block_qkv_wide.hip), same
                                            W_fwd stream, ldy % 8 == 0 */
} cvft_block_qkv_args;
int cvft_block_qkv_fwd(const cvft_block_qkv_args* a, void* stream);
typedef struct {
    int M;
    const void* dY; int lddy;            /* [M][3N] gradient of q | k | v */
    const void* dres;                    /* [M][256] gradient of the residual branch, or NULL */
    const void* x; const float* gamma; const float* mean; const float* rstd;
    const void* W_bwd; int N3;
    const void* At; int ldat; const void* Bbt; int ldbt;
    float alpha; float p; const int64_t* seed; unsigned sites[3];
    void* V; int ldv;                    /* [M][48] out */
    void* dx;                            /* [M][256] out */
    int wide;                            /* 0: W_bwd = wave w's quarter of 3N, [ks][ct] (block_qkv.hip); 1: 64 rows per workgroup
                                            (block_qkv_wide.hip, 8 waves), W_bwd = wave w's feature tile w over all of 3N, [ks] */
} cvft_block_qkv_bwd_args;
int cvft_block_qkv_bwd(const cvft_block_qkv_bwd_args* a, void* stream);
/* The tail of block i and the head of block i + 1 (the two halves either side of a block boundary of modules.py:349-375 inside one
 * stage of the estimator) on the same 32 rows in ONE launch: exactly cvft_block_tail_fwd(tail) followed by cvft_block_qkv_fwd(head)
 * with head->x == tail->out -- same outputs, bit for bit, same masks -- but the block output stays in registers between the two and
 * the weight ring never drains.  32-row forms only (tail->lean == 0, head->wide == 0), with the output projection (DI == 512),
 * 256 <= F <= 1024.  W_link replaces tail->W_fwd and head->W_fwd: per wave w its DI/8 + F/4 tail fragments (as in W_fwd) followed
 * by its 192 head fragments, [4 waves][DI/8 + F/4 + 192] + 32 fragments of padding (hipops/blockpack.py, BlockLinkPack). */
int cvft_block_link_fwd(const cvft_block_tail_args* tail, const cvft_block_qkv_args* head, const void* W_link, void* stream);
/* The same boundary backwards: exactly cvft_block_qkv_bwd(head) followed by cvft_block_tail_bwd(tail) with tail->dy == head->dx
 * (the gradient at the block boundary, still written) -- same outputs, bit for bit.  32-row forms only (head->wide == 0,
 * tail->lean == 0), DI == 512 with dout, 256 <= F <= 1024.  W_link: per wave w its 192 head fragments (as in the head's W_bwd)
 * followed by its F/4 + DI/8 tail fragments (as in the tail's W_bwd), + 32 fragments of padding. */
int cvft_block_link_bwd(const cvft_block_qkv_bwd_args* head, const cvft_block_tail_bwd_args* tail, const void* W_link, void* stream);

/* Diagnostics (development only): cycle stamps of the default 128x128 LDS-DMA GEMM kernel (CVFT_GLDS_BIG=15 launches its
 * stamped build; tools/glds_stamps.py); host_out receives 256 uint64. */
int cvft_debug_glds_stamps(unsigned long long* host_out);
/* Diagnostics: buf[slot] = the device's wall clock (s_memrealtime, 100 MHz ticks) when `stream` reaches this launch; usable
 * inside a captured hipGraph (llm_flow_model.py, CVFT_CHAIN_EVENTS: when does each chain of the step end). */
int cvft_debug_stamp(unsigned long long* buf, int slot, void* stream);
/* fp8 groundwork (BASELINE configs[4]): one v_mfma_scale_f32_16x16x128_f8f6f4 product, C[16][16] = A[16][128] . B[16][128]^T on
 * OCP e4m3 bytes with unit block scales -- pins the operand layout the fp8 GEMM will use (tests/test_ops_gpu.py). */
int cvft_debug_mfma_fp8_probe(const void* A, const void* B, float* C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVFT_H */
