/* avformer_hip.h - C ABI of libavformer_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the transformer hot path of
 * ColinWine/Multi-modal-Multi-label-Facial-Action-Unit-Detection-with-Transformer.
 * The reference has no native layer of its own (pure PyTorch eager), so there is no FFI to mirror;
 * each entry point below cites the reference Python it replaces (paths relative to the reference
 * repository root).  Conventions:
 *
 *   - every function returns 0 on success, non-zero on error; avf_last_error() returns a
 *     thread-local message for the last failure.  Nothing throws across the boundary.
 *   - all data pointers are CALLER-OWNED DEVICE pointers (PyTorch allocates inputs, outputs,
 *     saved activations and workspaces); the library allocates nothing persistent.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     All work is enqueued on it; no entry point synchronises the device (graph-capture safe).
 *   - activations are row-major [rows = batch*tokens, features]; the residual stream is fp32;
 *     `dtype` selects the compute/storage type of the non-residual activations:
 *     AVF_F32 (parity mode: fp32 MFMA / fp32 VALU) or AVF_BF16 (throughput mode: bf16 MFMA,
 *     fp32 accumulate, fp32 LayerNorm/softmax statistics).
 */
#ifndef AVFORMER_HIP_H
#define AVFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVF_F32 0
#define AVF_BF16 1

/* GEMM epilogues (avf_gemm) */
#define AVF_EPI_NONE 0      /* C = acc (+bias if given)                                           */
#define AVF_EPI_BIAS_RES 1  /* C(f32) = acc + bias + residual(f32)        heads.py:175,238,196    */
#define AVF_EPI_BIAS_GELU 2 /* aux = acc + bias ; C = tanh-GELU(aux)      heads.py:166,191-192    */
#define AVF_EPI_DGELU 3     /* C = acc * dGELU/du(aux)                    backward of heads.py:166 */

typedef struct avf_layer_cfg {
  int32_t batch;       /* clips B                                                              */
  int32_t tokens;      /* tokens per clip N                                                    */
  int32_t dim;         /* D      - Transformer(dim, ...)                  heads.py:243          */
  int32_t heads;       /* H                                               heads.py:204          */
  int32_t dim_head;    /* dh ; inner I = H*dh                             heads.py:206          */
  int32_t mlp_dim;     /* M                                               heads.py:189          */
  int32_t dtype;       /* AVF_F32 | AVF_BF16                                                    */
  int32_t project_out; /* 0 iff heads==1 && dim_head==dim (nn.Identity)   heads.py:207: pass the identity matrix as w_out and
                          zeros as b_out (the GEMM then returns the attention output exactly); no dropout site 0 */
  float ln_eps;        /* nn.LayerNorm default 1e-5                       heads.py:181          */
  float dropout_p;     /* nn.Dropout p of the three sites (after to_out, after GELU, after net.3)
                          heads.py:194-196,216; 0 = eval()/no dropout.  p > 0 needs dim % 4 == 0, dim <= 1536.  Masks are a pure
                          function of (seed, layer_index, site, element), regenerated in backward.               */
  uint32_t seed_lo, seed_hi; /* dropout seed: use a fresh value per forward, the same one in its backward        */
  int32_t layer_index; /* position of this layer in its stack (keys the dropout masks)                          */
  const void* seed_dev; /* optional device pointer to a uint64 seed; when non-null it replaces seed_lo/hi and is read
                          by the kernels at run time, so a captured hipGraph draws fresh masks on every replay
                          (the caller advances the value between forwards, e.g. with a captured add)            */
  int32_t grad_stream_bf16; /* 1 (AVF_BF16, dropout_p == 0, dim <= 1536): avf_layer_bwd keeps the residual gradient stream in
                          bf16 - it reads the incoming gradient from dx_out_lo (dx_out may be null; if only dx_out is given it
                          is cast first), the two LayerNorm backward kernels exchange bf16 only, and the fp32 dx_in is written
                          only when the pointer is non-null (a caller passes it where it consumes fp32, e.g. for the bottom
                          layer).  Per-layer parameter gradients stay fp32.                                            */
  int32_t mx8_fwd;     /* 1 (AVF_BF16 only; dim, mlp_dim % 128 == 0): the forward GEMMs of to_qkv, net.0 and net.3 take MX-FP8
                          operands (BASELINE config 5).  The bf16 weight-image buffer then also holds their e4m3 images:
                          refresh them with avf_stack_quant_weights_mx8 whenever the bf16 images changed.          */
  int32_t resid_bf16;  /* 1 (AVF_BF16 only; dim % 8 == 0, dim <= 1536): the FORWARD residual stream is stored in bf16 - x_in, x_out of
                          avf_layer_fwd / avf_layer_bwd and the saved mid-layer stream are bf16 tensors (LayerNorm statistics,
                          GEMM accumulation and the residual add itself stay fp32; one bf16 rounding per residual add).
                          Cuts the HBM bytes of the two LayerNorms and the two residual GEMM epilogues of a layer by a third. */
  int32_t mx8_bwd;     /* 1 (needs mx8_fwd): the backward GEMMs dX = dY W of net.3, net.0 and to_out take MX-FP8 operands too: the
                          LayerNorm backward kernels and the dGELU epilogue write the e4m3 image of their output beside the bf16
                          one, avf_stack_quant_weights_mx8 also makes the images of the transposed weights.  The bf16
                          gradient-stream buffers dx_out_lo / dx_in_lo of avf_layer_bwd are then avf_layer_grad_stream_bytes()
                          long: [R, D] bf16 | [R, D] e4m3 | [R, D/32] E8M0 (each part 256-byte aligned).               */
  int32_t dx_out_mx8;  /* mx8_bwd: 1 = dx_out_lo already carries that image (it was written as the dx_in_lo of the layer
                          above by avf_layer_bwd with mx8_bwd set); 0 = the layer quantises dx_out_lo itself (top layer) */
  const void* key_mask; /* optional token mask of Transformer.forward(x, mask) (heads.py:225-232; no reference caller passes one):
                          device bytes [batch, tokens], 1 = token kept - the reference's mask padded with a leading True.  A pair
                          (i, j) with either token dropped scores -FLT_MAX: a dropped query attends uniformly to all keys, a kept
                          query gives dropped keys zero weight; no gradient flows through a filled score.  bf16 layers with
                          dim_head 64 and <= 512 tokens apply it inside the MFMA attention kernels (head-resident forward, merged
                          backward); every other case runs the attention core on the fp32-arithmetic kernels (correct, not tuned). */
} avf_layer_cfg;

/* fp32 master parameters of one layer, in state_dict order:
 * layers.{i}.0.fn.norm.{weight,bias}, .0.fn.fn.to_qkv.weight [3I,D], .0.fn.fn.to_out.0.{weight [D,I],bias},
 * layers.{i}.1.fn.norm.{weight,bias}, .1.fn.fn.net.0.{weight [M,D],bias}, .1.fn.fn.net.3.{weight [D,M],bias} */
typedef struct avf_layer_params {
  const float *ln1_w, *ln1_b, *w_qkv, *w_out, *b_out, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
} avf_layer_params;

/* gradients, same shapes; every non-null pointer is OVERWRITTEN (not accumulated) */
typedef struct avf_layer_grads {
  float *ln1_w, *ln1_b, *w_qkv, *w_out, *b_out, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
} avf_layer_grads;

int avf_version(void);
/* sizeof(avf_layer_cfg) / sizeof(avf_layer_params) as this library was compiled: a binding checks its own struct
 * declarations against them before the first call (a shorter caller-side struct would be read past its end) */
size_t avf_sizeof_layer_cfg(void);
size_t avf_sizeof_layer_params(void);
const char* avf_last_error(void);
/* 1 if a gfx950 device is usable by this process */
int avf_device_ok(void);

/* ---- per-operator entry points --------------------------------------------------------- */

/* nn.LayerNorm(dim) forward - heads.py:178-185.  x fp32 [rows,dim] -> y (y_dtype) ; mean/rstd fp32 [rows]. */
int avf_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int64_t rows, int dim, float eps, void* stream);

/* LayerNorm backward.  dx = dres (nullable) + LN'(dy); optional low-precision copy dx_lo (bf16, nullable);
 * dgamma/dbeta [dim]; dcolsum (nullable) = column sums of dx (bias gradient of the Linear that produced x).
 * workspace >= avf_layernorm_bwd_workspace_bytes(rows, dim). */
size_t avf_layernorm_bwd_workspace_bytes(int64_t rows, int dim);
int avf_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx, void* dx_lo, float* dgamma,
                      float* dbeta, float* dcolsum, void* workspace, int64_t rows, int dim, void* stream);

/* column sums (bias gradients): out[c] = sum_r in[r,c].  workspace >= avf_colsum_workspace_bytes. */
size_t avf_colsum_workspace_bytes(int64_t rows, int cols);
int avf_colsum(const void* in, int in_dtype, int64_t rows, int cols, int64_t ld, float* out, void* workspace,
               void* stream);

/* fp32 -> bf16 cast (n elements) */
int avf_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* fp32 weight [rows,cols] -> bf16 copy [rows,cols] and bf16 transpose [cols,rows] (either may be null) */
int avf_prep_weight_bf16(const float* w, void* w_lo, void* w_t_lo, int rows, int cols, void* stream);

/* GEMM  C[M,N] = op(A)[M,K] * op(B)[K,N]  with fused epilogue.
 *   transA = 0: A stored [M,K] (lda)      transA = 1: A stored [K,M] (lda)
 *   transB = 0: B stored [K,N] (ldb)      transB = 1: B stored [N,K] (ldb)   (nn.Linear weight layout)
 * dtype AVF_F32: every form, any sizes.  dtype AVF_BF16: (transA=0,transB=1) "NT" with K%8==0, and
 * (transA=1,transB=0) "TN" (weight gradients, fp32 output, M%8==0, N%8==0).
 * c_dtype: storage type of C (and aux).  workspace (avf_gemm_workspace_bytes): the split-K slabs of a bf16 TN GEMM (required), of an
 * fp32 weight-gradient-shaped or skinny GEMM (optional: without it the unsplit general kernel runs). */
size_t avf_gemm_workspace_bytes(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K);
int avf_gemm(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
             const void* B, int64_t ldb, void* C, int64_t ldc, int c_dtype, int epilogue, const float* bias,
             const float* residual, int64_t ldres, void* aux, int64_t ldaux, void* workspace, void* stream);

/* Weight-stationary persistent NT GEMM (csrc/gemm_ws.hip): C[M,N] = epilogue(A[M,512] W[N,512]^T), the nn.Linear forward /
 * dX GEMMs of /root/reference/models/heads.py:191,195,212,215 at dim == 512.  W travels as its FRAGMENT-MAJOR image
 * (avf_pack_weight_ws of the row-major bf16 image, avf_pack_weight_ws_bytes(rows, cols) bytes; rows % 256 == 0, cols == 512):
 * each of the 8 wavefronts of a persistent workgroup keeps 32 weight rows in registers for the whole launch, only A streams.
 * Same epilogues, argument meaning and bit-for-bit the results of avf_gemm(AVF_BF16, 0, 1, ...); colsum (optional, with
 * workspace >= avf_gemm_nt_ws_workspace_bytes(M, N) - one partial row per persistent workgroup group, NOT the
 * avf_colsum_workspace_bytes of the standalone column sum): column sums of the stored C.  mx_q / mx_s (optional; DGELU with colsum and a
 * bf16 C, N % 32 == 0): also the MX-FP8 image of the fp32 values behind C (e4m3 [M][N], E8M0 [M][N/32]) - the fp8 mode's dGELU
 * GEMM keeps bf16 operands here and still feeds the fp8 GEMM behind it.  Errors if the shape does not qualify (M < 2048,
 * K != 512, N % 256 != 0). */
int avf_pack_weight_ws_ok(int64_t rows, int64_t cols);
size_t avf_pack_weight_ws_bytes(int64_t rows, int64_t cols);
size_t avf_gemm_nt_ws_workspace_bytes(int64_t M, int64_t N);
/* 1 when avf_gemm(AVF_BF16, 0, 1, ...) and the layer calls send this shape / epilogue to the persistent kernel (given 16-byte
 * aligned operands and no dropout); avf_gemm_nt_ws itself takes every shape the kernel can run */
int avf_gemm_nt_ws_dispatch(int64_t M, int64_t N, int64_t K, int epilogue, int c_dtype);
int avf_pack_weight_ws(const void* w_bf16, int64_t ldw, int64_t rows, int64_t cols, void* out, void* stream);
int avf_gemm_nt_ws(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B_packed, void* C, int64_t ldc,
                   int c_dtype, int epilogue, const float* bias, const void* residual, int64_t ldres, void* aux, int64_t ldaux,
                   void* workspace, float* colsum, void* mx_q, void* mx_s, void* stream);

/* The weight gradients of one layer as ONE grouped launch (autograd of nn.Linear, heads.py:191,195,212,215): up to four
 * C_i[M_i,N_i] (fp32, dense) = A_i[K,M_i]^T * B_i[K,N_i] (bf16, token-major, dense: lda = M_i, ldb = N_i) that share the
 * reduction length K (the B*N token rows); K % 64 == 0, M_i % 8 == 0, N_i % 8 == 0.  Split-K partial slabs live in
 * `workspace` (avf_gemm_tn_group_workspace_bytes) and are folded by a second launch. */
size_t avf_gemm_tn_group_workspace_bytes(int count, int64_t K, const int64_t* M, const int64_t* N);
int avf_gemm_tn_group(int count, int64_t K, const void* const* A, const void* const* B, float* const* C, const int64_t* M,
                      const int64_t* N, void* workspace, void* stream);

/* MX-FP8 operands (OCP microscaling: e4m3 elements, one E8M0 scale byte per 32 consecutive elements of a row) for the
 * forward nn.Linear GEMMs (heads.py:191,195,212) on v_mfma_scale_f32_16x16x128_f8f6f4 - BASELINE config 5.
 *   avf_quant_mx8:   x [rows,cols] (AVF_F32 | AVF_BF16, cols % 32 == 0) -> q [rows,cols] bytes, scales [rows,cols/32] bytes;
 *                    scale = floor(log2(block amax)) - 8 (+127), q = rne_e4m3(clamp(x * 2^-(scale-127), +-448)).
 *   avf_gemm_mx8_nt: C[M,N] = A[M,K] * B[N,K]^T from two such images (K % 128 == 0), fp32 accumulate, epilogues
 *                    AVF_EPI_NONE / BIAS_RES / BIAS_GELU / DGELU as avf_gemm; c_q / c_scales (optional, BIAS_GELU / DGELU, N % 32 == 0):
 *                    also the MX-FP8 image of the stored C, ready to be the next GEMM's A operand. */
/* e4m3 images of Wqkv, W1, W2, Wo (and, cfg.mx8_bwd, of the transposed W2, W1, Wo) of every layer of a stack from their
 * bf16 images, one launch (cfg.mx8_fwd = 1);
 * lowp[i] = the avf_layer_lowp_bytes buffer of layer i, after avf_layer_prepare_weights / the library Adam wrote it */
int avf_stack_quant_weights_mx8(const avf_layer_cfg* cfg, int layers, void* const* lowp, void* stream);
int avf_quant_mx8(int dtype, const void* x, int64_t rows, int64_t cols, void* q, void* scales, void* stream);
int avf_gemm_mx8_nt(int64_t M, int64_t N, int64_t K, const void* a_q, const void* a_scales, const void* b_q,
                    const void* b_scales, void* C, int64_t ldc, int c_dtype, int epilogue, const float* bias,
                    const float* residual, int64_t ldres, void* aux, int64_t ldaux, void* c_q, void* c_scales,
                    void* stream);
/* nn.LayerNorm forward (heads.py:178-185) writing the bf16 output AND its MX-FP8 image (dim % 32 == 0, dim <= 1536) */
int avf_layernorm_fwd_mx8(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                          void* y_q, void* y_scales, int64_t rows, int dim, float eps, void* stream);

/* nn.LayerNorm backward (avf_layernorm_bwd with bf16 dy) writing the bf16 image dx_lo of dx = dres + dLN AND its MX-FP8
 * image (dim % 32 == 0, dim <= 1536): the A operand of the dX GEMM below the LayerNorm in the fp8 mode; dx (fp32) optional */
int avf_layernorm_bwd_mx8(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, void* dx_lo, void* dx_q, void* dx_scales, float* dgamma, float* dbeta,
                          void* workspace, int64_t rows, int dim, void* stream);
/* avf_attn_fwd / avf_attn_bwd with the token mask of heads.py:225-232 (keep: device bytes [batch, tokens], 1 = kept; see
 * avf_layer_cfg.key_mask); fp32 arithmetic on fp32 or bf16 storage; workspace as avf_attn_bwd */
int avf_attn_fwd_masked(int dtype, const void* qkv, void* o, float* lse2, const void* keep, int batch, int tokens, int heads,
                        int dim_head, void* stream);
int avf_attn_bwd_masked(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv,
                        void* workspace, const void* keep, int batch, int tokens, int heads, int dim_head, void* stream);
/* bf16 attention forward (avf_attn_fwd) also writing the MX-FP8 image of o [B*N, I] - the A operand of to_out (heads.py:215)
 * in the fp8 mode.  Head-resident kernel only: dim_head 64, tokens <= 576 (error otherwise). */
int avf_attn_fwd_mx8(const void* qkv, void* o, float* lse2, void* o_q, void* o_scales, int batch, int tokens, int heads,
                     int dim_head, void* stream);
/* bf16 attention backward for PRE-SCALED queries (the q columns of qkv carry log2(e) / sqrt(dim_head), as the layer's Wqkv
 * image produces them) also writing the MX-FP8 image of dqkv [B*N, 3I] - the A operand of the dqkv -> dh1 GEMM (autograd of
 * heads.py:212) in the fp8 mode.  Merged kernel only: avf_attn_bwd_emits_mx8(tokens, dim_head) says (error otherwise). */
int avf_attn_bwd_emits_mx8(int tokens, int dim_head);
int avf_attn_bwd_mx8(const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, void* dqkv_q,
                     void* dqkv_scales, int batch, int tokens, int heads, int dim_head, void* stream);

/* Multi-head self-attention core - heads.py:222-237.  qkv [B*N, 3I] (q|k|v, head-major columns),
 * o [B*N, I], lse2 fp32 [B,H,N] = log2-domain log-sum-exp of the scaled scores (saved for backward). */
int avf_attn_fwd(int dtype, const void* qkv, void* o, float* lse2, int batch, int tokens, int heads,
                 int dim_head, void* stream);
/* backward: dqkv [B*N,3I] from do [B*N,I].  workspace >= avf_attn_bwd_workspace_bytes (holds delta [B,H,N]). */
size_t avf_attn_bwd_workspace_bytes(int batch, int tokens, int heads, int dim_head);
int avf_attn_bwd(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv,
                 void* workspace, int batch, int tokens, int heads, int dim_head, void* stream);

/* The same attention core (bf16 only) on a projection whose q columns are ALREADY multiplied by log2(e)/sqrt(dim_head):
 * the scores leave the MFMA in the log2 domain and the subtraction of the running maximum / of lse2 rides in the MFMA's
 * C operand (no per-score multiply-add).  This is what avf_layer_fwd / avf_layer_bwd run - they fold the factor into the
 * query rows of the bf16 copy of to_qkv.weight (heads.py:212) - and dq, dk, dv are still the gradients with respect to
 * the UNSCALED q, k, v.  workspace >= 2 * avf_attn_bwd_workspace_bytes(...). */
int avf_attn_fwd_qs(const void* qkv, void* o, float* lse2, int batch, int tokens, int heads, int dim_head, void* stream);
int avf_attn_bwd_qs(const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, void* workspace,
                    int batch, int tokens, int heads, int dim_head, void* stream);

/* Token-sequence plumbing of the callers either side of the stack (fp32, dim % 4 == 0, 16-byte aligned pointers).
 *  avf_fuse_tokens:   out[b, t, :] = (t < t_video ? clip[b, t, :] : audio[b, t - t_video, :]) + pos[t, :]  (pos nullable)
 *                     - the sequence-axis fusion torch.cat([clip, audio], 1) + pos_embedding of BASELINE.json's configs
 *                     (feature-axis fusion of the reference: models/avformer.py:95-103).
 *  avf_token_mean_fwd: out[b, :] = mean_t y[b, t, :]                      (x.mean(dim=1), models/tformer.py head)
 *  avf_token_mean_bwd: dy[b, t, :] = g[b, :] / tokens (nullable when dy_bf16 is given), the same in bf16 (dy_bf16, nullable), and colsum[d] =
 *                     sum_b g[b, d] (nullable) = the column sums of dy that the top layer's bias gradient needs. */
int avf_fuse_tokens(const float* clip, const float* audio, const float* pos, float* out, int batch, int t_video,
                    int t_audio, int dim, void* stream);
int avf_token_mean_fwd(const float* y, float* out, int batch, int tokens, int dim, void* stream);
/* the two ends of a bf16 residual stream (cfg.resid_bf16; dim % 8 == 0): the fused token build writing bf16, the token mean
 * reading bf16 (fp32 accumulation, fp32 result) */
int avf_fuse_tokens_bf16(const float* clip, const float* audio, const float* pos, void* out_bf16, int batch, int t_video,
                         int t_audio, int dim, void* stream);
int avf_token_mean_fwd_bf16(const void* y_bf16, float* out, int batch, int tokens, int dim, void* stream);
int avf_token_mean_bwd(const float* g, float* dy, void* dy_bf16, float* colsum, int batch, int tokens, int dim,
                       void* stream);

/* ---- token producers / consumers either side of the stack; fp32, one launch each ------------------
 * avf_bn1d_fwd / _bwd: nn.BatchNorm1d of AU_former (heads.py:263,293) on x [batch, features].  training != 0: batch
 *   statistics (and, when non-null, running_mean / running_var <- (1-momentum) r + momentum stat, unbiased variance, and
 *   *num_batches_tracked += 1); training == 0: the running statistics.  y = (x - mean) * invstd * gamma + beta; mean /
 *   invstd [features] are kept for backward.  Backward: dx (nullable), dgamma, dbeta (nullable) from dy.
 * avf_token_dots_fwd / _bwd: the per-token bias-free Linear(emb, 1) heads (heads.py:325-337, tformer.py:389-401):
 *   out[b, t] = tokens[b, t, :] . w[t, :] (w rows ldw apart); out rows are ldo apart and columns tokens_per_clip .. pad_to-1
 *   are zeroed (the [B,21] layout of train.py:136-138).  Backward: dtokens (nullable), dw [tokens_per_clip, emb] rows lddw apart
 *   (nullable).
 * avf_assemble_tokens: out[b, t, :] = (t < n_lead ? lead[t, :] : x[b, t - n_lead, :]) + pos[t, :] (pos nullable) - TFormer's
 *   cls token + positional table (vformer.py:279-287; n_lead = 1), or a bare positional add (n_lead = 0).
 * avf_cat_features: out[b, t, :] = concat(a[b, t, :emb_a], v[b, t, :emb_v]) + pos[t, :] - avformer.py:100 + tformer.py:383-386.
 * avf_transpose_add: in [batch, rows, cols] -> out [batch, cols, rows] (+ pos [cols, rows], nullable): the feature-map <-> token
 *   permutes of ResFormer.forward (sformer.py:316-318, 326-327).
 * avf_zero_cols: out[r, c0..c1) = 0. */
int avf_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                 int64_t* num_batches_tracked, float* y, float* mean, float* invstd, int batch, int features, float eps,
                 float momentum, int training, void* stream);
int avf_bn1d_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd, float* dx,
                 float* dgamma, float* dbeta, int batch, int features, int training, void* stream);
int avf_token_dots_fwd(const float* tokens, const float* w, int64_t ldw, float* out, int64_t ldo, int pad_to, int batch,
                       int tokens_per_clip, int emb, void* stream);
int avf_token_dots_bwd(const float* dout, int64_t ldo, const float* tokens, const float* w, int64_t ldw, float* dtokens,
                       float* dw, int64_t lddw, int batch, int tokens_per_clip, int emb, void* stream);
int avf_assemble_tokens(const float* x, const float* lead, const float* pos, float* out, int batch, int patches, int n_lead,
                        int dim, void* stream);
int avf_cat_features(const float* a, const float* v, const float* pos, float* out, int batch, int tokens_per_clip, int emb_a,
                     int emb_v, void* stream);
int avf_transpose_add(const float* in, const float* pos, float* out, int batch, int rows, int cols, void* stream);
int avf_zero_cols(float* out, int64_t ld, int rows, int c0, int c1, void* stream);
/* counter[0] += 1; snapshot[0] = counter[0] (device int64 scalars, one launch, graph-capturable): the dropout seed of one forward
 * of a stack (nn.Dropout at heads.py:194,196,216 draws fresh masks per call; the kernels of a forward and of its backward read
 * the snapshot through cfg.seed_dev). */
int avf_seed_advance(int64_t* counter, int64_t* snapshot, void* stream);
/* a small nn.Linear on a few rows written into a zero-padded row - the AU logits of a pooled feature in the reference's [B,21]
 * layout (avformer.py:101-105): out[r, o] = x[r,:] . w[o,:] + bias[o] (o < out_features), 0 up to `width`; one launch.
 * Backward: dx = dout[:, :O] w, dw = dout[:, :O]^T x, db = column sums (any of them may be null); dout rows are ldd apart. */
int avf_linear_pad_fwd(const float* x, const float* w, const float* bias, float* out, int rows, int in_features, int out_features,
                       int width, void* stream);
int avf_linear_pad_bwd(const float* dout, int64_t ldd, const float* x, const float* w, float* dx, float* dw, float* db, int rows,
                       int in_features, int out_features, void* stream);

/* AULoss - loss.py:63-103.  logits/labels fp32 [rows, 12] (ld given); rows whose FIRST label == ignore
 * are dropped; loss[0] = mean over kept rows x 12 of BCE-with-logits(pos_weight); grad_unit [rows,12]
 * (contiguous) = d loss / d logits.  All rows dropped => NaN (as the reference). */
int avf_au_loss(const float* logits, int64_t ld_logits, const float* labels, int64_t ld_labels, const float* pos_weight,
                float ignore, int rows, int ncls, float* loss, float* grad_unit, void* stream);
/* The same as a (sum, count) pair for batch-sharded data parallelism - loss.py:85-102 is a RATIO, so ranks with different
 * numbers of ignored rows must reduce numerator and denominator separately: sum_count[0] = sum over kept rows of the row's
 * mean BCE, sum_count[1] = kept rows, grad_unit = d sum_count[0] / d logits.  All rows dropped => (0, 0), zero gradient. */
int avf_au_loss_sum(const float* logits, int64_t ld_logits, const float* labels, int64_t ld_labels,
                    const float* pos_weight, float ignore, int rows, int ncls, float* sum_count, float* grad_unit,
                    void* stream);
/* Either of the two (sum_mode 0 / 1) with the gradient laid out as the model's OUTPUT row: grad_wide [rows, width] contiguous,
 * columns 0..ncls-1 as grad_unit above, ncls..width-1 zero - the gradient of the reference's [B,21] row whose first 12 slots are
 * the AU logits (train.py:136-138, loss on out[:, :12]: avformer.py get_au_loss), in the same launch instead of the fill + copy
 * autograd's slice backward adds.  `loss`: one float (sum_mode 0) or the (sum, count) pair (sum_mode 1). */
int avf_au_loss_wide(const float* logits, int64_t ld_logits, const float* labels, int64_t ld_labels, const float* pos_weight,
                     float ignore, int rows, int ncls, int width, int sum_mode, float* loss, float* grad_wide, void* stream);

/* ---- one transformer layer (heads.py:246-255), forward and backward ------------------------ */
size_t avf_layer_saved_bytes(const avf_layer_cfg* cfg);     /* activations kept for backward        */
size_t avf_layer_lowp_bytes(const avf_layer_cfg* cfg);      /* bf16 weight copies (+transposes)     */
size_t avf_layer_workspace_bytes(const avf_layer_cfg* cfg); /* scratch, reusable across layers      */
size_t avf_layer_grad_stream_bytes(const avf_layer_cfg* cfg); /* bytes of one dx_out_lo / dx_in_lo buffer of avf_layer_bwd:
                                                                 R*D bf16, plus the MX-FP8 image behind it when cfg.mx8_bwd */
/* refresh the bf16 weight copies from the fp32 masters (no-op in AVF_F32 mode) */
int avf_layer_prepare_weights(const avf_layer_cfg* cfg, const avf_layer_params* p, void* lowp, void* stream);

/* Adam step of one layer fused with the refresh of its bf16 weight copies - torch.optim.Adam(lr, betas, eps,
 * weight_decay) as the reference's training loop uses it (train.py:318-322; L2 decay added to the gradient, bias
 * correction by `step`, amsgrad off), arithmetic as torch's fused kernel.  p is updated IN PLACE (the const of
 * avf_layer_params is cast away), exp_avg / exp_avg_sq likewise; a tensor whose gradient pointer is null is not
 * updated (its bf16 copies are still rewritten).  `step` is a device float holding the number of THIS update (>= 1),
 * read at run time (graph-capturable); null means 1.  lowp as for avf_layer_fwd (ignored in AVF_F32 mode). */
int avf_layer_adam_step(const avf_layer_cfg* cfg, const avf_layer_params* p, const avf_layer_grads* g,
                        const avf_layer_grads* exp_avg, const avf_layer_grads* exp_avg_sq, void* lowp, float lr,
                        float beta1, float beta2, float eps, float weight_decay, const float* step, void* stream);

/* the same for `layers` consecutive layers of one stack (arrays of `layers` structs, lowp[l] per layer): thirteen layers per
 * launch (the descriptor table is a 12.6 KB kernel argument) */
int avf_stack_adam_step(const avf_layer_cfg* cfg, int layers, const avf_layer_params* p, const avf_layer_grads* g,
                        const avf_layer_grads* exp_avg, const avf_layer_grads* exp_avg_sq, void* const* lowp, float lr,
                        float beta1, float beta2, float eps, float weight_decay, const float* step, void* stream);

/* the same Adam update for `count` arbitrary fp32 tensors (host arrays of device pointers and element counts): the
 * parameters around the stacks (positional embedding, AU head).  A tensor whose gradient pointer is null is skipped. */
int avf_adam_step_tensors(int count, float* const* p, const float* const* g, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                          float weight_decay, const float* step, void* stream);

/* One optimizer step's launches, collected (torch.optim.Adam.step() over all parameters, train.py:237): between _begin and _end
 * on the calling thread, avf_layer_adam_step / avf_stack_adam_step / avf_adam_step_tensors append to one descriptor table
 * instead of launching; the table is launched when it is full (143 tensors), when the hyper-parameters, the step pointer or the
 * stream of a call differ from the pending ones, and by _end.  The reference's real model has five small stacks and a dozen
 * loose tensors: one launch instead of six.  Every pointer handed over must stay valid until _end returns.
 * The session is THREAD-LOCAL: _begin, every step call and _end / _abort must come from the same thread, with the same device
 * current (the table is launched on the stream of the calls it collected).  _abort closes the session without launching the
 * pending table (the caller failed while collecting the step); tables that were already launched stay launched.
 * The table is a ~12.6 KB by-value kernel argument: avf_selftest_adam_table() launches a full one (143 descriptors, one element
 * each) and checks every element - run it once per process where kernel arguments above 4 KB are in doubt. */
int avf_adam_batch_begin(void);
int avf_adam_batch_end(void);
int avf_adam_batch_abort(void);
int avf_selftest_adam_table(void* stream);

/* x_out = layer(x_in); x_in, x_out [B*N, D] (may not alias): fp32, or bf16 when cfg.resid_bf16 is set. */
int avf_layer_fwd(const avf_layer_cfg* cfg, const avf_layer_params* p, const void* lowp, const void* x_in,
                  void* x_out, void* saved, void* workspace, void* stream);

/* dx_in (fp32) and all parameter gradients from dx_out (fp32).  dx_out_lo: optional bf16 copy of dx_out
 * (null => made internally); dx_in_lo: optional bf16 copy of dx_in to hand to the previous layer (with dropout
 * active it already carries layer_index-1's site-2 mask, which is what that layer's MLP gradients consume).
 * dx_out_colsum: optional [D] column sums of dx_out (= this layer's b2 gradient) already computed by the
 * caller's previous call; dx_in_colsum: optional [D] output, column sums of dx_in for the next call.
 * dx_in may alias dx_out (dx_out_lo / dx_in_lo must then be distinct buffers).  x_in: the tensor avf_layer_fwd was given
 * (bf16 when cfg.resid_bf16); the gradients dx_* are fp32 / bf16 images independently of it. */
int avf_layer_bwd(const avf_layer_cfg* cfg, const avf_layer_params* p, const void* lowp, const void* x_in,
                  const void* saved, const float* dx_out, const void* dx_out_lo, const float* dx_out_colsum,
                  float* dx_in, void* dx_in_lo, float* dx_in_colsum, const avf_layer_grads* g, void* workspace,
                  void* stream);

/* test aid: the keep/(1-p) factors (0 or 1/(1-p_eff)) dropout site `site` (0 after to_out, 1 after GELU, 2 after
 * net.3) of layer `layer_index` applies to a [rows, cols] activation, as fp32 [rows, cols] (cols % 4 == 0). */
int avf_dropout_factors(uint32_t seed_lo, uint32_t seed_hi, int layer_index, int site, float p, int64_t rows, int cols,
                        float* out, void* stream);

/* ---- arithmetic of the AVF_F32 (parity) mode, process-wide (round 6) --------------------------------------------
 * The AVF_F32 GEMMs and the attention core (the fp32 aten::mm / bmm / softmax under models/heads.py:191-196, 212-238) run
 *   mode 1 ("bf16x3", default): every fp32 operand split as x = hi + lo (two bf16), a b ~ hi hi + hi lo + lo hi on
 *           v_mfma_f32_16x16x32_bf16 with fp32 accumulation: <= 3 * 2^-16 = 4.6e-5 relative error per product in the worst case,
 *           4e-6 typical (measured relative Frobenius error of a GEMM), 3 MFMAs per product;
 *   mode 0 ("f32"): the f32-input MFMA v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain, 1/16 of the bf16 rate).
 * Both hold north_star's logits rtol 1e-3; shapes the bf16x3 kernels do not take (tiles below 64 rows, ragged K on the
 * contiguous axis, token masks, bf16 storage) run the mode-0 kernels in either mode.  Returns the previous mode. */
int avf_set_f32_arith(int mode);
int avf_get_f32_arith(void);

/* After a FAILED hipGraph capture of a step (something that cannot be recorded was issued while `stream` was capturing - e.g. a
 * host-synchronising collective): ends a capture still open on `stream`, discards its graph and clears the runtime's sticky
 * last-error so that the caller can continue with eager launches.  Returns the HIP error code that was pending (0: none). */
int avf_hip_error_reset(void* stream);

/* A last line for a process that may die inside an OPTIONAL step (bench.py's multi-rank hipGraph attempt, which follows a completed
 * eager measurement): while armed, SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL write `line` to `fd` (write(2); fd < 0: nothing) and
 * leave with _exit(0).  _disarm restores the previous handlers.  Process-wide; not for product code paths. */
int avf_crash_line_arm(const char* line, int fd);
int avf_crash_line_disarm(void);

/* ---- optional HIP-event timing per kernel class (bench.py's roofline line) --------------------------
 * classes: 0 gemm_bf16_nt, 1 gemm_bf16_tn(+fold), 2 gemm_f32, 3 attn_fwd, 4 attn_bwd, 5 layernorm, 6 other, 7 gemm_mx8_nt.
 * enable(1) resets the records; read() synchronises the recorded events and sums them.  Every class but 2 and 6
 * attaches one event pair to each kernel dispatch (its own begin / end timestamps, as a profiler reports them; `launches`
 * counts dispatches); 2 and 6 record a pair around the launches of one call (~2 us of command-processor time each). */
int avf_timing_enable(int on);
int avf_timing_read(int cls, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* ---- hardware self-tests used by tests/ (MFMA fragment maps, transposed LDS read) ----------- */
int avf_selftest_mfma_bf16(const void* a_bf16_16x32, const void* b_bf16_32x16, float* c_16x16, void* stream);
int avf_selftest_mfma_f32(const float* a_16x4, const float* b_4x16, float* c_16x16, void* stream);
int avf_selftest_tr16(const void* tile_bf16_32x16, void* out_bf16_64x8, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVFORMER_HIP_H */
