// layer.hip - host-side orchestration of one transformer layer (forward and backward) on a HIP stream.
//
// Reference: models/heads.py:246-255 - one (Residual(PreNorm(Attention)), Residual(PreNorm(FeedForward)))
// pair; SURVEY.md appendix A gives the op order and the backward formulas.  Everything here is
// enqueue-only (no allocation, no synchronisation) so a caller may capture it into a hipGraph.
//
// forward:   h1 = LN1(x)            -> qkv = h1 Wqkv^T          -> o = attn(qkv)
//            x_mid = o Wo^T + bo + x -> h2 = LN2(x_mid)          -> u = h2 W1^T + b1, g = gelu(u)
//            x_out = g W2^T + b2 + x_mid
// backward:  du = (dx_out W2) o gelu'(u) ; dW2 = dx_out^T g ; db2 = colsum(dx_out)
//            dh2 = du W1 ; dW1 = du^T h2 ; db1 = colsum(du)
//            dx_mid = dx_out + LN2'(dh2) ; dgamma2, dbeta2 ; dbo = colsum(dx_mid)   [fused in LN bwd]
//            do = dx_mid Wo ; dWo = dx_mid^T o ; dqkv = attn'(do) ; dWqkv = dqkv^T h1 ; dh1 = dqkv Wqkv
//            dx_in = dx_mid + LN1'(dh1) ; dgamma1, dbeta1
#include <mutex>
#include <vector>

#include "common.hpp"

namespace avf {

namespace {

struct Dims {
  int64_t R;  // rows = batch * tokens
  int D, H, dh, I, M, B, N;
  int dt;     // compute dtype
  size_t es;  // element size of compute dtype
  float p;    // dropout probability (0 = off)
  uint64_t seed;
  const uint64_t* seed_dev;
  int layer;
  bool gs16;  // backward keeps the residual gradient stream in bf16 (no fp32 dx between the LayerNorm backward kernels)
  bool gsd;   // ... with LIVE dropout: every hand-off is two bf16 images, the stream (never masked) and, behind it in the same
              // buffer, what the Linear behind the dropout site sees (masked, rescaled) - all-bf16 streams only (resid_bf16)
  bool mx;    // forward nn.Linear GEMMs fed by LayerNorm / GELU (qkv, mlp1, mlp2) take MX-FP8 operands (config 5)
  bool rs16;  // the FORWARD residual stream (x_in, x_mid, x_out) is stored in bf16 (statistics / accumulation stay fp32)
  int xdt;    // storage type of the residual stream
  float p0;   // dropout probability of site 0 (after to_out): 0 when to_out is nn.Identity (no Dropout in it, heads.py:214-217)
  const void* keep;  // optional token mask [B, N] bytes (1 = kept), heads.py:225-232: attention then runs on the fp32-arithmetic kernels
  bool mxb;   // backward dX GEMMs fed by LayerNorm backward / the dGELU epilogue take MX-FP8 operands too (config 5)
  bool gy_mx; // the caller's dx_out_lo buffer already carries the MX-FP8 image behind the bf16 one
};

// bf16 gradient-stream buffers in the mx8_bwd mode: [R, D] bf16 | [R, D] e4m3 | [R, D / 32] E8M0, each part 256-aligned
inline size_t grad_q_off(int64_t R, int D) { return align_up((size_t)R * D * 2, 256); }
inline size_t grad_s_off(int64_t R, int D) { return grad_q_off(R, D) + align_up((size_t)R * D, 256); }
inline size_t grad_stream_bytes(int64_t R, int D, bool mxb, bool gsd = false) {
  if (gsd) return 2 * grad_q_off(R, D);  // [R, D] bf16 stream | [R, D] bf16 masked image (at grad_q_off)
  return mxb ? grad_s_off(R, D) + align_up((size_t)R * D / 32, 256) : (size_t)R * D * 2;
}

int make_dims(const avf_layer_cfg* c, Dims* d) {
  AVF_REQUIRE(c, "layer: null cfg");
  AVF_REQUIRE(c->batch > 0 && c->tokens > 0 && c->dim > 0 && c->heads > 0 && c->dim_head > 0 && c->mlp_dim > 0,
              "layer: non-positive shape in cfg");
  AVF_REQUIRE(c->dtype == AVF_F32 || c->dtype == AVF_BF16, "layer: bad dtype %d", c->dtype);
  AVF_REQUIRE(c->dropout_p >= 0.0f && c->dropout_p < 1.0f, "layer: dropout_p=%g out of range", (double)c->dropout_p);
  AVF_REQUIRE(c->dropout_p == 0.0f || (c->dim % 4 == 0 && c->dim <= 1536 && c->mlp_dim % 4 == 0),
              "layer: dropout needs dim %% 4 == 0, dim <= 1536");
  // the masks compare 16-bit uniforms with round(p * 65536): a p that rounds to 0 would leave "dropout_p > 0" (which selects the
  // masked gradient images of the bf16 streams) and the kernels' own predicate (thresh16 != 0) in disagreement - refuse it
  AVF_REQUIRE(c->dropout_p == 0.0f || make_drop(c->dropout_p, 0, 0, 0).thresh16 != 0,
              "layer: dropout_p=%g is below the mask resolution (2^-17): pass 0", (double)c->dropout_p);
  // nn.Identity to_out (heads == 1 && dim_head == dim, heads.py:207; never instantiated by the reference): the caller passes the
  // identity matrix as w_out and zeros as b_out (x 1.0 and + 0.0 are exact in fp32 and in bf16 with fp32 accumulation, so the
  // projection GEMM returns the attention output bit for bit) and the library drops the dropout site that nn.Identity lacks
  AVF_REQUIRE(c->project_out == 1 || (c->heads == 1 && c->dim_head == c->dim),
              "layer: project_out = 0 is the nn.Identity to_out case: it needs heads == 1 and dim_head == dim (heads.py:207)");
  d->B = c->batch; d->N = c->tokens; d->D = c->dim; d->H = c->heads; d->dh = c->dim_head;
  d->I = c->heads * c->dim_head; d->M = c->mlp_dim; d->R = (int64_t)c->batch * c->tokens;
  d->dt = c->dtype; d->es = c->dtype == AVF_BF16 ? 2 : 4;
  d->p = c->dropout_p; d->seed = ((uint64_t)c->seed_hi << 32) | c->seed_lo; d->layer = c->layer_index;
  d->seed_dev = (const uint64_t*)c->seed_dev;
  d->gs16 = c->grad_stream_bf16 != 0;
  d->gsd = d->gs16 && c->dropout_p > 0.0f;
  AVF_REQUIRE(!d->gs16 || (c->dtype == AVF_BF16 && c->dim <= 1536), "layer: grad_stream_bf16 needs the bf16 path and dim <= 1536");
  AVF_REQUIRE(!d->gsd || (c->resid_bf16 != 0 && c->mx8_fwd == 0 && c->dim % 8 == 0),
              "layer: grad_stream_bf16 with dropout_p > 0 needs resid_bf16 (all-bf16 streams), no mx8 and dim %% 8 == 0");
  d->mx = c->mx8_fwd != 0;
  AVF_REQUIRE(!d->mx || (c->dtype == AVF_BF16 && c->dim % 128 == 0 && c->mlp_dim % 128 == 0 && c->dim <= 1536),
              "layer: mx8_fwd needs the bf16 path with dim and mlp_dim multiples of 128 (dim=%d mlp_dim=%d)", c->dim,
              c->mlp_dim);
  d->p0 = c->project_out ? c->dropout_p : 0.0f;
  d->keep = c->key_mask;
  d->mxb = c->mx8_bwd != 0;
  d->gy_mx = c->dx_out_mx8 != 0;
  AVF_REQUIRE(!d->mxb || d->mx, "layer: mx8_bwd needs mx8_fwd");
  d->rs16 = c->resid_bf16 != 0;
  d->xdt = d->rs16 ? AVF_BF16 : AVF_F32;
  AVF_REQUIRE(!d->rs16 || (c->dtype == AVF_BF16 && c->dim % 8 == 0 && c->dim <= 1536),
              "layer: resid_bf16 needs the bf16 path, dim %% 8 == 0 and dim <= 1536 (dim=%d)", c->dim);
  if (c->dtype == AVF_BF16) {
    AVF_REQUIRE(d->D % 8 == 0 && d->I % 8 == 0 && d->M % 8 == 0, "layer(bf16): dim, inner and mlp_dim must be multiples of 8");
    AVF_REQUIRE(d->dh == 32 || d->dh == 64, "layer(bf16): dim_head must be 32 or 64 (got %d)", d->dh);
  } else {
    AVF_REQUIRE(d->D % 4 == 0 && d->I % 4 == 0 && d->M % 4 == 0, "layer(f32): dim, inner and mlp_dim must be multiples of 4");
  }
  return 0;
}

struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  }
};

struct Saved {
  void *h1, *qkv, *o, *h2, *u, *g, *x_mid;  // x_mid: fp32, or bf16 on the bf16 residual stream
  float *mean1, *rstd1, *lse2, *mean2, *rstd2;
};
size_t carve_saved(const Dims& d, void* base, Saved* s) {
  Carver c(base);
  Saved t;
  t.h1 = c.take(d.R * d.D * d.es);
  t.mean1 = (float*)c.take(d.R * 4);
  t.rstd1 = (float*)c.take(d.R * 4);
  t.qkv = c.take(d.R * 3 * d.I * d.es);
  t.o = c.take(d.R * d.I * d.es);
  t.lse2 = (float*)c.take((size_t)d.B * d.H * d.N * 4);
  t.x_mid = c.take(d.R * d.D * (d.rs16 ? 2 : 4));
  t.h2 = c.take(d.R * d.D * d.es);
  t.mean2 = (float*)c.take(d.R * 4);
  t.rstd2 = (float*)c.take(d.R * 4);
  t.u = c.take(d.R * d.M * d.es);
  t.g = c.take(d.R * d.M * d.es);
  if (s) *s = t;
  return c.off;
}

struct LowP {
  void *wqkv, *wqkv_t, *wo, *wo_t, *w1, *w1_t, *w2, *w2_t;
  void *wqkv_q, *wqkv_s, *w1_q, *w1_s, *w2_q, *w2_s;  // mx8_fwd only
  void *wo_q, *wo_s;                                  // mx8_fwd: out-projection (used when the attention kernel emits the image of o)
  void *w2t_q, *w2t_s, *w1t_q, *w1t_s, *wot_q, *wot_s;  // mx8_bwd: images of the transposed weights (K = out features)
  void *wqkvt_q, *wqkvt_s;                               // mx8_bwd: image of Wqkv^T [D, 3I] (dqkv -> dh1, K = 3I)
  LowpWs ws;                            // fragment-major images for the weight-stationary GEMM (gemm_ws.hip), null if unfit
};
size_t carve_lowp(const Dims& d, void* base, LowP* l) {
  if (d.dt != AVF_BF16) {
    if (l) memset(l, 0, sizeof(*l));
    return 0;
  }
  Carver c(base);
  LowP t;
  t.wqkv = c.take((size_t)3 * d.I * d.D * 2);
  t.wqkv_t = c.take((size_t)3 * d.I * d.D * 2);
  t.wo = c.take((size_t)d.D * d.I * 2);
  t.wo_t = c.take((size_t)d.D * d.I * 2);
  t.w1 = c.take((size_t)d.M * d.D * 2);
  t.w1_t = c.take((size_t)d.M * d.D * 2);
  t.w2 = c.take((size_t)d.D * d.M * 2);
  t.w2_t = c.take((size_t)d.D * d.M * 2);
  if (d.mx) {  // MX-FP8 forward images (e4m3 bytes + one scale byte per 32) of Wqkv, W1, W2
    t.wqkv_q = c.take((size_t)3 * d.I * d.D); t.wqkv_s = c.take((size_t)3 * d.I * d.D / 32);
    t.w1_q = c.take((size_t)d.M * d.D);       t.w1_s = c.take((size_t)d.M * d.D / 32);
    t.w2_q = c.take((size_t)d.D * d.M);       t.w2_s = c.take((size_t)d.D * d.M / 32);
    t.wo_q = c.take((size_t)d.D * d.I);       t.wo_s = c.take((size_t)d.D * d.I / 32);
    t.w2t_q = c.take((size_t)d.D * d.M);      t.w2t_s = c.take((size_t)d.D * d.M / 32);
    t.w1t_q = c.take((size_t)d.D * d.M);      t.w1t_s = c.take((size_t)d.D * d.M / 32);
    t.wot_q = c.take((size_t)d.D * d.I);      t.wot_s = c.take((size_t)d.D * d.I / 32);
    t.wqkvt_q = c.take((size_t)3 * d.I * d.D); t.wqkvt_s = c.take((size_t)3 * d.I * d.D / 32);
  } else {
    t.wqkv_q = t.wqkv_s = t.w1_q = t.w1_s = t.w2_q = t.w2_s = nullptr;
    t.wo_q = t.wo_s = t.w2t_q = t.w2t_s = t.w1t_q = t.w1t_s = t.wot_q = t.wot_s = nullptr;
    t.wqkvt_q = t.wqkvt_s = nullptr;
  }
  // LAST (avf_*_adam_step finds the eight bf16 images by their offsets from the front): the fragment-major images
  t.ws.wqkv_p = pack_ws_ok(3 * d.I, d.D) ? c.take(pack_ws_bytes(3 * d.I, d.D)) : nullptr;
  t.ws.wo_p = pack_ws_ok(d.D, d.I) ? c.take(pack_ws_bytes(d.D, d.I)) : nullptr;
  t.ws.w1_p = pack_ws_ok(d.M, d.D) ? c.take(pack_ws_bytes(d.M, d.D)) : nullptr;
  t.ws.w2t_p = pack_ws_ok(d.M, d.D) ? c.take(pack_ws_bytes(d.M, d.D)) : nullptr;
  t.ws.wot_p = pack_ws_ok(d.I, d.D) ? c.take(pack_ws_bytes(d.I, d.D)) : nullptr;
  if (!base) t.ws = LowpWs{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (l) *l = t;
  return c.off;
}

// the four weight-gradient GEMMs of a layer as one grouped launch: dW2 = gy^T g, dW1 = du^T h2, dWo = gm^T o,
// dWqkv = dqkv^T h1 (all reduce over the R token rows)
TnGroupArgs dw_group(const Dims& d, const void* gy, const void* g_act, const void* du, const void* h2, const void* gm,
                     const void* o, const void* dqkv, const void* h1, const avf_layer_grads* g) {
  TnGroupArgs a;
  memset(&a, 0, sizeof(a));
  a.count = 4;
  a.K = d.R;
  // order: attention half, then MLP half - the XCD-aware block order of the 256 x 128 kernel cuts the tile list in the
  // middle, and the two halves then share no operand panel (3 I D + D I tiles | 2 M D tiles: equal when I = D, M = 2 D)
  a.A[0] = dqkv; a.B[0] = h1; a.C[0] = g ? g->w_qkv : nullptr; a.M[0] = 3 * d.I; a.N[0] = d.D;
  a.A[1] = gm;   a.B[1] = o;  a.C[1] = g ? g->w_out : nullptr; a.M[1] = d.D;     a.N[1] = d.I;
  a.A[2] = du;   a.B[2] = h2; a.C[2] = g ? g->w1 : nullptr;    a.M[2] = d.M;     a.N[2] = d.D;
  a.A[3] = gy;   a.B[3] = g_act; a.C[3] = g ? g->w2 : nullptr; a.M[3] = d.D;     a.N[3] = d.M;
  for (int i = 0; i < 4; ++i) { a.lda[i] = a.M[i]; a.ldb[i] = a.N[i]; }
  return a;
}

struct Work {
  void *du, *dh, *d_o, *dqkv, *dx_mid_lo, *dx_out_lo, *ln_ws, *ln_ws1, *cs_ws, *gemm_ws;
  float *dx_mid, *delta;
  void *hq, *hs, *gq, *gs;  // mx8_fwd only: MX-FP8 images of the LayerNorm output and of gelu(u), forward scratch
  void *oq, *os;            // mx8_fwd: image of the attention output (forward scratch)
  void *duq, *dus, *mq, *ms, *gyq, *gys;  // mx8_bwd: images of du, of dx_mid, and of dx_out when the caller brought none
  void *dqq, *dqs;                        // mx8_bwd: image of dqkv (written by the merged attention backward)
  void *dx_out_s, *dx_mid_m;  // gsd: the unmasked bf16 stream image of dx_out (top of the stack), the masked image of dx_mid
  float *gy_m, *gm_m;       // fp32 mode with live dropout: masked copies of dx_out / dx_mid (what the Linears behind sites 2 / 0 see)
  float* small_part;        // short-sequence backward: per-clip partial rows (pb1 [B][M] | pln2 [B][3D] | pln1 [B][3D])
};
size_t carve_work(const Dims& d, void* base, Work* w) {
  Carver c(base);
  Work t;
  t.du = c.take(d.R * d.M * d.es);
  t.dh = c.take(d.R * d.D * d.es);
  t.d_o = c.take(d.R * d.I * d.es);
  t.dqkv = c.take(d.R * 3 * d.I * d.es);
  t.dx_mid = (float*)c.take(d.R * d.D * 4);
  t.dx_mid_lo = c.take(d.dt == AVF_BF16 ? d.R * d.D * 2 : 0);
  t.dx_out_lo = c.take(d.dt == AVF_BF16 ? d.R * d.D * 2 : 0);
  t.delta = (float*)c.take((size_t)d.B * d.H * d.N * 4 * 2);  // delta, then the negated lse2 rows (bf16 backward)
  t.ln_ws = c.take(layernorm_bwd_ws(d.R, d.D));
  t.ln_ws1 = c.take(layernorm_bwd_ws(d.R, d.D));  // LN1 partials (its fold may be deferred past LN2's)
  const int maxc = d.M > d.D ? d.M : d.D;
  size_t csb = colsum_ws(d.R, maxc);
  if (d.dt == AVF_BF16 && gemm_nt_colsum_ws(d.R, d.M) > csb) csb = gemm_nt_colsum_ws(d.R, d.M);
  t.cs_ws = c.take(csb);
  size_t g = 0;
  if (d.dt == AVF_BF16) {
    size_t a = gemm_bf16_tn_ws(3 * d.I, d.D, d.R), b = gemm_bf16_tn_ws(d.D, d.I, d.R);
    size_t e = gemm_bf16_tn_ws(d.M, d.D, d.R), f = gemm_bf16_tn_ws(d.D, d.M, d.R);
    g = a > b ? a : b;
    g = g > e ? g : e;
    g = g > f ? g : f;
    TnGroupArgs ga = dw_group(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    const size_t gg = gemm_bf16_tn_group_ws(ga);
    g = g > gg ? g : gg;
  } else {  // parity mode: the weight-gradient GEMMs of narrow layers split their long token reduction (gemm_f32_ws)
    const size_t a = gemm_f32_ws(3 * d.I, d.D, d.R), b = gemm_f32_ws(d.D, d.I, d.R);
    const size_t e = gemm_f32_ws(d.M, d.D, d.R), f = gemm_f32_ws(d.D, d.M, d.R);
    g = a > b ? a : b;
    g = g > e ? g : e;
    g = g > f ? g : f;
    TnGroupArgs ga = dw_group(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    const size_t gg = gemm_f32x3_tn_group_ws(ga);   // (independent of the arithmetic selected now: the buffer serves either)
    g = g > gg ? g : gg;
  }
  t.gemm_ws = c.take(g);
  t.small_part = (float*)c.take(small_layer_ok(d.dt, d.N, d.D, d.H, d.dh, d.M) ? small_bwd_partial_floats(d.B, d.D, d.M) * 4 : 0);
  if (d.mx) {
    t.hq = c.take(d.R * d.D); t.hs = c.take(d.R * d.D / 32);
    t.gq = c.take(d.R * d.M); t.gs = c.take(d.R * d.M / 32);
    t.oq = c.take(d.R * d.I); t.os = c.take(d.R * d.I / 32);
  } else {
    t.hq = t.hs = t.gq = t.gs = t.oq = t.os = nullptr;
  }
  if (d.mxb) {
    t.duq = c.take(d.R * d.M); t.dus = c.take(d.R * d.M / 32);
    t.mq = c.take(d.R * d.D);  t.ms = c.take(d.R * d.D / 32);
    t.gyq = c.take(d.R * d.D); t.gys = c.take(d.R * d.D / 32);
    t.dqq = c.take(d.R * 3 * d.I); t.dqs = c.take(d.R * 3 * d.I / 32);
  } else {
    t.duq = t.dus = t.mq = t.ms = t.gyq = t.gys = nullptr;
    t.dqq = t.dqs = nullptr;
  }
  t.dx_out_s = c.take(d.gsd ? d.R * d.D * 2 : 0);
  t.dx_mid_m = c.take(d.gsd ? d.R * d.D * 2 : 0);
  const bool f32_drop = d.dt == AVF_F32 && d.p > 0.f;
  t.gy_m = (float*)c.take(f32_drop ? d.R * d.D * 4 : 0);
  t.gm_m = (float*)c.take(f32_drop ? d.R * d.D * 4 : 0);
  if (w) *w = t;
  return c.off;
}

// C[R, out] = A[R, in] * W[out, in]^T  (nn.Linear forward)
int linear_fwd(const Dims& d, const void* A, int in, const void* W, int out, void* C, int c_dtype, int epi,
               const float* bias, const void* res, void* aux, hipStream_t s, const DropCfg& drop = kNoDrop,
               const void* Wp = nullptr) {  // Wp: fragment-major image of W (weight-stationary kernel, gemm_ws.hip)
  GemmArgs a;
  a.dtype = d.dt; a.transA = 0; a.transB = 1;
  a.M = d.R; a.N = out; a.K = in;
  a.A = A; a.lda = in; a.B = W; a.ldb = in;
  a.C = C; a.ldc = out; a.c_dtype = c_dtype; a.epilogue = epi;
  a.bias = bias; a.residual = res; a.ldres = out; a.aux = aux; a.ldaux = out; a.workspace = nullptr; a.colsum = nullptr;
  a.drop = drop; a.defer_fold = nullptr;
  a.Bp = Wp;
  return gemm(a, s);
}

// the same from MX-FP8 images of A and W; mx_q / mx_s: also emit the image of C (BIAS_GELU)
int linear_fwd_mx(const Dims& d, const void* Aq, const void* As, int in, const void* Wq, const void* Ws, int out, void* C,
                  int c_dtype, int epi, const float* bias, const void* res, void* aux, hipStream_t s, const DropCfg& drop,
                  void* mx_q = nullptr, void* mx_s = nullptr) {
  GemmArgs a;
  a.dtype = AVF_BF16; a.transA = 0; a.transB = 1;
  a.M = d.R; a.N = out; a.K = in;
  a.A = Aq; a.lda = in; a.B = Wq; a.ldb = in;
  a.C = C; a.ldc = out; a.c_dtype = c_dtype; a.epilogue = epi;
  a.bias = bias; a.residual = res; a.ldres = out; a.aux = aux; a.ldaux = out; a.workspace = nullptr; a.colsum = nullptr;
  a.drop = drop; a.defer_fold = nullptr;
  return gemm_mx8_nt(a, As, Ws, s, mx_q, mx_s);
}

// dX[R, in] = dY[R, out] * W[out, in].  bf16 mode consumes the transposed copy Wt[in, out] as an NT GEMM.
// colsum (bf16 mode only, optional): column sums of the produced dX, fused in the GEMM epilogue (ws = partials)
int linear_dx(const Dims& d, const void* dY, int out, const void* W_f32, const void* Wt_lo, int in, void* dX, int epi,
              void* aux, hipStream_t s, float* colsum_out = nullptr, void* ws = nullptr,
              const DropCfg& drop = kNoDrop, FoldJob* defer = nullptr, const void* Wtp = nullptr) {  // Wtp: as Wp of linear_fwd
  GemmArgs a;
  a.dtype = d.dt; a.transA = 0;
  a.M = d.R; a.N = in; a.K = out;
  a.A = dY; a.lda = out;
  if (d.dt == AVF_BF16) { a.transB = 1; a.B = Wt_lo; a.ldb = out; }
  else { a.transB = 0; a.B = W_f32; a.ldb = in; }
  a.C = dX; a.ldc = in; a.c_dtype = d.dt; a.epilogue = epi;
  a.bias = nullptr; a.residual = nullptr; a.ldres = 0; a.aux = aux; a.ldaux = in; a.workspace = ws; a.colsum = colsum_out;
  a.drop = drop; a.defer_fold = defer;
  a.Bp = d.dt == AVF_BF16 ? Wtp : nullptr;
  return gemm(a, s);
}

// the same from MX-FP8 images of dY [R, out] and of Wt [in, out] (blocks along out, the reduction); mx_q / mx_s: also emit
// the image of dX (DGELU)
int linear_dx_mx(const Dims& d, const void* dYq, const void* dYs, int out, const void* Wtq, const void* Wts, int in, void* dX,
                 int epi, void* aux, hipStream_t s, float* colsum_out, void* ws, const DropCfg& drop, FoldJob* defer,
                 void* mx_q = nullptr, void* mx_s = nullptr) {
  GemmArgs a;
  a.dtype = AVF_BF16; a.transA = 0; a.transB = 1;
  a.M = d.R; a.N = in; a.K = out;
  a.A = dYq; a.lda = out; a.B = Wtq; a.ldb = out;
  a.C = dX; a.ldc = in; a.c_dtype = AVF_BF16; a.epilogue = epi;
  a.bias = nullptr; a.residual = nullptr; a.ldres = 0; a.aux = aux; a.ldaux = in; a.workspace = ws; a.colsum = colsum_out;
  a.drop = drop; a.defer_fold = defer;
  return gemm_mx8_nt(a, dYs, Wts, s, mx_q, mx_s);
}

// dW[out, in] = dY[R, out]^T * X[R, in]   (fp32 result)
int linear_dw(const Dims& d, const void* dY, int out, const void* X, int in, float* dW, void* ws, hipStream_t s) {
  GemmArgs a;
  a.dtype = d.dt; a.transA = 1; a.transB = 0;
  a.M = out; a.N = in; a.K = d.R;
  a.A = dY; a.lda = out; a.B = X; a.ldb = in;
  a.C = dW; a.ldc = in; a.c_dtype = AVF_F32; a.epilogue = AVF_EPI_NONE;
  a.bias = nullptr; a.residual = nullptr; a.ldres = 0; a.aux = nullptr; a.ldaux = 0; a.workspace = ws; a.colsum = nullptr;
  a.drop = kNoDrop; a.defer_fold = nullptr;
  return gemm(a, s);
}

}  // namespace

size_t gemm_ws(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K) {
  if (dtype == AVF_BF16 && transA == 1 && transB == 0) return gemm_bf16_tn_ws(M, N, K);
  if (dtype == AVF_F32) return gemm_f32_ws(M, N, K);
  return 0;
}

int lowp_ws_images(const avf_layer_cfg* cfg, void* lowp, LowpWs* out) {
  Dims d;
  AVF_TRY(make_dims(cfg, &d));
  LowP l;
  carve_lowp(d, lowp, &l);
  *out = l.ws;
  return 0;
}

int gemm(const GemmArgs& a, hipStream_t s) {
  if (a.dtype == AVF_F32) return gemm_f32(a, s);
  if (a.dtype == AVF_BF16) {
    if (a.transA == 0 && a.transB == 1) return gemm_bf16_nt(a, s);
    if (a.transA == 1 && a.transB == 0) return gemm_bf16_tn(a, s);
    AVF_REQUIRE(false, "gemm(bf16): only NT (transA=0,transB=1) and TN (transA=1,transB=0) forms exist");
  }
  AVF_REQUIRE(false, "gemm: bad dtype %d", a.dtype);
}

}  // namespace avf

using namespace avf;

extern "C" size_t avf_layer_saved_bytes(const avf_layer_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, &d)) return 0;
  return carve_saved(d, nullptr, nullptr);
}
extern "C" size_t avf_layer_lowp_bytes(const avf_layer_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, &d)) return 0;
  return carve_lowp(d, nullptr, nullptr);
}
extern "C" size_t avf_layer_grad_stream_bytes(const avf_layer_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, &d)) return 0;
  return grad_stream_bytes(d.R, d.D, d.mxb, d.gsd);
}
extern "C" size_t avf_layer_workspace_bytes(const avf_layer_cfg* cfg) {
  Dims d;
  if (make_dims(cfg, &d)) return 0;
  return align_up(carve_work(d, nullptr, nullptr), 256);
}

extern "C" int avf_layer_prepare_weights(const avf_layer_cfg* cfg, const avf_layer_params* p, void* lowp,
                                         void* stream) {
  Dims d;
  AVF_TRY(make_dims(cfg, &d));
  if (d.dt != AVF_BF16) return 0;
  AVF_REQUIRE(p && lowp, "prepare_weights: null pointer");
  hipStream_t s = (hipStream_t)stream;
  LowP l;
  carve_lowp(d, lowp, &l);
  PrepBatch b;
  // the query rows of the forward image of Wqkv carry the softmax scale (the transposed image, which backward
  // multiplies dqkv with, does not): q' = h1 (c Wq)^T, so the attention kernels get log2-domain scores from the MFMA
  // ... and the fragment-major images for the weight-stationary GEMM ride in the same launch (the optimizer step rewrites all
  // of them itself: optim.hip): of the row-major images of Wqkv, Wo, W1 and of the transposed images of W2, Wo
  b.d[0] = PrepDesc{p->w_qkv, (bf16*)l.wqkv, (bf16*)l.wqkv_t, 3 * d.I, d.D, attn_q_prescale(d.dh), d.I, l.ws.wqkv_p, nullptr};
  b.d[1] = PrepDesc{p->w_out, (bf16*)l.wo, (bf16*)l.wo_t, d.D, d.I, 1.0f, 0, l.ws.wo_p, l.ws.wot_p};
  b.d[2] = PrepDesc{p->w1, (bf16*)l.w1, (bf16*)l.w1_t, d.M, d.D, 1.0f, 0, l.ws.w1_p, nullptr};
  b.d[3] = PrepDesc{p->w2, (bf16*)l.w2, (bf16*)l.w2_t, d.D, d.M, 1.0f, 0, nullptr, l.ws.w2t_p};
  AVF_TRY(prep_weights_multi(b, 4, s));
  return 0;
}

extern "C" int avf_stack_quant_weights_mx8(const avf_layer_cfg* cfg, int layers, void* const* lowp, void* stream) {
  Dims d;
  AVF_TRY(make_dims(cfg, &d));
  AVF_REQUIRE(d.mx, "stack_quant_weights_mx8: cfg.mx8_fwd is not set");
  AVF_REQUIRE(layers > 0 && layers <= 64 && lowp, "stack_quant_weights_mx8: bad arguments");
  MxQuantJob jobs[64 * 8];
  int n = 0;
  for (int i = 0; i < layers; ++i) {
    AVF_REQUIRE(lowp[i], "stack_quant_weights_mx8: null image buffer (layer %d)", i);
    LowP l;
    carve_lowp(d, lowp[i], &l);
    jobs[n++] = MxQuantJob{l.wqkv, l.wqkv_q, l.wqkv_s, 3 * d.I, d.D};
    jobs[n++] = MxQuantJob{l.w1, l.w1_q, l.w1_s, d.M, d.D};
    jobs[n++] = MxQuantJob{l.w2, l.w2_q, l.w2_s, d.D, d.M};
    if (d.I % 128 == 0) jobs[n++] = MxQuantJob{l.wo, l.wo_q, l.wo_s, d.D, d.I};
    if (d.mxb) {  // transposed images: Wt[in, out], blocks along out
      jobs[n++] = MxQuantJob{l.w2_t, l.w2t_q, l.w2t_s, d.M, d.D};
      jobs[n++] = MxQuantJob{l.w1_t, l.w1t_q, l.w1t_s, d.D, d.M};
      jobs[n++] = MxQuantJob{l.wo_t, l.wot_q, l.wot_s, d.I, d.D};
      if ((3 * d.I) % 128 == 0) jobs[n++] = MxQuantJob{l.wqkv_t, l.wqkvt_q, l.wqkvt_s, d.D, 3 * d.I};
    }
  }
  return quant_mx8_multi(jobs, n, (hipStream_t)stream);
}

extern "C" int avf_layer_fwd(const avf_layer_cfg* cfg, const avf_layer_params* p, const void* lowp, const void* x_in,
                             void* x_out, void* saved, void* workspace, void* stream) {
  Dims d;
  AVF_TRY(make_dims(cfg, &d));
  AVF_REQUIRE(p && x_in && x_out && saved, "layer_fwd: null pointer");
  AVF_REQUIRE(d.dt == AVF_F32 || lowp, "layer_fwd(bf16): lowp weights missing");
  hipStream_t s = (hipStream_t)stream;
  Saved sv;
  carve_saved(d, saved, &sv);
  LowP l;
  carve_lowp(d, (void*)lowp, &l);
  const bool lo = d.dt == AVF_BF16;
  const void* wqkv = lo ? l.wqkv : (const void*)p->w_qkv;
  const void* wo = lo ? l.wo : (const void*)p->w_out;
  const void* w1 = lo ? l.w1 : (const void*)p->w1;
  const void* w2 = lo ? l.w2 : (const void*)p->w2;

  if (d.mx) {
    // config 5: the three GEMMs whose A operand leaves a row-wise producer (LayerNorm, GELU epilogue) run on the MX-FP8
    // matrix path; the producers emit the e4m3 image beside the bf16 tensor backward needs, the weights' images come
    // from avf_stack_quant_weights_mx8.  (mx8_bwd: the dX GEMMs fed by LayerNorm backward / dGELU likewise, avf_layer_bwd.)
    AVF_REQUIRE(workspace, "layer_fwd(mx8): workspace missing");
    Work w;
    carve_work(d, workspace, &w);
    const DropCfg dr0 = make_drop(d.p0, d.seed, d.layer, 0, d.seed_dev), dr1 = make_drop(d.p, d.seed, d.layer, 1, d.seed_dev),
                  dr2 = make_drop(d.p, d.seed, d.layer, 2, d.seed_dev);
    AVF_TRY(layernorm_fwd(x_in, p->ln1_w, p->ln1_b, sv.h1, d.dt, sv.mean1, sv.rstd1, d.R, d.D, cfg->ln_eps, s, w.hq, w.hs, d.xdt));
    AVF_TRY(linear_fwd_mx(d, w.hq, w.hs, d.D, l.wqkv_q, l.wqkv_s, 3 * d.I, sv.qkv, d.dt, AVF_EPI_NONE, nullptr, nullptr,
                          nullptr, s, kNoDrop));
    // out-proj: its A operand is produced per head; the head-resident attention kernel writes the image from its epilogue
    static const int o_mx_on = [] {
      const char* e = tuning_env("AVF_MX8_OUTPROJ");  // tuning / A-B aid: 0 = out-projection on bf16 operands
      return (e && *e) ? atoi(e) : 1;
    }();
    const bool o_mx = o_mx_on && !d.keep && attn_fwd_emits_mx8(d.N, d.dh) && d.I % 128 == 0;
    if (d.keep)
      AVF_TRY(attn_fwd_vec(AVF_BF16, sv.qkv, sv.o, sv.lse2, d.B, d.N, d.H, d.dh, s, d.keep, attn_q_prescale_on()));
    else
      AVF_TRY(attn_fwd_bf16((const bf16*)sv.qkv, (bf16*)sv.o, sv.lse2, d.B, d.N, d.H, d.dh, s, attn_q_prescale_on(),
                            o_mx ? w.oq : nullptr, o_mx ? w.os : nullptr));
    if (o_mx)
      AVF_TRY(linear_fwd_mx(d, w.oq, w.os, d.I, l.wo_q, l.wo_s, d.D, sv.x_mid, d.xdt, AVF_EPI_BIAS_RES, p->b_out, x_in, nullptr,
                            s, dr0));
    else
      AVF_TRY(linear_fwd(d, sv.o, d.I, wo, d.D, sv.x_mid, d.xdt, AVF_EPI_BIAS_RES, p->b_out, x_in, nullptr, s, dr0, l.ws.wo_p));
    AVF_TRY(layernorm_fwd(sv.x_mid, p->ln2_w, p->ln2_b, sv.h2, d.dt, sv.mean2, sv.rstd2, d.R, d.D, cfg->ln_eps, s, w.hq, w.hs, d.xdt));
    AVF_TRY(linear_fwd_mx(d, w.hq, w.hs, d.D, l.w1_q, l.w1_s, d.M, sv.g, d.dt, AVF_EPI_BIAS_GELU, p->b1, nullptr, sv.u, s, dr1,
                          w.gq, w.gs));
    AVF_TRY(linear_fwd_mx(d, w.gq, w.gs, d.M, l.w2_q, l.w2_s, d.D, x_out, d.xdt, AVF_EPI_BIAS_RES, p->b2, sv.x_mid, nullptr,
                          s, dr2));
    return 0;
  }
  if (!d.rs16 && !d.keep && small_layer_ok(d.dt, d.N, d.D, d.H, d.dh, d.M)) {  // short sequences: the whole layer in one launch
    const float sc = attn_q_prescale_on() ? 1.0f : 1.4426950408889634f / sqrtf((float)d.dh);
    return layer_fwd_small(d.B, d.N, d.D, d.H, d.M, cfg->ln_eps, sc, p, wqkv, wo, w1, w2, (const float*)x_in, (float*)x_out, sv.h1,
                           sv.mean1, sv.rstd1, sv.qkv, sv.o, sv.lse2, (float*)sv.x_mid, sv.h2, sv.mean2, sv.rstd2, sv.u, sv.g,
                           make_drop(d.p0, d.seed, d.layer, 0, d.seed_dev), make_drop(d.p, d.seed, d.layer, 1, d.seed_dev),
                           make_drop(d.p, d.seed, d.layer, 2, d.seed_dev), s);
  }
  AVF_TRY(layernorm_fwd(x_in, p->ln1_w, p->ln1_b, sv.h1, d.dt, sv.mean1, sv.rstd1, d.R, d.D, cfg->ln_eps, s, nullptr, nullptr, d.xdt));
  AVF_TRY(linear_fwd(d, sv.h1, d.D, wqkv, 3 * d.I, sv.qkv, d.dt, AVF_EPI_NONE, nullptr, nullptr, nullptr, s, kNoDrop, l.ws.wqkv_p));
  // token mask (heads.py:225-232): on the MFMA kernels where they carry it (bf16, dim_head 64, up to 512 tokens), else on
  // the fp32-arithmetic ones
  const bool mask_mfma = d.keep && lo && attn_masked_bf16_ok(d.N, d.dh, attn_q_prescale_on());
  if (mask_mfma)
    AVF_TRY(attn_fwd_bf16((const bf16*)sv.qkv, (bf16*)sv.o, sv.lse2, d.B, d.N, d.H, d.dh, s, true, nullptr, nullptr, d.keep));
  else if (d.keep) AVF_TRY(attn_fwd_vec(d.dt, sv.qkv, sv.o, sv.lse2, d.B, d.N, d.H, d.dh, s, d.keep, lo && attn_q_prescale_on()));
  else if (lo) AVF_TRY(attn_fwd_bf16((const bf16*)sv.qkv, (bf16*)sv.o, sv.lse2, d.B, d.N, d.H, d.dh, s, attn_q_prescale_on()));
  else AVF_TRY(attn_fwd_f32((const float*)sv.qkv, (float*)sv.o, sv.lse2, d.B, d.N, d.H, d.dh, s));
  const DropCfg dr0 = make_drop(d.p0, d.seed, d.layer, 0, d.seed_dev), dr1 = make_drop(d.p, d.seed, d.layer, 1, d.seed_dev),
                dr2 = make_drop(d.p, d.seed, d.layer, 2, d.seed_dev);
  AVF_TRY(linear_fwd(d, sv.o, d.I, wo, d.D, sv.x_mid, d.xdt, AVF_EPI_BIAS_RES, p->b_out, x_in, nullptr, s, dr0, l.ws.wo_p));
  AVF_TRY(layernorm_fwd(sv.x_mid, p->ln2_w, p->ln2_b, sv.h2, d.dt, sv.mean2, sv.rstd2, d.R, d.D, cfg->ln_eps, s, nullptr, nullptr, d.xdt));
  AVF_TRY(linear_fwd(d, sv.h2, d.D, w1, d.M, sv.g, d.dt, AVF_EPI_BIAS_GELU, p->b1, nullptr, sv.u, s, dr1, l.ws.w1_p));
  AVF_TRY(linear_fwd(d, sv.g, d.M, w2, d.D, x_out, d.xdt, AVF_EPI_BIAS_RES, p->b2, sv.x_mid, nullptr, s, dr2));
  return 0;
}

extern "C" int avf_layer_bwd(const avf_layer_cfg* cfg, const avf_layer_params* p, const void* lowp, const void* x_in,
                             const void* saved, const float* dx_out, const void* dx_out_lo,
                             const float* dx_out_colsum, float* dx_in, void* dx_in_lo, float* dx_in_colsum,
                             const avf_layer_grads* g, void* workspace, void* stream) {
  Dims d;
  AVF_TRY(make_dims(cfg, &d));
  AVF_REQUIRE(p && x_in && saved && g && workspace, "layer_bwd: null pointer");
  // bf16 gradient stream: the incoming gradient may come as its bf16 image alone, and the fp32 dx_in is optional (a caller
  // asks for it only where it consumes it, e.g. below the bottom layer)
  AVF_REQUIRE(d.gs16 ? (dx_out || dx_out_lo) && (dx_in || dx_in_lo) : (dx_out && dx_in), "layer_bwd: null gradient pointer");
  AVF_REQUIRE(!d.gs16 || dx_in_lo, "layer_bwd(grad_stream_bf16): dx_in_lo missing");
  AVF_REQUIRE(d.dt == AVF_F32 || lowp, "layer_bwd(bf16): lowp weights missing");
  hipStream_t s = (hipStream_t)stream;
  Saved sv;
  carve_saved(d, (void*)saved, &sv);
  LowP l;
  carve_lowp(d, (void*)lowp, &l);
  const bool lo = d.dt == AVF_BF16;
  Work w;
  carve_work(d, workspace, &w);

  // gradient of the layer output in the compute dtype (GEMM operand)
  // dropout: the Linears behind a dropout site see the masked, rescaled gradient (the residual stream does not)
  const DropCfg dr0 = make_drop(d.p0, d.seed, d.layer, 0, d.seed_dev), dr1 = make_drop(d.p, d.seed, d.layer, 1, d.seed_dev),
                dr2 = make_drop(d.p, d.seed, d.layer, 2, d.seed_dev);
  const DropCfg dr_prev2 = d.layer > 0 ? make_drop(d.p, d.seed, d.layer - 1, 2, d.seed_dev) : kNoDrop;
  // short sequences (the reference's 12-token stacks): the six dependent launches around the attention backward run as two
  // fused kernels (layer_small.hip), which also make the bf16 image of the incoming gradient when the caller gave none
  static const int small_bwd_on = [] {
    const char* e = tuning_env("AVF_LAYER_SMALL_BWD");  // tuning / A-B aid
    return (e && *e) ? atoi(e) : 1;
  }();
  const bool small_bwd = lo && !d.rs16 && !d.mx && !d.keep && small_bwd_on && small_layer_ok(d.dt, d.N, d.D, d.H, d.dh, d.M);
  const void* gy = dx_out;
  const void* gy_stream = nullptr;  // gsd: the unmasked bf16 stream (LayerNorm-2 backward's residual gradient); gy is the masked image
  bool own_copy = false;
  if (lo) {
    if (dx_out_lo && d.gsd) {  // [stream | masked by this layer's site-2 mask] - written by the layer above's LayerNorm-1 backward
      gy_stream = dx_out_lo;
      gy = (const char*)dx_out_lo + grad_q_off(d.R, d.D);
    } else if (dx_out_lo) gy = dx_out_lo;  // the caller's previous call already applied this layer's site-2 mask
    else if (small_bwd) gy = w.dx_out_lo;  // written by the fused kernel below
    else {
      AVF_TRY(cast_f32_to_bf16(dx_out, w.dx_out_lo, d.R * d.D, s, dr2));
      gy = w.dx_out_lo;
      own_copy = true;
      if (d.gsd) {  // top of the stack: the stream image beside the masked one
        AVF_TRY(cast_f32_to_bf16(dx_out, w.dx_out_s, d.R * d.D, s, kNoDrop));
        gy_stream = w.dx_out_s;
      }
    }
  }
  AVF_REQUIRE(!d.gsd || !small_bwd, "layer_bwd: grad_stream_bf16 with dropout on a short-sequence layer (internal error)");
  const bool gm_masked = d.gsd && d.p0 > 0.f;  // a separate masked image of dx_mid exists (site 0 behind a real to_out)
  const bool f32_drop = !lo && d.p > 0.f;
  if (f32_drop) {  // fp32 mode with live dropout: net.3 sees the gradient through its site-2 mask
    AVF_TRY(mask_copy_f32(dx_out, w.gy_m, d.R * d.D, s, dr2));
    gy = w.gy_m;
    own_copy = true;
  }
  const bool f32_drop0 = f32_drop && d.p0 > 0.f;  // (no site 0 behind an nn.Identity to_out)
  const void* gm = lo ? (gm_masked ? (const void*)w.dx_mid_m : (const void*)w.dx_mid_lo)
                      : (f32_drop0 ? (const void*)w.gm_m : (const void*)w.dx_mid);
  // bf16 mode: the four dW GEMMs run as ONE grouped launch at the end of the layer (their operands all stay
  // alive in the workspace), when the shapes allow the LDS-DMA kernel
  TnGroupArgs grp = dw_group(d, gy, sv.g, w.du, sv.h2, gm, sv.o, w.dqkv, sv.h1, g);
  grp.workspace = w.gemm_ws;
  // parity mode, bf16x3 arithmetic (round 6): the four fp32 weight gradients likewise as one launch + one fold, issued right after
  // the attention backward - the last point where every operand exists and dx_out (which dx_in may alias) is still intact
  const bool grouped32 = !lo && !small_bwd && gemm_f32x3_tn_group_ok(grp);
  const bool grouped = lo && gemm_bf16_tn_group_ok(grp);   // (also defers the column folds of the layer to the grouped fold)
  FoldList folds;
  memset(&folds, 0, sizeof(folds));
  folds.count = 3;

  // the grouped dW launch and its fold are shared with the general path
  if (small_bwd) {
    float* pb1 = w.small_part;
    float* pln2 = pb1 + (size_t)d.B * d.M;
    float* pln1 = pln2 + (size_t)d.B * 3 * d.D;
    SmallBwdAHost ha;
    ha.dx_out = dx_out; ha.dx_out_lo = dx_out_lo; ha.gy_store = dx_out_lo ? nullptr : w.dx_out_lo; ha.x_mid = (const float*)sv.x_mid;
    ha.u = sv.u; ha.ln2_w = p->ln2_w; ha.mean2 = sv.mean2; ha.rstd2 = sv.rstd2;
    ha.w2_t = l.w2_t; ha.w1_t = l.w1_t; ha.wo_t = l.wo_t; ha.du = w.du; ha.dx_mid = d.gs16 ? nullptr : w.dx_mid;
    ha.dx_mid_lo = w.dx_mid_lo; ha.d_o = w.d_o; ha.pb1 = pb1; ha.pln2 = pln2; ha.gs16 = d.gs16 ? 1 : 0;
    ha.dr0 = dr0; ha.dr1 = dr1; ha.dr2 = dr2;
    AVF_TRY(layer_bwd_small_a(d.B, d.N, d.D, d.I, d.M, ha, s));
    // db2 (as below): handed over, or the column sums of the (masked) gradient image kernel A has just written / read
    if (dx_out_colsum && dx_out_colsum != g->b2)
      AVF_REQUIRE(hipMemcpyAsync(g->b2, dx_out_colsum, (size_t)d.D * 4, hipMemcpyDeviceToDevice, s) == hipSuccess,
                  "layer_bwd: memcpy failed");
    else if (!dx_out_colsum) {
      if (d.p > 0.f || !dx_out) AVF_TRY(colsum(gy, AVF_BF16, d.R, d.D, d.D, g->b2, w.cs_ws, s));
      else AVF_TRY(colsum(dx_out, AVF_F32, d.R, d.D, d.D, g->b2, w.cs_ws, s));
    }
    static const int small_att_on = [] {
      const char* e = tuning_env("AVF_LAYER_SMALL_ATT");  // tuning / A-B aid: 0 = per-operator attention backward
      return (e && *e) ? atoi(e) : 1;
    }();
    const bool fuse_att = small_att_on && d.H <= 16;
    if (!fuse_att)
      AVF_TRY(attn_bwd_bf16((const bf16*)sv.qkv, (const bf16*)sv.o, (const bf16*)w.d_o, sv.lse2, (bf16*)w.dqkv, w.delta,
                            d.B, d.N, d.H, d.dh, s, attn_q_prescale_on(), w.delta + (size_t)d.B * d.H * d.N));
    SmallBwdBHost hb;
    hb.attention = fuse_att ? 1 : 0;
    hb.qkv = sv.qkv; hb.o = sv.o; hb.d_o = w.d_o; hb.lse2 = sv.lse2; hb.H = d.H;
    {  // P = exp2(s c - lse2); dq = scale dS k; dk = scale dS^T q = dS^T q' / log2(e) when q' = q log2(e) scale (attn_bf16.hip)
      const float scale = 1.0f / sqrtf((float)d.dh), log2e = 1.4426950408889634f;
      hb.score_scale = attn_q_prescale_on() ? 1.0f : log2e * scale;
      hb.dq_scale = scale;
      hb.dk_scale = attn_q_prescale_on() ? 1.0f / log2e : scale;
    }
    hb.dqkv = w.dqkv; hb.wqkv_t = l.wqkv_t; hb.x_in = (const float*)x_in; hb.ln1_w = p->ln1_w; hb.mean1 = sv.mean1; hb.rstd1 = sv.rstd1;
    hb.dx_mid = d.gs16 ? nullptr : w.dx_mid; hb.dx_mid_lo = w.dx_mid_lo; hb.dx_in = dx_in; hb.dx_in_lo = dx_in_lo; hb.pln1 = pln1;
    hb.gs16 = d.gs16 ? 1 : 0; hb.dr_prev2 = dr_prev2;
    AVF_TRY(layer_bwd_small_b(d.B, d.N, d.D, d.I, hb, s));
    folds.job[0] = FoldJob{pb1, d.B, d.M, d.M, g->b1, nullptr, nullptr};
    folds.job[1] = FoldJob{pln2, d.B, 3 * d.D, d.D, g->ln2_w, g->ln2_b, g->b_out};
    folds.job[2] = FoldJob{pln1, d.B, 3 * d.D, d.D, g->ln1_w, g->ln1_b, dx_in_colsum};
    if (grouped) return gemm_bf16_tn_group(grp, s, &folds);
    AVF_TRY(linear_dw(d, gy, d.D, sv.g, d.M, g->w2, w.gemm_ws, s));
    AVF_TRY(linear_dw(d, w.du, d.M, sv.h2, d.D, g->w1, w.gemm_ws, s));
    AVF_TRY(linear_dw(d, gm, d.D, sv.o, d.I, g->w_out, w.gemm_ws, s));
    AVF_TRY(linear_dw(d, w.dqkv, 3 * d.I, sv.h1, d.D, g->w_qkv, w.gemm_ws, s));
    for (int j = 0; j < 3; ++j) AVF_TRY(fold_job(folds.job[j], s));
    return 0;
  }

  // ---- feed-forward half -------------------------------------------------------------------
  // db2 = column sums of dx_out: handed over by the caller (the next layer's LN1 backward produced them) or summed here
  // (the caller normally points the previous call's dx_in_colsum straight at this layer's b2 slot: nothing to do)
  if (dx_out_colsum && dx_out_colsum != g->b2)
    AVF_REQUIRE(hipMemcpyAsync(g->b2, dx_out_colsum, (size_t)d.D * 4, hipMemcpyDeviceToDevice, s) == hipSuccess,
                "layer_bwd: memcpy failed");
  else if (!dx_out_colsum) {
    if (d.p > 0.f) {  // db2 sums the MASKED gradient
      AVF_REQUIRE(own_copy, "layer_bwd: with dropout pass dx_out_colsum together with dx_out_lo");
      AVF_TRY(colsum(gy, d.dt, d.R, d.D, d.D, g->b2, w.cs_ws, s));
    } else if (dx_out) {
      AVF_TRY(colsum(dx_out, AVF_F32, d.R, d.D, d.D, g->b2, w.cs_ws, s));
    } else {
      AVF_TRY(colsum(gy, AVF_BF16, d.R, d.D, d.D, g->b2, w.cs_ws, s));
    }
  }
  if (!grouped && !grouped32) AVF_TRY(linear_dw(d, gy, d.D, sv.g, d.M, g->w2, w.gemm_ws, s));
  // config 5, backward half: the three dX GEMMs whose A operand leaves a row-wise producer (LayerNorm backward of this /
  // the next layer, the dGELU epilogue) read MX-FP8 images written by that producer; dqkv -> dh1 (A produced per head by the
  // attention backward) and the four weight-gradient GEMMs (reduction over tokens, not features) stay bf16
  const void *gyq = nullptr, *gys = nullptr;
  void *inq = nullptr, *ins = nullptr;  // image of dx_in, behind the bf16 one in the caller's buffer
  if (d.mxb) {
    if (d.gy_mx && dx_out_lo) {
      gyq = (const char*)dx_out_lo + grad_q_off(d.R, d.D);
      gys = (const char*)dx_out_lo + grad_s_off(d.R, d.D);
    } else {  // top of the stack (or a caller without the image): one quantiser pass over the bf16 gradient
      AVF_TRY(quant_mx8(gy, AVF_BF16, d.D, d.R, d.D, w.gyq, d.D, w.gys, s));
      gyq = w.gyq; gys = w.gys;
    }
    if (dx_in_lo) {
      inq = (char*)dx_in_lo + grad_q_off(d.R, d.D);
      ins = (char*)dx_in_lo + grad_s_off(d.R, d.D);
    }
  }
  if (d.mxb) {
    // du = (gy W2) * gelu'(u) with the image of du for the fp8 GEMM behind it.  Where the weight-stationary bf16 kernel takes
    // the shape it is the faster form of THIS launch even in the fp8 mode (C5: 49 us against 70 on MX-FP8 operands - the launch
    // is bound by the 168 MB it reads and writes beside the operands, which fp8 does not touch); it emits the same image
    GemmArgs a;
    a.dtype = AVF_BF16; a.transA = 0; a.transB = 1;
    a.M = d.R; a.N = d.M; a.K = d.D;
    a.A = gy; a.lda = d.D; a.B = l.w2_t; a.ldb = d.D;
    a.C = w.du; a.ldc = d.M; a.c_dtype = AVF_BF16; a.epilogue = AVF_EPI_DGELU;
    a.bias = nullptr; a.residual = nullptr; a.ldres = 0; a.aux = (void*)sv.u; a.ldaux = d.M; a.workspace = w.cs_ws; a.colsum = g->b1;
    a.drop = dr1; a.defer_fold = grouped ? &folds.job[0] : nullptr;
    a.Bp = l.ws.w2t_p; a.mx_q = w.duq; a.mx_s = w.dus;
    static const int dgelu_ws = [] {
      const char* e = tuning_env("AVF_MX8_DGELU_WS");  // A/B aid: 0 = the dGELU GEMM of the fp8 mode on MX-FP8 operands
      return (e && *e) ? atoi(e) : 1;
    }();
    if (dgelu_ws && gemm_bf16_nt_ws_ok(a)) AVF_TRY(gemm(a, s));
    else
      AVF_TRY(linear_dx_mx(d, gyq, gys, d.D, l.w2t_q, l.w2t_s, d.M, w.du, AVF_EPI_DGELU, sv.u, s, g->b1, w.cs_ws, dr1,
                           grouped ? &folds.job[0] : nullptr, w.duq, w.dus));
  } else if (lo) {  // db1 = colsum(du) fused into the dGELU GEMM epilogue
    AVF_TRY(linear_dx(d, gy, d.D, p->w2, l.w2_t, d.M, w.du, AVF_EPI_DGELU, sv.u, s, g->b1, w.cs_ws, dr1,
                      grouped ? &folds.job[0] : nullptr, l.ws.w2t_p));
  } else {
    AVF_TRY(linear_dx(d, gy, d.D, p->w2, l.w2_t, d.M, w.du, AVF_EPI_DGELU, sv.u, s, nullptr, nullptr, dr1));
    AVF_TRY(colsum(w.du, d.dt, d.R, d.M, d.M, g->b1, w.cs_ws, s));
  }
  if (d.mxb)
    AVF_TRY(linear_dx_mx(d, w.duq, w.dus, d.M, l.w1t_q, l.w1t_s, d.D, w.dh, AVF_EPI_NONE, nullptr, s, nullptr, nullptr, kNoDrop,
                         nullptr));
  else
    AVF_TRY(linear_dx(d, w.du, d.M, p->w1, l.w1_t, d.D, w.dh, AVF_EPI_NONE, nullptr, s));
  if (d.gs16)  // residual gradient in: the bf16 image the GEMMs read (gsd: the unmasked stream); out: the bf16 dx_mid only
    AVF_TRY(layernorm_bwd(w.dh, d.dt, sv.x_mid, p->ln2_w, sv.mean2, sv.rstd2, d.gsd ? gy_stream : gy, nullptr, w.dx_mid_lo,
                          g->ln2_w, g->ln2_b, g->b_out, w.ln_ws, d.R, d.D, s, dr0, grouped ? &folds.job[1] : nullptr, AVF_BF16,
                          d.xdt, w.mq, w.ms, gm_masked ? w.dx_mid_m : nullptr));
  else
    AVF_TRY(layernorm_bwd(w.dh, d.dt, sv.x_mid, p->ln2_w, sv.mean2, sv.rstd2, dx_out, w.dx_mid,
                          lo ? w.dx_mid_lo : nullptr, g->ln2_w, g->ln2_b, g->b_out, w.ln_ws, d.R, d.D, s, dr0,
                          (grouped || grouped32) ? &folds.job[1] : nullptr, AVF_F32, d.xdt, w.mq, w.ms));
  if (!grouped && !grouped32) AVF_TRY(linear_dw(d, w.du, d.M, sv.h2, d.D, g->w1, w.gemm_ws, s));
  // ---- attention half ----------------------------------------------------------------------
  if (f32_drop0) AVF_TRY(mask_copy_f32(w.dx_mid, w.gm_m, d.R * d.D, s, dr0));  // to_out sees dx_mid through its site-0 mask
  if (!grouped && !grouped32) AVF_TRY(linear_dw(d, gm, d.D, sv.o, d.I, g->w_out, w.gemm_ws, s));
  if (d.mxb)
    AVF_TRY(linear_dx_mx(d, w.mq, w.ms, d.D, l.wot_q, l.wot_s, d.I, w.d_o, AVF_EPI_NONE, nullptr, s, nullptr, nullptr, kNoDrop,
                         nullptr));
  else
    AVF_TRY(linear_dx(d, gm, d.D, p->w_out, l.wo_t, d.I, w.d_o, AVF_EPI_NONE, nullptr, s, nullptr, nullptr, kNoDrop, nullptr,
                      l.ws.wot_p));
  // dqkv -> dh1 on MX-FP8 operands, the image of dqkv written by the merged attention backward (DESIGN_HISTORY.md section 17, item 5):
  // built and bit-exact, but the image costs the attention kernel 18 us at B = 64, N = 512 (50 us before its stores were
  // made 16 bytes wide and dQ's 32-blocks wave-local) while the K = 1536 GEMM, already at 0.87 PFLOP/s on bf16 operands,
  // gains ~9 us: C5 5.06 ms per step with it against 4.93 without - OFF unless AVF_MX8_DQKV=1.
  const char* dq_env = tuning_env("AVF_MX8_DQKV");  // (read per call: a test flips it inside one process)
  const int dq_mx_on = (dq_env && *dq_env) ? atoi(dq_env) : 0;
  const bool dq_mx = d.mxb && dq_mx_on && !d.keep && (3 * d.I) % 128 == 0 && d.D % 128 == 0 &&
                     attn_bwd_emits_mx8(d.N, d.dh, attn_q_prescale_on());
  if (d.keep && lo && !d.mx && attn_masked_bf16_ok(d.N, d.dh, attn_q_prescale_on()))  // (as the forward chose)
    AVF_TRY(attn_bwd_bf16((const bf16*)sv.qkv, (const bf16*)sv.o, (const bf16*)w.d_o, sv.lse2, (bf16*)w.dqkv, w.delta,
                          d.B, d.N, d.H, d.dh, s, true, w.delta + (size_t)d.B * d.H * d.N, d.keep));
  else if (d.keep)
    AVF_TRY(attn_bwd_vec(d.dt, sv.qkv, sv.o, w.d_o, sv.lse2, w.dqkv, w.delta, d.B, d.N, d.H, d.dh, s, d.keep,
                         lo && attn_q_prescale_on()));
  else if (lo)
    AVF_TRY(attn_bwd_bf16((const bf16*)sv.qkv, (const bf16*)sv.o, (const bf16*)w.d_o, sv.lse2, (bf16*)w.dqkv, w.delta,
                          d.B, d.N, d.H, d.dh, s, attn_q_prescale_on(), w.delta + (size_t)d.B * d.H * d.N, nullptr,
                          dq_mx ? w.dqq : nullptr, dq_mx ? w.dqs : nullptr));
  else
    AVF_TRY(attn_bwd_f32((const float*)sv.qkv, (const float*)sv.o, (const float*)w.d_o, sv.lse2, (float*)w.dqkv,
                         w.delta, d.B, d.N, d.H, d.dh, s));
  if (grouped32) AVF_TRY(gemm_f32x3_tn_group(grp, s));
  if (dq_mx)  // dh1 = dqkv Wqkv on MX-FP8 operands: the image of dqkv left the attention backward's epilogue
    AVF_TRY(linear_dx_mx(d, w.dqq, w.dqs, 3 * d.I, l.wqkvt_q, l.wqkvt_s, d.D, w.dh, AVF_EPI_NONE, nullptr, s, nullptr, nullptr,
                         kNoDrop, nullptr));
  else
    AVF_TRY(linear_dx(d, w.dqkv, 3 * d.I, p->w_qkv, l.wqkv_t, d.D, w.dh, AVF_EPI_NONE, nullptr, s));
  // dx_in may alias dx_out, which the grouped dW2 GEMM does not read (it uses the bf16 copy gy)
  if (d.gs16)  // (gsd: the masked image for the layer below goes behind the stream in dx_in_lo; layer 0 has no site below it)
    AVF_TRY(layernorm_bwd(w.dh, d.dt, x_in, p->ln1_w, sv.mean1, sv.rstd1, w.dx_mid_lo, dx_in, dx_in_lo, g->ln1_w, g->ln1_b,
                          dx_in_colsum, w.ln_ws1, d.R, d.D, s, dr_prev2, grouped ? &folds.job[2] : nullptr, AVF_BF16, d.xdt,
                          inq, ins, (d.gsd && dr_prev2.thresh16) ? (char*)dx_in_lo + grad_q_off(d.R, d.D) : nullptr));
  else
    AVF_TRY(layernorm_bwd(w.dh, d.dt, x_in, p->ln1_w, sv.mean1, sv.rstd1, w.dx_mid, dx_in, lo ? dx_in_lo : nullptr,
                          g->ln1_w, g->ln1_b, dx_in_colsum, w.ln_ws1, d.R, d.D, s, dr_prev2,
                          (grouped || grouped32) ? &folds.job[2] : nullptr, AVF_F32, d.xdt, inq, ins));
  if (!grouped && !grouped32) AVF_TRY(linear_dw(d, w.dqkv, 3 * d.I, sv.h1, d.D, g->w_qkv, w.gemm_ws, s));
  // one launch folds the split-K slabs of the four weight gradients and the three deferred column folds
  // (db1; dgamma2/dbeta2/dbo; dgamma1/dbeta1/previous layer's db2)
  if (grouped) {
    AVF_TRY(gemm_bf16_tn_group(grp, s, &folds));
  }
  if (grouped32) AVF_TRY(fold_list(folds, s));  // the two LayerNorm column folds of the layer in one launch (job 0 stays empty)
  return 0;
}
