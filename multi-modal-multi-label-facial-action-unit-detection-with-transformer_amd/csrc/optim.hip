// optim.hip - Adam update of one transformer layer fused with the refresh of its bf16 weight copies.
//
// Reference: the training loop steps torch.optim.Adam(lr, weight_decay) (train.py:318-322, L2 weight decay added to
// the gradient, bias-corrected, eps outside the square root).  In throughput mode the stack then needs bf16 copies W
// and W^T of the four weight matrices (avf_layer_prepare_weights) - one more read of every master weight per step.
// Here ONE launch per layer reads p, g, m, v once and writes p, m, v AND the two bf16 images (the transpose through a
// 64x64 LDS tile), with 16-byte loads and 8-byte bf16 stores.  The arithmetic follows torch's fused kernel
// (exp_avg by lerp, exp_avg_sq by the two-term form) so the two optimizers agree to rounding.
#include "common.hpp"

namespace avf {

namespace {

// streaming accesses of the optimizer state (touched once per step): non-temporal, so they do not evict the bf16 weight
// images and activations the next forward re-reads from L2 / Infinity Cache
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
  const f32x4_nt t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(p));
}


struct AdamDesc {
  float* p;
  const float* g;  // null: no update, the bf16 images are still refreshed
  float* m;
  float* v;
  bf16* lo;  // row-major bf16 copy [R, C] (nullable)
  bf16* t;   // transposed bf16 copy [C, R] (nullable)
  bf16* lo_p;  // fragment-major image of `lo` (C == 512; common.hpp pack_ws_off) (nullable)
  bf16* t_p;   // fragment-major image of `t` (R == 512) (nullable)
  int R, C;
  float lo_scale;      // rows < lo_scaled_rows of `lo` (only) are written multiplied by this: the query rows of Wqkv carry
  int lo_scaled_rows;  // the softmax scale (attn_q_prescale)
  int tile0, tiles_c;  // first work item of this tensor, tiles per row of tiles (matrices)
};

// Descriptors per launch.  The table travels as a kernel argument (12.6 KB: tools/diag/kernarg_probe.hip - this stack takes it);
// rounds 1 - 4 kept it under 4 KB (33 = three layers), which made every small stack of the reference's real model its own 13 us
// launch.  13 layers per launch; avf_adam_batch_begin / _end collect the stacks and loose tensors of one optimizer step.
constexpr int ADAM_MAX = 143;

struct AdamBatch {
  AdamDesc d[ADAM_MAX];
  int count;
  float lr, b1, b2, eps, wd;
  const float* step;  // device scalar: the step number of THIS update (>= 1)
};

struct AdamCoef {
  float lr_c, rsq_c2, b1, b2, eps, wd;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamCoef& k) {
  g = fmaf(k.wd, p, g);
  const float w = 1.0f - k.b1;  // lerp(m, g, w) as torch: a + w (b - a) for w < 0.5
  m = w < 0.5f ? fmaf(w, g - m, m) : g - (g - m) * (1.0f - w);
  v = k.b2 * v + (1.0f - k.b2) * g * g;
  const float denom = sqrtf(v) * k.rsq_c2 + k.eps;
  p -= k.lr_c * (m / denom);
}

__global__ __launch_bounds__(256) void adam_layer_kernel(AdamBatch b) {
  __shared__ float tile[64][65];
  int di = 0;
  {  // the last descriptor whose first work item is <= this one (tile0 ascends): binary search over up to ADAM_MAX entries
    int hi = b.count - 1;
#pragma unroll 1
    while (di < hi) {
      const int mid = (di + hi + 1) >> 1;
      if ((int)blockIdx.x >= b.d[mid].tile0) di = mid;
      else hi = mid - 1;
    }
  }
  const AdamDesc d = b.d[di];
  const int item = (int)blockIdx.x - d.tile0;
  const float step = b.step ? b.step[0] : 1.0f;
  AdamCoef k;
  k.b1 = b.b1; k.b2 = b.b2; k.eps = b.eps; k.wd = b.wd;
  k.lr_c = b.lr / (1.0f - powf(b.b1, step));
  k.rsq_c2 = 1.0f / sqrtf(1.0f - powf(b.b2, step));
  const bool upd = d.g != nullptr;

  if (d.R == 1) {  // vector: 4096 elements per work item
    const int base = item * 4096;
    for (int i = threadIdx.x; i < 4096 && base + i < d.C; i += 256) {
      if (!upd) continue;
      float p = d.p[base + i], m = d.m[base + i], v = d.v[base + i];
      adam1(p, d.g[base + i], m, v, k);
      d.p[base + i] = p; d.m[base + i] = m; d.v[base + i] = v;
    }
    return;
  }
  const int r0 = (item / d.tiles_c) * 64, c0 = (item % d.tiles_c) * 64;
  const bool vec = (d.C & 3) == 0 && (d.R & 3) == 0;
  if (vec) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = c0 + tx * 4;
    // all sixteen 16-byte loads of the thread are issued before the first store (the pointers may alias as far as the
    // compiler knows, so it would not hoist them itself)
    float4 P[4], G[4], Mv[4], Vv[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 16 * i;
      ok[i] = r < d.R && c < d.C;
      const int64_t o = (int64_t)r * d.C + c;
      P[i] = G[i] = Mv[i] = Vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok[i]) {
        P[i] = nt_load4(d.p + o);
        if (upd) {
          G[i] = nt_load4(d.g + o);
          Mv[i] = nt_load4(d.m + o);
          Vv[i] = nt_load4(d.v + o);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 16 * i;
      float4 p = P[i];
      if (ok[i]) {
        const int64_t o = (int64_t)r * d.C + c;
        if (upd) {
          float4 m = Mv[i], v = Vv[i];
          const float4 g = G[i];
          adam1(p.x, g.x, m.x, v.x, k); adam1(p.y, g.y, m.y, v.y, k);
          adam1(p.z, g.z, m.z, v.z, k); adam1(p.w, g.w, m.w, v.w, k);
          nt_store4(d.p + o, p);
          nt_store4(d.m + o, m);
          nt_store4(d.v + o, v);
        }
        if (d.lo) {
          const float sc = r < d.lo_scaled_rows ? d.lo_scale : 1.0f;
          store4<bf16>(d.lo + o, make_float4(p.x * sc, p.y * sc, p.z * sc, p.w * sc));
          if (d.lo_p)  // the same values in the fragment-major image (8 bytes of a lane's 16-byte operand piece)
            store4<bf16>(reinterpret_cast<bf16*>(reinterpret_cast<char*>(d.lo_p) + pack_ws_off(r, c)),
                         make_float4(p.x * sc, p.y * sc, p.z * sc, p.w * sc));
        }
      }
      tile[ty + 16 * i][tx * 4 + 0] = p.x; tile[ty + 16 * i][tx * 4 + 1] = p.y;
      tile[ty + 16 * i][tx * 4 + 2] = p.z; tile[ty + 16 * i][tx * 4 + 3] = p.w;
    }
    if (!d.t) return;
    __syncthreads();
    const int cc = threadIdx.x >> 2, rg = (threadIdx.x & 3) * 16;  // output row c0+cc, 16 consecutive output columns
    if (c0 + cc < d.C) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = rg + 4 * q;
        if (r0 + rr < d.R) {
          const float4 tv = make_float4(tile[rr][cc], tile[rr + 1][cc], tile[rr + 2][cc], tile[rr + 3][cc]);
          store4<bf16>(d.t + (int64_t)(c0 + cc) * d.R + r0 + rr, tv);
          if (d.t_p) store4<bf16>(reinterpret_cast<bf16*>(reinterpret_cast<char*>(d.t_p) + pack_ws_off(c0 + cc, r0 + rr)), tv);
        }
      }
    }
    return;
  }
  // generic shapes: scalar accesses
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = r0 + (e >> 6), c = c0 + (e & 63);
    float p = 0.f;
    if (r < d.R && c < d.C) {
      const int64_t o = (int64_t)r * d.C + c;
      p = d.p[o];
      if (upd) {
        float m = d.m[o], v = d.v[o];
        adam1(p, d.g[o], m, v, k);
        d.p[o] = p; d.m[o] = m; d.v[o] = v;
      }
      if (d.lo) d.lo[o] = from_f32<bf16>(r < d.lo_scaled_rows ? p * d.lo_scale : p);
    }
    tile[e >> 6][e & 63] = p;
  }
  if (!d.t) return;
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int cc = e >> 6, rr = e & 63;
    if (c0 + cc < d.C && r0 + rr < d.R) d.t[(int64_t)(c0 + cc) * d.R + r0 + rr] = from_f32<bf16>(tile[rr][cc]);
  }
}

}  // namespace

}  // namespace avf

namespace avf {
namespace {
// append the eleven tensors of one layer to a batch; returns 0 / non-zero
int adam_add_layer(AdamBatch& b, int& n, int& tiles, const avf_layer_cfg* cfg, const avf_layer_params* p,
                   const avf_layer_grads* g, const avf_layer_grads* exp_avg, const avf_layer_grads* exp_avg_sq, void* lowp) {
  AVF_REQUIRE(cfg->dtype == AVF_F32 || lowp, "layer_adam_step(bf16): lowp buffer missing");
  const int D = cfg->dim, I = cfg->heads * cfg->dim_head, M = cfg->mlp_dim;
  // the bf16 images in the order avf_layer_lowp_bytes carves them: Wqkv, Wqkv^T, Wo, Wo^T, W1, W1^T, W2, W2^T
  bf16* img[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (cfg->dtype == AVF_BF16) {
    size_t off = 0;
    const size_t bytes[4] = {(size_t)3 * I * D * 2, (size_t)D * I * 2, (size_t)M * D * 2, (size_t)D * M * 2};
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 2; ++j) {
        img[2 * i + j] = reinterpret_cast<bf16*>((char*)lowp + off);
        off += (bytes[i] + 255) & ~(size_t)255;
      }
    // (with cfg->mx8_fwd the MX-FP8 images follow the bf16 ones; the caller re-derives them from these: transformer.py)
    AVF_REQUIRE(off <= avf_layer_lowp_bytes(cfg), "layer_adam_step: lowp layout mismatch (%zu vs %zu)", off,
                avf_layer_lowp_bytes(cfg));
  }
  // fragment-major images behind them (weight-stationary GEMM): rewritten by the same launch
  LowpWs wsi = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (cfg->dtype == AVF_BF16) AVF_TRY(lowp_ws_images(cfg, lowp, &wsi));
  const int first = n;
  auto add = [&](const float* pp, float* gg, float* mm, float* vv, bf16* lo, bf16* t, int R, int C, float lo_scale = 1.0f,
                 int lo_scaled_rows = 0, void* lo_p = nullptr, void* t_p = nullptr) {
    if (!pp || (!gg && !lo && !t)) return;  // absent tensor, or nothing to do for it
    AdamDesc& d = b.d[n++];
    d.p = const_cast<float*>(pp); d.g = gg; d.m = mm; d.v = vv; d.lo = lo; d.t = t; d.R = R; d.C = C;
    d.lo_p = lo ? (bf16*)lo_p : nullptr; d.t_p = t ? (bf16*)t_p : nullptr;
    d.lo_scale = lo_scale; d.lo_scaled_rows = lo_scaled_rows;
    d.tile0 = tiles;
    d.tiles_c = R == 1 ? 1 : (C + 63) / 64;
    tiles += R == 1 ? (C + 4095) / 4096 : ((R + 63) / 64) * d.tiles_c;
  };
  add(p->w_qkv, g->w_qkv, exp_avg->w_qkv, exp_avg_sq->w_qkv, img[0], img[1], 3 * I, D, attn_q_prescale(cfg->dim_head), I,
      wsi.wqkv_p, nullptr);
  add(p->w_out, g->w_out, exp_avg->w_out, exp_avg_sq->w_out, img[2], img[3], D, I, 1.0f, 0, wsi.wo_p, wsi.wot_p);
  add(p->w1, g->w1, exp_avg->w1, exp_avg_sq->w1, img[4], img[5], M, D, 1.0f, 0, wsi.w1_p, nullptr);
  add(p->w2, g->w2, exp_avg->w2, exp_avg_sq->w2, img[6], img[7], D, M, 1.0f, 0, nullptr, wsi.w2t_p);
  add(p->ln1_w, g->ln1_w, exp_avg->ln1_w, exp_avg_sq->ln1_w, nullptr, nullptr, 1, D);
  add(p->ln1_b, g->ln1_b, exp_avg->ln1_b, exp_avg_sq->ln1_b, nullptr, nullptr, 1, D);
  add(p->b_out, g->b_out, exp_avg->b_out, exp_avg_sq->b_out, nullptr, nullptr, 1, D);
  add(p->ln2_w, g->ln2_w, exp_avg->ln2_w, exp_avg_sq->ln2_w, nullptr, nullptr, 1, D);
  add(p->ln2_b, g->ln2_b, exp_avg->ln2_b, exp_avg_sq->ln2_b, nullptr, nullptr, 1, D);
  add(p->b1, g->b1, exp_avg->b1, exp_avg_sq->b1, nullptr, nullptr, 1, M);
  add(p->b2, g->b2, exp_avg->b2, exp_avg_sq->b2, nullptr, nullptr, 1, D);
  for (int i = first; i < n; ++i)
    AVF_REQUIRE(!b.d[i].g || (b.d[i].m && b.d[i].v), "layer_adam_step: exp_avg / exp_avg_sq missing for an updated tensor");
  return 0;
}
}  // namespace
}  // namespace avf

namespace avf {
namespace {
// One optimizer step's launches, collected (avf_adam_batch_begin ... avf_adam_batch_end on the calling thread): the stack and
// tensor entry points append their descriptors here instead of launching; a full table, a change of hyper-parameters or the end
// of the session launches what is pending.  Outside a session every call launches its own table, as before.
struct AdamPending {
  bool active = false;
  AdamBatch b;
  int n = 0, tiles = 0;
  hipStream_t stream = nullptr;
};
thread_local AdamPending g_adam_pending;

int adam_flush(AdamPending& P) {
  if (P.n == 0) return 0;
  P.b.count = P.n;
  adam_layer_kernel<<<P.tiles, 256, 0, P.stream>>>(P.b);
  P.n = 0;
  P.tiles = 0;
  return check_launch("adam_layer_kernel");
}

// the table the next descriptors go to: the session's (flushed first when the hyper-parameters or the stream change) or `local`
AdamPending& adam_target(AdamPending& local, float lr, float b1, float b2, float eps, float wd, const float* step, hipStream_t s,
                         int* rc) {
  AdamPending& P = g_adam_pending.active ? g_adam_pending : local;
  *rc = 0;
  if (P.n > 0 && (P.b.lr != lr || P.b.b1 != b1 || P.b.b2 != b2 || P.b.eps != eps || P.b.wd != wd || P.b.step != step || P.stream != s))
    *rc = adam_flush(P);
  if (P.n == 0) memset(&P.b, 0, sizeof(P.b));
  P.b.lr = lr; P.b.b1 = b1; P.b.b2 = b2; P.b.eps = eps; P.b.wd = wd; P.b.step = step;
  P.stream = s;
  return P;
}
}  // namespace
}  // namespace avf

extern "C" int avf_adam_batch_begin(void) {
  using namespace avf;
  AVF_REQUIRE(!g_adam_pending.active, "adam_batch_begin: a batch is already open on this thread");
  g_adam_pending.active = true;
  g_adam_pending.n = 0;
  g_adam_pending.tiles = 0;
  return 0;
}

extern "C" int avf_adam_batch_end(void) {
  using namespace avf;
  AVF_REQUIRE(g_adam_pending.active, "adam_batch_end: no batch is open on this thread");
  g_adam_pending.active = false;
  return adam_flush(g_adam_pending);
}

// close the session WITHOUT launching what is pending (the caller failed half-way through collecting the step: a table that
// holds part of the model must not run).  Tables already launched - full ones, or flushed by a change of hyper-parameters -
// stay launched.  No-op when no session is open.
extern "C" int avf_adam_batch_abort(void) {
  using namespace avf;
  g_adam_pending.active = false;
  g_adam_pending.n = 0;
  g_adam_pending.tiles = 0;
  return 0;
}

extern "C" int avf_stack_adam_step(const avf_layer_cfg* cfg, int layers, const avf_layer_params* p, const avf_layer_grads* g,
                                   const avf_layer_grads* exp_avg, const avf_layer_grads* exp_avg_sq, void* const* lowp,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, const float* step,
                                   void* stream) {
  using namespace avf;
  AVF_REQUIRE(cfg && layers >= 0 && (layers == 0 || (p && g && exp_avg && exp_avg_sq)), "stack_adam_step: null pointer");
  AVF_REQUIRE(cfg->dim > 0 && cfg->heads > 0 && cfg->dim_head > 0 && cfg->mlp_dim > 0, "stack_adam_step: bad dimensions");
  AVF_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f,
              "stack_adam_step: bad hyper-parameters");
  AVF_REQUIRE(cfg->dtype == AVF_F32 || lowp, "stack_adam_step(bf16): lowp buffers missing");
  AdamPending local;
  int rc = 0;
  AdamPending& P = adam_target(local, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream, &rc);
  AVF_TRY(rc);
  for (int l = 0; l < layers; ++l) {
    if (P.n + 11 > ADAM_MAX) AVF_TRY(adam_flush(P));
    AVF_TRY(adam_add_layer(P.b, P.n, P.tiles, cfg, p + l, g + l, exp_avg + l, exp_avg_sq + l, lowp ? lowp[l] : nullptr));
  }
  if (&P == &local) AVF_TRY(adam_flush(P));
  return 0;
}

extern "C" int avf_layer_adam_step(const avf_layer_cfg* cfg, const avf_layer_params* p, const avf_layer_grads* g,
                                   const avf_layer_grads* exp_avg, const avf_layer_grads* exp_avg_sq, void* lowp, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, const float* step,
                                   void* stream) {
  void* lp[1] = {lowp};
  return avf_stack_adam_step(cfg, 1, p, g, exp_avg, exp_avg_sq, lp, lr, beta1, beta2, eps, weight_decay, step, stream);
}

extern "C" int avf_adam_step_tensors(int count, float* const* p, const float* const* g, float* const* exp_avg,
                                     float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, const float* step, void* stream) {
  using namespace avf;
  AVF_REQUIRE(count >= 0 && (count == 0 || (p && g && exp_avg && exp_avg_sq && numel)), "adam_step_tensors: null pointer");
  AVF_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f,
              "adam_step_tensors: bad hyper-parameters");
  AdamPending local;
  int rc = 0;
  AdamPending& P = adam_target(local, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream, &rc);
  AVF_TRY(rc);
  for (int i = 0; i < count; ++i) {
    if (!g[i] || numel[i] <= 0) continue;
    AVF_REQUIRE(p[i] && exp_avg[i] && exp_avg_sq[i] && numel[i] < (1LL << 31), "adam_step_tensors: bad tensor %d", i);
    if (P.n + 1 > ADAM_MAX) AVF_TRY(adam_flush(P));
    AdamDesc& d = P.b.d[P.n++];
    memset(&d, 0, sizeof(d));
    d.p = p[i]; d.g = g[i]; d.m = exp_avg[i]; d.v = exp_avg_sq[i]; d.R = 1; d.C = (int)numel[i];
    d.lo_scale = 1.0f;
    d.tile0 = P.tiles; d.tiles_c = 1;
    P.tiles += (int)((numel[i] + 4095) / 4096);
  }
  if (&P == &local) AVF_TRY(adam_flush(P));
  return 0;
}

// A full descriptor table (ADAM_MAX one-element tensors) through the real kernel, every element checked on the host: the
// table is a ~12.6 KB by-value kernel argument, which this ROCm stack takes (tools/diag/kernarg_probe.hip) - a runtime that
// truncated it would update only the first tensors.  Allocates and frees its own scratch; synchronises the stream.
extern "C" int avf_selftest_adam_table(void* stream) {
  using namespace avf;
  AVF_REQUIRE(!g_adam_pending.active, "selftest_adam_table: a batch is open on this thread");
  hipStream_t s = (hipStream_t)stream;
  constexpr int n = ADAM_MAX;
  float* buf = nullptr;  // p | g | m | v | step
  AVF_REQUIRE(hipMalloc((void**)&buf, (4 * n + 1) * sizeof(float)) == hipSuccess, "selftest_adam_table: hipMalloc failed");
  float host[4 * n + 1];
  for (int i = 0; i < n; ++i) { host[i] = 1.0f + i; host[n + i] = 1.0f; host[2 * n + i] = 0.f; host[3 * n + i] = 0.f; }
  host[4 * n] = 1.0f;
  int rc = hipMemcpyAsync(buf, host, sizeof(host), hipMemcpyHostToDevice, s) == hipSuccess ? 0 : 1;
  float* p[n]; const float* g[n]; float* m[n]; float* v[n]; int64_t numel[n];
  for (int i = 0; i < n; ++i) { p[i] = buf + i; g[i] = buf + n + i; m[i] = buf + 2 * n + i; v[i] = buf + 3 * n + i; numel[i] = 1; }
  if (!rc) rc = avf_adam_step_tensors(n, p, g, m, v, numel, 0.5f, 0.9f, 0.999f, 0.f, 0.f, buf + 4 * n, stream);
  if (!rc) rc = hipMemcpyAsync(host, buf, n * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess ? 0 : 1;
  if (!rc) rc = hipStreamSynchronize(s) == hipSuccess ? 0 : 1;
  (void)hipFree(buf);
  AVF_REQUIRE(rc == 0, "selftest_adam_table: launch or copy failed");
  // first step, g = 1, eps = 0: m_hat / sqrt(v_hat) = 1, so every p moved by exactly lr
  for (int i = 0; i < n; ++i)
    AVF_REQUIRE(fabsf(host[i] - (0.5f + i)) < 1e-4f, "selftest_adam_table: tensor %d of %d not updated (%g)", i, n, (double)host[i]);
  return 0;
}
