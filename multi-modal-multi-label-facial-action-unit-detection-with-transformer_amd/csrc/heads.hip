// heads.hip - the token producers / consumers either side of the transformer stack:
//   BatchNorm1d of AU_former                         models/heads.py:263,293  (nn.BatchNorm1d, eps 1e-5, momentum 0.1)
//   the 12 per-token bias-free dots                  models/heads.py:325-337, models/tformer.py:389-401
//   TFormer's cls-token / positional assembly        models/vformer.py:279-287
//   avformer's feature-axis fusion + pos             models/avformer.py:100, models/tformer.py:383-386
//   ResFormer's feature-map <-> token transposes     models/sformer.py:313-327
// The 12-way projection itself (heads.py:294-319) and the AU logits Linear are GEMMs of the parity kernel (gemm_f32.hip)
// on the concatenated weights; these kernels are the launch-bound glue around them, fp32 throughout (tiny tensors:
// a few hundred rows), one launch each, no atomics (deterministic).
#include "common.hpp"

namespace avf {
namespace {

// ---------------------------------------------------------------------------------------------
// BatchNorm1d over [B, C]: a workgroup owns 32 features; its 256 threads are 32 features x 8 row lanes (consecutive
// threads read consecutive features: coalesced), partial sums meet in LDS in a fixed order (deterministic).  training:
// batch statistics (biased variance for the normalisation, unbiased for the running estimate, as torch); eval: the
// running statistics.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bn_reduce8(float v, float (*red)[32], int cl, int rl) {
  __syncthreads();  // the previous use of `red` is over
  red[rl][cl] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += red[j][cl];
  return s;
}

__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, int64_t* __restrict__ nbt,
                                                       float* __restrict__ y,
                                                       float* __restrict__ mean_out, float* __restrict__ invstd_out, int B,
                                                       int C, float eps, float momentum, int training) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool ok = c < C;
  if (blockIdx.x == 0 && threadIdx.x == 0 && training && nbt) nbt[0] += 1;  // nn.BatchNorm1d.num_batches_tracked
  float mu, var;
  if (training) {
    float s = 0.f;
    if (ok)
      for (int b = rl; b < B; b += 8) s += x[(int64_t)b * C + c];
    mu = bn_reduce8(s, red, cl, rl) / (float)B;
    float q = 0.f;
    if (ok)
      for (int b = rl; b < B; b += 8) {
        const float d = x[(int64_t)b * C + c] - mu;
        q += d * d;
      }
    q = bn_reduce8(q, red, cl, rl);
    var = q / (float)B;
    if (ok && rl == 0) {
      if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
      if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (B > 1 ? q / (float)(B - 1) : var);
    }
  } else {
    mu = ok ? run_mean[c] : 0.f;
    var = ok ? run_var[c] : 1.f;
  }
  if (!ok) return;
  const float is = 1.0f / sqrtf(var + eps);
  if (rl == 0) {
    mean_out[c] = mu;
    invstd_out[c] = is;
  }
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  for (int b = rl; b < B; b += 8) y[(int64_t)b * C + c] = (x[(int64_t)b * C + c] - mu) * is * g + bt;
}

// training: dx = gamma*invstd*(dy - mean_b(dy) - xhat*mean_b(dy*xhat)); eval: dx = dy*gamma*invstd (statistics are constants)
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                       int training) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool ok = c < C;
  const float mu = ok ? mean[c] : 0.f, is = ok ? invstd[c] : 0.f, g = (ok && gamma) ? gamma[c] : 1.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int b = rl; b < B; b += 8) {
      const float d = dy[(int64_t)b * C + c];
      s1 += d;
      s2 += d * (x[(int64_t)b * C + c] - mu) * is;
    }
  s1 = bn_reduce8(s1, red, cl, rl);
  s2 = bn_reduce8(s2, red, cl, rl);
  if (!ok) return;
  if (rl == 0) {
    if (dgamma) dgamma[c] = s2;
    if (dbeta) dbeta[c] = s1;
  }
  if (!dx) return;
  const float m1 = training ? s1 / (float)B : 0.f, m2 = training ? s2 / (float)B : 0.f;
  for (int b = rl; b < B; b += 8) {
    const float xh = (x[(int64_t)b * C + c] - mu) * is;
    dx[(int64_t)b * C + c] = g * is * (dy[(int64_t)b * C + c] - m1 - xh * m2);
  }
}

// ---------------------------------------------------------------------------------------------
// per-token dots: out[b, t] = sum_e tok[b, t, e] * w[t, e]  (w: T rows of E, row stride ldw - the bias-free Linear(E,1)
// number t+1 applied to token t).  One wavefront per (b, t); out has row stride ldo and columns T..pad_to-1 are zeroed
// (the reference's [B,21] layout).  Backward: dtok = dout[b,t] * w[t,:]; dw[t,:] = sum_b dout[b,t] * tok[b,t,:].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ w, int64_t ldw,
                                                         float* __restrict__ out, int64_t ldo, int pad_to, int B, int T, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, t)
  if (row >= (int64_t)B * T) return;
  const int64_t b = row / T;
  const int t = (int)(row - b * T);
  const float* tr = tok + row * E;
  const float* wr = w + (int64_t)t * ldw;
  float s = 0.f;
  for (int e = lane; e < E; e += 64) s += tr[e] * wr[e];
  s = wave_sum(s);
  if (lane == 0) out[b * ldo + t] = s;
  if (t == 0)
    for (int c = T + lane; c < pad_to; c += 64) out[b * ldo + c] = 0.f;
}
__global__ __launch_bounds__(256) void rowdot_bwd_tok_kernel(const float* __restrict__ dout, int64_t ldo,
                                                             const float* __restrict__ w, int64_t ldw, float* __restrict__ dtok,
                                                             int B, int T, int E) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T * E) return;
  const int e = (int)(i % E);
  const int64_t bt = i / E;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  dtok[i] = dout[b * ldo + t] * w[(int64_t)t * ldw + e];
}
// grid (ceil(E / 64), T); 256 threads = 64 columns x 4 batch lanes, combined through LDS in a fixed order
__global__ __launch_bounds__(256) void rowdot_bwd_w_kernel(const float* __restrict__ dout, int64_t ldo,
                                                           const float* __restrict__ tok, float* __restrict__ dw, int64_t lddw,
                                                           int B, int T, int E) {
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el, t = blockIdx.y;
  float s = 0.f;
  if (e < E)
    for (int b = bl; b < B; b += 4) s += dout[(int64_t)b * ldo + t] * tok[((int64_t)b * T + t) * E + e];
  red[bl][el] = s;
  __syncthreads();
  if (bl == 0 && e < E) dw[(int64_t)t * lddw + e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// ---------------------------------------------------------------------------------------------
// token assembly: out[b, t, :] = (t < n_lead ? lead[t, :] : x[b, t - n_lead, :]) + pos[t, :]
//   TFormer: lead = cls_token [1, D] (vformer.py:282-283); heads that only add a positional table: n_lead = 0.
// and the feature-axis fusion of avformer.py:100 / tformer.py:383-386:
//   out[b, t, :] = concat(a[b, t, :Ea], v[b, t, :Ev]) + pos[t, :]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ x, const float* __restrict__ lead,
                                                              const float* __restrict__ pos, float* __restrict__ out, int B,
                                                              int P, int n_lead, int D) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int T = P + n_lead;
  if (i >= (int64_t)B * T * D) return;
  const int d = (int)(i % D);
  const int64_t bt = i / D;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  const float v = t < n_lead ? lead[(int64_t)t * D + d] : x[(b * P + (t - n_lead)) * D + d];
  out[i] = v + (pos ? pos[(int64_t)t * D + d] : 0.f);
}
__global__ __launch_bounds__(256) void cat_features_kernel(const float* __restrict__ a, const float* __restrict__ v,
                                                           const float* __restrict__ pos, float* __restrict__ out,
                                                           int64_t rows, int T, int Ea, int Ev) {
  const int E = Ea + Ev;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * E) return;
  const int e = (int)(i % E);
  const int64_t r = i / E;  // (b, t)
  const int t = (int)(r % T);
  const float val = e < Ea ? a[r * Ea + e] : v[r * Ev + (e - Ea)];
  out[i] = val + (pos ? pos[(int64_t)t * E + e] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// [B, C, S] -> [B, S, C] (+ pos[S, C], nullable) through a 32 x 33 LDS tile: the feature-map <-> token permutes of
// ResFormer.forward (sformer.py:316-318 with the positional add, :326-327 back without)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_add_kernel(const float* __restrict__ in, const float* __restrict__ pos,
                                                            float* __restrict__ out, int C, int S) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const int c0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, s = s0 + tx;
    tile[r][tx] = (c < C && s < S) ? in[(b * C + c) * S + s] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r, c = c0 + tx;
    if (s < S && c < C) out[(b * S + s) * C + c] = tile[tx][r] + (pos ? pos[(int64_t)s * C + c] : 0.f);
  }
}

// out[r, c] = 0 for c in [c0, c1): the unused slots of the reference's [B,21] output (train.py:136-138)
__global__ __launch_bounds__(256) void zero_cols_kernel(float* __restrict__ out, int64_t ld, int rows, int c0, int c1) {
  const int w = c1 - c0;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * w) return;
  out[(i / w) * ld + c0 + (int)(i % w)] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// A small nn.Linear on a handful of rows, written into a zero-padded row: out[b, o] = x[b, :] . w[o, :] + bias[o] for o < O,
// 0 for O <= o < width - the AU logits of a pooled feature in the reference's [B, 21] layout (avformer.py:101-105) in ONE
// launch (the fp32 GEMM takes a split-K launch, a fold and a zero fill for the same 32 x 12 x 512 product).  One workgroup
// per row; a wave owns outputs wave, wave + 4, ...; the row sits in registers.
// ---------------------------------------------------------------------------------------------
constexpr int LP_MAXK = 4096;  // 16 values per lane
__global__ __launch_bounds__(256) void linear_pad_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out, int K,
                                                             int O, int width) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nper = (K + 255) / 256;  // float4 chunks per lane (K % 4 == 0)
  float4 xv[LP_MAXK / 256];
#pragma unroll
  for (int i = 0; i < LP_MAXK / 256; ++i) {
    const int c = (lane + 64 * i) * 4;
    xv[i] = (i < nper && c < K) ? *reinterpret_cast<const float4*>(x + (int64_t)b * K + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int o = wave; o < width; o += 4) {
    float v = 0.f;
    if (o < O) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < LP_MAXK / 256; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nper && c < K) {
          const float4 ww = *reinterpret_cast<const float4*>(w + (int64_t)o * K + c);
          a += (xv[i].x * ww.x + xv[i].y * ww.y) + (xv[i].z * ww.z + xv[i].w * ww.w);
        }
      }
      v = wave_sum(a) + (bias ? bias[o] : 0.f);
    }
    if (lane == 0) out[(int64_t)b * width + o] = v;
  }
}

// backward of the same: blocks [0, B): dx[b, :] = sum_o dl[b, o] w[o, :]; blocks [B, B + O): dw[o, :] = sum_b dl[b, o] x[b, :]
// and db[o] = sum_b dl[b, o] (rows summed in order: deterministic).  dl rows are ldd apart (the padded row is read in place).
__global__ __launch_bounds__(256) void linear_pad_bwd_kernel(const float* __restrict__ dl, int64_t ldd,
                                                             const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ dx, float* __restrict__ dw,
                                                             float* __restrict__ db, int B, int K, int O) {
  const int blk = blockIdx.x;
  if (blk < B) {
    if (!dx) return;
    for (int c = threadIdx.x * 4; c < K; c += 1024) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4  // independent loads of several iterations in flight (the loop is latency-bound: a dozen rows)
      for (int o = 0; o < O; ++o) {
        const float g = dl[(int64_t)blk * ldd + o];
        const float4 ww = *reinterpret_cast<const float4*>(w + (int64_t)o * K + c);
        a.x = fmaf(g, ww.x, a.x); a.y = fmaf(g, ww.y, a.y); a.z = fmaf(g, ww.z, a.z); a.w = fmaf(g, ww.w, a.w);
      }
      *reinterpret_cast<float4*>(dx + (int64_t)blk * K + c) = a;
    }
    return;
  }
  const int o = blk - B;
  if (dw) {
    for (int c = threadIdx.x * 4; c < K; c += 1024) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int b = 0; b < B; ++b) {
        const float g = dl[(int64_t)b * ldd + o];
        const float4 xx = *reinterpret_cast<const float4*>(x + (int64_t)b * K + c);
        a.x = fmaf(g, xx.x, a.x); a.y = fmaf(g, xx.y, a.y); a.z = fmaf(g, xx.z, a.z); a.w = fmaf(g, xx.w, a.w);
      }
      *reinterpret_cast<float4*>(dw + (int64_t)o * K + c) = a;
    }
  }
  if (db && threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dl[(int64_t)b * ldd + o];
    db[o] = a;
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)ceil_div(n, 256); }

}  // namespace
}  // namespace avf

using namespace avf;

extern "C" int avf_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* y, float* mean, float* invstd, int batch, int features,
                            float eps, float momentum, int training, void* stream) {
  AVF_REQUIRE(x && y && mean && invstd && batch > 0 && features > 0, "bn1d_fwd: bad arguments");
  AVF_REQUIRE(training || (running_mean && running_var), "bn1d_fwd: eval mode needs the running statistics");
  bn1d_fwd_kernel<<<(unsigned)ceil_div(features, 32), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, running_mean, running_var,
                                                                          num_batches_tracked, y, mean, invstd, batch, features,
                                                                          eps, momentum, training);
  return check_launch("bn1d_fwd_kernel");
}

extern "C" int avf_bn1d_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd,
                            float* dx, float* dgamma, float* dbeta, int batch, int features, int training, void* stream) {
  AVF_REQUIRE(x && dy && mean && invstd && batch > 0 && features > 0, "bn1d_bwd: bad arguments");
  bn1d_bwd_kernel<<<(unsigned)ceil_div(features, 32), 256, 0, (hipStream_t)stream>>>(x, dy, gamma, mean, invstd, dx, dgamma, dbeta, batch,
                                                                          features, training);
  return check_launch("bn1d_bwd_kernel");
}

extern "C" int avf_token_dots_fwd(const float* tokens, const float* w, int64_t ldw, float* out, int64_t ldo, int pad_to,
                                  int batch, int tokens_per_clip, int emb, void* stream) {
  AVF_REQUIRE(tokens && w && out && batch > 0 && tokens_per_clip > 0 && emb > 0 && ldo >= tokens_per_clip && pad_to <= ldo,
              "token_dots_fwd: bad arguments");
  rowdot_fwd_kernel<<<(unsigned)ceil_div((int64_t)batch * tokens_per_clip, 4), 256, 0, (hipStream_t)stream>>>(
      tokens, w, ldw, out, ldo, pad_to, batch, tokens_per_clip, emb);
  return check_launch("rowdot_fwd_kernel");
}

extern "C" int avf_token_dots_bwd(const float* dout, int64_t ldo, const float* tokens, const float* w, int64_t ldw,
                                  float* dtokens, float* dw, int64_t lddw, int batch, int tokens_per_clip, int emb,
                                  void* stream) {
  AVF_REQUIRE(dout && tokens && w && batch > 0 && tokens_per_clip > 0 && emb > 0, "token_dots_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtokens) {
    rowdot_bwd_tok_kernel<<<blocks_for((int64_t)batch * tokens_per_clip * emb), 256, 0, s>>>(dout, ldo, w, ldw, dtokens, batch,
                                                                                             tokens_per_clip, emb);
    AVF_TRY(check_launch("rowdot_bwd_tok_kernel"));
  }
  if (dw) {
    rowdot_bwd_w_kernel<<<dim3((unsigned)ceil_div(emb, 64), (unsigned)tokens_per_clip), 256, 0, s>>>(dout, ldo, tokens, dw, lddw,
                                                                                                     batch, tokens_per_clip, emb);
    AVF_TRY(check_launch("rowdot_bwd_w_kernel"));
  }
  return 0;
}

extern "C" int avf_assemble_tokens(const float* x, const float* lead, const float* pos, float* out, int batch, int patches,
                                   int n_lead, int dim, void* stream) {
  AVF_REQUIRE(out && batch > 0 && patches >= 0 && n_lead >= 0 && patches + n_lead > 0 && dim > 0 && (x || patches == 0) &&
                  (lead || n_lead == 0), "assemble_tokens: bad arguments");
  assemble_tokens_kernel<<<blocks_for((int64_t)batch * (patches + n_lead) * dim), 256, 0, (hipStream_t)stream>>>(
      x, lead, pos, out, batch, patches, n_lead, dim);
  return check_launch("assemble_tokens_kernel");
}

extern "C" int avf_cat_features(const float* a, const float* v, const float* pos, float* out, int batch, int tokens_per_clip,
                                int emb_a, int emb_v, void* stream) {
  AVF_REQUIRE(a && v && out && batch > 0 && tokens_per_clip > 0 && emb_a > 0 && emb_v > 0, "cat_features: bad arguments");
  const int64_t rows = (int64_t)batch * tokens_per_clip;
  cat_features_kernel<<<blocks_for(rows * (emb_a + emb_v)), 256, 0, (hipStream_t)stream>>>(a, v, pos, out, rows,
                                                                                           tokens_per_clip, emb_a, emb_v);
  return check_launch("cat_features_kernel");
}

extern "C" int avf_transpose_add(const float* in, const float* pos, float* out, int batch, int rows, int cols, void* stream) {
  AVF_REQUIRE(in && out && batch > 0 && batch <= 65535 && rows > 0 && cols > 0, "transpose_add: bad arguments");
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32), (unsigned)batch);
  AVF_REQUIRE(grid.y <= 65535, "transpose_add: too many rows");
  transpose_add_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, pos, out, rows, cols);
  return check_launch("transpose_add_kernel");
}

// counter += 1; snapshot = counter: the dropout seed of one forward of a stack (transformer.py: the kernels of that forward AND
// of its backward read the snapshot; an in-place add and a clone were two launches per stack and step)
__global__ void seed_advance_kernel(int64_t* __restrict__ counter, int64_t* __restrict__ snapshot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t v = counter[0] + 1;
    counter[0] = v;
    snapshot[0] = v;
  }
}

extern "C" int avf_seed_advance(int64_t* counter, int64_t* snapshot, void* stream) {
  AVF_REQUIRE(counter && snapshot && counter != snapshot, "seed_advance: bad arguments");
  seed_advance_kernel<<<1, 64, 0, (hipStream_t)stream>>>(counter, snapshot);
  return check_launch("seed_advance_kernel");
}

extern "C" int avf_zero_cols(float* out, int64_t ld, int rows, int c0, int c1, void* stream) {
  AVF_REQUIRE(out && rows > 0 && c0 >= 0 && c1 >= c0 && ld >= c1, "zero_cols: bad arguments");
  if (c1 == c0) return 0;
  zero_cols_kernel<<<blocks_for((int64_t)rows * (c1 - c0)), 256, 0, (hipStream_t)stream>>>(out, ld, rows, c0, c1);
  return check_launch("zero_cols_kernel");
}

extern "C" int avf_linear_pad_fwd(const float* x, const float* w, const float* bias, float* out, int rows, int in_features,
                                  int out_features, int width, void* stream) {
  AVF_REQUIRE(x && w && out && rows > 0 && in_features > 0 && out_features > 0 && width >= out_features,
              "linear_pad_fwd: bad arguments");
  AVF_REQUIRE(in_features % 4 == 0 && in_features <= LP_MAXK && (((uintptr_t)x | (uintptr_t)w) & 15) == 0,
              "linear_pad_fwd: in_features must be a multiple of 4, at most %d, operands 16-byte aligned", LP_MAXK);
  linear_pad_fwd_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(x, w, bias, out, in_features, out_features, width);
  return check_launch("linear_pad_fwd_kernel");
}

extern "C" int avf_linear_pad_bwd(const float* dout, int64_t ldd, const float* x, const float* w, float* dx, float* dw,
                                  float* db, int rows, int in_features, int out_features, void* stream) {
  AVF_REQUIRE(dout && x && w && rows > 0 && in_features > 0 && out_features > 0 && ldd >= out_features,
              "linear_pad_bwd: bad arguments");
  AVF_REQUIRE(in_features % 4 == 0 && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)dw) & 15) == 0,
              "linear_pad_bwd: in_features must be a multiple of 4, operands 16-byte aligned");
  if (!dx && !dw && !db) return 0;
  linear_pad_bwd_kernel<<<rows + out_features, 256, 0, (hipStream_t)stream>>>(dout, ldd, x, w, dx, dw, db, rows, in_features,
                                                                               out_features);
  return check_launch("linear_pad_bwd_kernel");
}
