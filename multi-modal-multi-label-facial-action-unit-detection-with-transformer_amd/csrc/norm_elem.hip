// norm_elem.hip - HBM-bound kernels of the hot path: LayerNorm fwd/bwd (wavefront reductions),
// column sums (bias gradients), fp32->bf16 casts / weight transposes, AU loss.
//
// Reference math: models/heads.py:178-185 (PreNorm/nn.LayerNorm), models/loss.py:63-103 (AULoss).
#include <algorithm>

#include "common.hpp"

namespace avf {

// =============================================================================================
// LayerNorm forward: one wavefront per row; fp32 statistics; output fp32 or bf16.
// Algorithmic bytes per row: 4*D read + sizeof(out)*D written + 8 (mean, rstd).
// =============================================================================================
template <typename OutT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, OutT* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  OutT* yr = y + row * D;
  float s = 0.f;
  if ((D & 3) == 0) {
    for (int c = lane * 4; c < D; c += 256) {
      float4 v = *reinterpret_cast<const float4*>(xr + c);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int c = lane; c < D; c += 64) s += xr[c];
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  if ((D & 3) == 0) {
    for (int c = lane * 4; c < D; c += 256) {
      float4 v = *reinterpret_cast<const float4*>(xr + c);
      float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  } else {
    for (int c = lane; c < D; c += 64) {
      float a = xr[c] - mu;
      q += a * a;
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  if ((D & 3) == 0) {
    for (int c = lane * 4; c < D; c += 256) {
      float4 v = *reinterpret_cast<const float4*>(xr + c);
      float4 g = *reinterpret_cast<const float4*>(gamma + c);
      float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v.x - mu) * rs * g.x + b.x;
      o.y = (v.y - mu) * rs * g.y + b.y;
      o.z = (v.z - mu) * rs * g.z + b.z;
      o.w = (v.w - mu) * rs * g.w + b.w;
      store4<OutT>(yr + c, o);
    }
  } else {
    for (int c = lane; c < D; c += 64) yr[c] = from_f32<OutT>((xr[c] - mu) * rs * gamma[c] + beta[c]);
  }
}

// D % 4 == 0 and D <= 256*NV: the row lives in registers (one HBM read, no re-reads from cache)
// MX: also emit the MX-FP8 image of the row (common.hpp mx8_encode4: D % 32 == 0, so the 8 lanes of a block are
// live together) - the A operand of the following forward GEMM in the fp8 mode (layer.hip)
template <typename OutT, int NV, bool MX = false, typename InT = float>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const InT* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, OutT* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         int64_t rows, int D, float eps, uint8_t* __restrict__ yq = nullptr,
                                                         uint8_t* __restrict__ ys = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    v[i] = c < D ? load4<InT>(x + row * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < D) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      const float4 o = make_float4((v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y,
                                   (v[i].z - mu) * rs * g.z + b.z, (v[i].w - mu) * rs * g.w + b.w);
      store4<OutT>(y + row * D + c, o);
      if (MX) {
        const float ov[4] = {o.x, o.y, o.z, o.w};
        uint32_t sb;
        const uint32_t qw = mx8_encode4(ov, &sb);
        *reinterpret_cast<uint32_t*>(yq + row * D + c) = qw;
        if ((lane & 7) == 0) ys[row * (D >> 5) + (c >> 5)] = (uint8_t)sb;
      }
    }
  }
}

static int ln_row8_on() {
  static const int on = [] {
    const char* e = tuning_env("AVF_LN_ROW8");  // tuning / A-B aid: 0 = the one-row-per-wave kernels
    return (e && *e) ? atoi(e) : 1;
  }();
  return on;
}

// ---- the bf16 residual stream's kernels (bf16 in, bf16 out, D % 8 == 0, D <= 512 * NV8) -----------------------------
// A lane owns 8 consecutive columns per 512 (one 16-byte access); a wave works on RU rows at once with all their loads in
// flight before the first reduction: 4 x 4 rows per workgroup, ~10 waves per CU x RU KiB per stream in flight (the
// one-row-per-wave form had 1 KiB per wave and ran at 2.5 TB/s; DESIGN_HISTORY.md section 14).
__device__ __forceinline__ void unpack8(const uint4& r, float (&v)[8]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// MX: also the MX-FP8 image of the rows (D % 32 == 0: a 32-block is the 8 columns of the four lanes of a quad)
template <int NV8, int RU, bool MX = false>
__global__ __launch_bounds__(256) void ln_fwd_row8_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                          int D, float eps, uint8_t* __restrict__ yq = nullptr,
                                                          uint8_t* __restrict__ ys = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RU;
  if (row0 >= rows) return;
  uint4 raw[RU][NV8];
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;  // (rows past the end re-read the last one; nothing is stored)
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int c = lane * 8 + 512 * i;
      raw[r][i] = c < D ? *reinterpret_cast<const uint4*>(x + row * D + c) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  float gm[NV8][8], bt[NV8][8];
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int c = lane * 8 + 512 * i;
    if (c < D) {
      load8f(gamma + c, gm[i]);
      load8f(beta + c, bt[i]);
    }
  }
  const float invD = 1.0f / (float)D;
  float mu[RU], rs[RU];
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      float v[8];
      unpack8(raw[r][i], v);
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    mu[r] = wave_sum(s) * invD;
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      if (lane * 8 + 512 * i < D) {
        float v[8];
        unpack8(raw[r][i], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float a = v[k] - mu[r];
          q = fmaf(a, a, q);
        }
      }
    }
    rs[r] = 1.0f / sqrtf(wave_sum(q) * invD + eps);
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    if (row0 + r >= rows) break;
    if (lane == 0) {
      mean[row0 + r] = mu[r];
      rstd[row0 + r] = rs[r];
    }
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int c = lane * 8 + 512 * i;
      if (c < D) {
        float v[8], o[8];
        unpack8(raw[r][i], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (v[k] - mu[r]) * rs[r] * gm[i][k] + bt[i][k];
        *reinterpret_cast<uint4*>(y + (row0 + r) * D + c) = pack8(o);
        if constexpr (MX) {  // (c < D is uniform over a quad: D % 32 == 0)
          const MxBlock mb = mx8_encode(o);
          *reinterpret_cast<uint2*>(yq + (row0 + r) * D + c) = mb.q;
          if ((lane & 3) == 0) ys[(row0 + r) * (D >> 5) + (c >> 5)] = (uint8_t)mb.scale;
        }
      }
    }
  }
}

int layernorm_fwd(const void* xv, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                  float* rstd, int64_t rows, int dim, float eps, hipStream_t s, void* mx_q, void* mx_s, int x_dtype) {
  AVF_REQUIRE(rows > 0 && dim > 0, "layernorm_fwd: bad shape rows=%lld dim=%d", (long long)rows, dim);
  AVF_REQUIRE(y_dtype == AVF_F32 || y_dtype == AVF_BF16, "layernorm_fwd: bad dtype %d", y_dtype);
  AVF_REQUIRE(x_dtype == AVF_F32 || (x_dtype == AVF_BF16 && y_dtype == AVF_BF16 && dim % 4 == 0 && dim <= 1536),
              "layernorm_fwd: a bf16 input needs a bf16 output, dim %% 4 == 0 and dim <= 1536 (dim=%d)", dim);
  const float* x = (const float*)xv;
  TimingScope ts(KC_LAYERNORM, 0.0, (double)rows * dim * ((x_dtype == AVF_BF16 ? 2.0 : 4.0) + (y_dtype == AVF_BF16 ? 2.0 : 4.0) +
                                                         (mx_q ? 1.03125 : 0.0)), s, /*per_kernel=*/true);
  dim3 grid((unsigned)ceil_div(rows, 4)), block(256);
  if (x_dtype == AVF_BF16 && dim % 8 == 0 && (!mx_q || (mx_s && dim % 32 == 0)) && ln_row8_on()) {  // four rows per wave, 16-byte accesses
    const bf16* xb = (const bf16*)xv;
    constexpr int RU = 4;
    dim3 g8((unsigned)ceil_div(rows, 4 * RU));
#define LAUNCH_R8(NVV)                                                                                                       \
  do {                                                                                                                       \
    if (mx_q) launch_in_scope(&ts, ln_fwd_row8_kernel<NVV, RU, true>, g8, block, 0, s, xb, gamma, beta, (bf16*)y, mean, rstd,  \
                              rows, dim, eps, (uint8_t*)mx_q, (uint8_t*)mx_s);                                               \
    else launch_in_scope(&ts, ln_fwd_row8_kernel<NVV, RU, false>, g8, block, 0, s, xb, gamma, beta, (bf16*)y, mean, rstd,      \
                         rows, dim, eps, (uint8_t*)nullptr, (uint8_t*)nullptr);                                              \
  } while (0)
    switch ((dim + 511) / 512) {
      case 1: LAUNCH_R8(1); break;
      case 2: LAUNCH_R8(2); break;
      default: LAUNCH_R8(3); break;
    }
#undef LAUNCH_R8
    return check_launch("ln_fwd_row8_kernel");
  }
  if (x_dtype == AVF_BF16) {  // bf16 residual stream: bf16 in, bf16 out (+ optional MX-FP8 image)
    const bf16* xb = (const bf16*)xv;
#define LAUNCH_LO(NVV)                                                                                                    \
  do {                                                                                                                    \
    if (mx_q) launch_in_scope(&ts, ln_fwd_reg_kernel<bf16, NVV, true, bf16>, grid, block, 0, s, xb, gamma, beta, (bf16*)y, mean, rstd, \
                              rows, dim, eps, (uint8_t*)mx_q, (uint8_t*)mx_s);                                            \
    else launch_in_scope(&ts, ln_fwd_reg_kernel<bf16, NVV, false, bf16>, grid, block, 0, s, xb, gamma, beta, (bf16*)y, mean, rstd, \
                         rows, dim, eps, (uint8_t*)nullptr, (uint8_t*)nullptr);                                           \
  } while (0)
    AVF_REQUIRE(!mx_q || (mx_s && dim % 32 == 0), "layernorm_fwd: the MX-FP8 image needs dim %% 32 == 0");
    switch ((dim + 255) / 256) {
      case 1: LAUNCH_LO(1); break;
      case 2: LAUNCH_LO(2); break;
      case 3: LAUNCH_LO(3); break;
      case 4: LAUNCH_LO(4); break;
      default: LAUNCH_LO(6); break;
    }
#undef LAUNCH_LO
    return check_launch("ln_fwd_reg_kernel(bf16 in)");
  }
  if (mx_q) {
    AVF_REQUIRE(mx_s && y_dtype == AVF_BF16 && dim % 32 == 0 && dim <= 1536,
                "layernorm_fwd: the MX-FP8 image needs bf16 output, dim %% 32 == 0 and dim <= 1536 (dim=%d)", dim);
#define LAUNCH_MX(NVV)                                                                                              \
  launch_in_scope(&ts, ln_fwd_reg_kernel<bf16, NVV, true>, grid, block, 0, s, x, gamma, beta, (bf16*)y, mean, rstd, rows, dim, eps, \
                  (uint8_t*)mx_q, (uint8_t*)mx_s)
    switch ((dim + 255) / 256) {
      case 1: LAUNCH_MX(1); break;
      case 2: LAUNCH_MX(2); break;
      case 3: LAUNCH_MX(3); break;
      case 4: LAUNCH_MX(4); break;
      default: LAUNCH_MX(6); break;
    }
#undef LAUNCH_MX
    return check_launch("ln_fwd_reg_kernel(mx)");
  }
  if (dim % 4 == 0 && dim <= 1536) {
    const int nv = (dim + 255) / 256;
#define LAUNCH_NV(T, NVV)                                                                                             \
  launch_in_scope(&ts, ln_fwd_reg_kernel<T, NVV, false>, grid, block, 0, s, x, gamma, beta, (T*)y, mean, rstd, rows, dim, eps, \
                  (uint8_t*)nullptr, (uint8_t*)nullptr)
#define LAUNCH_T(T)                 \
  switch (nv) {                     \
    case 1: LAUNCH_NV(T, 1); break; \
    case 2: LAUNCH_NV(T, 2); break; \
    case 3: LAUNCH_NV(T, 3); break; \
    case 4: LAUNCH_NV(T, 4); break; \
    default: LAUNCH_NV(T, 6); break;\
  }
    if (y_dtype == AVF_F32) { LAUNCH_T(float) } else { LAUNCH_T(bf16) }
#undef LAUNCH_T
#undef LAUNCH_NV
  } else if (y_dtype == AVF_F32) {
    launch_in_scope(&ts, ln_fwd_kernel<float>, grid, block, 0, s, x, gamma, beta, (float*)y, mean, rstd, rows, dim, eps);
  } else {
    launch_in_scope(&ts, ln_fwd_kernel<bf16>, grid, block, 0, s, x, gamma, beta, (bf16*)y, mean, rstd, rows, dim, eps);
  }
  return check_launch("ln_fwd_kernel");
}

// =============================================================================================
// LayerNorm backward.  One wavefront per row computes dx; the block accumulates the per-column sums
// (dgamma, dbeta, and the column sum of dx = bias gradient of the producing Linear) in LDS with
// ds_add_f32 and writes one partial per block; a second kernel folds the partials.
//   xhat = (x-mu)*rstd ; g = dy*gamma ; dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) + dres
// =============================================================================================
constexpr int LNB_ROWS_PER_BLOCK = 32;

template <typename DyT, bool VEC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const DyT* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16* __restrict__ dx_lo,
                                                     float* __restrict__ partial, int64_t rows, int D,
                                                     int want_colsum) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [3][D]
  float* s_dg = lds;
  float* s_db = lds + D;
  float* s_cs = lds + 2 * D;
  for (int i = threadIdx.x; i < 3 * D; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS_PER_BLOCK;
  const float invD = 1.0f / (float)D;
  for (int rr = wave; rr < LNB_ROWS_PER_BLOCK; rr += 4) {
    const int64_t row = row0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    const DyT* dyr = dy + row * D;
    const float* xr = x + row * D;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
      for (int c = lane * 4; c < D; c += 256) {
        float4 d = load4<DyT>(dyr + c);
        float4 v = *reinterpret_cast<const float4*>(xr + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        float g0 = d.x * g.x, g1 = d.y * g.y, g2 = d.z * g.z, g3 = d.w * g.w;
        s1 += (g0 + g1) + (g2 + g3);
        s2 += (g0 * (v.x - mu) + g1 * (v.y - mu)) + (g2 * (v.z - mu) + g3 * (v.w - mu));
      }
    } else {
      for (int c = lane; c < D; c += 64) {
        float g0 = to_f32<DyT>(dyr[c]) * gamma[c];
        s1 += g0;
        s2 += g0 * (xr[c] - mu);
      }
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * rs * invD;  // mean(g * xhat)
    float* dxr = dx + row * D;
    if (VEC) {
      for (int c = lane * 4; c < D; c += 256) {
        float4 d = load4<DyT>(dyr + c);
        float4 v = *reinterpret_cast<const float4*>(xr + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        float xh[4] = {(v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs};
        float dd[4] = {d.x, d.y, d.z, d.w};
        float gg[4] = {g.x, g.y, g.z, g.w};
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (dres) {
          float4 t = *reinterpret_cast<const float4*>(dres + row * D + c);
          r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = rs * (dd[j] * gg[j] - s1 - xh[j] * s2) + r[j];
          atomicAdd(&s_dg[c + j], dd[j] * xh[j]);
          atomicAdd(&s_db[c + j], dd[j]);
          if (want_colsum) atomicAdd(&s_cs[c + j], o[j]);
        }
        *reinterpret_cast<float4*>(dxr + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (dx_lo) store4<bf16>(dx_lo + row * D + c, make_float4(o[0], o[1], o[2], o[3]));
      }
    } else {
      for (int c = lane; c < D; c += 64) {
        float d = to_f32<DyT>(dyr[c]);
        float xh = (xr[c] - mu) * rs;
        float o = rs * (d * gamma[c] - s1 - xh * s2) + (dres ? dres[row * D + c] : 0.f);
        atomicAdd(&s_dg[c], d * xh);
        atomicAdd(&s_db[c], d);
        if (want_colsum) atomicAdd(&s_cs[c], o);
        dxr[c] = o;
        if (dx_lo) dx_lo[row * D + c] = from_f32<bf16>(o);
      }
    }
  }
  __syncthreads();
  float* out = partial + (int64_t)blockIdx.x * 3 * D;
  for (int i = threadIdx.x; i < 3 * D; i += 256) out[i] = lds[i];
}

// Fast path (D % 4 == 0, D <= 256*NV): every lane owns the same NV float4 column chunks for all the rows
// its wave processes, so the per-column sums (dgamma, dbeta, colsum(dx)) accumulate in registers; the
// four waves of a block are combined through LDS with plain adds (deterministic), one partial per block.
constexpr int LNR_ROWS_PER_BLOCK = 16;

// ResT: storage type of the incoming residual gradient dres (fp32, or bf16 when the gradient stream is kept in bf16:
// then dx is null and dx_lo is the stream the next LayerNorm backward reads as ITS dres)
template <typename DyT, int NV, typename ResT = float, typename XT = float>
__global__ __launch_bounds__(256) void ln_bwd_reg_kernel(const DyT* __restrict__ dy, const XT* __restrict__ x,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const ResT* __restrict__ dres, float* __restrict__ dx,
                                                         bf16* __restrict__ dx_lo, float* __restrict__ partial,
                                                         int64_t rows, int D, int want_colsum, DropCfg drop,
                                                         uint8_t* __restrict__ dxq = nullptr,
                                                         uint8_t* __restrict__ dxs = nullptr,
                                                         int rpb = LNR_ROWS_PER_BLOCK) {
  // rpb: rows per workgroup (16; 4 - one row per wave - for short inputs, where 16-row blocks leave most CUs empty and the
  //      four rows of a wave run one after the other)
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][3][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * rpb;
  const float invD = 1.0f / (float)D;
  const uint64_t dkey = drop.thresh16 ? drop_key(drop) : 0;
  float4 g[NV], adg[NV], adb[NV], acs[NV];
  bool act[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    act[i] = c < D;
    g[i] = act[i] ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    adg[i] = adb[i] = acs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int rr = wave; rr < rpb; rr += 4) {
    const int64_t row = row0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float4 d[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + 256 * i;
      d[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (act[i]) {
        d[i] = load4<DyT>(dy + row * D + c);
        float4 v;
        if (sizeof(XT) == 4) {
          typedef float f32x4_nt __attribute__((ext_vector_type(4)));  // last use of this x row in the step: non-temporal
          const f32x4_nt xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(x + row * D + c));
          v = make_float4(xv[0], xv[1], xv[2], xv[3]);
        } else {
          typedef uint32_t u32x2_nt __attribute__((ext_vector_type(2)));
          const u32x2_nt xv = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(x + row * D + c));
          v = make_float4(__uint_as_float(xv[0] << 16), __uint_as_float(xv[0] & 0xffff0000u), __uint_as_float(xv[1] << 16),
                          __uint_as_float(xv[1] & 0xffff0000u));
        }
        xh[i] = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
      }
      const float g0 = d[i].x * g[i].x, g1 = d[i].y * g[i].y, g2 = d[i].z * g[i].z, g3 = d[i].w * g[i].w;
      s1 += (g0 + g1) + (g2 + g3);
      s2 += (g0 * xh[i].x + g1 * xh[i].y) + (g2 * xh[i].z + g3 * xh[i].w);
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;  // mean(g * xhat)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (!act[i]) continue;
      const int c = lane * 4 + 256 * i;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dres) r = load4<ResT>(dres + row * D + c);
      float4 o;
      o.x = rs * (d[i].x * g[i].x - s1 - xh[i].x * s2) + r.x;
      o.y = rs * (d[i].y * g[i].y - s1 - xh[i].y * s2) + r.y;
      o.z = rs * (d[i].z * g[i].z - s1 - xh[i].z * s2) + r.z;
      o.w = rs * (d[i].w * g[i].w - s1 - xh[i].w * s2) + r.w;
      if (dx) *reinterpret_cast<float4*>(dx + row * D + c) = o;
      if (drop.thresh16) {  // what the Linear behind the dropout site sees: masked, rescaled
        const float4 f = drop_factor4(drop, dkey, (uint64_t)row * D + c);
        o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
      }
      if (dx_lo) store4<bf16>(dx_lo + row * D + c, o);
      if (dxq) {  // wave-uniform: MX-FP8 image of the same values (D % 32 == 0: the 8 lanes of a block are live together)
        const float ov[4] = {o.x, o.y, o.z, o.w};
        uint32_t sb;
        const uint32_t qw = mx8_encode4(ov, &sb);
        *reinterpret_cast<uint32_t*>(dxq + row * D + c) = qw;
        if ((lane & 7) == 0) dxs[row * (D >> 5) + (c >> 5)] = (uint8_t)sb;
      }
      adg[i].x += d[i].x * xh[i].x; adg[i].y += d[i].y * xh[i].y; adg[i].z += d[i].z * xh[i].z; adg[i].w += d[i].w * xh[i].w;
      adb[i].x += d[i].x; adb[i].y += d[i].y; adb[i].z += d[i].z; adb[i].w += d[i].w;
      acs[i].x += o.x; acs[i].y += o.y; acs[i].z += o.z; acs[i].w += o.w;
    }
  }
  float* mine = lds + wave * 3 * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (!act[i]) continue;
    const int c = lane * 4 + 256 * i;
    *reinterpret_cast<float4*>(mine + c) = adg[i];
    *reinterpret_cast<float4*>(mine + D + c) = adb[i];
    *reinterpret_cast<float4*>(mine + 2 * D + c) = acs[i];
  }
  __syncthreads();
  float* out = partial + (int64_t)blockIdx.x * 3 * D;
  const int n = want_colsum ? 3 * D : 2 * D;
  for (int i = threadIdx.x; i < n; i += 256)
    out[i] = (lds[i] + lds[3 * D + i]) + (lds[6 * D + i] + lds[9 * D + i]);
}

// out[j] = sum_b partial[b][j], j in [0, width): 32 columns x 8 partial-groups per block

__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partial, int nb, int width,
                                                            float* __restrict__ o0, float* __restrict__ o1,
                                                            float* __restrict__ o2, int seg) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  float a0 = 0.f, a1 = 0.f;
  if (col < width) {
    int b = grp;
    for (; b + 8 < nb; b += 16) {
      a0 += partial[(int64_t)b * width + col];
      a1 += partial[(int64_t)(b + 8) * width + col];
    }
    if (b < nb) a0 += partial[(int64_t)b * width + col];
  }
  red[grp][cl] = a0 + a1;
  __syncthreads();
  if (grp == 0 && col < width) {
    float v = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
    const int which = col / seg, c = col - which * seg;
    float* dst = which == 0 ? o0 : (which == 1 ? o1 : o2);
    if (dst) dst[c] = v;
  }
}

// width % 4 == 0: FOLD_COLS columns x FOLD_RG row groups per block, 16-byte loads, 8 loads in flight (common.hpp)
__global__ __launch_bounds__(256) void fold_partials_vec_kernel(FoldJob job) {
  __shared__ float4 red[FOLD_RG][FOLD_COLS / 4];
  fold_columns_vec(job, blockIdx.x, red);
}

static int launch_fold(const float* partial, int nb, int width, float* o0, float* o1, float* o2, int seg, hipStream_t s) {
  if (width % 4 == 0 && (((uintptr_t)partial) & 15) == 0)
    fold_partials_vec_kernel<<<(unsigned)ceil_div(width, FOLD_COLS), 256, 0, s>>>(FoldJob{partial, nb, width, seg, o0, o1, o2});
  else
    fold_partials_kernel<<<(unsigned)ceil_div(width, 32), 256, 0, s>>>(partial, nb, width, o0, o1, o2, seg);
  return check_launch("fold_partials_kernel");
}

int fold_partials(const float* partial, int nb, int width, float* out, hipStream_t s) {
  return launch_fold(partial, nb, width, out, nullptr, nullptr, width, s);
}

int fold_job(const FoldJob& j, hipStream_t s) { return launch_fold(j.partial, j.nb, j.width, j.o0, j.o1, j.o2, j.seg, s); }

// every job of a list in ONE launch (the parity mode's layer backward: its LayerNorm column folds were a launch each - 19 tiny
// launches per C2 step); jobs with a null partial are skipped.  Falls back to one launch per job for unaligned / odd widths.
__global__ __launch_bounds__(256) void fold_list_kernel(FoldList fl, int g0, int g1, int g2) {
  __shared__ float4 red[FOLD_RG][FOLD_COLS / 4];
  const int b = blockIdx.x;
  const int j = b < g0 ? 0 : (b < g0 + g1 ? 1 : 2);
  const int local = b - (j == 0 ? 0 : (j == 1 ? g0 : g0 + g1));
  (void)g2;
  fold_columns_vec(fl.job[j], local, red);
}
int fold_list(const FoldList& fl, hipStream_t s) {
  int g[3] = {0, 0, 0};
  bool vec = true;
  for (int j = 0; j < 3; ++j) {
    const FoldJob& job = fl.job[j];
    if (j >= fl.count || !job.partial) continue;
    g[j] = (int)ceil_div(job.width, FOLD_COLS);
    vec = vec && job.width % 4 == 0 && (((uintptr_t)job.partial) & 15) == 0;
  }
  if (g[0] + g[1] + g[2] == 0) return 0;
  if (!vec) {
    for (int j = 0; j < fl.count && j < 3; ++j)
      if (fl.job[j].partial) AVF_TRY(fold_job(fl.job[j], s));
    return 0;
  }
  fold_list_kernel<<<(unsigned)(g[0] + g[1] + g[2]), 256, 0, s>>>(fl, g[0], g[1], g[2]);
  return check_launch("fold_list_kernel");
}

// LayerNorm backward on the all-bf16 streams (dy, x, the incoming residual gradient and dx in bf16; no dropout, no MX image):
// the row8 layout of ln_fwd_row8_kernel.  A workgroup owns rpb rows = 4 waves x (rpb / 4 / RU) batches of RU rows; the per-column
// sums stay in registers and are combined through LDS in wave order, so the partial of a block - and the folded result - is
// deterministic.  rpb (lnr8_rows_per_block): 32 from 8192 rows up - two batches per wave halve the partial rows written here and
// folded later and amortise the LDS combine (C2: 12.7 -> 11.6 us per launch and the layer's fold 12.0 -> 10.4; C3: 16.7 -> 14.2 and
// 15.4 -> 12.2; 48 rows level with 32 at C2 and worse at C3, 64 and 8 worse) - 16 below (short inputs need the workgroups).
// DROP (round 5): live dropout on the all-bf16 streams.  The row gradient leaves TWICE: dx_lo = the residual-gradient stream
// (never masked: the next LayerNorm backward's dres) and dx_m = what the Linear behind the dropout site sees (masked, rescaled:
// the GEMM operand; the column sums - that Linear's bias gradient - are those of the MASKED values, as ln_bwd_reg_kernel's).
template <int NV8, int RU, bool DROP = false>
__global__ __launch_bounds__(256) void ln_bwd_row8_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const bf16* __restrict__ dres,
                                                          bf16* __restrict__ dx_lo, float* __restrict__ partial, int64_t rows,
                                                          int D, int want_colsum, uint8_t* __restrict__ dxq = nullptr,
                                                          uint8_t* __restrict__ dxs = nullptr, float* __restrict__ dx = nullptr,
                                                          DropCfg drop = kNoDrop, bf16* __restrict__ dx_m = nullptr,
                                                          int rpb = LNR_ROWS_PER_BLOCK) {
  // dx (optional, wave-uniform): the fp32 copy of the row gradient (the bottom layer hands it to the caller)
  // rpb: rows per workgroup, a multiple of 4 RU (host: lnr8_rows_per_block) - a wave's rows come in whole batches of RU
  uint64_t dkey = 0;
  if constexpr (DROP) dkey = drop_key(drop);
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][3][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gm[NV8][8];
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int c = lane * 8 + 512 * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) gm[i][k] = 0.f;
    if (c < D) load8f(gamma + c, gm[i]);
  }
  float adg[NV8][8], adb[NV8][8], acs[NV8][8];
#pragma unroll
  for (int i = 0; i < NV8; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) adg[i][k] = adb[i][k] = acs[i][k] = 0.f;
  const float invD = 1.0f / (float)D;
#pragma unroll 1
  for (int batch = 0; batch < rpb / 4 / RU; ++batch) {
  const int64_t row0 = (int64_t)blockIdx.x * rpb + wave * (rpb / 4) + batch * RU;
  uint4 rd[RU][NV8], rx[RU][NV8], rr[RU][NV8];
  float mu[RU], rs[RU];
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int c = lane * 8 + 512 * i;
      const bool ok = c < D;
      rd[r][i] = ok ? *reinterpret_cast<const uint4*>(dy + row * D + c) : make_uint4(0u, 0u, 0u, 0u);
      typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));  // last use of this x row in the step: non-temporal
      u32x4_nt xv = {0u, 0u, 0u, 0u};
      if (ok) xv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(x + row * D + c));
      rx[r][i] = make_uint4(xv[0], xv[1], xv[2], xv[3]);
      rr[r][i] = (ok && dres) ? *reinterpret_cast<const uint4*>(dres + row * D + c) : make_uint4(0u, 0u, 0u, 0u);
    }
    mu[r] = mean[row];
    rs[r] = rstd[row];
  }
  float s1[RU], s2[RU];
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      float d[8], v[8];
      unpack8(rd[r][i], d);
      unpack8(rx[r][i], v);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float g = d[k] * gm[i][k];
        a += g;
        b = fmaf(g, (v[k] - mu[r]) * rs[r], b);
      }
    }
    s1[r] = a;
    s2[r] = b;
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    s1[r] = wave_sum(s1[r]) * invD;
    s2[r] = wave_sum(s2[r]) * invD;  // mean(g * xhat)
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    if (row0 + r >= rows) break;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int c = lane * 8 + 512 * i;
      if (c >= D) continue;
      float d[8], v[8], e[8], o[8];
      unpack8(rd[r][i], d);
      unpack8(rx[r][i], v);
      unpack8(rr[r][i], e);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (v[k] - mu[r]) * rs[r];
        o[k] = rs[r] * (d[k] * gm[i][k] - s1[r] - xh * s2[r]) + e[k];
        adg[i][k] = fmaf(d[k], xh, adg[i][k]);
        adb[i][k] += d[k];
        if constexpr (!DROP) acs[i][k] += o[k];
      }
      *reinterpret_cast<uint4*>(dx_lo + (row0 + r) * D + c) = pack8(o);
      if (dx) {
        *reinterpret_cast<float4*>(dx + (row0 + r) * D + c) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dx + (row0 + r) * D + c + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
      if constexpr (DROP) {  // the masked image (element index = row * D + column, as every other user of this site's mask)
        const uint64_t e0 = (uint64_t)(row0 + r) * D + c;
        const float4 f0 = drop_factor4(drop, dkey, e0), f1 = drop_factor4(drop, dkey, e0 + 4);
        o[0] *= f0.x; o[1] *= f0.y; o[2] *= f0.z; o[3] *= f0.w;
        o[4] *= f1.x; o[5] *= f1.y; o[6] *= f1.z; o[7] *= f1.w;
#pragma unroll
        for (int k = 0; k < 8; ++k) acs[i][k] += o[k];
        *reinterpret_cast<uint4*>(dx_m + (row0 + r) * D + c) = pack8(o);
      }
      if (dxq) {  // wave-uniform: MX-FP8 image of the same values (D % 32 == 0: the four lanes of a block are live together)
        const MxBlock mb = mx8_encode(o);
        *reinterpret_cast<uint2*>(dxq + (row0 + r) * D + c) = mb.q;
        if ((lane & 3) == 0) dxs[(row0 + r) * (D >> 5) + (c >> 5)] = (uint8_t)mb.scale;
      }
    }
  }
  }  // batch
  float* mine = lds + wave * 3 * D;
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int c = lane * 8 + 512 * i;
    if (c >= D) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(mine + c + 4 * h) = make_float4(adg[i][4 * h], adg[i][4 * h + 1], adg[i][4 * h + 2], adg[i][4 * h + 3]);
      *reinterpret_cast<float4*>(mine + D + c + 4 * h) = make_float4(adb[i][4 * h], adb[i][4 * h + 1], adb[i][4 * h + 2], adb[i][4 * h + 3]);
      *reinterpret_cast<float4*>(mine + 2 * D + c + 4 * h) = make_float4(acs[i][4 * h], acs[i][4 * h + 1], acs[i][4 * h + 2], acs[i][4 * h + 3]);
    }
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * 3 * D);
  const float4* l4 = reinterpret_cast<const float4*>(lds);
  const int n4 = (want_colsum ? 3 * D : 2 * D) >> 2, w4 = (3 * D) >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4 a = l4[i], b = l4[w4 + i], c = l4[2 * w4 + i], d = l4[3 * w4 + i];
    out[i] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
  }
}

constexpr int64_t LNR_SHORT_ROWS = 4096;  // up to here the register-path backward runs 4 rows per workgroup
static inline int lnr_rows_per_block(int64_t rows) { return rows <= LNR_SHORT_ROWS ? 4 : LNR_ROWS_PER_BLOCK; }
constexpr int64_t LNR8_BIG_ROWS = 8192;  // from here the row8 backward runs two batches per wave (see the kernel)
static inline int lnr8_rows_per_block(int64_t rows) { return rows >= LNR8_BIG_ROWS ? 2 * LNR_ROWS_PER_BLOCK : LNR_ROWS_PER_BLOCK; }
size_t layernorm_bwd_ws(int64_t rows, int dim) {
  return (size_t)ceil_div(rows, lnr_rows_per_block(rows)) * 3 * dim * sizeof(float);  // LNR < LNB: covers both paths
}

int layernorm_bwd(const void* dy, int dy_dtype, const void* xv, const float* gamma, const float* mean,
                  const float* rstd, const void* dres, float* dx, void* dx_lo, float* dgamma, float* dbeta,
                  float* dcolsum, void* ws, int64_t rows, int dim, hipStream_t s, const DropCfg& drop,
                  FoldJob* defer_fold, int dres_dtype, int x_dtype, void* mx_q, void* mx_s, void* dx_m) {
  // dx_m (with a live `drop`, all-bf16 streams only): dx_lo is then the UNMASKED stream and dx_m the masked image (row8 DROP)
  AVF_REQUIRE(!dx_m || (drop.thresh16 && dx_lo && dy_dtype == AVF_BF16 && x_dtype == AVF_BF16 && (!dres || dres_dtype == AVF_BF16) &&
                        dim % 8 == 0 && dim <= 1536 && !mx_q),
              "layernorm_bwd: a separate masked image needs live dropout on the all-bf16 streams (dim %% 8 == 0, no MX-FP8 image)");
  AVF_REQUIRE(rows > 0 && dim > 0 && ws, "layernorm_bwd: bad arguments");
  AVF_REQUIRE(!mx_q || (mx_s && dy_dtype == AVF_BF16 && dim % 32 == 0 && dim <= 1536 && ((uintptr_t)mx_q & 3) == 0),
              "layernorm_bwd: the MX-FP8 image of dx needs bf16 dy, dim %% 32 == 0 and dim <= 1536 (dim=%d)", dim);
  AVF_REQUIRE(x_dtype == AVF_F32 || (x_dtype == AVF_BF16 && dy_dtype == AVF_BF16 && dim % 4 == 0 && dim <= 1536),
              "layernorm_bwd: a bf16 LayerNorm input needs bf16 dy, dim %% 4 == 0 and dim <= 1536");
  const float* x = (const float*)xv;
  AVF_REQUIRE(dres_dtype == AVF_F32 || (dres_dtype == AVF_BF16 && dy_dtype == AVF_BF16 && dim % 4 == 0 && dim <= 1536 &&
                                         (!drop.thresh16 || dx_m) && dx_lo),
              "layernorm_bwd: a bf16 residual gradient needs bf16 dy, a bf16 output, dim %% 4 == 0, dim <= 1536, and with dropout a "
              "separate masked image");
  AVF_REQUIRE(dx || dx_lo, "layernorm_bwd: no output");
  AVF_REQUIRE(!drop.thresh16 || (dim % 4 == 0 && dim <= 1536), "layernorm_bwd: dropout needs dim %% 4 == 0 and dim <= 1536");
  AVF_REQUIRE((size_t)3 * dim * sizeof(float) <= 64 * 1024, "layernorm_bwd: dim %d too large", dim);
  TimingScope ts(KC_LAYERNORM, 0.0,
                 (double)rows * dim * ((dy_dtype == AVF_BF16 ? 2.0 : 4.0) + (x_dtype == AVF_BF16 ? 2.0 : 4.0) + (dres ? (dres_dtype == AVF_BF16 ? 2.0 : 4.0) : 0.0) +
                                       (dx ? 4.0 : 0.0) + (dx_lo ? 2.0 : 0.0)), s, /*per_kernel=*/true);
  float* partial = (float*)ws;
  const int wc = dcolsum ? 1 : 0;
  int nb;
  const bool fast = (dim % 4 == 0) && dim <= 1536 && (dy_dtype == AVF_F32 || dy_dtype == AVF_BF16);
  if (fast) {
    const bool row8 = x_dtype == AVF_BF16 && dy_dtype == AVF_BF16 && (!dres || dres_dtype == AVF_BF16) && dx_lo &&
                      (!drop.thresh16 || dx_m) && (!mx_q || dim % 32 == 0) && dim % 8 == 0 && (ln_row8_on() || dx_m);
    const int rpb = row8 ? lnr8_rows_per_block(rows) : lnr_rows_per_block(rows);
    nb = (int)ceil_div(rows, rpb);
    const size_t lds = (size_t)4 * 3 * dim * sizeof(float);
    const int nv = (dim + 255) / 256;
    if (lds > 64 * 1024) {  // only the NV=6 instantiations (D up to 1536) can exceed the default dynamic-LDS limit
      static PerDeviceOnce raised;
      if (raised.need()) {
        hipError_t e1 = hipFuncSetAttribute((const void*)ln_bwd_reg_kernel<float, 6>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
        hipError_t e2 = hipFuncSetAttribute((const void*)ln_bwd_reg_kernel<bf16, 6>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
        hipError_t e3 = hipFuncSetAttribute((const void*)ln_bwd_reg_kernel<bf16, 6, bf16>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
        hipError_t e4 = hipFuncSetAttribute((const void*)ln_bwd_reg_kernel<bf16, 6, bf16, bf16>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
        hipError_t e5 = hipFuncSetAttribute((const void*)ln_bwd_reg_kernel<bf16, 6, float, bf16>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
        AVF_REQUIRE(e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess && e5 == hipSuccess,
                    "layernorm_bwd: cannot raise dynamic LDS limit");
        raised.mark();
      }
    }
    if (row8) {
      if (lds > 64 * 1024) {
        static PerDeviceOnce raised8;
        if (raised8.need()) {
          hipError_t e1 = hipFuncSetAttribute((const void*)ln_bwd_row8_kernel<3, 1>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
          hipError_t e2 = hipFuncSetAttribute((const void*)ln_bwd_row8_kernel<3, 1, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 1536 * 4);
          AVF_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "layernorm_bwd: cannot raise dynamic LDS limit");
          raised8.mark();
        }
      }
#define LAUNCH_R8(NVV, RU)                                                                                                   \
  do {                                                                                                                       \
    if (dx_m)                                                                                                                \
      launch_in_scope(&ts, ln_bwd_row8_kernel<NVV, RU, true>, dim3(nb), dim3(256), (uint32_t)lds, s, (const bf16*)dy,        \
                      (const bf16*)xv, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx_lo, partial, rows, dim, wc,            \
                      (uint8_t*)nullptr, (uint8_t*)nullptr, dx, drop, (bf16*)dx_m, rpb);                                     \
    else                                                                                                                     \
      launch_in_scope(&ts, ln_bwd_row8_kernel<NVV, RU>, dim3(nb), dim3(256), (uint32_t)lds, s, (const bf16*)dy,              \
                      (const bf16*)xv, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx_lo, partial, rows, dim, wc,            \
                      (uint8_t*)mx_q, (uint8_t*)mx_s, dx, kNoDrop, (bf16*)nullptr, rpb);                                     \
  } while (0)
      switch ((dim + 511) / 512) {  // rows in flight per wave: what the register file allows at two waves per SIMD or more
        case 1: LAUNCH_R8(1, 4); break;
        case 2: LAUNCH_R8(2, 2); break;
        default: LAUNCH_R8(3, 1); break;
      }
#undef LAUNCH_R8
      AVF_TRY(check_launch("ln_bwd_row8_kernel"));
    } else {
#define LAUNCH_NV(T, NVV)                                                                                                   \
  do {                                                                                                                      \
    if (x_dtype == AVF_BF16 && dres_dtype == AVF_BF16)                                                                      \
      launch_in_scope(&ts, ln_bwd_reg_kernel<bf16, NVV, bf16, bf16>, dim3(nb), dim3(256), (uint32_t)lds, s, (const bf16*)dy,   \
                      (const bf16*)xv, gamma, mean, rstd, (const bf16*)dres, dx, (bf16*)dx_lo, partial, rows, dim, wc, drop, (uint8_t*)mx_q, (uint8_t*)mx_s, rpb); \
    else if (x_dtype == AVF_BF16)                                                                                           \
      launch_in_scope(&ts, ln_bwd_reg_kernel<bf16, NVV, float, bf16>, dim3(nb), dim3(256), (uint32_t)lds, s, (const bf16*)dy,  \
                      (const bf16*)xv, gamma, mean, rstd, (const float*)dres, dx, (bf16*)dx_lo, partial, rows, dim, wc, drop, (uint8_t*)mx_q, (uint8_t*)mx_s, rpb); \
    else if (dres_dtype == AVF_BF16)                                                                                        \
      launch_in_scope(&ts, ln_bwd_reg_kernel<bf16, NVV, bf16>, dim3(nb), dim3(256), (uint32_t)lds, s, (const bf16*)dy, x, gamma, \
                      mean, rstd, (const bf16*)dres, dx, (bf16*)dx_lo, partial, rows, dim, wc, drop, (uint8_t*)mx_q,         \
                      (uint8_t*)mx_s, rpb);                                                                                 \
    else                                                                                                                    \
      launch_in_scope(&ts, ln_bwd_reg_kernel<T, NVV, float>, dim3(nb), dim3(256), (uint32_t)lds, s, (const T*)dy, x, gamma, mean, \
                      rstd, (const float*)dres, dx, (bf16*)dx_lo, partial, rows, dim, wc, drop, (uint8_t*)mx_q,              \
                      (uint8_t*)mx_s, rpb);                                                                                 \
  } while (0)
#define LAUNCH_T(T)                                   \
  switch (nv) {                                       \
    case 1: LAUNCH_NV(T, 1); break;                   \
    case 2: LAUNCH_NV(T, 2); break;                   \
    case 3: LAUNCH_NV(T, 3); break;                   \
    case 4: LAUNCH_NV(T, 4); break;                   \
    default: LAUNCH_NV(T, 6); break;                  \
  }
    if (dy_dtype == AVF_F32) { LAUNCH_T(float) } else { LAUNCH_T(bf16) }
#undef LAUNCH_T
#undef LAUNCH_NV
    AVF_TRY(check_launch("ln_bwd_reg_kernel"));
    }
  } else {
    nb = (int)ceil_div(rows, LNB_ROWS_PER_BLOCK);
    const size_t lds = (size_t)3 * dim * sizeof(float);
    const bool vec = (dim & 3) == 0;
#define LAUNCH(T, V)                                                                                            \
  launch_in_scope(&ts, ln_bwd_kernel<T, V>, dim3(nb), dim3(256), (uint32_t)lds, s, (const T*)dy, x, gamma, mean, rstd, (const float*)dres, \
                  dx, (bf16*)dx_lo, partial, rows, dim, wc)
    if (dy_dtype == AVF_F32) {
      if (vec) LAUNCH(float, true); else LAUNCH(float, false);
    } else if (dy_dtype == AVF_BF16) {
      if (vec) LAUNCH(bf16, true); else LAUNCH(bf16, false);
    } else {
      AVF_REQUIRE(false, "layernorm_bwd: bad dtype %d", dy_dtype);
    }
#undef LAUNCH
    AVF_TRY(check_launch("ln_bwd_kernel"));
  }
  const int width = 3 * dim;
  if (defer_fold) {
    AVF_REQUIRE(width % 4 == 0 && (((uintptr_t)partial) & 15) == 0, "layernorm_bwd: deferred fold needs dim %% 4 == 0");
    *defer_fold = FoldJob{partial, nb, width, dim, dgamma, dbeta, dcolsum};
    return 0;
  }
  return launch_fold(partial, nb, width, dgamma, dbeta, dcolsum, dim, s);
}

// =============================================================================================
// column sums: out[c] = sum_r in[r, c]  (bias gradients - "db = sum_rows dY")
// =============================================================================================
constexpr int CS_ROW_CHUNKS = 128;

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, int64_t rows, int cols, int64_t ld,
                                                     float* __restrict__ partial) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int64_t per = ceil_div(rows, (int64_t)gridDim.y);
  const int64_t r0 = (int64_t)blockIdx.y * per;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;
  if (col >= cols) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += to_f32<T>(in[(r + 0) * ld + col]);
    a1 += to_f32<T>(in[(r + 1) * ld + col]);
    a2 += to_f32<T>(in[(r + 2) * ld + col]);
    a3 += to_f32<T>(in[(r + 3) * ld + col]);
  }
  for (; r < r1; ++r) a0 += to_f32<T>(in[r * ld + col]);
  partial[(int64_t)blockIdx.y * cols + col] = (a0 + a1) + (a2 + a3);
}

static int colsum_chunks(int64_t rows) {
  int64_t c = ceil_div(rows, 64);
  if (c > CS_ROW_CHUNKS) c = CS_ROW_CHUNKS;
  if (c < 1) c = 1;
  return (int)c;
}
size_t colsum_ws(int64_t rows, int cols) { return (size_t)colsum_chunks(rows) * cols * sizeof(float); }

int colsum(const void* in, int in_dtype, int64_t rows, int cols, int64_t ld, float* out, void* ws, hipStream_t s) {
  AVF_REQUIRE(rows > 0 && cols > 0 && ws && out, "colsum: bad arguments");
  const int ch = colsum_chunks(rows);
  dim3 grid((unsigned)ceil_div(cols, 256), ch);
  // one row chunk (up to 64 rows - the batch sum behind d pos_embedding: 32 rows x 165 888 columns at C2): the kernel's "partial"
  // row IS the result, written straight to `out`; the fold of one row was a 10 us copy launch
  float* dst = ch == 1 ? out : (float*)ws;
  if (in_dtype == AVF_F32)
    colsum_kernel<float><<<grid, 256, 0, s>>>((const float*)in, rows, cols, ld, dst);
  else if (in_dtype == AVF_BF16)
    colsum_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)in, rows, cols, ld, dst);
  else
    AVF_REQUIRE(false, "colsum: bad dtype %d", in_dtype);
  AVF_TRY(check_launch("colsum_kernel"));
  if (ch == 1) return 0;
  return launch_fold((const float*)ws, ch, cols, out, nullptr, nullptr, cols, s);
}

// =============================================================================================
// casts / weight preparation
// =============================================================================================
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out,
                                                        int64_t n, DropCfg drop) {
  const uint64_t dkey = drop.thresh16 ? drop_key(drop) : 0;
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(in + i);
    if (drop.thresh16) {
      const float4 f = drop_factor4(drop, dkey, (uint64_t)i);
      v.x *= f.x; v.y *= f.y; v.z *= f.z; v.w *= f.w;
    }
    store4<bf16>(out + i, v);
  }
  // tail (n not a multiple of 4): handled by the thread whose i lands on it
  if (i < n && i + 3 >= n)
    for (int64_t j = i; j < n; ++j) out[j] = from_f32<bf16>(in[j]);
}

__global__ __launch_bounds__(256) void dropout_factors_kernel(DropCfg drop, float* __restrict__ out, int64_t n) {
  const uint64_t dkey = drop_key(drop);
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) *reinterpret_cast<float4*>(out + i) = drop_factor4(drop, dkey, (uint64_t)i);
}
int dropout_factors(const DropCfg& drop, float* out, int64_t n, hipStream_t s) {
  AVF_REQUIRE(n > 0 && n % 4 == 0 && drop.thresh16, "dropout_factors: n %% 4 == 0 and p > 0 required");
  int64_t blocks = ceil_div(n, 1024);
  if (blocks > 2048) blocks = 2048;
  dropout_factors_kernel<<<(unsigned)blocks, 256, 0, s>>>(drop, out, n);
  return check_launch("dropout_factors_kernel");
}

__global__ __launch_bounds__(256) void mask_copy_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                            DropCfg drop) {
  const uint64_t dkey = drop_key(drop);
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(in + i);
    const float4 f = drop_factor4(drop, dkey, (uint64_t)i);
    *reinterpret_cast<float4*>(out + i) = make_float4(v.x * f.x, v.y * f.y, v.z * f.z, v.w * f.w);
  }
}
int mask_copy_f32(const float* in, float* out, int64_t n, hipStream_t s, const DropCfg& drop) {
  AVF_REQUIRE(n > 0 && n % 4 == 0 && drop.thresh16, "mask_copy_f32: n %% 4 == 0 and p > 0 required");
  AVF_REQUIRE((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "mask_copy_f32: pointers must be 16B aligned");
  int64_t blocks = ceil_div(n, 1024);
  if (blocks > 2048) blocks = 2048;
  mask_copy_f32_kernel<<<(unsigned)blocks, 256, 0, s>>>(in, out, n, drop);
  return check_launch("mask_copy_f32_kernel");
}

int cast_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t s, const DropCfg& drop) {
  AVF_REQUIRE(n > 0, "cast: n must be positive");
  AVF_REQUIRE(!drop.thresh16 || n % 4 == 0, "cast: dropout needs n %% 4 == 0");
  AVF_REQUIRE((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "cast: pointers must be 16B/8B aligned");
  int64_t blocks = ceil_div(n, 1024);
  if (blocks > 2048) blocks = 2048;
  cast_bf16_kernel<<<(unsigned)blocks, 256, 0, s>>>(in, (bf16*)out, n, drop);
  return check_launch("cast_bf16_kernel");
}

// w [R,C] fp32 -> w_lo [R,C] bf16 and w_t_lo [C,R] bf16, 32x32 tiles through LDS
__global__ __launch_bounds__(256) void prep_weight_kernel(const float* __restrict__ w, bf16* __restrict__ w_lo,
                                                          bf16* __restrict__ w_t, int R, int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = (r < R && c < C) ? w[(int64_t)r * C + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (w_lo && r < R && c < C) w_lo[(int64_t)r * C + c] = from_f32<bf16>(v);
  }
  __syncthreads();
  if (w_t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int c = c0 + ty + 8 * i, r = r0 + tx;  // output row = c, output col = r
      if (c < C && r < R) w_t[(int64_t)c * R + r] = from_f32<bf16>(tile[tx][ty + 8 * i]);
    }
  }
}

// up to 4 weights in one launch (blockIdx.z selects the matrix)
__global__ __launch_bounds__(256) void prep_weights_multi_kernel(PrepBatch b) {
  __shared__ float tile[32][33];
  const PrepDesc d = b.d[blockIdx.z];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  if (c0 >= d.C || r0 >= d.R) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = (r < d.R && c < d.C) ? d.w[(int64_t)r * d.C + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (r < d.R && c < d.C) {
      const bf16 q = from_f32<bf16>(r < d.lo_scaled_rows ? v * d.lo_scale : v);
      d.lo[(int64_t)r * d.C + c] = q;
      // the fragment-major image of the same values (round 5: five pack_ws launches per layer before - 58 us of a 555 us
      // TFormer step that prepares its weights in every forward)
      if (d.lo_p) *reinterpret_cast<bf16*>(reinterpret_cast<char*>(d.lo_p) + pack_ws_off(r, c)) = q;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < d.C && r < d.R) {
      const bf16 q = from_f32<bf16>(tile[tx][ty + 8 * i]);
      d.t[(int64_t)c * d.R + r] = q;
      if (d.t_p) *reinterpret_cast<bf16*>(reinterpret_cast<char*>(d.t_p) + pack_ws_off(c, r)) = q;
    }
  }
}

int prep_weights_multi(const PrepBatch& b, int count, hipStream_t s) {
  AVF_REQUIRE(count >= 1 && count <= 4, "prep_weights_multi: count must be 1..4");
  int mr = 0, mc = 0;
  for (int i = 0; i < count; ++i) {
    AVF_REQUIRE(b.d[i].w && b.d[i].lo && b.d[i].t && b.d[i].R > 0 && b.d[i].C > 0, "prep_weights_multi: bad descriptor");
    mr = b.d[i].R > mr ? b.d[i].R : mr;
    mc = b.d[i].C > mc ? b.d[i].C : mc;
  }
  dim3 grid((unsigned)ceil_div(mc, 32), (unsigned)ceil_div(mr, 32), (unsigned)count);
  prep_weights_multi_kernel<<<grid, 256, 0, s>>>(b);
  return check_launch("prep_weights_multi_kernel");
}

int prep_weight_bf16(const float* w, void* w_lo, void* w_t_lo, int rows, int cols, hipStream_t s) {
  AVF_REQUIRE(rows > 0 && cols > 0, "prep_weight: bad shape");
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  prep_weight_kernel<<<grid, 256, 0, s>>>(w, (bf16*)w_lo, (bf16*)w_t_lo, rows, cols);
  return check_launch("prep_weight_kernel");
}

// =============================================================================================
// AULoss (models/loss.py:75-103): one block; rows are few (a batch of clips).
//   keep_b = (y[b,0] != ignore);  l = (1-y) z + (1 + (w-1) y) softplus(-z);  loss = mean over kept
//   dl/dz = sigmoid(z) (1 - y + w y) - w y, scaled by 1/(ncls * kept), 0 for dropped rows.
// =============================================================================================
__global__ __launch_bounds__(256) void au_loss_kernel(const float* __restrict__ z, int64_t ldz,
                                                      const float* __restrict__ y, int64_t ldy,
                                                      const float* __restrict__ pw, float ignore, int rows, int ncls,
                                                      float* __restrict__ loss, float* __restrict__ grad, int sum_mode,
                                                      int gwidth) {
  // gwidth >= ncls: rows of `grad` are gwidth wide, columns ncls .. gwidth-1 written as zeros (the gradient of the reference's
  // [B,21] output row whose slots 0..11 are the AU logits: what autograd's slice backward would build with a fill and a copy)
  __shared__ float red[4];
  __shared__ int redc[4];
  float acc = 0.f;
  int kept = 0;
  for (int r = threadIdx.x; r < rows; r += 256) kept += (y[(int64_t)r * ldy] != ignore) ? 1 : 0;
  for (int i = threadIdx.x; i < rows * ncls; i += 256) {
    const int r = i / ncls, c = i - r * ncls;
    if (y[(int64_t)r * ldy] != ignore) {
      const float zz = z[(int64_t)r * ldz + c], yy = y[(int64_t)r * ldy + c], w = pw[c];
      // softplus(-z) = max(-z,0) + log1p(exp(-|z|))
      const float sp = fmaxf(-zz, 0.f) + log1pf(expf(-fabsf(zz)));
      acc += (1.f - yy) * zz + (1.f + (w - 1.f) * yy) * sp;
    }
  }
  acc = wave_sum(acc);
  for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o, 64);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = acc;
    redc[threadIdx.x >> 6] = kept;
  }
  __syncthreads();
  const float total = (red[0] + red[1]) + (red[2] + red[3]);
  const int nk = redc[0] + redc[1] + redc[2] + redc[3];
  // sum_mode (data-parallel shards): loss[0] = sum over kept rows of the row mean, loss[1] = kept rows; the caller
  // divides by the GLOBAL kept count after reducing both over the ranks (loss.py:85-102 is a ratio)
  const float denom = (sum_mode ? 1.0f : (float)nk) * (float)ncls;
  if (threadIdx.x == 0) {
    loss[0] = total / denom;  // 0/0 -> NaN when every row is ignored (as the reference)
    if (sum_mode) loss[1] = (float)nk;
  }
  const float inv = 1.0f / denom;
  for (int i = threadIdx.x; i < rows * gwidth; i += 256) {
    const int r = i / gwidth, c = i - r * gwidth;
    float g = 0.f;
    if (c < ncls && y[(int64_t)r * ldy] != ignore) {
      const float zz = z[(int64_t)r * ldz + c], yy = y[(int64_t)r * ldy + c], w = pw[c];
      const float sg = 1.0f / (1.0f + expf(-zz));
      g = (sg * (1.f - yy + w * yy) - w * yy) * inv;
    }
    grad[i] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// Token-sequence plumbing of the callers either side of the stack (the fused [B, T_v + T_a, dim] sequence of
// BASELINE.json's configs; reference fusion: models/avformer.py:95-103, mean pooling: models/tformer.py head):
// sequence fusion + positional embedding in one pass, token-mean pooling, and its backward, which writes the top
// layer's incoming gradient in fp32 AND bf16 and its column sums analytically (= sum_b g[b,:]).
// ---------------------------------------------------------------------------------------------
constexpr int TOK_ROWS_PER_BLOCK = 4;

// one workgroup per TOK_ROWS_PER_BLOCK token rows (row = b * T + t): no per-element index division
template <typename OutT>  // float: float4 stores; bf16: the bf16 residual stream, 8-byte stores
__global__ __launch_bounds__(256) void fuse_tokens_kernel(const float4* __restrict__ clip, const float4* __restrict__ audio,
                                                         const float4* __restrict__ pos, OutT* __restrict__ out, int Tv,
                                                         int Ta, int D4, int64_t rows) {
  // the block's TOK_ROWS_PER_BLOCK x D4 quads are dealt to the 256 threads as one flat range (D4 = 128: two rows at a time,
  // every thread busy; one row per pass left half of them idle)
  const int T = Tv + Ta;
  const int64_t row0 = (int64_t)blockIdx.x * TOK_ROWS_PER_BLOCK;
  if constexpr (sizeof(OutT) == 2) {
    if ((D4 & 1) == 0 && ((uintptr_t)out & 15) == 0) {  // bf16 rows of a multiple of 8 columns: two quads in, ONE 16-byte store out (8-byte stores: 1.1 TB/s)
      const int D8 = D4 >> 1;
      for (int i = threadIdx.x; i < TOK_ROWS_PER_BLOCK * D8; i += 256) {
        const int rr = i / D8, c = i - rr * D8;
        const int64_t row = row0 + rr;
        if (row >= rows) return;  // (rr grows with i)
        const int64_t b = row / T;
        const int t = (int)(row - b * T);
        const float4* src = t < Tv ? clip + (b * Tv + t) * D4 : audio + (b * Ta + (t - Tv)) * D4;
        float4 v0 = src[2 * c], v1 = src[2 * c + 1];
        if (pos) {
          const float4 p0 = pos[(int64_t)t * D4 + 2 * c], p1 = pos[(int64_t)t * D4 + 2 * c + 1];
          v0.x += p0.x; v0.y += p0.y; v0.z += p0.z; v0.w += p0.w;
          v1.x += p1.x; v1.y += p1.y; v1.z += p1.z; v1.w += p1.w;
        }
        const float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        *reinterpret_cast<uint4*>(out + (row * D4 + 2 * c) * 4) = pack8(o);
      }
      return;
    }
  }
  for (int i = threadIdx.x; i < TOK_ROWS_PER_BLOCK * D4; i += 256) {
    const int rr = i / D4, c = i - rr * D4;
    const int64_t row = row0 + rr;
    if (row >= rows) return;  // (rr grows with i)
    const int64_t b = row / T;
    const int t = (int)(row - b * T);
    const float4* src = t < Tv ? clip + (b * Tv + t) * D4 : audio + (b * Ta + (t - Tv)) * D4;
    float4 v = src[c];
    if (pos) {
      const float4 p = pos[(int64_t)t * D4 + c];
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    store4<OutT>(out + (row * D4 + c) * 4, v);
  }
}

// grid (ceil(D4/16), B); 256 threads = 16 float4 column groups x 16 token lanes
template <typename InT>
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const InT* __restrict__ y, float4* __restrict__ out, int T,
                                                            int D4) {
  __shared__ float4 red[16][16];
  const int cg = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cg;
  const int64_t b = blockIdx.y;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < D4) {
    const InT* base = y + (b * T * D4 + c) * 4;
    const int64_t ld = (int64_t)D4 * 4;
    int t = tl;
    for (; t + 48 < T; t += 64) {  // four independent loads in flight
      const float4 v0 = load4<InT>(base + t * ld), v1 = load4<InT>(base + (t + 16) * ld), v2 = load4<InT>(base + (t + 32) * ld),
                   v3 = load4<InT>(base + (t + 48) * ld);
      acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
      acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; t < T; t += 16) {
      const float4 v = load4<InT>(base + t * ld);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[tl][cg] = acc;
  __syncthreads();
  if (tl == 0 && c < D4) {
    float4 r = red[0][cg];
#pragma unroll
    for (int j = 1; j < 16; ++j) { r.x += red[j][cg].x; r.y += red[j][cg].y; r.z += red[j][cg].z; r.w += red[j][cg].w; }
    const float inv = 1.0f / (float)T;
    out[b * D4 + c] = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
  }
}

// dy[b,t,:] = g[b,:] / T (fp32, and bf16 if dy_lo); the last blocks of the grid write colsum[d] = sum_b g[b,d]
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dy,
                                                            bf16* __restrict__ dy_lo, float* __restrict__ colsum, int B,
                                                            int T, int D4, int64_t rows, int main_blocks) {
  // the column-sum blocks come FIRST in the grid (they are one dependent chain of B rows each and would otherwise be the tail of
  // the launch), eight loads in flight; same summation order as before
  const int extra = (int)gridDim.x - main_blocks;
  if ((int)blockIdx.x < extra) {
    const int d = (int)blockIdx.x * 256 + threadIdx.x;
    if (d < 4 * D4) {
      const float* gs = reinterpret_cast<const float*>(g);
      float a = 0.f;
      int b = 0;
      for (; b + 8 <= B; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gs[(int64_t)(b + u) * 4 * D4 + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
      for (; b < B; ++b) a += gs[(int64_t)b * 4 * D4 + d];
      colsum[d] = a;
    }
    return;
  }
  const float inv = 1.0f / (float)T;
  const int64_t row0 = (int64_t)((int)blockIdx.x - extra) * TOK_ROWS_PER_BLOCK;
  if (!dy && (D4 & 1) == 0 && ((uintptr_t)dy_lo & 15) == 0) {  // the bf16 image alone (the bf16 gradient stream): 16-byte stores
    const int D8 = D4 >> 1;
    for (int i = threadIdx.x; i < TOK_ROWS_PER_BLOCK * D8; i += 256) {
      const int rr = i / D8, c = i - rr * D8;
      const int64_t row = row0 + rr;
      if (row >= rows) return;
      const float4 v0 = g[(row / T) * D4 + 2 * c], v1 = g[(row / T) * D4 + 2 * c + 1];
      const float o[8] = {v0.x * inv, v0.y * inv, v0.z * inv, v0.w * inv, v1.x * inv, v1.y * inv, v1.z * inv, v1.w * inv};
      *reinterpret_cast<uint4*>(dy_lo + 4 * (row * D4 + 2 * c)) = pack8(o);
    }
    return;
  }
  for (int i = threadIdx.x; i < TOK_ROWS_PER_BLOCK * D4; i += 256) {  // (flat range, as fuse_tokens_kernel)
    const int rr = i / D4, c = i - rr * D4;
    const int64_t row = row0 + rr;
    if (row >= rows) return;
    const float4 v = g[(row / T) * D4 + c];
    const float4 r = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    if (dy) dy[row * D4 + c] = r;
    if (dy_lo) store4<bf16>(dy_lo + 4 * (row * D4 + c), r);
  }
}

}  // namespace avf

extern "C" int avf_fuse_tokens(const float* clip, const float* audio, const float* pos, float* out, int batch, int t_video,
                               int t_audio, int dim, void* stream) {
  using namespace avf;
  AVF_REQUIRE(out && batch > 0 && t_video >= 0 && t_audio >= 0 && t_video + t_audio > 0 && (clip || t_video == 0) &&
                  (audio || t_audio == 0) && dim > 0 && dim % 4 == 0,
              "fuse_tokens: bad arguments (dim must be a multiple of 4)");
  AVF_REQUIRE((((uintptr_t)clip | (uintptr_t)audio | (uintptr_t)pos | (uintptr_t)out) & 15) == 0,
              "fuse_tokens: pointers must be 16-byte aligned");
  const int64_t rows = (int64_t)batch * (t_video + t_audio);
  AVF_REQUIRE(ceil_div(rows, TOK_ROWS_PER_BLOCK) < (1LL << 31), "fuse_tokens: too many rows");
  fuse_tokens_kernel<float><<<(unsigned)ceil_div(rows, TOK_ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(
      (const float4*)clip, (const float4*)audio, (const float4*)pos, out, t_video, t_audio, dim / 4, rows);
  return check_launch("fuse_tokens_kernel");
}

extern "C" int avf_fuse_tokens_bf16(const float* clip, const float* audio, const float* pos, void* out_bf16, int batch,
                                    int t_video, int t_audio, int dim, void* stream) {
  using namespace avf;
  AVF_REQUIRE(out_bf16 && batch > 0 && t_video >= 0 && t_audio >= 0 && t_video + t_audio > 0 && (clip || t_video == 0) &&
                  (audio || t_audio == 0) && dim > 0 && dim % 4 == 0,
              "fuse_tokens_bf16: bad arguments (dim must be a multiple of 4)");
  AVF_REQUIRE((((uintptr_t)clip | (uintptr_t)audio | (uintptr_t)pos) & 15) == 0 && ((uintptr_t)out_bf16 & 7) == 0,
              "fuse_tokens_bf16: misaligned pointers");
  const int64_t rows = (int64_t)batch * (t_video + t_audio);
  AVF_REQUIRE(ceil_div(rows, TOK_ROWS_PER_BLOCK) < (1LL << 31), "fuse_tokens_bf16: too many rows");
  fuse_tokens_kernel<bf16><<<(unsigned)ceil_div(rows, TOK_ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(
      (const float4*)clip, (const float4*)audio, (const float4*)pos, (bf16*)out_bf16, t_video, t_audio, dim / 4, rows);
  return check_launch("fuse_tokens_kernel");
}

extern "C" int avf_token_mean_fwd(const float* y, float* out, int batch, int tokens, int dim, void* stream) {
  using namespace avf;
  AVF_REQUIRE(y && out && batch > 0 && batch <= 65535 && tokens > 0 && dim > 0 && dim % 4 == 0,
              "token_mean_fwd: bad arguments (dim must be a multiple of 4)");
  AVF_REQUIRE((((uintptr_t)y | (uintptr_t)out) & 15) == 0, "token_mean_fwd: pointers must be 16-byte aligned");
  const int D4 = dim / 4;
  token_mean_fwd_kernel<float><<<dim3((D4 + 15) / 16, batch), 256, 0, (hipStream_t)stream>>>(y, (float4*)out, tokens, D4);
  return check_launch("token_mean_fwd_kernel");
}

extern "C" int avf_token_mean_fwd_bf16(const void* y_bf16, float* out, int batch, int tokens, int dim, void* stream) {
  using namespace avf;
  AVF_REQUIRE(y_bf16 && out && batch > 0 && batch <= 65535 && tokens > 0 && dim > 0 && dim % 4 == 0,
              "token_mean_fwd_bf16: bad arguments (dim must be a multiple of 4)");
  AVF_REQUIRE(((uintptr_t)y_bf16 & 7) == 0 && ((uintptr_t)out & 15) == 0, "token_mean_fwd_bf16: misaligned pointers");
  const int D4 = dim / 4;
  token_mean_fwd_kernel<bf16><<<dim3((D4 + 15) / 16, batch), 256, 0, (hipStream_t)stream>>>((const bf16*)y_bf16, (float4*)out,
                                                                                             tokens, D4);
  return check_launch("token_mean_fwd_kernel");
}

extern "C" int avf_token_mean_bwd(const float* g, float* dy, void* dy_bf16, float* colsum, int batch, int tokens, int dim,
                                  void* stream) {
  using namespace avf;
  AVF_REQUIRE(g && (dy || dy_bf16) && batch > 0 && tokens > 0 && dim > 0 && dim % 4 == 0,
              "token_mean_bwd: bad arguments (dim must be a multiple of 4; one of dy / dy_bf16 is required)");
  AVF_REQUIRE((((uintptr_t)g | (uintptr_t)dy) & 15) == 0 && ((uintptr_t)dy_bf16 & 7) == 0,
              "token_mean_bwd: misaligned pointers");
  const int D4 = dim / 4;
  const int64_t rows = (int64_t)batch * tokens;
  AVF_REQUIRE(ceil_div(rows, TOK_ROWS_PER_BLOCK) < (1LL << 30), "token_mean_bwd: too many rows");
  const int main_blocks = (int)ceil_div(rows, TOK_ROWS_PER_BLOCK);
  const int extra = colsum ? (dim + 255) / 256 : 0;
  token_mean_bwd_kernel<<<main_blocks + extra, 256, 0, (hipStream_t)stream>>>((const float4*)g, (float4*)dy, (bf16*)dy_bf16,
                                                                              colsum, batch, tokens, D4, rows, main_blocks);
  return check_launch("token_mean_bwd_kernel");
}

extern "C" int avf_au_loss(const float* logits, int64_t ld_logits, const float* labels, int64_t ld_labels,
                           const float* pos_weight, float ignore, int rows, int ncls, float* loss, float* grad_unit,
                           void* stream) {
  using namespace avf;
  AVF_REQUIRE(rows > 0 && ncls > 0 && logits && labels && pos_weight && loss && grad_unit, "au_loss: bad arguments");
  au_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, ld_logits, labels, ld_labels, pos_weight, ignore, rows,
                                                     ncls, loss, grad_unit, 0, ncls);
  return check_launch("au_loss_kernel");
}

extern "C" int avf_au_loss_sum(const float* logits, int64_t ld_logits, const float* labels, int64_t ld_labels,
                               const float* pos_weight, float ignore, int rows, int ncls, float* sum_count,
                               float* grad_unit, void* stream) {
  using namespace avf;
  AVF_REQUIRE(rows > 0 && ncls > 0 && logits && labels && pos_weight && sum_count && grad_unit, "au_loss_sum: bad arguments");
  au_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, ld_logits, labels, ld_labels, pos_weight, ignore, rows,
                                                     ncls, sum_count, grad_unit, 1, ncls);
  return check_launch("au_loss_kernel");
}

extern "C" int avf_au_loss_wide(const float* logits, int64_t ld_logits, const float* labels, int64_t ld_labels,
                                const float* pos_weight, float ignore, int rows, int ncls, int width, int sum_mode,
                                float* loss, float* grad_wide, void* stream) {
  using namespace avf;
  AVF_REQUIRE(rows > 0 && ncls > 0 && width >= ncls && logits && labels && pos_weight && loss && grad_wide,
              "au_loss_wide: bad arguments");
  au_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, ld_logits, labels, ld_labels, pos_weight, ignore, rows,
                                                     ncls, loss, grad_wide, sum_mode ? 1 : 0, width);
  return check_launch("au_loss_kernel");
}
