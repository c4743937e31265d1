// attn_f32.hip - parity-mode attention core (fp32 VALU, flash-style: never materialises [B,H,N,N]).
// (Round 5: fp32 storage without a mask at dim_head 32 / 64 - the parity mode's own calls - runs on attn_f32_mfma.hip instead;
//  these kernels keep the masked calls, bf16 storage and the other head widths.)
//
// Reference: models/heads.py:222-237 -  dots = q k^T * dh^-0.5 ; softmax(dim=-1) ; out = attn v ;
// 'b h n d -> b n (h d)'.  No dropout on the probabilities.  The token mask of heads.py:225-232 (dead in the reference: no
// caller passes one, but it is part of forward()'s signature) is built here: keep[b, n] (1 = token kept, the reference's
// mask padded with a leading True); a pair (i, j) with either token dropped scores -FLT_MAX, so a dropped query attends
// uniformly to ALL keys (every score equal) and a kept query gives dropped keys exactly zero weight; backward passes no
// gradient through a filled score (masked_fill_).  The kernels also serve the bf16 mode when a mask is given (T = bf16
// storage, fp32 arithmetic; QS: the q columns already carry log2(e)/sqrt(dh), layer.hip).
// One lane owns one query (forward, dQ) or one key (dK, dV); the opposite operand is staged in
// LDS and read as wave-wide broadcasts.  Scores are kept in the log2 domain:
//   s2 = (q . k) * dh^-0.5 * log2(e),  p = 2^(s2 - lse2),  lse2 = m2 + log2(sum 2^(s2 - m2)).
#include "common.hpp"

namespace avf {

namespace {

constexpr int KT = 32;  // keys (or queries) staged per step
constexpr float LOG2E = 1.4426950408889634f;

template <int DH, typename T>
__device__ __forceinline__ void stage_rows(float* dst, const T* src, int64_t ld, int row0, int nrows_valid) {
  // KT rows x DH floats, 64 threads, 4-element granules; rows >= nrows_valid are zero-filled
  constexpr int V = DH / 4;
  for (int i = threadIdx.x; i < KT * V; i += 64) {
    const int r = i / V, c = (i - r * V) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows_valid) v = load4<T>(src + (int64_t)(row0 + r) * ld + c);
    *reinterpret_cast<float4*>(dst + r * DH + c) = v;
  }
}
// keep flags of KT tokens (1 when no mask is given); tokens past the end are 0
__device__ __forceinline__ void stage_keep(uint8_t* dst, const uint8_t* keep, int64_t row0, int nvalid) {
  if (threadIdx.x < KT) dst[threadIdx.x] = (int)threadIdx.x < nvalid ? (keep ? keep[row0 + threadIdx.x] : 1) : 0;
}
constexpr float MASKV = -3.4028234663852886e38f;  // -finfo(float32).max, heads.py:225

template <int DH, typename T, bool MASKED>
__global__ __launch_bounds__(64) void attn_fwd_f32_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                          float* __restrict__ lse2, int /*B*/, int N, int H,
                                                          const uint8_t* __restrict__ keep, int qs) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * DH];
  __shared__ __attribute__((aligned(16))) float Vs[KT * DH];
  __shared__ uint8_t Ms[KT];
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const T* base = qkv + (int64_t)b * N * ld + h * DH;
  const int qi = blockIdx.x * 64 + threadIdx.x;
  const bool valid = qi < N;
  const bool mq = !MASKED || (valid && keep[(int64_t)b * N + qi]);
  const float c = qs ? 1.0f : LOG2E / sqrtf((float)DH);
  float q[DH], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = load4<T>(base + (int64_t)qi * ld + d);
    q[d] = v.x * c; q[d + 1] = v.y * c; q[d + 2] = v.z * c; q[d + 3] = v.w * c;
    acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int kt = 0; kt < N; kt += KT) {
    const int nk = (N - kt) < KT ? (N - kt) : KT;
    stage_rows<DH, T>(Ks, base + I, ld, kt, nk);
    stage_rows<DH, T>(Vs, base + 2 * I, ld, kt, nk);
    if (MASKED) stage_keep(Ms, keep, (int64_t)b * N + kt, nk);
    __syncthreads();
    float s[KT];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) a = fmaf(q[d], Ks[j * DH + d], a);
      if (MASKED) s[j] = (j < nk) ? ((mq && Ms[j]) ? a : MASKV) : -INFINITY;
      else s[j] = (j < nk) ? a : -INFINITY;
      tmax = fmaxf(tmax, s[j]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = exp2f(m - mn);
    l *= alpha;
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] *= alpha;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float pj = exp2f(s[j] - mn);
      l += pj;
#pragma unroll
      for (int d = 0; d < DH; ++d) acc[d] = fmaf(pj, Vs[j * DH + d], acc[d]);
    }
    m = mn;
    __syncthreads();
  }
  if (valid) {
    const float inv = 1.0f / l;
    T* orow = o + ((int64_t)b * N + qi) * I + h * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4)
      store4<T>(orow + d, make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv));
    lse2[(int64_t)bh * N + qi] = m + log2f(l);  // (a dropped query's row: backward uses p = 1/N, not this value)
  }
}

// dQ: one lane per query.  dS = P o (dP - delta),  dq = dS k * dh^-0.5; no gradient through a filled score
template <int DH, typename T, bool MASKED>
__global__ __launch_bounds__(64) void attn_dq_f32_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                         const float* __restrict__ lse2, const float* __restrict__ delta,
                                                         T* __restrict__ dqkv, int /*B*/, int N, int H,
                                                         const uint8_t* __restrict__ keep, int qs) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * DH];
  __shared__ __attribute__((aligned(16))) float Vs[KT * DH];
  __shared__ uint8_t Ms[KT];
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const T* base = qkv + (int64_t)b * N * ld + h * DH;
  const int qi = blockIdx.x * 64 + threadIdx.x;
  const bool valid = qi < N;
  const bool mq = !MASKED || (valid && keep[(int64_t)b * N + qi]);
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = qs ? 1.0f : LOG2E * scale;
  float q[DH], g[DH], dq[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
    if (valid) {
      v = load4<T>(base + (int64_t)qi * ld + d);
      w = load4<T>(d_o + ((int64_t)b * N + qi) * I + h * DH + d);
    }
    q[d] = v.x * c; q[d + 1] = v.y * c; q[d + 2] = v.z * c; q[d + 3] = v.w * c;
    g[d] = w.x; g[d + 1] = w.y; g[d + 2] = w.z; g[d + 3] = w.w;
    dq[d] = dq[d + 1] = dq[d + 2] = dq[d + 3] = 0.f;
  }
  const float L = valid ? lse2[(int64_t)bh * N + qi] : 0.f;
  const float dl = valid ? delta[(int64_t)bh * N + qi] : 0.f;
  for (int kt = 0; kt < N; kt += KT) {
    const int nk = (N - kt) < KT ? (N - kt) : KT;
    stage_rows<DH, T>(Ks, base + I, ld, kt, nk);
    stage_rows<DH, T>(Vs, base + 2 * I, ld, kt, nk);
    if (MASKED) stage_keep(Ms, keep, (int64_t)b * N + kt, nk);
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < KT; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        s = fmaf(q[d], Ks[j * DH + d], s);
        dp = fmaf(g[d], Vs[j * DH + d], dp);
      }
      const bool live = MASKED ? (j < nk && mq && Ms[j]) : (j < nk);  // filled scores: dS = 0
      const float pj = live ? exp2f(s - L) : 0.f;
      const float ds = pj * (dp - dl);
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, Ks[j * DH + d], dq[d]);
    }
    __syncthreads();
  }
  if (valid) {
    T* out = dqkv + ((int64_t)b * N + qi) * ld + h * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4)
      store4<T>(out + d, make_float4(dq[d] * scale, dq[d + 1] * scale, dq[d + 2] * scale, dq[d + 3] * scale));
  }
}

// dK / dV: one lane per key.  PASS 0: dv = P^T dO.  PASS 1: dk = dS^T q * dh^-0.5
// (a dropped query's row of P is the constant 1/N - dV receives it - and its dS is 0)
template <int DH, int PASS, typename T, bool MASKED>
__global__ __launch_bounds__(64) void attn_dkv_f32_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                          const float* __restrict__ lse2,
                                                          const float* __restrict__ delta, T* __restrict__ dqkv,
                                                          int /*B*/, int N, int H, const uint8_t* __restrict__ keep,
                                                          int qs) {
  __shared__ __attribute__((aligned(16))) float Qs[KT * DH];
  __shared__ __attribute__((aligned(16))) float Gs[KT * DH];
  __shared__ float Ls[KT], Ds[KT];
  __shared__ uint8_t Ms[KT];
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const T* base = qkv + (int64_t)b * N * ld + h * DH;
  const int ki = blockIdx.x * 64 + threadIdx.x;
  const bool valid = ki < N;
  const bool mk = !MASKED || (valid && keep[(int64_t)b * N + ki]);
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = qs ? 1.0f : LOG2E * scale;
  const float uniform = 1.0f / (float)N;
  float k[DH], v[PASS == 1 ? DH : 1], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) a = load4<T>(base + I + (int64_t)ki * ld + d);
    k[d] = a.x * c; k[d + 1] = a.y * c; k[d + 2] = a.z * c; k[d + 3] = a.w * c;
    if (PASS == 1) {
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) w = load4<T>(base + 2 * I + (int64_t)ki * ld + d);
      v[d] = w.x; v[d + 1] = w.y; v[d + 2] = w.z; v[d + 3] = w.w;
    }
    acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
  }
  for (int qt = 0; qt < N; qt += KT) {
    const int nq = (N - qt) < KT ? (N - qt) : KT;
    stage_rows<DH, T>(Qs, base, ld, qt, nq);
    stage_rows<DH, T>(Gs, d_o + (int64_t)b * N * I + h * DH, I, qt, nq);
    if (MASKED) stage_keep(Ms, keep, (int64_t)b * N + qt, nq);
    if (threadIdx.x < KT) {
      const bool ok = (int)threadIdx.x < nq;
      Ls[threadIdx.x] = ok ? lse2[(int64_t)bh * N + qt + threadIdx.x] : INFINITY;  // 2^(s - inf) = 0
      Ds[threadIdx.x] = ok ? delta[(int64_t)bh * N + qt + threadIdx.x] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < KT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(k[d], Qs[j * DH + d], s);
      const bool pair = !MASKED || (Ms[j] && mk);  // Ms[j] = 0 past the end and for a dropped query (Ls = inf past the end)
      float pj = pair ? exp2f(s - Ls[j]) : 0.f;
      if (MASKED && j < nq && !Ms[j]) pj = uniform;  // dropped query: every score was filled, softmax is uniform over all N keys
      if (PASS == 0) {
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = fmaf(pj, Gs[j * DH + d], acc[d]);
      } else {
        float dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) dp = fmaf(Gs[j * DH + d], v[d], dp);
        const float ds = pair ? pj * (dp - Ds[j]) : 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = fmaf(ds, Qs[j * DH + d], acc[d]);
      }
    }
    __syncthreads();
  }
  if (valid) {
    const float f = PASS == 1 ? (qs ? 1.0f / LOG2E : scale) : 1.0f;  // qs: q' = q log2(e) scale, dk = dS^T q' / log2(e)
    T* out = dqkv + ((int64_t)b * N + ki) * ld + (PASS == 1 ? I : 2 * I) + h * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4)
      store4<T>(out + d, make_float4(acc[d] * f, acc[d + 1] * f, acc[d + 2] * f, acc[d + 3] * f));
  }
}

// delta[b,h,n] = sum_d dO[b,n,h,d] * O[b,n,h,d]   (= rowsum(dP o P)).  Each lane takes one 16-byte granule of a
// token row (fully coalesced over [B*N, I]); the DH/VEC lanes of a head reduce with xor-shuffles.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o,
                                                         float* __restrict__ delta, int B, int N, int H, int DH) {
  const int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // granule index over [B*N*I / VEC]
  const int lph = DH / VEC;                                       // lanes per head (power of two, <= 64)
  const int64_t total = (int64_t)B * N * H * lph;
  float a = 0.f;
  if (gidx < total) {
    const T* po = o + gidx * VEC;
    const T* pg = d_o + gidx * VEC;
#pragma unroll
    for (int j = 0; j < VEC; j += 4) {
      const float4 x = load4<T>(po + j), y = load4<T>(pg + j);
      a += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
  }
  for (int off = lph >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  if (gidx < total && (gidx & (lph - 1)) == 0) {
    const int64_t head_idx = gidx / lph;  // over (b, n, h)
    const int h = (int)(head_idx % H);
    const int64_t bn = head_idx / H;
    const int n = (int)(bn % N);
    const int b = (int)(bn / N);
    delta[((int64_t)b * H + h) * N + n] = a;
  }
}

}  // namespace

int attn_delta(int dtype, const void* o, const void* d_o, float* delta, int B, int N, int H, int dh, hipStream_t s) {
  const int vec = dtype == AVF_F32 ? 4 : 8;
  AVF_REQUIRE(dh % vec == 0 && ((dh / vec) & (dh / vec - 1)) == 0 && dh / vec <= 64,
              "attn_delta: dim_head %d must be %d * a power of two", dh, vec);
  const int64_t total = (int64_t)B * N * H * (dh / vec);
  const unsigned grid = (unsigned)ceil_div(total, 256);
  if (dtype == AVF_F32)
    attn_delta_kernel<float, 4><<<grid, 256, 0, s>>>((const float*)o, (const float*)d_o, delta, B, N, H, dh);
  else
    attn_delta_kernel<bf16, 8><<<grid, 256, 0, s>>>((const bf16*)o, (const bf16*)d_o, delta, B, N, H, dh);
  return check_launch("attn_delta_kernel");
}

#define AVF_DH_DISPATCH(dh, MACRO)                                                           \
  switch (dh) {                                                                               \
    case 8: MACRO(8); break;                                                                  \
    case 16: MACRO(16); break;                                                                \
    case 32: MACRO(32); break;                                                                \
    case 64: MACRO(64); break;                                                                \
    default: AVF_REQUIRE(false, "attention (fp32): unsupported dim_head %d (8,16,32,64)", dh); \
  }

int attn_fwd_f32(const float* qkv, float* o, float* lse2, int B, int N, int H, int dh, hipStream_t s) {
  return attn_fwd_vec(AVF_F32, qkv, o, lse2, B, N, H, dh, s, nullptr, false);
}

// the fp32-arithmetic attention on fp32 or bf16 storage, with the optional token mask (keep [B, N] bytes, 1 = kept)
int attn_fwd_vec(int dtype, const void* qkv, void* o, float* lse2, int B, int N, int H, int dh, hipStream_t s,
                 const void* keep, bool q_prescaled) {
  AVF_REQUIRE(B > 0 && N > 0 && H > 0, "attn_fwd_f32: bad shape");
  AVF_REQUIRE((int64_t)B * H < 65536, "attn_fwd_f32: batch*heads too large for grid");
  AVF_REQUIRE(dtype == AVF_F32 || dtype == AVF_BF16, "attn_fwd_f32: bad dtype %d", dtype);
  TimingScope ts(KC_ATTN_FWD, 4.0 * B * H * (double)N * N * dh, 4.0 * 4.0 * B * N * H * dh, s);
  if (attn_f32x3_ok(dtype, dh, keep, H, qkv, o, q_prescaled))  // round 6: three bf16 products per fp32 product (attn_f32x3.hip)
    return attn_fwd_f32x3((const float*)qkv, (float*)o, lse2, B, N, H, s);
  if (attn_f32_mfma_ok(dtype, dh, keep, H, qkv, o))  // fp32 storage, no mask, dim_head 32 / 64: the fp32 matrix pipe
    return attn_fwd_f32_mfma((const float*)qkv, (float*)o, lse2, B, N, H, dh, s, q_prescaled);
  dim3 grid((unsigned)ceil_div(N, 64), (unsigned)(B * H));
  const uint8_t* kp = (const uint8_t*)keep;
  const int qs = q_prescaled ? 1 : 0;
#define LT(D, T, MK) attn_fwd_f32_kernel<D, T, MK><<<grid, 64, 0, s>>>((const T*)qkv, (T*)o, lse2, B, N, H, kp, qs)
#define L(D)                                                  \
  if (dtype == AVF_F32) {                                     \
    if (kp) LT(D, float, true); else LT(D, float, false);     \
  } else {                                                    \
    if (kp) LT(D, bf16, true); else LT(D, bf16, false);       \
  }
  AVF_DH_DISPATCH(dh, L)
#undef L
#undef LT
  return check_launch("attn_fwd_f32_kernel");
}

int attn_bwd_f32(const float* qkv, const float* o, const float* d_o, const float* lse2, float* dqkv, float* delta,
                 int B, int N, int H, int dh, hipStream_t s) {
  return attn_bwd_vec(AVF_F32, qkv, o, d_o, lse2, dqkv, delta, B, N, H, dh, s, nullptr, false);
}

int attn_bwd_vec(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, float* delta,
                 int B, int N, int H, int dh, hipStream_t s, const void* keep, bool q_prescaled) {
  AVF_REQUIRE(B > 0 && N > 0 && H > 0, "attn_bwd_f32: bad shape");
  AVF_REQUIRE((int64_t)B * H < 65536, "attn_bwd_f32: batch*heads too large for grid");
  AVF_REQUIRE(dtype == AVF_F32 || dtype == AVF_BF16, "attn_bwd_f32: bad dtype %d", dtype);
  TimingScope ts(KC_ATTN_BWD, 10.0 * B * H * (double)N * N * dh, 4.0 * 8.0 * B * N * H * dh, s);
  if (attn_f32x3_ok(dtype, dh, keep, H, qkv, d_o, q_prescaled) && ((uintptr_t)dqkv & 15) == 0 && ((uintptr_t)o & 15) == 0)
    return attn_bwd_f32x3((const float*)qkv, (const float*)o, (const float*)d_o, lse2, delta, (float*)dqkv, B, N, H, s);
  AVF_TRY(attn_delta(dtype, o, d_o, delta, B, N, H, dh, s));
  if (attn_f32_mfma_ok(dtype, dh, keep, H, qkv, d_o) && ((uintptr_t)dqkv & 15) == 0)
    return attn_bwd_f32_mfma((const float*)qkv, (const float*)d_o, lse2, delta, (float*)dqkv, B, N, H, dh, s, q_prescaled);
  dim3 grid((unsigned)ceil_div(N, 64), (unsigned)(B * H));
  const uint8_t* kp = (const uint8_t*)keep;
  const int qs = q_prescaled ? 1 : 0;
#define LT(D, T, MK)                                                                                                       \
  attn_dq_f32_kernel<D, T, MK><<<grid, 64, 0, s>>>((const T*)qkv, (const T*)d_o, lse2, delta, (T*)dqkv, B, N, H, kp, qs);     \
  attn_dkv_f32_kernel<D, 0, T, MK><<<grid, 64, 0, s>>>((const T*)qkv, (const T*)d_o, lse2, delta, (T*)dqkv, B, N, H, kp, qs); \
  attn_dkv_f32_kernel<D, 1, T, MK><<<grid, 64, 0, s>>>((const T*)qkv, (const T*)d_o, lse2, delta, (T*)dqkv, B, N, H, kp, qs)
#define L(D)                                                        \
  if (dtype == AVF_F32) {                                           \
    if (kp) { LT(D, float, true); } else { LT(D, float, false); }   \
  } else {                                                          \
    if (kp) { LT(D, bf16, true); } else { LT(D, bf16, false); }     \
  }
  AVF_DH_DISPATCH(dh, L)
#undef L
#undef LT
  return check_launch("attn_bwd_f32 kernels");
}

}  // namespace avf
