// attn_f32_mfma.hip - parity-mode attention core on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), round 5.
//
// Reference: models/heads.py:222-237 (dots = q k^T * dh^-0.5 ; softmax(dim=-1) ; out = attn v) and its autograd.
//
// The parity mode (compute_dtype="f32": the mode held to logits rtol 1e-3 against the fp32 CPU reference) ran its attention on
// attn_f32.hip's one-lane-per-query VALU kernels: 13.4 ms of a 27.8 ms step at C2, 9 TFLOP/s.  These kernels do the same
// arithmetic - fp32 operands, fp32 products, fp32 accumulation, flash-style with log2-domain statistics - on the f32-input MFMA,
// whose product-sum is a k-ordered fmaf chain: same precision class, different summation order.  No token mask and fp32 storage
// only (the masked and the bf16-storage calls stay on attn_f32.hip), dim_head 32 or 64.
//
// One workgroup = 4 wavefronts = 64 queries (forward, dQ) or 64 keys (dK/dV) of one (clip, head); the opposite operand streams
// through LDS in tiles of 32 rows.  Everything is computed TRANSPOSED so that the row a lane's statistics belong to sits on the
// lane index:   S^T = K Q^T  ->  acc[r] of lane (li, lg) = S[query li][key 4 lg + r]   (MFMA D layout: row 4 lg + r, column li).
// A 16 x 16 block of P (or dS) then becomes the B operand of the next product by a 4 x 4 transpose between the register index
// and the lane group (tr4x4: two v_permlane16_swap + two v_permlane32_swap per dword pair): register m of lane (li, lg) holds
// P[key 4 m + lg][query li], exactly what k-step m (keys 4 m .. 4 m + 3) of   O^T += V^T P   asks of that lane.
// The reduction index of the first product is permuted - k-step (c, e) of lane group g carries d = 16 c + 4 g + e - so that a
// lane fetches its four k-steps of a chunk with ONE 16-byte LDS read; both operands use the same permutation.
// LDS images: a row-fragment copy with row stride DH + 4 floats (16-byte reads of 16 consecutive rows: conflict-free) and, where a
// tile is also read transposed, a second copy with row stride DH + 16 (4-byte reads of 4 rows x 16 columns: conflict-free).
#include "common.hpp"

namespace avf {

namespace {

constexpr int MT = 32;  // rows of the streamed operand per tile
constexpr float kLog2e = 1.4426950408889634f;

#define AVF_MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// 4 x 4 transpose of one dword between the register index and the lane group (lanes l, l + 16, l + 32, l + 48):
// in: a[d] of lane group g = X[d][g]; out: a[k] of lane group g = X[g][k]
__device__ __forceinline__ void m_tr4x4(f32x4_t& v) {
  const uint32_t a0 = __float_as_uint(v[0]), a1 = __float_as_uint(v[1]), a2 = __float_as_uint(v[2]), a3 = __float_as_uint(v[3]);
  const auto p01 = __builtin_amdgcn_permlane16_swap(a0, a1, false, false);
  const auto p23 = __builtin_amdgcn_permlane16_swap(a2, a3, false, false);
  const auto x = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
  const auto y = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
  v[0] = __uint_as_float(x[0]); v[1] = __uint_as_float(y[0]); v[2] = __uint_as_float(x[1]); v[3] = __uint_as_float(y[1]);
}

// stage MT rows x DH floats of a row-major fp32 operand (row stride ld) into one or two LDS images; rows >= nvalid are zero
template <int DH, bool TWO>
__device__ __forceinline__ void m_stage(float* rowimg, float* trimg, const float* src, int64_t ld, int row0, int nvalid) {
  constexpr int V = DH / 4, LR = DH + 4, LT = DH + 16;
  for (int i = threadIdx.x; i < MT * V; i += 256) {
    const int r = i / V, c = (i - r * V) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nvalid) v = *reinterpret_cast<const float4*>(src + (int64_t)(row0 + r) * ld + c);
    *reinterpret_cast<float4*>(rowimg + r * LR + c) = v;
    if constexpr (TWO) *reinterpret_cast<float4*>(trimg + r * LT + c) = v;
  }
}

// this lane's fragment registers of one of its wave's 16 rows: f[c][e] = x[row][16 c + 4 lg + e] * scale (zeros past the end)
template <int DH>
__device__ __forceinline__ void m_row_regs(float (&f)[DH / 16][4], const float* row, bool valid, int lg, float scale) {
#pragma unroll
  for (int c = 0; c < DH / 16; ++c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(row + 16 * c + 4 * lg);
    f[c][0] = v.x * scale; f[c][1] = v.y * scale; f[c][2] = v.z * scale; f[c][3] = v.w * scale;
  }
}

// acc[b] (16 x 16, transposed: acc[b][r] of lane (li, lg) = sum_d img[16 b + 4 lg + r][d] * f[lane's row li][d]) for the two
// 16-row blocks of a staged tile: A operand = the tile's rows (one 16-byte read per chunk), B operand = the lane's registers
template <int DH>
__device__ __forceinline__ void m_tile_dot(f32x4_t (&acc)[2], const float* rowimg, const float (&f)[DH / 16][4], int li, int lg) {
  constexpr int LR = DH + 4;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < DH / 16; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(rowimg + (16 * b + li) * LR + 16 * c + 4 * lg);
      acc[b] = AVF_MFMA_F32(a.x, f[c][0], acc[b]);
      acc[b] = AVF_MFMA_F32(a.y, f[c][1], acc[b]);
      acc[b] = AVF_MFMA_F32(a.z, f[c][2], acc[b]);
      acc[b] = AVF_MFMA_F32(a.w, f[c][3], acc[b]);
    }
  }
}

// out[db] (transposed: out[db][r] of lane (li, lg) = column 16 db + 4 lg + r of this lane's row li) += sum over the tile's 32 rows
// of trimg[row][column] * w[row][lane's row li], w given as the two TRANSPOSED 16 x 16 blocks wt[b] (register m of lane (li, lg) =
// w[16 b + 4 m + lg][li])
template <int DH>
__device__ __forceinline__ void m_tile_acc(f32x4_t (&out)[DH / 16], const float* trimg, const f32x4_t (&wt)[2], int li, int lg) {
  constexpr int LT = DH + 16;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int db = 0; db < DH / 16; ++db)
        out[db] = AVF_MFMA_F32(trimg[(16 * b + 4 * m + lg) * LT + 16 * db + li], wt[b][m], out[db]);
}

__device__ __forceinline__ float m_max_groups(float v) {  // maximum over the four lane groups (lanes li, li + 16, + 32, + 48)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float m_sum_groups(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------- forward
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_f32m_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            float* __restrict__ lse2, int N, int H, int qs) {
  constexpr int NC = DH / 16, LR = DH + 4, LT = DH + 16;
  __shared__ __attribute__((aligned(16))) float Ks[MT * LR];
  __shared__ __attribute__((aligned(16))) float Vs[MT * LT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  const int qi = blockIdx.x * 64 + wave * 16 + li;
  const bool valid = qi < N;
  const float c = qs ? 1.0f : kLog2e / sqrtf((float)DH);
  float qf[NC][4];
  m_row_regs<DH>(qf, base + (int64_t)qi * ld, valid, lg, c);
  f32x4_t acc[NC];
#pragma unroll
  for (int db = 0; db < NC; ++db) acc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;  // l: this lane group's share of the row sum (its 8 keys per tile); summed over the groups at the end
  for (int kt = 0; kt < N; kt += MT) {
    const int nk = (N - kt) < MT ? (N - kt) : MT;
    __syncthreads();
    m_stage<DH, false>(Ks, nullptr, base + I, ld, kt, nk);
    {  // V: the transposed-read image only
      constexpr int V4 = DH / 4;
      for (int i = threadIdx.x; i < MT * V4; i += 256) {
        const int r = i / V4, cc = (i - r * V4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nk) v = *reinterpret_cast<const float4*>(base + 2 * I + (int64_t)(kt + r) * ld + cc);
        *reinterpret_cast<float4*>(Vs + r * LT + cc) = v;
      }
    }
    __syncthreads();
    f32x4_t s[2];
    m_tile_dot<DH>(s, Ks, qf, li, lg);  // s[b][r] = S[query li][key kt + 16 b + 4 lg + r] (log2 domain)
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * kb + 4 * lg + r >= nk) s[kb][r] = -INFINITY;
        tmax = fmaxf(tmax, s[kb][r]);
      }
    tmax = m_max_groups(tmax);
    const float mn = fmaxf(m, tmax);
    const float alpha = exp2f(m - mn);
    m = mn;
    l *= alpha;
#pragma unroll
    for (int db = 0; db < NC; ++db) {
      acc[db][0] *= alpha; acc[db][1] *= alpha; acc[db][2] *= alpha; acc[db][3] *= alpha;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kb][r] = exp2f(s[kb][r] - mn);
        l += s[kb][r];
      }
      m_tr4x4(s[kb]);
    }
    m_tile_acc<DH>(acc, Vs, s, li, lg);  // O^T += V^T P
  }
  l = m_sum_groups(l);
  if (valid) {
    const float inv = 1.0f / l;
    float* orow = o + ((int64_t)b * N + qi) * I + h * DH;
#pragma unroll
    for (int db = 0; db < NC; ++db)
      *reinterpret_cast<float4*>(orow + 16 * db + 4 * lg) =
          make_float4(acc[db][0] * inv, acc[db][1] * inv, acc[db][2] * inv, acc[db][3] * inv);
    if (lg == 0) lse2[(int64_t)bh * N + qi] = m + log2f(l);
  }
}

// ---------------------------------------------------------------------------------------------- dQ
// dS = P o (dP - delta),  dq = dS k * dh^-0.5   (64 queries per workgroup, keys streamed)
template <int DH>
__global__ __launch_bounds__(256) void attn_dq_f32m_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                           const float* __restrict__ lse2, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int N, int H, int qs) {
  constexpr int NC = DH / 16, LR = DH + 4, LT = DH + 16;
  __shared__ __attribute__((aligned(16))) float Ks[MT * LR];
  __shared__ __attribute__((aligned(16))) float Kt[MT * LT];
  __shared__ __attribute__((aligned(16))) float Vs[MT * LR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  const int qi = blockIdx.x * 64 + wave * 16 + li;
  const bool valid = qi < N;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = qs ? 1.0f : kLog2e * scale;
  float qf[NC][4], gf[NC][4];
  m_row_regs<DH>(qf, base + (int64_t)qi * ld, valid, lg, c);
  m_row_regs<DH>(gf, d_o + ((int64_t)b * N + qi) * I + h * DH, valid, lg, 1.0f);
  const float L = valid ? lse2[(int64_t)bh * N + qi] : INFINITY;  // rows past the end: P = 2^(s - inf) = 0
  const float dl = valid ? delta[(int64_t)bh * N + qi] : 0.f;
  f32x4_t dq[NC];
#pragma unroll
  for (int db = 0; db < NC; ++db) dq[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < N; kt += MT) {
    const int nk = (N - kt) < MT ? (N - kt) : MT;
    __syncthreads();
    m_stage<DH, true>(Ks, Kt, base + I, ld, kt, nk);
    m_stage<DH, false>(Vs, nullptr, base + 2 * I, ld, kt, nk);
    __syncthreads();
    f32x4_t s[2], dp[2];
    m_tile_dot<DH>(s, Ks, qf, li, lg);
    m_tile_dot<DH>(dp, Vs, gf, li, lg);  // dP[query li][key] = dO[q] . V[key]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool live = 16 * kb + 4 * lg + r < nk;
        const float p = live ? exp2f(s[kb][r] - L) : 0.f;
        s[kb][r] = p * (dp[kb][r] - dl);
      }
      m_tr4x4(s[kb]);
    }
    m_tile_acc<DH>(dq, Kt, s, li, lg);  // dQ^T += K^T dS^T
  }
  if (valid) {
    float* out = dqkv + ((int64_t)b * N + qi) * ld + h * DH;
#pragma unroll
    for (int db = 0; db < NC; ++db)
      *reinterpret_cast<float4*>(out + 16 * db + 4 * lg) =
          make_float4(dq[db][0] * scale, dq[db][1] * scale, dq[db][2] * scale, dq[db][3] * scale);
  }
}

// ---------------------------------------------------------------------------------------------- dK, dV
// dv = P^T dO,  dk = dS^T q * dh^-0.5   (64 keys per workgroup, queries streamed; one recomputation of P serves both)
template <int DH>
__global__ __launch_bounds__(256) void attn_dkv_f32m_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                            const float* __restrict__ lse2, const float* __restrict__ delta,
                                                            float* __restrict__ dqkv, int N, int H, int qs) {
  constexpr int NC = DH / 16, LR = DH + 4, LT = DH + 16;
  __shared__ __attribute__((aligned(16))) float Qs[MT * LR];
  __shared__ __attribute__((aligned(16))) float Qt[MT * LT];
  __shared__ __attribute__((aligned(16))) float Gs[MT * LR];
  __shared__ __attribute__((aligned(16))) float Gt[MT * LT];
  __shared__ __attribute__((aligned(16))) float Ls[MT];
  __shared__ __attribute__((aligned(16))) float Ds[MT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  const float* gbase = d_o + (int64_t)b * N * I + h * DH;
  const int ki = blockIdx.x * 64 + wave * 16 + li;
  const bool valid = ki < N;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = qs ? 1.0f : kLog2e * scale;
  float kf[NC][4], vf[NC][4];
  m_row_regs<DH>(kf, base + I + (int64_t)ki * ld, valid, lg, c);
  m_row_regs<DH>(vf, base + 2 * I + (int64_t)ki * ld, valid, lg, 1.0f);
  f32x4_t dk[NC], dv[NC];
#pragma unroll
  for (int db = 0; db < NC; ++db) dk[db] = dv[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int qt = 0; qt < N; qt += MT) {
    const int nq = (N - qt) < MT ? (N - qt) : MT;
    __syncthreads();
    m_stage<DH, true>(Qs, Qt, base, ld, qt, nq);
    m_stage<DH, true>(Gs, Gt, gbase, I, qt, nq);
    if (threadIdx.x < MT) {
      const bool ok = (int)threadIdx.x < nq;
      Ls[threadIdx.x] = ok ? lse2[(int64_t)bh * N + qt + threadIdx.x] : INFINITY;  // 2^(s - inf) = 0
      Ds[threadIdx.x] = ok ? delta[(int64_t)bh * N + qt + threadIdx.x] : 0.f;
    }
    __syncthreads();
    f32x4_t s[2], dp[2];
    m_tile_dot<DH>(s, Qs, kf, li, lg);   // s[b][r] = S[query qt + 16 b + 4 lg + r][key li]
    m_tile_dot<DH>(dp, Gs, vf, li, lg);  // dP[query][key li] = dO[q] . V[key]
    f32x4_t pt[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float4 Lq = *reinterpret_cast<const float4*>(Ls + 16 * qb + 4 * lg);
      const float4 Dq = *reinterpret_cast<const float4*>(Ds + 16 * qb + 4 * lg);
      const float Lr[4] = {Lq.x, Lq.y, Lq.z, Lq.w}, Dr[4] = {Dq.x, Dq.y, Dq.z, Dq.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s[qb][r] - Lr[r]);
        pt[qb][r] = p;
        s[qb][r] = p * (dp[qb][r] - Dr[r]);
      }
      m_tr4x4(pt[qb]);
      m_tr4x4(s[qb]);
    }
    m_tile_acc<DH>(dv, Gt, pt, li, lg);  // dV^T += dO^T P
    m_tile_acc<DH>(dk, Qt, s, li, lg);   // dK^T += Q^T dS
  }
  if (valid) {
    const float f = qs ? 1.0f / kLog2e : scale;  // qs: q' = q log2(e) scale, dk = dS^T q' / log2(e)
    float* outk = dqkv + ((int64_t)b * N + ki) * ld + I + h * DH;
    float* outv = outk + I;
#pragma unroll
    for (int db = 0; db < NC; ++db) {
      *reinterpret_cast<float4*>(outk + 16 * db + 4 * lg) = make_float4(dk[db][0] * f, dk[db][1] * f, dk[db][2] * f, dk[db][3] * f);
      *reinterpret_cast<float4*>(outv + 16 * db + 4 * lg) = make_float4(dv[db][0], dv[db][1], dv[db][2], dv[db][3]);
    }
  }
}

}  // namespace

// shapes these kernels take: fp32 storage, no token mask, dim_head 32 or 64, 16-byte aligned rows
bool attn_f32_mfma_ok(int dtype, int dh, const void* keep, int H, const void* qkv, const void* other) {
  return dtype == AVF_F32 && !keep && (dh == 32 || dh == 64) && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)other & 15) == 0 &&
         ((H * dh) % 4) == 0;
}

int attn_fwd_f32_mfma(const float* qkv, float* o, float* lse2, int B, int N, int H, int dh, hipStream_t s, bool q_prescaled) {
  dim3 grid((unsigned)ceil_div(N, 64), (unsigned)(B * H));
  const int qs = q_prescaled ? 1 : 0;
  if (dh == 64) attn_fwd_f32m_kernel<64><<<grid, 256, 0, s>>>(qkv, o, lse2, N, H, qs);
  else attn_fwd_f32m_kernel<32><<<grid, 256, 0, s>>>(qkv, o, lse2, N, H, qs);
  return check_launch("attn_fwd_f32m_kernel");
}

int attn_bwd_f32_mfma(const float* qkv, const float* d_o, const float* lse2, const float* delta, float* dqkv, int B, int N, int H,
                      int dh, hipStream_t s, bool q_prescaled) {
  dim3 grid((unsigned)ceil_div(N, 64), (unsigned)(B * H));
  const int qs = q_prescaled ? 1 : 0;
  if (dh == 64) {
    attn_dq_f32m_kernel<64><<<grid, 256, 0, s>>>(qkv, d_o, lse2, delta, dqkv, N, H, qs);
    attn_dkv_f32m_kernel<64><<<grid, 256, 0, s>>>(qkv, d_o, lse2, delta, dqkv, N, H, qs);
  } else {
    attn_dq_f32m_kernel<32><<<grid, 256, 0, s>>>(qkv, d_o, lse2, delta, dqkv, N, H, qs);
    attn_dkv_f32m_kernel<32><<<grid, 256, 0, s>>>(qkv, d_o, lse2, delta, dqkv, N, H, qs);
  }
  return check_launch("attn_bwd_f32m kernels");
}

}  // namespace avf
