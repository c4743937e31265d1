// layer_small.hip - one transformer layer, FORWARD, as a single launch for short sequences.
//
// The reference's real head shapes (AU_former / former_AU_head stacks over 12 tokens; dim 128 or 256,
// 8 heads of 32, mlp 256) are launch-bound: seven forward launches per layer of a few
// microseconds of work each.  Here ONE workgroup (8 or 16 wavefronts) per clip runs the whole layer - LN1, QKV, attention,
// out-projection + residual, LN2, MLP1 + GELU, MLP2 + residual (heads.py:246-255) - writing exactly the saved-activation
// block the per-operator backward reads (h1, stats, qkv, o, lse2, x_mid, h2, stats, u, g), so backward is unchanged.
// Between the phases (a workgroup barrier each) the clip's activations stay in LDS (<= 60 KB) and are ALSO stored into the
// saved block; the GEMMs keep the clip's A rows in registers and read the bf16 weight images straight from L2 as MFMA
// operands (no LDS staging of weights).  Same arithmetic as the per-operator path (bf16 MFMA, fp32 accumulate / LayerNorm / softmax,
// counter-based dropout with the same element indices), except that the softmax is single-pass (all keys at once).
//
// Eligibility (small_layer_ok): bf16, dim_head 32, tokens <= 16, dim / inner / mlp_dim in {128, 256} - the reference's
// AU_former / former_AU_head stacks.  Measured on the real avformer head model (B=64, 12 tokens, three stacks): -3 % per
// step under hipGraph replay, -5..10 % in the host-bound eager loop (42 fewer launches per step).
#include "common.hpp"

namespace avf {

namespace {

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;


struct SmallArgs {
  const float* x_in;
  float* x_out;
  // parameters (fp32 vectors) and bf16 weight images [out, in]
  const float *ln1_w, *ln1_b, *b_out, *ln2_w, *ln2_b, *b1, *b2;
  const bf16 *wqkv, *wo, *w1, *w2;
  // saved-activation block
  bf16 *h1, *qkv, *o, *h2, *u, *g;
  float *mean1, *rstd1, *lse2, *x_mid, *mean2, *rstd2;
  int N, H;
  float eps, score_scale;  // score_scale: 1 when the q rows of wqkv carry log2(e)/sqrt(dh), that factor otherwise
  DropCfg dr0, dr1, dr2;
};

__device__ __forceinline__ bf16x8_t ldg_frag(const bf16* p) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p));
}

__device__ __forceinline__ bf16x8_t pack_pair_s(const f32x4_t& a, const f32x4_t& b) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  u32x4_t r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int LD>  // LD: row stride in bytes of the image the fragment is read from
__device__ __forceinline__ bf16x8_t tr_frag_s(const lds_char* tile, int row_base, int col_base, int li, int lg) {
  const lds_char* p0 = tile + (row_base + 4 * lg + (li >> 2)) * LD + (col_base + 4 * (li & 3)) * 2;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * LD));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// max / sum over the four lanes {li, li+16, li+32, li+48}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// LayerNorm of the clip's rows: wave w takes rows w, w+NW, ... of the R-row block; x is fp32 with row stride ldx (global
// or LDS); the bf16 result goes to the LDS operand buffer ylds (row stride ldy elements; rows past N are zeroed, so that
// everything computed from them stays finite) and, for the rows of the clip, to the saved block in global memory.
template <int D32, int NW, int R>
__device__ __forceinline__ void small_ln(const float* x, int64_t ldx, const float* gamma, const float* beta, bf16* ylds,
                                         int ldy, bf16* ysave, float* mean, float* rstd, int64_t row0, int N, float eps,
                                         int wave, int lane) {
  constexpr int D = D32 * 32;
  const int c = lane * 4;
  const bool act = c < D;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f), b = g;
  if (act) {
    g = *reinterpret_cast<const float4*>(gamma + c);
    b = *reinterpret_cast<const float4*>(beta + c);
  }
  for (int r = wave; r < R; r += NW) {
    if (r >= N) {
      if (act) store4<bf16>(ylds + r * ldy + c, make_float4(0.f, 0.f, 0.f, 0.f));
      continue;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) / (float)D;
    float q = 0.f;
    if (act) {
      const float a0 = v.x - mu, a1 = v.y - mu, a2 = v.z - mu, a3 = v.w - mu;
      q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
      mean[row0 + r] = mu;
      rstd[row0 + r] = rs;
    }
    if (act) {
      const float4 y = make_float4((v.x - mu) * rs * g.x + b.x, (v.y - mu) * rs * g.y + b.y, (v.z - mu) * rs * g.z + b.z,
                                   (v.w - mu) * rs * g.w + b.w);
      store4<bf16>(ylds + r * ldy + c, y);
      store4<bf16>(ysave + (row0 + r) * D + c, y);
    }
  }
}

// C[R rows, Nout] = A[R, K] W[Nout, K]^T; A comes from an LDS operand buffer (row stride lda elements) and stays in
// registers, wave w takes the 16-column blocks w, w+NW, ...; epi(cb, acc): lane holds C[16 i + li][16 cb + 4 lg + 0..3]
template <int MB, int K32, int NW, typename Epi>
__device__ __forceinline__ void small_gemm(const bf16* A, int lda, const bf16* W, int Nout, int wave, int li, int lg,
                                           Epi&& epi) {
  constexpr int K = K32 * 32;
  bf16x8_t fa[MB][K32];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int ks = 0; ks < K32; ++ks)
      fa[i][ks] = *reinterpret_cast<const bf16x8_t*>(A + (16 * i + li) * lda + ks * 32 + 8 * lg);
  // the weight fragments of a column block are all requested before its first MFMA, and the next block's while this one
  // computes (the loop is latency-bound: a few MFMAs per L2 round trip)
  bf16x8_t fb[K32], fn[K32];
  int cb = wave;
  if (cb < Nout / 16) {
    const bf16* wrow = W + (int64_t)(cb * 16 + li) * K + 8 * lg;
#pragma unroll
    for (int ks = 0; ks < K32; ++ks) fb[ks] = ldg_frag(wrow + ks * 32);
  }
  for (; cb < Nout / 16; cb += NW) {
    const bool more = cb + NW < Nout / 16;
    if (more) {
      const bf16* wrow = W + (int64_t)((cb + NW) * 16 + li) * K + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < K32; ++ks) fn[ks] = ldg_frag(wrow + ks * 32);
    }
    f32x4_t acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < K32; ++ks)
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks], fa[i][ks], acc[i], 0, 0, 0);
    epi(cb, acc);
    if (more) {
#pragma unroll
      for (int ks = 0; ks < K32; ++ks) fb[ks] = fn[ks];
    }
  }
}

// NW wavefronts per clip (the GEMM column blocks, LayerNorm rows and heads spread over the waves: the kernel is a chain of
// latency-bound phases).  Between the phases the clip's activations stay in LDS - operand buffer (h1 -> o -> h2), qkv,
// x_mid (fp32), g - and every phase ALSO stores its result into the saved-activation block for backward; only the weight
// images and the layer input come from global memory.
template <int MB, int D32, int I32, int M32, int NW>
__global__ __launch_bounds__(NW * 64) void layer_fwd_small_kernel(SmallArgs a) {
  constexpr int D = D32 * 32, I = I32 * 32, M = M32 * 32, R = MB * 16;
  constexpr int KB = 2 * ((MB + 1) / 2);   // key blocks of 16, padded to pairs (one MFMA k-step = 32 keys)
  constexpr int RK = KB * 16;              // rows of the qkv buffer (>= R)
  constexpr int AMAX = (D > I ? D : I);    // widest operand that lives in abuf (h1, o, h2)
  constexpr int LDA = AMAX + 8, LDG = M + 8, LDQ = 3 * I + 16, LDX = D + 4;  // row strides: +16 B (bf16) keeps the 16-row
  // fragment reads conflict-free; +32 B on the qkv rows does the same for the transposed V reads
  __shared__ __attribute__((aligned(16))) bf16 abuf[R * LDA];
  __shared__ __attribute__((aligned(16))) bf16 gbuf[R * LDG];
  __shared__ __attribute__((aligned(16))) bf16 qbuf[RK * LDQ];
  __shared__ __attribute__((aligned(16))) float xmid[R * LDX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int N = a.N, H = a.H;
  const int64_t b = blockIdx.x, row0 = b * N;

  // rows of the qkv buffer beyond the row blocks the GEMM writes (odd MB): V must be finite there
  for (int e = tid; e < (RK - R) * LDQ / 8; e += NW * 64) reinterpret_cast<uint4*>(qbuf + R * LDQ)[e] = make_uint4(0, 0, 0, 0);

  // ---- LN1
  small_ln<D32, NW, R>(a.x_in + row0 * D, D, a.ln1_w, a.ln1_b, abuf, LDA, a.h1, a.mean1, a.rstd1, row0, N, a.eps, wave, lane);
  __syncthreads();

  // ---- QKV projection (no bias)
  small_gemm<MB, D32, NW>(abuf, LDA, a.wqkv, 3 * I, wave, li, lg, [&](int cb, f32x4_t(&acc)[MB]) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int r = 16 * i + li;
      const float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      store4<bf16>(qbuf + r * LDQ + cb * 16 + 4 * lg, v);
      if (r < N) store4<bf16>(a.qkv + (row0 + r) * (3 * I) + cb * 16 + 4 * lg, v);
    }
  });
  __syncthreads();

  // ---- attention: wave w takes heads w, w+NW, ...; single-pass softmax over the clip's keys; o -> abuf (h1 is dead)
  for (int h = wave; h < H; h += NW) {
    const bf16* qh = qbuf + h * 32;
    bf16x8_t fq[MB], fk[KB];
#pragma unroll
    for (int i = 0; i < MB; ++i) fq[i] = *reinterpret_cast<const bf16x8_t*>(qh + (16 * i + li) * LDQ + 8 * lg);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) fk[kb] = *reinterpret_cast<const bf16x8_t*>(qh + (16 * kb + li) * LDQ + I + 8 * lg);
#pragma unroll
    for (int qb = 0; qb < MB; ++qb) {
      // S^T[key][query]: lane owns query column 16 qb + li, keys 16 kb + 4 lg + r
      f32x4_t st[KB];
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[kb], fq[qb], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sc = (16 * kb + 4 * lg + r < N) ? st[kb][r] * a.score_scale : -INFINITY;
          st[kb][r] = sc;
          mx = fmaxf(mx, sc);
        }
      }
      mx = quad_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[kb][r] = __builtin_amdgcn_exp2f(st[kb][r] - mx);
          sum += st[kb][r];
        }
      sum = quad_sum(sum);
      f32x4_t ot[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s2 = 0; s2 < KB / 2; ++s2) {
        const bf16x8_t pp = pack_pair_s(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
        for (int d = 0; d < 2; ++d)
          ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              tr_frag_s<LDQ * 2>((const lds_char*)(qh + 2 * I), 32 * s2, 16 * d, li, lg), pp, ot[d], 0, 0, 0);
      }
      const int q = 16 * qb + li;
      const float inv = 1.0f / sum;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float4 v = make_float4(ot[d][0] * inv, ot[d][1] * inv, ot[d][2] * inv, ot[d][3] * inv);
        if (q < N) {
          store4<bf16>(abuf + q * LDA + h * 32 + 16 * d + 4 * lg, v);
          store4<bf16>(a.o + (row0 + q) * I + h * 32 + 16 * d + 4 * lg, v);
        }
      }
      if (q < N && lg == 0) a.lse2[(b * H + h) * N + q] = mx + log2f(sum);
    }
  }
  __syncthreads();

  // ---- out-projection + bias, dropout site 0, + residual -> x_mid (fp32)
  {
    const uint64_t key = a.dr0.thresh16 ? drop_key(a.dr0) : 0;
    small_gemm<MB, I32, NW>(abuf, LDA, a.wo, D, wave, li, lg, [&](int cb, f32x4_t(&acc)[MB]) {
      const int n = cb * 16 + 4 * lg;
      const float4 bj = *reinterpret_cast<const float4*>(a.b_out + n);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int r = 16 * i + li;
        if (r >= N) continue;
        const int64_t row = row0 + r;
        float4 df = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.dr0.thresh16) df = drop_factor4(a.dr0, key, (uint64_t)row * D + n);
        const float4 x = *reinterpret_cast<const float4*>(a.x_in + row * D + n);
        const float4 v = make_float4((acc[i][0] + bj.x) * df.x + x.x, (acc[i][1] + bj.y) * df.y + x.y,
                                     (acc[i][2] + bj.z) * df.z + x.z, (acc[i][3] + bj.w) * df.w + x.w);
        *reinterpret_cast<float4*>(xmid + r * LDX + n) = v;
        *reinterpret_cast<float4*>(a.x_mid + row * D + n) = v;
      }
    });
  }
  __syncthreads();

  // ---- LN2 (o in abuf is dead: h2 takes its place)
  small_ln<D32, NW, R>(xmid, LDX, a.ln2_w, a.ln2_b, abuf, LDA, a.h2, a.mean2, a.rstd2, row0, N, a.eps, wave, lane);
  __syncthreads();

  // ---- MLP1 + bias -> u (saved, unmasked), Dropout(GELU(u)) -> g
  {
    const uint64_t key = a.dr1.thresh16 ? drop_key(a.dr1) : 0;
    small_gemm<MB, D32, NW>(abuf, LDA, a.w1, M, wave, li, lg, [&](int cb, f32x4_t(&acc)[MB]) {
      const int n = cb * 16 + 4 * lg;
      const float4 bj = *reinterpret_cast<const float4*>(a.b1 + n);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int r = 16 * i + li;
        const int64_t row = row0 + (r < N ? r : 0);
        const float v0 = acc[i][0] + bj.x, v1 = acc[i][1] + bj.y, v2 = acc[i][2] + bj.z, v3 = acc[i][3] + bj.w;
        float4 df = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.dr1.thresh16) df = drop_factor4(a.dr1, key, (uint64_t)row * M + n);
        const float4 gv = make_float4(gelu_tanh_fast(v0) * df.x, gelu_tanh_fast(v1) * df.y, gelu_tanh_fast(v2) * df.z,
                                      gelu_tanh_fast(v3) * df.w);
        store4<bf16>(gbuf + r * LDG + n, gv);  // rows past N: finite values nobody stores
        if (r < N) {
          store4<bf16>(a.u + row * M + n, make_float4(v0, v1, v2, v3));
          store4<bf16>(a.g + row * M + n, gv);
        }
      }
    });
  }
  __syncthreads();

  // ---- MLP2 + bias, dropout site 2, + residual -> x_out (fp32)
  {
    const uint64_t key = a.dr2.thresh16 ? drop_key(a.dr2) : 0;
    small_gemm<MB, M32, NW>(gbuf, LDG, a.w2, D, wave, li, lg, [&](int cb, f32x4_t(&acc)[MB]) {
      const int n = cb * 16 + 4 * lg;
      const float4 bj = *reinterpret_cast<const float4*>(a.b2 + n);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int r = 16 * i + li;
        if (r >= N) continue;
        const int64_t row = row0 + r;
        float4 df = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.dr2.thresh16) df = drop_factor4(a.dr2, key, (uint64_t)row * D + n);
        const float4 x = *reinterpret_cast<const float4*>(xmid + r * LDX + n);
        *reinterpret_cast<float4*>(a.x_out + row * D + n) =
            make_float4((acc[i][0] + bj.x) * df.x + x.x, (acc[i][1] + bj.y) * df.y + x.y, (acc[i][2] + bj.z) * df.z + x.z,
                        (acc[i][3] + bj.w) * df.w + x.w);
      }
    });
  }
}


// =============================================================================================
// BACKWARD of the short-sequence layer in THREE pieces instead of six dependent launches around the attention backward:
//   kernel A (MLP half + LayerNorm 2 + out-projection):  du = (gy W2) o gelu'(u) -> dh2 = du W1 -> dx_mid = dres + LN2'(dh2)
//                                                        -> d_o = gm Wo            (replaces 4 launches)
//   [attention backward: the per-operator kernels]
//   kernel B (QKV projection + LayerNorm 1):             dh1 = dqkv Wqkv -> dx_in = dx_mid + LN1'(dh1)   (replaces 2 launches)
// One workgroup of 16 wavefronts per clip, the clip's gradients stay in LDS between the phases (dh2 / dh1 never leave the
// chip, and stay fp32 where the per-operator path rounds them to bf16); the GEMM operands the grouped weight-gradient launch
// needs (gy, du, gm, dqkv) are stored as the per-operator path stores them, and the bias / LayerNorm parameter gradients
// leave as one partial row per clip that the layer's fold launch sums.  Same dropout sites and element indices as layer.hip.
// =============================================================================================
struct SmallBwdA {
  const float* dx_out;     // fp32 incoming gradient (null on the bf16 gradient stream)
  const bf16* dx_out_lo;   // its bf16 image, already carrying this layer's site-2 mask (null: made here from dx_out)
  bf16* gy_store;          // where a locally made image is stored for the dW2 GEMM (null when dx_out_lo is given)
  const float* x_mid;      // saved LayerNorm-2 input
  const bf16* u;           // saved pre-activation
  const float *ln2_w, *mean2, *rstd2;
  const bf16 *w2_t, *w1_t, *wo_t;  // transposed bf16 weight images [in, out]
  bf16* du;                // [R, M] for the dW1 GEMM
  float* dx_mid;           // fp32 (null on the bf16 gradient stream)
  bf16* dx_mid_lo;         // bf16 image (site-0 mask applied): operand of dWo and of the d_o GEMM
  bf16* d_o;               // [R, I] for the attention backward
  float* pb1;              // [B][M]   per-clip column sums of du            -> db1
  float* pln2;             // [B][3 D] per-clip dgamma2 | dbeta2 | colsum(gm) -> dbo
  int N;
  int gs16;                // residual gradient arrives / leaves as bf16 only
  DropCfg dr0, dr1, dr2;
};

template <int D32, int I32, int M32, int NW>
__global__ __launch_bounds__(NW * 64) void layer_bwd_small_a_kernel(SmallBwdA a) {
  constexpr int D = D32 * 32, I = I32 * 32, M = M32 * 32, R = 16;
  constexpr int LDA = D + 8, LDG = M + 8, LDX = D + 4;
  __shared__ __attribute__((aligned(16))) bf16 abuf[R * LDA];   // gy, then gm
  __shared__ __attribute__((aligned(16))) bf16 gbuf[R * LDG];   // du
  __shared__ __attribute__((aligned(16))) float xbuf[R * LDX];  // dh2 (fp32)
  __shared__ __attribute__((aligned(16))) float mbuf[R * LDX];  // masked dx_mid (fp32), for its column sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int N = a.N;
  const int64_t b = blockIdx.x, row0 = b * N;

  // ---- phase 0: gy -> abuf (rows past N zero)
  {
    const uint64_t key2 = a.dr2.thresh16 ? drop_key(a.dr2) : 0;
    for (int e = tid; e < R * (D / 4); e += NW * 64) {
      const int r = e / (D / 4), c = (e - r * (D / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < N) {
        if (a.dx_out_lo) {
          v = load4<bf16>(a.dx_out_lo + (row0 + r) * D + c);
        } else {
          v = *reinterpret_cast<const float4*>(a.dx_out + (row0 + r) * D + c);
          if (a.dr2.thresh16) {
            const float4 f = drop_factor4(a.dr2, key2, (uint64_t)(row0 + r) * D + c);
            v.x *= f.x; v.y *= f.y; v.z *= f.z; v.w *= f.w;
          }
          if (a.gy_store) store4<bf16>(a.gy_store + (row0 + r) * D + c, v);
        }
      }
      store4<bf16>(abuf + r * LDA + c, v);
    }
  }
  __syncthreads();

  // ---- phase 1: du = (gy W2) o mask1 o gelu'(u) -> gbuf, global du; per-clip column sums -> pb1
  {
    const uint64_t key1 = a.dr1.thresh16 ? drop_key(a.dr1) : 0;
    small_gemm<1, D32, NW>(abuf, LDA, a.w2_t, M, wave, li, lg, [&](int cb, f32x4_t(&acc)[1]) {
      const int n = cb * 16 + 4 * lg, r = li;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < N) {
        const int64_t row = row0 + r;
        const float4 uu = load4<bf16>(a.u + row * M + n);
        float4 df = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.dr1.thresh16) df = drop_factor4(a.dr1, key1, (uint64_t)row * M + n);
        v = make_float4(acc[0][0] * df.x * dgelu_tanh_fast(uu.x), acc[0][1] * df.y * dgelu_tanh_fast(uu.y),
                        acc[0][2] * df.z * dgelu_tanh_fast(uu.z), acc[0][3] * df.w * dgelu_tanh_fast(uu.w));
        store4<bf16>(a.du + row * M + n, v);
      }
      store4<bf16>(gbuf + r * LDG + n, v);
      float cs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = cs[q];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
        cs[q] = t;
      }
      if (li == 0) *reinterpret_cast<float4*>(a.pb1 + b * M + n) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    });
  }
  __syncthreads();

  // ---- phase 2: dh2 = du W1 -> xbuf (fp32)
  small_gemm<1, M32, NW>(gbuf, LDG, a.w1_t, D, wave, li, lg, [&](int cb, f32x4_t(&acc)[1]) {
    *reinterpret_cast<float4*>(xbuf + li * LDX + cb * 16 + 4 * lg) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
  });
  __syncthreads();

  // ---- phase 3: LayerNorm-2 backward, one wave per row: dx_mid = dres + rstd (g - mean(g) - xhat mean(g xhat)), g = dh2 gamma
  {
    const uint64_t key0 = a.dr0.thresh16 ? drop_key(a.dr0) : 0;
    const int c = lane * 4;
    const bool act = c < D;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) g = *reinterpret_cast<const float4*>(a.ln2_w + c);
    for (int r = wave; r < R; r += NW) {
      if (r >= N) {
        if (act) {
          store4<bf16>(abuf + r * LDA + c, make_float4(0.f, 0.f, 0.f, 0.f));
          *reinterpret_cast<float4*>(mbuf + r * LDX + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        continue;
      }
      const int64_t row = row0 + r;
      const float mu = a.mean2[row], rs = a.rstd2[row];
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), xh = d, res = d;
      if (act) {
        d = *reinterpret_cast<const float4*>(xbuf + r * LDX + c);
        const float4 x = *reinterpret_cast<const float4*>(a.x_mid + row * D + c);
        xh = make_float4((x.x - mu) * rs, (x.y - mu) * rs, (x.z - mu) * rs, (x.w - mu) * rs);
        // bf16 gradient stream: the residual is the bf16 image (the caller's, or the one phase 0 made) - still in abuf
        res = a.gs16 ? load4<bf16>(abuf + r * LDA + c) : *reinterpret_cast<const float4*>(a.dx_out + row * D + c);
      }
      const float g0 = d.x * g.x, g1 = d.y * g.y, g2 = d.z * g.z, g3 = d.w * g.w;
      const float s1 = wave_sum((g0 + g1) + (g2 + g3)) / (float)D;
      const float s2 = wave_sum((g0 * xh.x + g1 * xh.y) + (g2 * xh.z + g3 * xh.w)) / (float)D;
      if (act) {
        float4 o = make_float4(rs * (g0 - s1 - xh.x * s2) + res.x, rs * (g1 - s1 - xh.y * s2) + res.y,
                               rs * (g2 - s1 - xh.z * s2) + res.z, rs * (g3 - s1 - xh.w * s2) + res.w);
        if (a.dx_mid) *reinterpret_cast<float4*>(a.dx_mid + row * D + c) = o;
        if (a.dr0.thresh16) {  // what the out-projection Linear sees: masked, rescaled
          const float4 f = drop_factor4(a.dr0, key0, (uint64_t)row * D + c);
          o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
        }
        store4<bf16>(abuf + r * LDA + c, o);
        store4<bf16>(a.dx_mid_lo + row * D + c, o);
        *reinterpret_cast<float4*>(mbuf + r * LDX + c) = o;
      }
    }
  }
  __syncthreads();
  // column sums over the clip's rows: dgamma2 = sum dh2 xhat, dbeta2 = sum dh2, colsum(gm) (-> dbo)
  for (int c = tid; c < D; c += NW * 64) {
    float dg = 0.f, db = 0.f, cs = 0.f;
    for (int r = 0; r < N; ++r) {
      const int64_t row = row0 + r;
      const float dy = xbuf[r * LDX + c];
      dg += dy * (a.x_mid[row * D + c] - a.mean2[row]) * a.rstd2[row];
      db += dy;
      cs += mbuf[r * LDX + c];
    }
    float* o = a.pln2 + b * 3 * D;
    o[c] = dg; o[D + c] = db; o[2 * D + c] = cs;
  }

  // ---- phase 4: d_o = gm Wo -> global (operand of the attention backward)
  small_gemm<1, D32, NW>(abuf, LDA, a.wo_t, I, wave, li, lg, [&](int cb, f32x4_t(&acc)[1]) {
    if (li < N) store4<bf16>(a.d_o + (row0 + li) * I + cb * 16 + 4 * lg, make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]));
  });
}

struct SmallBwdB {
  // fused attention backward (ATT builds): the saved projection / output / statistics and the incoming d_o; dqkv is then
  // an OUTPUT (the dWqkv GEMM reads it), otherwise the input the per-operator attention backward produced
  const bf16* qkv;         // saved [R, 3 I]
  const bf16* o;           // saved [R, I]
  const bf16* d_o;         // [R, I]
  const float* lse2;       // saved [B, H, N]
  float score_scale, dq_scale, dk_scale;
  int H;
  bf16* dqkv;              // [R, 3 I]
  const bf16* wqkv_t;      // [D, 3 I]
  const float* x_in;       // saved LayerNorm-1 input
  const float *ln1_w, *mean1, *rstd1;
  const float* dx_mid;     // fp32 residual gradient (null on the bf16 gradient stream)
  const bf16* dx_mid_lo;   // its bf16 image
  float* dx_in;            // fp32 out (nullable)
  bf16* dx_in_lo;          // bf16 out (mask of the PREVIOUS layer's site 2 applied)
  float* pln1;             // [B][3 D] per-clip dgamma1 | dbeta1 | colsum(dx_in_lo) -> previous layer's db2
  int N;
  int gs16;
  DropCfg dr_prev2;
};

// ATT: the attention backward of the clip (<= 16 tokens, dim_head 32, one wavefront per head) runs in front of the projection:
//   S^T = K Q^T, P = exp2(S c - lse2), dP^T = V dO^T, delta = rowsum(dO o O), dS = P o (dP - delta)       [2 MFMAs per head]
//   dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T    (the 16 x 16 tiles P / dS pass through LDS to change orientation)
// so that one launch takes d_o to dx_in.  Lane maps as layer_fwd_small_kernel (S^T: lane = query column, rows = keys).
template <int D32, int I32, int NW, bool ATT>
__global__ __launch_bounds__(NW * 64) void layer_bwd_small_b_kernel(SmallBwdB a) {
  constexpr int D = D32 * 32, I = I32 * 32, R = 16, K3 = 3 * I32;
  constexpr int LDQ = 3 * I + 8, LDX = D + 4;
  constexpr int LDS_ = 3 * I + 16;  // saved-projection tile (row stride as the forward's: conflict-free transposed reads)
  constexpr int LDO = I + 8;
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  // layout: [saved qkv tile 32 x LDS_ (ATT) | aliased afterwards by xbuf + mbuf] [dqkv tile] [d_o tile 32 x LDO (ATT)] [head scratch]
  constexpr int SAVED_BYTES = ATT ? 32 * LDS_ * 2 : 0;
  constexpr int X_BYTES = 2 * R * LDX * 4;
  constexpr int FIRST_BYTES = SAVED_BYTES > X_BYTES ? SAVED_BYTES : X_BYTES;
  bf16* sbuf = reinterpret_cast<bf16*>(bsm);
  float* xbuf = reinterpret_cast<float*>(bsm);
  float* mbuf = xbuf + R * LDX;
  bf16* qbuf = reinterpret_cast<bf16*>(bsm + FIRST_BYTES);
  bf16* dobuf = qbuf + R * LDQ;
  bf16* hscr = dobuf + (ATT ? 32 * LDO : 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int N = a.N;
  const int64_t b = blockIdx.x, row0 = b * N;
  if constexpr (ATT) {
    const int H = a.H;
    // saved projection rows (32-row image, rows past N zero) and the d_o rows (likewise)
    for (int e = tid; e < 32 * (3 * I / 8); e += NW * 64) {
      const int r = e / (3 * I / 8), c = (e - r * (3 * I / 8)) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < N) v = *reinterpret_cast<const uint4*>(a.qkv + (row0 + r) * (3 * I) + c);
      *reinterpret_cast<uint4*>(sbuf + r * LDS_ + c) = v;
    }
    for (int e = tid; e < 32 * (I / 8); e += NW * 64) {
      const int r = e / (I / 8), c = (e - r * (I / 8)) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < N) v = *reinterpret_cast<const uint4*>(a.d_o + (row0 + r) * I + c);
      *reinterpret_cast<uint4*>(dobuf + r * LDO + c) = v;
    }
    for (int e = tid; e < (R - N) * (3 * I / 8); e += NW * 64) {  // rows N..15 of the dqkv tile: operands of the projection
      const int r = N + e / (3 * I / 8), c = (e % (3 * I / 8)) * 8;
      *reinterpret_cast<uint4*>(qbuf + r * LDQ + c) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    bf16* pt = hscr + wave * 3 * 256;  // per-wave tiles: P^T [key][q], dS^T [key][q], dS [q][key] (16 x 16 bf16 each)
    bf16* dst = pt + 256;
    bf16* dsq = pt + 512;
    const bool head = wave < H;
    if (head) {
      const int h = wave;
      const bf16* qh = sbuf + h * 32;
      const bf16x8_t fq = *reinterpret_cast<const bf16x8_t*>(qh + li * LDS_ + 8 * lg);
      const bf16x8_t fk = *reinterpret_cast<const bf16x8_t*>(qh + li * LDS_ + I + 8 * lg);
      const bf16x8_t fv = *reinterpret_cast<const bf16x8_t*>(qh + li * LDS_ + 2 * I + 8 * lg);
      const bf16x8_t fdo = *reinterpret_cast<const bf16x8_t*>(dobuf + li * LDO + h * 32 + 8 * lg);
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      const f32x4_t st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq, z, 0, 0, 0);   // S^T[key 4lg+r][query li]
      const f32x4_t dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fdo, z, 0, 0, 0);  // dP^T[key][query]
      // delta[query li] = sum_d dO[q][d] O[q][d]
      float dl = 0.f;
      if (li < N) {
        const bf16x8_t fo = ldg_frag(a.o + (row0 + li) * I + h * 32 + 8 * lg);
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
        const u32x4_t x = __builtin_bit_cast(u32x4_t, fdo), y = __builtin_bit_cast(u32x4_t, fo);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dl = fmaf(__uint_as_float(x[i] << 16), __uint_as_float(y[i] << 16), dl);
          dl = fmaf(__uint_as_float(x[i] & 0xffff0000u), __uint_as_float(y[i] & 0xffff0000u), dl);
        }
      }
      dl = quad_sum(dl);
      const float lse = li < N ? a.lse2[(b * H + h) * N + li] : 0.f;
      float pv[4], dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool live = (4 * lg + r < N) && (li < N);
        pv[r] = live ? __builtin_amdgcn_exp2f(st[r] * a.score_scale - lse) : 0.f;
        dsv[r] = pv[r] * (dp[r] - dl);
        pt[(4 * lg + r) * 16 + li] = from_f32<bf16>(pv[r]);
        dst[(4 * lg + r) * 16 + li] = from_f32<bf16>(dsv[r]);
      }
      store4<bf16>(dsq + li * 16 + 4 * lg, make_float4(dsv[0], dsv[1], dsv[2], dsv[3]));
    }
    __syncthreads();
    if (head) {
      const int h = wave;
      const bf16* qh = sbuf + h * 32;
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
      // B operands with the k-slot order of the transposed A fragments: slots 0..3 = k 4 lg + 0..3, slots 4..7 = k 16 + ... = 0
      auto bfrag = [&](const bf16* tile) {
        const uint2 v = *reinterpret_cast<const uint2*>(tile + li * 16 + 4 * lg);
        const u32x4_t r = {v.x, v.y, 0u, 0u};
        return __builtin_bit_cast(bf16x8_t, r);
      };
      const bf16x8_t bp = bfrag(pt), bds_t = bfrag(dst), bds_q = bfrag(dsq);
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        // transposed A fragments: rows = d (16 d + ...), k = token (32-row images, rows past N zero)
        const bf16x8_t a_do = tr_frag_s<LDO * 2>((const lds_char*)(dobuf + h * 32), 0, 16 * d, li, lg);
        const bf16x8_t a_q = tr_frag_s<LDS_ * 2>((const lds_char*)qh, 0, 16 * d, li, lg);
        const bf16x8_t a_k = tr_frag_s<LDS_ * 2>((const lds_char*)(qh + I), 0, 16 * d, li, lg);
        const f32x4_t dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_do, bp, z, 0, 0, 0);     // dV^T[d][key li]
        const f32x4_t dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_q, bds_t, z, 0, 0, 0);   // dK^T[d][key li]
        const f32x4_t dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_k, bds_q, z, 0, 0, 0);   // dQ^T[d][query li]
        const int col = h * 32 + 16 * d + 4 * lg;
        const float4 vq = make_float4(dq[0] * a.dq_scale, dq[1] * a.dq_scale, dq[2] * a.dq_scale, dq[3] * a.dq_scale);
        const float4 vk = make_float4(dk[0] * a.dk_scale, dk[1] * a.dk_scale, dk[2] * a.dk_scale, dk[3] * a.dk_scale);
        const float4 vv = make_float4(dv[0], dv[1], dv[2], dv[3]);
        if (li < N) {
          store4<bf16>(qbuf + li * LDQ + col, vq);
          store4<bf16>(qbuf + li * LDQ + I + col, vk);
          store4<bf16>(qbuf + li * LDQ + 2 * I + col, vv);
          bf16* g = a.dqkv + (row0 + li) * (3 * I);
          store4<bf16>(g + col, vq);
          store4<bf16>(g + I + col, vk);
          store4<bf16>(g + 2 * I + col, vv);
        }
      }
    }
    __syncthreads();  // dqkv tile complete; the saved-projection tile is dead (xbuf / mbuf take its place)
  } else {
    for (int e = tid; e < R * (3 * I / 8); e += NW * 64) {
      const int r = e / (3 * I / 8), c = (e - r * (3 * I / 8)) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < N) v = *reinterpret_cast<const uint4*>(a.dqkv + (row0 + r) * (3 * I) + c);
      *reinterpret_cast<uint4*>(qbuf + r * LDQ + c) = v;
    }
    __syncthreads();
  }
  small_gemm<1, K3, NW>(qbuf, LDQ, a.wqkv_t, D, wave, li, lg, [&](int cb, f32x4_t(&acc)[1]) {
    *reinterpret_cast<float4*>(xbuf + li * LDX + cb * 16 + 4 * lg) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
  });
  __syncthreads();
  {
    const uint64_t keyp = a.dr_prev2.thresh16 ? drop_key(a.dr_prev2) : 0;
    const int c = lane * 4;
    const bool act = c < D;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) g = *reinterpret_cast<const float4*>(a.ln1_w + c);
    for (int r = wave; r < R; r += NW) {
      if (r >= N) {
        if (act) *reinterpret_cast<float4*>(mbuf + r * LDX + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      const int64_t row = row0 + r;
      const float mu = a.mean1[row], rs = a.rstd1[row];
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), xh = d, res = d;
      if (act) {
        d = *reinterpret_cast<const float4*>(xbuf + r * LDX + c);
        const float4 x = *reinterpret_cast<const float4*>(a.x_in + row * D + c);
        xh = make_float4((x.x - mu) * rs, (x.y - mu) * rs, (x.z - mu) * rs, (x.w - mu) * rs);
        res = a.gs16 ? load4<bf16>(a.dx_mid_lo + row * D + c) : *reinterpret_cast<const float4*>(a.dx_mid + row * D + c);
      }
      const float g0 = d.x * g.x, g1 = d.y * g.y, g2 = d.z * g.z, g3 = d.w * g.w;
      const float s1 = wave_sum((g0 + g1) + (g2 + g3)) / (float)D;
      const float s2 = wave_sum((g0 * xh.x + g1 * xh.y) + (g2 * xh.z + g3 * xh.w)) / (float)D;
      if (act) {
        float4 o = make_float4(rs * (g0 - s1 - xh.x * s2) + res.x, rs * (g1 - s1 - xh.y * s2) + res.y,
                               rs * (g2 - s1 - xh.z * s2) + res.z, rs * (g3 - s1 - xh.w * s2) + res.w);
        if (a.dx_in) *reinterpret_cast<float4*>(a.dx_in + row * D + c) = o;
        if (a.dr_prev2.thresh16) {
          const float4 f = drop_factor4(a.dr_prev2, keyp, (uint64_t)row * D + c);
          o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
        }
        if (a.dx_in_lo) store4<bf16>(a.dx_in_lo + row * D + c, o);
        *reinterpret_cast<float4*>(mbuf + r * LDX + c) = o;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < D; c += NW * 64) {
    float dg = 0.f, db = 0.f, cs = 0.f;
    for (int r = 0; r < N; ++r) {
      const int64_t row = row0 + r;
      const float dy = xbuf[r * LDX + c];
      dg += dy * (a.x_in[row * D + c] - a.mean1[row]) * a.rstd1[row];
      db += dy;
      cs += mbuf[r * LDX + c];
    }
    float* o = a.pln1 + b * 3 * D;
    o[c] = dg; o[D + c] = db; o[2 * D + c] = cs;
  }
}

template <int D32, int I32, int M32>
int launch_small(const SmallArgs& a, int B, hipStream_t s) {
  // one 16-row block, 16 wavefronts.  (The kernel is written for up to four row blocks, but measured on B=4 x 64 tokens
  // - BASELINE's C1 - the single launch is SLOWER than the seven per-operator ones under graph replay, 0.31 vs 0.25 ms per
  // step: with few clips the per-operator kernels spread over more CUs.  So only the 12-token shapes take this path.)
  AVF_REQUIRE(a.N <= 16, "layer_fwd_small: at most 16 tokens");
  layer_fwd_small_kernel<1, D32, I32, M32, 16><<<B, 1024, 0, s>>>(a);
  return check_launch("layer_fwd_small_kernel");
}

}  // namespace

// shapes the single-launch forward covers (AVF_LAYER_SMALL=0 turns it off: tuning / A-B aid)
bool small_layer_ok(int dtype, int tokens, int dim, int heads, int dim_head, int mlp_dim) {
  static const int on = [] {
    const char* e = tuning_env("AVF_LAYER_SMALL");
    return (e && *e) ? atoi(e) : 1;
  }();
  const int inner = heads * dim_head;
  auto ok = [](int v) { return v == 128 || v == 256; };
  return on && dtype == AVF_BF16 && dim_head == 32 && tokens >= 1 && tokens <= 16 && ok(dim) && ok(inner) && ok(mlp_dim);
}

int layer_fwd_small(int B, int N, int D, int H, int M, float eps, float score_scale, const avf_layer_params* p,
                    const void* wqkv, const void* wo, const void* w1, const void* w2, const float* x_in, float* x_out,
                    void* h1, float* mean1, float* rstd1, void* qkv, void* o, float* lse2, float* x_mid, void* h2,
                    float* mean2, float* rstd2, void* u, void* g, const DropCfg& dr0, const DropCfg& dr1,
                    const DropCfg& dr2, hipStream_t s) {
  SmallArgs a;
  a.x_in = x_in; a.x_out = x_out;
  a.ln1_w = p->ln1_w; a.ln1_b = p->ln1_b; a.b_out = p->b_out; a.ln2_w = p->ln2_w; a.ln2_b = p->ln2_b; a.b1 = p->b1; a.b2 = p->b2;
  a.wqkv = (const bf16*)wqkv; a.wo = (const bf16*)wo; a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2;
  a.h1 = (bf16*)h1; a.qkv = (bf16*)qkv; a.o = (bf16*)o; a.h2 = (bf16*)h2; a.u = (bf16*)u; a.g = (bf16*)g;
  a.mean1 = mean1; a.rstd1 = rstd1; a.lse2 = lse2; a.x_mid = x_mid; a.mean2 = mean2; a.rstd2 = rstd2;
  a.N = N; a.H = H; a.eps = eps; a.score_scale = score_scale;
  a.dr0 = dr0; a.dr1 = dr1; a.dr2 = dr2;
  const int I = H * 32;
  AVF_REQUIRE(((uintptr_t)x_in & 15) == 0 && ((uintptr_t)x_out & 15) == 0, "layer_fwd_small: misaligned activations");
#define AVF_SMALL(DD, II, MM) \
  if (D == DD * 32 && I == II * 32 && M == MM * 32) return launch_small<DD, II, MM>(a, B, s)
  AVF_SMALL(4, 4, 4); AVF_SMALL(4, 4, 8); AVF_SMALL(4, 8, 4); AVF_SMALL(4, 8, 8);
  AVF_SMALL(8, 4, 4); AVF_SMALL(8, 4, 8); AVF_SMALL(8, 8, 4); AVF_SMALL(8, 8, 8);
#undef AVF_SMALL
  AVF_REQUIRE(false, "layer_fwd_small: unsupported shape D=%d I=%d M=%d", D, I, M);
}

// bytes of the per-clip partial rows of the two backward kernels: pb1 [B][M], pln2 [B][3D], pln1 [B][3D]
size_t small_bwd_partial_floats(int B, int D, int M) { return (size_t)B * (M + 6 * (size_t)D); }

int layer_bwd_small_a(int B, int N, int D, int I, int M, const SmallBwdAHost& h, hipStream_t s) {
  SmallBwdA a;
  a.dx_out = h.dx_out; a.dx_out_lo = (const bf16*)h.dx_out_lo; a.gy_store = (bf16*)h.gy_store; a.x_mid = h.x_mid;
  a.u = (const bf16*)h.u; a.ln2_w = h.ln2_w; a.mean2 = h.mean2; a.rstd2 = h.rstd2;
  a.w2_t = (const bf16*)h.w2_t; a.w1_t = (const bf16*)h.w1_t; a.wo_t = (const bf16*)h.wo_t;
  a.du = (bf16*)h.du; a.dx_mid = h.dx_mid; a.dx_mid_lo = (bf16*)h.dx_mid_lo; a.d_o = (bf16*)h.d_o;
  a.pb1 = h.pb1; a.pln2 = h.pln2; a.N = N; a.gs16 = h.gs16; a.dr0 = h.dr0; a.dr1 = h.dr1; a.dr2 = h.dr2;
  AVF_REQUIRE(N >= 1 && N <= 16 && (a.dx_out || a.dx_out_lo) && (h.gs16 || a.dx_out) && (a.dx_out_lo || a.gy_store),
              "layer_bwd_small_a: bad arguments");
#define AVF_SMALL_A(DD, II, MM)                                                    \
  if (D == DD * 32 && I == II * 32 && M == MM * 32) {                              \
    layer_bwd_small_a_kernel<DD, II, MM, 16><<<B, 1024, 0, s>>>(a);                \
    return check_launch("layer_bwd_small_a_kernel");                               \
  }
  AVF_SMALL_A(4, 4, 4) AVF_SMALL_A(4, 4, 8) AVF_SMALL_A(4, 8, 4) AVF_SMALL_A(4, 8, 8)
  AVF_SMALL_A(8, 4, 4) AVF_SMALL_A(8, 4, 8) AVF_SMALL_A(8, 8, 4) AVF_SMALL_A(8, 8, 8)
#undef AVF_SMALL_A
  AVF_REQUIRE(false, "layer_bwd_small_a: unsupported shape D=%d I=%d M=%d", D, I, M);
}

template <int D32, int I32, bool ATT>
static int launch_small_b(const SmallBwdB& a, int B, hipStream_t s) {
  constexpr int D = D32 * 32, I = I32 * 32;
  constexpr int saved = ATT ? 32 * (3 * I + 16) * 2 : 0, xb = 2 * 16 * (D + 4) * 4;
  constexpr int smem = (saved > xb ? saved : xb) + 16 * (3 * I + 8) * 2 + (ATT ? 32 * (I + 8) * 2 + 16 * 3 * 256 * 2 : 0);
  static_assert(smem <= 160 * 1024, "LDS budget");
  static PerDeviceOnce raised;
  if (smem > 64 * 1024 && raised.need()) {
    AVF_REQUIRE(hipFuncSetAttribute((const void*)layer_bwd_small_b_kernel<D32, I32, 16, ATT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess,
                "layer_bwd_small_b: cannot raise dynamic LDS limit");
    raised.mark();
  }
  layer_bwd_small_b_kernel<D32, I32, 16, ATT><<<B, 1024, smem, s>>>(a);
  return check_launch("layer_bwd_small_b_kernel");
}

int layer_bwd_small_b(int B, int N, int D, int I, const SmallBwdBHost& h, hipStream_t s) {
  SmallBwdB a;
  a.qkv = (const bf16*)h.qkv; a.o = (const bf16*)h.o; a.d_o = (const bf16*)h.d_o; a.lse2 = h.lse2;
  a.score_scale = h.score_scale; a.dq_scale = h.dq_scale; a.dk_scale = h.dk_scale; a.H = h.H;
  a.dqkv = (bf16*)h.dqkv; a.wqkv_t = (const bf16*)h.wqkv_t; a.x_in = h.x_in; a.ln1_w = h.ln1_w; a.mean1 = h.mean1;
  a.rstd1 = h.rstd1; a.dx_mid = h.dx_mid; a.dx_mid_lo = (const bf16*)h.dx_mid_lo; a.dx_in = h.dx_in; a.dx_in_lo = (bf16*)h.dx_in_lo;
  a.pln1 = h.pln1; a.N = N; a.gs16 = h.gs16; a.dr_prev2 = h.dr_prev2;
  AVF_REQUIRE(N >= 1 && N <= 16 && a.dqkv && (a.dx_in || a.dx_in_lo) && (h.gs16 ? a.dx_mid_lo != nullptr : a.dx_mid != nullptr),
              "layer_bwd_small_b: bad arguments");
  AVF_REQUIRE(!h.attention || (a.qkv && a.o && a.d_o && a.lse2 && h.H >= 1 && h.H <= 16 && h.H * 32 == I),
              "layer_bwd_small_b: the fused attention backward needs the saved projection / output / lse2 and dim_head 32");
#define AVF_SMALL_B(DD, II)                                                                   \
  if (D == DD * 32 && I == II * 32)                                                           \
    return h.attention ? launch_small_b<DD, II, true>(a, B, s) : launch_small_b<DD, II, false>(a, B, s);
  AVF_SMALL_B(4, 4) AVF_SMALL_B(4, 8) AVF_SMALL_B(8, 4) AVF_SMALL_B(8, 8)
#undef AVF_SMALL_B
  AVF_REQUIRE(false, "layer_bwd_small_b: unsupported shape D=%d I=%d", D, I);
}

}  // namespace avf
