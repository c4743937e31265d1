// gemm_f32.hip - parity-mode GEMM: fp32 operands, fp32 accumulate, on v_mfma_f32_16x16x4_f32.
//
// The f32-input MFMA is bit-for-bit a k-ordered fmaf chain (one rounding per product), so this
// path reproduces an fp32 CPU matmul to ~1e-7 relative.  It serves every operand orientation
// (nn.Linear forward "NT", dX "NN", dW "TN") through element strides, because the f32 MFMA
// fragment is ONE value per lane (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]) and needs no particular
// LDS layout.  Replaces aten mm/addmm under models/heads.py:191-196, 212, 214-217.
#include "common.hpp"

namespace avf {

namespace {

// K-step: 16 for the general configuration, 32 for the small one (half the barriers per FLOP)
// block tile (32 T) x (32 T): 2 x 2 waves of (16 T) x (16 T) each.  T = 2 (64 x 64) is the general configuration; T = 1
// (32 x 32) serves the small GEMMs around the stacks - the 12-way projection of AU_former on a batch of a few dozen clips,
// the AU logits - where 64 x 64 tiles would leave most of the chip without a workgroup.

struct F32GemmParams {
  const float* A;
  int64_t a_sm, a_sk;  // element (m,k) at A[m*a_sm + k*a_sk]
  const float* B;
  int64_t b_sk, b_sn;  // element (k,n) at B[k*b_sk + n*b_sn]
  float* C;
  int64_t ldc;
  const float* bias;
  const float* residual;
  int64_t ldres;
  float* aux;
  int64_t ldaux;
  int M, N, K;
  DropCfg drop;    // dropout site fused in the epilogue (thresh16 == 0: none); element index = m * N + n, as the bf16 kernels
  int kchunk;      // split-K (gridDim.z > 1): k-range per split, a multiple of the K-step
  float* slabs;    // split-K: raw partial accumulators [split][M][N] (dense); the fold kernel applies the epilogue
};

template <int EPI, int T, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(F32GemmParams p) {
  constexpr int BM = 32 * T, BN = 32 * T, BK = T == 1 ? 32 : 16;
  constexpr int AS_LD = BK + 1;   // As[m][k] rows
  constexpr int BS_LD = BN + 16;  // Bs[k][n] rows (k -> +16 banks)
  __shared__ float As[BM * AS_LD];
  __shared__ float Bs[BK * BS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, (16 T) x (16 T) each
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int li = lane & 15, lg = lane >> 4;

  f32x4_t acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // thread -> tile element mapping follows the contiguous axis of each operand (wave-uniform choice)
  const bool a_kfast = (p.a_sk == 1);
  const bool b_nfast = (p.b_sn == 1);

  const int kbeg = SPLIT ? (int)blockIdx.z * p.kchunk : 0;
  const int kend = SPLIT ? ((kbeg + p.kchunk) < p.K ? kbeg + p.kchunk : p.K) : p.K;
  // register prefetch: the global loads of K-step t+1 are in flight while step t runs its MFMAs out of LDS
  constexpr int NA = BM * BK / 256, NB = BN * BK / 256;
  float ra[NA], rb[NB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int idx = tid + 256 * e;
      int m, k;
      if (a_kfast) { k = idx & (BK - 1); m = idx / BK; } else { m = idx & (BM - 1); k = idx / BM; }
      const int gm = m0 + m, gk = k0 + k;
      ra[e] = (gm < p.M && gk < kend) ? p.A[(int64_t)gm * p.a_sm + (int64_t)gk * p.a_sk] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int idx = tid + 256 * e;
      int n, k;
      if (b_nfast) { n = idx & (BN - 1); k = idx / BN; } else { k = idx & (BK - 1); n = idx / BK; }
      const int gn = n0 + n, gk = k0 + k;
      rb[e] = (gn < p.N && gk < kend) ? p.B[(int64_t)gk * p.b_sk + (int64_t)gn * p.b_sn] : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int idx = tid + 256 * e;
      int m, k;
      if (a_kfast) { k = idx & (BK - 1); m = idx / BK; } else { m = idx & (BM - 1); k = idx / BM; }
      As[m * AS_LD + k] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int idx = tid + 256 * e;
      int n, k;
      if (b_nfast) { n = idx & (BN - 1); k = idx / BN; } else { k = idx & (BK - 1); n = idx / BK; }
      Bs[k * BS_LD + n] = rb[e];
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    commit();
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[T], b[T];
#pragma unroll
      for (int i = 0; i < T; ++i) a[i] = As[(wm * 16 * T + i * 16 + li) * AS_LD + ks * 4 + lg];
#pragma unroll
      for (int j = 0; j < T; ++j) b[j] = Bs[(ks * 4 + lg) * BS_LD + wn * 16 * T + j * 16 + li];
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  const uint64_t dkey = (!SPLIT && p.drop.thresh16) ? drop_key(p.drop) : 0;
  // C/D map of the 16x16 tile: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * 16 * T + i * 16 + lg * 4 + r;
        const int gn = n0 + wn * 16 * T + j * 16 + li;
        if (gm >= p.M || gn >= p.N) continue;
        float v = acc[i][j][r];
        if (SPLIT) {  // raw partial: the fold kernel sums the splits in order and applies the epilogue
          p.slabs[((int64_t)blockIdx.z * p.M + gm) * p.N + gn] = v;
          continue;
        }
        if (p.bias) v += p.bias[gn];
        const float df = p.drop.thresh16 ? drop_factor1(p.drop, dkey, (uint64_t)gm * p.N + gn) : 1.0f;  // wave-uniform branch
        if (EPI == AVF_EPI_BIAS_RES) {  // x + Dropout(Linear(.))
          v = v * df + p.residual[(int64_t)gm * p.ldres + gn];
        } else if (EPI == AVF_EPI_BIAS_GELU) {  // Dropout(GELU(u)); u is saved unmasked
          p.aux[(int64_t)gm * p.ldaux + gn] = v;
          v = gelu_tanh_f(v) * df;
        } else if (EPI == AVF_EPI_DGELU) {  // backward through Dropout then GELU
          v *= df * dgelu_tanh_f(p.aux[(int64_t)gm * p.ldaux + gn]);
        }
        p.C[(int64_t)gm * p.ldc + gn] = v;
      }
}

// ------------------------------------------------------------------------------------------
// Fast form for the layer's own shapes (round 5): M % 128 == 0, N % 128 == 0, K % 32 == 0, 16-byte aligned dense operands, no
// dropout.  128 x 128 block tile, 4 waves of 64 x 64 (64 accumulator registers), K-step 32, register prefetch of the next
// K-step's global loads (16-byte loads along each operand's contiguous axis), ONE LDS buffer (two barriers per K-step: its 128
// MFMAs per wave are 4096 cycles).  The operands enter the MFMA SWAPPED - A-operand = the B matrix (column n on the lane), B-operand
// = the A matrix (row m on the lane) - so a lane owns C[m][n .. n + 3]: 16-byte stores, bias / residual / saved pre-activation as
// 16-byte loads (the general kernel above stores every value by itself).  An operand that is k-contiguous in memory (A of a forward
// / dX GEMM, the nn.Linear weight of a forward GEMM) is staged as [row][k] with row stride 36 floats and read 16 bytes at a time:
// k-step (c, e) of lane group g carries k = 16 c + 4 g + e on BOTH operands, one read per four MFMAs; an operand whose other axis is
// contiguous (the weight [k][n] of a dX GEMM, both operands of a weight-gradient GEMM) is staged as [k][row] with row stride 132 and
// read 4 bytes per MFMA (lanes = consecutive rows: conflict-free).  Split-K (SPLIT: the weight gradients, 16 - 48 tiles on 256
// CUs) writes raw partial slabs; gemm_f32_fold_kernel sums them in order.  Same products, same fp32 fmaf accumulation as the
// general kernel; the k ORDER inside a 16-chunk differs (a permutation), so results agree to fp32 rounding, not bit for bit.
// ------------------------------------------------------------------------------------------
template <int EPI, bool AKF, bool BKF, bool SPLIT, int BT = 128>
__global__ __launch_bounds__(256) void gemm_f32_fast_kernel(F32GemmParams p) {
  // BT = 128: 4 waves of 64 x 64; BT = 64: 4 waves of 32 x 32 (four times the workgroups: the forward / dX shapes, whose 128 x 128
  // grids are 1.3 rounds of the chip's workgroup slots)
  static_assert(BT == 128 || BT == 64, "block tile");
  constexpr int BK = 32, LR = BK + 4, LT = BT + 4;  // (LT: rows 4 lg + e -> 4 LT = 16 mod 64 banks per lane group)
  constexpr int NB = BT / 32;                       // 16 x 16 blocks per wave and dimension = 16-byte pieces per thread and operand
  constexpr int WT = BT / 2;                        // wave tile
  constexpr int A_FLOATS = AKF ? BT * LR : BK * LT, B_FLOATS = BKF ? BT * LR : BK * LT;
  __shared__ __attribute__((aligned(16))) float As[A_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[B_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
  const int kbeg = SPLIT ? (int)blockIdx.z * p.kchunk : 0;
  const int kend = SPLIT ? ((kbeg + p.kchunk) < p.K ? kbeg + p.kchunk : p.K) : p.K;

  f32x4_t acc[NB][NB];  // acc[im][jn][r] = C[m = WT wm + 16 im + li][n = WT wn + 16 jn + 4 lg + r]
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // staging: BT x 32 floats per operand = 8 BT float4, NB per thread.
  //   k-fast operand: thread -> row t / 8 + 32 e, k chunk (t % 8) * 4                       (128-byte row segments)
  //   row-fast operand: thread -> k t / (BT / 4) + (1024 / BT) e, row chunk (t % (BT / 4)) * 4   (whole k rows of the tile)
  constexpr int RPT = BT / 4, KPE = 1024 / BT;
  float4 ra[NB], rb[NB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      if constexpr (AKF) ra[e] = *reinterpret_cast<const float4*>(p.A + (int64_t)(m0 + (tid >> 3) + 32 * e) * p.a_sm + k0 + (tid & 7) * 4);
      else ra[e] = *reinterpret_cast<const float4*>(p.A + (int64_t)(k0 + tid / RPT + KPE * e) * p.a_sk + m0 + (tid % RPT) * 4);
      if constexpr (BKF) rb[e] = *reinterpret_cast<const float4*>(p.B + (int64_t)(n0 + (tid >> 3) + 32 * e) * p.b_sn + k0 + (tid & 7) * 4);
      else rb[e] = *reinterpret_cast<const float4*>(p.B + (int64_t)(k0 + tid / RPT + KPE * e) * p.b_sk + n0 + (tid % RPT) * 4);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      if constexpr (AKF) *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * e) * LR + (tid & 7) * 4) = ra[e];
      else *reinterpret_cast<float4*>(As + (tid / RPT + KPE * e) * LT + (tid % RPT) * 4) = ra[e];
      if constexpr (BKF) *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * e) * LR + (tid & 7) * 4) = rb[e];
      else *reinterpret_cast<float4*>(Bs + (tid / RPT + KPE * e) * LT + (tid % RPT) * 4) = rb[e];
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    commit();
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
    for (int c = 0; c < BK / 16; ++c) {
      float av[NB][4], bv[NB][4];  // [block][e]: the operand values of k-steps (c, e), k = 16 c + 4 lg + e
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if constexpr (AKF) {
          const float4 v = *reinterpret_cast<const float4*>(As + (WT * wm + 16 * i + li) * LR + 16 * c + 4 * lg);
          av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[i][e] = As[(16 * c + 4 * lg + e) * LT + WT * wm + 16 * i + li];
        }
        if constexpr (BKF) {
          const float4 v = *reinterpret_cast<const float4*>(Bs + (WT * wn + 16 * i + li) * LR + 16 * c + 4 * lg);
          bv[i][0] = v.x; bv[i][1] = v.y; bv[i][2] = v.z; bv[i][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[i][e] = Bs[(16 * c + 4 * lg + e) * LT + WT * wn + 16 * i + li];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j][e], av[i][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // D map with the operands swapped: row (= n) 4 lg + r, column (= m) li
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int gm = m0 + WT * wm + 16 * i + li;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int gn = n0 + WT * wn + 16 * j + 4 * lg;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if constexpr (SPLIT) {
        *reinterpret_cast<float4*>(p.slabs + ((int64_t)blockIdx.z * p.M + gm) * p.N + gn) = v;
      } else {
        if (p.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(p.bias + gn);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if constexpr (EPI == AVF_EPI_BIAS_RES) {
          const float4 rr = *reinterpret_cast<const float4*>(p.residual + (int64_t)gm * p.ldres + gn);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        } else if constexpr (EPI == AVF_EPI_BIAS_GELU) {
          *reinterpret_cast<float4*>(p.aux + (int64_t)gm * p.ldaux + gn) = v;
          v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w);
        } else if constexpr (EPI == AVF_EPI_DGELU) {
          const float4 u = *reinterpret_cast<const float4*>(p.aux + (int64_t)gm * p.ldaux + gn);
          v.x *= dgelu_tanh_f(u.x); v.y *= dgelu_tanh_f(u.y); v.z *= dgelu_tanh_f(u.z); v.w *= dgelu_tanh_f(u.w);
        }
        *reinterpret_cast<float4*>(p.C + (int64_t)gm * p.ldc + gn) = v;
      }
    }
  }
}

// the shapes the fast form takes, and its split count (1 = no split).  Split-K only where the fold can apply the epilogue.
static bool f32_fast_ok(const GemmArgs& a) {
  if (a.M % 128 || a.N % 128 || a.K % 32 || a.drop.thresh16) return false;
  if (a.transA && a.transB) return false;  // (no caller: A m-fast with B k-fast)
  if (a.lda % 4 || a.ldb % 4 || a.ldc % 4 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return false;
  if (a.bias && ((uintptr_t)a.bias & 15)) return false;
  if (a.epilogue == AVF_EPI_BIAS_RES && (!a.residual || a.ldres % 4 || ((uintptr_t)a.residual & 15))) return false;
  if ((a.epilogue == AVF_EPI_BIAS_GELU || a.epilogue == AVF_EPI_DGELU) && (!a.aux || a.ldaux % 4 || ((uintptr_t)a.aux & 15))) return false;
  return (a.M / 128) * (a.N / 128) >= 16;  // fewer tiles: the general kernel's 32 x 32 / 64 x 64 tiles fill the chip better
}
static int f32_fast_splits(int64_t M, int64_t N, int64_t K, int epilogue) {
  const int64_t tiles = (M / 128) * (N / 128);
  if (tiles >= 160 || (epilogue != AVF_EPI_NONE && epilogue != AVF_EPI_BIAS_RES)) return 1;
  int64_t sp = ceil_div(512, tiles);  // two workgroups (two waves per SIMD) per CU: one alone leaves the matrix pipe idle in its LDS phases
  if (sp > K / 256) sp = K / 256;     // at least 8 K-steps per split
  if (sp > 32) sp = 32;
  return sp < 2 ? 1 : (int)sp;
}

// C = sum_z slabs[z] (+ bias, + residual): the second half of a split-K launch (EPI_NONE / EPI_BIAS_RES)
__global__ __launch_bounds__(256) void gemm_f32_fold_kernel(F32GemmParams p, int S, int with_res) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t mn = (int64_t)p.M * p.N;
  if (i >= mn) return;
  const int m = (int)(i / p.N), n = (int)(i - (int64_t)m * p.N);
  float v = p.slabs[i];
  for (int z = 1; z < S; ++z) v += p.slabs[(int64_t)z * mn + i];
  if (p.bias) v += p.bias[n];
  if (with_res) {
    if (p.drop.thresh16) v *= drop_factor1(p.drop, drop_key(p.drop), (uint64_t)i);
    v += p.residual[(int64_t)m * p.ldres + n];
  }
  p.C[(int64_t)m * p.ldc + n] = v;
}

// split count of a small-grid GEMM: enough workgroups for the chip, at least 64 k per split
int f32_splits(int64_t M, int64_t N, int64_t K, int epilogue) {
  if (epilogue != AVF_EPI_NONE && epilogue != AVF_EPI_BIAS_RES) return 1;
  const int64_t wgs = ceil_div(N, 32) * ceil_div(M, 32);
  if (ceil_div(N, 64) * ceil_div(M, 64) >= 256 || wgs >= 192 || K < 256) return 1;
  int64_t sp = ceil_div(256, wgs);
  if (sp > K / 64) sp = K / 64;
  if (sp > 16) sp = 16;
  return sp < 2 ? 1 : (int)sp;
}

}  // namespace

size_t gemm_f32_ws(int64_t M, int64_t N, int64_t K) {
  int sp = f32_splits(M, N, K, AVF_EPI_NONE);
  if (M % 128 == 0 && N % 128 == 0 && K % 32 == 0 && (M / 128) * (N / 128) >= 16) {  // (the fast form may split where the general one does not)
    const int fs = f32_fast_splits(M, N, K, AVF_EPI_NONE);
    sp = fs > sp ? fs : sp;
  }
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}

int gemm_f32(const GemmArgs& a, hipStream_t s) {
  AVF_REQUIRE(a.c_dtype == AVF_F32, "gemm_f32: C must be fp32");
  AVF_REQUIRE(!a.drop.thresh16 || a.epilogue != AVF_EPI_NONE, "gemm_f32: dropout needs a fused epilogue");
  AVF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_f32: bad shape");
  AVF_REQUIRE(a.M < (1LL << 31) && a.N < (1LL << 31) && a.K < (1LL << 31), "gemm_f32: shape too large");
  F32GemmParams p;
  TimingScope ts(KC_GEMM_F32, 2.0 * a.M * a.N * a.K, 4.0 * (a.M * a.K + a.N * a.K + a.M * a.N), s);
  p.A = (const float*)a.A;
  p.B = (const float*)a.B;
  if (a.transA) { p.a_sm = 1; p.a_sk = a.lda; } else { p.a_sm = a.lda; p.a_sk = 1; }
  if (a.transB) { p.b_sk = 1; p.b_sn = a.ldb; } else { p.b_sk = a.ldb; p.b_sn = 1; }
  p.C = (float*)a.C;
  p.ldc = a.ldc;
  p.bias = a.bias;
  p.residual = (const float*)a.residual;
  p.ldres = a.ldres;
  p.aux = (float*)a.aux;
  p.ldaux = a.ldaux;
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  p.drop = a.drop;
  p.kchunk = 0; p.slabs = nullptr;
  // the layer's own shapes: the fast form (128 x 128 tiles, 16-byte accesses), split-K for the few-tile weight gradients
  // (measured at C2, us per launch, general 64 x 64 kernel | fast form with 128 x 128 tiles | with 64 x 64 tiles (BT = 64): the
  //  split-K weight gradients 262 | 171 + 9 fold | -; QKV 230 | 244 | 297; dX 165 | 175 | 232; bias + residual 122 | 167 | 183 -
  //  wherever the general kernel's 3888 small workgroups already fill the chip it stays ahead: the fast form is taken for the
  //  SPLIT shapes only)
  const int fs_probe = (f32_fast_ok(a) && a.workspace) ? f32_fast_splits(a.M, a.N, a.K, a.epilogue) : 1;
  if (fs_probe > 1) {
    const int fs = fs_probe;
    const bool akf = !a.transA, bkf = a.transB != 0;
    dim3 grid((unsigned)(a.N / 128), (unsigned)(a.M / 128), 1);
    AVF_REQUIRE(grid.y < 65536, "gemm_f32: M too large for grid");
#define FAST_E(E, SPL)                                                                                  \
  do {                                                                                                  \
    if (akf && bkf) gemm_f32_fast_kernel<E, true, true, SPL><<<grid, 256, 0, s>>>(p);                   \
    else if (akf) gemm_f32_fast_kernel<E, true, false, SPL><<<grid, 256, 0, s>>>(p);                    \
    else gemm_f32_fast_kernel<E, false, false, SPL><<<grid, 256, 0, s>>>(p);                            \
  } while (0)
    if (fs > 1) {
      p.slabs = (float*)a.workspace;
      p.kchunk = (int)(ceil_div(ceil_div(a.K, fs), 32) * 32);
      const int S = (int)ceil_div(a.K, p.kchunk);
      grid.z = (unsigned)S;
      FAST_E(AVF_EPI_NONE, true);
      AVF_TRY(check_launch("gemm_f32_fast_kernel(split)"));
      AVF_REQUIRE(a.epilogue == AVF_EPI_NONE || a.residual, "gemm_f32: residual missing");
      gemm_f32_fold_kernel<<<(unsigned)ceil_div(a.M * a.N, 256), 256, 0, s>>>(p, S, a.epilogue == AVF_EPI_BIAS_RES ? 1 : 0);
      return check_launch("gemm_f32_fold_kernel");
    }
    switch (a.epilogue) {
      case AVF_EPI_NONE: FAST_E(AVF_EPI_NONE, false); break;
      case AVF_EPI_BIAS_RES: FAST_E(AVF_EPI_BIAS_RES, false); break;
      case AVF_EPI_BIAS_GELU: FAST_E(AVF_EPI_BIAS_GELU, false); break;
      case AVF_EPI_DGELU: FAST_E(AVF_EPI_DGELU, false); break;
      default: AVF_REQUIRE(false, "gemm_f32: bad epilogue %d", a.epilogue);
    }
#undef FAST_E
    return check_launch("gemm_f32_fast_kernel");
  }
  // skinny GEMMs with a long reduction (the heads' projections on a few dozen clips): split K over the grid, raw partials
  // into the caller's workspace, one fold launch with the epilogue - deterministic (no atomics)
  const int sp = a.workspace ? f32_splits(a.M, a.N, a.K, a.epilogue) : 1;
  if (sp > 1) {
    p.slabs = (float*)a.workspace;
    p.kchunk = (int)(ceil_div(ceil_div(a.K, sp), 32) * 32);
    const int S = (int)ceil_div(a.K, p.kchunk);
    dim3 grid((unsigned)ceil_div(a.N, 32), (unsigned)ceil_div(a.M, 32), (unsigned)S);
    gemm_f32_kernel<AVF_EPI_NONE, 1, true><<<grid, 256, 0, s>>>(p);
    AVF_TRY(check_launch("gemm_f32_kernel(split)"));
    AVF_REQUIRE(a.epilogue == AVF_EPI_NONE || a.residual, "gemm_f32: residual missing");
    gemm_f32_fold_kernel<<<(unsigned)ceil_div(a.M * a.N, 256), 256, 0, s>>>(p, S, a.epilogue == AVF_EPI_BIAS_RES ? 1 : 0);
    return check_launch("gemm_f32_fold_kernel");
  }
  // 32 x 32 tiles when 64 x 64 ones would not give every CU a workgroup (the small GEMMs of the heads)
  const bool small = ceil_div(a.N, 64) * ceil_div(a.M, 64) < 256;
  const int bt = small ? 32 : 64;
  dim3 grid((unsigned)ceil_div(a.N, bt), (unsigned)ceil_div(a.M, bt));
  AVF_REQUIRE(grid.y < 65536, "gemm_f32: M too large for grid");
#define LAUNCH_F32(E)                                             \
  do {                                                            \
    if (small) gemm_f32_kernel<E, 1><<<grid, 256, 0, s>>>(p);     \
    else gemm_f32_kernel<E, 2><<<grid, 256, 0, s>>>(p);           \
  } while (0)
  switch (a.epilogue) {
    case AVF_EPI_NONE: LAUNCH_F32(AVF_EPI_NONE); break;
    case AVF_EPI_BIAS_RES:
      AVF_REQUIRE(a.residual, "gemm_f32: residual missing");
      LAUNCH_F32(AVF_EPI_BIAS_RES);
      break;
    case AVF_EPI_BIAS_GELU:
      AVF_REQUIRE(a.aux, "gemm_f32: aux missing");
      LAUNCH_F32(AVF_EPI_BIAS_GELU);
      break;
    case AVF_EPI_DGELU:
      AVF_REQUIRE(a.aux, "gemm_f32: aux missing");
      LAUNCH_F32(AVF_EPI_DGELU);
      break;
    default: AVF_REQUIRE(false, "gemm_f32: bad epilogue %d", a.epilogue);
  }
#undef LAUNCH_F32
  return check_launch("gemm_f32_kernel");
}

}  // namespace avf
