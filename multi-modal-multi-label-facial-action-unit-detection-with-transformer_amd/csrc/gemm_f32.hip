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

// K-step: 8 for the general configuration (round 5, back to back at 4096^3 / on the QKV shape of C2: K-step 4: 63 / 68 TFLOP/s, 8: 92 / 87,
// 16: 89.6 / 83, 32: 73 / 67 - short steps keep the staging registers few and more workgroups resident), 32 for the small one
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
  constexpr int BM = 32 * T, BN = 32 * T, BK = T == 1 ? 32 : 8;
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

// ... and of a weight-gradient-shaped GEMM (round 5): a few hundred 64 x 64 tiles with a reduction over thousands of token rows
// (the parity mode's dW at C2: 64 - 192 tiles, K = 10368) ran unsplit on 32 x 32 tiles (262 us per launch, 41 TFLOP/s); split over K
// on the 64 x 64 tiles - ~1024 workgroups, four per CU - the same kernel does 68 - 72 TFLOP/s (151 - 238 us + a 9 us fold).  A
// 128 x 128-tile kernel with 16-byte accesses (on v_mfma_f32_16x16x4_f32 and on 32x32x2) was built for these shapes and for the
// forward / dX ones and measured SLOWER than this kernel everywhere (tools/diag/f32_gemm_probe.py: 76 - 78 against 89 TFLOP/s at
// 4096^3, 57 - 69 against 68 - 72 on the split shapes): many small workgroups per CU hide its barriers better - removed.
int f32_split64(int64_t M, int64_t N, int64_t K, int epilogue) {
  if (epilogue != AVF_EPI_NONE && epilogue != AVF_EPI_BIAS_RES) return 1;
  const int64_t tiles = ceil_div(M, 64) * ceil_div(N, 64);
  if (tiles >= 512 || K < 2048) return 1;
  int64_t sp = ceil_div(1024, tiles);
  if (sp > K / 256) sp = K / 256;  // at least 16 K-steps per split
  if (sp > 16) sp = 16;
  return sp < 2 ? 1 : (int)sp;
}

}  // namespace

size_t gemm_f32_ws(int64_t M, int64_t N, int64_t K) {
  int sp = f32_splits(M, N, K, AVF_EPI_NONE);
  if (sp == 1) sp = f32_split64(M, N, K, AVF_EPI_NONE);
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
  // weight-gradient-shaped GEMMs: 64 x 64 tiles, split K (f32_split64)
  const int sp64 = a.workspace ? f32_split64(a.M, a.N, a.K, a.epilogue) : 1;
  if (sp64 > 1) {
    p.slabs = (float*)a.workspace;
    p.kchunk = (int)(ceil_div(ceil_div(a.K, sp64), 32) * 32);
    const int S = (int)ceil_div(a.K, p.kchunk);
    dim3 grid((unsigned)ceil_div(a.N, 64), (unsigned)ceil_div(a.M, 64), (unsigned)S);
    AVF_REQUIRE(grid.y < 65536, "gemm_f32: M too large for grid");
    gemm_f32_kernel<AVF_EPI_NONE, 2, true><<<grid, 256, 0, s>>>(p);
    AVF_TRY(check_launch("gemm_f32_kernel(split64)"));
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
