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

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int AS_LD = BK + 1;   // As[m][k], 17-float rows
constexpr int BS_LD = BN + 16;  // Bs[k][n], 80-float rows (k -> +16 banks)

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
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(F32GemmParams p) {
  __shared__ float As[BM * AS_LD];
  __shared__ float Bs[BK * BS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, 32 x 32 each
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int li = lane & 15, lg = lane >> 4;

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // thread -> tile element mapping follows the contiguous axis of each operand (wave-uniform choice)
  const bool a_kfast = (p.a_sk == 1);
  const bool b_nfast = (p.b_sn == 1);

  for (int k0 = 0; k0 < p.K; k0 += BK) {
    // A tile: 64 x 16 = 1024 elements, 4 per thread
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      int m, k;
      if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      float v = 0.f;
      if (gm < p.M && gk < p.K) v = p.A[(int64_t)gm * p.a_sm + (int64_t)gk * p.a_sk];
      As[m * AS_LD + k] = v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      int n, k;
      if (b_nfast) { n = idx & 63; k = idx >> 6; } else { k = idx & 15; n = idx >> 4; }
      const int gn = n0 + n, gk = k0 + k;
      float v = 0.f;
      if (gn < p.N && gk < p.K) v = p.B[(int64_t)gk * p.b_sk + (int64_t)gn * p.b_sn];
      Bs[k * BS_LD + n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + li) * AS_LD + ks * 4 + lg];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(ks * 4 + lg) * BS_LD + wn * 32 + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 16x16 tile: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * 32 + i * 16 + lg * 4 + r;
        const int gn = n0 + wn * 32 + j * 16 + li;
        if (gm >= p.M || gn >= p.N) continue;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[gn];
        if (EPI == AVF_EPI_BIAS_RES) {
          v += p.residual[(int64_t)gm * p.ldres + gn];
        } else if (EPI == AVF_EPI_BIAS_GELU) {
          p.aux[(int64_t)gm * p.ldaux + gn] = v;
          v = gelu_tanh_f(v);
        } else if (EPI == AVF_EPI_DGELU) {
          v *= dgelu_tanh_f(p.aux[(int64_t)gm * p.ldaux + gn]);
        }
        p.C[(int64_t)gm * p.ldc + gn] = v;
      }
}

}  // namespace

int gemm_f32(const GemmArgs& a, hipStream_t s) {
  AVF_REQUIRE(a.c_dtype == AVF_F32, "gemm_f32: C must be fp32");
  AVF_REQUIRE(!a.drop.thresh16, "gemm_f32: dropout is only implemented on the bf16 path");
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
  dim3 grid((unsigned)ceil_div(a.N, BN), (unsigned)ceil_div(a.M, BM));
  AVF_REQUIRE(grid.y < 65536, "gemm_f32: M too large for grid");
  switch (a.epilogue) {
    case AVF_EPI_NONE: gemm_f32_kernel<AVF_EPI_NONE><<<grid, 256, 0, s>>>(p); break;
    case AVF_EPI_BIAS_RES:
      AVF_REQUIRE(a.residual, "gemm_f32: residual missing");
      gemm_f32_kernel<AVF_EPI_BIAS_RES><<<grid, 256, 0, s>>>(p);
      break;
    case AVF_EPI_BIAS_GELU:
      AVF_REQUIRE(a.aux, "gemm_f32: aux missing");
      gemm_f32_kernel<AVF_EPI_BIAS_GELU><<<grid, 256, 0, s>>>(p);
      break;
    case AVF_EPI_DGELU:
      AVF_REQUIRE(a.aux, "gemm_f32: aux missing");
      gemm_f32_kernel<AVF_EPI_DGELU><<<grid, 256, 0, s>>>(p);
      break;
    default: AVF_REQUIRE(false, "gemm_f32: bad epilogue %d", a.epilogue);
  }
  return check_launch("gemm_f32_kernel");
}

}  // namespace avf
