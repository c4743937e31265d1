// gemm_bf16.hip - throughput-mode GEMMs on v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
//
//   NT:  C[M,N] = A[M,K] * B[N,K]^T      nn.Linear forward (heads.py:191,195,212,215) and, with the
//                                         pre-transposed bf16 weight copy as B, the dX GEMMs.
//        Fused epilogues: +bias, +fp32 residual (heads.py:175), tanh-GELU (heads.py:166), dGELU.
//   TN:  C[M,N] = A[K,M]^T * B[K,N]      weight gradients dW = dY^T X (reduction over tokens), fp32
//        out, split-K over the token axis with partial slabs folded by a second kernel.
//
// 128x128 block tile, 4 wavefronts (2x2), 64x64 per wave = 4x4 MFMA tiles, K-step 64, LDS double
// buffered, global->register->LDS staging issued before the MFMA phase and written after it.
//
// Fragment maps (checked on hardware by avf_selftest_mfma_bf16 / avf_selftest_tr16):
//   A_op[i=l&15][k=8(l>>4)+j], B_op[k=8(l>>4)+j][n=l&15], D col=l&15,row=4(l>>4)+r.
// The NT kernel issues mfma(Bfrag, Afrag) so that a lane ends up with 4 CONSECUTIVE n for one m
// (16-byte fp32 / 8-byte bf16 stores, float4 bias/residual loads).
#include "common.hpp"

namespace avf {

namespace {

typedef __attribute__((address_space(3))) char lds_char;

// ------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------
constexpr int TB = 128;            // block tile (both M and N)
constexpr int TK = 64;             // K per stage (128 bytes per tile row)
constexpr int NT_STAGE = TB * TK * 2;  // bytes per operand per stage = 16 KiB

struct NtParams {
  const bf16* A;
  int64_t lda;
  const bf16* B;
  int64_t ldb;
  void* C;
  int64_t ldc;
  const float* bias;
  const float* residual;
  int64_t ldres;
  bf16* aux;
  int64_t ldaux;
  int M, N, K;
};

// 16-byte chunk c (0..7) of tile row r lives at chunk slot c ^ (r & 7): conflict-free ds_read_b128
__device__ __forceinline__ int nt_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

template <int EPI, typename CT>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(NtParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * NT_STAGE];  // [stage][A|B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
  const int sc = tid & 7, sr = tid >> 3;  // staging: chunk, row (+32*i)

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto issue = [&](int k0) {
    const int kc = k0 + sc * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 32 * i;
      ra[i] = make_uint4(0, 0, 0, 0);
      rb[i] = make_uint4(0, 0, 0, 0);
      if (kc < p.K) {
        if (m0 + r < p.M) ra[i] = *reinterpret_cast<const uint4*>(p.A + (int64_t)(m0 + r) * p.lda + kc);
        if (n0 + r < p.N) rb[i] = *reinterpret_cast<const uint4*>(p.B + (int64_t)(n0 + r) * p.ldb + kc);
      }
    }
  };
  auto commit = [&](int stage) {
    char* sa = smem + stage * 2 * NT_STAGE;
    char* sb = sa + NT_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<uint4*>(sa + nt_off(r, sc)) = ra[i];
      *reinterpret_cast<uint4*>(sb + nt_off(r, sc)) = rb[i];
    }
  };

  const int nt = (p.K + TK - 1) / TK;
  issue(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) issue((t + 1) * TK);
    const char* sa = smem + cur * 2 * NT_STAGE;
    const char* sb = sa + NT_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + i * 16 + li, rb_ = wn * 64 + i * 16 + li;
        fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + nt_off(ra_, ks * 4 + lg));
        fb[i] = *reinterpret_cast<const bf16x8_t*>(sb + nt_off(rb_, ks * 4 + lg));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) commit(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds C[m = ..+li][n = ..+4*lg + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (EPI == AVF_EPI_BIAS_RES) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldres + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      } else if (EPI == AVF_EPI_BIAS_GELU) {
        store4<bf16>(p.aux + (int64_t)m * p.ldaux + n, make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
      } else if (EPI == AVF_EPI_DGELU) {
        const float4 u = load4<bf16>(p.aux + (int64_t)m * p.ldaux + n);
        v[0] *= dgelu_tanh_f(u.x); v[1] *= dgelu_tanh_f(u.y); v[2] *= dgelu_tanh_f(u.z); v[3] *= dgelu_tanh_f(u.w);
      }
      store4<CT>((CT*)p.C + (int64_t)m * p.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
    }
  }
}

// ------------------------------------------------------------------------------------------
// TN kernel (weight gradients)
// ------------------------------------------------------------------------------------------
constexpr int TR = 64;                 // reduction rows (tokens) per stage
constexpr int TN_LD = 256 + 32;        // bytes per staged row: 128 bf16 + 32 B pad -> consecutive rows shift 8 banks
constexpr int TN_STAGE = TR * TN_LD;   // bytes per operand per stage = 18 KiB
constexpr int TN_SMEM = 4 * TN_STAGE;  // 72 KiB

struct TnParams {
  const bf16* A;  // [K, M]
  int64_t lda;
  const bf16* B;  // [K, N]
  int64_t ldb;
  float* C;       // [M, N] or slabs [S][M][N]
  int64_t ldc;
  int64_t slab;   // elements per slab (0 when writing C directly)
  int M, N, K;
  int kchunk;     // reduction rows per split (multiple of TR)
};

// transposed fragment: 8 reduction rows x 16 columns -> lane (col = cb + li) gets rows
// k-slot j: 16*(j>>2) + 4*lg + (j&3) of the 32-row k-step (same slot map for both operands).
__device__ __forceinline__ bf16x8_t tr_frag(const lds_char* tile, int row_base, int col_base, int li, int lg) {
  const lds_char* p0 = tile + (row_base + 4 * lg + (li >> 2)) * TN_LD + (col_base + 4 * (li & 3)) * 2;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * TN_LD));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

__global__ __launch_bounds__(256) void gemm_bf16_tn_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = (kbeg + p.kchunk) < p.K ? (kbeg + p.kchunk) : p.K;
  const int sc = tid & 15, sr = tid >> 4;  // staging: 16-byte chunk (0..15), row (+16*i)

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = k0 + sr + 16 * i;
      ra[i] = make_uint4(0, 0, 0, 0);
      rb[i] = make_uint4(0, 0, 0, 0);
      if (r < kend) {
        if (m0 + sc * 8 < p.M) ra[i] = *reinterpret_cast<const uint4*>(p.A + (int64_t)r * p.lda + m0 + sc * 8);
        if (n0 + sc * 8 < p.N) rb[i] = *reinterpret_cast<const uint4*>(p.B + (int64_t)r * p.ldb + n0 + sc * 8);
      }
    }
  };
  auto commit = [&](int stage) {
    char* sa = dyn_smem + stage * 2 * TN_STAGE;
    char* sb = sa + TN_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 16 * i;
      *reinterpret_cast<uint4*>(sa + r * TN_LD + sc * 16) = ra[i];
      *reinterpret_cast<uint4*>(sb + r * TN_LD + sc * 16) = rb[i];
    }
  };

  const int nt = (kend - kbeg + TR - 1) / TR;
  if (nt > 0) {
    issue(kbeg);
    commit(0);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) issue(kbeg + (t + 1) * TR);
    const lds_char* sa = (const lds_char*)(dyn_smem + cur * 2 * TN_STAGE);
    const lds_char* sb = sa + TN_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = tr_frag(sa, ks * 32, wm * 64 + i * 16, li, lg);
        fb[i] = tr_frag(sb, ks * 32, wn * 64 + i * 16, li, lg);
      }
      // D[i = n][j = m]: A_op rows = N-side columns, B_op cols = M-side columns
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) commit(cur ^ 1);
    __syncthreads();
  }

  float* out = p.C + (p.slab ? (int64_t)blockIdx.z * p.slab : 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      *reinterpret_cast<float4*>(out + (int64_t)m * p.ldc + n) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

__global__ __launch_bounds__(256) void fold_slabs_kernel(const float* __restrict__ slabs, int S, int64_t slab,
                                                         float* __restrict__ out, int64_t n4) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(slabs)[i];
    for (int s = 1; s < S; ++s) {
      const float4 b = reinterpret_cast<const float4*>(slabs + (int64_t)s * slab)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

int tn_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ceil_div(M, TB) * ceil_div(N, TB);
  int64_t s = ceil_div(512, tiles);
  const int64_t maxs = K / 256 > 0 ? K / 256 : 1;  // at least 256 reduction rows per split
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

size_t gemm_bf16_tn_ws(int64_t M, int64_t N, int64_t K) {
  const int s = tn_splits(M, N, K);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

int gemm_bf16_nt(const GemmArgs& a, hipStream_t s) {
  AVF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_bf16_nt: bad shape");
  AVF_REQUIRE(a.K % 8 == 0 && a.N % 4 == 0, "gemm_bf16_nt: K%%8 and N%%4 must be 0 (K=%lld N=%lld)", (long long)a.K,
              (long long)a.N);
  AVF_REQUIRE(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 4 == 0, "gemm_bf16_nt: leading dimensions must be 16-byte multiples");
  AVF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && ((uintptr_t)a.C & 15) == 0,
              "gemm_bf16_nt: operands must be 16-byte aligned");
  AVF_REQUIRE(a.M < (1LL << 31) && a.N < (1LL << 31) && a.K < (1LL << 31), "gemm_bf16_nt: shape too large");
  NtParams p;
  TimingScope ts(KC_GEMM_BF16_NT, 2.0 * a.M * a.N * a.K,
                 2.0 * (a.M * a.K + a.N * a.K) + (a.c_dtype == AVF_F32 ? 4.0 : 2.0) * a.M * a.N, s);
  p.A = (const bf16*)a.A; p.lda = a.lda; p.B = (const bf16*)a.B; p.ldb = a.ldb;
  p.C = a.C; p.ldc = a.ldc; p.bias = a.bias; p.residual = a.residual; p.ldres = a.ldres;
  p.aux = (bf16*)a.aux; p.ldaux = a.ldaux;
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  dim3 grid((unsigned)ceil_div(a.N, TB), (unsigned)ceil_div(a.M, TB));
  AVF_REQUIRE(grid.y < 65536, "gemm_bf16_nt: M too large for grid");
  const bool cf32 = a.c_dtype == AVF_F32;
  AVF_REQUIRE(cf32 || a.c_dtype == AVF_BF16, "gemm_bf16_nt: bad c_dtype");
#define LAUNCH(E)                                                         \
  do {                                                                    \
    if (cf32) gemm_bf16_nt_kernel<E, float><<<grid, 256, 0, s>>>(p);      \
    else gemm_bf16_nt_kernel<E, bf16><<<grid, 256, 0, s>>>(p);            \
  } while (0)
  switch (a.epilogue) {
    case AVF_EPI_NONE: LAUNCH(AVF_EPI_NONE); break;
    case AVF_EPI_BIAS_RES:
      AVF_REQUIRE(a.residual && cf32 && a.ldres % 4 == 0, "gemm_bf16_nt: BIAS_RES needs fp32 C and residual");
      LAUNCH(AVF_EPI_BIAS_RES);
      break;
    case AVF_EPI_BIAS_GELU:
      AVF_REQUIRE(a.aux && a.ldaux % 4 == 0, "gemm_bf16_nt: aux missing");
      LAUNCH(AVF_EPI_BIAS_GELU);
      break;
    case AVF_EPI_DGELU:
      AVF_REQUIRE(a.aux && a.ldaux % 4 == 0, "gemm_bf16_nt: aux missing");
      LAUNCH(AVF_EPI_DGELU);
      break;
    default: AVF_REQUIRE(false, "gemm_bf16_nt: bad epilogue %d", a.epilogue);
  }
#undef LAUNCH
  return check_launch("gemm_bf16_nt_kernel");
}

int gemm_bf16_tn(const GemmArgs& a, hipStream_t s) {
  AVF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_bf16_tn: bad shape");
  AVF_REQUIRE(a.c_dtype == AVF_F32 && a.epilogue == AVF_EPI_NONE && !a.bias, "gemm_bf16_tn: fp32 output, no epilogue");
  AVF_REQUIRE(a.M % 8 == 0 && a.N % 8 == 0, "gemm_bf16_tn: M%%8 and N%%8 must be 0 (M=%lld N=%lld)", (long long)a.M,
              (long long)a.N);
  AVF_REQUIRE(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 4 == 0, "gemm_bf16_tn: leading dimensions must be 16-byte multiples");
  AVF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && ((uintptr_t)a.C & 15) == 0,
              "gemm_bf16_tn: operands must be 16-byte aligned");
  AVF_REQUIRE(a.M < (1LL << 31) && a.N < (1LL << 31) && a.K < (1LL << 31), "gemm_bf16_tn: shape too large");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM);
    AVF_REQUIRE(e == hipSuccess, "gemm_bf16_tn: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int S = tn_splits(a.M, a.N, a.K);
  AVF_REQUIRE(S == 1 || a.workspace, "gemm_bf16_tn: split-K workspace missing");
  TnParams p;
  TimingScope ts(KC_GEMM_BF16_TN, 2.0 * a.M * a.N * a.K, 2.0 * (a.M * a.K + a.N * a.K) + 4.0 * a.M * a.N, s);
  p.A = (const bf16*)a.A; p.lda = a.lda; p.B = (const bf16*)a.B; p.ldb = a.ldb;
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  p.kchunk = (int)(ceil_div(ceil_div(a.K, S), TR) * TR);
  if (S > 1) { p.C = (float*)a.workspace; p.ldc = a.N; p.slab = a.M * a.N; }
  else { p.C = (float*)a.C; p.ldc = a.ldc; p.slab = 0; }
  dim3 grid((unsigned)ceil_div(a.N, TB), (unsigned)ceil_div(a.M, TB), (unsigned)S);
  gemm_bf16_tn_kernel<<<grid, 256, TN_SMEM, s>>>(p);
  AVF_TRY(check_launch("gemm_bf16_tn_kernel"));
  if (S > 1) {
    AVF_REQUIRE(a.ldc == a.N, "gemm_bf16_tn: split-K path needs a dense C (ldc == N)");
    const int64_t n4 = a.M * a.N / 4;
    int64_t blocks = ceil_div(n4, 256);
    if (blocks > 2048) blocks = 2048;
    fold_slabs_kernel<<<(unsigned)blocks, 256, 0, s>>>((const float*)a.workspace, S, a.M * a.N, (float*)a.C, n4);
    AVF_TRY(check_launch("fold_slabs_kernel"));
  }
  return 0;
}

}  // namespace avf
