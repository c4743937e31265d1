// gemm_mx8.hip - MX-FP8 (OCP microscaling: e4m3 elements, one E8M0 scale per 32 consecutive K) operands on
// v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulate: the CDNA4 fp8 path of BASELINE config 5 for the forward
// nn.Linear GEMMs of the layer (heads.py:191,195,212).
//
//   quant:  x[R,K] (bf16 | fp32)  ->  q[R,K] e4m3 bytes + s[R,K/32] E8M0 bytes
//           s = floor(log2(amax of the block)) - 8 (+127), q = rne_e4m3(clamp(x * 2^-(s-127), +-448))     (OCP MX v1.0 6.3)
//   NT:     C[M,N] = (A_q, A_s)[M,K] * (B_q, B_s)[N,K]^T with the epilogues of the bf16 NT kernel (gemm_nt.hpp)
//
// Operand map of the instruction, pinned on hardware by tools/diag/mfma_fp8_probe.hip: lane (i = l & 15, g = l >> 4)
// holds row i's bytes k = 16 g .. 16 g + 15 and k = 64 + 16 g .. 64 + 16 g + 15 - the 16-byte chunks g and 4 + g of a
// 128-byte tile row, which is the same pair of ds_read_b128 the bf16 kernel issues for its two k-halves, so the LDS
// image, its XOR swizzle and the LDS-DMA source permutation are shared with gemm_bf16.hip - and supplies the scale
// byte of block k = 32 g .. 32 g + 31 of its row.  A K-step is therefore 128 deep at the LDS bytes, DMA instructions
// and fragment reads of a 64-deep bf16 step, with half as many (twice as long) MFMAs per unit of K.
// The scale bytes of a K-step ([row][4], one dword per row) ride along as one 4-byte LDS-DMA per 64 rows.
//
// The matrix core adds the 128 products of one instruction with a shared alignment (terms more than ~2^17 below the
// largest are truncated; measured by tools/diag/mfma_fp8_probe3.hip), so results differ from an fp32 dot product of the
// dequantised operands by up to ~1e-3 of the largest product: the tests state that tolerance.
#include "common.hpp"
#include "gemm_nt.hpp"

namespace avf {

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));

struct Mx8Params {
  NtParams nt;        // A, B are byte images here; lda, ldb in bytes
  const uint8_t* As;  // [M][K/32]
  const uint8_t* Bs;  // [N][K/32]
};

__device__ __forceinline__ void glds4(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 4, 0, 0);
}

// LEAN (bit flags): 0 = the general epilogue; 1 = nt_epilogue_lean (gemm_nt.hpp; the host has checked nt_lean_ok), + 2 = with
// column sums, + 4 = with the MX-FP8 image of C
template <int EPI, typename CT, int WM, int WN, int MI, int NI, int LEAN = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_mx8_nt_kernel(Mx8Params q, int tiles_n, int nwg) {
  const NtParams& p = q.nt;
  constexpr int WTM = 16 * MI, WTN = 16 * NI;
  constexpr int BMT = WTM * WM, BNT = WTN * WN, NW = WM * WN;

  constexpr int A_BYTES = BMT * 128, B_BYTES = BNT * 128;
  constexpr int S_PIECES = (BMT + BNT + 63) / 64;       // scale dwords: 64 rows per 4-byte DMA instruction
  constexpr int S_BYTES = S_PIECES * 256;               // [A rows | B rows], padded to whole pieces
  constexpr int S_INS = (S_PIECES + NW - 1) / NW;       // per wave (pieces wrap: a duplicate writes the same bytes)
  constexpr int STAGE = A_BYTES + B_BYTES + S_BYTES;
  constexpr int A_TOT = BMT / 8, B_TOT = BNT / 8;  // 8-row DMA instructions, dealt round-robin to the waves
  constexpr int A_INS = (A_TOT + NW - 1) / NW, B_INS = (B_TOT + NW - 1) / NW;
  constexpr int KS = 128;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;
  const int kb = p.K >> 5;  // scale bytes per row

  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const char* ga[A_INS];
  const char* gb[B_INS];
  const uint8_t* gs[S_INS];
  int spiece[S_INS];
#pragma unroll
  for (int j = 0; j < A_INS; ++j) {
    int r = m0 + (wave + j * NW) * 8 + lrow;
    r = r < p.M ? r : p.M - 1;
    ga[j] = (const char*)p.A + (int64_t)r * p.lda + lchunk * 16;
  }
#pragma unroll
  for (int j = 0; j < B_INS; ++j) {
    int r = n0 + (wave + j * NW) * 8 + lrow;
    r = r < p.N ? r : p.N - 1;
    gb[j] = (const char*)p.B + (int64_t)r * p.ldb + lchunk * 16;
  }
#pragma unroll
  for (int j = 0; j < S_INS; ++j) {
    spiece[j] = (wave * S_INS + j) % S_PIECES;
    const int rho = spiece[j] * 64 + lane;  // row of the combined [A | B] scale block
    if (rho < BMT) {
      int r = m0 + rho;
      r = r < p.M ? r : p.M - 1;
      gs[j] = q.As + (int64_t)r * kb;
    } else {
      int r = n0 + rho - BMT;
      r = r < n0 + BNT ? r : n0 + BNT - 1;  // padding lanes of the last piece
      r = r < p.N ? r : p.N - 1;
      gs[j] = q.Bs + (int64_t)r * kb;
    }
  }
  auto stage = [&](int st, int t) {
    char* sa = dsm + st * STAGE;
    char* sb = sa + A_BYTES;
    char* ss = sb + B_BYTES;
#pragma unroll
    for (int j = 0; j < A_INS; ++j)
      if (A_TOT % NW == 0 || wave + j * NW < A_TOT) glds16(ga[j] + t * KS, sa + (wave + j * NW) * 1024);
#pragma unroll
    for (int j = 0; j < B_INS; ++j)
      if (B_TOT % NW == 0 || wave + j * NW < B_TOT) glds16(gb[j] + t * KS, sb + (wave + j * NW) * 1024);
#pragma unroll
    for (int j = 0; j < S_INS; ++j) glds4(gs[j] + t * 4, ss + spiece[j] * 256);
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // LDS byte addresses of this lane's fragment rows / scale dwords in stage 0 (row blocks are 2048 / 64 bytes apart)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  uint32_t abase[2], bbase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    abase[h] = lds0 + nt_off(wm * WTM + li, h * 4 + lg);
    bbase[h] = lds0 + A_BYTES + nt_off(wn * WTN + li, h * 4 + lg);
  }
  const uint32_t sabase = lds0 + A_BYTES + B_BYTES + (wm * WTM + li) * 4;
  const uint32_t sbbase = lds0 + A_BYTES + B_BYTES + (BMT + wn * WTN + li) * 4;

  const int nt = p.K / KS;
  stage(0, 0);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) stage(cur ^ 1, t + 1);
    // fragment and scale reads as inline asm (gemm_nt.hpp): the DMA of tile t+1 stays in flight under them
    const uint32_t so = (uint32_t)cur * STAGE;
    i32x4_t fa[2][MI], fb[2][NI];
    uint32_t sca[MI], scb[NI];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      lds_read_frags<i32x4_t, 2048>(fa[h], abase[h] + so, std::make_integer_sequence<int, MI>{});
      lds_read_frags<i32x4_t, 2048>(fb[h], bbase[h] + so, std::make_integer_sequence<int, NI>{});
    }
    lds_read_words<64>(sca, sabase + so, std::make_integer_sequence<int, MI>{});
    lds_read_words<64>(scb, sbbase + so, std::make_integer_sequence<int, NI>{});
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) sca[i] = (sca[i] >> (8 * lg)) & 255u;
#pragma unroll
    for (int j = 0; j < NI; ++j) scb[j] = (scb[j] >> (8 * lg)) & 255u;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const v8i_t va = {fa[0][i][0], fa[0][i][1], fa[0][i][2], fa[0][i][3], fa[1][i][0], fa[1][i][1], fa[1][i][2], fa[1][i][3]};
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const v8i_t vb = {fb[0][j][0], fb[0][j][1], fb[0][j][2], fb[0][j][3], fb[1][j][0], fb[1][j][1], fb[1][j][2], fb[1][j][3]};
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vb, va, acc[i][j], 0, 0, 0, (int)scb[j], 0, (int)sca[i]);
      }
    }
    cur ^= 1;
  }

  if constexpr (LEAN != 0)
    nt_epilogue_lean<EPI, CT, MI, NI, (LEAN & 2) ? 1 : 0, false, (LEAN & 4) != 0>(p, acc, m0 + wm * WTM, n0 + wn * WTN, li, lg,
                                                                                (wg / tiles_n) * WM + wm);
  else
    nt_epilogue<EPI, CT, MI, NI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, li, lg,
                                 p.cs_partial ? (wg / tiles_n) * WM + wm : -1);
}

template <int EPI, typename CT, int WM, int WN, int MI, int NI, int LEAN = 0>
int launch_mx8(const Mx8Params& q, hipStream_t s, int* part_rows, TimingScope* ts) {
  constexpr int BMT = 16 * MI * WM, BNT = 16 * NI * WN;
  constexpr int SMEM = 2 * ((BMT + BNT) * 128 + ((BMT + BNT + 63) / 64) * 256);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static PerDeviceOnce raised;
  if (SMEM > 64 * 1024 && raised.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_mx8_nt_kernel<EPI, CT, WM, WN, MI, NI, LEAN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    AVF_REQUIRE(e == hipSuccess, "gemm_mx8_nt: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    raised.mark();
  }
  const int tiles_m = (q.nt.M + BMT - 1) / BMT, tiles_n = (q.nt.N + BNT - 1) / BNT;
  const int nwg = tiles_m * tiles_n;
  *part_rows = tiles_m * WM;
  if (shape_log_on()) {
    const NtParams& p = q.nt;
    const double csz = sizeof(CT);
    const double epi_b = (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) ? csz * p.M * p.N : 0.0;
    shape_log("gemm_mx8_nt,gemm_mx8_nt_kernel<%d, %s, %d, %d, %d, %d, %d>,%d,%d,%d,%d,%d,%.0f,%.0f", EPI,
              sizeof(CT) == 4 ? "float" : "bf16", WM, WN, MI, NI, LEAN, nwg, p.M, p.N, p.K, EPI + (p.mxq ? 10 : 0), 2.0 * p.M * p.N * p.K,
              ((double)p.M * p.K + (double)p.N * p.K) * (1.0 + 1.0 / 32) + csz * p.M * p.N + epi_b + (p.mxq ? (double)p.M * p.N * (1.0 + 1.0 / 32) : 0.0));
  }
  launch_in_scope(ts, gemm_mx8_nt_kernel<EPI, CT, WM, WN, MI, NI, LEAN>, dim3(nwg), dim3(WM * WN * 64), SMEM, s, q, tiles_n, nwg);
  return 0;
}

template <int EPI, typename CT>
int launch_mx8_any(const Mx8Params& q, hipStream_t s, int* part_rows, TimingScope* ts) {
  const int tile = pick_nt_tile(q.nt.M, q.nt.N, q.nt.K / 2);  // K/2: the same LDS bytes per row as a bf16 problem of that depth
  // the two 8-wave tiles with the lean epilogue (options fixed at compile time) when nothing asks for the general one's
  if ((tile == 2 || tile == 5) && nt_lean_ok<EPI, CT>(q.nt, 128, /*mx_ok=*/true)) {
    constexpr bool can_mx = EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU;
    constexpr bool can_cs = EPI == AVF_EPI_DGELU;
    const bool mx = q.nt.mxq != nullptr, cs = q.nt.cs_partial != nullptr;
#define AVF_MX8_LEAN(F)                                                                 \
    do {                                                                                \
      if (tile == 5) return launch_mx8<EPI, CT, 2, 4, 3, 2, F>(q, s, part_rows, ts);    \
      return launch_mx8<EPI, CT, 2, 4, 4, 2, F>(q, s, part_rows, ts);                   \
    } while (0)
    if (!mx && !cs) AVF_MX8_LEAN(1);
    if constexpr (can_mx) {
      if (mx && !cs) AVF_MX8_LEAN(5);
    }
    if constexpr (can_cs) {
      if (!mx && cs) AVF_MX8_LEAN(3);
      if (mx && cs) AVF_MX8_LEAN(7);
    }
#undef AVF_MX8_LEAN
  }
  switch (tile) {
    case 0: return launch_mx8<EPI, CT, 2, 2, 4, 4>(q, s, part_rows, ts);
    case 1: return launch_mx8<EPI, CT, 2, 2, 2, 4>(q, s, part_rows, ts);
    case 3: return launch_mx8<EPI, CT, 2, 2, 3, 4>(q, s, part_rows, ts);
    case 5: return launch_mx8<EPI, CT, 2, 4, 3, 2>(q, s, part_rows, ts);
    default: return launch_mx8<EPI, CT, 2, 4, 4, 2>(q, s, part_rows, ts);
  }
}

// ------------------------------------------------------------------------------------------
// quantiser: 8 elements per lane, 4 lanes per 32-block
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const T* __restrict__ x, int64_t ldx, uint8_t* __restrict__ qo,
                                                        int64_t ldq, uint8_t* __restrict__ so, int64_t R, int K) {
  const int per_row = K >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = R * per_row;
  const bool live = idx < total;
  const int64_t row = live ? idx / per_row : 0;
  const int c8 = live ? (int)(idx - row * per_row) : 0;
  float v[8];
  load8(x + row * ldx + c8 * 8, v);
  const MxBlock b = mx8_encode(v);
  if (live) {
    *reinterpret_cast<uint2*>(qo + row * ldq + c8 * 8) = b.q;
    if ((c8 & 3) == 0) so[row * (K >> 5) + (c8 >> 2)] = (uint8_t)b.scale;
  }
}

// several bf16 matrices in one launch (the forward weight images of a stack, once per step)
constexpr int MX_MAX = 24;
struct MxDesc {
  const bf16* x;
  uint8_t* q;
  uint8_t* s;
  int R, K, block0;
};
struct MxBatch {
  MxDesc d[MX_MAX];
  int n;
};
__global__ __launch_bounds__(256) void quant_mx8_multi_kernel(MxBatch b) {
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.d[i + 1].block0) ++i;
  const MxDesc& d = b.d[i];
  const int per_row = d.K >> 3;
  const int64_t idx = (int64_t)((int)blockIdx.x - d.block0) * 256 + threadIdx.x;
  const int64_t total = (int64_t)d.R * per_row;
  const bool live = idx < total;
  const int64_t row = live ? idx / per_row : 0;
  const int c8 = live ? (int)(idx - row * per_row) : 0;
  float v[8];
  load8(d.x + row * d.K + c8 * 8, v);
  const MxBlock blk = mx8_encode(v);
  if (live) {
    *reinterpret_cast<uint2*>(d.q + row * d.K + c8 * 8) = blk.q;
    if ((c8 & 3) == 0) d.s[row * (d.K >> 5) + (c8 >> 2)] = (uint8_t)blk.scale;
  }
}

}  // namespace

int quant_mx8_multi(const MxQuantJob* jobs, int n, hipStream_t s) {
  for (int base = 0; base < n; base += MX_MAX) {
    MxBatch b;
    b.n = n - base < MX_MAX ? n - base : MX_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const MxQuantJob& j = jobs[base + i];
      AVF_REQUIRE(j.x && j.q && j.s && j.R > 0 && j.K > 0 && j.K % 32 == 0 && j.R * j.K < (1LL << 31),
                  "quant_mx8_multi: bad job %d", base + i);
      b.d[i] = MxDesc{(const bf16*)j.x, (uint8_t*)j.q, (uint8_t*)j.s, (int)j.R, (int)j.K, blocks};
      blocks += (int)ceil_div(j.R * (j.K >> 3), (int64_t)256);
    }
    quant_mx8_multi_kernel<<<blocks, 256, 0, s>>>(b);
    AVF_TRY(check_launch("quant_mx8_multi_kernel"));
  }
  return 0;
}

int quant_mx8(const void* x, int dtype, int64_t ldx, int64_t R, int64_t K, void* q, int64_t ldq, void* scales, hipStream_t s) {
  AVF_REQUIRE(R > 0 && K > 0 && K % 32 == 0 && K < (1LL << 31), "quant_mx8: K must be a positive multiple of 32");
  AVF_REQUIRE(dtype == AVF_F32 || dtype == AVF_BF16, "quant_mx8: source must be f32 or bf16");
  AVF_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0,
              "quant_mx8: rows must be 16-byte (source) / 8-byte (image) aligned");
  const int64_t total = R * (K >> 3);
  const unsigned blocks = (unsigned)ceil_div(total, (int64_t)256);
  if (dtype == AVF_F32)
    quant_mx8_kernel<float><<<blocks, 256, 0, s>>>((const float*)x, ldx, (uint8_t*)q, ldq, (uint8_t*)scales, R, (int)K);
  else
    quant_mx8_kernel<bf16><<<blocks, 256, 0, s>>>((const bf16*)x, ldx, (uint8_t*)q, ldq, (uint8_t*)scales, R, (int)K);
  return check_launch("quant_mx8_kernel");
}

int gemm_mx8_nt(const GemmArgs& a, const void* a_scales, const void* b_scales, hipStream_t s, void* mx_q, void* mx_s) {
  AVF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_mx8_nt: bad shape");
  AVF_REQUIRE(a.K % 128 == 0 && a.N % 4 == 0, "gemm_mx8_nt: K%%128 and N%%4 must be 0 (K=%lld N=%lld)", (long long)a.K,
              (long long)a.N);
  AVF_REQUIRE(a.lda % 16 == 0 && a.ldb % 16 == 0 && a.ldc % 4 == 0, "gemm_mx8_nt: leading dimensions must be 16-byte multiples");
  AVF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && ((uintptr_t)a.C & 15) == 0,
              "gemm_mx8_nt: operands must be 16-byte aligned");
  AVF_REQUIRE(a_scales && b_scales && ((uintptr_t)a_scales & 3) == 0 && ((uintptr_t)b_scales & 3) == 0,
              "gemm_mx8_nt: scale images missing or not 4-byte aligned");
  AVF_REQUIRE(a.M < (1LL << 31) && a.N < (1LL << 31) && a.K < (1LL << 31), "gemm_mx8_nt: shape too large");
  Mx8Params q;
  NtParams& p = q.nt;
  const double csz = a.c_dtype == AVF_F32 ? 4.0 : 2.0;
  const double epi_bytes = a.epilogue == AVF_EPI_NONE ? 0.0 : csz * a.M * a.N;  // residual / saved pre-activation
  TimingScope ts(KC_GEMM_MX8_NT, 2.0 * a.M * a.N * a.K,
                 1.0 * (a.M * a.K + a.N * a.K) * (1.0 + 1.0 / 32) + csz * a.M * a.N + epi_bytes + (mx_q ? a.M * a.N * (1.0 + 1.0 / 32) : 0.0),
                 s, /*per_kernel=*/true);
  p.A = (const bf16*)a.A; p.lda = a.lda; p.B = (const bf16*)a.B; p.ldb = a.ldb;
  p.C = a.C; p.ldc = a.ldc; p.bias = a.bias; p.residual = a.residual; p.ldres = a.ldres;
  p.aux = a.aux; p.ldaux = a.ldaux;
  p.drop = a.drop;
  p.mxq = (uint8_t*)mx_q; p.mxs = (uint8_t*)mx_s;
  p.wide = nt_wide_stores();
  AVF_REQUIRE(!mx_q || (mx_s && (a.epilogue == AVF_EPI_BIAS_GELU || a.epilogue == AVF_EPI_DGELU) && a.N % 32 == 0 &&
                        ((uintptr_t)mx_q & 7) == 0),
              "gemm_mx8_nt: the MX-FP8 output image needs the BIAS_GELU / DGELU epilogue and N %% 32 == 0");
  AVF_REQUIRE(!a.drop.thresh16 || a.epilogue != AVF_EPI_NONE, "gemm_mx8_nt: dropout needs a fused epilogue");
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  q.As = (const uint8_t*)a_scales; q.Bs = (const uint8_t*)b_scales;
  const bool cf32 = a.c_dtype == AVF_F32;
  AVF_REQUIRE(cf32 || a.c_dtype == AVF_BF16, "gemm_mx8_nt: bad c_dtype");
  int part_rows = 0;
  p.cs_partial = nullptr;
  if (a.colsum) {
    AVF_REQUIRE(a.workspace, "gemm_mx8_nt: column-sum workspace missing");
    AVF_REQUIRE((size_t)ceil_div(a.M, 32) * a.N * sizeof(float) <= gemm_nt_colsum_ws(a.M, a.N),
                "gemm_mx8_nt: column-sum partials exceed their workspace (internal error)");
    p.cs_partial = (float*)a.workspace;
  }
#define LAUNCH(E)                                                     \
  do {                                                                \
    if (cf32) AVF_TRY((launch_mx8_any<E, float>(q, s, &part_rows, &ts)));   \
    else AVF_TRY((launch_mx8_any<E, bf16>(q, s, &part_rows, &ts)));        \
  } while (0)
  switch (a.epilogue) {
    case AVF_EPI_NONE: LAUNCH(AVF_EPI_NONE); break;
    case AVF_EPI_BIAS_RES:
      AVF_REQUIRE(a.residual && a.ldres % 4 == 0, "gemm_mx8_nt: BIAS_RES needs a residual (in C's storage type)");
      LAUNCH(AVF_EPI_BIAS_RES);
      break;
    case AVF_EPI_BIAS_GELU:
      AVF_REQUIRE(a.aux && a.ldaux % 4 == 0, "gemm_mx8_nt: aux missing");
      LAUNCH(AVF_EPI_BIAS_GELU);
      break;
    case AVF_EPI_DGELU:
      AVF_REQUIRE(a.aux && a.ldaux % 4 == 0, "gemm_mx8_nt: DGELU needs the saved pre-activation (in C's type)");
      LAUNCH(AVF_EPI_DGELU);
      break;
    default: AVF_REQUIRE(false, "gemm_mx8_nt: bad epilogue %d", a.epilogue);
  }
#undef LAUNCH
  AVF_TRY(check_launch("gemm_mx8_nt_kernel"));
  AVF_REQUIRE(!a.colsum || (size_t)part_rows * a.N * sizeof(float) <= gemm_nt_colsum_ws(a.M, a.N),
              "gemm_mx8_nt: column-sum partials exceed their workspace (internal error)");
  if (a.colsum) {
    if (a.defer_fold) *a.defer_fold = FoldJob{p.cs_partial, part_rows, (int)a.N, (int)a.N, a.colsum, nullptr, nullptr};
    else AVF_TRY(fold_partials(p.cs_partial, part_rows, (int)a.N, a.colsum, s));
  }
  return 0;
}

}  // namespace avf
