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


// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the same fp32 GEMM on the bf16 matrix pipe, operands split in three products ("bf16x3").
//
// gfx950 has no xf32 / tf32 MFMA: the f32-input MFMA above runs at 1/16 of the bf16 rate (157 TFLOP/s dense).  Here every fp32
// operand x is split while it is staged into LDS:  x = hi + lo + r,  hi = bf16(x) (RNE),  lo = bf16(x - hi)  (x - hi is exact in
// fp32: at most 16 significant bits remain); bf16 rounds to 8 significant bits (unit roundoff u = 2^-8), so |x - hi| <= u |x| and
// |r| <= u^2 |x| = 2^-16 |x|;  and  a b  ~  hi_a hi_b + hi_a lo_b + lo_a hi_b  on v_mfma_f32_16x16x32_bf16 with fp32 accumulation
// (every bf16 x bf16 product is exact in fp32).  Dropped: lo_a lo_b and the two residuals, each <= 2^-16 |a b|: a product carries
// <= 3 * 2^-16 = 4.6e-5 relative error in the worst case, ~4e-6 typically (measured: 4.4e-6 relative Frobenius on K = 512 ... 10368
// GEMMs; fp32: 4e-7) - oracle/bf16x3.py emulates exactly this arithmetic and the tests hold the kernel to it at 6e-7.  That is well
// inside north_star's logits rtol 1e-3, at 3 bf16 MFMAs per product: an effective roof of 2500 / 3 = 833 TFLOP/s against 157.
// (models/heads.py:191-196, 212, 214-217 and their autograd.)
//
// 128 x 128 tile, 32-deep K-step, 4 waves of 64 x 64 (4 x 4 MFMA blocks each, 48 MFMAs per K-step and wave); global -> registers
// (two K-steps of look-ahead) -> split -> LDS as four bf16 images [row][32 k] (64-byte rows, 16-byte chunks XOR-swizzled: fragment
// reads and stores conflict-free), double-buffered: one barrier per K-step, 64 KB per workgroup, two workgroups per CU.  Either operand may be "k-fast"
// (row-major along the reduction: 16-byte loads, K % 32 == 0) or "row-fast" (the transposed forms of dX / dW: lanes run along
// the rows, one dword per k, ragged K allowed).  The accumulators are kept TRANSPOSED (the weight side is the MFMA's A operand)
// so that a lane owns four consecutive columns of one row: 16-byte epilogue accesses.
constexpr int SBM = 128, SBN = 128, SBK = 32;
constexpr int SIMG = SBM * SBK;  // bf16 elements of one LDS image ([row][32 k], 64-byte rows, 16-byte chunks XOR-swizzled)

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// element offset of 16-byte chunk c (8 k) of row r.  ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... - MI355X_MICROARCH.md, LDS): a group holds all 16 fragment rows, rows 0-3 and
// 12-15 at k-chunk g, rows 4-11 at k-chunk g ^ 1, and rows r, r + 4, r + 8, r + 12 share the banks of a 256-byte line.  XORing
// the chunk with 2 * bit 2 of the row makes the four chunks of such a quadruple distinct (conflict-free fragment reads); bit 1
// of the row in the low bit makes the 16-byte stores of 8 consecutive rows (row-fast staging) conflict-free as well.
__device__ __forceinline__ int sw_off(int r, int c) { return r * SBK + ((c ^ ((((r >> 2) & 1) << 1) | ((r >> 1) & 1))) << 3); }

// KTAIL: the K range may end inside a K-step (both operands row-fast: the token reduction of a weight gradient); with a k-fast
// operand in the product K % 32 == 0 holds (f32x3_ok) and the row-fast side loads unconditionally
template <bool KF, bool KTAIL = true>
struct SplitStage {
  float v[16];  // fp32 values per thread and K-step
  // element (row, k) of the operand at X[row * s_r + k * s_k]; rows clamped (their results are never stored), k >= kend -> 0
  __device__ __forceinline__ void fetch(const float* X, int64_t s_r, int64_t s_k, int r0, int R, int k0, int kend) {
    const int tid = threadIdx.x;
    if constexpr (KF) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        const int row = idx >> 3, kq = idx & 7;
        int gr = r0 + row;
        gr = gr < R ? gr : R - 1;
        const float4 t = *reinterpret_cast<const float4*>(X + (int64_t)gr * s_r + k0 + kq * 4);
        v[4 * e] = t.x; v[4 * e + 1] = t.y; v[4 * e + 2] = t.z; v[4 * e + 3] = t.w;
      }
    } else {
      // lanes run along the rows; the k index is wave-uniform, so every load is (scalar row base) + (one 32-bit lane offset)
      const int row = tid & 127, kh = __builtin_amdgcn_readfirstlane(tid >> 7);
      int gr = r0 + row;
      gr = gr < R ? gr : R - 1;
      const float* base = X + (int64_t)(k0 + kh * 16) * s_k;
      if (!KTAIL || k0 + kh * 16 + 16 <= kend) {  // wave-uniform: every K-step but (possibly) the last loads unconditionally -
        // 16 predicated loads cost the mixed (NN) form 8 % and sit in 16 basic blocks
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (base + (int64_t)e * s_k)[(uint32_t)gr];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (k0 + kh * 16 + e < kend) ? (base + (int64_t)e * s_k)[(uint32_t)gr] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void commit(uint16_t* Xh, uint16_t* Xl) const {
    const int tid = threadIdx.x;
    if constexpr (KF) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        const int row = idx >> 3, kq = idx & 7;
        uint2 h, l;
        split2(v[4 * e], v[4 * e + 1], h.x, l.x);
        split2(v[4 * e + 2], v[4 * e + 3], h.y, l.y);
        const int off = sw_off(row, kq >> 1) + (kq & 1) * 4;
        *reinterpret_cast<uint2*>(Xh + off) = h;
        *reinterpret_cast<uint2*>(Xl + off) = l;
      }
    } else {
      const int row = tid & 127, kh = tid >> 7;
      uint32_t h[8], l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
      const int o0 = sw_off(row, 2 * kh), o1 = sw_off(row, 2 * kh + 1);
      *reinterpret_cast<uint4*>(Xh + o0) = make_uint4(h[0], h[1], h[2], h[3]);
      *reinterpret_cast<uint4*>(Xh + o1) = make_uint4(h[4], h[5], h[6], h[7]);
      *reinterpret_cast<uint4*>(Xl + o0) = make_uint4(l[0], l[1], l[2], l[3]);
      *reinterpret_cast<uint4*>(Xl + o1) = make_uint4(l[4], l[5], l[6], l[7]);
    }
  }
};

// XCD-aware tile order: workgroups go round-robin over the 8 XCDs (each with its own L2) in launch order, so the linear id is
// remapped to give every XCD a CONTIGUOUS run of tiles - the column tiles of one row panel (which share the A rows) and the
// neighbouring panels - instead of every eighth tile (cdna_hip_programming.md, T1)
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t id, uint32_t total) {
  const uint32_t xcd = id & 7, q = total >> 3, r = total & 7;
  return xcd * q + (xcd < r ? xcd : r) + (id >> 3);
}

// one 128 x 128 tile (bx, by) of C, K range bz (SPLIT): the whole kernel body; lds = 2 x 4 images (64 KB, two workgroups per CU)
template <bool AKF, bool BKF, int EPI, bool SPLIT>
__device__ __forceinline__ void f32x3_tile(const F32GemmParams& p, int vec, int bx, int by, int bz, uint16_t* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int m0 = by * SBM, n0 = bx * SBN;
  const int kbeg = SPLIT ? bz * p.kchunk : 0;
  const int kend = SPLIT ? ((kbeg + p.kchunk) < p.K ? kbeg + p.kchunk : p.K) : p.K;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // A: element (m, k) at A[m a_sm + k a_sk];  B: element (k, n) at B[k b_sk + n b_sn] (its tile rows are n)
  constexpr bool KTAIL = !AKF && !BKF;
  using StageA = SplitStage<AKF, KTAIL>;
  using StageB = SplitStage<BKF, KTAIL>;
  StageA sa0, sa1;
  StageB sb0, sb1;
  auto fetch = [&](StageA& sa, StageB& sb, int k0) {
    sa.fetch(p.A, p.a_sm, p.a_sk, m0, p.M, k0, kend);
    sb.fetch(p.B, p.b_sn, p.b_sk, n0, p.N, k0, kend);
  };
  const int frag = sw_off(li, lg);  // this lane's 16-byte fragment inside a 16-row block (the swizzle depends on li only)
  // one K-step: the MFMAs of the step staged in `cur`, while the NEXT step's registers are split into `nxt` and the loads of
  // the step after the one in flight are issued - two K-steps of look-ahead on the global loads, one barrier per step
  auto step = [&](StageA& sa, StageB& sb, const uint16_t* cur, uint16_t* nxt, int k0) {
    if (k0 + SBK < kend) {
      sa.commit(nxt, nxt + SIMG);
      sb.commit(nxt + 2 * SIMG, nxt + 3 * SIMG);
    }
    if (k0 + 3 * SBK < kend) fetch(sa, sb, k0 + 3 * SBK);
    bf16x8_t ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (wm * 64 + i * 16) * SBK + frag;
      ah[i] = *reinterpret_cast<const bf16x8_t*>(cur + off);
      al[i] = *reinterpret_cast<const bf16x8_t*>(cur + SIMG + off);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = (wn * 64 + j * 16) * SBK + frag;
      const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(cur + 2 * SIMG + off);
      const bf16x8_t bl = *reinterpret_cast<const bf16x8_t*>(cur + 3 * SIMG + off);
      // D[n][m] += B-side rows (MFMA A operand) x A-side rows (MFMA B operand): the small terms first
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  uint16_t* buf0 = lds;
  uint16_t* buf1 = lds + 4 * SIMG;
  if (kbeg < kend) {
    fetch(sa0, sb0, kbeg);
    if (kbeg + SBK < kend) fetch(sa1, sb1, kbeg + SBK);
    sa0.commit(buf0, buf0 + SIMG);
    sb0.commit(buf0 + 2 * SIMG, buf0 + 3 * SIMG);
    if (kbeg + 2 * SBK < kend) fetch(sa0, sb0, kbeg + 2 * SBK);
    __syncthreads();
  }
  for (int k0 = kbeg; k0 < kend; k0 += 2 * SBK) {
    step(sa1, sb1, buf0, buf1, k0);
    if (k0 + SBK < kend) step(sa0, sb0, buf1, buf0, k0 + SBK);
  }

  // transposed accumulators: register r of lane (li, lg) of acc[i][j] = C[m0 + wm 64 + 16 i + li][n0 + wn 64 + 16 j + 4 lg + r]
  const uint64_t dkey = (!SPLIT && p.drop.thresh16) ? drop_key(p.drop) : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + wm * 64 + i * 16 + li;
    if (gm >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + wn * 64 + j * 16 + 4 * lg;
      if (gn >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (SPLIT) {
        float* dst = p.slabs + ((int64_t)bz * p.M + gm) * p.N + gn;
        if (vec) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (gn + r < p.N) dst[r] = v[r];
        continue;
      }
      if (vec) {  // N % 4 == 0 and every row pointer 16-byte aligned: gn + 3 < N
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + gn);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        float4 df = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.drop.thresh16) df = drop_factor4(p.drop, dkey, (uint64_t)gm * p.N + gn);
        const float dfv[4] = {df.x, df.y, df.z, df.w};
        if (EPI == AVF_EPI_BIAS_RES) {
          const float4 r4 = *reinterpret_cast<const float4*>(p.residual + (int64_t)gm * p.ldres + gn);
          const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * dfv[r] + rr[r];
        } else if (EPI == AVF_EPI_BIAS_GELU) {
          *reinterpret_cast<float4*>(p.aux + (int64_t)gm * p.ldaux + gn) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]) * dfv[r];  // (exp2 / rcp form, ~1e-6 relative: the class of this
          // kernel's products; tanhf() is ~40 instructions per value and cost this launch 23 of 84 us)
        } else if (EPI == AVF_EPI_DGELU) {
          const float4 u4 = *reinterpret_cast<const float4*>(p.aux + (int64_t)gm * p.ldaux + gn);
          const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= dfv[r] * dgelu_tanh_fast(uu[r]);
        }
        *reinterpret_cast<float4*>(p.C + (int64_t)gm * p.ldc + gn) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = gn + r;
          if (n >= p.N) continue;
          float x = v[r];
          if (p.bias) x += p.bias[n];
          const float df = p.drop.thresh16 ? drop_factor1(p.drop, dkey, (uint64_t)gm * p.N + n) : 1.0f;
          if (EPI == AVF_EPI_BIAS_RES) {
            x = x * df + p.residual[(int64_t)gm * p.ldres + n];
          } else if (EPI == AVF_EPI_BIAS_GELU) {
            p.aux[(int64_t)gm * p.ldaux + n] = x;
            x = gelu_tanh_fast(x) * df;
          } else if (EPI == AVF_EPI_DGELU) {
            x *= df * dgelu_tanh_fast(p.aux[(int64_t)gm * p.ldaux + n]);
          }
          p.C[(int64_t)gm * p.ldc + n] = x;
        }
      }
    }
  }
}


template <bool AKF, bool BKF, int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_f32x3_kernel(F32GemmParams p, int vec) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 4 * SIMG];
  const uint32_t gx = gridDim.x, gxy = gridDim.x * gridDim.y;
  const uint32_t nid = xcd_contiguous(blockIdx.x + gx * blockIdx.y + gxy * blockIdx.z, gxy * gridDim.z);
  const int bz = nid / gxy;
  const uint32_t rem = nid - bz * gxy;
  const int by = rem / gx;
  f32x3_tile<AKF, BKF, EPI, SPLIT>(p, vec, rem - by * gx, by, bz, lds);
}

// The weight gradients of one layer (up to four C_i[M_i, N_i] = A_i[K, M_i]^T B_i[K, N_i] sharing the token reduction K) as ONE
// launch: the tiles of all problems in one list, every tile split S ways over K, tiles x S = one round of the chip's workgroup slots.
// Four separate launches each chose their own split count to fill the chip (10 / 32 / 16 / 16 at C2) and wrote 130 MB of slabs per
// layer where this writes 34, and each ended in its own under-filled tail and its own fold launch.
struct F32GroupParams {
  F32GemmParams p[4];
  int tile0[5];   // first tile of problem i in the list (tile0[count] = all tiles)
  int64_t el0[5]; // first C element of problem i in the concatenated fold index space
  int count, S, vec;
};
__global__ __launch_bounds__(256, 2) void gemm_f32x3_group_kernel(F32GroupParams g) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 4 * SIMG];
  const uint32_t tiles = g.tile0[g.count];
  const uint32_t nid = xcd_contiguous(blockIdx.x, tiles * g.S);
  const int bz = nid / tiles;
  const int t = nid - bz * tiles;
  int i = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < g.count && t >= g.tile0[k]) i = k;
  const int local = t - g.tile0[i];
  const int gxi = (g.p[i].N + SBN - 1) / SBN;
  const int by = local / gxi;
  f32x3_tile<false, false, AVF_EPI_NONE, true>(g.p[i], g.vec, local - by * gxi, by, bz, lds);
}
// C_i = sum over the S slabs of problem i, 16 bytes per thread (every M_i N_i is a multiple of 4)
__global__ __launch_bounds__(256) void gemm_f32x3_group_fold_kernel(F32GroupParams g) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= g.el0[g.count]) return;
  int i = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < g.count && e >= g.el0[k]) i = k;
  const int64_t le = e - g.el0[i], mn = (int64_t)g.p[i].M * g.p[i].N;
  const float* sl = g.p[i].slabs + le;
  float4 v = *reinterpret_cast<const float4*>(sl);
  for (int z = 1; z < g.S; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(sl + (int64_t)z * mn);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const int m = (int)(le / g.p[i].N), n = (int)(le - (int64_t)m * g.p[i].N);  // (ldc may exceed N)
  *reinterpret_cast<float4*>(g.p[i].C + (int64_t)m * g.p[i].ldc + n) = v;
}

// which arithmetic AVF_F32 GEMMs and attention run on: 1 = bf16x3 on the bf16 matrix pipe (default), 0 = the f32-input MFMA
int g_f32_arith = 1;

// shapes the bf16x3 kernel takes: tiles worth filling, one unit stride per operand, 16-byte loads where k is the fast axis
bool f32x3_ok(const F32GemmParams& p) {
  if (p.M < 96 || p.N < 96 || p.K < 32) return false;  // at least three quarters of a tile each way (smaller: the 32 / 64 tiles above)
  const bool akf = p.a_sk == 1, bkf = p.b_sk == 1;
  if (!akf && p.a_sm != 1) return false;
  if (!bkf && p.b_sn != 1) return false;
  if (akf && (p.K % 32 != 0 || p.a_sm % 4 != 0 || ((uintptr_t)p.A & 15))) return false;
  if (bkf && (p.K % 32 != 0 || p.b_sn % 4 != 0 || ((uintptr_t)p.B & 15))) return false;
  return true;
}
// split count over K for the 128 x 128 tiles: as many splits as keep tiles x splits within ONE round of the chip's 512 workgroup
// slots (two per CU) - 48 tiles x 11 splits = 528 workgroups ran a second, almost empty round (119 us; 10 splits: 85) - with at
// least 8 K-steps per split
int f32x3_splits(int64_t M, int64_t N, int64_t K, int epilogue) {
  if (epilogue != AVF_EPI_NONE && epilogue != AVF_EPI_BIAS_RES) return 1;
  const int64_t tiles = ceil_div(M, SBM) * ceil_div(N, SBN);
  if (tiles >= 256 || K < 512) return 1;
  int64_t sp = 512 / tiles;
  if (sp > K / 256) sp = K / 256;
  if (sp > 32) sp = 32;
  return sp < 2 ? 1 : (int)sp;
}

template <bool AKF, bool BKF>
int launch_f32x3(const F32GemmParams& p, int epilogue, int S, int vec, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(p.N, SBN), (unsigned)ceil_div(p.M, SBM), (unsigned)S);
  if (S > 1) {
    gemm_f32x3_kernel<AKF, BKF, AVF_EPI_NONE, true><<<grid, 256, 0, s>>>(p, vec);
    return check_launch("gemm_f32x3_kernel(split)");
  }
  switch (epilogue) {
    case AVF_EPI_NONE: gemm_f32x3_kernel<AKF, BKF, AVF_EPI_NONE, false><<<grid, 256, 0, s>>>(p, vec); break;
    case AVF_EPI_BIAS_RES: gemm_f32x3_kernel<AKF, BKF, AVF_EPI_BIAS_RES, false><<<grid, 256, 0, s>>>(p, vec); break;
    case AVF_EPI_BIAS_GELU: gemm_f32x3_kernel<AKF, BKF, AVF_EPI_BIAS_GELU, false><<<grid, 256, 0, s>>>(p, vec); break;
    case AVF_EPI_DGELU: gemm_f32x3_kernel<AKF, BKF, AVF_EPI_DGELU, false><<<grid, 256, 0, s>>>(p, vec); break;
    default: AVF_REQUIRE(false, "gemm_f32x3: bad epilogue %d", epilogue);
  }
  return check_launch("gemm_f32x3_kernel");
}

}  // namespace

size_t gemm_f32_ws(int64_t M, int64_t N, int64_t K) {
  int sp = f32_splits(M, N, K, AVF_EPI_NONE);
  if (sp == 1) sp = f32_split64(M, N, K, AVF_EPI_NONE);
  const int sx = (M >= 96 && N >= 96) ? f32x3_splits(M, N, K, AVF_EPI_NONE) : 1;  // the bf16x3 form of the same call
  if (sx > sp) sp = sx;
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}

void set_f32_arith(int mode) { g_f32_arith = mode ? 1 : 0; }
int get_f32_arith() { return g_f32_arith; }

// ---- the grouped weight-gradient launch of the parity mode (bf16x3 arithmetic) ---------------------------------------------------
namespace {
int f32x3_group_setup(const TnGroupArgs& a, F32GroupParams* g, bool any_arith = false) {
  if ((g_f32_arith != 1 && !any_arith) || a.count < 2 || a.count > 4 || a.K < 512 || a.K >= (1LL << 31)) return 0;
  int tiles = 0;
  int64_t el = 0;
  for (int i = 0; i < a.count; ++i) {
    if (a.M[i] < 96 || a.N[i] < 96 || a.N[i] % 4 != 0 || a.M[i] >= (1LL << 31) || a.N[i] >= (1LL << 31)) return 0;
    if (a.A[i] && (((uintptr_t)a.A[i] | (uintptr_t)a.B[i] | (uintptr_t)a.C[i]) & 15)) return 0;
    F32GemmParams& p = g->p[i];
    memset(&p, 0, sizeof(p));
    p.A = (const float*)a.A[i]; p.a_sm = 1; p.a_sk = a.lda[i];   // A_i [K, M_i]: element (m, k) at A[k lda + m]
    p.B = (const float*)a.B[i]; p.b_sk = a.ldb[i]; p.b_sn = 1;   // B_i [K, N_i]
    p.C = a.C[i]; p.ldc = a.N[i];
    p.M = (int)a.M[i]; p.N = (int)a.N[i]; p.K = (int)a.K;
    p.drop = kNoDrop;
    g->tile0[i] = tiles;
    g->el0[i] = el;
    tiles += (int)(ceil_div(a.M[i], SBM) * ceil_div(a.N[i], SBN));
    el += a.M[i] * a.N[i];
  }
  g->tile0[a.count] = tiles;
  g->el0[a.count] = el;
  g->count = a.count;
  g->vec = 1;
  int64_t sp = 512 / tiles;   // tiles x splits within one round of the 512 workgroup slots, at least 8 K-steps per split
  if (sp > a.K / 256) sp = a.K / 256;
  if (sp > 32) sp = 32;
  if (sp < 2) return 0;       // enough tiles to fill the chip unsplit: the per-problem launches serve that case
  const int kchunk = (int)(ceil_div(ceil_div(a.K, sp), SBK) * SBK);
  g->S = (int)ceil_div(a.K, kchunk);
  for (int i = 0; i < a.count; ++i) g->p[i].kchunk = kchunk;
  return 1;
}
}  // namespace

bool gemm_f32x3_tn_group_ok(const TnGroupArgs& a) {
  F32GroupParams g;
  return f32x3_group_setup(a, &g) != 0;
}
size_t gemm_f32x3_tn_group_ws(const TnGroupArgs& a) {  // (whatever arithmetic is selected NOW: a buffer sized under one serves both)
  F32GroupParams g;
  if (!f32x3_group_setup(a, &g, true)) return 0;
  return (size_t)g.S * (size_t)g.el0[g.count] * sizeof(float);
}
int gemm_f32x3_tn_group(const TnGroupArgs& a, hipStream_t s) {
  F32GroupParams g;
  AVF_REQUIRE(f32x3_group_setup(a, &g), "gemm_f32x3_tn_group: shapes not eligible (ask gemm_f32x3_tn_group_ok first)");
  AVF_REQUIRE(a.workspace && ((uintptr_t)a.workspace & 15) == 0, "gemm_f32x3_tn_group: workspace missing");
  double fl = 0, by = 0;
  for (int i = 0; i < a.count; ++i) {
    AVF_REQUIRE(a.A[i] && a.B[i] && a.C[i], "gemm_f32x3_tn_group: null operand %d", i);
    g.p[i].slabs = (float*)a.workspace + (size_t)g.S * (size_t)g.el0[i];
    fl += 2.0 * a.M[i] * a.N[i] * a.K;
    by += 4.0 * (a.K * (a.M[i] + a.N[i]) + a.M[i] * a.N[i]);
  }
  TimingScope ts(KC_GEMM_F32, fl, by, s);
  gemm_f32x3_group_kernel<<<(unsigned)(g.tile0[g.count] * g.S), 256, 0, s>>>(g);
  AVF_TRY(check_launch("gemm_f32x3_group_kernel"));
  gemm_f32x3_group_fold_kernel<<<(unsigned)ceil_div(g.el0[g.count] / 4, 256), 256, 0, s>>>(g);
  return check_launch("gemm_f32x3_group_fold_kernel");
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
  if (g_f32_arith == 1 && f32x3_ok(p)) {  // bf16x3 on the bf16 matrix pipe (round 6)
    AVF_REQUIRE(a.epilogue != AVF_EPI_BIAS_RES || a.residual, "gemm_f32: residual missing");
    AVF_REQUIRE((a.epilogue != AVF_EPI_BIAS_GELU && a.epilogue != AVF_EPI_DGELU) || a.aux, "gemm_f32: aux missing");
    const int S0 = a.workspace ? f32x3_splits(a.M, a.N, a.K, a.epilogue) : 1;
    int S = 1;
    if (S0 > 1) {
      p.slabs = (float*)a.workspace;
      p.kchunk = (int)(ceil_div(ceil_div(a.K, S0), SBK) * SBK);
      S = (int)ceil_div(a.K, p.kchunk);
    }
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    int vec = (p.N % 4 == 0);
    if (S > 1) vec = vec && al16(p.slabs);
    else
      vec = vec && p.ldc % 4 == 0 && al16(p.C) && (!p.bias || al16(p.bias)) &&
            (a.epilogue != AVF_EPI_BIAS_RES || (p.ldres % 4 == 0 && al16(p.residual))) &&
            ((a.epilogue != AVF_EPI_BIAS_GELU && a.epilogue != AVF_EPI_DGELU) || (p.ldaux % 4 == 0 && al16(p.aux)));
    AVF_REQUIRE(ceil_div(p.M, SBM) < 65536, "gemm_f32: M too large for grid");
    const bool akf = p.a_sk == 1, bkf = p.b_sk == 1;
    int rc;
    if (akf && bkf) rc = launch_f32x3<true, true>(p, a.epilogue, S, vec, s);
    else if (akf) rc = launch_f32x3<true, false>(p, a.epilogue, S, vec, s);
    else if (bkf) rc = launch_f32x3<false, true>(p, a.epilogue, S, vec, s);
    else rc = launch_f32x3<false, false>(p, a.epilogue, S, vec, s);
    AVF_TRY(rc);
    if (S > 1) {
      gemm_f32_fold_kernel<<<(unsigned)ceil_div(a.M * a.N, 256), 256, 0, s>>>(p, S, a.epilogue == AVF_EPI_BIAS_RES ? 1 : 0);
      return check_launch("gemm_f32_fold_kernel");
    }
    return 0;
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
