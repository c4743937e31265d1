// attn_bwd_merged.hip - throughput-mode attention backward as ONE kernel per (clip, head): dQ, dK and dV from a
// single recomputation of the probabilities, on v_mfma_f32_32x32x16_bf16 (dQ: v_mfma_f32_16x16x32_bf16).
//
// Reference: autograd of models/heads.py:222-237 (dots = q k^T * dh^-0.5, softmax over keys, out = attn v).
//
// The two head-resident kernels of attn_bf16.hip (dQ with the query on the lane, dK/dV with the key on the lane) each
// recompute S = q k^T and dP = dO v^T: 14 matrix products for 10 nominal.  Here every (query, key) tile is visited once:
//
//   one workgroup = 4 wavefronts (one per SIMD, the whole 512-register file each) = one (clip, head), N <= 512, dim_head 64;
//   wave w OWNS key blocks [w KB, (w+1) KB) of 32 keys: dK^T and dV^T of its keys stay in 64 KB accumulators for the whole
//   kernel (AGPRs), V fragments of its keys stay in registers, the head's K image is resident in LDS; the workgroup
//   sweeps the queries in SLICES of 32 rows (Q and dO slices arrive by LDS-DMA, one slice ahead, one barrier per slice).
//   Per (slice, key block), key on the lane:
//     A:  S  = Q K^T - lse2, dP = dO V^T - delta     (the row statistics ride in the C operand; q carries log2(e)/sqrt(dh))
//     B:  P = exp2(S), dS = P * dP, packed to bf16   (48 VALU instructions per lane)
//     C:  dV^T += dO^T P,  dK^T += Q^T dS            (the accumulator tiles ARE the B operands)
//         dS^T goes to a [key][query] image in LDS (8-byte stores of the accumulator rows).
//   dQ of slice s is computed DURING slice s+1 from the dS^T images of all the key blocks (written by all four waves,
//   visible after the slice barrier): wave w owns columns [16 w, 16 w + 16) of dQ and runs the reduction over the keys as
//   16x16x32 MFMAs (dQ^T = K^T dS^T, both operands by transposed LDS reads), interleaved with the VALU work of the new
//   slice's first key block - no cross-wave reduction, no atomics, the rows leave straight from the accumulators.
//   (A first version summed four per-wave dQ partials: ds_add_f32 into one tile measured 600 cycles per instruction - 55 %
//   of the kernel -, four fp32 slabs plus a summing pass still 15 %.)
//   delta = rowsum(dO * O) of the NEXT slice is computed at the bottom of each slice from the dO rows already in LDS
//   and an O chunk fetched at the top.
// One wave per SIMD: nothing but the wave's own instruction order overlaps VALU and matrix pipe, so the slice body is
// software-pipelined by hand (see the stages below) and fenced with sched_barrier.  Bitwise deterministic.
//
// Rows past N: K rows are zero in the image and V fragments zero (so dS K and the discarded dK/dV columns are finite and
// contribute nothing), padded keys are masked to P = 0 (RAGGED), query rows are copies of row N-1 with lse = +inf
// (P = 0) and delta = 0.
#include <type_traits>

#include "common.hpp"

namespace avf {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) short m_s16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t m_u32x4_t;
typedef __attribute__((address_space(3))) char m_lds_char;

// Phase stamps for tools/diag/attn_m4_phases.hip (that file defines these and includes this one); nothing in the product.
#ifndef AVF_PHASE_MARK
#define AVF_PHASE_INIT()
#define AVF_PHASE_MARK(slot)
#define AVF_PHASE_FLUSH()
#endif

#ifndef AVF_M4_MAXKB
#define AVF_M4_MAXKB 4
#endif

constexpr float kLog2E = 1.4426950408889634f;

#define AVF_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define AVF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define AVF_FENCE() __builtin_amdgcn_sched_barrier(0)

// The kernel is built with -mllvm -amdgpu-mfma-vgpr-form (see _build.py): the builtin MFMAs keep C / D in the
// architectural VGPRs, so S, dP and dQ feed the VALU without v_accvgpr_read copies.  The dK^T / dV^T accumulators - 64
// registers per key block, up to all 256 accumulation registers - are pinned to the AGPR half by the "+a" constraint of
// this statement (hipcc picks one MFMA form per function and cannot mix them itself).  hipcc pads nothing inside an asm
// string: the callers keep a VALU write of an operand at least two instructions away (m_operand_settle).
__device__ __forceinline__ void m_mfma_acc(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void m_operand_settle() { asm volatile("s_nop 1"); }

// 16-byte chunk permutation of a 128-byte row (8 chunks): conflict-free for the ds_read_b128 row fragments of the
// 32x32x16 operands (rows of equal parity inside one 16-lane service group get 8 distinct values) and for the
// ds_read_b64_tr_b16 fragments (rows r, r+2 of a 4-row block land in different 64-byte halves).
__device__ __forceinline__ int m_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
// 8-byte slot permutation of a 64-byte row of a dS^T image (8 slots): a permutation of the bits of (key>>1)&7, so the
// 8-byte stores of 16 consecutive keys hit distinct banks, with (key>>2)&1 on slot bit 2, so the two 4-row blocks a
// 16x16x32 transposed read takes from rows k .. k+3 and k+4 .. k+7 use different halves of their rows.
__device__ __forceinline__ int m_txor(int key) { return (((key >> 2) & 1) << 2) | (((key >> 1) & 1) << 1) | ((key >> 3) & 1); }

__device__ __forceinline__ void m_glds16(const void* g, char* l) {  // LDS-DMA, opaque to hipcc's wait-count pass
  const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(m_lds_char*)l);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(la), "v"(g)
               : "memory");
}

__device__ __forceinline__ bf16x8_t m_row_frag(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

__device__ __forceinline__ bf16x8_t m_tr_frag(const char* p0, const char* p1) {
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(const m_lds_char*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(const m_lds_char*)p1);
  m_s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

__device__ __forceinline__ float m_dot8(const uint4& x, const uint4& y) {
  const uint32_t a[4] = {x.x, x.y, x.z, x.w}, b[4] = {y.x, y.y, y.z, y.w};
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = fmaf(__uint_as_float(a[i] << 16), __uint_as_float(b[i] << 16), acc);
    acc = fmaf(__uint_as_float(a[i] & 0xffff0000u), __uint_as_float(b[i] & 0xffff0000u), acc);
  }
  return acc;
}

// sum over the 8 lanes 8 k .. 8 k + 7, valid in lane 8 k: two quad steps and row_shl:4 on the VALU (three dependent
// ds_bpermute round trips sat at the bottom of every slice, in front of its barrier)
__device__ __forceinline__ float m_sum8(float v) {
  v += AVF_DPP_F32(v, 0xB1);
  v += AVF_DPP_F32(v, 0x4E);
  v += AVF_DPP_F32(v, 0x104);
  return v;
}

__device__ __forceinline__ f32x16_t m_zero16() {
  f32x16_t z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// LDS map (bytes): K image [4 KB x 32 rows][128 B] | ring (2 slots x (Q 4 KB | dO 4 KB)) | NL | ND | T (2 buffers x 4 KB
// key blocks x 2 KB: the dS^T images of a slice, [32 keys][32 queries] bf16 each)
struct MLayout {
  int kimg, ring, nl, nd, t, total;
  __host__ __device__ MLayout(int KB, int NS) {
    kimg = 0;
    ring = kimg + 4 * KB * 32 * 128;
    nl = ring + 2 * 8192;
    nd = nl + NS * 32 * 4;
    t = nd + NS * 32 * 4;
    total = t + 2 * 4 * KB * 2048;
  }
};

// RAGGED: N < 4 KB 32 - key blocks that hold rows past N mask them (P = 0: -lse2 alone can be a large positive
// exponent when every score of a row is very negative)
// MASKED: the token mask of heads.py:225-232 (keep[b][n] != 0 keeps token n), as attn_fwd_res_kernel<.., MASKED> applied it:
// a kept query gives dropped keys P = 0; a dropped query attended uniformly (P = 1/N on every real key: its lse2 is
// log2 N exactly) and no gradient flows through its scores - dS = 0, so only dV sees the row.
// MXO: also write the MX-FP8 image of dqkv (e4m3 bytes [B N][3 I] + one E8M0 byte per 32 columns) - the A operand of the
// dqkv -> dh1 GEMM in the fp8 mode.  A 32-block of dK / dV is the 16 registers of a lane and of lane + 32; a 32-block of dQ
// stays inside one wave because the dQ job of this form gives a wave one query tile and 32 columns (not two tiles and 16).  The image is that
// of the bf16 values as STORED.
template <int KB, bool RAGGED, bool MASKED = false, bool MXO = false>
__global__ __launch_bounds__(256) void attn_bwd_m4_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                          const bf16* __restrict__ d_o, const float* __restrict__ lse2,
                                                          bf16* __restrict__ dqkv, int N, int H,
                                                          const uint8_t* __restrict__ keep = nullptr,
                                                          uint8_t* __restrict__ dq_q = nullptr,
                                                          uint8_t* __restrict__ dq_s = nullptr) {
  constexpr int DH = 64;
  constexpr int NKB = 4 * KB;  // key blocks of the head (padded ones included)
#ifdef AVF_MXO_NO_DQ
  constexpr bool MXQ = false;
#else
  constexpr bool MXQ = MXO;
#endif
#ifdef AVF_MXO_NO_DKV
  constexpr bool MXK = false;
#else
  constexpr bool MXK = MXO;
#endif
  extern __shared__ __attribute__((aligned(16))) char m_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const bf16* gbase = d_o + (int64_t)b * N * I + h * DH;
  const bf16* obase = o + (int64_t)b * N * I + h * DH;
  const int NS = (N + 31) >> 5;  // query slices
  const MLayout L(KB, NS);
  char* kimg = m_smem + L.kimg;
  char* ring = m_smem + L.ring;
  float* NLs = reinterpret_cast<float*>(m_smem + L.nl);
  float* NDs = reinterpret_cast<float*>(m_smem + L.nd);
  char* tbuf = m_smem + L.t;
  const int kb0 = wave * KB;  // first key block of this wave

  // ---- per-lane LDS byte offsets -------------------------------------------------------------------------------
  // row fragment of a 32-row block (128-byte rows): row r, logical chunk 2 ks + hf
  int off_row[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) off_row[ks] = r * 128 + (((2 * ks + hf) ^ m_swz(r)) << 4);
  // transposed fragments: 16-lane group G = lane>>4, lane 4 qq + p of the group supplies row qq, columns 4 p .. 4 p + 3
  const int li = lane & 15, G = lane >> 4, gg = G & 1, qq = li >> 2, p = li & 3;
  // (a) 32x32x16 A operand in accumulator k-order (element j of k-step s2 <-> row 16 s2 + 8 (j>>2) + 4 hf + (j&3)) of a
  //     32-row slice: rows 4 hf + qq and + 8, columns 32 db + 16 gg + 4 p
  int off_tra[2][2];
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int ch = 4 * db + 2 * gg + (p >> 1), byte = (p & 1) * 8;
    const int ra = 4 * hf + qq;
    off_tra[db][0] = ra * 128 + ((ch ^ m_swz(ra)) << 4) + byte;
    off_tra[db][1] = (ra + 8) * 128 + ((ch ^ m_swz(ra + 8)) << 4) + byte;
  }
  // (b) 16x16x32 operands of the dQ job, k-order of a 32-key block: element j of lane group G <-> key 4 G + (j&3) + 16 (j>>2):
  //     rows 4 G + qq and + 16.  A = K^T: columns d = 16 w + 4 p of the K image.  B = dS^T: query columns 16 qt + 4 p of a
  //     T image (64-byte rows; 8-byte slot 4 qt + p).
  //     MXQ (the fp8 image of dQ is written too): a wave owns ONE query tile (w & 1) and a whole 32-column block (w >> 1) -
  //     two A fragments (columns 32 (w >> 1) + 16 ct), one B fragment - so that a block's maximum stays inside the wave
  int off_kq[2], off_kq2[2], off_tq[2][2];
  {
    const int rk = 4 * G + qq;
    const int ch = (MXQ ? 4 * (wave >> 1) : 2 * wave) + (p >> 1), byte = (p & 1) * 8;
    off_kq[0] = rk * 128 + ((ch ^ m_swz(rk)) << 4) + byte;
    off_kq[1] = (rk + 16) * 128 + ((ch ^ m_swz(rk + 16)) << 4) + byte;
    off_kq2[0] = rk * 128 + (((ch + 2) ^ m_swz(rk)) << 4) + byte;
    off_kq2[1] = (rk + 16) * 128 + (((ch + 2) ^ m_swz(rk + 16)) << 4) + byte;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      off_tq[qt][0] = rk * 64 + (((4 * qt + p) ^ m_txor(rk)) << 3);
      off_tq[qt][1] = (rk + 16) * 64 + (((4 * qt + p) ^ m_txor(rk + 16)) << 3);
    }
  }
  // store of accumulator registers 4 g .. 4 g + 3 (queries 8 g + 4 hf ..) of key r into a T image: slot 2 g + hf
  int off_tw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) off_tw[g] = r * 64 + (((2 * g + hf) ^ m_txor(r)) << 3);

  // ---- prologue --------------------------------------------------------------------------------------------------
  AVF_PHASE_INIT();
  const int lrow = lane >> 3, lslot = lane & 7;
  // K image: pieces of 8 rows; rows past N are fetched from row N-1 and zeroed below
  for (int pc = wave; pc < 16 * KB; pc += 4) {
    const int row = pc * 8 + lrow;
    const int src = row < N ? row : N - 1;
    m_glds16(kbase + (int64_t)src * ld + ((lslot ^ m_swz(row)) << 3), kimg + pc * 1024);
  }
  auto issue_slice = [&](int s) {  // wave w brings rows 8 w .. 8 w + 7 of the slice's Q and dO
    char* slot = ring + (s & 1) * 8192;
    const int lr = 8 * wave + lrow;
    int row = 32 * s + lr;
    row = row < N ? row : N - 1;
    const int ch = (lslot ^ m_swz(lr)) << 3;
    m_glds16(qbase + (int64_t)row * ld + ch, slot + wave * 1024);
    m_glds16(gbase + (int64_t)row * I + ch, slot + 4096 + wave * 1024);
  };
  issue_slice(0);
  // V fragments of the wave's keys (B operand of dP = dO V^T: lane (r, hf) holds V[key r][16 ks + 8 hf ..])
  bf16x8_t vfr[KB][4];
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int key = 32 * (kb0 + j) + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (key < N) v = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * ld + 16 * ks + 8 * hf);
      vfr[j][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  // statistics: NL = -lse2 (past N: -3e38, finite so that it splits into bf16 pieces; P = exp2(s - 3e38) = 0)
  for (int q = tid; q < NS * 32; q += 256) NLs[q] = q < N ? -lse2[(int64_t)bh * N + q] : -3.0e38f;
  // delta of slice 0 straight from global memory: thread t = row t>>3, chunk t&7
  const int drow = tid >> 3, dch = tid & 7;
  {
    float part = 0.f;
    if (drow < N) {
      const uint4 ov = *reinterpret_cast<const uint4*>(obase + (int64_t)drow * I + dch * 8);
      const uint4 gv = *reinterpret_cast<const uint4*>(gbase + (int64_t)drow * I + dch * 8);
      part = m_dot8(ov, gv);
    }
    part = m_sum8(part);
    if (dch == 0) NDs[drow] = -part;
  }
  // MASKED: is this lane's key of block j a kept token (padded keys: no); the query flags of a slice as a bit mask,
  // fetched one slice ahead (bit i of qm: query 32 s + i, already shifted by this lane's 4 hf)
  bool kk[KB];
  uint32_t qf = 1u, qm = 0xffffffffu;
  const float dropv = -log2f((float)N);
#pragma unroll
  for (int j = 0; j < KB; ++j) kk[j] = true;
  if constexpr (MASKED) {
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int key = 32 * (kb0 + j) + r;
      kk[j] = key < N ? keep[(int64_t)b * N + key] != 0 : false;
    }
    qf = r < N ? keep[(int64_t)b * N + r] : 1u;
  }
  // the builtin form is VISIBLE to hipcc's wait-count pass (it retires the V fragment loads in its books: with the asm form
  // alone it put a vmcnt(0) in front of their first use in EVERY slice, exposing the latency of that slice's own DMA)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt unconstrained
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (N < NKB * 32) {  // zero the K rows past N (wave-uniform condition)
    for (int idx = N * 8 + tid; idx < NKB * 32 * 8; idx += 256)
      *reinterpret_cast<uint4*>(kimg + idx * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }

  AVF_PHASE_MARK(0);
  f32x16_t dkacc[KB][2], dvacc[KB][2];
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      dkacc[j][db] = m_zero16();
      dvacc[j][db] = m_zero16();
    }
  const float qscale = 0.125f;  // dh^-0.5

  // ---- the pieces of one (slice, key block) --------------------------------------------------------------------
  const char *qsl = nullptr, *gsl = nullptr;
  char* tcur = nullptr;  // the T images this slice writes (this wave's blocks)
  int kopq = 0;
  bf16x8_t qa[4], ga[4], kf[4];
  f32x16_t sacc, pacc;
  m_u32x4_t pk[2], dk[2];  // bf16 pairs of P and dS of the block whose stage C is pending, one k-step (8 registers) each

  // The row statistics enter S and dP through the matrix pipe: a fifth k-step whose A operand holds, for query row r, the
  // three bf16 pieces hi + mid + lo = -lse2[r] (exactly: 3 x 8 significand bits) in its first three k slots and whose B
  // operand is 1 there - S = Q K^T + (-lse2) 1^T.  Loading them into the C operand instead (the first version) cost 8
  // broadcast ds_read_b128 per key block and made the first MFMA of every block wait for LDS; this costs two MFMAs.
  bf16x8_t nlA, ndA;
  const bf16x8_t onesB = __builtin_bit_cast(bf16x8_t, m_u32x4_t{hf ? 0u : 0x3F803F80u, hf ? 0u : 0x00003F80u, 0u, 0u});
  auto split3 = [&](float x) {
    const float hi = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - hi;
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float lo = r1 - mid;
    const uint32_t w0 = (__float_as_uint(hi) >> 16) | (__float_as_uint(mid) & 0xffff0000u);
    const uint32_t w1 = pack_bf16x2(lo, 0.f);
    return __builtin_bit_cast(bf16x8_t, m_u32x4_t{hf ? 0u : w0, hf ? 0u : w1, 0u, 0u});
  };
  auto load_kf = [&](int j) {
    const char* kblk = kimg + (kb0 + j) * 4096 + kopq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = m_row_frag(kblk + off_row[ks]);
  };
  // stage A(j): ten MFMAs, no operand fresher than a whole phase
  auto stage_a = [&](auto jtag) {
    constexpr int j = decltype(jtag)::value;
    sacc = AVF_MFMA32(qa[0], kf[0], m_zero16());
    pacc = AVF_MFMA32(ga[0], vfr[j][0], m_zero16());
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
      sacc = AVF_MFMA32(qa[ks], kf[ks], sacc);
      pacc = AVF_MFMA32(ga[ks], vfr[j][ks], pacc);
    }
    sacc = AVF_MFMA32(nlA, onesB, sacc);
    pacc = AVF_MFMA32(ndA, onesB, pacc);
    if constexpr (RAGGED && !MASKED) {
      if (32 * (kb0 + j) + 32 > N) {  // wave-uniform: the block holds padded keys
        const bool dead = 32 * (kb0 + j) + r >= N;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = dead ? -INFINITY : sacc[i];
      }
    }
    if constexpr (MASKED) {  // element i of this lane = query 8 (i >> 2) + 4 hf + (i & 3) of the slice, key r of block j
      const bool real = 32 * (kb0 + j) + r < N;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool qkept = ((qm >> (8 * (i >> 2) + (i & 3))) & 1u) != 0;
        sacc[i] = qkept ? (kk[j] ? sacc[i] : -INFINITY) : (real ? dropv : -INFINITY);
      }
    }
  };
  // stage B in eight chunks of six VALU instructions: chunk c turns elements 2 c, 2 c + 1 into one dword of P and of dS
  auto b_chunk = [&](int c, m_u32x4_t (&pkn)[2], m_u32x4_t (&dkn)[2]) {
    const float p0 = __builtin_amdgcn_exp2f(sacc[2 * c]), p1 = __builtin_amdgcn_exp2f(sacc[2 * c + 1]);
    float d0 = p0 * pacc[2 * c], d1 = p1 * pacc[2 * c + 1];
    if constexpr (MASKED) {  // no gradient through the (filled) scores of a dropped query
      d0 = ((qm >> (8 * ((2 * c) >> 2) + ((2 * c) & 3))) & 1u) ? d0 : 0.f;
      d1 = ((qm >> (8 * ((2 * c + 1) >> 2) + ((2 * c + 1) & 3))) & 1u) ? d1 : 0.f;
    }
    pkn[c >> 2][c & 3] = pack_bf16x2(p0, p1);
    dkn[c >> 2][c & 3] = pack_bf16x2(d0, d1);
  };
  auto write_t = [&](auto jtag, const m_u32x4_t (&d)[2]) {  // dS^T of block j: registers 4 g .. 4 g + 3 are dwords 2 g, 2 g + 1
    constexpr int j = decltype(jtag)::value;
    char* timg = tcur + j * 2048;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      *reinterpret_cast<uint2*>(timg + off_tw[2 * s2]) = make_uint2(d[s2][0], d[s2][1]);
      *reinterpret_cast<uint2*>(timg + off_tw[2 * s2 + 1]) = make_uint2(d[s2][2], d[s2][3]);
    }
  };
  auto tr_qg = [&](int db, int s2, bf16x8_t& gt, bf16x8_t& qt) {
    gt = m_tr_frag(gsl + s2 * 2048 + off_tra[db][0], gsl + s2 * 2048 + off_tra[db][1]);
    qt = m_tr_frag(qsl + s2 * 2048 + off_tra[db][0], qsl + s2 * 2048 + off_tra[db][1]);
  };
  // stage C(j) - eight accumulating MFMAs - with stage B(j+1) riding between them (NX) and the operands of A(j+2)
  // preloaded into the registers B(j+1) has finished with (NX2).  The LDS reads run two slots ahead of their MFMAs.
  bf16x8_t cg0, cq0, cg1, cq1;  // the first two fragment pairs of the pending stage C
  auto prefetch_c = [&]() { tr_qg(0, 0, cg0, cq0); tr_qg(0, 1, cg1, cq1); };
  auto phase_c = [&](auto jtag) {
    constexpr int j = decltype(jtag)::value;
    constexpr bool NX = j + 1 < KB, NX2 = j + 2 < KB;
    bf16x8_t g0, q0, g1, q1;
    m_u32x4_t pkn[2], dkn[2];
    const bf16x8_t pk0 = __builtin_bit_cast(bf16x8_t, pk[0]), pk1 = __builtin_bit_cast(bf16x8_t, pk[1]);
    const bf16x8_t dk0 = __builtin_bit_cast(bf16x8_t, dk[0]), dk1 = __builtin_bit_cast(bf16x8_t, dk[1]);
    g0 = cg0; q0 = cq0; g1 = cg1; q1 = cq1;  // (0,0) and (0,1): requested before stage A(j+1)
    m_operand_settle();
    AVF_FENCE();
    if constexpr (NX2) load_kf(j + 2);  // K fragments of the block after next: a whole phase ahead of their MFMAs
    m_mfma_acc(dvacc[j][0], g0, pk0);
    if constexpr (NX) b_chunk(0, pkn, dkn);
    AVF_FENCE();
    m_mfma_acc(dkacc[j][0], q0, dk0);
    if constexpr (NX) b_chunk(1, pkn, dkn);
    tr_qg(1, 0, g0, q0);                // four slots ahead of its MFMAs
    AVF_FENCE();
    m_mfma_acc(dvacc[j][0], g1, pk1);
    if constexpr (NX) b_chunk(2, pkn, dkn);
    AVF_FENCE();
    m_mfma_acc(dkacc[j][0], q1, dk1);
    if constexpr (NX) b_chunk(3, pkn, dkn);
    tr_qg(1, 1, g1, q1);
    AVF_FENCE();
    m_mfma_acc(dvacc[j][1], g0, pk0);
    if constexpr (NX) b_chunk(4, pkn, dkn);
    AVF_FENCE();
    m_mfma_acc(dkacc[j][1], q0, dk0);
    if constexpr (NX) b_chunk(5, pkn, dkn);
    AVF_FENCE();
    m_mfma_acc(dvacc[j][1], g1, pk1);
    if constexpr (NX) b_chunk(6, pkn, dkn);
    AVF_FENCE();
    m_mfma_acc(dkacc[j][1], q1, dk1);
    if constexpr (NX) b_chunk(7, pkn, dkn);
    AVF_FENCE();
    if constexpr (NX) {
      write_t(std::integral_constant<int, j + 1>{}, dkn);
      pk[0] = pkn[0]; pk[1] = pkn[1]; dk[0] = dkn[0]; dk[1] = dkn[1];
    }
    AVF_FENCE();
  };
  // dQ of slice sp (the dS^T images of its key blocks in T buffer sp & 1): this wave's 16 columns, two 16x16x32 MFMAs per
  // key block; WITHB: the VALU work of the new slice's first key block (stage B(0)) rides between them
  auto dq_job = [&](int sp, auto withb_tag) {
    constexpr bool WITHB = decltype(withb_tag)::value;
    const char* tb = tbuf + (sp & 1) * (NKB * 2048);
    const char* kb = kimg + kopq;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    m_u32x4_t pkn[2], dkn[2];
    const int tq0 = MXQ ? (wave & 1) : 0;  // (MXQ: t1 carries the second A fragment, not a second query tile)
    bf16x8_t ka = m_tr_frag(kb + off_kq[0], kb + off_kq[1]);
    bf16x8_t t0 = m_tr_frag(tb + off_tq[tq0][0], tb + off_tq[tq0][1]);
    bf16x8_t t1 = MXQ ? m_tr_frag(kb + off_kq2[0], kb + off_kq2[1]) : m_tr_frag(tb + off_tq[1][0], tb + off_tq[1][1]);
#pragma unroll
    for (int kbk = 0; kbk < NKB; ++kbk) {
      AVF_FENCE();
      bf16x8_t kan = ka, t0n = t0, t1n = t1;
      if (kbk + 1 < NKB) {
        const char* kn = kb + (kbk + 1) * 4096;
        const char* tn = tb + (kbk + 1) * 2048;
        kan = m_tr_frag(kn + off_kq[0], kn + off_kq[1]);
        t0n = m_tr_frag(tn + off_tq[tq0][0], tn + off_tq[tq0][1]);
        t1n = MXQ ? m_tr_frag(kn + off_kq2[0], kn + off_kq2[1]) : m_tr_frag(tn + off_tq[1][0], tn + off_tq[1][1]);
      }
      acc0 = AVF_MFMA16(ka, t0, acc0);
      acc1 = MXQ ? AVF_MFMA16(t1, t0, acc1) : AVF_MFMA16(ka, t1, acc1);
      if constexpr (WITHB) {  // chunk c rides behind the MFMAs of key block floor(c NKB / 8)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c * NKB / 8 == kbk) b_chunk(c, pkn, dkn);
      }
      ka = kan; t0 = t0n; t1 = t1n;
    }
    AVF_FENCE();
    if constexpr (WITHB) {
      pk[0] = pkn[0]; pk[1] = pkn[1]; dk[0] = dkn[0]; dk[1] = dkn[1];
    }
    // rows leave from the accumulators: lane (query li of tile qt, group G) holds columns 16 w + 4 G .. + 3
    if constexpr (MXQ) {  // ... of query tile w & 1: columns 32 (w >> 1) + 16 ct + 4 G .. + 3, ct = 0 (acc0), 1 (acc1)
      const int q = 32 * sp + 16 * (wave & 1) + li;
      const int colb = h * DH + 32 * (wave >> 1);
      uint32_t sw[2][2];
      float t = 0.f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4_t a = ct ? acc1 : acc0;
        sw[ct][0] = pack_bf16x2(a[0] * qscale, a[1] * qscale);
        sw[ct][1] = pack_bf16x2(a[2] * qscale, a[3] * qscale);
        t = fmaxf(t, fmaxf(fmaxf(fabsf(__uint_as_float(sw[ct][0] << 16)), fabsf(__uint_as_float(sw[ct][0] & 0xffff0000u))),
                           fmaxf(fabsf(__uint_as_float(sw[ct][1] << 16)), fabsf(__uint_as_float(sw[ct][1] & 0xffff0000u)))));
      }
      {  // the row's 32 columns: lanes li, li + 16, li + 32, li + 48
        const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
        t = fmaxf(__uint_as_float(x[0]), __uint_as_float(x[1]));
        const auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
        t = fmaxf(__uint_as_float(y[0]), __uint_as_float(y[1]));
      }
      float inv;
      const uint32_t sb = mx8_scale_byte(t, &inv);
      if (q < N) {
        const int64_t row = (int64_t)b * N + q;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          *reinterpret_cast<uint2*>(dqkv + row * ld + colb + 16 * ct + 4 * G) = make_uint2(sw[ct][0], sw[ct][1]);
          const float v4[4] = {__uint_as_float(sw[ct][0] << 16), __uint_as_float(sw[ct][0] & 0xffff0000u),
                               __uint_as_float(sw[ct][1] << 16), __uint_as_float(sw[ct][1] & 0xffff0000u)};
          *reinterpret_cast<uint32_t*>(dq_q + row * ld + colb + 16 * ct + 4 * G) = mx8_pack4(v4, inv);
        }
        if (G == 0) dq_s[row * (ld >> 5) + (colb >> 5)] = (uint8_t)sb;
      }
    } else {
      // lane groups G, G + 1 trade words (v_permlane16_swap): an even group then holds 8 consecutive columns of query tile 0,
      // an odd one 8 of tile 1 - one 16-byte store per lane and slice instead of two 8-byte ones
      const uint32_t a0 = pack_bf16x2(acc0[0] * qscale, acc0[1] * qscale), a1 = pack_bf16x2(acc0[2] * qscale, acc0[3] * qscale);
      const uint32_t b0 = pack_bf16x2(acc1[0] * qscale, acc1[1] * qscale), b1 = pack_bf16x2(acc1[2] * qscale, acc1[3] * qscale);
      const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      const int q = 32 * sp + 16 * (G & 1) + li;
      if (q < N)
        *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + q) * ld + h * DH + 16 * wave + 4 * (G & ~1)) =
            make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
  };

  // ---- the sweep over the query slices ---------------------------------------------------------------------------
  for (int s = 0; s < NS; ++s) {
    qsl = ring + (s & 1) * 8192;
    gsl = qsl + 4096;
    tcur = tbuf + (s & 1) * (NKB * 2048) + kb0 * 2048;
    uint4 onext = make_uint4(0, 0, 0, 0);
    const bool more = s + 1 < NS;
    if constexpr (MASKED) {  // lanes r and r + 32 carry the same flag: the low word of the ballot is the slice's mask
      const uint32_t all = (uint32_t)__builtin_amdgcn_ballot_w64(qf != 0);
      qm = all >> (4 * hf);
    }
    if (more) {
      issue_slice(s + 1);
      const int q = 32 * (s + 1) + drow;
      if (q < N) onext = *reinterpret_cast<const uint4*>(obase + (int64_t)q * I + dch * 8);
      if constexpr (MASKED) {
        const int qn = 32 * (s + 1) + r;
        qf = qn < N ? keep[(int64_t)b * N + qn] : 1u;
      }
    }
    AVF_PHASE_MARK(1);

    nlA = split3(NLs[32 * s + r]);
    ndA = split3(NDs[32 * s + r]);
    // the K image never changes, so the compiler would hoist every fragment read of it out of the slice loop (48
    // registers per key block); an opaque per-slice copy of the image offset keeps the reads inside
    kopq = 0;
    asm volatile("" : "+v"(kopq));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qa[ks] = m_row_frag(qsl + off_row[ks]);
      ga[ks] = m_row_frag(gsl + off_row[ks]);
    }
    load_kf(0);
    // every wave runs all its KB blocks, padded ones included (zero K rows and V, masked scores): the slice barrier
    // would make a wave that skipped them wait for the others anyway, and one straight-line body keeps the 64 KB
    // accumulators in place (two code paths made the compiler shuffle all of them at the join)
    stage_a(std::integral_constant<int, 0>{});
    AVF_FENCE();
    if constexpr (KB > 1) load_kf(1);
    AVF_PHASE_MARK(2);
    if (s > 0) {
      dq_job(s - 1, std::true_type{});
    } else {
      m_u32x4_t pkn[2], dkn[2];
#pragma unroll
      for (int c = 0; c < 8; ++c) b_chunk(c, pkn, dkn);
      pk[0] = pkn[0]; pk[1] = pkn[1]; dk[0] = dkn[0]; dk[1] = dkn[1];
    }
    write_t(std::integral_constant<int, 0>{}, dk);
    prefetch_c();
    AVF_FENCE();
    AVF_PHASE_MARK(3);
    if constexpr (KB > 1) stage_a(std::integral_constant<int, 1>{});
    AVF_PHASE_MARK(4);
    phase_c(std::integral_constant<int, 0>{});
    AVF_PHASE_MARK(7);
    // (here, not at the bottom of the slice in front of its barrier: the DMA issued at the top has long landed, and the
    //  dozen VALU instructions run beside the matrix pipe instead of on the slice's critical path)
    if (more) {
      // delta of the next slice: its dO rows 8 w .. 8 w + 7 were brought in by THIS wave (own vmcnt suffices)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const char* gn = ring + ((s + 1) & 1) * 8192 + 4096;
      // onext is logical chunk dch of the row: its partner sits in physical slot dch ^ swz(row)
      const uint4 gv = *reinterpret_cast<const uint4*>(gn + (drow & 31) * 128 + ((dch ^ m_swz(drow & 31)) << 4));
      float part = m_dot8(onext, gv);
      part = m_sum8(part);
      if (dch == 0) NDs[32 * (s + 1) + drow] = (32 * (s + 1) + drow < N) ? -part : 0.f;
    }
    if constexpr (KB > 1) {
      prefetch_c();
      AVF_FENCE();
      if constexpr (KB > 2) stage_a(std::integral_constant<int, 2>{});
      AVF_PHASE_MARK(4);
      phase_c(std::integral_constant<int, 1>{});
      AVF_PHASE_MARK(7);
    }
    if constexpr (KB > 2) {
      prefetch_c();
      AVF_FENCE();
      if constexpr (KB > 3) stage_a(std::integral_constant<int, 3>{});
      AVF_PHASE_MARK(4);
      phase_c(std::integral_constant<int, 2>{});
      AVF_PHASE_MARK(7);
    }
    if constexpr (KB > 3) {
      prefetch_c();
      AVF_FENCE();
      phase_c(std::integral_constant<int, 3>{});
    }
    AVF_PHASE_MARK(7);
    __syncthreads();
    AVF_PHASE_MARK(5);
  }
  kopq = 0;
  asm volatile("" : "+v"(kopq));
  dq_job(NS - 1, std::false_type{});

  // ---- dK, dV of the wave's keys ---------------------------------------------------------------------------------
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last accumulating MFMA has retired before the AGPRs are read
  const float kscale = 1.0f / kLog2E;  // dK = dS^T q' / log2(e) with q' = q log2(e) / sqrt(dh)
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const int key = 32 * (kb0 + j) + r;
    // A lane holds columns 8 g + 4 hf .. + 3 (g = 0..3) of each 32-column block of its key row: the lane pair (r, 0), (r, 1)
    // trades words (v_permlane32_swap, every lane active: outside the key < N branch) so that lane (r, 0) stores columns
    // 0 .. 15 and lane (r, 1) columns 16 .. 31 as 16-byte pieces - half as many store instructions, each twice as wide, to
    // the 64 different rows a wave-instruction touches here
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        uint32_t w[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (which) {
            w[g][0] = pack_bf16x2(dvacc[j][db][4 * g], dvacc[j][db][4 * g + 1]);
            w[g][1] = pack_bf16x2(dvacc[j][db][4 * g + 2], dvacc[j][db][4 * g + 3]);
          } else {
            w[g][0] = pack_bf16x2(dkacc[j][db][4 * g] * kscale, dkacc[j][db][4 * g + 1] * kscale);
            w[g][1] = pack_bf16x2(dkacc[j][db][4 * g + 2] * kscale, dkacc[j][db][4 * g + 3] * kscale);
          }
        }
        const auto a0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[2][0], false, false);
        const auto a1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[2][1], false, false);
        const auto b0 = __builtin_amdgcn_permlane32_swap(w[1][0], w[3][0], false, false);
        const auto b1 = __builtin_amdgcn_permlane32_swap(w[1][1], w[3][1], false, false);
        if (key < N) {
          bf16* out = dqkv + ((int64_t)b * N + key) * ld + (which ? 2 * I : I) + h * DH + 32 * db + 16 * hf;
          // lane (r, 0): [g0 own | g0 partner | g1 own | g1 partner]; lane (r, 1): [g2 partner | g2 own | g3 partner | g3 own]
          *reinterpret_cast<uint4*>(out) = make_uint4(a0[0], a1[0], a0[1], a1[1]);
          *reinterpret_cast<uint4*>(out + 8) = make_uint4(b0[0], b1[0], b0[1], b1[1]);
        }
      }
    if constexpr (MXK) {  // (outside the key < N branch: the lane pair (r, hf) trades words with every lane active)
      const int64_t row = (int64_t)b * N + (key < N ? key : N - 1);
#pragma unroll
      for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          // the values as stored: one v_cvt_pk_bf16_f32 per pair, widened again
          float v[16];
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const float r0 = which ? dvacc[j][db][i] : dkacc[j][db][i] * kscale;
            const float r1 = which ? dvacc[j][db][i + 1] : dkacc[j][db][i + 1] * kscale;
            const uint32_t w = pack_bf16x2(r0, r1);
            v[i] = __uint_as_float(w << 16);
            v[i + 1] = __uint_as_float(w & 0xffff0000u);
            t = fmaxf(t, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
          }
          const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
          t = fmaxf(__uint_as_float(x[0]), __uint_as_float(x[1]));
          float inv;
          const uint32_t sb = mx8_scale_byte(t, &inv);
          uint32_t w[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float v4[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            w[g] = mx8_pack4(v4, inv);  // columns 8 g + 4 hf .. + 3 of the block
          }
          // lane (r, 0) collects columns 0 .. 15 of the block, lane (r, 1) columns 16 .. 31: one 16-byte store each instead of
          // four 4-byte ones to 64 different rows per instruction (which cost 34 us at B = 64, N = 512)
          const auto s02 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
          const int col = (which ? 2 * I : I) + h * DH + 32 * db;
          if (key < N) {
            *reinterpret_cast<uint4*>(dq_q + row * ld + col + 16 * hf) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
            if (hf == 0) dq_s[row * (ld >> 5) + (col >> 5)] = (uint8_t)sb;
          }
        }
    }
  }
  AVF_PHASE_MARK(6);
  AVF_PHASE_FLUSH();
}

template <int KB, bool RAGGED, bool MASKED = false, bool MXO = false>
int m4_launch(const TimingScope* ts, const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv, int B,
              int N, int H, hipStream_t s, const uint8_t* keep = nullptr, uint8_t* dq_q = nullptr, uint8_t* dq_s = nullptr) {
  static PerDeviceOnce once;
  const MLayout L(KB, (N + 31) >> 5);
  if (once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_m4_kernel<KB, RAGGED, MASKED, MXO>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    AVF_REQUIRE(e == hipSuccess, "attn_bwd_m4: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    once.mark();
  }
  AVF_REQUIRE(L.total <= 160 * 1024, "attn_bwd_m4: %d bytes of LDS", L.total);
  launch_in_scope(ts, attn_bwd_m4_kernel<KB, RAGGED, MASKED, MXO>, dim3(B * H), dim3(256), (uint32_t)L.total, s, qkv, o, d_o, lse2,
                  dqkv, N, H, keep, dq_q, dq_s);
  return check_launch("attn_bwd_m4_kernel");
}

}  // namespace

// merged backward: dim_head 64, pre-scaled q, N <= 512 - the default at every such N since its epilogue stores went to 16 bytes
// (back to back, B = 32: N = 128 11.6 us against 18.9 for the two head-resident kernels, 196: 23.1 / 24.2, 256: 26.3 / 29.3; TFormer's
// 17 tokens: 0.438 -> 0.428 ms per step).  History of the threshold (AVF_ATTN_MERGED_MIN_N), from three key blocks per wave up (N >= 257):
// back to back in the harness it wins at N = 512 (B = 32 / 64: 77.3 / 155 us against 80.1 / 160.2 us) and loses at N = 324
// (46.2 against 44.2 us), but IN THE STEP the single launch - which reads q, k, v, dO once, not twice - is the faster one there
// too: C2 (N = 324) 2.107 / 2.113 ms per step against 2.141 / 2.129 with the two kernels (same box, alternating runs).
// AVF_ATTN_MERGED=0 turns it off, AVF_ATTN_MERGED_MIN_N moves the threshold.
bool attn_bwd_merged_ok(int N, int dh, bool q_prescaled) {
  static const int allow = [] {
    const char* e = tuning_env("AVF_ATTN_MERGED");
    return (e && *e) ? atoi(e) : 1;
  }();
  static const int min_n = [] {
    const char* e = tuning_env("AVF_ATTN_MERGED_MIN_N");
    return (e && *e) ? atoi(e) : 1;
  }();
  return allow && q_prescaled && dh == 64 && N <= 512 && N >= min_n;
}

int attn_bwd_merged(const TimingScope* ts, const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv,
                    int B, int N, int H, hipStream_t s, const void* keep, void* dq_q, void* dq_s) {
  AVF_REQUIRE(!dq_q || (dq_s && !keep && ((uintptr_t)dq_q & 3) == 0 && (3 * H * 64) % 32 == 0),
              "attn_bwd_merged: the MX-FP8 image of dqkv needs its scale buffer, no token mask and 4-byte alignment");
  AVF_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0 &&
                  ((uintptr_t)dqkv & 15) == 0,
              "attn_bwd_merged: misaligned pointers");
  const int KB = (((N + 31) >> 5) + 3) >> 2;
  if (keep) {  // token mask: the masked instantiation (which treats padded keys as dropped ones)
    switch (KB) {
      case 1: return m4_launch<1, true, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, (const uint8_t*)keep);
      case 2: return m4_launch<2, true, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, (const uint8_t*)keep);
#if AVF_M4_MAXKB >= 3
      case 3: return m4_launch<3, true, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, (const uint8_t*)keep);
#endif
#if AVF_M4_MAXKB >= 4
      case 4: return m4_launch<4, true, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, (const uint8_t*)keep);
#endif
    }
    AVF_REQUIRE(false, "attn_bwd_merged: N=%d out of range", N);
  }
  const bool ragged = N != 4 * KB * 32;
#define AVF_M4(K)                                                                                                          \
  do {                                                                                                                     \
    if (dq_q)                                                                                                              \
      return ragged ? m4_launch<K, true, false, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, nullptr, (uint8_t*)dq_q, (uint8_t*)dq_s) \
                    : m4_launch<K, false, false, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, nullptr, (uint8_t*)dq_q, (uint8_t*)dq_s); \
    return ragged ? m4_launch<K, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s)                                            \
                  : m4_launch<K, false>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s);                                          \
  } while (0)
  switch (KB) {
    case 1: AVF_M4(1);
    case 2: AVF_M4(2);
#if AVF_M4_MAXKB >= 3
    case 3: AVF_M4(3);
#endif
#if AVF_M4_MAXKB >= 4
    case 4: AVF_M4(4);
#endif
  }
#undef AVF_M4
  AVF_REQUIRE(false, "attn_bwd_merged: N=%d out of range", N);
}

}  // namespace avf
