// attn_bwd_merged.hip - throughput-mode attention backward as ONE kernel per (clip, head): dQ, dK and dV from a
// single recomputation of the probabilities, on v_mfma_f32_32x32x16_bf16.
//
// Reference: autograd of models/heads.py:222-237 (dots = q k^T * dh^-0.5, softmax over keys, out = attn v).
//
// The two head-resident kernels of attn_bf16.hip (dQ with the query on the lane, dK/dV with the key on the lane) each
// recompute S = q k^T and dP = dO v^T: 14 matrix products for 10 nominal.  Here every (query, key) tile is visited once:
//
//   one workgroup = 4 wavefronts (one per SIMD, the whole 512-register file each) = one (clip, head), N <= 512, dim_head 64;
//   wave w OWNS key blocks [w KB, (w+1) KB) of 32 keys: dK^T and dV^T of its keys stay in 64 KB accumulators for the whole
//   kernel, V fragments of its keys stay in registers, the head's K image is resident in LDS (row reads for S, transposed
//   reads for dQ); the workgroup sweeps the queries in SLICES of 32 rows (Q and dO slices arrive by LDS-DMA, one slice
//   ahead, one barrier per slice).  Per (slice, key block), key on the lane:
//     S  = Q K^T - lse2      (the row statistic rides in the C operand; q carries log2(e)/sqrt(dh))
//     dP = dO V^T - delta    (likewise)
//     P = exp2(S), dS = P * dP                                   (32 values per lane)
//     dV^T += dO^T P,  dK^T += Q^T dS                            (the accumulator tiles ARE the B operands)
//     dS crosses LDS once (8-byte stores of the accumulator rows, transposed reads back), dQ_partial += dS K
//   The four waves' dQ partials of a slice go to four fp32 slabs in LDS (two sets in rotation), are summed in a fixed
//   order and leave as bf16 rows at the top of the next slice (ds_add_f32 into one shared tile measured 600 cycles per
//   instruction: 55 % of the kernel).  delta = rowsum(dO * O) of the NEXT slice is computed at the bottom of each slice
//   from the dO rows already in LDS and an O chunk fetched at the top.
// No atomics anywhere, fixed summation order: bitwise deterministic.
//
// Rows past N: K rows are zero in the image and V fragments zero (so dS K and the discarded dK/dV columns are finite and
// contribute nothing), query rows are copies of row N-1 with lse = +inf (P = 0) and delta = 0.
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

// The kernel is built with -mllvm -amdgpu-mfma-vgpr-form (see _build.py): the builtin MFMAs keep C / D in the
// architectural VGPRs, so S, dP and dQ feed the VALU without v_accvgpr_read copies.  The dK^T / dV^T accumulators - 64
// registers per key block, up to all 256 accumulation registers - are pinned to the AGPR half by the "+a" constraint of
// these two-instruction statements (hipcc picks one MFMA form per function and cannot mix them itself).  s_nop 1: the
// B operands were just written by v_cvt_pk (VALU write -> MFMA operand read, two wait states; hipcc pads nothing
// inside an asm string).
__device__ __forceinline__ void m_mfma_pair_acc(f32x16_t& acc0, const bf16x8_t& a0, const bf16x8_t& b0, f32x16_t& acc1,
                                                const bf16x8_t& a1, const bf16x8_t& b1) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1"
               : "+a"(acc0), "+a"(acc1)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}

// 16-byte chunk permutation of a 128-byte row (8 chunks): conflict-free for the ds_read_b128 row fragments of the
// 32x32x16 operands (rows of equal parity inside one 16-lane service group get 8 distinct values) and for the
// ds_read_b64_tr_b16 fragments (rows r, r+2 of a 4-row block land in different 64-byte halves).
__device__ __forceinline__ int m_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

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

__device__ __forceinline__ bf16x8_t m_pack8(const f32x16_t& a, int s2) {  // registers 8 s2 .. 8 s2 + 7 as one k-step
  m_u32x4_t r = {pack_bf16x2(a[8 * s2 + 0], a[8 * s2 + 1]), pack_bf16x2(a[8 * s2 + 2], a[8 * s2 + 3]),
                 pack_bf16x2(a[8 * s2 + 4], a[8 * s2 + 5]), pack_bf16x2(a[8 * s2 + 6], a[8 * s2 + 7])};
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

__device__ __forceinline__ f32x16_t m_zero16() {
  f32x16_t z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// LDS map (bytes): K image | ring (2 slots x (Q 4 KB | dO 4 KB)) | NL | ND | T (4 x 2 KB) | R (2 buffers x 4 wave slabs x
// 8 KB: the waves' dQ partials of a slice, [32 q][64 d] fp32 each) | V image of the key blocks whose V fragments do not
// stay in registers (blocks VR .. KB-1 of every wave, [wave][block][32][128 B])
struct MLayout {
  int kimg, ring, nl, nd, t, r, vimg, total;
  __host__ __device__ MLayout(int KB, int VR, int NS) {
    kimg = 0;
    ring = kimg + 4 * KB * 32 * 128;
    nl = ring + 2 * 8192;
    nd = nl + NS * 32 * 4;
    t = nd + NS * 32 * 4;
    r = t + 4 * 2048;
    vimg = r + 2 * 4 * 8192;
    total = vimg + 4 * (KB - VR) * 4096;
  }
};

// key blocks per wave whose V fragments stay in registers (the rest would be read from an LDS V image every slice; with
// the fp32 slabs of the dQ reduction there is no room for one at 4 blocks per wave, and the registers just suffice)
__host__ __device__ constexpr int m4_vr(int KB) { return KB; }
// key blocks per wave whose K fragments (row and transposed: 48 registers) may stay in registers
__host__ __device__ constexpr int m4_holdk(int KB) { return KB <= 2 ? KB : 0; }

// RAGGED: N < 4 KB 32 - key blocks that hold rows past N mask them (P = 0: -lse2 alone can be a large positive
// exponent when every score of a row is very negative)
template <int KB, bool RAGGED>
__global__ __launch_bounds__(256) void attn_bwd_m4_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                          const bf16* __restrict__ d_o, const float* __restrict__ lse2,
                                                          bf16* __restrict__ dqkv, int N, int H) {
  constexpr int DH = 64;
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
  const int NS = (N + 31) >> 5;   // query slices == key blocks with valid rows
  constexpr int VR = m4_vr(KB);
  constexpr int HOLDK = m4_holdk(KB);
  const MLayout L(KB, VR, NS);
  char* kimg = m_smem + L.kimg;
  char* vimg = m_smem + L.vimg + wave * (KB - VR) * 4096;
  char* ring = m_smem + L.ring;
  float* NLs = reinterpret_cast<float*>(m_smem + L.nl);
  float* NDs = reinterpret_cast<float*>(m_smem + L.nd);
  char* timg = m_smem + L.t + wave * 2048;
  float* Rs = reinterpret_cast<float*>(m_smem + L.r);
  const int kb0 = wave * KB;                                   // first key block of this wave

  // ---- per-lane LDS byte offsets -------------------------------------------------------------------------------
  // row fragment of a 32-row block (128-byte rows): row r, logical chunk 2 ks + hf
  int off_row[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) off_row[ks] = r * 128 + (((2 * ks + hf) ^ m_swz(r)) << 4);
  // transposed fragments: 16-lane group G = lane>>4 (gg = G&1: column half, hh = G>>1 = hf), lane 4 qq + p of the group
  const int li = lane & 15, gg = (lane >> 4) & 1, qq = li >> 2, p = li & 3;
  // (a) accumulator k-order (element j of k-step s2 <-> row 16 s2 + 8 (j>>2) + 4 hf + (j&3)): rows 4 hf + qq and + 8
  int off_tra[2][2];
  // (b) natural k-order (element j <-> row 16 ks + 8 hf + j): rows 8 hf + qq and + 4
  int off_trn[2][2];
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int ch = 4 * db + 2 * gg + (p >> 1), byte = (p & 1) * 8;
    const int ra = 4 * hf + qq, rn = 8 * hf + qq;
    off_tra[db][0] = ra * 128 + ((ch ^ m_swz(ra)) << 4) + byte;
    off_tra[db][1] = (ra + 8) * 128 + ((ch ^ m_swz(ra + 8)) << 4) + byte;
    off_trn[db][0] = rn * 128 + ((ch ^ m_swz(rn)) << 4) + byte;
    off_trn[db][1] = (rn + 4) * 128 + ((ch ^ m_swz(rn + 4)) << 4) + byte;
  }
  // T image (this wave's dS^T block, [key][query], 64-byte rows, 8-byte slot u at u ^ ((key>>1)&7))
  int off_tw[4];  // store of registers 4 g .. 4 g + 3: row r (key), slot 2 g + hf
#pragma unroll
  for (int g = 0; g < 4; ++g) off_tw[g] = r * 64 + (((2 * g + hf) ^ ((r >> 1) & 7)) << 3);
  int off_tr_t[2];  // transposed read: keys 8 hf + qq (+4), slot 4 gg + p; k-step ks adds 16 rows = 1024 bytes
  {
    const int k0 = 8 * hf + qq, k1 = k0 + 4, u = 4 * gg + p;
    off_tr_t[0] = k0 * 64 + ((u ^ ((k0 >> 1) & 7)) << 3);
    off_tr_t[1] = k1 * 64 + ((u ^ ((k1 >> 1) & 7)) << 3);
  }

  // ---- prologue --------------------------------------------------------------------------------------------------
  AVF_PHASE_INIT();
  const int lrow = lane >> 3, lslot = lane & 7;
  // K image: pieces of 8 rows; rows past N are fetched from row N-1 and zeroed below
  for (int pc = wave; pc < 16 * KB; pc += 4) {
    const int row = pc * 8 + lrow;
    const int src = row < N ? row : N - 1;
    m_glds16(kbase + (int64_t)src * ld + ((lslot ^ m_swz(row)) << 3), kimg + pc * 1024);
  }
  if constexpr (VR < KB) {  // V rows of this wave's blocks VR .. KB-1 (rows past N: fetched from row N-1, zeroed below)
    for (int pc = 0; pc < 4 * (KB - VR); ++pc) {
      const int lr = pc * 8 + lrow;                                  // row inside the wave's V image
      const int row = 32 * (kb0 + VR) + lr;                          // key
      const int src = row < N ? row : N - 1;
      m_glds16(vbase + (int64_t)src * ld + ((lslot ^ m_swz(lr)) << 3), vimg + pc * 1024);
    }
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
  bf16x8_t vfr[VR][4];
#pragma unroll
  for (int j = 0; j < VR; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int key = 32 * (kb0 + j) + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (key < N) v = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * ld + 16 * ks + 8 * hf);
      vfr[j][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  // statistics: NL = -lse2 (-inf past N: P = 0)
  for (int q = tid; q < NS * 32; q += 256) NLs[q] = q < N ? -lse2[(int64_t)bh * N + q] : -INFINITY;
  // delta of slice 0 straight from global memory: thread t = row t>>3, chunk t&7
  const int drow = tid >> 3, dch = tid & 7;
  {
    float part = 0.f;
    if (drow < N) {
      const uint4 ov = *reinterpret_cast<const uint4*>(obase + (int64_t)drow * I + dch * 8);
      const uint4 gv = *reinterpret_cast<const uint4*>(gbase + (int64_t)drow * I + dch * 8);
      part = m_dot8(ov, gv);
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (dch == 0) NDs[drow] = -part;
  }
  // the builtin form is VISIBLE to hipcc's wait-count pass (it retires the V fragment loads in its books: with the asm form
  // alone it put a vmcnt(0) in front of their first use in EVERY slice, exposing the latency of that slice's own DMA)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt unconstrained
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (N < 4 * KB * 32) {  // zero the K (and V image) rows past N (wave-uniform condition)
    for (int idx = N * 8 + tid; idx < 4 * KB * 32 * 8; idx += 256)
      *reinterpret_cast<uint4*>(kimg + idx * 16) = make_uint4(0, 0, 0, 0);
    if constexpr (VR < KB) {
      for (int idx = lane; idx < (KB - VR) * 32 * 8; idx += 64) {  // each wave its own image
        const int key = 32 * (kb0 + VR) + (idx >> 3);
        if (key >= N) *reinterpret_cast<uint4*>(vimg + idx * 16) = make_uint4(0, 0, 0, 0);
      }
    }
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

  // dQ rows of slice s leave the reduction tile (thread t: row t>>3, 8 columns)
  auto finish_slice = [&](int s) {
    const float4* rp = reinterpret_cast<const float4*>(Rs + (s & 1) * 8192 + drow * 64 + dch * 8);
    float4 a = rp[0], c = rp[1];
#pragma unroll
    for (int w = 1; w < 4; ++w) {  // fixed order: wave 0 + wave 1 + wave 2 + wave 3 (deterministic)
      const float4 a2 = rp[w * 512], c2 = rp[w * 512 + 1];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      c.x += c2.x; c.y += c2.y; c.z += c2.z; c.w += c2.w;
    }
    const int q = 32 * s + drow;
    if (q < N) {
      uint4 w;
      w.x = pack_bf16x2(a.x * qscale, a.y * qscale);
      w.y = pack_bf16x2(a.z * qscale, a.w * qscale);
      w.z = pack_bf16x2(c.x * qscale, c.y * qscale);
      w.w = pack_bf16x2(c.z * qscale, c.w * qscale);
      *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + q) * ld + h * DH + dch * 8) = w;
    }
  };

  // one (slice, key block): everything between the operand fragments and the three accumulations
  auto block = [&](auto jtag, const char* qsl, const char* gsl, const bf16x8_t (&qa)[4], const bf16x8_t (&ga)[4],
                   const float* nlp, const float* ndp, f32x16_t (&dq)[2], int kopq) {
    constexpr int j = decltype(jtag)::value;
    __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting the next block's reads over this one (registers)
    const char* kblk = kimg + (kb0 + j) * 4096 + (j < HOLDK ? 0 : kopq);
    const char* vblk = vimg + (j - VR) * 4096 + kopq;
    // the row statistics of the lane's 16 query rows ARE the initial accumulators (broadcast LDS reads)
    f32x16_t sacc, pacc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(nlp + 8 * g);
      const float4 c = *reinterpret_cast<const float4*>(ndp + 8 * g);
      sacc[4 * g + 0] = a.x; sacc[4 * g + 1] = a.y; sacc[4 * g + 2] = a.z; sacc[4 * g + 3] = a.w;
      pacc[4 * g + 0] = c.x; pacc[4 * g + 1] = c.y; pacc[4 * g + 2] = c.z; pacc[4 * g + 3] = c.w;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t kf = m_row_frag(kblk + off_row[ks]);
      sacc = AVF_MFMA32(qa[ks], kf, sacc);
      bf16x8_t vf;
      if constexpr (j < VR) vf = vfr[j][ks];
      else vf = m_row_frag(vblk + off_row[ks]);
      pacc = AVF_MFMA32(ga[ks], vf, pacc);
    }
    if constexpr (RAGGED) {
      if (32 * (kb0 + j) + 32 > N) {  // wave-uniform: the block holds padded keys
        const bool dead = 32 * (kb0 + j) + r >= N;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = dead ? -INFINITY : sacc[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sacc[i] = __builtin_amdgcn_exp2f(sacc[i]);
      pacc[i] = sacc[i] * pacc[i];
    }
    bf16x8_t pk[2], dk[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      pk[s2] = m_pack8(sacc, s2);
      dk[s2] = m_pack8(pacc, s2);
    }
    // dS^T to the wave's T image
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const m_u32x4_t w = __builtin_bit_cast(m_u32x4_t, dk[s2]);
      *reinterpret_cast<uint2*>(timg + off_tw[2 * s2]) = make_uint2(w[0], w[1]);
      *reinterpret_cast<uint2*>(timg + off_tw[2 * s2 + 1]) = make_uint2(w[2], w[3]);
    }
    // dV^T += dO^T P ; dK^T += Q^T dS
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t gt = m_tr_frag(gsl + s2 * 2048 + off_tra[db][0], gsl + s2 * 2048 + off_tra[db][1]);
        const bf16x8_t qt = m_tr_frag(qsl + s2 * 2048 + off_tra[db][0], qsl + s2 * 2048 + off_tra[db][1]);
        m_mfma_pair_acc(dvacc[j][db], gt, pk[s2], dkacc[j][db], qt, dk[s2]);
      }
    // dQ[q][d] += dS[q][key] K[key][d]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t da = m_tr_frag(timg + ks * 1024 + off_tr_t[0], timg + ks * 1024 + off_tr_t[1]);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x8_t kt = m_tr_frag(kblk + ks * 2048 + off_trn[db][0], kblk + ks * 2048 + off_trn[db][1]);
        dq[db] = AVF_MFMA32(da, kt, dq[db]);
      }
    }
  };

  // ---- the sweep over the query slices ---------------------------------------------------------------------------
  for (int s = 0; s < NS; ++s) {
    const char* qsl = ring + (s & 1) * 8192;
    const char* gsl = qsl + 4096;
    uint4 onext = make_uint4(0, 0, 0, 0);
    const bool more = s + 1 < NS;
    if (more) {
      issue_slice(s + 1);
      const int q = 32 * (s + 1) + drow;
      if (q < N) onext = *reinterpret_cast<const uint4*>(obase + (int64_t)q * I + dch * 8);
    }
    if (s > 0) finish_slice(s - 1);
    AVF_PHASE_MARK(1);

    bf16x8_t qa[4], ga[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qa[ks] = m_row_frag(qsl + off_row[ks]);
      ga[ks] = m_row_frag(gsl + off_row[ks]);
    }
    const float* nl = NLs + 32 * s + 4 * hf;
    const float* nd = NDs + 32 * s + 4 * hf;
    // the K (and V) images never change, so the compiler would hoist every fragment read of them out of the slice loop
    // - 48 registers per key block; an opaque per-slice copy of the image offset keeps the reads inside (HOLDK blocks
    // are left to the compiler: their fragments stay in registers for the whole kernel)
    int kopq = 0;
    asm volatile("" : "+v"(kopq));
    f32x16_t dq[2] = {m_zero16(), m_zero16()};
    // every wave runs all its KB blocks, padded ones included (zero K rows and V, masked scores): the slice barrier
    // would make a wave that skipped them wait for the others anyway, and one straight-line body keeps the 64 KB
    // accumulators in place (two code paths made the compiler shuffle all of them at the join)
    block(std::integral_constant<int, 0>{}, qsl, gsl, qa, ga, nl, nd, dq, kopq);
    if constexpr (KB > 1) block(std::integral_constant<int, 1>{}, qsl, gsl, qa, ga, nl, nd, dq, kopq);
    if constexpr (KB > 2) block(std::integral_constant<int, 2>{}, qsl, gsl, qa, ga, nl, nd, dq, kopq);
    if constexpr (KB > 3) block(std::integral_constant<int, 3>{}, qsl, gsl, qa, ga, nl, nd, dq, kopq);
    AVF_PHASE_MARK(2);
    // the wave's dQ partial joins the slice's reduction tile: register i <-> row (i&3) + 8 (i>>2) + 4 hf, column 32 db + r
    {
      float* rt = Rs + (s & 1) * 8192 + wave * 2048 + r;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) rt[((i & 3) + 8 * (i >> 2) + 4 * hf) * 64 + 32 * db] = dq[db][i];
    }
    AVF_PHASE_MARK(3);
    if (more) {
      // delta of the next slice: its dO rows 8 w .. 8 w + 7 were brought in by THIS wave (own vmcnt suffices)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const char* gn = ring + ((s + 1) & 1) * 8192 + 4096;
      // onext is logical chunk dch of the row: its partner sits in physical slot dch ^ swz(row)
      const uint4 gv = *reinterpret_cast<const uint4*>(gn + (drow & 31) * 128 + ((dch ^ m_swz(drow & 31)) << 4));
      float part = m_dot8(onext, gv);
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      if (dch == 0) NDs[32 * (s + 1) + drow] = (32 * (s + 1) + drow < N) ? -part : 0.f;
    }
    AVF_PHASE_MARK(4);
    __syncthreads();
    AVF_PHASE_MARK(5);
  }
  finish_slice(NS - 1);

  // ---- dK, dV of the wave's keys ---------------------------------------------------------------------------------
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last accumulating MFMA has retired before the AGPRs are read
  const float kscale = 1.0f / kLog2E;  // dK = dS^T q' / log2(e) with q' = q log2(e) / sqrt(dh)
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const int key = 32 * (kb0 + j) + r;
    if (key < N) {
      bf16* outk = dqkv + ((int64_t)b * N + key) * ld + I + h * DH;
      bf16* outv = outk + I;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * db + 8 * g + 4 * hf;
          store4<bf16>(outk + d, make_float4(dkacc[j][db][4 * g] * kscale, dkacc[j][db][4 * g + 1] * kscale,
                                             dkacc[j][db][4 * g + 2] * kscale, dkacc[j][db][4 * g + 3] * kscale));
          store4<bf16>(outv + d, make_float4(dvacc[j][db][4 * g], dvacc[j][db][4 * g + 1], dvacc[j][db][4 * g + 2],
                                             dvacc[j][db][4 * g + 3]));
        }
    }
  }
  AVF_PHASE_MARK(6);
  AVF_PHASE_FLUSH();
}

template <int KB, bool RAGGED>
int m4_launch(const TimingScope* ts, const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv, int B,
              int N, int H, hipStream_t s) {
  static PerDeviceOnce once;
  const MLayout L(KB, m4_vr(KB), (N + 31) >> 5);
  if (once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_m4_kernel<KB, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    AVF_REQUIRE(e == hipSuccess, "attn_bwd_m4: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    once.mark();
  }
  AVF_REQUIRE(L.total <= 160 * 1024, "attn_bwd_m4: %d bytes of LDS", L.total);
  launch_in_scope(ts, attn_bwd_m4_kernel<KB, RAGGED>, dim3(B * H), dim3(256), (uint32_t)L.total, s, qkv, o, d_o, lse2, dqkv, N, H);
  return check_launch("attn_bwd_m4_kernel");
}

}  // namespace

// merged backward: dim_head 64, pre-scaled q, N <= 512.  AVF_ATTN_MERGED=0 restores the two-kernel path;
// AVF_ATTN_MERGED_MIN_N sets the shortest sequence that takes it.
bool attn_bwd_merged_ok(int N, int dh, bool q_prescaled) {
  static const int allow = [] {
    const char* e = getenv("AVF_ATTN_MERGED");
    return (e && *e) ? atoi(e) : 0;  // off until it beats the two-kernel path
  }();
  static const int min_n = [] {
    const char* e = getenv("AVF_ATTN_MERGED_MIN_N");
    return (e && *e) ? atoi(e) : 1;
  }();
  return allow && q_prescaled && dh == 64 && N <= 512 && N >= min_n;
}

int attn_bwd_merged(const TimingScope* ts, const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv,
                    int B, int N, int H, hipStream_t s) {
  AVF_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0 &&
                  ((uintptr_t)dqkv & 15) == 0,
              "attn_bwd_merged: misaligned pointers");
  const int KB = (((N + 31) >> 5) + 3) >> 2;
  const bool ragged = N != 4 * KB * 32;
#define AVF_M4(K) return ragged ? m4_launch<K, true>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s) \
                                : m4_launch<K, false>(ts, qkv, o, d_o, lse2, dqkv, B, N, H, s)
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
