// attn_bf16.hip - throughput-mode attention core: LDS-tiled flash attention forward and backward on
// v_mfma_f32_16x16x32_bf16, fp32 online-softmax statistics, bf16 operands.
//
// Reference: models/heads.py:222-237 (scores * dh^-0.5, softmax over keys, P v, head merge).  The
// [B,H,N,N] score tensor the reference materialises never exists here.  q, k, v are read in place from
// the QKV GEMM output [B*N, 3I] (absorbing 'b n (h d) -> b h n d'), o is written as [B*N, I]
// ('b h n d -> b n (h d)').
//
// Orientation trick (the accumulator of one MFMA is the operand of the next, no LDS round trip):
//   forward / dQ : S^T = K Q^T    -> lane owns a query column; P^T (or dS^T) packs straight into the
//                  B operand of  O^T = V^T P^T  /  dQ^T = K^T dS^T ; V^T / K^T fragments come from
//                  ds_read_b64_tr_b16 on the row-major tile.
//   dK / dV      : S = Q K^T      -> lane owns a key column;  dV^T = dO^T P,  dK^T = Q^T dS.
// k-slot map of a packed accumulator pair / transposed fragment (32 reduction rows per k-step):
//   slot j of lane group g  <->  row 16*(j>>2) + 4*g + (j&3).
#include <type_traits>

#include "common.hpp"

namespace avf {

namespace {

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

constexpr float LOG2E = 1.4426950408889634f;

// Phase stamps for tools/diag/attn_phases.hip (that file defines these and includes this one); nothing in the product.
#ifndef AVF_PHASE_MARK
#define AVF_PHASE_INIT()
#define AVF_PHASE_MARK(slot)
#define AVF_PHASE_FLUSH()
#endif

// 1-D grid of nblk * B*H blocks.  Hardware deals consecutive block ids round-robin over the 8 XCDs; this
// bijection hands each XCD a CONTIGUOUS range of logical ids, so the row blocks of one (batch, head) - which
// share that head's K/V (or Q/dO) - run on one XCD and hit in its L2 instead of re-fetching from HBM.
__device__ __forceinline__ void block_coords(int nblk, int& blk, int& bh) {
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, x = id & 7;
  const int lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  bh = lid / nblk;
  blk = lid - bh * nblk;
}

template <int LD>
__device__ __forceinline__ bf16x8_t tr_frag(const lds_char* tile, int row_base, int col_base, int li, int lg) {
  const lds_char* p0 = tile + (row_base + 4 * lg + (li >> 2)) * LD + (col_base + 4 * (li & 3)) * 2;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * LD));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int LD>
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row, int col) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * LD + col * 2);
}

__device__ __forceinline__ bf16x8_t pack_pair(const f32x4_t& a, const f32x4_t& b) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  u32x4_t r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, r);
}

__device__ __forceinline__ bf16x8_t load_frag_global(const bf16* p, bool ok) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (ok) v = *reinterpret_cast<const uint4*>(p);
  return __builtin_bit_cast(bf16x8_t, v);
}

// stage ROWS rows x DH bf16 from global (row stride ld elements) into an LDS tile with row stride LD bytes;
// rows >= nvalid are zero-filled.  Split in issue (global -> regs) / commit (regs -> LDS).
template <int DH, int ROWS>
struct TileStager {
  static constexpr int CPR = DH / 8;                    // 16-byte chunks per row
  static constexpr int PER = (ROWS * CPR + 255) / 256;  // chunks per thread
  uint4 r[PER];
  __device__ __forceinline__ void issue(const bf16* src, int64_t ld, int row0, int nvalid, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR, c = idx - row * CPR;
      r[i] = make_uint4(0, 0, 0, 0);
      if (idx < ROWS * CPR && row < nvalid) r[i] = *reinterpret_cast<const uint4*>(src + (int64_t)(row0 + row) * ld + c * 8);
    }
  }
  template <int LD>
  __device__ __forceinline__ void commit(char* tile, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR, c = idx - row * CPR;
      if (idx < ROWS * CPR) *reinterpret_cast<uint4*>(tile + row * LD + c * 16) = r[i];
    }
  }
};

// =============================================================================================
// forward
// =============================================================================================
// One row's head columns (DB blocks of 16; a lane holds columns 16 d + 4 lg .. + 3 of every block d) as 16-byte stores: the
// lane groups lg, lg + 1 trade the words of blocks d, d + 1 (v_permlane16_swap), after which an even group holds 8 consecutive
// columns of block d and an odd one 8 of block d + 1.  The partner lane (same row, lg +- 1) must be active: call it under a
// predicate on the ROW only.  (8-byte stores to 16 rows per instruction cost the merged backward kernel 6 %.)
template <int DB, typename Get>
__device__ __forceinline__ void store_row_pairs(bf16* row, int lg, Get&& get) {
  static_assert(DB % 2 == 0, "column blocks come in pairs");
#pragma unroll
  for (int d = 0; d < DB; d += 2) {
    const f32x4_t a = get(d), b = get(d + 1);
    const uint32_t a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
    const uint32_t b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    *reinterpret_cast<uint4*>(row + ((lg & 1) ? (d + 1) * 16 + 4 * (lg - 1) : d * 16 + 4 * lg)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
  }
}

__device__ __forceinline__ float vmax(float a, float b) {  // v_max_f32 without the NaN-canonicalising pre-ops
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// max over the four lanes {li, li+16, li+32, li+48} (the four key groups of one query column), VALU only
__device__ __forceinline__ float colmax4(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = vmax(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return vmax(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// sum over the four lanes {li, li+16, li+32, li+48}
__device__ __forceinline__ float colsum4(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                            float* __restrict__ lse2, int /*B*/, int N, int H, int qs) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int KLD = DH * 2 + 32;  // K tile: row reads (ds_read_b128); +32 B keeps them conflict-free (PMC-checked)
  constexpr int VLD = DH * 2 + 32;  // V tile: transposed reads, 8 consecutive rows -> 8 distinct bank windows
  constexpr int STAGE = 64 * KLD + 64 * VLD;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int blk, bh;
  block_coords((N + 127) / 128, blk, bh);
  const int b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const int q0 = blk * 128 + wave * 32;
  const bool active = __builtin_amdgcn_readfirstlane(q0) < N;
  const float c = qs ? 1.0f : LOG2E / sqrtf((float)DH);  // qs: q already carries log2(e)/sqrt(dh)
  AVF_PHASE_INIT();

  bf16x8_t fq[2][KS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int q = q0 + qb * 16 + li;
      fq[qb][ks] = load_frag_global(qbase + (int64_t)q * ld + ks * 32 + 8 * lg, q < N);
    }

  f32x4_t ot[DB][2];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) ot[d][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};

  TileStager<DH, 64> sk, sv;
  const int nt = (N + 63) / 64;
  {
    const int nv = N < 64 ? N : 64;
    sk.issue(kbase, ld, 0, nv, tid);
    sv.issue(vbase, ld, 0, nv, tid);
    sk.template commit<KLD>(smem, tid);
    sv.template commit<VLD>(smem + 64 * KLD, tid);
  }
  __syncthreads();
  AVF_PHASE_MARK(0);
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const int r0 = (t + 1) * 64;
      const int nv = (N - r0) < 64 ? (N - r0) : 64;
      sk.issue(kbase, ld, r0, nv, tid);
      sv.issue(vbase, ld, r0, nv, tid);
    }
    AVF_PHASE_MARK(1);
    const char* kt = smem + cur * STAGE;
    const lds_char* vt = (const lds_char*)(smem + cur * STAGE + 64 * KLD);

    // One K/V tile.  TAIL = the tile holds keys past N: only its valid 16-key blocks are computed and the
    // rest is masked; full tiles take the branch-free instantiation.  Waves whose 32 query rows all lie past N
    // only help with staging and barriers.
    auto tile_body = [&](auto tail_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      const int nkb = TAIL ? (N - t * 64 + 15) / 16 : 4;
      // S^T[key][q] = K Q^T
      f32x4_t st[4][2];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        st[kb][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        st[kb][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (!TAIL || kb < nkb) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t fk = row_frag<KLD>(kt, kb * 16 + li, ks * 32 + 8 * lg);
            st[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[0][ks], st[kb][0], 0, 0, 0);
            st[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[1][ks], st[kb][1], 0, 0, 0);
          }
        }
      }
      AVF_PHASE_MARK(2);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (TAIL && (t * 64 + kb * 16 + 4 * lg + r >= N)) st[kb][qb][r] = -INFINITY;
            tmax = fmaxf(tmax, st[kb][qb][r]);  // raw scores: the scale c > 0 commutes with max
          }
        tmax = colmax4(tmax);
        const float mn = fmaxf(m[qb], tmax * c);
        const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
        m[qb] = mn;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][qb][r], c, -mn));  // 2^(c*s - m): one FMA + exp
            st[kb][qb][r] = pv;
            ps += pv;
          }
        lsum[qb] = lsum[qb] * alpha + ps;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          ot[d][qb][0] *= alpha; ot[d][qb][1] *= alpha; ot[d][qb][2] *= alpha; ot[d][qb][3] *= alpha;
        }
      }
      AVF_PHASE_MARK(3);
      // O^T[d][q] += V^T P^T
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (TAIL && 2 * s >= nkb) continue;
        const bf16x8_t p0 = pack_pair(st[2 * s][0], st[2 * s + 1][0]);
        const bf16x8_t p1 = pack_pair(st[2 * s][1], st[2 * s + 1][1]);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const bf16x8_t fv = tr_frag<VLD>(vt, 32 * s, d * 16, li, lg);
          ot[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, p0, ot[d][0], 0, 0, 0);
          ot[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, p1, ot[d][1], 0, 0, 0);
        }
      }
    };
    if (active) {
      if (t * 64 + 64 > N) tile_body(std::true_type{});
      else tile_body(std::false_type{});
    }
    AVF_PHASE_MARK(4);
    if (t + 1 < nt) {
      sk.template commit<KLD>(smem + (cur ^ 1) * STAGE, tid);
      sv.template commit<VLD>(smem + (cur ^ 1) * STAGE + 64 * KLD, tid);
    }
    AVF_PHASE_MARK(5);
    __syncthreads();
    AVF_PHASE_MARK(6);
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = lsum[qb];
    l = colsum4(l);
    const int q = q0 + qb * 16 + li;
    if (q < N) {
      const float inv = 1.0f / l;
      bf16* orow = o + ((int64_t)b * N + q) * I + h * DH;
      store_row_pairs<DB>(orow, lg, [&](int d) { return ot[d][qb] * inv; });
      if (lg == 0) lse2[(int64_t)bh * N + q] = m[qb] + log2f(l);
    }
  }
  AVF_PHASE_MARK(7);
  AVF_PHASE_FLUSH();
}

// =============================================================================================
// head-resident variants (dim_head 64, N <= RES_MAX_N): ONE workgroup per (batch, head) keeps the head's whole K and
// V (or Q and dO) in LDS - 128 B per row, N rounded up to 32 rows - so nothing is staged twice, there is no ring and
// no per-tile barrier.  The N rows are cut into V = ceil(N/32) groups of 32 query (key) rows; the workgroup has
// W = ceil(V / passes) wavefronts (passes = ceil(V/12): at most 12 waves, three per SIMD at <= 168 VGPRs) and wave w
// takes groups w, w+W, ...  After the load phase the waves free-run over the key (query) tiles, which lets the MFMA
// phase of one wave overlap the softmax VALU phase of another on the same SIMD.
//
// LDS image: written by LDS-DMA (global_load_lds_dwordx4, 64 lanes x 16 B = 8 rows per instruction, no staging
// VGPRs), so it is lane-linear per instruction; bank conflicts are avoided by permuting the SOURCE chunk instead:
// row r holds logical 16-B chunk c at slot c ^ res_swz(r), res_swz(r) = ((r>>1)&3)<<1.  That is conflict-free
// (SQ_LDS_BANK_CONFLICT = 0) for the ds_read_b128 row fragments (lane groups of 16: rows x two adjacent chunks) and
// for ds_read_b64_tr_b16 (lane groups of 32: 8 consecutive rows x two adjacent chunks).  Rows past N are loaded
// from row N-1 (finite values; their probabilities are exactly 0).
// Arrival: two barriers only - after the first RES_A tiles have landed, and after everything has (see ResLoader).
// Forward softmax: running maximum with LAZY rescaling - the accumulators are rescaled only when some row's maximum
// grew by more than 2^RES_TAU since the last rescale (a wave-uniform branch, rare after the first tile), so
// probabilities stay <= 2^RES_TAU; the row sums come from the MFMA (a ones fragment as a 65th value row), the
// cross-lane maximum from v_permlane{16,32}_swap instead of LDS permutes.  lse2 = m + log2(l) is exact whatever m is.
// =============================================================================================
constexpr int RES_MAX_N = 576;    // (N rounded to 32) * 256 B + the dK/dV kernel's statistics <= 160 KB
constexpr int RES_A = 1;          // tiles whose DMA is issued before any compute (1 vs 2: -2 % backward, same forward)
constexpr float RES_TAU = 6.0f;   // log2 of the largest probability kept before a rescale

__device__ __forceinline__ void glds16(const void* g, char* l) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 16, 0, 0);
}

// The same LDS-DMA issued as inline asm.  hipcc (ROCm 7.2) drains vmcnt to 0 in front of every compiler-visible LDS read
// that follows a compiler-visible LDS-DMA (its wait-count pass takes the DMA for a store that may alias the read): the
// pieces fed from hooks inside the first tile's compute then cost a full memory round trip each.  An asm DMA is opaque to
// that pass; the kernels' own wait_all_loads() + s_barrier order the LDS reads of a tile after its arrival, and vmcnt is
// in-order, so the compiler's own counted waits for its global loads only become more conservative.  AVF_ATTN_DMA_ASM=0
// at build time (-DAVF_ATTN_DMA_ASM=0) restores the builtin.
#ifndef AVF_ATTN_DMA_ASM
#define AVF_ATTN_DMA_ASM 1
#endif
__device__ __forceinline__ void glds16_res(const void* g, char* l) {
#if AVF_ATTN_DMA_ASM
  typedef __attribute__((address_space(3))) char lds_c;
  const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_c*)l);
  uint32_t keep;  // m0 is a reserved register: saved and restored, so whatever the compiler keeps in it survives
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(la), "v"(g)
               : "memory");
#else
  glds16(g, l);
#endif
}

__device__ __forceinline__ void glds4(const void* g, char* l) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 4, 0, 0);
}

__device__ __forceinline__ void wait_all_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// dot product of two 8-element bf16 fragments, fp32
__device__ __forceinline__ float dot8(const bf16x8_t& x, const bf16x8_t& y) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  const u32x4_t a = __builtin_bit_cast(u32x4_t, x), b = __builtin_bit_cast(u32x4_t, y);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = fmaf(__uint_as_float(a[i] << 16), __uint_as_float(b[i] << 16), acc);
    acc = fmaf(__uint_as_float(a[i] & 0xffff0000u), __uint_as_float(b[i] & 0xffff0000u), acc);
  }
  return acc;
}

__device__ __forceinline__ bf16x8_t ones_frag() {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  u32x4_t r = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  return __builtin_bit_cast(bf16x8_t, r);
}

__device__ __forceinline__ bf16x8_t lds_row_frag(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

__device__ __forceinline__ bf16x8_t lds_tr_frag(const char* p) {  // rows +0 and +16 of a 32-row k-step
  const lds_char* q = (const lds_char*)p;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)q);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q + 16 * 128));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// DMA schedule of a head-resident kernel.  Two [rows, 64] bf16 operands a, b (row strides lda, ldb elements) go to
// their LDS images in 8-row pieces, ordered a0 b0 a1 b1 ... (16 pieces per 64-row tile, so arrival is in tile order);
// piece p belongs to wave p mod W and each wave walks its pieces with a cursor.  Phase 1 (the first RES_A tiles) is
// issued up front; the rest is issued one piece at a time from hooks inside the first tile's compute, so no wave sits
// in a full memory queue while the SIMD idles; the second barrier (after the RES_A-th tile) waits for what is left.
struct ResLoader {
  const bf16 *a, *b;
  int64_t lda, ldb;
  char *la, *lb;
  int npieces, N, W, lrow, chunk, cur;
  __device__ __forceinline__ void init(const bf16* a_, int64_t lda_, char* la_, const bf16* b_, int64_t ldb_, char* lb_,
                                       int rows_padded, int N_, int wave, int W_, int lane) {
    a = a_; b = b_; lda = lda_; ldb = ldb_; la = la_; lb = lb_;
    npieces = rows_padded / 4;  // 2 operands x rows/8
    N = N_; W = W_; cur = wave;
    lrow = lane >> 3;
    chunk = (lane & 7) ^ (((lrow >> 1) & 3) << 1);
  }
  __device__ __forceinline__ void issue_one() {  // wave-uniform; no-op once the wave's pieces are out
    if (cur < npieces) {
      const int i = cur >> 1;
      int row = i * 8 + lrow;
      row = row < N ? row : N - 1;
      if (cur & 1) glds16_res(b + (int64_t)row * ldb + chunk * 8, lb + i * 1024);
      else glds16_res(a + (int64_t)row * lda + chunk * 8, la + i * 1024);
      cur += W;
    }
  }
  __device__ __forceinline__ void issue_until(int limit) {
    const int lim = limit < npieces ? limit : npieces;
    while (cur < lim) issue_one();
  }
};

// per-lane byte offsets inside a 64-row tile of an LDS image: row[ks] for the ds_read_b128 row fragment of row
// block kb (add kb*2048), tr[d] for the transposed fragment of 32-row k-step s2 (add s2*4096)
struct ResOffsets {
  int row[2], tr[4];
  __device__ __forceinline__ void init(int li, int lg) {
    const int swz_r = ((li >> 1) & 3) << 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) row[ks] = li * 128 + (((ks * 4 + lg) ^ swz_r) << 4);
    const int swz_t = ((2 * lg + (li >> 3)) & 3) << 1;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      tr[d] = (4 * lg + (li >> 2)) * 128 + (((2 * d + ((li & 3) >> 1)) ^ swz_t) << 4) + (li & 1) * 8;
  }
};

// Tile schedule shared by the three kernels: pass 0 of every wave gates on the two arrival barriers and feeds the
// remaining DMA from its first tile; later passes (more 32-row groups than waves) run without any synchronisation.
template <bool MULTI, typename TileFn, typename SyncFn>
__device__ __forceinline__ void res_sweep(int N, bool first, TileFn&& tile, SyncFn&& sync) {
  const int nt = (N + 63) / 64, nfull = N / 64;
  const bool first_pass = !MULTI || first;  // single-pass kernels: known at compile time
  int t0 = 0;
  if (first_pass) {
    sync();  // the first RES_A tiles are visible
    if (nfull > 0) tile(0, std::false_type{}, std::true_type{});
    else tile(0, std::true_type{}, std::true_type{});
    t0 = nt < RES_A ? nt : RES_A;
    for (int t = 1; t < t0; ++t) {
      if (t < nfull) tile(t, std::false_type{}, std::false_type{});
      else tile(t, std::true_type{}, std::false_type{});
    }
    if (nt > RES_A) sync();  // everything is visible
  }
  for (int t = t0; t < nfull; ++t) tile(t, std::false_type{}, std::false_type{});  // the hot loop: one variant only
  if (nfull < nt && t0 <= nfull) tile(nfull, std::true_type{}, std::false_type{});
}

// 4 x 4 transpose of one dword between the register index and the lane group (lanes l, l + 16, l + 32, l + 48):
// in: a[d] of lane group g = X[d][g]; out: a[k] of lane group g = X[g][k]
__device__ __forceinline__ void tr4x4(uint32_t (&a)[4]) {
  // lane groups g <-> g ^ 1: even groups keep X[d0][g] and take X[d0][g + 1], odd groups take X[d1][g - 1] and keep X[d1][g]
  const auto p01 = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false);
  const auto p23 = __builtin_amdgcn_permlane16_swap(a[2], a[3], false, false);
  // lane groups g <-> g ^ 2 on the first / second words of the two pairs
  const auto x = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
  const auto y = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
  a[0] = x[0]; a[1] = y[0]; a[2] = x[1]; a[3] = y[1];
}

// MULTI: more 32-row groups than waves (each wave loops over its groups); otherwise exactly one group per wave.
// QS: the q columns already carry log2(e)/sqrt(dh) (layer path) - the scores leave the MFMA in the log2 domain, and the
// subtraction of the running maximum (forward) / of lse2 (backward) rides in the MFMA's C operand: no per-score FMA.
// MASKED (QS only): the token mask of heads.py:225-232 - keep[b][n] != 0 keeps token n.  A pair (query, key) with either
// token dropped scores -FLT_MAX in the reference: a kept query gives dropped keys zero weight (score -inf here), a dropped
// query's row is one constant, i.e. uniform attention over ALL its keys (score 0 here: lse2 = log2 N exactly, which the
// backward kernel relies on).  The flags of the clip's keys sit in LDS behind the V image.
template <int MAXW, bool MULTI, bool QS, bool MASKED = false>
__global__ __launch_bounds__(MAXW * 64) void attn_fwd_res_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                                float* __restrict__ lse2, int N, int H,
                                                                uint8_t* __restrict__ oq = nullptr,
                                                                uint8_t* __restrict__ osc = nullptr,
                                                                const uint8_t* __restrict__ keep = nullptr) {
  static_assert(!MASKED || QS, "the masked form exists for pre-scaled queries only");
  constexpr int DH = 64, KS = 2, DB = 4;
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = blockDim.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const int NP = (N + 31) & ~31, V = NP >> 5;
  char* ksm = res_smem;
  char* vsm = res_smem + NP * 128;
  const float c = LOG2E / sqrtf((float)DH);
  AVF_PHASE_INIT();
  ResLoader loader;
  loader.init(kbase, ld, ksm, vbase, ld, vsm, NP, N, wave, W, lane);
  ResOffsets off;
  off.init(li, lg);
  const bf16x8_t ones = ones_frag();
  const uint8_t* keepl = reinterpret_cast<const uint8_t*>(res_smem + 2 * NP * 128);  // MASKED: [ceil64(N)] key flags
  if constexpr (MASKED) {
    uint8_t* kw = reinterpret_cast<uint8_t*>(res_smem + 2 * NP * 128);
    for (int i = tid; i < ((N + 63) & ~63); i += blockDim.x) kw[i] = i < N ? keep[(int64_t)b * N + i] : (uint8_t)0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the first arrival barrier of the sweep publishes them)
  }

  int grp = wave;
  do {
    const bool first_pass = !MULTI || grp == wave;
    const int q0 = grp * 32;
    bool qkeep[2] = {true, true};  // MASKED: is this lane's query row a kept token
    bool cen[2] = {false, false};  // MASKED: has the row's running maximum been centred on a finite score yet
    if constexpr (MASKED) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + qb * 16 + li;
        qkeep[qb] = q < N ? keep[(int64_t)b * N + q] != 0 : true;
      }
    }
    bf16x8_t fq[2][KS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int q = q0 + qb * 16 + li;
        fq[qb][ks] = load_frag_global(qbase + (int64_t)q * ld + ks * 32 + 8 * lg, q < N);
      }
    if (first_pass) loader.issue_until(16 * RES_A);

    f32x4_t ot[DB][2], ls[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      ls[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < DB; ++d) ot[d][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    float m[2] = {QS ? 0.f : -INFINITY, QS ? 0.f : -INFINITY};
    f32x4_t minit[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // QS: -m, the C operand of the first score MFMA
    bool first_tile = true;                                              // QS: the first tile centres m on its maximum

    auto tile = [&](int t, auto tail_tag, auto feed_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      constexpr bool FEED = decltype(feed_tag)::value;  // this tile also issues the remaining DMA pieces
      const int nkb = TAIL ? (N - t * 64 + 15) / 16 : 4;
      const char* kt = ksm + t * 8192;
      const char* vt = vsm + t * 8192;
      f32x4_t st[4][2];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        st[kb][0] = QS ? minit[0] : f32x4_t{0.f, 0.f, 0.f, 0.f};
        st[kb][1] = QS ? minit[1] : f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (!TAIL || kb < nkb) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t fk = lds_row_frag(kt + kb * 2048 + off.row[ks]);
            st[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[0][ks], st[kb][0], 0, 0, 0);
            st[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[1][ks], st[kb][1], 0, 0, 0);
          }
        }
        if (FEED) loader.issue_one();
      }
      AVF_PHASE_MARK(2);
      // 8-wave builds (two waves per SIMD, registers to spare): the V fragments of the first k-step are requested BEFORE
      // the softmax and those of the second one before the first k-step's MFMAs, so the transposed LDS reads land under
      // VALU / MFMA work instead of in front of the PV MFMAs (stamps: the PV section was 1100 cycles per tile for 20 MFMAs)
      constexpr bool VPF = MAXW == 8;
      bf16x8_t fvp[2][DB];
      if constexpr (VPF) {
        if (!TAIL || nkb > 0) {
#pragma unroll
          for (int d = 0; d < DB; ++d) fvp[0][d] = lds_tr_frag(vt + off.tr[d]);
        }
      }
      if constexpr (QS) {
        // st = log2-domain score - m.  Re-centre when a row maximum exceeds m by more than RES_TAU (and always on the
        // first tile, which fixes m): only then are the accumulators and this tile's scores shifted.
        float cm[2];
        bool grow = MASKED ? false : first_tile;
        uint32_t kflag[4] = {0u, 0u, 0u, 0u};
        if constexpr (MASKED) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) kflag[kb] = *reinterpret_cast<const uint32_t*>(keepl + t * 64 + kb * 16 + 4 * lg);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          float tmax = -INFINITY;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (MASKED) {
                const bool kk = ((kflag[kb] >> (8 * r)) & 255u) != 0;  // (rows past N carry flag 0)
                const bool real = !TAIL || (t * 64 + kb * 16 + 4 * lg + r < N);
                st[kb][qb][r] = qkeep[qb] ? (kk ? st[kb][qb][r] : -INFINITY) : (real ? -m[qb] : -INFINITY);
              } else {
                if (TAIL && (t * 64 + kb * 16 + 4 * lg + r >= N)) st[kb][qb][r] = -INFINITY;
              }
              tmax = fmaxf(tmax, st[kb][qb][r]);
            }
          cm[qb] = colmax4(tmax);
          if constexpr (MASKED) grow = grow || (cen[qb] ? cm[qb] > RES_TAU : cm[qb] > -1.0e30f);  // (a tile may hold no kept key)
          else grow = grow || (cm[qb] > RES_TAU);
        }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {  // wave-uniform
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
            float shift, alpha;
            if constexpr (MASKED) {
              // a row is centred on its first tile with a finite score (any sign); its accumulators are 0 until then
              const bool fin = cm[qb] > -1.0e30f;
              shift = cen[qb] ? fmaxf(cm[qb], 0.f) : (fin ? cm[qb] : 0.f);
              alpha = cen[qb] ? __builtin_amdgcn_exp2f(-shift) : 1.0f;
              cen[qb] = cen[qb] || fin;
            } else {
              shift = first_tile ? cm[qb] : fmaxf(cm[qb], 0.f);
              alpha = first_tile ? 1.0f : __builtin_amdgcn_exp2f(-shift);  // accumulators are 0 on the first tile
            }
            m[qb] += shift;
            minit[qb] = f32x4_t{-m[qb], -m[qb], -m[qb], -m[qb]};
            ls[qb][0] *= alpha; ls[qb][1] *= alpha; ls[qb][2] *= alpha; ls[qb][3] *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
              ot[d][qb][0] *= alpha; ot[d][qb][1] *= alpha; ot[d][qb][2] *= alpha; ot[d][qb][3] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) st[kb][qb][r] -= shift;
          }
        }
        first_tile = false;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kb][qb][r] = __builtin_amdgcn_exp2f(st[kb][qb][r]);
      } else {
      float cand[2];
      bool grow = false;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (TAIL && (t * 64 + kb * 16 + 4 * lg + r >= N)) st[kb][qb][r] = -INFINITY;
            tmax = fmaxf(tmax, st[kb][qb][r]);
          }
        cand[qb] = colmax4(tmax) * c;
        grow = grow || (cand[qb] > m[qb] + RES_TAU);
      }
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {  // wave-uniform
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const float mn = fmaxf(m[qb], cand[qb]);
          const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
          m[qb] = mn;
          ls[qb][0] *= alpha; ls[qb][1] *= alpha; ls[qb][2] *= alpha; ls[qb][3] *= alpha;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            ot[d][qb][0] *= alpha; ot[d][qb][1] *= alpha; ot[d][qb][2] *= alpha; ot[d][qb][3] *= alpha;
          }
        }
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[kb][qb][r] = __builtin_amdgcn_exp2f(fmaf(st[kb][qb][r], c, -m[qb]));
      }
      AVF_PHASE_MARK(3);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (TAIL && 2 * s2 >= nkb) continue;
        if constexpr (!VPF) __builtin_amdgcn_sched_barrier(0);  // keep the V fragments of this k-step from being hoisted over the softmax
        const bf16x8_t p0 = pack_pair(st[2 * s2][0], st[2 * s2 + 1][0]);
        const bf16x8_t p1 = pack_pair(st[2 * s2][1], st[2 * s2 + 1][1]);
        if constexpr (VPF) {
          if (s2 == 0 && (!TAIL || nkb > 2)) {
#pragma unroll
            for (int d = 0; d < DB; ++d) fvp[1][d] = lds_tr_frag(vt + 4096 + off.tr[d]);
          }
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          bf16x8_t fv;
          if constexpr (VPF) fv = fvp[s2][d];
          else fv = lds_tr_frag(vt + s2 * 4096 + off.tr[d]);
          ot[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, p0, ot[d][0], 0, 0, 0);
          ot[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, p1, ot[d][1], 0, 0, 0);
        }
        ls[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, ls[0], 0, 0, 0);
        ls[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, ls[1], 0, 0, 0);
        if (FEED) loader.issue_one();
      }
      if (FEED) loader.issue_until(1 << 30);
      AVF_PHASE_MARK(4);
    };
    AVF_PHASE_MARK(1);
    res_sweep<MULTI>(N, first_pass, tile, [&] {
      wait_all_loads();
      __builtin_amdgcn_s_barrier();
      AVF_PHASE_MARK(0);
    });

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = q0 + qb * 16 + li;
      const float l = ls[qb][0];
      const float inv = 1.0f / l;
      {
        // lane groups lg, lg + 1 trade the words of column blocks d, d + 1 (v_permlane16_swap, all lanes active): an even
        // group then holds 8 consecutive columns of block d, an odd one 8 of block d + 1 - 16-byte stores
        bf16* orow = o + ((int64_t)b * N + (q < N ? q : N - 1)) * I + h * DH;
#pragma unroll
        for (int d = 0; d < DB; d += 2) {
          const uint32_t a0 = pack_bf16x2(ot[d][qb][0] * inv, ot[d][qb][1] * inv), a1 = pack_bf16x2(ot[d][qb][2] * inv, ot[d][qb][3] * inv);
          const uint32_t b0 = pack_bf16x2(ot[d + 1][qb][0] * inv, ot[d + 1][qb][1] * inv),
                         b1 = pack_bf16x2(ot[d + 1][qb][2] * inv, ot[d + 1][qb][3] * inv);
          const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          if (q < N)
            *reinterpret_cast<uint4*>(orow + ((lg & 1) ? (d + 1) * 16 + 4 * (lg - 1) : d * 16 + 4 * lg)) =
                make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
        if (q < N && lg == 0) lse2[(int64_t)bh * N + q] = m[qb] + log2f(l);
      }
      if (oq) {  // wave-uniform: MX-FP8 image of the output rows (A operand of the out-projection in the fp8 mode); a
                 // 32-block of the head's 64 columns is two 16-column blocks x the 4 lane groups x 4 registers
        // A lane holds columns 16 d + 4 lg .. + 3 of its row for d = 0..3: one dword of the image each, 16 bytes apart.  A
        // 4 x 4 transpose between the register index d and the lane group lg (tr4x4) leaves lane group lg with the 16
        // contiguous bytes of column block d = lg: one 16-byte store (64-byte row segments per instruction; 67.6 -> 65.7 us
        // at B = 64, N = 512 against per-dword stores.  The same transpose on the bf16 output itself measured +2 %: not used)
        const int64_t row = (int64_t)b * N + (q < N ? q : N - 1);
        uint32_t qw[DB], sb[2];
        float vb[DB][4];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) vb[d][r] = to_f32<bf16>(from_f32<bf16>(ot[d][qb][r] * inv));  // image of the STORED values
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          float am = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) am = fmaxf(am, fmaxf(fabsf(vb[2 * blk][r]), fabsf(vb[2 * blk + 1][r])));
          am = fmaxf(am, __shfl_xor(am, 16, 64));
          am = fmaxf(am, __shfl_xor(am, 32, 64));
          float sinv;
          sb[blk] = mx8_scale_byte(am, &sinv);
          qw[2 * blk] = mx8_pack4(vb[2 * blk], sinv);
          qw[2 * blk + 1] = mx8_pack4(vb[2 * blk + 1], sinv);
        }
        tr4x4(qw);  // lane group lg: bytes 16 lg .. 16 lg + 15 of the head's 64
        if (q < N) {
          *reinterpret_cast<uint4*>(oq + row * I + h * DH + lg * 16) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
          if (lg == 0) *reinterpret_cast<uint16_t*>(osc + row * (I >> 5) + h * 2) = (uint16_t)(sb[0] | (sb[1] << 8));
        }
      }
    }
    AVF_PHASE_MARK(7);
  } while (MULTI && (grp += W) < V);
  AVF_PHASE_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// head-resident dQ: K and V images as in the forward; 32 query rows per group.  Also produces
// delta[q] = sum_d O[q,d] dO[q,d] for its rows (from the dO fragments it holds anyway) and writes it for the
// dK/dV kernel that follows on the stream - no separate delta launch on this path.
// ---------------------------------------------------------------------------------------------
// MULTI: more 32-row groups than waves (each wave loops over its groups); otherwise exactly one group per wave.
// QS: the q columns already carry log2(e)/sqrt(dh) (layer path) - the scores leave the MFMA in the log2 domain, and the
// subtraction of the running maximum (forward) / of lse2 (backward) rides in the MFMA's C operand: no per-score FMA.
template <int MAXW, bool MULTI, bool QS>
__global__ __launch_bounds__(MAXW * 64) void attn_dq_res_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                               const bf16* __restrict__ d_o,
                                                               const float* __restrict__ lse2, float* __restrict__ delta,
                                                               float* __restrict__ nlse, bf16* __restrict__ dqkv, int N,
                                                               int H) {
  constexpr int DH = 64, KS = 2, DB = 4;
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = blockDim.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const bf16* gbase = d_o + (int64_t)b * N * I + h * DH;
  const bf16* obase = o + (int64_t)b * N * I + h * DH;
  const int NP = (N + 31) & ~31, V = NP >> 5;
  char* ksm = res_smem;
  char* vsm = res_smem + NP * 128;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = LOG2E * scale;
  ResLoader loader;
  loader.init(kbase, ld, ksm, vbase, ld, vsm, NP, N, wave, W, lane);
  ResOffsets off;
  off.init(li, lg);

  int grp = wave;
  do {
    const bool first_pass = !MULTI || grp == wave;
    const int q0 = grp * 32;
    bf16x8_t fq[2][KS], fg[2][KS];
    float L[2], dl[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = q0 + qb * 16 + li;
      const bool ok = q < N;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        fq[qb][ks] = load_frag_global(qbase + (int64_t)q * ld + ks * 32 + 8 * lg, ok);
        fg[qb][ks] = load_frag_global(gbase + (int64_t)q * I + ks * 32 + 8 * lg, ok);
      }
      L[qb] = ok ? lse2[(int64_t)bh * N + q] : 0.f;
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        part += dot8(load_frag_global(obase + (int64_t)q * I + ks * 32 + 8 * lg, ok), fg[qb][ks]);
      dl[qb] = colsum4(part);
      if (ok && lg == 0) {
        delta[(int64_t)bh * N + q] = -dl[qb];        // NEGATED: the dK/dV kernel feeds it to its dP MFMAs as the C operand
        if (QS) nlse[(int64_t)bh * N + q] = -L[qb];  // likewise for the score MFMAs
      }
    }
    const f32x4_t linit[2] = {{-L[0], -L[0], -L[0], -L[0]}, {-L[1], -L[1], -L[1], -L[1]}};
    // dP - delta leaves the MFMA directly: -delta of the lane's query row rides in the C operand (no per-score subtract)
    const f32x4_t dinit[2] = {{-dl[0], -dl[0], -dl[0], -dl[0]}, {-dl[1], -dl[1], -dl[1], -dl[1]}};
    if (first_pass) loader.issue_until(16 * RES_A);

    f32x4_t dqt[DB][2];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) dqt[d][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto tile = [&](int t, auto tail_tag, auto feed_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      constexpr bool FEED = decltype(feed_tag)::value;
      const int nkb = TAIL ? (N - t * 64 + 15) / 16 : 4;
      const char* kt = ksm + t * 8192;
      const char* vt = vsm + t * 8192;
      f32x4_t ds[4][2];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        ds[kb][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        ds[kb][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (!TAIL || kb < nkb) {
          const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
          f32x4_t p0 = dinit[0], p1 = dinit[1];
          f32x4_t s0 = QS ? linit[0] : zero, s1 = QS ? linit[1] : zero;  // QS: the MFMA returns log2-score - lse2
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t fk = lds_row_frag(kt + kb * 2048 + off.row[ks]);
            const bf16x8_t fv = lds_row_frag(vt + kb * 2048 + off.row[ks]);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[0][ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[1][ks], s1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fg[0][ks], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fg[1][ks], p1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool dead = TAIL && (t * 64 + kb * 16 + 4 * lg + r >= N);
            const float e0 = dead ? 0.f : __builtin_amdgcn_exp2f(QS ? s0[r] : fmaf(s0[r], c, -L[0]));
            const float e1 = dead ? 0.f : __builtin_amdgcn_exp2f(QS ? s1[r] : fmaf(s1[r], c, -L[1]));
            ds[kb][0][r] = e0 * p0[r];
            ds[kb][1][r] = e1 * p1[r];
          }
        }
        if (FEED) loader.issue_one();
      }
      // dQ^T[d][q] += K^T dS^T
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (TAIL && 2 * s2 >= nkb) continue;
        const bf16x8_t a0 = pack_pair(ds[2 * s2][0], ds[2 * s2 + 1][0]);
        const bf16x8_t a1 = pack_pair(ds[2 * s2][1], ds[2 * s2 + 1][1]);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const bf16x8_t fkt = lds_tr_frag(kt + s2 * 4096 + off.tr[d]);
          dqt[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkt, a0, dqt[d][0], 0, 0, 0);
          dqt[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkt, a1, dqt[d][1], 0, 0, 0);
        }
        if (FEED) loader.issue_one();
      }
      if (FEED) loader.issue_until(1 << 30);
    };
    res_sweep<MULTI>(N, first_pass, tile, [&] {
      wait_all_loads();
      __builtin_amdgcn_s_barrier();
    });

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = q0 + qb * 16 + li;
      if (q < N) {
        bf16* out = dqkv + ((int64_t)b * N + q) * ld + h * DH;
store_row_pairs<DB>(out, lg, [&](int d) { return dqt[d][qb] * scale; });
      }
    }
   } while (MULTI && (grp += W) < V);
}

// ---------------------------------------------------------------------------------------------
// head-resident dK, dV: Q and dO images (+ the head's lse2 / delta rows, by 4-byte LDS-DMA) in LDS; 32 key rows per
// group.  Query rows past N are copies of row N-1: their probabilities are zeroed in the tail tile.
// ---------------------------------------------------------------------------------------------
// MULTI: more 32-row groups than waves (each wave loops over its groups); otherwise exactly one group per wave.
// QS: the q columns already carry log2(e)/sqrt(dh) (layer path) - the scores leave the MFMA in the log2 domain, and the
// subtraction of the running maximum (forward) / of lse2 (backward) rides in the MFMA's C operand: no per-score FMA.
template <int MAXW, bool MULTI, bool QS>
__global__ __launch_bounds__(MAXW * 64) void attn_dkv_res_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                                const float* __restrict__ lse2,
                                                                const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                                int N, int H) {
  constexpr int DH = 64, KS = 2, DB = 4;
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = blockDim.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const bf16* gbase = d_o + (int64_t)b * N * I + h * DH;
  const int NP = (N + 31) & ~31, V = NP >> 5;
  const int NP64 = (N + 63) & ~63;
  char* qsm = res_smem;
  char* gsm = res_smem + NP * 128;
  float* Ls = reinterpret_cast<float*>(res_smem + 2 * NP * 128);
  float* Ds = Ls + NP64;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = LOG2E * scale;
  const float kscale = QS ? 1.0f / LOG2E : scale;  // dK = scale dS^T q = dS^T q' / log2(e)
  AVF_PHASE_INIT();
  // QS: `lse2` points at the NEGATED statistics the dQ kernel wrote (they become the C operand of the score MFMAs)
  // the softmax statistics of the head's query rows: 64 floats per DMA piece, lse2 pieces then delta pieces
  for (int j = wave; j < NP64 / 32; j += W) {
    const int blk = j >> 1;
    int row = blk * 64 + lane;
    row = row < N ? row : N - 1;
    const float* src = (j & 1) ? delta : lse2;
    glds4(src + (int64_t)bh * N + row, reinterpret_cast<char*>(((j & 1) ? Ds : Ls) + blk * 64));
  }
  ResLoader loader;
  loader.init(qbase, ld, qsm, gbase, I, gsm, NP, N, wave, W, lane);
  ResOffsets off;
  off.init(li, lg);

  int grp = wave;
  do {
    const bool first_pass = !MULTI || grp == wave;
    const int k0 = grp * 32;
    bf16x8_t fk[2][KS], fv[2][KS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int key = k0 + kb * 16 + li;
        fk[kb][ks] = load_frag_global(kbase + (int64_t)key * ld + ks * 32 + 8 * lg, key < N);
        fv[kb][ks] = load_frag_global(vbase + (int64_t)key * ld + ks * 32 + 8 * lg, key < N);
      }
    if (first_pass) loader.issue_until(16 * RES_A);

    f32x4_t dvt[DB][2], dkt[DB][2];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        dvt[d][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dkt[d][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }

    auto tile = [&](int t, auto tail_tag, auto feed_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      constexpr bool FEED = decltype(feed_tag)::value;
      const int nqb = TAIL ? (N - t * 64 + 15) / 16 : 4;  // 16-query blocks with valid rows
      const char* qt = qsm + t * 8192;
      const char* gt = gsm + t * 8192;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (TAIL && 2 * s2 >= nqb) continue;
        f32x4_t pm[2][2], dsm[2][2];  // P and dS, [q-block of the pair][key-block]
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int qb = 2 * s2 + h2;
          pm[h2][0] = pm[h2][1] = dsm[h2][0] = dsm[h2][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (!TAIL || qb < nqb) {
            const float4 l4 = *reinterpret_cast<const float4*>(Ls + t * 64 + qb * 16 + 4 * lg);
            const float4 d4 = *reinterpret_cast<const float4*>(Ds + t * 64 + qb * 16 + 4 * lg);
            const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
            f32x4_t p0 = {d4.x, d4.y, d4.z, d4.w}, p1 = p0;  // Ds holds -delta (written negated by the dQ kernel)
            f32x4_t s0 = QS ? f32x4_t{l4.x, l4.y, l4.z, l4.w} : zero, s1 = s0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const bf16x8_t fqr = lds_row_frag(qt + qb * 2048 + off.row[ks]);
              const bf16x8_t fgr = lds_row_frag(gt + qb * 2048 + off.row[ks]);
              s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqr, fk[0][ks], s0, 0, 0, 0);
              s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqr, fk[1][ks], s1, 0, 0, 0);
              p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgr, fv[0][ks], p0, 0, 0, 0);
              p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgr, fv[1][ks], p1, 0, 0, 0);
            }
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool dead = TAIL && (t * 64 + qb * 16 + 4 * lg + r >= N);
              const float e0 = dead ? 0.f : __builtin_amdgcn_exp2f(QS ? s0[r] : fmaf(s0[r], c, -lv[r]));
              const float e1 = dead ? 0.f : __builtin_amdgcn_exp2f(QS ? s1[r] : fmaf(s1[r], c, -lv[r]));
              pm[h2][0][r] = e0;
              pm[h2][1][r] = e1;
              dsm[h2][0][r] = e0 * p0[r];
              dsm[h2][1][r] = e1 * p1[r];
            }
          }
          if (FEED) loader.issue_one();
        }
        AVF_PHASE_MARK(2);
        // dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS
        const bf16x8_t pa0 = pack_pair(pm[0][0], pm[1][0]);
        const bf16x8_t pa1 = pack_pair(pm[0][1], pm[1][1]);
        const bf16x8_t da0 = pack_pair(dsm[0][0], dsm[1][0]);
        const bf16x8_t da1 = pack_pair(dsm[0][1], dsm[1][1]);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const bf16x8_t fgt = lds_tr_frag(gt + s2 * 4096 + off.tr[d]);
          const bf16x8_t fqt = lds_tr_frag(qt + s2 * 4096 + off.tr[d]);
          dvt[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgt, pa0, dvt[d][0], 0, 0, 0);
          dvt[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgt, pa1, dvt[d][1], 0, 0, 0);
          dkt[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, da0, dkt[d][0], 0, 0, 0);
          dkt[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, da1, dkt[d][1], 0, 0, 0);
        }
        if (FEED) loader.issue_one();
        AVF_PHASE_MARK(4);
      }
      if (FEED) loader.issue_until(1 << 30);
    };
    AVF_PHASE_MARK(1);
    res_sweep<MULTI>(N, first_pass, tile, [&] {
      wait_all_loads();
      __builtin_amdgcn_s_barrier();
      AVF_PHASE_MARK(0);
    });

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int key = k0 + kb * 16 + li;
      if (key < N) {
        bf16* outk = dqkv + ((int64_t)b * N + key) * ld + I + h * DH;
        bf16* outv = outk + I;
store_row_pairs<DB>(outk, lg, [&](int d) { return dkt[d][kb] * kscale; });
        store_row_pairs<DB>(outv, lg, [&](int d) { return dvt[d][kb]; });
      }
    }
    AVF_PHASE_MARK(7);
  } while (MULTI && (grp += W) < V);
  AVF_PHASE_FLUSH();
}

// =============================================================================================
// backward: dQ  (query on the lane; sweeps key tiles)
// =============================================================================================
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_dq_bf16_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                           const float* __restrict__ lse2,
                                                           const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                           int /*B*/, int N, int H, int qs) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int KLD = DH * 2 + 32;  // K tile: row reads AND transposed reads
  constexpr int VLD = DH * 2 + 32;  // V tile: row reads only (+32 B: conflict-free, PMC-checked)
  constexpr int STAGE = 64 * KLD + 64 * VLD;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int blk, bh;
  block_coords((N + 127) / 128, blk, bh);
  const int b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const bf16* gbase = d_o + (int64_t)b * N * I + h * DH;
  const int q0 = blk * 128 + wave * 32;
  const bool active = __builtin_amdgcn_readfirstlane(q0) < N;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = qs ? 1.0f : LOG2E * scale;

  bf16x8_t fq[2][KS], fg[2][KS];
  float L[2], dl[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 16 + li;
    const bool ok = q < N;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fq[qb][ks] = load_frag_global(qbase + (int64_t)q * ld + ks * 32 + 8 * lg, ok);
      fg[qb][ks] = load_frag_global(gbase + (int64_t)q * I + ks * 32 + 8 * lg, ok);
    }
    L[qb] = ok ? lse2[(int64_t)bh * N + q] : 0.f;
    dl[qb] = ok ? delta[(int64_t)bh * N + q] : 0.f;
  }
  f32x4_t dqt[DB][2];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) dqt[d][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  TileStager<DH, 64> sk, sv;
  const int nt = (N + 63) / 64;
  {
    const int nv = N < 64 ? N : 64;
    sk.issue(kbase, ld, 0, nv, tid);
    sv.issue(vbase, ld, 0, nv, tid);
    sk.template commit<KLD>(smem, tid);
    sv.template commit<VLD>(smem + 64 * KLD, tid);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const int r0 = (t + 1) * 64;
      const int nv = (N - r0) < 64 ? (N - r0) : 64;
      sk.issue(kbase, ld, r0, nv, tid);
      sv.issue(vbase, ld, r0, nv, tid);
    }
    const char* kt = smem + cur * STAGE;
    const char* vt = smem + cur * STAGE + 64 * KLD;
    auto tile_body = [&](auto tail_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      const int nkb = TAIL ? (N - t * 64 + 15) / 16 : 4;
      f32x4_t ds[4][2];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        ds[kb][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        ds[kb][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (TAIL && kb >= nkb) continue;
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t fk = row_frag<KLD>(kt, kb * 16 + li, ks * 32 + 8 * lg);
          const bf16x8_t fv = row_frag<VLD>(vt, kb * 16 + li, ks * 32 + 8 * lg);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[0][ks], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[1][ks], s1, 0, 0, 0);
          p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fg[0][ks], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fg[1][ks], p1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = TAIL && (t * 64 + kb * 16 + 4 * lg + r >= N);
          const float e0 = dead ? 0.f : __builtin_amdgcn_exp2f(s0[r] * c - L[0]);
          const float e1 = dead ? 0.f : __builtin_amdgcn_exp2f(s1[r] * c - L[1]);
          ds[kb][0][r] = e0 * (p0[r] - dl[0]);
          ds[kb][1][r] = e1 * (p1[r] - dl[1]);
        }
      }
      // dQ^T[d][q] += K^T dS^T
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (TAIL && 2 * s >= nkb) continue;
        const bf16x8_t a0 = pack_pair(ds[2 * s][0], ds[2 * s + 1][0]);
        const bf16x8_t a1 = pack_pair(ds[2 * s][1], ds[2 * s + 1][1]);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const bf16x8_t fkt = tr_frag<KLD>((const lds_char*)kt, 32 * s, d * 16, li, lg);
          dqt[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkt, a0, dqt[d][0], 0, 0, 0);
          dqt[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkt, a1, dqt[d][1], 0, 0, 0);
        }
      }
    };
    if (active) {
      if (t * 64 + 64 > N) tile_body(std::true_type{});
      else tile_body(std::false_type{});
    }
    if (t + 1 < nt) {
      sk.template commit<KLD>(smem + (cur ^ 1) * STAGE, tid);
      sv.template commit<VLD>(smem + (cur ^ 1) * STAGE + 64 * KLD, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 16 + li;
    if (q < N) {
      bf16* out = dqkv + ((int64_t)b * N + q) * ld + h * DH;
store_row_pairs<DB>(out, lg, [&](int d) { return dqt[d][qb] * scale; });
    }
  }
}

// =============================================================================================
// backward: dK, dV  (key on the lane; sweeps query tiles)
// =============================================================================================
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_dkv_bf16_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                            const float* __restrict__ lse2,
                                                            const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                            int /*B*/, int N, int H, int qs) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int TLD = DH * 2 + 32;  // Q and dO tiles: row reads AND transposed reads
  constexpr int STAGE = 2 * 64 * TLD + 2 * 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int blk, bh;
  block_coords((N + 127) / 128, blk, bh);
  const int b = bh / H, h = bh - b * H;
  const int I = H * DH;
  const int64_t ld = 3 * (int64_t)I;
  const bf16* qbase = qkv + (int64_t)b * N * ld + h * DH;
  const bf16* kbase = qbase + I;
  const bf16* vbase = qbase + 2 * I;
  const bf16* gbase = d_o + (int64_t)b * N * I + h * DH;
  const int k0 = blk * 128 + wave * 32;
  const bool active = __builtin_amdgcn_readfirstlane(k0) < N;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c = qs ? 1.0f : LOG2E * scale;
  const float kscale = qs ? 1.0f / LOG2E : scale;  // dK = scale dS^T q = dS^T q' / log2(e) when q' = q log2(e) scale

  bf16x8_t fk[2][KS], fv[2][KS];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int key = k0 + kb * 16 + li;
      fk[kb][ks] = load_frag_global(kbase + (int64_t)key * ld + ks * 32 + 8 * lg, key < N);
      fv[kb][ks] = load_frag_global(vbase + (int64_t)key * ld + ks * 32 + 8 * lg, key < N);
    }
  f32x4_t dvt[DB][2], dkt[DB][2];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      dvt[d][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dkt[d][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

  TileStager<DH, 64> sq, sg;
  float rl = 0.f, rd = 0.f;  // staged lse2 / delta (threads 0..63)
  auto issue_stats = [&](int r0) {
    if (tid < 64) {
      const bool ok = r0 + tid < N;
      rl = ok ? lse2[(int64_t)bh * N + r0 + tid] : INFINITY;  // 2^(x - inf) = 0 for padded queries
      rd = ok ? delta[(int64_t)bh * N + r0 + tid] : 0.f;
    }
  };
  auto commit_stats = [&](char* stage) {
    if (tid < 64) {
      reinterpret_cast<float*>(stage + 2 * 64 * TLD)[tid] = rl;
      reinterpret_cast<float*>(stage + 2 * 64 * TLD + 256)[tid] = rd;
    }
  };
  const int nt = (N + 63) / 64;
  {
    const int nv = N < 64 ? N : 64;
    sq.issue(qbase, ld, 0, nv, tid);
    sg.issue(gbase, I, 0, nv, tid);
    issue_stats(0);
    sq.template commit<TLD>(smem, tid);
    sg.template commit<TLD>(smem + 64 * TLD, tid);
    commit_stats(smem);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const int r0 = (t + 1) * 64;
      const int nv = (N - r0) < 64 ? (N - r0) : 64;
      sq.issue(qbase, ld, r0, nv, tid);
      sg.issue(gbase, I, r0, nv, tid);
      issue_stats(r0);
    }
    const char* qt = smem + cur * STAGE;
    const char* gt = qt + 64 * TLD;
    const float* Ls = reinterpret_cast<const float*>(qt + 2 * 64 * TLD);
    const float* Ds = Ls + 64;

    // One Q/dO tile, processed as two 32-query k-steps so that only one pair of P / dS blocks is live at a time
    // (keeps the kernel under 256 registers -> two waves per SIMD).
    auto tile_body = [&](auto tail_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      const int nqb = TAIL ? (N - t * 64 + 15) / 16 : 4;  // 16-query blocks with valid rows
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (TAIL && 2 * s >= nqb) continue;
        f32x4_t pm[2][2], dsm[2][2];  // P and dS, [q-block of the pair][key-block]
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int qb = 2 * s + h2;
          pm[h2][0] = pm[h2][1] = dsm[h2][0] = dsm[h2][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (TAIL && qb >= nqb) continue;
          f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t fqr = row_frag<TLD>(qt, qb * 16 + li, ks * 32 + 8 * lg);
            const bf16x8_t fgr = row_frag<TLD>(gt, qb * 16 + li, ks * 32 + 8 * lg);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqr, fk[0][ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqr, fk[1][ks], s1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgr, fv[0][ks], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgr, fv[1][ks], p1, 0, 0, 0);
          }
          const float4 l4 = *reinterpret_cast<const float4*>(Ls + qb * 16 + 4 * lg);
          const float4 d4 = *reinterpret_cast<const float4*>(Ds + qb * 16 + 4 * lg);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e0 = __builtin_amdgcn_exp2f(s0[r] * c - lv[r]);
            const float e1 = __builtin_amdgcn_exp2f(s1[r] * c - lv[r]);
            pm[h2][0][r] = e0;
            pm[h2][1][r] = e1;
            dsm[h2][0][r] = e0 * (p0[r] - dv[r]);
            dsm[h2][1][r] = e1 * (p1[r] - dv[r]);
          }
        }
        // dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS
        const bf16x8_t pa0 = pack_pair(pm[0][0], pm[1][0]);
        const bf16x8_t pa1 = pack_pair(pm[0][1], pm[1][1]);
        const bf16x8_t da0 = pack_pair(dsm[0][0], dsm[1][0]);
        const bf16x8_t da1 = pack_pair(dsm[0][1], dsm[1][1]);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const bf16x8_t fgt = tr_frag<TLD>((const lds_char*)gt, 32 * s, d * 16, li, lg);
          const bf16x8_t fqt = tr_frag<TLD>((const lds_char*)qt, 32 * s, d * 16, li, lg);
          dvt[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgt, pa0, dvt[d][0], 0, 0, 0);
          dvt[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fgt, pa1, dvt[d][1], 0, 0, 0);
          dkt[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, da0, dkt[d][0], 0, 0, 0);
          dkt[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, da1, dkt[d][1], 0, 0, 0);
        }
      }
    };
    if (active) {
      if (t * 64 + 64 > N) tile_body(std::true_type{});
      else tile_body(std::false_type{});
    }
    if (t + 1 < nt) {
      char* nx = smem + (cur ^ 1) * STAGE;
      sq.template commit<TLD>(nx, tid);
      sg.template commit<TLD>(nx + 64 * TLD, tid);
      commit_stats(nx);
    }
    __syncthreads();
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int key = k0 + kb * 16 + li;
    if (key < N) {
      bf16* outk = dqkv + ((int64_t)b * N + key) * ld + I + h * DH;
      bf16* outv = outk + I;
store_row_pairs<DB>(outk, lg, [&](int d) { return dkt[d][kb] * kscale; });
        store_row_pairs<DB>(outv, lg, [&](int d) { return dvt[d][kb]; });
    }
  }
}

}  // namespace

namespace {
// head-resident kernels: dim_head 64, the head's two operands fit in LDS; AVF_ATTN_RESIDENT=0 forces the streaming ones
bool use_resident(int N, int dh) {
  static const int allow = [] {
    const char* e = tuning_env("AVF_ATTN_RESIDENT");
    return (e && *e) ? atoi(e) : 1;
  }();
  return allow && dh == 64 && N <= RES_MAX_N;
}

// waves per workgroup: V = ceil(N/32) row groups spread over ceil(V/12) passes (multi-pass kernels: at most 8 waves)
bool res_multi(int N) { return ceil_div(N, 32) > 12; }
int res_waves(int N) {
  const int V = (int)ceil_div(N, 32);
  if (V <= 12) return V;
  const int passes = (int)ceil_div(V, 8);
  return (int)ceil_div(V, passes);
}

template <typename K, typename... Args>
int res_launch(const TimingScope* ts, K kernel, const char* name, int blocks, int waves, size_t smem, hipStream_t s, Args... args) {
  // raise the dynamic-LDS limit once per kernel (nine instantiations share this function template per signature)
  // (the table is append-only under a lock: the forward thread and autograd's device thread may both get here)
  static struct { const void* fn; PerDeviceOnce once; } raised[32];
  static int nraised = 0;
  static int lock = 0;
  PerDeviceOnce* once = nullptr;
  while (__atomic_exchange_n(&lock, 1, __ATOMIC_ACQUIRE)) {}
  for (int i = 0; i < nraised; ++i)
    if (raised[i].fn == (const void*)kernel) once = &raised[i].once;
  if (!once && nraised < 32) {
    raised[nraised].fn = (const void*)kernel;
    once = &raised[nraised++].once;
  }
  __atomic_store_n(&lock, 0, __ATOMIC_RELEASE);
  if (!once || once->need()) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    AVF_REQUIRE(e == hipSuccess, "%s: cannot raise dynamic LDS limit: %s", name, hipGetErrorString(e));
    if (once) once->mark();
  }
  AVF_REQUIRE(smem <= 160 * 1024, "%s: %zu bytes of LDS", name, smem);
  launch_in_scope(ts, kernel, dim3(blocks), dim3(waves * 64), (uint32_t)smem, s, args...);
  return check_launch(name);
}
}  // namespace

// log2(e)/sqrt(dh): the factor the layer path folds into the query rows of its bf16 Wqkv image
// (AVF_ATTN_QS=0, a tuning aid, turns the folding off: factor 1 and the kernels that scale the scores themselves)
bool attn_q_prescale_on() {
  static const int on = [] {
    const char* e = tuning_env("AVF_ATTN_QS");
    return (e && *e) ? atoi(e) : 1;
  }();
  return on != 0;
}
float attn_q_prescale(int dh) { return attn_q_prescale_on() ? LOG2E / sqrtf((float)dh) : 1.0f; }

// the head-resident forward kernel can also write the MX-FP8 image of its output
bool attn_fwd_emits_mx8(int N, int dh) { return use_resident(N, dh); }

bool attn_masked_bf16_ok(int N, int dh, bool q_prescaled) {
  static const int on = [] {
    const char* e = tuning_env("AVF_ATTN_MASK_MFMA");  // A/B aid: 0 = every masked call on the fp32-arithmetic kernels
    return (e && *e) ? atoi(e) : 1;
  }();
  return on && q_prescaled && use_resident(N, dh) && N >= 1 && N <= 512;
}

int attn_fwd_bf16(const bf16* qkv, bf16* o, float* lse2, int B, int N, int H, int dh, hipStream_t s, bool q_prescaled,
                  void* mx_q, void* mx_s, const void* keep) {
  AVF_REQUIRE(!keep || (attn_masked_bf16_ok(N, dh, q_prescaled) && !mx_q),
              "attn_fwd_bf16: the token mask runs on the MFMA kernels for dim_head 64, N <= 512, pre-scaled q, no fp8 image");
  AVF_REQUIRE(!mx_q || (mx_s && attn_fwd_emits_mx8(N, dh) && ((uintptr_t)mx_q & 3) == 0),
              "attn_fwd_bf16: the MX-FP8 output image exists on the head-resident kernel only (N=%d dh=%d)", N, dh);
  AVF_REQUIRE(B > 0 && N > 0 && H > 0, "attn_fwd_bf16: bad shape");
  AVF_REQUIRE(ceil_div(N, 128) * B * H < (1LL << 31), "attn_fwd_bf16: grid too large");
  AVF_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 7) == 0, "attn_fwd_bf16: misaligned pointers");
  TimingScope ts(KC_ATTN_FWD, 4.0 * B * H * (double)N * N * dh, 2.0 * 4.0 * B * N * H * dh, s, /*per_kernel=*/true);
  if (shape_log_on())
    shape_log("attn_fwd,attn_fwd,%d,%d,%d,%d,%d,%.0f,%.0f", B * H, B, N, H * dh, -1, 4.0 * B * H * (double)N * N * dh,
              2.0 * 4.0 * B * N * H * dh);
  if (use_resident(N, dh)) {
    const int W = res_waves(N);
    const size_t smem = (size_t)((N + 31) & ~31) * 128 * 2;
#define AVF_FWD_RES(MW, MU, Q, NAME) res_launch(&ts, attn_fwd_res_kernel<MW, MU, Q>, NAME, B * H, W, smem, s, qkv, o, lse2, N, H, \
                                                (uint8_t*)mx_q, (uint8_t*)mx_s, (const uint8_t*)nullptr)
    if (keep) {
      const size_t smem_k = smem + (size_t)((N + 63) & ~63);
#define AVF_FWD_RES_M(MW, MU, NAME) res_launch(&ts, attn_fwd_res_kernel<MW, MU, true, true>, NAME, B * H, W, smem_k, s, qkv, o, lse2, N, \
                                              H, (uint8_t*)nullptr, (uint8_t*)nullptr, (const uint8_t*)keep)
      if (res_multi(N)) return AVF_FWD_RES_M(8, true, "attn_fwd_res<8,multi,qs,mask>");
      if (W <= 8) return AVF_FWD_RES_M(8, false, "attn_fwd_res<8,qs,mask>");
      return AVF_FWD_RES_M(12, false, "attn_fwd_res<12,qs,mask>");
#undef AVF_FWD_RES_M
    }
    if (q_prescaled) {
      if (res_multi(N)) return AVF_FWD_RES(8, true, true, "attn_fwd_res<8,multi,qs>");
      if (W <= 8) return AVF_FWD_RES(8, false, true, "attn_fwd_res<8,qs>");
      return AVF_FWD_RES(12, false, true, "attn_fwd_res<12,qs>");
    }
    if (res_multi(N)) return AVF_FWD_RES(8, true, false, "attn_fwd_res<8,multi>");
    if (W <= 8) return AVF_FWD_RES(8, false, false, "attn_fwd_res<8>");
    return AVF_FWD_RES(12, false, false, "attn_fwd_res<12>");
#undef AVF_FWD_RES
  }
  const unsigned grid = (unsigned)(ceil_div(N, 128) * B * H);
  const int qs = q_prescaled ? 1 : 0;
  if (dh == 64) launch_in_scope(&ts, attn_fwd_bf16_kernel<64>, dim3(grid), dim3(256), 0, s, qkv, o, lse2, B, N, H, qs);
  else if (dh == 32) launch_in_scope(&ts, attn_fwd_bf16_kernel<32>, dim3(grid), dim3(256), 0, s, qkv, o, lse2, B, N, H, qs);
  else AVF_REQUIRE(false, "attention (bf16): unsupported dim_head %d (32 or 64)", dh);
  return check_launch("attn_fwd_bf16_kernel");
}

bool attn_bwd_emits_mx8(int N, int dh, bool q_prescaled) { return attn_bwd_merged_ok(N, dh, q_prescaled); }

int attn_bwd_bf16(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv, float* delta, int B,
                  int N, int H, int dh, hipStream_t s, bool q_prescaled, float* nlse, const void* keep, void* dq_q, void* dq_s) {
  AVF_REQUIRE(!dq_q || (attn_bwd_emits_mx8(N, dh, q_prescaled) && !keep),
              "attn_bwd_bf16: the MX-FP8 image of dqkv exists on the merged kernel only (N=%d dh=%d)", N, dh);
  AVF_REQUIRE(!keep || attn_masked_bf16_ok(N, dh, q_prescaled),
              "attn_bwd_bf16: the token mask runs on the MFMA kernels for dim_head 64, N <= 512, pre-scaled q");
  AVF_REQUIRE(B > 0 && N > 0 && H > 0, "attn_bwd_bf16: bad shape");
  AVF_REQUIRE(ceil_div(N, 128) * B * H < (1LL << 31), "attn_bwd_bf16: grid too large");
  AVF_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)d_o & 15) == 0 && ((uintptr_t)dqkv & 7) == 0,
              "attn_bwd_bf16: misaligned pointers");
  TimingScope ts(KC_ATTN_BWD, 10.0 * B * H * (double)N * N * dh, 2.0 * 8.0 * B * N * H * dh, s, /*per_kernel=*/true);
  if (shape_log_on())
    shape_log("attn_bwd,%s,%d,%d,%d,%d,%d,%.0f,%.0f", attn_bwd_merged_ok(N, dh, q_prescaled) ? "attn_bwd_m4_kernel" : "attn_dq+attn_dkv",
              B * H, B, N, H * dh, -1, 10.0 * B * H * (double)N * N * dh, 2.0 * 8.0 * B * N * H * dh);
  if (keep) return attn_bwd_merged(&ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, keep);  // (the merged kernel at every N <= 512)
  if (attn_bwd_merged_ok(N, dh, q_prescaled)) return attn_bwd_merged(&ts, qkv, o, d_o, lse2, dqkv, B, N, H, s, nullptr, dq_q, dq_s);
  if (use_resident(N, dh)) {  // delta comes out of the dQ kernel
    const int W = res_waves(N);
    const size_t smem = (size_t)((N + 31) & ~31) * 128 * 2, smem_kv = smem + (size_t)((N + 63) & ~63) * 8;
    AVF_REQUIRE(!q_prescaled || nlse, "attn_bwd_bf16: scratch for the negated statistics missing");
#define AVF_BWD_RES(MW, MU, Q, TAG)                                                                                      \
  do {                                                                                                                   \
    AVF_TRY(res_launch(&ts, attn_dq_res_kernel<MW, MU, Q>, "attn_dq_res" TAG, B * H, W, smem, s, qkv, o, d_o, lse2, delta, \
                       nlse, dqkv, N, H));                                                                               \
    return res_launch(&ts, attn_dkv_res_kernel<MW, MU, Q>, "attn_dkv_res" TAG, B * H, W, smem_kv, s, qkv, d_o,           \
                      Q ? (const float*)nlse : lse2, (const float*)delta, dqkv, N, H);                                   \
  } while (0)
    if (q_prescaled) {
      if (res_multi(N)) AVF_BWD_RES(8, true, true, "<8,multi,qs>");
      if (W <= 8) AVF_BWD_RES(8, false, true, "<8,qs>");
      AVF_BWD_RES(12, false, true, "<12,qs>");
    }
    if (res_multi(N)) AVF_BWD_RES(8, true, false, "<8,multi>");
    if (W <= 8) AVF_BWD_RES(8, false, false, "<8>");
    AVF_BWD_RES(12, false, false, "<12>");
#undef AVF_BWD_RES
  }
  AVF_TRY(attn_delta(AVF_BF16, o, d_o, delta, B, N, H, dh, s));
  const unsigned grid = (unsigned)(ceil_div(N, 128) * B * H);
  const int qs = q_prescaled ? 1 : 0;
  if (dh == 64) {
    launch_in_scope(&ts, attn_dq_bf16_kernel<64>, dim3(grid), dim3(256), 0, s, qkv, d_o, lse2, (const float*)delta, dqkv, B, N, H, qs);
    launch_in_scope(&ts, attn_dkv_bf16_kernel<64>, dim3(grid), dim3(256), 0, s, qkv, d_o, lse2, (const float*)delta, dqkv, B, N, H, qs);
  } else if (dh == 32) {
    launch_in_scope(&ts, attn_dq_bf16_kernel<32>, dim3(grid), dim3(256), 0, s, qkv, d_o, lse2, (const float*)delta, dqkv, B, N, H, qs);
    launch_in_scope(&ts, attn_dkv_bf16_kernel<32>, dim3(grid), dim3(256), 0, s, qkv, d_o, lse2, (const float*)delta, dqkv, B, N, H, qs);
  } else {
    AVF_REQUIRE(false, "attention (bf16): unsupported dim_head %d (32 or 64)", dh);
  }
  return check_launch("attn_bwd_bf16 kernels");
}

}  // namespace avf
