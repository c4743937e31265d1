// gemm_nt.hpp - pieces shared by the NT GEMM kernels (gemm_bf16.hip: bf16 operands; gemm_mx8.hip: MX-FP8 operands):
// parameter block, LDS tile addressing, the fused epilogue, LDS-DMA and launch-order helpers.
#pragma once
#include <stdlib.h>

#include <utility>

#include "common.hpp"

namespace avf {
namespace {

typedef __attribute__((address_space(3))) char lds_char;

constexpr int TB = 128;            // block tile (both M and N)
constexpr int TK = 64;             // bf16 K per stage (128 bytes per tile row)
constexpr int NT_STAGE = TB * TK * 2;  // bytes per operand per stage = 16 KiB

struct NtParams {
  const bf16* A;
  int64_t lda;
  const bf16* B;
  int64_t ldb;
  void* C;
  int64_t ldc;
  const float* bias;
  const void* residual;  // fp32, or bf16 when C is bf16 (BIAS_RES on the bf16 residual stream)
  int64_t ldres;
  void* aux;          // saved pre-activation, stored in C's type (BIAS_GELU writes it, DGELU reads it)
  int64_t ldaux;
  float* cs_partial;  // optional [tiles_m * WM][N] column-sum partials of the stored C values (bias gradients)
  DropCfg drop;       // dropout site fused in the epilogue (thresh16 == 0: none); element index = m * N + n
  int wide;           // 1: 2-byte outputs are stored 16 bytes per lane after a lane-pair exchange (store_pair16)
  uint8_t* mxq;       // optional (BIAS_GELU / DGELU, N % 32 == 0): MX-FP8 image of the stored C, [M][N] e4m3 bytes ...
  uint8_t* mxs;       // ... and [M][N/32] E8M0 scale bytes - the A operand of the next GEMM in the fp8 mode
  int M, N, K;
};

// 16-byte chunk c (0..7) of tile row r lives at chunk slot c ^ (r & 7): conflict-free ds_read_b128
__device__ __forceinline__ int nt_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// Two column blocks (j, j+1) of one output row in a 2-byte type.  A lane holds columns 4 lg .. 4 lg + 3 of each block as two
// packed dwords; v_permlane16_swap trades the (j+1) words of the even lane groups for the j words of the odd ones, after
// which a lane owns EIGHT consecutive columns (even groups: block j, columns 4 lg .. 4 lg + 7; odd groups: block j+1,
// columns 4 (lg-1) .. 4 (lg-1) + 7): one 16-byte store where the plain path issues two 8-byte ones to other rows' segments.
template <bool NT = false>
__device__ __forceinline__ void store_pair16(bf16* row, int n_j, int n_j1, int lg, uint32_t a0, uint32_t a1, uint32_t b0,
                                             uint32_t b1, bool row_ok, int N) {
  const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  const int col = (lg & 1) ? n_j1 - 4 : n_j;
  if (row_ok && col < N) {
    if constexpr (NT) {  // written now, read again only in backward: keep it out of the caches the next kernels work from
      typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
      const u32x4_nt v = {s0[0], s1[0], s0[1], s1[1]};
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4_nt*>(row + col));
    } else {
      *reinterpret_cast<uint4*>(row + col) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
  }
}
// the same for one dword per block (4 e4m3 bytes): one 8-byte store
__device__ __forceinline__ void store_pair8(uint8_t* row, int n_j, int n_j1, int lg, uint32_t a, uint32_t b, bool row_ok, int N) {
  const auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const int col = (lg & 1) ? n_j1 - 4 : n_j;
  if (row_ok && col < N) *reinterpret_cast<uint2*>(row + col) = make_uint2(s[0], s[1]);
}

// Operands of the epilogue fetched AHEAD of it (the persistent kernel of gemm_ws.hip issues these loads before a tile's MFMAs
// and consumes them behind them): the bias of the lane's columns and, per row block, the residual / saved pre-activation -
// raw 16-byte pieces of 2-byte rows in the paired layout of nt_epilogue's wide path, or fp32 rows.  nt_epi_prefetch() mirrors
// the address arithmetic of nt_epilogue (template flag PRE there).
template <int MI, int NI>
struct NtPre {
  float4 bj[NI];
  uint4 raw[MI][(NI + 1) / 2];
  float4 exf[MI][NI];
};
// BIAS: also (re)load the bias (a persistent kernel loads it once and passes false afterwards: the per-tile loads are then
// exactly MI * NI / 2 (2-byte rows) or MI * NI (fp32 rows) instructions, which its vmcnt bookkeeping counts on)
template <int EPI, typename CT, int MI, int NI, bool BIAS = true>
__device__ __forceinline__ void nt_epi_prefetch(const NtParams& p, int m_base, int n_base, int li, int lg, NtPre<MI, NI>& pre) {
  int nn[NI], nc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    nn[j] = n_base + j * 16 + 4 * lg;
    nc[j] = nn[j] < p.N ? nn[j] : 0;
    if constexpr (BIAS) pre.bj[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nc[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the caller has checked the conditions of nt_epilogue's wide path (2-byte rows: p.wide == 1, N % 8 == 0, leading dimension
  // % 8 == 0, 16-byte aligned base)
  if constexpr (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_DGELU) {
    if constexpr (sizeof(CT) == 2) {
      static_assert((NI & 1) == 0, "paired column blocks");
      const bf16* src = (EPI == AVF_EPI_BIAS_RES) ? (const bf16*)p.residual : (const bf16*)p.aux;
      const int64_t ldx = (EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int mm = m_base + i * 16 + li;
        const int mc = mm < p.M ? mm : p.M - 1;
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          int cw = (lg & 1) ? nn[j + 1] - 4 : nn[j];
          cw = cw + 8 <= p.N ? cw : 0;
          pre.raw[i][j >> 1] = *reinterpret_cast<const uint4*>(src + (int64_t)mc * ldx + cw);
        }
      }
    } else {  // fp32 C: fp32 residual / saved pre-activation rows
      const float* src = (EPI == AVF_EPI_BIAS_RES) ? (const float*)p.residual : (const float*)p.aux;
      const int64_t ldx = (EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int mm = m_base + i * 16 + li;
        const int mc = mm < p.M ? mm : p.M - 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) pre.exf[i][j] = *reinterpret_cast<const float4*>(src + (int64_t)mc * ldx + nc[j]);
      }
    }
  }
}

// Shared epilogue.  A lane holds C[m = m_base + 16 i + li][n = n_base + 16 j + 4 lg + 0..3] in acc[i][j].
// All global reads of the epilogue (bias, fp32 residual, saved pre-activation) are issued up front from CLAMPED
// addresses - no branch sits between them, so their latencies overlap instead of serialising - and only the
// stores are predicated on the tile edge.
// part_row >= 0: also emit the column sums of this wave's 64 rows into cs_partial[part_row][n] (plain stores;
// a fold kernel adds the tiles_m*WM partial rows) - fuses the bias gradient "db = sum_rows dY" into the GEMM.
template <int EPI, typename CT, int MI, int NI, bool PRE = false>
__device__ __forceinline__ void nt_epilogue(const NtParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base, int li,
                                            int lg, int part_row, const NtPre<MI, NI>* pre = nullptr) {
  int nn[NI], nc[NI];
  float4 bj[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    nn[j] = n_base + j * 16 + 4 * lg;
    nc[j] = nn[j] < p.N ? nn[j] : 0;
    if constexpr (PRE) bj[j] = pre->bj[j];
    else bj[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nc[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const uint64_t dkey = p.drop.thresh16 ? drop_key(p.drop) : 0;
  // 2-byte outputs: lane pairs trade words so that every store is 16 bytes (wave-uniform conditions)
  const bool wide_c = (NI & 1) == 0 && sizeof(CT) == 2 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && p.wide;
  const bool wide_aux = (NI & 1) == 0 && sizeof(CT) == 2 && EPI == AVF_EPI_BIAS_GELU && (p.N & 7) == 0 && (p.ldaux & 7) == 0 && p.wide;
  float cs[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
#pragma unroll
  for (int half = 0; half < (MI + 1) / 2; ++half) {
    float4 ex[2][NI];  // residual (fp32) or saved pre-activation (bf16 -> fp32) for this pair of row blocks
    int mm[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      if (half * 2 + ii >= MI) continue;  // odd MI: the last pair has one row block
      mm[ii] = m_base + (half * 2 + ii) * 16 + li;
      const int mc = mm[ii] < p.M ? mm[ii] : p.M - 1;
      // 2-byte rows (bf16 residual stream, saved pre-activation): one 16-byte load per lane and block PAIR instead of two
      // 8-byte ones - an even lane group reads 8 consecutive columns of block j (its own 4 and its right neighbour's), an odd
      // one 8 of block j + 1 (its left neighbour's 4 and its own); v_permlane16_swap hands each lane its own columns of both
      // blocks (the mirror image of store_pair16).  Wave-uniform condition.
      const bool wide_ex = (PRE && sizeof(CT) == 2 && (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_DGELU)) ||  // PRE: the caller checked
                           ((NI & 1) == 0 && sizeof(CT) == 2 && (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_DGELU) && p.wide && p.wide != 2 &&  // (AVF_NT_WIDE=2: wide stores only)
                            (p.N & 7) == 0 && (((EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux) & 7) == 0 &&
                            (((uintptr_t)((EPI == AVF_EPI_BIAS_RES) ? p.residual : (const void*)p.aux)) & 15) == 0);
      if (wide_ex) {
        const bf16* src = (EPI == AVF_EPI_BIAS_RES) ? (const bf16*)p.residual : (const bf16*)p.aux;
        const int64_t ldx = (EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux;
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          int cw = (lg & 1) ? nn[j + 1] - 4 : nn[j];
          cw = cw + 8 <= p.N ? cw : 0;
          uint4 raw;
          if constexpr (PRE) raw = pre->raw[half * 2 + ii][j >> 1];
          else raw = *reinterpret_cast<const uint4*>(src + (int64_t)mc * ldx + cw);
          const auto s0 = __builtin_amdgcn_permlane16_swap(raw.x, raw.z, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(raw.y, raw.w, false, false);
          ex[ii][j] = make_float4(__uint_as_float(s0[0] << 16), __uint_as_float(s0[0] & 0xffff0000u),
                                  __uint_as_float(s1[0] << 16), __uint_as_float(s1[0] & 0xffff0000u));
          ex[ii][j + 1] = make_float4(__uint_as_float(s0[1] << 16), __uint_as_float(s0[1] & 0xffff0000u),
                                      __uint_as_float(s1[1] << 16), __uint_as_float(s1[1] & 0xffff0000u));
        }
      } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (EPI == AVF_EPI_BIAS_RES) {
          if (sizeof(CT) == 2) ex[ii][j] = load4<bf16>((const bf16*)p.residual + (int64_t)mc * p.ldres + nc[j]);
          else if constexpr (PRE) ex[ii][j] = pre->exf[half * 2 + ii][j];
          else ex[ii][j] = *reinterpret_cast<const float4*>((const float*)p.residual + (int64_t)mc * p.ldres + nc[j]);
        }
        else if (EPI == AVF_EPI_DGELU) ex[ii][j] = load4<CT>((const CT*)p.aux + (int64_t)mc * p.ldaux + nc[j]);
      }
      }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int i = half * 2 + ii;
      if (i >= MI) continue;
      const bool mok = mm[ii] < p.M;
      float mxv[NI][4];
      uint32_t cw[NI][2], aw[NI][2];  // packed bf16 words of C / aux when a lane pair shares its stores (store_pair16)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        float v[4];
        v[0] = acc[i][j][0] + bj[j].x; v[1] = acc[i][j][1] + bj[j].y; v[2] = acc[i][j][2] + bj[j].z; v[3] = acc[i][j][3] + bj[j].w;
        const bool ok = mok && nn[j] < p.N;
        float4 df = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.drop.thresh16) df = drop_factor4(p.drop, dkey, (uint64_t)mm[ii] * p.N + nn[j]);  // wave-uniform branch
        if (EPI == AVF_EPI_BIAS_RES) {  // x + Dropout(Linear(.))
          v[0] = v[0] * df.x + ex[ii][j].x; v[1] = v[1] * df.y + ex[ii][j].y;
          v[2] = v[2] * df.z + ex[ii][j].z; v[3] = v[3] * df.w + ex[ii][j].w;
        } else if (EPI == AVF_EPI_BIAS_GELU) {  // Dropout(GELU(u)); u is saved unmasked
          if (wide_aux) {
            aw[j][0] = pack_bf16x2(v[0], v[1]); aw[j][1] = pack_bf16x2(v[2], v[3]);
          } else if (ok) {
            store4<CT>((CT*)p.aux + (int64_t)mm[ii] * p.ldaux + nn[j], make_float4(v[0], v[1], v[2], v[3]));
          }
          v[0] = gelu_tanh_fast(v[0]) * df.x; v[1] = gelu_tanh_fast(v[1]) * df.y;
          v[2] = gelu_tanh_fast(v[2]) * df.z; v[3] = gelu_tanh_fast(v[3]) * df.w;
        } else if (EPI == AVF_EPI_DGELU) {  // backward through Dropout then GELU
          v[0] *= df.x * dgelu_tanh_fast(ex[ii][j].x); v[1] *= df.y * dgelu_tanh_fast(ex[ii][j].y);
          v[2] *= df.z * dgelu_tanh_fast(ex[ii][j].z); v[3] *= df.w * dgelu_tanh_fast(ex[ii][j].w);
        }
        if (wide_c) {
          cw[j][0] = pack_bf16x2(v[0], v[1]); cw[j][1] = pack_bf16x2(v[2], v[3]);
        } else if (ok) {
          store4<CT>((CT*)p.C + (int64_t)mm[ii] * p.ldc + nn[j], make_float4(v[0], v[1], v[2], v[3]));
        }
        if (ok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[j][r] += v[r];
        }
        if ((EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) && (NI & 1) == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) mxv[j][r] = v[r];
        }
      }
      if ((NI & 1) == 0) {
        if (wide_c) {
#pragma unroll
          for (int j = 0; j < NI; j += 2)
            store_pair16((bf16*)p.C + (int64_t)mm[ii] * p.ldc, nn[j], nn[j + 1], lg, cw[j][0], cw[j][1], cw[j + 1][0],
                         cw[j + 1][1], mok, p.N);
        }
        if (EPI == AVF_EPI_BIAS_GELU && wide_aux) {
#pragma unroll
          for (int j = 0; j < NI; j += 2)
            // the saved pre-activation is read again in backward only: non-temporal stores keep it out of the caches the
            // next kernels work from (C2 2.078 -> 2.063 ms per step, C3 2.757 -> 2.747, same box; AVF_NT_WIDE=3: plain stores.
            // The same hint on its LOAD in the dGELU epilogue changed nothing; on the folded weight gradients it cost 0.8 %:
            // the optimizer then reads them from HBM instead of the Infinity Cache)
            if (p.wide != 3)
              store_pair16<true>((bf16*)p.aux + (int64_t)mm[ii] * p.ldaux, nn[j], nn[j + 1], lg, aw[j][0], aw[j][1], aw[j + 1][0],
                                 aw[j + 1][1], mok, p.N);
            else
            store_pair16((bf16*)p.aux + (int64_t)mm[ii] * p.ldaux, nn[j], nn[j + 1], lg, aw[j][0], aw[j][1], aw[j + 1][0],
                         aw[j + 1][1], mok, p.N);
        }
      }
      // MX-FP8 image of the row segment: a 32-block is the column blocks (j, j+1) x the 4 lane groups x 4 registers
      if ((EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) && (NI & 1) == 0 && p.mxq) {  // wave-uniform
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          float am = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) am = fmaxf(am, fmaxf(fabsf(mxv[j][r]), fabsf(mxv[j + 1][r])));
          am = fmaxf(am, __shfl_xor(am, 16, 64));
          am = fmaxf(am, __shfl_xor(am, 32, 64));
          float inv;
          const uint32_t sb = mx8_scale_byte(am, &inv);
          store_pair8(p.mxq + (int64_t)mm[ii] * p.N, nn[j], nn[j + 1], lg, mx8_pack4(mxv[j], inv), mx8_pack4(mxv[j + 1], inv),
                      mok, p.N);
          if (lg == 0 && mok && nn[j] < p.N) p.mxs[(int64_t)mm[ii] * (p.N >> 5) + (nn[j] >> 5)] = (uint8_t)sb;
        }
      }
    }
  }
  if (part_row >= 0) {  // wave-uniform
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = cs[j][r];  // sum over the 16 lanes of a DPP row (= the 16 tile rows li): VALU only, no ds_bpermute
        t += AVF_DPP_F32(t, 0xB1);
        t += AVF_DPP_F32(t, 0x4E);
        t += AVF_DPP_F32(t, 0x124);
        t += AVF_DPP_F32(t, 0x128);
        cs[j][r] = t;
      }
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
        if (nn[j] < p.N)
          *reinterpret_cast<float4*>(p.cs_partial + (int64_t)part_row * p.N + nn[j]) =
              make_float4(cs[j][0], cs[j][1], cs[j][2], cs[j][3]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Lean epilogue: the fast path of nt_epilogue with every option fixed at COMPILE time (dropout included: DROP) - the general one decides dropout,
// MX image, column edge, store width and column sums with wave-uniform run-time branches, and its ~550 executed
// instructions per 16 values and lane (2200 cycles of one wave's issue, 3700 beside a partner streaming MFMAs: phase stamps of
// gemm_ws.hip, DESIGN_HISTORY.md section 18) cost more than the 64 MFMAs that produce those values.  Same arithmetic, same order, same
// bits.  Preconditions, checked by the host (nt_lean_ok): no MX image beside a dropout site; the
// wave tile lies inside N; 2-byte C: N % 8 == 0, every leading dimension % 8 == 0, 16-byte aligned bases (16-byte paired
// stores and loads, p.wide == 1).  CS: column sums of the stored values into cs_partial[part_row].  PRE: bias / residual /
// saved pre-activation were fetched by nt_epi_prefetch.  FULL: every row of the tile is inside M (no row predicate).
// ------------------------------------------------------------------------------------------
// the column sums of a wave tile, summed over the lanes that share a column (the 16 lanes of a DPP row) and stored as one
// partial row: the same DPP tree as nt_epilogue
template <int NI>
__device__ __forceinline__ void nt_cs_flush(const NtParams& p, float (&cs)[NI][4], int part_row, int n_base, int li, int lg) {
  const int n0 = n_base + 4 * lg;
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = cs[j][r];
      t += AVF_DPP_F32(t, 0xB1);
      t += AVF_DPP_F32(t, 0x4E);
      t += AVF_DPP_F32(t, 0x124);
      t += AVF_DPP_F32(t, 0x128);
      cs[j][r] = t;
    }
  if (li == 0) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
      *reinterpret_cast<float4*>(p.cs_partial + (int64_t)part_row * p.N + n0 + 16 * j) = make_float4(cs[j][0], cs[j][1], cs[j][2], cs[j][3]);
  }
}
// CS: 0 = no column sums; 1 = summed and stored per call (row part_row of cs_partial); 2 = added into the caller's per-lane
// accumulators cs_acc (a persistent kernel sums every tile it owns in registers and calls nt_cs_flush once)
// MXO: also emit the MX-FP8 image of the values as computed (fp32, before their rounding to CT) into p.mxq / p.mxs - a 32-block
// is the column-block pair (j, j + 1) x the four lane groups x 4 registers; the block maximum crosses the lane groups on the
// VALU (v_permlane32_swap / v_permlane16_swap) where nt_epilogue takes two LDS round trips (__shfl_xor)
// BIAS = false: the caller vouches for p.bias == nullptr (no bias registers, no adds of zeros)
// DROP: the dropout site of the epilogue (p.drop; nn.Dropout of heads.py:194,196,216) - the mask factors of nt_epilogue, same
// hash of the same element index, applied at the same place of the same expression (bit-identical results): BIAS_RES masks
// (acc + bias) before the residual, BIAS_GELU masks gelu(u) (u is saved unmasked), DGELU multiplies by mask * gelu'(u).  One
// splitmix64 per 4 values (~40 integer instructions), where the general epilogue pays its 550 for everything else as well.
// dkey = drop_key(p.drop), computed once per kernel by the caller.
template <int EPI, typename CT, int MI, int NI, int CS, bool PRE, bool FULL, bool MXO = false, bool BIAS = true, bool DROP = false>
__device__ __forceinline__ void nt_epilogue_lean_body(const NtParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base, int li,
                                                      int lg, int part_row, const NtPre<MI, NI>* pre, float (*cs_acc)[4],
                                                      uint64_t dkey = 0) {
  static_assert(!DROP || EPI != AVF_EPI_NONE, "dropout sits in a fused epilogue");
  static_assert(sizeof(CT) == 4 || (NI & 1) == 0, "2-byte outputs are stored in column-block pairs");
  const int n0 = n_base + 4 * lg;                    // this lane's 4 columns of block 0 (block j: + 16 j)
  const int cp0 = n0 + ((lg & 1) ? 12 : 0);          // its 8 columns of the block pair (0, 1) after the lane-pair exchange
  float4 bj[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    if constexpr (!BIAS) bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    else if constexpr (PRE) bj[j] = pre->bj[j];
    else bj[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float cs[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m_base + 16 * i + li;
    const bool mok = FULL || m < p.M;
    const int mc = mok ? m : p.M - 1;
    // addresses as (wave-uniform tile offset) + (per-lane offset that does not depend on the tile): a persistent kernel's tile
    // loop then keeps the per-lane part in a register and the 64-bit arithmetic on the scalar unit - every vector instruction
    // of the epilogue competes with the MFMA stream for the SIMD's issue (DESIGN_HISTORY.md section 19)
    const uint32_t lrow = (uint32_t)(16 * i + li);
    float4 ex[NI];
    if constexpr (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_DGELU) {
      if constexpr (sizeof(CT) == 2) {
        const bf16* src = (EPI == AVF_EPI_BIAS_RES) ? (const bf16*)p.residual : (const bf16*)p.aux;
        const int64_t ldx = (EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux;
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          uint4 raw;
          if constexpr (PRE) raw = pre->raw[i][j >> 1];
          else if constexpr (FULL) raw = *reinterpret_cast<const uint4*>(src + (int64_t)m_base * ldx + (uint32_t)(lrow * (uint32_t)ldx + (uint32_t)(cp0 + 16 * j)));
          else raw = *reinterpret_cast<const uint4*>(src + (int64_t)mc * ldx + cp0 + 16 * j);
          const auto s0 = __builtin_amdgcn_permlane16_swap(raw.x, raw.z, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(raw.y, raw.w, false, false);
          ex[j] = make_float4(__uint_as_float(s0[0] << 16), __uint_as_float(s0[0] & 0xffff0000u), __uint_as_float(s1[0] << 16),
                              __uint_as_float(s1[0] & 0xffff0000u));
          ex[j + 1] = make_float4(__uint_as_float(s0[1] << 16), __uint_as_float(s0[1] & 0xffff0000u), __uint_as_float(s1[1] << 16),
                                  __uint_as_float(s1[1] & 0xffff0000u));
        }
      } else {  // fp32 C: fp32 residual / saved pre-activation rows
        const float* src = (EPI == AVF_EPI_BIAS_RES) ? (const float*)p.residual : (const float*)p.aux;
        const int64_t ldx = (EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (PRE) ex[j] = pre->exf[i][j];
          else ex[j] = *reinterpret_cast<const float4*>(src + (int64_t)mc * ldx + n0 + 16 * j);
        }
      }
    }
    uint32_t cw[NI][2], aw[NI][2];
    float4 vf[NI];
    float mxv[MXO ? NI : 1][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float v[4];
      if constexpr (BIAS) {
        v[0] = acc[i][j][0] + bj[j].x; v[1] = acc[i][j][1] + bj[j].y; v[2] = acc[i][j][2] + bj[j].z; v[3] = acc[i][j][3] + bj[j].w;
      } else {  // (x + 0.0f is x for every x but -0.0f, which no consumer distinguishes; the product paths pass no bias here)
        v[0] = acc[i][j][0]; v[1] = acc[i][j][1]; v[2] = acc[i][j][2]; v[3] = acc[i][j][3];
      }
      float4 df;
      if constexpr (DROP) df = drop_factor4(p.drop, dkey, (uint64_t)m * p.N + (uint64_t)(n0 + 16 * j));
      if constexpr (EPI == AVF_EPI_BIAS_RES) {
        if constexpr (DROP) {
          v[0] = v[0] * df.x + ex[j].x; v[1] = v[1] * df.y + ex[j].y; v[2] = v[2] * df.z + ex[j].z; v[3] = v[3] * df.w + ex[j].w;
        } else {
          v[0] += ex[j].x; v[1] += ex[j].y; v[2] += ex[j].z; v[3] += ex[j].w;
        }
      } else if constexpr (EPI == AVF_EPI_BIAS_GELU) {
        if constexpr (sizeof(CT) == 2) {
          aw[j][0] = pack_bf16x2(v[0], v[1]); aw[j][1] = pack_bf16x2(v[2], v[3]);
        } else if (mok) {
          store4<CT>((CT*)p.aux + (int64_t)m_base * p.ldaux + (uint32_t)(lrow * (uint32_t)p.ldaux + (uint32_t)(n0 + 16 * j)),
                     make_float4(v[0], v[1], v[2], v[3]));
        }
        v[0] = gelu_tanh_fast(v[0]); v[1] = gelu_tanh_fast(v[1]); v[2] = gelu_tanh_fast(v[2]); v[3] = gelu_tanh_fast(v[3]);
        if constexpr (DROP) {
          v[0] *= df.x; v[1] *= df.y; v[2] *= df.z; v[3] *= df.w;
        }
      } else if constexpr (EPI == AVF_EPI_DGELU) {
        if constexpr (DROP) {
          v[0] *= df.x * dgelu_tanh_fast(ex[j].x); v[1] *= df.y * dgelu_tanh_fast(ex[j].y);
          v[2] *= df.z * dgelu_tanh_fast(ex[j].z); v[3] *= df.w * dgelu_tanh_fast(ex[j].w);
        } else {
          v[0] *= dgelu_tanh_fast(ex[j].x); v[1] *= dgelu_tanh_fast(ex[j].y);
          v[2] *= dgelu_tanh_fast(ex[j].z); v[3] *= dgelu_tanh_fast(ex[j].w);
        }
      }
      if constexpr (sizeof(CT) == 2) {
        cw[j][0] = pack_bf16x2(v[0], v[1]); cw[j][1] = pack_bf16x2(v[2], v[3]);
      } else {
        vf[j] = make_float4(v[0], v[1], v[2], v[3]);
      }
      if constexpr (MXO) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mxv[j][r] = v[r];
      }
      if constexpr (CS == 1) {
        if (mok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[j][r] += v[r];
        }
      } else if constexpr (CS == 2) {
        if (mok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cs_acc[j][r] += v[r];
        }
      }
    }
    if constexpr (sizeof(CT) == 2) {
      // lane pairs trade words (store_pair16): every lane stores 8 consecutive columns of one block
      uint4 sc[NI / 2], sa[NI / 2];
#pragma unroll
      for (int j = 0; j < NI; j += 2) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(cw[j][0], cw[j + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(cw[j][1], cw[j + 1][1], false, false);
        sc[j >> 1] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        if constexpr (EPI == AVF_EPI_BIAS_GELU) {
          const auto t0 = __builtin_amdgcn_permlane16_swap(aw[j][0], aw[j + 1][0], false, false);
          const auto t1 = __builtin_amdgcn_permlane16_swap(aw[j][1], aw[j + 1][1], false, false);
          sa[j >> 1] = make_uint4(t0[0], t1[0], t0[1], t1[1]);
        }
      }
      if (mok) {
        bf16* crow = (bf16*)p.C + (int64_t)m_base * p.ldc + (uint32_t)(lrow * (uint32_t)p.ldc + (uint32_t)cp0);
#pragma unroll
        for (int j = 0; j < NI; j += 2) *reinterpret_cast<uint4*>(crow + 16 * j) = sc[j >> 1];
        if constexpr (EPI == AVF_EPI_BIAS_GELU) {
          // the saved pre-activation is read again in backward only: non-temporal (as nt_epilogue; p.wide == 3: plain)
          typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
          bf16* arow = (bf16*)p.aux + (int64_t)m_base * p.ldaux + (uint32_t)(lrow * (uint32_t)p.ldaux + (uint32_t)cp0);
#pragma unroll
          for (int j = 0; j < NI; j += 2) {
            const u32x4_nt v4 = {sa[j >> 1].x, sa[j >> 1].y, sa[j >> 1].z, sa[j >> 1].w};
            __builtin_nontemporal_store(v4, reinterpret_cast<u32x4_nt*>(arow + 16 * j));
          }
        }
      }
    } else if (mok) {
      float* crow = (float*)p.C + (int64_t)m_base * p.ldc + (uint32_t)(lrow * (uint32_t)p.ldc + (uint32_t)n0);
#pragma unroll
      for (int j = 0; j < NI; ++j) *reinterpret_cast<float4*>(crow + 16 * j) = vf[j];
    }
    if constexpr (MXO) {
      static_assert((NI & 1) == 0, "MX-FP8 image: 32-blocks are column-block pairs");
#pragma unroll
      for (int j = 0; j < NI; j += 2) {
        float am = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) am = fmaxf(am, fmaxf(fabsf(mxv[j][r]), fabsf(mxv[j + 1][r])));
        {  // maximum over the lanes li, li + 16, li + 32, li + 48 (exact: the order does not matter)
          const auto a32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(am), __float_as_uint(am), false, false);
          am = fmaxf(__uint_as_float(a32[0]), __uint_as_float(a32[1]));
          const auto a16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(am), __float_as_uint(am), false, false);
          am = fmaxf(__uint_as_float(a16[0]), __uint_as_float(a16[1]));
        }
        float inv;
        const uint32_t sb = mx8_scale_byte(am, &inv);
        const auto sq = __builtin_amdgcn_permlane16_swap(mx8_pack4(mxv[j], inv), mx8_pack4(mxv[j + 1], inv), false, false);
        if (mok) {
          *reinterpret_cast<uint2*>(p.mxq + (int64_t)m * p.N + cp0 + 16 * j) = make_uint2(sq[0], sq[1]);
          if (lg == 0) p.mxs[(int64_t)m * (p.N >> 5) + ((n_base + 16 * j) >> 5)] = (uint8_t)sb;
        }
      }
    }
  }
  if constexpr (CS == 1) nt_cs_flush<NI>(p, cs, part_row, n_base, li, lg);
}
template <int EPI, typename CT, int MI, int NI, int CS, bool PRE, bool MXO = false, bool BIAS = true, bool DROP = false>
__device__ __forceinline__ void nt_epilogue_lean(const NtParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base, int li, int lg,
                                                 int part_row, const NtPre<MI, NI>* pre = nullptr, float (*cs_acc)[4] = nullptr,
                                                 uint64_t dkey = 0) {
  if (m_base + 16 * MI <= p.M)
    nt_epilogue_lean_body<EPI, CT, MI, NI, CS, PRE, true, MXO, BIAS, DROP>(p, acc, m_base, n_base, li, lg, part_row, pre, cs_acc, dkey);
  else
    nt_epilogue_lean_body<EPI, CT, MI, NI, CS, PRE, false, MXO, BIAS, DROP>(p, acc, m_base, n_base, li, lg, part_row, pre, cs_acc, dkey);
}

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 16, 0, 0);
}

__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// Fragment reads as inline asm.  With an LDS-DMA in flight hipcc (ROCm 7.2) drains vmcnt to 0 in front of the first
// compiler-visible ds_read of a K-step (its wait-count pass takes the DMA for an LDS store that may alias the read), which
// serialises the next tile's DMA with this tile's reads and MFMAs.  These reads are opaque to that pass: the kernel's own
// vmcnt + s_barrier order them against the DMA, and the caller waits with wait_lgkmcnt<N>() + sched_barrier(0) before the
// first use of the destination registers (the hardware does not interlock LDS returns).  DESIGN_HISTORY.md section 12(a).
template <typename V, int OFF>
__device__ __forceinline__ V lds_read_b128(uint32_t addr) {
  V v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ uint32_t lds_read_b32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// dst[i] = the 16 bytes at addr + i * STRIDE, i = 0 .. sizeof...(Is) - 1 (immediate offsets)
template <typename V, int STRIDE, int... Is>
__device__ __forceinline__ void lds_read_frags(V* dst, uint32_t addr, std::integer_sequence<int, Is...>) {
  ((dst[Is] = lds_read_b128<V, Is * STRIDE>(addr)), ...);
}
template <int STRIDE, int... Is>
__device__ __forceinline__ void lds_read_words(uint32_t* dst, uint32_t addr, std::integer_sequence<int, Is...>) {
  ((dst[Is] = lds_read_b32<Is * STRIDE>(addr)), ...);
}

// tile configurations (block tile, wavefronts, LDS stages, resident blocks per CU):
//   0: 128x128, 4 waves of 64x64, 2 stages (64 KiB, 2/CU)
//   1:  64x128, 4 waves of 32x64, 2 stages (48 KiB, 3/CU)   - finer grain for grids that cannot fill the chip
//   2: 128x128, 8 waves of 64x32, 2 stages (64 KiB, 2/CU)   - twice the waves per CU hide the DMA / epilogue
//                                                             latency better (+5..8 % measured at K = 512..1536)
//   3:  96x128, 4 waves of 48x64, 2 stages (56 KiB, 2/CU)   - M = 10368, N = 512: 432 workgroups fill the 512 slots in
//                                                             one round; 8..11 % faster than (1) once K >= 1024
//   5:  96x128, 8 waves of 48x32, 2 stages (56 KiB, 2/CU)   - the same tile with twice the waves (12 DMA instructions per
//                                                             operand dealt round-robin to 8 waves): 0.3..2.2 us faster than
//                                                             (1) / (3) on every N = 512 shape of C2, replaces both there
// (3- and 4-stage rings, 256x128 / 256x256 / 192x128 tiles, 64x64 tiles and a persistent tile loop were all measured slower
//  on this path's shapes - M = 10k..16k, N = 512..1536, K = 512..1536 - and removed: the waves wait ~55 % of their
//  cycles (SQ_WAIT_ANY) on LDS/barrier latency, which more resident waves hide better than deeper DMA rings)
int nt_lean_on() {
  static const int on = [] {
    const char* e = tuning_env("AVF_NT_LEAN");  // A/B aid: 0 = the general epilogue everywhere in the tiled kernels
    return (e && *e) ? atoi(e) : 1;
  }();
  return on;
}
int nt_wide_stores();
// the preconditions of nt_epilogue_lean for a block tile of bn columns (host side)
template <int EPI, typename CT>
bool nt_lean_ok(const NtParams& p, int bn, bool mx_ok = false, bool drop_ok = false) {
  if (!nt_lean_on() || p.N % bn != 0) return false;
  // a dropout site: only where the caller has the DROP instantiation (the bf16 kernels), and only on a fused epilogue
  if (p.drop.thresh16 && !(drop_ok && EPI != AVF_EPI_NONE && !p.mxq)) return false;
  if (p.mxq && !(mx_ok && (EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) && p.mxs && p.N % 32 == 0 && ((uintptr_t)p.mxq & 7) == 0))
    return false;
  if (p.cs_partial && EPI != AVF_EPI_DGELU) return false;
  if (sizeof(CT) == 2) {
    if (nt_wide_stores() != 1 || p.wide != 1 || (p.N & 7) || (p.ldc & 7) || ((uintptr_t)p.C & 15)) return false;
    if (EPI == AVF_EPI_BIAS_RES && ((p.ldres & 7) || ((uintptr_t)p.residual & 15))) return false;
    if ((EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) && ((p.ldaux & 7) || ((uintptr_t)p.aux & 15))) return false;
  } else {
    if ((p.ldc & 3) || ((uintptr_t)p.C & 15)) return false;
    if (EPI == AVF_EPI_BIAS_RES && ((p.ldres & 3) || ((uintptr_t)p.residual & 15))) return false;
    if ((EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) && ((p.ldaux & 3) || ((uintptr_t)p.aux & 15))) return false;
  }
  return true;
}
int nt_wide_stores() {
  static const int on = [] {
    const char* e = tuning_env("AVF_NT_WIDE");  // tuning aid: 0 = the plain per-block stores
    return (e && *e) ? atoi(e) : 1;
  }();
  return on;
}

int pick_nt_tile(int64_t M, int64_t N, int64_t K) {
  static const int override_tile = [] {
    const char* e = tuning_env("AVF_NT_TILE");  // tuning aid: force one configuration
    return (e && *e) ? atoi(e) : -1;
  }();
  if (override_tile >= 0) return override_tile;
  const int64_t wg128 = ceil_div(M, 128) * ceil_div(N, 128);
  if (wg128 >= 512) return 2;
  (void)K;
  return ceil_div(M, 96) * ceil_div(N, 128) <= 512 ? 5 : 1;
}

// bf16 NT kernel only: problems with few token rows (the reference's 12- / 17-token stacks at batch 64: ~1 k rows) put a few
// dozen 96 x 128 tiles on 256 CUs, and with one stage of look-ahead every K-step of such a lone workgroup pays a full memory
// round trip (measured: 10.9 us for 1088 x 512 x 512).  Tile 6 = 32 x 64, four waves, a ring of 12 KiB stages: eight times the
// workgroups.  Round 5: THREE slots (36 KiB, four workgroups per CU) instead of six (72 KiB, two per CU) - the wide shapes of a
// 1088-row layer (816 workgroups at N = 1536) then fit the chip's slots in one round, and co-resident workgroups hide a round trip as
// well as ring depth does: TFormer (17 tokens, d = 512, B = 64) 144 -> 133 us per layer replayed (2 slots 146, 4 slots 134).
int pick_nt_tile_bf16(int64_t M, int64_t N, int64_t K) {
  static const int small_on = [] {
    const char* e = tuning_env("AVF_NT_SMALL_M");  // A/B aid: 0 = the large tiles for every shape
    return (e && *e) ? atoi(e) : 1;
  }();
  static const int forced = [] {
    const char* e = tuning_env("AVF_NT_TILE");
    return (e && *e) ? atoi(e) : -1;
  }();
  if (forced < 0 && small_on && M <= 2048 && ceil_div(M, 96) * ceil_div(N, 128) <= 160) return 6;
  return pick_nt_tile(M, N, K);
}

}  // namespace
}  // namespace avf
