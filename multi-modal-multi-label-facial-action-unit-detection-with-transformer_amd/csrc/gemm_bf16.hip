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
#include <stdlib.h>

#include "common.hpp"
#include "gemm_nt.hpp"

namespace avf {

namespace {

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

  nt_epilogue<EPI, CT, 4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, li, lg,
                             p.cs_partial ? (int)blockIdx.y * 2 + wm : -1);
}

// ------------------------------------------------------------------------------------------
// NT kernel, LDS-DMA staged (K % 64 == 0): global_load_lds_dwordx4 writes each tile straight into LDS
// (no staging VGPRs, no ds_write).  The LDS image must be lane-linear per wave-instruction (64 lanes x 16 B =
// 8 tile rows), so the XOR swizzle of nt_off() is applied to the per-lane SOURCE chunk instead:
// lane l fills row 8*j + (l>>3), slot l&7, from source chunk (l&7) ^ (l>>3).
// (64*WM) x (64*WN) block tile, WM*WN wavefronts of 64x64 each; two LDS stages; the next tile's DMA is in
// flight while the current one feeds the MFMAs; one barrier per K-step.  Block ids are remapped so that the
// workgroups dealt to one XCD (ids congruent mod 8) cover a contiguous range of tiles and share A panels in its L2.
// ------------------------------------------------------------------------------------------
// LEAN: 0 = the general epilogue (run-time options), 1 = nt_epilogue_lean (every option fixed at compile time; the host has
// checked nt_lean_ok), 2 = the same with column sums; + 4 = with the epilogue's dropout site
template <int EPI, typename CT, int WM, int WN, int MI, int NI, int NS, int LEAN = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nt_glds_kernel(NtParams p, int tiles_n, int nwg, int wpf) {
  constexpr int WTM = 16 * MI, WTN = 16 * NI;  // per-wave output tile
  constexpr int BMT = WTM * WM, BNT = WTN * WN, NW = WM * WN;
  constexpr int A_BYTES = BMT * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;
  // wave-instructions (8 tile rows each) are dealt to the waves round-robin; a wave's last one may not exist (96-row tiles
  // on 8 waves: 12 instructions), which is fine with two stages - every K-step drains vmcnt to 0
  constexpr int A_TOT = BMT / 8, B_TOT = BNT / 8;
  constexpr int A_INS = (A_TOT + NW - 1) / NW, B_INS = (B_TOT + NW - 1) / NW;
  static_assert(NS == 2 || (A_TOT % NW == 0 && B_TOT % NW == 0), "deeper rings count instructions: even split required");
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;

  // per-lane source pointers (row clamped: rows past the edge re-read the last row and are never stored)
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const bf16* ga[A_INS];
  const bf16* gb[B_INS];
#pragma unroll
  for (int j = 0; j < A_INS; ++j) {
    int r = m0 + (wave + j * NW) * 8 + lrow;
    r = r < p.M ? r : p.M - 1;
    ga[j] = p.A + (int64_t)r * p.lda + lchunk * 8;
  }
#pragma unroll
  for (int j = 0; j < B_INS; ++j) {
    int r = n0 + (wave + j * NW) * 8 + lrow;
    r = r < p.N ? r : p.N - 1;
    gb[j] = p.B + (int64_t)r * p.ldb + lchunk * 8;
  }
  auto stage = [&](int st, int k0) {
    char* sa = dsm + st * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INS; ++j)
      if (A_TOT % NW == 0 || wave + j * NW < A_TOT) glds16(ga[j] + k0, sa + (wave + j * NW) * 1024);
#pragma unroll
    for (int j = 0; j < B_INS; ++j)
      if (B_TOT % NW == 0 || wave + j * NW < B_TOT) glds16(gb[j] + k0, sb + (wave + j * NW) * 1024);
  };

  // Weight warm-up.  In the step a weight image was last touched a whole pass ago: it is in no cache, every workgroup walks K
  // in the same order, and with one stage of look-ahead each K-step then exposes a full HBM round trip for the SAME few lines
  // of B in all eight L2s (tools/diag/fresh_operand.py: 15.2 -> 21.2 us at K = 1024, 20.0 -> 30.2 at K = 1536 with nothing but
  // the 1 - 1.5 MB weight out of cache).  So the workgroups dealt to one XCD (ids congruent mod 8 - a speed assumption only)
  // split B between them and request ALL of it once, up front: one dword per 128-byte line, the value never used.  The
  // loads are PLAIN C++ loads (round 6; rounds 4-5 issued them through inline asm into "+v" registers, whose flight the
  // compiler's wait-count and liveness tracking could not see): hipcc owns the destination registers for the whole flight, the
  // empty asm after the K loop is the values' only use (hipcc puts an s_waitcnt vmcnt(0) in front of it, free there), and the
  // "memory" clobbers of the LDS-DMA statements keep the loads in front of the K loop.  They return in order, so the loop's own
  // counted waits (which count the DMA instructions issued AFTER them) retire them first, as before.
  constexpr int WPF = 2;
  uint32_t wpf_sink[WPF];
#pragma unroll
  for (int i = 0; i < WPF; ++i) wpf_sink[i] = 0u;
  if (wpf) {  // (host: B is dense, ldb == K)
    const int nx = (nwg - (int)(blockIdx.x & 7) + 7) >> 3;  // workgroups of this XCD
    const int lines = p.N * (p.K >> 6);
    const int per = (lines + nx - 1) / nx;
    const int lo = (int)(blockIdx.x >> 3) * per;
    const int hi = (lo + per) < lines ? (lo + per) : lines;
#pragma unroll
    for (int i = 0; i < WPF; ++i) {
      const int ln = lo + tid + i * NW * 64;
      if (ln < hi) {
        const bf16* src = p.B + (int64_t)ln * 64;
        wpf_sink[i] = *reinterpret_cast<const uint32_t*>(src);
      }
    }
  }
  asm volatile("" ::: "memory");  // nothing of the warm-up moves below this point

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // LDS byte addresses of this lane's fragment rows in stage 0: row blocks i / j are 16 rows = 2048 bytes apart, and the
  // swizzle term of nt_off() depends on (row & 7) = (li & 7) only, so one base per k-half serves all of them
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  uint32_t abase[2], bbase[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    abase[ks] = lds0 + nt_off(wm * WTM + li, ks * 4 + lg);
    bbase[ks] = lds0 + A_BYTES + nt_off(wn * WTN + li, ks * 4 + lg);
  }

  // NS-stage ring: tiles t .. t+NS-2 are in flight while tile t is consumed.  Each wave counts its own DMA
  // instructions (INS per tile) with s_waitcnt vmcnt(N) - never draining to 0 inside the loop - and one raw
  // s_barrier per K-step makes every wave's landed tile visible and frees the slot of tile t-1 for tile t+NS-1.
  constexpr int INS = A_INS + B_INS;
  const int nt = p.K / TK;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nt) stage(i, i * TK);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const int ahead = (nt - 1 - t) < (NS - 2) ? (nt - 1 - t) : (NS - 2);  // tiles issued after tile t
    if constexpr (NS <= 4) {
      if (ahead >= 2) wait_vmcnt<2 * INS>();
      else if (ahead == 1) wait_vmcnt<INS>();
      else wait_vmcnt<0>();
    } else {  // deep rings (the small-M tile: the whole K panel of a short problem in flight)
      switch (ahead) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<INS>(); break;
        case 2: wait_vmcnt<2 * INS>(); break;
        case 3: wait_vmcnt<3 * INS>(); break;
        case 4: wait_vmcnt<4 * INS>(); break;
        case 5: wait_vmcnt<5 * INS>(); break;
        default: wait_vmcnt<6 * INS>(); break;
      }
    }
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < nt) {
      int slot = cur + NS - 1;
      slot = slot >= NS ? slot - NS : slot;
      stage(slot, (t + NS - 1) * TK);
    }
    // all fragment reads of the K-step are issued (inline asm: gemm_nt.hpp) before its first MFMA; the first k-half's
    // MFMAs start when its MI + NI reads have returned, the second half's reads overlap them
    const uint32_t so = (uint32_t)cur * STAGE;
    bf16x8_t fa[2][MI], fb[2][NI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lds_read_frags<bf16x8_t, 2048>(fa[ks], abase[ks] + so, std::make_integer_sequence<int, MI>{});
      lds_read_frags<bf16x8_t, 2048>(fb[ks], bbase[ks] + so, std::make_integer_sequence<int, NI>{});
    }
    wait_lgkmcnt<MI + NI>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
    cur = (cur + 1 == NS) ? 0 : cur + 1;
  }
#pragma unroll
  for (int i = 0; i < WPF; ++i) asm volatile("" ::"v"(wpf_sink[i]));  // (end of the warm-up registers' live range)

  if constexpr (LEAN != 0)
    nt_epilogue_lean<EPI, CT, MI, NI, (LEAN & 3) == 2 ? 1 : 0, false, false, true, (LEAN & 4) != 0>(
        p, acc, m0 + wm * WTM, n0 + wn * WTN, li, lg, (wg / tiles_n) * WM + wm, nullptr, nullptr,
        (LEAN & 4) ? drop_key(p.drop) : 0);
  else
    nt_epilogue<EPI, CT, MI, NI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, li, lg, p.cs_partial ? (wg / tiles_n) * WM + wm : -1);
}

template <int EPI, typename CT, int WM, int WN, int MI, int NI, int NS, int LEAN = 0>
int launch_nt_glds(const NtParams& p, hipStream_t s, int* part_rows, TimingScope* ts) {
  constexpr int BMT = 16 * MI * WM, BNT = 16 * NI * WN;
  constexpr int SMEM = NS * (BMT + BNT) * 128;
  static_assert(NS >= 2 && NS <= 8 && SMEM <= 160 * 1024, "stage count / LDS budget");
  static PerDeviceOnce raised;
  if (SMEM > 64 * 1024 && raised.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_glds_kernel<EPI, CT, WM, WN, MI, NI, NS, LEAN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    AVF_REQUIRE(e == hipSuccess, "gemm_bf16_nt: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    raised.mark();
  }
  const int tiles_m = (p.M + BMT - 1) / BMT, tiles_n = (p.N + BNT - 1) / BNT;
  const int nwg = tiles_m * tiles_n;
  *part_rows = tiles_m * WM;
  if (shape_log_on()) {
    const double csz = sizeof(CT);
    const double epi_b = (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) ? csz * p.M * p.N : 0.0;
    shape_log("gemm_bf16_nt,gemm_bf16_nt_glds_kernel<%d, %s, %d, %d, %d, %d, %d, %d>,%d,%d,%d,%d,%d,%.0f,%.0f", EPI,
              sizeof(CT) == 4 ? "float" : "bf16", WM, WN, MI, NI, NS, LEAN, nwg, p.M, p.N, p.K,
              EPI, 2.0 * p.M * p.N * p.K,
              2.0 * ((double)p.M * p.K + (double)p.N * p.K) + csz * p.M * p.N + epi_b);
  }
  // Weight warm-up (see the kernel): on for launches that leave workgroup slots of the chip empty (C2: 432 workgroups on 512
  // slots).  It turns stall time into MFMA-dense time, and the replayed step runs against the package power limit (amd-smi: PPT
  // violation active during replay, shader clock 2.15 - 2.37 GHz of 2.4): on the full launches of C3 the tiled shapes ran 6 -
  // 12 % shorter and the firmware lowered the clock of the WHOLE step by 6 % (2305 -> 2155 MHz) - C3 +3.0 %, C4 +0.5 %,
  // C5 +0.3 % in wall time, C2 -1.9 ... -3.8 % (profiles/r05_weight_warmup.txt).
  static const int wpf_env = [] {
    const char* e = tuning_env("AVF_NT_WPF");  // A/B aid: 0 = never, 2 = every launch
    return (e && *e) ? atoi(e) : 1;
  }();
  // the 8-wave tiles only: on the small-M tile (four workgroups of 8 K-steps per CU, the reference's 17-token layers) the warm-up
  // loads queue in front of a short K loop's own - TFormer 133 -> 140 us per layer
  constexpr int kSlots = (WM * WN == 8) ? 512 : 0;  // two 8-wave workgroups per CU, 256 CUs
  const int wpf = (wpf_env && (wpf_env == 2 || nwg < kSlots) && p.ldb == p.K && p.K >= TK) ? 1 : 0;  // (K >= TK: the K loop's
                                                                                                       //  first wait retires the loads)
  launch_in_scope(ts, gemm_bf16_nt_glds_kernel<EPI, CT, WM, WN, MI, NI, NS, LEAN>, dim3(nwg), dim3(WM * WN * 64), SMEM, s, p,
                  tiles_n, nwg, wpf);
  return 0;
}

template <int EPI, typename CT>
int launch_nt_glds_any(const NtParams& p, hipStream_t s, int* part_rows, TimingScope* ts) {
  const int tile = pick_nt_tile_bf16(p.M, p.N, p.K);
  // the two 8-wave tiles with the lean epilogue when nothing asks for the general one's options
  if ((tile == 2 || tile == 5) && nt_lean_ok<EPI, CT>(p, 128, false, /*drop_ok=*/true)) {
    constexpr bool csv = EPI == AVF_EPI_DGELU;  // column sums ride on the dGELU epilogue only
    if constexpr (EPI != AVF_EPI_NONE) {
      if (p.drop.thresh16) {  // the reference's real instantiations train at p = 0.2 (heads.py:277): same lean path + the mask
        if (p.cs_partial == nullptr) {
          if (tile == 5) return launch_nt_glds<EPI, CT, 2, 4, 3, 2, 2, 5>(p, s, part_rows, ts);
          return launch_nt_glds<EPI, CT, 2, 4, 4, 2, 2, 5>(p, s, part_rows, ts);
        }
        if constexpr (csv) {
          if (tile == 5) return launch_nt_glds<EPI, CT, 2, 4, 3, 2, 2, 6>(p, s, part_rows, ts);
          return launch_nt_glds<EPI, CT, 2, 4, 4, 2, 2, 6>(p, s, part_rows, ts);
        }
      }
    }
    if (!p.drop.thresh16) {
      if (p.cs_partial == nullptr) {
        if (tile == 5) return launch_nt_glds<EPI, CT, 2, 4, 3, 2, 2, 1>(p, s, part_rows, ts);
        return launch_nt_glds<EPI, CT, 2, 4, 4, 2, 2, 1>(p, s, part_rows, ts);
      }
      if constexpr (csv) {
        if (tile == 5) return launch_nt_glds<EPI, CT, 2, 4, 3, 2, 2, 2>(p, s, part_rows, ts);
        return launch_nt_glds<EPI, CT, 2, 4, 4, 2, 2, 2>(p, s, part_rows, ts);
      }
    }
  }
  switch (tile) {
    case 6: return launch_nt_glds<EPI, CT, 1, 4, 2, 1, 3>(p, s, part_rows, ts);  // 36 KiB: four workgroups per CU
    case 0: return launch_nt_glds<EPI, CT, 2, 2, 4, 4, 2>(p, s, part_rows, ts);
    case 1: return launch_nt_glds<EPI, CT, 2, 2, 2, 4, 2>(p, s, part_rows, ts);
    case 3: return launch_nt_glds<EPI, CT, 2, 2, 3, 4, 2>(p, s, part_rows, ts);
    case 5: return launch_nt_glds<EPI, CT, 2, 4, 3, 2, 2>(p, s, part_rows, ts);
    default: return launch_nt_glds<EPI, CT, 2, 4, 4, 2, 2>(p, s, part_rows, ts);
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

// ------------------------------------------------------------------------------------------
// grouped TN kernel, LDS-DMA staged: up to 4 weight-gradient GEMMs that share the reduction axis
// (the B*N token rows of one layer) in ONE launch, so the chip is filled even though each dW has only
// 16..48 output tiles.  K % 64 == 0.  Staged rows are 256 B (128 bf16) with no padding (LDS-DMA images are
// lane-linear); transposed-read bank conflicts are removed by XOR-ing the 16-byte chunk index with
// (row & 7) << 1 on the SOURCE address and on the read address (8 consecutive rows -> 8 distinct 32-byte
// windows of the 256-byte bank row).
// ------------------------------------------------------------------------------------------
struct TnProblem {
  const bf16* A;  // [K, M]
  const bf16* B;  // [K, N]
  float* C;       // [M, N] dense
  float* slabs;   // [S][M][N] (S > 1)
  int lda, ldb, M, N, tiles_n, tile_start;
};
struct TnGroup {
  TnProblem p[4];
  int nprob, K, kchunk, S, total_tiles;
  int xcd_groups;  // 256 x 128 kernel: > 0 = XCD-aware block order (see gemm_bf16_tn_group_big_kernel), 0 = tile-major ids
};

__device__ __forceinline__ bf16x8_t tr_frag_swz(const lds_char* tile, int row_base, int col_base, int li, int lg) {
  const int row = row_base + 4 * lg + (li >> 2);  // row & 7 is the same for the +16 read
  const int chunk = (col_base >> 3) + ((li & 3) >> 1);
  const int off = ((chunk ^ ((row & 7) << 1)) << 4) + ((li & 1) << 3);
  const lds_char* p0 = tile + row * 256 + off;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * 256));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// WNW = wavefronts along N (2 -> 4 waves of 64x64, 4 -> 8 waves of 64x32)
template <int WNW>
__global__ __launch_bounds__(128 * WNW) void gemm_bf16_tn_group_kernel(TnGroup g) {
  constexpr int OPB = TR * 256;  // bytes per operand per stage (64 rows x 256 B) = 16 KiB
  constexpr int NI = 8 / WNW;    // 16-column blocks per wave
  constexpr int INS = 8 / WNW;   // LDS-DMA instructions per wave per operand per stage (16 in total)
  __shared__ __attribute__((aligned(16))) char smem[4 * OPB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int li = lane & 15, lg = lane >> 4;
  // block -> (tile, split); splits of one tile are adjacent ids
  const int tile = blockIdx.x / g.S, split = blockIdx.x - tile * g.S;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < g.nprob && tile >= g.p[i].tile_start) pi = i;
  const TnProblem& P = g.p[pi];
  const int lt = tile - P.tile_start;
  const int m0 = (lt / P.tiles_n) * TB, n0 = (lt % P.tiles_n) * TB;
  const int kbeg = split * g.kchunk;
  const int kend = (kbeg + g.kchunk) < g.K ? (kbeg + g.kchunk) : g.K;

  // LDS-DMA source pointers: wave-instruction j of this wave fills rows 4*(4*wave + j) .. +3 (1 KiB)
  const int lrow = lane >> 4, lslot = lane & 15;
  const bf16* ga[INS];
  const bf16* gb[INS];
#pragma unroll
  for (int j = 0; j < INS; ++j) {
    const int row = (wave * INS + j) * 4 + lrow;
    const int chunk = lslot ^ ((row & 7) << 1);
    const int ca = (m0 + chunk * 8 < P.M) ? m0 + chunk * 8 : 0;  // columns past the edge: any valid address
    const int cb = (n0 + chunk * 8 < P.N) ? n0 + chunk * 8 : 0;
    ga[j] = P.A + (int64_t)(kbeg + row) * P.lda + ca;
    gb[j] = P.B + (int64_t)(kbeg + row) * P.ldb + cb;
  }
  const int64_t astep = (int64_t)TR * P.lda, bstep = (int64_t)TR * P.ldb;
  auto stage = [&](int st, int t) {
    char* sa = smem + st * 2 * OPB;
    char* sb = sa + OPB;
#pragma unroll
    for (int j = 0; j < INS; ++j) glds16(ga[j] + t * astep, sa + (wave * INS + j) * 1024);
#pragma unroll
    for (int j = 0; j < INS; ++j) glds16(gb[j] + t * bstep, sb + (wave * INS + j) * 1024);
  };

  f32x4_t acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = kend > kbeg ? (kend - kbeg) / TR : 0;
  if (nt > 0) stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    __syncthreads();
    if (t + 1 < nt) stage(cur ^ 1, t + 1);
    const lds_char* sa = (const lds_char*)(smem + cur * 2 * OPB);
    const lds_char* sb = sa + OPB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[4], fb[NI];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = tr_frag_swz(sa, ks * 32, wm * 64 + i * 16, li, lg);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = tr_frag_swz(sb, ks * 32, wn * (16 * NI) + j * 16, li, lg);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }
  float* out = g.S > 1 ? P.slabs + (int64_t)split * P.M * P.N : P.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= P.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * (16 * NI) + j * 16 + 4 * lg;
      if (n >= P.N) continue;
      *reinterpret_cast<float4*>(out + (int64_t)m * P.N + n) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// grouped TN kernel, 256 x 128 tile: ONE workgroup of 8 wavefronts (4 x 2, 64 x 64 each) per CU.
// The 128 x 128 kernel above moves 32 KB from L2 into LDS per 2.1 MFLOP K-step (64 FLOP/B) and two co-resident
// workgroups saturate the CU's L2 -> LDS path (~38 B/clk: 1690 cycles per pair of K-steps against 1024 of MFMA time);
// this tile moves 48 KB per 4.2 MFLOP (85 FLOP/B).  A rows are 512 B in LDS (two rows per LDS-DMA instruction), B rows
// 256 B (four per instruction); the same chunk XOR (row & 7) << 1 keeps ds_read_b64_tr_b16 conflict-free for both row
// lengths (either is a multiple of the 256-byte bank row).  NS-stage ring (3: 144 KB), counted vmcnt, one raw
// s_barrier per K-step - weight gradients reduce over thousands of token rows, so the ring runs at depth for the whole
// launch and there is no short-K prologue / epilogue problem here.
// ------------------------------------------------------------------------------------------
constexpr int TBM = 256;  // big-tile M extent (columns of A)

template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_frag_swz_rb(const lds_char* tile, int row_base, int col_base, int li, int lg) {
  const int row = row_base + 4 * lg + (li >> 2);  // row & 7 is the same for the +16 read
  const int chunk = (col_base >> 3) + ((li & 3) >> 1);
  const int off = ((chunk ^ ((row & 7) << 1)) << 4) + ((li & 1) << 3);
  const lds_char* p0 = tile + row * ROWB + off;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * ROWB));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// Fragment reads as INLINE ASM.  hipcc treats an LDS-DMA in flight as a pending LDS store that may alias any later
// ds_read of the same array and puts s_waitcnt vmcnt(0) in front of the first read of every K-step - which drains the ring
// to depth one whatever the counted waits in the source say.  The asm is opaque to that pass; ordering against the DMA is
// the kernel's own counted vmcnt + s_barrier, and every use of the results sits behind an explicit s_waitcnt lgkmcnt(0)
// followed by sched_barrier(0) (the compiler may otherwise hoist register-only MFMAs above the asm wait).
template <int OFF>
__device__ __forceinline__ s16x4_t lds_tr16_asm(uint32_t addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ bf16x8_t tr_frag_asm(uint32_t addr) {
  const s16x4_t lo = lds_tr16_asm<OFF_LO>(addr), hi = lds_tr16_asm<OFF_HI>(addr);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int NS>
__global__ __launch_bounds__(512) void gemm_bf16_tn_group_big_kernel(TnGroup g) {
  static_assert(NS == 3, "the ping-pong schedule below is written for a 3-slot ring");
  constexpr int A_BYTES = TR * 512, B_BYTES = TR * 256, STAGE = A_BYTES + B_BYTES;  // 32 + 16 KiB
  constexpr int A_INS = 4, B_INS = 2, INS = A_INS + B_INS;                          // per wave per stage
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  // Block -> (tile, split).  Workgroups are dealt to the 8 XCDs round-robin (ids congruent mod 8 share an L2 - a speed
  // assumption only).  With S splits dividing 8, XCD x takes K-range x % S of the tiles of group x / S, the tile list being
  // cut into 8 / S contiguous groups: an XCD then streams ONE K-range of operand panels that no other group needs (the
  // list is ordered so that the cut falls between problems: dWqkv + dWo | dW1 + dW2), and every operand byte crosses the
  // fabric once.  The tile-major order (tile = id / S) spread the tiles of one panel over two XCD groups: 235 MB per launch
  // at C2 against 161 MB of operands + slabs.
  int tile, split;
  if (g.xcd_groups > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    split = xcd % g.S;
    const int grp = xcd / g.S;
    const int lo = (int)((int64_t)g.total_tiles * grp / g.xcd_groups), hi = (int)((int64_t)g.total_tiles * (grp + 1) / g.xcd_groups);
    tile = lo + j;
    if (tile >= hi) return;  // (whole workgroup: uniform)
  } else if (g.xcd_groups < 0) {
    // any split count (S = 3 at d = 768: round 3 ran that shape tile-major - every XCD fetched every panel, 1.6 GB measured for
    // 0.32 GB of operands and slabs): the S * tiles work items in split-major order, XCD x takes the x-th eighth of them -
    // a contiguous run of tiles of one K-range (two at a seam), so a panel is fetched by one XCD (two at a seam)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int W = g.S * g.total_tiles;
    const int lo = (int)((int64_t)W * xcd / 8), hi = (int)((int64_t)W * (xcd + 1) / 8);
    const int w = lo + j;
    if (w >= hi) return;  // (whole workgroup: uniform)
    split = w / g.total_tiles;
    tile = w - split * g.total_tiles;
  } else {
    tile = blockIdx.x / g.S;
    split = blockIdx.x - tile * g.S;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < g.nprob && tile >= g.p[i].tile_start) pi = i;
  const TnProblem& P = g.p[pi];
  const int lt = tile - P.tile_start;
  const int m0 = (lt / P.tiles_n) * TBM, n0 = (lt % P.tiles_n) * TB;
  const int kbeg = split * g.kchunk;
  const int kend = (kbeg + g.kchunk) < g.K ? (kbeg + g.kchunk) : g.K;

  // LDS-DMA sources.  A: instruction q = 4 wave + j fills rows 2q, 2q+1 (lanes 0..31 / 32..63, 32 chunks of 16 B each);
  // B: instruction q = 2 wave + j fills rows 4q .. 4q+3 (16 lanes, 16 chunks each).
  const bf16* ga[A_INS];
  const bf16* gb[B_INS];
#pragma unroll
  for (int j = 0; j < A_INS; ++j) {
    const int row = (wave * A_INS + j) * 2 + (lane >> 5);
    const int chunk = (lane & 31) ^ ((row & 7) << 1);
    const int ca = (m0 + chunk * 8 < P.M) ? m0 + chunk * 8 : 0;  // columns past the edge: any valid address
    ga[j] = P.A + (int64_t)(kbeg + row) * P.lda + ca;
  }
#pragma unroll
  for (int j = 0; j < B_INS; ++j) {
    const int row = (wave * B_INS + j) * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ ((row & 7) << 1);
    const int cb = (n0 + chunk * 8 < P.N) ? n0 + chunk * 8 : 0;
    gb[j] = P.B + (int64_t)(kbeg + row) * P.ldb + cb;
  }
  const int64_t astep = (int64_t)TR * P.lda, bstep = (int64_t)TR * P.ldb;
  auto stage = [&](int st, int t) {
    char* sa = dsm + st * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INS; ++j) glds16(ga[j] + t * astep, sa + (wave * A_INS + j) * 1024);
#pragma unroll
    for (int j = 0; j < B_INS; ++j) glds16(gb[j] + t * bstep, sb + (wave * B_INS + j) * 1024);
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Ping-pong between the two halves of the workgroup (waves 0-3 / 4-7; a SIMD hosts one wave of each).  Every wave
  // alternates a LOAD phase (fragment reads of tile t, LDS-DMA issue for tile t+2, waits) with a COMPUTE phase (its 32
  // MFMAs), one s_barrier after each; the second half runs one phase behind (one extra barrier up front), so while one
  // wave of a SIMD computes its partner loads.  With all eight waves in lock-step (same code, one barrier per K-step)
  // this tile ran 20 % SLOWER than two independent 128 x 128 workgroups: both waves of a SIMD read, then both compute.
  // Hazards (interval = time between two barriers; half 0 loads tile t in interval 2t+1, half 1 in 2t+2):
  //   RAW  a wave waits for its own pieces of tile t+1 (vmcnt) at the end of its load phase t, i.e. by the end of interval
  //        2t+2 for every wave; the first read of tile t+1 is in interval 2t+3.
  //   WAR  tile t+2 lands in the slot of tile t-1, first written in interval 2t+1; its last reads ended (lgkmcnt(0) before
  //        the barrier) in interval 2t.
  // per-lane fragment offsets inside a stage (row = 4 lg + (li >> 2) (+16, +32, +48 by immediate), swizzled 16-byte chunk)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  uint32_t aoff[4], boff[4];
  {
    const int row = 4 * lg + (li >> 2), sw = (row & 7) << 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ca = ((wm * 64 + i * 16) >> 3) + ((li & 3) >> 1), cb = ((wn * 64 + i * 16) >> 3) + ((li & 3) >> 1);
      aoff[i] = (uint32_t)(row * 512 + ((ca ^ sw) << 4) + ((li & 1) << 3));
      boff[i] = (uint32_t)(row * 256 + ((cb ^ sw) << 4) + ((li & 1) << 3));
    }
  }
  const int nt = kend > kbeg ? (kend - kbeg) / TR : 0;
  const int half = wave >> 2;
  if (nt > 0) stage(0, 0);
  if (nt > 1) stage(1, 1);
  if (nt > 1) wait_vmcnt<INS>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (half == 1) __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const uint32_t sbase = lds0 + (uint32_t)cur * STAGE;
    bf16x8_t fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][i] = tr_frag_asm<0, 16 * 512>(sbase + aoff[i]);
      fa[1][i] = tr_frag_asm<32 * 512, 48 * 512>(sbase + aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fb[0][j] = tr_frag_asm<A_BYTES, A_BYTES + 16 * 256>(sbase + boff[j]);
      fb[1][j] = tr_frag_asm<A_BYTES + 32 * 256, A_BYTES + 48 * 256>(sbase + boff[j]);
    }
    if (t + 2 < nt) {
      int slot = cur + 2;
      slot = slot >= 3 ? slot - 3 : slot;
      stage(slot, t + 2);
      wait_vmcnt<INS>();
    } else {
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur = (cur + 1 == 3) ? 0 : cur + 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();
  float* out = g.S > 1 ? P.slabs + (int64_t)split * P.M * P.N : P.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= P.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (n >= P.N) continue;
      *reinterpret_cast<float4*>(out + (int64_t)m * P.N + n) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// C_i = sum_s slabs_i[s]  for every problem of the group, plus the layer's deferred column folds (one launch):
// blockIdx.y < nslab selects a weight-gradient problem, the remaining rows select a FoldJob.
template <int SS>  // SS > 0: the split count, fully unrolled (all slab loads of an element in flight); 0: runtime g.S
__global__ __launch_bounds__(256) void fold_group_kernel(TnGroup g, int nslab, FoldList fl) {
  if ((int)blockIdx.y >= nslab) {
    __shared__ float4 red[FOLD_RG][FOLD_COLS / 4];
    const FoldJob& job = fl.job[blockIdx.y - nslab];
    const int groups = (job.width + FOLD_COLS - 1) / FOLD_COLS;
    for (int cg = blockIdx.x; cg < groups; cg += gridDim.x) fold_columns_vec(job, cg, red);
    return;
  }
  const TnProblem& P = g.p[blockIdx.y];
  const int64_t n4 = (int64_t)P.M * P.N / 4, slab = (int64_t)P.M * P.N;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n4; i += stride) {
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));  // slabs are read exactly once: non-temporal
    f32x4_nt a;
    if (SS > 0) {
      f32x4_nt v[SS > 0 ? SS : 1];
#pragma unroll
      for (int s = 0; s < SS; ++s) v[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(P.slabs + (int64_t)s * slab) + i);
      a = v[0];
#pragma unroll
      for (int s = 1; s < SS; ++s) a += v[s];  // same order as the runtime loop: bitwise the same sums
    } else {
      a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(P.slabs) + i);
      for (int s = 1; s < g.S; ++s) a += __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(P.slabs + (int64_t)s * slab) + i);
    }
    reinterpret_cast<float4*>(P.C)[i] = make_float4(a[0], a[1], a[2], a[3]);
  }
}

int tn_group_splits(int total_tiles, int64_t K, int slots = 512) {
  static const int forced = [] {
    const char* e = tuning_env("AVF_TN_SPLITS");  // tuning aid
    return (e && *e) ? atoi(e) : 0;
  }();
  if (forced > 0) return forced;
  // 512 workgroup slots (2 per CU at 72 KiB of LDS).  Pick the split count whose total workgroup count fills whole rounds of
  // them best, charging 5 % per extra slab for the fold's traffic: 128 tiles (d = 512) -> 4 splits = exactly one round;
  // 288 tiles (d = 768) -> 3 splits = 1.7 rounds (84 % full) instead of 2 splits = 1.125 rounds (56 %): 3.90 -> 3.13 ms per
  // step at C4 (measured sweep: 3 splits 3.13, 4: 3.22, 5: 3.26, 6: 3.20, 8: 3.17 ms)
  // at least 16 K-steps per workgroup - 4 where the unsplit tiles leave more than half of the slots empty (short inputs: ~1 k
  // token rows; each K-step of a lone workgroup is a full memory round trip, so more, shorter workgroups win: real avformer
  // heads 0.671 -> 0.637 ms per step, TFormer at 17 tokens 0.444 -> 0.437)
  const int64_t per = 2 * (int64_t)total_tiles < slots ? 256 : 1024;
  const int64_t maxs = K / per > 0 ? K / per : 1;
  if (per == 256 && (int64_t)total_tiles * (maxs < 8 ? maxs : 8) <= slots) return (int)(maxs < 8 ? maxs : 8);  // one round even fully split
  int best = 1;
  double best_score = -1.0;
  for (int s = 1; s <= 8 && s <= maxs; ++s) {
    const int64_t wgs = (int64_t)total_tiles * s;
    const double eff = (double)wgs / (double)(ceil_div(wgs, slots) * slots);
    const double score = eff - 0.05 * (s - 1);
    if (score > best_score + 1e-9) { best_score = score; best = s; }
  }
  return best;
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

// one partial row per (tile row, wave row); the finest configuration has 32-row wave tiles, and the last tile may be
// ragged in both its block and its wave rows - size for ceil(M/32) plus a block's worth of slack
size_t gemm_nt_colsum_ws(int64_t M, int64_t N) { return (size_t)(ceil_div(M, 32) + 8) * N * sizeof(float); }

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
  // algorithmic bytes: both operands once, C once, plus what the fused epilogue reads / writes beside C (fp32 residual;
  // the saved pre-activation, in C's type)
  const double csz = a.c_dtype == AVF_F32 ? 4.0 : 2.0;
  const double epi_bytes = a.epilogue == AVF_EPI_BIAS_RES ? csz * a.M * a.N
                           : (a.epilogue == AVF_EPI_BIAS_GELU || a.epilogue == AVF_EPI_DGELU) ? csz * a.M * a.N : 0.0;
  TimingScope ts(KC_GEMM_BF16_NT, 2.0 * a.M * a.N * a.K, 2.0 * (a.M * a.K + a.N * a.K) + csz * a.M * a.N + epi_bytes, s,
                 /*per_kernel=*/true);
  p.A = (const bf16*)a.A; p.lda = a.lda; p.B = (const bf16*)a.B; p.ldb = a.ldb;
  p.C = a.C; p.ldc = a.ldc; p.bias = a.bias; p.residual = a.residual; p.ldres = a.ldres;
  p.aux = a.aux; p.ldaux = a.ldaux;
  p.drop = a.drop;
  p.mxq = nullptr; p.mxs = nullptr;
  p.wide = nt_wide_stores();
  AVF_REQUIRE(!a.drop.thresh16 || a.epilogue != AVF_EPI_NONE, "gemm_bf16_nt: dropout needs a fused epilogue");
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  dim3 grid((unsigned)ceil_div(a.N, TB), (unsigned)ceil_div(a.M, TB));
  AVF_REQUIRE(grid.y < 65536, "gemm_bf16_nt: M too large for grid");
  const bool cf32 = a.c_dtype == AVF_F32;
  AVF_REQUIRE(cf32 || a.c_dtype == AVF_BF16, "gemm_bf16_nt: bad c_dtype");
  AVF_REQUIRE(!a.mx_q || gemm_bf16_nt_ws_ok(a), "gemm_bf16_nt: an MX-FP8 image of C exists on the weight-stationary kernel only "
              "(ask gemm_bf16_nt_ws_ok first)");
  // K = 512 with a fragment-major weight image: the weight-stationary persistent kernel - where it is the faster one, or asked for
  if ((a.ws_force || a.mx_q) ? gemm_bf16_nt_ws_ok(a) : gemm_bf16_nt_ws_preferred(a)) {
    int ws_rows = 0;
    AVF_TRY(gemm_bf16_nt_ws(a, s, &ws_rows));
    if (a.colsum) {
      if (a.defer_fold) *a.defer_fold = FoldJob{(float*)a.workspace, ws_rows, (int)a.N, (int)a.N, a.colsum, nullptr, nullptr};
      else AVF_TRY(fold_partials((float*)a.workspace, ws_rows, (int)a.N, a.colsum, s));
    }
    return 0;
  }
  const bool dma = (a.K % TK == 0);
  int part_rows = (int)grid.y * 2;  // register-staged kernel: 2 wave rows per 128-row tile
  p.cs_partial = nullptr;
  if (a.colsum) {
    AVF_REQUIRE(a.workspace, "gemm_bf16_nt: column-sum workspace missing");
    p.cs_partial = (float*)a.workspace;
    // the partial rows the chosen tile will write must fit the workspace - checked BEFORE anything is enqueued
    int planned = part_rows;
    if (dma) {
      const int t = pick_nt_tile_bf16(a.M, a.N, a.K);
      const int bmt = (t == 0 || t == 2) ? 128 : (t == 1 ? 64 : (t == 6 ? 32 : 96));
      planned = (int)ceil_div(a.M, bmt) * (t == 6 ? 1 : 2);  // two wave rows per block tile (the small-M tile: one)
    }
    AVF_REQUIRE((size_t)planned * a.N * sizeof(float) <= gemm_nt_colsum_ws(a.M, a.N),
                "gemm_bf16_nt: column-sum partials exceed their workspace (internal error)");
  }
#define LAUNCH(E)                                                         \
  do {                                                                    \
    if (dma) {                                                            \
      if (cf32) AVF_TRY((launch_nt_glds_any<E, float>(p, s, &part_rows, &ts))); \
      else AVF_TRY((launch_nt_glds_any<E, bf16>(p, s, &part_rows, &ts)));      \
    } else if (cf32) launch_in_scope(&ts, gemm_bf16_nt_kernel<E, float>, grid, dim3(256), 0, s, p); \
    else launch_in_scope(&ts, gemm_bf16_nt_kernel<E, bf16>, grid, dim3(256), 0, s, p); \
  } while (0)
  switch (a.epilogue) {
    case AVF_EPI_NONE: LAUNCH(AVF_EPI_NONE); break;
    case AVF_EPI_BIAS_RES:
      AVF_REQUIRE(a.residual && a.ldres % 4 == 0, "gemm_bf16_nt: BIAS_RES needs a residual (in C's storage type)");
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
  AVF_TRY(check_launch("gemm_bf16_nt_kernel"));
  if (a.colsum) {
    if (a.defer_fold) *a.defer_fold = FoldJob{p.cs_partial, part_rows, (int)a.N, (int)a.N, a.colsum, nullptr, nullptr};
    else AVF_TRY(fold_partials(p.cs_partial, part_rows, (int)a.N, a.colsum, s));
  }
  return 0;
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
  static PerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM);
    AVF_REQUIRE(e == hipSuccess, "gemm_bf16_tn: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    attr_set.mark();
  }
  const int S = tn_splits(a.M, a.N, a.K);
  AVF_REQUIRE(S == 1 || a.workspace, "gemm_bf16_tn: split-K workspace missing");
  TnParams p;
  TimingScope ts(KC_GEMM_BF16_TN, 2.0 * a.M * a.N * a.K, 2.0 * (a.M * a.K + a.N * a.K) + 4.0 * a.M * a.N, s, /*per_kernel=*/true);
  p.A = (const bf16*)a.A; p.lda = a.lda; p.B = (const bf16*)a.B; p.ldb = a.ldb;
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  p.kchunk = (int)(ceil_div(ceil_div(a.K, S), TR) * TR);
  if (S > 1) { p.C = (float*)a.workspace; p.ldc = a.N; p.slab = a.M * a.N; }
  else { p.C = (float*)a.C; p.ldc = a.ldc; p.slab = 0; }
  dim3 grid((unsigned)ceil_div(a.N, TB), (unsigned)ceil_div(a.M, TB), (unsigned)S);
  launch_in_scope(&ts, gemm_bf16_tn_kernel, grid, dim3(256), TN_SMEM, s, p);
  AVF_TRY(check_launch("gemm_bf16_tn_kernel"));
  if (S > 1) {
    AVF_REQUIRE(a.ldc == a.N, "gemm_bf16_tn: split-K path needs a dense C (ldc == N)");
    const int64_t n4 = a.M * a.N / 4;
    int64_t blocks = ceil_div(n4, 256);
    if (blocks > 2048) blocks = 2048;
    launch_in_scope(&ts, fold_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a.workspace, S, (int64_t)(a.M * a.N),
                    (float*)a.C, n4);
    AVF_TRY(check_launch("fold_slabs_kernel"));
  }
  return 0;
}

bool gemm_bf16_tn_group_ok(const TnGroupArgs& a) {
  if (a.count < 1 || a.count > 4 || a.K <= 0 || a.K % TR != 0) return false;
  for (int i = 0; i < a.count; ++i) {
    if (a.M[i] % 8 || a.N[i] % 8 || a.lda[i] % 8 || a.ldb[i] % 8) return false;
    if (((uintptr_t)a.A[i] & 15) || ((uintptr_t)a.B[i] & 15) || ((uintptr_t)a.C[i] & 15)) return false;
  }
  return true;
}

// 256 x 128 tiles (one 8-wave workgroup per CU) when the group has enough work to fill the chip with them
static bool tn_group_big(const TnGroupArgs& a) {
  static const int forced = [] {
    const char* e = tuning_env("AVF_TN_BIG");  // tuning aid: 0 = always the 128 x 128 kernel, 1 = always the 256 x 128 one
    return (e && *e) ? atoi(e) : -1;
  }();
  if (forced >= 0) return forced != 0;
  int64_t tiles = 0;
  for (int i = 0; i < a.count; ++i) tiles += ceil_div(a.M[i], TBM) * ceil_div(a.N[i], TB);
  // (round 5: also a FEW tiles under a very long reduction - ResFormer's token section, 16 tiles over 50 176 token rows: 921 -> 895 us
  //  per layer with the split count filling the chip)
  return (tiles >= 32 && a.K >= 2048) || (tiles >= 8 && a.K >= 16384);
}
static int tn_group_tiles(const TnGroupArgs& a, bool big) {
  int tiles = 0;
  for (int i = 0; i < a.count; ++i) tiles += (int)(ceil_div(a.M[i], big ? TBM : TB) * ceil_div(a.N[i], TB));
  return tiles;
}

size_t gemm_bf16_tn_group_ws(const TnGroupArgs& a) {
  size_t elems = 0;
  for (int i = 0; i < a.count; ++i) elems += (size_t)a.M[i] * a.N[i];
  // the larger of the two kernels' needs (the choice can be overridden by the environment at run time)
  const int S0 = tn_group_splits(tn_group_tiles(a, false), a.K, 512), S1 = tn_group_splits(tn_group_tiles(a, true), a.K, 256);
  const int S = S0 > S1 ? S0 : S1;
  return S > 1 ? (size_t)S * elems * sizeof(float) : 0;
}

int gemm_bf16_tn_group(const TnGroupArgs& a, hipStream_t s, const FoldList* extra_folds) {
  AVF_REQUIRE(gemm_bf16_tn_group_ok(a), "gemm_bf16_tn_group: unsupported shapes/alignment (K%%64, M%%8, N%%8, 16-byte alignment)");
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.nprob = a.count;
  g.K = (int)a.K;
  int tiles = 0;
  double flops = 0, bytes = 0;
  const bool big = tn_group_big(a);
  for (int i = 0; i < a.count; ++i) {
    TnProblem& P = g.p[i];
    P.A = (const bf16*)a.A[i]; P.B = (const bf16*)a.B[i]; P.C = a.C[i];
    P.lda = (int)a.lda[i]; P.ldb = (int)a.ldb[i]; P.M = (int)a.M[i]; P.N = (int)a.N[i];
    P.tiles_n = (int)ceil_div(a.N[i], TB);
    P.tile_start = tiles;
    tiles += (int)ceil_div(a.M[i], big ? TBM : TB) * P.tiles_n;
    flops += 2.0 * a.M[i] * a.N[i] * a.K;
    bytes += 2.0 * (a.M[i] + a.N[i]) * a.K + 4.0 * a.M[i] * a.N[i];
  }
  g.total_tiles = tiles;
  g.S = tn_group_splits(tiles, a.K, big ? 256 : 512);
  g.kchunk = (int)(ceil_div(ceil_div(a.K, g.S), TR) * TR);
  if (g.S > 1) {
    AVF_REQUIRE(a.workspace, "gemm_bf16_tn_group: split-K workspace missing");
    float* w = (float*)a.workspace;
    for (int i = 0; i < a.count; ++i) {
      g.p[i].slabs = w;
      w += (size_t)g.S * a.M[i] * a.N[i];
    }
  }
  TimingScope ts(KC_GEMM_BF16_TN, flops, bytes, s, /*per_kernel=*/true);
  if (shape_log_on())
    shape_log("gemm_bf16_tn,gemm_bf16_tn_group%s,%d,%d,%d,%lld,%d,%.0f,%.0f", big ? "_big_kernel<3>" : "_kernel", tiles * g.S,
              a.count, g.S, (long long)a.K, -1, flops, bytes);
  static const int tn_waves = [] {
    const char* e = tuning_env("AVF_TN_WAVES");  // tuning aid
    return (e && *e) ? atoi(e) : 4;  // 8 waves measured 5 % slower here (unlike the NT kernel)
  }();
  if (big) {
    static PerDeviceOnce raised3;
    if (raised3.need()) {
      AVF_REQUIRE(hipFuncSetAttribute((const void*)gemm_bf16_tn_group_big_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      3 * 48 * 1024) == hipSuccess, "gemm_bf16_tn_group: cannot raise dynamic LDS limit");
      raised3.mark();
    }
    static const int xcd_order = [] {
      const char* e = tuning_env("AVF_TN_XCD");  // tuning aid: 0 = tile-major block ids
      return (e && *e) ? atoi(e) : 1;
    }();
    int nblocks = tiles * g.S;
    g.xcd_groups = 0;
    if (xcd_order && (g.S == 1 || g.S == 2 || g.S == 4 || g.S == 8)) {
      g.xcd_groups = 8 / g.S;
      int per = 0;  // tiles of the largest group
      for (int grp = 0; grp < g.xcd_groups; ++grp) {
        const int n = (int)((int64_t)tiles * (grp + 1) / g.xcd_groups - (int64_t)tiles * grp / g.xcd_groups);
        per = n > per ? n : per;
      }
      nblocks = 8 * per;
    } else if (xcd_order) {
      g.xcd_groups = -1;
      const int W = tiles * g.S;
      int per = 0;
      for (int x = 0; x < 8; ++x) {
        const int n = (int)((int64_t)W * (x + 1) / 8 - (int64_t)W * x / 8);
        per = n > per ? n : per;
      }
      nblocks = 8 * per;
    }
    launch_in_scope(&ts, gemm_bf16_tn_group_big_kernel<3>, dim3(nblocks), dim3(512), 3 * 48 * 1024, s, g);
  } else if (tn_waves == 4) launch_in_scope(&ts, gemm_bf16_tn_group_kernel<2>, dim3(tiles * g.S), dim3(256), 0, s, g);
  else launch_in_scope(&ts, gemm_bf16_tn_group_kernel<4>, dim3(tiles * g.S), dim3(512), 0, s, g);
  AVF_TRY(check_launch("gemm_bf16_tn_group_kernel"));
  FoldList fl;
  memset(&fl, 0, sizeof(fl));
  if (extra_folds) fl = *extra_folds;
  const int nslab = g.S > 1 ? a.count : 0;
  if (nslab + fl.count > 0) {
    dim3 grid(256, nslab + fl.count);
    switch (nslab ? g.S : 0) {
      case 2: launch_in_scope(&ts, fold_group_kernel<2>, grid, dim3(256), 0, s, g, nslab, fl); break;
      case 3: launch_in_scope(&ts, fold_group_kernel<3>, grid, dim3(256), 0, s, g, nslab, fl); break;
      case 4: launch_in_scope(&ts, fold_group_kernel<4>, grid, dim3(256), 0, s, g, nslab, fl); break;
      case 5: launch_in_scope(&ts, fold_group_kernel<5>, grid, dim3(256), 0, s, g, nslab, fl); break;
      case 6: launch_in_scope(&ts, fold_group_kernel<6>, grid, dim3(256), 0, s, g, nslab, fl); break;
      case 7: launch_in_scope(&ts, fold_group_kernel<7>, grid, dim3(256), 0, s, g, nslab, fl); break;
      case 8: launch_in_scope(&ts, fold_group_kernel<8>, grid, dim3(256), 0, s, g, nslab, fl); break;
      default: launch_in_scope(&ts, fold_group_kernel<0>, grid, dim3(256), 0, s, g, nslab, fl); break;
    }
    AVF_TRY(check_launch("fold_group_kernel"));
  }
  return 0;
}

}  // namespace avf
