// gemm_ws.hip - weight-stationary persistent NT GEMM for the layer's K = 512 shapes (DESIGN_HISTORY.md section 18).
//
//   C[M, N] = epilogue(A[M, 512] * W[N, 512]^T)      nn.Linear forward / dX of heads.py:191,195,212,215 at dim = 512
//
// One persistent workgroup of 8 wavefronts per CU.  A workgroup owns a PANEL of 256 output columns for its whole life and
// a contiguous range of 32-row tiles of A:
//   * the weight panel lives in REGISTERS: wave w holds W[n0 + 32 w .. + 31][0 .. 511] as its 32 MFMA operand fragments
//     (2 column blocks x 16 k-steps x 4 VGPRs = 128 VGPRs), loaded once from a fragment-major image (pack_ws) with
//     perfectly coalesced 1 KiB wave-instructions.  The register file of a CU (512 KiB) is three times its LDS: the panel
//     costs no LDS bytes and no LDS reads;
//   * only A streams: tiles of 32 rows x 512 k (32 KiB) arrive by LDS-DMA into a ring of NSLOT slots, two tiles ahead; every
//     wave multiplies the WHOLE tile by its 32 columns (one ds_read_b128 per two MFMAs).  L2 -> LDS bytes per FLOP are a
//     quarter of the 128 x 128 tile's (1/256 against 1/64);
//   * ONE s_barrier per tile (64 MFMAs per wave), none inside the K-loop: the K = 512 reduction of a tile is fully unrolled;
//   * two schedules (AVF_WS_INPHASE, chosen by the epilogue).  Plain stores: waves 4..7 (the SIMD partners of waves 0..3)
//     run half a period out of phase - they take the tile barrier between their MFMAs and their epilogue, waves 0..3 take
//     it before their MFMAs - so on every SIMD one wave streams MFMAs while its partner converts and stores.  Epilogues
//     with arithmetic (bias + residual, GELU, dGELU): every wave in phase, MFMA phases together and epilogues together (a
//     wave's vector instructions lose most of their issue rate beside a partner's MFMA stream; DESIGN_HISTORY.md section 19).
// The epilogue is nt_epilogue_lean (gemm_nt.hpp) - the arithmetic of nt_epilogue - and the k order of the accumulation is that
// of the tiled kernel, so results are bit-identical to gemm_bf16_nt_glds_kernel.
#include <utility>

#include "common.hpp"
#include "gemm_nt.hpp"

namespace avf {
namespace {

constexpr int WS_K = 512;          // the reduction length this kernel is built for (16 k-steps of 32)
constexpr int WS_BN = 256;         // panel width: 8 waves x 32 columns
constexpr int WS_KS = WS_K / 32;   // MFMA k-steps per tile

// Diagnostic build only (-DAVF_WS_STAMPS=1, tools/diag/ws_phases.py): s_memtime stamps of waves 0 and 4 of every workgroup,
// kept in LDS behind the ring (LDS writes count in lgkmcnt: the kernel's counted vmcnt waits see the same queue as the
// product build) and copied out at the end.  No stamp executes in the product build.
#ifndef AVF_WS_STAMPS
#define AVF_WS_STAMPS 0
#endif
#ifndef AVF_WS_EPI_PRIO
#define AVF_WS_EPI_PRIO 2
#endif
#ifndef AVF_WS_DBG
#define AVF_WS_DBG 0  // timing experiments (WRONG results): bit 0 = no DMA after the prologue, bit 1 = no stores (epilogue sees M = 0)
#endif
constexpr int WS_NSTAMP = 64;
#if AVF_WS_STAMPS
__device__ unsigned long long g_ws_stamps[512 * 8 * WS_NSTAMP];
#define WS_STAMP(i)                                                                              \
  do {                                                                                           \
    if (stamp_on && (i) < WS_NSTAMP) stamp_lds[(i)] = (unsigned long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void ws_glds16(const void* g, uint32_t lds_addr) {
  uint32_t keep;  // m0 is compiler-reserved: saved and restored around the DMA
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(g)
               : "memory");
}

// fragment-major image of W[N][512] (row-major, ldw): chunk (panel, wave, j, s, lane) = the 16 bytes lane (li, lg) feeds to the
// MFMA of column block j, k-step s:  W[256 panel + 32 wave + 16 j + li][32 s + 8 lg .. + 7]
__global__ __launch_bounds__(256) void pack_ws_kernel(const bf16* __restrict__ w, int64_t ldw, uint4* __restrict__ out, int n_chunks) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n_chunks) return;
  const int lane = o & 63, s = (o >> 6) & 15, j = (o >> 10) & 1, wave = (o >> 11) & 7, panel = o >> 14;
  const int li = lane & 15, lg = lane >> 4;
  const int n = panel * WS_BN + wave * 32 + j * 16 + li;
  out[o] = *reinterpret_cast<const uint4*>(w + (int64_t)n * ldw + s * 32 + lg * 8);
}

template <int EPI, typename CT, int MI, int NSLOT, bool CS, bool MXO = false, bool DROP = false>
__global__ __launch_bounds__(512) void gemm_bf16_nt_ws_kernel(NtParams p, const uint4* __restrict__ wp, int P, int G, int tq, int tr) {
  constexpr int NI = 2;
  constexpr int BM = 16 * MI;            // rows per tile
  constexpr int CH = BM * 128;           // bytes of one 64-deep k-chunk of a tile (the tiled kernel's stage image)
  constexpr int SLOT = 8 * CH;           // one tile: 8 chunks
  constexpr int PPW = BM / 8;            // DMA wave-instructions per wave and tile (8 rows x 128 B each)
  static_assert(BM % 8 == 0 && 7 * CH + (MI - 1) * 2048 < 65536, "tile shape");
  static_assert(NSLOT == 3, "the vmcnt bookkeeping below is written for two tiles of look-ahead");
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  // workgroups dealt to one XCD (ids congruent mod 8) take consecutive logical ids: the P workgroups of a row range (one per
  // panel) sit on one XCD and find the A tiles their neighbours fetched in its L2
  const int nwg = gridDim.x;
  const int L = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int grp = L / P, panel = L - grp * P;
  if (grp >= G) return;
  // group g takes tq tiles (+ 1 for the first tr groups) of the T = G tq + tr row tiles
  const int t0 = grp * tq + (grp < tr ? grp : tr);
  const int nt = tq + (grp < tr ? 1 : 0);
  if (nt <= 0) return;
  const int n0 = panel * WS_BN + wave * 32;
#if AVF_WS_STAMPS
  const bool stamp_on = lane == 0;
  unsigned long long* stamp_lds = reinterpret_cast<unsigned long long*>(dsm + NSLOT * SLOT + 8 * 2 * MI * 1024) + wave * WS_NSTAMP;
  if (stamp_on)
    for (int i = 0; i < WS_NSTAMP; ++i) stamp_lds[i] = 0;
  WS_STAMP(0);
#if AVF_WS_STAMPS
  if (stamp_on) stamp_lds[62] = (unsigned long long)__builtin_amdgcn_s_memrealtime();  // the 100 MHz constant clock: calibrates s_memtime
#endif
#endif

  // LDS-DMA pieces of a tile: piece q = wave + 8 jj covers k-chunk q / PPW, rows 8 (q % PPW) .. + 7; lane l fills row l >> 3,
  // slot l & 7 from source chunk (l & 7) ^ (l >> 3) (the XOR swizzle of nt_off() on the source side)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  auto dma_tile = [&](int tile, uint32_t slot_addr) {
    const int row0 = tile * BM;
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) {
      const int q = wave + 8 * jj;
      const int c = q / PPW, rg = q % PPW;
      int r = row0 + rg * 8 + lrow;
      r = r < p.M ? r : p.M - 1;  // rows past the edge re-read the last row and are never stored
      ws_glds16(p.A + (int64_t)r * p.lda + c * 64 + lchunk * 8,
                __builtin_amdgcn_readfirstlane(slot_addr + (uint32_t)(c * CH + rg * 1024)));
    }
  };
  uint32_t abase[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) abase[ks] = lds0 + nt_off(li, ks * 4 + lg);

#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i)
    if (i < nt) dma_tile(t0 + i, lds0 + i * SLOT);
  // (the staged epilogue operands of tile 0 ride with them: issued below, once the lambda exists - still ahead of the
  //  weight loads whose vmcnt(0) covers everything)
  // this wave's weight fragments (coalesced: 64 lanes x 16 B per instruction), behind the first tiles' DMA in the queue.
  // The 256 KiB of a workgroup's panel come out of the XCD's L2 at the rate all its CUs share (3.5 us when the eight waves of
  // every CU ask at once), so the two wave halves take TURNS: waves 0..3 load first and run tile 0 while waves 4..7 - which
  // trail them by half a period for the whole launch anyway - load theirs.
  // Two schedules.  Plain stores (EPI_NONE): the two wave halves run half a period apart - waves 4..7 take the tile barrier
  // between their MFMAs and their epilogue -, so one half's stores drain under the other half's MFMAs.  Epilogues with
  // arithmetic (bias + residual, GELU, dGELU): every wave takes the barrier at the top of the tile, MFMA phases together and
  // epilogues together - a wave's vector instructions get ~1 issue slot per MFMA of its SIMD partner
  // (tools/diag/mfma_valu_overlap.hip), so an epilogue of 130 - 220 of them runs 1.7x longer beside a partner in its MFMA
  // phase than beside one in its own epilogue (MLP1 27.9 -> 25.8 us, dGELU 25.5 -> 24.2 alone; QKV 29.6 -> 30.4 the other way)
#ifndef AVF_WS_INPHASE
#define AVF_WS_INPHASE (EPI != AVF_EPI_NONE)
#endif
  const bool late = !(AVF_WS_INPHASE) && wave >= 4;  // wave-uniform
  bf16x8_t wr[NI][WS_KS];
  const bf16x8_t* wsrc = reinterpret_cast<const bf16x8_t*>(wp) + ((size_t)(panel * 8 + wave) * NI * WS_KS) * 64 + lane;
  if (!late) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int s = 0; s < WS_KS; ++s) wr[j][s] = wsrc[(j * WS_KS + s) * 64];
    // ONE compiler-visible vmcnt(0): the weights (and, the queue being in order, the first tiles) have landed.  As a builtin
    // the wait-count pass knows it: without it hipcc waits for each weight register at its first use INSIDE the tile loop,
    // where a counted vmcnt(3..23) in front of the MFMAs also waits for the DMA issued at the top of the iteration
    WS_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    WS_STAMP(2);
  } else {
    wait_vmcnt<0>();  // own pieces of the first tiles
    __builtin_amdgcn_s_barrier();
    WS_STAMP(1);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int s = 0; s < WS_KS; ++s) wr[j][s] = wsrc[(j * WS_KS + s) * 64];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    WS_STAMP(2);
  }

  f32x4_t acc[MI][NI];
  // CS: the column sums of ALL the tiles this wave owns, per lane; one DPP reduction and one partial row per workgroup
  // group at the end (row `grp` of cs_partial) instead of one per tile
  float cs_acc[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs_acc[j][r] = 0.f;
  // The epilogue's operands.  The bias of the lane's columns is loaded once.  The residual / saved pre-activation rows (2-byte
  // rows: the bf16 streams) of tile t + 1 travel by LDS-DMA into a per-wave staging buffer while tile t is being finished -
  // requested BEHIND the DMA of tile t + 2 and read a whole period later: in the pipeline these rows are HBM-cold (the saved
  // pre-activation was written a forward pass ago), and a register prefetch issued in front of a tile's MFMAs gave them one
  // MFMA phase in flight (16 KiB per CU: the dGELU GEMM ran at 2.6 TB/s).  No VGPR destination, no compiler-visible load in the
  // tile loop: the wait-count pass inserts nothing there, the kernel's own vmcnt counts every instruction.  fp32 rows (the
  // fp32 residual stream) keep the register prefetch in front of the MFMAs.
  constexpr bool PRE_LDS = (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_DGELU) && sizeof(CT) == 2;
  constexpr int KPRE = PRE_LDS ? MI * (NI / 2) : 0;  // LDS-DMA instructions per tile and wave (16 rows x 64 B each)
  static_assert(NI == 2, "one column-block pair per wave");
  const uint32_t pre_lds = lds0 + NSLOT * SLOT + (uint32_t)wave * (2 * MI * 1024);
  const int cp0 = n0 + 4 * lg + ((lg & 1) ? 12 : 0);  // nt_epilogue_lean: the lane's 8 columns of the pair after the exchange
  auto pre_dma = [&](int tile, int buf) {
    if constexpr (PRE_LDS) {
      const bf16* src = (EPI == AVF_EPI_BIAS_RES) ? (const bf16*)p.residual : (const bf16*)p.aux;
      const int64_t ldx = (EPI == AVF_EPI_BIAS_RES) ? p.ldres : p.ldaux;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int r = tile * BM + 16 * i + li;
        r = r < p.M ? r : p.M - 1;
        ws_glds16(src + (int64_t)r * ldx + cp0, __builtin_amdgcn_readfirstlane(pre_lds + (uint32_t)((buf * MI + i) * 1024)));
      }
    }
  };
  pre_dma(t0, 0);  // tile 0's rows: landed by the first counted wait of the loop (older than everything it leaves in flight)
  NtPre<MI, NI> pre;
  nt_epi_prefetch<EPI, CT, MI, NI, true>(p, t0 * BM, n0, li, lg, pre);  // the bias (and, fp32 rows, tile 0's operands)
  // compiler-visible, so that the wait-count pass knows the bias has landed: it would otherwise wait vmcnt(0) at the bias'
  // first use in EVERY iteration's epilogue - behind the DMA that iteration has just issued
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // DROP: the mask key of the epilogue's dropout site, once per kernel (a scalar load when the seed lives in device memory)
  uint64_t dkey = 0;
  if constexpr (DROP) dkey = drop_key(p.drop);
  uint32_t rd_slot = 0, wr_slot = (NSLOT - 1) * SLOT;
  for (int t = 0; t < nt; ++t) {
    WS_STAMP(4 + 6 * t);
    if (!late) __builtin_amdgcn_s_barrier();  // tile t landed for every wave; every wave is done reading tile t - 1
    WS_STAMP(5 + 6 * t);
    if constexpr (!PRE_LDS) {
      if (t > 0) nt_epi_prefetch<EPI, CT, MI, NI, false>(p, (t0 + t) * BM, n0, li, lg, pre);  // (covered by the wait below)
    }
    WS_STAMP(6 + 6 * t);
    // ---- the tile's 16 k-steps, fully unrolled ----
    // three fragment buffers: the reads of chunk c + 2 are issued before the MFMAs of chunk c (an LDS read under load takes
    // longer than the 8 MFMAs of one chunk: with one chunk of look-ahead the 64 MFMAs took 2200 cycles instead of 1024)
    bf16x8_t fa[3][2][MI];
    const uint32_t so = rd_slot;
#pragma unroll
    for (int c0 = 0; c0 < 2; ++c0)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        lds_read_frags<bf16x8_t, 2048>(fa[c0][ks], abase[ks] + so + c0 * CH, std::make_integer_sequence<int, MI>{});
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // reads of chunk c have returned; those of chunk c + 1 (2 MI instructions) may still be in flight.  LDS returns in
      // order, so a compiler-issued scalar load in the queue can only make this wait longer, never shorter
      if (c + 1 < 8) wait_lgkmcnt<2 * MI>();
      else wait_lgkmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < 8) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          lds_read_frags<bf16x8_t, 2048>(fa[(c + 2) % 3][ks], abase[ks] + so + (c + 2) * CH, std::make_integer_sequence<int, MI>{});
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            if (c == 0 && ks == 0)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[j][0], fa[0][0][i], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[j][2 * c + ks], fa[c % 3][ks][i], acc[i][j], 0, 0, 0);
          }
    }
    // the DMA of tile t + 2 is issued BEHIND the MFMAs of tile t: an LDS-DMA in flight counts in the issuing wave's lgkmcnt as
    // well as in its vmcnt, so a DMA issued in front of the K-loop turns every counted lgkmcnt wait of the fragment reads into
    // a wait for the DMA (64 MFMAs: 1850 cycles with the DMA in front, 1280 with none in flight)
    const bool more2 = t + 2 < nt, more1 = t + 1 < nt;
    if (more2 && !(AVF_WS_DBG & 1)) dma_tile(t0 + t + 2, lds0 + wr_slot);
    if (more1) pre_dma(t0 + t + 1, (t + 1) & 1);
    WS_STAMP(7 + 6 * t);
    // everything OLDER than what this iteration issued has landed - this wave's pieces of tile t + 1 and the staged epilogue
    // operands of tile t (both issued an iteration ago); the queue is in order, so the count is that of this iteration's own
    // instructions: PPW DMA pieces and KPRE staging pieces
    if (more2) wait_vmcnt<PPW + KPRE>();
    else if (more1) wait_vmcnt<KPRE>();
    else wait_vmcnt<0>();
    WS_STAMP(8 + 6 * t);
    if (late && more1) __builtin_amdgcn_s_barrier();
    WS_STAMP(9 + 6 * t);
    if constexpr (PRE_LDS) {  // this wave's staged rows of tile t: the 16 bytes a lane would have loaded (lane-linear image)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = lds_read_b128<u32x4_t, 0>(pre_lds + (uint32_t)(((t & 1) * MI + i) * 1024) + (uint32_t)lane * 16);
        pre.raw[i][0] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      wait_lgkmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
    // the epilogue's vector instructions outrank the SIMD partner's MFMA stream: a wave in its epilogue is on the workgroup's
    // critical path (2480 cycles beside a partner issuing MFMAs against 1450 alone), the matrix pipe has slack
    __builtin_amdgcn_s_setprio(AVF_WS_EPI_PRIO);
    if constexpr ((AVF_WS_DBG & 2) != 0) {
      NtParams p2 = p;
      p2.M = 0;  // every store predicated off
      nt_epilogue_lean<EPI, CT, MI, NI, 0, true>(p2, acc, (t0 + t) * BM, n0, li, lg, -1, &pre);
    } else {
      // (no bias on the plain and the dGELU form: gemm_bf16_nt_ws_ok)
      nt_epilogue_lean<EPI, CT, MI, NI, CS ? 2 : 0, true, MXO, EPI != AVF_EPI_NONE && EPI != AVF_EPI_DGELU, DROP>(
          p, acc, (t0 + t) * BM, n0, li, lg, -1, &pre, cs_acc, dkey);
    }
    __builtin_amdgcn_s_setprio(0);
    rd_slot = rd_slot + SLOT == NSLOT * SLOT ? 0 : rd_slot + SLOT;
    wr_slot = wr_slot + SLOT == NSLOT * SLOT ? 0 : wr_slot + SLOT;
  }
  if constexpr (CS) nt_cs_flush<NI>(p, cs_acc, grp, n0, li, lg);
#if AVF_WS_STAMPS
  WS_STAMP(3);
  if (stamp_on) stamp_lds[63] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  if (stamp_on && blockIdx.x < 512)
    for (int i = 0; i < WS_NSTAMP; ++i) g_ws_stamps[(blockIdx.x * 8 + wave) * WS_NSTAMP + i] = stamp_lds[i];
#endif
}

int ws_enabled() {
  static const int on = [] {
    const char* e = tuning_env("AVF_NT_WS");  // 0: every NT GEMM on the tiled kernel (A/B aid)
    return (e && *e) ? atoi(e) : 1;
  }();
  return on;
}
int ws_grid() {
  static const int n = [] {
    const char* e = tuning_env("AVF_NT_WS_GRID");  // tuning aid: persistent workgroups (a multiple of 8)
    int v = (e && *e) ? atoi(e) : 0;
    if (v <= 0) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
      v = cus;
    }
    return v & ~7;
  }();
  return n;
}

template <int EPI, typename CT, int MI, int NSLOT, bool CS, bool MXO = false, bool DROP = false>
int launch_ws(const NtParams& p, const void* bp, hipStream_t s, int* part_rows, TimingScope* ts) {
  constexpr int BM = 16 * MI;
  // the A ring, the per-wave staging of the epilogue's 2-byte operand rows (two tiles), [diagnostic build: the stamps]
  constexpr int SMEM = NSLOT * BM * 1024 + 8 * 2 * MI * 1024 + (AVF_WS_STAMPS ? 8 * WS_NSTAMP * 8 : 0);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static PerDeviceOnce raised;
  if (SMEM > 64 * 1024 && raised.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_ws_kernel<EPI, CT, MI, NSLOT, CS, MXO, DROP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    AVF_REQUIRE(e == hipSuccess, "gemm_bf16_nt_ws: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    raised.mark();
  }
  const int P = p.N / WS_BN;
  const int T = (p.M + BM - 1) / BM;
  const int nwg = ws_grid();
  int G = nwg / P;
  if (G > T) G = T;
  AVF_REQUIRE(G >= 1, "gemm_bf16_nt_ws: more column panels (%d) than persistent workgroups (%d)", P, nwg);
  *part_rows = CS ? G : T;  // column sums: one partial row per workgroup group
  // ... which must fit the caller's workspace (avf_gemm_nt_ws_workspace_bytes / gemm_nt_colsum_ws) - checked BEFORE the launch
  AVF_REQUIRE(!CS || (size_t)G * p.N * sizeof(float) <= gemm_nt_colsum_ws(p.M, p.N),
              "gemm_bf16_nt_ws: column-sum partials exceed their workspace (internal error)");
  if (shape_log_on()) {
    const double csz = sizeof(CT);
    const double epi_b = (EPI == AVF_EPI_BIAS_RES || EPI == AVF_EPI_BIAS_GELU || EPI == AVF_EPI_DGELU) ? csz * p.M * p.N : 0.0;
    shape_log("gemm_bf16_nt,gemm_bf16_nt_ws_kernel<%d, %s, %d, %d, %s, %s, %s>,%d,%d,%d,%d,%d,%.0f,%.0f", EPI,
              sizeof(CT) == 4 ? "float" : "bf16", MI, NSLOT, CS ? "true" : "false", MXO ? "true" : "false", DROP ? "true" : "false", nwg, p.M, p.N, p.K, EPI, 2.0 * p.M * p.N * p.K,
              2.0 * ((double)p.M * p.K + (double)p.N * p.K) + csz * p.M * p.N + epi_b);
  }
  launch_in_scope(ts, gemm_bf16_nt_ws_kernel<EPI, CT, MI, NSLOT, CS, MXO, DROP>, dim3(nwg), dim3(512), SMEM, s, p, (const uint4*)bp, P, G, T / G, T % G);
  return 0;
}

template <int EPI, typename CT>
int launch_ws_any(const NtParams& p, const void* bp, hipStream_t s, int* part_rows, TimingScope* ts) {
  // column sums ride on the dGELU epilogue only (db1 of the layer's backward); other epilogues with a colsum request go to
  // the tiled kernel (gemm_bf16_nt_ws_ok)
  if constexpr (EPI != AVF_EPI_NONE) {
    if (p.drop.thresh16) {  // the epilogue's dropout site (training at p > 0: heads.py:277 and every real instantiation)
      AVF_REQUIRE(!p.mxq, "gemm_bf16_nt_ws: no MX-FP8 image beside a dropout site (internal error)");
      if constexpr (EPI == AVF_EPI_DGELU && sizeof(CT) == 2) {
        if (p.cs_partial) return launch_ws<EPI, CT, 2, 3, true, false, true>(p, bp, s, part_rows, ts);
      }
      AVF_REQUIRE(!p.cs_partial, "gemm_bf16_nt_ws: no instantiation for this combination of options (internal error)");
      return launch_ws<EPI, CT, 2, 3, false, false, true>(p, bp, s, part_rows, ts);
    }
  }
  if constexpr (EPI == AVF_EPI_DGELU && sizeof(CT) == 2) {
    if (p.cs_partial && p.mxq) return launch_ws<EPI, CT, 2, 3, true, true>(p, bp, s, part_rows, ts);  // (the fp8 mode's dGELU)
    if (p.cs_partial) return launch_ws<EPI, CT, 2, 3, true>(p, bp, s, part_rows, ts);
  }
  AVF_REQUIRE(!p.mxq && !p.cs_partial, "gemm_bf16_nt_ws: no instantiation for this combination of options (internal error)");
  return launch_ws<EPI, CT, 2, 3, false>(p, bp, s, part_rows, ts);
}

}  // namespace

#if AVF_WS_STAMPS
extern "C" int avf_ws_stamps_read(void* host_out, size_t bytes) {
  if (bytes > sizeof(g_ws_stamps)) bytes = sizeof(g_ws_stamps);
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

size_t pack_ws_bytes(int64_t N, int64_t K) { return (size_t)N * K * 2; }
bool pack_ws_ok(int64_t N, int64_t K) { return K == WS_K && N > 0 && N % WS_BN == 0; }

int pack_ws(const void* w_bf16, int64_t ldw, int64_t N, int64_t K, void* out, hipStream_t s) {
  AVF_REQUIRE(pack_ws_ok(N, K), "pack_ws: the weight-stationary image needs K == 512 and N %% 256 == 0 (N=%lld K=%lld)",
              (long long)N, (long long)K);
  AVF_REQUIRE(w_bf16 && out && ldw % 8 == 0 && ((uintptr_t)w_bf16 & 15) == 0 && ((uintptr_t)out & 15) == 0,
              "pack_ws: null or misaligned pointer");
  const int n_chunks = (int)(N * (K / 8));
  pack_ws_kernel<<<dim3((n_chunks + 255) / 256), dim3(256), 0, s>>>((const bf16*)w_bf16, ldw, (uint4*)out, n_chunks);
  return check_launch("pack_ws_kernel");
}

// shapes the persistent kernel takes (the caller falls back to the tiled kernel otherwise)
bool gemm_bf16_nt_ws_ok(const GemmArgs& a) {
  if (!ws_enabled() || !a.Bp) return false;
  if (!pack_ws_ok(a.N, a.K) || a.M < 2048) return false;
  if (a.lda % 8 || a.ldc % 8 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.C & 15) || ((uintptr_t)a.Bp & 15)) return false;
  // the epilogue's prefetch takes nt_epilogue's wide path for 2-byte residual / pre-activation rows unconditionally
  if (nt_wide_stores() != 1) return false;
  if (a.epilogue == AVF_EPI_BIAS_RES && (a.ldres % 8 || ((uintptr_t)a.residual & 15))) return false;
  if ((a.epilogue == AVF_EPI_DGELU || a.epilogue == AVF_EPI_BIAS_GELU) && (a.ldaux % 8 || ((uintptr_t)a.aux & 15))) return false;
  if (a.epilogue == AVF_EPI_DGELU && a.c_dtype != AVF_BF16) return false;
  if ((a.epilogue == AVF_EPI_DGELU || a.epilogue == AVF_EPI_NONE) && a.bias) return false;  // (compiled without the bias add)
  // the lean epilogue (gemm_nt.hpp) carries a dropout site on a fused epilogue only (never beside an MX-FP8 image) and sums
  // columns on the dGELU epilogue only
  if (a.drop.thresh16 && (a.epilogue == AVF_EPI_NONE || a.mx_q)) return false;
  if (a.colsum && a.epilogue != AVF_EPI_DGELU) return false;
  // the MX-FP8 image of C: on the dGELU form with column sums only (what the fp8 mode's backward asks for)
  if (a.mx_q && !(a.mx_s && a.epilogue == AVF_EPI_DGELU && a.colsum && a.c_dtype == AVF_BF16 && a.N % 32 == 0 && ((uintptr_t)a.mx_q & 7) == 0))
    return false;
  if (a.c_dtype == AVF_BF16 && (a.N % 8 || a.ldc % 8)) return false;
  return a.N / WS_BN <= ws_grid();
}

// ... and the shapes gemm_bf16_nt() SENDS there (avf_gemm_nt_ws takes every shape the kernel can run)
bool gemm_bf16_nt_ws_preferred(const GemmArgs& a) {
  if (!gemm_bf16_nt_ws_ok(a)) return false;
  {
    static const int mask = [] {
      const char* e = tuning_env("AVF_NT_WS_EPI");  // tuning aid: bit e set = epilogue e (AVF_EPI_*) may take the persistent kernel
      return (e && *e) ? atoi(e) : 15;
    }();
    if (!((mask >> a.epilogue) & 1) && !a.mx_q) return false;
  }
  // bias + residual (the out-projection, N = dim): the persistent kernel pays its prologue (a CU's 256 KiB of weights out of
  // L2, ~3 us) once per workgroup; with 4 row tiles per workgroup (C3: 512 tiles over 128 groups) the tiled kernel is ahead -
  // 15.5 against 16.8 us at C3, 13.1 against 14.4 at C2, per-shape rocprof of the pipeline - so this form needs 6 tiles
  if (a.epilogue == AVF_EPI_BIAS_RES) {
    const int64_t P = a.N / WS_BN, T = (a.M + 31) / 32;
    int64_t G = ws_grid() / P;
    G = G > T ? T : G;
    static const int min_tiles = [] {
      const char* e = tuning_env("AVF_NT_WS_RES_TILES");  // tuning aid
      return (e && *e) ? atoi(e) : 6;
    }();
    if (T / G < min_tiles) return false;
  }
  // plain, N = dim (d_o in backward): the same prologue serves two column panels only - below three row tiles per workgroup the
  // tiled kernel (whose launches at these sizes leave slots empty and warm their weights, gemm_bf16.hip) is ahead with the weight
  // image cold as in the step: 10.4 against 12.2 us at C2's 10368 rows, 8.1 / 9.7 at 5184, 6.1 / 7.8 at 2048; N = 1536 stays
  // here at every row count (tools/diag/ws_vs_tiled_small_m.py)
  if (a.epilogue == AVF_EPI_NONE && a.N / WS_BN <= 2) {
    const int64_t P = a.N / WS_BN, T = (a.M + 31) / 32;
    int64_t G = ws_grid() / P;
    G = G > T ? T : G;
    static const int min_tiles0 = [] {
      const char* e = tuning_env("AVF_NT_WS_PLAIN_TILES");  // tuning aid
      return (e && *e) ? atoi(e) : 3;
    }();
    if (T < min_tiles0 * G) return false;
  }
  return true;
}

int gemm_bf16_nt_ws(const GemmArgs& a, hipStream_t s, int* part_rows_out) {
  AVF_REQUIRE(gemm_bf16_nt_ws_ok(a), "gemm_bf16_nt_ws: unsupported shape / arguments");
  NtParams p;
  const double csz = a.c_dtype == AVF_F32 ? 4.0 : 2.0;
  const double epi_bytes = (a.epilogue == AVF_EPI_BIAS_RES || a.epilogue == AVF_EPI_BIAS_GELU || a.epilogue == AVF_EPI_DGELU)
                               ? csz * a.M * a.N : 0.0;
  TimingScope ts(KC_GEMM_BF16_NT, 2.0 * a.M * a.N * a.K, 2.0 * (a.M * a.K + a.N * a.K) + csz * a.M * a.N + epi_bytes, s,
                 /*per_kernel=*/true);
  p.A = (const bf16*)a.A; p.lda = a.lda; p.B = nullptr; p.ldb = 0;
  p.C = a.C; p.ldc = a.ldc; p.bias = a.bias; p.residual = a.residual; p.ldres = a.ldres;
  p.aux = a.aux; p.ldaux = a.ldaux;
  p.drop = a.drop;
  p.mxq = (uint8_t*)a.mx_q; p.mxs = (uint8_t*)a.mx_s;
  p.wide = nt_wide_stores();
  p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
  p.cs_partial = a.colsum ? (float*)a.workspace : nullptr;
  AVF_REQUIRE(!a.colsum || a.workspace, "gemm_bf16_nt_ws: column-sum workspace missing");
  const bool cf32 = a.c_dtype == AVF_F32;
  int part_rows = 0;
#define LAUNCH_WS(E)                                                               \
  do {                                                                             \
    if (cf32) AVF_TRY((launch_ws_any<E, float>(p, a.Bp, s, &part_rows, &ts)));     \
    else AVF_TRY((launch_ws_any<E, bf16>(p, a.Bp, s, &part_rows, &ts)));           \
  } while (0)
  switch (a.epilogue) {
    case AVF_EPI_NONE: LAUNCH_WS(AVF_EPI_NONE); break;
    case AVF_EPI_BIAS_RES:
      AVF_REQUIRE(a.residual && a.ldres % 4 == 0, "gemm_bf16_nt_ws: BIAS_RES needs a residual (in C's storage type)");
      LAUNCH_WS(AVF_EPI_BIAS_RES);
      break;
    case AVF_EPI_BIAS_GELU:
      AVF_REQUIRE(a.aux && a.ldaux % 4 == 0, "gemm_bf16_nt_ws: aux missing");
      LAUNCH_WS(AVF_EPI_BIAS_GELU);
      break;
    case AVF_EPI_DGELU:
      AVF_REQUIRE(a.aux && a.ldaux % 4 == 0 && !cf32, "gemm_bf16_nt_ws: aux missing (or an fp32 C)");
      AVF_TRY((launch_ws_any<AVF_EPI_DGELU, bf16>(p, a.Bp, s, &part_rows, &ts)));
      break;
    default: AVF_REQUIRE(false, "gemm_bf16_nt_ws: bad epilogue %d", a.epilogue);
  }
#undef LAUNCH_WS
  *part_rows_out = part_rows;
  return check_launch("gemm_bf16_nt_ws_kernel");
}

}  // namespace avf
