// mlp_fused.hip - the FeedForward sublayer (reference models/heads.py:188-199) as ONE kernel per direction.
//
//   forward  (MODE 0):  u = h W1^T + b1,  g = gelu(u),  x_out = g W2^T + b2 + x_mid        (u and g are saved for backward)
//   backward (MODE 1):  du = (dy W2) o gelu'(u),  dh = du W1                                 (du is kept for dW1, db1 = colsum du)
//
// Both are the same chain  T = X Wa^T -> elementwise -> OUT = T Wb^T  with the [R, M] intermediate T never re-read from
// memory: a workgroup of 8 waves owns a 64-row panel of X, walks the hidden dimension M in 128-column chunks, computes the
// chunk's T tile (64 x 128, K = D) into registers, runs the elementwise epilogue there, leaves the bf16 tile in LDS as the A
// operand of the second product and accumulates OUT (64 x D, fp32, in registers for the whole kernel) over the chunks.
// Weights stream through one 3-slot LDS ring of 32 KiB stages by LDS-DMA, two stages ahead of the MFMAs; the stage list of a
// chunk is  D/64 stages of (X 64x64 | Wa 128x64)  then  2 D/256 stages of (Wb 256x64),  i.e. the DMA of the next chunk's first
// product runs under the second product of this one.  Every wave counts its own DMA instructions (s_waitcnt vmcnt(N), N a
// compile-time function of the position in the chunk - the chunk body is fully unrolled), one s_barrier per stage.
// DESIGN.md section 15 has the roofline of this shape (the per-CU L2 -> LDS rate, not the MFMA) and the measurements.
#include "gemm_nt.hpp"

namespace avf {
namespace {

struct MlpParams {
  const bf16* X;      // [R][D]   MODE 0: LayerNorm output h;      MODE 1: dy (bf16 image of the gradient of the layer output)
  const bf16* Wa;     // [M][D]   MODE 0: W1;                      MODE 1: W2^T
  const bf16* Wb;     // [D][M]   MODE 0: W2;                      MODE 1: W1^T
  const float* ba;    // [M]      MODE 0: b1
  const float* bb;    // [D]      MODE 0: b2
  const void* res;    // [R][D]   MODE 0: x_mid, in OT
  void* out;          // [R][D]   MODE 0: x_out in OT;             MODE 1: dh (bf16)
  bf16* u;            // [R][M]   MODE 0: written;                 MODE 1: read (saved pre-activation)
  bf16* g;            // [R][M]   MODE 0: gelu(u), written;        MODE 1: du, written
  float* cs_partial;  // MODE 1: [2 R / 64][M] column sums of du over groups of 32 rows (the fold gives db1)
  int R, M;
  int dbg;  // diagnostic (AVF_MLPF_DBG): bit 0 = no DMA, bit 1 = no fragment reads / MFMAs, bit 2 = no first-product epilogue
};

template <int L>
struct IC {
  static constexpr int v = L;
};
template <typename F, int... Ls>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Ls...>) {
  (f(IC<Ls>{}), ...);
}

__device__ __forceinline__ void f_glds16(const void* g, uint32_t lds_addr) {  // LDS-DMA, opaque to hipcc's wait-count pass
  const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(la), "v"(g)
               : "memory");
}

constexpr int F_SLOT = 32768;  // one ring stage: (X 8 KiB | Wa 16 KiB) or Wb 32 KiB

template <int NH, int MODE>
struct FusedLayout {
  static constexpr int G_OFF = 3 * F_SLOT;                        // T tile, bf16, two 64-wide k-tiles of 8 KiB
  static constexpr int U_OFF = G_OFF + 16384;                     // MODE 1: the u tile (64 x 128 bf16), DMA'd per chunk
  static constexpr int B_OFF = U_OFF + (MODE == 1 ? 16384 : 0);   // MODE 0: b1 [M] fp32
};

// NH = D / 256.  OT: storage type of the residual stream (MODE 0); MODE 1 writes bf16.
template <int NH, int MODE, typename OT>
__global__ __launch_bounds__(512) void mlp_fused_kernel(MlpParams p) {
  constexpr int D = 256 * NH, KT1 = D / 64, NP2 = 2 * NH, SPC = KT1 + NP2;  // stages per chunk (a multiple of 3)
  using LY = FusedLayout<NH, MODE>;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // first product: 2 x 4 waves of 32 x 32
  const int li = lane & 15, lg = lane >> 4;
  const int r0 = blockIdx.x * 64;
  const int M = p.M, nchunk = M >> 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;

  // ---- DMA sources (8-row pieces of 1 KiB; the XOR swizzle of nt_off() sits on the source chunk) ----
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const bf16* xa = p.X + (int64_t)(r0 + wave * 8 + lrow) * D + lchunk * 8;
  const bf16* wa = p.Wa + (int64_t)(wave * 8 + lrow) * D + lchunk * 8;           // + chunk * 128 * D; second piece + 64 * D
  const bf16* wb = p.Wb + (int64_t)(wave * 8 + lrow) * M + lchunk * 8;           // + j * 64 * M, + 256 nh * M, + chunk * 128 + 64 ks
  // MODE 1: the u tile as 4-row pieces of 1 KiB (row = 256 bytes), 16-byte chunk c of row r at slot c ^ (r & 15)
  const int urow = lane >> 4;
  const bf16* ua = nullptr;
  if constexpr (MODE == 1) ua = p.u + (int64_t)(r0 + wave * 8 + urow) * M;  // pieces 2 wave, 2 wave + 1: rows 8 wave + 0..7

  auto issue = [&](auto lc, int c) {  // stage l of chunk c into slot l % 3
    constexpr int l = decltype(lc)::v;
    if (p.dbg & 1) return;
    const uint32_t slot = lds0 + (l % 3) * F_SLOT;
    if constexpr (l < KT1) {
      const int64_t wo = (int64_t)c * 128 * D + l * 64;
      f_glds16(xa + l * 64, slot + wave * 1024);
      f_glds16(wa + wo, slot + 8192 + wave * 1024);
      f_glds16(wa + wo + 64 * D, slot + 8192 + (wave + 8) * 1024);
      if constexpr (MODE == 1 && l == 2) {  // the chunk's u tile rides with its third stage (consumed after the KT1-th)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = wave * 8 + 4 * h + urow;  // row inside the panel
          f_glds16(ua + (int64_t)(4 * h) * M + c * 128 + (((lane & 15) ^ (row & 15)) << 3), lds0 + LY::U_OFF + (wave * 2 + h) * 1024);
        }
      }
    } else {
      constexpr int t = l - KT1, ks = t / NH, nh = t % NH;
      const bf16* src = wb + (int64_t)(256 * nh) * M + c * 128 + 64 * ks;
#pragma unroll
      for (int j = 0; j < 4; ++j) f_glds16(src + (int64_t)(64 * j) * M, slot + (wave + 8 * j) * 1024);
    }
  };

  f32x4_t U[2][2], OUT[NH][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) U[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) OUT[h][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a slot (row blocks 16 rows = 2048 bytes apart; swizzle depends on li & 7 only)
  uint32_t a1[2], b1[2], a2[2], b2[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    a1[kh] = lds0 + nt_off(wm * 32 + li, kh * 4 + lg);
    b1[kh] = lds0 + 8192 + nt_off(wn * 32 + li, kh * 4 + lg);
    a2[kh] = lds0 + LY::G_OFF + nt_off(li, kh * 4 + lg);
    b2[kh] = lds0 + nt_off(wave * 32 + li, kh * 4 + lg);
  }

  // prologue: the first two stages; MODE 0: b1 into LDS meanwhile (ordered by the first stage barrier)
  issue(IC<0>{}, 0);
  issue(IC<1>{}, 0);
  if constexpr (MODE == 0) {
    float* bl = reinterpret_cast<float*>(dsm + LY::B_OFF);
    for (int i = tid; i < M; i += 512) bl[i] = p.ba[i];
    wait_lgkmcnt<0>();  // (raw s_barrier below: the LDS writes must have completed before it)
  }

  // VM operations a wave issues in step l behind that step's DMA (they sit in the in-order queue between two stages):
  // the epilogue of the first product stores u and g (MODE 0) / du and the column-sum partial (MODE 1): 4 instructions
  auto extra = [](int l) constexpr { return l == KT1 - 1 ? 4 : 0; };
  auto ndma = [](int l) constexpr { return l < KT1 ? ((MODE == 1 && l == 2) ? 5 : 3) : 4; };

  auto chunk_body = [&](int c, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    static_for(
        [&](auto lc) {
          constexpr int l = decltype(lc)::v;
          // stage l has landed when all but the operations issued after it are done: those of step l-2 behind its DMA, the
          // DMA of stage l+1 (issued in step l-1) and what step l-1 issued behind it
          constexpr int l1 = (l + 1) % SPC, lm1 = (l + SPC - 1) % SPC, lm2 = (l + SPC - 2) % SPC;
          constexpr int pend = extra(lm2) + ((LAST && l + 1 >= SPC) ? 0 : ndma(l1)) + extra(lm1);
          wait_vmcnt<pend>();
          __builtin_amdgcn_s_barrier();
          if constexpr (l + 2 < SPC) issue(IC<l + 2>{}, c);
          else if constexpr (!LAST) issue(IC<l + 2 - SPC>{}, c + 1);
          const uint32_t so = (l % 3) * F_SLOT;
          if (p.dbg & 2) return;
          if constexpr (l < KT1) {
            bf16x8_t fa[2][2], fb[2][2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              lds_read_frags<bf16x8_t, 2048>(fa[kh], a1[kh] + so, std::make_integer_sequence<int, 2>{});
              lds_read_frags<bf16x8_t, 2048>(fb[kh], b1[kh] + so, std::make_integer_sequence<int, 2>{});
            }
            wait_lgkmcnt<4>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) U[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], U[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) U[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], U[i][j], 0, 0, 0);
            if (l == KT1 - 1 && !(p.dbg & 4)) {
              // ---- epilogue of the first product: elementwise, stores, the bf16 tile into LDS ----
              const int cb = c * 128 + wn * 32 + 4 * lg;  // this lane's first column (block j adds 16)
              float cs[2][4];
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int row = wm * 32 + 16 * i + li;  // inside the panel
                uint32_t tw[2][2], uw[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  float v[4] = {U[i][j][0], U[i][j][1], U[i][j][2], U[i][j][3]};
                  if constexpr (MODE == 0) {
                    const float4 bj = *reinterpret_cast<const float4*>(dsm + LY::B_OFF + (cb + 16 * j) * 4);
                    v[0] += bj.x; v[1] += bj.y; v[2] += bj.z; v[3] += bj.w;
                    uw[j][0] = pack_bf16x2(v[0], v[1]);
                    uw[j][1] = pack_bf16x2(v[2], v[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]);
                  } else {
                    const int cc = wn * 32 + 16 * j + 4 * lg;  // column inside the chunk
                    const uint2 ur = *reinterpret_cast<const uint2*>(dsm + LY::U_OFF + row * 256 + ((((cc >> 3) ^ (row & 15))) << 4) +
                                                                    (cc & 7) * 2);
                    v[0] *= dgelu_tanh_fast(__uint_as_float(ur.x << 16));
                    v[1] *= dgelu_tanh_fast(__uint_as_float(ur.x & 0xffff0000u));
                    v[2] *= dgelu_tanh_fast(__uint_as_float(ur.y << 16));
                    v[3] *= dgelu_tanh_fast(__uint_as_float(ur.y & 0xffff0000u));
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs[j][r] += v[r];
                  }
                  tw[j][0] = pack_bf16x2(v[0], v[1]);
                  tw[j][1] = pack_bf16x2(v[2], v[3]);
                  // A operand of the second product: k-tile wn / 2, 16-byte chunk (wn & 1) * 4 + 2 j + lg / 2, half lg & 1
                  *reinterpret_cast<uint2*>(dsm + LY::G_OFF + (wn >> 1) * 8192 + nt_off(row, (wn & 1) * 4 + 2 * j + (lg >> 1)) +
                                            (lg & 1) * 8) = make_uint2(tw[j][0], tw[j][1]);
                  U[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
                // 16-byte stores: lane pairs trade words (store_pair16 without its edge predicate - full tiles only)
                const int64_t ro = (int64_t)(r0 + row) * M;
                const int col = (lg & 1) ? cb + 16 - 4 : cb;
                {
                  const auto s0 = __builtin_amdgcn_permlane16_swap(tw[0][0], tw[1][0], false, false);
                  const auto s1 = __builtin_amdgcn_permlane16_swap(tw[0][1], tw[1][1], false, false);
                  *reinterpret_cast<uint4*>(p.g + ro + col) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
                if constexpr (MODE == 0) {
                  const auto s0 = __builtin_amdgcn_permlane16_swap(uw[0][0], uw[1][0], false, false);
                  const auto s1 = __builtin_amdgcn_permlane16_swap(uw[0][1], uw[1][1], false, false);
                  *reinterpret_cast<uint4*>(p.u + ro + col) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
              }
              if constexpr (MODE == 1) {  // column sums of this wave's 32 rows (two stores: the count in extra())
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  float4 t;
                  float* tp = &t.x;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    float s = cs[j][r];
                    s += AVF_DPP_F32(s, 0xB1);
                    s += AVF_DPP_F32(s, 0x4E);
                    s += AVF_DPP_F32(s, 0x124);
                    s += AVF_DPP_F32(s, 0x128);
                    tp[r] = s;
                  }
                  // (all 64 lanes store: the 16 lanes of a group hold the same sums and write the same 16 bytes - one
                  //  instruction, no exec-mask branch the count above would have to know about)
                  *reinterpret_cast<float4*>(p.cs_partial + (int64_t)(blockIdx.x * 2 + wm) * M + cb + 16 * j) = t;
                }
              }
              wait_lgkmcnt<0>();  // the tile is in LDS before the next stage's barrier lets the other waves read it
            }
          } else {
            constexpr int t = l - KT1, ks = t / NH, nh = t % NH;
            bf16x8_t fa[2][4], fb[2][2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              lds_read_frags<bf16x8_t, 2048>(fa[kh], a2[kh] + ks * 8192, std::make_integer_sequence<int, 4>{});
              lds_read_frags<bf16x8_t, 2048>(fb[kh], b2[kh] + so, std::make_integer_sequence<int, 2>{});
            }
            wait_lgkmcnt<6>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                OUT[nh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], OUT[nh][i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                OUT[nh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], OUT[nh][i][j], 0, 0, 0);
          }
        },
        std::make_integer_sequence<int, SPC>{});
  };

  for (int c = 0; c < nchunk - 1; ++c) chunk_body(c, std::false_type{});
  chunk_body(nchunk - 1, std::true_type{});

  // ---- final epilogue: the 64 x D accumulator ----
#pragma unroll
  for (int nh = 0; nh < NH; ++nh) {
    float4 bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 256 * nh + wave * 32 + 16 * j + 4 * lg;
      bj[j] = MODE == 0 ? *reinterpret_cast<const float4*>(p.bb + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t ro = (int64_t)(r0 + 16 * i + li) * D;
      uint32_t w[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = 256 * nh + wave * 32 + 16 * j + 4 * lg;
        float4 v = make_float4(OUT[nh][i][j][0] + bj[j].x, OUT[nh][i][j][1] + bj[j].y, OUT[nh][i][j][2] + bj[j].z,
                               OUT[nh][i][j][3] + bj[j].w);
        if constexpr (MODE == 0) {
          const float4 rv = load4<OT>((const OT*)p.res + ro + col);
          v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if constexpr (MODE == 0 && sizeof(OT) == 4) {
          *reinterpret_cast<float4*>((float*)p.out + ro + col) = v;
        } else {
          w[j][0] = pack_bf16x2(v.x, v.y);
          w[j][1] = pack_bf16x2(v.z, v.w);
        }
      }
      if constexpr (!(MODE == 0 && sizeof(OT) == 4)) {
        const int cb = 256 * nh + wave * 32 + 4 * lg;
        const auto s0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
        *reinterpret_cast<uint4*>((bf16*)p.out + ro + ((lg & 1) ? cb + 12 : cb)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    }
  }
}

template <int NH, int MODE, typename OT>
int launch_fused(const MlpParams& p, hipStream_t s, TimingScope* ts) {
  using LY = FusedLayout<NH, MODE>;
  const int smem = LY::B_OFF + (MODE == 0 ? p.M * 4 : 0);
  static PerDeviceOnce raised;
  if (raised.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_fused_kernel<NH, MODE, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    AVF_REQUIRE(e == hipSuccess, "mlp_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    raised.mark();
  }
  launch_in_scope(ts, mlp_fused_kernel<NH, MODE, OT>, dim3(p.R / 64), dim3(512), (uint32_t)smem, s, p);
  return check_launch("mlp_fused_kernel");
}

}  // namespace

bool mlp_fused_ok(int64_t R, int D, int M) {
  return R > 0 && R % 64 == 0 && (D == 256 || D == 512 || D == 768) && M >= 128 && M % 128 == 0 && M * 4 <= 24 * 1024;
}

int mlp_fused_fwd(const void* h, const void* w1, const float* b1, const void* w2, const float* b2, const void* x_mid, int x_dtype,
                  void* x_out, void* u, void* g, int64_t R, int D, int M, hipStream_t s) {
  AVF_REQUIRE(mlp_fused_ok(R, D, M), "mlp_fused_fwd: unsupported shape R=%lld D=%d M=%d", (long long)R, D, M);
  AVF_REQUIRE(h && w1 && b1 && w2 && b2 && x_mid && x_out && u && g, "mlp_fused_fwd: null pointer");
  MlpParams p;
  memset(&p, 0, sizeof(p));
  p.X = (const bf16*)h; p.Wa = (const bf16*)w1; p.Wb = (const bf16*)w2; p.ba = b1; p.bb = b2; p.res = x_mid; p.out = x_out;
  p.u = (bf16*)u; p.g = (bf16*)g; p.R = (int)R; p.M = M;
  if (const char* e = getenv("AVF_MLPF_DBG")) p.dbg = atoi(e);
  const double bytes = (double)R * D * 2 + 4.0 * D * M + (double)R * M * 4 + (double)R * D * (x_dtype == AVF_BF16 ? 4 : 8);
  TimingScope ts(KC_GEMM_BF16_NT, 4.0 * (double)R * D * M, bytes, s, /*per_kernel=*/true);
  shape_log("mlp_fused,mlp_fused_kernel<%d, 0, %s>,%lld,%lld,%d,%d,%d,%.0f,%.0f", D / 256, x_dtype == AVF_BF16 ? "bf16" : "float",
            (long long)(R / 64), (long long)R, D, M, 0, 4.0 * (double)R * D * M, bytes);
#define AVF_FUSED(NHV)                                                            \
  (x_dtype == AVF_BF16 ? launch_fused<NHV, 0, bf16>(p, s, &ts) : launch_fused<NHV, 0, float>(p, s, &ts))
  switch (D / 256) {
    case 1: return AVF_FUSED(1);
    case 2: return AVF_FUSED(2);
    default: return AVF_FUSED(3);
  }
#undef AVF_FUSED
}

// cs_partial: [2 R / 64][M] floats; *fold describes the fold that turns them into db1
int mlp_fused_bwd(const void* dy, const void* w2_t, const void* w1_t, const void* u, void* du, void* dh, float* cs_partial,
                  int64_t R, int D, int M, hipStream_t s) {
  AVF_REQUIRE(mlp_fused_ok(R, D, M), "mlp_fused_bwd: unsupported shape R=%lld D=%d M=%d", (long long)R, D, M);
  AVF_REQUIRE(dy && w2_t && w1_t && u && du && dh && cs_partial, "mlp_fused_bwd: null pointer");
  MlpParams p;
  memset(&p, 0, sizeof(p));
  p.X = (const bf16*)dy; p.Wa = (const bf16*)w2_t; p.Wb = (const bf16*)w1_t; p.out = dh; p.u = (bf16*)u; p.g = (bf16*)du;
  p.cs_partial = cs_partial; p.R = (int)R; p.M = M;
  if (const char* e = getenv("AVF_MLPF_DBG")) p.dbg = atoi(e);
  const double bytes = (double)R * D * 4 + 4.0 * D * M + (double)R * M * 4;
  TimingScope ts(KC_GEMM_BF16_NT, 4.0 * (double)R * D * M, bytes, s, /*per_kernel=*/true);
  shape_log("mlp_fused,mlp_fused_kernel<%d, 1, bf16>,%lld,%lld,%d,%d,%d,%.0f,%.0f", D / 256, (long long)(R / 64), (long long)R, D, M, 3,
            4.0 * (double)R * D * M, bytes);
  switch (D / 256) {
    case 1: return launch_fused<1, 1, bf16>(p, s, &ts);
    case 2: return launch_fused<2, 1, bf16>(p, s, &ts);
    default: return launch_fused<3, 1, bf16>(p, s, &ts);
  }
}

}  // namespace avf

/* ---- C ABI (include/avformer_hip.h) ---- */
extern "C" int avf_mlp_fused_ok(int64_t rows, int dim, int mlp_dim) { return avf::mlp_fused_ok(rows, dim, mlp_dim) ? 1 : 0; }
extern "C" int avf_mlp_fused_fwd(const void* h, const void* w1, const float* b1, const void* w2, const float* b2, const void* x_mid,
                                 int x_dtype, void* x_out, void* u, void* g, int64_t rows, int dim, int mlp_dim, void* stream) {
  return avf::mlp_fused_fwd(h, w1, b1, w2, b2, x_mid, x_dtype, x_out, u, g, rows, dim, mlp_dim, (hipStream_t)stream);
}
extern "C" size_t avf_mlp_fused_bwd_partial_rows(int64_t rows) { return (size_t)(2 * (rows / 64)); }
extern "C" int avf_mlp_fused_bwd(const void* dy, const void* w2_t, const void* w1_t, const void* u, void* du, void* dh,
                                 float* colsum_partial, int64_t rows, int dim, int mlp_dim, void* stream) {
  return avf::mlp_fused_bwd(dy, w2_t, w1_t, u, du, dh, colsum_partial, rows, dim, mlp_dim, (hipStream_t)stream);
}
