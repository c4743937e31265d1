// common.hpp - shared device/host helpers for libavformer_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/avformer_hip.h"

namespace avf {

// ---------------------------------------------------------------------------------------------
// error plumbing (nothing throws across the C ABI)
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// AVF_SHAPE_LOG=<file>: one line per hot-path launch ("class,kernel-template,grid,M,N,K,epilogue,flops,bytes") appended to
// <file> - tools/shape_table.py joins it with a rocprofv3 kernel trace into the per-shape table under profiles/.
bool shape_log_on();
void shape_log(const char* fmt, ...);
// Every tuning / A-B switch of the library (AVF_NT_WS, AVF_NT_TILE, AVF_ATTN_MERGED, ...: INTEGRATION.md lists them) is read
// through this: unless the process was started with AVF_TUNING=1 the switches do not exist - a product process runs ONE
// dispatch, the one the parity tests exercised, whatever else is in its environment.
inline const char* tuning_env(const char* name) {
  static const bool on = [] {
    const char* t = getenv("AVF_TUNING");
    return t && *t && atoi(t) != 0;
  }();
  return on ? getenv(name) : nullptr;
}

#define AVF_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::avf::set_error(__VA_ARGS__);    \
      return 1;                         \
    }                                   \
  } while (0)

#define AVF_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != 0) return _rc;    \
  } while (0)

// ---------------------------------------------------------------------------------------------
// optional per-kernel-class HIP-event timing (bench.py's roofline line); off by default
// ---------------------------------------------------------------------------------------------
enum KernelClass {
  KC_GEMM_BF16_NT = 0, KC_GEMM_BF16_TN, KC_GEMM_F32, KC_ATTN_FWD, KC_ATTN_BWD, KC_LAYERNORM, KC_OTHER, KC_GEMM_MX8_NT, KC_COUNT
};
struct TimingScope {  // HIP-event timing of the launches issued during its lifetime (off unless avf_timing_enable(1))
  int slot;
  hipStream_t stream;
  bool per_kernel;
  int cls;
  double flops, bytes;
  mutable int issued;
  // plain scope: one start/stop event pair recorded AROUND everything launched while it lives (includes ~2 us of
  // command-processor time per pair).  per_kernel scope: records nothing itself; every launch made through
  // launch_in_scope() gets its own pair ATTACHED to the dispatch (hipExtLaunchKernelGGL: the kernel's own begin / end
  // timestamps, the duration rocprofv3 reports); the first one carries the scope's flops / bytes.
  TimingScope(int cls, double flops, double bytes, hipStream_t s, bool per_kernel = false);
  ~TimingScope();
  bool events(hipEvent_t* start, hipEvent_t* stop) const;
};
// launch `kernel` inside a per_kernel scope (ts may be null or a plain scope: ordinary launch)
template <typename... Args, typename F = void (*)(Args...)>
inline void launch_in_scope(const TimingScope* ts, F kernel, dim3 grid, dim3 block, uint32_t smem, hipStream_t stream,
                            Args... args) {
  hipEvent_t e0, e1;
  if (ts && ts->events(&e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, e0, e1, 0, args...);
  else kernel<<<grid, block, smem, stream>>>(args...);
}

// ---------------------------------------------------------------------------------------------
// element types
// ---------------------------------------------------------------------------------------------
struct bf16 {
  uint16_t x;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;  // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 for the pair (a scalar cast per element costs two converts plus shift/or)
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// the scale byte and its reciprocal as a float, from a block maximum
__device__ __forceinline__ uint32_t mx8_scale_byte(float amax, float* inv) {
  const uint32_t eb = (__float_as_uint(amax) >> 23) & 255u;
  const uint32_t sb = eb > 8u ? eb - 8u : 0u;
  *inv = __uint_as_float((254u - sb) << 23);
  return sb;
}
__device__ __forceinline__ uint32_t mx8_pack4(const float (&v)[4], float inv) {
  float t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = fminf(fmaxf(v[i] * inv, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w, true);
  return (uint32_t)w;
}
// 4 consecutive elements per lane, the 8 lanes 8k .. 8k+7 form a block (LayerNorm rows)
__device__ __forceinline__ uint32_t mx8_encode4(const float (&v)[4], uint32_t* scale) {
  float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  am = fmaxf(am, __shfl_xor(am, 1, 64));
  am = fmaxf(am, __shfl_xor(am, 2, 64));
  am = fmaxf(am, __shfl_xor(am, 4, 64));
  float inv;
  *scale = mx8_scale_byte(am, &inv);
  return mx8_pack4(v, inv);
}

// MX-FP8 block encoder (OCP microscaling v1.0, e4m3 elements): the 4 lanes 4k .. 4k+3 hold 8 consecutive elements
// each of one 32-element block and must all be active.  scale = E8M0 byte of 2^(floor(log2 amax) - 8); elements are
// x * 2^-(scale-127), clamped to +-448, rounded to nearest even by v_cvt_pk_fp8_f32.
struct MxBlock {
  uint2 q;         // this lane's 8 e4m3 bytes
  uint32_t scale;  // the block's E8M0 byte (same on the 4 lanes)
};
__device__ __forceinline__ MxBlock mx8_encode(const float (&v)[8]) {
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(v[i]));
  // the other three lanes of the quad through DPP (quad_perm [1,0,3,2], [2,3,0,1]): VALU only
  am = fmaxf(am, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(am), 0xB1, 0xF, 0xF, true)));
  am = fmaxf(am, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(am), 0x4E, 0xF, 0xF, true)));
  float inv;
  MxBlock b;
  b.scale = mx8_scale_byte(am, &inv);  // biased exponent of the block maximum - 8; inv = 2^(127 - scale)
  const float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
  b.q = make_uint2(mx8_pack4(lo, inv), mx8_pack4(hi, inv));
  return b;
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16>(bf16 v) { return bf16_bits_to_f32(v.x); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) { bf16 r; r.x = f32_to_bf16_bits(v); return r; }

// 4 consecutive elements
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 load4<bf16>(const bf16* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}
template <typename T>
__device__ __forceinline__ void store4(T* p, float4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <>
__device__ __forceinline__ void store4<bf16>(bf16* p, float4 v) {
  uint2 r;
  r.x = pack_bf16x2(v.x, v.y);
  r.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = r;
}

// ---------------------------------------------------------------------------------------------
// wave (64 lanes) reductions
// ---------------------------------------------------------------------------------------------
// Wave-wide reductions on the VALU: four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2] and [2,3,0,1], row_ror 4
// and 8), then v_permlane16_swap / v_permlane32_swap across the four rows.  (__shfl_xor is a ds_bpermute - an LDS round trip
// - and six dependent ones per reduction were most of a LayerNorm row's latency.)  Every lane ends with the same value.
#define AVF_DPP_F32(v, ctrl) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float wave_sum(float v) {
  v += AVF_DPP_F32(v, 0xB1);
  v += AVF_DPP_F32(v, 0x4E);
  v += AVF_DPP_F32(v, 0x124);
  v += AVF_DPP_F32(v, 0x128);
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, AVF_DPP_F32(v, 0xB1));
  v = fmaxf(v, AVF_DPP_F32(v, 0x4E));
  v = fmaxf(v, AVF_DPP_F32(v, 0x124));
  v = fmaxf(v, AVF_DPP_F32(v, 0x128));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// tanh-GELU exactly as models/heads.py:166 and its derivative
__device__ __forceinline__ float gelu_tanh_f(float u) {
  const float c = 0.7978845608028654f;  // sqrt(2/pi)
  float inner = c * (u + 0.044715f * u * u * u);
  return 0.5f * u * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float dgelu_tanh_f(float u) {
  const float c = 0.7978845608028654f;
  float u2 = u * u;
  float t = tanhf(c * (u + 0.044715f * u * u2));
  return 0.5f * (1.0f + t) + 0.5f * u * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * u2);
}

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
// fast forms for the bf16 throughput path: tanh(z) = 1 - 2/(1 + 2^(2 z log2 e)) on v_exp_f32 / v_rcp_f32
// (|error| ~1e-6, far below one bf16 ulp)
__device__ __forceinline__ float fast_tanh(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.885390081777927f);  // 2*log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
// tanh-GELU through the logistic form: 0.5 u (1 + tanh z) = u * sigma(2 z), 2 z log2(e) = u (K1 + K2 u^2) - 13 VALU
// slots per element instead of 18 (the transcendental pair exp2 / rcp is 8 of them); same value to fp32 rounding
constexpr float GELU_K1 = 2.302208198f;    // 2 sqrt(2/pi) log2(e)
constexpr float GELU_K2 = 0.1029432396f;   // K1 * 0.044715
__device__ __forceinline__ float gelu_tanh_fast(float u) {
  const float e = __builtin_amdgcn_exp2f(-u * fmaf(GELU_K2, u * u, GELU_K1));
  return u * __builtin_amdgcn_rcpf(1.0f + e);
}
// d/du [u sigma(2z)] = s + u s (1 - s) d(2z)/du,  d(2z)/du = 2 sqrt(2/pi) (1 + 3 * 0.044715 u^2)
__device__ __forceinline__ float dgelu_tanh_fast(float u) {
  const float u2 = u * u;
  const float e = __builtin_amdgcn_exp2f(-u * fmaf(GELU_K2, u2, GELU_K1));
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  const float q = fmaf(0.2140644488f, u2, 1.5957691216f);  // 6 c a, 2 c  (c = sqrt(2/pi), a = 0.044715)
  return fmaf(u * s * (1.0f - s), q, s);
}

// ---------------------------------------------------------------------------------------------
// dropout (nn.Dropout at heads.py:194,196,216): counter-based, stateless.  One splitmix64 hash per group of 4
// consecutive elements gives four 16-bit uniforms; an element is dropped when its uniform < thresh16 = round(p*65536)
// and kept elements are scaled by 1/(1-p_eff).  The mask of element (row, col) of a site depends only on
// (seed, layer, site, row*ld + col), so forward and backward regenerate identical masks without storing them.
// ---------------------------------------------------------------------------------------------
struct DropCfg {
  uint64_t key;              // mix64(seed + site_const) for a host-side seed; meaningless when thresh16 == 0
  const uint64_t* seed_dev;  // optional: the seed lives in device memory (graph-replayable); key is then derived
  uint64_t site_const;       // in the kernel as mix64(*seed_dev + site_const)
  uint32_t thresh16;         // 0 = dropout disabled
  float scale;
};
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
// sites of one layer: 0 = after to_out (heads.py:216), 1 = after GELU (heads.py:194), 2 = after net.3 (heads.py:196)
inline DropCfg make_drop(float p, uint64_t seed, int layer, int site, const uint64_t* seed_dev = nullptr) {
  DropCfg d;
  uint32_t t = (uint32_t)(p * 65536.0f + 0.5f);
  if (t > 65535u) t = 65535u;
  d.thresh16 = p > 0.f ? t : 0u;
  d.scale = 1.0f / (1.0f - (float)d.thresh16 / 65536.0f);
  d.site_const = 0xD1B54A32D192ED03ULL * (uint64_t)(layer * 4 + site + 1);
  d.key = mix64(seed + d.site_const);
  d.seed_dev = seed_dev;
  return d;
}
static const DropCfg kNoDrop = {0, nullptr, 0, 0, 1.0f};
// the mask key of a site: from the device-resident seed when there is one (wave-uniform scalar load), else the host's
__device__ __forceinline__ uint64_t drop_key(const DropCfg& d) {
  return d.seed_dev ? mix64(*d.seed_dev + d.site_const) : d.key;
}
// keep*scale factors of the 4 elements starting at linear index idx (idx % 4 == 0)
__device__ __forceinline__ float4 drop_factor4(const DropCfg& d, uint64_t key, uint64_t idx) {
  const uint64_t h = mix64(key + idx * 0x9E3779B97F4A7C15ULL);
  float4 f;
  f.x = ((uint32_t)(h) & 0xffffu) >= d.thresh16 ? d.scale : 0.f;
  f.y = ((uint32_t)(h >> 16) & 0xffffu) >= d.thresh16 ? d.scale : 0.f;
  f.z = ((uint32_t)(h >> 32) & 0xffffu) >= d.thresh16 ? d.scale : 0.f;
  f.w = ((uint32_t)(h >> 48) & 0xffffu) >= d.thresh16 ? d.scale : 0.f;
  return f;
}

// the factor of the single element idx (the same hash word as its group of four)
__device__ __forceinline__ float drop_factor1(const DropCfg& d, uint64_t key, uint64_t idx) {
  const uint64_t h = mix64(key + (idx & ~3ULL) * 0x9E3779B97F4A7C15ULL);
  return ((uint32_t)(h >> (16 * (idx & 3))) & 0xffffu) >= d.thresh16 ? d.scale : 0.f;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// "dynamic-LDS limit raised" bookkeeping, one bit per device (hipFuncSetAttribute is per device; a process that runs stacks
// on a second GPU must raise it there too).  Racing threads (forward thread, autograd's device thread) may both call
// hipFuncSetAttribute - idempotent - and the bit is set with an atomic OR.
struct PerDeviceOnce {
  uint64_t done = 0;
  bool need() const {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return ((__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> (dev & 63)) & 1u) == 0;
  }
  void mark() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    __atomic_fetch_or(&done, 1ull << (dev & 63), __ATOMIC_RELEASE);
  }
};

// ---------------------------------------------------------------------------------------------
// internal launchers shared between translation units (all return 0 / non-zero)
// ---------------------------------------------------------------------------------------------
struct FoldJob;
// mx_q / mx_s (optional, bf16 output only): also write the MX-FP8 image of y ([rows,dim] e4m3 + [rows,dim/32] E8M0)
// x_dtype AVF_BF16 (bf16 residual stream): x is read as bf16 (dim % 4 == 0, dim <= 1536)
int layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                  float* rstd, int64_t rows, int dim, float eps, hipStream_t s, void* mx_q = nullptr, void* mx_s = nullptr,
                  int x_dtype = AVF_F32);
size_t layernorm_bwd_ws(int64_t rows, int dim);
// drop: mask applied to the bf16 copy dx_lo AND to the column sums (they feed the Linear behind a dropout site);
// dx itself (the residual stream gradient) is never masked.
// dres_dtype AVF_BF16 (bf16 gradient stream): dres is read as bf16, dx may be null (dx_lo is then the only output).
// x_dtype AVF_BF16 (bf16 residual stream): the saved LayerNorm input is bf16 (bf16 dy, dim % 4 == 0, dim <= 1536)
int layernorm_bwd(const void* dy, int dy_dtype, const void* x, const float* gamma, const float* mean,
                  const float* rstd, const void* dres, float* dx, void* dx_lo, float* dgamma, float* dbeta,
                  float* dcolsum, void* ws, int64_t rows, int dim, hipStream_t s, const DropCfg& drop = kNoDrop,
                  FoldJob* defer_fold = nullptr, int dres_dtype = AVF_F32, int x_dtype = AVF_F32, void* mx_q = nullptr,
                  void* mx_s = nullptr, void* dx_m = nullptr);  // mx_q / mx_s: also the MX-FP8 image of the values written to dx_lo
size_t colsum_ws(int64_t rows, int cols);
int colsum(const void* in, int in_dtype, int64_t rows, int cols, int64_t ld, float* out, void* ws, hipStream_t s);
int cast_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t s, const DropCfg& drop = kNoDrop);
// out = in * dropout factors (fp32 parity mode: the gradient a Linear behind a dropout site sees); n % 4 == 0
int mask_copy_f32(const float* in, float* out, int64_t n, hipStream_t s, const DropCfg& drop);
int dropout_factors(const DropCfg& drop, float* out, int64_t n, hipStream_t s);  // test aid: keep*scale per element
// out[c] = sum_b partial[b][c]
int fold_partials(const float* partial, int nb, int width, float* out, hipStream_t s);
struct FoldJob;
int fold_job(const FoldJob& job, hipStream_t s);  // one FoldJob (up to three output segments) as its own launch

// A deferred column fold: out_k[c] = sum_b partial[b][k*seg + c] for the (up to 3) segments of width/seg.
// Kernels that produce per-block partial sums can hand the fold back to the caller, which runs all folds of a layer
// in ONE launch together with the split-K slab fold of the weight gradients (fewer launches: each costs ~6 us here).
struct FoldJob {
  const float* partial;
  int nb, width, seg;
  float *o0, *o1, *o2;
};
struct FoldList {
  FoldJob job[3];
  int count;
};
// 16 columns (4 quads) x 64 row groups per 256-thread block, eight 16-byte loads per thread in flight; width % 4 == 0, 16-byte
// aligned partials.  (Round 4: 32 columns x 32 row groups with four loads in flight put 48 blocks on the 1536 columns of a
// LayerNorm fold, each walking 1024 partial rows with 16 KiB in flight - latency-bound at ~8 GB/s per CU, 12 of the 19.5 us of
// the layer's fold launch at C3.  Half the columns per block and twice the loads in flight: four times the bytes in flight.)
int fold_list(const FoldList& fl, hipStream_t s);  // the (up to three) jobs of a list in one launch
constexpr int FOLD_COLS = 16, FOLD_RG = 64;
__device__ __forceinline__ void fold_columns_vec(const FoldJob& j, int colgroup, float4 (*red)[FOLD_COLS / 4]) {
  const int cq = threadIdx.x & (FOLD_COLS / 4 - 1), grp = threadIdx.x / (FOLD_COLS / 4);
  const int col = colgroup * FOLD_COLS + cq * 4;
  float4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < j.width) {
    int b = grp;
    for (; b + 7 * FOLD_RG < j.nb; b += 8 * FOLD_RG) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(j.partial + (int64_t)(b + FOLD_RG * u) * j.width + col);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u].x += t[u].x; a[u].y += t[u].y; a[u].z += t[u].z; a[u].w += t[u].w; }
    }
    for (; b < j.nb; b += FOLD_RG) {
      const float4 t = *reinterpret_cast<const float4*>(j.partial + (int64_t)b * j.width + col);
      a[0].x += t.x; a[0].y += t.y; a[0].z += t.z; a[0].w += t.w;
    }
  }
  red[grp][cq] = make_float4(((a[0].x + a[1].x) + (a[2].x + a[3].x)) + ((a[4].x + a[5].x) + (a[6].x + a[7].x)),
                             ((a[0].y + a[1].y) + (a[2].y + a[3].y)) + ((a[4].y + a[5].y) + (a[6].y + a[7].y)),
                             ((a[0].z + a[1].z) + (a[2].z + a[3].z)) + ((a[4].z + a[5].z) + (a[6].z + a[7].z)),
                             ((a[0].w + a[1].w) + (a[2].w + a[3].w)) + ((a[4].w + a[5].w) + (a[6].w + a[7].w)));
  __syncthreads();
  if (threadIdx.x < FOLD_COLS) {  // thread -> (quad, component)
    const int q = threadIdx.x >> 2, comp = threadIdx.x & 3;
    const int c = colgroup * FOLD_COLS + threadIdx.x;
    if (c < j.width) {
      float v = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < FOLD_RG; ++g2) v += reinterpret_cast<const float*>(&red[g2][q])[comp];
      const int which = c / j.seg, cc = c - which * j.seg;
      float* dst = which == 0 ? j.o0 : (which == 1 ? j.o1 : j.o2);
      if (dst) dst[cc] = v;
    }
  }
  __syncthreads();
}
// workspace bytes for the fused column-sum partials of an NT GEMM with M rows and N columns
size_t gemm_nt_colsum_ws(int64_t M, int64_t N);
int prep_weight_bf16(const float* w, void* w_lo, void* w_t_lo, int rows, int cols, hipStream_t s);
// lo_scale / lo_scaled_rows: the first rows of the row-major image `lo` (NOT of the transposed image) are multiplied by a
// constant - the query rows of Wqkv carry the softmax scale (see attn_q_prescale in attn_bf16.hip)
// lo_p / t_p (optional): the fragment-major images (pack_ws_off) of lo (C == 512) / of t (R == 512) for the weight-stationary GEMM
struct PrepDesc { const float* w; bf16* lo; bf16* t; int R, C; float lo_scale; int lo_scaled_rows; void* lo_p = nullptr; void* t_p = nullptr; };
struct PrepBatch { PrepDesc d[4]; };
int prep_weights_multi(const PrepBatch& b, int count, hipStream_t s);

struct GemmArgs {
  int dtype, transA, transB;
  int64_t M, N, K;
  const void* A;
  int64_t lda;
  const void* B;
  int64_t ldb;
  void* C;
  int64_t ldc;
  int c_dtype;
  int epilogue;
  const float* bias;
  const void* residual;  // BIAS_RES: fp32 when C is fp32; bf16 when C is bf16 (bf16 residual stream)
  int64_t ldres;
  void* aux;
  int64_t ldaux;
  void* workspace;
  float* colsum;  // optional: column sums of the stored C (bf16 NT only; partials go to workspace)
  DropCfg drop;   // bf16 NT only: BIAS_RES masks (acc+bias) before the residual, BIAS_GELU masks gelu(u), DGELU masks acc
  FoldJob* defer_fold;  // with colsum: do not launch the fold, describe it here instead
  // bf16 NT only, optional: the fragment-major image of B (pack_ws; K == 512, N % 256 == 0) - the GEMM then runs on the
  // weight-stationary persistent kernel (gemm_ws.hip) when the shape qualifies
  const void* Bp = nullptr;
  int ws_force = 0;  // 1: take that kernel whenever it can run the shape (avf_gemm_nt_ws); 0: where it is the faster one
  // ... and on that kernel only (BIAS_GELU / DGELU, N % 32 == 0): also the MX-FP8 image of C (e4m3 bytes [M][N], E8M0 scale
  // bytes [M][N / 32]) - the fp8 mode's dGELU GEMM keeps bf16 operands there and still feeds the fp8 GEMM behind it
  void* mx_q = nullptr;
  void* mx_s = nullptr;
};
// weight-stationary persistent NT GEMM (gemm_ws.hip; DESIGN_HISTORY.md section 18)
// byte offset of element (n, k) of a weight [N][512] in its fragment-major image: 1 KiB pieces (panel of 256 rows, wave's 32
// rows, 16-row block j, k-step s of 32), inside a piece lane (li = n % 16, lg = (k % 32) / 8) owns 16 bytes = 8 consecutive k
__host__ __device__ __forceinline__ size_t pack_ws_off(int n, int k) {
  const size_t piece = (size_t)((((n >> 8) * 8 + ((n >> 5) & 7)) * 2 + ((n >> 4) & 1)) * 16 + (k >> 5));
  return piece * 1024 + (size_t)((((k >> 3) & 3) * 16 + (n & 15)) * 16 + (k & 7) * 2);
}
// the fragment-major images a layer keeps behind its bf16 weight images (null where the shape does not qualify):
// Wqkv [3I, D], Wo [D, I], W1 [M, D] (forward) and W2^T [M, D], Wo^T [I, D] (the dX GEMMs with a 512-deep reduction)
struct LowpWs {
  void *wqkv_p, *wo_p, *w1_p, *w2t_p, *wot_p;
};
int lowp_ws_images(const avf_layer_cfg* cfg, void* lowp, LowpWs* out);
size_t pack_ws_bytes(int64_t N, int64_t K);
bool pack_ws_ok(int64_t N, int64_t K);
int pack_ws(const void* w_bf16, int64_t ldw, int64_t N, int64_t K, void* out, hipStream_t s);
bool gemm_bf16_nt_ws_ok(const GemmArgs& a);
bool gemm_bf16_nt_ws_preferred(const GemmArgs& a);
int gemm_bf16_nt_ws(const GemmArgs& a, hipStream_t s, int* part_rows_out);
size_t gemm_ws(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K);
int gemm(const GemmArgs& a, hipStream_t s);
int gemm_f32(const GemmArgs& a, hipStream_t s);
size_t gemm_f32_ws(int64_t M, int64_t N, int64_t K);  // split-K slabs of skinny shapes (0: no split)
// arithmetic of the AVF_F32 mode (round 6): 1 = operands split in three bf16 products on the bf16 matrix pipe ("bf16x3", default),
// 0 = the f32-input MFMA (v_mfma_f32_16x16x4_f32).  Process-wide; avf_set_f32_arith / avf_get_f32_arith.
void set_f32_arith(int mode);
int get_f32_arith();
int gemm_bf16_nt(const GemmArgs& a, hipStream_t s);
int gemm_bf16_tn(const GemmArgs& a, hipStream_t s);
// MX-FP8 NT GEMM (gemm_mx8.hip): A, B are e4m3 byte images (lda, ldb in bytes), scales [rows][K/32] E8M0 bytes
// mx_q / mx_s (optional, BIAS_GELU only): also write the MX-FP8 image of C ([M,N] e4m3 + [M,N/32] E8M0)
int gemm_mx8_nt(const GemmArgs& a, const void* a_scales, const void* b_scales, hipStream_t s, void* mx_q = nullptr,
                void* mx_s = nullptr);
struct MxQuantJob {
  const void* x;  // bf16 [R,K]
  void* q;        // e4m3 [R,K]
  void* s;        // E8M0 [R,K/32]
  int64_t R, K;
};
int quant_mx8_multi(const MxQuantJob* jobs, int n, hipStream_t s);
int quant_mx8(const void* x, int dtype, int64_t ldx, int64_t R, int64_t K, void* q, int64_t ldq, void* scales, hipStream_t s);
size_t gemm_bf16_tn_ws(int64_t M, int64_t N, int64_t K);
// up to four C_i[M_i,N_i] = A_i[K,M_i]^T B_i[K,N_i] sharing K, one launch (weight gradients of one layer)
struct TnGroupArgs {
  int count;
  int64_t K;
  const void* A[4];
  const void* B[4];
  float* C[4];
  int64_t M[4], N[4], lda[4], ldb[4];
  void* workspace;
};
bool gemm_bf16_tn_group_ok(const TnGroupArgs& a);
size_t gemm_bf16_tn_group_ws(const TnGroupArgs& a);
int gemm_bf16_tn_group(const TnGroupArgs& a, hipStream_t s, const FoldList* extra_folds = nullptr);
// the same group for fp32 operands in the parity mode's bf16x3 arithmetic (gemm_f32.hip, round 6): one launch + one fold
bool gemm_f32x3_tn_group_ok(const TnGroupArgs& a);
size_t gemm_f32x3_tn_group_ws(const TnGroupArgs& a);
int gemm_f32x3_tn_group(const TnGroupArgs& a, hipStream_t s);

int attn_fwd_f32(const float* qkv, float* o, float* lse2, int B, int N, int H, int dh, hipStream_t s);
// fp32-arithmetic attention on fp32 / bf16 storage with the optional token mask keep [B, N] (bytes, 1 = kept; heads.py:225-232)
// parity-mode attention on the fp32 matrix pipe (attn_f32_mfma.hip): fp32 storage, no mask, dim_head 32 / 64
bool attn_f32_mfma_ok(int dtype, int dh, const void* keep, int H, const void* qkv, const void* other);
int attn_fwd_f32_mfma(const float* qkv, float* o, float* lse2, int B, int N, int H, int dh, hipStream_t s, bool q_prescaled);
int attn_bwd_f32_mfma(const float* qkv, const float* d_o, const float* lse2, const float* delta, float* dqkv, int B, int N, int H,
                      int dh, hipStream_t s, bool q_prescaled);
// parity-mode attention with three bf16 products per fp32 product (attn_f32x3.hip): fp32 storage, no mask, dim_head 64, raw q
bool attn_f32x3_ok(int dtype, int dh, const void* keep, int H, const void* qkv, const void* other, bool q_prescaled);
int attn_fwd_f32x3(const float* qkv, float* o, float* lse2, int B, int N, int H, hipStream_t s);
int attn_bwd_f32x3(const float* qkv, const float* o, const float* d_o, const float* lse2, float* delta, float* dqkv, int B, int N,
                   int H, hipStream_t s);  // (computes delta itself: no attn_delta launch in front of it)
int attn_fwd_vec(int dtype, const void* qkv, void* o, float* lse2, int B, int N, int H, int dh, hipStream_t s, const void* keep,
                 bool q_prescaled);
int attn_bwd_vec(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, float* delta,
                 int B, int N, int H, int dh, hipStream_t s, const void* keep, bool q_prescaled);
int attn_bwd_f32(const float* qkv, const float* o, const float* d_o, const float* lse2, float* dqkv, float* delta,
                 int B, int N, int H, int dh, hipStream_t s);
// q_prescaled: the q columns of qkv already hold q * log2(e)/sqrt(dh) (the layer path: folded into the bf16 copy of
// Wqkv's query rows, attn_q_prescale); the operator-level C entry points pass false.  nlse (backward, q_prescaled
// only): B*H*N floats of scratch next to delta.
// single-launch forward of one layer for short sequences (layer_small.hip)
bool small_layer_ok(int dtype, int tokens, int dim, int heads, int dim_head, int mlp_dim);
int layer_fwd_small(int B, int N, int D, int H, int M, float eps, float score_scale, const avf_layer_params* p,
                    const void* wqkv, const void* wo, const void* w1, const void* w2, const float* x_in, float* x_out,
                    void* h1, float* mean1, float* rstd1, void* qkv, void* o, float* lse2, float* x_mid, void* h2,
                    float* mean2, float* rstd2, void* u, void* g, const DropCfg& dr0, const DropCfg& dr1,
                    const DropCfg& dr2, hipStream_t s);
// the fused backward pieces of the short-sequence layer (layer_small.hip): host-side argument blocks
struct SmallBwdAHost {
  const float* dx_out; const void* dx_out_lo; void* gy_store; const float* x_mid; const void* u;
  const float *ln2_w, *mean2, *rstd2; const void *w2_t, *w1_t, *wo_t; void* du; float* dx_mid; void* dx_mid_lo; void* d_o;
  float *pb1, *pln2; int gs16; DropCfg dr0, dr1, dr2;
};
struct SmallBwdBHost {
  int attention;  // 1: the clip's attention backward runs inside the kernel (qkv / o / d_o / lse2 in, dqkv out)
  const void *qkv, *o, *d_o; const float* lse2; float score_scale, dq_scale, dk_scale; int H;
  void* dqkv; const void* wqkv_t; const float* x_in; const float *ln1_w, *mean1, *rstd1; const float* dx_mid; const void* dx_mid_lo;
  float* dx_in; void* dx_in_lo; float* pln1; int gs16; DropCfg dr_prev2;
};
size_t small_bwd_partial_floats(int B, int D, int M);
int layer_bwd_small_a(int B, int N, int D, int I, int M, const SmallBwdAHost& h, hipStream_t s);
int layer_bwd_small_b(int B, int N, int D, int I, const SmallBwdBHost& h, hipStream_t s);
float attn_q_prescale(int dh);
bool attn_q_prescale_on();
// keep (optional, [B][N] bytes): the token mask of heads.py:225-232 on the MFMA kernels - attn_masked_bf16_ok says where
int attn_fwd_bf16(const bf16* qkv, bf16* o, float* lse2, int B, int N, int H, int dh, hipStream_t s,
                  bool q_prescaled = false, void* mx_q = nullptr, void* mx_s = nullptr, const void* keep = nullptr);
bool attn_masked_bf16_ok(int N, int dh, bool q_prescaled);  // head-resident forward + merged backward: dh 64, N <= 512, pre-scaled q
bool attn_fwd_emits_mx8(int N, int dh);  // mx_q / mx_s: MX-FP8 image of o, written by the head-resident kernel only
int attn_bwd_bf16(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv, float* delta,
                  int B, int N, int H, int dh, hipStream_t s, bool q_prescaled = false, float* nlse = nullptr,
                  const void* keep = nullptr, void* dq_q = nullptr, void* dq_s = nullptr);
// dq_q / dq_s: MX-FP8 image of dqkv ([B N][3 I] e4m3 + [B N][3 I / 32] E8M0), written by the merged kernel only
bool attn_bwd_emits_mx8(int N, int dh, bool q_prescaled);
int attn_delta(int dtype, const void* o, const void* d_o, float* delta, int B, int N, int H, int dh, hipStream_t s);
// merged dQ + dK/dV kernel (attn_bwd_merged.hip): dim_head 64, pre-scaled q, N <= 512
bool attn_bwd_merged_ok(int N, int dh, bool q_prescaled);
int attn_bwd_merged(const TimingScope* ts, const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse2, bf16* dqkv,
                    int B, int N, int H, hipStream_t s, const void* keep = nullptr, void* dq_q = nullptr, void* dq_s = nullptr);

}  // namespace avf
