// ln_fuse.hip - the small kernels around the LayerNorm-folded GEMMs (gemm_nt.hpp, NtParams::ln_*).
//
// Reference: PreNorm (models/heads.py:178-185): fn(LayerNorm(x)).  With fn's first operation an nn.Linear,
//   LayerNorm(x) W^T + b = rstd (x (gamma o W)^T - mean s) + c,   s[n] = sum_k gamma[k] W[n][k],  c[n] = beta . W[n] + b[n],
// so the GEMM can read the raw residual stream and the normalised rows are never written in forward.
//   row_stats        : per-row partial (sum, sum of squares) over groups of 32 columns - the form the residual GEMM epilogues
//                      emit - for a LayerNorm input that no such epilogue produced (the first layer of a stack)
//   ln_fold_weights  : W'[n][k] = bf16(scale[n] gamma[k] W[n][k]), s[n] = sum_k W'[n][k] (of the ROUNDED image, so that
//                      mean * s cancels the mean part of the accumulated product exactly), c[n] = scale[n] beta . W[n] + b[n]
#include "common.hpp"

namespace avf {

namespace {

template <typename XT>
__global__ __launch_bounds__(256) void row_stats_kernel(const XT* __restrict__ x, float* __restrict__ part, int64_t rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int np = D >> 5;
  for (int c0 = 0; c0 < D; c0 += 256) {  // a wave covers 256 columns per step: 8 lanes x 4 values = one group of 32
    const int c = c0 + lane * 4;
    float s1 = 0.f, s2 = 0.f;
    if (c < D) {
      const float4 v = load4<XT>(x + row * D + c);
      s1 = (v.x + v.y) + (v.z + v.w);
      s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    s1 += __shfl_xor(s1, 4, 64); s2 += __shfl_xor(s2, 4, 64);
    if ((lane & 7) == 0 && c < D) *reinterpret_cast<float2*>(part + (row * np + (c >> 5)) * 2) = make_float2(s1, s2);
  }
}

constexpr int LNF_MAX_JOBS = 48;  // 24 layers per launch: the table travels as a kernel argument (3.6 KB)
struct LnFoldBatch {
  LnFoldJob job[LNF_MAX_JOBS];
  int first_row[LNF_MAX_JOBS + 1];  // prefix sums of job rows
  int count;
};

// one wavefront per output row (4 rows per block)
__global__ __launch_bounds__(256) void ln_fold_kernel(const LnFoldBatch b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gr = blockIdx.x * 4 + wave;
  if (gr >= b.first_row[b.count]) return;
  int j = 0;
  while (gr >= b.first_row[j + 1]) ++j;
  const LnFoldJob& J = b.job[j];
  const int n = gr - b.first_row[j];
  const float sc = n < J.lo_scaled_rows ? J.lo_scale : 1.0f;
  const float* wrow = J.w + (int64_t)n * J.dim;
  float s = 0.f, c = 0.f;
  for (int k = lane * 4; k < J.dim; k += 256) {
    const float4 w = *reinterpret_cast<const float4*>(wrow + k);
    const float4 g = *reinterpret_cast<const float4*>(J.gamma + k);
    const float4 be = *reinterpret_cast<const float4*>(J.beta + k);
    const uint32_t p0 = pack_bf16x2(sc * g.x * w.x, sc * g.y * w.y), p1 = pack_bf16x2(sc * g.z * w.z, sc * g.w * w.w);
    *reinterpret_cast<uint2*>(J.w_ln + (int64_t)n * J.dim + k) = make_uint2(p0, p1);
    s += (__uint_as_float(p0 << 16) + __uint_as_float(p0 & 0xffff0000u)) + (__uint_as_float(p1 << 16) + __uint_as_float(p1 & 0xffff0000u));
    c += (be.x * w.x + be.y * w.y) + (be.z * w.z + be.w * w.w);
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if (lane == 0) {
    J.s[n] = s;
    J.c[n] = sc * c + (J.bias ? J.bias[n] : 0.f);
  }
}

}  // namespace

int row_stats(const void* x, int x_dtype, int64_t rows, int dim, float* part, hipStream_t s) {
  AVF_REQUIRE(x && part && rows > 0 && dim > 0 && dim % 32 == 0, "row_stats: bad arguments (dim %% 32 == 0)");
  AVF_REQUIRE(ceil_div(rows, 4) < (1LL << 31), "row_stats: too many rows");
  TimingScope ts(KC_LAYERNORM, 0.0, (double)rows * dim * (x_dtype == AVF_BF16 ? 2.0 : 4.0), s, /*per_kernel=*/true);
  const dim3 grid((unsigned)ceil_div(rows, 4));
  if (x_dtype == AVF_BF16) launch_in_scope(&ts, row_stats_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, part, rows, dim);
  else if (x_dtype == AVF_F32) launch_in_scope(&ts, row_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x, part, rows, dim);
  else AVF_REQUIRE(false, "row_stats: bad dtype %d", x_dtype);
  return check_launch("row_stats_kernel");
}

int ln_fold_weights(const LnFoldJob* jobs, int count, hipStream_t s) {
  AVF_REQUIRE(jobs && count > 0, "ln_fold_weights: no jobs");
  for (int base = 0; base < count; base += LNF_MAX_JOBS) {
    LnFoldBatch b;
    b.count = count - base < LNF_MAX_JOBS ? count - base : LNF_MAX_JOBS;
    b.first_row[0] = 0;
    double bytes = 0.0;
    for (int i = 0; i < b.count; ++i) {
      const LnFoldJob& j = jobs[base + i];
      AVF_REQUIRE(j.w && j.gamma && j.beta && j.w_ln && j.s && j.c && j.rows > 0 && j.dim > 0 && j.dim % 4 == 0 &&
                      ((uintptr_t)j.w & 15) == 0 && ((uintptr_t)j.w_ln & 7) == 0,
                  "ln_fold_weights: bad job %d", base + i);
      b.job[i] = j;
      b.first_row[i + 1] = b.first_row[i] + j.rows;
      bytes += 6.0 * j.rows * j.dim;
    }
    TimingScope ts(KC_OTHER, 0.0, bytes, s, /*per_kernel=*/true);
    launch_in_scope(&ts, ln_fold_kernel, dim3((unsigned)ceil_div(b.first_row[b.count], 4)), dim3(256), 0, s, b);
    AVF_TRY(check_launch("ln_fold_kernel"));
  }
  return 0;
}

}  // namespace avf
