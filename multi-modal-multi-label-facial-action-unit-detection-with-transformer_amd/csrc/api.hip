// api.hip - extern "C" surface of libavformer_hip.so (see include/avformer_hip.h), error plumbing,
// and the hardware self-tests that pin the MFMA fragment maps / transposed LDS read semantics the
// bf16 kernels rely on.
#include <stdarg.h>

#include "common.hpp"

namespace avf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// event timing
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int TIMING_POOL = 32768;
struct TimingRec { hipEvent_t start, stop; int cls; double flops, bytes; };
bool g_timing_on = false;
TimingRec* g_recs = nullptr;
int g_created = 0, g_used = 0;
}  // namespace

namespace {
int timing_take_slot(int cls, double flops, double bytes) {
  if (!g_timing_on || g_used >= TIMING_POOL) return -1;
  if (!g_recs) g_recs = new TimingRec[TIMING_POOL];
  if (g_used >= g_created) {
    if (hipEventCreate(&g_recs[g_created].start) != hipSuccess || hipEventCreate(&g_recs[g_created].stop) != hipSuccess) return -1;
    ++g_created;
  }
  const int slot = g_used++;
  g_recs[slot].cls = cls; g_recs[slot].flops = flops; g_recs[slot].bytes = bytes;
  return slot;
}
}  // namespace

static FILE* shape_log_file() {
  static FILE* f = [] {
    const char* e = getenv("AVF_SHAPE_LOG");
    return (e && *e) ? fopen(e, "a") : (FILE*)nullptr;
  }();
  return f;
}
bool shape_log_on() { return shape_log_file() != nullptr; }
void shape_log(const char* fmt, ...) {
  FILE* f = shape_log_file();
  if (!f) return;
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fputc('\n', f);
  fflush(f);
}

TimingScope::TimingScope(int c, double f, double b, hipStream_t s, bool pk)
    : slot(-1), stream(s), per_kernel(pk), cls(c), flops(f), bytes(b), issued(0) {
  if (per_kernel) return;  // records are taken per launch
  slot = timing_take_slot(cls, flops, bytes);
  if (slot >= 0) (void)hipEventRecord(g_recs[slot].start, s);
}
TimingScope::~TimingScope() {
  if (slot >= 0 && !per_kernel) (void)hipEventRecord(g_recs[slot].stop, stream);
}
bool TimingScope::events(hipEvent_t* start, hipEvent_t* stop) const {
  if (!per_kernel) return false;
  const int sl = timing_take_slot(cls, issued ? 0.0 : flops, issued ? 0.0 : bytes);
  if (sl < 0) return false;
  ++issued;
  *start = g_recs[sl].start;
  *stop = g_recs[sl].stop;
  return true;
}

// ---------------------------------------------------------------------------------------------
// self-tests
// ---------------------------------------------------------------------------------------------
// C[16x16] = A[16x32] * B[32x16] with one v_mfma_f32_16x16x32_bf16, operands fetched with the
// lane maps the kernels use: A[i=l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][n=l&15]; C col=l&15,row=4(l>>4)+r.
__global__ void selftest_mfma_bf16_kernel(const bf16* a, const bf16* b, float* c) {
  const int l = threadIdx.x, li = l & 15, lg = l >> 4;
  bf16x8_t fa, fb;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    fa[j] = __builtin_bit_cast(__bf16, a[li * 32 + 8 * lg + j].x);
    fb[j] = __builtin_bit_cast(__bf16, b[(8 * lg + j) * 16 + li].x);
  }
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[(4 * lg + r) * 16 + li] = acc[r];
}

__global__ void selftest_mfma_f32_kernel(const float* a, const float* b, float* c) {
  const int l = threadIdx.x, li = l & 15, lg = l >> 4;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[li * 4 + lg], b[lg * 16 + li], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[(4 * lg + r) * 16 + li] = acc[r];
}

// ds_read_b64_tr_b16 on a [32 rows][16 cols] bf16 tile (32-byte rows): lane l (i=l&15, g=l>>4) issues two
// reads with row blocks 4g.. (h=0) and 16+4g.. (h=1), address = &T[blk + (i>>2)][4*(i&3)], and stores
// its 8 received values: out[l][h*4 + e].  Expected (kernels' assumption): out[l][h*4+e] = T[16h+4g+e][i].
__global__ void selftest_tr16_kernel(const bf16* tile, bf16* out) {
  __shared__ __attribute__((aligned(16))) bf16 T[32 * 16];
  const int l = threadIdx.x, li = l & 15, lg = l >> 4;
  for (int i = l; i < 32 * 16; i += 64) T[i] = tile[i];
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const bf16* src = &T[(16 * h + 4 * lg + (li >> 2)) * 16 + 4 * (li & 3)];
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)src);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[l * 8 + h * 4 + e].x = (uint16_t)v[e];
  }
}

}  // namespace avf

using namespace avf;

extern "C" int avf_version(void) { return 1; }
extern "C" size_t avf_sizeof_layer_cfg(void) { return sizeof(avf_layer_cfg); }
extern "C" size_t avf_sizeof_layer_params(void) { return sizeof(avf_layer_params); }

extern "C" int avf_timing_enable(int on) {
  g_timing_on = on != 0;
  if (on) g_used = 0;
  return 0;
}
extern "C" int avf_timing_read(int cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  AVF_REQUIRE(cls >= 0 && cls < KC_COUNT && total_ms && launches && flops && bytes, "timing_read: bad arguments");
  double ms = 0, fl = 0, by = 0;
  int64_t n = 0;
  for (int i = 0; i < g_used; ++i) {
    if (g_recs[i].cls != cls) continue;
    float t = 0.f;
    hipError_t e = hipEventSynchronize(g_recs[i].stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, g_recs[i].start, g_recs[i].stop);
    AVF_REQUIRE(e == hipSuccess, "timing_read: %s", hipGetErrorString(e));
    ms += t; fl += g_recs[i].flops; by += g_recs[i].bytes; ++n;
  }
  *total_ms = ms; *launches = n; *flops = fl; *bytes = by;
  return 0;
}
extern "C" const char* avf_last_error(void) { return g_err; }

extern "C" int avf_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible");
    return 0;
  }
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    set_error("cannot query HIP device");
    return 0;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device arch %s is not gfx950", prop.gcnArchName);
    return 0;
  }
  return 1;
}

extern "C" int avf_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                                 float* mean, float* rstd, int64_t rows, int dim, float eps, void* stream) {
  AVF_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  return layernorm_fwd(x, gamma, beta, y, y_dtype, mean, rstd, rows, dim, eps, (hipStream_t)stream);
}
extern "C" size_t avf_layernorm_bwd_workspace_bytes(int64_t rows, int dim) { return layernorm_bwd_ws(rows, dim); }
extern "C" int avf_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, const float* dres, float* dx, void* dx_lo, float* dgamma,
                                 float* dbeta, float* dcolsum, void* workspace, int64_t rows, int dim, void* stream) {
  AVF_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "layernorm_bwd: null pointer");
  return layernorm_bwd(dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dx_lo, dgamma, dbeta, dcolsum, workspace, rows,
                       dim, (hipStream_t)stream);
}
extern "C" size_t avf_colsum_workspace_bytes(int64_t rows, int cols) { return colsum_ws(rows, cols); }
extern "C" int avf_colsum(const void* in, int in_dtype, int64_t rows, int cols, int64_t ld, float* out,
                          void* workspace, void* stream) {
  return colsum(in, in_dtype, rows, cols, ld, out, workspace, (hipStream_t)stream);
}
extern "C" int avf_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  AVF_REQUIRE(in && out, "cast: null pointer");
  return cast_f32_to_bf16(in, out, n, (hipStream_t)stream);
}
extern "C" int avf_dropout_factors(uint32_t seed_lo, uint32_t seed_hi, int layer_index, int site, float p, int64_t rows,
                                   int cols, float* out, void* stream) {
  AVF_REQUIRE(out && rows > 0 && cols > 0 && cols % 4 == 0 && p > 0.f && p < 1.f && site >= 0 && site < 3,
              "dropout_factors: bad arguments");
  const DropCfg d = make_drop(p, ((uint64_t)seed_hi << 32) | seed_lo, layer_index, site);
  return dropout_factors(d, out, rows * cols, (hipStream_t)stream);
}
extern "C" int avf_prep_weight_bf16(const float* w, void* w_lo, void* w_t_lo, int rows, int cols, void* stream) {
  AVF_REQUIRE(w, "prep_weight: null pointer");
  return prep_weight_bf16(w, w_lo, w_t_lo, rows, cols, (hipStream_t)stream);
}

extern "C" size_t avf_gemm_workspace_bytes(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K) {
  return gemm_ws(dtype, transA, transB, M, N, K);
}
extern "C" int avf_gemm(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                        const void* B, int64_t ldb, void* C, int64_t ldc, int c_dtype, int epilogue, const float* bias,
                        const float* residual, int64_t ldres, void* aux, int64_t ldaux, void* workspace, void* stream) {
  AVF_REQUIRE(A && B && C, "gemm: null pointer");
  GemmArgs a;
  a.dtype = dtype; a.transA = transA; a.transB = transB;
  a.M = M; a.N = N; a.K = K;
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.c_dtype = c_dtype; a.epilogue = epilogue; a.bias = bias; a.residual = residual; a.ldres = ldres;
  a.aux = aux; a.ldaux = ldaux; a.workspace = workspace; a.colsum = nullptr; a.drop = kNoDrop; a.defer_fold = nullptr;
  return gemm(a, (hipStream_t)stream);
}

// ---- weight-stationary persistent NT GEMM (gemm_ws.hip) ----
extern "C" int avf_pack_weight_ws_ok(int64_t rows, int64_t cols) { return pack_ws_ok(rows, cols) ? 1 : 0; }
extern "C" size_t avf_pack_weight_ws_bytes(int64_t rows, int64_t cols) { return pack_ws_ok(rows, cols) ? pack_ws_bytes(rows, cols) : 0; }
extern "C" size_t avf_gemm_nt_ws_workspace_bytes(int64_t M, int64_t N) { return gemm_nt_colsum_ws(M, N); }
// 1 when avf_gemm / the layer calls send this shape and epilogue to the persistent kernel (aligned operands, no dropout, no
// column sums except on DGELU): what a test or a profile tool asks to know which kernel it is looking at
extern "C" int avf_gemm_nt_ws_dispatch(int64_t M, int64_t N, int64_t K, int epilogue, int c_dtype) {
  GemmArgs a;
  a.dtype = AVF_BF16; a.transA = 0; a.transB = 1;
  a.M = M; a.N = N; a.K = K;
  void* aligned = (void*)(uintptr_t)256;  // never dereferenced: the predicates look at alignment only
  a.A = aligned; a.lda = K; a.B = nullptr; a.ldb = K; a.C = aligned; a.ldc = N;
  a.c_dtype = c_dtype; a.epilogue = epilogue; a.bias = (epilogue == AVF_EPI_BIAS_RES || epilogue == AVF_EPI_BIAS_GELU) ? (const float*)aligned : nullptr;
  a.residual = epilogue == AVF_EPI_BIAS_RES ? aligned : nullptr; a.ldres = N;
  a.aux = (epilogue == AVF_EPI_BIAS_GELU || epilogue == AVF_EPI_DGELU) ? aligned : nullptr; a.ldaux = N;
  a.workspace = nullptr; a.colsum = nullptr; a.drop = kNoDrop; a.defer_fold = nullptr;
  a.Bp = aligned; a.ws_force = 0; a.mx_q = nullptr; a.mx_s = nullptr;
  return gemm_bf16_nt_ws_preferred(a) ? 1 : 0;
}
extern "C" int avf_pack_weight_ws(const void* w_bf16, int64_t ldw, int64_t rows, int64_t cols, void* out, void* stream) {
  return pack_ws(w_bf16, ldw, rows, cols, out, (hipStream_t)stream);
}
extern "C" int avf_gemm_nt_ws(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B_packed, void* C,
                              int64_t ldc, int c_dtype, int epilogue, const float* bias, const void* residual, int64_t ldres,
                              void* aux, int64_t ldaux, void* workspace, float* colsum, void* mx_q, void* mx_s, void* stream) {
  AVF_REQUIRE(A && B_packed && C, "gemm_nt_ws: null pointer");
  GemmArgs a;
  a.dtype = AVF_BF16; a.transA = 0; a.transB = 1;
  a.M = M; a.N = N; a.K = K;
  a.A = A; a.lda = lda; a.B = nullptr; a.ldb = K; a.C = C; a.ldc = ldc;
  a.c_dtype = c_dtype; a.epilogue = epilogue; a.bias = bias; a.residual = residual; a.ldres = ldres;
  a.aux = aux; a.ldaux = ldaux; a.workspace = workspace; a.colsum = colsum; a.drop = kNoDrop; a.defer_fold = nullptr;
  a.Bp = B_packed; a.ws_force = 1;
  a.mx_q = mx_q; a.mx_s = mx_s;
  AVF_REQUIRE(gemm_bf16_nt_ws_ok(a), "gemm_nt_ws: the weight-stationary kernel takes K == 512, N %% 256 == 0, M >= 2048, 16-byte "
              "aligned operands; an MX-FP8 image only with DGELU + colsum + a bf16 C (M=%lld N=%lld K=%lld)", (long long)M,
              (long long)N, (long long)K);
  return gemm_bf16_nt(a, (hipStream_t)stream);
}

static int fill_tn_group(TnGroupArgs* g, int count, int64_t K, const void* const* A, const void* const* B, float* const* C,
                         const int64_t* M, const int64_t* N) {
  AVF_REQUIRE(count >= 1 && count <= 4 && M && N, "gemm_tn_group: 1..4 problems");
  memset(g, 0, sizeof(*g));
  g->count = count;
  g->K = K;
  for (int i = 0; i < count; ++i) {
    g->A[i] = A ? A[i] : nullptr; g->B[i] = B ? B[i] : nullptr; g->C[i] = C ? C[i] : nullptr;
    g->M[i] = M[i]; g->N[i] = N[i]; g->lda[i] = M[i]; g->ldb[i] = N[i];
  }
  return 0;
}
extern "C" size_t avf_gemm_tn_group_workspace_bytes(int count, int64_t K, const int64_t* M, const int64_t* N) {
  TnGroupArgs g;
  if (fill_tn_group(&g, count, K, nullptr, nullptr, nullptr, M, N)) return 0;
  return gemm_bf16_tn_group_ws(g);
}
extern "C" int avf_gemm_tn_group(int count, int64_t K, const void* const* A, const void* const* B, float* const* C,
                                 const int64_t* M, const int64_t* N, void* workspace, void* stream) {
  AVF_REQUIRE(A && B && C, "gemm_tn_group: null pointer");
  TnGroupArgs g;
  AVF_TRY(fill_tn_group(&g, count, K, A, B, C, M, N));
  for (int i = 0; i < count; ++i) AVF_REQUIRE(A[i] && B[i] && C[i], "gemm_tn_group: null operand %d", i);
  g.workspace = workspace;
  return gemm_bf16_tn_group(g, (hipStream_t)stream, nullptr);
}

extern "C" int avf_layernorm_fwd_mx8(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                     float* rstd, void* y_q, void* y_scales, int64_t rows, int dim, float eps, void* stream) {
  AVF_REQUIRE(x && gamma && beta && y_bf16 && mean && rstd && y_q && y_scales, "layernorm_fwd_mx8: null pointer");
  return layernorm_fwd(x, gamma, beta, y_bf16, AVF_BF16, mean, rstd, rows, dim, eps, (hipStream_t)stream, y_q, y_scales);
}
extern "C" int avf_layernorm_bwd_mx8(const void* dy_bf16, const float* x, const float* gamma, const float* mean,
                                     const float* rstd, const float* dres, float* dx, void* dx_lo, void* dx_q, void* dx_scales,
                                     float* dgamma, float* dbeta, void* workspace, int64_t rows, int dim, void* stream) {
  AVF_REQUIRE(dy_bf16 && x && gamma && mean && rstd && dx_lo && dx_q && dx_scales && dgamma && dbeta && workspace,
              "layernorm_bwd_mx8: null pointer");
  return layernorm_bwd(dy_bf16, AVF_BF16, x, gamma, mean, rstd, dres, dx, dx_lo, dgamma, dbeta, nullptr, workspace, rows, dim,
                       (hipStream_t)stream, kNoDrop, nullptr, AVF_F32, AVF_F32, dx_q, dx_scales);
}
extern "C" int avf_attn_fwd_masked(int dtype, const void* qkv, void* o, float* lse2, const void* keep, int batch, int tokens,
                                   int heads, int dim_head, void* stream) {
  AVF_REQUIRE(qkv && o && lse2 && keep, "attn_fwd_masked: null pointer");
  return attn_fwd_vec(dtype, qkv, o, lse2, batch, tokens, heads, dim_head, (hipStream_t)stream, keep, false);
}
extern "C" int avf_attn_bwd_masked(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv,
                                   void* workspace, const void* keep, int batch, int tokens, int heads, int dim_head,
                                   void* stream) {
  AVF_REQUIRE(qkv && o && d_o && lse2 && dqkv && workspace && keep, "attn_bwd_masked: null pointer");
  return attn_bwd_vec(dtype, qkv, o, d_o, lse2, dqkv, (float*)workspace, batch, tokens, heads, dim_head, (hipStream_t)stream,
                      keep, false);
}
extern "C" int avf_attn_fwd_mx8(const void* qkv, void* o, float* lse2, void* o_q, void* o_scales, int batch, int tokens,
                                int heads, int dim_head, void* stream) {
  AVF_REQUIRE(qkv && o && lse2 && o_q && o_scales, "attn_fwd_mx8: null pointer");
  AVF_REQUIRE(attn_fwd_emits_mx8(tokens, dim_head),
              "attn_fwd_mx8: only the head-resident kernel (dim_head 64, tokens <= 576) writes the image (tokens=%d dim_head=%d)",
              tokens, dim_head);
  return attn_fwd_bf16((const bf16*)qkv, (bf16*)o, lse2, batch, tokens, heads, dim_head, (hipStream_t)stream, false, o_q,
                       o_scales);
}
extern "C" int avf_attn_bwd_emits_mx8(int tokens, int dim_head) { return attn_bwd_emits_mx8(tokens, dim_head, true) ? 1 : 0; }
extern "C" int avf_attn_bwd_mx8(const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, void* dqkv_q,
                                void* dqkv_scales, int batch, int tokens, int heads, int dim_head, void* stream) {
  AVF_REQUIRE(qkv && o && d_o && lse2 && dqkv && dqkv_q && dqkv_scales, "attn_bwd_mx8: null pointer");
  AVF_REQUIRE(attn_bwd_emits_mx8(tokens, dim_head, true),
              "attn_bwd_mx8: only the merged backward kernel writes the image (tokens=%d dim_head=%d)", tokens, dim_head);
  return attn_bwd_bf16((const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse2, (bf16*)dqkv, nullptr, batch, tokens, heads,
                       dim_head, (hipStream_t)stream, true, nullptr, nullptr, dqkv_q, dqkv_scales);
}
extern "C" int avf_quant_mx8(int dtype, const void* x, int64_t rows, int64_t cols, void* q, void* scales, void* stream) {
  AVF_REQUIRE(x && q && scales, "quant_mx8: null pointer");
  return quant_mx8(x, dtype, cols, rows, cols, q, cols, scales, (hipStream_t)stream);
}
extern "C" int avf_gemm_mx8_nt(int64_t M, int64_t N, int64_t K, const void* a_q, const void* a_scales, const void* b_q,
                               const void* b_scales, void* C, int64_t ldc, int c_dtype, int epilogue, const float* bias,
                               const float* residual, int64_t ldres, void* aux, int64_t ldaux, void* c_q, void* c_scales,
                               void* stream) {
  AVF_REQUIRE(a_q && a_scales && b_q && b_scales && C, "gemm_mx8_nt: null pointer");
  GemmArgs a;
  a.dtype = AVF_BF16; a.transA = 0; a.transB = 1;
  a.M = M; a.N = N; a.K = K;
  a.A = a_q; a.lda = K; a.B = b_q; a.ldb = K; a.C = C; a.ldc = ldc;
  a.c_dtype = c_dtype; a.epilogue = epilogue; a.bias = bias; a.residual = residual; a.ldres = ldres;
  a.aux = aux; a.ldaux = ldaux; a.workspace = nullptr; a.colsum = nullptr; a.drop = kNoDrop; a.defer_fold = nullptr;
  return gemm_mx8_nt(a, a_scales, b_scales, (hipStream_t)stream, c_q, c_scales);
}

extern "C" int avf_attn_fwd(int dtype, const void* qkv, void* o, float* lse2, int batch, int tokens, int heads,
                            int dim_head, void* stream) {
  AVF_REQUIRE(qkv && o && lse2, "attn_fwd: null pointer");
  if (dtype == AVF_F32)
    return attn_fwd_f32((const float*)qkv, (float*)o, lse2, batch, tokens, heads, dim_head, (hipStream_t)stream);
  if (dtype == AVF_BF16)
    return attn_fwd_bf16((const bf16*)qkv, (bf16*)o, lse2, batch, tokens, heads, dim_head, (hipStream_t)stream);
  AVF_REQUIRE(false, "attn_fwd: bad dtype %d", dtype);
}
extern "C" size_t avf_attn_bwd_workspace_bytes(int batch, int tokens, int heads, int dim_head) {
  (void)dim_head;
  return (size_t)batch * tokens * heads * sizeof(float);
}
extern "C" int avf_attn_bwd(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv,
                            void* workspace, int batch, int tokens, int heads, int dim_head, void* stream) {
  AVF_REQUIRE(qkv && o && d_o && lse2 && dqkv && workspace, "attn_bwd: null pointer");
  if (dtype == AVF_F32)
    return attn_bwd_f32((const float*)qkv, (const float*)o, (const float*)d_o, lse2, (float*)dqkv, (float*)workspace,
                        batch, tokens, heads, dim_head, (hipStream_t)stream);
  if (dtype == AVF_BF16)
    return attn_bwd_bf16((const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse2, (bf16*)dqkv, (float*)workspace,
                         batch, tokens, heads, dim_head, (hipStream_t)stream);
  AVF_REQUIRE(false, "attn_bwd: bad dtype %d", dtype);
}

// bf16 attention on a projection whose q columns already carry log2(e)/sqrt(dim_head) - what avf_layer_fwd/bwd run
// (the factor is folded into the query rows of the bf16 Wqkv image).  workspace: 2 * avf_attn_bwd_workspace_bytes.
extern "C" int avf_attn_fwd_qs(const void* qkv, void* o, float* lse2, int batch, int tokens, int heads, int dim_head,
                               void* stream) {
  AVF_REQUIRE(qkv && o && lse2, "attn_fwd_qs: null pointer");
  return attn_fwd_bf16((const bf16*)qkv, (bf16*)o, lse2, batch, tokens, heads, dim_head, (hipStream_t)stream, true);
}
extern "C" int avf_attn_bwd_qs(const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv,
                               void* workspace, int batch, int tokens, int heads, int dim_head, void* stream) {
  AVF_REQUIRE(qkv && o && d_o && lse2 && dqkv && workspace, "attn_bwd_qs: null pointer");
  float* w = (float*)workspace;
  return attn_bwd_bf16((const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse2, (bf16*)dqkv, w, batch, tokens, heads,
                       dim_head, (hipStream_t)stream, true, w + (size_t)batch * tokens * heads);
}

// After a FAILED stream capture (an operation that cannot be recorded was issued while capturing): end a capture that is still
// open on `stream` (discarding its graph) and clear the runtime's sticky last-error, which would otherwise surface at the
// caller's next, unrelated HIP call.  Returns the error code that was pending (0: none).
extern "C" int avf_hip_error_reset(void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(s, &g);
    if (g) (void)hipGraphDestroy(g);
  }
  int first = 0;
  for (int i = 0; i < 8; ++i) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) break;
    if (!first) first = (int)e;
  }
  return first;
}

// A last line for a process that may die inside an OPTIONAL step (bench.py: the multi-rank hipGraph attempt that follows a completed
// eager measurement).  While armed, a fatal signal (SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL - a crash inside the runtime's capture /
// instantiate path cannot be caught as an exception) writes `line` to `fd` with write(2) and leaves with _exit(0): the measurement
// that was already taken is reported, nothing is retried.  fd < 0 or an empty line: just leave.  Disarm restores the handlers.
#include <signal.h>
#include <unistd.h>
namespace {
char g_crash_line[16384];
int g_crash_len = 0, g_crash_fd = -1;
struct sigaction g_crash_old[5];
const int g_crash_sigs[5] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
bool g_crash_armed = false;
void crash_line_handler(int) {
  if (g_crash_fd >= 0 && g_crash_len > 0) {
    ssize_t w = write(g_crash_fd, g_crash_line, (size_t)g_crash_len);
    (void)w;
  }
  _exit(0);
}
}  // namespace
extern "C" int avf_crash_line_arm(const char* line, int fd) {
  AVF_REQUIRE(!g_crash_armed, "crash_line_arm: already armed");
  const size_t n = line ? strlen(line) : 0;
  AVF_REQUIRE(n < sizeof(g_crash_line), "crash_line_arm: line too long");
  if (n) memcpy(g_crash_line, line, n);
  g_crash_len = (int)n;
  g_crash_fd = fd;
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = crash_line_handler;
  sigemptyset(&sa.sa_mask);
  for (int i = 0; i < 5; ++i) sigaction(g_crash_sigs[i], &sa, &g_crash_old[i]);
  g_crash_armed = true;
  return 0;
}
extern "C" int avf_crash_line_disarm(void) {
  if (!g_crash_armed) return 0;
  for (int i = 0; i < 5; ++i) sigaction(g_crash_sigs[i], &g_crash_old[i], nullptr);
  g_crash_armed = false;
  g_crash_len = 0;
  g_crash_fd = -1;
  return 0;
}

extern "C" int avf_set_f32_arith(int mode) {
  const int prev = get_f32_arith();
  set_f32_arith(mode);
  return prev;
}
extern "C" int avf_get_f32_arith(void) { return get_f32_arith(); }

extern "C" int avf_selftest_mfma_bf16(const void* a, const void* b, float* c, void* stream) {
  selftest_mfma_bf16_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const bf16*)a, (const bf16*)b, c);
  return check_launch("selftest_mfma_bf16");
}
extern "C" int avf_selftest_mfma_f32(const float* a, const float* b, float* c, void* stream) {
  selftest_mfma_f32_kernel<<<1, 64, 0, (hipStream_t)stream>>>(a, b, c);
  return check_launch("selftest_mfma_f32");
}
extern "C" int avf_selftest_tr16(const void* tile, void* out, void* stream) {
  selftest_tr16_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const bf16*)tile, (bf16*)out);
  return check_launch("selftest_tr16");
}
