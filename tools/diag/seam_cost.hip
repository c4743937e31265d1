// Diagnostic (not part of the product): what does a grid-wide seam cost INSIDE one launch against a kernel boundary?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag/seam_cost.hip -o tools/diag/bin/seam_cost && tools/diag/bin/seam_cost
//
// The question behind it: the 17- / 49-token layers of the reference (vformer.py:271, sformer.py:240) run
// ~17 dependent launches of 5 - 9 us per layer; would ONE persistent launch per layer, its phases (LN + QKV | attention |
// out-proj + LN + MLP1 | MLP2 ...) separated by a device-scope barrier on an atomic counter, be faster?  Every phase boundary of
// such a kernel is an all-to-all seam: each workgroup's output is read by other workgroups on other XCDs, so the seam needs a
// release of the XCD's L2, an arrival counter, a poll, and an acquire (per-XCD L2s are not coherent).
//
// Both forms run the SAME phase body: workgroup w reads the 16 KiB another workgroup (other XCD) wrote in the previous phase,
// reduces it, writes its own 16 KiB.  (a) P kernels on one stream, captured into a hipGraph and replayed; (b) one persistent
// kernel of P phases with an XCD-hierarchical barrier (per-XCD arrival counter, the last arriver of an XCD releases and arrives
// at a top counter, then publishes the generation to its XCD; everyone acquires).  Prints us per phase of both and checks that
// they computed the same thing (a wrong barrier shows up as a different checksum).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

constexpr int SLAB = 16384 / 16;  // uint4 per workgroup and phase
constexpr int THREADS = 512;

__device__ __forceinline__ void phase_body(const uint4* __restrict__ in, uint4* __restrict__ out, int w, int nwg, int p, bool sc1) {
  const int src = (w * 37 + 11 + p) % nwg;  // another workgroup's slab (ids 37 apart: another XCD)
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < SLAB; i += THREADS) {
    uint4 v;
    if (sc1) {  // device-coherent load (bypasses this CU's L1): what a consumer of another workgroup's fresh data must use
      const uint4* a = in + (size_t)src * SLAB + i;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    } else {
      v = in[(size_t)src * SLAB + i];
    }
    acc.x += v.x * 3u + 1u; acc.y += v.y ^ (uint32_t)p; acc.z += v.z + v.x; acc.w += v.w * 5u;
  }
  for (int i = threadIdx.x; i < SLAB; i += THREADS)
    out[(size_t)w * SLAB + i] = make_uint4(acc.x + i, acc.y + w, acc.z + p, acc.w);
}

__global__ __launch_bounds__(THREADS) void phase_kernel(const uint4* in, uint4* out, int nwg, int p) {
  phase_body(in, out, blockIdx.x, nwg, p, false);
}

// monotonic counters, never reset: generation g completes when the top counter reaches 8 g
struct Barrier {
  unsigned int* xcc;  // [8][32] (one 128-byte line per XCD): arrivals
  unsigned int* gen;  // [8][32]: published generation per XCD
  unsigned int* top;  // arrivals of XCD leaders
};

__device__ __forceinline__ unsigned int ld_sc1(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void grid_barrier(const Barrier& b, unsigned int g, int per_xcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned int n = __hip_atomic_fetch_add(b.xcc + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n + 1 == g * (unsigned int)per_xcd) {  // last arriver of this XCD in generation g
      __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (ld_sc1(b.top) < 8u * g && ++spins < (1L << 26)) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(b.gen + x * 32, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (ld_sc1(b.gen + x * 32) < g && ++spins < (1L << 26)) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(THREADS) void persistent_kernel(uint4* a, uint4* b, int nwg, int phases, Barrier bar, unsigned int g0) {
  for (int p = 0; p < phases; ++p) {
    phase_body((p & 1) ? b : a, (p & 1) ? a : b, blockIdx.x, nwg, p, false);  // plain loads: the acquire invalidated this CU's L1
    if (p + 1 < phases) grid_barrier(bar, g0 + p + 1, nwg / 8);
  }
}

static unsigned long long checksum(const uint4* d, size_t n) {
  uint4* h = (uint4*)malloc(n * sizeof(uint4));
  CHECK(hipMemcpy(h, d, n * sizeof(uint4), hipMemcpyDeviceToHost));
  unsigned long long s = 0;
  for (size_t i = 0; i < n; ++i) s = s * 1315423911ull + h[i].x + 3ull * h[i].y + 5ull * h[i].z + 7ull * h[i].w;
  free(h);
  return s;
}

int main() {
  const int nwg = 256, P = 26;  // one workgroup per CU; 26 phases ~ forward + backward of two layers
  uint4 *a, *b;
  const size_t n = (size_t)nwg * SLAB;
  CHECK(hipMalloc(&a, n * sizeof(uint4)));
  CHECK(hipMalloc(&b, n * sizeof(uint4)));
  unsigned int* ctr;
  CHECK(hipMalloc(&ctr, (2 * 8 * 32 + 32) * sizeof(unsigned int)));
  CHECK(hipMemset(ctr, 0, (2 * 8 * 32 + 32) * sizeof(unsigned int)));
  Barrier bar{ctr, ctr + 8 * 32, ctr + 2 * 8 * 32};
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto init = [&] { CHECK(hipMemsetAsync(a, 1, n * sizeof(uint4), s)); CHECK(hipMemsetAsync(b, 2, n * sizeof(uint4), s)); };

  // (a) P kernels, captured
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int p = 0; p < P; ++p) phase_kernel<<<nwg, THREADS, 0, s>>>((p & 1) ? b : a, (p & 1) ? a : b, nwg, p);
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  float best_a = 1e9f, best_b = 1e9f;
  unsigned long long sum_a = 0, sum_b = 0;
  for (int it = 0; it < 12; ++it) {
    init();
    CHECK(hipEventRecord(e0, s));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 1 && ms < best_a) best_a = ms;
    sum_a = checksum((P & 1) ? b : a, n);
  }
  // (b) one persistent kernel with P - 1 grid barriers
  unsigned int gen = 0;
  for (int it = 0; it < 12; ++it) {
    init();
    CHECK(hipEventRecord(e0, s));
    persistent_kernel<<<nwg, THREADS, 0, s>>>(a, b, nwg, P, bar, gen);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    gen += P - 1;
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 1 && ms < best_b) best_b = ms;
    sum_b = checksum((P & 1) ? b : a, n);
  }
  // (c) the phase body alone (one kernel, no dependency): what both forms pay per phase besides their seam
  float best_c = 1e9f;
  for (int it = 0; it < 12; ++it) {
    CHECK(hipEventRecord(e0, s));
    phase_kernel<<<nwg, THREADS, 0, s>>>(a, b, nwg, 0);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 1 && ms < best_c) best_c = ms;
  }
  printf("phases %d, %d workgroups x %d threads, 16 KiB read + 16 KiB written per workgroup and phase\n", P, nwg, THREADS);
  printf("(a) %d kernels, hipGraph replay      : %8.2f us total, %6.2f us per phase\n", P, best_a * 1e3, best_a * 1e3 / P);
  printf("(b) one persistent kernel, XCD barrier: %8.2f us total, %6.2f us per phase\n", best_b * 1e3, best_b * 1e3 / P);
  printf("(c) one phase kernel alone (event pair): %7.2f us\n", best_c * 1e3);
  printf("checksums %s (%llx / %llx)\n", sum_a == sum_b ? "EQUAL" : "DIFFER", sum_a, sum_b);
  return sum_a == sum_b ? 0 : 2;
}
