// graph_fork.hip - does a hipGraph captured from a FORKED two-stream schedule replay as fast as the eager schedule?
//
// Round 3 saw a captured training step that forked one launch per layer to a side stream replay 2.2x slower than the same step
// captured on one stream (DESIGN_HISTORY.md section 14).  The data-parallel step forks too (the process group's communication stream
// joins the capture through events), so the question decides the multi-rank launch default of bench.py.  This is the pattern
// alone: a chain of L "layers" on stream A, each of NK short dependent kernels; layer l forks ONE longer kernel to stream B
// (event record on A, wait on B) which the NEXT layer joins before its last kernel (event record on B, wait on A).
//   eager      : the launches as written, two streams
//   graph-fork : the same, captured (hipStreamCaptureModeGlobal) and replayed
//   graph-1s   : everything on stream A, captured and replayed (no fork: the side kernels run in line)
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/graph_fork.hip -o tools/diag/bin/graph_fork
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void spin(float* p, int iters) {  // ~iters * 4 cycles per thread: a kernel of known, short duration
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

static void schedule(hipStream_t a, hipStream_t b, hipEvent_t* fork, hipEvent_t* join, float* bufa, float* bufb, int L, int NK, bool two) {
  for (int l = 0; l < L; ++l) {
    for (int k = 0; k < NK; ++k) {
      if (k == NK - 1 && l > 0 && two) CK(hipStreamWaitEvent(a, join[l - 1], 0));  // join the previous layer's side kernel
      spin<<<256, 256, 0, a>>>(bufa, 2000);                                         // ~8 us chain kernel, whole chip
    }
    if (two) {
      CK(hipEventRecord(fork[l], a));
      CK(hipStreamWaitEvent(b, fork[l], 0));
      spin<<<256, 256, 0, b>>>(bufb, 12000);  // ~45 us side kernel
      CK(hipEventRecord(join[l], b));
    } else {
      spin<<<256, 256, 0, a>>>(bufb, 12000);
    }
  }
  if (two) CK(hipStreamWaitEvent(a, join[L - 1], 0));
}

int main() {
  const int L = 6, NK = 10, REP = 50;
  float *bufa, *bufb;
  CK(hipMalloc(&bufa, 256 * 256 * 4));
  CK(hipMalloc(&bufb, 256 * 256 * 4));
  CK(hipMemset(bufa, 0, 256 * 256 * 4));
  CK(hipMemset(bufb, 0, 256 * 256 * 4));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t fork[L], join[L];
  for (int l = 0; l < L; ++l) {
    CK(hipEventCreateWithFlags(&fork[l], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join[l], hipEventDisableTiming));
  }
  auto wall = [&](auto fn) {
    for (int i = 0; i < 5; ++i) fn();
    CK(hipStreamSynchronize(a));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < REP; ++i) fn();
    CK(hipStreamSynchronize(a));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
  };
  const double eager2 = wall([&] { schedule(a, b, fork, join, bufa, bufb, L, NK, true); });
  const double eager1 = wall([&] { schedule(a, b, fork, join, bufa, bufb, L, NK, false); });
  hipGraph_t g2, g1;
  hipGraphExec_t x2, x1;
  CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
  schedule(a, b, fork, join, bufa, bufb, L, NK, true);
  CK(hipStreamEndCapture(a, &g2));
  CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
  schedule(a, b, fork, join, bufa, bufb, L, NK, false);
  CK(hipStreamEndCapture(a, &g1));
  CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
  const double graph2 = wall([&] { CK(hipGraphLaunch(x2, a)); });
  const double graph1 = wall([&] { CK(hipGraphLaunch(x1, a)); });
  printf("%d layers x (%d chain kernels + 1 side kernel), us per pass:\n", L, NK);
  printf("  eager, two streams (fork / join)   %8.1f\n", eager2);
  printf("  eager, one stream                  %8.1f\n", eager1);
  printf("  graph replay of the forked capture %8.1f   (%.2fx the eager forked schedule)\n", graph2, graph2 / eager2);
  printf("  graph replay, one stream           %8.1f   (%.2fx the eager one-stream schedule)\n", graph1, graph1 / eager1);
  return 0;
}
