// mfma_fp8_probe.hip - pins the operand / scale / result maps of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands)
// on the hardware before gemm_fp8.hip relies on them.
//
//   hipcc --offload-arch=gfx950 -O2 tools/diag/mfma_fp8_probe.hip -o /tmp/mfma_fp8_probe && /tmp/mfma_fp8_probe
//
// Map checked (found with mfma_fp8_probe2.hip): with X the first and Y the second operand, both [16][128] e4m3,
//   D[row = 4 (l >> 4) + r][col = l & 15] = sum_k 2^(sx[row][k/32]-127) 2^(sy[col][k/32]-127) X[row][k] Y[col][k]
//   lane (i = l & 15, g = l >> 4) holds X[i][16 g + b] in bytes b = 0..15 and X[i][64 + 16 g + b] in bytes 16..31
//   (so the two 16-byte halves of a lane's operand are the 16-byte chunks g and 4 + g of a 128-byte row), and
//   supplies the E8M0 scale of the 32-element block k = 32 g .. 32 g + 31 of its row in the byte of the scale VGPR
//   that op_sel names (the other three bytes are ignored).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OA, int OB>
__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OA, sa[l], OB, sb[l]);
  c[l] = acc;
}

static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -x : x;
}

#define CK(x)                                                             \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 2;                                                           \
    }                                                                     \
  } while (0)

int main() {
  uint8_t ha[64][32], hb[64][32];
  int hsa[64], hsb[64];
  srand(7);
  int bad = 0;
  for (int sel = 0; sel < 4; ++sel) {
    for (int l = 0; l < 64; ++l) {
      for (int j = 0; j < 32; ++j) {
        ha[l][j] = (uint8_t)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
        hb[l][j] = (uint8_t)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
      }
      const int ea = 124 + rand() % 7, eb = 124 + rand() % 7;
      hsa[l] = (int)(0x11223344u & ~(0xffu << (8 * sel))) | (ea << (8 * sel));  // other bytes: junk
      hsb[l] = (int)(0x55667788u & ~(0xffu << (8 * sel))) | (eb << (8 * sel));
    }
    void *da, *db, *dsa, *dsb, *dc;
    CK(hipMalloc(&da, sizeof(ha))); CK(hipMalloc(&db, sizeof(hb)));
    CK(hipMalloc(&dsa, sizeof(hsa))); CK(hipMalloc(&dsb, sizeof(hsb))); CK(hipMalloc(&dc, 64 * 16));
    CK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice));
    switch (sel) {
      case 0: probe<0, 0><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
      case 1: probe<1, 1><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
      case 2: probe<2, 2><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
      default: probe<3, 3><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
    }
    CK(hipDeviceSynchronize());
    float hc[64][4];
    CK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        double ref = 0;
        for (int k = 0; k < 128; ++k) {
          const int g = (k & 63) >> 4, b = (k & 15) + (k >= 64 ? 16 : 0), kb = k >> 5;
          const int ea = (hsa[row + 16 * kb] >> (8 * sel)) & 255, eb = (hsb[col + 16 * kb] >> (8 * sel)) & 255;
          ref += ldexp((double)e4m3(ha[row + 16 * g][b]) * e4m3(hb[col + 16 * g][b]), ea - 127 + eb - 127);
        }
        const double err = fabs(hc[l][r] - ref) / (fabs(ref) + 1.0);
        if (err > worst) worst = err;
      }
    printf("op_sel %d: worst relative deviation from the map = %.3g %s\n", sel, worst, worst < 1e-5 ? "OK" : "MISMATCH");
    bad += worst >= 1e-5;
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dc);
  }
  return bad ? 1 : 0;
}
