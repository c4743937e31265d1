// exploratory: (1) accumulation exactness with unit scales, (2) scale map of the SECOND operand
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  c[l] = acc;
}
static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -x : x;
}
static uint8_t enc_pow2(int p) { return (uint8_t)((p + 7) << 3); }
uint8_t ha[64][32], hb[64][32];
int hsa[64], hsb[64];
float hc[64][4];
void *da, *db, *dsa, *dsb, *dc;
static void run() {
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc);
  hipDeviceSynchronize();
  hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
}
static double ref_at(int row, int col, bool use_sa, bool use_sb) {
  double ref = 0;
  for (int k = 0; k < 128; ++k) {
    const int g = (k & 63) >> 4, b = (k & 15) + (k >= 64 ? 16 : 0), kb = k >> 5;
    const int ea = use_sa ? (hsa[row + 16 * kb] & 255) : 127, eb = use_sb ? (hsb[col + 16 * kb] & 255) : 127;
    ref += ldexp((double)e4m3(ha[row + 16 * g][b]) * e4m3(hb[col + 16 * g][b]), ea - 127 + eb - 127);
  }
  return ref;
}
static double worst(bool usa, bool usb) {
  double w = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const double ref = ref_at(4 * (l >> 4) + r, l & 15, usa, usb);
      const double e = fabs(hc[l][r] - ref) / (fabs(ref) + 1.0);
      if (e > w) w = e;
    }
  return w;
}
int main() {
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
  srand(3);
  auto rnd = [&](bool sign) {
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        ha[l][j] = (uint8_t)(((sign ? rand() & 1 : 0) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
        hb[l][j] = (uint8_t)(((sign ? rand() & 1 : 0) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
      }
  };
  for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7f;
  rnd(true); run();
  printf("unit scales, random signs:   worst rel dev %.3g\n", worst(false, false));
  rnd(false); run();
  printf("unit scales, positive only:  worst rel dev %.3g\n", worst(false, false));
  rnd(true);
  for (int l = 0; l < 64; ++l) hsa[l] = 124 + rand() % 7;
  run();
  printf("random X scales, unit Y:     worst rel dev %.3g\n", worst(true, false));
  for (int l = 0; l < 64; ++l) { hsa[l] = 0x7f; hsb[l] = 124 + rand() % 7; }
  run();
  printf("unit X, random Y scales:     worst rel dev %.3g\n", worst(false, true));
  for (int l = 0; l < 64; ++l) { hsa[l] = 124 + rand() % 7; }
  run();
  printf("random X and Y scales:       worst rel dev %.3g\n", worst(true, true));
  // one large term among small ones: does the adder keep the small ones?
  memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
  for (int l = 0; l < 64; ++l) { hsa[l] = hsb[l] = 0x7f; }
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { ha[l][j] = enc_pow2(-6); hb[l][j] = enc_pow2(-6); }  // 2^-12 each
  ha[0][0] = enc_pow2(8); hb[0][0] = enc_pow2(8);                                                                   // 2^16
  run();
  printf("one 2^16 product + 127 products of 2^-12: D[0][0] - 65536 = %.9g (exact: %.9g)\n", hc[0][0] - 65536.f, 127 * ldexp(1.0, -12));
  ha[0][0] = enc_pow2(4); hb[0][0] = enc_pow2(4);
  run();
  printf("one 2^8 product + 127 products of 2^-12:  D[0][0] - 256   = %.9g (exact: %.9g)\n", hc[0][0] - 256.f, 127 * ldexp(1.0, -12));
  return 0;
}
