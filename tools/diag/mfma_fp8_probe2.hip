// exploratory: which (row, k-block) does the scale byte q of lane L scale under op_sel S?  (first operand only)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int OA, int OB>
__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OA, sa[l], OB, sb[l]);
  c[l] = acc;
}
static uint8_t enc_pow2(int p) { return (uint8_t)((p + 7) << 3); }  // 2^p, e4m3
int main() {
  uint8_t ha[64][32], hb[64][32];
  int hsa[64], hsb[64];
  void *da, *db, *dsa, *dsb, *dc;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
  // data-map test: X[l][b] = 1 only at one (l0, b0); Y all ones with distinct powers per (g) -> find which D row lights up
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { ha[l][j] = enc_pow2(0); hb[l][j] = enc_pow2(l >> 4); }
  for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7f7f7f7f;
  float base[64][4], hc[64][4];
  auto run = [&](int S, float (*out)[4]) {
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    switch (S) {
      case 0: probe<0, 0><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
      case 1: probe<1, 0><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
      case 2: probe<2, 0><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
      default: probe<3, 0><<<1, 64>>>((v8i*)da, (v8i*)db, (int*)dsa, (int*)dsb, (v4f*)dc); break;
    }
    hipDeviceSynchronize();
    hipMemcpy(out, dc, 1024, hipMemcpyDeviceToHost);
  };
  run(0, base);
  printf("baseline D[0][0] = %g (expect 32*(1+2+4+8) = 480), D lane 17 r2 = %g\n", base[0][0], base[17][2]);
  for (int S = 0; S < 4; ++S)
    for (int q = 0; q < 4; ++q) {
      printf("op_sel %d, scale byte %d:", S, q);
      int shown = 0;
      for (int L = 0; L < 64; ++L) {
        for (int l = 0; l < 64; ++l) hsa[l] = 0x7f7f7f7f;
        hsa[L] = (int)((0x7f7f7f7fu & ~(0xffu << (8 * q))) | (0x80u << (8 * q)));
        run(S, hc);
        // find affected rows and k-blocks from column 0 (lane l = col 0 + 16*(row/4), r = row%4)
        for (int row = 0; row < 16; ++row) {
          const float d = hc[16 * (row >> 2)][row & 3] - base[16 * (row >> 2)][row & 3];
          if (d != 0.f) {
            if (shown < 70) printf(" L%d->(row %d,+%g)", L, row, d);
            ++shown;
          }
        }
      }
      printf("  [%d hits]\n", shown);
    }
  // data map: a single non-zero byte in X
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7f7f7f7f;
  printf("data map (X lane L byte b set to 1, rest 0; Y[l][j] = 2^(l>>4) for j<16, 2^(l>>4) * 16 for j >= 16):\n");
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) hb[l][j] = enc_pow2((l >> 4) + (j >= 16 ? 4 : 0));
  int Ls[] = {0, 5, 16, 37, 63};
  for (int t = 0; t < 5; ++t)
    for (int b = 0; b < 32; b += 9) {
      memset(ha, 0, sizeof(ha));
      ha[Ls[t]][b] = enc_pow2(0);
      run(0, hc);
      for (int row = 0; row < 16; ++row) {
        const float d = hc[16 * (row >> 2)][row & 3];
        if (d != 0.f) printf("  X[L%d][b%d] -> row %d value %g\n", Ls[t], b, row, d);
      }
    }
  return 0;
}
