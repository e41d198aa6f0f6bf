// Round 4 (N1): checks, on the device, the two register layouts k_jacobian_mfma relies on:
//   1. the 4 x 4 transpose between four registers and the four 16-lane rows of a wave made of v_permlane32_swap /
//      v_permlane16_swap (gfx950): out_t[row g][n] = in_g[row t][n];
//   2. v_mfma_f64_16x16x4_f64: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
//      D[row = (lane >> 4) + 4 reg][col = lane & 15].
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_layout_check.hip -o /tmp/mfma_layout_check && /tmp/mfma_layout_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ inline void swap32(double& x, double& y) {
  const unsigned xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
  const v2u a = __builtin_amdgcn_permlane32_swap(xl, yl, false, false), b = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
  x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}
__device__ inline void swap16(double& x, double& y) {
  const unsigned xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
  const v2u a = __builtin_amdgcn_permlane16_swap(xl, yl, false, false), b = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
  x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}
__global__ void k(double* out) {
  const int l = threadIdx.x;
  double r0 = 1000 + l, r1 = 2000 + l, r2 = 3000 + l, r3 = 4000 + l;
  swap32(r0, r2); swap32(r1, r3); swap16(r0, r1); swap16(r2, r3);
  out[l] = r0; out[64 + l] = r1; out[128 + l] = r2; out[192 + l] = r3;
  // A[m][k] = 1 + m + 100 k, B[k][n] = 2 + n + 10 k
  const double a = 1.0 + (l & 15) + 100.0 * (l >> 4), b = 2.0 + (l & 15) + 10.0 * (l >> 4);
  v4d acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + 64 * r + l] = acc[r];
}
int main() {
  double* d;
  if (hipMalloc(&d, 512 * sizeof(double)) != hipSuccess) { printf("no device\n"); return 2; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(512);
  if (hipMemcpy(h.data(), d, 512 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
  int bad = 0;
  for (int t = 0; t < 4; ++t)
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, n = l & 15;
      const double want = 1000.0 * (g + 1) + 16 * t + n;          // register g of the input, taken at row t
      if (h[64 * t + l] != want) { if (bad < 8) printf("transpose: out_%d lane %d = %.0f, want %.0f\n", t, l, h[64 * t + l], want); ++bad; }
    }
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      const int row = (l >> 4) + 4 * r, col = l & 15;
      double want = 0.0;
      for (int kk = 0; kk < 4; ++kk) want += (1.0 + row + 100.0 * kk) * (2.0 + col + 10.0 * kk);
      if (std::fabs(h[256 + 64 * r + l] - want) > 1e-9) { if (bad < 16) printf("mfma: reg %d lane %d = %.1f, want %.1f\n", r, l, h[256 + 64 * r + l], want); ++bad; }
    }
  printf(bad ? "LAYOUT CHECK FAILED (%d)\n" : "layout check ok: permlane transpose and mfma_f64_16x16x4 maps as assumed\n", bad);
  return bad ? 1 : 0;
}
