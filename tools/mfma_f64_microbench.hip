// N1 (BASELINE north_star "MFMA only for the dense per-element contractions"): does the FP64 matrix
// pipe of gfx950 pay for the element contraction of k_residual / k_jacobian?
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_microbench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
//
// The contraction of the residual kernel is, per cell,  r[(field, component)][a] = sum_{k, m} S[k][slot(field, component, m)]
// T[k][a][m]  - a [6 x 96] x [96 x 10] product (24 quadrature points x (value, 3 gradient slots); 10 P2 nodes); k_jacobian
// has the same shape with 64 columns of S.  On v_mfma_f64_16x16x4_f64 tiles two cells fill 12 of 16 rows and the nodes 10
// of 16 columns.  Measured here, all on one kernel launch per case with every SIMD of the chip busy:
//   1. peak rates of the two pipes on register operands (the guide's 78.6 / 78.6 TFLOP/s),
//   2. the two pipes fed from the same wave's instruction stream and from different waves of a SIMD (do they overlap?),
//   3. the contraction itself from LDS operands in the layout of k_residual: VALU form (the shipped one) vs MFMA form,
//   4. the contraction next to a stand-in for the flux phase (a VALU FMA chain of the flux's length): VALU + VALU vs VALU
//      + MFMA - the case the north star has in mind (matrix pipe used while the vector pipe does the pointwise physics).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NQ = 24, NSLOT = 25;

// ---- 1 / 2: raw pipes -------------------------------------------------------------------------------------------
// mode bit 0: VALU chain, bit 1: MFMA chain; waves with (wave id & split) != 0 run only the MFMA part when split != 0
__global__ __launch_bounds__(256) void k_pipes(int iters, int mode, int split, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a0 = 1.0 + lane * 1e-9, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
  const double m = 1.0000001, c = 1e-9;
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  const double ma = 1.0 + lane * 1e-6, mb = 0.5;
  const bool do_v = (mode & 1) && (split == 0 || (wave & 1) == 0);
  const bool do_m = (mode & 2) && (split == 0 || (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_v) {        // 8 independent FMA chains x 4 = 32 v_fma_f64 per iteration
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
        a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
      }
    }
    if (do_m) {        // 4 independent accumulators: 4 v_mfma_f64_16x16x4 per iteration
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc3, 0, 0, 0);
    }
  }
  const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc2[2] + acc3[3];
  if (s == 123.456) out[0] = s;
}

// ---- 3 / 4: the contraction from LDS, one cell pair per wave and round ------------------------------------------------
// sS[cell][k][slot] and sT[k][a] = (N, dN0, dN1, dN2) as in k_residual; `flux` FMAs per lane stand in for the quadrature
// phase (0: contraction only).  form 0: VALU (60 + 4 lanes, 96 FMAs each, per cell); form 1: MFMA (24 steps per pair).
template <int FORM>
__global__ __launch_bounds__(64) void k_contract(int rounds, int flux, double* out) {
  // FORM 0 (the shipped layout): sT 7.7 KB + sS 9.6 KB + sL; FORM 1: the MFMA operand images A[row][kk] (12 of 16 rows
  // used: (cell, field, component); kk = 4 k + m), B[kk][a] (10 of 16 columns) + the pressure slots: 26 KB, i.e. six
  // workgroups per CU instead of eight
  __shared__ __attribute__((aligned(32))) double4 sT[FORM == 0 ? NQ : 1][10];
  __shared__ double sL[NQ][4];
  __shared__ double sS[FORM == 0 ? 2 : 1][FORM == 0 ? NQ : 1][NSLOT];
  __shared__ double sA[FORM == 1 ? 16 : 1][NQ * 4 + 1], sB[FORM == 1 ? NQ * 4 : 1][16], sP[2][NQ];
  const int lane = threadIdx.x;
  if (FORM == 0) {
    for (int t = lane; t < NQ * 10; t += 64) sT[t / 10][t % 10] = make_double4(0.1 + t * 1e-3, 0.2, 0.3 + t * 1e-4, 0.4);
    for (int t = lane; t < 2 * NQ * NSLOT; t += 64) (&sS[0][0][0])[t] = 1.0 + t * 1e-5;
  } else {
    for (int t = lane; t < NQ * 4 * 16; t += 64) sB[t / 16][t % 16] = (t % 16) < 10 ? 0.1 + t * 1e-4 : 0.0;
    for (int t = lane; t < 16 * (NQ * 4 + 1); t += 64) (&sA[0][0])[t] = (t / (NQ * 4 + 1)) < 12 ? 1.0 + t * 1e-5 : 0.0;
  }
  for (int t = lane; t < NQ * 4; t += 64) sL[t / 4][t % 4] = 0.25;
  if (lane < 2 * NQ) sP[lane / NQ][lane % NQ] = 1.0;
  __syncthreads();
  double acc = 0.0, f0 = 1.0 + lane * 1e-9, f1 = 1.1, f2 = 1.2, f3 = 1.3;
  for (int round = 0; round < rounds; ++round) {
    // stand-in for the quadrature phase: a dependent-free VALU chain of `flux` FMAs per lane whose result feeds the contraction
    for (int i = 0; i < flux; i += 4) { f0 = f0 * 1.0000001 + 1e-9; f1 = f1 * 1.0000001 + 1e-9; f2 = f2 * 1.0000001 + 1e-9; f3 = f3 * 1.0000001 + 1e-9; }
    if (lane < 2 * NQ) {
      if (FORM == 0) sS[lane / NQ][lane % NQ][round % NSLOT] = f0 + f1;
      else sA[(lane / NQ) * 6 + round % 6][4 * (lane % NQ) + round % 4] = f0 + f1;
    }
    __syncthreads();
    if (FORM == 0) {
      for (int t = 0; t < 2; ++t) {
        double r = 0.0;
        if (lane < 60) {
          const int fld = lane / 30, comp = (lane % 30) / 10, a = lane % 10;
          const int vo = fld * 12 + comp, go = fld * 12 + 3 + 3 * comp;
#pragma unroll 4
          for (int k = 0; k < NQ; ++k) {
            const double4 tb = sT[k][a];
            r += sS[t][k][vo] * tb.x + sS[t][k][go] * tb.y + sS[t][k][go + 1] * tb.z + sS[t][k][go + 2] * tb.w;
          }
        } else {
          const int a = lane - 60;
          for (int k = 0; k < NQ; ++k) r += sS[t][k][24] * sL[k][a];
        }
        acc += r;
      }
    } else {
      v4d d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
      const int ar = lane & 15, ak = lane >> 4;
#pragma unroll 4
      for (int s = 0; s < NQ; s += 2) {            // two accumulators: consecutive MFMAs are independent
        d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[ar][4 * s + ak], sB[4 * s + ak][ar], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[ar][4 * s + 4 + ak], sB[4 * s + 4 + ak][ar], d1, 0, 0, 0);
      }
      acc += d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
      // the pressure rows (24 MACs x 4 lanes per cell) stay on the vector pipe
      if (lane < 8) { double r = 0.0; for (int k = 0; k < NQ; ++k) r += sP[lane >> 2][k] * sL[k][lane & 3]; acc += r; }
    }
    __syncthreads();
  }
  if (acc + f2 + f3 == 123.456) out[0] = acc;
}

template <class F>
double time_ms(F&& launch, int reps = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, %.0f MHz\n", prop.gcnArchName, cus, prop.clockRate / 1e3);
  double* out;
  CHECK(hipMalloc(&out, 64));
  const int iters = 20000;
  const int blocks = cus * 2;                      // 256-thread blocks: 2 per CU = 2 waves per SIMD
  auto pipes = [&](int mode, int split) {
    return time_ms([&] { hipLaunchKernelGGL(k_pipes, dim3(blocks), dim3(256), 0, 0, iters, mode, split, out); });
  };
  const double waves = (double)blocks * 4;
  const double fl_v = waves * iters * 32.0 * 64 * 2, fl_m = waves * iters * 4.0 * 2048;
  const double tv = pipes(1, 0), tm = pipes(2, 0), tb = pipes(3, 0);
  printf("1. pipes alone, 2 waves per SIMD: VALU v_fma_f64 %.2f ms = %.1f TFLOP/s | MFMA f64 16x16x4 %.2f ms = %.1f TFLOP/s\n", tv,
         fl_v / tv / 1e9, tm, fl_m / tm / 1e9);
  printf("2a. both in ONE instruction stream (every wave): %.2f ms (sum of the two alone %.2f, max %.2f) -> overlap %.0f %%\n", tb, tv + tm,
         tv > tm ? tv : tm, 100.0 * (tv + tm - tb) / (tv < tm ? tv : tm));
  const double tsv = pipes(1, 1), tsm = pipes(2, 1), tsb = pipes(3, 1);
  printf("2b. split by wave (even waves VALU, odd waves MFMA, one of each per SIMD): VALU waves alone %.2f ms, MFMA waves alone %.2f ms, "
         "together %.2f ms -> overlap %.0f %%\n", tsv, tsm, tsb, 100.0 * (tsv + tsm - tsb) / (tsv < tsm ? tsv : tsm));
  // every workgroup that is launched must be resident at once (a second batch would run alone and double the time), and
  // both forms do the same number of cell pairs: blocks = CUs x (resident workgroups per CU), rounds = pairs / blocks
  int occ0 = 0, occ1 = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, k_contract<0>, 64, 0));
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, k_contract<1>, 64, 0));
  const double pairs = 4.0e6;
  printf("   resident 64-lane workgroups per CU: VALU form %d, MFMA form %d (LDS: operand images of 26 KB)\n", occ0, occ1);
  auto contract = [&](int form, int flux) {
    const int blocks_c = cus * (form == 0 ? occ0 : occ1);
    const int rounds = (int)(pairs / blocks_c);
    const double t = time_ms([&] {
      if (form == 0) hipLaunchKernelGGL(k_contract<0>, dim3(blocks_c), dim3(64), 0, 0, rounds, flux, out);
      else hipLaunchKernelGGL(k_contract<1>, dim3(blocks_c), dim3(64), 0, 0, rounds, flux, out);
    });
    return t * pairs / ((double)blocks_c * rounds);        // normalised to exactly `pairs`
  };
  const double c0 = contract(0, 0), c1 = contract(1, 0);
  printf("3. contraction alone, LDS operands, per cell pair (chip-wide throughput): VALU form %.2f ns | MFMA form %.2f ns  (x%.2f)\n",
         1e6 * c0 / pairs, 1e6 * c1 / pairs, c1 / c0);
  // flux phase of k_residual: (55 - 11) kflop x 2 cells / 48 lanes / 2 flop ~ 920 FMAs per lane
  for (int flux : {480, 920, 1840}) {
    const double f0 = contract(0, flux), f1 = contract(1, flux);
    printf("4. flux stand-in of %4d FMAs per lane + contraction, per cell pair: VALU form %.2f ns | MFMA form %.2f ns  (x%.2f)\n", flux,
           1e6 * f0 / pairs, 1e6 * f1 / pairs, f1 / f0);
  }
  CHECK(hipFree(out));
  return 0;
}
