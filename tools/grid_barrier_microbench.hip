// What does a device-wide barrier inside one kernel cost on gfx950, against the boundary between two dependent launches?
//
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_microbench.hip -o tools/grid_barrier_microbench && tools/grid_barrier_microbench
//
// Background ("what was not reached"): at the per-GPU size of an 8-rank run a preconditioner application is
// ~85 dependent launches of 5 - 10 us on its critical stream, each within a factor 1 - 3 of a launch boundary.  A sweep chain in
// ONE launch pays a barrier per sweep instead.  Cases:
//   1. chain of K dependent launches of a kernel that writes one value per thread and reads its neighbour workgroup's
//   2. the same exchange K times inside one launch: counter barrier, release / acquire at agent scope (cross-XCD visibility)
//   3. as 2 with the data exchanged through relaxed agent-scope atomics (sc1 stores / loads) and no cache-wide fences
// Every spin is bounded: a barrier that does not complete within ~2^22 polls sets a flag and every workgroup leaves.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_step(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = (i + 256) % n;                    // the next workgroup's entry
  out[i] = in[j] + 1.0f;
}

// bar[0]: arrival counter (monotone), bar[1]: failure flag
__device__ inline bool grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned polls = 0;
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++polls > (1u << 22) || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;      // (only thread 0's value matters: the others re-read the flag through it below)
}

template <int MODE>      // 0: plain loads / stores + release / acquire barrier; 1: relaxed agent-scope atomics for the data
__global__ __launch_bounds__(256) void k_persist(float* a, float* b, int n, int K, unsigned* bar) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = (i + 256) % n;
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  float* in = a;
  float* out = b;
  for (int k = 0; k < K; ++k) {
    float v;
    if (MODE == 0) v = in[j];
    else v = __hip_atomic_load(in + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 0) out[i] = v + 1.0f;
    else __hip_atomic_store(out + i, v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = grid_barrier(bar, (unsigned)(k + 1) * gridDim.x);
    if (threadIdx.x == 0 && !ok) dead = 1;
    __syncthreads();
    if (dead) return;
    float* t = in; in = out; out = t;
  }
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
  const int K = 2000;
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  unsigned* bar;
  CHECK(hipMalloc(&bar, 64));
  for (int G : {64, 256, 512, 1024, 2048}) {
    const int n = G * 256;
    float *a, *b;
    CHECK(hipMalloc(&a, n * sizeof(float)));
    CHECK(hipMalloc(&b, n * sizeof(float)));
    std::vector<float> h(n);
    auto reset = [&]() { CHECK(hipMemset(a, 0, n * sizeof(float))); CHECK(hipMemset(b, 0, n * sizeof(float))); CHECK(hipMemset(bar, 0, 64)); };
    auto check = [&](const char* what, float ms) {
      CHECK(hipMemcpy(h.data(), (K % 2 == 0) ? a : b, n * sizeof(float), hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < n; ++i) bad += h[i] != (float)K;
      unsigned hb[2];
      CHECK(hipMemcpy(hb, bar, sizeof hb, hipMemcpyDeviceToHost));
      printf("  G = %4d  %-58s %8.3f us per step   wrong entries %d   barrier failed %u\n", G, what, 1e3 * ms / K, bad, hb[1]);
    };
    float ms;
    // 1. launch chain
    reset();
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) reset();
      CHECK(hipEventRecord(e0, st));
      for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n);
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
    }
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    check("chain of dependent launches", ms);
    // co-residency of the persistent forms
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_persist<0>, 256, 0));
    if (G > per_cu * prop.multiProcessorCount) { printf("  G = %4d  not co-resident (%d per CU)\n", G, per_cu); continue; }
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        reset();
        CHECK(hipEventRecord(e0, st));
        if (mode == 0) hipLaunchKernelGGL(k_persist<0>, dim3(G), dim3(256), 0, st, a, b, n, K, bar);
        else hipLaunchKernelGGL(k_persist<1>, dim3(G), dim3(256), 0, st, a, b, n, K, bar);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
      }
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      check(mode == 0 ? "one launch, barrier with release / acquire, plain data" : "one launch, barrier with release / acquire, sc1 data", ms);
    }
    CHECK(hipFree(a));
    CHECK(hipFree(b));
  }
  return 0;
}
