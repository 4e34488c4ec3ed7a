// Hardware / runtime probe (not product code): does hipExtAnyOrderLaunch clear the
// AQL barrier bit on this part, how many dispatches of one queue run concurrently,
// are workgroups of consecutive dispatches handed out in order, and what does a
// launch boundary cost with and without the barrier bit.
//   hipcc --offload-arch=gfx950 -O2 -o any_order_probe any_order_probe.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

// 100 MHz ticks
__global__ void spin_kernel(unsigned ticks, unsigned long long* stamps, int slot) {
  const unsigned long long t0 = now();
  while (now() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = now(); }
}

// one "substep": wait until flag[w] >= seq (bounded), work for `ticks`, add 1 to
// data[w * 64 + lane] (read-modify-write: crosses the launch boundary through
// memory), publish flag[w] = seq + 1.  big LDS: two workgroups per CU.
__global__ __launch_bounds__(64) void chain_kernel(int* flag, float* data, int seq, unsigned ticks,
                                                   int* timeouts, int use_flags) {
  __shared__ float pad[20000];   // 80 KB: 2 workgroups per CU
  const int w = blockIdx.x;
  pad[threadIdx.x] = 0.0f;
  if (use_flags) {
    int spins = 0;
    while (__hip_atomic_load(flag + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
      __builtin_amdgcn_s_sleep(16);
      if (++spins > (1 << 16)) { if (threadIdx.x == 0) atomicAdd(timeouts, 1); break; }
    }
  }
  float v = data[w * 64 + threadIdx.x];
  const unsigned long long t0 = now();
  while (now() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  data[w * 64 + threadIdx.x] = v + 1.0f + pad[threadIdx.x];
  if (use_flags) {
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0)
      __hip_atomic_store(flag + w, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* stamps;
  CHECK(hipMalloc(&stamps, 64 * sizeof(unsigned long long)));
  CHECK(hipMemset(stamps, 0, 64 * sizeof(unsigned long long)));
  unsigned long long h[64];

  // 1. overlap: A spins 200 us, then B (any order) 20 us
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 20000u, stamps, 0);
  hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr,
                        hipExtAnyOrderLaunch, 2000u, stamps, 1);
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
  printf("1. A [%.1f, %.1f] us  B(any order) [%.1f, %.1f] us  -> %s\n", 0.0,
         (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0,
         ((double)h[3] - (double)h[0]) / 100.0,
         h[2] < h[1] ? "OVERLAPPED (barrier bit cleared)" : "serialised (flag ignored)");

  // 2. how many dispatches of one queue are in flight at once: 16 any-order spins of 100 us
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 100u, stamps, 0);
  for (int i = 0; i < 16; ++i)
    hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr,
                          hipExtAnyOrderLaunch, 10000u, stamps, 1 + i);
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
  int conc = 0;
  for (int i = 1; i <= 16; ++i) if (h[2 * i] < h[3]) ++conc;   // started before the first one ended
  printf("2. 16 any-order 100-us spins: %d started before the first ended; starts (us):", conc);
  for (int i = 1; i <= 16; ++i) printf(" %.0f", ((double)h[2 * i] - (double)h[2]) / 100.0);
  printf("\n");

  // 3. chains of dependent launches over 1024 workgroups (machine: 512 resident), 10 us of work
  const int W = 1024, Q = 200;
  int *flag, *timeouts;
  float* data;
  CHECK(hipMalloc(&flag, W * sizeof(int)));
  CHECK(hipMalloc(&timeouts, sizeof(int)));
  CHECK(hipMalloc(&data, W * 64 * sizeof(float)));
  for (int mode = 0; mode < 3; ++mode) {   // 0: barrier-bit chain  1: any order + flags  2: same, 512 WGs
    const int grid = mode == 2 ? 512 : W;
    CHECK(hipMemsetAsync(flag, 0, W * sizeof(int), s));
    CHECK(hipMemsetAsync(timeouts, 0, sizeof(int), s));
    CHECK(hipMemsetAsync(data, 0, W * 64 * sizeof(float), s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, s));
    for (int q = 0; q < Q; ++q) {
      if (mode == 0 || q == 0)
        hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(64), 0, s, flag, data, q, 1000u, timeouts,
                           mode != 0);
      else
        hipExtLaunchKernelGGL(chain_kernel, dim3(grid), dim3(64), 0, s, nullptr, nullptr,
                              hipExtAnyOrderLaunch, flag, data, q, 1000u, timeouts, 1);
    }
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 0u, stamps, 40);   // barrier-bit fence
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> hd(grid * 64);
    int to = 0;
    CHECK(hipMemcpy(hd.data(), data, hd.size() * sizeof(float), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&to, timeouts, sizeof(int), hipMemcpyDeviceToHost));
    int bad = 0;
    for (float v : hd) if (v != (float)Q) ++bad;
    const int rounds = (grid + 511) / 512;
    printf("3.%d %s grid %d: %.2f us per launch (work alone %d x 10 us), wrong values %d, timeouts %d\n",
           mode, mode == 0 ? "barrier-bit chain" : "any-order + per-group flags", grid,
           1000.0 * ms / Q, rounds, bad, to);
  }
  return 0;
}
