// Where do scattered f64 atomic adds execute, and how fast?  (MI355X: 8 XCDs, one L2 each, not coherent with each other)
//   agent : unsafeAtomicAdd / agent scope -> sc1: performed at the memory side so every XCD sees it
//   wg    : workgroup scope -> performed in the XCD's own L2; only correct across workgroups if every
//           XCD adds into its OWN copy of the target (index = hardware XCC_ID), copies summed afterwards
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atom_scope_probe atom_scope_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xf;
}
__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
template <int MODE>  // 0 agent, 1 workgroup scope into per-XCD copies, 2 workgroup scope into ONE copy (wrong across XCDs: rate only)
__global__ __launch_bounds__(256) void scatter(double *t, size_t n_addr, int per_thread, unsigned long long seed, unsigned *xcc_seen) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  double *base = t;
  if (MODE == 1) {
    const unsigned x = xcc_id();
    base = t + (size_t)x * n_addr;
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1u << x);
  }
  for (int i = 0; i < per_thread; i++) {
    const size_t a = mix(seed + tid * 1315423911ull + i) % n_addr;
    if (MODE == 0)
      unsafeAtomicAdd(&base[a], 1.0);
    else
      __hip_atomic_fetch_add(&base[a], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ void sum_copies(const double *t, size_t n_addr, int copies, double *out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_addr; i += (size_t)gridDim.x * 256)
    for (int c = 0; c < copies; c++) s += t[(size_t)c * n_addr + i];
  atomicAdd(out, s);
}
int main() {
  const size_t n_addr = 512 * 512;  // one H x H matrix of doubles (2 MB)
  const int copies = 16, per_thread = 16, blocks = 8192;
  double *t, *out;
  unsigned *seen;
  hipMalloc(&t, n_addr * copies * sizeof(double));
  hipMalloc(&out, sizeof(double));
  hipMalloc(&seen, sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double total = (double)blocks * 256 * per_thread;
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9;
    double got = 0;
    unsigned hseen = 0;
    for (int rep = 0; rep < 4; rep++) {
      hipMemset(t, 0, n_addr * copies * sizeof(double));
      hipMemset(out, 0, sizeof(double));
      hipMemset(seen, 0, sizeof(unsigned));
      hipEventRecord(e0);
      if (mode == 0) scatter<0><<<blocks, 256>>>(t, n_addr, per_thread, 17 + rep, seen);
      if (mode == 1) scatter<1><<<blocks, 256>>>(t, n_addr, per_thread, 17 + rep, seen);
      if (mode == 2) scatter<2><<<blocks, 256>>>(t, n_addr, per_thread, 17 + rep, seen);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      sum_copies<<<256, 256>>>(t, n_addr, copies, out);
      hipMemcpy(&got, out, sizeof(double), hipMemcpyDeviceToHost);
      hipMemcpy(&hseen, seen, sizeof(unsigned), hipMemcpyDeviceToHost);
    }
    const char *nm[3] = {"agent scope, one copy", "workgroup scope, per-XCD copies", "workgroup scope, ONE copy (rate only)"};
    printf("%-40s %8.3f ms  %7.2f G atomics/s  sum %.0f of %.0f %s  xcc mask 0x%x\n", nm[mode], best, total / best / 1e6, got, total,
           got == total ? "exact" : "MISMATCH", hseen);
  }
  return 0;
}
