// LDS atomic throughput probe (MI355X): lane-operations per cycle per CU for ds_add_f64 / ds_add_f32 / ds_add_u32 /
// plain ds read-modify-write, conflict-free and with random addresses in a 512-entry array.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o tools/probes/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int RANDOM>
__global__ __launch_bounds__(1024) void probe(double *out, int reps, unsigned seed) {
  __shared__ double buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 1024) buf[i] = 0.0;
  __syncthreads();
  unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float *fb = (float *)buf;
  unsigned *ub = (unsigned *)buf;
  for (int r = 0; r < reps; r++) {
    x = x * 1664525u + 1013904223u;
    const int idx = RANDOM ? (int)((x >> 10) & 511) : (int)threadIdx.x;  // conflict-free: one address per thread
    if (MODE == 0) unsafeAtomicAdd(&buf[idx], 1.0);
    if (MODE == 1) unsafeAtomicAdd(&fb[idx], 1.0f);
    if (MODE == 2) atomicAdd(&ub[idx], 1u);
    if (MODE == 3) buf[idx] = buf[idx] + 1.0;  // plain read-modify-write (only meaningful conflict-free)
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = buf[5] + fb[7] + (double)ub[9];
}
int main() {
  double *d;
  hipMalloc(&d, 256 * sizeof(double));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int reps = 2000;
  const char *names[4] = {"ds_add_f64", "ds_add_f32", "ds_add_u32", "plain rmw f64"};
#define RUN(M, R)                                                                                         \
  {                                                                                                       \
    probe<M, R><<<256, 1024>>>(d, 10, 1);                                                                 \
    hipDeviceSynchronize();                                                                               \
    hipEventRecord(a);                                                                                    \
    probe<M, R><<<256, 1024>>>(d, reps, 1);                                                               \
    hipEventRecord(b);                                                                                    \
    hipEventSynchronize(b);                                                                               \
    float ms;                                                                                             \
    hipEventElapsedTime(&ms, a, b);                                                                       \
    const double ops = 256.0 * 1024 * reps;                                                               \
    printf("%-14s %-13s %8.3f ms  %7.2f G lane-ops/s  %6.3f lane-ops / cycle / CU (2.4 GHz)\n", names[M], \
           R ? "random(512)" : "conflict-free", ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);   \
  }
  RUN(0, 0) RUN(0, 1) RUN(1, 0) RUN(1, 1) RUN(2, 0) RUN(2, 1) RUN(3, 0)
  return 0;
}
