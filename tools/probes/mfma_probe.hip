// v_mfma_f64_16x16x4_f64 issue rate on gfx950: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void mfma_loop(double *out, int iters) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(const char *name, int blocks, int threads, double *out) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, threads>>>(out, 100);
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)blocks * threads / 64;
  double flops = waves * iters * NACC * 2048.0;
  printf("%-34s blocks %5d x %4d thr: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per wave at 2.4 GHz)\n", name, blocks, threads, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * NACC));
}
int main() {
  double *out; hipMalloc(&out, 1 << 26);
  run<4>("4 independent accumulators", 256, 256, out);     // 1 wave per SIMD
  run<4>("4 independent accumulators", 256 * 2, 256, out); // 2 waves per SIMD
  run<4>("4 independent accumulators", 256 * 4, 256, out);
  run<1>("1 accumulator (dependent chain)", 256, 256, out);
  run<2>("2 accumulators", 256, 256, out);
  run<8>("8 accumulators", 256, 256, out);
  return 0;
}
