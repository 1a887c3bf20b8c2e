// Shader clock and barrier cost probe (gfx950).  hipcc --offload-arch=gfx950 -O3 clk_probe.hip -o clk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void chain(double *out, long long *cyc, int iters) {
  double x = out[0], y = 1.0000001;
  long long w0 = wall_clock64();
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) x = fma(x, y, 1e-9);
  long long c1 = clock64();
  long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void barriers(double *out, long long *cyc, int iters) {
  __shared__ double sh[1024];
  long long w0 = wall_clock64();
  long long c0 = clock64();
  double x = threadIdx.x;
  for (int i = 0; i < iters; i++) {
    sh[threadIdx.x] = x;
    __syncthreads();
    x += sh[(threadIdx.x + 64) & (blockDim.x - 1)];
  }
  long long c1 = clock64();
  long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
  out[threadIdx.x] = x;
}
int main() {
  double *out; long long *cyc, h[2];
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 16);
  hipMemset(out, 0, 1 << 24);
  int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
  int sclk = 0; hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, max shader clock %d kHz\n", wclk, sclk);
  for (int blocks : {1, 2, 256, 2048}) {
    for (int rep = 0; rep < 3; rep++) {
      chain<<<blocks, 64>>>(out, cyc, 200000);
      hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      double us = h[1] * 1e3 / wclk;
      printf("chain blocks=%4d: %lld clock64 ticks, %.1f us -> clock64 %.0f MHz, %.2f ns per dependent f64 fma\n", blocks, h[0], us, h[0] / us, us * 1e3 / 200000);
    }
  }
  for (int threads : {128, 512, 1024}) {
    barriers<<<1, threads>>>(out, cyc, 20000);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    double us = h[1] * 1e3 / wclk;
    printf("barrier loop threads=%4d: %.1f ns per LDS write + barrier + LDS read iteration\n", threads, us * 1e3 / 20000);
  }
  return 0;
}
