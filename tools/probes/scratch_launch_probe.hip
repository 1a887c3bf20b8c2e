// Does a kernel that needs scratch memory cost more to LAUNCH than one that does not (MI355X, ROCm 7.2)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/scratch_probe tools/probes/scratch_launch_probe.hip && /tmp/scratch_probe
// Three kernels of the same trivial body -- no scratch, 64 B of scratch per lane, 1 KB per lane -- launched back to back
// 2000 times each (dependent launches on one stream, grid of 256 x 256 threads), alone and alternating with the
// scratch-free kernel; HIP events around each series.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_plain(double *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
template <int WORDS>
__global__ void k_scratch(double *p, int n, int sel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  volatile double priv[WORDS];  // dynamically indexed: lives in scratch
  for (int j = 0; j < WORDS; j++) priv[j] = (double)(i + j);
  if (i < n) p[i] = p[i] * 1.0000001 + priv[(sel + i) % WORDS];
}

int main() {
  const int n = 256 * 256;
  double *d;
  CHECK(hipMalloc(&d, n * sizeof(double)));
  CHECK(hipMemset(d, 0, n * sizeof(double)));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int R = 2000;
  auto run = [&](const char *name, int kind, bool alternate) -> int {
    for (int w = 0; w < 2; w++) {
      CHECK(hipEventRecord(a, s));
      for (int r = 0; r < R; r++) {
        if (kind == 0) k_plain<<<256, 256, 0, s>>>(d, n);
        if (kind == 1) k_scratch<8><<<256, 256, 0, s>>>(d, n, r);
        if (kind == 2) k_scratch<128><<<256, 256, 0, s>>>(d, n, r);
        if (alternate) k_plain<<<256, 256, 0, s>>>(d, n);
      }
      CHECK(hipEventRecord(b, s));
      CHECK(hipEventSynchronize(b));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.2f us per %s\n", name, 1e3 * ms / R, alternate ? "pair" : "launch");
    return 0;
  };
  if (run("no scratch", 0, false)) return 1;
  if (run("64 B of scratch per lane", 1, false)) return 1;
  if (run("1 KB of scratch per lane", 2, false)) return 1;
  if (run("no scratch + no scratch", 0, true)) return 1;
  if (run("64 B scratch + no scratch (alternating)", 1, true)) return 1;
  if (run("1 KB scratch + no scratch (alternating)", 2, true)) return 1;
  return 0;
}
