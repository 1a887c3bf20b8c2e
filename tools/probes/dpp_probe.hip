// DPP row_newbcast availability on gfx950 (lane 3 of every 16-lane row to the whole row).
// hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
  int v = threadIdx.x;
  o[threadIdx.x] = __builtin_amdgcn_update_dpp(v, v, 0x153, 0xF, 0xF, false);  // row_newbcast:3
}
int main() {
  int *d, h[64];
  if (hipMalloc(&d, 256) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  if (hipMemcpy(h, d, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int i = 0; i < 64; i++) printf("%d ", h[i]);  // expected 3 x16, 19 x16, 35 x16, 51 x16
  printf("\n");
  return 0;
}
