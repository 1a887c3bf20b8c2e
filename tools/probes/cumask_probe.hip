// CU-mask probe (MI355X): hipExtStreamCreateWithCUMask as an ordinary user.
//   1. which (XCC, SE, CU) does mask bit i enable?  (one bit per stream, a kernel reports HW_ID / XCC_ID)
//   2. does a small dependent-launch chain on a stream masked to a few CUs per XCD run at its stand-alone speed
//      while a persistent chip-filling kernel holds the complementary CUs?  (the Theta-update chain beside the
//      stream-K contraction: without masks the chain's workgroups queue behind the persistent grid)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cumask_probe.hip -o tools/probes/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

__global__ void where_kernel(unsigned *out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}
// persistent filler: spins for `cycles` of the shader clock, uses 64 KB of LDS so that two workgroups fill a CU
__global__ __launch_bounds__(256) void filler_kernel(long long cycles, double *sink) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  double a = lds[threadIdx.x];
  while (wall_clock64() - t0 < cycles) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) sink[0] = a;
}
// one "chain step": 128 workgroups, each ~2 us of dependent work
__global__ __launch_bounds__(256) void step_kernel(double *buf, int n) {
  __shared__ double s[256];
  double v = buf[(blockIdx.x * 256 + threadIdx.x) % n];
  for (int i = 0; i < 200; i++) v = v * 0.999999 + 1e-7;
  s[threadIdx.x] = v;
  __syncthreads();
  buf[(blockIdx.x * 256 + threadIdx.x) % n] = s[(threadIdx.x + 1) & 255];
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.gcnArchName, ncu);
  const int words = (ncu + 31) / 32;
  unsigned *d_out;
  CK(hipMalloc(&d_out, 2 * 4096 * sizeof(unsigned)));
  std::vector<unsigned> h(2 * 4096);
  // 1. which CUs does a mask enable?  distinct (xcc, se, cu) seen by 2048 one-wave workgroups
  auto census = [&](const char *name, const std::vector<uint32_t> &mask) -> int {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
    if (e != hipSuccess) {
      printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e));
      return 2;
    }
    where_kernel<<<2048, 64, 0, s>>>(d_out);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, 2 * 2048 * sizeof(unsigned), hipMemcpyDeviceToHost));
    int cnt[8] = {0};
    std::vector<char> seen(8 * 8 * 2 * 16, 0);
    for (int b = 0; b < 2048; b++) {
      const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 7;
      const unsigned id = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15);
      if (!seen[id]) {
        seen[id] = 1;
        cnt[xcc]++;
      }
    }
    printf("%-28s distinct CUs per XCC:", name);
    for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
    printf("\n");
    CK(hipStreamDestroy(s));
    return 0;
  };
  {
    std::vector<uint32_t> m(words, 0u);
    for (int i = 0; i < 8; i++) m[i / 32] |= 1u << (i % 32);
    if (census("bits 0..7", m)) return 1;
    std::fill(m.begin(), m.end(), 0u);
    for (int i = 0; i < 32; i++) m[i / 32] |= 1u << (i % 32);
    if (census("bits 0..31", m)) return 1;
    std::fill(m.begin(), m.end(), 0u);
    for (int i = 32; i < ncu; i++) m[i / 32] |= 1u << (i % 32);
    if (census("bits 32..255", m)) return 1;
    std::fill(m.begin(), m.end(), 0xffffffffu);
    if (census("all bits", m)) return 1;
  }
  // 2. chain beside a persistent filler
  double *d_buf, *d_sink;
  CK(hipMalloc(&d_buf, 1 << 20));
  CK(hipMemset(d_buf, 0, 1 << 20));
  CK(hipMalloc(&d_sink, 64));
  CK(hipFuncSetAttribute((const void *)filler_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  hipEvent_t a, b, fa, fb;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventCreate(&fa));
  CK(hipEventCreate(&fb));
  auto run = [&](int chain_cus_per_xcd, bool masked, bool with_filler) -> int {
    // chain mask: bits i with (i / 8) < chain_cus_per_xcd (bit i -> XCC i % 8 if the KFD mapping holds)
    std::vector<uint32_t> mc(words, 0u), mg(words, 0u);
    for (int i = 0; i < ncu; i++) {
      if (i / 8 < chain_cus_per_xcd) mc[i / 32] |= 1u << (i % 32);
      else mg[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sc, sg;
    if (masked) {
      CK(hipExtStreamCreateWithCUMask(&sc, words, mc.data()));
      CK(hipExtStreamCreateWithCUMask(&sg, words, mg.data()));
    } else {
      CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
      CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    }
    const int gemm_wgs = 2 * (ncu - (masked ? 8 * chain_cus_per_xcd : 0));
    const long long cycles = 100000000LL / 1000 * 2;  // wall_clock64 ticks at 100 MHz: 2 ms
    for (int rep = 0; rep < 2; rep++) {
      if (with_filler) {
        CK(hipEventRecord(fa, sg));
        filler_kernel<<<gemm_wgs, 256, 72 * 1024, sg>>>(cycles, d_sink);
        CK(hipEventRecord(fb, sg));
      }
      CK(hipEventRecord(a, sc));
      for (int i = 0; i < 16; i++) step_kernel<<<128, 256, 0, sc>>>(d_buf, 1 << 17);
      CK(hipEventRecord(b, sc));
      CK(hipDeviceSynchronize());
    }
    float ms = 0, fms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (with_filler) CK(hipEventElapsedTime(&fms, fa, fb));
    printf("chain CUs/XCD %2d  masked %d  filler %d (%d wgs): chain of 16 steps %.3f ms, filler %.3f ms\n", chain_cus_per_xcd,
           (int)masked, (int)with_filler, gemm_wgs, ms, fms);
    CK(hipStreamDestroy(sc));
    CK(hipStreamDestroy(sg));
    return 0;
  };
  if (run(0, false, false)) return 1;
  if (run(0, false, true)) return 1;
  for (int c : {1, 2, 4, 8})
    if (run(c, true, false) || run(c, true, true)) return 1;
  return 0;
}
