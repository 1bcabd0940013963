// Microbenchmark: VALU issue rate of one wave against the number of independent dependency chains,
// and of several waves per SIMD (gfx950).  Prints ns and cycles (s_memtime) per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
template <int CH> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < 16 / CH; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) x[c] = __builtin_fmaf(x[c], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < CH; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH> void run(int bpc, float *d, unsigned long long *c) {
  int blocks = 256 * bpc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<blocks, 256>>>(d, c, 1.0001f, 0.5f); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CH><<<blocks, 256>>>(d, c, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  double per_wave_instr = (double)ITERS * 16;
  printf("chains=%2d waves/SIMD=%d  %.2f ns per wave-instr per SIMD, %.2f s_memtime ticks per instr of one wave\n", CH, bpc,
         ms * 1e6 / (per_wave_instr * bpc), (double)hc / per_wave_instr);
}
int main() {
  float *d; unsigned long long *c; hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); hipMalloc(&c, 8);
  for (int bpc : {1, 2, 3, 4}) {
    run<1>(bpc, d, c); run<2>(bpc, d, c); run<4>(bpc, d, c); run<8>(bpc, d, c); run<16>(bpc, d, c);
  }
  return 0;
}
