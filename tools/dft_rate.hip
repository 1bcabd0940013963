// Microbenchmark: issue rate of the register-resident radix-16 butterfly + twiddle code of the demod
// kernel, free of any memory traffic, against waves per SIMD.  Instruction counts come from the ISA
// (hipcc -S), times from HIP events.
#include "../gr-lora_amd/csrc/lora_hip_kernels.hip.h"
#include <cstdio>
using namespace lora_hip;
#define ITERS 512
#ifndef LB
#define LB 3
#endif
template <int MODE> __global__ __launch_bounds__(256, LB) void k(float2 *out, float2 w1, unsigned long long *clk) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float2 v[16];
  for (int i = 0; i < 16; i++) v[i] = make_float2(threadIdx.x + i, threadIdx.x - i);
  for (int it = 0; it < ITERS; it++) {
    if (MODE >= 1) {
      float2 w[16];
      w[1] = w1;
      asm volatile("" : "+v"(w[1].x), "+v"(w[1].y));
      w[2] = csqr(w[1]); w[4] = csqr(w[2]);
      w[3] = cmul(w[1], w[2]); w[5] = cmul(w[4], w[1]); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
      w[8] = csqr(w[4]);
      for (int m = 1; m < 8; m++) w[8 + m] = cmul(w[8], w[m]);
      for (int m = 1; m < 16; m++) v[m] = cmul(v[m], w[m]);
    }
    Dft<16>::run(v);
    if (MODE >= 2) {
      Best b = {-1.0f, 0};
      for (int kk = 0; kk < 16; kk++) { const float mag = magsq(v[kk]); if (mag > b.v) { b.v = mag; b.i = kk; } }
      v[0].x += b.v; v[1].y += b.i;
    }
  }
  float2 s = make_float2(0, 0);
  for (int i = 0; i < 16; i++) s = cadd(s, v[i]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
static unsigned long long *clk;
template <int MODE> void run(const char *name, int bpc, float2 *d, int instr_per_iter) {
  int blocks = 256 * bpc;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, make_float2(0.6f, 0.8f), clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, make_float2(0.6f, 0.8f), clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per iteration per SIMD", name, bpc, ms, ms * 1e6 / (ITERS * (double)bpc));
  if (instr_per_iter) printf("  = %.2f ns per wave-instr (%d VALU per iteration)", ms * 1e6 / (ITERS * (double)bpc * instr_per_iter), instr_per_iter);
  unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  printf("  [s_memtime/s_memrealtime = %.2f]\n", (double)hc[0] / (double)hc[1]);
}
int main(int argc, char **argv) {
  int n0 = argc > 1 ? atoi(argv[1]) : 0, n1 = argc > 2 ? atoi(argv[2]) : 0, n2 = argc > 3 ? atoi(argv[3]) : 0;
  float2 *d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float2)); (void)hipMalloc(&clk, 16);
  for (int bpc : {1, 2, 3, 4}) {
    run<0>("dft16", bpc, d, n0);
    run<1>("powers+twiddle+dft16", bpc, d, n1);
    run<2>("..+magsq+argmax", bpc, d, n2);
  }
  return 0;
}
