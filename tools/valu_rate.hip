// Microbenchmark: issue rate of f32 VALU forms on gfx950 (plain vs packed), to decide whether
// the FFT butterflies should be written with v_pk_*_f32.  Prints Gops/s (lane-operations).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, A = {a, a}, B = {b, b};
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) {  // 8 x v_fma_f32
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    } else if (MODE == 1) {  // 4 x v_pk_fma_f32 (same flops as mode 0)
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A), "v"(B));
    } else if (MODE == 2) {  // 8 x v_add_f32
      x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a;
    } else if (MODE == 3) {  // 4 x v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A));
    } else if (MODE == 4) {  // 4 x v_pk_mul_f32
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, int blocks_per_cu, float *d) {
  int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double lane_elems = 5.0 * blocks * 256.0 * ITERS * 8.0;   // float results produced
  printf("%-14s blocks/CU=%d  %.1f G float-results/s  (%.3f ms)\n", name, blocks_per_cu, lane_elems / (ms * 1e-3) / 1e9, ms / 5);
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int bpc : {1, 2, 4, 8}) {
    if (bpc == 1) { run<0>("v_fma_f32", 1, d); run<1>("v_pk_fma_f32", 1, d); run<2>("v_add_f32", 1, d); run<3>("v_pk_add_f32", 1, d); run<4>("v_pk_mul_f32", 1, d); }
    if (bpc == 2) { run<0>("v_fma_f32", 2, d); run<1>("v_pk_fma_f32", 2, d); run<2>("v_add_f32", 2, d); run<3>("v_pk_add_f32", 2, d); run<4>("v_pk_mul_f32", 2, d); }
    if (bpc == 4) { run<0>("v_fma_f32", 4, d); run<1>("v_pk_fma_f32", 4, d); run<2>("v_add_f32", 4, d); run<3>("v_pk_add_f32", 4, d); run<4>("v_pk_mul_f32", 4, d); }
    if (bpc == 8) { run<0>("v_fma_f32", 8, d); run<1>("v_pk_fma_f32", 8, d); run<2>("v_add_f32", 8, d); run<3>("v_pk_add_f32", 8, d); run<4>("v_pk_mul_f32", 8, d); }
  }
  return 0;
}
