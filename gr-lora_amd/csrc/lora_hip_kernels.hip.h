// lora_hip_kernels.hip.h -- gfx950 device code of the LoRa demod hot path.
//
// One fused kernel per (spreading factor, fft_factor): for every symbol window it does what one
// general_work call of the reference does between lib/demod_impl.cc:177 and :206 (or one
// iteration of the SFD loop, :297-322):
//   dechirp (x chirp table) -> Kaiser window -> zero-padded forward DFT of M = N*F points
//   -> |X|^2 -> argmax (lowest bin wins ties) ; only (idx, max) ever leaves the CU.
//
// MI355X mapping (not a translation of the reference's FFTW/VOLK calls):
//  * a window of N = 2^SF samples is owned by T = N/R threads, R = points per thread held in
//    VGPRs; a 256-thread workgroup therefore carries 256/T windows (1 at SF12, 16 at SF8).
//  * the zero padding is never materialised: an M = N*F point DFT of N samples is F DFTs of
//    N points of x[n]*exp(-2 pi i r n / M), r < F (decimation in frequency by F), bin F*k + r.
//  * each N-point DFT is a Stockham autosort FFT with register-resident radix-16/8/4
//    butterflies; passes exchange data through a padded LDS tile (<= 34 KiB per workgroup).
//  * the spectrum is reduced to (max, argmax) in registers, then with wave64 DPP steps and one
//    small LDS step; the IQ is read from HBM exactly once with coalesced 8-byte lane loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lora_hip {

#ifndef LORA_PF_NT
#define LORA_PF_NT 1    // sweeps of stride >= N (SF >= 8) run the instantiation whose IQ loads carry the non-temporal hint (0: never)
#endif
// buffer_load_dwordx2 through the LLVM intrinsic (the clang builtin of this toolchain narrows b64 loads to one dword):
// descriptor in four SGPRs, per-lane byte offset in one VGPR, a scalar byte offset per load
typedef int buf_rsrc_t __attribute__((ext_vector_type(4)));
typedef float buf_v2f_t __attribute__((ext_vector_type(2)));
__device__ buf_v2f_t buffer_load_v2f32(buf_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
// one IQ sample; NT: with the non-temporal hint (a sample of a sweep whose windows do not overlap is read once)
template <bool NT>
__device__ __forceinline__ float2 iq_load(const float2 *p) {
  if constexpr (NT) {
    const buf_v2f_t t = __builtin_nontemporal_load(reinterpret_cast<const buf_v2f_t *>(p));
    return make_float2(t.x, t.y);
  } else {
    return *p;
  }
}

struct KParams {
  const float2 *iq;             // capture (device)
  unsigned long long n_samples;
  const float2 *up, *down;      // 2N-entry chirp tables (demod_impl.cc:92-101)
  const float *win;             // N Kaiser taps (demod_impl.cc:85)
  const void *twp[4];           // per pass ps>=1: merged twiddles: float2 [F][r-1][NS] (LDS) or [F][NS] (base only)
  const void *desc;             // lora_win_desc[n_win] or nullptr for the strided form
  unsigned long long pos0, stride;
  unsigned off0, off_step, kind;
  unsigned n_win;
  unsigned short *out_idx;
  float *out_val;
  unsigned *ticket;             // kTicketQueues counters, 128 B apart, zero at the launch (the kernel leaves them so)
  const float2 *twist;          // any-factor kernels only: W_M^e, e < M = N*fr
  unsigned fr;                  // any-factor kernels only: the run-time fft_factor
  unsigned long long *dbg;      // diagnostic builds only (LORA_STAMPS): per-workgroup phase cycle sums
};

struct WinDesc { unsigned long long pos; unsigned short off; unsigned char kind; unsigned char pad[5]; };

// One ds_read_b64 per call.  The backend otherwise pairs LDS reads into ds_read2_b64, which moves its 1 KiB per wave in 8
// LDS cycles where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table): the exchanges of the FFT are the
// LDS's main load (measured: LDS array busy 48 % of the kernel with read2, SQ_WAIT_INST_LDS 19 % of wave cycles).
// A volatile access is never merged; the reads keep their program order, which is the order they are written in.
// Measured per spreading factor: SF12 +3-4 %, SF9 +2 %, SF11 none, SF6-SF8 and SF10 -1 % (the fixed order costs the
// small windows more than the LDS cycles gain them), hence lds_single().
constexpr bool lds_single(int logn) { return logn == 9 || logn >= 11; }
template <bool SINGLE>
__device__ __forceinline__ float2 lds_read(const float2 *p) {
  if constexpr (SINGLE) {
    typedef float lds_v2f __attribute__((ext_vector_type(2)));
    typedef const volatile __attribute__((address_space(3))) lds_v2f *lds_ptr_t;      // LDS, not a flat access
    const lds_v2f t = *(lds_ptr_t)(p);
    return make_float2(t.x, t.y);
  } else {
    return *p;
  }
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// complex multiply of the FFT (twiddles): explicit FMA form so that every kernel that shares this
// code rounds identically, whatever the compiler's contraction heuristics
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
// LORA_REF_ROUNDING 1: the two places where this kernel rounds differently from the reference's arithmetic follow it
// instead (measured cost and effect: profiles/r02_ref_rounding.md):
//   dechirp and window as VOLK's generic kernels do them -- (x*c) with four separately rounded products, one subtraction
//   and one addition, then each component times the window tap (lib/demod_impl.cc:179/:183, :192) -- instead of one FMA
//   complex multiply by the premultiplied tap;
//   |X|^2 as demod_impl::argmax does under C++11 (std::pow(float, int) is double): (float)((double)re*re + (double)im*im)
//   (:118, :125) instead of an f32 FMA.
#ifndef LORA_REF_ROUNDING
#define LORA_REF_ROUNDING 0
#endif
__device__ __forceinline__ float magsq(float2 a) {
#if LORA_REF_ROUNDING
  return (float)((double)a.x * (double)a.x + (double)a.y * (double)a.y);
#else
  return __builtin_fmaf(a.x, a.x, a.y * a.y);
#endif
}
// dechirp (+ window): tap = chirp (x window premultiplied unless LORA_REF_ROUNDING), wk = window tap or 1
__device__ __forceinline__ float2 dechirp(float2 x, float2 tap, float wk) {
#if LORA_REF_ROUNDING
  const float2 t = make_float2(x.x * tap.x - x.y * tap.y, x.x * tap.y + x.y * tap.x);
  return make_float2(t.x * wk, t.y * wk);
#else
  (void)wk;
  return make_float2(__builtin_fmaf(x.x, tap.x, -(x.y * tap.y)), __builtin_fmaf(x.x, tap.y, x.y * tap.x));
#endif
}
__device__ __forceinline__ float2 make_tap(float2 chirp, float wk) {
#if LORA_REF_ROUNDING
  (void)wk;
  return chirp;
#else
  return make_float2(chirp.x * wk, chirp.y * wk);
#endif
}
// a * (-j)
__device__ __forceinline__ float2 mul_mj(float2 a) { return make_float2(a.y, -a.x); }

// 64th roots of unity, W_64^k = exp(-2 pi i k / 64), as compile-time constants: the pass-0 input
// constants of the zero-padding twist, W_(F*R)^(rr*m), are all among them (F*R <= 64), so they
// cost no load at all.  Values are cos(2 pi k / 64) evaluated in double and rounded once.
constexpr float kCosQ64[17] = {
    1.0f, 0.99518472f, 0.980785251f, 0.956940353f, 0.923879504f, 0.881921291f, 0.831469595f, 0.773010433f,
    0.707106769f, 0.634393275f, 0.555570245f, 0.471396744f, 0.382683426f, 0.290284663f, 0.195090324f,
    0.0980171412f, 0.0f};
__device__ __forceinline__ constexpr float cos64(int k) {
  k &= 63;
  return k <= 16 ? kCosQ64[k] : k <= 32 ? -kCosQ64[32 - k] : k <= 48 ? -kCosQ64[k - 32] : kCosQ64[64 - k];
}
// a * W_64^k; k is a compile-time constant after unrolling, so the table lookups fold to literals
__device__ __forceinline__ float2 mul_w64(float2 a, int k) {
  k &= 63;
  if (k == 0) return a;
  if (k == 16) return make_float2(a.y, -a.x);     // -j
  if (k == 32) return make_float2(-a.x, -a.y);
  if (k == 48) return make_float2(-a.y, a.x);     // +j
  const float c = cos64(k), sn = -cos64(k - 16);  // W = c + j*sn, sn = -sin(2 pi k / 64)
  return make_float2(__builtin_fmaf(a.x, c, -(a.y * sn)), __builtin_fmaf(a.x, sn, a.y * c));
}
// W_64^k itself (folds to literals like the above)
__device__ __forceinline__ float2 w64c(int k) { return make_float2(cos64(k & 63), -cos64((k - 16) & 63)); }
__device__ __forceinline__ float2 csqr(float2 a) {
  const float ab = a.x * a.y;
  return make_float2(__builtin_fmaf(a.x, a.x, -(a.y * a.y)), ab + ab);
}

// ---- register-resident forward DFTs, natural order in and out ---------------------------
template <int R> struct Dft;

template <> struct Dft<2> {
  __device__ static __forceinline__ void run(float2 *x) {
    float2 a = x[0], b = x[1];
    x[0] = cadd(a, b); x[1] = csub(a, b);
  }
};

__device__ __forceinline__ void dft4(float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
  float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = mul_mj(csub(x1, x3));
  x0 = cadd(a0, a2); x2 = csub(a0, a2); x1 = cadd(a1, a3); x3 = csub(a1, a3);
}

template <> struct Dft<4> {
  __device__ static __forceinline__ void run(float2 *x) { dft4(x[0], x[1], x[2], x[3]); }
};

#define LORA_SQRT1_2 0.70710678118654752440f
#define LORA_COS_PI_8 0.92387953251128675613f
#define LORA_SIN_PI_8 0.38268343236508977173f

template <> struct Dft<8> {
  __device__ static __forceinline__ void run(float2 *x) {
    float2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    float2 o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    // W8^1 = (1-j)/sqrt2, W8^2 = -j, W8^3 = (-1-j)/sqrt2
    o1 = make_float2((o1.x + o1.y) * LORA_SQRT1_2, (o1.y - o1.x) * LORA_SQRT1_2);
    o2 = mul_mj(o2);
    o3 = make_float2((o3.y - o3.x) * LORA_SQRT1_2, -(o3.x + o3.y) * LORA_SQRT1_2);
    x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
    x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
    x[2] = cadd(e2, o2); x[6] = csub(e2, o2);
    x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
  }
};

// Butterflies with the multiply folded in (every operation an FMA): a0 = p + w*b, a1 = p - w*b in 6 instructions instead of
// the 8 of a complex multiply followed by an add and a subtract; a1 = 2p - a0 reuses the sum.  The error stays
// O(eps * max(|p|, |b|)), as for the separate form.
__device__ __forceinline__ void bfly_tw(const float2 p, const float2 w, const float2 b, float2 &a0, float2 &a1) {
  a0.x = __builtin_fmaf(-w.y, b.y, __builtin_fmaf(w.x, b.x, p.x));
  a0.y = __builtin_fmaf(w.y, b.x, __builtin_fmaf(w.x, b.y, p.y));
  a1.x = __builtin_fmaf(2.0f, p.x, -a0.x);
  a1.y = __builtin_fmaf(2.0f, p.y, -a0.y);
}
// the same with the factor -j*w: -j*(w*b) swaps the components of w*b, which the FMAs absorb
__device__ __forceinline__ void bfly_tw_mj(const float2 p, const float2 w, const float2 b, float2 &a0, float2 &a1) {
  a0.x = __builtin_fmaf(w.y, b.x, __builtin_fmaf(w.x, b.y, p.x));
  a0.y = __builtin_fmaf(w.y, b.y, __builtin_fmaf(-w.x, b.x, p.y));
  a1.x = __builtin_fmaf(2.0f, p.x, -a0.x);
  a1.y = __builtin_fmaf(2.0f, p.y, -a0.y);
}
// Twiddled DFT8 as a polynomial evaluation (see Dft<16>::run_poly): y[j] = sum_k x[k] (w*W8^j)^k from the four factors
// t = {w^4, w^2, w, w*W8}: three levels of four butterflies, 72 instructions, against 28 for seven products and 56 for the DFT8
__device__ __forceinline__ void dft8_poly(float2 *x, const float2 (&t)[4]) {
  float2 a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; k++) bfly_tw(x[k], t[0], x[k + 4], a[k], b[k]);
  float2 c[2][4];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    bfly_tw(a[k], t[1], a[k + 2], c[k][0], c[k][2]);
    bfly_tw_mj(b[k], t[1], b[k + 2], c[k][1], c[k][3]);
  }
  bfly_tw(c[0][0], t[2], c[1][0], x[0], x[4]);
  bfly_tw(c[0][1], t[3], c[1][1], x[1], x[5]);
  bfly_tw_mj(c[0][2], t[2], c[1][2], x[2], x[6]);
  bfly_tw_mj(c[0][3], t[3], c[1][3], x[3], x[7]);
}
__device__ __forceinline__ void dft8_poly_factors(const float2 w, float2 (&t)[4]) {
  t[2] = w;
  t[1] = csqr(w); t[0] = csqr(t[1]);
  t[3] = make_float2((w.x + w.y) * LORA_SQRT1_2, (w.y - w.x) * LORA_SQRT1_2);                     // * W8 = (1 - j)/sqrt2
}
#ifndef LORA_POLY_TWIST
#define LORA_POLY_TWIST 0   // 1: pass 0 of sub-transform rr >= 1 -- the zero-padding twist W_(F*R)^(rr*k) on input k is a power of
                            // one constant -- runs as one polynomial-form transform with literal factors (192 / 72
                            // instructions) instead of up to 15 / 7 constant multiplies in front of the plain DFT.
                            // Measured 0.5-1.5 % slower at every SF and F (the plain DFT's internal constants are
                            // cheaper than general factors, and several twist constants are trivial): off
#endif
// factors of the base W_64^s for dft8_poly: {c^4, c^2, c, c*W8}
__device__ __forceinline__ void twist_factors8(const int s, float2 (&t)[4]) {
  t[0] = w64c(4 * s); t[1] = w64c(2 * s); t[2] = w64c(s); t[3] = w64c(s + 8);
}
// w = (1 - j)/sqrt2 (W8^1): w*b = ((b.x + b.y), (b.y - b.x)) / sqrt2
__device__ __forceinline__ void bfly_w8_1(const float2 p, const float2 b, float2 &a0, float2 &a1) {
  const float t1 = b.x + b.y, t2 = b.y - b.x;
  a0 = make_float2(__builtin_fmaf(LORA_SQRT1_2, t1, p.x), __builtin_fmaf(LORA_SQRT1_2, t2, p.y));
  a1 = make_float2(__builtin_fmaf(-LORA_SQRT1_2, t1, p.x), __builtin_fmaf(-LORA_SQRT1_2, t2, p.y));
}
// w = (-1 - j)/sqrt2 (W8^3): w*b = ((b.y - b.x), -(b.x + b.y)) / sqrt2
__device__ __forceinline__ void bfly_w8_3(const float2 p, const float2 b, float2 &a0, float2 &a1) {
  const float t1 = b.y - b.x, t2 = b.x + b.y;
  a0 = make_float2(__builtin_fmaf(LORA_SQRT1_2, t1, p.x), __builtin_fmaf(-LORA_SQRT1_2, t2, p.y));
  a1 = make_float2(__builtin_fmaf(-LORA_SQRT1_2, t1, p.x), __builtin_fmaf(LORA_SQRT1_2, t2, p.y));
}
__device__ __forceinline__ void dft4_tail(const float2 a0, const float2 a1, const float2 a2, const float2 a3d,
                                          float2 &y0, float2 &y1, float2 &y2, float2 &y3) {
  const float2 a3 = mul_mj(a3d);
  y0 = cadd(a0, a2); y2 = csub(a0, a2); y1 = cadd(a1, a3); y3 = csub(a1, a3);
}

template <> struct Dft<16> {
  // TW: inputs 1..15 are first multiplied by w[1..15] (the pass twiddles); folded into the first butterflies
  // 150 instructions untwiddled, 194 with the 15 twiddles (164 and 224 with separate multiplies)
  // out(k1), k1 < 4: called as soon as outputs k1, k1 + 4, k1 + 8, k1 + 12 are final (the caller may store them right there, so
  // that its sixteen stores go out between the last butterflies instead of in one burst behind them)
  struct NoOut { __device__ __forceinline__ void operator()(int) const {} };
  template <bool TW, typename Out = NoOut>
  __device__ static __forceinline__ void run_impl(float2 *x, const float2 *w, const Out &out = Out()) {
    // 4x4: u[c][k1] = DFT4 over m of x[c + 4m]
    float2 u[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float2 a0, a1, a2, a3;
      if constexpr (TW) {
        const float2 p0 = (c == 0) ? x[0] : cmul(x[c], w[c]);
        const float2 p1 = cmul(x[c + 4], w[c + 4]);
        bfly_tw(p0, w[c + 8], x[c + 8], a0, a1);
        bfly_tw(p1, w[c + 12], x[c + 12], a2, a3);
      } else {
        a0 = cadd(x[c], x[c + 8]); a1 = csub(x[c], x[c + 8]);
        a2 = cadd(x[c + 4], x[c + 12]); a3 = csub(x[c + 4], x[c + 12]);
      }
      dft4_tail(a0, a1, a2, a3, u[c][0], u[c][1], u[c][2], u[c][3]);
    }
    // y[k1 + 4 k2] = DFT4 over c of u[c][k1] * W16^(c*k1)
    const float2 w1 = make_float2(LORA_COS_PI_8, -LORA_SIN_PI_8);     // W16^1
    const float2 w3 = make_float2(LORA_SIN_PI_8, -LORA_COS_PI_8);     // W16^3
    const float2 w9 = make_float2(-LORA_COS_PI_8, LORA_SIN_PI_8);     // W16^9
    {
      dft4(u[0][0], u[1][0], u[2][0], u[3][0]);
      x[0] = u[0][0]; x[4] = u[1][0]; x[8] = u[2][0]; x[12] = u[3][0];
      out(0);
    }
    {   // k1 = 1: W^1, W^2 = (1-j)/sqrt2, W^3
      float2 a0, a1, a2, a3;
      const float2 p1 = cmul(u[1][1], w1);
      bfly_w8_1(u[0][1], u[2][1], a0, a1);
      bfly_tw(p1, w3, u[3][1], a2, a3);
      dft4_tail(a0, a1, a2, a3, x[1], x[5], x[9], x[13]);
      out(1);
    }
    {   // k1 = 2: W^2 = (1-j)/sqrt2, W^4 = -j, W^6 = (-1-j)/sqrt2
      float2 a0, a1, a2, a3;
      const float2 p1 = make_float2((u[1][2].x + u[1][2].y) * LORA_SQRT1_2, (u[1][2].y - u[1][2].x) * LORA_SQRT1_2);
      const float2 b2 = mul_mj(u[2][2]);
      a0 = cadd(u[0][2], b2); a1 = csub(u[0][2], b2);
      bfly_w8_3(p1, u[3][2], a2, a3);
      dft4_tail(a0, a1, a2, a3, x[2], x[6], x[10], x[14]);
      out(2);
    }
    {   // k1 = 3: W^3, W^6 = (-1-j)/sqrt2, W^9
      float2 a0, a1, a2, a3;
      const float2 p1 = cmul(u[1][3], w3);
      bfly_w8_3(u[0][3], u[2][3], a0, a1);
      bfly_tw(p1, w9, u[3][3], a2, a3);
      dft4_tail(a0, a1, a2, a3, x[3], x[7], x[11], x[15]);
      out(3);
    }
  }
  __device__ static __forceinline__ void run(float2 *x) { run_impl<false>(x, nullptr); }
  template <typename Out>
  __device__ static __forceinline__ void run_out(float2 *x, const Out &out) { run_impl<false, Out>(x, nullptr, out); }
  __device__ static __forceinline__ void run_tw(float2 *x, const float2 *w) { run_impl<true>(x, w); }

  // The same transform as run_tw with w[k] = w^k (pass twiddles are powers of one base per butterfly), from eight
  // values instead of fifteen: y[j] = sum_k x[k] z_j^k with z_j = w * W16^j is the polynomial with coefficients x
  // evaluated at sixteen points, and splitting it into even and odd coefficients four times (z_(j+8) = -z_j,
  // z_j^2 = w^2 * W8^j, ...) leaves four levels of eight butterflies p +- t*q whose factors are w^8; w^4 and -j*w^4;
  // w^2, w^2*W8 and -j times those; w*W16^j, j < 4, and -j times those.  -j costs nothing (operands swap), so
  // t[8] = {w^8, w^4, w^2, w^2*W8, w, w*W16, w*W16^2, w*W16^3} is all it takes: 192 instructions (32 butterflies of
  // 6) and no internal constants, against 194 plus 53 for forming fifteen powers.
  template <typename Out = NoOut>
  __device__ static __forceinline__ void run_poly(float2 *x, const float2 (&t)[8], const Out &out = Out()) {
    float2 a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) bfly_tw(x[k], t[0], x[k + 8], a[k], b[k]);
    float2 c[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      bfly_tw(a[k], t[1], a[k + 4], c[k][0], c[k][2]);
      bfly_tw_mj(b[k], t[1], b[k + 4], c[k][1], c[k][3]);
    }
    float2 d[2][8];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      bfly_tw(c[k][0], t[2], c[k + 2][0], d[k][0], d[k][4]);
      bfly_tw(c[k][1], t[3], c[k + 2][1], d[k][1], d[k][5]);
      bfly_tw_mj(c[k][2], t[2], c[k + 2][2], d[k][2], d[k][6]);
      bfly_tw_mj(c[k][3], t[3], c[k + 2][3], d[k][3], d[k][7]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      bfly_tw(d[0][j], t[4 + j], d[1][j], x[j], x[j + 8]);
      bfly_tw_mj(d[0][j + 4], t[4 + j], d[1][j + 4], x[j + 4], x[j + 12]);
      out(j);
    }
  }
  // t[8] of run_poly from the base twiddle: three squarings and four products (25 instructions)
  __device__ static __forceinline__ void poly_factors(const float2 w, float2 (&t)[8]) {
    t[4] = w;
    t[2] = csqr(w); t[1] = csqr(t[2]); t[0] = csqr(t[1]);
    t[3] = make_float2((t[2].x + t[2].y) * LORA_SQRT1_2, (t[2].y - t[2].x) * LORA_SQRT1_2);       // * W8 = (1 - j)/sqrt2
    t[5] = cmul(w, make_float2(LORA_COS_PI_8, -LORA_SIN_PI_8));                                   // * W16
    t[6] = make_float2((w.x + w.y) * LORA_SQRT1_2, (w.y - w.x) * LORA_SQRT1_2);                   // * W16^2
    t[7] = cmul(w, make_float2(LORA_SIN_PI_8, -LORA_COS_PI_8));                                   // * W16^3
  }
  // the same for the constant base W_64^s (pass-0 twist): literals
  __device__ static __forceinline__ void twist_factors(const int s, float2 (&t)[8]) {
    t[0] = w64c(8 * s); t[1] = w64c(4 * s); t[2] = w64c(2 * s); t[3] = w64c(2 * s + 8);
    t[4] = w64c(s); t[5] = w64c(s + 4); t[6] = w64c(s + 8); t[7] = w64c(s + 12);
  }
};

// ---- geometry ---------------------------------------------------------------------------
// Radix plan per spreading factor: N = product of the pass radices; R = points per thread.
#ifndef LORA_SF12_R8
#define LORA_SF12_R8 0   // experiment: SF12 with 8 points per thread, 512-thread workgroups, 8.8.8.8 and 4 waves per SIMD
                         // (121 VGPRs, no spills): correct, but 27 % slower than 16 points per thread (0.545 vs 0.40 ms at F=2):
                         // one more LDS exchange per transform outweighs the even wave count
#endif
constexpr int plan_points(int logn) { return (logn == 9 || (LORA_SF12_R8 && logn == 12)) ? 8 : 16; }
constexpr int plan_passes(int logn) { return logn <= 8 ? 2 : (LORA_SF12_R8 && logn == 12) ? 4 : 3; }
constexpr int plan_radix(int logn, int ps) {
  switch (logn) {
    case 6:  return ps == 0 ? 16 : 4;
    case 7:  return ps == 0 ? 16 : 8;
    case 8:  return 16;
    case 9:  return 8;
    case 10: return ps == 0 ? 16 : 8;
    case 11: return ps == 2 ? 8 : 16;
    default: return LORA_SF12_R8 ? 8 : 16;   // 12
  }
}
constexpr int plan_ns(int logn, int ps) { int v = 1; for (int i = 0; i < ps; i++) v *= plan_radix(logn, i); return v; }

// Threads per workgroup.  SF11 at F >= 2: one window (two waves) per workgroup, so that a barrier never makes
// the waves of one window wait for another window's (+5 %); at F = 1 its LDS table (16 KiB per workgroup)
// would cost residency, so two windows share a 256-thread workgroup there.
constexpr int block_threads(int logn, int f) { return (LORA_SF12_R8 && logn == 12) ? 512 : (logn == 11 && f >= 2) ? 128 : 256; }
constexpr unsigned kTicketQueues = 8, kWaveTicketQueues = 32, kTicketStride = 32;   // counters 128 bytes apart
constexpr unsigned kTicketDoneWord = 16;      // workgroups that have finished drawing; in no counter's word
#ifndef LORA_ABL
#define LORA_ABL 0   // timing-only ablations (wrong results): 1 no IQ loads, 2 no chirp/window loads, 4 no twiddle loads, 8 no wave reduction, 16 no argmax, 32 no LDS exchanges (SF12), 64 IQ from L2
#endif
#ifndef LORA_STAMPS
#define LORA_STAMPS 0
#endif
#ifndef LORA_CLOCK
#define LORA_CLOCK 0      // diagnostic build: every workgroup records s_memtime and s_memrealtime over its lifetime (in-kernel clock)
#endif
#if LORA_STAMPS
#define LORA_STAMP_PARAMS , unsigned long long (&stamp_acc)[8], unsigned long long &stamp_last
#define LORA_STAMP_ARGS , stamp_acc, stamp_last
#define LORA_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define LORA_STAMP_PARAMS
#define LORA_STAMP_ARGS
#define LORA_STAMP(i) do { } while (0)
#endif
// scheduling fence: the backend's scheduler otherwise stretches live ranges across phases (hoisted LDS reads and
// twiddle powers) until the register budget spills
#define LORA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef LORA_FENCES
#define LORA_FENCES 0     // 1: fft_sf12 / demod_kernel<12> carry scheduling fences and form the twiddle powers right before use
#endif
#define LORA_FENCE1() do { if (LORA_FENCES) LORA_SCHED_FENCE(); } while (0)
#ifndef LORA_TWPRE
#define LORA_TWPRE 1      // pass twiddles that live in LDS are requested before the pass's barrier (their latency hides behind
                          // the scatter drain and the barrier) instead of one by one between the multiplies
#endif
#ifndef LORA_POLY16
#define LORA_POLY16 1     // SF12: the twiddled radix-16 passes run as Dft<16>::run_poly (eight factors per butterfly instead of
                          // fifteen powers: 8 instead of 15 LDS reads in pass 1, 25 instead of 53 instructions of powers in pass 2)
#endif
#ifndef LORA_POLY_GEN
#define LORA_POLY_GEN 1   // the generic passes' twiddled radix-16 and radix-8 butterflies in the same form (tables hold 8 / 4 factors)
#endif
#ifndef LORA_KTRACK
#define LORA_KTRACK 0     // 1: the per-thread argmax scan tracks the register index only (fewer instructions, but measured
                          // neutral, and the allocator spills in several variants)
#endif
#ifndef LORA_SPREAD
#define LORA_SPREAD 1     // SF12 sweep: the next window's IQ loads are spread over the last sub-transform as scalar-addressed buffer loads
#endif
#ifndef LORA_SF12_DIGITS
#define LORA_SF12_DIGITS 1
#endif
#ifndef LORA_F1_W4
#define LORA_F1_W4 0      // 1: fft_factor 1 at SF10-SF12 is cut for four waves per SIMD (128 registers)
#endif
#ifndef LORA_MIN_WAVES
#define LORA_MIN_WAVES 3
#endif
// Round-3 co-issue experiments on the SF12 kernel (profiles/r03_kernel_experiments.md); all off in the product build.
#ifndef LORA_PRIO
#define LORA_PRIO 0       // wave priority (s_setprio).  1: static, by the wave's slot on its SIMD (HW_ID.wave_id): older slots higher;
                          // 2: static, younger slots higher; 3: raised from the end of an LDS gather to the end of the butterfly
                          // block and scatter that follow; 4: raised only while a wave issues its LDS scatter and gather
#endif
#ifndef LORA_STAGGER
#define LORA_STAGGER 0    // 1: a workgroup delays its first window by (wave slot mod 3) thirds of a window period (s_sleep), so
                          // that the three waves of a SIMD do not start in phase
#endif
#ifndef LORA_EARLY_BARRIER
#define LORA_EARLY_BARRIER 0   // SF12: the barrier that protects a transform's rows from the next transform's scatter sits right behind
                               // the wave-local exchange (its reads have landed) instead of in front of the scatter, so that the
                               // scatter's stores can go out between the last butterflies of pass 0 instead of in one burst
#endif
#define LORA_PRIO_PHASE(on_butterflies) do { \
    if (LORA_PRIO == 3) __builtin_amdgcn_s_setprio((on_butterflies) ? 1 : 0); \
    else if (LORA_PRIO == 4) __builtin_amdgcn_s_setprio((on_butterflies) ? 0 : 1); } while (0)
// Waves per SIMD the register budget is cut for.  SF9 keeps 8 points per thread (72-127 VGPRs), so it can run
// six (F <= 2) or four (F = 4) instead of three: measured +17 % at F = 2, nothing at F = 1 (memory-bound).
constexpr int min_waves(int logn, int f) {
  return (LORA_SF12_R8 && logn == 12) ? 4 : (logn == 9) ? (f <= 2 ? 6 : 4) : (LORA_F1_W4 && f == 1 && logn >= 10) ? 4 : LORA_MIN_WAVES;
}

// Requesting a pass's LDS twiddles before its barrier keeps up to 30 more registers live across it: only where the
// variant's register budget takes that without spilling (checked with -Rpass-analysis=kernel-resource-usage)
constexpr bool twpre_fits(int logn, int f) { return f <= 2 && logn != 9 && logn != 11; }

// Twiddle table of pass ps >= 1 (merged with the zero-padding twist, see fft_pass).  Vector loads
// return to the registers at only ~32 B/clk per CU (measured: TD busy 84 % with tables in L2), so
// per-window table traffic is kept off that path:
//   small tables live in LDS as float2 [F][r-1][NS]      (copied once per workgroup)
//   big tables (last pass of SF12, SF11 at F >= 2, SF10 at F = 4) are not tables at all in the loop: a
//   thread's butterflies are the same for every window, so the base twiddle W^1 of each is read once
//   per launch from a float2 [F][NS] table and kept in registers; the other powers are products.
constexpr int kLdsTableBytes = 16384;
constexpr int pass_entries(int logn, int f, int ps) { return f * (plan_radix(logn, ps) - 1) * plan_ns(logn, ps); }
constexpr int lds_table_bytes_before(int logn, int f, int ps);
constexpr bool pass_table_in_lds(int logn, int f, int ps) {
  return ps > 0 && lds_table_bytes_before(logn, f, ps) + pass_entries(logn, f, ps) * 8 <= kLdsTableBytes;
}
constexpr int lds_table_bytes_before(int logn, int f, int ps) {
  int o = 0;
  for (int i = 1; i < ps; i++) if (pass_table_in_lds(logn, f, i)) o += pass_entries(logn, f, i) * 8;
  return o;
}
constexpr int pass_lds_offset(int logn, int f, int ps) {   // in float2, inside the LDS table area
  int o = 0;
  for (int i = 1; i < ps; i++) if (pass_table_in_lds(logn, f, i)) o += pass_entries(logn, f, i);
  return o;
}

// LDS tile padding: one complex of padding per 16 keeps the stride-16 scatter of the first
// pass conflict-free for ds_write_b64.  Every access is written as pad(thread part) +
// pad(compile-time part) so that the address is one VGPR plus an immediate; that split is exact
// because the low four bits of the two parts never carry (see the static_asserts at the uses).
__device__ __forceinline__ constexpr int lds_pad(int i) { return i + (i >> 4); }

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) {
  // demod_impl::argmax (:123-131): strict '>' scanning upwards == lowest index of the maximum
  const bool take_b = (b.v > a.v) || (b.v == a.v && b.i < a.i);
  return take_b ? b : a;
}

// One Stockham pass (radix r, sub-transform length NS so far) over the thread's R points, then
// the following passes.  v[k] always holds element tw + k*T of the current stage's input; the
// last pass leaves bin tw + k*T in v[k] (up to a unit-magnitude factor, see below).
//
// Zero padding by F without padding: bins F*k + rr of the M-point DFT are the N-point DFT of
// x[n] * W_M^(rr*n).  That twist is never applied per sample.  With n = tw + m*T it splits into
// W_M^(rr*m*T) (uniform constants, applied to the pass-0 inputs by the caller) and W_M^(rr*tw),
// common to a whole butterfly; by linearity the common factor moves to the butterfly's outputs
// and is merged into the NEXT pass's twiddle table:
//     twiddle(ps, rr, jm, m) = W_M^( m * (N/L_ps) * (F*jm + rr) ),  jm = j mod NS, L_ps = NS*r
// What remains after the last pass is one factor per thread of magnitude 1, which |X|^2 drops.
// A window of at most 64 threads lives inside one wave: its LDS exchanges need no workgroup barrier, only
// program order (the DS operations of a wave execute in order; wave_barrier keeps the compiler from moving
// them across), so the waves of such a workgroup never wait for each other.
#define LORA_PASS_BARRIER() do { if constexpr (T > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier(); } while (0)
template <int LOGN, int F, int PS, int NS>
__device__ __forceinline__ void fft_pass(float2 (&v)[plan_points(LOGN)], float2 *tile, const float2 *lds_tw,
                                         const int tw, const float2 *wb, const int rr LORA_STAMP_PARAMS) {
  constexpr int N = 1 << LOGN, R = plan_points(LOGN), T = N / R;
  constexpr int NP = plan_passes(LOGN), r = plan_radix(LOGN, PS), nb = R / r;
  constexpr bool last = (PS == NP - 1);
  static_assert(T % 16 == 0 || T == 8 || T == 4, "pad split: tw + k*T must not carry out of bit 3");
  static_assert(NS % 16 == 0 || NS == 8 || NS == 1, "pad split: o + m*NS must not carry out of bit 3");
  LORA_STAMP(2);                     // compute (butterflies, scatter) since the last stamp
  constexpr bool kTwPre = LORA_TWPRE && twpre_fits(LOGN, F) && PS > 0 && pass_table_in_lds(LOGN, F, PS);
  // polynomial form (Dft<16>::run_poly, dft8_poly): the table's first NF rows hold the butterfly's factors (host: create())
  constexpr bool kPoly = LORA_POLY16 && LORA_POLY_GEN && PS > 0 && (r == 16 || r == 8);
  constexpr int NF = kPoly ? r / 2 : r - 1;
  float2 tpre[kTwPre ? nb * NF : 1];
  if constexpr (kTwPre) {            // this pass's twiddles (read-only LDS table): requested before the barrier
#pragma unroll
    for (int q = 0; q < nb; q++) {
      const float2 *tp = lds_tw + pass_lds_offset(LOGN, F, PS) + (rr * (r - 1)) * NS + ((tw + q * T) & (NS - 1));
#pragma unroll
      for (int e = 0; e < NF; e++) tpre[q * NF + e] = lds_read<lds_single(LOGN)>(&tp[e * NS]);
    }
  }
  if constexpr (PS > 0) {
    LORA_PASS_BARRIER();             // previous pass's scatter is visible
    LORA_STAMP(5);                   // barrier wait
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = lds_read<lds_single(LOGN)>(&tile[lds_pad(tw) + lds_pad(k * T)]);
    LORA_STAMP(6);                   // LDS gather latency
    if constexpr (!last) LORA_PASS_BARRIER();   // everyone has read before the tile is overwritten
    LORA_STAMP(5);
  } else {
    LORA_PASS_BARRIER();             // readers of the previous sub-transform / window are done
    LORA_STAMP(5);
  }
#pragma unroll
  for (int q = 0; q < nb; q++) {
    const int j = tw + q * T;        // butterfly index, j < N/r
    float2 a[r];
#pragma unroll
    for (int m = 0; m < r; m++) a[m] = v[q + m * nb];
    if constexpr (PS > 0) {
      const int jm = j & (NS - 1);
      if constexpr (kPoly) {
        float2 t[NF];                // the butterfly's factors
        if constexpr (kTwPre) {
#pragma unroll
          for (int e = 0; e < NF; e++) t[e] = tpre[q * NF + e];
        } else if constexpr (pass_table_in_lds(LOGN, F, PS)) {
          const float2 *tp = lds_tw + pass_lds_offset(LOGN, F, PS) + (rr * (r - 1)) * NS + jm;
#pragma unroll
          for (int e = 0; e < NF; e++) t[e] = lds_read<lds_single(LOGN)>(&tp[e * NS]);
        } else {
          // base twiddle of this butterfly from the caller's registers; the other factors are products
          static_assert(last, "only the last pass has a table too big for LDS");
          float2 w1 = wb[q];
          asm volatile("" : "+v"(w1.x), "+v"(w1.y));     // loop-invariant, but hoisted factors would cost r registers
          if constexpr (r == 16) Dft<16>::poly_factors(w1, t);
          else dft8_poly_factors(w1, t);
        }
        if constexpr (r == 16) Dft<16>::run_poly(a, t);
        else dft8_poly(a, t);
      } else {
      float2 w[r];                   // w[m], m >= 1: this butterfly's pass twiddles
      if constexpr (kTwPre) {
#pragma unroll
        for (int m = 1; m < r; m++) w[m] = tpre[q * (r - 1) + m - 1];
      } else if constexpr (pass_table_in_lds(LOGN, F, PS)) {
        const float2 *tp = lds_tw + pass_lds_offset(LOGN, F, PS) + (rr * (r - 1)) * NS + jm;
#pragma unroll
        for (int m = 1; m < r; m++) w[m] = lds_read<lds_single(LOGN)>(&tp[(m - 1) * NS]);
      } else {
        // base twiddle of this butterfly from the caller's registers; every other power is a product
        static_assert(r == 8 || r == 16, "power chain covers radix 8 and 16");
        static_assert(last, "only the last pass has a table too big for LDS");
        w[1] = wb[q];
        asm volatile("" : "+v"(w[1].x), "+v"(w[1].y));   // loop-invariant, but hoisted powers would cost 2r registers
        w[2] = csqr(w[1]); w[4] = csqr(w[2]);
        w[3] = cmul(w[1], w[2]); w[5] = cmul(w[4], w[1]); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
        if constexpr (r == 16) {
          w[8] = csqr(w[4]);
#pragma unroll
          for (int m = 1; m < 8; m++) w[8 + m] = cmul(w[8], w[m]);
        }
      }
      if constexpr (r == 16) {
        Dft<16>::run_tw(a, w);       // twiddles folded into the first butterflies
      } else {
#pragma unroll
        for (int m = 1; m < r; m++) a[m] = cmul(a[m], w[m]);
        Dft<r>::run(a);
      }
      }
    } else {
      if constexpr (LORA_POLY16 && LORA_POLY_GEN && LORA_POLY_TWIST && (r == 16 || r == 8) && 64 % (F * r) == 0) {
        // pass 0: the caller left the zero-padding twist W_(F*R)^(rr*k) of sub-transform rr to this transform
        if (rr > 0) {
          const int s64 = rr * (64 / (F * r));
          if constexpr (r == 16) { float2 t[8]; Dft<16>::twist_factors(s64, t); Dft<16>::run_poly(a, t); }
          else { float2 t[4]; twist_factors8(s64, t); dft8_poly(a, t); }
        } else {
          Dft<r>::run(a);
        }
      } else {
        Dft<r>::run(a);
      }
    }
    if constexpr (!last) {
      const int o = (j / NS) * NS * r + (j & (NS - 1));
#pragma unroll
      for (int m = 0; m < r; m++) tile[lds_pad(o) + lds_pad(m * NS)] = a[m];
    } else {
#pragma unroll
      for (int m = 0; m < r; m++) v[q + m * nb] = a[m];
    }
  }
  if constexpr (!last) fft_pass<LOGN, F, PS + 1, NS * r>(v, tile, lds_tw, tw, wb, rr LORA_STAMP_ARGS);
}

template <int LOGN, int F>
constexpr int demod_lds_table_entries() { return pass_lds_offset(LOGN, F, plan_passes(LOGN)); }

// SF12 (16 x 16 x 16, one window per workgroup): the same three radix-16 passes with the data
// mapped by digits instead of the generic Stockham scatter.  n = 256*n2 + 16*n1 + n0,
// k = ka + 16*kb + 256*kc, thread = (hi, lo) = (tw >> 4, tw & 15), 4 values of hi per wave:
//   pass 0: thread (n1, n0) transforms n2 -> ka          exchange 1: all waves, tile[ka][n1][n0]
//   pass 1: thread (ka, n0) transforms n1 -> kb          exchange 2: inside each 16-lane row
//   pass 2: thread (ka, kb) transforms n0 -> kc          bin = ka + 16*kb + 256*kc
// After exchange 1 a wave reads exactly the four rows ka = 4w..4w+3 of the tile and nobody else
// does, so exchange 2 reuses those rows with wave-local ordering only (DS operations of one wave
// execute in order): two workgroup barriers per transform instead of four.  Rows are 272 = 16*17
// elements long: conflict-free for the row-contiguous accesses of exchange 1 and for the
// stride-17 transpose of exchange 2.
// hook(stage), stage 1..3: where the caller may issue a share of the next window's IQ loads (after the scatter, after
// the pass-1 twiddles, behind the wave-local exchange), so that a wave never sits behind a burst of sixteen loads
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <int F, typename Hook = NoHook>
__device__ __forceinline__ void fft_sf12(float2 (&v)[16], float2 *tile, const float2 *lds_tw, const int tw,
                                         const float2 w1_in, const int rr LORA_STAMP_PARAMS, const Hook &hook = Hook()) {
  static_assert(pass_table_in_lds(12, F, 1) && !pass_table_in_lds(12, F, 2), "table placement assumed below");
  constexpr int ROW = 272;
  const int hi = tw >> 4, lo = tw & 15;
  if (LORA_POLY16 && LORA_POLY_GEN && LORA_POLY_TWIST && rr > 0) {   // twist of sub-transform rr and DFT in one (see fft_pass)
    float2 t[8];
    Dft<16>::twist_factors(rr * (64 / (F * 16)), t);
    Dft<16>::run_poly(v, t);                         // n2 -> ka
  } else if (LORA_EARLY_BARRIER >= 2 && !(LORA_ABL & 32)) {
    // the scatter of exchange 1 (no barrier in front of it any more) rides on the last butterflies
    float2 *o = tile + tw;
    Dft<16>::run_out(v, [&](const int k1) {
      LORA_SCHED_FENCE();
#pragma unroll
      for (int m = k1; m < 16; m += 4) o[m * ROW] = v[m];
      LORA_SCHED_FENCE();
    });
  } else {
    Dft<16>::run(v);                                 // n2 -> ka
  }
  constexpr bool kTwPre12 = LORA_TWPRE && twpre_fits(12, F) && !(LORA_ABL & 32);
#ifndef LORA_P1_BASE
#define LORA_P1_BASE 0     // 1: pass 1 reads only its base twiddle from LDS and forms the other seven factors (25 instructions)
#endif
  constexpr int NT1 = LORA_POLY16 ? 8 : 15;          // rows of the pass-1 table this kernel reads (the host fills them to match)
  float2 t1[15];                                     // pass-1 twiddles (read-only table): in flight across the barrier
#if !(LORA_ABL & 32)
  LORA_STAMP(2);
  if (!(LORA_ABL & 128) && !LORA_EARLY_BARRIER) __syncthreads();   // every wave is done with the previous transform's rows
  LORA_STAMP(5);
  LORA_PRIO_PHASE(false);
  if (LORA_EARLY_BARRIER < 2) {
    float2 *o = tile + tw;
#pragma unroll
    for (int m = 0; m < 16; m++) o[m * ROW] = v[m];  // tile[ka = m][n1 = hi][n0 = lo]
  }
  if constexpr (kTwPre12) {
    const float2 *tp = lds_tw + pass_lds_offset(12, F, 1) + (rr * 15) * 16 + hi;
    if constexpr (LORA_P1_BASE && LORA_POLY16) t1[4] = lds_read<lds_single(12)>(&tp[4 * 16]);
    else {
#pragma unroll
      for (int m = 0; m < NT1; m++) t1[m] = lds_read<lds_single(12)>(&tp[m * 16]);
    }
  }
  hook(1);
  LORA_STAMP(6);
  if (!(LORA_ABL & 128)) __syncthreads();
  LORA_STAMP(5);
  {
    const float2 *i = tile + hi * ROW + lo;
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = lds_read<lds_single(12)>(&i[m * 16]);   // thread (ka = hi, n0 = lo), register n1 = m
  }
  LORA_PRIO_PHASE(true);
  LORA_STAMP(6);
#endif
  LORA_FENCE1();
  {
    if constexpr (!kTwPre12) {
      const float2 *tp = lds_tw + pass_lds_offset(12, F, 1) + (rr * 15) * 16 + hi;
#pragma unroll
      for (int m = 0; m < NT1; m++) t1[m] = lds_read<lds_single(12)>(&tp[m * 16]);
    }
    hook(2);
    if constexpr (LORA_POLY16 && LORA_P1_BASE) {
      float2 t8[8];
      Dft<16>::poly_factors(t1[4], t8);
      Dft<16>::run_poly(v, t8);
    } else if constexpr (LORA_POLY16 && LORA_EARLY_BARRIER >= 3 && !(LORA_ABL & (32 | 256))) {
      const float2 (&t8)[8] = reinterpret_cast<const float2 (&)[8]>(t1);
      float2 *o = tile + hi * ROW + lo;              // exchange 2's stores ride on the last butterflies
      Dft<16>::run_poly(v, t8, [&](const int j) {
        LORA_SCHED_FENCE();
#pragma unroll
        for (int m = j; m < 16; m += 4) o[m * 17] = v[m];
        LORA_SCHED_FENCE();
      });
    } else if constexpr (LORA_POLY16) {
      const float2 (&t8)[8] = reinterpret_cast<const float2 (&)[8]>(t1);
      Dft<16>::run_poly(v, t8);                      // n1 -> kb
    } else {
      Dft<16>::run_tw(v, t1 - 1);                    // twiddles folded into the first butterflies; n1 -> kb
    }
  }
  LORA_FENCE1();
  LORA_FENCE1();
  LORA_STAMP(2);
#if !(LORA_ABL & (32 | 256))
  LORA_PRIO_PHASE(false);
  {
    float2 *o = tile + hi * ROW + lo;
    if (!(LORA_POLY16 && LORA_EARLY_BARRIER >= 3)) {
#pragma unroll
      for (int m = 0; m < 16; m++) o[m * 17] = v[m];   // own row: [kb = m][n0 = lo], wave-local
    }
    const float2 *i = tile + hi * ROW + lo * 17;
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = lds_read<lds_single(12)>(&i[m]);        // thread (ka = hi, kb = lo), register n0 = m
  }
  LORA_PRIO_PHASE(true);
  LORA_STAMP(6);
#endif
  hook(3);
  {
    // powers of this butterfly's base twiddle w1 = W_M^(F*g + rr), g = ka + 16*kb: w1 is a per-thread
    // constant of the whole launch (kept in registers by the caller), everything else is products,
    // so that the loop holds no vector load besides the IQ prefetch
    float2 w[16];
    w[1] = w1_in;
    asm volatile("" : "+v"(w[1].x), "+v"(w[1].y));     // loop-invariant, but 30 registers of hoisted powers would spill
    if constexpr (LORA_POLY16) {
      float2 t8[8];
      Dft<16>::poly_factors(w[1], t8);
#if LORA_EARLY_BARRIER && !(LORA_ABL & (32 | 128 | 256))
      __syncthreads();                               // every wave has read its rows: the next transform may scatter at once
#endif
      Dft<16>::run_poly(v, t8);                      // n0 -> kc: v[kc] is bin g + 256*kc
      return;
    }
    w[2] = csqr(w[1]); w[4] = csqr(w[2]); w[8] = csqr(w[4]);
    w[3] = cmul(w[1], w[2]); w[5] = cmul(w[4], w[1]); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
#pragma unroll
    for (int m = 1; m < 8; m++) w[8 + m] = cmul(w[8], w[m]);
    Dft<16>::run_tw(v, w);                           // twiddles folded in; n0 -> kc: v[kc] is bin g + 256*kc
  }
}

// Everything the kernel needs to know about one window, resolved one iteration ahead.
struct WinCtx {
  unsigned long long pos;
  unsigned win_id, off, kind;
  bool valid, in_range;
};

// STRIDED: the launch has no descriptor list (known at compile time for the register-resident-taps kernels, which keeps
// descriptor loads and their waits out of the steady-state loop)
template <int N, bool STRIDED = false>
__device__ __forceinline__ WinCtx window_ctx(const KParams &p, const unsigned win_id) {
  WinCtx c;
  c.win_id = win_id;
  c.valid = win_id < p.n_win;
  unsigned off, kind;
  if (!STRIDED && p.desc != nullptr) {
    const WinDesc d = c.valid ? reinterpret_cast<const WinDesc *>(p.desc)[win_id] : WinDesc{0, 0, 0, {0}};
    c.pos = d.pos; off = d.off; kind = d.kind;
  } else {
    c.pos = p.pos0 + (unsigned long long)win_id * p.stride;
    off = (p.off0 + win_id * p.off_step) & (N - 1);
    kind = p.kind;
  }
  // guard: never read outside the capture or the tables.  An invalid window is redirected to
  // sample 0 (the host guarantees n_samples >= N) and reported as (0xFFFF, -1).
  c.in_range = c.valid && off < (unsigned)N && kind <= 1u &&
               c.pos <= p.n_samples && p.n_samples - c.pos >= (unsigned long long)N;
  if (!c.in_range) { c.pos = 0; off = 0; kind = 0; }
  c.off = off; c.kind = kind;
  return c;
}

template <int LOGN, int F>
constexpr size_t demod_lds_bytes() {
  constexpr int kBlock = block_threads(LOGN, F);
  constexpr int N = 1 << LOGN, T = N / plan_points(LOGN), WPB = kBlock / T, PADN = N + N / 16;
  return (size_t)WPB * PADN * sizeof(float2) + (size_t)demod_lds_table_entries<LOGN, F>() * sizeof(float2) +
         2 * (kBlock / 64) * sizeof(Best) + 16 /* ticket ring */;
}
// The windows of a list (SFD searches, header/payload reads) each have their own chirp offset, so their taps are read per
// window: 2 KiB per SF8 window through the vector-memory return path, the scarce one (with these loads compiled out the
// SF8 SFD sweep runs 1.2x faster, profiles/r03_sfd_shared_tile_bound.txt).  Where the LDS has room (SF <= 8: 36 N bytes
// per workgroup at three workgroups per CU) the list kernels keep both chirp tables and the window there instead.
#ifndef LORA_TAP_LDS
#define LORA_TAP_LDS 1
#endif
constexpr bool tap_lds(int logn, bool const_off, bool anyf) { return LORA_TAP_LDS && !const_off && !anyf && logn <= 8; }
template <int LOGN>
constexpr size_t demod_tap_lds_bytes() { return (size_t)(4 << LOGN) * sizeof(float2) + ((size_t)1 << LOGN) * sizeof(float); }
template <int LOGN, int F>
constexpr size_t demod_tap_lds_offset() { return (demod_lds_bytes<LOGN, F>() + 15) / 16 * 16; }

// Persistent workgroups take groups of WPB windows from a global ticket counter.  The shares must be
// dynamic: a SIMD issues VALU work for its waves in pairs, so with three resident workgroups per CU
// one of them runs up to twice as fast as the other two in VALU-bound phases (measured: equal static
// shares finish 28 % apart; profiles/r01_dft_rate_microbench.txt).  The ticket of the group that is
// two or three iterations ahead is drawn by thread 0 at the top of an iteration and published through a
// small LDS ring at its end, so that every thread finds the next group behind barriers that exist anyway.  One counter for the whole chip would serialise at ~13 ns per atomic
// (measured: 32768 groups = 0.42 ms), so there are 8 queues, chosen by blockIdx % 8 (workgroups that
// share an XCD), each owning every 8th group: static shares between XCDs, dynamic ones inside.  Software pipelining, so that a wave never waits on HBM with nothing
// else to do:
//   * the raw IQ of window i+1 is loaded into registers while the last sub-transform of window i
//     runs (its registers are those of x, dead by then);
//   * CONST_OFF (strided sweep with one chirp offset: the steady-state path): the chirp and
//     Kaiser taps of a thread are the same for every window, loaded once, kept in VGPRs;
//   * otherwise they are requested first thing in iteration i, and the reduction + store of
//     window i-1 runs while they are in flight.
//
// ANYF (fft_factor other than 1, 2, 4; the reference takes any): the same kernel with the N-point geometry and
// tables of F = 1 and a run-time number of sub-transforms p.fr.  Sub-transform rr gets the whole twist
// x[n] * W_M^(rr*n) at its input, one complex multiply per sample by a table value that L2 keeps
// (rr*n < M, so the table index needs no reduction), and then runs the plain N-point transform.
// NT: the sweep's IQ loads carry the non-temporal hint (only launched when no window overlaps another)
template <int LOGN, int F, bool CONST_OFF, bool ANYF = false, bool NT = false>
__global__ __launch_bounds__(block_threads(LOGN, F), min_waves(LOGN, F)) void demod_kernel(const KParams p) {
  static_assert(!ANYF || F == 1, "the any-factor kernel is built on the F = 1 geometry");
  constexpr int kBlock = block_threads(LOGN, F);
  constexpr int N = 1 << LOGN, R = plan_points(LOGN), T = N / R, WPB = kBlock / T;
  constexpr int PADN = N + N / 16;
  static_assert(T >= 1 && T <= kBlock && WPB >= 1, "geometry");
  static_assert(plan_radix(LOGN, 0) == R, "pass 0 must consume all of a thread's points");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *smem = reinterpret_cast<float2 *>(smem_raw);
  float2 *lds_tw = smem + WPB * PADN;
  // per-wave partial results for windows spanning several waves
  Best *red = reinterpret_cast<Best *>(lds_tw + demod_lds_table_entries<LOGN, F>());

  const int tid = threadIdx.x;
  const int tw = tid % T;          // thread within its window
  // window within the workgroup; wave-uniform when a window fills whole waves, which keeps the window
  // arithmetic (64-bit positions, validity) in scalar registers
  const int wl = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;
  float2 *tile = smem + wl * PADN;

  // once per workgroup: small twiddle tables -> LDS (visible after the first barrier of pass 0)
#pragma unroll
  for (int ps = 1; ps < plan_passes(LOGN); ps++) {
    if (pass_table_in_lds(LOGN, F, ps)) {
      const float2 *__restrict__ g = reinterpret_cast<const float2 *>(p.twp[ps]);
      for (int i = tid; i < pass_entries(LOGN, F, ps); i += kBlock) lds_tw[pass_lds_offset(LOGN, F, ps) + i] = g[i];
    }
  }

  // list kernels at SF <= 8: both chirp tables and the window -> LDS (tap_lds above; visible after the barrier below)
  constexpr bool kTapLds = tap_lds(LOGN, CONST_OFF, ANYF);
  float2 *lds_up = reinterpret_cast<float2 *>(smem_raw + demod_tap_lds_offset<LOGN, F>());   // [2N], then down [2N], then window [N]
  float2 *lds_down = lds_up + 2 * N;
  float *lds_win = reinterpret_cast<float *>(lds_down + 2 * N);
  if constexpr (kTapLds) {
    for (int i = tid; i < 2 * N; i += kBlock) { lds_up[i] = p.up[i]; lds_down[i] = p.down[i]; }
    for (int i = tid; i < N; i += kBlock) lds_win[i] = p.win[i];
  }

#if LORA_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if LORA_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
  const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz reference
#endif

  // Reduction over the T threads of a window.  Part 1 (end of window i): wave64 DPP steps, then one
  // partial per wave into red[i & 1][wave] -- no barrier.  Part 2 (inside window i+1, after its
  // first pass barriers): thread 0 of the window combines the partials and stores (idx, max).
  // The two red[] halves alternate, so a half is rewritten two windows later, behind barriers.
  auto reduce_partial = [&](Best best, const int par) -> Best {
    constexpr int LANES = T < 64 ? T : 64;
    // DPP cross-lane steps (VALU data path, a few cycles each) instead of ds_bpermute shuffles (~100 cycles each).
    // |X|^2 is never negative and never NaN here (a NaN never wins the per-thread scan), so the float order is the
    // order of the bit patterns as integers (the initial -1 sorts below everything): first the maximum by integer max
    // steps, then the lowest index among the lanes that hold it by integer min steps -- demod_impl::argmax's
    // lowest-bin rule (:123-131), in 2 instructions per step instead of 9 for a (value, index) pair.
    int vb = __builtin_bit_cast(int, best.v);
    const int own = vb;
#define LORA_DPP_MAX(ctrl, row_mask) { const int o = __builtin_amdgcn_update_dpp(vb, vb, ctrl, row_mask, 0xF, false); vb = o > vb ? o : vb; }
#define LORA_DPP_MIN(ctrl, row_mask) { const int o = __builtin_amdgcn_update_dpp(ci, ci, ctrl, row_mask, 0xF, false); ci = o < ci ? o : ci; }
    if (!(LORA_ABL & 8)) {
      if (LANES >= 2) LORA_DPP_MAX(0xB1, 0xF)      // quad_perm [1,0,3,2]: lane ^ 1
      if (LANES >= 4) LORA_DPP_MAX(0x4E, 0xF)      // quad_perm [2,3,0,1]: lane ^ 2
      if (LANES >= 8) LORA_DPP_MAX(0x141, 0xF)     // row_half_mirror: every lane of 8 has the best of 8
      if (LANES >= 16) LORA_DPP_MAX(0x140, 0xF)    // row_mirror: every lane of a row has the row's best
      if (LANES >= 32) LORA_DPP_MAX(0x142, 0xA)    // row_bcast:15 into rows 1 and 3
      if (LANES >= 64) LORA_DPP_MAX(0x143, 0xC)    // row_bcast:31 into rows 2 and 3: lane 63 has the wave's best
    }
    static_assert(LANES <= 16 || LANES == 64, "a 32-lane group would leave its result in the upper row only");
    if (LANES == 64) vb = __builtin_amdgcn_readlane(vb, 63);          // uniform
    int ci = (own == vb) ? best.i : 0x7fffffff;
    if (!(LORA_ABL & 8)) {
      if (LANES >= 2) LORA_DPP_MIN(0xB1, 0xF)
      if (LANES >= 4) LORA_DPP_MIN(0x4E, 0xF)
      if (LANES >= 8) LORA_DPP_MIN(0x141, 0xF)
      if (LANES >= 16) LORA_DPP_MIN(0x140, 0xF)
      if (LANES >= 32) LORA_DPP_MIN(0x142, 0xA)
      if (LANES >= 64) LORA_DPP_MIN(0x143, 0xC)
    }
#undef LORA_DPP_MAX
#undef LORA_DPP_MIN
    if (LANES == 64) ci = __builtin_amdgcn_readlane(ci, 63);
    best.v = __builtin_bit_cast(float, vb);
    best.i = ci;
    if (T > 64) {
      if ((tid & 63) == 0) red[par * (kBlock / 64) + tid / 64] = best;
    }
    return best;
  };
  auto combine_store = [&](const WinCtx &w, Best best, const bool bin0_nan, const int par) {
    if (tw == 0) {
      if (T > 64) {
        constexpr int WPW = T / 64 > 0 ? T / 64 : 1;   // waves per window
        Best b = red[par * (kBlock / 64) + wl * WPW];
#pragma unroll
        for (int ww = 1; ww < WPW; ww++) b = better(b, red[par * (kBlock / 64) + wl * WPW + ww]);
        best = b;
      }
      if (w.valid) {
        unsigned short oi; float ov;
        if (!w.in_range) { oi = 0xFFFFu; ov = -1.0f; }
        else if (bin0_nan) { oi = 0; ov = __builtin_nanf(""); }   // magsq[0] NaN: nothing ever compares greater
        else { oi = (unsigned short)best.i; ov = best.v; }
        p.out_idx[w.win_id] = oi;
        p.out_val[w.win_id] = ov;
      }
    }
  };

  // Work is drawn in groups.  Windows that span several waves (T > 64): a group is the WPB windows of the
  // whole workgroup, G(0) = blockIdx.x, G(i >= 1) = gridDim.x + ticket * queues + queue; a ticket is
  // published through the LDS ring at the end of an iteration and may be read once a barrier has passed: at
  // the prefetch point of the next iteration when a whole sub-transform lies before it (F = 2: two groups
  // ahead, the smallest tail), else at the top of the iteration after the next (three groups ahead; also
  // used at F = 4, where the window arithmetic at the prefetch point would spill).
  // Windows inside one wave (T <= 64): every wave is a worker of its own, a group is the 64/T windows of
  // one wave, and the ticket drawn by lane 0 one iteration earlier simply waits in a register.
  constexpr bool kWaveLocal = (T <= 64);
  constexpr int GW = kWaveLocal ? 64 / T : WPB;                    // windows per group
  constexpr unsigned kWorkers = kWaveLocal ? kBlock / 64 : 1;       // workers per workgroup
  constexpr unsigned kQueues = kWaveLocal ? kWaveTicketQueues : kTicketQueues;
  constexpr bool kLateRead = (F == 2) && !kWaveLocal;
  constexpr int LOOK = kLateRead ? 2 : 3;
  unsigned *ring = reinterpret_cast<unsigned *>(red + 2 * (kBlock / 64));
  const unsigned n_groups = (p.n_win + GW - 1) / GW;
  const unsigned n_workers = gridDim.x * kWorkers;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool drawer = kWaveLocal ? ((tid & 63) == 0) : (tid == 0);
  const unsigned my_queue = kWaveLocal ? (blockIdx.x % kTicketQueues) * kWorkers + wave : blockIdx.x % kTicketQueues;
  unsigned *const my_ticket = p.ticket + my_queue * kTicketStride;
  const int wig = kWaveLocal ? (tid & 63) / T : wl;                 // window within the group
  unsigned tk1 = 0;                                                 // wave-local: ticket of G(it + 1)
  if constexpr (kWaveLocal) {
    if (drawer) tk1 = atomicAdd(my_ticket, 1u);
  } else {
    if (drawer) {
      for (int s2 = 1; s2 < LOOK; s2++) ring[s2] = n_workers + atomicAdd(my_ticket, 1u) * kQueues + my_queue;
    }
  }
  __syncthreads();                 // the LDS twiddle tables and the ring are visible
#if LORA_PRIO == 1 || LORA_PRIO == 2 || LORA_PRIO == 5 || LORA_PRIO == 6 || LORA_STAGGER
  if constexpr (LOGN == 12) {
    // HW_REG_HW_ID (id 4), bits 3:0: the wave's slot on its SIMD; three resident workgroups per CU fill slots in dispatch order.
    // LORA_PRIO 5 / 6: the workgroup's rank in the grid instead (blockIdx.x / (gridDim.x / 3)): the same for its four waves
#if LORA_PRIO == 5 || LORA_PRIO == 6 || LORA_STAGGER == 2
    const unsigned slot = (blockIdx.x / (gridDim.x / 3u)) % 3u;
#else
    const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) % 3u;
#endif
#if LORA_PRIO == 1 || LORA_PRIO == 5
    if (slot == 0) __builtin_amdgcn_s_setprio(2); else if (slot == 1) __builtin_amdgcn_s_setprio(1);
#elif LORA_PRIO == 2 || LORA_PRIO == 6
    if (slot == 2) __builtin_amdgcn_s_setprio(2); else if (slot == 1) __builtin_amdgcn_s_setprio(1);
#endif
#if LORA_STAGGER
    // a window takes a workgroup ~14 k cycles at three workgroups per CU: a third is ~4.7 k cycles = 73 x 64
    if (slot == 1) __builtin_amdgcn_s_sleep(73);
    else if (slot == 2) { __builtin_amdgcn_s_sleep(73); __builtin_amdgcn_s_sleep(73); }
#endif
  }
#endif
  unsigned cur_g = kWaveLocal ? blockIdx.x * kWorkers + wave : blockIdx.x;
  WinCtx cur = window_ctx<N, CONST_OFF>(p, cur_g * GW + wig);
  // CONST_OFF: this thread's chirp and Kaiser taps for the whole sweep
  // (taps = chirp x Kaiser window, multiplied once: see the dechirp below)
  float2 cc[CONST_OFF ? R : 1];
  float cwin[(CONST_OFF && LORA_REF_ROUNDING) ? R : 1];
  if constexpr (CONST_OFF) {
    const float2 *__restrict__ chirp = (p.kind == 0 ? p.down : p.up) + p.off0 + tw;
    const float *__restrict__ wsrc = p.win + tw;
#pragma unroll
    for (int k = 0; k < R; k++) {
      const float2 ck = chirp[k * T];
      const float wk = (p.kind == 0) ? wsrc[k * T] : 1.0f;     // the SFD path has no window (:310)
      cc[k] = make_tap(ck, wk);
      if (LORA_REF_ROUNDING) cwin[k] = wk;
    }
  }
  // last pass with a table too big for LDS: the base twiddle W_M^((N/L)*(F*jm + rr)) of each of this
  // thread's butterflies, per sub-transform, for the whole launch
  constexpr int NPS = plan_passes(LOGN), NSL = plan_ns(LOGN, NPS - 1), NBL = R / plan_radix(LOGN, NPS - 1);
  constexpr bool kBaseRegs = !pass_table_in_lds(LOGN, F, NPS - 1);
  constexpr bool kDigits = (LOGN == 12 && LORA_SF12_DIGITS && !LORA_SF12_R8);
  float2 wb[kBaseRegs ? F * NBL : 1];
  if constexpr (kBaseRegs) {
    const float2 *__restrict__ tab = reinterpret_cast<const float2 *>(p.twp[NPS - 1]);
#pragma unroll
    for (int rr = 0; rr < F; rr++) {
#pragma unroll
      for (int q = 0; q < NBL; q++) {
        const int jm = kDigits ? ((tw >> 4) + 16 * (tw & 15)) : ((tw + q * T) & (NSL - 1));
        wb[rr * NBL + q] = tab[rr * NSL + jm];
      }
    }
  }
  float2 nx[R];                    // raw IQ of the window about to be processed: sample tw + k*T
  {
    const float2 *__restrict__ src = p.iq + cur.pos + tw;
#pragma unroll
    for (int k = 0; k < R; k++) nx[k] = iq_load<NT>(&src[k * T]);
  }
  // The store of window i is deferred into window i + 1 when the window spans several waves (the partials
  // meet behind the next window's barriers) and, measured, whenever F > 1; at F = 1 with a window inside one
  // wave storing at once frees the registers of the deferred state (SF7/SF8: +28 % / +41 %).
  constexpr bool kDeferStore = (T > 64) || (F > 1) || ANYF;
  const int FR = ANYF ? (int)p.fr : F;            // sub-transforms per window
  WinCtx prev = cur;
  Best prev_best = {-1.0f, 0};
  bool prev_nan = false, have_prev = false;
  int par = 0;

  for (unsigned it = 0; cur_g < n_groups; it++) {
    LORA_STAMP(0);
    // ---- request this window's chirp/window taps, retire the previous window meanwhile -----
    float2 c[R];
    float cw[LORA_REF_ROUNDING ? R : 1];
    if constexpr (CONST_OFF) {
#pragma unroll
      for (int k = 0; k < R; k++) { c[k] = cc[k]; if (LORA_REF_ROUNDING) cw[k] = cwin[k]; }
    } else {
      const float2 *__restrict__ chirp = (kTapLds ? (cur.kind == 0 ? lds_down : lds_up) : (cur.kind == 0 ? p.down : p.up)) + cur.off + tw;
      const float *__restrict__ wsrc = (kTapLds ? lds_win : p.win) + tw;
#pragma unroll
      for (int k = 0; k < R; k++) {
#if LORA_ABL & 2
        c[k] = make_float2(0.5f, tw * 0.001f);
#else
        const float2 ck = chirp[k * T];
        const float wk = (cur.kind == 0) ? wsrc[k * T] : 1.0f;   // the SFD path has no window (:310)
        c[k] = make_tap(ck, wk);                                  // same products as the CONST_OFF taps: both paths agree bit for bit
        if (LORA_REF_ROUNDING) cw[k] = wk;
#endif
      }
    }
    unsigned nxt_g = 0;
    WinCtx nxt;
    if constexpr (kWaveLocal) {
      nxt_g = n_workers + __builtin_amdgcn_readfirstlane(tk1) * kQueues + my_queue;   // drawn one iteration ago
      nxt = window_ctx<N, CONST_OFF>(p, nxt_g * GW + wig);
    } else if constexpr (!kLateRead) {
      nxt_g = __builtin_amdgcn_readfirstlane(ring[(it + 1) % LOOK]);   // G(it + 1); uniform: scalar window arithmetic
      nxt = window_ctx<N, CONST_OFF>(p, nxt_g * GW + wig);
    }

    // ---- dechirp + window: x[k] <-> sample n = tw + k*T -----------------------------------
    float2 x[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
      // :179/:183/:192/:310 as one complex multiply by (chirp x window): x*(c*w) instead of the reference's
      // (x*c)*w and FMA instead of VOLK's separately rounded products -- 4 VALU per sample instead of 8 and 16
      // fewer live registers; the FFT input differs from the reference's by at most ~1.5 ulp, three orders of
      // magnitude inside the 1e-4 bar (measured agreement with the oracle: profiles/r01_v13_accuracy.txt)
      x[k] = dechirp(nx[k], c[k], cw[LORA_REF_ROUNDING ? k : 0]);
    }
    LORA_FENCE1();
    LORA_STAMP(1);    // operands landed + dechirp
    // G(it + LOOK) (wave-local: G(it + 2)); lands while this group is processed.  Drawn behind the dechirp: in vmcnt order
    // a wait for the last IQ load would otherwise also be a wait for this atomic
    unsigned tk = 0;
    {
      static_assert(R <= 16, "asm operand count");
      // every IQ load has been waited for once all of x exists; only then may the atomic be issued
      if constexpr (R == 16)
        asm volatile("" :: "v"(x[0].x), "v"(x[1].x), "v"(x[2].x), "v"(x[3].x), "v"(x[4].x), "v"(x[5].x), "v"(x[6].x), "v"(x[7].x),
                     "v"(x[8].x), "v"(x[9].x), "v"(x[10].x), "v"(x[11].x), "v"(x[12].x), "v"(x[13].x), "v"(x[14].x), "v"(x[15].x));
      else
        asm volatile("" :: "v"(x[0].x), "v"(x[1].x), "v"(x[2].x), "v"(x[3].x), "v"(x[4].x), "v"(x[5].x), "v"(x[6].x), "v"(x[7].x));
    }
    if (drawer) tk = atomicAdd(my_ticket, 1u);

    Best best = {-1.0f, 0};
    bool bin0_nan = false;
#pragma unroll ANYF ? 1 : F
    for (int rr = 0; rr < FR; rr++) {
      float2 v[R];
      // decimation in frequency by F: bins F*k + rr come from x[n] * exp(-2 pi i rr n / M)
      if (rr == 0) {
#pragma unroll
        for (int k = 0; k < R; k++) v[k] = x[k];
      } else if constexpr (ANYF) {
        const float2 *__restrict__ tws = p.twist + rr * tw;         // W_M^(rr*(tw + k*T))
#pragma unroll
        for (int k = 0; k < R; k++) v[k] = cmul(x[k], tws[k * (rr * T)]);
      } else if constexpr (LORA_POLY16 && LORA_POLY_GEN && LORA_POLY_TWIST && (R == 16 || R == 8)) {
        static_assert(64 % (F * R) == 0, "pass-0 twist constants are 64th roots of unity");
#pragma unroll
        for (int k = 0; k < R; k++) v[k] = x[k];     // the twist W_M^(rr*k*T) is applied inside pass 0 (polynomial form)
      } else {
        static_assert(64 % (F * R) == 0, "pass-0 twist constants are 64th roots of unity");
#pragma unroll
        for (int k = 0; k < R; k++) v[k] = mul_w64(x[k], rr * k * (64 / (F * R)));   // W_M^(rr*k*T)
      }
      LORA_FENCE1();
      constexpr bool kSpread = LORA_SPREAD && kDigits && CONST_OFF && !ANYF;     // SF12 steady-state sweep
      if (!ANYF && rr == F - 1) {
        // x is dead from here on: prefetch the next window's raw IQ, a whole sub-transform early
        // (any-factor kernels: the compiler cannot see that in a run-time loop, x and nx would both stay
        // live and spill; they load after the loop instead)
        asm volatile("" ::: "memory");
        if constexpr (kLateRead) {
          nxt_g = __builtin_amdgcn_readfirstlane(ring[(it + 1) % LOOK]);
          nxt = window_ctx<N, CONST_OFF>(p, nxt_g * GW + wig);
        }
        if constexpr (!kSpread) {
          const float2 *__restrict__ src = p.iq + ((LORA_ABL & 64) ? 0 : nxt.pos) + tw;   // ABL 64: every window reads window 0 (L2)
#pragma unroll
          for (int k = 0; k < R; k++) {
#if LORA_ABL & 1
            nx[k] = make_float2(1.0f + tw, k);
#else
            nx[k] = iq_load<NT>(&src[k * T]);
#endif
          }
        }
      }
      // kSpread: the sixteen loads of a thread go out in four groups along the last sub-transform, as buffer loads: one
      // scalar descriptor per window (base = first sample of the window, 8N bytes), the lane's byte offset in one VGPR, the
      // register index in the scalar offset -- no vector address arithmetic at all
      buf_rsrc_t pf_rsrc;
      {
        // the window position is wave-uniform (T >= 64): say so, or every load is wrapped in a waterfall loop
        const unsigned long long a = (unsigned long long)(p.iq + ((kSpread && rr == F - 1) ? nxt.pos : 0ull));
        pf_rsrc.x = __builtin_amdgcn_readfirstlane((int)a);
        pf_rsrc.y = __builtin_amdgcn_readfirstlane((int)(a >> 32)) & 0xffff;       // stride 0
        pf_rsrc.z = N * (int)sizeof(float2);                                          // bytes: loads past the window return 0
        pf_rsrc.w = 0x00020000;                                                       // raw 32-bit data format
      }
      constexpr int kPfGroups = (LORA_SPREAD == 2 || (LORA_SPREAD == 3 && F > 1)) ? 1 : 4;   // 1: all sixteen at stage 0
      auto pf_issue = [&](const int g) {
        if constexpr (kSpread) {
          if (rr == F - 1 && g < kPfGroups) {
            LORA_SCHED_FENCE();
#pragma unroll
            for (int k = (R / kPfGroups) * g; k < (R / kPfGroups) * (g + 1); k++) {
              // cache policy 2 = nt: a sample of a stride >= N sweep is read once, and streaming it past the L2's retention is
              // worth +1.5 % at F=2 and +2.5 % at F=1 (profiles/r03_ab_nt_loads.txt); an overlapped sweep (the block's lattice,
              // stride N/16) lives on L2 hits and loses 11-15 % with the hint, and choosing at run time inside one kernel costs
              // more than the hint gives: hence two instantiations, picked by the host from the stride
              const buf_v2f_t d = buffer_load_v2f32(pf_rsrc, tw * (int)sizeof(float2), k * T * (int)sizeof(float2), NT ? 2 : 0);
              nx[k] = make_float2(d.x, d.y);
            }
            LORA_SCHED_FENCE();
          }
        }
      };
      pf_issue(0);
      const int trr = ANYF ? 0 : rr;                // row of the twiddle tables: the any-factor twist is already in v
      if constexpr (kDigits) fft_sf12<F>(v, tile, lds_tw, tw, wb[trr], trr LORA_STAMP_ARGS, pf_issue);
      else fft_pass<LOGN, F, 0, 1>(v, tile, lds_tw, tw, wb + (kBaseRegs ? trr * NBL : 0), trr LORA_STAMP_ARGS);
      LORA_STAMP(2);  // all passes of this sub-transform
      if constexpr (kDeferStore) {
        if (rr == 0 && have_prev) combine_store(prev, prev_best, prev_nan, par ^ 1);   // previous window, behind barriers
      }
      LORA_STAMP(4);
      // last pass left bin kout = tw + k*T in v[k] (SF12 digit mapping: (tw>>4) + 16*(tw&15) + 256*k); scan k upwards so
      // that strict '>' keeps the lowest index of this thread's maximum for this rr; only k is tracked in the scan
      Best b = {-1.0f, 0};
      const int kbase = kDigits ? ((tw >> 4) + 16 * (tw & 15)) : tw, kstep = kDigits ? 256 : T;
#pragma unroll
      for (int k = 0; k < R; k++) {
        const float mag = magsq(v[k]);
        if (rr == 0 && k == 0 && tw == 0) bin0_nan = (mag != mag);      // thread 0 holds bin 0 in both mappings
#if LORA_ABL & 16
        b.v += mag;
#else
        if (mag > b.v) { b.v = mag; b.i = LORA_KTRACK ? k : FR * (kbase + kstep * k) + rr; }
#endif
      }
      if (LORA_KTRACK) b.i = FR * (kbase + kstep * b.i) + rr;         // bin FR*kout + rr of the padded spectrum
      best = (rr == 0) ? b : better(best, b);
      LORA_FENCE1();
      LORA_STAMP(3);  // |X|^2 + per-thread argmax
      asm volatile("" ::: "memory");   // keep the next sub-transform's loads out of this one
    }
    if constexpr (ANYF) {
      const float2 *__restrict__ src = p.iq + nxt.pos + tw;
#pragma unroll
      for (int k = 0; k < R; k++) nx[k] = src[k * T];
    }
    if constexpr (kDeferStore) {
      prev_best = reduce_partial(best, par);
      prev = cur; prev_nan = bin0_nan; have_prev = true;
    } else {
      combine_store(cur, reduce_partial(best, par), bin0_nan, par);   // a window inside one wave is final already
    }
    // The ticket is first touched here: the compiler must not form anything from it at the draw (it sinks such arithmetic
    // into the drawing branch, and the wait for the atomic -- which in vmcnt order is also a wait for the IQ prefetch --
    // would then stall thread 0's wave for an atomic round trip at the top of every iteration, with the workgroup's
    // other waves waiting for it at the next barrier).  Built with -amdgpu-atomic-optimizer-strategy=None for the same reason.
    asm volatile("" : "+v"(tk));
    if constexpr (kWaveLocal) tk1 = tk;
    else if (drawer) ring[it % LOOK] = n_workers + tk * kQueues + my_queue;   // read LOOK - 1 iterations from now, behind barriers
    cur = nxt; cur_g = nxt_g;
    par ^= 1;
  }
  if (have_prev) {
    if (T > 64) __syncthreads();
    combine_store(prev, prev_best, prev_nan, par ^ 1);
  }
  // The counters go back to zero for the next launch on this buffer: a workgroup checks out once all its waves have
  // drawn their last ticket, and the workgroup that checks out last -- nobody draws any more -- clears the counters and
  // the check-out count.  This replaces a 4 KiB memset in front of every launch (a kernel of its own: 4.7 us plus a
  // dispatch gap, 1.5-2 % of an SF12 launch and a fifth of a small one).
  __syncthreads();
  if (tid == 0) {
    unsigned *const done = p.ticket + kTicketDoneWord;
    if (atomicAdd(done, 1u) == gridDim.x - 1) {
      for (unsigned q = 0; q < kWaveTicketQueues; q++) atomicExch(p.ticket + q * kTicketStride, 0u);
      atomicExch(done, 0u);
    }
  }
#if LORA_CLOCK
  if (p.dbg && tid == 0) {
    p.dbg[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - clk_t0;
    p.dbg[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
#if LORA_STAMPS
  stamp_acc[7] = __builtin_amdgcn_s_memrealtime() - stamp_rt0;
  if (p.dbg && (tid & 63) == 0) {
    for (int i = 0; i < 8; i++) p.dbg[(blockIdx.x * (kBlock / 64) + tid / 64) * 8 + i] = stamp_acc[i];
  }
#endif
}

template <int LOGN, int F>
constexpr int demod_wpb() { return block_threads(LOGN, F) / ((1 << LOGN) / plan_points(LOGN)); }

}  // namespace lora_hip
