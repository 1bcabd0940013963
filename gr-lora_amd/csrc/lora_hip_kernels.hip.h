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
//  * the spectrum is reduced to (max, argmax) in registers, then with wave64 shuffles and one
//    small LDS step; the IQ is read from HBM exactly once with coalesced 8-byte lane loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lora_hip {

struct KParams {
  const float2 *iq;             // capture (device)
  unsigned long long n_samples;
  const float2 *up, *down;      // 2N-entry chirp tables (demod_impl.cc:92-101)
  const float *win;             // N Kaiser taps (demod_impl.cc:85)
  const float2 *twc;            // [F][R]   pass-0 input constants W_(F*R)^(rr*m)           (uniform)
  const float2 *twp[3];         // per pass ps>=1: [F][r-1][NS] merged twiddles, see fft_pass
  const void *desc;             // lora_win_desc[n_win] or nullptr for the strided form
  unsigned long long pos0, stride;
  unsigned off0, off_step, kind;
  unsigned n_win;
  unsigned short *out_idx;
  float *out_val;
};

struct WinDesc { unsigned long long pos; unsigned short off; unsigned char kind; unsigned char pad[5]; };

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * (-j)
__device__ __forceinline__ float2 mul_mj(float2 a) { return make_float2(a.y, -a.x); }

// volk_32fc_x2_multiply_32fc, generic/SSE3 arithmetic: every product and sum rounded once, no
// FMA contraction (demod_impl.cc:179/183/310)
__device__ __forceinline__ float2 cmul_volk(float2 a, float2 b) {
  return make_float2(__fsub_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)),
                     __fadd_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x)));
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

template <> struct Dft<16> {
  __device__ static __forceinline__ void run(float2 *x) {
    // 4x4: u[c][k1] = DFT4 over m of x[c + 4m]
    float2 u[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      u[c][0] = x[c]; u[c][1] = x[c + 4]; u[c][2] = x[c + 8]; u[c][3] = x[c + 12];
      dft4(u[c][0], u[c][1], u[c][2], u[c][3]);
    }
    // twiddles W16^(c*k1)
    const float2 w1 = make_float2(LORA_COS_PI_8, -LORA_SIN_PI_8);
    const float2 w3 = make_float2(LORA_SIN_PI_8, -LORA_COS_PI_8);
    u[1][1] = cmul(u[1][1], w1);
    u[1][2] = make_float2((u[1][2].x + u[1][2].y) * LORA_SQRT1_2, (u[1][2].y - u[1][2].x) * LORA_SQRT1_2);
    u[1][3] = cmul(u[1][3], w3);
    u[2][1] = make_float2((u[2][1].x + u[2][1].y) * LORA_SQRT1_2, (u[2][1].y - u[2][1].x) * LORA_SQRT1_2);
    u[2][2] = mul_mj(u[2][2]);
    u[2][3] = make_float2((u[2][3].y - u[2][3].x) * LORA_SQRT1_2, -(u[2][3].x + u[2][3].y) * LORA_SQRT1_2);
    u[3][1] = cmul(u[3][1], w3);
    u[3][2] = make_float2((u[3][2].y - u[3][2].x) * LORA_SQRT1_2, -(u[3][2].x + u[3][2].y) * LORA_SQRT1_2);
    u[3][3] = cmul(u[3][3], make_float2(-LORA_COS_PI_8, LORA_SIN_PI_8));   // W16^9
    // y[k1 + 4 k2] = DFT4 over c of u[c][k1]
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
      dft4(u[0][k1], u[1][k1], u[2][k1], u[3][k1]);
      x[k1] = u[0][k1]; x[k1 + 4] = u[1][k1]; x[k1 + 8] = u[2][k1]; x[k1 + 12] = u[3][k1];
    }
  }
};

// ---- geometry ---------------------------------------------------------------------------
// Radix plan per spreading factor: N = product of the pass radices; R = points per thread.
constexpr int plan_points(int logn) { return logn == 9 ? 8 : 16; }
constexpr int plan_passes(int logn) { return logn <= 8 ? 2 : 3; }
constexpr int plan_radix(int logn, int ps) {
  switch (logn) {
    case 6:  return ps == 0 ? 16 : 4;
    case 7:  return ps == 0 ? 16 : 8;
    case 8:  return 16;
    case 9:  return 8;
    case 10: return ps == 0 ? 16 : 8;
    case 11: return ps == 2 ? 8 : 16;
    default: return 16;   // 12
  }
}

constexpr int kBlock = 256;

// LDS tile padding: one complex of padding per 16 keeps the stride-16 scatter of the first
// pass conflict-free for ds_write_b64
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
template <int LOGN, int F, int PS, int NS>
__device__ __forceinline__ void fft_pass(float2 (&v)[plan_points(LOGN)], float2 *tile, const int tw,
                                         const KParams &p, const int rr) {
  constexpr int N = 1 << LOGN, R = plan_points(LOGN), T = N / R;
  constexpr int NP = plan_passes(LOGN), r = plan_radix(LOGN, PS), nb = R / r;
  constexpr bool last = (PS == NP - 1);
  if constexpr (PS > 0) {
    __syncthreads();                 // previous pass's scatter is visible
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = tile[lds_pad(tw + k * T)];
    if constexpr (!last) __syncthreads();   // everyone has read before the tile is overwritten
  } else if constexpr (NP > 1) {
    if (rr > 0) __syncthreads();     // readers of the previous sub-transform are done
  }
#pragma unroll
  for (int q = 0; q < nb; q++) {
    const int j = tw + q * T;        // butterfly index, j < N/r
    float2 a[r];
#pragma unroll
    for (int m = 0; m < r; m++) a[m] = v[q + m * nb];
    if constexpr (PS > 0) {
      const float2 *__restrict__ tp = p.twp[PS] + (rr * (r - 1)) * NS + (j & (NS - 1));
#pragma unroll
      for (int m = 1; m < r; m++) a[m] = cmul(a[m], tp[(m - 1) * NS]);
    }
    Dft<r>::run(a);
    if constexpr (!last) {
      const int o = (j / NS) * NS * r + (j & (NS - 1));
#pragma unroll
      for (int m = 0; m < r; m++) tile[lds_pad(o + m * NS)] = a[m];
    } else {
#pragma unroll
      for (int m = 0; m < r; m++) v[q + m * nb] = a[m];
    }
  }
  if constexpr (!last) fft_pass<LOGN, F, PS + 1, NS * r>(v, tile, tw, p, rr);
}

template <int LOGN, int F>
__global__ __launch_bounds__(kBlock, 2) void demod_kernel(const KParams p) {
  constexpr int N = 1 << LOGN, R = plan_points(LOGN), T = N / R, WPB = kBlock / T;
  constexpr int PADN = N + N / 16;
  static_assert(T >= 1 && T <= kBlock && WPB >= 1, "geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *smem = reinterpret_cast<float2 *>(smem_raw);
  // second region: per-wave partial results for windows spanning several waves
  Best *red = reinterpret_cast<Best *>(smem + WPB * PADN);

  const int tid = threadIdx.x;
  const int tw = tid % T;          // thread within its window
  const int wl = tid / T;          // window within the workgroup
  const unsigned win_id = blockIdx.x * WPB + wl;
  float2 *tile = smem + wl * PADN;

  // window descriptor
  unsigned long long pos; unsigned off, kind;
  bool valid = win_id < p.n_win;
  if (p.desc != nullptr) {
    const WinDesc d = valid ? reinterpret_cast<const WinDesc *>(p.desc)[win_id] : WinDesc{0, 0, 0, {0}};
    pos = d.pos; off = d.off; kind = d.kind;
  } else {
    pos = p.pos0 + (unsigned long long)win_id * p.stride;
    off = (p.off0 + win_id * p.off_step) & (N - 1);
    kind = p.kind;
  }
  // guard: never read outside the capture or the 2N-entry tables
  const bool in_range = valid && off < (unsigned)N && pos <= p.n_samples && p.n_samples - pos >= (unsigned long long)N;

  // ---- load + dechirp + window: x[k] <-> sample n = tw + k*T ----------------------------
  float2 x[R];
  {
    const float2 *src = p.iq + (in_range ? pos : 0ull);
    const float2 *chirp = (kind == 0 ? p.down : p.up) + (in_range ? off : 0u);
#pragma unroll
    for (int k = 0; k < R; k++) {
      const int n = tw + k * T;
      float2 s = make_float2(0.f, 0.f), c = make_float2(0.f, 0.f);
      float w = 1.0f;
      if (in_range) {
        s = src[n];
        c = chirp[n];
        if (kind == 0) w = p.win[n];
      }
      float2 t = cmul_volk(s, c);                                     // :179/:183/:310
      if (kind == 0) t = make_float2(__fmul_rn(t.x, w), __fmul_rn(t.y, w));   // :192
      x[k] = t;
    }
  }

  Best best = {-1.0f, 0};
  bool bin0_nan = false;

#pragma nounroll
  for (int rr = 0; rr < F; rr++) {
    float2 v[R];
    // decimation in frequency by F: bins F*k + rr come from x[n] * exp(-2 pi i rr n / M)
    static_assert(plan_radix(LOGN, 0) == R, "pass 0 must consume all of a thread's points");
    if (rr == 0) {
#pragma unroll
      for (int k = 0; k < R; k++) v[k] = x[k];
    } else {
      const float2 *__restrict__ tc = p.twc + rr * R;     // wave-uniform -> scalar loads
      v[0] = x[0];
#pragma unroll
      for (int k = 1; k < R; k++) v[k] = cmul(x[k], tc[k]);
    }

    fft_pass<LOGN, F, 0, 1>(v, tile, tw, p, rr);
    // last pass left bin kout = tw + k*T in v[k]; scan k upwards so that strict '>' keeps the
    // lowest index of this thread's maximum for this rr
    Best b = {-1.0f, 0};
#pragma unroll
    for (int k = 0; k < R; k++) {
      const float mag = v[k].x * v[k].x + v[k].y * v[k].y;
      if (rr == 0 && k == 0 && tw == 0) bin0_nan = (mag != mag);
      if (mag > b.v) { b.v = mag; b.i = F * (tw + k * T) + rr; }
    }
    best = (rr == 0) ? b : better(best, b);
  }

  // ---- reduce over the T threads of the window --------------------------------------------
  constexpr int LANES = T < 64 ? T : 64;
#pragma unroll
  for (int s = LANES / 2; s >= 1; s >>= 1) {
    Best o;
    o.v = __shfl_xor(best.v, s, 64);
    o.i = __shfl_xor(best.i, s, 64);
    best = better(best, o);
  }
  if (T > 64) {
    constexpr int WPW = T / 64 > 0 ? T / 64 : 1;   // waves per window
    const int wave = tid / 64, lane = tid & 63;
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tw == 0) {
      Best b = red[wl * WPW];
#pragma unroll
      for (int w = 1; w < WPW; w++) b = better(b, red[wl * WPW + w]);
      best = b;
    }
  }
  if (tw == 0 && valid) {
    unsigned short oi; float ov;
    if (!in_range) { oi = 0xFFFFu; ov = -1.0f; }
    else if (bin0_nan) { oi = 0; ov = __builtin_nanf(""); }   // magsq[0] NaN: nothing ever compares greater
    else { oi = (unsigned short)best.i; ov = best.v; }
    p.out_idx[win_id] = oi;
    p.out_val[win_id] = ov;
  }
}

template <int LOGN, int F>
constexpr size_t demod_lds_bytes() {
  constexpr int N = 1 << LOGN, T = N / plan_points(LOGN), WPB = kBlock / T, PADN = N + N / 16;
  return (size_t)WPB * PADN * sizeof(float2) + (kBlock / 64) * sizeof(Best);
}
template <int LOGN>
constexpr int demod_wpb() { return kBlock / ((1 << LOGN) / plan_points(LOGN)); }

}  // namespace lora_hip
