/* chan_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.  See chan_oracle.h for what it restates and why
 * its parity is unpinned. */
#include "chan_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- taps: pfb.py arb_resampler_ccf.create_taps (rate < 1 branch), firdes::low_pass_2 ------------------- */
/* window.cc: blackman_harris(ntaps, atten = 92) -> coswindow(ntaps, c0, c1, c2, c3), float arithmetic on a
 * float M = ntaps - 1 */
static void blackman_harris(int ntaps, float *w) {
  const float c0 = 0.35875f, c1 = 0.48829f, c2 = 0.14128f, c3 = 0.01168f;
  const float M = (float)(ntaps - 1);
  for (int n = 0; n < ntaps; n++)
    w[n] = (float)(c0 - c1 * cos((2.0 * M_PI * n) / M) + c2 * cos((4.0 * M_PI * n) / M) - c3 * cos((6.0 * M_PI * n) / M));
}

size_t lo_chan_design_taps(double rate, unsigned nfilts, double atten_db, float *taps, size_t cap) {
  /* pfb.py: percent = 0.80; halfband = 0.5*rate; bw = percent*halfband; tb = (percent/2.0)*halfband;
   * firdes.low_pass_2(flt_size, flt_size, bw, tb, atten, WIN_BLACKMAN_HARRIS) */
  const double percent = 0.80, halfband = 0.5 * rate, bw = percent * halfband, tb = (percent / 2.0) * halfband;
  const double gain0 = nfilts, fs = nfilts;
  /* firdes::compute_ntaps_windes */
  int ntaps = (int)(atten_db * fs / (22.0 * tb));
  if ((ntaps & 1) == 0) ntaps++;
  float *w = (float *)malloc(sizeof(float) * (size_t)ntaps);
  float *t = (float *)malloc(sizeof(float) * (size_t)ntaps);
  if (!w || !t) { free(w); free(t); return 0; }
  blackman_harris(ntaps, w);
  const int M = (ntaps - 1) / 2;
  const double fwT0 = 2 * M_PI * bw / fs;
  for (int n = -M; n <= M; n++) {
    if (n == 0) t[n + M] = (float)(fwT0 / M_PI * w[n + M]);
    else t[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w[n + M]);
  }
  double fmax = t[0 + M];
  for (int n = 1; n <= M; n++) fmax += 2 * t[n + M];
  const double gain = gain0 / fmax;
  for (int i = 0; i < ntaps; i++) t[i] = (float)(t[i] * gain);
  for (size_t i = 0; i < (size_t)ntaps && i < cap; i++) taps[i] = t[i];
  free(w); free(t);
  return (size_t)ntaps;
}

/* ---- rotator.h ------------------------------------------------------------------------------------------ */
void lo_rotator(double phase_inc, const lo_c32 *in, lo_c32 *out, size_t n, lo_c32 *phase, unsigned *counter) {
  /* rotator_cc_impl: d_r.set_phase_incr(exp(gr_complex(0, phase_inc))); set_phase_incr: incr / abs(incr) */
  const float pf = (float)phase_inc;
  float ir = cosf(pf), ii = sinf(pf);
  const float mag = hypotf(ir, ii);
  ir /= mag; ii /= mag;
  float pr = phase->re, pi = phase->im;
  unsigned c = *counter;
  for (size_t k = 0; k < n; k++) {
    c++;
    const float xr = in[k].re, xi = in[k].im;
    out[k].re = xr * pr - xi * pi;          /* z = in * d_phase */
    out[k].im = xr * pi + xi * pr;
    const float nr = pr * ir - pi * ii, ni = pr * ii + pi * ir;   /* d_phase *= d_phase_incr */
    pr = nr; pi = ni;
    if ((c % 512) == 0) { const float a = hypotf(pr, pi); pr /= a; pi /= a; }
  }
  phase->re = pr; phase->im = pi; *counter = c;
}

/* ---- pfb_arb_resampler.cc ------------------------------------------------------------------------------- */
struct lo_pfb {
  unsigned int_rate, dec_rate, taps_per_filter, last_filter;
  float flt_rate, acc;
  float *filt;      /* [int_rate][taps_per_filter], already reversed as fir_filter::set_taps stores them */
  float *dfilt;
  lo_c32 *hist;     /* taps_per_filter - 1 samples before the next input */
  size_t skip;      /* inputs of the next call already stepped over (i_in ran past n_to_read) */
};

static void make_bank(const float *taps, size_t ntaps, unsigned int_rate, unsigned tpf, float *bank) {
  /* create_taps: zero-fill to int_rate*tpf, ourtaps[i][j] = tmp[i + j*int_rate]; fir_filter::set_taps
   * reverses: d_taps[k] = ourtaps[i][tpf-1-k] */
  for (unsigned i = 0; i < int_rate; i++)
    for (unsigned j = 0; j < tpf; j++) {
      const size_t src = (size_t)i + (size_t)j * int_rate;
      bank[(size_t)i * tpf + (tpf - 1 - j)] = src < ntaps ? taps[src] : 0.0f;
    }
}

lo_pfb *lo_pfb_create(float rate, const float *taps, size_t ntaps, unsigned nfilts) {
  if (!(rate > 0.0f) || ntaps < 2 || nfilts == 0) return NULL;
  lo_pfb *p = (lo_pfb *)calloc(1, sizeof(*p));
  if (!p) return NULL;
  p->acc = 0;
  p->int_rate = nfilts;
  /* set_rate */
  p->dec_rate = (unsigned int)floor(p->int_rate / rate);
  p->flt_rate = (p->int_rate / rate) - p->dec_rate;
  p->last_filter = (unsigned)((ntaps / 2) % nfilts);
  p->taps_per_filter = (unsigned int)ceil((double)ntaps / (double)p->int_rate);
  const size_t bank = (size_t)p->int_rate * p->taps_per_filter;
  p->filt = (float *)malloc(sizeof(float) * bank);
  p->dfilt = (float *)malloc(sizeof(float) * bank);
  float *dt = (float *)malloc(sizeof(float) * ntaps);
  p->hist = (lo_c32 *)calloc(p->taps_per_filter, sizeof(lo_c32));
  if (!p->filt || !p->dfilt || !dt || !p->hist) { free(dt); lo_pfb_destroy(p); return NULL; }
  /* create_diff_taps: difftaps[i] = -taps[i] + taps[i+1]; last = 0 */
  for (size_t i = 0; i + 1 < ntaps; i++) {
    float tap = 0;
    tap += -1.0f * taps[i];
    tap += 1.0f * taps[i + 1];
    dt[i] = tap;
  }
  dt[ntaps - 1] = 0;
  make_bank(taps, ntaps, p->int_rate, p->taps_per_filter, p->filt);
  make_bank(dt, ntaps, p->int_rate, p->taps_per_filter, p->dfilt);
  free(dt);
  return p;
}

void lo_pfb_destroy(lo_pfb *p) {
  if (!p) return;
  free(p->filt); free(p->dfilt); free(p->hist); free(p);
}
unsigned lo_pfb_taps_per_filter(const lo_pfb *p) { return p->taps_per_filter; }
unsigned lo_pfb_dec_rate(const lo_pfb *p) { return p->dec_rate; }
float lo_pfb_flt_rate(const lo_pfb *p) { return p->flt_rate; }
unsigned lo_pfb_last_filter(const lo_pfb *p) { return p->last_filter; }

/* volk_32fc_32f_dot_prod_32fc_generic: sequential float accumulation */
static lo_c32 dot_f(const lo_c32 *x, const float *t, unsigned n) {
  float re = 0, im = 0;
  for (unsigned k = 0; k < n; k++) { re += x[k].re * t[k]; im += x[k].im * t[k]; }
  lo_c32 r = {re, im};
  return r;
}
static void dot_d(const lo_c32 *x, const float *t, unsigned n, double *re, double *im) {
  double a = 0, b = 0;
  for (unsigned k = 0; k < n; k++) { a += (double)x[k].re * t[k]; b += (double)x[k].im * t[k]; }
  *re = a; *im = b;
}

size_t lo_pfb_work(lo_pfb *p, const lo_c32 *in, size_t n_in, lo_c32 *out, size_t out_cap, int use_double) {
  const unsigned K = p->taps_per_filter;
  /* flat stream: history ++ in */
  lo_c32 *s = (lo_c32 *)malloc(sizeof(lo_c32) * (n_in + K));
  if (!s) return 0;
  memcpy(s, p->hist, sizeof(lo_c32) * (K - 1));
  memcpy(s + (K - 1), in, sizeof(lo_c32) * n_in);
  size_t i_out = 0, i_in = p->skip;
  unsigned j = p->last_filter;
  /* kernel::filter; a j carried over from the previous call is < int_rate, as there */
  while (i_in < n_in && i_out < out_cap) {
    while (j < p->int_rate && i_out < out_cap) {
      if (i_in >= n_in) break;   /* cannot happen here (i_in only moves in the outer loop); keeps reads in range */
      const float *f0 = p->filt + (size_t)j * K, *f1 = p->dfilt + (size_t)j * K;
      if (use_double) {
        double ar, ai, br, bi;
        dot_d(s + i_in, f0, K, &ar, &ai);
        dot_d(s + i_in, f1, K, &br, &bi);
        out[i_out].re = (float)(ar + br * (double)p->acc);
        out[i_out].im = (float)(ai + bi * (double)p->acc);
      } else {
        const lo_c32 o0 = dot_f(s + i_in, f0, K), o1 = dot_f(s + i_in, f1, K);
        out[i_out].re = o0.re + o1.re * p->acc;      /* o0 + o1*d_acc */
        out[i_out].im = o0.im + o1.im * p->acc;
      }
      i_out++;
      p->acc += p->flt_rate;
      j += p->dec_rate + (int)floor(p->acc);
      p->acc = fmodf(p->acc, 1.0f);
    }
    i_in += (size_t)(j / p->int_rate);
    j = j % p->int_rate;
  }
  p->last_filter = j;
  p->skip = i_in > n_in ? i_in - n_in : 0;
  /* history for the next call: the last K-1 samples of the flat stream */
  memmove(p->hist, s + n_in, sizeof(lo_c32) * (K - 1));
  free(s);
  return i_out;
}

size_t lo_pfb_schedule(lo_pfb *p, size_t n_in, uint64_t *pos, float *acc, size_t cap) {
  size_t i_out = 0, i_in = p->skip;
  unsigned j = p->last_filter;
  while (i_in < n_in && i_out < cap) {
    while (j < p->int_rate && i_out < cap) {
      pos[i_out] = (uint64_t)i_in * p->int_rate + j;
      acc[i_out] = p->acc;
      i_out++;
      p->acc += p->flt_rate;
      j += p->dec_rate + (int)floor(p->acc);
      p->acc = fmodf(p->acc, 1.0f);
    }
    i_in += (size_t)(j / p->int_rate);
    j = j % p->int_rate;
  }
  p->last_filter = j;
  p->skip = i_in > n_in ? i_in - n_in : 0;
  return i_out;
}
