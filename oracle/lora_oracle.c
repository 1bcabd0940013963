/*
 * lora_oracle.c -- CPU ORACLE for the gr-lora demod hot path.  TEST INFRASTRUCTURE ONLY.
 * See lora_oracle.h for scope, citations and the "parity unpinned" statement.
 *
 * Build with -ffp-contract=off: the reference's multiplies are plain (round each product, then
 * round the sum) in VOLK's generic/SSE3 kernels, and the oracle's arithmetic must not depend on
 * whether the host compiler fuses a*b+c*d.
 */
#define _POSIX_C_SOURCE 200809L
#include "lora_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* include/lora/demod.h:28-33 */
#define DEMOD_HISTORY_DEPTH       3
#define REQUIRED_PREAMBLE_CHIRPS  4
#define REQUIRED_SFD_CHIRPS       2
#define LORA_SFD_TOLERANCE        1
#define LORA_PREAMBLE_TOLERANCE   1
#define DEMOD_SYNC_RECOVERY_COUNT (8-REQUIRED_PREAMBLE_CHIRPS)+(2-REQUIRED_SFD_CHIRPS)+4
/* lib/demod_impl.cc:35-36 */
#define OVERLAP_DEFAULT 1
#define OVERLAP_FACTOR  16
/* lib/mod_impl.h:30 */
#define NUM_PREAMBLE_CHIRPS 8

typedef struct { double re, im; } lo_c64;

/* ------------------------------------------------------------------------------------------
 * Forward DFT, unnormalised, X[k] = sum x[n] exp(-j 2 pi k n / M)  -- what
 * gr::fft::fft_complex(M, true, 1) / fftwf_plan_dft_1d(FFTW_FORWARD) computes
 * (call sites demod_impl.cc:81,201-203,317-319).  FFTW itself is not available; this is a
 * textbook iterative radix-2 decimation-in-time transform in single precision with twiddles
 * rounded from double, and the same transform in double precision for margin analysis.
 * A size that is not a power of two (fft_factor 3, 5, 6 ...: the reference hands any M to FFTW)
 * goes through a textbook recursive mixed-radix decimation-in-time transform instead: split by
 * the smallest prime factor p, transform the p decimated sequences, combine with a direct
 * p-point DFT.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  unsigned m, logm;
  uint32_t *rev;     /* bit reversal permutation */
  lo_c32 *tw;        /* tw[k] = exp(-j 2 pi k / M), k < M/2, rounded to float */
  lo_c64 *twd;       /* same in double */
  lo_c32 *in, *out;  /* the plan's buffers, like gr::fft's get_inbuf()/get_outbuf() */
  lo_c64 *dwork;
  int mixed;         /* m is not a power of two */
  lo_c32 *mtw;       /* mixed radix: mtw[k] = exp(-j 2 pi k / M), k < M */
  lo_c64 *mtwd, *din;
} lo_fft;

static lo_fft *fft_create(unsigned m)
{
  lo_fft *p = (lo_fft *)calloc(1, sizeof *p);
  p->m = m;
  while ((1u << p->logm) < m) p->logm++;
  p->rev = (uint32_t *)malloc(sizeof(uint32_t) * m);
  for (unsigned i = 0; i < m; i++) {
    unsigned r = 0;
    for (unsigned b = 0; b < p->logm; b++) if (i & (1u << b)) r |= 1u << (p->logm - 1 - b);
    p->rev[i] = r;
  }
  unsigned h = m / 2 ? m / 2 : 1;
  p->tw = (lo_c32 *)malloc(sizeof(lo_c32) * h);
  p->twd = (lo_c64 *)malloc(sizeof(lo_c64) * h);
  for (unsigned k = 0; k < h; k++) {
    double a = -2.0 * M_PI * (double)k / (double)m;
    p->twd[k].re = cos(a); p->twd[k].im = sin(a);
    p->tw[k].re = (float)p->twd[k].re; p->tw[k].im = (float)p->twd[k].im;
  }
  p->in = (lo_c32 *)aligned_alloc(64, sizeof(lo_c32) * m);
  p->out = (lo_c32 *)aligned_alloc(64, sizeof(lo_c32) * m);
  p->dwork = (lo_c64 *)malloc(sizeof(lo_c64) * m);
  p->mixed = (m & (m - 1)) != 0;
  if (p->mixed) {
    p->mtw = (lo_c32 *)malloc(sizeof(lo_c32) * m);
    p->mtwd = (lo_c64 *)malloc(sizeof(lo_c64) * m);
    p->din = (lo_c64 *)malloc(sizeof(lo_c64) * m);
    for (unsigned k = 0; k < m; k++) {
      double a = -2.0 * M_PI * (double)k / (double)m;
      p->mtwd[k].re = cos(a); p->mtwd[k].im = sin(a);
      p->mtw[k].re = (float)p->mtwd[k].re; p->mtw[k].im = (float)p->mtwd[k].im;
    }
  }
  return p;
}

static unsigned smallest_prime_factor(unsigned n)
{
  for (unsigned q = 2; q * q <= n; q++) if (n % q == 0) return q;
  return n;
}

/* out[0..n) = DFT_n of in[0], in[stride], in[2 stride] ...; tw = exp(-j 2 pi k / m) with n | m */
#define LO_MIXED_FFT(NAME, CT, RT)                                                                   \
static void NAME(const CT *in, CT *out, unsigned n, unsigned stride, const CT *tw, unsigned m)      \
{                                                                                                    \
  if (n == 1) { out[0] = in[0]; return; }                                                            \
  const unsigned p = smallest_prime_factor(n), q = n / p;                                            \
  for (unsigned r = 0; r < p; r++) NAME(in + r * stride, out + r * q, q, stride * p, tw, m);         \
  CT y[p];                                                                                           \
  for (unsigned k = 0; k < q; k++) {                                                                 \
    for (unsigned r = 0; r < p; r++) {          /* y_r = W_n^(r k) Y_r[k] */                         \
      const CT w = tw[(size_t)r * k * (m / n) % m], a = out[r * q + k];                              \
      y[r].re = a.re * w.re - a.im * w.im;                                                           \
      y[r].im = a.re * w.im + a.im * w.re;                                                           \
    }                                                                                                \
    for (unsigned t = 0; t < p; t++) {          /* X[k + q t] = sum_r W_p^(r t) y_r */               \
      RT sr = y[0].re, si = y[0].im;                                                                 \
      for (unsigned r = 1; r < p; r++) {                                                             \
        const CT w = tw[(size_t)(r * t % p) * (m / p)];                                              \
        sr += y[r].re * w.re - y[r].im * w.im;                                                       \
        si += y[r].re * w.im + y[r].im * w.re;                                                       \
      }                                                                                              \
      out[k + q * t].re = sr; out[k + q * t].im = si;                                                \
    }                                                                                                \
  }                                                                                                  \
}
LO_MIXED_FFT(mixed_fft_f, lo_c32, float)
LO_MIXED_FFT(mixed_fft_d, lo_c64, double)

static void fft_destroy(lo_fft *p)
{
  if (!p) return;
  free(p->rev); free(p->tw); free(p->twd); free(p->in); free(p->out); free(p->dwork);
  free(p->mtw); free(p->mtwd); free(p->din); free(p);
}

/* in -> out, single precision */
static void fft_execute(lo_fft *p)
{
  const unsigned m = p->m;
  if (p->mixed) { mixed_fft_f(p->in, p->out, m, 1, p->mtw, m); return; }
  lo_c32 *x = p->out;
  for (unsigned i = 0; i < m; i++) x[p->rev[i]] = p->in[i];
  for (unsigned len = 2; len <= m; len <<= 1) {
    const unsigned half = len >> 1, step = m / len;
    for (unsigned base = 0; base < m; base += len) {
      lo_c32 *a = x + base, *b = x + base + half;
      for (unsigned j = 0; j < half; j++) {
        const lo_c32 w = p->tw[j * step];
        const float tr = b[j].re * w.re - b[j].im * w.im;
        const float ti = b[j].re * w.im + b[j].im * w.re;
        const float ur = a[j].re, ui = a[j].im;
        a[j].re = ur + tr; a[j].im = ui + ti;
        b[j].re = ur - tr; b[j].im = ui - ti;
      }
    }
  }
}

/* same transform in double on the plan's float input (promoted), result in p->dwork */
static void fft_execute_double(lo_fft *p)
{
  const unsigned m = p->m;
  if (p->mixed) {
    for (unsigned i = 0; i < m; i++) { p->din[i].re = p->in[i].re; p->din[i].im = p->in[i].im; }
    mixed_fft_d(p->din, p->dwork, m, 1, p->mtwd, m);
    return;
  }
  lo_c64 *x = p->dwork;
  for (unsigned i = 0; i < m; i++) { x[p->rev[i]].re = p->in[i].re; x[p->rev[i]].im = p->in[i].im; }
  for (unsigned len = 2; len <= m; len <<= 1) {
    const unsigned half = len >> 1, step = m / len;
    for (unsigned base = 0; base < m; base += len) {
      lo_c64 *a = x + base, *b = x + base + half;
      for (unsigned j = 0; j < half; j++) {
        const lo_c64 w = p->twd[j * step];
        const double tr = b[j].re * w.re - b[j].im * w.im;
        const double ti = b[j].re * w.im + b[j].im * w.re;
        const double ur = a[j].re, ui = a[j].im;
        a[j].re = ur + tr; a[j].im = ui + ti;
        b[j].re = ur - tr; b[j].im = ui - ti;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * VOLK restatements (generic / SSE3 arithmetic: every product and every sum rounded once).
 * noinline so that the dead down-path multiply of demod_impl.cc:175 really runs in the
 * timing leg, as it does in the reference.
 * ---------------------------------------------------------------------------------------- */
__attribute__((noinline))
static void volk_32fc_x2_multiply_32fc(lo_c32 *c, const lo_c32 *a, const lo_c32 *b, unsigned n)
{
  for (unsigned i = 0; i < n; i++) {
    const float ar = a[i].re, ai = a[i].im, br = b[i].re, bi = b[i].im;
    c[i].re = ar * br - ai * bi;
    c[i].im = ar * bi + ai * br;
  }
}

__attribute__((noinline))
static void volk_32fc_32f_multiply_32fc(lo_c32 *c, const lo_c32 *a, const float *b, unsigned n)
{
  for (unsigned i = 0; i < n; i++) { c[i].re = a[i].re * b[i]; c[i].im = a[i].im * b[i]; }
}

/* ------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------- */
/* demod_impl.cc:92-101 (identical loop at mod_impl.cc:60-69): phase is a FLOAT that is
 * re-rounded after every increment, the accumulator is a double, std::polar(1.0, acc) is
 * evaluated in double and narrowed to complex<float>. */
void lo_chirp_tables(unsigned sf, lo_c32 *up, lo_c32 *down)
{
  const int n = 1 << sf;
  float phase = -M_PI;
  double accumulator = 0;
  for (int i = 0; i < 2 * n; i++) {
    accumulator += phase;
    const double c = cos(accumulator), s = sin(accumulator);   /* std::polar(1.0, acc) */
    down[i].re = (float)c; down[i].im = (float)(-s);           /* conj */
    up[i].re = (float)c;   up[i].im = (float)s;
    phase += (2 * M_PI) / n;
  }
}

/* GNU Radio 3.7.x gr-fft/lib/window.cc (not under /root/reference, version unpinned):
 *   Izero(x): power series of the modified Bessel function I0, terms added until
 *             u < 1e-21 * sum;
 *   kaiser(ntaps, beta): taps[i] = Izero(beta*sqrt(1 - t*t)) / Izero(beta),
 *             t = 2*i/(ntaps-1) - 1, computed in double, stored as float. */
static double Izero(double x)
{
  double sum, u, halfx, temp;
  int n;
  sum = u = n = 1;
  halfx = x / 2.0;
  do {
    temp = halfx / (double)n;
    n += 1;
    temp *= temp;
    u *= temp;
    sum += u;
  } while (u >= 1E-21 * sum);
  return sum;
}

void lo_kaiser(unsigned ntaps, double beta, float *taps)
{
  const double IBeta = 1.0 / Izero(beta);
  const double inm1 = 1.0 / ((double)(ntaps - 1));
  for (unsigned i = 0; i < ntaps; i++) {
    double temp = 2 * (int)i * inm1 - 1;
    taps[i] = (float)(Izero(beta * sqrt(1.0 - temp * temp)) * IBeta);
  }
}

/* ------------------------------------------------------------------------------------------
 * Capture synthesis
 * ---------------------------------------------------------------------------------------- */
size_t lo_mod_frame_len(unsigned sf, size_t n_sym)
{
  const size_t n = (size_t)1 << sf;
  return 4 * n + (NUM_PREAMBLE_CHIRPS + 2) * n + (2 * n + n / 4) + n_sym * n + 4 * n + 128;
}

/* mod_impl.cc:80-139, for a modulate() call on an empty d_iq_out queue */
void lo_modulate(unsigned sf, unsigned sync_word, const uint16_t *syms, size_t n_sym, lo_c32 *out)
{
  const int n = 1 << sf;
  lo_c32 *up = (lo_c32 *)malloc(sizeof(lo_c32) * 2 * n);
  lo_c32 *down = (lo_c32 *)malloc(sizeof(lo_c32) * 2 * n);
  lo_chirp_tables(sf, up, down);
  const lo_c32 zero = {0.0f, 0.0f};                      /* gr_complex(std::polar(0.0, 0.0)) */
  size_t k = 0;
  for (int i = 0; i < 4 * n; i++) out[k++] = zero;                               /* :124 */
  for (int i = 0; i < NUM_PREAMBLE_CHIRPS * n; i++) out[k++] = up[i % n];        /* :91-94 */
  for (int i = 0; i < n; i++) out[k++] = up[(8 * ((sync_word & 0xF0) >> 4) + i) % n]; /* :97-100 */
  for (int i = 0; i < n; i++) out[k++] = up[(8 * (sync_word & 0x0F) + i) % n];   /* :103-106 */
  for (int i = 0; i < 2 * n + n / 4; i++) out[k++] = down[i % n];                /* :109-112 */
  for (size_t i = 0; i < n_sym; i++)                                             /* :115-121 */
    for (int j = 0; j < n; j++) out[k++] = up[(syms[i] + (n / 4) + j) % n];
  for (int i = 0; i < 4 * n + 128; i++) out[k++] = zero;                         /* :133 */
  free(up); free(down);
}

uint64_t lo_splitmix64(uint64_t *state)
{
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void lo_awgn(lo_c32 *iq, size_t n, double sigma2, uint64_t seed, float gain)
{
  uint64_t st = seed;
  const double s = sqrt(sigma2 / 2.0);
  for (size_t i = 0; i < n; i++) {
    /* u1 in (0,1], u2 in [0,1) from the top 53 bits */
    const double u1 = ((double)(lo_splitmix64(&st) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(lo_splitmix64(&st) >> 11) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    const double gr = r * cos(2.0 * M_PI * u2), gi = r * sin(2.0 * M_PI * u2);
    const float re = (float)((double)iq[i].re + s * gr);
    const float im = (float)((double)iq[i].im + s * gi);
    iq[i].re = re * gain;
    iq[i].im = im * gain;
  }
}

/* ------------------------------------------------------------------------------------------
 * demod instance (members of lib/demod_impl.h:41-72)
 * ---------------------------------------------------------------------------------------- */
struct lo_demod {
  int d_state;
  unsigned short d_sf;
  int d_ldr;
  unsigned short d_num_symbols, d_fft_size_factor, d_fft_size, d_overlaps, d_offset;
  float d_power, d_threshold;
  int d_squelched;
  unsigned short d_preamble_idx, d_sfd_idx;
  unsigned short d_argmax_history[REQUIRED_PREAMBLE_CHIRPS + 1]; unsigned n_argmax_history;
  unsigned short d_sfd_history[REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR + 1]; unsigned n_sfd_history;
  unsigned short d_sync_recovery_counter;
  lo_fft *d_fft;
  float *d_window;
  float d_beta;
  lo_c32 *d_upchirp, *d_downchirp;
  unsigned short *d_symbols; size_t n_symbols, cap_symbols;
  /* L1 scratch */
  lo_c32 *scratch;
  /* instrumentation */
  size_t n_ffts;
  unsigned last_n_sfd;
  /* optimised CPU variant (lo_demod_time_opt): NULL = the faithful loop */
  struct lo_fast *fast;
};

static void fast_destroy(struct lo_fast *q);

static void state_reset_members(lo_demod *d)
{
  d->d_state = LO_S_RESET;                                /* :77 */
  d->d_overlaps = OVERLAP_DEFAULT;                        /* :82 */
  d->d_offset = 0;                                        /* :83 */
  d->d_power = .000000001f;                               /* :87 */
  d->d_threshold = 0.005f;                                /* :88 */
  d->d_squelched = 0;  /* uninitialised in the reference; always written by the first argmax */
  d->d_preamble_idx = 0; d->d_sfd_idx = 0;
  d->n_argmax_history = 0; d->n_sfd_history = 0;
  d->d_sync_recovery_counter = 0;
  d->n_symbols = 0;
  d->n_ffts = 0;
}

lo_demod *lo_demod_create(unsigned sf, int ldr, float beta, unsigned fft_factor)
{
  if (sf < 6 || sf > 12 || fft_factor == 0) return NULL;  /* asserts at :70-72 */
  if (((size_t)fft_factor << sf) > 65535u) return NULL; /* d_fft_size is unsigned short: 65536 would wrap to 0 */
  lo_demod *d = (lo_demod *)calloc(1, sizeof *d);
  d->d_sf = (unsigned short)sf; d->d_ldr = ldr; d->d_beta = beta;
  d->d_fft_size_factor = (unsigned short)fft_factor;
  d->d_num_symbols = (unsigned short)(1 << sf);                              /* :79 */
  d->d_fft_size = (unsigned short)(d->d_fft_size_factor * d->d_num_symbols); /* :80 */
  d->d_fft = fft_create(d->d_fft_size);                                      /* :81 */
  d->d_window = (float *)malloc(sizeof(float) * d->d_num_symbols);
  lo_kaiser(d->d_num_symbols, d->d_beta, d->d_window);                       /* :85 */
  d->d_upchirp = (lo_c32 *)malloc(sizeof(lo_c32) * 2 * d->d_num_symbols);
  d->d_downchirp = (lo_c32 *)malloc(sizeof(lo_c32) * 2 * d->d_num_symbols);
  lo_chirp_tables(sf, d->d_upchirp, d->d_downchirp);                         /* :92-101 */
  d->cap_symbols = 1024;
  d->d_symbols = (unsigned short *)malloc(sizeof(unsigned short) * d->cap_symbols);
  d->scratch = (lo_c32 *)malloc(sizeof(lo_c32) * d->d_num_symbols);
  state_reset_members(d);
  return d;
}

void lo_demod_destroy(lo_demod *d)
{
  if (!d) return;
  fft_destroy(d->d_fft);
  fast_destroy(d->fast);
  free(d->d_window); free(d->d_upchirp); free(d->d_downchirp); free(d->d_symbols);
  free(d->scratch); free(d);
}

const lo_c32 *lo_demod_upchirp(const lo_demod *d) { return d->d_upchirp; }
const lo_c32 *lo_demod_downchirp(const lo_demod *d) { return d->d_downchirp; }
const float *lo_demod_window(const lo_demod *d) { return d->d_window; }

/* demod_impl.cc:114-140.  `pow(float, 2)` resolves to a double-valued pow under C++11
 * (std::pow(float,int) promotes to double; ::pow is the C double function), so
 * magsq = (float)((double)re*re + (double)im*im).  The loop starts its comparisons at i = 0
 * against max_val = magsq[0], strict '>', so the lowest index of the maximum wins and a NaN
 * never replaces the running maximum. */
static unsigned short argmax(lo_demod *d, const lo_c32 *fft_result, int update_squelch, float *max_out)
{
  float magsq = (float)((double)fft_result[0].re * (double)fft_result[0].re +
                        (double)fft_result[0].im * (double)fft_result[0].im);
  float max_val = magsq;
  unsigned short max_idx = 0;
  for (unsigned i = 0; i < d->d_fft_size; i++) {
    magsq = (float)((double)fft_result[i].re * (double)fft_result[i].re +
                    (double)fft_result[i].im * (double)fft_result[i].im);
    if (magsq > max_val) { max_idx = (unsigned short)i; max_val = magsq; }
  }
  if (update_squelch) {
    d->d_power = max_val;
    d->d_squelched = (d->d_power > d->d_threshold) ? 0 : 1;
  }
  if (max_out) *max_out = max_val;
  return max_idx;
}

/* zero-pad + FFT of an N-sample block: demod_impl.cc:201-203 / :317-319 */
static void padded_fft(lo_demod *d, const lo_c32 *block)
{
  memset(d->d_fft->in, 0, d->d_fft_size * sizeof(lo_c32));
  memcpy(d->d_fft->in, block, d->d_num_symbols * sizeof(lo_c32));
  fft_execute(d->d_fft);
  d->n_ffts++;
}

static double top2_margin(lo_demod *d)
{
  fft_execute_double(d->d_fft);
  double t1 = -1.0, t2 = -1.0;
  for (unsigned i = 0; i < d->d_fft_size; i++) {
    const lo_c64 x = d->d_fft->dwork[i];
    const double m = x.re * x.re + x.im * x.im;
    if (m > t1) { t2 = t1; t1 = m; } else if (m > t2) { t2 = m; }
  }
  if (!(t1 > 0.0)) return 1.0;
  if (t2 < 0.0) t2 = 0.0;
  return (t1 - t2) / t1;
}

unsigned lo_window_argmax(lo_demod *d, const lo_c32 *in, unsigned chirp_off, int kind,
                          float *val, double *margin)
{
  const unsigned n = d->d_num_symbols;
  lo_c32 *blk = d->scratch;
  if (kind == LO_KIND_UP) {
    volk_32fc_x2_multiply_32fc(blk, in, d->d_downchirp + chirp_off, n);   /* :179/:183 */
    volk_32fc_32f_multiply_32fc(blk, blk, d->d_window, n);                /* :192 */
  } else {
    volk_32fc_x2_multiply_32fc(blk, in, d->d_upchirp + chirp_off, n);     /* :310 */
  }
  const float save_power = d->d_power; const int save_sq = d->d_squelched;
  padded_fft(d, blk);
  float mv;
  unsigned idx = argmax(d, d->d_fft->out, 0, &mv);
  d->d_power = save_power; d->d_squelched = save_sq;
  if (val) *val = mv;
  if (margin) *margin = top2_margin(d);
  return idx;
}

void lo_window_argmax_batch(lo_demod *d, const lo_c32 *iq, size_t n, const uint64_t *pos,
                            const uint16_t *chirp_off, const uint8_t *kind, size_t n_win,
                            uint16_t *idx, float *val, double *margin)
{
  for (size_t i = 0; i < n_win; i++) {
    if (pos[i] + d->d_num_symbols > n) { idx[i] = 0xFFFF; val[i] = -1.0f; if (margin) margin[i] = 0; continue; }
    float v; double m;
    idx[i] = (uint16_t)lo_window_argmax(d, iq + pos[i], chirp_off[i], kind[i], &v, margin ? &m : NULL);
    val[i] = v;
    if (margin) margin[i] = m;
  }
}

static void hist_push_front(unsigned short *h, unsigned *n, unsigned short v)
{
  memmove(h + 1, h, sizeof(unsigned short) * (*n));
  h[0] = v;
  (*n)++;
}

static void symbols_push(lo_demod *d, unsigned short s)
{
  if (d->n_symbols == d->cap_symbols) {
    d->cap_symbols *= 2;
    d->d_symbols = (unsigned short *)realloc(d->d_symbols, sizeof(unsigned short) * d->cap_symbols);
  }
  d->d_symbols[d->n_symbols++] = s;
}

/* ---- optimised CPU variant of one window evaluation (cpu_baseline "optimised") -----------------------
 * What a tuned CPU build of the reference's loop would do without changing what is computed: no per-call allocation,
 * no dead multiply (:175), no memset/memcpy of the padded buffer; the zero-padded M = N*F point DFT as F pruned N-point
 * transforms of x[n]*W_M^(r n) (bins F*k + r), each a radix-4 Stockham autosort FFT on split re/im arrays (contiguous inner
 * loops that gcc vectorises) with per-stage twiddle tables; |X|^2 and the lowest-bin argmax fused into the last pass's
 * output scan.  Same dechirp/window order as the faithful loop.  Results agree with it to rounding; it is a timing
 * baseline, not the parity oracle. */
typedef struct lo_fast {
  unsigned n, f, n1, n2;             /* N = n1 * n2 (n1 <= n2, powers of two) */
  float *ar, *ai, *br, *bi;          /* ping-pong work arrays, n each */
  float *xr, *xi;                    /* dechirped (and windowed) input */
  float *w1r, *w1i, *w2r, *w2i;      /* stage twiddles of the length-n1 and length-n2 column transforms */
  float *mr, *mi;                    /* middle twiddles W_N^(k1*n2), n entries laid out [k1][n2] */
  float *tr, *ti;                    /* input twist W_M^(r n), f*n entries */
} lo_fast;

static int fast_stage_table(unsigned len, float **wr, float **wi)
{
  size_t tw = 0;
  for (unsigned l = len; l >= 4; l /= 4) tw += 3 * (l / 4);
  if (posix_memalign((void **)wr, 64, sizeof(float) * (tw + 16)) || posix_memalign((void **)wi, 64, sizeof(float) * (tw + 16))) return -1;
  size_t o = 0;
  for (unsigned l = len; l >= 4; l /= 4)
    for (unsigned k = 1; k <= 3; k++)
      for (unsigned p = 0; p < l / 4; p++) {
        const double a = -2.0 * M_PI * (double)(k * p) / (double)l;
        (*wr)[o] = (float)cos(a); (*wi)[o] = (float)sin(a); o++;
      }
  return 0;
}

static lo_fast *fast_create(unsigned n, unsigned f)
{
  lo_fast *q = (lo_fast *)calloc(1, sizeof *q);
  q->n = n; q->f = f;
  unsigned logn = 0;
  while ((1u << logn) < n) logn++;
  q->n1 = 1u << (logn / 2); q->n2 = n / q->n1;
  float **arr[] = { &q->ar, &q->ai, &q->br, &q->bi, &q->xr, &q->xi, &q->mr, &q->mi };
  for (unsigned i = 0; i < 8; i++) if (posix_memalign((void **)arr[i], 64, sizeof(float) * n)) return NULL;
  if (fast_stage_table(q->n1, &q->w1r, &q->w1i) || fast_stage_table(q->n2, &q->w2r, &q->w2i)) return NULL;
  for (unsigned k1 = 0; k1 < q->n1; k1++)
    for (unsigned c = 0; c < q->n2; c++) {
      const double a = -2.0 * M_PI * (double)(k1 * c) / (double)n;
      q->mr[k1 * q->n2 + c] = (float)cos(a); q->mi[k1 * q->n2 + c] = (float)sin(a);
    }
  if (posix_memalign((void **)&q->tr, 64, sizeof(float) * n * f) || posix_memalign((void **)&q->ti, 64, sizeof(float) * n * f)) return NULL;
  for (unsigned r = 0; r < f; r++)
    for (unsigned i = 0; i < n; i++) {
      const double a = -2.0 * M_PI * (double)((unsigned long long)r * i % ((unsigned long long)n * f)) / (double)((unsigned long long)n * f);
      q->tr[r * n + i] = (float)cos(a); q->ti[r * n + i] = (float)sin(a);
    }
  return q;
}

static void fast_destroy(struct lo_fast *q)
{
  if (!q) return;
  free(q->ar); free(q->ai); free(q->br); free(q->bi); free(q->xr); free(q->xi); free(q->mr); free(q->mi);
  free(q->w1r); free(q->w1i); free(q->w2r); free(q->w2i); free(q->tr); free(q->ti); free(q);
}

/* `len`-point forward Stockham transforms (decimation in frequency, radix 4 with a closing radix-2 pass) down the columns of a
 * [len][v] array: element = a whole row of v contiguous values, so every inner loop is contiguous.  Input in (*xr, *xi), the
 * other buffer in (*yr, *yi); on return (*xr, *xi) point at the result. */
static void fast_col_fft(unsigned len, unsigned v, const float *wr, const float *wi, float **pxr, float **pxi, float **pyr, float **pyi)
{
  float *xr = *pxr, *xi = *pxi, *yr = *pyr, *yi = *pyi;
  unsigned s = v;
  size_t o = 0;
  while (len >= 4) {
    const unsigned n1 = len / 4;
    const float *w1r = wr + o, *w1i = wi + o, *w2r = w1r + n1, *w2i = w1i + n1, *w3r = w2r + n1, *w3i = w2i + n1;
    for (unsigned p = 0; p < n1; p++) {
      const float c1 = w1r[p], s1 = w1i[p], c2 = w2r[p], s2 = w2i[p], c3 = w3r[p], s3 = w3i[p];
      const float *restrict a_r = xr + s * p, *restrict a_i = xi + s * p, *restrict b_r = a_r + s * n1, *restrict b_i = a_i + s * n1;
      const float *restrict c_r = b_r + s * n1, *restrict c_i = b_i + s * n1, *restrict d_r = c_r + s * n1, *restrict d_i = c_i + s * n1;
      float *restrict o0r = yr + s * 4 * p, *restrict o0i = yi + s * 4 * p, *restrict o1r = o0r + s, *restrict o1i = o0i + s,
            *restrict o2r = o1r + s, *restrict o2i = o1i + s, *restrict o3r = o2r + s, *restrict o3i = o2i + s;
#pragma GCC ivdep
      for (unsigned k = 0; k < s; k++) {
        const float apc_r = a_r[k] + c_r[k], apc_i = a_i[k] + c_i[k], amc_r = a_r[k] - c_r[k], amc_i = a_i[k] - c_i[k];
        const float bpd_r = b_r[k] + d_r[k], bpd_i = b_i[k] + d_i[k];
        const float jb_r = -(b_i[k] - d_i[k]), jb_i = b_r[k] - d_r[k];           /* j*(b - d) */
        o0r[k] = apc_r + bpd_r; o0i[k] = apc_i + bpd_i;
        const float t1r = amc_r - jb_r, t1i = amc_i - jb_i;
        o1r[k] = t1r * c1 - t1i * s1; o1i[k] = t1r * s1 + t1i * c1;
        const float t2r = apc_r - bpd_r, t2i = apc_i - bpd_i;
        o2r[k] = t2r * c2 - t2i * s2; o2i[k] = t2r * s2 + t2i * c2;
        const float t3r = amc_r + jb_r, t3i = amc_i + jb_i;
        o3r[k] = t3r * c3 - t3i * s3; o3i[k] = t3r * s3 + t3i * c3;
      }
    }
    o += 3 * n1;
    { float *t = xr; xr = yr; yr = t; t = xi; xi = yi; yi = t; }
    len /= 4; s *= 4;
  }
  if (len == 2) {                              /* odd log2: one radix-2 pass closes the transform */
#pragma GCC ivdep
    for (unsigned k = 0; k < s; k++) {
      yr[k] = xr[k] + xr[k + s]; yi[k] = xi[k] + xi[k + s];
      yr[k + s] = xr[k] - xr[k + s]; yi[k + s] = xi[k] - xi[k + s];
    }
    { float *t = xr; xr = yr; yr = t; t = xi; xi = yi; yi = t; }
  }
  *pxr = xr; *pxi = xi; *pyr = yr; *pyi = yi;
}

/* forward N-point FFT of (ar, ai) by the four-step method: N = n1*n2, sample n = n2*r + c seen as row r, column c.
 * n1-point transforms down the columns, times W_N^(k1*c), transpose, n2-point transforms down the columns again: the
 * result X[k1 + n1*k2] lies at [k2][k1], i.e. in natural order.  Returns pointers to the result. */
static void fast_fft(lo_fast *q, float **out_r, float **out_i)
{
  float *xr = q->ar, *xi = q->ai, *yr = q->br, *yi = q->bi;
  const unsigned n1 = q->n1, n2 = q->n2;
  fast_col_fft(n1, n2, q->w1r, q->w1i, &xr, &xi, &yr, &yi);                   /* x: [k1][c] */
  for (unsigned i = 0; i < q->n; i++) {                                       /* middle twiddles */
    const float re = xr[i] * q->mr[i] - xi[i] * q->mi[i], im = xr[i] * q->mi[i] + xi[i] * q->mr[i];
    xr[i] = re; xi[i] = im;
  }
  for (unsigned r0 = 0; r0 < n1; r0 += 8)                                      /* transpose into y: [c][k1], 8x8 blocks */
    for (unsigned c0 = 0; c0 < n2; c0 += 8)
      for (unsigned r = r0; r < r0 + 8 && r < n1; r++)
        for (unsigned c = c0; c < c0 + 8 && c < n2; c++) { yr[c * n1 + r] = xr[r * n2 + c]; yi[c * n1 + r] = xi[r * n2 + c]; }
  fast_col_fft(n2, n1, q->w2r, q->w2i, &yr, &yi, &xr, &xi);                   /* y: [k2][k1] */
  *out_r = yr; *out_i = yi;
}

/* one window: in[0..N) x chirp[0..N) (x window), zero-padded M-point spectrum, lowest-bin argmax of |X|^2 */
static unsigned short fast_eval(lo_demod *d, const lo_c32 *in, const lo_c32 *chirp, int windowed, float *max_out)
{
  lo_fast *q = d->fast;
  const unsigned n = q->n, f = q->f;
  for (unsigned i = 0; i < n; i++) {          /* VOLK generic order: (x*c) then *w */
    float re = in[i].re * chirp[i].re - in[i].im * chirp[i].im;
    float im = in[i].re * chirp[i].im + in[i].im * chirp[i].re;
    if (windowed) { re *= d->d_window[i]; im *= d->d_window[i]; }
    q->xr[i] = re; q->xi[i] = im;
  }
  float best = -1.0f; unsigned best_idx = 0;
  for (unsigned r = 0; r < f; r++) {
    if (r == 0) { memcpy(q->ar, q->xr, sizeof(float) * n); memcpy(q->ai, q->xi, sizeof(float) * n); }
    else {
      const float *tr = q->tr + (size_t)r * n, *ti = q->ti + (size_t)r * n;
      for (unsigned i = 0; i < n; i++) { q->ar[i] = q->xr[i] * tr[i] - q->xi[i] * ti[i]; q->ai[i] = q->xr[i] * ti[i] + q->xi[i] * tr[i]; }
    }
    float *yr, *yi;
    fast_fft(q, &yr, &yi);
    float *restrict pw = yr == q->ar ? q->br : q->ar;                          /* the work array the result is not in */
    float m = -1.0f; unsigned mi = 0;
#pragma GCC ivdep
    for (unsigned k = 0; k < n; k++)            /* |X|^2 through double as the reference's argmax does */
      pw[k] = (float)((double)yr[k] * (double)yr[k] + (double)yi[k] * (double)yi[k]);
    for (unsigned k = 0; k < n; k++) m = pw[k] > m ? pw[k] : m;
    while (mi < n && pw[mi] != m) mi++;
    if (mi == n) mi = 0;                        /* all NaN: the reference's scan keeps index 0 */
    const unsigned bin = f * mi + r;
    if (m > best || (m == best && bin < best_idx)) { best = m; best_idx = bin; }
  }
  d->n_ffts++;
  if (max_out) *max_out = best;
  return (unsigned short)best_idx;
}

/* PDU sink used by general_work's S_OUT case */
typedef struct {
  uint16_t *sym_out; size_t max_syms, n_syms;
  uint32_t *pdu_len; size_t max_pdus, n_pdus;
} lo_sink;

/* demod_impl.cc:149-450, one call.  `in` points at the read index of the GR-framed stream.
 * Returns num_consumed.  tr (optional) receives the trace row (pos filled by the caller). */
static unsigned short general_work(lo_demod *d, const lo_c32 *in, lo_sink *sink, lo_trace *tr)
{
  unsigned short num_consumed = d->d_num_symbols;
  unsigned short max_index = 0;
  int preamble_found = 0, sfd_found = 0;
  const size_t bytes = d->d_fft_size * sizeof(lo_c32);
  /* :165-167 three per-call allocations (volk_malloc -> posix_memalign) */
  lo_c32 *buffer = NULL, *up_block = NULL, *down_block = NULL;
  const int fast = d->fast != NULL;             /* optimised CPU variant: no allocations, no dead multiply, pruned transform */
  if (!fast) {
    if (posix_memalign((void **)&buffer, 64, bytes)) buffer = NULL;
    if (posix_memalign((void **)&up_block, 64, bytes)) up_block = NULL;
    if (posix_memalign((void **)&down_block, 64, bytes)) down_block = NULL;
  }

  const int st_in = d->d_state;
  unsigned short off_used = 0;
  d->last_n_sfd = 0;

  float up_val;
  if (fast) {
    if (d->d_state == LO_S_READ_HEADER || d->d_state == LO_S_READ_PAYLOAD) off_used = d->d_offset;
    max_index = fast_eval(d, in, &d->d_downchirp[off_used], 1, &up_val);
    d->d_power = up_val;                                                      /* argmax(..., update_squelch = true), :133-137 */
    d->d_squelched = (d->d_power > d->d_threshold) ? 0 : 1;
  } else {
  /* :175 dead multiply (result only used by DUMP_IQ / overwritten in the SFD loop) */
  volk_32fc_x2_multiply_32fc(down_block, in, &d->d_upchirp[0], d->d_num_symbols);

  if (d->d_state == LO_S_READ_HEADER || d->d_state == LO_S_READ_PAYLOAD) {   /* :177-184 */
    off_used = d->d_offset;
    volk_32fc_x2_multiply_32fc(up_block, in, &d->d_downchirp[d->d_offset], d->d_num_symbols);
  } else {
    volk_32fc_x2_multiply_32fc(up_block, in, &d->d_downchirp[0], d->d_num_symbols);
  }
  volk_32fc_32f_multiply_32fc(up_block, up_block, &d->d_window[0], d->d_num_symbols); /* :192 */

  padded_fft(d, up_block);                                                    /* :201-203 */
  max_index = argmax(d, d->d_fft->out, 1, &up_val);                           /* :206 */
  }
  const unsigned short up_idx = max_index;
  hist_push_front(d->d_argmax_history, &d->n_argmax_history, max_index);      /* :207 */
  if (d->n_argmax_history > REQUIRED_PREAMBLE_CHIRPS) d->n_argmax_history--;  /* :209-212 */

  switch (d->d_state) {
  case LO_S_RESET:                                                            /* :215-229 */
    d->d_overlaps = OVERLAP_DEFAULT;
    d->d_offset = 0;
    d->n_symbols = 0;
    d->n_argmax_history = 0;
    d->n_sfd_history = 0;
    d->d_sync_recovery_counter = 0;
    d->d_state = LO_S_PREFILL;
    break;

  case LO_S_PREFILL:                                                          /* :233-242 */
    if (d->n_argmax_history >= REQUIRED_PREAMBLE_CHIRPS) d->d_state = LO_S_DETECT_PREAMBLE;
    break;

  case LO_S_DETECT_PREAMBLE:                                                  /* :247-273 */
    d->d_preamble_idx = d->d_argmax_history[0];
    preamble_found = 1;
    for (int i = 1; i < REQUIRED_PREAMBLE_CHIRPS; i++)
      if (abs((int)d->d_preamble_idx - (int)d->d_argmax_history[i]) > LORA_PREAMBLE_TOLERANCE)
        preamble_found = 0;
    if (preamble_found && !d->d_squelched) d->d_state = LO_S_SFD_SYNC;
    break;

  case LO_S_SFD_SYNC:                                                         /* :279-359 */
    d->d_overlaps = OVERLAP_FACTOR;
    if (d->d_sync_recovery_counter++ > DEMOD_SYNC_RECOVERY_COUNT) {
      d->d_state = LO_S_RESET;
      d->d_overlaps = OVERLAP_DEFAULT;
    }
    for (int ol = 0; ol < d->d_overlaps; ol++) {
      d->d_offset = (unsigned short)(((ol * d->d_num_symbols) / d->d_overlaps) % d->d_num_symbols);
      if (fast) {
        max_index = fast_eval(d, in + d->d_offset, &d->d_upchirp[d->d_offset], 0, NULL);
      } else {
        for (int j = 0; j < d->d_num_symbols; j++) buffer[j] = in[d->d_offset + j];
        volk_32fc_x2_multiply_32fc(down_block, buffer, &d->d_upchirp[d->d_offset], d->d_num_symbols);
        padded_fft(d, down_block);
        max_index = argmax(d, d->d_fft->out, 0, NULL);
      }
      d->last_n_sfd++;
      hist_push_front(d->d_sfd_history, &d->n_sfd_history, max_index);
      if (d->n_sfd_history > REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR) {
        d->n_sfd_history--;
        sfd_found = 1;
        d->d_sfd_idx = d->d_sfd_history[0];
        for (int i = 1; i < REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR; i++)
          if (abs((int)d->d_sfd_idx - (int)d->d_sfd_history[i]) > d->d_fft_size_factor * LORA_SFD_TOLERANCE)
            sfd_found = 0;
        if (sfd_found) {
          num_consumed = (unsigned short)((ol * d->d_num_symbols) / d->d_overlaps + 5 * d->d_num_symbols / 4);
          d->d_offset = (unsigned short)((d->d_offset + (d->d_num_symbols / 4)) % d->d_num_symbols);
          d->d_state = LO_S_READ_HEADER;
          d->d_overlaps = OVERLAP_DEFAULT;
          break;
        }
      }
    }
    break;

  case LO_S_READ_HEADER:                                                      /* :363-387 */
    if (d->d_squelched) d->d_state = LO_S_OUT;
    else if (d->n_symbols == 7) d->d_state = LO_S_READ_PAYLOAD;
    symbols_push(d, (unsigned short)(((d->d_num_symbols + (d->d_argmax_history[0] / d->d_fft_size_factor)
                     - (d->d_preamble_idx / d->d_fft_size_factor)) % d->d_num_symbols) / 4));
    break;

  case LO_S_READ_PAYLOAD:                                                     /* :391-413 */
    if (d->d_squelched) d->d_state = LO_S_OUT;
    if (d->d_ldr)
      symbols_push(d, (unsigned short)(((d->d_num_symbols + (d->d_argmax_history[0] / d->d_fft_size_factor)
                       - (d->d_preamble_idx / d->d_fft_size_factor)) % d->d_num_symbols) / 4));
    else
      symbols_push(d, (unsigned short)((d->d_num_symbols + (d->d_argmax_history[0] / d->d_fft_size_factor)
                       - (d->d_preamble_idx / d->d_fft_size_factor)) % d->d_num_symbols));
    break;

  case LO_S_OUT:                                                              /* :418-431 */
    if (sink) {
      if (sink->n_pdus < sink->max_pdus) sink->pdu_len[sink->n_pdus] = (uint32_t)d->n_symbols;
      sink->n_pdus++;
      for (size_t i = 0; i < d->n_symbols && sink->n_syms < sink->max_syms; i++)
        sink->sym_out[sink->n_syms++] = d->d_symbols[i];
    }
    d->d_state = LO_S_RESET;
    break;

  default:
    break;
  }

  if (tr) {
    tr->state = (uint16_t)st_in; tr->chirp_off = off_used; tr->idx = up_idx;
    tr->consumed = num_consumed; tr->val = up_val; tr->n_sfd = d->last_n_sfd;
  }

  if (!fast) { free(down_block); free(up_block); free(buffer); }              /* :445-447 */
  return num_consumed;                                                        /* :443 */
}

static size_t run_stream(lo_demod *d, const lo_c32 *capture, size_t n, lo_sink *sink,
                         lo_trace *trace, size_t max_trace, size_t *n_calls)
{
  const size_t N = d->d_num_symbols;
  const size_t hist = DEMOD_HISTORY_DEPTH * N - 1;                            /* :103 */
  const size_t total = hist + n;
  lo_c32 *s = (lo_c32 *)calloc(total ? total : 1, sizeof(lo_c32));
  if (n) memcpy(s + hist, capture, n * sizeof(lo_c32));
  size_t pos = 0, calls = 0;
  const size_t need = N + hist;                         /* forecast(1) + history()-1 */
  while (pos + need <= total) {
    lo_trace row; memset(&row, 0, sizeof row);
    row.pos = pos;
    unsigned short c = general_work(d, s + pos, sink, &row);
    if (trace && calls < max_trace) trace[calls] = row;
    calls++;
    pos += c;
  }
  free(s);
  if (n_calls) *n_calls = calls;
  return sink ? sink->n_pdus : 0;
}

size_t lo_demod_stream(lo_demod *d, const lo_c32 *capture, size_t n,
                       uint16_t *sym_out, size_t max_syms, uint32_t *pdu_len, size_t max_pdus,
                       lo_trace *trace, size_t max_trace, size_t *n_calls, size_t *n_ffts)
{
  lo_sink sink = { sym_out, max_syms, 0, pdu_len, max_pdus, 0 };
  state_reset_members(d);
  size_t np = run_stream(d, capture, n, &sink, trace, max_trace, n_calls);
  if (n_ffts) *n_ffts = d->n_ffts;
  return np;
}

/* The same loop over a stream that is framed already (test stand-in for the product's lora_demod_core_seek): framed[0] is
 * the first sample the block reads, in S_RESET; trace positions are relative to framed[0]. */
size_t lo_demod_stream_framed(lo_demod *d, const lo_c32 *framed, size_t n,
                              uint16_t *sym_out, size_t max_syms, uint32_t *pdu_len, size_t max_pdus,
                              lo_trace *trace, size_t max_trace, size_t *n_calls)
{
  lo_sink sink = { sym_out, max_syms, 0, pdu_len, max_pdus, 0 };
  state_reset_members(d);
  const size_t N = d->d_num_symbols;
  const size_t need = N + DEMOD_HISTORY_DEPTH * N - 1;
  size_t pos = 0, calls = 0;
  while (pos + need <= n) {
    lo_trace row; memset(&row, 0, sizeof row);
    row.pos = pos;
    unsigned short c = general_work(d, framed + pos, &sink, &row);
    if (trace && calls < max_trace) trace[calls] = row;
    calls++;
    pos += c;
  }
  if (n_calls) *n_calls = calls;
  return sink.n_pdus;
}

static double demod_time(int optimised, unsigned sf, int ldr, float beta, unsigned fft_factor,
                         const lo_c32 *capture, size_t n, unsigned reps,
                         size_t *n_calls, size_t *n_ffts, size_t *n_pdus)
{
  lo_demod *d = lo_demod_create(sf, ldr, beta, fft_factor);
  if (!d) return -1.0;
  if (optimised) {
    d->fast = fast_create(d->d_num_symbols, d->d_fft_size_factor);
    if (!d->fast) { lo_demod_destroy(d); return -1.0; }
  }
  const size_t max_syms = 1 << 20, max_pdus = 1 << 16;
  uint16_t *syms = (uint16_t *)malloc(sizeof(uint16_t) * max_syms);
  uint32_t *plen = (uint32_t *)malloc(sizeof(uint32_t) * max_pdus);
  size_t calls = 0, ffts = 0, pdus = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned r = 0; r < reps; r++) {
    lo_sink sink = { syms, max_syms, 0, plen, max_pdus, 0 };
    state_reset_members(d);
    pdus = run_stream(d, capture, n, &sink, NULL, 0, &calls);
    ffts = d->n_ffts;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (n_calls) *n_calls = calls;
  if (n_ffts) *n_ffts = ffts;
  if (n_pdus) *n_pdus = pdus;
  free(syms); free(plen);
  lo_demod_destroy(d);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

double lo_demod_time(unsigned sf, int ldr, float beta, unsigned fft_factor, const lo_c32 *capture, size_t n, unsigned reps,
                     size_t *n_calls, size_t *n_ffts, size_t *n_pdus)
{
  return demod_time(0, sf, ldr, beta, fft_factor, capture, n, reps, n_calls, n_ffts, n_pdus);
}

/* the same loop with the optimised window evaluation (fast_eval); power-of-two N*F only */
double lo_demod_time_opt(unsigned sf, int ldr, float beta, unsigned fft_factor, const lo_c32 *capture, size_t n, unsigned reps,
                         size_t *n_calls, size_t *n_ffts, size_t *n_pdus)
{
  return demod_time(1, sf, ldr, beta, fft_factor, capture, n, reps, n_calls, n_ffts, n_pdus);
}
