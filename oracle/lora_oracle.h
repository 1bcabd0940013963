/*
 * lora_oracle.h -- CPU ORACLE for the gr-lora demod hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm in the reference's
 *   lib/demod_impl.cc  (demod block: tables :92-101, argmax :114-140, general_work :149-450)
 *   lib/mod_impl.cc    (mod block, used only to synthesise captures: tables :60-69, frame :80-139)
 *   include/lora/demod.h:28-46 (tunables and the state enum)
 * plus the published algorithms of the third-party routines the reference calls and that are
 * NOT under /root/reference (GNU Radio 3.7.x gr-fft window::kaiser, FFTW3f forward DFT,
 * VOLK complex multiplies; no version is pinned by the reference's CMake files).
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or fixtures for
 * this path (python/qa_demod.py:34-37 and lib/qa_lora.cc:30-36 are empty) and it cannot be built
 * in this image (GNU Radio, VOLK, FFTW, Boost are absent).  The oracle is pinned only against an
 * independent numpy/scipy double-precision recomputation (oracle/numpy_check.py) and the golden
 * vectors it generated itself (tests/golden/, made by tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (gr-lora_amd/, include/) never links, imports or calls it.
 */
#ifndef LORA_ORACLE_H
#define LORA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interleaved complex64, same memory layout as gr_complex / std::complex<float> */
typedef struct { float re, im; } lo_c32;

/* include/lora/demod.h:38-46 */
enum lo_state {
  LO_S_RESET = 0, LO_S_PREFILL, LO_S_DETECT_PREAMBLE, LO_S_SFD_SYNC,
  LO_S_READ_HEADER, LO_S_READ_PAYLOAD, LO_S_OUT
};

/* window kinds at the L1 (per-window) level */
enum { LO_KIND_UP = 0,   /* in * downchirp[off..] * kaiser  (demod_impl.cc:177-192) */
       LO_KIND_SFD = 1   /* in * upchirp[off..], no window  (demod_impl.cc:310)     */ };

typedef struct lo_demod lo_demod;

/* ---- tables ---------------------------------------------------------------------------- */
/* demod_impl.cc:92-101 == mod_impl.cc:60-69.  up/down hold 2*N entries each, N = 1<<sf. */
void lo_chirp_tables(unsigned sf, lo_c32 *up_2n, lo_c32 *down_2n);
/* gr::fft::window::build(WIN_KAISER, ntaps, beta), call site demod_impl.cc:85 */
void lo_kaiser(unsigned ntaps, double beta, float *taps);

/* ---- capture synthesis (mod block + the harness's own AWGN) ---------------------------- */
/* samples produced by one modulate() call on an empty queue: 4N + (12.25+n_sym)N + 4N+128 */
size_t lo_mod_frame_len(unsigned sf, size_t n_sym);
/* mod_impl.cc:80-139; out must hold lo_mod_frame_len() samples */
void lo_modulate(unsigned sf, unsigned sync_word, const uint16_t *syms, size_t n_sym, lo_c32 *out);
/* seeded complex AWGN, variance sigma2 per complex sample (sigma2/2 per rail), then scale by
 * gain.  splitmix64 + Box-Muller; harness-defined, not part of the reference. */
void lo_awgn(lo_c32 *iq, size_t n, double sigma2, uint64_t seed, float gain);
/* splitmix64 stream exposed so that tests can draw the same symbols from C and numpy */
uint64_t lo_splitmix64(uint64_t *state);

/* ---- demod ------------------------------------------------------------------------------ */
/* demod_impl.cc:41-104 */
lo_demod *lo_demod_create(unsigned sf, int low_data_rate, float beta, unsigned fft_factor);
void lo_demod_destroy(lo_demod *d);
/* read-only views of the tables the instance built (lengths 2N, 2N, N) */
const lo_c32 *lo_demod_upchirp(const lo_demod *d);
const lo_c32 *lo_demod_downchirp(const lo_demod *d);
const float *lo_demod_window(const lo_demod *d);

/* L1: one window.  in points at N samples.  Returns the reference's argmax index
 * (first index of the strict maximum of (float)((double)re*re + (double)im*im)), *val = that
 * maximum.  If margin != NULL also computes the spectrum in double precision and stores the
 * relative gap between its two largest |X|^2: (top1-top2)/top1 (1.0 when top1 == 0). */
unsigned lo_window_argmax(lo_demod *d, const lo_c32 *in, unsigned chirp_off, int kind,
                          float *val, double *margin);
/* L1 over many windows of one buffer: window i starts at pos[i], needs pos[i]+N <= n. */
void lo_window_argmax_batch(lo_demod *d, const lo_c32 *iq, size_t n, const uint64_t *pos,
                            const uint16_t *chirp_off, const uint8_t *kind, size_t n_win,
                            uint16_t *idx, float *val, double *margin);

/* One row of the per-call trace of general_work */
typedef struct {
  uint64_t pos;        /* read index into the GR-framed stream (3N-1 zeros ++ capture) */
  uint16_t state;      /* d_state on entry */
  uint16_t chirp_off;  /* downchirp offset used for the up path of this call */
  uint16_t idx;        /* argmax of the up path */
  uint16_t consumed;   /* num_consumed */
  float    val;        /* max |X|^2 of the up path (d_power) */
  uint32_t n_sfd;      /* SFD windows evaluated in this call (0, 1..16) */
} lo_trace;

/* L2: the whole block on one finite capture, GNU Radio framing restated:
 *  - the stream seen by general_work is (history-1 = 3N-1 zeros) ++ capture (demod_impl.cc:103)
 *  - general_work runs while at least N + 3N-1 samples are available from the read index
 *    (forecast :142-147 with noutput_items = 1, plus history-1)
 *  - every call evaluates exactly one symbol window and consumes num_consumed (:443).
 * PDUs are appended to sym_out (concatenated) with their lengths in pdu_len.
 * trace may be NULL.  Returns the number of PDUs (may exceed max_pdus: then only the first
 * max_pdus lengths are stored; symbols are dropped once sym_out is full).
 * *n_calls / *n_ffts receive the number of general_work calls and of FFTs executed. */
size_t lo_demod_stream(lo_demod *d, const lo_c32 *capture, size_t n,
                       uint16_t *sym_out, size_t max_syms, uint32_t *pdu_len, size_t max_pdus,
                       lo_trace *trace, size_t max_trace, size_t *n_calls, size_t *n_ffts);

/* the same loop over an already framed stream (no history zeros are put in front): framed[0] is read first, in S_RESET */
size_t lo_demod_stream_framed(lo_demod *d, const lo_c32 *framed, size_t n,
                              uint16_t *sym_out, size_t max_syms, uint32_t *pdu_len, size_t max_pdus,
                              lo_trace *trace, size_t max_trace, size_t *n_calls);
/* CPU baseline leg: same loop as lo_demod_stream (faithful: 3 allocations per call, the dead
 * down-path multiply, memset+memcpy zero padding, full M-point FFT, scalar argmax), repeated
 * over the capture `reps` times on a fresh instance state; returns seconds of wall time and
 * stores calls / FFTs per repetition. */
double lo_demod_time(unsigned sf, int ldr, float beta, unsigned fft_factor,
                     const lo_c32 *capture, size_t n, unsigned reps,
                     size_t *n_calls, size_t *n_ffts, size_t *n_pdus);
/* the same loop with an optimised window evaluation (no per-call allocation, no dead multiply, pruned radix-4 Stockham
 * transform, fused argmax): the "optimised CPU" baseline of SURVEY 8d; results agree with the faithful loop to rounding */
double lo_demod_time_opt(unsigned sf, int ldr, float beta, unsigned fft_factor,
                         const lo_c32 *capture, size_t n, unsigned reps,
                         size_t *n_calls, size_t *n_ffts, size_t *n_pdus);

#ifdef __cplusplus
}
#endif
#endif
