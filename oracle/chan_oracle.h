/*
 * chan_oracle.h -- CPU ORACLE for the channeliser in front of the demod block.
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and benchmark CPU legs may use it.
 *
 * The reference's receive flowgraph puts two stock GNU Radio blocks before lora.demod:
 *   examples/rx_usrp.grc:489-503  blocks_rotator_cc, phase_inc = 2*pi*offset/samp_rate
 *   examples/rx_usrp.grc:825-860  pfb_arb_resampler_ccf, rrate = bw/samp_rate, nfilts 32, atten 100, taps ""
 * Neither block's source is under /root/reference (GNU Radio is an un-vendored, un-versioned dependency:
 * find_package(Gnuradio) in CMakeLists.txt:123; the flowgraph says 3.7.10), so this file restates their
 * published algorithms:
 *   gr-blocks  rotator.h (rotate(): out = in*phase; phase *= incr; every 512 samples phase /= |phase|)
 *   gr-filter  pfb_arb_resampler.cc (kernel::pfb_arb_resampler_ccf: create_taps, create_diff_taps, set_rate,
 *              filter) on top of kernel::fir_filter_ccf (reversed taps, VOLK generic dot product, sequential
 *              float accumulation)
 *   gr-filter  python/pfb.py arb_resampler_ccf.create_taps (rate < 1: firdes.low_pass_2 with a
 *              Blackman-harris window), firdes.cc low_pass_2 / compute_ntaps_windes, window.cc blackman_harris
 * PARITY UNPINNED: no fixture of the reference covers these blocks; the oracle is pinned only against an
 * independent numpy double-precision recomputation (tests/test_chan_oracle.py).
 */
#ifndef CHAN_ORACLE_H
#define CHAN_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "lora_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pfb.arb_resampler_ccf.create_taps for rate < 1.  Returns the number of taps; writes min(cap, n) of them. */
size_t lo_chan_design_taps(double rate, unsigned nfilts, double atten_db, float *taps, size_t cap);

/* blocks::rotator_cc(phase_inc): in-place allowed.  *phase (re, im) and *counter carry the state across calls;
 * start with phase = (1, 0), counter = 0. */
void lo_rotator(double phase_inc, const lo_c32 *in, lo_c32 *out, size_t n, lo_c32 *phase, unsigned *counter);

typedef struct lo_pfb lo_pfb;
lo_pfb *lo_pfb_create(float rate, const float *taps, size_t ntaps, unsigned nfilts);
void lo_pfb_destroy(lo_pfb *p);
unsigned lo_pfb_taps_per_filter(const lo_pfb *p);
unsigned lo_pfb_dec_rate(const lo_pfb *p);
float lo_pfb_flt_rate(const lo_pfb *p);
unsigned lo_pfb_last_filter(const lo_pfb *p);
/* The block on a flat array: the stream is (taps_per_filter - 1 zeros of history, kept from the previous call)
 * ++ in[0..n_in); kernel::filter runs with n_to_read = n_in.  Returns the number of outputs written (stops early
 * if out_cap is reached).  use_double != 0 accumulates the two dot products in double (margin analysis);
 * otherwise sequential float, as VOLK's generic kernel does. */
size_t lo_pfb_work(lo_pfb *p, const lo_c32 *in, size_t n_in, lo_c32 *out, size_t out_cap, int use_double);
/* The index recurrence alone (data-independent): for the next n_in inputs writes, per output, the polyphase
 * position 32*i_in + j relative to the first input of this call, and the interpolation weight d_acc. */
size_t lo_pfb_schedule(lo_pfb *p, size_t n_in, uint64_t *pos, float *acc, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
