/*
 * lora_hip_chan.h -- C ABI of the MI355X (gfx950) channeliser that feeds the demod hot path.
 *
 * The reference's receive flowgraph (examples/rx_usrp.grc) brings one LoRa channel of a wideband capture to
 * the 1 sample/chip stream lora.demod needs (README.md:15) with two stock GNU Radio blocks:
 *     blocks_rotator_cc        phase_inc = 2*pi*offset/samp_rate            examples/rx_usrp.grc:489-503
 *     pfb_arb_resampler_ccf    rrate = bw/samp_rate, nfilts 32, atten 100   examples/rx_usrp.grc:825-860
 * This library does both in one pass over the samples in HBM: frequency shift while the tile is loaded, then
 * the polyphase filter bank (32 arms, linear interpolation between an arm and its difference filter).  The
 * blocks' sources are GNU Radio's (not under the reference tree, version unpinned): the algorithms restated
 * are gr-blocks rotator.h and gr-filter pfb_arb_resampler.cc / fir_filter.cc / firdes.cc / pfb.py.
 *
 * Plain C: pointers and sizes.  Error codes, lora_c32 and the flag bits are those of lora_hip_demod.h.
 * A handle is a stream block with state (filter arm, interpolation accumulator, history, phase); it is used by
 * one thread at a time; distinct handles are independent (one per channel and per device).
 */
#ifndef LORA_HIP_CHAN_H
#define LORA_HIP_CHAN_H

#include "lora_hip_demod.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lora_hip_chan lora_hip_chan_t;

/* Replaces pfb.arb_resampler_ccf.create_taps (gr-filter python/pfb.py; what the flowgraph's empty `taps`
 * parameter selects, rx_usrp.grc:853-854) for rate < 1: firdes.low_pass_2(nfilts, nfilts, 0.4*rate, 0.2*rate,
 * atten_db, WIN_BLACKMAN_HARRIS).  Writes min(cap, n) taps, stores n in *ntaps.  rate >= 1 (Parks-McClellan
 * design in GNU Radio) is not restated: LORA_HIP_EUNSUPPORTED, pass taps of your own to create(). */
LORA_HIP_API int lora_hip_chan_design_taps(double rate, unsigned nfilts, double atten_db, float *taps, size_t cap,
                                           size_t *ntaps);

/* Replaces blocks.rotator_cc(phase_inc) + filter.pfb_arb_resampler_ccf(rate, taps, nfilts)
 * (rx_usrp.grc:489-503, :825-860).  taps: the prototype filter at nfilts times the input rate (copied);
 * phase_inc 0 leaves the rotation out. */
LORA_HIP_API int lora_hip_chan_create(lora_hip_chan_t **h, int device, double phase_inc, float rate,
                                      const float *taps, size_t ntaps, unsigned nfilts);
LORA_HIP_API void lora_hip_chan_destroy(lora_hip_chan_t *h);
LORA_HIP_API int lora_hip_chan_set_stream(lora_hip_chan_t *h, void *hip_stream);
LORA_HIP_API const char *lora_hip_chan_last_hip_error(const lora_hip_chan_t *h);
/* back to the state after create(): history zeroed, first arm, phase 0 */
LORA_HIP_API int lora_hip_chan_reset(lora_hip_chan_t *h);

/* Put the stream state where it is after n_consumed input samples of a stream that started at reset(): filter
 * arm, interpolation weight, input overshoot and rotation phase come from the closed form of the reference's
 * recurrence (no pass over the samples); `history` holds the taps_per_filter - 1 samples before that position
 * (host pointer, or device pointer with LORA_HIP_IQ_ON_DEVICE; NULL = zeros).  work() on the samples from
 * n_consumed on then produces exactly what the sequential stream produces there, so one long wideband capture
 * can be cut into time segments, one per GPU, with taps_per_filter - 1 samples of overlap and no exchange
 * between the GPUs (no counterpart in the reference, which is one sequential thread per block). */
LORA_HIP_API int lora_hip_chan_seek(lora_hip_chan_t *h, uint64_t n_consumed, const lora_c32 *history, unsigned flags);

/* Number of output samples the next work() call with n_in inputs will produce (the index recurrence of
 * kernel::pfb_arb_resampler_ccf::filter is data-independent).  Does not change the state. */
LORA_HIP_API size_t lora_hip_chan_output_count(lora_hip_chan_t *h, size_t n_in);

/* Replaces one rotator work() + one pfb_arb_resampler general_work() over n_in input samples:
 *     x[n]   = in[n] * exp(i * phase_inc * n_abs)                      rotator::rotate
 *     out[m] = fir(arm j_m, x at i_m) + d_acc_m * fir(diff arm j_m, x at i_m)   pfb_arb_resampler_ccf::filter
 * with (i_m, j_m, d_acc_m) from the reference recurrence (j += dec_rate + floor(d_acc); d_acc = fmodf(..)).
 * The taps_per_filter - 1 samples of history and all counters carry over to the next call, so that any
 * chunking of a stream gives the same output.  out_cap must be at least lora_hip_chan_output_count(n_in)
 * (else LORA_HIP_EINVAL, nothing is consumed).  flags: LORA_HIP_IQ_ON_DEVICE (in), LORA_HIP_OUT_ON_DEVICE
 * (out), LORA_HIP_ASYNC (both on the device: return without synchronising the stream). */
LORA_HIP_API int lora_hip_chan_work(lora_hip_chan_t *h, const lora_c32 *in, size_t n_in, lora_c32 *out,
                                    size_t out_cap, size_t *n_out, unsigned flags);

/* geometry of the handle */
LORA_HIP_API unsigned lora_hip_chan_taps_per_filter(const lora_hip_chan_t *h);
LORA_HIP_API unsigned lora_hip_chan_dec_rate(const lora_hip_chan_t *h);     /* d_dec_rate */
LORA_HIP_API float lora_hip_chan_flt_rate(const lora_hip_chan_t *h);        /* d_flt_rate */
/* the rotation applied per sample, in radians: the angle of the reference's increment exp(i*(float)phase_inc)
 * after rounding to complex<float> and normalising (rotator::set_phase_incr); differs from phase_inc by up to
 * ~1e-7 rad, which is 6 rad over 64 M samples */
LORA_HIP_API double lora_hip_chan_phase_inc_effective(const lora_hip_chan_t *h);
/* name of the kernel work() dispatches to: "chan_decim_kernel<D,..>" for whole-number decimation (d_flt_rate
 * 0, d_dec_rate = D * nfilts, D in 2..8, 10, 12, 16), "chan_arb_kernel" for other rates whose filter banks fit LDS,
 * "chan_general_kernel" otherwise */
LORA_HIP_API const char *lora_hip_chan_kernel_name(const lora_hip_chan_t *h);
/* Non-finite input.  pfb_arb_resampler_ccf multiplies exactly the taps_per_filter samples of an output's window.  The
 * whole-number-decimation kernel pads every polyphase component with zero taps to a multiple of eight and sums over
 * lanes that carry zero taps, so an Inf or NaN sample up to 8*D samples outside an output's true window still meets
 * a zero tap (0*Inf = NaN) and that output becomes NaN where the reference's stays finite.  Finite input -- anything a
 * receiver delivers -- is not affected; callers that must reproduce the reference on non-finite samples use
 * lora_hip_chan_force_general. */
/* Rates above nfilts keep per-output positions in 32 bits: work() feeds such inputs through in pieces of at most this
 * many samples (2^31 / nfilts), which gives the same stream as one call. */
LORA_HIP_API size_t lora_hip_chan_piece_limit(unsigned nfilts);
/* developer knob: 1 = always run the general kernel (the one that follows the reference's summation order) */
LORA_HIP_API int lora_hip_chan_force_general(lora_hip_chan_t *h, int on);

#ifdef __cplusplus
}
#endif
#endif /* LORA_HIP_CHAN_H */
