/*
 * lora_hip_demod.h -- C ABI of the MI355X (gfx950) LoRa chirp demodulator hot path.
 *
 * Drop-in boundary for the per-symbol numeric loop of gr-lora's demod block
 * (reference: lib/demod_impl.cc:149-450, demod_impl::general_work).  Everything here is plain C:
 * pointers and sizes, no C++/torch/HIP types in the signatures.  All functions return 0 on
 * success or a negative LORA_HIP_E* code; nothing throws across the boundary.
 *
 * Threading: a handle is used by one thread at a time; distinct handles are independent and may
 * live on different GPUs (one handle per device for multi-GPU).  No global mutable state.
 */
#ifndef LORA_HIP_DEMOD_H
#define LORA_HIP_DEMOD_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define LORA_HIP_API __attribute__((visibility("default")))   /* mirrors LORA_API, include/lora/api.h */
#else
#define LORA_HIP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* interleaved complex64 == gr_complex == std::complex<float> == float2 */
typedef struct { float re, im; } lora_c32;

typedef struct lora_hip_demod lora_hip_demod_t;

enum {
  LORA_HIP_OK = 0,
  LORA_HIP_EINVAL = -1,        /* bad argument (NULL, sf outside 6..12, window past the buffer) */
  LORA_HIP_EUNSUPPORTED = -2,  /* 2^sf * fft_factor >= 65536: wraps the reference's unsigned short d_fft_size */
  LORA_HIP_ENOMEM = -3,        /* host or device allocation failed */
  LORA_HIP_EHIP = -4,          /* a HIP runtime call failed; see lora_hip_last_hip_error() */
  LORA_HIP_ENODEVICE = -5      /* no usable gfx950 device / device index out of range */
};

/* window kinds: which dechirp the window gets */
enum {
  LORA_WIN_UP = 0,   /* in * downchirp[chirp_off + n] * kaiser[n]   demod_impl.cc:177-192 */
  LORA_WIN_SFD = 1   /* in * upchirp[chirp_off + n], no window       demod_impl.cc:300-310 */
};

/* flags for the batch calls */
enum {
  LORA_HIP_IQ_ON_DEVICE = 1,    /* iq is a device pointer (HBM resident), else host memory */
  LORA_HIP_OUT_ON_DEVICE = 2,   /* max_idx / max_val are device pointers, else host memory */
  LORA_HIP_DESC_ON_DEVICE = 4,  /* desc is a device pointer */
  LORA_HIP_ASYNC = 8,           /* do not synchronise the handle's stream before returning: valid when the samples are on
                                   the device or staged, descriptors (if any) on the device, and the results go to device
                                   memory or, with LORA_HIP_OUT_PINNED, to page-locked host memory, which then holds them
                                   once the lane has been synchronised (lora_hip_demod_sync) */
  LORA_HIP_IQ_STAGED = 16,      /* iq is ignored: use the samples last given to
                                   lora_hip_demod_stage_iq (n_samples must not exceed them) */
  LORA_HIP_OUT_PINNED = 32,     /* max_idx / max_val are page-locked host memory (lora_hip_host_alloc): the results
                                   are copied straight into them instead of through the handle's own pinned mirror */
  LORA_HIP_NO_VAL = 64,         /* max_val may be NULL and is not written (host outputs only): half the copy back
                                   for callers that need the bins only, like the SFD search (demod_impl.cc:322) */
  LORA_HIP_OUT_DIRECT = 256,    /* with LORA_HIP_OUT_PINNED: the kernel writes the results into the page-locked arrays itself
                                   instead of into the handle's device block with two copies behind it; for small launches,
                                   where the copies' dispatch is what takes time (a large one is faster through the copies) */
  LORA_HIP_LANE1 = 128          /* run on the handle's second lane: a stream of the handle's own with its own counters and
                                   staging, so that a small launch (a block's header/payload windows, :363-413) overtakes
                                   the long sweeps queued on the first lane.  Samples must be staged or on the device.
                                   Launches on one lane are ordered; the two lanes may overlap in time */
};

/* one symbol window: N = 2^sf samples starting at iq[pos] */
typedef struct {
  uint64_t pos;        /* first sample of the window                                   */
  uint16_t chirp_off;  /* offset into the 2N-entry chirp table (d_offset), 0 <= off < N  */
  uint8_t  kind;       /* LORA_WIN_UP / LORA_WIN_SFD                                    */
  uint8_t  reserved[5];
} lora_win_desc;       /* 16 bytes */

LORA_HIP_API const char *lora_hip_strerror(int code);
/* text of the last HIP error seen by this handle (valid until the next call on it) */
LORA_HIP_API const char *lora_hip_last_hip_error(const lora_hip_demod_t *h);

/*
 * Host-side table builders.  They restate, in double/float exactly as the reference does:
 *   chirp tables  -- lib/demod_impl.cc:92-101 (float phase, double accumulator, polar in double)
 *   Kaiser window -- fft::window::build(WIN_KAISER, N, beta), call site lib/demod_impl.cc:85
 * up_2n / down_2n hold 2*2^sf entries, window_n holds 2^sf.
 */
LORA_HIP_API int lora_hip_make_chirp_tables(unsigned sf, lora_c32 *up_2n, lora_c32 *down_2n);
LORA_HIP_API int lora_hip_make_kaiser(unsigned ntaps, double beta, float *window_n);

/*
 * Replaces the constructor state of demod_impl (lib/demod_impl.cc:54-104): FFT plan (:81),
 * window (:85) and chirp tables (:92-101) are copied to HBM of `device`; the caller keeps
 * ownership of the three host arrays.  Passing NULL for all three makes the library build them
 * with the functions above from (sf, beta).  One HIP stream is created per handle.
 * fft_factor: any value >= 1 with 2^sf * fft_factor < 65536, as in the reference (its FFT size is
 * fft_factor * 2^sf handed to FFTW, demod_impl.cc:80-81, kept in an unsigned short).  1, 2 and 4 -- the
 * values the reference's flowgraphs use -- run specialised kernels; every other value runs a generic one
 * (run-time number of N-point sub-transforms), same results contract, lower throughput.
 */
LORA_HIP_API int lora_hip_demod_create(lora_hip_demod_t **h, int device, unsigned sf,
                                       unsigned fft_factor, float beta, const float *window_n,
                                       const lora_c32 *upchirp_2n, const lora_c32 *downchirp_2n);
LORA_HIP_API void lora_hip_demod_destroy(lora_hip_demod_t *h);

/* Run the batch calls on a caller-owned hipStream_t (passed as void*); NULL restores the
 * handle's own stream. */
LORA_HIP_API int lora_hip_demod_set_stream(lora_hip_demod_t *h, void *hip_stream);

/*
 * The hot path.  For each window i < n_win this computes what one general_work call computes
 * between lib/demod_impl.cc:177 and :206 (kind UP) or one iteration of the SFD loop
 * :297-322 (kind SFD):
 *     x[n]  = iq[pos+n] * chirp[chirp_off+n]  (* kaiser[n] for UP), n < N
 *     X     = forward DFT of x zero-padded to M = N*fft_factor points
 *     max_idx[i] = lowest k with maximal |X[k]|^2   (demod_impl::argmax, :114-140)
 *     max_val[i] = that |X[k]|^2                    (d_power when update_squelch)
 * A window with pos+N > n_samples is an error (LORA_HIP_EINVAL) when the descriptors are on the
 * host; descriptors on the device are guarded in the kernel and yield max_idx 0xFFFF, val -1.
 */
LORA_HIP_API int lora_hip_demod_batch(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples,
                                      const lora_win_desc *desc, size_t n_win,
                                      uint16_t *max_idx, float *max_val, unsigned flags);

/* Same for `count` windows at pos0 + i*stride with one chirp_off and kind: the stride-N sweep of
 * S_DETECT_PREAMBLE / S_READ_* (consume_each(N), :443) and the stride-N/16 sweep of S_SFD_SYNC
 * when chirp_off_step != 0: window i uses chirp_off = (chirp_off + i*chirp_off_step) mod N
 * (d_offset = (ol*N/16) % N, :297). */
LORA_HIP_API int lora_hip_demod_batch_strided(lora_hip_demod_t *h, const lora_c32 *iq,
                                              size_t n_samples, uint64_t pos0, uint64_t stride,
                                              size_t count, unsigned chirp_off,
                                              unsigned chirp_off_step, unsigned kind,
                                              uint16_t *max_idx, float *max_val, unsigned flags);

/*
 * Several strided sweeps in ONE launch: run r contributes `count` windows at pos0 + i*stride with chirp offset
 * (chirp_off + i*chirp_off_step) mod N, and the results of all runs lie back to back in run order.  This is what lets
 * a host-side block evaluate every SFD search (:295-357) or every header/payload read (:363-413) of a long input in
 * one launch instead of one per packet.  A small kernel expands the runs into window descriptors in HBM, then the
 * same kernel as lora_hip_demod_batch runs over them.  Runs are validated on the host like host descriptors.
 */
typedef struct {
  uint64_t pos0;            /* first window                                   */
  uint32_t stride;          /* samples between windows                        */
  uint32_t count;           /* windows in this run                            */
  uint16_t chirp_off;       /* chirp offset of the first window, < N          */
  uint16_t chirp_off_step;  /* added per window, mod N                        */
  uint8_t  kind;            /* LORA_WIN_UP / LORA_WIN_SFD                     */
  uint8_t  reserved[3];
} lora_win_run;             /* 24 bytes */
LORA_HIP_API int lora_hip_demod_batch_runs(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples,
                                           const lora_win_run *runs, size_t n_runs,
                                           uint16_t *max_idx, float *max_val, unsigned flags);

/* Wait until everything queued on a lane (0: the handle's stream or the one given to lora_hip_demod_set_stream, 1: the
 * second lane) has run: what a caller of LORA_HIP_ASYNC calls before it reads page-locked results.  No counterpart in the
 * reference (its general_work is synchronous, demod_impl.cc:149-450). */
LORA_HIP_API int lora_hip_demod_sync(lora_hip_demod_t *h, int lane);

/* Marks, for a caller that has several LORA_HIP_ASYNC calls queued on a lane and wants the results of one of them:
 * lora_hip_demod_mark sets mark `mark` (0 <= mark < LORA_HIP_MARKS) behind everything queued on `lane` so far,
 * lora_hip_demod_wait returns once the lane has reached it (at once for a mark that was never set).  Setting a mark
 * again moves it.  No counterpart in the reference. */
#define LORA_HIP_MARKS 8
LORA_HIP_API int lora_hip_demod_mark(lora_hip_demod_t *h, int lane, int mark);
LORA_HIP_API int lora_hip_demod_wait(lora_hip_demod_t *h, int mark);

/* Page-locked host memory for LORA_HIP_OUT_PINNED result buffers (hipHostMalloc / hipHostFree). */
LORA_HIP_API int lora_hip_host_alloc(void **p, size_t bytes);
LORA_HIP_API void lora_hip_host_free(void *p);

/*
 * lora_hip_demod_batch_strided for callers that cannot allocate.  It has no counterpart in the reference: it is
 * for code that must not reach hipMalloc or hipFree -- work recorded under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same windows, same kernels and same results, bit for bit, as
 * lora_hip_demod_batch_strided with LORA_HIP_IQ_ON_DEVICE | LORA_HIP_OUT_ON_DEVICE | LORA_HIP_ASYNC; the only device
 * memory the kernels use beyond the caller's buffers and the handle's tables, their window-group ticket counters
 * (4 KiB), lies in a file-scope __device__ array of the library (LORA_HIP_NOALLOC_MAX_HANDLES slots of 4 KiB = 64 KiB
 * per GPU, held until the process ends) instead of a buffer the handle allocated.
 *   - iq_dev, max_idx_dev and max_val_dev are device pointers on the handle's GPU; nothing is copied to or from
 *     the host and nothing is allocated or freed (no hipMalloc/hipFree/hipHostMalloc, no growth of staging buffers).
 *   - The call only enqueues the kernel on the handle's stream (lora_hip_demod_set_stream); the caller synchronises.
 *     The counters are zero when the library is loaded and the kernel's last workgroup returns them to zero, so no
 *     memset is enqueued (only after a HIP error on this handle does the next call clear them first).
 *   - count above LORA_HIP_NOALLOC_MAX_WINDOWS is refused with LORA_HIP_EINVAL (text in lora_hip_last_hip_error).
 *   - Every handle owns one slot of its GPU's scratch array, claimed at lora_hip_demod_create and given back at
 *     lora_hip_demod_destroy: calls on different handles never share counters, whatever their streams or threads.  A
 *     handle created while LORA_HIP_NOALLOC_MAX_HANDLES others are alive on the same GPU has no slot: every other entry
 *     point works on it, this one returns LORA_HIP_ENOMEM (text in lora_hip_last_hip_error).
 *   - Calls on ONE handle share that handle's slot, which keeps the last call's state (zeros after a launch that ran to
 *     its end): queued on one stream they are ordered and safe; overlapping in time (the stream changed between two
 *     calls that are both still running) they race on the counters.  A handle is used by one thread at a time anyway.
 *   - Diagnostic builds of the library (LORA_CLOCK, LORA_STAMPS) allocate inside a launch; there this entry point
 *     returns LORA_HIP_EUNSUPPORTED.
 */
#define LORA_HIP_NOALLOC_MAX_WINDOWS ((size_t)1 << 30)
#define LORA_HIP_NOALLOC_MAX_HANDLES 16
LORA_HIP_API int lora_hip_demod_batch_strided_noalloc(lora_hip_demod_t *h, const lora_c32 *iq_dev,
                                                      size_t n_samples, uint64_t pos0, uint64_t stride,
                                                      size_t count, unsigned chirp_off,
                                                      unsigned chirp_off_step, unsigned kind,
                                                      uint16_t *max_idx_dev, float *max_val_dev);

/* Copy n_samples host samples into the handle's own HBM staging buffer once, so that several
 * batch calls (flag LORA_HIP_IQ_STAGED) can sweep them without another host-to-device copy.
 * This is what replaces the scheduler-owned input buffer of general_work (:155). */
LORA_HIP_API int lora_hip_demod_stage_iq(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples);
/* The same for a capture that lies in two host pieces, staged back to back (a then b): the samples a
 * block carried over from its previous call (set_history, :103) and the new input items, without
 * joining them on the host first. */
LORA_HIP_API int lora_hip_demod_stage_iq2(lora_hip_demod_t *h, const lora_c32 *a, size_t n_a,
                                          const lora_c32 *b, size_t n_b);

/* The same with the second piece already in HBM (a device pointer on the handle's GPU): a host-side history in
 * front of samples that an upstream GPU stage (the channeliser, lora_hip_chan.h) produced. */
LORA_HIP_API int lora_hip_demod_stage_iq2_dev(lora_hip_demod_t *h, const lora_c32 *a_host, size_t n_a,
                                              const lora_c32 *b_device, size_t n_b);
/* Copy n samples from HBM to host memory on the handle's stream and wait for them: how a host-side block
 * keeps the few unconsumed samples of a device-resident input as its history (set_history, :103). */
LORA_HIP_API int lora_hip_demod_fetch(lora_hip_demod_t *h, const lora_c32 *src_device, size_t n, lora_c32 *dst_host);

/* geometry of the handle */
LORA_HIP_API unsigned lora_hip_demod_sf(const lora_hip_demod_t *h);
LORA_HIP_API unsigned lora_hip_demod_fft_factor(const lora_hip_demod_t *h);
/* name of the kernel the handle dispatches to (for profiles) */
LORA_HIP_API const char *lora_hip_demod_kernel_name(const lora_hip_demod_t *h);

#ifdef __cplusplus
}
#endif
#endif /* LORA_HIP_DEMOD_H */
