/*
 * lora_demod_core.h -- GNU-Radio-free host side of the demod block, above lora_hip_demod.h.
 *
 * Replaces the body of gr::lora::demod_impl (reference: lib/demod_impl.cc:54-104 constructor,
 * :149-450 general_work) minus the GNU Radio plumbing: it owns the stream framing
 * (set_history(3N), :103), the integer sync/framing state machine (:214-437), the squelch
 * (:133-137) and PDU assembly (:420-422), and obtains every (argmax, max) pair from the GPU
 * through lora_hip_demod_batch_strided in speculative batches instead of one FFT per call.
 * The sequence of PDUs equals what the reference emits for the same sample stream.
 *
 * The GNU Radio adapter (gr-lora_amd/gr_adapter/) subclasses gr::lora::demod, feeds
 * general_work's new input items to lora_demod_core_work and publishes each PDU as
 * pmt::cons(pmt::make_dict(), pmt::init_u16vector(n, symbols)) on message port "out".
 */
#ifndef LORA_DEMOD_CORE_H
#define LORA_DEMOD_CORE_H

#include "lora_hip_demod.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lora_demod_core lora_demod_core_t;

/* include/lora/demod.h:38-46 */
enum lora_demod_state {
  LORA_S_RESET = 0, LORA_S_PREFILL, LORA_S_DETECT_PREAMBLE, LORA_S_SFD_SYNC,
  LORA_S_READ_HEADER, LORA_S_READ_PAYLOAD, LORA_S_OUT
};

/* One PDU: the u16vector of lib/demod_impl.cc:420 (first 8 symbols /4, then full SF bits or /4
 * under low_data_rate, trailing squelch symbol included; not gray-mapped). */
typedef void (*lora_pdu_cb)(void *user, const uint16_t *symbols, size_t n_symbols);

/* One row per emulated general_work call (for tests and debugging). */
typedef struct {
  uint64_t pos;        /* read index into the framed stream (3N-1 zeros ++ samples fed so far) */
  uint16_t state;      /* d_state on entry */
  uint16_t chirp_off;  /* downchirp offset of the up path (d_offset in READ_*, else 0) */
  uint16_t idx;        /* argmax of the up path */
  uint16_t consumed;   /* num_consumed (:443) */
  float    val;        /* d_power */
  uint32_t n_sfd;      /* SFD windows evaluated in this call */
} lora_core_trace;

/* gr::lora::demod::make(spreading_factor, low_data_rate, beta, fft_factor)
 * (include/lora/demod.h:66-69, lib/demod_impl.cc:41-49) plus the GPU index. */
LORA_HIP_API int lora_demod_core_create(lora_demod_core_t **c, int device, unsigned spreading_factor,
                                        int low_data_rate, float beta, unsigned fft_factor);
LORA_HIP_API void lora_demod_core_destroy(lora_demod_core_t *c);
/* Start a new, independent stream on the same block: everything is as after create (state S_RESET,
 * empty histories, 3N-1 zeros of history, counters and trace position at zero; a trace buffer stays set),
 * but the tables and the GPU buffers are kept.  Equivalent to destroying the block and making it again
 * with the same arguments; meant for demodulating many captures one after the other. */
LORA_HIP_API int lora_demod_core_reset(lora_demod_core_t *c);

/* Feed n_in new samples (host memory).  Runs every general_work call that GNU Radio's scheduler
 * could run with the samples fed so far (a call needs N + history-1 = 4N-1 samples from its read
 * index: forecast :142-147 with noutput_items = 1) and reports PDUs through cb in stream order.
 * Samples that are not yet consumable stay buffered inside the core. */
LORA_HIP_API int lora_demod_core_work(lora_demod_core_t *c, const lora_c32 *in, size_t n_in,
                                      lora_pdu_cb cb, void *user);

/* The same for samples that are already in HBM of the block's GPU (a device pointer), e.g. the output of the
 * channeliser (lora_hip_chan.h): the windows are read where they lie, nothing goes through the host except the
 * (argmax, max) pairs and the < 4N samples of history a call sequence leaves unconsumed.  No counterpart in the
 * reference (its input buffer is the scheduler's host memory, :155).  The samples must be complete when the
 * call is made, or be produced on the stream given to lora_demod_core_set_stream. */
LORA_HIP_API int lora_demod_core_work_device(lora_demod_core_t *c, const lora_c32 *in_device, size_t n_in,
                                             lora_pdu_cb cb, void *user);
/* PDUs without a callback per PDU (a foreign-function callback costs more than the block spends on a packet): with
 * collecting on, each work() call also keeps its PDUs inside the block -- PDU i is symbols[offsets[i] .. offsets[i+1]) --
 * until the next work(), reset() or set_collect() call.  cb may then be NULL. */
LORA_HIP_API int lora_demod_core_set_collect(lora_demod_core_t *c, int on);
LORA_HIP_API int lora_demod_core_collected(const lora_demod_core_t *c, const uint16_t **symbols,
                                           const uint64_t **offsets, size_t *n_pdus);
/* Run the block's GPU work on a caller-owned hipStream_t (void*); NULL restores the block's own stream. */
LORA_HIP_API int lora_demod_core_set_stream(lora_demod_core_t *c, void *hip_stream);

/* Time-segment sharding of one long capture (design A; no counterpart in the
 * reference, which is one sequential thread per channel).  Outside READ_HEADER / READ_PAYLOAD the
 * up-path result of a call (lib/demod_impl.cc:177-206 with downchirp offset 0) depends only on its
 * read index, and read indices stay on the N/16 lattice (num_consumed is N or ol*N/16 + 5N/4, :344).
 * Those results can therefore be produced ahead of time, by any GPU, with
 * lora_hip_demod_batch_strided(pos0, stride = N, chirp_off = 0, kind = LORA_WIN_UP) for each of the
 * 16 lattice phases, and handed over here before the samples are fed: work() then replays the
 * reference's state machine over them, so the window grid is the sequential run's by construction,
 * and launches only the SFD sweeps and header/payload windows itself.
 * idx[k], val[k] = result of the window starting at framed-stream index start + k*N (the framed
 * stream is 3N-1 zeros followed by the samples, as in lora_core_trace.pos); start must be a multiple
 * of N/16.  The arrays are copied.  Results that never match a call are simply not used. */
LORA_HIP_API int lora_demod_core_preload(lora_demod_core_t *c, uint64_t start, size_t count,
                                         const uint16_t *idx, const float *val);

/* Planned evaluation.  A long input is not evaluated one packet at a time: per work() call the block sweeps the whole
 * N/16 lattice for preambles (all 16 phases), evaluates every SFD search the preamble rule (:247-273) can start, dry-runs
 * the state machine to find the header/payload windows, evaluates those, and only then runs the state machine for
 * real -- a fixed number of launches whatever the number of packets; whatever the plan did not foresee is evaluated on
 * demand.  The PDUs and the trace are the same either way (a plan only chooses which windows are evaluated early).
 * mode 0: off (every window on demand, in speculative sweeps); 1 (default): inputs of at least 1024 symbols;
 * 2: every input.  The environment variable LORA_CORE_PLAN sets the default of new blocks. */
LORA_HIP_API int lora_demod_core_set_plan(lora_demod_core_t *c, int mode);
/* plans started, READ stages redone after a surprise, windows that had to be evaluated on demand (each = one launch) */
LORA_HIP_API uint64_t lora_demod_core_plans(const lora_demod_core_t *c);
LORA_HIP_API uint64_t lora_demod_core_replans(const lora_demod_core_t *c);
LORA_HIP_API uint64_t lora_demod_core_on_demand(const lora_demod_core_t *c);
/* times the first pass of a plan met a header/payload window whose real maximum fell on the other side of the squelch
 * threshold than predicted: the block then resumes from the S_RESET call before that packet, evaluating on demand */
LORA_HIP_API uint64_t lora_demod_core_rollbacks(const lora_demod_core_t *c);

/* For tests: the preamble rule of :247-273 over a sweep of lattice results, as the planned block's SFD stage evaluates it
 * (16-bit vector compares, squelch included): position j >= 48 is reported when the bins at j-16, j-32, j-48 -- the three
 * older windows of j's lattice phase -- lie within LORA_PREAMBLE_TOLERANCE of the bin at j and val[j] > threshold.  hits
 * must hold count entries; returns how many were written (ascending).  path: 0 = what the block uses on this CPU,
 * 1 = the SSE2 form, 2 = the AVX2 + BMI2 form (returns (size_t)-1 when the CPU lacks them), 3 = the scalar definition. */
LORA_HIP_API size_t lora_demod_core_rule_hits(const uint16_t *idx, const float *val, float threshold, size_t count,
                                              uint32_t *hits, int path);

/* Sharded replay of one long capture (no counterpart in the reference).  The block's state is determined
 * by its read index alone at every S_RESET call (:215-229 clears the rest) and at every S_DETECT_PREAMBLE call (the
 * argmax history then holds the four windows before the read index, :207-212, :233-242): two blocks that make such a call
 * at the same framed-stream index emit the same PDUs from there on.  So a GPU can demodulate a time segment on its own,
 * started anywhere, and its output is the sequential block's from the first index at which both are idle together.
 *   seek: like reset, but the stream starts at framed-stream index pos (a multiple of N/16) in S_RESET with no history
 *   zeros: the first sample fed afterwards is framed-stream index pos (trace rows and preload use the same indices).
 *   idle_log: every maximal run of consecutive S_DETECT_PREAMBLE calls since the last reset/seek, as first and last read
 *   index (N apart) and the number of PDUs emitted before it; valid until the next call on the block. */
typedef struct { uint64_t first, last, pdus_before; } lora_core_idle;
LORA_HIP_API int lora_demod_core_seek(lora_demod_core_t *c, uint64_t pos);
LORA_HIP_API int lora_demod_core_idle_log(const lora_demod_core_t *c, const lora_core_idle **runs, size_t *n_runs);

/* Optional per-call trace: rows are appended to `rows` (caller-owned) up to max_rows; the number
 * of calls so far (may exceed max_rows) is returned by lora_demod_core_calls(). */
LORA_HIP_API int lora_demod_core_set_trace(lora_demod_core_t *c, lora_core_trace *rows, size_t max_rows);
LORA_HIP_API uint64_t lora_demod_core_calls(const lora_demod_core_t *c);
/* GPU work issued so far: symbol windows evaluated and kernel launches */
LORA_HIP_API uint64_t lora_demod_core_windows(const lora_demod_core_t *c);
LORA_HIP_API uint64_t lora_demod_core_launches(const lora_demod_core_t *c);
LORA_HIP_API int lora_demod_core_state(const lora_demod_core_t *c);
LORA_HIP_API const char *lora_demod_core_last_error(const lora_demod_core_t *c);

#ifdef __cplusplus
}
#endif
#endif /* LORA_DEMOD_CORE_H */
