// demod_core.cpp -- host side of the demod block (include/lora_demod_core.h).
//
// The reference evaluates one FFT per general_work call and decides what to look at next from
// the result (lib/demod_impl.cc:214-437).  Here the same integer state machine runs on the host
// over (argmax, max) pairs that the GPU produced in speculative batches:
//   * up-path windows lie on a stride-N lattice until the next SFD lock, so a miss at (pos, off)
//     launches one strided batch from pos to the end of the buffered samples; the batches of
//     every (lattice phase, offset) seen so far are kept, so a lattice that comes back after a
//     later packet (there are only 16 phases) costs no new launch;
//   * entering S_SFD_SYNC launches one stride-N/16 batch that covers the at most 10 calls x 16
//     overlapped windows the reference can evaluate before it bails out (:283-292).
// Every decision (tolerances, counters, integer divisions, squelch compare) is the reference's.
// Product code: never calls anything under oracle/.
#include "lora_demod_core.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// include/lora/demod.h:28-33
#define DEMOD_HISTORY_DEPTH 3
#define REQUIRED_PREAMBLE_CHIRPS 4
#define REQUIRED_SFD_CHIRPS 2
#define LORA_SFD_TOLERANCE 1
#define LORA_PREAMBLE_TOLERANCE 1
#define DEMOD_SYNC_RECOVERY_COUNT (8 - REQUIRED_PREAMBLE_CHIRPS) + (2 - REQUIRED_SFD_CHIRPS) + 4
// lib/demod_impl.cc:35-36
#define OVERLAP_DEFAULT 1
#define OVERLAP_FACTOR 16

namespace {

// results of one strided GPU batch: window k starts at start + k*stride
struct Run {
  uint64_t start = 0, stride = 1;
  unsigned off0 = 0, off_step = 0, kind = 0;
  std::vector<uint16_t> idx;
  std::vector<float> val;
  bool find(uint64_t pos, unsigned off, unsigned n, size_t *k) const {
    if (idx.empty() || pos < start) return false;
    const uint64_t d = pos - start;
    if (d % stride) return false;
    const uint64_t q = d / stride;
    if (q >= idx.size()) return false;
    if (((off0 + q * off_step) & (n - 1)) != off) return false;
    *k = (size_t)q;
    return true;
  }
};

}  // namespace

struct lora_demod_core {
  // configuration (demod_impl.h:44-49)
  unsigned d_sf = 0; bool d_ldr = false; float d_beta = 0; unsigned d_fft_size_factor = 0;
  unsigned d_num_symbols = 0, d_fft_size = 0;
  lora_hip_demod_t *hip = nullptr;
  // reference state (demod_impl.h:43-70)
  int d_state = LORA_S_RESET;
  unsigned d_overlaps = OVERLAP_DEFAULT, d_offset = 0;
  float d_power = .000000001f, d_threshold = 0.005f;      // :87-88
  bool d_squelched = false;
  unsigned d_preamble_idx = 0, d_sfd_idx = 0;
  std::vector<unsigned short> d_argmax_history, d_sfd_history;
  unsigned d_sync_recovery_counter = 0;
  std::vector<unsigned short> d_symbols;
  // stream framing: the samples [base, base + total()) of the framed stream are buf (carried over from
  // earlier calls) followed, during a work() call, by that call's input, which is never copied on the
  // host: only the GPU reads samples, the host keeps what a later call may still need; pos = read index
  std::vector<lora_c32> buf;
  const lora_c32 *cur_in = nullptr; size_t cur_n = 0;
  // device-resident input (lora_demod_core_work_device): cur_in is a device pointer; only the carry and the first
  // head_n input samples are staged (windows that start inside the carry end there), every other window is read
  // from cur_in where it lies
  bool cur_dev = false; size_t head_n = 0;
  size_t total() const { return buf.size() + cur_n; }
  uint64_t base = 0, pos = 0;
  // GPU result caches
  std::vector<Run> up_runs;      // launched by this core, bounded (kMaxRuns)
  size_t last_up = 0;            // index of the sweep that answered the last up-path lookup
  std::vector<Run> pre_runs;     // handed in through lora_demod_core_preload
  std::vector<Run> sfd_runs;     // one per detect-lattice phase that has needed a sweep, bounded (16 phases)
  unsigned sfd_misses[OVERLAP_FACTOR] = {0};
  uint64_t sfd_last_miss[OVERLAP_FACTOR] = {0};
  bool staged = false;
  // instrumentation
  lora_core_trace *trace = nullptr; size_t max_trace = 0;
  uint64_t n_calls = 0, n_windows = 0, n_launches = 0;
  std::string err;
};

namespace {

constexpr size_t kMaxSpec = 1u << 16, kReadSpec = 2048, kMaxRuns = 48;

int fail(lora_demod_core *c, int rc, const char *what) {
  c->err = std::string(what) + ": " + lora_hip_strerror(rc);
  if (rc == LORA_HIP_EHIP) c->err += std::string(" [") + lora_hip_last_hip_error(c->hip) + "]";
  return rc;
}

int ensure_staged(lora_demod_core *c) {
  if (c->staged) return LORA_HIP_OK;
  int rc;
  if (c->cur_dev) {
    // a window that starts inside the carry (< 4N samples) ends at most N samples into the input
    const size_t head_cap = (size_t)(DEMOD_HISTORY_DEPTH + 1) * c->d_num_symbols;
    c->head_n = c->cur_n < head_cap ? c->cur_n : head_cap;
    rc = lora_hip_demod_stage_iq2_dev(c->hip, c->buf.data(), c->buf.size(), c->cur_in, c->head_n);
  } else {
    rc = lora_hip_demod_stage_iq2(c->hip, c->buf.data(), c->buf.size(), c->cur_in, c->cur_n);
  }
  if (rc) return fail(c, rc, "stage_iq");
  c->staged = true;
  return LORA_HIP_OK;
}

int launch_run(lora_demod_core *c, Run &r, uint64_t start, uint64_t stride, size_t count, unsigned off0,
               unsigned off_step, unsigned kind) {
  int rc = ensure_staged(c);
  if (rc) return rc;
  r.start = start; r.stride = stride; r.off0 = off0; r.off_step = off_step; r.kind = kind;
  r.idx.assign(count, 0); r.val.assign(count, 0.f);
  const uint64_t s0 = start - c->base;
  if (!c->cur_dev || c->head_n == c->cur_n) {
    // everything the windows touch is in the staging buffer
    rc = lora_hip_demod_batch_strided(c->hip, nullptr, c->total(), s0, stride, count, off0, off_step,
                                      kind, r.idx.data(), r.val.data(), LORA_HIP_IQ_STAGED);
    c->n_launches++;
  } else {
    // device-resident input: the windows that start inside the carry come from the staged head, all others
    // from the caller's buffer in place
    const uint64_t carry = c->buf.size();
    size_t k0 = 0;
    if (s0 < carry) {
      const uint64_t k = (carry - s0 + stride - 1) / stride;
      k0 = k < count ? (size_t)k : count;
    }
    rc = LORA_HIP_OK;
    if (k0) {
      rc = lora_hip_demod_batch_strided(c->hip, nullptr, carry + c->head_n, s0, stride, k0, off0, off_step, kind,
                                        r.idx.data(), r.val.data(), LORA_HIP_IQ_STAGED);
      c->n_launches++;
    }
    if (!rc && count > k0) {
      const unsigned n = c->d_num_symbols;
      const unsigned off_k0 = (unsigned)((off0 + (uint64_t)k0 * off_step) & (n - 1));
      rc = lora_hip_demod_batch_strided(c->hip, c->cur_in, c->cur_n, s0 + (uint64_t)k0 * stride - carry, stride,
                                        count - k0, off_k0, off_step, kind, r.idx.data() + k0, r.val.data() + k0,
                                        LORA_HIP_IQ_ON_DEVICE);
      c->n_launches++;
    }
  }
  if (rc) { r.idx.clear(); r.val.clear(); return fail(c, rc, "batch_strided"); }
  c->n_windows += count;
  return LORA_HIP_OK;
}

// (argmax, max) of the up path at (pos, off): lib/demod_impl.cc:177-206
int get_up(lora_demod_core *c, uint64_t pos, unsigned off, unsigned short *idx, float *val) {
  const unsigned n = c->d_num_symbols;
  size_t k;
  // consecutive calls walk along one sweep: try the one that answered last before searching them all
  if (c->last_up < c->up_runs.size() && c->up_runs[c->last_up].find(pos, off, n, &k)) {
    *idx = c->up_runs[c->last_up].idx[k]; *val = c->up_runs[c->last_up].val[k];
    return LORA_HIP_OK;
  }
  for (size_t r = c->up_runs.size(); r-- > 0;) {
    if (c->up_runs[r].find(pos, off, n, &k)) { c->last_up = r; *idx = c->up_runs[r].idx[k]; *val = c->up_runs[r].val[k]; return LORA_HIP_OK; }
  }
  for (size_t r = c->pre_runs.size(); r-- > 0;) {
    if (c->pre_runs[r].find(pos, off, n, &k)) { *idx = c->pre_runs[r].idx[k]; *val = c->pre_runs[r].val[k]; return LORA_HIP_OK; }
  }
  // miss: drop batches that lie entirely behind the read index
  auto behind = [&](const Run &u) { return u.start + (uint64_t)u.idx.size() * u.stride <= c->pos; };
  for (size_t r = 0; r < c->up_runs.size();) {
    if (behind(c->up_runs[r])) c->up_runs.erase(c->up_runs.begin() + r);
    else r++;
  }
  for (size_t r = 0; r < c->pre_runs.size();) {
    if (behind(c->pre_runs[r])) c->pre_runs.erase(c->pre_runs.begin() + r);
    else r++;
  }
  if (c->up_runs.size() >= kMaxRuns) c->up_runs.erase(c->up_runs.begin());
  // Sweep this (lattice phase, offset) far ahead: a window costs the GPU nanoseconds, a launch and its
  // synchronisation tens of microseconds, and the same lattice tends to come back after later packets
  // (there are only 16 phases).  A header/payload lattice ends with its packet, so it gets a shorter sweep
  // than a preamble search.
  const uint64_t end = c->base + c->total();
  const size_t avail = (size_t)((end - pos - n) / n + 1);
  const bool reading = (c->d_state == LORA_S_READ_HEADER || c->d_state == LORA_S_READ_PAYLOAD);
  const size_t want = reading ? kReadSpec : kMaxSpec;
  c->up_runs.emplace_back();
  int rc = launch_run(c, c->up_runs.back(), pos, n, avail < want ? avail : want, off, 0, LORA_WIN_UP);
  if (rc) { c->up_runs.pop_back(); return rc; }
  c->last_up = c->up_runs.size() - 1;
  *idx = c->up_runs.back().idx[0]; *val = c->up_runs.back().val[0];
  return LORA_HIP_OK;
}

// argmax of one overlapped SFD window: lib/demod_impl.cc:297-322.
// The window at absolute position p of a sync attempt whose calls sit on the detect lattice of phase
// phi = call_pos mod N uses the chirp offset (p - phi) mod N (:297, d_offset = ol*N/16 and read indices move by N
// between the calls of one attempt): its result depends on (p, phi) only, not on which call or attempt asks.  One
// stride-N/16 sweep therefore serves every later attempt on the same lattice phase.  How far a sweep runs ahead
// follows the traffic: a launch and its synchronisation cost about 30 us, a window the GPU, the copy back and the
// host about N*F/1200 ns (more at high SF), so sweeping the gap to the next attempt on this phase pays when the
// gap -- taken from the last two misses of the phase -- is shorter than the windows one launch is worth.  Dense
// traffic (a packet every ~85 symbol times): 458 -> 92 launches per 386 packets; sparse traffic (the 1-hour
// capture of BASELINE configs[4], a packet per second): gaps are too long, every attempt keeps its own sweep of at
// most 10 calls x 16 windows (:283-292), as before.
int get_sfd(lora_demod_core *c, uint64_t call_pos, unsigned off, unsigned short *idx) {
  const unsigned n = c->d_num_symbols, step = n / OVERLAP_FACTOR;
  size_t k;
  for (size_t r = c->sfd_runs.size(); r-- > 0;) {
    if (c->sfd_runs[r].find(call_pos + off, off, n, &k)) { *idx = c->sfd_runs[r].idx[k]; return LORA_HIP_OK; }
  }
  // miss: drop sweeps that lie entirely behind the read index
  for (size_t r = 0; r < c->sfd_runs.size();) {
    const Run &u = c->sfd_runs[r];
    if (u.start + (uint64_t)u.idx.size() * u.stride <= c->pos) c->sfd_runs.erase(c->sfd_runs.begin() + r);
    else r++;
  }
  if (c->sfd_runs.size() >= 2 * OVERLAP_FACTOR) c->sfd_runs.erase(c->sfd_runs.begin());
  const uint64_t end = c->base + c->total();
  const uint64_t first = call_pos + off;
  const size_t avail = (size_t)((end - first - n) / step + 1);
  const unsigned phase = (unsigned)((call_pos % n) / step);
  const size_t one_attempt = (size_t)OVERLAP_FACTOR * 10 + OVERLAP_FACTOR;
  size_t want = one_attempt;
  if (c->sfd_misses[phase]++ > 0) {
    const double window_ns = (double)n * c->d_fft_size_factor / 1200.0 * (1.0 + 0.17 * ((double)c->d_sf - 8.0));
    const double worth = 30000.0 / (window_ns > 0.1 ? window_ns : 0.1);          // windows one launch is worth (15 us measured worse)
    const uint64_t gap = (first - c->sfd_last_miss[phase]) / step;               // windows since this phase's last miss
    if ((double)gap <= worth) {
      const uint64_t ahead = 8 * gap, cap = (uint64_t)1 << 17;
      want = (size_t)(ahead < one_attempt ? one_attempt : (ahead < cap ? ahead : cap));
    }
  }
  c->sfd_last_miss[phase] = first;
  c->sfd_runs.emplace_back();
  int rc = launch_run(c, c->sfd_runs.back(), first, step, avail < want ? avail : want, off, step, LORA_WIN_SFD);
  if (rc) { c->sfd_runs.pop_back(); return rc; }
  *idx = c->sfd_runs.back().idx[0];
  return LORA_HIP_OK;
}

// One general_work call at c->pos (lib/demod_impl.cc:149-450).  *consumed = num_consumed.
int general_work(lora_demod_core *c, lora_pdu_cb cb, void *user, unsigned *consumed) {
  const unsigned N = c->d_num_symbols, F = c->d_fft_size_factor;
  unsigned num_consumed = N;
  unsigned short max_index = 0;
  bool preamble_found = false, sfd_found = false;
  const int st_in = c->d_state;
  unsigned n_sfd = 0;

  // :177-206 up path; chirp offset is d_offset only while reading header/payload
  const unsigned off = (c->d_state == LORA_S_READ_HEADER || c->d_state == LORA_S_READ_PAYLOAD) ? c->d_offset : 0;
  float max_val = 0.f;
  int rc = get_up(c, c->pos, off, &max_index, &max_val);
  if (rc) return rc;
  const unsigned short up_idx = max_index;
  // argmax(..., update_squelch = true), :133-137
  c->d_power = max_val;
  c->d_squelched = (c->d_power > c->d_threshold) ? false : true;
  // :207-212
  c->d_argmax_history.insert(c->d_argmax_history.begin(), max_index);
  if (c->d_argmax_history.size() > REQUIRED_PREAMBLE_CHIRPS) c->d_argmax_history.pop_back();

  switch (c->d_state) {
    case LORA_S_RESET:                                              // :215-229
      c->d_overlaps = OVERLAP_DEFAULT;
      c->d_offset = 0;
      c->d_symbols.clear();
      c->d_argmax_history.clear();
      c->d_sfd_history.clear();
      c->d_sync_recovery_counter = 0;
      c->d_state = LORA_S_PREFILL;
      break;

    case LORA_S_PREFILL:                                            // :233-242
      if (c->d_argmax_history.size() >= REQUIRED_PREAMBLE_CHIRPS) c->d_state = LORA_S_DETECT_PREAMBLE;
      break;

    case LORA_S_DETECT_PREAMBLE:                                    // :247-273
      c->d_preamble_idx = c->d_argmax_history[0];
      preamble_found = true;
      for (int i = 1; i < REQUIRED_PREAMBLE_CHIRPS; i++)
        if (abs(int(c->d_preamble_idx) - int(c->d_argmax_history[i])) > LORA_PREAMBLE_TOLERANCE) preamble_found = false;
      if (preamble_found && !c->d_squelched) c->d_state = LORA_S_SFD_SYNC;
      break;

    case LORA_S_SFD_SYNC:                                           // :279-359
      c->d_overlaps = OVERLAP_FACTOR;
      if (c->d_sync_recovery_counter++ > DEMOD_SYNC_RECOVERY_COUNT) {
        c->d_state = LORA_S_RESET;
        c->d_overlaps = OVERLAP_DEFAULT;
      }
      for (int ol = 0; ol < (int)c->d_overlaps; ol++) {
        c->d_offset = ((ol * N) / c->d_overlaps) % N;
        if ((rc = get_sfd(c, c->pos, c->d_offset, &max_index))) return rc;
        n_sfd++;
        c->d_sfd_history.insert(c->d_sfd_history.begin(), max_index);
        if (c->d_sfd_history.size() > REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR) {
          c->d_sfd_history.pop_back();
          sfd_found = true;
          c->d_sfd_idx = c->d_sfd_history[0];
          for (int i = 1; i < REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR; i++)
            if (abs(int(c->d_sfd_idx) - int(c->d_sfd_history[i])) > int(F) * LORA_SFD_TOLERANCE) sfd_found = false;
          if (sfd_found) {
            num_consumed = (ol * N) / c->d_overlaps + 5 * N / 4;   // skip last quarter chirp
            c->d_offset = (c->d_offset + (N / 4)) % N;
            c->d_state = LORA_S_READ_HEADER;
            c->d_overlaps = OVERLAP_DEFAULT;
            break;
          }
        }
      }
      break;

    case LORA_S_READ_HEADER:                                        // :363-387
      if (c->d_squelched) c->d_state = LORA_S_OUT;
      else if (c->d_symbols.size() == 7) c->d_state = LORA_S_READ_PAYLOAD;
      c->d_symbols.push_back((unsigned short)(((N + (c->d_argmax_history[0] / F) - (c->d_preamble_idx / F)) % N) / 4));
      break;

    case LORA_S_READ_PAYLOAD:                                       // :391-413
      if (c->d_squelched) c->d_state = LORA_S_OUT;
      if (c->d_ldr)
        c->d_symbols.push_back((unsigned short)(((N + (c->d_argmax_history[0] / F) - (c->d_preamble_idx / F)) % N) / 4));
      else
        c->d_symbols.push_back((unsigned short)((N + (c->d_argmax_history[0] / F) - (c->d_preamble_idx / F)) % N));
      break;

    case LORA_S_OUT:                                                // :418-431
      if (cb) cb(user, c->d_symbols.data(), c->d_symbols.size());
      c->d_state = LORA_S_RESET;
      break;

    default:
      break;
  }

  if (c->trace && c->n_calls < c->max_trace) {
    lora_core_trace &t = c->trace[c->n_calls];
    t.pos = c->pos; t.state = (uint16_t)st_in; t.chirp_off = (uint16_t)off; t.idx = up_idx;
    t.consumed = (uint16_t)num_consumed; t.val = max_val; t.n_sfd = n_sfd;
  }
  c->n_calls++;
  *consumed = num_consumed & 0xFFFFu;     // num_consumed is an unsigned short in the reference (:157)
  return LORA_HIP_OK;
}

}  // namespace

extern "C" {

int lora_demod_core_create(lora_demod_core_t **out, int device, unsigned sf, int ldr, float beta, unsigned fft_factor) {
  if (!out) return LORA_HIP_EINVAL;
  *out = nullptr;
  lora_demod_core *c = new (std::nothrow) lora_demod_core();
  if (!c) return LORA_HIP_ENOMEM;
  int rc = lora_hip_demod_create(&c->hip, device, sf, fft_factor, beta, nullptr, nullptr, nullptr);
  if (rc) { delete c; return rc; }
  c->d_sf = sf; c->d_ldr = ldr != 0; c->d_beta = beta; c->d_fft_size_factor = fft_factor;
  c->d_num_symbols = 1u << sf;                                      // :79
  c->d_fft_size = c->d_fft_size_factor * c->d_num_symbols;          // :80
  // set_history(DEMOD_HISTORY_DEPTH * N), :103: the scheduler presents history()-1 zeros first
  c->buf.assign((size_t)DEMOD_HISTORY_DEPTH * c->d_num_symbols - 1, lora_c32{0.f, 0.f});
  *out = c;
  return LORA_HIP_OK;
}

int lora_demod_core_reset(lora_demod_core_t *c) {
  if (!c) return LORA_HIP_EINVAL;
  // back to the state of a freshly made block (:77-88, :103): same tables and GPU buffers, new stream
  c->d_state = LORA_S_RESET;
  c->d_overlaps = OVERLAP_DEFAULT; c->d_offset = 0;
  c->d_power = .000000001f; c->d_squelched = false;
  c->d_preamble_idx = 0; c->d_sfd_idx = 0;
  c->d_argmax_history.clear(); c->d_sfd_history.clear(); c->d_symbols.clear();
  c->d_sync_recovery_counter = 0;
  c->buf.assign((size_t)DEMOD_HISTORY_DEPTH * c->d_num_symbols - 1, lora_c32{0.f, 0.f});
  c->cur_in = nullptr; c->cur_n = 0;
  c->base = 0; c->pos = 0;
  c->up_runs.clear(); c->pre_runs.clear(); c->sfd_runs.clear();
  for (unsigned &m : c->sfd_misses) m = 0;
  for (uint64_t &m : c->sfd_last_miss) m = 0;
  c->staged = false;
  c->n_calls = 0; c->n_windows = 0; c->n_launches = 0;
  c->err.clear();
  return LORA_HIP_OK;
}

void lora_demod_core_destroy(lora_demod_core_t *c) {
  if (!c) return;
  lora_hip_demod_destroy(c->hip);
  delete c;
}

static int core_work(lora_demod_core_t *c, const lora_c32 *in, size_t n_in, bool on_device, lora_pdu_cb cb, void *user);

int lora_demod_core_work(lora_demod_core_t *c, const lora_c32 *in, size_t n_in, lora_pdu_cb cb, void *user) {
  return core_work(c, in, n_in, false, cb, user);
}

int lora_demod_core_work_device(lora_demod_core_t *c, const lora_c32 *in_device, size_t n_in, lora_pdu_cb cb, void *user) {
  return core_work(c, in_device, n_in, true, cb, user);
}

int lora_demod_core_set_stream(lora_demod_core_t *c, void *hip_stream) {
  if (!c) return LORA_HIP_EINVAL;
  return lora_hip_demod_set_stream(c->hip, hip_stream);
}

static int core_work(lora_demod_core_t *c, const lora_c32 *in, size_t n_in, bool on_device, lora_pdu_cb cb, void *user) {
  if (!c || (!in && n_in)) return LORA_HIP_EINVAL;
  c->cur_in = in; c->cur_n = n_in; c->cur_dev = on_device && n_in; c->head_n = 0;
  if (n_in) c->staged = false;
  const uint64_t need = (uint64_t)c->d_num_symbols + (uint64_t)DEMOD_HISTORY_DEPTH * c->d_num_symbols - 1;
  int rc = LORA_HIP_OK;
  while (c->base + c->total() >= c->pos + need) {
    unsigned consumed = 0;
    rc = general_work(c, cb, user, &consumed);
    if (rc) break;
    c->pos += consumed;                                             // consume_each, :443
  }
  // keep what can still be read, [pos, end), as the next call's carry; cached runs keep absolute positions
  try {
    const uint64_t carry_end = c->base + c->buf.size();
    if (n_in == 0) {
      c->buf.erase(c->buf.begin(), c->buf.begin() + (size_t)(c->pos - c->base));
    } else if (c->pos >= carry_end) {
      const size_t skip = (size_t)(c->pos - carry_end);             // pos <= end: a call never consumes past need
      const size_t from = skip < n_in ? skip : n_in;
      if (c->cur_dev) {                                             // the unconsumed tail (< 4N samples) comes back from HBM
        c->buf.resize(n_in - from);
        const int rf = lora_hip_demod_fetch(c->hip, in + from, n_in - from, c->buf.data());
        if (rf) rc = rc ? rc : fail(c, rf, "fetch");
      } else {
        c->buf.assign(in + from, in + n_in);
      }
    } else {
      c->buf.erase(c->buf.begin(), c->buf.begin() + (size_t)(c->pos - c->base));
      if (c->cur_dev) {                                             // fewer than 4N new samples: all of them are kept
        const size_t old = c->buf.size();
        c->buf.resize(old + n_in);
        const int rf = lora_hip_demod_fetch(c->hip, in, n_in, c->buf.data() + old);
        if (rf) rc = rc ? rc : fail(c, rf, "fetch");
      } else {
        c->buf.insert(c->buf.end(), in, in + n_in);
      }
    }
  } catch (const std::bad_alloc &) { rc = rc ? rc : LORA_HIP_ENOMEM; }
  c->base = c->pos > c->base ? c->pos : c->base;
  c->cur_in = nullptr; c->cur_n = 0; c->cur_dev = false; c->head_n = 0;
  c->staged = false;
  if (rc) return rc;
  return LORA_HIP_OK;
}

int lora_demod_core_preload(lora_demod_core_t *c, uint64_t start, size_t count, const uint16_t *idx, const float *val) {
  if (!c || (count && (!idx || !val))) return LORA_HIP_EINVAL;
  if (count == 0) return LORA_HIP_OK;
  if (start % (c->d_num_symbols / OVERLAP_FACTOR)) return LORA_HIP_EINVAL;      // read indices stay on the N/16 lattice
  try {
    c->pre_runs.emplace_back();
    Run &r = c->pre_runs.back();
    r.start = start; r.stride = c->d_num_symbols; r.off0 = 0; r.off_step = 0; r.kind = LORA_WIN_UP;
    r.idx.assign(idx, idx + count); r.val.assign(val, val + count);
  } catch (const std::bad_alloc &) { return LORA_HIP_ENOMEM; }
  return LORA_HIP_OK;
}

int lora_demod_core_set_trace(lora_demod_core_t *c, lora_core_trace *rows, size_t max_rows) {
  if (!c) return LORA_HIP_EINVAL;
  c->trace = rows; c->max_trace = rows ? max_rows : 0;
  return LORA_HIP_OK;
}

uint64_t lora_demod_core_calls(const lora_demod_core_t *c) { return c ? c->n_calls : 0; }
uint64_t lora_demod_core_windows(const lora_demod_core_t *c) { return c ? c->n_windows : 0; }
uint64_t lora_demod_core_launches(const lora_demod_core_t *c) { return c ? c->n_launches : 0; }
int lora_demod_core_state(const lora_demod_core_t *c) { return c ? c->d_state : -1; }
const char *lora_demod_core_last_error(const lora_demod_core_t *c) { return c ? c->err.c_str() : ""; }

}  // extern "C"
