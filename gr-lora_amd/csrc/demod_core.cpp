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

#include <chrono>
#include <immintrin.h>
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

// one planned sweep: its results lie at [at, at + count) of the plan's page-locked result arrays
struct PRun {
  uint64_t start = 0; uint32_t stride = 1, count = 0;      // stride: N or N/16, a power of two
  unsigned off0 = 0, off_step = 0;
  size_t at = 0;
  bool find(uint64_t pos, unsigned off, unsigned n, size_t *k, size_t *rem = nullptr) const {
    if (pos < start) return false;
    const uint64_t d = pos - start;
    if (d & (stride - 1)) return false;
    const uint64_t q = d >> __builtin_ctz(stride);
    if (q >= count) return false;
    if (((off0 + q * off_step) & (n - 1)) != off) return false;
    *k = at + (size_t)q;
    if (rem) *rem = count - (size_t)q;        // windows of this sweep from k on
    return true;
  }
};

// page-locked host array that only grows
template <typename T>
struct Pin {
  T *p = nullptr; size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return LORA_HIP_OK;
    lora_hip_host_free(p); p = nullptr; cap = 0;
    const size_t nc = n + n / 4 + 1024;
    void *q = nullptr;
    const int rc = lora_hip_host_alloc(&q, nc * sizeof(T));
    if (rc) return rc;
    p = static_cast<T *>(q); cap = nc;
    return LORA_HIP_OK;
  }
  Pin() = default;
  Pin(const Pin &) = delete;
  Pin &operator=(const Pin &) = delete;
  ~Pin() { lora_hip_host_free(p); }
};

// What one work() call (or one stretch of a very long one) evaluates ahead of the state machine, in a fixed number of
// launches whatever the number of packets:
//   A    every up-path window with chirp offset 0 on the N/16 lattice, all 16 phases (the read index never leaves that
//        lattice: num_consumed is N or ol*N/16 + 5N/4, :314), one strided sweep;
//   SFD  behind every lattice position at which the preamble rule (:247-273) holds, on its phase, the at most
//        9*16 + 1 overlapped windows one sync attempt can evaluate (:283-292), all of them in one launch;
//   READ the header/payload windows of the chain of packets a dry run of the state machine (same code, squelch taken
//        from A: a chirp offset only rotates the spectrum) predicts, one launch.
// The state machine then runs over these results; whatever the plan did not foresee is evaluated on demand as before
// and the READ plan is redone from that point.  Results are exact either way: a plan only decides WHICH windows are
// evaluated early, never what a window's result is.
struct Plan {
  bool active = false;
  uint64_t a_start = 0; size_t a_count = 0;          // window j of A starts at a_start + j*N/16
  uint64_t walk_end = 0;                             // the state machine runs on this plan while its read index is <= walk_end
  Pin<uint16_t> a_idx; Pin<float> a_val;
  std::vector<PRun> sfd[OVERLAP_FACTOR]; size_t sfd_cur[OVERLAP_FACTOR] = {0};
  Pin<uint16_t> s_idx;
  std::vector<PRun> rd; size_t rd_cur = 0;
  Pin<uint16_t> r_idx; Pin<float> r_val;
  unsigned replans = 0;
};

// a header/payload call the first pass answered from a prediction: where its exact result goes once it is known
struct Patch { uint64_t pos; uint32_t row, flat; uint16_t off, pre; uint8_t div4, sq; };
// the read index, call count and PDU count at a call that enters S_RESET: everything else is cleared by that call (:215-229)
struct Snap { uint64_t pos, calls; size_t pdus; };

}  // namespace

// the reference's std::vector histories (newest first, :207-212, :318-320) as fixed arrays: they never hold more than
// REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR + 1 entries
template <unsigned CAP>
struct Hist {
  // entries live at buf[head .. head + n): a push writes in front of them; when the front is reached they move back to
  // the end of the array, once per kSlack pushes
  static constexpr unsigned kSlack = 224;
  unsigned short buf[CAP + 1 + kSlack]; unsigned head = kSlack, n = 0;
  size_t size() const { return n; }
  void clear() { n = 0; head = kSlack; }
  void push_front(unsigned short x) {
    if (head == 0) { memmove(buf + kSlack, buf, n * sizeof(unsigned short)); head = kSlack; }
    buf[--head] = x; n++;
  }
  void pop_back() { n--; }
  unsigned short operator[](size_t i) const { return buf[head + i]; }
  const unsigned short *data() const { return buf + head; }
};

// x / fft_size_factor without a division when the factor is a power of two (it almost always is; N always is, so `% N`
// is a mask): the state machine makes three such divisions per header/payload call and two per SFD window
struct DivF {
  unsigned f = 1, sh = 0; bool p2 = true;
  void set(unsigned factor) { f = factor; p2 = factor && !(factor & (factor - 1)); sh = p2 ? (unsigned)__builtin_ctz(factor) : 0; }
  unsigned operator()(unsigned x) const { return p2 ? x >> sh : x / f; }
};

// everything general_work reads and writes between calls (demod_impl.h:43-70) plus the read index: a dry run saves and
// restores exactly this
struct WalkState {
  int d_state = LORA_S_RESET;
  unsigned d_overlaps = OVERLAP_DEFAULT, d_offset = 0;
  float d_power = .000000001f, d_threshold = 0.005f;      // :87-88
  bool d_squelched = false;
  unsigned d_preamble_idx = 0, d_sfd_idx = 0;
  Hist<REQUIRED_PREAMBLE_CHIRPS> d_argmax_history; Hist<REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR> d_sfd_history;
  unsigned d_sync_recovery_counter = 0;
  unsigned sfd_agree = 0;       // bookkeeping of the SFD test (general_work), cleared with d_sfd_history
  std::vector<unsigned short> d_symbols;
  uint64_t pos = 0;
  uint64_t n_pdus = 0;          // PDUs emitted on this stream so far
};

constexpr unsigned kPlans = 4;
static_assert(2 * kPlans <= LORA_HIP_MARKS, "one mark per plan for its lattice sweep, one for its SFD stage");

struct lora_demod_core : WalkState {
  // configuration (demod_impl.h:44-49)
  unsigned d_sf = 0; bool d_ldr = false; float d_beta = 0; unsigned d_fft_size_factor = 0;
  unsigned d_num_symbols = 0, d_fft_size = 0;
  DivF div_f;
  lora_hip_demod_t *hip = nullptr;
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
  uint64_t base = 0;
  // planned evaluation (Plan above): mode 0 off, 1 when the input is long enough to pay for it, 2 always
  // Three plans: a long input is planned in stretches, and while the state machine walks one stretch (`planp`) the GPU
  // already evaluates the SFD stage of the next and the lattice sweep of the one after (pipeline_run)
  Plan plans[kPlans]; Plan *planp = &plans[0]; int plan_mode = 1;
  Plan &cur() { return *planp; }
  uint64_t dens_pdus = 0, dens_syms = 0;    // traffic of the last calls (halved every 4096 symbols): PDUs emitted, symbols consumed
  unsigned walk_lane = 0;        // LORA_HIP_LANE1 while stretches are in flight: what the walk launches must not queue behind them
  // Lattice positions per stretch at least (LORA_CORE_STRETCH: developer knob).  A stretch costs its launches, a READ round
  // trip that nothing hides (~0.06 ms, more when the READ launch has to wait for a running sweep to let go of the CUs) and the
  // state machine's wait for its SFD stage; measured on 524 k positions (8.4 M samples at SF8, 386 packets,
  // profiles/r03_pipeline.txt, end of round 3): 1 stretch 1.40 ms, 2 stretches 1.16, 3 stretches 1.19, 4 stretches 1.24, 7 stretches 1.46
  size_t stretch_min = 180000;
  bool shadow = false;           // dry run: no launches, no PDUs, misses are predicted and recorded in `needs`
  std::vector<PRun> needs;
  std::vector<uint32_t> fire;      // plan_sfd: the lattice positions at which the preamble rule holds
  // first pass of a plan (plan_walk): the dry run keeps its trace rows and PDUs, to be completed from the READ stage
  bool deferred = false, pred = false;
  std::vector<Patch> patches; std::vector<Snap> snaps;
  std::vector<uint16_t> pdu_syms; std::vector<size_t> pdu_off;
  // every maximal run of consecutive S_DETECT_PREAMBLE calls (lora_demod_core_idle_log); calls0 = call number of its first
  std::vector<lora_core_idle> idle; std::vector<uint64_t> idle_calls0;
  bool collect = false;                                        // lora_demod_core_set_collect
  std::vector<uint16_t> col_syms; std::vector<uint64_t> col_off;
  long fault_at = -1; uint64_t n_pred = 0, n_rollbacks = 0;    // LORA_CORE_FAULT: test hook, mispredict that window
  unsigned read_flags = LORA_HIP_OUT_DIRECT;      // LORA_CORE_READ_DIRECT=0: the READ stage's results through the copies (developer knob)
  bool no_fast = false;          // LORA_CORE_NO_FAST=1: every call of the first pass through general_work (tests compare the two)
  uint64_t n_plans = 0, n_replans = 0, n_demand = 0;
  // LORA_CORE_TIMING=1: seconds per stage of the last work() call on stderr (lattice sweep, SFD scan, SFD launch, dry
  // run, READ launch, state machine)
  bool timing = false; double t_stage[6] = {0, 0, 0, 0, 0, 0}, t_cb = 0, t_wait = 0, t_real = 0;
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

// one strided sweep into caller-given result arrays (val may be NULL with LORA_HIP_NO_VAL in xflags)
int launch_strided(lora_demod_core *c, uint64_t start, uint64_t stride, size_t count, unsigned off0, unsigned off_step,
                   unsigned kind, uint16_t *idx, float *val, unsigned xflags) {
  int rc = ensure_staged(c);
  if (rc) return rc;
  const uint64_t s0 = start - c->base;
  if (!c->cur_dev || c->head_n == c->cur_n) {
    // everything the windows touch is in the staging buffer
    rc = lora_hip_demod_batch_strided(c->hip, nullptr, c->total(), s0, stride, count, off0, off_step,
                                      kind, idx, val, LORA_HIP_IQ_STAGED | xflags);
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
                                        idx, val, LORA_HIP_IQ_STAGED | xflags);
      c->n_launches++;
    }
    if (!rc && count > k0) {
      const unsigned n = c->d_num_symbols;
      const unsigned off_k0 = (unsigned)((off0 + (uint64_t)k0 * off_step) & (n - 1));
      rc = lora_hip_demod_batch_strided(c->hip, c->cur_in, c->cur_n, s0 + (uint64_t)k0 * stride - carry, stride,
                                        count - k0, off_k0, off_step, kind, idx + k0, val ? val + k0 : nullptr,
                                        LORA_HIP_IQ_ON_DEVICE | xflags);
      c->n_launches++;
    }
  }
  if (rc) return fail(c, rc, "batch_strided");
  c->n_windows += count;
  return LORA_HIP_OK;
}

int launch_run(lora_demod_core *c, Run &r, uint64_t start, uint64_t stride, size_t count, unsigned off0,
               unsigned off_step, unsigned kind) {
  r.start = start; r.stride = stride; r.off0 = off0; r.off_step = off_step; r.kind = kind;
  r.idx.assign(count, 0); r.val.assign(count, 0.f);
  const int rc = launch_strided(c, start, stride, count, off0, off_step, kind, r.idx.data(), r.val.data(), c->walk_lane);
  if (rc) { r.idx.clear(); r.val.clear(); }
  return rc;
}

// ---- planned evaluation (struct Plan) ----------------------------------------------------------------------------
// one finished PDU (:420-422): to the callback, and into the block's own list when collecting
inline void deliver(lora_demod_core *c, lora_pdu_cb cb, void *user, const uint16_t *syms, size_t n) {
  if (c->collect) {
    c->col_syms.insert(c->col_syms.end(), syms, syms + n);
    c->col_off.push_back(c->col_syms.size());
  }
  if (cb) cb(user, syms, n);
}

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int kShadowStop = 1;                    // dry run met something it cannot predict: stop it there
// plan_mode 1: shorter inputs are evaluated on demand.  A plan costs five launches, three synchronisations and 16 times the
// lattice windows; on demand a packet costs two to four launches.  Measured on configs[3] (one packet per 280-symbol
// capture, four blocks in flight): 0.19 s planned against 0.12 s on demand, so the plan starts where several packets fit.
constexpr size_t kPlanMinSymbols = 1024;
// ... unless the stream has been dense lately: with a packet per kDenseSymbols symbols or more over the last calls a plan
// repays itself from kPlanMinDense symbols on (SF8 in 256-symbol chunks of back-to-back packets: 162 us planned against 275 us
// on demand per call, profiles/r03_stream.txt; one packet per 276-symbol capture, configs[3]: 0.15 s on demand against 0.22)
constexpr size_t kPlanMinDense = 192, kDenseSymbols = 160;
constexpr size_t kPlanMaxSfd = (size_t)1 << 23;   // SFD windows per plan
constexpr unsigned kMaxReplans = 16;
constexpr unsigned kSfdSpan = 9 * OVERLAP_FACTOR + 1;   // :283-292: nine full calls, then the call that gives up evaluates one

inline bool split_input(const lora_demod_core *c) { return c->cur_dev && c->head_n != c->cur_n; }

// append a planned sweep; with a device-resident input it is cut where its windows stop starting inside the carry
void push_planned(lora_demod_core *c, std::vector<PRun> &list, PRun r) {
  if (split_input(c)) {
    const uint64_t carry = c->buf.size(), s0 = r.start - c->base;
    if (s0 < carry) {
      const uint64_t k = (carry - s0 + r.stride - 1) / r.stride;
      if (k < r.count) {
        PRun a = r; a.count = (uint32_t)k;
        list.push_back(a);
        r.start += k * r.stride; r.count -= (uint32_t)k;
        r.off0 = (unsigned)((r.off0 + k * r.off_step) & (c->d_num_symbols - 1));
      }
    }
  }
  list.push_back(r);
}

// all sweeps of `runs` in one launch (two with a device-resident input when some windows start inside the carry)
int launch_planned(lora_demod_core *c, std::vector<PRun *> &runs, unsigned kind, Pin<uint16_t> &idx, Pin<float> *val,
                   unsigned xflags = 0) {
  if (runs.empty()) return LORA_HIP_OK;
  int rc = ensure_staged(c);
  if (rc) return rc;
  const bool split = split_input(c);
  const uint64_t carry = c->buf.size();
  std::vector<lora_win_run> early, late;
  size_t n_early = 0, n_late = 0;
  auto mk = [&](const PRun &r, uint64_t pos0) {
    lora_win_run u; memset(&u, 0, sizeof u);
    u.pos0 = pos0; u.stride = r.stride; u.count = r.count; u.chirp_off = (uint16_t)r.off0; u.chirp_off_step = (uint16_t)r.off_step;
    u.kind = (uint8_t)kind;
    return u;
  };
  for (PRun *r : runs) {
    const uint64_t s0 = r->start - c->base;
    if (!split || s0 < carry) { r->at = n_early; n_early += r->count; early.push_back(mk(*r, s0)); }
  }
  if (split)
    for (PRun *r : runs) {
      const uint64_t s0 = r->start - c->base;
      if (s0 >= carry) { r->at = n_early + n_late; n_late += r->count; late.push_back(mk(*r, s0 - carry)); }
    }
  const size_t total = n_early + n_late;
  if ((rc = idx.ensure(total))) return fail(c, rc, "host_alloc");
  if (val && (rc = val->ensure(total))) return fail(c, rc, "host_alloc");
  const unsigned flags = LORA_HIP_OUT_PINNED | (val ? 0u : (unsigned)LORA_HIP_NO_VAL) | xflags;
  if (n_early) {
    rc = lora_hip_demod_batch_runs(c->hip, nullptr, split ? carry + c->head_n : c->total(), early.data(), early.size(),
                                   idx.p, val ? val->p : nullptr, flags | LORA_HIP_IQ_STAGED);
    c->n_launches += 2;
    if (rc) return fail(c, rc, "batch_runs");
  }
  if (n_late) {
    rc = lora_hip_demod_batch_runs(c->hip, c->cur_in, c->cur_n, late.data(), late.size(), idx.p + n_early,
                                   val ? val->p + n_early : nullptr, flags | LORA_HIP_IQ_ON_DEVICE);
    c->n_launches += 2;
    if (rc) return fail(c, rc, "batch_runs");
  }
  c->n_windows += total;
  return LORA_HIP_OK;
}

inline bool plan_a_find(const Plan &P, uint64_t pos, unsigned step, size_t *k) {
  if (pos < P.a_start) return false;
  const uint64_t d = pos - P.a_start;
  if (d & (step - 1)) return false;
  const uint64_t q = d >> __builtin_ctz(step);
  if (q >= P.a_count) return false;
  *k = (size_t)q;
  return true;
}

// sorted sweeps, read indices only grow: a cursor finds the sweep
inline bool planned_find(const std::vector<PRun> &L, size_t &cur, uint64_t pos, unsigned off, unsigned n, size_t *k,
                         size_t *rem = nullptr) {
  if (L.empty()) return false;
  if (cur >= L.size()) cur = L.size() - 1;
  while (cur + 1 < L.size() && L[cur + 1].start <= pos) cur++;
  for (size_t b = 0; b < 3 && b <= cur; b++)
    if (L[cur - b].find(pos, off, n, k, rem)) return true;
  return false;
}

int general_work(lora_demod_core *c, lora_pdu_cb cb, void *user, unsigned *consumed);

// The preamble rule (:247-273) at every lattice position j >= 48 of a sweep: the bins at j-16, j-32, j-48 (the three older
// windows of j's phase) within LORA_PREAMBLE_TOLERANCE of the bin at j, and the window at j not squelched (val > thr).  Writes
// the positions at which it holds, in order, and returns their number.  |a - b| <= 1  <=>  (u16)(a - b + 1) <= 2 for bins
// below 65535.  (A capture with exact zeros between its packets has the same bin everywhere: the squelch has to be part of
// the vector test, or every position of an hour of silence is a candidate.)
static_assert(LORA_PREAMBLE_TOLERANCE == 1, "the wrap-around compare below is written for a tolerance of 1");
static_assert(OVERLAP_FACTOR == 16 && REQUIRED_PREAMBLE_CHIRPS == 4, "rule_hits looks 16, 32 and 48 positions back");
inline bool rule_at(const uint16_t *ai, const float *av, float thr, size_t j) {
  const unsigned short i0 = ai[j];
  return (((unsigned short)(i0 - ai[j - 16] + 1) <= 2) & ((unsigned short)(i0 - ai[j - 32] + 1) <= 2) & ((unsigned short)(i0 - ai[j - 48] + 1) <= 2)) &&
         av[j] > thr;
}
size_t rule_hits_sse2(const uint16_t *ai, const float *av, float thr, size_t count, uint32_t *hits) {
  size_t nh = 0, j = 48;
  const __m128i one = _mm_set1_epi16(1), two = _mm_set1_epi16(2), zero = _mm_setzero_si128();
  for (; j + 8 <= count; j += 8) {
    const __m128i i1 = _mm_add_epi16(_mm_loadu_si128((const __m128i *)(ai + j)), one);
    const __m128i d1 = _mm_subs_epu16(_mm_sub_epi16(i1, _mm_loadu_si128((const __m128i *)(ai + j - 16))), two);
    const __m128i d2 = _mm_subs_epu16(_mm_sub_epi16(i1, _mm_loadu_si128((const __m128i *)(ai + j - 32))), two);
    const __m128i d3 = _mm_subs_epu16(_mm_sub_epi16(i1, _mm_loadu_si128((const __m128i *)(ai + j - 48))), two);
    unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_or_si128(_mm_or_si128(d1, d2), d3), zero));
    while (m) {
      const unsigned b = (unsigned)__builtin_ctz(m);
      if (av[j + b / 2] > thr) hits[nh++] = (uint32_t)(j + b / 2);
      m &= ~(3u << b);
    }
  }
  for (; j < count; j++) if (rule_at(ai, av, thr, j)) hits[nh++] = (uint32_t)j;
  return nh;
}
__attribute__((target("avx2,bmi2"))) size_t rule_hits_avx2(const uint16_t *ai, const float *av, float thr, size_t count, uint32_t *hits) {
  size_t nh = 0, j = 48;
  const __m256i one = _mm256_set1_epi16(1), two = _mm256_set1_epi16(2), zero = _mm256_setzero_si256();
  const __m256 vthr = _mm256_set1_ps(thr);
  __m256i b1, b2, b3;
  if (count >= 64) {
    b3 = _mm256_loadu_si256((const __m256i *)(ai)); b2 = _mm256_loadu_si256((const __m256i *)(ai + 16)); b1 = _mm256_loadu_si256((const __m256i *)(ai + 32));
    for (; j + 16 <= count; j += 16) {                  // the three older vectors are the last three loaded
      const __m256i i0 = _mm256_loadu_si256((const __m256i *)(ai + j));
      const __m256i i1 = _mm256_add_epi16(i0, one);
      const __m256i d1 = _mm256_subs_epu16(_mm256_sub_epi16(i1, b1), two);
      const __m256i d2 = _mm256_subs_epu16(_mm256_sub_epi16(i1, b2), two);
      const __m256i d3 = _mm256_subs_epu16(_mm256_sub_epi16(i1, b3), two);
      unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi16(_mm256_or_si256(_mm256_or_si256(d1, d2), d3), zero));
      b3 = b2; b2 = b1; b1 = i0;
      if (!m) continue;
      m = _pext_u32(m, 0x55555555u);                    // one bit per position
      m &= (unsigned)_mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(av + j), vthr, _CMP_GT_OQ)) |
           (unsigned)_mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(av + j + 8), vthr, _CMP_GT_OQ)) << 8;
      while (m) { const unsigned b = (unsigned)__builtin_ctz(m); hits[nh++] = (uint32_t)(j + b); m &= m - 1; }
    }
  }
  for (; j < count; j++) if (rule_at(ai, av, thr, j)) hits[nh++] = (uint32_t)j;
  return nh;
}

// SFD stage: behind every lattice position of A at which the preamble rule holds (:247-273: the newest four windows of
// one phase within +-1 of the newest, not squelched), the windows one sync attempt entered there can evaluate
int plan_sfd(lora_demod_core *c, Plan &P, unsigned xflags = 0) {
  const double t0 = now_s();
  const unsigned n = c->d_num_symbols, step = n / OVERLAP_FACTOR;
  const uint16_t *ai = P.a_idx.p; const float *av = P.a_val.p;
  const uint64_t end = c->base + c->total();
  const uint64_t lat0 = P.a_start / step;
  const uint64_t jmax = (end - n - P.a_start) / step;      // last lattice index at which a window still fits
  const float thr = c->d_threshold;
  size_t total = 0;
  const size_t lag = OVERLAP_FACTOR;
  // the rule for every lattice position at once (16-bit lanes, 8 or 16 positions per operation), then the few hits
  c->fire.resize(P.a_count + 1);
  uint32_t *fire = c->fire.data();
  static const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
  const size_t n_fire = have_avx2 ? rule_hits_avx2(ai, av, thr, P.a_count, fire) : rule_hits_sse2(ai, av, thr, P.a_count, fire);
  const double t_rule = now_s();
  uint64_t back_lo[OVERLAP_FACTOR] = {0};                  // per phase: lattice index at which its last sweep starts
  for (unsigned ph = 0; ph < OVERLAP_FACTOR; ph++) P.sfd[ph].clear();
  for (size_t f = 0; f < n_fire && total < kPlanMaxSfd; f++) {
    const size_t j = fire[f];                           // (not squelched, :268: rule_hits has looked)
    const uint64_t lo = j + lag;                           // the sync attempt's first call is the next one on this phase
    uint64_t hi = lo + kSfdSpan;
    if (hi > jmax + 1) hi = jmax + 1;
    if (lo >= hi) continue;
    const unsigned ph = (unsigned)((lat0 + j) % OVERLAP_FACTOR);
    std::vector<PRun> &L = P.sfd[ph];
    if (!L.empty()) {
      PRun &b = L.back();
      const uint64_t blo = back_lo[ph];
      if (lo <= blo + b.count) {                           // overlaps or touches the last sweep of this phase
        // An attempt entered later in the same preamble locks at the same place or not at all: the sweep grows by one
        // call beyond what its first attempt can use, not by a full attempt per position (the rest, if ever asked
        // for, is evaluated on demand)
        const uint64_t cap = blo + kSfdSpan + OVERLAP_FACTOR;
        if (hi > cap && lo < cap) hi = cap;
        if (lo < cap) {
          if (hi > blo + b.count) { total += (size_t)(hi - blo - b.count); b.count = (uint32_t)(hi - blo); }
          continue;
        }
      }
    }
    PRun r;
    r.start = P.a_start + lo * step; r.stride = step; r.count = (uint32_t)(hi - lo);
    r.off0 = (unsigned)(((r.start % n) + n - ph * step) % n); r.off_step = step;
    L.push_back(r); back_lo[ph] = lo;
    total += r.count;
  }
  const double t_hits = now_s();
  std::vector<PRun *> all;
  for (unsigned ph = 0; ph < OVERLAP_FACTOR; ph++) {
    if (split_input(c)) {                                  // cut the sweeps that straddle the end of the carry
      std::vector<PRun> cut;
      for (const PRun &r : P.sfd[ph]) push_planned(c, cut, r);
      P.sfd[ph].swap(cut);
    }
    for (PRun &r : P.sfd[ph]) all.push_back(&r);
    P.sfd_cur[ph] = 0;
  }
  const double t1 = now_s();
  const int rc = launch_planned(c, all, LORA_WIN_SFD, P.s_idx, nullptr, xflags);
  c->t_stage[1] += t1 - t0; c->t_stage[2] += now_s() - t1;
  if (c->timing) fprintf(stderr, "[core]   SFD stage: %zu lattice positions, rule holds at %zu, %zu sweeps, %zu windows; scan %.3f ms (rule %.3f, hits %.3f, lists %.3f)\n", P.a_count, n_fire,
                         all.size(), total, (t1 - t0) * 1e3, (t_rule - t0) * 1e3, (t_hits - t_rule) * 1e3, (t1 - t_hits) * 1e3);
  return rc;
}

// READ stage: a dry run of the state machine from its present state over what is known (A, the SFD sweeps, every
// cached result); the header/payload windows it asks for are recorded and then evaluated in one launch
int plan_reads(lora_demod_core *c) {
  Plan &P = c->cur();
  const unsigned n = c->d_num_symbols;
  P.rd.clear(); P.rd_cur = 0; c->needs.clear();
  const double t0 = now_s();
  const WalkState saved = *c;
  const uint64_t saved_calls = c->n_calls;
  size_t saved_cur[OVERLAP_FACTOR];
  for (unsigned ph = 0; ph < OVERLAP_FACTOR; ph++) saved_cur[ph] = P.sfd_cur[ph];
  const size_t saved_last_up = c->last_up;
  c->shadow = true;
  const uint64_t need = (uint64_t)n + (uint64_t)DEMOD_HISTORY_DEPTH * n - 1;
  const uint64_t a_last = P.walk_end;
  int rc = LORA_HIP_OK;
  try {
    while (c->base + c->total() >= c->pos + need && c->pos <= a_last) {
      unsigned consumed = 0;
      rc = general_work(c, nullptr, nullptr, &consumed);
      if (rc) break;
      c->pos += consumed;
    }
  } catch (const std::bad_alloc &) { rc = LORA_HIP_ENOMEM; }
  c->shadow = false;
  static_cast<WalkState &>(*c) = saved;
  c->n_calls = saved_calls; c->last_up = saved_last_up;
  for (unsigned ph = 0; ph < OVERLAP_FACTOR; ph++) P.sfd_cur[ph] = saved_cur[ph];
  if (rc != LORA_HIP_OK && rc != kShadowStop) return rc;
  // two windows more per packet: the squelch of the real windows may fall a symbol later than predicted
  const uint64_t end = c->base + c->total();
  for (PRun &r : c->needs) {
    for (int extra = 0; extra < 2; extra++)
      if (r.start + (uint64_t)(r.count + 1) * r.stride <= end - (need - n)) r.count++;
    push_planned(c, P.rd, r);
  }
  c->needs.clear();
  std::vector<PRun *> all;
  for (PRun &r : P.rd) all.push_back(&r);
  const double t1 = now_s();
  const int rl = launch_planned(c, all, LORA_WIN_UP, P.r_idx, &P.r_val, c->walk_lane);
  c->t_stage[3] += t1 - t0; c->t_stage[4] += now_s() - t1;
  return rl;
}

// The idle log after the call counter went back to c->n_calls (a dry run undone, or the real run resuming at an earlier
// S_RESET call): runs whose first call lies at or behind the counter go, and the run that began before it loses the calls
// the undone pass had added in place (a run is one S_DETECT_PREAMBLE call per N samples, numbered consecutively).
void trim_idle(lora_demod_core *c) {
  while (!c->idle_calls0.empty() && c->idle_calls0.back() >= c->n_calls) { c->idle.pop_back(); c->idle_calls0.pop_back(); }
  if (c->idle.empty()) return;
  lora_core_idle &r = c->idle.back();
  const uint64_t kept = c->n_calls - c->idle_calls0.back();             // calls of this run that stay made (>= 1)
  const uint64_t last = r.first + (kept - 1) * (uint64_t)c->d_num_symbols;
  if (last < r.last) r.last = last;
}

// The header/payload calls of one packet in the first pass, behind the packet's first (which general_work has made and
// which opened the READ sweep in `needs`): what general_work does for such a call -- get_up's prediction from the lattice
// results, :133-137, :207-212, :363-413 -- with everything that cannot change between two of them looked up once.  The
// calls go on for as long as the state machine stays in READ_HEADER/READ_PAYLOAD, lie one symbol apart (num_consumed = N)
// and are answered from A at every 16th lattice position.  Returns false without having done anything when a condition
// of the short cut does not hold; the caller then makes the call through general_work.
bool first_pass_reads(lora_demod_core *c, Plan &P, uint64_t need, uint64_t a_last) {
  const unsigned N = c->d_num_symbols, step = N / OVERLAP_FACTOR;
  const unsigned off = c->d_offset;
  // get_up would look in these first (exact results), and the test hook counts predictions
  if (off == 0 || !c->up_runs.empty() || !c->pre_runs.empty() || c->fault_at >= 0 || !P.rd.empty() || c->needs.empty()) return false;
  PRun &nr = c->needs.back();
  if (nr.off0 != off || nr.start + (uint64_t)nr.count * N != c->pos) return false;
  size_t k;
  if (!plan_a_find(P, c->pos, step, &k)) return false;
  const uint16_t *ai = P.a_idx.p; const float *av = P.a_val.p;
  const float thr = c->d_threshold;
  const uint64_t end = c->base + c->total();
  const DivF div_f = c->div_f;
  const unsigned pre = div_f(c->d_preamble_idx);
  const bool ldr = c->d_ldr;
  uint64_t pos = c->pos, calls = c->n_calls;
  int state = c->d_state;
  bool any = false;
  while ((state == LORA_S_READ_HEADER || state == LORA_S_READ_PAYLOAD) && k < P.a_count && end >= pos + need && pos <= a_last) {
    const unsigned short idx = ai[k]; const float val = av[k];
    const bool sq = !(val > thr);                                      // :133-137
    c->d_argmax_history.push_front(idx);                               // :207-212
    if (c->d_argmax_history.size() > REQUIRED_PREAMBLE_CHIRPS) c->d_argmax_history.pop_back();
    const int st_in = state;
    const unsigned sym = (N + div_f(idx) - pre) & (N - 1);
    uint8_t div4;
    if (state == LORA_S_READ_HEADER) {                                 // :363-387
      if (sq) state = LORA_S_OUT;
      else if (c->d_symbols.size() == 7) state = LORA_S_READ_PAYLOAD;
      div4 = 1;
    } else {                                                           // :391-413
      if (sq) state = LORA_S_OUT;
      div4 = ldr ? 1 : 0;
    }
    c->d_symbols.push_back((unsigned short)(div4 ? sym / 4 : sym));
    c->patches.push_back(Patch{pos, (uint32_t)calls, (uint32_t)(c->pdu_syms.size() + c->d_symbols.size() - 1),
                               (uint16_t)off, (uint16_t)c->d_preamble_idx, div4, (uint8_t)sq});
    if (c->trace && calls < c->max_trace) {
      lora_core_trace &t = c->trace[calls];
      t.pos = pos; t.state = (uint16_t)st_in; t.chirp_off = (uint16_t)off; t.idx = idx;
      t.consumed = (uint16_t)N; t.val = val; t.n_sfd = 0;
    }
    c->d_power = val; c->d_squelched = sq;
    nr.count++; calls++; pos += N; k += OVERLAP_FACTOR;
    any = true;
  }
  c->d_state = state; c->pos = pos; c->n_calls = calls;
  return any;
}

// First pass of a plan: the state machine runs ONCE over what is known, as in plan_reads, but keeps what it produces --
// trace rows, PDUs -- with the header/payload calls answered from predictions (the squelch from A, the bin unknown).
// After the READ stage has evaluated those windows, every predicted squelch decision is checked against the exact
// maximum; if all hold, the path the first pass took IS the reference's path, and the bins are filled in: symbols
// (:363-413 arithmetic) and trace rows.  The state machine is then put back at the last call that entered S_RESET
// (which clears all state, :215-229) and runs for real from there, so that its final state never rests on a prediction.
// A decision that does not hold puts it back at the S_RESET call before that packet instead.
template <typename Between>
int plan_walk(lora_demod_core *c, lora_pdu_cb cb, void *user, bool overlap, Between between) {
  Plan &P = c->cur();
  const unsigned n = c->d_num_symbols;
  P.rd.clear(); P.rd_cur = 0; c->needs.clear();
  c->patches.clear(); c->snaps.clear(); c->pdu_syms.clear(); c->pdu_off.clear();
  const double t0 = now_s();
  const WalkState saved = *c;
  const uint64_t saved_calls = c->n_calls;
  c->shadow = true; c->deferred = true;
  const uint64_t need = (uint64_t)n + (uint64_t)DEMOD_HISTORY_DEPTH * n - 1;
  const uint64_t a_last = P.walk_end;
  int rc = LORA_HIP_OK;
  try {
    while (c->base + c->total() >= c->pos + need && c->pos <= a_last) {
      if ((c->d_state == LORA_S_READ_HEADER || c->d_state == LORA_S_READ_PAYLOAD) && !c->no_fast && first_pass_reads(c, P, need, a_last)) continue;
      unsigned consumed = 0;
      rc = general_work(c, nullptr, nullptr, &consumed);
      if (rc) break;
      c->pos += consumed;
    }
  } catch (const std::bad_alloc &) { rc = LORA_HIP_ENOMEM; }
  c->shadow = false; c->deferred = false;
  const double t_walked = now_s();
  auto to_start = [&]() {
    static_cast<WalkState &>(*c) = saved; c->n_calls = saved_calls;
    trim_idle(c);
  };
  if (rc != LORA_HIP_OK && rc != kShadowStop) { to_start(); return rc; }
  const uint64_t end = c->base + c->total();
  for (PRun &r : c->needs) {
    for (int extra = 0; extra < 2; extra++)
      if (r.start + (uint64_t)(r.count + 1) * r.stride <= end - (need - n)) r.count++;
    push_planned(c, P.rd, r);
  }
  c->needs.clear();
  std::vector<PRun *> all;
  for (PRun &r : P.rd) all.push_back(&r);
  const double t1 = now_s();
  // the READ stage's round trip (run list to HBM, launch, results back: ~60 us) passes while the host does `between`
  rc = launch_planned(c, all, LORA_WIN_UP, P.r_idx, &P.r_val, c->walk_lane | (overlap ? (unsigned)LORA_HIP_ASYNC : 0u) | c->read_flags);
  const double t1b = now_s();
  double t_between = 0, t_rsync = 0;
  if (overlap) {
    const int rb2 = rc ? LORA_HIP_OK : between();
    const double t1c = now_s();
    const int rs = lora_hip_demod_sync(c->hip, c->walk_lane ? 1 : 0);
    t_between = t1c - t1b; t_rsync = now_s() - t1c;
    c->t_stage[4] += t_rsync;
    if (!rc && rb2) rc = rb2;
    if (!rc && rs) rc = fail(c, rs, "sync");
  }
  const double t2 = now_s();
  c->t_stage[3] += t1 - t0; c->t_stage[4] += t1b - t1;
  for (unsigned ph = 0; ph < OVERLAP_FACTOR; ph++) P.sfd_cur[ph] = 0;     // the read index goes back
  P.rd_cur = 0;
  if (rc) { to_start(); return rc; }
  // where the real run resumes: the last S_RESET call, or one before the first prediction that did not hold
  bool from_start = c->snaps.empty();
  Snap rb = from_start ? Snap{0, saved_calls, 0} : c->snaps.back();
  // one pass over the predicted calls, in the order they were made (= the order of the READ sweeps): check the squelch
  // decision, then fill in the bin.  A decision that does not hold ends the pass: the state machine goes back to the
  // S_RESET call before that packet (what was filled in behind that call is never delivered and is overwritten).
  {
    size_t r = 0;
    const unsigned sh = __builtin_ctz(n);
    for (const Patch &pt : c->patches) {
      if (pt.row >= rb.calls) break;
      while (r < P.rd.size() && !(P.rd[r].off0 == pt.off && pt.pos >= P.rd[r].start &&
                                  pt.pos < P.rd[r].start + ((uint64_t)P.rd[r].count << sh))) r++;
      bool ok = r < P.rd.size() && ((pt.pos - P.rd[r].start) & (n - 1)) == 0;
      size_t k = 0;
      if (ok) {
        k = P.rd[r].at + (size_t)((pt.pos - P.rd[r].start) >> sh);
        ok = (!(P.r_val.p[k] > c->d_threshold)) == (pt.sq != 0);
      }
      if (!ok) {
        c->n_rollbacks++;
        from_start = true; rb = Snap{0, saved_calls, 0};
        for (size_t s = c->snaps.size(); s-- > 0;)
          if (c->snaps[s].calls <= pt.row) { rb = c->snaps[s]; from_start = false; break; }
        break;
      }
      const unsigned idx = P.r_idx.p[k];
      const unsigned sym = (n + c->div_f(idx) - c->div_f(pt.pre)) & (n - 1);  // :373, :398-401
      c->pdu_syms[pt.flat] = (unsigned short)(pt.div4 ? sym / 4 : sym);
      if (c->trace && pt.row < c->max_trace) { c->trace[pt.row].idx = (uint16_t)idx; c->trace[pt.row].val = P.r_val.p[k]; }
    }
  }
  const double t3 = now_s();
  for (size_t i = 0; i < rb.pdus; i++) {
    const size_t b = c->pdu_off[i], e = i + 1 < c->pdu_off.size() ? c->pdu_off[i + 1] : c->pdu_syms.size();
    deliver(c, cb, user, c->pdu_syms.data() + b, e - b);
  }
  if (from_start) to_start();
  else { c->d_state = LORA_S_RESET; c->pos = rb.pos; c->n_calls = rb.calls; c->n_pdus = saved.n_pdus + rb.pdus; }
  trim_idle(c);
  P.rd_cur = 0;
  c->t_stage[3] += t3 - t2; c->t_cb += now_s() - t3;
  if (c->timing) fprintf(stderr, "[core]   first pass: state machine %.3f ms (%llu calls) | run list %.3f | READ launch %.3f, meanwhile %.3f, then waited %.3f | check + fill in %.3f (%zu predicted calls)\n",
                         (t_walked - t0) * 1e3, (unsigned long long)(c->n_calls - saved_calls), (t1 - t_walked) * 1e3, (t1b - t1) * 1e3, t_between * 1e3, t_rsync * 1e3,
                         (t3 - t2) * 1e3, c->patches.size());
  return LORA_HIP_OK;
}

// Lattice sweep of one stretch into plan P: every lattice position from a_start on, all 16 phases, chirp offset 0.  Results
// handed in through lora_demod_core_preload (time sharding: other GPUs swept the lattice) are used as they are.  The sweep is
// only queued (xflags carries LORA_HIP_ASYNC): its results are in P's page-locked arrays once lane 0 has been synchronised.
int stage_lattice(lora_demod_core *c, Plan &P, uint64_t a_start, size_t count, uint64_t walk_end, unsigned xflags) {
  const unsigned n = c->d_num_symbols, step = n / OVERLAP_FACTOR;
  int rc;
  P.active = false; P.replans = 0;
  for (unsigned ph = 0; ph < OVERLAP_FACTOR; ph++) { P.sfd[ph].clear(); P.sfd_cur[ph] = 0; }
  P.rd.clear(); P.rd_cur = 0;
  if ((rc = P.a_idx.ensure(count)) || (rc = P.a_val.ensure(count))) return fail(c, rc, "host_alloc");
  bool have = false;
  if (!c->pre_runs.empty()) {
    memset(P.a_idx.p, 0xFF, count * sizeof(uint16_t));     // 0xFFFF is no bin: N*F < 65536
    for (const Run &r : c->pre_runs) {
      if (r.stride != n || r.off0 != 0 || r.off_step != 0 || r.kind != LORA_WIN_UP || (r.start % step)) continue;
      size_t k = 0;
      if (r.start < a_start) k = (size_t)((a_start - r.start + n - 1) / n);
      for (; k < r.idx.size(); k++) {
        const uint64_t p = r.start + (uint64_t)k * n;
        if ((p - a_start) % step) break;
        const uint64_t j = (p - a_start) / step;
        if (j >= count) break;
        P.a_idx.p[j] = r.idx[k]; P.a_val.p[j] = r.val[k];
      }
    }
    have = true;
    for (size_t j = 0; j < count; j++) if (P.a_idx.p[j] == 0xFFFF) { have = false; break; }
  }
  const double t0 = now_s();
  if (!have && (rc = launch_strided(c, a_start, step, count, 0, 0, LORA_WIN_UP, P.a_idx.p, P.a_val.p, LORA_HIP_OUT_PINNED | xflags))) return rc;
  c->t_stage[0] += now_s() - t0;
  P.a_start = a_start; P.a_count = count; P.walk_end = walk_end;
  c->n_plans++;
  return LORA_HIP_OK;
}

// Planned evaluation of one work() call (see struct Plan), in stretches that overlap in time.  The host's part of a plan --
// the preamble rule over the lattice results, the first pass of the state machine, the check of its predictions -- is
// strictly serial and as long as the GPU's part, so one plan for the whole input left each side idle half the time.  The
// input is therefore cut into stretches of the lattice, each with a plan of its own (kPlans of them, used in turn).  Lane 0
// is filled ahead of the state machine -- sweep 0, sweep 1, SFD 0, sweep 2, SFD 1, ... -- with a mark behind every item
// (lora_hip_demod_mark), so that the host waits for the one result it needs next and not for the lane.  The host
//     evaluates the preamble rule of stretches 0 and 1 as their sweeps arrive and queues their SFD stages, then per stretch k
//     waits for the SFD stage of k,
//     walks k: first pass, READ stage queued (lane 1, so that it does not wait behind the sweeps),
//     evaluates the rule of stretch k+2 and queues its SFD stage while the READ stage is on its way,
//     waits for lane 1, checks the first pass against the READ results and runs the state machine for real.
// Stretches overlap by kStretchOverlap lattice
// positions and the state machine changes plans in the middle of the overlap: far enough into the next stretch for the
// preamble rule's three windows of history and for a sync attempt still in progress (at most 10 calls, :283-292) to have
// been entered inside it.  What a plan holds for a window is exact whichever stretch evaluated it, and whatever no plan
// holds is evaluated on demand, so the cut changes which launch evaluates a window, never a result.
constexpr size_t kStretchMax = (size_t)1 << 21;
constexpr unsigned kMaxStretches = 8;
constexpr size_t kStretchOverlap = 1024;              // 64 symbols
int pipeline_run(lora_demod_core *c, lora_pdu_cb cb, void *user) {
  for (Plan &P : c->plans) P.active = false;
  c->planp = &c->plans[0]; c->walk_lane = 0;
  if (c->plan_mode == 0) return LORA_HIP_OK;
  const unsigned n = c->d_num_symbols, step = n / OVERLAP_FACTOR;
  const uint64_t end = c->base + c->total();
  const uint64_t need = (uint64_t)n + (uint64_t)DEMOD_HISTORY_DEPTH * n - 1;
  if (end < c->pos + need) return LORA_HIP_OK;
  const size_t total = (size_t)((end - c->pos - n) / step + 1);
  const bool dense = c->dens_pdus > 0 && c->dens_pdus * kDenseSymbols >= c->dens_syms;
  if (c->plan_mode == 1 && total < (dense ? kPlanMinDense : kPlanMinSymbols) * OVERLAP_FACTOR) return LORA_HIP_OK;
  size_t K = total / c->stretch_min;
  if (K < 1) K = 1;
  if (K > kMaxStretches) K = kMaxStretches;
  size_t S = (total + K - 1) / K;
  if (S > kStretchMax) { S = kStretchMax; K = (total + S - 1) / S; }
  S = (S + OVERLAP_FACTOR - 1) / OVERLAP_FACTOR * OVERLAP_FACTOR;
  const uint64_t p0 = c->pos;
  const uint64_t last = p0 + (uint64_t)(total - 1) * step;          // last lattice position at which a window fits
  auto a_start = [&](size_t k) { return p0 + (uint64_t)(k ? k * S - kStretchOverlap : 0) * step; };
  auto a_last = [&](size_t k) { const uint64_t e = p0 + (uint64_t)((k + 1) * S - 1) * step; return (k + 1 == K || e > last) ? last : e; };
  auto walk_end = [&](size_t k) { return k + 1 == K ? a_last(k) : a_last(k) - (uint64_t)(kStretchOverlap / 2) * step; };
  if (K == 1) {                                          // one stretch: the three stages in order, on the one lane
    Plan &P = c->plans[0];
    int rc = stage_lattice(c, P, a_start(0), (size_t)((a_last(0) - a_start(0)) / step + 1), walk_end(0), 0);
    if (!rc) rc = plan_sfd(c, P, 0);
    if (rc) return rc;
    c->planp = &P; P.active = true;
    return plan_walk(c, cb, user, false, [] { return (int)LORA_HIP_OK; });   // the caller's loop runs it to the end of the input
  }
  const unsigned async = LORA_HIP_ASYNC;
  c->walk_lane = LORA_HIP_LANE1;
  size_t next_lat = 0, next_scan = 0, walked = 0;
  int rc = LORA_HIP_OK;
  auto marked = [&](int r, const char *what) { return r ? fail(c, r, what) : (int)LORA_HIP_OK; };
  auto issue_lat = [&]() {
    const size_t k = next_lat++;
    int r = stage_lattice(c, c->plans[k % kPlans], a_start(k), (size_t)((a_last(k) - a_start(k)) / step + 1), walk_end(k), async);
    if (!r) r = marked(lora_hip_demod_mark(c->hip, 0, (int)(k % kPlans)), "mark");
    return r;
  };
  // the preamble rule over the next stretch and its SFD stage, queued; behind it the lattice sweep two stretches further on,
  // so that lane 0 holds sweep, sweep, SFD, sweep, SFD, ... and neither the GPU nor the first walk waits for long
  auto scan = [&]() {
    const size_t k = next_scan++;
    const double tw = now_s();
    int r = marked(lora_hip_demod_wait(c->hip, (int)(k % kPlans)), "wait");
    c->t_wait += now_s() - tw;
    if (!r) r = plan_sfd(c, c->plans[k % kPlans], async);
    if (!r) r = marked(lora_hip_demod_mark(c->hip, 0, (int)(kPlans + k % kPlans)), "mark");
    if (!r && next_lat < K && next_lat < walked + kPlans) r = issue_lat();
    return r;
  };
  if (!(rc = issue_lat()) && !(rc = issue_lat()) && !(rc = scan())) rc = scan();
  for (size_t k = 0; !rc && k < K; k++) {
    const double tw = now_s();
    if ((rc = marked(lora_hip_demod_wait(c->hip, (int)(kPlans + k % kPlans)), "wait"))) break;
    c->t_wait += now_s() - tw;
    c->planp = &c->plans[k % kPlans];
    Plan &P = c->cur();
    if (c->pos <= P.walk_end) {                          // (else the state machine is already past this stretch: a long SFD skip)
      P.active = true;
      // the rule over stretch k+2 is evaluated while the READ stage of stretch k is on its way
      if ((rc = plan_walk(c, cb, user, true, [&] { return next_scan < K && next_scan <= k + 2 ? scan() : (int)LORA_HIP_OK; }))) break;
      if (k + 1 == K) break;                             // the last stretch: the caller's loop runs it to the end of the input
      const double tr = now_s();
      while (c->base + c->total() >= c->pos + need && c->pos <= P.walk_end) {
        unsigned consumed = 0;
        if ((rc = general_work(c, cb, user, &consumed))) break;
        c->pos += consumed;
      }
      c->t_real += now_s() - tr;
      P.active = false;
      if (rc) break;
    }
    walked = k + 1;                                      // this stretch's plan is free for the stretch kPlans further on
    while (!rc && next_scan < K && next_scan <= k + 2) rc = scan();
    while (!rc && next_lat < K && next_lat < walked + kPlans) rc = issue_lat();
  }
  if (rc) {
    (void)lora_hip_demod_sync(c->hip, 0);              // nothing may still write a plan's arrays when the caller sees the error
    (void)lora_hip_demod_sync(c->hip, 1);
    for (Plan &P : c->plans) P.active = false;
  }
  c->walk_lane = 0;
  return rc;
}

// (argmax, max) of the up path at (pos, off): lib/demod_impl.cc:177-206
int get_up(lora_demod_core *c, uint64_t pos, unsigned off, unsigned short *idx, float *val) {
  const unsigned n = c->d_num_symbols;
  size_t k;
  Plan &P = c->cur();
  if (P.active) {
    if (off == 0 && plan_a_find(P, pos, n / OVERLAP_FACTOR, &k)) { *idx = P.a_idx.p[k]; *val = P.a_val.p[k]; return LORA_HIP_OK; }
    if (planned_find(P.rd, P.rd_cur, pos, off, n, &k)) { *idx = P.r_idx.p[k]; *val = P.r_val.p[k]; return LORA_HIP_OK; }
  }
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
  if (c->shadow) {
    // dry run: the window is asked for, not evaluated.  Its maximum is that of the offset-0 window at the same place
    // (a chirp offset only rotates the spectrum), which is all the state machine's path depends on while reading.
    if (!P.active || !plan_a_find(P, pos, n / OVERLAP_FACTOR, &k)) return kShadowStop;
    *idx = P.a_idx.p[k]; *val = P.a_val.p[k];
    c->pred = true;
    if (c->deferred && c->fault_at >= 0 && (long)c->n_pred++ == c->fault_at) *val = 0.f;
    if (!c->needs.empty() && c->needs.back().off0 == off &&
        c->needs.back().start + (uint64_t)c->needs.back().count * n == pos) c->needs.back().count++;
    else { PRun r; r.start = pos; r.stride = n; r.count = 1; r.off0 = off; r.off_step = 0; c->needs.push_back(r); }
    return LORA_HIP_OK;
  }
  if (P.active && P.replans < kMaxReplans) {
    // the plan did not foresee this window (an SFD search it had no sweep for, a squelch that fell elsewhere): redo
    // its READ stage from here
    P.replans++; c->n_replans++;
    const int rp = plan_reads(c);
    if (rp) return rp;
    if (planned_find(P.rd, P.rd_cur, pos, off, n, &k)) { *idx = P.r_idx.p[k]; *val = P.r_val.p[k]; return LORA_HIP_OK; }
  }
  c->n_demand++;
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
int get_sfd(lora_demod_core *c, uint64_t call_pos, unsigned off, unsigned short *idx, const uint16_t **seq, size_t *seq_n) {
  const unsigned n = c->d_num_symbols, step = n / OVERLAP_FACTOR;
  size_t k;
  if (c->cur().active) {
    const unsigned ph = (unsigned)((call_pos & (n - 1)) / step);
    size_t rem = 0;
    if (planned_find(c->cur().sfd[ph], c->cur().sfd_cur[ph], call_pos + off, off, n, &k, &rem)) {
      *idx = c->cur().s_idx.p[k];
      // the windows of one call lie N/16 apart with offsets N/16 apart: the same sweep holds the next ones in order
      if (seq) { *seq = c->cur().s_idx.p + k; *seq_n = rem; }
      return LORA_HIP_OK;
    }
  }
  for (size_t r = c->sfd_runs.size(); r-- > 0;) {
    if (c->sfd_runs[r].find(call_pos + off, off, n, &k)) { *idx = c->sfd_runs[r].idx[k]; return LORA_HIP_OK; }
  }
  if (c->shadow) return kShadowStop;
  c->n_demand++;
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
  const uint16_t *seq = nullptr; size_t seq_n = 0;      // planned results of this call's SFD windows, when one sweep holds them
  const int st_in = c->d_state;
  unsigned n_sfd = 0;

  // :177-206 up path; chirp offset is d_offset only while reading header/payload
  const unsigned off = (c->d_state == LORA_S_READ_HEADER || c->d_state == LORA_S_READ_PAYLOAD) ? c->d_offset : 0;
  float max_val = 0.f;
  if (c->deferred && c->d_state == LORA_S_RESET) c->snaps.push_back(Snap{c->pos, c->n_calls, c->pdu_off.size()});
  if (c->d_state == LORA_S_DETECT_PREAMBLE && (!c->shadow || c->deferred)) {
    if (!c->idle.empty() && c->idle.back().last + N == c->pos) c->idle.back().last = c->pos;
    else { c->idle.push_back(lora_core_idle{c->pos, c->pos, c->n_pdus}); c->idle_calls0.push_back(c->n_calls); }
  }
  c->pred = false;
  int rc = get_up(c, c->pos, off, &max_index, &max_val);
  if (rc) return rc;
  const bool pred = c->pred;
  const unsigned short up_idx = max_index;
  // argmax(..., update_squelch = true), :133-137
  c->d_power = max_val;
  c->d_squelched = (c->d_power > c->d_threshold) ? false : true;
  // :207-212
  c->d_argmax_history.push_front(max_index);
  if (c->d_argmax_history.size() > REQUIRED_PREAMBLE_CHIRPS) c->d_argmax_history.pop_back();

  switch (c->d_state) {
    case LORA_S_RESET:                                              // :215-229
      c->d_overlaps = OVERLAP_DEFAULT;
      c->d_offset = 0;
      c->d_symbols.clear();
      c->d_argmax_history.clear();
      c->d_sfd_history.clear(); c->sfd_agree = 0;
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

    case LORA_S_SFD_SYNC: {                                         // :279-359
      c->d_overlaps = OVERLAP_FACTOR;
      if (c->d_sync_recovery_counter++ > DEMOD_SYNC_RECOVERY_COUNT) {
        c->d_state = LORA_S_RESET;
        c->d_overlaps = OVERLAP_DEFAULT;
      }
      const unsigned ovl_sh = (unsigned)__builtin_ctz(c->d_overlaps);
      const int tol = int(F) * LORA_SFD_TOLERANCE;
      for (int ol = 0; ol < (int)c->d_overlaps; ol++) {
        c->d_offset = ((ol * N) >> ovl_sh) & (N - 1);                // ((ol * N) / d_overlaps) % N, d_overlaps 1 or 16
        if ((size_t)ol < seq_n) max_index = seq[ol];
        else if ((rc = get_sfd(c, c->pos, c->d_offset, &max_index, ol == 0 ? &seq : nullptr, &seq_n))) return rc;
        n_sfd++;
        // sfd_agree: how many of the newest entries in a row lie within 2*tol of their predecessor.  The reference's test
        // (every older entry within tol of the newest, :318-330) cannot hold unless all 31 pairs of the history do, so it
        // is only evaluated then: one compare per window instead of a pass over the history whenever two windows agree
        const bool adj = c->d_sfd_history.size() && abs(int(max_index) - int(c->d_sfd_history[0])) <= 2 * tol;
        c->sfd_agree = adj ? c->sfd_agree + 1 : 0;
        c->d_sfd_history.push_front(max_index);
        if (c->d_sfd_history.size() > REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR) {
          c->d_sfd_history.pop_back();
          c->d_sfd_idx = max_index;
          if (c->sfd_agree >= REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR - 1) {
            const unsigned short *hs = c->d_sfd_history.data();
            int worst = 0;
            for (int i = 1; i < REQUIRED_SFD_CHIRPS * OVERLAP_FACTOR; i++) {
              const int dv = abs(int(max_index) - int(hs[i]));
              worst = dv > worst ? dv : worst;
            }
            sfd_found = !(worst > tol);
          }
          if (sfd_found) {
            num_consumed = ((ol * N) >> ovl_sh) + 5 * N / 4;       // skip last quarter chirp
            c->d_offset = (c->d_offset + (N / 4)) & (N - 1);
            c->d_state = LORA_S_READ_HEADER;
            c->d_overlaps = OVERLAP_DEFAULT;
            break;
          }
        }
      }
      break;
    }

    case LORA_S_READ_HEADER:                                        // :363-387
      if (c->d_squelched) c->d_state = LORA_S_OUT;
      else if (c->d_symbols.size() == 7) c->d_state = LORA_S_READ_PAYLOAD;
      c->d_symbols.push_back((unsigned short)(((N + c->div_f(c->d_argmax_history[0]) - c->div_f(c->d_preamble_idx)) & (N - 1)) / 4));
      if (c->deferred && pred)
        c->patches.push_back(Patch{c->pos, (uint32_t)c->n_calls, (uint32_t)(c->pdu_syms.size() + c->d_symbols.size() - 1),
                                   (uint16_t)off, (uint16_t)c->d_preamble_idx, 1, (uint8_t)c->d_squelched});
      break;

    case LORA_S_READ_PAYLOAD:                                       // :391-413
      if (c->d_squelched) c->d_state = LORA_S_OUT;
      if (c->d_ldr)
        c->d_symbols.push_back((unsigned short)(((N + c->div_f(c->d_argmax_history[0]) - c->div_f(c->d_preamble_idx)) & (N - 1)) / 4));
      else
        c->d_symbols.push_back((unsigned short)((N + c->div_f(c->d_argmax_history[0]) - c->div_f(c->d_preamble_idx)) & (N - 1)));
      if (c->deferred && pred)
        c->patches.push_back(Patch{c->pos, (uint32_t)c->n_calls, (uint32_t)(c->pdu_syms.size() + c->d_symbols.size() - 1),
                                   (uint16_t)off, (uint16_t)c->d_preamble_idx, (uint8_t)(c->d_ldr ? 1 : 0), (uint8_t)c->d_squelched});
      break;

    case LORA_S_OUT:                                                // :418-431
      c->n_pdus++;
      if (c->deferred) {
        c->pdu_off.push_back(c->pdu_syms.size());
        c->pdu_syms.insert(c->pdu_syms.end(), c->d_symbols.begin(), c->d_symbols.end());
      } else if (!c->shadow) deliver(c, cb, user, c->d_symbols.data(), c->d_symbols.size());
      c->d_state = LORA_S_RESET;
      break;

    default:
      break;
  }

  if (c->shadow && !c->deferred) { *consumed = num_consumed & 0xFFFFu; return LORA_HIP_OK; }
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
  if (const char *e = getenv("LORA_CORE_FAULT")) c->fault_at = atol(e);
  if (const char *e = getenv("LORA_CORE_TIMING")) c->timing = atoi(e) != 0;
  if (const char *e = getenv("LORA_CORE_NO_FAST")) c->no_fast = atoi(e) != 0;
  if (const char *e = getenv("LORA_CORE_READ_DIRECT")) c->read_flags = atoi(e) ? (unsigned)LORA_HIP_OUT_DIRECT : 0u;
  if (const char *e = getenv("LORA_CORE_STRETCH")) { const long v = atol(e); if (v >= 2 * (long)kStretchOverlap) c->stretch_min = (size_t)v; }
  if (const char *e = getenv("LORA_CORE_PLAN")) { const int m = atoi(e); if (m >= 0 && m <= 2) c->plan_mode = m; }
  c->d_sf = sf; c->d_ldr = ldr != 0; c->d_beta = beta; c->d_fft_size_factor = fft_factor; c->div_f.set(fft_factor);
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
  c->d_argmax_history.clear(); c->d_sfd_history.clear(); c->d_symbols.clear(); c->sfd_agree = 0;
  c->d_sync_recovery_counter = 0;
  c->buf.assign((size_t)DEMOD_HISTORY_DEPTH * c->d_num_symbols - 1, lora_c32{0.f, 0.f});
  c->cur_in = nullptr; c->cur_n = 0;
  c->base = 0; c->pos = 0; c->n_pdus = 0;
  c->dens_pdus = 0; c->dens_syms = 0;
  c->idle.clear(); c->idle_calls0.clear();
  c->up_runs.clear(); c->pre_runs.clear(); c->sfd_runs.clear();
  for (unsigned &m : c->sfd_misses) m = 0;
  for (uint64_t &m : c->sfd_last_miss) m = 0;
  c->staged = false;
  c->n_calls = 0; c->n_windows = 0; c->n_launches = 0;
  c->n_plans = 0; c->n_replans = 0; c->n_demand = 0;
  for (Plan &P : c->plans) P.active = false;
  c->planp = &c->plans[0]; c->shadow = false; c->deferred = false;
  c->n_pred = 0; c->n_rollbacks = 0;
  c->err.clear();
  return LORA_HIP_OK;
}

int lora_demod_core_seek(lora_demod_core_t *c, uint64_t pos) {
  if (!c) return LORA_HIP_EINVAL;
  if (pos % (c->d_num_symbols / OVERLAP_FACTOR)) return LORA_HIP_EINVAL;          // read indices stay on the N/16 lattice
  const int rc = lora_demod_core_reset(c);
  if (rc) return rc;
  c->buf.clear();                        // no scheduler history: the next sample fed IS framed-stream index pos
  c->base = pos; c->pos = pos;
  return LORA_HIP_OK;
}

int lora_demod_core_idle_log(const lora_demod_core_t *c, const lora_core_idle **runs, size_t *n_runs) {
  if (!c || !runs || !n_runs) return LORA_HIP_EINVAL;
  *runs = c->idle.data(); *n_runs = c->idle.size();
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
  c->col_syms.clear(); c->col_off.assign(1, 0);
  for (double &t : c->t_stage) t = 0;
  c->t_cb = 0; c->t_wait = 0; c->t_real = 0;
  const double t_begin = now_s();
  const uint64_t pdus0 = c->n_pdus, pos0 = c->pos;
  try { rc = pipeline_run(c, cb, user); } catch (const std::bad_alloc &) { rc = LORA_HIP_ENOMEM; }
  const double t_walk = now_s();
  while (!rc && c->base + c->total() >= c->pos + need) {
    unsigned consumed = 0;
    rc = general_work(c, cb, user, &consumed);
    if (rc) break;
    c->pos += consumed;                                             // consume_each, :443
  }
  if (c->timing) {
    const double t_end = now_s();
    fprintf(stderr, "[core]   waits for lane 0 %.3f ms | real runs inside the pipeline %.3f | real run behind it %.3f\n", c->t_wait * 1e3, c->t_real * 1e3,
            (t_end - t_walk) * 1e3);
    c->t_stage[5] = t_end - t_walk - (c->t_stage[0] + c->t_stage[1] + c->t_stage[2] + c->t_stage[3] + c->t_stage[4] - (t_walk - t_begin));
    fprintf(stderr, "[core] work %zu samples: lattice sweep %.3f ms | SFD scan %.3f | SFD launch %.3f | dry run %.3f | READ launch %.3f | "
                    "state machine %.3f | PDU callbacks of the first pass %.3f | total %.3f ms; %llu launches so far\n", n_in, c->t_stage[0] * 1e3, c->t_stage[1] * 1e3,
            c->t_stage[2] * 1e3, c->t_stage[3] * 1e3, c->t_stage[4] * 1e3, (c->t_stage[5] - c->t_cb) * 1e3, c->t_cb * 1e3, (t_end - t_begin) * 1e3,
            (unsigned long long)c->n_launches);
  }
  c->dens_pdus += c->n_pdus - pdus0; c->dens_syms += (c->pos - pos0) / c->d_num_symbols;
  if (c->dens_syms > 4096) { c->dens_pdus /= 2; c->dens_syms /= 2; }
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
  for (Plan &P : c->plans) P.active = false;      // planned results describe this call's samples
  c->planp = &c->plans[0];
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

int lora_demod_core_set_collect(lora_demod_core_t *c, int on) {
  if (!c) return LORA_HIP_EINVAL;
  c->collect = on != 0;
  c->col_syms.clear(); c->col_off.assign(1, 0);
  return LORA_HIP_OK;
}
int lora_demod_core_collected(const lora_demod_core_t *c, const uint16_t **symbols, const uint64_t **offsets, size_t *n_pdus) {
  if (!c || !symbols || !offsets || !n_pdus) return LORA_HIP_EINVAL;
  *symbols = c->col_syms.data(); *offsets = c->col_off.data();
  *n_pdus = c->col_off.empty() ? 0 : c->col_off.size() - 1;
  return LORA_HIP_OK;
}

int lora_demod_core_set_plan(lora_demod_core_t *c, int mode) {
  if (!c || mode < 0 || mode > 2) return LORA_HIP_EINVAL;
  c->plan_mode = mode;
  return LORA_HIP_OK;
}
size_t lora_demod_core_rule_hits(const uint16_t *idx, const float *val, float threshold, size_t count, uint32_t *hits, int path) {
  if (!idx || !val || !hits) return 0;
  const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
  if (path == 2 && !have_avx2) return (size_t)-1;
  if (path == 3) {
    size_t nh = 0;
    for (size_t j = 48; j < count; j++) if (rule_at(idx, val, threshold, j)) hits[nh++] = (uint32_t)j;
    return nh;
  }
  if (path == 2 || (path == 0 && have_avx2)) return rule_hits_avx2(idx, val, threshold, count, hits);
  return rule_hits_sse2(idx, val, threshold, count, hits);
}

uint64_t lora_demod_core_plans(const lora_demod_core_t *c) { return c ? c->n_plans : 0; }
uint64_t lora_demod_core_replans(const lora_demod_core_t *c) { return c ? c->n_replans : 0; }
uint64_t lora_demod_core_on_demand(const lora_demod_core_t *c) { return c ? c->n_demand : 0; }
uint64_t lora_demod_core_rollbacks(const lora_demod_core_t *c) { return c ? c->n_rollbacks : 0; }

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
