// lora_hip_demod.hip -- C ABI (include/lora_hip_demod.h) over the gfx950 kernels.
// Product code: never links or calls anything under oracle/.
#include "lora_hip_demod.h"
#include "lora_hip_kernels.hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
#include <algorithm>

using lora_hip::KParams;

// Scratch of lora_hip_demod_batch_strided_noalloc (callers that cannot allocate): the window-group ticket counters
// of the persistent kernels, the only device memory they use beyond the caller's inputs and outputs and the
// handle's tables.  One copy per GPU the library runs on, held until the process ends: kScratchSlots slots of 4 KiB,
// one per handle (claimed at lora_hip_demod_create, given back at lora_hip_demod_destroy), so that two blocks on two
// scheduler threads never share counters.  In host code this name is only a stand-in for hipGetSymbolAddress: what
// reaches a kernel or a memset is the per-device address kept in g_scratch_ptr, taken once per device at
// lora_hip_demod_create.
constexpr size_t kScratchWords = lora_hip::kWaveTicketQueues * lora_hip::kTicketStride;   // 32 counters, 128 B apart: 4 KiB
constexpr int kScratchSlots = LORA_HIP_NOALLOC_MAX_HANDLES;                                // 16 slots: 64 KiB per GPU
static __device__ __align__(256) unsigned scratch[kScratchSlots * kScratchWords];

struct lora_hip_demod {
  int device = 0;
  unsigned sf = 0, fft_factor = 0, n = 0, m = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  float2 *d_up = nullptr, *d_down = nullptr;
  float *d_win = nullptr;
  float2 *d_tw = nullptr;       // the pass tables
  float2 *d_twist = nullptr;    // any-factor kernels: W_M^e, e < M
  bool anyf = false;            // fft_factor outside {1, 2, 4}: run-time number of sub-transforms
  size_t tw_off[4] = {0, 0, 0, 0};
  // staging for host-resident arguments (grown on demand, reused)
  float2 *d_iq = nullptr; size_t cap_iq = 0, staged_n = 0;
  void *d_desc = nullptr; size_t cap_desc = 0;
  unsigned char *d_runs = nullptr; size_t cap_runs = 0;   // lora_hip_demod_batch_runs: the run list in HBM
  // results for host callers: one device block [n_win floats | n_win u16] and its pinned mirror, so that
  // a call costs one device-to-host copy
  unsigned char *d_out = nullptr, *h_out = nullptr; size_t cap_out = 0;
  unsigned *d_ticket = nullptr;          // window-group ticket counters of the persistent kernel (the allocating entry points)
  bool ticket_dirty = false;             // a launch on d_ticket may have failed: clear those counters before the next
  // Second lane (flag LORA_HIP_LANE1): a stream of the handle's own with its own counters, result staging, descriptor
  // and run buffers, so that a small launch (a block's header/payload windows) can overtake the long sweeps queued on the
  // first lane instead of waiting behind them.  Launches on the two lanes may overlap in time; launches on one lane are ordered.
  hipStream_t aux_stream = nullptr;
  unsigned *d_ticket1 = nullptr; bool ticket1_dirty = false;
  void *d_desc1 = nullptr; size_t cap_desc1 = 0;
  unsigned char *d_runs1 = nullptr; size_t cap_runs1 = 0;
  unsigned char *d_out1 = nullptr, *h_out1 = nullptr; size_t cap_out1 = 0;
  unsigned char *h_fetch = nullptr; size_t cap_fetch = 0;     // lora_hip_demod_fetch: page-locked landing block for small copies back
  // run lists of lora_hip_demod_batch_runs: page-locked staging per lane, copied to HBM behind the call (no synchronisation);
  // the event says when the lane's last copy is done with it
  // two blocks per lane, used in turn, so that a call waits for the copy of the call before the last, not for the last
  unsigned char *h_runs[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; size_t cap_hruns[2][2] = {{0, 0}, {0, 0}};
  hipEvent_t runs_done[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; bool runs_pending[2][2] = {{false, false}, {false, false}};
  unsigned runs_turn[2] = {0, 0};
  hipEvent_t marks[LORA_HIP_MARKS] = {nullptr};           // lora_hip_demod_mark / lora_hip_demod_wait
  int scratch_slot = -1;                 // this handle's slot of the per-device `scratch` (-1: all were taken at create)
  bool scratch_dirty = false;            // the same for the handle's scratch slot (each buffer has its own flag)
  long fault_at = -1, n_launch = 0;      // LORA_HIP_FAULT=n: test hook, the n-th launch on the handle fails and leaves its counters undefined
  unsigned long long *d_dbg = nullptr;   // LORA_STAMPS builds only
  char hip_err[256] = {0};
  const char *kernel_name = "";
  int num_cu = 256;
  unsigned blocks_per_cu = 0;      // 0: the kernel variant's own choice (LORA_HIP_BLOCKS_PER_CU overrides)
};

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned *g_scratch_ptr[kMaxDevices] = {nullptr};   // device address of `scratch` on each GPU, by device index
unsigned g_scratch_used[kMaxDevices] = {0};         // bit s: slot s of that GPU's scratch belongs to a live handle
unsigned g_scratch_stale[kMaxDevices] = {0};        // bit s: slot s was given back with counters that may not be zero

int hip_fail(lora_hip_demod *h, hipError_t e, const char *what) {
  // any HIP failure on the handle may have been a launch that died half way: both counter buffers are suspect, and each
  // is cleared (and its own flag reset) by the next launch that uses it
  if (h) { snprintf(h->hip_err, sizeof h->hip_err, "%s: %s", what, hipGetErrorString(e)); h->ticket_dirty = true; h->ticket1_dirty = true; h->scratch_dirty = true; }
  return LORA_HIP_EHIP;
}
#define HIP_TRY(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail((h), e_, #call); } while (0)

// Makes the handle's device current for the scope of one entry point and puts the caller's device back
// afterwards: the library runs inside processes (GNU Radio flowgraphs with several blocks, torch) that keep
// their own idea of the current device per thread.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err != hipSuccess) { prev = -1; return; }
    if (prev == device) { prev = -1; return; }       // nothing to restore
    err = hipSetDevice(device);
    if (err != hipSuccess) prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

template <typename T>
int grow(lora_hip_demod *h, T **buf, size_t *cap, size_t want) {
  if (want <= *cap) return LORA_HIP_OK;
  if (*buf) { hipError_t e = hipFree(*buf); *buf = nullptr; *cap = 0; if (e != hipSuccess) return hip_fail(h, e, "hipFree"); }
  size_t ncap = want + want / 4;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(buf), ncap * sizeof(T));
  if (e != hipSuccess) { *buf = nullptr; hip_fail(h, e, "hipMalloc"); return LORA_HIP_ENOMEM; }
  *cap = ncap;
  return LORA_HIP_OK;
}

#ifndef LORA_NO_BOUNCE
#define LORA_NO_BOUNCE 0   // developer knob: 1 = small pageable copies are left to the runtime, as before
#endif
#ifndef LORA_NO_MIRROR_DIRECT
#define LORA_NO_MIRROR_DIRECT 0   // developer knob: 1 = small host-output calls go through the device block and a copy, as before
#endif
constexpr size_t kMirrorDirectMax = 8192;   // windows
#ifndef LORA_NO_CONST
#define LORA_NO_CONST 0   // developer knob: 1 = never use the register-resident-taps variant
#endif
typedef void (*kernel_fn)(const KParams);
// fn: any window list; fn_const: the strided sweep with one chirp offset (taps in registers), which may have its own geometry
struct Variant {
  kernel_fn fn, fn_const; size_t lds; int wpb, threads, blocks_per_cu; const char *name;
  size_t lds_const; int wpb_const, blocks_per_cu_const; const char *name_const;
  kernel_fn fn_const_nt = nullptr;     // fn_const with non-temporal IQ loads, for sweeps whose windows do not overlap (SF >= 8)
};

template <int LOGN, int F>
Variant make_variant(const char *name) {
  Variant v{&lora_hip::demod_kernel<LOGN, F, false>, &lora_hip::demod_kernel<LOGN, F, true>,
            lora_hip::demod_lds_bytes<LOGN, F>(), lora_hip::demod_wpb<LOGN, F>(), lora_hip::block_threads(LOGN, F),
            lora_hip::min_waves(LOGN, F) * 4 / (lora_hip::block_threads(LOGN, F) / 64), name, 0, 0, 0, name};
  v.lds_const = v.lds; v.wpb_const = v.wpb; v.blocks_per_cu_const = v.blocks_per_cu;
  // measured +1.7 ... +5 % from SF8 up (profiles/r03_ab_nt_loads.txt), -0.8 % at SF7
  if constexpr (LORA_PF_NT && LOGN >= 8) v.fn_const_nt = &lora_hip::demod_kernel<LOGN, F, true, false, true>;
  // the list kernel keeps the chirp tables and the window in LDS where there is room (lora_hip::tap_lds)
  if (lora_hip::tap_lds(LOGN, false, false)) v.lds = lora_hip::demod_tap_lds_offset<LOGN, F>() + lora_hip::demod_tap_lds_bytes<LOGN>();
  return v;
}

template <int LOGN>
Variant make_variant_anyf(const char *name) {
  // no register-resident-taps variant: x stays live across the run-time loop and the registers are needed there
  Variant v{&lora_hip::demod_kernel<LOGN, 1, false, true>, &lora_hip::demod_kernel<LOGN, 1, false, true>,
            lora_hip::demod_lds_bytes<LOGN, 1>(), lora_hip::demod_wpb<LOGN, 1>(), lora_hip::block_threads(LOGN, 1),
            lora_hip::min_waves(LOGN, 1) * 4 / (lora_hip::block_threads(LOGN, 1) / 64), name, 0, 0, 0, name};
  v.lds_const = v.lds; v.wpb_const = v.wpb; v.blocks_per_cu_const = v.blocks_per_cu;
  return v;
}

// one thread per window: find its run by bisection over the runs' first-window numbers, write its descriptor
struct RunDev { unsigned long long pos0; unsigned stride, first; unsigned short off, off_step; unsigned kind; };
__global__ void expand_runs_kernel(const RunDev *runs, unsigned n_runs, unsigned n_win, unsigned nmask, lora_hip::WinDesc *desc) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_win) return;
  unsigned lo = 0, hi = n_runs - 1;
  while (lo < hi) {
    const unsigned mid = (lo + hi + 1) >> 1;
    if (runs[mid].first <= i) lo = mid; else hi = mid - 1;
  }
  const RunDev r = runs[lo];
  const unsigned k = i - r.first;
  lora_hip::WinDesc d;
  d.pos = r.pos0 + (unsigned long long)k * r.stride;
  d.off = (unsigned short)((r.off + k * r.off_step) & nmask);
  d.kind = (unsigned char)r.kind;
  for (int b = 0; b < 5; b++) d.pad[b] = 0;
  desc[i] = d;
}

#define LORA_VARIANTS_SF(SF) \
  case SF: \
    if (f == 1) { *v = make_variant<SF, 1>("demod_kernel<" #SF ",1>"); return true; } \
    if (f == 2) { *v = make_variant<SF, 2>("demod_kernel<" #SF ",2>"); return true; } \
    if (f == 4) { *v = make_variant<SF, 4>("demod_kernel<" #SF ",4>"); return true; } \
    *v = make_variant_anyf<SF>("demod_kernel<" #SF ",any>"); return true;

bool pick_variant(unsigned sf, unsigned f, Variant *v) {
  switch (sf) {
    LORA_VARIANTS_SF(6) LORA_VARIANTS_SF(7) LORA_VARIANTS_SF(8) LORA_VARIANTS_SF(9)
    LORA_VARIANTS_SF(10) LORA_VARIANTS_SF(11) LORA_VARIANTS_SF(12)
    default: return false;
  }
}

// own_ticket: the counters are the handle's (d_ticket on lane 0, d_ticket1 on lane 1; the allocating entry points);
// otherwise the handle's scratch slot (lane 0 only)
int launch(lora_hip_demod *h, KParams &p, const bool own_ticket, const int lane = 0) {
  unsigned *const ticket = !own_ticket ? g_scratch_ptr[h->device] + (size_t)h->scratch_slot * kScratchWords : (lane ? h->d_ticket1 : h->d_ticket);
  bool &dirty = !own_ticket ? h->scratch_dirty : (lane ? h->ticket1_dirty : h->ticket_dirty);
  const hipStream_t stream = lane ? h->aux_stream : h->stream;
#if LORA_CLOCK || LORA_STAMPS
  if (!own_ticket) {   // the diagnostic builds allocate and synchronise in here: not what the no-allocation entry point promises
    snprintf(h->hip_err, sizeof h->hip_err, "lora_hip_demod_batch_strided_noalloc is not available in LORA_CLOCK / LORA_STAMPS builds");
    return LORA_HIP_EUNSUPPORTED;
  }
#endif
  Variant v;
  if (!pick_variant(h->sf, h->fft_factor, &v)) return LORA_HIP_EUNSUPPORTED;
  if (p.n_win == 0) return LORA_HIP_OK;
  // one chirp offset for the whole sweep (the steady-state stride-N path): taps stay in registers
  const bool const_off = (p.desc == nullptr && p.off_step == 0) && !LORA_NO_CONST;
  const int wpb = const_off ? v.wpb_const : v.wpb;
  const size_t lds = const_off ? v.lds_const : v.lds;
  // persistent workgroups: at most blocks_per_cu per CU; they share the window groups through ticket counters
  unsigned blocks = (p.n_win + wpb - 1) / wpb;
  const unsigned resident = (unsigned)h->num_cu * (h->blocks_per_cu ? h->blocks_per_cu : (unsigned)(const_off ? v.blocks_per_cu_const : v.blocks_per_cu));
  if (blocks > resident) blocks = resident;
#if LORA_CLOCK
  {
    static unsigned long long n_launch = 0;
    if (!h->d_dbg) (void)hipMalloc((void **)&h->d_dbg, 4096 * 2 * sizeof(unsigned long long));
    p.dbg = h->d_dbg;
    if ((++n_launch % 400) == 0) {     // report now and then: the previous launch's numbers
      std::vector<unsigned long long> hd(2 * blocks);
      (void)hipStreamSynchronize(h->stream);
      (void)hipMemcpy(hd.data(), h->d_dbg, hd.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> mhz;
      for (unsigned b = 0; b < blocks; b++) if (hd[2 * b + 1]) mhz.push_back(100.0 * (double)hd[2 * b] / (double)hd[2 * b + 1]);
      std::sort(mhz.begin(), mhz.end());
      if (!mhz.empty()) fprintf(stderr, "[clock] launch %llu: in-kernel clock median %.0f MHz (min %.0f max %.0f), lifetime median %.1f us\n",
                                n_launch, mhz[mhz.size() / 2], mhz.front(), mhz.back(), (double)hd[2 * (blocks / 2) + 1] / 100.0);
    }
  }
#endif
#if LORA_STAMPS
  const size_t ndbg = (size_t)blocks * (v.threads / 64) * 8;
  if (!h->d_dbg) (void)hipMalloc((void **)&h->d_dbg, (size_t)h->num_cu * 8 * 4 * 8 * sizeof(unsigned long long));
  (void)hipMemsetAsync(h->d_dbg, 0, ndbg * sizeof(unsigned long long), h->stream);
  p.dbg = h->d_dbg;
#endif
  // The counters must be zero at the launch.  They are when the buffer is new (create() clears the handle's own; the
  // library's __device__ array starts as zeros) and after every launch that ran to its end: the kernel's last workgroup
  // clears them.  A launch that failed leaves them undefined, so the next one clears them itself.
  if (dirty) {
    hipError_t em = hipMemsetAsync(ticket, 0, kScratchWords * sizeof(unsigned), stream);
    if (em != hipSuccess) return hip_fail(h, em, "hipMemsetAsync");
    dirty = false;
  }
  if (h->fault_at > 0 && ++h->n_launch == h->fault_at) {   // test hook: what a launch that died half way leaves behind
    (void)hipMemsetAsync(ticket, 0xA5, kScratchWords * sizeof(unsigned), stream);
    return hip_fail(h, hipErrorLaunchFailure, "kernel launch (LORA_HIP_FAULT test hook)");
  }
  p.ticket = ticket;
  p.twist = h->d_twist; p.fr = h->fft_factor;
  const kernel_fn fn = !const_off ? v.fn : (v.fn_const_nt && p.stride >= h->n) ? v.fn_const_nt : v.fn_const;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(v.threads), lds, stream, p);
#if LORA_STAMPS
  {
    std::vector<unsigned long long> hd(ndbg);
    (void)hipMemcpyAsync(hd.data(), h->d_dbg, ndbg * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
    (void)hipStreamSynchronize(h->stream);
    double sum[8] = {0};
    for (size_t i = 0; i < ndbg; i++) sum[i % 8] += (double)hd[i];
    const double nw = (double)blocks * (v.threads / 64);
    {   // spread of per-wave lifetimes (100 MHz ticks)
      std::vector<double> life;
      for (size_t w = 0; w < ndbg / 8; w++) life.push_back((double)hd[w * 8 + 7]);
      std::sort(life.begin(), life.end());
      fprintf(stderr, "[stamps] wave lifetime (us): min %.1f p10 %.1f p33 %.1f p50 %.1f p67 %.1f p90 %.1f max %.1f\n", life.front() / 100,
              life[life.size() / 10] / 100, life[life.size() / 3] / 100, life[life.size() / 2] / 100, life[2 * life.size() / 3] / 100,
              life[9 * life.size() / 10] / 100, life.back() / 100);
    }
    fprintf(stderr, "[stamps] %s blocks=%u per-wave cycles: top %.0f | taps+dechirp %.0f | fft compute %.0f | barrier wait %.0f | lds gather %.0f | magsq %.0f | reduce+store %.0f | s_memtime/s_memrealtime %.2f (x100 MHz)\n",
            v.name, blocks, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[5] / nw, sum[6] / nw, sum[3] / nw, sum[4] / nw, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6]) / sum[7]);
  }
#endif
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dirty = true; return hip_fail(h, e, "kernel launch"); }
  return LORA_HIP_OK;
}

// GNU Radio 3.7 gr-fft window.cc Izero(): I0 by power series until the term < 1e-21 * sum
double izero(double x) {
  double sum = 1, u = 1, halfx = x / 2.0;
  int n = 1;
  do {
    double temp = halfx / (double)n;
    n += 1;
    temp *= temp;
    u *= temp;
    sum += u;
  } while (u >= 1E-21 * sum);
  return sum;
}

}  // namespace

extern "C" {

const char *lora_hip_strerror(int code) {
  switch (code) {
    case LORA_HIP_OK: return "ok";
    case LORA_HIP_EINVAL: return "invalid argument";
    case LORA_HIP_EUNSUPPORTED: return "unsupported fft_factor (2^sf * fft_factor must be below 65536, the reference's 16-bit FFT size)";
    case LORA_HIP_ENOMEM: return "out of memory";
    case LORA_HIP_EHIP: return "HIP runtime error";
    case LORA_HIP_ENODEVICE: return "no usable GPU device";
    default: return "unknown error";
  }
}

const char *lora_hip_last_hip_error(const lora_hip_demod_t *h) { return h ? h->hip_err : ""; }

int lora_hip_make_chirp_tables(unsigned sf, lora_c32 *up, lora_c32 *down) {
  if (sf < 6 || sf > 12 || !up || !down) return LORA_HIP_EINVAL;
  const int n = 1 << sf;
  // lib/demod_impl.cc:92-101: float phase re-rounded every step, double accumulator,
  // std::polar(1.0, acc) in double, narrowed to complex<float>
  float phase = -M_PI;
  double accumulator = 0;
  for (int i = 0; i < 2 * n; i++) {
    accumulator += phase;
    const double c = std::cos(accumulator), s = std::sin(accumulator);
    down[i].re = (float)c; down[i].im = (float)(-s);
    up[i].re = (float)c;   up[i].im = (float)s;
    phase += (2 * M_PI) / n;
  }
  return LORA_HIP_OK;
}

int lora_hip_make_kaiser(unsigned ntaps, double beta, float *w) {
  if (ntaps < 2 || !w || beta < 0) return LORA_HIP_EINVAL;
  const double ibeta = 1.0 / izero(beta);
  const double inm1 = 1.0 / ((double)(ntaps - 1));
  for (unsigned i = 0; i < ntaps; i++) {
    double temp = 2 * (int)i * inm1 - 1;
    w[i] = (float)(izero(beta * std::sqrt(1.0 - temp * temp)) * ibeta);
  }
  return LORA_HIP_OK;
}

int lora_hip_demod_create(lora_hip_demod_t **out, int device, unsigned sf, unsigned fft_factor, float beta,
                          const float *window_n, const lora_c32 *up_2n, const lora_c32 *down_2n) {
  if (!out) return LORA_HIP_EINVAL;
  *out = nullptr;
  if (sf < 6 || sf > 12 || fft_factor == 0) return LORA_HIP_EINVAL;   // asserts at demod_impl.cc:70-72
  // d_fft_size is an unsigned short (demod_impl.h:52): a product of 65536 or more wraps in the reference
  if (((unsigned long long)fft_factor << sf) > 65535ull) return LORA_HIP_EUNSUPPORTED;
  const bool all_null = !window_n && !up_2n && !down_2n;
  if (!all_null && (!window_n || !up_2n || !down_2n)) return LORA_HIP_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LORA_HIP_ENODEVICE;
  if (device < 0 || device >= ndev) return LORA_HIP_ENODEVICE;

  lora_hip_demod *h = new (std::nothrow) lora_hip_demod();
  if (!h) return LORA_HIP_ENOMEM;
  h->device = device; h->sf = sf; h->fft_factor = fft_factor;
  h->n = 1u << sf; h->m = h->n * fft_factor;
  h->anyf = (fft_factor != 1 && fft_factor != 2 && fft_factor != 4);
  Variant v;
  pick_variant(sf, fft_factor, &v);
  h->kernel_name = v.name_const;   // the kernel of the steady-state sweep
  if (const char *e = getenv("LORA_HIP_FAULT")) h->fault_at = atol(e);
  if (const char *e = getenv("LORA_HIP_BLOCKS_PER_CU")) { int b = atoi(e); if (b >= 1 && b <= 8) h->blocks_per_cu = (unsigned)b; }   // tuning knob

  const size_t n = h->n, m = h->m;
  std::vector<lora_c32> up(2 * n), down(2 * n);
  std::vector<float> win(n);
  if (all_null) {
    lora_hip_make_chirp_tables(sf, up.data(), down.data());
    lora_hip_make_kaiser((unsigned)n, beta, win.data());
  } else {
    memcpy(up.data(), up_2n, 2 * n * sizeof(lora_c32));
    memcpy(down.data(), down_2n, 2 * n * sizeof(lora_c32));
    memcpy(win.data(), window_n, n * sizeof(float));
  }
  // Twiddle tables, evaluated in double and rounded once (layouts: lora_hip_kernels.hip.h):
  //   (the pass-0 input constants W_M^(rr*m*N/R) are compile-time constants of the kernel)
  //   twp[ps][rr][m-1][jm]  = W_M^(m*(N/L_ps)*(F*jm + rr))         merged pass twiddles, ps >= 1
  //   (a last pass too big for LDS stores m = 1 only; the kernel keeps it in registers)
  std::vector<lora_c32> tw;
  // the any-factor kernel runs N-point transforms (F = 1 tables) behind a full input twist
  const unsigned F = h->anyf ? 1u : fft_factor;
  const size_t mt = n * F;
  auto wm = [&](unsigned long long e) {
    const double a = -2.0 * M_PI * (double)(e % mt) / (double)mt;
    return lora_c32{(float)std::cos(a), (float)std::sin(a)};
  };
  const int NP = lora_hip::plan_passes((int)sf);
  {
    unsigned long long NS = lora_hip::plan_radix((int)sf, 0);
    for (int ps = 1; ps < NP; ps++) {
      const unsigned long long r = lora_hip::plan_radix((int)sf, ps), L = NS * r;
      h->tw_off[ps] = tw.size();
      const bool small = lora_hip::pass_table_in_lds((int)sf, (int)F, ps);
      for (unsigned rr = 0; rr < F; rr++) {
        const bool poly_gen = LORA_POLY16 && LORA_POLY_GEN && (r == 16 || r == 8);
        if (small && LORA_POLY16 && ((sf == 12 && ps == 1) || poly_gen)) {
          // twiddled radix-16 / radix-8 butterflies run as a polynomial evaluation (Dft<16>::run_poly, dft8_poly): the first
          // r/2 rows hold the factors of the butterfly's base twiddle b = W_M^((N/L)*(F*jm + rr)) --
          //   r = 16: b^8, b^4, b^2, b^2*W8, b, b*W16, b*W16^2, b*W16^3        r = 8: b^4, b^2, b, b*W8
          // -- and the other rows of the [r-1] layout are not read
          const unsigned long long q = mt / r;                  // W_r = W_M^(M/r)
          const unsigned long long kk16[15] = {8, 4, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, cc16[15] = {0, 0, 0, 2, 0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 0};
          const unsigned long long kk8[7] = {4, 2, 1, 1, 0, 0, 0}, cc8[7] = {0, 0, 0, 1, 0, 0, 0};
          const unsigned long long *kk = r == 16 ? kk16 : kk8, *cc = r == 16 ? cc16 : cc8;
          for (unsigned long long row = 0; row + 1 < r; row++)
            for (unsigned long long jm = 0; jm < NS; jm++) tw.push_back(wm(kk[row] * (n / L) * (F * jm + rr) + cc[row] * q));
        } else if (small) {   // float2 [F][r-1][NS]
          for (unsigned long long mm = 1; mm < r; mm++)
            for (unsigned long long jm = 0; jm < NS; jm++) tw.push_back(wm(mm * (n / L) * (F * jm + rr)));
        } else {              // float2 [F][NS]: the base twiddle of each butterfly only
          for (unsigned long long jm = 0; jm < NS; jm++) tw.push_back(wm((n / L) * (F * jm + rr)));
        }
      }
      NS = L;
    }
  }
  auto fail = [&](int code) { lora_hip_demod_destroy(h); return code; };
  DeviceGuard guard(device);
  if (guard.err != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(LORA_HIP_EHIP);
  h->stream = h->own_stream;
  if (hipMalloc((void **)&h->d_up, 2 * n * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMalloc((void **)&h->d_down, 2 * n * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMalloc((void **)&h->d_win, n * sizeof(float)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMalloc((void **)&h->d_tw, tw.size() * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMalloc((void **)&h->d_ticket, kScratchWords * sizeof(unsigned)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMemset(h->d_ticket, 0, kScratchWords * sizeof(unsigned)) != hipSuccess) return fail(LORA_HIP_EHIP);
  {
    // the second lane is for small launches that must not wait behind the first lane's queue: highest priority, so that its
    // workgroups go first when a running sweep lets go of the CUs
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
    if (getenv("LORA_HIP_NO_PRIORITY")) greatest = least;
    if (hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, greatest) != hipSuccess) return fail(LORA_HIP_EHIP);
  }
  if (hipMalloc((void **)&h->d_ticket1, kScratchWords * sizeof(unsigned)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMemset(h->d_ticket1, 0, kScratchWords * sizeof(unsigned)) != hipSuccess) return fail(LORA_HIP_EHIP);
  for (int l = 0; l < 2; l++)
    for (int b = 0; b < 2; b++)
      if (hipEventCreateWithFlags(&h->runs_done[l][b], hipEventDisableTiming) != hipSuccess) return fail(LORA_HIP_EHIP);
  for (int m = 0; m < LORA_HIP_MARKS; m++)
    if (hipEventCreateWithFlags(&h->marks[m], hipEventDisableTiming) != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipMemcpy(h->d_up, up.data(), 2 * n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipMemcpy(h->d_down, down.data(), 2 * n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipMemcpy(h->d_win, win.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  if (h->anyf) {
    std::vector<lora_c32> twist(m);
    for (size_t e = 0; e < m; e++) {
      const double a = -2.0 * M_PI * (double)e / (double)m;
      twist[e] = lora_c32{(float)std::cos(a), (float)std::sin(a)};
    }
    if (hipMalloc((void **)&h->d_twist, m * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
    if (hipMemcpy(h->d_twist, twist.data(), m * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(LORA_HIP_EHIP);
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // the no-allocation entry point's scratch: its address on this device, once per device, with the device current
  if (device < kMaxDevices) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
      unsigned *sp = nullptr;
      hipError_t e = hipGetSymbolAddress((void **)&sp, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !sp) { hip_fail(h, e, "hipGetSymbolAddress(scratch)"); return fail(LORA_HIP_EHIP); }
      g_scratch_ptr[device] = sp;
    }
    // this handle's slot of the scratch; none left is not an error here (only the no-allocation entry point needs one)
    for (int sl = 0; sl < kScratchSlots; sl++) {
      if (!(g_scratch_used[device] >> sl & 1u)) {
        g_scratch_used[device] |= 1u << sl;
        h->scratch_slot = sl;
        h->scratch_dirty = g_scratch_stale[device] >> sl & 1u;
        g_scratch_stale[device] &= ~(1u << sl);
        break;
      }
    }
  }
  // kernels with more dynamic LDS than the runtime's default limit need the opt-in
  if (v.lds > 48 * 1024 || v.lds_const > 48 * 1024) {
    if (hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)v.fn_const, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds_const) != hipSuccess ||
        (v.fn_const_nt && hipFuncSetAttribute((const void *)v.fn_const_nt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds_const) != hipSuccess))
      return fail(LORA_HIP_EHIP);
  }
  *out = h;
  return LORA_HIP_OK;
}

void lora_hip_demod_destroy(lora_hip_demod_t *h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); (void)hipStreamDestroy(h->own_stream); }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  for (int l = 0; l < 2; l++)
    for (int b = 0; b < 2; b++) {
      if (h->runs_done[l][b]) (void)hipEventDestroy(h->runs_done[l][b]);
      if (h->h_runs[l][b]) (void)hipHostFree(h->h_runs[l][b]);
    }
  for (int m = 0; m < LORA_HIP_MARKS; m++)
    if (h->marks[m]) (void)hipEventDestroy(h->marks[m]);
  (void)hipFree(h->d_ticket1); (void)hipFree(h->d_desc1); (void)hipFree(h->d_runs1); (void)hipFree(h->d_out1);
  if (h->h_out1) (void)hipHostFree(h->h_out1);
  if (h->h_fetch) (void)hipHostFree(h->h_fetch);
  if (h->scratch_slot >= 0 && h->device < kMaxDevices) {   // the slot goes back; the next owner clears it if this one left it undefined
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch_used[h->device] &= ~(1u << h->scratch_slot);
    if (h->scratch_dirty) g_scratch_stale[h->device] |= 1u << h->scratch_slot;
    h->scratch_slot = -1;
  }
  (void)hipFree(h->d_up); (void)hipFree(h->d_down); (void)hipFree(h->d_win); (void)hipFree(h->d_tw); (void)hipFree(h->d_twist);
  (void)hipFree(h->d_iq); (void)hipFree(h->d_desc); (void)hipFree(h->d_runs); (void)hipFree(h->d_out); (void)hipFree(h->d_ticket);
  if (h->h_out) (void)hipHostFree(h->h_out);
  delete h;
}

int lora_hip_demod_set_stream(lora_hip_demod_t *h, void *hip_stream) {
  if (!h) return LORA_HIP_EINVAL;
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return LORA_HIP_OK;
}

// Small copies between pageable host memory and the device go through one page-locked block of the handle: left to the
// runtime, a pageable copy of a few hundred KiB pins and unpins the caller's pages, and what that costs varied between 10 and
// several hundred microseconds from process to process (profiles/r03_stream.txt: the same 128 KiB chunks at 100 or 550 us
// per call).  All users are synchronous, so one block serves them.
constexpr size_t kBounceMax = 1u << 20;
static int bounce(lora_hip_demod_t *h, size_t bytes) {
  if (bytes <= h->cap_fetch) return LORA_HIP_OK;
  if (h->h_fetch) { (void)hipHostFree(h->h_fetch); h->h_fetch = nullptr; h->cap_fetch = 0; }
  const size_t ncap = bytes < 65536 ? 65536 : kBounceMax;
  hipError_t e = hipHostMalloc((void **)&h->h_fetch, ncap, hipHostMallocDefault);
  if (e != hipSuccess) { h->h_fetch = nullptr; hip_fail(h, e, "hipHostMalloc"); return LORA_HIP_ENOMEM; }
  h->cap_fetch = ncap;
  return LORA_HIP_OK;
}

int lora_hip_demod_stage_iq2(lora_hip_demod_t *h, const lora_c32 *a, size_t n_a, const lora_c32 *b, size_t n_b) {
  if (!h || (!a && n_a) || (!b && n_b)) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  h->staged_n = 0;
  const size_t n = n_a + n_b;
  int rc = grow(h, &h->d_iq, &h->cap_iq, n ? n : 1);
  if (rc != LORA_HIP_OK) return rc;
  if (n && n * sizeof(float2) <= kBounceMax && !LORA_NO_BOUNCE) {
    if ((rc = bounce(h, n * sizeof(float2))) != LORA_HIP_OK) return rc;
    if (n_a) memcpy(h->h_fetch, a, n_a * sizeof(float2));
    if (n_b) memcpy(h->h_fetch + n_a * sizeof(float2), b, n_b * sizeof(float2));
    HIP_TRY(h, hipMemcpyAsync(h->d_iq, h->h_fetch, n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  } else {
    if (n_a) HIP_TRY(h, hipMemcpyAsync(h->d_iq, a, n_a * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    if (n_b) HIP_TRY(h, hipMemcpyAsync(h->d_iq + n_a, b, n_b * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));     // the host buffers may be reused by the caller
  h->staged_n = n;
  return LORA_HIP_OK;
}

int lora_hip_demod_stage_iq2_dev(lora_hip_demod_t *h, const lora_c32 *a, size_t n_a, const lora_c32 *b_dev, size_t n_b) {
  if (!h || (!a && n_a) || (!b_dev && n_b)) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  h->staged_n = 0;
  const size_t n = n_a + n_b;
  int rc = grow(h, &h->d_iq, &h->cap_iq, n ? n : 1);
  if (rc != LORA_HIP_OK) return rc;
  if (n_a && n_a * sizeof(float2) <= kBounceMax && !LORA_NO_BOUNCE) {
    if ((rc = bounce(h, n_a * sizeof(float2))) != LORA_HIP_OK) return rc;
    memcpy(h->h_fetch, a, n_a * sizeof(float2));
    HIP_TRY(h, hipMemcpyAsync(h->d_iq, h->h_fetch, n_a * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  } else if (n_a) HIP_TRY(h, hipMemcpyAsync(h->d_iq, a, n_a * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  if (n_b) HIP_TRY(h, hipMemcpyAsync(h->d_iq + n_a, b_dev, n_b * sizeof(float2), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->staged_n = n;
  return LORA_HIP_OK;
}

int lora_hip_demod_fetch(lora_hip_demod_t *h, const lora_c32 *src_dev, size_t n, lora_c32 *dst) {
  if (!h || (n && (!src_dev || !dst))) return LORA_HIP_EINVAL;
  if (!n) return LORA_HIP_OK;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  // A copy into pageable memory makes the runtime pin and unpin the destination (tens of microseconds for a few KiB); what
  // the block fetches per work() call is the unconsumed tail, < 4N samples, so it lands in a page-locked block of the handle
  const size_t bytes = n * sizeof(float2);
  if (bytes <= kBounceMax) {
    const int rb = bounce(h, bytes);
    if (rb != LORA_HIP_OK) return rb;
    HIP_TRY(h, hipMemcpyAsync(h->h_fetch, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(dst, h->h_fetch, bytes);
    return LORA_HIP_OK;
  }
  HIP_TRY(h, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return LORA_HIP_OK;
}

int lora_hip_demod_stage_iq(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples) {
  return lora_hip_demod_stage_iq2(h, iq, n_samples, nullptr, 0);
}

unsigned lora_hip_demod_sf(const lora_hip_demod_t *h) { return h ? h->sf : 0; }
unsigned lora_hip_demod_fft_factor(const lora_hip_demod_t *h) { return h ? h->fft_factor : 0; }
const char *lora_hip_demod_kernel_name(const lora_hip_demod_t *h) { return h ? h->kernel_name : ""; }

static int run_batch(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples, KParams &p,
                     const lora_win_desc *desc, size_t n_win, uint16_t *max_idx, float *max_val, unsigned flags) {
  const bool iq_staged = flags & LORA_HIP_IQ_STAGED;
  const bool no_val = (flags & LORA_HIP_NO_VAL) && !(flags & LORA_HIP_OUT_ON_DEVICE);
  if (!h || (!iq && n_samples && !iq_staged) || (n_win && (!max_idx || (!max_val && !no_val)))) return LORA_HIP_EINVAL;
  if (iq_staged && (n_samples > h->staged_n || (flags & LORA_HIP_IQ_ON_DEVICE))) return LORA_HIP_EINVAL;
  if (n_win > 0xFFFFFFFFull / 2) return LORA_HIP_EINVAL;
  const bool iq_dev = flags & LORA_HIP_IQ_ON_DEVICE, out_dev = flags & LORA_HIP_OUT_ON_DEVICE;
  const bool desc_dev = flags & LORA_HIP_DESC_ON_DEVICE;
  const bool out_pinned = (flags & LORA_HIP_OUT_PINNED) && !out_dev;
  // without a synchronisation the caller's buffers must be such that the copies can run behind the call's return:
  // device memory, or page-locked host memory for the results
  if ((flags & LORA_HIP_ASYNC) && ((!iq_dev && !iq_staged) || (!out_dev && !out_pinned) || (desc && !desc_dev))) return LORA_HIP_EINVAL;
  const int lane = (flags & LORA_HIP_LANE1) ? 1 : 0;
  if (lane && !iq_dev && !iq_staged) return LORA_HIP_EINVAL;           // the staging copy belongs to lane 0's stream
  if (n_win == 0) return LORA_HIP_OK;
  if (n_samples < h->n) return LORA_HIP_EINVAL;      // the kernel redirects bad windows to sample 0
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  const hipStream_t stream = lane ? h->aux_stream : h->stream;
  void *&d_desc = lane ? h->d_desc1 : h->d_desc;
  size_t &cap_desc = lane ? h->cap_desc1 : h->cap_desc;
  unsigned char *&d_out = lane ? h->d_out1 : h->d_out;
  unsigned char *&h_out = lane ? h->h_out1 : h->h_out;
  size_t &cap_out = lane ? h->cap_out1 : h->cap_out;
  int rc;
  // inputs
  if (iq_dev) p.iq = reinterpret_cast<const float2 *>(iq);
  else if (iq_staged) p.iq = h->d_iq;
  else {
    h->staged_n = 0;
    if ((rc = grow(h, &h->d_iq, &h->cap_iq, n_samples)) != LORA_HIP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_iq, iq, n_samples * sizeof(float2), hipMemcpyHostToDevice, stream));
    p.iq = h->d_iq;
  }
  p.n_samples = n_samples;
  if (desc) {
    if (desc_dev) p.desc = desc;
    else {
      // host descriptors: validate here so that a bad window is an error, not a sentinel
      for (size_t i = 0; i < n_win; i++) {
        if (desc[i].kind > LORA_WIN_SFD || desc[i].chirp_off >= h->n) return LORA_HIP_EINVAL;
        if (desc[i].pos > n_samples || n_samples - desc[i].pos < h->n) return LORA_HIP_EINVAL;
      }
      size_t bytes = n_win * sizeof(lora_win_desc);
      if ((rc = grow(h, reinterpret_cast<unsigned char **>(&d_desc), &cap_desc, bytes)) != LORA_HIP_OK) return rc;
      HIP_TRY(h, hipMemcpyAsync(d_desc, desc, bytes, hipMemcpyHostToDevice, stream));
      p.desc = d_desc;
    }
  } else p.desc = nullptr;
  // outputs
  // LORA_HIP_OUT_DIRECT: the kernel stores its results into the caller's page-locked arrays itself (they are mapped into the
  // device's address space; the stores are visible to the host once the lane has been synchronised): two copy operations
  // fewer behind a small launch, for which their dispatch, not their size, is what takes time
  const bool out_direct = out_pinned && (flags & LORA_HIP_OUT_DIRECT) && !no_val;
  bool mirror_direct = false;
  if (out_dev || out_direct) { p.out_idx = max_idx; p.out_val = max_val; }
  else {
    if (n_win > cap_out) {
      // (hipFree waits for the device: a launch still reading the old block on this lane is done before it goes)
      const size_t ncap = n_win + n_win / 4 + 64;
      if (d_out) { (void)hipFree(d_out); d_out = nullptr; }
      if (h_out) { (void)hipHostFree(h_out); h_out = nullptr; }
      cap_out = 0;
      hipError_t e = hipMalloc((void **)&d_out, ncap * 6);
      if (e != hipSuccess) { d_out = nullptr; hip_fail(h, e, "hipMalloc"); return LORA_HIP_ENOMEM; }
      e = hipHostMalloc((void **)&h_out, ncap * 6, hipHostMallocDefault);
      if (e != hipSuccess) { h_out = nullptr; hip_fail(h, e, "hipHostMalloc"); return LORA_HIP_ENOMEM; }
      cap_out = ncap;
    }
    p.out_val = reinterpret_cast<float *>(d_out);
    p.out_idx = reinterpret_cast<unsigned short *>(d_out + n_win * 4);     // packed behind this call's values
    // a small synchronous call with pageable results (the block's on-demand launches): the kernel stores straight into the
    // handle's page-locked mirror, which saves the copy operation between the kernel and the synchronisation
    mirror_direct = !out_pinned && !(flags & LORA_HIP_ASYNC) && n_win <= kMirrorDirectMax && !LORA_NO_MIRROR_DIRECT;
    if (mirror_direct) {
      p.out_val = reinterpret_cast<float *>(h_out);
      p.out_idx = reinterpret_cast<unsigned short *>(h_out + n_win * 4);
    }
  }
  p.up = h->d_up; p.down = h->d_down; p.win = h->d_win;
  for (int ps = 0; ps < 4; ps++) p.twp[ps] = h->d_tw + h->tw_off[ps];
  p.n_win = (unsigned)n_win;
  if ((rc = launch(h, p, true, lane)) != LORA_HIP_OK) return rc;
  if (out_direct) {
    if (!(flags & LORA_HIP_ASYNC)) HIP_TRY(h, hipStreamSynchronize(stream));
    return LORA_HIP_OK;
  }
  if (!out_dev) {
    const size_t idx_off = n_win * 4;
    if (out_pinned) {
      // (an asynchronous call leaves the lane's device block busy until these copies have run: calls on one lane are
      // ordered, so the next launch on it writes the block only behind them)
      if (!no_val) HIP_TRY(h, hipMemcpyAsync(max_val, d_out, n_win * 4, hipMemcpyDeviceToHost, stream));
      HIP_TRY(h, hipMemcpyAsync(max_idx, d_out + idx_off, n_win * 2, hipMemcpyDeviceToHost, stream));
      if (!(flags & LORA_HIP_ASYNC)) HIP_TRY(h, hipStreamSynchronize(stream));
      return LORA_HIP_OK;
    }
    if (mirror_direct) {}
    else if (no_val) HIP_TRY(h, hipMemcpyAsync(h_out + idx_off, d_out + idx_off, n_win * 2, hipMemcpyDeviceToHost, stream));
    else HIP_TRY(h, hipMemcpyAsync(h_out, d_out, n_win * 6, hipMemcpyDeviceToHost, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    if (!no_val) memcpy(max_val, h_out, n_win * 4);
    memcpy(max_idx, h_out + idx_off, n_win * 2);
    return LORA_HIP_OK;
  }
  if (!(flags & LORA_HIP_ASYNC)) HIP_TRY(h, hipStreamSynchronize(stream));
  return LORA_HIP_OK;
}

int lora_hip_demod_batch(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples, const lora_win_desc *desc,
                         size_t n_win, uint16_t *max_idx, float *max_val, unsigned flags) {
  if (n_win && !desc) return LORA_HIP_EINVAL;
  KParams p;
  memset(&p, 0, sizeof p);
  return run_batch(h, iq, n_samples, p, desc, n_win, max_idx, max_val, flags);
}

int lora_hip_demod_batch_strided(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples, uint64_t pos0,
                                 uint64_t stride, size_t count, unsigned chirp_off, unsigned chirp_off_step,
                                 unsigned kind, uint16_t *max_idx, float *max_val, unsigned flags) {
  if (!h) return LORA_HIP_EINVAL;
  if (kind > LORA_WIN_SFD || chirp_off >= h->n || chirp_off_step >= h->n) return LORA_HIP_EINVAL;
  if (count) {
    // last window must end inside the capture (overflow-safe)
    if (pos0 > n_samples || n_samples - pos0 < h->n) return LORA_HIP_EINVAL;
    const uint64_t room = n_samples - pos0 - h->n;
    if (count > 1 && (stride == 0 ? false : (count - 1) > room / stride)) return LORA_HIP_EINVAL;
  }
  KParams p;
  memset(&p, 0, sizeof p);
  p.pos0 = pos0; p.stride = stride; p.off0 = chirp_off; p.off_step = chirp_off_step; p.kind = kind;
  return run_batch(h, iq, n_samples, p, nullptr, count, max_idx, max_val, flags);
}

int lora_hip_demod_batch_runs(lora_hip_demod_t *h, const lora_c32 *iq, size_t n_samples, const lora_win_run *runs,
                              size_t n_runs, uint16_t *max_idx, float *max_val, unsigned flags) {
  if (!h || (n_runs && !runs)) return LORA_HIP_EINVAL;
  if (flags & LORA_HIP_DESC_ON_DEVICE) return LORA_HIP_EINVAL;   // the run list is host memory
  if (n_runs > 0x7FFFFFFFull) return LORA_HIP_EINVAL;
  const int lane = (flags & LORA_HIP_LANE1) ? 1 : 0;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  const hipStream_t stream = lane ? h->aux_stream : h->stream;
  // The run list goes into page-locked memory of the handle and from there to HBM with an asynchronous copy (it used to be
  // copied from a pageable vector, which cost a stream synchronisation per call).  The lane's previous copy must be done
  // with that memory before it is overwritten.
  const unsigned tb = h->runs_turn[lane]++ & 1u;
  if (h->runs_pending[lane][tb]) { HIP_TRY(h, hipEventSynchronize(h->runs_done[lane][tb])); h->runs_pending[lane][tb] = false; }
  if (n_runs * sizeof(RunDev) > h->cap_hruns[lane][tb]) {
    if (h->h_runs[lane][tb]) { (void)hipHostFree(h->h_runs[lane][tb]); h->h_runs[lane][tb] = nullptr; h->cap_hruns[lane][tb] = 0; }
    const size_t ncap = (n_runs + n_runs / 4 + 64) * sizeof(RunDev);
    hipError_t e = hipHostMalloc((void **)&h->h_runs[lane][tb], ncap, hipHostMallocDefault);
    if (e != hipSuccess) { h->h_runs[lane][tb] = nullptr; hip_fail(h, e, "hipHostMalloc"); return LORA_HIP_ENOMEM; }
    h->cap_hruns[lane][tb] = ncap;
  }
  RunDev *rd = reinterpret_cast<RunDev *>(h->h_runs[lane][tb]);
  size_t n_rd = 0;
  uint64_t total = 0;
  for (size_t r = 0; r < n_runs; r++) {
    const lora_win_run &u = runs[r];
    if (u.kind > LORA_WIN_SFD || u.chirp_off >= h->n || u.chirp_off_step >= h->n) return LORA_HIP_EINVAL;
    if (u.count == 0) continue;
    if (u.pos0 > n_samples || n_samples - u.pos0 < h->n) return LORA_HIP_EINVAL;
    if (u.count > 1 && u.stride != 0 && (uint64_t)(u.count - 1) > (n_samples - u.pos0 - h->n) / u.stride) return LORA_HIP_EINVAL;
    rd[n_rd++] = RunDev{u.pos0, u.stride, (unsigned)total, u.chirp_off, u.chirp_off_step, u.kind};
    total += u.count;
    if (total > 0xFFFFFFFFull / 2) return LORA_HIP_EINVAL;
  }
  if (total == 0) return LORA_HIP_OK;
  void *&d_desc = lane ? h->d_desc1 : h->d_desc;
  size_t &cap_desc = lane ? h->cap_desc1 : h->cap_desc;
  unsigned char *&d_runs = lane ? h->d_runs1 : h->d_runs;
  size_t &cap_runs = lane ? h->cap_runs1 : h->cap_runs;
  int rc;
  if ((rc = grow(h, &d_runs, &cap_runs, n_rd * sizeof(RunDev))) != LORA_HIP_OK) return rc;
  if ((rc = grow(h, reinterpret_cast<unsigned char **>(&d_desc), &cap_desc, (size_t)total * sizeof(lora_win_desc))) != LORA_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpyAsync(d_runs, rd, n_rd * sizeof(RunDev), hipMemcpyHostToDevice, stream));
  HIP_TRY(h, hipEventRecord(h->runs_done[lane][tb], stream));
  h->runs_pending[lane][tb] = true;
  hipLaunchKernelGGL(expand_runs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const RunDev *>(d_runs), (unsigned)n_rd, (unsigned)total, h->n - 1,
                     reinterpret_cast<lora_hip::WinDesc *>(d_desc));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, e, "expand_runs launch");
  KParams p;
  memset(&p, 0, sizeof p);
  return run_batch(h, iq, n_samples, p, reinterpret_cast<const lora_win_desc *>(d_desc), (size_t)total, max_idx, max_val,
                   flags | LORA_HIP_DESC_ON_DEVICE);
}

int lora_hip_demod_sync(lora_hip_demod_t *h, int lane) {
  if (!h || lane < 0 || lane > 1) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  HIP_TRY(h, hipStreamSynchronize(lane ? h->aux_stream : h->stream));
  return LORA_HIP_OK;
}

int lora_hip_demod_mark(lora_hip_demod_t *h, int lane, int mark) {
  if (!h || lane < 0 || lane > 1 || mark < 0 || mark >= LORA_HIP_MARKS) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  HIP_TRY(h, hipEventRecord(h->marks[mark], lane ? h->aux_stream : h->stream));
  return LORA_HIP_OK;
}

int lora_hip_demod_wait(lora_hip_demod_t *h, int mark) {
  if (!h || mark < 0 || mark >= LORA_HIP_MARKS) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  HIP_TRY(h, hipEventSynchronize(h->marks[mark]));       // a mark never set counts as reached
  return LORA_HIP_OK;
}

int lora_hip_host_alloc(void **p, size_t bytes) {
  if (!p) return LORA_HIP_EINVAL;
  *p = nullptr;
  hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) { *p = nullptr; return LORA_HIP_ENOMEM; }
  return LORA_HIP_OK;
}
void lora_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

int lora_hip_demod_batch_strided_noalloc(lora_hip_demod_t *h, const lora_c32 *iq_dev, size_t n_samples, uint64_t pos0,
                                         uint64_t stride, size_t count, unsigned chirp_off, unsigned chirp_off_step,
                                         unsigned kind, uint16_t *max_idx_dev, float *max_val_dev) {
  if (!h) return LORA_HIP_EINVAL;
  if (count > LORA_HIP_NOALLOC_MAX_WINDOWS) {
    snprintf(h->hip_err, sizeof h->hip_err, "lora_hip_demod_batch_strided_noalloc: %zu windows exceed the maximum of %llu per call",
             count, (unsigned long long)LORA_HIP_NOALLOC_MAX_WINDOWS);
    return LORA_HIP_EINVAL;
  }
  if (kind > LORA_WIN_SFD || chirp_off >= h->n || chirp_off_step >= h->n) return LORA_HIP_EINVAL;
  if (count == 0) return LORA_HIP_OK;
  if (!iq_dev || !max_idx_dev || !max_val_dev || n_samples < h->n) return LORA_HIP_EINVAL;
  if (pos0 > n_samples || n_samples - pos0 < h->n) return LORA_HIP_EINVAL;
  if (count > 1 && stride != 0 && (count - 1) > (n_samples - pos0 - h->n) / stride) return LORA_HIP_EINVAL;
  if (h->device >= kMaxDevices || !g_scratch_ptr[h->device]) return LORA_HIP_ENODEVICE;   // set by create() on this device
  if (h->scratch_slot < 0) {
    snprintf(h->hip_err, sizeof h->hip_err, "lora_hip_demod_batch_strided_noalloc: all %d scratch slots of GPU %d were taken when this handle was created",
             kScratchSlots, h->device);
    return LORA_HIP_ENOMEM;
  }
  DeviceGuard guard(h->device);
  HIP_TRY(h, guard.err);
  KParams p;
  memset(&p, 0, sizeof p);
  p.pos0 = pos0; p.stride = stride; p.off0 = chirp_off; p.off_step = chirp_off_step; p.kind = kind;
  p.iq = reinterpret_cast<const float2 *>(iq_dev);
  p.n_samples = n_samples;
  p.desc = nullptr;
  p.out_idx = max_idx_dev; p.out_val = max_val_dev;
  p.up = h->d_up; p.down = h->d_down; p.win = h->d_win;
  for (int ps = 0; ps < 4; ps++) p.twp[ps] = h->d_tw + h->tw_off[ps];
  p.n_win = (unsigned)count;
  // enqueue only, on the handle's stream: the kernel (its last workgroup returns the counters to zero), in front of it a
  // memset of the handle's 4 KiB slot only after a failed launch; no allocation, no synchronisation
  return launch(h, p, false);
}

}  // extern "C"
