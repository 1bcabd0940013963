// Host code under AddressSanitizer: a plain C++ program (so that the sanitizer's runtime is the program's own, nothing is
// preloaded) that drives the block through the C ABI the way the Python mirror does -- whole input, scheduler-sized chunks,
// ragged chunks; plan modes 0 / 1 / 2; short stretches (LORA_CORE_STRETCH in the environment); reset and reuse; trace and
// collected PDUs -- and checks that every way of feeding gives the same PDUs.  GPU code is not instrumented (the pool has no
// GPU sanitizers).  Built and run by tools/asan_run.sh.
//   asan_driver <capture.c64> <sf> <fft_factor>
#include "lora_demod_core.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
struct Sink { std::vector<uint16_t> syms; std::vector<size_t> ends; };
void on_pdu(void *user, const uint16_t *s, size_t n) {
  Sink *k = static_cast<Sink *>(user);
  k->syms.insert(k->syms.end(), s, s + n);
  k->ends.push_back(k->syms.size());
}
int feed(lora_demod_core_t *c, const std::vector<lora_c32> &cap, size_t chunk, unsigned ragged, Sink *out) {
  size_t at = 0;
  unsigned lcg = 12345u + ragged;
  while (at < cap.size()) {
    size_t n = chunk ? chunk : cap.size();
    if (ragged) { lcg = lcg * 1664525u + 1013904223u; n = 1 + (lcg >> 8) % (2 * chunk); }
    if (n > cap.size() - at) n = cap.size() - at;
    const int rc = lora_demod_core_work(c, cap.data() + at, n, on_pdu, out);
    if (rc) { fprintf(stderr, "work: %s (%s)\n", lora_hip_strerror(rc), lora_demod_core_last_error(c)); return rc; }
    at += n;
  }
  return 0;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s capture.c64 sf fft_factor\n", argv[0]); return 2; }
  const unsigned sf = (unsigned)atoi(argv[2]), F = (unsigned)atoi(argv[3]), N = 1u << sf;
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  std::vector<lora_c32> cap;
  lora_c32 buf[4096];
  size_t got;
  while ((got = fread(buf, sizeof(lora_c32), 4096, f)) > 0) cap.insert(cap.end(), buf, buf + got);
  fclose(f);
  cap.resize(cap.size() + 4 * N, lora_c32{0.f, 0.f});          // flush the last packet
  lora_demod_core_t *c = nullptr;
  int rc = lora_demod_core_create(&c, 0, sf, 0, 25.0f, F);
  if (rc) { fprintf(stderr, "create: %s\n", lora_hip_strerror(rc)); return 1; }
  std::vector<lora_core_trace> rows(1 << 20);
  Sink ref;
  int bad = 0;
  struct Way { int plan; size_t chunk; unsigned ragged; const char *name; };
  const Way ways[] = {
    {2, 0, 0, "planned, whole input"},          {0, 0, 0, "on demand, whole input"},
    {1, 16 * N, 0, "default, 16-symbol chunks"}, {2, 300 * N, 0, "planned, 300-symbol chunks"},
    {1, 256 * N, 0, "default, 256-symbol chunks (dense streams switch to plans)"},
    {2, 40 * N, 7, "planned, ragged chunks"},    {0, 3 * N, 3, "on demand, ragged short chunks"},
  };
  for (const Way &w : ways) {
    if ((rc = lora_demod_core_reset(c)) || (rc = lora_demod_core_set_plan(c, w.plan)) ||
        (rc = lora_demod_core_set_trace(c, rows.data(), rows.size())) || (rc = lora_demod_core_set_collect(c, w.chunk == 0))) {
      fprintf(stderr, "setup: %s\n", lora_hip_strerror(rc)); return 1;
    }
    Sink s;
    if (feed(c, cap, w.chunk, w.ragged, &s)) return 1;
    if (w.chunk == 0) {                                            // the block's own list must say the same as the callback
      const uint16_t *cs = nullptr; const uint64_t *co = nullptr; size_t np = 0;
      if (lora_demod_core_collected(c, &cs, &co, &np) || np != s.ends.size()) { fprintf(stderr, "collected: %zu PDUs against %zu\n", np, s.ends.size()); bad++; }
      else for (size_t i = 0; i < np; i++) if (co[i + 1] != s.ends[i]) { bad++; break; }
    }
    const lora_core_idle *runs = nullptr; size_t n_runs = 0;
    (void)lora_demod_core_idle_log(c, &runs, &n_runs);
    printf("%-62s %5zu PDUs, %8llu calls, %6llu launches, %4zu idle runs, %llu plans, %llu on demand\n", w.name, s.ends.size(),
           (unsigned long long)lora_demod_core_calls(c), (unsigned long long)lora_demod_core_launches(c), n_runs,
           (unsigned long long)lora_demod_core_plans(c), (unsigned long long)lora_demod_core_on_demand(c));
    if (&w == &ways[0]) ref = s;
    else if (s.syms != ref.syms || s.ends != ref.ends) { fprintf(stderr, "MISMATCH: %s\n", w.name); bad++; }
  }
  // a block started in the middle of the stream (sharded replay)
  if ((rc = lora_demod_core_seek(c, (uint64_t)(cap.size() / 2 / (N / 16)) * (N / 16))) == 0) {
    Sink s;
    (void)lora_demod_core_set_plan(c, 2);
    rc = lora_demod_core_work(c, cap.data() + cap.size() / 2, cap.size() - cap.size() / 2, on_pdu, &s);
    printf("%-62s %5zu PDUs (rc %d)\n", "seek to the middle, planned", s.ends.size(), rc);
    if (rc) bad++;
  } else { fprintf(stderr, "seek: %s\n", lora_hip_strerror(rc)); bad++; }
  lora_demod_core_destroy(c);
  if (ref.ends.empty()) { fprintf(stderr, "no PDUs at all\n"); bad++; }
  printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
