// per_bin_native.cpp -- the per-bin commands timed from C++, without an interpreter between the calls: what a compiled
// caller (the MEX gateway's mexFunction) pays per encodeBin / getNumBits / decodeBin (ref: CABAC/SimpleCABACMex.cpp:210-333,
// one call per bin from cabacDemo.m:92-131 and +coder/cabacEncode.m:45-70).  tools/per_bin_rate.py measures the same calls
// through ctypes, where the interpreter's own 4-5 us per call hide the queued encodeBin entirely.
//   g++ -O2 -std=c++17 -I include tests/perf/per_bin_native.cpp -o tests/_build/per_bin_native -L isscabac_amd/lib -lisscabac_hip
//       -Wl,-rpath,'$ORIGIN/../../isscabac_amd/lib'
//   tests/_build/per_bin_native [bins]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "isscabac.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x)                                                                                    \
  do {                                                                                              \
    const int rc_ = (x);                                                                            \
    if (rc_) {                                                                                      \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, e ? isscabac_last_error(e) : "");                 \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)

// the same three loops on the host twin of the handle (isscabac_cpu_stream_*: no engine, no GPU)
static int host_twin(const std::vector<uint8_t> &bins, const std::vector<uint8_t> &ctx, uint32_t n) {
  isscabac_engine *e = nullptr;  // (CHECK prints its error text)
  const uint8_t init[3] = {0, 0, 0};
  isscabac_cpu_stream *s = nullptr, *d = nullptr;
  CHECK(isscabac_cpu_stream_open(init, 3, &s));
  CHECK(isscabac_cpu_stream_encode_start(s));
  const double t0 = now();
  for (uint32_t i = 0; i < n; i++) CHECK(isscabac_cpu_stream_encode_bins(s, &bins[i], &ctx[i], 1));
  const double t1 = now();
  std::vector<uint8_t> payload(n / 4 + 64);
  uint64_t nbytes = 0, bits = 0, sum = 0;
  CHECK(isscabac_cpu_stream_encode_finish(s, payload.data(), payload.size(), &nbytes));
  CHECK(isscabac_cpu_stream_encode_start(s));
  const double t2 = now();
  for (uint32_t i = 0; i < n; i++) {
    CHECK(isscabac_cpu_stream_encode_bins(s, &bins[i], &ctx[i], 1));
    CHECK(isscabac_cpu_stream_num_bits(s, &bits));
    sum += bits;
  }
  const double t3 = now();
  CHECK(isscabac_cpu_stream_open(init, 3, &d));
  CHECK(isscabac_cpu_stream_decode_start(d, payload.data(), nbytes));
  uint32_t wrong = 0;
  const double t4 = now();
  for (uint32_t i = 0; i < n; i++) {
    uint8_t b = 2;
    CHECK(isscabac_cpu_stream_decode_bins(d, &ctx[i], 1, &b));
    wrong += b != bins[i];
  }
  const double t5 = now();
  uint32_t status = 99;
  CHECK(isscabac_cpu_stream_decode_finish(d, &status));
  printf("the same on the host twin (isscabac_cpu_stream_*, one host core, %u bins): encodeBin %.1f Mbin/s (%.3f us per call), encodeBin + "
         "getNumBits %.1f Mbin/s (%.3f us per pair), decodeBin %.1f Mbin/s (%.3f us per call); %llu bytes, %u wrong bins, status %u (checksum %llu)\n",
         n, n / (t1 - t0) / 1e6, (t1 - t0) / n * 1e6, n / (t3 - t2) / 1e6, (t3 - t2) / n * 1e6, n / (t5 - t4) / 1e6, (t5 - t4) / n * 1e6,
         (unsigned long long)nbytes, wrong, status, (unsigned long long)sum);
  isscabac_cpu_stream_close(s);
  isscabac_cpu_stream_close(d);
  return wrong != 0 || status != 0;
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000u;
  const bool host_only = argc > 2 && argv[2][0] == 'h';  // `per_bin_native 200000 host`: no GPU needed
  isscabac_engine *e = nullptr;
  if (!host_only) CHECK(isscabac_create(0, &e));
  // BASELINE config 1's shape: one stream, three contexts with different skews (a seeded generator of this file)
  std::vector<uint8_t> bins(n), ctx(n);
  uint32_t x = 12345;
  for (uint32_t i = 0; i < n; i++) {
    x = x * 1103515245u + 12345u;
    ctx[i] = (uint8_t)((x >> 20) % 3u);
    const uint32_t thr = ctx[i] == 0 ? 13107u : ctx[i] == 1 ? 32768u : 52429u;  // P(1) = 0.2 / 0.5 / 0.8
    bins[i] = (uint8_t)(((x >> 4) & 0xFFFFu) < thr);
  }
  if (host_only) return host_twin(bins, ctx, n);
  const uint8_t init[3] = {0, 0, 0};
  isscabac_stream *s = nullptr, *d = nullptr;
  CHECK(isscabac_stream_open(e, init, 3, &s));
  CHECK(isscabac_stream_encode_start(s));
  const double t0 = now();
  for (uint32_t i = 0; i < n; i++) CHECK(isscabac_stream_encode_bins(s, &bins[i], &ctx[i], 1));
  uint64_t bits = 0;
  CHECK(isscabac_stream_num_bits(s, &bits));  // the queued bins are coded here at the latest
  const double t1 = now();
  const uint32_t np = n < 20000u ? n : 20000u;
  std::vector<uint8_t> payload(n / 4 + 64);
  uint64_t nbytes = 0;
  CHECK(isscabac_stream_encode_finish(s, payload.data(), payload.size(), &nbytes));
  // encodeBin + getNumBits after every bin (what the statistics loop of +coder/cabacEncode.m:60-67 does)
  CHECK(isscabac_stream_encode_start(s));
  const double t2 = now();
  for (uint32_t i = 0; i < np; i++) {
    CHECK(isscabac_stream_encode_bins(s, &bins[i], &ctx[i], 1));
    CHECK(isscabac_stream_num_bits(s, &bits));
  }
  const double t3 = now();
  std::vector<uint8_t> scratch(n / 4 + 64);
  uint64_t nb2 = 0;
  CHECK(isscabac_stream_encode_finish(s, scratch.data(), scratch.size(), &nb2));
  // decodeBin, one call per bin
  CHECK(isscabac_stream_open(e, init, 3, &d));
  CHECK(isscabac_stream_decode_start(d, payload.data(), nbytes));
  const uint32_t nd = n < 50000u ? n : 50000u;
  uint32_t wrong = 0;
  const double t4 = now();
  for (uint32_t i = 0; i < nd; i++) {
    uint8_t b = 2;
    CHECK(isscabac_stream_decode_bins(d, &ctx[i], 1, &b));
    wrong += b != bins[i];
  }
  const double t5 = now();
  // the rest in one call, then the integrity word
  if (n > nd) {
    std::vector<uint8_t> rest(n - nd);
    CHECK(isscabac_stream_decode_bins(d, &ctx[nd], n - nd, rest.data()));
    for (uint32_t i = nd; i < n; i++) wrong += rest[i - nd] != bins[i];
  }
  uint32_t status = 99;
  CHECK(isscabac_stream_decode_finish(d, &status));
  printf("per-bin commands from C++ (no interpreter), one stream, 3 contexts: encodeBin %.2f Mbin/s (%.3f us per call, queued; %u bins, "
         "%llu bytes), encodeBin + getNumBits %.1f kbin/s (%.2f us per pair, %u pairs), decodeBin %.1f kbin/s (%.2f us per call, %u "
         "calls); %u wrong bins, status %u\n",
         n / (t1 - t0) / 1e6, (t1 - t0) / n * 1e6, n, (unsigned long long)nbytes, np / (t3 - t2) / 1e3, (t3 - t2) / np * 1e6, np,
         nd / (t5 - t4) / 1e3, (t5 - t4) / nd * 1e6, nd, wrong, status);
  isscabac_stream_close(s);
  isscabac_stream_close(d);
  isscabac_destroy(e);
  return (wrong != 0 || status != 0) | host_twin(bins, ctx, n);
}
