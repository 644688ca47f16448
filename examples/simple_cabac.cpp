// simple_cabac.cpp -- the smallest complete use of the C ABI (include/isscabac.h) from a compiled program: one bitstream
// through the per-bin handle, then the same bins through the batch call, both checked against a known answer.
//
// The op sequence is the one the reference's stand-alone driver codes (ref: CABAC/SimpleCABAC.cpp:63-103): six bins on a
// context initialised to (mps 0, state 20), five bypass bins one by one, the same five as one multi-bin bypass write, six
// bins on a default context (mps 1, state 0).  The reference writes the four bytes AB B1 2F 09 for it (pinned by the
// compiled reference: tests/golden/cabac_golden.json, ops[0]); decoding them gives the bins back and a clean status word.
//
//   g++ -O2 -std=c++17 -I include examples/simple_cabac.cpp -o /tmp/simple_cabac -L isscabac_amd/lib -lisscabac_hip
//       -Wl,-rpath,$PWD/isscabac_amd/lib -Wl,--allow-shlib-undefined && /tmp/simple_cabac
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "isscabac.h"

static int fail(isscabac_engine *e, const char *what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, isscabac_last_error(e));
  return 1;
}
#define TRY(call)                                  \
  do {                                             \
    const int rc_ = (call);                        \
    if (rc_ != ISSCABAC_OK) return fail(e, #call, rc_); \
  } while (0)

int main() {
  isscabac_engine *e = nullptr;
  if (const int rc = isscabac_create(0, &e)) return fail(nullptr, "isscabac_create", rc);

  // packed initial states: mps | state << 1 (isscabac_init_by_state takes the reference's [ctxIdx, mps, state] triples)
  const double triples[6] = {0, 0, 20, 1, 1, 0};
  uint8_t init[2];
  TRY(isscabac_init_by_state(triples, 2, init));

  // bins and the contexts they are coded with; 0xFF = a bypass bin
  const uint8_t bins[] = {0, 0, 1, 0, 1, 1, /* bypass */ 1, 0, 0, 1, 0, /* 18 in five bits */ 1, 0, 0, 1, 0, 1, 1, 0, 1, 1, 1};
  const uint8_t ctx[] = {0, 0, 0, 0, 0, 0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 1, 1, 1, 1, 1, 1};
  const uint32_t n = sizeof bins;
  const uint8_t expect[4] = {0xAB, 0xB1, 0x2F, 0x09};

  // 1. the per-bin handle, one bin per call like the reference's MATLAB loops
  isscabac_stream *s = nullptr;
  TRY(isscabac_stream_open(e, init, 2, &s));
  TRY(isscabac_stream_encode_start(s));
  for (uint32_t i = 0; i < n; i++) TRY(isscabac_stream_encode_bins(s, &bins[i], &ctx[i], 1));
  uint64_t bits = 0;
  TRY(isscabac_stream_num_bits(s, &bits));
  uint8_t out[64];
  uint64_t nbytes = 0;
  TRY(isscabac_stream_encode_finish(s, out, sizeof out, &nbytes));
  printf("handle: %llu bits before finish, %llu bytes:", (unsigned long long)bits, (unsigned long long)nbytes);
  for (uint64_t i = 0; i < nbytes; i++) printf(" %02X", out[i]);
  printf("\n");
  int bad = nbytes != 4 || memcmp(out, expect, 4) != 0;
  TRY(isscabac_stream_decode_start(s, out, nbytes));
  for (uint32_t i = 0; i < n; i++) {
    uint8_t b = 2;
    TRY(isscabac_stream_decode_bins(s, &ctx[i], 1, &b));
    bad |= b != bins[i];
  }
  uint32_t status = 1;
  TRY(isscabac_stream_decode_finish(s, &status));
  bad |= status != 0;
  isscabac_stream_close(s);

  // 2. the batch call: the same sequence as 70 independent streams (rows of a [S][pitch] array; a MATLAB B x S matrix)
  const uint64_t S = 70;
  const uint32_t pitch = 32;
  std::vector<uint8_t> bb(S * pitch, 0), cc(S * pitch, 0), dec(S * pitch, 9);
  for (uint64_t r = 0; r < S; r++) {
    memcpy(&bb[r * pitch], bins, n);
    memcpy(&cc[r * pitch], ctx, n);
  }
  std::vector<uint8_t> payload(S * isscabac_stream_bound(n));
  std::vector<uint64_t> offsets(S + 1);
  std::vector<uint32_t> st(S, 1);
  uint64_t total = 0;
  TRY(isscabac_encode_batch(e, bb.data(), cc.data(), nullptr, S, n, pitch, init, 2, payload.data(), payload.size(), offsets.data(), &total));
  bad |= total != 4 * S;
  for (uint64_t r = 0; r < S && !bad; r++) bad |= offsets[r] != 4 * r || memcmp(&payload[4 * r], expect, 4) != 0;
  TRY(isscabac_decode_batch(e, payload.data(), offsets.data(), cc.data(), nullptr, S, n, pitch, init, 2, dec.data(), st.data()));
  for (uint64_t r = 0; r < S; r++) bad |= st[r] != 0 || memcmp(&dec[r * pitch], bins, n) != 0;
  printf("batch:  %llu streams, %llu bytes, every stream %s\n", (unsigned long long)S, (unsigned long long)total, bad ? "WRONG" : "AB B1 2F 09 and back");
  isscabac_destroy(e);
  return bad;
}
