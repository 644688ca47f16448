// cabac_container.cpp -- the host edge of the engine: where bitstreams meet the file system.
//
// The reference hands an encoder's output to the decoder through ONE FILE PER STREAM (ref: CABAC/SimpleCABACMex.cpp:195
// opens the handle's file for writing, :288 for reading) and its callers size a result by `dir(fn).bytes * 8`
// (ref: CABAC/cabacDemo.m:190-191, ISS/+coder/cabacEncode.m:100-101).  The engine keeps a batch as payload + offsets;
// these calls convert between the two forms, so that files written here are read by the reference's decoder and the
// other way round (tests/test_abi_symbols.py::test_stream_files_interoperate_with_the_reference_decoder and
// tests/test_mex_gateway.py do both with oracle/_ref).
// The second half is the side information ISS.m transmits next to its two CABAC streams (ref: ISS/ISS.m:197-201:
// centroids cW, cH, the mixing matrix Q and, unless equalProb, the uint8 context initialisations xw, xh).  The reference
// only measures its size through MATLAB's `save`; here it gets a concrete little-endian container, written and read back
// together with the two streams.
// Host-only: no GPU work in this file.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/isscabac.h"
#include "cabac_internal.h"

using isscabac::engine_fail;

namespace {

struct File {
  FILE *f;
  explicit File(const char *path, const char *mode) : f(path ? fopen(path, mode) : nullptr) {}
  ~File() {
    if (f) fclose(f);
  }
  // for files written: flush and close now, and say whether everything reached the file (a full disk shows up here)
  bool close_ok() {
    if (!f) return false;
    const bool ok = fflush(f) == 0 && !ferror(f);
    const bool closed = fclose(f) == 0;
    f = nullptr;
    return ok && closed;
  }
};

const char kMagic[8] = {'I', 'S', 'S', 'C', 'A', 'B', 'A', 'C'};
const uint32_t kVersion = 1;

bool put(FILE *f, const void *p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
bool get(FILE *f, void *p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }

}  // namespace

extern "C" {

int isscabac_stream_file_path(const char *prefix, uint64_t s, char *buf, uint64_t cap) {
  if (!prefix || !buf) return ISSCABAC_ERR_ARG;
  const int n = snprintf(buf, (size_t)cap, "%s%06llu.bin", prefix, (unsigned long long)s);
  return n > 0 && (uint64_t)n < cap ? ISSCABAC_OK : engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_stream_file_path: buffer too small");
}

int isscabac_write_stream_files(const uint8_t *payload, const uint64_t *offsets, uint64_t S, const char *const *paths) {
  if ((!payload && S && offsets && offsets[S]) || !offsets || !paths)
    return engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_write_stream_files: null argument");
  for (uint64_t s = 0; s < S; s++) {
    if (offsets[s + 1] < offsets[s]) return engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_write_stream_files: offsets not monotone at stream %llu", (unsigned long long)s);
    File f(paths[s], "wb");
    if (!f.f || !put(f.f, payload + offsets[s], (size_t)(offsets[s + 1] - offsets[s])) || !f.close_ok())
      return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_write_stream_files: cannot write '%s'", paths[s] ? paths[s] : "(null)");
  }
  return ISSCABAC_OK;
}

int isscabac_read_stream_files(const char *const *paths, uint64_t S, uint8_t *payload, uint64_t cap, uint64_t *offsets,
                               uint64_t *total_bytes) {
  if (!paths || !offsets) return engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_read_stream_files: null argument");
  uint64_t off = 0;
  bool fits = true;  // (a null payload is fine as long as no byte has to go there: S == 0, or nothing but empty files)
  for (uint64_t s = 0; s < S; s++) {
    File f(paths[s], "rb");
    if (!f.f || fseek(f.f, 0, SEEK_END) != 0) return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_read_stream_files: cannot read '%s'", paths[s] ? paths[s] : "(null)");
    const long len = ftell(f.f);
    if (len < 0 || fseek(f.f, 0, SEEK_SET) != 0) return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_read_stream_files: cannot read '%s'", paths[s]);
    offsets[s] = off;
    if (len == 0) {
    } else if (fits && payload && off + (uint64_t)len <= cap) {
      if (!get(f.f, payload + off, (size_t)len)) return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_read_stream_files: short read on '%s'", paths[s]);
    } else {
      fits = false;
    }
    off += (uint64_t)len;
  }
  offsets[S] = off;
  if (total_bytes) *total_bytes = off;
  return fits ? ISSCABAC_OK
              : engine_fail(nullptr, ISSCABAC_ERR_CAPACITY, "isscabac_read_stream_files: the files hold %llu bytes, the buffer %llu",
                            (unsigned long long)off, (unsigned long long)cap);
}

// layout: magic[8] version:u32 ncW:u32 ncH:u32 q_rows:u32 q_cols:u32 nxw:u32 nxh:u32 reserved:u32 w_bytes:u64 h_bytes:u64
//         cW:f64[ncW] cH:f64[ncH] Q:f64[q_rows*q_cols] (column-major, as MATLAB holds it) xw:u8[nxw] xh:u8[nxh] W[w_bytes] H[h_bytes]
int isscabac_iss_container_write(const char *path, const isscabac_iss_side_info *si, const uint8_t *w_stream, uint64_t w_bytes,
                                 const uint8_t *h_stream, uint64_t h_bytes) {
  if (!path || !si || (!w_stream && w_bytes) || (!h_stream && h_bytes) || (!si->cW && si->ncW) || (!si->cH && si->ncH) ||
      (!si->Q && si->q_rows * si->q_cols) || (!si->xw && si->nxw) || (!si->xh && si->nxh))
    return engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_iss_container_write: null argument");
  File f(path, "wb");
  const uint32_t head[8] = {kVersion, si->ncW, si->ncH, si->q_rows, si->q_cols, si->nxw, si->nxh, 0};
  const uint64_t lens[2] = {w_bytes, h_bytes};
  if (!f.f || !put(f.f, kMagic, 8) || !put(f.f, head, sizeof head) || !put(f.f, lens, sizeof lens) ||
      !put(f.f, si->cW, 8ull * si->ncW) || !put(f.f, si->cH, 8ull * si->ncH) ||
      !put(f.f, si->Q, 8ull * si->q_rows * si->q_cols) || !put(f.f, si->xw, si->nxw) || !put(f.f, si->xh, si->nxh) ||
      !put(f.f, w_stream, (size_t)w_bytes) || !put(f.f, h_stream, (size_t)h_bytes) || !f.close_ok())
    return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_iss_container_write: cannot write '%s'", path);
  return ISSCABAC_OK;
}

int isscabac_iss_container_sizes(const char *path, isscabac_iss_container_info *out) {
  if (!path || !out) return engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_iss_container_sizes: null argument");
  File f(path, "rb");
  char magic[8];
  uint32_t head[8];
  uint64_t lens[2];
  if (!f.f || !get(f.f, magic, 8) || !get(f.f, head, sizeof head) || !get(f.f, lens, sizeof lens))
    return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_iss_container_sizes: cannot read '%s'", path);
  if (memcmp(magic, kMagic, 8) != 0 || head[0] != kVersion)
    return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_iss_container_sizes: '%s' is not an ISS container of version %u", path, kVersion);
  out->ncW = head[1];
  out->ncH = head[2];
  out->q_rows = head[3];
  out->q_cols = head[4];
  out->nxw = head[5];
  out->nxh = head[6];
  out->w_bytes = lens[0];
  out->h_bytes = lens[1];
  return ISSCABAC_OK;
}

int isscabac_iss_container_read(const char *path, double *cW, double *cH, double *Q, uint8_t *xw, uint8_t *xh, uint8_t *w_stream,
                                uint8_t *h_stream) {
  isscabac_iss_container_info n;
  const int rc = isscabac_iss_container_sizes(path, &n);
  if (rc) return rc;
  if ((!cW && n.ncW) || (!cH && n.ncH) || (!Q && n.q_rows * n.q_cols) || (!xw && n.nxw) || (!xh && n.nxh) ||
      (!w_stream && n.w_bytes) || (!h_stream && n.h_bytes))
    return engine_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_iss_container_read: null buffer for a non-empty field");
  File f(path, "rb");
  if (!f.f || fseek(f.f, 8 + 32 + 16, SEEK_SET) != 0 || !get(f.f, cW, 8ull * n.ncW) || !get(f.f, cH, 8ull * n.ncH) ||
      !get(f.f, Q, 8ull * n.q_rows * n.q_cols) || !get(f.f, xw, n.nxw) || !get(f.f, xh, n.nxh) ||
      !get(f.f, w_stream, (size_t)n.w_bytes) || !get(f.f, h_stream, (size_t)n.h_bytes))
    return engine_fail(nullptr, ISSCABAC_ERR_IO, "isscabac_iss_container_read: '%s' is truncated", path);
  return ISSCABAC_OK;
}

}  // extern "C"
