// cabac_multi.hip -- several GPUs behind ONE handle in ONE process (SURVEY 8(b) "cabac_create(device_ids, n)", 8(e)).
//
// A MEX / FFI caller is a single process; torchrun-style one-process-per-GPU sharding (isscabac_amd/sharding.py, bench.py)
// is out of its reach.  isscabac_multi owns one engine per listed device and codes a batch the way the multi-process
// path does: streams are independent (ref: one `class CABAC` = one stream, CABAC/SimpleCABACMex.cpp:69-80), device g takes
// the contiguous block [S*g/n, S*(g+1)/n) of the stream range, nothing is exchanged while coding, and the only
// communication is the final bitstream gather -- here hipMemcpyPeerAsync from every device straight into the root
// device's payload buffer (all xGMI links of the root busy at once, no ring), or plain D2H copies into the caller's host
// buffer.  One host thread per device drives its engine, so the devices run concurrently.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cabac_internal.h"

struct isscabac_multi {
  std::vector<isscabac_engine *> eng;
  std::vector<int> dev;
  std::string err;
  // per-device staging for the device-resident form: shard payload + offsets before the gather
  std::vector<uint8_t *> d_pay;
  std::vector<uint64_t> pay_cap;
  std::vector<uint64_t *> d_off;
  std::vector<uint64_t> off_cap;
  // how device g reaches the root's memory for the gather: 0 = it IS the root's device, 1 = peer access (xGMI / PCIe P2P,
  // hipDeviceCanAccessPeer said yes and it is enabled), 2 = no peer access: hipMemcpyPeerAsync stages through the host
  std::vector<int> peer;
};

namespace isscabac {

__global__ void k_rebase_offsets(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] + base;
}

// the public entry points leave the caller's current HIP device as they found it (they, and the calling thread's share of
// on_all(), switch devices: an FFI or torch caller would otherwise find later raw hipMalloc / launches on the wrong GPU)
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceGuard() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};

static void shard_of(uint64_t S, int g, int n, uint64_t *lo, uint64_t *hi) {  // sharding.shard_range
  const uint64_t base = S / n, rem = S % n;
  *lo = g * base + ((uint64_t)g < rem ? (uint64_t)g : rem);
  *hi = *lo + base + ((uint64_t)g < rem ? 1 : 0);
}

static int multi_fail(isscabac_multi *m, int code, const std::string &msg) {
  if (m) m->err = msg;
  else engine_fail(nullptr, code, "%s", msg.c_str());
  return code;
}

// run fn(g) on one host thread per device; returns the first non-zero status (and that engine's error text)
template <class F>
static int on_all(isscabac_multi *m, F fn) {
  const int n = (int)m->eng.size();
  std::vector<int> rc(n, 0);
  std::vector<std::thread> th;
  for (int g = 1; g < n; g++) th.emplace_back([&, g] { rc[g] = fn(g); });
  rc[0] = fn(0);
  for (auto &t : th) t.join();
  for (int g = 0; g < n; g++)
    if (rc[g]) {
      m->err = std::string("device ") + std::to_string(m->dev[g]) + ": " + isscabac_last_error(m->eng[g]);
      return rc[g];
    }
  return ISSCABAC_OK;
}

}  // namespace isscabac

using namespace isscabac;

extern "C" {

int isscabac_create_multi(const int *device_ids, int n, isscabac_multi **out) {
  if (!out) return multi_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_create_multi: out is null");
  *out = nullptr;
  if (!device_ids || n < 1 || n > 64) return multi_fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_create_multi: 1..64 devices");
  DeviceGuard restore;  // first: isscabac_create() switches the device and never switches back; both exits below restore it
  isscabac_multi *m = new isscabac_multi;
  for (int g = 0; g < n; g++) {
    isscabac_engine *e = nullptr;
    const int rc = isscabac_create(device_ids[g], &e);
    if (rc) {
      isscabac_destroy_multi(m);  // (the error text of the failed create stays in isscabac_last_error(NULL))
      return rc;
    }
    m->eng.push_back(e);
    m->dev.push_back(device_ids[g]);
  }
  m->d_pay.assign(n, nullptr);
  m->pay_cap.assign(n, 0);
  m->d_off.assign(n, nullptr);
  m->off_cap.assign(n, 0);
  // peer access root <-> others where the hardware offers it (hipMemcpyPeerAsync works without, through the host)
  m->peer.assign(n, 0);
  const char *log = getenv("ISSCABAC_MULTI_LOG");  // set: one line per device pair on stderr (a first multi-GPU run explains itself)
  for (int g = 1; g < n; g++) {
    int can = 0;
    if (m->dev[g] == m->dev[0]) continue;
    m->peer[g] = 2;
    if (hipDeviceCanAccessPeer(&can, m->dev[g], m->dev[0]) == hipSuccess && can) {
      (void)hipSetDevice(m->dev[g]);
      const hipError_t pe = hipDeviceEnablePeerAccess(m->dev[0], 0);  // "already enabled" is fine
      (void)hipGetLastError();
      if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) m->peer[g] = 1;
    }
    if (log && log[0] && log[0] != '0')
      fprintf(stderr, "isscabac_multi: device %d -> root device %d: hipDeviceCanAccessPeer=%d, gather %s\n", m->dev[g], m->dev[0], can,
              m->peer[g] == 1 ? "peer to peer" : "staged through the host");
  }
  *out = m;
  return ISSCABAC_OK;
}

void isscabac_destroy_multi(isscabac_multi *m) {
  if (!m) return;
  DeviceGuard restore;
  for (size_t g = 0; g < m->eng.size(); g++) {
    (void)hipSetDevice(m->dev[g]);
    if (g < m->d_pay.size() && m->d_pay[g]) (void)hipFree(m->d_pay[g]);
    if (g < m->d_off.size() && m->d_off[g]) (void)hipFree(m->d_off[g]);
    isscabac_destroy(m->eng[g]);
  }
  delete m;
}

int isscabac_multi_peer_access(const isscabac_multi *m, int g) { return m && g >= 0 && g < (int)m->peer.size() ? m->peer[g] : -1; }
int isscabac_multi_device_count(const isscabac_multi *m) { return m ? (int)m->eng.size() : 0; }
isscabac_engine *isscabac_multi_engine(isscabac_multi *m, int g) { return m && g >= 0 && g < (int)m->eng.size() ? m->eng[g] : nullptr; }
const char *isscabac_multi_last_error(const isscabac_multi *m) { return m ? m->err.c_str() : isscabac_last_error(nullptr); }
void isscabac_multi_shard(const isscabac_multi *m, uint64_t S, int g, uint64_t *first, uint64_t *end) {
  uint64_t lo = 0, hi = 0;
  if (m && g >= 0 && g < (int)m->eng.size()) shard_of(S, g, (int)m->eng.size(), &lo, &hi);
  if (first) *first = lo;
  if (end) *end = hi;
}

// ---- host buffers ---------------------------------------------------------------------------------------------------
}  // extern "C"

// Shard-wise host encode: code(g, lo, hi, part, cap, poff, &tot) codes streams [lo, hi) on engine g into a scratch of its own
// (sizes are not known before); then the pieces are placed and the offsets rebased.  bound = worst-case bytes per stream.
template <class F>
static int multi_encode_host(isscabac_multi *m, const char *who, uint64_t S, uint64_t bound, uint8_t *payload, uint64_t payload_cap,
                             uint64_t *offsets, uint64_t *total_bytes, F code) {
  const int n = (int)m->eng.size();
  std::vector<std::unique_ptr<uint8_t[]>> part(n);  // (new uint8_t[k]: not zeroed -- a value-initialised vector would touch
  std::vector<uint64_t> part_cap(n, 0);             //  every page of ~3 GiB per Mi streams before the engine writes them)
  std::vector<std::vector<uint64_t>> poff(n);
  DeviceGuard restore;
  std::vector<uint64_t> tot(n, 0);
  int rc = on_all(m, [&](int g) -> int {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (hi == lo) return ISSCABAC_OK;
    part_cap[g] = (hi - lo) * bound;
    part[g].reset(new (std::nothrow) uint8_t[part_cap[g]]);
    if (!part[g]) return engine_fail(m->eng[g], ISSCABAC_ERR_NOMEM, "%s: no host memory for %llu scratch bytes", who, (unsigned long long)part_cap[g]);
    poff[g].resize(hi - lo + 1);
    return code(g, lo, hi, part[g].get(), part_cap[g], poff[g].data(), &tot[g]);
  });
  if (rc) return rc;
  uint64_t total = 0;
  for (int g = 0; g < n; g++) total += tot[g];
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap) return multi_fail(m, ISSCABAC_ERR_CAPACITY, std::string(who) + ": payload needs " + std::to_string(total) + " bytes");
  uint64_t base = 0;
  for (int g = 0; g < n; g++) {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (tot[g]) memcpy(payload + base, part[g].get(), tot[g]);
    for (uint64_t s = lo; s < hi; s++) offsets[s] = poff[g][s - lo] + base;
    base += tot[g];
  }
  offsets[S] = total;
  return ISSCABAC_OK;
}

// Shard-wise host decode: decode(g, lo, hi, payload slice, offsets from 0) -- after the WHOLE offsets array has been checked,
// before any shard touches the payload: the device threads run at the same time, and a shard whose own slice looks fine
// would otherwise upload payload[offsets[lo], offsets[hi]) from the caller's buffer although a later entry shows the array
// to be malformed (e.g. [0, 1000000, 5]) -- a host read out of bounds.  After this check every shard's slice lies inside
// [0, offsets[S]), which is what the caller vouches for.
template <class F>
static int multi_decode_host(isscabac_multi *m, const char *who, const uint8_t *payload, const uint64_t *offsets, uint64_t S, F decode) {
  const int n = (int)m->eng.size();
  for (uint64_t s = 0; s < S; s++) {
    if (offsets[s + 1] < offsets[s]) return multi_fail(m, ISSCABAC_ERR_ARG, std::string(who) + ": offsets not monotone at stream " + std::to_string(s));
    if (offsets[s + 1] - offsets[s] > 0xFFFFFFFFull) return multi_fail(m, ISSCABAC_ERR_ARG, std::string(who) + ": stream " + std::to_string(s) + " longer than 4 GiB");
  }
  DeviceGuard restore;
  return on_all(m, [&](int g) -> int {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (hi == lo) return ISSCABAC_OK;
    std::vector<uint64_t> off(hi - lo + 1);
    for (uint64_t s = lo; s <= hi; s++) off[s - lo] = offsets[s] - offsets[lo];
    return decode(g, lo, hi, payload + offsets[lo], off.data());
  });
}

extern "C" {

int isscabac_multi_encode_batch(isscabac_multi *m, const uint8_t *bins, const uint8_t *ctx, const uint32_t *nbins_per_stream,
                                uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                                uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!bins || !ctx || !payload || !offsets || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_encode_batch: null buffer");
  return multi_encode_host(m, "isscabac_multi_encode_batch", S, isscabac_stream_bound(nbins), payload, payload_cap, offsets, total_bytes,
                           [&](int g, uint64_t lo, uint64_t hi, uint8_t *part, uint64_t cap, uint64_t *poff, uint64_t *tot) {
                             return isscabac_encode_batch(m->eng[g], bins + lo * pitch, ctx + lo * pitch,
                                                          nbins_per_stream ? nbins_per_stream + lo : nullptr, hi - lo, nbins, pitch,
                                                          ctx_init, nctx, part, cap, poff, tot);
                           });
}

int isscabac_multi_decode_batch(isscabac_multi *m, const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                                const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                                const uint8_t *ctx_init, int nctx, uint8_t *bins_out, uint32_t *status) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!payload || !offsets || !ctx || !bins_out || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_decode_batch: null buffer");
  return multi_decode_host(m, "isscabac_multi_decode_batch", payload, offsets, S,
                           [&](int g, uint64_t lo, uint64_t hi, const uint8_t *pay, const uint64_t *off) {
                             return isscabac_decode_batch(m->eng[g], pay, off, ctx + lo * pitch,
                                                          nbins_per_stream ? nbins_per_stream + lo : nullptr, hi - lo, nbins, pitch,
                                                          ctx_init, nctx, bins_out + lo * pitch, status ? status + lo : nullptr);
                           });
}

// the same with the bins as bits on the caller's side (row s of bins_bits: pitch / 8 bytes)
int isscabac_multi_encode_batch_packed(isscabac_multi *m, const uint8_t *bins_bits, const uint8_t *ctx, const uint32_t *nbins_per_stream,
                                       uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                                       uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!bins_bits || !ctx || !payload || !offsets || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_encode_batch_packed: null buffer");
  return multi_encode_host(m, "isscabac_multi_encode_batch_packed", S, isscabac_stream_bound(nbins), payload, payload_cap, offsets, total_bytes,
                           [&](int g, uint64_t lo, uint64_t hi, uint8_t *part, uint64_t cap, uint64_t *poff, uint64_t *tot) {
                             return isscabac_encode_batch_packed(m->eng[g], bins_bits + lo * (pitch / 8), ctx + lo * pitch,
                                                                 nbins_per_stream ? nbins_per_stream + lo : nullptr, hi - lo, nbins,
                                                                 pitch, ctx_init, nctx, part, cap, poff, tot);
                           });
}

int isscabac_multi_decode_batch_packed(isscabac_multi *m, const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                                       const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                                       const uint8_t *ctx_init, int nctx, uint8_t *bins_bits_out, uint32_t *status) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!payload || !offsets || !ctx || !bins_bits_out || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_decode_batch_packed: null buffer");
  return multi_decode_host(m, "isscabac_multi_decode_batch_packed", payload, offsets, S,
                           [&](int g, uint64_t lo, uint64_t hi, const uint8_t *pay, const uint64_t *off) {
                             return isscabac_decode_batch_packed(m->eng[g], pay, off, ctx + lo * pitch,
                                                                 nbins_per_stream ? nbins_per_stream + lo : nullptr, hi - lo, nbins,
                                                                 pitch, ctx_init, nctx, bins_bits_out + lo * (pitch / 8),
                                                                 status ? status + lo : nullptr);
                           });
}

// symbol-level calls over the shards (config 4's path: every column a stream; a whole-matrix stream is one stream = one device)
int isscabac_multi_encode_symbols_stats(isscabac_multi *m, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S,
                                        uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init,
                                        int nctx, uint8_t *payload, uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes,
                                        uint32_t *nbins_out, const isscabac_symbol_stats *stats) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!sym || !params || !payload || !offsets || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_encode_symbols: null buffer");
  const uint32_t mb = isscabac_symbol_max_bins(params);
  if (mb == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_encode_symbols: unsupported symbol parameters");
  return multi_encode_host(m, "isscabac_multi_encode_symbols", S, isscabac_stream_bound(nsym * mb), payload, payload_cap, offsets, total_bytes,
                           [&](int g, uint64_t lo, uint64_t hi, uint8_t *part, uint64_t cap, uint64_t *poff, uint64_t *tot) {
                             isscabac_symbol_stats st = {nullptr, nullptr, nullptr, nullptr, 0};
                             if (stats) {  // per-stream arrays: this shard's rows
                               st.ctx_hist = stats->ctx_hist ? stats->ctx_hist + lo * ISSCABAC_STAT_CTX : nullptr;
                               st.ctx_cost = stats->ctx_cost ? stats->ctx_cost + lo * ISSCABAC_STAT_CTX : nullptr;
                               st.sym_bits = stats->sym_bits ? stats->sym_bits + lo * pitch : nullptr;
                               st.bin_trace = stats->bin_trace ? stats->bin_trace + lo * stats->bin_trace_pitch : nullptr;
                               st.bin_trace_pitch = stats->bin_trace_pitch;
                             }
                             return isscabac_encode_symbols_stats(m->eng[g], sym + lo * pitch, nsym_per_stream ? nsym_per_stream + lo : nullptr,
                                                                  hi - lo, nsym, pitch, params, ctx_init, nctx, part, cap, poff, tot,
                                                                  nbins_out ? nbins_out + lo : nullptr, stats ? &st : nullptr);
                           });
}

int isscabac_multi_encode_symbols(isscabac_multi *m, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym,
                                  uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                                  uint8_t *payload, uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out) {
  return isscabac_multi_encode_symbols_stats(m, sym, nsym_per_stream, S, nsym, pitch, params, ctx_init, nctx, payload, payload_cap, offsets,
                                             total_bytes, nbins_out, nullptr);
}

int isscabac_multi_decode_symbols(isscabac_multi *m, const uint8_t *payload, const uint64_t *offsets, const uint32_t *nsym_per_stream,
                                  uint64_t S, uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                  const uint8_t *ctx_init, int nctx, uint8_t *sym_out, uint32_t *status) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!payload || !offsets || !params || !sym_out || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_decode_symbols: null buffer");
  return multi_decode_host(m, "isscabac_multi_decode_symbols", payload, offsets, S,
                           [&](int g, uint64_t lo, uint64_t hi, const uint8_t *pay, const uint64_t *off) {
                             return isscabac_decode_symbols(m->eng[g], pay, off, nsym_per_stream ? nsym_per_stream + lo : nullptr,
                                                            hi - lo, nsym, pitch, params, ctx_init, nctx, sym_out + lo * pitch,
                                                            status ? status + lo : nullptr);
                           });
}

// ---- device-resident shards, bitstream gathered on the root device ----------------------------------------------------------
int isscabac_multi_encode_batch_dev(isscabac_multi *m, const uint8_t *const *d_bins, const uint8_t *const *d_ctx,
                                    const uint32_t *const *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                                    const uint8_t *ctx_init, int nctx, uint8_t *d_payload_root, uint64_t payload_cap,
                                    uint64_t *d_offsets_root, uint64_t *total_bytes) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!d_bins || !d_ctx || !d_payload_root || !d_offsets_root || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_encode_batch_dev: null buffer");
  const int n = (int)m->eng.size();
  std::vector<uint64_t> tot(n, 0);
  DeviceGuard restore;
  // 1. every device codes its shard into its own buffers (capacity = the provable bound, so that this cannot fail late)
  int rc = on_all(m, [&](int g) -> int {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (hi == lo) return ISSCABAC_OK;
    isscabac_engine *e = m->eng[g];
    HIP_TRY(e, hipSetDevice(m->dev[g]));
    const uint64_t need = (hi - lo) * isscabac_stream_bound(nbins), oneed = (hi - lo + 1) * sizeof(uint64_t);
    if (m->pay_cap[g] < need) {
      if (m->d_pay[g]) HIP_TRY(e, hipFree(m->d_pay[g]));
      m->d_pay[g] = nullptr;
      m->pay_cap[g] = 0;
      HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&m->d_pay[g]), need));
      m->pay_cap[g] = need;
    }
    if (m->off_cap[g] < oneed) {
      if (m->d_off[g]) HIP_TRY(e, hipFree(m->d_off[g]));
      m->d_off[g] = nullptr;
      m->off_cap[g] = 0;
      HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&m->d_off[g]), oneed));
      m->off_cap[g] = oneed;
    }
    return isscabac_encode_batch_dev(e, d_bins[g], d_ctx[g], d_nbins ? d_nbins[g] : nullptr, hi - lo, nbins, pitch, ctx_init, nctx,
                                     m->d_pay[g], need, m->d_off[g], &tot[g]);
  });
  if (rc) return rc;
  uint64_t total = 0;
  std::vector<uint64_t> base(n, 0);
  for (int g = 0; g < n; g++) {
    base[g] = total;
    total += tot[g];
  }
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap) return multi_fail(m, ISSCABAC_ERR_CAPACITY, "isscabac_multi_encode_batch_dev: payload needs " + std::to_string(total) + " bytes");
  // 2. the gather: every device pushes its piece to the root concurrently, then the root shifts the offsets by the piece's base
  rc = on_all(m, [&](int g) -> int {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (hi == lo) return ISSCABAC_OK;
    isscabac_engine *e = m->eng[g];
    HIP_TRY(e, hipSetDevice(m->dev[g]));
    if (tot[g]) HIP_TRY(e, hipMemcpyPeerAsync(d_payload_root + base[g], m->dev[0], m->d_pay[g], m->dev[g], tot[g], e->stream));
    // offsets: the piece's first hi - lo entries go to the root as they are (the last one, the piece's size, is the next
    // piece's first entry once rebased: not sent, so that no two devices write the same word) and are rebased there
    HIP_TRY(e, hipMemcpyPeerAsync(d_offsets_root + lo, m->dev[0], m->d_off[g], m->dev[g], (hi - lo) * sizeof(uint64_t), e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return ISSCABAC_OK;
  });
  if (rc) return rc;
  isscabac_engine *r = m->eng[0];
  HIP_TRY(r, hipSetDevice(m->dev[0]));
  for (int g = 0; g < n; g++) {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (hi == lo || base[g] == 0) continue;
    const uint64_t cnt = hi - lo;
    hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, r->stream, d_offsets_root + lo,
                       d_offsets_root + lo, cnt, base[g]);
  }
  HIP_TRY(r, hipGetLastError());
  HIP_TRY(r, hipMemcpyAsync(d_offsets_root + S, &total, sizeof total, hipMemcpyHostToDevice, r->stream));
  HIP_TRY(r, hipStreamSynchronize(r->stream));
  return ISSCABAC_OK;
}

// The decode side of the device-resident form: payload and offsets live on the ROOT device (where encode_batch_dev left them);
// every device pulls its slice [offsets[lo], offsets[hi]) and its offsets straight from the root (hipMemcpyPeerAsync, all
// links of the root busy at once), rebases the offsets to its slice on the device and decodes its shard into its own
// d_bins_out[g] / d_status[g].  The only host visit: the n + 1 shard boundaries out of the offsets array.
int isscabac_multi_decode_batch_dev(isscabac_multi *m, const uint8_t *d_payload_root, uint64_t payload_bytes,
                                    const uint64_t *d_offsets_root, const uint8_t *const *d_ctx, const uint32_t *const *d_nbins,
                                    uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                                    uint8_t *const *d_bins_out, uint32_t *const *d_status) {
  if (!m) return ISSCABAC_ERR_ARG;
  if (!d_payload_root || !d_offsets_root || !d_ctx || !d_bins_out || S == 0) return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_decode_batch_dev: null buffer");
  const int n = (int)m->eng.size();
  DeviceGuard restore;
  isscabac_engine *r = m->eng[0];
  HIP_TRY(r, hipSetDevice(m->dev[0]));
  std::vector<uint64_t> edge(n + 1, 0);  // offsets[first stream of shard g], and offsets[S]
  for (int g = 0; g <= n; g++) {
    uint64_t lo = S, hi = S;
    if (g < n) shard_of(S, g, n, &lo, &hi);
    HIP_TRY(r, hipMemcpyAsync(&edge[g], d_offsets_root + lo, sizeof(uint64_t), hipMemcpyDeviceToHost, r->stream));
  }
  HIP_TRY(r, hipStreamSynchronize(r->stream));
  for (int g = 0; g < n; g++)
    if (edge[g + 1] < edge[g] || edge[g + 1] > payload_bytes)
      return multi_fail(m, ISSCABAC_ERR_ARG, "isscabac_multi_decode_batch_dev: offsets not monotone or beyond the payload at shard " + std::to_string(g));
  return on_all(m, [&](int g) -> int {
    uint64_t lo, hi;
    shard_of(S, g, n, &lo, &hi);
    if (hi == lo) return ISSCABAC_OK;
    isscabac_engine *e = m->eng[g];
    HIP_TRY(e, hipSetDevice(m->dev[g]));
    const uint64_t bytes = edge[g + 1] - edge[g], need = bytes + 16, oneed = (hi - lo + 1) * sizeof(uint64_t);
    if (m->pay_cap[g] < need) {
      if (m->d_pay[g]) HIP_TRY(e, hipFree(m->d_pay[g]));
      m->d_pay[g] = nullptr;
      m->pay_cap[g] = 0;
      HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&m->d_pay[g]), need));
      m->pay_cap[g] = need;
    }
    if (m->off_cap[g] < oneed) {
      if (m->d_off[g]) HIP_TRY(e, hipFree(m->d_off[g]));
      m->d_off[g] = nullptr;
      m->off_cap[g] = 0;
      HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&m->d_off[g]), oneed));
      m->off_cap[g] = oneed;
    }
    if (bytes) HIP_TRY(e, hipMemcpyPeerAsync(m->d_pay[g], m->dev[g], d_payload_root + edge[g], m->dev[0], bytes, e->stream));
    HIP_TRY(e, hipMemcpyPeerAsync(m->d_off[g], m->dev[g], d_offsets_root + lo, m->dev[0], oneed, e->stream));
    const uint64_t cnt = hi - lo + 1;
    hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream, m->d_off[g], m->d_off[g], cnt,
                       (uint64_t)0 - edge[g]);  // (unsigned wrap-around: subtracts the slice's first byte)
    HIP_TRY(e, hipGetLastError());
    return isscabac_decode_batch_dev(e, m->d_pay[g], bytes, m->d_off[g], d_ctx[g], d_nbins ? d_nbins[g] : nullptr, hi - lo, nbins, pitch,
                                     ctx_init, nctx, d_bins_out[g], d_status ? d_status[g] : nullptr);
  });
}

}  // extern "C"
