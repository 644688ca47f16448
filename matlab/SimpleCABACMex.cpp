// SimpleCABACMex.cpp -- MEX gateway of the MI355X CABAC engine: a THIN translator from the reference's MEX command
// strings (ref: CABAC/SimpleCABACMex.cpp:100-472) onto the C ABI of include/isscabac.h.  No CABAC logic lives here.
//
//   mex CXXFLAGS="\$CXXFLAGS -std=c++11" -I<repo>/include SimpleCABACMex.cpp -L<repo>/isscabac_amd/lib -lisscabac_hip
//
// Same command strings, arity, return shapes and error texts as the reference for
//   initByProb / initByState / encodeStart / encodeBin / getNumBits / encodeFinish / decodeStart / decodeBin /
//   decodeFinish / getEncoderStats / getDecoderStats     (the last two exist only in the reference's Windows build,
//                                                          ref: CommonDef.h:39-43; here they always work)
// plus the additive commands
//   SimpleCABACMex('encodeBinEP', h, bin) / bin = ('decodeBinEP', h)                 equiprobable bins (ref: encodeBinEP)
//   SimpleCABACMex('encodeBinProb', h, bin, p) / bin = ('decodeBinProb', h, p)       fixed probability p % of a 1 (ref: encodeBinProb)
//   [payload, offsets] = ('encodeBatch', h, bins, ctx [, nbins])      bins/ctx: B x S uint8, one stream per column,
//   [bins, status]     = ('decodeBatch', h, payload, offsets, ctx [, nbins])                        ctx 255 = bypass
//       ctx as a uint16 matrix (65535 = bypass) takes handles of up to 999 contexts -- the reference's limit, CommonDef.h:56 --
//       through isscabac_encode_batch16 / isscabac_decode_batch16
//   [payload, offsets] = ('encodeBatchPacked', h, bits, ctx [, nbins])   bits: (ceil(B/16)*2) x S uint8, bin i of a column at
//   [bits, status]     = ('decodeBatchPacked', h, payload, offsets, ctx [, nbins])      bit i % 8 of byte i / 8
//   [payload, offsets, nbins]                         = ('encodeSymbols', h, sym, params [, nsym])   sym: R x S uint8 symbols,
//   [payload, offsets, nbins, ctxHist, ctxCost, symBits] = ('encodeSymbolsStats', h, sym, params [, nsym])   one stream per column;
//   [sym, status]                                     = ('decodeSymbols', h, payload, offsets, R, params [, nsym])
//        params = [binarization k nq selector issTypes nlbp colRows] (isscabac_symbol_params, include/isscabac.h)
//   [p0, p0bytes] = ('issInitContextModel', sym, params)              empirical context initialisation (cabacInitContextModel.m)
//   ('writeStreamFiles', payload, offsets, prefix) / [payload, offsets] = ('readStreamFiles', prefix, S)
//        one file per stream, "<prefix>%06d.bin": what the reference's per-stream files hold
//   ('setDevices', [d0 d1 ...])     code the batch commands on several GPUs (isscabac_multi); ('setDevices', []) = back to one
//   SimpleCABACMex('delete', h)
// The bitstream FILE named at init stays the encoder -> decoder hand-off of the per-bin commands (ref: :195,288):
// encodeFinish writes it, decodeStart reads it.
//
// Deliberate differences from the reference (INTEGRATION.md has the table):
//   * a context index outside the initialised contexts is an error ("Error: invalid context index") -- the reference does
//     not check (ref: :223-224) and indexes its 1000-entry array with whatever it is given;
//   * decodeFinish reports a broken stream (terminate bin / stop bit) as a warning and, if asked for, as its result; the
//     reference asserts there, which a release MEX build compiles out (ref: CABAC_ArithmeticDecoder.cpp:78-81).
#include "mex.h"
#include "matrix.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <set>
#include <string>
#include <vector>

#include "isscabac.h"

namespace {

isscabac_engine *g_engine = NULL;
isscabac_multi *g_multi = NULL;  // set by 'setDevices': the batch commands then run on several GPUs

// The per-bin commands are served by the host twin of the handle (isscabac_cpu_stream_*: the engine's own per-lane code compiled
// for the host -- one bin per call has no parallelism to give a GPU: ~20 ns against 2.3 us per decodeBin), or, with
// ISSCABAC_MEX_PERBIN=gpu in the environment at init time, by the device-resident handle (isscabac_stream_*).  Same results
// either way; the batch / symbol commands always run on the GPU.
struct Handle {
  isscabac_stream *stream;   // device handle, or NULL
  isscabac_cpu_stream *cpu;  // host twin, or NULL
  std::string fn;
  std::vector<uint8_t> ctx_init;
};
// A few whole-matrix streams (col_rows > 0: the reference's one-file-per-matrix format) are serial work as well: up to this many
// of them per call are coded by the host twin (a 400 x 20 matrix: 0.2 ms on one core against 1.3 / 2.5 ms on one GPU lane);
// more streams, or columns as streams, go to the GPU
const size_t kSerialMatrixStreams = 4;
bool perbin_on_gpu();
bool serial_on_host(const isscabac_symbol_params &sp, size_t S) { return sp.col_rows > 0 && S <= kSerialMatrixStreams && !perbin_on_gpu() && !g_multi; }
bool perbin_on_gpu() {
  const char *v = getenv("ISSCABAC_MEX_PERBIN");
  return v && (v[0] == 'g' || v[0] == 'G');
}
std::set<Handle *> g_handles;  // every live handle: 'delete' and at_exit() only touch what is really there

void close_handle(Handle *h) {
  if (h->stream) isscabac_stream_close(h->stream);
  if (h->cpu) isscabac_cpu_stream_close(h->cpu);
  g_handles.erase(h);
  delete h;
  if (g_handles.empty() && mexIsLocked()) mexUnlock();
}

void at_exit() {
  // streams first (they point into the engine), then the engines
  while (!g_handles.empty()) close_handle(*g_handles.begin());
  if (g_multi) isscabac_destroy_multi(g_multi);
  g_multi = NULL;
  if (g_engine) isscabac_destroy(g_engine);
  g_engine = NULL;
}

isscabac_engine *engine() {
  if (!g_engine) {
    const char *dev = getenv("ISSCABAC_DEVICE");
    if (isscabac_create(dev ? atoi(dev) : 0, &g_engine) != ISSCABAC_OK) {
      mexPrintf("Error: %s\n", isscabac_last_error(NULL));
      mexErrMsgTxt("Error: no MI355X engine available\n");
    }
    mexAtExit(at_exit);
  }
  return g_engine;
}

void check(int rc, const char *what) {
  if (rc == ISSCABAC_OK) return;
  mexPrintf("Error: %s: %s\n", what, isscabac_last_error(g_engine));
  mexErrMsgTxt("Error: CABAC engine call failed\n");
}
void check_h(const Handle *h, int rc, const char *what) {  // a per-bin command's status: the text comes from whoever served it
  if (rc == ISSCABAC_OK) return;
  mexPrintf("Error: %s: %s\n", what, h->cpu ? isscabac_cpu_stream_last_error(h->cpu) : isscabac_last_error(g_engine));
  mexErrMsgTxt("Error: CABAC engine call failed\n");
}
// the per-bin calls, on whichever form of the handle exists
int h_encode_start(Handle *h) { return h->cpu ? isscabac_cpu_stream_encode_start(h->cpu) : isscabac_stream_encode_start(h->stream); }
int h_encode_ops(Handle *h, const uint8_t *b, const uint16_t *op, uint32_t n) {
  return h->cpu ? isscabac_cpu_stream_encode_ops(h->cpu, b, op, n) : isscabac_stream_encode_ops(h->stream, b, op, n);
}
int h_num_bits(Handle *h, uint64_t *bits) { return h->cpu ? isscabac_cpu_stream_num_bits(h->cpu, bits) : isscabac_stream_num_bits(h->stream, bits); }
int h_encode_finish(Handle *h, uint8_t *out, uint64_t cap, uint64_t *n) {
  return h->cpu ? isscabac_cpu_stream_encode_finish(h->cpu, out, cap, n) : isscabac_stream_encode_finish(h->stream, out, cap, n);
}
int h_decode_start(Handle *h, const uint8_t *p, uint64_t n) {
  return h->cpu ? isscabac_cpu_stream_decode_start(h->cpu, p, n) : isscabac_stream_decode_start(h->stream, p, n);
}
int h_decode_ops(Handle *h, const uint16_t *op, uint32_t n, uint8_t *out) {
  return h->cpu ? isscabac_cpu_stream_decode_ops(h->cpu, op, n, out) : isscabac_stream_decode_ops(h->stream, op, n, out);
}
int h_decode_finish(Handle *h, uint32_t *st) { return h->cpu ? isscabac_cpu_stream_decode_finish(h->cpu, st) : isscabac_stream_decode_finish(h->stream, st); }
int h_get_stats(Handle *h, int decoder, int ctx, uint8_t *t5, uint64_t cap, uint64_t *n, uint32_t *tr) {
  return h->cpu ? isscabac_cpu_stream_get_stats(h->cpu, decoder, ctx, t5, cap, n, tr) : isscabac_stream_get_stats(h->stream, decoder, ctx, t5, cap, n, tr);
}
void check_multi(int rc, const char *what) {
  if (rc == ISSCABAC_OK) return;
  mexPrintf("Error: %s: %s\n", what, isscabac_multi_last_error(g_multi));
  mexErrMsgTxt("Error: CABAC engine call failed\n");
}
void check_host(int rc, const char *what) {  // container calls: no engine involved
  if (rc == ISSCABAC_OK) return;
  mexPrintf("Error: %s: %s\n", what, isscabac_last_error(NULL));
  mexErrMsgTxt("Error: bitstreamfile access error\n");
}

// ref: getPointer, SimpleCABACMex.cpp:84-97
Handle *get_handle(int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("Error: You need to provide the pointer to the initialized CABAC engine \n");
  if (!mxIsClass(prhs[1], "double") || mxGetNumberOfElements(prhs[1]) < 1) mexErrMsgTxt("Error: No initialized CABAC instance provided \n");
  uintptr_t v = (uintptr_t)*mxGetPr(prhs[1]);
  Handle *h = (Handle *)v;
  if (h == NULL || !g_handles.count(h)) mexErrMsgTxt("Error: No initialized CABAC instance provided \n");
  return h;
}

const uint8_t *u8_arg(const mxArray *a, const char *msg) {
  if (!mxIsClass(a, "uint8")) mexErrMsgTxt(msg);
  return (const uint8_t *)mxGetData(a);
}

double scalar_arg(const mxArray *a, const char *msg) {
  if (!mxIsClass(a, "double") || mxGetNumberOfElements(a) < 1) mexErrMsgTxt(msg);
  return *mxGetPr(a);
}

// the context index arrives as a double: range-check it BEFORE narrowing, so that 256 never becomes 0 and -1 or 255 never
// become a bypass bin
uint16_t ctx_op(const Handle *h, const mxArray *a) {
  const double c = scalar_arg(a, "Error: invalid context index\n");
  if (!(c >= 0 && c < (double)h->ctx_init.size()) || c != (double)(long)c) mexErrMsgTxt("Error: invalid context index\n");
  return ISSCABAC_OP_CTX((unsigned)c);
}
uint16_t prob_op(const mxArray *a) {
  const double p = scalar_arg(a, "Error: invalid probability\n");
  if (!(p >= 1 && p <= 99) || p != (double)(long)p) mexErrMsgTxt("Error: invalid probability, provide the probability of a 1 in percent (1..99)\n");
  return ISSCABAC_OP_PROB((unsigned)p);
}
uint8_t bin_arg(const mxArray *a) {
  const double b = scalar_arg(a, "Error: invalid input 3, bin to be encoded should either be 1 or 0\n");
  if (b != 0 && b != 1) mexErrMsgTxt("Error: invalid input 3, bin to be encoded should either be 1 or 0\n");
  return (uint8_t)b;
}

// scratch memory MATLAB frees by itself when the command returns or raises
uint8_t *scratch(size_t n) { return (uint8_t *)mxCalloc(n ? n : 1, 1); }

// a B x S uint8 matrix (one stream per column) in the ABI's layout: rows padded to a multiple of 16 bytes
const uint8_t *padded(const uint8_t *a, size_t B, size_t S, uint32_t pitch) {
  if (pitch == B) return a;
  uint8_t *p = scratch((size_t)pitch * S);
  for (size_t s = 0; s < S; s++) memcpy(p + s * pitch, a + s * B, B);
  return p;
}

// ... and a B x S uint16 matrix of context indices (the batch commands' form for more than 255 contexts)
const uint16_t *padded16(const uint16_t *a, size_t B, size_t S, uint32_t pitch) {
  if (pitch == B) return a;
  uint16_t *p = (uint16_t *)scratch((size_t)pitch * S * sizeof(uint16_t));
  for (size_t s = 0; s < S; s++) memcpy(p + s * pitch, a + s * B, B * sizeof(uint16_t));
  return p;
}

void symbol_params(const mxArray *a, isscabac_symbol_params *sp) {
  if (!mxIsClass(a, "double") || mxGetNumberOfElements(a) != 7)
    mexErrMsgTxt("Error: params must be [binarization k nq selector issTypes nlbp colRows]\n");
  const double *p = mxGetPr(a);
  for (int i = 0; i < 7; i++)
    if (!(p[i] >= 0 && p[i] <= 4294967295.0)) mexErrMsgTxt("Error: params must be [binarization k nq selector issTypes nlbp colRows]\n");
  sp->binarization = (uint32_t)p[0];
  sp->k = (uint32_t)p[1];
  sp->nq = (uint32_t)p[2];
  sp->selector = (uint32_t)p[3];
  sp->iss_types = (uint32_t)p[4];
  sp->nlbp = (uint32_t)p[5];
  sp->col_rows = (uint32_t)p[6];
}

const uint32_t *u32_per_stream(int present, const mxArray *a, size_t S, const char *msg) {
  if (!present) return NULL;
  if (!mxIsClass(a, "uint32") || mxGetNumberOfElements(a) != S) mexErrMsgTxt(msg);
  return (const uint32_t *)mxGetData(a);
}

mxArray *doubles_of_u64(const uint64_t *v, size_t n) {
  mxArray *a = mxCreateDoubleMatrix(n, 1, mxREAL);
  for (size_t i = 0; i < n; i++) mxGetPr(a)[i] = (double)v[i];
  return a;
}
mxArray *doubles_of_u32(const uint32_t *v, size_t m, size_t n) {
  mxArray *a = mxCreateDoubleMatrix(m, n, mxREAL);
  for (size_t i = 0; i < m * n; i++) mxGetPr(a)[i] = (double)v[i];
  return a;
}

void stats_command(int decoder, int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  // ref: SimpleCABACMex.cpp:356-411 (encoder), :413-466 (decoder)
  if (nrhs < 2) mexErrMsgTxt("Error: You need to provide the pointer to the initialized CABAC engine \n");
  if (nlhs != 2) mexErrMsgTxt("Error: invalid command, provide two variables to store the trace \n");
  Handle *h = get_handle(nrhs, prhs);
  if (nrhs < 3) mexErrMsgTxt("Error: invalid context index\n");
  const int ctx = (int)ctx_op(h, prhs[2]);
  uint64_t n = 0;
  check_h(h, h_get_stats(h, decoder, ctx, NULL, 0, &n, NULL), decoder ? "getDecoderStats" : "getEncoderStats");
  plhs[0] = mxCreateNumericMatrix(5, (size_t)n, mxUINT8_CLASS, mxREAL);
  plhs[1] = mxCreateNumericMatrix(128, 128, mxUINT32_CLASS, mxREAL);
  check_h(h, h_get_stats(h, decoder, ctx, (uint8_t *)mxGetData(plhs[0]), n, &n, (uint32_t *)mxGetData(plhs[1])),
          decoder ? "getDecoderStats" : "getEncoderStats");
}

}  // namespace

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  char cmd[64];
  if (nrhs < 1 || !mxIsClass(prhs[0], "char") || mxGetString(prhs[0], cmd, sizeof(cmd))) {
    mexErrMsgTxt("Error: input 0 must be a valid keyword - init, encodeStart, encodeBin, encodeFinish, decodeStart, decodeBin, decodeFinish\n");
  }
  const std::string c(cmd);

  if (c == "initByState" || c == "initByProb") {
    if (nrhs < 2) mexErrMsgTxt("Error: please provide the filename string and the context initializations\n");
    if (nrhs > 3) mexErrMsgTxt("Error: too many parameters, provide the filename and the context initializations");
    if (!mxIsClass(prhs[1], "char")) mexErrMsgTxt("Error: invalid filename \n");
    if (nrhs < 3 || !mxIsClass(prhs[2], "double")) mexErrMsgTxt("Error: invalid context initialization\n");
    const int nel = (int)mxGetNumberOfElements(prhs[2]);
    const int nctx = c == "initByState" ? nel / 3 : nel;  // ref: :165,172
    if (nctx < 1 || nctx > ISSCABAC_STREAM_MAX_CTX) mexErrMsgTxt("Error: invalid context initialization\n");  // ref asserts n < 1000, CABAC_ContextModelsInit.cpp:53,84
    uint8_t *init = scratch(nctx);
    const int rc = c == "initByState" ? isscabac_init_by_state(mxGetPr(prhs[2]), nctx, init) : isscabac_init_by_prob(mxGetPr(prhs[2]), nctx, init);
    if (rc != ISSCABAC_OK) mexErrMsgTxt("Error: invalid context initialization\n");
    isscabac_stream *stream = NULL;
    isscabac_cpu_stream *cpu = NULL;
    if (perbin_on_gpu()) {
      check(isscabac_stream_open(engine(), init, nctx, &stream), "init");  // (nothing of ours is allocated yet if this raises)
      check(isscabac_stream_set_trace(stream, 1), "init");  // what the reference's Windows build always records
    } else {  // the host twin: no engine, no GPU touched until a batch command asks for one
      if (isscabac_cpu_stream_open(init, nctx, &cpu) != ISSCABAC_OK || isscabac_cpu_stream_set_trace(cpu, 1) != ISSCABAC_OK)
        mexErrMsgTxt("Error: invalid context initialization\n");
      mexAtExit(at_exit);
    }
    Handle *h = new Handle;
    h->stream = stream;
    h->cpu = cpu;
    char *fn = mxArrayToString(prhs[1]);
    h->fn = fn ? fn : "";
    if (fn) mxFree(fn);
    h->ctx_init.assign(init, init + nctx);
    g_handles.insert(h);
    if (!mexIsLocked()) mexLock();  // `clear mex` must not unload the engine under live handles
    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    *mxGetPr(plhs[0]) = (double)(uintptr_t)h;  // handle = pointer in a double, ref: :154-156
  } else if (c == "encodeStart") {
    Handle *h = get_handle(nrhs, prhs);
    FILE *f = fopen(h->fn.c_str(), "wb");  // ref: :195-199
    if (!f) {
      mexPrintf("Error: filename %s cannot be opened for writing\n", h->fn.c_str());
      mexErrMsgTxt("Error: bitstreamfile access error\n");
    }
    fclose(f);
    check_h(h, h_encode_start(h), "encodeStart");
  } else if (c == "encodeBin") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs != 4) mexErrMsgTxt("Error: invalid input, provide the bin and context index to be encoded with\n");
    const uint8_t b = bin_arg(prhs[2]);
    const uint16_t op = ctx_op(h, prhs[3]);
    check_h(h, h_encode_ops(h, &b, &op, 1), "encodeBin");
  } else if (c == "encodeBinEP" || c == "encodeBinProb") {
    Handle *h = get_handle(nrhs, prhs);
    const bool ep = c == "encodeBinEP";
    if (nrhs != (ep ? 3 : 4)) mexErrMsgTxt(ep ? "Error: invalid input, provide the bin to be encoded\n" : "Error: invalid input, provide the bin and its probability of being 1 in percent\n");
    const uint8_t b = bin_arg(prhs[2]);
    const uint16_t op = ep ? ISSCABAC_OP_BYPASS : prob_op(prhs[3]);
    check_h(h, h_encode_ops(h, &b, &op, 1), cmd);
  } else if (c == "getNumBits") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs != 2) mexErrMsgTxt("Error: invalid input\n");
    uint64_t bits = 0;
    check_h(h, h_num_bits(h, &bits), "getNumBits");
    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    *mxGetPr(plhs[0]) = (double)bits;
  } else if (c == "encodeFinish") {
    Handle *h = get_handle(nrhs, prhs);
    uint64_t n = 0;
    uint8_t *buf = scratch(1 << 16);
    int rc = h_encode_finish(h, buf, 1 << 16, &n);
    if (rc == ISSCABAC_ERR_CAPACITY) {
      buf = scratch((size_t)n);
      rc = h_encode_finish(h, buf, n, &n);
    }
    check_h(h, rc, "encodeFinish");
    FILE *f = fopen(h->fn.c_str(), "wb");
    const bool ok = f && fwrite(buf, 1, (size_t)n, f) == n;
    if (f) fclose(f);
    if (!ok) mexErrMsgTxt("Error: bitstreamfile access error\n");
  } else if (c == "decodeStart") {
    Handle *h = get_handle(nrhs, prhs);
    FILE *f = fopen(h->fn.c_str(), "rb");  // ref: :288-292
    if (!f) {
      mexPrintf("Error: filename %s cannot be opened for reading\n", h->fn.c_str());
      mexErrMsgTxt("Error: bitstreamfile access error\n");
    }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = scratch(len > 0 ? (size_t)len : 1);
    const bool ok = len >= 0 && fread(buf, 1, (size_t)len, f) == (size_t)len;
    fclose(f);
    if (!ok) mexErrMsgTxt("Error: bitstreamfile access error\n");
    check_h(h, h_decode_start(h, buf, (uint64_t)len), "decodeStart");
  } else if (c == "decodeBin" || c == "decodeBinEP" || c == "decodeBinProb") {
    if (nrhs < 2) mexErrMsgTxt("Error: You need to provide the pointer to the initialized CABAC engine \n");
    if (nlhs != 1) mexErrMsgTxt("Error: invalid command, provide a variable to store the decoded bin \n");
    Handle *h = get_handle(nrhs, prhs);
    uint16_t op = ISSCABAC_OP_BYPASS;
    if (c != "decodeBinEP") {
      if (nrhs < 3) mexErrMsgTxt("Error: invalid input\n");
      op = c == "decodeBin" ? ctx_op(h, prhs[2]) : prob_op(prhs[2]);
    }
    uint8_t bin = 0;
    check_h(h, h_decode_ops(h, &op, 1, &bin), cmd);
    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    *mxGetPr(plhs[0]) = bin;
  } else if (c == "decodeFinish") {
    Handle *h = get_handle(nrhs, prhs);
    uint32_t status = 0;
    check_h(h, h_decode_finish(h, &status), "decodeFinish");
    // the reference asserts here (CABAC_ArithmeticDecoder.cpp:78-81) and a release MEX build compiles the asserts out:
    // report, do not raise
    if (status != 0) mexWarnMsgTxt("bitstream not terminated properly");
    if (nlhs > 0) {
      plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
      *mxGetPr(plhs[0]) = status;
    }
  } else if (c == "getEncoderStats") {
    stats_command(0, nlhs, plhs, nrhs, prhs);
  } else if (c == "getDecoderStats") {
    stats_command(1, nlhs, plhs, nrhs, prhs);
  } else if (c == "encodeBatch") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 4 || nrhs > 5) mexErrMsgTxt("Error: encodeBatch needs the handle, bins, ctx and optionally nbins\n");
    const uint8_t *bins = u8_arg(prhs[2], "Error: bins must be a uint8 matrix (one stream per column)\n");
    const bool wide = mxIsClass(prhs[3], "uint16");  // 16-bit context indices: up to 999 contexts, 65535 = bypass
    const uint8_t *ctx = wide ? NULL : u8_arg(prhs[3], "Error: ctx must be a uint8 (or, for more than 255 contexts, uint16) matrix (one stream per column)\n");
    const size_t B = mxGetM(prhs[2]), S = mxGetN(prhs[2]);
    if (mxGetM(prhs[3]) != B || mxGetN(prhs[3]) != S || B == 0 || S == 0) mexErrMsgTxt("Error: bins and ctx must have the same size\n");
    if (!wide && h->ctx_init.size() > ISSCABAC_MAX_CTX) mexErrMsgTxt("Error: a uint8 ctx matrix names a context in one byte: at most 255 contexts (pass ctx as uint16)\n");
    const uint32_t *nb = u32_per_stream(nrhs == 5, nrhs == 5 ? prhs[4] : NULL, S, "Error: nbins must be uint32 with one entry per stream\n");
    // MATLAB's B x S column-major matrix is the ABI's stream-major layout; only the row pitch has to be a multiple of 16
    const uint32_t pitch = (uint32_t)((B + 15) / 16 * 16);
    bins = padded(bins, B, S, pitch);
    if (!wide) ctx = padded(ctx, B, S, pitch);
    const uint64_t cap = S * isscabac_stream_bound((uint32_t)B);
    uint8_t *payload = scratch(cap);
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    uint64_t total = 0;
    engine();
    if (wide)
      check(isscabac_encode_batch16(g_engine, bins, padded16((const uint16_t *)mxGetData(prhs[3]), B, S, pitch), nb, S, (uint32_t)B, pitch,
                                    h->ctx_init.data(), (int)h->ctx_init.size(), payload, cap, offsets, &total), "encodeBatch");
    else if (g_multi)
      check_multi(isscabac_multi_encode_batch(g_multi, bins, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(), (int)h->ctx_init.size(),
                                              payload, cap, offsets, &total), "encodeBatch");
    else
      check(isscabac_encode_batch(g_engine, bins, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(), (int)h->ctx_init.size(), payload,
                                  cap, offsets, &total), "encodeBatch");
    plhs[0] = mxCreateNumericMatrix(total, 1, mxUINT8_CLASS, mxREAL);
    memcpy(mxGetData(plhs[0]), payload, total);
    if (nlhs > 1) plhs[1] = doubles_of_u64(offsets, S + 1);
  } else if (c == "decodeBatch") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 5 || nrhs > 6) mexErrMsgTxt("Error: decodeBatch needs the handle, payload, offsets, ctx and optionally nbins\n");
    const uint8_t *payload = u8_arg(prhs[2], "Error: payload must be uint8\n");
    if (!mxIsClass(prhs[3], "double")) mexErrMsgTxt("Error: offsets must be double\n");
    const bool wide = mxIsClass(prhs[4], "uint16");
    const uint8_t *ctx = wide ? NULL : u8_arg(prhs[4], "Error: ctx must be a uint8 (or, for more than 255 contexts, uint16) matrix (one stream per column)\n");
    const size_t B = mxGetM(prhs[4]), S = mxGetN(prhs[4]);
    if (mxGetNumberOfElements(prhs[3]) != S + 1 || B == 0 || S == 0) mexErrMsgTxt("Error: offsets must have one entry per stream plus one\n");
    if (!wide && h->ctx_init.size() > ISSCABAC_MAX_CTX) mexErrMsgTxt("Error: a uint8 ctx matrix names a context in one byte: at most 255 contexts (pass ctx as uint16)\n");
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    for (size_t s = 0; s <= S; s++) offsets[s] = (uint64_t)mxGetPr(prhs[3])[s];
    if (offsets[S] > mxGetNumberOfElements(prhs[2])) mexErrMsgTxt("Error: offsets exceed the payload\n");
    const uint32_t *nb = u32_per_stream(nrhs == 6, nrhs == 6 ? prhs[5] : NULL, S, "Error: nbins must be uint32 with one entry per stream\n");
    const uint32_t pitch = (uint32_t)((B + 15) / 16 * 16);
    if (!wide) ctx = padded(ctx, B, S, pitch);
    uint8_t *bins = scratch((size_t)pitch * S);
    uint32_t *status = (uint32_t *)scratch(S * sizeof(uint32_t));
    engine();
    if (wide)
      check(isscabac_decode_batch16(g_engine, payload, offsets, padded16((const uint16_t *)mxGetData(prhs[4]), B, S, pitch), nb, S, (uint32_t)B,
                                    pitch, h->ctx_init.data(), (int)h->ctx_init.size(), bins, status), "decodeBatch");
    else if (g_multi)
      check_multi(isscabac_multi_decode_batch(g_multi, payload, offsets, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(),
                                              (int)h->ctx_init.size(), bins, status), "decodeBatch");
    else
      check(isscabac_decode_batch(g_engine, payload, offsets, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(), (int)h->ctx_init.size(),
                                  bins, status), "decodeBatch");
    plhs[0] = mxCreateNumericMatrix(B, S, mxUINT8_CLASS, mxREAL);
    uint8_t *out = (uint8_t *)mxGetData(plhs[0]);
    for (size_t s = 0; s < S; s++) memcpy(out + s * B, bins + s * pitch, B);
    if (nlhs > 1) plhs[1] = doubles_of_u32(status, S, 1);
  } else if (c == "encodeBatchPacked") {
    // bins as BITS: a (pitch / 8) x S uint8 matrix, bin i of a stream at bit i % 8 of byte i / 8 of its column
    // (pitch = B rounded up to 16; B = rows of ctx) -- an eighth of the bins' bytes crosses the PCIe link
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 4 || nrhs > 5) mexErrMsgTxt("Error: encodeBatchPacked needs the handle, the bins as bits, ctx and optionally nbins\n");
    const uint8_t *bits = u8_arg(prhs[2], "Error: the packed bins must be a uint8 matrix (one stream per column)\n");
    const uint8_t *ctx = u8_arg(prhs[3], "Error: ctx must be a uint8 matrix (one stream per column; no packed form for 16-bit context indices)\n");
    const size_t B = mxGetM(prhs[3]), S = mxGetN(prhs[3]);
    const uint32_t pitch = (uint32_t)((B + 15) / 16 * 16);
    if (B == 0 || S == 0 || mxGetM(prhs[2]) != pitch / 8 || mxGetN(prhs[2]) != S)
      mexErrMsgTxt("Error: the packed bins must have ceil(rows(ctx) / 16) * 2 rows and one column per stream\n");
    if (h->ctx_init.size() > ISSCABAC_MAX_CTX) mexErrMsgTxt("Error: the batch commands name a context in one byte: at most 255 contexts\n");
    const uint32_t *nb = u32_per_stream(nrhs == 5, nrhs == 5 ? prhs[4] : NULL, S, "Error: nbins must be uint32 with one entry per stream\n");
    ctx = padded(ctx, B, S, pitch);
    const uint64_t cap = S * isscabac_stream_bound((uint32_t)B);
    uint8_t *payload = scratch(cap);
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    uint64_t total = 0;
    engine();
    if (g_multi)
      check_multi(isscabac_multi_encode_batch_packed(g_multi, bits, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(), (int)h->ctx_init.size(),
                                                     payload, cap, offsets, &total), "encodeBatchPacked");
    else
      check(isscabac_encode_batch_packed(g_engine, bits, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(), (int)h->ctx_init.size(), payload,
                                         cap, offsets, &total), "encodeBatchPacked");
    plhs[0] = mxCreateNumericMatrix(total, 1, mxUINT8_CLASS, mxREAL);
    memcpy(mxGetData(plhs[0]), payload, total);
    if (nlhs > 1) plhs[1] = doubles_of_u64(offsets, S + 1);
  } else if (c == "decodeBatchPacked") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 5 || nrhs > 6) mexErrMsgTxt("Error: decodeBatchPacked needs the handle, payload, offsets, ctx and optionally nbins\n");
    const uint8_t *payload = u8_arg(prhs[2], "Error: payload must be uint8\n");
    if (!mxIsClass(prhs[3], "double")) mexErrMsgTxt("Error: offsets must be double\n");
    // (no packed form behind 16-bit context indices: a uint16 ctx is refused by u8_arg like any other class, as in encodeBatchPacked)
    const uint8_t *ctx = u8_arg(prhs[4], "Error: ctx must be a uint8 matrix (one stream per column; no packed form for 16-bit context indices)\n");
    const size_t B = mxGetM(prhs[4]), S = mxGetN(prhs[4]);
    if (mxGetNumberOfElements(prhs[3]) != S + 1 || B == 0 || S == 0) mexErrMsgTxt("Error: offsets must have one entry per stream plus one\n");
    if (h->ctx_init.size() > ISSCABAC_MAX_CTX) mexErrMsgTxt("Error: the batch commands name a context in one byte: at most 255 contexts\n");
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    for (size_t s = 0; s <= S; s++) offsets[s] = (uint64_t)mxGetPr(prhs[3])[s];
    if (offsets[S] > mxGetNumberOfElements(prhs[2])) mexErrMsgTxt("Error: offsets exceed the payload\n");
    const uint32_t *nb = u32_per_stream(nrhs == 6, nrhs == 6 ? prhs[5] : NULL, S, "Error: nbins must be uint32 with one entry per stream\n");
    const uint32_t pitch = (uint32_t)((B + 15) / 16 * 16);
    ctx = padded(ctx, B, S, pitch);
    uint32_t *status = (uint32_t *)scratch(S * sizeof(uint32_t));
    plhs[0] = mxCreateNumericMatrix(pitch / 8, S, mxUINT8_CLASS, mxREAL);  // the bits, straight into the result
    engine();
    if (g_multi)
      check_multi(isscabac_multi_decode_batch_packed(g_multi, payload, offsets, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(),
                                                     (int)h->ctx_init.size(), (uint8_t *)mxGetData(plhs[0]), status), "decodeBatchPacked");
    else
      check(isscabac_decode_batch_packed(g_engine, payload, offsets, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(),
                                         (int)h->ctx_init.size(), (uint8_t *)mxGetData(plhs[0]), status), "decodeBatchPacked");
    if (nlhs > 1) plhs[1] = doubles_of_u32(status, S, 1);
  } else if (c == "encodeSymbols" || c == "encodeSymbolsStats") {
    // one call per matrix instead of one per bin: binarisation, context selection and coding run on the GPU
    // (replaces the loops of ref: ISS/+coder/cabacEncode.m:45-70; the statistics are those of :39-42,60-67)
    Handle *h = get_handle(nrhs, prhs);
    const bool stats = c == "encodeSymbolsStats";
    if (nrhs < 4 || nrhs > 5) mexErrMsgTxt("Error: encodeSymbols needs the handle, the symbols, params and optionally nsym\n");
    const uint8_t *sym = u8_arg(prhs[2], "Error: symbols must be a uint8 matrix (one stream per column)\n");
    const size_t R = mxGetM(prhs[2]), S = mxGetN(prhs[2]);
    if (R == 0 || S == 0) mexErrMsgTxt("Error: symbols must be a uint8 matrix (one stream per column)\n");
    isscabac_symbol_params sp;
    symbol_params(prhs[3], &sp);
    const uint32_t *ns = u32_per_stream(nrhs == 5, nrhs == 5 ? prhs[4] : NULL, S, "Error: nsym must be uint32 with one entry per stream\n");
    const uint32_t mb = isscabac_symbol_max_bins(&sp);
    if (mb == 0) mexErrMsgTxt("Error: unsupported symbol parameters\n");
    const uint32_t pitch = (uint32_t)((R + 15) / 16 * 16);
    sym = padded(sym, R, S, pitch);
    const uint64_t cap = S * isscabac_stream_bound((uint32_t)(R * mb));
    uint8_t *payload = scratch(cap);
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    uint32_t *nbins = (uint32_t *)scratch(S * sizeof(uint32_t));
    uint64_t total = 0;
    isscabac_symbol_stats st = {NULL, NULL, NULL, NULL, 0};
    if (stats) {
      st.ctx_hist = (uint32_t *)scratch(S * ISSCABAC_STAT_CTX * sizeof(uint32_t));
      st.ctx_cost = (uint32_t *)scratch(S * ISSCABAC_STAT_CTX * sizeof(uint32_t));
      st.sym_bits = (uint16_t *)scratch(S * (size_t)pitch * sizeof(uint16_t));
    }
    if (serial_on_host(sp, S)) {
      if (isscabac_cpu_encode_symbols(sym, ns, S, (uint32_t)R, pitch, &sp, h->ctx_init.data(), (int)h->ctx_init.size(), payload, cap, offsets,
                                      &total, nbins, stats ? &st : NULL) != ISSCABAC_OK)
        mexErrMsgTxt("Error: unsupported symbol parameters\n");
    } else if (engine(), g_multi)  // the streams (columns, or whole matrices) split over the devices of 'setDevices'
      check_multi(isscabac_multi_encode_symbols_stats(g_multi, sym, ns, S, (uint32_t)R, pitch, &sp, h->ctx_init.data(), (int)h->ctx_init.size(),
                                                      payload, cap, offsets, &total, nbins, stats ? &st : NULL), cmd);
    else
      check(isscabac_encode_symbols_stats(g_engine, sym, ns, S, (uint32_t)R, pitch, &sp, h->ctx_init.data(), (int)h->ctx_init.size(), payload,
                                          cap, offsets, &total, nbins, stats ? &st : NULL), cmd);
    plhs[0] = mxCreateNumericMatrix(total, 1, mxUINT8_CLASS, mxREAL);
    memcpy(mxGetData(plhs[0]), payload, total);
    if (nlhs > 1) plhs[1] = doubles_of_u64(offsets, S + 1);
    if (nlhs > 2) plhs[2] = doubles_of_u32(nbins, S, 1);
    if (stats) {
      if (nlhs > 3) plhs[3] = doubles_of_u32(st.ctx_hist, ISSCABAC_STAT_CTX, S);  // column s = stream s
      if (nlhs > 4) plhs[4] = doubles_of_u32(st.ctx_cost, ISSCABAC_STAT_CTX, S);
      if (nlhs > 5) {
        plhs[5] = mxCreateDoubleMatrix(R, S, mxREAL);
        for (size_t s = 0; s < S; s++)
          for (size_t i = 0; i < R; i++) mxGetPr(plhs[5])[s * R + i] = st.sym_bits[s * pitch + i];
      }
    }
  } else if (c == "decodeSymbols") {
    // (replaces the loops of ref: ISS/+coder/cabacDecode.m:29-55)
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 6 || nrhs > 7) mexErrMsgTxt("Error: decodeSymbols needs the handle, payload, offsets, the symbols per stream, params and optionally nsym\n");
    const uint8_t *payload = u8_arg(prhs[2], "Error: payload must be uint8\n");
    if (!mxIsClass(prhs[3], "double") || mxGetNumberOfElements(prhs[3]) < 2) mexErrMsgTxt("Error: offsets must be double\n");
    const size_t S = mxGetNumberOfElements(prhs[3]) - 1;
    const double Rd = scalar_arg(prhs[4], "Error: invalid number of symbols per stream\n");
    if (!(Rd >= 1 && Rd <= 33554432.0)) mexErrMsgTxt("Error: invalid number of symbols per stream\n");
    const size_t R = (size_t)Rd;
    isscabac_symbol_params sp;
    symbol_params(prhs[5], &sp);
    const uint32_t *ns = u32_per_stream(nrhs == 7, nrhs == 7 ? prhs[6] : NULL, S, "Error: nsym must be uint32 with one entry per stream\n");
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    for (size_t s = 0; s <= S; s++) offsets[s] = (uint64_t)mxGetPr(prhs[3])[s];
    if (offsets[S] > mxGetNumberOfElements(prhs[2])) mexErrMsgTxt("Error: offsets exceed the payload\n");
    const uint32_t pitch = (uint32_t)((R + 15) / 16 * 16);
    uint8_t *sym = scratch((size_t)pitch * S);
    uint32_t *status = (uint32_t *)scratch(S * sizeof(uint32_t));
    if (serial_on_host(sp, S)) {
      if (isscabac_cpu_decode_symbols(payload, offsets, ns, S, (uint32_t)R, pitch, &sp, h->ctx_init.data(), (int)h->ctx_init.size(), sym, status) != ISSCABAC_OK)
        mexErrMsgTxt("Error: unsupported symbol parameters\n");
    } else if (engine(), g_multi)
      check_multi(isscabac_multi_decode_symbols(g_multi, payload, offsets, ns, S, (uint32_t)R, pitch, &sp, h->ctx_init.data(),
                                                (int)h->ctx_init.size(), sym, status), "decodeSymbols");
    else
      check(isscabac_decode_symbols(g_engine, payload, offsets, ns, S, (uint32_t)R, pitch, &sp, h->ctx_init.data(), (int)h->ctx_init.size(), sym,
                                    status), "decodeSymbols");
    plhs[0] = mxCreateNumericMatrix(R, S, mxUINT8_CLASS, mxREAL);
    uint8_t *out = (uint8_t *)mxGetData(plhs[0]);
    for (size_t s = 0; s < S; s++) memcpy(out + s * R, sym + s * pitch, R);
    if (nlhs > 1) plhs[1] = doubles_of_u32(status, S, 1);
  } else if (c == "issInitContextModel") {
    // (replaces ref: ISS/+coder/cabacInitContextModel.m and the uint8 quantisation of cabacEncode.m:30)
    if (nrhs != 3) mexErrMsgTxt("Error: issInitContextModel needs the symbols and params\n");
    const uint8_t *sym = u8_arg(prhs[1], "Error: symbols must be a uint8 matrix (one column per stream)\n");
    const size_t R = mxGetM(prhs[1]), S = mxGetN(prhs[1]);
    if (R == 0 || S == 0) mexErrMsgTxt("Error: symbols must be a uint8 matrix (one column per stream)\n");
    isscabac_symbol_params sp;
    symbol_params(prhs[2], &sp);
    if (sp.nlbp < 1 || sp.nlbp > 8) mexErrMsgTxt("Error: unsupported symbol parameters\n");
    const uint32_t pitch = (uint32_t)((R + 15) / 16 * 16);
    sym = padded(sym, R, S, pitch);
    const size_t nctx = 7 * sp.nlbp + 2;
    plhs[0] = mxCreateDoubleMatrix(1, nctx, mxREAL);
    mxArray *bytes = mxCreateNumericMatrix(1, nctx, mxUINT8_CLASS, mxREAL);
    check(isscabac_iss_init_context_model(engine(), sym, S, (uint32_t)R, pitch, &sp, mxGetPr(plhs[0]), (uint8_t *)mxGetData(bytes)),
          "issInitContextModel");
    if (nlhs > 1) plhs[1] = bytes;
  } else if (c == "writeStreamFiles") {
    if (nrhs != 4) mexErrMsgTxt("Error: writeStreamFiles needs payload, offsets and the file name prefix\n");
    const uint8_t *payload = u8_arg(prhs[1], "Error: payload must be uint8\n");
    if (!mxIsClass(prhs[2], "double") || mxGetNumberOfElements(prhs[2]) < 1 || !mxIsClass(prhs[3], "char")) mexErrMsgTxt("Error: offsets must be double, the prefix a char array\n");
    const size_t S = mxGetNumberOfElements(prhs[2]) - 1;
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    for (size_t s = 0; s <= S; s++) offsets[s] = (uint64_t)mxGetPr(prhs[2])[s];
    if (offsets[S] > mxGetNumberOfElements(prhs[1])) mexErrMsgTxt("Error: offsets exceed the payload\n");
    char *prefix = mxArrayToString(prhs[3]);
    char *names = (char *)scratch(S * 4096);
    const char **paths = (const char **)scratch(S * sizeof(char *));
    for (size_t s = 0; s < S; s++) {
      paths[s] = names + s * 4096;
      check_host(isscabac_stream_file_path(prefix, s, names + s * 4096, 4096), "writeStreamFiles");
    }
    mxFree(prefix);
    check_host(isscabac_write_stream_files(payload, offsets, S, paths), "writeStreamFiles");
  } else if (c == "readStreamFiles") {
    if (nrhs != 3 || !mxIsClass(prhs[1], "char")) mexErrMsgTxt("Error: readStreamFiles needs the file name prefix and the number of streams\n");
    const double Sd = scalar_arg(prhs[2], "Error: invalid number of streams\n");
    if (!(Sd >= 0 && Sd <= 1e9)) mexErrMsgTxt("Error: invalid number of streams\n");
    const size_t S = (size_t)Sd;
    char *prefix = mxArrayToString(prhs[1]);
    char *names = (char *)scratch(S * 4096);
    const char **paths = (const char **)scratch(S * sizeof(char *));
    for (size_t s = 0; s < S; s++) {
      paths[s] = names + s * 4096;
      check_host(isscabac_stream_file_path(prefix, s, names + s * 4096, 4096), "readStreamFiles");
    }
    mxFree(prefix);
    uint64_t *offsets = (uint64_t *)scratch((S + 1) * sizeof(uint64_t));
    uint64_t total = 0;
    const int rc = isscabac_read_stream_files(paths, S, NULL, 0, offsets, &total);
    if (rc != ISSCABAC_OK && rc != ISSCABAC_ERR_CAPACITY) check_host(rc, "readStreamFiles");
    plhs[0] = mxCreateNumericMatrix(total, 1, mxUINT8_CLASS, mxREAL);
    check_host(isscabac_read_stream_files(paths, S, (uint8_t *)mxGetData(plhs[0]), total ? total : 1, offsets, &total), "readStreamFiles");
    if (nlhs > 1) plhs[1] = doubles_of_u64(offsets, S + 1);
  } else if (c == "setDevices") {
    if (nrhs != 2 || !mxIsClass(prhs[1], "double")) mexErrMsgTxt("Error: setDevices needs a vector of device indices\n");
    const size_t n = mxGetNumberOfElements(prhs[1]);
    engine();
    if (g_multi) isscabac_destroy_multi(g_multi);
    g_multi = NULL;
    if (n > 0) {
      int ids[64];
      if (n > 64) mexErrMsgTxt("Error: at most 64 devices\n");
      for (size_t i = 0; i < n; i++) ids[i] = (int)mxGetPr(prhs[1])[i];
      if (isscabac_create_multi(ids, (int)n, &g_multi) != ISSCABAC_OK) {
        mexPrintf("Error: setDevices: %s\n", isscabac_last_error(NULL));
        mexErrMsgTxt("Error: CABAC engine call failed\n");
      }
    }
  } else if (c == "delete") {
    close_handle(get_handle(nrhs, prhs));
  } else {
    mexErrMsgTxt("Error: Invalid Command\n");
  }
}
