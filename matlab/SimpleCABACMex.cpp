// SimpleCABACMex.cpp -- MEX gateway of the MI355X CABAC engine: a THIN translator from the reference's MEX command
// strings (ref: CABAC/SimpleCABACMex.cpp:100-472) onto the C ABI of include/isscabac.h.  No CABAC logic lives here.
//
//   mex CXXFLAGS="\$CXXFLAGS -std=c++11" -I<repo>/include SimpleCABACMex.cpp -L<repo>/isscabac_amd/lib -lisscabac_hip
//
// Same command strings, arity, return shapes and error texts as the reference for
//   initByProb / initByState / encodeStart / encodeBin / getNumBits / encodeFinish / decodeStart / decodeBin /
//   decodeFinish        (getEncoderStats / getDecoderStats exist only in the reference's Windows build; like its Linux
//                        build this gateway answers "Error: Invalid Command")
// plus the additive batch commands
//   [payload, offsets] = SimpleCABACMex('encodeBatch', h, bins, ctx [, nbins])   bins/ctx: B x S uint8 (one stream
//                                                                               per column), ctx 255 = bypass
//   [bins, status]     = SimpleCABACMex('decodeBatch', h, payload, offsets, ctx [, nbins])
//   SimpleCABACMex('delete', h)
// The bitstream FILE named at init stays the encoder -> decoder hand-off of the per-bin commands (ref: :195,288):
// encodeFinish writes it, decodeStart reads it.
#include "mex.h"
#include "matrix.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "isscabac.h"

namespace {

isscabac_engine *g_engine = NULL;

struct Handle {
  isscabac_stream *stream;
  std::string fn;
  std::vector<uint8_t> ctx_init;
};

void at_exit() {
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

// ref: getPointer, SimpleCABACMex.cpp:84-97
Handle *get_handle(int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("Error: You need to provide the pointer to the initialized CABAC engine \n");
  uintptr_t v = (uintptr_t)*mxGetPr(prhs[1]);
  Handle *h = (Handle *)v;
  if (h == NULL) mexErrMsgTxt("Error: No initialized CABAC instance provided \n");
  return h;
}

const uint8_t *u8_arg(const mxArray *a, const char *msg) {
  if (!mxIsClass(a, "uint8")) mexErrMsgTxt(msg);
  return (const uint8_t *)mxGetData(a);
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
    Handle *h = new Handle;
    char *fn = mxArrayToString(prhs[1]);
    h->fn = fn ? fn : "";
    const int nel = (int)mxGetNumberOfElements(prhs[2]);
    const int nctx = c == "initByState" ? nel / 3 : nel;  // ref: :165,172
    if (nctx < 1 || nctx > ISSCABAC_MAX_CTX) {
      delete h;
      mexErrMsgTxt("Error: invalid context initialization\n");
    }
    h->ctx_init.resize(nctx);
    const int rc = c == "initByState" ? isscabac_init_by_state(mxGetPr(prhs[2]), nctx, h->ctx_init.data())
                                      : isscabac_init_by_prob(mxGetPr(prhs[2]), nctx, h->ctx_init.data());
    if (rc != ISSCABAC_OK) {
      delete h;
      mexErrMsgTxt("Error: invalid context initialization\n");
    }
    h->stream = NULL;
    check(isscabac_stream_open(engine(), h->ctx_init.data(), nctx, &h->stream), "init");
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
    check(isscabac_stream_encode_start(h->stream), "encodeStart");
  } else if (c == "encodeBin") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs != 4) mexErrMsgTxt("Error: invalid input, provide the bin and context index to be encoded with\n");
    const unsigned bin = (unsigned)*mxGetPr(prhs[2]);
    const int ctx = (int)*mxGetPr(prhs[3]);
    if (bin != 0 && bin != 1) mexErrMsgTxt("Error: invalid input 3, bin to be encoded should either be 1 or 0\n");
    const uint8_t b = (uint8_t)bin, cx = (uint8_t)ctx;
    check(isscabac_stream_encode_bins(h->stream, &b, &cx, 1), "encodeBin");
  } else if (c == "getNumBits") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs != 2) mexErrMsgTxt("Error: invalid input\n");
    uint64_t bits = 0;
    check(isscabac_stream_num_bits(h->stream, &bits), "getNumBits");
    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    *mxGetPr(plhs[0]) = (double)bits;
  } else if (c == "encodeFinish") {
    Handle *h = get_handle(nrhs, prhs);
    uint64_t n = 0;
    std::vector<uint8_t> buf(1 << 16);
    int rc = isscabac_stream_encode_finish(h->stream, buf.data(), buf.size(), &n);
    if (rc == ISSCABAC_ERR_CAPACITY) {
      buf.resize(n);
      rc = isscabac_stream_encode_finish(h->stream, buf.data(), buf.size(), &n);
    }
    check(rc, "encodeFinish");
    FILE *f = fopen(h->fn.c_str(), "wb");
    if (!f || fwrite(buf.data(), 1, n, f) != n) mexErrMsgTxt("Error: bitstreamfile access error\n");
    fclose(f);
  } else if (c == "decodeStart") {
    Handle *h = get_handle(nrhs, prhs);
    FILE *f = fopen(h->fn.c_str(), "rb");  // ref: :288-292
    if (!f) {
      mexPrintf("Error: filename %s cannot be opened for reading\n", h->fn.c_str());
      mexErrMsgTxt("Error: bitstreamfile access error\n");
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[4096];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    check(isscabac_stream_decode_start(h->stream, buf.data(), buf.size()), "decodeStart");
  } else if (c == "decodeBin") {
    if (nrhs < 2) mexErrMsgTxt("Error: You need to provide the pointer to the initialized CABAC engine \n");
    if (nlhs != 1) mexErrMsgTxt("Error: invalid command, provide a variable to store the decoded bin \n");
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 3) mexErrMsgTxt("Error: invalid input\n");
    const uint8_t cx = (uint8_t)(int)*mxGetPr(prhs[2]);
    uint8_t bin = 0;
    check(isscabac_stream_decode_bins(h->stream, &cx, 1, &bin), "decodeBin");
    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    *mxGetPr(plhs[0]) = bin;
  } else if (c == "decodeFinish") {
    Handle *h = get_handle(nrhs, prhs);
    uint32_t status = 0;
    check(isscabac_stream_decode_finish(h->stream, &status), "decodeFinish");
    // the reference asserts here (CABAC_ArithmeticDecoder.cpp:78-81); a MATLAB error is the closest safe thing
    if (status != 0) mexErrMsgTxt("Error: bitstream not terminated properly\n");
  } else if (c == "encodeBatch") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 4 || nrhs > 5) mexErrMsgTxt("Error: encodeBatch needs the handle, bins, ctx and optionally nbins\n");
    const uint8_t *bins = u8_arg(prhs[2], "Error: bins must be a uint8 matrix (one stream per column)\n");
    const uint8_t *ctx = u8_arg(prhs[3], "Error: ctx must be a uint8 matrix (one stream per column)\n");
    const size_t B = mxGetM(prhs[2]), S = mxGetN(prhs[2]);
    if (mxGetM(prhs[3]) != B || mxGetN(prhs[3]) != S || B == 0 || S == 0) mexErrMsgTxt("Error: bins and ctx must have the same size\n");
    const uint32_t *nb = NULL;
    if (nrhs == 5) {
      if (!mxIsClass(prhs[4], "uint32") || mxGetNumberOfElements(prhs[4]) != S) mexErrMsgTxt("Error: nbins must be uint32 with one entry per stream\n");
      nb = (const uint32_t *)mxGetData(prhs[4]);
    }
    // MATLAB's B x S column-major matrix is the ABI's stream-major layout; only the row pitch has to be a multiple of 16
    const uint32_t pitch = (uint32_t)((B + 15) / 16 * 16);
    std::vector<uint8_t> pb, pc;
    if (pitch != B) {
      pb.assign((size_t)pitch * S, 0);
      pc.assign((size_t)pitch * S, 0);
      for (size_t s = 0; s < S; s++) {
        memcpy(&pb[s * pitch], bins + s * B, B);
        memcpy(&pc[s * pitch], ctx + s * B, B);
      }
      bins = pb.data();
      ctx = pc.data();
    }
    const uint64_t cap = S * isscabac_stream_bound((uint32_t)B);
    std::vector<uint8_t> payload(cap);
    std::vector<uint64_t> offsets(S + 1);
    uint64_t total = 0;
    check(isscabac_encode_batch(engine(), bins, ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(), (int)h->ctx_init.size(),
                                payload.data(), cap, offsets.data(), &total), "encodeBatch");
    plhs[0] = mxCreateNumericMatrix(total, 1, mxUINT8_CLASS, mxREAL);
    memcpy(mxGetData(plhs[0]), payload.data(), total);
    if (nlhs > 1) {
      plhs[1] = mxCreateDoubleMatrix(S + 1, 1, mxREAL);
      for (size_t s = 0; s <= S; s++) mxGetPr(plhs[1])[s] = (double)offsets[s];
    }
  } else if (c == "decodeBatch") {
    Handle *h = get_handle(nrhs, prhs);
    if (nrhs < 5 || nrhs > 6) mexErrMsgTxt("Error: decodeBatch needs the handle, payload, offsets, ctx and optionally nbins\n");
    const uint8_t *payload = u8_arg(prhs[2], "Error: payload must be uint8\n");
    if (!mxIsClass(prhs[3], "double")) mexErrMsgTxt("Error: offsets must be double\n");
    const uint8_t *ctx = u8_arg(prhs[4], "Error: ctx must be a uint8 matrix (one stream per column)\n");
    const size_t B = mxGetM(prhs[4]), S = mxGetN(prhs[4]);
    if (mxGetNumberOfElements(prhs[3]) != S + 1 || B == 0 || S == 0) mexErrMsgTxt("Error: offsets must have one entry per stream plus one\n");
    std::vector<uint64_t> offsets(S + 1);
    for (size_t s = 0; s <= S; s++) offsets[s] = (uint64_t)mxGetPr(prhs[3])[s];
    if (offsets[S] > mxGetNumberOfElements(prhs[2])) mexErrMsgTxt("Error: offsets exceed the payload\n");
    const uint32_t *nb = NULL;
    if (nrhs == 6) {
      if (!mxIsClass(prhs[5], "uint32") || mxGetNumberOfElements(prhs[5]) != S) mexErrMsgTxt("Error: nbins must be uint32 with one entry per stream\n");
      nb = (const uint32_t *)mxGetData(prhs[5]);
    }
    const uint32_t pitch = (uint32_t)((B + 15) / 16 * 16);
    std::vector<uint8_t> pc;
    if (pitch != B) {
      pc.assign((size_t)pitch * S, 0);
      for (size_t s = 0; s < S; s++) memcpy(&pc[s * pitch], ctx + s * B, B);
      ctx = pc.data();
    }
    std::vector<uint8_t> bins((size_t)pitch * S);
    std::vector<uint32_t> status(S);
    check(isscabac_decode_batch(engine(), payload, offsets.data(), ctx, nb, S, (uint32_t)B, pitch, h->ctx_init.data(),
                                (int)h->ctx_init.size(), bins.data(), status.data()), "decodeBatch");
    plhs[0] = mxCreateNumericMatrix(B, S, mxUINT8_CLASS, mxREAL);
    uint8_t *out = (uint8_t *)mxGetData(plhs[0]);
    for (size_t s = 0; s < S; s++) memcpy(out + s * B, &bins[s * pitch], B);
    if (nlhs > 1) {
      plhs[1] = mxCreateDoubleMatrix(S, 1, mxREAL);
      for (size_t s = 0; s < S; s++) mxGetPr(plhs[1])[s] = status[s];
    }
  } else if (c == "delete") {
    Handle *h = get_handle(nrhs, prhs);
    isscabac_stream_close(h->stream);
    delete h;
  } else {
    mexErrMsgTxt("Error: Invalid Command\n");
  }
}
