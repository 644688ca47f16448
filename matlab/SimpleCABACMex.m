% SimpleCABACMex  MEX gateway of the MI355X batched CABAC engine (help text; the code is SimpleCABACMex.cpp).
%
% The per-bin commands of christianrohlfing/ISScabac's MEX file, with the same strings, argument counts, result shapes
% and error texts -- every numeric argument and result is a double, the handle h is a double too:
%
%   h    = SimpleCABACMex('initByProb',  fileName, p0)        p0(c) = probability of a 0 in context c (one value per context)
%   h    = SimpleCABACMex('initByState', fileName, triples)   [ctxID mps state  ctxID mps state ...], ctxID is ignored
%          SimpleCABACMex('encodeStart', h)
%          SimpleCABACMex('encodeBin', h, bin, ctx)            bin 0 / 1, ctx = 0-based context index
%   bits = SimpleCABACMex('getNumBits', h)                     bits written to the stream so far
%          SimpleCABACMex('encodeFinish', h)                   writes fileName
%          SimpleCABACMex('decodeStart', h)                    reads fileName
%   bin  = SimpleCABACMex('decodeBin', h, ctx)
%   st   = SimpleCABACMex('decodeFinish', h)                   0 = clean; a broken stream is a warning and a non-zero st, not an abort
%   [trace, transitions] = SimpleCABACMex('getEncoderStats', h, ctx)     5 x N uint8, 128 x 128 uint32
%   [trace, transitions] = SimpleCABACMex('getDecoderStats', h, ctx)     (these two work on every platform here)
%
% Commands this engine adds (one independent bitstream per matrix COLUMN; the GPU codes all columns of a call at once):
%
%          SimpleCABACMex('encodeBinEP', h, bin)                bin = SimpleCABACMex('decodeBinEP', h)
%          SimpleCABACMex('encodeBinProb', h, bin, p)           bin = SimpleCABACMex('decodeBinProb', h, p)    p in percent
%   [payload, offsets] = SimpleCABACMex('encodeBatch', h, bins, ctx [, nbins])            bins, ctx: B x S uint8, ctx 255 = bypass
%   [bins, status]     = SimpleCABACMex('decodeBatch', h, payload, offsets, ctx [, nbins])  (ctx as uint16: up to 999 contexts,
%                                                                                            65535 = bypass)
%   [payload, offsets] = SimpleCABACMex('encodeBatchPacked', h, bits, ctx [, nbins])      bins as bits, 8 to a byte
%   [bits, status]     = SimpleCABACMex('decodeBatchPacked', h, payload, offsets, ctx [, nbins])
%   [payload, offsets, nbins] = SimpleCABACMex('encodeSymbols', h, sym, params [, nsym])  sym: R x S uint8 symbols; binarisation and
%   [payload, offsets, nbins, ctxHist, ctxCost, symBits] = SimpleCABACMex('encodeSymbolsStats', h, sym, params [, nsym])   context
%   [sym, status]      = SimpleCABACMex('decodeSymbols', h, payload, offsets, R, params [, nsym])       selection on the GPU
%   [p0, p0bytes]      = SimpleCABACMex('issInitContextModel', sym, params)               empirical p(0) of the ISS contexts
%          SimpleCABACMex('writeStreamFiles', payload, offsets, prefix)                   one file per stream, <prefix>%06d.bin
%   [payload, offsets] = SimpleCABACMex('readStreamFiles', prefix, S)
%          SimpleCABACMex('setDevices', [d0 d1 ...])            batch commands over several GPUs; [] = back to one
%          SimpleCABACMex('delete', h)
%
% payload is a uint8 column, offsets a double column of S + 1 byte positions: stream s is payload(offsets(s)+1 : offsets(s+1)),
% byte for byte what the original writes into the file of stream s.  params: see coder.cabacSymbolParams.
%
% See also cabacWrapper, coder.cabacEncode, coder.cabacDecode.
