function [nbits, ctxInit0, ctxHist, ctxCost, H] = cabacEncode(G, Nq, param)
% cabacEncode  Encode the integer matrix G (values 0 .. Nq-1) with CABAC on the MI355X engine.
%
%   [nbits, ctxInit0] = coder.cabacEncode(G, Nq, param)
%   [nbits, ctxInit0, ctxHist, ctxCost, H] = coder.cabacEncode(G, Nq, param)
%
% Same call, same results and the same file as coder.cabacEncode of christianrohlfing/ISScabac: param.fn receives the
% bitstream of the whole matrix (one stream, column after column), nbits is its size in bits, ctxInit0 the uint8 context
% initialisation the decoder needs.  The optional outputs are the statistics the original gathers for its plots: how often
% every context was used, how many bits were written while it was, and the bits written per matrix element.
%
% What differs is how it gets there: binarisation, the empirical context initialisation, context selection and the
% arithmetic coder all run on the GPU, in TWO gateway calls per matrix (initialisation, coding) instead of three calls
% per bin.  param: binMethod, cmTypes, Nlbp, equalProb, fn (ISS.m fills them in).

  if nargin < 2, Nq = 2; end
  G = uint8(G);
  [rows, cols] = size(G); %#ok<ASGLU>

  % initial probability of a zero for every context, as one byte each -- what is transmitted as side information
  nctx = 7 * param.Nlbp + 2;
  if param.equalProb
    ctxInit0 = uint8(0.5 * ones(1, nctx) * 255);
  else
    [~, ctxInit0] = SimpleCABACMex('issInitContextModel', G, coder.cabacSymbolParams(Nq, param, 0));
  end
  ctxInit = double(ctxInit0) / 255;   % the decoder dequantises the same way

  c = cabacWrapper(ctxInit, param.fn);
  params = coder.cabacSymbolParams(Nq, param, rows);
  if nargout > 2
    [payload, ~, ~, hist, cost, bits] = c.encodeSymbolsStats(G(:), params);
    ctxHist = hist(1:nctx)';
    ctxCost = cost(1:nctx)';
    H = reshape(bits, size(G));
  else
    payload = c.encodeSymbols(G(:), params);
  end
  delete(c);

  fid = fopen(param.fn, 'w');
  if fid < 0, error('coder:file', 'cannot open %s for writing', param.fn); end
  fwrite(fid, payload, 'uint8');
  fclose(fid);
  nbits = numel(payload) * 8;
end
