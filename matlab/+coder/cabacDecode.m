function G = cabacDecode(Nq, param, ctxInit, siz)
% cabacDecode  Decode a matrix of integers (0 .. Nq-1) that coder.cabacEncode wrote to param.fn.
%
%   G = coder.cabacDecode(Nq, param, ctxInit, siz)
%
% Same call and result as coder.cabacDecode of christianrohlfing/ISScabac: ctxInit is the uint8 context initialisation
% the encoder returned, siz the size of the matrix.  Context selection, the end-of-symbol test and de-binarisation
% depend on bins that were just decoded, so the whole loop runs inside one GPU kernel: ONE gateway call per matrix.

  if nargin < 2, Nq = 2; end
  ctxInit = double(ctxInit) / 255;

  fid = fopen(param.fn, 'r');
  if fid < 0, error('coder:file', 'cannot open %s for reading', param.fn); end
  payload = uint8(fread(fid, inf, 'uint8'));
  fclose(fid);

  c = cabacWrapper(ctxInit, param.fn);
  params = coder.cabacSymbolParams(Nq, param, siz(1));
  [sym, status] = c.decodeSymbols(payload, [0 numel(payload)], prod(siz), params);
  delete(c);
  if status ~= 0
    error('coder:stream', 'the CABAC stream in %s is damaged (status %d)', param.fn, status);
  end
  G = double(reshape(sym, siz));
end
