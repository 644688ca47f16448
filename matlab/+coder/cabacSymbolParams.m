function params = cabacSymbolParams(Nq, param, colRows)
% cabacSymbolParams  The engine's symbol-level parameter vector for an ISS coder run.
%
%   params = coder.cabacSymbolParams(Nq, param, colRows)
%
% param is the struct ISS.m hands to coder.cabacEncode / coder.cabacDecode (fields binMethod, cmTypes, Nlbp); colRows is
% the number of rows of the matrix when the whole matrix is coded as ONE stream, column after column, or 0 when every
% column is a stream of its own.  The result is [binarization k nq selector issTypes nlbp colRows], the fields of
% isscabac_symbol_params in include/isscabac.h, as the gateway commands encodeSymbols / decodeSymbols /
% issInitContextModel take them.

  if nargin < 3, colRows = 0; end
  if Nq <= 2
    % no binarisation: the symbols are the bins -- truncated unary with a single bin is exactly that
    binarization = 0; k = 0; nq = 2;
  else
    method = param.binMethod;
    nq = Nq;
    switch method
      case 'DEC2TU'
        binarization = 0; k = 0;
      case {'DEC2EG0', 'DEC2EG1', 'DEC2EG2'}
        binarization = 1; k = str2double(method(end));
      case {'DEC2TR0', 'DEC2TR1', 'DEC2TR2'}   % encode side only, as in the original: its decoder has no end-of-symbol rule
        binarization = 2; k = str2double(method(end));
      case 'DEC2FL32'                          % encode side only
        binarization = 3; k = 0;
      otherwise
        error('coder:binMethod', 'unknown binarisation method %s', method);
    end
  end
  names = {'cond0', 'cond1', 'conds0', 'conds1', 'condbinlft'};
  issTypes = 0;
  for it = 1:numel(names)
    if any(ismember(param.cmTypes, names{it}))
      issTypes = issTypes + 2^(it - 1);
    end
  end
  selectorISS = 1;
  params = double([binarization, k, nq, selectorISS, issTypes, param.Nlbp, colRows]);
end
