classdef cabacWrapper < handle
% cabacWrapper  MATLAB front end of the MI355X batched CABAC engine.
%
%   w = cabacWrapper(ctxInit, fileName)            contexts initialised by p(0), one value per context
%   w = cabacWrapper(ctxInit, fileName, false)     contexts initialised by [ctxID mps state ...] triples
%
% Per-bin calls (what CABAC/cabacDemo.m and ISS/+coder/cabacEncode.m / cabacDecode.m of christianrohlfing/ISScabac
% use; same names, arguments and results as that repository's class, so those scripts run unchanged):
%   encodeStart, encodeBin(bin, ctx), encodeFinish, getNumBits,
%   decodeStart, bin = decodeBin(ctx), decodeFinish,
%   [trace, stats] = getEncoderStats(ctx) / getDecoderStats(ctx)
% Two more kinds of bin, per call (new; the engine of the original has them, its MEX file does not reach them):
%   encodeBinEP(bin), bin = decodeBinEP()                  equiprobable bins
%   encodeBinProb(bin, p), bin = decodeBinProb(p)          p = probability of a 1 in percent, 1..99 (not 46..49, 51..54)
% Batch calls (new; one independent bitstream per matrix column, coded on the GPU in one go; ctx value 255 marks a
% bypass bin; nbins optionally gives the number of bins actually coded in every column):
%   [payload, offsets] = encodeBatch(bins, ctx [, nbins])
%   [bins, status]     = decodeBatch(payload, offsets, ctx [, nbins])
% Symbol calls (new; columns of integer symbols, binarisation and context selection done by the engine, see
% coder.cabacSymbolParams for params):
%   [payload, offsets, nbins] = encodeSymbols(sym, params [, nsym])
%   [payload, offsets, nbins, ctxHist, ctxCost, symBits] = encodeSymbolsStats(sym, params [, nsym])
%   [sym, status] = decodeSymbols(payload, offsets, symbolsPerStream, params [, nsym])
%
% Every method is a thin call into the MEX gateway SimpleCABACMex (matlab/SimpleCABACMex.cpp), which forwards to the C
% ABI of include/isscabac.h; no coding logic lives here.

  properties
    cabac_handle = [];          % engine handle returned by the gateway (a double)
    bitStreamName = '';         % file the per-bin calls write to / read from
    contextModelInitOptions;    % what the constructor was given (p(0) values or state triples)
  end

  methods
    % ---- life cycle ---------------------------------------------------------------------------------------------
    function obj = cabacWrapper(ctxInit, fileName, byProbability)
      if nargin < 3
        byProbability = true;
      end
      if ~ischar(fileName)
        error('cabacWrapper:fileName', 'the bitstream file name has to be a character array');
      end
      if ~ismatrix(ctxInit) || ~isnumeric(ctxInit)
        error('cabacWrapper:ctxInit', ...
              'context initialisation: p(0) per context, or [ctxID mps state, ctxID mps state, ...]');
      end
      obj.contextModelInitOptions = ctxInit;
      obj.bitStreamName = fileName;
      obj.init(byProbability);
    end

    function init(obj, byProbability)
      % (re)creates the engine handle; the gateway keeps the command names of the reference MEX file
      if nargin < 2 || byProbability
        command = 'initByProb';
      else
        command = 'initByState';
      end
      obj.cabac_handle = SimpleCABACMex(command, obj.bitStreamName, obj.contextModelInitOptions);
    end

    function delete(obj)
      if ~isempty(obj.cabac_handle)
        obj.gateway('delete');
        obj.cabac_handle = [];
      end
    end

    % ---- per-bin encoder ----------------------------------------------------------------------------------------
    function encodeStart(obj)
      obj.gateway('encodeStart');
    end

    function encodeBin(obj, binValue, ctxID)
      if ~(isequal(binValue, 0) || isequal(binValue, 1))
        error('cabacWrapper:bin', 'a bin is 0 or 1');
      end
      obj.gateway('encodeBin', binValue, ctxID);
    end

    function encodeFinish(obj)
      obj.gateway('encodeFinish');
    end

    function bits = getNumBits(obj)
      bits = SimpleCABACMex('getNumBits', obj.cabac_handle);
    end

    % ---- per-bin decoder ----------------------------------------------------------------------------------------
    function decodeStart(obj)
      obj.gateway('decodeStart');
    end

    function bin = decodeBin(obj, ctxID)
      bin = SimpleCABACMex('decodeBin', obj.cabac_handle, ctxID);
    end

    function decodeFinish(obj)
      obj.gateway('decodeFinish');
    end

    function status = decodeFinishStatus(obj)
      % like decodeFinish, but hands back the decoder's integrity flags (0 = the stream ended where it should)
      status = SimpleCABACMex('decodeFinish', obj.cabac_handle);
    end

    % ---- equiprobable and fixed-probability bins ----------------------------------------------------------------
    function encodeBinEP(obj, binValue)
      obj.gateway('encodeBinEP', binValue);
    end

    function bin = decodeBinEP(obj)
      bin = SimpleCABACMex('decodeBinEP', obj.cabac_handle);
    end

    function encodeBinProb(obj, binValue, percent)
      obj.gateway('encodeBinProb', binValue, percent);
    end

    function bin = decodeBinProb(obj, percent)
      bin = SimpleCABACMex('decodeBinProb', obj.cabac_handle, percent);
    end

    % ---- state traces: trace is 5 x N (bin, state before, mps before, state after, mps after), stats 128 x 128 -----
    function [trace, stats] = getEncoderStats(obj, ctxID)
      [trace, stats] = SimpleCABACMex('getEncoderStats', obj.cabac_handle, ctxID);
    end

    function [trace, stats] = getDecoderStats(obj, ctxID)
      [trace, stats] = SimpleCABACMex('getDecoderStats', obj.cabac_handle, ctxID);
    end

    % ---- batches: B x S uint8 matrices, column s = stream s ------------------------------------------------------
    % (a uint16 ctx matrix -- 65535 = bypass bin -- takes the up to 999 contexts of the reference's coder, CommonDef.h:56)
    function [payload, offsets] = encodeBatch(obj, bins, ctx, nbins)
      if ~isa(ctx, 'uint16'), ctx = uint8(ctx); end
      args = {uint8(bins), ctx};
      if nargin >= 4
        args{end + 1} = uint32(nbins);
      end
      [payload, offsets] = SimpleCABACMex('encodeBatch', obj.cabac_handle, args{:});
    end

    function [bins, status] = decodeBatch(obj, payload, offsets, ctx, nbins)
      if ~isa(ctx, 'uint16'), ctx = uint8(ctx); end
      args = {uint8(payload), double(offsets), ctx};
      if nargin >= 5
        args{end + 1} = uint32(nbins);
      end
      [bins, status] = SimpleCABACMex('decodeBatch', obj.cabac_handle, args{:});
    end
  end

  methods
    % ---- symbols: R x S uint8 matrices, column s = stream s -------------------------------------------------------
    function [payload, offsets, nbins] = encodeSymbols(obj, sym, params, nsym)
      args = {uint8(sym), double(params)};
      if nargin >= 4
        args{end + 1} = uint32(nsym);
      end
      [payload, offsets, nbins] = SimpleCABACMex('encodeSymbols', obj.cabac_handle, args{:});
    end

    function [payload, offsets, nbins, ctxHist, ctxCost, symBits] = encodeSymbolsStats(obj, sym, params, nsym)
      args = {uint8(sym), double(params)};
      if nargin >= 4
        args{end + 1} = uint32(nsym);
      end
      [payload, offsets, nbins, ctxHist, ctxCost, symBits] = SimpleCABACMex('encodeSymbolsStats', obj.cabac_handle, args{:});
    end

    function [sym, status] = decodeSymbols(obj, payload, offsets, symbolsPerStream, params, nsym)
      args = {uint8(payload), double(offsets), double(symbolsPerStream), double(params)};
      if nargin >= 6
        args{end + 1} = uint32(nsym);
      end
      [sym, status] = SimpleCABACMex('decodeSymbols', obj.cabac_handle, args{:});
    end
  end

  methods (Access = private)
    function gateway(obj, command, varargin)
      % commands without a result
      SimpleCABACMex(command, obj.cabac_handle, varargin{:});
    end
  end
end
