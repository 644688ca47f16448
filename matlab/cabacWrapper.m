classdef cabacWrapper < handle
%-------------------------------------------------------------------------%
% MATLAB class wrapper around SimpleCABACMex, the MEX gateway of the MI355X
% batched CABAC engine.  Same constructor and methods as the reference class
% (christianrohlfing/ISScabac, CABAC/cabacWrapper.m) so that cabacDemo.m and
% ISS/+coder/cabac{Encode,Decode}.m run unchanged; encodeBatch / decodeBatch
% are additive and code one independent stream per matrix column on the GPU.
%-------------------------------------------------------------------------%
  properties
    bitStreamName;
    cabac_handle;
    contextModelInitOptions; % initByProb: p(0) per context; initByState: [ctxID mps state ...]
  end
  methods
    function obj = cabacWrapper(cm,fn,initByProb)
      if nargin < 3, initByProb=1; end
      if ischar(fn)
        obj.bitStreamName = fn;
      else
        error('bitStreamName should be string')
      end
      if ismatrix(cm)
        obj.contextModelInitOptions = cm;
      else
        error('context model initialization options should be like this: [ctxID mps probIndex, ctxID mps probIndex, ...]')
      end
      obj.init(initByProb)
    end
    function init(obj,initByProb)
      if nargin < 2, initByProb=1; end
      if initByProb
        obj.cabac_handle = SimpleCABACMex('initByProb', obj.bitStreamName,obj.contextModelInitOptions);
      else
        obj.cabac_handle = SimpleCABACMex('initByState', obj.bitStreamName,obj.contextModelInitOptions);
      end
    end
    function encodeStart(obj)
      SimpleCABACMex('encodeStart', obj.cabac_handle);
    end
    function encodeBin(obj,binValue, ctxID)
      if binValue == 1 || binValue == 0
        SimpleCABACMex('encodeBin', obj.cabac_handle, binValue, ctxID);
      else
        error('bin value to be encoded should either be 1 or 0');
      end
    end
    function encodeFinish(obj)
      SimpleCABACMex('encodeFinish', obj.cabac_handle);
    end
    function decodeStart(obj)
      SimpleCABACMex('decodeStart', obj.cabac_handle);
    end
    function decodedBin = decodeBin(obj, ctxID)
      decodedBin = SimpleCABACMex('decodeBin', obj.cabac_handle, ctxID);
    end
    function decodeFinish(obj)
      SimpleCABACMex('decodeFinish', obj.cabac_handle);
    end
    function [trace, stats] = getEncoderStats(obj, ctxID)
      [trace, stats] = SimpleCABACMex('getEncoderStats',obj.cabac_handle,ctxID);
    end
    function [trace, stats] = getDecoderStats(obj, ctxID)
      [trace, stats] = SimpleCABACMex('getDecoderStats',obj.cabac_handle,ctxID);
    end
    function [bits] = getNumBits(obj)
      [bits] = SimpleCABACMex('getNumBits',obj.cabac_handle);
    end
    % ---- batch extension: one independent stream per column, ctx 255 = bypass bin ----
    function [payload, offsets] = encodeBatch(obj, bins, ctx, nbins)
      if nargin < 4
        [payload, offsets] = SimpleCABACMex('encodeBatch', obj.cabac_handle, uint8(bins), uint8(ctx));
      else
        [payload, offsets] = SimpleCABACMex('encodeBatch', obj.cabac_handle, uint8(bins), uint8(ctx), uint32(nbins));
      end
    end
    function [bins, status] = decodeBatch(obj, payload, offsets, ctx, nbins)
      if nargin < 5
        [bins, status] = SimpleCABACMex('decodeBatch', obj.cabac_handle, uint8(payload), double(offsets), uint8(ctx));
      else
        [bins, status] = SimpleCABACMex('decodeBatch', obj.cabac_handle, uint8(payload), double(offsets), uint8(ctx), uint32(nbins));
      end
    end
    function delete(obj)
      if ~isempty(obj.cabac_handle)
        SimpleCABACMex('delete', obj.cabac_handle);
        obj.cabac_handle = [];
      end
    end
  end
end
