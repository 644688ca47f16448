#!/bin/bash
# round 6, run y2: the decoder pair pinned like the encoder's, and used up to 2 / 4 / 8 waves of streams per CU
O=gpurun_out/r06y2; mkdir -p $O
ISSCABAC_LIB=$PWD/tools/ab/variant_dpairpin.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k latency > $O/gputests.txt 2>&1; tail -2 $O/gputests.txt
for r in 1 2; do
echo "== product library"; python tools/s_sweep.py --streams 1048576,131072,65536,32768,16384 2>&1 | grep -v amdgpu
for w in 2 4 8; do
  echo "== decoder pair pinned, up to $w waves of streams per CU"; ISSCABAC_EXP_DEC_PAIR_WAVES=$w ISSCABAC_LIB=$PWD/tools/ab/variant_dpairpin.so python tools/s_sweep.py --streams 1048576,131072,65536,32768,16384 2>&1 | grep -v amdgpu
done; done > $O/sweep.txt 2>&1; cat $O/sweep.txt
