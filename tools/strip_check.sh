#!/bin/bash
# usage: bash tools/strip_check.sh <outdir>   -- device ISA of every .hip source (default build) into <outdir>/*.s, normalised for diffing
# (two trees produce the same kernels iff `diff -r` of two such directories is empty)
set -e
O=$1; mkdir -p $O
for f in cabac_kernels cabac_pair cabac_wide cabac_stream cabac_symbols cabac_multi cabac_order; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -x hip isscabac_amd/csrc/$f.hip -o $O/$f.raw.s 2>/dev/null
    # drop what depends on line numbers / paths only: debug-ish comments, .file / .loc, the ident strings
    grep -v '^\s*;\|^\s*\.file\|^\s*\.loc\|^\s*\.ident\|\.amdgcn_target\|^\s*\.section\s*\.AMDGPU\.csdata' $O/$f.raw.s | sed 's/;.*$//; s/__hip_cuid_[0-9a-f]*/__hip_cuid_X/g' > $O/$f.s; rm $O/$f.raw.s ) &
done
wait
ls -la $O
