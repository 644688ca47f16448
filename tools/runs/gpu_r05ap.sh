#!/bin/bash
# round 5, call ap: does the row pitch matter?  bins / ctx rows at 4096 bytes (every row of a wave's tile starts on the same 4 KiB grid)
# against 4096 + 64 / + 128 / + 256 bytes
set -u
OUT=gpurun_out/r05ap; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pad in 0 64 128 256 0 128; do
  echo "== pitch pad $pad" >> $OUT/pitch.txt
  ISSCABAC_BENCH_PITCH_PAD=$pad timeout 400 python3 tools/s_sweep.py --out /tmp/s.json 2>&1 | grep "1048576\|131072\|65536 streams" >> $OUT/pitch.txt
done
cat $OUT/pitch.txt
