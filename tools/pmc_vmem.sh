#!/bin/bash
# Issue / wait split of the coder kernels for one build (GPU box, repo root):
#   ISSCABAC_LIB=... bash tools/pmc_vmem.sh <outdir> [bench.py args]
# (TCP_* / TA_* counters crash rocprofv3 on this image and hang the pass: SQ counters only, every pass under a timeout)
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BARGS="$*"
run() { name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify $BARGS > "$OUT/$name.log" 2>&1
  echo "$OUT $name rc=$?"; }
run g1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAIT_ANY
