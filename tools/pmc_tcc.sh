#!/bin/bash
# L2 (TCC) request counters of the coder kernels for one build (GPU box, repo root): request sizes, DRAM share, stalls.
#   ISSCABAC_LIB=... bash tools/pmc_tcc.sh <outdir> [bench.py args]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BARGS="$*"
run() { name=$1; shift
  timeout 60 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify $BARGS > "$OUT/$name.log" 2>&1
  echo "$OUT $name rc=$?"; }
run t1 TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_DRAM
run t2 TCC_REQ TCC_HIT TCC_MISS TCC_STREAMING_REQ TCC_READ TCC_WRITE
run t3 TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL TCC_BUSY TCC_CYCLE
