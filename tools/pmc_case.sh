#!/bin/bash
# Counters of the coder kernels for ONE bench.py configuration (GPU box, from the repo root):
#   bash tools/pmc_case.sh <outdir> <bench.py args...>
# One rocprofv3 pass per counter group, --kernel-trace only beside --pmc.  Summarise with tools/pmc_case_table.py.
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # name, counters...
  name=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify $BARGS > "$OUT/$name.log" 2>&1
  echo "$OUT $name rc=$?"
}
BARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run grbm GRBM_GUI_ACTIVE
if [ "${PMC_MEM:-0}" = "1" ]; then
  run fetch FETCH_SIZE
  run write WRITE_SIZE
fi
