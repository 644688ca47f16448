#!/bin/bash
# Counters of the symbol-level kernels on tools/symbol_rate.py's shapes (GPU box, from the repo root):
#   bash tools/pmc_symbols.sh <outdir>        then: python tools/pmc_summary.py <outdir>
# One rocprofv3 pass per counter group, --kernel-trace only beside --pmc.
set -u
OUT=$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/symbol_rate.py > "$OUT/$name.log" 2>&1
  echo "$OUT $name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
