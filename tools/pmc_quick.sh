#!/bin/bash
# usage: bash tools/pmc_quick.sh <outdir>   (GPU box, repo root)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "sq3 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LEVEL_WAVES"; do
  set -- $grp; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/$name.log 2>&1
done
echo done
