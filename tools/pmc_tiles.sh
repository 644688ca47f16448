set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 32 64; do
  export ISSCABAC_TILE=$t
  OUT=gpurun_out/pmc_v2_t$t; mkdir -p $OUT
  for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "tcc TCC_HIT_sum TCC_MISS_sum"; do
    set -- $grp; name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/$name.log 2>&1
  done
done
echo done
