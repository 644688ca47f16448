mkdir -p gpurun_out/r06r
for w in 1 2 3 4; do echo "== ISSCABAC_EXP_WPB=$w"; ISSCABAC_EXP_WPB=$w ISSCABAC_LIB=$PWD/tools/ab/variant_wpbexp.so python tools/ctx_rates.py 2>&1 | grep "127\|190\|253"; done > gpurun_out/r06r/wpb_sweep.txt 2>&1
cat gpurun_out/r06r/wpb_sweep.txt
