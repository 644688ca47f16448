#!/bin/bash
# round 6, run w: workgroup size by hand at 190 contexts, repeated (the decoder's four-wave workgroups measured 3.69 ms in one run, 4.0-4.1 in another)
O=gpurun_out/r06w; mkdir -p $O
for r in 1 2 3; do for w in 3 4 1; do echo "== ISSCABAC_EXP_WPB=$w"; ISSCABAC_EXP_WPB=$w ISSCABAC_LIB=$PWD/tools/ab/variant_wpbexp.so python tools/ctx_rates.py 2>&1 | grep "190\|253"; done; done > $O/wpb_190.txt 2>&1
cat $O/wpb_190.txt
