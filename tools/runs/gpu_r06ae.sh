#!/bin/bash
# round 6, run ae: workgroups of fewer than four waves start four and keep theirs on the SIMDs a ticket names (k_encode / k_decode above 64 contexts) -- parity, sizes by hand
O=gpurun_out/r06ae; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2; do for w in 4 3 2 1; do echo "== ISSCABAC_EXP_WPB=$w (pinned below four)"; ISSCABAC_EXP_WPB=$w ISSCABAC_LIB=$PWD/tools/ab/variant_wpbexp.so python tools/ctx_rates.py 2>&1 | grep "127\|190\|253"; done; done > $O/wpb_pinned.txt 2>&1; cat $O/wpb_pinned.txt
