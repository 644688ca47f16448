#!/usr/bin/env python3
"""one line per coder kernel from a tools/pmc_quick.sh directory (config 3: 16384 waves x 4096 bins)"""
import json, subprocess, sys
d = json.loads(subprocess.check_output([sys.executable, "tools/pmc_summary.py", sys.argv[1]]))
steps = 16384 * 4096
for k, x in d.items():
    if k.startswith(("k_encode<", "k_decode<")):  # the coder kernels proper (not k_decode_exact: returns at once)
        g = lambda n: x.get(n, float("nan"))
        print("%-14s FETCH %.2f GB WRITE %.2f GB | per wave-bin: VALU %.1f SALU %.1f LDS %.2f | LDS conflict %.0f%% | "
              "wave cyc/bin %.0f active %.0f%% wait_any %.0f%% wait_inst %.0f%% | waves/SIMD avg %.2f" % (
                  k, g("FETCH_SIZE") * 1024 / 1e9, g("WRITE_SIZE") * 1024 / 1e9, g("SQ_INSTS_VALU") / steps,
                  g("SQ_INSTS_SALU") / steps, g("SQ_INSTS_LDS") / steps, 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
                  4 * g("SQ_WAVE_CYCLES") / steps, 100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                  100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                  g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES") * 4 / 4 if g("SQ_BUSY_CYCLES") else float("nan")))
