#!/usr/bin/env python3
"""One line per coder kernel from a tools/pmc_case.sh directory.
usage: python tools/pmc_case_table.py <dir> <streams> <bins> [kernel-trace csv for durations]"""
import json
import subprocess
import sys

d = json.loads(subprocess.check_output([sys.executable, "tools/pmc_summary.py", sys.argv[1]]))
S, B = int(sys.argv[2]), int(sys.argv[3])
steps = (S // 64) * B  # wave-bins per launch (one wave of streams x bins: the pair kernels run two waves per wave of streams)
for k, x in d.items():
    if k.startswith(("k_encode<", "k_decode<", "k_encode_pair", "k_decode_pair")):
        g = lambda n: x.get(n, float("nan"))
        wc = g("SQ_WAVE_CYCLES")
        print("%-22s per wave-bin: VALU %.1f SALU %.1f LDS %.2f VMEM %.2f | LDS conflict %.0f%% | wave cyc/bin %.0f: active %.0f%% "
              "(valu %.0f%% lds %.0f%%) wait_any %.0f%% wait_inst %.0f%% (lds %.0f%%) | GUI_ACTIVE/8 %.3g cycles | FETCH %.3f GB WRITE %.3f GB"
              % (k, g("SQ_INSTS_VALU") / steps, g("SQ_INSTS_SALU") / steps, g("SQ_INSTS_LDS") / steps,
                 (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) / steps,
                 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4 * wc / steps,
                 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc,
                 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc,
                 g("GRBM_GUI_ACTIVE") / 8, g("FETCH_SIZE") * 1024 / 1e9, g("WRITE_SIZE") * 1024 / 1e9))
