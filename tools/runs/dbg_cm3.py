import sys, os, subprocess
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import numpy as np, torch
    from isscabac_amd import Engine
    from oracle.pyoracle import Oracle
    eng, orc = Engine(0), Oracle()
    S, B, C, ep, tuning = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5]
    eng.set_tuning(tuning)
    bins, ctx = orc.synth(0x99, 0, S, B, C, ep)
    init = np.zeros(C, np.uint8)
    wp, wo = orc.encode_batch(bins, ctx, init)
    d, st = eng.decode(wp, wo, ctx, init)
    print("  dec ok", (d == bins).all(), (st == 0).all(), eng.last_instance("decode"), flush=True)
    sys.exit(0)
for case in [(130, 500, 128, 1, "throughput"), (130, 500, 128, 0, "throughput"), (128, 500, 128, 1, "throughput"), (64, 512, 128, 1, "throughput"),
             (64, 128, 128, 1, "throughput"), (64, 2000, 100, 1, "throughput"), (130, 500, 100, 1, "throughput"), (130, 500, 128, 1, "latency_single"),
             (130, 500, 190, 1, "throughput"), (64, 16, 128, 1, "throughput")]:
    r = subprocess.run([sys.executable, __file__] + [str(x) for x in case], capture_output=True, text=True)
    print(case, "rc", r.returncode, r.stdout.strip()[-80:], (r.stderr.strip().splitlines() or [""])[-1][-100:], flush=True)
