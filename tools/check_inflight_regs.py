#!/usr/bin/env python3
"""Static check of the kernels' ISA: no instruction may touch the destination registers of an LDS read before the wave has
waited on lgkmcnt.  The compiler guarantees this for the loads it knows; the latency-oriented encoder issues its state pass
in inline assembly (cabac_kernels.hip: EncLatPipe), where a register copy or spill the compiler put between the read and
the wait would read a register whose data has not arrived -- silently, and only sometimes.

    python tools/check_inflight_regs.py [file.s]      (without a file: compiles isscabac_amd/csrc/cabac_kernels.hip to ISA)

Prints the number of LDS reads checked per kernel; exit status 1 and the offending lines on a violation."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    """v12 -> {12}; v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    ops = [t.strip() for t in re.split(r",|\s", parts[1]) if t.strip()]
    return parts[0], ops


def check(text):
    bad, counts = [], {}
    kernel = None
    pending = []  # (line number, mnemonic, dest regs)
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, pending = m.group(1), []
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue  # (labels too: the text is followed in layout order, which is how the rare branches fall through)
        mnem, ops = operands(s)
        if mnem == "s_waitcnt" and "lgkmcnt" in s:
            pending = []
            continue
        used = set()
        for t in ops:
            used |= regs_of(t)
        for (pl, pm, dest) in pending:
            if used & dest:
                bad.append("%s: line %d `%s` touches v%s, destination of `%s` (line %d), before any lgkmcnt wait"
                           % (kernel, ln, s, sorted(used & dest), pm, pl))
        if mnem.startswith("ds_read") and ops:
            pending.append((ln, s, regs_of(ops[0])))
            counts[kernel] = counts.get(kernel, 0) + 1
        if mnem.startswith(("s_branch", "s_endpgm", "s_setpc")):
            pending = []  # nothing falls through an unconditional branch
    return bad, counts


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        sys.path.insert(0, ROOT)
        from isscabac_amd import build
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                   "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "isscabac_amd", "csrc", "cabac_kernels.hip")],
                                  stderr=subprocess.DEVNULL)
            text = open(out).read()
    bad, counts = check(text)
    for k, n in sorted(counts.items()):
        print("%6d LDS reads checked  %s" % (n, k))
    for b in bad:
        print("VIOLATION", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
