#!/usr/bin/env python3
"""Static checks of the kernels' gfx950 ISA for the three hazards the hand-written parts of the engine manage themselves
(inline assembly is opaque to the compiler's hazard recogniser, so nothing else looks at these):

  A  LDS read in flight.  No instruction may touch the destination registers of an LDS read before the wave has waited on
     lgkmcnt.  The compiler guarantees it for the loads it knows; the latency-oriented encoder issues its state pass in
     inline assembly (cabac_kernels.hip: EncLatPipe), where a register copy or spill the compiler put between the read and
     the wait would read a register whose data has not arrived -- silently, and only sometimes.
  B  Wide-store data.  A vector-memory store of more than 64 bits (global / buffer / flat / scratch ..._dwordx3/x4) reads
     its data registers late: a VALU instruction that WRITES one of them must not follow within two wait states (an
     instruction = 1, `s_nop N` = N + 1; LLVM's 12-dword store hazard, two wait states on gfx940+; LDS stores have none).  The decode kernels issue their output stores by hand with `s_nop 1`
     behind each (cabac_kernels.hip: store_out, cabac_pair.hip: store_block); without it one full-size run once decoded
     wrong bytes with clean status words.
  C  SDWA partial write.  A VALU result written with a destination select other than DWORD (dst_sel:BYTE_n / WORD_n) must
     not be read by the very next VALU instruction (cabac_kernels.hip: EncLatPipe keeps an address add or an `s_nop 0`
     in between; cabac_lane.h: put_out_byte's results are consumed later).

    python tools/check_inflight_regs.py [file.s ...]   (without files: compiles every .hip of isscabac_amd/csrc to ISA)

Prints the number of sites checked per kernel and hazard; exit status 1 and the offending lines on a violation."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_SOURCES = ["cabac_kernels.hip", "cabac_pair.hip", "cabac_wide.hip", "cabac_stream.hip", "cabac_symbols.hip", "cabac_multi.hip", "cabac_order.hip"]


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


WIDE_STORE = re.compile(r"^(global|buffer|flat|scratch)_store_dwordx[34]$")


def wide_store_data(mnem, ops):
    """data registers of a store of more than 64 bits (else the empty set)"""
    if not WIDE_STORE.match(mnem):
        return set()
    if mnem.startswith("buffer"):  # buffer_store_dwordx4 data, vaddr, srsrc, soffset
        return regs_of(ops[0]) if ops else set()
    return regs_of(ops[1]) if len(ops) > 1 else set()  # global / flat / scratch: addr, data[, saddr]


def valu_written(mnem, ops):
    """registers a VALU instruction writes (its first operand; v_cmp with an SGPR / VCC destination writes none)"""
    if not mnem.startswith("v_") or not ops:
        return set()
    if mnem.startswith(("v_cmp", "v_cmpx")):
        return set()
    return regs_of(ops[0])  # (a second destination, where there is one, is an SGPR pair / VCC)


def valu_read(mnem, ops):
    """registers a VALU instruction reads: every VGPR operand but the destination (dst_unused:UNUSED_PRESERVE reads it too)"""
    if not mnem.startswith("v_") or not ops:
        return set()
    src = ops if mnem.startswith(("v_cmp", "v_cmpx")) else ops[1:]
    r = set()
    for t in src:
        r |= regs_of(t)
    if "UNUSED_PRESERVE" in " ".join(ops):
        r |= regs_of(ops[0])
    return r


def check(text):
    """-> (violations, counts).  counts maps kernel -> LDS reads checked (hazard A: the historical return value); the other
    two hazards' site counts are in check.sites after the call ({kernel: [A, B, C]})."""
    bad, counts = [], {}
    sites = {}
    kernel = None
    pending = []     # A: (line number, text, dest regs) of LDS reads not yet waited for
    store = None     # B: [line number, text, data regs, wait states seen since]
    partial = None   # C: (line number, text, dest regs) when the previous instruction was an SDWA partial write
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, pending, store, partial = m.group(1), [], None, None
            sites.setdefault(kernel, [0, 0, 0])
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue  # (labels too: the text is followed in layout order, which is how the rare branches fall through)
        mnem, ops = operands(s)
        if kernel is None:
            continue
        # ---- A
        if mnem == "s_waitcnt" and "lgkmcnt" in s:
            pending = []
        else:
            used = set()
            for t in ops:
                used |= regs_of(t)
            for (pl, pm, dest) in pending:
                if used & dest:
                    bad.append("A %s: line %d `%s` touches v%s, destination of `%s` (line %d), before any lgkmcnt wait"
                               % (kernel, ln, s, sorted(used & dest), pm, pl))
            if mnem.startswith("ds_read") and ops:
                pending.append((ln, s, regs_of(ops[0])))
                counts[kernel] = counts.get(kernel, 0) + 1
                sites[kernel][0] += 1
        # ---- B
        if store is not None:
            if store[3] < 2:
                hit = valu_written(mnem, ops) & store[2]
                if hit:
                    bad.append("B %s: line %d `%s` writes v%s, data of `%s` (line %d), %d wait state(s) behind the store"
                               % (kernel, ln, s, sorted(hit), store[1], store[0], store[3]))
            nop = re.fullmatch(r"s_nop\s+(\d+)", s.split(";")[0].strip())
            store[3] += int(nop.group(1)) + 1 if nop else 1
            if store[3] >= 2:
                store = None
        data = wide_store_data(mnem, ops)
        if data:
            store = [ln, s, data, 0]
            sites[kernel][1] += 1
        # ---- C
        if partial is not None:
            hit = valu_read(mnem, ops) & partial[2]
            if hit and not (mnem == "s_nop"):
                bad.append("C %s: line %d `%s` reads v%s right behind the partial write `%s` (line %d)"
                           % (kernel, ln, s, sorted(hit), partial[1], partial[0]))
            partial = None
        msel = re.search(r"dst_sel:(\w+)", s)
        if mnem.startswith("v_") and msel and msel.group(1) != "DWORD" and not mnem.startswith(("v_cmp", "v_cmpx")):
            partial = (ln, s, regs_of(ops[0]) if ops else set())
            sites[kernel][2] += 1
        # ---- control flow: nothing falls through an unconditional branch
        if mnem.startswith(("s_branch", "s_endpgm", "s_setpc")):
            pending, store, partial = [], None, None
    check.sites = sites
    return bad, counts


check.sites = {}


def compile_isa(src, outdir):
    sys.path.insert(0, ROOT)
    from isscabac_amd import build
    out = os.path.join(outdir, os.path.basename(src) + ".s")
    subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    return out


def main():
    rc = 0
    with tempfile.TemporaryDirectory() as d:
        files = sys.argv[1:]
        if not files:
            srcs = [os.path.join(ROOT, "isscabac_amd", "csrc", f) for f in HIP_SOURCES]
            with ThreadPoolExecutor(4) as ex:
                files = list(ex.map(lambda s: compile_isa(s, d), srcs))
        for f in files:
            bad, _ = check(open(f).read())
            for k, (na, nb, nc) in sorted(check.sites.items()):
                if na or nb or nc:
                    print("%6d LDS reads, %5d wide stores, %5d partial writes checked  %s" % (na, nb, nc, k))
            for b in bad:
                print("VIOLATION", b)
            rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
