#!/usr/bin/env python3
"""Partial preprocessor used ONCE in round 6 to take the build-time experiment switches out of the shipped sources
(VERDICT r5 item 7): every `#if` / `#ifdef` / `#ifndef` / `#elif` whose condition mentions only the switches listed in VALUES
(and literals) is decided with the switch's default value, the dead branch is dropped, the `#ifndef X / #define X v / #endif`
default blocks of those switches go, and remaining uses of a switch in the code are replaced by its value.  Conditions that
mention anything else (`__HIP_DEVICE_COMPILE__`, the tunables that stay, include guards) are left alone.

    python tools/strip_switches.py isscabac_amd/csrc/*.h isscabac_amd/csrc/*.hip isscabac_amd/csrc/*.cpp

The default build's ISA before and after is compared by tools/strip_check.sh; what the switches were, and the measurements
made with them, stay in profiles/EXPERIMENTS.md and in the history (the last commit with the switches is named there).
"""
import re
import sys

VALUES = {
    "ISSCABAC_DEC_LAT_BARRIER": "1", "ISSCABAC_DEC_LAT_EXP": "0", "ISSCABAC_DEC_LAT_MAD": "0", "ISSCABAC_DEC_LDS_PAD": "0",
    "ISSCABAC_DEC_LPR": "8", "ISSCABAC_DEC_LPR_PERM": "8", "ISSCABAC_DEC_NOSTORE": "0", "ISSCABAC_DEC_NT": "1",
    "ISSCABAC_DEC_NT_ST": "1", "ISSCABAC_DEC_OCC": "4", "ISSCABAC_DEC_PAIR": "1", "ISSCABAC_DEC_ROLLED": "0",
    "ISSCABAC_DEC_SHL16": "0", "ISSCABAC_DEC_SPLIT_TAB": "0", "ISSCABAC_DEC_TAB_REPL": "0", "ISSCABAC_DEC_TILE_PAIR": "0",
    "ISSCABAC_DPAIR_ALT_ROLES": "0", "ISSCABAC_DPAIR_PRIO_B": "0", "ISSCABAC_DPAIR_TILE_PAIR": "0", "ISSCABAC_ENC_BINADDR": "0",
    "ISSCABAC_ENC_DEPTH2": "0", "ISSCABAC_ENC_LINE_B": "1", "ISSCABAC_ENC_LINE_C": "1", "ISSCABAC_ENC_LPR_B": "4",
    "ISSCABAC_ENC_LPR_C": "4", "ISSCABAC_ENC_MIXED2": "1", "ISSCABAC_ENC_NOSTORE": "0", "ISSCABAC_ENC_NT": "0",
    "ISSCABAC_ENC_PAIR": "1", "ISSCABAC_ENC_PICK_VCC": "2", "ISSCABAC_ENC_ROLLED": "0", "ISSCABAC_ENC_SPLIT_FETCH": "0",
    "ISSCABAC_ENC_TILE_PAIR": "0", "ISSCABAC_EXP_CLK": "0", "ISSCABAC_EXP_INPLACE": "0", "ISSCABAC_EXP_NOLOAD": "0",
    "ISSCABAC_EXP_NOPAYLOAD": "0", "ISSCABAC_EXP_PRIO": "0", "ISSCABAC_EXP_SB32": "0", "ISSCABAC_EXP_SPREAD_ST": "0",
    "ISSCABAC_EXP_STEP": "0", "ISSCABAC_FLUSH_FN": "enc_flush16_wave", "ISSCABAC_PAIR_ALT_ROLES": "0", "ISSCABAC_PAIR_EXP": "0",
    "ISSCABAC_PAIR_NOSYNC": "0", "ISSCABAC_PAIR_PRIO_A": "1", "ISSCABAC_PAIR_PRIO_B": "0", "ISSCABAC_PAIR_TILE_PAIR": "1",
    "ISSCABAC_SHL16": "1", "ISSCABAC_SYM_PICK_VCC": "0",
}
TOKEN = re.compile(r"\b(ISSCABAC_[A-Z0-9_]+)\b")


def decide(expr):
    """-> True / False, or None when the condition mentions something this tool does not own"""
    e = re.sub(r"//.*$", "", expr).strip()
    e = re.sub(r"/\*.*?\*/", "", e)
    names = set(re.findall(r"\b[A-Za-z_][A-Za-z0-9_]*\b", e)) - {"defined"}
    if not names or not names <= set(VALUES):
        return None
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", "1", e)
    e = TOKEN.sub(lambda m: VALUES[m.group(1)], e)
    e = re.sub(r"(\d+)[uU]\b", r"\1", e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    try:
        return bool(eval(e, {"__builtins__": {}}, {}))
    except Exception:
        return None


def strip(text):
    lines = text.split("\n")
    out = []
    # stack of frames: dict(owned, taken, live, parent_live)
    stack = []
    i = 0

    def live():
        return all(f["live"] for f in stack)

    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$", ln)
        if not m:
            if live():
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            # the default block of an owned switch:  #ifndef X / #define X v / #endif
            if kind == "ifndef":
                name = rest.strip().split()[0] if rest.strip() else ""
                if name in VALUES and i + 2 < len(lines) and re.match(r"\s*#\s*define\s+%s\b" % name, lines[i + 1]) \
                        and re.match(r"\s*#\s*endif\b", lines[i + 2]):
                    i += 3
                    continue
            if kind == "if":
                d = decide(rest)
            else:
                name = rest.strip().split()[0]
                d = None if name not in VALUES else (kind == "ifdef")
            if d is None:
                stack.append({"owned": False, "live": True, "taken": True})
                if live():
                    out.append(ln)
            else:
                stack.append({"owned": True, "live": d, "taken": d})
        elif kind == "elif":
            f = stack[-1]
            if f["owned"]:
                if f["taken"]:
                    f["live"] = False
                else:
                    d = decide(rest)
                    if d is None:
                        raise SystemExit("cannot decide #elif %s" % rest)
                    f["live"] = d
                    f["taken"] = d
            elif live():
                out.append(ln)
        elif kind == "else":
            f = stack[-1]
            if f["owned"]:
                f["live"] = not f["taken"]
                f["taken"] = True
            elif live():
                out.append(ln)
        else:  # endif
            f = stack.pop()
            if not f["owned"] and live():
                out.append(ln)
        i += 1
    assert not stack
    res = "\n".join(out)
    # remaining uses in the code
    res = TOKEN.sub(lambda m: VALUES.get(m.group(1), m.group(1)), res)
    return res


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        if new != src:
            open(path, "w").write(new)
            print("stripped", path, src.count("\n") - new.count("\n"), "lines shorter")
