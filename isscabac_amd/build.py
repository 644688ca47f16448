"""Build the HIP engine in-tree: isscabac_amd/lib/libisscabac_hip.so (gfx950 only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ISSCABAC_LIB selects another build of the same ABI (A/B experiments: tools/ab_bench.py)
LIB = os.environ.get("ISSCABAC_LIB") or os.path.join(HERE, "lib", "libisscabac_hip.so")
SOURCES = ["cabac_kernels.hip", "cabac_pair.hip", "cabac_wide.hip", "cabac_stream.hip", "cabac_symbols.hip", "cabac_multi.hip", "cabac_order.hip", "cabac_capi.cpp", "cabac_container.cpp"]
HEADERS = ["cabac_lane.h", "cabac_wave.h", "cabac_symbols.h", "cabac_internal.h", os.path.join("..", "..", "include", "isscabac.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X engine cannot be built on this machine")
    return exe


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> lib/libisscabac_hip.so.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag, defines):
    """Same sources with extra -D flags -> tools/ab/variant_<tag>.so (A/B experiments, see tools/ab_bench.py)."""
    out = os.path.join(HERE, "..", "tools", "ab", "variant_%s.so" % tag)  # A/B builds live beside the tools, not beside the product library
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", out] + \
          ["-D" + d for d in defines] + [os.path.join(CSRC, f) for f in SOURCES]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
