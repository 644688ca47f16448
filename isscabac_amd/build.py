"""Build the HIP engine in-tree: isscabac_amd/lib/libisscabac_hip.so (gfx950 only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ISSCABAC_LIB selects another build of the same ABI (A/B experiments: tools/ab_bench.py)
LIB = os.environ.get("ISSCABAC_LIB") or os.path.join(HERE, "lib", "libisscabac_hip.so")
SOURCES = ["cabac_kernels.hip", "cabac_pair.hip", "cabac_wide.hip", "cabac_stream.hip", "cabac_symbols.hip", "cabac_multi.hip", "cabac_order.hip", "cabac_capi.cpp", "cabac_container.cpp", "cabac_host.cpp"]
HOST_ONLY = ["cabac_host.cpp"]  # the host twins of the serial entry points (isscabac_cpu_*)
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


def _compile_objects(objdir, flags, verbose=False, csrc=None):
    """one object per source, compiled side by side (the whole library in one hipcc call takes a minute, the slowest file 25 s);
    an object is reused while it is newer than its source and every header"""
    from concurrent.futures import ThreadPoolExecutor
    csrc = csrc or CSRC
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs, objs = [], []
    for f in SOURCES:
        src, obj = os.path.join(csrc, f), os.path.join(objdir, f + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(src)):
            continue
        if f in HOST_ONLY:  # plain C++: the per-lane headers compile without the HIP runtime (no device pass, no hip_runtime.h)
            jobs.append([hipcc(), "-O2", "-std=c++17", "-fPIC", "-Wall"] + flags + ["-c", "-x", "c++", src, "-o", obj])
            continue
        lang = ["-x", "hip"] if f.endswith(".hip") else []
        jobs.append([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall"] + flags + ["-c"] + lang + [src, "-o", obj])
    if verbose:
        for j in jobs:
            print(" ".join(j))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for rc, j in zip(ex.map(subprocess.call, jobs), jobs):
            if rc:
                raise subprocess.CalledProcessError(rc, j)
    return objs


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> lib/libisscabac_hip.so.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(os.path.dirname(LIB), "obj")
    if force and os.path.isdir(objdir):
        shutil.rmtree(objdir)
    objs = _compile_objects(objdir, [], verbose)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag, defines=(), patches=()):
    """The library with extra -D flags (the tunables of DESIGN.md) and / or patches applied to a scratch copy of the sources
    -> tools/ab/variant_<tag>.so, for same-box A/B runs (tools/ab_bench.py).  patches: files of tools/ab_patches/ (`patch -p1`
    against a tree that holds csrc/): timing-only builds -- no input loads, no output stores, ... -- live there, not behind
    switches in the shipped sources."""
    import tempfile
    out = os.path.join(HERE, "..", "tools", "ab", "variant_%s.so" % tag)  # A/B builds live beside the tools, not beside the product library
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objdir = os.path.join(os.path.dirname(out), "obj_%s" % tag)
    if os.path.isdir(objdir):
        shutil.rmtree(objdir)
    with tempfile.TemporaryDirectory() as tmp:
        csrc = CSRC
        if patches:
            # the sources include ../../include/isscabac.h: keep that shape in the scratch tree
            csrc = os.path.join(tmp, "isscabac_amd", "csrc")
            shutil.copytree(CSRC, csrc)
            shutil.copytree(os.path.join(HERE, "..", "include"), os.path.join(tmp, "include"))
            for p in patches:
                subprocess.check_call(["patch", "-s", "-p1", "-d", os.path.join(tmp, "isscabac_amd"), "-i", os.path.abspath(p)])
        objs = _compile_objects(objdir, ["-D" + d for d in defines], csrc=csrc)
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    shutil.rmtree(objdir)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
