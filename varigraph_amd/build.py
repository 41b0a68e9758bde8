"""Build the native libraries in-tree.

  varigraph_amd/libvgmi.so   HIP kernels + C ABI (include/vgmi.h), hipcc --offload-arch=gfx950
  oracle/liboracle.so        the CPU checker (test infrastructure; see oracle/vg_oracle.h)
  oracle/_ref/*              the real reference, only when /root/reference is present

hipcc cross-compiles without a GPU, so this runs in the build container too.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "varigraph_amd", "csrc")
LIB = os.path.join(ROOT, "varigraph_amd", "libvgmi.so")
HOSTLIB = os.path.join(ROOT, "varigraph_amd", "libvghost.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


VGMI_SOURCES = ("vgmi_kernels.hip", "vgmi_xtable.hip", "vgmi_ctable.hip", "vgmi_ctdefer.hip", "vgmi_ptable.hip", "vgmi_fastq.hip", "vgmi_inflate.hip",
                "vgmi_gunzip.hip", "vgmi_bloom_bin.hip", "vgmi_hmm.hip", "vgmi_api.cpp", "vgmi_api_table.cpp", "vgmi_api_rccl.cpp",
                "vgmi_api_fastq.cpp", "vgmi_api_bloom.cpp", "vgmi_api_hmm.cpp")


def source_digest():
    """sha256 over the sources libvgmi.so is built from (names and bytes, in a fixed order): what a committed counter profile names, so
    that bench.py reports it as `roofline.traffic` for this code only."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in VGMI_SOURCES] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "vgmi.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def lib_digest():
    import hashlib
    return hashlib.sha256(open(LIB, "rb").read()).hexdigest() if os.path.exists(LIB) else None


def build_vgmi(force=False, verbose=False):
    """One object per source (in parallel, under build/obj; an object is rebuilt when its source or ANY header of csrc/ is newer --
    vg_x80.h defines the HMM kernels' arithmetic: editing it must rebuild the library), then the link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, f) for f in VGMI_SOURCES]
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "vgmi.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    if os.environ.get("VGMI_ABLATION") == "1":   # profiling builds only: the VGMI_DBG ablations (wrong results on purpose)
        flags.append("-DVGMI_ABLATION")
    flags += os.environ.get("VGMI_HIPCC_DEFS", "").split()   # A/B builds of compile-time choices (e.g. -DINFW_WHOLE=0)
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")
    # a library newer than every source and header, built with these flags, is up to date WHATEVER the object directory holds: build/ does
    # not travel to the GPU box, the library does -- a fresh box must not spend a minute compiling what it was sent
    custom = os.environ.get("VGMI_ABLATION") == "1" or bool(os.environ.get("VGMI_HIPCC_DEFS", "").split())
    stamp_ok = (open(stamp).read() == " ".join(flags)) if os.path.exists(stamp) else not custom
    if not force and stamp_ok and not _newer(LIB, srcs + hdrs):
        return LIB
    if not stamp_ok:
        force = True
    todo, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            todo.append([_hipcc(), *flags, "-c", src, "-o", obj])
    if not todo and not _newer(LIB, objs):
        return LIB

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=ROOT)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(run, todo))
    open(stamp, "w").write(" ".join(flags))
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


SYNTHLIB = os.path.join(ROOT, "varigraph_amd", "libvgsynth.so")


def build_synth(force=False, verbose=False):
    """libvgsynth.so: bench / test tooling (csrc/bench/vgsynth.h) -- the seeded synthetic workloads.  Not part of the product library."""
    bdir = os.path.join(CSRC, "bench")
    src = os.path.join(bdir, "vgsynth.hip")
    deps = [src, os.path.join(bdir, "vgsynth.h"), os.path.join(bdir, "vg_synth.h"), os.path.join(CSRC, "vgmi_device.h")]
    if not force and not _newer(SYNTHLIB, deps):
        return SYNTHLIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function", src, "-o", SYNTHLIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return SYNTHLIB


def build_host(force=False, verbose=False):
    hdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hdir):
        return None
    srcs = sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".cpp") and not f.startswith("main_"))
    if not srcs:
        return None
    deps = srcs + [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".hpp") or f.endswith(".h")] + [
        os.path.join(ROOT, "include", "vgmi.h"), os.path.join(ROOT, "include", "vghost.h")]
    deps = [d for d in deps if os.path.exists(d)]
    if not force and not _newer(HOSTLIB, deps + [LIB]):
        return HOSTLIB
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "include"), *srcs,
           "-o", HOSTLIB, "-L", os.path.dirname(LIB), "-lvgmi", "-Wl,-rpath,$ORIGIN", "-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return HOSTLIB


CLI = os.path.join(os.path.dirname(LIB), "bin", "varigraph-mi")


def build_cli(force=False, verbose=False):
    """`varigraph-mi genotype`: the reference's genotype sub-command on this build (csrc/host/main_genotype.cpp)."""
    src = os.path.join(CSRC, "host", "main_genotype.cpp")
    if not os.path.exists(src):
        return None
    if not force and not _newer(CLI, [src, HOSTLIB, LIB]):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O3", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", CLI,
           "-L", os.path.dirname(LIB), "-lvghost", "-lvgmi", "-Wl,-rpath,$ORIGIN/..", "-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return CLI


def build_oracle(force=False, with_ref=None):
    """Compile the checker. Building it is not using it (only tests/smoke/cpu_baseline call it)."""
    args = ["make", "-s", "-C", ORACLE_DIR, "all"]
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference/src")
    if with_ref:
        args.append("ref")
    if force:
        args.insert(1, "-B")
    subprocess.run(args, check=True)
    return ORACLE_LIB


def build_all(force=False, verbose=False):
    build_vgmi(force, verbose)
    build_synth(force, verbose)
    build_host(force, verbose)
    build_cli(force, verbose)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
