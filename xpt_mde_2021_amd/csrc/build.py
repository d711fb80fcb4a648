"""Builds libxpt_hip.so (all gfx950 kernels + the C ABI of include/xpt_hip.h) in-tree with hipcc.

Every .hip file is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked
into the shared library: a one-file edit rebuilds in seconds instead of minutes."""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libxpt_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
# the same sources once more with IEEE-half activations (xpt_common.h, XPT_HALF_F16): BASELINE configs[4] "fp16 convs"
OUT_F16 = os.path.join(PKG, "libxpt_hip_f16.so")
OBJ_DIR_F16 = os.path.join(HERE, "build_f16")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _headers():
    return glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(os.path.dirname(PKG), "include", "xpt_hip.h")]


def needs_build(out=None):
    out = out or OUT
    if not os.path.isfile(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


CODEGEN_FLAGS = ["-O3", "-fno-slp-vectorize", f"--offload-arch={ARCH}", "-std=c++17"]     # also used by tests/test_pipelined_isa.py


def _object_of(src, obj_dir=None):
    return os.path.join(obj_dir or OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def build(force=False, verbose=True, extra_flags=()):
    """Both libraries: libxpt_hip.so (bfloat16 activations) and libxpt_hip_f16.so (IEEE half)."""
    _build_one(OUT, OBJ_DIR, force, verbose, tuple(extra_flags))
    _build_one(OUT_F16, OBJ_DIR_F16, force, verbose, ("-DXPT_HALF_F16", *extra_flags))
    return OUT


def _build_one(OUT, OBJ_DIR, force, verbose, extra_flags):
    def _object_of(src):
        return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libxpt_hip.so")
    if not force and not needs_build(OUT):
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())
    flags = [*CODEGEN_FLAGS, "-fPIC", "-Wall", "-Wno-unused-function", *extra_flags]
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    same_flags = os.path.isfile(stamp) and open(stamp).read() == " ".join(flags)
    jobs = []
    for src in sources():
        obj = _object_of(src)
        if (force or not same_flags or not os.path.isfile(obj)
                or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header)):
            jobs.append([hipcc, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[xpt build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    for stale in set(glob.glob(os.path.join(OBJ_DIR, "*.o"))) - {_object_of(s) for s in sources()}:
        os.remove(stale)
    run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT, *[_object_of(s) for s in sources()]])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
