"""Builds libxpt_hip.so (all gfx950 kernels + the C ABI of include/xpt_hip.h) in-tree with hipcc."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libxpt_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(os.path.dirname(PKG), "include", "xpt_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


CODEGEN_FLAGS = ["-O3", "-fno-slp-vectorize", f"--offload-arch={ARCH}", "-std=c++17"]     # also used by tests/test_pipelined_isa.py


def build(force=False, verbose=True, extra_flags=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libxpt_hip.so")
    if not force and not needs_build():
        return OUT
    cmd = [hipcc, *CODEGEN_FLAGS, "-shared", "-fPIC", "-Wall", "-Wno-unused-function", "-o", OUT] + list(extra_flags) + sources()
    if verbose:
        print("[xpt build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
