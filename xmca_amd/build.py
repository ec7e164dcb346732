"""Builds libxmca_hip.so (gfx950) in-tree with hipcc.  No GPU is needed to build.

    python -m xmca_amd.build            # rebuild if sources are newer than the library
    python -m xmca_amd.build --force
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
REPO = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libxmca_hip.so")
SOURCES = ["xmca_hip.cpp"]
HEADERS = ["chol64.h", "cholesky.h", "comm.h", "common.h", "fft.h", "gemm.h", "jacobi.h", "jacobi_impl.inc", "kernels.h", "rotate.h", "solver.h", "tridiag.h", "tridiag_vec.h"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(REPO, "include", "xmca_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-x", "hip", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
           "-I", os.path.join(REPO, "include")]
    cmd += os.environ.get("XMCA_EXTRA_CXXFLAGS", "").split()      # (experiments: -D switches of a variant build)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-ldl", "-o", LIB + ".tmp"]          # (dlopen: RCCL is bound at run time, csrc/comm.h)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
