"""Builds libmdgrad_hip.so in-tree with hipcc for gfx950 (no CMake, no torch extension
machinery: the library has a plain C ABI and does not link against torch)."""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmdgrad_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# The SLP vectoriser pairs adjacent scalar f32 multiply-adds into v_pk_fma_f32 plus the v_mov's that line their operands up; on
# gfx950 a packed f32 instruction costs 1.5-1.8 x a scalar one (profiles/r05_valu_rate.txt), so beside MFMAs the pairing is a loss:
# the stashed cfconv sweeps 55.9 -> 50.3 us and no spill, the stacked SchNet pass + 2 % (profiles/r06_noslp_ab.txt).  The ring /
# workgroup trajectory kernels keep it (their packed forms are written by hand, and the headline measured 0.6 % lower without).
NO_SLP = ["-fno-slp-vectorize"]
SLP_SOURCES = {"traj_small.hip"}


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    m = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > m for p in [src] + deps)


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    deps = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    if _newer(src, obj, deps):
        extra = [] if os.path.basename(src) in SLP_SOURCES else NO_SLP
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj, True, r.stderr
    return obj, False, ""


def build_library(verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources under %s" % CSRC)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _, _ in res]
    rebuilt = any(c for _, c, _ in res)
    if verbose:
        for (o, c, err) in res:
            if c and err.strip():
                print(err, file=sys.stderr)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("libmdgrad_hip.so: %s (%d sources, %s)" % (LIB, len(srcs), "rebuilt" if rebuilt else "up to date"))
    return LIB


TORCH_SRC = os.path.join(HERE, "csrc_torch", "mdgrad_torch.cpp")
TORCH_LIB = os.path.join(LIBDIR, "libmdgrad_torch.so")


def build_torch_ops(verbose=True):
    """torch.ops.mdgrad.*: the TORCH_LIBRARY layer over the C ABI (csrc_torch/mdgrad_torch.cpp), host-only C++ built
    with g++ against this interpreter's torch; links libmdgrad_hip.so through $ORIGIN."""
    import torch
    from torch.utils import cpp_extension as ce
    deps = [os.path.join(HERE, "..", "include", "mdgrad_hip.h"), LIB]
    if not _newer(TORCH_SRC, TORCH_LIB, deps):
        if verbose:
            print("libmdgrad_torch.so: %s (up to date)" % TORCH_LIB)
        return TORCH_LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
           + ["-I" + d for d in ce.include_paths()] + ["-I/opt/rocm/include", TORCH_SRC, "-o", TORCH_LIB,
              "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-L" + LIBDIR, "-lmdgrad_hip",
              "-Wl,-rpath,$ORIGIN"])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the torch op library failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("libmdgrad_torch.so: %s (rebuilt)" % TORCH_LIB)
    return TORCH_LIB


if __name__ == "__main__":
    build_library()
    build_torch_ops()
