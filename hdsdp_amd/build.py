"""Build the HIP extension (gfx950) in-tree.

    python -m hdsdp_amd.build                  # hdsdp_amd/libhdsdp_mi355x.so  (the product: tested code paths only)
    python -m hdsdp_amd.build --diagnostics    # hdsdp_amd/libhdsdp_mi355x_diag.so: the same sources with -DHDM_DIAGNOSTICS
                                               # (stamped kernels, wrong-result timing ablations, A/B bodies selected by
                                               # HDM_VAR / HDM_CONG2_DIRECT).  Measurement tools load it through
                                               # HDSDP_MI355X_LIB; the package, the tests and bench.py never do.

hipcc cross-compiles without a GPU.  The built .so files are git-ignored but travel to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libhdsdp_mi355x.so")
SOURCES = ["gemm_f64.hip", "gemm_persist.hip", "chol.hip", "schur.hip", "lanczos.hip", "lu.hip", "small.hip", "bsparse.hip", "engine.hip", "probes.hip", "coeff.cpp", "sdpa.cpp", "alloc.cpp"]
# every header under csrc/ is a dependency of every object (a header missing from a hand-kept list once left a stale
# library in place after an edit)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "hdsdp_mi355x.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, diagnostics=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    lib = LIB.replace(".so", "_diag.so") if diagnostics else LIB
    if not force and not _stale(lib, srcs + hdrs):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(CSRC, "build_diag" if diagnostics else "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if not force and not _stale(o, [s] + hdrs):
            continue
        # -fvisibility=hidden: the library exports what include/hdsdp_mi355x.h declares (default visibility there) and nothing else
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-x", "hip", "-c", s, "-o", o]
        if diagnostics:
            cmd.insert(1, "-DHDM_DIAGNOSTICS")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    # RCCL (the in-process device group's transport, csrc/group_impl.h) is linked, not dlopen'ed: the library's collectives
    # are part of the product, and a missing librccl should fail at load time, not in the middle of a solve
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, diagnostics="--diagnostics" in sys.argv)
