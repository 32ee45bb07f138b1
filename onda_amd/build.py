"""Build libonda_hip.so (all HIP kernels + the C ABI of include/onda_hip.h) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so stays
in-tree (git-ignored) and travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libonda_hip.so")
SOURCES = ["conv.hip", "conv_bf3.hip", "conv_h2.hip", "conv_l2.hip", "norm.hip", "norm_l2.hip", "pointwise.hip", "loss_proto.hip", "pipeline.hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(INCLUDE, "onda_hip.h")]
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + INCLUDE, "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
