"""Build libonda_hip.so (all HIP kernels + the C ABI of include/onda_hip.h) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so stays in-tree (git-ignored) and
travels to the GPU box with the snapshot.  A rebuild is decided by CONTENT, not by mtime: the sha256 of every source
and header is compiled into the library (``onda_version()`` ends in ``src=<hash>``) and compared with the sources on
disk, so a stale library -- or one built from other sources -- is rebuilt here and detected anywhere else
(``check_fresh``; bench.py prints both hashes).
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libonda_hip.so")
SOURCES = ["conv.hip", "conv_h2.hip", "conv_l2.hip", "norm.hip", "norm_l2.hip", "pointwise.hip", "loss_proto.hip",
           "pipeline.hip", "switch.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(INCLUDE, "onda_hip.h")]


def source_hash():
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def built_hash():
    """The source hash compiled into the library on disk (None: no library, or one from before the hash existed)."""
    if not os.path.exists(LIB):
        return None
    m = re.search(rb"onda_hip [0-9.]+ \(gfx950\) src=([0-9a-f]{16})", open(LIB, "rb").read())
    return m.group(1).decode() if m else None


def check_fresh():
    """(library hash, source hash): equal when the library was compiled from the sources beside it."""
    return built_hash(), source_hash()


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    want = source_hash()
    if not force and built_hash() == want:
        return LIB
    objs = []
    stamp = os.path.join(CSRC, ".objhash")
    old = dict(l.split() for l in open(stamp).read().splitlines()) if os.path.exists(stamp) else {}
    new = {}
    hdr = hashlib.sha256(b"".join(open(h, "rb").read() for h in HEADERS)).hexdigest()[:16]
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        key = hashlib.sha256(open(path, "rb").read() + hdr.encode() + (want.encode() if src == "loss_proto.hip" else b"")).hexdigest()[:16]
        new[src] = key
        if force or old.get(src) != key or not os.path.exists(obj):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + INCLUDE, f'-DONDA_SRC_HASH="{want}"', "-c", path,
                   "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp, "w").write("".join(f"{k} {v}\n" for k, v in new.items()))
    assert built_hash() == want, "the library does not carry the hash of its sources"
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
