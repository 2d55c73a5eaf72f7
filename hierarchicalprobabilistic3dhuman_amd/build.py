"""Build recipe for libhps.so (hand-written HIP for gfx950, C ABI declared in include/hps.h).

hipcc cross-compiles without a GPU; the shared object is written next to this file so that it
travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_PATH = os.path.join(PKG_DIR, "libhps.so")
STAMP_PATH = os.path.join(PKG_DIR, "libhps.stamp")

SOURCES = ["api.hip", "smpl.hip", "blend_gemm.hip", "mf_sample.hip", "head.hip", "conv.hip", "conv_pad.hip", "composite.hip", "host_svd.hip", "frontend.hip", "metrics.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _source_digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "hps_common.h"),
                                                         os.path.join(INCLUDE, "hps.h")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    with open(STAMP_PATH) as f:
        return f.read().strip() == _source_digest()


def build(force=False, verbose=True):
    """Compile every translation unit for gfx950 and link libhps.so. Returns the library path."""
    if not force and is_current():
        return LIB_PATH
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP_PATH, "w") as f:
        f.write(_source_digest())
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
