"""Build recipe for libhps.so (hand-written HIP for gfx950, C ABI declared in include/hps.h) and for
libhps_dev.so, the same sources plus csrc/conv.hip compiled with -DHPS_DEV_BUILD (include/hps_dev.h: earlier kernel
generations kept as bit-level cross-checks, alternate variants, tuning switches, profiling ablations).  The product
library contains the product path only; only tests/ and tests/dev/ load the dev library.

hipcc cross-compiles without a GPU; the shared objects are written next to this file so that they
travel with the source tree (git-ignored, not gpurun-ignored).
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
DEV_LIB_PATH = os.path.join(PKG_DIR, "libhps_dev.so")

SOURCES = ["api.hip", "smpl.hip", "blend_gemm.hip", "mesh_fused.hip", "mesh_split.hip", "mf_sample.hip", "head.hip", "conv_pad.hip", "conv_wino.hip", "stem_wino.hip", "composite.hip",
           "host_svd.hip", "frontend.hip", "metrics.hip"]
DEV_ONLY_SOURCES = ["conv.hip"]
# per-file flags.  mesh_fused.hip: hipcc's SLP vectoriser turns the skinning epilogue into v_pk_fma_f32 plus one v_mov
# per packed operand (525 moves, 2 051 instructions); unpacked it is 1 963 instructions with 109 moves, and packed fp32
# VALU next to MFMAs is slower on gfx950 (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
# frontend.hip: the row-marching Canny kernel holds its windows in ~250 registers; SLP-packed pairs add alignment moves and push it
# into AGPR spills (310 registers, 6 126 VALU instructions per six steps against 260 / 5 864 unpacked).
# frontend.hip, max-ilp scheduling: the default (max-occupancy) strategy at ~250 registers emits the Gaussian / Sobel sums as runs of
# dependent FMAs into one accumulator; a wave alone on its SIMD issues a dependent v_fma_f32 every 8.25 cycles and an independent
# one every 5.0 (tools/valu_dep_probe.hip; two waves per SIMD together: one per 2.5).  Interleaved chains: edge map 0.031 -> 0.029 ms.  (Tried on stem_wino.hip and mesh_fused.hip:
# no change -- 0.603 / 0.574 ms; conv_wino.hip goes to scratch with it.)
FILE_FLAGS = {"mesh_fused.hip": ["-fno-slp-vectorize"], "mesh_split.hip": ["-fno-slp-vectorize"],
              "frontend.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _config(dev):
    if dev:
        return DEV_LIB_PATH, SOURCES + DEV_ONLY_SOURCES, FLAGS + ["-DHPS_DEV_BUILD"], "build_dev"
    return LIB_PATH, SOURCES, FLAGS, "build"


def _source_digest(dev=False):
    _, sources, flags, _ = _config(dev)
    h = hashlib.sha256()
    headers = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))          # every header under csrc/ (svd3_gesdd.h, ...)
    files = [os.path.join(CSRC, s) for s in sources] + [os.path.join(CSRC, h) for h in headers] + [
        os.path.join(INCLUDE, "hps.h"), os.path.join(INCLUDE, "hps_dev.h")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(flags).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    with open(os.path.abspath(__file__), "rb") as fh:       # the recipe itself (link line, version script)
        h.update(fh.read())
    return h.hexdigest()


def is_current(dev=False):
    lib = _config(dev)[0]
    stamp = lib.replace(".so", ".stamp")
    if not (os.path.exists(lib) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _source_digest(dev)


def build(force=False, verbose=True, dev=False):
    """Compile every translation unit for gfx950 and link libhps.so (dev=True: libhps_dev.so). Returns the library path."""
    lib_path, sources, flags, objname = _config(dev)
    if not force and is_current(dev):
        return lib_path
    objdir = os.path.join(PKG_DIR, objname)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in sources:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + flags + FILE_FLAGS.get(src, []) + ["-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
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
    # Dynamic symbol table = the C ABI and nothing else.  -fvisibility=hidden takes care of the C++ helpers and the kernels' host
    # stubs; hipcc still gives every __global__ kernel's host-side handle (and its per-TU __hip_cuid_* marker) default visibility,
    # so a linker version script makes everything but hps_* local.
    vscript = os.path.join(objdir, "exports.map")
    with open(vscript, "w") as f:
        f.write("{ global: hps_*; local: *; };\n")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vscript, "-o", lib_path] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(lib_path.replace(".so", ".stamp"), "w") as f:
        f.write(_source_digest(dev))
    return lib_path


TOOLS_DIR = os.path.join(REPO_ROOT, "tools")


def build_tools(force=False, verbose=True):
    """The stand-alone gfx950 microbenchmarks under tools/ (mfma_peak: the sustained fp32 MFMA rate on constant / random data;
    mfma_valu_overlap: does fp32 VALU work overlap fp32 MFMA) -> tools/bin/.  Their outputs are what DESIGN.md section 4's
    "sustained ceiling" and "VALU is additive" statements rest on (tools/collect_ablations.sh runs them on the GPU box)."""
    out_dir = os.path.join(TOOLS_DIR, "bin")
    os.makedirs(out_dir, exist_ok=True)
    built = []
    for name in sorted(f for f in os.listdir(TOOLS_DIR) if f.endswith(".hip")):
        src, exe = os.path.join(TOOLS_DIR, name), os.path.join(out_dir, name[:-4])
        if not force and os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
            built.append(exe)
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-I", INCLUDE, "-I", CSRC, src, "-o", exe]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        built.append(exe)
    return built


def build_all(force=False, verbose=True):
    """Product library first, then the dev library; the microbenchmarks under tools/ are best effort (a probe that does not
    compile on some ROCm version -- inline asm, LDS-DMA builtins -- must never block libhps.so)."""
    paths = build(force, verbose, dev=False), build(force, verbose, dev=True)
    try:
        build_tools(force, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        print("[build] WARNING: tools/ microbenchmarks not built (%s); libhps.so / libhps_dev.so are unaffected" % e, flush=True)
    return paths


if __name__ == "__main__":
    for path in build_all(force="--force" in sys.argv):
        print(path)
